# per-kernel table of the train step under several settings of environment variables:
#   bash tools/micro/env_ab2.sh "A=1 B=2" "A=3" ""        (each argument: one setting, space-separated VAR=value pairs)
i=0
for setting in "$@"; do
  env $setting python bench.py --train-only --no-cpu-baseline --steps 40 2>/tmp/err.txt > /tmp/ab_$i.json || tail -3 /tmp/err.txt
  i=$((i+1))
done
python - "$@" <<'PY'
import json, sys
sets = sys.argv[1:]
rs = [json.load(open('/tmp/ab_%d.json' % i)) for i in range(len(sets))]
for st, r in zip(sets, rs):
  print('%-60s ms_per_step %.3f' % (st or '(default)', r['ms_per_step']))
ks = [{k['kernel']: k for k in r['roofline']['kernels']} for r in rs]
names = sorted(set().union(*[set(k) for k in ks]), key=lambda n: -max(k[n]['share_of_conv_stack'] for k in ks if n in k))
for n in names[:int(__import__('os').environ.get('ROWS', '14'))]:
  print('%-44s %s' % (n[:44], ' '.join(('%8.4f' % k[n]['avg_launch_ms']) if n in k else ' ' * 8 for k in ks)))
PY
