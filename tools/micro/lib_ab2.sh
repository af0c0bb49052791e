# A/B of library builds on the train step, per-kernel: bash tools/micro/lib_ab2.sh lib1.so lib2.so ...   (paths relative to the repo)
i=0
for lib in "$@"; do
  ADVOC_HIP_LIB=$PWD/$lib python bench.py --train-only --no-cpu-baseline --steps 40 2>/tmp/err.txt > /tmp/ab_$i.json || tail -3 /tmp/err.txt
  i=$((i+1))
done
python - "$@" <<'PY'
import json, sys
libs = sys.argv[1:]
rs = [json.load(open('/tmp/ab_%d.json' % i)) for i in range(len(libs))]
print('ms_per_step: ' + '  '.join('%s %.3f' % (l.split('/')[-1], r['ms_per_step']) for l, r in zip(libs, rs)))
ks = [{k['kernel']: k for k in r['roofline']['kernels']} for r in rs]
names = sorted(set().union(*[set(k) for k in ks]), key=lambda n: -max(k[n]['share_of_conv_stack'] for k in ks if n in k))
for n in names[:22]:
  print('%-52s %s' % (n[:52], '  '.join(('%8.4f ms x%-4.1f' % (k[n]['avg_launch_ms'], k[n].get('launches_per_step', 0))) if n in k else ' ' * 20 for k in ks)))
PY
