# A/B of one environment knob on the train step: bash tools/micro/env_ab.sh NAME value_a value_b [repeats]
name=$1; a=$2; b=$3; n=${4:-2}
for i in $(seq $n); do
for v in $a $b; do
  t=$(echo $v | tr '/' '_')
  env $name=$v python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null > /tmp/b_$t.json
  python - <<PY
import json
r = json.load(open('/tmp/b_$t.json'))
ks = {k['kernel']: k for k in r['roofline']['kernels']}
w = ks.get('wgrad_h3_256_kernel') or ks.get('wgrad_h3_256s_kernel') or {}
print('$name', '$v', 'ms_per_step', round(r['ms_per_step'], 3), 'wgrad256 avg ms', w.get('avg_launch_ms'), 'frac', w.get('frac'))
PY
done
done
