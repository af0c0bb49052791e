# per-kernel launch times of the train step under two settings of one environment variable:
#   bash tools/micro/kern_ab.sh NAME value_a value_b
name=$1
i=0
for v in $2 $3; do
  env $name=$v python bench.py --train-only --no-cpu-baseline --steps 40 2>/tmp/err.txt > /tmp/k_$i.json || tail -3 /tmp/err.txt
  i=$((i+1))
done
python - <<PY
import json
a = json.load(open('/tmp/k_0.json')); b = json.load(open('/tmp/k_1.json'))
print('$name', '$2', round(a['ms_per_step'], 3), '$3', round(b['ms_per_step'], 3))
ka = {k['kernel']: k for k in a['roofline']['kernels']}; kb = {k['kernel']: k for k in b['roofline']['kernels']}
for n in sorted(set(ka) | set(kb), key=lambda n: -(ka.get(n) or kb.get(n))['share_of_conv_stack']):
  x, y = ka.get(n), kb.get(n)
  f = lambda k: '%8.4f ms x%-3d %5.1f%%' % (k['avg_launch_ms'], k.get('launches_per_step', 0), 100 * k['share_of_conv_stack']) if k else ' ' * 24
  print('%-58s %s | %s' % (n[:58], f(x), f(y)))
PY
