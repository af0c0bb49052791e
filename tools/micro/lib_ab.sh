# A/B of library builds on the train step: bash tools/micro/lib_ab.sh lib1.so lib2.so ...   (paths relative to the repo)
for lib in "$@"; do
  ADVOC_HIP_LIB=$PWD/$lib python bench.py --train-only --no-cpu-baseline --steps 40 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
  python - <<PY
import json
r = json.load(open('/tmp/b.json'))
ks = {k['kernel']: k for k in r['roofline']['kernels']}
w = ks.get('wgrad_h3_256_kernel') or {}
w2 = ks.get('wgrad_h3_kernel') or {}
g = [(k['kernel'], round(k['avg_launch_ms'], 4)) for k in r['roofline']['kernels'] if 'gather_gemm_h3' in k['kernel']]
print('$lib', g, 'ms_per_step', round(r['ms_per_step'], 3), 'wgrad256 ms', round(w.get('avg_launch_ms', 0), 4), 'frac', round(w.get('frac', 0), 3), 'wgrad128 ms', round(w2.get('avg_launch_ms', 0), 4), 'D loss', r['losses']['disc_loss'])
PY
done
