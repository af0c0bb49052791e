cd $GRAFT_REPO_ROOT
for abl in 0 2 3 4; do
  for v in "1 2"; do set -- $v
    echo "ABLATE=$abl tile=$1 stages=$2"
    export ADVOC_H3_ABLATE=$abl
    ADVOC_H3_TILE=$1 ADVOC_H3_STAGES=$2 python - <<'PY'
import os, sys
sys.path.insert(0, 'tools/micro')
import torch
from h3_sweep_shapes import build
from advoc_amd import _lib
def t(fn, reps=10):
  for _ in range(2): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3
for name in ('d4', 'enc4', 'dec4'):
  L, dy, dx0, dx1 = build(name)
  L.forward()
  os.environ['ADVOC_H3_SKIP_PREP'] = '1'; _lib.reload_env()
  us = t(L.forward)
  os.environ.pop('ADVOC_H3_SKIP_PREP'); _lib.reload_env()
  print('  %-5s %-36s %8.1f us %6.1f TF' % (name, L.kernel_name(0), us, L.flops / us / 1e6), flush=True)
PY
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/h3_ablate.txt
unset ADVOC_H3_ABLATE
ADVOC_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 10 --warmup 2 --train-only --no-cpu-baseline > gpurun_out/b2_h3.json 2> gpurun_out/b2_h3.err; head -12 gpurun_out/b2_h3.err; tail -2 gpurun_out/b2_h3.err; head -c 400 gpurun_out/b2_h3.json
