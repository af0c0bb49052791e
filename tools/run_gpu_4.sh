cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3 4; do
  for v in "2 2"; do set -- $v
    echo "ABLATE=$abl tile=$1 stages=$2"
    export ADVOC_X6D_ABLATE=$abl
    ADVOC_X6D_TILE=$1 ADVOC_X6D_STAGES=$2 python - <<'PY'
import os, sys
sys.path.insert(0, 'tools/micro')
import torch
from x6d_sweep_shapes import build
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
  os.environ['ADVOC_X6D_SKIP_PREP'] = '1'; _lib.reload_env()
  us = t(L.forward)
  os.environ.pop('ADVOC_X6D_SKIP_PREP'); _lib.reload_env()
  print('  %-5s %-36s %8.1f us %6.1f TF' % (name, L.kernel_name(0), us, L.flops / us / 1e6), flush=True)
PY
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x6d_ablate.txt
