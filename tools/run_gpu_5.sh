cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_lws.py tests/test_hip_inversion.py tests/test_infer.py tests/test_loader.py tests/test_cli.py -q -m gpu -x 2>&1 | tail -15
python - <<'PY'
import time, torch, numpy as np
from advoc_amd import spectral
mag = torch.rand(32, 256, 513, device='cuda')
for name, fn in (('lws', lambda: spectral.lws_batch(mag, 1024, 256)), ('gl60', lambda: spectral.griffin_lim_batch(mag, 1024, 256, 60, torch.rand_like(mag)))):
  fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); print(name, '%.2f ms / 32 clips' % ((time.perf_counter() - t0) * 1e3))
PY
