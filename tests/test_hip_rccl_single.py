"""The RCCL transport itself on ONE GPU: a single-rank process group (all-reduce = identity) run through the
same code path as N ranks (ADVOC_DP_FORCE=1) -- asynchronous bucketed all-reduces issued from the weight-gradient
side stream, the synchronous discriminator reduce, broadcast, barrier.  The N-rank numerics are covered by
tests/test_hip_parallel.py (gloo transport); this one checks that the RCCL calls, their streams and the joins
work, and that a step with them equals a step without."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpu = pytest.mark.gpu

SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, %r)
from advoc_amd.parallel import DataParallel
from advoc_amd.model import AdvocSmall, Modes
force = os.environ.get('ADVOC_DP_FORCE') == '1'
dp = DataParallel(bucket_bytes=1 << 20).init_from_env()
assert dp.enabled == force, (dp.enabled, force)
if force:
  assert dp.backend == 'nccl', dp.backend
m = AdvocSmall(Modes.TRAIN)
m.subseq_len = 64
m.train_batch_size = 4
m.build(batch_size=4, seed=5)
dp.attach(m)
dp.broadcast_parameters(m)
g = torch.Generator().manual_seed(3)
t = torch.rand(4, 64, 513, 1, generator=g) * 2
x = t * (0.5 + torch.rand(4, 64, 513, 1, generator=g)) - 0.1
dev = torch.device('cuda', dp.local_rank)
m((x.to(dev), t.to(dev)))
for _ in range(3):
  m.train_loop()
dp.barrier()
torch.cuda.synchronize()
ls = m.losses()
print('LOSSES %%.6f %%.6f %%.6f' %% (ls['gen_loss_L1'], ls['disc_loss'], ls['gen_loss_GAN']))
'''


def _run(force):
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
             HSA_ENABLE_IPC_MODE_LEGACY='0')
  env.pop('ADVOC_DP_BACKEND', None)
  if force:
    env['ADVOC_DP_FORCE'] = '1'
  else:
    env.pop('ADVOC_DP_FORCE', None)
  out = subprocess.run([sys.executable, '-c', SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  line = [l for l in out.stdout.splitlines() if l.startswith('LOSSES')][-1]
  return [float(v) for v in line.split()[1:]]


@gpu
def test_single_rank_rccl_step_equals_plain_step(hip):
  with_rccl = _run(True)
  plain = _run(False)
  for a, b in zip(with_rccl, plain):
    assert abs(a - b) <= 1e-3 * abs(b) + 1e-5, (with_rccl, plain)
