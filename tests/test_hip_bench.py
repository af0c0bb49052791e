"""bench.py is the driver's contract: the plain `python bench.py --gpus N` form must launch its own ranks, and the JSON
line must carry the fields the driver and the judge read.  The 2-rank run shares this box's one GPU
(ADVOC_DP_BACKEND=gloo ADVOC_DP_DEVICE=0: host-staged collectives, a wiring check of the N-rank path -- RCCL refuses two
ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpu = pytest.mark.gpu


def _run(args, extra_env=None, timeout=900):
  env = dict(os.environ)
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  env.update(extra_env or {})
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  return json.loads(lines[0])


@gpu
def test_bench_line_single_gpu_small_and_quick(hip):
  r = _run(['--model', 'small', '--batch', '8', '--steps', '3', '--warmup', '1', '--prof-steps', '1',
            '--train-only', '--no-cpu-baseline'])
  for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
    assert key in r, key
  assert r['n_gpus'] == 1 and r['steps'] == 3 and r['warmup'] == 1 and r['higher_is_better'] is True
  assert r['dist'] is None
  assert r['unit'] == 'mel-frames/s' and r['vs_baseline'] is None and r['data'] == 'synthetic'
  assert abs(r['value'] - 8 * 256 * 3 / (r['ms_per_step'] * 3e-3)) < 1e-6 * r['value']
  assert 'workload' in r['config'] and 'model' not in r['config']
  roof = r['roofline']
  assert roof['bound'] in ('hbm', 'mfma') and 0 < roof['frac'] < 1 and roof['unit'] in ('GB/s', 'TFLOP/s')
  assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9


@gpu
def test_plain_python_bench_gpus_2_spawns_its_own_ranks(hip):
  r = _run(['--gpus', '2', '--model', 'small', '--batch', '4', '--steps', '2', '--warmup', '1', '--prof-steps', '0',
            '--train-only', '--no-cpu-baseline'],
           extra_env={'ADVOC_DP_BACKEND': 'gloo', 'ADVOC_DP_DEVICE': '0'})
  assert r['n_gpus'] == 2 and r['config']['global_batch'] == 8 and r['config']['parallelism'] == 'dp2'
  assert r['scaling'] == 'weak' and r['cpu_baseline'] is None
  assert abs(r['value'] - 8 * 256 * 2 / (r['ms_per_step'] * 2e-3)) < 1e-6 * r['value']
  assert abs(r['per_gpu_value'] * 2 - r['value']) < 1e-6 * r['value']
  # what the collectives ran on: so that a SCALE record shows the transport saw N ranks
  d = r['dist']
  assert d['backend'] == 'gloo' and d['world_size'] == 2 and sorted(x['rank'] for x in d['ranks']) == [0, 1]
  assert all(x['device'] == 'cuda:0' and x['name'] for x in d['ranks'])
  assert d['g_arena_bytes'] > 0 and d['g_arena_allreduce_ms'] > 0


@gpu
def test_bench_gpus_8_first_run_wiring_on_one_device(hip):
  """The driver may run `bench.py --gpus 8` on an 8-GPU node without anybody having run it there before: the same command
  with eight ranks sharing this box's one GPU (host-staged collectives) has to come back with eight ranks in `dist`, the CU
  reserve the N > 1 path picks for RCCL recorded, and IDENTICAL parameters on every rank after the optimizer steps."""
  r = _run(['--gpus', '8', '--model', 'small', '--batch', '2', '--steps', '2', '--warmup', '0', '--prof-steps', '0',
            '--train-only', '--no-cpu-baseline'],
           extra_env={'ADVOC_DP_BACKEND': 'gloo', 'ADVOC_DP_DEVICE': '0'}, timeout=1500)
  assert r['n_gpus'] == 8 and r['config']['global_batch'] == 16 and r['config']['parallelism'] == 'dp8'
  d = r['dist']
  assert d['world_size'] == 8 and sorted(x['rank'] for x in d['ranks']) == list(range(8))
  assert len(set(x['pid'] for x in d['ranks'])) == 8
  assert d['reserve_cus'] == 8 and 'bench.py default' in d['reserve_cus_source']
  assert d['params_equal_across_ranks'] is True and len(set(x['param_sha1_16'] for x in d['ranks'])) == 1
  # (r6) correctness beside the throughput: one global-batch step over the 8 ranks equals the single-process step on rank 0
  assert d['step_equals_single_gpu'] is True, d


@gpu
def test_an_explicit_cu_reserve_is_not_overridden(hip):
  r = _run(['--gpus', '2', '--model', 'small', '--batch', '2', '--steps', '1', '--warmup', '0', '--prof-steps', '0',
            '--train-only', '--no-cpu-baseline'],
           extra_env={'ADVOC_DP_BACKEND': 'gloo', 'ADVOC_DP_DEVICE': '0', 'ADVOC_RESERVE_CUS': '16'})
  assert r['dist']['reserve_cus'] == 16 and r['dist']['reserve_cus_source'] == 'ADVOC_RESERVE_CUS'
