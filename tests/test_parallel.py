"""Data-parallel path on CPU: world_size-2 gloo processes run advoc_amd.parallel.DataParallel
(the same code the GPU ranks run over RCCL) and check the one contract the reference implies:
an N-rank step on a sharded global batch == the single-rank step on the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.set_num_threads(2)
  from advoc_amd.parallel import DataParallel
  from oracle import advoc_torch as A
  dp = DataParallel(bucket_bytes=1 << 16).init_from_env(backend='gloo')
  assert dp.enabled and dp.world_size == world and dp.rank == rank

  cfg = A.Config(small=True, subseq_len=32)
  P = A.init_params(cfg, seed=0)
  B = 4
  g = torch.Generator().manual_seed(11)
  target = torch.rand(B, 32, 513, 1, generator=g)
  x = target * 0.8 + 0.05
  masks = A.make_dropout_masks(cfg, B, seed=2)
  lo, hi = rank * B // world, (rank + 1) * B // world
  local_masks = {k: v[lo:hi] for k, v in masks.items()}
  grads, _ = A.grads(P, x[lo:hi], target[lo:hi], cfg, local_masks, 'G')
  flat = torch.cat([v.reshape(-1) for v in grads.values()])
  assert flat.numel() * 4 > 4 * dp.bucket_elems            # several buckets
  # the overlapped path: ranges of the arena reduced asynchronously as the "backward pass" completes them
  flat2 = flat.clone()
  n = flat2.numel()
  cuts = [0, n // 5, n // 2, n]
  for a, b in zip(cuts[:-1], cuts[1:]):
    dp.reduce_range_async(flat2, a, b)
  dp.finish_reductions()
  dp.allreduce_(flat)
  assert torch.equal(flat, flat2)
  flat /= world                                             # what the fused Adam's grad_scale applies
  if rank == 0:
    full, _ = A.grads(P, x, target, cfg, masks, 'G')
    ref = torch.cat([v.reshape(-1) for v in full.values()])
    err = float((flat - ref).norm() / ref.norm())
    torch.save(dict(err=err, shard=dp.shard(list(range(10))), mx=None), os.path.join(out_dir, 'r0.pt'))
  mx = dp.max_over_ranks(1.0 + rank)
  assert mx == float(world)
  assert dp.shard(list(range(10))) == list(range(10))[rank::world]
  dp.barrier()


def test_two_rank_gradient_average_equals_full_batch(tmp_path):
  port = _free_port()
  mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  r = torch.load(os.path.join(str(tmp_path), 'r0.pt'))
  assert r['err'] < 1e-5, r['err']
  assert r['shard'] == [0, 2, 4, 6, 8]


def test_single_process_is_a_noop():
  from advoc_amd.parallel import DataParallel
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
    os.environ.pop(k, None)
  dp = DataParallel()
  assert not dp.enabled and dp.world_size == 1
  t = torch.arange(5.)
  assert dp.allreduce_(t) is t and dp.max_over_ranks(3.5) == 3.5
  assert dp.shard([1, 2, 3]) == [1, 2, 3]
