"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no multi-GPU code; this is new functionality with one
contract: an N-GPU step on a global batch equals the 1-GPU step on the same global batch.
Each rank owns batch/N clips; after each backward pass the network's flat gradient arena
(advoc_amd.model) is summed across ranks with a few large all-reduces -- xGMI is
point-to-point, ring all-reduce is per-link bound, so buckets are big (default 8 MiB) -- and the
fused Adam kernel applies 1/N.  The generator's arena is laid out in backward-completion order
and its buckets are reduced asynchronously as soon as the backward pass has filled them, next to
the remaining backward kernels; the (small) discriminator arena is reduced in one go, asynchronously too: its sum runs
under the generator forward of the G step that follows and the discriminator's Adam step is applied when that pass is
through (model._flush_d_adam).  Dropout masks are Philox streams keyed by the GLOBAL clip index,
so sharding does not change them, and with use_batchnorm=True the batch statistics are summed over the
replicas (advoc_bn_*_stats / _finalize / _apply) -- synchronised batch norm.
"""
import os

import torch
import torch.distributed as dist


class DataParallel(object):
  def __init__(self, bucket_bytes=None):
    # bucket_bytes: size of one all-reduce (default 8 MiB).  xGMI is point-to-point: a ring all-reduce is bound per
    # link, so buckets are few and large; 8 MiB = 27 collectives for the generator's 217.6 MB arena
    if bucket_bytes is None:
      bucket_bytes = 8 << 20
    self.bucket_elems = max(1, bucket_bytes // 4)
    self.world_size = 1
    self.rank = 0
    self.local_rank = 0
    self.enabled = False
    self.backend = None
    self.reserve_source = 'none'   # who set ADVOC_RESERVE_CUS for the persistent launches: 'user' | 'dp' | 'none'
    self._pending = []

  def init_from_env(self, backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run)."""
    self.world_size = int(os.environ.get('WORLD_SIZE', '1'))
    self.rank = int(os.environ.get('RANK', '0'))
    self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # ADVOC_DP_BACKEND=gloo / ADVOC_DP_DEVICE=<index>: run the N-rank code path on a box with fewer
    # GPUs than ranks (wiring check only -- RCCL refuses two ranks on one device, so collectives go
    # through host memory there)
    backend = backend or os.environ.get('ADVOC_DP_BACKEND')
    if 'ADVOC_DP_DEVICE' in os.environ:
      self.local_rank = int(os.environ['ADVOC_DP_DEVICE'])
    # ADVOC_DP_FORCE=1: take the N-rank code path even with ONE rank (RCCL all-reduce over a single rank is
    # the identity): exercises the real RCCL calls, streams and bucket logic on a 1-GPU box
    if self.world_size > 1 or os.environ.get('ADVOC_DP_FORCE') == '1':
      os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
      if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
      if backend == 'nccl':
        torch.cuda.set_device(self.local_rank)
      if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
      self.backend = dist.get_backend()
      self.enabled = True
      # ADVOC_DP_RESERVE_CUS=k (default 0; multiples of 8): the persistent launches (patch kernels and image weight
      # gradient: one 110-160 KB-LDS workgroup per CU for the whole launch) fill CUs - k instead of every CU, so that
      # RCCL's kernels become co-resident DURING those launches instead of between them.  Only honoured here, i.e. when
      # there is more than one rank; unmeasured on hardware (no multi-GPU box was available to the builder): k = 8 costs
      # the persistent kernels 3 % of the chip
      # (r5) an ADVOC_RESERVE_CUS the user set explicitly wins (it used to be overwritten); bench.py passes 8
      # for N > 1 unless told otherwise.  Still experimental: no multi-GPU box has measured it.
      k = os.environ.get('ADVOC_DP_RESERVE_CUS')
      if 'ADVOC_RESERVE_CUS' in os.environ:
        self.reserve_source = 'user'
      elif k is not None:
        self.reserve_source = 'dp'
        os.environ['ADVOC_RESERVE_CUS'] = str(max(0, int(k)))
        from advoc_amd import _lib
        if torch.cuda.is_available():
          _lib.reload_env()
    elif torch.cuda.is_available():
      torch.cuda.set_device(self.local_rank)
    return self

  def allreduce_(self, flat):
    """Sum a flat fp32 tensor across ranks, in place, in large buckets."""
    if not self.enabled:
      return flat
    n = flat.numel()
    for lo in range(0, n, self.bucket_elems):
      self._collective(dist.all_reduce, flat[lo:min(n, lo + self.bucket_elems)], op=dist.ReduceOp.SUM)
    return flat

  # ---- overlapped reduction: ranges of a flat gradient arena are reduced as soon as the backward
  # pass has produced them (the arena is laid out in backward-completion order) ----
  def reduce_range_async(self, flat, lo, hi):
    """Starts the sum of flat[lo:hi] across ranks.  RCCL: asynchronous on its own stream -- it waits
    for the work already enqueued on the compute stream (the kernels that wrote this range) and runs
    next to the kernels enqueued afterwards.  Completion is joined by `finish_reductions`."""
    if not self.enabled or hi <= lo:
      return
    n = hi
    for a in range(lo, n, self.bucket_elems):
      piece = flat[a:min(n, a + self.bucket_elems)]
      if self.backend == 'gloo' and piece.is_cuda:
        self._collective(dist.all_reduce, piece, op=dist.ReduceOp.SUM)      # host-staged, synchronous
      else:
        self._pending.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))

  def finish_reductions(self):
    """Makes the current stream (RCCL) / the host (gloo) wait for every reduction started above."""
    for w in self._pending:
      w.wait()
    self._pending = []

  def _collective(self, fn, t, **kw):
    """RCCL works on device tensors directly; gloo (CPU tests, wiring checks) goes through host memory."""
    if self.backend == 'gloo' and t.is_cuda:
      host = t.cpu()
      fn(host, **kw)
      t.copy_(host)
    else:
      fn(t, **kw)

  def sum_small_(self, t):
    """In-place cross-rank sum of a small tensor (batch-norm statistics: 2*c float64 values)."""
    if self.enabled:
      self._collective(dist.all_reduce, t, op=dist.ReduceOp.SUM)
    return t

  def attach(self, model):
    """Makes `model` (advoc_amd.model.Advoc) average gradients across ranks before Adam."""
    model._world_size = self.world_size
    model._rank = self.rank
    model._allreduce = self.allreduce_ if self.enabled else None
    # generator gradients: reduce arena ranges while the rest of the backward pass still runs
    model._reduce_async = (self.reduce_range_async, self.finish_reductions, self.bucket_elems) if self.enabled else None
    # batch norm (use_batchnorm=True): statistics over the global batch, one small float64 all-reduce per
    # normalised tensor and direction, so that N replicas still equal one device on the global batch
    model._sync_bn = self.sum_small_ if self.enabled else None
    return model

  def broadcast_parameters(self, model):
    """Rank 0's parameters and optimiser slots to everyone (identical start)."""
    if not self.enabled:
      return
    model._flush_d_adam()
    st = model._built
    for k in ('g_param', 'd_param', 'g_m', 'g_v', 'd_m', 'd_v'):
      self._collective(dist.broadcast, st[k], src=0)
    model.parameters_changed()

  def barrier(self):
    if self.enabled:
      if self.backend == 'nccl':
        dist.barrier(device_ids=[self.local_rank])
      else:
        dist.barrier()

  def max_over_ranks(self, value):
    if not self.enabled:
      return value
    dev = torch.device('cuda', self.local_rank) if self.backend == 'nccl' else torch.device('cpu')
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  def shard(self, items):
    """This rank's strided share of a list (files, chunks)."""
    return items[self.rank::self.world_size]
