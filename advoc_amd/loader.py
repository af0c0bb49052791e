"""Audio files -> fixed-length (features, audio) batches in HBM.  Drop-in for the reference's
``advoc.loader.decode_extract_and_batch`` (/root/reference/advoc/loader.py:8-218): same 21
parameters, defaults, error behaviour and slicing semantics.

What is different by construction
  * the reference builds a tf.data graph and returns two lazy tensors bound to a one-shot
    iterator; this returns a ``BatchPipeline`` whose ``next()`` yields the same
    ``(features, audio)`` pair (also iterable, and unpackable into two per-field views so
    ``x_feats, x_audio = decode_extract_and_batch(...)`` keeps working: the k-th ``next()`` of either
    view returns its field of the SAME k-th batch, as one ``sess.run`` on both reference tensors does);
  * host threads only decode WAVs and plan slices (integer arithmetic); the STFT / mel features
    of each file are computed on the GPU by the HIP kernels (advoc_amd.spectral) and sliced,
    shuffled and batched as device tensors -- features never cross PCIe.

Semantics kept (reference line numbers): per-epoch shuffle of all file paths (:69-70), repeat
(:73-74), features ``None | 'magspec' | 'melspec'`` (:99-130), ``slice_hop =
round(slice_len * (1 - overlap))`` (:137), audio slices of ``slice_len * fs / feature_fs`` samples
(:144-148), optional random start offset in [0, slice_len) frames (:155-161), zero ``pad_end``
framing of both streams (:165-178; tf.contrib.signal.frame: ceil(N / hop) frames when padding,
1 + (N - L) // hop otherwise), ``slice_first_only`` (:180-182), example shuffle buffer
(:199-200), ``batch(drop_remainder=True)`` (:203).  ``ValueError`` for an unknown
``extract_type`` (:130), a negative overlap (:136) and an overlap that leaves no hop (:139).
"""
import collections
import queue
import threading

import numpy as np
import torch

from advoc_amd import _lib
from advoc_amd.audioio import decode_audio


def frame_count(n, length, hop, pad_end):
  """Number of frames tf.contrib.signal.frame produces for a signal of n steps."""
  if pad_end:
    return -(-n // hop)
  return max(0, 1 + (n - length) // hop)


def slice_geometry(slice_len, audio_fs, feature_fs, slice_overlap_ratio):
  """(slice_hop, audio_slice_len, audio_slice_hop) exactly as loader.py:135-148 computes them."""
  if slice_overlap_ratio < 0:
    raise ValueError('Slice overlap must be nonnegative')
  slice_hop = int(round(slice_len * (1. - slice_overlap_ratio)))
  if slice_hop < 1:
    raise ValueError('Overlap ratio too high')
  nsamps_per_tstep = float(audio_fs) / float(feature_fs)
  audio_slice_len = int(round(slice_len * nsamps_per_tstep) + 1e-4)
  audio_slice_hop = int(round(slice_hop * nsamps_per_tstep) + 1e-4)
  return slice_hop, audio_slice_len, audio_slice_hop


def plan_slices(ntsteps, nsamps, slice_len, slice_hop, audio_slice_len, audio_slice_hop,
                nsamps_per_tstep, pad_end, first_only, start=0):
  """Integer slice plan for one file: (feature start rows, audio start samples, count).

  `start` is the random frame offset (0 when slice_randomize_offset is off); the audio offset is
  round(start * nsamps_per_tstep + 1e-4) (loader.py:157-161).  The two streams are zipped, so the
  count is the smaller of the two frame counts (tf.data.Dataset.zip, loader.py:190-196)."""
  start_audio = int(np.round(np.float32(start) * np.float32(nsamps_per_tstep) + np.float32(1e-4)))
  nt = max(ntsteps - start, 0)
  ns = max(nsamps - start_audio, 0)
  count = min(frame_count(nt, slice_len, slice_hop, pad_end),
              frame_count(ns, audio_slice_len, audio_slice_hop, pad_end))
  if first_only:
    count = min(count, 1)
  return start, start_audio, count


def _frames(x, start, count, length, hop):
  """[N, ...] device tensor -> [count, length, ...] zero-padded frames beginning at `start`."""
  if count == 0:
    return x.new_zeros((0, length) + tuple(x.shape[1:]))
  need = start + (count - 1) * hop + length
  if need > x.shape[0]:
    pad = x.new_zeros((need - x.shape[0],) + tuple(x.shape[1:]))
    x = torch.cat([x, pad], dim=0)
  win = x[start:need].unfold(0, length, hop)            # [count, ..., length]
  return win.movedim(-1, 1).contiguous()


class _FieldView(object):
  """One field of the pipeline's output.  The reference returns both tensors from ONE iterator `get_next`
  (loader.py:209-216), so a `sess.run([x_feats, x_audio])` fetches a PAIRED batch: the two views therefore share
  the current batch -- a view's `next()` returns its field of the current batch if it has not read that batch yet,
  otherwise it pulls a new one (which the other view then reads too).  `f = feats.next(); a = audio.next()` is a
  pair; a loop over one view alone gets a fresh batch per call, like repeated `sess.run` of one tensor."""

  def __init__(self, pipe, index):
    self._pipe, self._index = pipe, index
    self._seen = -1          # serial number of the last batch this view returned

  def next(self):
    serial, batch = self._pipe._view_batch(self._seen)
    self._seen = serial
    return batch[self._index]

  __next__ = next

  def __iter__(self):
    return self


class BatchPipeline(object):
  def __init__(self, fps, batch_size, slice_len, audio_fs, audio_mono, audio_normalize, decode_fastwav,
               decode_parallel_calls, extract_type, extract_nfft, extract_nhop, repeat, shuffle,
               shuffle_buffer_size, slice_first_only, slice_randomize_offset, slice_overlap_ratio,
               slice_pad_end, prefetch_size, seed=None):
    if extract_type not in (None, 'melspec', 'magspec'):
      raise ValueError()
    self.fps = list(fps)
    self.batch_size = int(batch_size)
    self.slice_len = int(slice_len)
    self.audio_fs = audio_fs
    self.audio_mono = audio_mono
    self.audio_normalize = bool(audio_normalize)
    self.decode_fastwav = bool(decode_fastwav)
    self.extract_type = extract_type
    self.nfft, self.nhop = int(extract_nfft), int(extract_nhop)
    self.repeat, self.shuffle = bool(repeat), bool(shuffle)
    self.shuffle_buffer_size = shuffle_buffer_size
    self.first_only = bool(slice_first_only)
    self.randomize = bool(slice_randomize_offset)
    self.pad_end = bool(slice_pad_end)
    feature_fs = audio_fs if extract_type is None else audio_fs / self.nhop
    self.nsamps_per_tstep = float(audio_fs) / float(feature_fs)
    self.slice_hop, self.audio_slice_len, self.audio_slice_hop = slice_geometry(
        self.slice_len, audio_fs, feature_fs, slice_overlap_ratio)
    # two independent streams: the producer thread draws the per-epoch file order, the consumer thread the
    # slice offsets and shuffle-buffer slots -- one shared generator would interleave non-deterministically
    self.rng = np.random.RandomState(seed)
    self._order_rng = np.random.RandomState(None if seed is None else (int(seed) * 2654435761 + 97) % (2 ** 32))
    self._view_cur = None       # (serial, batch) the field views share
    self._view_next = 0
    self._examples = self._example_stream()
    self._buffer = []
    self._exhausted = False
    # decoded-file prefetch: host threads run scipy decode, nothing else
    self._nthreads = max(1, int(decode_parallel_calls))
    depth = max(2 * self._nthreads, 4)
    if prefetch_size:
      depth = max(depth, min(int(prefetch_size), 64))
    self._decoded = queue.Queue(maxsize=depth)
    self._stop = threading.Event()
    self._producer = threading.Thread(target=self._produce, daemon=True)
    self._producer.start()

  # ---- host side: file order + decode ----
  def _file_order(self):
    while True:
      order = list(range(len(self.fps)))
      if self.shuffle:
        self._order_rng.shuffle(order)
      for i in order:
        yield self.fps[i]
      if not self.repeat:
        return

  def _decode(self, fp):
    return decode_audio(fp, fs=self.audio_fs, mono=self.audio_mono, normalize=self.audio_normalize,
                        fastwav=self.decode_fastwav)[1]

  def _produce(self):
    from concurrent.futures import ThreadPoolExecutor
    pending = collections.deque()
    try:
      with ThreadPoolExecutor(self._nthreads) as pool:
        for fp in self._file_order():
          if self._stop.is_set():
            return
          pending.append(pool.submit(self._decode, fp))
          while len(pending) >= 2 * self._nthreads:
            self._put(pending.popleft())
        while pending:
          self._put(pending.popleft())
    finally:
      # end-of-stream marker; after close() nobody reads the queue any more: never block on it
      while True:
        try:
          self._decoded.put(None, timeout=0.1)
          break
        except queue.Full:
          if self._stop.is_set():
            break

  def _put(self, fut):
    try:
      item = fut.result()
    except Exception as e:   # surfaced on the consumer thread
      item = e
    while not self._stop.is_set():
      try:
        self._decoded.put(item, timeout=0.1)
        return
      except queue.Full:
        continue

  # ---- device side: features + slices ----
  def _features(self, wav_dev):
    from advoc_amd import spectral
    if self.extract_type is None:
      return wav_dev
    if self.extract_type == 'magspec':
      return spectral.stft_magnitude(wav_dev[None], self.nfft, self.nhop)[0]
    return spectral.waveform_to_melspec_tf(wav_dev[None], fs=self.audio_fs, nfft=self.nfft,
                                           nhop=self.nhop)[0]

  def _example_stream(self):
    dev = _lib.device()
    while True:
      item = self._decoded.get()
      if item is None:
        return
      if isinstance(item, Exception):
        raise item
      wav = torch.from_numpy(np.ascontiguousarray(item)).to(dev, non_blocking=True)   # [n,1,ch]
      feats = self._features(wav)
      start = int(self.rng.randint(0, self.slice_len)) if self.randomize else 0
      fstart, astart, count = plan_slices(
          feats.shape[0], wav.shape[0], self.slice_len, self.slice_hop, self.audio_slice_len,
          self.audio_slice_hop, self.nsamps_per_tstep, self.pad_end, self.first_only, start)
      if count == 0:
        continue
      f = _frames(feats, fstart, count, self.slice_len, self.slice_hop)
      a = _frames(wav, astart, count, self.audio_slice_len, self.audio_slice_hop)
      for i in range(count):
        yield f[i], a[i]

  def _next_example(self):
    """tf.data shuffle-buffer semantics: fill the buffer, then swap a random slot per draw."""
    if not self.shuffle or not self.shuffle_buffer_size:
      return next(self._examples)
    while not self._exhausted and len(self._buffer) < self.shuffle_buffer_size:
      try:
        self._buffer.append(next(self._examples))
      except StopIteration:
        self._exhausted = True
    if not self._buffer:
      raise StopIteration
    j = int(self.rng.randint(0, len(self._buffer)))
    ex = self._buffer[j]
    self._buffer[j] = self._buffer[-1]
    self._buffer.pop()
    return ex

  def next(self):
    """(features [b, slice_len, nfeats, nch], audio [b, audio_slice_len, 1, nch]) float32 in HBM.
    Raises StopIteration at the end of a non-repeating dataset (incomplete batch dropped)."""
    feats, audio = [], []
    for _ in range(self.batch_size):
      f, a = self._next_example()      # StopIteration propagates: drop_remainder=True
      feats.append(f)
      audio.append(a)
    return torch.stack(feats), torch.stack(audio)

  __next__ = next

  def _view_batch(self, seen):
    """(serial, batch) for a field view that last returned batch `seen`: the current batch if it is newer, else a
    freshly pulled one."""
    if self._view_cur is None or self._view_cur[0] <= seen:
      self._view_cur = (self._view_next, self.next())
      self._view_next += 1
    return self._view_cur

  def __iter__(self):
    # unpacking `feats, audio = pipeline` yields the two field views
    return iter((_FieldView(self, 0), _FieldView(self, 1)))

  def batches(self):
    while True:
      try:
        yield self.next()
      except StopIteration:
        return

  def close(self):
    self._stop.set()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


def decode_extract_and_batch(
    fps,
    batch_size,
    slice_len,
    audio_fs=22050,
    audio_mono=True,
    audio_normalize=False,
    decode_fastwav=False,
    decode_parallel_calls=1,
    extract_type=None,
    extract_nfft=1024,
    extract_nhop=256,
    extract_parallel_calls=1,
    repeat=False,
    shuffle=False,
    shuffle_buffer_size=None,
    slice_first_only=False,
    slice_randomize_offset=False,
    slice_overlap_ratio=0,
    slice_pad_end=False,
    prefetch_size=None,
    prefetch_gpu_num=None):
  """Decodes audio files directly into [b, slice_len, nfeats, nch] batches on the GPU.

  Args: identical to the reference (loader.py:30-59).  `extract_parallel_calls` and
  `prefetch_gpu_num` are accepted for compatibility: feature extraction is one GPU kernel per
  file on the current HIP device, so there is nothing to parallelise on the host.

  Returns:
    A BatchPipeline; `.next()` -> (features, audio).  Unpacks into two field views.
  """
  return BatchPipeline(fps, batch_size, slice_len, audio_fs, audio_mono, audio_normalize,
                       decode_fastwav, decode_parallel_calls, extract_type, extract_nfft,
                       extract_nhop, repeat, shuffle, shuffle_buffer_size, slice_first_only,
                       slice_randomize_offset, slice_overlap_ratio, slice_pad_end, prefetch_size)
