#!/usr/bin/env python
"""Where the LWS time goes: the time-ordered (causal) pass vs the 100 batch sweeps, for 64 clips of 256 frames, on uniform
noise magnitudes (every tile active from sweep ~40 on) and on a real speech spectrogram tiled to the batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from advoc_amd import _lib, audioio, spectral as S
lib = _lib.load()
fs, wav = audioio.decode_audio(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'mono.wav'), fastwav=True)
speech = S.stft_magnitude(wav[None], 1024, 256)[0, :, :, 0]
for clips in (64, 256):
  for name in ('noise', 'speech'):
    if name == 'noise':
      mag = torch.rand(clips, 256, 513, device='cuda') + 0.01
    else:
      mag = torch.stack([speech[(37 * i) % (speech.shape[0] - 256):][:256] for i in range(clips)]).contiguous()
    for _ in range(2):
      S.lws_spectrogram_batch(mag, 1024, 256)
    torch.cuda.synchronize()
    def timed(fn, n=5):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(n):
        fn()
      e1.record(); torch.cuda.synchronize()
      return e0.elapsed_time(e1) / n
    full = timed(lambda: S.lws_spectrogram_batch(mag, 1024, 256))
    nob = timed(lambda: S.lws_spectrogram_batch(mag, 1024, 256, batch=(0, 1.0, 0.0, 1.0)))
    wavs = timed(lambda: S.lws_batch(mag, 1024, 256))
    print('%3d clips %-6s: spectrogram %.2f ms (time-ordered pass %.2f ms, 100 sweeps %.2f ms), with istft %.2f ms' % (
        clips, name, full, nob, full - nob, wavs), flush=True)
