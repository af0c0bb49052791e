"""HIP-event timing of a 30-120 us kernel: one event pair per launch (what bench.event_timed did until r4) against one pair around
the same launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from advoc_amd import spectral, _lib
lib = _lib.load()
wav = bench.synth_waveforms(64, 1, torch.device('cuda'))
win = spectral._device_window(1024, 256); tw = spectral._device_twiddle(1024)
for clips in (512, 128):
  x = wav[:, :, 0, 0].repeat((clips + 63) // 64, 1)[:clips].contiguous()
  out = torch.empty(clips, 256, 513, dtype=torch.float32, device=x.device)
  call = lambda: _lib.check(lib.advoc_stft_mag_f32(_lib.ptr(x), clips, x.shape[1], _lib.ptr(win), _lib.ptr(tw), 1024, 256, 256, _lib.ptr(out), _lib.stream()), 'stft')
  def per_launch(n):
    for _ in range(3): call()
    evs = []
    for _ in range(n):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record(); call(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / n
  per = [per_launch(30) for _ in range(5)]
  tot = []
  for _ in range(5):
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): call()
    e1.record(); torch.cuda.synchronize()
    tot.append(e0.elapsed_time(e1) / 30)
  print(clips, 'per-launch events us', [round(v * 1e3, 1) for v in per], 'one pair around 30 launches', [round(v * 1e3, 1) for v in tot])
