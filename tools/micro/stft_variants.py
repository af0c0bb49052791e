#!/usr/bin/env python
"""stft1024_hop256_kernel variants (ADVOC_STFT_V, ADVOC_STFT_BLOCKS; -1 = stft1024_kernel) at 512 clips and at the
training feed (128 clips): one process per setting (the library reads the switches once)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, bench
from advoc_amd import spectral, _lib
lib = _lib.load()
wav = bench.synth_waveforms(64, 1, torch.device('cuda'))
win = spectral._device_window(1024, 256); tw = spectral._device_twiddle(1024)
res = {}
for clips in (512, 128):
  x = wav[:, :, 0, 0].repeat((clips + 63) // 64, 1)[:clips].contiguous()
  out = torch.empty(clips, bench.CLIP_FRAMES, 513, dtype=torch.float32, device=x.device)
  call = lambda: _lib.check(lib.advoc_stft_mag_f32(_lib.ptr(x), clips, x.shape[1], _lib.ptr(win), _lib.ptr(tw), 1024, 256,
                                                   bench.CLIP_FRAMES, _lib.ptr(out), _lib.stream()), 'stft')
  ms = min(bench.event_timed(torch, call, 50) for _ in range(3))
  by = clips * (bench.CLIP_SAMPLES * 4 + bench.CLIP_FRAMES * 513 * 4)
  res[clips] = (round(ms * 1e3, 1), round(by / (ms * 1e-3) / 1e9 / 8000, 3))
  if clips == 128:
    ref = torch.stft(x[:8].double(), 1024, 256, window=win.double(), center=False, return_complex=True).abs().transpose(1, 2)
    res['err'] = float((out[:8].double() - ref).norm() / ref.norm())
print(json.dumps(res))
''' % ROOT
settings = [(-1, 2048), (0, 2048), (1, 2048), (0, 1024), (1, 768), (1, 1536), (0, 4096), (1, 4096)]
if len(sys.argv) > 1:
  settings = [tuple(int(t) for t in a.split(',')) for a in sys.argv[1:]]
for st in settings:
  v, blocks = st[0], st[1]
  env = dict(os.environ, ADVOC_STFT_V=str(v), ADVOC_STFT_BLOCKS=str(blocks))
  if len(st) > 2:
    env.update(ADVOC_STFT_SKEW_WG=str(st[2]), ADVOC_STFT_SKEW_WAVE=str(st[3]))
    print('skew', st[2:], end=' ')
  r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
  print('V=%2d blocks=%4d  %s %s' % (v, blocks, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else '',
                                    r.stderr.strip()[-400:] if r.returncode else ''), flush=True)
