"""Full-model layer shapes shared by the h3 micro-benchmarks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import conv
dev = torch.device('cuda')
SHAPES = {
    # name: (kind, B, H, W, c0, c1, cout, stride, trim, act)
    'enc4':  (0, 64, 32, 65, 256, 0, 512, (2, 2), 0, 1),
    'enc3':  (0, 64, 64, 129, 128, 0, 256, (2, 2), 0, 1),
    'enc2':  (0, 64, 128, 257, 64, 0, 128, (2, 2), 0, 1),
    'd4':    (0, 128, 32, 64, 256, 0, 512, (1, 1), 0, 1),
    'dec4':  (1, 64, 16, 33, 512, 512, 256, (2, 2), 1, 2),
    'dec3':  (1, 64, 32, 65, 256, 256, 128, (2, 2), 1, 2),
    # model geometry (AdVoc-full, B=64; D on 2B): the stride-1 gathers igemm_patch.hip takes
    'dec2m': (1, 64, 64, 128, 128, 128, 64, (2, 2), 1, 2),      # decoder_2 forward: 64 output channels
    'dec3m': (1, 64, 32, 64, 256, 256, 128, (2, 2), 1, 2),
    'dec4m': (1, 64, 16, 32, 512, 512, 256, (2, 2), 1, 2),
    'enc2m': (0, 64, 128, 256, 64, 0, 128, (2, 2), 0, 1),       # backward-data: 64 columns, phases over a 64 x 128 grid
    'enc3m': (0, 64, 64, 128, 128, 0, 256, (2, 2), 0, 1),
    'enc4m': (0, 64, 32, 64, 256, 0, 512, (2, 2), 0, 1),
    'd2m':   (0, 128, 128, 256, 64, 0, 128, (2, 2), 0, 1),
    'd3m':   (0, 128, 64, 128, 128, 0, 256, (2, 2), 0, 1),
    'd4b':   (0, 64, 32, 64, 256, 0, 512, (1, 1), 0, 1),        # layer_4 on the B-clip pass of the G step
    # the deep layers (few grid points, 16-33 MB of weights): split-K launches
    'enc5m': (0, 64, 16, 33, 512, 0, 512, (2, 2), 0, 1),
    'enc6m': (0, 64, 8, 17, 512, 0, 512, (2, 2), 0, 1),
    'dec5m': (1, 64, 8, 17, 512, 512, 512, (2, 2), 1, 2),
    'dec6m': (1, 64, 4, 9, 512, 512, 512, (2, 2), 1, 2),
    'enc4o': (0, 64, 32, 65, 256, 0, 512, (2, 2), 0, 1),        # encoder_4 at the model's odd width
    'enc7m': (0, 64, 4, 9, 512, 0, 512, (2, 2), 0, 1),
    'enc8m': (0, 64, 2, 5, 512, 0, 512, (2, 2), 0, 1),
    'dec7m': (1, 64, 2, 5, 512, 512, 512, (2, 2), 1, 2),
    'dec8m': (1, 64, 1, 3, 512, 0, 512, (2, 2), 0, 2),
}


def build(name):
  kind, B, H, W, c0, c1, cout, stride, trim, act = SHAPES[name]
  x0 = torch.randn(B, H, W + trim, c0, device=dev)
  x1 = torch.randn(B, H, W, c1, device=dev) if c1 else None
  if kind == 0:
    if stride == (1, 1):
      oh, ow = H - 1, W - 1
    else:
      oh, ow = -(-H // 2), -(-W // 2)
    w = torch.randn(4, 4, c0 + c1, cout, device=dev) * 0.05
  else:
    oh, ow = 2 * H, 2 * W
    w = torch.randn(4, 4, cout, c0 + c1, device=dev) * 0.05
  y = torch.empty(B, oh, ow, cout, device=dev)
  L = conv.Layer(kind, x0, y, w, None, x1=x1, in_w=W, stride=stride, pad=(1, 1), in_act=act)
  dy = torch.randn_like(y)
  dx0 = torch.empty_like(x0)
  dx1 = torch.empty_like(x1) if x1 is not None else None
  return L, dy, dx0, dx1




def setenv(**kw):
  from advoc_amd import _lib
  for k, v in kw.items():
    if v is None:
      os.environ.pop(k, None)
    else:
      os.environ[k] = str(v)
  _lib.reload_env()


def timed_us(fn, reps=10):
  for _ in range(2):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3
