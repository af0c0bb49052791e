import sys, os
sys.path.insert(0, os.getcwd())
import torch
from advoc_amd import model as M
m = M.AdvocSmall(M.Modes.TRAIN)
m.build(batch_size=32)
st = m._built
def show(tag, layers):
  it = layers.items() if isinstance(layers, dict) else enumerate(layers)
  for k, L in it:
    print(tag, k, [L.kernel_name(d) for d in (0, 1, 2)], tuple(L.x0.shape), tuple(L.y.shape))
show('G', st['g_layers'])
show('Dreal', st['d_layers'] if 'd_layers' in st else st['d_layers_fake'])
