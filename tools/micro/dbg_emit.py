import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ.update(ADVOC_H3_MIN_TILES='1', ADVOC_H3_PATCH_MIN_WGS='1', ADVOC_WGRAD_H3_MIN_M='1', ADVOC_H3_PATCH=sys.argv[1])
import torch
from advoc_amd import conv
dev = torch.device('cuda')
g = torch.Generator().manual_seed(31)
xin1 = torch.randn(2, 36, 36, 64, generator=g).to(dev)
xin2 = torch.randn(2, 9, 9, 64, generator=g).to(dev)
w1 = (torch.randn(4, 4, 64, 128, generator=g) * 0.05).to(dev)
w2 = (torch.randn(4, 4, 128, 64, generator=g) * 0.05).to(dev)
wc1 = (torch.randn(4, 4, 128, 256, generator=g) * 0.05).to(dev)
wc2 = (torch.randn(4, 4, 64, 256, generator=g) * 0.05).to(dev)
b1 = (torch.randn(128, generator=g) * 0.1).to(dev)
mask = (torch.rand(2, 18, 18, 128, generator=g) >= 0.5).to(torch.uint8).to(dev)
def make(register):
  y1 = torch.empty(2, 18, 18, 128, device=dev)
  y2 = torch.empty(2, 18, 18, 128, device=dev)
  P1 = conv.Layer(conv.CONV, xin1.clone(), y1, w1, b1, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
  P2 = conv.Layer(conv.DECONV, xin2.clone(), y2, w2, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_RELU, drop_mask=mask, drop_scale=2.0)
  C1 = conv.Layer(conv.CONV, y1, torch.empty(2, 9, 9, 256, device=dev), wc1, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
  C2 = conv.Layer(conv.DECONV, y2, torch.empty(2, 36, 36, 64, device=dev), wc2, None, x1=y1, in_w=18, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_RELU)
  for L in (P1, P2, C1, C2):
    L.delayed_scale, L.reuse_images = True, True
  if register:
    P1.add_image_consumer(C1, 0); P1.add_image_consumer(C2, 1); P2.add_image_consumer(C2, 0)
  return P1, P2, C1, C2
A, R = make(True), make(False)
print([L.kernel_name(0) for L in A])
for step, scale in enumerate((1.0, 0.8, 1.3)):
  for Ls in (A, R):
    Ls[0].x0.copy_(xin1 * scale); Ls[1].x0.copy_(xin2 * scale)
    for L in Ls: L.forward()
  for ci in (2, 3):
    a, r = A[ci]._img[0], R[ci]._img[0]
    d = (a != r).nonzero().flatten()
    print('step', step, 'consumer', ci - 1, 'diff', d.numel(), 'of', a.numel(), 'hdr', A[ci]._img[1].cpu().tolist()[:6], R[ci]._img[1].cpu().tolist()[:6], 'y equal', torch.equal(A[ci].y, R[ci].y))
    if d.numel():
      print('   first', [(i, int(a[i]), int(r[i])) for i in d[:8].tolist()], 'last', d[-3:].tolist())
      half = a.numel() // 2
      print('   in source0 region:', int((d < half).sum()), ' source1 region:', int((d >= half).sum()))
