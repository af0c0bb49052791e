#!/usr/bin/env python
"""Winograd F(2x2, 4x4) for the discriminator's layer_4 (4x4, stride 1, 256 -> 512 channels), costed in a
numerics microbench ONLY: error growth of the transform pipeline under this code base's arithmetic (fp32 transforms, every
contraction as fp16 pairs under one power-of-two scale per operand, three of four partial products, fp32 accumulation)
against float64, next to the direct form under the same arithmetic.  CPU, numpy; no GPU kernel exists or is planned.
    python tools/micro/winograd_f2x2_4x4_numerics.py"""
import numpy as np

rng = np.random.default_rng(5)


def cook_toom(points, m, r):
  """1-D F(m, r) over the finite `points` + infinity: AT [m, a], G [a, r], BT [a, a], a = m + r - 1, float64."""
  a = m + r - 1
  p = np.array(points, dtype=np.float64)
  assert len(p) == a - 1
  AT = np.zeros((m, a)); G = np.zeros((a, r)); BT = np.zeros((a, a))
  for j in range(a - 1):
    for i in range(m):
      AT[i, j] = p[j] ** i
    n = np.prod([p[j] - p[l] for l in range(a - 1) if l != j])
    for k in range(r):
      G[j, k] = p[j] ** k / n
  AT[m - 1, a - 1] = 1.0
  G[a - 1, r - 1] = 1.0
  M = np.poly1d(np.poly(p))                       # prod (x - p_l)
  for j in range(a - 1):
    q, _ = np.polydiv(M, np.poly1d([1.0, -p[j]]))
    c = q.coeffs[::-1]                             # ascending powers, degree a - 2
    BT[j, :len(c)] = c
  c = M.coeffs[::-1]
  BT[a - 1, :len(c)] = c
  return AT, G, BT


def check(AT, G, BT, m, r):
  d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
  want = np.array([np.dot(d[i:i + r], g) for i in range(m)])
  got = AT @ ((G @ g) * (BT @ d))
  return np.abs(got - want).max()


def pair(x):
  """x -> the value the fp16 pair image holds: x 2^s = h0 + h1 (fp16, round to nearest), s from max |x| -> [2^13, 2^14)"""
  x = np.asarray(x, dtype=np.float32)
  amax = float(np.abs(x).max())
  s = 13 - int(np.floor(np.log2(amax))) if amax > 0 else 0
  y = x * np.float32(2.0 ** s)
  h0 = y.astype(np.float16)
  h1 = (y - h0.astype(np.float32)).astype(np.float16)
  return (h0.astype(np.float32) + h1.astype(np.float32)) * np.float32(2.0 ** -s)


def gemm_pairs(a, b):
  """[M, K] x [K, N] as the kernels do it: both operands as fp16 pairs under one scale each, fp32 accumulation (the dropped
  a1 b1 term is below the fp32 accumulation error and not modelled)"""
  return pair(a).astype(np.float32) @ pair(b).astype(np.float32)


def direct(x, w, dtype):
  """x [H, W, C], w [4, 4, C, K] -> valid correlation [H - 3, W - 3, K]"""
  H, W, C = x.shape
  out = np.zeros((H - 3, W - 3, w.shape[3]), dtype=dtype)
  for ky in range(4):
    for kx in range(4):
      out += (x[ky:ky + H - 3, kx:kx + W - 3].reshape(-1, C).astype(dtype) @ w[ky, kx].astype(dtype)).reshape(H - 3, W - 3, -1)
  return out


def direct_pairs(x, w):
  H, W, C = x.shape
  xp, wp = pair(x), pair(w)
  out = np.zeros((H - 3, W - 3, w.shape[3]), dtype=np.float32)
  for ky in range(4):
    for kx in range(4):
      out += (xp[ky:ky + H - 3, kx:kx + W - 3].reshape(-1, C) @ wp[ky, kx]).reshape(H - 3, W - 3, -1)
  return out


def winograd_pairs(x, w, AT, G, BT):
  """fp32 transforms, 25 contractions over channels as fp16 pairs (one scale per transformed operand), fp32 output transform"""
  H, W, C = x.shape
  K = w.shape[3]
  th, tw = (H - 3) // 2, (W - 3) // 2
  AT32, G32, BT32 = AT.astype(np.float32), G.astype(np.float32), BT.astype(np.float32)
  U = np.einsum('ia,abck,jb->ijck', G32, w.astype(np.float32), G32).astype(np.float32)          # [5, 5, C, K]
  tiles = np.stack([x[2 * ty:2 * ty + 5, 2 * tx:2 * tx + 5] for ty in range(th) for tx in range(tw)]).astype(np.float32)
  V = np.einsum('ia,tabc,jb->tijc', BT32, tiles, BT32).astype(np.float32)                          # [T, 5, 5, C]
  M = np.zeros((tiles.shape[0], 5, 5, K), dtype=np.float32)
  for i in range(5):
    for j in range(5):
      M[:, i, j] = gemm_pairs(V[:, i, j], U[i, j])
  Y = np.einsum('ia,tabk,jb->tijk', AT32, M, AT32).astype(np.float32)                              # [T, 2, 2, K]
  out = np.zeros((2 * th, 2 * tw, K), dtype=np.float32)
  t = 0
  for ty in range(th):
    for tx in range(tw):
      out[2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y[t]
      t += 1
  return out


def rel(a, b):
  a, b = a.astype(np.float64), b.astype(np.float64)
  return float(np.linalg.norm(a - b) / np.linalg.norm(b))


if __name__ == '__main__':
  H = W = 19                   # 16 x 16 outputs = 8 x 8 tiles
  C, K = 256, 64
  datasets = {
      'gaussian activations': rng.standard_normal((H, W, C)),
      'log-normal activations (sigma = 3), leaky ReLU': None,
      'spiky output gradients (1e-6 with 1e3 x outliers)': None,
  }
  ln = np.exp(3.0 * rng.standard_normal((H, W, C))) * np.sign(rng.standard_normal((H, W, C)))
  datasets['log-normal activations (sigma = 3), leaky ReLU'] = np.where(ln > 0, ln, 0.2 * ln)
  sp = 1e-6 * rng.standard_normal((H, W, C))
  sp[rng.random((H, W, C)) < 1e-3] *= 1e3
  datasets['spiky output gradients (1e-6 with 1e3 x outliers)'] = sp
  w = (0.05 * rng.standard_normal((4, 4, C, K)))
  print('| point set (+ infinity) | exactness of F(2,4) in float64 | data | direct, fp16 pairs | Winograd F(2x2,4x4), fp16 pairs | Winograd, float32 everywhere |')
  print('|---|---|---|---|---|---|')
  for pts in ([0, 1, -1, 2], [0, 1, -1, 0.5], [0, 0.5, -0.5, 1]):
    AT, G, BT = cook_toom(pts, 2, 4)
    ex = check(AT, G, BT, 2, 4)
    for name, x in datasets.items():
      ref = direct(x, w, np.float64)[:16, :16]
      d_p = rel(direct_pairs(x, w)[:16, :16], ref)
      w_p = rel(winograd_pairs(x, w, AT, G, BT), ref)
      # float32 everywhere: the same pipeline with plain float32 contractions
      _pair = pair
      globals()['pair'] = lambda v: np.asarray(v, dtype=np.float32)
      w_f = rel(winograd_pairs(x, w, AT, G, BT), ref)
      globals()['pair'] = _pair
      print('| %s | %.1e | %s | %.2e | %.2e | %.2e |' % (pts, ex, name, d_p, w_p, w_f))
