"""advoc_amd.loader: host-side slice planning against the numpy oracle (CPU), and the whole
pipeline -- threads, GPU features, slicing, batching -- against the oracle pipeline (GPU)."""
import os

import numpy as np
import pytest

from advoc_amd import loader as L
from advoc_amd.audioio import save_as_wav
from oracle import loader_np as O

gpu = pytest.mark.gpu


@pytest.mark.parametrize('pad_end', [True, False])
@pytest.mark.parametrize('n,length,hop', [(0, 4, 2), (1, 4, 2), (3, 4, 2), (4, 4, 2), (5, 4, 2), (9, 4, 3),
                                          (63, 256, 192), (600, 256, 192), (259, 256, 256), (1000, 256, 64)])
def test_frame_count_matches_explicit_framing(n, length, hop, pad_end):
  x = np.arange(n, dtype=np.float32)[:, None]
  assert L.frame_count(n, length, hop, pad_end) == O.frame(x, length, hop, pad_end).shape[0]


def test_slice_geometry_ljspeech_and_sc09():
  # datacfg/ljspeech.txt: overlap 0.25 -> hop 192 frames, 65536 / 49152 samples
  assert L.slice_geometry(256, 22050, 22050 / 256, 0.25) == (192, 65536, 49152)
  assert L.slice_geometry(256, 16000, 16000 / 256, 0.) == (256, 65536, 65536)
  assert L.slice_geometry(16384, 16000, 16000, 0.) == (16384, 16384, 16384)
  with pytest.raises(ValueError):
    L.slice_geometry(256, 22050, 22050 / 256, -0.1)
  with pytest.raises(ValueError):
    L.slice_geometry(256, 22050, 22050 / 256, 0.999)


@pytest.mark.parametrize('nsamps', [100, 16000, 65536, 65537, 145000, 220500])
@pytest.mark.parametrize('pad_end,first_only,overlap', [(True, False, 0.25), (False, False, 0.25),
                                                        (True, True, 0.), (False, False, 0.)])
def test_plan_matches_oracle_slicing(nsamps, pad_end, first_only, overlap):
  nt = -(-nsamps // 256)
  feats = np.arange(nt, dtype=np.float32)[:, None, None]
  audio = np.arange(nsamps, dtype=np.float32)[:, None, None]
  f_o, a_o = O.parallel_slice(feats, audio, 256, 22050, 22050 / 256, overlap, pad_end, first_only)
  hop, alen, ahop = L.slice_geometry(256, 22050, 22050 / 256, overlap)
  fs, as_, count = L.plan_slices(nt, nsamps, 256, hop, alen, ahop, 256.0, pad_end, first_only, 0)
  assert (fs, as_) == (0, 0) and count == f_o.shape[0] == a_o.shape[0]
  for i in range(count):
    assert f_o[i, 0, 0, 0] == fs + i * hop and a_o[i, 0, 0, 0] == as_ + i * ahop


def test_random_offset_audio_alignment():
  hop, alen, ahop = L.slice_geometry(256, 22050, 22050 / 256, 0.25)
  for start in (0, 1, 17, 255):
    fs, as_, count = L.plan_slices(600, 600 * 256, 256, hop, alen, ahop, 256.0, True, False, start)
    assert fs == start and as_ == start * 256
    assert count == min(-(-(600 - start) // hop), -(-(600 * 256 - as_) // ahop))


def test_constructor_errors_need_no_gpu():
  with pytest.raises(ValueError):
    L.decode_extract_and_batch([], 1, 256, extract_type='cqt')
  with pytest.raises(ValueError):
    L.decode_extract_and_batch([], 1, 256, slice_overlap_ratio=-1)
  with pytest.raises(ValueError):
    L.decode_extract_and_batch([], 1, 256, slice_overlap_ratio=1.0)


def _write_wavs(tmpdir, lengths, fs):
  rng = np.random.default_rng(0)
  fps = []
  for i, n in enumerate(lengths):
    t = np.arange(n) / fs
    x = 0.3 * np.sin(2 * np.pi * (200 + 150 * i) * t) + 0.1 * rng.uniform(-1, 1, n)
    fp = os.path.join(str(tmpdir), 'clip%02d.wav' % i)
    save_as_wav(fp, fs, x.astype(np.float32)[:, None, None])
    fps.append(fp)
  return fps


@gpu
@pytest.mark.parametrize('extract,overlap,pad_end,first_only,bs', [
    ('magspec', 0.25, True, False, 3), ('magspec', 0., False, False, 2), ('magspec', 0., True, True, 2),
    ('melspec', 0.25, True, False, 4), (None, 0., True, False, 2)])
def test_pipeline_matches_oracle(hip, tmp_path, extract, overlap, pad_end, first_only, bs):
  fps = _write_wavs(tmp_path, [70000, 16000, 150000, 66304, 30000], 22050)
  slice_len = 256 if extract else 16384
  want = O.batches(fps, bs, slice_len, audio_fs=22050, extract_type=extract, first_only=first_only,
                   overlap=overlap, pad_end=pad_end)
  pipe = L.decode_extract_and_batch(
      fps, bs, slice_len, audio_fs=22050, audio_mono=True, decode_fastwav=True, decode_parallel_calls=3,
      extract_type=extract, slice_first_only=first_only, slice_overlap_ratio=overlap,
      slice_pad_end=pad_end, prefetch_size=4)
  got = list(pipe.batches())
  assert len(got) == len(want) and len(got) > 0
  for (gf, ga), (wf, wa) in zip(got, want):
    assert gf.is_cuda and ga.is_cuda
    assert tuple(gf.shape) == wf.shape and tuple(ga.shape) == wa.shape
    assert np.array_equal(ga.cpu().numpy(), wa)                     # audio slices: bit exact
    g = gf.cpu().numpy()
    if extract is None:
      assert np.array_equal(g, wf)
    else:
      assert np.linalg.norm(g - wf) / np.linalg.norm(wf) < 1e-5


@gpu
def test_training_configuration_runs_and_shuffles(hip, tmp_path):
  """The train() configuration of train_evaluate.py:31-50: repeat + shuffle + random offsets."""
  fps = _write_wavs(tmp_path, [90000, 120000, 70000], 22050)
  pipe = L.decode_extract_and_batch(
      fps, batch_size=4, slice_len=256, audio_fs=22050, audio_mono=True, audio_normalize=False,
      decode_fastwav=True, decode_parallel_calls=4, extract_type='magspec', extract_parallel_calls=8,
      repeat=True, shuffle=True, shuffle_buffer_size=16, slice_first_only=False,
      slice_randomize_offset=True, slice_overlap_ratio=0.25, slice_pad_end=True, prefetch_size=32,
      prefetch_gpu_num=0)
  x_magspec, x_wav = pipe          # unpacks like the reference's two tensors
  seen = []
  for _ in range(6):
    f = x_magspec.next()
    assert tuple(f.shape) == (4, 256, 513, 1) and f.is_cuda and float(f.min()) >= 0
    seen.append(float(f.sum()))
  a = x_wav.next()
  assert tuple(a.shape) == (4, 65536, 1, 1)
  assert len(set(seen)) > 1
  pipe.close()


@gpu
def test_decode_errors_surface(hip, tmp_path):
  bad = os.path.join(str(tmp_path), 'bad.wav')
  open(bad, 'wb').write(b'not a wav')
  pipe = L.decode_extract_and_batch([bad], 1, 256, decode_fastwav=True, extract_type='magspec')
  with pytest.raises(ValueError):
    pipe.next()


def test_field_views_share_the_current_batch():
  """`x_feats, x_audio = decode_extract_and_batch(...)`: reading both views in turn yields PAIRED fields (one
  iterator get_next feeds both reference tensors, loader.py:209-216); a view read twice in a row pulls a new batch."""
  class Stub(object):
    _view_batch = L.BatchPipeline._view_batch

    def __init__(self):
      self._view_cur, self._view_next, self.n = None, 0, 0

    def next(self):
      self.n += 1
      return ('f%d' % self.n, 'a%d' % self.n)
  p = Stub()
  f, a = L._FieldView(p, 0), L._FieldView(p, 1)
  got = [f.next(), a.next(), f.next(), f.next(), a.next(), a.next(), f.next()]
  assert got == ['f1', 'a1', 'f2', 'f3', 'a3', 'a4', 'f4']
