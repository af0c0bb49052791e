"""Vocoding inference path (scripts/spectrogram_advoc.py:15-22,80-94 semantics)."""
import numpy as np
import pytest
import torch

from oracle import advoc_torch as A
from oracle import spectral_np as S

gpu = pytest.mark.gpu


def test_chunk_plan_always_adds_a_chunk():
  from advoc_amd.infer import chunk_plan
  assert chunk_plan(1, 256) == (256, 1)
  assert chunk_plan(255, 256) == (256, 1)
  assert chunk_plan(256, 256) == (512, 2)       # the reference pads a full extra chunk here
  assert chunk_plan(600, 256) == (768, 3)
  assert chunk_plan(64, 32) == (96, 3)


def oracle_vocode(P, cfg, spec, masks):
  Winv = S.create_inverse_mel_filterbank(22050, 1024, fmin=125, fmax=7600, n_mels=80)
  X = S.tacotron_mel_to_mag(spec[:, :, 0], Winv)
  T = X.shape[0]
  L = cfg.subseq_len
  target = int(T / L) * L + L
  X = np.pad(X, ([0, target - T], [0, 0]), 'constant')
  n = int(target / L)
  chunks = torch.from_numpy(X.reshape(n, L, 513, 1).astype(np.float32))
  gen = A.build_generator(P, chunks, cfg, masks)
  return gen.reshape(target, 513, 1)[:T].numpy()


@gpu
@pytest.mark.parametrize('T', [70, 64])
def test_vocode_matches_oracle(hip, T):
  from advoc_amd.infer import vocode_melspec
  from advoc_amd.model import AdvocSmall, Modes
  L = 32
  cfg = A.Config(small=True, subseq_len=L)
  P = A.init_params(cfg, seed=3)
  rng = np.random.default_rng(T)
  spec = rng.uniform(0.2, 0.9, size=(T, 80, 1))
  n = int(T / L) + 1
  masks = A.make_dropout_masks(cfg, n, seed=4)
  want = oracle_vocode(P, cfg, spec, masks)
  m = AdvocSmall(Modes.INFER)
  m.subseq_len = L
  m.build(batch_size=n)
  m.load_state_dict(P)
  m.set_dropout_masks({k: v.to(torch.uint8) for k, v in masks.items()})
  got = vocode_melspec(m, spec, chunk_batch=n)
  assert got.shape == (T, 513, 1) and got.dtype == np.float32
  assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-4


@gpu
def test_inference_is_stochastic_like_the_reference(hip):
  """Dropout stays on at inference (advoc_model.py:145-149): two calls differ."""
  from advoc_amd.infer import vocode_melspec
  from advoc_amd.model import AdvocSmall, Modes
  m = AdvocSmall(Modes.INFER)
  m.subseq_len = 32
  m.build(batch_size=2)
  spec = np.random.default_rng(0).uniform(0.2, 0.9, size=(40, 80, 1))
  a = vocode_melspec(m, spec, chunk_batch=2)
  b = vocode_melspec(m, spec, chunk_batch=2)
  assert a.shape == b.shape == (40, 513, 1) and np.isfinite(a).all()
  assert not np.array_equal(a, b)


@gpu
def test_vocoding_script_writes_wavs(hip, tmp_path):
  """scripts/spectrogram_advoc.py end to end in its checkpoint-free (pseudo-inverse) mode and
  SpectralUtil.audio_from_mag_spec: mel .npy in, PCM16 .wav out (reference :75-97)."""
  import os
  import subprocess
  import sys
  from advoc_amd.audioio import decode_audio
  from advoc_amd.spectral_util import SpectralUtil
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec_dir, out_dir = tmp_path / 'specs', tmp_path / 'out'
  spec_dir.mkdir()
  rng = np.random.default_rng(0)
  np.save(spec_dir / 'utt0.npy', rng.uniform(0.3, 0.8, size=(50, 80, 1)))
  np.save(spec_dir / 'utt1.npy', rng.uniform(0.3, 0.8, size=(7, 80, 1)))
  r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'spectrogram_advoc.py'), '--spec_dir', str(spec_dir),
                      '--out_dir', str(out_dir), '--phase_estimation', 'gl5'], capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-2000:]
  for name, T in (('utt0', 50), ('utt1', 7)):
    fs, wav = decode_audio(str(out_dir / (name + '.wav')), fastwav=True)
    assert fs == 22050 and wav.shape == ((T - 1) * 256 + 1024, 1, 1)
    assert np.isfinite(wav).all() and float(np.abs(wav).max()) > 0
  su = SpectralUtil()
  mag = np.abs(rng.standard_normal((12, 513, 1))).astype(np.float32)
  np.random.seed(1)
  x = su.audio_from_mag_spec(mag, phase_estimation='gl2')
  assert x.shape == (11 * 256 + 1024, 1, 1) and x.dtype == np.float32
  x = su.audio_from_mag_spec(mag, phase_estimation='lws')
  assert x.shape == (11 * 256 + 1024, 1, 1) and x.dtype == np.float32 and np.isfinite(x).all()
  with pytest.raises(ValueError):
    su.audio_from_mag_spec(mag, phase_estimation='nope')


@gpu
def test_vocode_batch_equals_per_sample_vocoding(hip):
  """The batched z -> mel -> magnitude -> waveform path (BASELINE configs[4]: 64-frame SC09 clips through
  the full model with its (1,2)-stride layers) gives, per sample, what the per-utterance path gives."""
  from advoc_amd import spectral
  from advoc_amd.infer import vocode_batch, vocode_melspec
  from advoc_amd.model import Advoc, Modes
  m = Advoc(Modes.INFER)
  m.subseq_len = 64
  m.audio_fs = 16000
  m.build(batch_size=6, seed=1)
  rng = np.random.default_rng(5)
  specs = rng.uniform(0.1, 0.9, size=(3, 64, 80, 1)).astype(np.float32)
  masks = {k: (torch.rand(buf.shape, generator=torch.Generator().manual_seed(i)) >= 0.5).to(torch.uint8)
           for i, (k, buf) in enumerate((('decoder_%d' % idx, b[0]) for idx, b in m._built['masks'].items()))}
  m.set_dropout_masks(masks)
  gen, wav = vocode_batch(m, specs, phase_estimation='gl2', chunk_batch=6,
                          unit_phase=torch.full((3, 64, 513), 0.25, device='cuda'))
  assert tuple(gen.shape) == (3, 64, 513) and tuple(wav.shape) == (3, 63 * 256 + 1024)
  assert torch.isfinite(wav).all()
  for i in range(3):
    m.build(batch_size=2)
    m.set_dropout_masks({k: v[2 * i:2 * i + 2] for k, v in masks.items()})
    one = vocode_melspec(m, specs[i].astype(np.float64), chunk_batch=2)
    assert np.linalg.norm(one[:, :, 0] - gen[i].cpu().numpy()) / np.linalg.norm(one) < 1e-5
  g0, none = vocode_batch(m, specs[:1], phase_estimation=None, chunk_batch=2)
  assert none is None and tuple(g0.shape) == (1, 64, 513)
  g1, w1 = vocode_batch(m, specs[:1], phase_estimation='lws', chunk_batch=2)
  assert tuple(w1.shape) == (1, 63 * 256 + 1024) and bool(torch.isfinite(w1).all())


@gpu
def test_feature_dump_script(hip, tmp_path, golden_dir):
  """scripts/audio_to_spectrogram.py: WAV directory in, one float64 [T, 80, 1] .npy per file out, equal
  to waveform_to_r9y9_melspec of the normalised mono decode (reference scripts/audio_to_spectrogram.py:36-51)."""
  import os
  import shutil
  import subprocess
  import sys
  from advoc_amd import spectral
  from advoc_amd.audioio import decode_audio
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  wavs = tmp_path / 'wavs'
  wavs.mkdir()
  shutil.copy(os.path.join(golden_dir, 'sc09.wav'), wavs / 'a.wav')
  out = tmp_path / 'mels'
  r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'audio_to_spectrogram.py'), '--wave_dir', str(wavs),
                      '--out_dir', str(out), '--fs', '16000', '--data_fast_wav'], capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-2000:]
  got = np.load(out / 'a.npy')
  _, x = decode_audio(str(wavs / 'a.wav'), fs=16000, mono=True, normalize=True, fastwav=True)
  want = spectral.waveform_to_r9y9_melspec(x, fs=16000)
  assert got.dtype == np.float64 and got.shape == want.shape and got.shape[1:] == (80, 1)
  assert np.abs(got - want).max() < 1e-6
