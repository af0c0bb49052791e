"""The reference's own import lines work unchanged against this repository: the `advoc`
package (advoc/{spectral,loader,audioio,util}.py) and the flat modules next to
models/advoc/train_evaluate.py (/root/reference/models/advoc/train_evaluate.py:1-10,
scripts/spectrogram_advoc.py:10-12, advoc/util.py:4).  Run in a fresh interpreter so that nothing this test
session imported earlier can make them pass."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REFERENCE_STYLE = r'''
import sys
# the reference's scripts run with their own directory first on sys.path and `advoc` installed (setup.py)
sys.path.insert(0, %(models)r)
sys.path.insert(1, %(root)r)
from advoc.loader import decode_extract_and_batch            # train_evaluate.py:2
from model import Modes                                       # :3
from util import override_model_attrs                         # :4
import advoc.spectral                                         # :7
from advoc_model import Advoc                                 # :8
from advoc_model_small import Advoc as AdvocSmall             # :9
from spectral_util import SpectralUtil                        # :10
from advoc.audioio import decode_audio, save_as_wav           # scripts/audio_to_spectrogram.py, spectrogram_advoc.py:10
from advoc.spectral import r9y9_melspec_to_waveform, magspec_to_waveform_lws        # spectrogram_advoc.py:11
from advoc.spectral import create_inverse_mel_filterbank, create_mel_filterbank     # :12
from advoc.spectral import stft, stft_tf, lws_hann_default, waveform_to_melspec, waveform_to_melspec_tf
from advoc.spectral import waveform_to_r9y9_melspec, waveform_to_r9y9_melspec_tf, waveform_to_tacotron2_melspec
from advoc.spectral import magspec_to_waveform_griffin_lim, melspec_to_waveform
from advoc import util as advoc_util
import advoc_amd.spectral, advoc_amd.loader, advoc_amd.model
assert advoc.spectral is advoc_amd.spectral and advoc_util.r9y9_melspec_norm(0.5) == 0.0
assert decode_extract_and_batch is advoc_amd.loader.decode_extract_and_batch
assert Advoc is advoc_amd.model.Advoc and AdvocSmall is advoc_amd.model.AdvocSmall
assert (Advoc.ngf, AdvocSmall.ngf, Advoc.train_batch_size, Modes.TRAIN) == (64, 32, 8, 'train')
m, summary = override_model_attrs(AdvocSmall(Modes.TRAIN), 'train_batch_size=32,use_batchnorm=True')
assert m.train_batch_size == 32 and m.use_batchnorm is True and 'ngf,32' in summary
import inspect
sig = inspect.signature(decode_extract_and_batch)
assert list(sig.parameters)[:3] == ['fps', 'batch_size', 'slice_len'] and len(sig.parameters) == 21
print('reference imports ok')
'''


def test_reference_import_lines_work_unchanged():
  code = REFERENCE_STYLE % dict(models=os.path.join(ROOT, 'models', 'advoc'), root=ROOT)
  env = dict(os.environ)
  env.pop('PYTHONPATH', None)
  out = subprocess.run([sys.executable, '-c', code], cwd='/', env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr
  assert 'reference imports ok' in out.stdout


def test_reference_cli_path_exists():
  for rel in ('models/advoc/train_evaluate.py', 'models/advoc/advoc_model.py', 'models/advoc/advoc_model_small.py',
              'models/advoc/spectral_util.py', 'models/advoc/model.py', 'models/advoc/util.py',
              'advoc/__init__.py', 'advoc/spectral.py', 'advoc/loader.py', 'advoc/audioio.py', 'advoc/util.py',
              'scripts/audio_to_spectrogram.py', 'scripts/spectrogram_advoc.py', 'datacfg/ljspeech.txt'):
    assert os.path.isfile(os.path.join(ROOT, rel)), rel
