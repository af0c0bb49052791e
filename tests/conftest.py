import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN


@pytest.fixture(scope='session')
def hip():
  """The loaded C-ABI library; GPU tests fail loudly (not skip) when it is missing."""
  import torch
  from advoc_amd import _lib
  assert torch.cuda.is_available(), 'gpu-marked test running without a HIP device'
  return _lib.load()


@pytest.fixture
def hipenv(hip):
  """Sets ADVOC_* diagnostic switches for one test: the library caches its environment on first use, so every
  change is followed by advoc_tuning_reload(); the previous values come back (and are re-read) afterwards."""
  from advoc_amd import _lib
  saved = {}

  def set_env(**kw):
    for k, v in kw.items():
      if k not in saved:
        saved[k] = os.environ.get(k)
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = str(v)
    _lib.reload_env()
  yield set_env
  for k, v in saved.items():
    if v is None:
      os.environ.pop(k, None)
    else:
      os.environ[k] = v
  _lib.reload_env()
