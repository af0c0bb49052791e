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
