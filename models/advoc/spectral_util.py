"""Same module path as the reference (models/advoc/spectral_util.py, imported flat -- `from spectral_util import ...` -- by the
scripts next to it); the implementation lives in advoc_amd/spectral_util.py."""
import os
import sys

_ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
if _ROOT not in sys.path:
  sys.path.insert(0, _ROOT)

from advoc_amd.spectral_util import SpectralUtil  # noqa: E402,F401
