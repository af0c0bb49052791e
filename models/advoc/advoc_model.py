"""Same module path as the reference (models/advoc/advoc_model.py, imported flat -- `from advoc_model import ...` -- by the
scripts next to it); the implementation lives in advoc_amd/model.py."""
import os
import sys

_ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
if _ROOT not in sys.path:
  sys.path.insert(0, _ROOT)

from advoc_amd.model import Advoc, Model, Modes, EPS  # noqa: E402,F401
