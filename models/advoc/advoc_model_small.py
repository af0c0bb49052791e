"""Same module path as the reference (models/advoc/advoc_model_small.py): its class is called ``Advoc`` there
too (``from advoc_model_small import Advoc as AdvocSmall``, train_evaluate.py:9); the implementation is
advoc_amd.model.AdvocSmall."""
import os
import sys

_ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
if _ROOT not in sys.path:
  sys.path.insert(0, _ROOT)

from advoc_amd.model import EPS, Model, Modes  # noqa: E402,F401
from advoc_amd.model import AdvocSmall as Advoc  # noqa: E402,F401
