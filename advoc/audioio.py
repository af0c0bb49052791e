"""`advoc.audioio` of the reference (/root/reference/advoc/audioio.py): an alias of `advoc_amd.audioio`."""
import sys

from advoc_amd import audioio as _impl

sys.modules[__name__] = _impl
