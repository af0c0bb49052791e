"""`advoc.spectral` of the reference (/root/reference/advoc/spectral.py): an alias of `advoc_amd.spectral`."""
import sys

from advoc_amd import spectral as _impl

sys.modules[__name__] = _impl
