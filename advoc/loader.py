"""`advoc.loader` of the reference (/root/reference/advoc/loader.py): an alias of `advoc_amd.loader`."""
import sys

from advoc_amd import loader as _impl

sys.modules[__name__] = _impl
