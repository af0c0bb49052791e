"""`advoc.util` of the reference (/root/reference/advoc/util.py): an alias of `advoc_amd.util`."""
import sys

from advoc_amd import util as _impl

sys.modules[__name__] = _impl
