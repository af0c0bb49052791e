"""advoc_amd -- MI355X-native (gfx950) adversarial-vocoder hot path.

Drop-in for the ``advoc.spectral`` / ``advoc.loader`` / ``advoc.audioio`` API
surface and the ``models/advoc`` train / eval CLI of paarthneekhara/advoc; the
arithmetic runs in hand-written HIP kernels behind the C ABI in
``include/advoc_hip.h`` (``advoc_amd/csrc/libadvoc_hip.so``).  There is no CPU
fallback: importing is cheap, but the first kernel call raises if the library
or a HIP device is missing.
"""
__version__ = '0.1.0'
