"""The reference's package name (/root/reference/advoc/__init__.py): ``import advoc.spectral``,
``from advoc.loader import decode_extract_and_batch``, ``from advoc.audioio import decode_audio`` resolve to
the MI355X implementation in ``advoc_amd`` -- each submodule here IS the advoc_amd module of the same name
(same module object, so monkeypatching and isinstance checks see one thing)."""
from advoc_amd import __version__  # noqa: F401
