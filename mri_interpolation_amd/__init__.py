"""MI355X-native hot path for Benjamin-Fouquet/mri_interpolation.

Drop-in class surface of the reference's `encoding.py` / `models.py` / `datamodules.py`
(hash-grid encoders, SIREN / tiny-MLP models, coordinate-batch producer) computed by
hand-written gfx950 HIP kernels behind the C ABI of `include/mri_inr.h`.
"""
from . import _lib  # noqa: F401


def build(force: bool = False):
    """Compile libmri_inr.so in-tree (hipcc, gfx950)."""
    from .build import build as _build
    return _build(force=force)


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import mri_interpolation_amd`
    # stays cheap for tools that only want the build step
    if name in ("ops", "encoding", "models", "optim", "datamodules", "trainer", "parallel",
                "nifti", "config"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
