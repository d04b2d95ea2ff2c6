"""MI355X-native (gfx950) implementation of the GAN training inner loop of Cameronr11/GAN-Variant-Research.

Import name: ``gan_variant_research_amd`` (the shim next to this directory maps it here).  The compute path is the
C-ABI shared library ``libmi355x_gan.so`` (include/mi355x_gan.h); this package is its host side.
"""
from . import _lib  # noqa: F401
from ._lib import BF16, F32, FP8, GanError  # noqa: F401

__all__ = ["_lib", "BF16", "F32", "FP8", "GanError"]
