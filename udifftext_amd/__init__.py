"""udifftext_amd — MI355X-native (gfx950) implementation of UDiffText's latent-diffusion denoising path.

Sub-modules:
  build     compile libudt_kernels.so (hipcc, gfx950)
  lib/ops   ctypes binding + torch-tensor front end of the C ABI (include/udt_kernels.h)
  packing   checkpoint layout -> device layout
  sgm/      the reference's plugin surface (same dotted ``target:`` paths), backed by the HIP kernels

``import udifftext_amd`` puts this directory on ``sys.path`` so that ``import sgm`` resolves to the
mirror package — which is what makes the reference's ``test.py`` / ``util.py`` / ``configs/*.yaml`` drop-in.
"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
if _HERE not in _sys.path:
    _sys.path.insert(0, _HERE)

__version__ = "0.1.0"
