"""`sgm` — the reference's plugin surface (same dotted ``target:`` paths as ZYM-PKU/UDiffText's sgm/),
backed by the gfx950 kernels of libudt_kernels.so.  See INTEGRATION.md."""
from .models import AutoencodingEngine, DiffusionEngine
from .util import instantiate_from_config

__all__ = ["AutoencodingEngine", "DiffusionEngine", "instantiate_from_config"]
