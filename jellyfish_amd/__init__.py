"""jellyfish_amd: MI355X (gfx950) implementation of the jellyfish PLONK proving hot path.

The product is the C-ABI library ``libjfhip.so`` (``include/jfhip.h``), built from the HIP
sources in ``jellyfish_amd/csrc``.  This package is the thin Python plumbing around it
(ctypes binding, torch device-memory interop, the ``torch.distributed`` sharding layer).
There is no CPU fallback: every call fails loudly if the HIP library is missing.
"""
from .lib import JfHipError, load_library, library_path  # noqa: F401
