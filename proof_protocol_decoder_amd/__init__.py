"""MI355X-native block-proof hot path (host-side Python binding of libbpg.so).

The product path is the HIP library behind the C ABI in include/bpg.h.  There is NO CPU fallback:
importing the kernels on a machine without the built library raises, and every op raises when no
gfx950 device is present.
"""
from ._lib import BpgError, lib, lib_path  # noqa: F401
from . import ops  # noqa: F401
from . import proof_gen  # noqa: F401
from . import compact  # noqa: F401
from . import trace_protocol  # noqa: F401
