"""lra_amd -- MI355X-native (gfx950) per-read alignment hot path of ChaissonLab/LRA.

The package is a thin host-side mirror of the reference's per-stage functions over the
C ABI of ``liblra_hip.so`` (include/lra_hip.h).  There is NO CPU fallback: importing the
compute entry points without the built HIP library raises.
"""
from ._lib import load_library, library_path, LraError  # noqa: F401
