"""mevi_amd -- MI355X-native inference hot path of MEVI (Model-enhanced Vector Index).

Host side is Python (as the reference is), device side is hand-written HIP for
gfx950 behind the C ABI in include/mevi_hip.h.  There is no CPU fallback: every
op raises if libmevi_hip.so is missing or no GPU is present.
"""

__version__ = "0.1.0"
