"""pings_amd — MI355X-native render + SDF-query core for PINGS (hot path only).

The compute lives in `pings_amd/lib/libpings_hip.so` (hand-written HIP for gfx950,
C ABI in include/pings_hip.h); the modules here are the host-side mirror of the
reference's operator interface for that path.  There is no CPU fallback.
"""
__version__ = "0.1.0"
