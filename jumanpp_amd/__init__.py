"""jumanpp_amd -- MI355X-native Juman++ analysis hot path."""
from .native import Context, JppGpuError, Result, load_library  # noqa: F401
