"""snickery_amd -- MI355X-native unit-selection search for Snickery voices.

The compute path is libsnkhip.so (hand-written HIP for gfx950 behind the C ABI in
include/snk.h).  There is no CPU fallback: constructing an engine without the library or
without a gfx950 device raises.
"""
from .engine import HipSearchEngine, QueryBatch, SnkError, configure_runtime, device_count, library_path, load_library  # noqa: F401

__all__ = ['HipSearchEngine', 'QueryBatch', 'SnkError', 'configure_runtime', 'library_path', 'load_library']
