"""pwicp_amd — thin ctypes mirror of libpwicp.so (the MI355X-native Piecewise-ICP hot path).

The library is HIP-only: importing works anywhere, but every compute call raises PwicpError when
no HIP device is present (there is no CPU fallback)."""
import os as _os

# A series worker runs five HIP streams; the runtime serves a process's streams from four hardware queues unless this variable says
# otherwise, and reads it when it starts.  The library does not edit its host's environment, so the binding does it here, at import
# (no effect if the process - e.g. torch - has already started the runtime; an explicit setting of the caller is kept).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .binding import (PwicpError, Context, Pair, Series, Target, Params, Result, Step, lib_path, load_library,  # noqa: F401
                      device_count, f4, frontend_segment, preprocess, sor_filter, pc_resolution, PiecewiseICP_pair_call,
                      PiecewiseICP_4D_call, series_run_distributed, series_release_parked, frontend_fallback_counts, run_pairs_concurrent)
