"""pwicp_amd — thin ctypes mirror of libpwicp.so (the MI355X-native Piecewise-ICP hot path).

The library is HIP-only: importing works anywhere, but every compute call raises PwicpError when
no HIP device is present (there is no CPU fallback)."""
from .binding import (PwicpError, Context, Pair, Series, Target, Params, Result, Step, lib_path, load_library,  # noqa: F401
                      device_count, f4, frontend_segment, preprocess, sor_filter, pc_resolution, PiecewiseICP_pair_call,
                      PiecewiseICP_4D_call, series_run_distributed, series_release_parked)
