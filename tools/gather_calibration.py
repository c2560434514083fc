"""What FETCH_SIZE reports for the dense search's kind of load: N divergent 12-byte gathers (global_load_dwordx3), one per
`stride` bytes of a buffer larger than the Infinity Cache, i.e. N distinct lines from HBM.  Run under
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o gcal -- python tools/gather_calibration.py
and read k_gather_calibration's rows: FETCH_SIZE [KiB] * 1024 / N = bytes the counter reports per gathered line
(tools/collect_profiles.sh does it for strides 64, 128 and 256 and writes profiles/rNN_gather_calibration.txt)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
ctx = P.Context(0)
L = P.load_library()
L.pwicp_debug_gather_calibration.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int]
n = int(os.environ.get("GCAL_N", str(8 * 1024 * 1024)))
for stride in (64, 128, 256):
    rc = L.pwicp_debug_gather_calibration(ctx._h, n, stride, 3)
    print("stride %d bytes: %d gathers per launch, 3 launches, rc %d" % (stride, n, rc))
