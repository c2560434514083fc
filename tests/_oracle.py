"""ctypes binding of the CPU oracle (oracle/libpwicp_oracle.so) and of the reference's own
front end (oracle/_ref/libref_frontend.so).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product."""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORC_DIR = os.path.join(_ROOT, "oracle")
ORC_MAX_OUTER = 256


class LoopIO(C.Structure):
    _fields_ = [
        ("Res1", C.c_float), ("Res2", C.c_float), ("SVRes1", C.c_float), ("SVRes2", C.c_float),
        ("isManualDTinit", C.c_int), ("DTinit", C.c_float), ("DTmin", C.c_float),
        ("faithful_cost", C.c_int),
        ("status", C.c_int), ("n_outer", C.c_int),
        ("T16", C.c_float * 16), ("VCM", C.c_double * 36),
        ("DTseries", C.c_float * (ORC_MAX_OUTER + 1)),
        ("n_inner", C.c_int * ORC_MAX_OUTER),
        ("n_stable", C.c_int * ORC_MAX_OUTER),
        ("n_stable_pts", C.c_int * ORC_MAX_OUTER),
        ("LoDmin", C.c_float * ORC_MAX_OUTER),
        ("maxBB", C.c_float * ORC_MAX_OUTER),
        ("d75", C.c_double * ORC_MAX_OUTER),
        ("Tk", (C.c_float * 16) * ORC_MAX_OUTER),
        ("n_corr", C.c_longlong), ("t_loop_s", C.c_double), ("t_inner_s", C.c_double),
        ("n_inner_total", C.c_longlong),
    ]


def build(force=False):
    so = os.path.join(_ORC_DIR, "libpwicp_oracle.so")
    src = os.path.join(_ORC_DIR, "pwicp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORC_DIR, "-s"])
    return so


_lib = None
_ref = None
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [fp, C.c_int]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_kdtree_nn1.argtypes = [C.c_void_p, fp, C.c_int, ip, fp]
        L.orc_kdtree_knn.argtypes = [C.c_void_p, fp, C.c_int, ip, fp]
        L.orc_determine_correspondences.argtypes = [fp, C.c_int, fp, C.c_int, ip, fp]
        L.orc_cal_patch_normal.argtypes = [fp, C.c_int, fp, fp, fp]
        L.orc_cal_patch_normal.restype = C.c_int
        L.orc_cal_patch_std.argtypes = [fp, C.c_int]
        L.orc_cal_patch_std.restype = C.c_float
        L.orc_patch_refinement.argtypes = [fp, C.c_int, C.c_double, C.POINTER(C.c_ubyte)]
        L.orc_patch_refinement.restype = C.c_int
        L.orc_cal_patch_feature.argtypes = [fp, C.c_int, fp, fp, fp]
        L.orc_cal_patch_ct_bp.argtypes = [fp, C.c_int, fp, fp]
        L.orc_select_patches.argtypes = [fp, C.c_int, ip, C.c_int] + [C.POINTER(C.c_void_p)] * 7
        L.orc_select_patches.restype = C.c_int
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_p2p_icp.argtypes = [fp, fp, C.c_int, fp, fp, C.c_int, C.c_double, fp, C.POINTER(C.c_longlong)]
        L.orc_p2p_icp.restype = C.c_int
        L.orc_p2p_lls.argtypes = [fp, fp, fp, ip, C.c_int, dp, dp, dp, fp]
        L.orc_cal_trans_para_vcm.argtypes = [fp, fp, C.c_int, fp, C.c_int, dp]
        L.orc_percentile_dist.argtypes = [fp, C.c_int, fp, C.c_int, C.c_float]
        L.orc_percentile_dist.restype = C.c_double
        L.orc_octree_bbox.argtypes = [fp, C.c_int, C.c_double, dp]
        L.orc_bb_corner_change.argtypes = [dp, fp]
        L.orc_bb_corner_change.restype = C.c_float
        L.orc_matrix2angle.argtypes = [fp, fp]
        L.orc_transform_points.argtypes = [fp, C.c_int, fp]
        L.orc_mat4_mul.argtypes = [fp, fp, fp]
        L.orc_voxel_grid.argtypes = [fp, C.c_int, C.c_float, fp]
        L.orc_voxel_grid.restype = C.c_int
        L.orc_sor_filter.argtypes = [fp, C.c_int, C.c_int, C.c_double, fp]
        L.orc_sor_filter.restype = C.c_int
        L.orc_pc_resolution.argtypes = [fp, C.c_int]
        L.orc_pc_resolution.restype = C.c_float
        L.orc_overlap_ratio.argtypes = [fp, C.c_int, fp, C.c_int, C.c_float]
        L.orc_overlap_ratio.restype = C.c_float
        L.orc_piecewise_icp_loop.argtypes = [fp, C.c_int, fp, C.c_int, fp, ip, C.c_int, fp, fp,
                                             fp, ip, C.c_int, fp, fp, C.POINTER(LoopIO)]
        L.orc_piecewise_icp_loop.restype = C.c_int
    return _lib


def ref_frontend_available():
    return os.path.exists(os.path.join(_ORC_DIR, "_ref", "libref_frontend.so"))


def ref_lib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(_ORC_DIR, "_ref", "libref_frontend.so"))
        _ref.ref_frontend_run.argtypes = [fp, C.c_int, C.c_int, C.c_double, C.c_int, ip, dp, ip]
        _ref.ref_frontend_run.restype = C.c_int
    return _ref


def f4(a):
    """(n,3|4) array -> contiguous float32 (n,4) PointXYZ layout (pad = 1)."""
    a = np.asarray(a, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("expected (n,3) or (n,4)")
    if a.shape[1] == 3:
        out = np.ones((a.shape[0], 4), dtype=np.float32)
        out[:, :3] = a
        return out
    return np.ascontiguousarray(a)


def _p(a, t=fp):
    return a.ctypes.data_as(t)


def nn1(tgt, qry):
    tgt, qry = f4(tgt), f4(qry)
    idx = np.empty(len(qry), np.int32)
    d2 = np.empty(len(qry), np.float32)
    lib().orc_determine_correspondences(_p(tgt), len(tgt), _p(qry), len(qry), _p(idx, ip), _p(d2))
    return idx, d2


def ref_frontend(cloud, sv_resolution, knn=45, want_aux=False):
    cloud = f4(cloud)
    n = len(cloud)
    labels = np.empty(n, np.int32)
    normals = np.empty((n, 3), np.float64) if want_aux else None
    neigh = np.empty((n, knn), np.int32) if want_aux else None
    nsv = ref_lib().ref_frontend_run(_p(cloud), n, 4, float(sv_resolution), knn, _p(labels, ip),
                                     _p(normals, dp) if want_aux else None,
                                     _p(neigh, ip) if want_aux else None)
    if nsv < 0:
        raise RuntimeError("reference front end failed")
    return (labels, nsv, normals, neigh) if want_aux else (labels, nsv)


class Patches:
    """Selected patches of one cloud (S.cpp:97-150 + 306-321)."""

    def __init__(self, pat, off, src, ct, bp, bpstd, ctstd):
        self.pat, self.off, self.src, self.ct, self.bp, self.bpstd, self.ctstd = pat, off, src, ct, bp, bpstd, ctstd

    @property
    def m(self):
        return len(self.off) - 1


def _take(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


def select_patches(cloud, labels, nsv):
    cloud = f4(cloud)
    labels = np.ascontiguousarray(labels, np.int32)
    outs = [C.c_void_p() for _ in range(7)]
    m = lib().orc_select_patches(_p(cloud), len(cloud), _p(labels, ip), int(nsv), *[C.byref(o) for o in outs])
    off = _take(outs[1], (m + 1,), np.int32)
    tot = int(off[m])
    res = Patches(_take(outs[0], (tot, 4), np.float32), off, _take(outs[2], (tot,), np.int32),
                  _take(outs[3], (m, 4), np.float32), _take(outs[4], (m * 6, 4), np.float32),
                  _take(outs[5], (m,), np.float32), _take(outs[6], (m,), np.float32))
    for o in outs:
        lib().orc_free(o)
    return res


def run_loop(cloud1, cloud2, P1, P2, Res1, Res2, SVRes1, SVRes2, DTinit, DTmin, manual_dt=True, faithful=False):
    """orc_piecewise_icp_loop on copies of the source arrays. Returns the LoopIO struct."""
    c1 = f4(cloud1)
    c2 = f4(cloud2).copy()
    pat2, ct2, bp2 = P2.pat.copy(), P2.ct.copy(), P2.bp.copy()
    io = LoopIO()
    io.Res1, io.Res2, io.SVRes1, io.SVRes2 = Res1, Res2, SVRes1, SVRes2
    io.isManualDTinit, io.DTinit, io.DTmin = int(manual_dt), DTinit, DTmin
    io.faithful_cost = int(faithful)
    lib().orc_piecewise_icp_loop(_p(c1), len(c1), _p(c2), len(c2),
                                 _p(P1.pat), _p(P1.off, ip), P1.m, _p(P1.ct), _p(P1.bp),
                                 _p(pat2), _p(P2.off, ip), P2.m, _p(ct2), _p(bp2), C.byref(io))
    return io


def voxel_grid(cloud, leaf):
    cloud = f4(cloud)
    out = np.empty_like(cloud)
    m = lib().orc_voxel_grid(_p(cloud), len(cloud), leaf, _p(out))
    return out[:m].copy()


def sor(cloud, k, mult):
    cloud = f4(cloud)
    out = np.empty_like(cloud)
    m = lib().orc_sor_filter(_p(cloud), len(cloud), k, mult, _p(out))
    return out[:m].copy()


def pc_resolution(cloud):
    cloud = f4(cloud)
    return float(lib().orc_pc_resolution(_p(cloud), len(cloud)))


def set_num_threads(n):
    """Host threads of the oracle's batch NN searches (1 = the faithful single-threaded cost)."""
    L = lib()
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_set_num_threads.restype = None
    L.orc_set_num_threads(int(n))


def max_threads():
    L = lib()
    L.orc_get_max_threads.restype = C.c_int
    return int(L.orc_get_max_threads())


def matrix2angle(T):
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    a = np.zeros(3, np.float32)
    lib().orc_matrix2angle(_p(T), _p(a))
    return a
