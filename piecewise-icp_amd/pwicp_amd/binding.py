"""ctypes binding of include/pwicp.h.  Names mirror the reference's own functions where one
exists (calPercentileDistBetween2PC, P2PICPwithPatchNormal, calTransParaVCM, Piecewise_ICP ...)."""
import atexit
import ctypes as C
import os
import weakref
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PWICP_MAX_OUTER = 256

STATUS = {0: "OK", -1: "NO_DEVICE", -2: "INVALID", -3: "TOO_FEW_PATCHES", -4: "TOO_FEW_STABLE",
          -5: "NOMEM", -6: "INTERNAL", -7: "NOT_CONVERGED"}


# Handles still open when the interpreter exits are closed here, dependants first (pairs, targets, series, then contexts), while
# the HIP runtime is still up: a handle collected during interpreter shutdown - or never - would otherwise meet a runtime that
# is already tearing down (seen as a std::bad_variant_access abort at exit with 8 contexts left open).
_LIVE = {k: weakref.WeakSet() for k in ("pair", "target", "series", "context")}


def _track(kind, obj):
    _LIVE[kind].add(obj)


@atexit.register
def _close_all():
    for kind in ("pair", "target", "series", "context"):
        for obj in list(_LIVE[kind]):
            try:
                obj.close()
            except Exception:
                pass


class PwicpError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("pwicp: %s (%d) %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("Res1", C.c_float), ("Res2", C.c_float), ("SVRes1", C.c_float), ("SVRes2", C.c_float),
                ("isManualDTinit", C.c_int), ("DTinit", C.c_float), ("DTmin", C.c_float)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("n_outer", C.c_int), ("T16", C.c_float * 16), ("VCM", C.c_double * 36),
                ("DTseries", C.c_float * (PWICP_MAX_OUTER + 1)),
                ("n_inner", C.c_int * PWICP_MAX_OUTER), ("n_stable", C.c_int * PWICP_MAX_OUTER),
                ("n_stable_pts", C.c_int * PWICP_MAX_OUTER), ("LoDmin", C.c_float * PWICP_MAX_OUTER),
                ("maxBB", C.c_float * PWICP_MAX_OUTER), ("d75", C.c_double * PWICP_MAX_OUTER),
                ("Tk", (C.c_float * 16) * PWICP_MAX_OUTER),
                ("n_corr", C.c_longlong), ("n_corr_dense", C.c_longlong), ("n_inner_total", C.c_longlong),
                ("t_loop_ms", C.c_double), ("t_dense_nn_ms", C.c_double), ("n_dense_nn_launches", C.c_int),
                ("t_inner_ms", C.c_double), ("dense_kbar", C.c_double),
                ("dense_rows", C.c_int32), ("n_dense_bounded", C.c_int32)]


class Step(C.Structure):
    """pwicp_step: the state PwICP_singleIteration keeps between calls (R.h:181-188, R.cpp:11-14) + this call's outputs."""
    _fields_ = [("currDT", C.c_float), ("BBchange_1", C.c_float), ("BBchange_2", C.c_float),
                ("toStage2", C.c_int), ("toStage3", C.c_int), ("status", C.c_int), ("T16", C.c_float * 16),
                ("VCM", C.c_double * 36), ("n_stable", C.c_int), ("n_stable_pts", C.c_int), ("n_inner", C.c_int),
                ("LoDmin", C.c_float), ("maxBB", C.c_float), ("d75", C.c_double)]


def lib_path():
    # $PWICP_LIB: another build of the same library (A/B of compile-time variants, tools/build_variant.sh); never a fallback
    return os.environ.get("PWICP_LIB") or os.path.join(os.path.dirname(_HERE), "libpwicp.so")


_lib = None
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
dp = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)


def load_library():
    """Loads libpwicp.so (built by __graft_entry__.build()).  Raises if it is missing — the
    product has no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise PwicpError(-6, "libpwicp.so not built: run `python __graft_entry__.py` (build())")
    L = C.CDLL(p)
    vp = C.c_void_p
    L.pwicp_version.restype = C.c_char_p
    L.pwicp_last_error.restype = C.c_char_p
    L.pwicp_last_error.argtypes = [vp]
    L.pwicp_create.argtypes = [C.POINTER(vp), C.c_int]
    L.pwicp_destroy.argtypes = [vp]
    L.pwicp_nn_search.argtypes = [vp, fp, C.c_int, fp, C.c_int, ip, fp]
    L.pwicp_percentile_dist.argtypes = [vp, fp, C.c_int, fp, C.c_int, C.c_float, dp]
    L.pwicp_overlap_ratio.argtypes = [vp, fp, C.c_int, fp, C.c_int, C.c_float, fp]
    for name, args in [
        ("pwicp_patch_normals", [vp, fp, ip, C.c_int, fp, u8p]),
        ("pwicp_select_patches", [vp, fp, C.c_int, ip, C.c_int, ip, ip, fp, ip, ip, fp, fp, fp, fp]),
        ("pwicp_p2p_icp", [vp, fp, fp, C.c_int, fp, fp, C.c_int, C.c_double, fp, ip]),
        ("pwicp_trans_para_vcm", [vp, fp, fp, C.c_int, fp, C.c_int, dp]),
        ("pwicp_pair_create", [vp, fp, C.c_int, ip, C.c_int, fp, C.c_int, ip, C.c_int, C.POINTER(Params),
                               C.POINTER(vp)]),
        ("pwicp_pair_create_from_patches", [vp, fp, C.c_int, fp, ip, C.c_int, fp, C.c_int, fp, ip, C.c_int,
                                            C.POINTER(Params), C.POINTER(vp)]),
        ("pwicp_pair_destroy", [vp]),
        ("pwicp_pair_num_patches", [vp, ip, ip]),
        ("pwicp_pair_reset", [vp]),
        ("pwicp_pair_run", [vp, C.POINTER(Result)]),
        ("pwicp_pair_download_source", [vp, fp]),
        ("pwicp_pair_step", [vp, C.POINTER(Step)]),
        ("pwicp_pair_auto_dtinit", [vp, fp]),
        ("pwicp_pair_bench_dense_nn", [vp, C.c_int, dp, C.POINTER(C.c_longlong), dp, dp]),
        ("pwicp_pair_dense_distances", [vp, C.c_int, fp]),
        ("pwicp_pair_num_patch_points", [vp, ip, ip]),
        ("pwicp_pair_download_state", [vp, fp, fp, fp, fp]),
        ("pwicp_pair_set_profiling", [vp, C.c_int]),
        ("pwicp_target_create", [vp, fp, C.c_int, ip, C.c_int, C.c_float, C.c_float, C.POINTER(vp)]),
        ("pwicp_target_destroy", [vp]),
        ("pwicp_pair_create_with_target", [vp, fp, C.c_int, ip, C.c_int, C.POINTER(Params), C.POINTER(vp)]),
        ("pwicp_pair_create_with_target_on", [vp, vp, fp, C.c_int, ip, C.c_int, C.POINTER(Params), C.POINTER(vp)]),
    ]:
        if hasattr(L, name):
            getattr(L, name).argtypes = args
    L.pwicp_frontend_segment.argtypes = [fp, C.c_int, C.c_float, C.c_int, ip, ip]
    L.pwicp_knn.argtypes = [vp, fp, C.c_int, C.c_int, C.c_float, ip]
    L.pwicp_frontend_segment_dev.argtypes = [vp, fp, C.c_int, C.c_float, C.c_int, C.c_float, ip, ip]
    L.pwicp_series_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, ip, C.c_int, C.POINTER(vp)]
    L.pwicp_series_close.argtypes = [vp]
    L.pwicp_series_close.restype = None
    L.pwicp_series_num_pairs.argtypes = [vp]
    L.pwicp_series_num_scans.argtypes = [vp]
    L.pwicp_series_pair_epochs.argtypes = [vp, C.c_int, ip, ip, C.POINTER(C.c_long)]
    L.pwicp_series_adaptive_targets.argtypes = [vp, ip, C.c_int]
    L.pwicp_series_stage_times.argtypes = [vp, C.POINTER(C.c_double)]
    L.pwicp_series_overlap_ratios.argtypes = [vp, ip, C.c_int, fp]
    L.pwicp_series_adaptive_from_ratios.argtypes = [vp, fp, C.c_float, C.c_int]
    L.pwicp_series_run_pair.argtypes = [vp, C.c_int, vp]
    L.pwicp_series_run_pairs.argtypes = [vp, ip, C.c_int, vp]
    L.pwicp_series_write_results.argtypes = [vp, vp, C.c_int]
    L.pwicp_series_expect_target_labels.argtypes = [vp, C.c_int]
    L.pwicp_series_supply_target_labels.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip]
    L.pwicp_series_wait_target_labels.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), ip, C.c_int]
    L.pwicp_series_close_target_labels.argtypes = [vp]
    L.pwicp_series_target_label_counts.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pwicp_series_close_target_labels.restype = None
    L.pwicp_pc_resolution_dev.argtypes = [vp, fp, C.c_int, fp]
    L.pwicp_preprocess_dev.argtypes = [vp, fp, C.c_int, C.c_float, C.c_int, C.c_double, fp, ip]
    L.pwicp_preprocess.argtypes = [fp, C.c_int, C.c_float, C.c_int, C.c_double, fp, ip]
    L.pwicp_sor_filter.argtypes = [fp, C.c_int, C.c_int, C.c_double, fp, ip]
    L.pwicp_sor_filter_dev.argtypes = [vp, fp, C.c_int, C.c_int, C.c_double, C.c_float, fp, ip]
    L.pwicp_pc_resolution.argtypes = [fp, C.c_int]
    L.pwicp_pc_resolution.restype = C.c_float
    L.PiecewiseICP_pair_call.argtypes = [C.c_char_p, C.c_char_p]
    L.PiecewiseICP_pair_call.restype = C.c_bool
    L.pwicp_series_run_distributed.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.pwicp_series_run_distributed.restype = C.c_bool
    L.pwicp_series_set_devices.argtypes = [vp, ip, C.c_int]
    L.PiecewiseICP_4D_call.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float]
    L.PiecewiseICP_4D_call.restype = C.c_bool
    _lib = L
    return L


def device_count():
    return int(load_library().pwicp_device_count())


def f4(a):
    """(n,3|4) -> contiguous float32 (n,4) pcl::PointXYZ layout (pad = 1)."""
    a = np.asarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (3, 4):
        raise ValueError("expected (n,3) or (n,4)")
    if a.shape[1] == 3:
        out = np.ones((a.shape[0], 4), dtype=np.float32)
        out[:, :3] = a
        return out
    return np.ascontiguousarray(a)


def _p(a, t=fp):
    return a.ctypes.data_as(t)


# ---- host-side setup stages (no GPU needed) ------------------------------------------------------------------
def frontend_segment(cloud, sv_resolution, knn=45):
    """Supervoxel labels (PatchGenerationAndRefinement, S.cpp:18-68). Returns (labels int32, n_supervoxels)."""
    L = load_library()
    c = f4(cloud)
    lab = np.empty(len(c), np.int32)
    nsv = C.c_int32()
    rc = L.pwicp_frontend_segment(_p(c), len(c), float(sv_resolution), int(knn), _p(lab, ip), C.byref(nsv))
    if rc != 0:
        raise PwicpError(rc, "pwicp_frontend_segment")
    return lab, nsv.value


def pc_resolution(cloud):
    """calPCresolution (C.cpp:239-263) on the host."""
    c = f4(cloud)
    return float(load_library().pwicp_pc_resolution(_p(c), len(c)))


def preprocess(cloud, voxel_size, sor_k=14, sor_mult=5.0):
    """PCpreprocessing (VoxelGrid + SOR, C.cpp:423-452). Returns float32 (m,4)."""
    L = load_library()
    c = f4(cloud)
    out = np.empty_like(c)
    m = C.c_int32()
    rc = L.pwicp_preprocess(_p(c), len(c), float(voxel_size), int(sor_k), float(sor_mult), _p(out), C.byref(m))
    if rc != 0:
        raise PwicpError(rc, "pwicp_preprocess")
    return out[:m.value].copy()


def sor_filter(cloud, sor_k=14, sor_mult=5.0):
    """SORfilter (C.cpp:441-452) = PCpreprocessing with isDownSamp = false, host version."""
    L = load_library()
    c = f4(cloud)
    out = np.empty_like(c)
    m = C.c_int32()
    rc = L.pwicp_sor_filter(_p(c), len(c), int(sor_k), C.c_double(sor_mult), _p(out), C.byref(m))
    if rc != 0:
        raise PwicpError(rc, "pwicp_sor_filter")
    return out[:m.value].copy()


def series_run_distributed(confile, startEpoch, epochNum, pairMode, overlapThd, rank, world, device, id_file):
    """pwicp_series_run_distributed: one rank of the RCCL-gathered multi-process series (C++, no torch)."""
    return bool(load_library().pwicp_series_run_distributed(str(confile).encode(), int(startEpoch), int(epochNum), int(pairMode),
                                                            float(overlapThd), int(rank), int(world), int(device),
                                                            str(id_file).encode()))


def PiecewiseICP_pair_call(confile, outfile):
    return bool(load_library().PiecewiseICP_pair_call(str(confile).encode(), str(outfile).encode()))


def PiecewiseICP_4D_call(confile, startEpoch, epochNum, pairMode, overlapThd=0.75):
    return bool(load_library().PiecewiseICP_4D_call(str(confile).encode(), int(startEpoch), int(epochNum), int(pairMode),
                                                    float(overlapThd)))


def series_release_parked():
    """Frees the device contexts / front-end work spaces closed series have left parked for the next one (pwicp.h)."""
    L = load_library()
    L.pwicp_series_release_parked.restype = None
    L.pwicp_series_release_parked()


class Series:
    """A 4D series (PiecewiseICP_4D_call, R.cpp:17-215) as a handle whose pairs can be run one by one, on any GPU.
    Records are rows of pwicp_amd.fourd.RECORD (= pwicp_pair_record, 384 bytes)."""

    def __init__(self, confile, startEpoch, epochNum, pairMode, overlapThd=0.75, device=0, adaptive_targets=None):
        self._L = load_library()
        h = C.c_void_p()
        at, n_at = None, 0
        if isinstance(adaptive_targets, str) and adaptive_targets == "deferred":
            n_at = -1                  # adaptive map supplied later: overlap_ratios() + adaptive_from_ratios()
        elif adaptive_targets is not None:
            self._at = np.ascontiguousarray(adaptive_targets, np.int32)
            at, n_at = _p(self._at, ip), len(self._at)
        rc = self._L.pwicp_series_open(str(confile).encode(), int(startEpoch), int(epochNum), int(pairMode),
                                       float(overlapThd), int(device), at, n_at, C.byref(h))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_open")
        self._h = h
        _track("series", self)
        self._start = int(startEpoch)
        self.pair_mode = int(pairMode)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pwicp_series_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def num_pairs(self):
        return int(self._L.pwicp_series_num_pairs(self._h))

    def pair_epochs(self, pair):
        """(target file index, source file index, source epoch stamp) of pair `pair`."""
        t, s, st = C.c_int32(), C.c_int32(), C.c_long()
        rc = self._L.pwicp_series_pair_epochs(self._h, int(pair), C.byref(t), C.byref(s), C.byref(st))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_pair_epochs")
        return t.value, s.value, st.value

    @property
    def num_scans(self):
        return int(self._L.pwicp_series_num_scans(self._h))

    def stage_times(self):
        """Wall time per stage of the pairs run so far (ms) and the raw scan bytes handed to the GPU."""
        v = (C.c_double * 5)()
        if self._L.pwicp_series_stage_times(self._h, v) != 0:
            raise PwicpError(-2, "pwicp_series_stage_times")
        return {"read_scans_ms": v[0], "gpu_preparation_ms": v[1], "front_ends_rest_ms": v[2], "registrations_ms": v[3], "scan_bytes": int(v[4])}

    def overlap_ratios(self, pairs):
        """calOverlapRatioByC2Cdist (R.cpp:593-614) for (target, source) file-index pairs; returns float32 ratios."""
        ij = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        out = np.empty(len(ij), np.float32)
        rc = self._L.pwicp_series_overlap_ratios(self._h, _p(ij, ip), len(ij), _p(out, fp))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_overlap_ratios")
        return out

    def adaptive_from_ratios(self, table, overlapThd=0.75, write_pair_file=False):
        """calAdaptivePairSequence (R.cpp:552-589) replayed on a (#files x #files) table of overlap ratios (NaN = unknown)."""
        t = np.ascontiguousarray(table, np.float32)
        n = self.num_scans
        if t.shape != (n, n):
            raise ValueError("ratio table must be (%d, %d)" % (n, n))
        rc = self._L.pwicp_series_adaptive_from_ratios(self._h, _p(t, fp), float(overlapThd), 1 if write_pair_file else 0)
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_adaptive_from_ratios")

    def adaptive_targets(self):
        """Adaptive pair map: entry k = target of source k+1, both relative to startEpoch."""
        n = self.num_scans - self._start - 1
        out = np.empty(n, np.int32)
        rc = self._L.pwicp_series_adaptive_targets(self._h, _p(out, ip), int(n))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_adaptive_targets")
        return out

    def run_pair(self, pair):
        """One pair of the loop R.cpp:89-150 on this handle's GPU.  Returns a 1-element record array; a failed step
        comes back with status != 0 (the reference skips it), a missing GPU raises."""
        from .fourd import RECORD
        rec = np.zeros(1, RECORD)
        rc = self._L.pwicp_series_run_pair(self._h, int(pair), rec.ctypes.data_as(C.c_void_p))
        if rc == -1:
            raise PwicpError(rc, "pwicp_series_run_pair: no usable HIP device")
        return rec

    def run_pairs(self, pairs):
        """Several pairs, pipelined (scans read and supervoxels computed on host threads for a window of pairs at once).
        Returns a record array with one row per requested pair, in the order given."""
        from .fourd import RECORD
        ids = np.ascontiguousarray(pairs, np.int32)
        recs = np.zeros(len(ids), RECORD)
        if len(ids) == 0:
            return recs
        rc = self._L.pwicp_series_run_pairs(self._h, _p(ids, ip), len(ids), recs.ctypes.data_as(C.c_void_p))
        if rc == -1:
            raise PwicpError(rc, "pwicp_series_run_pairs: no usable HIP device")
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_run_pairs")
        return recs

    # ---- labels of a target that several processes share (pwicp.h: pwicp_series_*_target_labels) ----
    def expect_target_labels(self, scan):
        """The labels of target `scan` will come from another rank (supply_target_labels, from another thread, during run_pairs)."""
        rc = self._L.pwicp_series_expect_target_labels(self._h, int(scan))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_expect_target_labels")

    def supply_target_labels(self, scan, labels, n_supervoxels):
        """labels None: the supplier failed - the run segments the target itself."""
        if labels is None:
            rc = self._L.pwicp_series_supply_target_labels(self._h, int(scan), -1, 0, None)
        else:
            lab = np.ascontiguousarray(labels, np.int32)
            rc = self._L.pwicp_series_supply_target_labels(self._h, int(scan), len(lab), int(n_supervoxels), _p(lab, ip))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_supply_target_labels")

    def wait_target_labels(self, scan, timeout_s=3600.0):
        """Blocks until a run_pairs of this process (another thread) has segmented target `scan`; (labels, n_supervoxels), or None
        when that run failed on the target or ended without it."""
        m, nsv = C.c_int(0), C.c_int(0)
        rc = self._L.pwicp_series_wait_target_labels(self._h, int(scan), int(timeout_s * 1000), C.byref(m), C.byref(nsv), None, 0)
        if rc != 0:
            return None
        lab = np.zeros(max(m.value, 1), np.int32)
        rc = self._L.pwicp_series_wait_target_labels(self._h, int(scan), 0, C.byref(m), C.byref(nsv), _p(lab, ip), len(lab))
        return (lab[:m.value], nsv.value) if rc == 0 else None

    def target_label_counts(self):
        """(targets whose labels came from another rank, targets segmented by this one)"""
        a, b = C.c_int(0), C.c_int(0)
        self._L.pwicp_series_target_label_counts(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close_target_labels(self):
        self._L.pwicp_series_close_target_labels.restype = None
        self._L.pwicp_series_close_target_labels(self._h)

    def set_devices(self, devices):
        """Several GPUs in this process: pairs of a run_pairs call are dealt to them (pair k -> device k mod n)."""
        d = np.ascontiguousarray(devices, np.int32)
        rc = self._L.pwicp_series_set_devices(self._h, _p(d, ip), len(d))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_set_devices")

    def write_results(self, records):
        """records: structured array of RECORD rows (all pairs, any order).  Writes the reference's result files."""
        from .fourd import RECORD
        recs = np.ascontiguousarray(records, RECORD)
        rc = self._L.pwicp_series_write_results(self._h, recs.ctypes.data_as(C.c_void_p), len(recs))
        if rc != 0:
            raise PwicpError(rc, "pwicp_series_write_results")


class Context:
    """One per GPU (per process rank).  Replaces the reference's module-level state."""

    def __init__(self, device_id=0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.pwicp_create(C.byref(h), int(device_id))
        if rc != 0:
            raise PwicpError(rc, "pwicp_create(device %d): no usable HIP device" % device_id)
        self._h = h
        _track("context", self)
        self.device = int(self._L.pwicp_context_device(h)) if hasattr(self._L, "pwicp_context_device") else int(device_id)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pwicp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PwicpError(rc, self._L.pwicp_last_error(self._h).decode("utf-8", "replace"))

    # -- building blocks ---------------------------------------------------------------------
    def determineCorrespondences(self, target, source):
        """CorrespondenceEstimation::determineCorrespondences(.., DBL_MAX): (index_match, sq_distance)."""
        t, q = f4(target), f4(source)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float32)
        self._chk(self._L.pwicp_nn_search(self._h, _p(t), len(t), _p(q), len(q), _p(idx, ip), _p(d2)))
        return idx, d2

    def calPercentileDistBetween2PC(self, cloud1, cloud2, percentile=0.75):
        c1, c2 = f4(cloud1), f4(cloud2)
        out = C.c_double()
        self._chk(self._L.pwicp_percentile_dist(self._h, _p(c1), len(c1), _p(c2), len(c2), percentile, C.byref(out)))
        return out.value

    def calOverlapRatioByC2Cdist(self, cloud1, cloud2, DTinit):
        c1, c2 = f4(cloud1), f4(cloud2)
        out = C.c_float()
        self._chk(self._L.pwicp_overlap_ratio(self._h, _p(c1), len(c1), _p(c2), len(c2), DTinit, C.byref(out)))
        return out.value

    def knn(self, cloud, k, cell_edge=0.0):
        """k nearest neighbours of every point within the cloud (GPU): int32 (n, k), self first."""
        c = f4(cloud)
        nb = np.empty((len(c), k), np.int32)
        self._chk(self._L.pwicp_knn(self._h, _p(c), len(c), int(k), float(cell_edge), _p(nb, ip)))
        return nb

    def frontend_segment(self, cloud, sv_resolution, knn=45, point_spacing=0.0):
        """Supervoxel labels with the k-NN graph built on the GPU (same labels as frontend_segment)."""
        c = f4(cloud)
        lab = np.empty(len(c), np.int32)
        nsv = C.c_int32()
        self._chk(self._L.pwicp_frontend_segment_dev(self._h, _p(c), len(c), float(sv_resolution), int(knn),
                                                      float(point_spacing), _p(lab, ip), C.byref(nsv)))
        return lab, nsv.value

    def preprocess(self, cloud, voxel_size, sor_k=14, sor_mult=5.0):
        """PCpreprocessing on the GPU (same output as the module-level preprocess)."""
        c = f4(cloud)
        out = np.empty_like(c)
        m = C.c_int32()
        self._chk(self._L.pwicp_preprocess_dev(self._h, _p(c), len(c), float(voxel_size), int(sor_k), float(sor_mult),
                                               _p(out), C.byref(m)))
        return out[:m.value].copy()

    def sor_filter(self, cloud, sor_k=14, sor_mult=5.0, spacing_hint=0.0):
        """SORfilter (C.cpp:441-452) with the k-NN statistic on the GPU (same output as the module-level sor_filter)."""
        c = f4(cloud)
        out = np.empty_like(c)
        m = C.c_int32()
        self._chk(self._L.pwicp_sor_filter_dev(self._h, _p(c), len(c), int(sor_k), C.c_double(sor_mult),
                                               C.c_float(spacing_hint), _p(out), C.byref(m)))
        return out[:m.value].copy()

    def pc_resolution(self, cloud):
        """calPCresolution (C.cpp:239-263) with the nearest-neighbour search on the GPU."""
        c = f4(cloud)
        r = C.c_float()
        self._chk(self._L.pwicp_pc_resolution_dev(self._h, _p(c), len(c), C.byref(r)))
        return r.value

    def patchNormals(self, patch_xyz4, offsets):
        pat = f4(patch_xyz4)
        off = np.ascontiguousarray(offsets, np.int32)
        m = len(off) - 1
        nrm = np.zeros((m, 4), np.float32)
        ok = np.zeros(m, np.uint8)
        self._chk(self._L.pwicp_patch_normals(self._h, _p(pat), _p(off, ip), m, _p(nrm), _p(ok, u8p)))
        return nrm, ok

    def selectPatches(self, cloud, labels, nsv):
        """PatchGenerationAndRefinement after segmentation + calBPandCTSTD.
        Returns dict(pat, off, src, ct, bp, bpstd, ctstd)."""
        c = f4(cloud)
        lab = np.ascontiguousarray(labels, np.int32)
        m = C.c_int32()
        tot = C.c_int32()
        self._chk(self._L.pwicp_select_patches(self._h, _p(c), len(c), _p(lab, ip), int(nsv), C.byref(m), C.byref(tot),
                                               None, None, None, None, None, None, None))
        m, tot = m.value, tot.value
        pat = np.zeros((max(tot, 1), 4), np.float32)
        off = np.zeros(m + 1, np.int32)
        src = np.zeros(max(tot, 1), np.int32)
        ct = np.zeros((max(m, 1), 4), np.float32)
        bp = np.zeros((max(m, 1) * 6, 4), np.float32)
        sbp = np.zeros(max(m, 1), np.float32)
        sct = np.zeros(max(m, 1), np.float32)
        mm, tt = C.c_int32(), C.c_int32()
        self._chk(self._L.pwicp_select_patches(self._h, _p(c), len(c), _p(lab, ip), int(nsv), C.byref(mm), C.byref(tt),
                                               _p(pat), _p(off, ip), _p(src, ip), _p(ct), _p(bp), _p(sbp), _p(sct)))
        return dict(pat=pat[:tot], off=off, src=src[:tot], ct=ct[:m], bp=bp[:6 * m], bpstd=sbp[:m], ctstd=sct[:m])

    def P2PICPwithPatchNormal(self, tgt, tgt_n, src, src_n, euclid_eps=1e-6):
        t, tn, s, sn = f4(tgt), f4(tgt_n), f4(src), f4(src_n)
        T = np.zeros(16, np.float32)
        it = C.c_int32()
        self._chk(self._L.pwicp_p2p_icp(self._h, _p(t), _p(tn), len(t), _p(s), _p(sn), len(s), euclid_eps, _p(T),
                                        C.byref(it)))
        return T.reshape(4, 4), it.value

    def calTransParaVCM(self, tgt, tgt_n, src_stable):
        t, tn, s = f4(tgt), f4(tgt_n), f4(src_stable)
        V = np.zeros(36, np.float64)
        self._chk(self._L.pwicp_trans_para_vcm(self._h, _p(t), _p(tn), len(t), _p(s), len(s), _p(V, dp)))
        return V.reshape(6, 6)


PROF_DENSE, PROF_INNER, PROF_REPLAY = 1, 2, 4


class Target:
    """The static target side of a pair (cloud, patches, normals, grids), shareable by all pairs with this target scan
    (every pair of a Direct2Ref series).  Must outlive the pairs created with it."""

    def __init__(self, ctx, cloud1, labels1, nsv1, Res1, SVRes1):
        self._ctx = ctx
        self._L = ctx._L
        c1 = f4(cloud1)
        l1 = np.ascontiguousarray(labels1, np.int32)
        h = C.c_void_p()
        ctx._chk(self._L.pwicp_target_create(ctx._h, _p(c1), len(c1), _p(l1, ip), int(nsv1), float(Res1), float(SVRes1),
                                             C.byref(h)))
        self._h = h
        _track("target", self)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pwicp_target_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pair:
    """A target/source pair resident in HBM; run() is the Piecewise_ICP while-loop."""

    def __init__(self, ctx, cloud1, labels1, nsv1, cloud2, labels2, nsv2, params, patches=None, target=None):
        self._ctx = ctx
        self._L = ctx._L
        self.n2 = len(cloud2)
        h = C.c_void_p()
        if target is not None:                      # cloud1 / labels1 / nsv1 are ignored: the target handle carries them
            self._target = target                   # keep it alive
            c2 = f4(cloud2)
            l2 = np.ascontiguousarray(labels2, np.int32)
            # (ctx may be another context of the target's device: the pair then lives on that context's stream)
            ctx._chk(self._L.pwicp_pair_create_with_target_on(ctx._h, target._h, _p(c2), len(c2), _p(l2, ip), int(nsv2),
                                                              C.byref(params), C.byref(h)))
            self._h = h
            _track("pair", self)
            return
        c1, c2 = f4(cloud1), f4(cloud2)
        if patches is None:
            l1 = np.ascontiguousarray(labels1, np.int32)
            l2 = np.ascontiguousarray(labels2, np.int32)
            ctx._chk(self._L.pwicp_pair_create(ctx._h, _p(c1), len(c1), _p(l1, ip), int(nsv1), _p(c2), len(c2),
                                               _p(l2, ip), int(nsv2), C.byref(params), C.byref(h)))
        else:
            (pat1, off1), (pat2, off2) = patches
            pat1, pat2 = f4(pat1), f4(pat2)
            off1 = np.ascontiguousarray(off1, np.int32)
            off2 = np.ascontiguousarray(off2, np.int32)
            ctx._chk(self._L.pwicp_pair_create_from_patches(ctx._h, _p(c1), len(c1), _p(pat1), _p(off1, ip),
                                                            len(off1) - 1, _p(c2), len(c2), _p(pat2), _p(off2, ip),
                                                            len(off2) - 1, C.byref(params), C.byref(h)))
        self._h = h
        _track("pair", self)

    def set_profiling(self, flags):
        """PROF_DENSE = 1 (default), PROF_INNER = 2, PROF_REPLAY = 4 (include/pwicp.h)."""
        self._ctx._chk(self._L.pwicp_pair_set_profiling(self._h, int(flags)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.pwicp_pair_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_patches(self):
        a, b = C.c_int32(), C.c_int32()
        self._ctx._chk(self._L.pwicp_pair_num_patches(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reset(self):
        self._ctx._chk(self._L.pwicp_pair_reset(self._h))

    def run(self, check=True):
        r = Result()
        rc = self._L.pwicp_pair_run(self._h, C.byref(r))
        if check:
            self._ctx._chk(rc)
        return r

    def step(self, st):
        """PwICP_singleIteration (R.cpp:704-972) on the resident pair; `st` (Step) carries currDT / BBchange / stage flags."""
        return self._L.pwicp_pair_step(self._h, C.byref(st))

    def auto_dtinit(self):
        v = C.c_float()
        rc = self._L.pwicp_pair_auto_dtinit(self._h, C.byref(v))
        if rc != 0:
            raise PwicpError(rc, "pwicp_pair_auto_dtinit")
        return v.value

    def download_source(self):
        out = np.zeros((self.n2, 4), np.float32)
        self._ctx._chk(self._L.pwicp_pair_download_source(self._h, _p(out)))
        return out

    def num_patch_points(self):
        a, b = C.c_int32(), C.c_int32()
        self._ctx._chk(self._L.pwicp_pair_num_patch_points(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def source_patch_points(self):
        """The points of the selected source patches (current positions), float32 (tot2, 4), in the order of the patch arrays."""
        out = np.empty((self.num_patch_points()[1], 4), np.float32)
        self._ctx._chk(self._L.pwicp_pair_download_state(self._h, None, None, None, _p(out)))
        return out

    def dense_distances(self, far_group=-1):
        """Squared distance of every source patch point (order of the source patch arrays) to its nearest target point: the
        dense search of calPercentileDistBetween2PC with every patch taken as stable."""
        tot = self.num_patch_points()[1]
        out = np.empty(tot, np.float32)
        self._ctx._chk(self._L.pwicp_pair_dense_distances(self._h, int(far_group), _p(out)))
        return out

    def bench_dense_nn(self, n_launches=10):
        ms, nq, kb, edge = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        self._ctx._chk(self._L.pwicp_pair_bench_dense_nn(self._h, n_launches, C.byref(ms), C.byref(nq), C.byref(kb),
                                                         C.byref(edge)))
        return ms.value, nq.value, kb.value, edge.value


def frontend_fallback_counts():
    """pwicp_frontend_fallback_counts: how often the serial host passes of the front end took over in this process -
    {clouds, fusion_on_host, refinement_on_host, arena_doubled, fusion_host_by_env, frontend_host_by_env}."""
    L = load_library()
    v = (C.c_longlong * 6)()
    L.pwicp_frontend_fallback_counts.argtypes = [C.POINTER(C.c_longlong)]
    if L.pwicp_frontend_fallback_counts(v) != 0:
        raise PwicpError(-2, "pwicp_frontend_fallback_counts")
    names = ("clouds_on_device", "fusion_taken_over_by_host", "refinement_taken_over_by_host", "list_arena_doubled",
             "fusion_on_host_by_env", "frontend_on_host_by_env")
    return {k: int(v[i]) for i, k in enumerate(names)}


def run_pairs_concurrent(pairs, reset_first=True):
    """pwicp_pairs_run_concurrent: the registrations of independent pairs (each on a Context of its own) side by side, one host
    thread per pair inside the library; returns their Results (bit for bit what Pair.run gives for each alone)."""
    L = load_library()
    n = len(pairs)
    hs = (C.c_void_p * n)(*[p._h for p in pairs])
    res = (Result * n)()
    L.pwicp_pairs_run_concurrent.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Result), C.c_int]
    rc = L.pwicp_pairs_run_concurrent(hs, n, res, 1 if reset_first else 0)
    if rc != 0:
        raise PwicpError(rc, "pwicp_pairs_run_concurrent")
    return [res[k] for k in range(n)]
