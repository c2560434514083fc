"""Host-side setup stages of the product (preprocessing, segmentation front end, file formats) against the oracle /
the reference's own front end, and the reference's file-in/file-out entry points end to end (GPU)."""
import os

import numpy as np
import pytest

import _data
import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_preprocess_matches_oracle(oracle):
    import pwicp_amd as P
    tgt, src, _ = _data.pair(60000, reduce=False)
    rng = np.random.default_rng(0)
    cloud = np.vstack([tgt, tgt[:500] + rng.normal(0, 0.05, (500, 3)).astype(np.float32)])     # a few outliers
    for leaf, mult in ((0.01, 5.0), (0.0075, 2.7)):
        a = P.preprocess(cloud, leaf, 14, mult)
        b = oracle.sor(oracle.voxel_grid(cloud, leaf), 14, mult)
        assert a.shape == b.shape and np.array_equal(a, b)
        assert len(a) < len(cloud)


def test_voxel_grid_heap_sort_fallback_of_the_std_sort_order(oracle, monkeypatch):
    """pcl::VoxelGrid sums the points of a voxel in the order MSVC's std::sort leaves them in (host/msvc_sort.h).  Where that
    sort's depth budget runs out it heap-sorts the sub-range: unreachable with the fixtures, so the budget is forced small here
    (product: $PWICP_MSVC_SORT_BUDGET, oracle: orc_set_msvc_sort_budget).  Product == oracle bit for bit on every budget, the
    heap sort really ran, and the centroids then DIFFER in last bits from the full-budget ones (the order matters: that is why
    the fall-back is restated instead of replaced by input order)."""
    import ctypes as C
    import pwicp_amd as P
    tgt, _, _ = _data.pair(60000, reduce=False)
    rng = np.random.default_rng(3)
    cloud = np.vstack([tgt, tgt + rng.normal(0, 0.002, tgt.shape).astype(np.float32)])       # several points per voxel
    L = oracle.lib()
    L.orc_set_msvc_sort_budget.argtypes = [C.c_longlong]
    L.orc_set_msvc_sort_budget.restype = None
    L.orc_msvc_heap_used.restype = C.c_int
    full = oracle.voxel_grid(cloud, 0.02)
    outs = []
    try:
        for budget in (0, 1, 8, 200):
            L.orc_set_msvc_sort_budget(budget)
            monkeypatch.setenv("PWICP_MSVC_SORT_BUDGET", str(budget))
            b = oracle.voxel_grid(cloud, 0.02)
            a = P.preprocess(cloud, 0.02, 14, 1.0e9)            # (a multiplier no point exceeds: SOR keeps everything)
            assert L.orc_msvc_heap_used() == 1
            assert a.shape == b.shape == full.shape and np.array_equal(a, b), budget
            assert np.allclose(a, full, rtol=0, atol=1e-5)
            outs.append(a)
    finally:
        L.orc_set_msvc_sort_budget(-1)
    assert any(not np.array_equal(o, full) for o in outs)


def test_preprocess_voxel_index_overflow_passes_the_cloud_through(oracle):
    """pcl::VoxelGrid (PCL 1.8.1): a leaf whose voxel indices would overflow int32 -> warning, output = input; the
    reference then runs SOR on the full cloud (C.cpp:423-439).  Same here, and SORfilter alone
    (isDownSamp = false) is the same function."""
    import pwicp_amd as P
    tgt, _, _ = _data.pair(20000, reduce=False)
    cloud = tgt.copy()
    cloud[:50] += np.float32(3.0)                         # a few outliers and a large extent: 4 m / 1e-4 m per axis
    a = P.preprocess(cloud, 1.0e-4, 14, 2.7)
    assert np.array_equal(oracle.voxel_grid(cloud, 1.0e-4)[:, :3], cloud)
    b = oracle.sor(oracle.f4(cloud), 14, 2.7)
    assert a.shape == b.shape and np.array_equal(a, b) and len(a) < len(cloud)
    c = P.sor_filter(cloud, 14, 2.7)
    assert np.array_equal(a, c)


def test_frontend_matches_reference_front_end(oracle):
    """Product segmentation == the reference's own codelibrary front end (oracle/_ref), label for label."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    import pwicp_amd as P
    tgt, src, _ = _data.pair(20000)
    for cloud in (tgt, src):
        lab, nsv = P.frontend_segment(cloud, 0.05)
        lab_ref, nsv_ref = oracle.ref_frontend(cloud, 0.05)
        assert nsv == nsv_ref and np.array_equal(lab, lab_ref)
        assert lab.min() == 0 and lab.max() == nsv - 1


def test_frontend_on_reference_epoch(oracle):
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    p1 = P.preprocess(read_pcd(G.epoch_path(1)), 0.005, 14, 5.0)
    assert np.array_equal(p1, G.preprocess_4d(oracle, read_pcd(G.epoch_path(1))))
    r1, _, _ = G.reduce_pair(p1, p1)
    lab, nsv = P.frontend_segment(r1, 0.05)
    lab_ref, nsv_ref = oracle.ref_frontend(r1, 0.05)
    assert nsv == nsv_ref == 2073
    assert np.array_equal(lab, lab_ref)


@pytest.mark.parametrize("epoch", list(range(2, 21)))
def test_frontend_on_every_reference_epoch(oracle, epoch):
    """The serial passes of the product (the ones the device pipeline is tested against, tests/test_gpu_frontend.py) reproduce
    the reference's own compiled front end on every scan of its 4D series (here: wherever /root/reference is present; the six
    epochs kept as fixtures run everywhere)."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    path = G.epoch_path(epoch)
    if path is None:
        pytest.skip("scan not available here")
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    p = P.preprocess(read_pcd(path), 0.005, 14, 5.0)
    r, _, _ = G.reduce_pair(p, p)
    lab, nsv = P.frontend_segment(r, 0.05)
    lab_ref, nsv_ref = oracle.ref_frontend(r, 0.05)
    assert nsv == nsv_ref and np.array_equal(lab, lab_ref)


def test_pcd_roundtrip(tmp_path):
    from pwicp_amd.pcd import read_pcd, write_pcd_binary
    pts = np.random.default_rng(1).normal(size=(1000, 3)).astype(np.float32)
    f = tmp_path / "a.pcd"
    write_pcd_binary(str(f), pts)
    assert np.array_equal(read_pcd(str(f)), pts)
    # ascii variant with an extra field
    g = tmp_path / "b.pcd"
    with open(g, "w") as o:
        o.write("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                "WIDTH 3\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 3\nDATA ascii\n1 2 3 9\n4 5 6 9\n7 8 9 9\n")
    assert np.array_equal(read_pcd(str(g)), np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32))


def _write_config(path, p1, p2, res=0.005, sv=0.05, dtinit=0.05, dtmin=0.004):
    with open(path, "w") as f:      # layout of configuration_files/configuration_4d.txt (11 positional lines)
        f.write("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                "float PCres1 (m): %g\nfloat PCres2 (m): %g\nfloat SVsize1 (m): %g\nfloat SVsize2 (m): %g\n"
                "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): %g\nfloat DTmin (m): %g\nbool isVisual (yes-1, no-0): 0"
                % (p1, p2, res, res, sv, sv, dtinit, dtmin))


def test_entry_points_reject_bad_config(tmp_path):
    import pwicp_amd as P
    assert P.PiecewiseICP_pair_call(str(tmp_path / "missing.txt"), str(tmp_path) + "/") is False
    bad = tmp_path / "bad.txt"
    _write_config(bad, "a.pcd", "b.pcd", res=-1.0)                 # PCres1 <= 0 -> readConfigFile returns false
    assert P.PiecewiseICP_pair_call(str(bad), str(tmp_path) + "/") is False
    bad2 = tmp_path / "bad2.txt"
    _write_config(bad2, "a.pcd", "b.pcd", dtinit=0.001, dtmin=0.004)   # DTinit < DTmin
    assert P.PiecewiseICP_4D_call(str(bad2), 0, 2, 0, 0.75) is False


@pytest.mark.gpu
def test_4d_entry_point_reproduces_reference_result(tmp_path, ctx):
    """PiecewiseICP_4D_call (the reference's exported function) on the reference's own Epoch_001/002 files:
    result files in the reference's format, numbers equal to the reference's checked-in result."""
    import pwicp_amd as P
    out = str(tmp_path) + "/"
    cfg = tmp_path / "cfg.txt"
    _write_config(cfg, os.path.join(G.GOLD, "inputs"), out)
    cwd = os.getcwd()
    os.chdir(tmp_path)                                  # RegPairFile.txt / GT lookup are CWD-relative (R.cpp:56, 210)
    try:
        assert P.PiecewiseICP_4D_call(str(cfg), 0, 2, 0, 0.75) is True
    finally:
        os.chdir(cwd)
    for f in ("2_Direct2Ref_TransMatrix.txt", "TransMatrices.txt", "TransParameters.txt", "TransMatrices_toRef.txt",
              "TransParameters_toRef.txt"):
        assert os.path.exists(out + f), f
    T, V, stds = G.parse_transmatrix_file(out + "2_Direct2Ref_TransMatrix.txt")
    gold = os.path.join(G.GOLD, "reference_results", "2_Direct2Ref_TransMatrix.txt")
    Tg, Vg, stdg = G.parse_transmatrix_file(gold)
    assert np.abs(G.euler(T) - G.euler(Tg)).max() < 5e-6 and np.abs(T[:3, 3] - Tg[:3, 3]).max() < 5e-6
    assert np.allclose(stds, stdg, rtol=5e-3)
    # same text layout as the reference's file (labels and line structure)
    mine = [l.split("=")[0].strip() if "=" in l else l.strip() for l in open(out + "2_Direct2Ref_TransMatrix.txt")]
    ref = [l.split("=")[0].strip() if "=" in l else l.strip() for l in open(gold)]
    assert [m for m in mine if not m or m[0].isalpha() or m[0] == "4" or m[0] == "6"] == \
           [r for r in ref if not r or r[0].isalpha() or r[0] == "4" or r[0] == "6"]
    assert len(mine) == len(ref)
    # TransParameters.txt: header + one row equal to the reference's first row
    rows = open(out + "TransParameters.txt").read().strip().split("\n")
    gold_rows = open(os.path.join(G.GOLD, "reference_results", "TransParameters.txt")).read().strip().split("\n")
    assert rows[0].split() == gold_rows[0].split()
    a = np.array(rows[1].split(), float)
    b = np.array(gold_rows[1].split(), float)
    assert a[0] == b[0] == 2 and np.abs(a[1:4] - b[1:4]).max() < 5e-4 and np.abs(a[4:7] - b[4:7]).max() < 5e-6


@pytest.mark.gpu
def test_pair_entry_point(tmp_path, ctx, oracle):
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    out = str(tmp_path) + "/"
    cfg = tmp_path / "cfg_pair.txt"
    _write_config(cfg, os.path.join(G.GOLD, "inputs", "Epoch_001.pcd"), os.path.join(G.GOLD, "inputs", "Epoch_002.pcd"))
    assert P.PiecewiseICP_pair_call(str(cfg), out) is True
    T, V, stds = G.parse_transmatrix_file(out + "TransMatrix.txt")
    Tg, _, _ = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "2_Direct2Ref_TransMatrix.txt"))
    # pair path uses SOR multiplier 2.7 instead of 5.0: same registration up to the accuracy level of
    # the data set (reference result vs ground truth for this epoch: 1e-4 rad / 1.1e-3 m)
    assert np.abs(G.euler(T) - G.euler(Tg)).max() < 3e-4 and np.abs(T[:3, 3] - Tg[:3, 3]).max() < 1.5e-3
    # ... and equal to the oracle driven through the same pair path (SOR 2.7)
    if oracle.ref_frontend_available():
        p1 = oracle.sor(oracle.voxel_grid(read_pcd(os.path.join(G.GOLD, "inputs", "Epoch_001.pcd")), 0.005), 14, 2.7)
        p2 = oracle.sor(oracle.voxel_grid(read_pcd(os.path.join(G.GOLD, "inputs", "Epoch_002.pcd")), 0.005), 14, 2.7)
        r1, r2, shift = G.reduce_pair(p1, p2)
        l1, n1 = oracle.ref_frontend(r1, 0.05)
        l2, n2 = oracle.ref_frontend(r2, 0.05)
        io = oracle.run_loop(r1, r2, oracle.select_patches(r1, l1, n1), oracle.select_patches(r2, l2, n2),
                             0.005, 0.005, 0.05, 0.05, 0.05, 0.004)
        To = G.final_matrix(io.T16, shift)
        assert np.abs(G.euler(T) - G.euler(To)).max() < 1e-5 and np.abs(T[:3, 3] - To[:3, 3].astype(float)).max() < 1e-5
    src = read_pcd(os.path.join(G.GOLD, "inputs", "Epoch_002.pcd"))
    moved = read_pcd(out + "RegisteredSourceCloud.pcd")
    assert moved.shape == src.shape
    expect = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    assert np.abs(moved - expect).max() < 1e-5


@pytest.mark.gpu
def test_pair_entry_point_again_with_parked_and_recycled_contexts(tmp_path):
    """The entry point parks its context (front-end streams and work spaces) between calls and contexts that are destroyed leave
    their streams to the next ones: the files of a second and third call - after contexts have come and gone, and after the parked
    set was released - are byte-identical to the first call's."""
    import pwicp_amd as P
    cfg = tmp_path / "cfg_pair.txt"
    _write_config(cfg, os.path.join(G.GOLD, "inputs", "Epoch_001.pcd"), os.path.join(G.GOLD, "inputs", "Epoch_002.pcd"))
    outs = [str(tmp_path / ("run%d_" % k)) for k in range(3)]
    assert P.PiecewiseICP_pair_call(str(cfg), outs[0]) is True
    for _ in range(3):                                   # contexts come and go (their streams are recycled)
        cs = [P.Context(0) for _ in range(3)]
        for c in cs:
            c.close()
    assert P.PiecewiseICP_pair_call(str(cfg), outs[1]) is True          # on the parked context
    P.series_release_parked()
    assert P.PiecewiseICP_pair_call(str(cfg), outs[2]) is True          # on a new one
    ref_t = open(outs[0] + "TransMatrix.txt", "rb").read()
    ref_c = open(outs[0] + "RegisteredSourceCloud.pcd", "rb").read()
    for o in outs[1:]:
        assert open(o + "TransMatrix.txt", "rb").read() == ref_t
        assert open(o + "RegisteredSourceCloud.pcd", "rb").read() == ref_c


@pytest.mark.gpu
def test_gpu_knn_and_frontend_match_host(ctx, oracle):
    """k-NN graph on the GPU == host KD-tree lists (order included); labels == host front end == reference front end."""
    import pwicp_amd as P
    from scipy.spatial import cKDTree
    tgt, src, _ = _data.pair(30000)
    nb = ctx.knn(src, 45, 2 * _data.R)
    assert nb.shape == (len(src), 45) and np.array_equal(nb[:, 0], np.arange(len(src)))
    d, ii = cKDTree(src.astype(np.float64)).query(src.astype(np.float64), k=45)
    # same neighbour sets and same order up to exact distance ties
    dn = np.linalg.norm(src[nb].astype(np.float64) - src[:, None, :].astype(np.float64), axis=2)
    assert np.allclose(dn, d, rtol=0, atol=1e-12) and np.all(np.diff(dn, axis=1) >= 0)
    assert (nb != ii).mean() < 1e-3
    lab_g, n_g = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
    lab_h, n_h = P.frontend_segment(src, 10 * _data.R)
    assert n_g == n_h and np.array_equal(lab_g, lab_h)
    if oracle.ref_frontend_available():
        lab_r, n_r = oracle.ref_frontend(src, 10 * _data.R)
        assert n_g == n_r and np.array_equal(lab_g, lab_r)
    # sparse / clustered input: k larger than a cell block holds -> the search radius must grow
    rng = np.random.default_rng(3)
    pts = np.vstack([rng.normal(0, 0.01, (300, 3)), rng.normal(1.0, 0.3, (200, 3))]).astype(np.float32)
    nb2 = ctx.knn(pts, 45)
    d2, i2 = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=45)
    dn2 = np.linalg.norm(pts[nb2].astype(np.float64) - pts[:, None, :].astype(np.float64), axis=2)
    assert np.allclose(dn2, d2, rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_gpu_preprocess_matches_host(ctx, oracle):
    """VoxelGrid + SOR on the GPU == host stage == oracle, point for point (order and bits)."""
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    cases = []
    tgt, src, _ = _data.pair(60000)
    rng = np.random.default_rng(11)
    noisy = np.vstack([src[:, :3], src[:500, :3] + rng.normal(0, 0.3, (500, 3)).astype(np.float32)]).astype(np.float32)
    cases.append((noisy, 1.5 * _data.R, 14, 2.7))
    cases.append((tgt[:, :3] + np.float32(1234.5), 2.0 * _data.R, 14, 5.0))           # unreduced coordinates
    g = os.path.join(os.path.dirname(__file__), "golden", "inputs", "Epoch_001.pcd")
    cases.append((read_pcd(g)[:, :3], 0.02, 14, 5.0))
    cases.append((rng.normal(0, 1, (40, 3)).astype(np.float32), 0.05, 14, 1.0))        # fewer voxels than a block
    for cloud, leaf, k, mult in cases:
        got = ctx.preprocess(cloud, leaf, k, mult)
        want = P.preprocess(cloud, leaf, k, mult)
        assert got.shape == want.shape and got.shape[0] > 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # SORfilter alone (PCpreprocessing with isDownSamp = false) and the voxel-index-overflow pass-through (PCL semantics)
    c0 = cases[0][0]
    want = P.sor_filter(c0, 14, 2.7)
    assert np.array_equal(ctx.sor_filter(c0, 14, 2.7).view(np.uint32), want.view(np.uint32))
    assert np.array_equal(ctx.sor_filter(c0, 14, 2.7, 0.005).view(np.uint32), want.view(np.uint32))
    assert np.array_equal(ctx.preprocess(c0, 1.0e-5, 14, 2.7).view(np.uint32), want.view(np.uint32))
    o = oracle.sor(oracle.voxel_grid(cases[0][0], cases[0][1]), 14, 2.7)
    assert np.array_equal(ctx.preprocess(*cases[0])[:, :3].view(np.uint32), np.ascontiguousarray(o[:, :3]).view(np.uint32))


@pytest.mark.gpu
def test_4d_series_reuses_target_and_auto_spacing(tmp_path, ctx, oracle):
    """Three-epoch Direct2Ref series (epoch 3 = a copy of epoch 2): the second pair runs with the cached target
    (preprocessed cloud + supervoxels) and must give exactly the first pair's result; then the same series with the
    point spacing estimated by calPCresolution on the GPU (isSetResSVsize = 0)."""
    import shutil
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    inp = tmp_path / "in"
    inp.mkdir()
    src = os.path.join(G.GOLD, "inputs")
    shutil.copy(os.path.join(src, "Epoch_001.pcd"), inp / "Epoch_001.pcd")
    shutil.copy(os.path.join(src, "Epoch_002.pcd"), inp / "Epoch_002.pcd")
    shutil.copy(os.path.join(src, "Epoch_002.pcd"), inp / "Epoch_003.pcd")
    out = str(tmp_path) + "/o_"
    cfg = tmp_path / "cfg.txt"
    _write_config(cfg, str(inp), out)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert P.PiecewiseICP_4D_call(str(cfg), 0, 3, 0, 0.75) is True
    finally:
        os.chdir(cwd)
    a = open(out + "2_Direct2Ref_TransMatrix.txt").read()
    b = open(out + "3_Direct2Ref_TransMatrix.txt").read()
    assert a == b
    T, _, _ = G.parse_transmatrix_file(out + "2_Direct2Ref_TransMatrix.txt")
    Tg, _, _ = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "2_Direct2Ref_TransMatrix.txt"))
    assert np.abs(G.euler(T) - G.euler(Tg)).max() < 5e-6
    # point spacing on the GPU == host == oracle, bit for bit
    c = read_pcd(os.path.join(src, "Epoch_001.pcd"))
    r_gpu, r_host = ctx.pc_resolution(c), P.pc_resolution(c)
    assert np.float32(r_gpu) == np.float32(r_host) == np.float32(oracle.pc_resolution(c))
    assert 0.001 < r_gpu < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("pair_mode", [0, -1, 2])
def test_series_driver_two_ranks_matches_single_process(tmp_path, ctx, pair_mode):
    """pwicp_amd.series (pairs sharded over two ranks, records all-gathered, rank 0 writes) produces the same files,
    byte for byte, as the exported single-process PiecewiseICP_4D_call.  Both ranks share GPU 0 (gloo), which
    exercises everything but the RCCL transport."""
    import shutil
    import socket
    import subprocess
    import sys
    import pwicp_amd as P
    inp = tmp_path / "in"
    inp.mkdir()
    src = os.path.join(G.GOLD, "inputs")
    shutil.copy(os.path.join(src, "Epoch_001.pcd"), inp / "Epoch_001.pcd")
    shutil.copy(os.path.join(src, "Epoch_002.pcd"), inp / "Epoch_002.pcd")
    shutil.copy(os.path.join(src, "Epoch_002.pcd"), inp / "Epoch_003.pcd")
    shutil.copy(os.path.join(src, "Epoch_001.pcd"), inp / "Epoch_004.pcd")
    outs = []
    for tag in ("single", "sharded"):
        d = tmp_path / tag
        d.mkdir()
        out = str(d) + "/"
        cfg = d / "cfg.txt"
        _write_config(cfg, str(inp), out)
        if tag == "single":
            cwd = os.getcwd()
            os.chdir(d)
            try:
                assert P.PiecewiseICP_4D_call(str(cfg), 0, 4, pair_mode, 0.75) is True
            finally:
                os.chdir(cwd)
        else:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            env = dict(os.environ)
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env["PYTHONPATH"] = os.path.join(root, "piecewise-icp_amd") + os.pathsep + env.get("PYTHONPATH", "")
            res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pwicp_amd.series",
                                  str(cfg), "0", "4", str(pair_mode), "0.75", "--backend", "gloo", "--single-device"],
                                 capture_output=True, text=True, timeout=600, cwd=str(d), env=env)
            assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
        outs.append(out)
    mode = "Direct2Ref" if pair_mode == 0 else ("Adaptive" if pair_mode < 0 else "Fixed")
    names = ["TransMatrices.txt", "TransParameters.txt", "TransMatrices_toRef.txt", "TransParameters_toRef.txt"] + \
            ["%d_%s_TransMatrix.txt" % (e, mode) for e in (2, 3, 4)]
    for f in names:
        a, b = open(outs[0] + f).read(), open(outs[1] + f).read()
        assert a == b and len(a) > 50, f
    if pair_mode < 0:
        assert open(str(tmp_path / "single" / "RegPairFile.txt")).read() == open(str(tmp_path / "sharded" / "RegPairFile.txt")).read()


@pytest.mark.gpu
@pytest.mark.parametrize("pair_mode", [0, -1])
def test_cpp_multi_device_and_rccl_paths_match_single_device(tmp_path, ctx, pair_mode):
    """C++-native multi-GPU inside libpwicp.so: (a) PiecewiseICP_4D_call with PWICP_DEVICES=0,0 — two
    workers (context + host thread each) sharing the one GPU of this box, pairs dealt round-robin, the shared target built
    per worker; (b) pwicp_series_run_distributed at world 1 — the RCCL all-gather / broadcast path through librccl
    (ncclCommInitRank with one rank).  Both must write the files of the plain single-device call, byte for byte."""
    import shutil
    import subprocess
    import sys
    import pwicp_amd as P
    inp = tmp_path / "in"
    inp.mkdir()
    src = os.path.join(G.GOLD, "inputs")
    for k, e in enumerate((1, 2, 8, 2, 1), start=1):
        shutil.copy(os.path.join(src, "Epoch_%03d.pcd" % e), inp / ("Epoch_%03d.pcd" % k))
    outs = {}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tag in ("single", "two_workers", "rccl_world1"):
        d = tmp_path / tag
        d.mkdir()
        out = str(d) + "/"
        cfg = d / "cfg.txt"
        _write_config(cfg, str(inp), out)
        env = dict(os.environ)
        env["PYTHONPATH"] = os.path.join(root, "piecewise-icp_amd") + os.pathsep + env.get("PYTHONPATH", "")
        if tag == "single":
            env["PWICP_DEVICES"] = "0"
            code = "import pwicp_amd as P, sys; sys.exit(0 if P.PiecewiseICP_4D_call(%r, 0, 5, %d, 0.75) else 1)" % (str(cfg), pair_mode)
        elif tag == "two_workers":
            env["PWICP_DEVICES"] = "0,0"
            code = "import pwicp_amd as P, sys; sys.exit(0 if P.PiecewiseICP_4D_call(%r, 0, 5, %d, 0.75) else 1)" % (str(cfg), pair_mode)
        else:
            code = ("import pwicp_amd as P, sys; sys.exit(0 if P.series_run_distributed(%r, 0, 5, %d, 0.75, 0, 1, 0, %r) else 1)"
                    % (str(cfg), pair_mode, str(d / "rccl.id")))
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(d), env=env)
        assert res.returncode == 0, tag + ": " + res.stdout[-2000:] + res.stderr[-2000:]
        outs[tag] = out
    mode = "Direct2Ref" if pair_mode == 0 else "Adaptive"
    names = ["TransMatrices.txt", "TransParameters.txt", "TransMatrices_toRef.txt", "TransParameters_toRef.txt"] + \
            ["%d_%s_TransMatrix.txt" % (e, mode) for e in (2, 3, 4, 5)]
    for f in names:
        a = open(outs["single"] + f).read()
        assert len(a) > 50, f
        assert a == open(outs["two_workers"] + f).read(), "two workers: " + f
        assert a == open(outs["rccl_world1"] + f).read(), "rccl: " + f


def test_pcd_with_non_finite_points(tmp_path):
    """PCL keeps NaN points of a file and every consumer on this path skips them; the reader drops them."""
    from pwicp_amd.pcd import read_pcd
    import pwicp_amd as P
    g = tmp_path / "n.pcd"
    with open(g, "w") as o:
        o.write("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\n"
                "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 4\nDATA ascii\n1 2 3\nnan nan nan\n4 5 6\n7 inf 9\n")
    # the product's C++ reader is exercised through the pair entry point's loader: the config points at this file twice;
    # two valid points are too few to register, so the call returns False — but it must get past the loader cleanly
    cfg = tmp_path / "cfg.txt"
    _write_config(cfg, str(g), str(g))
    assert P.PiecewiseICP_pair_call(str(cfg), str(tmp_path) + "/x_") is False
    a = read_pcd(str(g))                                   # the python-side reader of the test utilities keeps the raw rows
    assert a.shape[0] == 4 and np.isnan(a[1]).all()


def test_composition_and_error_entry_points_reproduce_reference_files(tmp_path):
    """The C entry points behind calTransToReferenceEpoch (R.cpp:977-1153) and calAbsErrorOfTransPara (R.cpp:1157-1251), no GPU
    needed: from the reference's own TransMatrices.txt + the adaptive pair map recovered in SURVEY 4 they must reproduce the
    reference's TransMatrices_toRef.txt, and from that file + the ground truth its TransPara_AbsError.txt."""
    import ctypes as C
    import pwicp_amd as P
    from test_distributed_cpu import _read_matrices, ADAPTIVE_REL
    L = P.load_library()
    gold = os.path.join(G.GOLD, "reference_results")
    pair_file = tmp_path / "RegPairFile.txt"
    pair_file.write_text("".join("%d %d\n" % (k, ADAPTIVE_REL[k]) for k in range(1, 20)))
    n = 19
    stamps = (C.c_int32 * n)()
    T = (C.c_float * (16 * n))()
    V = (C.c_double * (36 * n))()
    L.pwicp_trans_to_reference_epoch.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p,
                                                 C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)]
    out_tm, out_tp = str(tmp_path / "toRef.txt"), str(tmp_path / "toRefPara.txt")
    rc = L.pwicp_trans_to_reference_epoch(os.path.join(gold, "TransMatrices.txt").encode(), -1, str(pair_file).encode(), n,
                                          out_tm.encode(), out_tp.encode(), stamps, T, V)
    assert rc == 0 and list(stamps) == list(range(2, 21))
    # byte for byte (the reference's files were written on Windows: CRLF line ends)
    for mine, theirs in ((out_tm, "TransMatrices_toRef.txt"), (out_tp, "TransParameters_toRef.txt")):
        assert open(mine, "rb").read() == open(os.path.join(gold, theirs), "rb").read().replace(b"\r", b""), theirs
    Tr, Vr = _read_matrices(os.path.join(gold, "TransMatrices_toRef.txt"), n)
    Tm, Vm = _read_matrices(out_tm, n)
    for i in range(n):
        assert np.abs(Tm[i] - Tr[i]).max() < 5e-6 and np.allclose(Vm[i], Vr[i], rtol=2e-3, atol=3e-12)
        assert np.array_equal(np.array(T[16 * i:16 * i + 16], np.float32).reshape(4, 4), Tm[i])
    # a missing pair file is an error code, not an exit()
    assert L.pwicp_trans_to_reference_epoch(os.path.join(gold, "TransMatrices.txt").encode(), -1, b"/nonexistent/pairs.txt", n,
                                            out_tm.encode(), out_tp.encode(), stamps, T, V) != 0
    L.pwicp_abs_error_of_trans_para.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    err = str(tmp_path / "err.txt")
    assert L.pwicp_abs_error_of_trans_para(os.path.join(gold, "TransMatrices_toRef.txt").encode(),
                                           os.path.join(gold, "defined_transformations.txt").encode(), 20, 0, err.encode()) == 0
    assert open(err, "rb").read() == open(os.path.join(gold, "TransPara_AbsError.txt"), "rb").read().replace(b"\r", b"")
    a = np.loadtxt(err, skiprows=1)
    b = np.loadtxt(os.path.join(gold, "TransPara_AbsError.txt"), skiprows=1)
    assert a.shape == b.shape == (19, 6)
    assert np.allclose(a, b, rtol=2e-3, atol=2e-3)          # mgon / mm, printed with 6 significant digits by the reference


def test_result_file_writers_reproduce_the_references_files_byte_for_byte(tmp_path):
    """Layout known-answer test of the text writers (R.cpp:341-388 / 492-539, 152-167): the reference's own numbers, parsed from
    its 57 checked-in <e>_{Direct2Ref,Adaptive,Fixed}_TransMatrix.txt files, written again by the product's writer.  Every line
    must come back byte for byte (CRLF aside) - labels, blank lines, trailing blanks, `fixed` with 12 / 10 digits, the gon
    conversion of matrix2angle - except the six Std_ lines, whose inputs (the full-precision VCM) the files only hold to 12
    decimals: there the label / unit / digit layout must match and the value agree to 3 digits.  TransMatrices.txt (id line,
    4 + 6 rows per pair) is reproduced in full through the direct-to-reference composition (a pass-through in pairMode 0)."""
    import ctypes as C
    import re
    import pwicp_amd as P
    L = P.load_library()
    gold = os.path.join(G.GOLD, "reference_results")
    L.pwicp_write_trans_matrix_file.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    out = str(tmp_path / "tm.txt")
    std_line = re.compile(rb"^Std_(R|t)[xyz] = \d+\.\d{10} (mgon|mm)$")
    for mode in ("Direct2Ref", "Adaptive", "Fixed"):
        for e in range(2, 21):
            f = os.path.join(gold, "%d_%s_TransMatrix.txt" % (e, mode))
            T, V, _ = G.parse_transmatrix_file(f)
            T32 = np.ascontiguousarray(T, np.float32).reshape(16)
            V64 = np.ascontiguousarray(V, np.float64).reshape(36)
            assert L.pwicp_write_trans_matrix_file(out.encode(), T32.ctypes.data_as(C.POINTER(C.c_float)),
                                                   V64.ctypes.data_as(C.POINTER(C.c_double))) == 0
            theirs = open(f, "rb").read().replace(b"\r", b"").split(b"\n")
            mine = open(out, "rb").read().split(b"\n")
            assert len(mine) == len(theirs) == 31
            n_std = 0
            for a, b in zip(mine, theirs):
                if b.startswith(b"Std_"):
                    n_std += 1
                    assert std_line.match(a) and std_line.match(b) and a.split(b"=")[0] == b.split(b"=")[0] and a.split()[-1] == b.split()[-1]
                    assert abs(float(a.split()[2]) / float(b.split()[2]) - 1) < 2e-3
                else:
                    assert a == b, (mode, e, a, b)
            assert n_std == 6
    # TransMatrices.txt through reader + writer (pairMode 0: the matrices to the reference epoch are the matrices themselves)
    n = 19
    stamps = (C.c_int32 * n)()
    T = (C.c_float * (16 * n))()
    V = (C.c_double * (36 * n))()
    L.pwicp_trans_to_reference_epoch.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p,
                                                 C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)]
    out_tm, out_tp = str(tmp_path / "tms.txt"), str(tmp_path / "tps.txt")
    assert L.pwicp_trans_to_reference_epoch(os.path.join(gold, "TransMatrices.txt").encode(), 0, b"", n, out_tm.encode(),
                                            out_tp.encode(), stamps, T, V) == 0
    assert open(out_tm, "rb").read() == open(os.path.join(gold, "TransMatrices.txt"), "rb").read().replace(b"\r", b"")
    # TransParameters.txt: header and row layout (its sigmas come from full-precision VCMs the text files do not hold)
    theirs = open(os.path.join(gold, "TransParameters.txt"), "rb").read().replace(b"\r", b"").split(b"\n")
    mine = open(out_tp, "rb").read().split(b"\n")
    assert mine[0] == theirs[0] and len(mine) == len(theirs)
    row = re.compile(rb"^\d+( -?\d+\.\d{10}){12}$")
    for a, b in zip(mine[1:20], theirs[1:20]):
        assert row.match(a) and row.match(b) and a.split()[:7] == b.split()[:7]


def test_host_thread_pool_respects_the_cpu_share_of_a_rank():
    """VERDICT r4 item 3b (host/parallel.h): the library's host-thread pool is sized by the CPUs the PROCESS may really use - the
    affinity mask, cut by the cgroup CPU quota - divided by $LOCAL_WORLD_SIZE; $PWICP_HOST_THREADS overrides.  (Round 4 took
    hardware_concurrency() capped at 32, whatever the quota and however many ranks shared the node.)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import pwicp_amd as P; print(P.load_library().pwicp_host_threads())"
            % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "piecewise-icp_amd"))

    def ask(env_extra, cpus=None):
        env = {k: v for k, v in os.environ.items() if k not in ("PWICP_HOST_THREADS", "LOCAL_WORLD_SIZE")}
        env.update(env_extra)
        cmd = [sys.executable, "-c", code]
        if cpus is not None:
            cmd = ["taskset", "-c", cpus] + cmd
        return int(subprocess.run(cmd, env=env, capture_output=True, text=True, check=True).stdout.split()[-1])

    usable = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            usable = min(usable, max(1, int(int(q) / int(per) + 0.5)))
    except OSError:
        pass
    base = ask({})
    assert base == max(1, min(usable, 32))
    assert ask({"LOCAL_WORLD_SIZE": "8"}) == max(1, min(int(usable / 8 + 0.5), 32))
    assert ask({"PWICP_HOST_THREADS": "5", "LOCAL_WORLD_SIZE": "8"}) == 5
    if usable >= 2:
        assert ask({}, cpus="0-1") <= 2
        assert ask({"LOCAL_WORLD_SIZE": "2"}, cpus="0-1") == 1
