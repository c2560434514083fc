"""BASELINE.json configs on the GPU, each against the oracle (through the C ABI):
  configs[1]  1 M-point pair, product supervoxel labels        -> bit-exact loop parity with the oracle
  configs[0]+ the flip-sensitive golden pairs e8/e11/e13/e19   -> GPU == oracle, GPU vs the reference's result files
  configs[3]  8 pairs x 1 M points against one shared target   -> every record == stand-alone pair == oracle
  configs[4]  4 pairs x 5 M points streamed, shared target      -> every record == stand-alone pair; one vs the oracle
              one 5 M-point pair                               -> oracle loop parity + sampled brute-force NN
  the reference's own run (main.cpp) through the exported entry point, pairMode 0 / -1 / 3 -> all 57 result files
"""
import json
import os

import numpy as np
import pytest

import _data
import _golden as G
from test_gpu_parity import _assert_loop_parity

pytestmark = pytest.mark.gpu

R = _data.R


def _oracle_loop(oracle, tgt, l1, n1, src, l2, n2, r=R, sv=10 * R, dtinit=10 * R, dtmin=0.8 * R):
    P1 = oracle.select_patches(tgt, l1, n1)
    P2 = oracle.select_patches(src, l2, n2)
    return oracle.run_loop(tgt, src, P1, P2, r, r, sv, sv, dtinit, dtmin)


def test_loop_parity_1m_supervoxel_labels(ctx, oracle):
    """BASELINE configs[1] at full size with the PRODUCT's supervoxel labels (what bench.py runs): DT series, stable
    counts, d75, LoD, maxBB, inner-iteration counts bit-exact against the oracle; T within north_star's tolerance."""
    import pwicp_amd as P
    tgt, src, _ = _data.pair(1000000)
    l1, n1 = ctx.frontend_segment(tgt, 10 * R, 45, R)
    l2, n2 = ctx.frontend_segment(src, 10 * R, 45, R)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
    res = pair.run(check=False)
    io = _oracle_loop(oracle, tgt, l1, n1, src, l2, n2)
    _assert_loop_parity(res, io)
    assert res.n_dense_nn_launches >= 1 and res.n_corr_dense > 500000
    moved = pair.download_source()
    c2 = oracle.f4(src).copy()
    for i in range(io.n_outer):
        oracle.lib().orc_transform_points(oracle._p(c2), len(c2), oracle._p(np.array(io.Tk[i], np.float32)))
    assert np.array_equal(moved[:, :3], c2[:, :3])
    pair.close()


# tolerance of every result file of the reference (rad, m): tests/golden/tolerance_table.json, the table the oracle is held to
with open(os.path.join(G.GOLD, "tolerance_table.json")) as _f:
    _TT = json.load(_f)
TOL = {m: {int(e): tuple(v) for e, v in t.items()} for m, t in _TT["tol"].items()}
GOLD_TOL = TOL["Direct2Ref"]
AMAP = {int(e): t for e, t in _TT["pair_map"]["Adaptive"].items()}
FMAP = {int(e): t for e, t in _TT["pair_map"]["Fixed"].items()}
# a11: [relative tolerance of the six printed sigmas, absolute tolerance of the 36 printed VCM entries] per file, same table
STOL = {m: {int(e): tuple(v) for e, v in t.items()} for m, t in _TT["sigma_vcm_tol"].items()}
# the series files of the run (pairMode -1) against the reference's checked-in ones: twice the distances measured on the GPU
# (profiles/r05_entry_point_distances.json), floored at print precision.  Matrices: (entry, VCM entry relative to the largest of
# its matrix); parameter files: (angles in gon, translations in m, sigmas relative); error report: (mgon, mm)
COMPOSED_TOL = {"TransMatrices.txt": (2.5e-7, 2e-4), "TransMatrices_toRef.txt": (2.5e-7, 1e-4),
                "TransParameters.txt": (5e-6, 2.5e-7, 2e-5), "TransParameters_toRef.txt": (7e-6, 2.5e-7, 2e-5),
                "TransPara_AbsError.txt": (7e-3, 3e-4)}
# (measured: one float ulp, 1.2e-7, on the matrix entries of both files; 2.3e-6 / 3.2e-6 gon = 3.6e-8 / 5e-8 rad on the angles;
#  1.2e-7 m; sigmas 7e-6 relative; 3.2e-3 mgon / 1.2e-4 mm on the error report, which prints six significant digits)


def _record_distances(name, d):
    """Measured distances to the reference's files, kept beside the run (tools/publish_profiles.sh copies them to profiles/)."""
    dst = os.path.join(os.path.dirname(G.HERE), "gpurun_out")
    if os.path.isdir(dst):
        with open(os.path.join(dst, "r05_distances_%s.json" % name), "w") as f:
            json.dump({str(k): v for k, v in d.items()}, f, indent=1)


@pytest.mark.parametrize("epoch", list(range(2, 21)))
def test_golden_pairs_through_gpu(ctx, oracle, epoch):
    """Every pair of the reference's synthetic 4D series (Direct2Ref: epoch 1 against epoch e), among them the pairs that end
    with <= 65-300 stable patches (e8, e11, e13, e19) where one patch classified differently moves the result.
    GPU == oracle on every discrete quantity, and the GPU result within the file's entry of tests/golden/tolerance_table.json
    (float print precision for all 19)."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    p1 = G.preprocess_4d(oracle, read_pcd(G.epoch_path(1)))
    p2 = G.preprocess_4d(oracle, read_pcd(G.epoch_path(epoch)))
    r1, r2, shift = G.reduce_pair(p1, p2)
    l1, n1 = oracle.ref_frontend(r1, 0.05)
    l2, n2 = oracle.ref_frontend(r2, 0.05)
    pair = P.Pair(ctx, r1, l1, n1, r2, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
    res = pair.run(check=False)
    io = _oracle_loop(oracle, r1, l1, n1, r2, l2, n2, r=0.005, sv=0.05, dtinit=0.05, dtmin=0.004)
    _assert_loop_parity(res, io)
    Tf = G.final_matrix(res.T16, shift)
    Tg, _, _ = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "%d_Direct2Ref_TransMatrix.txt" % epoch))
    assert np.abs(G.euler(Tf) - G.euler(Tg)).max() < GOLD_TOL[epoch][0]
    assert np.abs(Tf[:3, 3].astype(float) - Tg[:3, 3]).max() < GOLD_TOL[epoch][1]
    pair.close()


def _mat4_mul_f32(A, B):
    """float 4x4 product in the element order of the product's mat4_mul (Eigen's order for Matrix4f)."""
    A = np.asarray(A, np.float32).reshape(4, 4)
    B = np.asarray(B, np.float32).reshape(4, 4)
    out = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            s = np.float32(A[i, 0] * B[0, j])
            for k in range(1, 4):
                s = np.float32(s + np.float32(A[i, k] * B[k, j]))
            out[i, j] = s
    return out


def _final_matrix_exact(T16, shift):
    """T_final = S^-1 * T * S (R.cpp:461) with sequential float sums."""
    S = np.eye(4, dtype=np.float32); S[:3, 3] = shift
    Si = np.eye(4, dtype=np.float32); Si[:3, 3] = np.float32(-1) * shift
    return _mat4_mul_f32(_mat4_mul_f32(Si, T16), S)


def _write_series_config(path, p1, p2, res=R, sv=10 * R, dtinit=10 * R, dtmin=0.8 * R):
    with open(path, "w") as f:      # layout of configuration_files/configuration_4d.txt (11 positional lines)
        f.write("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                "float PCres1 (m): %.9g\nfloat PCres2 (m): %.9g\nfloat SVsize1 (m): %.9g\nfloat SVsize2 (m): %.9g\n"
                "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): %.9g\nfloat DTmin (m): %.9g\nbool isVisual (yes-1, no-0): 0"
                % (p1, p2, res, res, sv, sv, dtinit, dtmin))


def test_series_8_pairs_1m_shared_target(tmp_path, ctx, oracle):
    """BASELINE configs[3] on one GPU: reference epoch + 8 source epochs of 1 M points (SURVEY 8d cfg 4), Direct2Ref,
    through pwicp_series_run_pairs with the device-side target shared by all pairs.  Every record must equal a
    stand-alone pwicp_pair_create run on the same preprocessed clouds, which must match the oracle."""
    import pwicp_amd as P
    from pwicp_amd import synth
    from pwicp_amd.pcd import write_pcd_binary
    n = 1000000
    inp = tmp_path / "scans"
    inp.mkdir()
    tgt, _ = synth.make_tile(n, R)
    write_pcd_binary(str(inp / "Epoch_001.pcd"), tgt)
    srcs = []
    for e in range(1, 9):
        s, _ = synth.make_source(n, R, epoch=e)
        srcs.append(s)
        write_pcd_binary(str(inp / ("Epoch_%03d.pcd" % (e + 1))), s)
    out = str(tmp_path) + "/res_"
    cfg = tmp_path / "cfg.txt"
    _write_series_config(cfg, str(inp), out)
    with P.Series(str(cfg), 0, 9, 0, 0.75, 0) as series:
        assert series.num_pairs == 8
        recs = series.run_pairs(list(range(8)))
        series.write_results(recs)
    assert np.all(recs["status"] == 0) and list(recs["pair"]) == list(range(8))
    assert os.path.exists(out + "TransMatrices_toRef.txt") and os.path.exists(out + "9_Direct2Ref_TransMatrix.txt")
    # stand-alone: the same preprocessing (VoxelGrid + SOR 5.0), reduction, labels; a pair of its own per source epoch
    p1 = ctx.preprocess(tgt, R, 14, 5.0)
    r1, _, shift = G.reduce_pair(p1, p1)
    l1, n1 = ctx.frontend_segment(r1, 10 * R, 45, R)
    for k in (0, 3, 7):
        p2 = ctx.preprocess(srcs[k], R, 14, 5.0)
        r2 = p2.copy()
        r2[:, :3] = (p2[:, :3] + shift[None, :]).astype(np.float32)
        l2, n2 = ctx.frontend_segment(r2, 10 * R, 45, R)
        pair = P.Pair(ctx, r1, l1, n1, r2, l2, n2, _data.params())
        res = pair.run()
        pair.close()
        Tf = _final_matrix_exact(res.T16, shift)
        assert np.array_equal(Tf.reshape(16), recs["T"][k])                         # shared target == own target, bit for bit
        assert np.array_equal(np.array(res.VCM), recs["VCM"][k])
        assert int(recs["n_outer"][k]) == res.n_outer and int(recs["n_corr"][k]) == res.n_corr
        io = _oracle_loop(oracle, r1, l1, n1, r2, l2, n2)
        _assert_loop_parity(res, io)


def test_two_workers_of_one_process_both_take_the_supplied_target_labels(tmp_path, ctx):
    """ADVICE r5 (host/registration.cpp target_labels): a rank that EXPECTS the labels of the shared target from another rank and runs
    two workers (pwicp_series_set_devices: every worker prepares the target on its own device) used to hand the supplied labels to the
    first worker only - the second waited out $PWICP_LABEL_TIMEOUT_S (600 s) and then segmented the target itself.  Both workers must
    take them (received == 2, segmented == 0), quickly, and the records must equal the ones of a plain run."""
    import threading
    import time
    import pwicp_amd as P
    from pwicp_amd import synth
    from pwicp_amd.pcd import write_pcd_binary
    n = 120000
    inp = tmp_path / "scans"
    inp.mkdir()
    tgt, _ = synth.make_tile(n, R)
    write_pcd_binary(str(inp / "Epoch_001.pcd"), tgt)
    for e in range(1, 5):
        s, _ = synth.make_source(n, R, epoch=e)
        write_pcd_binary(str(inp / ("Epoch_%03d.pcd" % (e + 1))), s)
    cfg = tmp_path / "cfg.txt"
    _write_series_config(cfg, str(inp), str(tmp_path) + "/res_")
    with P.Series(str(cfg), 0, 5, 0, 0.75, 0) as plain:
        want = plain.run_pairs([0, 1, 2, 3])
    # the labels "another rank" would broadcast: the target preprocessed and segmented the way the series does it
    p1 = ctx.preprocess(tgt, R, 14, 5.0)
    r1, _, _ = G.reduce_pair(p1, p1)
    lab, nsv = ctx.frontend_segment(r1, 10 * R, 45, R)
    os.environ["PWICP_LABEL_TIMEOUT_S"] = "120"
    try:
        with P.Series(str(cfg), 0, 5, 0, 0.75, 0) as series:
            series.set_devices([0, 0])
            series.expect_target_labels(0)
            th = threading.Thread(target=lambda: (time.sleep(0.5), series.supply_target_labels(0, lab, nsv)))
            th.start()
            t0 = time.time()
            recs = series.run_pairs([0, 1, 2, 3])
            wall = time.time() - t0
            th.join()
            received, segmented = series.target_label_counts()
    finally:
        os.environ.pop("PWICP_LABEL_TIMEOUT_S", None)
    assert wall < 60.0, wall                       # (one worker waiting out the time-out would take 120 s)
    assert (received, segmented) == (2, 0), (received, segmented)
    assert np.all(recs["status"] == 0)
    assert recs["T"].tobytes() == want["T"].tobytes() and recs["VCM"].tobytes() == want["VCM"].tobytes()


def test_series_4_pairs_5m_streamed(tmp_path, ctx, oracle):
    """BASELINE configs[4] shape on one GPU: what each GPU of the 8-GPU run gets - a reference epoch and 4 source epochs of
    5 M points (L = 11.2 m), Direct2Ref, streamed through pwicp_series_run_pairs two pairs per window ($PWICP_SERIES_WINDOW) so
    that a window boundary is crossed with the device-side target kept.  Every record must equal a stand-alone pair on the
    same preprocessed clouds; one of them is checked against the oracle."""
    import pwicp_amd as P
    from pwicp_amd import synth
    from pwicp_amd.pcd import write_pcd_binary
    n = 5000000
    inp = tmp_path / "scans"
    inp.mkdir()
    tgt, _ = synth.make_tile(n, R)
    write_pcd_binary(str(inp / "Epoch_001.pcd"), tgt)
    srcs = {}
    for e in range(1, 5):
        s, _ = synth.make_source(n, R, epoch=e)
        if e in (1, 4):
            srcs[e - 1] = s
        write_pcd_binary(str(inp / ("Epoch_%03d.pcd" % (e + 1))), s)
        del s
    out = str(tmp_path) + "/res_"
    cfg = tmp_path / "cfg.txt"
    _write_series_config(cfg, str(inp), out)
    os.environ["PWICP_SERIES_WINDOW"] = "2"
    try:
        with P.Series(str(cfg), 0, 5, 0, 0.75, 0) as series:
            assert series.num_pairs == 4
            recs = series.run_pairs(list(range(4)))
            series.write_results(recs)
    finally:
        os.environ.pop("PWICP_SERIES_WINDOW", None)
    assert np.all(recs["status"] == 0) and list(recs["pair"]) == list(range(4))
    assert os.path.exists(out + "TransMatrices_toRef.txt") and os.path.exists(out + "5_Direct2Ref_TransMatrix.txt")
    p1 = ctx.preprocess(tgt, R, 14, 5.0)
    r1, _, shift = G.reduce_pair(p1, p1)
    l1, n1 = ctx.frontend_segment(r1, 10 * R, 45, R)
    for k in (0, 3):
        p2 = ctx.preprocess(srcs[k], R, 14, 5.0)
        r2 = p2.copy()
        r2[:, :3] = (p2[:, :3] + shift[None, :]).astype(np.float32)
        l2, n2 = ctx.frontend_segment(r2, 10 * R, 45, R)
        pair = P.Pair(ctx, r1, l1, n1, r2, l2, n2, _data.params())
        res = pair.run()
        pair.close()
        assert np.array_equal(_final_matrix_exact(res.T16, shift).reshape(16), recs["T"][k])
        assert np.array_equal(np.array(res.VCM), recs["VCM"][k])
        assert int(recs["n_outer"][k]) == res.n_outer and int(recs["n_corr"][k]) == res.n_corr
        if k == 3:
            _assert_loop_parity(res, _oracle_loop(oracle, r1, l1, n1, r2, l2, n2))


def test_pair_5m_points(ctx, oracle):
    """BASELINE configs[4] point count (5 M points per cloud, L = 11.2 m): the cell tables of this size, oracle loop
    parity, sampled brute-force NN at full size."""
    import pwicp_amd as P
    from pwicp_amd import synth
    n = 5000000
    tgt, src, Tgt = _data.pair(n)
    l1, n1 = synth.grid_labels(tgt, 10 * R)
    l2, n2 = synth.grid_labels(src, 10 * R)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
    res = pair.run(check=False)
    io = _oracle_loop(oracle, tgt, l1, n1, src, l2, n2)
    _assert_loop_parity(res, io)
    m1, m2 = pair.num_patches()
    assert m1 > 40000 and m2 > 40000
    idx, d2 = ctx.determineCorrespondences(tgt, src[:4000])
    t64 = tgt.astype(np.float64)
    for i in range(0, 4000, 100):
        d = ((t64 - src[i].astype(np.float64)) ** 2).sum(1)
        assert abs(d.min() - d2[i]) <= 1e-6 * max(d.min(), 1e-12) + 1e-12
    pair.close()


@pytest.mark.parametrize("pair_mode", [0, -1, 3])
def test_the_references_own_run_through_the_exported_entry_point(tmp_path, ctx, pair_mode):
    """src/main.cpp:27-28 of the reference: PiecewiseICP_4D_call(configuration_4d.txt, 0, 20, pairMode, 0.75) on its 20 scans
    (kept as fixtures), here through libpwicp.so's function of the same name - preprocessing, front end, loop, composition and
    result files all by the product.  pairMode 0 / -1 (what main.cpp passes) / 3 are the three runs whose per-pair files the
    reference checked in (<e>_Direct2Ref_ / _Adaptive_ / _Fixed_TransMatrix.txt): every one of the 57 files within its entry of
    tests/golden/tolerance_table.json (the table the oracle is held to), and within north_star's 1e-5 rad / 1e-4 m.
    pairMode -1 additionally: RegPairFile.txt equal to the pair map recovered from the reference's Adaptive results."""
    import pwicp_amd as P
    out = str(tmp_path) + "/"
    cfg = tmp_path / "cfg.txt"
    with open(cfg, "w") as f:      # configuration_files/configuration_4d.txt
        f.write("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                "float PCres1 (m): 0.005\nfloat PCres2 (m): 0.005\nfloat SVsize1 (m): 0.05\nfloat SVsize2 (m): 0.05\n"
                "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): 0.05\nfloat DTmin (m): 0.004\nbool isVisual (yes-1, no-0): 0"
                % (os.path.join(G.GOLD, "inputs"), out))
    # the ground truth of the series where the reference looks for it (R.cpp:209-211: relative to the working directory)
    os.makedirs(tmp_path / "data" / "data_synthetic")
    with open(os.path.join(G.GOLD, "reference_results", "defined_transformations.txt"), "rb") as f:
        (tmp_path / "data" / "data_synthetic" / "defined_transformations.txt").write_bytes(f.read())
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert P.PiecewiseICP_4D_call(str(cfg), 0, 20, pair_mode, 0.75) is True
    finally:
        os.chdir(cwd)
    mode = {0: "Direct2Ref", -1: "Adaptive", 3: "Fixed"}[pair_mode]
    if pair_mode < 0:
        pairs = [tuple(int(v) for v in line.split()[:2]) for line in open(str(tmp_path / "RegPairFile.txt")) if line.strip() and line.split()[0].isdigit()]
        got = {}
        for a, b in pairs:
            got[max(a, b) + 1] = min(a, b) + 1          # the file is 0-based (R.cpp:578-586)
        assert got == AMAP, got
    bad, dist = {}, {}
    for e in range(2, 21):
        T, V, stds = G.parse_transmatrix_file(out + "%d_%s_TransMatrix.txt" % (e, mode))
        Tg, Vg, stds_g = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "%d_%s_TransMatrix.txt" % (e, mode)))
        da, dt = float(np.abs(G.euler(T) - G.euler(Tg)).max()), float(np.abs(T[:3, 3] - Tg[:3, 3]).max())
        # a11: the six printed sigmas and the 6x6 VCM of the SAME file (calTransParaVCM, R.cpp:1273-1343, written R.cpp:520-537)
        ds, dv = float(np.abs(stds / stds_g - 1).max()), float(np.abs(V - Vg).max())
        dist[e] = (da, dt, ds, dv)
        if not (da < TOL[mode][e][0] and dt < TOL[mode][e][1] and da < 1e-5 and dt < 1e-4 and
                ds < STOL[mode][e][0] and dv < STOL[mode][e][1]):
            bad[e] = (da, dt, TOL[mode][e], ds, dv, STOL[mode][e])
    _record_distances("entry_point_%s" % mode, dist)
    assert not bad, bad
    # every printed digit of the VCM and five digits of the sigmas on (nearly) all files, as for the oracle
    assert sum(1 for v in dist.values() if v[2] < 2e-5 and v[3] < 1.5e-12) >= 14      # (the oracle: 14 / 19 / 17)
    assert os.path.exists(out + "TransMatrices_toRef.txt") and os.path.exists(out + "TransParameters_toRef.txt")
    if pair_mode < 0:
        # the reference's checked-in series files are those of main.cpp's own call (pairMode -1): the run's composed outputs
        # (calTransToReferenceEpoch, R.cpp:977-1153) and its error report (calAbsErrorOfTransPara, R.cpp:1157-1251) against them
        from test_distributed_cpu import _read_matrices
        gold = os.path.join(G.GOLD, "reference_results")
        comp = {}
        for name in ("TransMatrices.txt", "TransMatrices_toRef.txt"):
            Tm, Vm = _read_matrices(out + name, 19)
            Tr, Vr = _read_matrices(os.path.join(gold, name), 19)
            comp[name] = (max(float(np.abs(Tm[i].astype(float) - Tr[i]).max()) for i in range(19)),
                          max(float(np.abs(Vm[i] - Vr[i]).max() / np.abs(Vr[i]).max()) for i in range(19)))
        for name in ("TransParameters.txt", "TransParameters_toRef.txt"):
            a, b = np.loadtxt(out + name, skiprows=1), np.loadtxt(os.path.join(gold, name), skiprows=1)
            assert a.shape == b.shape == (19, 13) and np.array_equal(a[:, 0], b[:, 0])
            comp[name] = (float(np.abs(a[:, 1:4] - b[:, 1:4]).max()), float(np.abs(a[:, 4:7] - b[:, 4:7]).max()),
                          float(np.abs(a[:, 7:] / b[:, 7:] - 1).max()))
        a, b = np.loadtxt(out + "TransPara_AbsError.txt", skiprows=1), np.loadtxt(os.path.join(gold, "TransPara_AbsError.txt"), skiprows=1)
        assert a.shape == b.shape == (19, 6)
        comp["TransPara_AbsError.txt"] = (float(np.abs(a[:, :3] - b[:, :3]).max()), float(np.abs(a[:, 3:] - b[:, 3:]).max()))
        _record_distances("entry_point_composed", comp)
        for name in ("TransMatrices.txt", "TransMatrices_toRef.txt"):
            # matrix entries: float print precision of a chain of up to six float products; VCM entries relative to the largest
            assert comp[name][0] < COMPOSED_TOL[name][0] and comp[name][1] < COMPOSED_TOL[name][1], (name, comp[name])
        for name in ("TransParameters.txt", "TransParameters_toRef.txt"):
            assert all(c < t for c, t in zip(comp[name], COMPOSED_TOL[name])), (name, comp[name])      # gon, m, relative sigma
        assert all(c < t for c, t in zip(comp["TransPara_AbsError.txt"], COMPOSED_TOL["TransPara_AbsError.txt"])), comp   # mgon, mm


@pytest.mark.parametrize("shape", ["steep_z", "face_yz", "diagonal"])
def test_loop_parity_on_cliff_like_scenes(ctx, oracle, shape):
    """Steep / volumetric scenes (the rockfall use case): a column of the search grid holds a tall stack of points, the
    per-pair choice between the levels of columns and of cells goes to cells, the far path of the dense search is exercised.
    steep_z: slopes up to ~9 on the tile; face_yz: the tile turned into the y-z plane (a cliff looked at along x: no column
    layout fits); diagonal: the tile tilted 45 degrees about y.  GPU == oracle bit for bit on every discrete quantity."""
    import pwicp_amd as P
    tgt, src, _ = _data.pair(120000)
    out = []
    for a in (tgt, src):
        a = a.copy()
        if shape == "steep_z":
            a[:, 2] += (1.5 * np.sin(6.0 * a[:, 0])).astype(np.float32)
        elif shape == "face_yz":
            a = np.ascontiguousarray(a[:, [2, 0, 1]])
        else:
            c, s_ = np.float32(np.cos(np.pi / 4)), np.float32(np.sin(np.pi / 4))
            x, z = a[:, 0].copy(), a[:, 2].copy()
            a[:, 0] = c * x + s_ * z
            a[:, 2] = -s_ * x + c * z
        out.append(a.astype(np.float32))
    tgt, src = out
    l1, n1 = ctx.frontend_segment(tgt, 10 * R, 45, R)
    l2, n2 = ctx.frontend_segment(src, 10 * R, 45, R)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
    res = pair.run(check=False)
    io = _oracle_loop(oracle, tgt, l1, n1, src, l2, n2)
    _assert_loop_parity(res, io)
    pair.close()


# ---- the reference's SECOND input set: data/data_synthetic/syntheticPC_no_transformations -----------------------------------
# (SELF-CONSISTENCY, not parity: the reference holds no result files for this set - the expected values are this repository's own
#  oracle's, tests/golden/README.md; the parity gate is the 57 files of tests/golden/reference_results)
_NT = json.load(open(os.path.join(G.GOLD, "no_transformations_expected.json")))["pairs"]
NT_INPUTS = os.path.join(G.GOLD, "inputs_no_transformations")
# the reference's own accuracy on the transformed set (largest entries of its TransPara_AbsError.txt: 57.1 mgon, 1.14 mm)
NT_ANG, NT_TR = 58.0 * np.pi / 200000.0, 1.2e-3
# pairs on which the METHOD (oracle == product) does not come back to the identity that closely: its result slides along y
# (e2, e6, e7: 4.6 / 4.7 / 35 mm) or rests on a few hundred stable patches (e11, e16, e19) - recorded, not hidden
NT_DRIFT = {2, 6, 7, 11, 16, 19}


@pytest.mark.parametrize("epoch", list(range(2, 21)))
def test_no_transformations_pairs_through_gpu(ctx, oracle, epoch):
    """The 19 Direct2Ref pairs of the reference's untransformed series (main.cpp:27-28's call on the other folder of
    data/data_synthetic): GPU == oracle on every discrete quantity, GPU == the committed oracle results
    (tests/golden/no_transformations_expected.json), and the identity property: |angles|, |t| within the reference's own
    accuracy level wherever the oracle's are."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    p1 = G.preprocess_4d(oracle, read_pcd(os.path.join(NT_INPUTS, "Epoch_001.pcd")))
    p2 = G.preprocess_4d(oracle, read_pcd(os.path.join(NT_INPUTS, "Epoch_%03d.pcd" % epoch)))
    r1, r2, shift = G.reduce_pair(p1, p2)
    l1, n1 = oracle.ref_frontend(r1, 0.05)
    l2, n2 = oracle.ref_frontend(r2, 0.05)
    pair = P.Pair(ctx, r1, l1, n1, r2, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
    res = pair.run(check=False)
    io = _oracle_loop(oracle, r1, l1, n1, r2, l2, n2, r=0.005, sv=0.05, dtinit=0.05, dtmin=0.004)
    _assert_loop_parity(res, io)
    pair.close()
    exp = _NT[str(epoch)]
    k = res.n_outer
    assert k == exp["n_outer"] and list(res.n_inner[:k]) == exp["n_inner"] and list(res.n_stable[:k]) == exp["n_stable"]
    assert [float(v) for v in res.DTseries[:k + 1]] == exp["DTseries"]
    Tf = G.final_matrix(res.T16, shift)
    Te = np.array(exp["T_final"]).reshape(4, 4)
    assert np.abs(G.euler(Tf) - G.euler(Te)).max() < 1e-5 and np.abs(Tf[:3, 3] - Te[:3, 3]).max() < 1e-4
    if epoch not in NT_DRIFT:
        assert np.abs(G.euler(Tf)).max() < NT_ANG and np.abs(Tf[:3, 3]).max() < NT_TR


def test_no_transformations_series_through_the_exported_entry_point(tmp_path, ctx):
    """PiecewiseICP_4D_call(cfg, 0, 20, 0, 0.75) on the untransformed folder, everything by the product (preprocessing, device
    front end, loop, files): every <e>_Direct2Ref_TransMatrix.txt within 1e-5 rad / 1e-4 m of the oracle's result."""
    import pwicp_amd as P
    out = str(tmp_path) + "/"
    cfg = tmp_path / "cfg.txt"
    with open(cfg, "w") as f:
        f.write("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                "float PCres1 (m): 0.005\nfloat PCres2 (m): 0.005\nfloat SVsize1 (m): 0.05\nfloat SVsize2 (m): 0.05\n"
                "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): 0.05\nfloat DTmin (m): 0.004\nbool isVisual (yes-1, no-0): 0"
                % (NT_INPUTS, out))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert P.PiecewiseICP_4D_call(str(cfg), 0, 20, 0, 0.75) is True
    finally:
        os.chdir(cwd)
    bad, near_identity = {}, 0
    for e in range(2, 21):
        T, _, _ = G.parse_transmatrix_file(out + "%d_Direct2Ref_TransMatrix.txt" % e)
        Te = np.array(_NT[str(e)]["T_final"]).reshape(4, 4)
        da, dt = float(np.abs(G.euler(T) - G.euler(Te)).max()), float(np.abs(T[:3, 3] - Te[:3, 3]).max())
        if not (da < 1e-5 and dt < 1e-4):
            bad[e] = (da, dt)
        near_identity += int(np.abs(G.euler(T)).max() < NT_ANG and np.abs(T[:3, 3]).max() < NT_TR)
    assert not bad, bad
    assert near_identity == 19 - len(NT_DRIFT)


def test_a_second_series_takes_over_the_parked_contexts_of_the_first(tmp_path, ctx):
    """A closed series parks its device contexts and front-end work spaces for the next one of the process
    (pwicp_series_release_parked frees them): the same series run three times - cold, on parked resources, after a release -
    gives byte-identical records."""
    import pwicp_amd as P
    from pwicp_amd import synth
    from pwicp_amd.pcd import write_pcd_binary
    inp = tmp_path / "scans"
    inp.mkdir()
    t, _ = synth.make_tile(60000, R)
    write_pcd_binary(str(inp / "Epoch_001.pcd"), t.astype(np.float32))
    for e in (1, 2):
        s, _ = synth.make_source(60000, R, epoch=e)
        write_pcd_binary(str(inp / ("Epoch_%03d.pcd" % (e + 1))), s.astype(np.float32))
    cfg = tmp_path / "cfg.txt"
    _write_series_config(str(cfg), str(inp), str(tmp_path / "out_"))
    runs = []
    for k in range(3):
        if k == 2:
            P.series_release_parked()
        series = P.Series(str(cfg), 0, 3, 0, 0.75, 0)
        recs = series.run_pairs([0, 1])
        series.close()
        assert np.all(recs["status"] == 0)
        rec = recs.copy()
        rec["t_loop_ms"] = 0
        rec["t_pair_ms"] = 0
        runs.append(rec.tobytes())
    assert runs[0] == runs[1] == runs[2]
    P.series_release_parked()
