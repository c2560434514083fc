"""Parity of the HIP hot path (through the C ABI) against the CPU oracle on identical inputs.
Integer / index results and every float the control flow depends on must be BIT-EXACT; the final transform is
additionally checked against north_star's tolerance (1e-5 rad / 1e-4 m)."""
import ctypes as C
import os

import numpy as np
import pytest

import _data
import _golden as G

pytestmark = pytest.mark.gpu

ANG_TOL, TR_TOL = 1e-5, 1e-4     # BASELINE.json north_star


def test_native_library_is_the_one_running(ctx):
    import pwicp_amd
    assert os.path.exists(pwicp_amd.lib_path())
    maps = open("/proc/self/maps").read()
    assert "libpwicp.so" in maps


@pytest.mark.parametrize("n", [1, 7, 1000, 50000])
def test_nn_bit_exact(ctx, oracle, n):
    rng = np.random.default_rng(n)
    tgt, src, _ = _data.pair(max(n, 400))
    tgt = tgt[:n] if n < 400 else tgt
    q = src[: max(n, 64)].copy()
    q[:10] += np.float32(2.0)                      # far and outside the bounding box
    q[10:20, 0] -= np.float32(9.0)
    idx, d2 = ctx.determineCorrespondences(tgt, q)
    oi, od = oracle.nn1(tgt, q)
    assert np.array_equal(idx, oi)
    assert np.array_equal(d2, od)


def test_nn_ties_duplicates_and_empty_query(ctx, oracle):
    t = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0], [-1, 0, 0], [1, 0, 0]], np.float32)
    q = np.array([[1, 0, 0], [0.5, 0, 0], [0, 0, 0], [100, 100, 100]], np.float32)
    idx, d2 = ctx.determineCorrespondences(t, q)
    oi, od = oracle.nn1(t, q)
    assert list(idx) == list(oi) == [1, 0, 0, 1]
    assert np.array_equal(d2, od)
    idx, d2 = ctx.determineCorrespondences(t, np.zeros((0, 3), np.float32))
    assert len(idx) == 0


def test_nn_large_offset_coordinates(ctx, oracle):
    """Unreduced coordinates (+1e4 m, the rockfall-like case of SURVEY §8d cfg 3): float cell assignment slack."""
    tgt, src, _ = _data.pair(30000, offset=(1.0e4, -2.0e4, 3.0e3), reduce=False)
    idx, d2 = ctx.determineCorrespondences(tgt, src)
    oi, od = oracle.nn1(tgt, src)
    assert np.array_equal(d2, od)
    assert np.array_equal(idx, oi)


def test_percentile_and_overlap(ctx, oracle):
    tgt, src, _ = _data.pair(60000)
    p = ctx.calPercentileDistBetween2PC(tgt, src, 0.75)
    po = oracle.lib().orc_percentile_dist(oracle._p(oracle.f4(tgt)), len(tgt), oracle._p(oracle.f4(src)), len(src), 0.75)
    assert p == po
    r = ctx.calOverlapRatioByC2Cdist(tgt, src, 0.01)
    ro = oracle.lib().orc_overlap_ratio(oracle._p(oracle.f4(tgt)), len(tgt), oracle._p(oracle.f4(src)), len(src), 0.01)
    assert r == ro


def _labels(cloud, which):
    from pwicp_amd import synth
    import _oracle as O
    if which == "ref" and O.ref_frontend_available():
        return O.ref_frontend(cloud, 10 * _data.R)
    return synth.grid_labels(cloud, 10 * _data.R)


@pytest.mark.parametrize("n,which", [(20000, "ref"), (100000, "grid")])
def test_select_patches_and_normals_bit_exact(ctx, oracle, n, which):
    tgt, src, _ = _data.pair(n)
    lab, nsv = _labels(src, which)
    Po = oracle.select_patches(src, lab, nsv)
    Pg = ctx.selectPatches(src, lab, nsv)
    assert np.array_equal(Po.off, Pg["off"])
    for k in ("pat", "src", "ct", "bp", "bpstd", "ctstd"):
        assert np.array_equal(getattr(Po, k), Pg[k]), k
    nrm, ok = ctx.patchNormals(Po.pat, Po.off)
    for i in range(0, Po.m, max(1, Po.m // 200)):
        seg = np.ascontiguousarray(Po.pat[Po.off[i]:Po.off[i + 1]])
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        o = oracle.lib().orc_cal_patch_normal(oracle._p(seg), len(seg), C.byref(a), C.byref(b), C.byref(c))
        assert o == ok[i]
        assert (a.value, b.value, c.value) == tuple(nrm[i, :3])


def test_select_patches_edge_cases(ctx, oracle):
    tgt, _, _ = _data.pair(5000)
    # everything in one label; tiny labels; labels with no points
    for lab, nsv in [(np.zeros(len(tgt), np.int32), 1), (np.arange(len(tgt), dtype=np.int32) % 997, 997),
                     ((np.arange(len(tgt), dtype=np.int32) % 3) * 5, 20)]:
        Po = oracle.select_patches(tgt, lab, nsv)
        Pg = ctx.selectPatches(tgt, lab, nsv)
        assert np.array_equal(Po.off, Pg["off"]) and np.array_equal(Po.pat, Pg["pat"])
    import pwicp_amd
    with pytest.raises(pwicp_amd.PwicpError):
        ctx.selectPatches(tgt, np.full(len(tgt), 7, np.int32), 3)      # label out of range -> error, not UB


def test_inner_icp_and_vcm(ctx, oracle):
    tgt, src, _ = _data.pair(40000)
    lab1, n1 = _labels(tgt, "grid")
    lab2, n2 = _labels(src, "grid")
    P1 = oracle.select_patches(tgt, lab1, n1)
    P2 = oracle.select_patches(src, lab2, n2)
    n1v, _ = ctx.patchNormals(P1.pat, P1.off)
    n2v, _ = ctx.patchNormals(P2.pat, P2.off)
    Tg, it = ctx.P2PICPwithPatchNormal(P1.ct, n1v, P2.ct, n2v, 1e-6)
    To = np.zeros(16, np.float32)
    ito = oracle.lib().orc_p2p_icp(oracle._p(oracle.f4(P1.ct)), oracle._p(oracle.f4(n1v)), P1.m, oracle._p(oracle.f4(P2.ct)),
                                   oracle._p(oracle.f4(n2v)), P2.m, 1e-6, oracle._p(To), None)
    assert it == ito
    assert np.abs(_data.euler(Tg) - _data.euler(To)).max() < ANG_TOL
    assert np.abs(Tg[:3, 3] - To.reshape(4, 4)[:3, 3]).max() < TR_TOL
    assert np.abs(Tg.reshape(16) - To).max() <= 2 ** -23           # at most one float ulp of ~1
    Vg = ctx.calTransParaVCM(P1.ct, n1v, P2.ct)
    Vo = np.zeros(36)
    oracle.lib().orc_cal_trans_para_vcm(oracle._p(oracle.f4(P1.ct)), oracle._p(oracle.f4(n1v)), P1.m,
                                        oracle._p(oracle.f4(P2.ct)), P2.m, oracle._p(Vo, oracle.dp))
    assert np.allclose(Vg.reshape(36), Vo, rtol=1e-9, atol=1e-18)


def test_vcm_when_the_fit_explains_nearly_everything(ctx, oracle):
    """calTransParaVCM forms v^T v as L^T L - x^T A^T L unless the two terms cancel (icp.hip: vcm_block); then - this case, a
    noise-free rigid motion: v is only the model's second-order error - it forms the residuals point by point as the reference
    does (R.cpp:1331-1333).  Both branches have to agree with the oracle's explicit form."""
    tgt, _, _ = _data.pair(40000)
    lab1, n1 = _labels(tgt, "grid")
    P1 = oracle.select_patches(tgt, lab1, n1)
    n1v, _ = ctx.patchNormals(P1.pat, P1.off)
    th = 1.0e-3
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    ct = np.asarray(P1.ct, np.float64)[:, :3]
    moved = (ct @ R.T + np.array([2e-4, -1e-4, 3e-4])).astype(np.float32)
    Vg = ctx.calTransParaVCM(P1.ct, n1v, moved)
    Vo = np.zeros(36)
    oracle.lib().orc_cal_trans_para_vcm(oracle._p(oracle.f4(P1.ct)), oracle._p(oracle.f4(n1v)), P1.m,
                                        oracle._p(oracle.f4(moved)), len(moved), oracle._p(Vo, oracle.dp))
    assert np.isfinite(Vg).all() and (np.diag(Vg.reshape(6, 6)) > 0).all()
    assert np.allclose(Vg.reshape(36), Vo, rtol=1e-8, atol=1e-22)


def test_vcm_on_an_ill_conditioned_patch_layout(ctx, oracle):
    """v^T v from the sums must not lose digits when A^T A is ill-conditioned.  Target centroids on a strip 5 m long
    and 2 cm wide with nearly parallel normals (the rotation about the strip's axis and two translations are barely
    constrained: cond(A^T A) ~ 1e9 and more), sources = targets + noise (v^T v / L^T L ~ 1: the branch that uses the sums) and
    sources = targets + a tiny rigid motion (ratio ~ 0: the point-by-point branch).  Both against the oracle's explicit residuals
    (R.cpp:1331-1333) at the tolerance of the well-conditioned test."""
    rng = np.random.default_rng(7)
    m = 3000
    ct = np.zeros((m, 4), np.float32)
    ct[:, 0] = rng.uniform(0, 5, m)
    ct[:, 1] = rng.uniform(0, 0.02, m)
    ct[:, 2] = 0.05 * np.sin(ct[:, 0])
    ct[:, 3] = 1.0
    nrm = np.zeros((m, 4), np.float32)
    nrm[:, 0] = -0.05 * np.cos(ct[:, 0]) + rng.normal(0, 1e-3, m)
    nrm[:, 1] = rng.normal(0, 1e-3, m)
    nrm[:, 2] = 1.0
    nrm[:, :3] /= np.linalg.norm(nrm[:, :3], axis=1)[:, None]
    for moved in ((ct[:, :3] + rng.normal(0, 2e-3, (m, 3))).astype(np.float32),
                  (ct[:, :3] + np.array([1e-5, 0, 2e-5])).astype(np.float32)):
        Vg = ctx.calTransParaVCM(ct, nrm, moved)
        Vo = np.zeros(36)
        oracle.lib().orc_cal_trans_para_vcm(oracle._p(oracle.f4(ct)), oracle._p(oracle.f4(nrm)), m,
                                            oracle._p(oracle.f4(moved)), len(moved), oracle._p(Vo, oracle.dp))
        assert np.isfinite(Vg).all() and np.isfinite(Vo).all()
        # sigma0^2 scales the whole matrix: compare it through the diagonal, to 1e-8 relative whatever the conditioning ...
        d_g, d_o = np.diag(Vg.reshape(6, 6)), np.diag(Vo.reshape(6, 6))
        assert np.allclose(d_g, d_o, rtol=1e-8, atol=0)
        # ... and the ratio of any two entries is Qxx's, which both sides invert with the same LU
        assert np.allclose(Vg.reshape(36) / d_g[0], Vo / d_o[0], rtol=1e-7, atol=1e-12 * np.abs(Vo / d_o[0]).max())


def test_pairs_of_one_target_on_two_contexts(ctx):
    """pwicp_pair_create_with_target_on: the pairs of a shared target may live on other contexts (= streams) of its device; a pair
    created and run on a second context while another one runs on the first gives bit for bit what it gives alone."""
    import threading
    import pwicp_amd as P
    tgt, s1, _ = _data.pair(60000, epoch=1)
    _, s2, _ = _data.pair(60000, epoch=2)
    prm = _data.params()
    lt, nt = _labels(tgt, "grid")
    l1, n1 = _labels(s1, "grid")
    l2, n2 = _labels(s2, "grid")
    T = P.Target(ctx, tgt, lt, nt, prm.Res1, prm.SVRes1)
    alone = []
    for s, l, n in ((s1, l1, n1), (s2, l2, n2)):
        pr = P.Pair(ctx, None, None, 0, s, l, n, prm, target=T)
        alone.append(pr.run())
        pr.close()
    ctx2 = P.Context(0)
    out = [None, None]

    def work(i, c, s, l, n):
        for _ in range(5):
            pr = P.Pair(c, None, None, 0, s, l, n, prm, target=T)
            out[i] = pr.run()
            pr.close()

    th = [threading.Thread(target=work, args=(0, ctx, s1, l1, n1)), threading.Thread(target=work, args=(1, ctx2, s2, l2, n2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for a, b in zip(alone, out):
        assert b is not None and b.status == 0
        assert list(a.T16) == list(b.T16) and list(a.VCM) == list(b.VCM) and a.n_outer == b.n_outer
        assert list(a.DTseries[:a.n_outer + 1]) == list(b.DTseries[:b.n_outer + 1])
    T.close()
    ctx2.close()


def test_four_registrations_side_by_side(ctx):
    """Independent pairs of a series on four contexts (= streams, mailboxes, pools) of one GPU, re-registered from four host threads
    at once (bench.py: pairs_side_by_side, tools/loop_concurrency.py): every repetition of every pair gives bit for bit what the
    pair gives alone - T, VCM, the DT series, the per-iteration counts."""
    import threading
    import pwicp_amd as P
    prm = _data.params()
    K, REP = 4, 25
    tgt = _data.pair(200000, epoch=1)[0]
    lt, nt = _labels(tgt, "grid")
    T = P.Target(ctx, tgt, lt, nt, prm.Res1, prm.SVRes1)
    ctxs = [ctx] + [P.Context(0) for _ in range(K - 1)]
    pairs, alone = [], []
    for e in range(K):
        s = _data.pair(200000, epoch=e + 1)[1]
        l, n = _labels(s, "grid")
        pairs.append(P.Pair(ctxs[e], None, None, 0, s, l, n, prm, target=T))
    for pr in pairs:
        alone.append(pr.run())

    def key(r):
        no = r.n_outer
        return (r.status, no, list(r.T16), list(r.VCM), list(r.DTseries[:no + 1]), list(r.n_inner[:no]), list(r.n_stable[:no]))
    bad = []

    def work(i):
        want = key(alone[i])
        for rep in range(REP):
            pairs[i].reset()
            if key(pairs[i].run()) != want:
                bad.append((i, rep))
    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad
    assert all(a.status == 0 for a in alone)
    # the same through the library's own call (pwicp_pairs_run_concurrent: one host thread per pair inside the library)
    for rep in range(5):
        got = P.run_pairs_concurrent(pairs)
        assert [key(g) for g in got] == [key(a) for a in alone], rep
    # pairs that share a context run one after the other on that context's thread, beside the other contexts' pairs: same results
    s2 = _data.pair(200000, epoch=2)[1]
    extra = P.Pair(ctxs[0], None, None, 0, s2, *_labels(s2, "grid"), prm, target=T)
    try:
        got = P.run_pairs_concurrent(pairs + [extra])
        assert [key(g) for g in got[:K]] == [key(a) for a in alone] and key(got[K]) == key(alone[1])
        with pytest.raises(P.PwicpError):          # the same pair twice: refused
            P.run_pairs_concurrent([pairs[0], pairs[0]])
    finally:
        extra.close()
    for pr in pairs:
        pr.close()
    T.close()
    for c in ctxs[1:]:
        c.close()


def _loop_both(ctx, oracle, tgt, src, l1, n1, l2, n2, manual=True):
    import pwicp_amd as P
    R = _data.R
    prm = _data.params(manual)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, prm)
    res = pair.run(check=False)
    P1 = oracle.select_patches(tgt, l1, n1)
    P2 = oracle.select_patches(src, l2, n2)
    io = oracle.run_loop(tgt, src, P1, P2, R, R, 10 * R, 10 * R, 10 * R, 0.8 * R, manual_dt=manual)
    return pair, res, io


def _assert_loop_parity(res, io):
    assert res.status == 0 and io.status == 0
    assert res.n_outer == io.n_outer
    k = io.n_outer
    assert list(res.n_inner[:k]) == list(io.n_inner[:k])
    assert list(res.n_stable[:k]) == list(io.n_stable[:k])
    assert list(res.n_stable_pts[:k]) == list(io.n_stable_pts[:k])
    assert [float(x) for x in res.DTseries[:k + 1]] == [float(x) for x in io.DTseries[:k + 1]]      # bit-exact floats
    assert [res.d75[i] for i in range(k)] == [io.d75[i] for i in range(k)]
    assert [float(res.LoDmin[i]) for i in range(k)] == [float(io.LoDmin[i]) for i in range(k)]
    assert [float(res.maxBB[i]) for i in range(k)] == [float(io.maxBB[i]) for i in range(k)]
    assert res.n_corr == io.n_corr
    Tg = np.array(res.T16, np.float64).reshape(4, 4)
    To = np.array(io.T16, np.float64).reshape(4, 4)
    assert np.abs(_data.euler(Tg) - _data.euler(To)).max() < ANG_TOL
    assert np.abs(Tg[:3, 3] - To[:3, 3]).max() < TR_TOL
    assert np.allclose(np.array(res.VCM), np.array(io.VCM), rtol=1e-6, atol=1e-20)


@pytest.mark.parametrize("n,which,epoch", [(20000, "ref", 1), (60000, "ref", 2), (200000, "grid", 3)])
def test_loop_parity_with_oracle(ctx, oracle, n, which, epoch):
    tgt, src, Tgt = _data.pair(n, epoch=epoch)
    l1, n1 = _labels(tgt, which)
    l2, n2 = _labels(src, which)
    pair, res, io = _loop_both(ctx, oracle, tgt, src, l1, n1, l2, n2)
    _assert_loop_parity(res, io)
    # reset + rerun is deterministic (no atomics on floats anywhere in the path)
    pair.reset()
    res2 = pair.run()
    assert np.array_equal(np.array(res2.T16), np.array(res.T16)) and np.array_equal(np.array(res2.VCM), np.array(res.VCM))
    # the source cloud on the device was transformed exactly like the oracle's (R.cpp:943-945)
    moved = pair.download_source()
    c2 = oracle.f4(src).copy()
    for i in range(io.n_outer):
        Tk = np.array(io.Tk[i], np.float32)
        oracle.lib().orc_transform_points(oracle._p(c2), len(c2), oracle._p(Tk))
    assert np.array_equal(moved[:, :3], c2[:, :3])
    pair.close()


def test_loop_rockfall_scale_unreduced_coordinates(ctx, oracle):
    """Stand-in for BASELINE configs[2] (rockfall data is not in the reference tree): spacing 0.3 m, patches 3 m,
    coordinates offset by ~1e4 m and NOT reduced, so the float rounding slack of the cell assignment matters."""
    import pwicp_amd as P
    from pwicp_amd import synth
    r = 0.3
    tgt, _ = synth.make_tile(40000, r, offset=(1.0e4, -2.0e4, 1.5e3))
    src, _ = synth.make_source(40000, r, epoch=4, offset=(1.0e4, -2.0e4, 1.5e3))
    l1, n1 = synth.grid_labels(tgt, 10 * r)
    l2, n2 = synth.grid_labels(src, 10 * r)
    prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, prm)
    res = pair.run(check=False)
    io = oracle.run_loop(tgt, src, oracle.select_patches(tgt, l1, n1), oracle.select_patches(src, l2, n2), r, r, 10 * r, 10 * r,
                         10 * r, 0.8 * r)
    _assert_loop_parity(res, io)
    pair.close()


def test_loop_auto_dtinit(ctx, oracle):
    tgt, src, _ = _data.pair(30000)
    l1, n1 = _labels(tgt, "grid")
    l2, n2 = _labels(src, "grid")
    pair, res, io = _loop_both(ctx, oracle, tgt, src, l1, n1, l2, n2, manual=False)
    _assert_loop_parity(res, io)
    pair.close()


def test_too_few_patches_is_an_error_code_not_exit(ctx):
    import pwicp_amd as P
    tgt, src, _ = _data.pair(3000)
    lab = np.zeros(len(src), np.int32)
    pair = P.Pair(ctx, tgt, np.zeros(len(tgt), np.int32), 1, src, lab, 1, _data.params())
    res = pair.run(check=False)
    assert res.status == -3          # PWICP_E_TOO_FEW_PATCHES (reference: std::exit, R.cpp:728-731)
    pair.close()


def test_full_size_properties_1m(ctx):
    """BASELINE configs[1] size (1 M points per cloud): size-independent properties instead of the oracle."""
    import pwicp_amd as P
    from pwicp_amd import synth
    tgt, src, Tgt = _data.pair(1000000)
    l1, n1 = synth.grid_labels(tgt, 10 * _data.R)
    l2, n2 = synth.grid_labels(src, 10 * _data.R)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
    res = pair.run()
    k = res.n_outer
    dts = np.array(res.DTseries[:k + 1])
    assert res.status == 0 and 2 <= k <= 20
    assert np.all(np.diff(dts) <= 0) and dts[-1] >= 0.8 * _data.R * (1 - 1e-6)         # monotone schedule, floor DTmin
    # ground truth of the generator (source -> target, both reduced by the target centroid)
    c = synth.make_tile(1000000, _data.R)[0].mean(0).astype(np.float64)
    S = np.eye(4); S[:3, 3] = -c
    Tfull = np.linalg.inv(S) @ np.array(res.T16, np.float64).reshape(4, 4) @ S
    assert np.abs(_data.euler(Tfull) - _data.euler(Tgt)).max() < 1e-4
    assert np.abs(Tfull[:3, 3] - Tgt[:3, 3]).max() < 1e-3
    # product of the per-iteration matrices equals the accumulated matrix (float Eigen order)
    acc = np.eye(4, dtype=np.float32)
    for i in range(k):
        Tk = np.array(res.Tk[i], np.float32).reshape(4, 4)
        new = np.zeros((4, 4), np.float32)
        for a in range(4):
            for b in range(4):
                s = np.float32(Tk[a, 0] * acc[0, b])
                for kk in range(1, 4):
                    s = np.float32(s + np.float32(Tk[a, kk] * acc[kk, b]))
                new[a, b] = s
        acc = new
    assert np.array_equal(acc.reshape(16), np.array(res.T16, np.float32))
    # idempotence: registering the registered source again moves it by < the detection level
    moved = pair.download_source()
    pair2 = P.Pair(ctx, tgt, l1, n1, moved[:, :3], l2, n2, _data.params())
    res2 = pair2.run()
    T2 = np.array(res2.T16, np.float64).reshape(4, 4)
    assert np.abs(_data.euler(T2)).max() < 5e-5 and np.abs(T2[:3, 3]).max() < 5e-4
    # NN: sampled brute-force check at full size
    idx, d2 = ctx.determineCorrespondences(tgt, src[:2000])
    for i in range(0, 2000, 40):
        d = ((tgt - src[i]) ** 2).sum(1)
        assert abs(d.min() - d2[i]) <= 1e-6 * max(d.min(), 1e-12) + 1e-12
    pair.close(); pair2.close()


def test_golden_epoch2_through_gpu(ctx, oracle):
    """The reference's own result for Epoch_002 -> Epoch_001, hot path on the GPU."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    p1 = G.preprocess_4d(oracle, read_pcd(G.epoch_path(1)))
    p2 = G.preprocess_4d(oracle, read_pcd(G.epoch_path(2)))
    r1, r2, shift = G.reduce_pair(p1, p2)
    l1, n1 = oracle.ref_frontend(r1, 0.05)
    l2, n2 = oracle.ref_frontend(r2, 0.05)
    prm = P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004)
    pair = P.Pair(ctx, r1, l1, n1, r2, l2, n2, prm)
    res = pair.run()
    assert pair.num_patches() == (1822, 1846)
    Tf = G.final_matrix(res.T16, shift)
    Tg, Vg, stds = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "2_Direct2Ref_TransMatrix.txt"))
    assert np.abs(G.euler(Tf) - G.euler(Tg)).max() < 5e-6
    assert np.abs(Tf[:3, 3].astype(float) - Tg[:3, 3]).max() < 5e-6
    V = np.array(res.VCM).reshape(6, 6)
    mine = np.concatenate([1000 * 63.6619772368 * np.sqrt(np.diag(V)[:3]), 1000 * np.sqrt(np.diag(V)[3:])])
    assert np.allclose(mine, stds, rtol=5e-3)
    # honest accounting (round 6): far queries of a dense search that were cut short at the percentile's edge are counted, never more
    # than the dense queries there were, and the reference-defined query count is untouched by the count
    assert 0 <= res.n_dense_bounded <= res.n_corr_dense <= res.n_corr
    pair.close()


@pytest.mark.gpu
def test_loop_is_deterministic_run_to_run(ctx):
    """The same resident pair, run 80 times: one single result, bit for bit (T, VCM, thresholds, counts).  Guards the
    'last block finishes the job' kernels (bounding-box fold, percentile pick, ICP / VCM solve), whose completion counts
    rely on the partial results having been PERFORMED, not merely issued."""
    import pwicp_amd as P
    from pwicp_amd.pcd import read_pcd
    g = os.path.join(os.path.dirname(__file__), "golden", "inputs")
    p1 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_001.pcd")), 0.005, 14, 2.7)
    p2 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_002.pcd")), 0.005, 14, 2.7)
    cen = p1[:, :3].mean(0)
    p1[:, :3] -= cen
    p2[:, :3] -= cen
    l1, n1 = ctx.frontend_segment(p1, 0.05, 45, 0.005)
    l2, n2 = ctx.frontend_segment(p2, 0.05, 45, 0.005)
    pair = P.Pair(ctx, p1, l1, n1, p2, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
    seen = set()
    for _ in range(80):
        pair.reset()
        r = pair.run()
        seen.add((bytes(np.array(r.T16, np.float32)), bytes(np.array(r.VCM, np.float64)), tuple(r.n_inner[:r.n_outer]),
                  tuple(r.n_stable[:r.n_outer]), bytes(np.array(r.DTseries[:r.n_outer + 1], np.float32)),
                  bytes(np.array(r.maxBB[:r.n_outer], np.float32)), bytes(np.array(r.d75[:r.n_outer], np.float64))))
    pair.close()
    assert len(seen) == 1


SCHEDULE_WORKER = r'''
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import pwicp_amd as P
import _data
ctx = P.Context(0)
h = hashlib.sha256()
for (n, ep, manual) in ((60000, 1, True), (60000, 3, False), (200000, 2, True)):
    tgt, src, _ = _data.pair(n, epoch=ep)
    l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
    l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
    for rep in range(2):                                  # the second pair of a size reuses the first one's pooled buffers
        pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params(manual))
        r = pair.run()
        no = r.n_outer
        for a in (np.array(r.T16, np.float32), np.array(r.VCM, np.float64), np.array(r.n_inner[:no]), np.array(r.n_stable[:no]),
                  np.array(r.DTseries[:no + 1], np.float32), np.array(r.maxBB[:no], np.float32), np.array(r.d75[:no], np.float64),
                  pair.download_source()):
            h.update(np.ascontiguousarray(a).tobytes())
        pair.close()
print("FINGERPRINT", h.hexdigest())
'''


@pytest.mark.gpu
def test_scheduling_switches_do_not_change_a_bit(tmp_path):
    """Stage guard (the Stage-1 -> Stage-2 decision on the device), speculative first dense search, fused percentile selection,
    the closing stream synchronisation, the allocation pool, the form of the dense search's far path, the stream the first dense search
    runs on, the memory it takes its candidates from and the lanes per front
    query only change WHEN things are launched, by how many lanes, and where buffers come from: T, VCM, every per-iteration series and the moved source cloud are bit-identical with each of them switched off.
    So they are without the per-iteration SOURCE patch normals (R.cpp:823-824): PCL's point-to-plane estimate reads the target's
    normals only, the source's are dead values in the reference (loop.hip: source_normals())."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(SCHEDULE_WORKER % {"root": root})
    prints = {}
    for name, extra in (("default", {}), ("guard off", {"PWICP_STAGE_GUARD": "0"}), ("no speculation", {"PWICP_SPECULATE_DENSE": "0"}),
                        ("selection on its own launches", {"PWICP_FUSED_SELECT": "0"}),
                        ("no pool, closing synchronisation", {"PWICP_POOL_MB": "0", "PWICP_RUN_SYNC": "1"}),
                        ("far queries of every dense search on their own launch", {"PWICP_DENSE_FAR_GROUP": "1"}),
                        ("... every one of them searched to the end, whatever the selection's bins say already",
                         {"PWICP_DENSE_FAR_GROUP": "1", "PWICP_DENSE_FAR_EDGE": "0"}),
                        ("far queries inside the search's blocks, 8 lanes per front query", {"PWICP_DENSE_FAR_GROUP": "0", "PWICP_FRONT_QUERY_LANES": "8"}),
                        ("4 lanes per front query", {"PWICP_FRONT_QUERY_LANES": "4"}),
                        ("source patch normals left out", {"PWICP_SOURCE_NORMALS": "0"}),
                        ("dense search gathers its queries through the order array", {"PWICP_DENSE_QUERY_COPY": "0"})):
        env = dict(os.environ)
        for k in ("PWICP_STAGE_GUARD", "PWICP_SPECULATE_DENSE", "PWICP_FUSED_SELECT", "PWICP_POOL_MB", "PWICP_RUN_SYNC",
                  "PWICP_DENSE_FAR_GROUP", "PWICP_DENSE_FAR_EDGE", "PWICP_FRONT_QUERY_LANES", "PWICP_SOURCE_NORMALS", "PWICP_DENSE_QUERY_COPY"):
            env.pop(k, None)
        env.update(extra)
        out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0 and "FINGERPRINT" in out.stdout, name + ": " + out.stdout[-2000:] + out.stderr[-2000:]
        prints[name] = out.stdout.split("FINGERPRINT")[1].split()[0]
    assert len(set(prints.values())) == 1, prints


LAYOUT_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import pwicp_amd as P
import _oracle, _data
ctx = P.Context(0)
tgt, src, _ = _data.pair(40000)
rng = np.random.default_rng(5)
far = (src[:2000] + rng.normal(0, 0.5, (2000, 3))).astype(np.float32)          # queries far from the surface too
q = np.vstack([src, far]).astype(np.float32)
idx, d2 = ctx.determineCorrespondences(tgt, q)
oi, od = _oracle.nn1(tgt, q)
assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
nb = ctx.knn(src, 20)
from scipy.spatial import cKDTree
d, ii = cKDTree(src.astype(np.float64)).query(src.astype(np.float64), k=20)
dn = np.linalg.norm(src[nb].astype(np.float64) - src[:, None, :].astype(np.float64), axis=2)
assert np.allclose(dn, d, rtol=0, atol=1e-12)
print("LAYOUT_OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["0", "1", "2"])
def test_every_grid_layout_gives_the_exact_minimum(tmp_path, layout):
    """Cells, y-columns and z-columns (PWICP_GRID_LAYOUT, read once per process) return the oracle's (index, d2) for
    every query and the exact k-NN lists — the layout is a cost choice only."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(LAYOUT_WORKER % {"root": root})
    env = dict(os.environ)
    env["PWICP_GRID_LAYOUT"] = layout
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "LAYOUT_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_too_few_stable_patches_is_an_error_code_and_the_pair_stays_usable(ctx, oracle):
    """Source shifted far beyond the distance threshold: no stable patch (reference: std::exit, R.cpp:864-867) ->
    PWICP_E_TOO_FEW_STABLE, no hang; the same context then registers a proper pair bit-identically to a fresh one."""
    import pwicp_amd as P
    tgt, src, _ = _data.pair(30000)
    l1, n1 = _labels(tgt, "grid")
    l2, n2 = _labels(src, "grid")
    far = src.copy()
    far[:, 2] += np.float32(1.0)                       # 1 m off, DTinit = 5 cm
    bad = P.Pair(ctx, tgt, l1, n1, far, l2, n2, _data.params())
    res = bad.run(check=False)
    assert res.status == -4                            # PWICP_E_TOO_FEW_STABLE
    bad.close()
    good = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
    r1 = good.run()
    io = oracle.run_loop(tgt, src, oracle.select_patches(tgt, l1, n1), oracle.select_patches(src, l2, n2),
                         _data.R, _data.R, 10 * _data.R, 10 * _data.R, 10 * _data.R, 0.8 * _data.R)
    assert r1.status == 0 and r1.n_outer == io.n_outer
    assert np.abs(np.array(r1.T16, np.float64) - np.array(io.T16, np.float64)).max() < 1e-6
    good.close()


@pytest.mark.gpu
def test_non_finite_input_is_rejected(ctx):
    import pwicp_amd as P
    tgt, src, _ = _data.pair(20000)
    l1, n1 = _labels(tgt, "grid")
    l2, n2 = _labels(src, "grid")
    broken = tgt.copy()
    broken[123, 1] = np.nan
    with pytest.raises(P.PwicpError):
        P.Pair(ctx, broken, l1, n1, src, l2, n2, _data.params())
    with pytest.raises(P.PwicpError):
        ctx.determineCorrespondences(broken, src)


@pytest.mark.gpu
def test_shared_target_gives_the_same_results(ctx):
    """Two source epochs registered against ONE device-side target (pwicp_target) == each registered with its own
    pwicp_pair_create, bit for bit; the target is not modified by a run."""
    import pwicp_amd as P
    tgt, src1, _ = _data.pair(40000, epoch=1)
    _, src2, _ = _data.pair(40000, epoch=2)
    l1, n1 = _labels(tgt, "grid")
    prm = _data.params()
    T = P.Target(ctx, tgt, l1, n1, prm.Res1, prm.SVRes1)
    outs = []
    for src in (src1, src2, src1):
        l2, n2 = _labels(src, "grid")
        a = P.Pair(ctx, None, None, 0, src, l2, n2, prm, target=T)
        ra = a.run()
        b = P.Pair(ctx, tgt, l1, n1, src, l2, n2, prm)
        rb = b.run()
        assert ra.status == 0 and list(ra.T16) == list(rb.T16) and list(ra.VCM) == list(rb.VCM)
        assert list(ra.DTseries[:ra.n_outer + 1]) == list(rb.DTseries[:rb.n_outer + 1])
        outs.append(list(ra.T16))
        a.close()
        b.close()
    assert outs[0] == outs[2] and outs[0] != outs[1]
    T.close()


@pytest.mark.gpu
@pytest.mark.parametrize("manual", [True, False])
def test_single_iteration_steps_equal_the_whole_loop(ctx, oracle, manual):
    """pwicp_pair_step = PwICP_singleIteration (R.h:181-188): the caller keeps currDT / BBchange_1,2 / the stage flags and
    runs the loop of Piecewise_ICP (R.cpp:680-694) itself.  Stepping a reset pair to Stage 3 must give, bit for bit, what
    pwicp_pair_run gives (and therefore the oracle's DT series and counts)."""
    import pwicp_amd as P
    tgt, src, _ = _data.pair(60000, epoch=2)
    l1, n1 = _labels(tgt, "grid")
    l2, n2 = _labels(src, "grid")
    pair, res, io = _loop_both(ctx, oracle, tgt, src, l1, n1, l2, n2, manual=manual)
    _assert_loop_parity(res, io)
    pair.reset()
    st = P.Step()
    st.currDT = res.DTseries[0] if manual else pair.auto_dtinit()
    assert float(st.currDT) == float(res.DTseries[0])
    T = np.eye(4, dtype=np.float32)
    k = 0
    while not st.toStage3:
        assert pair.step(st) == 0 and st.status == 0
        assert list(st.T16) == list(res.Tk[k])
        assert (st.n_stable, st.n_stable_pts, st.n_inner) == (res.n_stable[k], res.n_stable_pts[k], res.n_inner[k])
        assert float(st.LoDmin) == float(res.LoDmin[k]) and float(st.maxBB) == float(res.maxBB[k]) and st.d75 == res.d75[k]
        assert float(st.currDT) == float(res.DTseries[k + 1])
        Tk = np.array(st.T16, np.float32).reshape(4, 4)
        new = np.zeros((4, 4), np.float32)
        for a in range(4):
            for b in range(4):
                acc = np.float32(Tk[a, 0] * T[0, b])
                for kk in range(1, 4):
                    acc = np.float32(acc + np.float32(Tk[a, kk] * T[kk, b]))
                new[a, b] = acc
        T = new
        k += 1
        assert k <= res.n_outer
    assert k == res.n_outer
    assert np.array_equal(T.reshape(16), np.array(res.T16, np.float32))
    assert list(st.VCM) == list(res.VCM)
    pair.close()


@pytest.mark.parametrize("far_group", [0, 1])
@pytest.mark.parametrize("scene", ["aligned", "offset", "beyond_coverage"])
def test_dense_search_against_brute_force(ctx, oracle, scene, far_group):
    """The dense 1-NN search of calPercentileDistBetween2PC (C.cpp:266-281) by itself (pwicp_pair_dense_distances: every
    source patch point, both forms of its far path) against the oracle's exhaustive search: the float d2 of every query,
    bit for bit.  `offset`: the clouds 3 cm apart (every ball wide, most queries far: the levels of larger cells);
    `beyond_coverage`: a third of the source lies 0.2 - 0.6 m outside the target's extent (nothing within 2.75 coarse cells:
    the general search, taken by whole blocks - more such queries per block than a block keeps, too)."""
    import pwicp_amd as P
    tgt, src, _ = _data.pair(60000, epoch=2)
    src = src.copy()
    if scene == "offset":
        src[:, 0] += np.float32(0.03); src[:, 2] += np.float32(0.012)
    elif scene == "beyond_coverage":
        far = src[:, 0] > np.quantile(src[:, 0], 0.67)
        src[far, 0] += np.float32(0.2) + (src[far, 0] - src[far, 0].min()) * np.float32(1.5)
    l1, n1 = _labels(tgt, "grid")
    l2, n2 = _labels(src, "grid")
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
    q = pair.source_patch_points()
    d2 = pair.dense_distances(far_group)
    _, ref = oracle.nn1(tgt, q[:, :3])
    assert len(q) > 10000
    assert d2.tobytes() == np.asarray(ref, np.float32).tobytes()
    if scene == "beyond_coverage":
        assert (np.sqrt(d2) > 0.1).sum() > 2000
    pair.close()
