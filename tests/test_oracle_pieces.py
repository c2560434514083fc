"""The oracle's pieces against independent numpy/scipy computations (the reference holds no golden
vectors at this level)."""
import ctypes as C

import numpy as np
import pytest
from scipy.spatial import cKDTree

import _data


def test_nn_matches_ckdtree(oracle):
    rng = np.random.default_rng(1)
    t = rng.normal(size=(5000, 3)).astype(np.float32)
    q = rng.normal(size=(3000, 3)).astype(np.float32) * 1.5
    idx, d2 = oracle.nn1(t, q)
    dd, ii = cKDTree(t.astype(np.float64)).query(q.astype(np.float64))
    # float d2 as FLANN computes it
    diff = q - t[idx]
    ref = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
    assert np.array_equal(d2, ref.astype(np.float32))
    # the oracle's neighbour is the true nearest up to float ties
    dtrue = np.linalg.norm(q.astype(np.float64) - t[ii].astype(np.float64), axis=1)
    dmine = np.linalg.norm(q.astype(np.float64) - t[idx].astype(np.float64), axis=1)
    assert np.all(dmine <= dtrue * (1 + 1e-6) + 1e-12)
    assert (idx != ii).mean() < 1e-3


def test_nn_ties_lowest_index_and_edge_cases(oracle):
    t = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0], [-1, 0, 0]], np.float32)
    q = np.array([[1, 0, 0], [0.5, 0, 0], [0, 0, 0]], np.float32)
    idx, d2 = oracle.nn1(t, q)
    assert list(idx) == [1, 0, 0]
    assert list(d2) == [0.0, 0.25, 0.0]
    idx, d2 = oracle.nn1(t[:1], q)
    assert list(idx) == [0, 0, 0]


def test_patch_normal_vs_eigh(oracle):
    rng = np.random.default_rng(2)
    for _ in range(20):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        u = np.cross(n, [1, 0, 0.3]); u /= np.linalg.norm(u)
        v = np.cross(n, u)
        pts = (rng.uniform(-0.03, 0.03, (70, 1)) * u + rng.uniform(-0.03, 0.03, (70, 1)) * v +
               rng.normal(0, 0.0005, (70, 1)) * n + rng.uniform(-0.5, 0.5, 3)).astype(np.float32)
        p4 = oracle.f4(pts)
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        ok = oracle.lib().orc_cal_patch_normal(oracle._p(p4), len(p4), C.byref(a), C.byref(b), C.byref(c))
        assert ok == 1
        nn = np.array([a.value, b.value, c.value])
        w, V = np.linalg.eigh(np.cov(pts.astype(np.float64).T))
        # float single-pass covariance is noisy: agreement to ~1e-2 rad, unit length to 1e-5
        assert abs(abs(nn @ V[:, 0]) - 1) < 1e-3
        assert abs(np.linalg.norm(nn) - 1) < 1e-5


def test_lls_matches_lstsq(oracle):
    rng = np.random.default_rng(3)
    n = 400
    src = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tgt = (src + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
    s4, t4, n4 = oracle.f4(src), oracle.f4(tgt), oracle.f4(nrm.astype(np.float32))
    match = np.arange(n, dtype=np.int32)
    ATA = np.zeros(36); ATb = np.zeros(6); x = np.zeros(6); T = np.zeros(16, np.float32)
    oracle.lib().orc_p2p_lls(oracle._p(s4), oracle._p(t4), oracle._p(n4), oracle._p(match, oracle.ip), n,
                             oracle._p(ATA, oracle.dp), oracle._p(ATb, oracle.dp), oracle._p(x, oracle.dp), oracle._p(T))
    nf = n4[:, :3].astype(np.float64); sf = src.astype(np.float64); tf = tgt.astype(np.float64)
    A = np.hstack([np.cross(sf, nf), nf])
    b = np.sum(nf * (tf - sf), axis=1)
    xr = np.linalg.lstsq(A, b, rcond=None)[0]
    assert np.allclose(x, xr, rtol=1e-4, atol=1e-7)
    assert np.allclose(ATA.reshape(6, 6), A.T @ A, rtol=1e-5)
    assert abs(T[15] - 1) == 0 and abs(np.linalg.det(T.reshape(4, 4)[:3, :3].astype(np.float64)) - 1) < 1e-5


def test_icp_recovers_small_motion(oracle):
    from pwicp_amd import synth
    rng = np.random.default_rng(4)
    t, _ = synth.make_tile(4000, 0.02)
    t = (t - t.mean(0)).astype(np.float32)
    nrm = synth._normal(t[:, 0].astype(float) + 0, t[:, 1].astype(float), 4000 ** 0.5 * 0.02).astype(np.float32)
    M = synth.euler_matrix(0.002, -0.001, 0.0015, [0.003, -0.002, 0.001])
    s = (t.astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
    sn = (nrm.astype(np.float64) @ M[:3, :3].T).astype(np.float32)
    T = np.zeros(16, np.float32)
    it = oracle.lib().orc_p2p_icp(oracle._p(oracle.f4(t)), oracle._p(oracle.f4(nrm)), len(t), oracle._p(oracle.f4(s)),
                                  oracle._p(oracle.f4(sn)), len(s), 1e-6, oracle._p(T), None)
    assert 1 <= it <= 100
    Tinv = np.linalg.inv(M)
    assert np.abs(T.reshape(4, 4) - Tinv).max() < 5e-4


def test_voxel_grid_and_sor_vs_numpy(oracle):
    rng = np.random.default_rng(5)
    pts = rng.uniform(0, 1, (20000, 3)).astype(np.float32)
    leaf = np.float32(0.05)
    out = oracle.voxel_grid(pts, float(leaf))
    inv = np.float32(1.0) / leaf
    ijk = np.floor(pts * inv).astype(np.int64)
    ijk -= np.floor(pts.min(0) * inv).astype(np.int64)
    dims = ijk.max(0) + 1
    key = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    uk = np.unique(key)
    assert len(out) == len(uk)
    cen = np.array([pts[key == k].astype(np.float64).mean(0) for k in uk[:50]])
    assert np.allclose(out[:50, :3], cen, atol=2e-6)
    # SOR: k nearest other points
    kept = oracle.sor(out, 14, 1.0)
    tree = cKDTree(out[:, :3].astype(np.float64))
    d, _ = tree.query(out[:, :3].astype(np.float64), k=15)
    md = d[:, 1:].mean(1)
    thr = md.mean() + 1.0 * md.std(ddof=1)
    expect = (md <= thr).sum()
    assert abs(len(kept) - expect) <= 2


def test_percentile_and_bbox_and_angles(oracle):
    rng = np.random.default_rng(6)
    a = rng.uniform(0, 1, (3000, 3)).astype(np.float32)
    b = rng.uniform(0, 1, (1001, 3)).astype(np.float32)
    p = oracle.lib().orc_percentile_dist(oracle._p(oracle.f4(a)), len(a), oracle._p(oracle.f4(b)), len(b), 0.75)
    d, _ = cKDTree(a.astype(np.float64)).query(b.astype(np.float64))
    assert abs(p - np.sort(d)[int(np.float32(len(b)) * np.float32(0.75))]) < 1e-6
    bb = np.zeros(6)
    oracle.lib().orc_octree_bbox(oracle._p(oracle.f4(a)), len(a), 0.01, oracle._p(bb, oracle.dp))
    side = bb[3:] - bb[:3]
    assert np.allclose(side, side[0]) and abs(side[0] - 1.28) < 1e-9      # 2^7 * 0.01 cube
    assert np.all(bb[:3] <= a.min(0)) and np.all(bb[3:] >= a.max(0))
    from pwicp_amd import synth
    M = synth.euler_matrix(0.01, -0.02, 0.03, [1, 2, 3]).astype(np.float32)
    ang = oracle.matrix2angle(M)
    assert np.allclose(ang, [0.01, -0.02, 0.03], atol=1e-6)


def test_select_patches_contract(oracle):
    tgt, src, _ = _data.pair(20000)
    from pwicp_amd import synth
    lab, nsv = synth.grid_labels(tgt, 0.05)
    P = oracle.select_patches(tgt, lab, nsv)
    sizes = np.diff(P.off)
    assert P.m > 50 and sizes.min() >= 20
    assert np.array_equal(P.pat[:, :3], tgt[P.src])
    # points inside a patch keep the point order of the cloud (S.cpp:99-103)
    for i in range(0, P.m, 17):
        s = P.src[P.off[i]:P.off[i + 1]]
        assert np.all(np.diff(s) > 0)
        assert len(set(lab[s])) == 1
    # boundary points are patch points, order Xmax,Xmin,Ymax,Ymin,Zmax,Zmin
    i = 3
    seg = P.pat[P.off[i]:P.off[i + 1], :3]
    bp = P.bp[6 * i:6 * i + 6, :3]
    assert bp[0, 0] == seg[:, 0].max() and bp[1, 0] == seg[:, 0].min() and bp[5, 2] == seg[:, 2].min()
    assert np.allclose(P.ct[i, :3], seg.astype(np.float64).mean(0), atol=1e-6)
    assert np.allclose(P.ctstd, P.bpstd / sizes, rtol=1e-6)                  # sigma_CT = sigma_BP / N (B.2)


def test_loop_converges_to_ground_truth(oracle):
    tgt, src, Tgt = _data.pair(50000)
    from pwicp_amd import synth
    l1, n1 = synth.grid_labels(tgt, 0.05)
    l2, n2 = synth.grid_labels(src, 0.05)
    P1 = oracle.select_patches(tgt, l1, n1)
    P2 = oracle.select_patches(src, l2, n2)
    io = oracle.run_loop(tgt, src, P1, P2, _data.R, _data.R, 0.05, 0.05, 0.05, 0.004)
    assert io.status == 0 and 2 <= io.n_outer <= 12
    dts = np.array(io.DTseries[:io.n_outer + 1])
    assert np.all(np.diff(dts) <= 1e-9) and dts[-1] >= 0.004 - 1e-9        # monotone DT schedule
    T = np.array(io.T16, np.float64).reshape(4, 4)
    # T maps source -> target frame (both reduced by the target centroid)
    c = synth.make_tile(50000, _data.R)[0].mean(0).astype(np.float64)
    S = np.eye(4); S[:3, 3] = -c
    Tfull = np.linalg.inv(S) @ T @ S
    assert np.abs(_data.euler(Tfull) - _data.euler(Tgt)).max() < 2e-4
    assert np.abs(Tfull[:3, 3] - Tgt[:3, 3]).max() < 1e-3
    # faithful-cost mode (reference call structure) gives identical results
    io2 = oracle.run_loop(tgt, src, P1, P2, _data.R, _data.R, 0.05, 0.05, 0.05, 0.004, faithful=True)
    assert np.array_equal(np.array(io.T16), np.array(io2.T16)) and io.n_corr == io2.n_corr
