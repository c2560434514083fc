"""Pins the oracle end-to-end against the reference's own checked-in results
(results/4DPCReg/<e>_Direct2Ref_TransMatrix.txt, produced by the reference's Windows/PCL-1.8.1 build with
configuration_files/configuration_4d.txt).  Pipeline: VoxelGrid(5 mm) + SOR(14, 5.0) -> centroid reduction ->
the REFERENCE'S OWN front end (oracle/_ref, compiled from /root/reference/codelibrary) -> oracle patch
selection -> oracle loop -> T_final.  The committed inputs are Epoch_001/002 (data files of the reference);
the remaining epochs are read from /root/reference when it is mounted."""
import json
import os

import numpy as np
import pytest

import _golden as G

# tolerance per epoch (rad, m): float print precision for the well-conditioned pairs; e8 and e19 are the
# flip-sensitive ones (one patch classified differently; <= 65 stable patches left), see SURVEY App. D
TOL = {e: (5e-6, 5e-6) for e in range(2, 21)}
TOL[8] = (5e-5, 5e-5)
TOL[19] = (2e-3, 3e-3)


@pytest.fixture(scope="module")
def target(oracle):
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref/libref_frontend.so not built (needs /root/reference at build time)")
    from pwicp_amd.pcd import read_pcd
    c1 = read_pcd(G.epoch_path(1))
    assert len(c1) == 174474
    return G.preprocess_4d(oracle, c1)


def register(oracle, p1, e):
    from pwicp_amd.pcd import read_pcd
    p2 = G.preprocess_4d(oracle, read_pcd(G.epoch_path(e)))
    r1, r2, shift = G.reduce_pair(p1, p2)
    l1, n1 = oracle.ref_frontend(r1, 0.05)
    l2, n2 = oracle.ref_frontend(r2, 0.05)
    P1 = oracle.select_patches(r1, l1, n1)
    P2 = oracle.select_patches(r2, l2, n2)
    io = oracle.run_loop(r1, r2, P1, P2, 0.005, 0.005, 0.05, 0.05, 0.05, 0.004)
    return io, G.final_matrix(io.T16, shift), (len(r1), len(r2), P1.m, P2.m)


def test_epoch2_matches_reference_result(oracle, target):
    io, Tf, sizes = register(oracle, target, 2)
    assert sizes == (142402, 140662, 1822, 1846)
    assert io.status == 0 and io.n_outer == 4 and list(io.n_inner[:4]) == [5, 3, 2, 2]
    Tg, Vg, stds = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "2_Direct2Ref_TransMatrix.txt"))
    assert np.abs(G.euler(Tf) - G.euler(Tg)).max() < 5e-6
    assert np.abs(Tf[:3, 3].astype(float) - Tg[:3, 3]).max() < 5e-6
    V = np.array(io.VCM).reshape(6, 6)
    mine = np.concatenate([1000 * 63.6619772368 * np.sqrt(np.diag(V)[:3]), 1000 * np.sqrt(np.diag(V)[3:])])
    assert np.allclose(mine, stds, rtol=5e-3)
    assert np.allclose(V, Vg, atol=2e-11, rtol=2e-2)


@pytest.mark.skipif(not os.path.isdir(G.REF_ROOT), reason="reference tree not mounted")
def test_all_direct2ref_pairs(oracle, target):
    rows = {}
    for e in range(2, 21):
        io, Tf, sizes = register(oracle, target, e)
        Tg, Vg, stds = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "%d_Direct2Ref_TransMatrix.txt" % e))
        da = float(np.abs(G.euler(Tf) - G.euler(Tg)).max())
        dt = float(np.abs(Tf[:3, 3].astype(float) - Tg[:3, 3]).max())
        rows[e] = dict(d_angle_rad=da, d_trans_m=dt, outer=io.n_outer, inner=list(io.n_inner[:io.n_outer]),
                       stable=list(io.n_stable[:io.n_outer]), patches=[sizes[2], sizes[3]])
        assert io.status == 0
        assert da < TOL[e][0] and dt < TOL[e][1], (e, da, dt)
    tight = [e for e in rows if rows[e]["d_angle_rad"] < 1e-6 and rows[e]["d_trans_m"] < 1e-6]
    assert len(tight) >= 16
    # the committed report tests/golden/oracle_vs_reference.json is only rewritten on request (keeps the work tree clean)
    if os.environ.get("PWICP_WRITE_GOLDEN_REPORT"):
        with open(os.path.join(G.GOLD, "oracle_vs_reference.json"), "w") as f:
            json.dump(rows, f, indent=1)


@pytest.mark.skipif(not os.path.isdir(G.REF_ROOT), reason="reference tree not mounted")
def test_adaptive_pair_sequence_matches_reference_outputs(oracle):
    """calAdaptivePairSequence (R.cpp:552-589) with the oracle's overlap ratio on the reference's 20 raw epochs
    reproduces the pair map recovered from the reference's own Adaptive result files."""
    from pwicp_amd.pcd import read_pcd
    clouds = [oracle.f4(read_pcd(G.epoch_path(e))) for e in range(1, 21)]
    expect = {2: 1, 3: 1, 4: 1, 5: 1, 6: 1, 7: 3, 8: 4, 9: 4, 10: 5, 11: 6, 12: 6, 13: 7, 14: 9, 15: 12, 16: 13, 17: 14,
              18: 14, 19: 14, 20: 14}
    got = {}
    idx_target = 0
    for j in range(1, 20):
        for i in range(idx_target, j):
            r = oracle.lib().orc_overlap_ratio(oracle._p(clouds[i]), len(clouds[i]), oracle._p(clouds[j]), len(clouds[j]), 0.05)
            idx_target = i
            if r > 0.75:
                break
        got[j + 1] = idx_target + 1
    assert got == expect


AMAP = {2: 1, 3: 1, 4: 1, 5: 1, 6: 1, 7: 3, 8: 4, 9: 4, 10: 5, 11: 6, 12: 6, 13: 7, 14: 9, 15: 12, 16: 13, 17: 14, 18: 14,
        19: 14, 20: 14}


@pytest.mark.skipif(not os.path.isdir(G.REF_ROOT), reason="reference tree not mounted")
def test_adaptive_pairs_match_reference_results(oracle):
    """Second family of known answers: the reference's <e>_Adaptive_TransMatrix.txt files (source e registered to
    the adaptive target AMAP[e], not to epoch 1).  Only pairs whose target differs from epoch 1 are new information."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    from pwicp_amd.pcd import read_pcd
    prep = {}

    def cloud(e):
        if e not in prep:
            prep[e] = G.preprocess_4d(oracle, read_pcd(G.epoch_path(e)))
        return prep[e]

    rows = {}
    for e in (7, 9, 12, 14, 15, 17):
        t = AMAP[e]
        r1, r2, shift = G.reduce_pair(cloud(t), cloud(e))
        l1, n1 = oracle.ref_frontend(r1, 0.05)
        l2, n2 = oracle.ref_frontend(r2, 0.05)
        io = oracle.run_loop(r1, r2, oracle.select_patches(r1, l1, n1), oracle.select_patches(r2, l2, n2),
                             0.005, 0.005, 0.05, 0.05, 0.05, 0.004)
        Tf = G.final_matrix(io.T16, shift)
        Tg, Vg, stds = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "%d_Adaptive_TransMatrix.txt" % e))
        da = float(np.abs(G.euler(Tf) - G.euler(Tg)).max())
        dt = float(np.abs(Tf[:3, 3].astype(float) - Tg[:3, 3]).max())
        rows[e] = (da, dt)
        assert io.status == 0
    # all within registration-noise level, most at float print precision
    assert all(v[0] < 1e-4 and v[1] < 2e-4 for v in rows.values()), rows
    assert sum(1 for v in rows.values() if v[0] < 2e-6 and v[1] < 2e-6) >= 4, rows
