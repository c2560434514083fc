"""Pins the oracle end-to-end against ALL of the reference's own checked-in per-pair results
(results/4DPCReg/<e>_{Direct2Ref,Adaptive,Fixed}_TransMatrix.txt: 57 files, 46 distinct (target, source) pairs, produced by
the reference's Windows/PCL-1.8.1 build with configuration_files/configuration_4d.txt and pairMode 0 / -1 / 3).
Pipeline: VoxelGrid(5 mm) + SOR(14, 5.0) -> centroid reduction -> the REFERENCE'S OWN front end (oracle/_ref, compiled from
/root/reference/codelibrary) -> oracle patch selection -> oracle loop -> T_final.  The committed inputs are the reference's
20 scans (tests/golden/inputs).  Per-file tolerances: tests/golden/tolerance_table.json (tools/golden_report.py: twice the
measured distance, floored at float print precision 2e-7 rad / 3e-7 m; 55 files sit on the floor, Fixed e11 at 6.3e-6 rad).

What it took to get there (tools/rootcause_golden.py): the reference's VoxelGrid sums the points of a voxel in the order an
UNSTABLE std::sort leaves them in; with input order instead, Direct2Ref e8 / e19 miss their files by 1.8e-5 / 8.7e-4 rad."""
import json
import os

import numpy as np
import pytest

import _golden as G

with open(os.path.join(G.GOLD, "tolerance_table.json")) as _f:
    _TT = json.load(_f)
TOL = {m: {int(e): tuple(v) for e, v in t.items()} for m, t in _TT["tol"].items()}
AMAP = {int(e): t for e, t in _TT["pair_map"]["Adaptive"].items()}
FMAP = {int(e): t for e, t in _TT["pair_map"]["Fixed"].items()}       # pairMode 3: target = max(e - 3, 1) (R.cpp:94-97)
# a11 (calTransParaVCM, R.cpp:1273-1343; written at R.cpp:492-540): per file [rel. tolerance of the six Std_ values, abs. tolerance
# of the 36 VCM entries] - same rule as TOL: twice the oracle's measured distance, floored at 2e-5 / 1.5e-12 (the file prints the
# matrix with 12 decimals; 52 of the 57 files sit on that floor, i.e. every printed digit agrees)
STOL = {m: {int(e): tuple(v) for e, v in t.items()} for m, t in _TT["sigma_vcm_tol"].items()}


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
    assert np.abs(G.euler(Tf) - G.euler(Tg)).max() < TOL["Direct2Ref"][2][0]
    assert np.abs(Tf[:3, 3].astype(float) - Tg[:3, 3]).max() < TOL["Direct2Ref"][2][1]
    V = np.array(io.VCM).reshape(6, 6)
    assert np.allclose(G.printed_sigmas(V), stds, rtol=1e-5)          # the printed sigmas (10 digits): same stable set, same residuals
    assert np.abs(V - Vg).max() < STOL["Direct2Ref"][2][1]            # every printed digit of the 6x6 matrix


_PREP, _PAIR = {}, {}       # preprocessed epochs and finished (target, source) pairs, shared by the three families


def _family(oracle, mode, pair_map):
    """Every file of one family against the oracle; returns {epoch: (d_angle, d_trans, io)}."""
    from pwicp_amd.pcd import read_pcd

    def cloud(e):
        if e not in _PREP:
            _PREP[e] = G.preprocess_4d(oracle, read_pcd(G.epoch_path(e)))
        return _PREP[e]

    rows = {}
    for e in range(2, 21):
        if (pair_map[e], e) not in _PAIR:
            r1, r2, shift = G.reduce_pair(cloud(pair_map[e]), cloud(e))
            l1, n1 = oracle.ref_frontend(r1, 0.05)
            l2, n2 = oracle.ref_frontend(r2, 0.05)
            io = oracle.run_loop(r1, r2, oracle.select_patches(r1, l1, n1), oracle.select_patches(r2, l2, n2),
                                 0.005, 0.005, 0.05, 0.05, 0.05, 0.004)
            _PAIR[(pair_map[e], e)] = (io, G.final_matrix(io.T16, shift))
        io, Tf = _PAIR[(pair_map[e], e)]
        Tg, Vg, stds = G.parse_transmatrix_file(os.path.join(G.GOLD, "reference_results", "%d_%s_TransMatrix.txt" % (e, mode)))
        V = np.array(io.VCM).reshape(6, 6)
        rows[e] = (float(np.abs(G.euler(Tf) - G.euler(Tg)).max()), float(np.abs(Tf[:3, 3].astype(float) - Tg[:3, 3]).max()), io,
                   float(np.abs(G.printed_sigmas(V) / stds - 1).max()), float(np.abs(V - Vg).max()))
    return rows


@pytest.mark.parametrize("mode", ["Direct2Ref", "Adaptive", "Fixed"])
def test_every_result_file_of_the_reference(oracle, mode):
    """19 files per family, each within its entry of tests/golden/tolerance_table.json; north_star's 1e-5 rad / 1e-4 m holds
    for all 57 with a margin of 3x (worst: Fixed e11, 3.1e-6 rad - one refinement decision |d| < 2 sigma of epoch 11 at a
    relative margin of 5.8e-6, S.cpp:220-225)."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    pair_map = {"Direct2Ref": {e: 1 for e in range(2, 21)}, "Adaptive": AMAP, "Fixed": FMAP}[mode]
    rows = _family(oracle, mode, pair_map)
    for e, (da, dt, io, ds, dv) in rows.items():
        assert io.status == 0
        assert da < TOL[mode][e][0] and dt < TOL[mode][e][1], (mode, e, da, dt)
        assert da < 1e-5 and dt < 1e-4
        # the reference's own VCM and sigmas of the SAME file (a11): all 57, not one
        assert ds < STOL[mode][e][0] and dv < STOL[mode][e][1], (mode, e, ds, dv)
    assert sum(1 for v in rows.values() if v[0] < 2e-7 and v[1] < 3e-7) >= 18
    assert sum(1 for v in rows.values() if v[3] < 2e-5 and v[4] < 1.5e-12) >= 14      # (Direct2Ref: 14, Adaptive: 19, Fixed: 17)


def test_adaptive_pair_sequence_matches_reference_outputs(oracle):
    """calAdaptivePairSequence (R.cpp:552-589) with the oracle's overlap ratio on the reference's 20 raw epochs
    reproduces the pair map recovered from the reference's own Adaptive result files."""
    from pwicp_amd.pcd import read_pcd
    clouds = [oracle.f4(read_pcd(G.epoch_path(e))) for e in range(1, 21)]
    expect = AMAP
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


@pytest.mark.parametrize("e", [2, 11, 16])
def test_no_transformations_fixture_is_the_oracles_result(oracle, e):
    """tests/golden/no_transformations_expected.json (what the GPU tests on the reference's SECOND input set,
    data/data_synthetic/syntheticPC_no_transformations, compare against) is re-derived here for three of its 19 pairs:
    identical counts and DT series, the matrix to the last float bit."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_nt", os.path.join(G.GOLD, "make_no_transformations_expected.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from pwicp_amd.pcd import read_pcd
    inputs = os.path.join(G.GOLD, "inputs_no_transformations")
    if "nt" not in _PREP:
        _PREP["nt"] = G.preprocess_4d(oracle, read_pcd(os.path.join(inputs, "Epoch_001.pcd")))
    got = gen.oracle_pair(_PREP["nt"], e, inputs)
    exp = json.load(open(os.path.join(G.GOLD, "no_transformations_expected.json")))["pairs"][str(e)]
    assert got == exp


def test_no_transformations_identity_property():
    """Expected transformation of the untransformed series = identity.  13 of the 19 pairs come back to it within the
    reference's own accuracy on the transformed series (largest entries of its TransPara_AbsError.txt: 57.1 mgon, 1.14 mm);
    e2 / e6 / e7 keep the angles but slide along y (4.6 / 4.7 / 35 mm), e11 / e16 / e19 rest on few stable patches - the
    method's behaviour on these scans (the oracle is pinned by the 57 result files above), recorded here as it is."""
    exp = json.load(open(os.path.join(G.GOLD, "no_transformations_expected.json")))["pairs"]
    ang, tr = 58.0 * np.pi / 200000.0, 1.2e-3
    ok = {int(e) for e, v in exp.items() if np.abs(v["euler_rad"]).max() < ang and np.abs(v["t_m"]).max() < tr}
    assert ok == set(range(2, 21)) - {2, 6, 7, 11, 16, 19}
    for e in (2, 6, 7):
        assert np.abs(exp[str(e)]["euler_rad"]).max() < ang and abs(exp[str(e)]["t_m"][1]) > tr
