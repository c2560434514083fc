"""Segmentation front end on the device (csrc/frontend.hip): supervoxel fusion and boundary refinement as speculative fixed
points must reproduce the serial passes of the reference (supervoxel_segmentation.h:65-248) label for label.  The host
pipeline (host/frontend.cpp, $PWICP_FRONTEND=host) is the serial restatement; tests/test_host_stages.py and
tests/test_oracle_golden.py pin it to the reference's own compiled front end (oracle/_ref)."""
import os

import numpy as np
import pytest

import _data
import _golden as G

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _both(ctx, cloud, sv, spacing):
    out = {}
    for mode in ("host", "device"):
        os.environ["PWICP_FRONTEND"] = mode
        try:
            out[mode] = ctx.frontend_segment(cloud, sv, 45, spacing)
        finally:
            os.environ.pop("PWICP_FRONTEND", None)
    return out["host"], out["device"]


def _assert_same(ctx, cloud, sv, spacing):
    (lh, nh), (ld, nd) = _both(ctx, cloud, sv, spacing)
    assert nh == nd
    assert np.array_equal(lh, ld), "%d of %d labels differ" % (int((lh != ld).sum()), len(lh))
    return nd


@pytest.mark.parametrize("n", [2000, 30000, 200000])
def test_device_labels_equal_serial_labels_synthetic(ctx, n):
    tgt, src, _ = _data.pair(n)
    for cloud in (tgt, src):
        nsv = _assert_same(ctx, cloud, 10 * _data.R, _data.R)
        assert nsv > 0


@pytest.mark.parametrize("sv_factor", [3.0, 6.0, 25.0, 60.0])
def test_device_labels_equal_serial_labels_other_supervoxel_sizes(ctx, sv_factor):
    """Small supervoxels: the target count is reached in the first rounds (the round that stops inside a centre is a big one);
    large supervoxels: many rounds, long adjacency lists (search queues beyond the wavefront's LDS queue fall back to the host
    pass - same labels either way)."""
    _, src, _ = _data.pair(60000)
    _assert_same(ctx, src, sv_factor * _data.R, _data.R)


def test_device_labels_equal_serial_labels_rough_surface(ctx):
    """A noisy cliff: normals vary quickly, lambda0 is large, the fusion takes few rounds with big absorptions."""
    rng = np.random.default_rng(5)
    tgt, _, _ = _data.pair(80000)
    cloud = tgt.copy()
    cloud[:, 2] += (0.3 * np.sin(9.0 * cloud[:, 0]) + rng.normal(0, 0.004, len(cloud))).astype(np.float32)
    _assert_same(ctx, cloud, 10 * _data.R, _data.R)


@pytest.mark.parametrize("epoch", ["%03d" % e for e in range(1, 21)])
def test_device_labels_equal_serial_labels_golden_epochs(ctx, epoch):
    """The reference's own 4D series (data/data_synthetic/syntheticPC_with_transformations/Epoch_001..020.pcd, kept as
    fixtures), preprocessed as the entry points do (Res 0.005, SV 0.05)."""
    from pwicp_amd.pcd import read_pcd
    raw = read_pcd(os.path.join(HERE, "golden", "inputs", "Epoch_%s.pcd" % epoch))
    cloud = ctx.preprocess(raw, 0.005, 14, 2.0)
    cloud = (cloud - cloud.mean(axis=0)).astype(np.float32)
    nsv = _assert_same(ctx, cloud, 0.05, 0.005)
    assert nsv > 100


def _device_labels(ctx, cloud, sv, spacing):
    os.environ["PWICP_FRONTEND"] = "device"
    try:
        return ctx.frontend_segment(cloud, sv, 45, spacing)
    finally:
        os.environ.pop("PWICP_FRONTEND", None)


@pytest.mark.parametrize("epoch", list(range(1, 21)))
def test_device_labels_equal_the_references_compiled_front_end_golden_epochs(ctx, oracle, epoch):
    """Directly against the REFERENCE'S OWN front end (oracle/_ref: /root/reference/codelibrary compiled by oracle/Makefile;
    what Segmentation.cpp:18-68 runs), not against the product's serial passes: each of the reference's 20 scans, preprocessed
    and reduced as the 4D entry point does (VoxelGrid 5 mm, SOR 14 / 5.0, minus the float centroid; R.cpp:412-436)."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    from pwicp_amd.pcd import read_pcd
    cloud = G.preprocess_4d(oracle, read_pcd(G.epoch_path(epoch)))
    red, _, _ = G.reduce_pair(cloud, cloud)
    ld, nd = _device_labels(ctx, red, 0.05, 0.005)
    lr, nr = oracle.ref_frontend(red, 0.05)
    assert nd == nr
    assert np.array_equal(ld, lr), "%d of %d labels differ" % (int((ld != lr).sum()), len(lr))


def test_device_labels_equal_the_references_compiled_front_end_300k(ctx, oracle):
    """300 k synthetic points (the bench generator), device pipeline against oracle/_ref."""
    if not oracle.ref_frontend_available():
        pytest.skip("oracle/_ref not built")
    tgt, src, _ = _data.pair(300000)
    for cloud in (tgt, src):
        ld, nd = _device_labels(ctx, cloud, 10 * _data.R, _data.R)
        lr, nr = oracle.ref_frontend(cloud, 10 * _data.R)
        assert nd == nr and np.array_equal(ld, lr)


def test_one_million_points(ctx):
    tgt, _, _ = _data.pair(1000000)
    _assert_same(ctx, tgt, 10 * _data.R, _data.R)


def test_search_queue_overflow_falls_back_to_the_serial_pass(ctx, capfd):
    """A search wider than the wavefront's queue makes the device fusion give up; the serial host pass takes over (refinement
    stays on the device) - same labels.  $PWICP_FUSION_QUEUE shrinks the queue to force it."""
    _, src, _ = _data.pair(30000)
    os.environ["PWICP_FUSION_QUEUE"] = "50"
    os.environ["PWICP_TRACE"] = "1"
    try:
        _assert_same(ctx, src, 10 * _data.R, _data.R)
    finally:
        os.environ.pop("PWICP_FUSION_QUEUE", None)
        os.environ.pop("PWICP_TRACE", None)
    assert "fusion gives up" in capfd.readouterr().err


@pytest.mark.parametrize("knobs", [{"PWICP_FUSION_CHUNK": "1"}, {"PWICP_FUSION_CHUNK": "2"}, {"PWICP_FUSION_CHUNK": "4"},
                                   {"PWICP_FUSION_CHUNK": "16", "PWICP_FUSION_WAKE_DIV": "1"},
                                   {"PWICP_FUSION_CHUNK": "3", "PWICP_FUSION_WAKE_DIV": "100000"},
                                   {"PWICP_FUSION_TILE": "0"}, {"PWICP_FUSION_TILE": "4", "PWICP_FUSION_CHUNK_DIV": "64"},
                                   {"PWICP_FUSION_TILE": "300", "PWICP_FUSION_CHUNK": "5"},
                                   {"PWICP_FUSION_COLOURS": "1"}, {"PWICP_FUSION_COLOURS": "4", "PWICP_FUSION_TILE": "7"},
                                   {"PWICP_FE_AHEAD": "0"}, {"PWICP_FE_PIECES": "3"}, {"PWICP_FE_PIECES": "1"}])
def test_labels_do_not_depend_on_the_sweep_schedule(ctx, knobs):
    """The fixed point is the serial result whatever the schedule: chunks of 1 (Jacobi) ... 16 centres per wavefront
    (Gauss-Seidel inside a chunk), work lists from the first sweep on (WAKE_DIV 1) or hardly ever (100000), the full sweeps in
    index order (TILE 0) or tile by tile with tiles of 4 / 300 centres, 64 instead of 2048 chunks wanted per sweep, the tiles of a
    sweep all at once, in two colours (the default) or in four.  And whatever the pipeline around the normals: everything after them
    (FE_AHEAD 0: no reverse index / cell count beside the host's eigen step, the counts not taken inside the k-NN launch), the scatter
    sums in three pieces or in one."""
    tgt, _, _ = _data.pair(400000)
    os.environ.update(knobs)
    try:
        _assert_same(ctx, tgt, 10 * _data.R, _data.R)
    finally:
        for k in knobs:
            os.environ.pop(k, None)


def test_degenerate_inputs(ctx):
    """Exact ties and degenerate neighbourhoods: a perfectly regular planar lattice (every distance tied, normals exactly +-z,
    metric exactly 0 between neighbours), a cloud with 10 % duplicated points (zero distances, rank-deficient scatter), a
    cloud barely larger than k."""
    g = np.arange(120, dtype=np.float32) * np.float32(0.005)
    lattice = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    lattice = np.concatenate([lattice, np.zeros((len(lattice), 1), np.float32)], 1).astype(np.float32)
    _assert_same(ctx, lattice, 10 * _data.R, _data.R)
    tgt, _, _ = _data.pair(20000)
    rng = np.random.default_rng(11)
    dup = np.concatenate([tgt, tgt[rng.integers(0, len(tgt), len(tgt) // 10)]]).astype(np.float32)
    _assert_same(ctx, dup, 10 * _data.R, _data.R)
    _assert_same(ctx, tgt[:60], 10 * _data.R, _data.R)


def test_refinement_sweep_cap_falls_back_to_the_serial_pass(ctx, capfd):
    """A queue generation that has not settled after $PWICP_REFINE_SWEEPS sweeps (default 4096; 2-5 are normal) is redone by the
    serial host pass from the fused labels - same labels."""
    _, src, _ = _data.pair(30000)
    os.environ["PWICP_REFINE_SWEEPS"] = "1"
    os.environ["PWICP_TRACE"] = "1"
    try:
        _assert_same(ctx, src, 10 * _data.R, _data.R)
    finally:
        os.environ.pop("PWICP_REFINE_SWEEPS", None)
        os.environ.pop("PWICP_TRACE", None)
    assert "refinement gives up" in capfd.readouterr().err


def test_rockfall_scale(ctx):
    """BASELINE configs[2] stand-in (SURVEY 8d cfg 3): point spacing 0.3 m, supervoxels of 3 m, coordinates of a few hundred
    metres after the centroid reduction."""
    tgt, _, _ = _data.pair(150000)
    big = (tgt * np.float32(60.0)).astype(np.float32)
    _assert_same(ctx, big, 3.0, 0.3)


def test_randomised_clouds_and_schedules():
    """tools/fe_fuzz.py: 24 random combinations of size, roughness, point order, supervoxel size, chunk size, wake threshold
    and queue limit, device labels against the serial passes."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "fe_fuzz.py"), "24", "2026"], capture_output=True,
                         text=True, timeout=600)
    assert "different: 0 of 24" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("k", [10, 48, 49, 64])
def test_other_neighbourhood_sizes(ctx, k):
    """k = 45 in the reference (CommonFunc.h:41); the device pipeline takes any k <= 64 (k > 48: the k-NN kernel with its lists
    in global memory)."""
    _, src, _ = _data.pair(40000)
    out = {}
    for mode in ("host", "device"):
        os.environ["PWICP_FRONTEND"] = mode
        try:
            out[mode] = ctx.frontend_segment(src, 10 * _data.R, k, _data.R)
        finally:
            os.environ.pop("PWICP_FRONTEND", None)
    assert out["host"][1] == out["device"][1] and np.array_equal(out["host"][0], out["device"][0])
