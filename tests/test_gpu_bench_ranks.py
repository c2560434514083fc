"""bench.py's N > 1 path, rehearsed on the 1-GPU box: `--gpus 2` must start two ranks BY ITSELF (no launcher environment),
report n_gpus == 2, and the records it gathers (pairs of the end-to-end series dealt p -> rank p mod 2, R.cpp:89-187: the
reference's pair loop, whose iterations are independent) must equal the ones a single rank produces.  Both ranks share
GPU 0 (--single-device); the exchange runs over gloo, and — where RCCL accepts two ranks on one device — over RCCL too."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--points", "200000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-inner-timing",
          "--series-epochs", "4", "--pairs-in-flight", "0", "--large-points", "0"]


def _bench(extra, dump, timeout=900, env_extra=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + ["--dump-records", str(dump)] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.fixture(scope="module")
def one_rank(tmp_path_factory):
    d = tmp_path_factory.mktemp("bench1")
    p, line = _bench(["--gpus", "1"], d / "r1.npy")
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 1 and line["series_end_to_end"]["all_pairs_ok"]
    return line, np.load(d / "r1.npy")


def _same_records(a, b):
    assert a.dtype == b.dtype and len(a) == len(b)
    for f in ("pair", "status", "n_outer", "n_inner", "n_corr"):
        assert np.array_equal(a[f], b[f]), f
    assert a["T"].tobytes() == b["T"].tobytes()
    assert a["VCM"].tobytes() == b["VCM"].tobytes()


def test_bench_gpus_2_starts_two_ranks_and_gathers_the_same_records(tmp_path, one_rank):
    line1, rec1 = one_rank
    p, line = _bench(["--gpus", "2", "--single-device", "--backend", "gloo"], tmp_path / "r2.npy")
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 2
    assert line["config"]["parallelism"] == "pair-per-gpu x2"
    s = line["series_end_to_end"]
    assert s["n_gpus"] == 2 and s["pairs"] == 4 and s["all_pairs_ok"]
    # weak scaling of the loop-only figure: both ranks ran `steps` registrations
    assert line["config"]["correspondences_per_step"] == line1["config"]["correspondences_per_step"]
    _same_records(np.load(tmp_path / "r2.npy"), rec1)
    # the shared target of the series was segmented ONCE: by rank 0; rank 1 preprocessed it and took the labels from the broadcast
    #  - and the records above are byte-equal to one rank's all the same
    assert s["target_labels_by_rank"] == [[0, 1], [1, 0]], s["target_labels_by_rank"]
    assert line1["series_end_to_end"]["target_labels_by_rank"] == [[0, 1]]


def test_bench_two_ranks_each_segmenting_the_target_give_the_same_records(tmp_path, one_rank):
    """PWICP_SHARE_TARGET=0 (round 4: every rank runs the target's front end itself): the same records."""
    _, rec1 = one_rank
    p, line = _bench(["--gpus", "2", "--single-device", "--backend", "gloo"], tmp_path / "r2.npy", env_extra={"PWICP_SHARE_TARGET": "0"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["series_end_to_end"]["target_labels_by_rank"] == [[0, 1], [0, 1]]
    _same_records(np.load(tmp_path / "r2.npy"), rec1)


def test_bench_refuses_a_world_other_than_gpus(tmp_path):
    """never an n_gpus smaller than asked: a launcher environment of another size is an error, not a 1-rank run"""
    p, line = _bench(["--gpus", "2"], tmp_path / "x.npy", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and line is None
    assert "--gpus 2" in p.stderr
    # two ranks asked for on real devices, one device present, no --single-device: refused as well
    import torch
    if torch.cuda.device_count() < 2:
        p, line = _bench(["--gpus", "2", "--backend", "gloo"], tmp_path / "y.npy")
        assert p.returncode != 0 and line is None


def test_bench_gpus_2_over_rccl_on_one_device(tmp_path, one_rank):
    """The same over RCCL (backend nccl).  RCCL refuses two ranks of one communicator on the same device
    ("Duplicate GPU detected") unless that check is relaxed; where it still refuses, that is what this test records."""
    _, rec1 = one_rank
    p, line = _bench(["--gpus", "2", "--single-device", "--backend", "nccl"], tmp_path / "r2n.npy", timeout=600,
                     env_extra={"NCCL_DEBUG": "WARN"})
    if p.returncode != 0:
        txt = (p.stderr + p.stdout)
        assert "uplicate GPU" in txt or "invalid usage" in txt.lower() or "ncclInvalidUsage" in txt, txt[-3000:]
        pytest.skip("RCCL does not take two ranks on one device on this box (Duplicate GPU): needs a 2-GPU node")
    assert line["n_gpus"] == 2
    _same_records(np.load(tmp_path / "r2n.npy"), rec1)
