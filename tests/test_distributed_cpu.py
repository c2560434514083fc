"""4D series: pair schedule, sharding, gather (world_size 2, gloo, CPU) and composition to the reference."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import _golden as G
from pwicp_amd import fourd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_schedule_modes():
    assert fourd.pair_schedule(0, 5, 0) == [(0, 1), (0, 2), (0, 3), (0, 4)]
    assert fourd.pair_schedule(0, 6, 2) == [(0, 1), (0, 2), (1, 3), (2, 4), (3, 5)]
    assert fourd.pair_schedule(0, 4, -1, {1: 0, 2: 0, 3: 1}) == [(0, 1), (0, 2), (1, 3)]
    assert fourd.shard(list(range(5)), 1, 2) == [(1, 1), (3, 3)]


def _read_matrices(path, n):
    vals = open(path).read().split()
    T, V, pos = [], [], 0
    for _ in range(n):
        pos += 1
        T.append(np.array(vals[pos:pos + 16], float).reshape(4, 4).astype(np.float32)); pos += 16
        V.append(np.array(vals[pos:pos + 36], float).reshape(6, 6)); pos += 36
    return T, V


def test_compose_adaptive_matches_reference_files():
    """calTransToReferenceEpoch on the reference's own TransMatrices.txt reproduces its TransMatrices_toRef.txt
    with the adaptive pair map recovered in SURVEY §4."""
    gold = os.path.join(G.GOLD, "reference_results")
    T, V = _read_matrices(os.path.join(gold, "TransMatrices.txt"), 19)
    Tr, Vr = _read_matrices(os.path.join(gold, "TransMatrices_toRef.txt"), 19)
    amap = {2: 1, 3: 1, 4: 1, 5: 1, 6: 1, 7: 3, 8: 4, 9: 4, 10: 5, 11: 6, 12: 6, 13: 7, 14: 9, 15: 12, 16: 13, 17: 14,
            18: 14, 19: 14, 20: 14}
    rel = {s - 1: t - 1 for s, t in amap.items()}          # relative to startEpoch (R.cpp:570)
    T2, V2 = fourd.compose_to_reference(T, V, -1, rel)
    for i in range(19):
        assert np.abs(T2[i] - Tr[i]).max() < 5e-6
        assert np.allclose(V2[i], Vr[i], rtol=2e-3, atol=3e-12)


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(%r, "piecewise-icp_amd"))
    from pwicp_amd import fourd
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_pairs = int(os.environ.get("PWICP_TEST_PAIRS", "5"))
    pairs = fourd.pair_schedule(0, n_pairs + 1, 0)            # 5 pairs over 2 ranks; 8 over 8 (BASELINE configs[3]: an epoch per GPU)
    mine = fourd.shard(pairs, rank, world)
    recs = []
    for pid, (t, s) in mine:                                  # stand-in registrar: deterministic function of the pair
        T = np.eye(4, dtype=np.float32); T[0, 3] = 0.5 * s + t
        V = np.full((6, 6), float(pid))
        recs.append(fourd.pack_record(pid, 0, 3 + pid, 7, T, V, 1000 * pid))
    table = fourd.gather_records(recs, len(pairs), world, dist=dist)
    assert sorted(table) == list(range(n_pairs)), sorted(table)
    assert [pid for pid, _ in mine] == [p for p in range(n_pairs) if p %% world == rank]      # pair p -> rank p mod G
    for pid, (t, s) in enumerate(pairs):
        r = table[pid]
        assert r["n_outer"] == 3 + pid and r["n_corr"] == 1000 * pid and abs(r["T"][3] - (0.5 * s + t)) < 1e-6
        assert r["VCM"][35] == float(pid)
    dist.barrier()
    if rank == 0:
        print("GATHER_OK")
    dist.destroy_process_group()
''')


def test_shard_and_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout


def test_shard_and_gather_world8_gloo(tmp_path):
    """The shape the 8-GPU node will run (BASELINE configs[3]: 8 source epochs, one pair per rank; R.cpp:89-187's iterations are
    independent): pair p -> rank p mod 8, ONE all-gather of the 384-byte records, every rank holding the whole table afterwards.
    Eight gloo ranks on the CPU."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER % ROOT)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, PWICP_TEST_PAIRS="8", OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout


def test_record_is_384_bytes():
    r = fourd.pack_record(3, 0, 4, 9, np.eye(4), np.zeros((6, 6)), 12345)
    assert r.nbytes == 384 == fourd.RECORD_BYTES


SERIES_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(%(root)r, "piecewise-icp_amd"))
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import pwicp_amd as P
    from pwicp_amd import fourd
    from test_distributed_cpu import _read_matrices, ADAPTIVE_REL
    import _golden as G
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    targets = np.array([ADAPTIVE_REL[k] for k in range(1, 20)], np.int32)
    # adaptive map handed over (as rank 0 would broadcast it): no GPU is needed to open the series or to write results
    with P.Series(%(cfg)r, 0, 20, -1, 0.75, 0, targets) as s:
        assert s.num_pairs == 19 and s.num_scans == 20
        assert s.pair_epochs(6) == (int(targets[6]), 7, 8)
        assert np.array_equal(s.adaptive_targets(), targets)
        T, V = _read_matrices(os.path.join(G.GOLD, "reference_results", "TransMatrices.txt"), 19)
        mine = [fourd.pack_record(p, 0, 5, 12, T[p], V[p], 1000) for p in range(19) if p %% world == rank]
        table = fourd.gather_records(mine, 19, world, dist=dist)
        if rank == 0:
            recs = np.concatenate([table[p].reshape(1) for p in sorted(table)])
            s.write_results(recs[::-1].copy())               # any order
        try:
            s.run_pair(0)
            raise SystemExit("run_pair must fail without a GPU")
        except P.PwicpError:
            pass
    dist.barrier()
    if rank == 0:
        print("SERIES_OK")
    dist.destroy_process_group()
''')

ADAPTIVE_REL = {s - 1: t - 1 for s, t in {2: 1, 3: 1, 4: 1, 5: 1, 6: 1, 7: 3, 8: 4, 9: 4, 10: 5, 11: 6, 12: 6, 13: 7, 14: 9,
                                          15: 12, 16: 13, 17: 14, 18: 14, 19: 14, 20: 14}.items()}


def test_series_records_to_reference_files_world2_gloo(tmp_path):
    """The multi-GPU series path without its GPU step: two ranks open the series, contribute the reference's own
    pairwise results as records, rank 0 writes the files through libpwicp.so: TransMatrices.txt and the composition
    TransMatrices_toRef.txt must reproduce the reference's checked-in files."""
    from pwicp_amd.pcd import write_pcd_binary as write_pcd
    inp = tmp_path / "scans"
    inp.mkdir()
    for e in range(1, 21):
        write_pcd(str(inp / ("Epoch_%03d.pcd" % e)), np.zeros((3, 3), np.float32))
    out = str(tmp_path) + "/res_"
    cfg = tmp_path / "cfg.txt"
    cfg.write_text("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                   "float PCres1 (m): 0.005\nfloat PCres2 (m): 0.005\nfloat SVsize1 (m): 0.05\nfloat SVsize2 (m): 0.05\n"
                   "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): 0.05\nfloat DTmin (m): 0.004\n"
                   "bool isVisual (yes-1, no-0): 0" % (str(inp), out))
    script = tmp_path / "worker.py"
    script.write_text(SERIES_WORKER % {"root": ROOT, "cfg": str(cfg)})
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "SERIES_OK" in res.stdout
    gold = os.path.join(G.GOLD, "reference_results")
    T, V = _read_matrices(out + "TransMatrices.txt", 19)
    Tg, Vg = _read_matrices(os.path.join(gold, "TransMatrices.txt"), 19)
    Tr, Vr = _read_matrices(out + "TransMatrices_toRef.txt", 19)
    Trg, Vrg = _read_matrices(os.path.join(gold, "TransMatrices_toRef.txt"), 19)
    for i in range(19):
        assert np.array_equal(T[i], Tg[i]) and np.allclose(V[i], Vg[i], rtol=1e-9, atol=0)
        assert np.abs(Tr[i] - Trg[i]).max() < 5e-6 and np.allclose(Vr[i], Vrg[i], rtol=2e-3, atol=3e-12)
    assert os.path.exists(out + "8_Adaptive_TransMatrix.txt") and os.path.exists(out + "TransParameters_toRef.txt")
    a = np.loadtxt(out + "TransParameters.txt", skiprows=1)
    b = np.loadtxt(os.path.join(gold, "TransParameters.txt"), skiprows=1)
    # the standard deviations come from the VCM as printed (12 decimals) in the reference's TransMatrices.txt
    assert a.shape == b.shape and np.array_equal(a[:, :7], b[:, :7]) and np.allclose(a[:, 7:], b[:, 7:], rtol=1e-3)


def test_series_driver_failure_on_one_rank_ends_every_rank(tmp_path):
    """a rank that fails before a collective (here: the configuration file is unreadable on every rank, and
    in a second run only rank 1 fails) must not leave the others blocked in a broadcast / all-gather: every rank agrees
    on the failure (all-reduce MIN) and returns False."""
    worker = tmp_path / "w.py"
    worker.write_text(textwrap.dedent('''
        import os, sys
        sys.path.insert(0, os.path.join(%r, "piecewise-icp_amd"))
        from pwicp_amd import series
        rank = int(os.environ["RANK"])
        cfg = sys.argv[1] if (sys.argv[2] == "all" or rank == 1) else sys.argv[3]
        ok = series.run_series(cfg, 0, 3, int(sys.argv[4]), 0.75, backend="gloo", single_device=True, timeout_s=60)
        print("RANK%%d_RETURNED_%%s" %% (rank, ok))
        sys.exit(0 if not ok else 3)
    ''' % ROOT))
    from pwicp_amd.pcd import write_pcd_binary as write_pcd
    inp = tmp_path / "scans"
    inp.mkdir()
    for e in range(1, 4):
        write_pcd(str(inp / ("Epoch_%03d.pcd" % e)), np.zeros((3, 3), np.float32))
    good = tmp_path / "cfg.txt"
    good.write_text("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                    "float PCres1 (m): 0.005\nfloat PCres2 (m): 0.005\nfloat SVsize1 (m): 0.05\nfloat SVsize2 (m): 0.05\n"
                    "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): 0.05\nfloat DTmin (m): 0.004\n"
                    "bool isVisual (yes-1, no-0): 0" % (str(inp), str(tmp_path) + "/res_"))
    for who, mode in (("all", 0), ("one", 0), ("all", -1)):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(worker),
                              str(tmp_path / "missing.txt"), who, str(good), str(mode)],
                             capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        assert "RANK0_RETURNED_False" in res.stdout and "RANK1_RETURNED_False" in res.stdout


def test_bench_refuses_a_launcher_world_other_than_gpus():
    """bench.py --gpus N must never report fewer ranks than asked: with a launcher environment of another
    size it exits non-zero before touching a device."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus 8" in p.stderr and not p.stdout.strip()


def test_rccl_id_file_of_an_earlier_launch_with_the_same_token_is_not_taken(tmp_path):
    """ADVICE r4 (host/comm.cpp): under a static rendezvous two launches carry the same job token; the id file an earlier one left
    behind when it was killed is younger than the 600 s age limit - a rank that polls before rank 0 has replaced it must not take
    it (ncclCommInitRank would hang).  The file is accepted only if it was not written more than 30 s before the reading
    process started.  No GPU: the acceptance rules through their test hook."""
    import ctypes as C
    import pwicp_amd as P
    L = P.load_library()
    L.pwicp_comm_debug_id_file.argtypes = [C.c_char_p, C.c_int, C.c_long]
    L.pwicp_comm_debug_id_file.restype = C.c_int
    path = str(tmp_path / "rccl.id").encode()
    assert L.pwicp_comm_debug_id_file(path, 1, 0) == 0                # nothing there yet
    assert L.pwicp_comm_debug_id_file(path, 0, 0) == 1                # this launch's rank 0 writes now
    assert L.pwicp_comm_debug_id_file(path, 1, 0) == 1
    # the same token, 120 s old (< 600 s): this process (started with the test session, about as long ago as that or later)
    # must refuse a file dated before its own start minus the launcher's stagger
    import psutil, time
    since_start = time.time() - psutil.Process().create_time()
    assert L.pwicp_comm_debug_id_file(path, 0, int(since_start) + 60) == 1
    assert L.pwicp_comm_debug_id_file(path, 1, 0) == 0
    # ... unless the launcher is known to start its ranks far apart ($PWICP_ID_STAGGER_S), or the launch has a name of its own
    # ($PWICP_JOB_ID: the token IS unique then and the start-time rule is off) - ADVICE r5.  Same token needed: the reader with
    # $PWICP_JOB_ID must have written the file itself.
    rd = ("import ctypes as C, sys; sys.path.insert(0, %r); import pwicp_amd as P; L = P.load_library(); "
          "L.pwicp_comm_debug_id_file.argtypes = [C.c_char_p, C.c_int, C.c_long]; "
          % os.path.join(ROOT, "piecewise-icp_amd"))
    wide = dict(os.environ, PWICP_ID_STAGGER_S="100000")
    two = ("ok = L.pwicp_comm_debug_id_file(%r, 0, 400) == 1 and L.pwicp_comm_debug_id_file(%r, 1, 0) == %d; sys.exit(0 if ok else 1)")
    assert subprocess.run([sys.executable, "-c", rd + two % (path + b".w", path + b".w", 1)], env=wide).returncode == 0
    assert subprocess.run([sys.executable, "-c", rd + two % (path + b".d", path + b".d", 0)], env=dict(os.environ)).returncode == 0
    named = dict(os.environ, PWICP_JOB_ID="named-launch")
    assert subprocess.run([sys.executable, "-c", rd + two % (path + b".n", path + b".n", 1)], env=named).returncode == 0
    assert L.pwicp_comm_debug_id_file(path, 0, 700 + int(since_start)) == 1   # older than the age limit
    assert L.pwicp_comm_debug_id_file(path, 1, 0) == 0
    # another launch's token: not taken however fresh (a subprocess with another PWICP_JOB_ID writes, this process reads)
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import pwicp_amd as P; L = P.load_library(); "
            "L.pwicp_comm_debug_id_file.argtypes = [C.c_char_p, C.c_int, C.c_long]; "
            "sys.exit(0 if L.pwicp_comm_debug_id_file(%r, 0, 0) == 1 else 1)" % (os.path.join(ROOT, "piecewise-icp_amd"), path))
    env = dict(os.environ, PWICP_JOB_ID="another-launch")
    assert subprocess.run([sys.executable, "-c", code], env=env).returncode == 0
    assert L.pwicp_comm_debug_id_file(path, 1, 0) == 0
    assert L.pwicp_comm_debug_id_file(path, 0, 0) == 1 and L.pwicp_comm_debug_id_file(path, 1, 0) == 1


LABEL_WORKER = textwrap.dedent('''
    import os, sys, threading, time
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(%(root)r, "piecewise-icp_amd"))
    from pwicp_amd import fourd
    from pwicp_amd.series import run_pairs_sharing_target
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import torch
    dev = torch.device("cpu")

    class FakeSeries:
        """the label hand-over of a series without its GPU work: run_pairs 'segments' the target after a while on rank 0 and
        waits for the supplied labels on the others (what pwicp_series_run_pairs does, host/registration.cpp target_labels)"""
        def __init__(self, fail):
            self.cv = threading.Condition(); self.done = None; self.closed = False; self.supplied = "none"; self.expected = False
            self.fail = fail
        def expect_target_labels(self, scan): self.expected = True
        def supply_target_labels(self, scan, labels, nsv):
            with self.cv: self.supplied = (None if labels is None else (np.array(labels), nsv)); self.cv.notify_all()
        def wait_target_labels(self, scan, timeout_s=60.0):
            with self.cv:
                self.cv.wait_for(lambda: self.done is not None or self.closed, timeout_s)
                return self.done if self.done else None
        def close_target_labels(self):
            with self.cv: self.closed = True; self.cv.notify_all()
        def run_pairs(self, mine):
            recs = np.zeros(len(mine), fourd.RECORD)
            if self.expected:
                with self.cv: assert self.cv.wait_for(lambda: self.supplied != "none", 60.0)
                self.got = self.supplied
            else:
                time.sleep(0.2)
                if not self.fail:
                    with self.cv: self.done = (np.arange(1000, dtype=np.int32) %% 37, 37); self.cv.notify_all()
            recs["pair"] = mine
            return recs

    for fail in (False, True):
        s = FakeSeries(fail)
        mine = [p for p in range(3) if p %% world == rank]
        recs = run_pairs_sharing_target(s, mine, 0, rank, world, dist, dev)
        assert list(recs["pair"]) == mine
        if rank != 0 and mine:                                        # (a rank without pairs prepares no target: nothing to receive into)
            if fail:
                assert s.got is None                                   # rank 0 failed on the target: this rank segments for itself
            else:
                lab, nsv = s.got
                assert nsv == 37 and np.array_equal(lab, np.arange(1000, dtype=np.int32) %% 37)
        dist.barrier()
    # a rank without pairs of its own still takes part in the two broadcasts
    s = FakeSeries(False)
    recs = run_pairs_sharing_target(s, [0] if rank == 0 else [], 0, rank, world, dist, dev)
    assert len(recs) == (1 if rank == 0 else 0)
    dist.barrier()
    if rank == 0:
        print("LABELS_OK")
    dist.destroy_process_group()
''')


def test_shared_target_labels_travel_from_rank0_world2_gloo(tmp_path):
    """in a Direct2Ref series on several ranks the target is segmented once, by rank 0; the other ranks take
    its labels from a broadcast that runs beside their own preparation (pwicp_amd.series.run_pairs_sharing_target).  The
    collective part without the GPU work: labels arrive intact, a failure on rank 0 arrives as 'segment it yourself', a rank
    without pairs takes part.  (The GPU side - records byte-equal to one rank's, the front end run once - is
    tests/test_gpu_bench_ranks.py.)"""
    script = tmp_path / "labels.py"
    script.write_text(LABEL_WORKER % {"root": ROOT})
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "LABELS_OK" in res.stdout


def test_shared_target_labels_travel_from_rank0_world8_gloo(tmp_path):
    """The same hand-over with eight ranks (three pairs: five ranks have none and still take part in every collective)."""
    script = tmp_path / "labels8.py"
    script.write_text(LABEL_WORKER % {"root": ROOT})
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "LABELS_OK" in res.stdout


def test_target_label_exchange_entry_points_without_a_run(tmp_path):
    """The C side of the hand-over (include/pwicp.h: pwicp_series_*_target_labels) as far as it goes without a GPU: a waiter
    learns at once that a run has ended without the target (close), supplied labels are kept for the run that expects them,
    bad arguments are error codes."""
    import ctypes as C
    import pwicp_amd as P
    from pwicp_amd.pcd import write_pcd_binary as write_pcd
    inp = tmp_path / "scans"
    inp.mkdir()
    for e in range(1, 4):
        write_pcd(str(inp / ("Epoch_%03d.pcd" % e)), np.zeros((3, 3), np.float32))
    cfg = tmp_path / "cfg.txt"
    cfg.write_text("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                   "float PCres1 (m): 0.005\nfloat PCres2 (m): 0.005\nfloat SVsize1 (m): 0.05\nfloat SVsize2 (m): 0.05\n"
                   "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): 0.05\nfloat DTmin (m): 0.004\n"
                   "bool isVisual (yes-1, no-0): 0" % (str(inp), str(tmp_path) + "/res_"))
    with P.Series(str(cfg), 0, 3, 0, 0.75, 0) as s:
        assert s.target_label_counts() == (0, 0)
        s.expect_target_labels(0)
        s.supply_target_labels(0, np.arange(10, dtype=np.int32), 4)
        s.supply_target_labels(0, None, 0)
        with pytest.raises(P.PwicpError):
            s.expect_target_labels(99)
        import threading, time
        out = []
        th = threading.Thread(target=lambda: out.append(s.wait_target_labels(0, 30.0)))
        th.start()
        time.sleep(0.2)
        assert th.is_alive()                       # nothing has segmented the target yet
        s.close_target_labels()
        th.join(10.0)
        assert not th.is_alive() and out == [None]
