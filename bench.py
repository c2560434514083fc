#!/usr/bin/env python3
"""Benchmark of the Piecewise-ICP fine-registration loop on MI355X (BASELINE.json metric:
correspondences/s and ms per ICP iteration on a synthetic 1 M-point pair).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU under torch.distributed.run — the driver's launch, or, when no launcher environment is set,
   bench.py starts the N ranks itself; a world size other than N is an error; every rank registers its own source epoch of a
   synthetic 4D series against the shared reference epoch — independent pairs, weak scaling — and the 384-byte
   result records of all steps are all-gathered over RCCL once, inside the timed region: the series' one exchange.)

A "step" = one complete Piecewise-ICP loop (Piecewise_ICP's while-loop, reference src/Registration.cpp:680-694)
on data already resident in HBM: pwicp_pair_reset (device-to-device restore of the source arrays) +
pwicp_pair_run.  Patch generation (front end) and uploads are setup, outside the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
# The HIP runtime serves a process's streams from 4 hardware queues unless told otherwise; registrations side by side on as many
# contexts (the `pairs_side_by_side` figure; front ends of several clouds) want one each.  Read when the runtime starts: set before
# anything touches the GPU.  The timed steps use one stream and do not depend on it (0.259 ms either way).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
R_SPACING = 0.005


def cpu_budget():
    """CPUs this process may really use: the affinity mask, cut by the cgroup's CPU quota (v2 cpu.max / v1 cfs_quota_us).  A box
    that shows 256 hardware threads under a 16-CPU quota freezes the whole cgroup once its threads overrun the quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def kernel_source_hash():
    """sha256 over the HIP sources of libpwicp.so: ties profiles/traffic_latest.json (PMC passes, collected with
    tools/collect_profiles.sh) to the kernels that are being timed."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "piecewise-icp_amd", "csrc", "*"))):
        if os.path.basename(f) in ("frontend.hip", "prep.hip", "api.hip"):      # setup stages: not on the timed path
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def make_pair(n_points, epoch, ctx):
    """Synthetic reference tile + source epoch, reduced to the target centroid
    (Registration.cpp:277-294) and labelled."""
    from pwicp_amd import synth
    r = R_SPACING
    tgt, _ = synth.make_tile(n_points, r)
    src, Tgt = synth.make_source(n_points, r, epoch=epoch)
    c = tgt.mean(axis=0)
    tgt = (tgt - c).astype(np.float32)
    src = (src - c).astype(np.float32)
    t0 = time.time()
    l1, n1 = segment(tgt, 10 * r, ctx)
    l2, n2 = segment(src, 10 * r, ctx)
    global FRONTEND_S
    FRONTEND_S = time.time() - t0
    return tgt, l1, n1, src, l2, n2, Tgt


LABELS = "supervoxel"
FRONTEND_S = 0.0


def segment(cloud, sv, ctx):
    """Supervoxel labels from the product's own front end (csrc/frontend.hip: k-NN graph, fusion and refinement on the GPU, only the
    closed-form eigen step of the normals on host threads; setup, outside the timed hot path); `--labels grid`
    substitutes square grid cells."""
    from pwicp_amd import synth
    if LABELS == "grid":
        return synth.grid_labels(cloud, sv)
    return ctx.frontend_segment(cloud, sv, 45, R_SPACING)


def cpu_baseline(tgt, l1, n1, src, l2, n2, passes=25, passes_mt=10):
    """The CPU oracle (single-threaded C restatement of the reference path, KD-trees rebuilt and patch normals
    recomputed at the reference's call sites) timed on this host, same inputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    r = R_SPACING
    P1 = O.select_patches(tgt, l1, n1)
    P2 = O.select_patches(src, l2, n2)
    best = None
    O.set_num_threads(1)
    for _ in range(passes):
        io = O.run_loop(tgt, src, P1, P2, r, r, 10 * r, 10 * r, 10 * r, 0.8 * r, faithful=True)
        if best is None or io.t_loop_s < best.t_loop_s:
            best = io
    # SURVEY 8d's second CPU figure: the same path with its nearest-neighbour queries spread over all host cores
    # (OpenMP; tree builds and reductions stay serial, results identical)
    mt, cores = None, min(O.max_threads(), 64, cpu_budget())
    if cores > 1:
        O.set_num_threads(cores)
        for _ in range(passes_mt):
            io = O.run_loop(tgt, src, P1, P2, r, r, 10 * r, 10 * r, 10 * r, 0.8 * r, faithful=True)
            if mt is None or io.t_loop_s < mt.t_loop_s:
                mt = io
        O.set_num_threads(1)
    return best, mt, cores


def series_workload(args, ctx, P, rank, world):
    """BASELINE configs[4] shape on one GPU: the source epochs of a Direct2Ref series, one after the other, against ONE
    device-side target (pwicp_target).  Per epoch: host->device upload of the cloud and its labels, patch selection + statistics,
    query ordering, work buffers (pwicp_pair_create_with_target), then the registration loop.  Labels (front end) are setup.
    Prints its own JSON line (metric pairs/s); the loop-only rate of the pair workload is not affected by it."""
    from pwicp_amd import synth
    r, n = R_SPACING, args.points
    tgt, _ = synth.make_tile(n, r)
    c = tgt.mean(axis=0)
    tgt = (tgt - c).astype(np.float32)
    prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
    l1, n1 = segment(tgt, 10 * r, ctx)
    t0 = time.perf_counter()
    T = P.Target(ctx, tgt, l1, n1, prm.Res1, prm.SVRes1)
    t_target = time.perf_counter() - t0
    epochs = []
    for e in range(args.epochs):
        s, _ = synth.make_source(n, r, epoch=1 + e + rank * args.epochs)
        s = (s - c).astype(np.float32)
        l2, n2 = segment(s, 10 * r, ctx)
        epochs.append((P.f4(s), np.ascontiguousarray(l2, np.int32), n2))
    import threading
    import torch
    torch.cuda.synchronize()
    # plain host->device rate of one epoch's cloud (what PCIe gives; the scans sit in pageable memory)
    dbuf = torch.empty(epochs[0][0].shape, dtype=torch.float32, device="cuda")
    hsrc = torch.from_numpy(epochs[0][0])
    t_copy = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        ta = time.perf_counter()
        dbuf.copy_(hsrc, non_blocking=True)
        torch.cuda.synchronize()
        t_copy = min(t_copy, time.perf_counter() - ta)
    del dbuf
    # one epoch after the other on one context (create, loop, destroy) ...
    t_create = t_loop = t_destroy = 0.0
    n_corr = up_bytes = 0
    t0 = time.perf_counter()
    for s4, l2, n2 in epochs:
        ta = time.perf_counter()
        pair = P.Pair(ctx, None, None, 0, s4, l2, n2, prm, target=T)
        tb = time.perf_counter()
        res = pair.run()
        tc = time.perf_counter()
        pair.close()
        td = time.perf_counter()
        t_create += tb - ta
        t_loop += tc - tb
        t_destroy += td - tc
        n_corr += int(res.n_corr)
        up_bytes += s4.nbytes + l2.nbytes
    wall_serial = time.perf_counter() - t0
    # ... and pipelined over two contexts (= streams) and two host threads: epoch k + 1 is uploaded and its patches selected
    # (pwicp_pair_create_with_target_on, second context) while the loop of epoch k runs; the series is streamed twice so that
    # both contexts are warm in the timed pass
    ctx2 = P.Context(ctx.device)
    ctxs = (ctx, ctx2)

    def pipelined():
        nxt = {}

        def make(k):
            s4, l2, n2 = epochs[k]
            nxt[k] = P.Pair(ctxs[k % 2], None, None, 0, s4, l2, n2, prm, target=T)

        make(0)
        ok = True
        for k in range(len(epochs)):
            th = None
            if k + 1 < len(epochs):
                th = threading.Thread(target=make, args=(k + 1,))
                th.start()
            pair = nxt.pop(k)
            ok = ok and pair.run().status == 0
            pair.close()
            if th:
                th.join()
        return ok

    pipelined()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok_pipe = pipelined()
    wall = time.perf_counter() - t0
    ctx2.close()
    if rank == 0:
        print(json.dumps({
            "metric": "pairs/sec (streamed Direct2Ref series, secondary line)", "value": round(args.epochs / wall, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": args.epochs, "warmup": 0, "ms_per_step": round(1e3 * wall / args.epochs, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Direct2Ref series, %d source epochs x %d points against one shared device-side target, one GPU, "
                                   "epochs streamed through two contexts: upload + patch selection of epoch k + 1 beside the loop of "
                                   "epoch k (BASELINE configs[4] shape)" % (args.epochs, n),
                       "points_per_cloud": n, "epochs": args.epochs},
            "all_pairs_ok": bool(ok_pipe),
            "ms_per_pair_one_after_the_other": {"upload_select_grids": round(1e3 * t_create / args.epochs, 3),
                                                "loop": round(1e3 * t_loop / args.epochs, 3),
                                                "destroy": round(1e3 * t_destroy / args.epochs, 3),
                                                "total": round(1e3 * wall_serial / args.epochs, 3),
                                                "shared_target_once": round(1e3 * t_target, 3)},
            "host_to_device": {"bytes_per_pair": up_bytes // args.epochs,
                               "copy_gbs": round(epochs[0][0].nbytes / t_copy / 1e9, 1),
                               "note": "copy_gbs: a plain copy of one epoch's cloud from pageable host memory; the rest of "
                                       "pwicp_pair_create_with_target is patch selection, query ordering, grids and buffers"},
            "correspondences_per_s_incl_setup": round(n_corr / wall, 1)}))
    T.close()
    ctx.close()


def concurrent_pairs(args, P, local_rank, data, prm, expect_T16, in_flight=4):
    """A SECOND figure beside `value`, never mixed into it: the same registration as the timed steps, `in_flight` of them side by
    side on as many contexts (= HIP streams) from as many host threads - independent pairs of a 4D series (R.cpp:89-150) need
    not wait for each other, and one registration is a chain of ~14 short launches that leaves most of the chip idle."""
    import threading
    tgt, l1, n1, src, l2, n2 = data
    reps = max(3 * args.steps, 30)
    ctxs = [P.Context(local_rank) for _ in range(in_flight)]
    pairs = [P.Pair(c, tgt, l1, n1, src, l2, n2, prm) for c in ctxs]
    last = [None] * in_flight

    def loop(i, n):
        for _ in range(n):
            pairs[i].reset()
            last[i] = pairs[i].run()

    def side_by_side(n):
        th = [threading.Thread(target=loop, args=(i, n)) for i in range(in_flight)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return time.perf_counter() - t0
    side_by_side(5)                                  # warm-up
    dt = side_by_side(reps)
    same = all(list(r.T16) == list(expect_T16) for r in last)
    corr = float(sum(r.n_corr for r in last)) * reps
    for pr in pairs: pr.close()
    for c in ctxs: c.close()
    return {"in_flight": in_flight, "registrations": reps * in_flight, "ms_per_registration": round(1e3 * dt / (reps * in_flight), 4),
            "value": round(corr / dt, 1), "unit": "correspondences/s", "results_identical_to_the_timed_steps": bool(same),
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "note": "throughput of independent pairs on ONE GPU (one context, stream and host thread per pair in flight, "
                    "GPU_MAX_HW_QUEUES hardware queues: 0.16 ms with the runtime's default 4, 0.13 with 8); `value` and "
                    "ms_per_step above are one pair at a time"}


def series_end_to_end(args, P, rank, world, local_rank, dist, dev, barrier):
    """BASELINE configs[3] end to end, beside the loop-only figure: ONE Direct2Ref series of `--series-epochs` source epochs of
    `--points` points (PCD files written by rank 0), its pairs dealt p -> rank p mod world, every rank going from the files through
    preprocessing, front end, registration (pwicp_series_run_pairs); then the series' one exchange (all-gather of the 384-byte
    records).  Strong scaling of a fixed series: what contends on a node - scan I/O, host threads of the front ends, PCIe uploads
    from pageable memory - shows here and not in the loop-only line.  Returns a dict for the main JSON line (rank 0), else None."""
    import shutil
    import tempfile
    from pwicp_amd import fourd, synth
    from pwicp_amd.pcd import write_pcd_binary
    r, n, E = R_SPACING, args.points, args.series_epochs
    root = os.environ.get("PWICP_BENCH_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    d = os.path.join(root, "pwicp_bench_series_%s" % os.environ.get("MASTER_PORT", str(os.getppid())))
    inp = os.path.join(d, "scans")
    cfg = os.path.join(d, "cfg.txt")
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(inp)
        t, _ = synth.make_tile(n, r)
        write_pcd_binary(os.path.join(inp, "Epoch_001.pcd"), t.astype(np.float32))
        for e in range(1, E + 1):
            s, _ = synth.make_source(n, r, epoch=e)
            write_pcd_binary(os.path.join(inp, "Epoch_%03d.pcd" % (e + 1)), s.astype(np.float32))
        with open(cfg, "w") as f:
            f.write("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                    "float PCres1 (m): %g\nfloat PCres2 (m): %g\nfloat SVsize1 (m): %g\nfloat SVsize2 (m): %g\n"
                    "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): %g\nfloat DTmin (m): %g\nbool isVisual (yes-1, no-0): 0"
                    % (inp, os.path.join(d, "out_"), r, r, 10 * r, 10 * r, 10 * r, 0.8 * r))
    barrier()
    out = None
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    try:
        os.dup2(devnull, 1)                      # the entry point prints the reference's progress lines on stdout
        mine = [p for p in range(E) if p % world == rank]

        def one_series():
            """the whole series through a fresh Series object (nothing prepared: the shared target too is read, preprocessed and
            segmented again), then the series' one exchange"""
            series = P.Series(cfg, 0, E + 1, 0, 0.75, local_rank)
            barrier()
            t0 = time.perf_counter()
            from pwicp_amd.series import run_pairs_sharing_target
            recs = run_pairs_sharing_target(series, mine, 0, rank, world, dist, dev)       # (the target is segmented once, by rank 0)
            table = fourd.gather_records([recs[k:k + 1] for k in range(len(recs))], E, world, dist=dist, device=dev)
            assert len(table) == E, "record gather incomplete"
            barrier()
            wall = time.perf_counter() - t0
            stages = series.stage_times()
            stages["target_labels_received"], stages["target_labels_segmented"] = series.target_label_counts()
            series.close()                       # (its contexts and front-end work spaces stay parked for the next series)
            return wall, recs, table, stages
        # first series of the process: device contexts, ~2 GB of front-end work space per stream, pinned staging buffers and kernel
        # code are set up on the way (reported as cold_wall_s) - the warm-up of this figure; the timed one finds them parked
        wall_cold = one_series()[0]
        warm = [one_series() for _ in range(3)]  # three warm series: the figure is their median (one alone scatters by +-5 %)
    finally:
        import ctypes
        ctypes.CDLL(None).fflush(None)           # what the library has buffered for stdout goes to /dev/null too
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    import torch
    walls, tcold = [w[0] for w in warm], wall_cold
    if dist is not None:
        t = torch.tensor(walls + [wall_cold], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        walls, tcold = [float(x) for x in t[:-1].tolist()], float(t[-1].item())
    mid = sorted(range(len(walls)), key=lambda i: walls[i])[len(walls) // 2]      # (the same series on every rank: reduced walls)
    tmax = walls[mid]
    _, recs, table, stages = warm[mid]
    ok = all(bool(np.all(w[1]["status"] == 0)) if len(w[1]) else True for w in warm)
    # who segmented the shared target: [received from rank 0, segmented here] of every rank (Direct2Ref: one target)
    labels_by_rank = [[stages["target_labels_received"], stages["target_labels_segmented"]]]
    if dist is not None:
        t = torch.tensor(labels_by_rank[0], dtype=torch.int32, device=dev)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        labels_by_rank = [[int(x) for x in p_.tolist()] for p_ in parts]
    # the stage walls of EVERY rank (the series ends with its slowest rank: that one's stages are what the wall time is made of)
    stages_by_rank = [dict(stages, rank=rank)]
    if dist is not None:
        stages_by_rank = [None] * world
        dist.all_gather_object(stages_by_rank, dict(stages, rank=rank))
    if rank == 0:
        def _busy(st):
            return sum(v for k, v in st.items() if k.endswith("_ms"))
        slowest = max(stages_by_rank, key=_busy)
    if rank == 0:
        out = {"metric": "pairs/sec of a WARM process, PCD files -> transforms (Direct2Ref series of %d source epochs x %d pts, pairs dealt over "
                         "the GPUs; cold_value: the first series of the process, what a fresh reference process or rounds 1 - 3 are comparable with)" % (E, n),
               "value": round(E / tmax, 3), "cold_value": round(E / tcold, 3), "unit": "pairs/s", "scaling": "strong", "n_gpus": world,
               "pairs": E, "wall_s": round(tmax, 3), "cold_wall_s": round(tcold, 3), "warm_walls_s": [round(x, 3) for x in walls], "all_pairs_ok": ok,
               "rank0_stage_wall_ms": {k: round(v, 1) for k, v in stages.items() if k.endswith("_ms")},
               "rank0_scan_bytes_to_gpu": stages["scan_bytes"],
               "frontend_host_takeovers_rank0": P.frontend_fallback_counts(),
               "slowest_rank": int(slowest["rank"]),
               "slowest_rank_stage_wall_ms": {k: round(v, 1) for k, v in slowest.items() if k.endswith("_ms")},
               "stage_wall_ms_by_rank": [{k: (round(v, 1) if k.endswith("_ms") else v) for k, v in st.items() if k.endswith("_ms") or k == "rank"}
                                         for st in stages_by_rank],
               "target_labels_by_rank": labels_by_rank,
               "target_labels_note": "[taken from rank 0's broadcast, made by the rank's own front end] per rank: the shared target of "
                                     "the series is segmented once, by rank 0 (pwicp_amd.series.run_pairs_sharing_target)",
               "note": "stages of rank 0 (its share of the pairs + the shared target): reading scans, GPU preparation (voxel grid incl. the "
                       "host-side std::sort order, SOR, reduction), what is left of the front ends after that, registrations; the "
                       "front end of a cloud (~65 ms per 1 M points alone, ~45 ms with several side by side) is what a pair costs, the "
                       "loop is 0.25 ms of it; a pair is registered as soon as its source is segmented, beside the later front ends.  "
                       "wall_s: the median of three warm series of the process (warm_walls_s; each through a fresh Series object, "
                       "nothing prepared - the device contexts and work spaces of the first one are reused); cold_wall_s: the first one"}
        shutil.rmtree(d, ignore_errors=True)
        if args.dump_records:
            # the gathered table of the series, pair order (tests compare N ranks against one rank: tests/test_gpu_configs.py);
            # the two timing fields are not results
            tab = np.concatenate([table[p].reshape(1) for p in range(E)])
            tab["t_loop_ms"] = 0
            tab["t_pair_ms"] = 0
            np.save(args.dump_records, tab)
    return out


def frontend_workload(args, ctx, P, rank, world):
    """the segmentation front end of ONE cloud (S.cpp:18-68: k-NN-45 graph, PCA normals, supervoxel fusion,
    boundary refinement) through pwicp_frontend_segment_dev, timed host buffer in -> labels out.  The device pipeline
    (csrc/frontend.hip) against the serial host passes ($PWICP_FRONTEND=host, the restatement the labels are checked
    against); the labels of both must be identical.  Prints its own JSON line (metric points/s)."""
    from pwicp_amd import synth
    r, n = R_SPACING, args.points
    cloud, _ = synth.make_tile(n, r, offset=(float(rank), 0.0, 0.0))
    cloud = (cloud - cloud.mean(axis=0)).astype(np.float32)
    ctx.frontend_segment(cloud[: max(n // 10, 2000)], 10 * r, 45, r)                     # warm-up (allocations, code objects)
    times = []
    for _ in range(max(args.steps, 1)):
        t0 = time.perf_counter()
        lab_d, nsv_d = ctx.frontend_segment(cloud, 10 * r, 45, r)
        times.append(time.perf_counter() - t0)
    os.environ["PWICP_FRONTEND"] = "host"
    try:
        t0 = time.perf_counter()
        lab_h, nsv_h = ctx.frontend_segment(cloud, 10 * r, 45, r)
        t_host = time.perf_counter() - t0
    finally:
        os.environ.pop("PWICP_FRONTEND", None)
    t = float(np.median(times))
    # the reference's OWN front end (codelibrary k-d tree, PCA normals, supervoxel segmentation compiled into oracle/_ref) on a
    # bounded sample of the same cloud: its first 300 k points (a contiguous strip of the tile), single-threaded as the reference runs it
    ref = None
    if rank == 0 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        if O.ref_frontend_available():
            m = min(n, 300000)
            sample = np.ascontiguousarray(cloud[:m])
            t0 = time.perf_counter()
            lab_r, nsv_r = O.ref_frontend(sample, 10 * r)
            t_ref = time.perf_counter() - t0
            lab_s, nsv_s = ctx.frontend_segment(sample, 10 * r, 45, r)
            ref = {"value": round(m / t_ref, 1), "unit": "points/s", "cores": 1, "kind": "reference",
                   "sample": "the first %d points of the same cloud through the reference's own codelibrary front end (oracle/_ref, k-d tree "
                             "k-NN included), %.2f s" % (m, t_ref),
                   "device_labels_identical_on_sample": bool(nsv_r == nsv_s and np.array_equal(lab_r, lab_s))}
    if rank == 0:
        print(json.dumps({
            "metric": "points/sec segmented (front end of one cloud, secondary line)", "value": round(n / t, 1), "unit": "points/s",
            "n_gpus": world, "steps": len(times), "warmup": 1, "ms_per_step": round(1e3 * t, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "supervoxel labels of one synthetic %d-pt cloud: k-NN-45 graph + PCA normals + fusion + boundary "
                                   "refinement, host buffer in -> labels out" % n, "points_per_cloud": n, "supervoxels": int(nsv_d)},
            "labels_identical_to_serial_passes": bool(nsv_d == nsv_h and np.array_equal(lab_d, lab_h)),
            "host_takeovers_in_this_process": P.frontend_fallback_counts(),
            "cpu_baseline": {"value": round(n / t_host, 1), "unit": "points/s", "kind": "port",
                             "sample": "the same cloud through the serial host passes (host/frontend.cpp; k-NN graph still on the "
                                       "device), host threads only for normals / lambda0 / seeds", "ms": round(1e3 * t_host, 3)},
            "cpu_reference": ref}))
    ctx.close()


def large_roofline(args, P, ctx):
    """A second roofline record of the dense 1-NN launch on a `--large-points` pair (BASELINE configs[4]'s point count is 5 M; 4 M by
    default): at 1 M points the launch is two generations of blocks and its ramp and drain are a third of it.  Same measurement as the
    main record: HIP events around the launch inside the loop, on untimed steps.  Physical traffic only if profiles/traffic_latest.json
    holds it for these kernel sources and this size."""
    n = args.large_points
    global FRONTEND_S
    keep = FRONTEND_S                      # (make_pair times the front ends of ITS clouds into the global the main line reports)
    tgt, l1, n1, src, l2, n2, _ = make_pair(n, epoch=1, ctx=ctx)
    FRONTEND_S = keep
    r = R_SPACING
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r))
    try:
        pair.set_profiling(0)
        walls = []
        for _ in range(2):
            pair.reset(); pair.run()
        for _ in range(5):
            pair.reset()
            t0 = time.perf_counter(); pair.run(); walls.append(time.perf_counter() - t0)
        pair.set_profiling(1)
        prof = []
        for _ in range(max(args.roofline_steps, 1)):
            pair.reset()
            prof.append(pair.run())
        n_launch = sum(rr.n_dense_nn_launches for rr in prof)
        t_ms = sum(rr.t_dense_nn_ms for rr in prof)
        if not n_launch or t_ms <= 0:
            return None
        nq = sum(rr.n_corr_dense for rr in prof) / n_launch
        dur_s = t_ms / n_launch * 1e-3
        kbar = prof[-1].dense_kbar
        compulsory = (16 + 4) * nq + 12.0 * len(tgt)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            if tj.get("kernel_source_sha256") == kernel_source_hash() and tj.get("large_points") == n:
                traffic = tj.get("k_nn_dense_bytes_per_launch_large")
        except Exception:
            traffic = None
        b = traffic if traffic else compulsory
        return {"bound": "hbm", "kernel": "k_nn_dense_disc", "points_per_cloud": n, "achieved": round(b / dur_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(b / dur_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_is": ("PMC FETCH_SIZE x correction + WRITE_SIZE per launch" if traffic else
                               "not measured for these sources / this size: achieved uses the compulsory bytes (lower bound)"),
                "frac_useful": round(compulsory / dur_s / 1e9 / HBM_PEAK_GBS, 4), "compulsory_bytes_per_launch": int(compulsory),
                "traffic_over_compulsory": (round(traffic / compulsory, 2) if traffic else None),
                "queries_per_launch": int(nq), "kbar": round(kbar, 2), "avg_launch_us": round(dur_s * 1e6, 2),
                "ms_per_step": round(1e3 * sorted(walls)[len(walls) // 2], 4), "outer_iterations": int(prof[-1].n_outer),
                "correspondences_per_step": int(prof[-1].n_corr),
                "note": "same pair generator and parameters as the main line at %d points per cloud; not part of `value`" % n}
    finally:
        pair.close()


def launch_ranks(n):
    """`python bench.py --gpus N` started by hand (no launcher environment): start the N ranks the way the driver does —
    torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 and a free port — and pass their exit code on."""
    import socket
    import subprocess
    import uuid
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("PWICP_JOB_ID", uuid.uuid4().hex)          # the token of the library's own RCCL rendezvous (host/comm.cpp)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, cpu_budget() // n)))
    # the library's host-thread pool (host/parallel.h) divides the CPUs it may use by $LOCAL_WORLD_SIZE by itself; said here too
    env.setdefault("PWICP_HOST_THREADS", str(max(1, cpu_budget() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--series-epochs", type=int, default=8,
                    help="source epochs of the end-to-end series measured beside the loop-only figure (BASELINE configs[3]); 0: skip")
    ap.add_argument("--pairs-in-flight", type=int, default=4,
                    help="a second, separate figure at N=1: this many registrations side by side on as many contexts; <= 1: skip")
    ap.add_argument("--no-inner-timing", action="store_true",
                    help="skip the two extra untimed steps that time the inner ICP with HIP events (kernel-trace runs: the last "
                         "step of the process is then a step as timed)")
    ap.add_argument("--roofline-steps", type=int, default=5,
                    help="extra untimed steps with HIP events around the dense 1-NN launch (the roofline record's duration)")
    ap.add_argument("--large-points", type=int, default=4000000,
                    help="a SECOND roofline record (roofline_large) of the dense 1-NN launch on a pair of this many points per cloud, where "
                         "the launch's fixed part is not the whole story (N=1 only); 0: skip")
    ap.add_argument("--labels", choices=["supervoxel", "grid"], default="supervoxel")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for debugging)")
    ap.add_argument("--workload", choices=["pair", "series", "frontend"], default="pair",
                    help="pair (default): the BASELINE metric line.  series: a SECOND kind of line — a Direct2Ref 4D series streamed "
                         "through one GPU (shared device-side target, one source epoch after the other: upload, patch selection, grids, "
                         "loop), per-pair wall time and the host->device rate; never mixed into the pair line's `value`")
    ap.add_argument("--epochs", type=int, default=4, help="source epochs of --workload series (BASELINE configs[4]: 4 per GPU at 5 M points)")
    ap.add_argument("--dump-records", default=None, metavar="FILE.npy",
                    help="rank 0 saves the gathered 384-byte records of the end-to-end series (pair order) for comparison across N")
    ap.add_argument("--single-device", action="store_true",
                    help="debug: every rank uses GPU 0 (functional check of the N>1 path on a 1-GPU box; needs --backend gloo)")
    args = ap.parse_args()
    global LABELS
    LABELS = args.labels

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # never print an n_gpus other than the one asked for
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or let bench.py --gpus N launch the ranks itself)"
                 % (args.gpus, world, args.gpus))
    import torch
    import pwicp_amd as P
    from pwicp_amd import fourd
    if world > 1 and not args.single_device and torch.cuda.device_count() < world:
        sys.exit("bench.py: --gpus %d but only %d device(s) visible (--single-device --backend gloo shares one GPU for a functional check)"
                 % (world, torch.cuda.device_count()))

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend=args.backend)
        dist = dist_mod
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if args.backend == "nccl" or world == 1 else torch.device("cpu")

    ctx = P.Context(local_rank)
    if args.workload == "series":
        return series_workload(args, ctx, P, rank, world)
    if args.workload == "frontend":
        return frontend_workload(args, ctx, P, rank, world)
    # ---- setup (untimed): data, labels, upload, patch selection/statistics, grids ------------------------
    tgt, l1, n1, src, l2, n2, Tgt = make_pair(args.points, epoch=rank + 1, ctx=ctx)
    r = R_SPACING
    prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
    t0 = time.time()
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, prm)
    t_setup = time.time() - t0
    pair.set_profiling(0)        # the timed steps carry NO event records (round 6; a record is a ~5 us bubble on the stream): the
                                 # roofline kernel's duration comes from extra, untimed steps behind the timed region

    def barrier():
        if dist is not None:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # A step is one pair of the 4D series on this rank (pair index = step * world + rank).  The pairs are independent
    # (R.cpp:89-187); the series' ONE exchange is the all-gather of the 384-byte result records once every rank has
    # run its pairs (R.cpp:197-203 composes them afterwards), so it happens once per timed region, inside it.
    def step(k, recs):
        pair.reset()
        res = pair.run()
        if dist is not None:
            recs.append(fourd.pack_record(k * world + rank, res.status, res.n_outer, int(res.n_inner_total), res.T16, res.VCM,
                                          res.n_corr))
        return res

    def exchange(recs):
        if dist is not None and recs:
            table = fourd.gather_records(recs, len(recs) * world, world, dist=dist, device=dev)
            assert len(table) == len(recs) * world, "record gather incomplete"

    wrecs = []
    for k in range(args.warmup):
        step(k, wrecs)
    exchange(wrecs)
    barrier()
    t0 = time.perf_counter()
    results, recs = [], []
    for k in range(args.steps):
        results.append(step(k, recs))
    exchange(recs)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = elapsed
    bounded_local = float(sum(rr.n_dense_bounded for rr in results))
    corr_ref_local = float(sum(rr.n_corr for rr in results))
    corr_local = corr_ref_local - bounded_local          # completed 1-NN queries only
    corr_total, corr_ref_total, bounded_total = corr_local, corr_ref_local, bounded_local
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        c = torch.tensor([corr_local, corr_ref_local, bounded_local], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        corr_total, corr_ref_total, bounded_total = [float(x) for x in c.tolist()]

    res = results[-1]
    # who took part (the first multi-GPU record must be readable on its own): every rank's device as the runtime sees it, and the size
    # of the communicator the exchange above went through (backend nccl = RCCL)
    me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "device": int(torch.cuda.current_device())}
    try:
        pr_ = torch.cuda.get_device_properties(me["device"])
        me.update({"name": pr_.name, "uuid": str(getattr(pr_, "uuid", "")), "pci_bus_id": getattr(pr_, "pci_bus_id", None),
                   "visible_devices": torch.cuda.device_count()})
    except Exception:
        pass
    ranks_seen = [me]
    comm_info = {"backend": (args.backend if dist is not None else None), "world_size": (dist.get_world_size() if dist is not None else 1)}
    if dist is not None:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
        if args.backend == "nccl":
            try:
                comm_info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                pass
        comm_info["distinct_devices"] = len({(r_.get("uuid") or r_.get("pci_bus_id") or r_["device"]) for r_ in ranks_seen})
        comm_info["records_gathered_per_timed_region"] = len(recs) * world
    side_by_side = None
    if world == 1 and args.pairs_in_flight > 1:
        try:
            side_by_side = concurrent_pairs(args, P, local_rank, (tgt, l1, n1, src, l2, n2), prm, res.T16, args.pairs_in_flight)
        except Exception as e:                 # a secondary figure must not take the metric line with it
            side_by_side = {"error": "%s: %s" % (type(e).__name__, e)}
    series_line = None
    if args.series_epochs > 0:
        if world == 1:
            try:
                series_line = series_end_to_end(args, P, rank, world, local_rank, dist, dev, barrier)
            except Exception as e:             # (a secondary figure: the metric line survives it; with several ranks a failure
                series_line = {"error": "%s: %s" % (type(e).__name__, e)}      # must stay loud - the others would wait for this one)
        else:
            series_line = series_end_to_end(args, P, rank, world, local_rank, dist, dev, barrier)
    # inner-iteration timing needs two more HIP events per ICP call (a ~6 us stream bubble each): measured on two
    # extra, untimed steps
    roofline_large = None
    if world == 1 and args.large_points > 0:
        try:
            roofline_large = large_roofline(args, P, ctx)
        except Exception as e:                 # a secondary record must not take the metric line with it
            roofline_large = {"error": "%s: %s" % (type(e).__name__, e)}
    pair.set_profiling(1 | 2)
    t_inner_ms = n_inner_prof = 0
    for _ in range(0 if args.no_inner_timing else 2):
        pair.reset()
        rp = pair.run()
        t_inner_ms += rp.t_inner_ms
        n_inner_prof += int(rp.n_inner_total)
    # ---- roofline of the dominant kernel (dense 1-NN, k_nn_dense_disc): HIP events around its launches inside the loop, on
    # `--roofline-steps` extra UNTIMED steps of the same pair (same launches, same stream; the timed steps stay free of events)
    pair.set_profiling(1)
    prof = []
    for _ in range(max(args.roofline_steps, 1)):
        pair.reset()
        prof.append(pair.run())
    pair.set_profiling(0)
    n_launch = sum(rr.n_dense_nn_launches for rr in prof)
    t_dense_ms = sum(rr.t_dense_nn_ms for rr in prof)
    roofline = None
    if n_launch > 0 and t_dense_ms > 0:
        nq = sum(rr.n_corr_dense for rr in prof) / n_launch
        kbar = prof[-1].dense_kbar
        dur_s = (t_dense_ms / n_launch) * 1e-3
        # SURVEY 8d's ALGORITHMIC stream (every query re-reads its candidates: query 16 + d2 out 4 + ~5 cell rows x (begin,
        # end) 8 + 16 per target point examined, Kbar measured by the kernel): mostly L1/L2 hits -> reported as model_gbs only
        b_nn = 16.0 + 4.0 + 5 * 8.0 + 16.0 * kbar
        model_gbs = b_nn * nq / dur_s / 1e9
        # PHYSICAL bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md
        # prescribes; profiles/traffic_latest.json) — only if they were collected on exactly these kernel sources
        traffic, pmc, stale, tj = None, {}, None, {}
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        sha = kernel_source_hash()
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("kernel_source_sha256") == sha:
                    traffic = tj.get("k_nn_dense_bytes_per_launch")
                    pmc = tj.get("dense_sq_counters", {})
                else:
                    stale = "profiles/traffic_latest.json was collected on other kernel sources: traffic not reported"
                    tj = {}
            except Exception:
                traffic, tj = None, {}
        # compulsory: every query in (16 B), its d2 out (4 B), the target once (the search reads its packed 12-byte copy)
        compulsory = (16 + 4) * nq + 12.0 * len(tgt)
        hbm_bytes = traffic if traffic else compulsory
        achieved = hbm_bytes / dur_s / 1e9
        # vector-ALU issue: wave instructions x 4 cycles (measured: SQ_ACTIVE_INST_ANY / instructions) over the chip's
        # 1024 SIMDs at 2.4 GHz
        valu = pmc.get("SQ_INSTS_VALU")
        roofline = {"bound": "hbm", "kernel": "k_nn_dense_disc" if res.dense_rows == 0 else "k_nn_dense_direct",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_is": ("PMC FETCH_SIZE x correction + WRITE_SIZE per launch" if traffic else
                                                       "not measured for these sources: achieved uses the compulsory bytes (lower bound)"),
                    "model_gbs": round(model_gbs, 1), "bytes_per_correspondence_model": round(b_nn, 1), "kbar": round(kbar, 2),
                    "queries_per_launch": int(nq), "avg_launch_us": round(dur_s * 1e6, 2),
                    "compulsory_bytes_per_launch": int(compulsory),
                    "valu_issue_frac": (round(valu * 4.0 / (1024 * 2.4e9 * dur_s), 3) if valu else None),
                    "traffic_over_compulsory": (round(traffic / compulsory, 2) if traffic else None),
                    # the same fraction on the USEFUL bytes only (queries in, d2 out, the target once): what an ideal kernel would move
                    "frac_useful": round(compulsory / dur_s / 1e9 / HBM_PEAK_GBS, 4),
                    # SURVEY 8d's formula taken literally (B_nn x queries / duration / peak): a CACHE-stream figure - every query
                    # re-reads its candidates from L1 / L2 - that exceeds 1 and is no HBM fraction; printed so nobody has to recompute it
                    "frac_survey8d": round(model_gbs / HBM_PEAK_GBS, 3),
                    "frac_survey8d_is": "L1/L2 stream incl. cache hits (not HBM traffic; earns no roofline credit)",
                    "duration_source": "HIP events around the launch inside the loop, %d launches on %d extra untimed steps" % (n_launch, len(prof)),
                    "traffic_source": ({"file": "profiles/traffic_latest.json", "kernel_source_sha256": sha,
                                        "collected_by": tj.get("source"), "fetch_correction_factor": tj.get("fetch_correction_factor")}
                                       if traffic else {"file": None, "kernel_source_sha256_now": sha}),
                    "fetch_correction": "FETCH_SIZE x 1.974 (k_transform_all's 16-B stream); holds for this kernel's 12-B gathers: the "
                                        "L2 fetches 128-B lines whatever the load width and the counter tallies 64 B per request "
                                        "(profiles/r05_gather_calibration.txt)",
                    "note": "achieved/frac = PHYSICAL fabric bytes per launch / HIP-event time / 8 TB/s; frac_useful = the compulsory bytes "
                            "over the same time; model_gbs = SURVEY 8d's algorithmic stream (cache hits included) for reference.  The 27 MB "
                            "working set of the pair stays in the 256 MiB Infinity Cache across the timed steps (the counters include its "
                            "hits), so 'HBM' here is fabric traffic.  What bounds the launch is the CU's address / tag pipe and L1 "
                            "(every query gathers ~27 candidates of 12 B: 0.5 GB per launch through the L1s) together with the vector "
                            "ALU, not HBM: profiles/r06_dense_variants.txt, DESIGN.md 4.1"}
        if stale:
            roofline["stale_profile"] = stale

    if rank == 0:
        value = corr_total / tmax
        n_outer = res.n_outer
        n_inner = int(res.n_inner_total)
        out = {
            "metric": "correspondences/sec", "value": round(value, 1), "unit": "correspondences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * tmax / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic %d-pt source/target pair per GPU, full Piecewise-ICP loop to convergence "
                                   "(BASELINE configs[1]; N>1: every step is one independent pair per GPU of a 4D series, the "
                                   "384-byte result records of all steps are all-gathered over RCCL once, inside the timed region)" % args.points,
                       "points_per_cloud": args.points, "spacing_m": r, "patches_target_source": list(pair.num_patches()),
                       "outer_iterations": n_outer, "inner_iterations": n_inner,
                       "correspondences_per_step": int(res.n_corr), "parallelism": "pair-per-gpu x%d" % world,
                       "other_configs": "BASELINE configs[0] (the reference's own scans), [2] and [3] / [4] are parity-test cases "
                                        "(tests/test_gpu_configs.py); configs[2], the rockfall pair, is NOT in the reference's tree - wherever "
                                        "this repo says cfg 3 it means a synthetic rockfall-scale stand-in, checked against the oracle only",
                       "segmentation": ("boundary-preserving supervoxels, product front end on the device (csrc/frontend.hip; setup, untimed)"
                                        if args.labels == "supervoxel" else "grid cells (setup, untimed)")},
            "ms_per_outer_iteration": round(res.t_loop_ms / max(n_outer, 1), 4),
            "ms_per_inner_iteration": round(t_inner_ms / max(n_inner_prof, 1), 4),
            # setup stages, reported separately and never part of `value`: supervoxel labels of both clouds
            # (one after the other here), then upload + patch selection + grids
            "frontend_s": round(FRONTEND_S, 3), "setup_s": round(t_setup, 3),
            "roofline": roofline,
            "roofline_large": roofline_large,
            # honest accounting of the metric's unit (one correspondence = one COMPLETED 1-NN query): far queries of a dense
            # search whose distance was proved to lie above the percentile are cut short (csrc/grid.hip k_nn_dense_far; C.cpp:266-281
            # returns Dist75 only).  `value` counts completed queries only; value_reference_equivalent counts every query the reference
            # issues (what rounds 1 - 5 printed as `value`)
            "queries_bounded_not_completed": int(bounded_total),
            "value_reference_equivalent": round(corr_ref_total / tmax, 1),
            "series_end_to_end": series_line,
            "pairs_side_by_side": side_by_side,
            "communicator": comm_info, "ranks": ranks_seen,
        }
        if world == 1 and not args.no_cpu_baseline:
            io, io_mt, mt_cores = cpu_baseline(tgt, l1, n1, src, l2, n2)
            cpu_val = io.n_corr / io.t_loop_s
            same = (io.n_outer == res.n_outer and
                    np.abs(np.array(io.T16, dtype=np.float64) - np.array(res.T16, dtype=np.float64)).max() < 1e-5)
            out["cpu_baseline"] = {"value": round(cpu_val, 1), "unit": "correspondences/s", "cores": 1, "kind": "port",
                                   "sample": "the full %d-pt pair loop, best of 25 passes (~12 s of CPU work), %.2f s per pass; single-threaded "
                                             "C oracle with KD-trees rebuilt at the reference's call sites" %
                                             (args.points, io.t_loop_s),
                                   "host_cores_available": os.cpu_count(), "host_cpus_usable": cpu_budget(),
                                   "gpu_matches_cpu_transform": bool(same)}
            out["speedup_vs_cpu"] = round(value / cpu_val, 1)
            if io_mt is not None:
                out["cpu_baseline_all_cores"] = {"value": round(io_mt.n_corr / io_mt.t_loop_s, 1), "unit": "correspondences/s",
                                                 "cores": mt_cores, "kind": "port",
                                                 "sample": "same loop, nearest-neighbour queries over all host cores (OpenMP), "
                                                           "best of 10 passes, %.3f s per pass" % io_mt.t_loop_s,
                                                 "same_result_as_single_thread": bool(
                                                     io_mt.n_outer == io.n_outer and list(io_mt.T16) == list(io.T16))}
        print(json.dumps(out))
    pair.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
