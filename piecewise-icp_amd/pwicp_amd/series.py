"""4D series on a multi-GPU node: PiecewiseICP_4D_call (R.cpp:17-215) with its pair loop (R.cpp:89-187) sharded over
one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        -m pwicp_amd.series configuration_4d.txt 0 20 0 0.75

(without a launcher: one GPU, same as the exported C entry point).  Every rank opens the same configuration, runs the
pairs p with p mod world == rank through libpwicp.so, the 384-byte records are all-gathered (backend nccl = RCCL over
xGMI; gloo for debugging) and rank 0 writes the reference's result files and the composition to the reference epoch.
In adaptive mode the overlap ratios behind the pair map (dense NN of raw scans, R.cpp:593-614) are dealt to the ranks and
every rank replays the target scan (R.cpp:552-589) on the gathered table.
In Direct2Ref mode the shared target scan is segmented ONCE, by rank 0; the other ranks preprocess it themselves and take its labels
from a broadcast that runs beside their own preparation (run_pairs_sharing_target)."""
import argparse
import os
import sys

import numpy as np

from . import fourd
from .binding import Series


def _agree(dist, dev, ok):
    """All ranks learn whether EVERY rank got here in good order (all-reduce MIN of a flag) — called before each
    data collective so that a rank that failed locally (bad configuration, no device, a raised error) makes all ranks
    leave together instead of leaving the others blocked in a broadcast / all-gather."""
    if dist is None:
        return bool(ok)
    import torch
    f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return bool(f.item())


def run_pairs_sharing_target(series, mine, scan, rank, world, dist, dev):
    """series.run_pairs(mine) of a Direct2Ref series on `world` ranks whose pairs all have target `scan`: rank 0 (the owner of pair
    0) segments the target as part of its run, the others preprocess it themselves and take its supervoxel labels - 4 bytes per
    point - from a broadcast that a helper thread runs beside the rank's own preparation, instead of running the front end once
    more per rank (pwicp.h: pwicp_series_*_target_labels).  EVERY rank calls this (two broadcasts are collective), with or
    without pairs of its own.  A failure on rank 0 travels as a count of -1 and the others segment for themselves.  Returns the
    records of `mine`; $PWICP_SHARE_TARGET=0: every rank segments the target itself (round 4)."""
    from .fourd import RECORD
    if dist is None or world <= 1 or os.environ.get("PWICP_SHARE_TARGET", "1") == "0":
        return series.run_pairs(mine) if len(mine) else np.zeros(0, RECORD)
    import threading
    import torch
    if rank != 0:
        series.expect_target_labels(scan)
    err = []

    def exchange():
        try:
            if dev is not None and dev.type == "cuda":
                torch.cuda.set_device(dev)                   # (the current device is per thread)
            hdr = torch.tensor([-1, 0], dtype=torch.int32)
            lab = None
            if rank == 0:
                got = series.wait_target_labels(scan)
                if got is not None:
                    lab = torch.from_numpy(got[0].copy())
                    hdr = torch.tensor([len(got[0]), got[1]], dtype=torch.int32)
            hdr = hdr.to(dev)
            dist.broadcast(hdr, src=0)
            m, nsv = int(hdr[0].item()), int(hdr[1].item())
            if m > 0:
                lab = (lab if rank == 0 else torch.empty(m, dtype=torch.int32)).to(dev)
                dist.broadcast(lab, src=0)
            if rank != 0:
                series.supply_target_labels(scan, lab.cpu().numpy() if m > 0 else None, nsv)
        except Exception as e:                               # noqa: BLE001 - the run must not wait for labels that will not come
            err.append(e)
            if rank != 0:
                series.supply_target_labels(scan, None, 0)

    th = threading.Thread(target=exchange, daemon=True)
    th.start()
    try:
        recs = series.run_pairs(mine) if len(mine) else np.zeros(0, RECORD)
    finally:
        series.close_target_labels()                         # (rank 0: a run that never reached its target wakes the helper)
        th.join()
    if err:
        print("pwicp series: rank %d: label exchange of target %d failed (%s); segmented locally" % (rank, scan, err[0]), file=sys.stderr)
    return recs


def run_series(confile, start_epoch, epoch_num, pair_mode, overlap_thd=0.75, backend="nccl", single_device=False,
               timeout_s=1800):
    """Returns True on success (on every rank).  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.
    A failure on any rank is agreed on by all ranks before the next collective: every rank returns False, none hangs
    (and the process group carries a timeout as the last line of defence)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = 0 if single_device else local_rank
    dist, dev = None, None
    if world > 1:
        import datetime
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=timeout_s)
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device), timeout=tmo)
            dev = torch.device("cuda", device)
        else:
            dist.init_process_group(backend=backend, timeout=tmo)
            dev = torch.device("cpu")
    ok = False
    series = None
    try:
        # ---- phase 1: open the series.  Adaptive mode: the overlap ratios of the candidate pairs (source j against the targets
        # j-W .. j-1) are independent (R.cpp:593-614): dealt to the ranks, all-gathered as one table, and every rank replays the
        # sequential target scan (R.cpp:552-589) on it - the same map on every rank, nothing to broadcast --------------------
        targets, good = None, True
        if pair_mode < 0 and world > 1:
            import torch
            table = None
            try:
                series = Series(confile, start_epoch, epoch_num, pair_mode, overlap_thd, device, "deferred")
                nf = series.num_scans
                W = max(1, int(os.environ.get("PWICP_ADAPTIVE_WINDOW", "6")))
                cand = [(i, j) for j in range(start_epoch + 1, nf) for i in range(max(start_epoch, j - W), j)]
                # contiguous blocks of the (source, target) order: a rank reads its block of sources + the W scans before it
                mine_ij = cand[len(cand) * rank // world: len(cand) * (rank + 1) // world]
                table = np.full((nf, nf), np.nan, np.float32)
                if mine_ij:
                    r = series.overlap_ratios(mine_ij)
                    for (i, j), v in zip(mine_ij, r):
                        table[i, j] = v
            except Exception as e:                          # noqa: BLE001 - reported, then agreed on by all ranks
                print("pwicp series: rank %d failed on its share of the overlap ratios: %s" % (rank, e), file=sys.stderr)
                good = False
            if not _agree(dist, dev, good):
                return False
            t = torch.from_numpy(table).to(dev)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            for part in parts:
                a = part.cpu().numpy()
                known = ~np.isnan(a)
                table[known] = a[known]
            try:
                series.adaptive_from_ratios(table, overlap_thd, write_pair_file=(rank == 0))
            except Exception as e:                          # noqa: BLE001
                print("pwicp series: rank %d failed to determine the pair map: %s" % (rank, e), file=sys.stderr)
                good = False
            if not _agree(dist, dev, good):
                return False
        if series is None:
            try:
                series = Series(confile, start_epoch, epoch_num, pair_mode, overlap_thd, device, targets)
            except Exception as e:                          # noqa: BLE001
                print("pwicp series: rank %d failed to open the series: %s" % (rank, e), file=sys.stderr)
                good = False
        if not _agree(dist, dev, good):
            return False
        # ---- phase 2: this rank's pairs, then the series' one exchange ---------------------------------------
        n = series.num_pairs
        mine = []
        try:
            todo = [p for p in range(n) if p % world == rank]
            if pair_mode == 0 and n > 0:
                done = run_pairs_sharing_target(series, todo, start_epoch, rank, world, dist, dev)
            else:
                done = series.run_pairs(todo)
            mine = [done[k:k + 1] for k in range(len(done))]
        except Exception as e:                              # noqa: BLE001
            print("pwicp series: rank %d failed while running its pairs: %s" % (rank, e), file=sys.stderr)
            good = False
        if not _agree(dist, dev, good):
            return False
        table = fourd.gather_records(mine, n, world, dist=dist, device=dev)
        if rank == 0:
            try:
                recs = np.concatenate([table[p].reshape(1) for p in sorted(table)]) if table else np.zeros(0, fourd.RECORD)
                series.write_results(recs)
                ok = len(recs) == n and bool(np.all(recs["status"] == 0))
            except Exception as e:                          # noqa: BLE001
                print("pwicp series: writing the results failed: %s" % e, file=sys.stderr)
                ok = False
        else:
            ok = True
        ok = _agree(dist, dev, ok)
    finally:
        if series is not None:
            series.close()
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()
    return ok


def main(argv=None):
    ap = argparse.ArgumentParser(description="Piecewise-ICP 4D series, pairs sharded over the GPUs of one node")
    ap.add_argument("confile")
    ap.add_argument("startEpoch", type=int)
    ap.add_argument("epochNum", type=int)
    ap.add_argument("pairMode", type=int, help="0: all to the reference epoch, k>0: fixed interval, <0: adaptive")
    ap.add_argument("overlapThd", type=float, nargs="?", default=0.75)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--single-device", action="store_true", help="debug: every rank uses GPU 0 (with --backend gloo)")
    a = ap.parse_args(argv)
    ok = run_series(a.confile, a.startEpoch, a.epochNum, a.pairMode, a.overlapThd, a.backend, a.single_device)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
