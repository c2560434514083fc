"""4D series on a multi-GPU node: PiecewiseICP_4D_call (R.cpp:17-215) with its pair loop (R.cpp:89-187) sharded over
one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        -m pwicp_amd.series configuration_4d.txt 0 20 0 0.75

(without a launcher: one GPU, same as the exported C entry point).  Every rank opens the same configuration, runs the
pairs p with p mod world == rank through libpwicp.so, the 384-byte records are all-gathered (backend nccl = RCCL over
xGMI; gloo for debugging) and rank 0 writes the reference's result files and the composition to the reference epoch.
In adaptive mode rank 0 determines the pair map (dense NN of raw scans on its GPU, R.cpp:552-589) and broadcasts it.
In Direct2Ref mode every rank prepares the shared target scan once (preprocessing + supervoxels), not once per pair."""
import argparse
import os
import sys

import numpy as np

from . import fourd
from .binding import Series


def run_series(confile, start_epoch, epoch_num, pair_mode, overlap_thd=0.75, backend="nccl", single_device=False):
    """Returns True on success (on every rank).  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = 0 if single_device else local_rank
    dist, dev = None, None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
            dev = torch.device("cuda", device)
        else:
            dist.init_process_group(backend=backend)
            dev = torch.device("cpu")
    ok = False
    try:
        targets = None
        if pair_mode < 0 and world > 1:
            # rank 0 computes the adaptive pair map and writes RegPairFile.txt; the others receive it
            import torch
            s0 = Series(confile, start_epoch, epoch_num, pair_mode, overlap_thd, device) if rank == 0 else None
            n_t = torch.zeros(1, dtype=torch.int32, device=dev)
            if rank == 0:
                targets = s0.adaptive_targets()
                n_t[0] = len(targets)
            dist.broadcast(n_t, src=0)
            t = torch.zeros(int(n_t.item()), dtype=torch.int32, device=dev)
            if rank == 0:
                t.copy_(torch.from_numpy(targets))
            dist.broadcast(t, src=0)
            targets = t.cpu().numpy()
            series = s0 if rank == 0 else Series(confile, start_epoch, epoch_num, pair_mode, overlap_thd, device, targets)
        else:
            series = Series(confile, start_epoch, epoch_num, pair_mode, overlap_thd, device)
        with series:
            n = series.num_pairs
            done = series.run_pairs([p for p in range(n) if p % world == rank])
            mine = [done[k:k + 1] for k in range(len(done))]
            table = fourd.gather_records(mine, n, world, dist=dist, device=dev)
            if rank == 0:
                recs = np.concatenate([table[p].reshape(1) for p in sorted(table)]) if table else np.zeros(0, fourd.RECORD)
                series.write_results(recs)
                ok = len(recs) == n and bool(np.all(recs["status"] == 0))
            else:
                ok = True
        if dist is not None:
            import torch
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.broadcast(flag, src=0)
            ok = bool(flag.item())
    finally:
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
