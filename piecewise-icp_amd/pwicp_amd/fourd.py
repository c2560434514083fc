"""4D series: pair schedule, sharding of independent pairs over ranks, gather of the per-pair result
records, composition to the reference epoch.

Reference: PiecewiseICP_4D_call src/Registration.cpp:17-215 (pair choice 94-103), the pair loop 89-187 whose
iterations are mutually independent, calTransToReferenceEpoch Registration.cpp:977-1153.
One process per GPU; the only exchange is an all-gather of a fixed 384-byte record per pair
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests)."""
import numpy as np

RECORD_BYTES = 384
# = struct pwicp_pair_record of include/pwicp.h
_REC = np.dtype([("pair", "<i4"), ("status", "<i4"), ("n_outer", "<i4"), ("n_inner", "<i4"),
                 ("T", "<f4", (16,)), ("VCM", "<f8", (36,)), ("n_corr", "<i8"), ("t_loop_ms", "<f4"), ("t_pair_ms", "<f4")])
assert _REC.itemsize == RECORD_BYTES
RECORD = _REC


def pair_schedule(start_epoch, epoch_num, pair_mode, adaptive_pairs=None):
    """List of (target_index, source_index) in the order of the loop R.cpp:89-103.
    pair_mode 0: all to the reference; k>0: fixed interval k; <0: adaptive (map source->target, relative
    to start_epoch, as calAdaptivePairSequence returns it, R.cpp:570)."""
    out = []
    for i in range(start_epoch, epoch_num - 1):
        step = i - start_epoch + 1
        if pair_mode > 0:
            ref = start_epoch if pair_mode >= step else (i + 1 - pair_mode)
        elif pair_mode < 0:
            ref = adaptive_pairs[i + 1]          # B.8: relative indices used as absolute (only correct for start 0)
        else:
            ref = start_epoch
        out.append((ref, i + 1))
    return out


def shard(pairs, rank, world):
    """pair p -> rank p mod world."""
    return [(p, tp) for p, tp in enumerate(pairs) if p % world == rank]


def pack_record(pair_id, status, n_outer, n_inner, T16, VCM36, n_corr):
    r = np.zeros(1, _REC)
    r["pair"], r["status"], r["n_outer"], r["n_inner"] = pair_id, status, n_outer, n_inner
    r["T"][0] = np.asarray(T16, np.float32).reshape(16)
    r["VCM"][0] = np.asarray(VCM36, np.float64).reshape(36)
    r["n_corr"] = n_corr
    return r


def gather_records(local_records, n_pairs, world, dist=None, device=None):
    """All-gather of the per-pair records. local_records: list of 1-element _REC arrays owned by this rank.
    Every rank contributes ceil(n_pairs/world) slots (unused slots have pair = -1).  Returns a dict
    pair_id -> record on every rank."""
    slots = (n_pairs + world - 1) // world
    buf = np.zeros(slots, _REC)
    buf["pair"] = -1
    for k, rec in enumerate(local_records):
        buf[k] = rec[0]
    if dist is None or world == 1:
        allb = [buf]
    else:
        import torch
        t = torch.from_numpy(buf.view(np.uint8).copy())
        if device is not None:
            t = t.to(device)
        if t.is_cuda and hasattr(dist, "all_gather_into_tensor"):
            # one flat RCCL all-gather (world * slots * 384 bytes)
            flat = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
            dist.all_gather_into_tensor(flat, t)
            allb = [c.cpu().numpy().view(_REC) for c in flat.view(world, -1)]
        else:
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            allb = [o.cpu().numpy().view(_REC) for o in outs]
    table = {}
    for b in allb:
        for rec in b:
            if rec["pair"] >= 0:
                table[int(rec["pair"])] = rec.copy()
    return table


def compose_to_reference(T_list, VCM_list, pair_mode, adaptive_pairs=None):
    """calTransToReferenceEpoch, R.cpp:1051-1109.  T_list[i] (4x4 float32), VCM_list[i] (6x6 float64) of the
    i-th pair in file order. Returns (T2ref list, VCM2ref list)."""
    n = len(T_list)
    outT, outV = [], []
    for i in range(n):
        if pair_mode < 0:
            accT = T_list[i].astype(np.float32)
            accV = VCM_list[i].astype(np.float64)
            idx = i + 1
            for _ in range(i + 1):
                idx = adaptive_pairs[idx]
                if idx == 0:
                    break
                M = T_list[idx - 1].astype(np.float32)
                accT = (M @ accT).astype(np.float32)
                Md = M.astype(np.float64)
                R = Md[:3, :3]
                t = Md[:3, 3]
                SS = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float64)
                Adj = np.zeros((6, 6))
                Adj[:3, :3] = R
                Adj[3:, 3:] = R
                Adj[3:, :3] = SS @ R
                accV = VCM_list[idx - 1] + Adj @ accV @ Adj.T
        else:
            if pair_mode == 0 or i < pair_mode:
                accT = T_list[i].astype(np.float32)
                accV = VCM_list[i].astype(np.float64)
            else:
                accT = np.eye(4, dtype=np.float32)
                accV = np.zeros((6, 6))
                for j in range(n):
                    accT = (T_list[i - pair_mode * j].astype(np.float32) @ accT).astype(np.float32)
                    accV = VCM_list[i - pair_mode * j] + accV
                    if i - pair_mode * j < pair_mode:
                        break
        outT.append(accT)
        outV.append(accV)
    return outT, outV
