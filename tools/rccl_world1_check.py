"""RCCL sanity on one GPU: init the nccl backend with world size 1 and push one record buffer through the same
all_gather_into_tensor / all_reduce / broadcast / barrier calls the N>1 paths of bench.py and pwicp_amd.series use."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "piecewise-icp_amd"))
from pwicp_amd import fourd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
buf = np.zeros(3, fourd.RECORD); buf["pair"] = [0, 1, -1]; buf["T"][0] = np.arange(16)
t = torch.from_numpy(buf.view(np.uint8).copy()).to(dev)
flat = torch.empty(t.numel(), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(flat, t)
back = flat.cpu().numpy().view(fourd.RECORD)
assert np.array_equal(back["pair"], [0, 1, -1]) and np.array_equal(back["T"][0], np.arange(16, dtype=np.float32))
x = torch.tensor([2.5], dtype=torch.float64, device=dev); dist.all_reduce(x, op=dist.ReduceOp.MAX); assert x.item() == 2.5
f = torch.tensor([1], dtype=torch.int32, device=dev); dist.broadcast(f, src=0)
dist.barrier(device_ids=[0]); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
