#!/usr/bin/env python3
"""Can the RCCL leg of the sharded retrieval (retrieval.RowGather: one all_gather_into_tensor) be executed with two ranks on ONE
GPU?  (The development boxes have a single MI355X; the N > 1 bench is the driver's.)  Launch:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/rccl_one_gpu_check.py
Prints what happened; exit code 0 either way (a refusal by RCCL is an answer)."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import retrieval

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    n_total, E = 11, 1024
    per = (n_total + world - 1) // world
    rows = torch.arange(n_total * E, dtype=torch.float32).reshape(n_total, E)
    local = rows[rank * per:min((rank + 1) * per, n_total)].cuda()
    out = retrieval.RowGather()(local, n_total)
    torch.cuda.synchronize()
    ok = torch.equal(out.cpu(), rows)
    print(f"rank {rank}: RCCL all_gather_into_tensor of two ranks on one GPU: gathered rows equal the corpus: {ok}", flush=True)
    dist.destroy_process_group()
except Exception as e:      # noqa: BLE001
    print(f"rank {rank}: RCCL refused two ranks on one device: {type(e).__name__}: {str(e)[:300]}", flush=True)
