#!/usr/bin/env python3
"""Stand-in for bench.py on a box without GPUs: the same launcher (hirest_amd.launch.ensure_ranks / init_ranks), the same
timing protocol (timed_steps) and the same exchange step (retrieval.gather_rows), with a CPU stub in place of the encoder
and gloo in place of RCCL.  Driven by tests/test_launch.py; prints ONE JSON line on rank 0 like bench.py does."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from hirest_amd import launch  # noqa: E402

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--visible", type=int, default=None, help="pretend this many devices are visible")
    args = ap.parse_args()
    launch.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], visible_devices=args.visible)
    rank, local_rank, world = launch.init_ranks(args.gpus, "gloo")
    import torch.distributed as dist
    from hirest_amd import retrieval
    V_local, E = 4, 8
    calls = []

    def step():
        calls.append(1)
        pooled = torch.full((V_local, E), float(rank + 1))            # the "encoder": rank r produces rows of r + 1
        allv = retrieval.gather_rows(pooled, V_local * world)
        return allv.sum().item()

    info = {}
    elapsed, last = launch.timed_steps(step, args.warmup, args.steps, lambda: None, info=info)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
                          "warmup": args.warmup, "step_calls": len(calls), "elapsed": elapsed, "per_rank_s": info["per_rank_s"], "gathered_sum": last,
                          "expected_sum": float(V_local * E * sum(range(1, world + 1)))}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
