"""N>1 data path on CPU: world-size-2 gloo run of the video shard -> all-gather -> rank merge.
The collective layer (hirest_amd.retrieval.shard_range / gather_rows) is backend-agnostic: RCCL on GPU
tensors in production, gloo on CPU tensors here.  Embedding rows are produced per shard by the CPU
oracle's pooling so the gathered matrix can be checked against the unsharded computation bit-exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hirest_amd import retrieval, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, V, F, E, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_cpu as O
        frames = synth.tensor("dist.fe", (V, F, E), 1.0, 3, mean=0.05)          # the whole corpus (cheap here)
        lo, hi, per = retrieval.shard_range(V, rank, world)
        local = O.pool_video(frames[lo:hi]) if hi > lo else torch.zeros((0, E))
        allv = retrieval.gather_rows(local, V)
        want = O.pool_video(frames)
        ok = torch.equal(allv, want)
        # rank merge is replicated: every rank must produce identical top-k
        te = O.l2_normalize(synth.tensor("dist.te", (7, E), 1.0, 3))
        top = O.topk_with_ties(O.similarity(te, allv), torch.arange(V), 5)
        gathered = [torch.zeros_like(top) for _ in range(world)]
        dist.all_gather(gathered, top)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        q.put((rank, ok, same, (lo, hi, per)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V", [8, 7, 1])
def test_gloo_world2_shard_gather(V):
    world, F, E = 2, 4, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, V, F, E, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, same, _ in res:
        assert ok and same, f"rank {rank}: gathered rows differ from the unsharded result"


def test_shard_range_covers_everything():
    for V in (1, 7, 8, 4096, 4282):
        for world in (1, 2, 4, 8):
            got = []
            for r in range(world):
                lo, hi, per = retrieval.shard_range(V, r, world)
                assert hi - lo <= per
                got += list(range(lo, hi))
            assert got == list(range(V))


def test_gather_rows_without_group_is_identity():
    x = torch.arange(12.0).reshape(4, 3)
    assert torch.equal(retrieval.gather_rows(x, 4), x)


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hirest_amd import train
        lin, other = torch.nn.Linear(5, 3), torch.nn.Linear(2, 2)
        for p in lin.parameters():
            p.grad = torch.full_like(p, float(rank + 1))               # rank 0: 1, rank 1: 2 -> mean 1.5
        other.weight.grad = torch.full_like(other.weight, 4.0) if rank == 0 else None    # a tensor only rank 0 reached
        params = list(lin.parameters()) + list(other.parameters())
        train.allreduce_gradients(params, bucket_bytes=32)                # several small buckets
        ok = all(torch.equal(p.grad, torch.full_like(p, 1.5)) for p in lin.parameters())
        # a tensor with a gradient on SOME rank is averaged (zeros from the ranks that did not reach it); one that no rank reached
        # stays None, as in a single-process run, so the optimizer skips it (no weight decay / moment decay on unused heads)
        ok = ok and torch.equal(other.weight.grad, torch.full_like(other.weight, 2.0)) and other.bias.grad is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2():
    """The exchange step of data-parallel training (run.py:93): gradients averaged over ranks in flat buckets."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
