"""One process per GPU: the launcher and the timing protocol shared by ``bench.py`` and the multi-rank tools.

The reference's multi-process entry is per-``LOCAL_RANK`` initialisation under an external launcher
(/root/reference/run.py:853,873-878: ``torch.cuda.set_device(local_rank)`` + ``init_process_group('nccl')``); its
retrieval script shards by ``ids[process_id::num_process]`` with no launcher at all
(inference_video_retrieval.py:226-237).  Here a script asked for N ranks either *is* one of N ranks already
(``WORLD_SIZE`` set by ``torch.distributed.run``) or re-executes itself under ``torch.distributed.run`` with N ranks on
this node.  There is no silent degradation: asking for N ranks and getting another world size raises.

    ensure_ranks(n, script, argv)   called first thing by the script: returns in a rank process, never returns in the
                                    launching parent (it exits with the children's status)
    init_ranks(n, backend)          rank / local_rank / world from the environment, process group, world-size check
    timed_steps(step, warmup, steps, sync)
                                    W untimed steps, then exactly K steps bracketed by barrier + device sync on both
                                    sides, MAX over ranks
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Callable, Optional, Sequence, Tuple

# the host driver only supports dmabuf IPC: RCCL's intra-node transport needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n: int, script: str, argv: Sequence[str], port: Optional[int] = None) -> list:
    """The command line that starts `script argv...` as n ranks on this node (rendezvous on 127.0.0.1: the container
    hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def ensure_ranks(n: int, script: str, argv: Sequence[str], visible_devices: Optional[int] = None) -> None:
    """Make `script` run as n ranks.

    * Already a rank (``WORLD_SIZE`` in the environment): the world size must equal n, otherwise SystemExit — a launcher
      that started 4 ranks for ``--gpus 8`` is an error, not a 4-GPU result labelled 8.
    * n == 1: nothing to do.
    * Otherwise re-execute under ``torch.distributed.run`` with n ranks and exit with its status.  `visible_devices`
      (the caller's ``torch.cuda.device_count()``; None = do not check, e.g. CPU/gloo tests) must be >= n: one rank per
      GPU, never two ranks sharing one.
    """
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != n:
            raise SystemExit(f"{os.path.basename(script)}: asked for {n} ranks but the launcher started WORLD_SIZE={env_world}")
        return
    if n == 1:
        return
    if n < 1:
        raise SystemExit(f"{os.path.basename(script)}: --gpus must be >= 1, got {n}")
    if visible_devices is not None and visible_devices < n:
        raise SystemExit(f"{os.path.basename(script)}: asked for {n} ranks (one per GPU) but only {visible_devices} GPU(s) "
                         "are visible; refusing to report a smaller run as an N-GPU result")
    rc = subprocess.call(launch_command(n, script, list(argv)), env=dict(os.environ))
    raise SystemExit(rc)


def init_ranks(n_expected: int, backend: str = "nccl", device=None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from RANK / LOCAL_RANK / WORLD_SIZE; initialises the process group when world > 1
    ("nccl" is RCCL on ROCm) and checks the world size against what the caller was asked to run."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_expected:
        raise SystemExit(f"expected {n_expected} ranks, environment says WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, **kw)
        if dist.get_world_size() != n_expected:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {n_expected}")
    return rank, local_rank, world


def timed_steps(step: Callable[[], object], warmup: int, steps: int, sync: Callable[[], None],
                on_timed_start: Optional[Callable[[], None]] = None, reduce_device=None,
                info: Optional[dict] = None) -> Tuple[float, object]:
    """The bench contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier and a device sync
    on both sides; returns (elapsed seconds = MAX over ranks, last step's result).  `info` (a dict), when given, receives
    ``per_rank_s``: every rank's own time from the common start to its own device sync, BEFORE the closing barrier — the skew
    between ranks that the MAX hides (a slow GPU, a late gather)."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    for _ in range(warmup):
        step()
    sync()
    if multi:
        dist.barrier()
    if on_timed_start is not None:
        on_timed_start()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    own = time.perf_counter() - t0
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        rd = reduce_device if reduce_device is not None else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=rd)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if info is not None:
            mine = torch.tensor([own], dtype=torch.float64, device=rd)
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            info["per_rank_s"] = [float(e.item()) for e in every]
    elif info is not None:
        info["per_rank_s"] = [own]
    return elapsed, out
