"""`bench.py --gpus N` must run N ranks or fail loudly (args.gpus used to be ignored).  The launcher, the
world-size checks and the timing protocol live in hirest_amd/launch.py and are shared with bench.py; here they are driven
on CPU at world size 2 over gloo through tests/_rank_stub.py (a stub step in place of the GPU encoder)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
STUB = os.path.join(HERE, "_rank_stub.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, STUB] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_world2():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    assert d["step_calls"] == 5                       # W untimed + exactly K timed
    assert d["gathered_sum"] == d["expected_sum"]     # both ranks' rows arrived, in the preallocated gather buffer
    assert d["elapsed"] > 0
    # every rank's own time up to its own sync, before the closing barrier: one entry per rank, none longer than the reported MAX + barrier
    assert len(d["per_rank_s"]) == 2 and all(0 < t <= d["elapsed"] * 1.001 for t in d["per_rank_s"])


def test_single_rank_needs_no_launcher():
    d = _json_line(_run(["--gpus", "1"]).stdout)
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["gathered_sum"] == d["expected_sum"] and len(d["per_rank_s"]) == 1


def test_fewer_devices_than_ranks_fails_loudly():
    r = _run(["--gpus", "2", "--visible", "1"])
    assert r.returncode != 0 and "only 1 GPU(s) are visible" in r.stderr and "{" not in r.stdout


def test_world_size_mismatch_fails_loudly():
    # a launcher that started one rank for --gpus 2 (what `python bench.py --gpus 8` used to be, silently)
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout


def test_bench_uses_the_launcher():
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "launch.ensure_ranks(args.gpus" in src and "launch.init_ranks(args.gpus" in src and "launch.timed_steps(" in src
    assert '"rccl_ranks"' in src
    from hirest_amd import launch
    cmd = launch.launch_command(8, "bench.py", ["--gpus", "8", "--steps", "2"], port=29517)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                       "--master-port", "29517", "bench.py", "--gpus", "8", "--steps", "2"]
