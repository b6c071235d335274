"""CPU: `python bench.py --gpus N` must start its own N ranks when no launcher did (the driver's scaling command), form the
process group and print ONE JSON line from rank 0.  Without a GPU the ranks form a gloo group and the line says so (there is
no CPU path for the workloads themselves); on a GPU node the same path initialises RCCL."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=240)


def test_self_launch_two_ranks_forms_a_group():
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("CPU launch-path check (a GPU box runs the real bench)")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_size_seen"] == 2 and j["backend"] == "gloo" and "no GPU" in j["error"]
    assert r.returncode == 3
    assert "torch.distributed.run" in r.stderr           # the ranks were started as a child launcher, not by exec


def test_world_size_mismatch_is_a_clear_error():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
