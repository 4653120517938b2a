"""bench.py's own launcher: `python bench.py --gpus N` (no torchrun around it) must become N ranks of one process group.

The GPU work itself cannot run here; `--dry-ranks` stops every rank after the rendezvous (gloo) and reports who was there.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_is_the_drivers_command():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "2"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "2"]


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=e)


def test_bare_invocation_spawns_its_ranks():
    r = _run(["--gpus", "2", "--dry-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    got = json.loads(line)
    assert got["world"] == 2 and got["backend"] == "gloo"
    assert sorted(d["rank"] for d in got["dry_ranks"]) == [0, 1]
    assert len({d["pid"] for d in got["dry_ranks"]}) == 2


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "3", "--dry-ranks"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
