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


def _torchrun(args, env=None, nproc=2):
    """The DRIVER's route: torch.distributed.run around bench.py (bench.self_launch is not involved)."""
    import bench
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", *bench.RANK_ENV_DEFAULTS):
        e.pop(k, None)
    e.update(env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(bench.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=e)


def test_drivers_own_torchrun_command_gets_the_rank_environment():
    """The RCCL environment defaults must hold when the driver starts the ranks itself (missing 1)."""
    import bench
    r = _torchrun(["--dry-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got["world"] == 2 and sorted(d["rank"] for d in got["dry_ranks"]) == [0, 1]
    for d in got["dry_ranks"]:
        assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == bench.RANK_ENV_DEFAULTS["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert d["env"]["OMP_NUM_THREADS"] is not None


def test_rank_environment_respects_what_the_caller_set():
    r = _torchrun(["--dry-ranks"], env={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert all(d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" for d in got["dry_ranks"])


def test_a_failing_rank_leaves_one_json_error_line():
    for bad in ("0", "1"):
        r = _torchrun(["--dry-ranks"], env={"BENCH_DRY_FAIL_RANK": bad})
        assert r.returncode != 0
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, (bad, r.stdout, r.stderr[-1500:])
        got = json.loads(lines[0])
        assert got["value"] is None and "error" in got and got["n_gpus"] == 2
        assert got["rccl"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and got["rccl"]["ranks"] == 2


def test_eight_ranks_as_the_driver_launches_them():
    """`--gpus 8`: the launch the driver's scaling run uses -- eight ranks of one group (gloo here), each with its shard's geometry:
    rank r of 8 holds rows [r B/8, (r + 1) B/8) of configs[3]'s 1 048 576."""
    r = _torchrun(["--dry-ranks"], nproc=8)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got["world"] == 8 and sorted(d["rank"] for d in got["dry_ranks"]) == list(range(8))
    assert len({d["pid"] for d in got["dry_ranks"]}) == 8
    assert sorted(d["local_rank"] for d in got["dry_ranks"]) == list(range(8))
    from ldpc_amd.sharding import shard_range
    B = 1048576
    cuts = [shard_range(B, rank, 8) for rank in range(8)]
    assert cuts[0][0] == 0 and cuts[-1][1] == B and all(b - a == 131072 for a, b in cuts) and all(cuts[k][1] == cuts[k + 1][0] for k in range(7))
    r2 = _run(["--gpus", "8", "--dry-ranks"])  # and bench.py's own launcher
    assert r2.returncode == 0, r2.stderr[-2000:]
    got2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert got2["world"] == 8 and len({d["pid"] for d in got2["dry_ranks"]}) == 8
