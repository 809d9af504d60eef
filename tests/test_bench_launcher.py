"""`python3 bench.py --gpus N` with no launcher in front starts the N ranks itself: before
torch or HIP are touched, as child processes with the environment torch.distributed.run would give them; rank 0's ONE
JSON line is handed on; a failing rank fails the run.  CPU only: the ranks are a stub that records what it was given."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import json, os, sys
d = os.environ["SHK_STUB_DIR"]
rank = int(os.environ["RANK"])
with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
    json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")} | {"argv": sys.argv[1:]}, f)
if os.environ.get("SHK_STUB_FAIL") == str(rank):
    sys.exit(3)
if rank == 0:
    print("RCCL version chatter that is not the line")
    print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}))
'''


def _run(tmp_path, n, extra_env=None, args=()):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(SHK_BENCH_CHILD=f"{sys.executable} {stub}", SHK_STUB_DIR=str(tmp_path))
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *args], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


@pytest.mark.parametrize("n", [2, 8])
def test_gpus_n_starts_n_ranks_with_the_launcher_environment(tmp_path, n):
    r = _run(tmp_path, n, args=("--steps", "7", "--config", "4"))
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "n_gpus": n}   # rank 0's JSON line, nothing else
    ports = set()
    for rank in range(n):
        d = json.load(open(tmp_path / f"rank{rank}.json"))
        assert d["RANK"] == str(rank) and d["LOCAL_RANK"] == str(rank) and d["WORLD_SIZE"] == str(n)
        assert d["MASTER_ADDR"] == "127.0.0.1" and d["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert d["argv"] == ["--gpus", str(n), "--steps", "7", "--config", "4"]     # the ranks see the caller's arguments
        ports.add(d["MASTER_PORT"])
    assert len(ports) == 1 and int(ports.pop()) > 0


def test_a_failing_rank_fails_the_run(tmp_path):
    r = _run(tmp_path, 4, extra_env={"SHK_STUB_FAIL": "2"})
    assert r.returncode != 0
    assert "rank 2 exited with 3" in r.stderr.decode()


def test_under_a_launcher_bench_does_not_launch_again(tmp_path):
    """WORLD_SIZE set (torch.distributed.run in front): bench.py is a rank, not a launcher — it must go on to parse its
    own arguments (here it fails on an unknown one, which shows it did not spawn the stub)."""
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", SHK_BENCH_CHILD=f"{sys.executable} {stub}", SHK_STUB_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-such-flag"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 2 and b"unrecognized arguments" in r.stderr
    assert not (tmp_path / "rank0.json").exists()
