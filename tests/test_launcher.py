"""The N > 1 launch path of bench.py on CPU: a plain `python bench.py --gpus N` starts N fresh rank processes itself
(shimmer_amd/launch.py), a `torch.distributed.run` launch is recognised and not re-launched, rank processes never import torch, the
128-byte id travels through the file store, a failing rank ends the whole launch with a non-zero code instead of a hang. `--dry-run`
stops short of the GPU (no communicator can exist here); the same launcher with a real communicator runs in tests/test_gpu_multi.py."""
import json
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from shimmer_amd import launch  # noqa: E402


def _run(cmd, timeout=180, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SHM_STORE_DIR")}
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_plain_start_launches_one_process_per_rank():
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr.decode()
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1  # ONE JSON line: rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launcher"] == "shimmer_amd.launch"
    ranks = out["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1] and [x["local_rank"] for x in ranks] == [0, 1]
    assert ranks[0]["pid"] != ranks[1]["pid"] and ranks[0]["ppid"] == ranks[1]["ppid"]  # two children of the one launcher
    assert all(x["id_ok"] and not x["torch_imported"] for x in ranks)
    assert out["max_dt_s"] >= 0.02  # the max over ranks, not rank 0's own 10 ms


def test_failing_rank_ends_the_launch_with_an_error_code():
    t0 = time.monotonic()
    r = _run([sys.executable, "bench.py", "--gpus", "3", "--dry-run", "--dry-run-fail-rank", "2"], timeout=120)
    assert r.returncode != 0
    assert time.monotonic() - t0 < 60  # the peers saw the failure through the store; nobody waited for the 600 s limit
    assert not [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]  # and no JSON line claims a result


def test_launch_under_torch_distributed_run_is_not_relaunched():
    """The driver's way: the agent exports RANK / WORLD_SIZE; bench.py must run as that rank (no second level of children) and find its
    peers through the store named after MASTER_PORT and the agent's pid."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(29700 + os.getpid() % 200), "bench.py", "--gpus", "2", "--dry-run"], timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["launcher"] == "external" and len(out["ranks"]) == 2
    assert all(not x["torch_imported"] for x in out["ranks"])  # the agent imports torch; the rank processes do not


def test_launcher_single_rank_and_exit_code_passthrough(tmp_path):
    script = tmp_path / "child.py"
    script.write_text("import os, sys\nprint('rank', os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'])\n"
                      "sys.exit(5 if os.environ['RANK'] == '1' else 0)\n")
    r = _run([sys.executable, "-c", f"import sys; sys.path.insert(0, {str(ROOT)!r}); from shimmer_amd import launch; "
              f"sys.exit(launch.spawn_ranks([sys.executable, {str(script)!r}], 2))"])
    assert r.returncode == 5
    assert r.stdout.decode().strip() == "rank 0 2 0 127.0.0.1"          # rank 0's stdout is the launcher's stdout
    assert "[rank 1] rank 1 2 1 127.0.0.1" in r.stderr.decode()           # the others go to stderr, prefixed


def test_launcher_timeout_stops_the_ranks(tmp_path):
    script = tmp_path / "sleep.py"
    script.write_text("import time\ntime.sleep(600)\n")
    t0 = time.monotonic()
    r = _run([sys.executable, "-c", f"import sys; sys.path.insert(0, {str(ROOT)!r}); from shimmer_amd import launch; "
              f"sys.exit(launch.spawn_ranks([sys.executable, {str(script)!r}], 2, timeout_s=2.0))"], timeout=120)
    assert r.returncode == 124 and time.monotonic() - t0 < 60


def test_file_store_collectives(tmp_path):
    world, res = 4, {}

    def worker(rank):
        st = launch.FileStore(tmp_path / "store", rank, world, timeout_s=30)
        uid = st.broadcast("id", os.urandom(128) if rank == 0 else None)
        st.barrier()
        mx = st.allreduce_max("t", 0.5 + rank)
        got = st.allgather("g", str(rank * rank).encode())
        res[rank] = (uid, mx, got)
        st.finish()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert len(res) == world
    assert len({res[r][0] for r in range(world)}) == 1 and len(res[0][0]) == 128
    assert all(res[r][1] == 3.5 for r in range(world))
    assert all(res[r][2] == [b"0", b"1", b"4", b"9"] for r in range(world))
    assert not (tmp_path / "store").exists()  # rank 0 removed it after everyone said goodbye


def test_file_store_reports_a_failed_peer_and_times_out(tmp_path):
    a = launch.FileStore(tmp_path / "s", 0, 2, timeout_s=0.3)
    with pytest.raises(launch.StoreTimeout):
        a.get("never")
    b = launch.FileStore(tmp_path / "s", 1, 2, timeout_s=5)
    b.fail("out of memory")
    with pytest.raises(RuntimeError, match="failed.1: out of memory"):
        a.barrier()
