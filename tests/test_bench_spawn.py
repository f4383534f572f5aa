"""`python bench.py --gpus N` started PLAINLY must spawn its own ranks (the driver starts benches that way); the
torch.distributed.run entry must keep working.  CSS_BENCH_DRY=1 stops each rank after the rendezvous and the rank
census, so this runs without a GPU (gloo).  Env bootstrap as the reference's utils/torch_utils.py:102-113 reads it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, CSS_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _one_json_line(stdout):
    # the communication libraries chat on stdout themselves (gloo: "[Gloo] Rank 0 is connected to ...", RCCL 2.26: a
    # version banner at exit); bench.py hands file descriptor 1 to stderr and writes its record to the real stdout
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout          # rank 0 prints ONE line, the other ranks nothing
    return json.loads(lines[0])


def test_plain_launch_spawns_its_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["dry_run"] and line["n_gpus"] == 2
    col = line["collective"]
    assert col["world"] == 2 and col["ranks_seen_by_all_gather"] == [0, 1]
    assert sorted(i["rank"] for i in col["ranks"]) == [0, 1]
    assert len({i["pid"] for i in col["ranks"]}) == 2      # two real processes


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_torch_distributed_run_entry_still_works():
    for attempt in range(2):    # (a port can be taken between the probe and the rendezvous)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
        r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["collective"]["world"] == 2


def test_a_failing_rank_stops_the_launch():
    env = _env()
    env["CSS_BENCH_BACKEND"] = "gloo"
    env.pop("CSS_BENCH_DRY")                 # without a GPU every rank exits with "bench.py needs a GPU"
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr


def test_one_rank_can_take_the_sharded_path():
    """CSS_BENCH_FORCE_SHARDED=1: the N > 1 code path with one rank (on a 1-GPU box that is the RCCL smoke,
    profiles/r03_rccl_world1.log); here: rendezvous + census at world 1."""
    env = _env()
    env["CSS_BENCH_FORCE_SHARDED"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["dry_run"] and line["collective"]["world"] == 1 and line["collective"]["ranks_seen_by_all_gather"] == [0]
