"""`python bench.py --gpus N` must start its own N ranks (the driver's N = 1 command has no launcher around it, and an
N = 8 command of the same shape must not die before touching a GPU), and the torchrun form must keep working.  No GPU
here: `--dry-run` runs everything of the N-rank path that needs none — launch, rendezvous (gloo), the packed all-gather of
the per-rank lists (image_search_amd.search.ShardExchange) and the merge through the C ABI (mi_knn_merge)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bare_gpus_2_launches_its_own_ranks(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--k", "10"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["dry_run"] is True and out["exchange_ok"] is True


def test_torchrun_form_still_works(built):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run", "--k", "1000"], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["exchange_ok"] is True


def test_the_drivers_scale_command_at_its_own_n_of_8(built):
    """The round-end SCALE run is `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 --steps K --warmup W`; with --dry-run the same command line rehearses here on 8 gloo
    ranks: rendezvous, the packed all-gather of 8 per-rank lists, mi_knn_merge, the known merged answer on EVERY rank."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = _env()
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2",
                        "--dry-run"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 8 and out["steps"] == 5 and out["warmup"] == 2 and out["exchange_ok"] is True and out["scaling"] == "weak"


def test_bare_gpus_8_launches_its_own_ranks(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run", "--k", "1000"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 8 and out["dry_run"] is True and out["exchange_ok"] is True


def test_a_failing_rank_fails_the_launcher(built):
    # an argument the ranks reject: the child's exit code must come through, and no result line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--k", "0"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
