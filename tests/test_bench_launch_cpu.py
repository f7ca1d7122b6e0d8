"""`python bench.py --gpus N` must start N ranks by itself.  CPU rehearsal of exactly that path:
no WORLD_SIZE in the environment -> bench.py becomes the launcher (a torch.distributed.run child with N fresh rank
processes), the ranks form the process group (gloo here, RCCL on the GPU node), all-gather their rank ids and rank 0
prints the one JSON line with n_gpus == N and rccl_ranks == N.  Also: a launcher whose world size disagrees with
--gpus is refused with a non-zero status instead of silently measuring one GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bench_starts_its_own_ranks_world2():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=_env(ZVEC_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["ranks"] == [0, 1] and line["backend"] == "gloo"


def test_world_size_mismatch_is_refused():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert p.returncode == 2
    assert "WORLD_SIZE=1 but --gpus 8" in p.stderr
