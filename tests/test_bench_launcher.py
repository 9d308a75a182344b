"""bench.py's own launcher: `python bench.py --gpus N` without RANK / WORLD_SIZE must start N rank processes itself (the form the
round driver uses for N = 1), relay rank 0's single JSON line and fail when a rank fails. --dry-run keeps the GPU out of it."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


@pytest.mark.parametrize("n", [1, 3])
def test_self_launch_starts_n_ranks(n):
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # ONE JSON line, from rank 0
    assert p.stdout.strip().splitlines() == lines, p.stdout   # ... and NOTHING else on stdout (gloo announces its peers on std::cout: bench.py turns fd 1 into stderr)
    res = json.loads(lines[0])
    assert res["n_gpus"] == n and [r["rank"] for r in res["config"]["ranks"]] == list(range(n))
    assert len({r["pid"] for r in res["config"]["ranks"]}) == n          # n distinct processes
    assert res["config"]["self_launched"] == (n > 1)


def test_under_a_launcher_the_process_is_a_rank():
    """With RANK / WORLD_SIZE set (torch.distributed.run form) bench.py must NOT spawn: world 1 given explicitly runs in this process."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=120,
                       env=_clean_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert p.returncode == 0, p.stdout + p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["config"]["self_launched"] is False


def test_a_failing_rank_fails_the_launch():
    """No GPU here: every rank exits with the product's "no HIP device" error; the launcher reports it and exits non-zero (no hang)."""
    import frt
    if frt.lib().frt_device_count() > 0:
        pytest.skip("needs a machine without a HIP device")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert p.returncode != 0
    assert "no HIP device" in p.stderr and "exited with code" in p.stderr
