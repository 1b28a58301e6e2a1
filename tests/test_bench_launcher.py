"""`python bench.py --gpus N` with no launcher around it starts its own N ranks (the driver's command form): the
parent never touches a GPU, relays rank 0's JSON line and fails when a rank fails.  Runs on the CPU: the ranks meet
over gloo (MPPI_BENCH_LAUNCH_CHECK makes them stop there instead of benchmarking)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MPPI_BENCH_LAUNCH_CHECK"] = mode
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "7", "--warmup", "1",
                           "--workload", "c4"], env=env, capture_output=True, text=True, timeout=300)


def test_bench_starts_its_own_ranks_and_relays_rank_zero():
    res = _run("1", 2)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout  # ONE JSON line on stdout, whatever else rank 0 printed
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rank_sum"] == 3.0 and d["steps"] == 7  # both ranks took part; the flags travelled
    assert "a line that is not the result" in res.stderr


def test_bench_fails_when_a_rank_fails():
    res = _run("fail", 2)
    assert res.returncode != 0
    assert "ranks exited with" in res.stderr
