"""not-gpu tier: bench.py's N-rank launch path.  `python bench.py --gpus N` must produce an N-rank run by itself (fresh child
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one JSON line from rank 0, non-zero exit on any failure); here the
compute is stubbed (BBP_BENCH_STUB=1: gloo, no engine) -- the real ranks differ only in backend (nccl = RCCL) and workload."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(BBP_BENCH_STUB="1", **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_launches_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["data"] == "stub" and "not a measurement" in out["metric"]
    # whole-job aggregate: units of BOTH ranks over the max-over-ranks time
    assert abs(out["value"] - 2 * 5 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    # ... and every rank's own figure beside it (a straggler shows): the slowest rank's time is the step time, no rank is faster than 1/N of `value`
    pr = out["per_rank"]
    assert len(pr["value"]) == 2 and len(pr["ms_per_step"]) == 2
    assert abs(max(pr["ms_per_step"]) - out["ms_per_step"]) < 1e-9 + 1e-6 * out["ms_per_step"]
    assert all(v >= out["value"] / 2 * (1 - 1e-6) for v in pr["value"])


def test_world_size_mismatch_is_refused():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)


def test_failing_rank_fails_the_launch():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "stream"], {"BBP_BENCH_STUB_FAIL_RANK": "1"})
    assert p.returncode != 0
