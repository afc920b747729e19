"""GPU tier: everything that involves MORE THAN ONE engine instance, rehearsed on the one card a GPU box has --
  * bench.py's multi-rank path with the REAL engine: N fresh rank processes, one context each, the block split of
    dusk_blindbidproof_amd/sharding.py, the path's one collective (proof records / flags gathered to rank 0) and rank 0's
    check of a sample from every rank's block against the C oracle (BBP_BENCH_BACKEND=gloo BBP_BENCH_DEVICE=0: both ranks on
    device 0, rendezvous over gloo; on an 8-GPU node the same code runs over RCCL, one rank per card);
  * two live contexts in ONE process driven concurrently (what a host holding one context per GPU does; also the state the
    per-context dynamic-LDS attribute of the sort kernel exists for);
  * the device pool (bbp_pool_init): the reference's prove() / verify() callers and the batch calls spread over several
    contexts, results in request order.
SURVEY.md 8e; reference concurrency model src/main.rs:55, src/futures/main.rs:46-56."""
import json
import os
import subprocess
import sys
import threading

import pytest

from tests import oracle_c
from tests.test_gpu_prove_verify import _synth_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def oc(built):
    return oracle_c.load(built.build_oracle())


def _bench_two_ranks(args):
    env = dict(os.environ, BBP_BENCH_BACKEND="gloo", BBP_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-build", "--no-also", "--no-cpu-baseline"] + args,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout.decode()[-1500:], p.stderr.decode()[-3000:])
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-1500:]
    return json.loads(lines[0])


def test_two_ranks_prove_records_gathered_and_checked(ctx, built):
    """bench.py --gpus 2, the real engine in both ranks: 2 x 128 proofs, records gathered to rank 0 through
    sharding.gather_records, and rank 0 compares records 0 / 64 / 127 of EACH rank's block with the C oracle's proof of that
    rank's inputs (bench_workloads.ProveWorkload.check_gathered raises SystemExit on any difference -> non-zero exit)."""
    out = _bench_two_ranks(["--batch", "128", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["batch_per_gpu"] == 128
    g = out["gather"]
    assert g["ranks"] == 2 and g["backend"] == "gloo" and g["bytes"] == 2 * 128 * (1121 + 32 * 12)
    assert g["gather_ms"] > 0 and "every rank" in g["checked"]
    assert out["value"] > 0 and abs(out["value"] - 2 * 128 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


def test_two_ranks_verify_flags_gathered_and_checked(ctx, built):
    """configs[3] in miniature: every rank verifies its own 256 proofs (1 % corrupted at known indices), 4-byte flags gathered to
    rank 0, every rank's block checked against the expected pattern."""
    out = _bench_two_ranks(["--workload", "verify", "--batch", "256", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["gather"]["bytes"] == 2 * 256 * 4 and out["gather"]["ranks"] == 2
    assert out["unit"] == "verifies/s"


def test_two_live_contexts_driven_from_two_threads(ctx, oc, bbp):
    """A second context on the same card beside the session's: two host threads, one context each, prove + verify batches at the
    same time (64 proofs of N = 8 and 48 of N = 3, so the two contexts also compile different circuits and launch different MSM
    shapes concurrently).  Every record byte-equal to the C oracle, every verdict right, both contexts healthy afterwards."""
    other = bbp.Context(0)
    try:
        jobs = [(ctx, 64, 8, 777), (other, 48, 3, 778)]
        data = []
        for c, B, N, seed in jobs:
            ins, ents, vins = _synth_batch(ctx, B, N, seed=seed)
            exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
            assert est == [0] * B
            data.append((ins, ents, vins, exp))
        results, errors = [None, None], []
        barrier = threading.Barrier(2)

        def worker(k):
            c, B, N, _ = jobs[k]
            ins, ents, vins, _ = data[k]
            try:
                barrier.wait()
                rounds = []
                for _ in range(3):
                    out, st = c.prove_batch(B, N, b"".join(ins), b"".join(ents))
                    rs_ = bbp.record_size(N)
                    rows = [bytearray(out[i * rs_:(i + 1) * rs_] + b"".join(vins[i])) for i in range(B)]
                    rows[5][300] ^= 0x10
                    vst = c.verify_batch(B, N, b"".join(bytes(r) for r in rows))
                    rounds.append((out, st, vst))
                results[k] = rounds
            except Exception as e:  # noqa: BLE001
                errors.append((k, repr(e)))

        th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for k, (c, B, N, _) in enumerate(jobs):
            for out, st, vst in results[k]:
                assert st == [0] * B
                assert out == data[k][3], "context %d: records differ from the oracle's" % k
                assert [i for i, v in enumerate(vst) if v != 0] == [5], (k, vst)
        assert other.health() == 0
    finally:
        other.close()
