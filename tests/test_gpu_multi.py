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
    if p.returncode != 0:  # keep the ranks' whole stderr where gpurun brings it back
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "two_ranks_failure.err"), "wb") as f:
            f.write(p.stderr)
    err = [ln for ln in p.stderr.decode().splitlines() if "Gloo" not in ln and "amdgpu.ids" not in ln and "hostname of the client" not in ln]
    assert p.returncode == 0, (p.stdout.decode()[-1500:], "\n".join(err)[-3000:])
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


# ---- device pool (include/bbp.h "Device pool"; csrc/pool.cpp) ---------------------------------------------------------------------

@pytest.fixture(scope="module")
def pool(bbp, ctx):
    p = bbp.Pool([0, 0])  # two member contexts on the one card of a GPU box
    yield p
    flags = p.health()
    p.close()
    assert flags == 0


def test_pool_batch_calls_split_by_index_and_keep_order(pool, ctx, oc, bbp):
    """bbp_prove_batch / bbp_verify_batch[_aggregated] / bbp_msm_batch on a pool handle: contiguous block split over the two
    members (ragged: 37 = 19 + 18), records byte-equal to the C oracle IN REQUEST ORDER, per-item statuses at the right indices
    (a non-canonical input and a toggle >= N in different blocks), verification verdicts at the right indices."""
    N, B = 8, 37
    ins, ents, vins = _synth_batch(ctx, B, N, seed=4242)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
    assert est == [0] * B
    rs_ = bbp.record_size(N)
    bad_in = [bytearray(x) for x in ins]
    bad_in[3][32:64] = b"\xff" * 32                                    # k >= l in member 0's block -> FormatError for that item only
    bad_in[30][-8:] = (N).to_bytes(8, "little")                        # toggle >= N in member 1's block -> BAD_ARG for that item only
    out, st = pool.prove_batch(B, N, b"".join(bytes(x) for x in bad_in), b"".join(ents))
    assert [i for i, s in enumerate(st) if s != 0] == [3, 30] and st[3] == 3 and st[30] == 4
    for i in range(B):
        if i not in (3, 30):
            assert out[i * rs_:(i + 1) * rs_] == exp[i * rs_:(i + 1) * rs_], i
    out, st = pool.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert st == [0] * B and out == exp
    rows = [bytearray(out[i * rs_:(i + 1) * rs_] + b"".join(vins[i])) for i in range(B)]
    rows[2][400] ^= 1
    rows[25][rs_ + 5] ^= 1
    rows[36][1 + 32 * 9:1 + 32 * 10] = b"\xff" * 32
    blob = b"".join(bytes(r) for r in rows)
    vst = pool.verify_batch(B, N, blob)
    assert [i for i, s in enumerate(vst) if s != 0] == [2, 25, 36] and vst[36] == 3
    assert vst == ctx.verify_batch(B, N, blob)
    agg, nfb = pool.verify_batch_aggregated(B, N, blob, 4)
    assert agg == vst and nfb >= 3
    # the MSM hook splits the same way
    import random
    rnd = random.Random(9)
    n_terms = 129
    sc = b"".join(rnd.getrandbits(252).to_bytes(32, "little") for _ in range(5 * n_terms))
    assert pool.msm_batch(5, n_terms, sc, bbp.LAYOUT_BLIND_G_H) == ctx.msm_batch(5, n_terms, sc, bbp.LAYOUT_BLIND_G_H)
    # one item, zero items, and a batch smaller than the pool
    o1, s1 = pool.prove_batch(1, N, ins[0], ents[0])
    assert s1 == [0] and o1 == exp[:rs_]
    assert pool.verify_batch(1, N, bytes(rows[0])) == [0]


def test_pool_single_calls_are_dealt_to_both_members(pool, ctx, oc, bbp):
    """The reference's calling pattern on a multi-GPU node: worker threads call prove() / verify() on ONE shared handle
    (src/main.rs:55, src/futures/main.rs:46-56).  12 threads x 4 ops on the pool: every record byte-equal to the C oracle under
    its own entropy, every verdict right, and BOTH members ran combined device calls."""
    N, T, per = 8, 12, 4
    ins, ents, vins = _synth_batch(ctx, T * per, N, seed=515)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), T * per, N, threads=8)
    assert est == [0] * (T * per)
    rs_ = bbp.record_size(N)
    results, errors = {}, []
    barrier = threading.Barrier(T)

    def worker(t):
        try:
            barrier.wait()
            for j in range(per):
                i = t * per + j
                rec = pool.prove(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i])
                ok = pool.verify(rec, *vins[i])
                bad = bytearray(rec)
                bad[150 + i] ^= 0x02
                results[i] = (rec, ok, pool.verify(bytes(bad), *vins[i]))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for i in range(T * per):
        rec, ok, rej = results[i]
        assert rec == exp[i * rs_:(i + 1) * rs_], i
        assert ok == 0 and rej in (1, 3), i
    calls, reqs, _ = pool.batching_stats()
    per_member = [pool.member_stats(i) for i in range(len(pool))]
    assert len(pool) == 2 and reqs == T * per * 3 and calls < reqs
    assert sum(q for _, q in per_member) == reqs and all(c > 0 for c, _ in per_member), per_member
    assert pool.member(0).batching_stats()[:2] == per_member[0]


def test_pool_refuses_device_pointer_calls(pool, bbp):
    import torch
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(bbp.BbpError) as e:
        pool.prove_batch_dev(1, 8, buf.data_ptr(), buf.data_ptr(), buf.data_ptr())
    assert e.value.status == 4 and "pool" in str(e.value)
    assert pool.stream is None and pool.member(0).stream is not None
    # a member is an ordinary context: the setup read-backs agree with the pool's (served by member 0)
    assert pool.generator(bbp.BASE_G0 + 5) == pool.member(1).generator(bbp.BASE_G0 + 5)


def test_async_calls_complete_through_callbacks(ctx, pool, oc, bbp):
    """bbp_prove_async / bbp_verify_async (what the epoll server and a Rust Future use): 24 requests queued from ONE thread
    without waiting, on a context and on the pool; callbacks deliver records byte-equal to the oracle and the right verdicts;
    a request the host-side structural parse already decides returns its status at once and never calls back."""
    N, B = 8, 24
    ins, ents, vins = _synth_batch(ctx, B, N, seed=6001)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
    rs_ = bbp.record_size(N)
    for h in (ctx, pool):
        got, keep, ev = {}, [], threading.Event()
        lock = threading.Lock()

        def on_proof(i):
            def cb(status, rec):
                with lock:
                    got[i] = (status, rec)
                    if len(got) == B:
                        ev.set()
            return cb
        for i in range(B):
            keep.append(h.prove_async(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i], on_proof(i)))
        assert ev.wait(120)
        for i in range(B):
            assert got[i] == (0, exp[i * rs_:(i + 1) * rs_]), i
        verdicts, ev2 = {}, threading.Event()

        def on_verdict(i):
            def cb(status):
                with lock:
                    verdicts[i] = status
                    if len(verdicts) == B:
                        ev2.set()
            return cb
        for i in range(B):
            rec = bytearray(got[i][1])
            if i % 5 == 0:
                rec[600] ^= 1                       # rejected on the device
            if i == 7:
                rec = rec[:-32]                     # wrong length: decided by the host-side parse, status returned at once
            keep.append(h.verify_async(bytes(rec), *vins[i], on_verdict(i)))
        assert ev2.wait(120)
        assert [verdicts[i] for i in range(B)] == [3 if i == 7 else (1 if i % 5 == 0 else 0) for i in range(B)]


@pytest.fixture(scope="module")
def pool_server(built, bbp):
    import signal
    import tempfile
    import time
    built.build_server()
    d = tempfile.mkdtemp(prefix="bbp-uds-pool-")
    path = os.path.join(d, "sock")
    err = open(os.path.join(d, "log"), "w+")
    p = subprocess.Popen([built.SERVER_BIN, "-b", path, "-l", "info", "--engine", bbp.lib_path, "--devices", "0,0", "--window-us", "500"], stderr=err)
    for _ in range(3000):
        if os.path.exists(path) or p.poll() is not None:
            break
        time.sleep(0.02)
    assert os.path.exists(path), "server did not bind: " + open(err.name).read()[-800:]
    yield {"path": path, "proc": p, "log": err}
    if p.poll() is None:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=60)


def test_uds_server_on_a_two_member_pool(ctx, oc, pool_server):
    """bbp-uds-server --devices 0,0 with the REAL engine: the epoll front end feeds ONE pool handle through bbp_prove_async /
    bbp_verify_async; 16 concurrent connections x 3 ops; every proof that comes back is accepted by the C oracle and by opcode 2,
    tampered ones are refused; the shutdown log shows both device contexts ran device calls."""
    import re
    import signal
    from tests import uds_client as uc
    N, T, per = 8, 16, 3
    ins, _, vins = _synth_batch(ctx, T * per, N, seed=9099)
    out, errors = {}, []

    def worker(t):
        try:
            for j in range(per):
                i = t * per + j
                blob = uc.prove(pool_server["path"], ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"))
                ok = uc.verify(pool_server["path"], blob, *vins[i])
                bad = bytearray(blob)
                bad[300 + i] ^= 0x04
                out[i] = (blob, ok, uc.verify(pool_server["path"], bytes(bad), *vins[i]))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
    for i in range(T * per):
        blob, ok, rej = out[i]
        proof, c, t = uc.decode_proof(blob)
        assert oc.verify(proof + b"".join(c) + b"".join(t), *vins[i]) == 0, i
        assert ok == b"\x01" and rej == b"\x00", i
    pool_server["proc"].send_signal(signal.SIGTERM)
    pool_server["proc"].wait(timeout=60)
    pool_server["log"].seek(0)
    log = pool_server["log"].read()
    assert "2 device context(s)" in log
    per_dev = re.findall(r"device context (\d) \(device 0\): (\d+) device calls, (\d+) requests", log)
    assert len(per_dev) == 2 and all(int(c) > 0 for _, c, _ in per_dev), log[-1200:]
    assert sum(int(q) for _, _, q in per_dev) == T * per * 3
    assert "ERROR" not in log, log[-1200:]


def test_reserve_sizes_buffers_and_changes_no_result(pool, ctx, oc, bbp):
    """bbp_reserve on a context and on a pool (every member): dummy batches size every per-batch buffer; proofs made afterwards
    are the oracle's byte for byte, and a second reserve of the same size is a no-op that still succeeds."""
    N, B = 5, 96
    for h in (ctx, pool):
        h.reserve(B, N)
        h.reserve(B, N)
    ins, ents, vins = _synth_batch(ctx, B, N, seed=8642)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
    assert est == [0] * B
    for h in (ctx, pool):
        out, st = h.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert st == [0] * B and out == exp
    with pytest.raises(bbp.BbpError) as e:
        ctx.reserve(16, 0)
    assert e.value.status == 4
    with pytest.raises(bbp.BbpError) as e:
        pool.reserve(16, 203)
    assert e.value.status == 2


def test_free_runs_the_asynchronous_requests_still_queued(bbp, ctx, oc):
    """bbp_free with asynchronous requests still in the combiner's queue: they are run first -- every callback fires with the
    oracle's record -- and only then is the device state taken away (include/bbp.h)."""
    N, B = 3, 10
    ins, ents, vins = _synth_batch(ctx, B, N, seed=777001)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=4)
    rs_ = bbp.record_size(N)
    c = bbp.Context(0)
    c.set_batching(20000, 0)   # a 20 ms window: the requests are still queued when close() is called
    got, keep = {}, []
    for i in range(B):
        keep.append(c.prove_async(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i],
                                  (lambda i: lambda status, rec: got.__setitem__(i, (status, rec)))(i)))
    c.close()
    assert len(got) == B
    for i in range(B):
        assert got[i] == (0, exp[i * rs_:(i + 1) * rs_]), i
