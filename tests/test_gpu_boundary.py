"""GPU tier: the C-ABI boundary under the reference's calling pattern -- many worker threads, one process-wide state
(src/main.rs:55, src/futures/main.rs:46-56: one Proof::prove / Verify::verify per dusk-uds worker thread) -- plus the argument
semantics the device-resident entry points inherit from the reference (u64 toggle compare, canonical-only public inputs) and
the stream contract of include/bbp.h."""
import threading

import pytest

from oracle.ref_py import ristretto as rs
from tests import oracle_c
from tests.test_gpu_prove_verify import _synth_batch

pytestmark = pytest.mark.gpu
L = rs.L


@pytest.fixture(scope="module")
def oc(built):
    return oracle_c.load(built.build_oracle())


def test_eight_threads_share_one_context(ctx, oc, bbp):
    """8 host threads call bbp_prove / bbp_verify on ONE context concurrently (what the reference's worker pool does with its
    process-wide generators): every record byte-equal to the C oracle under its own entropy, every verification correct, and
    the call combiner really merged concurrent callers into shared device batches."""
    N, T, per = 8, 8, 3
    ins, ents, vins = _synth_batch(ctx, T * per, N, seed=2024)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), T * per, N, threads=8)
    assert est == [0] * (T * per)
    rs_ = bbp.record_size(N)
    calls0, reqs0, _ = ctx.batching_stats()
    results, errors = {}, []
    barrier = threading.Barrier(T)

    def worker(t):
        try:
            barrier.wait()
            for j in range(per):
                i = t * per + j
                rec = ctx.prove(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i])
                ok = ctx.verify(rec, *vins[i])
                bad = bytearray(rec)
                bad[200 + i] ^= 0x08
                rej = ctx.verify(bytes(bad), *vins[i])
                results[i] = (rec, ok, rej)
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
    calls, reqs, biggest = ctx.batching_stats()
    assert reqs - reqs0 == T * per * 3
    assert calls - calls0 < reqs - reqs0          # concurrent callers shared device batches ...
    assert biggest >= 2                             # ... of more than one request


def test_threads_mix_batch_and_single_calls(ctx, oc, bbp):
    """Host-pointer batch calls, single calls and setup read-backs from different threads at once: the context lock keeps the
    shared staging buffers consistent (every output equals the sequential answer)."""
    N, B = 3, 40
    ins, ents, vins = _synth_batch(ctx, B, N, seed=77077)
    ref, rst = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert rst == [0] * B
    rs_ = bbp.record_size(N)
    vin = b"".join(ref[i * rs_:(i + 1) * rs_] + b"".join(v) for i, v in enumerate(vins))
    g0 = ctx.generator(7)
    out, errors = {}, []

    def batch_prover():
        try:
            for _ in range(3):
                o, s = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
                assert s == [0] * B and o == ref
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def batch_verifier():
        try:
            for _ in range(4):
                assert ctx.verify_batch(B, N, vin) == [0] * B
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def singles(k):
        try:
            for i in range(k, B, 8):
                assert ctx.prove(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i]) == ref[i * rs_:(i + 1) * rs_]
                assert ctx.verify(ref[i * rs_:(i + 1) * rs_], *vins[i]) == 0
                assert ctx.generator(7) == g0
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=batch_prover), threading.Thread(target=batch_verifier)] + [threading.Thread(target=singles, args=(k,)) for k in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    del out


def test_exception_inside_a_device_call_is_a_status(ctx, bbp, monkeypatch):
    """A C++ exception below the boundary (injected at the first synthesis of a list length, and a bad_alloc in the batch prover)
    comes back as BBP_ERR_INTERNAL with a message; the context stays usable."""
    ins, ents, vins = _synth_batch(ctx, 1, 11, seed=5)                       # N = 11: not compiled by any other test
    monkeypatch.setenv("BBP_FAULT_INJECT", "compile")
    with pytest.raises(bbp.BbpError) as e:
        ctx.prove(ins[0][:224], ins[0][224:224 + 32 * 11], 0, ents[0])
    assert e.value.status == 6 and "injected" in str(e.value)
    monkeypatch.setenv("BBP_FAULT_INJECT", "alloc")
    with pytest.raises(bbp.BbpError) as e:
        ctx.prove_batch(1, 11, ins[0], ents[0])
    assert e.value.status == 6
    monkeypatch.delenv("BBP_FAULT_INJECT")
    rec = ctx.prove(ins[0][:224], ins[0][224:224 + 32 * 11], int.from_bytes(ins[0][-8:], "little"), ents[0])
    assert ctx.verify(rec, *vins[0]) == 0


def test_toggle_is_compared_as_u64_on_the_device_path(ctx, oc, bbp):
    """`x as u64 == toggle` (src/blindbid/proof.rs:63): toggle = 2^32 + 3 sets NO bit.  bbp_prove_batch_dev has no host screening,
    so the kernel itself must compare all 64 bits: the record equals the oracle's for that toggle (a proof of a false statement
    that no verifier accepts), not the record of toggle = 3."""
    import torch
    N = 8
    ins, ents, vins = _synth_batch(ctx, 4, N, seed=404)
    row = bytearray(ins[3])                                                     # its honest toggle is 3
    assert int.from_bytes(row[-8:], "little") == 3
    row[-8:] = (2**32 + 3).to_bytes(8, "little")
    dev = torch.device("cuda", 0)
    rs_ = bbp.record_size(N)
    d_in = torch.frombuffer(bytearray(bytes(row)), dtype=torch.uint8).to(dev)
    d_ent = torch.frombuffer(bytearray(ents[3]), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(rs_, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    ctx.prove_batch_dev(1, N, d_in.data_ptr(), d_ent.data_ptr(), d_out.data_ptr(), s.cuda_stream)
    s.synchronize()
    got = bytes(d_out.cpu().numpy().tobytes())
    # the C oracle screens toggle >= N like the host-pointer API; the big-int oracle follows proof.rs:60-67 literally
    from oracle.ref_py import blindbid as pbb
    f = lambda k: int.from_bytes(bytes(row[32 * k:32 * k + 32]), "little")
    pub_int = [rs.sc_from_bits(bytes(row[224 + 32 * i:256 + 32 * i])) for i in range(N)]
    exp = pbb.prove(f(0), f(1), f(2), f(3), f(4), f(5), f(6), pub_int, 2**32 + 3, ents[3]).to_record()
    assert got == exp
    rc, honest = oc.prove(ins[3][:224], ins[3][224:224 + 32 * N], 3, ents[3])
    assert got != honest
    assert ctx.verify(got, *vins[3]) == 1 and oc.verify(got, *vins[3]) == 1


def test_noncanonical_public_inputs_are_format_errors(ctx, oc, bbp):
    """score, z_img, seed reach Verify::new as serde-deserialised Scalars (src/blindbid/verify.rs:100-104): canonical encodings
    only.  x + l encodes the same residue but is rejected with FormatError on the single, batch and aggregated paths (the
    prove side already did this for its seven scalars)."""
    N = 2
    ins, ents, vins = _synth_batch(ctx, 3, N, seed=606060)
    out, st = ctx.prove_batch(3, N, b"".join(ins), b"".join(ents))
    assert st == [0, 0, 0]
    rs_ = bbp.record_size(N)
    plus_l = lambda b: (int.from_bytes(b, "little") + L).to_bytes(32, "little")
    rec0 = out[:rs_]
    q, z, sd, pub = vins[0]
    assert ctx.verify(rec0, q, z, sd, pub) == 0
    assert ctx.verify(rec0, plus_l(q), z, sd, pub) == 3
    assert ctx.verify(rec0, q, plus_l(z), sd, pub) == 3
    assert ctx.verify(rec0, q, z, plus_l(sd), pub) == 3
    rows = [bytearray(out[i * rs_:(i + 1) * rs_] + b"".join(vins[i])) for i in range(3)]
    rows[1][rs_:rs_ + 32] = plus_l(bytes(rows[1][rs_:rs_ + 32]))              # score + l
    rows[2][rs_ + 64:rs_ + 96] = plus_l(bytes(rows[2][rs_ + 64:rs_ + 96]))    # seed + l
    blob = b"".join(bytes(r) for r in rows)
    assert ctx.verify_batch(3, N, blob) == [0, 3, 3]
    assert ctx.verify_batch_aggregated(3, N, blob, 2)[0] == [0, 3, 3]


def test_null_stream_is_the_callers_default_stream(ctx, oc, bbp):
    """include/bbp.h: a NULL `stream` is the legacy default stream and is honoured as such -- torch's default stream has handle 0,
    so work torch enqueues there after the call sees the records without any host synchronisation; BBP_STREAM_CONTEXT (None in
    the binding) selects the context's own stream instead."""
    import torch
    N, B = 8, 96
    ins, ents, _ = _synth_batch(ctx, B, N, seed=1717)
    dev = torch.device("cuda", 0)
    rs_ = bbp.record_size(N)
    d_in = torch.frombuffer(bytearray(b"".join(ins)), dtype=torch.uint8).to(dev)
    d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    assert torch.cuda.current_stream().cuda_stream == 0
    ctx.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), d_out.data_ptr(), 0)
    snap = d_out.clone()                      # enqueued on the default stream right behind the call, no synchronisation
    d_out.zero_()                             # and this after the clone: the clone must have seen the records
    host = bytes(snap.cpu().numpy().tobytes())
    ref, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert st == [0] * B and host == ref
    ctx.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), d_out.data_ptr())   # context stream: caller synchronises the device
    torch.cuda.synchronize()
    assert bytes(d_out.cpu().numpy().tobytes()) == ref


def test_corrupted_scratch_fails_the_call_instead_of_returning_a_wrong_proof(bbp, ctx, oc):
    """include/bbp.h "never aborts, never lies": an out-of-range entry in the MSM's sorted scratch (the state behind round 1's one
    GPU fault; injected here through bbp_debug_corrupt_scratch) is clamped by the accumulate kernel -- no fault -- and raises the
    context's health bit.  The call whose results are fetched next must come back BBP_ERR_DEVICE as a whole, NOT status 0 with a
    record computed from garbage; the flag is sticky (every later host-pointer call on that context fails the same way, single
    calls through the combiner included) and other contexts are unaffected."""
    N, B = 4, 6
    ins, ents, vins = _synth_batch(ctx, B, N, seed=31415)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=4)
    assert est == [0] * B
    c = bbp.Context(0)
    try:
        out, st = c.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert st == [0] * B and out == exp and c.health() == 0
        c.debug_corrupt_scratch()
        with pytest.raises(bbp.BbpError) as e:
            c.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert e.value.status == 5 and "health" in str(e.value)
        assert c.health() == 1
        with pytest.raises(bbp.BbpError) as e:                       # sticky: the batch API ...
            c.verify_batch(1, N, exp[:bbp.record_size(N)] + b"".join(vins[0]))
        assert e.value.status == 5
        with pytest.raises(bbp.BbpError) as e:                       # ... the single calls through the combiner ...
            c.prove(ins[0][:224], ins[0][224:224 + 32 * N], 0, ents[0])
        assert e.value.status == 5
        assert c.verify(exp[:bbp.record_size(N)], *vins[0]) == 5
        import random
        rnd = random.Random(1)
        sc = b"".join(rnd.getrandbits(252).to_bytes(32, "little") for _ in range(9))
        with pytest.raises(bbp.BbpError) as e:                       # ... and the MSM hook
            c.msm_batch(1, 9, sc, bbp.LAYOUT_BLIND_G_H)
        assert e.value.status == 5
    finally:
        c.close()
    out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))   # the session's context never saw it
    assert st == [0] * B and out == exp and ctx.health() == 0


def test_describe_reports_device_and_warns_about_hardware_queues(ctx, bbp):
    """bbp_describe: the device line is there, and the hardware-queue condition that silently costs up to 30 % (INTEGRATION.md
    section 5) is either satisfied in this process (the binding exports GPU_MAX_HW_QUEUES=16 before HIP initialises) or WARNed about."""
    import os
    text = ctx.describe()
    assert "gfx950" in text and "CUs" in text and "verifier: 4 lanes" in text
    hwq = os.environ.get("GPU_MAX_HW_QUEUES")
    assert ("WARNING: GPU_MAX_HW_QUEUES" in text) == (hwq is None or int(hwq) < 8)
    assert "hardware queues: GPU_MAX_HW_QUEUES=" in text and "allocation(s) since bbp_init" in text


_HWQ_PROBE = r"""
import ctypes, os, sys
pre = sys.argv[2]
if pre == "hip_first":  # some other component of the host process touched HIP before the engine was created
    hip = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
    n = ctypes.c_int(0)
    assert hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 1
L = ctypes.CDLL(sys.argv[1])
h = ctypes.c_void_p()
L.bbp_init.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
assert L.bbp_init(0, ctypes.byref(h)) == 0
buf = ctypes.create_string_buffer(8192)
L.bbp_describe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32]
assert L.bbp_describe(h, buf, 8192) == 0
libc = ctypes.CDLL(None)
libc.getenv.restype = ctypes.c_char_p
print("ENV", libc.getenv(b"GPU_MAX_HW_QUEUES"))
print(buf.value.decode())
L.bbp_free.argtypes = [ctypes.c_void_p]
L.bbp_free(h)
"""


def test_library_owns_the_hardware_queue_setting(bbp):
    """a Rust / Go / C host bound per INTEGRATION.md knows nothing about GPU_MAX_HW_QUEUES.  bbp_init exports
    it itself when the process has not initialised HIP yet, leaves a caller's own value alone, and WARNs through bbp_describe when
    something else initialised HIP first with the variable unset.  Each case is a fresh process that loads only libbbp_hip.so."""
    import os
    import subprocess
    import sys

    def probe(pre, env_extra):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        env.update(env_extra)
        p = subprocess.run([sys.executable, "-c", _HWQ_PROBE, bbp.lib_path, pre], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        return p.stdout
    out = probe("engine_first", {})
    assert "ENV b'16'" in out and "exported by the library before HIP initialised" in out and "WARNING: GPU_MAX_HW_QUEUES" not in out, out
    out = probe("engine_first", {"GPU_MAX_HW_QUEUES": "8"})
    assert "ENV b'8'" in out and "from the caller's environment" in out and "WARNING: GPU_MAX_HW_QUEUES" not in out, out
    out = probe("hip_first", {})
    assert "ENV None" in out and "HIP was initialised before bbp_init could set it" in out, out


def _allocs(c):
    import re
    m = re.search(r"in (\d+) allocation\(s\) since bbp_init", c.describe())
    assert m
    return int(m.group(1))


def test_reserve_below_1024_leaves_nothing_to_allocate(bbp, oc):
    """bbp_reserve(max_batch < 1024) used to warm three of the five rotating prover buffers; the 4th / 5th call under
    load then grew its buffer with hipFree + hipMalloc (a 0.1-1 s stall for everybody).  After the reservation, seven consecutive
    prove calls and five verify calls of that size -- every buffer of the rotation, every staging slot, every verifier lane --
    perform no scratch allocation at all; the records are still the oracle's."""
    N, B = 8, 192
    c = bbp.Context(0)
    try:
        c.reserve(B, N)
        ins, ents, vins = _synth_batch(c, B, N, seed=4242)  # (bbp_witness_batch has staging of its own: not what is counted here)
        a0 = _allocs(c)
        assert a0 > 0
        exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
        for k in range(7):
            out, st = c.prove_batch(B, N, b"".join(ins), b"".join(ents))
            assert st == [0] * B and out == exp, k
        rs_ = bbp.record_size(N)
        vin = b"".join(exp[i * rs_:(i + 1) * rs_] + b"".join(vins[i]) for i in range(B))
        for k in range(5):
            assert c.verify_batch(B, N, vin) == [0] * B
        assert _allocs(c) == a0, (a0, _allocs(c))
    finally:
        c.close()
