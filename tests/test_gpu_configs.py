"""GPU parity for the BASELINE.json configurations and the SURVEY.md 8a edge that round 1 left without a `-m gpu` test:
  configs[3]  one GPU's shard of the 65 536-proof verification (8192 verifications, 1 % corrupted at known indices),
  configs[4]  streaming prove -> verify over many chunks with no host synchronisation in between,
  row a9      bid lists holding non-canonical 32-byte values (Scalar::from_bits, src/blindbid/bid.rs:20-29, verify.rs:112-116).
Everything goes through the C-ABI (libbbp_hip.so); the C / Python oracles are the checkers."""
import hashlib
import json
import os

import pytest

from oracle.ref_py import ristretto as rs
from tests import oracle_c
from tests.test_gpu_prove_verify import _synth_batch

pytestmark = pytest.mark.gpu
L = rs.L


@pytest.fixture(scope="module")
def oc(built):
    return oracle_c.load(built.build_oracle())


def test_config3_shard_8192_verifications(ctx, oc, bbp):
    """BASELINE.json configs[3], one GPU's share: 8192 verifications = 128 distinct device-made proofs tiled 64x, 1 % corrupted at
    known indices (proof bytes, public inputs, non-canonical scalars, undecodable points), through bbp_verify_batch AND
    bbp_verify_batch_aggregated; statuses == expected; a 64-row sample (corrupted rows included) also through the C oracle."""
    N, distinct, B = 8, 128, 8192
    ins, ents, vins = _synth_batch(ctx, distinct, N, seed=31337)
    out, st = ctx.prove_batch(distinct, N, b"".join(ins), b"".join(ents))
    assert st == [0] * distinct
    rs_ = bbp.record_size(N)
    rows = [bytearray(out[(i % distinct) * rs_:(i % distinct + 1) * rs_] + b"".join(vins[i % distinct])) for i in range(B)]
    bad = sorted({(i * 101 + 7) % B for i in range(82)})
    assert len(bad) == 82
    fmt = set()
    for j, i in enumerate(bad):
        kind = j % 4
        if kind == 0:
            rows[i][1 + (i * 37) % 1100] ^= 1 << (i % 8)             # a bit somewhere in the R1CSProof
        elif kind == 1:
            rows[i][rs_ + 32 + (i % 32)] ^= 0x04                      # z_img (public input)
        elif kind == 2:
            rows[i][1 + 32 * 9:1 + 32 * 10] = b"\xff" * 32            # t_x_blinding non-canonical -> FormatError
            fmt.add(i)
        else:
            rows[i][rs_ - 32 * (4 + N):rs_ - 32 * (3 + N)] = b"\xff" * 32  # commitment V_0 is not a ristretto encoding
    blob = b"".join(bytes(r) for r in rows)
    plain = ctx.verify_batch(B, N, blob)
    assert [i for i, s in enumerate(plain) if s != 0] == bad
    for i in bad:
        assert plain[i] == (3 if i in fmt else plain[i]) and plain[i] in (1, 3), i
    agg, nfb = ctx.verify_batch_aggregated(B, N, blob, 0)
    assert agg == plain
    assert 0 < nfb <= 32 * len(bad)
    # oracle on a sample: the first 24 corrupted rows and 40 honest ones spread over the shard
    sample = bad[:24] + [i for i in range(0, B, B // 44) if i not in bad][:40]
    stride = rs_ + 96 + 32 * N
    ost = oc.verify_many(b"".join(blob[i * stride:(i + 1) * stride] for i in sample), len(sample), N, threads=8)
    for i, o in zip(sample, ost):
        assert (o == 0) == (plain[i] == 0), i
        if i in fmt:
            assert o == 3, i


def test_config3_flags_gather_layout(ctx, bbp):
    """The shard's flags as the one collective of the path carries them (sharding.gather_records on a 1-rank group is the
    identity): int32 statuses viewed as 4-byte records keep their order."""
    import torch
    from dusk_blindbidproof_amd import sharding
    st = torch.arange(8192, dtype=torch.int32, device="cuda")
    got = sharding.gather_records(None, st.view(torch.uint8), 4, 8192, 0, 1)
    assert got.view(torch.int32).cpu().tolist() == list(range(8192))


def _stream_bytes(tag, n):
    return hashlib.shake_256(b"bbp-stream-test/" + tag).digest(n)


def test_config4_streaming_prove_verify_no_host_sync(ctx, oc, bbp):
    """BASELINE.json configs[4] shape: 16 chunks x 1024 bids streamed bids -> witness -> prove -> verify entirely on the device,
    with NO host synchronisation between chunks (device-resident APIs on one caller stream, the bid pass on a second one).  Zero
    failed verifications over all 16 384; ALL 1024 records of the last chunk byte-equal to the C oracle under the same entropy."""
    import torch
    dev = torch.device("cuda", 0)
    N, C, n_chunks = 8, 1024, 16
    total = C * n_chunks
    seed_sc = rs.sc_bytes(rs.sc_wide(_stream_bytes(b"seed", 64)))
    dk = _stream_bytes(b"dk", 40 * total)
    bids = bytearray()
    for i in range(total):
        k = bytearray(dk[40 * i + 8:40 * i + 40])
        k[31] &= 0x0f                                              # < 2^252 < l: canonical
        bids += dk[40 * i:40 * i + 8] + bytes(24) + bytes(k) + seed_sc
    lists = bytearray(_stream_bytes(b"lists", 32 * N * total))
    for j in range(31, len(lists), 32):
        lists[j] &= 0x0f
    toggles = [i % N for i in range(total)]
    ent = bytearray(_stream_bytes(b"ent", bbp.entropy_size(N) * total))
    es = bbp.entropy_size(N)
    for i in range(total):
        for b in range(4 + N):
            ent[es * i + 32 * b + 31] &= 0x0f                     # canonical blindings
    d_bids = torch.frombuffer(bids, dtype=torch.uint8).to(dev)
    d_lists = torch.frombuffer(lists, dtype=torch.uint8).to(dev)
    d_tog = torch.tensor(toggles, dtype=torch.int64, device=dev)
    d_ent = torch.frombuffer(ent, dtype=torch.uint8).to(dev)
    in_stride, rs_, vt = 224 + 32 * N + 8, bbp.record_size(N), 96 + 32 * N
    d_in = [torch.zeros(C * in_stride, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_vt = [torch.zeros(C * vt, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_out = torch.zeros((n_chunks, C * rs_), dtype=torch.uint8, device=dev)
    vin = [torch.empty((C, rs_ + vt), dtype=torch.uint8, device=dev) for _ in range(2)]
    vent = torch.frombuffer(bytearray(_stream_bytes(b"vent", 32 * total)), dtype=torch.uint8).to(dev)
    status = torch.full((total,), -1, dtype=torch.int32, device=dev)
    keep_in = torch.zeros(C * in_stride, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    prep, main = torch.cuda.Stream(), torch.cuda.Stream()
    done = [None, None]
    for c in range(n_chunks):
        b = c & 1
        sl = slice(c * C, (c + 1) * C)
        if done[b] is not None:
            prep.wait_event(done[b])                               # buffer b was last read by chunk c - 2's prove / verify
        ctx.prepare_bids_dev(C, N, d_bids[96 * c * C:].data_ptr(), d_lists[32 * N * c * C:].data_ptr(), d_tog[sl].data_ptr(),
                             d_in[b].data_ptr(), d_vt[b].data_ptr(), prep.cuda_stream)
        with torch.cuda.stream(main):
            ctx.prove_batch_dev(C, N, d_in[b].data_ptr(), d_ent[es * c * C:].data_ptr(), d_out[c].data_ptr(), main.cuda_stream)
            main.wait_stream(prep)
            vin[b][:, :rs_] = d_out[c].view(C, rs_)
            vin[b][:, rs_:] = d_vt[b].view(C, vt)
            ctx.verify_batch_dev(C, N, vin[b].data_ptr(), vent[32 * c * C:].data_ptr(), status[sl].data_ptr(), main.cuda_stream)
            if c == n_chunks - 1:
                keep_in.copy_(d_in[b])
            done[b] = torch.cuda.Event()
            done[b].record(main)
    torch.cuda.synchronize()                                       # the only host synchronisation of the stream
    st = status.cpu().tolist()
    assert st == [0] * total, [i for i, s in enumerate(st) if s != 0][:10]
    last_in = bytes(keep_in.cpu().numpy().tobytes())
    last_out = bytes(d_out[n_chunks - 1].cpu().numpy().tobytes())
    # the WHOLE last chunk against the oracle (round 4; two records before): 1024 records from the device-made input rows and the same entropy
    g0 = (n_chunks - 1) * C
    exp, est = oc.prove_many(last_in, bytes(ent[es * g0:es * (g0 + C)]), C, N, threads=os.cpu_count() or 8)
    assert est == [0] * C
    bad = [r for r in range(C) if last_out[r * rs_:(r + 1) * rs_] != exp[r * rs_:(r + 1) * rs_]]
    assert not bad, bad[:10]
    # and an earlier chunk did not get overwritten by a later one (double-buffered inputs, per-chunk outputs)
    first = bytes(d_out[0][:rs_].cpu().numpy().tobytes())
    w = ctx.witness_batch(bytes(bids[:96]))
    x0 = w[32:64]
    pub0 = bytearray(lists[:32 * N])
    pub0[0:32] = x0
    assert oc.verify(first, w[128:160], w[160:192], seed_sc, bytes(pub0)) == 0


def test_aggregated_verification_is_stream_ordered(ctx, bbp):
    """bbp_verify_batch_aggregated_dev with n_fallback = NULL makes no host round trip: which groups failed is decided on the
    device and the per-proof pass sizes itself from a device counter.  Six calls back to back on one stream over different
    corruption patterns (none, a few, a whole group, most of the batch), status buffers read only after ONE synchronisation at
    the end; every call must report exactly what the per-proof path reports."""
    import torch
    dev = torch.device("cuda", 0)
    N, distinct, B = 8, 48, 700
    ins, ents, vins = _synth_batch(ctx, distinct, N, seed=161803)
    out, st = ctx.prove_batch(distinct, N, b"".join(ins), b"".join(ents))
    assert st == [0] * distinct
    rs_ = bbp.record_size(N)
    base = [out[(i % distinct) * rs_:(i % distinct + 1) * rs_] + b"".join(vins[i % distinct]) for i in range(B)]
    patterns = [set(), {5, 6, 7, 300}, set(range(64, 96)), {B - 1}, set(range(0, B, 2)), set()]
    blobs, d_in, d_st = [], [], []
    for bad in patterns:
        rows = [bytearray(r) for r in base]
        for i in bad:
            rows[i][1 + (i * 13) % 1000] ^= 1 << (i % 7)
        blobs.append(b"".join(bytes(r) for r in rows))
        d_in.append(torch.frombuffer(bytearray(blobs[-1]), dtype=torch.uint8).to(dev))
        d_st.append(torch.full((B,), -1, dtype=torch.int32, device=dev))
    d_ent = torch.zeros(32 * B, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for k in range(len(patterns)):
            assert ctx.verify_batch_aggregated_dev(B, N, d_in[k].data_ptr(), d_ent.data_ptr(), d_st[k].data_ptr(), 32, s.cuda_stream, want_count=False) is None
    s.synchronize()
    for k, bad in enumerate(patterns):
        got = d_st[k].cpu().tolist()
        assert got == ctx.verify_batch(B, N, blobs[k]), k
        assert {i for i, v in enumerate(got) if v != 0} == bad, k


def test_verifier_lanes_share_nothing(ctx, bbp):
    """bbp_context_verify_stream(lane): calls on different lanes' streams run concurrently on the device (own batch buffer, scratch,
    MSM scratch slot and aggregation buffers each; four lanes since round 3).  Twelve calls rotate over the lanes with no
    synchronisation in between, per-proof and aggregated mixed, every call over a different corruption pattern and batch size:
    each must report exactly its own pattern."""
    import torch
    dev = torch.device("cuda", 0)
    N, distinct = 8, 40
    ins, ents, vins = _synth_batch(ctx, distinct, N, seed=271828)
    out, st = ctx.prove_batch(distinct, N, b"".join(ins), b"".join(ents))
    assert st == [0] * distinct
    rs_ = bbp.record_size(N)
    n_lanes = 0
    while ctx.verify_stream(n_lanes):
        n_lanes += 1
    assert n_lanes == 4 and ctx.verify_stream(n_lanes) is None
    lanes = [torch.cuda.ExternalStream(ctx.verify_stream(i), device=dev) for i in range(n_lanes)]
    assert len({x.cuda_stream for x in lanes}) == n_lanes
    calls = []
    for k in range(12):
        B = (300, 517, 64, 1000)[k % 4]
        bad = {(7 * k + 11 * j) % B for j in range(k)}          # call 0 is all honest
        rows = [bytearray(out[(i % distinct) * rs_:(i % distinct + 1) * rs_] + b"".join(vins[i % distinct])) for i in range(B)]
        for i in bad:
            rows[i][1 + 32 * 3 + (i % 200)] ^= 0x40
        d_in = torch.frombuffer(bytearray(b"".join(bytes(r) for r in rows)), dtype=torch.uint8).to(dev)
        calls.append((B, bad, d_in, torch.zeros(32 * B, dtype=torch.uint8, device=dev), torch.full((B,), -1, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    for k, (B, bad, d_in, d_ent, d_st) in enumerate(calls):
        s = lanes[k % n_lanes].cuda_stream
        if k % 3 == 2:
            ctx.verify_batch_aggregated_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), d_st.data_ptr(), 16, s, want_count=False)
        else:
            ctx.verify_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    for k, (B, bad, _, _, d_st) in enumerate(calls):
        got = d_st.cpu().tolist()
        assert {i for i, v in enumerate(got) if v != 0} == bad, k
        assert all(v in (0, 1, 3) for v in got), k


# ---- a9: Scalar::from_bits semantics on the device path ------------------------------------------------------------------------
NONCANON = [L, L + 1, 2**255 - 1, 2**255 + 12345, 2**256 - 1, 2**255 + L + 7]


def test_a9_noncanonical_bids_golden(ctx, oc, golden):
    """Committed fixture made by the big-int oracle (tests/golden/make_golden.py --noncanonical): a list holding l, l + 1,
    2^255 - 1 and values with bit 255 set, handed over as RAW bytes.  Device prover == fixture bytes; device, C and Python
    verifiers accept; the reduced form of the same list is the same statement (same proof verifies)."""
    from oracle.ref_py import blindbid as pbb
    c = golden("proofs_noncanonical.json")["noncanonical"][0]
    s7 = b"".join(bytes.fromhex(c[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])
    raw = b"".join(bytes.fromhex(p) for p in c["pub_list"])
    ent = bytes.fromhex(c["entropy"])
    rec = ctx.prove(s7, raw, c["toggle"], ent)
    assert rec.hex() == c["record"]
    rc, crec = oc.prove(s7, raw, c["toggle"], ent)
    assert rc == 0 and crec == rec
    args = (bytes.fromhex(c["q"]), bytes.fromhex(c["z_img"]), bytes.fromhex(c["seed"]))
    assert ctx.verify(rec, *args, raw) == 0 and oc.verify(rec, *args, raw) == 0
    reduced = b"".join(rs.sc_bytes(rs.sc_from_bits(raw[32 * i:32 * i + 32])) for i in range(c["N"]))
    assert reduced != raw
    assert ctx.verify(rec, *args, reduced) == 0 and oc.verify(rec, *args, reduced) == 0
    f = lambda k: int.from_bytes(bytes.fromhex(c[k]), "little")
    pub_int = [rs.sc_from_bits(raw[32 * i:32 * i + 32]) for i in range(c["N"])]
    assert pbb.verify(pbb.Proof.from_record(rec, c["N"]), f("q"), f("z_img"), f("seed"), pub_int)
    # bit 255 is CLEARED, not an error and not part of the value: flipping it on any entry changes nothing
    flipped = bytearray(raw)
    for i in range(c["N"]):
        flipped[32 * i + 31] ^= 0x80
    assert ctx.verify(rec, *args, bytes(flipped)) == 0
    # but the low 255 bits matter: l + 1 is not l
    other = bytearray(raw)
    other[0] ^= 1
    assert ctx.verify(rec, *args, bytes(other)) == 1


@pytest.mark.parametrize("N,B", [(6, 5), (2, 3)])
def test_a9_noncanonical_bids_batch_paths(ctx, oc, bbp, N, B):
    """The same edge through the batch entry points (host-pointer and device-resident): every record byte-equal to the C oracle,
    every verification accepted by device and oracle, with a different non-canonical value next to each bid."""
    import torch
    ins, ents, vins = _synth_batch(ctx, B, N, seed=900 + N)
    rows, vrows = [], []
    for i in range(B):
        row = bytearray(ins[i])
        toggle = int.from_bytes(row[-8:], "little")
        pub = bytearray(vins[i][3])
        for j in range(N):
            if j != toggle:
                pub[32 * j:32 * j + 32] = NONCANON[(i + j) % len(NONCANON)].to_bytes(32, "little")
        row[224:224 + 32 * N] = pub
        rows.append(bytes(row))
        vrows.append((vins[i][0], vins[i][1], vins[i][2], bytes(pub)))
    out, st = ctx.prove_batch(B, N, b"".join(rows), b"".join(ents))
    assert st == [0] * B
    cout, cst = oc.prove_many(b"".join(rows), b"".join(ents), B, N, threads=4)
    assert cst == [0] * B and cout == out
    rs_ = bbp.record_size(N)
    vin = b"".join(out[i * rs_:(i + 1) * rs_] + b"".join(v) for i, v in enumerate(vrows))
    assert ctx.verify_batch(B, N, vin) == [0] * B
    assert oc.verify_many(vin, B, N, threads=4) == [0] * B
    got, _ = ctx.verify_batch_aggregated(B, N, vin, 2)
    assert got == [0] * B
    # device-resident path: no host screening at all between the raw bytes and the kernels
    dev = torch.device("cuda", 0)
    d_in = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(dev)
    d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        ctx.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), d_out.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert bytes(d_out.cpu().numpy().tobytes()) == out


def test_config4_one_gpu_share_streams_125k_bids_without_a_failed_verdict(built):
    """BASELINE.json configs[4] at one GPU's share of its stated volume: 1 M bids over 8 GPUs = 125 k per GPU.  `bench.py --workload
    stream` pushes 123 chunks of 1024 bids (+ 2 warm-up chunks: 128 000 bids) through H2D -> witness -> prove -> verify -> D2H with
    three chunks in flight; every verdict must be OK (the workload itself aborts on a failed one and checks a chunk against the
    oracle), the line must say so, and the latency figures must be present.  ~10 s of GPU time."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "stream", "--steps", "123", "--warmup", "2", "--no-build", "--no-also",
                        "--no-cpu-baseline", "--no-exclusive"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["failed_verifications"] == 0 and d["chunks"] >= 123 and d["steps"] == 123
    assert d["value"] > 1000 and d["unit"] == "proofs/s" and "stream" in d["config"]["workload"]
