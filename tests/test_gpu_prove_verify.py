"""GPU parity (through the C-ABI): full blind-bid prove / verify on the device against the golden fixtures (big-int oracle,
fixed entropy) and the C oracle -- bytes identical, challenges identical, cross-acceptance both ways, tamper rejection."""
import hashlib
import random

import pytest

from oracle.ref_py import ristretto as rs
from tests import oracle_c

pytestmark = pytest.mark.gpu
L = rs.L
b32 = lambda x: x.to_bytes(32, "little")


@pytest.fixture(scope="module")
def oc(built):
    return oracle_c.load(built.build_oracle())


def _s7(c):
    return b"".join(bytes.fromhex(c[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])


def _pub(c):
    return b"".join(bytes.fromhex(p) for p in c["pub_list"])


def test_witness_batch_vs_oracle(ctx, oc):
    rnd = random.Random(11)
    rows = [b32(1) + b32(2) + b32(3)] + [b32(rnd.getrandbits(64)) + b32(rnd.randrange(L)) + b32(rnd.randrange(L)) for _ in range(70)]
    out = ctx.witness_batch(b"".join(rows))
    assert out[:32].hex() == "e9c12933df0565e65eabf6436296300b1a8f9eb7355ebc1af7d0661cd8f23805"  # SURVEY.md App. B witness KAT (m)
    for i, r in enumerate(rows):
        assert out[192 * i:192 * i + 192] == oc.witness(r), i


@pytest.mark.parametrize("idx", [0, 1])
def test_prove_matches_golden_bytes(ctx, golden, idx):
    """Fixed entropy -> byte-identical record, and the device's Fiat-Shamir challenges equal the oracle's trace."""
    c = golden("proofs_full.json")["full"][idx]
    rec = ctx.prove(_s7(c), _pub(c), c["toggle"], bytes.fromhex(c["entropy"]))
    ch = ctx.debug_challenges(1, c["N"], 0)
    for k in ["y", "z", "u", "x", "w"]:
        assert ch[k] == c["trace"][k], k
    assert [ch["t%d" % i] for i in range(1, 7)] == c["trace"]["t_coeffs"]
    assert rec.hex() == c["record"]


def test_verify_golden_and_tamper(ctx, golden):
    for c in golden("proofs_full.json")["full"]:
        rec = bytes.fromhex(c["record"])
        args = (bytes.fromhex(c["q"]), bytes.fromhex(c["z_img"]), bytes.fromhex(c["seed"]), _pub(c))
        assert ctx.verify(rec, *args) == 0
        assert ctx.verify(rec, bytes.fromhex(c["z_img"]), *args[1:]) == 1            # wrong score
        assert ctx.verify(rec, args[0], bytes.fromhex(c["q"]), *args[2:]) == 1       # wrong z_img
        assert ctx.verify(rec, args[0], args[1], bytes.fromhex(c["q"]), args[3]) == 1  # wrong seed
        pub = bytearray(args[3])
        pub[32 * c["toggle"]] ^= 1
        assert ctx.verify(rec, args[0], args[1], args[2], bytes(pub)) == 1           # x not in list
        for pos in (1, 40, 1 + 32 * 8 + 3, 1 + 32 * 11 + 5, c["proof_len"] - 40, c["proof_len"] + 5, len(rec) - 1):
            bad = bytearray(rec)
            bad[pos] ^= 1
            assert ctx.verify(bytes(bad), *args) in (1, 3), pos
        assert ctx.verify(b"\x02" + rec[1:], *args) == 3                             # bad version byte
        nc = bytearray(rec)
        nc[1 + 32 * 8:1 + 32 * 9] = b"\xff" * 32                                     # non-canonical t_x -> FormatError
        assert ctx.verify(bytes(nc), *args) == 3
        zero = bytearray(rec)
        zero[1:33] = bytes(32)                                                       # identity A_I1 -> VerificationError
        assert ctx.verify(bytes(zero), *args) == 1
        assert ctx.verify(rec[:-1], *args) == 3                                      # wrong length


def _synth_batch(ctx, B, N, seed):
    """SURVEY.md 8d recipe: SHA-512 counter stream; d = u64, k uniform, one seed per batch, x_i at toggle_i = i mod N."""
    def stream(i, tag):
        return hashlib.sha512(b"bbp-bench-v1" + (seed).to_bytes(8, "little") + (i).to_bytes(8, "little") + tag).digest()
    sd = rs.sc_wide(stream(0, b"seed"))
    dks = b"".join(stream(i, b"d")[:8] + bytes(24) + rs.sc_bytes(rs.sc_wide(stream(i, b"k"))) + rs.sc_bytes(sd) for i in range(B))
    w = ctx.witness_batch(dks)
    ins, ents, vins = [], [], []
    for i in range(B):
        m, x, y, yi, q, z = (w[192 * i + 32 * j:192 * i + 32 * j + 32] for j in range(6))
        toggle = i % N
        pub = [rs.sc_bytes(rs.sc_wide(stream(i, b"pub%d" % j))) for j in range(N)]
        pub[toggle] = x
        d, k = dks[96 * i:96 * i + 32], dks[96 * i + 32:96 * i + 64]
        ins.append(d + k + y + yi + q + z + rs.sc_bytes(sd) + b"".join(pub) + toggle.to_bytes(8, "little"))
        ent = b"".join(rs.sc_bytes(rs.sc_wide(stream(i, b"ent%d" % j))) for j in range(4 + N)) + stream(i, b"entseed")[:32]
        ents.append(ent)
        vins.append((q, z, rs.sc_bytes(sd), b"".join(pub)))
    return ins, ents, vins


@pytest.mark.parametrize("N,B", [(8, 6), (1, 3), (3, 4)])
def test_batch_prove_three_way(ctx, oc, bbp, N, B):
    """Device prover == C oracle byte for byte under injected entropy; each side's verifier accepts the other's proofs."""
    ins, ents, vins = _synth_batch(ctx, B, N, seed=7 + N)
    out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert st == [0] * B
    rs_ = bbp.record_size(N)
    cout, cst = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=4)
    assert cst == [0] * B
    for i in range(B):
        assert out[i * rs_:(i + 1) * rs_] == cout[i * rs_:(i + 1) * rs_], i
    vin = b"".join(out[i * rs_:(i + 1) * rs_] + v[0] + v[1] + v[2] + v[3] for i, v in enumerate(vins))
    assert oc.verify_many(vin, B, N, threads=4) == [0] * B          # oracle accepts device proofs
    assert ctx.verify_batch(B, N, vin) == [0] * B                   # device accepts (== oracle's) proofs
    bad = bytearray(vin)
    stride = rs_ + 96 + 32 * N
    bad[1 * stride + 200] ^= 0x10
    exp = [0] * B
    exp[1] = 1
    assert ctx.verify_batch(B, N, bytes(bad)) == exp


def test_prove_os_entropy_roundtrip(ctx, oc):
    """entropy = NULL: OS randomness; two runs differ (like the reference's thread_rng) and both verify everywhere."""
    ins, _, vins = _synth_batch(ctx, 2, 8, seed=99)
    s7, pub = ins[0][:224], ins[0][224:224 + 256]
    r1 = ctx.prove(s7, pub, 0)
    r2 = ctx.prove(s7, pub, 0)
    assert r1 != r2
    for r in (r1, r2):
        assert ctx.verify(r, *vins[0]) == 0
        assert oc.verify(r, *vins[0]) == 0


def test_argument_screening(ctx, bbp, golden):
    c = golden("proofs_full.json")["full"][1]
    with pytest.raises(bbp.BbpError) as e:
        ctx.prove(_s7(c), _pub(c), 8, bytes.fromhex(c["entropy"]))       # toggle >= N
    assert e.value.status == 4
    with pytest.raises(bbp.BbpError) as e:
        ctx.prove(_s7(c), b"", 0, None)                                    # N = 0 (reference panics)
    assert e.value.status == 4
    with pytest.raises(bbp.BbpError) as e:
        ctx.prove(_s7(c), bytes(32 * 203), 0, bytes(bbp.entropy_size(203)))  # 2051 multipliers > 2048 generators
    assert e.value.status == 2
    bad = bytearray(_s7(c))
    bad[0:32] = b"\xff" * 32
    with pytest.raises(bbp.BbpError) as e:
        ctx.prove(bytes(bad), _pub(c), 0, bytes.fromhex(c["entropy"]))   # non-canonical d
    assert e.value.status == 3


def test_max_list_length(ctx, oc, bbp):
    """N = 202 is the largest list the 2048-generator capacity admits (1442 + 3*202 = 2048 multipliers, no padding)."""
    ins, ents, vins = _synth_batch(ctx, 1, 202, seed=5)
    out, st = ctx.prove_batch(1, 202, ins[0], ents[0])
    assert st == [0]
    assert ctx.verify(out, *vins[0]) == 0
    assert oc.verify(out, *vins[0]) == 0
    rc, crec = oc.prove(ins[0][:224], ins[0][224:224 + 32 * 202], 0, ents[0])
    assert rc == 0 and crec == out


def test_record_layout_variants_match_reference_parse_rules(ctx, oc, golden):
    """R1CSProof::from_bytes accepts the compact (0x00) and the 2-phase (0x01) layouts; a well-formed proof of the wrong IPA depth
    is a VerificationError, structural damage a FormatError -- same classification on the device, the C oracle and the Python oracle."""
    from oracle.ref_py import blindbid as pbb, r1cs as pr1cs
    c = golden("proofs_full.json")["full"][1]
    rec = bytes.fromhex(c["record"])
    args = (bytes.fromhex(c["q"]), bytes.fromhex(c["z_img"]), bytes.fromhex(c["seed"]), _pub(c))
    plen = c["proof_len"]
    two_phase = b"\x01" + rec[1:97] + bytes(96) + rec[97:]
    assert ctx.verify(two_phase, *args) == 0 and oc.verify(two_phase, *args) == 0
    f = lambda k: int.from_bytes(bytes.fromhex(c[k]), "little")
    pub_int = [int.from_bytes(bytes.fromhex(p), "little") for p in c["pub_list"]]
    assert pbb.verify(pbb.Proof.from_record(two_phase, c["N"]), f("q"), f("z_img"), f("seed"), pub_int)
    forged = b"\x01" + rec[1:97] + rec[1:97] + rec[97:]            # non-identity phase-2 points
    assert ctx.verify(forged, *args) == 1 and oc.verify(forged, *args) == 1
    shallow = rec[:1 + 32 * 11] + rec[1 + 32 * 11 + 64:]              # drop (L_1, R_1): lg_n = 10, still well formed
    assert ctx.verify(shallow, *args) == 1 and oc.verify(shallow, *args) == 1
    with pytest.raises(pr1cs.VerificationError):
        pbb.verify(pbb.Proof.from_record(shallow, c["N"]), f("q"), f("z_img"), f("seed"), pub_int)
    odd = rec[:plen - 32] + rec[plen:]                                 # ipp with an odd element count
    assert ctx.verify(odd, *args) == 3 and oc.verify(odd, *args) == 3
    assert ctx.verify(rec[:-1], *args) == 3 and ctx.verify(rec[1:], *args) == 3
    assert ctx.verify(rec[plen - 1:], *args) == 3                      # far too short
    nc_b = bytearray(rec)
    nc_b[plen - 32:plen] = b"\xff" * 32                                # non-canonical ipp.b
    assert ctx.verify(bytes(nc_b), *args) == 3 and oc.verify(bytes(nc_b), *args) == 3


@pytest.mark.parametrize("B,N", [(1, 8), (127, 2), (129, 8), (64, 5)])
def test_batch_geometries_and_buffer_regrowth(ctx, oc, bbp, B, N):
    """Single-half (< 128) and uneven two-half batches, changing B and N between calls on one context."""
    ins, ents, vins = _synth_batch(ctx, B, N, seed=1000 + B)
    out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert st == [0] * B
    rs_ = bbp.record_size(N)
    vin = b"".join(out[i * rs_:(i + 1) * rs_] + v[0] + v[1] + v[2] + v[3] for i, v in enumerate(vins))
    assert ctx.verify_batch(B, N, vin) == [0] * B
    for i in sorted({0, B // 2 - 1 if B > 1 else 0, B // 2, B - 1}):     # the seams of the two half-batches
        rc, exp = oc.prove(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i])
        assert rc == 0 and out[i * rs_:(i + 1) * rs_] == exp, i
    # back-to-back calls reuse the alternate pipeline buffer: results must not depend on call parity
    out2, _ = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert out2 == out


@pytest.mark.parametrize("N,B", [(8, 70), (1, 5), (5, 33)])
def test_prepare_bids_on_device_feeds_prove_and_verify(ctx, oc, bbp, N, B):
    """(d, k, seed) + bid list + toggle -> prover rows and verifier tails on the device, byte-equal to the rows the
    host assembles from bbp_witness_batch (whose values the oracle tests pin); the rows then go straight into the device prover
    and verifier without touching the host."""
    import torch
    dev = torch.device("cuda", 0)
    ins, ents, vins = _synth_batch(ctx, B, N, seed=5150 + N)
    bids = b"".join(r[:64] + r[192:224] for r in ins)                                 # d || k || seed
    lists = bytearray(b"".join(r[224:224 + 32 * N] for r in ins))
    toggles = [int.from_bytes(r[-8:], "little") for r in ins]
    for i, t in enumerate(toggles):
        lists[32 * (N * i + t):32 * (N * i + t + 1)] = b"\xee" * 32                  # the bid's own slot is filled in on the device
    d_bids = torch.frombuffer(bytearray(bids), dtype=torch.uint8).to(dev)
    d_lists = torch.frombuffer(lists, dtype=torch.uint8).to(dev)
    d_tog = torch.tensor(toggles, dtype=torch.int64, device=dev)
    in_stride, rs_ = 224 + 32 * N + 8, bbp.record_size(N)
    vt_stride = 96 + 32 * N
    d_in = torch.zeros(B * in_stride, dtype=torch.uint8, device=dev)
    d_vt = torch.zeros(B * vt_stride, dtype=torch.uint8, device=dev)
    d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
    vin = torch.empty((B, rs_ + vt_stride), dtype=torch.uint8, device=dev)
    vent = torch.zeros(B * 32, dtype=torch.uint8, device=dev)
    status = torch.full((B,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    # no host synchronisation from here on: the bid pass on a stream of its own (as a pipelined caller would run it), prove /
    # verify on another
    prep, main = torch.cuda.Stream(), torch.cuda.Stream()
    ctx.prepare_bids_dev(B, N, d_bids.data_ptr(), d_lists.data_ptr(), d_tog.data_ptr(), d_in.data_ptr(), d_vt.data_ptr(), prep.cuda_stream)
    with torch.cuda.stream(main):
        ctx.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), d_out.data_ptr(), main.cuda_stream)  # orders itself after the bid pass
        main.wait_stream(prep)  # for the copy of the verifier tails
        vin[:, :rs_] = d_out.view(B, rs_)
        vin[:, rs_:] = d_vt.view(B, vt_stride)
        ctx.verify_batch_dev(B, N, vin.data_ptr(), vent.data_ptr(), status.data_ptr(), main.cuda_stream)
    torch.cuda.synchronize()
    assert bytes(d_in.cpu().numpy().tobytes()) == b"".join(ins)
    assert bytes(d_vt.cpu().numpy().tobytes()) == b"".join(b"".join(v) for v in vins)
    out = bytes(d_out.cpu().numpy().tobytes())
    rc, exp = oc.prove(ins[B - 1][:224], ins[B - 1][224:224 + 32 * N], toggles[B - 1], ents[B - 1])
    assert rc == 0 and out[(B - 1) * rs_:] == exp
    assert status.cpu().tolist() == [0] * B


def test_host_batches_are_chunked_transparently(ctx, bbp, monkeypatch):
    """The host-pointer batch calls cut large batches into equal engine calls (BBP_HOST_CHUNK_*, default 16384 / 32768): records
    and statuses must not depend on the chunking, including ragged last chunks and rejected items."""
    N, B = 2, 131
    ins, ents, vins = _synth_batch(ctx, B, N, seed=606)
    bad = bytearray(ins[77])
    bad[-8:] = (N).to_bytes(8, "little")                                  # toggle == N -> BAD_ARG, zeroed record
    ins[77] = bytes(bad)
    ref, rst = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert rst == [0] * 77 + [4] + [0] * (B - 78)
    rs_ = bbp.record_size(N)
    rows = [bytearray(ref[i * rs_:(i + 1) * rs_] + b"".join(vins[i])) for i in range(B)]
    rows[5][40] ^= 2
    rows[130][rs_ + 1] ^= 1
    blob = b"".join(bytes(r) for r in rows)
    vref = ctx.verify_batch(B, N, blob)
    assert [i for i, v in enumerate(vref) if v] == [5, 77, 130]
    monkeypatch.setenv("BBP_HOST_CHUNK_PROVE", "50")                      # 3 chunks of 44, 44, 43
    monkeypatch.setenv("BBP_HOST_CHUNK_VERIFY", "60")
    out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert st == rst and out == ref
    assert ctx.verify_batch(B, N, blob) == vref
    got, nfb = ctx.verify_batch_aggregated(B, N, blob, 8)
    assert got == vref and nfb > 0


def test_batch_status_per_item(ctx, bbp):
    ins, ents, _ = _synth_batch(ctx, 4, 3, seed=77)
    bad_toggle = bytearray(ins[1])
    bad_toggle[-8:] = (3).to_bytes(8, "little")                          # toggle == N
    bad_scalar = bytearray(ins[2])
    bad_scalar[32:64] = b"\xff" * 32                                      # non-canonical k
    out, st = ctx.prove_batch(4, 3, ins[0] + bytes(bad_toggle) + bytes(bad_scalar) + ins[3], b"".join(ents))
    assert st == [0, 4, 3, 0]
    rs_ = bbp.record_size(3)
    assert out[rs_:3 * rs_] == bytes(2 * rs_) and out[:rs_] != bytes(rs_)


def test_verify_large_batch_with_corruptions(ctx, bbp):
    """Config-4 shaped check at one-GPU scale: 4096 verifications = 64 distinct proofs tiled, ~1 % corrupted at known indices."""
    N, distinct, B = 8, 64, 4096
    ins, ents, vins = _synth_batch(ctx, distinct, N, seed=4242)
    out, st = ctx.prove_batch(distinct, N, b"".join(ins), b"".join(ents))
    assert st == [0] * distinct
    rs_ = bbp.record_size(N)
    rows = [bytearray(out[(i % distinct) * rs_:(i % distinct + 1) * rs_] + b"".join(vins[i % distinct])) for i in range(B)]
    bad = sorted({(i * 101 + 7) % B for i in range(41)})
    for i in bad:
        rows[i][1 + (i * 37) % 1100] ^= 1 << (i % 8)
    got = ctx.verify_batch(B, N, b"".join(bytes(r) for r in rows))
    assert [i for i, s in enumerate(got) if s != 0] == bad
    assert all(got[i] in (1, 3) for i in bad)


@pytest.mark.parametrize("group", [0, 7, 1000])
def test_aggregated_verification_reports_per_proof_statuses(ctx, bbp, group):
    """SURVEY.md 8f-4 extension: groups of proofs share one weighted generator MSM; failing groups are re-verified proof by
    proof.  The statuses must be exactly those of the per-proof path for every kind of rejection (bad proof bytes, wrong public
    input, non-canonical scalar -> FormatError, undecodable / identity point), also with several bad proofs in one group, a
    ragged last group and one group spanning the whole batch; an all-honest batch must not take the per-proof path at all."""
    N, distinct, B = 3, 24, 203
    ins, ents, vins = _synth_batch(ctx, distinct, N, seed=808)
    out, st = ctx.prove_batch(distinct, N, b"".join(ins), b"".join(ents))
    assert st == [0] * distinct
    rs_ = bbp.record_size(N)
    rows = [bytearray(out[(i % distinct) * rs_:(i % distinct + 1) * rs_] + b"".join(vins[i % distinct])) for i in range(B)]
    honest = b"".join(bytes(r) for r in rows)
    got, nfb = ctx.verify_batch_aggregated(B, N, honest, group)
    assert got == [0] * B and nfb == 0
    kinds = {}
    for j, i in enumerate(sorted({(k * 29 + 3) % B for k in range(17)} | {8, 9, 10, B - 1})):  # 8, 9, 10: neighbours in one group
        kind = j % 5
        if kind == 0:
            rows[i][1 + 32 * 14 + 5] ^= 0x10             # a bit of an IPA point / scalar region
        elif kind == 1:
            rows[i][rs_ + 3] ^= 1                        # wrong score (public input)
        elif kind == 2:
            rows[i][1 + 32 * 8:1 + 32 * 9] = b"\xff" * 32   # t_x non-canonical -> FormatError
        elif kind == 3:
            rows[i][1:33] = b"\xff" * 32                 # A_I1 is not a valid ristretto encoding
        else:
            rows[i][1 + 32:1 + 64] = bytes(32)           # A_O1 = identity encoding: validate_and_append_point rejects
        kinds[i] = kind
    blob = b"".join(bytes(r) for r in rows)
    plain = ctx.verify_batch(B, N, blob)
    assert sorted(i for i, v in enumerate(plain) if v != 0) == sorted(kinds)
    assert all(plain[i] == 3 for i, k in kinds.items() if k == 2)
    got, nfb = ctx.verify_batch_aggregated(B, N, blob, group)
    assert got == plain
    assert 0 < nfb <= B


def test_calls_from_different_streams_are_serialised(ctx, oc, bbp):
    """One context shares scratch between calls; calls issued on different caller streams must still be ordered (stream guard).
    Regression: two chunks in flight on two streams once raced on the MSM scratch and faulted the GPU."""
    import torch
    dev = torch.device("cuda", 0)
    B, N = 160, 4
    ins, ents, vins = _synth_batch(ctx, B, N, seed=31337)
    rs_ = bbp.record_size(N)
    d_in = torch.frombuffer(bytearray(b"".join(ins)), dtype=torch.uint8).to(dev)
    d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
    vtail = torch.frombuffer(bytearray(b"".join(b"".join(v) for v in vins)), dtype=torch.uint8).to(dev).view(B, 96 + 32 * N)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs, sts = [], []
    for it in range(6):
        st = streams[it % 3]
        with torch.cuda.stream(st):
            out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
            ctx.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), out.data_ptr(), st.cuda_stream)
            vin = torch.empty((B, rs_ + 96 + 32 * N), dtype=torch.uint8, device=dev)
            vin[:, :rs_] = out.view(B, rs_)
            vin[:, rs_:] = vtail
            status = torch.full((B,), -1, dtype=torch.int32, device=dev)
            ctx.verify_batch_dev(B, N, vin.data_ptr(), torch.zeros(B * 32, dtype=torch.uint8, device=dev).data_ptr(), status.data_ptr(),
                                 st.cuda_stream)
            outs.append((out, vin))
            sts.append(status)
    torch.cuda.synchronize()
    ref = bytes(outs[0][0].cpu().numpy().tobytes())
    rc, exp = oc.prove(ins[B - 1][:224], ins[B - 1][224:224 + 32 * N], int.from_bytes(ins[B - 1][-8:], "little"), ents[B - 1])
    assert rc == 0 and ref[(B - 1) * rs_:] == exp
    for (out, _), status in zip(outs, sts):
        assert bytes(out.cpu().numpy().tobytes()) == ref
        assert status.cpu().tolist() == [0] * B


@pytest.mark.parametrize("knobs", [
    {"BBP_TAIL_ROUND": "12"},                                   # all 11 IPA rounds on the fixed-base MSM kernels (no folded-generator tail)
    {"BBP_SLICES": "1", "BBP_SERIAL_LDS": "0"},                 # one heavy-stage stream, serial kernels not fenced off
    {"BBP_SLICES": "4", "BBP_SERIAL_BLOCK": "256"},             # four slices (two of them share a hardware queue by default)
    {"BBP_SLICES": "2", "BBP_STAGGER": "1"},
    {"BBP_VARBASE_LANES": "1000", "BBP_DUAL_OPEN_BELOW": "0"},  # verifier: 3 lanes per proof, ~15 points per lane on one doubling chain
    {"BBP_VARBASE_LANES": "64", "BBP_VERIFY_OVERLAP": "0"},     # verifier: one lane per proof, variable-base kernel in line
    {"BBP_RNG_COOP": "0"},                                      # TranscriptRng draw chain on one lane per proof (round-1 path)
    {"BBP_RNG_COOP": "1", "BBP_RNG_BLOCK": "64", "BBP_SERIAL_LDS": "0"},  # cooperative rng forced: one wavefront (two proofs) per workgroup, not fenced
    {"BBP_RNG_COOP": "1", "BBP_RNG_BLOCK": "1024"},             # cooperative rng forced: 32 proofs per reserved CU
    {"BBP_RNG_COOP": "1", "BBP_WITNESS_NATIVE": "0"},           # ... with the gates interpreted from the compiled gadget program (default: written from the wiring itself)
    {"BBP_RNG_COOP": "1", "BBP_RNG_DPP": "1"},                  # ... one word per lane with DPP / permlane-swap theta (round 3; the default is round 4's half-word form, keccak_wave.h)
    {"BBP_RNG_COOP": "1", "BBP_RNG_DPP": "1", "BBP_RNG_BLOCK": "64", "BBP_SERIAL_LDS": "0"},
    {"BBP_RNG_COOP": "1", "BBP_RNG_DPP": "0"},                  # ... in its 25-lane ds_bpermute form (two proofs per wavefront)
    {"BBP_RNG_COOP": "1", "BBP_RNG_DPP": "0", "BBP_RNG_BLOCK": "64", "BBP_SERIAL_LDS": "0"},
    {"BBP_TAIL_SMALL_BELOW": "0", "BBP_SLICES": "1"},           # small heavy stages take the folded-generator tail as well
    {"BBP_FOLD_HALF_FROM": "1"},                                # every MSM launch folds on half a wavefront per MSM
    {"BBP_FOLD_HALF_FROM": "1000000"},                          # ... and none does (128 lanes per MSM)
])
def test_engine_schedules_give_identical_bytes(bbp, oc, knobs):
    """The scheduling knobs (slices, tail round, serial-kernel fencing, stagger) change WHEN and HOW work runs, never the bytes:
    every variant must reproduce the C oracle's records under the same entropy, for a batch that spans every slice."""
    import os
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        c2 = bbp.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        B, N = 261, 3
        ins, ents, vins = _synth_batch(c2, B, N, seed=4242)
        rs_ = bbp.record_size(N)
        for _ in range(2):  # second call runs on the other buffer parity, overlapped with the first
            out, st = c2.prove_batch(B, N, b"".join(ins), b"".join(ents))
            assert st == [0] * B
        cout, cst = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
        assert cst == [0] * B
        assert out == cout
        vin = b"".join(out[i * rs_:(i + 1) * rs_] + v[0] + v[1] + v[2] + v[3] for i, v in enumerate(vins))
        assert c2.verify_batch(B, N, vin) == [0] * B
    finally:
        c2.close()


@pytest.mark.parametrize("knob,value", [("BBP_MSM_SMALL", "1"), ("BBP_MSM_SMALL", "0"), ("BBP_COMMIT_SPLIT_BELOW", "0"), ("BBP_WITNESS_NATIVE", "0"), ("BBP_IPA_WIDE_BELOW", "0"), ("BBP_TR_WAVE_BELOW", "0"),
                                        ("BBP_TAIL_SMALL_BELOW", "0")])
def test_small_call_paths_give_identical_bytes(bbp, oc, knob, value):
    """What only small launches take.  Launches of fewer than 128 MSMs are cut into sub-MSMs; those use 128 buckets and width-9
    digits (msm.hip msm_geom<2>) or, BBP_MSM_SMALL=0, the 1024 buckets of the unsplit kernels.  Pedersen-commitment launches of
    at most 1024 commitments put each on eight lanes (BBP_COMMIT_SPLIT_BELOW=0: one lane).  The cooperative opening launches write
    the gates from the gadget wiring (BBP_WITNESS_NATIVE=0: interpreted); launches of at most 32 proofs run the transcript kernels
    with one proof per wavefront (BBP_TR_WAVE_BELOW=0: one lane) -- also when the IPA tail kernels, which stay one-lane, follow them
    (BBP_TAIL_SMALL_BELOW=0).  Every way the records are the C oracle's, for one proof
    (sixteen sub-MSMs per MSM) and for batches that are cut in fewer pieces."""
    import os
    old = os.environ.get(knob)
    os.environ[knob] = value
    try:
        c2 = bbp.Context(0)
    finally:
        if old is None:
            os.environ.pop(knob, None)
        else:
            os.environ[knob] = old
    try:
        for B, N, seed in ((1, 8, 77), (23, 3, 78), (100, 1, 79)):
            ins, ents, vins = _synth_batch(c2, B, N, seed=seed)
            rs_ = bbp.record_size(N)
            out, st = c2.prove_batch(B, N, b"".join(ins), b"".join(ents))
            assert st == [0] * B
            cout, cst = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
            assert cst == [0] * B and out == cout, (B, N)
            vin = b"".join(out[i * rs_:(i + 1) * rs_] + v[0] + v[1] + v[2] + v[3] for i, v in enumerate(vins))
            assert c2.verify_batch(B, N, vin) == [0] * B
        assert c2.health() == 0
    finally:
        c2.close()


def test_pipelined_calls_of_mixed_batch_sizes(bbp, oc):
    """Back-to-back device calls without host synchronisation, alternating between batch sizes below and above the
    dual-opening threshold (two opening streams over three buffers vs one stream over two): every call must reproduce the
    records of the same inputs proven alone, and the oracle's record for a sampled proof."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    old = os.environ.get("BBP_DUAL_OPEN_BELOW")
    os.environ["BBP_DUAL_OPEN_BELOW"] = "128"
    try:
        c2 = bbp.Context(0)
    finally:
        if old is None:
            os.environ.pop("BBP_DUAL_OPEN_BELOW", None)
        else:
            os.environ["BBP_DUAL_OPEN_BELOW"] = old
    try:
        N = 2
        rs_ = bbp.record_size(N)
        sets = []
        for k, B in enumerate((70, 130, 96, 192, 33)):  # 70 / 96 / 33 open on alternating streams, 130 / 192 on the single one
            ins, ents, _ = _synth_batch(c2, B, N, seed=900 + k)
            d_in = torch.frombuffer(bytearray(b"".join(ins)), dtype=torch.uint8).to(dev)
            d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
            solo, st = c2.prove_batch(B, N, b"".join(ins), b"".join(ents))
            assert st == [0] * B
            rc, exp = oc.prove(ins[B - 1][:224], ins[B - 1][224:224 + 32 * N], int.from_bytes(ins[B - 1][-8:], "little"), ents[B - 1])
            assert rc == 0 and solo[(B - 1) * rs_:] == exp
            sets.append((B, d_in, d_ent, solo))
        torch.cuda.synchronize()
        s = torch.cuda.current_stream().cuda_stream
        outs = []
        for it in range(15):
            B, d_in, d_ent, solo = sets[(it * 3) % 5]
            out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
            c2.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), out.data_ptr(), s)
            outs.append((out, solo))
        torch.cuda.synchronize()
        for it, (out, solo) in enumerate(outs):
            assert bytes(out.cpu().numpy().tobytes()) == solo, it
    finally:
        c2.close()


def test_deep_pipeline_takes_the_rotating_path_with_the_same_bytes(bbp, oc, capfd):
    """A caller that keeps three or more prove calls in flight is moved from slices of one call to whole calls in rotation
    (prover.hip "deep"); a batch that arrives behind a sliced one and is not small takes the sliced path.  Neither changes a byte:
    every pipelined call must reproduce the records of the same inputs proven alone, and the engine's own trace must show that the
    rotating path was actually taken."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    knobs = {"BBP_ROTATE_BELOW": "0", "BBP_DUAL_OPEN_BELOW": "0", "BBP_ROTATE_MIXED_FROM": "64", "BBP_TRACE_PROVE": "1"}  # every batch size counts as large
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        c2 = bbp.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        N = 2
        rs_ = bbp.record_size(N)
        sets = []
        for k, B in enumerate((70, 200, 33)):
            ins, ents, _ = _synth_batch(c2, B, N, seed=1500 + k)
            d_in = torch.frombuffer(bytearray(b"".join(ins)), dtype=torch.uint8).to(dev)
            d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
            solo, st = c2.prove_batch(B, N, b"".join(ins), b"".join(ents))
            assert st == [0] * B
            rc, exp = oc.prove(ins[B - 1][:224], ins[B - 1][224:224 + 32 * N], int.from_bytes(ins[B - 1][-8:], "little"), ents[B - 1])
            assert rc == 0 and solo[(B - 1) * rs_:] == exp
            sets.append((B, d_in, d_ent, solo))
        torch.cuda.synchronize()
        capfd.readouterr()
        s = torch.cuda.current_stream().cuda_stream
        outs = []
        for it in range(14):
            B, d_in, d_ent, solo = sets[it % 3]
            out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
            c2.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), out.data_ptr(), s)
            outs.append((out, solo))
        torch.cuda.synchronize()
        for it, (out, solo) in enumerate(outs):
            assert bytes(out.cpu().numpy().tobytes()) == solo, it
        trace = [l for l in capfd.readouterr().err.splitlines() if l.startswith("prove call")]
        assert len(trace) == 14
        assert " rotate 0 " in trace[0] and " deep 0 " in trace[0]          # nothing in flight yet: slices
        assert sum(" deep 1 " in l and " rotate 1 " in l for l in trace) >= 8  # the pipeline filled up: whole calls in rotation
        assert c2.health() == 0
    finally:
        c2.close()


def test_a_lone_large_call_takes_the_cooperative_draw_chain(bbp, oc, capfd):
    """A batch above the cooperative threshold still gets one wavefront per proof for its TranscriptRng chain when it finds the
    device without an earlier prove call (prover.hip `rng_coop_idle_below`); the calls queued behind it keep the single-lane chain.
    Same bytes either way: every call must reproduce the solo records, the last record the C oracle's."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    knobs = {"BBP_RNG_COOP_BELOW": "16", "BBP_RNG_COOP_IDLE_BELOW": "4096", "BBP_TRACE_PROVE": "1"}
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        c2 = bbp.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        N, B = 2, 90
        rs_ = bbp.record_size(N)
        ins, ents, _ = _synth_batch(c2, B, N, seed=1700)
        d_in = torch.frombuffer(bytearray(b"".join(ins)), dtype=torch.uint8).to(dev)
        d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
        solo, st = c2.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert st == [0] * B
        rc, exp = oc.prove(ins[B - 1][:224], ins[B - 1][224:224 + 32 * N], int.from_bytes(ins[B - 1][-8:], "little"), ents[B - 1])
        assert rc == 0 and solo[(B - 1) * rs_:] == exp
        torch.cuda.synchronize()
        capfd.readouterr()
        s = torch.cuda.current_stream().cuda_stream
        outs = []
        for it in range(4):
            out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
            c2.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), out.data_ptr(), s)
            outs.append(out)
        torch.cuda.synchronize()
        for it, out in enumerate(outs):
            assert bytes(out.cpu().numpy().tobytes()) == solo, it
        trace = [l for l in capfd.readouterr().err.splitlines() if l.startswith("prove call")]
        assert len(trace) == 4
        assert " inflight 0 " in trace[0] and trace[0].rstrip().endswith("coop 1")   # the device was idle: a wavefront per proof
        assert any(l.rstrip().endswith("coop 0") for l in trace[1:])                    # behind it: the single-lane chain
        assert c2.health() == 0
    finally:
        c2.close()


def test_pipelined_calls_with_different_list_lengths(ctx, oc, bbp):
    """Consecutive device calls with different N use different compiled circuits (index lists, constraint tables) while the
    previous call is still in flight: outputs must equal the solo runs and the oracle's record."""
    import torch
    dev = torch.device("cuda", 0)
    sets = []
    for k, (B, N) in enumerate(((40, 2), (33, 9), (70, 1), (25, 30))):
        ins, ents, _ = _synth_batch(ctx, B, N, seed=1200 + k)
        d_in = torch.frombuffer(bytearray(b"".join(ins)), dtype=torch.uint8).to(dev)
        d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
        solo, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert st == [0] * B
        rs_ = bbp.record_size(N)
        rc, exp = oc.prove(ins[0][:224], ins[0][224:224 + 32 * N], int.from_bytes(ins[0][-8:], "little"), ents[0])
        assert rc == 0 and solo[:rs_] == exp
        sets.append((B, N, d_in, d_ent, solo))
    torch.cuda.synchronize()
    main = torch.cuda.Stream()
    outs = []
    for it in range(12):
        B, N, d_in, d_ent, solo = sets[(it * 3) % 4]
        out = torch.zeros(B * bbp.record_size(N), dtype=torch.uint8, device=dev)
        ctx.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), out.data_ptr(), main.cuda_stream)
        outs.append((out, solo))
    torch.cuda.synchronize()
    for it, (out, solo) in enumerate(outs):
        assert bytes(out.cpu().numpy().tobytes()) == solo, it


def test_config3_all_1024_records_under_both_schedules(bbp, oc, capfd):
    """BASELINE configs[2] at its stated size with EVERY record checked (20 of 1024 were): six device calls of 1024
    proofs (N = 8) back to back.  The first finds an idle device and is cut into three slices; from the fourth on three earlier calls
    are in flight and whole calls run in rotation over five buffers (prover.hip "deep": the schedule behind the headline number).
    The engine's own trace must show both paths, and all 1024 records of a sliced call AND of a rotating call must equal what the C
    oracle proves from the same inputs and entropy (~25 s of oracle time on the box's cores)."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    old = os.environ.get("BBP_TRACE_PROVE")
    os.environ["BBP_TRACE_PROVE"] = "1"
    try:
        c2 = bbp.Context(0)
    finally:
        if old is None:
            os.environ.pop("BBP_TRACE_PROVE", None)
        else:
            os.environ["BBP_TRACE_PROVE"] = old
    try:
        B, N = 1024, 8
        rs_ = bbp.record_size(N)
        ins, ents, _ = _synth_batch(c2, B, N, seed=31337)
        exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=os.cpu_count() or 8)
        assert est == [0] * B
        d_in = torch.frombuffer(bytearray(b"".join(ins)), dtype=torch.uint8).to(dev)
        d_ent = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
        torch.cuda.synchronize()
        capfd.readouterr()
        s = torch.cuda.current_stream().cuda_stream
        outs = []
        for it in range(6):
            out = torch.zeros(B * rs_, dtype=torch.uint8, device=dev)
            c2.prove_batch_dev(B, N, d_in.data_ptr(), d_ent.data_ptr(), out.data_ptr(), s)
            outs.append(out)
        torch.cuda.synchronize()
        trace = [l for l in capfd.readouterr().err.splitlines() if l.startswith("prove call")]
        assert len(trace) == 6, trace
        sliced = [i for i, l in enumerate(trace) if " deep 0 " in l and " rotate 0 " in l]
        rotating = [i for i, l in enumerate(trace) if " deep 1 " in l and " rotate 1 " in l]
        assert sliced and sliced[0] == 0 and rotating, trace
        for which in (sliced[0], rotating[-1]):
            got = bytes(outs[which].cpu().numpy().tobytes())
            bad = [i for i in range(B) if got[i * rs_:(i + 1) * rs_] != exp[i * rs_:(i + 1) * rs_]]
            assert not bad, (which, trace[which], bad[:10])
        assert c2.health() == 0
    finally:
        c2.close()


def test_config3_full_batch(ctx, oc, bbp):
    """SURVEY.md 8d config 3 at full size: 1024 full proves (N = 8) in one batch call; the first 16 and the last 4 records
    byte-compared with the C oracle under the same entropy, every proof accepted by the device verifier, a spread sample of 48
    accepted by the oracle's verifier, and the batch is identical when proved again (other buffer parity, pipeline warm)."""
    B, N = 1024, 8
    ins, ents, vins = _synth_batch(ctx, B, N, seed=1)
    rs_ = bbp.record_size(N)
    out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert st == [0] * B
    out2, st2 = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    assert st2 == st and out2 == out
    pick = list(range(16)) + list(range(B - 4, B))
    cout, cst = oc.prove_many(b"".join(ins[i] for i in pick), b"".join(ents[i] for i in pick), len(pick), N, threads=8)
    assert cst == [0] * len(pick)
    for j, i in enumerate(pick):
        assert out[i * rs_:(i + 1) * rs_] == cout[j * rs_:(j + 1) * rs_], i
    vin = [out[i * rs_:(i + 1) * rs_] + b"".join(vins[i]) for i in range(B)]
    assert ctx.verify_batch(B, N, b"".join(vin)) == [0] * B
    sample = list(range(0, B, 22))[:48]
    assert oc.verify_many(b"".join(vin[i] for i in sample), len(sample), N, threads=8) == [0] * len(sample)
