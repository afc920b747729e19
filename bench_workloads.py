"""Synthetic workloads for bench.py (SURVEY.md 8d recipes). Inputs live in HBM before the timed region starts."""
import hashlib
import os
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
TAG_NAMES = {1: "k_msm_acc", 11: "k_msm_sort", 2: "k_encode", 3: "k_witness_head", 4: "k_open_serial(witness+rng)", 5: "k_poly/powers/flatten", 6: "k_ipa_round", 7: "k_commit",
             8: "k_transcript", 9: "k_vscalars", 10: "k_varbase/tail", 12: "k_msm_fold"}
L = 2**252 + 27742317777372353535851937790883648493


def _oracle_lib():
    """C restatement under oracle/ -- used ONLY as checker and as the timed CPU baseline."""
    import __graft_entry__ as ge
    path = ge.build_oracle()
    if not path or not os.path.exists(path):
        return None
    from tests import oracle_c
    return oracle_c.load(path)


def host_threads():
    """CPU threads this process may really use: cgroup quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("BBP_CPU_THREADS", "64"))))


def _stream(seed, i, tag):
    """SURVEY.md 8d PRNG: SHA-512 counter stream."""
    return hashlib.sha512(b"bbp-bench-v1" + seed.to_bytes(8, "little") + i.to_bytes(8, "little") + tag).digest()


def _wide(b):
    return (int.from_bytes(b, "little") % L).to_bytes(32, "little")


def _bid_row(seed, i, N, sd, dk, w):
    """One synthetic bid: prover input row, prover entropy, public list, (q, z_img, seed) from its witness row w."""
    m, x, y, yi, q, z = (w[32 * j:32 * j + 32] for j in range(6))
    toggle = i % N
    pub = [_wide(_stream(seed, i, b"pub%d" % j)) for j in range(N)]
    pub[toggle] = x
    pub = b"".join(pub)
    row = dk[:64] + y + yi + q + z + sd + pub + toggle.to_bytes(8, "little")
    ent = b"".join(_wide(_stream(seed, i, b"ent%d" % j)) for j in range(4 + N)) + _stream(seed, i, b"entseed")[:32]
    return row, ent, pub, q + z + sd


def _bid_dks(seed, i, sd):
    return _stream(seed, i, b"d")[:8] + bytes(24) + _wide(_stream(seed, i, b"k")) + sd


def synth_bids(ctx, B, N, seed):
    """d = uniform u64, k uniform scalar, ONE seed per batch, witness on the device, x_i placed at toggle_i = i mod N."""
    sd = _wide(_stream(seed, 0, b"seed"))
    dks = b"".join(_bid_dks(seed, i, sd) for i in range(B))
    w = ctx.witness_batch(dks)
    ins, ents, pubs, qz = [], [], [], []
    for i in range(B):
        row, ent, pub, t = _bid_row(seed, i, N, sd, dks[96 * i:96 * i + 96], w[192 * i:192 * i + 192])
        ins.append(row)
        ents.append(ent)
        pubs.append(pub)
        qz.append(t)
    return ins, ents, pubs, qz


def oracle_bid_rows(lib, seed, idxs, N):
    """The same synthetic bids for the rows `idxs` of the batch seeded `seed`, made WITHOUT the engine (witness by the C oracle):
    what rank 0 needs to check records gathered from another rank, whose inputs it never held."""
    sd = _wide(_stream(seed, 0, b"seed"))
    out = []
    for i in idxs:
        dk = _bid_dks(seed, i, sd)
        out.append(_bid_row(seed, i, N, sd, dk, lib.witness(dk)))
    return out


def _to_dev(torch, device, data):
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8)
    return t.to(device)


def _kernel_table(timings):
    agg = {}
    for tag, us in timings:
        a = agg.setdefault(TAG_NAMES.get(tag, str(tag)), [0, 0.0])
        a[0] += 1
        a[1] += us
    return {k: {"launches": v[0], "total_us": round(v[1], 1)} for k, v in agg.items()}


# average non-zero digits per scalar of the engine's recodings (tests/test_host_arith.py::test_naf_recoding measures them):
# width-12 NAF in the <0> MSM kernels, width-9 NAF in the <1> (generator fold) instances; one digit = one table-row addition
NAF12_DIGITS, NAF9_DIGITS = 19.85, 25.66
MERGED_AI_TERMS = 4 * 90 * 4  # A_I1 terms that ride on another term's merged base (prover.hip circuit_get), any N


def _traffic_from_profiles(key):
    """HBM-side bytes of the dominant kernel per STEP (one batch) as MEASURED by the rocprofv3 PMC passes of the named profile (FETCH_SIZE and
    WRITE_SIZE in separate runs; FETCH_SIZE x 2 on gfx950 per MI355X_MICROARCH.md's HBM section).  The figure is read from
    profiles/traffic.json, which tools/pmc_aggregate.py writes from the counter CSVs -- not measured in this run, so the bench
    line names the file it comes from; (None, None) when there is no profile of this workload."""
    try:
        import json
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]
        return int(t["bytes_per_step"]), t["source"] + (" [measured %s]" % t["measured_on"] if t.get("measured_on") else "")
    except Exception:
        return None, None


class _Base:
    dominant_tag = 1
    dominant_kernel = "k_msm_acc"
    traffic_key = None
    data = "synthetic"

    def measured_traffic(self):
        return _traffic_from_profiles(self.traffic_key) if self.traffic_key else (None, None)

    def extra_report(self, timings):
        return {"kernels_us": _kernel_table(timings)}

    def gather(self, dist, rank, world):
        """The path's one collective: this rank's results to rank 0, in global proof order; rank 0 gets the tensor, others None."""
        return None

    def check_gathered(self, gathered, world):
        """Rank 0, outside the timed region: a sample from EVERY rank's block of the gathered results against the oracle."""

    def drain(self):
        pass

    def clone_for(self, ctx2):
        """The same inputs on another context of the same device, with outputs of its own (bench.py's exclusive pass); None = this
        workload has no exclusive form."""
        return None

    def describe_difference(self, other):
        return ""


class StubWorkload(_Base):
    """BBP_BENCH_STUB=1 only: no engine, no GPU -- lets tests/test_bench_launcher.py drive bench.py's N-rank launch path (process
    start, rendezvous, barrier, max-over-ranks timing, one JSON line) on a CPU box over gloo.  Never a measurement."""
    metric, unit, data = "stub (launcher plumbing test, not a measurement)", "steps/s", "stub"

    def __init__(self, batch):
        self.units_per_step = batch
        self.config = {"workload": "stub: sleeps 1 ms per step", "batch_per_gpu": batch}

    def step(self, stream):
        time.sleep(0.001)

    def check(self):
        pass

    def extra_report(self, timings):
        return {}


class MsmWorkload(_Base):
    """BASELINE.json configs[1]: B proofs x (A_I1: 1+2n1, A_O1: 1+n1, S1: 1+2n1 terms), n1 = 1442 + 3N."""

    metric = "blind-bid proofs/sec (commitment-MSM stage only: A_I1+A_O1+S1 per proof)"
    unit = "proofs/s"

    def __init__(self, ctx, bbp, torch, device, batch, items, seed):
        self.ctx, self.bbp, self.torch, self.B, self.seed = ctx, bbp, torch, batch, seed
        self.n1 = 1442 + 3 * items
        self.shapes = [(1 + 2 * self.n1, bbp.LAYOUT_BLIND_G_H), (1 + self.n1, bbp.LAYOUT_BLIND_G),
                       (1 + 2 * self.n1, bbp.LAYOUT_BLIND_G_H)]
        from bench import synth_scalars_device
        self.scal = [synth_scalars_device(torch, batch, n, seed * 16 + i, device) for i, (n, _) in enumerate(self.shapes)]
        self.out = [torch.zeros((batch, 32), dtype=torch.uint8, device=device) for _ in self.shapes]
        self.units_per_step = batch
        terms = sum(n for n, _ in self.shapes)
        # 160 B read per term (32 B scalar + 128 B extended point) + 32 B written per MSM
        self.alg_bytes_per_step = batch * (terms * 160 + 32 * len(self.shapes))
        self.row_additions_per_step = batch * terms * NAF12_DIGITS
        self.dominant_launches_per_step = len(self.shapes)
        self.traffic_key = "msm_b%d_n%d" % (batch, items)
        self.config = {"workload": "configs[1]: batch of %d blind-bid proofs, commitment MSMs only (N=%d: %s terms)"
                       % (batch, items, "+".join(str(n) for n, _ in self.shapes)),
                       "batch_per_gpu": batch, "bid_list_len": items, "msm_recoding": "NAF-12", "parallelism": "batch-sharded"}

    def step(self, stream):
        for (n, layout), s, o in zip(self.shapes, self.scal, self.out):
            self.ctx.msm_batch_dev(self.B, n, s.data_ptr(), layout, o.data_ptr(), stream)

    def clone_for(self, ctx2):
        import copy
        w = copy.copy(self)
        w.ctx = ctx2
        w.out = [self.torch.zeros_like(o) for o in self.out]
        self.torch.cuda.synchronize()  # (zero-fill on torch's stream before another context's stream writes the tensors)
        return w

    def same_results(self, other):
        return all(bool((a == b).all()) for a, b in zip(self.out, other.out))

    def check(self):
        lib = _oracle_lib()
        for (n, layout), s, o in zip(self.shapes, self.scal, self.out):
            for r in (0, self.B - 1):
                sc = bytes(s[r].cpu().numpy().tobytes())
                got = bytes(o[r].cpu().numpy().tobytes())
                if got != lib.msm_layout(sc, n, layout):
                    raise SystemExit("PARITY FAILURE in bench msm workload row %d" % r)

    def gather(self, dist, rank, world):
        from dusk_blindbidproof_amd import sharding
        rec = self.torch.cat(self.out, dim=1).contiguous().view(-1)  # per proof: A_I1 || A_O1 || S1, 96 bytes
        return sharding.gather_records(dist, rec, 96, self.B * world, rank, world)

    def check_gathered(self, gathered, world):
        from bench import synth_scalars_device
        lib = _oracle_lib()
        got = bytes(gathered.cpu().numpy().tobytes())
        for r in range(world):  # rank r's scalars are regenerated here from its seed (same generator, same device type)
            for i, (n, layout) in enumerate(self.shapes):
                s = synth_scalars_device(self.torch, self.B, n, (self.seed + r) * 16 + i, self.scal[0].device)
                for row in (0, self.B - 1):
                    exp = lib.msm_layout(bytes(s[row].cpu().numpy().tobytes()), n, layout)
                    o = 96 * (r * self.B + row) + 32 * i
                    if got[o:o + 32] != exp:
                        raise SystemExit("PARITY FAILURE in gathered msm results: rank %d row %d msm %d" % (r, row, i))
                del s

    def cpu_baseline(self):
        lib = _oracle_lib()
        threads = host_threads()
        sample = max(2 * threads, 8)
        rows = [bytes(self.scal[i][r % self.B].cpu().numpy().tobytes()) for r in range(sample) for i in range(3)]
        t0 = time.perf_counter()
        lib.msm_layout_many(rows, [self.shapes[i % 3][0] for i in range(len(rows))],
                            [self.shapes[i % 3][1] for i in range(len(rows))], threads)
        dt = time.perf_counter() - t0
        return {"value": sample / dt, "unit": self.unit, "cores": threads, "kind": "port",
                "sample": "%d proofs' A_I1+A_O1+S1 MSMs (vartime Pippenger, C oracle, %d threads) in %.1f s" % (sample, threads, dt)}


class ProveWorkload(_Base):
    """BASELINE.json configs[2]: batch of B full R1CS proves (gadgets + MSMs + polynomial sweep + IPA) on one GPU."""

    metric = "blind-bid proofs/sec"
    unit = "proofs/s"

    def __init__(self, ctx, bbp, torch, device, batch, items, seed):
        self.ctx, self.bbp, self.torch, self.B, self.N, self.device, self.seed = ctx, bbp, torch, batch, items, device, seed
        self.ins, self.ents, self.pubs, self.qz = synth_bids(ctx, batch, items, seed)
        self.in_dev = _to_dev(torch, device, b"".join(self.ins))
        self.ent_dev = _to_dev(torch, device, b"".join(self.ents))
        self.rec = bbp.record_size(items)
        self.out_dev = torch.zeros(batch * self.rec, dtype=torch.uint8, device=device)
        # The timed loop rotates over several DISTINCT input sets (round 4; one set proven 20 times before): set 0 is the one every check,
        # the exclusive pass and the gather look at (out_dev); sets 1.. have their own bids, entropy and output buffers.  The work is
        # data-independent apart from NAF digit counts, so the figure does not move -- but no step repeats the previous step's inputs.
        self.k = 0
        self.sets = [(self.in_dev, self.ent_dev, self.out_dev)]
        for j in range(1, max(1, int(os.environ.get("BBP_BENCH_INPUT_SETS", "3")))):
            ins_j, ents_j, _, _ = synth_bids(ctx, batch, items, seed + 7919 * j)
            self.sets.append((_to_dev(torch, device, b"".join(ins_j)), _to_dev(torch, device, b"".join(ents_j)),
                              torch.zeros(batch * self.rec, dtype=torch.uint8, device=device)))
        if hasattr(ctx, "reserve") and not os.environ.get("BBP_BENCH_NO_RESERVE"):
            ctx.reserve(batch, items)  # bbp_reserve: every buffer of every schedule sized before the clock starts (what a server does at start-up)
        self.units_per_step = batch
        n1 = 1442 + 3 * items
        commit_terms = (1 + 2 * n1) * 2 + (1 + n1)
        # the REFERENCE algorithm's MSM terms per prove (V, A_I1, A_O1, S1, T, Q, IPA L/R with shrinking halves, generator
        # folds) at 160 B read per term + 32 B per output point: 23 766 terms = 3.80 MB per proof at N = 8
        ref_terms = 2 * (4 + items) + 5 * n1 + 3 + 11 + 8210 + 8188
        self.alg_bytes_per_step = batch * (ref_terms * 160 + 32 * ((4 + items) + 8 + 22))
        # what the engine's MSM kernels actually add (fold-free IPA: rounds 1-6 are 2 x 2049-term MSMs over the original generators,
        # then one composite-bucket pass over all 4096 generators; the tail rounds are variable-base work outside the MSM kernels)
        # (round 1 walks the 2048 - n1 zero-padded multipliers' common-scalar terms of L as ONE term on a precomputed sum)
        # (A_I1: the 4 x 90 MiMC rounds wire a to three multiplier inputs and a^2 to three more; each triple is ONE term on a merged
        #  base, MERGED_AI_TERMS terms fewer -- DESIGN.md "Merged bases")
        engine_terms = commit_terms - MERGED_AI_TERMS + 6 * 2 * 2049 - max(2048 - n1 - 1, 0)
        self.row_additions_per_step = batch * (engine_terms * NAF12_DIGITS + 4096 * NAF9_DIGITS)
        self.dominant_launches_per_step = 3 + 6 + 1
        self.traffic_key = "prove_b1024_n8" if batch == 1024 and items == 8 else None
        self.config = {"workload": "configs[2]: batch of %d full blind-bid R1CS proves (N=%d, 1466 multipliers, 11 IPA rounds)" % (batch, items),
                       "batch_per_gpu": batch, "bid_list_len": items, "msm_recoding": "NAF-12", "parallelism": "batch-sharded",
                       "ref_msm_terms_per_proof": ref_terms, "engine_msm_terms_per_proof": engine_terms + 4096,
                       "distinct_input_sets_in_rotation": len(self.sets)}

    def step(self, stream):
        in_dev, ent_dev, out_dev = self.sets[self.k % len(self.sets)]
        self.k += 1
        self.ctx.prove_batch_dev(self.B, self.N, in_dev.data_ptr(), ent_dev.data_ptr(), out_dev.data_ptr(), stream)

    def records(self):
        return bytes(self.out_dev.cpu().numpy().tobytes())

    def clone_for(self, ctx2):
        import copy
        w = copy.copy(self)
        w.ctx = ctx2
        w.out_dev = self.torch.zeros_like(self.out_dev)
        w.sets = [(i, e, w.out_dev if j == 0 else self.torch.zeros_like(o)) for j, (i, e, o) in enumerate(self.sets)]
        w.k = 0  # its first step proves set 0 into its own out_dev: what same_results compares
        self.torch.cuda.synchronize()  # (zero-fill on torch's stream before another context's stream writes the tensor)
        return w

    def same_results(self, other):
        return all(bool((a[2] == b[2]).all()) for a, b in zip(self.sets, other.sets) if bool(b[2].any()) and bool(a[2].any()))

    def check(self):
        lib = _oracle_lib()
        out = self.records()
        for r in (0, self.B - 1):
            got = out[r * self.rec:(r + 1) * self.rec]
            ins = self.ins[r]
            rc, exp = lib.prove(ins[:224], ins[224:224 + 32 * self.N], int.from_bytes(ins[-8:], "little"), self.ents[r])
            if rc != 0 or got != exp:
                raise SystemExit("PARITY FAILURE in bench prove workload row %d" % r)
            if lib.verify(got, self.qz[r][:32], self.qz[r][32:64], self.qz[r][64:96], self.pubs[r]) != 0:
                raise SystemExit("oracle verifier rejected device proof %d" % r)

    def gather(self, dist, rank, world):
        """The one collective of the path: fixed-stride proof records to rank 0 (RCCL over xGMI)."""
        from dusk_blindbidproof_amd import sharding
        return sharding.gather_records(dist, self.out_dev, self.rec, self.B * world, rank, world)

    def check_gathered(self, gathered, world):
        """Records as rank 0 received them: the first, a middle and the last record of EVERY rank's block byte-equal to what the
        C oracle proves from that rank's inputs (re-made here from the rank's seed, witness by the oracle) and accepted by it."""
        lib = _oracle_lib()
        got = bytes(gathered.cpu().numpy().tobytes())
        assert len(got) == self.B * world * self.rec
        idxs = sorted({0, self.B // 2, self.B - 1})
        for r in range(world):
            for i, (row, ent, pub, qz) in zip(idxs, oracle_bid_rows(lib, self.seed + r, idxs, self.N)):
                rc, exp = lib.prove(row[:224], row[224:224 + 32 * self.N], int.from_bytes(row[-8:], "little"), ent)
                rec = got[(r * self.B + i) * self.rec:(r * self.B + i + 1) * self.rec]
                if rc != 0 or rec != exp:
                    raise SystemExit("PARITY FAILURE in gathered records: rank %d record %d" % (r, i))
                if lib.verify(rec, qz[:32], qz[32:64], qz[64:96], pub) != 0:
                    raise SystemExit("oracle verifier rejected gathered record: rank %d record %d" % (r, i))

    def cpu_baseline(self):
        lib = _oracle_lib()
        threads = host_threads()
        sample = 4 * threads  # ~22 core-seconds of oracle work on a 16-thread box (0.34 s per proof), ~1.4 s of wall time
        ins = b"".join(self.ins[i % self.B] for i in range(sample))
        ents = b"".join(self.ents[i % self.B] for i in range(sample))
        t0 = time.perf_counter()
        _, st = lib.prove_many(ins, ents, sample, self.N, threads)
        dt = time.perf_counter() - t0
        assert st == [0] * sample
        return {"value": sample / dt, "unit": self.unit, "cores": threads, "kind": "port",
                "sample": "%d full proves (C oracle: the reference's algorithm incl. generator folding, serial 64-bit limbs, one proof per "
                          "thread, %d threads) in %.1f s; reference's own published figure: 0.261 s per prove+verify on an i7-8559U "
                          "(docs/benchmarks.png)" % (sample, threads, dt)}


class VerifyWorkload(_Base):
    """B full verifications of device-made proofs (1 % corrupted at known indices)."""

    metric = "blind-bid verifies/sec"
    unit = "verifies/s"

    def __init__(self, ctx, bbp, torch, device, batch, items, seed, prove_wl=None):
        self.ctx, self.bbp, self.torch, self.B, self.N = ctx, bbp, torch, batch, items
        # BASELINE.json configs[3] allows distinct proofs tiled: at most 1024 are made, larger batches repeat them
        distinct = min(batch, 1024)
        pw = prove_wl if prove_wl is not None and prove_wl.B == distinct else ProveWorkload(ctx, bbp, torch, device, distinct, items, seed)
        pw.step(None)
        torch.cuda.synchronize()
        recs = pw.records()
        rec = pw.rec
        self.stride = rec + 96 + 32 * items
        rows = [bytearray(recs[(i % distinct) * rec:(i % distinct + 1) * rec] + pw.qz[i % distinct] + pw.pubs[i % distinct]) for i in range(batch)]
        self.bad = sorted(set((i * 97 + 13) % batch for i in range(max(batch // 100, 1))))
        for i in self.bad:
            rows[i][100 + (i % 900)] ^= 0x20
        self.rows = rows
        self.in_dev = _to_dev(torch, device, b"".join(bytes(r) for r in rows))
        self.ent_dev = _to_dev(torch, device, hashlib.shake_256(b"verifier-entropy%d" % seed).digest(32 * batch))
        # consecutive steps rotate over the verifier's lanes (include/bbp.h bbp_context_verify_stream): the front end of one call
        # runs under the generator MSM of another; BBP_BENCH_VERIFY_LANES=n uses only n of them (1: every step on one stream)
        avail = 0
        while avail < 8 and ctx.verify_stream(avail):
            avail += 1
        self.lanes = max(1, min(avail, int(os.environ.get("BBP_BENCH_VERIFY_LANES", avail))))
        self.lane_streams = [torch.cuda.ExternalStream(ctx.verify_stream(i), device=device) for i in range(self.lanes)]
        self.lane_status = [torch.full((batch,), -1, dtype=torch.int32, device=device) for _ in range(self.lanes)]
        self.status = self.lane_status[0]
        self.k = 0
        torch.cuda.synchronize()  # (the status tensors are filled by torch's stream, written by the lanes' streams: order them once)
        self.units_per_step = batch
        self.alg_bytes_per_step = batch * ((4135 + items) * 160 + 32)   # verify = 4135 + N terms
        self.row_additions_per_step = batch * 4098 * NAF12_DIGITS      # the fixed-base mega-check MSM launch
        self.dominant_launches_per_step = 1
        self.traffic_key = "verify_b%d_n%d" % (batch, items)
        self.config = {"workload": "%sbatch of %d full blind-bid verifications (N=%d), %d corrupted at known indices, %d distinct proofs"
                       % ("configs[3] shard: " if batch == 8192 else "", batch, items, len(self.bad), distinct),
                       "batch_per_gpu": batch, "bid_list_len": items, "parallelism": "batch-sharded, flags gathered to rank 0"}

    def clone_for(self, ctx2):
        """One verifier lane on the other context: consecutive calls are then ordered one behind the other, no two of their
        launches overlap."""
        import copy
        w = copy.copy(self)
        w.ctx = ctx2
        w.lanes = 1
        w.lane_streams = [self.torch.cuda.ExternalStream(ctx2.verify_stream(0), device=self.in_dev.device)]
        w.lane_status = [self.torch.full((self.B,), -1, dtype=self.torch.int32, device=self.in_dev.device)]
        w.status = w.lane_status[0]
        w.k = 0
        w.config = dict(self.config)
        # the fill above is a kernel on torch's CURRENT stream; the verification writes the same tensor from a lane's stream: without
        # this the fill may land after the verdicts (seen once two ranks shared a card: 256 statuses of -1)
        self.torch.cuda.synchronize()
        return w

    def same_results(self, other):
        return bool((self.status == other.status).all())

    def describe_difference(self, other):
        a, b = self.status.cpu().tolist(), other.status.cpu().tolist()
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
        return ": %d of %d statuses differ, first (index, shipped, exclusive): %r; corrupted indices %r; lanes %d, calls so far %d" % (
            len(diff), len(a), diff[:8], self.bad[:8], self.lanes, self.k)

    def _lane(self):
        i = self.k % self.lanes
        self.k += 1
        self.status = self.lane_status[i]
        return self.lane_streams[i].cuda_stream, self.status

    def step(self, stream):
        s, st = self._lane()
        self.ctx.verify_batch_dev(self.B, self.N, self.in_dev.data_ptr(), self.ent_dev.data_ptr(), st.data_ptr(), s)

    def check(self):
        self.torch.cuda.synchronize()
        st = self.status.cpu().tolist()
        exp = [0] * self.B
        for i in self.bad:
            exp[i] = None
        for i, (a, b) in enumerate(zip(st, exp)):
            if (b == 0 and a != 0) or (b is None and a not in (1, 3)):
                raise SystemExit("verify workload: wrong status %d at %d" % (a, i))
        lib = _oracle_lib()
        for i in (0, self.bad[0]):
            r = bytes(self.rows[i])
            rec = self.bbp.record_size(self.N)
            c = lib.verify(r[:rec], r[rec:rec + 32], r[rec + 32:rec + 64], r[rec + 64:rec + 96], r[rec + 96:])
            if (c == 0) != (st[i] == 0):
                raise SystemExit("verify workload: oracle disagrees at %d" % i)

    def gather(self, dist, rank, world):
        from dusk_blindbidproof_amd import sharding
        return sharding.gather_records(dist, self.status.view(self.torch.uint8), 4, self.B * world, rank, world)

    def check_gathered(self, gathered, world):
        """Flags as rank 0 received them: in EVERY rank's block exactly the known corrupted indices are rejected (every rank corrupts
        the same positions of its own proofs; rank 0's rows were also put to the oracle by check())."""
        st = gathered.cpu().view(self.torch.int32).tolist()
        assert len(st) == self.B * world
        for r in range(world):
            blk = st[r * self.B:(r + 1) * self.B]
            if [i for i, v in enumerate(blk) if v != 0] != self.bad or any(blk[i] not in (1, 3) for i in self.bad):
                raise SystemExit("gathered verify flags: rank %d block differs from the expected pattern" % r)

    def cpu_baseline(self):
        lib = _oracle_lib()
        threads = host_threads()
        sample = 16 * threads
        vin = b"".join(bytes(self.rows[i % self.B]) for i in range(sample))
        t0 = time.perf_counter()
        lib.verify_many(vin, sample, self.N, threads)
        dt = time.perf_counter() - t0
        return {"value": sample / dt, "unit": self.unit, "cores": threads, "kind": "port",
                "sample": "%d verifications (C oracle, %d threads) in %.1f s" % (sample, threads, dt)}


class VerifyAggregatedWorkload(VerifyWorkload):
    """The same B verifications (1 % corrupted) through bbp_verify_batch_aggregated_dev: groups of 32 proofs share one weighted
    generator MSM, the members of failing groups are re-verified one by one; statuses as the per-proof path reports them.
    Stream-ordered since round 2 (failing groups are compacted on the device)."""

    metric = "blind-bid verifies/sec (aggregated, SURVEY 8f-4 extension)"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.group = 32
        self.n_fallback = 0
        self.config = dict(self.config, workload=self.config["workload"] + ", aggregated in groups of %d with per-proof fallback" % self.group)

    def step(self, stream):
        s, st = self._lane()
        nf = self.ctx.verify_batch_aggregated_dev(self.B, self.N, self.in_dev.data_ptr(), self.ent_dev.data_ptr(), st.data_ptr(), self.group, s,
                                                  want_count=self.k <= self.lanes)  # the count once per lane (it synchronises), then stream-ordered
        if nf is not None:
            self.n_fallback = nf
        # MSMs the engine actually ran: one per group plus one per re-verified proof
        n_msm = (self.B + self.group - 1) // self.group + self.n_fallback
        self.row_additions_per_step = n_msm * 4098 * NAF12_DIGITS
        self.dominant_launches_per_step = 2 if self.n_fallback else 1
        self.config["reverified_per_step"] = self.n_fallback


class StreamWorkload(_Base):
    """BASELINE.json configs[4] on this rank: streaming prove + verify at sustained ingest.  Bids arrive in pinned host memory in
    the caller's raw form -- (d, k, seed) || bid list || toggle, plus prover entropy -- and one step pushes ONE chunk of B bids
    through: H2D copies and the witness pass (bbp_prepare_bids_dev) on a copy stream, prove and verify on the engine stream,
    records + flags D2H into pinned memory.  No host synchronisation on the chunk itself: the host only waits for the slot it is
    about to reuse (`depth` chunks back).  Chunk latency = device timeline from the chunk's first H2D copy to its last D2H copy.
    The distinct bids are a tile of 256 (building a million witnesses' inputs in Python would dominate the run); every chunk gets
    fresh prover entropy, so every proof differs."""

    metric = "blind-bid proofs/sec (streaming prove+verify, host ingest: PCIe-inclusive)"
    unit = "proofs/s"

    def __init__(self, ctx, bbp, torch, device, batch, items, seed, depth=3):
        import numpy as np
        self.np, self.ctx, self.bbp, self.torch, self.B, self.N, self.depth = np, ctx, bbp, torch, batch, items, depth
        C, N, tile = batch, items, min(batch, 256)
        sd = _wide(_stream(seed, 0, b"seed"))
        bids = [_stream(seed, i, b"d")[:8] + bytes(24) + _wide(_stream(seed, i, b"k")) + sd for i in range(tile)]
        lists = [b"".join(_wide(_stream(seed, i, b"pub%d" % j)) for j in range(N)) for i in range(tile)]
        ents = [b"".join(_wide(_stream(seed, i, b"ent%d" % j)) for j in range(4 + N)) + bytes(32) for i in range(tile)]
        pin = lambda data: torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
        self.h_bids = pin(b"".join(bids[i % tile] for i in range(C)))
        self.h_lists = pin(b"".join(lists[i % tile] for i in range(C)))
        self.h_tog = torch.tensor([i % N for i in range(C)], dtype=torch.int64).pin_memory()
        self.es, self.rec, self.in_stride, self.vt = bbp.entropy_size(N), bbp.record_size(N), 224 + 32 * N + 8, 96 + 32 * N
        self.ent_np = np.frombuffer(bytearray(b"".join(ents[i % tile] for i in range(C))), dtype=np.uint8).reshape(C, self.es).copy()
        # ingest side on the context's own second stream (include/bbp.h: a stream from torch's pool may share a hardware queue
        # with one of the engine's); BBP_BENCH_COPY_STREAM=torch for the comparison
        self.cp = torch.cuda.Stream() if os.environ.get("BBP_BENCH_COPY_STREAM") == "torch" else torch.cuda.ExternalStream(ctx.copy_stream, device=device)
        z = lambda n, dt=torch.uint8: torch.zeros(n, dtype=dt, device=device)
        self.slots = [dict(h_ent=torch.empty(C * self.es, dtype=torch.uint8).pin_memory(), d_bids=z(C * 96), d_lists=z(C * 32 * N),
                           d_tog=z(C, torch.int64), d_ent=z(C * self.es), d_in=z(C * self.in_stride), d_vt=z(C * self.vt), d_rec=z(C * self.rec),
                           d_vin=torch.empty((C, self.rec + self.vt), dtype=torch.uint8, device=device), d_vent=z(C * 32),
                           d_st=torch.full((C,), -1, dtype=torch.int32, device=device),
                           h_rec=torch.empty(C * self.rec, dtype=torch.uint8).pin_memory(), h_st=torch.empty(C, dtype=torch.int32).pin_memory(),
                           ev_in=torch.cuda.Event(), ev0=torch.cuda.Event(enable_timing=True), ev1=torch.cuda.Event(enable_timing=True),
                           ev_v=torch.cuda.Event(), ev_rec=torch.cuda.Event(),
                           busy=False, chunk=-1) for _ in range(depth)]
        # verification of chunk k on one of the verifier's lanes (rotating), so that it runs beside the prove of chunk k + 1 instead of
        # in line on the engine stream (round 3: 17.7 k -> see profiles/); BBP_BENCH_STREAM_VERIFY_INLINE=1 for the old shape
        self.vlanes = []
        if os.environ.get("BBP_BENCH_STREAM_VERIFY_INLINE") != "1":
            i = 0
            while ctx.verify_stream(i):
                self.vlanes.append(torch.cuda.ExternalStream(ctx.verify_stream(i), device=device))
                i += 1
        self.k = 0
        self.lat_ms, self.failed, self.done_chunks = [], 0, 0
        self.units_per_step = batch
        n1 = 1442 + 3 * items
        ref_terms = 2 * (4 + items) + 5 * n1 + 3 + 11 + 8210 + 8188
        self.alg_bytes_per_step = batch * (ref_terms * 160 + 32 * ((4 + items) + 8 + 22)) + batch * ((4135 + items) * 160 + 32)
        engine_terms = (1 + 2 * n1) * 2 + (1 + n1) - MERGED_AI_TERMS + 6 * 2 * 2049 - max(2048 - n1 - 1, 0)
        self.row_additions_per_step = batch * (engine_terms * NAF12_DIGITS + 4096 * NAF9_DIGITS + 4098 * NAF12_DIGITS)
        self.config = {"workload": "configs[4]: streaming prove+verify at sustained host ingest, chunks of %d bids (N=%d), %d chunk slots"
                       % (batch, items, depth), "batch_per_gpu": batch, "bid_list_len": items, "parallelism": "batch-sharded",
                       "ingest": "pinned host memory -> H2D -> witness on device -> prove -> verify -> D2H"}

    def _retire(self, sl):
        sl["ev1"].synchronize()
        self.lat_ms.append(sl["ev0"].elapsed_time(sl["ev1"]))
        self.failed += int((sl["h_st"] != 0).sum())
        self.done_chunks += 1
        sl["busy"] = False

    def _stage(self, k):
        """Host side of the ingest for chunk k, issued one chunk AHEAD of its prove call: with the GPU saturated a copy or the
        witness kernel can wait milliseconds for its turn, and the prover's opening stage needs complete inputs when it starts."""
        torch, C, N = self.torch, self.B, self.N
        sl = self.slots[k % self.depth]
        if sl["busy"]:
            self._retire(sl)  # the only host wait: the slot used `depth` chunks ago
        seeds = self.np.frombuffer(hashlib.shake_256(b"chunk%d" % k).digest(32 * C), dtype=self.np.uint8).reshape(C, 32)
        self.ent_np[:, self.es - 32:] = seeds
        sl["h_ent"].copy_(torch.from_numpy(self.ent_np.reshape(-1)))
        with torch.cuda.stream(self.cp):
            sl["ev0"].record(self.cp)
            sl["d_bids"].copy_(self.h_bids, non_blocking=True)
            sl["d_lists"].copy_(self.h_lists, non_blocking=True)
            sl["d_tog"].copy_(self.h_tog, non_blocking=True)
            sl["d_ent"].copy_(sl["h_ent"], non_blocking=True)
            # witness + row assembly on the copy stream: the prover's opening stage waits for this pass by itself (bbp.h), and
            # it runs behind the entropy copy on the same stream, so every prover input is complete when that stage starts
            self.ctx.prepare_bids_dev(C, N, sl["d_bids"].data_ptr(), sl["d_lists"].data_ptr(), sl["d_tog"].data_ptr(), sl["d_in"].data_ptr(),
                                      sl["d_vt"].data_ptr(), self.cp.cuda_stream)
            sl["ev_in"].record(self.cp)
        sl["chunk"] = k

    def step(self, stream):
        torch, C, N = self.torch, self.B, self.N
        sl = self.slots[self.k % self.depth]
        if sl["chunk"] != self.k:
            self._stage(self.k)  # first step, or the first after a drain
        eng = torch.cuda.current_stream()
        self.ctx.prove_batch_dev(C, N, sl["d_in"].data_ptr(), sl["d_ent"].data_ptr(), sl["d_rec"].data_ptr(), stream)
        self._stage(self.k + 1)  # after the prove call (which waited for ITS bid pass), so the next chunk's ingest runs under this one
        eng.wait_event(sl["ev_in"])  # the verifier tails
        sl["d_vin"][:, :self.rec] = sl["d_rec"].view(C, self.rec)
        sl["d_vin"][:, self.rec:] = sl["d_vt"].view(C, self.vt)
        if self.vlanes:
            vs = self.vlanes[self.k % len(self.vlanes)]
            sl["ev_v"].record(eng)                      # records and verifier rows assembled
            sl["h_rec"].copy_(sl["d_rec"], non_blocking=True)
            sl["ev_rec"].record(eng)
            vs.wait_event(sl["ev_v"])
            self.ctx.verify_batch_dev(C, N, sl["d_vin"].data_ptr(), sl["d_vent"].data_ptr(), sl["d_st"].data_ptr(), vs.cuda_stream)
            with torch.cuda.stream(vs):
                sl["h_st"].copy_(sl["d_st"], non_blocking=True)
                vs.wait_event(sl["ev_rec"])             # the chunk is done when its flags AND its records are on the host
                sl["ev1"].record(vs)
        else:
            self.ctx.verify_batch_dev(C, N, sl["d_vin"].data_ptr(), sl["d_vent"].data_ptr(), sl["d_st"].data_ptr(), stream)
            sl["h_rec"].copy_(sl["d_rec"], non_blocking=True)
            sl["h_st"].copy_(sl["d_st"], non_blocking=True)
            sl["ev1"].record(eng)
        sl["busy"] = True
        self.k += 1

    def drain(self):
        for i in range(self.depth):
            sl = self.slots[(self.k + i) % self.depth]
            if sl["busy"]:
                self._retire(sl)
        self.torch.cuda.synchronize()
        for sl in self.slots:
            sl["chunk"] = -1  # the chunk staged ahead is dropped: whatever step comes next stages its own (fresh ev0)

    def check(self):
        if self.failed:
            raise SystemExit("stream workload: %d failed verifications" % self.failed)
        lib = _oracle_lib()
        sl = self.slots[(self.k - 1) % self.depth]
        row = bytes(sl["d_in"][:self.in_stride].cpu().numpy().tobytes())
        ent = bytes(sl["h_ent"][:self.es].numpy().tobytes())
        rc, exp = lib.prove(row[:224], row[224:224 + 32 * self.N], int.from_bytes(row[-8:], "little"), ent)
        if rc != 0 or bytes(sl["h_rec"][:self.rec].numpy().tobytes()) != exp:
            raise SystemExit("PARITY FAILURE in stream workload (chunk %d, record 0)" % sl["chunk"])
        self.lat_ms.clear()

    def extra_report(self, timings):
        lat = sorted(self.lat_ms)
        pick = lambda q: lat[min(len(lat) - 1, int(len(lat) * q))] if lat else None
        return {"kernels_us": _kernel_table(timings), "failed_verifications": self.failed, "chunks": len(lat),
                "chunk_latency_ms": {"p50": pick(0.5), "p99": pick(0.99), "max": lat[-1] if lat else None,
                                     "definition": "device timeline, first H2D copy of the chunk -> last D2H copy of its records and flags"}}

    def gather(self, dist, rank, world):
        from dusk_blindbidproof_amd import sharding
        sl = self.slots[(self.k - 1) % self.depth]
        return sharding.gather_records(dist, sl["d_st"].view(self.torch.uint8), 4, self.B * world, rank, world)

    def check_gathered(self, gathered, world):
        st = gathered.cpu().view(self.torch.int32)
        if st.numel() != self.B * world or int((st != 0).sum()):
            raise SystemExit("gathered stream flags: %d failed verifications" % int((st != 0).sum()))

    def cpu_baseline(self):
        lib = _oracle_lib()
        threads = host_threads()
        sample = 2 * threads
        sl = self.slots[(self.k - 1) % self.depth]
        ins = bytes(sl["d_in"][:sample * self.in_stride].cpu().numpy().tobytes())
        ents = bytes(sl["h_ent"][:sample * self.es].numpy().tobytes())
        t0 = time.perf_counter()
        recs, st = lib.prove_many(ins, ents, sample, self.N, threads)
        vt = bytes(sl["d_vt"][:sample * self.vt].cpu().numpy().tobytes())
        vin = b"".join(recs[i * self.rec:(i + 1) * self.rec] + vt[i * self.vt:(i + 1) * self.vt] for i in range(sample))
        vst = lib.verify_many(vin, sample, self.N, threads)
        dt = time.perf_counter() - t0
        assert st == [0] * sample and vst == [0] * sample
        return {"value": sample / dt, "unit": self.unit, "cores": threads, "kind": "port",
                "sample": "%d bids proved then verified (C oracle, one per thread, %d threads) in %.1f s" % (sample, threads, dt)}


def make_workload(name, ctx, bbp, torch, device, batch, items, seed):
    if name == "stream":
        return StreamWorkload(ctx, bbp, torch, device, batch, items, seed, depth=int(os.environ.get("BBP_BENCH_STREAM_DEPTH", "3")))
    if name in ("auto", "prove"):
        return ProveWorkload(ctx, bbp, torch, device, batch, items, seed)
    if name == "msm":
        return MsmWorkload(ctx, bbp, torch, device, batch, items, seed)
    if name == "verify":
        return VerifyWorkload(ctx, bbp, torch, device, batch, items, seed)
    if name == "verify_aggregated":
        return VerifyAggregatedWorkload(ctx, bbp, torch, device, batch, items, seed)
    raise SystemExit("unknown workload %r" % name)
