"""Synthetic workloads for bench.py (SURVEY.md 8d recipes). Inputs live in HBM before the timed region starts."""
import hashlib
import os
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
TAG_NAMES = {1: "k_msm_acc", 11: "k_msm_sort", 2: "k_encode", 3: "k_witness_head", 4: "k_open_serial(witness+rng)", 5: "k_poly/powers/flatten", 6: "k_ipa_round", 7: "k_commit",
             8: "k_transcript", 9: "k_vscalars", 10: "k_varbase/tail"}
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


def synth_bids(ctx, B, N, seed):
    """d = uniform u64, k uniform scalar, ONE seed per batch, witness on the device, x_i placed at toggle_i = i mod N."""
    sd = _wide(_stream(seed, 0, b"seed"))
    dks = b"".join(_stream(seed, i, b"d")[:8] + bytes(24) + _wide(_stream(seed, i, b"k")) + sd for i in range(B))
    w = ctx.witness_batch(dks)
    ins, ents, pubs, qz = [], [], [], []
    for i in range(B):
        m, x, y, yi, q, z = (w[192 * i + 32 * j:192 * i + 32 * j + 32] for j in range(6))
        toggle = i % N
        pub = [_wide(_stream(seed, i, b"pub%d" % j)) for j in range(N)]
        pub[toggle] = x
        pub = b"".join(pub)
        ins.append(dks[96 * i:96 * i + 64] + y + yi + q + z + sd + pub + toggle.to_bytes(8, "little"))
        ents.append(b"".join(_wide(_stream(seed, i, b"ent%d" % j)) for j in range(4 + N)) + _stream(seed, i, b"entseed")[:32])
        pubs.append(pub)
        qz.append(q + z + sd)
    return ins, ents, pubs, qz


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


# HBM bytes per accumulate-kernel launch (k_msm_acc, a third of the batch per launch) from the rocprofv3 PMC passes on B = 1024,
# N = 8 (profiles/r01_rocprofv3_pmc_hbm.csv, weighted over the <0> and <1> instances): FETCH_SIZE x 2 (gfx950 reports half of
# wide reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE
MEASURED_TRAFFIC_PROVE_1024_8 = (2 * 1651238 + 508352) * 1024


class _Base:
    dominant_tag = 1
    dominant_kernel = "k_msm_acc"
    measured_traffic_bytes = None

    def extra_report(self, timings):
        return {"kernels_us": _kernel_table(timings)}

    def gather(self, dist, rank, world):
        pass


class MsmWorkload(_Base):
    """BASELINE.json configs[1]: B proofs x (A_I1: 1+2n1, A_O1: 1+n1, S1: 1+2n1 terms), n1 = 1442 + 3N."""

    metric = "blind-bid proofs/sec (commitment-MSM stage only: A_I1+A_O1+S1 per proof)"
    unit = "proofs/s"

    def __init__(self, ctx, bbp, torch, device, batch, items, seed):
        self.ctx, self.bbp, self.torch, self.B = ctx, bbp, torch, batch
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
        self.config = {"workload": "configs[1]: batch of %d blind-bid proofs, commitment MSMs only (N=%d: %s terms)"
                       % (batch, items, "+".join(str(n) for n, _ in self.shapes)),
                       "batch_per_gpu": batch, "bid_list_len": items, "msm_recoding": "NAF-12", "parallelism": "batch-sharded"}

    def step(self, stream):
        for (n, layout), s, o in zip(self.shapes, self.scal, self.out):
            self.ctx.msm_batch_dev(self.B, n, s.data_ptr(), layout, o.data_ptr(), stream)

    def check(self):
        lib = _oracle_lib()
        for (n, layout), s, o in zip(self.shapes, self.scal, self.out):
            for r in (0, self.B - 1):
                sc = bytes(s[r].cpu().numpy().tobytes())
                got = bytes(o[r].cpu().numpy().tobytes())
                if got != lib.msm_layout(sc, n, layout):
                    raise SystemExit("PARITY FAILURE in bench msm workload row %d" % r)

    def gather(self, dist, rank, world):
        t = self.out[0]
        dist.gather(t, [self.torch.empty_like(t) for _ in range(world)] if rank == 0 else None, dst=0)

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
        self.ctx, self.bbp, self.torch, self.B, self.N, self.device = ctx, bbp, torch, batch, items, device
        self.ins, self.ents, self.pubs, self.qz = synth_bids(ctx, batch, items, seed)
        self.in_dev = _to_dev(torch, device, b"".join(self.ins))
        self.ent_dev = _to_dev(torch, device, b"".join(self.ents))
        self.rec = bbp.record_size(items)
        self.out_dev = torch.zeros(batch * self.rec, dtype=torch.uint8, device=device)
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
        self.measured_traffic_bytes = None  # set from profiles/ (rocprofv3 PMC passes) for the B = 1024, N = 8 configuration
        if batch == 1024 and items == 8:
            self.measured_traffic_bytes = MEASURED_TRAFFIC_PROVE_1024_8
        self.config = {"workload": "configs[2]: batch of %d full blind-bid R1CS proves (N=%d, 1466 multipliers, 11 IPA rounds)" % (batch, items),
                       "batch_per_gpu": batch, "bid_list_len": items, "msm_recoding": "NAF-12", "parallelism": "batch-sharded",
                       "ref_msm_terms_per_proof": ref_terms, "engine_msm_terms_per_proof": engine_terms + 4096}

    def step(self, stream):
        self.ctx.prove_batch_dev(self.B, self.N, self.in_dev.data_ptr(), self.ent_dev.data_ptr(), self.out_dev.data_ptr(), stream)

    def records(self):
        return bytes(self.out_dev.cpu().numpy().tobytes())

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

    def cpu_baseline(self):
        lib = _oracle_lib()
        threads = host_threads()
        sample = 2 * threads
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
        pw = prove_wl or ProveWorkload(ctx, bbp, torch, device, batch, items, seed)
        pw.step(0)
        torch.cuda.synchronize()
        recs = pw.records()
        rec = pw.rec
        self.stride = rec + 96 + 32 * items
        rows = [bytearray(recs[i * rec:(i + 1) * rec] + pw.qz[i] + pw.pubs[i]) for i in range(batch)]
        self.bad = sorted(set((i * 97 + 13) % batch for i in range(max(batch // 100, 1))))
        for i in self.bad:
            rows[i][100 + (i % 900)] ^= 0x20
        self.rows = rows
        self.in_dev = _to_dev(torch, device, b"".join(bytes(r) for r in rows))
        self.ent_dev = _to_dev(torch, device, hashlib.shake_256(b"verifier-entropy%d" % seed).digest(32 * batch))
        self.status = torch.full((batch,), -1, dtype=torch.int32, device=device)
        self.units_per_step = batch
        self.alg_bytes_per_step = batch * ((4135 + items) * 160 + 32)   # verify = 4135 + N terms
        self.row_additions_per_step = batch * 4098 * NAF12_DIGITS      # the fixed-base mega-check MSM launch
        self.dominant_launches_per_step = 1
        self.config = {"workload": "batch of %d full blind-bid verifications (N=%d), %d corrupted" % (batch, items, len(self.bad)),
                       "batch_per_gpu": batch, "bid_list_len": items, "parallelism": "batch-sharded"}

    def step(self, stream):
        self.ctx.verify_batch_dev(self.B, self.N, self.in_dev.data_ptr(), self.ent_dev.data_ptr(), self.status.data_ptr(), stream)

    def check(self):
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
    The call synchronises the stream itself (the host reads the group verdicts)."""

    metric = "blind-bid verifies/sec (aggregated, SURVEY 8f-4 extension)"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.group = 32
        self.n_fallback = 0
        self.config = dict(self.config, workload=self.config["workload"] + ", aggregated in groups of %d with per-proof fallback" % self.group)

    def step(self, stream):
        self.n_fallback = self.ctx.verify_batch_aggregated_dev(self.B, self.N, self.in_dev.data_ptr(), self.ent_dev.data_ptr(),
                                                               self.status.data_ptr(), self.group, stream)
        # MSMs the engine actually ran: one per group plus one per re-verified proof
        n_msm = (self.B + self.group - 1) // self.group + self.n_fallback
        self.row_additions_per_step = n_msm * 4098 * NAF12_DIGITS
        self.dominant_launches_per_step = 2 if self.n_fallback else 1
        self.config["reverified_per_step"] = self.n_fallback


def make_workload(name, ctx, bbp, torch, device, batch, items, seed):
    if name in ("auto", "prove"):
        return ProveWorkload(ctx, bbp, torch, device, batch, items, seed)
    if name == "msm":
        return MsmWorkload(ctx, bbp, torch, device, batch, items, seed)
    if name == "verify":
        return VerifyWorkload(ctx, bbp, torch, device, batch, items, seed)
    if name == "verify_aggregated":
        return VerifyAggregatedWorkload(ctx, bbp, torch, device, batch, items, seed)
    raise SystemExit("unknown workload %r" % name)
