"""Synthetic workloads for bench.py (SURVEY.md 8d recipes). Inputs live in HBM before the timed region starts."""
import ctypes
import os
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
TAG_NAMES = {1: "k_msm", 2: "k_encode", 3: "k_witness", 4: "k_rng", 5: "k_poly", 6: "k_ipa_scalars", 7: "k_commit",
             8: "k_transcript", 9: "k_verify_scalars", 10: "k_varbase"}


def _oracle_lib():
    """C restatement under oracle/ -- used ONLY as checker and as the timed CPU baseline."""
    import __graft_entry__ as ge
    path = ge.build_oracle()
    if not path or not os.path.exists(path):
        return None
    from tests import oracle_c
    return oracle_c.load(path)


class MsmWorkload:
    """BASELINE.json configs[1]: B proofs x (A_I1: 1+2n1, A_O1: 1+n1, S1: 1+2n1 terms), n1 = 1442 + 3N."""

    metric = "blind-bid proofs/sec (commitment-MSM stage: A_I1+A_O1+S1 per proof)"
    unit = "proofs/s"
    dominant_tag = 1
    dominant_kernel = "k_msm"
    measured_traffic_bytes = None

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
        self.dominant_alg_bytes_per_launch = self.alg_bytes_per_step / len(self.shapes)
        self.config = {"workload": "configs[1]: batch of %d blind-bid proofs, commitment MSMs only (N=%d: %s terms)"
                       % (batch, items, "+".join(str(n) for n, _ in self.shapes)),
                       "batch_per_gpu": batch, "bid_list_len": items, "msm_window_bits": 11, "parallelism": "batch-sharded"}

    def step(self, stream):
        for (n, layout), s, o in zip(self.shapes, self.scal, self.out):
            self.ctx.msm_batch_dev(self.B, n, s.data_ptr(), layout, o.data_ptr(), stream)

    def check(self):
        lib = _oracle_lib()
        rows = [0, self.B - 1]
        for (n, layout), s, o in zip(self.shapes, self.scal, self.out):
            for r in rows:
                sc = bytes(s[r].cpu().numpy().tobytes())
                got = bytes(o[r].cpu().numpy().tobytes())
                if lib is not None:
                    exp = lib.msm_layout(sc, n, layout)
                else:
                    from oracle.ref_py import blindbid as bb, ristretto as rs
                    pc, bp = bb.gens(2048)
                    m = (n - 1) // 2 if layout == 0 else n - 1
                    bases = [pc.B_blinding] + bp.G[:m] + (bp.H[:m] if layout == 0 else [])
                    exp = rs.encode(rs.msm([int.from_bytes(sc[32 * i:32 * i + 32], "little") for i in range(n)], bases))
                if got != exp:
                    raise SystemExit("PARITY FAILURE in bench msm workload row %d" % r)

    def gather(self, dist, rank, world):
        t = self.out[0]
        if rank == 0:
            bufs = [self.torch.empty_like(t) for _ in range(world)]
            dist.gather(t, bufs, dst=0)
        else:
            dist.gather(t, None, dst=0)

    def extra_report(self, timings):
        agg = {}
        for tag, us in timings:
            a = agg.setdefault(TAG_NAMES.get(tag, str(tag)), [0, 0.0])
            a[0] += 1
            a[1] += us
        return {"kernels_us": {k: {"launches": v[0], "total_us": round(v[1], 1)} for k, v in agg.items()}}

    def cpu_baseline(self):
        lib = _oracle_lib()
        if lib is None:
            return None
        threads = os.cpu_count() or 1
        sample = max(threads, 8)
        rows = [bytes(self.scal[i][r % self.B].cpu().numpy().tobytes()) for r in range(sample) for i in range(3)]
        t0 = time.perf_counter()
        lib.msm_layout_many(rows, [self.shapes[i % 3][0] for i in range(len(rows))],
                            [self.shapes[i % 3][1] for i in range(len(rows))], threads)
        dt = time.perf_counter() - t0
        return {"value": sample / dt, "unit": self.unit, "cores": threads, "kind": "port",
                "sample": "%d proofs' A_I1+A_O1+S1 MSMs (Pippenger, C oracle, %d threads) in %.1f s" % (sample, threads, dt)}


def make_workload(name, ctx, bbp, torch, device, batch, items, seed):
    if name in ("auto", "msm"):
        return MsmWorkload(ctx, bbp, torch, device, batch, items, seed)
    raise SystemExit("unknown workload %r" % name)
