#!/usr/bin/env python3
"""bench.py -- blind-bid Bulletproofs hot path on MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W [--workload prove|verify|verify_aggregated|msm|stream] [--batch B] [--items N_ITEMS]

A "step" is one pass of the hot path over one batch of B synthetic bids whose inputs are already resident in HBM.
  workload prove (default): BASELINE.json configs[2] -- full R1CS prove of B = 1024 bids (gadget witness, Merlin, commitment
                   MSMs, polynomial sweep, 11 IPA rounds), proof records out.  `value` = proofs/s.
  workload verify: B full verifications (one 4098-term fixed-base MSM + ~45 proof points each); `--batch 8192` is one GPU's
                   shard of BASELINE.json configs[3] (65 536 verifications over 8 GPUs, flags gathered to rank 0).
  workload msm   : configs[1] -- per proof only the three commitment MSMs A_I1/A_O1/S1 (2933 + 1467 + 2933 terms at N = 8).
  workload stream: configs[4] -- sustained ingest: bids arrive in pinned host memory, each step pushes one chunk of B bids
                   through H2D copy -> witness -> prove -> verify -> D2H with no host synchronisation on the chunk itself;
                   reports sustained proofs/s (= verifies/s) and p50 / p99 chunk latency.  PCIe-inclusive by nature.
After the timed region the default run also measures the other workloads for a few steps and reports them as `also` (same
JSON line), so one run carries proofs/s, verifies/s and the MSM-stage rate.
Each rank works on its own B proofs (independent units, no data-path collective; weak scaling); the only collective is the
final gather of proof records / flags to rank 0.  cpu_baseline (rank 0, N=1 only) times the C oracle on the host cores.

Launch: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (fresh child processes,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, backend nccl = RCCL) before this process has touched the GPU, relays rank 0's JSON
line and fails if any rank fails or the line does not say n_gpus == N.  Under torchrun (WORLD_SIZE set) it is one of the ranks;
--gpus must then equal WORLD_SIZE.
"""
import os

# The engine keeps four HIP streams busy per GPU (the caller's, the opening stage's, two more heavy-stage slices) -- HIP's default
# number of hardware queues.  Any other stream used before it (torch.distributed's RCCL stream in the multi-rank run) makes
# two of them share a queue and serialise: measured 84.5 vs 61.6 ms per batch (tools/hwq_probe.py).  Must be set before the HIP
# runtime initialises, i.e. before the first torch.cuda call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import argparse
import json
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
STUB = os.environ.get("BBP_BENCH_STUB") == "1"  # launcher-plumbing test on a CPU box: gloo, no engine, NOT a measurement


def synth_scalars_device(torch, n_rows, n_terms, seed, device):
    """Uniform canonical scalars, generated on the device: 252 random bits each (< 2^252 < l)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.randint(0, 2**31 - 1, (n_rows, n_terms, 8), generator=g, device=device, dtype=torch.int64)
    w2 = torch.randint(0, 2, (n_rows, n_terms, 8), generator=g, device=device, dtype=torch.int64)
    w = (w | (w2 << 31)) & 0xFFFFFFFF
    w[..., 7] &= 0x0FFFFFFF
    return w.to(torch.int32).contiguous()  # bit pattern of u32 LE limbs


def timed(wl, ctx, torch, dist, world, steps, stream):
    def barrier():
        if world > 1:
            dist.barrier()
        if not STUB:
            torch.cuda.synchronize()
    if ctx is not None:
        ctx.set_profiling(True)
        ctx.last_timings()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step(stream)
    wl.drain()
    barrier()
    dt = time.perf_counter() - t0
    timings = []
    if ctx is not None:
        timings = ctx.last_timings()
        ctx.set_profiling(False)
    return dt, timings


def roofline(wl, timings, steps, alu_peak=None, wall_s=None):
    dom = [us for tag, us in timings if tag == wl.dominant_tag]
    avg_us = sum(dom) / max(len(dom), 1)
    # algorithmic bytes of the step's dominant-kernel work, spread over the launches that were actually observed
    # (the engine may cut a batch into slices on several streams, multiplying the launch count)
    alg_per_launch = wl.alg_bytes_per_step * steps / max(len(dom), 1)
    achieved = alg_per_launch / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    traffic_step, traffic_src = wl.measured_traffic()  # per step, from the named profile; per launch = spread over what this run launched
    traffic = traffic_step * steps / max(len(dom), 1) if traffic_step else None
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "traffic": traffic, "traffic_source": traffic_src, "kernel": wl.dominant_kernel, "avg_launch_us": avg_us, "launches": len(dom),
           "alg_bytes_per_launch": alg_per_launch,
           "note": "modular-integer path: the binding roofline is 32-bit integer multiply issue, not HBM (DESIGN.md section 5); "
                   "avg_launch_us is co-residency-stretched when slices overlap -- the exclusive figure is in the named profile"}
    if traffic and wall_s and dom:
        # measured HBM-side bytes of the dominant kernel over the WALL time of the region (launches x traffic per launch)
        out["hbm_side_GBps_whole_step"] = traffic_step * steps / wall_s / 1e9
    if alu_peak:
        # secondary, honest roofline (SURVEY.md 7 hard part 2): table-row additions the MSM kernels actually perform per second
        # (one per non-zero NAF digit) against the register-resident mixed-addition rate bbp_ubench measures live (no memory traffic)
        adds = wl.row_additions_per_step * steps / (sum(dom) * 1e-6) if dom else 0.0
        out["alu"] = {"bound": "v_mad_i64_i32 issue", "achieved": adds, "peak": alu_peak, "unit": "point additions/s",
                      "frac": adds / alu_peak}
        if wall_s:
            # launches of the concurrent slices overlap, so their summed durations exceed wall time; this is the same count of
            # additions over the WALL time of the timed region: what the whole pipeline gets out of the integer units
            out["alu"]["whole_step_frac"] = wl.row_additions_per_step * steps / wall_s / alu_peak
    return out


def exclusive_context(bbp, dev_index):
    """A second context created with BBP_SLICES=1 and BBP_VERIFY_OVERLAP=0 (the environment is read at bbp_init): one heavy-stage
    stream for the prover, and the verifier's variable-base kernels in line on the caller's stream instead of beside its MSM -- with
    them on a side stream the "exclusive" accumulate launch of a verification shared the GPU with 2.3 ms of other kernels."""
    knobs = {"BBP_SLICES": "1", "BBP_VERIFY_OVERLAP": "0"}
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        return bbp.Context(dev_index)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def exclusive_pass(wl, bbp, torch, device, dev_index, alu_peak, steps=3, ctx2=None):
    """The reproducible roofline figure.  The shipped schedule cuts a batch into slices whose launches overlap in time, so the
    per-launch durations behind `roofline.frac` are stretched by co-residency.  Here the same inputs go through a SECOND context
    created with BBP_SLICES=1 (verifier: one lane): no two launches of the dominant kernel overlap, a launch covers the whole batch,
    and (summed launch time of the dominant kernel per step) is what rocprofv3's exclusive kernel-stats pass shows as well
    (profiles/README.md).  Results must equal the shipped schedule's byte for byte."""
    own = ctx2 is None
    if own:
        ctx2 = exclusive_context(bbp, dev_index)
    try:
        w2 = wl.clone_for(ctx2)
        if w2 is None:
            return None
        s2 = torch.cuda.ExternalStream(ctx2.stream, device=device)
        with torch.cuda.stream(s2):
            w2.step(s2.cuda_stream)
            w2.drain()
            torch.cuda.synchronize()
            if os.environ.get("BBP_BENCH_NO_CHECK") != "1" and not wl.same_results(w2):
                raise SystemExit("PARITY FAILURE: the exclusive (one-slice) schedule and the shipped schedule disagree" + wl.describe_difference(w2))
            ctx2.set_profiling(True)
            ctx2.last_timings()
            t0 = time.perf_counter()
            for _ in range(steps):
                w2.step(s2.cuda_stream)
            w2.drain()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            timings = ctx2.last_timings()
            ctx2.set_profiling(False)
        dom = [us for tag, us in timings if tag == wl.dominant_tag]
        dom_ms = sum(dom) / steps / 1e3
        achieved = wl.alg_bytes_per_step / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        out = {"frac": achieved / HBM_PEAK_GBPS, "achieved": achieved, "unit": "GB/s", "dominant_ms_per_step": dom_ms, "launches_per_step": len(dom) / steps,
               "steps": steps, "ms_per_step": wall / steps * 1e3,
               "how": "second context with BBP_SLICES=1 (verifier: one lane, variable-base kernels in line), same inputs, results byte-equal to the shipped schedule's; "
                      "achieved = algorithmic bytes per step / summed launch time of %s per step" % wl.dominant_kernel}
        if alu_peak and dom_ms > 0:
            out["alu_frac"] = wl.row_additions_per_step / (dom_ms * 1e-3) / alu_peak
        del w2, s2
        return out
    finally:
        import gc
        torch.cuda.synchronize()
        gc.collect()
        if own:
            ctx2.close()


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher: start N fresh ranks (this process has not touched the GPU), relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if not args.no_build and not STUB:
        build_all()  # once, in the parent, before any rank exists; the ranks then only check
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ["--no-build"], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained on a thread while this loop watches every rank: one rank dying must not leave the others (and
    # this process) waiting in a rendezvous or barrier forever
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            time.sleep(2.0)  # let the others notice by themselves first
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = b"".join(chunks).decode()
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %r\n" % rcs)
        sys.stdout.write(out0)
        return 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write("bench.py: expected ONE JSON line from rank 0, got %d\n" % len(lines))
        return 1
    if json.loads(lines[0]).get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: rank 0 reports n_gpus=%r, --gpus=%d\n" % (json.loads(lines[0]).get("n_gpus"), args.gpus))
        return 1
    print(lines[0], flush=True)
    return 0


def build_all(check_only=False):
    """Compile (or, with check_only, insist on up-to-date) native artefacts.  Runs BEFORE the first torch.cuda / HIP call of this
    process, under a file lock so that concurrently started ranks build once: a profiler's preloaded library must never see this
    process spawn a compiler after the GPU is initialised (tools/profile_round.sh passes --no-build)."""
    import fcntl
    import __graft_entry__ as ge
    if check_only:
        stale = ge.stale_artefacts()
        if stale:
            raise SystemExit("bench.py --no-build: stale or missing artefacts %r (run `python __graft_entry__.py` first)" % stale)
        return
    with open(os.path.join(ROOT, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        ge.build_hip()
        ge.build_oracle()


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=os.environ.get("BBP_BENCH_WORKLOAD", "prove"))
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--items", type=int, default=8, help="bid-list length N (SURVEY.md 8d default 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary verify / msm-stage measurements")
    ap.add_argument("--no-build", action="store_true", help="do not compile anything: fail if the native artefacts are stale")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the exclusive (one-slice) pass behind roofline.exclusive")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a run of a different size" % (args.gpus, world))
    if not STUB:
        build_all(check_only=args.no_build)  # before anything touches the GPU

    import torch
    import torch.distributed as dist

    # rehearsal knobs (one-GPU box): BBP_BENCH_BACKEND=gloo BBP_BENCH_DEVICE=0 run N ranks against a single card
    backend = "gloo" if STUB else os.environ.get("BBP_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("BBP_BENCH_DEVICE", local_rank))
    device = torch.device("cpu") if STUB else torch.device("cuda", dev_index)
    if not STUB:
        torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    red_dev = device if backend == "nccl" else torch.device("cpu")

    from bench_workloads import make_workload, MsmWorkload, VerifyWorkload, VerifyAggregatedWorkload, StubWorkload
    if STUB:
        if os.environ.get("BBP_BENCH_STUB_FAIL_RANK") == str(rank):
            sys.exit(3)  # tests/test_bench_launcher.py: a dying rank must fail the whole launch
        ctx, bbp, wl, stream, engine_stream = None, None, StubWorkload(args.batch), None, None
    else:
        import dusk_blindbidproof_amd as bbp
        if os.environ.get("BBP_BENCH_STREAM") == "torch_first":
            pre_stream = torch.cuda.Stream()  # experiment: torch's stream pool exists before the engine's streams
        ctx = bbp.Context(dev_index)
        wl = make_workload(args.workload, ctx, bbp, torch, device, args.batch, args.items, seed=1 + rank)
        # a real caller stream (not handle 0): the ordering contract of include/bbp.h is exercised, and torch's own work of this
        # script (copies, clones in the workloads) is issued on the same stream
        mode = os.environ.get("BBP_BENCH_STREAM", "external")
        if mode == "context":    # experiment knobs (DESIGN.md section 4 "whose stream")
            engine_stream, stream = None, None
        elif mode == "torch_first":
            engine_stream = pre_stream
            torch.cuda.set_stream(engine_stream)
            stream = engine_stream.cuda_stream
        elif mode == "torch":
            engine_stream = torch.cuda.Stream()
            torch.cuda.set_stream(engine_stream)
            stream = engine_stream.cuda_stream
        else:
            engine_stream = torch.cuda.ExternalStream(ctx.stream, device=device)
            torch.cuda.set_stream(engine_stream)
            stream = engine_stream.cuda_stream
    for _ in range(args.warmup):
        wl.step(stream)
    wl.drain()
    if not STUB:
        torch.cuda.synchronize()
    if os.environ.get("BBP_BENCH_NO_CHECK") != "1":  # only for timing experiments with deliberately wrong variants (tools/build_variant.py)
        wl.check()  # parity of the warmed-up output against the oracle on a sample (not timed)

    alu_peak = None if STUB else max(ctx.ubench(3, 8192, 2000) for _ in range(2))  # register-resident ge_madd chains: the integer-ALU ceiling
    # ... and the same loop with a table operand the compiler cannot see through (kind 5: the 19 x limb products of a freshly gathered
    # row cannot be hoisted): the ceiling the accumulate loop can actually be held against.  Both run the engine's own field
    # multiplication, so they move with it (round 4: +12 %) -- read `frac` / `dominant_ms_per_step` for progress, `alu_frac` for distance.
    alu_peak_fresh = None if STUB else max(ctx.ubench(5, 8192, 2000) for _ in range(2))
    dt, timings = timed(wl, ctx, torch, dist, world, args.steps, stream)
    tmax = torch.tensor([dt], device=red_dev, dtype=torch.float64)
    per_rank_s = [dt]
    if world > 1:
        # every rank's own clock beside the max-over-ranks figure: a straggler GPU (clocks, a shared PCIe switch, a noisy neighbour)
        # is visible in the one SCALE line instead of only dragging `value` down
        mine = tmax.clone()
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_s = [float(t.item()) for t in allt]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # The path's one collective: every rank's proof records / flags to rank 0 in global proof order, after the timed
    # region.  Timed on its own between barriers -- the first call also builds the point-to-point connections, the second is the
    # steady-state cost -- and the tensor rank 0 received is CHECKED: a sample from every rank's block against the oracle.
    gather_info = None
    if not STUB:
        gms = []
        for _ in range(2 if world > 1 else 1):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            gathered = wl.gather(dist, rank, world)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            gms.append((time.perf_counter() - t0) * 1e3)
        if rank == 0 and gathered is not None:
            if os.environ.get("BBP_BENCH_NO_CHECK") != "1":
                wl.check_gathered(gathered, world)
            gather_info = {"gather_ms": gms[-1], "gather_first_ms": gms[0], "bytes": int(gathered.numel()), "ranks": world,
                           "backend": backend if world > 1 else "none (single rank: local copy)", "checked": "sample of every rank's block vs the C oracle"}
        del gathered

    exclusive, ctx_excl = None, None
    if rank == 0 and not STUB and not args.no_exclusive:
        ctx_excl = exclusive_context(bbp, dev_index)  # kept for the secondary workloads' exclusive passes below
        exclusive = exclusive_pass(wl, bbp, torch, device, dev_index, alu_peak, ctx2=ctx_excl)

    also = {}
    if args.workload == "prove" and not args.no_also and not STUB:
        for name, cls in (("verify", VerifyWorkload), ("verify_aggregated", VerifyAggregatedWorkload), ("msm_stage", MsmWorkload)):
            kw = {"prove_wl": wl} if issubclass(cls, VerifyWorkload) else {}
            w2 = cls(ctx, bbp, torch, device, args.batch, args.items, 1 + rank, **kw)
            w2.step(stream)
            torch.cuda.synchronize()
            w2.check()
            # enough calls for the four verifier lanes / the MSM pipeline to reach their steady state (12 calls on four lanes read
            # 13 % low: 170 k instead of 196 k verifications/s; 48 calls still 3-4 % under what `--workload verify` reads over 60):
            # eight untimed calls first, then 96
            ks = 96 if issubclass(cls, VerifyWorkload) else 12
            if issubclass(cls, VerifyWorkload):
                for _ in range(8):
                    w2.step(stream)
                w2.drain()
                torch.cuda.synchronize()
            d2, t2 = timed(w2, ctx, torch, dist, world, ks, stream)
            t2max = torch.tensor([d2], device=red_dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t2max, op=dist.ReduceOp.MAX)
            d2 = float(t2max.item())
            also[name] = {"metric": w2.metric, "value": w2.units_per_step * ks * world / d2, "unit": w2.unit, "steps": ks,
                          "ms_per_step": d2 / ks * 1e3, "config": w2.config, "roofline": roofline(w2, t2, ks, alu_peak, d2)}
            if ctx_excl is not None:  # the un-stretched figure for this workload too (one lane / one slice on the second context)
                ex = exclusive_pass(w2, bbp, torch, device, dev_index, alu_peak, ctx2=ctx_excl)
                if ex:
                    also[name]["roofline"]["exclusive"] = ex
            if rank == 0 and world == 1 and not args.no_cpu_baseline and name == "verify":
                also[name]["cpu_baseline"] = dict(w2.cpu_baseline(), cpu_model=cpu_model())
            del w2

    if rank == 0:
        units = wl.units_per_step * args.steps * world
        out = {
            "metric": wl.metric, "value": units / dt, "unit": wl.unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": wl.data, "config": wl.config,
        }
        if world > 1:
            out["per_rank"] = {"value": [wl.units_per_step * args.steps / t for t in per_rank_s], "ms_per_step": [t / args.steps * 1e3 for t in per_rank_s],
                               "note": "each rank's own units/s over its own clock between the barriers; `value` = all units / the slowest rank's time"}
        if not STUB:
            out["roofline"] = roofline(wl, timings, args.steps, alu_peak, dt)
            if alu_peak_fresh:
                out["roofline"]["alu"]["peak_fresh_row"] = alu_peak_fresh
            if exclusive:
                if alu_peak_fresh and exclusive.get("alu_frac"):
                    exclusive["alu_frac_fresh_row"] = exclusive["alu_frac"] * alu_peak / alu_peak_fresh
                out["roofline"]["exclusive"] = exclusive
        out.update(wl.extra_report(timings))
        if gather_info:
            out["gather"] = gather_info
        if also:
            out["also"] = also
        if world == 1 and not args.no_cpu_baseline and not STUB:
            out["cpu_baseline"] = dict(wl.cpu_baseline(), cpu_model=cpu_model())
        print(json.dumps(out), flush=True)
    if ctx_excl is not None:
        torch.cuda.synchronize()
        ctx_excl.close()
    if ctx is not None:
        # torch objects that wrap the context's streams (ExternalStream, events recorded on them) must go before bbp_free destroys
        # the streams: an event destructor touching a destroyed stream at interpreter exit is a segfault
        import gc
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())
        del wl, engine_stream
        gc.collect()
        ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
