#!/usr/bin/env python3
"""GPU box: ONE parametrised A/B harness (replaces the per-knob tools/*_ab.sh scripts of rounds 1-3).

    python tools/ab.py --knob BBP_SERIAL_BLOCK --values 64 256 [--repeats 3] [--bench "--workload verify --batch 1024"]
                       [--steps 20 --warmup 4] [--exclusive] [--tag NAME] [--commit SHA] [--out gpurun_out/ab/NAME.jsonl]
    python tools/ab.py --knob variant --values - coop            # "-" = the product library, others = tools/build_variant.py builds
    python tools/ab.py --knob "BBP_A,BBP_B" --values 1,0 0,1      # several environment knobs moved together
    python tools/ab.py --harness uds --knob hwq --values 8 16 --bench "--connections 3072 --no-verify --ops 110592"   # through the UDS server
    python tools/ab.py --harness script --script tools/midsize.py --knob BBP_RNG_DPP --values 1 0 --bench "256"         # any other tool

Arms alternate inside every repeat (a b a b ...), so box drift and clock state hit both alike.  Every run is one child process
(fresh context); one JSON object per run goes to the jsonl with the box id, the commit given on the command line, the arm, and the
figures that were being compared by hand before: value, ms_per_step, roofline.alu, roofline.exclusive, per-kernel averages (bench.py),
or the tool's own JSON line (uds / script).  A summary table (median per arm) is printed at the end.  Nothing here touches oracle/
beyond bench.py's own sampled check (BBP_BENCH_NO_CHECK=1 is set only with --no-check, for deliberately wrong knock-out builds)."""
import argparse, json, os, socket, statistics, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def box_id():
    gpu = ""
    try:
        for d in sorted(os.listdir("/sys/class/drm")):
            p = "/sys/class/drm/%s/device/unique_id" % d
            if os.path.exists(p):
                gpu = open(p).read().strip()
                break
    except OSError:
        pass
    return socket.gethostname() + (":" + gpu if gpu else "")


def kernel_avg(d, name):
    e = (d.get("kernels_us") or {}).get(name)
    return round(e["total_us"] / e["launches"], 1) if e and e.get("launches") else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", required=True, help="environment variable name(s, comma separated), or 'variant' for BBP_LIB_VARIANT builds")
    ap.add_argument("--values", nargs="+", required=True)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--harness", default="bench", choices=["bench", "uds", "script"],
                    help="what one run is: bench.py (default); tools/uds_bench.py (the UDS server + load generator; knob 'hwq' = its --hwq); "
                         "or any python tool given with --script (its stdout lines are kept)")
    ap.add_argument("--script", default=None, help="--harness script: the tool, e.g. tools/midsize.py")
    ap.add_argument("--bench", default="", help="extra arguments of the harness (bench.py: workload, batch; uds_bench.py: --connections ...; script: its own)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--exclusive", action="store_true", help="keep bench.py's exclusive (one-slice) pass: roofline.exclusive per run")
    ap.add_argument("--also", action="store_true", help="keep the secondary workloads (also.verify ...)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--tag", default=None)
    ap.add_argument("--commit", default=os.environ.get("BBP_COMMIT", "unknown"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--timeout", type=int, default=300)
    a = ap.parse_args()
    tag = a.tag or a.knob.replace(",", "+")
    out = a.out or os.path.join(ROOT, "gpurun_out", "ab", tag + ".jsonl")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    knobs = ["BBP_LIB_VARIANT"] if a.knob == "variant" else a.knob.split(",")
    # native artefacts once, before the first run (every run then passes --no-build and would refuse a stale library)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if b.returncode != 0:
        sys.stderr.write(b.stdout.decode()[-2000:])
        return 1
    box = box_id()
    rows = []
    with open(out, "a") as f:
        for rep in range(a.repeats):
            for val in a.values:
                env = dict(os.environ)
                vals = val.split(",") if len(knobs) > 1 else [val]
                for k, v in zip(knobs, vals):
                    if v in ("-", "default"):
                        env.pop(k, None)
                    else:
                        env[k] = v
                if a.no_check:
                    env["BBP_BENCH_NO_CHECK"] = "1"
                if a.harness == "bench":
                    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", str(a.warmup), "--no-cpu-baseline", "--no-build"]
                    if not a.exclusive:
                        cmd.append("--no-exclusive")
                    if not a.also:
                        cmd.append("--no-also")
                    cmd += a.bench.split()
                elif a.harness == "uds":
                    cmd = [sys.executable, os.path.join(ROOT, "tools", "uds_bench.py")] + a.bench.split()
                    if a.knob == "hwq":  # the server process's GPU_MAX_HW_QUEUES is uds_bench.py's own flag
                        env.pop("hwq", None)
                        cmd += ["--hwq", val]
                else:
                    cmd = [sys.executable, os.path.join(ROOT, a.script)] + a.bench.split()
                t0 = time.time()
                try:
                    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=a.timeout)
                    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
                    d = json.loads(lines[-1]) if lines else None
                    err = None if d else (p.stderr.decode()[-400:] or "no JSON line")
                except subprocess.TimeoutExpired:
                    d, err = None, "timeout"
                rec = {"tag": tag, "knob": a.knob, "arm": val, "repeat": rep, "box": box, "commit": a.commit, "bench": a.bench, "steps": a.steps,
                       "wall_s": round(time.time() - t0, 1)}
                if d and a.harness != "bench":  # uds_bench.py / a script: keep its own line(s)
                    rec["result"] = d
                    rec["value"] = d.get("proofs_per_s") or d.get("value") or 0.0   # the load generator's whole-run rate
                    rec["unit"] = "proofs/s (through the socket)" if "proofs_per_s" in d else d.get("unit", "")
                    rec["ms_per_step"] = (d.get("prove_latency_ms") or {}).get("p50") or d.get("ms_per_step") or 0.0  # prove p50 for the UDS harness
                elif a.harness == "script" and not d:
                    rec["stdout_tail"] = p.stdout.decode()[-1500:]
                    rec.pop("error", None)
                    err = None
                    d = {}
                elif d:
                    rl = d.get("roofline") or {}
                    rec.update({"value": d["value"], "unit": d.get("unit"), "ms_per_step": d["ms_per_step"],
                                "alu_frac": (rl.get("alu") or {}).get("frac"), "whole_step_frac": (rl.get("alu") or {}).get("whole_step_frac"),
                                "avg_launch_us": rl.get("avg_launch_us"), "exclusive": rl.get("exclusive"), "alu_peak": (rl.get("alu") or {}).get("peak"),
                                "acc_us": kernel_avg(d, "k_msm_acc"), "sort_us": kernel_avg(d, "k_msm_sort"), "fold_us": kernel_avg(d, "k_msm_fold"),
                                "also": {k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("also") or {}).items()} or None})
                else:
                    rec["error"] = err
                    if err == "timeout":  # a GPU step that had to be killed: start no further GPU step in this call
                        f.write(json.dumps(rec) + "\n")
                        print("TIMEOUT in arm %s: stopping" % val, flush=True)
                        return 1
                f.write(json.dumps(rec) + "\n")
                f.flush()
                rows.append(rec)
                ex = rec.get("exclusive") or {}
                print("[%s=%s #%d] %s %s  ms/step %s  alu %s  excl.frac %s excl.alu %s excl.dom_ms %s" % (
                    a.knob, val, rep, "%.0f" % rec["value"] if "value" in rec else "ERR " + str(rec.get("error"))[:200], rec.get("unit", ""),
                    "%.2f" % rec["ms_per_step"] if "ms_per_step" in rec else "-", "%.3f" % rec["alu_frac"] if rec.get("alu_frac") else "-",
                    "%.4f" % ex["frac"] if ex.get("frac") else "-", "%.3f" % ex["alu_frac"] if ex.get("alu_frac") else "-",
                    "%.2f" % ex["dominant_ms_per_step"] if ex.get("dominant_ms_per_step") else "-"), flush=True)
    print("== medians (%s, %s, commit %s)" % (tag, box, a.commit))
    for val in a.values:
        rs = [r for r in rows if r["arm"] == val and "value" in r]
        if not rs:
            print("  %-24s no successful run" % val)
            continue
        med = lambda k: statistics.median([r[k] for r in rs if r.get(k) is not None]) if any(r.get(k) is not None for r in rs) else float("nan")
        exs = [r["exclusive"] for r in rs if r.get("exclusive")]
        exm = lambda k: statistics.median([e[k] for e in exs if e.get(k) is not None]) if exs else float("nan")
        print("  %-24s value %9.0f  ms/step %7.2f  alu %.3f  acc_us %7.1f | exclusive frac %.4f alu %.3f dom_ms %.2f" % (
            val, med("value"), med("ms_per_step"), med("alu_frac"), med("acc_us"), exm("frac"), exm("alu_frac"), exm("dominant_ms_per_step")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
