#!/usr/bin/env python3
"""GPU box: shader clock and board power while the prove workload runs back to back (is the heavy stage issue-bound at full clock
or power-limited?).  python tools/power_probe.py [seconds=6] [workload=prove|msm]
Reads the amdgpu hwmon / sysfs files if the box exposes them, else rocm-smi as a child process; the workload runs in this process."""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
name = sys.argv[2] if len(sys.argv) > 2 else "prove"
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
ctx = bbp.Context(0)
wl = make_workload(name, ctx, bbp, torch, device, 1024, 8, seed=1)
es = torch.cuda.ExternalStream(ctx.stream, device=device)
torch.cuda.set_stream(es)
stream = es.cuda_stream


def read_sysfs():
    out, allc = {}, []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for f in glob.glob(card + "/hwmon/hwmon*/power1_average") + glob.glob(card + "/hwmon/hwmon*/power1_input"):
            try:
                out["power_W"] = int(open(f).read()) / 1e6
            except Exception:
                pass
        for f in glob.glob(card + "/hwmon/hwmon*/freq1_input"):
            try:
                out["sclk_MHz"] = int(open(f).read()) / 1e6
            except Exception:
                pass
        try:
            cur = [l for l in open(card + "/pp_dpm_sclk").read().splitlines() if l.strip().endswith("*")]
            if cur:
                out["dpm_sclk"] = cur[0].strip()
        except Exception:
            pass
        if out:
            allc.append((os.path.basename(os.path.dirname(card)), dict(out)))
            out = {}
    return {"cards": allc} if allc else {}


def read_smi():
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in t.splitlines() if any(k in l for k in ("sclk", "Power", "GPU use", "mclk"))]
        return {"smi": " | ".join(keep)}
    except Exception as e:
        return {"smi_error": str(e)}


samples, stop = [], threading.Event()


def sampler():
    use_smi = not read_sysfs()
    while not stop.is_set():
        s = read_smi() if use_smi else read_sysfs()
        s["t"] = round(time.perf_counter() - t0, 2)
        samples.append(s)
        time.sleep(0.25 if not use_smi else 0.05)


for _ in range(3):
    wl.step(stream)
torch.cuda.synchronize()
print("idle:", read_sysfs() or read_smi(), flush=True)
try:
    print("this GPU:", torch.cuda.get_device_properties(0).name, getattr(torch.cuda.get_device_properties(0), "pci_bus_id", None), os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"))
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        print(card, os.path.realpath(card))
except Exception as e:
    print(e)
t0 = time.perf_counter()
th = threading.Thread(target=sampler)
th.start()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(4):
        wl.step(stream)
        n += 1
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop.set()
th.join()
print("%s: %d steps in %.2f s = %.2f ms per step" % (name, n, dt, dt / n * 1e3))
for s in samples:
    print(s)
