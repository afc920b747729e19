#!/usr/bin/env python3
"""Experiment builds: python tools/build_variant.py NAME [-DKNOB=value ...] -> dusk_blindbidproof_amd/libbbp_hip.NAME.so
(same sources, extra hipcc flags; load it with BBP_LIB_VARIANT=NAME).  The product library is untouched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dusk_blindbidproof_amd")
name, flags = sys.argv[1], sys.argv[2:]
objdir = os.path.join(PKG, "build", "variant_" + name)
os.makedirs(objdir, exist_ok=True)
srcs = sorted(f for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith((".hip", ".cpp")))
procs, objs = [], []
for s in srcs:
    o = os.path.join(objdir, s + ".o")
    objs.append(o)
    procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w"] + flags + ["-c", os.path.join(PKG, "csrc", s), "-o", o]))
for p in procs:
    if p.wait() != 0:
        raise SystemExit("compile failed")
out = os.path.join(PKG, "libbbp_hip.%s.so" % name)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
