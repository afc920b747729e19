#!/usr/bin/env python3
"""Does the register-resident ge_madd rate hold when the kernel runs for seconds (clock / power behaviour)?
python tools/ubench_sustained.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dusk_blindbidproof_amd as bbp
ctx = bbp.Context(0)
for iters in (2000, 20000, 100000, 100000, 100000):
    t = time.time()
    r = ctx.ubench(3, 8192, iters)
    print("ge_madd iters=%6d  %.3e /s   (%.2f s wall)" % (iters, r, time.time() - t), flush=True)
for iters in (20000, 400000):
    r = ctx.ubench(0, 8192, iters)
    print("v_mad_u64_u32 iters=%6d  %.3e /s" % (iters, r), flush=True)
