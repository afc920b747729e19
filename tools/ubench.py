#!/usr/bin/env python3
"""Integer-ALU roofline numbers for DESIGN.md: python tools/ubench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dusk_blindbidproof_amd as bbp
ctx = bbp.Context(0)
names = ["v_mad_u64_u32", "fe_mul", "fe_sq", "ge_madd", "sc_montmul"]
for blocks in (2048, 8192):
    for k, n in enumerate(names):
        best = max(ctx.ubench(k, blocks, 3000 if k else 20000) for _ in range(3))
        print("blocks=%5d %-14s %.3e ops/s" % (blocks, n, best))
