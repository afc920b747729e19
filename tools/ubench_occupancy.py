#!/usr/bin/env python3
"""Register-resident ge_madd / fe_mul rate against waves per SIMD (LDS ballast throttles occupancy; 256-lane blocks = one wave
per SIMD of a CU per block): BBP_UBENCH_LDS is read per call.  python tools/ubench_occupancy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dusk_blindbidproof_amd as bbp
ctx = bbp.Context(0)
for waves, lds in ((1, 160 * 1024), (2, 80 * 1024), (3, 53 * 1024), (4, 40 * 1024), (8, 20 * 1024)):
    os.environ["BBP_UBENCH_LDS"] = str(lds)
    for kind, name in ((3, "ge_madd"), (5, "ge_madd/fresh operand"), (1, "fe_mul")):
        r = max(ctx.ubench(kind, 256 * waves * 4, 4000) for _ in range(2))
        print("waves/SIMD %d  %-22s %.3e /s" % (waves, name, r), flush=True)
