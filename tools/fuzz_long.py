#!/usr/bin/env python3
"""Longer run of the randomised differential tests (tests/test_gpu_fuzz.py) over many seeds: python tools/fuzz_long.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
import dusk_blindbidproof_amd as bbp
from tests import oracle_c
import tests.test_gpu_fuzz as fz
oc = oracle_c.load(ge.build_oracle())
ctx = bbp.Context(0)
for seed in range(100, 140):
    fz.test_msm_fuzz.__wrapped__(ctx, bbp, oc, seed) if hasattr(fz.test_msm_fuzz, "__wrapped__") else fz.test_msm_fuzz(ctx, bbp, oc, seed)
print("msm fuzz: 40 seeds ok", flush=True)
for seed in range(200, 212):
    fz.test_prove_verify_fuzz(ctx, bbp, oc, seed)
print("prove/verify fuzz: 12 seeds ok", flush=True)
