#!/usr/bin/env python3
"""BASELINE config 4 (bench.py's mixed64 leg) alone, with the time of every kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

import bench
import motioncam_decoder_amd as M

L = bench.synth_lib()
dev = torch.device("cuda:0")
ctx = M.Context(0)
ctx.profile(True)
r = bench.mixed64_leg(torch, ctx, M, L, dev, reps=10)
km = {}
for k in M.KERNELS:
    try:
        ms, n = ctx.kernel_ms(k, reset=True)
        km[k] = round(ms / max(n, 1), 4)
    except Exception:
        pass
r["kernel_ms_per_launch"] = km
print(json.dumps(r))
