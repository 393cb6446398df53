#!/usr/bin/env python3
"""The strip kernels (k7_tiles with the fused post stage) of two builds in ONE process on one set of buffers:
   gpurun -- 'python3 tools/ab_post.py cur prev'   (lib/libmcraw_hip_<name>.so; cur = lib/libmcraw_hip.so)"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch
import bench
import motioncam_decoder_amd as M

names = sys.argv[1:] or ["cur", "prev"]
args = types.SimpleNamespace(width=3840, height=2160, frames=240, distinct=48, config=3, nbits=int(os.environ.get("NBITS", "12")), sigma=12.0, streams=1)
dev = torch.device("cuda", 0)
L = bench.synth_lib()
wl = bench.Workload(torch, M, L, dev, args, "nat", list(range(args.frames)))
ctxs = []
for n in names:
    if n != "cur":
        os.environ["MCRAW_LIB_PATH"] = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "libmcraw_hip_%s.so" % n)
    M._lib = None
    ctxs.append((n, M.Context(0)))
    os.environ.pop("MCRAW_LIB_PATH", None)
for bits in (12, 10, 14):
    black = [256] * 4 if bits != 10 else [64] * 4
    res = {}
    for n, ctx in ctxs:
        ctx.set_post(black=black, bits=bits)
        ctx.profile(only=("k7_tiles",), every=4)
        for _ in range(24):
            ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, want_status=False)
        st = ctx.synchronize(wl.frames)
        assert all(s == 0 for s in st)
        rb = L.post_row_bytes(wl.w, bits=bits)
        got = wl.t_out[:wl.h * rb].cpu().numpy().reshape(wl.h, rb)
        assert os.environ.get("MCRAW_NOCHECK") or np.array_equal(got, L.oracle_post(wl.pairs[0][0], black, bits=bits)), (n, bits)
        ctx.kernel_ms("k7_tiles", reset=True)
    for r in range(6):
        for n, ctx in (ctxs if r % 2 == 0 else ctxs[::-1]):
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, want_status=False)
            ctx.synchronize()
            res.setdefault(n, []).append((time.perf_counter() - t0) * 50.0)
    for n, ctx in ctxs:
        ms, k = ctx.kernel_ms("k7_tiles", reset=True)
        rb = L.post_row_bytes(wl.w, bits=bits)
        byts = wl.in_bytes + wl.frames * wl.h * rb
        print("%d-bit strips %-6s step %.4f ms  k7_tiles %.4f ms  frac %.4f  xcd %s" % (bits, n, sorted(res[n])[len(res[n]) // 2], ms / max(k, 1), byts / (ms / max(k, 1) * 1e-3) / 8e12, ctx.xcd_runs()), flush=True)
        ctx.set_post()
for n, ctx in ctxs:
    ctx.close()
