#!/usr/bin/env python3
"""End-to-end (host buffers in and out) rate of mcraw_decode_batch in MCRAW_MEM_HOST mode."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import numpy as np
import torch
import _libs as L
import motioncam_decoder_amd as M

w, h, n = 3840, 2160, int(os.environ.get("NFRAMES", "96"))
lib = M.load(); ctx = M.Context(0)
pairs = []
for i in range(8):
    img = L.synth_image(w, h, 12, 1, 12.0, 3000 + i); pairs.append((img, L.encode7(img)))
ins, outs, descs = [], [], []
for i in range(n):
    buf = pairs[i % 8][1]
    pi = lib.mcraw_host_alloc(buf.size); po = lib.mcraw_host_alloc(w * h * 2)
    C.memmove(pi, buf.ctypes.data, buf.size); ins.append(pi); outs.append(po)
    descs.append((pi, buf.size, w, h, 7, po, w * h))
frames = M.Context.make_frames(descs)
ctx.decode_batch(frames, mem=M.MEM_HOST)
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    written, status = ctx.decode_batch(frames, mem=M.MEM_HOST)
t = (time.perf_counter() - t0) / reps
got = np.ctypeslib.as_array(C.cast(outs[n - 1], C.POINTER(C.c_uint16)), shape=(h, w))
print(json.dumps({"sub_mb": os.environ.get("MCRAW_SUB_MB", "96"), "frames": n, "fps": round(n / t, 1), "d2h_GBs": round(n * w * h * 2 / t / 1e9, 1),
                  "h2d_GBs": round(sum(pairs[i % 8][1].size for i in range(n)) / t / 1e9, 1), "ok": bool(all(s == 0 for s in status) and np.array_equal(got, pairs[(n - 1) % 8][0]))}))
