#!/usr/bin/env python3
"""Event timeline of one k7_side workgroup, from a library built with -DMCRAW_DIAG (tools/side_prof.sh):
shader cycles since the workgroup's start at every stamp of wave 0 (the walker) and wave 1 (a decoder).
   N=240 DIST=nat|u python3 tools/side_prof.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import _libs as L
import motioncam_decoder_amd as M

w, h, n = int(os.environ.get("W", "3840")), int(os.environ.get("H", "2160")), int(os.environ.get("N", "240"))
dist = 1 if os.environ.get("DIST", "nat") == "nat" else 0
dev = torch.device("cuda:0")
nbits = int(os.environ.get("NB", "12"))
imgs = [L.synth_image(w, h, nbits, dist, float(os.environ.get("SIGMA", "12")), 3000 + i) for i in range(4)]
bufs = [L.encode7(im) for im in imgs]
tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(n)]
tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 7, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
ctx = M.Context(0)
lib = M.load()
for _ in range(3):
    ctx.decode_batch(frames)
prof = (C.c_ulonglong * 512)()
ctx.profile(True)
for k in M.KERNELS:
    ctx.kernel_ms(k, reset=True)
reps = 3
for _ in range(reps):
    ctx.decode_batch(frames, want_status=False)
torch.cuda.synchronize()
lib.mcraw_diag_side_prof(prof, 1)
names = {0: "top(after barrier)", 1: "moved/build done", 2: "walk done", 5: "scan done", 3: "decode done", 10: "plan read", 11: "header+count",
         20: "count run over", 21: "part in front has spoken", 12: "piece0 stored", 13: "strides0+barrier", 14: "first walk", 15: "end"}
for wv in (0, 1):
    row = prof[256 * wv: 256 * wv + 256]
    cnt = int(row[0])
    print("-- wave %d: %d events%s" % (wv, cnt, (", walk steps %d" % row[255]) if wv == 0 else ""))
    prev = 0
    for i in range(1, cnt + 1):
        ev, t = row[i] >> 48, row[i] & ((1 << 48) - 1)
        print("   %-20s %8d  (+%d)" % (names.get(ev, str(ev)), t, t - prev))
        prev = t
print("k7_side ms/launch", ctx.kernel_ms("k7_side")[0] / reps, "tiles", ctx.kernel_ms("k7_tiles")[0] / reps)
