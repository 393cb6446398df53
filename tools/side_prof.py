#!/usr/bin/env python3
"""Phase breakdown of one k7_side workgroup (refs stream of frame 0), from a library built with MCRAW_DIAG=1:
    MCRAW_DIAG=1 python -m motioncam_decoder_amd.build hip --force   (into a side copy: see tools/side_prof.sh)
Prints s_memtime ticks (100 MHz constant clock on gfx950? -> compare ratios) per phase, summed over the pieces."""
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

w, h, n = 3840, 2160, int(os.environ.get("N", "240"))
dist = 1 if os.environ.get("DIST", "nat") == "nat" else 0
dev = torch.device("cuda:0")
imgs = [L.synth_image(w, h, 12, dist, 12.0, 3000 + i) for i in range(4)]
bufs = [L.encode7(im) for im in imgs]
tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(n)]
tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 7, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
ctx = M.Context(0)
lib = M.load()
ctx.decode_batch(frames)
prof = (C.c_ulonglong * 32)()
lib.mcraw_diag_side_prof(prof, 1)
ctx.profile(True)
reps = 5
for _ in range(reps):
    ctx.decode_batch(frames, want_status=False)
torch.cuda.synchronize()
lib.mcraw_diag_side_prof(prof, 1)
names0 = ["wait units", "build..barrier", "walk(w0)", "decode(w1)", "scan wait", "scan", "walk steps", "walk calls",
          "walk: entry", "walk: first stride", "walk: loop", "top: s_st read", "top: build", "top: load issue", "-", "-"]
names = ["w0 " + x for x in names0] + ["w1 " + x for x in names0]
tot = sum(prof[:6])
tot1 = sum(prof[16:22])
for i, nm in enumerate(names):
    print("%-24s %10.0f ticks/launch  %5.1f %%" % (nm, prof[i] / reps, 100.0 * prof[i] / max(tot if i < 16 else tot1, 1)))
print("total ticks/launch", tot / reps, " k7_side ms/launch", ctx.kernel_ms("k7_side")[0] / reps, "tiles", ctx.kernel_ms("k7_tiles")[0] / reps)
