#!/usr/bin/env python3
"""k7_side / k7_tiles time against the number of frames in the batch (UHD 12-bit Nat or U)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _libs as L
import motioncam_decoder_amd as M

w, h = int(os.environ.get("W", "3840")), int(os.environ.get("H", "2160"))
nbits = int(os.environ.get("NB", "12"))
dist = 1 if os.environ.get("DIST", "nat") == "nat" else 0
dev = torch.device("cuda:0")
imgs = [L.synth_image(w, h, nbits, dist, float(os.environ.get("SIGMA", "12")), 3000 + i) for i in range(4)]
bufs = [L.encode7(im) for im in imgs]
ctx = M.Context(0)
ctx.profile(True)
nmax = int(os.environ.get("NMAX", "480"))
tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(nmax)]
tout = torch.zeros(nmax * w * h * 2, dtype=torch.uint8, device=dev)
for n in [int(x) for x in os.environ.get("NS", "1,2,16,64,120,128,180,240,360,480").split(",")]:
    frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 7, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
    ctx.decode_batch(frames, want_status=False); torch.cuda.synchronize()
    for k in M.KERNELS:
        ctx.kernel_ms(k, reset=True)
    reps = 5
    for _ in range(reps):
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    print("frames %4d  k7_side %.4f ms  k7_tiles %.4f ms" % (n, ctx.kernel_ms("k7_side")[0] / reps, ctx.kernel_ms("k7_tiles")[0] / reps), flush=True)
