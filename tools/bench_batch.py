#!/usr/bin/env python3
"""Per-batch time against the number of frames in the batch (UHD 12-bit Nat, buffers in HBM, back-to-back batches on one
stream): us per frame / GPix/s, both encodings."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

import _libs as L
import motioncam_decoder_amd as M

w, h = 3840, 2160
dev = torch.device("cuda:0")
ctx = M.Context(0)
imgs = [L.synth_image(w, h, 12, 1, 12.0, 3000 + i) for i in range(4)]
for typ, enc in ((7, L.encode7), (6, L.encode6)):
    bufs = [enc(im) for im in imgs]
    nmax = 256
    tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(nmax)]
    tout = torch.zeros(nmax * w * h * 2, dtype=torch.uint8, device=dev)
    row = []
    for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        fr = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, typ, tout.data_ptr() + i * w * h * 2, w * h)
                                    for i in range(n)])
        ctx.decode_batch(fr)
        reps = max(5, 400 // n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.decode_batch(fr, want_status=False)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        row.append("%d: %.1f / %.0f" % (n, t / n * 1e6, n * w * h / t / 1e9))
    print("type %d | " % typ + " | ".join(row), flush=True)
