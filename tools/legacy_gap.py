#!/usr/bin/env python3
"""Wall time per legacy batch (32 x 4000x3000 12-bit, buffers in HBM) over many back-to-back batches, kernel timing off:
what a batch costs around its one kernel (table upload, launch, slot bookkeeping)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import _libs as L
import motioncam_decoder_amd as M

dev = torch.device("cuda:0")
ctx = M.Context(0)
w, h, n = 4000, 3000, 32
imgs = [L.synth_image(w, h, 12, 1, 12.0, 6000 + i) for i in range(4)]
bufs = [L.encode6(im) for im in imgs]
tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(n)]
tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 6, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
written, status = ctx.decode_batch(frames)
assert all(s == 0 for s in status)
for reps in (50, 200, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.decode_batch(frames, want_status=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("reps %d: %.4f ms per batch (host submitted them in %.4f ms each)" % (reps, (t2 - t0) / reps * 1e3, (t1 - t0) / reps * 1e3), flush=True)
