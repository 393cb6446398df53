#!/usr/bin/env python3
"""k6_decode of a -DMCRAW_DIAG library in ONE replay mode (env K6_MODE: 0 normal, 2 maps read back, 3 read back with the loads
issued early), for rocprofv3 --pmc (tools/k6_replay_pmc.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import _libs as L
import motioncam_decoder_amd as M

w, h, n = 4000, 3000, 32
mode = int(os.environ.get("K6_MODE", "0"))
dev = torch.device("cuda:0")
imgs = [L.synth_image(w, h, 12, 1, 12.0, 6000 + i) for i in range(4)]
bufs = [L.encode6(im) for im in imgs]
tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(n)]
tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 6, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
ctx = M.Context(0)
lib = M.load()
lib.mcraw_diag_k6_maps(1)
ctx.decode_batch(frames)
lib.mcraw_diag_k6_maps(mode)
for _ in range(12):
    ctx.decode_batch(frames, want_status=False)
torch.cuda.synchronize()
print("mode", mode, "done")
