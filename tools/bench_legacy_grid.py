#!/usr/bin/env python3
"""Legacy (type 6) kernel time by content: bit depth x distribution x size, 32 frames each."""
import json
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
res = {}
for (w, h) in ((1920, 1080), (4000, 3000)):
    for nb in (10, 12, 14):
        for dist in (0, 1):
            n = 32
            imgs = [L.synth_image(w, h, nb, dist, 12.0, 7000 + i) for i in range(2)]
            bufs = [L.encode6(im) for im in imgs]
            tin = [torch.from_numpy(bufs[i % 2]).to(dev) for i in range(n)]
            tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
            frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 6, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
            written, status = ctx.decode_batch(frames)
            assert all(s == 0 for s in status)
            assert np.array_equal(tout[: w * h * 2].cpu().numpy().view(np.uint16).reshape(h, w), imgs[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.decode_batch(frames, want_status=False)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 10
            res["%dx%d_%dbit_%s" % (w, h, nb, "U" if dist == 0 else "Nat")] = {"ms": round(t * 1e3, 3), "gpix_s": round(n * w * h / t / 1e9, 1),
                                                                              "bpp": round(8 * bufs[0].size / (w * h), 2)}
print(json.dumps(res))
