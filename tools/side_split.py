#!/usr/bin/env python3
"""k7_side (and the whole type-7 step) with the side streams in 1, 2 and 4 parts (MCRAW_SIDE_SPLIT), several batch shapes."""
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
ctx.profile(True)
shapes = [("8K nat x120", 7680, 4320, 12, 1, 12.0, 120), ("12MP u14 x16", 4032, 3024, 14, 0, 0.0, 16), ("12MP nat x16", 4032, 3024, 12, 1, 12.0, 16),
          ("UHD nat x1", 3840, 2160, 12, 1, 12.0, 1), ("UHD nat x32", 3840, 2160, 12, 1, 12.0, 32), ("UHD nat x120", 3840, 2160, 12, 1, 12.0, 120),
          ("UHD u12 x120", 3840, 2160, 12, 0, 0.0, 120)]
if os.environ.get("SHAPE_EXTRA"):  # "name,w,h,bits,dist,sigma,frames"
    e = os.environ["SHAPE_EXTRA"].split(",")
    shapes.append((e[0], int(e[1]), int(e[2]), int(e[3]), int(e[4]), float(e[5]), int(e[6])))
only = os.environ.get("SHAPES")
for name, w, h, nb, dist, sig, n in shapes:
    if only and not any(o in name for o in only.split(",")):
        continue
    imgs = [L.synth_image(w, h, nb, dist, sig, 900 + i) for i in range(2)]
    bufs = [L.encode7(im) for im in imgs]
    tin = [torch.from_numpy(b).to(dev) for b in bufs]
    tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
    fr = M.Context.make_frames([(tin[i % 2].data_ptr(), tin[i % 2].numel(), w, h, 7, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
    line = {}
    for sp in os.environ.get("SPLITS", "auto;1;2;4").split(";"):
        if sp == "auto":
            os.environ.pop("MCRAW_SIDE_SPLIT", None)
        else:
            os.environ["MCRAW_SIDE_SPLIT"] = sp
        wr, st = ctx.decode_batch(fr)
        ok = all(s == 0 for s in st) and np.array_equal(tout[(n - 1) * w * h * 2:].cpu().numpy().view(np.uint16).reshape(h, w), imgs[(n - 1) % 2])
        for k in M.KERNELS:
            ctx.kernel_ms(k, reset=True)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.decode_batch(fr, want_status=False)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        line[sp] = {"side_us": round(ctx.kernel_ms("k7_side", reset=True)[0] / reps * 1e3, 1), "tiles_us": round(ctx.kernel_ms("k7_tiles", reset=True)[0] / reps * 1e3, 1),
                    "step_us": round(t * 1e6, 1), "ok": bool(ok)}
    print(name, json.dumps(line))
