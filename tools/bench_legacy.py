#!/usr/bin/env python3
"""Throughput of the legacy (type 6) path and of BASELINE config 4's mixed batch (not the bench line)."""
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


def main():
    dev = torch.device("cuda:0")
    ctx = M.Context(0)
    ctx.profile(True)
    res = {}
    for name, w, h, nb, dist, sig in (("legacy_4000x3000_nat12", 4000, 3000, 12, 1, 12.0), ("legacy_1920x1080_u10", 1920, 1080, 10, 0, 0.0)):
        n = int(os.environ.get("N", "32"))
        imgs = [L.synth_image(w, h, nb, dist, sig, 6000 + i) for i in range(4)]
        bufs = [L.encode6(im) for im in imgs]
        tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(4 if os.environ.get("DEDUP") else n)]  # DEDUP=1: four input buffers
        tin = [tin[i % len(tin)] for i in range(n)]                                                       # for all frames (reads hit the caches)
        tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
        descs = [(tin[i].data_ptr(), tin[i].numel(), w, h, 6, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)]
        frames = M.Context.make_frames(descs)
        written, status = ctx.decode_batch(frames)
        assert os.environ.get('MCRAW_NOCHECK') or all(s == 0 for s in status)
        got = tout[: w * h * 2].cpu().numpy().view(np.uint16).reshape(h, w)
        assert os.environ.get('MCRAW_NOCHECK') or np.array_equal(got, imgs[0])
        for k in M.KERNELS:
            ctx.kernel_ms(k, reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            ctx.decode_batch(frames, want_status=False)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        kms = {k: round(ctx.kernel_ms(k, reset=True)[0] / reps, 4) for k in ("k6_decode",)}
        byts = sum(b.size for b in bufs) * n // 4 + n * w * h * 2
        res[name] = {"ms_per_batch": round(t * 1e3, 3), "mpix_s": round(n * w * h / t / 1e6, 1),
                     "gbs_in_plus_out": round(byts / t / 1e9, 1), "bpp": round(8 * bufs[0].size / (w * h), 2), "kernels_ms": kms}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
