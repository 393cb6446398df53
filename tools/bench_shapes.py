#!/usr/bin/env python3
"""Throughput over image content that real footage has and the synthetic bench does not: clipped
regions, letterboxing, low light, stripes, dead pixels, flat frames -- both encodings, UHD.  A data shape
that takes a slow path shows up here (flat legacy frames did: 0.78 ms per 32 frames before the bulk
listing of 2-byte records in the legacy unpack, 0.25 ms after)."""
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


def shapes(w, h):
    rng = np.random.default_rng(1)
    nat = lambda seed: L.synth_image(w, h, 12, 1, 12.0, seed).copy()
    out = {"nat12": nat(5), "flat": np.full((h, w), 517, np.uint16),
           "lowlight": (64 + rng.integers(0, 24, (h, w))).astype(np.uint16),
           "checker": ((np.indices((h, w)).sum(0) % 2) * 4095).astype(np.uint16),
           "u14": rng.integers(0, 16384, (h, w), dtype=np.uint16)}
    x = rng.integers(0, 4096, (h, w), dtype=np.uint16); x[: h // 2] = 4095; out["halfclipped"] = x
    x = nat(6); x[:, : w // 4] = 0; x[:, -w // 4:] = 0; out["letterbox"] = x
    x = nat(7); x[::16] = 4095; out["stripes"] = x
    x = nat(8); x[rng.random((h, w)) < 0.0005] = 0; out["deadpix"] = x
    return out


def main():
    dev = torch.device("cuda:0")
    ctx = M.Context(0)
    ctx.profile(True)
    w, h = 3840, 2160
    res = {}
    for name, img in shapes(w, h).items():
        for typ, enc, n in ((7, L.encode7, 120), (6, L.encode6, 32)):
            if os.environ.get("SHAPES_TYPES") and str(typ) not in os.environ["SHAPES_TYPES"]:
                continue
            buf = enc(img)
            tin = torch.from_numpy(buf).to(dev)
            tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
            fr = M.Context.make_frames([(tin.data_ptr(), tin.numel(), w, h, typ, tout.data_ptr() + i * w * h * 2, w * h)
                                        for i in range(n)])
            wr, st = ctx.decode_batch(fr)
            ok = all(s == 0 for s in st) and np.array_equal(
                tout[(n - 1) * w * h * 2:].cpu().numpy().view(np.uint16).reshape(h, w), img)
            for _ in range(2):
                for k in M.KERNELS:
                    ctx.kernel_ms(k, reset=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    ctx.decode_batch(fr, want_status=False)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / 5
            res["%s_type%d" % (name, typ)] = {"frames": n, "bpp": round(8 * buf.size / (w * h), 2), "ms": round(t * 1e3, 3),
                                             "gpix_s": round(n * w * h / t / 1e9, 1), "bit_exact": bool(ok)}
            del tout
    print(json.dumps(res))


if __name__ == "__main__":
    main()
