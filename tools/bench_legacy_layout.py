#!/usr/bin/env python3
"""Does k6_decode's time depend on WHERE the frames' buffers lie?  The launch keeps all frames of a batch at the same segment index
at any moment, so their streams sit at the same offset from their buffers' starts: with buffers a power of two apart those
accesses could meet on memory channels.  LAYOUT=sep (one allocation per frame: torch's allocator), packed (one arena, frames back
to back), aligned (arena, 16 MiB apart), skew (arena, 16 MiB + i * 4352 bytes apart); OUT=packed|aligned|skew likewise."""
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


def place(layout, sizes, dev):
    n = len(sizes)
    if layout == "sep":
        ts = [torch.empty(s, dtype=torch.uint8, device=dev) for s in sizes]
        return ts, [t.data_ptr() for t in ts]
    big = max(sizes)
    if layout == "packed":
        offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])[:-1]])
    elif layout == "aligned":
        st = (big + (16 << 20) - 1) // (16 << 20) * (16 << 20)
        offs = [i * st for i in range(n)]
    else:
        st = (big + (16 << 20) - 1) // (16 << 20) * (16 << 20)
        offs = [i * (st + 4352 * 7) for i in range(n)]
    arena = torch.empty(int(offs[-1]) + big + 4096, dtype=torch.uint8, device=dev)
    base = (arena.data_ptr() + 255) // 256 * 256
    return [arena], [base + int(o) for o in offs]


def main():
    dev = torch.device("cuda:0")
    ctx = M.Context(0)
    ctx.profile(True)
    w, h, nb, n = 4000, 3000, 12, 32
    imgs = [L.synth_image(w, h, nb, 1, 12.0, 6000 + i) for i in range(4)]
    bufs = [L.encode6(im) for im in imgs]
    res = {}
    for lin in os.environ.get("LAYOUTS", "sep packed aligned skew").split():
        for lout in os.environ.get("OUTS", "packed aligned skew").split():
            keep_i, pin = place(lin, [bufs[i % 4].size for i in range(n)], dev)
            keep_o, pout = place(lout, [w * h * 2] * n, dev)
            for i in range(n):
                src = torch.from_numpy(bufs[i % 4]).to(dev)
                # copy into place through a view on the raw pointer
                dst = torch.empty(0, dtype=torch.uint8, device=dev)
                import ctypes
                torch.cuda.synchronize()
                M.load()  # (library loaded)
                hip = ctypes.CDLL("libamdhip64.so")
                hip.hipMemcpy(ctypes.c_void_p(pin[i]), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(src.numel()), 3)
            descs = [(pin[i], bufs[i % 4].size, w, h, 6, pout[i], w * h) for i in range(n)]
            frames = M.Context.make_frames(descs)
            written, status = ctx.decode_batch(frames)
            assert all(s == 0 for s in status)
            for k in M.KERNELS:
                ctx.kernel_ms(k, reset=True)
            torch.cuda.synchronize()
            reps = 12
            for _ in range(reps):
                ctx.decode_batch(frames, want_status=False)
            torch.cuda.synchronize()
            res["%s/%s" % (lin, lout)] = round(ctx.kernel_ms("k6_decode", reset=True)[0] / reps, 4)
            del keep_i, keep_o
    print(json.dumps(res))


if __name__ == "__main__":
    main()
