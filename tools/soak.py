#!/usr/bin/env python3
"""Soak: batches of varying shapes, both encodings, device and host memory, synchronous, asynchronous and
ticketed, post stage on and off, for SOAK_SECONDS (default 60); reports host RSS and free HBM before and after
(the context's arenas only ever grow to the largest batch seen, then stay)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import psutil
import torch

import _libs as L
import motioncam_decoder_amd as M


def main():
    secs = float(os.environ.get("SOAK_SECONDS", "60"))
    dev = torch.device("cuda:0")
    ctx = M.Context(0)
    rng = np.random.default_rng(7)
    pool = []
    for (w, h) in ((640, 480), (1920, 1080), (4032, 3024), (1000, 37), (256, 16)):
        img = L.natural_image_np(w, h, 12, 12.0, w)
        for typ, enc in ((7, L.encode7), (6, L.encode6)):
            buf = enc(img)
            pool.append((typ, w, h, buf, torch.from_numpy(buf).to(dev), img))
    proc = psutil.Process()
    big = torch.empty(40 * 4032 * 3024 * 2, dtype=torch.uint8, device=dev)  # outputs are slices of this: no allocator traffic

    def snapshot():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return proc.memory_info().rss / 1e6, free / 1e6

    def one_round():
        n = int(rng.integers(1, 40))
        picks = [pool[int(rng.integers(0, len(pool)))] for _ in range(n)]
        mode = int(rng.integers(0, 4))
        pack12 = bool(rng.integers(0, 2)) and mode != 1
        if pack12:
            ctx.set_post(black=[64, 64, 64, 64], pack12=True)
        try:
            if mode in (0, 1):  # device memory, with statuses / fire-and-forget
                outs, o0 = [], 0
                for (_, w, h, _, _, _) in picks:
                    outs.append(big[o0:o0 + w * h * 2])
                    o0 += (w * h * 2 + 255) // 256 * 256
                fr = M.Context.make_frames([(t.data_ptr(), t.numel(), w, h, typ, o.data_ptr(), w * h)
                                            for (typ, w, h, _, t, _), o in zip(picks, outs)])
                if mode == 0:
                    wr, st = ctx.decode_batch(fr)
                    assert all(s == 0 for s in st)
                else:
                    ctx.decode_batch(fr, want_status=False)
                    assert all(s == 0 for s in ctx.synchronize(n))
            else:               # host memory, synchronous / ticketed
                outs = [np.empty(w * h * 2, np.uint8) for (_, w, h, _, _, _) in picks]
                fr = M.Context.make_frames([(b.ctypes.data, b.size, w, h, typ, o.ctypes.data, w * h)
                                            for (typ, w, h, b, _, _), o in zip(picks, outs)])
                if mode == 2:
                    wr, st = ctx.decode_batch(fr, mem=M.MEM_HOST)
                else:
                    wr, st = ctx.wait(ctx.decode_batch_async(fr))
                assert all(s == 0 for s in st)
                for (typ, w, h, _, _, img), o in zip(picks, outs):  # every frame: the copies of neighbours are merged
                    if pack12:
                        rb = L.post_row_bytes(w, True)
                        assert np.array_equal(o[: h * rb].reshape(h, rb), L.oracle_post(img, [64, 64, 64, 64], True))
                    else:
                        assert np.array_equal(o.view(np.uint16).reshape(h, w), img)
        finally:
            if pack12:
                ctx.set_post()

    for _ in range(200):  # let every arena reach its largest size first
        one_round()
    rss0, free0 = snapshot()
    t0 = time.time()
    rounds = 0
    mark = t0
    while time.time() - t0 < secs:
        one_round()
        rounds += 1
        if time.time() - mark > 20:
            mark = time.time()
            print("  t=%3.0f s rounds %d RSS %.0f MB free HBM %.0f MB" % ((mark - t0,) + (rounds,) + snapshot()), flush=True)
    rss1, free1 = snapshot()
    print("soak ok: %d rounds in %.0f s; host RSS %.0f -> %.0f MB, free HBM %.0f -> %.0f MB" % (rounds, time.time() - t0, rss0, rss1, free0, free1))


if __name__ == "__main__":
    main()
