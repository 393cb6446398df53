#!/usr/bin/env python3
"""Soak of the round-4 paths that hand work between workgroups: k7_side's parts (speculative count, hand-off, time-out fallback)
and k6_decode's look-back, under changing residency -- resident batches of 1..40 large frames (12 MP and 8K; natural, uniform
noise, flat, letterboxed; both codecs mixed), random MCRAW_SIDE_SPLIT per context, every output compared on the GPU with the
image it was encoded from.  SOAK_SECONDS (default 120)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import _libs as L
import motioncam_decoder_amd as M


def images():
    out = []
    for (w, h) in ((4032, 3024), (7680, 4320), (4000, 3000)):
        nat = L.synth_image(w, h, 12, 1, 12.0, w)
        out.append(nat)
        rng = np.random.default_rng(h)
        out.append(rng.integers(0, 1 << 14, size=(h, w), dtype=np.uint16))  # every block raw: the refs stream's records change size
        flat = np.full((h, w), 700, np.uint16)
        flat[h // 3: h // 2, :] = nat[h // 3: h // 2, :]                        # flat with a band of detail: long runs, then short ones
        out.append(flat)
    return out


def main():
    secs = float(os.environ.get("SOAK_SECONDS", "120"))
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "11")))
    pool = []
    for img in images():
        h, w = img.shape
        want = torch.from_numpy(img.view(np.int16)).to(dev)
        for typ, enc in ((7, L.encode7), (6, L.encode6)):
            buf = enc(img)
            pool.append((typ, w, h, torch.from_numpy(buf).to(dev), want))
    big = torch.empty(24 * 7680 * 4320 * 2, dtype=torch.uint8, device=dev)
    t0 = time.time()
    rounds = frames = 0
    splits = [None, "2,2", "4,4", "4,1", "1,3", "3,2"]
    while time.time() - t0 < secs:
        sp = splits[int(rng.integers(0, len(splits)))]
        if sp:
            os.environ["MCRAW_SIDE_SPLIT"] = sp
        else:
            os.environ.pop("MCRAW_SIDE_SPLIT", None)
        ctx = M.Context(0)  # (the variable is read per batch; a fresh context also starts its measurements over)
        for _ in range(12):
            n = int(rng.integers(1, 41))
            picks, o0, outs = [], 0, []
            for _k in range(n):
                p = pool[int(rng.integers(0, len(pool)))]
                need = p[1] * p[2] * 2
                if o0 + need > big.numel():
                    break
                picks.append(p)
                outs.append(big[o0:o0 + need])
                o0 += (need + 255) // 256 * 256
            big[:o0].zero_()
            torch.cuda.synchronize()
            fr = M.Context.make_frames([(t.data_ptr(), t.numel(), w, h, typ, o.data_ptr(), w * h) for (typ, w, h, t, _), o in zip(picks, outs)])
            wr, st = ctx.decode_batch(fr)
            assert all(s == 0 for s in st), (sp, st)
            for (typ, w, h, _, want), o in zip(picks, outs):
                assert torch.equal(o.view(torch.int16).view(h, w), want), (sp, typ, w, h)
            rounds += 1
            frames += len(picks)
        ctx.close()
    print("soak_split ok: %d batches, %d frames in %.0f s" % (rounds, frames, time.time() - t0))


if __name__ == "__main__":
    main()
