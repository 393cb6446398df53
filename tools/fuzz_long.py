#!/usr/bin/env python3
"""Extended mutation fuzzing of the HIP path against the oracle (same rule as tests/test_gpu_fuzz.py:
whatever the oracle makes of a mutant, the GPU must agree).  FUZZ_SECONDS (default 120), FUZZ_SEED;
FUZZ_POST=1 runs every round with the fused post stage on (random black levels, 12-bit strips on or off)
and compares with the oracle's post stage applied to the oracle's decode."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import _libs as L
import motioncam_decoder_amd as M
from _gpu import decode_batch_device
from test_gpu_fuzz import _mutants


def main():
    secs = float(os.environ.get("FUZZ_SECONDS", "120"))
    seed = int(os.environ.get("FUZZ_SEED", str(int(time.time()))))
    rng = np.random.default_rng(seed)
    ctx = M.Context(0)
    t0 = time.time()
    rounds = frames = decoded = 0
    use_post = bool(os.environ.get("FUZZ_POST"))
    while time.time() - t0 < secs:
        black, pack12 = None, False
        if use_post:
            black = [int(x) for x in rng.integers(0, 5000, 4)] if rng.random() < 0.7 else None
            pack12 = bool(rng.random() < 0.6) or black is None
            ctx.set_post(black=black, pack12=pack12)
        typ = 7 if rng.random() < 0.6 else 6
        w = int(rng.choice([64, 77, 200, 256, 640, 1000, 1920, 4032]))
        h = int(rng.choice([4, 12, 30, 64, 270, 1080])) if w < 3000 else int(rng.choice([8, 64, 256]))
        kind = rng.random()
        if kind < 0.4:
            img = L.natural_image_np(w, h, int(rng.choice([10, 12, 14])), float(rng.choice([2.0, 12.0, 40.0])), int(rng.integers(1 << 30)))
        elif kind < 0.7:
            img = rng.integers(0, 1 << int(rng.integers(1, 17)), size=(h, w), dtype=np.uint16)
        else:  # flat bands between noise: dense and sparse chunks side by side
            img = rng.integers(0, 4096, size=(h, w), dtype=np.uint16)
            band = max(1, h // int(rng.integers(2, 6)))
            img[:band] = int(rng.integers(0, 4096))
        enc, dec = (L.encode7, L.oracle_decode7) if typ == 7 else (L.encode6, L.oracle_decode6)
        buf = enc(img)
        hot = []
        if typ == 7:
            bits_off = int(np.frombuffer(buf[8:12].tobytes(), np.uint32)[0])
            hot = [(0, 16), (bits_off, buf.size)]
        bufs = [buf] + _mutants(buf, rng, 24, hot) + [buf]
        written, status, outs = decode_batch_device(ctx, [(typ, w, h, b) for b in bufs], fill=0)
        for i, b in enumerate(bufs):
            ret, want = dec(b, w, h)
            if ret == 0:
                assert status[i] != 0 and written[i] == 0, (seed, rounds, i, typ, w, h, status[i], written[i])
            else:
                assert status[i] == 0 and written[i] == ret, (seed, rounds, i, typ, w, h, status[i], written[i], ret)
                rows = ret // w
                if use_post:
                    rb = L.post_row_bytes(w, pack12)
                    got = outs[i].reshape(-1).view(np.uint8)[: rows * rb].reshape(rows, rb)
                    assert np.array_equal(got, L.oracle_post(want[:rows], black, pack12)), (seed, rounds, i, typ, w, h, black, pack12)
                else:
                    assert np.array_equal(outs[i][:rows], want[:rows]), (seed, rounds, i, typ, w, h)
                decoded += 1
        frames += len(bufs)
        rounds += 1
    print("fuzz ok: seed %d, %d rounds, %d frames (%d decoded, %d rejected) in %.0f s" % (seed, rounds, frames, decoded, frames - decoded, time.time() - t0))


if __name__ == "__main__":
    main()
