#!/usr/bin/env python3
"""Small host-memory batches in a row (7 UHD frames each, pinned buffers): synchronous calls against
tickets with one batch queued behind the one being waited for (mcraw_decode_batch_async)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import _libs as L
import motioncam_decoder_amd as M

w, h, per, nb = 3840, 2160, int(os.environ.get("PER", "7")), int(os.environ.get("BATCHES", "24"))
lib = M.load()
ctx = M.Context(0)
pairs = [(im, L.encode7(im)) for im in (L.synth_image(w, h, 12, 1, 12.0, 3000 + i) for i in range(4))]
ring = 3  # buffer sets in rotation
sets = []
for r in range(ring):
    descs, outs = [], []
    for i in range(per):
        buf = pairs[(r + i) % 4][1]
        pi = lib.mcraw_host_alloc(buf.size)
        po = lib.mcraw_host_alloc(w * h * 2)
        C.memmove(pi, buf.ctypes.data, buf.size)
        descs.append((pi, buf.size, w, h, 7, po, w * h))
        outs.append(po)
    sets.append((M.Context.make_frames(descs), outs))


def check(r):
    got = np.ctypeslib.as_array(C.cast(sets[r][1][0], C.POINTER(C.c_uint16)), shape=(h, w))
    return np.array_equal(got, pairs[r % 4][0])


ctx.decode_batch(sets[0][0], mem=M.MEM_HOST)
t0 = time.perf_counter()
for b in range(nb):
    ctx.decode_batch(sets[b % ring][0], mem=M.MEM_HOST)
t_sync = time.perf_counter() - t0
ok = all(check(r) for r in range(ring))
t0 = time.perf_counter()
prev = None
for b in range(nb):
    t = ctx.decode_batch_async(sets[b % ring][0])
    if prev is not None:
        wr, st = ctx.wait(prev)
        ok = ok and all(s == 0 for s in st)
    prev = t
wr, st = ctx.wait(prev)
t_async = time.perf_counter() - t0
ok = ok and all(s == 0 for s in st) and all(check(r) for r in range(ring))
print(json.dumps({"frames_per_batch": per, "batches": nb, "sync_fps": round(per * nb / t_sync, 1),
                  "async_fps": round(per * nb / t_async, 1), "ok": bool(ok)}))
