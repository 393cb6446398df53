#!/usr/bin/env python3
"""Host-memory batches in a row on tickets, by frames per batch and tickets in flight (pinned buffers, the frames of a batch
neighbours in one allocation as in the facade's staging slots): where does a stream of small batches lose against one large one?"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import _libs as L
import motioncam_decoder_amd as M

w, h, total = 3840, 2160, int(os.environ.get("TOTAL", "240"))
if os.environ.get("TORCH_FIRST"):  # a torch operation on the GPU BEFORE the context exists (the hardware-queue lottery: tools/pcie_order.py)
    import torch
    torch.ones(4, device="cuda:0").sum().item()
lib = M.load()
# POOL=1: through the pool of one member (what the facade calls); SEPOUT=1: every output buffer an allocation of its own
ctx = M.Pool([0]) if os.environ.get("POOL") else M.Context(0)
SEPOUT = bool(os.environ.get("SEPOUT"))
pairs = [(im, L.encode7(im)) for im in (L.synth_image(w, h, 12, 1, 12.0, 3000 + i) for i in range(4))]
up = lambda v: (v + 255) // 256 * 256
isz = max(up(p[1].size) for p in pairs)
osz = up(w * h * 2)


def make_sets(per, ring):
    sets = []
    for r in range(ring):
        pin = lib.mcraw_host_alloc(isz * per)
        pouts = [lib.mcraw_host_alloc(osz) for i in range(per)] if SEPOUT else [lib.mcraw_host_alloc(osz * per)]
        descs = []
        for i in range(per):
            buf = pairs[(r + i) % 4][1]
            C.memmove(pin + i * isz, buf.ctypes.data, buf.size)
            descs.append((pin + i * isz, buf.size, w, h, 7, pouts[i] if SEPOUT else pouts[0] + i * osz, w * h))
        sets.append((M.Context.make_frames(descs), pin, pouts))
    return sets


def free_sets(sets):
    for _, pin, pouts in sets:
        lib.mcraw_host_free(pin)
        for p in pouts:
            lib.mcraw_host_free(p)


for per in [int(x) for x in os.environ.get("PERS", "3 7 16 40").split()]:
    row = {"frames_per_batch": per, "pool": bool(os.environ.get("POOL")), "separate_outputs": SEPOUT}
    for depth in [int(x) for x in os.environ.get("DEPTHS", "1 2 3 4").split()]:
        sets = make_sets(per, depth + 1)
        nb = max(4, total // per)
        for rep in range(2):  # (the first pass warms the slots)
            q = []
            t0 = time.perf_counter()
            for b in range(nb):
                if os.environ.get("SYNC") and depth == 1:  # the synchronous entry point (a large batch is dealt out inside)
                    wr, st = ctx.decode_batch(sets[b % (depth + 1)][0]) if os.environ.get("POOL") else ctx.decode_batch(sets[b % (depth + 1)][0], mem=M.MEM_HOST)
                    assert all(s == 0 for s in st)
                    continue
                q.append(ctx.decode_batch_async(sets[b % (depth + 1)][0]))
                if len(q) >= depth:
                    wr, st = ctx.wait(q.pop(0))
                    assert all(s == 0 for s in st)
            while q:
                ctx.wait(q.pop(0))
            dt = time.perf_counter() - t0
        got = np.ctypeslib.as_array(C.cast(sets[0][2][0], C.POINTER(C.c_uint16)), shape=(h, w))
        assert np.array_equal(got, pairs[0][0])
        row["depth%d_fps" % depth] = round(per * nb / dt, 1)
        free_sets(sets)
    print(json.dumps(row), flush=True)
