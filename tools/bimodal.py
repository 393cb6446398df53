#!/usr/bin/env python3
"""Is the spread of k7_tiles between runs a property of the buffers' addresses or of the box's state?
One process, several allocations of the same workload at different addresses, timed in turn; a 1 GB fill between
them as the box's own yardstick."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _libs as L
import motioncam_decoder_amd as M

w, h, n = 3840, 2160, 240
dev = torch.device("cuda:0")
imgs = [L.synth_image(w, h, 12, 1, 12.0, 3000 + i) for i in range(8)]
bufs = [L.encode7(im) for im in imgs]
ctx = M.Context(0)
ctx.profile(True)
fillbuf = torch.empty(1 << 30, dtype=torch.int32, device=dev)


def fill_rate():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fillbuf.fill_(3)
    e1.record()
    torch.cuda.synchronize()
    return 5 * fillbuf.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12


sets = []
pads = []
for k in range(int(os.environ.get("SETS", "4"))):
    pads.append(torch.empty((k * 37 + 1) << 20, dtype=torch.uint8, device=dev))  # shift the next allocations
    tin = [torch.from_numpy(bufs[i % 8]).to(dev) for i in range(n)]
    tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
    frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 7, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
    sets.append((tin, tout, frames))
    print("set", k, "out at", hex(tout.data_ptr()), "in0 at", hex(tin[0].data_ptr()))
for rnd in range(4):
    for k, (tin, tout, frames) in enumerate(sets):
        ctx.decode_batch(frames, want_status=False)
        torch.cuda.synchronize()
        for kk in M.KERNELS:
            ctx.kernel_ms(kk, reset=True)
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.decode_batch(frames, want_status=False)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 20
        print("round %d set %d: step %.4f ms  k7_tiles %.4f  k7_side %.4f   fill %.2f TB/s" % (
            rnd, k, el * 1e3, ctx.kernel_ms("k7_tiles")[0] / 20, ctx.kernel_ms("k7_side")[0] / 20, fill_rate()), flush=True)
