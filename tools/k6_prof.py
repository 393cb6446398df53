#!/usr/bin/env python3
"""Phase breakdown of k6_decode summed over all workgroups, from a library built with -DMCRAW_DIAG (tools/k6_prof.sh)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import _libs as L
import motioncam_decoder_amd as M

w, h, n = 4000, 3000, int(os.environ.get("N", "32"))
dev = torch.device("cuda:0")
NB, DIST = int(os.environ.get("NB", "12")), int(os.environ.get("DIST", "1"))  # bits per sample; 1 = natural, 0 = uniform noise
imgs = [L.synth_image(w, h, NB, DIST, 12.0, 6000 + i) for i in range(4)]
bufs = [L.encode6(im) for im in imgs]
tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(n)]
tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 6, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
ctx = M.Context(0)
lib = M.load()
written, status = ctx.decode_batch(frames)
assert all(s == 0 for s in status)
NWG = 1 << 16
prof = np.zeros((NWG, 32), np.uint32)
pp = prof.ctypes.data_as(C.POINTER(C.c_uint32))
lib.mcraw_diag_k6_prof(pp, NWG, 1)
ctx.profile(True)
reps = 5
for _ in range(reps):
    ctx.decode_batch(frames, want_status=False)
torch.cuda.synchronize()
lib.mcraw_diag_k6_prof(pp, NWG, 1)   # stamps of the last launch
live = prof[:, 14] == 1
P = prof[live].astype(np.float64)
names = {0: "w0 ticket+load", 1: "w0 stage", 2: "w0 wait for wave 3 (sure entry)", 3: "w0 wait for wave 4", 4: "w0 (lists ready)", 5: "w0 unpack",
         8: "w4 ticket+load", 9: "w4 stage", 10: "w4 warm-up walk", 11: "w4 quarter walks + lane checks + wait for wave 3", 26: "w4 check from the sure entry",
         12: "w4 scan+look-back", 13: "w4 entries+lists", 25: "w4 walk rounds (count)"}
for i, nm in names.items():
    print("%-30s mean %8.0f  p50 %8.0f  p90 %8.0f ticks" % (nm, P[:, i].mean(), np.median(P[:, i]), np.percentile(P[:, i], 90)))
print("workgroups", live.sum(), "lifetime mean", P[:, 0:6].sum(axis=1).mean(), "spins/wg", P[:, 15].mean(), "max", P[:, 15].max())
Q = P[P[:, 7] > 0]
dt = (Q[:, 7] - Q[:, 6]) % 2**32          # s_memtime ticks of a workgroup's life
dr = (Q[:, 17] - Q[:, 16]) % 2**32        # the same in 100 MHz ticks
ok = dr > 0
clk = np.median(dt[ok] / dr[ok]) * 100.0
ms = ctx.kernel_ms("k6_decode")[0] / reps
life_us = np.mean(dr[ok]) / 100.0
print("in-kernel clock MHz (median of workgroups)", clk, " workgroup life us", life_us, " -> workgroups in flight", life_us * live.sum() / (ms * 1e3),
      "= per CU", life_us * live.sum() / (ms * 1e3) / 256)
print("ms/launch", ctx.kernel_ms("k6_decode")[0] / reps)

# residency: workgroups per CU over time, from the stamps of every wave's end and the hardware ids of wave 0
R = prof[live].astype(np.int64)
start = R[:, 16]
end = R[:, 18:23].max(axis=1)
t0 = start.min()
start = (start - t0) % 2**32
end = (end - t0) % 2**32
good = (end > start) & (end - start < 10**6)
cu = (R[:, 24] & 15) * 65536 + (R[:, 23] & 0xFF00)          # XCC, then SE / SH / CU bits of HW_ID
ids = np.unique(cu[good])
tot_busy = 0.0; peak = []; span = (end[good].max() - start[good].min())
for c in ids:
    m = good & (cu == c)
    ev = np.concatenate([np.stack([start[m], np.ones(m.sum(), np.int64)], 1), np.stack([end[m], -np.ones(m.sum(), np.int64)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    conc = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    tot_busy += float((conc[:-1] * dt).sum())
    peak.append(conc.max())
print("distinct CU ids", len(ids), " mean resident workgroups per CU %.2f" % (tot_busy / span / len(ids)), " peak per CU: min %d median %d max %d" % (min(peak), np.median(peak), max(peak)),
      " workgroup life incl. store drain us: mean %.2f" % ((end[good] - start[good]).mean() / 100.0))
hw = prof[live][:, 27:32]
print("wave slot ids seen (all waves):", np.bincount((hw & 15).ravel(), minlength=10), " SIMD ids:", np.bincount(((hw >> 4) & 3).ravel()))
if os.environ.get("K6_PROF_SAVE"):
    np.save(os.environ["K6_PROF_SAVE"], prof[live])
