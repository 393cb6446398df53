#!/usr/bin/env python3
"""Host cost of one device-memory batch (gpurun -- 'python3 tools/submit_cost.py'): the time mcraw_decode_batch takes to
return when it only queues (plans + launches), and a synchronous call, for 240 / 32 / 1 UHD frames."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import motioncam_decoder_amd as M
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import _libs as L

dev = torch.device("cuda:0")
w, h = 3840, 2160
img = L.synth_image(w, h, 12, 1, 12.0, 5)
buf = L.encode7(img)
t_in = torch.from_numpy(buf).to(dev)
ctx = M.Context(0)
res = {}
for n in (240, 32, 1):
    out = torch.empty((n, h, w), dtype=torch.int16, device=dev)
    frames = M.Context.make_frames([(t_in.data_ptr(), t_in.numel(), w, h, 7, out[i].data_ptr(), w * h) for i in range(n)])
    for _ in range(12):
        ctx.decode_batch(frames)
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.decode_batch(frames, want_status=False)
        t1 = time.perf_counter()
        ctx.synchronize(n) if hasattr(ctx, "synchronize") else torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    sy = []
    for _ in range(20):
        t0 = time.perf_counter()
        ctx.decode_batch(frames)
        sy.append(time.perf_counter() - t0)
    res[n] = {"queue_only_us": round(1e6 * float(np.median([a for a, b in ts])), 1), "queue_then_wait_us": round(1e6 * float(np.median([b for a, b in ts])), 1),
              "synchronous_call_us": round(1e6 * float(np.median(sy)), 1)}
print(json.dumps(res))

# the same through a pool of one member (a thread of its own drives the context)
pool = M.Pool([0])
resp = {}
for n in (240, 32, 1):
    out = torch.empty((n, h, w), dtype=torch.int16, device=dev)
    frames = M.Context.make_frames([(t_in.data_ptr(), t_in.numel(), w, h, 7, out[i].data_ptr(), w * h) for i in range(n)])
    for _ in range(12):
        pool.decode_batch_device(frames)
    sy, qo = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        pool.decode_batch_device(frames)
        sy.append(time.perf_counter() - t0)
    for _ in range(20):
        t0 = time.perf_counter()
        pool.decode_batch_device(frames, want_status=False)
        t1 = time.perf_counter()
        pool.synchronize(n)
        qo.append((t1 - t0, time.perf_counter() - t0))
    resp[n] = {"synchronous_call_us": round(1e6 * float(np.median(sy)), 1), "queue_only_us": round(1e6 * float(np.median([a for a, b in qo])), 1),
               "queue_then_wait_us": round(1e6 * float(np.median([b for a, b in qo])), 1)}
print(json.dumps({"pool_of_one": resp}))
