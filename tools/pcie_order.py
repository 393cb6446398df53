#!/usr/bin/env python3
"""Does what a process did with the GPU BEFORE mcraw_ctx_create change the host-memory pipeline's rate?  (HIP streams of one priority
share four hardware queues; which of a context's twenty-odd streams end up together depends on the queues that exist already.)
   python3 tools/pcie_order.py ctx_first | torch_first | torch_stream_first"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
import motioncam_decoder_amd as M
from motioncam_decoder_amd import benchlib

mode = sys.argv[1] if len(sys.argv) > 1 else "ctx_first"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if mode == "torch_first":
    torch.ones(4, device=dev).sum().item()
elif mode == "torch_stream_first":
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.ones(4, device=dev).sum().item()
ctx = M.Context(0)
args = types.SimpleNamespace(width=3840, height=2160, frames=240, distinct=8, config=3, nbits=12, sigma=12.0, streams=1)
L = bench.synth_lib()
wl = bench.Workload(torch, M, L, dev, args, "nat", list(range(args.frames)))
comm = benchlib.Comm(None)
r = [bench.pcie_inclusive(M, L, ctx, wl, comm, None, nframes=240)["frames_per_s"] for _ in range(3)]
print(mode, r, flush=True)
ctx.close()
