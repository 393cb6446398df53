#!/usr/bin/env python3
"""One workload of the bench line, alone, for rocprofv3 (tools/profile_round.sh): the program goes directly behind `--`.

    python3 tools/prof_workload.py nat|u|legacy|mixed64|post12|post10|post14|config5 [launches]

nat / u: BASELINE config 3 (240 x 3840x2160 12-bit type 7, Nat or uniform); legacy: 32 x 4000x3000 12-bit type 6;
mixed64: BASELINE config 4; post12: config 3 with black levels + 12-bit strips; config5: 120 x 7680x4320 12-bit.
Prints one JSON line: the workload, launches, wall ms per batch and this box's yardstick (fill / copy rate)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import bench
import motioncam_decoder_amd as M
from motioncam_decoder_amd import shard


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "nat"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    L = bench.synth_lib()
    ctx = M.Context(0)
    calib = bench.box_calibration(torch, dev)
    out = {"workload": name, "launches": reps, "box_calibration": calib}
    if name in ("nat", "u", "post12", "post10", "post14", "config5"):
        a = argparse.Namespace(config=5 if name == "config5" else 3, width=7680 if name == "config5" else 3840,
                               height=4320 if name == "config5" else 2160, frames=120 if name == "config5" else 240, distinct=48,
                               nbits=12, sigma=12.0, streams=1)
        wl = bench.Workload(torch, M, L, dev, a, "u" if name == "u" else "nat", shard.shard_frames(a.frames, 0, 1))
        pbits = {"post12": 12, "post10": 10, "post14": 14}.get(name)
        if pbits:
            ctx.set_post(black=[256, 256, 256, 256] if pbits != 10 else [64, 64, 64, 64], bits=pbits)
        stream = torch.cuda.current_stream().cuda_stream
        written, status = ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=stream, want_status=True)
        assert all(s == 0 for s in status)
        for _ in range(24):  # (the XCD mapping of k7_tiles and the split of the side streams are measured on the first launches)
            ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=stream, want_status=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=stream, want_status=False)
        torch.cuda.synchronize()
        out["ms_per_batch"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        out["algorithmic_bytes_per_batch"] = wl.in_bytes + (wl.frames * wl.h * L.post_row_bytes(wl.w, bits=pbits) if pbits else wl.out_bytes)
        ctx.set_post()
    elif name == "legacy":
        r = bench.legacy_leg(torch, ctx, M, L, dev, reps=reps)
        out.update({k: r[k] for k in ("ms_per_batch", "kernels_ms", "bit_exact")})
        out["algorithmic_bytes_per_batch"] = r.get("algorithmic_bytes_per_batch")
    elif name == "mixed64":
        r = bench.mixed64_leg(torch, ctx, M, L, dev, reps=reps)
        out.update({k: r[k] for k in ("ms_per_batch", "bit_exact")})
    else:
        raise SystemExit("unknown workload " + name)
    out["box_calibration_after"] = bench.box_calibration(torch, dev)
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
