#!/usr/bin/env python3
"""A/B of library settings that are read when a context is created, in ONE process on ONE set of buffers: the same batch is decoded
in turn by contexts created under different environments.  (Fresh processes differ by +-4 % on the tile kernel alone, as the
physical pages of their buffers fall; this tool takes that out of a comparison.)

    gpurun -- 'python3 tools/ab_ctx.py auto: "split22:MCRAW_SIDE_SPLIT=2,2" "chunk16:MCRAW_XCD_CHUNK=16" [--config 5] [--rounds 9] [--steps 20]'
"""
import argparse
import os
import statistics
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help="name:ENV=value;ENV=value (environment while the variant's context is created)")
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dist", default="nat")
    ap.add_argument("--lib", action="append", default=[], help="name=path: the variant of that name loads another build (own process-wide "
                    "library: only ONE library per process, so this is for the environment variants of that build)")
    a = ap.parse_args()
    import torch
    import bench
    import motioncam_decoder_amd as M
    cw, ch, cf = {3: (3840, 2160, 240), 5: (7680, 4320, 120), 2: (4032, 3024, 96)}[a.config]
    args = types.SimpleNamespace(width=cw, height=ch, frames=a.frames or cf, distinct=48 if a.config != 5 else 4, config=a.config, nbits=12,
                                 sigma=12.0, streams=1)
    dev = torch.device("cuda", 0)
    L = bench.synth_lib()
    wl = bench.Workload(torch, M, L, dev, args, a.dist, list(range(args.frames)))
    ctxs = []
    for v in a.variants:
        name, _, envs = v.partition(":")
        kv = [e.split("=", 1) for e in envs.split(";") if e]
        old = {k: os.environ.get(k) for k, _ in kv}
        for k, val in kv:
            os.environ[k] = val
        lib = dict(kv).get("LIB")  # LIB=<name>: this variant's context comes from lib/libmcraw_hip_<name>.so (another build, same process)
        if lib:
            os.environ["MCRAW_LIB_PATH"] = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "libmcraw_hip_%s.so" % lib)
        M._lib = None  # (every variant binds its own handle: Context keeps the one it was created with)
        ctx = M.Context(0)
        os.environ.pop("MCRAW_LIB_PATH", None)
        for k, _ in kv:
            if old[k] is None:
                del os.environ[k]
            else:
                os.environ[k] = old[k]
        ctxs.append((name, ctx))
    for name, ctx in ctxs:  # first launches: statuses, the library's own measurements (XCD mapping, side parts)
        _, status = ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, want_status=True)
        assert all(s == 0 for s in status), name
        for _ in range(32):
            ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, want_status=False)
        ctx.synchronize()
        assert wl.verify(torch, [0, args.frames - 1]), name
        ctx.profile(only=("k7_tiles",), every=4)
        ctx.kernel_ms("k7_tiles", reset=True)
    res = {name: [] for name, _ in ctxs}
    for r in range(a.rounds):
        for name, ctx in (ctxs if r % 2 == 0 else ctxs[::-1]):
            for _ in range(3):
                ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, want_status=False)
            ctx.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, want_status=False)
            ctx.synchronize()
            res[name].append((time.perf_counter() - t0) * 1e3 / a.steps)
    bytes_step = wl.in_bytes + wl.out_bytes
    for name, ctx in ctxs:
        ms, n = ctx.kernel_ms("k7_tiles", reset=True)
        med = statistics.median(res[name])
        print("%-10s ms_per_step median %.4f min %.4f max %.4f  step_frac %.4f  k7_tiles avg %.4f (%d launches)  gap %.4f  xcd %s parts %s" % (
            name, med, min(res[name]), max(res[name]), bytes_step / (med * 1e-3) / 8e12, ms / max(n, 1), n, med - ms / max(n, 1),
            ctx.xcd_runs(), ctx.side_parts()), flush=True)
        st = ctx.synchronize(args.frames)
        assert all(s == 0 for s in st)
        ctx.close()


if __name__ == "__main__":
    main()
