#!/usr/bin/env python3
"""bench.py -- MCRAW frame-decode throughput on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the decode path (mcraw_decode_batch through the C ABI,
kernels k7_side -> k7_tiles) over one batch of synthetic
frames that are already resident in HBM.  Workload = BASELINE.json config 3:
240 frames of 3840x2160 12-bit, current (type 7) encoding, per GPU (--config 5:
120 frames of 7680x4320 per GPU).  Frames shard by index, no collective on the
data path (weak scaling: every rank decodes its own frames); torch.distributed
only carries the barriers and the max of the per-rank times.

Timed region: W warm-up steps, then rounds of exactly K steps, each round
bracketed by barrier + synchronise; as many rounds as it takes to cover 0.5 s
(motioncam_decoder_amd/benchlib.py).  `ms_per_step` is the median round / K,
the spread is reported beside it.

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import struct
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):  # the package; the checkers' doors (oracle/doors.py: verification and cpu_baseline only)
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=(3, 5),
                    help="BASELINE.json config: 3 = 240 x 3840x2160 per GPU (the metric's workload), 5 = 120 x 7680x4320 per GPU")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default: the config's)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="timed rounds cover at least this long")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) legs")
    ap.add_argument("--dist-backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--all-on-device0", action="store_true",
                    help="rehearsal of the multi-rank path on a one-GPU box: every rank decodes on cuda:0 (use with --dist-backend gloo)")
    ap.add_argument("--stub-decode", action="store_true",
                    help="CPU rehearsal of the multi-rank protocol (gloo, no GPU, the step is a sleep): tests only")
    ap.add_argument("--distinct", type=int, default=48, help="distinct synthetic frames generated per rank")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--nbits", type=int, default=12)
    ap.add_argument("--sigma", type=float, default=12.0)
    ap.add_argument("--dist", choices=["nat", "u"], default="nat")
    ap.add_argument("--streams", type=int, default=1,
                    help="experiment: submit the (independent) steps round-robin on this many HIP streams, own output "
                         "buffers each, so that the chain-resolve kernels of one batch run beside the HBM-bound unpack "
                         "kernel of the previous one (2 streams: +4 %% throughput, but the overlapped unpack launches "
                         "each get 4-5 %% longer, so the default stays 1)")
    ap.add_argument("--no-also", action="store_true", help="skip the second (other distribution) measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work per cpu_baseline pass (all threads, then one thread)")
    a = ap.parse_args()
    cw, ch, cf = {3: (3840, 2160, 240), 5: (7680, 4320, 120)}[a.config]
    a.width = a.width or cw
    a.height = a.height or ch
    a.frames = a.frames or cf
    return a


def synth_lib():
    """One namespace for the helpers the legs use: the build's encoder and image generator (the package's synthlib) and the
    checkers' doors (oracle/doors.py -- the oracle and the real reference: algorithmic byte counts, verification outside the timed
    regions, cpu_baseline; never the thing measured)."""
    import types
    import doors
    from motioncam_decoder_amd import synthlib
    ns = {k: v for k, v in vars(synthlib).items() if not k.startswith("__")}
    ns.update({k: v for k, v in vars(doors).items() if not k.startswith("__")})
    return types.SimpleNamespace(**ns)


def make_frames(L, seeds, w, h, nbits, dist, sigma):
    """One (image, encoded buffer) pair per seed, generated and encoded in parallel on the host."""
    def one(seed):
        img = L.synth_image(w, h, nbits, 1 if dist == "nat" else 0, sigma, seed)
        return img, L.encode7(img)
    with ThreadPoolExecutor(max_workers=max(1, min(os.cpu_count() or 1, 32))) as ex:
        return list(ex.map(one, seeds))


def frame_seeds(gidx, distinct, frames, config):
    """Seeds of the images a rank generates: global frame g holds image g mod `distinct` (seed 1000 * config + g mod distinct)
    whatever the world size; the rank's local frame i is global frame gidx[i] = rank + i * world, and the images it needs repeat
    with period d = distinct / gcd(world, distinct), so that seeds[i % d] is local frame i's."""
    D = max(1, distinct)
    stride_g = (gidx[1] - gidx[0]) if len(gidx) > 1 else 1
    d = max(1, min(D // math.gcd(stride_g, D), frames))
    return [1000 * config + (g % D) for g in gidx[:d]]


class Workload:
    """A batch of `frames` type-7 frames resident in HBM (inputs at distinct addresses)."""

    def __init__(self, torch, M, L, dev, args, dist, gidx):
        """gidx: global frame indices of this rank (frame i of the job lives on rank i % world);
        frame g is generated from seed 1000 * 3 + g (config 3 of SURVEY 8d)."""
        self.w, self.h = args.width, args.height
        self.frames = args.frames
        # What a frame holds depends on its GLOBAL index alone -- frame g is image (g mod distinct) -- so that the per-frame
        # checksums of a job do not depend on how many ranks share it.  Local frame i is global frame rank + i * world; the
        # images a rank needs repeat with period d = distinct / gcd(world, distinct): pairs[i % d] is local frame i's.
        self.gidx = list(gidx[: self.frames])
        seeds = frame_seeds(gidx, args.distinct, self.frames, args.config)
        self.pairs = make_frames(L, seeds, self.w, self.h, args.nbits, dist, args.sigma)
        lens = [p[1].size for p in self.pairs]
        d = len(self.pairs)
        stride = [(x + 255) // 256 * 256 for x in lens]
        offs, o = [], 0
        for i in range(self.frames):
            offs.append(o)
            o += stride[i % d]
        self.t_in = torch.empty(o, dtype=torch.uint8, device=dev)
        for i in range(self.frames):
            src = torch.from_numpy(self.pairs[i % d][1])
            self.t_in[offs[i]: offs[i] + lens[i % d]].copy_(src, non_blocking=False)
        self.out_stride = self.w * self.h * 2
        # one set of output buffers (and frame descriptors) per stream the steps cycle over
        self.t_outs, self.desc_sets = [], []
        for _ in range(max(1, args.streams)):
            t_out = torch.zeros(self.frames * self.out_stride, dtype=torch.uint8, device=dev)
            descs = [(self.t_in.data_ptr() + offs[i], lens[i % d], self.w, self.h, M.TYPE_BLOCK,
                      t_out.data_ptr() + i * self.out_stride, self.w * self.h) for i in range(self.frames)]
            self.t_outs.append(t_out)
            self.desc_sets.append(M.Context.make_frames(descs))
        self.t_out, self.descs = self.t_outs[0], self.desc_sets[0]
        if os.environ.get("MCRAW_BENCH_DEBUG"):
            print("buffers: in %#x (%d B) out %#x (%d B)" % (self.t_in.data_ptr(), self.t_in.numel(), self.t_out.data_ptr(), self.t_out.numel()),
                  file=sys.stderr, flush=True)
        orc = L.oracle()
        used = [orc.mcraw_oracle_len_used7(L._ptr(p[1]), p[1].size) for p in self.pairs]
        assert all(u > 0 for u in used)
        self.in_bytes = sum(used[i % d] for i in range(self.frames))      # algorithmic input bytes / step
        self.out_bytes = self.frames * self.out_stride                    # algorithmic output bytes / step
        self.pixels = self.frames * self.w * self.h
        self.bpp = 8.0 * sum(lens) / (d * self.w * self.h)

    def verify(self, torch, idx, which=0):
        """Round trip: decoded frame == the image the encoder was given."""
        d = len(self.pairs)
        for i in idx:
            got = self.t_outs[which][i * self.out_stride:(i + 1) * self.out_stride].cpu().numpy().view(np.uint16)
            if not np.array_equal(got.reshape(self.h, self.w), self.pairs[i % d][0]):
                return False
        return True


def buffer_checksum(torch, t, wgt_cache={}):
    """32-bit position-weighted checksum of one decoded frame (a uint8 tensor on a GPU), computed there."""
    n32 = t.numel() // 4
    key = (str(t.device), n32)
    if key not in wgt_cache:
        wgt_cache.clear()
        wgt_cache[key] = (torch.arange(n32, device=t.device, dtype=torch.int64) % 65521) + 1
    return int((t[: n32 * 4].view(torch.int32).to(torch.int64) * wgt_cache[key]).sum().item()) & 0xFFFFFFFF


def frame_checksums(torch, comm, wl, nglobal):
    """32-bit checksums of the decoded global frames 0 .. nglobal - 1 of the job (every rank sums the ones it holds, on its GPU;
    one SUM all-reduce merges the slots) and a digest over them: equal at every rank count, since frame g's content depends on g
    alone (Workload) and its pixels on nothing but its bytes (lib/RawData.cpp:528-612 is a pure function of one buffer)."""
    slots = [0.0] * nglobal
    for i, g in enumerate(wl.gidx):
        if g < nglobal:
            slots[g] = float(buffer_checksum(torch, wl.t_out[i * wl.out_stride:(i + 1) * wl.out_stride]))
    vals = [int(v) for v in comm.sum(slots)]
    return vals, hashlib.sha1(struct.pack("<%dI" % len(vals), *vals)).hexdigest()[:16]


def run_timed(torch, comm, ctx, M, wl, args):
    """Warm-up, the timed rounds, then k7_side's time (untimed).  Returns (round times [s, max over ranks], {k7_side ms, the
    XCD mapping, this rank's own ms per step}, (k7_tiles ms summed over the sampled launches of the timed rounds, launches), ok)."""
    from motioncam_decoder_amd import benchlib
    steps, warmup = args.steps, args.warmup
    nset = len(wl.desc_sets)
    cur = torch.cuda.current_stream()
    streams = [cur] if nset == 1 else [torch.cuda.Stream() for _ in range(nset)]
    for s in streams:
        s.wait_stream(cur)
    # first pass with statuses: every frame must decode
    written, status = ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=streams[0].cuda_stream, want_status=True)
    assert all(s == 0 for s in status), "decode failed: %s" % [hex(s) for s in status if s][:4]
    assert all(wr == wl.w * wl.h for wr in written)
    ok = wl.verify(torch, sorted({0, wl.frames // 2, wl.frames - 1}))
    for _ in range(6):  # (the library times its XCD mapping of k7_tiles on the first launches on a new set of buffers)
        ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=streams[0].cuda_stream, want_status=False)
    torch.cuda.synchronize()
    for t in wl.t_outs:
        t.zero_()
    # timed: only the roofline kernel carries events, and only every eighth launch of it (a bracket is two event
    # records in the stream, several microseconds; the sampled launches give the same average duration)
    ctx.profile(only=("k7_tiles",), every=8)
    ctx.kernel_ms("k7_tiles", reset=True)
    counted = [0]

    def step(i):
        ctx.decode_batch(wl.desc_sets[i % nset], mem=M.MEM_DEVICE, stream=streams[i % nset].cuda_stream, want_status=False)
        counted[0] += 1

    own = []
    times = benchlib.timed_rounds(step, torch.cuda.synchronize, comm, steps, max(0, warmup - 1), args.min_seconds, own=own)
    st = ctx.synchronize(wl.frames)
    ok = ok and all(s == 0 for s in st)
    for which in range(min(nset, steps)): # every buffer set the timed steps wrote
        ok = ok and wl.verify(torch, sorted({min(1, wl.frames - 1), wl.frames // 3, max(0, wl.frames - 2)}), which)
    tiles = ctx.kernel_ms("k7_tiles", reset=True) # (ms summed over warm-up + timed rounds, launches)
    # untimed, behind the timed rounds: a few steps with k7_side between two event records (the tile kernel's time is the one
    # sampled in the timed rounds: a bracket around every launch costs the stream several microseconds and made the tile
    # kernel read longer than the whole step)
    ctx.profile(only=("k7_side",))
    for i in range(7):
        if i == 1: # the first step after the synchronise runs on an idle, down-clocked GPU
            ctx.kernel_ms("k7_side", reset=True)
        ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=streams[0].cuda_stream, want_status=False)
    torch.cuda.synchronize()
    kms = {"k7_side": ctx.kernel_ms("k7_side", reset=True)[0] / 6.0}
    kms["xcd_runs"] = ctx.xcd_runs()  # the mapping of k7_tiles' workgroups the library measured as the faster one here
    kms["own_ms_per_step"] = 1e3 * sorted(own)[len(own) // 2] / steps  # this rank's own median round (in front of the closing barrier)
    ctx.profile(True)
    return times, kms, tiles, ok


def box_calibration(torch, dev, nbytes=1 << 30, reps=8):
    """This box's own yardstick, measured in this process: what a plain write-only stream (torch fill, 1 GiB, beyond
    the Infinity Cache) and a plain copy reach right now.  Boxes of the pool -- and runs on one box -- differ by several
    per cent on the HBM-bound kernel; roofline.frac can be normalised with these."""
    buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    src = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)

    def rate(fn, moved):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return reps * moved / (e0.elapsed_time(e1) * 1e-3) / 1e9

    res = {"fill_write_only_GBs": round(rate(lambda: buf.fill_(7), nbytes), 1),
           "copy_read_plus_write_GBs": round(rate(lambda: buf.copy_(src), 2 * nbytes), 1)}
    del buf, src
    return res


def link_probe(torch, dev, nbytes=256 << 20, reps=3):
    """The host link of this GPU, measured now: pinned copies up, down, and both at once (GB/s)."""
    h1 = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    h2 = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d1 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def t(fn):
        for _ in range(2):  # the first copies of a fresh pinned buffer run far below the link rate
            fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def both():
        with torch.cuda.stream(s1):
            d1.copy_(h1, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(d2, non_blocking=True)

    up = nbytes / t(lambda: d1.copy_(h1, non_blocking=True)) / 1e9
    down = nbytes / t(lambda: h2.copy_(d2, non_blocking=True)) / 1e9
    duplex = nbytes / t(both) / 1e9
    return {"h2d_GBs": round(up, 1), "d2h_GBs": round(down, 1), "each_way_when_both_GBs": round(duplex, 1)}


def mixed64_leg(torch, ctx, M, L, dev, reps=40):
    """BASELINE config 4, timed: 64 frames, 14-bit type 7 (U and Nat) interleaved with legacy frames (10/12/14-bit,
    one width with w % 32 != 0), 1920x1080 and 12 MP; verified against the images the encoder was given."""
    items = []
    for i in range(64):
        if i % 2 == 0:
            w, h = ((1920, 1080), (4032, 3024))[(i // 2) % 2]
            img = L.synth_image(w, h, 14, (i // 4) % 2, 40.0, 4000 + i)
            items.append((M.TYPE_BLOCK, img, L.encode7(img)))
        else:
            w, h = ((1920, 1080), (4000, 3000))[(i // 2) % 2]
            img = L.synth_image(w, h, (10, 12, 14)[(i // 2) % 3], (i // 4) % 2, 12.0, 4000 + i)
            items.append((M.TYPE_LEGACY, img, L.encode6(img, None, flags=1)))
    tin = [torch.from_numpy(it[2]).to(dev) for it in items]
    tout = [torch.zeros(it[1].size * 2, dtype=torch.uint8, device=dev) for it in items]
    descs = [(tin[i].data_ptr(), tin[i].numel(), it[1].shape[1], it[1].shape[0], it[0], tout[i].data_ptr(), it[1].size)
             for i, it in enumerate(items)]
    frames = M.Context.make_frames(descs)
    written, status = ctx.decode_batch(frames)
    ok = all(s == 0 for s in status)
    for i, it in enumerate(items):
        ok = ok and np.array_equal(tout[i].cpu().numpy().view(np.uint16).reshape(it[1].shape), it[1])
    ctx.profile(False) # (no events between the batches: they follow each other on the stream, as the bench line's steps do)
    for _ in range(24): # (the split of the side streams is measured on the first launches of a geometry)
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    ctx.profile(True)
    px = sum(it[1].size for it in items)
    byts = sum(it[2].size for it in items) + 2 * px
    return {"workload": "config 4: 64 frames, type 7 (14-bit U/Nat) and type 6 (10/12/14-bit) interleaved, 1920x1080 / 4032x3024 / 4000x3000",
            "ms_per_batch": round(t * 1e3, 4), "batches_timed": reps, "mpix_s": round(px / t / 1e6, 1),
            "in_plus_out_GBs": round(byts / t / 1e9, 1), "side_parts": ctx.side_parts(), "bit_exact": bool(ok)}


def cpu_baseline(L, wl, seconds):
    """The reference codec (oracle/_ref, built from the reference's own sources) or, when that
    build is absent, the oracle port, timed on this host: frame-parallel over all cores, and
    single-threaded (the reference itself is single-threaded)."""
    ref = L.ref()
    if ref is not None:
        kind, fn = "reference", ref.mcraw_ref_time_batch
        # (the library was built in the build container and travels prebuilt: -march=native there would be another CPU's)
        flags = "g++ -O3 -march=x86-64-%s -include cstring -include algorithm (oracle/Makefile ref; %s)" % (
            "v3" if "_v3" in (L.ref_path() or "") else "v2", os.path.basename(L.ref_path() or ""))
    else:
        kind = "port"
        flags = "gcc -O3 -march=native (oracle/mcraw_oracle.c, built on this host)"
        native = os.path.join(ROOT, "oracle", "libmcraw_oracle_native.so")
        try:
            subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-shared", "-o", native,
                            os.path.join(ROOT, "oracle", "mcraw_oracle.c"), "-lpthread"], check=True)
            lib = C.CDLL(native)
        except Exception:
            lib = L.oracle()
        fn = lib.mcraw_oracle_time_batch
        fn.restype = C.c_double
        fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    bufs = [p[1] for p in wl.pairs][:32]
    n = len(bufs)
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    try:
        cores = len(os.sched_getaffinity(0)) # (the CPUs this process may run on)
    except AttributeError:
        cores = os.cpu_count() or 1
    res = {}
    for label, th in (("all", cores), ("one", 1)):
        t = fn(7, wl.w, wl.h, ptrs, lens, n, th, 1)  # one calibration pass
        if t <= 0:
            return None
        reps = max(1, int(seconds / t))
        t = fn(7, wl.w, wl.h, ptrs, lens, n, th, reps)
        res[label] = (n * reps * wl.w * wl.h) / t / 1e6
        res[label + "_s"] = t
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(res["all"], 1), "unit": "MPixels/s", "cores": cores, "kind": kind,
            "flags": flags,
            "value_1thread": round(res["one"], 1), "cpu": model,
            "sample": "%d distinct %dx%d %d-bit frames of this workload, decode only (inputs in RAM), "
                      "%.1f s on %d threads + %.1f s on 1 thread" % (n, wl.w, wl.h, 12, res["all_s"], cores, res["one_s"])}


def pcie_inclusive(M, L, ctx, wl, comm, link=None, nframes=240, reps=2, pack12=False, bits=None):
    """End-to-end rate when the boundary hands over HOST buffers (never `value`): frames in pinned
    memory -> H2D -> decode -> D2H into pinned memory, sub-batches pipelined on the context's
    streams (mcraw_decode_batch, MCRAW_MEM_HOST).  pack12: with the fused 12-bit strip stage, which
    sends 1.5 instead of 2 bytes per sample back over the link.
    Every rank runs the SAME sequence of collectives whatever happens to it: a rank that fails (allocation, a frame
    status, a mismatch) says so in the reduction behind each phase, and all ranks leave the leg together -- no rank
    is left waiting in a barrier for one that raised."""
    lib = M.load()
    d = len(wl.pairs)
    n = min(nframes, wl.frames)
    bits = bits or (12 if pack12 else None)
    row_bytes = L.post_row_bytes(wl.w, bits=bits)
    out_bytes = wl.h * row_bytes
    ins, outs, descs = [], [], []
    err = None

    def all_ok():
        return comm.min([0.0 if err else 1.0])[0] > 0.5

    try:
        try:
            if bits:
                ctx.set_post(bits=bits)
            for i in range(n):
                buf = wl.pairs[i % d][1]
                pi = lib.mcraw_host_alloc(buf.size)
                po = lib.mcraw_host_alloc(out_bytes)
                if not pi or not po:
                    raise MemoryError("mcraw_host_alloc failed")
                ins.append(pi)
                outs.append(po)
                C.memmove(pi, buf.ctypes.data, buf.size)
                descs.append((pi, buf.size, wl.w, wl.h, M.TYPE_BLOCK, po, out_bytes // 2))
            frames = M.Context.make_frames(descs)
            ctx.decode_batch(frames, mem=M.MEM_HOST)  # warm-up: staging buffers get allocated
            for _ in range(3):  # (the context compares its second and fourth large batch: one fetches its status words, one has them sent home)
                ctx.decode_batch(frames, mem=M.MEM_HOST)
        except Exception as e:
            err = e
        if not all_ok():
            return {"error": repr(err) if err else "another rank failed in the set-up of this leg"}
        comm.barrier()                             # every rank pulls on the host's memory and its own link at the same time
        t_local, ok = 1e9, False
        try:
            t0 = time.perf_counter()
            for _ in range(reps):
                written, status = ctx.decode_batch(frames, mem=M.MEM_HOST)
            t_local = (time.perf_counter() - t0) / reps
            ok = all(s == 0 for s in status)
            got = np.ctypeslib.as_array(C.cast(outs[0], C.POINTER(C.c_uint8)), shape=(wl.h, row_bytes))
            ok = ok and np.array_equal(got, L.oracle_post(wl.pairs[0][0], None, bits=bits))
        except Exception as e:
            err = e
        comm.barrier()
        t = comm.max([t_local])[0]                 # the job's rate is set by its slowest rank
        own_rates = comm.gather(n / t_local if t_local < 1e8 else 0.0)
        ok = bool(comm.min([1.0 if ok else 0.0])[0] > 0.5)
        if not all_ok():
            return {"error": repr(err) if err else "another rank failed in the timed part of this leg"}
        in_b = sum(wl.pairs[i % d][1].size for i in range(n))
        world = comm.world
        res = {"frames_per_rank": n, "mpix_s": round(world * n * wl.w * wl.h / t / 1e6, 1), "frames_per_s": round(world * n / t, 1),
               "frames_per_s_per_rank": round(n / t, 1), "h2d_GBs_per_rank": round(in_b / t / 1e9, 2),
               "d2h_GBs_per_rank": round(n * out_bytes / t / 1e9, 2), "bit_exact": ok,
               "note": "pinned host buffers in and out; sub-batches flow through upload stream / kernels / download stream; PCIe-bound"
                       + (("; %d-bit strips out (mcraw_ctx_set_post)" % bits) if bits else "")
                       + ("; all ranks at once" if world > 1 else "")}
        if world > 1:
            res["frames_per_s_by_rank"] = [round(v, 1) for v in own_rates]
        res["status_words"] = {None: "undecided", 0: "fetched", 1: "sent"}[ctx.host_way()]
        if link:
            # time the two directions need at the rates this link showed with both directions busy
            t_link = max(in_b / (link["each_way_when_both_GBs"] * 1e9), n * out_bytes / (link["each_way_when_both_GBs"] * 1e9))
            res["link_frac"] = round(t_link / t_local, 3)
        return res
    finally:
        try:
            ctx.set_post()
        except Exception:
            pass
        for p in ins + outs:
            lib.mcraw_host_free(p)


def pool_leg(torch, M, L, wl, devices, link=None, nframes=240, reps=2):
    """The PRODUCT's multi-GPU driver (mcraw_pool_*: one context and one NUMA-bound host thread per GPU, frame i ->
    member i mod G) on this workload: host buffers in and out (pinned memory allocated by each member's own thread), and
    buffers resident in each member's HBM (mcraw_pool_decode_batch_device, BASELINE config 5's form).  One process
    drives every GPU in `devices` -- the counterpart of the one-process-per-GPU numbers above."""
    lib = M.load()
    pool = M.Pool(devices)
    G = pool.size
    d = len(wl.pairs)
    n = min(nframes, wl.frames)
    out_bytes = wl.w * wl.h * 2
    ins, outs, descs = [], [], []
    try:
        for i in range(n):
            buf = wl.pairs[i % d][1]
            pi, po = pool.host_alloc(i % G, buf.size), pool.host_alloc(i % G, out_bytes)
            assert pi and po
            ins.append(pi)
            outs.append(po)
            C.memmove(pi, buf.ctypes.data, buf.size)
            descs.append((pi, buf.size, wl.w, wl.h, M.TYPE_BLOCK, po, out_bytes // 2))
        frames = M.Context.make_frames(descs)
        pool.decode_batch(frames)  # warm-up: staging buffers get allocated
        t0 = time.perf_counter()
        for _ in range(reps):
            written, status = pool.decode_batch(frames)
        t = (time.perf_counter() - t0) / reps
        ok = all(s == 0 for s in status)
        got = np.ctypeslib.as_array(C.cast(outs[n - 1], C.POINTER(C.c_uint16)), shape=(wl.h, wl.w))
        ok = ok and np.array_equal(got, wl.pairs[(n - 1) % d][0])
        in_b = sum(wl.pairs[i % d][1].size for i in range(n))
        res = {"devices": pool.devices(), "numa_cpus": pool.numa_cpus(),
               "host_buffers": {"frames": n, "frames_per_s": round(n / t, 1), "mpix_s": round(n * wl.w * wl.h / t / 1e6, 1),
                                "h2d_GBs": round(in_b / t / 1e9, 2), "d2h_GBs": round(n * out_bytes / t / 1e9, 2), "bit_exact": bool(ok),
                                "note": "the pool's contexts share their GPU with this process's own context (and with each other where a device "
                                        "is named twice): the host-memory pipeline then takes its long way (status words fetched at the wait, no "
                                        "pieces), as before round 5's change; pcie_inclusive is the leg of a context that has its device to itself"}}
        if link:
            per = max(in_b, n * out_bytes) / G / (link["each_way_when_both_GBs"] * 1e9)  # every member has a link of its own
            res["host_buffers"]["link_frac"] = round(per / t, 3)
    finally:
        for p_ in ins + outs:
            lib.mcraw_host_free(p_)
    # resident: frame i in the HBM of member i % G
    try:
        devs = pool.devices()
        tin, tout, descs = [], [], []
        for i in range(n):
            dev = torch.device("cuda", devs[i % G])
            ti = torch.from_numpy(wl.pairs[i % d][1]).to(dev)
            to = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
            tin.append(ti)
            tout.append(to)
            descs.append((ti.data_ptr(), ti.numel(), wl.w, wl.h, M.TYPE_BLOCK, to.data_ptr(), out_bytes // 2))
        for dv in sorted(set(devs)):
            torch.cuda.synchronize(dv)
        frames = M.Context.make_frames(descs)
        for _ in range(12): # (a member's context first measures its XCD mapping on these buffers: not what a call costs)
            pool.decode_batch_device(frames)
        t0 = time.perf_counter()
        for _ in range(10):
            written, status = pool.decode_batch_device(frames)
        t = (time.perf_counter() - t0) / 10
        ok = all(s == 0 for s in status)
        got = tout[n - 1].cpu().numpy().view(np.uint16).reshape(wl.h, wl.w)
        ok = ok and np.array_equal(got, wl.pairs[(n - 1) % d][0])
        # every frame against what ONE context on ONE GPU made of it (the bench line's buffers): per-frame checksums
        same = sum(1 for i in range(n) if buffer_checksum(torch, tout[i]) == buffer_checksum(torch, wl.t_out[i * wl.out_stride:(i + 1) * wl.out_stride]))
        res["resident"] = {"frames": n, "ms_per_batch": round(t * 1e3, 4), "frames_per_s": round(n / t, 1),
                           "mpix_s": round(n * wl.w * wl.h / t / 1e6, 1), "bit_exact": bool(ok),
                           "frames_equal_to_single_gpu_decode": same,
                           "note": "synchronous call: host hand-off, kernels and status read-back of every member"}
        # the same batches queued back to back (no status asked for), one mcraw_pool_synchronize behind them
        for _ in range(8):
            pool.decode_batch_device(frames, want_status=False)
        st = pool.synchronize(n)
        t0 = time.perf_counter()
        for _ in range(20):
            pool.decode_batch_device(frames, want_status=False)
        st = pool.synchronize(n)
        t = (time.perf_counter() - t0) / 20
        res["resident_queued"] = {"frames": n, "ms_per_batch": round(t * 1e3, 4), "frames_per_s": round(n / t, 1),
                                  "mpix_s": round(n * wl.w * wl.h / t / 1e6, 1), "bit_exact": bool(ok and all(x == 0 for x in st))}
    finally:
        pool.close()
    return res


def post_stage(torch, ctx, M, L, wl, steps, bits=12):
    """The same batch with the fused post-decode stage: black levels subtracted and rows
    written as `bits`-bit DNG strips -- 1.5 (1.25, 1.75) instead of 2 output bytes per sample.  Not the bench line."""
    stream = torch.cuda.current_stream().cuda_stream
    black = [256, 256, 256, 256] if bits != 10 else [64, 64, 64, 64]
    ctx.set_post(black=black, bits=bits)
    try:
        for t in wl.t_outs[:1]:
            t.zero_()
        ctx.profile(only=("k7_tiles",))
        for _ in range(8):  # (the XCD mapping is chosen per row format: measured on the first launches with this one)
            ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=stream, want_status=False)
        torch.cuda.synchronize()
        ctx.kernel_ms("k7_tiles", reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.decode_batch(wl.descs, mem=M.MEM_DEVICE, stream=stream, want_status=False)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = ctx.synchronize(wl.frames)
        tile_ms, tile_n = ctx.kernel_ms("k7_tiles", reset=True)
        rb = L.post_row_bytes(wl.w, bits=bits)
        ok = all(s == 0 for s in st)
        d = len(wl.pairs)
        for i in sorted({0, wl.frames - 1}):
            got = wl.t_out[i * wl.out_stride: i * wl.out_stride + wl.h * rb].cpu().numpy().reshape(wl.h, rb)
            ok = ok and np.array_equal(got, L.oracle_post(wl.pairs[i % d][0], black, bits=bits))
        out_b = wl.frames * wl.h * rb
        ach = (wl.in_bytes + out_b) / (tile_ms / max(tile_n, 1) * 1e-3) / 1e9
        bound = {12: "vector issue and the memory pipeline's instruction rate: two stores per lane and call like the plain kernel, 12 instead of 16 "
                     "bytes each, every 64-byte line written once (L1->L2 write requests = bytes / 64, profiles/ and docs/lab_notes.md); the "
                     "stage's arithmetic is 14 vector instructions per lane and call for items that take the lean path (DESIGN 3)",
                 10: "four stores per lane and call (8 + 2 bytes per row piece at 2-byte alignment): every line of a row piece is touched by two "
                     "instructions, 2.0 x the L1->L2 write requests its bytes need; the merged form (16-byte stores 10 bytes apart) writes six "
                     "bytes twice and is slower still (these 12-bit frames saturate at 1023 on the way, like the oracle's)",
                 14: "ONE 16-byte store per lane and row in the frame's interior (round 5: the next row piece's first two bytes are fetched "
                     "across lanes and written by both lanes; 12 + 2 bytes per lane, round 4's form, touched every line twice: 1.24 -> 1.15 ms)"}[bits]
        out = {"stage": "black levels %s subtracted, rows as %d-bit strips" % (black, bits), "ms_per_step": round(1e3 * el / steps, 4),
               "mpix_s": round(wl.pixels * steps / el / 1e6, 1), "tiles_ms_per_launch": round(tile_ms / max(tile_n, 1), 4),
               "algorithmic_bytes_per_launch": wl.in_bytes + out_b, "achieved_gbs": round(ach, 1),
               "frac": round(ach / HBM_PEAK_GBS, 4), "xcd_runs": ctx.xcd_runs(), "bit_exact": bool(ok), "bound": bound}
        return with_frac_profile(out, "post%d" % bits, "k7_tiles", wl.in_bytes + out_b)
    finally:
        ctx.set_post()
        ctx.profile(True)


def legacy_leg(torch, ctx, M, L, dev, n=32, w=4000, h=3000, nbits=12, sigma=12.0, reps=40):
    """The legacy (type 6) encoding on a BASELINE config 4 geometry (width % 32 != 0): not the bench line,
    reported so that both codecs of the path are measured by the same program."""
    imgs = [L.synth_image(w, h, nbits, 1, sigma, 6000 + i) for i in range(4)]
    bufs = [L.encode6(im) for im in imgs]
    tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(n)]
    tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
    descs = [(tin[i].data_ptr(), tin[i].numel(), w, h, M.TYPE_LEGACY, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)]
    frames = M.Context.make_frames(descs)
    written, status = ctx.decode_batch(frames)
    ok = all(s == 0 for s in status) and all(wr == w * h for wr in written)
    for i in (0, n - 1):
        got = tout[i * w * h * 2:(i + 1) * w * h * 2].cpu().numpy().view(np.uint16).reshape(h, w)
        ok = ok and np.array_equal(got, imgs[i % 4])
    # wall time of batches that follow each other on the stream, no events between them (as the bench line's steps are
    # timed); then the kernel alone from a sample of launches bracketed by events (every 4th: an event pair costs the stream
    # a few microseconds and keeps the next launch from starting while the previous one drains)
    ctx.profile(False)
    for _ in range(3):
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    ctx.profile(only=("k6_decode",), every=4)
    ctx.kernel_ms("k6_decode", reset=True)
    for _ in range(96):  # (24 sampled launches)
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    kt, kn = ctx.kernel_ms("k6_decode", reset=True)
    kms = {"k6_decode": round(kt / max(kn, 1), 4)}
    ctx.profile(True)
    byts = sum(bufs[i % 4].size for i in range(n)) + n * w * h * 2
    out = {"workload": "%d x %dx%d %d-bit type-6 frames, Nat" % (n, w, h, nbits), "ms_per_batch": round(t * 1e3, 4),
           "batches_timed": reps, "kernel_launches_sampled": kn,
           "mpix_s": round(n * w * h / t / 1e6, 1), "in_plus_out_GBs": round(byts / t / 1e9, 1),
           "input_bpp": round(8.0 * bufs[0].size / (w * h), 2), "kernels_ms": kms, "bit_exact": bool(ok)}
    # the same roofline figures as for the bench line: the batch (wall) and its one kernel against the HBM peak, and the
    # HBM traffic of that kernel from the committed counter passes (tools/profile_round.sh; not measured in this run)
    out["step_frac"] = round(byts / t / 1e9 / 8000.0, 4)
    out["algorithmic_bytes_per_batch"] = byts
    if kms["k6_decode"] > 0:
        out["frac"] = round(byts / (kms["k6_decode"] * 1e-3) / 1e9 / 8000.0, 4)
    with_frac_profile(out, "legacy", "k6_decode", byts)
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_legacy_traffic.json")) as f:
                tj = json.load(f)
            out["traffic"] = tj["total_bytes_per_batch"]
            out["traffic_source"] = ("profiles/%s_legacy_traffic.json (rocprofv3 --pmc passes, committed; not measured in this run)" % tag)
            break
        except Exception:
            continue
    return out


def config5_leg(torch, ctx, M, L, dev, n=120, w=7680, h=4320, nbits=12, sigma=12.0, reps=8, distinct=4):
    """One rank's share of BASELINE config 5 (120 frames of 7680x4320, 12-bit, type 7, resident in HBM): the step against the HBM
    peak, and its two kernels -- an 8K frame's side streams are four times as long as a UHD frame's, and k7_side follows a chain.
    Not the bench line; `python bench.py --config 5` times the same workload with rounds and spread."""
    def one(seed):
        img = L.synth_image(w, h, nbits, 1, sigma, seed)
        return img, L.encode7(img)
    with ThreadPoolExecutor(max_workers=distinct) as ex:
        pairs = list(ex.map(one, [5000 + i for i in range(distinct)]))
    lens = [p[1].size for p in pairs]
    tin = [torch.from_numpy(pairs[i % distinct][1]).to(dev) for i in range(n)]  # (distinct addresses: every frame's bytes come from HBM)
    tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
    frames = M.Context.make_frames([(tin[i].data_ptr(), lens[i % distinct], w, h, M.TYPE_BLOCK, tout.data_ptr() + i * w * h * 2, w * h)
                                    for i in range(n)])
    written, status = ctx.decode_batch(frames)
    ok = all(s == 0 for s in status) and all(wr == w * h for wr in written)
    for i in (0, n - 1):
        got = tout[i * w * h * 2:(i + 1) * w * h * 2].cpu().numpy().view(np.uint16).reshape(h, w)
        ok = ok and np.array_equal(got, pairs[i % distinct][0])
    orc = L.oracle()
    used = [orc.mcraw_oracle_len_used7(L._ptr(p[1]), p[1].size) for p in pairs]
    byts = sum(used[i % distinct] for i in range(n)) + n * w * h * 2
    ctx.profile(False)
    for _ in range(28):  # (the library times the XCD mapping of k7_tiles and the split of the side streams on the first launches of a geometry)
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    kms = {}
    for kname, every, launches in (("k7_tiles", 2, 8), ("k7_side", 1, 4)):  # (one kernel at a time between event records, the tile kernel every 2nd launch)
        ctx.profile(only=(kname,), every=every)
        ctx.kernel_ms(kname, reset=True)
        for _ in range(launches):
            ctx.decode_batch(frames, want_status=False)
        torch.cuda.synchronize()
        kt, kn = ctx.kernel_ms(kname, reset=True)
        kms[kname] = round(kt / max(kn, 1), 4)
    ctx.profile(True)
    out = {"workload": "config 5, one rank's share: %d x %dx%d %d-bit type-7 frames, Nat" % (n, w, h, nbits), "ms_per_step": round(t * 1e3, 4),
           "steps_timed": reps, "mpix_s": round(n * w * h / t / 1e6, 1), "algorithmic_bytes_per_step": byts,
           "step_frac": round(byts / t / 1e9 / HBM_PEAK_GBS, 4), "kernels_ms_per_step": kms, "xcd_runs": ctx.xcd_runs(),
           "side_parts": ctx.side_parts(), "bit_exact": bool(ok)}
    if kms["k7_tiles"] > 0:
        out["frac"] = round(byts / (kms["k7_tiles"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    with_frac_profile(out, "config5", "k7_tiles", byts)
    del tin, tout
    return out


def rotating_outputs_leg(torch, ctx, M, wl, dev, sets=8, reps=64):
    """The bench line's workload for a caller that never writes the same output buffer twice in a row: `sets` sets of output
    buffers taken in turn.  The choice of k7_tiles' XCD mapping is made per geometry and re-checked by one timed launch in 64,
    so such a caller runs on a decided mapping like the bench line does (round 3 keyed the choice on the output pointer and
    kept this caller measuring)."""
    outs = [torch.zeros(wl.frames * wl.out_stride, dtype=torch.uint8, device=dev) for _ in range(sets)]
    d = len(wl.pairs)
    base = wl.t_in.data_ptr()
    offs, o = [], 0
    lens = [p[1].size for p in wl.pairs]
    for i in range(wl.frames):
        offs.append(o)
        o += (lens[i % d] + 255) // 256 * 256
    fsets = [M.Context.make_frames([(base + offs[i], lens[i % d], wl.w, wl.h, M.TYPE_BLOCK, t.data_ptr() + i * wl.out_stride, wl.w * wl.h)
                                    for i in range(wl.frames)]) for t in outs]
    ctx.profile(False)
    for b in range(sets):
        ctx.decode_batch(fsets[b], want_status=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(reps):
        ctx.decode_batch(fsets[b % sets], want_status=False)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    st = ctx.synchronize(wl.frames)
    ok = all(x == 0 for x in st)
    for t_out in (outs[0], outs[sets - 1]):
        got = t_out[(wl.frames - 1) * wl.out_stride: wl.frames * wl.out_stride].cpu().numpy().view(np.uint16).reshape(wl.h, wl.w)
        ok = ok and np.array_equal(got, wl.pairs[(wl.frames - 1) % d][0])
    ctx.profile(True)
    res = {"output_sets": sets, "steps_timed": reps, "ms_per_step": round(t * 1e3, 4), "mpix_s": round(wl.pixels / t / 1e6, 1),
           "step_frac": round((wl.in_bytes + wl.out_bytes) / t / 1e9 / HBM_PEAK_GBS, 4), "xcd_runs": ctx.xcd_runs(), "bit_exact": bool(ok)}
    del outs
    return res


def traffic_from_profile(workload_key, algorithmic_bytes=None):
    """HBM bytes per k7_tiles launch from the committed rocprofv3 --pmc summary (profiles/traffic.json): the entry of this
    (geometry, bit depth, frames, distribution) when there is one; else the measured traffic / algorithmic ratio of the entry
    with the same distribution (any geometry: the ratio is a property of the kernel's access pattern -- 1.007 at UHD and 8K alike)
    times this run's algorithmic bytes.  Returns (bytes or None, where it comes from)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
    except Exception:
        return None, None
    e = t.get(workload_key)
    if e and e.get("hbm_bytes_per_launch"):
        return e["hbm_bytes_per_launch"], ("profiles/traffic.json[%s] (rocprofv3 --pmc passes of this workload, profile %s, committed; not "
                                           "measured in this run)" % (workload_key, e.get("profile")))
    dist = workload_key.rsplit("_", 1)[-1]
    for k, e in t.items():
        if k.rsplit("_", 1)[-1] == dist and e.get("hbm_bytes_per_launch") and e.get("algorithmic_bytes") and algorithmic_bytes:
            ratio = e["hbm_bytes_per_launch"] / e["algorithmic_bytes"]
            return ratio * algorithmic_bytes, ("ratio %.4f of profiles/traffic.json[%s] (profile %s) x this run's algorithmic bytes: no --pmc "
                                               "pass of this geometry is committed" % (ratio, k, e.get("profile")))
    return None, None


def profile_kernel_ms(workload, kernel="k7_tiles"):
    """Average launch of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of a workload
    (profiles/rNN_<workload>_kernel_stats.csv of the newest round that has one), or (None, None).  Every leg of the line carries
    the fraction this gives (`frac_profile`) beside the one its own events measure, so no leg prints a fraction that profiles/ does
    not show."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_kernel_stats.csv" % workload)), reverse=True):
        try:
            with open(path) as f:
                rows = list(csv.reader(line for line in f if not line.startswith("#")))
            hdr = rows[0]
            best = None
            for r in rows[1:]:
                if kernel in r[hdr.index("Name")]:
                    tot = float(r[hdr.index("TotalDurationNs")]) if "TotalDurationNs" in hdr else float(r[hdr.index("AverageNs")])
                    if best is None or tot > best[0]:  # (several instances of a kernel: the one the workload spends its time in)
                        best = (tot, float(r[hdr.index("AverageNs")]) * 1e-6)
            if best:
                return best[1], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def with_frac_profile(out, workload, kernel, byts):
    ms, path = profile_kernel_ms(workload, kernel)
    out["frac_profile"] = round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None
    out["frac_profile_source"] = ("%s: %s AverageNs %.0f" % (path, kernel, ms * 1e6)) if ms else None
    return out


def profile_launch_ms(dist):
    return profile_kernel_ms(dist, "k7_tiles")


def stub_main(args):
    """CPU rehearsal of the multi-rank protocol (tests/test_bench_dist_gloo.py): the same process-group set-up,
    timed rounds, reductions and rank-0 JSON line as the GPU run, over gloo, with a sleep standing in for the decode."""
    import torch
    import torch.distributed as dist_mod
    from motioncam_decoder_amd import benchlib, shard
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
        dist_mod.init_process_group("gloo")
    comm = benchlib.Comm(dist_mod, "cpu")
    ranks_seen = int(round(comm.sum([1.0])[0]))
    if ranks_seen != args.gpus:
        print("bench.py: %d rank(s) took part, --gpus %d" % (ranks_seen, args.gpus), file=sys.stderr)
        return 2
    mine = shard.shard_frames(world * args.frames, rank, world)
    assert len(mine) == args.frames
    per_step = 0.002 * (1 + rank)  # the slower rank sets the job's time
    own = []
    times = benchlib.timed_rounds(lambda i: time.sleep(per_step), lambda: None, comm, args.steps, args.warmup, args.min_seconds, own=own)
    ok = bool(comm.min([1.0])[0] > 0.5)
    per_rank = {"ms_per_step": [round(v, 4) for v in comm.gather(1e3 * sorted(own)[len(own) // 2] / args.steps)]} if world > 1 else None
    if comm.rank == 0:
        st = benchlib.round_stats(times, args.steps)
        pixels = world * args.frames * args.width * args.height
        print(json.dumps({"metric": METRIC, "value": round(pixels / (st["median"] * 1e-3) / 1e6, 1), "unit": "MPixels/s",
                          "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(st["median"], 4),
                          "ms_per_step_min": round(st["min"], 4), "ms_per_step_max": round(st["max"], 4), "rounds": st["rounds"],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16",
                          "data": "stub: no decode (CPU rehearsal of the multi-rank protocol)", "bit_exact": ok, "per_rank": per_rank,
                          "config": {"workload": "stub", "frames_per_gpu": args.frames}}), flush=True)
    if comm.dist:
        dist_mod.barrier()
        dist_mod.destroy_process_group()


METRIC = "MPixels/s unpacked + achieved HBM GB/s %peak, 4K 12-bit, 1/2/4/8 GPUs"


def visible_gpus():
    """GPUs this process could use, counted WITHOUT touching HIP (sysfs: AMD render nodes, HIP_VISIBLE_DEVICES honoured)."""
    from motioncam_decoder_amd import benchlib
    n = len(benchlib.gpu_pci_devices())
    vis = benchlib._visible_ordinals()
    return n if vis is None else min(n, len(vis)) if n else len(vis)


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: this process starts the N ranks itself, the way the
    driver does (torch.distributed.run, one process per GPU), as a CHILD process -- it never touches HIP and never exec()s --,
    relays their output (rank 0's JSON line) and returns their exit code.  An N-GPU request never prints a 1-GPU line."""
    need = 1 if (args.all_on_device0 or args.stub_decode) else args.gpus
    have = visible_gpus()
    if not args.stub_decode and have < need:
        print("bench.py: --gpus %d asked for, %d GPU(s) visible here: refusing to measure fewer GPUs than requested "
              "(use --all-on-device0 --dist-backend gloo to rehearse the multi-rank path on one GPU)" % (args.gpus, have), file=sys.stderr)
        return 3
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else None
    if world_env is None and args.gpus > 1:
        return spawn_ranks(args)
    if world_env is not None and world_env != args.gpus:
        print("bench.py: launched with WORLD_SIZE=%d but --gpus %d: the two must agree" % (world_env, args.gpus), file=sys.stderr)
        return 2
    if args.stub_decode:
        return stub_main(args)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from motioncam_decoder_amd import benchlib
    # before the first GPU call: this process, and the pinned buffers it will first-touch, next to its GPU
    if args.all_on_device0:
        local = 0
    numa = benchlib.bind_to_gpu_numa(local) if world > 1 else None
    import torch
    import torch.distributed as dist_mod

    if not args.all_on_device0 and torch.cuda.device_count() < max(world, local + 1):  # (counting devices does not initialise HIP)
        print("bench.py: rank %d of %d needs cuda:%d, %d GPU(s) visible: refusing to measure fewer GPUs than requested" % (
            rank, world, local, torch.cuda.device_count()), file=sys.stderr)
        return 3
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import motioncam_decoder_amd as M
    # The library's context BEFORE this process's first torch (or RCCL) operation on the GPU, as in a C++ host that has neither: HIP
    # streams of one priority share four hardware queues, and which of the context's streams end up beside each other -- hence how
    # well the three lanes of its host-memory pipeline overlap, pcie_inclusive below: 2 690 or 2 500 UHD frames/s -- depends on the
    # queues that exist when it is created (tools/pcie_order.py, docs/lab_notes.md).  (Only where the library is built already.)
    ctx = M.Context(local) if os.path.exists(M.lib_path()) else None
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    if use_dist:
        if args.dist_backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist_mod.init_process_group(args.dist_backend)
    comm = benchlib.Comm(dist_mod, dev if args.dist_backend == "nccl" else "cpu")

    from motioncam_decoder_amd import build as B
    from motioncam_decoder_amd import shard
    if local == 0:
        if not os.path.exists(M.lib_path()):
            B.build_hip()
        B.build_synth()
    comm.barrier()
    L = synth_lib()
    if ctx is None:
        ctx = M.Context(local)
    ctx.profile(True)
    ranks_seen = int(round(comm.sum([1.0])[0]))  # every rank that takes part says so: the line's n_gpus is counted, not assumed
    if ranks_seen != args.gpus:
        print("bench.py: %d rank(s) took part, --gpus %d" % (ranks_seen, args.gpus), file=sys.stderr)
        return 2

    dists = [args.dist] + ([] if args.no_also else [("u" if args.dist == "nat" else "nat")])  # (at every N: a SCALE line is as complete as the N = 1 line)
    results = {}
    calib = box_calibration(torch, dev) if rank == 0 else None
    for d in dists:
        # weak scaling: the job is world * frames frames, frame i decoded by rank i % world
        wl = Workload(torch, M, L, dev, args, d, shard.shard_frames(world * args.frames, rank, world))
        times, kms, tiles, ok = run_timed(torch, comm, ctx, M, wl, args)
        results[d] = dict(wl=wl, times=times, kms=kms, tiles=tiles, ok=bool(comm.min([1.0 if ok else 0.0])[0] > 0.5))

    wl = results[args.dist]["wl"]
    extra = {}
    sums, digest = frame_checksums(torch, comm, wl, args.frames)
    # which GPU every rank decoded on (PCI bus of its device), gathered like the checksums
    prop = torch.cuda.get_device_properties(local)
    code = float((getattr(prop, "pci_domain_id", 0) << 16) | (getattr(prop, "pci_bus_id", 0) << 8) | getattr(prop, "pci_device_id", 0))
    codes = [int(v) for v in comm.sum([code if r == rank else 0.0 for r in range(world)])]
    devices_seen = ["%04x:%02x:%02x" % (c >> 16, (c >> 8) & 0xff, c & 0xff) for c in codes]
    if not args.no_pcie and not args.no_cpu:
        # the host-buffer legs run on EVERY rank at once: what limits a node is its PCIe links and the host
        # memory feeding them, not the HBM-resident kernels
        link = None
        try:
            link = link_probe(torch, dev)  # (no collective inside)
            extra["pcie_link"] = link
        except Exception as e:
            extra["pcie_link"] = {"error": repr(e)}
        link = link if bool(comm.min([1.0 if link else 0.0])[0] > 0.5) else None
        # (every leg is collective-safe: a rank that fails inside one leaves it together with the others)
        extra["pcie_inclusive"] = pcie_inclusive(M, L, ctx, wl, comm, link, nframes=min(240, args.frames))
        extra["pcie_inclusive_pack12"] = pcie_inclusive(M, L, ctx, wl, comm, link, nframes=min(240, args.frames), pack12=True)
        # what 10-bit footage ships: 1.25 bytes per sample (these 12-bit frames saturate at 1023 on the way, like the oracle's)
        extra["pcie_inclusive_pack10"] = pcie_inclusive(M, L, ctx, wl, comm, link, nframes=min(240, args.frames), bits=10)

    if not args.no_pcie and not args.no_cpu:
        # the product's own multi-GPU driver (mcraw_pool_*), one process over the GPUs: on one GPU as pools of one and of
        # two members, at N > 1 from rank 0 over every visible GPU while the other ranks idle at the barrier
        if rank == 0:
            try:
                if world == 1:
                    extra["pool"] = [pool_leg(torch, M, L, wl, [local], extra.get("pcie_link")),
                                     pool_leg(torch, M, L, wl, [local, local], extra.get("pcie_link"))]
                else:
                    ndev = min(torch.cuda.device_count(), world)
                    extra["pool"] = [pool_leg(torch, M, L, wl, list(range(ndev)), extra.get("pcie_link"), nframes=min(240, args.frames))]
            except Exception as e:
                extra["pool"] = {"error": repr(e)}
        comm.barrier()

    # what an N > 1 line says about every rank (each figure a rank's own, in front of the closing barriers): a sub-linear curve
    # can then be read -- one slow GPU, one NUMA node, one mapping of the tile kernel, one way home of the status words
    per_rank = None
    if world > 1:
        r0 = results[args.dist]
        tile_ms, tile_n = r0["tiles"]
        per_rank = {
            "ms_per_step": [round(v, 4) for v in comm.gather(r0["kms"]["own_ms_per_step"])],
            "k7_tiles_avg_launch_ms": [round(v, 4) for v in comm.gather(tile_ms / max(tile_n, 1))],
            "k7_side_ms": [round(v, 4) for v in comm.gather(r0["kms"]["k7_side"])],
            "xcd_runs": [int(v) for v in comm.gather(r0["kms"].get("xcd_runs") if r0["kms"].get("xcd_runs") is not None else -1)],
            "numa_node": [int(v) for v in comm.gather(numa["node"] if numa else -1)],
            "host_way": [int(v) for v in comm.gather(ctx.host_way() if ctx.host_way() is not None else -1)],
            "note": "by rank; ms_per_step: the rank's own median round, taken in front of the round's closing barrier (the line's "
                    "ms_per_step is the max over ranks per round); xcd_runs / host_way: what the rank's context measured and chose "
                    "(-1: nothing decided); pcie_inclusive*.frames_per_s_by_rank: every rank's own host-to-host rate",
        }
    if rank == 0:
        def summarize(r):
            wl = r["wl"]
            st = benchlib.round_stats(r["times"], args.steps)
            tile_ms, tile_n = r["tiles"]                    # summed over every launch since the counters' reset
            bytes_step = wl.in_bytes + wl.out_bytes
            tile_avg = tile_ms / max(tile_n, 1)
            ach = bytes_step / (tile_avg * 1e-3) / 1e9 if tile_avg > 0 else 0.0
            return {
                "mpix_s": world * wl.pixels / (st["median"] * 1e-3) / 1e6,
                "st": st,
                "tiles_ms_per_launch": tile_avg,
                "achieved_gbs": ach,
                "step_gbs": bytes_step / (st["median"] * 1e-3) / 1e9,
                "bytes_per_launch": bytes_step,
                "bpp": wl.bpp,
                "kms": r["kms"],
            }

        s = summarize(results[args.dist])
        wname = "%dx%d %d-bit type-7 x %d frames/GPU, %s" % (args.width, args.height, args.nbits, args.frames,
                                                             "Nat (smooth field + noise sigma %g)" % args.sigma if args.dist == "nat" else "U (uniform)")
        key = "%dx%d_%dbit_%d_%s" % (args.width, args.height, args.nbits, args.frames, args.dist)
        traffic, traffic_src = traffic_from_profile(key, s["bytes_per_launch"])
        prof_ms, prof_path = profile_launch_ms(args.dist) if (args.width, args.height, args.frames, args.nbits) == (3840, 2160, 240, 12) else (None, None)
        out = {
            "metric": METRIC,
            "value": round(s["mpix_s"], 1),
            "unit": "MPixels/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "devices": devices_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(s["st"]["median"], 4),
            "ms_per_step_min": round(s["st"]["min"], 4),
            "ms_per_step_max": round(s["st"]["max"], 4),
            "rounds": s["st"]["rounds"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic: %d seeded distinct frames per rank (own encoder), replicated to %d frames at distinct "
                    "HBM addresses; inputs resident in HBM before the timed region" % (len(wl.pairs), args.frames),
            "bit_exact": results[args.dist]["ok"],
            "frame_checksums": {"frames": len(sums), "digest": digest, "first": sums[:4],
                                "note": "32-bit checksums of the decoded global frames 0..%d of the job (frame g = image g mod %d, decoded by "
                                        "rank g mod n_gpus), gathered over the ranks; sha1 digest over them: the same at every --gpus N"
                                        % (len(sums) - 1, args.distinct)},
            "distinct_devices": len(set(devices_seen)),
            "bit_exact_scope": "sanity flag: sampled output frames of the timed buffers equal the images the encoder was given; "
                               "parity against the oracle / reference is what tests/ -m gpu checks",
            "frames_per_s": round(world * args.frames / (s["st"]["median"] * 1e-3), 1),
            "config": {"workload": wname, "baseline_config": args.config, "frames_per_gpu": args.frames, "width": args.width,
                       "height": args.height, "bits": args.nbits, "encoding": 7, "input_bpp": round(s["bpp"], 2),
                       "sharding": "frame index, no collective", "streams": args.streams, "numa": numa},
            "roofline": {"bound": "hbm", "kernel": "k7_tiles", "achieved": round(s["achieved_gbs"], 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(s["achieved_gbs"] / HBM_PEAK_GBS, 4),
                         # the whole step (every launch of the path) against the same peak: what a caller sees
                         "step_achieved": round(s["step_gbs"], 1), "step_frac": round(s["step_gbs"] / HBM_PEAK_GBS, 4),
                         # the same bytes over the committed rocprofv3 average of this kernel on this workload (another box, another
                         # day: boxes differ by several per cent): what a reader can reproduce from profiles/ alone
                         "frac_profile": round(s["bytes_per_launch"] / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if prof_ms else None,
                         "frac_profile_source": ("%s: k7_tiles AverageNs %.0f" % (prof_path, prof_ms * 1e6)) if prof_ms else None,
                         "traffic": round(traffic) if traffic else None,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(s["bytes_per_launch"]),
                         "avg_launch_ms": round(s["tiles_ms_per_launch"], 4),
                         "launches_per_step": 1.0, "kernel_launches_per_step": 2,
                         "xcd_runs": s["kms"].get("xcd_runs"),
                         "xcd_recheck_share": "1 launch in 64 is timed, the chosen mapping and the other one in turn: 1 in 128 runs the mapping that "
                                              "was NOT chosen (+0 .. 8 % for that launch, < 0.07 % of the step)",
                         "timed_with": "HIP events on the launch stream around every 8th k7_tiles launch of the timed rounds"},
            "kernels_ms_per_step": {"k7_tiles": round(s["tiles_ms_per_launch"], 4), "k7_side": round(s["kms"]["k7_side"], 4)},
            "kernels_ms_per_step_note": "k7_tiles: every 8th launch of the timed rounds between two event records (= roofline.avg_launch_ms); "
                                        "k7_side: every launch of six untimed steps behind them; the rest of ms_per_step is two kernel boundaries",
        }
        if world > 1:
            out["per_rank"] = per_rank
        calib_after = box_calibration(torch, dev)
        out["box_calibration"] = {"before": calib, "after": calib_after,
                                  "note": "torch fill / copy of 1 GiB in this process, before and after the timed rounds: this box's own HBM yardstick"}
        for d in dists[1:]:
            s2 = summarize(results[d])
            key2 = "%dx%d_%dbit_%d_%s" % (args.width, args.height, args.nbits, args.frames, d)
            out["also_" + d] = {"mpix_s": round(s2["mpix_s"], 1), "ms_per_step": round(s2["st"]["median"], 4),
                                "input_bpp": round(s2["bpp"], 2), "achieved_gbs": round(s2["achieved_gbs"], 1),
                                "frac": round(s2["achieved_gbs"] / HBM_PEAK_GBS, 4),
                                "step_frac": round(s2["step_gbs"] / HBM_PEAK_GBS, 4),
                                "algorithmic_bytes_per_launch": round(s2["bytes_per_launch"]), "avg_launch_ms": round(s2["tiles_ms_per_launch"], 4),
                                "traffic": (lambda tv: round(tv) if tv else None)(traffic_from_profile(key2, s2["bytes_per_launch"])[0]),
                                "kernels_ms_per_step": {"k7_tiles": round(s2["tiles_ms_per_launch"], 4), "k7_side": round(s2["kms"]["k7_side"], 4)},
                                "xcd_runs": s2["kms"].get("xcd_runs"),
                                "bit_exact": results[d]["ok"]}
            with_frac_profile(out["also_" + d], d, "k7_tiles", s2["bytes_per_launch"])
        out.update(extra)
        if not args.no_cpu: # (rank 0 alone, at every N: the other ranks wait at the barrier below)
            try:
                out["legacy"] = legacy_leg(torch, ctx, M, L, dev)
            except Exception as e:
                out["legacy"] = {"error": repr(e)}
            try:
                out["rotating_outputs"] = rotating_outputs_leg(torch, ctx, M, wl, dev)
            except Exception as e:
                out["rotating_outputs"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            if args.config != 5:
                try:
                    out["config5"] = config5_leg(torch, ctx, M, L, dev)
                except Exception as e:
                    out["config5"] = {"error": repr(e)}
                torch.cuda.empty_cache()
            try:
                out["mixed64"] = mixed64_leg(torch, ctx, M, L, dev)
            except Exception as e:
                out["mixed64"] = {"error": repr(e)}
            for key_, bits_ in (("post_stage", 12), ("post_stage10", 10), ("post_stage14", 14)):
                try:
                    out[key_] = post_stage(torch, ctx, M, L, wl, max(2, args.steps // 2), bits=bits_)
                except Exception as e:
                    out[key_] = {"error": repr(e)}
            try:
                if world > 1: # this rank was bound to its GPU's NUMA node: the baseline is the NODE's cores
                    try:
                        os.sched_setaffinity(0, range(os.cpu_count() or 1))
                    except OSError:
                        pass
                out["cpu_baseline"] = cpu_baseline(L, wl, args.cpu_seconds)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU line
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist_mod.barrier()
        dist_mod.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
