"""GPU (-m gpu): size-independent properties at BASELINE's full batch size (config 3: 240 frames of
3840x2160 12-bit): every frame round-trips, the result does not depend on the order or the
neighbours of a frame in the batch, and decoding twice gives the same bytes (checksum of checksums)."""
import numpy as np
import pytest

import _libs as L

pytestmark = pytest.mark.gpu

W, H, N, DISTINCT = 3840, 2160, 240, 12


def _sums(torch, t_out, n):
    v = t_out.view(torch.int32).view(n, -1)
    # two independent folds per frame: plain sum and position-weighted sum (wrapping int64)
    idx = torch.arange(v.shape[1], device=v.device, dtype=torch.int64) % 65521 + 1
    return (v.to(torch.int64).sum(1).cpu().numpy(), (v.to(torch.int64) * idx).sum(1).cpu().numpy())


def test_full_batch_order_and_repeat_invariance(gpu_ctx):
    import torch
    import motioncam_decoder_amd as M
    dev = torch.device("cuda:0")
    imgs = [L.synth_image(W, H, 12, i % 2, 12.0, 3000 + i) for i in range(DISTINCT)]  # U and Nat alternate
    bufs = [L.encode7(im) for im in imgs]
    t_bufs = [torch.from_numpy(b).to(dev) for b in bufs]
    t_out = torch.zeros(N * W * H * 2, dtype=torch.uint8, device=dev)

    def run(order):
        t_out.zero_()
        descs = [(t_bufs[k].data_ptr(), t_bufs[k].numel(), W, H, 7, t_out.data_ptr() + i * W * H * 2, W * H)
                 for i, k in enumerate(order)]
        torch.cuda.synchronize()
        written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs))
        assert status == [0] * N and written == [W * H] * N
        return _sums(torch, t_out, N)

    rng = np.random.default_rng(3)
    order_a = [i % DISTINCT for i in range(N)]
    order_b = list(rng.permutation(order_a))
    a1 = run(order_a)
    a2 = run(order_a)
    b = run(order_b)
    assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1])       # deterministic
    # the expected folds of each distinct frame, from the source images (round trip)
    ref = {}
    for k, im in enumerate(imgs):
        v = torch.from_numpy(im.view(np.int32).reshape(1, -1).copy()).to(dev).view(torch.uint8)
        s = _sums(torch, v, 1)
        ref[k] = (int(s[0][0]), int(s[1][0]))
    for i, k in enumerate(order_a):
        assert (int(a1[0][i]), int(a1[1][i])) == ref[k], (i, k)
    for i, k in enumerate(order_b):                                             # order / neighbours do not matter
        assert (int(b[0][i]), int(b[1][i])) == ref[int(k)], (i, k)


def test_frame_checksums_do_not_depend_on_the_gpu_count(gpu_ctx):
    """BASELINE config 5: 7680x4320 frames sharded over 1, 2, 4, 8 ranks (frame i -> rank i mod world,
    motioncam_decoder_amd/shard.py); the per-frame checksums of the job must be the same for every world
    size.  One GPU plays the ranks one after another: what this proves is the PARTITION RULE (every frame decoded exactly
    once, by the rank the rule names, to the same pixels whatever the world size) -- not that eight physical devices give the
    same bytes; no test on a one-GPU box can.  `bench.py --gpus N` carries the check to real devices: its line holds a digest of
    the job's per-frame checksums that must equal the N = 1 line's (`frame_checksums`; compared for two ranks on cuda:0 in
    test_bench_multi_rank_path_on_one_gpu below)."""
    import torch
    import motioncam_decoder_amd as M
    from motioncam_decoder_amd import shard
    dev = torch.device("cuda:0")
    w, h, total, distinct = 7680, 4320, 16, 4
    imgs = [L.synth_image(w, h, 12, 1, 12.0, 5000 + i) for i in range(distinct)]
    bufs = [torch.from_numpy(L.encode7(im)).to(dev) for im in imgs]
    want = [shard.frame_checksum(imgs[i % distinct]) for i in range(total)]
    for world in (1, 2, 4, 8):
        got = {}
        for rank in range(world):
            mine = shard.shard_frames(total, rank, world)
            t_out = torch.zeros(len(mine) * w * h * 2, dtype=torch.uint8, device=dev)
            descs = [(bufs[g % distinct].data_ptr(), bufs[g % distinct].numel(), w, h, 7,
                      t_out.data_ptr() + k * w * h * 2, w * h) for k, g in enumerate(mine)]
            written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs))
            assert status == [0] * len(mine) and written == [w * h] * len(mine)
            out = t_out.cpu().numpy().view(np.uint16).reshape(len(mine), h, w)
            for k, g in enumerate(mine):
                got[g] = shard.frame_checksum(out[k])
            del t_out
        assert [got[i] for i in range(total)] == want, world


def test_config5_at_its_stated_size_120_frames_of_8k_in_one_batch(gpu_ctx):
    """BASELINE config 5, one rank's share: 120 frames of 7680x4320 12-bit in ONE batch (8 GB of output).  Every
    frame is compared on the device with the image the encoder was given; sampled frames against the oracle."""
    import torch
    import motioncam_decoder_amd as M
    dev = torch.device("cuda:0")
    w, h, n, distinct = 7680, 4320, 120, 4
    imgs = [L.synth_image(w, h, 12, 1, 12.0, 5000 + i) for i in range(distinct)]
    bufs = [L.encode7(im) for im in imgs]
    for k in (0, 3):
        ret, out = L.oracle_decode7(bufs[k], w, h)
        assert ret == w * h and np.array_equal(out, imgs[k])
    t_exp = [torch.from_numpy(im.view(np.int16)).to(dev) for im in imgs]
    t_in = [torch.from_numpy(bufs[i % distinct]).to(dev) for i in range(n)]  # every frame at its own address
    t_out = torch.zeros(n * w * h, dtype=torch.int16, device=dev)
    descs = [(t_in[i].data_ptr(), t_in[i].numel(), w, h, 7, t_out.data_ptr() + 2 * i * w * h, w * h) for i in range(n)]
    written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs))
    assert status == [0] * n and written == [w * h] * n
    out = t_out.view(n, h, w)
    for i in range(n):
        assert torch.equal(out[i], t_exp[i % distinct]), i
    # the same batch again without statuses (the path bench.py --config 5 times), then the statuses
    t_out.zero_()
    assert gpu_ctx.decode_batch(M.Context.make_frames(descs), want_status=False) is None
    assert gpu_ctx.synchronize(n) == [0] * n
    torch.cuda.synchronize()
    assert all(torch.equal(out[i], t_exp[i % distinct]) for i in (0, 59, 119))


def test_config3_host_memory_mode_at_240_frames_every_frame_compared(gpu_ctx):
    """BASELINE config 3 as stated: 240 UHD 12-bit frames, pinned host buffers in and out, H2D / decode / D2H
    overlapped on the library's streams -- all 240 outputs compared."""
    import ctypes as C
    import motioncam_decoder_amd as M
    lib = M.load()
    w, h, n, distinct = 3840, 2160, 240, 6
    imgs = [L.synth_image(w, h, 12, i % 2, 12.0, 3000 + i) for i in range(distinct)]  # Nat and U alternate
    bufs = [L.encode7(im) for im in imgs]
    ins, outs = [], []
    try:
        descs = []
        for i in range(n):
            b = bufs[i % distinct]
            pi, po = lib.mcraw_host_alloc(b.size), lib.mcraw_host_alloc(w * h * 2)
            assert pi and po
            ins.append(pi)
            outs.append(po)
            C.memmove(pi, b.ctypes.data, b.size)
            descs.append((pi, b.size, w, h, 7, po, w * h))
        written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs), mem=M.MEM_HOST)
        assert status == [0] * n and written == [w * h] * n
        for i in range(n):
            got = np.ctypeslib.as_array(C.cast(outs[i], C.POINTER(C.c_uint16)), shape=(h, w))
            assert np.array_equal(got, imgs[i % distinct]), i
    finally:
        for p in ins + outs:
            lib.mcraw_host_free(p)


def test_bench_multi_rank_path_on_one_gpu(tmp_path):
    """The N > 1 path of bench.py on hardware, as far as a one-GPU box allows: two ranks launched the way the driver
    launches them (torch.distributed.run), both decoding on cuda:0, reductions over gloo instead of RCCL.  Everything
    else is the real thing: NUMA binding, sharded workloads, timed rounds between barriers, the host-buffer (PCIe)
    legs on every rank at once, the rank-0 JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # `bench.py --gpus 2` starts its two ranks itself (a child process: torch.distributed.run); the line says how many took part
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", "24", "--distinct", "4", "--min-seconds", "0.05", "--dist-backend", "gloo", "--all-on-device0",
           "--cpu-seconds", "0.2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["bit_exact"] is True and d["scaling"] == "weak"
    assert d["distinct_devices"] == 1 and len(d["devices"]) == 2  # (both ranks on cuda:0 here, and the line says so)
    assert d["config"]["frames_per_gpu"] == 24
    assert abs(d["value"] / (2 * 24 * 3840 * 2160 / (d["ms_per_step"] * 1e-3) / 1e6) - 1.0) < 2e-3  # (ms_per_step is rounded)
    assert d["roofline"]["step_frac"] > 0 and d["roofline"]["frac"] > 0
    # the job's first 24 frames decode to the same pixels whether one rank holds them all or two share them
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--frames", "24",
                         "--distinct", "4", "--min-seconds", "0.02", "--no-pcie", "--no-cpu", "--no-also"], capture_output=True, text=True,
                        timeout=600, cwd=root, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["ranks_seen"] == 1
    assert d1["frame_checksums"]["frames"] == 24 and d1["frame_checksums"] == d["frame_checksums"]
    p = d["pcie_inclusive"]
    assert p["bit_exact"] is True and p["frames_per_rank"] == 24 and abs(p["frames_per_s"] - 2 * p["frames_per_s_per_rank"]) < 1.0


def test_xcd_mapping_is_measured_per_buffer_set_and_kept(gpu_ctx):
    """Large resident batches: the library times two mappings of the tile kernel's workgroups on the first launches on a
    new set of buffers and keeps the faster (mcraw_ctx_xcd_runs).  A caller that alternates between two sets of frame
    buffers (double buffering) gets a choice for each, once; every launch on the way is bit-exact whatever it ran with."""
    import torch
    import motioncam_decoder_amd as M
    dev = torch.device("cuda:0")
    imgs = [L.natural_image_np(256, 32, 12, 12.0, 7700 + i) for i in range(40)]
    bufs = [L.encode7(im) for im in imgs]
    tin = [torch.from_numpy(b).to(dev) for b in bufs]
    sets = []
    for k in range(2):
        touts = [torch.zeros(im.size * 2, dtype=torch.uint8, device=dev) for im in imgs]
        frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), 256, 32, 7, touts[i].data_ptr(), imgs[i].size)
                                        for i in range(len(imgs))])
        sets.append((touts, frames))
    seen = [set(), set()]
    for rnd in range(16):
        k = rnd & 1
        touts, frames = sets[k]
        for t in touts:
            t.zero_()
        written, status = gpu_ctx.decode_batch(frames)
        assert all(s == 0 for s in status)
        for im, t in zip(imgs, touts):
            assert np.array_equal(t.cpu().numpy().view(np.uint16).reshape(im.shape), im), rnd
        seen[k].add(gpu_ctx.xcd_runs())
    last = [None, None]
    for rnd in range(4):  # the choices are made by now and stay
        k = rnd & 1
        gpu_ctx.decode_batch(sets[k][1])
        r = gpu_ctx.xcd_runs()
        assert r in (0, 128), r
        assert last[k] in (None, r)
        last[k] = r
