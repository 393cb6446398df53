"""GPU (-m gpu): the drop-in five-argument entry points (host pointers, same
signature and return convention as motioncam::raw::Decode / DecodeLegacy,
RawData.hpp:25-37) and the host-memory batch with overlapped copies."""
import ctypes as C

import numpy as np
import pytest

import _libs as L
import motioncam_decoder_amd as M

pytestmark = pytest.mark.gpu


def test_decode7_and_decode6_host_pointers(gpu_ctx, golden):
    lib = M.load()
    for name in sorted(golden):
        c = golden[name]
        out = np.full((c["h"], c["w"]), 0xA5A5, np.uint16)
        buf = np.ascontiguousarray(c["buf"])
        fn = lib.mcraw_decode7 if c["type"] == 7 else lib.mcraw_decode6
        ret = fn(out.ctypes.data, c["w"], c["h"], buf.ctypes.data, buf.size)
        assert ret == c["ret"], name
        assert np.array_equal(out, c["out"]), name


def test_decode7_failure_returns_zero(gpu_ctx):
    lib = M.load()
    img = L.natural_image_np(128, 8, 12, 12.0, 3)
    buf = L.encode7(img)[:-9].copy()
    out = np.zeros((8, 128), np.uint16)
    assert lib.mcraw_decode7(out.ctypes.data, 128, 8, buf.ctypes.data, buf.size) == 0


def test_host_memory_batch_many_sub_batches(gpu_ctx):
    # enough frames that the host pipeline cycles through all its slots (config 3 shape, reduced count)
    w, h, n = 3840, 2160, 24
    imgs = [L.synth_image(w, h, 12, 1, 12.0, 3000 + i) for i in range(4)]
    bufs = [L.encode7(im) for im in imgs]
    outs = [np.zeros((h, w), np.uint16) for _ in range(n)]
    descs = [(bufs[i % 4].ctypes.data, bufs[i % 4].size, w, h, 7, outs[i].ctypes.data, w * h) for i in range(n)]
    frames = M.Context.make_frames(descs)
    written, status = gpu_ctx.decode_batch(frames, mem=M.MEM_HOST)
    assert status == [0] * n and written == [w * h] * n
    for i in range(n):
        assert np.array_equal(outs[i], imgs[i % 4]), i


def test_concurrent_host_threads(gpu_ctx, golden):
    # SURVEY 8b "threading": the ABI is callable from several host threads at once -- each with its
    # own context (one per GPU/stream in production), and several sharing the default context of the
    # five-argument entry points.
    import threading
    lib = M.load()
    names = sorted(golden)
    errors = []

    def own_context(tid):
        try:
            ctx = M.Context(0)
            import torch
            dev = torch.device("cuda:0")
            for rep in range(3):
                for n in names[tid::4]:
                    c = golden[n]
                    ti = torch.from_numpy(np.ascontiguousarray(c["buf"])).to(dev)
                    to = torch.zeros(c["w"] * c["h"] * 2, dtype=torch.uint8, device=dev)
                    torch.cuda.synchronize()
                    fr = M.Context.make_frames([(ti.data_ptr(), ti.numel(), c["w"], c["h"], c["type"], to.data_ptr(), c["w"] * c["h"])])
                    written, status = ctx.decode_batch(fr)
                    got = to.cpu().numpy().view(np.uint16).reshape(c["h"], c["w"])
                    if status != [0] or written != [c["ret"]] or not np.array_equal(got, c["out"]):
                        errors.append(("ctx", tid, n))
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append(("ctx", tid, repr(e)))

    def default_context(tid):
        try:
            for rep in range(3):
                for n in names[tid::4]:
                    c = golden[n]
                    out = np.zeros((c["h"], c["w"]), np.uint16)
                    buf = np.ascontiguousarray(c["buf"])
                    fn = lib.mcraw_decode7 if c["type"] == 7 else lib.mcraw_decode6
                    ret = fn(out.ctypes.data, c["w"], c["h"], buf.ctypes.data, buf.size)
                    if ret != c["ret"] or not np.array_equal(out, c["out"]):
                        errors.append(("default", tid, n))
        except Exception as e:  # pragma: no cover
            errors.append(("default", tid, repr(e)))

    threads = [threading.Thread(target=own_context, args=(t,)) for t in range(4)]
    threads += [threading.Thread(target=default_context, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[:5]
