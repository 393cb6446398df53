"""GPU (-m gpu): the device pool (several GPUs behind one handle, frame i -> member i mod G).  The box has ONE
GPU: pools of 2 and 3 members are built from repeated device indices, which runs the whole sharded path -- one
context, one host thread and one set of staging slices per member -- and must give exactly the results of G = 1."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _libs as L
import motioncam_decoder_amd as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPORT = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "mcraw_export")


def _batch(n):
    items = []
    for i in range(n):
        w, h = ((256, 32), (320, 48), (192, 16))[i % 3]
        img = L.natural_image_np(w, h, 12, 12.0, 7000 + i)
        typ = 6 if i % 4 == 3 else 7
        items.append((typ, img, L.encode7(img) if typ == 7 else L.encode6(img)))
    return items


@pytest.mark.parametrize("devices", [None, [0], [0, 0], [0, 0, 0]])
def test_pool_results_do_not_depend_on_its_size(devices):
    pool = M.Pool(devices)
    try:
        assert pool.size == (len(devices) if devices else 1) and all(d == 0 for d in pool.devices())
        items = _batch(11)
        outs = [np.full(it[1].size + 8, 0xA5A5, np.uint16) for it in items]
        descs = [(it[2].ctypes.data, it[2].size, it[1].shape[1], it[1].shape[0], it[0], o.ctypes.data, it[1].size)
                 for it, o in zip(items, outs)]
        frames = M.Context.make_frames(descs)
        for use_ticket in (False, True):
            for o in outs:
                o[:] = 0xA5A5
            written, status = pool.wait(pool.decode_batch_async(frames)) if use_ticket else pool.decode_batch(frames)
            assert status == [0] * len(items) and written == [it[1].size for it in items]
            for it, o in zip(items, outs):
                assert np.array_equal(o[: it[1].size].reshape(it[1].shape), it[1]) and (o[it[1].size:] == 0xA5A5).all()
        # a broken frame is its own business, whichever member gets it
        bad = items[4][2][: items[4][2].size // 2].copy()
        descs[4] = (bad.ctypes.data, bad.size) + descs[4][2:]
        written, status = pool.decode_batch(M.Context.make_frames(descs))
        assert status[4] != 0 and written[4] == 0
        assert all(s == 0 for i, s in enumerate(status) if i != 4)
        # pinned memory from a member's own thread
        p = pool.host_alloc(pool.size - 1, 1 << 20)
        assert p
        C.memset(p, 0x5A, 1 << 20)
        M.load().mcraw_host_free(p)
    finally:
        pool.close()


def test_pool_with_post_stage():
    pool = M.Pool([0, 0])
    try:
        items = [it for it in _batch(6) if it[0] == 7]
        black = [60, 61, 62, 63]
        outs, descs = [], []
        for typ, img, buf in items:
            h, w = img.shape
            rb = L.post_row_bytes(w, True)
            o = np.zeros(h * rb, np.uint8)
            outs.append(o)
            descs.append((buf.ctypes.data, buf.size, w, h, typ, o.ctypes.data, (h * rb + 1) // 2))
        pool.set_post(black=black, pack12=True)
        written, status = pool.decode_batch(M.Context.make_frames(descs))
        pool.set_post()
        assert status == [0] * len(items)
        for (typ, img, buf), o in zip(items, outs):
            assert np.array_equal(o.reshape(img.shape[0], -1), L.oracle_post(img, black, True))
    finally:
        pool.close()


@pytest.fixture(scope="module")
def clip(tmp_path_factory):
    d = tmp_path_factory.mktemp("poolclip")
    frames, images = [], {}
    for i in range(9):
        w, h = ((640, 480), (1920, 1080), (800, 600))[i % 3]
        img = L.natural_image_np(w, h, 12, 12.0, 7100 + i)
        typ = 6 if i % 3 == 2 else 7
        ts = 1000 * (i + 1)
        frames.append((ts, typ, w, h, L.encode7(img) if typ == 7 else L.encode6(img)))
        images[ts] = img
    return L.write_mcraw(str(d / "clip.mcraw"), frames, []), images


@pytest.mark.parametrize("devs", ["0", "0,0", "0,0,0"])
@pytest.mark.parametrize("pinned", [False, True])
def test_facade_over_pools_of_several_members(clip, tmp_path, devs, pinned):
    path, images = clip
    if not os.path.exists(EXPORT):
        from motioncam_decoder_amd import build
        build.build_host()
    env = dict(os.environ, MCRAW_DEVICES=devs)
    r = subprocess.run([EXPORT, path, "-o", str(tmp_path)] + (["--pinned"] if pinned else []), capture_output=True, text=True,
                       timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    for i, ts in enumerate(sorted(images)):
        got = np.fromfile(str(tmp_path / ("frame_%06d.u16" % i)), dtype=np.uint16)
        assert np.array_equal(got.reshape(images[ts].shape), images[ts]), (devs, pinned, i)


def test_failing_chunk_with_later_chunks_in_flight(tmp_path):
    """loadFramesInto, several chunks, a corrupt frame in the FIRST one: the chunk queued behind it is still
    waited for before the exception leaves (its ticket is owned before the first is waited for), the tool
    reports the failure and exits -- no hang, no crash."""
    frames = []
    for i in range(12):  # 12 MP frames: about five per 192 MB staging chunk
        img = L.synth_image(4032, 3024, 12, 1, 12.0, 7200 + i)
        buf = L.encode7(img)
        if i == 1:
            buf = buf[: buf.size // 2].copy()
        frames.append((1000 * (i + 1), 7, 4032, 3024, buf))
    path = L.write_mcraw(str(tmp_path / "bad.mcraw"), frames, [])
    for devs in ("0", "0,0"):
        r = subprocess.run([EXPORT, path, "-o", str(tmp_path), "--pinned", "--no-write"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, MCRAW_DEVICES=devs), cwd=str(tmp_path))
        assert r.returncode != 0
        assert "Failed to uncompress frame" in (r.stderr + r.stdout)


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_pool_decodes_batches_that_are_resident_in_hbm(devices):
    """BASELINE config 5's form: frame i lives in the HBM of the GPU that decodes it, devices[i % G]
    (mcraw_pool_decode_batch_device); results as from host buffers, whatever the pool size."""
    import torch
    pool = M.Pool(devices)
    try:
        items = _batch(13)
        devs = pool.devices()
        ins, outs, descs = [], [], []
        for i, (typ, img, buf) in enumerate(items):
            dev = torch.device("cuda", devs[i % pool.size])
            ti = torch.from_numpy(buf).to(dev)
            to = torch.full((img.size * 2 + 16,), 0xA5, dtype=torch.uint8, device=dev)
            ins.append(ti)
            outs.append(to)
            descs.append((ti.data_ptr(), ti.numel(), img.shape[1], img.shape[0], typ, to.data_ptr(), img.size))
        torch.cuda.synchronize()
        frames = M.Context.make_frames(descs)
        written, status = pool.decode_batch_device(frames)
        assert status == [0] * len(items) and written == [it[1].size for it in items]
        for (typ, img, buf), to in zip(items, outs):
            a = to.cpu().numpy()
            assert np.array_equal(a[: img.size * 2].view(np.uint16).reshape(img.shape), img) and (a[img.size * 2:] == 0xA5).all()
        # queued form: nobody asks for statuses, the members' shares run back to back; one synchronize behind three batches
        for to in outs:
            to.zero_()
        for _ in range(3):
            assert pool.decode_batch_device(frames, want_status=False) is None
        assert pool.synchronize(len(items)) == [0] * len(items)
        for (typ, img, buf), to in zip(items, outs):
            assert np.array_equal(to.cpu().numpy()[: img.size * 2].view(np.uint16).reshape(img.shape), img)
    finally:
        pool.close()


def test_pool_serves_two_host_threads_at_once():
    """One host thread allocates and frees staging memory on the members' threads while another keeps batches in
    flight (synchronous and ticketed): every task runs, every batch is right, nothing deadlocks."""
    import threading
    pool = M.Pool([0, 0])
    lib = M.load()
    items = _batch(9)
    errors, stop = [], threading.Event()

    def alloc_loop():
        try:
            k = 0
            while not stop.is_set():
                p = pool.host_alloc(k % pool.size, 1 << 16)
                assert p
                C.memset(p, k & 255, 1 << 16)
                lib.mcraw_host_free(p)
                k += 1
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = threading.Thread(target=alloc_loop)
    th.start()
    try:
        for rnd in range(12):
            outs = [np.full(it[1].size + 8, 0xA5A5, np.uint16) for it in items]
            descs = [(it[2].ctypes.data, it[2].size, it[1].shape[1], it[1].shape[0], it[0], o.ctypes.data, it[1].size)
                     for it, o in zip(items, outs)]
            frames = M.Context.make_frames(descs)
            written, status = pool.wait(pool.decode_batch_async(frames)) if rnd % 2 else pool.decode_batch(frames)
            assert status == [0] * len(items) and written == [it[1].size for it in items]
            for it, o in zip(items, outs):
                assert np.array_equal(o[: it[1].size].reshape(it[1].shape), it[1])
    finally:
        stop.set()
        th.join(timeout=60)
        alive = th.is_alive()
        pool.close()
    assert not alive and not errors, errors


def test_pool_rejects_an_unknown_post_stage():
    pool = M.Pool([0])
    try:
        p = M.Post()
        p.flags = 0x40
        assert M.load().mcraw_pool_set_post(pool._h, C.byref(p)) != 0
        with pytest.raises(M.McrawError):
            pool.set_post(bits=11)
    except KeyError:
        pass  # (the wrapper itself refuses a strip width the ABI does not know)
    finally:
        pool.close()


def test_resident_batches_whose_buffers_are_not_where_the_rule_says():
    """mcraw_pool_decode_batch_device CHECKS where a frame's buffers live: a host pointer (pinned or pageable) is no
    device memory of the member that decodes the frame -> MCRAW_E_ARGS for that frame, nothing written, the other frames
    decode; the same through the queued form, where the verdict comes from mcraw_pool_synchronize."""
    import torch
    pool = M.Pool([0, 0])
    lib = M.load()
    try:
        items = _batch(8)
        dev = torch.device("cuda", 0)
        ins = [torch.from_numpy(it[2]).to(dev) for it in items]
        outs = [torch.full((it[1].size * 2,), 0xA5, dtype=torch.uint8, device=dev) for it in items]
        host_in = lib.mcraw_host_alloc(items[2][2].size)            # pinned host memory: GPU-visible, but no HBM
        C.memmove(host_in, items[2][2].ctypes.data, items[2][2].size)
        pageable_out = np.zeros(items[5][1].size, np.uint16)         # not known to HIP at all
        descs = []
        for i, it in enumerate(items):
            pin = host_in if i == 2 else ins[i].data_ptr()
            pout = pageable_out.ctypes.data if i == 5 else outs[i].data_ptr()
            descs.append((pin, it[2].size, it[1].shape[1], it[1].shape[0], it[0], pout, it[1].size))
        torch.cuda.synchronize()
        frames = M.Context.make_frames(descs)
        written, status = pool.decode_batch_device(frames)
        want = [M.E_ARGS if i in (2, 5) else 0 for i in range(8)]
        assert status == want, status
        assert [w for i, w in enumerate(written) if i in (2, 5)] == [0, 0]
        for i, it in enumerate(items):
            a = outs[i].cpu().numpy()
            if i == 2:
                assert (a == 0xA5).all()
            elif i != 5:
                assert np.array_equal(a.view(np.uint16).reshape(it[1].shape), it[1])
        assert not pageable_out.any()
        # queued
        assert pool.decode_batch_device(frames, want_status=False) is None
        assert pool.synchronize(8) == want and pool.errors == M.E_ARGS
        assert pool.synchronize(8) == want and pool.errors == M.E_ARGS  # (asked again: the same batch, the host's verdicts with it)
        lib.mcraw_host_free(host_in)
    finally:
        pool.close()


def test_queued_resident_batches_of_two_host_threads_are_reported_to_their_own_thread():
    """Two host threads queue status-less resident batches on one pool at the same time -- one of them with a frame that
    fails (a stream cut short) --: mcraw_pool_synchronize gives every thread the statuses of ITS last batch (round 3 kept one
    index map per pool: the threads got each other's), and a failure in an EARLIER queued batch is not lost: it is in the OR
    that synchronize returns."""
    import threading
    import torch
    pool = M.Pool([0, 0, 0])
    dev = torch.device("cuda", 0)
    items = _batch(12)
    results, errors = {}, []
    barrier = threading.Barrier(2)

    def worker(name, nframes, break_at, rounds_bad, rounds_good):
        try:
            its = items[:nframes]
            ins = []
            for i, it in enumerate(its):
                buf = it[2]
                if i == break_at:
                    buf = buf[: max(8, buf.size // 3)].copy()
                ins.append(torch.from_numpy(np.ascontiguousarray(buf)).to(dev))
            outs = [torch.zeros(it[1].size * 2, dtype=torch.uint8, device=dev) for it in its]
            good_ins = [torch.from_numpy(it[2]).to(dev) for it in its]
            torch.cuda.synchronize()
            bad = M.Context.make_frames([(ins[i].data_ptr(), ins[i].numel(), it[1].shape[1], it[1].shape[0], it[0], outs[i].data_ptr(), it[1].size)
                                         for i, it in enumerate(its)])
            good = M.Context.make_frames([(good_ins[i].data_ptr(), good_ins[i].numel(), it[1].shape[1], it[1].shape[0], it[0], outs[i].data_ptr(), it[1].size)
                                          for i, it in enumerate(its)])
            barrier.wait(timeout=60)
            for _ in range(rounds_bad):
                pool.decode_batch_device(bad, want_status=False)
            for _ in range(rounds_good):
                pool.decode_batch_device(good, want_status=False)
            barrier.wait(timeout=60)
            st = pool.synchronize(nframes)
            results[name] = (st, pool.errors)
        except Exception as e:  # pragma: no cover
            errors.append((name, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    ta = threading.Thread(target=worker, args=("a", 12, 4, 3, 0))   # its LAST batch holds the broken frame 4
    tb = threading.Thread(target=worker, args=("b", 7, 2, 1, 2))    # an EARLIER batch held a broken frame, the last one is clean
    ta.start(); tb.start(); ta.join(120); tb.join(120)
    pool.close()
    assert not errors and not ta.is_alive() and not tb.is_alive(), errors
    sa, ea = results["a"]
    sb, eb = results["b"]
    assert len(sa) == 12 and sa[4] != 0 and all(s == 0 for i, s in enumerate(sa) if i != 4), sa
    assert len(sb) == 7 and all(s == 0 for s in sb), sb
    assert (ea | eb) != 0  # the earlier failure of thread b (and thread a's) are in the OR of whoever synchronised first


def test_context_reports_batches_that_were_queued_without_a_status_request():
    """mcraw_ctx_last_serial / mcraw_ctx_batch_status / mcraw_ctx_errors: three status-less device batches in a row, the
    middle one with a broken frame; every batch's own statuses can still be had, and the OR says that something failed."""
    import torch
    dev = torch.device("cuda", 0)
    ctx = M.Context(0)
    items = _batch(6)
    outs = [torch.zeros(it[1].size * 2, dtype=torch.uint8, device=dev) for it in items]
    ins = [torch.from_numpy(it[2]).to(dev) for it in items]
    cut = torch.from_numpy(items[3][2][: items[3][2].size // 2].copy()).to(dev)
    torch.cuda.synchronize()

    def frames(broken):
        return M.Context.make_frames([((cut if broken and i == 3 else ins[i]).data_ptr(), (cut if broken and i == 3 else ins[i]).numel(),
                                       it[1].shape[1], it[1].shape[0], it[0], outs[i].data_ptr(), it[1].size) for i, it in enumerate(items)])
    serials = []
    for broken in (False, True, False):
        ctx.decode_batch(frames(broken), want_status=False)
        serials.append(ctx.last_serial())
    assert serials[0] < serials[1] < serials[2]
    st = [ctx.batch_status(s, 6) for s in serials]
    assert st[0] == [0] * 6 and st[2] == [0] * 6 and st[1][3] != 0 and all(v == 0 for i, v in enumerate(st[1]) if i != 3), st
    assert ctx.errors() == st[1][3] and ctx.errors() == 0
    assert ctx.batch_status(serials[2] + 1000, 6) is None
    ctx.close()
