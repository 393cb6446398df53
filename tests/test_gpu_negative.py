"""GPU (-m gpu): malformed frames fail cleanly -- nonzero status, 0 written, no fault,
and well-formed neighbours in the same batch still decode."""
import numpy as np
import pytest

import _libs as L
import motioncam_decoder_amd as M

pytestmark = pytest.mark.gpu


def _u32(v):
    return np.frombuffer(np.uint32(v).tobytes(), np.uint8)


def test_malformed_type7_frames(gpu_ctx):
    from _gpu import decode_batch_device
    img = L.natural_image_np(256, 16, 12, 12.0, 3)
    good = L.encode7(img)
    bits_off = int(np.frombuffer(good[8:12].tobytes(), np.uint32)[0])
    cases = [("good", good, 0)]
    cases.append(("cut1", good[:-1], M.E_TRUNCATED))
    cases.append(("cut_half", good[: good.size // 2], None))
    cases.append(("tiny", good[:8], None))
    b = good.copy(); b[0:4] = _u32(100); cases.append(("encW%64", b, M.E_HEADER))
    b = good.copy(); b[0:4] = _u32(64); cases.append(("encW<w", b, M.E_HEADER))
    b = good.copy(); b[8:12] = _u32(1 << 30); cases.append(("bitsOff>len", b, M.E_HEADER))
    b = good.copy(); b[12:16] = _u32(1 << 30); cases.append(("refsOff>len", b, M.E_HEADER))
    b = good.copy(); b[bits_off + 4] |= 0x0F; cases.append(("bits>16", b, M.E_SIDESTREAM))
    b = good.copy(); b[bits_off:bits_off + 4] = _u32(3); cases.append(("count<N", b, M.E_SIDESTREAM))
    b = good.copy(); b[4:8] = _u32(1 << 20); cases.append(("encH huge", b, None))
    cases.append(("good2", good, 0))
    items = [(7, 256, 16, c[1]) for c in cases]
    written, status, outs = decode_batch_device(gpu_ctx, items)
    for (name, _, want), wr, st, out in zip(cases, written, status, outs):
        if want == 0:
            assert st == 0 and wr == 256 * 16 and np.array_equal(out, img), name
        else:
            assert st != 0 and wr == 0, (name, st, wr)
            if want is not None:
                assert st & want, (name, hex(st))


def test_malformed_legacy_frames(gpu_ctx):
    from _gpu import decode_batch_device
    img = L.natural_image_np(160, 24, 10, 4.0, 9)
    good = L.encode6(img)
    cases = [("good", good, True), ("no_trailing_byte", good[:-1], False), ("half", good[: good.size // 2], False),
             ("two_bytes", good[:2], False), ("good2", good, True)]
    written, status, outs = decode_batch_device(gpu_ctx, [(6, 160, 24, c[1]) for c in cases])
    for (name, _, ok), wr, st, out in zip(cases, written, status, outs):
        if ok:
            assert st == 0 and wr == 160 * 24 and np.array_equal(out, img), name
        else:
            assert st & M.E_TRUNCATED and wr == 0, (name, st)


def test_bad_arguments(gpu_ctx):
    from _gpu import decode_batch_device
    img = L.natural_image_np(128, 8, 12, 12.0, 1)
    buf = L.encode7(img)
    written, status, _ = decode_batch_device(gpu_ctx, [(5, 128, 8, buf), (7, 128, 8, buf)])
    assert status[0] & M.E_ARGS and written[0] == 0
    assert status[1] == 0 and written[1] == 128 * 8


def test_header_geometry_differs_from_rounded_size(gpu_ctx):
    # a frame coded wider/taller than ceil64(w) x ceil4(h) is re-planned from its real header
    from _gpu import decode_batch_device
    big = L.natural_image_np(320, 24, 12, 12.0, 11)
    buf = L.encode7(big)          # encW=320, encH=24
    ret, want = L.oracle_decode7(buf, 200, 16)   # caller keeps a 200x16 window
    assert ret == 200 * 16
    written, status, outs = decode_batch_device(gpu_ctx, [(7, 200, 16, buf)])
    assert status == [0] and written == [200 * 16]
    assert np.array_equal(outs[0], want) and np.array_equal(want, big[:16, :200])


def test_aliased_side_streams_are_replanned(gpu_ctx):
    # A frame whose bits stream runs past the start of its refs stream (the two chains share
    # bytes).  The reference follows each chain wherever it goes, and so does k7_side: a stream's
    # extent is bounded by the frame's length only, never by where the other stream starts.
    from _gpu import decode_batch_device
    w, h, val = 256, 16, 5
    img = np.full((h, w), val, np.uint16)
    nblk = (w // 64) * (h // 4) * 4          # 64 blocks -> one side-stream record per stream
    enc = L.encode7(img, np.full(nblk, 5, np.uint8))
    bits_off = int(np.frombuffer(enc[8:12].tobytes(), np.uint32)[0])
    nrec, m = (nblk + 63) // 64, 3
    # rebuild the tail: one long run of {hbits 0, ref 5} records read by BOTH chains
    tail = np.concatenate([_u32(nrec * 64), np.tile(np.array([0, 5], np.uint8), nrec + m + 4)])
    buf = np.concatenate([enc[:bits_off], tail]).copy()
    buf[12:16] = _u32(bits_off + 4 + 2 * m)   # refsOffset inside the bits stream
    ret, want = L.oracle_decode7(buf, w, h)
    assert ret == w * h and np.array_equal(want, img)
    if L.ref() is not None:
        rr, orr = L.ref_decode7(buf, w, h)
        assert rr == w * h and np.array_equal(orr[:h], img)
    written, status, outs = decode_batch_device(gpu_ctx, [(7, w, h, buf), (7, w, h, enc)])
    assert status == [0, 0] and written == [w * h, w * h]
    assert np.array_equal(outs[0], img) and np.array_equal(outs[1], img)


def test_abi_edge_arguments(gpu_ctx):
    """Every per-frame argument error is a status bit of that frame, not a crash and not a failed call;
    an empty batch is a no-op; NULL where an array is required fails the call."""
    import ctypes as C
    import torch
    dev = torch.device("cuda:0")
    w, h = 128, 8
    img = L.natural_image_np(w, h, 12, 12.0, 5)
    buf = L.encode7(img)
    t_in = torch.zeros(buf.size + 64, dtype=torch.uint8, device=dev)
    t_in[:buf.size].copy_(torch.from_numpy(buf))
    t_in2 = torch.zeros(buf.size + 64, dtype=torch.uint8, device=dev)
    t_in2[8:8 + buf.size].copy_(torch.from_numpy(buf))  # not 16-byte aligned
    t_out = torch.zeros(w * h * 2 + 64, dtype=torch.uint8, device=dev)
    t_out6 = torch.zeros(w * h * 2 + 64, dtype=torch.uint8, device=dev)
    ip, ip_mis, op = t_in.data_ptr(), t_in2.data_ptr() + 8, t_out.data_ptr()
    assert ip % 16 == 0
    ret6, out6 = L.oracle_decode6(buf, w, h)  # a type-7 buffer read as a legacy stream: whatever the oracle says
    cases = [
        ((ip, buf.size, w, h, 7, op, w * h), 0),
        ((ip_mis, buf.size, w, h, 7, op, w * h), 0),                 # device input at any alignment
        ((ip, buf.size, w, h, 7, op + 1, w * h), M.E_ARGS),          # odd output address
        ((ip, 0, w, h, 7, op, w * h), M.E_ARGS),                     # empty input
        ((ip, buf.size, 0, h, 7, op, w * h), M.E_ARGS),
        ((ip, buf.size, w, -3, 7, op, w * h), M.E_ARGS),
        ((ip, buf.size, 1 << 20, 1 << 12, 7, op, w * h), M.E_ARGS),  # 2^32 samples
        ((0, buf.size, w, h, 7, op, w * h), M.E_ARGS),
        ((ip, buf.size, w, h, 7, 0, w * h), M.E_ARGS),
        ((ip, buf.size, w, h, 7, op, w * h - 1), M.E_CAPACITY),
        ((ip, buf.size, w, h, 6, t_out6.data_ptr(), w * h), None),
        ((ip, buf.size, w, h, 7, op, w * h), 0),
    ]
    frames = M.Context.make_frames([c[0] for c in cases])
    written, status = gpu_ctx.decode_batch(frames)
    for i, (_, want) in enumerate(cases):
        if want == 0:
            assert status[i] == 0 and written[i] == w * h, (i, status[i])
        elif want is None:
            assert (status[i] == 0 and written[i] == ret6) if ret6 else (status[i] != 0 and written[i] == 0), (i, status[i], ret6)
            if ret6:
                assert np.array_equal(t_out6[: w * h * 2].cpu().numpy().view(np.uint16).reshape(h, w), out6)
        else:
            assert status[i] & want and written[i] == 0, (i, status[i], want)
    torch.cuda.synchronize()
    assert np.array_equal(t_out[: w * h * 2].cpu().numpy().view(np.uint16).reshape(h, w), img)
    # empty batch
    empty = M.Context.make_frames([])
    assert gpu_ctx._lib.mcraw_decode_batch(gpu_ctx._h, empty, 0, M.MEM_DEVICE, None, None, None) == 0
    # NULL frame array with n > 0, NULL context
    assert gpu_ctx._lib.mcraw_decode_batch(gpu_ctx._h, None, 3, M.MEM_DEVICE, None, None, None) != 0
    assert gpu_ctx._lib.mcraw_decode_batch(None, frames, len(cases), M.MEM_DEVICE, None, None, None) != 0
    # the context is still good
    written, status = gpu_ctx.decode_batch(M.Context.make_frames([cases[0][0]]))
    assert status == [0] and written == [w * h]


def test_replanned_frames_in_host_memory_batches(gpu_ctx):
    # the re-plan path of the host-memory pipeline (synchronous and ticketed), with and without the
    # post stage: a frame coded larger than ceil64(w) x ceil4(h), between two ordinary frames
    big = L.natural_image_np(320, 24, 12, 12.0, 11)
    odd = L.encode7(big)                       # encW=320, encH=24, decoded into a 200x16 window
    plain = L.natural_image_np(256, 16, 12, 12.0, 12)
    pbuf = L.encode7(plain)
    window = big[:16, :200]
    for black, pack12 in ((None, False), ([64, 65, 66, 67], True)):
        for use_ticket in (False, True):
            items = [(pbuf, 256, 16, plain), (odd, 200, 16, window), (pbuf, 256, 16, plain)]
            outs, descs = [], []
            for buf, w, h, img in items:
                rb = L.post_row_bytes(w, pack12)
                o = np.full(h * rb + 16, 0xA5, np.uint8)
                outs.append(o)
                descs.append((buf.ctypes.data, buf.size, w, h, 7, o.ctypes.data, (h * rb + 1) // 2))
            frames = M.Context.make_frames(descs)
            gpu_ctx.set_post(black=black, pack12=pack12)
            try:
                if use_ticket:
                    t = gpu_ctx.decode_batch_async(frames)
                    gpu_ctx.set_post()  # the ticket keeps the stage it was submitted with
                    written, status = gpu_ctx.wait(t)
                else:
                    written, status = gpu_ctx.decode_batch(frames, mem=M.MEM_HOST)
            finally:
                gpu_ctx.set_post()
            for (buf, w, h, img), o, wr, st in zip(items, outs, written, status):
                assert st == 0 and wr == w * h, (black, pack12, use_ticket, st, wr)
                rb = L.post_row_bytes(w, pack12)
                want = L.oracle_post(img, black, pack12)
                assert np.array_equal(o[: h * rb].reshape(h, rb), want), (black, pack12, use_ticket, w, h)
                assert (o[h * rb:] == 0xA5).all()


def _geometry_and_aliased_cases():
    """(name, type, w, h, buffer, expected image): a frame coded larger than the caller's window, and a frame
    whose two side streams share bytes -- both decoded by the reference (the header alone says where things are)."""
    big = L.natural_image_np(320, 24, 12, 12.0, 11)
    cases = [("window", 7, 200, 16, L.encode7(big), big[:16, :200])]
    w, h = 256, 16
    img = np.full((h, w), 5, np.uint16)
    nblk = (w // 64) * (h // 4) * 4
    enc = L.encode7(img, np.full(nblk, 5, np.uint8))
    bits_off = int(np.frombuffer(enc[8:12].tobytes(), np.uint32)[0])
    nrec, m = (nblk + 63) // 64, 3
    tail = np.concatenate([_u32(nrec * 64), np.tile(np.array([0, 5], np.uint8), nrec + m + 4)])
    buf = np.concatenate([enc[:bits_off], tail]).copy()
    buf[12:16] = _u32(bits_off + 4 + 2 * m)
    cases.append(("aliased", 7, w, h, buf, img))
    plain = L.natural_image_np(256, 16, 12, 12.0, 12)
    cases.append(("plain", 7, 256, 16, L.encode7(plain), plain))
    return cases


def test_status_less_device_batches_are_resolved_by_synchronize(gpu_ctx):
    """No status requested: frames whose header asks for more than the plan gave them are decoded by
    mcraw_ctx_synchronize (or when their slot comes round again), not left failed."""
    import torch
    dev = torch.device("cuda:0")
    cases = _geometry_and_aliased_cases()
    for rounds in (1, 7):  # 7 > the slot ring: the first batches are settled when their slots are taken again
        keep = []
        for r in range(rounds):
            ins, outs, descs = [], [], []
            for name, typ, w, h, buf, want in cases:
                ti = torch.from_numpy(buf).to(dev)
                to = torch.full((w * h * 2 + 16,), 0xA5, dtype=torch.uint8, device=dev)
                ins.append(ti)
                outs.append(to)
                descs.append((ti.data_ptr(), ti.numel(), w, h, typ, to.data_ptr(), w * h))
            torch.cuda.synchronize()
            assert gpu_ctx.decode_batch(M.Context.make_frames(descs), want_status=False) is None
            keep.append((ins, outs))
        status = gpu_ctx.synchronize(len(cases))
        assert status == [0] * len(cases), [hex(s) for s in status]
        torch.cuda.synchronize()
        for ins, outs in keep:
            for (name, typ, w, h, buf, want), to in zip(cases, outs):
                a = to.cpu().numpy()
                assert np.array_equal(a[: w * h * 2].view(np.uint16).reshape(h, w), want), (rounds, name)
                assert (a[w * h * 2:] == 0xA5).all()


def test_geometry_and_aliased_frames_through_async_host_batches(gpu_ctx):
    cases = _geometry_and_aliased_cases()
    outs, descs = [], []
    for name, typ, w, h, buf, want in cases:
        o = np.full(w * h + 8, 0xA5A5, np.uint16)
        outs.append(o)
        descs.append((buf.ctypes.data, buf.size, w, h, typ, o.ctypes.data, w * h))
    frames = M.Context.make_frames(descs)
    t1 = gpu_ctx.decode_batch_async(frames)
    written, status = gpu_ctx.wait(t1)
    assert status == [0] * len(cases) and written == [c[2] * c[3] for c in cases]
    for (name, typ, w, h, buf, want), o in zip(cases, outs):
        assert np.array_equal(o[: w * h].reshape(h, w), want) and (o[w * h:] == 0xA5A5).all(), name


def test_legacy_stream_with_more_than_2_24_records_does_not_wrap_into_the_frame(gpu_ctx):
    """A small legacy frame followed by 40 MB of two-byte records (zeros): the chunk entries carry record indices in
    24 bits; indices past that saturate instead of wrapping, so nothing behind the frame's own records is decoded
    over its pixels."""
    import torch
    dev = torch.device("cuda:0")
    w, h = 256, 64
    img = L.natural_image_np(w, h, 10, 4.0, 77)
    enc = L.encode6(img)
    buf = np.concatenate([enc, np.zeros(40 << 20, np.uint8)])
    ret, want = L.oracle_decode6(buf, w, h)
    assert ret == w * h and np.array_equal(want, img)
    from _gpu import decode_batch_device
    written, status, outs = decode_batch_device(gpu_ctx, [(6, w, h, buf)])
    assert status == [0] and written == [w * h]
    assert np.array_equal(outs[0], img)


def test_host_batches_do_not_trust_a_header_for_their_workspace(gpu_ctx):
    """Host-memory batches plan every frame from its own 16-byte header (RawData.cpp:500-524), which is untrusted input.
    A header that claims more blocks than the buffer could describe gets no workspace for them, a frame that claims a
    huge (but possible) geometry does not multiply its workspace by the frames around it, and in both cases only that
    frame fails: its neighbours decode, the call succeeds."""
    plain = L.natural_image_np(256, 16, 12, 12.0, 21)
    pbuf = L.encode7(plain)
    # (a) impossible: 32768 x 32764 coded pixels claimed by a 2 KB buffer
    liar = pbuf.copy()
    liar[0:4] = _u32(32768)
    liar[4:8] = _u32(32764)
    # (b) possible on paper: the same claim in a buffer long enough for that many (empty) side-stream records
    nrec = (32768 * 32764 // 64 + 63) // 64
    big = np.zeros(16 + 2 * (4 + 2 * nrec) + 64, np.uint8)
    big[: pbuf.size] = pbuf
    big[0:4] = _u32(32768)
    big[4:8] = _u32(32764)
    for bogus in (liar, big):
        items = [pbuf] * 40 + [bogus] + [pbuf] * 40
        outs, descs = [], []
        for buf in items:
            o = np.full(256 * 16 + 8, 0xA5A5, np.uint16)
            outs.append(o)
            descs.append((buf.ctypes.data, buf.size, 256, 16, 7, o.ctypes.data, 256 * 16))
        written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs), mem=M.MEM_HOST)
        for i, (o, wr, st) in enumerate(zip(outs, written, status)):
            if i == 40:
                assert st != 0 and wr == 0, (bogus.size, hex(st), wr)
            else:
                assert st == 0 and wr == 256 * 16, (bogus.size, i, hex(st))
                assert np.array_equal(o[: 256 * 16].reshape(16, 256), plain) and (o[256 * 16:] == 0xA5A5).all()
