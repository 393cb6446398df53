"""Frame layouts other writers may produce (the synthetic encoder's flags, motioncam_decoder_amd/synth/mcraw_synth.c):
refs stream in front of the bits stream, unused bytes between the parts (streams on odd addresses), more side-stream
records than the frame uses, entry counts not rounded to 64, legacy trailers with many restart records.
CPU part: reference (where it was built here) = oracle = the encoder's input.  GPU part: the HIP path = oracle."""
import itertools

import numpy as np
import pytest

import _libs as L

F7 = [0, 2, 4, 8, 2 | 4, 2 | 8, 4 | 8, 2 | 4 | 8, 1 | 2 | 4 | 8]
F6 = [0, 1, 2]


def _cases():
    out = []
    for k, fl in enumerate(F7):
        w, h = ((256, 16), (320, 24), (200, 12), (448, 40))[k % 4]
        img = L.natural_image_np(w, h, 12, 12.0, 9000 + k)
        # flag 1 (unrounded count) makes the REFERENCE overflow its vectors unless the count is a multiple of 64
        nblk = ((w + 63) // 64) * ((h + 3) // 4) * 4
        if fl & 1 and nblk % 64:
            continue
        out.append((7, fl, img, L.encode7(img, None, fl)))
    for k, fl in enumerate(F6):
        w, h = ((160, 24), (75, 9), (352, 40))[k % 3]
        img = L.natural_image_np(w, h, 10, 4.0, 9100 + k)
        out.append((6, fl, img, L.encode6(img, None, fl)))
    return out


def test_variants_oracle_and_reference():
    n_ref = 0
    for typ, fl, img, buf in _cases():
        h, w = img.shape
        ret, out = (L.oracle_decode7 if typ == 7 else L.oracle_decode6)(buf, w, h)
        assert ret == w * h and np.array_equal(out[:h], img), (typ, fl)
        if L.ref() is not None and h % 4 == 0:  # (the reference writes encodedHeight rows: SURVEY 0.5b)
            rr, ro = (L.ref_decode7 if typ == 7 else L.ref_decode6)(buf, w, h)
            assert rr == w * h and np.array_equal(ro[:h], img), (typ, fl)
            n_ref += 1
    assert L.ref() is None or n_ref >= 8


def test_variant_headers_say_what_the_flags_promise():
    img = L.natural_image_np(256, 16, 12, 12.0, 9001)
    plain, swapped, gaps, extra = (L.encode7(img, None, f) for f in (0, 2, 4, 8))
    u32 = lambda b, o: int(np.frombuffer(b[o:o + 4].tobytes(), np.uint32)[0])
    assert u32(plain, 8) < u32(plain, 12) and u32(swapped, 12) < u32(swapped, 8)
    assert u32(gaps, 8) % 2 == 1 and gaps.size == plain.size + 13 + 5 + 3
    assert u32(extra, u32(extra, 8)) == u32(plain, u32(plain, 8)) + 3 * 64 and extra.size > plain.size
    l0, l2 = L.encode6(img, None, 0), L.encode6(img, None, 2)
    assert l2.size == l0.size + 5 * 2 and l2[-1] == 0xFF


@pytest.mark.gpu
def test_variants_gpu(gpu_ctx):
    from _gpu import decode_batch_device
    cases = _cases()
    items = [(typ, img.shape[1], img.shape[0], buf) for typ, fl, img, buf in cases]
    written, status, outs = decode_batch_device(gpu_ctx, items)
    for (typ, fl, img, buf), wr, st, out in zip(cases, written, status, outs):
        assert st == 0 and wr == img.size and np.array_equal(out, img), (typ, fl, hex(st))
    # the same through host-memory batches (headers read on the host) with the post stage on
    import motioncam_decoder_amd as M
    black = [16, 17, 18, 19]
    bufs, descs = [], []
    for typ, fl, img, buf in cases:
        h, w = img.shape
        rb = L.post_row_bytes(w, True)
        o = np.zeros(h * rb, np.uint8)
        bufs.append(o)
        descs.append((buf.ctypes.data, buf.size, w, h, typ, o.ctypes.data, (h * rb + 1) // 2))
    gpu_ctx.set_post(black=black, pack12=True)
    try:
        written, status = gpu_ctx.wait(gpu_ctx.decode_batch_async(M.Context.make_frames(descs)))
    finally:
        gpu_ctx.set_post()
    for (typ, fl, img, buf), o, st in zip(cases, bufs, status):
        assert st == 0 and np.array_equal(o.reshape(img.shape[0], -1), L.oracle_post(img, black, True)), (typ, fl)
