"""Fused post-decode stage (mcraw_ctx_set_post) through the C ABI: decode + black levels / 12-bit
strip rows on the GPU == oracle decode followed by the oracle's post stage, byte for byte."""
import ctypes as C

import numpy as np
import pytest
import torch

import _libs as L
import motioncam_decoder_amd as M

pytestmark = pytest.mark.gpu


def _run(ctx, items, black, pack12, misalign=0, mem=M.MEM_DEVICE, bits=None):
    """items: (type, w, h, buf, img).  Returns the list of output byte arrays [h, row_bytes]."""
    dev = torch.device("cuda:0")
    if bits:
        pack12 = bits
    ctx.set_post(black=black, bits=bits if bits else (12 if pack12 else None))
    try:
        keep, descs, outs = [], [], []
        for typ, w, h, buf, img in items:
            rb = L.post_row_bytes(w, bits=bits) if bits else L.post_row_bytes(w, pack12)
            cap16 = (h * rb + 1) // 2
            if mem == M.MEM_DEVICE:
                t_in = torch.from_numpy(np.ascontiguousarray(buf)).to(dev)
                t_out = torch.full((cap16 * 2 + 64,), 0xA5, dtype=torch.uint8, device=dev)
                keep += [t_in, t_out]
                descs.append((t_in.data_ptr(), t_in.numel(), w, h, typ, t_out.data_ptr() + misalign, cap16))
                outs.append(t_out)
            else:
                a_in = np.ascontiguousarray(buf)
                a_out = np.full(cap16 * 2 + 64, 0xA5, dtype=np.uint8)
                keep += [a_in, a_out]
                descs.append((a_in.ctypes.data, a_in.size, w, h, typ, a_out.ctypes.data + misalign, cap16))
                outs.append(a_out)
        torch.cuda.synchronize() # (the fills above run on torch's stream, the batch on the context's own)
        written, status = ctx.decode_batch(M.Context.make_frames(descs), mem=mem)
        torch.cuda.synchronize()
        res = []
        for (typ, w, h, buf, img), o, wr, st in zip(items, outs, written, status):
            assert st == 0 and wr == w * h, (typ, w, h, st, wr)
            a = o.cpu().numpy() if mem == M.MEM_DEVICE else o
            rb = L.post_row_bytes(w, bits=bits) if bits else L.post_row_bytes(w, pack12)
            res.append(a[misalign: misalign + h * rb].reshape(h, rb))
            tail = a[misalign + h * rb: misalign + h * rb + 8]
            assert (tail == 0xA5).all(), "wrote past the strip"
        return res
    finally:
        ctx.set_post()


def _items(shapes, seed):
    rng = np.random.default_rng(seed)
    items = []
    for (w, h, nbits) in shapes:
        img = rng.integers(0, 1 << nbits, size=(h, w), dtype=np.uint16)
        img[: max(1, h // 3), : max(1, w // 2)] = 64 # a flat corner: empty blocks / 2-byte records
        items.append((7, w, h, L.encode7(img), img))
        items.append((6, w, h, L.encode6(img), img))
    return items


SHAPES = [(64, 4, 12), (256, 16, 12), (1000, 37, 12), (1001, 9, 10), (77, 6, 16), (1920, 64, 12), (4032, 24, 14)]


@pytest.mark.parametrize("black,pack12", [([64, 64, 64, 64], False), (None, True), ([60, 64, 68, 4000], True)])
def test_post_stage_matches_oracle(gpu_ctx, black, pack12):
    items = _items(SHAPES, 11)
    got = _run(gpu_ctx, items, black, pack12)
    for (typ, w, h, buf, img), g in zip(items, got):
        want = L.oracle_post(img, black, pack12)
        assert np.array_equal(g, want), (typ, w, h, np.argwhere(g != want)[:3])


def test_post_stage_unaligned_output_and_host_memory(gpu_ctx):
    items = _items([(256, 16, 12), (1000, 21, 12)], 12)
    for mem in (M.MEM_DEVICE, M.MEM_HOST):
        got = _run(gpu_ctx, items, [64, 65, 66, 67], True, misalign=2, mem=mem)
        for (typ, w, h, buf, img), g in zip(items, got):
            assert np.array_equal(g, L.oracle_post(img, [64, 65, 66, 67], True)), (typ, w, h, mem)


def test_post_stage_off_again_is_plain_mosaic(gpu_ctx):
    items = _items([(256, 16, 12)], 13)
    _run(gpu_ctx, items, [64, 64, 64, 64], True)
    got = _run(gpu_ctx, items, None, False)
    for (typ, w, h, buf, img), g in zip(items, got):
        assert np.array_equal(g.view("<u2"), img)


def test_post_capacity_is_counted_in_strip_bytes(gpu_ctx):
    w, h = 256, 16
    img = np.full((h, w), 100, np.uint16)
    buf = L.encode7(img)
    dev = torch.device("cuda:0")
    t_in = torch.from_numpy(buf).to(dev)
    t_out = torch.zeros(w * h * 2, dtype=torch.uint8, device=dev)
    gpu_ctx.set_post(pack12=True)
    try:
        need16 = (h * L.post_row_bytes(w, True) + 1) // 2
        for cap, want in ((need16 - 1, M.E_CAPACITY), (need16, 0)):
            fr = M.Context.make_frames([(t_in.data_ptr(), t_in.numel(), w, h, 7, t_out.data_ptr(), cap)])
            written, status = gpu_ctx.decode_batch(fr)
            assert status[0] == want, (cap, status)
    finally:
        gpu_ctx.set_post()


def test_unknown_post_flags_are_rejected(gpu_ctx):
    p = M.Post()
    p.flags = 16  # (1, 2, 4, 8 are black levels and the three strip widths)
    assert gpu_ctx._lib.mcraw_ctx_set_post(gpu_ctx._h, C.byref(p)) != 0
    assert b"post" in gpu_ctx._lib.mcraw_last_error()
    # and the context still decodes plain mosaics
    items = _items([(64, 4, 12)], 14)
    got = _run(gpu_ctx, items, None, False)
    for (typ, w, h, buf, img), g in zip(items, got):
        assert np.array_equal(g.view("<u2"), img)


@pytest.mark.parametrize("bits", [10, 14])
@pytest.mark.parametrize("black", [None, [60, 64, 68, 1000]])
def test_post_stage_10_and_14_bit_strips(gpu_ctx, bits, black):
    """MCRAW_POST_PACK10 / PACK14: both codecs, cropped and odd widths (rows that end inside a byte), saturation."""
    items = _items(SHAPES, 13)
    got = _run(gpu_ctx, items, black, False, bits=bits)
    for (typ, w, h, buf, img), g in zip(items, got):
        want = L.oracle_post(img, black, bits=bits)
        assert np.array_equal(g, want), (bits, typ, w, h, np.argwhere(g != want)[:3])
    # 2-byte aligned output addresses and host-memory batches
    items = _items([(256, 16, 12), (1000, 21, 10)], 14)
    for mem in (M.MEM_DEVICE, M.MEM_HOST):
        got = _run(gpu_ctx, items, black, False, misalign=2, mem=mem, bits=bits)
        for (typ, w, h, buf, img), g in zip(items, got):
            assert np.array_equal(g, L.oracle_post(img, black, bits=bits)), (bits, typ, w, h, mem)


def test_at_most_one_strip_width(gpu_ctx):
    p = M.Post()
    p.flags = M.POST_PACK10 | M.POST_PACK12
    assert gpu_ctx._lib.mcraw_ctx_set_post(gpu_ctx._h, C.byref(p)) != 0
    gpu_ctx.set_post()


@pytest.mark.parametrize("typ,bits,bright", [(7, 12, False), (7, None, False), (7, 10, False), (7, 14, False), (7, 12, True),
                                             (6, 12, False), (6, 10, False), (6, None, False)])
def test_post_stage_large_batch_under_load(gpu_ctx, typ, bits, bright):
    """120 UHD frames per batch (32 for the legacy kernel), every frame checked (on the GPU, against an upload of the oracle's
    rows), every store form of both kernels: what goes wrong only when the memory pipeline is backed up -- a store whose data
    registers are overwritten before it has read them wrote wrong 12-bit strips for the bench's 240-frame batch while every
    small-frame test passed.  `bright`: frames whose samples reach 4095 and whose references lie below the black levels in
    places, so that the items of one batch take both paths of the 12-bit stage (the lean one: no clamp, black levels off the
    references; and the general one)."""
    dev = torch.device("cuda:0")
    w, h, n, distinct = 3840, 2160, (120 if typ == 7 else 32), 4
    black = [64, 60, 68, 72] if not bright else [300, 2, 1200, 40]
    imgs = [L.synth_image(w, h, 12, 1, 12.0, 900 + i).copy() for i in range(distinct)]
    if bright:
        for k, im in enumerate(imgs):
            im[200 * k: 200 * k + 300, :] = np.minimum(4095, im[200 * k: 200 * k + 300, :].astype(np.uint32) * 3).astype(np.uint16)  # clipped highlights
            im[1000:1100, 64 * k: 64 * k + 1024] //= 16                                                                         # shadows below the black levels
    bufs = [(L.encode7 if typ == 7 else L.encode6)(im) for im in imgs]
    want = [torch.from_numpy(np.ascontiguousarray(L.oracle_post(im, black, bits=bits))).to(dev) for im in imgs]
    rb = L.post_row_bytes(w, bits=bits)
    cap16 = (h * rb + 1) // 2
    t_in = [torch.from_numpy(b).to(dev) for b in bufs]
    t_out = torch.zeros(n * cap16 * 2, dtype=torch.uint8, device=dev)
    descs = [(t_in[i % distinct].data_ptr(), t_in[i % distinct].numel(), w, h, typ, t_out.data_ptr() + i * cap16 * 2, cap16) for i in range(n)]
    frames = M.Context.make_frames(descs)
    gpu_ctx.set_post(black=black, bits=bits)
    try:
        for rnd in range(3):
            t_out.zero_()
            torch.cuda.synchronize() # (the batch runs on the context's own stream, not on torch's)
            written, status = gpu_ctx.decode_batch(frames)
            torch.cuda.synchronize()
            assert all(s == 0 for s in status) and all(wr == w * h for wr in written)
            for i in range(n):
                got = t_out[i * cap16 * 2: i * cap16 * 2 + h * rb].view(h, rb)
                if not torch.equal(got, want[i % distinct]):
                    bad = (got != want[i % distinct]).nonzero()
                    raise AssertionError("type %d bits %s round %d frame %d: %d bytes differ, first at (row, byte) %s" %
                                         (typ, bits, rnd, i, bad.shape[0], bad[:4].tolist()))
    finally:
        gpu_ctx.set_post()


def test_14_bit_rows_one_store_per_lane_every_neighbour_case(gpu_ctx):
    """Round 5: in the interior of a frame a lane writes its 14 bytes of a 14-bit row AND the first two bytes of the row's next
    8 samples (fetched across lanes), or -- when that piece is not decoded in the same pass: the last tile of a pass, the
    frame's last tile column -- the last two bytes of the previous piece in front of its own.  Geometries that put every case
    on every position: one to nine tile columns (a pass holds four tiles: rows wrap inside passes), heights that end
    inside a tile row, widths off the 64 grid (their cropped tiles take the old path), natural content (lean items) and
    noise with raw blocks (not lean), with and without black levels, dword-misaligned buffers."""
    rng = np.random.default_rng(77)
    items = []
    for w in (64, 128, 192, 256, 320, 448, 576, 200, 1000):
        for h in (4, 10, 36):
            img = rng.integers(0, 1 << 14, size=(h, w), dtype=np.uint16) if (w // 64 + h) % 2 else L.natural_image_np(w, h, 12, 12.0, w + h)
            items.append((7, w, h, L.encode7(img), img))
    for black in (None, [256, 256, 256, 256], [60, 64, 68, 4000]):
        for mis in (0, 2):
            got = _run(gpu_ctx, items, black, False, misalign=mis, bits=14)
            for (typ, w, h, buf, img), g in zip(items, got):
                want = L.oracle_post(img, black, bits=14)
                assert np.array_equal(g, want), (w, h, black, mis, np.argwhere(g != want)[:3])
