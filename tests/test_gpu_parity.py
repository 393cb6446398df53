"""GPU (-m gpu): the HIP decode path, called through the C ABI with HBM-resident
buffers, must be bit-exact against (a) the golden vectors produced by the real
reference and (b) the CPU oracle on seeded synthetic frames.  Integer work: the
bar is memcmp equality, including the return value."""
import numpy as np
import pytest

import _libs as L
import motioncam_decoder_amd as M

pytestmark = pytest.mark.gpu


def _ref_safe(t, w, h, buf):
    """Frames the real reference decodes without leaving its buffers: it writes whole tile rows (RawData.cpp:598-608)
    and whole side-stream records of 64 entries into vectors sized by the stream's own count (:463-498)."""
    if t == 6:
        return True
    if h % 4 or buf.size < 16:
        return False
    bo, ro = (int(x) for x in np.frombuffer(buf[8:16].tobytes(), np.uint32))
    counts = [int(np.frombuffer(buf[o:o + 4].tobytes(), np.uint32)[0]) if o + 4 <= buf.size else 1 for o in (bo, ro)]
    return all(c % 64 == 0 for c in counts)


def _check(ctx, items, expect, with_ref=True):
    """HIP output == `expect` (the oracle's, or a golden vector's); and, where the real reference codec travelled to this box
    (oracle/_ref, built in the container), == what IT makes of the same bytes -- every frame of every test below is then
    compared with the reference itself, not only through the oracle."""
    from _gpu import decode_batch_device
    written, status, outs = decode_batch_device(ctx, [(t, w, h, b) for (t, w, h, b) in items])
    ref = L.ref() if with_ref else None
    for i, ((t, w, h, b), (ret, img)) in enumerate(zip(items, expect)):
        assert status[i] == 0, (i, t, w, h, status[i])
        assert written[i] == ret, (i, t, w, h, written[i], ret)
        if not np.array_equal(outs[i], img):
            bad = np.argwhere(outs[i] != img)
            raise AssertionError("frame %d type %d %dx%d: %d mismatches, first at %s got %d want %d" % (
                i, t, w, h, len(bad), bad[0], outs[i][tuple(bad[0])], img[tuple(bad[0])]))
        if ref is not None and _ref_safe(t, w, h, b):  # (frames the reference would overrun its buffers on are left to the oracle)
            rr, orr = (L.ref_decode7 if t == 7 else L.ref_decode6)(b, w, h)
            assert rr == ret and np.array_equal(outs[i], orr[:h]), "frame %d type %d %dx%d differs from the reference codec" % (i, t, w, h)


def test_golden_vectors_one_batch(gpu_ctx, golden):
    names = sorted(golden)
    items = [(golden[n]["type"], golden[n]["w"], golden[n]["h"], golden[n]["buf"]) for n in names]
    expect = [(golden[n]["ret"], golden[n]["out"]) for n in names]
    _check(gpu_ctx, items, expect)


def test_golden_vectors_one_by_one(gpu_ctx, golden):
    for n in sorted(golden):
        c = golden[n]
        _check(gpu_ctx, [(c["type"], c["w"], c["h"], c["buf"])], [(c["ret"], c["out"])])


@pytest.mark.parametrize("w,h", [(64, 4), (128, 8), (256, 64), (192, 12), (100, 8), (77, 4), (1000, 20), (1024, 256)])
def test_type7_random_classes_vs_oracle(gpu_ctx, w, h):
    rng = np.random.default_rng(w * 7 + h)
    encW, encH = (w + 63) // 64 * 64, (h + 3) // 4 * 4
    items, expect = [], []
    for trial in range(8):
        nb = int(rng.integers(1, 17))
        img = rng.integers(0, 1 << nb, size=(h, w), dtype=np.uint16)
        mb = rng.integers(0, 17, size=encW * encH // 64).astype(np.uint8) if trial % 2 else None
        buf = L.encode7(img, mb, flags=(trial >> 1) & 1)  # also unrounded side-stream counts
        ret, out = L.oracle_decode7(buf, w, h)
        assert ret == w * h and np.array_equal(out, img)
        items.append((7, w, h, buf))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


@pytest.mark.parametrize("w,h", [(32, 1), (96, 4), (80, 6), (75, 5), (33, 3), (256, 16), (1000, 30), (2048, 64)])
def test_type6_random_classes_vs_oracle(gpu_ctx, w, h):
    rng = np.random.default_rng(w * 5 + h)
    items, expect = [], []
    for trial in range(8):
        nb = int(rng.integers(1, 17))
        img = rng.integers(0, 1 << nb, size=(h, w), dtype=np.uint16)
        nrec = ((w + 31) // 32) * 2 * h
        mb = rng.integers(0, 16, size=nrec).astype(np.uint8) if trial % 2 else None
        buf = L.encode6(img, mb, flags=trial & 1)
        ret, out = L.oracle_decode6(buf, w, h)
        assert ret == w * h and np.array_equal(out, img)
        items.append((6, w, h, buf))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_height_not_multiple_of_4_never_writes_past_the_frame(gpu_ctx):
    # the reference writes width*encodedHeight; the build clips at `height`
    from _gpu import decode_batch_device
    img = L.natural_image_np(128, 6, 12, 12.0, 6)
    buf = L.encode7(img)
    written, status, outs = decode_batch_device(gpu_ctx, [(7, 128, 6, buf)], fill=0xA5, out_rows_extra=2)
    assert status == [0] and written == [128 * 6]
    assert np.array_equal(outs[0][:6], img)
    assert np.all(outs[0][6:] == 0xA5A5)


def test_baseline_config1_and_config2_full_size(gpu_ctx):
    # config 1: 1920x1080 10-bit; config 2: 4032x3024 12-bit; U and Nat
    items, expect = [], []
    for (w, h, nb, dist, sig, seed) in [(1920, 1080, 10, 0, 0, 1000), (1920, 1080, 10, 1, 4, 1001),
                                        (4032, 3024, 12, 0, 0, 2000), (4032, 3024, 12, 1, 12, 2001)]:
        img = L.synth_image(w, h, nb, dist, sig, seed)
        buf = L.encode7(img)
        ret, out = L.oracle_decode7(buf, w, h)
        assert ret == w * h
        items.append((7, w, h, buf))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_baseline_config4_mixed_batch(gpu_ctx):
    # 14-bit type 7 (U and Nat) interleaved with legacy frames incl. width % 32 != 0;
    # every storage class forced at least once; 64 frames interleaved by frame index
    rng = np.random.default_rng(4000)
    items, expect = [], []
    for i in range(64):
        if i % 2 == 0:
            w, h = ((1920, 1080), (4032, 3024))[(i // 2) % 2]
            img = L.synth_image(w, h, 14, (i // 4) % 2, 40.0, 4000 + i)
            nblk = ((w + 63) // 64) * ((h + 3) // 4) * 4
            mb = rng.integers(0, 17, nblk).astype(np.uint8) if i % 8 == 0 else None
            buf = L.encode7(img, mb)
            ret, out = L.oracle_decode7(buf, w, h)
            items.append((7, w, h, buf))
        else:
            w, h = ((1920, 1080), (4000, 3000))[(i // 2) % 2]
            nb = (10, 12, 14)[(i // 2) % 3]
            img = L.synth_image(w, h, nb, (i // 4) % 2, 12.0, 4000 + i)
            nrec = ((w + 31) // 32) * 2 * h
            mb = rng.integers(0, 16, nrec).astype(np.uint8) if i % 8 == 1 else None
            buf = L.encode6(img, mb, flags=1)
            ret, out = L.oracle_decode6(buf, w, h)
            items.append((6, w, h, buf))
        assert ret == w * h and np.array_equal(out, img)
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_roundtrip_property_8k(gpu_ctx):
    # size-independent property at the largest BASELINE geometry (config 5): decode(encode(x)) == x
    w, h = 7680, 4320
    for dist, seed in ((0, 5000), (1, 5001)):
        img = L.synth_image(w, h, 12, dist, 12.0, seed)
        buf = L.encode7(img)
        _check(gpu_ctx, [(7, w, h, buf)], [(w * h, img)])


def _shift_streams(buf, pad_bits, pad_refs):
    """Re-lay a type-7 frame with `pad_*` junk bytes in front of each side stream (the header
    offsets are free-form: lib/RawData.cpp:513-523 just reads them)."""
    bo = int(np.frombuffer(buf[8:12].tobytes(), np.uint32)[0])
    ro = int(np.frombuffer(buf[12:16].tobytes(), np.uint32)[0])
    out = np.concatenate([buf[:bo], np.full(pad_bits, 0xEE, np.uint8), buf[bo:ro], np.full(pad_refs, 0x77, np.uint8), buf[ro:]])
    out = out.copy()
    out[8:12] = np.frombuffer(np.uint32(bo + pad_bits).tobytes(), np.uint8)
    out[12:16] = np.frombuffer(np.uint32(ro + pad_bits + pad_refs).tobytes(), np.uint8)
    return out


@pytest.mark.parametrize("pads", [(1, 0), (0, 1), (3, 2), (5, 7), (2, 2), (1030, 515)])
def test_side_streams_at_odd_offsets(gpu_ctx, pads):
    items, expect = [], []
    for (w, h, nb, sig, seed) in ((256, 32, 12, 12.0, 1), (1000, 20, 14, 40.0, 2), (640, 480, 10, 4.0, 3)):
        img = L.natural_image_np(w, h, nb, sig, seed)
        buf = _shift_streams(L.encode7(img), *pads)
        ret, out = L.oracle_decode7(buf, w, h)
        assert ret == w * h and np.array_equal(out, img)
        if L.ref() is not None:
            rr, orr = L.ref_decode7(buf, w, h)
            assert np.array_equal(orr[:h], img)
        items.append((7, w, h, buf))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_flat_frames_densest_chains(gpu_ctx):
    # constant images: every legacy record is 2 bytes (512 records per KiB chunk -> k6_decode lists its
    # 8-chunk window in several rounds), every type-7 block is 0 bytes (empty payload spans) and the
    # side streams are runs of 2-byte records
    items, expect = [], []
    for (w, h, val) in ((4096, 64, 0), (1000, 37, 4095), (2048, 16, 65535)):
        img = np.full((h, w), val, np.uint16)
        b6 = L.encode6(img)
        ret, out = L.oracle_decode6(b6, w, h)
        assert ret == w * h and np.array_equal(out, img)
        items.append((6, w, h, b6))
        expect.append((ret, out))
        b7 = L.encode7(img)
        ret, out = L.oracle_decode7(b7, w, h)
        assert ret == w * h and np.array_equal(out, img)
        items.append((7, w, h, b7))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_legacy_frames_of_one_record_size_each_over_several_segments(gpu_ctx):
    # every record of a frame has the same size (header nibble forced): the lists of k6_decode's unpacking waves then hold
    # 120 .. 2048 records per wave, on both sides of every threshold of the lean path -- every record listed (up to 352 records),
    # one entry per record pair (up to 736), several rounds of 768 -- and frames of 5 .. 40 segments, the last one partly filled
    items, expect = [], []
    rng = np.random.default_rng(66)
    for nib in range(16):
        nb = nib if nib <= 10 else 16
        w = 1024 + 32 * nib
        h = max(96, int(6 * 16384 / (2 + 2 * nb) * 16 / w) + 1 + nib) # at least six segments of 16 KiB
        img = rng.integers(0, 1 << max(nb, 1), size=(h, w), dtype=np.uint16) if nb else np.full((h, w), 321, np.uint16)
        nrec = ((w + 31) // 32) * 2 * h
        buf = L.encode6(img, np.full(nrec, nib, np.uint8))
        ret, out = L.oracle_decode6(buf, w, h)
        assert ret == w * h and np.array_equal(out, img)
        items.append((6, w, h, buf))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_legacy_frames_of_mixed_record_sizes_around_the_list_thresholds(gpu_ctx):
    # round 6: a wave of k6_decode lists every record up to 352 records, every second one (from the quarter walks' notes) up to 736,
    # and leaves the lean path above (its lanes then walk a quarter chunk each); frames whose records are drawn from two or three
    # neighbouring sizes put the waves of ONE frame on both sides of each threshold, quarter by quarter
    items, expect = [], []
    rng = np.random.default_rng(606)
    for nibs, w in (((4, 5, 6), 1504), ((3, 4, 5), 1280), ((2, 3), 1056), ((1, 2, 3), 2016), ((0, 1, 2), 992), ((0, 5), 1600), ((2, 12), 1184)):
        h = 260
        rpr = w // 32 * 2                                           # records per row (w is a multiple of 32)
        nrec = rpr * h
        run = rng.integers(1, 400, size=nrec)                      # sizes change in runs of 1 .. 400 records
        idx = np.repeat(np.arange(nrec), run)[:nrec] % len(nibs)
        nib = np.asarray(nibs, np.int64)[rng.permutation(len(nibs))][idx].reshape(h, w // 32, 2)
        # record (y, g, p) holds the samples of columns 32 g + 2 i + p: residuals of nib[y, g, p] bits above a common reference
        bits = np.repeat(nib, 16, axis=1).reshape(h, w // 32, 16, 2).reshape(h, w)
        img = (100 + (rng.random((h, w)) * (1 << bits)).astype(np.int64)).astype(np.uint16)
        buf = L.encode6(img)
        ret, out = L.oracle_decode6(buf, w, h)
        assert ret == w * h and np.array_equal(out, img)
        assert 0.8 * nrec * (2 + 2 * min(nibs)) <= buf.size <= 1.2 * nrec * (2 + 2 * max(nibs)) + 64  # (the sizes came out as meant)
        items.append((6, w, h, buf))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_flat_and_textured_bands_mix_dense_and_sparse_chunks(gpu_ctx):
    # bands of constant rows between bands of noise: the side streams alternate between runs of
    # 2-byte records (dense chunks, listed by pointer doubling) and ordinary records (sparse chunks,
    # grouped four to a work item), so groups that hold both kinds occur; likewise the legacy chain
    # alternates between 2-byte and long records inside one k6_decode window
    rng = np.random.default_rng(77)
    items, expect = [], []
    for (w, h, band) in ((2048, 192, 24), (1024, 1536, 512), (1000, 150, 10)):
        img = rng.integers(0, 4096, size=(h, w), dtype=np.uint16)
        for y0 in range(0, h, 2 * band):
            img[y0:y0 + band] = 517
        for typ, enc, dec in ((7, L.encode7, L.oracle_decode7), (6, L.encode6, L.oracle_decode6)):
            buf = enc(img)
            ret, out = dec(buf, w, h)
            assert ret == w * h and np.array_equal(out, img)
            items.append((typ, w, h, buf))
            expect.append((ret, out))
    _check(gpu_ctx, items, expect)


def test_rows_wider_than_24_bits(gpu_ctx):
    # a strip wider than 2^24 pixels: row offsets and column arithmetic must not assume 24-bit widths
    rng = np.random.default_rng(2424)
    w, h = (1 << 24) + 4000, 3
    img = rng.integers(0, 1024, size=(h, w), dtype=np.uint16)
    img[:, 5_000_000:9_000_000] = 77 # a long run of 2-byte records / empty blocks in the middle
    items, expect = [], []
    for typ, enc, dec in ((6, L.encode6, L.oracle_decode6), (7, L.encode7, L.oracle_decode7)):
        buf = enc(img)
        ret, out = dec(buf, w, h)
        assert ret == w * h and np.array_equal(out, img)
        items.append((typ, w, h, buf))
        expect.append((w * h, img))
    _check(gpu_ctx, items, expect)


def test_batch_of_many_different_sizes(gpu_ctx):
    # more distinct frame sizes than the unpack kernel has size classes (8), in random order: the plans
    # are sorted into classes for the launches, statuses and outputs must still land on the caller's frames
    rng = np.random.default_rng(1212)
    sizes = [(64 * k, 4 * (k + 1)) for k in range(1, 13)] + [(1000, 37), (77, 6)]
    items, expect = [], []
    for i in rng.permutation(len(sizes) * 3):
        w, h = sizes[int(i) % len(sizes)]
        img = rng.integers(0, 1 << int(rng.integers(1, 15)), size=(h, w), dtype=np.uint16)
        bad = int(i) % 7 == 3
        buf = L.encode7(img)
        if bad:
            buf = buf[: buf.size - 5].copy()  # the refs stream loses its tail
        ret, out = L.oracle_decode7(buf, w, h)
        items.append((7, w, h, buf))
        expect.append((ret, out))
    from _gpu import decode_batch_device
    written, status, outs = decode_batch_device(gpu_ctx, items)
    for i, ((t, w, h, b), (ret, img)) in enumerate(zip(items, expect)):
        if ret == 0:
            assert status[i] != 0 and written[i] == 0, (i, w, h, status[i])
        else:
            assert status[i] == 0 and written[i] == ret and np.array_equal(outs[i], img), (i, w, h, status[i])


def test_many_small_frames_one_batch(gpu_ctx):
    # 1200 frames of assorted small geometries, both encodings interleaved: exercises the batch
    # indexing (uniform-stride workspace sized by the largest frame, work lists, status mapping)
    rng = np.random.default_rng(1200)
    shapes = [(64, 4), (128, 8), (192, 12), (100, 8), (77, 4), (256, 16), (320, 20), (33, 3)]
    pool = []
    for i, (w, h) in enumerate(shapes):
        img = rng.integers(0, 1 << int(rng.integers(1, 15)), size=(h, w), dtype=np.uint16)
        pool.append((7, w, h, L.encode7(img), img))
        pool.append((6, w, h, L.encode6(img), img))
    items, expect = [], []
    for i in range(1200):
        t, w, h, buf, img = pool[int(rng.integers(0, len(pool)))]
        items.append((t, w, h, buf))
        expect.append((w * h, img))
    _check(gpu_ctx, items, expect)


def test_frames_decoded_in_place_from_a_file_image_in_hbm(gpu_ctx, tmp_path):
    # the whole .mcraw file is uploaded once; every frame is decoded from where its BUFFER payload sits
    # in that image (arbitrary byte alignment -- container items are packed back to back)
    import struct
    import torch
    import motioncam_decoder_amd as M
    rng = np.random.default_rng(99)
    specs = [(1000 + i, (7, 6)[i % 2], (640, 1000, 200, 1920)[i % 4], (480, 37, 12, 64)[i % 4]) for i in range(12)]
    frames, images = [], {}
    for ts, typ, w, h in specs:
        img = rng.integers(0, 1 << int(rng.integers(4, 15)), size=(h, w), dtype=np.uint16)
        frames.append((ts, typ, w, h, L.encode7(img) if typ == 7 else L.encode6(img)))
        images[ts] = img
    path = L.write_mcraw(str(tmp_path / "clip.mcraw"), frames)
    raw = np.fromfile(path, dtype=np.uint8)
    # walk the items (8-byte header: type u32, size u32; BUFFER = 2), note the payload offsets
    off, payloads = 8, []
    while off + 8 <= raw.size:
        t, size = struct.unpack_from("<II", raw, off)
        if t == 2:
            payloads.append((off + 8, size))
        off += 8 + size
    assert len(payloads) == len(specs) and any(p % 16 for p, _ in payloads) and any(p % 2 for p, _ in payloads if True)
    dev = torch.device("cuda:0")
    image = torch.from_numpy(raw).to(dev)
    outs, descs = [], []
    for (ts, typ, w, h), (p, size) in zip(specs, payloads):
        o = torch.zeros(w * h * 2, dtype=torch.uint8, device=dev)
        outs.append(o)
        descs.append((image.data_ptr() + p, size, w, h, typ, o.data_ptr(), w * h))
    written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs))
    torch.cuda.synchronize()
    for (ts, typ, w, h), o, wr, st in zip(specs, outs, written, status):
        assert st == 0 and wr == w * h, (ts, typ, st)
        assert np.array_equal(o.cpu().numpy().view(np.uint16).reshape(h, w), images[ts]), ts


def test_legacy_batch_of_very_different_stream_lengths(gpu_ctx):
    # k6_decode launches one workgroup per 16 KiB segment, round by round over the frames sorted by size: a batch of a
    # one-chunk frame, a few-segment frame and many-segment frames (the rounds in which only the largest frames are still
    # in play), decoded twice (the second batch finds the look-back state of the first: older epoch, must read as empty)
    rng = np.random.default_rng(4242)
    shapes = ((64, 4, 10), (4000, 3000, 12), (320, 200, 12), (1920, 1080, 14), (1000, 37, 10), (4032, 3024, 10), (96, 8, 16))
    items, expect = [], []
    for k, (w, h, nb) in enumerate(shapes):
        img = L.natural_image_np(w, h, min(nb, 14), 12.0, 900 + k) if k % 2 else rng.integers(0, 1 << nb, size=(h, w), dtype=np.uint16)
        buf = L.encode6(img)
        ret, out = L.oracle_decode6(buf, w, h)
        assert ret == w * h and np.array_equal(out, img)
        items.append((6, w, h, buf))
        expect.append((ret, out))
    _check(gpu_ctx, items, expect)
    _check(gpu_ctx, items[::-1], expect[::-1])


def test_legacy_streams_whose_chunk_maps_are_never_unanimous(gpu_ctx):
    # Uniform 16-bit samples: every record is raw (stride 34) and the payload is noise, and -- the harder case -- a
    # frame of ONE record size whose payload repeats the header pattern, so that all 17 phases of a chunk are chains of
    # their own that never meet: no chunk map is unanimous, every segment has to take its entry phase from the
    # segment in front of it (k6_decode's look-back carries it)
    rng = np.random.default_rng(99)
    items, expect = [], []
    w, h = 1920, 270
    img = rng.integers(0, 65536, size=(h, w), dtype=np.uint16)
    buf = L.encode6(img)
    ret, out = L.oracle_decode6(buf, w, h)
    assert ret == w * h and np.array_equal(out, img)
    items.append((6, w, h, buf))
    expect.append((ret, out))
    # payload bytes that read as headers of the same record size: the constant 0xBFFF is coded with the reference 0xFFF
    # and raw residuals 0xB000, so every even byte of the stream -- the headers' 0xFF and the payload's 0xB0 -- has a
    # nibble >= 11 = "raw record, 34 bytes"
    img2 = np.full((h, w), 0xBFFF, np.uint16)
    buf2 = L.encode6(img2)
    assert all((int(b) >> 4) >= 11 for b in buf2[0:4096:2])
    ret2, out2 = L.oracle_decode6(buf2, w, h)
    assert ret2 == w * h and np.array_equal(out2, img2)
    items.append((6, w, h, buf2))
    expect.append((ret2, out2))
    _check(gpu_ctx, items, expect)


def test_legacy_stream_cut_inside_a_late_segment(gpu_ctx):
    # a many-segment stream that ends in the middle of a record far from its start: the chain dies there, every later
    # chunk is entered by nothing, the frame fails as truncated and its neighbours decode
    from _gpu import decode_batch_device
    w, h = 1920, 540
    img = L.natural_image_np(w, h, 12, 12.0, 31)
    good = L.encode6(img)
    cut = good[: (good.size * 2 // 3) | 1].copy()
    ret, _ = L.oracle_decode6(cut, w, h)
    assert ret == 0
    written, status, outs = decode_batch_device(gpu_ctx, [(6, w, h, good), (6, w, h, cut), (6, w, h, good)])
    assert status[0] == 0 and status[2] == 0 and written[0] == w * h and written[2] == w * h
    assert np.array_equal(outs[0], img) and np.array_equal(outs[2], img)
    assert status[1] & M.E_TRUNCATED and written[1] == 0
