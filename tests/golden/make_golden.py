#!/usr/bin/env python3
"""Generate tests/golden/mcraw_golden.npz from the REAL reference codec.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Inputs come from the repo's own encoder (the reference has none);
every expected output and return value is produced by the reference's
motioncam::raw::Decode / DecodeLegacy compiled from its own sources
(oracle/Makefile target `ref`).  The fixture is data only: encoded bytes,
geometry, expected uint16 mosaic, expected return value.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _libs as L  # noqa: E402


def cases():
    rng = np.random.default_rng(20250824)
    out = []

    # ---- type 7: every `bits` value 0..16 forced on every block (RawData.cpp:424-458)
    for b in range(17):
        w, h = 128, 8
        img = rng.integers(0, max(1, 1 << b), size=(h, w), dtype=np.uint16) if b else np.full((h, w), 777, np.uint16)
        nblk = (w // 64) * (h // 4) * 4
        out.append(("t7_bits%02d" % b, 7, img, np.full(nblk, b, np.uint8), 0))
    # mixed classes chosen per block
    w, h = 256, 64
    img = L.natural_image_np(w, h, 12, 12.0, 1000 * 2 + 0)
    nblk = (w // 64) * (h // 4) * 4
    out.append(("t7_nat12_256x64", 7, img, None, 0))
    out.append(("t7_forced_mix_256x64", 7, img, rng.integers(0, 17, nblk).astype(np.uint8), 0))
    out.append(("t7_uniform10_192x16", 7, L.uniform_image_np(192, 16, 10, 1000 * 1 + 0), None, 0))
    out.append(("t7_uniform14_128x8", 7, L.uniform_image_np(128, 8, 14, 1000 * 4 + 0), None, 0))
    out.append(("t7_uniform16_128x8", 7, L.uniform_image_np(128, 8, 16, 5), None, 0))  # refs > 4095, u16 wrap range
    # cropped: width < encodedWidth (RawData.cpp:598-608)
    out.append(("t7_crop_200x12", 7, L.natural_image_np(200, 12, 12, 12.0, 31), None, 0))
    out.append(("t7_crop_100x8", 7, L.natural_image_np(100, 8, 10, 4.0, 32), None, 0))
    out.append(("t7_crop_odd_77x4", 7, L.natural_image_np(77, 4, 14, 40.0, 33), None, 0))
    # side-stream entry count that is a whole number of records (3 x 64)
    out.append(("t7_n192_192x16", 7, L.natural_image_np(192, 16, 12, 12.0, 34), None, 0))

    # ---- type 6: every header nibble 0..15 forced on every record (RawData_Legacy.cpp:401-439)
    for b in range(16):
        w, h = 96, 4
        lim = b if b <= 10 else 12
        img = rng.integers(0, max(1, 1 << lim), size=(h, w), dtype=np.uint16) if b else np.full((h, w), 300, np.uint16)
        nrec = (w // 32) * 2 * h
        out.append(("t6_bits%02d" % b, 6, img, np.full(nrec, b, np.uint8), b & 1))
    out.append(("t6_nat10_160x24", 6, L.natural_image_np(160, 24, 10, 4.0, 41), None, 0))
    out.append(("t6_pad_80x6", 6, L.natural_image_np(80, 6, 12, 12.0, 42), None, 1))  # width % 32 != 0 (:34-36,490)
    out.append(("t6_pad_odd_75x5", 6, L.uniform_image_np(75, 5, 14, 43), None, 0))
    out.append(("t6_uniform16_64x4", 6, L.uniform_image_np(64, 4, 16, 44), None, 0))  # refs clamp at 4095
    nrec = (256 // 32) * 2 * 16
    out.append(("t6_forced_mix_256x16", 6, L.natural_image_np(256, 16, 12, 12.0, 45),
                rng.integers(0, 16, nrec).astype(np.uint8), 1))

    # ---- streams long enough for the GPU path's hand-offs (round 6; the npz is compressed, so these are built from few
    # distinct values): legacy streams of more than three 16 KiB segments (look-back between workgroups), one of them with
    # a padded width and a trailer; a type-7 frame whose bits AND refs streams are longer than one 32 KiB piece
    rng2 = np.random.default_rng(20261003)
    out.append(("t6_segments_640x128", 6, L.natural_image_np(640, 128, 12, 12.0, 46), None, 0))
    w, h = 600, 100
    nrec = ((w + 31) // 32) * 2 * h
    out.append(("t6_segments_forced_600x100", 6, L.natural_image_np(w, h, 10, 4.0, 47),
                rng2.integers(0, 16, nrec).astype(np.uint8), 1))
    w, h = 2048, 1600  # 51 200 blocks = 800 records per side stream
    tiles = rng2.integers(0, 1 << 16, size=(h // 4, w // 64, 4), dtype=np.uint16)  # one value per block: refs of 16 bits
    yy, xx = np.arange(h)[:, None], np.arange(w)[None, :]
    img = tiles[yy // 4, xx // 64, (yy & 1) * 2 + (xx & 1)]
    out.append(("t7_long_side_streams_2048x1600", 7, np.ascontiguousarray(img),
                rng2.integers(0, 17, (w // 64) * (h // 4) * 4).astype(np.uint8), 0))
    return out


def main():
    if L.ref() is None:
        sys.exit("reference library not built: run `make -C oracle ref` where /root/reference exists")
    store = {}
    names = []
    for name, typ, img, min_bits, flags in cases():
        h, w = img.shape
        if typ == 7:
            buf = L.encode7(img, min_bits, flags)
            ret, out = L.ref_decode7(buf, w, h)
            out = out[:h]
        else:
            buf = L.encode6(img, min_bits, flags)
            ret, out = L.ref_decode6(buf, w, h)
        assert ret == w * h, (name, ret)
        assert np.array_equal(out, img), name  # the reference inverts the encoder
        store[name + "/buf"] = buf
        store[name + "/out"] = out
        store[name + "/meta"] = np.array([typ, w, h, ret], dtype=np.int64)
        names.append(name)
    store["names"] = np.array(names)
    path = os.path.join(HERE, "mcraw_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote %s: %d cases, %.1f KiB" % (path, len(names), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
