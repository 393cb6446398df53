"""The post-decode stage (black levels, 12-bit strips) has no reference function: its definition is
oracle/mcraw_oracle.c:mcraw_oracle_post.  Pin that definition against an independent numpy
statement, and against hand-computed bytes."""
import numpy as np
import pytest

import _libs as L


def test_hand_computed_strip():
    # samples 0xABC, 0x123, 0xFFF, 0x001, 0x800 -> AB C1 23 | FF F0 01 | 80 0(pad)
    img = np.array([[0xABC, 0x123, 0xFFF, 0x001, 0x800]], dtype=np.uint16)
    got = L.oracle_post(img, None, True)
    assert got.tolist() == [[0xAB, 0xC1, 0x23, 0xFF, 0xF0, 0x01, 0x80, 0x00]]
    # black levels by CFA position (row & 1) * 2 + (col & 1), saturating at 0; 12-bit saturation at 4095
    img = np.array([[100, 50, 5000, 7], [10, 65535, 0, 9]], dtype=np.uint16)
    got = L.oracle_post(img, [64, 60, 8, 1], False).view("<u2")
    assert got.tolist() == [[36, 0, 4936, 0], [2, 65534, 0, 8]]
    got = L.oracle_post(img, [64, 60, 8, 1], True)
    assert got.tolist() == [[0x02, 0x40, 0x00, 0xFF, 0xF0, 0x00], [0x00, 0x2F, 0xFF, 0x00, 0x00, 0x08]]


@pytest.mark.parametrize("w,h", [(2, 2), (8, 2), (13, 5), (64, 4), (101, 7), (1000, 20)])
def test_oracle_post_matches_numpy(w, h):
    rng = np.random.default_rng(w * 31 + h)
    for nbits in (10, 12, 16):
        img = rng.integers(0, 1 << nbits, size=(h, w), dtype=np.uint16)
        for black in (None, [64, 64, 64, 64], [0, 1, 4095, 65535]):
            for pack12 in (False, True):
                assert np.array_equal(L.oracle_post(img, black, pack12), L.post_np(img, black, pack12))


def test_hand_computed_10_and_14_bit_strips():
    # 10 bits: 0x3FF, 0x001, 0x200, 0x155 -> 1111111111 0000000001 1000000000 0101010101 = FF C0 18 01 55
    img = np.array([[0x3FF, 0x001, 0x200, 0x155, 0x7FF]], dtype=np.uint16)  # (0x7FF saturates to 0x3FF)
    assert L.oracle_post(img, None, bits=10).tolist() == [[0xFF, 0xC0, 0x18, 0x01, 0x55, 0xFF, 0xC0]]
    # 14 bits: 0x3FFF, 0x0001 -> 11111111111111 00000000000001 + 4 bits of padding = FF FC 00 10
    img = np.array([[0x3FFF, 0x0001], [0x2AAA, 0xFFFF]], dtype=np.uint16)
    assert L.oracle_post(img, None, bits=14).tolist() == [[0xFF, 0xFC, 0x00, 0x10], [0xAA, 0xAB, 0xFF, 0xF0]]


@pytest.mark.parametrize("w,h", [(2, 2), (8, 2), (13, 5), (77, 3), (1000, 6)])
def test_oracle_post_10_14_matches_numpy(w, h):
    rng = np.random.default_rng(w * 131 + h)
    for nbits in (10, 14, 16):
        img = rng.integers(0, 1 << nbits, size=(h, w), dtype=np.uint16)
        for black in (None, [64, 0, 1023, 3]):
            for bits in (10, 14):
                assert np.array_equal(L.oracle_post(img, black, bits=bits), L.post_np(img, black, bits=bits)), (nbits, bits)
