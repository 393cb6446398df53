"""CPU: malformed frames make the oracle fail cleanly (return 0) where the reference
has undefined behaviour."""
import numpy as np

import _libs as L


def _frame7(w=128, h=8, seed=3):
    img = L.natural_image_np(w, h, 12, 12.0, seed)
    return img, L.encode7(img)


def test_truncated_type7_fails():
    img, buf = _frame7()
    for cut in (1, 8, 40, buf.size // 2, buf.size - 17):
        ret, _ = L.oracle_decode7(buf[: buf.size - cut], 128, 8)
        assert ret == 0


def test_bad_header_type7_fails():
    img, buf = _frame7()
    for off, val in ((0, 100), (0, 64), (8, 1 << 30), (12, 1 << 30)):  # encW%64, encW<width, offsets > len
        b = buf.copy()
        b[off:off + 4] = np.frombuffer(np.uint32(val).tobytes(), np.uint8)
        ret, _ = L.oracle_decode7(b, 128, 8)
        assert ret == 0, (off, val)


def test_bits_above_16_fails():
    img, buf = _frame7()
    bits_off = int(np.frombuffer(buf[8:12].tobytes(), np.uint32)[0])
    b = buf.copy()
    b[bits_off + 4] = (b[bits_off + 4] & 0xF0) | 0x0F  # reference 0xF?? -> bits values 3840+ > 16
    ret, _ = L.oracle_decode7(b, 128, 8)
    assert ret == 0


def test_unrounded_side_stream_count_is_accepted():
    # real files may carry an entry count that is not a multiple of 64; the reference
    # overflows there, the build decodes ceil(count/64) records
    img = L.natural_image_np(192, 4, 12, 12.0, 5)  # 12 blocks
    buf = L.encode7(img, None, flags=1)
    ret, out = L.oracle_decode7(buf, 192, 4)
    assert ret == 192 * 4 and np.array_equal(out, img)


def test_height_not_multiple_of_4_is_clipped():
    img = L.natural_image_np(128, 6, 12, 12.0, 6)
    buf = L.encode7(img)
    ret, out = L.oracle_decode7(buf, 128, 6)
    assert ret == 128 * 6 and np.array_equal(out, img)


def test_truncated_legacy_fails():
    img = L.natural_image_np(96, 4, 10, 4.0, 7)
    buf = L.encode6(img)
    ret, out = L.oracle_decode6(buf, 96, 4)
    assert ret == 96 * 4 and np.array_equal(out, img)
    assert L.oracle_decode6(buf[:-1], 96, 4)[0] == 0  # the trailing byte is required (RawData_Legacy.cpp:387,398)
    assert L.oracle_decode6(buf[: buf.size // 2], 96, 4)[0] == 0
