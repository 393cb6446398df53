"""CPU: oracle == real reference on randomized frames (skipped where the reference
build is absent, e.g. no /root/reference and no prebuilt oracle/_ref)."""
import numpy as np
import pytest

import _libs as L

pytestmark = pytest.mark.skipif(L.ref() is None, reason="oracle/_ref not built")


@pytest.mark.parametrize("w,h", [(128, 8), (256, 64), (192, 12), (100, 8), (77, 4), (64, 4), (4032 // 4, 3024 // 4)])
def test_type7_random(w, h):
    rng = np.random.default_rng(w * 131 + h)
    encW, encH = (w + 63) // 64 * 64, (h + 3) // 4 * 4
    for trial in range(6):
        nb = int(rng.integers(1, 17))
        img = rng.integers(0, 1 << nb, size=(h, w), dtype=np.uint16)
        mb = rng.integers(0, 17, size=encW * encH // 64).astype(np.uint8) if trial % 2 else None
        buf = L.encode7(img, mb)
        ro, oo = L.oracle_decode7(buf, w, h)
        rr, orr = L.ref_decode7(buf, w, h)
        assert ro == rr == w * encH
        assert np.array_equal(oo, orr[:h])
        assert np.array_equal(oo, img)


@pytest.mark.parametrize("w,h", [(96, 4), (80, 6), (75, 5), (256, 16), (33, 3), (1920 // 2, 1080 // 4)])
def test_type6_random(w, h):
    rng = np.random.default_rng(w * 17 + h)
    for trial in range(6):
        nb = int(rng.integers(1, 17))
        img = rng.integers(0, 1 << nb, size=(h, w), dtype=np.uint16)
        nrec = ((w + 31) // 32) * 2 * h
        mb = rng.integers(0, 16, size=nrec).astype(np.uint8) if trial % 2 else None
        buf = L.encode6(img, mb, flags=trial & 1)
        ro, oo = L.oracle_decode6(buf, w, h)
        rr, orr = L.ref_decode6(buf, w, h)
        assert ro == rr == w * h
        assert np.array_equal(oo, orr)
        assert np.array_equal(oo, img)


@pytest.mark.parametrize("w,h,nb,dist,sig", [
    (1920, 1080, 10, 0, 0), (1920, 1080, 10, 1, 4),          # BASELINE config 1, both distributions
    (4032, 3024, 12, 1, 12), (4032, 3024, 12, 0, 0),         # config 2: Nat and U
    (3840, 2160, 12, 1, 12), (3840, 2160, 12, 0, 0),         # config 3's frame (the bench line's): Nat and U
    (4032, 3024, 14, 0, 0),                                  # config 4's 14-bit frames
    (7680, 4320, 12, 1, 12),                                 # config 5's frame
])
def test_full_size_type7(w, h, nb, dist, sig):
    img = L.synth_image(w, h, nb, dist, sig, 1000 + dist)
    buf = L.encode7(img)
    ro, oo = L.oracle_decode7(buf, w, h)
    rr, orr = L.ref_decode7(buf, w, h)
    assert ro == rr == w * h
    assert np.array_equal(oo, orr[:h]) and np.array_equal(oo, img)


@pytest.mark.parametrize("w,h,nb,dist,sig,flags", [
    (4000, 3000, 12, 1, 12, 0),                              # the bench's legacy leg; width % 32 == 0
    (4000, 3000, 14, 0, 0, 1),                               # config 4's 14-bit legacy frames, with a trailer
    (1920, 1080, 10, 1, 4, 0), (1920, 1080, 10, 0, 0, 1),    # config 1's frame as legacy
    (4032 - 5, 3024 // 4, 12, 1, 12, 1),                     # a padded width at full row length
])
def test_full_size_legacy(w, h, nb, dist, sig, flags):
    img = L.synth_image(w, h, nb, dist, sig, 2000 + dist)
    buf = L.encode6(img, None, flags)
    ro, oo = L.oracle_decode6(buf, w, h)
    rr, orr = L.ref_decode6(buf, w, h)
    assert ro == rr == w * h
    assert np.array_equal(oo, orr) and np.array_equal(oo, img)
