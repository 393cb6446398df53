"""CPU: the oracle (own C restatement) against the golden vectors produced by the
real reference (tests/golden/make_golden.py) -- this is what pins the oracle."""
import numpy as np
import pytest

import _libs as L


def test_golden_covers_every_storage_class(golden):
    assert {"t7_bits%02d" % b for b in range(17)} <= set(golden)
    assert {"t6_bits%02d" % b for b in range(16)} <= set(golden)
    assert len(golden) >= 40


def test_oracle_matches_reference_vectors(golden):
    for name, c in golden.items():
        fn = L.oracle_decode7 if c["type"] == 7 else L.oracle_decode6
        ret, out = fn(c["buf"], c["w"], c["h"])
        assert ret == c["ret"], name
        assert np.array_equal(out, c["out"]), name


@pytest.mark.parametrize("bits", range(17))
def test_block7_unpack_inverts_pack(bits):
    rng = np.random.default_rng(bits)
    v = rng.integers(0, 1 << bits if bits else 1, 64, dtype=np.uint16) if bits < 11 else rng.integers(0, 1 << 16, 64, dtype=np.uint16)
    buf = np.zeros(128, np.uint8)
    n = L.synth().mcraw_synth_pack_block7(L._ptr(buf), bits, L._ptr(v))
    out = np.zeros(64, np.uint16)
    m = L.oracle().mcraw_oracle_block7(L._ptr(out), bits, L._ptr(buf))
    assert n == m == [0, 8, 16, 24, 32, 40, 48, 64, 64, 80, 80, 128, 128, 128, 128, 128, 128][bits]
    assert np.array_equal(out, v)
