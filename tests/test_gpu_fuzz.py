"""GPU (-m gpu): mutated frames.  Random byte flips anywhere in valid frames (payload, side
streams, headers, legacy records).  Whatever the oracle makes of a mutant -- decodes it to some
pixels, or rejects it -- the HIP path must agree: same pixels and return value, or a nonzero
status with 0 written.  Nothing may fault, and clean frames batched with mutants still decode."""
import numpy as np
import pytest

import _libs as L

pytestmark = pytest.mark.gpu


def _mutants(buf, rng, n, hot=()):
    out = []
    for i in range(n):
        b = buf.copy()
        k = int(rng.integers(1, 4))
        for _ in range(k):
            if hot and rng.random() < 0.5:
                lo, hi = hot[int(rng.integers(0, len(hot)))]
                pos = int(rng.integers(lo, min(hi, b.size)))
            else:
                pos = int(rng.integers(0, b.size))
            b[pos] = rng.integers(0, 256)
        out.append(b)
    return out


def _compare(ctx, typ, w, h, bufs, decode):
    from _gpu import decode_batch_device
    written, status, outs = decode_batch_device(ctx, [(typ, w, h, b) for b in bufs], fill=0)
    n_ok = 0
    for i, b in enumerate(bufs):
        ret, want = decode(b, w, h)
        if ret == 0:
            assert status[i] != 0 and written[i] == 0, (i, status[i], written[i])
        else:
            assert status[i] == 0 and written[i] == ret, (i, status[i], written[i], ret)
            rows = ret // w  # a frame coded shorter than `height` leaves the rows below untouched
            assert np.array_equal(outs[i][:rows], want[:rows]), i
            n_ok += 1
    return n_ok


def test_type7_mutants(gpu_ctx):
    rng = np.random.default_rng(77)
    total_ok = 0
    for (w, h, seed) in ((256, 32, 1), (200, 12, 2), (640, 64, 3)):
        img = L.natural_image_np(w, h, 12, 12.0, seed)
        encW, encH = (w + 63) // 64 * 64, (h + 3) // 4 * 4
        mb = rng.integers(0, 17, encW * encH // 64).astype(np.uint8)
        for buf in (L.encode7(img), L.encode7(img, mb)):
            bits_off = int(np.frombuffer(buf[8:12].tobytes(), np.uint32)[0])
            hot = [(0, 16), (bits_off, buf.size)]  # header and side streams
            bufs = [buf] + _mutants(buf, rng, 60, hot) + [buf]
            total_ok += _compare(gpu_ctx, 7, w, h, bufs, L.oracle_decode7)
    assert total_ok > 50  # plenty of mutants still decode (payload flips) -- and bit-exactly so


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_type7_side_stream_mutants_many_chunks(gpu_ctx, seed):
    # frames whose side streams span many transition-map chunks; mutations confined to the streams
    # (record headers change the chain, references and bits change the pixels or fail the frame)
    rng = np.random.default_rng(700 + seed)
    w, h = 1920, 1080
    img = L.synth_image(w, h, 12, 1, 12.0, 9000 + seed)
    buf = L.encode7(img)
    bits_off = int(np.frombuffer(buf[8:12].tobytes(), np.uint32)[0])
    bufs = [buf] + _mutants(buf, rng, 40, [(bits_off, buf.size)] * 3) + [buf]
    # structured header mutants: offsets nudged, geometry changed
    for off, delta in ((8, 1), (8, -2), (12, 1), (12, -7), (12, 2), (0, 64), (4, 4), (4, -4)):
        b = buf.copy()
        v = int(np.frombuffer(b[off:off + 4].tobytes(), np.uint32)[0]) + delta
        b[off:off + 4] = np.frombuffer(np.uint32(v).tobytes(), np.uint8)
        bufs.append(b)
    n_ok = _compare(gpu_ctx, 7, w, h, bufs, L.oracle_decode7)
    assert n_ok >= 2


def test_type6_mutants(gpu_ctx):
    rng = np.random.default_rng(66)
    total_ok = 0
    for (w, h, seed) in ((160, 24, 1), (75, 9, 2), (1000, 16, 3)):
        img = L.natural_image_np(w, h, 12, 12.0, seed)
        buf = L.encode6(img, None, flags=seed & 1)
        bufs = [buf] + _mutants(buf, rng, 80) + [buf]
        total_ok += _compare(gpu_ctx, 6, w, h, bufs, L.oracle_decode6)
    assert total_ok > 20
