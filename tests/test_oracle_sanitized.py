"""CPU: the oracle rebuilt with AddressSanitizer + UBSan decodes the golden vectors and a pile of
mutated / truncated frames without a single report (GPU sanitizers are unavailable on the pool, so
memory safety of the restatement -- the thing parity is judged against -- is checked here)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import _libs as L
lib = C.CDLL(os.path.join(%(root)r, "oracle", "libmcraw_oracle_asan.so"))
for name in ("mcraw_oracle_decode7", "mcraw_oracle_decode6"):
    fn = getattr(lib, name); fn.restype = C.c_size_t
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
z = np.load(os.path.join(%(root)r, "tests", "golden", "mcraw_golden.npz"))
rng = np.random.default_rng(5)
n = 0
for name in z["names"]:
    name = str(name)
    typ, w, h, ret = (int(v) for v in z[name + "/meta"])
    fn = lib.mcraw_oracle_decode7 if typ == 7 else lib.mcraw_oracle_decode6
    buf = np.ascontiguousarray(z[name + "/buf"])
    out = np.zeros((h, w), np.uint16)
    assert fn(out.ctypes.data, w, h, buf.ctypes.data, buf.size) == ret
    assert np.array_equal(out, z[name + "/out"])
    for trial in range(40):                      # mutants: exact-size heap copies so any overrun is caught
        b = buf.copy()
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, b.size))] = rng.integers(0, 256)
        cut = int(rng.integers(0, 3)) and int(rng.integers(1, b.size))
        b = np.ascontiguousarray(b[:b.size - cut]) if cut else b
        out = np.zeros((h, w), np.uint16)
        fn(out.ctypes.data, w, h, b.ctypes.data, b.size)
        n += 1
# the post stage on exact-size buffers (odd widths: the last byte of a 12-bit row is half padding)
lib.mcraw_oracle_post.restype = C.c_size_t
lib.mcraw_oracle_post.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_void_p]
black = np.array([64, 65, 66, 4000], np.uint16)
for (w, h) in ((1, 1), (2, 3), (13, 5), (64, 4), (101, 7)):
    img = rng.integers(0, 65536, size=(h, w), dtype=np.uint16)
    for flags in (0, 1, 2, 3):
        rb = (w * 12 + 7) // 8 if flags & 2 else 2 * w
        out = np.zeros(h * rb, np.uint8)
        assert lib.mcraw_oracle_post(out.ctypes.data, img.ctypes.data, w, h, flags, black.ctypes.data) == out.size
        assert np.array_equal(out.reshape(h, rb), L.post_np(img, black if flags & 1 else None, bool(flags & 2)))
        n += 1
print("sanitized ok", n)
'''


def _asan_runtime():
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_asan_runtime() is None, reason="libasan not available")
def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    env = dict(os.environ, LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "sanitized ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
