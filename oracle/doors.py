"""ctypes doors for the checkers (TEST INFRASTRUCTURE: tests/, __graft_entry__.smoke() and bench.py's checking / cpu_baseline
legs only -- the product never loads anything from this directory).

* oracle/libmcraw_oracle.so      -- own scalar C restatement of the reference codecs (oracle/mcraw_oracle.c)
* oracle/_ref/libmcraw_ref_*.so  -- the real reference codec (built only where /root/reference exists; travels to the
                                    GPU box prebuilt)
"""
import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def _ensure(path, cmd, cwd):
    if not os.path.exists(path):
        subprocess.run(cmd, cwd=cwd, check=True, stdout=subprocess.DEVNULL)
    return path


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        p = _ensure(os.path.join(ORACLE_DIR, "libmcraw_oracle.so"), ["make", "-s"], ORACLE_DIR)
        lib = C.CDLL(p)
        for name in ("mcraw_oracle_decode7", "mcraw_oracle_decode6"):
            fn = getattr(lib, name)
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.mcraw_oracle_block7.restype = C.c_int
        lib.mcraw_oracle_block7.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.mcraw_oracle_block6.restype = C.c_int
        lib.mcraw_oracle_block6.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.mcraw_oracle_len_used7.restype = C.c_size_t
        lib.mcraw_oracle_len_used7.argtypes = [C.c_void_p, C.c_size_t]
        lib.mcraw_oracle_post.restype = C.c_size_t
        lib.mcraw_oracle_post.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_void_p]
        lib.mcraw_oracle_time_batch.restype = C.c_double
        lib.mcraw_oracle_time_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_int, C.c_int]
        _oracle = lib
    return _oracle


def ref_path():
    for v in (("v3",) if _cpu_has("avx2") else ()) + ("v2",):
        p = os.path.join(ORACLE_DIR, "_ref", "libmcraw_ref_%s.so" % v)
        if os.path.exists(p):
            return p
    return None


def ref():
    """The real reference codec, or None when it was never built (no /root/reference)."""
    global _ref
    if _ref is None:
        p = ref_path()
        if p is None and os.path.isdir("/root/reference/lib"):
            subprocess.run(["make", "-s", "ref"], cwd=ORACLE_DIR, check=True, stdout=subprocess.DEVNULL)
            p = ref_path()
        if p is None:
            return None
        lib = C.CDLL(p)
        for name in ("mcraw_ref_decode7", "mcraw_ref_decode6"):
            fn = getattr(lib, name)
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.mcraw_ref_time_batch.restype = C.c_double
        lib.mcraw_ref_time_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int]
        _ref = lib
    return _ref


def _decode(fn, buf, w, h, rows_alloc=None, fill=0xA5A5):
    """Call a 5-argument decode entry; returns (ret, out[h_alloc, w])."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    rows = rows_alloc if rows_alloc is not None else h
    out = np.full((rows, w), fill, dtype=np.uint16)
    ret = fn(_ptr(out), w, h, _ptr(buf), buf.size)
    return ret, out


def oracle_decode7(buf, w, h, **kw):
    return _decode(oracle().mcraw_oracle_decode7, buf, w, h, **kw)


def oracle_decode6(buf, w, h, **kw):
    return _decode(oracle().mcraw_oracle_decode6, buf, w, h, **kw)


def ref_decode7(buf, w, h, **kw):
    # the reference writes width*encodedHeight: give it 4 spare rows
    kw.setdefault("rows_alloc", h + 4)
    return _decode(ref().mcraw_ref_decode7, buf, w, h, **kw)


def ref_decode6(buf, w, h, **kw):
    return _decode(ref().mcraw_ref_decode6, buf, w, h, **kw)


def _strip_bits(pack12, bits):
    b = int(bits) if bits else (12 if pack12 else 16)
    assert b in (10, 12, 14, 16)
    return b


def post_row_bytes(w, pack12=False, bits=None):
    return (w * _strip_bits(pack12, bits) + 7) // 8


def oracle_post(img, black=None, pack12=False, bits=None):
    """The post stage (mcraw_ctx_set_post) applied to a decoded mosaic by the oracle: bytes [h, row_bytes]."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    b = _strip_bits(pack12, bits)
    out = np.zeros((h, post_row_bytes(w, bits=b)), dtype=np.uint8)
    bl = np.ascontiguousarray(black if black is not None else [0, 0, 0, 0], dtype=np.uint16)
    flags = (1 if black is not None else 0) | {16: 0, 12: 2, 10: 4, 14: 8}[b]
    n = oracle().mcraw_oracle_post(_ptr(out), _ptr(img), w, h, flags, _ptr(bl))
    assert n == out.size
    return out


def post_np(img, black=None, pack12=False, bits=None):
    """Independent numpy statement of the same stage (checks the oracle's)."""
    v = img.astype(np.int64)
    h, w = v.shape
    nb = _strip_bits(pack12, bits)
    if black is not None:
        b = np.asarray(black, dtype=np.int64).reshape(2, 2)
        v = np.maximum(v - np.tile(b, ((h + 1) // 2, (w + 1) // 2))[:h, :w], 0)
    if nb == 16:
        return v.astype("<u2").view(np.uint8).reshape(h, w * 2)
    v = np.minimum(v, (1 << nb) - 1)
    bits = ((v[:, :, None] >> np.arange(nb - 1, -1, -1)) & 1).astype(np.uint8).reshape(h, w * nb)
    pad = (-bits.shape[1]) % 8
    if pad:
        bits = np.concatenate([bits, np.zeros((h, pad), np.uint8)], axis=1)
    return np.packbits(bits, axis=1)


