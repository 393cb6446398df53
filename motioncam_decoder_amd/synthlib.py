"""The build's own encoder and image generator (motioncam_decoder_amd/synth/mcraw_synth.c) behind ctypes.

The reference has no encoder and no sample clip: inputs for tests, tools and bench.py come from this encoder,
which is validated by the real reference decoding its output (tests/test_oracle_vs_reference.py).  CPU only; nothing here
decodes anything.
"""
import ctypes as C
import os
import subprocess

import numpy as np

SYNTH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth")

_synth = None


def synth():
    global _synth
    if _synth is None:
        p = os.path.join(SYNTH_DIR, "libmcraw_synth.so")
        if not os.path.exists(p):
            subprocess.run(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-std=c11", "-shared", "-o", p,
                            os.path.join(SYNTH_DIR, "mcraw_synth.c"), "-lm"], check=True)
        lib = C.CDLL(p)
        lib.mcraw_synth_bound7.restype = C.c_size_t
        lib.mcraw_synth_bound7.argtypes = [C.c_int, C.c_int]
        lib.mcraw_synth_bound6.restype = C.c_size_t
        lib.mcraw_synth_bound6.argtypes = [C.c_int, C.c_int]
        for name in ("mcraw_synth_encode7", "mcraw_synth_encode6"):
            fn = getattr(lib, name)
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.mcraw_synth_image.restype = None
        lib.mcraw_synth_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                          C.c_uint64]
        lib.mcraw_synth_pack_block7.restype = C.c_int
        lib.mcraw_synth_pack_block7.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _synth = lib
    return _synth


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def encode7(img, min_bits=None, flags=0):
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    s = synth()
    buf = np.zeros(s.mcraw_synth_bound7(w, h), dtype=np.uint8)
    mb = None if min_bits is None else np.ascontiguousarray(min_bits, dtype=np.uint8)
    n = s.mcraw_synth_encode7(_ptr(buf), buf.size, _ptr(img), w, h, _ptr(mb), flags)
    assert n > 0
    return buf[:n].copy()


def encode6(img, min_bits=None, flags=0):
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    s = synth()
    buf = np.zeros(s.mcraw_synth_bound6(w, h), dtype=np.uint8)
    mb = None if min_bits is None else np.ascontiguousarray(min_bits, dtype=np.uint8)
    n = s.mcraw_synth_encode6(_ptr(buf), buf.size, _ptr(img), w, h, _ptr(mb), flags)
    assert n > 0
    return buf[:n].copy()


def synth_image(w, h, nbits, dist, sigma, seed):
    img = np.empty((h, w), dtype=np.uint16)
    synth().mcraw_synth_image(_ptr(img), w, h, nbits, dist, float(sigma), seed)
    return img


def _strip_bits(pack12, bits):
    """Bits per sample of a strip row: `bits` (10, 12, 14 or 16/None), or 12 for the older pack12=True."""
    b = int(bits) if bits else (12 if pack12 else 16)
    assert b in (10, 12, 14, 16)
    return b


def post_row_bytes(w, pack12=False, bits=None):
    return (w * _strip_bits(pack12, bits) + 7) // 8


def natural_image_np(w, h, nbits, sigma, seed):
    """SURVEY 8(d) "Nat" distribution, numpy flavour (used for golden vectors)."""
    rng = np.random.default_rng(seed)
    maxv = (1 << nbits) - 1
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    field = 0.8 * maxv * (0.5 + 0.45 * np.sin(x / 211.0) * np.cos(y / 173.0)) + maxv / 16.0
    img = field + rng.normal(0.0, sigma, size=(h, w))
    return np.clip(np.rint(img), 0, maxv).astype(np.uint16)


def uniform_image_np(w, h, nbits, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 1 << nbits, size=(h, w), dtype=np.uint16)


# ---------------------------------------------------------------- .mcraw container writer

def write_mcraw(path, frames, audio_chunks=(), camera_extra=None, audio_rate=48000, audio_channels=2):
    """Write a synthetic .mcraw container (layout: SURVEY Appendix A.5).

    frames: list of (timestamp, type, width, height, encoded bytes), written in the given order
    (the reader sorts by timestamp).  audio_chunks: list of (timestamp_ns or None, int16 array).
    """
    import json
    import struct
    camera = {"blackLevel": [64, 64, 64, 64], "whiteLevel": 1023.0, "sensorArrangment": "rggb",
              "colorMatrix1": [1, 0, 0, 0, 1, 0, 0, 0, 1], "colorMatrix2": [1, 0, 0, 0, 1, 0, 0, 0, 1],
              "forwardMatrix1": [1, 0, 0, 0, 1, 0, 0, 0, 1], "forwardMatrix2": [1, 0, 0, 0, 1, 0, 0, 0, 1],
              "extraData": {"audioSampleRate": audio_rate, "audioChannels": audio_channels}}
    if camera_extra:
        camera.update(camera_extra)
    BUFFER_INDEX, BUFFER_INDEX_DATA, BUFFER, METADATA, AUDIO_INDEX, AUDIO_DATA, AUDIO_DATA_METADATA = range(7)

    def item(t, size):
        return struct.pack("<II", t, size)

    out = bytearray(b"MOTION " + bytes([3]))
    cj = json.dumps(camera).encode()
    out += item(METADATA, len(cj)) + cj
    offsets = []
    for ts, typ, w, h, buf in frames:
        offsets.append((len(out), ts))
        b = bytes(np.ascontiguousarray(buf, dtype=np.uint8))
        out += item(BUFFER, len(b)) + b
        fj = json.dumps({"width": w, "height": h, "compressionType": typ, "asShotNeutral": [1.0, 1.0, 1.0],
                         "timestamp": str(ts)}).encode()
        out += item(METADATA, len(fj)) + fj
    audio_offsets = []
    for ts, samples in audio_chunks:
        audio_offsets.append((len(out), ts if ts is not None else -1))
        b = np.ascontiguousarray(samples, dtype=np.int16).tobytes()
        out += item(AUDIO_DATA, len(b)) + b
        if ts is not None:
            out += item(AUDIO_DATA_METADATA, 8) + struct.pack("<q", ts)
    if audio_offsets:
        out += item(AUDIO_INDEX, 16 + 16 * len(audio_offsets)) + struct.pack("<qq", len(audio_offsets), 0)
        for off, ts in audio_offsets:
            out += struct.pack("<qq", off, ts)
    out += item(BUFFER_INDEX_DATA, 16 * len(offsets))
    index_data_offset = len(out)
    for off, ts in offsets:
        out += struct.pack("<qq", off, ts)
    out += item(BUFFER_INDEX, 16) + struct.pack("<iiq", np.int32(np.uint32(0x8A905612)), len(offsets), index_data_offset)
    with open(path, "wb") as f:
        f.write(out)
    return path
