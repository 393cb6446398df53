#!/usr/bin/env python3
"""Latency of the five-argument drop-in entry points (mcraw_decode7 / mcraw_decode6: pageable host
pointers in and out, synchronous), next to the reference codec on one CPU thread."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401  (one HIP runtime for the process)

import _libs as L
import motioncam_decoder_amd as M


def main():
    lib = M.load()
    res = {}
    for name, w, h, nb, sig in (("4032x3024_12bit", 4032, 3024, 12, 12.0), ("1920x1080_10bit", 1920, 1080, 10, 4.0)):
        img = L.synth_image(w, h, nb, 1, sig, 77)
        for typ, enc, fn in ((7, L.encode7, lib.mcraw_decode7), (6, L.encode6, lib.mcraw_decode6)):
            buf = enc(img)
            out = np.zeros((h, w), np.uint16)
            for _ in range(3):
                r = fn(out.ctypes.data, w, h, buf.ctypes.data, buf.size)
            assert r == w * h and np.array_equal(out, img)
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                fn(out.ctypes.data, w, h, buf.ctypes.data, buf.size)
            t = (time.perf_counter() - t0) / reps
            e = {"ms": round(t * 1e3, 3), "in_MB": round(buf.size / 1e6, 2), "out_MB": round(out.nbytes / 1e6, 2),
                 "GBs_moved": round((buf.size + out.nbytes) / t / 1e9, 1)}
            ref = L.ref()
            if ref is not None:
                rfn = ref.mcraw_ref_decode7 if typ == 7 else ref.mcraw_ref_decode6
                o2 = np.zeros((h + 4, w), np.uint16)
                t0 = time.perf_counter()
                for _ in range(5):
                    rfn(o2.ctypes.data, w, h, buf.ctypes.data, buf.size)
                e["reference_cpu_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
            res["%s_type%d" % (name, typ)] = e
    print(json.dumps(res))


if __name__ == "__main__":
    main()
