#!/usr/bin/env python3
"""End-to-end timing of the C++ facade (mcraw_export --no-write) on a synthetic UHD .mcraw."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _libs as L
n = int(os.environ.get("NFRAMES", "48"))
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
pairs = [L.encode7(L.synth_image(3840, 2160, 12, 1, 12.0, 3000 + i)) for i in range(8)]
path = L.write_mcraw(os.path.join(d, "uhd.mcraw"), [(1000 + i, 7, 3840, 2160, pairs[i % 8]) for i in range(n)])
exe = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "mcraw_export")
for mode in ([], ["--single"], ["--single", "--reuse"], [], ["--pinned"], ["--pinned", "--bits", "12"]):
    r = subprocess.run([exe, path, "-o", d, "--no-write"] + mode, capture_output=True, text=True, env=dict(os.environ, MCRAW_TRACE="1"))
    tr = [l for l in r.stderr.splitlines() if l.startswith("[mcraw]")]
    print(mode, [l for l in r.stdout.splitlines() if l.startswith("decoded") or l.startswith("pinned")], tr[:1], tr[-1:] if len(tr) > 1 else "")
    if "--single" in mode and len(tr) > 2:  # per-frame calls: the means over every call but the first (which makes the context)
        import re
        rows = [[float(x) for x in re.findall(r"(?:pipeline|wait-read|gpu batch|wait-copy|tail copy) ([0-9.]+)", l)] for l in tr[1:]]
        mean = [sum(c) / len(c) for c in zip(*rows)]
        print("   per call, mean of %d: pipeline %.3f ms = wait-read %.3f + gpu batch %.3f + wait-copy %.3f + tail copy %.3f"
              % ((len(rows),) + tuple(mean)))
os.remove(path)
