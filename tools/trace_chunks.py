#!/usr/bin/env python3
"""loadFrames chunk by chunk on the host's clock (MCRAW_TRACE=2): mcraw_export --no-write [--pinned] on a synthetic 240-frame UHD clip."""
import os, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _libs as L
n = int(os.environ.get("NFRAMES", "240"))
d = tempfile.mkdtemp(dir="/dev/shm")
pairs = [L.encode7(L.synth_image(3840, 2160, 12, 1, 12.0, 3000 + i)) for i in range(int(os.environ.get("DISTINCT", "8")))]
path = L.write_mcraw(os.path.join(d, "uhd.mcraw"), [(1000 + i, 7, 3840, 2160, pairs[i % len(pairs)]) for i in range(n)])
exe = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "mcraw_export")
for mode in ([], ["--pinned"]) * int(os.environ.get("REPS", "1")):
    r = subprocess.run([exe, path, "-o", d, "--no-write"] + mode, capture_output=True, text=True, env=dict(os.environ, MCRAW_TRACE="2"))
    tr = [l for l in r.stderr.splitlines() if l.startswith("[mcraw")]
    print(mode)
    for l in (tr[-1:] if os.environ.get("BRIEF") else tr[:6] + tr[-8:]):
        print("   ", l)
os.remove(path)
