#!/usr/bin/env python3
"""The facade's staging slot size (MCRAW_SLOT_MB) against pipeline and set-up time: mcraw_export --no-write on a synthetic 240-frame UHD clip."""
import os, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _libs as L
n = 240
d = tempfile.mkdtemp(dir="/dev/shm")
pairs = [L.encode7(L.synth_image(3840, 2160, 12, 1, 12.0, 3000 + i)) for i in range(8)]
path = L.write_mcraw(os.path.join(d, "uhd.mcraw"), [(1000 + i, 7, 3840, 2160, pairs[i % 8]) for i in range(n)])
exe = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "mcraw_export")
for mb in (os.environ.get("SLOTS") or "96 192 384 768 192 384").split():
    for rep in range(2):
        r = subprocess.run([exe, path, "-o", d, "--no-write"], capture_output=True, text=True, env=dict(os.environ, MCRAW_TRACE="1", MCRAW_SLOT_MB=mb))
        tr = [l for l in r.stderr.splitlines() if l.startswith("[mcraw]")]
        print(mb, [l.split(" s (")[0] for l in r.stdout.splitlines() if l.startswith("decoded")], tr[-1][18:] if tr else "", flush=True)
os.remove(path)
