#!/usr/bin/env python3
"""The per-frame loop of the facade (mcraw_export --single --reuse: the reference example's loop) with its trace lines: the
last calls in full, and the means of every part."""
import os, re, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _libs as L
n = int(os.environ.get("NFRAMES", "120"))
d = tempfile.mkdtemp(dir="/dev/shm")
pairs = [L.encode7(L.synth_image(3840, 2160, 12, 1, 12.0, 3000 + i)) for i in range(8)]
path = L.write_mcraw(os.path.join(d, "uhd.mcraw"), [(1000 + i, 7, 3840, 2160, pairs[i % 8]) for i in range(n)])
exe = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "mcraw_export")
for rep in range(int(os.environ.get("REPS", "3"))):
    for mode in (["--single", "--reuse"], ["--single"]):
        r = subprocess.run([exe, path, "-o", d, "--no-write"] + mode, capture_output=True, text=True, env=dict(os.environ, MCRAW_TRACE="1"))
        tr = [l for l in r.stderr.splitlines() if l.startswith("[mcraw]")]
        rows = [[float(x) for x in re.findall(r" ([0-9]+\.[0-9]+)", l)] for l in tr[3:]]
        width = min(len(x) for x in rows)
        print(" ".join(mode), [l for l in r.stdout.splitlines() if l.startswith("decoded")])
        print("   means over %d calls:" % len(rows), ["%.3f" % (sum(x[i] for x in rows) / len(rows)) for i in range(width)])
        if rep == 0:
            for l in tr[:4] + tr[-2:]:
                print("   ", l[:260])
os.remove(path)
