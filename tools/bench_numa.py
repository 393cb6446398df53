#!/usr/bin/env python3
"""Where the facade's threads run against the GPU stage of loadFrames: mcraw_export --no-write on a synthetic 240-frame UHD clip
under taskset masks (one line per setting: total, set-up, pipeline and its parts)."""
import os, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _libs as L
n = int(os.environ.get("NFRAMES", "240"))
print(subprocess.run("lscpu | grep -i 'numa\\|socket\\|^CPU(s)'; rocm-smi --showtoponuma 2>/dev/null | grep -i numa", shell=True, capture_output=True, text=True).stdout)
d = tempfile.mkdtemp(dir="/dev/shm")
pairs = [L.encode7(L.synth_image(3840, 2160, 12, 1, 12.0, 3000 + i)) for i in range(8)]
path = L.write_mcraw(os.path.join(d, "uhd.mcraw"), [(1000 + i, 7, 3840, 2160, pairs[i % 8]) for i in range(n)])
exe = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "mcraw_export")
ncpu = os.cpu_count()
masks = ["", "0-%d" % (ncpu // 4 - 1), "%d-%d" % (ncpu // 4, ncpu // 2 - 1), "0-%d" % (ncpu // 2 - 1), "0-15", "%d-%d" % (ncpu // 4, ncpu // 4 + 15)]
for rep in range(2):
    for mode in ([], ["--pinned"]):
        for m in masks:
            cmd = (["taskset", "-c", m] if m else []) + [exe, path, "-o", d, "--no-write"] + mode
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MCRAW_TRACE="1"))
            tr = [l for l in r.stderr.splitlines() if l.startswith("[mcraw]")]
            print("%-10s %-12s" % (" ".join(mode), m or "all"), tr[-1][18:] if tr else r.stderr[-200:], flush=True)
os.remove(path)
