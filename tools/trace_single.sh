#!/bin/bash
# gpurun -- 'bash tools/trace_single.sh': the GPU-side timeline (copies and kernels, rocprofv3 traces) of the facade's per-frame loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_single; rm -rf $OUT; mkdir -p $OUT
cd $R && python3 - <<'PY'
import os, sys
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _libs as L
pairs = [L.encode7(L.synth_image(3840, 2160, 12, 1, 12.0, 3000 + i)) for i in range(4)]
L.write_mcraw("/dev/shm/uhd60.mcraw", [(1000 + i, 7, 3840, 2160, pairs[i % 4]) for i in range(int(os.environ.get("NFRAMES", "60")))])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- $R/motioncam_decoder_amd/lib/mcraw_export /dev/shm/uhd60.mcraw -o /tmp --no-write ${ARGS---single --reuse} > $OUT/run.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.getcwd(), "gpurun_out", "trace_single")
ev = []
for f in glob.glob(out + "/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "copy"))))
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
ev.sort()
t0 = ev[0][0]
big = [e for e in ev if e[1] - e[0] > 20000 or "k7" in e[2]]
for s, e, n in big[-int(os.environ.get("LAST", "40")):]:
    print("%10.3f ms  +%7.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
# how busy each lane was between the first and the last event of the run's second half
half = big[len(big) // 2:]
span = (half[-1][1] - half[0][0]) / 1e6
for key in ("HOST_TO_DEVICE", "DEVICE_TO_HOST", "k7_side", "k7_tiles"):
    busy = sum(e - s for s, e, n in half if key in n) / 1e6
    print("second half: %-16s busy %8.3f of %8.3f ms (%.0f %%), %d events" % (key, busy, span, 100 * busy / span, sum(1 for x in half if key in x[2])))
PY
rm -f /dev/shm/uhd60.mcraw
