#!/bin/bash
# The host-memory pipeline on the GPU's own clock, without a profiler: a library built with -DMCRAW_TIMELINE (events with
# timing; one line per sub-batch when it is drained: queued / upload from-to / decoded / downloaded, ms since the first) under
# mcraw_export.  Build here first (the built library travels to the GPU box with the snapshot):
#   python3 -c "from motioncam_decoder_amd import build; build.build_timeline()"
#   gpurun -- 'ARGS="--pinned" bash tools/timeline_host.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R && python3 - <<'PY'
import os, sys
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _libs as L
pairs = [L.encode7(L.synth_image(3840, 2160, 12, 1, 12.0, 3000 + i)) for i in range(4)]
L.write_mcraw("/dev/shm/uhd_tl.mcraw", [(1000 + i, 7, 3840, 2160, pairs[i % 4]) for i in range(int(os.environ.get("NFRAMES", "120")))])
PY
LD_LIBRARY_PATH=$R/motioncam_decoder_amd/lib/timeline:$LD_LIBRARY_PATH MCRAW_TRACE=2 $R/motioncam_decoder_amd/lib/mcraw_export /dev/shm/uhd_tl.mcraw -o /tmp --no-write ${ARGS:-} 2>&1 | grep "^\[tl\]\|^\[mcraw" | tail -${LAST:-60}
rm -f /dev/shm/uhd_tl.mcraw
