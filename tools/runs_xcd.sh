#!/bin/bash
# gpurun -- 'bash tools/runs_xcd.sh': the bench's headline leg in fresh processes, k7_tiles dealt to the XCDs in runs of
# different lengths (MCRAW_XCD_CHUNK: 0 = the grid in eight parts, 1 = plain blockIdx order)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${N:-4}); do for C in ${CHUNKS:-0 1 64 2048}; do
  MCRAW_XCD_CHUNK=$C python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-also --no-pcie 2>/dev/null | grep "^{" > /tmp/line.json
  python3 - <<PY
import json
d = json.load(open("/tmp/line.json"))
print("chunk $C", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["box_calibration"]["before"], d["bit_exact"])
PY
done; done
