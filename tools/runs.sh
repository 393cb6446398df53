#!/bin/bash
# gpurun -- 'bash tools/runs.sh': the bench's headline leg in N fresh processes on one box (run-to-run spread, box yardstick beside it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${N:-8}); do
  MCRAW_BENCH_DEBUG=1 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-also --no-pcie ${RUN_ARGS:-} 2>/tmp/err.txt | grep "^{" > /tmp/line.json
  python3 - <<'PY'
import json
d = json.load(open("/tmp/line.json"))
print(d["ms_per_step"], d["kernels_ms_bracketed"], d["roofline"]["avg_launch_ms"], d.get("box_calibration"))
PY
  grep buffers /tmp/err.txt
done
