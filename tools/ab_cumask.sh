#!/bin/bash
# gpurun -- 'bash tools/ab_cumask.sh': the bench's headline leg with r CUs of every XCD reserved for k7_side (MCRAW_SIDE_CUS=r; 0: no
# partition), interleaved on one box.  AB_MASKS="0 1 2 3 4 6", AB_N rounds, AB_ARGS for bench.py (e.g. "--config 5").
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${AB_N:-2}); do for m in ${AB_MASKS:-0 1 2 3 4 6}; do
  MCRAW_SIDE_CUS=$m python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-also --no-pcie ${AB_ARGS:-} 2>/tmp/err.txt | grep "^{" > /tmp/line.json
  python3 - "$m" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/line.json"))
    print("side_cus/xcd", sys.argv[1], "ms_per_step", d["ms_per_step"], "step_frac", d["roofline"]["step_frac"], "tiles_avg", d["roofline"]["avg_launch_ms"],
          "frac", d["roofline"]["frac"], d["kernels_ms_bracketed"], "ok", d["bit_exact"], flush=True)
except Exception as e:
    print("side_cus/xcd", sys.argv[1], "failed", e, open("/tmp/err.txt").read()[-600:], flush=True)
PY
done; done
