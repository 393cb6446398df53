#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "--dist u" "--width 4032 --height 3024 --frames 96" "--config 5" "--width 1920 --height 1080 --frames 480"; do
for i in 1 2; do for C in 0 64 128; do
  MCRAW_XCD_CHUNK=$C python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-also --no-pcie $cfg 2>/dev/null | grep "^{" > /tmp/line.json
  python3 - <<PY
import json
d = json.load(open("/tmp/line.json"))
print("$cfg chunk $C", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["bit_exact"])
PY
done; done; done
