#!/bin/bash
# gpurun -- 'bash tools/ab_thin.sh': k7_side variants (workgroup size / piece size, built as lib/libmcraw_hip_<name>.so) in line
# (MCRAW_SIDE_CUS=0) and on a stream of their own beside the previous batch's tile kernel (-1: no CU partition; r: r CUs per XCD)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${AB_N:-2}); do for v in ${AB_LIBS:-fat t256l2 t256l3 t256l4}; do for m in ${AB_MASKS:--1}; do
  if [ $v = fat ]; then unset MCRAW_LIB_PATH; else export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_$v.so; fi
  MCRAW_SIDE_CUS=$m python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-also --no-pcie ${AB_ARGS:-} 2>/tmp/err.txt | grep "^{" > /tmp/line.json
  python3 - "$v" "$m" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/line.json"))
    print(sys.argv[1], "side_cus", sys.argv[2], "ms_per_step", d["ms_per_step"], "step_frac", d["roofline"]["step_frac"], "tiles_avg", d["roofline"]["avg_launch_ms"],
          "frac", d["roofline"]["frac"], d["kernels_ms_bracketed"], "ok", d["bit_exact"], flush=True)
except Exception as e:
    print(sys.argv[1], sys.argv[2], "failed", e, open("/tmp/err.txt").read()[-600:], flush=True)
PY
done; done; done
