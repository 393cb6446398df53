#!/bin/bash
# gpurun -- 'bash tools/ab_lib.sh name1 name2 ..': bench.py's headline leg with lib/libmcraw_hip_<name>.so ("cur": lib/libmcraw_hip.so),
# interleaved fresh processes on one box (AB_N rounds); AB_ARGS: more arguments for bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${AB_N:-3}); do for v in "$@"; do
  if [ $v = cur ]; then unset MCRAW_LIB_PATH; else export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_$v.so; fi
  python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-pcie ${AB_ARGS:-} 2>/tmp/err.txt | grep "^{" > /tmp/line.json
  python3 - "$v" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/line.json"))
    u = d.get("also_u", {})
    print("%-8s" % sys.argv[1], "ms_per_step", d["ms_per_step"], "tiles_avg", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "xcd", d["roofline"]["xcd_runs"],
          "| U step", u.get("ms_per_step"), "tiles", u.get("avg_launch_ms"), "ok", d["bit_exact"], u.get("bit_exact"), flush=True)
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/err.txt").read()[-600:], flush=True)
PY
done; done
