#!/bin/bash
# gpurun -- 'AB_ENVS="A=1;B=2 C=3" bash tools/ab_env.sh': bench.py's headline leg under several environments (entries separated by
# blanks, variables inside an entry by ';'), interleaved on one box.  AB_ARGS: more arguments for bench.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${AB_N:-2}); do for e in ${AB_ENVS:-X=0}; do
  ( IFS=';'; for kv in $e; do export "$kv"; done; unset IFS
    python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-also --no-pcie ${AB_ARGS:-} 2>/tmp/err.txt | grep "^{" > /tmp/line.json )
  python3 - "$e" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/line.json"))
    print(sys.argv[1], "ms_per_step", d["ms_per_step"], "step_frac", d["roofline"]["step_frac"], "tiles_avg", d["roofline"]["avg_launch_ms"],
          "frac", d["roofline"]["frac"], d.get("kernels_ms_per_step") or d.get("kernels_ms_bracketed"), "ok", d["bit_exact"], flush=True)
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/err.txt").read()[-600:], flush=True)
PY
done; done
