#!/bin/bash
# A/B two builds of libmcraw_hip.so on ONE box, interleaved rounds (guide rule 24):
#   lib/libmcraw_hip_prev.so (baseline) vs lib/libmcraw_hip.so (candidate)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${AB_N:-3}); do for v in prev cur; do
  if [ $v = prev ]; then export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_prev.so; else unset MCRAW_LIB_PATH; fi
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu ${AB_ARGS:-} 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernels_ms_bracketed'], d['roofline']['avg_launch_ms'], d.get('also_u',{}).get('ms_per_step'), d['bit_exact'])"
done; done
