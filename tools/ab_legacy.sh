#!/bin/bash
# A/B two builds of libmcraw_hip.so on ONE box, legacy path (tools/bench_legacy.py), interleaved rounds:
#   lib/libmcraw_hip_prev.so (baseline) vs lib/libmcraw_hip.so (candidate)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in prev cur; do
  if [ $v = prev ]; then export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_prev.so; else unset MCRAW_LIB_PATH; fi
  python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k:(v['ms_per_batch'],v['kernels_ms']) for k,v in d.items()})"
done; done
