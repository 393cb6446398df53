#!/bin/bash
# A/B of the legacy path: lib/libmcraw_hip_prev.so vs lib/libmcraw_hip.so, interleaved on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in prev cur; do
  if [ $v = prev ]; then export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_prev.so; else unset MCRAW_LIB_PATH; fi
  echo $v $(python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-200)
done; done
