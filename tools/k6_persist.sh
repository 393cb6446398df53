#!/bin/bash
# gpurun -- 'bash tools/k6_persist.sh': k6_decode with n resident workgroups per CU taking segment after segment (MCRAW_K6_WGS_PER_CU;
# 0: one workgroup per segment), against lib/libmcraw_hip_prev.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  echo prev $(MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_prev.so python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-150)
  for n in 0 4 5 6; do
    echo per_cu $n $(MCRAW_K6_WGS_PER_CU=$n python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-150)
  done
  echo default $(python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-150)
done
