#!/bin/bash
# gpurun -- 'bash tools/k6_occ.sh': k6_decode against the number of workgroups a CU holds (extra LDS per workgroup, diag build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $R/motioncam_decoder_amd/csrc/mcraw_abi.hip $R/motioncam_decoder_amd/csrc/mcraw_pool.hip $R/motioncam_decoder_amd/csrc/mcraw_type7.hip $R/motioncam_decoder_amd/csrc/mcraw_type6.hip -lpthread || exit 1
export MCRAW_LIB_PATH=/tmp/libmcraw_diag.so
# 163840 / n - 26996: 6 -> 0, 5 -> 5700, 4 -> 13900, 3 -> 27600, 2 -> 54900, 1 -> 100000
for pad in 0 5700 13900 27600 54900 100000 0; do
  echo pad $pad $(MCRAW_K6_LDSPAD=$pad python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-160)
done
