#!/bin/bash
# gpurun -- 'bash tools/k6_front_pad.sh': does the LDS a workgroup ALLOCATES (not what it does) change how fast workgroups are launched?
# The front-only build (-DK6_ABL=5, 10.5 KB LDS, six workgroups per CU by waves) with 0 and 16 400 bytes of unused dynamic LDS (still six per CU)
R=${GRAFT_REPO_ROOT:-$(pwd)}
S="$R/motioncam_decoder_amd/csrc/mcraw_abi.hip $R/motioncam_decoder_amd/csrc/mcraw_pool.hip $R/motioncam_decoder_amd/csrc/mcraw_type7.hip $R/motioncam_decoder_amd/csrc/mcraw_type6.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -DK6_ABL=5 -o /tmp/libabl5d.so $S -lpthread || exit 1
export MCRAW_LIB_PATH=/tmp/libabl5d.so
for i in 1 2; do for pad in 0 8000 16400; do
  echo pad $pad $(MCRAW_K6_LDSPAD=$pad MCRAW_NOCHECK=1 python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-160)
done; done
