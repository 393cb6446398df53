#!/bin/bash
# gpurun -- 'bash tools/k6_prof_occ.sh': stage times of k6_decode at 1, 2, 4 and 6 workgroups per CU (extra LDS; diag build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
for pad in 100000 54900 13900 0; do echo "== pad $pad"; MCRAW_K6_LDSPAD=$pad NB=12 DIST=1 MCRAW_LIB_PATH=/tmp/libmcraw_diag.so python3 $R/tools/k6_prof.py 2>&1 | grep -v "amdgpu.ids"; done
