#!/bin/bash
# gpurun -- 'bash tools/k6_replay.sh': diag build of the library into /tmp, k6_decode with its maps walked / read back
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $R/motioncam_decoder_amd/csrc/mcraw_abi.hip $R/motioncam_decoder_amd/csrc/mcraw_pool.hip $R/motioncam_decoder_amd/csrc/mcraw_type7.hip $R/motioncam_decoder_amd/csrc/mcraw_type6.hip -lpthread || exit 1
MCRAW_LIB_PATH=/tmp/libmcraw_diag.so python3 $R/tools/k6_replay.py 2>&1 | grep -v "amdgpu.ids"
