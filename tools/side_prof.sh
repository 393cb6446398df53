#!/bin/bash
# gpurun -- 'bash tools/side_prof.sh': diag build of the library into /tmp, phase stamps of k7_side
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $R/motioncam_decoder_amd/csrc/mcraw_abi.hip $R/motioncam_decoder_amd/csrc/mcraw_type7.hip $R/motioncam_decoder_amd/csrc/mcraw_type6.hip || exit 1
MCRAW_LIB_PATH=/tmp/libmcraw_diag.so python3 $R/tools/side_prof.py
DIST=u MCRAW_LIB_PATH=/tmp/libmcraw_diag.so python3 $R/tools/side_prof.py
