#!/bin/bash
# gpurun -- 'bash tools/k6_prof.sh [extra hipcc flags]': diag build of the library into /tmp, phase stamps of k6_decode
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG "$@" -o /tmp/libmcraw_diag.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
for cfg in ${K6_PROF_CFGS:-"12,1 14,1 14,0"}; do IFS=, read nb dist <<< "$cfg"; echo "== bits $nb dist $dist"; NB=$nb DIST=$dist MCRAW_LIB_PATH=/tmp/libmcraw_diag.so python3 $R/tools/k6_prof.py 2>&1 | grep -v "amdgpu.ids"; done
