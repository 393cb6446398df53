#!/bin/bash
# gpurun -- 'bash tools/side_prof.sh': diag builds of the library into /tmp, event timeline of one k7_side workgroup
# (workgroup 1 = refs stream of frame 0, workgroup 0 = its bits stream)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for B in ${BLOCKS:-1 0}; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -DSIDE_PROF_BLOCK=${B}u ${XDEF:-} -o /tmp/libmcraw_diag$B.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
for NN in ${NS:-240 1}; do for D in nat u; do echo "== block $B dist $D frames $NN"; N=$NN DIST=$D W=${W:-3840} H=${H:-2160} NB=${NB:-12} SIGMA=${SIGMA:-12} MCRAW_LIB_PATH=/tmp/libmcraw_diag$B.so python3 $R/tools/side_prof.py 2>&1 | grep -v "amdgpu.ids"; done; done
done
