#!/bin/bash
# gpurun -- 'bash tools/k6_lds_steps.sh': where the LDS allocator's steps are -- resident workgroups per CU (from the hardware ids, tools/k6_prof.py)
# and the kernel's time against a few hundred bytes more or less of (unused, dynamic) LDS per workgroup
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
for pad in ${PADS:-0 128 300 600 900 1200 1800 2600 4000 5800}; do echo "== pad $pad: $(MCRAW_K6_LDSPAD=$pad NB=12 DIST=1 MCRAW_LIB_PATH=/tmp/libmcraw_diag.so python3 $R/tools/k6_prof.py 2>&1 | grep -E "ms/launch|distinct" | tr '\n' ' ' | cut -c1-230)"; done
