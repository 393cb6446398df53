#!/bin/bash
# gpurun -- 'bash tools/k6_occ.sh': the legacy kernel at 7 .. 2 workgroups per CU (unused dynamic LDS: 22 144 + pad bytes per workgroup, LDS is
# handed out in units of 1 280 bytes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
S="$(ls $R/motioncam_decoder_amd/csrc/*.hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG "$@" -o /tmp/libpadenv.so $S -lpthread || exit 1
for i in 1 2; do for a in ${K6_PADS:-0 4096 9216 17408 30720 58368}; do
  echo "pad=$a wg/cu=$((163840 / ((22144 + a + 1279) / 1280 * 1280))) $(MCRAW_K6_LDSPAD=$a MCRAW_LIB_PATH=/tmp/libpadenv.so python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-170)"
done; done
