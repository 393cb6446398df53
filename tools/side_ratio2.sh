#!/bin/bash
# gpurun -- 'bash tools/side_ratio2.sh': records per pass below which k7_side's segment walkers take over, on the shapes of tools/side_split.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=${VARS:-"3u 6u 9u"}
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_SEGW_RATIO=$V ${XDEF:-} -o /tmp/libr_$V.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
done
for V in $VARS; do
  echo "== ratio $V"; SPLITS=${SPLITS:-"auto;1"} MCRAW_LIB_PATH=/tmp/libr_$V.so python3 $R/tools/side_split.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
done
