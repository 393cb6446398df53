#!/bin/bash
# gpurun -- 'bash tools/side_warm2.sh': how far in front of its pieces a speculative part of k7_side starts (candidates of 2 bytes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=${VARS:-"1024 4096"}
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_SPEC_WARM=$V ${XDEF:-} -o /tmp/libw_$V.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
done
for V in $VARS; do
  echo "== spec warm $V"; SPLITS=${SPLITS:-"1;2;4"} SHAPES=${SHAPES:-8K,12MP} MCRAW_LIB_PATH=/tmp/libw_$V.so python3 $R/tools/side_split.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
