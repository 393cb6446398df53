#!/bin/bash
# gpurun -- 'bash tools/side_variants.sh': k7_side piece / unit sizes on one box (product kernels, side builds in /tmp)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for V in "4 256" "4 512" "4 1024" "2 256" "2 512" "2 1024"; do
  set -- $V
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_SIDE_LPT=$1 -DMCRAW_SIDE_LCAP=$2 -o /tmp/libv_$1_$2.so $R/motioncam_decoder_amd/csrc/mcraw_abi.hip $R/motioncam_decoder_amd/csrc/mcraw_type7.hip $R/motioncam_decoder_amd/csrc/mcraw_type6.hip || exit 1
done
for rep in 1 2; do for V in "4 256" "4 512" "4 1024" "2 256" "2 512" "2 1024"; do
  set -- $V
  echo "LPT $1 LCAP $2: $(MCRAW_LIB_PATH=/tmp/libv_$1_$2.so python3 $R/tools/side_scale.py 2>&1 | grep 'frames    1 \|frames  240' | tr '\n' ' ')"
done; done
