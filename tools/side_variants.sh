#!/bin/bash
# gpurun -- 'bash tools/side_variants.sh': k7_side workgroup / piece / unit sizes on one box (side builds in /tmp)
#   VARS="threads:lines per thread:records per unit ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=${VARS:-"512:4:512 512:2:512 1024:2:512"}
for V in $VARS; do
  IFS=: read T LPT LCAP <<< "$V"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_SIDE_T=$T -DMCRAW_SIDE_LPT=$LPT -DMCRAW_SIDE_LCAP=$LCAP -o /tmp/libv_$V.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
done
for rep in 1 2; do for V in $VARS; do
  echo "$V: $(MCRAW_LIB_PATH=/tmp/libv_$V.so python3 $R/tools/side_scale.py 2>&1 | grep 'frames' | sed 's/k7_tiles.*//' | tr '\n' ' ')"
done; done
