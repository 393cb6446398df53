#!/bin/bash
# gpurun -- 'XDEFS="-DA -DB" bash tools/side_xdef.sh': k7_side of the current sources against builds with extra -D flags (one per entry)
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for X in "" ${XDEFS:-}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc $X -o /tmp/libx_$i.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
  i=$((i+1))
done
for rep in 1 2 3; do i=0; for X in "base" ${XDEFS:-}; do
  echo "nat [$X]: $(DIST=nat NS=${NS:-1,240} MCRAW_LIB_PATH=/tmp/libx_$i.so python3 $R/tools/side_scale.py 2>&1 | grep frames | sed 's/k7_tiles.*//' | tr '\n' ' ')"
  echo "u   [$X]: $(DIST=u NS=${NS:-1,240} MCRAW_LIB_PATH=/tmp/libx_$i.so python3 $R/tools/side_scale.py 2>&1 | grep frames | sed 's/k7_tiles.*//' | tr '\n' ' ')"
  i=$((i+1))
done; done
