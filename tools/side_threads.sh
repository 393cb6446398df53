#!/bin/bash
# gpurun -- 'bash tools/side_threads.sh': k7_side with 512 and 1024 threads per workgroup over batch and frame sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}
for T in 512 1024; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_SIDE_T=$T -DMCRAW_SIDE_LPT=$((2048/T)) -o /tmp/libt_$T.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
done
for rep in 1 2; do for T in 512 1024; do
  echo "uhd nat T=$T: $(DIST=nat NS=1,16,64,120,128 MCRAW_LIB_PATH=/tmp/libt_$T.so python3 $R/tools/side_scale.py 2>&1 | grep frames | sed 's/k7_tiles.*//' | tr '\n' ' ')"
  echo "uhd u   T=$T: $(DIST=u NS=1,64,120 MCRAW_LIB_PATH=/tmp/libt_$T.so python3 $R/tools/side_scale.py 2>&1 | grep frames | sed 's/k7_tiles.*//' | tr '\n' ' ')"
  echo "8k nat  T=$T: $(DIST=nat W=7680 H=4320 NMAX=120 NS=1,30,120 MCRAW_LIB_PATH=/tmp/libt_$T.so python3 $R/tools/side_scale.py 2>&1 | grep frames | sed 's/k7_tiles.*//' | tr '\n' ' ')"
done; done
