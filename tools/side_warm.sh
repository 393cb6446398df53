#!/bin/bash
# gpurun -- 'bash tools/side_warm.sh': warm-up length of k7_side's segment walkers on one box (side builds in /tmp)
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=${VARS:-"3 4 6 8 16"}
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_WARM_SEGS=$V ${XDEF:-} -o /tmp/libw_$V.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
done
for rep in 1 2; do for V in $VARS prev; do
  if [ $V = prev ]; then LIB=$R/motioncam_decoder_amd/lib/libmcraw_hip_prev.so; else LIB=/tmp/libw_$V.so; fi
  echo "uhd12 u   warm $V: $(DIST=u NS=1,240 MCRAW_LIB_PATH=$LIB python3 $R/tools/side_scale.py 2>&1 | grep 'frames' | sed 's/k7_tiles.*//' | tr '\n' ' ')"
  echo "12mp14 u  warm $V: $(DIST=u W=4032 H=3024 NB=14 NMAX=32 NS=1,32 MCRAW_LIB_PATH=$LIB python3 $R/tools/side_scale.py 2>&1 | grep 'frames' | sed 's/k7_tiles.*//' | tr '\n' ' ')"
  echo "uhd10 u   warm $V: $(DIST=u NB=10 NMAX=64 NS=1,64 MCRAW_LIB_PATH=$LIB python3 $R/tools/side_scale.py 2>&1 | grep 'frames' | sed 's/k7_tiles.*//' | tr '\n' ' ')"
done; done
