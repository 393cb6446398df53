#!/bin/bash
# gpurun -- 'bash tools/side_ratio.sh': records per pass below which k7_side's segment walkers take over (side builds in /tmp)
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=${VARS:-"3u 8u 16u 24u"}
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_SEGW_RATIO=$V ${XDEF:-} -o /tmp/libr_$V.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
done
for rep in 1 2; do for V in $VARS; do
  for D in u nat; do echo "$D ratio $V: $(DIST=$D NS=${NS:-1,240} MCRAW_LIB_PATH=/tmp/libr_$V.so python3 $R/tools/side_scale.py 2>&1 | grep 'frames' | sed 's/k7_tiles.*//' | tr '\n' ' ')"; done
  echo "mixed ratio $V: $(MCRAW_LIB_PATH=/tmp/libr_$V.so python3 $R/tools/bench_mixed.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_batch'], d['kernel_ms_per_launch'])")"
done; done
