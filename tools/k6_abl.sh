#!/bin/bash
# gpurun -- 'bash tools/k6_abl.sh': what the legacy kernel's stages cost -- timing-only builds (-DK6_ABL=n: wrong pixels) beside the product
R=${GRAFT_REPO_ROOT:-$(pwd)}
S="$(ls $R/motioncam_decoder_amd/csrc/*.hip)"
for a in ${K6_ABLS:-1 7 8 9}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DK6_ABL=$a "$@" -o /tmp/libabl$a.so $S -lpthread || exit 1
done
for i in 1 2; do
  echo "product $(python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-170)"
  for a in ${K6_ABLS:-1 7 8 9}; do
    echo "abl=$a $(MCRAW_NOCHECK=1 MCRAW_LIB_PATH=/tmp/libabl$a.so python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-170)"
  done
done
