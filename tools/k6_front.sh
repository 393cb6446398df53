#!/bin/bash
# gpurun -- 'bash tools/k6_front.sh': what the legacy kernel's front (everything up to the record lists) costs on its own:
# -DK6_ABL=5 (no stream bytes in LDS, waves 0-3 leave behind the maps, no unpack), =6 (the same, the waves stay), the product
R=${GRAFT_REPO_ROOT:-$(pwd)}
S="$R/motioncam_decoder_amd/csrc/mcraw_abi.hip $R/motioncam_decoder_amd/csrc/mcraw_pool.hip $R/motioncam_decoder_amd/csrc/mcraw_type7.hip $R/motioncam_decoder_amd/csrc/mcraw_type6.hip"
for a in 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DK6_ABL=$a -o /tmp/libabl$a.so $S -lpthread || exit 1
done
for i in 1 2; do for v in 0 5 6; do
  if [ $v = 0 ]; then unset MCRAW_LIB_PATH; else export MCRAW_LIB_PATH=/tmp/libabl$v.so; fi
  echo abl$v $(MCRAW_NOCHECK=1 python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-180)
done; done
