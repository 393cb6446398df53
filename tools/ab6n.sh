#!/bin/bash
# A/B/.. of the legacy path between several builds of the library, interleaved on one box:
#   bash tools/ab6n.sh prev w5 cur     (lib/libmcraw_hip_<name>.so; "cur" = lib/libmcraw_hip.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in "$@"; do
  if [ $v = cur ]; then unset MCRAW_LIB_PATH; else export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_$v.so; fi
  echo $v $(python3 $R/tools/bench_legacy.py 2>/dev/null | tail -1 | cut -c1-330)
done; done
