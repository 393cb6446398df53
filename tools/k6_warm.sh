#!/bin/bash
# How far in front of its quarter chunk a speculative walker of k6_decode starts (WARM6, csrc/mcraw_type6.hip).  Build the variants
# HERE (no GPU needed), then A/B them on one box:
#   bash tools/k6_warm.sh build 384 640 768 1024 && gpurun -- 'bash tools/k6_warm.sh run 384 640 768 1024'
R=${GRAFT_REPO_ROOT:-$(pwd)}
mode=$1; shift
if [ "$mode" = build ]; then
  for w in "$@"; do python3 -m motioncam_decoder_amd.build variant $R/motioncam_decoder_amd/lib/libmcraw_hip_w$w.so -DMCRAW_WARM6=$w || exit 1; done
else
  names=(cur); for w in "$@"; do names+=(w$w); done
  bash $R/tools/ab6n.sh "${names[@]}" | cut -c1-150 | sort
fi
