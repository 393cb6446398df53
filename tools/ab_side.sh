#!/bin/bash
# gpurun -- 'bash tools/ab_side.sh [names]': k7_side by shape and split, lib/libmcraw_hip_<name>.so ("cur": lib/libmcraw_hip.so), interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${AB_N:-2}); do for v in ${@:-prev cur}; do
  if [ $v = cur ]; then unset MCRAW_LIB_PATH; else export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_$v.so; fi
  echo "== $v"; SPLITS="${SPLITS:-1;2;4;4,1;4,2}" python3 $R/tools/side_split.py 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for line in sys.stdin:
    i = line.find('{')
    try:
        d = json.loads(line[i:])
        print('%-14s' % line[:i].strip(), '  '.join('%s: %6.1f%s' % (k, v['side_us'], '' if v['ok'] else ' BAD') for k, v in d.items()))
    except Exception:
        print(line.rstrip())
"
done; done
