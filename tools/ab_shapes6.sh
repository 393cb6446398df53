#!/bin/bash
# gpurun -- 'bash tools/ab_shapes6.sh prev cur': legacy content shapes (tools/bench_shapes.py, type 6 only) under several builds of the library
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  if [ $v = cur ]; then unset MCRAW_LIB_PATH; else export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_$v.so; fi
  echo "$v $(SHAPES_TYPES=6 python3 $R/tools/bench_shapes.py 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join('%s=%s%s' % (k.replace('_type6',''), v['gpix_s'], '' if v['bit_exact'] else '(WRONG)') for k,v in d.items() if k.endswith('type6')))")"
done
