#!/bin/bash
# A/B of the fused post stage (12-bit strips + black levels) on one box: lib/libmcraw_hip_prev.so vs lib/libmcraw_hip.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in prev cur; do
  if [ $v = prev ]; then export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_prev.so; else unset MCRAW_LIB_PATH; fi
  python3 $R/bench.py --steps 10 --warmup 2 --no-also --no-pcie --cpu-seconds 0.2 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['post_stage']; print('$v', 'plain', d['ms_per_step'], 'post step', p['ms_per_step'], 'tiles', p['tiles_ms_per_launch'], p['bit_exact'])"
done; done
