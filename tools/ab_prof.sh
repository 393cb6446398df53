#!/bin/bash
# rocprofv3 kernel durations of two builds on one box: lib/libmcraw_hip_prev.so vs lib/libmcraw_hip.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in prev cur prev cur; do
  if [ $v = prev ]; then export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_prev.so; else unset MCRAW_LIB_PATH; fi
  rm -rf /tmp/abp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-also > /tmp/abp_$v.log 2>&1
  f=$(find /tmp/abp_$v -name "*_kernel_stats.csv" | head -1)
  echo "== $v: $(grep '^{' /tmp/abp_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
  grep "mcraw\|copyBuffer" $f | cut -d, -f1-4 | sed 's/(mcraw::Work7[^"]*//'
done
