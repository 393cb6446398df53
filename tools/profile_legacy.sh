#!/bin/bash
# rocprofv3 kernel statistics of the legacy (type 6) kernels on tools/bench_legacy.py:
#   gpurun --timeout 600 -- 'bash tools/profile_legacy.sh r01'  ->  profiles/<tag>_legacy_kernel_stats.csv
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_legacy
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/bench_legacy.py" > "$OUT/stats.log" 2>&1
tail -1 "$OUT/stats.log" | cut -c1-300
