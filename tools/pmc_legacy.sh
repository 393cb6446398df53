#!/bin/bash
# HBM traffic counters of the legacy kernels (FETCH_SIZE / WRITE_SIZE in separate passes, kernel trace only):
#   gpurun --timeout 600 -- 'bash tools/pmc_legacy.sh'   ->  gpurun_out/pmc_legacy/{fetch,write}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_legacy
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -- python3 "$R/tools/bench_legacy.py" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -- python3 "$R/tools/bench_legacy.py" > "$OUT/write.log" 2>&1
echo done
