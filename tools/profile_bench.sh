#!/bin/bash
# Profile the default bench.py workload on the GPU box (run through gpurun):
#   gpurun --timeout 900 -- 'bash tools/profile_bench.sh r01'
# Pass 1: kernel trace + stats.  Passes 2/3: HBM traffic counters, one TCC group per
# pass (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# counters alone with --kernel-trace only.  Raw CSVs land in gpurun_out/prof_<tag>/;
# tools/summarize_profile.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-also > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-also > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d "$OUT/sq" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-also > "$OUT/sq.log" 2>&1
cd "$R" && python3 tools/summarize_profile.py "$TAG" > "$OUT/summary.log" 2>&1
tail -5 "$OUT/summary.log"
