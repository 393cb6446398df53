#!/bin/bash
# Everything profiles/<tag>_* is made from, in one gpurun call (the program goes directly behind `--`):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r04'
# Per workload of the bench line: rocprofv3 --kernel-trace --stats; for nat, u and legacy also the HBM traffic counters
# (FETCH_SIZE and WRITE_SIZE in separate passes, counters with --kernel-trace only) and one pass of SQ counters.
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in ${WORKLOADS:-nat u legacy mixed64 post12 post10 post14 config5}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/stats" -- python3 "$R/tools/prof_workload.py" $W 20 > "$OUT/$W.stats.log" 2>&1
done
for W in nat u legacy; do
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/$W/fetch" -- python3 "$R/tools/prof_workload.py" $W 4 > "$OUT/$W.fetch.log" 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/$W/write" -- python3 "$R/tools/prof_workload.py" $W 4 > "$OUT/$W.write.log" 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/$W/sq" -- python3 "$R/tools/prof_workload.py" $W 4 > "$OUT/$W.sq.log" 2>&1
done
cd "$R" && python3 tools/summarize_round.py "$TAG" > "$OUT/summary.log" 2>&1
tail -20 "$OUT/summary.log"
