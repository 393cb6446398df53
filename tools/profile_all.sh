#!/bin/bash
# Everything profiles/<tag>_* is made from, in one gpurun call:
#   gpurun --timeout 1800 -- 'bash tools/profile_all.sh r02'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err
bash $R/tools/profile_bench.sh $TAG > $R/gpurun_out/prof_$TAG.log 2>&1
bash $R/tools/profile_legacy.sh $TAG > $R/gpurun_out/prof_${TAG}_legacy.log 2>&1
bash $R/tools/pmc_legacy.sh > $R/gpurun_out/pmc_legacy.log 2>&1
tail -3 $R/gpurun_out/prof_$TAG.log
