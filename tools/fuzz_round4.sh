#!/bin/bash
# gpurun -- 'bash tools/fuzz_round4.sh': long mutation fuzz of round 4's kernels (FUZZ_SECONDS each, default 120): the product
# library; every side stream in four parts (MCRAW_SIDE_SPLIT=4,4) on the run speculation and on the segment walkers; the post stage
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_FORCE_SEGW -o /tmp/libmcraw_segw.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
echo "== product"; FUZZ_SEED=41 timeout 600 python3 $R/tools/fuzz_long.py 2>&1 | tail -2
echo "== four parts per side stream"; FUZZ_SEED=42 MCRAW_SIDE_SPLIT=4,4 timeout 600 python3 $R/tools/fuzz_long.py 2>&1 | tail -2
echo "== two + three parts, segment walkers"; FUZZ_SEED=43 MCRAW_SIDE_SPLIT=2,3 MCRAW_LIB_PATH=/tmp/libmcraw_segw.so timeout 600 python3 $R/tools/fuzz_long.py 2>&1 | tail -2
echo "== product, post stage"; FUZZ_SEED=44 FUZZ_POST=1 timeout 600 python3 $R/tools/fuzz_long.py 2>&1 | tail -2
