#!/bin/bash
# gpurun -- 'bash tools/fuzz_segw.sh': long mutation fuzz against the product library and against a build whose k7_side puts
# every side stream on the segment walkers (FUZZ_SECONDS each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_FORCE_SEGW -o /tmp/libmcraw_segw.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
echo "== product"; FUZZ_SEED=31 timeout 600 python3 $R/tools/fuzz_long.py 2>&1 | tail -3
echo "== forced segment walkers"; FUZZ_SEED=32 MCRAW_LIB_PATH=/tmp/libmcraw_segw.so timeout 600 python3 $R/tools/fuzz_long.py 2>&1 | tail -3
echo "== product, post stage"; FUZZ_SEED=33 FUZZ_POST=1 timeout 600 python3 $R/tools/fuzz_long.py 2>&1 | tail -3
