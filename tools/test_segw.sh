#!/bin/bash
# gpurun -- 'bash tools/test_segw.sh': the whole GPU suite against a library whose k7_side follows EVERY side stream with the
# segment walkers (-DMCRAW_FORCE_SEGW): parity, fuzz and negative tests then cover that path on all their inputs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_FORCE_SEGW ${XDEF:-} -o /tmp/libmcraw_segw.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
MCRAW_LIB_PATH=/tmp/libmcraw_segw.so timeout ${TMO:-900} python3 -m pytest $R/tests -m gpu -x -q ${PYTEST_ARGS:-}
