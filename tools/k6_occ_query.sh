#!/bin/bash
# gpurun -- 'bash tools/k6_occ_query.sh': workgroups of k6_decode per CU as the runtime computes them, against extra LDS
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
python3 - <<PY
import ctypes
l = ctypes.CDLL("/tmp/libmcraw_diag.so")
for pad in (0, 300, 1000, 5700, 13900, 27600, 54900, 100000):
    print("pad", pad, "workgroups per CU", l.mcraw_diag_k6_occupancy(pad))
PY
/opt/rocm/bin/rocminfo | grep -i -E "lds|wave|compute unit|simd|max waves" | sort | uniq -c | head -20
