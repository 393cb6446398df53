#!/bin/bash
# gpurun -- 'bash tools/runs_nt.sh': k7_tiles with plain instead of non-temporal output stores (diag build, MCRAW_ABLATE=4)
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $(ls $R/motioncam_decoder_amd/csrc/*.hip) -lpthread || exit 1
for i in 1 2 3; do for A in 0 4; do for C in 128 0; do
  MCRAW_LIB_PATH=/tmp/libmcraw_diag.so MCRAW_ABLATE=$A MCRAW_XCD_CHUNK=$C python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-also --no-pcie 2>/dev/null | grep "^{" > /tmp/line.json
  python3 - <<PY
import json
d = json.load(open("/tmp/line.json"))
print("ablate $A chunk $C", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["bit_exact"])
PY
done; done; done
