#!/bin/bash
# gpurun -- 'bash tools/store_sc.sh': k7_tiles' output stores with other cache policies (-DMCRAW_STORE_POLICY: "sc1 nt" = the
# product, "nt", "sc1", "sc0 sc1", "sc0 sc1 nt"), the headline leg in fresh processes, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
S="$(ls $R/motioncam_decoder_amd/csrc/*.hip)"
POL=("sc1 nt" "nt" "sc1" "sc0 sc1" "sc0 sc1 nt")
for k in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc "-DMCRAW_STORE_POLICY=\"${POL[$k]}\"" -o /tmp/libsc$k.so $S -lpthread || exit 1
done
for i in $(seq 1 ${N:-4}); do for k in 0 1 2 3 4; do
  MCRAW_LIB_PATH=/tmp/libsc$k.so python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-also --no-pcie 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${POL[$k]}:', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['bit_exact'])"
done; done
