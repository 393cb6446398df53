#!/bin/bash
# latency / level counters of k6_decode: gpurun -- 'bash tools/pmc_legacy_lat.sh [lib-variant]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "${1:-}" ] && [ "$1" != cur ] && export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_$1.so
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/pl_lat$i
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pl_lat$i -- python3 $R/tools/bench_legacy.py > /tmp/pl_lat$i.log 2>&1 || tail -3 /tmp/pl_lat$i.log
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 7):
    d = "/tmp/pl_lat%d" % i
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs: print(d, "no data"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void mcraw::", "").replace("mcraw::", "")
        if not k.startswith("k6"): continue
        g = r.get("Grid_Size", "")
        agg[(k, g)][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, g)].add(r["Dispatch_Id"])
    for k in agg:
        n = max(len(disp[k]), 1)
        print(k, n, {c: round(v / n) for c, v in agg[k].items()})
PY
