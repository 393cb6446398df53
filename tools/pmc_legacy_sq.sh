#!/bin/bash
# issue / wait counters of the legacy kernels: gpurun -- 'bash tools/pmc_legacy_sq.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl_sq
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/pl_sq -- python3 $R/tools/bench_legacy.py > /tmp/pl_sq.log 2>&1
rm -rf /tmp/pl_sq2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d /tmp/pl_sq2 -- python3 $R/tools/bench_legacy.py > /tmp/pl_sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("/tmp/pl_sq", "/tmp/pl_sq2"):
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs: print(d, "no data"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void mcraw::", "").replace("mcraw::", "")
        if not k.startswith("k6"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in agg:
        n = max(len(disp[k]), 1)
        print(k, {c: round(v / n) for c, v in agg[k].items()})
PY
