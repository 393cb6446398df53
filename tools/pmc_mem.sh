#!/bin/bash
# memory-side counters of k7_tiles (diagnostic): gpurun -- 'bash tools/pmc_mem.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu --no-also"
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum -d $R/gpurun_out/pmc_m1 -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum -d $R/gpurun_out/pmc_m2 -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_WRITE_sum TCC_READ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum -d $R/gpurun_out/pmc_m3 -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_m4 -- python3 $R/bench.py $A > /dev/null 2>&1
cd $R; python3 - <<PY
import csv, glob, collections
for d in ("pmc_m1","pmc_m2","pmc_m3","pmc_m4"):
    fs = glob.glob("gpurun_out/%s/**/*_counter_collection.csv" % d, recursive=True)
    if not fs: print(d, "no data"); continue
    agg = collections.defaultdict(float); disp=set()
    for r in csv.DictReader(open(fs[0])):
        if "k7_tiles" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n=max(len(disp),1)
    print(d, {k: round(v/n) for k,v in agg.items()})
PY
