#!/bin/bash
# gpurun -- 'bash tools/pmc_post.sh': HBM traffic and memory-side counters of k7_tiles by row format (plain, 12-, 10-, 14-bit strips)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for W in ${WORKLOADS:-nat post12 post10 post14}; do
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum"; do
    D=$R/gpurun_out/pmc_post/$W/$(echo $C | cut -d' ' -f1)
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d $D -- python3 $R/tools/prof_workload.py $W 3 > /dev/null 2>&1
  done
done
cd $R; python3 - <<'PY'
import csv, glob, collections, json
out = {}
for w in ("nat", "post12", "post10", "post14"):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/pmc_post/%s/**/*_counter_collection.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k7_tiles" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    out[w] = {k: round(v / max(len(n[k]), 1)) for k, v in agg.items()}
    print(w, out[w])
json.dump(out, open("gpurun_out/pmc_post/summary.json", "w"), indent=1)
PY
