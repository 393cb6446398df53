#!/bin/bash
# gpurun -- 'bash tools/traffic6.sh': HBM traffic of k6_decode on the 32 x 12 MP batch (FETCH_SIZE x 2, WRITE_SIZE; KiB)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/t6_$c
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/t6_$c -- python3 $R/tools/prof_workload.py legacy 4 > /tmp/t6_$c.log 2>&1
  python3 - <<PY
import csv, glob
fs = glob.glob("/tmp/t6_$c/**/*_counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "k6_decode" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
print("$c", "per launch KiB", sum(v) / len(v), "launches", len(v), " MB:", sum(v) / len(v) * 1024 / 1e6 * (2 if "$c" == "FETCH_SIZE" else 1))
PY
done
