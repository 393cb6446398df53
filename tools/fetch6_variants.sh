#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
# gpurun -- 'bash tools/fetch6_variants.sh base cur <name> ...': FETCH_SIZE (x 2: gfx950) of k6_decode per launch for several builds of the library
# (lib/libmcraw_hip_<name>.so; "cur" = the product), 32 x 12 MP legacy frames
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = cur ]; then unset MCRAW_LIB_PATH; else export MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/libmcraw_hip_$v.so; fi
  rm -rf /tmp/t6_$v
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/t6_$v -- python3 $R/tools/prof_workload.py legacy 4 > /tmp/t6_$v.log 2>&1
  python3 - <<PY
import csv, glob
fs = glob.glob("/tmp/t6_$v/**/*_counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "k6_decode" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("$v FETCH MB per launch:", sum(v) / len(v) * 1024 / 1e6 * 2, "launches", len(v))
PY
done
