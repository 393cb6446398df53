#!/bin/bash
# gpurun -- 'bash tools/k6_replay_pmc.sh': counters of k6_decode with its maps walked (mode 0) and read back (mode 2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DMCRAW_DIAG -o /tmp/libmcraw_diag.so $R/motioncam_decoder_amd/csrc/mcraw_abi.hip $R/motioncam_decoder_amd/csrc/mcraw_pool.hip $R/motioncam_decoder_amd/csrc/mcraw_type7.hip $R/motioncam_decoder_amd/csrc/mcraw_type6.hip -lpthread || exit 1
export MCRAW_LIB_PATH=/tmp/libmcraw_diag.so
cd /tmp && export TMPDIR=/tmp
for m in 0 2; do
  export K6_MODE=$m
  rm -rf /tmp/rp_$m
  rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY -d /tmp/rp_$m -- python3 $R/tools/k6_replay_one.py > /tmp/rp_$m.log 2>&1
  tail -2 /tmp/rp_$m.log
  python3 - <<PY
import csv, glob, collections
d = "/tmp/rp_$m"
fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        if "k6_decode" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "k6_decode" not in r["Kernel_Name"]: continue
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(agg, key=int)[2:]   # (the first two dispatches: the decode that recorded, and a warm-up)
for i in ids[-4:]:
    c = agg[i]; ns = dur.get(i, 0)
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / ns if ns else 0
    print("mode $m dispatch", i, "ns", ns, {k: round(v) for k, v in c.items()}, "eff clock GHz %.3f" % clk)
PY
done
