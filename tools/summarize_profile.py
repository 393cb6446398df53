#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 CSVs from tools/profile_bench.sh) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_counters.json and profiles/traffic.json
(the HBM bytes per k7_tiles launch that bench.py reports as roofline.traffic).

gfx950 corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming
read, so the read side is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    files = sorted(glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: {c: v / max(len(disp[k]), 1) for c, v in cs.items()} for k, cs in agg.items()}, \
        {k: len(v) for k, v in disp.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:  # the newest run (gpurun merges every run's files into the same directory)
        shutil.copy(stats[-1], os.path.join(dst, tag + "_kernel_stats.csv"))
    out = {"tag": tag, "command": "rocprofv3 ... -- python3 bench.py --steps N --warmup W --no-cpu --no-also (tools/profile_bench.sh)"}
    per = {}
    for name in ("fetch", "write", "sq"):
        c, d = counters(os.path.join(src, name))
        for k, v in c.items():
            if "mcraw" in k:
                per.setdefault(k, {}).update(v)
                per[k]["dispatches_" + name] = d[k]
    out["per_dispatch"] = per
    tiles = next((v for k, v in per.items() if "k7_tiles" in k), None)
    if tiles and "FETCH_SIZE" in tiles and "WRITE_SIZE" in tiles:
        rd = 2.0 * tiles["FETCH_SIZE"] * 1024.0   # gfx950: FETCH_SIZE counts 64 B per 128-B request
        wr = tiles["WRITE_SIZE"] * 1024.0
        out["k7_tiles_hbm"] = {"read_bytes_corrected": rd, "write_bytes": wr, "total": rd + wr,
                               "fetch_size_kib_raw": tiles["FETCH_SIZE"], "write_size_kib_raw": tiles["WRITE_SIZE"]}
        tpath = os.path.join(dst, "traffic.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        # bench.py's default workload key; dispatches may be split into sub-batches: sum per step
        sub = 1
        try:
            for line in open(os.path.join(src, "stats.log")):
                if line.startswith("{"):
                    j = json.loads(line)
                    sub = max(1, round(j["roofline"].get("launches_per_step", 1)))
        except Exception:
            pass
        t["3840x2160_12bit_240_nat"] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                                        "launches_per_step": sub, "profile": tag,
                                        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950)"}
        json.dump(t, open(tpath, "w"), indent=1)
    json.dump(out, open(os.path.join(dst, tag + "_counters.json"), "w"), indent=1)
    print(json.dumps(out.get("k7_tiles_hbm", {})))
    for line in open(os.path.join(dst, tag + "_kernel_stats.csv")):
        print(line.strip()[:160])


if __name__ == "__main__":
    main()
