#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/<workload>/ (rocprofv3 CSVs of tools/profile_round.sh) -> profiles/<tag>_<workload>_kernel_stats.csv
(the mcraw kernels' rows, with the run's JSON line -- box yardstick included -- as a comment header), profiles/<tag>_counters.json
and the HBM traffic of the dominant kernels (profiles/traffic.json for k7_tiles, profiles/<tag>_legacy_traffic.json).

gfx950 corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 64 B per 128-B request of a
wide coalesced read, so the read side is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    try:
        for line in reversed(open(path).read().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


def counters(path):
    files = sorted(glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "mcraw" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: dict({c: v / max(len(disp[k]), 1) for c, v in cs.items()}, dispatches=len(disp[k])) for k, cs in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    summary = {"tag": tag, "made_by": "tools/profile_round.sh (rocprofv3 ... -- python3 tools/prof_workload.py <workload> <launches>)", "workloads": {}}
    for w in ("nat", "u", "legacy", "mixed64", "post12", "post10", "post14", "config5"):
        info = last_json(os.path.join(src, w + ".stats.log"))
        stats = sorted(glob.glob(os.path.join(src, w, "stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        entry = {"run": info}
        if stats:
            rows = list(csv.reader(open(stats[-1])))
            keep = [rows[0]] + [r for r in rows[1:] if r and "mcraw" in r[0]]
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, w)), "w", newline="") as f:
                f.write("# rocprofv3 --kernel-trace --stats -- python3 tools/prof_workload.py %s 20 (the mcraw kernels' rows)\n" % w)
                f.write("# run: %s\n" % json.dumps(info))
                csv.writer(f, quoting=csv.QUOTE_ALL).writerows(keep)
            entry["kernels"] = {r[0].split("(")[0].replace("void ", ""): {"calls": int(r[1]), "avg_ns": float(r[3]), "min_ns": float(r[5]), "max_ns": float(r[6])}
                                for r in keep[1:]}
        per = {}
        for name in ("fetch", "write", "sq"):
            for k, v in counters(os.path.join(src, w, name)).items():
                per.setdefault(k, {}).update({(c if c != "dispatches" else "dispatches_" + name): x for c, x in v.items()})
        if per:
            for k, v in per.items():
                if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    v["hbm_read_bytes_corrected"] = 2.0 * v["FETCH_SIZE"] * 1024.0
                    v["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024.0
                    v["hbm_bytes_per_launch"] = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
            entry["counters_per_dispatch"] = per
        summary["workloads"][w] = entry
        print(w, json.dumps({k: v for k, v in entry.items() if k != "counters_per_dispatch"})[:600])
    json.dump(summary, open(os.path.join(dst, tag + "_counters.json"), "w"), indent=1)
    # traffic of the dominant kernels, as bench.py reports it
    tpath = os.path.join(dst, "traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for w, key in (("nat", "3840x2160_12bit_240_nat"), ("u", "3840x2160_12bit_240_u")):
        per = summary["workloads"].get(w, {}).get("counters_per_dispatch", {})
        tiles = next((v for k, v in per.items() if "k7_tiles" in k and "hbm_bytes_per_launch" in v), None)
        if tiles:
            t[key] = {"hbm_bytes_per_launch": tiles["hbm_bytes_per_launch"], "read_bytes": tiles["hbm_read_bytes_corrected"],
                      "write_bytes": tiles["hbm_write_bytes"], "launches_per_step": 1, "profile": tag,
                      "algorithmic_bytes": (summary["workloads"][w].get("run") or {}).get("algorithmic_bytes_per_batch"),
                      "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950)"}
    json.dump(t, open(tpath, "w"), indent=1)
    per = summary["workloads"].get("legacy", {}).get("counters_per_dispatch", {})
    k6 = next((v for k, v in per.items() if "k6_decode" in k and "hbm_bytes_per_launch" in v), None)
    if k6:
        alg = (summary["workloads"]["legacy"].get("run") or {}).get("algorithmic_bytes_per_batch")
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/prof_workload.py legacy: 32 x 4000x3000 12-bit Nat "
                           "type-6 frames, per launch of k6_decode; FETCH_SIZE x2 (gfx950 wide reads), KiB -> bytes",
                   "kernels": {"k6_decode": {"read_bytes": k6["hbm_read_bytes_corrected"], "write_bytes": k6["hbm_write_bytes"]}},
                   "total_bytes_per_batch": k6["hbm_bytes_per_launch"], "algorithmic_bytes_per_batch": alg,
                   "ratio": round(k6["hbm_bytes_per_launch"] / alg, 4) if alg else None},
                  open(os.path.join(dst, tag + "_legacy_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
