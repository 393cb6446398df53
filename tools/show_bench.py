#!/usr/bin/env python3
"""One bench line (file with the JSON line of bench.py) as a few readable rows: python3 tools/show_bench.py gpurun_out/x.json ..."""
import json, sys
for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().split("\n")[-1])
    except Exception as e:
        print(path, e)
        continue
    r = d["roofline"]
    print(path, "ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "step_frac", r.get("step_frac"), "xcd", r.get("xcd_runs"),
          "traffic_x", round(r["traffic"] / r["algorithmic_bytes_per_launch"], 4) if r.get("traffic") and r.get("algorithmic_bytes_per_launch") else None)
    for k in ("also_u", "legacy", "config5", "mixed64", "post_stage", "post_stage10", "post_stage14", "rotating_outputs"):
        v = d.get(k)
        if isinstance(v, dict):
            print("   ", k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "ms_per_batch", "frac", "step_frac", "side_parts", "tiles_ms_per_launch", "kernels_ms", "kernels", "traffic")})
    c = d.get("cpu_baseline") or {}
    print("    cpu", c.get("value"), c.get("cores"), c.get("value_1thread"))
    for p in d.get("pool") or []:
        print("    pool", p.get("devices"), {k: (v.get("ms_per_batch") or v.get("frames_per_s")) for k, v in p.items() if isinstance(v, dict)})
    print("    pcie", {k: (v.get("frames_per_s") if isinstance(v, dict) else v) for k, v in d.items() if k.startswith("pcie")})
