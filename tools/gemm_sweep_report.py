"""Compare tools/gemm_sweep.py outputs: per shape the launcher's default against every forced tiling, weighted by launch count."""
import glob
import json
import sys

d = sys.argv[1]
runs = {f.split("sweep_")[-1][:-5]: json.load(open(f)) for f in sorted(glob.glob(f"{d}/sweep_*.json"))}
base = runs["default"]
tot_def = tot_best = 0.0
rows = []
for key, v in base.items():
    best_name, best = "default", v["us"]
    for name, r in runs.items():
        if key in r and r[key]["us"] < best:
            best_name, best = name, r[key]["us"]
    tot_def += v["us"] * v["count"]
    tot_best += best * v["count"]
    rows.append((v["us"] * v["count"], key, v["us"], best_name, best, {n: round(r[key]["us"], 1) for n, r in runs.items() if key in r}))
print(f"GEMM time per 16-frame pass: default {tot_def / 1e3:.2f} ms, best-per-shape {tot_best / 1e3:.2f} ms ({100 * (1 - tot_best / tot_def):.1f} % less)")
for w, key, us, bn, b, allv in sorted(rows, reverse=True)[:45]:
    print(f"{w / 1e3:7.2f} ms  {key:48s} default {us:7.1f} us  best {bn:8s} {b:7.1f}  {allv}")
