"""Condense rocprofv3 --kernel-trace --stats output (kernel_stats.csv) into the text table kept under profiles/."""
import csv
import glob
import sys

root, cmd = sys.argv[1], sys.argv[2]
f = glob.glob(f"{root}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(cmd + "   (MI355X; one HIP stream so that a kernel's duration is its own)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
    t = float(r["TotalDurationNs"])
    print(f"{t / 1e6:9.2f} ms {100 * t / tot:6.2f}% calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:150]}")
print(f"total GPU ms {tot / 1e6:.1f}")
