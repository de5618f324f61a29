#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile.sh) into a small text summary for profiles/."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 4:
                print(",".join(x[:80] for x in row))
print()
print("== per-kernel mean duration from the kernel trace ==")
for f in find("trace/**/*kernel_trace.csv"):
    d = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (row["Kernel_Name"][:70], row.get("Grid_Size", "?"), row.get("VGPR_Count", "?"),
                   row.get("Accum_VGPR_Count", "?"), row.get("SGPR_Count", "?"))
            d[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"{k[0]:70s} grid={k[1]:>9s} vgpr={k[2]} agpr={k[3]} sgpr={k[4]} n={len(v):5d} "
              f"mean={sum(v)/len(v)/1e3:10.2f} us  min={min(v)/1e3:10.2f} us")
print()
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    print(f"== counters: {name} ==")
    for f in find(f"{name}/**/*counter_collection.csv"):
        d = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                d[row["Kernel_Name"][:60] + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in d.items():
            if "minco" not in k and "anet" not in k:
                continue
            for cn, v in cs.items():
                print(f"{k[:80]:80s} {cn:22s} n={len(v):4d} mean={sum(v)/len(v):.6g} max={max(v):.6g}")
    print()
