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
            key = (row["Kernel_Name"][:70], row.get("Grid_Size_X", row.get("Grid_Size", "?")), row.get("VGPR_Count", "?"),
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

# machine-readable PMC summary for bench.py's roofline.traffic
if len(sys.argv) > 2:
    import json, re
    tag = sys.argv[2]
    acc = defaultdict(dict)
    for name, cn in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for f in find(f"{name}/**/*counter_collection.csv"):
            tmp = defaultdict(list)
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    m = re.match(r"void anet::k_minco_solve<(\d+), (\d+)", row["Kernel_Name"])
                    if m and row["Counter_Name"] == cn:
                        tmp[(int(m.group(1)), int(m.group(2)), int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
            for k, v in tmp.items():
                acc[k][cn] = sum(v) / len(v)
    entries = [{"order": k[0], "pieces": k[1], "grid": k[2], "fetch_kib": v.get("FETCH_SIZE"),
                "write_kib": v.get("WRITE_SIZE")} for k, v in sorted(acc.items())
               if "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    with open(os.path.join(out, f"{tag}_pmc.json"), "w") as fh:
        json.dump({"k_minco_solve": entries, "note": "mean per launch; KiB as reported by rocprofv3"}, fh, indent=1)
