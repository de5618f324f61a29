#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_leg.sh for one bench.py leg into the three files that go into profiles/:
<tag>_<leg>_rocprof_summary.txt (per (kernel, grid): launches, mean / min / max duration from the kernel trace, registers, scratch,
LDS; the leg's own line without and under the profiler), <tag>_<leg>_pmc.json (per (kernel, grid): mean FETCH_SIZE / WRITE_SIZE in
KiB per launch -- what bench.py's pmc_leg_traffic reads) and <tag>_<leg>_line.json (the leg's line without a profiler)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag, leg = sys.argv[1], sys.argv[2], sys.argv[3]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def last_json_line(path):
    try:
        for ln in reversed(open(path).read().strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
    except Exception:
        pass
    return None


lines = []
lines.append(f"== bench.py --workload {leg} --main-only --no-cpu-baseline: rocprofv3 --kernel-trace --stats ==")
trace = defaultdict(list)
meta = {}
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            n = row["Kernel_Name"]
            if "anet" not in n:
                continue
            gx, gy, gz = (int(row.get(k, 1) or 1) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            key = (n, gx * gy * gz)
            trace[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[key] = (row.get("VGPR_Count", "?"), row.get("Accum_VGPR_Count", "?"), row.get("SGPR_Count", "?"),
                         row.get("Scratch_Size", "?"), row.get("LDS_Block_Size", "?"), f"{gx}x{gy}x{gz}")
lines.append("kernel | grid (work-items) | vgpr agpr sgpr scratch lds | launches | mean us | min us | max us | total ms")
for k, v in sorted(trace.items(), key=lambda kv: -sum(kv[1])):
    m = meta[k]
    lines.append(f"{k[0][:96]} | {k[1]} ({m[5]}) | {m[0]} {m[1]} {m[2]} {m[3]} {m[4]} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | "
                 f"{min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {sum(v) / 1e6:.3f}")
lines.append("")
acc = defaultdict(dict)
for name, cn in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    lines.append(f"== counters: --pmc {cn} (own run, --kernel-trace only) ==")
    tmp = defaultdict(list)
    for f in find(f"{name}/**/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "anet" in row["Kernel_Name"] and row["Counter_Name"] == cn:
                    tmp[(row["Kernel_Name"], int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    for k, v in sorted(tmp.items(), key=lambda kv: -sum(kv[1])):
        acc[k][cn] = (sum(v) / len(v), len(v), max(v), min(v))
        lines.append(f"{k[0][:96]} | grid {k[1]} | n={len(v)} | mean {sum(v) / len(v):.6g} KiB | min {min(v):.6g} | max {max(v):.6g}")
    lines.append("")
kernels = []
for k, v in sorted(acc.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        kernels.append({"name": k[0], "grid": k[1], "n": v["FETCH_SIZE"][1], "fetch_kib": v["FETCH_SIZE"][0], "write_kib": v["WRITE_SIZE"][0],
                        "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024.0})
clean = last_json_line(os.path.join(out, "line.json"))
under = last_json_line(os.path.join(out, "trace_line.json"))
lines.append("== the leg's line without a profiler / under --kernel-trace (ms per step) ==")


def steps_of(d):
    if not d:
        return None
    r = {"ms_per_step": d.get("ms_per_step")}
    for key in ("config3", "config4", "config5", "qp_solve"):
        if key in d:
            o = d[key]
            for sub in ("b4096", "saturating", "snap8", "jerk5"):
                if sub in o:
                    r[sub] = o[sub].get("stream_ms_per_step", o[sub].get("ms_per_batch"))
            for f in ("stream_seconds", "kernel_ms"):
                if f in o:
                    r[f] = o[f]
    return r


lines.append("without: " + json.dumps(steps_of(clean)))
lines.append("under:   " + json.dumps(steps_of(under)))
with open(os.path.join(out, f"{tag}_{leg}_rocprof_summary.txt"), "w") as fh:
    fh.write("\n".join(lines) + "\n")
with open(os.path.join(out, f"{tag}_{leg}_pmc.json"), "w") as fh:
    json.dump({"leg": leg, "command": f"bench.py --workload {leg} --main-only --no-cpu-baseline",
               "note": "mean per launch; KiB as reported by rocprofv3; hbm_bytes_per_launch = (WRITE_SIZE + 2 x FETCH_SIZE) x 1024 "
                       "(gfx950 read correction, MI355X_MICROARCH.md)", "kernels": kernels}, fh, indent=1)
if clean:
    with open(os.path.join(out, f"{tag}_{leg}_line.json"), "w") as fh:
        fh.write(json.dumps(clean, separators=(",", ":")) + "\n")
