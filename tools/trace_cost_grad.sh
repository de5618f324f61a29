#!/bin/bash
# Kernel-only durations and start-to-start gaps of the launches of a cost + gradient evaluation from rocprofv3's kernel trace:
#   gpurun -- 'bash tools/trace_cost_grad.sh 4,3,8,4096'     (ANET_RES=1: no sample arithmetic -- the latency floor)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tcg && rocprofv3 --kernel-trace --output-format csv -d /tmp/tcg -o t -- python $ROOT/tools/time_cost_grad.py "$@" > /tmp/tcg.log 2>&1
F=$(find /tmp/tcg -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "anet" in r["Kernel_Name"]]
rows = rows[len(rows) // 2:]                      # the steady state
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = a["Kernel_Name"].split("(")[0][-60:]
    dur[n].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[n].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for n in dur:
    d, g = sorted(dur[n]), sorted(gap[n])
    print(f"{n:62s} calls {len(d):5d}  kernel us med {d[len(d)//2]/1e3:7.2f}  idle until the next kernel starts us med {g[len(g)//2]/1e3:6.2f}")
PY
