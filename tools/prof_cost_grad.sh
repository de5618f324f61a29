#!/bin/bash
# Per-kernel times of three cost + gradient evaluations at B = 131072 (tools/run_cost_grad.py) from the kernel trace:
#   gpurun --timeout 600 -- 'bash tools/prof_cost_grad.sh [tag]'     (env such as ANET_PIECE_LIST_MIN_BATCH is passed on)
TAG=${1:-a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_cg_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/t -o t -- python $ROOT/tools/run_cost_grad.py > $OUT/run.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "anet" not in n: continue
        key = (n[:60], row["Grid_Size_X"], row["Grid_Size_Y"], row["VGPR_Count"], row["Accum_VGPR_Count"], row["Scratch_Size"])
        d[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
print("kernel | grid x | grid y | vgpr | agpr | scratch | calls | mean us | min us")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(" | ".join(k), "| %d | %.1f | %.1f" % (len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
find $OUT -name "*.csv" -size +2M -delete
