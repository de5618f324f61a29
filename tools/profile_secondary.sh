#!/bin/bash
# rocprofv3 kernel stats for the secondary workloads (configs 3/4, QP solve):
#   gpurun --timeout 900 -- 'bash tools/profile_secondary.sh r01'
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/cfg -o cfg -- python $ROOT/tools/bench_configs.py > $OUT/configs.json 2> $OUT/cfg.log
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/qp -o qp -- python $ROOT/tools/bench_qp.py > $OUT/qp.json 2> $OUT/qp.log
python - <<PY > $OUT/summary.txt
import csv
from collections import defaultdict
for name in ("cfg", "qp"):
    d = defaultdict(list)
    for row in csv.DictReader(open("$OUT/%s/%s_kernel_trace.csv" % (name, name))):
        n = row["Kernel_Name"]
        if "anet" not in n:
            continue
        key = (n[:64], row["Grid_Size_X"], row["Grid_Size_Y"], row["VGPR_Count"], row["Accum_VGPR_Count"],
               row["Scratch_Size"], row["LDS_Block_Size"])
        d[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("== %s: kernel, grid x, grid y, vgpr, agpr, scratch, lds: calls, mean us ==" % name)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(" | ".join(k), "| n=%d | mean %.1f us | total %.2f ms" % (len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
PY
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
