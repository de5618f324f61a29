#!/bin/bash
# k_qp_ipm under rocprofv3 (kernel trace) and its per-section cycle counters:
#   gpurun --timeout 900 -- 'bash tools/prof_qp.sh > gpurun_out/prof_qp.txt 2>&1'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_qp
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_qp -o t -- python $ROOT/tools/time_qp_dev.py > /tmp/prof_qp.log 2>&1
grep "^s " /tmp/prof_qp.log
python3 - <<PY
import csv, glob
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob("/tmp/prof_qp/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "qp_ipm" in row["Kernel_Name"]:
            d[(row["Kernel_Name"][:48], row["Grid_Size_X"], row["VGPR_Count"], row["Accum_VGPR_Count"], row["Scratch_Size"])].append(
                (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
print("kernel | grid | vgpr | agpr | scratch | launches | mean ms | min ms   (rocprofv3 --kernel-trace)")
for k, v in d.items():
    print(" | ".join(k), "| %d | %.2f | %.2f" % (len(v), sum(v) / len(v), min(v)))
PY
echo "--- per-section cycle counters (library rebuilt with -DANET_IPM_PROF in /tmp/anet_ab)"
bash $ROOT/tools/ab_build.sh "-DANET_IPM_PROF" python tools/ipm_prof.py 4,8,16,1 4,8,16,4096 3,5,16,1 2>&1 | grep -v amdgpu
