#!/bin/bash
# Instruction-mix / stall counters of the interior-point QP kernel (4096 x 8-seg snap, 4096 x 5-seg jerk) (separate --pmc passes, kernel trace only):
#   gpurun --timeout 900 -- 'bash tools/pmc_qp.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_qp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
           "SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d $OUT/p$i -o p -- python $ROOT/tools/time_qp_dev.py 4,8,16,4096 3,5,16,4096 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "anet" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f  (per launch, %d launches)" % (c, sum(v) / len(v), len(v)))
PY
find $OUT -name "*.csv" -size +1M -delete
