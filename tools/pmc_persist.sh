#!/bin/bash
# Instruction mix of the one-launch MINCO L-BFGS kernel per evaluation (B = 1 and B = 4096; tools/persist_prof.py):
#   gpurun --timeout 900 -- 'bash tools/pmc_persist.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_persist
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $OUT/p$i -o p -- python $ROOT/tools/persist_prof.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "persistent" not in r["Kernel_Name"]: continue
        acc[(r["Kernel_Name"][:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("==", k)
    for c, v in sorted(d.items()):
        print("   %-26s %16.0f  (mean of %d launches)" % (c, sum(v) / len(v), len(v)))
PY
grep -h "evals of problem" $OUT/p1.log | head -8
find $OUT -name "*.csv" -size +1M -delete
