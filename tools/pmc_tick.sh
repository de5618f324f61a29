#!/bin/bash
# Issue / stall counters of the four kernels of one L-BFGS evaluation step at the config-4 shape
# (B = 4096, 16-piece jerk): separate --pmc passes, kernel trace only.
#   gpurun --timeout 900 -- 'bash tools/pmc_tick.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TICK_FIRST_ONLY=1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d $OUT/p$i -o p -- python $ROOT/tools/lbfgs_tick_probe.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "anet" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"][:56]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f  (per launch, %d launches)" % (c, sum(v) / len(v), len(v)))
PY
find $OUT -name "*.csv" -size +1M -delete
