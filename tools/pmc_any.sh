#!/bin/bash
# LDS / issue counters of every anet kernel a command launches (separate --pmc passes, kernel trace only):
#   gpurun --timeout 900 -- 'bash tools/pmc_any.sh python tools/bench_firi.py'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_any
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  (cd $ROOT && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $OUT/p$i -o p -- "$@" > $OUT/p$i.log 2>&1)
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "anet" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"][:56]][r["Counter_Name"]].append(float(r["Counter_Value"]))
order = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = len(d.get("SQ_WAVES", [1]))
    print("==", k, "(%d launches, totals)" % n)
    print("   " + "  ".join("%s %.3g" % (c[3:], sum(d[c])) for c in order if c in d))
PY
find $OUT -name "*.csv" -size +1M -delete
