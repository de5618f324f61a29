#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun):
#   gpurun --timeout 900 -- 'bash tools/profile.sh r01'
# Writes raw output under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --main-only"
# 1. kernel trace + stats
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
# 2. PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -type f -size +8M -delete; find $OUT -type f | head -50
python $ROOT/tools/summarize_prof.py $OUT $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
