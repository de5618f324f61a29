#!/bin/bash
# rocprofv3 evidence for ONE leg of bench.py, taken from bench.py's own leg (same generator, same timing code):
#   gpurun --timeout 1500 -- 'bash tools/profile_leg.sh r06 config3 config5 config4 qp'
# Per leg: (0) the leg's line without a profiler, (1) --kernel-trace --stats, (2) --pmc FETCH_SIZE and (3) --pmc WRITE_SIZE, each in
# its own run (counter passes carry --kernel-trace only).  Raw output under gpurun_out/prof_<tag>_<leg>/; the three files
# tools/summarize_leg.py writes there -- <tag>_<leg>_rocprof_summary.txt, <tag>_<leg>_pmc.json, <tag>_<leg>_line.json -- are what
# goes into profiles/ (bench.py's pmc_leg_traffic reads the pmc.json).
set -u
TAG=${1:-r06}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for LEG in "$@"; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$LEG
  rm -rf $OUT; mkdir -p $OUT
  BENCH="python $ROOT/bench.py --workload $LEG --main-only --no-cpu-baseline"
  timeout 600 $BENCH > $OUT/line.json 2> $OUT/line.log
  timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace_line.json 2> $OUT/trace.log
  timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log
  timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.json 2> $OUT/pmc_write.log
  find $OUT -name "*.db" -delete
  python $ROOT/tools/summarize_leg.py $OUT $TAG $LEG
  find $OUT -type f -size +4M -delete
  cat $OUT/${TAG}_${LEG}_rocprof_summary.txt
done
