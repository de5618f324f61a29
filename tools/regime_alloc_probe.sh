#!/bin/bash
# Fresh processes per allocation variant of the headline solve's output buffer (tools/micro/regime_alloc.hip), rounds
# interleaved so that every variant sees the same box at the same times; every process under its own timeout:
#   gpurun -- 'bash tools/regime_alloc_probe.sh | tee gpurun_out/regime_alloc.txt'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
BIN=$ROOT/tools/micro/regime_alloc
VARIANTS=${VARIANTS:-"0 1 2 3 4 5 6 7 8"}
for round in 1 2 3 4 5 6; do
  for v in $VARIANTS; do
    out=$(timeout 30 $BIN $v 2>&1 | tail -1)
    echo "round $round ${out:-variant $v: no output (timeout or crash)}"
  done
done
