#!/bin/bash
# A/B of a compile-time switch on the GPU box: builds a second copy of the library with extra flags and runs a command in it.
#   gpurun -- 'bash tools/ab_build.sh "-DANET_PG_SW_MINB=2" python tools/time_cost_grad.py 4,3,8,4096'
FLAGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/anet_ab && cp -r $ROOT /tmp/anet_ab && cd /tmp/anet_ab
ANET_BUILD_FLAGS="$FLAGS" python -m allocnet_amd.build --force > /tmp/anet_ab_build.log 2>&1 || { tail -5 /tmp/anet_ab_build.log; exit 1; }
GRAFT_REPO_ROOT=/tmp/anet_ab "$@"
