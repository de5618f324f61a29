#!/bin/bash
# Phase cycle counters of k_lbfgs_minco_persistent (a second copy of the library built with -DANET_PERSIST_PROF on the
# GPU box, the shipped one is left alone):   gpurun --timeout 900 -- 'bash tools/persist_prof.sh > gpurun_out/persist_prof.txt 2>&1'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cp -r $ROOT /tmp/anet_prof && cd /tmp/anet_prof
ANET_BUILD_FLAGS=-DANET_PERSIST_PROF python -m allocnet_amd.build --force > /dev/null 2>&1
python tools/persist_prof.py 2>&1 | grep -v amdgpu.ids
