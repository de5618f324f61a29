#!/bin/bash
# kernel times of one L-BFGS tick by launch shape (see tools/lbfgs_tick_probe.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/tick
rocprofv3 --output-format csv --kernel-trace -d $ROOT/gpurun_out/tick -o t -- python $ROOT/tools/lbfgs_tick_probe.py > /dev/null 2>&1
python $ROOT/tools/lbfgs_tick_probe.py --summarize $ROOT/gpurun_out/tick/t_kernel_trace.csv | grep -E "${1:-.}"
rm -rf $ROOT/gpurun_out/tick
