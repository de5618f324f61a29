#!/bin/bash
# Which of the two throughput regimes of the headline kernel does a fresh process land in?  ("Row stride")
# Prints the steady-state % of 8 TB/s for repeated processes under a few runtime settings.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
one() { timeout 120 python $ROOT/tools/time_series.py 576 2>/dev/null | tail -1 | awk '{print $NF}'; }
echo "default:            $(one) $(one) $(one) $(one) $(one) $(one)"
echo "sleep 3 between:    $(one; sleep 3) $(one; sleep 3) $(one; sleep 3) $(one)"
echo "GPU_MAX_HW_QUEUES=1: $(GPU_MAX_HW_QUEUES=1 one) $(GPU_MAX_HW_QUEUES=1 one) $(GPU_MAX_HW_QUEUES=1 one) $(GPU_MAX_HW_QUEUES=1 one)"
echo "HSA_ENABLE_SDMA=0:  $(HSA_ENABLE_SDMA=0 one) $(HSA_ENABLE_SDMA=0 one) $(HSA_ENABLE_SDMA=0 one) $(HSA_ENABLE_SDMA=0 one)"
echo "HSA_XNACK=0:        $(HSA_XNACK=0 one) $(HSA_XNACK=0 one) $(HSA_XNACK=0 one) $(HSA_XNACK=0 one)"
echo "HIP_FORCE_DEV_KERNARG=1: $(HIP_FORCE_DEV_KERNARG=1 one) $(HIP_FORCE_DEV_KERNARG=1 one) $(HIP_FORCE_DEV_KERNARG=1 one) $(HIP_FORCE_DEV_KERNARG=1 one)"
echo "PYTORCH_NO_HIP_MEMORY_CACHING=1: $(PYTORCH_NO_HIP_MEMORY_CACHING=1 one) $(PYTORCH_NO_HIP_MEMORY_CACHING=1 one) $(PYTORCH_NO_HIP_MEMORY_CACHING=1 one) $(PYTORCH_NO_HIP_MEMORY_CACHING=1 one)"
echo "default again:      $(one) $(one) $(one) $(one)"
