// The FUSE form of the interior-point QP kernel (qp_ipm.h: lone problems, batches of up to 512, problems whose LDS fills a CU) as a
// translation unit of its own, built with -mllvm -amdgpu-sched-strategy=max-ilp (allocnet_amd/build.py): one workgroup per CU,
// nothing to gain from occupancy, every phase a dependent chain.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qp_ipm.h"

namespace anet {

int launch_qp_ipm_fuse(int s, int64_t batch, size_t lds_bytes, hipStream_t st, const IpmArgs &a) {
  hipError_t e;
  if (s == 4) {
    e = hipFuncSetAttribute((const void *)k_qp_ipm<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_qp_ipm<4, 1, true>), dim3((unsigned)batch), dim3(256), lds_bytes, st, a);
  } else {
    e = hipFuncSetAttribute((const void *)k_qp_ipm<3, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_qp_ipm<3, 1, true>), dim3((unsigned)batch), dim3(256), lds_bytes, st, a);
  }
  return (int)hipGetLastError();
}

}  // namespace anet
