// pdh_inst.hip — one translation unit per (PDH_DIM, PDH_NT): instantiates k_diag / k_offdiag for the
// four live-block counts LB of the last fragment and exposes a plain launcher.
#include "pdh_kernels.h"

#ifndef PDH_DIM
#error "compile with -DPDH_DIM=2|3 -DPDH_NT=1..4"
#endif

#define PDH_CAT3(a, b, c) a##b##_##c
#define PDH_NAME(d, n) PDH_CAT3(pdh_launch_, d, n)

// which: 0 = k_diag, 1 = k_offdiag
extern "C" hipError_t PDH_NAME(PDH_DIM, PDH_NT)(int lb, int which, const PdhDev *P, int count, size_t lds,
                                               hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count), block(PDH_WAVE);
#define PDH_CASE(LB)                                                                                   \
  case LB:                                                                                             \
    if (which == 0)                                                                                    \
      hipLaunchKernelGGL((pdh::k_diag<PDH_DIM, PDH_NT, LB>), grid, block, lds, stream, *P, count);      \
    else                                                                                               \
      hipLaunchKernelGGL((pdh::k_offdiag<PDH_DIM, PDH_NT, LB>), grid, block, lds, stream, *P, count);   \
    break;
  switch (lb)
    {
      PDH_CASE(1)
      PDH_CASE(2)
      PDH_CASE(3)
      PDH_CASE(4)
      default:
        return hipErrorInvalidValue;
    }
#undef PDH_CASE
  return hipGetLastError();
}
