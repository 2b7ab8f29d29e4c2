// pdh_tiled.hip — instantiations and launcher of the tiled kernels (pdh_tiled.h): 3-D, N1D = degree + 1 = 5 .. 8.
#include "pdh_tiled.h"

template <int N1D>
static hipError_t launch_n1d(int which, const PdhDev *P, int count, int ntile, hipStream_t stream)
{
  const size_t lds = which == 1 ? pdht2::lds_bytes_toffdiag(3, N1D) : pdht2::lds_bytes_tdiag(3, N1D);
  const dim3 block(PDH_WAVE);
  if (count <= 0)
    return hipSuccess;
  if (which == 1)
    {
      const long long blocks = (long long)ntile * ntile * count;
      if (blocks > 0x7fffffffLL)
        return hipErrorInvalidValue;
      hipLaunchKernelGGL((pdht2::k_toffdiag<3, N1D>), dim3((unsigned)blocks), block, lds, stream, *P, count, ntile);
      return hipGetLastError();
    }
  // own blocks: the symmetric tiles ti == tj, then the pairs ti < tj
  const long long bd = (long long)ntile * count, bo = (long long)ntile * (ntile - 1) / 2 * count;
  if (bd > 0x7fffffffLL || bo > 0x7fffffffLL)
    return hipErrorInvalidValue;
  if (which == 0)
    {
      hipLaunchKernelGGL((pdht2::k_tdiag<3, N1D, false, true>), dim3((unsigned)bd), block, lds, stream, *P, count, ntile);
      if (bo > 0)
        hipLaunchKernelGGL((pdht2::k_tdiag<3, N1D, false, false>), dim3((unsigned)bo), block, lds, stream, *P, count, ntile);
    }
  else
    {
      hipLaunchKernelGGL((pdht2::k_tdiag<3, N1D, true, true>), dim3((unsigned)bd), block, lds, stream, *P, count, ntile);
      if (bo > 0)
        hipLaunchKernelGGL((pdht2::k_tdiag<3, N1D, true, false>), dim3((unsigned)bo), block, lds, stream, *P, count, ntile);
    }
  return hipGetLastError();
}

// which: 0 = own blocks, 2 = own blocks with reaction term, 1 = coupling blocks; count = owned polytopes / face items
extern "C" hipError_t pdh_launch_tiled(int dim, int n1d, int which, const PdhDev *P, int count, hipStream_t stream)
{
  if (dim != 3)
    return hipErrorInvalidValue;
  const int ntile = (P->n + 63) / 64;
  switch (n1d)
    {
    case 5:
      return launch_n1d<5>(which, P, count, ntile, stream);
    case 6:
      return launch_n1d<6>(which, P, count, ntile, stream);
    case 7:
      return launch_n1d<7>(which, P, count, ntile, stream);
    case 8:
      return launch_n1d<8>(which, P, count, ntile, stream);
    default:
      return hipErrorInvalidValue;
    }
}

extern "C" int pdh_tiled_has_kind(int dim, int n1d, int n) { return dim == 3 && n1d >= 5 && n1d <= 8 && n > 64; }
