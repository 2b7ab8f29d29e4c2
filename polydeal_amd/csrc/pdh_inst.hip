// pdh_inst.hip — one translation unit per PDH_GROUP: instantiates k_diag (with and without reaction
// term) and k_offdiag for the combos of that group (pdh_combos.h) and exposes a plain launcher.
#include "pdh_combos.h"
#include "pdh_kernels.h"

#ifndef PDH_GROUP
#error "compile with -DPDH_GROUP=0..7"
#endif

#define PDH_CAT_(a, b) a##b
#define PDH_CAT(a, b) PDH_CAT_(a, b)

// which: 0 = k_diag, 2 = k_diag with reaction term, 1 = k_offdiag
#define PDH_LAUNCH_CASE(D, N, T, L)                                                                        \
  if (dim == D && n1d == N && nt == T && lb == L)                                                          \
    {                                                                                                      \
      if (which == 0)                                                                                      \
        hipLaunchKernelGGL((pdh::k_diag<D, N, T, L, false>), grid, block, lds, stream, *P, count);          \
      else if (which == 2)                                                                                 \
        hipLaunchKernelGGL((pdh::k_diag<D, N, T, L, true>), grid, block, lds, stream, *P, count);           \
      else                                                                                                 \
        hipLaunchKernelGGL((pdh::k_offdiag<D, N, T, L>), grid, block, lds, stream, *P, count);              \
      return hipGetLastError();                                                                            \
    }
#define PDH_SKIP(D, N, T, L)
#define PDH_SEL_0 PDH_SKIP
#define PDH_SEL_1 PDH_SKIP
#define PDH_SEL_2 PDH_SKIP
#define PDH_SEL_3 PDH_SKIP
#define PDH_SEL_4 PDH_SKIP
#define PDH_SEL_5 PDH_SKIP
#define PDH_SEL_6 PDH_SKIP
#define PDH_SEL_7 PDH_SKIP
#if PDH_GROUP == 0
#undef PDH_SEL_0
#define PDH_SEL_0 PDH_LAUNCH_CASE
#elif PDH_GROUP == 1
#undef PDH_SEL_1
#define PDH_SEL_1 PDH_LAUNCH_CASE
#elif PDH_GROUP == 2
#undef PDH_SEL_2
#define PDH_SEL_2 PDH_LAUNCH_CASE
#elif PDH_GROUP == 3
#undef PDH_SEL_3
#define PDH_SEL_3 PDH_LAUNCH_CASE
#elif PDH_GROUP == 4
#undef PDH_SEL_4
#define PDH_SEL_4 PDH_LAUNCH_CASE
#elif PDH_GROUP == 5
#undef PDH_SEL_5
#define PDH_SEL_5 PDH_LAUNCH_CASE
#elif PDH_GROUP == 6
#undef PDH_SEL_6
#define PDH_SEL_6 PDH_LAUNCH_CASE
#elif PDH_GROUP == 7
#undef PDH_SEL_7
#define PDH_SEL_7 PDH_LAUNCH_CASE
#endif
#define PDH_X(G, D, N, T, L) PDH_CAT(PDH_SEL_, G)(D, N, T, L)

extern "C" hipError_t PDH_CAT(pdh_launch_g, PDH_GROUP)(int dim, int n1d, int nt, int lb, int which, const PdhDev *P,
                                                        int count, size_t lds, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count), block(PDH_WAVE);
  PDH_COMBOS(PDH_X)
  return hipErrorInvalidValue;
}
