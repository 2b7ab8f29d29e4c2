// pdh_moment.hip — instantiations and launcher of the moment-form kernels (pdh_moment.h): 3-D, degree <= 3.
#include <cstdlib>
#include "pdh_moment.h"
#include "pdh_rows.h"

// which: 0 = diagonal blocks (count = owned polytopes), 1 = coupling blocks (count = interior-face items)
extern "C" hipError_t pdh_launch_moment(int n1d, int which, const PdhDev *P, const double *mtab, int count, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count), block(PDH_WAVE);
  if (n1d == 4 && P->n == 64) // FE_DGQ(3): contraction stages 2 and 3 on the MFMA
    {
      if (which == 0)
        hipLaunchKernelGGL((pdhm::k_mdiag<4, true>), grid, block, pdhm::lds_doubles_diag<4>() * sizeof(double), stream, *P, mtab, count);
      else
        hipLaunchKernelGGL((pdhm::k_moffdiag<4, true>), grid, block, pdhm::lds_doubles_offdiag<4>() * sizeof(double), stream, *P,
                           mtab, count);
      return hipGetLastError();
    }
#define PDH_MOM_CASE(N)                                                                                              \
  if (n1d == N)                                                                                                      \
    {                                                                                                                \
      if (which == 0)                                                                                                \
        hipLaunchKernelGGL((pdhm::k_mdiag<N, false>), grid, block, pdhm::lds_doubles_diag<N>() * sizeof(double), stream, *P, \
                           mtab, count);                                                                             \
      else                                                                                                           \
        hipLaunchKernelGGL((pdhm::k_moffdiag<N, false>), grid, block, pdhm::lds_doubles_offdiag<N>() * sizeof(double),       \
                           stream, *P, mtab, count);                                                                 \
      return hipGetLastError();                                                                                      \
    }
  PDH_MOM_CASE(2) PDH_MOM_CASE(3) PDH_MOM_CASE(4)
#undef PDH_MOM_CASE
  return hipErrorInvalidValue;
}

extern "C" int pdh_moment_table_doubles(int n1d)
{
  switch (n1d)
    {
    case 2: return pdhm::MT<2>::SIZE;
    case 3: return pdhm::MT<3>::SIZE;
    case 4: return pdhm::MT<4>::SIZE;
    }
  return 0;
}

// Row kernel (pdh_rows.h): one wave per owned polytope writes all blocks of its rows; FE_DGQ(3) or FE_AggloDGP(3) in 3-D,
// axis-aligned planar faces.
extern "C" hipError_t pdh_launch_rows(const PdhDev *P, const PdhRows *R, const double *mtab, int count, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  // PDH_ROWS_LDS_PAD (bytes, diagnostics only): extra dynamic LDS per workgroup = fewer resident waves per CU, to see how
  // the kernel's time scales with occupancy (tools/README)
  static const size_t pad = [] {
    const char *e = getenv("PDH_ROWS_LDS_PAD");
    return e ? (size_t)atol(e) : (size_t)0;
  }();
  // persistent waves: as many single-wave workgroups as fit on the device at once (8 per CU by LDS and registers), each
  // working through slots blockIdx.x, blockIdx.x + gridDim.x, ...; PDH_ROWS_WAVES_PER_CU overrides (diagnostics)
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  static const int per_cu_env = [] {
    const char *e = getenv("PDH_ROWS_WAVES_PER_CU");
    return e ? atoi(e) : 0;
  }();
  const bool dgp = P->n == pdhr::DGP_N; // FE_AggloDGP(3): 26 KB of LDS per wave = 6 per CU
  const int per_cu = per_cu_env > 0 ? per_cu_env : (dgp ? 6 : 8);
  const int resident = cus * per_cu;
  const unsigned grid = (unsigned)(count < resident ? count : resident);
  constexpr size_t lds_q = pdhr::lds_doubles_rows<4, 0>() * sizeof(double), lds_p = pdhr::lds_doubles_rows<4, 1>() * sizeof(double);
  if (dgp)
    hipLaunchKernelGGL((pdhr::k_rows<4, 1>), dim3(grid), dim3(PDH_WAVE), lds_p + pad, stream, *P, *R, mtab, count);
  else
    hipLaunchKernelGGL((pdhr::k_rows<4, 0>), dim3(grid), dim3(PDH_WAVE), lds_q + pad, stream, *P, *R, mtab, count);
  return hipGetLastError();
}
extern "C" int pdh_rows_max_faces(void) { return pdhr::MAXF; }
