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
  // kinds of the kernel (pdh_rows.h: RowsKind): FE_DGQ(3) is the row-piece kernel, the others stream their rows
  const int full = P->n1d * P->n1d * P->n1d;
  const int basis = P->n == full ? 0 : 1;
  hipError_t rc = hipErrorInvalidValue;
  auto launch = [&](auto n1d_, auto basis_) {
    constexpr int N = decltype(n1d_)::value, B = decltype(basis_)::value;
    if (P->n != pdhr::RowsKind<N, B>::NF)
      return;
    // MULTI instantiation (FE_DGQ(3), PdhRows::multi): the coupling-moment slots (one per interior plane entry) are sized for
    // the resident problem
    const bool multi = N == 4 && B == 0 && R->multi != 0;
    // (MULTI: the coupling moments of the interior entries are parked in PdhRows::m2c_scratch, not in LDS - the layout of the
    // block-shaped kernel, whose six slots serve as staging there)
    const size_t lds = pdhr::lds_doubles_rows<N, B>() * sizeof(double);
    // resident single-wave workgroups per CU: by LDS (160 KB, handed out in granules of 1280 bytes - measured: 26 624 bytes
    // fit six times, 27 136 do not) and by the registers of the instantiation (the runtime's occupancy query: two waves per
    // SIMD above 168 VGPRs, three up to 168)
    const size_t granules = (lds + pad + 1279) / 1280 * 1280;
    const int fit_lds = (int)(160 * 1024 / granules);
    // degree 3: the instantiation without general-point paths when the host verified tensor rules everywhere
    // (PdhRows::tensor_only, pdh_capi.cpp: rows_kind_applies)
    auto go = [&](auto general_, auto shifted_, auto multi_) {
      constexpr bool G = decltype(general_)::value, S = decltype(shifted_)::value, MU = decltype(multi_)::value;
      static thread_local size_t occ_lds = ~(size_t)0; // (per instantiation and thread)
      static thread_local int occ_fit = 8;
      if (occ_lds != lds + pad)
        {
          int nb = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pdhr::k_rows<N, B, G, S, MU>, PDH_WAVE, lds + pad) != hipSuccess || nb < 1)
            nb = 8;
          occ_fit = nb, occ_lds = lds + pad;
          if (getenv("PDH_ROWS_VERBOSE"))
            fprintf(stderr, "k_rows<%d,%d,%d,%d,%d>: lds %zu bytes, resident waves per CU: %d by LDS, %d by the occupancy query\n", N, B, (int)G,
                    (int)S, (int)MU, lds + pad, fit_lds, nb);
        }
      const int fit = fit_lds < occ_fit ? fit_lds : occ_fit;
      const int per_cu = per_cu_env > 0 ? per_cu_env : fit;
      int resident = cus * per_cu;
      if (MU && resident > R->scratch_waves)
        resident = R->scratch_waves; // (one row of the moment scratch per workgroup)
      const unsigned grid = (unsigned)(count < resident ? count : resident);
      // the work counter and the count of leavers start every launch at zero.  The last wave out of a launch resets them, but a
      // launch that was aborted, or two launches of one context overlapping after a change of stream, would leave them dirty -
      // and a dirty counter silently skips or repeats polytopes.  Eight bytes, stream-ordered in front of the kernel.
      if constexpr (!pdhr::RowsKind<N, B>::SMALL)
        if (hipMemsetAsync(R->sched, 0, 2 * sizeof(unsigned int), stream) != hipSuccess)
          return;
      hipLaunchKernelGGL((pdhr::k_rows<N, B, G, S, MU>), dim3(grid), dim3(PDH_WAVE), lds + pad, stream, *P, *R, mtab, count);
    };
    using std::true_type;
    using std::false_type;
    const bool general = N == 4 && !R->tensor_only, shifted = P->diag_first != 0;
    if constexpr (N == 4 && B == 0)
      if (multi)
        {
          if (general && shifted)
            go(true_type{}, true_type{}, true_type{});
          else if (general)
            go(true_type{}, false_type{}, true_type{});
          else if (shifted)
            go(false_type{}, true_type{}, true_type{});
          else
            go(false_type{}, false_type{}, true_type{});
          rc = hipGetLastError();
          return;
        }
    if constexpr (N == 4)
      {
        if (general && shifted)
          go(true_type{}, true_type{}, false_type{});
        else if (general)
          go(true_type{}, false_type{}, false_type{});
      }
    if (!general && shifted)
      go(false_type{}, true_type{}, false_type{});
    else if (!general)
      go(false_type{}, false_type{}, false_type{});
    rc = hipGetLastError();
  };
  using std::integral_constant;
  if (P->n1d == 4 && basis == 0)
    launch(integral_constant<int, 4>{}, integral_constant<int, 0>{});
  else if (P->n1d == 4)
    launch(integral_constant<int, 4>{}, integral_constant<int, 1>{});
  else if (P->n1d == 3 && basis == 0)
    launch(integral_constant<int, 3>{}, integral_constant<int, 0>{});
  else if (P->n1d == 3)
    launch(integral_constant<int, 3>{}, integral_constant<int, 1>{});
  else if (P->n1d == 2 && basis == 0)
    launch(integral_constant<int, 2>{}, integral_constant<int, 0>{});
  else if (P->n1d == 2)
    launch(integral_constant<int, 2>{}, integral_constant<int, 1>{});
  return rc;
}
extern "C" int pdh_rows_n_dofs(int n1d, int basis)
{
  switch (n1d * 2 + (basis ? 1 : 0))
    {
    case 8: return pdhr::RowsKind<4, 0>::NF;
    case 9: return pdhr::RowsKind<4, 1>::NF;
    case 6: return pdhr::RowsKind<3, 0>::NF;
    case 7: return pdhr::RowsKind<3, 1>::NF;
    case 4: return pdhr::RowsKind<2, 0>::NF;
    case 5: return pdhr::RowsKind<2, 1>::NF;
    }
  return 0;
}
extern "C" int pdh_rows_max_faces(void) { return pdhr::MAXF; }
