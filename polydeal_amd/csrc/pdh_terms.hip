// pdh_terms.hip — instantiations and launcher of the term kernel (pdh_terms.h): 3-D FE_DGQ(1,2), FE_AggloDGP(1..3).
#include <cstdlib>
#include "pdh_terms.h"
#include "pdh_terms_wg.h"

namespace
{
template <class F>
bool for_kind(int n1d, int basis, F &&f)
{
  using std::integral_constant;
  if (n1d == 4 && basis == 1)
    f(integral_constant<int, 4>{}, integral_constant<int, 1>{});
  else if (n1d == 3 && basis == 0)
    f(integral_constant<int, 3>{}, integral_constant<int, 0>{});
  else if (n1d == 3 && basis == 1)
    f(integral_constant<int, 3>{}, integral_constant<int, 1>{});
  else if (n1d == 2 && basis == 0)
    f(integral_constant<int, 2>{}, integral_constant<int, 0>{});
  else if (n1d == 2 && basis == 1)
    f(integral_constant<int, 2>{}, integral_constant<int, 1>{});
  else
    return false;
  return true;
}
} // namespace

// 1 if the term kernel (pdh_terms.h, a wave per polytope) is instantiated for this element, 2: FE_DGQ(3), which has the
// workgroup-per-polytope kernel of pdh_terms_wg.h instead
extern "C" int pdh_terms_has_kind(int n1d, int basis)
{
  if (n1d == 4 && basis == 0)
    return 2;
  return for_kind(n1d, basis, [](auto, auto) {}) ? 1 : 0;
}

// dynamic LDS of a workgroup for the maxima of a resident problem, bytes (0: no such kind)
// split: the two-phase form of the wave-per-polytope kernel (pdh_terms.h: SPLIT; ignored for the workgroup kernel)
extern "C" int pdh_terms_lds_bytes(int n1d, int basis, int maxruns, int maxsf, int maxsi, int maxcell, int split, int task_pts)
{
  int bytes = 0;
  (void)task_pts;
  if (n1d == 4 && basis == 0)
    return 8 * pdht::terms_lds_doubles<4, 0, false>(maxruns, maxsf, maxsi, maxcell);
  for_kind(n1d, basis, [&](auto n_, auto b_) {
    constexpr int N = decltype(n_)::value, B = decltype(b_)::value;
    bytes = 8 * (split ? pdht::terms_lds_doubles<N, B, true, true>(maxruns, maxsf, maxsi, maxcell)
                       : pdht::terms_lds_doubles<N, B, true, false>(maxruns, maxsf, maxsi, maxcell));
  });
  return bytes;
}

// Set-up, once per problem: the records of 1-D rules (PdhTerms::tdata) from the point arrays resident in HBM
__global__ void __launch_bounds__(PDH_WAVE) k_terms_gather(const PdhDev P, const PdhTerms T, double *__restrict__ out, const int n_owned)
{
  const int slot = blockIdx.x;
  if (slot < n_owned)
    pdht::terms_gather_record(P, T, slot, threadIdx.x, PDH_WAVE, out + (int64_t)slot * T.tstride);
}
extern "C" hipError_t pdh_launch_terms_gather(const PdhDev *P, const PdhTerms *T, double *out, int count, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_terms_gather, dim3((unsigned)count), dim3(PDH_WAVE), 0, stream, *P, *T, out, count);
  return hipGetLastError();
}
// doubles of a polytope's record of 1-D rules
extern "C" int pdh_terms_task_doubles(int maxsf, int maxcell, int pm) { return pdht::terms_task_doubles(maxsf, maxcell, pm); }

extern "C" hipError_t pdh_launch_terms(const PdhDev *P, const PdhTerms *T, int count, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const int full = P->n1d * P->n1d * P->n1d;
  const int basis = P->n == full ? 0 : 1;
  hipError_t rc = hipErrorInvalidValue;
  if (P->n1d == 4 && basis == 0)
    {
      // FE_DGQ(3): a workgroup of W waves per polytope (pdh_terms_wg.h); PDH_TERMS_WG_WAVES = 4 | 8 (diagnostics)
      static const int waves = [] {
        const char *e = getenv("PDH_TERMS_WG_WAVES");
        const int w = e ? atoi(e) : 4;
        return w == 8 ? 8 : 4;
      }();
      const size_t lds = (size_t)T->lds_bytes;
      const bool small = T->task_pts <= 4;
      auto go = [&](auto w_, auto shifted_, auto pmax_) {
        constexpr int W = decltype(w_)::value;
        hipLaunchKernelGGL((pdht::k_terms_wg<W, decltype(shifted_)::value, decltype(pmax_)::value>), dim3((unsigned)count), dim3(PDH_WAVE * W),
                           lds, stream, *P, *T, count);
      };
      using std::integral_constant;
      auto go_w = [&](auto w_) {
        if (P->diag_first && small)
          go(w_, std::true_type{}, integral_constant<int, 4>{});
        else if (P->diag_first)
          go(w_, std::true_type{}, integral_constant<int, 8>{});
        else if (small)
          go(w_, std::false_type{}, integral_constant<int, 4>{});
        else
          go(w_, std::false_type{}, integral_constant<int, 8>{});
      };
      if (waves == 8)
        go_w(integral_constant<int, 8>{});
      else
        go_w(integral_constant<int, 4>{});
      return hipGetLastError();
    }
  for_kind(P->n1d, basis, [&](auto n_, auto b_) {
    constexpr int N = decltype(n_)::value, B = decltype(b_)::value;
    if (P->n != pdht::Kind<N, B>::NF)
      return;
    const size_t lds = (size_t)T->lds_bytes;
    const bool small = T->task_pts <= 4; // (rules of up to 4 / up to 8 points per direction)
    auto go = [&](auto shifted_, auto pmax_) {
      constexpr bool S = decltype(shifted_)::value;
      constexpr int PM = decltype(pmax_)::value;
      if (T->split)
        hipLaunchKernelGGL((pdht::k_terms<N, B, S, PM, true>), dim3((unsigned)count), dim3(PDH_WAVE), lds, stream, *P, *T, count);
      else
        hipLaunchKernelGGL((pdht::k_terms<N, B, S, PM, false>), dim3((unsigned)count), dim3(PDH_WAVE), lds, stream, *P, *T, count);
    };
    using std::integral_constant;
    if (P->diag_first && small)
      go(std::true_type{}, integral_constant<int, 4>{});
    else if (P->diag_first)
      go(std::true_type{}, integral_constant<int, 8>{});
    else if (small)
      go(std::false_type{}, integral_constant<int, 4>{});
    else
      go(std::false_type{}, integral_constant<int, 8>{});
    rc = hipGetLastError();
  });
  return rc;
}
