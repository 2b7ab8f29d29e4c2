// pdh_eval.hip — evaluation of a polytopal DG function (and its gradient) at caller-given points, per
// polytope, in the bounding-box frame.  It is the device part of the reference's
// post-processing: PolyUtils::interpolate_to_fine_grid evaluates the polytopal solution at the support points
// of the sub-cells (include/poly_utils.h:1145-1274) and PolyUtils::compute_global_error at the quadrature
// points (include/poly_utils.h:1686-1731); the weighted sums over points stay with the caller.
//
// One wavefront per owned polytope, lanes = points.  Each lane evaluates the 1-D basis records of its own
// point into LDS (same records as the assembly kernels), then runs over the n basis functions: the multi-index
// and the coefficient of function i are wave-uniform, the three table entries come from the lane's own record.
#include "pdh_kernels.h"

namespace pdh
{
template <int DIM, int N1D, bool GRAD>
__global__ void __launch_bounds__(PDH_WAVE) k_eval(const PdhDev P, const int n_owned, const double *__restrict__ coef,
                                                   const int64_t *__restrict__ pt_ptr, const double *__restrict__ pts,
                                                   const int64_t pts_stride, double *__restrict__ out_u,
                                                   double *__restrict__ out_g, const int by_agg)
{
  // by_agg: pt_ptr / pts / out are the CALLER's arrays, indexed by the polytope numbers of the description ([n_agg+1]);
  // otherwise they are compacted over the owned slots
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using RC = Rec<DIM, N1D>;
  constexpr int STRIDE = RC::LEN + 1; // odd number of doubles per lane: own-record reads are conflict-free
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;
  if (slot >= n_owned)
    return;
  const int agg = P.own_agg[slot];
  double lo[DIM], h[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      h[c] = P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - lo[c];
    }
  const double *cf = coef + P.own_row[slot];
  double *rec = lds + lane * STRIDE;
  const int64_t qb = pt_ptr[by_agg ? agg : slot], qe = pt_ptr[(by_agg ? agg : slot) + 1];
  for (int64_t base = qb; base < qe; base += PDH_WAVE)
    {
      const int64_t q = base + lane;
      const bool on = q < qe;
      double x[DIM];
      for (int c = 0; c < DIM; ++c)
        x[c] = on ? pts[c * pts_stride + q] : lo[c];
      eval_point_record<DIM, N1D, false>(P.tab, lo, h, x, 1.0, nullptr, rec);
      double u = 0.0, g[DIM];
      for (int c = 0; c < DIM; ++c)
        g[c] = 0.0;
      for (int i = 0; i < P.n; ++i)
        {
          const uint32_t packed = (uint32_t)P.midx[i]; // wave-uniform
          const double ci = cf[i];                     // wave-uniform
          double v[DIM], d[DIM];
          for (int c = 0; c < DIM; ++c)
            {
              const int k = (int)((packed >> (8 * c)) & 0xff);
              v[c] = rec[(c * N1D + k) * 2];
              d[c] = rec[(c * N1D + k) * 2 + 1];
            }
          double phi = v[0];
          for (int c = 1; c < DIM; ++c)
            phi *= v[c];
          u += ci * phi;
          if constexpr (GRAD)
            for (int gc = 0; gc < DIM; ++gc)
              {
                double t = d[gc];
                for (int c = 0; c < DIM; ++c)
                  if (c != gc)
                    t *= v[c];
                g[gc] += ci * t;
              }
        }
      if (on)
        {
          out_u[q] = u;
          if constexpr (GRAD)
            for (int c = 0; c < DIM; ++c)
              out_g[c * pts_stride + q] = g[c];
        }
    }
}

// Values of all n basis functions of box `blockIdx.x` at that box's points:  out[q][j] = phi_j(x_q).
// This is the local matrix of Utils::fill_injection_matrix (reference include/utils.h:219-229:
// local_matrix(i,j) = fe.shape_value(j, coarse_bbox.real_to_unit(real_qpoints[i]))).  Phase 1: lanes = points
// (1-D records into LDS); phase 2: lanes = basis functions, rows of `out` are written coalesced.
template <int DIM, int N1D>
__global__ void __launch_bounds__(PDH_WAVE) k_shape(const PdhDev P, const int n_boxes, const int64_t *__restrict__ pt_ptr,
                                                    const double *__restrict__ pts, const int64_t pts_stride,
                                                    double *__restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using RC = Rec<DIM, N1D>;
  const int lane = threadIdx.x;
  const int box = blockIdx.x;
  if (box >= n_boxes)
    return;
  double lo[DIM], h[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)box * 2 * DIM + c];
      h[c] = P.bbox[(int64_t)box * 2 * DIM + DIM + c] - lo[c];
    }
  const bool live = lane < P.n;
  int off[DIM];
  {
    const uint32_t packed = live ? (uint32_t)P.midx[lane] : 0u;
    for (int c = 0; c < DIM; ++c)
      off[c] = (c * N1D + (int)((packed >> (8 * c)) & 0xff)) * 2;
  }
  const int64_t qb = pt_ptr[box], qe = pt_ptr[box + 1];
  for (int64_t base = qb; base < qe; base += PDH_WAVE)
    {
      const int cnt = (int)((qe - base < PDH_WAVE) ? (qe - base) : PDH_WAVE);
      __syncthreads();
      if (lane < cnt)
        {
          double x[DIM];
          for (int c = 0; c < DIM; ++c)
            x[c] = pts[c * pts_stride + base + lane];
          eval_point_record<DIM, N1D, false>(P.tab, lo, h, x, 1.0, nullptr, lds + lane * RC::LEN);
        }
      __syncthreads();
      if (live)
        for (int q = 0; q < cnt; ++q)
          {
            const double *r = lds + q * RC::LEN;
            double phi = r[off[0]];
            for (int c = 1; c < DIM; ++c)
              phi *= r[off[c]];
            out[(base + q) * P.n + lane] = phi;
          }
    }
}
} // namespace pdh

extern "C" hipError_t pdh_launch_shape(int dim, int n1d, const PdhDev *P, int n_boxes, const int64_t *pt_ptr,
                                       const double *pts, int64_t pts_stride, double *out, hipStream_t stream)
{
  if (n_boxes <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)n_boxes), block(PDH_WAVE);
#define PDH_SHAPE_CASE(D, N)                                                                                         \
  if (dim == D && n1d == N)                                                                                          \
    {                                                                                                                \
      const size_t lds = (size_t)PDH_WAVE * pdh::Rec<D, N>::LEN * sizeof(double);                                    \
      hipLaunchKernelGGL((pdh::k_shape<D, N>), grid, block, lds, stream, *P, n_boxes, pt_ptr, pts, pts_stride, out);  \
      return hipGetLastError();                                                                                      \
    }
  PDH_SHAPE_CASE(2, 1) PDH_SHAPE_CASE(2, 2) PDH_SHAPE_CASE(2, 3) PDH_SHAPE_CASE(2, 4)
  PDH_SHAPE_CASE(2, 5) PDH_SHAPE_CASE(2, 6) PDH_SHAPE_CASE(2, 7) PDH_SHAPE_CASE(2, 8)
  PDH_SHAPE_CASE(3, 1) PDH_SHAPE_CASE(3, 2) PDH_SHAPE_CASE(3, 3) PDH_SHAPE_CASE(3, 4)
  PDH_SHAPE_CASE(3, 5) PDH_SHAPE_CASE(3, 6)
#undef PDH_SHAPE_CASE
  return hipErrorInvalidValue;
}

extern "C" hipError_t pdh_launch_eval(int dim, int n1d, int grad, const PdhDev *P, int count, const double *coef,
                                      const int64_t *pt_ptr, const double *pts, int64_t pts_stride, double *out_u,
                                      double *out_g, int by_agg, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count), block(PDH_WAVE);
#define PDH_EVAL_CASE(D, N)                                                                                          \
  if (dim == D && n1d == N)                                                                                          \
    {                                                                                                                \
      const size_t lds = (size_t)PDH_WAVE * (pdh::Rec<D, N>::LEN + 1) * sizeof(double);                              \
      if (grad)                                                                                                      \
        hipLaunchKernelGGL((pdh::k_eval<D, N, true>), grid, block, lds, stream, *P, count, coef, pt_ptr, pts,         \
                           pts_stride, out_u, out_g, by_agg);                                                        \
      else                                                                                                           \
        hipLaunchKernelGGL((pdh::k_eval<D, N, false>), grid, block, lds, stream, *P, count, coef, pt_ptr, pts,        \
                           pts_stride, out_u, out_g, by_agg);                                                        \
      return hipGetLastError();                                                                                      \
    }
  PDH_EVAL_CASE(2, 1) PDH_EVAL_CASE(2, 2) PDH_EVAL_CASE(2, 3) PDH_EVAL_CASE(2, 4)
  PDH_EVAL_CASE(2, 5) PDH_EVAL_CASE(2, 6) PDH_EVAL_CASE(2, 7) PDH_EVAL_CASE(2, 8)
  PDH_EVAL_CASE(3, 1) PDH_EVAL_CASE(3, 2) PDH_EVAL_CASE(3, 3) PDH_EVAL_CASE(3, 4)
  PDH_EVAL_CASE(3, 5) PDH_EVAL_CASE(3, 6)
#undef PDH_EVAL_CASE
  return hipErrorInvalidValue;
}
