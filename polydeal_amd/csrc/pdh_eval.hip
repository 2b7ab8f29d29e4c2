// pdh_eval.hip — evaluation of a polytopal DG function (and its gradient) at caller-given points, per
// polytope, in the bounding-box frame.  It is the device part of the reference's
// post-processing: PolyUtils::interpolate_to_fine_grid evaluates the polytopal solution at the support points
// of the sub-cells (include/poly_utils.h:1145-1274) and PolyUtils::compute_global_error at the quadrature
// points (include/poly_utils.h:1686-1731); the weighted sums over points stay with the caller.
//
// One wavefront per owned polytope, lanes = points.  Each lane evaluates the 1-D basis values and derivatives of its own
// point into registers and sums over the basis by sum factorisation - both bases are products of 1-D functions numbered
// with the x index fastest (pdh_basis.h: multi_indices; FE_AggloDGP restricts the index set to total degree <= p):
//   u = sum_k2 B2_k2 ( sum_k1 B1_k1 ( sum_k0 B0_k0 c_(k0,k1,k2) ) ),   the gradient shares the inner sums
// - 2 n + 3 N1D^2 + 4 N1D multiply-adds per point with the gradient (192 for FE_DGQ(3)) instead of ~14 n, no per-function
// LDS traffic; the polytope's coefficients are read as LDS broadcasts at compile-time offsets.  HBM-bound: 8 (d + 1) bytes
// read and 8 (1 + d) written per point.
#include "pdh_kernels.h"

namespace pdh
{
// number of basis functions that precede multi-index (k0, k1, k2) in the numbering of pdh_basis.h
template <int DIM, int N1D, int BASIS>
__host__ __device__ constexpr int function_index(int k0, int k1, int k2)
{
  if (BASIS == 0)
    return k0 + N1D * (k1 + (DIM == 3 ? N1D * k2 : 0));
  int cnt = 0;
  if (DIM == 2)
    {
      for (int iy = 0; iy < N1D; ++iy)
        for (int ix = 0; ix < N1D - iy; ++ix)
          {
            if (ix == k0 && iy == k1)
              return cnt;
            ++cnt;
          }
      return -1;
    }
  for (int iz = 0; iz < N1D; ++iz)
    for (int iy = 0; iy < N1D - iz; ++iy)
      for (int ix = 0; ix < N1D - iy - iz; ++ix)
        {
          if (ix == k0 && iy == k1 && iz == k2)
            return cnt;
          ++cnt;
        }
  return -1;
}

// ERR: instead of writing u_h / grad u_h, form the squared error sums of PolyUtils::compute_global_error (reference
// include/poly_utils.h:1699-1745) per polytope:  err[slot] = { sum_q w (u - u_h)^2, sum_q w |grad u - grad u_h|^2 }  with the
// exact values sampled by the caller at the same points - nothing per point leaves the device.
struct EvalErr
{
  const double *w, *exact_u, *exact_g; // [N], [N], [dim][N]
  double *err;                         // [n_owned][2]
};
template <int DIM, int N1D, bool GRAD, int BASIS, bool ERR = false>
__global__ void __launch_bounds__(PDH_WAVE) k_eval(const PdhDev P, const int n_owned, const double *__restrict__ coef,
                                                   const int64_t *__restrict__ pt_ptr, const double *__restrict__ pts,
                                                   const int64_t pts_stride, double *__restrict__ out_u,
                                                   double *__restrict__ out_g, const int by_agg, const EvalErr E = EvalErr{})
{
  // by_agg: pt_ptr / pts / out are the CALLER's arrays, indexed by the polytope numbers of the description ([n_agg+1]);
  // otherwise they are compacted over the owned slots
  extern __shared__ __attribute__((aligned(16))) double lds[]; // [n] coefficients of the polytope
  using RC = Rec<DIM, N1D>;
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
  for (int i = lane; i < P.n; i += PDH_WAVE) // (n > 64: FE_DGQ(4..7) / FE_AggloDGP(6, 7) in 3-D)
    lds[i] = coef[P.own_row[slot] + i];
  PDH_WAVE_SYNC();
  const int64_t qb = pt_ptr[by_agg ? agg : slot], qe = pt_ptr[(by_agg ? agg : slot) + 1];
  double e_l2 = 0.0, e_h1 = 0.0;
  for (int64_t base = qb; base < qe; base += PDH_WAVE)
    {
      const int64_t q = base + lane;
      const bool on = q < qe;
      double x[DIM];
      for (int c = 0; c < DIM; ++c)
        x[c] = on ? pts[c * pts_stride + q] : lo[c];
      // (error mode: the exact values of this point are requested before the evaluation, consumed after it)
      double ew = 0.0, eu = 0.0, eg[DIM];
      for (int c = 0; c < DIM; ++c)
        eg[c] = 0.0;
      if constexpr (ERR)
        if (on)
          {
            ew = E.w[q];
            eu = E.exact_u[q];
            if constexpr (GRAD)
              for (int c = 0; c < DIM; ++c)
                eg[c] = E.exact_g[c * pts_stride + q];
          }
      double r[RC::LEN]; // (value, derivative / h) of every 1-D function, in registers (static indices only)
      eval_point_record<DIM, N1D, false>(P.tab, lo, h, x, 1.0, nullptr, r);
      // (the coefficient reads stay inside the loop: hoisted, the n coefficients would occupy 2 n VGPRs for the whole kernel -
      // 208 registers, two waves per SIMD - where a streaming kernel wants many waves in flight; broadcasts are cheap)
      int zoff = 0;
      asm volatile("" : "+v"(zoff));
      const double *cfl = lds + zoff;
      double u = 0.0, g[3] = {0.0, 0.0, 0.0};
      constexpr int NZ = DIM == 3 ? N1D : 1;
      static_for<0, NZ>([&](auto k2_) {
        constexpr int k2 = k2_;
        constexpr int NY = BASIS == 0 ? N1D : N1D - k2;
        double a = 0.0, ad0 = 0.0, ad1 = 0.0;
        static_for<0, NY>([&](auto k1_) {
          constexpr int k1 = k1_;
          constexpr int NX = BASIS == 0 ? N1D : N1D - k1 - k2;
          double s0 = 0.0, sd = 0.0;
          static_for<0, NX>([&](auto k0_) {
            constexpr int k0 = k0_;
            const double c = cfl[function_index<DIM, N1D, BASIS>(k0, k1, k2)];
            s0 += r[(0 * N1D + k0) * 2] * c;
            if constexpr (GRAD)
              sd += r[(0 * N1D + k0) * 2 + 1] * c;
          });
          a += r[(1 * N1D + k1) * 2] * s0;
          if constexpr (GRAD)
            {
              ad0 += r[(1 * N1D + k1) * 2] * sd;
              ad1 += r[(1 * N1D + k1) * 2 + 1] * s0;
            }
        });
        if constexpr (DIM == 3)
          {
            u += r[(2 * N1D + k2) * 2] * a;
            if constexpr (GRAD)
              {
                g[0] += r[(2 * N1D + k2) * 2] * ad0;
                g[1] += r[(2 * N1D + k2) * 2] * ad1;
                g[2] += r[(2 * N1D + k2) * 2 + 1] * a;
              }
          }
        else
          {
            u += a;
            if constexpr (GRAD)
              {
                g[0] += ad0;
                g[1] += ad1;
              }
          }
      });
      if constexpr (ERR)
        {
          e_l2 += ew * (eu - u) * (eu - u); // ew = 0 on lanes without a point
          if constexpr (GRAD)
            {
              double s = 0.0;
              for (int c = 0; c < DIM; ++c)
                s += (eg[c] - g[c]) * (eg[c] - g[c]);
              e_h1 += ew * s;
            }
        }
      else if (on)
        {
          out_u[q] = u;
          if constexpr (GRAD)
            for (int c = 0; c < DIM; ++c)
              out_g[c * pts_stride + q] = g[c];
        }
    }
  if constexpr (ERR)
    {
      // fixed-order butterfly over the wave: the sums of a polytope do not depend on anything but its points
      for (int off = 32; off > 0; off >>= 1)
        {
          e_l2 += __shfl_xor(e_l2, off);
          e_h1 += __shfl_xor(e_h1, off);
        }
      if (lane == 0)
        {
          E.err[2 * (int64_t)slot] = e_l2;
          E.err[2 * (int64_t)slot + 1] = e_h1;
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
  const int fi = (int)blockIdx.y * PDH_WAVE + lane; // blockIdx.y: 64 functions each (n > 64)
  const bool live = fi < P.n;
  int off[DIM];
  {
    const uint32_t packed = live ? (uint32_t)P.midx[fi] : 0u;
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
            out[(base + q) * P.n + fi] = phi;
          }
    }
}
} // namespace pdh

extern "C" hipError_t pdh_launch_shape(int dim, int n1d, const PdhDev *P, int n_boxes, const int64_t *pt_ptr,
                                       const double *pts, int64_t pts_stride, double *out, hipStream_t stream)
{
  if (n_boxes <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)n_boxes, (unsigned)((P->n + PDH_WAVE - 1) / PDH_WAVE)), block(PDH_WAVE);
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
  PDH_SHAPE_CASE(3, 5) PDH_SHAPE_CASE(3, 6) PDH_SHAPE_CASE(3, 7) PDH_SHAPE_CASE(3, 8)
#undef PDH_SHAPE_CASE
  return hipErrorInvalidValue;
}

extern "C" hipError_t pdh_launch_eval_err(int dim, int n1d, const PdhDev *P, int count, const double *coef, const int64_t *pt_ptr,
                                          const double *pts, int64_t pts_stride, const double *w, const double *exact_u,
                                          const double *exact_g, double *err, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count), block(PDH_WAVE);
  int full = 1;
  for (int c = 0; c < dim; ++c)
    full *= n1d;
  const bool dgq = P->n == full;
  const pdh::EvalErr E{w, exact_u, exact_g, err};
  const size_t lds = (size_t)((P->n + PDH_WAVE - 1) / PDH_WAVE) * PDH_WAVE * sizeof(double);
#define PDH_ERR_CASE(D, N)                                                                                           \
  if (dim == D && n1d == N)                                                                                          \
    {                                                                                                                \
      if (dgq)                                                                                                       \
        hipLaunchKernelGGL((pdh::k_eval<D, N, true, 0, true>), grid, block, lds, stream, *P, count, coef, pt_ptr, pts, \
                           pts_stride, nullptr, nullptr, 1, E);                                                      \
      else                                                                                                           \
        hipLaunchKernelGGL((pdh::k_eval<D, N, true, 1, true>), grid, block, lds, stream, *P, count, coef, pt_ptr, pts, \
                           pts_stride, nullptr, nullptr, 1, E);                                                      \
      return hipGetLastError();                                                                                      \
    }
  PDH_ERR_CASE(2, 1) PDH_ERR_CASE(2, 2) PDH_ERR_CASE(2, 3) PDH_ERR_CASE(2, 4)
  PDH_ERR_CASE(2, 5) PDH_ERR_CASE(2, 6) PDH_ERR_CASE(2, 7) PDH_ERR_CASE(2, 8)
  PDH_ERR_CASE(3, 1) PDH_ERR_CASE(3, 2) PDH_ERR_CASE(3, 3) PDH_ERR_CASE(3, 4)
  PDH_ERR_CASE(3, 5) PDH_ERR_CASE(3, 6) PDH_ERR_CASE(3, 7) PDH_ERR_CASE(3, 8)
#undef PDH_ERR_CASE
  return hipErrorInvalidValue;
}

extern "C" hipError_t pdh_launch_eval(int dim, int n1d, int grad, const PdhDev *P, int count, const double *coef,
                                      const int64_t *pt_ptr, const double *pts, int64_t pts_stride, double *out_u,
                                      double *out_g, int by_agg, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count), block(PDH_WAVE);
  int full = 1;
  for (int c = 0; c < dim; ++c)
    full *= n1d;
  const bool dgq = P->n == full; // tensor-product index set; else total degree <= p (FE_AggloDGP)
#define PDH_EVAL_LAUNCH(D, N, G, B)                                                                                  \
  hipLaunchKernelGGL((pdh::k_eval<D, N, G, B>), grid, block, lds, stream, *P, count, coef, pt_ptr, pts, pts_stride,   \
                     out_u, out_g, by_agg)
#define PDH_EVAL_CASE(D, N)                                                                                          \
  if (dim == D && n1d == N)                                                                                          \
    {                                                                                                                \
      const size_t lds = (size_t)((P->n + PDH_WAVE - 1) / PDH_WAVE) * PDH_WAVE * sizeof(double);                                                          \
      if (grad && dgq)                                                                                               \
        PDH_EVAL_LAUNCH(D, N, true, 0);                                                                              \
      else if (grad)                                                                                                 \
        PDH_EVAL_LAUNCH(D, N, true, 1);                                                                              \
      else if (dgq)                                                                                                  \
        PDH_EVAL_LAUNCH(D, N, false, 0);                                                                             \
      else                                                                                                           \
        PDH_EVAL_LAUNCH(D, N, false, 1);                                                                             \
      return hipGetLastError();                                                                                      \
    }
  PDH_EVAL_CASE(2, 1) PDH_EVAL_CASE(2, 2) PDH_EVAL_CASE(2, 3) PDH_EVAL_CASE(2, 4)
  PDH_EVAL_CASE(2, 5) PDH_EVAL_CASE(2, 6) PDH_EVAL_CASE(2, 7) PDH_EVAL_CASE(2, 8)
  PDH_EVAL_CASE(3, 1) PDH_EVAL_CASE(3, 2) PDH_EVAL_CASE(3, 3) PDH_EVAL_CASE(3, 4)
  PDH_EVAL_CASE(3, 5) PDH_EVAL_CASE(3, 6) PDH_EVAL_CASE(3, 7) PDH_EVAL_CASE(3, 8)
#undef PDH_EVAL_CASE
#undef PDH_EVAL_LAUNCH
  return hipErrorInvalidValue;
}
