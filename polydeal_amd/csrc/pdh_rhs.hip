// pdh_rhs.hip — right-hand-side kernel of the SIP path.
//
//   rhs_i(P) = sum_q phi_i f(x_q) JxW                                          reference examples/poisson.cc:745-759
//            + sum_{q on boundary faces of P} (sigma g phi_i - grad phi_i . n g) JxW          examples/poisson.cc:788-828
//
// One wavefront per owned polytope, lanes = basis functions (n <= 64).  Per chunk of 64 quadrature points each
// lane first evaluates the 1-D basis of one POINT (lanes = points), then every lane runs over the chunk's points and
// accumulates ITS function's value - all lanes read the same point's entries.
//   volume: the point phase stores  F0[k0] = f JxW B0_k0  and  P12[k1,k2] = B1_k1 B2_k2, so a (function, point) pair costs
//           two LDS reads and ONE multiply-add;
//   faces : only chunks that hold boundary points are visited (interior faces carry no datum); value and normal derivative
//           from the full 1-D records.
// Reads 8 (d + 2) bytes per volume point; bound by LDS reads + multiply-adds of the pair loop, not by HBM.
#include "pdh_kernels.h"

namespace pdh
{
template <int DIM, int N1D>
constexpr int rhs_lds_doubles()
{
  constexpr int face = Rec<DIM, N1D>::LEN + 1 + DIM;
  constexpr int vol = (N1D + (DIM == 3 ? N1D * N1D : N1D)) | 1;
  return PDH_WAVE * (face > vol ? face : vol);
}

template <int DIM, int N1D>
__global__ void __launch_bounds__(PDH_WAVE) k_rhs(const PdhDev P, const int n_owned, const double *__restrict__ f_vol,
                                                  const double *__restrict__ g_face, double *__restrict__ rhs,
                                                  const int64_t *__restrict__ vq_src, const int64_t *__restrict__ ap_src)
{
  // f_vol / g_face are indexed in the CALLER's point order (the order of vq_x / fq_x of the description): vq_src[slot] is
  // the caller index of the slot's first volume point, ap_src[q] the caller index of packed face point q (-1: not a boundary
  // point).  nullptr maps: the arrays are already in packed order.
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using RC = Rec<DIM, N1D>;
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;
  if (slot >= n_owned)
    return;
  const int agg = P.own_agg[slot];
  double *rec = lds;                      // [64][RC::LEN]
  double *aux = lds + PDH_WAVE * RC::LEN; // [64][1+DIM]
  constexpr int AUXN = 1 + DIM;
  // volume records: [64][VLEN] = F0[N1D] | P12[N1D^(DIM-1)]; odd stride
  constexpr int NP12 = DIM == 3 ? N1D * N1D : N1D;
  constexpr int VLEN = (N1D + NP12) | 1;
  // (the launcher sizes the LDS for the larger of the two record kinds: rhs_lds_doubles)
  double lo[DIM], h[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      h[c] = P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - lo[c];
    }
  // this lane's basis function
  const bool live = lane < P.n;
  int off[DIM];
  {
    const uint32_t packed = live ? (uint32_t)P.midx[lane] : 0u;
    for (int c = 0; c < DIM; ++c)
      off[c] = live ? (c * N1D + (int)((packed >> (8 * c)) & 0xff)) * 2 : RC::ZERO_OFF / 8;
  }
  double acc = 0.0;
  // offsets of this lane's function into a volume record
  int voff0 = 0, voff12 = N1D;
  {
    const uint32_t packed = live ? (uint32_t)P.midx[lane] : 0u;
    const int k0 = (int)(packed & 0xff), k1 = (int)((packed >> 8) & 0xff), k2 = (int)((packed >> 16) & 0xff);
    voff0 = k0;
    voff12 = N1D + k1 + (DIM == 3 ? N1D * k2 : 0);
  }

  // volume: sum_q phi_i f JxW
  if (f_vol)
    {
      const int64_t qb = P.vq_ptr[slot], qe = P.vq_ptr[slot + 1];
      for (int64_t base = qb; base < qe; base += PDH_WAVE)
        {
          const int cnt = (int)((qe - base < PDH_WAVE) ? (qe - base) : PDH_WAVE);
          PDH_WAVE_SYNC();
          {
            // lanes = points (dead lanes: zero weight, any valid coordinates)
            const bool on = lane < cnt;
            double x[DIM];
            for (int c = 0; c < DIM; ++c)
              x[c] = on ? P.vq_x[c * P.vq_stride + base + lane] : lo[c];
            const int64_t src = vq_src ? vq_src[slot] + (base + lane - qb) : base + lane;
            const double fw = on ? f_vol[src] * P.vq_w[base + lane] : 0.0;
            double r[RC::LEN];
            eval_point_record<DIM, N1D, false>(P.tab, lo, h, x, 1.0, nullptr, r);
            double *v = lds + lane * VLEN;
            static_for<0, N1D>([&](auto k_) {
              constexpr int k = k_;
              v[k] = fw * r[(0 * N1D + k) * 2];
            });
            static_for<0, NP12>([&](auto m_) {
              constexpr int m = m_;
              constexpr int k1 = m % N1D, k2 = m / N1D;
              if constexpr (DIM == 3)
                v[N1D + m] = r[(1 * N1D + k1) * 2] * r[(2 * N1D + k2) * 2];
              else
                v[N1D + m] = r[(1 * N1D + k1) * 2];
            });
          }
          PDH_WAVE_SYNC();
          // (all 64 records are valid: dead points contribute zero)
#pragma unroll 8
          for (int q = 0; q < PDH_WAVE; ++q)
            acc += lds[q * VLEN + voff0] * lds[q * VLEN + voff12];
        }
    }

  // Nitsche boundary terms; points of interior faces carry g = 0 (set by the host)
  if (g_face)
    {
      const int64_t pb = P.ap_ptr[slot], pe = P.ap_ptr[slot + 1];
      for (int64_t base = pb; base < pe; base += PDH_WAVE)
        {
          const int cnt = (int)((pe - base < PDH_WAVE) ? (pe - base) : PDH_WAVE);
          // a chunk without boundary points contributes nothing (ap_src = -1 on interior faces)
          const int64_t src_l = lane < cnt ? (ap_src ? ap_src[base + lane] : base + lane) : -1;
          if (__ballot(src_l >= 0) == 0ull)
            continue;
          PDH_WAVE_SYNC();
          if (lane < cnt)
            {
              double x[DIM];
              for (int c = 0; c < DIM; ++c)
                x[c] = P.ap_x[c * P.ap_stride + base + lane];
              eval_point_record<DIM, N1D, false>(P.tab, lo, h, x, 1.0, nullptr, rec + lane * RC::LEN);
              // packed boundary points carry w = 2 JxW and sigma/2:  2w (sigma/2 g phi - 1/2 g grad phi.n)
              const int64_t src = src_l;
              const double gw = (src >= 0 ? g_face[src] : 0.0) * P.ap_wself[base + lane];
              aux[lane * AUXN] = gw * P.ap_sig[base + lane];
              for (int c = 0; c < DIM; ++c)
                aux[lane * AUXN + 1 + c] = -0.5 * gw * P.ap_n[c * P.ap_stride + base + lane];
            }
          PDH_WAVE_SYNC();
          for (int q = 0; q < cnt; ++q)
            {
              const double *r = rec + q * RC::LEN;
              const double *a = aux + q * AUXN;
              double v[DIM], d[DIM];
              for (int c = 0; c < DIM; ++c)
                {
                  v[c] = r[off[c]];
                  d[c] = r[off[c] + 1];
                }
              double phi = v[0];
              for (int c = 1; c < DIM; ++c)
                phi *= v[c];
              double s = a[0] * phi;
              for (int g = 0; g < DIM; ++g)
                {
                  double t = d[g];
                  for (int c = 0; c < DIM; ++c)
                    if (c != g)
                      t *= v[c];
                  s += a[1 + g] * t;
                }
              acc += s;
            }
        }
    }
  if (live)
    rhs[(int64_t)P.own_row[slot] + lane] = acc;
}
} // namespace pdh

extern "C" hipError_t pdh_launch_rhs(int dim, int n1d, const PdhDev *P, int count, const double *f_vol,
                                     const double *g_face, double *rhs, const int64_t *vq_src, const int64_t *ap_src,
                                     hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count), block(PDH_WAVE);
#define PDH_RHS_CASE(D, N)                                                                                 \
  if (dim == D && n1d == N)                                                                                \
    {                                                                                                      \
      const size_t lds = (size_t)pdh::rhs_lds_doubles<D, N>() * sizeof(double);                            \
      hipLaunchKernelGGL((pdh::k_rhs<D, N>), grid, block, lds, stream, *P, count, f_vol, g_face, rhs, vq_src, ap_src);      \
      return hipGetLastError();                                                                            \
    }
  PDH_RHS_CASE(2, 1) PDH_RHS_CASE(2, 2) PDH_RHS_CASE(2, 3) PDH_RHS_CASE(2, 4)
  PDH_RHS_CASE(2, 5) PDH_RHS_CASE(2, 6) PDH_RHS_CASE(2, 7) PDH_RHS_CASE(2, 8)
  PDH_RHS_CASE(3, 1) PDH_RHS_CASE(3, 2) PDH_RHS_CASE(3, 3) PDH_RHS_CASE(3, 4)
  PDH_RHS_CASE(3, 5) PDH_RHS_CASE(3, 6)
#undef PDH_RHS_CASE
  return hipErrorInvalidValue;
}
