// pdh_rhs.hip — right-hand-side kernel of the SIP path.
//
//   rhs_i(P) = sum_q phi_i f(x_q) JxW                                          reference examples/poisson.cc:745-759
//            + sum_{q on boundary faces of P} (sigma g phi_i - grad phi_i . n g) JxW          examples/poisson.cc:788-828
//
// One wavefront per owned polytope (and per 64 basis functions of it: blockIdx.y, for n > 64), lanes = basis functions.  Per chunk of 64 quadrature points each
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
// points per chunk of the face loop: 16 where the volume part runs on the MFMA and needs 2 KB of LDS only (then 4 KB per
// wave instead of 15 KB: the kernel is bound by the latency of its point loads, it wants waves)
template <int DIM, int N1D>
constexpr int rhs_face_chunk()
{
  return (DIM == 3 && N1D == 4) ? 16 : PDH_WAVE;
}
template <int DIM, int N1D>
constexpr int rhs_lds_doubles()
{
  constexpr int face = rhs_face_chunk<DIM, N1D>() * (Rec<DIM, N1D>::LEN + 1 + DIM);
  constexpr int vol = (DIM == 3 && N1D == 4) ? PDH_WAVE * 4 : PDH_WAVE * ((N1D + (DIM == 3 ? N1D * N1D : N1D)) | 1);
  return face > vol ? face : vol;
}

template <int DIM, int N1D>
__global__ void __launch_bounds__(PDH_WAVE) k_rhs(const PdhDev P, const int n_owned, const double *__restrict__ f_vol,
                                                  const double *__restrict__ g_face, double *__restrict__ rhs,
                                                  const int64_t *__restrict__ vq_src, const int64_t *__restrict__ ap_src,
                                                  const int64_t *__restrict__ bd_rng)
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
  constexpr int FCHK = rhs_face_chunk<DIM, N1D>();
  double *aux = lds + FCHK * RC::LEN; // [FCHK][1+DIM]
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
  const int fi = (int)blockIdx.y * PDH_WAVE + lane;
  const bool live = fi < P.n;
  int off[DIM];
  {
    const uint32_t packed = live ? (uint32_t)P.midx[fi] : 0u;
    for (int c = 0; c < DIM; ++c)
      off[c] = live ? (c * N1D + (int)((packed >> (8 * c)) & 0xff)) * 2 : RC::ZERO_OFF / 8;
  }
  double acc = 0.0;
  // offsets of this lane's function into a volume record
  int voff0 = 0, voff12 = N1D;
  {
    const uint32_t packed = live ? (uint32_t)P.midx[fi] : 0u;
    const int k0 = (int)(packed & 0xff), k1 = (int)((packed >> 8) & 0xff), k2 = (int)((packed >> 16) & 0xff);
    voff0 = k0;
    voff12 = N1D + k1 + (DIM == 3 ? N1D * k2 : 0);
  }

  // volume: sum_q phi_i f JxW
  if constexpr (DIM == 3 && N1D == 4)
    {
      // Degree 3 in 3-D: the sum over the points as a GEMM on the f64 MFMA.  rhs[k0; (k1,k2)] = sum_q (f JxW B0_k0)(q) (B1_k1
      // B2_k2)(q) is a 4 x 16 product with K = the points: v_mfma_f64_4x4x4_4b with block b = k2, A_b[i][k] = f JxW B0_i at
      // point 4 s + k (the same for all blocks), B_b[k][j] = B1_j B2_b.  Lane (k, blk, idx) of a step evaluates three cubics
      // at its point (read from LDS: centred box coordinates and f JxW, 4 doubles) - 11 VALU instructions and 2 LDS reads per
      // 4 points and lane instead of 2 LDS reads and a multiply-add per POINT and lane.  FE_AggloDGP(3) is the subset
      // k0 + k1 + k2 <= 3 of the same 64 sums.
      if (f_vol)
        {
          const int kq = lane >> 4, blk = (lane >> 2) & 3, idx = lane & 3;
          double cI[4], cB[4]; // monomial coefficients (centred variable) of the 1-D functions idx and blk
          static_for<0, 4>([&](auto m_) {
            constexpr int m = m_;
            cI[m] = idx == 0 ? P.tab.coef[0][m] : (idx == 1 ? P.tab.coef[1][m] : (idx == 2 ? P.tab.coef[2][m] : P.tab.coef[3][m]));
            cB[m] = blk == 0 ? P.tab.coef[0][m] : (blk == 1 ? P.tab.coef[1][m] : (blk == 2 ? P.tab.coef[2][m] : P.tab.coef[3][m]));
          });
          double vacc = 0.0, vacc2 = 0.0; // (two accumulators: consecutive MFMAs do not wait for each other)
          const int64_t qb = P.vq_ptr[slot], qe = P.vq_ptr[slot + 1];
          double *pts = lds; // [64][4]: xhat0, xhat1, xhat2 (centred), f JxW
          // point data of the NEXT chunk is requested before the current one is worked on (one wave per polytope: nothing
          // else hides the latency of these loads)
          struct Raw
          {
            double x0, x1, x2, f, w;
          };
          auto issue = [&](int64_t base) {
            Raw r;
            const bool on = base + lane < qe;
            const int64_t q = on ? base + lane : qb;
            const int64_t src = vq_src ? vq_src[slot] + (q - qb) : q;
            r.x0 = P.vq_x[0 * P.vq_stride + q];
            r.x1 = P.vq_x[1 * P.vq_stride + q];
            r.x2 = P.vq_x[2 * P.vq_stride + q];
            r.f = f_vol[src];
            r.w = on ? P.vq_w[q] : 0.0;
            return r;
          };
          Raw nxt = issue(qb);
          for (int64_t base = qb; base < qe; base += PDH_WAVE)
            {
              const int cnt = (int)((qe - base < PDH_WAVE) ? (qe - base) : PDH_WAVE);
              PDH_WAVE_SYNC();
              pts[lane * 4 + 0] = (nxt.x0 - lo[0]) / h[0] - 0.5;
              pts[lane * 4 + 1] = (nxt.x1 - lo[1]) / h[1] - 0.5;
              pts[lane * 4 + 2] = (nxt.x2 - lo[2]) / h[2] - 0.5;
              pts[lane * 4 + 3] = nxt.f * nxt.w; // dead points: w = 0
              if (base + PDH_WAVE < qe)
                nxt = issue(base + PDH_WAVE);
              PDH_WAVE_SYNC();
              (void)cnt; // all 16 steps of 4 points: dead points carry f JxW = 0
              static_for<0, PDH_WAVE / 4>([&](auto st_) {
                constexpr int st = st_;
                const double *pq = pts + (4 * st + kq) * 4;
                const double x0 = pq[0], x1 = pq[1], x2 = pq[2], fw = pq[3];
                double b0 = cI[3], b1 = cI[3], b2 = cB[3];
                static_for<0, 3>([&](auto t_) {
                  constexpr int m = 2 - t_;
                  b0 = b0 * x0 + cI[m];
                  b1 = b1 * x1 + cI[m];
                  b2 = b2 * x2 + cB[m];
                });
                if constexpr (st % 2 == 0)
                  vacc = mfma4(fw * b0, b1 * b2, vacc);
                else
                  vacc2 = mfma4(fw * b0, b1 * b2, vacc2);
              });
            }
          // D layout: lane (i, blk, j) holds rhs[k0 = i, k1 = j, k2 = blk]; hand over to the lanes = functions of this kernel
          PDH_WAVE_SYNC();
          lds[(lane >> 4) + 4 * (lane & 3) + 16 * ((lane >> 2) & 3)] = vacc + vacc2;
          PDH_WAVE_SYNC();
          if (live)
            {
              const uint32_t packed = (uint32_t)P.midx[fi];
              acc += lds[(int)(packed & 0xff) + 4 * (int)((packed >> 8) & 0xff) + 16 * (int)((packed >> 16) & 0xff)];
            }
          PDH_WAVE_SYNC();
        }
    }
  else if (f_vol)
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
      // (bd_rng: the packed points of this slot that lie on the boundary - none for most polytopes; without it every chunk's
      // map entries would be loaded one after the other just to find that out)
      const int64_t pb = bd_rng ? bd_rng[2 * (int64_t)slot] : P.ap_ptr[slot], pe = bd_rng ? bd_rng[2 * (int64_t)slot + 1] : P.ap_ptr[slot + 1];
      for (int64_t base = pb; base < pe; base += FCHK)
        {
          const int cnt = (int)((pe - base < FCHK) ? (pe - base) : FCHK);
          // a chunk without boundary points contributes nothing (ap_src = -1 on interior faces)
          const int64_t src_l = lane < cnt ? (ap_src ? ap_src[base + lane] : base + lane) : -1;
          if (!bd_rng && __ballot(src_l >= 0) == 0ull) // (with bd_rng the range holds boundary points only: no need to wait)
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
    rhs[(int64_t)P.own_row[slot] + fi] = acc;
}
} // namespace pdh

extern "C" hipError_t pdh_launch_rhs(int dim, int n1d, const PdhDev *P, int count, const double *f_vol,
                                     const double *g_face, double *rhs, const int64_t *vq_src, const int64_t *ap_src,
                                     const int64_t *bd_rng, hipStream_t stream)
{
  if (count <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)count, (unsigned)((P->n + PDH_WAVE - 1) / PDH_WAVE)), block(PDH_WAVE);
#define PDH_RHS_CASE(D, N)                                                                                 \
  if (dim == D && n1d == N)                                                                                \
    {                                                                                                      \
      const size_t lds = (size_t)pdh::rhs_lds_doubles<D, N>() * sizeof(double);                            \
      hipLaunchKernelGGL((pdh::k_rhs<D, N>), grid, block, lds, stream, *P, count, f_vol, g_face, rhs, vq_src, ap_src,  \
                         bd_rng);                                                                          \
      return hipGetLastError();                                                                            \
    }
  PDH_RHS_CASE(2, 1) PDH_RHS_CASE(2, 2) PDH_RHS_CASE(2, 3) PDH_RHS_CASE(2, 4)
  PDH_RHS_CASE(2, 5) PDH_RHS_CASE(2, 6) PDH_RHS_CASE(2, 7) PDH_RHS_CASE(2, 8)
  PDH_RHS_CASE(3, 1) PDH_RHS_CASE(3, 2) PDH_RHS_CASE(3, 3) PDH_RHS_CASE(3, 4)
  PDH_RHS_CASE(3, 5) PDH_RHS_CASE(3, 6) PDH_RHS_CASE(3, 7) PDH_RHS_CASE(3, 8)
#undef PDH_RHS_CASE
  return hipErrorInvalidValue;
}
