// pdh_kernels.h — hand-written gfx950 (CDNA4) kernels of the SIP assembly path.
//
// What they compute (reference include/poly_utils.h:2034-2193, 1870-1926):
//   k_diag    : one wavefront per polytope P.  Writes the dense block A[P,P] =
//                 sum_q grad phi_i . grad phi_j JxW (+ c phi_i phi_j JxW)            (volume, A7)
//               + for every face of P the own-side SIP block
//                 sum_q ( -1/2 g_i phi_j - 1/2 phi_i g_j + sigma phi_i phi_j ) JxW   (M11 / M22, A9)
//                 with g = grad phi . n_own; Nitsche boundary faces are the same form with weight 2 JxW
//                 and sigma/2 (packed that way by the host, exact in fp).
//   k_offdiag : one wavefront per (polytope P, interior face F).  Writes the coupling block A[P,Q] =
//                 sum_q ( 1/2 g^P_i phi^Q_j - 1/2 phi^P_i g^Q_j - sigma phi^P_i phi^Q_j ) JxW_1
//               (M12 seen from side 0, M21 seen from side 1; both are this form with P's outward normal).
//   Every CSR value is written exactly once by exactly one wave: no atomics, no zero-fill pass,
//   deterministic, and rows are owned by the wave of their polytope (multi-GPU needs no exchange).
//
// How (MI355X specifics):
//   * basis functions are never tabulated in HBM.  Per chunk of 64 quadrature points each lane evaluates
//     the 1-D basis polynomials and derivatives of ONE point in the bounding-box frame
//     (x_hat = (x-lo)/h, d/dx = 1/h d/dx_hat: reference source/mapping_box.cc:210-222, 522-531) and
//     leaves them in LDS; MFMA operand fragments are then formed in registers from 3 LDS reads.
//   * the contraction over quadrature points runs on v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks
//     per instruction).  Measured on MI355X (tools/probes): 73 TFLOP/s vs 48 for v_mfma_f64_16x16x4 and
//     59 for v_fma_f64.  Operand lane map (probed): A/B lane = 16*k + 4*blk + idx, D lane = 16*i + 4*blk + j.
//     A fragment register holds 16 consecutive basis functions x 4 quadrature points; the 4x4 blocks off
//     the block diagonal are reached by rotating the B operand inside its 16-lane row (DPP row_ror).
//   * output blocks are transposed/mirrored through a 16-row LDS strip and written as contiguous row
//     segments in final CSR order (deal.II diagonal-first layout handled in the epilogue).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define PDH_MAX_N1D 8
#define PDH_WAVE 64

struct PdhBasisTab
{
  double coef[PDH_MAX_N1D][PDH_MAX_N1D]; // coef[k][m]: monomial coefficients of 1-D basis function k
};

struct PdhDev
{
  int32_t dim, n, n1d, diag_first;
  double reaction_c;
  const double *bbox;   // [n_agg][2][dim]
  const int32_t *midx;  // [16*NT] packed multi-index (k0 | k1<<8 | k2<<16), 0xffffffff = dead
  // volume quadrature of the owned polytopes (SoA), indexed by owned slot
  const int64_t *vq_ptr;
  const double *vq_x;
  int64_t vq_stride;
  const double *vq_w;
  // own-side face points, packed per owned polytope
  const int64_t *ap_ptr; // [n_owned+1]
  const double *ap_x;    // [dim][P]
  const double *ap_n;    // [dim][P] outward normal of the owning polytope
  int64_t ap_stride;
  const double *ap_wself;  // [P] JxW used by the diagonal block (2 JxW on the boundary)
  const double *ap_wcross; // [P] JxW used by the coupling block (JxW of side 1)
  const double *ap_sig;    // [P] sigma (sigma/2 on the boundary)
  // diagonal-block items
  const int32_t *own_agg;  // [n_owned]
  const int64_t *row_base; // [n_owned] value offset of the polytope's first row
  const int32_t *row_len;  // [n_owned] entries per row
  const int32_t *diag_L;   // [n_owned] ascending column position of the own block inside the row
  // coupling-block items
  const int32_t *it_own;  // owned slot
  const int32_t *it_nbr;  // neighbour polytope id
  const int64_t *it_pbeg; // first packed point
  const int32_t *it_pcnt; // number of points
  const int32_t *it_pos;  // position of the neighbour block inside the row (diag-first shift included)
  double *values;
  PdhBasisTab tab;
};

namespace pdh
{
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
  if constexpr (I < N)
    {
      f(std::integral_constant<int, I>{});
      static_for<I + 1, N>(f);
    }
}

// Rotate the four 4-lane blocks of every 16-lane row: result block bb = input block (bb+N)&3.
template <int N>
__device__ __forceinline__ double rot_blocks(double x)
{
  if constexpr (N == 0)
    return x;
  else
    {
      constexpr int ROR = 16 - 4 * N; // row_ror:R gives lane m the value of lane (m-R) mod 16
      int lo = __double2loint(x), hi = __double2hiint(x);
      lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + ROR, 0xf, 0xf, false);
      hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + ROR, 0xf, 0xf, false);
      return __hiloint2double(hi, lo);
    }
}

__device__ __forceinline__ double mfma4(double a, double b, double c)
{
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// Compile-time product schedule.  NT fragments of 16 basis functions; the last fragment has LB live
// 4-function blocks.  If LB is 1 or 2 the last fragment is stored replicated ([F,F,F,F] / [F0,F1,F0,F1])
// so that one instruction pairs it with four different blocks of the other operand.
template <int NT, int LB>
struct Sched
{
  static constexpr int T = 4 * (NT - 1) + LB; // live 4x4 tile rows
  static constexpr int rep(int f) { return (f == NT - 1) ? (LB == 1 ? 1 : (LB == 2 ? 2 : 4)) : 4; }
  static constexpr int tile(int f, int blk) { return 4 * f + (blk % rep(f)); }
  static constexpr bool live(int a, int b, int r, int bb) { return tile(a, bb) < T && tile(b, (bb + r) & 3) < T; }
  // first producer of the (un)ordered tile pair inside the (a,b) group, in (r,bb) order
  static constexpr bool canon(int a, int b, int r, int bb, bool sym)
  {
    if (!live(a, b, r, bb))
      return false;
    const int ti = tile(a, bb), tj = tile(b, (bb + r) & 3);
    for (int r2 = 0; r2 <= r; ++r2)
      for (int b2 = 0; b2 < 4; ++b2)
        {
          if (r2 == r && b2 >= bb)
            break;
          if (!live(a, b, r2, b2))
            continue;
          const int ui = tile(a, b2), uj = tile(b, (b2 + r2) & 3);
          if ((ui == ti && uj == tj) || (sym && ui == tj && uj == ti))
            return false;
        }
    return true;
  }
  static constexpr unsigned mask(int a, int b, int r, bool sym)
  {
    unsigned m = 0;
    for (int bb = 0; bb < 4; ++bb)
      if (canon(a, b, r, bb, sym))
        m |= 1u << bb;
    return m;
  }
  static constexpr unsigned sym_mask(int a, int b, int r) { return (a <= b) ? mask(a, b, r, a == b) : 0u; }
  static constexpr unsigned full_mask(int a, int b, int r) { return mask(a, b, r, false); }
};

// Per-lane addressing of the 1-D tables: for fragment a the lane owns basis function
// i = 16a + (m % (4*rep)) (replication, see Sched), m = lane & 15.
template <int DIM, int NT, int LB>
struct LaneBasis
{
  int off[NT][DIM]; // byte offset inside a point record of the (val,der) pair of direction c
  double livef[NT]; // 1.0 / 0.0
  __device__ __forceinline__ void init(const PdhDev &P, int lane)
  {
    using S = Sched<NT, LB>;
    const int m = lane & 15;
    static_for<0, NT>([&](auto a_) {
      constexpr int a = a_;
      constexpr int rp = S::rep(a);
      const int mm = (rp == 4) ? m : (m % (4 * rp));
      const int i = 16 * a + mm;
      const uint32_t packed = (i < P.n) ? (uint32_t)P.midx[i] : 0xffffffffu;
      const bool ok = packed != 0xffffffffu;
      livef[a] = ok ? 1.0 : 0.0;
      for (int c = 0; c < DIM; ++c)
        {
          const int k = ok ? (int)((packed >> (8 * c)) & 0xff) : 0;
          off[a][c] = (c * P.n1d + k) * 16;
        }
    });
  }
};

// Evaluate the 1-D basis of one point (this lane's) in the frame of a bounding box and store the
// record [c][k] -> (value, derivative / h_c) in LDS.
template <int DIM>
__device__ __forceinline__ void eval_point_record(const PdhDev &P, const double *lo, const double *h,
                                                  const double *x, double *rec)
{
  const int p = P.n1d - 1;
  for (int c = 0; c < DIM; ++c)
    {
      const double xh = (x[c] - lo[c]) / h[c]; // BoundingBox::real_to_unit (agglomeration_handler.cc:703-704)
      const double ih = 1.0 / h[c];            // inverse_cell_extents (mapping_box.cc:222)
      for (int k = 0; k < P.n1d; ++k)
        {
          double val = P.tab.coef[k][p], der = 0.0;
          for (int mth = p - 1; mth >= 0; --mth)
            {
              der = der * xh + val;
              val = val * xh + P.tab.coef[k][mth];
            }
          rec[(c * P.n1d + k) * 2 + 0] = val;
          rec[(c * P.n1d + k) * 2 + 1] = der * ih;
        }
    }
}

typedef double d2_t __attribute__((ext_vector_type(2)));

// phi and the DIM partial derivatives of fragment a's function at the lane's point.
template <int DIM>
__device__ __forceinline__ void frag_eval(const char *rec_bytes, const int *off, double livef, double &phi,
                                          double *dphi)
{
  d2_t t[DIM];
  for (int c = 0; c < DIM; ++c)
    t[c] = *reinterpret_cast<const d2_t *>(rec_bytes + off[c]);
  if constexpr (DIM == 2)
    {
      const double v0 = t[0].x * livef, v1 = t[1].x;
      phi = v0 * v1;
      dphi[0] = (t[0].y * livef) * v1;
      dphi[1] = v0 * t[1].y;
    }
  else
    {
      const double v0 = t[0].x * livef, v1 = t[1].x, v2 = t[2].x;
      const double v12 = v1 * v2, v01 = v0 * v1;
      phi = v0 * v12;
      dphi[0] = (t[0].y * livef) * v12;
      dphi[1] = (v0 * v2) * t[1].y;
      dphi[2] = v01 * t[2].y;
    }
}

template <int NT>
__device__ __forceinline__ constexpr int acc_idx(int a, int b, int r)
{
  return (a * NT + b) * 4 + r;
}

// acc[a,b,r] += A[a] (x) rot_r(B[b]) for the symmetric (a<=b) or the full schedule.
template <int NT, int LB, bool SYM>
__device__ __forceinline__ void product(double *acc, const double *A, const double *B)
{
  using S = Sched<NT, LB>;
  static_for<0, NT>([&](auto b_) {
    constexpr int b = b_;
    const double R0 = B[b];
    const double R1 = rot_blocks<1>(R0), R2 = rot_blocks<2>(R0), R3 = rot_blocks<3>(R0);
    static_for<0, NT>([&](auto a_) {
      constexpr int a = a_;
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        constexpr unsigned msk = SYM ? S::sym_mask(a, b, r) : S::full_mask(a, b, r);
        if constexpr (msk != 0u)
          {
            const double Rr = (r == 0) ? R0 : (r == 1) ? R1 : (r == 2) ? R2 : R3;
            acc[acc_idx<NT>(a, b, r)] = mfma4(A[a], Rr, acc[acc_idx<NT>(a, b, r)]);
          }
      });
    });
  });
}

// Scatter the accumulators' entries that fall into row strip `strip_a` (rows 16a..16a+15 of the block)
// into the LDS strip [16][ncol_pad]; SYM additionally mirrors (i,j) -> (j,i).
template <int NT, int LB, bool SYM, int STRIP>
__device__ __forceinline__ void fill_strip(const double *acc, double *strip, int ncol_pad, int lane, int n)
{
  using S = Sched<NT, LB>;
  const int i = lane >> 4, bb = (lane >> 2) & 3, j = lane & 3;
  static_for<0, NT>([&](auto a_) {
    constexpr int a = a_;
    static_for<0, NT>([&](auto b_) {
      constexpr int b = b_;
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        constexpr unsigned msk = SYM ? S::sym_mask(a, b, r) : S::full_mask(a, b, r);
        if constexpr (msk != 0u)
          {
            if ((msk >> bb) & 1u)
              {
                const int ti = S::tile(a, bb), tj = S::tile(b, (bb + r) & 3);
                const int R = 4 * ti + i, C = 4 * tj + j;
                const double v = acc[acc_idx<NT>(a, b, r)];
                if (R < n && C < n)
                  {
                    if ((R >> 4) == STRIP)
                      strip[(R & 15) * ncol_pad + C] = v;
                    if (SYM && ti != tj && (C >> 4) == STRIP)
                      strip[(C & 15) * ncol_pad + R] = v;
                  }
              }
          }
      });
    });
  });
}

// ------------------------------------------------------------------------------------------------
// Diagonal-block kernel: one wave per owned polytope.
// ------------------------------------------------------------------------------------------------
template <int DIM, int NT, int LB>
__global__ void __launch_bounds__(PDH_WAVE) k_diag(const PdhDev P, const int n_owned)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;
  if (slot >= n_owned)
    return;
  const int agg = P.own_agg[slot];
  const int rec_len = DIM * P.n1d * 2; // doubles per point record
  double *rec = lds;                   // [64][rec_len]
  double *aux = lds + PDH_WAVE * rec_len; // [64][2+DIM]: w, sigma, normal
  constexpr int AUXN = 2 + DIM;

  double lo[DIM], h[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      h[c] = P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - lo[c];
    }

  LaneBasis<DIM, NT, LB> lb;
  lb.init(P, lane);

  double acc[NT * NT * 4];
  for (int t = 0; t < NT * NT * 4; ++t)
    acc[t] = 0.0;

  const int kq = lane >> 4;

  // ---- volume term ---------------------------------------------------------------------------
  {
    const int64_t qb = P.vq_ptr[slot], qe = P.vq_ptr[slot + 1];
    for (int64_t base = qb; base < qe; base += PDH_WAVE)
      {
        const int cnt = (int)((qe - base < PDH_WAVE) ? (qe - base) : PDH_WAVE);
        __syncthreads();
        {
          double x[DIM], w = 0.0;
          if (lane < cnt)
            {
              for (int c = 0; c < DIM; ++c)
                x[c] = P.vq_x[c * P.vq_stride + base + lane];
              w = P.vq_w[base + lane];
            }
          else
            for (int c = 0; c < DIM; ++c)
              x[c] = lo[c];
          eval_point_record<DIM>(P, lo, h, x, rec + lane * rec_len);
          aux[lane * AUXN] = w;
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        for (int step = 0; step < nsteps; ++step)
          {
            const int pt = 4 * step + kq;
            const char *rb = reinterpret_cast<const char *>(rec + pt * rec_len);
            const double w = aux[pt * AUXN];
            double phi[NT], dphi[NT][DIM];
            static_for<0, NT>([&](auto a_) {
              constexpr int a = a_;
              frag_eval<DIM>(rb, lb.off[a], lb.livef[a], phi[a], dphi[a]);
            });
            static_for<0, DIM>([&](auto c_) {
              constexpr int c = c_;
              double A[NT], B[NT];
              static_for<0, NT>([&](auto a_) {
                constexpr int a = a_;
                B[a] = dphi[a][c];
                A[a] = w * dphi[a][c];
              });
              product<NT, LB, true>(acc, A, B);
            });
            if (P.reaction_c != 0.0)
              {
                double A[NT];
                const double cw = P.reaction_c * w;
                static_for<0, NT>([&](auto a_) {
                  constexpr int a = a_;
                  A[a] = cw * phi[a];
                });
                product<NT, LB, true>(acc, A, phi);
              }
          }
      }
  }

  // ---- own-side face terms (all faces of the polytope, boundary included) ----------------------
  {
    const int64_t pb = P.ap_ptr[slot], pe = P.ap_ptr[slot + 1];
    for (int64_t base = pb; base < pe; base += PDH_WAVE)
      {
        const int cnt = (int)((pe - base < PDH_WAVE) ? (pe - base) : PDH_WAVE);
        __syncthreads();
        {
          double x[DIM], nr[DIM], w = 0.0, sg = 0.0;
          if (lane < cnt)
            {
              for (int c = 0; c < DIM; ++c)
                {
                  x[c] = P.ap_x[c * P.ap_stride + base + lane];
                  nr[c] = P.ap_n[c * P.ap_stride + base + lane];
                }
              w = P.ap_wself[base + lane];
              sg = P.ap_sig[base + lane];
            }
          else
            for (int c = 0; c < DIM; ++c)
              {
                x[c] = lo[c];
                nr[c] = 0.0;
              }
          eval_point_record<DIM>(P, lo, h, x, rec + lane * rec_len);
          aux[lane * AUXN] = w;
          aux[lane * AUXN + 1] = sg;
          for (int c = 0; c < DIM; ++c)
            aux[lane * AUXN + 2 + c] = nr[c];
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        for (int step = 0; step < nsteps; ++step)
          {
            const int pt = 4 * step + kq;
            const char *rb = reinterpret_cast<const char *>(rec + pt * rec_len);
            const double w = aux[pt * AUXN], hs = 0.5 * aux[pt * AUXN + 1];
            double nr[DIM];
            for (int c = 0; c < DIM; ++c)
              nr[c] = aux[pt * AUXN + 2 + c];
            double Phi[NT], U[NT], AU[NT], APhi[NT];
            static_for<0, NT>([&](auto a_) {
              constexpr int a = a_;
              double ph, dp[DIM];
              frag_eval<DIM>(rb, lb.off[a], lb.livef[a], ph, dp);
              double g = nr[0] * dp[0];
              for (int c = 1; c < DIM; ++c)
                g += nr[c] * dp[c];
              Phi[a] = ph;
              U[a] = hs * ph - 0.5 * g; // -1/2 grad phi . n + sigma/2 phi
              AU[a] = w * U[a];
              APhi[a] = w * ph;
            });
            product<NT, LB, true>(acc, AU, Phi);
            product<NT, LB, true>(acc, APhi, U);
          }
      }
  }

  // ---- epilogue: mirror + write rows in CSR order ------------------------------------------------
  const int n = P.n;
  const int ncol_pad = 16 * NT + 2;
  double *strip = lds; // overlays the point records
  const int64_t rbase = P.row_base[slot];
  const int rlen = P.row_len[slot];
  const int L = P.diag_L[slot];
  static_for<0, NT>([&](auto s_) {
    constexpr int s = s_;
    __syncthreads();
    fill_strip<NT, LB, true, s>(acc, strip, ncol_pad, lane, n);
    __syncthreads();
    const int rows = (n - 16 * s < 16) ? (n - 16 * s) : 16;
    for (int idx = lane; idx < rows * n; idx += PDH_WAVE)
      {
        const int rr = idx / n, c = idx - rr * n;
        const int R = 16 * s + rr;
        int pos;
        if (P.diag_first)
          pos = (c == R) ? 0 : (L + c + (c < R ? 1 : 0));
        else
          pos = L + c;
        P.values[rbase + (int64_t)R * rlen + pos] = strip[rr * ncol_pad + c];
      }
  });
}

// ------------------------------------------------------------------------------------------------
// Coupling-block kernel: one wave per (owned polytope, interior face).
// ------------------------------------------------------------------------------------------------
template <int DIM, int NT, int LB>
__global__ void __launch_bounds__(PDH_WAVE) k_offdiag(const PdhDev P, const int n_items)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  if (item >= n_items)
    return;
  const int slot = P.it_own[item];
  const int agg = P.own_agg[slot];
  const int nbr = P.it_nbr[item];
  const int rec_len = DIM * P.n1d * 2;
  double *recP = lds;                        // own frame
  double *recQ = lds + PDH_WAVE * rec_len;   // neighbour frame
  double *aux = lds + 2 * PDH_WAVE * rec_len; // [64][2+DIM]
  constexpr int AUXN = 2 + DIM;

  double loP[DIM], hP[DIM], loQ[DIM], hQ[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      loP[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      hP[c] = P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - loP[c];
      loQ[c] = P.bbox[(int64_t)nbr * 2 * DIM + c];
      hQ[c] = P.bbox[(int64_t)nbr * 2 * DIM + DIM + c] - loQ[c];
    }

  LaneBasis<DIM, NT, LB> lb;
  lb.init(P, lane);

  double acc[NT * NT * 4];
  for (int t = 0; t < NT * NT * 4; ++t)
    acc[t] = 0.0;

  const int kq = lane >> 4;
  const int64_t pb = P.it_pbeg[item], pe = pb + P.it_pcnt[item];
  for (int64_t base = pb; base < pe; base += PDH_WAVE)
    {
      const int cnt = (int)((pe - base < PDH_WAVE) ? (pe - base) : PDH_WAVE);
      __syncthreads();
      {
        double x[DIM], nr[DIM], w = 0.0, sg = 0.0;
        if (lane < cnt)
          {
            for (int c = 0; c < DIM; ++c)
              {
                x[c] = P.ap_x[c * P.ap_stride + base + lane];
                nr[c] = P.ap_n[c * P.ap_stride + base + lane];
              }
            w = P.ap_wcross[base + lane];
            sg = P.ap_sig[base + lane];
            eval_point_record<DIM>(P, loP, hP, x, recP + lane * rec_len);
            eval_point_record<DIM>(P, loQ, hQ, x, recQ + lane * rec_len);
          }
        else
          {
            for (int c = 0; c < DIM; ++c)
              nr[c] = 0.0;
            eval_point_record<DIM>(P, loP, hP, loP, recP + lane * rec_len);
            eval_point_record<DIM>(P, loQ, hQ, loQ, recQ + lane * rec_len);
          }
        aux[lane * AUXN] = w;
        aux[lane * AUXN + 1] = sg;
        for (int c = 0; c < DIM; ++c)
          aux[lane * AUXN + 2 + c] = nr[c];
      }
      __syncthreads();
      const int nsteps = (cnt + 3) >> 2;
      for (int step = 0; step < nsteps; ++step)
        {
          const int pt = 4 * step + kq;
          const char *rbP = reinterpret_cast<const char *>(recP + pt * rec_len);
          const char *rbQ = reinterpret_cast<const char *>(recQ + pt * rec_len);
          const double w = aux[pt * AUXN], sg = aux[pt * AUXN + 1];
          double nr[DIM];
          for (int c = 0; c < DIM; ++c)
            nr[c] = aux[pt * AUXN + 2 + c];
          double A1[NT], A2[NT], B1[NT], B2[NT];
          static_for<0, NT>([&](auto a_) {
            constexpr int a = a_;
            double ph, dp[DIM];
            frag_eval<DIM>(rbP, lb.off[a], lb.livef[a], ph, dp);
            double g = nr[0] * dp[0];
            for (int c = 1; c < DIM; ++c)
              g += nr[c] * dp[c];
            A1[a] = w * (0.5 * g - sg * ph); // (1/2 g^P - sigma phi^P) JxW
            A2[a] = -0.5 * w * ph;           // -1/2 phi^P JxW
            frag_eval<DIM>(rbQ, lb.off[a], lb.livef[a], ph, dp);
            g = nr[0] * dp[0];
            for (int c = 1; c < DIM; ++c)
              g += nr[c] * dp[c];
            B1[a] = ph; // phi^Q
            B2[a] = g;  // grad phi^Q . n_P
          });
          product<NT, LB, false>(acc, A1, B1);
          product<NT, LB, false>(acc, A2, B2);
        }
    }

  const int n = P.n;
  const int ncol_pad = 16 * NT + 2;
  double *strip = lds;
  const int64_t rbase = P.row_base[slot];
  const int rlen = P.row_len[slot];
  const int pos0 = P.it_pos[item];
  static_for<0, NT>([&](auto s_) {
    constexpr int s = s_;
    __syncthreads();
    fill_strip<NT, LB, false, s>(acc, strip, ncol_pad, lane, n);
    __syncthreads();
    const int rows = (n - 16 * s < 16) ? (n - 16 * s) : 16;
    for (int idx = lane; idx < rows * n; idx += PDH_WAVE)
      {
        const int rr = idx / n, c = idx - rr * n;
        const int R = 16 * s + rr;
        P.values[rbase + (int64_t)R * rlen + pos0 + c] = strip[rr * ncol_pad + c];
      }
  });
}

// LDS bytes needed by the two kernels (host side helper).
inline size_t lds_bytes_diag(int dim, int n1d, int nt)
{
  const size_t recs = (size_t)PDH_WAVE * (dim * n1d * 2 + 2 + dim) * sizeof(double);
  const size_t strip = (size_t)16 * (16 * nt + 2) * sizeof(double);
  return recs > strip ? recs : strip;
}
inline size_t lds_bytes_offdiag(int dim, int n1d, int nt)
{
  const size_t recs = (size_t)PDH_WAVE * (2 * dim * n1d * 2 + 2 + dim) * sizeof(double);
  const size_t strip = (size_t)16 * (16 * nt + 2) * sizeof(double);
  return recs > strip ? recs : strip;
}
} // namespace pdh
