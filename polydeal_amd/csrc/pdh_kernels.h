// pdh_kernels.h — hand-written gfx950 (CDNA4) kernels of the SIP assembly path.
//
// What they compute (reference include/poly_utils.h:2034-2193, 1870-1926):
//   k_diag    : one wavefront per polytope P.  Writes the dense block A[P,P] =
//                 sum_q grad phi_i . grad phi_j JxW (+ c phi_i phi_j JxW)            (volume, A7)
//               + for every face of P the own-side SIP block
//                 sum_q ( -1/2 g_i phi_j - 1/2 phi_i g_j + sigma phi_i phi_j ) JxW   (M11 / M22, A9)
//                 with g = grad phi . n_own; Nitsche boundary faces are the same form with weight 2 JxW
//                 and sigma/2 (packed that way by the host, exact in fp).
//   k_offdiag : one wavefront per interior face (P,Q).  Computes the coupling block A[P,Q] =
//                 sum_q ( 1/2 g^P_i phi^Q_j - 1/2 phi^P_i g^Q_j - sigma phi^P_i phi^Q_j ) JxW_1
//               (M12 seen from side 0, M21 seen from side 1; both are this form with P's outward normal)
//               and writes it into P's rows and its transpose into Q's rows (M21 = M12^T: the reference
//               uses JxW_1 and sigma for both, poly_utils.h:1899-1914).
//   Every CSR value is written exactly once by exactly one wave: no atomics, no zero-fill pass,
//   deterministic, and rows are owned by the context of their polytope (multi-GPU needs no exchange).
//
// How (MI355X specifics):
//   * basis functions are never tabulated in HBM.  Per chunk of 64 quadrature points each lane evaluates
//     the 1-D basis polynomials and derivatives of ONE point in the bounding-box frame
//     (x_hat = (x-lo)/h, d/dx = 1/h d/dx_hat: reference source/mapping_box.cc:210-222, 522-531) and
//     leaves them in LDS, scaled by sqrt(JxW) so that operands need no further weighting and the
//     volume block comes out exactly symmetric; MFMA operand fragments are formed in registers from
//     DIM ds_read_b128 + 2 DIM multiplies.
//   * the contraction over quadrature points runs on v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks
//     per instruction).  Measured on MI355X (tools/probes): 73 TFLOP/s vs 48 for v_mfma_f64_16x16x4 and
//     59 for v_fma_f64.  Operand lane map (probed): A/B lane = 16*k + 4*blk + idx, D lane = 16*i + 4*blk + j.
//     A fragment register holds 16 consecutive basis functions x 4 quadrature points; the 4x4 blocks off
//     the block diagonal are reached by rotating the B operand inside its 16-lane row.  The rotation goes
//     through ds_bpermute_b32 (LDS crossbar, no LDS memory): with DPP moves the kernel was bound by
//     VALU issue (2.4 VALU per MFMA, profiles/r01_v2_pmc_sq.txt); CBSZ/ABID broadcast does not act on the
//     f64 MFMA (tools/probes/mfma_cbsz_probe.hip).
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
  // coupling-block items: one per interior face with at least one owned side
  const int32_t *it_own;  // owned slot of P (the side whose packed points are used)
  const int32_t *it_nbr;  // neighbour polytope id Q
  const int64_t *it_pbeg; // first packed point
  const int32_t *it_pcnt; // number of points
  const int32_t *it_pos;  // position of Q's block inside P's rows (diag-first shift included)
  const int32_t *it_nbr_slot; // owned slot of Q, or -1: A[Q,P] = A[P,Q]^T is then not written here
  const int32_t *it_pos_t;    // position of P's block inside Q's rows
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

__device__ __forceinline__ double mfma4(double a, double b, double c)
{
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// Rotation of the four 4-lane blocks of every 16-lane row: result block bb = input block (bb+R)&3.
// Both cross-lane paths are used so that neither pipe limits the MFMA rate: R = 1 goes through DPP
// (VALU, row_ror), R = 2, 3 through ds_bpermute_b32 (LDS crossbar, ~8 LDS-pipe cycles each, measured).
struct Rotator
{
  int addr[2]; // 4 * source lane for R = 2, 3
  __device__ __forceinline__ void init(int lane)
  {
    for (int r = 2; r < 4; ++r)
      addr[r - 2] = ((lane & ~15) | ((lane + 4 * r) & 15)) * 4;
  }
  template <int R>
  __device__ __forceinline__ double rot(double x) const
  {
    if constexpr (R == 0)
      return x;
    else if constexpr (R == 1)
      {
        // row_ror:12 gives lane m the value of lane (m+4) mod 16 of its row (tools/probes/dpp_probe.hip)
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x120 + 12, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x120 + 12, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
      }
    else
      {
        const int lo = __builtin_amdgcn_ds_bpermute(addr[R - 2], __double2loint(x));
        const int hi = __builtin_amdgcn_ds_bpermute(addr[R - 2], __double2hiint(x));
        return __hiloint2double(hi, lo);
      }
  }
};

// Compile-time product schedule.  NT fragments of 16 basis functions; the last fragment has LB live
// 4-function blocks.  If LB is 1 or 2 the last fragment is stored replicated ([F,F,F,F] / [F0,F1,F0,F1])
// so that one instruction pairs it with four different blocks of the other operand.
template <int NT, int LB>
struct Sched
{
  static constexpr int T = 4 * (NT - 1) + LB; // live 4x4 tile rows
  static constexpr int rep(int f) { return (f == NT - 1) ? (LB == 1 ? 1 : (LB == 2 ? 2 : 4)) : 4; }
  static constexpr int tile(int f, int blk) { return 4 * f + (blk % rep(f)); }
  // Product r of fragment pair (a,b) multiplies rot_sa(A[a]) with rot_sb(B[b]); lane block bb then holds
  // tile (tile(a,bb+sa), tile(b,bb+sb)).  Relative rotation sb-sa = r.  In the symmetric case A and B are
  // the same fragments, so r = 3 is realised as (sa,sb) = (1,0) and rotation 3 is never needed.
  static constexpr int sa(int r, bool sym) { return (sym && r == 3) ? 1 : 0; }
  static constexpr int sb(int r, bool sym) { return (sym && r == 3) ? 0 : r; }
  static constexpr int ti(int a, int r, int bb, bool sym) { return tile(a, (bb + sa(r, sym)) & 3); }
  static constexpr int tj(int b, int r, int bb, bool sym) { return tile(b, (bb + sb(r, sym)) & 3); }
  static constexpr bool live(int a, int b, int r, int bb, bool sym) { return ti(a, r, bb, sym) < T && tj(b, r, bb, sym) < T; }
  // first producer of the (un)ordered tile pair inside the (a,b) group, in (r,bb) order
  static constexpr bool canon(int a, int b, int r, int bb, bool sym)
  {
    if (!live(a, b, r, bb, sym))
      return false;
    const bool unordered = sym && a == b;
    const int i0 = ti(a, r, bb, sym), j0 = tj(b, r, bb, sym);
    for (int r2 = 0; r2 <= r; ++r2)
      for (int b2 = 0; b2 < 4; ++b2)
        {
          if (r2 == r && b2 >= bb)
            break;
          if (!live(a, b, r2, b2, sym))
            continue;
          const int ui = ti(a, r2, b2, sym), uj = tj(b, r2, b2, sym);
          if ((ui == i0 && uj == j0) || (unordered && ui == j0 && uj == i0))
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
  static constexpr unsigned sym_mask(int a, int b, int r) { return (a <= b) ? mask(a, b, r, true) : 0u; }
  static constexpr unsigned full_mask(int a, int b, int r) { return mask(a, b, r, false); }
  // is rot_s of fragment f needed as a B operand (any a) / as an A operand (any b)?
  static constexpr bool needs_rot_b(int f, int s, bool sym)
  {
    for (int a = 0; a < NT; ++a)
      for (int r = 0; r < 4; ++r)
        if (sb(r, sym) == s && (sym ? sym_mask(a, f, r) : full_mask(a, f, r)) != 0u)
          return true;
    return false;
  }
  static constexpr bool needs_rot_a(int f, int s, bool sym)
  {
    for (int b = 0; b < NT; ++b)
      for (int r = 0; r < 4; ++r)
        if (sa(r, sym) == s && (sym ? sym_mask(f, b, r) : full_mask(f, b, r)) != 0u)
          return true;
    return false;
  }
};

// Point record in LDS: [c][k] -> (value, derivative / h_c), then one (0,0) pair that dead basis functions
// (padding of the last fragment) point to.  All entries of direction 0 carry the factor sqrt(weight).
template <int DIM, int N1D>
struct Rec
{
  static constexpr int LEN = DIM * N1D * 2 + 2; // doubles (even: ds_read_b128 stays 16-B aligned)
  static constexpr int ZERO_OFF = DIM * N1D * 16; // byte offset of the zero pair
};

// Per-lane addressing of the 1-D tables: for fragment a the lane owns basis function
// i = 16a + (m % (4*rep)) (replication, see Sched), m = lane & 15.
template <int DIM, int N1D, int NT, int LB>
struct LaneBasis
{
  int off[NT][DIM]; // byte offset inside a point record of the (val,der) pair of direction c
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
      for (int c = 0; c < DIM; ++c)
        off[a][c] = ok ? (c * N1D + (int)((packed >> (8 * c)) & 0xff)) * 16 : Rec<DIM, N1D>::ZERO_OFF;
    });
  }
};

// Evaluate the 1-D basis of one point (this lane's) in the frame of a bounding box and store its record.
// sw = sqrt(weight) is folded into direction 0.
template <int DIM, int N1D>
__device__ __forceinline__ void eval_point_record(const PdhBasisTab &tab, const double *lo, const double *h,
                                                  const double *x, double sw, double *rec)
{
  constexpr int p = N1D - 1;
  static_for<0, DIM>([&](auto c_) {
    constexpr int c = c_;
    const double xh = (x[c] - lo[c]) / h[c]; // BoundingBox::real_to_unit (agglomeration_handler.cc:703-704)
    double ih = 1.0 / h[c];                  // inverse_cell_extents (mapping_box.cc:222)
    double sv = 1.0;
    if constexpr (c == 0)
      {
        sv = sw;
        ih *= sw;
      }
    static_for<0, N1D>([&](auto k_) {
      constexpr int k = k_;
      double val = tab.coef[k][p], der = 0.0;
      static_for<0, p>([&](auto t_) {
        constexpr int mth = p - 1 - t_;
        der = der * xh + val;
        val = val * xh + tab.coef[k][mth];
      });
      rec[(c * N1D + k) * 2 + 0] = (c == 0) ? val * sv : val;
      rec[(c * N1D + k) * 2 + 1] = der * ih;
    });
  });
  rec[DIM * N1D * 2 + 0] = 0.0;
  rec[DIM * N1D * 2 + 1] = 0.0;
}

typedef double d2_t __attribute__((ext_vector_type(2)));

// sqrt(w) phi and sqrt(w) d_c phi of fragment a's function at the lane's point.
template <int DIM>
__device__ __forceinline__ void frag_eval(const char *rec_bytes, const int *off, double &phi, double *dphi)
{
  d2_t t[DIM];
  for (int c = 0; c < DIM; ++c)
    t[c] = *reinterpret_cast<const d2_t *>(rec_bytes + off[c]);
  if constexpr (DIM == 2)
    {
      phi = t[0].x * t[1].x;
      dphi[0] = t[0].y * t[1].x;
      dphi[1] = t[0].x * t[1].y;
    }
  else
    {
      const double v12 = t[1].x * t[2].x, v01 = t[0].x * t[1].x;
      phi = t[0].x * v12;
      dphi[0] = t[0].y * v12;
      dphi[1] = (t[0].x * t[2].x) * t[1].y;
      dphi[2] = v01 * t[2].y;
    }
}

template <int NT>
__device__ __forceinline__ constexpr int acc_idx(int a, int b, int r)
{
  return (a * NT + b) * 4 + r;
}

// acc[a,b,r] += rot_sa(A[a]) (x) rot_sb(B[b]) for the symmetric (a<=b) or the full schedule.
template <int NT, int LB, bool SYM>
__device__ __forceinline__ void product(double *acc, const double *A, const double *B, const Rotator &rt)
{
  using S = Sched<NT, LB>;
  double A1[NT]; // rot_1(A[a]), only used by the symmetric schedule (r = 3)
  static_for<0, NT>([&](auto a_) {
    constexpr int a = a_;
    if constexpr (S::needs_rot_a(a, 1, SYM))
      A1[a] = rt.template rot<1>(A[a]);
    else
      A1[a] = 0.0;
  });
  static_for<0, NT>([&](auto b_) {
    constexpr int b = b_;
    const double R0 = B[b];
    double R1 = 0.0, R2 = 0.0, R3 = 0.0;
    if constexpr (S::needs_rot_b(b, 1, SYM))
      R1 = rt.template rot<1>(R0);
    if constexpr (S::needs_rot_b(b, 2, SYM))
      R2 = rt.template rot<2>(R0);
    if constexpr (S::needs_rot_b(b, 3, SYM))
      R3 = rt.template rot<3>(R0);
    static_for<0, NT>([&](auto a_) {
      constexpr int a = a_;
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        constexpr unsigned msk = SYM ? S::sym_mask(a, b, r) : S::full_mask(a, b, r);
        if constexpr (msk != 0u)
          {
            constexpr int sb = S::sb(r, SYM);
            const double Rr = (sb == 0) ? R0 : (sb == 1) ? R1 : (sb == 2) ? R2 : R3;
            const double Aa = (S::sa(r, SYM) == 0) ? A[a] : A1[a];
            acc[acc_idx<NT>(a, b, r)] = mfma4(Aa, Rr, acc[acc_idx<NT>(a, b, r)]);
          }
      });
    });
  });
}

// Scatter the accumulators' entries that fall into row strip STRIP (rows 16*STRIP.. of the block) into the
// LDS strip [16][ncol_pad]; SYM additionally mirrors (i,j) -> (j,i); TRANSPOSE scatters the transposed
// block instead.
template <int NT, int LB, bool SYM, bool TRANSPOSE, int STRIP>
__device__ __forceinline__ void fill_strip(const double *acc, double *strip, int ncol_pad, int lane, int n)
{
  using S = Sched<NT, LB>;
  const int i = lane >> 4, bb = (lane >> 2) & 3, j = lane & 3;
  static_for<0, NT>([&](auto a_) {
    constexpr int a = a_;
    static_for<0, NT>([&](auto b_) {
      constexpr int b = b_;
      // rows 4*tile(a,.)+i lie in row strip a and columns 4*tile(b,.)+j in column strip b, so only the
      // accumulators of fragment row STRIP (direct) / fragment column STRIP (mirror or transpose) contribute
      constexpr bool do_direct = !TRANSPOSE && (a == STRIP);
      constexpr bool do_swap = (TRANSPOSE || SYM) && (b == STRIP);
      if constexpr (do_direct || do_swap)
        static_for<0, 4>([&](auto r_) {
          constexpr int r = r_;
          constexpr unsigned msk = SYM ? S::sym_mask(a, b, r) : S::full_mask(a, b, r);
          if constexpr (msk != 0u)
            {
              if ((msk >> bb) & 1u)
                {
                  const int ti = S::ti(a, r, bb, SYM), tj = S::tj(b, r, bb, SYM);
                  const int R = 4 * ti + i, C = 4 * tj + j; // entry (R,C) of the computed block
                  const double v = acc[acc_idx<NT>(a, b, r)];
                  if (R < n && C < n)
                    {
                      if constexpr (do_direct)
                        strip[(R & 15) * ncol_pad + C] = v;
                      if constexpr (do_swap)
                        if (TRANSPOSE || ti != tj)
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
template <int DIM, int N1D, int NT, int LB, bool REACT>
__global__ void __launch_bounds__(PDH_WAVE, 2) k_diag(const PdhDev P, const int n_owned)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using RC = Rec<DIM, N1D>;
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;
  if (slot >= n_owned)
    return;
  const int agg = P.own_agg[slot];
  double *rec = lds;                      // [64][RC::LEN]
  double *aux = lds + PDH_WAVE * RC::LEN; // [64][2+DIM]: (unused), sigma/2, normal
  constexpr int AUXN = 2 + DIM;

  double lo[DIM], h[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      h[c] = P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - lo[c];
    }

  LaneBasis<DIM, N1D, NT, LB> lb;
  lb.init(P, lane);
  Rotator rt;
  rt.init(lane);

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
          eval_point_record<DIM, N1D>(P.tab, lo, h, x, sqrt(w), rec + lane * RC::LEN);
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        for (int step = 0; step < nsteps; ++step)
          {
            const int pt = 4 * step + kq;
            const char *rb = reinterpret_cast<const char *>(rec + pt * RC::LEN);
            double phi[NT], dphi[NT][DIM];
            static_for<0, NT>([&](auto a_) {
              constexpr int a = a_;
              frag_eval<DIM>(rb, lb.off[a], phi[a], dphi[a]);
            });
            static_for<0, DIM>([&](auto c_) {
              constexpr int c = c_;
              double G[NT];
              static_for<0, NT>([&](auto a_) {
                constexpr int a = a_;
                G[a] = dphi[a][c];
              });
              product<NT, LB, true>(acc, G, G, rt); // sum_q (sqrt(w) d_c phi_i)(sqrt(w) d_c phi_j)
            });
            if constexpr (REACT) // compile-time: a run-time branch here makes hipcc double the accumulators
              {
                double A[NT];
                static_for<0, NT>([&](auto a_) {
                  constexpr int a = a_;
                  A[a] = P.reaction_c * phi[a];
                });
                product<NT, LB, true>(acc, A, phi, rt);
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
          eval_point_record<DIM, N1D>(P.tab, lo, h, x, sqrt(w), rec + lane * RC::LEN);
          aux[lane * AUXN + 1] = 0.5 * sg;
          for (int c = 0; c < DIM; ++c)
            aux[lane * AUXN + 2 + c] = -0.5 * nr[c];
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        for (int step = 0; step < nsteps; ++step)
          {
            const int pt = 4 * step + kq;
            const char *rb = reinterpret_cast<const char *>(rec + pt * RC::LEN);
            const double hs = aux[pt * AUXN + 1];
            double nh[DIM]; // -n/2
            for (int c = 0; c < DIM; ++c)
              nh[c] = aux[pt * AUXN + 2 + c];
            double Phi[NT], U[NT];
            static_for<0, NT>([&](auto a_) {
              constexpr int a = a_;
              double ph, dp[DIM];
              frag_eval<DIM>(rb, lb.off[a], ph, dp);
              double u = hs * ph;
              for (int c = 0; c < DIM; ++c)
                u += nh[c] * dp[c];
              Phi[a] = ph; // sqrt(w) phi
              U[a] = u;    // sqrt(w) (-1/2 grad phi . n + sigma/2 phi)
            });
            product<NT, LB, true>(acc, U, Phi, rt);
            product<NT, LB, true>(acc, Phi, U, rt);
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
    fill_strip<NT, LB, true, false, s>(acc, strip, ncol_pad, lane, n);
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
// Coupling-block kernel: one wave per interior face with an owned side.
// ------------------------------------------------------------------------------------------------
template <int DIM, int N1D, int NT, int LB>
__global__ void __launch_bounds__(PDH_WAVE, 2) k_offdiag(const PdhDev P, const int n_items)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using RC = Rec<DIM, N1D>;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  if (item >= n_items)
    return;
  const int slot = P.it_own[item];
  const int agg = P.own_agg[slot];
  const int nbr = P.it_nbr[item];
  // Chunks of 32 points: lanes 0-31 evaluate the records in P's frame, lanes 32-63 the same points in Q's
  // frame (half the LDS of a 64-point chunk, which would limit the CU to 5 waves).
  constexpr int CH = 32;
  double *recP = lds;                  // own frame        [32][RC::LEN]
  double *recQ = lds + CH * RC::LEN;   // neighbour frame  [32][RC::LEN]
  double *aux = lds + 2 * CH * RC::LEN; // [32][2+DIM]
  constexpr int AUXN = 2 + DIM;

  const int half = lane >> 5, pl = lane & 31;
  double lo[DIM], h[DIM]; // frame this lane evaluates records in
  for (int c = 0; c < DIM; ++c)
    {
      const int box = half ? nbr : agg;
      lo[c] = P.bbox[(int64_t)box * 2 * DIM + c];
      h[c] = P.bbox[(int64_t)box * 2 * DIM + DIM + c] - lo[c];
    }

  LaneBasis<DIM, N1D, NT, LB> lb;
  lb.init(P, lane);
  Rotator rt;
  rt.init(lane);

  double acc[NT * NT * 4];
  for (int t = 0; t < NT * NT * 4; ++t)
    acc[t] = 0.0;

  const int kq = lane >> 4;
  const int64_t pb = P.it_pbeg[item], pe = pb + P.it_pcnt[item];
  for (int64_t base = pb; base < pe; base += CH)
    {
      const int cnt = (int)((pe - base < CH) ? (pe - base) : CH);
      __syncthreads();
      {
        double x[DIM], nr[DIM], w = 0.0, sg = 0.0;
        if (pl < cnt)
          {
            for (int c = 0; c < DIM; ++c)
              {
                x[c] = P.ap_x[c * P.ap_stride + base + pl];
                nr[c] = P.ap_n[c * P.ap_stride + base + pl];
              }
            w = P.ap_wcross[base + pl];
            sg = P.ap_sig[base + pl];
          }
        else
          for (int c = 0; c < DIM; ++c)
            {
              x[c] = lo[c]; // any point with finite basis values; its weight is zero
              nr[c] = 0.0;
            }
        eval_point_record<DIM, N1D>(P.tab, lo, h, x, sqrt(w), (half ? recQ : recP) + pl * RC::LEN);
        if (half == 0)
          {
            aux[pl * AUXN + 1] = -sg;
            for (int c = 0; c < DIM; ++c)
              aux[pl * AUXN + 2 + c] = 0.5 * nr[c];
          }
      }
      __syncthreads();
      const int nsteps = (cnt + 3) >> 2;
      for (int step = 0; step < nsteps; ++step)
        {
          const int pt = 4 * step + kq;
          const char *rbP = reinterpret_cast<const char *>(recP + pt * RC::LEN);
          const char *rbQ = reinterpret_cast<const char *>(recQ + pt * RC::LEN);
          const double msg = aux[pt * AUXN + 1]; // -sigma
          double nh[DIM];                         // n_P / 2
          for (int c = 0; c < DIM; ++c)
            nh[c] = aux[pt * AUXN + 2 + c];
          double A1[NT], A2[NT], B1[NT], B2[NT];
          static_for<0, NT>([&](auto a_) {
            constexpr int a = a_;
            double ph, dp[DIM];
            frag_eval<DIM>(rbP, lb.off[a], ph, dp);
            double u = msg * ph;
            for (int c = 0; c < DIM; ++c)
              u += nh[c] * dp[c];
            A1[a] = u;  // sqrt(w) (1/2 grad phi^P . n_P - sigma phi^P)
            A2[a] = ph; // sqrt(w) phi^P
            frag_eval<DIM>(rbQ, lb.off[a], ph, dp);
            double g = -nh[0] * dp[0];
            for (int c = 1; c < DIM; ++c)
              g -= nh[c] * dp[c];
            B1[a] = ph; // sqrt(w) phi^Q
            B2[a] = g;  // sqrt(w) (-1/2 grad phi^Q . n_P)
          });
          product<NT, LB, false>(acc, A1, B1, rt);
          product<NT, LB, false>(acc, A2, B2, rt);
        }
    }

  const int n = P.n;
  const int ncol_pad = 16 * NT + 2;
  double *strip = lds;
  {
    const int64_t rbase = P.row_base[slot];
    const int rlen = P.row_len[slot];
    const int pos0 = P.it_pos[item];
    static_for<0, NT>([&](auto s_) {
      constexpr int s = s_;
      __syncthreads();
      fill_strip<NT, LB, false, false, s>(acc, strip, ncol_pad, lane, n);
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
  // A[Q,P] = A[P,Q]^T, written into Q's rows when this context owns them
  const int qslot = P.it_nbr_slot[item];
  if (qslot >= 0)
    {
      const int64_t qbase = P.row_base[qslot];
      const int qlen = P.row_len[qslot];
      const int post = P.it_pos_t[item];
      static_for<0, NT>([&](auto s_) {
        constexpr int s = s_;
        __syncthreads();
        fill_strip<NT, LB, false, true, s>(acc, strip, ncol_pad, lane, n);
        __syncthreads();
        const int rows = (n - 16 * s < 16) ? (n - 16 * s) : 16;
        for (int idx = lane; idx < rows * n; idx += PDH_WAVE)
          {
            const int rr = idx / n, c = idx - rr * n;
            const int R = 16 * s + rr;
            P.values[qbase + (int64_t)R * qlen + post + c] = strip[rr * ncol_pad + c];
          }
      });
    }
}

// LDS bytes needed by the two kernels (host side helper).
inline size_t lds_bytes_diag(int dim, int n1d, int nt)
{
  const size_t recs = (size_t)PDH_WAVE * (dim * n1d * 2 + 2 + 2 + dim) * sizeof(double);
  const size_t strip = (size_t)16 * (16 * nt + 2) * sizeof(double);
  return recs > strip ? recs : strip;
}
inline size_t lds_bytes_offdiag(int dim, int n1d, int nt)
{
  const size_t recs = (size_t)32 * (2 * (dim * n1d * 2 + 2) + 2 + dim) * sizeof(double); // 32-point chunks
  const size_t strip = (size_t)16 * (16 * nt + 2) * sizeof(double);
  return recs > strip ? recs : strip;
}
} // namespace pdh
