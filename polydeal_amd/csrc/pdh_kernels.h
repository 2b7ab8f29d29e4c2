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

// Ordering of LDS traffic inside ONE wave.  Every kernel of this library runs single-wave workgroups; the lanes of a wave
// exchange data through LDS (records, staging tiles).  The LDS unit executes the DS instructions of a wave in issue
// order, so a ds_write followed by a ds_read of another lane needs no s_waitcnt and no s_barrier - only the COMPILER must
// be kept from moving memory operations across the hand-off.  __syncthreads() does that too, but it is also a full
// fence: s_waitcnt vmcnt(0), i.e. it waits for every global load and store in flight - which silently undid all prefetching
// of point data and made each hand-off wait for the row stores before it (measured with in-kernel stamps, r02: a 64-point
// chunk of the row kernel took 5.6k cycles instead of 2.8k).
#define PDH_WAVE_SYNC()                                                                                             \
  do                                                                                                                \
    {                                                                                                               \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                        \
      __builtin_amdgcn_wave_barrier();                                                                              \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                        \
    }                                                                                                               \
  while (0)

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
  const int32_t *own_row;  // [n_owned] first dof row of the polytope, relative to the owned row range
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

// Experiment switches for tools/ab_bench.py (never defined in the shipped build): time the kernels without their
// epilogue (-DPDH_EXP_NOSTORE) or without their accumulation loops (-DPDH_EXP_NOCOMPUTE).  Measured on the
// default bench workload: k_diag 5.75 ms compute-only / 0.42 ms store-only (5.9 together); k_offdiag 1.85 ms
// compute-only / 1.90 ms store-only (2.47 together).
#if defined(PDH_EXP_NOCOMPUTE)
#define PDH_EXP_LOOPCOND &&(P.n < 0)
#else
#define PDH_EXP_LOOPCOND
#endif
#if defined(PDH_EXP_NOSTORE)
#define PDH_EXP_EPILOGUE_GUARD                                                                     \
  if (acc[0] != 1.2345e300)                                                                        \
    return;
#else
#define PDH_EXP_EPILOGUE_GUARD
#endif

namespace pdh
{
// Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md: blocks b and b + 8 share one); each XCD has its own
// L2.  Work item of block b such that every XCD works through ONE contiguous eighth of the n items: items that write neighbouring
// pieces of the same rows (the blocks of a polytope's rows in the two-kernel forms) then meet in one L2, where partial lines merge
// before they leave for HBM.  Speed only - correctness never depends on placement.  -DPDH_NO_XCD_CHUNKS: item = block.
__device__ __forceinline__ int xcd_chunked(int b, int n)
{
#ifdef PDH_NO_XCD_CHUNKS
  (void)n;
  return b;
#else
  const int x = b & 7, k = b >> 3, q = n >> 3, r = n & 7;
  return x * q + (x < r ? x : r) + k;
#endif
}

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
// R = 1, 2 go through DPP row_ror (VALU), R = 3 - needed only by the non-symmetric coupling schedule -
// through ds_bpermute_b32 (LDS crossbar).  Measured alternatives on MI355X (profiles/README.md): all three
// through DPP made the kernels VALU-issue bound (every VALU instruction, DPP moves included, takes issue
// time from the f64 MFMA of its SIMD even across waves - tools/probes/coissue_probe.hip); all through
// ds_bpermute made them LDS-pipe bound; a write/read round trip through LDS scratch exposed its latency.
struct Rotator
{
  int addr3; // 4 * source lane for R = 3
  __device__ __forceinline__ void init(int lane) { addr3 = ((lane & ~15) | ((lane + 12) & 15)) * 4; }
  template <int R>
  __device__ __forceinline__ double rot(double x) const
  {
    if constexpr (R == 0)
      return x;
    else if constexpr (R == 1 || R == 2)
      {
        // row_ror:K gives lane m the value of lane (m-K) mod 16 of its row (tools/probes/dpp_probe.hip);
        // mov_dpp (undefined old value): every lane is written, so no zero-initialising v_mov is needed
        constexpr int K = 16 - 4 * R;
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x120 + K, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x120 + K, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
      }
    else
      {
        const int lo = __builtin_amdgcn_ds_bpermute(addr3, __double2loint(x));
        const int hi = __builtin_amdgcn_ds_bpermute(addr3, __double2hiint(x));
        return __hiloint2double(hi, lo);
      }
  }
};

// The rotated copies of one set of NT fragments: v[s][f] = rot_s(fragment f).
template <int NT>
struct RotSet
{
  double v[4][NT];
};

// SMASK: bit s set = rot_s is needed (bit 0 is implied).
template <int NT, int SMASK>
__device__ __forceinline__ void make_rot(const double *X, const Rotator &rt, RotSet<NT> &out)
{
  static_for<0, NT>([&](auto f_) {
    constexpr int f = f_;
    out.v[0][f] = X[f];
    out.v[1][f] = ((SMASK >> 1) & 1) ? rt.template rot<1>(X[f]) : 0.0;
    out.v[2][f] = ((SMASK >> 2) & 1) ? rt.template rot<2>(X[f]) : 0.0;
    out.v[3][f] = ((SMASK >> 3) & 1) ? rt.template rot<3>(X[f]) : 0.0;
  });
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
// sw = sqrt(weight) is folded into direction 0.  dscale[c] (face points only) additionally scales the
// derivative entries of direction c by a per-point factor - the face kernels fold -n_c/2 (or +n_c/2) in, so
// that grad(phi).n-type combinations cost no extra multiplies per basis function.
template <int DIM, int N1D, bool DSCALE>
__device__ __forceinline__ void eval_point_record(const PdhBasisTab &tab, const double *lo, const double *h,
                                                  const double *x, double sw, const double *dscale, double *rec)
{
  constexpr int p = N1D - 1;
  static_for<0, DIM>([&](auto c_) {
    constexpr int c = c_;
    // BoundingBox::real_to_unit (agglomeration_handler.cc:703-704), then centred: the tables are in t = x^ - 1/2
    const double xh = (x[c] - lo[c]) / h[c] - 0.5;
    double ih = 1.0 / h[c];                  // inverse_cell_extents (mapping_box.cc:222)
    double sv = 1.0;
    if constexpr (c == 0)
      {
        sv = sw;
        ih *= sw;
      }
    if constexpr (DSCALE)
      ih *= dscale[c];
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

// Raw table entries (value, derivative) per direction of one fragment's function at the lane's point.
template <int DIM>
struct FragRaw
{
  d2_t t[DIM];
  __device__ __forceinline__ void load(const char *rec_bytes, const int *off)
  {
    for (int c = 0; c < DIM; ++c)
      t[c] = *reinterpret_cast<const d2_t *>(rec_bytes + off[c]);
  }
  // With derivative entries pre-scaled by s_c (face records):  phi  and  u = m phi + sum_c s_c d_c phi
  __device__ __forceinline__ void eval_u(double m, double &phi, double &u) const
  {
    if constexpr (DIM == 2)
      {
        phi = t[0].x * t[1].x;
        u = t[1].x * (m * t[0].x + t[0].y) + t[0].x * t[1].y;
      }
    else
      {
        const double v12 = t[1].x * t[2].x;
        phi = t[0].x * v12;
        u = v12 * (m * t[0].x + t[0].y) + t[0].x * (t[1].y * t[2].x + t[1].x * t[2].y);
      }
  }
  // sqrt(w) phi and sqrt(w) d_c phi
  __device__ __forceinline__ void eval(double &phi, double *dphi) const
  {
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
};

template <int DIM>
__device__ __forceinline__ void frag_eval(const char *rec_bytes, const int *off, double &phi, double *dphi)
{
  FragRaw<DIM> r;
  r.load(rec_bytes, off);
  r.eval(phi, dphi);
}

// Table entries of ONE 4-point step for all NT fragments of this lane, loaded one step ahead of their use.
// Generic form: DIM (value, derivative) pairs per fragment.  Tensor form (3-D FE_DGQ(3): n1d^2 = 16, i.e. a
// fragment is exactly one z-layer, function 16a+m = (m&3, m>>2, a)): the x and y entries are shared by the
// four fragments and the z entries are lane-uniform, so a step needs 6 instead of 12 LDS reads and 15
// instead of 24 multiplies for the gradients (16 instead of 28 operations for the face operands).
template <int DIM, int N1D, int NT, int LB>
struct StepRaw
{
  // Measured on MI355X (tools/ab_bench.py): k_diag +2.6 %, k_offdiag -1 % with the tensor form, so it is
  // switched off; the code is kept because the operation count argument may hold on other parts.
  static constexpr bool TENS = false && (DIM == 3 && N1D == 4 && NT == 4 && LB == 4);
  FragRaw<DIM> r[TENS ? 1 : NT];
  d2_t z[TENS ? 4 : 1];

  __device__ __forceinline__ void load(const char *rb, const LaneBasis<DIM, N1D, NT, LB> &lb)
  {
    if constexpr (TENS)
      {
        r[0].t[0] = *reinterpret_cast<const d2_t *>(rb + lb.off[0][0]);
        r[0].t[1] = *reinterpret_cast<const d2_t *>(rb + lb.off[0][1]);
        static_for<0, 4>([&](auto a_) {
          constexpr int a = a_;
          z[a] = *reinterpret_cast<const d2_t *>(rb + (2 * N1D + a) * 16);
        });
      }
    else
      static_for<0, NT>([&](auto a_) { r[a_].load(rb, lb.off[a_]); });
  }
  // sqrt(w) d_c phi of every fragment (phi only if WANT_PHI)
  template <bool WANT_PHI>
  __device__ __forceinline__ void eval_grad(double *phi, double (*dphi)[DIM]) const
  {
    if constexpr (TENS)
      {
        const double tx = r[0].t[0].y * r[0].t[1].x, ty = r[0].t[0].x * r[0].t[1].y, tz = r[0].t[0].x * r[0].t[1].x;
        static_for<0, 4>([&](auto a_) {
          constexpr int a = a_;
          dphi[a][0] = tx * z[a].x;
          dphi[a][1] = ty * z[a].x;
          dphi[a][2] = tz * z[a].y;
          if constexpr (WANT_PHI)
            phi[a] = tz * z[a].x;
        });
      }
    else
      static_for<0, NT>([&](auto a_) {
        constexpr int a = a_;
        double ph;
        r[a].eval(ph, dphi[a]);
        if constexpr (WANT_PHI)
          phi[a] = ph;
      });
  }
  // face records (derivatives pre-scaled by s_c): phi and u = m phi + sum_c s_c d_c phi of every fragment
  __device__ __forceinline__ void eval_u(double m, double *phi, double *u) const
  {
    if constexpr (TENS)
      {
        const double c_ = r[0].t[0].x * r[0].t[1].x;
        const double a_ = r[0].t[1].x * (m * r[0].t[0].x + r[0].t[0].y) + r[0].t[0].x * r[0].t[1].y;
        static_for<0, 4>([&](auto f_) {
          constexpr int f = f_;
          phi[f] = c_ * z[f].x;
          u[f] = a_ * z[f].x + c_ * z[f].y;
        });
      }
    else
      static_for<0, NT>([&](auto a_) {
        constexpr int a = a_;
        r[a].eval_u(m, phi[a], u[a]);
      });
  }
};

template <int NT>
__device__ __forceinline__ constexpr int acc_idx(int a, int b, int r)
{
  return (a * NT + b) * 4 + r;
}

// acc[a,b,r] += rot_sa(A[a]) (x) rot_sb(B[b]) for the symmetric (a<=b) or the full schedule.
// Needs A rotations {0} (+{1} if SYM) and B rotations {0,1,2} (+{3} if !SYM).
template <int NT, int LB, bool SYM>
__device__ __forceinline__ void product(double *acc, const RotSet<NT> &A, const RotSet<NT> &B)
{
  using S = Sched<NT, LB>;
  static_for<0, NT>([&](auto b_) {
    constexpr int b = b_;
    static_for<0, NT>([&](auto a_) {
      constexpr int a = a_;
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        constexpr unsigned msk = SYM ? S::sym_mask(a, b, r) : S::full_mask(a, b, r);
        if constexpr (msk != 0u)
          acc[acc_idx<NT>(a, b, r)] =
            mfma4(A.v[S::sa(r, SYM)][a], B.v[S::sb(r, SYM)][b], acc[acc_idx<NT>(a, b, r)]);
      });
    });
  });
}
// Full (non-symmetric) schedule with the B rotations formed per fragment right before use: keeps only three
// rotated copies live at a time (the coupling kernel has 64 accumulators and no registers to spare).
template <int NT, int LB>
__device__ __forceinline__ void product_full(double *acc, const double *A, const double *B, const Rotator &rt)
{
  using S = Sched<NT, LB>;
  static_for<0, NT>([&](auto b_) {
    constexpr int b = b_;
    const double R0 = B[b];
    const double R1 = rt.template rot<1>(R0), R2 = rt.template rot<2>(R0), R3 = rt.template rot<3>(R0);
    static_for<0, NT>([&](auto a_) {
      constexpr int a = a_;
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        if constexpr (S::full_mask(a, b, r) != 0u)
          {
            const double Rr = (r == 0) ? R0 : (r == 1) ? R1 : (r == 2) ? R2 : R3;
            acc[acc_idx<NT>(a, b, r)] = mfma4(A[a], Rr, acc[acc_idx<NT>(a, b, r)]);
          }
      });
    });
  });
}
constexpr int ROT_SYM = 0x7;  // rotations 0,1,2: enough for both operands of the symmetric schedule
constexpr int ROT_FULL = 0xf; // B operand of the full schedule
constexpr int ROT_NONE = 0x1; // A operand of the full schedule

// Epilogue helper.  Per-lane constants of the D layout (lane = 16 i + 4 bb + j): for a fragment with `rep`
// distinct blocks and operand rotation s the lane's local tile is (bb+s) & (rep-1).  For full fragments
// (rep = 4) the LDS offsets of an accumulator's entry depend on (sa, sb) only and are precomputed, so that
// scattering an accumulator into the strip is one ds_write with an immediate offset.
template <int NT, int LB>
struct StripMap
{
  using S = Sched<NT, LB>;
  int i, bb, j;
  int dir4[2][4]; // (row & 15) * ncol_pad + (col - 16 b)   for rotations (sa, sb) of full fragments
  int swp4[2][4]; // (col & 15) * ncol_pad + (row - 16 a)
  __device__ __forceinline__ void init(int lane, int ncol_pad)
  {
    i = lane >> 4;
    bb = (lane >> 2) & 3;
    j = lane & 3;
    // static indices only: a run-time indexed member array would move the whole object to scratch memory
    static_for<0, 2>([&](auto sa_) {
      static_for<0, 4>([&](auto sb_) {
        constexpr int sa = sa_, sb = sb_;
        const int r = 4 * ((bb + sa) & 3) + i, c = 4 * ((bb + sb) & 3) + j;
        dir4[sa][sb] = r * ncol_pad + c;
        swp4[sa][sb] = c * ncol_pad + r;
      });
    });
  }
  // row / column of the computed block held by this lane for product (a,b,r)
  template <int A, int R, bool SYM>
  __device__ __forceinline__ int row() const
  {
    constexpr int rp = S::rep(A);
    return 16 * A + 4 * ((bb + S::sa(R, SYM)) & (rp - 1)) + i;
  }
  template <int B, int R, bool SYM>
  __device__ __forceinline__ int col() const
  {
    constexpr int rp = S::rep(B);
    return 16 * B + 4 * ((bb + S::sb(R, SYM)) & (rp - 1)) + j;
  }
};

// Scatter the accumulators' entries that fall into row strip STRIP (rows 16*STRIP.. of the block) into the
// LDS strip [16][ncol_pad]; SYM additionally mirrors (i,j) -> (j,i); TRANSPOSE scatters the transposed
// block instead.  FULL: n == 16 NT, every tile of every fragment is live - no bounds checks, precomputed offsets.
template <int NT, int LB, bool SYM, bool TRANSPOSE, int STRIP, bool FULL>
__device__ __forceinline__ void fill_strip(const double *acc, double *strip, int ncol_pad, const StripMap<NT, LB> &sm, int n)
{
  using S = Sched<NT, LB>;
  static_for<0, NT>([&](auto a_) {
    constexpr int a = a_;
    static_for<0, NT>([&](auto b_) {
      constexpr int b = b_;
      // rows of fragment a lie in row strip a and columns of fragment b in column strip b, so only the
      // accumulators of fragment row STRIP (direct) / fragment column STRIP (mirror or transpose) contribute
      constexpr bool do_direct = !TRANSPOSE && (a == STRIP);
      constexpr bool do_swap = (TRANSPOSE || SYM) && (b == STRIP);
      if constexpr (do_direct || do_swap)
        static_for<0, 4>([&](auto r_) {
          constexpr int r = r_;
          constexpr unsigned msk = SYM ? S::sym_mask(a, b, r) : S::full_mask(a, b, r);
          if constexpr (msk != 0u)
            {
              const double v = acc[acc_idx<NT>(a, b, r)];
              constexpr int sa = S::sa(r, SYM), sb = S::sb(r, SYM);
              if constexpr (FULL && LB == 4)
                {
                  bool ok = true;
                  if constexpr (msk != 0xfu)
                    ok = ((msk >> sm.bb) & 1u) != 0u;
                  if (ok)
                    {
                      if constexpr (do_direct)
                        strip[sm.dir4[sa][sb] + 16 * b] = v;
                      if constexpr (do_swap)
                        {
                          // mirror only strictly off-diagonal tiles (a == b: tile (bb+sa) vs (bb+sb), i.e. r != 0)
                          if constexpr (TRANSPOSE || a != b || r != 0)
                            strip[sm.swp4[sa][sb] + 16 * a] = v;
                        }
                    }
                }
              else
                {
                  const int R = sm.template row<a, r, SYM>(), C = sm.template col<b, r, SYM>();
                  bool ok = (R < n) && (C < n);
                  if constexpr (msk != 0xfu)
                    ok = ok && (((msk >> sm.bb) & 1u) != 0u);
                  if (ok)
                    {
                      if constexpr (do_direct)
                        strip[(R & 15) * ncol_pad + C] = v;
                      if constexpr (do_swap)
                        if (TRANSPOSE || (R >> 2) != (C >> 2))
                          strip[(C & 15) * ncol_pad + R] = v;
                    }
                }
            }
        });
    });
  });
}

// Write rows [16*STRIP, 16*STRIP+rows) of a block from the LDS strip to their CSR positions: row R of the
// block goes to values[base + R*row_len + pos(R,c)], one contiguous segment per row (n <= 64 = one wave store).
// DIAG: own block in deal.II diagonal-first layout (diagonal entry at position 0, columns before it shifted).
template <bool DIAG>
__device__ __forceinline__ void store_strip(double *values, int64_t base, int row_len, int pos0, int diag_first,
                                            const double *strip, int ncol_pad, int strip_idx, int n, int lane)
{
  const int rows = (n - 16 * strip_idx < 16) ? (n - 16 * strip_idx) : 16;
  if (lane < n)
    {
      double *dst = values + base + (int64_t)(16 * strip_idx) * row_len;
      const double *src = strip + lane;
      // batches of 8 rows: all LDS reads of a batch are in flight before the first store waits for its data
      // (a read -> wait -> store chain per row exposed the LDS latency 16 times per strip).  The strip always
      // holds 16 rows, so reading past `rows` is harmless; the stores are predicated.
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8)
        {
          double v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            v[k] = src[(r0 + k) * ncol_pad];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            {
              const int rr = r0 + k;
              int pos = pos0 + lane;
              if constexpr (DIAG)
                if (diag_first)
                  {
                    const int R = 16 * strip_idx + rr;
                    pos = (lane == R) ? 0 : (pos0 + lane + (lane < R ? 1 : 0));
                  }
              if (rr < rows)
                dst[(int64_t)rr * row_len + pos] = v[k];
            }
        }
    }
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
  if ((int)blockIdx.x >= n_owned)
    return;
  const int slot = xcd_chunked((int)blockIdx.x, n_owned); // (neighbouring blocks of a row from one L2: see xcd_chunked)
  const int agg = P.own_agg[slot];
  // Points per chunk.  Large blocks (NT >= 3) run at 2 waves per SIMD (register-limited): 64-point chunks
  // (one point per lane in the point phase, 15.9 KB of LDS per wave at n1d = 4, DIM = 3 = 10 waves per CU).
  // Small blocks reach 4-5 waves per SIMD, where LDS would be the cap: 32-point chunks (measured faster).
  constexpr int CH = (NT >= 3) ? PDH_WAVE : 32;
  double *rec = lds;                // [CH][RC::LEN]
  double *aux = lds + CH * RC::LEN; // [CH][2+DIM]: (unused), sigma/2, -normal/2
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
    for (int64_t base = qb; base < qe PDH_EXP_LOOPCOND; base += CH)
      {
        const int cnt = (int)((qe - base < CH) ? (qe - base) : CH);
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
          if (lane < CH)
            eval_point_record<DIM, N1D, false>(P.tab, lo, h, x, sqrt(w), nullptr, rec + lane * RC::LEN);
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        StepRaw<DIM, N1D, NT, LB> raw; // table entries of the step being computed, loaded one step ahead
        raw.load(reinterpret_cast<const char *>(rec + kq * RC::LEN), lb);
        for (int step = 0; step < nsteps; ++step)
          {
            double phi[NT], dphi[NT][DIM];
            raw.template eval_grad<REACT>(phi, dphi);
            {
              // prefetch the next step (the last iteration re-reads its own point: always a valid record)
              const int ptn = 4 * ((step + 1 < nsteps) ? step + 1 : step) + kq;
              raw.load(reinterpret_cast<const char *>(rec + ptn * RC::LEN), lb);
            }
            RotSet<NT> RG[DIM];
            static_for<0, DIM>([&](auto c_) {
              constexpr int c = c_;
              double G[NT];
              static_for<0, NT>([&](auto a_) {
                constexpr int a = a_;
                G[a] = dphi[a][c];
              });
              make_rot<NT, ROT_SYM>(G, rt, RG[c]);
            });
            static_for<0, DIM>([&](auto c_) {
              constexpr int c = c_;
              product<NT, LB, true>(acc, RG[c], RG[c]); // sum_q (sqrt(w) d_c phi_i)(sqrt(w) d_c phi_j)
            });
            if constexpr (REACT) // compile-time: a run-time branch here makes hipcc double the accumulators
              {
                RotSet<NT> RP, RA;
                make_rot<NT, ROT_SYM>(phi, rt, RP);
                static_for<0, NT>([&](auto a_) {
                  constexpr int a = a_;
                  RA.v[0][a] = P.reaction_c * RP.v[0][a];
                  RA.v[1][a] = P.reaction_c * RP.v[1][a];
                });
                product<NT, LB, true>(acc, RA, RP);
              }
          }
      }
  }

  // ---- own-side face terms (all faces of the polytope, boundary included) ----------------------
  {
    const int64_t pb = P.ap_ptr[slot], pe = P.ap_ptr[slot + 1];
    for (int64_t base = pb; base < pe PDH_EXP_LOOPCOND; base += CH)
      {
        const int cnt = (int)((pe - base < CH) ? (pe - base) : CH);
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
          if (lane < CH)
            {
              double ds[DIM]; // -n_c / 2 folded into the derivative entries
              for (int c = 0; c < DIM; ++c)
                ds[c] = -0.5 * nr[c];
              eval_point_record<DIM, N1D, true>(P.tab, lo, h, x, sqrt(w), ds, rec + lane * RC::LEN);
              aux[lane * AUXN + 1] = 0.5 * sg;
            }
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        StepRaw<DIM, N1D, NT, LB> raw;
        double hs; // sigma/2 of the step being computed
        raw.load(reinterpret_cast<const char *>(rec + kq * RC::LEN), lb);
        hs = aux[kq * AUXN + 1];
        for (int step = 0; step < nsteps; ++step)
          {
            double Phi[NT], U[NT];
            // Phi = sqrt(w) phi,  U = sqrt(w) (-1/2 grad phi . n + sigma/2 phi)
            raw.eval_u(hs, Phi, U);
            {
              const int ptn = 4 * ((step + 1 < nsteps) ? step + 1 : step) + kq;
              raw.load(reinterpret_cast<const char *>(rec + ptn * RC::LEN), lb);
              hs = aux[ptn * AUXN + 1];
            }
            RotSet<NT> RU, RPhi;
            make_rot<NT, ROT_SYM>(U, rt, RU);
            make_rot<NT, ROT_SYM>(Phi, rt, RPhi);
            product<NT, LB, true>(acc, RU, RPhi);
            product<NT, LB, true>(acc, RPhi, RU);
          }
      }
  }

  // ---- epilogue: mirror + write rows in CSR order ------------------------------------------------
  PDH_EXP_EPILOGUE_GUARD
  const int n = P.n;
  const int ncol_pad = 16 * NT + 2;
  double *strip = lds; // overlays the point records
  const int64_t rbase = P.row_base[slot];
  const int rlen = P.row_len[slot];
  const int L = P.diag_L[slot];
  StripMap<NT, LB> sm;
  sm.init(lane, ncol_pad);
  const bool full = (n == 16 * NT);
  static_for<0, NT>([&](auto s_) {
    constexpr int s = s_;
    __syncthreads();
    if constexpr (LB == 4) // the unchecked variant exists for completely filled blocks only
      {
        if (full)
          fill_strip<NT, LB, true, false, s, true>(acc, strip, ncol_pad, sm, n);
        else
          fill_strip<NT, LB, true, false, s, false>(acc, strip, ncol_pad, sm, n);
      }
    else
      fill_strip<NT, LB, true, false, s, false>(acc, strip, ncol_pad, sm, n);
    __syncthreads();
    store_strip<true>(P.values, rbase, rlen, L, P.diag_first, strip, ncol_pad, s, n, lane);
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
  if ((int)blockIdx.x >= n_items)
    return;
  const int item = xcd_chunked((int)blockIdx.x, n_items);
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
  for (int64_t base = pb; base < pe PDH_EXP_LOOPCOND; base += CH)
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
        double ds[DIM]; // P frame: +n_c/2, Q frame: -n_c/2 folded into the derivative entries
        for (int c = 0; c < DIM; ++c)
          ds[c] = (half ? -0.5 : 0.5) * nr[c];
        eval_point_record<DIM, N1D, true>(P.tab, lo, h, x, sqrt(w), ds, (half ? recQ : recP) + pl * RC::LEN);
        if (half == 0)
          aux[pl * AUXN + 1] = -sg;
      }
      __syncthreads();
      const int nsteps = (cnt + 3) >> 2;
      for (int step = 0; step < nsteps; ++step)
        {
          const int pt = 4 * step + kq;
          const char *rbP = reinterpret_cast<const char *>(recP + pt * RC::LEN);
          const char *rbQ = reinterpret_cast<const char *>(recQ + pt * RC::LEN);
          const double msg = aux[pt * AUXN + 1]; // -sigma
          double A1[NT], A2[NT], B1[NT], B2[NT];
          static_for<0, NT>([&](auto a_) {
            constexpr int a = a_;
            FragRaw<DIM> r;
            r.load(rbP, lb.off[a]);
            // A2 = sqrt(w) phi^P,  A1 = sqrt(w) (1/2 grad phi^P . n_P - sigma phi^P)
            r.eval_u(msg, A2[a], A1[a]);
            r.load(rbQ, lb.off[a]);
            // B1 = sqrt(w) phi^Q,  B2 = sqrt(w) (-1/2 grad phi^Q . n_P)
            r.eval_u(0.0, B1[a], B2[a]);
          });
          product_full<NT, LB>(acc, A1, B1, rt);
          product_full<NT, LB>(acc, A2, B2, rt);
        }
    }

  PDH_EXP_EPILOGUE_GUARD
  const int n = P.n;
  const int ncol_pad = 16 * NT + 2;
  double *strip = lds;
  StripMap<NT, LB> sm;
  sm.init(lane, ncol_pad);
  const bool full = (n == 16 * NT);
  {
    const int64_t rbase = P.row_base[slot];
    const int rlen = P.row_len[slot];
    const int pos0 = P.it_pos[item];
    static_for<0, NT>([&](auto s_) {
      constexpr int s = s_;
      __syncthreads();
      if constexpr (LB == 4)
        {
          if (full)
            fill_strip<NT, LB, false, false, s, true>(acc, strip, ncol_pad, sm, n);
          else
            fill_strip<NT, LB, false, false, s, false>(acc, strip, ncol_pad, sm, n);
        }
      else
        fill_strip<NT, LB, false, false, s, false>(acc, strip, ncol_pad, sm, n);
      __syncthreads();
      store_strip<false>(P.values, rbase, rlen, pos0, 0, strip, ncol_pad, s, n, lane);
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
        if constexpr (LB == 4)
          {
            if (full)
              fill_strip<NT, LB, false, true, s, true>(acc, strip, ncol_pad, sm, n);
            else
              fill_strip<NT, LB, false, true, s, false>(acc, strip, ncol_pad, sm, n);
          }
        else
          fill_strip<NT, LB, false, true, s, false>(acc, strip, ncol_pad, sm, n);
        __syncthreads();
        store_strip<false>(P.values, qbase, qlen, post, 0, strip, ncol_pad, s, n, lane);
      });
    }
}

// LDS bytes needed by the two kernels (host side helper).
inline size_t lds_bytes_diag(int dim, int n1d, int nt)
{
  const size_t recs = (size_t)(nt >= 3 ? PDH_WAVE : 32) * (dim * n1d * 2 + 2 + 2 + dim) * sizeof(double); // CH-point chunks
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
