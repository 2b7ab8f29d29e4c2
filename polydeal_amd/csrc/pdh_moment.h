// pdh_moment.h — the moment ("quadrature-free contraction") form of the SIP blocks, 3-D tensor-product bases.
//
// The direct kernels (pdh_kernels.h) contract  sum_q w_q f_i(x_q) g_j(x_q)  for all n^2 pairs: 2 d Nq n^2 flops per
// polytope, MFMA-bound for n = 64.  Here the same sums are reorganised exactly (no approximation, the quadrature
// rule stays the caller's): every integrand of the path is a product of 1-D polynomials of the bounding-box
// coordinates, e.g. for the volume term
//     d_c phi_i d_c phi_j = h_c^-2 [B'_k B'_l](x^_c)  prod_{d != c} [B_k B_l](x^_d),       i = (k0,k1,k2), j = (l0,l1,l2)
// and each 1-D product of degree <= 2p is expanded ONCE (host tables, exact Gauss quadrature in long double) in the
// L2-orthonormal Legendre polynomials L_a of [0,1]:   B_k B_l = sum_a E[k,l,a] L_a,  B'_k B'_l = sum_a D[k,l,a] L_a,
// (B_k B_l)' = sum_a Fs[k,l,a] L_a.   Then with the MOMENTS of the quadrature
//     M[a0,a1,a2] = sum_q w_q L_a0(x^_0) L_a1(x^_1) L_a2(x^_2)                         ((2p+1)^3 numbers)
// the block is   A_ij = sum_a X[k0,l0,a0] Y[k1,l1,a1] Z[k2,l2,a2] M[a]   - evaluated by sum factorisation.
// Face terms use moments weighted by  w sigma  and  -w n_c / 2;  a coupling block A[P,Q] uses tables of the mixed
// products B^P_k(xi^P(t)) B^Q_l(xi^Q(t)) (the two bounding-box frames differ by an axis-aligned affine map), built
// per face on the device by an exact Gauss rule.  Cost per polytope: O(Nq (2p+1)^3) for the moments plus
// O((p+1)^2d (2p+1)) for the contraction, instead of O(Nq n^2) - for FE_DGQ(3) about 8x fewer operations, which moves
// the headline workload off the f64 MFMA roof towards the HBM write roof of the matrix values.
// Where the work runs: moment accumulation on the f64 MFMA (a GEMM over the points, struct MomentAcc); contraction
// stage 1 on the VALU; stages 2 and 3 on the MFMA for FE_DGQ(3) (k_*<4, true>), on the VALU otherwise (k_*<N, false>).
// Numerics: every expansion is exact for polynomials, so results differ from the direct form by rounding only
// (1e-14 relative measured; tests/test_gpu_parity.py runs both forms against the oracle, tests/test_moment_math.py
// checks the algebra in NumPy).
//
// The statement of WHAT is computed is unchanged (reference include/poly_utils.h:2040-2084, 1870-1926, see
// pdh_kernels.h); this file only changes the order of summation.
#pragma once
#include "pdh_kernels.h"

// experiment switches (tools/ab_bench.py; never defined in the shipped build, results are garbage with them):
// -DPDHM_EXP=1 no per-face tables, 2 no moment accumulation, 3 no contraction, 4 no stores
#ifndef PDHM_EXP
#define PDHM_EXP 0
#endif
#ifndef PDHM_OFF_FULL
#define PDHM_OFF_FULL 0
#endif
#if PDHM_EXP == 4
#define PDHM_STORE_OK (P.n < 0)
#else
#define PDHM_STORE_OK true
#endif
#if PDHM_EXP == 3
#define PDHM_SLABS (P.n < 0 ? 4 : 0)
#else
#define PDHM_SLABS 4
#endif

namespace pdhm
{
using pdh::static_for;

template <int N1D>
struct MT
{
  static constexpr int NA = 2 * N1D - 1; // Legendre modes 0 .. 2p
  static constexpr int NAP = NA + 1;     // padded (even: 16-byte rows)
  static constexpr int NG = 2 * N1D;     // Gauss points of the per-face table rule: exact to degree 4p+3 >= 4p
  static constexpr int PAIRS = N1D * N1D;
  static constexpr int TAB = PAIRS * NAP; // one expansion table [k][l][NAP] in the global buffer
  // In LDS the table rows and the T2 rows use a stride of NAP + 2 doubles: with 64-byte rows the 16 rows a wave
  // touches in one ds_read_b128 (lanes differing in two 1-D indices) fall on four bank groups only - a 16-way conflict
  // that made the contraction 8x slower than its instruction count; 80-byte rows are conflict-free.
  static constexpr int RS = NAP + 2;
  static constexpr int LTAB = PAIRS * RS;
  // layout of the device table buffer (doubles); filled by pdh_capi.cpp:build_moment_tables
  static constexpr int OFF_E = 0, OFF_D = TAB, OFF_FS = 2 * TAB, OFF_GX = 3 * TAB, OFF_GL = OFF_GX + NG /* [NA][NG] */,
                       OFF_BV = OFF_GL + NA * NG /* [N1D][NG] */, OFF_BD = OFF_BV + N1D * NG, SIZE = OFF_BD + N1D * NG;
};

// L_a(x) = sqrt(2a+1) P_a(2x-1), a = 0 .. NA-1, by the three-term recurrence of the ORTHONORMAL polynomials
//   L_{k+1} = A_k t L_k - C_k L_{k-1},  t = 2x-1,  A_k = sqrt((2k+1)(2k+3))/(k+1),  C_k = k/(k+1) sqrt((2k+3)/(2k-1))
// (three operations per degree; the record phase is VALU time that the f64 MFMA of the same SIMD cannot overlap).
template <int NA>
__device__ __forceinline__ void legendre01(double x, double *L)
{
  const double t = 2.0 * x - 1.0;
  L[0] = 1.0;
  if constexpr (NA > 1)
    L[1] = 1.7320508075688772935 * t;
  static_for<1, NA - 1>([&](auto k_) {
    constexpr int k = k_;
    const double Ak = __builtin_sqrt((double)((2 * k + 1) * (2 * k + 3))) / (double)(k + 1);
    const double Ck = (double)k / (double)(k + 1) * __builtin_sqrt((double)(2 * k + 3) / (double)(2 * k - 1));
    L[k + 1] = Ak * (t * L[k]) - Ck * L[k - 1];
  });
}

typedef double d2_t __attribute__((ext_vector_type(2)));

// The unrolled contraction loops below are split into basic blocks by branches on a value the compiler cannot see
// through (a fresh `asm volatile` zero per iteration; identical conditions would be merged again).  Without them the
// scheduler hoists the LDS row loads of all iterations to the top of one huge block and spills them at once
// (454 VGPRs wanted); sched_barrier / memory-clobber fences made it worse.
#define PDHM_BLOCK(...)                                                                                              \
  do                                                                                                                 \
    {                                                                                                                \
      int z_ = 0;                                                                                                    \
      asm volatile("" : "+s"(z_));                                                                                   \
      if (z_ == 0)                                                                                                   \
        {                                                                                                            \
          __VA_ARGS__;                                                                                                  \
        }                                                                                                            \
    }                                                                                                                \
  while (0)

// sum_a t[a] * r[a]: t in registers (never addressed through a vector pointer: that would pin it to scratch memory),
// r a 16-byte aligned row in LDS
template <int NA>
__device__ __forceinline__ double dot_row(const double *t, const double *r)
{
  double s = 0.0;
  static_for<0, (NA + 1) / 2>([&](auto h_) {
    constexpr int hh = h_;
    const d2_t b = *reinterpret_cast<const d2_t *>(r + 2 * hh);
    s += t[2 * hh] * b.x;
    if constexpr (2 * hh + 1 < NA)
      s += t[2 * hh + 1] * b.y;
  });
  return s;
}

// VALU form of stages 2 and 3 (all elements but FE_DGQ(3)).  T1 holds the stage-1 arrays [N1D][NA (a0)][NAP (a1)] of the current slab;
// stage 2 for one X-type: T2[(l2,l1,k1)][NAP (a0)] = sum over the listed (Y table, T1 array, scale) terms of
//   scale * sum_a1 Y[k1,l1,a1] T1[arr][l2][a0][a1]
template <int N1D>
struct Term
{
  const double *ytab; // [PAIRS][NAP] in LDS
  int arr;            // T1 array
  double scale;
};

// SWAP = false: this lane owns the column index l (function of the second side), the loops run over row indices k;
// SWAP = true: the lane owns k, the loops run over l.  Tables are always indexed [k][l].
template <int N1D, int NTERM, bool SWAP>
__device__ __forceinline__ void stage2(const Term<N1D> (&terms)[NTERM], const double *T1, double *T2, int lane)
{
  using M = MT<N1D>;
  constexpr int NA = M::NA, NAP = M::NAP;
  if (lane < N1D * N1D * N1D)
    {
      const int low = lane % N1D, mid = (lane / N1D) % N1D, hi = lane / (N1D * N1D);
      const int pair = SWAP ? (mid * N1D + low) : (low * N1D + mid); // (k1, l1)
      double out[NAP];
      for (int a0 = 0; a0 < NAP; ++a0)
        out[a0] = 0.0;
      static_for<0, NTERM>([&](auto t_) {
        constexpr int t = t_;
        double y[NAP];
        static_for<0, NAP / 2>([&](auto h_) {
          constexpr int hh = h_;
          const d2_t v = *reinterpret_cast<const d2_t *>(terms[t].ytab + pair * M::RS + 2 * hh);
          y[2 * hh] = v.x * terms[t].scale;
          y[2 * hh + 1] = v.y * terms[t].scale;
        });
        const double *t1 = T1 + ((terms[t].arr * N1D + hi) * NA) * NAP;
        // blocks of four rows: 16 row loads in flight, then 28 FMAs
        static_for<0, (NA + 3) / 4>([&](auto g_) {
          constexpr int g = g_;
          PDHM_BLOCK(static_for<4 * g, (4 * g + 4 < NA ? 4 * g + 4 : NA)>([&](auto a0_) {
            constexpr int a0 = a0_;
            out[a0] += dot_row<NA>(y, t1 + a0 * NAP);
          }));
        });
      });
      static_for<0, NAP / 2>([&](auto h_) {
        constexpr int hh = h_;
        d2_t v;
        v.x = out[2 * hh];
        v.y = out[2 * hh + 1];
        *reinterpret_cast<d2_t *>(T2 + ((low * N1D + hi) * N1D + mid) * M::RS + 2 * hh) = v;
      });
    }
}

// stage 3 for one X-type; (o0,o1,o2) = multi-index of the function this lane owns, the slab's other-side functions are
// enumerated as r = s1 * N1D + s0 (all N1D^2 combinations; those that are not in the space - FE_AggloDGP - are computed
// and dropped by the caller):   out[r] += sum_a0 X[k0, l0, a0] T2[(o2, o1, s1)][a0],  (k0,l0) = SWAP ? (o0,s0) : (s0,o0)
template <int N1D, bool SWAP>
__device__ __forceinline__ void stage3(const double *xtab, const double *T2, int o0, int o1, int o2, double *out)
{
  using M = MT<N1D>;
  constexpr int NA = M::NA, NAP = M::NAP;
  // the N1D rows of T2 this lane needs (one per s1) are read once and stay in registers for all s0: LDS return
  // bandwidth (128 B/clk per CU), not the FMA count, bounds this kernel
  double t2[N1D][NAP], xv[2][NAP];
  auto load_x = [&](int s0, double *dst) {
    const int pair = SWAP ? (o0 * N1D + s0) : (s0 * N1D + o0);
    static_for<0, NAP / 2>([&](auto h_) {
      constexpr int hh = h_;
      const d2_t v = *reinterpret_cast<const d2_t *>(xtab + pair * M::RS + 2 * hh);
      dst[2 * hh] = v.x;
      dst[2 * hh + 1] = v.y;
    });
  };
  PDHM_BLOCK(static_for<0, N1D>([&](auto s1_) {
    constexpr int s1 = s1_;
    static_for<0, NAP / 2>([&](auto h_) {
      constexpr int hh = h_;
      const d2_t v = *reinterpret_cast<const d2_t *>(T2 + ((s1 * N1D + o2) * N1D + o1) * M::RS + 2 * hh);
      t2[s1][2 * hh] = v.x;
      t2[s1][2 * hh + 1] = v.y;
    });
  });
             load_x(0, xv[0]));
  static_for<0, N1D>([&](auto s0_) {
    constexpr int s0 = s0_;
    // the X row of the next s0 is requested before this one is consumed (LDS latency behind the FMAs)
    PDHM_BLOCK(
      if constexpr (s0 + 1 < N1D) load_x(s0 + 1, xv[(s0 + 1) & 1]);
      static_for<0, N1D>([&](auto s1_) {
        constexpr int s1 = s1_;
        double sum = out[s1 * N1D + s0];
        static_for<0, NA>([&](auto a_) {
          constexpr int a = a_;
          sum += xv[s0 & 1][a] * t2[s1][a];
        });
        out[s1 * N1D + s0] = sum;
      }));
  });
}

// row index of the basis function with multi-index (k0,k1,k2), or -1 if the space does not contain it (FE_AggloDGP);
// every lane holds the packed multi-index of its own function
__device__ __forceinline__ int row_of(bool live, int packed_own, int k0, int k1, int k2)
{
  const unsigned long long m = __ballot(live && packed_own == (k0 | (k1 << 8) | (k2 << 16)));
  return m ? (int)__builtin_ctzll(m) : -1;
}

// one past the last row whose k2 is <= `k2` (rows are ordered with k2 non-decreasing: FE_DGQ lexicographic, x fastest;
// FE_AggloDGP "for iz: for iy: for ix"); l2 = this lane's own k2, live = lane < n
__device__ __forceinline__ int slab_end(bool live, int l2, int k2)
{
  return (int)__builtin_popcountll(__ballot(live && l2 <= k2));
}


// ------------------------------------------------------------------------------------------------------------
// Moment accumulation on the f64 MFMA.  M[(a0,a1)][(a2,t)] = sum_q [L_a0 L_a1](q) [s_t L_a2](q) is a GEMM over the
// points: A rows = the NA^2 pairs (a0,a1) in NFA fragments of 16, B columns = blocks of the four weights t for one
// a2 (NFB fragments of four blocks).  A first version kept the per-point factors wave-uniform and broadcast them
// from LDS into VALU FMAs: LDS return bandwidth (128 B/clk per CU, a broadcast still returns 64 lanes) made it 9x
// slower than this.  Operand layout as in pdh_kernels.h: lane = 16 k + 4 blk + idx (k = point of the 4-point step).
// ------------------------------------------------------------------------------------------------------------
template <int N1D>
struct MomentAcc
{
  using M = MT<N1D>;
  static constexpr int NA = M::NA, NAP = M::NAP, ROWS = NA * NA, NFA = (ROWS + 15) / 16, NFB = (NA + 3) / 4;
  // record strides are ODD numbers of doubles: the record phase writes one record per lane, and with an even stride
  // such as 24 doubles (48 dwords) the 64 lanes of a ds_write fall on 4 bank offsets - a 16-way conflict on every write
  // (the face records keep an even stride: 2-way conflicts only with 32 points per chunk, and one more double per record
  // would push the coupling kernel over 20 KB of LDS = 7 instead of 8 waves per CU)
  static constexpr int REC = 2 * NA + 4 * NAP + 4; // L0[NA], L1[NA], (s_t L2)[4][NAP], zero pair, pad
  static constexpr int ZERO = (2 * NA + 4 * NAP) * 8;
  static constexpr int CH = 32; // face points per chunk
  // volume points need one weight only: smaller records, 64 points per chunk (one point per lane in the record phase)
  static constexpr int VREC = (2 * NA + NAP + 2) | 1; // L0[NA], L1[NA], (w L2)[NAP], zero pair, pad
  static constexpr int VZERO = (2 * NA + NAP) * 8;
  static constexpr int VCH = 64;
  // If the last A fragment holds a single live row (NA^2 = 16 m + 1: 49 pairs for FE_DGQ(3), 25 for p = 2, 9 for p = 1) its
  // NFB * 4 (+2) MFMAs per step would run at 1/16 row utilisation.  That row - the pair (NA-1, NA-1) - is computed
  // transposed instead: the B fragments act as rows and one column carries y = L_{NA-1}(x0) L_{NA-1}(x1), NFB (+1) MFMAs.
  static constexpr bool LAST_ROW = (ROWS % 16 == 1) && NFA > 1;
  static constexpr int NFAM = LAST_ROW ? NFA - 1 : NFA; // A fragments multiplied in the regular way
  int offA0[NFA], offA1[NFA], offA0v[NFA], offA1v[NFA], offBv, offBf[NFB], offY0, offY0v;
  double accyv, accyf[NFB]; // transposed products of the last row
  pdh::Rotator rt;
  double accv[NFA][2];      // volume: columns a2 (two blocks, B replicated [F0,F1,F0,F1]), rotations 0,1
  double accf[NFA][NFB][4]; // faces
  __device__ __forceinline__ void init(int lane)
  {
    const int blk = (lane >> 2) & 3, idx = lane & 3;
    static_for<0, NFA>([&](auto a_) {
      constexpr int a = a_;
      const int row = 16 * a + 4 * blk + idx;
      offA0[a] = row < ROWS ? (row / NA) * 8 : ZERO;
      offA1[a] = row < ROWS ? (NA + row % NA) * 8 : ZERO;
      offA0v[a] = row < ROWS ? (row / NA) * 8 : VZERO;
      offA1v[a] = row < ROWS ? (NA + row % NA) * 8 : VZERO;
      accv[a][0] = accv[a][1] = 0.0;
      static_for<0, NFB>([&](auto b_) {
        constexpr int b = b_;
        for (int r = 0; r < 4; ++r)
          accf[a][b][r] = 0.0;
      });
    });
    // y operand: column 0 of every lane block carries L_{NA-1}(x0) * L_{NA-1}(x1) (offY0 / the L1 entry NA doubles later)
    offY0 = idx == 0 ? (NA - 1) * 8 : ZERO - NA * 8;
    offY0v = idx == 0 ? (NA - 1) * 8 : VZERO - NA * 8;
    accyv = 0.0;
    static_for<0, NFB>([&](auto b_) {
      constexpr int b = b_;
      accyf[b] = 0.0;
    });
    const int a2v = 4 * (blk & 1) + idx;
    offBv = a2v < NA ? (2 * NA + a2v) * 8 : VZERO;
    static_for<0, NFB>([&](auto b_) {
      constexpr int b = b_;
      const int a2 = 4 * b + blk;
      offBf[b] = a2 < NA ? (2 * NA + idx * NAP + a2) * 8 : ZERO;
    });
    rt.init(lane);
  }
  // volume record of one point (all 64 lanes, one point each); dead points: w = 0
  static __device__ __forceinline__ void write_volume_record(double *r, const double *xu, double w)
  {
    double L[NA];
    legendre01<NA>(xu[0], L);
    for (int a = 0; a < NA; ++a)
      r[a] = L[a];
    legendre01<NA>(xu[1], L);
    for (int a = 0; a < NA; ++a)
      r[NA + a] = L[a];
    legendre01<NA>(xu[2], L);
    for (int a = 0; a < NA; ++a)
      r[2 * NA + a] = w * L[a];
    r[2 * NA + NA] = 0.0;
    r[2 * NA + NAP] = 0.0;
    r[2 * NA + NAP + 1] = 0.0;
  }
  // face record of one point, written by two lanes: half 0 (lanes 0-31) the factors of directions 0 and 1, half 1
  // (lanes 32-63) the four weighted copies of direction 2;  xa, xb = the unit coordinates this half needs
  static __device__ __forceinline__ void write_face_record_half(double *r, int half, double xa, double xb, const double *s)
  {
    double L[NA];
    legendre01<NA>(xa, L);
    if (half == 0)
      {
        for (int a = 0; a < NA; ++a)
          r[a] = L[a];
        legendre01<NA>(xb, L);
        for (int a = 0; a < NA; ++a)
          r[NA + a] = L[a];
      }
    else
      {
        for (int t = 0; t < 4; ++t)
          {
            for (int a = 0; a < NA; ++a)
              r[2 * NA + t * NAP + a] = s[t] * L[a];
            r[2 * NA + t * NAP + NA] = 0.0;
          }
        r[2 * NA + 4 * NAP] = 0.0;
        r[2 * NA + 4 * NAP + 1] = 0.0;
      }
  }
  // ---- full chunks: software-pipelined, fully unrolled step loops ------------------------------------------------------
  // hipcc turns the prefetching loops below (volume_chunk / face_chunk) back into "load, wait, multiply, MFMA" with ten
  // address computations per step (measured r02: the MFMA pipe of k_mdiag was busy 45 % of the time, 43 % of the wave
  // cycles were issue stalls).  For full chunks every operand address is  per-lane base + compile-time step offset, so
  // the bases are computed once per kernel and the reads are written as `ds_read_b64 ... offset:imm` in inline asm, one
  // step ahead of the MFMAs that consume them; sched_barriers keep the compiler from undoing the order.  The rotated B
  // operands come straight from LDS as well (an LDS read overlaps with the f64 MFMA, a DPP move does not).
  unsigned adA0v[NFAM], adA1v[NFAM], adYv, adBv[2];     // volume records
  unsigned adA0[NFAM], adA1[NFAM], adY, adB[NFB][4];    // face records
  static constexpr int VSTEP = 4 * VREC * 8, FSTEP = 4 * REC * 8; // bytes per 4-point step
  static_assert(VSTEP * (VCH / 4) < 65536 && FSTEP * (CH / 4) < 65536, "ds_read immediate offsets");
  __device__ __forceinline__ void init_addr(const double *rec, int lane)
  {
    typedef __attribute__((address_space(3))) const char lds_cchar;
    const unsigned base = (unsigned)(uintptr_t)(lds_cchar *)reinterpret_cast<const char *>(rec);
    const int kq = lane >> 4, blk = (lane >> 2) & 3, idx = lane & 3;
    const unsigned bv = base + kq * VREC * 8, bf = base + kq * REC * 8;
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      adA0v[a] = bv + offA0v[a];
      adA1v[a] = bv + offA1v[a];
      adA0[a] = bf + offA0[a];
      adA1[a] = bf + offA1[a];
    });
    adYv = bv + offY0v;
    adY = bf + offY0;
    static_for<0, 2>([&](auto r_) {
      constexpr int r = r_;
      const int a2 = 4 * ((blk + r) & 1) + idx;
      adBv[r] = bv + (a2 < NA ? (2 * NA + a2) * 8 : VZERO);
    });
    static_for<0, NFB>([&](auto b_) {
      constexpr int b = b_;
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        const int a2 = 4 * b + ((blk + r) & 3);
        adB[b][r] = bf + (a2 < NA ? (2 * NA + idx * NAP + a2) * 8 : ZERO);
      });
    });
  }
  template <int IMM>
  static __device__ __forceinline__ void lds_read(double &dst, unsigned addr)
  {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
  }
  struct VBuf
  {
    double r0[NFAM], r1[NFAM], y0, y1, b0, b1;
  };
  struct FBuf
  {
    double r0[NFAM], r1[NFAM], y0, y1, b[NFB][4];
  };
  template <int STEP>
  __device__ __forceinline__ void vload(VBuf &v) const
  {
    constexpr int imm = STEP * VSTEP;
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      lds_read<imm>(v.r0[a], adA0v[a]);
      lds_read<imm>(v.r1[a], adA1v[a]);
    });
    if constexpr (LAST_ROW)
      {
        lds_read<imm>(v.y0, adYv);
        lds_read<imm + NA * 8>(v.y1, adYv);
      }
    lds_read<imm>(v.b0, adBv[0]);
    lds_read<imm>(v.b1, adBv[1]);
  }
  template <int STEP>
  __device__ __forceinline__ void fload(FBuf &v) const
  {
    constexpr int imm = STEP * FSTEP;
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      lds_read<imm>(v.r0[a], adA0[a]);
      lds_read<imm>(v.r1[a], adA1[a]);
    });
    if constexpr (LAST_ROW)
      {
        lds_read<imm>(v.y0, adY);
        lds_read<imm + NA * 8>(v.y1, adY);
      }
    static_for<0, NFB>([&](auto b_) {
      constexpr int b = b_;
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        lds_read<imm>(v.b[b][r], adB[b][r]);
      });
    });
  }
  // s_waitcnt lgkmcnt(0) that the consumers of the buffer depend on (the compiler does not track the asm reads)
  static __device__ __forceinline__ void vwait(VBuf &v)
  {
    static_assert(NFAM <= 3, "operand list written for at most three regular A fragments");
    if constexpr (NFAM == 3)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(v.r0[0]), "+v"(v.r1[0]), "+v"(v.r0[1]), "+v"(v.r1[1]), "+v"(v.r0[NFAM - 1]), "+v"(v.r1[NFAM - 1]), "+v"(v.y0),
                     "+v"(v.y1), "+v"(v.b0), "+v"(v.b1));
    else if constexpr (NFAM == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v.r0[0]), "+v"(v.r1[0]), "+v"(v.r0[NFAM - 1]), "+v"(v.r1[NFAM - 1]), "+v"(v.y0), "+v"(v.y1), "+v"(v.b0), "+v"(v.b1));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v.r0[0]), "+v"(v.r1[0]), "+v"(v.y0), "+v"(v.y1), "+v"(v.b0), "+v"(v.b1));
  }
  static __device__ __forceinline__ void fwait(FBuf &v)
  {
    static_assert(NFB <= 2, "operand list written for at most two B fragments");
    if constexpr (NFAM == 3)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(v.r0[0]), "+v"(v.r1[0]), "+v"(v.r0[1]), "+v"(v.r1[1]), "+v"(v.r0[NFAM - 1]), "+v"(v.r1[NFAM - 1]), "+v"(v.y0),
                     "+v"(v.y1));
    else if constexpr (NFAM == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v.r0[0]), "+v"(v.r1[0]), "+v"(v.r0[NFAM - 1]), "+v"(v.r1[NFAM - 1]), "+v"(v.y0), "+v"(v.y1));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v.r0[0]), "+v"(v.r1[0]), "+v"(v.y0), "+v"(v.y1));
    if constexpr (NFB == 2)
      asm volatile("" : "+v"(v.b[0][0]), "+v"(v.b[0][1]), "+v"(v.b[0][2]), "+v"(v.b[0][3]), "+v"(v.b[NFB - 1][0]), "+v"(v.b[NFB - 1][1]),
                   "+v"(v.b[NFB - 1][2]), "+v"(v.b[NFB - 1][3]));
    else
      asm volatile("" : "+v"(v.b[0][0]), "+v"(v.b[0][1]), "+v"(v.b[0][2]), "+v"(v.b[0][3]));
  }
  __device__ __forceinline__ void vstep(const VBuf &v)
  {
    double A[NFA];
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      A[a] = v.r0[a] * v.r1[a];
    });
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      accv[a][0] = pdh::mfma4(A[a], v.b0, accv[a][0]);
      accv[a][1] = pdh::mfma4(A[a], v.b1, accv[a][1]);
    });
    if constexpr (LAST_ROW)
      accyv = pdh::mfma4(v.b0, v.y0 * v.y1, accyv);
  }
  __device__ __forceinline__ void fstep(const FBuf &v)
  {
    double A[NFA];
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      A[a] = v.r0[a] * v.r1[a];
    });
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      static_for<0, NFB>([&](auto b_) {
        constexpr int b = b_;
        static_for<0, 4>([&](auto r_) {
          constexpr int r = r_;
          accf[a][b][r] = pdh::mfma4(A[a], v.b[b][r], accf[a][b][r]);
        });
      });
    });
    if constexpr (LAST_ROW)
      {
        const double Y = v.y0 * v.y1;
        static_for<0, NFB>([&](auto b_) {
          constexpr int b = b_;
          accyf[b] = pdh::mfma4(v.b[b][0], Y, accyf[b]);
        });
      }
  }
  // all VCH points of a chunk (dead points carry zero weights)
  __device__ __forceinline__ void volume_chunk_full()
  {
    constexpr int NS = VCH / 4;
    VBuf buf[2];
    vload<0>(buf[0]);
    static_for<0, NS>([&](auto s_) {
      constexpr int s = s_;
      vwait(buf[s & 1]);
      if constexpr (s + 1 < NS)
        vload<s + 1>(buf[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      vstep(buf[s & 1]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  __device__ __forceinline__ void face_chunk_full()
  {
    constexpr int NS = CH / 4;
    FBuf buf[2];
    fload<0>(buf[0]);
    static_for<0, NS>([&](auto s_) {
      constexpr int s = s_;
      fwait(buf[s & 1]);
      if constexpr (s + 1 < NS)
        fload<s + 1>(buf[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      fstep(buf[s & 1]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  static __device__ __forceinline__ double ld(const char *rb, int off) { return *reinterpret_cast<const double *>(rb + off); }
  // One step = 4 points.  The LDS factors of the next step are requested before the MFMAs of the current one.
  __device__ __forceinline__ void volume_chunk(const double *rec, int cnt, int lane)
  {
    const int kq = lane >> 4;
    const int nsteps = (cnt + 3) >> 2;
    double r0[NFA], r1[NFA], rb0, ry0 = 0.0, ry1 = 0.0;
    auto fetch = [&](int step) {
      const char *rb = reinterpret_cast<const char *>(rec + (4 * step + kq) * VREC);
      static_for<0, NFAM>([&](auto a_) {
        constexpr int a = a_;
        r0[a] = ld(rb, offA0v[a]);
        r1[a] = ld(rb, offA1v[a]);
      });
      if constexpr (LAST_ROW)
        {
          ry0 = ld(rb, offY0v);
          ry1 = ld(rb, offY0v + NA * 8);
        }
      rb0 = ld(rb, offBv);
    };
    fetch(0);
    for (int step = 0; step < nsteps; ++step)
      {
        double A[NFA];
        static_for<0, NFAM>([&](auto a_) {
          constexpr int a = a_;
          A[a] = r0[a] * r1[a];
        });
        const double Y = ry0 * ry1;
        const double B0 = rb0;
        fetch(step + 1 < nsteps ? step + 1 : step);
        const double B1 = rt.template rot<1>(B0);
        static_for<0, NFAM>([&](auto a_) {
          constexpr int a = a_;
          accv[a][0] = pdh::mfma4(A[a], B0, accv[a][0]);
          accv[a][1] = pdh::mfma4(A[a], B1, accv[a][1]);
        });
        if constexpr (LAST_ROW)
          accyv = pdh::mfma4(B0, Y, accyv);
      }
  }
  __device__ __forceinline__ void face_chunk(const double *rec, int cnt, int lane)
  {
    const int kq = lane >> 4;
    const int nsteps = (cnt + 3) >> 2;
    double r0[NFA], r1[NFA], rb[NFB], ry0 = 0.0, ry1 = 0.0;
    auto fetch = [&](int step) {
      const char *p = reinterpret_cast<const char *>(rec + (4 * step + kq) * REC);
      static_for<0, NFAM>([&](auto a_) {
        constexpr int a = a_;
        r0[a] = ld(p, offA0[a]);
        r1[a] = ld(p, offA1[a]);
      });
      if constexpr (LAST_ROW)
        {
          ry0 = ld(p, offY0);
          ry1 = ld(p, offY0 + NA * 8);
        }
      static_for<0, NFB>([&](auto b_) {
        constexpr int b = b_;
        rb[b] = ld(p, offBf[b]);
      });
    };
    fetch(0);
    for (int step = 0; step < nsteps; ++step)
      {
        double A[NFA], B[NFB][4];
        static_for<0, NFAM>([&](auto a_) {
          constexpr int a = a_;
          A[a] = r0[a] * r1[a];
        });
        const double Y = ry0 * ry1;
        static_for<0, NFB>([&](auto b_) {
          constexpr int b = b_;
          B[b][0] = rb[b];
        });
        fetch(step + 1 < nsteps ? step + 1 : step);
        static_for<0, NFB>([&](auto b_) {
          constexpr int b = b_;
          B[b][1] = rt.template rot<1>(B[b][0]);
          B[b][2] = rt.template rot<2>(B[b][0]);
          B[b][3] = rt.template rot<3>(B[b][0]);
        });
        static_for<0, NFAM>([&](auto a_) {
          constexpr int a = a_;
          static_for<0, NFB>([&](auto b_) {
            constexpr int b = b_;
            static_for<0, 4>([&](auto r_) {
              constexpr int r = r_;
              accf[a][b][r] = pdh::mfma4(A[a], B[b][r], accf[a][b][r]);
            });
          });
        });
        if constexpr (LAST_ROW)
          static_for<0, NFB>([&](auto b_) {
            constexpr int b = b_;
            accyf[b] = pdh::mfma4(B[b][0], Y, accyf[b]);
          });
      }
  }
  // Scatter the accumulators into Mx[tensor][row][NA] (VOL: tensor 0 = volume, 1 + t = face weight t; else t).  D layout:
  // lane = 16 i + 4 blk + j holds row 16 a + 4 blk + i; faces: a2 = 4 b + ((blk + r) & 3), t = j; volume: a2 =
  // 4 ((blk + r) & 1) + j.
  template <bool VOL>
  __device__ __forceinline__ void scatter(double *Mx, int lane) const
  {
    const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
    if constexpr (LAST_ROW)
      if (j == 0)
        {
          // transposed products: D lane (i, blk, 0) = sum_q B[row' = 4 blk + i](q) y(q); faces: row' = (a2 = 4 b + blk, t = i),
          // volume: row' = a2 = 4 (blk & 1) + i (replicated, blocks 0 and 1 are the live copies)
          if constexpr (VOL)
            if (blk < 2 && 4 * blk + i < NA)
              Mx[(ROWS - 1) * NA + 4 * blk + i] = accyv;
          static_for<0, NFB>([&](auto b_) {
            constexpr int b = b_;
            if (4 * b + blk < NA)
              Mx[(((VOL ? 1 : 0) + i) * ROWS + ROWS - 1) * NA + 4 * b + blk] = accyf[b];
          });
        }
    static_for<0, NFAM>([&](auto a_) {
      constexpr int a = a_;
      const int row = 16 * a + 4 * blk + i;
      if (row < ROWS)
        {
          if constexpr (VOL)
            static_for<0, 2>([&](auto r_) {
              constexpr int r = r_;
              const int a2 = 4 * ((blk + r) & 1) + j;
              if (a2 < NA)
                Mx[row * NA + a2] = accv[a][r];
            });
          static_for<0, NFB>([&](auto b_) {
            constexpr int b = b_;
            static_for<0, 4>([&](auto r_) {
              constexpr int r = r_;
              const int a2 = 4 * b + ((blk + r) & 3);
              if (a2 < NA)
                Mx[(((VOL ? 1 : 0) + j) * ROWS + row) * NA + a2] = accf[a][b][r];
            });
          });
        }
    });
  }
};


// ------------------------------------------------------------------------------------------------------------
// Contraction stages 2 and 3 on the f64 MFMA - FE_DGQ(3): N1D = 4, n = 64, function index = k0 + 4 k1 + 16 k2.
// Both stages are small GEMMs with K = 8 (seven Legendre modes + a zero pad):
//   stage 2   C2[(s1,o1)][(o2,a0)]    = sum_terms scale * sum_a1 Y[pair(s1,o1)][a1] T1[arr][o2][a0][a1]
//   stage 3   OUT[(s0,o0)][(s1,o1,o2)] = sum_X sum_a0 X[pair(s0,o0)][a0] T2_X[a0][(s1,o1,o2)]
// (s = index of the side the slab loop runs over - the rows -, o = index on the side of the columns).  A operands are the
// expansion tables; they do not depend on the slab and stay in registers, one copy per s with all four lane blocks
// holding the same table block (so that a result register is a complete row; no block rotations needed).  B operands are
// written to LDS by the producing stage in exactly the lane order the consumer reads (T1B, T2B: one conflict-free
// 512-byte read per fragment).  In the VALU form of these stages (kept below for the other elements) the LDS return
// bandwidth of the row reads and the FMA issue bounded the kernels; here a slab costs 112-176 MFMA.
// Result layout (D of stage 3, register [cf][s0], lane = 16 i + 4 blk + j):  row S = s0 + 4 cf + 16 s2, column
// O = i + 4 j + 16 blk  - a register is one complete row of the block, a store instruction writes its 512 bytes.
// ------------------------------------------------------------------------------------------------------------
struct ASet
{
  double a[4][2]; // [s][k-step]
};

// tab: [16 pairs][stride] in LDS, pair = 4 k + l;  SWAP = false: rows s are k, columns o are l
template <bool SWAP>
__device__ __forceinline__ void load_aset(const double *tab, int stride, int lane, ASet &A)
{
  const int kq = lane >> 4, idx = lane & 3;
  static_for<0, 4>([&](auto r_) {
    constexpr int r = r_;
    constexpr int sidx = r;
    const int pair = SWAP ? (idx * 4 + sidx) : (sidx * 4 + idx);
    A.a[r][0] = tab[pair * stride + kq];
    A.a[r][1] = tab[pair * stride + kq + 4]; // entry 7 is the zero pad
  });
}

// T1B / T2B hold the B operands of stages 2 / 3 in the order the MFMA lanes read them: 64 consecutive doubles per
// fragment.  The PRODUCING stage writes them scattered, and in the plain order its 16-lane write groups hit 2 (T1B: the
// bank depends on a0 only, 7-way conflicts) or 2 (T2B: 8-way) of the 16 double-wide banks: measured with SQ counters on the
// row kernel, two thirds of all its bank-conflict cycles came from these stores.  Both layouts are therefore rotated
// inside every aligned block of 16 doubles by an amount the reader can recompute from its lane:
//   T1B: position 16 (a1 & 3) + x  ->  16 (a1 & 3) + ((x + rot(a1)) & 15),  rot = {0,7,2,14,12,5,10,3}  (searched: <= 2-way)
//   T2B: position 16 j + x         ->  16 j + ((x + j + 8 ks) & 15)                                  (conflict-free)
// A reader's 32-lane group still covers 32 distinct doubles of one 256-byte row: conflict-free as before.
__device__ __forceinline__ int t1b_rot(int a1) { return (int)((0x3A5CE270u >> (4 * a1)) & 15u); }

// position of T1[arr][hi][a0][a1] in the B-operand order of stage 2
__device__ __forceinline__ int t1b_index(int arr, int hi, int a0, int a1)
{
  return ((arr * 2 + (hi >> 1)) * 2 + (a1 >> 2)) * 64 + (a1 & 3) * 16 + ((8 * (hi & 1) + a0 + t1b_rot(a1)) & 15);
}

// Offsets of this lane's stage-1 outputs inside one [hi] slice of T1B (everything that depends on (a0,a1) only, computed
// once per kernel): the value itself and the zero pads the lane owns (a1 = 6: column 7; a0 = 6: row 7; both: the corner).
struct T1Off
{
  int val[2], padc[2], padr[2], padx[2]; // [hi & 1]; pad offsets are -1 when the lane does not own that pad
  __device__ __forceinline__ void init(int a0, int a1)
  {
    for (int h = 0; h < 2; ++h)
      {
        val[h] = t1b_index(0, h, a0, a1);
        padc[h] = a1 == 6 ? t1b_index(0, h, a0, 7) : -1;
        padr[h] = a0 == 6 ? t1b_index(0, h, 7, a1) : -1;
        padx[h] = (a0 == 6 && a1 == 6) ? t1b_index(0, h, 7, 7) : -1;
      }
  }
};

// ARR, HI compile-time: T1B offset of the slice = ARR * 256 + (HI >> 1) * 128 (+ the lane's offset for HI & 1)
template <int ARR, int HI>
__device__ __forceinline__ void t1b_store(double *T1B, const T1Off &o, double v)
{
  constexpr int base = ARR * 256 + (HI >> 1) * 128, h = HI & 1;
  T1B[base + o.val[h]] = v;
  if (o.padc[h] >= 0)
    T1B[base + o.padc[h]] = 0.0;
  if (o.padr[h] >= 0)
    T1B[base + o.padr[h]] = 0.0;
  if (o.padx[h] >= 0)
    T1B[base + o.padx[h]] = 0.0;
}

// one term of stage 2: D2[bf][r] += (scale * A) . T1B[arr]      (ASets are passed by reference and indexed statically
// only: a pointer to one would move it to scratch memory)
__device__ __forceinline__ void mstage2_term(const ASet &A, int arr, double scale, const double *T1B, int lane, double (&D2)[2][4])
{
  static_for<0, 2>([&](auto ks_) {
    constexpr int ks = ks_;
    double ar[4];
    static_for<0, 4>([&](auto r_) {
      constexpr int r = r_;
      ar[r] = A.a[r][ks] * scale;
    });
    static_for<0, 2>([&](auto bf_) {
      constexpr int bf = bf_;
      const double b = T1B[((arr * 2 + bf) * 2 + ks) * 64 + ((lane & 48) | ((lane + t1b_rot(4 * ks + (lane >> 4))) & 15))];
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        D2[bf][r] = pdh::mfma4(ar[r], b, D2[bf][r]);
      });
    });
  });
}

// scatter the stage-2 result into T2B (B-operand order of stage 3):
// D2[bf][s1] lane (i, blk, j) = T2[a0 = 4 (blk & 1) + j][(s1, o1 = i, o2 = 2 bf + (blk >> 1))]
__device__ __forceinline__ void mstage2_scatter(const double (&D2)[2][4], double *T2B, int lane)
{
  const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
  static_for<0, 2>([&](auto bf_) {
    constexpr int bf = bf_;
    static_for<0, 4>([&](auto r_) {
      constexpr int r = r_;
      T2B[((blk & 1) * 4 + r) * 64 + j * 16 + (((2 * bf + (blk >> 1)) * 4 + i + j + 8 * (blk & 1)) & 15)] = D2[bf][r];
    });
  });
}

// stage 3 for one X-type: D3[cf][s0] += X . T2B
__device__ __forceinline__ void mstage3(const ASet &X, const double *T2B, int lane, double (&D3)[4][4])
{
  static_for<0, 2>([&](auto ks_) {
    constexpr int ks = ks_;
    static_for<0, 4>([&](auto cf_) {
      constexpr int cf = cf_;
      const double b = T2B[(ks * 4 + cf) * 64 + ((lane & 48) | ((lane + (lane >> 4) + 8 * ks) & 15))];
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        D3[cf][r] = pdh::mfma4(X.a[r][ks], b, D3[cf][r]);
      });
    });
  });
}

// Stage 3 with the operand ROLES SWAPPED (row kernel, pdh_rows.h): the T2 fragment is the A operand, the table the B operand.
//   D lane (i, blk, j) = sum_a0 T2[a0][(s1; o1 = blk, o2 = i)] X[s0][o0 = j][a0]   ->   column O = o0 + 4 o1 + 16 o2 = lane
// - a result register is still one complete row, and now in the NATURAL column order: lane l holds column l, so the
// epilogue stores it as it stands (no transposition through LDS).  The table registers are the same (an ASet holds
// X[s][idx][k] with idx = lane & 3 in all four lane blocks - A and B fragments are addressed alike); the T2 fragment wants
// (o1, o2) = (blk, idx) where the B-operand order had (idx, blk), so the producer scatters accordingly.  Rotation inside
// aligned blocks of 16 doubles: 2 kq + 8 ks (conflict-free for the writer's 16-lane groups: banks h + 2 j + 8 (blk & 1)).
__device__ __forceinline__ void mstage2_scatter_t(const double (&D2)[2][4], double *T2B, int lane)
{
  const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
  static_for<0, 2>([&](auto bf_) {
    constexpr int bf = bf_;
    static_for<0, 4>([&](auto r_) {
      constexpr int r = r_;
      // D2[bf][s1 = r] lane (i, blk, j) = T2[a0 = 4 (blk & 1) + j][(s1, o1 = i, o2 = 2 bf + (blk >> 1))]
      // reader lane (kq = j, blk_r = o1 = i, idx_r = o2) of fragment (ks = blk & 1, cf = s1)
      T2B[((blk & 1) * 4 + r) * 64 + j * 16 + ((4 * i + 2 * bf + (blk >> 1) + 2 * j + 8 * (blk & 1)) & 15)] = D2[bf][r];
    });
  });
}
__device__ __forceinline__ void mstage3_t(const ASet &X, const double *T2B, int lane, double (&D3)[4][4])
{
  static_for<0, 2>([&](auto ks_) {
    constexpr int ks = ks_;
    static_for<0, 4>([&](auto cf_) {
      constexpr int cf = cf_;
      const double a = T2B[(ks * 4 + cf) * 64 + ((lane & 48) | ((lane + 2 * (lane >> 4) + 8 * ks) & 15))];
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        D3[cf][r] = pdh::mfma4(a, X.a[r][ks], D3[cf][r]);
      });
    });
  });
}

template <int N1D>
constexpr int lds_doubles_diag()
{
  using M = MT<N1D>;
  using A = MomentAcc<N1D>;
  constexpr int tb = 3 * M::LTAB;
  constexpr int recs = (A::CH * A::REC > A::VCH * A::VREC) ? A::CH * A::REC : A::VCH * A::VREC;
  constexpr int mx = 5 * A::ROWS * M::NA;                                     // gathered moments
  constexpr int work = 4 * N1D * M::NA * M::NAP + N1D * N1D * N1D * M::RS; // T1 (4 arrays) + T2
  constexpr int m1 = recs > mx ? recs : mx;
  return tb + (m1 > work ? m1 : work);
}
template <int N1D>
constexpr int lds_doubles_offdiag()
{
  using M = MT<N1D>;
  using A = MomentAcc<N1D>;
  constexpr int tb = 6 * M::LTAB;
  constexpr int recs = A::CH * A::REC;
  constexpr int mx = 4 * A::ROWS * M::NA;
  constexpr int work = 3 * N1D * M::NA * M::NAP + N1D * N1D * N1D * M::RS; // T1 (3 arrays) + T2
  constexpr int stage = 64 * (N1D * N1D + 1);                  // transposed-store staging
  constexpr int m1 = recs > mx ? recs : mx;
  constexpr int m2 = work > stage ? work : stage;
  return tb + (m1 > m2 ? m1 : m2);
}

// ------------------------------------------------------------------------------------------------------------
// Diagonal blocks: one wave per owned polytope.
// ------------------------------------------------------------------------------------------------------------
template <int N1D, bool MFMA>
__global__ void __launch_bounds__(PDH_WAVE, 2) k_mdiag(const PdhDev P, const double *__restrict__ mt, const int n_owned)
{
  using M = MT<N1D>;
  constexpr int NA = M::NA, NAP = M::NAP, DIM = 3;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= n_owned)
    return;
  const int slot = pdh::xcd_chunked((int)blockIdx.x, n_owned); // (an XCD per contiguous eighth of the polytopes: see k_moffdiag)
  const int agg = P.own_agg[slot];
  double lo[DIM], ih[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      ih[c] = 1.0 / (P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - lo[c]);
    }
  double *tabE = lds, *tabD = lds + M::LTAB, *tabF = lds + 2 * M::LTAB;
  double *work = lds + 3 * M::LTAB;
  for (int t = lane; t < 3 * M::TAB; t += PDH_WAVE)
    lds[(t / NAP) * M::RS + t % NAP] = mt[t]; // rows re-strided from NAP to RS

  const bool act = lane < NA * NA;
  const int a0 = act ? lane / NA : 0, a1 = act ? lane % NA : 0;
  using Acc = MomentAcc<N1D>;
  Acc ma;
  ma.init(lane);
  ma.init_addr(work, lane);
  // Point data of a chunk is loaded one chunk ahead (registers) so that the global-load latency sits behind the MFMA
  // loop of the previous chunk: measured, the record phase of this kernel was 0.53 of its 2.14 ms, mostly waiting.
  // ---- volume moments -----------------------------------------------------------------------------------
  {
    const int64_t qb = P.vq_ptr[slot], qe = P.vq_ptr[slot + 1];
    double px[DIM] = {0.0, 0.0, 0.0}, pw = 0.0;
    auto fetch = [&](int64_t base) {
      const bool on = base + lane < qe;
      for (int c = 0; c < DIM; ++c)
        px[c] = on ? P.vq_x[c * P.vq_stride + base + lane] : 0.0;
      pw = on ? P.vq_w[base + lane] : 0.0;
    };
    fetch(qb);
#if PDHM_EXP == 2
    for (int64_t base = qb; base < qe && P.n < 0; base += Acc::VCH)
#else
    for (int64_t base = qb; base < qe; base += Acc::VCH)
#endif
      {
        const int cnt = (int)((qe - base < Acc::VCH) ? (qe - base) : Acc::VCH);
        PDH_WAVE_SYNC();
        {
          double xu[DIM] = {0.5, 0.5, 0.5}, w = 0.0;
          if (lane < cnt)
            {
              for (int c = 0; c < DIM; ++c)
                xu[c] = (px[c] - lo[c]) * ih[c];
              w = pw;
            }
          fetch(base + Acc::VCH);
#if PDHM_EXP != 5
          Acc::write_volume_record(work + lane * Acc::VREC, xu, w);
#endif
        }
        PDH_WAVE_SYNC();
#if PDHM_EXP != 6
        if (cnt == Acc::VCH)
          ma.volume_chunk_full();
        else
          ma.volume_chunk(work, cnt, lane);
#endif
      }
  }
  // ---- moments of the own-side face points (all faces of the polytope, boundary included) ------------------
  {
    const int64_t pb = P.ap_ptr[slot], pe = P.ap_ptr[slot + 1];
    const int half = lane >> 5, pt = lane & 31;
    // half 0 needs x0, x1; half 1 needs x2, w, sigma, n0, n1, n2
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0, f4 = 0.0, f5 = 0.0;
    auto fetch = [&](int64_t base) {
      const bool on = base + pt < pe;
      const int64_t q = on ? base + pt : pb;
      if (half == 0)
        {
          f0 = P.ap_x[0 * P.ap_stride + q];
          f1 = P.ap_x[1 * P.ap_stride + q];
        }
      else
        {
          f0 = P.ap_x[2 * P.ap_stride + q];
          f1 = on ? P.ap_wself[q] : 0.0;
          f2 = P.ap_sig[q];
          f3 = P.ap_n[0 * P.ap_stride + q];
          f4 = P.ap_n[1 * P.ap_stride + q];
          f5 = P.ap_n[2 * P.ap_stride + q];
        }
    };
    if (pb < pe)
      fetch(pb);
#if PDHM_EXP == 2
    for (int64_t base = pb; base < pe && P.n < 0; base += Acc::CH)
#else
    for (int64_t base = pb; base < pe; base += Acc::CH)
#endif
      {
        const int cnt = (int)((pe - base < Acc::CH) ? (pe - base) : Acc::CH);
        PDH_WAVE_SYNC();
        {
          double xa = 0.5, xb = 0.5, s[4] = {0.0, 0.0, 0.0, 0.0};
          if (pt < cnt)
            {
              if (half == 0)
                {
                  xa = (f0 - lo[0]) * ih[0];
                  xb = (f1 - lo[1]) * ih[1];
                }
              else
                {
                  xa = (f0 - lo[2]) * ih[2];
                  const double w = f1;
                  // U_i phi_j + phi_i U_j  with  U = -1/2 grad phi . n + (sig/2) phi   (pdh_kernels.h, k_diag)
                  s[0] = w * f2;
                  s[1] = -0.5 * w * f3;
                  s[2] = -0.5 * w * f4;
                  s[3] = -0.5 * w * f5;
                }
            }
          if (base + Acc::CH < pe)
            fetch(base + Acc::CH);
#if PDHM_EXP != 5
          Acc::write_face_record_half(work + pt * Acc::REC, half, xa, xb, s);
#endif
        }
        PDH_WAVE_SYNC();
#if PDHM_EXP != 6
        if (cnt == Acc::CH)
          ma.face_chunk_full();
        else
          ma.face_chunk(work, cnt, lane);
#endif
      }
  }
  // gather: this lane's (a0,a1) row of every moment tensor
  double accM[NAP], accS[NAP], accN[DIM][NAP];
  PDH_WAVE_SYNC();
  ma.template scatter<true>(work, lane);
  PDH_WAVE_SYNC();
  for (int a = 0; a < NA; ++a)
    {
      const int row = act ? lane : 0;
      accM[a] = work[(0 * Acc::ROWS + row) * NA + a];
      accS[a] = work[(1 * Acc::ROWS + row) * NA + a];
      for (int c = 0; c < DIM; ++c)
        accN[c][a] = work[((2 + c) * Acc::ROWS + row) * NA + a];
    }
  // reaction term  c phi_i phi_j  shares the (E,E,E) contraction with the penalty moments
  if (P.reaction_c != 0.0)
    for (int a = 0; a < NA; ++a)
      accS[a] += P.reaction_c * accM[a];

  // ---- FE_DGQ(3): contraction stages 2 and 3 on the MFMA (MFMA = true is launched for n = 64 only) ------------------
  if constexpr (MFMA)
      {
        static_assert(N1D == 4, "the MFMA contraction is written for FE_DGQ(3)");
        double *T1B = work;               // [4 arrays][2 bf][2 ks][64]
        double *T2B = work + 4 * 2 * 2 * 64; // [2 ks][4 cf][64]
        const int64_t rbase = P.row_base[slot];
        const int rlen = P.row_len[slot];
        const int L = P.diag_L[slot];
        const double ih0 = ih[0], ih1 = ih[1], ih2 = ih[2];
        const int di = lane >> 4, dblk = (lane >> 2) & 3, dj = lane & 3;
        const int O = di + 4 * dj + 16 * dblk; // this lane's column
        T1Off t1o;
        t1o.init(a0, a1);
        // zero pads of the K = 8 fragments (row 7 / column 7 of every [a0][a1] plane): T1B is cleared once, stage 1 then writes
        // its 49 values only - nothing else touches T1B inside the slab loop of this kernel (see pdh_rows.h, P4)
        PDH_WAVE_SYNC();
        for (int k = 0; k < 16; ++k)
          T1B[k * 64 + lane] = 0.0;
#pragma unroll 1
        for (int k2 = 0; k2 < PDHM_SLABS; ++k2)
          {
            PDH_WAVE_SYNC();
            if (act)
              static_for<0, 4>([&](auto ll_) {
                constexpr int ll = ll_;
                const int pr = (k2 * 4 + ll) * M::RS;
                double g1 = 0.0, ee = 0.0, n0 = 0.0, n1 = 0.0;
                static_for<0, NA>([&](auto a_) {
                  constexpr int a = a_;
                  const double e = tabE[pr + a];
                  g1 += e * accM[a];
                  ee += (tabD[pr + a] * (ih2 * ih2)) * accM[a] + e * accS[a] + (tabF[pr + a] * ih2) * accN[2][a];
                  n0 += e * accN[0][a];
                  n1 += e * accN[1][a];
                });
                T1B[(0 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = g1;
                T1B[(1 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = ee;
                T1B[(2 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = n0;
                T1B[(3 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = n1;
              });
            // the A operands (24 doubles) are re-read per slab instead of living through stage 1 (register pressure);
            // the opaque zero keeps the compiler from hoisting the loads out of the loop again
            int zero = 0;
            asm volatile("" : "+s"(zero));
            ASet AE, AD, AF;
            load_aset<false>(tabE + zero, M::RS, lane, AE);
            load_aset<false>(tabD + zero, M::RS, lane, AD);
            load_aset<false>(tabF + zero, M::RS, lane, AF);
            double D3[4][4];
            for (int c = 0; c < 4; ++c)
              for (int r = 0; r < 4; ++r)
                D3[c][r] = 0.0;
            {
              PDH_WAVE_SYNC();
              double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = D
              mstage2_term(AE, 0, ih0 * ih0, T1B, lane, D2);
              mstage2_scatter(D2, T2B, lane);
              PDH_WAVE_SYNC();
              mstage3(AD, T2B, lane, D3);
            }
            {
              PDH_WAVE_SYNC();
              double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = E
              mstage2_term(AD, 0, ih1 * ih1, T1B, lane, D2);
              mstage2_term(AE, 1, 1.0, T1B, lane, D2);
              mstage2_term(AF, 3, ih1, T1B, lane, D2);
              mstage2_scatter(D2, T2B, lane);
              PDH_WAVE_SYNC();
              mstage3(AE, T2B, lane, D3);
            }
            {
              PDH_WAVE_SYNC();
              double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = Fs
              mstage2_term(AE, 2, ih0, T1B, lane, D2);
              mstage2_scatter(D2, T2B, lane);
              PDH_WAVE_SYNC();
              mstage3(AF, T2B, lane, D3);
            }
            static_for<0, 4>([&](auto cf_) {
              constexpr int cf = cf_;
              static_for<0, 4>([&](auto s0_) {
                constexpr int s0 = s0_;
                const int R = s0 + 4 * cf + 16 * k2;
                int pos = L + O;
                if (P.diag_first)
                  pos = (O == R) ? 0 : (L + O + (O < R ? 1 : 0));
                if (PDHM_STORE_OK)
                  P.values[rbase + (int64_t)R * rlen + pos] = D3[cf][s0];
              });
            });
          }
      }
  else
  {
  // ---- contraction, one slab of rows (fixed k2) at a time ------------------------------------------------
  double *T1 = work;                          // [4][N1D][NA][NAP]: 0 = E.M, 1 = (D.M/h2^2 + E.S + Fs.N2/h2), 2 = E.N0, 3 = E.N1
  double *T2 = work + 4 * N1D * NA * NAP;      // [N1D (s1)][N1D^2 (o2,o1)][RS]
  const int n = P.n;
  const bool live = lane < n;
  int l0 = 0, l1 = 0, l2 = 0, packed_own = 0;
  if (live)
    {
      packed_own = P.midx[lane];
      l0 = packed_own & 0xff;
      l1 = (packed_own >> 8) & 0xff;
      l2 = (packed_own >> 16) & 0xff;
    }
  const int64_t rbase = P.row_base[slot];
  const int rlen = P.row_len[slot];
  const int L = P.diag_L[slot];
  const double ih0 = ih[0], ih1 = ih[1], ih2 = ih[2];
  int row_begin = 0;
#pragma unroll 1
  for (int k2 = 0; k2 < N1D && row_begin < n; ++k2)
    {
      const int row_end = slab_end(live, l2, k2);
      // The X / Y table entries of this lane do not depend on k2: hoisted out of this loop by the compiler they would
      // occupy 192 VGPRs for its whole duration (measured: 240 spills).  An opaque zero ties their address to the iteration.
      int zero = 0;
      asm volatile("" : "+s"(zero));
      tabE = lds + zero;
      tabD = lds + M::LTAB + zero;
      tabF = lds + 2 * M::LTAB + zero;
      PDH_WAVE_SYNC();
      // stage 1 (in registers: this lane's (a0,a1), contraction over a2)
      if (act)
        for (int ll = 0; ll < N1D; ++ll)
          {
            const int pr = (k2 * N1D + ll) * M::RS;
            double g1 = 0.0, ee = 0.0, n0 = 0.0, n1 = 0.0;
            static_for<0, NA>([&](auto a_) {
              constexpr int a = a_;
              const double e = tabE[pr + a];
              g1 += e * accM[a];
              ee += (tabD[pr + a] * (ih2 * ih2)) * accM[a] + e * accS[a] + (tabF[pr + a] * ih2) * accN[2][a];
              n0 += e * accN[0][a];
              n1 += e * accN[1][a];
            });
            const int o = (ll * NA + a0) * NAP + a1;
            T1[0 * N1D * NA * NAP + o] = g1;
            T1[1 * N1D * NA * NAP + o] = ee;
            T1[2 * N1D * NA * NAP + o] = n0;
            T1[3 * N1D * NA * NAP + o] = n1;
            if (a1 == 0)
              for (int g = 0; g < 4; ++g)
                T1[g * N1D * NA * NAP + (ll * NA + a0) * NAP + NA] = 0.0;
          }
      double out[N1D * N1D];
      for (int r = 0; r < N1D * N1D; ++r)
        out[r] = 0.0;
      // X = D :  D(dir 0) E(dir 1) on E.M / h0^2
      {
        PDH_WAVE_SYNC();
        const Term<N1D> terms[1] = {{tabE, 0, ih0 * ih0}};
        stage2<N1D, 1, false>(terms, T1, T2, lane);
        PDH_WAVE_SYNC();
        stage3<N1D, false>(tabD, T2, l0, l1, l2, out);
      }
      // X = E :  D(dir 1) on E.M / h1^2  +  E(dir 1) on the merged array  +  Fs(dir 1) on E.N1 / h1
      {
        PDH_WAVE_SYNC();
        const Term<N1D> terms[3] = {{tabD, 0, ih1 * ih1}, {tabE, 1, 1.0}, {tabF, 3, ih1}};
        stage2<N1D, 3, false>(terms, T1, T2, lane);
        PDH_WAVE_SYNC();
        stage3<N1D, false>(tabE, T2, l0, l1, l2, out);
      }
      // X = Fs :  E(dir 1) on E.N0 / h0
      {
        PDH_WAVE_SYNC();
        const Term<N1D> terms[1] = {{tabE, 2, ih0}};
        stage2<N1D, 1, false>(terms, T1, T2, lane);
        PDH_WAVE_SYNC();
        stage3<N1D, false>(tabF, T2, l0, l1, l2, out);
      }
      // rows of the slab: one contiguous segment per row at its CSR position (diagonal-first shift as in store_strip)
      static_for<0, N1D * N1D>([&](auto r_) {
        constexpr int r = r_;
        const int R = row_of(live, packed_own, r % N1D, r / N1D, k2);
        if (R >= 0 && live)
          {
            int pos = L + lane;
            if (P.diag_first)
              pos = (lane == R) ? 0 : (L + lane + (lane < R ? 1 : 0));
            P.values[rbase + (int64_t)R * rlen + pos] = out[r];
          }
      });
      row_begin = row_end;
    }
  } // VALU contraction
}

// ------------------------------------------------------------------------------------------------------------
// Coupling blocks: one wave per interior face item (P = side whose packed points are used, Q = neighbour).
//   A[P,Q]_ij = sum_q w [ (1/2 g^P_i - sigma phi^P_i) phi^Q_j - 1/2 phi^P_i g^Q_j ],   g = grad phi . n_P
// ------------------------------------------------------------------------------------------------------------
template <int N1D, bool MFMA>
__global__ void __launch_bounds__(PDH_WAVE, 2) k_moffdiag(const PdhDev P, const double *__restrict__ mt, const int n_items)
{
  using M = MT<N1D>;
  constexpr int NA = M::NA, NAP = M::NAP, NG = M::NG, DIM = 3;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  // an XCD per contiguous eighth of the items: in deal.II's diagonal-first rows the blocks left of the diagonal are shifted by one entry,
  // so neighbouring blocks of a row share 128-byte lines; written from one L2 they merge there instead of reaching HBM as partial lines
  const int item = (int)blockIdx.x < n_items ? pdh::xcd_chunked((int)blockIdx.x, n_items) : (int)blockIdx.x;
  if (item >= n_items)
    return;
  const int slot = P.it_own[item];
  const int agg = P.own_agg[slot];
  const int nbr = P.it_nbr[item];
  double lo[DIM], ih[DIM], loq[DIM], ihq[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      ih[c] = 1.0 / (P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - lo[c]);
      loq[c] = P.bbox[(int64_t)nbr * 2 * DIM + c];
      ihq[c] = 1.0 / (P.bbox[(int64_t)nbr * 2 * DIM + DIM + c] - loq[c]);
    }
  // The moments of a face are taken in a frame F chosen PER DIRECTION as the shorter of the two bounding-box intervals
  // (the face points lie in both boxes, so x^_F stays in [0,1]).  Both bases are then polynomials of the frame
  // coordinate with a slope <= 1:  xi^P = aP t + bP,  xi^Q = aQ t + bQ.  Taking P's own frame instead loses digits fast
  // when Q is smaller than P (B^Q(alpha t + beta) with alpha = h^P/h^Q > 1 has coefficients ~ alpha^p that cancel in the
  // expansion): measured 2e-12 / 1e-9 / 3e-7 relative for size ratios 2 / 4 / 8, tools/debug/ratio_check.py.
  //   EQ_c[k,l,a] = <B_k(xi^P) B_l(xi^Q), L_a>,    HQ_c = <B'_k B_l>/h^P_c - <B_k B'_l>/h^Q_c
  double loF[DIM], ihF[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      const bool use_q = ihq[c] > ih[c]; // Q's interval is the shorter one
      loF[c] = use_q ? loq[c] : lo[c];
      ihF[c] = use_q ? ihq[c] : ih[c];
    }
  double *tabEQ = lds;                // [3][PAIRS][RS]
  double *tabHQ = lds + 3 * M::LTAB;  // [3][PAIRS][RS]
  double *work = lds + 6 * M::LTAB;
#if PDHM_EXP == 1
  if (lane < DIM * N1D * N1D && P.n < 0)
#else
  if (lane < DIM * N1D * N1D)
#endif
    {
      const int c = lane / (N1D * N1D), k = (lane / N1D) % N1D, l = lane % N1D;
      // this lane's direction: selected with exact 0/1 factors (an if-chain over the arrays is turned into an indexed
      // scratch-memory access by the compiler: a store -> load round trip through global memory at the head of every item)
      const double m1 = c == 1 ? 1.0 : 0.0, m2 = c == 2 ? 1.0 : 0.0, m0 = 1.0 - m1 - m2;
      const double lo_c = m0 * lo[0] + m1 * lo[1] + m2 * lo[2], ih_c = m0 * ih[0] + m1 * ih[1] + m2 * ih[2];
      const double loq_c = m0 * loq[0] + m1 * loq[1] + m2 * loq[2], ihq_c = m0 * ihq[0] + m1 * ihq[1] + m2 * ihq[2];
      const double loF_c = m0 * loF[0] + m1 * loF[1] + m2 * loF[2], ihF_c = m0 * ihF[0] + m1 * ihF[1] + m2 * ihF[2];
      const double hF = 1.0 / ihF_c;
      const double aP = hF * ih_c, bP = (loF_c - lo_c) * ih_c - 0.5;    // centred unit coordinate of P at frame coordinate t
      const double aQ = hF * ihq_c, bQ = (loF_c - loq_c) * ihq_c - 0.5; // ... of Q
      double e[NA], f[NA], g[NA];
      for (int a = 0; a < NA; ++a)
        e[a] = f[a] = g[a] = 0.0;
      constexpr int p = N1D - 1;
      for (int gq = 0; gq < NG; ++gq)
        {
          const double t = mt[M::OFF_GX + gq];
          const double xp = aP * t + bP, xq = aQ * t + bQ;
          double vk = P.tab.coef[k][p], dk = 0.0, vl = P.tab.coef[l][p], dl = 0.0;
          for (int m = p - 1; m >= 0; --m)
            {
              dk = dk * xp + vk;
              vk = vk * xp + P.tab.coef[k][m];
              dl = dl * xq + vl;
              vl = vl * xq + P.tab.coef[l][m];
            }
          const double vv = vk * vl, dv = dk * vl, vd = vk * dl;
          for (int a = 0; a < NA; ++a)
            {
              const double gl = mt[M::OFF_GL + a * NG + gq];
              e[a] += vv * gl;
              f[a] += dv * gl;
              g[a] += vd * gl;
            }
        }
      double *te = tabEQ + c * M::LTAB + (k * N1D + l) * M::RS, *th = tabHQ + c * M::LTAB + (k * N1D + l) * M::RS;
      for (int a = 0; a < NA; ++a)
        {
          te[a] = e[a];
          th[a] = f[a] * ih_c - g[a] * ihq_c;
        }
      te[NA] = 0.0;
      th[NA] = 0.0;
    }

  const bool act = lane < NA * NA;
  const int a0 = act ? lane / NA : 0, a1 = act ? lane % NA : 0;
  using Acc = MomentAcc<N1D>;
  Acc ma;
  ma.init(lane);
  ma.init_addr(work, lane);
  {
    const int64_t pb = P.it_pbeg[item], pe = pb + P.it_pcnt[item];
    const int half = lane >> 5, pt = lane & 31;
    // point data one chunk ahead (half 0: x0, x1; half 1: x2, w, sigma, n0, n1, n2), as in k_mdiag
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0, f4 = 0.0, f5 = 0.0;
    auto fetch = [&](int64_t base) {
      const bool on = base + pt < pe;
      const int64_t q = on ? base + pt : pb;
      if (half == 0)
        {
          f0 = P.ap_x[0 * P.ap_stride + q];
          f1 = P.ap_x[1 * P.ap_stride + q];
        }
      else
        {
          f0 = P.ap_x[2 * P.ap_stride + q];
          f1 = on ? P.ap_wcross[q] : 0.0;
          f2 = P.ap_sig[q];
          f3 = P.ap_n[0 * P.ap_stride + q];
          f4 = P.ap_n[1 * P.ap_stride + q];
          f5 = P.ap_n[2 * P.ap_stride + q];
        }
    };
    if (pb < pe)
      fetch(pb);
#if PDHM_EXP == 2
    for (int64_t base = pb; base < pe && P.n < 0; base += Acc::CH)
#else
    for (int64_t base = pb; base < pe; base += Acc::CH)
#endif
      {
        const int cnt = (int)((pe - base < Acc::CH) ? (pe - base) : Acc::CH);
        PDH_WAVE_SYNC();
        {
          double xa = 0.5, xb = 0.5, s[4] = {0.0, 0.0, 0.0, 0.0};
          if (pt < cnt)
            {
              if (half == 0)
                {
                  xa = (f0 - loF[0]) * ihF[0];
                  xb = (f1 - loF[1]) * ihF[1];
                }
              else
                {
                  xa = (f0 - loF[2]) * ihF[2];
                  const double w = f1;
                  s[0] = -w * f2;
                  s[1] = 0.5 * w * f3;
                  s[2] = 0.5 * w * f4;
                  s[3] = 0.5 * w * f5;
                }
            }
          if (base + Acc::CH < pe)
            fetch(base + Acc::CH);
          Acc::write_face_record_half(work + pt * Acc::REC, half, xa, xb, s);
        }
        PDH_WAVE_SYNC();
#if PDHM_OFF_FULL
        if (cnt == Acc::CH)
          ma.face_chunk_full();
        else
#endif
          ma.face_chunk(work, cnt, lane);
      }
  }
  double accS[NAP], accN[DIM][NAP];
  PDH_WAVE_SYNC();
  ma.template scatter<false>(work, lane);
  PDH_WAVE_SYNC();
  for (int a = 0; a < NA; ++a)
    {
      const int row = act ? lane : 0;
      accS[a] = work[(0 * Acc::ROWS + row) * NA + a];
      for (int c = 0; c < DIM; ++c)
        accN[c][a] = work[((1 + c) * Acc::ROWS + row) * NA + a];
    }

  // ---- FE_DGQ(3): contraction stages 2 and 3 on the MFMA -------------------------------------------------------------
  // Swapped orientation: the slabs run over Q's functions l (rows S of the result), this lane's column O is a function k
  // of P.  The result rows are therefore rows of A[Q,P] - the block left of the diagonal, shifted by one double in
  // deal.II's layout - and are written complete (512 B per instruction); A[P,Q] itself (aligned) goes through an LDS
  // staging tile as full 128-byte lines.
  if constexpr (MFMA)
      {
        static_assert(N1D == 4, "the MFMA contraction is written for FE_DGQ(3)");
        ASet AE0, AH0, AE1, AH1;
        PDH_WAVE_SYNC(); // per-face tables complete
        load_aset<true>(tabEQ, M::RS, lane, AE0);
        load_aset<true>(tabHQ, M::RS, lane, AH0);
        load_aset<true>(tabEQ + M::LTAB, M::RS, lane, AE1);
        load_aset<true>(tabHQ + M::LTAB, M::RS, lane, AH1);
        double *T1B = work;                  // [3 arrays][2 bf][2 ks][64]
        double *T2B = work + 3 * 2 * 2 * 64; // [2 ks][4 cf][64]
        const int64_t rbase = P.row_base[slot];
        const int rlen = P.row_len[slot];
        const int pos0 = P.it_pos[item];
        const int qslot = P.it_nbr_slot[item];
        const int64_t qbase = qslot >= 0 ? P.row_base[qslot] : 0;
        const int qlen = qslot >= 0 ? P.row_len[qslot] : 0;
        const int post = P.it_pos_t[item];
        const int di = lane >> 4, dblk = (lane >> 2) & 3, dj = lane & 3;
        const int O = di + 4 * dj + 16 * dblk; // this lane's function of P
        T1Off t1o;
        t1o.init(a0, a1);
#pragma unroll 1
        for (int s2 = 0; s2 < PDHM_SLABS; ++s2)
          {
            PDH_WAVE_SYNC();
            if (act)
              static_for<0, 4>([&](auto kk_) {
                constexpr int kk = kk_;
                const int pr = 2 * M::LTAB + (kk * 4 + s2) * M::RS; // direction 2 tables, pair (k2 = kk, l2 = s2)
                double ee = 0.0, n0 = 0.0, n1 = 0.0;
                static_for<0, NA>([&](auto a_) {
                  constexpr int a = a_;
                  const double e = tabEQ[pr + a];
                  ee += e * accS[a] + tabHQ[pr + a] * accN[2][a];
                  n0 += e * accN[0][a];
                  n1 += e * accN[1][a];
                });
                t1b_store<0, kk>(T1B, t1o, ee);
                t1b_store<1, kk>(T1B, t1o, n0);
                t1b_store<2, kk>(T1B, t1o, n1);
              });
            double D3[4][4];
            for (int c = 0; c < 4; ++c)
              for (int r = 0; r < 4; ++r)
                D3[c][r] = 0.0;
            {
              PDH_WAVE_SYNC();
              double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = EQ0
              mstage2_term(AE1, 0, 1.0, T1B, lane, D2);
              mstage2_term(AH1, 2, 1.0, T1B, lane, D2);
              mstage2_scatter(D2, T2B, lane);
              PDH_WAVE_SYNC();
              mstage3(AE0, T2B, lane, D3);
            }
            {
              PDH_WAVE_SYNC();
              double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = HQ0
              mstage2_term(AE1, 1, 1.0, T1B, lane, D2);
              mstage2_scatter(D2, T2B, lane);
              PDH_WAVE_SYNC();
              mstage3(AH0, T2B, lane, D3);
            }
            // A[Q,P] = A[P,Q]^T: rows 16 s2 .. 16 s2 + 15 of Q's block, complete
            if (qslot >= 0)
              static_for<0, 4>([&](auto cf_) {
                constexpr int cf = cf_;
                static_for<0, 4>([&](auto s0_) {
                  constexpr int s0 = s0_;
                  const int R = s0 + 4 * cf + 16 * s2;
                  if (PDHM_STORE_OK)
                    P.values[qbase + (int64_t)R * qlen + post + O] = D3[cf][s0];
                });
              });
            // A[P,Q]: columns 16 s2 .. 16 s2 + 15 of every row O, through an LDS staging tile (full 128-byte lines)
            {
              constexpr int SR = 17;
              PDH_WAVE_SYNC();
              double *stage = work;
              static_for<0, 4>([&](auto cf_) {
                constexpr int cf = cf_;
                static_for<0, 4>([&](auto r_) {
                  constexpr int r = r_;
                  stage[O * SR + r + 4 * cf] = D3[cf][r];
                });
              });
              PDH_WAVE_SYNC();
              const int cc = lane & 15;
              for (int jb = 0; jb < 64; jb += 4)
                {
                  const int j = jb + (lane >> 4);
                  if (PDHM_STORE_OK)
                    P.values[rbase + (int64_t)j * rlen + pos0 + 16 * s2 + cc] = stage[j * SR + cc];
                }
            }
          }
      }
  else
  {
  // Contraction in the SWAPPED orientation: this lane owns ROW i of A[P,Q] (a function of P), slabs run over the k2-type
  // index l2 of Q's functions.  A slab therefore yields complete rows j of the transposed block A[Q,P] - the block that
  // sits left of the diagonal in Q's rows and is shifted by one double in deal.II's diagonal-first layout - as coalesced
  // 512 B row stores, while the piece of A[P,Q] itself (right of the diagonal, aligned) goes through an LDS staging tile
  // and is written as complete 128 B lines.
  double *T1 = work;                      // [3][N1D (k2)][NA][NAP]: 0 = EQ2.S + HQ2.N2, 1 = EQ2.N0, 2 = EQ2.N1
  double *T2 = work + 3 * N1D * NA * NAP;  // [N1D (s1)][N1D^2 (o2,o1)][RS]
  const int n = P.n;
  const bool live = lane < n;
  int o0 = 0, o1 = 0, o2 = 0, packed_own = 0;
  if (live)
    {
      packed_own = P.midx[lane];
      o0 = packed_own & 0xff;
      o1 = (packed_own >> 8) & 0xff;
      o2 = (packed_own >> 16) & 0xff;
    }
  const int64_t rbase = P.row_base[slot];
  const int rlen = P.row_len[slot];
  const int pos0 = P.it_pos[item];
  const int qslot = P.it_nbr_slot[item];
  const int64_t qbase = qslot >= 0 ? P.row_base[qslot] : 0;
  const int qlen = qslot >= 0 ? P.row_len[qslot] : 0;
  const int post = P.it_pos_t[item];
  constexpr int SROW = N1D * N1D + 1; // staging row stride (odd: conflict-free)
  int col_begin = 0;
#if PDHM_EXP == 3
  if (P.n < 0)
#endif
#pragma unroll 1
  for (int s2 = 0; s2 < N1D && col_begin < n; ++s2)
    {
      const int col_end = slab_end(live, o2, s2);
      int zero = 0; // see k_mdiag: keeps the loop-invariant table loads inside the loop
      asm volatile("" : "+s"(zero));
      tabEQ = lds + zero;
      tabHQ = lds + 3 * M::LTAB + zero;
      PDH_WAVE_SYNC();
      if (act)
        for (int kk = 0; kk < N1D; ++kk)
          {
            const int pr = 2 * M::LTAB + (kk * N1D + s2) * M::RS; // direction 2 tables, pair (k2 = kk, l2 = s2)
            double ee = 0.0, n0 = 0.0, n1 = 0.0;
            static_for<0, NA>([&](auto a_) {
              constexpr int a = a_;
              const double e = tabEQ[pr + a];
              ee += e * accS[a] + tabHQ[pr + a] * accN[2][a];
              n0 += e * accN[0][a];
              n1 += e * accN[1][a];
            });
            const int o = (kk * NA + a0) * NAP + a1;
            T1[0 * N1D * NA * NAP + o] = ee;
            T1[1 * N1D * NA * NAP + o] = n0;
            T1[2 * N1D * NA * NAP + o] = n1;
            if (a1 == 0)
              for (int g = 0; g < 3; ++g)
                T1[g * N1D * NA * NAP + (kk * NA + a0) * NAP + NA] = 0.0;
          }
      double out[N1D * N1D]; // out[r], r = l1 * N1D + l0: A[P,Q][i = lane][j = (l0, l1, s2)]
      for (int r = 0; r < N1D * N1D; ++r)
        out[r] = 0.0;
      // X = EQ0 :  EQ1 on array 0  +  HQ1 on EQ2.N1
      {
        PDH_WAVE_SYNC();
        const Term<N1D> terms[2] = {{tabEQ + M::LTAB, 0, 1.0}, {tabHQ + M::LTAB, 2, 1.0}};
        stage2<N1D, 2, true>(terms, T1, T2, lane);
        PDH_WAVE_SYNC();
        stage3<N1D, true>(tabEQ, T2, o0, o1, o2, out);
      }
      // X = HQ0 :  EQ1 on EQ2.N0
      {
        PDH_WAVE_SYNC();
        const Term<N1D> terms[1] = {{tabEQ + M::LTAB, 1, 1.0}};
        stage2<N1D, 1, true>(terms, T1, T2, lane);
        PDH_WAVE_SYNC();
        stage3<N1D, true>(tabHQ, T2, o0, o1, o2, out);
      }
      // A[Q,P] = A[P,Q]^T: rows j of the slab, one contiguous segment per row (columns = this wave's lanes)
      int cols_of[N1D * N1D];
      static_for<0, N1D * N1D>([&](auto r_) {
        constexpr int r = r_;
        cols_of[r] = row_of(live, packed_own, r % N1D, r / N1D, s2);
        if (qslot >= 0 && cols_of[r] >= 0 && live && PDHM_STORE_OK)
          P.values[qbase + (int64_t)cols_of[r] * qlen + post + lane] = out[r];
      });
      // A[P,Q]: this slab holds columns [col_begin,col_end) of every row i.  Through an LDS staging tile so that one
      // store instruction writes 16-column (128 B) pieces of four rows.
      {
        PDH_WAVE_SYNC();
        double *stage = work;
        if (live)
          static_for<0, N1D * N1D>([&](auto r_) {
            constexpr int r = r_;
            if (cols_of[r] >= 0)
              stage[lane * SROW + cols_of[r] - col_begin] = out[r];
          });
        PDH_WAVE_SYNC();
        const int cols = col_end - col_begin;
        constexpr int W = N1D * N1D; // lanes per row piece (<= 16)
        const int cc = lane % W;
        for (int ib = 0; ib < n; ib += PDH_WAVE / W)
          {
            const int i = ib + lane / W;
            if (i < n && cc < cols && lane < (PDH_WAVE / W) * W && PDHM_STORE_OK)
              P.values[rbase + (int64_t)i * rlen + pos0 + col_begin + cc] = stage[i * SROW + cc];
          }
      }
      col_begin = col_end;
    }
  } // VALU contraction
}
} // namespace pdhm
