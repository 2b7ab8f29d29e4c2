// pdh_rows.h — the row kernel: 3-D, degree 1 .. 3, on polytopes whose faces are axis-aligned planes (agglomerates of
// Cartesian cells: every configuration of BASELINE.json but the piston mesh).  Described here for FE_DGQ(3), n = 64, the
// element it was designed around; the other elements run as "streamed" kinds of the same template (RowsKind below: shared
// moment phases, their own diagonal-block and store phases).  ONE wave per owned polytope computes and
// writes ALL values of the polytope's 64 rows - the diagonal block and every coupling block - as complete, aligned
// 512-byte pieces.  It replaces the pair k_mdiag / k_moffdiag (pdh_moment.h) where it applies; what is computed is
// unchanged (reference include/poly_utils.h:2040-2084 volume + boundary, :1870-1926 interface blocks), only the order
// of summation differs, so results agree with the other forms to rounding (tests/test_gpu_parity.py runs all three
// against the oracle).
//
// Why a separate kernel.  On a face that lies in the plane x_c = const with normal +-e_c
//   * the face moments of the diagonal block (pdh_moment.h) are rank one in direction c:
//       M[a0,a1,a2] = L_{a_c}(zeta) (x) M2[a_i,a_j]   - 49 x 2 numbers from a 2-D moment GEMM (2 MFMA per 4 points
//       instead of 26), expanded into the 3-D tensors with 14 FMAs per lane;
//   * a coupling block is a Kronecker product:  A[P,Q]_(k,l) = C[k_c,l_c] * S[(k_i,k_j),(l_i,l_j)]  with ONE 4x4 matrix
//       C = (1/2 s B'_k(zeta^P)/h^P - sigma B_k(zeta^P)) B_l(zeta^Q) - 1/2 s B_k(zeta^P) B'_l(zeta^Q)/h^Q
//     and ONE 16x16 matrix S = sum_q w B^P_ki B^P_kj B^Q_li B^Q_lj, obtained from 49 2-D moments by two small
//     contractions - a few hundred FMAs per face instead of the three-stage contraction of 4 x 343 moments.
//   The arithmetic of a coupling block shrinks so much that a polytope can afford to compute the blocks of BOTH sides of
//   each of its faces itself ("owner computes rows" taken literally).  Then one wave holds every value of a row and
//   the stores can be laid out for the memory system instead of for the arithmetic: in deal.II's diagonal-first rows
//   the blocks left of the diagonal start one double late, so a block-wise writer leaves two partial 128-byte lines per
//   block and row (2.7 TB/s measured against 6.1 TB/s for whole lines, profiles/r01_probe_store_pattern.txt).  Here every
//   store instruction writes positions 64 m .. 64 m + 63 of a row: lane 0 carries the last column of the block before
//   (or the diagonal entry), lanes 1..63 the first 63 columns of block m.
//
// Phases (one persistent wave per resident slot, LDS region W is re-used from phase to phase):
//   P1 volume moments: tensor rules (found on the points by the host) -> three 1-D moment vectors per cell, outer product
//      per lane; general points -> moment GEMM on the f64 MFMA (MomentAcc of pdh_moment.h)
//   P2 per face: 2-D moments (weights: w sigma, -w n_c/2 for the diagonal block in P's frame; w for the coupling block in
//      the per-direction shorter frame F, see pdh_moment.h), expansion into this lane's rows of the S / N_c tensors;
//      tensor sub-face rules -> all faces as one batch of lane tasks, general points -> 2 MFMA per 4 points
//   P3 carry into the own block's piece: last column of the block left of it
//   P4 diagonal block: three-stage contraction on the MFMA (pdh_moment.h), rows written slab by slab
//   P5 coupling blocks in ascending column order: tables -> S, C -> 64 products per lane -> stores
// Instantiations: <N1D, BASIS, GENERAL, SHIFTED, MULTI> - GENERAL = false drops the general-point paths (the host verified tensor
// rules everywhere), SHIFTED = diagonal-first rows; both are facts of a resident problem, and keeping the other variant's
// code out of the kernel is worth 5-10 % (register allocation).  MULTI (FE_DGQ(3) only): polytopes that meet a neighbour along
// several planes or have more than 6 interior / 16 plane entries (METIS-like agglomerates of Cartesian cells) - a coupling
// block is then the SUM over the neighbour's plane entries of C_e (x) S_e, accumulated in registers 32 rows at a time; the
// coupling data of the entries are parked in a per-workgroup row of global memory between P2 and P5 (PdhRows::m2c_scratch:
// LDS stays at the block kernel's 20 KB whatever the number of entries), as 2-D moments (GENERAL) or, with tensor rules, as
// the two 4 x 4 factors of every sub-face (FACT below).
//
// Row stores (P4 epilogue, P5): BUFFER stores through one resource per polytope - address = base of the polytope's rows (four
// SGPRs) + scalar offset (row and piece, advanced by SALU adds) + 32-bit lane offset; with a plain pointer the compiler forms a
// 64-bit vector address per row.  Written with the compiler's own builtin (__builtin_amdgcn_raw_buffer_store_b64), so the
// hazard recogniser and the scheduler see them: rounds 2's version wrote `global_store_dwordx2 v, v[..], s[..]` as inline asm,
// which they cannot see - correct behind VALU / LDS results and SALU-made row pointers, and twice found wrong by the parity
// tests otherwise (a store directly behind an MFMA result; a row pointer restored from a spilled SGPR by v_readlane right in
// front of the store: "VALU writes SGPR -> VMEM reads it" needs five wait states, the store went to a stale address).  The
// resource's size is the polytope's n rows: an offset that is off is DROPPED by the bounds check instead of written.
// Carries travel through LDS and are read back by lane 0 alone (PDHR_CARRY_READ8 below, plain C++ in the shipped build): no asm
// statement of this file issues a memory instruction the compiler cannot see or changes EXEC (tools/isa_lint.py checks the
// generated code object for both).
#pragma once
#include "pdh_moment.h"

#include "pdh_rows_tables.h"

// experiment switches (tools/ab_bench.py; never defined in the shipped build): -DPDHR_EXP=1 no coupling blocks (P5),
// 2 no diagonal block (P4), 3 no faces (P2), 4 no volume (P1), 5 P5 without its global stores, 6 no row stores at all
#ifndef PDHR_EXP
#define PDHR_EXP 0
#endif
// 1: the blocks RIGHT of the diagonal (aligned pieces, no carries) are written before the diagonal block, the blocks left of it
// after: the row stores of a polytope come in three bursts (192 | 64 | 192 for a block-shaped polytope) instead of one of
// 448 at the end - the kernel is bound by how well its stores overlap with the arithmetic of the other waves of the CU
#ifndef PDHR_SPLIT
#define PDHR_SPLIT 1
#endif
#ifndef PDHR_STORE_AUX
#define PDHR_STORE_AUX 18 // gfx940+ cache-policy bits of the row stores: 1 = sc0, 2 = nt, 16 = sc1
#endif

// -DPDHR_CHECK (diagnostic builds only): every data-dependent point index of a global load is checked against the size of
// its array; a violation is recorded in the stamp buffer (slot 15: code | index << 8, slot 14: the wave's slot) and the index is
// clamped to 0 - a software bounds check for a kernel whose fault would otherwise take the process (and the box) down.
#ifdef PDHR_CHECK
#define PDHR_IDX(idx, n, code) pdhr_checked((int64_t)(idx), (int64_t)(n), (code), Rw.stamps, slot)
__device__ __forceinline__ int64_t pdhr_checked(int64_t idx, int64_t n, int code, long long *dbg, int slot)
{
  if (idx < 0 || idx >= n)
    {
      if (dbg)
        {
          dbg[(int64_t)slot * 16 + 15] = (long long)code | (long long)(idx << 8);
          dbg[(int64_t)slot * 16 + 14] = slot;
        }
      return 0;
    }
  return idx;
}
#else
#define PDHR_IDX(idx, n, code) (idx)
#endif

#ifdef PDHR_STAMP
#define PDHR_MARK(k)                                                                                                  \
  do                                                                                                                  \
    {                                                                                                                 \
      const long long tm_ = (long long)__builtin_readcyclecounter();                                                  \
      if (lane == 0 && Rw.stamps)                                                                                     \
        Rw.stamps[(int64_t)slot * 16 + (k)] = tm_;                                                                     \
    }                                                                                                                 \
  while (0)
#define PDHR_T0() const long long t0_ = (long long)__builtin_readcyclecounter()
#define PDHR_ACC(var) var += (long long)__builtin_readcyclecounter() - t0_
#else
#define PDHR_MARK(k)
#define PDHR_T0()
#define PDHR_ACC(var)
#endif

// Eight carries of a shifted piece, read from LDS by lane 0 ALONE straight into the registers that hold the products of rows
// 8 g .. 8 g + 7 (the other lanes keep theirs): the statement narrows EXEC to lane 0 around eight ds_read_b64 and waits for them
// itself, so nothing is outstanding that the compiler does not know of.  It SAVES the mask it finds and restores exactly that
// (round 3 restored the constant -1, i.e. relied on the compiler never placing the statement under a partial mask - true of the
// code it generated, checked by tools/isa_lint.py rule B, but not something an asm string may assume).  -DPDHR_EXEC_CONST
// (diagnostic builds only) brings the old form back.
#if !defined(PDHR_CARRY_ASM) && !defined(PDHR_EXEC_CONST)
// Default since round 4: plain C++ - `if (lane == 0)` around eight LDS reads.  hipcc turns it into s_and_saveexec / ds_read2_b64 x 4
// / s_or exec, i.e. the same instructions as the hand-written statement below (fewer: the reads pair up), counts the reads in its
// own s_waitcnt bookkeeping and owns EXEC; no asm statement of the library changes EXEC any more.  -DPDHR_CARRY_ASM brings the
// asm form back (A/B, tools/ab_bench.py).
#define PDHR_CARRY_READ8(V0, V1, V2, V3, V4, V5, V6, V7, CADDR, G)                                                    \
  do                                                                                                                  \
    {                                                                                                                 \
      if (lane == 0)                                                                                                  \
        {                                                                                                             \
          const double *cp_ = csrc + 8 * (G);                                                                         \
          V0 = cp_[0], V1 = cp_[1], V2 = cp_[2], V3 = cp_[3], V4 = cp_[4], V5 = cp_[5], V6 = cp_[6], V7 = cp_[7];     \
        }                                                                                                             \
      (void)(CADDR);                                                                                                  \
    }                                                                                                                 \
  while (0)
#elif defined(PDHR_EXEC_CONST)
#define PDHR_CARRY_READ8(V0, V1, V2, V3, V4, V5, V6, V7, CADDR, G)                                                    \
  asm volatile("s_mov_b64 exec, 1\n\t"                                                                               \
               "ds_read_b64 %0, %8 offset:%9\n\t"                                                                    \
               "ds_read_b64 %1, %8 offset:%10\n\t"                                                                   \
               "ds_read_b64 %2, %8 offset:%11\n\t"                                                                   \
               "ds_read_b64 %3, %8 offset:%12\n\t"                                                                   \
               "ds_read_b64 %4, %8 offset:%13\n\t"                                                                   \
               "ds_read_b64 %5, %8 offset:%14\n\t"                                                                   \
               "ds_read_b64 %6, %8 offset:%15\n\t"                                                                   \
               "ds_read_b64 %7, %8 offset:%16\n\t"                                                                   \
               "s_mov_b64 exec, -1\n\t"                                                                              \
               "s_waitcnt lgkmcnt(0)"                                                                                \
               : "+v"(V0), "+v"(V1), "+v"(V2), "+v"(V3), "+v"(V4), "+v"(V5), "+v"(V6), "+v"(V7)                      \
               : "v"(CADDR), "n"(64 * (G) + 0), "n"(64 * (G) + 8), "n"(64 * (G) + 16), "n"(64 * (G) + 24),           \
                 "n"(64 * (G) + 32), "n"(64 * (G) + 40), "n"(64 * (G) + 48), "n"(64 * (G) + 56))
#else
#define PDHR_CARRY_READ8(V0, V1, V2, V3, V4, V5, V6, V7, CADDR, G)                                                    \
  do                                                                                                                  \
    {                                                                                                                 \
      unsigned long long exec_saved_;                                                                                 \
      asm volatile("s_mov_b64 %8, exec\n\t"                                                                          \
                   "s_mov_b64 exec, 1\n\t"                                                                           \
                   "ds_read_b64 %0, %9 offset:%10\n\t"                                                               \
                   "ds_read_b64 %1, %9 offset:%11\n\t"                                                               \
                   "ds_read_b64 %2, %9 offset:%12\n\t"                                                               \
                   "ds_read_b64 %3, %9 offset:%13\n\t"                                                               \
                   "ds_read_b64 %4, %9 offset:%14\n\t"                                                               \
                   "ds_read_b64 %5, %9 offset:%15\n\t"                                                               \
                   "ds_read_b64 %6, %9 offset:%16\n\t"                                                               \
                   "ds_read_b64 %7, %9 offset:%17\n\t"                                                               \
                   "s_mov_b64 exec, %8\n\t"                                                                          \
                   "s_waitcnt lgkmcnt(0)"                                                                            \
                   : "+v"(V0), "+v"(V1), "+v"(V2), "+v"(V3), "+v"(V4), "+v"(V5), "+v"(V6), "+v"(V7), "=&s"(exec_saved_) \
                   : "v"(CADDR), "n"(64 * (G) + 0), "n"(64 * (G) + 8), "n"(64 * (G) + 16), "n"(64 * (G) + 24),       \
                     "n"(64 * (G) + 32), "n"(64 * (G) + 40), "n"(64 * (G) + 48), "n"(64 * (G) + 56));                \
      (void)exec_saved_;                                                                                              \
    }                                                                                                                 \
  while (0)
#endif

namespace pdhr
{
using pdh::static_for;
using pdhm::d2_t;
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

constexpr int ROWS_HDR = 12, ROWS_MAXE = 16, ROWS_ENT = 12, ROWS_REC = ROWS_HDR + ROWS_MAXE * ROWS_ENT; // per-slot record
constexpr int MAXF = 6;  // INTERIOR faces per polytope the LDS layout provides for (6: 20.3 KB per wave = 8 waves per CU)
constexpr int FREC = 33; // face record: L_i[8], (s_t L_j)[3][8], pad (odd stride)
constexpr int FCH = 32;  // face points per chunk
constexpr int FSTEP = 4 * FREC * 8;

// Kinds of the kernel.  (N1D, BASIS) = (4, 0) - FE_DGQ(3), n = 64 - is the row-piece kernel described above.  Every other
// kind (FE_DGQ(1,2), FE_AggloDGP(1..3)) is the "streamed" variant: its n rows are one contiguous range of n * rlen values
// that is written front to back (P4 / P5 below), and it exists for verified tensor rules only (no general-point paths).
// BASIS = 0: all (k0,k1,k2) < N1D; BASIS = 1: k0 + k1 + k2 <= p (pdh_basis.h: multi_indices; x index fastest in both).
template <int N1D, int BASIS>
struct RowsKind
{
  static constexpr bool SMALL = !(N1D == 4 && BASIS == 0);
  static constexpr int NA = 2 * N1D - 1;
  static constexpr int NF = BASIS == 0 ? N1D * N1D * N1D : N1D * (N1D + 1) * (N1D + 2) / 6; // functions
  static constexpr int NS = BASIS == 0 ? N1D * N1D : N1D * (N1D + 1) / 2;                   // pairs of tangential digits
  static constexpr int SS = SMALL ? ((NS * NS > 64 ? NS * NS : 64) + 7) / 8 * 8 : 64; // per-face slot: 8x8 moments, later S
  // task vectors of P2 (tensor sub-face rules): [64 tasks][self | cross], 8 + 8 doubles per task - packed to NA + NA (odd
  // stride) where that buys a resident wave: W, and with it the LDS of a wave, is sized by them for degree <= 2, and the
  // streamed kinds run at the speed their occupancy allows (FE_DGQ(2): 1.14 / 0.99 / 0.86 / 0.78 ms at 4 / 5 / 6 / 7 waves
  // per CU).  FE_DGQ(2): 21 336 -> 19 288 bytes = 8 waves per CU instead of 7 (0.76 -> 0.70 ms); degree 1: 13 480 -> 11 432
  // bytes = the 12 waves per CU their 156 VGPRs allow (0.27 -> 0.21 ms).  FE_AggloDGP(2,3) stay at 8 / 6 waves either way
  // (registers / the stage buffers of P4) and are 1-2 % slower with the odd stride: not packed.
  static constexpr bool PACKED = SMALL && (N1D == 2 || (N1D == 3 && BASIS == 0));
  static constexpr int MVC = PACKED ? NA : 8, MVS = PACKED ? 2 * NA + 1 : 16;
  // Stages 2 / 3 of the diagonal block run in passes over K1P values of k1 (all of them, except for FE_AggloDGP(3): there the
  // stage buffers size the LDS of a wave, and T2 for two values of k1 at a time makes 23 032 instead of 26 616 bytes = 7
  // resident waves per CU instead of 6 - these kinds run at the speed their occupancy allows, see MVS below)
  static constexpr int K1P = (N1D == 4 && BASIS == 1) ? 2 : N1D;
  static constexpr int NC = K1P * NS;                                                  // stage-2 work items per a0 and pass (upper bound)
  static constexpr int T1 = 4 * N1D * NA * NA, T2 = 3 * NC * NA;                       // stage buffers of the diagonal block
  // index of the digit pair (ka, kb) among the pairs that occur
  __host__ __device__ static constexpr int pair(int ka, int kb) { return BASIS == 0 ? ka + N1D * kb : kb * N1D - kb * (kb - 1) / 2 + ka; }
  // number of the function with digits (k0, k1, k2)
  __device__ __forceinline__ static int findex(int k0, int k1, int k2)
  {
    if constexpr (BASIS == 0)
      return k0 + N1D * (k1 + N1D * k2);
    else
      {
        int zoff = 0;
        for (int z = 0; z < k2; ++z)
          zoff += (N1D - z) * (N1D - z + 1) / 2;
        const int nn = N1D - k2;
        return zoff + k1 * nn - k1 * (k1 - 1) / 2 + k0;
      }
  }
};

template <int N1D, int BASIS = 0>
constexpr int w_doubles_rows()
{
  using A = pdhm::MomentAcc<N1D>;
  using K = RowsKind<N1D, BASIS>;
  using M = pdhm::MT<N1D>;
  if constexpr (!K::SMALL)
    {
      constexpr int w_rec = A::VCH * A::VREC > FCH * FREC ? A::VCH * A::VREC : FCH * FREC;
      constexpr int w_con = 4 * 2 * 2 * 64 + 2 * 4 * 64 + 64; // T1B + T2B + carry of the own piece
      return w_rec > w_con ? w_rec : w_con;
    }
  else
    {
      // task vectors of P2 [64][16]; S / C scratch (two tables + T); stage buffers of the diagonal block
      constexpr int w_s = 2 * M::LTAB + 128, w_d = (K::T1 + K::T2 + 63) / 64 * 64;
      // (P1: [16 cells][3][8]; S / C of all faces in lock-step: T_f at W + 128 f)
      constexpr int w_t = 64 * K::MVS > 128 * MAXF ? 64 * K::MVS : 128 * MAXF;
      constexpr int w0 = w_s > w_t ? w_s : w_t;
      return w0 > w_d ? w0 : w_d;
    }
}
template <int N1D, int BASIS = 0>
constexpr int lds_doubles_rows()
{
  using M = pdhm::MT<N1D>;
  using K = RowsKind<N1D, BASIS>;
  constexpr int w = w_doubles_rows<N1D, BASIS>();
  // streamed kinds add: C of every interior face [MAXF][16], the diagonal block [n][n], digit table [3][n] ints
  constexpr int extra = !K::SMALL ? 0 : MAXF * 16 + K::NF * K::NF + (3 * K::NF + 1) / 2 + 1;
  // (160 KB of LDS per CU hold 6 waves of the FE_AggloDGP(3) kind only up to 26 624 bytes each - measured: 512 bytes more
  // and the sixth workgroup of a CU waits for a second round)
  return 3 * M::LTAB + MAXF * K::SS + (!K::SMALL ? 64 : 0) /* diagv */ + 16 /* C */ + 16 /* coef */ + w + extra;
}

// multi-index digits of a function index i = k0 + 4 k1 + 16 k2: digit of axis c, and u = k_i + 4 k_j of the other two (i < j)
__device__ __forceinline__ int digit_c(int i, int c) { return (i >> (2 * c)) & 3; }
__device__ __forceinline__ int digits_t(int i, int c)
{
  const int k0 = i & 3, k1 = (i >> 2) & 3, k2 = (i >> 4) & 3;
  return c == 0 ? (k1 + 4 * k2) : (c == 1 ? (k0 + 4 * k2) : (k0 + 4 * k1));
}

// GENERAL = false: instantiation without the general-point paths (tensor rules verified on every sub-cell and sub-face, no
// face entry with more than 32 sub-faces) - the MFMA moment accumulators and their operand addresses are then not part of
// the kernel at all, which the register allocation of the remaining phases feels.
// SHIFTED: deal.II's diagonal-first rows (else ascending) - a compile-time fact of the instantiation like GENERAL, for the
// same reason: the code of the other layout is not there to be allocated for.
// MULTI (FE_DGQ(3) only): the interface with a neighbour may span SEVERAL planes - "staircase" faces of METIS-like agglomerates of
// Cartesian cells (reference examples/poisson.cc:543-566 partitions the cell graph; source/agglomeration_handler.cc:1129-1165
// collects all sub-faces shared with one neighbour into one polytopal face).  Every plane is one entry of the record; the
// entries of a neighbour share its block, and the block is the SUM of their Kronecker products C_e (x) S_e.  The numbers of
// entries per polytope (PdhRows::maxe, <= 48) and of interior entries (coupling-moment slots in LDS, PdhRows::maxf) are then
// run-time facts of the resident problem, and the slots move to the end of the LDS allocation, sized at launch.  A separate
// instantiation, so that the kernel of the block-shaped polytopes keeps its registers and its LDS footprint.
template <int N1D, int BASIS = 0, bool GENERAL = (N1D == 4), bool SHIFTED = true, bool MULTI = false>
__global__ void __launch_bounds__(PDH_WAVE, 2) k_rows(const PdhDev P, const PdhRows Rw, const double *__restrict__ mt, const int n_owned)
{
  static_assert(N1D >= 2 && N1D <= 4, "the row kernel is written for degree 1 .. 3");
  static_assert(!GENERAL || N1D == 4, "general-point paths exist for degree 3 only");
  static_assert(!MULTI || (N1D == 4 && BASIS == 0), "several planes per neighbour: FE_DGQ(3) only");
  using RK = RowsKind<N1D, BASIS>;
  constexpr bool SMALL = RK::SMALL;
  // MULTI with verified tensor rules: the coupling data of an interior entry are kept per SUB-FACE in factorised form - on an
  // axis-aligned rectangle with a tensor rule  S_sub[(k_i,k_j),(l_i,l_j)] = X[k_i,l_i] Y[k_j,l_j],  X[k,l] = sum_alpha a_alpha
  // B^P_k(x_alpha) B^Q_l(x_alpha) (16 + 16 numbers from 4 + 4 points, both bases evaluated in their own boxes) - instead of
  // as 49 two-dimensional moments in a common frame that need direction tables and two contraction stages per entry (6 k
  // cycles each, a third of this instantiation's time; its entries are small: 1.4 sub-faces on average)
  constexpr bool FACT = MULTI && !GENERAL;
  constexpr int MS = RK::SS; // doubles per interior-face slot
  constexpr int NF = RK::NF, NS = RK::NS;
  using M = pdhm::MT<N1D>;
  using Acc = pdhm::MomentAcc<N1D>;
  constexpr int NA = M::NA, NAP = M::NAP, NG = M::NG, DIM = 3;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  // LDS tables: once per wave (the kernel is persistent: a wave works through slots blockIdx.x, blockIdx.x + gridDim.x, ...)
  auto sel3 = [](int c, double x0, double x1, double x2) { return c == 0 ? x0 : (c == 1 ? x1 : x2); };
  double *tabE = lds, *tabD = lds + M::LTAB, *tabF = lds + 2 * M::LTAB;
  double *diagv = lds + 3 * M::LTAB + (MULTI ? 0 : MAXF * MS); // [64] diagonal entries A[R,R]
  double *Cbuf = diagv + (!SMALL ? 64 : 0); // [4][4]  (the streamed kinds have no diagv)
  double *coefL = Cbuf + 16;        // [4][4] monomial coefficients of the 1-D basis (centred variable)
  double *W = coefL + 16;           // phase-local
  // [maxf][8][8] coupling moments of every interior entry: in front of diagv.  MULTI (up to 40 interior entries): parked in
  // global memory between P2 and P5 - one row of PdhRows::m2c_scratch per workgroup, 512 bytes per entry, written once and
  // read once or twice per polytope by the same wave (it stays in L2) - and staged through ONE slot behind W when S of an
  // entry is built.  In LDS they cost 512 bytes per entry of the polytope with the MOST entries: 36 KB per wave on the bench
  // mesh of grown agglomerates = 4 resident waves per CU, one per SIMD, and this kernel lives on the latency the other wave
  // of a SIMD hides.
  double *M2c = MULTI ? W + w_doubles_rows<N1D, BASIS>() : lds + 3 * M::LTAB;
  double *M2g = nullptr;
  if constexpr (MULTI)
    M2g = Rw.m2c_scratch + (size_t)blockIdx.x * (size_t)Rw.scratch_stride; // (FACT: [sub-face][X 16 | Y 16])
  const int maxf = MULTI ? Rw.maxf : MAXF, maxe = MULTI ? Rw.maxe : ROWS_MAXE;
  // streamed kinds only: behind W
  double *Call = W + w_doubles_rows<N1D, BASIS>(); // [MAXF][4][4] C of every interior face
  double *Dblk = Call + MAXF * 16;          // [n][n] diagonal block
  int *dig = reinterpret_cast<int *>(Dblk + NF * NF); // [3 axes][n]: k_c | index of the pair of the other two digits << 4
  if constexpr (SMALL)
    if (lane < NF)
      {
        int k0 = 0, k1 = 0, k2 = 0, cnt = 0;
        for (int iz = 0; iz < N1D; ++iz)
          for (int iy = 0; iy < (BASIS == 0 ? N1D : N1D - iz); ++iy)
            for (int ix = 0; ix < (BASIS == 0 ? N1D : N1D - iy - iz); ++ix)
              {
                if (cnt == lane)
                  k0 = ix, k1 = iy, k2 = iz;
                ++cnt;
              }
        dig[0 * NF + lane] = k0 | (RK::pair(k1, k2) << 4);
        dig[1 * NF + lane] = k1 | (RK::pair(k0, k2) << 4);
        dig[2 * NF + lane] = k2 | (RK::pair(k0, k1) << 4);
      }
  for (int t = lane; t < 3 * M::TAB; t += PDH_WAVE)
    lds[(t / NAP) * M::RS + t % NAP] = mt[t];
  if (lane < 16)
    coefL[lane] = P.tab.coef[lane >> 2][lane & 3];
  auto rl_i = [](int v, int t) { return __builtin_amdgcn_readlane(v, t); };
  auto rl_d = [](double v, int t) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), t), __builtin_amdgcn_readlane(__double2loint(v), t));
  };
  // Per-slot record (pdh_capi.cpp:build_rows_tables): ROWS_HDR header doubles, then ROWS_MAXE face entries of ROWS_ENT
  // doubles (neighbour boxes included) - ONE round of loads per polytope, requested a whole polytope ahead: under the store
  // traffic of this kernel a dependent global load takes ~7k cycles, and the chain slot -> face table -> neighbour boxes
  // used to cost three of them at the head of every polytope.
  struct Meta
  {
    double e[ROWS_ENT]; // lanes 0 .. maxe-1: the face entry; lanes maxe .. maxe+ROWS_HDR-1: e[0] = header value
  };
  const int hb = maxe; // first header lane (maxe <= 48, pdh_capi.cpp: build_rows_tables)
  auto load_meta = [&](int s_) {
    Meta m;
    const double *r = Rw.meta + (int64_t)s_ * (ROWS_HDR + maxe * ROWS_ENT);
    for (int k = 0; k < ROWS_ENT; ++k)
      m.e[k] = 0.0;
    if (lane < maxe)
      for (int k = 0; k < ROWS_ENT; ++k)
        m.e[k] = r[ROWS_HDR + lane * ROWS_ENT + k];
    else if (lane < maxe + ROWS_HDR)
      m.e[0] = r[lane - maxe];
    return m;
  };
  // Work distribution: the first polytope of a wave is its block number, every further one comes from a device-wide counter
  // (PdhRows::sched[0], one returning atomic add per polytope, requested a whole polytope ahead of its use).  A static stride
  // (slot += gridDim.x) left the waves of the faster CUs / XCDs idle while the slowest finished: per-wave spans of 0.81 .. 1.15
  // of the mean (in-kernel stamps, 2048 waves x 16 polytopes), i.e. a tail of 13 % of the kernel.  The wave that leaves last
  // (sched[1] counts leavers) puts both words back to zero for the next launch - launches of one context are stream-ordered.
  auto leave = [&]() {
    // (every wave has made its last request to sched[0] before it counts itself out: the last one out may reset both)
    if constexpr (!RowsKind<N1D, BASIS>::SMALL)
    if (threadIdx.x == 0)
      if (__hip_atomic_fetch_add(Rw.sched + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
        {
          __hip_atomic_store(Rw.sched, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(Rw.sched + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
  };
  int slot = blockIdx.x;
  if (slot >= n_owned)
    { // (the launcher never starts more waves than polytopes; counted out all the same, or the counters would not be reset)
      leave();
      return;
    }
  Meta cur = load_meta(slot);
  const int lane_outer = lane;
#pragma unroll 1
  for (;;)
  {
  // (streamed kinds: static stride - their polytopes are short, 45 .. 80 requests per microsecond would sit at the ~88 per
  // microsecond one counter word sustains: measured 0.41 -> 0.52 ms for FE_AggloDGP(2) with the counter)
  constexpr bool DYNAMIC = !SMALL;
  int nslot_v = slot + (int)gridDim.x;
  if constexpr (DYNAMIC)
    if (lane_outer == 0)
      nslot_v = (int)gridDim.x + (int)__hip_atomic_fetch_add(Rw.sched, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Everything derived from the lane number below is loop-invariant; hoisted out of this loop it would sit in ~60 VGPRs for
  // the whole kernel (the compiler did exactly that: 256 VGPRs + spills).  An opaque copy ties it to the iteration.
  int lane = lane_outer;
  asm volatile("" : "+v"(lane));
  const bool act = lane < NA * NA;
  const int a0 = act ? lane / NA : 0, a1 = act ? lane % NA : 0;
  PDHR_MARK(0);
  PDH_WAVE_SYNC();
  // ---- decode this polytope's record; request the next one's
  const double lo0 = rl_d(cur.e[0], hb + 1), lo1 = rl_d(cur.e[0], hb + 2), lo2 = rl_d(cur.e[0], hb + 3);
  const double ih0 = rl_d(cur.e[0], hb + 4), ih1 = rl_d(cur.e[0], hb + 5), ih2 = rl_d(cur.e[0], hb + 6);
  const int nfaces = (int)__double_as_longlong(rl_d(cur.e[0], hb + 0));
  const int64_t rbase = __double_as_longlong(rl_d(cur.e[0], hb + 7));
  const int rlen = (int)__double_as_longlong(rl_d(cur.e[0], hb + 8));
  const int L = (int)__double_as_longlong(rl_d(cur.e[0], hb + 9));
  const int64_t vq_b = __double_as_longlong(rl_d(cur.e[0], hb + 10)), vq_e = __double_as_longlong(rl_d(cur.e[0], hb + 11));
  const int m0 = !SMALL ? (L >> 6) : L / NF;
  // the polytope's rows as a buffer: n rows of rlen values from rbase on (see "Row stores" at the top of this file)
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(P.values + rbase, 0, NF * rlen * 8, 0x00020000);
  auto row_store = [&](double v, uint32_t lane_bytes, uint32_t row_bytes) {
#if PDHR_EXP == 6
    if (P.n < 0) // (experiment: everything computed, nothing stored)
#endif
    // cache policy sc1 | nt: the values are written once and not read again by this kernel; written through (the line is not kept in
    // the XCD's L2) the same stores run 5 % faster in the store-only twin of this kernel (tools/probes/rows_store_probe.hip:
    // 1.37 -> 1.30 ms at 8 waves per CU)
    // (the streamed kinds keep the default policy: their rows are not multiples of 128 bytes, neighbouring pieces share lines, and
    // written through the halves of a shared line no longer merge in L2 - FE_DGQ(2) 0.74 -> 0.79 ms with sc1)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, v), vrs, (int)lane_bytes, (int)row_bytes, SMALL ? 0 : PDHR_STORE_AUX);
  };
  // the face table of the polytope lives in the lanes (lane t = face t); a face's entries are read with v_readlane
  const long long pb_ = __double_as_longlong(cur.e[0]);
  const int t_pblo = (int)(uint32_t)pb_, t_pbhi = (int)(pb_ >> 32);
  const long long i1_ = __double_as_longlong(cur.e[1]), i2_ = __double_as_longlong(cur.e[2]);
  const int t_pcnt = lane < nfaces ? (int)(uint32_t)i1_ : 0, t_nbr = lane < nfaces ? (int)(i1_ >> 32) : -1;
  const int t_axis = (int)(i2_ & 0xff), t_flags = (int)((i2_ >> 8) & 0xff), t_blk = lane < nfaces ? (int)(i2_ >> 32) : -1;
  const double t_coord = cur.e[3], t_sigma = cur.e[4], t_nsign = cur.e[5];
  const double t_qlo0 = cur.e[6], t_qlo1 = cur.e[7], t_qlo2 = cur.e[8], t_qih0 = cur.e[9], t_qih1 = cur.e[10], t_qih2 = cur.e[11];
  int n_bdry = 0; // boundary entries come first; coupling moments are kept per interior face
  while (n_bdry < nfaces && rl_i(t_blk, n_bdry) < 0)
    ++n_bdry;
  // (general volume path only: MFMA accumulators and operand addresses; dead registers in the tensor path)
  pdhm::MomentAcc<N1D> ma;
  if constexpr (GENERAL) // (the other kinds exist for verified tensor rules only)
    if (Rw.vq_tensor_n == 0)
      {
        ma.init(lane);
        ma.init_addr(W, lane);
      }

  PDHR_MARK(1);
  // ================= P1: volume moments ========================================================================
  double accM[NAP];
  for (int a = 0; a < NAP; ++a)
    accM[a] = 0.0;
  const int tn = Rw.vq_tensor_n;
  if (tn > 0)
    {
      // Verified tensor rules (pdh_problem::vq_tensor_n): the points come cell by cell, x_(i,j,k) = (X_i, Y_j, Z_k),
      // JxW = a_i b_j c_k, so the cell's moment tensor is the outer product of three 1-D moment vectors
      //   m_d[a] = sum_i W_d[i] L_a(x^_d[i]),   W_0 = w_(i,0,0), W_1 = w_(0,j,0) / w_000, W_2 = w_(0,0,k) / w_000
      // - 3 n sums per cell instead of n^3 points through the MFMA.  Lanes (cell of the batch, direction) form the vectors,
      // then every lane adds its (a0, a1) row:  M[a0,a1,.] += m_0[a0] m_1[a1] m_2[.].
      const int64_t qb = vq_b, qe = vq_e;
      const int m3 = tn * tn * tn;
      const int ncell = (int)((qe - qb) / m3);
      double *mv = W; // [16 cells][3][8]
#if PDHR_EXP == 4
      for (int c0 = 0; c0 < ncell && P.n < 0; c0 += 16)
#else
      for (int c0 = 0; c0 < ncell; c0 += 16)
#endif
        {
          const int nb = ncell - c0 < 16 ? ncell - c0 : 16;
          PDH_WAVE_SYNC();
          if (lane < 3 * nb)
            {
              const int cl = lane / 3, d = lane - 3 * cl;
              const int64_t base = qb + (int64_t)(c0 + cl) * m3;
              const int64_t step = d == 0 ? 1 : (d == 1 ? tn : tn * tn);
              const double lo_d = sel3(d, lo0, lo1, lo2), ih_d = sel3(d, ih0, ih1, ih2);
              const double w000 = P.vq_w[PDHR_IDX(base, P.vq_stride, 1)];
              const double sc = d == 0 ? 1.0 : 1.0 / w000;
              double mo[NA];
              for (int a = 0; a < NA; ++a)
                mo[a] = 0.0;
              // all loads of the task first (tn <= 8, checked by the host): in a rolled loop each point would wait for
              // its own global load - four serial round trips per task
              double xr[8], wr[8];
              static_for<0, 8>([&](auto i_) {
                constexpr int i = i_;
                xr[i] = i < tn ? P.vq_x[(int64_t)d * P.vq_stride + PDHR_IDX(base + i * step, P.vq_stride, 2)] : 0.0;
                wr[i] = i < tn ? P.vq_w[PDHR_IDX(base + i * step, P.vq_stride, 3)] : 0.0;
              });
              static_for<0, 8>([&](auto i_) {
                constexpr int i = i_;
                if (i < tn)
                  {
                    const double x = (xr[i] - lo_d) * ih_d;
                    const double w = wr[i] * sc;
                    double Lx[NA];
                    pdhm::legendre01<NA>(x, Lx);
                    for (int a = 0; a < NA; ++a)
                      mo[a] += w * Lx[a];
                  }
              });
              for (int a = 0; a < NA; ++a)
                mv[(cl * 3 + d) * 8 + a] = mo[a];
            }
          PDH_WAVE_SYNC();
          for (int cl = 0; cl < nb; ++cl)
            {
              const double pm = mv[(cl * 3 + 0) * 8 + a0] * mv[(cl * 3 + 1) * 8 + a1];
              for (int a = 0; a < NA; ++a)
                accM[a] += pm * mv[(cl * 3 + 2) * 8 + a];
            }
        }
      PDH_WAVE_SYNC();
    }
  else if constexpr (GENERAL)
  {
    // point data two chunks ahead, in two statically addressed register sets (see P2 on why no copies)
    const int64_t qb = vq_b, qe = vq_e;
    struct VRaw
    {
      double x0, x1, x2, w;
    };
    auto vissue = [&](int64_t base) {
      VRaw r;
      const bool on = base + lane < qe;
      r.x0 = on ? P.vq_x[0 * P.vq_stride + base + lane] : 0.0;
      r.x1 = on ? P.vq_x[1 * P.vq_stride + base + lane] : 0.0;
      r.x2 = on ? P.vq_x[2 * P.vq_stride + base + lane] : 0.0;
      r.w = on ? P.vq_w[base + lane] : 0.0;
      return r;
    };
    VRaw va = vissue(qb), vb = vissue(qb + Acc::VCH);
    auto vchunk = [&](VRaw &v, int64_t base) {
      const int cnt = (int)((qe - base < Acc::VCH) ? (qe - base) : Acc::VCH);
      double xu[DIM] = {0.5, 0.5, 0.5}, w = 0.0;
      if (lane < cnt)
        {
          xu[0] = (v.x0 - lo0) * ih0;
          xu[1] = (v.x1 - lo1) * ih1;
          xu[2] = (v.x2 - lo2) * ih2;
          w = v.w;
        }
      v = vissue(base + 2 * Acc::VCH);
      Acc::write_volume_record(W + lane * Acc::VREC, xu, w);
      PDH_WAVE_SYNC();
      ma.volume_chunk_full(); // dead points carry zero weights
      PDH_WAVE_SYNC();
    };
#if PDHR_EXP == 4
    for (int64_t base = qb; base < qe && P.n < 0; base += 2 * Acc::VCH)
#else
    for (int64_t base = qb; base < qe; base += 2 * Acc::VCH)
#endif
      {
        vchunk(va, base);
        if (base + Acc::VCH < qe)
          vchunk(vb, base + Acc::VCH);
      }
  }

  PDHR_MARK(2);
  // ================= P2: faces ==================================================================================
  int t_nsub = 0, t_soff = 0; // FACT: lane t = entry t: its sub-faces and how many the interior entries before it have
  double accS[NAP], accN[DIM][NAP];
  for (int a = 0; a < NAP; ++a)
    {
      accS[a] = 0.0;
      for (int c = 0; c < DIM; ++c)
        accN[c][a] = 0.0;
    }
  {
    // operand addresses of the 2-D moment GEMM: rows a_i (two blocks, replicated), columns (t, a_j)
    typedef __attribute__((address_space(3))) const char lds_cchar;
    const unsigned wbase = (unsigned)(uintptr_t)(lds_cchar *)reinterpret_cast<const char *>(W);
    const int kq = lane >> 4, blk = (lane >> 2) & 3, idx = lane & 3;
    const unsigned rb = wbase + kq * FREC * 8;
    const unsigned adA = rb + (4 * (blk & 1) + idx) * 8;
    const unsigned adB0 = rb + (8 + 4 * blk + idx) * 8;
    const unsigned adB1 = rb + (8 + 4 * ((blk + 1) & 3) + idx) * 8;
    const unsigned adB2 = rb + (24 + 4 * ((blk == 1 || blk == 2) ? 1 : 0) + idx) * 8;
    const int half = lane >> 5, pt = lane & 31;
    // uniform parameters of face t
    struct FP
    {
      int c, ti, tj, nbr, npass;
      int64_t pb, pe;
      bool sep;
      double lo_t0, lo_t1, ih_t0, ih_t1, loF0, loF1, ihF0, ihF1, nsg, xpl, ptol, sig;
      bool masked, fast_j;
    };
    auto face_params = [&](int t) {
      FP fp;
      fp.c = rl_i(t_axis, t);
      fp.ti = fp.c == 0 ? 1 : 0;
      fp.tj = fp.c == 2 ? 1 : 2;
      fp.nbr = rl_i(t_nbr, t);
      fp.pb = ((int64_t)rl_i(t_pbhi, t) << 32) | (uint32_t)rl_i(t_pblo, t);
      fp.pe = fp.pb + rl_i(t_pcnt, t);
      fp.lo_t0 = fp.ti == 0 ? lo0 : lo1;
      fp.lo_t1 = fp.tj == 1 ? lo1 : lo2;
      fp.ih_t0 = fp.ti == 0 ? ih0 : ih1;
      fp.ih_t1 = fp.tj == 1 ? ih1 : ih2;
      fp.loF0 = fp.lo_t0, fp.loF1 = fp.lo_t1, fp.ihF0 = fp.ih_t0, fp.ihF1 = fp.ih_t1;
      fp.sep = false;
      if (fp.nbr >= 0)
        { // frames of the coupling moments: per tangential direction the shorter of the two box intervals
          const double q0 = rl_d(t_qlo0, t), q1 = rl_d(t_qlo1, t), q2 = rl_d(t_qlo2, t);
          const double i0 = rl_d(t_qih0, t), i1 = rl_d(t_qih1, t), i2 = rl_d(t_qih2, t);
          const double lqa = fp.ti == 0 ? q0 : q1, lqb = fp.tj == 1 ? q1 : q2;
          const double iqa = fp.ti == 0 ? i0 : i1, iqb = fp.tj == 1 ? i1 : i2;
          if (iqa > fp.ih_t0)
            fp.loF0 = lqa, fp.ihF0 = iqa, fp.sep = true;
          if (iqb > fp.ih_t1)
            fp.loF1 = lqb, fp.ihF1 = iqb, fp.sep = true;
        }
      fp.npass = (fp.nbr >= 0 && fp.sep) ? 2 : 1;
      fp.nsg = rl_d(t_nsign, t);
      fp.xpl = rl_d(t_coord, t);
      fp.ptol = 1e-9 / sel3(fp.c, ih0, ih1, ih2);
      fp.sig = rl_d(t_sigma, t);
      fp.masked = (rl_i(t_flags, t) & 1) != 0;
      fp.fast_j = (rl_i(t_flags, t) & 2) != 0; // tensor sub-face rules: the j direction runs fastest
      return fp;
    };
    // point data of one chunk, loaded one chunk ahead of its use (half 0: x_i; half 1: x_j and the weights)
    struct Raw
    {
      double x, wS, wC, n, xc;
      bool on;
    };
    auto issue = [&](const FP &fp, int64_t base) {
      Raw r;
      r.on = base + pt < fp.pe;
      const int64_t q = r.on ? base + pt : fp.pb;
      r.x = P.ap_x[(int64_t)(half == 0 ? fp.ti : fp.tj) * P.ap_stride + q];
      r.wS = r.wC = r.n = r.xc = 0.0;
      if (half == 1)
        {
          r.wS = P.ap_wself[q];
          r.wC = P.ap_wcross[q];
          if (fp.masked)
            { // only boundary runs that span several planes need the per-point normal and plane coordinate
              r.n = P.ap_n[(int64_t)fp.c * P.ap_stride + q];
              r.xc = P.ap_x[(int64_t)fp.c * P.ap_stride + q];
            }
        }
      return r;
    };
    // expansion of a face's 2-D moments M2 = [3][8][8] into this lane's (a0, a1) rows of the 3-D tensors
    auto expand = [&](const FP &fp, const double *M2) {
      const int c = fp.c;
      // expansion into this lane's (a0, a1) rows of the 3-D tensors: M[a0,a1,a2] += L_{a_c}(zeta) M2[a_i][a_j]
      const double zeta = (fp.xpl - sel3(c, lo0, lo1, lo2)) * sel3(c, ih0, ih1, ih2);
      double Lc[NA];
      pdhm::legendre01<NA>(zeta, Lc);
      if (c == 2)
        {
          const double mS = M2[0 * 64 + a0 * 8 + a1], mN = M2[1 * 64 + a0 * 8 + a1];
          for (int a = 0; a < NA; ++a)
            {
              accS[a] += Lc[a] * mS;
              accN[2][a] += Lc[a] * mN;
            }
        }
      else
        {
          // c == 1: (i, j) = (0, 2): factor L_{a1}(zeta), row a0;  c == 0: (i, j) = (1, 2): factor L_{a0}(zeta), row a1
          const int asel = c == 1 ? a1 : a0, arow = c == 1 ? a0 : a1;
          double lc_ = Lc[0];
          static_for<1, NA>([&](auto a_) {
            constexpr int a = a_;
            lc_ = asel == a ? Lc[a] : lc_;
          });
          // (both tensors updated, one with a zero factor: "if (c == 1) accN[1] else accN[0]" became an array indexed by a
          // run-time value, i.e. scratch memory - twelve scratch loads / stores per face in the middle of the kernel)
          const double l0 = c == 1 ? 0.0 : lc_, l1 = c == 1 ? lc_ : 0.0;
          for (int a = 0; a < NA; ++a)
            {
              const double mS = M2[0 * 64 + arow * 8 + a], mN = M2[1 * 64 + arow * 8 + a];
              accS[a] += lc_ * mS;
              accN[0][a] += l0 * mN;
              accN[1][a] += l1 * mN;
            }
        }
    };
    const int fn = Rw.fq_tensor_n;
    const int nf2 = fn > 0 ? fn * fn : 1;
    const int my_nsub = (fn > 0 && lane < nfaces) ? t_pcnt / nf2 : 0;
    // (an entry with more than 32 sub-faces does not fit the 64 lane tasks of a batch: such polytopes take the MFMA path)
    const bool face_tensor = fn > 0 && __ballot(my_nsub > 32) == 0ull;
    if constexpr (FACT)
      {
        t_nsub = my_nsub;
        int run = 0;
        for (int t = n_bdry; t < nfaces; ++t)
          {
            t_soff = lane == t ? run : t_soff;
            run += rl_i(my_nsub, t);
          }
      }
    if (face_tensor)
      {
        // Verified tensor rules on the sub-faces (pdh_problem::fq_tensor_n): every group of fn^2 points is a rule
        // x = (X_alpha, Y_beta), JxW = a_alpha b_beta on an axis-aligned rectangle, so a sub-face's 2-D moments are the outer
        // product of two 1-D moment vectors.  sigma and the normal are constant on the face: the two diagonal-block tensors
        // are multiples of ONE sum G = sum_sub m_i (x) m_j; the coupling tensor uses JxW of the other side (and the frame F).
        // Lane tasks = (face entry, sub-face, direction), up to 64 at a time (all faces of a box polytope in one batch): the
        // global loads of a whole batch are in flight together.
        double *mv = W;         // [64 tasks][self 8 | cross 8]
        int tb = 0;
#ifdef PDHR_STAMP
        long long tt_task = 0, tt_face = 0;
#endif
#if PDHR_EXP == 3
        while (tb < nfaces && P.n < 0)
#else
        while (tb < nfaces)
#endif
          {
            int te = tb, ntask = 0;
            while (te < nfaces)
              {
                const int ns2 = 2 * rl_i(my_nsub, te);
                if (ntask + ns2 > 64)
                  break;
                ntask += ns2;
                ++te;
              }
            // this lane's task
            int mt = -1, rel = 0;
            {
              int s0 = 0;
              for (int t = tb; t < te; ++t)
                {
                  const int ns2 = 2 * rl_i(my_nsub, t);
                  if (lane >= s0 && lane < s0 + ns2)
                    {
                      mt = t;
                      rel = lane - s0;
                    }
                  s0 += ns2;
                }
            }
            PDH_WAVE_SYNC();
            {
              PDHR_T0();
              const int src = mt >= 0 ? mt : 0;
              // the entry's parameters live in lane `src` of the face table
              const int c = __shfl(t_axis, src), flags = __shfl(t_flags, src), nbr = __shfl(t_nbr, src);
              const double nsg = __shfl(t_nsign, src), xpl = __shfl(t_coord, src);
              const int64_t pb = ((int64_t)__shfl(t_pbhi, src) << 32) | (uint32_t)__shfl(t_pblo, src);
              const double q0 = __shfl(t_qlo0, src), q1 = __shfl(t_qlo1, src), q2 = __shfl(t_qlo2, src);
              const double i0 = __shfl(t_qih0, src), i1 = __shfl(t_qih1, src), i2 = __shfl(t_qih2, src);
              const int soff_src = FACT ? __shfl(t_soff, src) : 0; // (cross-lane reads stay outside the divergent part)
              if (mt >= 0)
                {
                  const int sb = rel >> 1, dir = rel & 1;
                  const int ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2;
                  const int ax = dir ? tj : ti;
                  const bool fast_j = (flags & 2) != 0, masked = (flags & 1) != 0;
                  const bool is_fast = (dir == 1) == fast_j;
                  const int64_t stp = is_fast ? 1 : fn;
                  const int64_t base = pb + (int64_t)sb * nf2;
                  const double lo_d = sel3(ax, lo0, lo1, lo2), ih_d = sel3(ax, ih0, ih1, ih2);
                  // frame of the coupling moments in this direction: the shorter of the two box intervals
                  double loF_d = lo_d, ihF_d = ih_d;
                  if (nbr >= 0)
                    {
                      const double lq = sel3(ax, q0, q1, q2), iq = sel3(ax, i0, i1, i2);
                      if (iq > ih_d)
                        loF_d = lq, ihF_d = iq;
                    }
                  bool member = true;
                  if (masked) // boundary run of a corner polytope: sub-faces of the other planes do not count
                    member = P.ap_n[(int64_t)c * P.ap_stride + PDHR_IDX(base, P.ap_stride, 4)] * nsg > 0.5 &&
                             fabs(P.ap_x[(int64_t)c * P.ap_stride + PDHR_IDX(base, P.ap_stride, 5)] - xpl) <= 1e-9 / sel3(c, ih0, ih1, ih2);
                  const double sS = dir ? 1.0 / P.ap_wself[PDHR_IDX(base, P.ap_stride, 6)] : 1.0,
                               sC = dir ? 1.0 / P.ap_wcross[PDHR_IDX(base, P.ap_stride, 7)] : 1.0;
                  double ms[NA], mc[NA];
                  for (int a = 0; a < NA; ++a)
                    ms[a] = mc[a] = 0.0;
                  double Fd[FACT ? 16 : 1]; // FACT: X (dir 0) or Y (dir 1) of this sub-face, [k][l]
                  for (int i = 0; i < (FACT ? 16 : 1); ++i)
                    Fd[i] = 0.0;
                  const double loq_d = sel3(ax, q0, q1, q2), ihq_d = sel3(ax, i0, i1, i2);
                  // (all loads of the task first, like the volume tasks; fn <= 8)
                  double xr[8], wsr[8], wcr[8];
                  static_for<0, 8>([&](auto i_) {
                    constexpr int al = i_;
                    const int64_t q = PDHR_IDX(al < fn ? base + al * stp : 0, P.ap_stride, 8);
                    xr[al] = al < fn ? P.ap_x[(int64_t)ax * P.ap_stride + q] : 0.0;
                    wsr[al] = al < fn ? P.ap_wself[q] : 0.0;
                    wcr[al] = (al < fn && nbr >= 0) ? P.ap_wcross[q] : 0.0;
                  });
                  static_for<0, 8>([&](auto i_) {
                    constexpr int al = i_;
                    if (al < fn)
                      {
                        const double x = xr[al];
                        const double wS = member ? wsr[al] * sS : 0.0;
                        const double wC = (member && nbr >= 0) ? wcr[al] * sC : 0.0;
                        double Lx[NA];
                        pdhm::legendre01<NA>((x - lo_d) * ih_d, Lx);
                        for (int a = 0; a < NA; ++a)
                          ms[a] += wS * Lx[a];
                        if constexpr (FACT)
                          {
                            // both bases in the centred variable of their own box (pdh_basis.h: monomial coefficients,
                            // uniform: scalar operands)
                            const double zp = (x - lo_d) * ih_d - 0.5, zq = (x - loq_d) * ihq_d - 0.5;
                            double bp[4], bq[4];
                            for (int k = 0; k < 4; ++k)
                              {
                                double vp = P.tab.coef[k][3], vq = P.tab.coef[k][3];
                                for (int m = 2; m >= 0; --m)
                                  {
                                    vp = vp * zp + P.tab.coef[k][m];
                                    vq = vq * zq + P.tab.coef[k][m];
                                  }
                                bp[k] = wC * vp, bq[k] = vq;
                              }
                            for (int k = 0; k < 4; ++k)
                              for (int l = 0; l < 4; ++l)
                                Fd[k * 4 + l] += bp[k] * bq[l];
                          }
                        else
                          {
                            if (ihF_d != ih_d || loF_d != lo_d)
                              pdhm::legendre01<NA>((x - loF_d) * ihF_d, Lx);
                            for (int a = 0; a < NA; ++a)
                              mc[a] += wC * Lx[a];
                          }
                      }
                  });
                  for (int a = 0; a < NA; ++a)
                    {
                      mv[lane * RK::MVS + a] = ms[a];
                      if constexpr (!FACT)
                        mv[lane * RK::MVS + RK::MVC + a] = mc[a];
                    }
                  if constexpr (FACT)
                    if (nbr >= 0)
                      {
                        double *row = M2g + (size_t)(soff_src + sb) * 32 + dir * 16;
                        for (int i = 0; i < 16; ++i)
                          row[i] = Fd[i];
                      }
                }
              PDHR_ACC(tt_task);
            }
            PDH_WAVE_SYNC();
            PDHR_T0();
            // Per face, every lane forms what its (a0, a1) rows need straight from the task vectors (read-only now): no
            // staging of the 2-D moments, no hand-off inside the loop.  M2[a_i][a_j] = sum_sub m_i[a_i] m_j[a_j]; the rows
            // take L_{a_c}(zeta) M2 (expansion as in the general path below).
            int s0 = 0;
            if constexpr (!SMALL)
              {
                // FE_DGQ(3): the 2-D moments are staged through LDS (M2) and expanded like in the general path - fewer
                // VALU instructions per face than forming the rows from the task vectors, and this kind is the one whose
                // CU time is shared with 450 row stores per polytope (A/B: 1.3 % faster, half the scratch)
                double *M2 = W + 1024; // [3][8][8]
                for (int t = tb; t < te; ++t)
                  {
                    const FP fp = face_params(t);
                    const int ns = rl_i(my_nsub, t);
                    double G = 0.0, Gc = 0.0;
                    for (int sb = 0; sb < ns; ++sb)
                      {
                        G += mv[(s0 + 2 * sb) * RK::MVS + a0] * mv[(s0 + 2 * sb + 1) * RK::MVS + a1];
                        if constexpr (!FACT)
                          Gc += mv[(s0 + 2 * sb) * RK::MVS + RK::MVC + a0] * mv[(s0 + 2 * sb + 1) * RK::MVS + RK::MVC + a1];
                      }
                    s0 += 2 * ns;
                    if (act)
                      {
                        M2[0 * 64 + a0 * 8 + a1] = fp.sig * G;
                        M2[1 * 64 + a0 * 8 + a1] = -0.5 * fp.nsg * G;
                        if constexpr (!FACT)
                          M2[2 * 64 + a0 * 8 + a1] = Gc;
                      }
                    PDH_WAVE_SYNC();
                    expand(fp, M2);
                    const int fl = t - n_bdry;
                    if constexpr (!FACT)
                    if (fp.nbr >= 0 && fl >= 0 && fl < maxf)
                      {
                        if constexpr (MULTI)
                          M2g[fl * 64 + lane] = M2[2 * 64 + lane];
                        else
                          M2c[fl * MS + lane] = M2[2 * 64 + lane];
                      }
                    PDH_WAVE_SYNC();
                  }
              }
            else
            for (int t = tb; t < te; ++t)
              {
                const int c = rl_i(t_axis, t), nbr = rl_i(t_nbr, t), ns = rl_i(my_nsub, t);
                const double kS = rl_d(t_sigma, t), kN = -0.5 * rl_d(t_nsign, t);
                const double zeta = (rl_d(t_coord, t) - sel3(c, lo0, lo1, lo2)) * sel3(c, ih0, ih1, ih2);
                double Lc[NA];
                pdhm::legendre01<NA>(zeta, Lc);
                const double *mvt = mv + s0 * RK::MVS;
                if (c == 2)
                  {
                    double G = 0.0;
                    for (int sb = 0; sb < ns; ++sb)
                      G += mvt[(2 * sb) * RK::MVS + a0] * mvt[(2 * sb + 1) * RK::MVS + a1];
                    const double gS = kS * G, gN = kN * G;
                    for (int a = 0; a < NA; ++a)
                      {
                        accS[a] += Lc[a] * gS;
                        accN[2][a] += Lc[a] * gN;
                      }
                  }
                else
                  {
                    const int asel = c == 1 ? a1 : a0, arow = c == 1 ? a0 : a1;
                    double lc_ = Lc[0];
                    static_for<1, NA>([&](auto a_) {
                      constexpr int a = a_;
                      lc_ = asel == a ? Lc[a] : lc_;
                    });
                    double g[NA];
                    for (int a = 0; a < NA; ++a)
                      g[a] = 0.0;
                    for (int sb = 0; sb < ns; ++sb)
                      {
                        const double mi = mvt[(2 * sb) * RK::MVS + arow];
                        for (int a = 0; a < NA; ++a)
                          g[a] += mi * mvt[(2 * sb + 1) * RK::MVS + a];
                      }
                    const double fS = lc_ * kS, fN = lc_ * kN;
                    for (int a = 0; a < NA; ++a)
                      {
                        accS[a] += fS * g[a];
                        if (c == 1)
                          accN[1][a] += fN * g[a];
                        else
                          accN[0][a] += fN * g[a];
                      }
                  }
                const int fl = t - n_bdry;
                if (nbr >= 0 && fl >= 0 && fl < maxf)
                  {
                    double Gc = 0.0;
                    for (int sb = 0; sb < ns; ++sb)
                      Gc += mvt[(2 * sb) * RK::MVS + RK::MVC + a0] * mvt[(2 * sb + 1) * RK::MVS + RK::MVC + a1];
                    if (act)
                      M2c[fl * MS + a0 * 8 + a1] = Gc;
                  }
                s0 += 2 * ns;
              }
            PDHR_ACC(tt_face);
            tb = te;
          }
        PDH_WAVE_SYNC();
#ifdef PDHR_STAMP
        if (lane == 0 && Rw.stamps)
          {
            Rw.stamps[(int64_t)slot * 16 + 12] = tt_task;
            Rw.stamps[(int64_t)slot * 16 + 13] = tt_face;
          }
#endif
      }
    else if constexpr (GENERAL)
#if PDHR_EXP == 3
    if (nfaces > 0 && P.n < 0)
#else
    if (nfaces > 0)
#endif
      {
        // Chunk cursors: (face, pass, first point).  The point data of a chunk is requested DEPTH chunks ahead of its use
        // into a ring of register sets - a global load takes ~7k cycles under the store traffic of this kernel (in-kernel
        // stamps), a chunk ~3k.  Ring slots are addressed statically (the loop body is instantiated per slot): copying a
        // prefetched value into another variable would wait for the load.
        struct Cur
        {
          int t, pass;
          int64_t base;
          FP fp;
        };
        auto step = [&](const Cur &c0) {
          Cur n = c0;
          n.base = c0.base + FCH;
          if (n.base >= c0.fp.pe)
            {
              if (c0.pass + 1 < c0.fp.npass)
                n.pass = c0.pass + 1, n.base = c0.fp.pb;
              else
                {
                  n.t = c0.t + 1, n.pass = 0;
                  if (n.t < nfaces)
                    {
                      n.fp = face_params(n.t);
                      n.base = n.fp.pb;
                    }
                }
            }
          return n;
        };
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
#ifdef PDHR_STAMP
        long long tm_issue = 0, tm_rec = 0, tm_mfma = 0, tm_flush = 0;
#endif
        Cur cc, lc;
        cc.t = 0, cc.pass = 0;
        cc.fp = face_params(0);
        cc.base = cc.fp.pb;
        lc = cc;
        constexpr int DEPTH = 4;
        Raw ring[DEPTH];
        static_for<0, DEPTH>([&](auto d_) {
          constexpr int d = d_;
          ring[d] = ring[0];
          if (lc.t < nfaces)
            {
              ring[d] = issue(lc.fp, lc.base);
              lc = step(lc);
            }
        });
        auto chunk = [&](Raw &cur) {
            const FP fp = cc.fp;
            const int t = cc.t, pass = cc.pass;
            const bool last_chunk = cc.base + FCH >= fp.pe;
            // ---- consume this chunk's point data, THEN request a later chunk's into the same registers
            const int c = fp.c;
            const bool cur_on = cur.on;
            double xh, s0 = 0.0, s1 = 0.0, s2 = 0.0;
            {
              const double flo = half == 0 ? (pass ? fp.loF0 : fp.lo_t0) : (pass ? fp.loF1 : fp.lo_t1);
              const double fih = half == 0 ? (pass ? fp.ihF0 : fp.ih_t0) : (pass ? fp.ihF1 : fp.ih_t1);
              xh = (cur.x - flo) * fih;
              // a boundary run may hold the points of several planes (one entry per plane): the others get zero weights
              bool mine = half == 1 && cur.on;
              if (fp.masked)
                mine = mine && cur.n * fp.nsg > 0.5 && fabs(cur.xc - fp.xpl) <= fp.ptol;
              if (mine)
                {
                  if (pass == 0)
                    {
                      s0 = cur.wS * fp.sig;
                      s1 = -0.5 * cur.wS * fp.nsg;
                      s2 = (fp.nbr >= 0 && !fp.sep) ? cur.wC : 0.0;
                    }
                  else
                    s2 = cur.wC;
                }
            }
            {
              PDHR_T0();
              if (lc.t < nfaces)
                {
                  cur = issue(lc.fp, lc.base);
                  lc = step(lc);
                }
              PDHR_ACC(tm_issue);
            }
            // ---- record phase
            {
              PDHR_T0();
              double Lh[NA];
              pdhm::legendre01<NA>(cur_on ? xh : 0.5, Lh);
              double *r = W + pt * FREC;
              if (half == 0)
                {
                  for (int a = 0; a < NA; ++a)
                    r[a] = Lh[a];
                  r[NA] = 0.0;
                }
              else
                {
                  for (int a = 0; a < NA; ++a)
                    {
                      r[8 + a] = s0 * Lh[a];
                      r[16 + a] = s1 * Lh[a];
                      r[24 + a] = s2 * Lh[a];
                    }
                  r[8 + NA] = 0.0;
                  r[16 + NA] = 0.0;
                  r[24 + NA] = 0.0;
                }
              PDH_WAVE_SYNC();
              PDHR_ACC(tm_rec);
            }
            // ---- 8 steps of 4 points, operands one step ahead of the MFMAs (see MomentAcc::volume_chunk_full)
            {
              PDHR_T0();
              double ra[2], rb0[2], rb1[2], rb2[2];
              Acc::template lds_read<0>(ra[0], adA);
              Acc::template lds_read<0>(rb0[0], adB0);
              Acc::template lds_read<0>(rb1[0], adB1);
              Acc::template lds_read<0>(rb2[0], adB2);
              static_for<0, FCH / 4>([&](auto s_) {
                constexpr int s = s_;
                constexpr int cu = s & 1, nx = (s + 1) & 1;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[cu]), "+v"(rb0[cu]), "+v"(rb1[cu]), "+v"(rb2[cu]));
                if constexpr (s + 1 < FCH / 4)
                  {
                    Acc::template lds_read<(s + 1) * FSTEP>(ra[nx], adA);
                    Acc::template lds_read<(s + 1) * FSTEP>(rb0[nx], adB0);
                    Acc::template lds_read<(s + 1) * FSTEP>(rb1[nx], adB1);
                    Acc::template lds_read<(s + 1) * FSTEP>(rb2[nx], adB2);
                  }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = pdh::mfma4(ra[cu], rb0[cu], acc0);
                acc1 = pdh::mfma4(ra[cu], rb1[cu], acc1);
                acc2 = pdh::mfma4(ra[cu], rb2[cu], acc2);
                __builtin_amdgcn_sched_barrier(0);
              });
              PDH_WAVE_SYNC();
              PDHR_ACC(tm_mfma);
            }
            if (last_chunk)
              {
                PDHR_T0();
                // flush: D lane (i, blk, j) = rows a_i = 4 (blk & 1) + i; acc0: column block blk, acc1: block blk + 1, acc2: t = 2
                double *M2 = W; // [3][8][8]
                {
                  const int i = lane >> 4, j = lane & 3;
                  const int ai = 4 * (blk & 1) + i;
                  const int b1 = (blk + 1) & 3;
                  M2[(blk >> 1) * 64 + ai * 8 + 4 * (blk & 1) + j] = acc0;
                  M2[(b1 >> 1) * 64 + ai * 8 + 4 * (b1 & 1) + j] = acc1;
                  M2[2 * 64 + ai * 8 + 4 * ((blk == 1 || blk == 2) ? 1 : 0) + j] = acc2;
                }
                acc0 = acc1 = acc2 = 0.0;
                PDH_WAVE_SYNC();
                if (pass == 0)
                  expand(fp, M2);
                const int fl = t - n_bdry;
                if (fp.nbr >= 0 && (pass == 1 || !fp.sep) && fl >= 0 && fl < maxf)
                  {
                    if constexpr (MULTI)
                      M2g[fl * 64 + lane] = M2[2 * 64 + lane];
                    else
                      M2c[fl * MS + lane] = M2[2 * 64 + lane];
                  }
                PDH_WAVE_SYNC();
                PDHR_ACC(tm_flush);
              }
            cc = step(cc);
        };
        while (cc.t < nfaces)
          {
            chunk(ring[0]);
            if (cc.t >= nfaces)
              break;
            chunk(ring[1]);
            if (cc.t >= nfaces)
              break;
            chunk(ring[2]);
            if (cc.t >= nfaces)
              break;
            chunk(ring[3]);
          }
#ifdef PDHR_STAMP
        if (lane == 0 && Rw.stamps)
          {
            Rw.stamps[(int64_t)slot * 16 + 8] = tm_issue;
            Rw.stamps[(int64_t)slot * 16 + 9] = tm_rec;
            Rw.stamps[(int64_t)slot * 16 + 10] = tm_mfma;
            Rw.stamps[(int64_t)slot * 16 + 11] = tm_flush;
          }
#endif
      }
  }

  if constexpr (FACT)
    {
      // The factors were written by the task lanes of P2 and are read by OTHER lanes of this wave from here on (P3 follows at
      // once): global memory orders a work-item's own accesses only, so the wave releases its stores before it goes on
      // (vmcnt(0): they have reached L2).  [The moment form parks a value in the lane that reads it back: no fence needed.]
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  PDHR_MARK(3);
  // ---- S and C of one face into W (tables for the tangential directions, T, S) ----------------------------------------
  double *tabQ = W;               // [2][PAIRS][RS]
  double *Tst = W + 2 * M::LTAB;  // [16 pairs (k_j,l_j)][8]
  double *Sbuf = Tst + 128;       // [16 u][16 v]
  double *Sdst = Sbuf, *Cdst = Cbuf; // where build_S leaves S and C (MULTI keeps the planes of a neighbour side by side)
  // MULTI: this lane's value of an entry's coupling moments from the scratch row - entries are taken in ascending order nearly
  // always, so the value of the NEXT entry is requested whenever one is handed out (a load behind this kernel's own stores
  // takes thousands of cycles; one double per lane is what the head start costs.  Two entries ahead: no faster, A/B 3.01 vs
  // 3.05 ms - S and C of an entry cost ~6k cycles of arithmetic and hand-offs, not of waiting for this load)
  double m2_ahead = 0.0;
  int m2_ahead_fl = -1;
  auto moments_of = [&](int fl) { // (uniform)
    double v = m2_ahead;
    if (fl != m2_ahead_fl)
      v = M2g[fl * 64 + lane];
    const int nx = fl + 1;
    m2_ahead_fl = -1;
    if (nx < nfaces - n_bdry && nx < maxf)
      {
        m2_ahead = M2g[nx * 64 + lane];
        m2_ahead_fl = nx;
      }
    return v;
  };
  // FACT: the first 64 doubles (two sub-faces) of the NEXT entry's factors are requested when an entry is built
  double fa_v = 0.0;
  int fa_t = -1;
  auto build_S = [&](int t) { // t = local face index
    const int fl = t - n_bdry;
    const int c = rl_i(t_axis, t);
    const int ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2;
    const double q0 = rl_d(t_qlo0, t), q1 = rl_d(t_qlo1, t), q2 = rl_d(t_qlo2, t);
    const double i0 = rl_d(t_qih0, t), i1 = rl_d(t_qih1, t), i2 = rl_d(t_qih2, t);
    if constexpr (FACT)
      {
        // S[(k_i,k_j),(l_i,l_j)] = sum over the entry's sub-faces of X[k_i,l_i] Y[k_j,l_j]: the factors (P2) are staged from the
        // scratch row through W[0, 448) (14 sub-faces at a time), every lane forms its four entries of S
        const int ns = rl_i(t_nsub, t), sf0 = rl_i(t_soff, t);
        double v0 = fa_v;
        if (t != fa_t)
          v0 = M2g[(size_t)sf0 * 32 + lane];
        fa_t = -1;
        if (t + 1 < nfaces)
          {
            fa_v = M2g[(size_t)rl_i(t_soff, t + 1) * 32 + lane];
            fa_t = t + 1;
          }
        double *F = W;
        const int pairI = lane & 15, ki = pairI >> 2, li = pairI & 3;
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int sb0 = 0; sb0 < ns; sb0 += 14)
          {
            const int nsc = ns - sb0 < 14 ? ns - sb0 : 14;
            if (sb0 > 0)
              PDH_WAVE_SYNC(); // (the chunk before has been read)
            if (lane < nsc * 32)
              F[lane] = sb0 == 0 ? v0 : M2g[(size_t)(sf0 + sb0) * 32 + lane];
            for (int r = lane + 64; r < nsc * 32; r += 64)
              F[r] = M2g[(size_t)(sf0 + sb0) * 32 + r];
            PDH_WAVE_SYNC();
            for (int sb = 0; sb < nsc; ++sb)
              {
                const double x = F[sb * 32 + ki * 4 + li];
                static_for<0, 4>([&](auto r_) {
                  constexpr int r = r_;
                  const int pairJ = (lane >> 4) + 4 * r, kj = pairJ >> 2, lj = pairJ & 3;
                  sacc[r] += x * F[sb * 32 + 16 + kj * 4 + lj];
                });
              }
          }
        static_for<0, 4>([&](auto r_) {
          constexpr int r = r_;
          const int pairJ = (lane >> 4) + 4 * r, kj = pairJ >> 2, lj = pairJ & 3;
          Sdst[(ki + 4 * kj) * 16 + li + 4 * lj] = sacc[r];
        });
      }
    else
    {
    if constexpr (MULTI)
      M2c[lane] = moments_of(fl); // staging slot (read in the T stage below, behind the first hand-off)
    bool same[2];
    {
      const int dd = (lane >> 4) & 1;
      const int d = dd == 0 ? ti : tj;
      const double lo_d = sel3(d, lo0, lo1, lo2), ih_d = sel3(d, ih0, ih1, ih2);
      const double loq_d = sel3(d, q0, q1, q2), ihq_d = sel3(d, i0, i1, i2);
      const bool same_d = (loq_d == lo_d) && (ihq_d == ih_d);
      same[0] = __shfl(same_d ? 1 : 0, 0) != 0;
      same[1] = __shfl(same_d ? 1 : 0, 16) != 0;
      PDH_WAVE_SYNC();
      if (lane < 32 && !same_d)
        {
          // E^Q_d[k,l,a] = <B_k(xi^P(t)) B_l(xi^Q(t)), L_a>, frame = the shorter interval (pdh_moment.h, k_moffdiag)
          const bool use_q = ihq_d > ih_d;
          const double loF = use_q ? loq_d : lo_d, ihF = use_q ? ihq_d : ih_d;
          const double hF = 1.0 / ihF;
          const double aP = hF * ih_d, bP = (loF - lo_d) * ih_d - 0.5;
          const double aQ = hF * ihq_d, bQ = (loF - loq_d) * ihq_d - 0.5;
          const int k = (lane >> 2) & 3, l = lane & 3;
          double e[NA];
          for (int a = 0; a < NA; ++a)
            e[a] = 0.0;
          for (int gq = 0; gq < NG; ++gq)
            {
              const double tg = mt[M::OFF_GX + gq];
              const double xp = aP * tg + bP, xq = aQ * tg + bQ;
              double vk = coefL[k * 4 + 3], vl = coefL[l * 4 + 3];
              for (int m = 2; m >= 0; --m)
                {
                  vk = vk * xp + coefL[k * 4 + m];
                  vl = vl * xq + coefL[l * 4 + m];
                }
              const double vv = vk * vl;
              for (int a = 0; a < NA; ++a)
                e[a] += vv * mt[M::OFF_GL + a * NG + gq];
            }
          if (k < N1D && l < N1D)
            {
              double *te = tabQ + dd * M::LTAB + (k * N1D + l) * M::RS;
              for (int a = 0; a < NA; ++a)
                te[a] = e[a];
              te[NA] = 0.0;
            }
        }
      PDH_WAVE_SYNC();
    }
    const double *EQi = same[0] ? tabE : tabQ, *EQj = same[1] ? tabE : (tabQ + M::LTAB);
    {
      // T[(k_j,l_j)][alpha] = sum_beta EQj[(k_j,l_j)][beta] M2c[alpha][beta]
      // (lanes are grouped by 16 = 4 x 4 digit pairs whatever N1D is; pairs with a digit >= N1D are idle)
      const int pair = lane & 15;
      const int kj_ = pair >> 2, lj_ = pair & 3;
      const int pairT = (kj_ < N1D && lj_ < N1D) ? kj_ * N1D + lj_ : 0;
      const double *m2 = MULTI ? M2c : M2c + fl * MS;
      double ej[NA];
      for (int b = 0; b < NA; ++b)
        ej[b] = EQj[pairT * M::RS + b];
      static_for<0, 2>([&](auto h_) {
        constexpr int h = h_;
        const int al = (lane >> 4) + 4 * h;
        if (al < NA)
          {
            double s = 0.0;
            for (int b = 0; b < NA; ++b)
              s += ej[b] * m2[al * 8 + b];
            Tst[pair * 8 + al] = s;
          }
      });
    }
    PDH_WAVE_SYNC();
    {
      // S[k_i + 4 k_j][l_i + 4 l_j] = sum_alpha EQi[(k_i,l_i)][alpha] T[(k_j,l_j)][alpha]
      const int pairI = lane & 15, ki = pairI >> 2, li = pairI & 3;
      const int pairIT = (ki < N1D && li < N1D) ? ki * N1D + li : 0;
      double ei[NA];
      for (int a = 0; a < NA; ++a)
        ei[a] = EQi[pairIT * M::RS + a];
      static_for<0, 4>([&](auto r_) {
        constexpr int r = r_;
        const int pairJ = (lane >> 4) + 4 * r, kj = pairJ >> 2, lj = pairJ & 3;
        double s = 0.0;
        for (int a = 0; a < NA; ++a)
          s += ei[a] * Tst[pairJ * 8 + a];
        if constexpr (!SMALL)
          Sdst[(ki + 4 * kj) * 16 + li + 4 * lj] = s;
        else
          { // compact S over the face's moment slot (read for the last time in the T stage)
            const bool okk = ki < N1D && kj < N1D && (BASIS == 0 || ki + kj < N1D);
            const bool okl = li < N1D && lj < N1D && (BASIS == 0 || li + lj < N1D);
            if (okk && okl)
              M2c[fl * MS + RK::pair(ki, kj) * NS + RK::pair(li, lj)] = s;
          }
      });
    }
    } // (moment form of S)
    if (lane < 16)
      {
        // C[k][l] = (1/2 s B'_k(zP)/hP - sigma B_k(zP)) B_l(zQ) - 1/2 s B_k(zP) B'_l(zQ)/hQ    (centred variable)
        const double x = rl_d(t_coord, t);
        const double lo_c = sel3(c, lo0, lo1, lo2), ih_c = sel3(c, ih0, ih1, ih2);
        const double loq_c = sel3(c, q0, q1, q2), ihq_c = sel3(c, i0, i1, i2);
        const double zp = (x - lo_c) * ih_c - 0.5, zq = (x - loq_c) * ihq_c - 0.5;
        const int k = lane >> 2, l = lane & 3;
        double vk = coefL[k * 4 + 3], dk = 0.0, vl = coefL[l * 4 + 3], dl = 0.0;
        for (int m = 2; m >= 0; --m)
          {
            dk = dk * zp + vk;
            vk = vk * zp + coefL[k * 4 + m];
            dl = dl * zq + vl;
            vl = vl * zq + coefL[l * 4 + m];
          }
        const double sg = rl_d(t_nsign, t), sig = rl_d(t_sigma, t);
        const double cv = (0.5 * sg * dk * ih_c - sig * vk) * vl - 0.5 * sg * vk * dl * ihq_c;
        if constexpr (!SMALL)
          Cdst[lane] = cv;
        else
          Call[fl * 16 + lane] = cv;
      }
    PDH_WAVE_SYNC();
  };
  // last column (63: l = (3,3,3)) of the block just built, row R = lane
  auto last_column = [&](int c) { return Cdst[digit_c(lane, c) * 4 + 3] * Sdst[digits_t(lane, c) * 16 + 15]; };

  // ================= P3: carry into the own block's piece ==========================================================
  constexpr bool shifted = SHIFTED;
  double carry_own = 0.0;
  if constexpr (SMALL)
    { // streamed kinds: S and C of every interior face now (P4 takes W over), kept until the rows are streamed out
      // Uniform neighbourhoods (every neighbour's box agrees with the own one in both tangential directions of the shared
      // face - block agglomerates): no per-face tables are needed, and the two contraction stages run for ALL faces in
      // lock-step (two hand-offs per polytope instead of five per face).
      bool all_same = nfaces - n_bdry <= MAXF;
      for (int t = n_bdry; t < nfaces; ++t)
        {
          const int c = rl_i(t_axis, t);
          const int ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2;
          const double q0 = rl_d(t_qlo0, t), q1 = rl_d(t_qlo1, t), q2 = rl_d(t_qlo2, t);
          const double i0 = rl_d(t_qih0, t), i1 = rl_d(t_qih1, t), i2 = rl_d(t_qih2, t);
          all_same = all_same && sel3(ti, q0, q1, q2) == sel3(ti, lo0, lo1, lo2) && sel3(ti, i0, i1, i2) == sel3(ti, ih0, ih1, ih2) &&
                     sel3(tj, q0, q1, q2) == sel3(tj, lo0, lo1, lo2) && sel3(tj, i0, i1, i2) == sel3(tj, ih0, ih1, ih2);
        }
      if (all_same)
        {
          const int nint = nfaces - n_bdry;
          PDH_WAVE_SYNC();
          {
            // T_f[(k_j,l_j)][alpha] = sum_beta E[(k_j,l_j)][beta] M2c_f[alpha][beta], every face; T_f at W + 128 f
            const int pair = lane & 15, kj_ = pair >> 2, lj_ = pair & 3;
            const int pairT = (kj_ < N1D && lj_ < N1D) ? kj_ * N1D + lj_ : 0;
            double ej[NA];
            for (int b = 0; b < NA; ++b)
              ej[b] = tabE[pairT * M::RS + b];
            for (int f = 0; f < nint; ++f)
              {
                const double *m2 = M2c + f * MS;
                static_for<0, 2>([&](auto h_) {
                  constexpr int h = h_;
                  const int al = (lane >> 4) + 4 * h;
                  if (al < NA)
                    {
                      double sm = 0.0;
                      for (int b = 0; b < NA; ++b)
                        sm += ej[b] * m2[al * 8 + b];
                      W[f * 128 + pair * 8 + al] = sm;
                    }
                });
              }
          }
          PDH_WAVE_SYNC();
          {
            const int pairI = lane & 15, ki = pairI >> 2, li = pairI & 3;
            const int pairIT = (ki < N1D && li < N1D) ? ki * N1D + li : 0;
            double ei[NA];
            for (int a = 0; a < NA; ++a)
              ei[a] = tabE[pairIT * M::RS + a];
            for (int f = 0; f < nint; ++f)
              static_for<0, 4>([&](auto r_) {
                constexpr int r = r_;
                const int pairJ = (lane >> 4) + 4 * r, kj = pairJ >> 2, lj = pairJ & 3;
                double sm = 0.0;
                for (int a = 0; a < NA; ++a)
                  sm += ei[a] * W[f * 128 + pairJ * 8 + a];
                const bool okk = ki < N1D && kj < N1D && (BASIS == 0 || ki + kj < N1D);
                const bool okl = li < N1D && lj < N1D && (BASIS == 0 || li + lj < N1D);
                if (okk && okl)
                  M2c[f * MS + RK::pair(ki, kj) * NS + RK::pair(li, lj)] = sm;
              });
            // C of every face: lanes = (face, k, l)
            for (int r = 0; r < 2; ++r)
              {
                const int id = lane + 64 * r, f = id >> 4, kl = id & 15;
                const int src = n_bdry + (f < nint ? f : 0);
                const int c = __shfl(t_axis, src);
                const double x = __shfl(t_coord, src), sg = __shfl(t_nsign, src), sig = __shfl(t_sigma, src);
                const double q0 = __shfl(t_qlo0, src), q1 = __shfl(t_qlo1, src), q2 = __shfl(t_qlo2, src);
                const double i0 = __shfl(t_qih0, src), i1 = __shfl(t_qih1, src), i2 = __shfl(t_qih2, src);
                if (f < nint)
                  {
                    const double lo_c = sel3(c, lo0, lo1, lo2), ih_c = sel3(c, ih0, ih1, ih2);
                    const double loq_c = sel3(c, q0, q1, q2), ihq_c = sel3(c, i0, i1, i2);
                    const double zp = (x - lo_c) * ih_c - 0.5, zq = (x - loq_c) * ihq_c - 0.5;
                    const int k = kl >> 2, l = kl & 3;
                    double vk = coefL[k * 4 + 3], dk = 0.0, vl = coefL[l * 4 + 3], dl = 0.0;
                    for (int m = 2; m >= 0; --m)
                      {
                        dk = dk * zp + vk;
                        vk = vk * zp + coefL[k * 4 + m];
                        dl = dl * zq + vl;
                        vl = vl * zq + coefL[l * 4 + m];
                      }
                    Call[f * 16 + kl] = (0.5 * sg * dk * ih_c - sig * vk) * vl - 0.5 * sg * vk * dl * ihq_c;
                  }
              }
          }
          PDH_WAVE_SYNC();
        }
      else
      for (int t = n_bdry; t < nfaces; ++t)
        {
          PDH_WAVE_SYNC();
          build_S(t);
        }
    }
  else if (shifted && m0 > 0)
    for (int t = n_bdry; t < nfaces; ++t)
      if (rl_i(t_blk, t) == m0 - 1)
        {
          PDH_WAVE_SYNC();
          build_S(t);
          // (MULTI: the block left of the own one may be the sum over several planes)
          carry_own = (MULTI ? carry_own : 0.0) + last_column(rl_i(t_axis, t));
        }

  // ================= coupling blocks (FE_DGQ(3), one plane per neighbour): P5 of the phase list ===========================
  // LEFT_PASS = false: the blocks right of the diagonal (every block in the ascending layout) - plain aligned pieces, no
  // dependence on anything but their own face: written HERE, before the diagonal block (PDHR_SPLIT).  LEFT_PASS = true: the
  // blocks left of the diagonal, whose pieces start with the last column of the block before (the first one with the
  // diagonal entry, which P4 leaves in diagv): after P4.
  auto coupling_blocks = [&](auto left_pass_) {
    constexpr bool LEFT_PASS = left_pass_;
    // (the ascending layout has aligned pieces only: one pass, after the diagonal block - before it the moment accumulators
    // are still live and the pass would spill)
    constexpr bool SPLIT = PDHR_SPLIT && SHIFTED;
    double carry = 0.0; // lane R: the value that lane 0 stores in row R of the next piece
    bool first_left = true;
#if PDHR_EXP == 1
    for (int t = n_bdry; t < nfaces && P.n < 0; ++t)
#else
    for (int t = n_bdry; t < nfaces; ++t)
#endif
      {
        const int b = rl_i(t_blk, t);
        const int c = rl_i(t_axis, t);
        const bool left = shifted && b < m0;
        if (SPLIT && left != LEFT_PASS)
          continue;
        PDH_WAVE_SYNC();
        const bool first_piece = left && first_left; // piece 0 starts with the diagonal entry (diagv, written in P4)
        if (left)
          first_left = false;
        build_S(t);
        // this lane's column of the block: shifted pieces hold columns -1 .. 62 (lane 0: the carry)
        const int jcol = left ? (lane > 0 ? lane - 1 : 0) : lane;
        const int lc = digit_c(jcol, c), vt = digits_t(jcol, c);
        double Cl[4], sc[16];
        for (int k = 0; k < 4; ++k)
          Cl[k] = Cbuf[k * 4 + lc];
        for (int u = 0; u < 16; ++u)
          sc[u] = Sbuf[u * 16 + vt];
        const double next_carry = left ? last_column(c) : 0.0;
        // rows: value = Cl[k_c(R)] * sc[u(R)].  Shifted pieces: lane 0 stores the carry of row R instead.  The carries of all
        // rows (lane R = row R: `carry` of the block before, or the diagonal entries P4 left in diagv for piece 0) stand in
        // LDS; eight rows at a time, lane 0 ALONE reads its eight carries straight into the product registers (EXEC = 1 around
        // the eight ds_read: the other lanes keep their products), one wait, eight stores.  No VALU instruction per row but
        // the multiply - the carries used to travel by v_readlane + v_writelane pairs, four VALU instructions per row, 768 per
        // block-shaped polytope (an eighth of all the kernel's VALU work).  The row address is a scalar offset + lane.
        double *ctab = W + 1536; // [64] (behind the scratch of build_S)
        if (left && !first_piece)
          ctab[lane] = carry;
        const double *csrc = first_piece ? diagv : ctab; // uniform
        typedef __attribute__((address_space(3))) const char lds_cchar;
        const unsigned caddr = (unsigned)(uintptr_t)(lds_cchar *)reinterpret_cast<const char *>(csrc);
        if (left)
          PDH_WAVE_SYNC();
        const uint32_t rowp = 64u * 8u * (uint32_t)b; // uniform: byte offset of the piece in row 0
        const uint32_t lane_off = (uint32_t)lane * 8u;
        auto rows = [&, lane_off](auto c_, auto left_) {
          constexpr int cc = c_;
          constexpr bool LEFT = left_;
          const uint32_t loff = lane_off;
          uint32_t rowrun = rowp;
          static_for<0, 8>([&](auto g_) {
            constexpr int g = g_;
            double v[8];
            static_for<0, 8>([&](auto r_) {
              constexpr int R = 8 * g + r_;
              constexpr int kc = (R >> (2 * cc)) & 3;
              constexpr int k0 = R & 3, k1 = (R >> 2) & 3, k2 = (R >> 4) & 3;
              constexpr int u = cc == 0 ? (k1 + 4 * k2) : (cc == 1 ? (k0 + 4 * k2) : (k0 + 4 * k1));
              v[r_] = Cl[kc] * sc[u];
            });
            if constexpr (LEFT)
              PDHR_CARRY_READ8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], caddr, g); // (one self-contained statement)
            static_for<0, 8>([&](auto r_) {
#if PDHR_EXP == 5
              if (P.n < 0)
#endif
                row_store(v[r_], loff, rowrun); // (a running offset: 64 precomputed row offsets would be spilled scalars)
              rowrun += (uint32_t)rlen * 8u;
            });
          });
        };
        using std::integral_constant;
        // (with PDHR_SPLIT a pass meets one kind of piece only: the other kind's code is not instantiated for it)
        if (left)
          {
            if constexpr (LEFT_PASS || !SPLIT)
              {
                if (c == 0)
                  rows(integral_constant<int, 0>{}, std::true_type{});
                else if (c == 1)
                  rows(integral_constant<int, 1>{}, std::true_type{});
                else
                  rows(integral_constant<int, 2>{}, std::true_type{});
              }
          }
        else
          {
            if constexpr (!LEFT_PASS || !SPLIT)
              {
                if (c == 0)
                  rows(integral_constant<int, 0>{}, std::false_type{});
                else if (c == 1)
                  rows(integral_constant<int, 1>{}, std::false_type{});
                else
                  rows(integral_constant<int, 2>{}, std::false_type{});
              }
          }
        PDH_WAVE_SYNC();
        carry = next_carry;
      }
  };
  if constexpr (!SMALL && !MULTI && PDHR_SPLIT && SHIFTED)
    coupling_blocks(std::false_type{});

  m2_ahead_fl = -1; // (MULTI: a value requested ahead by P3 is not carried through the diagonal block)
  fa_t = -1;
  PDHR_MARK(4);
  // ================= P4: diagonal block ============================================================================
  PDH_WAVE_SYNC();
  if constexpr (GENERAL)
  if (tn == 0)
  {
  {
    // volume accumulators -> W[(a0,a1) row][a2] (scatter of MomentAcc, volume part)
    const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
    if constexpr (Acc::LAST_ROW)
      if (j == 0 && blk < 2 && 4 * blk + i < NA)
        W[(Acc::ROWS - 1) * NA + 4 * blk + i] = ma.accyv;
    static_for<0, Acc::NFAM>([&](auto a_) {
      constexpr int a = a_;
      const int row = 16 * a + 4 * blk + i;
      if (row < Acc::ROWS)
        static_for<0, 2>([&](auto r_) {
          constexpr int r = r_;
          const int a2 = 4 * ((blk + r) & 1) + j;
          if (a2 < NA)
            W[row * NA + a2] = ma.accv[a][r];
        });
    });
  }
  PDH_WAVE_SYNC();
  for (int a = 0; a < NA; ++a)
    accM[a] = W[(act ? lane : 0) * NA + a];
  }
  if (P.reaction_c != 0.0)
    for (int a = 0; a < NA; ++a)
      accS[a] += P.reaction_c * accM[a];
  if constexpr (SMALL)
    {
      // streamed kinds: n x n entries - the same three-stage sum factorisation on the VALU, restricted to the index set of the
      // basis on both sides, slab by slab over k2.  Stage 1 (a2, in registers) -> T1[g1 | ee | n0 | n1][l2][a0,a1]; stage 2 (a1):
      // work items (k1, (l1,l2), a0) -> T2[X = D | E | Fs][k1][(l1,l2)][a0] with the h factors folded in; stage 3 (a0):
      // work items (row (k0,k1), column j) -> the LDS copy of the block.
      double *T1 = W, *T2 = W + RK::T1;
      constexpr int NA2 = NA * NA, NC = RK::NC;
      // stage-2 work items of this lane, one per round of 64: item q = (k1, pair (l1,l2), a0) does not depend on the slab -
      // decoded once (offsets of its T1 rows, its table row and its T2 entry)
      // (item q = (k1 - first k1 of the pass, pair (l1,l2), a0): the table row of the pass adds k1b N1D RS)
      constexpr int K1P = RK::K1P;
      constexpr int S2R = (K1P * NS * NA + 63) / 64;
      int s2_r0[S2R], s2_tp[S2R], s2_out[S2R];
      static_for<0, S2R>([&](auto rr_) {
        constexpr int rr = rr_;
        const int q = lane + 64 * rr;
        const int cc = q / NA, a0_ = q - NA * cc;
        const int k1 = cc / NS, lp = cc - NS * k1; // lp = RK::pair(l1, l2)
        int l1, l2;
        if constexpr (BASIS == 0)
          l2 = lp / N1D, l1 = lp - N1D * l2;
        else
          {
            l2 = 0, l1 = lp;
            for (int w_ = N1D; l1 >= w_ && w_ > 0; --w_)
              l1 -= w_, ++l2;
          }
        const bool ok = k1 < K1P && l1 < N1D && l2 < N1D;
        s2_tp[rr] = ok ? (k1 * N1D + l1) * M::RS : 0;
        s2_r0[rr] = ok ? l2 * NA2 + a0_ * NA : 0;
        s2_out[rr] = ok ? cc * NA + a0_ : 0;
      });
      const double s00 = ih0 * ih0, s11 = ih1 * ih1, s22 = ih2 * ih2;
#ifdef PDHR_STAMP
      long long tp1 = 0, tp2 = 0, tp3 = 0;
#endif
#pragma unroll 1
#if PDHR_EXP == 2
      for (int k2 = 0; k2 < N1D && P.n < 0; ++k2)
#else
      for (int k2 = 0; k2 < N1D; ++k2)
#endif
        {
          PDH_WAVE_SYNC();
          PDHR_T0();
          if (act)
            static_for<0, N1D>([&](auto ll_) {
              constexpr int ll = ll_;
              // (uniform table rows: scalar loads from the global table, not LDS broadcasts)
              const double *tE = mt + M::OFF_E + (k2 * N1D + ll) * NAP, *tD = mt + M::OFF_D + (k2 * N1D + ll) * NAP,
                           *tF = mt + M::OFF_FS + (k2 * N1D + ll) * NAP;
              double g1 = 0.0, eD = 0.0, eE = 0.0, eF = 0.0, n0 = 0.0, n1 = 0.0;
              static_for<0, NA>([&](auto a_) {
                constexpr int a = a_;
                const double e = tE[a], d = tD[a], f = tF[a];
                g1 += e * accM[a];
                eD += d * accM[a];
                eE += e * accS[a];
                eF += f * accN[2][a];
                n0 += e * accN[0][a];
                n1 += e * accN[1][a];
              });
              T1[(0 * N1D + ll) * NA2 + lane] = g1;
              T1[(1 * N1D + ll) * NA2 + lane] = s22 * eD + eE + ih2 * eF;
              T1[(2 * N1D + ll) * NA2 + lane] = n0;
              T1[(3 * N1D + ll) * NA2 + lane] = n1;
            });
          PDH_WAVE_SYNC();
          PDHR_ACC(tp1);
          const int nk1 = BASIS == 0 ? N1D : N1D - k2; // values of k1 (and, below, rows (k0, k1)) of this slab
#pragma unroll 1
          for (int k1b = 0; k1b < nk1; k1b += K1P)
            {
              const int npk = nk1 - k1b < K1P ? nk1 - k1b : K1P; // values of k1 in this pass
              if (k1b > 0)
                PDH_WAVE_SYNC(); // (stage 3 of the pass before has read T2)
              const long long t1_ = (long long)__builtin_readcyclecounter();
              const int tpb = k1b * N1D * M::RS;
              static_for<0, S2R>([&](auto rr_) {
                constexpr int rr = rr_;
                if (lane + 64 * rr < npk * NS * NA)
                {
                  const int tp = s2_tp[rr] + tpb, cc_a = s2_out[rr];
                  const double *r0 = T1 + s2_r0[rr], *r1 = r0 + N1D * NA2, *r2 = r0 + 2 * N1D * NA2, *r3 = r0 + 3 * N1D * NA2;
                  double sD = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, sF = 0.0;
                  static_for<0, NA>([&](auto a_) {
                    constexpr int a = a_;
                    const double e = tabE[tp + a], d = tabD[tp + a], f = tabF[tp + a], g = r0[a];
                    sD += e * g;
                    s1 += d * g;
                    s2 += e * r1[a];
                    s3 += f * r3[a];
                    sF += e * r2[a];
                  });
                  T2[0 * NC * NA + cc_a] = s00 * sD;
                  T2[1 * NC * NA + cc_a] = s11 * s1 + s2 + ih1 * s3;
                  T2[2 * NC * NA + cc_a] = ih0 * sF;
                }
              });
              PDH_WAVE_SYNC();
              const long long t2_ = (long long)__builtin_readcyclecounter();
#ifdef PDHR_STAMP
              tp2 += t2_ - t1_;
#endif
              // rows (k0, k1) of the slab with k1 in [k1b, k1b + npk): k1 outer, k0 inner (k0 < nk1 - k1 for the P_p basis)
              const int rb = BASIS == 0 ? k1b * N1D : k1b * nk1 - k1b * (k1b - 1) / 2;
              const int k1e = k1b + npk;
              const int re = BASIS == 0 ? k1e * N1D : k1e * nk1 - k1e * (k1e - 1) / 2;
              for (int q = lane + rb * NF; q < re * NF; q += 64)
                {
                  const int rr = q / NF, j = q - NF * rr;
                  int k1, k0;
                  if constexpr (BASIS == 0)
                    k1 = rr / N1D, k0 = rr - N1D * k1;
                  else
                    {
                      k1 = 0, k0 = rr;
                      for (int w_ = nk1; k0 >= w_; --w_)
                        k0 -= w_, ++k1;
                    }
                  const int dj_ = dig[j]; // axis 0: l0 | pair(l1, l2) << 4
                  const int tp = (k0 * N1D + (dj_ & 15)) * M::RS;
                  const double *t2 = T2 + ((k1 - k1b) * NS + (dj_ >> 4)) * NA;
                  double sv = 0.0;
                  static_for<0, NA>([&](auto a_) {
                    constexpr int a = a_;
                    sv += tabD[tp + a] * t2[a] + tabE[tp + a] * t2[NC * NA + a] + tabF[tp + a] * t2[2 * NC * NA + a];
                  });
                  Dblk[RK::findex(k0, k1, k2) * NF + j] = sv;
                }
#ifdef PDHR_STAMP
              tp3 += (long long)__builtin_readcyclecounter() - t2_;
#endif
            }
        }
#ifdef PDHR_STAMP
      if (lane == 0 && Rw.stamps)
        {
          Rw.stamps[(int64_t)slot * 16 + 7] = tp1; // (slots 8 .. 13 belong to P2)
          Rw.stamps[(int64_t)slot * 16 + 14] = tp2;
          Rw.stamps[(int64_t)slot * 16 + 15] = tp3;
        }
#endif
    }
  else
  {
    double *T1B = W;                  // [4 arrays][2 bf][2 ks][64]
    double *T2B = W + 4 * 2 * 2 * 64; // [2 ks][4 cf][64]
    // (stage 3 runs with the operand roles swapped, pdh_moment.h: mstage3_t - a result register D3[cf][s0] is row
    // s0 + 4 cf + 16 k2 of the block with lane l holding column l, stored as it stands)
    double dg = 0.0; // lane R: the diagonal entry A[R,R] (diagonal-first rows: kept for piece 0 of the row, P5)
    pdhm::T1Off t1o;
    t1o.init(a0, a1);
    // The K = 8 fragments of stage 2 read an eighth Legendre mode and an eighth (a0) row that do not exist: zero pads (row 7,
    // column 7 of every [a0][a1] plane of T1B).  Nothing but stage 1 writes T1B between here and the end of P4, and stage 1
    // writes the 49 values only: the pads are set ONCE per polytope - all 1024 doubles are cleared, 16 stores per lane, where
    // the version that re-wrote its pads with every value spent three exec-masked stores per value (64 values per lane).
    PDH_WAVE_SYNC();
    for (int k = 0; k < 16; ++k)
      T1B[k * 64 + lane] = 0.0;
#ifdef PDHR_STAMP
    long long tq1 = 0, tq2 = 0, tq3 = 0;
#endif
#pragma unroll 1
#if PDHR_EXP == 2
    for (int k2 = 0; k2 < 4 && P.n < 0; ++k2)
#else
    for (int k2 = 0; k2 < 4; ++k2)
#endif
      {
        PDH_WAVE_SYNC();
#ifdef PDHR_STAMP
        const long long ts0_ = (long long)__builtin_readcyclecounter();
#endif
        if (act)
          static_for<0, 4>([&](auto ll_) {
            constexpr int ll = ll_;
            // (uniform table rows: scalar loads from the global table, not LDS broadcasts)
            const double *tE = mt + M::OFF_E + (k2 * 4 + ll) * NAP, *tD = mt + M::OFF_D + (k2 * 4 + ll) * NAP,
                         *tF = mt + M::OFF_FS + (k2 * 4 + ll) * NAP;
            double g1 = 0.0, ee = 0.0, n0 = 0.0, n1 = 0.0;
            static_for<0, NA>([&](auto a_) {
              constexpr int a = a_;
              const double e = tE[a];
              g1 += e * accM[a];
              ee += (tD[a] * (ih2 * ih2)) * accM[a] + e * accS[a] + (tF[a] * ih2) * accN[2][a];
              n0 += e * accN[0][a];
              n1 += e * accN[1][a];
            });
            // (values only: the zero pads of T1B were written once for this polytope, below)
            T1B[(0 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = g1;
            T1B[(1 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = ee;
            T1B[(2 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = n0;
            T1B[(3 * 256 + (ll >> 1) * 128) + t1o.val[ll & 1]] = n1;
          });
#ifdef PDHR_STAMP
        const long long ts1_ = (long long)__builtin_readcyclecounter();
        tq1 += ts1_ - ts0_;
#endif
        int zero = 0;
        asm volatile("" : "+s"(zero));
        pdhm::ASet AE, AD, AF;
        pdhm::load_aset<false>(tabE + zero, M::RS, lane, AE);
        pdhm::load_aset<false>(tabD + zero, M::RS, lane, AD);
        pdhm::load_aset<false>(tabF + zero, M::RS, lane, AF);
        double D3[4][4];
        for (int c = 0; c < 4; ++c)
          for (int r = 0; r < 4; ++r)
            D3[c][r] = 0.0;
        {
          PDH_WAVE_SYNC();
          double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = D
          pdhm::mstage2_term(AE, 0, ih0 * ih0, T1B, lane, D2);
          pdhm::mstage2_scatter_t(D2, T2B, lane);
          PDH_WAVE_SYNC();
          pdhm::mstage3_t(AD, T2B, lane, D3);
        }
        {
          PDH_WAVE_SYNC();
          double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = E
          pdhm::mstage2_term(AD, 0, ih1 * ih1, T1B, lane, D2);
          pdhm::mstage2_term(AE, 1, 1.0, T1B, lane, D2);
          pdhm::mstage2_term(AF, 3, ih1, T1B, lane, D2);
          pdhm::mstage2_scatter_t(D2, T2B, lane);
          PDH_WAVE_SYNC();
          pdhm::mstage3_t(AE, T2B, lane, D3);
        }
        {
          PDH_WAVE_SYNC();
          double D2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; // X = Fs
          pdhm::mstage2_term(AE, 2, ih0, T1B, lane, D2);
          pdhm::mstage2_scatter_t(D2, T2B, lane);
          PDH_WAVE_SYNC();
          pdhm::mstage3_t(AF, T2B, lane, D3);
        }
#ifdef PDHR_STAMP
        const long long ts2_ = (long long)__builtin_readcyclecounter();
        tq2 += ts2_ - ts1_;
#endif
        if (!shifted)
          { // ascending layout: the own block is aligned, a register is a complete row in column order
            const uint32_t loff4 = (uint32_t)lane * 8u;
            static_for<0, 4>([&](auto cf_) {
              constexpr int cf = cf_;
              static_for<0, 4>([&](auto s0_) {
                constexpr int s0 = s0_;
                const int R = s0 + 4 * cf + 16 * k2;
                // (D3 comes straight out of the MFMA: the hazard between an MFMA result and the store that reads it is the
                // compiler's to resolve)
                row_store(D3[cf][s0], loff4, (uint32_t)(R * rlen + L) * 8u);
              });
            });
          }
        else
          {
            // diagonal-first layout: piece m0 of row R = [carry | own columns without the diagonal]: column c sits at position
            // c + 1 (c < R) or c (c > R) of the piece; the diagonal entry belongs to position 0 of the ROW (= of this piece if
            // the own block is the first one, m0 = 0; else piece 0 takes it in P5 from diagv), and position 0 of this piece
            // takes the carry (last column of the block before).  Lane R - the one holding the diagonal entry - is the lane
            // whose slot is free: it stores the carry of row R, which carry_own holds in exactly that lane.  So a row is ONE
            // store of 512 contiguous bytes with the lanes 0 .. R rotated by one position: no transposition, no LDS.
            const uint32_t offA = (uint32_t)(lane + 1) * 8u, offB = (uint32_t)lane * 8u;
            uint32_t rowrun = (uint32_t)(16 * k2 * rlen + L) * 8u; // uniform: byte offset of the row's piece
            auto own_rows = [&](auto carry_) {
              constexpr bool CARRY = carry_;
              static_for<0, 4>([&](auto cf_) {
                constexpr int cf = cf_;
                static_for<0, 4>([&](auto s0_) {
                  constexpr int s0 = s0_;
                  const int R = s0 + 4 * cf + 16 * k2; // uniform
                  const bool below = lane < R, diag = lane == R;
                  double v = D3[cf][s0];
                  uint32_t off = below ? offA : offB;
                  off = diag ? 0u : off;
                  if constexpr (CARRY)
                    {
                      dg = diag ? v : dg;
                      v = diag ? carry_own : v;
                    }
                  row_store(v, off, rowrun);
                  rowrun += (uint32_t)rlen * 8u;
                });
              });
            };
            if (m0 > 0)
              own_rows(std::true_type{});
            else
              own_rows(std::false_type{});
          }
#ifdef PDHR_STAMP
        tq3 += (long long)__builtin_readcyclecounter() - ts2_;
#endif
      }
    if (shifted && m0 > 0)
      diagv[lane] = dg; // (read by P5 for piece 0, after its wave-level synchronisation)
#ifdef PDHR_STAMP
    if (lane == 0 && Rw.stamps)
      {
        Rw.stamps[(int64_t)slot * 16 + 7] = tq1; // (slots 8 .. 13 belong to P2)
        Rw.stamps[(int64_t)slot * 16 + 14] = tq2;
        Rw.stamps[(int64_t)slot * 16 + 15] = tq3;
      }
#endif
  }

  PDHR_MARK(5);
  // the next polytope's record is requested here: the store phase is long enough to hide the loads (~7k cycles), and
  // before it the 24 registers of the record would sit through the phases with the highest register pressure
  Meta nxt = cur;
  const int nslot = DYNAMIC ? __builtin_amdgcn_readfirstlane(nslot_v) : nslot_v; // (requested at the top of this polytope)
  if (nslot < n_owned)
    nxt = load_meta(nslot);
  if constexpr (SMALL)
    {
      // ================= P5 (streamed kinds): the polytope's n rows are ONE contiguous range of n rlen values ==============
      // Streamed out in 512-byte pieces that start on 128-byte lines.  Position e -> (row R, position p in the row) ->
      // column c (diagonal-first rows: p = 0 is the diagonal entry, the rest ascending without it) -> block b, function j.
      // Own block: the LDS copy of P4; neighbour block: C[k_c(R), l_c(j)] S[pair(R), pair(j)] of that face.
      PDH_WAVE_SYNC();
      const int total = NF * rlen;
      const int mis = (int)(rbase & 15);
      int e = lane - mis;
      double *out = P.values + rbase;
      const int nit = (total + mis + 63) >> 6;
      // digits of every function along each interior face's axis, face by face (MAXF x n ints at the start of W, dead now)
      int *digf = reinterpret_cast<int *>(W);
      for (int r = 0; r < (MAXF * NF + 63) / 64; ++r) // (uniform: the face table is read across lanes)
        {
          const int q0 = lane + 64 * r, q = q0 < MAXF * NF ? q0 : MAXF * NF - 1;
          const int f = q / NF, i = q - f * NF;
          int ax = __shfl(t_axis, n_bdry + f) & 3;
          ax = ax > 2 ? 2 : ax;
          if (q0 < MAXF * NF)
            digf[q] = dig[ax * NF + i];
        }
      PDH_WAVE_SYNC();
      // Row by row, a row in pieces of 64 positions.  What a lane needs to know about ITS position p of a row - block,
      // face, digits of the column function - does not depend on the row, except in diagonal-first rows, where the column is
      // p (rows R <= p - 1 - L) or p - 1 (the rows after): both candidates are decoded once per polytope into registers, a
      // row then costs a few selects, four LDS reads and one multiply per piece.  (Decoding every value's position from its
      // flat index - the first version - was 35 VALU instructions per value and the largest phase of these kinds.)
      constexpr int MAXPC = (7 * NF + 63) / 64; // pieces per row: at most 1 + MAXF blocks of n columns
      struct Cand
      {
        int cvb, svb, dvb, drb; // Call index without 4 k_c(R); slot index without NS u(R); column j; first digit entry of the face
        bool own;
      };
      auto make = [&](int c) {
        const int b = (int)(((float)c + 0.5f) * (1.0f / (float)NF)); // c / n (c < 8 n: exact)
        const int j = c - b * NF;
        int fl = b < m0 ? b : b - 1;
        fl = fl < 0 ? 0 : (fl >= MAXF ? MAXF - 1 : fl);
        const int dJ = digf[fl * NF + j];
        Cand k;
        k.cvb = fl * 16 + (dJ & 15), k.svb = fl * MS + (dJ >> 4), k.dvb = j, k.drb = fl * NF, k.own = b == m0;
        return k;
      };
      Cand cA[MAXPC], cB[MAXPC];
      int thr[MAXPC];
      bool valid[MAXPC];
      static_for<0, MAXPC>([&](auto pc_) {
        constexpr int pc = pc_;
        const int p = pc * 64 + lane;
        valid[pc] = p < rlen;
        const int pv = valid[pc] ? p : 0;
        cA[pc] = make(pv);
        cB[pc] = shifted ? make(pv > 0 ? pv - 1 : 0) : cA[pc];
        thr[pc] = shifted ? pv - 1 - L : (1 << 30);
      });
      (void)total, (void)mis, (void)e, (void)nit;
      const uint32_t loff5 = (uint32_t)lane * 8u;
#pragma unroll 2
#if PDHR_EXP == 1
      for (int R = 0; R < NF && P.n < 0; ++R)
#else
      for (int R = 0; R < NF; ++R)
#endif
        {
          static_for<0, MAXPC>([&](auto pc_) {
            constexpr int pc = pc_;
            if (pc * 64 < rlen)
              {
                const bool useB = R > thr[pc];
                const int cvb = useB ? cB[pc].cvb : cA[pc].cvb, svb = useB ? cB[pc].svb : cA[pc].svb;
                const int dvb = useB ? cB[pc].dvb : cA[pc].dvb, drb = useB ? cB[pc].drb : cA[pc].drb;
                const bool own = useB ? cB[pc].own : cA[pc].own;
                const int dR = digf[drb + R];
                const double cv = Call[cvb + 4 * (dR & 15)];
                const double sv = M2c[svb + (dR >> 4) * NS];
                double dv = Dblk[R * NF + dvb];
                double val = own ? dv : cv * sv;
                if constexpr (pc == 0)
                  if (shifted)
                    { // position 0 of a diagonal-first row: the diagonal entry
                      const double dd = Dblk[R * NF + R];
                      val = lane == 0 ? dd : val;
                    }
#if PDHR_EXP == 5
                if (valid[pc] && P.n < 0)
#else
                if (valid[pc])
#endif
                  {
                    row_store(val, loff5, (uint32_t)(R * rlen + pc * 64) * 8u); // scalar row / piece offset + lane offset
                  }
              }
          });
        }
    }
  else if constexpr (MULTI)
  // ================= P5 (MULTI): coupling blocks in ascending column order, a block = sum over the planes of its neighbour ===
  {
    double carry = 0.0; // lane R: the value that lane 0 stores in row R of the next piece
    bool first_left = true;
    int t = n_bdry;
#ifdef PDHR_STAMP
    long long tb_S = 0, tb_out = 0;
#endif
#if PDHR_EXP == 1
    while (t < nfaces && P.n < 0)
#else
    while (t < nfaces)
#endif
      {
        const int b = rl_i(t_blk, t);
        int te = t + 1;
        while (te < nfaces && rl_i(t_blk, te) == b)
          ++te; // entries [t, te): the planes shared with this neighbour
        const bool left = shifted && b < m0;
        PDH_WAVE_SYNC();
        const bool first_piece = left && first_left; // piece 0 starts with the diagonal entry (diagv, written in P4)
        if (left)
          first_left = false;
        // this lane's column of the block: shifted pieces hold columns -1 .. 62 (lane 0: the carry)
        const int jcol = left ? (lane > 0 ? lane - 1 : 0) : lane;
        const uint32_t rowp = 64u * 8u * (uint32_t)b; // uniform: byte offset of the piece in row 0
        const uint32_t lane_off = (uint32_t)lane * 8u;
        // carries of the shifted pieces: in LDS, read by lane 0 alone (EXEC = 1) straight into the value registers, eight rows
        // at a time - as in the single-plane kernel (coupling_blocks)
        double *ctab = W + 1536; // [64]
        if (left && !first_piece)
          ctab[lane] = carry;
        const double *csrc = first_piece ? diagv : ctab; // uniform
        typedef __attribute__((address_space(3))) const char lds_cchar;
        const unsigned caddr = (unsigned)(uintptr_t)(lds_cchar *)reinterpret_cast<const char *>(csrc);
        double next_carry = 0.0;
        using std::integral_constant;
#ifdef PDHR_MULTI_FUSED
        // (diagnostic build only, never shipped: round 3's first MULTI version, which multiplied and stored the rows of a
        // single-plane neighbour directly - the block kernel's fused path - next to the accumulating path, and faulted on the
        // device.  Kept so that tools/isa_lint.py can be run on exactly that code: DESIGN.md 4c, "the fused-path fault".)
        if (te - t == 1)
          {
            PDH_WAVE_SYNC();
            build_S(t);
            const int c = rl_i(t_axis, t);
            const int lc = digit_c(jcol, c), vt = digits_t(jcol, c);
            double Cl[4], sc[16];
            for (int k = 0; k < 4; ++k)
              Cl[k] = Cbuf[k * 4 + lc];
            for (int u = 0; u < 16; ++u)
              sc[u] = Sbuf[u * 16 + vt];
            next_carry = left ? last_column(c) : 0.0;
            if (left)
              PDH_WAVE_SYNC();
            auto rows = [&, lane_off](auto c_, auto left_) {
              constexpr int cc = c_;
              constexpr bool LEFT = left_;
              const uint32_t loff = lane_off;
              uint32_t rowrun = rowp;
              static_for<0, 8>([&](auto g_) {
                constexpr int g = g_;
                double v[8];
                static_for<0, 8>([&](auto r_) {
                  constexpr int R = 8 * g + r_;
                  constexpr int kc = (R >> (2 * cc)) & 3;
                  constexpr int k0 = R & 3, k1 = (R >> 2) & 3, k2 = (R >> 4) & 3;
                  constexpr int u = cc == 0 ? (k1 + 4 * k2) : (cc == 1 ? (k0 + 4 * k2) : (k0 + 4 * k1));
                  v[r_] = Cl[kc] * sc[u];
                });
                if constexpr (LEFT)
                  PDHR_CARRY_READ8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], caddr, g);
                static_for<0, 8>([&](auto r_) {
                  row_store(v[r_], loff, rowrun);
                  rowrun += (uint32_t)rlen * 8u;
                });
              });
            };
            if (left)
              {
                if (c == 0)
                  rows(integral_constant<int, 0>{}, std::true_type{});
                else if (c == 1)
                  rows(integral_constant<int, 1>{}, std::true_type{});
                else
                  rows(integral_constant<int, 2>{}, std::true_type{});
              }
            else
              {
                if (c == 0)
                  rows(integral_constant<int, 0>{}, std::false_type{});
                else if (c == 1)
                  rows(integral_constant<int, 1>{}, std::false_type{});
                else
                  rows(integral_constant<int, 2>{}, std::false_type{});
              }
          }
        else
#endif
        {
          // acc[r] = sum_e C_e[k_c(R), l_c(j)] S_e[u(R), v(j)], in two halves of 32 rows (64 accumulators do not fit the
          // register file next to what lives across this phase).  S_e / C_e of the first three planes of a neighbour stay in
          // LDS side by side (W + 704 ..., behind the scratch of build_S) for the second half; further planes are rebuilt.
          // (A variant that multiplied and stored the rows of single-plane neighbours directly, like the block-shaped
          // kernel does, faulted on the device in all its four instantiations - stores through a corrupted base - while this
          // form is correct; the cause was not found in the generated code, see profiles/README.md, r03.)
          uint32_t rowrun = rowp;
          static_for<0, 2>([&](auto half_) {
            constexpr int half = half_;
            double acc[32];
            static_for<0, 32>([&](auto r_) { acc[r_] = 0.0; });
            for (int e = t; e < te; ++e)
              {
                const int kslot = e - t;
                Sdst = kslot < 3 ? W + 704 + 272 * kslot : Sbuf;
                Cdst = kslot < 3 ? W + 704 + 272 * kslot + 256 : Cbuf;
                if (half == 0 || kslot >= 3)
                  {
                    PDHR_T0();
                    PDH_WAVE_SYNC();
                    build_S(e);
                    PDHR_ACC(tb_S);
                  }
                const int c = rl_i(t_axis, e);
                const int lc = digit_c(jcol, c), vt = digits_t(jcol, c);
                double Cl[4], sc[16];
                for (int k = 0; k < 4; ++k)
                  Cl[k] = Cdst[k * 4 + lc];
                for (int u = 0; u < 16; ++u)
                  sc[u] = Sdst[u * 16 + vt];
                if (half == 0 && left)
                  next_carry += last_column(c);
                auto add = [&](auto c_) {
                  constexpr int cc = c_;
                  static_for<0, 32>([&](auto r_) {
                    constexpr int R = 32 * half + r_;
                    constexpr int kc = (R >> (2 * cc)) & 3;
                    constexpr int k0 = R & 3, k1 = (R >> 2) & 3, k2 = (R >> 4) & 3;
                    constexpr int u = cc == 0 ? (k1 + 4 * k2) : (cc == 1 ? (k0 + 4 * k2) : (k0 + 4 * k1));
                    acc[r_] += Cl[kc] * sc[u];
                  });
                };
                if (c == 0)
                  add(integral_constant<int, 0>{});
                else if (c == 1)
                  add(integral_constant<int, 1>{});
                else
                  add(integral_constant<int, 2>{});
              }
            if (half == 0 && left)
              PDH_WAVE_SYNC(); // (ctab written above is read below)
            auto out = [&](auto left_) {
              constexpr bool LEFT = left_;
              static_for<0, 4>([&](auto g_) {
                constexpr int g = 4 * half + g_;
                if constexpr (LEFT)
                  PDHR_CARRY_READ8(acc[8 * g_ + 0], acc[8 * g_ + 1], acc[8 * g_ + 2], acc[8 * g_ + 3], acc[8 * g_ + 4], acc[8 * g_ + 5],
                                   acc[8 * g_ + 6], acc[8 * g_ + 7], caddr, g);
                static_for<0, 8>([&](auto r_) {
#if PDHR_EXP == 5
                  if (P.n < 0)
#endif
                    row_store(acc[8 * g_ + r_], lane_off, rowrun);
                  rowrun += (uint32_t)rlen * 8u;
                });
              });
            };
            {
              PDHR_T0();
              if (left)
                out(std::true_type{});
              else
                out(std::false_type{});
              PDHR_ACC(tb_out);
            }
          });
          Sdst = Sbuf, Cdst = Cbuf;
        }
        PDH_WAVE_SYNC();
        carry = next_carry;
        t = te;
      }
#ifdef PDHR_STAMP
    if (lane == 0 && Rw.stamps)
      {
        Rw.stamps[(int64_t)slot * 16 + 8] = tb_S; // (MULTI: slots 8, 9 are free - the tensor path of P2 uses 12, 13)
        Rw.stamps[(int64_t)slot * 16 + 9] = tb_out;
      }
#endif
  }
  else
  // ================= P5: coupling blocks left of the diagonal, in ascending column order ================================
  {
    coupling_blocks(std::true_type{});
  }
  PDHR_MARK(6);
  if (nslot >= n_owned)
    break;
  slot = nslot;
  cur = nxt;
  } // persistent loop over the wave's slots
  leave();
}
} // namespace pdhr
