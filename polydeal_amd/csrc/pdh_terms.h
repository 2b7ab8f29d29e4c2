// pdh_terms.h — the term kernel: the row kernel of the SMALL elements (3-D FE_DGQ(1,2), FE_AggloDGP(1..3)) on agglomerates of
// Cartesian cells with tensor-product rules on every sub-cell and sub-face.  One wave per owned polytope computes and writes ALL
// values of the polytope's n rows (owner computes rows, like pdh_rows.h); what is computed is unchanged (reference
// include/poly_utils.h:2040-2084 volume + boundary, :1870-1926 interface blocks), only the order of summation differs.
//
// Why another form.  On an axis-aligned cell with a tensor rule every integrand of the path is a product of three 1-D sums, so
// every entry of the polytope's rows is a short sum of products of three small-matrix entries ("terms"):
//   A[P,P]_(k,l) = sum over cells   (K0' M1 M2 + M0 K1 M2 + M0 M1 K2)[k_d, l_d]        M_d = sum_a w_a B_k B_l,  K_d = sum_a w_a B'_k B'_l / h_d^2
//                + sum over sub-faces D0 D1 D2 [k_d, l_d]                              (K0' = K0 + c M0 with a reaction term)
//                  tangential d: D_d = sum_a w_a B_k B_l (own-side JxW; 2 JxW on the boundary),
//                  normal c:     D_c = sigma B_k B_l - 1/2 s (B'_k B_l + B_k B'_l) / h_c  at the plane (sigma / 2 on the boundary)
//   A[P,Q]_(k,l) = sum over the sub-faces shared with Q   X0 X1 X2 [k_d, l_d]
//                  tangential d: X_d = sum_a w_a B^P_k B^Q_l (JxW of side 1, each basis in its own box),
//                  normal c:     X_c = (1/2 s B'_k(zP)/hP - sigma B_k(zP)) B_l(zQ) - 1/2 s B_k(zP) B'_l(zQ)/hQ
// (the same identities as pdh_rows.h uses for its Kronecker form C (x) S, written per sub-face: no Legendre moments, no contraction
// stages, no limit on the number of planes a neighbour is met along - "staircase" faces of METIS-like agglomerates of Cartesian
// cells, reference examples/poisson.cc:543-566, are just more terms).  The streamed kinds of pdh_rows.h spent 6 100 VALU
// instructions and two dozen LDS hand-offs per polytope of FE_AggloDGP(3) on moments and a three-stage sum factorisation
// (profiles/r02v4_pmc_sq_dgp.txt: 2.0e8 VALU wave-instructions per launch, time ~ 1 / resident waves); this form needs a third
// of the instructions and three hand-offs.  Cost grows with cells + sub-faces per polytope (the moment form's does not): the
// host takes this kernel while a polytope's tables fit the LDS budget below, else the kinds of pdh_rows.h.
//
// Phases (one wave per polytope; LDS: record copy, digit table, coupling tables X, diagonal tables D / M / K):
//   A  lane tasks build the small matrices: (sub-face, tangential direction) - fn points each -, (sub-face, normal direction),
//      (cell, direction) - tn points each; symmetric ones packed (10 of 16 entries for degree 3)
//   B1 diagonal block: lane = (column j, subset s of the terms); per term 3 column vectors from LDS, NS + n multiply-adds;
//      the subsets are summed across lanes (ds_bpermute) and the block is left in LDS
//   B2 rows: a row in pieces of 64 positions, lane = position = column; a coupling column sums the terms of its neighbour's
//      sub-faces for all n rows in registers, an own column reads the block of B1; n buffer stores of 512 contiguous bytes per piece
#pragma once
#include "pdh_kernels.h"
#include "pdh_terms_tables.h"

#ifdef PDHT_STAMP
#define PDHT_MARK(k)                                                                                                  \
  do                                                                                                                  \
    {                                                                                                                 \
      const long long tm_ = (long long)__builtin_readcyclecounter();                                                  \
      if (lane == 0 && T.stamps)                                                                                      \
        T.stamps[(int64_t)slot * 16 + (k)] = tm_;                                                                      \
    }                                                                                                                 \
  while (0)
#else
#define PDHT_MARK(k)
#endif

namespace pdht
{
using pdh::static_for;
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

constexpr int TERMS_HDR = 12, TERMS_ENT = 10;

template <int N1D, int BASIS>
struct Kind
{
  static constexpr int NF = BASIS == 0 ? N1D * N1D * N1D : N1D * (N1D + 1) * (N1D + 2) / 6; // functions
  static constexpr int NS = BASIS == 0 ? N1D * N1D : N1D * (N1D + 1) / 2;                   // pairs (k1, k2) that occur
  static constexpr int NSYM = N1D * (N1D + 1) / 2, FULL = N1D * N1D;
  static constexpr int SYMS = NSYM | 1, FULLS = FULL | 1; // odd strides: the lane tasks of phase A write table after table
  static constexpr int NSUB = 64 / NF > 0 ? 64 / NF : 1;  // subsets of the diagonal block's terms (lanes = NSUB x NF columns)
  struct Dig
  {
    int k0, k1, k2;
  };
  // digits of function R (x fastest; BASIS 1: k0 + k1 + k2 <= p, pdh_basis.h: multi_indices)
  __host__ __device__ static constexpr Dig dig(int R)
  {
    if (BASIS == 0)
      return Dig{R % N1D, (R / N1D) % N1D, R / (N1D * N1D)};
    int cnt = 0;
    for (int iz = 0; iz < N1D; ++iz)
      for (int iy = 0; iy < N1D - iz; ++iy)
        for (int ix = 0; ix < N1D - iy - iz; ++ix)
          {
            if (cnt == R)
              return Dig{ix, iy, iz};
            ++cnt;
          }
    return Dig{0, 0, 0};
  }
  __host__ __device__ static constexpr int pair(int k1, int k2) { return BASIS == 0 ? k1 + N1D * k2 : k2 * N1D - k2 * (k2 - 1) / 2 + k1; }
  __host__ __device__ static constexpr bool pair_ok(int k1, int k2) { return BASIS == 0 || k1 + k2 < N1D; }
  __host__ __device__ static constexpr int sym(int k, int l) { return k <= l ? l * (l + 1) / 2 + k : k * (k + 1) / 2 + l; }
};

// LDS of a workgroup in doubles (host and device agree through this one function)
__host__ __device__ constexpr int terms_rec_doubles(int maxruns) { return (TERMS_HDR + maxruns * TERMS_ENT + 1) & ~1; }
// Two-phase tables (SPLIT): the D / M / K tables are needed by the diagonal block only, the X tables by the row pieces only.  Made
// in one pass they cost FE_AggloDGP(3) on block polytopes 21 KB of LDS per wave = 7 resident waves per CU, 30-40 KB on METIS-like
// agglomerates, and the kernel runs at the speed its occupancy allows; made one after the other - the X tables from the point data
// still held in registers, behind the finished block - 13.7 KB = 11 waves, for 250 more VALU instructions per polytope (the bases
// at the tangential points are evaluated twice): 0.384 -> 0.373 ms on the bench mesh, 0.71 -> 0.56 ms on its grown agglomerates
// (profiles/r04_terms_split.txt).  The host takes the form that gives a polytope's workgroup more resident waves (PdhTerms::split).
// (BLOCK_IN_LDS: the wave-per-polytope kernel leaves the diagonal block in LDS over the dead tables; the workgroup kernel of
// pdh_terms_wg.h stores it from registers)
template <int N1D, int BASIS, bool BLOCK_IN_LDS = true, bool SPLIT = false>
__host__ __device__ constexpr int terms_lds_doubles(int maxruns, int maxsf, int maxsi, int maxcell)
{
  using K = Kind<N1D, BASIS>;
  const int dg = (K::NF + 1) / 2 + ((K::NF + 1) / 2 & 1);
  const int xa = maxsi * 3 * K::FULLS + ((maxsi * 3 * K::FULLS) & 1);
  int da = maxsf * 3 * K::SYMS + maxcell * 6 * K::SYMS;
  if (BLOCK_IN_LDS && SPLIT)
    { // the X tables are made after the diagonal block and stand BEHIND it, over the D tables (dead by then)
      const int xb = K::NF * K::NF + xa;
      da = da > xb ? da : xb;
      return terms_rec_doubles(maxruns) + dg + da + (da & 1);
    }
  if (BLOCK_IN_LDS)
    da = da > K::NF * K::NF ? da : K::NF * K::NF;
  return terms_rec_doubles(maxruns) + dg + xa + da + (da & 1);
}

// A polytope's record of 1-D rules (PdhTerms::tdata): the points of every (sub-face, tangential direction) and (cell, direction) task
// [task][x | w_self | w_cross][pmax], then per sub-face its plane coordinate and descriptor
__host__ __device__ constexpr int terms_task_doubles(int maxsf, int maxcell, int pmax)
{
  return (2 * maxsf + 3 * maxcell) * pmax * 3 + 2 * maxsf;
}

// Phase A of the term kernels: the lane tasks that build the small matrices of a polytope in LDS.  Shared by the wave-per-polytope
// kernel below and the workgroup-per-polytope kernel of pdh_terms_wg.h.  PMAX: most points per direction of a rule (4 or 8): the
// point data of a task sit in registers.
// PAIR: a task of up to 8 points (composite rules of merged cells / sub-faces, rules of more than four points) is worked on by TWO
// neighbouring lanes, four slots each, whose partial matrices are added through one DPP exchange - half the serial chain of an 8-slot
// lane task and its 48 registers of point data (measured: 156 -> 122 VGPRs in the workgroup kernel).  PMAX = slots per lane (4);
// the records have TPM = 8 slots per task then.
template <int N1D, int BASIS, int PMAX, bool PAIR = false>
struct TermTasks
{
  static constexpr int TPM = PAIR ? 2 * PMAX : PMAX; // slots per 1-D rule in the polytope's record
  using K = Kind<N1D, BASIS>;
  static constexpr int NSYM = K::NSYM, SYMS = K::SYMS, FULL = K::FULL, FULLS = K::FULLS;
  const PdhDev &P;
  const double *rec; // LDS copy of the run entries (behind TERMS_HDR unused doubles)
  double *Xa, *Da, *Ca;
  double lo0, lo1, lo2, ih0, ih1, ih2; // own box: lower corner, 1 / side
  int nsfb, fn, tn;
  const double *gr; // the polytope's record of 1-D rules (PdhTerms::tdata)
  int maxsf, maxcell;
  __device__ __forceinline__ static double sel3(int c, double x0, double x1, double x2) { return c == 0 ? x0 : (c == 1 ? x1 : x2); }
  // 1-D basis in the centred variable of a box (pdh_basis.h: monomial coefficients, uniform -> scalar operands)
  __device__ __forceinline__ void basis(double t, double *b) const
  {
    static_for<0, N1D>([&](auto k_) {
      constexpr int k = k_;
      double v = P.tab.coef[k][N1D - 1];
      for (int m = N1D - 2; m >= 0; --m)
        v = v * t + P.tab.coef[k][m];
      b[k] = v;
    });
  }
  __device__ __forceinline__ void basis_d(double t, double *b, double *db) const
  {
    static_for<0, N1D>([&](auto k_) {
      constexpr int k = k_;
      double v = P.tab.coef[k][N1D - 1], d = 0.0;
      for (int m = N1D - 2; m >= 0; --m)
        {
          d = d * t + v;
          v = v * t + P.tab.coef[k][m];
        }
      b[k] = v, db[k] = d;
    });
  }
  // (sub-face, tangential direction): D_d (own x own, own-side weights) and, towards a neighbour, X_d (own x neighbour, JxW of
  // side 1).  The rule of a sub-face is a_alpha b_beta: direction 0 takes w_(alpha,0), direction 1 takes w_(0,beta) / w_(0,0)
  struct Pts // point data of one lane task: tangential x | w_self | w_cross; cell x | w; normal-direction task: x[0] = plane coordinate
  {
    double x[PMAX], ws[PMAX], wc[PMAX];
    double ws0, wc0; // weights of the task's FIRST point (the normalisation of the second tangential / the other cell directions)
    int npts;        // live points of this lane: intervals x points of a rule (PAIR: its half of them)
  };
  using TPts = Pts;
  using CPts = Pts;
  // sum of v over the two lanes of a pair (lanes 2 k and 2 k + 1, both active)
  __device__ __forceinline__ static double pair_sum(double v)
  {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
  }
  // half: which four slots of the task this lane takes (PAIR only; else 0)
  __device__ __forceinline__ TPts tang_load(int info, int sf, int dir, int half = 0) const
  { // all loads of a task at once: one contiguous piece of the polytope's record
    TPts r;
    const double *g = gr + (2 * sf + dir) * (3 * TPM);
    const int npts = ((info >> (dir ? 15 : 12)) & 7) * fn;
    r.npts = PAIR ? (npts - PMAX * half < 0 ? 0 : (npts - PMAX * half > PMAX ? PMAX : npts - PMAX * half)) : npts;
    r.ws0 = g[TPM];
    r.wc0 = g[2 * TPM];
    const double *gh = g + (PAIR ? PMAX * half : 0);
    static_for<0, PMAX>([&](auto i_) {
      constexpr int al = i_;
      r.x[al] = gh[al];
      r.ws[al] = gh[TPM + al];
      r.wc[al] = gh[2 * TPM + al]; // (zero on the boundary)
    });
    return r;
  }
  __device__ __forceinline__ double zeta_load(int sf) const { return gr[(2 * maxsf + 3 * maxcell) * (3 * TPM) + sf]; }
  // WANT_D / WANT_X: which of the two kinds of table a call produces (both by default; the kernel of FE_AggloDGP(3) makes them in
  // two phases so that the X tables can take the place of the D tables in LDS)
  template <bool WANT_D = true, bool WANT_X = true>
  __device__ __forceinline__ void tang_compute(const TPts &r, int sf, int dir, int info, int half = 0) const
  {
    const int run = info & 0xff, c = (info >> 8) & 3;
    const int ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2, ax = dir ? tj : ti;
    const double *re = rec + TERMS_HDR + run * TERMS_ENT;
    const bool interior = (int)__double_as_longlong(re[1]) >= 0;
    const double lo_d = sel3(ax, lo0, lo1, lo2), ih_d = sel3(ax, ih0, ih1, ih2);
    const double loq = re[3 + ax], ihq = re[6 + ax];
    const double sS = dir ? 1.0 / r.ws0 : 1.0, sC = (dir && interior) ? 1.0 / r.wc0 : 1.0;
    double Dm[NSYM], Xm[FULL];
    for (int i = 0; i < NSYM; ++i)
      Dm[i] = 0.0;
    for (int i = 0; i < FULL; ++i)
      Xm[i] = 0.0;
    static_for<0, PMAX>([&](auto i_) {
      constexpr int al = i_;
      if (al < r.npts)
        {
          double bp[N1D], bq[N1D];
          basis((r.x[al] - lo_d) * ih_d - 0.5, bp);
          basis((r.x[al] - loq) * ihq - 0.5, bq);
          const double wS = r.ws[al] * sS, wC = r.wc[al] * sC;
          static_for<0, N1D>([&](auto k_) {
            constexpr int k = k_;
            const double ws = wS * bp[k], wc = wC * bp[k];
            if constexpr (WANT_D)
              static_for<k, N1D>([&](auto l_) { Dm[K::sym(k, l_)] += ws * bp[l_]; });
            if constexpr (WANT_X)
              static_for<0, N1D>([&](auto l_) { Xm[l_ * N1D + k] += wc * bq[l_]; });
          });
        }
    });
    if constexpr (PAIR)
      { // the other half of the task's points sits in the neighbouring lane
        if constexpr (WANT_D)
          for (int i = 0; i < NSYM; ++i)
            Dm[i] = pair_sum(Dm[i]);
        if constexpr (WANT_X)
          for (int i = 0; i < FULL; ++i)
            Xm[i] = pair_sum(Xm[i]);
        if (half != 0)
          return;
      }
    if constexpr (WANT_D)
      {
        double *dd = Da + (sf * 3 + ax) * SYMS;
        for (int i = 0; i < NSYM; ++i)
          dd[i] = Dm[i];
      }
    if (WANT_X && interior)
      {
        double *xd = Xa + ((sf - nsfb) * 3 + ax) * FULLS;
        for (int i = 0; i < FULL; ++i)
          xd[i] = Xm[i];
      }
  }
  // (sub-face, normal direction): D_c and X_c at the plane
  template <bool WANT_D = true, bool WANT_X = true>
  __device__ __forceinline__ void norm_compute(double zeta, int sf, int info) const
  {
    const int run = info & 0xff, c = (info >> 8) & 3;
    const double sg = ((info >> 10) & 1) ? 1.0 : -1.0;
    const double *re = rec + TERMS_HDR + run * TERMS_ENT;
    const bool interior = (int)__double_as_longlong(re[1]) >= 0;
    const double sig = re[2];
    const double lo_c = sel3(c, lo0, lo1, lo2), ih_c = sel3(c, ih0, ih1, ih2);
    const double loq = re[3 + c], ihq = re[6 + c];
    double bp[N1D], dp[N1D], bq[N1D], dq[N1D];
    basis_d((zeta - lo_c) * ih_c - 0.5, bp, dp);
    basis_d((zeta - loq) * ihq - 0.5, bq, dq);
    double *dd = Da + (sf * 3 + c) * SYMS;
    double *xd = Xa + ((sf - nsfb) * 3 + c) * FULLS;
    const double hs = 0.5 * sg * ih_c, hq = 0.5 * sg * ihq;
    static_for<0, N1D>([&](auto k_) {
      constexpr int k = k_;
      if constexpr (WANT_D)
        static_for<k, N1D>([&](auto l_) {
          constexpr int l = l_;
          dd[K::sym(k, l)] = sig * bp[k] * bp[l] - hs * (dp[k] * bp[l] + bp[k] * dp[l]);
        });
      if (WANT_X && interior)
        static_for<0, N1D>([&](auto l_) {
          constexpr int l = l_;
          xd[l * N1D + k] = (hs * dp[k] - sig * bp[k]) * bq[l] - hq * bp[k] * dq[l];
        });
    });
  }
  // (cell, direction): M_d and K_d.  The rule of a cell is a_i b_j c_k: direction 0 takes w_(i,0,0), the others w / w_000
  __device__ __forceinline__ CPts cell_load(int ct, int half = 0) const
  {
    CPts r;
    const double *g = gr + (2 * maxsf + ct) * (3 * TPM);
    r.npts = 0;
    r.ws0 = g[TPM];
    r.wc0 = 0.0;
    const double *gh = g + (PAIR ? PMAX * half : 0);
    static_for<0, PMAX>([&](auto i_) {
      constexpr int i = i_;
      r.x[i] = gh[i];
      r.ws[i] = gh[TPM + i];
      r.wc[i] = 0.0;
      r.npts += r.ws[i] != 0.0 ? 1 : 0; // (slots behind the last interval carry zero weights)
    });
    return r;
  }
  __device__ __forceinline__ void cell_compute(const CPts &r, int ct, int half = 0) const
  {
    const int cell = ct / 3, d = ct - 3 * cell;
    const double lo_d = sel3(d, lo0, lo1, lo2), ih_d = sel3(d, ih0, ih1, ih2);
    const double sc = d == 0 ? 1.0 : 1.0 / r.ws0;
    double Mm[NSYM], Km[NSYM];
    for (int i = 0; i < NSYM; ++i)
      Mm[i] = Km[i] = 0.0;
    static_for<0, PMAX>([&](auto i_) {
      constexpr int i = i_;
      if (i < r.npts)
        {
          double bp[N1D], dp[N1D];
          basis_d((r.x[i] - lo_d) * ih_d - 0.5, bp, dp);
          const double w = r.ws[i] * sc;
          static_for<0, N1D>([&](auto k_) {
            constexpr int k = k_;
            const double wb = w * bp[k], wd = w * (dp[k] * ih_d);
            static_for<k, N1D>([&](auto l_) {
              Mm[K::sym(k, l_)] += wb * bp[l_];
              Km[K::sym(k, l_)] += wd * (dp[l_] * ih_d);
            });
          });
        }
    });
    if constexpr (PAIR)
      {
        for (int i = 0; i < NSYM; ++i)
          Mm[i] = pair_sum(Mm[i]), Km[i] = pair_sum(Km[i]);
        if (half != 0)
          return;
      }
    if (d == 0 && P.reaction_c != 0.0) // c phi_i phi_j rides on the first term of the cell
      for (int i = 0; i < NSYM; ++i)
        Km[i] += P.reaction_c * Mm[i];
    double *cd = Ca + (cell * 3 + d) * 2 * SYMS;
    for (int i = 0; i < NSYM; ++i)
      {
        cd[i] = Mm[i];
        cd[SYMS + i] = Km[i];
      }
  }
};

// Set-up (once per problem): the 1-D rules of polytope `slot` gathered from the point arrays through the descriptors of PdhTerms into its
// record out[tstride] (layout: PdhTerms::tdata).  nt threads share the work.
__device__ __forceinline__ void terms_gather_record(const PdhDev &P, const PdhTerms &T, int slot, int tid, int nt, double *__restrict__ out)
{
  const int REC = TERMS_HDR + T.maxruns * TERMS_ENT;
  const double *g = T.meta + (int64_t)slot * REC;
  const int ncell = (int)((__double_as_longlong(g[0]) >> 16) & 0xffff), nsf = (int)__double_as_longlong(g[11]);
  const int64_t vq_b = __double_as_longlong(g[10]);
  const int fn = T.fq_tensor_n, tn = T.vq_tensor_n, pm = T.tpm;
  const int ntt = 2 * nsf, ntc = 3 * ncell;
  const int64_t sfb = (int64_t)slot * T.maxsf;
  for (int it = tid; it < (ntt + ntc) * pm; it += nt)
    {
      const int task = it / pm, al = it - task * pm;
      double x = 0.0, ws = 0.0, wc = 0.0;
      int at;
      if (task < ntt)
        {
          const int sf = task >> 1, dir = task & 1;
          const int info = T.sf_info[sfb + sf];
          const int64_t pb = T.sf_pt[sfb + sf];
          const int c = (info >> 8) & 3;
          const bool fast_j = ((info >> 11) & 1) != 0;
          const int ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2, ax = dir ? tj : ti;
          const int64_t stp = ((dir == 1) == fast_j) ? 1 : fn;
          const int npts = ((info >> (dir ? 15 : 12)) & 7) * fn;
          if (al < npts)
            {
              const int ii = (al >= fn ? 1 : 0) + (al >= 2 * fn ? 1 : 0) + (al >= 3 * fn ? 1 : 0);
              const int64_t q = pb + T.sf_ivl[((sfb + sf) * 2 + dir) * TERMS_MI + ii] + (al - ii * fn) * stp;
              x = P.ap_x[(int64_t)ax * P.ap_stride + q];
              ws = P.ap_wself[q];
              wc = P.ap_wcross[q]; // (zero on the boundary)
            }
          at = task;
        }
      else
        {
          const int ct = task - ntt, cell = ct / 3, d = ct - 3 * cell;
          const int ii = (al >= tn ? 1 : 0) + (al >= 2 * tn ? 1 : 0) + (al >= 3 * tn ? 1 : 0);
          const int ci = ii < TERMS_MI ? T.cell_ivl[((int64_t)slot * T.maxcell * 3 + ct) * TERMS_MI + ii] : -1;
          if (ci >= 0)
            {
              const int64_t step = d == 0 ? 1 : (d == 1 ? tn : tn * tn);
              const int64_t q = vq_b + (int64_t)ci * (tn * tn * tn) + (al - ii * tn) * step;
              x = P.vq_x[(int64_t)d * P.vq_stride + q];
              ws = P.vq_w[q];
            }
          at = 2 * T.maxsf + ct;
        }
      double *pt = out + at * (3 * pm);
      pt[al] = x;
      pt[pm + al] = ws;
      pt[2 * pm + al] = wc;
    }
  double *zt = out + (2 * T.maxsf + 3 * T.maxcell) * (3 * pm);
  for (int sf = tid; sf < nsf; sf += nt)
    {
      const int info = T.sf_info[sfb + sf];
      zt[sf] = P.ap_x[(int64_t)((info >> 8) & 3) * P.ap_stride + T.sf_pt[sfb + sf]];
      zt[T.maxsf + sf] = __longlong_as_double((long long)info);
    }
}

// PMAX: most points per direction of a rule the instantiation takes (4 or 8): the point data of a lane task sit in registers
#ifndef PDHT_WAVES
#define PDHT_WAVES 3 // waves per SIMD the register allocation aims at (4: 128 VGPRs - the 8-slot lane tasks then spill 50-190 bytes; measured
                     // equal or slower, profiles/r04_terms_waves.txt)
#endif
template <int N1D, int BASIS, bool SHIFTED, int PMAX, bool SPLIT>
__global__ void __launch_bounds__(PDH_WAVE, PDHT_WAVES) k_terms(const PdhDev P, const PdhTerms T, const int n_owned)
{
  using K = Kind<N1D, BASIS>;
  constexpr int NF = K::NF, NS = K::NS, NSYM = K::NSYM, SYMS = K::SYMS, FULL = K::FULL, FULLS = K::FULLS, NSUB = K::NSUB;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int slot = blockIdx.x; // (an XCD-chunked order of the polytopes, as in the two-kernel forms, changes nothing here: whole lines)
  if (slot >= n_owned)
    return;
  PDHT_MARK(0);
  const int REC = TERMS_HDR + T.maxruns * TERMS_ENT;
  double *rec = lds;
  int *dig = reinterpret_cast<int *>(lds + terms_rec_doubles(T.maxruns));
  // (SPLIT: X tables in a second pass, behind the diagonal block - see above)
  double *Xa = lds + terms_rec_doubles(T.maxruns) + ((NF + 1) / 2 + ((NF + 1) / 2 & 1));
  double *Da = SPLIT ? Xa : Xa + (T.maxsi * 3 * FULLS + ((T.maxsi * 3 * FULLS) & 1));
  if constexpr (SPLIT)
    Xa = Da + NF * NF;
  auto sel3 = [](int c, double x0, double x1, double x2) { return c == 0 ? x0 : (c == 1 ? x1 : x2); };
  // ---- level 1 of the loads: header (uniform address: scalar loads), run entries (-> LDS), sub-face descriptors of the first
  // round of lane tasks of either kind.  The descriptors of a polytope stand at a fixed stride (maxsf per polytope), so nothing
  // here waits for anything else; the point data (level 2) need the descriptors only.
  const double *g = T.meta + (int64_t)slot * REC;
  const long long h0 = __double_as_longlong(g[0]);
  const int ncell = (int)((h0 >> 16) & 0xffff), nsfb = (int)(h0 >> 32);
  const int nsf = (int)__double_as_longlong(g[11]);
  const double lo0 = g[1], lo1 = g[2], lo2 = g[3];
  const double ih0 = g[4], ih1 = g[5], ih2 = g[6];
  const int64_t rbase = __double_as_longlong(g[7]);
  const int rlen = (int)__double_as_longlong(g[8]);
  const int L = (int)__double_as_longlong(g[9]);
  const int fn = T.fq_tensor_n, tn = T.vq_tensor_n;
  // the polytope's record of 1-D rules (PdhTerms::tdata): everything phase A reads from HBM besides the run entries, at addresses
  // that depend on the slot alone
  const double *gr = T.tdata + (int64_t)slot * T.tstride;
  const double *gr_info = gr + (2 * T.maxsf + 3 * T.maxcell) * (3 * PMAX) + T.maxsf;
  auto desc = [&](int sf, int &info) { // (entries behind a polytope's sub-faces are zero: harmless to load)
    info = (int)__double_as_longlong(gr_info[sf < T.maxsf ? sf : 0]);
  };
  int infoT = 0, infoN = 0;
  desc(lane, infoN);
  for (int k0 = 0; k0 < T.maxruns * TERMS_ENT; k0 += 4 * PDH_WAVE)
    { // (four loads in flight per lane: a rolled copy loop would wait for every single one)
      double rv[4];
      static_for<0, 4>([&](auto i_) {
        const int k = k0 + lane + PDH_WAVE * i_;
        rv[i_] = k < T.maxruns * TERMS_ENT ? g[TERMS_HDR + k] : 0.0;
      });
      static_for<0, 4>([&](auto i_) {
        const int k = k0 + lane + PDH_WAVE * i_;
        if (k < T.maxruns * TERMS_ENT)
          rec[TERMS_HDR + k] = rv[i_];
      });
    }
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
      dig[lane] = k0 | (k1 << 4) | (k2 << 8);
    }
  double *Ca = Da + nsf * 3 * SYMS; // [cell][direction][M | K] behind the sub-faces' tables
  // PMAX = slots per 1-D rule in the records (PdhTerms::tpm); rules of more than four slots are worked on by pairs of lanes (TermTasks)
  constexpr bool PAIR = PMAX > 4;
  constexpr int PL = PAIR ? PMAX / 2 : PMAX, TS = PAIR ? 2 : 1; // slots per lane, lanes per task
  using TT = TermTasks<N1D, BASIS, PL, PAIR>;
  using TPts = typename TT::TPts;
  using CPts = typename TT::CPts;
  const TT tt{P, rec, Xa, Da, Ca, lo0, lo1, lo2, ih0, ih1, ih2, nsfb, fn, tn, gr, T.maxsf, T.maxcell};
  // lanes of the first kind: (sub-face, tangential direction[, half]); of the second: the normal-direction tasks, then - from an even
  // lane on, pairs must be neighbours - (cell, direction[, half])
  const int n1 = 2 * nsf * TS, ncs = PAIR ? (nsf + 1) & ~1 : nsf, n2 = ncs + 3 * ncell * TS;
  auto dec1 = [&](int tid, int &sf, int &dir, int &half) {
    const int task = tid / TS;
    half = tid - task * TS, sf = task >> 1, dir = task & 1;
  };
  auto dec2 = [&](int tid, int &ct, int &half) {
    const int u = tid - ncs;
    ct = u / TS, half = u - ct * TS;
  };
  auto tang_load = [&](int info, int sf, int dir, int half) { return tt.tang_load(info, sf < T.maxsf ? sf : 0, dir, half); };
  auto tang_compute = [&](const TPts &r, int sf, int dir, int info, int half) { tt.template tang_compute<true, !SPLIT>(r, sf, dir, info, half); };
  auto norm_compute = [&](double zeta, int sf, int info) { tt.template norm_compute<true, !SPLIT>(zeta, sf, info); };
  PDHT_MARK(1);
  TPts tp0; // (first round of the task form; the two-pass form uses its point data twice)
  double zeta0 = 0.0;
  tp0.npts = 0;
  // (an "entry form" of phase A - lane = one entry (k, l) of one task's matrix, the points staged in LDS - was measured against these
  // lane tasks: 1.6 x the instructions for a shorter critical path; FE_AggloDGP(3) 0.32 -> 0.39 ms, FE_DGQ(2) 0.41 -> 0.50 here, 1.30 vs
  // 1.28 ms in the workgroup kernel of pdh_terms_wg.h whose waves share the entries - profiles/r04_wg_forms.txt; not kept)
  // ---- level 2: the point data of the first round of BOTH kinds of task are requested before anything is computed
  int sfT, dirT, halfT;
  dec1(lane, sfT, dirT, halfT);
  desc(sfT, infoT);
  tp0 = tang_load(infoT, sfT, dirT, halfT);
  CPts cp0;
  for (int i = 0; i < PL; ++i)
    cp0.x[i] = cp0.ws[i] = cp0.wc[i] = 0.0;
  cp0.npts = 0, cp0.ws0 = 1.0, cp0.wc0 = 0.0;
  int ct0 = 0, halfC = 0;
  if (lane < nsf)
    zeta0 = tt.zeta_load(lane);
  else if (lane >= ncs && lane < n2)
    {
      dec2(lane, ct0, halfC);
      cp0 = tt.cell_load(ct0, halfC);
    }
  __builtin_amdgcn_sched_barrier(0);
  PDH_WAVE_SYNC(); // (run entries and digit table are in LDS)
  if (lane < n1)
    tang_compute(tp0, sfT, dirT, infoT, halfT);
  for (int t0 = PDH_WAVE; t0 < n1; t0 += PDH_WAVE)
    {
      const int tid = t0 + lane;
      if (tid < n1)
        {
          int info, sf, dir, half;
          dec1(tid, sf, dir, half);
          desc(sf, info);
          const TPts tp = tang_load(info, sf, dir, half);
          tang_compute(tp, sf, dir, info, half);
        }
    }
  PDHT_MARK(2);
  if (lane < nsf)
    norm_compute(zeta0, lane, infoN);
  else if (lane >= ncs && lane < n2)
    tt.cell_compute(cp0, ct0, halfC);
  for (int t0 = PDH_WAVE; t0 < n2; t0 += PDH_WAVE)
    {
      const int tid = t0 + lane;
      if (tid < nsf)
        {
          int info;
          desc(tid, info);
          norm_compute(tt.zeta_load(tid), tid, info);
        }
      else if (tid >= ncs && tid < n2)
        {
          int ct, half;
          dec2(tid, ct, half);
          const CPts cp = tt.cell_load(ct, half);
          tt.cell_compute(cp, ct, half);
        }
    }
  PDH_WAVE_SYNC();
  PDHT_MARK(3);

  // ================= B1: the diagonal block ========================================================================
  {
    double acc[NF];
    for (int r = 0; r < NF; ++r)
      acc[r] = 0.0;
    const int s = lane / NF, j = lane - s * NF;
    const bool on = s < NSUB;
    const int dj = dig[j];
    const int l0 = dj & 15, l1 = (dj >> 4) & 15, l2 = dj >> 8;
    int o0[N1D], o1[N1D], o2[N1D]; // packed positions of column l_d of a symmetric table
    for (int k = 0; k < N1D; ++k)
      {
        o0[k] = K::sym(k, l0);
        o1[k] = K::sym(k, l1) + SYMS;
        o2[k] = K::sym(k, l2) + 2 * SYMS;
      }
    // cells: (K0' M1 M2 + M0 (K1 M2 + M1 K2))[k, l]
    for (int u = on ? s : ncell; u < ncell; u += NSUB)
      {
        const double *b = Ca + u * 6 * SYMS;
        double M0[N1D], K0[N1D], M1[N1D], K1[N1D], M2[N1D], K2[N1D];
        for (int k = 0; k < N1D; ++k)
          {
            M0[k] = b[o0[k]], K0[k] = b[o0[k] + SYMS];
            M1[k] = b[o1[k] + SYMS], K1[k] = b[o1[k] + 2 * SYMS];
            M2[k] = b[o2[k] + 2 * SYMS], K2[k] = b[o2[k] + 3 * SYMS];
          }
        double mm[NS], km[NS];
        static_for<0, N1D>([&](auto k2_) {
          static_for<0, N1D>([&](auto k1_) {
            constexpr int k1 = k1_, k2 = k2_;
            if constexpr (K::pair_ok(k1, k2))
              {
                mm[K::pair(k1, k2)] = M1[k1] * M2[k2];
                km[K::pair(k1, k2)] = K1[k1] * M2[k2] + M1[k1] * K2[k2];
              }
          });
        });
        static_for<0, NF>([&](auto r_) {
          constexpr typename K::Dig g = K::dig(r_);
          acc[r_] += K0[g.k0] * mm[K::pair(g.k1, g.k2)] + M0[g.k0] * km[K::pair(g.k1, g.k2)];
        });
      }
    // sub-faces: (D0 D1 D2)[k, l]
    for (int u = on ? s : nsf; u < nsf; u += NSUB)
      {
        const double *b = Da + u * 3 * SYMS;
        double f0[N1D], f1[N1D], f2[N1D];
        for (int k = 0; k < N1D; ++k)
          f0[k] = b[o0[k]], f1[k] = b[o1[k]], f2[k] = b[o2[k]];
        double yz[NS];
        static_for<0, N1D>([&](auto k2_) {
          static_for<0, N1D>([&](auto k1_) {
            constexpr int k1 = k1_, k2 = k2_;
            if constexpr (K::pair_ok(k1, k2))
              yz[K::pair(k1, k2)] = f1[k1] * f2[k2];
          });
        });
        static_for<0, NF>([&](auto r_) {
          constexpr typename K::Dig g = K::dig(r_);
          acc[r_] += f0[g.k0] * yz[K::pair(g.k1, g.k2)];
        });
      }
    // sum over the subsets (lanes j, j + n, j + 2 n, ...), then the block into LDS over the tables (dead now)
    if constexpr (NSUB > 1)
      static_for<0, NF>([&](auto r_) {
        double v = acc[r_];
        for (int q = 1; q < NSUB; ++q)
          v += __shfl(acc[r_], lane + q * NF);
        acc[r_] = v;
      });
    PDH_WAVE_SYNC();
    if (lane < NF)
      static_for<0, NF>([&](auto r_) { Da[r_ * NF + lane] = acc[r_]; });
    PDH_WAVE_SYNC();
  }

  if constexpr (SPLIT)
    {
      // ---- second pass of the lane tasks: the X tables, from the point data of the first round still in registers (later rounds:
      // loaded again), into the space behind the block
      if (lane < n1)
        tt.template tang_compute<false, true>(tp0, sfT, dirT, infoT, halfT);
      for (int t0 = PDH_WAVE; t0 < n1; t0 += PDH_WAVE)
        {
          const int tid = t0 + lane;
          if (tid < n1)
            {
              int info, sf, dir, half;
              dec1(tid, sf, dir, half);
              desc(sf, info);
              const TPts tp = tang_load(info, sf, dir, half);
              tt.template tang_compute<false, true>(tp, sf, dir, info, half);
            }
        }
      if (lane < nsf)
        tt.template norm_compute<false, true>(zeta0, lane, infoN);
      for (int t0 = PDH_WAVE; t0 < nsf; t0 += PDH_WAVE)
        {
          const int tid = t0 + lane;
          if (tid < nsf)
            {
              int info;
              desc(tid, info);
              tt.template norm_compute<false, true>(tt.zeta_load(tid), tid, info);
            }
        }
      PDH_WAVE_SYNC();
    }
  PDHT_MARK(4);
  // ================= B2: the rows ===================================================================================
  {
    const double *Dblk = Da;
    const int m0 = L / NF;
    const int first_int = nsfb > 0 ? 1 : 0; // the boundary run, if any, is run 0
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(P.values + rbase, 0, NF * rlen * 8, 0x00020000);
    const uint32_t loff = (uint32_t)lane * 8u;
    for (int p0 = 0; p0 < rlen; p0 += PDH_WAVE)
      {
        const int p = p0 + lane;
        const bool valid = p < rlen;
        bool own;
        int c;
        if constexpr (SHIFTED)
          { // diagonal-first rows: position 0 is the diagonal entry, then the columns in ascending order without it
            own = valid && (p == 0 || (p > L && p < L + NF));
            c = p <= L ? p - 1 : p;
          }
        else
          {
            own = valid && p >= L && p < L + NF;
            c = p;
          }
        const bool cpl = valid && !own;
        const int cc = cpl ? c : 0;
        const int b = cc / NF, j = cc - b * NF;
        const int run = first_int + (b < m0 ? b : b - 1);
        const long long r0 = __double_as_longlong(rec[TERMS_HDR + (cpl ? run : 0) * TERMS_ENT]);
        const int ns = cpl ? (int)(r0 >> 32) : 0;
        const double *xb = Xa + ((int)(uint32_t)r0 - nsfb) * 3 * FULLS;
        const int dj = dig[j];
        const int q0 = (dj & 15) * N1D, q1 = ((dj >> 4) & 15) * N1D + FULLS, q2 = (dj >> 8) * N1D + 2 * FULLS;
        double acc[NF];
        for (int r = 0; r < NF; ++r)
          acc[r] = 0.0;
        for (int st = 0; __any(st < ns); ++st)
          {
            if (st < ns)
              {
                double f0[N1D], f1[N1D], f2[N1D];
                for (int k = 0; k < N1D; ++k)
                  f0[k] = xb[q0 + k], f1[k] = xb[q1 + k], f2[k] = xb[q2 + k];
                double yz[NS];
                static_for<0, N1D>([&](auto k2_) {
                  static_for<0, N1D>([&](auto k1_) {
                    constexpr int k1 = k1_, k2 = k2_;
                    if constexpr (K::pair_ok(k1, k2))
                      yz[K::pair(k1, k2)] = f1[k1] * f2[k2];
                  });
                });
                static_for<0, NF>([&](auto r_) {
                  constexpr typename K::Dig g = K::dig(r_);
                  acc[r_] += f0[g.k0] * yz[K::pair(g.k1, g.k2)];
                });
              }
            xb += 3 * FULLS;
          }
        if (__any(own))
          if (own)
            static_for<0, NF>([&](auto r_) {
              constexpr int R = r_;
              int col;
              if constexpr (SHIFTED)
                col = p == 0 ? R : (R >= p - L ? p - 1 - L : p - L);
              else
                col = p - L;
              acc[R] = Dblk[R * NF + col];
            });
        if (valid)
          {
            uint32_t rowrun = (uint32_t)p0 * 8u;
            static_for<0, NF>([&](auto r_) {
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, acc[r_]), vrs, (int)loff, (int)rowrun, 0);
              rowrun += (uint32_t)rlen * 8u;
            });
          }
      }
  }
  PDHT_MARK(5);
}
} // namespace pdht
