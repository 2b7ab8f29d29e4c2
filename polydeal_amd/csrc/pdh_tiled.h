// pdh_tiled.h — blocks of MORE than 64 dofs per polytope: 3-D FE_DGQ(4..7) (n = 125 .. 512) and FE_AggloDGP(6, 7) (n = 84, 120), the
// degrees the reference's examples/3D_piston.cc:912-914 sweeps (degree 1 .. 6 with FE_DGQ<3>).
//
// Same sums as pdh_kernels.h (reference include/poly_utils.h:2034-2193, 1870-1926), same point records, same f64-MFMA contraction over
// the quadrature points - but an n x n block no longer fits the registers of one wave (n^2 / 64 accumulators per lane), so a block is
// cut into (i, j) TILES of 64 x 64 basis functions and one wave computes one tile:
//   k_tdiag    : tile (ti <= tj) of the own block A[P,P] (two launches: the symmetric tiles ti == tj, then the pairs ti < tj): sum_q grad phi_i . grad phi_j JxW (+ c phi_i phi_j JxW) over the volume points
//                plus the own-side SIP terms over all face points of P, rows i in tile ti against columns j in tile tj; written into the
//                rows of tile ti and, for ti < tj, transposed into the rows of tile tj (the block is symmetric).
//   k_toffdiag : tile (ti, tj) of the coupling block A[P,Q] of one interior face, P's functions of tile ti against Q's of tile tj;
//                written into P's rows and transposed into Q's rows when those are owned here (M21 = M12^T, as k_offdiag).
// A tile is the NT = 4, LB = 4 product of pdh_kernels.h with the FULL (non-symmetric) schedule; the lane's functions are
// i0 + 16 a + (lane & 15), functions past n point at the zero pair of the point record (their tiles stay zero and are not stored).
// The point records do not depend on the tile (all N1D 1-D values of a point), every tile re-evaluates them: at these n the MFMA work per
// point (192 / 128 instructions per 4 points) dwarfs the 2 DIM N1D Horner evaluations.
// Owner-computes-rows, every value written once, no atomics - as the rest of the library.
#pragma once
#include "pdh_kernels.h"

namespace pdht2
{
using namespace pdh;

template <int DIM, int N1D>
__device__ __forceinline__ void init_tile_basis(LaneBasis<DIM, N1D, 4, 4> &lb, const PdhDev &P, int lane, int i0)
{
  const int m = lane & 15;
  static_for<0, 4>([&](auto a_) {
    constexpr int a = a_;
    const int i = i0 + 16 * a + m;
    const uint32_t packed = (i < P.n) ? (uint32_t)P.midx[i] : 0xffffffffu;
    const bool ok = packed != 0xffffffffu;
    for (int c = 0; c < DIM; ++c)
      lb.off[a][c] = ok ? (c * N1D + (int)((packed >> (8 * c)) & 0xff)) * 16 : Rec<DIM, N1D>::ZERO_OFF;
  });
}

// Rows [16 s, 16 s + 16) of a tile from the LDS strip to their CSR positions.  The strip row r holds the block row i0 + 16 s + r, its
// column c the block column j0 + c; pos0 = position of the block's column 0 inside the row.  DIAG: own block in deal.II's
// diagonal-first layout (the diagonal entry at position 0, the columns in front of it one further right).
template <bool DIAG>
__device__ __forceinline__ void store_tile_strip(double *values, int64_t base, int row_len, int pos0, int diag_first, const double *strip,
                                                 int ncol_pad, int s, int i0, int j0, int n, int lane)
{
  const int C = j0 + lane;
  if (C >= n)
    return;
  const double *src = strip + lane;
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
          const int R = i0 + 16 * s + r0 + k;
          int pos = pos0 + C;
          if constexpr (DIAG)
            if (diag_first)
              pos = (C == R) ? 0 : (pos0 + C + (C < R ? 1 : 0));
          if (R < n)
            values[base + (int64_t)R * row_len + pos] = v[k];
        }
    }
}

// SYM: the accumulators hold the upper tile pairs of a symmetric tile (ti == tj of the own block): mirrored while the strip is filled
template <bool DIAG, bool TRANSPOSE, bool SYM = false>
__device__ __forceinline__ void write_tile(const double *acc, double *strip, double *values, int64_t base, int row_len, int pos0,
                                           int diag_first, int i0, int j0, int n, int lane)
{
  constexpr int ncol_pad = 66;
  StripMap<4, 4> sm;
  sm.init(lane, ncol_pad);
  static_for<0, 4>([&](auto s_) {
    constexpr int s = s_;
    __syncthreads();
    fill_strip<4, 4, SYM, TRANSPOSE, s, true>(acc, strip, ncol_pad, sm, 64);
    __syncthreads();
    store_tile_strip<DIAG>(values, base, row_len, pos0, diag_first, strip, ncol_pad, s, i0, j0, n, lane);
  });
}

// ------------------------------------------------------------------------------------------------
// Own block: one wave per (owned polytope, tile pair ti <= tj).
// ------------------------------------------------------------------------------------------------
// DIAGT: the ntile tiles ti == tj of a polytope - symmetric, so only the upper 4 x 4-tile pairs are multiplied (the schedule of k_diag:
// 40 instead of 64 MFMA per product) and mirrored in the epilogue; else the ntile (ntile - 1) / 2 pairs ti < tj with the full schedule.
template <int DIM, int N1D, bool REACT, bool DIAGT>
__global__ void __launch_bounds__(PDH_WAVE, 2) k_tdiag(const PdhDev P, const int n_owned, const int ntile)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using RC = Rec<DIM, N1D>;
  const int lane = threadIdx.x;
  const int npairs = DIAGT ? ntile : ntile * (ntile - 1) / 2;
  if ((int)blockIdx.x >= n_owned * npairs)
    return;
  // (an XCD per contiguous eighth of the work items: the tiles of a block and the blocks of a row - n is no multiple of 16, so all of
  // them share 128-byte lines with their neighbours - are then written from one L2; pdh_kernels.h: xcd_chunked)
  const int bid = xcd_chunked((int)blockIdx.x, n_owned * npairs);
  const int slot = bid / npairs;
  int pr = bid - slot * npairs, ti = 0;
  if constexpr (!DIAGT)
    while (pr >= ntile - 1 - ti) // pairs (ti, ti + 1), (ti, ti + 2), ...
      {
        pr -= ntile - 1 - ti;
        ++ti;
      }
  else
    ti = pr;
  const int tj = DIAGT ? ti : ti + 1 + pr;
  const int i0 = 64 * ti, j0 = 64 * tj;
  const int agg = P.own_agg[slot];
  constexpr int CH = PDH_WAVE;
  double *rec = lds;                // [CH][RC::LEN]
  double *aux = lds + CH * RC::LEN; // [CH][2+DIM]: (unused), sigma/2
  constexpr int AUXN = 2 + DIM;

  double lo[DIM], h[DIM];
  for (int c = 0; c < DIM; ++c)
    {
      lo[c] = P.bbox[(int64_t)agg * 2 * DIM + c];
      h[c] = P.bbox[(int64_t)agg * 2 * DIM + DIM + c] - lo[c];
    }
  LaneBasis<DIM, N1D, 4, 4> lbI, lbJ;
  init_tile_basis<DIM, N1D>(lbI, P, lane, i0);
  init_tile_basis<DIM, N1D>(lbJ, P, lane, j0);
  Rotator rt;
  rt.init(lane);
  double acc[64];
  for (int t = 0; t < 64; ++t)
    acc[t] = 0.0;
  const int kq = lane >> 4;

  // ---- volume term ---------------------------------------------------------------------------
  {
    const int64_t qb = P.vq_ptr[slot], qe = P.vq_ptr[slot + 1];
    for (int64_t base = qb; base < qe; base += CH)
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
          eval_point_record<DIM, N1D, false>(P.tab, lo, h, x, sqrt(w), nullptr, rec + lane * RC::LEN);
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        for (int step = 0; step < nsteps; ++step)
          {
            const char *rb = reinterpret_cast<const char *>(rec + (4 * step + kq) * RC::LEN);
            double phiI[4], phiJ[4], dI[4][DIM], dJ[4][DIM];
            static_for<0, 4>([&](auto a_) {
              constexpr int a = a_;
              frag_eval<DIM>(rb, lbI.off[a], phiI[a], dI[a]);
              frag_eval<DIM>(rb, lbJ.off[a], phiJ[a], dJ[a]);
            });
            static_for<0, DIM>([&](auto c_) {
              constexpr int c = c_;
              double GI[4], GJ[4];
              static_for<0, 4>([&](auto a_) {
                constexpr int a = a_;
                GI[a] = dI[a][c];
                GJ[a] = dJ[a][c];
              });
              // sum_q (sqrt(w) d_c phi_i)(sqrt(w) d_c phi_j)
              if constexpr (DIAGT)
                {
                  RotSet<4> RG;
                  make_rot<4, ROT_SYM>(GI, rt, RG);
                  product<4, 4, true>(acc, RG, RG);
                }
              else
                product_full<4, 4>(acc, GI, GJ, rt);
            });
            if constexpr (REACT)
              {
                double RI[4];
                static_for<0, 4>([&](auto a_) {
                  constexpr int a = a_;
                  RI[a] = P.reaction_c * phiI[a];
                });
                if constexpr (DIAGT)
                  {
                    RotSet<4> RP, RA;
                    make_rot<4, ROT_SYM>(phiI, rt, RP);
                    make_rot<4, ROT_SYM>(RI, rt, RA);
                    product<4, 4, true>(acc, RA, RP);
                  }
                else
                  product_full<4, 4>(acc, RI, phiJ, rt);
              }
          }
      }
  }

  // ---- own-side face terms (all faces of the polytope, boundary included) ----------------------
  {
    const int64_t pb = P.ap_ptr[slot], pe = P.ap_ptr[slot + 1];
    for (int64_t base = pb; base < pe; base += CH)
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
          double ds[DIM]; // -n_c / 2 folded into the derivative entries
          for (int c = 0; c < DIM; ++c)
            ds[c] = -0.5 * nr[c];
          eval_point_record<DIM, N1D, true>(P.tab, lo, h, x, sqrt(w), ds, rec + lane * RC::LEN);
          aux[lane * AUXN + 1] = 0.5 * sg;
        }
        __syncthreads();
        const int nsteps = (cnt + 3) >> 2;
        for (int step = 0; step < nsteps; ++step)
          {
            const int pt = 4 * step + kq;
            const char *rb = reinterpret_cast<const char *>(rec + pt * RC::LEN);
            const double hs = aux[pt * AUXN + 1];
            // Phi = sqrt(w) phi,  U = sqrt(w) (-1/2 grad phi . n + sigma/2 phi)
            double PhiI[4], UI[4], PhiJ[4], UJ[4];
            static_for<0, 4>([&](auto a_) {
              constexpr int a = a_;
              FragRaw<DIM> r;
              r.load(rb, lbI.off[a]);
              r.eval_u(hs, PhiI[a], UI[a]);
              r.load(rb, lbJ.off[a]);
              r.eval_u(hs, PhiJ[a], UJ[a]);
            });
            if constexpr (DIAGT)
              {
                RotSet<4> RU, RPhi;
                make_rot<4, ROT_SYM>(UI, rt, RU);
                make_rot<4, ROT_SYM>(PhiI, rt, RPhi);
                product<4, 4, true>(acc, RU, RPhi);
                product<4, 4, true>(acc, RPhi, RU);
              }
            else
              {
                product_full<4, 4>(acc, UI, PhiJ, rt);
                product_full<4, 4>(acc, PhiI, UJ, rt);
              }
          }
      }
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  double *strip = lds; // overlays the point records
  const int64_t rbase = P.row_base[slot];
  const int rlen = P.row_len[slot];
  const int L = P.diag_L[slot];
  if constexpr (DIAGT)
    write_tile<true, false, true>(acc, strip, P.values, rbase, rlen, L, P.diag_first, i0, j0, P.n, lane);
  else
    {
      write_tile<true, false>(acc, strip, P.values, rbase, rlen, L, P.diag_first, i0, j0, P.n, lane);
      write_tile<true, true>(acc, strip, P.values, rbase, rlen, L, P.diag_first, j0, i0, P.n, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// Coupling block: one wave per (interior face with an owned side, tile (ti, tj)).
// ------------------------------------------------------------------------------------------------
template <int DIM, int N1D>
__global__ void __launch_bounds__(PDH_WAVE, 2) k_toffdiag(const PdhDev P, const int n_items, const int ntile)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using RC = Rec<DIM, N1D>;
  const int lane = threadIdx.x;
  const int nt2 = ntile * ntile;
  if ((int)blockIdx.x >= n_items * nt2)
    return;
  const int bid = xcd_chunked((int)blockIdx.x, n_items * nt2);
  const int item = bid / nt2;
  const int pr = bid - item * nt2;
  const int ti = pr / ntile, tj = pr - ti * ntile;
  const int i0 = 64 * ti, j0 = 64 * tj;
  const int slot = P.it_own[item];
  const int agg = P.own_agg[slot];
  const int nbr = P.it_nbr[item];
  constexpr int CH = 32; // lanes 0-31: the records in P's frame, lanes 32-63: the same points in Q's frame
  double *recP = lds;
  double *recQ = lds + CH * RC::LEN;
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
  LaneBasis<DIM, N1D, 4, 4> lbI, lbJ;
  init_tile_basis<DIM, N1D>(lbI, P, lane, i0);
  init_tile_basis<DIM, N1D>(lbJ, P, lane, j0);
  Rotator rt;
  rt.init(lane);
  double acc[64];
  for (int t = 0; t < 64; ++t)
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
          double A1[4], A2[4], B1[4], B2[4];
          static_for<0, 4>([&](auto a_) {
            constexpr int a = a_;
            FragRaw<DIM> r;
            r.load(rbP, lbI.off[a]);
            // A2 = sqrt(w) phi^P,  A1 = sqrt(w) (1/2 grad phi^P . n_P - sigma phi^P)
            r.eval_u(msg, A2[a], A1[a]);
            r.load(rbQ, lbJ.off[a]);
            // B1 = sqrt(w) phi^Q,  B2 = sqrt(w) (-1/2 grad phi^Q . n_P)
            r.eval_u(0.0, B1[a], B2[a]);
          });
          product_full<4, 4>(acc, A1, B1, rt);
          product_full<4, 4>(acc, A2, B2, rt);
        }
    }

  double *strip = lds;
  write_tile<false, false>(acc, strip, P.values, P.row_base[slot], P.row_len[slot], P.it_pos[item], 0, i0, j0, P.n, lane);
  // A[Q,P] = A[P,Q]^T, written into Q's rows when this context owns them
  const int qslot = P.it_nbr_slot[item];
  if (qslot >= 0)
    write_tile<false, true>(acc, strip, P.values, P.row_base[qslot], P.row_len[qslot], P.it_pos_t[item], 0, j0, i0, P.n, lane);
}

inline size_t lds_bytes_tdiag(int dim, int n1d)
{
  const size_t recs = (size_t)PDH_WAVE * (dim * n1d * 2 + 2 + 2 + dim) * sizeof(double);
  const size_t strip = (size_t)16 * 66 * sizeof(double);
  return recs > strip ? recs : strip;
}
inline size_t lds_bytes_toffdiag(int dim, int n1d)
{
  const size_t recs = (size_t)32 * (2 * (dim * n1d * 2 + 2) + 2 + dim) * sizeof(double);
  const size_t strip = (size_t)16 * 66 * sizeof(double);
  return recs > strip ? recs : strip;
}
} // namespace pdht2
