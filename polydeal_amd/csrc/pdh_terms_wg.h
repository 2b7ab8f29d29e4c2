// pdh_terms_wg.h — FE_DGQ(3) (n = 64) on agglomerates of Cartesian cells with tensor rules: a WORKGROUP per polytope.
//
// Same sums as pdh_rows.h / pdh_terms.h (reference include/poly_utils.h:2040-2084, 1870-1926), same owner-computes-rows formulation,
// same stores (whole aligned 512-byte pieces of a row through one buffer resource per polytope) - what changes is who works on a
// polytope.  pdh_rows.h gives a polytope to ONE wave; at 8 waves per CU a CU then streams into 8 regions of 458 KB at a time, the
// device into 2048, and the store-only twin of that kernel tops out at 5.2-5.5 TB/s (profiles/r03_probe_rows_store_only.txt),
// while the same stores issued by all the waves of a CU into ONE polytope's rows reach 6.5 TB/s (k_rows_shared there).  Here the W
// waves of a workgroup share one polytope:
//   A  the lane tasks of pdh_terms.h (TermTasks: 1-D matrices of every sub-cell and sub-face) spread over the waves, one kind of
//      task per wave where they fit - one round of loads for the whole polytope, then ONE workgroup barrier;
//   B  wave w owns rows [64 w / W, 64 (w + 1) / W) of every block: lane = column.  A value is a short sum of products of three
//      1-D matrix entries (pdh_terms.h), so a wave accumulates its 64 / W rows of a block in registers straight from the LDS
//      tables - no moments, no contraction stages, no MFMA, no hand-off between the waves after the barrier - and stores each row
//      piece as 512 contiguous bytes.  Diagonal-first rows: piece m <= m0 holds [carry | columns 0 .. 62]: lane l computes column
//      l - 1 (its own table offsets), lane 0 the last column of the block before (another run's tables) or, in piece 0, receives
//      the diagonal entry; the own block is computed in natural order and stored with lanes 0 .. R rotated by one position
//      (lane R, whose diagonal entry belongs to position 0 of the ROW, carries the last column of the block before).
// Status (round 4, profiles/r04_wg_variants.txt): parity green on blocks and staircase agglomerates in both layouts; 1.45-1.61 ms on
// the bench mesh where pdh_rows.h takes 1.55-1.65 on the same boxes - within the spread between two allocations of the values, so
// AUTO keeps pdh_rows.h and this kernel is taken on request (PDH_TERMS_DGQ3=1).  Without its stores it needs 1.02 ms (pdh_rows.h:
// 0.86): the term form costs this element more VALU work than the moment form + MFMA contraction (8 000 against 5 900 instructions
// per polytope), which eats what the shared store pattern gains.  A persistent variant that requests the next polytope's
// descriptors and point data during phase B (raw s_barrier hand-offs, no vmcnt(0)) was measured equal at 156 VGPRs and 15 % slower
// when held to 128: not kept.
#pragma once
#include "pdh_terms.h"

#ifndef PDHW_STORE_AUX
#define PDHW_STORE_AUX 2 // gfx940+ cache-policy bits of the row stores: 1 = sc0, 2 = nt, 16 = sc1 (nt alone: 1.48-1.52 ms where sc1|nt, the
                         // policy of pdh_rows.h, gives 1.51-1.61, profiles/r04_wg_variants.txt)
#endif

// -DPDHT_STAMP (diagnostic builds): cycle counter of wave w at four points -> stamps[slot][4 w + k] (W = 4 only)
#ifdef PDHT_STAMP
#define PDHW_MARK(k)                                                                                                  \
  do                                                                                                                  \
    {                                                                                                                 \
      const long long tm_ = (long long)__builtin_readcyclecounter();                                                  \
      if (W == 4 && lane == 0 && T.stamps)                                                                            \
        T.stamps[(int64_t)slot * 16 + 4 * wave + (k)] = tm_;                                                           \
    }                                                                                                                 \
  while (0)
#else
#define PDHW_MARK(k)
#endif

namespace pdht
{
template <int W, bool SHIFTED, int PMAX>
__global__ void __launch_bounds__(PDH_WAVE *W, 3) k_terms_wg(const PdhDev P, const PdhTerms T, const int n_owned)
{
  constexpr int N1D = 4, BASIS = 0;
  using K = Kind<N1D, BASIS>;
  constexpr int NF = K::NF, SYMS = K::SYMS, FULLS = K::FULLS;
  constexpr int RPW = 64 / W, NK1 = RPW / 4; // rows per wave; values of k1 among them (k2 is fixed per wave)
  static_assert(NF == 64 && (W == 4 || W == 8), "FE_DGQ(3), four or eight waves per polytope");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & (PDH_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = blockIdx.x; // (an XCD-chunked order of the polytopes, as in the two-kernel forms, changes nothing here: whole lines)
  if (slot >= n_owned)
    return;
  PDHW_MARK(0);
  const int REC = TERMS_HDR + T.maxruns * TERMS_ENT;
  double *rec = lds;
  double *Xa = lds + terms_rec_doubles(T.maxruns) + ((NF + 1) / 2 + ((NF + 1) / 2 & 1));
  double *Da = Xa + (T.maxsi * 3 * FULLS + ((T.maxsi * 3 * FULLS) & 1));
  // ---- header (uniform address: scalar loads), run entries -> LDS (all threads)
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
  const double *gr = T.tdata + (int64_t)slot * T.tstride; // the polytope's record of 1-D rules
  const double *gr_info = gr + (2 * T.maxsf + 3 * T.maxcell) * (3 * PMAX) + T.maxsf;
  for (int k = threadIdx.x; k < T.maxruns * TERMS_ENT; k += PDH_WAVE * W)
    rec[TERMS_HDR + k] = g[TERMS_HDR + k];
  double *Ca = Da + nsf * 3 * SYMS;
  // PMAX = slots per 1-D rule in the records (PdhTerms::tpm); rules of more than four slots are worked on by pairs of lanes (TermTasks)
  constexpr bool PAIR = PMAX > 4;
  constexpr int PL = PAIR ? PMAX / 2 : PMAX, TS = PAIR ? 2 : 1; // slots per lane, lanes per task
  using TT = TermTasks<N1D, BASIS, PL, PAIR>;
  const TT tt{P, rec, Xa, Da, Ca, lo0, lo1, lo2, ih0, ih1, ih2, nsfb, fn, tn, gr, T.maxsf, T.maxcell};
  auto desc = [&](int sf, int &info) { info = (int)__double_as_longlong(gr_info[sf < T.maxsf ? sf : 0]); };
  // ================= A ===============================================================================================
  // lane tasks, one kind per wave where they fit: waves [0, W/2): (sub-face, tangential direction[, half]) tasks; waves [W/2, W): the
  // normal-direction tasks, then - from an even lane on - the (cell, direction[, half]) tasks; each wave a contiguous, even share.
  // The point data of a wave's first round are requested before the run entries are needed.
  {
    constexpr int H = W / 2;
    const bool first_kind = wave < H;
    const int hw = first_kind ? wave : wave - H;
    const int ncs = PAIR ? (nsf + 1) & ~1 : nsf;
    const int ntask = first_kind ? 2 * nsf * TS : ncs + 3 * ncell * TS;
    const int share = ((ntask + H - 1) / H + 1) & ~1;
    const int t_begin = hw * share, t_end = t_begin + share < ntask ? t_begin + share : ntask;
    auto dec1 = [&](int tid, int &sf, int &dir, int &half) {
      const int task = tid / TS;
      half = tid - task * TS, sf = task >> 1, dir = task & 1;
    };
    auto dec2 = [&](int tid, int &ct, int &half) {
      const int u = tid - ncs;
      ct = u / TS, half = u - ct * TS;
    };
    // first round of this wave: loads now, arithmetic behind the barrier-free part (the run entries must be in LDS first)
    const int tid0 = t_begin + lane;
    const bool on0 = tid0 < t_end;
    int info0 = 0, a0 = 0, b0 = 0, h0_ = 0; // (sf, dir, half) or (cell task, -, half)
    typename TT::TPts tp0;
    typename TT::CPts cp0;
    double zeta0 = 0.0;
    for (int i = 0; i < PL; ++i)
      tp0.x[i] = tp0.ws[i] = tp0.wc[i] = cp0.x[i] = cp0.ws[i] = cp0.wc[i] = 0.0;
    tp0.npts = cp0.npts = 0;
    tp0.ws0 = tp0.wc0 = cp0.ws0 = 1.0, cp0.wc0 = 0.0;
    if (first_kind)
      {
        dec1(on0 ? tid0 : 0, a0, b0, h0_);
        desc(a0, info0);
        tp0 = tt.tang_load(info0, a0 < T.maxsf ? a0 : 0, b0, h0_);
      }
    else if (on0 && tid0 < nsf)
      {
        desc(tid0, info0);
        zeta0 = tt.zeta_load(tid0);
      }
    else if (on0 && tid0 >= ncs)
      {
        dec2(tid0, a0, h0_);
        cp0 = tt.cell_load(a0, h0_);
      }
    __syncthreads(); // run entries in LDS (nothing of this workgroup is in flight towards HBM yet: the wait costs nothing)
    if (first_kind)
      {
        if (on0)
          tt.tang_compute(tp0, a0, b0, info0, h0_);
        for (int tid = tid0 + PDH_WAVE; tid < t_end; tid += PDH_WAVE)
          {
            int info, sf, dir, half;
            dec1(tid, sf, dir, half);
            desc(sf, info);
            const typename TT::TPts tp = tt.tang_load(info, sf, dir, half);
            tt.tang_compute(tp, sf, dir, info, half);
          }
      }
    else
      {
        if (on0 && tid0 < nsf)
          tt.norm_compute(zeta0, tid0, info0);
        else if (on0 && tid0 >= ncs)
          tt.cell_compute(cp0, a0, h0_);
        for (int tid = tid0 + PDH_WAVE; tid < t_end; tid += PDH_WAVE)
          {
            if (tid < nsf)
              {
                int info;
                desc(tid, info);
                tt.norm_compute(tt.zeta_load(tid), tid, info);
              }
            else if (tid >= ncs)
              {
                int ct, half;
                dec2(tid, ct, half);
                const typename TT::CPts cp = tt.cell_load(ct, half);
                tt.cell_compute(cp, ct, half);
              }
          }
      }
  }
  __syncthreads(); // the only hand-off between the waves: from here on every wave works from the tables alone
  PDHW_MARK(1);

  // ================= B: this wave's rows of every block ===============================================================
  const int R0 = RPW * wave;                                  // first row of the wave
  const int k2 = RPW == 16 ? wave : wave >> 1;                // third digit of all its rows
  const int K1B = RPW == 16 ? 0 : 2 * (wave & 1);             // first value of the second digit (NK1 values)
  const int m0 = L >> 6, nblk = rlen >> 6;
  const int first_int = nsfb > 0 ? 1 : 0;                     // the boundary run, if any, is run 0
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(P.values + rbase, 0, NF * rlen * 8, 0x00020000);
  auto row_store = [&](double v, uint32_t lane_bytes, uint32_t row_bytes) {
#ifdef PDHW_NOSTORE
    if (P.n < 0) // (experiment: everything computed, nothing stored)
#endif
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, v), vrs, (int)lane_bytes, (int)row_bytes, PDHW_STORE_AUX);
  };
  const uint32_t row0_bytes = (uint32_t)R0 * (uint32_t)rlen * 8u, rstep = (uint32_t)rlen * 8u;
  // rows r = k0 + 4 i (i-th value of k1):  acc[r] += f0[k0] * (f1[i] * f2)
  auto add_term = [&](double (&acc)[RPW], const double (&f0)[4], const double (&f1)[NK1], double f2) __attribute__((always_inline)) {
    double yz[NK1];
    static_for<0, NK1>([&](auto i_) { yz[i_] = f1[i_] * f2; });
    static_for<0, RPW>([&](auto r_) {
      constexpr int r = r_;
      acc[r] += f0[r & 3] * yz[r >> 2];
    });
  };
  // coupling column `col` (0 .. 63) of the block of interior run `fl` (ascending block order without the own block), per lane
  // (always inlined: called from three places, and an out-of-line call would put the accumulators into scratch memory)
  auto coupling = [&](double (&acc)[RPW], int fl, int col, bool on) __attribute__((always_inline)) {
    const long long e0 = __double_as_longlong(rec[TERMS_HDR + (first_int + fl) * TERMS_ENT]);
    const int ns = on ? (int)(e0 >> 32) : 0;
    const double *xb = Xa + ((int)(uint32_t)e0 - nsfb) * 3 * FULLS;
    const int q0 = (col & 3) * 4, q1 = ((col >> 2) & 3) * 4 + FULLS + K1B, q2 = (col >> 4) * 4 + 2 * FULLS + k2;
    for (int st = 0; __any(st < ns); ++st)
      {
        if (st < ns)
          {
            double f0[4], f1[NK1];
            for (int k = 0; k < 4; ++k)
              f0[k] = xb[q0 + k];
            for (int i = 0; i < NK1; ++i)
              f1[i] = xb[q1 + i];
            add_term(acc, f0, f1, xb[q2]);
          }
        xb += 3 * FULLS;
      }
  };
  // ---- the own block, natural order (lane = column)
  double own[RPW];
  static_for<0, RPW>([&](auto r_) { own[r_] = 0.0; });
  {
    const int l0 = lane & 3, l1 = (lane >> 2) & 3, l2 = lane >> 4;
    int o0[4], o1[NK1];
    for (int k = 0; k < 4; ++k)
      o0[k] = K::sym(k, l0);
    for (int i = 0; i < NK1; ++i)
      o1[i] = K::sym(K1B + i, l1) + SYMS;
    const int o2 = K::sym(k2, l2) + 2 * SYMS;
    for (int u = 0; u < ncell; ++u)
      { // (K0' M1 M2 + M0 (K1 M2 + M1 K2))[k, l]
        const double *b = Ca + u * 6 * SYMS;
        double M0[4], K0[4], mm[NK1], km[NK1];
        const double M2 = b[o2 + 2 * SYMS], K2 = b[o2 + 3 * SYMS];
        for (int k = 0; k < 4; ++k)
          M0[k] = b[o0[k]], K0[k] = b[o0[k] + SYMS];
        for (int i = 0; i < NK1; ++i)
          {
            const double M1 = b[o1[i] + SYMS], K1 = b[o1[i] + 2 * SYMS];
            mm[i] = M1 * M2;
            km[i] = K1 * M2 + M1 * K2;
          }
        static_for<0, RPW>([&](auto r_) {
          constexpr int r = r_;
          own[r] += K0[r & 3] * mm[r >> 2] + M0[r & 3] * km[r >> 2];
        });
      }
    for (int u = 0; u < nsf; ++u)
      { // (D0 D1 D2)[k, l]
        const double *b = Da + u * 3 * SYMS;
        double f0[4], f1[NK1];
        for (int k = 0; k < 4; ++k)
          f0[k] = b[o0[k]];
        for (int i = 0; i < NK1; ++i)
          f1[i] = b[o1[i]];
        add_term(own, f0, f1, b[o2]);
      }
  }
  PDHW_MARK(2);
  if constexpr (!SHIFTED)
    {
      // ascending rows: every piece is a block, the own block among them
      for (int b = 0; b < nblk; ++b)
        {
          double acc[RPW];
          if (b == m0)
            static_for<0, RPW>([&](auto r_) { acc[r_] = own[r_]; });
          else
            {
              static_for<0, RPW>([&](auto r_) { acc[r_] = 0.0; });
              coupling(acc, b < m0 ? b : b - 1, lane, true);
            }
          uint32_t rowrun = row0_bytes + (uint32_t)b * 512u;
          static_for<0, RPW>([&](auto r_) {
            row_store(acc[r_], (uint32_t)lane * 8u, rowrun);
            rowrun += rstep;
          });
        }
    }
  else
    {
      // ---- the own block's piece m0: lanes 0 .. R rotated by one position; lane R stores position 0 of the piece - the diagonal
      // entry itself if the own block is the first of the row, else the last column of the block before, which every lane
      // computes for the wave's rows (one column: the terms of that run once)
      double dg[RPW]; // diagonal entries of the wave's rows (uniform), for position 0 of piece 0
      {
        double left63[RPW];
        static_for<0, RPW>([&](auto r_) { left63[r_] = 0.0; });
        if (m0 > 0)
          coupling(left63, m0 - 1, 63, true);
        uint32_t rowrun = row0_bytes + (uint32_t)m0 * 512u;
        static_for<0, RPW>([&](auto r_) {
          constexpr int r = r_;
          const int R = R0 + r; // uniform
          dg[r] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(own[r]), R), __builtin_amdgcn_readlane(__double2loint(own[r]), R));
          const bool diag = lane == R;
          // (an opaque copy: `c ? a[r] : b[r]` of two register arrays is turned into ONE load from a selected address, which
          // puts both arrays into scratch memory)
          double lv = left63[r];
          asm volatile("" : "+v"(lv));
          const double v = (diag && m0 > 0) ? lv : own[r];
          const uint32_t off = diag ? 0u : (lane < R ? (uint32_t)(lane + 1) * 8u : (uint32_t)lane * 8u);
          row_store(v, off, rowrun);
          rowrun += rstep;
        });
      }
      // ---- pieces left of the own block: [carry | columns 0 .. 62 of block m]
      for (int m = 0; m < m0; ++m)
        {
          double acc[RPW];
          static_for<0, RPW>([&](auto r_) { acc[r_] = 0.0; });
          const bool l0_ = lane == 0;
          // lane 0: last column of block m - 1 (piece 0: the diagonal entry, set below); the others: column lane - 1 of block m
          coupling(acc, l0_ ? (m > 0 ? m - 1 : 0) : m, l0_ ? 63 : lane - 1, !(l0_ && m == 0));
          if (m == 0)
            static_for<0, RPW>([&](auto r_) {
              double dv = dg[r_];
              asm volatile("" : "+v"(dv));
              acc[r_] = l0_ ? dv : acc[r_];
            });
          uint32_t rowrun = row0_bytes + (uint32_t)m * 512u;
          static_for<0, RPW>([&](auto r_) {
            row_store(acc[r_], (uint32_t)lane * 8u, rowrun);
            rowrun += rstep;
          });
        }
      // ---- pieces right of it: aligned blocks
      for (int b = m0 + 1; b < nblk; ++b)
        {
          double acc[RPW];
          static_for<0, RPW>([&](auto r_) { acc[r_] = 0.0; });
          coupling(acc, b - 1, lane, true);
          uint32_t rowrun = row0_bytes + (uint32_t)b * 512u;
          static_for<0, RPW>([&](auto r_) {
            row_store(acc[r_], (uint32_t)lane * 8u, rowrun);
            rowrun += rstep;
          });
        }
    }
  PDHW_MARK(3);
}
} // namespace pdht
