// pdh_terms_tables.h — the PdhTerms struct shared by the term kernel (pdh_terms.h, device) and pdh_capi.cpp (host).
#pragma once
#include <stdint.h>
struct PdhTerms
{
  // per owned polytope one record of TERMS_HDR + maxruns * TERMS_ENT doubles (pdh_terms.h):
  //   header: [0] runs | cells << 16 | boundary sub-faces << 32 (integers), [1..3] lower corner of the box, [4..6] 1 / side,
  //           [7] first value of the polytope's rows, [8] row length, [9] ascending position L of the own block,
  //           [10] first volume point, [11] sub-faces (integer)
  //   run t (a polytopal face = all sub-faces shared with one neighbour; the boundary run first, then ascending block rank):
  //           [0] first sub-face of the run in the polytope's list | sub-faces << 32, [1] block rank (-1: boundary),
  //           [2] sigma as stored per point, [3..5] lower corner of the neighbour's box, [6..8] 1 / side
  const double *meta;
  const int64_t *sf_pt;   // [n_owned][maxsf] first own-side point (ap_* arrays) of every sub-face of a polytope, run by run (rest: 0)
  const int32_t *sf_info; // run | axis << 8 | (own outward normal along +axis) << 10 | (second tangential axis runs fastest) << 11
  int32_t maxruns;        // runs a record provides for
  int32_t maxsf, maxsi, maxcell; // most sub-faces / interior sub-faces / cells of one owned polytope
  int32_t vq_tensor_n, fq_tensor_n; // verified points per direction of the sub-cell / sub-face rules
  int32_t lds_bytes;      // dynamic LDS of a workgroup for these maxima
  int32_t split;          // 1: X tables made in a second pass over the D tables (pdh_terms.h: SPLIT) - lds_bytes is that form's
  long long *stamps;      // [n_owned][16] cycle counter at the phase boundaries; written by -DPDHT_STAMP builds only
};
