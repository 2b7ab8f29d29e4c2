// pdh_terms_tables.h — the PdhTerms struct shared by the term kernel (pdh_terms.h, device) and pdh_capi.cpp (host).
#pragma once
#include <stdint.h>
#define TERMS_MI 4 // most intervals of a composite 1-D rule (8 points of 2-point rules)
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
  // "Sub-face" and "cell" below are what the kernel sums over; they may be MERGED ones (pdh_capi.cpp: build_terms_tables): where the
  // cells of a polytope (the sub-faces of a plane of a run) form a tensor grid, the sum over a sub-grid of them of products of three 1-D
  // matrices is the product of the three 1-D sums - one cell (sub-face) with composite rules of several intervals per direction.
  const int64_t *sf_pt;   // [n_owned][maxsf] first own-side point (ap_* arrays) of every sub-face of a polytope, run by run (rest: 0)
  const int32_t *sf_info; // run | axis << 8 | (own outward normal along +axis) << 10 | (second tangential axis runs fastest) << 11
                          // | intervals along the first tangential axis << 12 | along the second << 15
  const int32_t *sf_ivl;  // [n_owned][maxsf][2][TERMS_MI] per tangential direction: first point of every interval's rule, relative to sf_pt
  const int32_t *cell_ivl; // [n_owned][maxcell][3][TERMS_MI] per direction: the polytope's cell (rule of vq_tensor_n^3 points) that carries
                          // the 1-D rule of every interval, -1: no such interval
  int32_t task_pts;       // most points of a 1-D composite rule (vq_tensor_n / fq_tensor_n x intervals): 4 or 8 register slots per lane task
  // The 1-D rules the kernels work from, gathered ONCE per problem on the device (pdh_terms.hip: k_terms_gather) from the point arrays
  // through the descriptors above - per owned polytope one record of tstride doubles (pdh_terms.h: terms_task_doubles):
  //   task (sub-face sf, tangential direction dir) at (2 sf + dir) * 3 tpm: coordinates [tpm] | own-side weights [tpm] | side-1 weights [tpm]
  //   task (cell c, direction d) at (2 maxsf + 3 c + d) * 3 tpm: coordinates | weights | unused          (slots behind a rule: zeros)
  //   then the plane coordinate of every sub-face [maxsf] and its descriptor sf_info [maxsf] (integer bits)
  // An assembly reads these 3 KB per polytope (block agglomerates) in one contiguous request instead of chasing descriptor -> interval
  // -> point through three dependent loads into the point arrays.
  const double *tdata;
  int32_t tstride, tpm;   // tpm = 4 or 8: slots per 1-D rule in the records = PMAX of the kernel instantiation that is launched
  int32_t maxruns;        // runs a record provides for
  int32_t maxsf, maxsi, maxcell; // most sub-faces / interior sub-faces / cells of one owned polytope
  int32_t vq_tensor_n, fq_tensor_n; // verified points per direction of the sub-cell / sub-face rules
  int32_t lds_bytes;      // dynamic LDS of a workgroup for these maxima
  int32_t split;          // 1: X tables made in a second pass over the D tables (pdh_terms.h: SPLIT) - lds_bytes is that form's
  long long *stamps;      // [n_owned][16] cycle counter at the phase boundaries; written by -DPDHT_STAMP builds only
};
