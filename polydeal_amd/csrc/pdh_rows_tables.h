// pdh_rows_tables.h — the PdhRows struct shared by the row kernel (pdh_rows.h, device) and pdh_capi.cpp (host).
#pragma once
#include <stdint.h>
struct PdhRows
{
  const int32_t *fr_ptr;   // [n_owned+1] faces of every owned polytope: boundary faces first, then ascending block rank
  const int64_t *fr_pbeg;  // first own-side point in the ap_* arrays
  const int32_t *fr_pcnt;  // number of points
  const int32_t *fr_nbr;   // neighbour polytope, -1 on the boundary
  const int32_t *fr_axis;  // normal axis c
  const int32_t *fr_blk;   // ascending rank of the neighbour's block in the row (-1 on the boundary)
  const int32_t *fr_flags; // bit 0: the point range holds points of other planes too (boundary run of a corner polytope);
                           // bit 1: tensor sub-face rules run fastest along the SECOND tangential axis
  const double *fr_coord;  // x_c of the plane
  const double *fr_sigma;  // penalty as stored per point (sigma; sigma/2 on the boundary)
  const double *fr_nsign;  // +-1: own outward normal along c
  const double *meta;      // [n_owned][ROWS_REC] per-slot records (header + face entries incl. neighbour boxes), pdh_rows.h
  int32_t fq_tensor_n;     // > 0: face points are verified tensor rules of this many points per direction on every sub-face
  int32_t vq_tensor_n;     // > 0: volume points are verified tensor rules of this many points per direction (else 0)
  int32_t tensor_only;     // 1: both kinds of rule are tensor rules and no face entry has more than 32 sub-faces
  // MULTI instantiation (pdh_rows.h): polytopes whose interface with a neighbour spans several planes ("staircase" faces of
  // METIS-like agglomerates of Cartesian cells), more than 6 interior plane entries or more than 16 entries in all
  int32_t multi;           // 1: take the MULTI instantiation
  int32_t maxe;            // face entries a record provides for (16 unless multi): record = 12 + 12 maxe doubles
  int32_t maxf;            // interior entries (coupling-moment slots) a polytope may have (6 unless multi)
  double *m2c_scratch;     // multi: [resident waves][maxf][64] coupling moments of a wave's current polytope (pdh_rows.h);
  int32_t scratch_waves;   // resident waves the scratch provides for (the launcher starts no more workgroups than that)
  int64_t scratch_stride;  // doubles per resident wave: max(maxf 64, 32 x most interior sub-faces of a polytope + 64)
  unsigned int *sched;     // [2] work counter of the persistent waves and count of the waves that have left (both zero
                           // between launches: the last wave out resets them)
  long long *stamps;       // [n_owned][16] s_memtime at the phase boundaries; written by -DPDHR_STAMP builds only
};
