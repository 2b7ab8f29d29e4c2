// pdh_combos.h — the (DIM, N1D, NT, LB) kernel instantiations that exist, with the translation-unit group
// that compiles each.  N1D = degree+1; n = dofs per polytope; T = ceil(n/4) tile rows; NT = ceil(T/4)
// fragments; LB = T - 4(NT-1) live blocks in the last fragment.
//   2-D FE_DGQ p=0..7 : n = 1,4,9,16,25,36,49,64      2-D FE_AggloDGP p=1..7 : n = 3,6,10,15,21,28,36
//   3-D FE_DGQ p=0..3 : n = 1,8,27,64                 3-D FE_AggloDGP p=1..5 : n = 4,10,20,35,56
#pragma once
#define PDH_N_GROUPS 8
#define PDH_COMBOS(X)                                                                              \
  X(0, 3, 4, 4, 4)                                                                                 \
  X(1, 3, 6, 4, 2)                                                                                 \
  X(2, 2, 8, 4, 4)                                                                                 \
  X(3, 2, 7, 4, 1) X(3, 3, 1, 1, 1) X(3, 2, 1, 1, 1)                                               \
  X(4, 3, 5, 3, 1) X(4, 3, 2, 1, 2) X(4, 3, 2, 1, 1) X(4, 2, 2, 1, 1)                              \
  X(5, 2, 6, 3, 1) X(5, 3, 3, 2, 3) X(5, 3, 3, 1, 3)                                               \
  X(6, 2, 8, 3, 1) X(6, 3, 4, 2, 1) X(6, 2, 3, 1, 3) X(6, 2, 3, 1, 2)                              \
  X(7, 2, 5, 2, 3) X(7, 2, 6, 2, 2) X(7, 2, 7, 2, 3) X(7, 2, 4, 1, 4) X(7, 2, 4, 1, 3) X(7, 2, 5, 1, 4)
