/* polydeal_hip.h — C ABI of the MI355X (gfx950) SIP assembly path.
 *
 * This is the drop-in boundary for ONE hot path of polyDEAL: the DG/SIP system-matrix assembly
 * over agglomerated polytopal elements, i.e. what
 *
 *   PolyUtils::assemble_dg_matrix(MatrixType&, const FiniteElement<dim>&, const AgglomerationHandler<dim>&)
 *       reference include/poly_utils.h:2000-2195
 *   and the hand-written loops of examples/poisson.cc:694-988, examples/minimal_SIP.cc:143-365,
 *   examples/diffusion_reaction.cc:420-696
 *
 * do on the CPU through AgglomerationHandler::reinit / reinit_interface
 * (reference source/agglomeration_handler.cc:729-906, 1103-1243), MappingBox
 * (source/mapping_box.cc:393-531) and FE_DGQ / FE_AggloDGP (source/fe_agglodgp.cc:27-55).
 *
 * The reference has no FFI layer (it is a set of C++ templates over deal.II types); the entry
 * points below are what a deal.II-side adapter binds (INTEGRATION.md shows it).  Plain pointers and
 * sizes only, no exceptions cross the boundary: every function returns 0 on success or a negative
 * PDH_E* code, and pdh_last_error() gives the message.  One pdh_ctx per device and per host thread
 * (not re-entrant, like the reference's handler: include/agglomeration_handler.h:834-851).
 *
 * All arithmetic is fp64.  Indices are 32-bit except CSR row pointers / point offsets (64-bit).
 */
#ifndef POLYDEAL_HIP_H
#define POLYDEAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDH_BASIS_DGQ 0      /* FE_DGQ<dim>(p): tensor Lagrange on Gauss-Lobatto nodes, (p+1)^dim dofs   */
#define PDH_BASIS_AGGLODGP 1 /* FE_AggloDGP<dim>(p): Legendre P_p, C(p+dim,dim) dofs (fe_agglodgp.cc:27) */

#define PDH_OK 0
#define PDH_EINVAL -1     /* malformed problem description                */
#define PDH_EUNSUPPORTED -2 /* valid but outside what the kernels implement */
#define PDH_EDEVICE -3    /* HIP runtime error (no GPU, OOM, launch failure) */
#define PDH_ESTATE -4     /* call order (e.g. assemble before set_problem)  */

typedef struct pdh_ctx pdh_ctx; /* opaque; owns all device memory */

/* Flattened description of one agglomerated mesh + SIP variant.  All pointers are caller-owned host
 * memory, read-only, and need to stay valid only for the duration of the call they are passed to.
 * Letters in brackets refer to SURVEY.md section 8(a).                                            */
typedef struct pdh_problem
{
  int32_t dim;     /* 2 or 3                                                                       */
  int32_t degree;  /* polynomial degree p, 0 .. 7.  Polytopes of up to 64 dofs run one wavefront per
                      block; more (3-D only: FE_DGQ(4..7), FE_AggloDGP(6, 7)) in 64 x 64 tiles
                      (pdh_tiled.h; owner-computes-rows only, no ghost-block exchange)             */
  int32_t basis;   /* PDH_BASIS_*                                                                  */
  int32_t n_agg;   /* number of polytopes (agglomerates) in this description                       */
  int32_t n_faces; /* polytopal faces; each interior face stored ONCE (in/out), boundary: out = -1 */
  int32_t n_rows;  /* GLOBAL number of dofs (= n_dofs_per_cell * n_agg for a global description)   */
  int32_t diag_first; /* 1: deal.II SparsityPattern row layout (diagonal first, then ascending);
                         0: plain ascending columns (Epetra local CSR)                       [A4] */
  int32_t local;   /* 0: GLOBAL description - all polytopes, rowptr [n_rows+1] over all rows.
                      1: RANK-LOCAL description, what an MPI rank of the reference holds
                         (source/agglomeration_handler.cc:1026-1091: locally owned polytopes + the ghost
                         polytopes across its partition boundary, with their bounding boxes and GLOBAL dof
                         indices).  Polytopes are numbered locally 0..n_agg-1: the owned ones - exactly those
                         whose dof_offset lies in [row_begin,row_end) - and at least every neighbour of an owned
                         one.  dof_offset holds GLOBAL dof numbers; ghosts need bbox and dof_offset only (their
                         volume range may be empty); every face with an owned side must be listed, with the
                         quadrature data of side 0 as always; faces between two ghosts are ignored.
                         rowptr has row_end - row_begin + 1 entries (owned rows only, rowptr[0] = 0), colind
                         - if given - global column numbers of those rows.                              */
  double reaction_c; /* adds c * phi_i phi_j to the volume term (diffusion_reaction.cc:495-501)    */

  const double  *bbox;       /* [n_agg][2][dim] lower, upper corner of the bounding box      [A1] */
  const int32_t *dof_offset; /* [n_agg] first global dof of each polytope                    [A2] */

  const int64_t *vq_ptr; /* [n_agg+1] CSR offsets into the volume quadrature arrays           [A5] */
  const double  *vq_x;   /* [dim][Nq_tot] REAL quadrature points (structure of arrays)             */
  const double  *vq_w;   /* [Nq_tot] JxW of the sub-cell rules (agglomeration_handler.cc:639-653)  */

  const int32_t *face_in;  /* [n_faces] polytope on side 0                                    [A3] */
  const int32_t *face_out; /* [n_faces] polytope on side 1, or -1 on the domain boundary           */
  const int64_t *fq_ptr;   /* [n_faces+1] CSR offsets into the face quadrature arrays         [A8] */
  const double  *fq_x;     /* [dim][Nqf_tot] REAL points                                           */
  const double  *fq_n;     /* [dim][Nqf_tot] outward unit normal of side 0 (poly_utils.h:1881)     */
  const double  *fq_w;     /* [Nqf_tot] JxW seen from side 0                                       */
  const double  *fq_w_out; /* [Nqf_tot] JxW seen from side 1, or NULL if identical     */
  const double  *face_sigma; /* [n_faces] penalty sigma = C / h_f, resolved per caller variant     */

  const int64_t *rowptr; /* [n_rows+1] (local: [row_end-row_begin+1]) target CSR row pointers    [A4] */
  const int32_t *colind; /* [nnz] target CSR columns, or NULL: canonical DG block pattern assumed  */

  /* Optional (NULL = dof_offset): column number of the first dof of every polytope in the numbering that ORDERS the
   * entries of a row.  An Epetra_CrsMatrix row is sorted by LOCAL column id, and the column map lists the owned
   * columns first and the ghost columns after them, so on rank > 0 a ghost block with a smaller global number sits
   * BEHIND the owned blocks of the row (TrilinosWrappers::SparseMatrix, reference examples/diffusion_reaction.cc:
   * 448-466).  With col_offset = local column ids the values come out in exactly that order (colind, if given, is
   * then checked against these numbers).  Requires diag_first = 0.                                              */
  const int32_t *col_offset;
  /* Optional: owning rank of every polytope (the calling rank's own number for the owned ones).  Needed only by the
   * ghost-block exchange variant (pdh_set_exchange_mode), which ships the M21/M22 blocks of the faces cut by the
   * partition to the rank that owns their rows (reference include/poly_utils.h:1930-1992, 2134-2194).             */
  const int32_t *agg_rank;
  /* Structure of the volume rules.  The volume points of every polytope may come in consecutive groups of
   * vq_tensor_n^dim points, each group a tensor-product rule on an axis-aligned box, first index fastest:
   *   x_(i,j,k) = (X_i, Y_j, Z_k),  JxW_(i,j,k) = a_i b_j c_k        (QGauss<dim>(n) on Cartesian sub-cells,
   *   source/agglomeration_handler.cc:639-653 with MappingCartesian-like cells).
   * > 0: a claim, which pdh_set_problem VERIFIES on the points (a few ulp); 0: pdh_set_problem finds out by itself
   * (n = 8 .. 2 are tried; a candidate that does not hold fails on the first group); < 0: do not look.  Where the
   * structure holds for all owned polytopes the row kernel integrates the volume moments cell by cell in factorised form
   * (3 n sums instead of n^3 points).  A wrong claim is harmless: the check fails and the general path is taken.   */
  int32_t vq_tensor_n;
  /* The same for the face points: groups of fq_tensor_n^(dim-1) points, each a tensor rule on an axis-aligned rectangle
   * (either tangential direction may run fastest).  Verified on the points like vq_tensor_n.                        */
  int32_t fq_tensor_n;
} pdh_problem;

/* Lifetime -------------------------------------------------------------------------------------- */
int pdh_create(pdh_ctx **out, int device_id);
void pdh_destroy(pdh_ctx *ctx);
const char *pdh_last_error(const pdh_ctx *ctx); /* valid until the next call on ctx; ctx may be NULL */

/* Setup: validates the description, derives the per-polytope block positions inside the CSR rows and
 * uploads the repacked tables to HBM.  Plays the role of the caches AgglomerationHandler builds in
 * distribute_agglomerated_dofs / initialize_fe_values (agglomeration_handler.cc:210-236, 326-379).
 * [row_begin,row_end) selects the dof rows this context owns (multi-GPU: one contiguous range per
 * rank, whole polytopes only, like the reference asserts at agglomeration_handler.cc:83-87).     */
/* With problem->local = 1 the description itself is rank-local (owned + ghost polytopes, global dof numbers): a
 * rank never needs the global mesh.  With local = 0 every rank passes the same global description and its range. */
int pdh_set_problem(pdh_ctx *ctx, const pdh_problem *problem);
int pdh_set_problem_local(pdh_ctx *ctx, const pdh_problem *problem, int32_t row_begin, int32_t row_end);

/* Set-up from a COMPACT description of agglomerates of CARTESIAN cells: the quadrature data of the problem are generated ON THE
 * DEVICE instead of being gathered on the host and uploaded.  The reference gathers them per polytope inside the span it times
 * (source/agglomeration_handler.cc:622-707 agglomerated_quadrature, :1103-1243 reinit_master; examples/poisson.cc:1099-1106); for
 * sub-cells that are axis-aligned boxes with QGauss rules - every BASELINE configuration but the piston mesh - a group of nq^3
 * volume points is a function of its cell's box, a group of nqf^2 face points of its cell's box and local face number.
 *   problem : as for pdh_set_problem_local (global or rank-local), with vq_x, vq_w, fq_x, fq_n, fq_w, fq_w_out = NULL; dim = 3;
 *             vq_ptr / fq_ptr count points as always (nq^3 per sub-cell, nqf^2 per sub-face); vq_tensor_n / fq_tensor_n ignored
 *   points  : where the points come from.  Volume group g (the g-th group of nq^3 points in the order of vq_ptr) is QGauss<3>(nq)
 *             on cell vq_cell[g] (x fastest); face group s (order of fq_ptr) is QGauss<2>(nqf) on local face fq_face[s] (deal.II
 *             numbering 2 * axis + side) of cell fq_cell[s] - the sub-cell on side 0 of the polytopal face, whose outward normal
 *             is the face's normal (reference include/poly_utils.h:1881) - lower tangential axis fastest.
 * The generated arrays are exactly what the caller would have passed (to rounding of lo + h * xi), so everything downstream is
 * unchanged; the assembly runs through the term kernels (PDH_ROWS_TERMS: a problem whose polytopes are too large for them is
 * refused with PDH_EUNSUPPORTED - describe it with points then).                                                              */
typedef struct pdh_cartesian_points
{
  int32_t n_cells;          /* cells of the background grid referred to below                                   */
  const double *cell_box;   /* [n_cells][2][3] lower, upper corner                                              */
  const int32_t *vq_cell;   /* [vq_ptr[n_agg] / nq^3] cell of every group of volume points                      */
  const int32_t *fq_cell;   /* [fq_ptr[n_faces] / nqf^2] side-0 cell of every sub-face                          */
  const int32_t *fq_face;   /* [same] its local face number 0 .. 5                                              */
  int32_t nq, nqf;          /* points per direction of the cell / face rule (1 .. 8)                            */
} pdh_cartesian_points;
int pdh_set_problem_cartesian(pdh_ctx *ctx, const pdh_problem *problem, const pdh_cartesian_points *points, int32_t row_begin,
                              int32_t row_end);

/* The hot path.  Replaces the body of assemble_dg_matrix (poly_utils.h:2034-2193): volume term,
 * boundary (Nitsche) term and the four interface blocks, written straight into CSR value order.
 *   pdh_assemble_device : values stay in HBM (pointer from pdh_device_values); asynchronous on the
 *                         context's stream; nothing crosses PCIe.
 *   pdh_assemble        : same + copy of the owned rows' values into `values`
 *                         (length rowptr[row_end]-rowptr[row_begin]); overwrites, does not add.
 *   pdh_assemble_sip    : set_problem + assemble in one call (all rows).                           */
int pdh_assemble_device(pdh_ctx *ctx);
int pdh_assemble(pdh_ctx *ctx, double *values);
int pdh_assemble_sip(pdh_ctx *ctx, const pdh_problem *problem, double *values);
int pdh_assemble_sip_local(pdh_ctx *ctx, const pdh_problem *problem, int32_t row_begin, int32_t row_end,
                           double *values);

/* Right-hand side of the SIP Poisson problem for the resident problem, i.e. what the
 * callers assemble in the same loops as the matrix (examples/poisson.cc:745-759, 788-828):
 *   rhs_i(P) = sum_q phi_i f(x_q) JxW  +  sum_{q on boundary faces} (sigma g phi_i - grad phi_i . n g) JxW
 * f_vol : [Nq_tot]  f sampled at the volume quadrature points, in the order of vq_x; NULL = no volume term
 * g_bdry: [Nqf_tot] Dirichlet datum sampled at the face quadrature points, in the order of fq_x (only the
 *         entries of boundary faces are read); NULL = homogeneous
 * rhs   : [row_end-row_begin] host output for the owned rows, overwritten.  Rows are in dof order.      */
int pdh_assemble_rhs(pdh_ctx *ctx, const double *f_vol, const double *g_bdry, double *rhs);

/* Evaluation of a polytopal DG function at caller-given points: the device part of
 * PolyUtils::interpolate_to_fine_grid (include/poly_utils.h:1145-1274: points = support points of the sub-cells)
 * and PolyUtils::compute_global_error (:1686-1731: points = quadrature points; the JxW-weighted sums stay with the
 * caller).  solution: coefficients of the owned rows [row_end-row_begin]; pt_ptr [n_agg+1]: CSR offsets of the
 * points of every polytope; pts [dim][N] real coordinates; u [N] (required) and grad [dim][N] (may be NULL) receive
 * u_h and grad u_h at the points of the polytopes owned by this context (others are left untouched).        */
int pdh_evaluate(pdh_ctx *ctx, const double *solution, const int64_t *pt_ptr, const double *pts, double *u, double *grad);

/* Values of all basis functions of a set of bounding boxes at caller-given points - the local matrices of
 * Utils::fill_injection_matrix (include/utils.h:219-229: local_matrix(i,j) = fe.shape_value(j,
 * coarse_bbox.real_to_unit(real_qpoints[i]))); needs no resident problem.  bbox [n_boxes][2][dim] (lower corner,
 * upper corner); pt_ptr [n_boxes+1] CSR offsets of the points of every box; pts [dim][N] real coordinates;
 * values [N][n] row-major, n = dofs per cell of (dim, degree, basis).                                         */
int pdh_shape_values(pdh_ctx *ctx, int dim, int degree, int basis, int n_boxes, const double *bbox,
                     const int64_t *pt_ptr, const double *pts, double *values);

/* Device-resident variants of the three calls above: every pointer is DEVICE memory, nothing is allocated or copied,
 * the kernels are queued on pdh_stream() and the call returns (pdh_synchronize waits).  Arrays are in the CALLER's order,
 * exactly as for the host variants: d_f_vol [Nq_tot] / d_g_bdry [Nqf_tot] sampled at vq_x / fq_x of the resident
 * description (NULL = absent), d_rhs [row_end-row_begin]; d_pt_ptr [n_agg+1], d_pts [dim][n_points] (component stride
 * n_points), d_u [n_points], d_grad [dim][n_points] or NULL - only the points of polytopes owned by the context are written.
 * The host variants are these plus the transfers through grow-only scratch buffers of the context.              */
int pdh_assemble_rhs_device(pdh_ctx *ctx, const double *d_f_vol, const double *d_g_bdry, double *d_rhs);
int pdh_evaluate_device(pdh_ctx *ctx, const double *d_solution, const int64_t *d_pt_ptr, const double *d_pts, int64_t n_points,
                        double *d_u, double *d_grad);
int pdh_shape_values_device(pdh_ctx *ctx, int dim, int degree, int basis, int n_boxes, const double *d_bbox,
                            const int64_t *d_pt_ptr, const double *d_pts, int64_t n_points, double *d_values);

/* PolyUtils::compute_global_error (reference include/poly_utils.h:1647-1750) with the weighted sums formed ON THE DEVICE:
 *   sums[0] = sum_q JxW_q (u(x_q) - u_h(x_q))^2,   sums[1] = sum_q JxW_q |grad u(x_q) - grad u_h(x_q)|^2
 * over the points of the polytopes owned by the context (the SQUARES: ranks add them before the root, like the reference's
 * Utilities::MPI::sum, :1736-1745).  pt_ptr / pts as for pdh_evaluate; w [N] the JxW of the points; exact_u [N] and
 * exact_grad [dim][N] the analytical solution and its gradient sampled by the caller at the same points (the library has no
 * callbacks into host code).  One kernel evaluates u_h, grad u_h and the two sums per polytope (fixed summation order:
 * the result is reproducible); 16 bytes per polytope cross PCIe instead of 8 (dim + 1) per point.  sums is HOST memory in
 * both variants; the _device variant takes every array in device memory and synchronises the stream before it returns. */
int pdh_global_error(pdh_ctx *ctx, const double *solution, const int64_t *pt_ptr, const double *pts, const double *w,
                     const double *exact_u, const double *exact_grad, double *sums /* [2] */);
int pdh_global_error_device(pdh_ctx *ctx, const double *d_solution, const int64_t *d_pt_ptr, const double *d_pts, int64_t n_points,
                            const double *d_w, const double *d_exact_u, const double *d_exact_grad, double *sums /* [2], host */);

/* Access to device-resident results and synchronisation. */
int pdh_device_values(pdh_ctx *ctx, double **device_ptr, int64_t *n_values);
int pdh_synchronize(pdh_ctx *ctx);
void *pdh_stream(pdh_ctx *ctx); /* hipStream_t the kernels are launched on */

/* Multi-GPU, two realisations of the one exchange step of the path:
 *   PDH_EXCHANGE_NONE  (default) owner-computes-rows: a face cut by the partition is contracted on BOTH ranks, each writing
 *                      only its own rows - no matrix traffic at all (the reference's compress(VectorOperation::add),
 *                      include/poly_utils.h:2194, disappears).
 *   PDH_EXCHANGE_GHOST the reference's scheme (include/poly_utils.h:1930-1992, 2134-2194; examples/diffusion_reaction.cc:
 *                      615-692): the rank owning side 0 of a cut face assembles all four blocks, keeps M11/M12 and ships
 *                      M21 (one n x n block per face) and M22 (summed per remote polytope) to the owner of those rows.
 *                      Needs problem->agg_rank.  A step is then
 *                         pdh_assemble_device(ctx);                 owned rows minus the foreign faces + outgoing blocks
 *                         pdh_exchange_get_send(ctx, d_send);       [sum send_count] doubles, peer by peer in rank order
 *                         <caller moves them: grouped ncclSend/ncclRecv or all-to-all-v over RCCL/xGMI, d_send -> peers' d_recv>
 *                         pdh_exchange_apply(ctx, d_recv);          stores the M21 blocks, adds the M22 sums
 *                      The library stays transport-free (no RCCL/MPI dependency in the C ABI); bench.py drives it with
 *                      torch.distributed (RCCL).  Both ranks derive the block order of a peer buffer from the global dof
 *                      numbers of the cut faces, so no index lists are exchanged.  Results equal PDH_EXCHANGE_NONE up to
 *                      rounding (1e-14 relative).  Mode changes take effect at the next pdh_set_problem*.              */
#define PDH_EXCHANGE_NONE 0
#define PDH_EXCHANGE_GHOST 1
int pdh_set_exchange_mode(pdh_ctx *ctx, int mode);
int pdh_exchange_layout(pdh_ctx *ctx, int n_ranks, int64_t *send_count /* [n_ranks] doubles */, int64_t *recv_count);
int pdh_exchange_get_send(pdh_ctx *ctx, double *d_send /* device memory */);
int pdh_exchange_apply(pdh_ctx *ctx, const double *d_recv /* device memory */);
/* Launch everything on a stream of the caller (hipStream_t; NULL = the context's own): collectives issued on that
 * stream are then ordered with the kernels without host synchronisation.                                       */
int pdh_set_stream(pdh_ctx *ctx, void *stream);
/* Host-only (no GPU): the per-peer sizes pdh_exchange_layout would report for this description and row range. */
int pdh_check_exchange(const pdh_problem *problem, int32_t row_begin, int32_t row_end, int n_ranks, int64_t *send_count,
                       int64_t *recv_count);
/* Copy of the owned rows' values as they stand in HBM (after pdh_assemble_device / pdh_exchange_apply). */
int pdh_copy_values(pdh_ctx *ctx, double *values);
/* One pass over the values in HBM: out4 = { sum, sum of |.|, max |.|, number of non-finite entries } of the owned rows -
 * a validity signal for matrices too large to copy back (for FE_DGQ the sum is 1^T A 1, known in closed form).        */
int pdh_values_checksum(pdh_ctx *ctx, double *out4);

/* Two algebraically identical forms of the same sums exist (results differ by rounding only, both are tested against
 * the oracle): DIRECT contracts basis values over the quadrature points for all n^2 pairs (f64 MFMA, pdh_kernels.h;
 * pdh_tiled.h for more than 64 dofs per polytope);
 * MOMENT first reduces the quadrature to (2p+1)^3 Legendre moments per polytope / face and obtains the blocks by sum
 * factorisation (pdh_moment.h; 3-D, degree 1..3).  AUTO picks the faster one for the resident problem.          */
#define PDH_ALG_AUTO 0
#define PDH_ALG_DIRECT 1
#define PDH_ALG_MOMENT 2
#define PDH_ALG_MIXED 3 /* reported only: AUTO chose MOMENT for the diagonal blocks and DIRECT for the coupling blocks */
/* ROWS: the owner of a polytope writes ALL blocks of its rows as whole 128-byte lines (owner computes rows).  Exists in 3-D for
 * FE_DGQ / FE_AggloDGP of degree 1 .. 3 when every face of every owned polytope lies in axis-aligned planes (agglomerates of
 * Cartesian cells; a neighbour may be met along several planes - METIS-like "staircase" agglomerates - for every element).  Every
 * element but FE_DGQ(3) additionally needs tensor-product rules on the sub-cells and sub-faces (vq_tensor_n / fq_tensor_n).
 * Several kernels serve it (pdh_rows_kernel_in_use below).  AUTO takes it whenever the resident problem qualifies (tested on the
 * quadrature points at pdh_set_problem - or known by construction after pdh_set_problem_cartesian - no mesh flag needed).   */
#define PDH_ALG_ROWS 4
/* Host-only: 1 if PDH_ALG_ROWS applies to this description / row range, 0 if not (pdh_last_error(NULL) says why). */
int pdh_check_rows(const pdh_problem *problem, int32_t row_begin, int32_t row_end);
/* Host-only: 1 if the term kernels (PDH_ROWS_TERMS) apply to this description / row range, 0 if not (pdh_last_error(NULL) says
 * why).  stats5 (may be NULL): most runs (polytopal faces), sub-faces, interior sub-faces and cells of one owned polytope, and
 * the LDS bytes a workgroup needs for them (the kernels apply while that stays within their budget).                          */
int pdh_check_terms(const pdh_problem *problem, int32_t row_begin, int32_t row_end, int64_t *stats5);
/* Term kernels sum over the cells and sub-faces of a polytope; where those form tensor grids (block agglomerates: the R-tree levels of a
 * structured grid; the sub-faces shared with one neighbour in one plane) a sub-grid of them is ONE cell / sub-face with composite 1-D
 * rules - found on the data at pdh_set_problem.  out4 = { cells, cells after merging, sub-faces, sub-faces after merging } of the
 * resident problem's owned polytopes (zeros if another kernel serves it).                                                          */
int pdh_terms_merge_stats(pdh_ctx *ctx, int64_t *out4);
int pdh_set_algorithm(pdh_ctx *ctx, int algorithm);
/* Which row kernel serves the resident problem (PDH_ALG_ROWS has several; all write whole rows, owner computes rows):
 *   TERMS    every element of degree 1 .. 3 on agglomerates of Cartesian cells with tensor-product rules, while a polytope's 1-D tables
 *            fit the kernels' LDS budget: every entry a short sum of products of three 1-D matrix entries per sub-cell / sub-face, any
 *            number of planes per neighbour; cells / sub-faces that form tensor grids summed over as one with composite rules
 *            (pdh_terms.h, one wave per polytope; FE_DGQ(3): pdh_terms_wg.h, a workgroup per polytope - PDH_TERMS_DGQ3=0 in the
 *            environment keeps pdh_rows.h for that element; reference examples/poisson.cc:413, 543-566 - FE_AggloDGP on METIS
 *            agglomerates - is this case)
 *   PIECES   FE_DGQ(3), one plane per neighbour: moments + Kronecker form, rows in aligned 512-byte pieces (pdh_rows.h) - also
 *            for rules without tensor structure
 *   MULTI    FE_DGQ(3), several planes per neighbour (METIS-like agglomerates of Cartesian cells; pdh_rows.h)
 *   STREAMED FE_DGQ(1,2) / FE_AggloDGP(1..3), moment form, one plane per neighbour (pdh_rows.h): polytopes beyond the LDS budget of
 *            TERMS                                                                                                              */
#define PDH_ROWS_NONE 0
#define PDH_ROWS_PIECES 1
#define PDH_ROWS_MULTI 2
#define PDH_ROWS_STREAMED 3
#define PDH_ROWS_TERMS 4
int pdh_rows_kernel_in_use(pdh_ctx *ctx);
int pdh_algorithm_in_use(pdh_ctx *ctx); /* PDH_ALG_DIRECT, PDH_ALG_MOMENT, PDH_ALG_MIXED or PDH_ALG_ROWS for the resident problem, < 0 on error */

/* On large problems the two kernels of a step (diagonal blocks / coupling blocks: disjoint values, complementary
 * bottlenecks) run concurrently, the second on an internal stream forked from and joined into pdh_stream() - callers
 * still see one ordered stream.  pdh_set_overlap(ctx, 0) serialises them (cleaner per-kernel timings, ~4 % slower).   */
int pdh_set_overlap(pdh_ctx *ctx, int enabled);

/* Measurement helpers (HIP events on the context's stream).  kernel 0 = diagonal-block kernel
 * (volume + own-side face terms), kernel 1 = off-diagonal (interface coupling) kernel.            */
#define PDH_N_KERNELS 2
int pdh_set_profiling(pdh_ctx *ctx, int enabled); /* enabling (re)starts the accumulation */
int pdh_kernel_times_ms(pdh_ctx *ctx, float *ms /* [PDH_N_KERNELS] average per launch */,
                        int *n_launches /* launches averaged over, may be NULL */);
int pdh_problem_stats(pdh_ctx *ctx, int64_t *stats /* [8]: n_owned_agg, n_offdiag_items, n_vq_points,
                        n_face_side_points, n_values, dofs_per_cell, lds_bytes_diag, lds_bytes_offdiag */);

/* Executed work of one pdh_assemble_device launch: v_mfma_f64_4x4x4_4b_f64 instructions issued by k_diag and
 * k_offdiag (512 flop each).  Smaller than the algorithmic count of SURVEY.md 8(d): only the upper tile
 * pairs of the symmetric diagonal blocks are computed and A[Q,P] is written as A[P,Q]^T.             */
int pdh_kernel_work(pdh_ctx *ctx, int64_t *mfma_instr /* [PDH_N_KERNELS] */);

/* Host-only validation of a problem description: runs every check of pdh_set_problem_local without
 * touching a GPU (usable on a build machine).  stats as in pdh_problem_stats, may be NULL.          */
int pdh_check_problem(const pdh_problem *problem, int32_t row_begin, int32_t row_end, int64_t *stats);

/* Version / build info: "polydeal_hip <version> gfx950". */
const char *pdh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* POLYDEAL_HIP_H */
