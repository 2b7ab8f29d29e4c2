/* sip_ref.c — plain-C restatement of the reference's SIP assembly loops (CPU).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): used by tests as a second, independent checker and by bench.py's
 * cpu_baseline leg as the timed "port" of the reference's CPU path.  Never linked into the product.
 *
 * Parity status: this file restates include/poly_utils.h:2034-2193 (assemble_dg_matrix),
 * :1870-1926 (assemble_local_jumps_and_averages), source/agglomeration_handler.cc:729-767 (reinit),
 * :805-834 (reinit_interface), source/mapping_box.cc:393-439, 465-531 (MappingBox) and the FE bases
 * (source/fe_agglodgp.cc:27-55; FE_DGQ [deal.II]).  It is pinned through the NumPy oracle
 * (oracle/polydeal_oracle.py, itself pinned to the reference's golden outputs): tests require
 * entry-wise agreement of the two to 1e-13.  The reference itself cannot be built here (needs deal.II).
 *
 * It keeps the reference's algorithmic SHAPE so that its timing is a fair stand-in for the reference's
 * own CPU assembly: per-polytope scratch allocation and generic basis evaluation on the bounding box at
 * every call (reinit / reinit_interface allocate fresh FEValues objects: agglomeration_handler.cc:753-766,
 * 1223-1242), quadrature-point-outer scalar i/j loops, four dense blocks per interior face assembled by
 * the owner, and scatter by per-entry column search (AffineConstraints::distribute_local_to_global).
 *
 * Input: the flat arrays of a pdh_problem (include/polydeal_hip.h) — the same bytes the GPU path gets.
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/sip_ref.c -o oracle/_build/libsip_ref.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN1D 8

typedef struct
{
  int32_t dim, degree, basis, n_agg, n_faces, n_rows, diag_first, reserved;
  double reaction_c;
  const double *bbox;
  const int32_t *dof_offset;
  const int64_t *vq_ptr;
  const double *vq_x, *vq_w;
  const int32_t *face_in, *face_out;
  const int64_t *fq_ptr;
  const double *fq_x, *fq_n, *fq_w, *fq_w_out, *face_sigma;
  const int64_t *rowptr;
  const int32_t *colind;
} sipref_problem; /* layout-identical to pdh_problem */

/* ---- 1-D bases -------------------------------------------------------------------------------- */
static void gauss_lobatto(int p, double *x)
{
  if (p == 0)
    {
      x[0] = 0.5;
      return;
    }
  for (int i = 0; i <= p; ++i)
    {
      long double t = -cosl(3.14159265358979323846264338327950288L * i / p);
      if (i != 0 && i != p)
        for (int it = 0; it < 100; ++it)
          {
            long double p0 = 1, p1 = t;
            for (int k = 1; k < p; ++k)
              {
                long double p2 = ((2 * k + 1) * t * p1 - k * p0) / (k + 1);
                p0 = p1;
                p1 = p2;
              }
            long double dP = p * (p0 - t * p1) / (1 - t * t);
            long double d2P = (2 * t * dP - (long double)p * (p + 1) * p1) / (1 - t * t);
            long double dt = dP / d2P;
            t -= dt;
            if (fabsl(dt) < 1e-19L)
              break;
          }
      x[i] = (double)(0.5L * (t + 1));
    }
  x[0] = 0.0;
  x[p] = 1.0;
}

/* Lagrange basis on nodes (product form) */
static void eval_lagrange(int p, const double *nodes, double x, double *val, double *der)
{
  for (int k = 0; k <= p; ++k)
    {
      double denom = 1.0, v = 1.0, d = 0.0;
      for (int j = 0; j <= p; ++j)
        if (j != k)
          {
            denom *= nodes[k] - nodes[j];
            v *= x - nodes[j];
            double t = 1.0;
            for (int l = 0; l <= p; ++l)
              if (l != k && l != j)
                t *= x - nodes[l];
            d += t;
          }
      val[k] = v / denom;
      der[k] = d / denom;
    }
}

/* Polynomials::Legendre: sqrt(2k+1) P_k(2x-1) by the three-term recurrence */
static void eval_legendre(int p, double x, double *val, double *der)
{
  const double t = 2.0 * x - 1.0;
  double P[MAXN1D], D[MAXN1D];
  P[0] = 1.0;
  D[0] = 0.0;
  if (p >= 1)
    {
      P[1] = t;
      D[1] = 1.0;
    }
  for (int k = 1; k < p; ++k)
    {
      P[k + 1] = ((2 * k + 1) * t * P[k] - k * P[k - 1]) / (k + 1);
      D[k + 1] = ((2 * k + 1) * (P[k] + t * D[k]) - k * D[k - 1]) / (k + 1);
    }
  for (int k = 0; k <= p; ++k)
    {
      const double s = sqrt(2.0 * k + 1.0);
      val[k] = s * P[k];
      der[k] = 2.0 * s * D[k];
    }
}

typedef struct
{
  int dim, p, basis, n;
  double nodes[MAXN1D];
  int mi[512][3];
} fe_t;

static int fe_init(fe_t *fe, int dim, int p, int basis)
{
  fe->dim = dim;
  fe->p = p;
  fe->basis = basis;
  const int n1 = p + 1;
  int n = 0;
  if (basis == 0)
    {
      gauss_lobatto(p, fe->nodes);
      const int tot = dim == 2 ? n1 * n1 : n1 * n1 * n1;
      if (tot > 512)
        return -1;
      for (int i = 0; i < tot; ++i)
        {
          fe->mi[i][0] = i % n1;
          fe->mi[i][1] = (i / n1) % n1;
          fe->mi[i][2] = i / (n1 * n1);
        }
      n = tot;
    }
  else
    {
      for (int iz = 0; iz < (dim == 3 ? n1 : 1); ++iz)
        for (int iy = 0; iy < n1 - iz; ++iy)
          for (int ix = 0; ix < n1 - iy - iz; ++ix)
            {
              fe->mi[n][0] = ix;
              fe->mi[n][1] = iy;
              fe->mi[n][2] = iz;
              ++n;
            }
    }
  fe->n = n;
  return 0;
}

/* "FEValues on the bounding box": values[q*n+i], grads[(q*n+i)*dim+c] at real points x (SoA, stride),
 * freshly allocated on every call like the reference's ScratchData / FEImmersedSurfaceValues. */
typedef struct
{
  int nq;
  double *val, *grad;
} fev_t;

static void fev_reinit(fev_t *fv, const fe_t *fe, const double *bbox /*[2][dim]*/, const double *x, int64_t stride,
                       int64_t q0, int nq)
{
  const int dim = fe->dim, n = fe->n, n1 = fe->p + 1;
  fv->nq = nq;
  fv->val = (double *)malloc(sizeof(double) * (size_t)nq * n);
  fv->grad = (double *)malloc(sizeof(double) * (size_t)nq * n * dim);
  double inv_h[3];
  for (int c = 0; c < dim; ++c)
    inv_h[c] = 1.0 / (bbox[dim + c] - bbox[c]); /* inverse_cell_extents, mapping_box.cc:222 */
  for (int q = 0; q < nq; ++q)
    {
      double v1[3][MAXN1D], d1[3][MAXN1D];
      for (int c = 0; c < dim; ++c)
        {
          const double xh = (x[c * stride + q0 + q] - bbox[c]) / (bbox[dim + c] - bbox[c]); /* real_to_unit */
          if (fe->basis == 0)
            eval_lagrange(fe->p, fe->nodes, xh, v1[c], d1[c]);
          else
            eval_legendre(fe->p, xh, v1[c], d1[c]);
        }
      (void)n1;
      for (int i = 0; i < n; ++i)
        {
          double v = 1.0;
          for (int c = 0; c < dim; ++c)
            v *= v1[c][fe->mi[i][c]];
          fv->val[(size_t)q * n + i] = v;
          for (int g = 0; g < dim; ++g)
            {
              double t = 1.0;
              for (int c = 0; c < dim; ++c)
                t *= (c == g) ? d1[c][fe->mi[i][c]] : v1[c][fe->mi[i][c]];
              fv->grad[((size_t)q * n + i) * dim + g] = t * inv_h[g]; /* covariant transform, mapping_box.cc:528-530 */
            }
        }
    }
}
static void fev_free(fev_t *fv)
{
  free(fv->val);
  free(fv->grad);
}

/* distribute_local_to_global: add a dense block by searching every column in the (sorted, possibly
 * diagonal-first) row */
static void scatter(const sipref_problem *P, double *values, const double *M, int n, int row0, int col0, int atomic)
{
  for (int i = 0; i < n; ++i)
    {
      const int r = row0 + i;
      const int64_t b = P->rowptr[r], e = P->rowptr[r + 1];
      for (int j = 0; j < n; ++j)
        {
          const int c = col0 + j;
          int64_t pos;
          if (P->diag_first && c == r)
            pos = b;
          else
            {
              int64_t lo = b + (P->diag_first ? 1 : 0), hi = e;
              if (P->colind)
                {
                  while (lo < hi)
                    {
                      const int64_t mid = (lo + hi) >> 1;
                      if (P->colind[mid] < c)
                        lo = mid + 1;
                      else
                        hi = mid;
                    }
                  pos = lo;
                }
              else
                pos = -1;
            }
          if (pos < 0)
            continue;
          if (atomic)
            {
#pragma omp atomic
              values[pos] += M[(size_t)i * n + j];
            }
          else
            values[pos] += M[(size_t)i * n + j];
        }
    }
}

/* One polytope of the reference loop (include/poly_utils.h:2036-2192) */
static void assemble_polytope(const sipref_problem *P, const fe_t *fe, int a, const int64_t *face_begin, double *values,
                              int atomic)
{
  const int dim = P->dim, n = fe->n;
  const int64_t nq_tot = P->vq_ptr[P->n_agg];
  const int64_t nqf_tot = P->n_faces ? P->fq_ptr[P->n_faces] : 0;
  double *cell = (double *)calloc((size_t)n * n, sizeof(double));
  double *M11 = (double *)malloc(sizeof(double) * 4 * (size_t)n * n);
  double *M12 = M11 + (size_t)n * n, *M21 = M12 + (size_t)n * n, *M22 = M21 + (size_t)n * n;
  const double *bbP = P->bbox + (size_t)a * 2 * dim;

  /* volume: poly_utils.h:2039-2052 */
  {
    fev_t fv;
    const int64_t q0 = P->vq_ptr[a];
    fev_reinit(&fv, fe, bbP, P->vq_x, nq_tot, q0, (int)(P->vq_ptr[a + 1] - q0));
    for (int q = 0; q < fv.nq; ++q)
      {
        const double JxW = P->vq_w[q0 + q];
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j)
            {
              double s = 0.0;
              for (int c = 0; c < dim; ++c)
                s += fv.grad[((size_t)q * n + i) * dim + c] * fv.grad[((size_t)q * n + j) * dim + c];
              cell[(size_t)i * n + j] += s * JxW;
              if (P->reaction_c != 0.0)
                cell[(size_t)i * n + j] += P->reaction_c * fv.val[(size_t)q * n + i] * fv.val[(size_t)q * n + j] * JxW;
            }
      }
    fev_free(&fv);
  }

  /* faces this polytope assembles (it is 'in' = the owner or the boundary polytope) */
  for (int64_t f = face_begin[a]; f < face_begin[a + 1]; ++f)
    {
      const int out = P->face_out[f];
      const int64_t q0 = P->fq_ptr[f];
      const int nq = (int)(P->fq_ptr[f + 1] - q0);
      const double sigma = P->face_sigma[f];
      fev_t f0;
      fev_reinit(&f0, fe, bbP, P->fq_x, nqf_tot, q0, nq);
      if (out < 0)
        { /* boundary: poly_utils.h:2060-2085 */
          for (int q = 0; q < nq; ++q)
            {
              const double JxW = P->fq_w[q0 + q];
              for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                  {
                    double gi = 0.0, gj = 0.0;
                    for (int c = 0; c < dim; ++c)
                      {
                        gi += f0.grad[((size_t)q * n + i) * dim + c] * P->fq_n[c * nqf_tot + q0 + q];
                        gj += f0.grad[((size_t)q * n + j) * dim + c] * P->fq_n[c * nqf_tot + q0 + q];
                      }
                    const double vi = f0.val[(size_t)q * n + i], vj = f0.val[(size_t)q * n + j];
                    cell[(size_t)i * n + j] += (-vi * gj - gi * vj + sigma * vi * vj) * JxW;
                  }
            }
        }
      else
        { /* interior, assembled once by the owner: poly_utils.h:2086-2133 + 1884-1925 */
          fev_t f1;
          fev_reinit(&f1, fe, P->bbox + (size_t)out * 2 * dim, P->fq_x, nqf_tot, q0, nq);
          memset(M11, 0, sizeof(double) * 4 * (size_t)n * n);
          for (int q = 0; q < nq; ++q)
            {
              const double w0 = P->fq_w[q0 + q];
              const double w1 = P->fq_w_out ? P->fq_w_out[q0 + q] : w0;
              for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                  {
                    double g0i = 0, g0j = 0, g1i = 0, g1j = 0;
                    for (int c = 0; c < dim; ++c)
                      {
                        const double nc = P->fq_n[c * nqf_tot + q0 + q]; /* normal of side 0 for all four */
                        g0i += f0.grad[((size_t)q * n + i) * dim + c] * nc;
                        g0j += f0.grad[((size_t)q * n + j) * dim + c] * nc;
                        g1i += f1.grad[((size_t)q * n + i) * dim + c] * nc;
                        g1j += f1.grad[((size_t)q * n + j) * dim + c] * nc;
                      }
                    const double v0i = f0.val[(size_t)q * n + i], v0j = f0.val[(size_t)q * n + j];
                    const double v1i = f1.val[(size_t)q * n + i], v1j = f1.val[(size_t)q * n + j];
                    M11[(size_t)i * n + j] += (-0.5 * g0i * v0j - 0.5 * g0j * v0i + sigma * v0i * v0j) * w0;
                    M12[(size_t)i * n + j] += (0.5 * g0i * v1j - 0.5 * g1j * v0i - sigma * v0i * v1j) * w1;
                    M21[(size_t)i * n + j] += (-0.5 * g1i * v0j + 0.5 * g0j * v1i - sigma * v1i * v0j) * w1;
                    M22[(size_t)i * n + j] += (0.5 * g1i * v1j + 0.5 * g1j * v1i + sigma * v1i * v1j) * w1;
                  }
            }
          const int ra = P->dof_offset[a], rb = P->dof_offset[out];
          scatter(P, values, M11, n, ra, ra, atomic);
          scatter(P, values, M12, n, ra, rb, atomic);
          scatter(P, values, M21, n, rb, ra, atomic);
          scatter(P, values, M22, n, rb, rb, atomic);
          fev_free(&f1);
        }
      fev_free(&f0);
    }
  scatter(P, values, cell, n, P->dof_offset[a], P->dof_offset[a], atomic);
  free(cell);
  free(M11);
}

/* ------------------------------------------------------------------------------------------------------
 * "Honest best-effort" CPU variant: same algorithmic content as the reference
 * (volume block + Nitsche block per polytope, four n x n blocks per owned interior face, no symmetry or
 * transpose shortcuts), but implemented the way a performance-minded CPU author would: basis values and
 * normal derivatives evaluated once per point into contiguous arrays, j-contiguous inner loops that the
 * compiler vectorises (build this file with -O3 -march=native), scratch reused across polytopes, and block
 * scatter by one column search per row.  Reported next to the reference-shaped port so that GPU/CPU
 * ratios are not quoted against a strawman only.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct
{
  double *val, *grad, *gn, *val1, *gn1, *blk; /* scratch */
  size_t cap_q;
} fast_scratch;

static void fast_eval(const fe_t *fe, const double *bbox, const double *x, int64_t stride, int64_t q0, int nq,
                      double *val /*[q][n]*/, double *grad /*[c][q][n]*/)
{
  const int dim = fe->dim, n = fe->n;
  double inv_h[3];
  for (int c = 0; c < dim; ++c)
    inv_h[c] = 1.0 / (bbox[dim + c] - bbox[c]);
  for (int q = 0; q < nq; ++q)
    {
      double v1[3][MAXN1D], d1[3][MAXN1D];
      for (int c = 0; c < dim; ++c)
        {
          const double xh = (x[c * stride + q0 + q] - bbox[c]) / (bbox[dim + c] - bbox[c]);
          if (fe->basis == 0)
            eval_lagrange(fe->p, fe->nodes, xh, v1[c], d1[c]);
          else
            eval_legendre(fe->p, xh, v1[c], d1[c]);
          for (int k = 0; k <= fe->p; ++k)
            d1[c][k] *= inv_h[c];
        }
      for (int i = 0; i < n; ++i)
        {
          const int *m = fe->mi[i];
          if (dim == 2)
            {
              val[(size_t)q * n + i] = v1[0][m[0]] * v1[1][m[1]];
              grad[((size_t)0 * nq + q) * n + i] = d1[0][m[0]] * v1[1][m[1]];
              grad[((size_t)1 * nq + q) * n + i] = v1[0][m[0]] * d1[1][m[1]];
            }
          else
            {
              const double a = v1[0][m[0]], b = v1[1][m[1]], cc = v1[2][m[2]];
              val[(size_t)q * n + i] = a * b * cc;
              grad[((size_t)0 * nq + q) * n + i] = d1[0][m[0]] * b * cc;
              grad[((size_t)1 * nq + q) * n + i] = a * d1[1][m[1]] * cc;
              grad[((size_t)2 * nq + q) * n + i] = a * b * d1[2][m[2]];
            }
        }
    }
}

/* block scatter: one column search per row (columns of a block are contiguous in the DG pattern) */
static void scatter_block(const sipref_problem *P, double *values, const double *M, int n, int row0, int col0, int atomic)
{
  for (int i = 0; i < n; ++i)
    {
      const int r = row0 + i;
      const int64_t b = P->rowptr[r], e = P->rowptr[r + 1];
      int64_t lo = b + (P->diag_first ? 1 : 0), hi = e;
      while (lo < hi)
        {
          const int64_t mid = (lo + hi) >> 1;
          if (P->colind[mid] < col0)
            lo = mid + 1;
          else
            hi = mid;
        }
      /* lo = position of the first column >= col0 among the ascending part */
      for (int j = 0; j < n; ++j)
        {
          const int c = col0 + j;
          int64_t pos;
          if (P->diag_first && c == r)
            pos = b;
          else if (P->diag_first && row0 == col0 && c > r)
            pos = lo + j - 1; /* the diagonal entry was moved to the front of the row */
          else
            pos = lo + j;
          if (atomic)
            {
#pragma omp atomic
              values[pos] += M[(size_t)i * n + j];
            }
          else
            values[pos] += M[(size_t)i * n + j];
        }
    }
}

static void assemble_polytope_fast(const sipref_problem *P, const fe_t *fe, int a, const int64_t *face_begin,
                                   double *values, int atomic, fast_scratch *S)
{
  const int dim = P->dim, n = fe->n;
  const int64_t nq_tot = P->vq_ptr[P->n_agg];
  const int64_t nqf_tot = P->n_faces ? P->fq_ptr[P->n_faces] : 0;
  const double *bbP = P->bbox + (size_t)a * 2 * dim;
  size_t need = (size_t)(P->vq_ptr[a + 1] - P->vq_ptr[a]);
  for (int64_t f = face_begin[a]; f < face_begin[a + 1]; ++f)
    if ((size_t)(P->fq_ptr[f + 1] - P->fq_ptr[f]) > need)
      need = (size_t)(P->fq_ptr[f + 1] - P->fq_ptr[f]);
  if (need > S->cap_q)
    {
      free(S->val);
      free(S->grad);
      free(S->gn);
      free(S->val1);
      free(S->gn1);
      S->cap_q = need;
      S->val = (double *)malloc(sizeof(double) * need * n);
      S->grad = (double *)malloc(sizeof(double) * need * n * 3);
      S->gn = (double *)malloc(sizeof(double) * need * n);
      S->val1 = (double *)malloc(sizeof(double) * need * n);
      S->gn1 = (double *)malloc(sizeof(double) * need * n);
    }
  if (!S->blk)
    S->blk = (double *)malloc(sizeof(double) * 5 * (size_t)n * n);
  double *restrict cell = S->blk, *restrict M11 = cell + (size_t)n * n, *restrict M12 = M11 + (size_t)n * n,
                   *restrict M21 = M12 + (size_t)n * n, *restrict M22 = M21 + (size_t)n * n;
  memset(cell, 0, sizeof(double) * (size_t)n * n);
  {
    const int64_t q0 = P->vq_ptr[a];
    const int nq = (int)(P->vq_ptr[a + 1] - q0);
    fast_eval(fe, bbP, P->vq_x, nq_tot, q0, nq, S->val, S->grad);
    for (int c = 0; c < dim; ++c)
      for (int q = 0; q < nq; ++q)
        {
          const double w = P->vq_w[q0 + q];
          const double *restrict g = S->grad + ((size_t)c * nq + q) * n;
          for (int i = 0; i < n; ++i)
            {
              const double gi = g[i] * w;
              double *restrict row = cell + (size_t)i * n;
              for (int j = 0; j < n; ++j)
                row[j] += gi * g[j];
            }
        }
    if (P->reaction_c != 0.0)
      for (int q = 0; q < nq; ++q)
        {
          const double w = P->reaction_c * P->vq_w[q0 + q];
          const double *restrict v = S->val + (size_t)q * n;
          for (int i = 0; i < n; ++i)
            {
              const double vi = v[i] * w;
              double *restrict row = cell + (size_t)i * n;
              for (int j = 0; j < n; ++j)
                row[j] += vi * v[j];
            }
        }
  }
  for (int64_t f = face_begin[a]; f < face_begin[a + 1]; ++f)
    {
      const int out = P->face_out[f];
      const int64_t q0 = P->fq_ptr[f];
      const int nq = (int)(P->fq_ptr[f + 1] - q0);
      const double sigma = P->face_sigma[f];
      fast_eval(fe, bbP, P->fq_x, nqf_tot, q0, nq, S->val, S->grad);
      for (int q = 0; q < nq; ++q)
        for (int i = 0; i < n; ++i)
          {
            double s = 0.0;
            for (int c = 0; c < dim; ++c)
              s += S->grad[((size_t)c * nq + q) * n + i] * P->fq_n[c * nqf_tot + q0 + q];
            S->gn[(size_t)q * n + i] = s;
          }
      if (out < 0)
        {
          for (int q = 0; q < nq; ++q)
            {
              const double w = P->fq_w[q0 + q];
              const double *restrict v = S->val + (size_t)q * n, *restrict g = S->gn + (size_t)q * n;
              for (int i = 0; i < n; ++i)
                {
                  const double vi = v[i] * w, gi = g[i] * w;
                  double *restrict row = cell + (size_t)i * n;
                  for (int j = 0; j < n; ++j)
                    row[j] += -vi * g[j] - gi * v[j] + sigma * vi * v[j];
                }
            }
        }
      else
        {
          fast_eval(fe, P->bbox + (size_t)out * 2 * dim, P->fq_x, nqf_tot, q0, nq, S->val1, S->grad);
          for (int q = 0; q < nq; ++q)
            for (int i = 0; i < n; ++i)
              {
                double s = 0.0;
                for (int c = 0; c < dim; ++c)
                  s += S->grad[((size_t)c * nq + q) * n + i] * P->fq_n[c * nqf_tot + q0 + q];
                S->gn1[(size_t)q * n + i] = s;
              }
          memset(M11, 0, sizeof(double) * 4 * (size_t)n * n);
          for (int q = 0; q < nq; ++q)
            {
              const double w0 = P->fq_w[q0 + q];
              const double w1 = P->fq_w_out ? P->fq_w_out[q0 + q] : w0;
              const double *restrict v0 = S->val + (size_t)q * n, *restrict g0 = S->gn + (size_t)q * n;
              const double *restrict v1 = S->val1 + (size_t)q * n, *restrict g1 = S->gn1 + (size_t)q * n;
              for (int i = 0; i < n; ++i)
                {
                  const double v0i = v0[i], g0i = g0[i], v1i = v1[i], g1i = g1[i];
                  double *restrict r11 = M11 + (size_t)i * n, *restrict r12 = M12 + (size_t)i * n;
                  double *restrict r21 = M21 + (size_t)i * n, *restrict r22 = M22 + (size_t)i * n;
                  for (int j = 0; j < n; ++j)
                    {
                      r11[j] += (-0.5 * g0i * v0[j] - 0.5 * g0[j] * v0i + sigma * v0i * v0[j]) * w0;
                      r12[j] += (0.5 * g0i * v1[j] - 0.5 * g1[j] * v0i - sigma * v0i * v1[j]) * w1;
                      r21[j] += (-0.5 * g1i * v0[j] + 0.5 * g0[j] * v1i - sigma * v1i * v0[j]) * w1;
                      r22[j] += (0.5 * g1i * v1[j] + 0.5 * g1[j] * v1i + sigma * v1i * v1[j]) * w1;
                    }
                }
            }
          const int ra = P->dof_offset[a], rb = P->dof_offset[out];
          scatter_block(P, values, M11, n, ra, ra, atomic);
          scatter_block(P, values, M12, n, ra, rb, atomic);
          scatter_block(P, values, M21, n, rb, ra, atomic);
          scatter_block(P, values, M22, n, rb, rb, atomic);
        }
    }
  scatter_block(P, values, cell, n, P->dof_offset[a], P->dof_offset[a], atomic);
}

/* Assemble polytopes [a_begin, a_end) (all: 0..n_agg).  values must be zero-initialised by the caller
 * and have nnz entries; colind is required.  nthreads <= 1: serial, in polytope order (the reference's
 * per-rank behaviour); > 1: OpenMP over polytopes with atomic scatter (stand-in for mpirun -np N).
 * Returns 0, or <0 on bad input. */
static int sipref_assemble_mode(const sipref_problem *P, double *values, int a_begin, int a_end, int nthreads, int fast);

int sipref_assemble(const sipref_problem *P, double *values, int a_begin, int a_end, int nthreads)
{
  return sipref_assemble_mode(P, values, a_begin, a_end, nthreads, 0);
}

/* hoisted / vectorisable variant (see "Honest best-effort" above) */
int sipref_assemble_fast(const sipref_problem *P, double *values, int a_begin, int a_end, int nthreads)
{
  return sipref_assemble_mode(P, values, a_begin, a_end, nthreads, 1);
}

static int sipref_assemble_mode(const sipref_problem *P, double *values, int a_begin, int a_end, int nthreads, int fast)
{
  if (!P || !values || !P->colind)
    return -1;
  fe_t fe;
  if (fe_init(&fe, P->dim, P->degree, P->basis) != 0)
    return -2;
  /* faces are listed grouped by their assembling polytope (face_in non-decreasing) */
  int64_t *face_begin = (int64_t *)calloc((size_t)P->n_agg + 1, sizeof(int64_t));
  for (int f = 0; f < P->n_faces; ++f)
    {
      if (f > 0 && P->face_in[f] < P->face_in[f - 1])
        {
          free(face_begin);
          return -3;
        }
      ++face_begin[P->face_in[f] + 1];
    }
  for (int a = 0; a < P->n_agg; ++a)
    face_begin[a + 1] += face_begin[a];
  if (fast)
    {
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 1 ? nthreads : 1)
#endif
      {
        fast_scratch S;
        memset(&S, 0, sizeof(S));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int a = a_begin; a < a_end; ++a)
          assemble_polytope_fast(P, &fe, a, face_begin, values, nthreads > 1, &S);
        free(S.val);
        free(S.grad);
        free(S.gn);
        free(S.val1);
        free(S.gn1);
        free(S.blk);
      }
      free(face_begin);
      return 0;
    }
#ifdef _OPENMP
  if (nthreads > 1)
    {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
      for (int a = a_begin; a < a_end; ++a)
        assemble_polytope(P, &fe, a, face_begin, values, 1);
    }
  else
#endif
    for (int a = a_begin; a < a_end; ++a)
      assemble_polytope(P, &fe, a, face_begin, values, 0);
  free(face_begin);
  return 0;
}

int sipref_dofs_per_cell(int dim, int degree, int basis)
{
  fe_t fe;
  if (fe_init(&fe, dim, degree, basis) != 0)
    return -1;
  return fe.n;
}

int sipref_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
