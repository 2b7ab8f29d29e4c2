// pdh_host_capi.cpp — C entry points onto the C++ host mirror (host/polydeal_host.h) so that Python
// (bench.py, tests, the polydeal_amd package) can drive it.  Handle based; errors are returned as
// negative codes with the message available from pdhh_last_error().  Nothing here needs a GPU except
// pdhh_assemble_dg_matrix, which goes through the C ABI of polydeal_hip.h.
#include "host/polydeal_host.h"

#include <cstring>
#include <memory>
#include <string>

using namespace polydeal_hip;

namespace
{
thread_local std::string g_host_err;
struct GridH
{
  BackgroundGrid g;
};
struct HandlerH
{
  explicit HandlerH(const BackgroundGrid &g)
    : ah(g)
  {}
  AgglomerationHandler ah;
};
// A flattened problem owns its arrays independently of the handler it came from: a second flatten() of the same
// handler (another variant / layout / row range) must not invalidate views onto an earlier one.
struct FlatH
{
  FlatProblem flat;
  std::vector<int> local_of; // rank-local descriptions: global polytope index of every local polytope
};
template <class F>
int guarded(F &&f)
{
  try
    {
      return f();
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return -1;
    }
}
} // namespace

extern "C" {
const char *pdhh_last_error(void) { return g_host_err.c_str(); }

// GridGenerator::subdivided_hyper_rectangle: repetitions[dim], p1[dim], p2[dim]
void *pdhh_grid_create_rectangle(int dim, const int *repetitions, const double *lo, const double *hi)
{
  try
    {
      auto *h = new GridH;
      h->g = BackgroundGrid::subdivided_hyper_rectangle(dim, repetitions, lo, hi);
      return h;
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return nullptr;
    }
}

void *pdhh_grid_create(int dim, int n_per_dir, int morton, double lo, double hi)
{
  try
    {
      auto *h = new GridH;
      if (morton)
        {
          int lev = 0;
          while ((1 << lev) < n_per_dir)
            ++lev;
          if ((1 << lev) != n_per_dir)
            throw std::invalid_argument("Morton grids need a power-of-two number of cells per direction");
          h->g = BackgroundGrid::hyper_cube_refined(dim, lo, hi, lev);
        }
      else
        h->g = BackgroundGrid::subdivided_hyper_cube(dim, n_per_dir, lo, hi);
      return h;
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return nullptr;
    }
}
// GridIn::read_msh (gmsh 4.1 ASCII, quadrilaterals) followed by refine_global(n_refine)
void *pdhh_grid_read_msh(const char *path, int n_refine)
{
  try
    {
      auto *h = new GridH;
      h->g = BackgroundGrid::read_msh(path);
      h->g.refine_global(n_refine);
      return h;
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return nullptr;
    }
}
int pdhh_grid_neighbor(void *g, int cell, int f) { return static_cast<GridH *>(g)->g.neighbor(cell, f); }
void pdhh_grid_destroy(void *g) { delete static_cast<GridH *>(g); }
int pdhh_grid_n_cells(void *g) { return static_cast<GridH *>(g)->g.n_active_cells(); }
int pdhh_grid_distort(void *g, double factor, unsigned seed)
{
  return guarded([&] {
    static_cast<GridH *>(g)->g.distort(factor, seed);
    return 0;
  });
}
int pdhh_grid_vertices(void *g, int cell, double *out /*[2^dim][dim]*/)
{
  return guarded([&] {
    const BackgroundGrid &G = static_cast<GridH *>(g)->g;
    std::memcpy(out, G.vertex(cell, 0), sizeof(double) * G.nv() * G.dim);
    return 0;
  });
}

void *pdhh_handler_create(void *g)
{
  return new HandlerH(static_cast<GridH *>(g)->g);
}
void pdhh_handler_destroy(void *h) { delete static_cast<HandlerH *>(h); }
#define AH (static_cast<HandlerH *>(h)->ah)
int pdhh_define_agglomerate(void *h, const int32_t *cells, int n)
{
  return guarded([&] { return AH.define_agglomerate(std::vector<int>(cells, cells + n)); });
}
int pdhh_define_block_agglomerates(void *h, int b)
{
  return guarded([&] {
    define_block_agglomerates(AH, b);
    return 0;
  });
}
int pdhh_define_grown_agglomerates(void *h, int cells_per_polytope, unsigned seed)
{
  return guarded([&] {
    define_grown_agglomerates(AH, cells_per_polytope, seed);
    return 0;
  });
}
int pdhh_partition_into_grown_agglomerates(void *h, int n_subdomains, unsigned seed)
{
  return guarded([&] {
    partition_into_grown_agglomerates(AH, n_subdomains, seed);
    return 0;
  });
}
int pdhh_get_agglomerate(void *h, int P, int32_t *cells, int cap)
{
  return guarded([&] {
    const std::vector<int> c = AH.get_agglomerate(P);
    for (int i = 0; i < (int)c.size() && i < cap; ++i)
      cells[i] = c[i];
    return (int)c.size();
  });
}
int pdhh_initialize_fe_values(void *h, int nq, int nqf)
{
  return guarded([&] {
    AH.initialize_fe_values(nq, nqf);
    return 0;
  });
}
int pdhh_distribute_agglomerated_dofs(void *h, int basis, int degree)
{
  return guarded([&] {
    FiniteElement fe;
    fe.dim = AH.get_triangulation().dim;
    fe.degree = degree;
    fe.basis = basis;
    AH.distribute_agglomerated_dofs(fe);
    return 0;
  });
}
int pdhh_n_agglomerates(void *h) { return (int)AH.n_agglomerates(); }
int pdhh_n_dofs(void *h) { return (int)AH.n_dofs(); }
int pdhh_agglomerate_size(void *h, int P)
{
  return guarded([&] { return (int)AH.get_agglomerate(P).size(); });
}
int pdhh_n_dofs_per_cell(void *h) { return (int)AH.n_dofs_per_cell(); }
int pdhh_master_index(void *h, int P)
{
  return guarded([&] { return AH.master_index(P); });
}
int pdhh_master_slave_value(void *h, int cell)
{
  return guarded([&] { return AH.master_slave_value(cell); });
}
int pdhh_n_faces(void *h, int P)
{
  return guarded([&] { return (int)AH.n_faces(P); });
}
// Blocks per row of every polytope (1 + its neighbours), in DOF order (position = dof_offset / n): the weights of a work-balanced
// partition into contiguous row ranges - rows x row length = non-zeros, which is what a rank writes (reference: the cell graph is
// partitioned by work, include/poly_utils.h:553-704; here the polytopes are, by non-zeros).  out [n_agglomerates].
int pdhh_blocks_per_row(void *h, int32_t *out)
{
  return guarded([&] {
    const int nA = (int)AH.n_agglomerates(), n = (int)AH.n_dofs_per_cell();
    for (int P = 0; P < nA; ++P)
      {
        int nb = 1;
        for (unsigned f = 0; f < AH.n_faces(P); ++f)
          if (!AH.at_boundary(P, f))
            ++nb;
        out[AH.dof_offset_of(P) / n] = nb;
      }
    return 0;
  });
}
int pdhh_at_boundary(void *h, int P, int f)
{
  return guarded([&] { return AH.at_boundary(P, (unsigned)f) ? 1 : 0; });
}
int pdhh_neighbor(void *h, int P, int f)
{
  try
    {
      return AH.neighbor(P, (unsigned)f);
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return -2;
    }
}
int pdhh_neighbor_of_agglomerated_neighbor(void *h, int P, int f)
{
  try
    {
      return AH.neighbor_of_agglomerated_neighbor(P, (unsigned)f);
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return -2;
    }
}
int pdhh_interface(void *h, int P, int Q, int32_t *cells, int32_t *faces, int cap)
{
  return guarded([&] {
    const auto &v = AH.get_interface(P, Q);
    for (int i = 0; i < (int)v.size() && i < cap; ++i)
      {
        cells[i] = v[i].first;
        faces[i] = v[i].second;
      }
    return (int)v.size();
  });
}
int pdhh_bbox(void *h, int P, double *lo_hi /*[2][dim]*/)
{
  return guarded([&] {
    const int dim = AH.get_triangulation().dim;
    for (int c = 0; c < dim; ++c)
      {
        lo_hi[c] = AH.bbox(P)[c];
        lo_hi[dim + c] = AH.bbox(P)[3 + c];
      }
    return 0;
  });
}
double pdhh_diameter(void *h, int P) { return AH.diameter(P); }
int pdhh_dof_offset(void *h, int P)
{
  return guarded([&] { return AH.dof_offset_of(P); });
}
double pdhh_volume_jxw_sum(void *h, int P)
{
  QPoints q;
  AH.agglomerated_quadrature(P, q);
  double s = 0;
  for (double w : q.w)
    s += w;
  return s;
}
double pdhh_face_jxw_sum(void *h, int P, int f)
{
  QPoints q;
  AH.face_quadrature_of(P, (unsigned)f, q);
  double s = 0;
  for (double w : q.w)
    s += w;
  return s;
}
// Sparsity: first call with colind == NULL to get nnz (rowptr must have n_dofs+1 entries).
int64_t pdhh_sparsity(void *h, int diag_first, int64_t *rowptr, int32_t *colind)
{
  try
    {
      std::vector<int64_t> rp;
      std::vector<int32_t> ci;
      AH.create_agglomeration_sparsity_pattern(rp, colind ? &ci : nullptr, diag_first != 0);
      std::memcpy(rowptr, rp.data(), rp.size() * sizeof(int64_t));
      if (colind)
        std::memcpy(colind, ci.data(), ci.size() * sizeof(int32_t));
      return rp.back();
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return -1;
    }
}

// Flatten into a new, independently owned problem description; free it with pdhh_flat_destroy.
void *pdhh_flatten(void *h, double penalty_constant, int owner_rule, int h_rule, int boundary, double reaction_c,
                   int diag_first, int with_colind)
{
  try
    {
      SipVariant v;
      v.penalty_constant = penalty_constant;
      v.owner_rule = owner_rule;
      v.h_rule = h_rule;
      v.boundary = boundary;
      v.reaction_c = reaction_c;
      std::unique_ptr<FlatH> F(new FlatH);
      AH.flatten(v, F->flat, diag_first != 0, with_colind != 0);
      return F.release();
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return nullptr;
    }
}
// Rank-local description of the dof rows [row_begin,row_end) (pdh_problem::local = 1).  row_splits (may be NULL):
// [n_ranks+1] first row of every rank, fills agg_rank; epetra_columns: col_offset = Epetra local column ids.
void *pdhh_flatten_local(void *h, double penalty_constant, int owner_rule, int h_rule, int boundary, double reaction_c,
                         int diag_first, int with_colind, int row_begin, int row_end, const int32_t *row_splits, int n_ranks,
                         int epetra_columns)
{
  try
    {
      SipVariant v;
      v.penalty_constant = penalty_constant;
      v.owner_rule = owner_rule;
      v.h_rule = h_rule;
      v.boundary = boundary;
      v.reaction_c = reaction_c;
      std::unique_ptr<FlatH> F(new FlatH);
      std::vector<int> splits;
      if (row_splits)
        splits.assign(row_splits, row_splits + n_ranks + 1);
      AH.flatten_local(v, F->flat, row_begin, row_end, diag_first != 0, with_colind != 0, &F->local_of,
                       row_splits ? &splits : nullptr, epetra_columns != 0);
      return F.release();
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return nullptr;
    }
}
// The description without the points (agglomerates of Cartesian cells): pdh_problem with NULL point arrays + pdh_cartesian_points
// (pdhh_flat_cartesian), for pdh_set_problem_cartesian.  row_end <= row_begin: all rows (global description).
void *pdhh_flatten_cartesian(void *h, double penalty_constant, int owner_rule, int h_rule, int boundary, double reaction_c,
                             int diag_first, int with_colind, int row_begin, int row_end, const int32_t *row_splits, int n_ranks)
{
  try
    {
      SipVariant v;
      v.penalty_constant = penalty_constant;
      v.owner_rule = owner_rule;
      v.h_rule = h_rule;
      v.boundary = boundary;
      v.reaction_c = reaction_c;
      std::unique_ptr<FlatH> F(new FlatH);
      if (row_end <= row_begin)
        AH.flatten_cartesian(v, F->flat, diag_first != 0, with_colind != 0);
      else
        {
          std::vector<int> splits;
          if (row_splits)
            splits.assign(row_splits, row_splits + n_ranks + 1);
          AH.flatten_local_cartesian(v, F->flat, row_begin, row_end, diag_first != 0, with_colind != 0, &F->local_of,
                                     row_splits ? &splits : nullptr, false);
        }
      return F.release();
    }
  catch (const std::exception &e)
    {
      g_host_err = e.what();
      return nullptr;
    }
}
const pdh_cartesian_points *pdhh_flat_cartesian(void *fh)
{
  FlatH *H = static_cast<FlatH *>(fh);
  return H->flat.cartesian ? &H->flat.cart : nullptr;
}
const pdh_problem *pdhh_flat_problem(void *fh) { return &static_cast<FlatH *>(fh)->flat.c; }
void pdhh_flat_destroy(void *fh) { delete static_cast<FlatH *>(fh); }
int64_t pdhh_flat_sizes(void *fh, int64_t *out /*[4]: Nq_tot, Nqf_tot, nnz, n_faces*/)
{
  FlatH *H = static_cast<FlatH *>(fh);
  out[0] = H->flat.vq_ptr.empty() ? 0 : H->flat.vq_ptr.back();
  out[1] = H->flat.fq_ptr.empty() ? 0 : H->flat.fq_ptr.back();
  out[2] = H->flat.rowptr.empty() ? 0 : H->flat.rowptr.back();
  out[3] = (int64_t)H->flat.face_in.size();
  return 0;
}
// global polytope index of every local polytope of a rank-local description (n_agg entries); 0 for global ones
int pdhh_flat_local_of(void *fh, int32_t *out)
{
  FlatH *H = static_cast<FlatH *>(fh);
  for (size_t i = 0; i < H->local_of.size(); ++i)
    out[i] = H->local_of[i];
  return (int)H->local_of.size();
}

// PolyUtilsHIP::assemble_dg_matrix through C: values must hold nnz doubles.
int pdhh_assemble_dg_matrix(void *h, double penalty_constant, int owner_rule, int h_rule, int boundary,
                            double reaction_c, int diag_first, int device, double *values, int64_t n_values)
{
  return guarded([&] {
    SipVariant v;
    v.penalty_constant = penalty_constant;
    v.owner_rule = owner_rule;
    v.h_rule = h_rule;
    v.boundary = boundary;
    v.reaction_c = reaction_c;
    std::vector<double> vals;
    PolyUtilsHIP::assemble_dg_matrix(vals, AH.get_fe(), AH, v, diag_first != 0, device);
    if ((int64_t)vals.size() != n_values)
      throw std::invalid_argument("values buffer has the wrong length");
    std::memcpy(values, vals.data(), vals.size() * sizeof(double));
    return 0;
  });
}

// Utils::fill_injection_matrix through C: rowptr [n_fine_dofs+1], colind / values [n_fine_dofs * n].
int pdhh_fill_injection_matrix(void *coarse, void *fine, int device, int64_t *rowptr, int32_t *colind, double *values,
                               int64_t n_values)
{
  return guarded([&] {
    std::vector<int64_t> rp;
    std::vector<int32_t> ci;
    std::vector<double> va;
    Utils::fill_injection_matrix(static_cast<HandlerH *>(coarse)->ah, static_cast<HandlerH *>(fine)->ah, rp, ci, va, device);
    if ((int64_t)va.size() != n_values)
      throw std::invalid_argument("values buffer has the wrong length");
    std::memcpy(rowptr, rp.data(), rp.size() * sizeof(int64_t));
    std::memcpy(colind, ci.data(), ci.size() * sizeof(int32_t));
    std::memcpy(values, va.data(), va.size() * sizeof(double));
    return 0;
  });
}
#undef AH
}
