// polydeal_hip_dealii_adapter.h — deal.II-side binding of the C ABI (include/polydeal_hip.h).
//
// STATUS: this header needs deal.II (>= 9.7) and polyDEAL's own headers.  Neither is installed in the
// environment this repository is built and tested in, so the header is NOT compiled here; it is the
// concrete form of the binding sketched in INTEGRATION.md, written against the reference's public API
// (include/agglomeration_handler.h:203-452, include/agglomeration_accessor.h:55-203) only.  Its logic is
// mirrored 1:1 by polydeal_hip::AgglomerationHandler::flatten (polydeal_amd/csrc/host/polydeal_host.h),
// which IS compiled and tested (golden outputs of the reference's tests, oracle parity).
//
// Usage inside polyDEAL (examples/poisson.cc, examples/minimal_SIP.cc keep their structure; the matrix part
// of assemble_system() becomes one call):
//
//   #include <polydeal_hip_dealii_adapter.h>
//   ...
//   PolyUtilsHIP::SipOptions<dim> opt;                  // scalars of the caller's variant, SURVEY.md 8(a)
//   opt.penalty_constant = 10. * (p + 1) * (p + dim);   // examples/poisson.cc:476
//   opt.owner_by_index   = true;                        // polytope->index() < neigh->index()  (:841)
//   opt.face_quadrature  = &face_quad;                  // the rule given to ah->initialize_fe_values (:702-709)
//   PolyUtilsHIP::assemble_dg_matrix(system_matrix, dg_fe, *ah, opt);   // instead of the loops :733-987
//
// or, with the reference's own signature (include/poly_utils.h:2000-2004):
//
//   PolyUtilsHIP::assemble_dg_matrix(system_matrix, fe_dg, ah);
#ifndef POLYDEAL_HIP_DEALII_ADAPTER_H
#define POLYDEAL_HIP_DEALII_ADAPTER_H

#include <deal.II/base/exceptions.h>
#include <deal.II/base/mpi.h>

#include <deal.II/fe/fe_dgq.h>
#include <deal.II/fe/fe_nothing.h>
#include <deal.II/fe/fe_values.h>

#include <deal.II/lac/sparse_matrix.h>
#include <deal.II/lac/trilinos_sparse_matrix.h>

#include <agglomeration_handler.h>
#include <fe_agglodgp.h>
#include <polydeal_hip.h>

#include <algorithm>
#include <map>
#include <type_traits>
#include <vector>

namespace PolyUtilsHIP
{
  using namespace dealii;

  template <int dim>
  struct SipOptions
  {
    double penalty_constant = -1.;  // < 0: 10 (p + dim)(p + 1)            (include/poly_utils.h:2018-2019)
    bool   owner_by_index   = false; // false: id() < id() (poly_utils.h:2089); true: index() < index()
    int    h_rule           = 0;     // 0: C / diameter(owner); 1: C (h_f = 1); 2: C max(1/h_in, 1/h_out)
    bool   zero_boundary    = false; // examples/minimal_SIP.cc:230-248
    double reaction_c       = 0.;    // examples/diffusion_reaction.cc:495-501
    int    device           = 0;
    // The face rule the caller handed to ah.initialize_fe_values(quad, flags, FACE_QUAD, face_flags) (the handler keeps it
    // private, agglomeration_handler.h:869).  With it the adapter gathers the face data with ONE FEFaceValues on FE_Nothing
    // (what reinit_master does internally, source/agglomeration_handler.cc:1146-1165) and never calls ah.reinit(polytope,f)
    // / reinit_interface - each of which allocates an FEImmersedSurfaceValues and evaluates all n shape values and gradients
    // on the host (SURVEY trap T9: the reference's dominant CPU cost; the kernels recompute them on the device anyway).
    // nullptr: fall back to ah.reinit* (correct, slow).
    const Quadrature<dim - 1> *face_quadrature = nullptr;
    // Hint (include/polydeal_hip.h: vq_tensor_n / fq_tensor_n): the cell rule is a tensor rule with this many points per
    // direction (QGauss<dim>(n): n) and the background cells are axis-aligned boxes, ditto the face rule.  The library
    // verifies the claim on the points and weights it is given and ignores it if it does not hold (distorted cells), so a
    // caller may always pass the 1-D size of its QGauss rules.  0: no claim.
    unsigned int cell_rule_points_1d = 0;
    unsigned int face_rule_points_1d = 0;
  };

  namespace internal
  {
    // The handler is walked like assemble_dg_matrix walks it (include/poly_utils.h:2034-2192), but only quadrature data is
    // gathered.  Polytopes are numbered LOCALLY: the locally owned ones in polytope_iterators() order, then the ghost
    // polytopes across the partition boundary (rank-local description of include/polydeal_hip.h: pdh_problem.local = 1).
    template <int dim>
    struct Flattened
    {
      pdh_problem               p{};
      std::vector<double>       bbox, vq_x, vq_w, fq_x, fq_n, fq_w, fq_w_out, sigma;
      std::vector<std::int32_t> dof_offset, face_in, face_out, colind, col_offset, agg_rank;
      std::vector<std::int64_t> vq_ptr{0}, fq_ptr{0}, rowptr;
      std::vector<std::vector<double>> vx = std::vector<std::vector<double>>(dim), fx = vx, fn = vx;
      std::map<CellId, int>     local_of; // polytope id -> local number
      std::int32_t              row_begin = 0, row_end = 0;

      void
      bind()
      {
        for (unsigned int c = 0; c < dim; ++c)
          {
            vq_x.insert(vq_x.end(), vx[c].begin(), vx[c].end());
            fq_x.insert(fq_x.end(), fx[c].begin(), fx[c].end());
            fq_n.insert(fq_n.end(), fn[c].begin(), fn[c].end());
          }
        p.n_agg      = dof_offset.size();
        p.n_faces    = face_in.size();
        p.bbox       = bbox.data();
        p.dof_offset = dof_offset.data();
        p.vq_ptr     = vq_ptr.data();
        p.vq_x       = vq_x.data();
        p.vq_w       = vq_w.data();
        p.face_in    = face_in.data();
        p.face_out   = face_out.data();
        p.fq_ptr     = fq_ptr.data();
        p.fq_x       = fq_x.data();
        p.fq_n       = fq_n.data();
        p.fq_w       = fq_w.data();
        p.fq_w_out   = fq_w_out.data();
        p.face_sigma = sigma.data();
        p.agg_rank   = agg_rank.empty() ? nullptr : agg_rank.data();
      }
    };

    // points / normals / JxW of the sub-faces listed in get_interface()[{id_in, id_out}], in list order
    template <int dim>
    void
    gather_face(const AgglomerationHandler<dim> &ah,
                FEFaceValues<dim>               &fv,
                const CellId                    &id_in,
                const CellId                    &id_out,
                std::vector<Point<dim>>         *pts,
                std::vector<Tensor<1, dim>>     *normals,
                std::vector<double>             &jxw)
    {
      for (const auto &[cell, f] : ah.get_interface().at({id_in, id_out}))
        {
          fv.reinit(cell, f);
          for (unsigned int q = 0; q < fv.n_quadrature_points; ++q)
            {
              if (pts)
                pts->push_back(fv.quadrature_point(q));
              if (normals)
                normals->push_back(fv.normal_vector(q));
              jxw.push_back(fv.JxW(q));
            }
        }
    }

    template <int dim>
    void
    flatten(const FiniteElement<dim> &fe, const AgglomerationHandler<dim> &ah, const SipOptions<dim> &opt, Flattened<dim> &F,
            const bool distributed, const unsigned int my_rank)
    {
      const unsigned int p_deg = fe.degree;
      const double       C     = opt.penalty_constant >= 0 ? opt.penalty_constant : 10. * (p_deg + dim) * (p_deg + 1);
      F.p.dim    = dim;
      F.p.degree = p_deg;
      if (dynamic_cast<const FE_DGQ<dim> *>(&fe))
        F.p.basis = PDH_BASIS_DGQ;
      else if (dynamic_cast<const FE_AggloDGP<dim> *>(&fe))
        F.p.basis = PDH_BASIS_AGGLODGP;
      else
        AssertThrow(false, ExcMessage("polydeal_hip: FE type not supported (FE_DGQ or FE_AggloDGP)."));
      F.p.n_rows     = ah.n_dofs(); // global
      F.p.reaction_c = opt.reaction_c;
      F.p.local      = distributed ? 1 : 0;
      F.p.vq_tensor_n = opt.cell_rule_points_1d;
      F.p.fq_tensor_n = opt.face_rule_points_1d;

      std::vector<types::global_dof_index> dofs(fe.n_dofs_per_cell());
      auto add_polytope = [&](const auto &polytope) {
        const auto it = F.local_of.find(polytope->id());
        if (it != F.local_of.end())
          return it->second;
        const int l = F.dof_offset.size();
        F.local_of[polytope->id()] = l;
        const auto bp = polytope->get_bounding_box().get_boundary_points(); // ghosts: recv_ghosted_bbox (accessor.h:612)
        for (unsigned int c = 0; c < dim; ++c)
          F.bbox.push_back(bp.first[c]);
        for (unsigned int c = 0; c < dim; ++c)
          F.bbox.push_back(bp.second[c]);
        polytope->get_dof_indices(dofs); // ghosts: recv_ghost_dofs (accessor.h:552)
        F.dof_offset.push_back(static_cast<std::int32_t>(dofs[0]));
        F.agg_rank.push_back(distributed ? static_cast<std::int32_t>(polytope->subdomain_id()) : 0);
        return l;
      };

      // pass 1: the locally owned polytopes get the first local numbers, with their volume quadrature
      for (const auto &polytope : ah.polytope_iterators())
        if (polytope->is_locally_owned())
          {
            add_polytope(polytope);
            // concatenated sub-cell rules (source/agglomeration_handler.cc:622-707): unit points on the bounding box + JxW.
            // No FEValues of the polytopal space is built (ah.reinit(polytope) would evaluate n shape functions per point).
            const Quadrature<dim> q =
              ah.agglomerated_quadrature(polytope->get_agglomerate(), polytope.master_cell());
            const BoundingBox<dim> &box = polytope->get_bounding_box();
            for (unsigned int k = 0; k < q.size(); ++k)
              {
                const Point<dim> x = box.unit_to_real(q.point(k));
                for (unsigned int c = 0; c < dim; ++c)
                  F.vx[c].push_back(x[c]);
                F.vq_w.push_back(q.weight(k));
              }
            F.vq_ptr.push_back(static_cast<std::int64_t>(F.vq_w.size()));
          }
      const std::size_t n_owned = F.dof_offset.size();

      std::unique_ptr<FEFaceValues<dim>> fv;
      const FE_Nothing<dim>              fe_nothing;
      if (opt.face_quadrature)
        fv = std::make_unique<FEFaceValues<dim>>(ah.get_mapping(), fe_nothing, *opt.face_quadrature,
                                                 update_quadrature_points | update_JxW_values | update_normal_vectors);

      // pass 2: faces, each described from its OWNER side (normal of side 0, JxW_0, JxW_1, sigma of the owner's rule)
      for (const auto &polytope : ah.polytope_iterators())
        {
          if (!polytope->is_locally_owned())
            continue;
          const int P = F.local_of.at(polytope->id());
          for (unsigned int f = 0; f < polytope->n_faces(); ++f)
            {
              std::vector<Point<dim>>     pts;
              std::vector<Tensor<1, dim>> nrm;
              std::vector<double>         w0, w1;
              int                         in = P, out = -1;
              double                      s  = 0.;
              if (polytope->at_boundary(f))
                {
                  if (opt.zero_boundary)
                    continue;
                  if (fv)
                    gather_face(ah, *fv, polytope->id(), polytope->id(), &pts, &nrm, w0);
                  else
                    {
                      const auto &v = ah.reinit(polytope, f);
                      pts           = v.get_quadrature_points();
                      nrm           = v.get_normal_vectors();
                      w0            = v.get_JxW_values();
                    }
                  w1 = w0;
                  s  = opt.h_rule == 1 ? C : C / std::fabs(polytope->diameter());
                }
              else
                {
                  const auto        &neigh   = polytope->neighbor(f);
                  const bool         p_owns  = opt.owner_by_index ? (polytope->index() < neigh->index()) : (polytope->id() < neigh->id());
                  const bool         n_local = neigh->is_locally_owned();
                  if (!p_owns && n_local)
                    continue; // listed when the neighbour's faces are walked
                  const unsigned int nofn = polytope->neighbor_of_agglomerated_neighbor(f);
                  const int          Q    = add_polytope(neigh); // a ghost gets its local number here
                  // own side: points, own outward normal, own JxW
                  std::vector<double> w_own, w_other;
                  if (fv)
                    gather_face(ah, *fv, polytope->id(), neigh->id(), &pts, &nrm, w_own);
                  else
                    {
                      const auto &v = ah.reinit(polytope, f);
                      pts           = v.get_quadrature_points();
                      nrm           = v.get_normal_vectors();
                      w_own         = v.get_JxW_values();
                    }
                  // other side's JxW in matching order: the mirrored interface list (locally owned neighbour) or what
                  // exchange_interface_values() received (source/agglomeration_handler.cc:531-618)
                  if (n_local && fv)
                    gather_face(ah, *fv, neigh->id(), polytope->id(), nullptr, nullptr, w_other);
                  else if (n_local)
                    w_other = ah.reinit(neigh, nofn).get_JxW_values();
                  else
                    w_other = ah.recv_jxws.at(neigh->subdomain_id()).at({neigh->id(), nofn});
                  AssertThrow(w_other.size() == w_own.size(), ExcMessage("interface lists of the two sides differ"));
                  if (p_owns)
                    {
                      in = P, out = Q, w0 = w_own, w1 = w_other;
                    }
                  else
                    { // the ghost neighbour owns the face: same points, opposite normal, JxW roles swapped
                      in = Q, out = P, w0 = w_other, w1 = w_own;
                      for (auto &n : nrm)
                        n *= -1.;
                    }
                  const auto  &owner = p_owns ? polytope : neigh, &other = p_owns ? neigh : polytope;
                  s = C / std::fabs(owner->diameter());
                  if (opt.h_rule == 1)
                    s = C;
                  else if (opt.h_rule == 2)
                    s = C * std::max(1. / owner->diameter(), 1. / other->diameter());
                }
              for (unsigned int q = 0; q < pts.size(); ++q)
                {
                  for (unsigned int c = 0; c < dim; ++c)
                    {
                      F.fx[c].push_back(pts[q][c]);
                      F.fn[c].push_back(nrm[q][c]);
                    }
                  F.fq_w.push_back(w0[q]);
                  F.fq_w_out.push_back(w1[q]);
                }
              F.fq_ptr.push_back(static_cast<std::int64_t>(F.fq_w.size()));
              F.face_in.push_back(in);
              F.face_out.push_back(out);
              F.sigma.push_back(s);
            }
        }
      // ghosts carry no volume points
      while (F.vq_ptr.size() < F.dof_offset.size() + 1)
        F.vq_ptr.push_back(F.vq_ptr.back());
      // owned rows: [min, max + n) of the owned polytopes' dofs (deal.II numbers the dofs of a subdomain contiguously)
      F.row_begin = F.p.n_rows;
      F.row_end   = 0;
      for (std::size_t l = 0; l < n_owned; ++l)
        {
          F.row_begin = std::min<std::int32_t>(F.row_begin, F.dof_offset[l]);
          F.row_end   = std::max<std::int32_t>(F.row_end, F.dof_offset[l] + fe.n_dofs_per_cell());
        }
      (void)my_rank;
      F.bind();
    }

    inline void
    run(const pdh_problem &p, const std::int32_t row_begin, const std::int32_t row_end, double *values, const int device)
    {
      pdh_ctx *ctx = nullptr;
      AssertThrow(pdh_create(&ctx, device) == PDH_OK, ExcMessage(pdh_last_error(nullptr)));
      const int         rc  = pdh_assemble_sip_local(ctx, &p, row_begin, row_end, values);
      const std::string msg = rc == PDH_OK ? "" : pdh_last_error(ctx);
      pdh_destroy(ctx);
      AssertThrow(rc == PDH_OK, ExcMessage("polydeal_hip: " + msg));
    }
  } // namespace internal


  // Serial deal.II matrix: SparsityPattern stores the diagonal first, then ascending columns.  The matrix
  // must already be reinit()-ed on the pattern of ah.create_agglomeration_sparsity_pattern().
  template <int dim>
  void
  assemble_dg_matrix(SparseMatrix<double>            &system_matrix,
                     const FiniteElement<dim>        &fe_dg,
                     const AgglomerationHandler<dim> &ah,
                     const SipOptions<dim>           &opt = SipOptions<dim>())
  {
    internal::Flattened<dim> F;
    internal::flatten(fe_dg, ah, opt, F, false, 0);
    const SparsityPattern &sp = system_matrix.get_sparsity_pattern();
    F.rowptr.assign(1, 0);
    for (types::global_dof_index r = 0; r < sp.n_rows(); ++r)
      {
        for (unsigned int k = 0; k < sp.row_length(r); ++k)
          F.colind.push_back(static_cast<std::int32_t>(sp.column_number(r, k)));
        F.rowptr.push_back(static_cast<std::int64_t>(F.colind.size()));
      }
    F.p.rowptr     = F.rowptr.data();
    F.p.colind     = F.colind.data(); // every row is verified against the DG block layout by pdh_set_problem
    F.p.diag_first = 1;
    // the value array of a SparseMatrix is contiguous in pattern order; global_entry(0) is its first element
    internal::run(F.p, 0, F.p.n_rows, &system_matrix.global_entry(0), opt.device);
  }


  // Trilinos matrix, any number of ranks (examples/diffusion_reaction.cc:448-466, include/poly_utils.h:2000-2195 with
  // MatrixType = TrilinosWrappers::SparseMatrix).  Every rank hands over ONLY what it holds: its locally owned polytopes,
  // the ghost polytopes across its partition boundary (bounding boxes and global dofs from setup_ghost_polytopes(),
  // JxW of the other side from exchange_interface_values()), and the local CSR of its Epetra_CrsMatrix.  An Epetra row is
  // sorted by LOCAL column id (owned columns first, ghost columns behind): col_offset tells the library that order, so
  // the kernels write straight into the matrix storage.  Owner-computes-rows: no entry of another rank's rows is
  // produced, compress(add) has nothing to ship.
  template <int dim>
  void
  assemble_dg_matrix(TrilinosWrappers::SparseMatrix  &system_matrix,
                     const FiniteElement<dim>        &fe_dg,
                     const AgglomerationHandler<dim> &ah,
                     const SipOptions<dim>           &opt = SipOptions<dim>())
  {
    const MPI_Comm     comm = system_matrix.get_mpi_communicator();
    const unsigned int nr   = Utilities::MPI::n_mpi_processes(comm);
    internal::Flattened<dim> F;
    internal::flatten(fe_dg, ah, opt, F, nr > 1, Utilities::MPI::this_mpi_process(comm));
    Epetra_CrsMatrix &M = const_cast<Epetra_CrsMatrix &>(system_matrix.trilinos_matrix());
    int    *rp = nullptr, *ci = nullptr;
    double *v = nullptr;
    AssertThrow(M.ExtractCrsDataPointers(rp, ci, v) == 0, ExcMessage("matrix storage is not optimised"));
    const int n_my_rows = M.NumMyRows();
    AssertThrow(n_my_rows == F.row_end - F.row_begin, ExcMessage("owned rows of the matrix and of the handler differ"));
    F.rowptr.assign(rp, rp + n_my_rows + 1);
    F.colind.assign(ci, ci + rp[n_my_rows]); // local column ids
    // local column id of the first dof of every polytope of the description
    F.col_offset.resize(F.dof_offset.size());
    for (std::size_t l = 0; l < F.dof_offset.size(); ++l)
      {
        const int lid = M.ColMap().LID(static_cast<TrilinosWrappers::types::int_type>(F.dof_offset[l]));
        AssertThrow(lid >= 0, ExcMessage("a ghost polytope's dofs are missing from the column map"));
        F.col_offset[l] = lid;
      }
    F.p.rowptr     = F.rowptr.data();
    F.p.colind     = F.colind.data();
    F.p.col_offset = F.col_offset.data();
    F.p.diag_first = 0;
    internal::run(F.p, F.row_begin, F.row_end, v, opt.device);
    system_matrix.compress(VectorOperation::add); // include/poly_utils.h:2194 (nothing to exchange: rows are owned)
  }
} // namespace PolyUtilsHIP

#endif // POLYDEAL_HIP_DEALII_ADAPTER_H
