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
//   PolyUtilsHIP::SipOptions opt;                       // scalars of the caller's variant, SURVEY.md 8(a)
//   opt.penalty_constant = 10. * (p + 1) * (p + dim);   // examples/poisson.cc:476
//   opt.owner_by_index   = true;                        // polytope->index() < neigh->index()  (:841)
//   PolyUtilsHIP::assemble_dg_matrix(system_matrix, dg_fe, *ah, opt);   // instead of the loops :733-987
//
// or, with the reference's own signature (include/poly_utils.h:2000-2004):
//
//   PolyUtilsHIP::assemble_dg_matrix(system_matrix, fe_dg, ah);
#ifndef POLYDEAL_HIP_DEALII_ADAPTER_H
#define POLYDEAL_HIP_DEALII_ADAPTER_H

#include <deal.II/base/exceptions.h>

#include <deal.II/fe/fe_dgq.h>

#include <deal.II/lac/sparse_matrix.h>
#include <deal.II/lac/trilinos_sparse_matrix.h>

#include <agglomeration_handler.h>
#include <fe_agglodgp.h>
#include <polydeal_hip.h>

#include <algorithm>
#include <type_traits>
#include <vector>

namespace PolyUtilsHIP
{
  using namespace dealii;

  struct SipOptions
  {
    double penalty_constant = -1.;  // < 0: 10 (p + dim)(p + 1)            (include/poly_utils.h:2018-2019)
    bool   owner_by_index   = false; // false: id() < id() (poly_utils.h:2089); true: index() < index()
    int    h_rule           = 0;     // 0: C / diameter(owner); 1: C (h_f = 1); 2: C max(1/h_in, 1/h_out)
    bool   zero_boundary    = false; // examples/minimal_SIP.cc:230-248
    double reaction_c       = 0.;    // examples/diffusion_reaction.cc:495-501
    int    device           = 0;
  };

  namespace internal
  {
    // Walks the handler like assemble_dg_matrix does (include/poly_utils.h:2034-2192) but only to collect
    // the quadrature data the kernels need; no shape function is evaluated on the host.
    template <int dim>
    struct Flattened
    {
      pdh_problem                               p{};
      std::vector<double>                       bbox, vq_x, vq_w, fq_x, fq_n, fq_w, fq_w_out, sigma;
      std::vector<std::int32_t>                 dof_offset, face_in, face_out, colind;
      std::vector<std::int64_t>                 vq_ptr{0}, fq_ptr{0}, rowptr;
      std::vector<std::vector<double>>          vx = std::vector<std::vector<double>>(dim);
      std::vector<std::vector<double>>          fx = std::vector<std::vector<double>>(dim);
      std::vector<std::vector<double>>          fn = std::vector<std::vector<double>>(dim);

      void
      push_face(const FEValuesBase<dim> &fv0, const FEValuesBase<dim> *fv1)
      {
        const auto &pts     = fv0.get_quadrature_points();
        const auto &normals = fv0.get_normal_vectors(); // normal of side 0 for all four blocks (:1881)
        for (unsigned int q = 0; q < pts.size(); ++q)
          {
            for (unsigned int c = 0; c < dim; ++c)
              {
                fx[c].push_back(pts[q][c]);
                fn[c].push_back(normals[q][c]);
              }
            fq_w.push_back(fv0.JxW(q));
            fq_w_out.push_back(fv1 ? fv1->JxW(q) : fv0.JxW(q)); // M12, M21, M22 use JxW of side 1 (:1906-1922)
          }
        fq_ptr.push_back(static_cast<std::int64_t>(fq_w.size()));
      }
    };

    template <int dim>
    void
    flatten(const FiniteElement<dim> &fe, const AgglomerationHandler<dim> &ah, const SipOptions &opt, Flattened<dim> &F)
    {
      const unsigned int p_deg = fe.degree;
      const double       C =
        opt.penalty_constant >= 0 ? opt.penalty_constant : 10. * (p_deg + dim) * (p_deg + 1);
      F.p.dim    = dim;
      F.p.degree = p_deg;
      if (dynamic_cast<const FE_DGQ<dim> *>(&fe))
        F.p.basis = PDH_BASIS_DGQ;
      else if (dynamic_cast<const FE_AggloDGP<dim> *>(&fe))
        F.p.basis = PDH_BASIS_AGGLODGP;
      else
        AssertThrow(false, ExcMessage("polydeal_hip: FE type not supported (FE_DGQ or FE_AggloDGP)."));
      F.p.n_agg      = ah.n_agglomerates();
      F.p.n_rows     = ah.n_dofs();
      F.p.reaction_c = opt.reaction_c;

      const auto &boxes = ah.get_local_bboxes(); // indexed by polytope->index()
      F.bbox.resize(2 * dim * F.p.n_agg);
      F.dof_offset.resize(F.p.n_agg);
      std::vector<types::global_dof_index> dofs(fe.n_dofs_per_cell());

      // NOTE: arrays are indexed by polytope->index(); polytope_iterators() visits polytopes in index order
      // (master_cells_container order, include/agglomeration_handler.h:1213ff).
      for (const auto &polytope : ah.polytope_iterators())
        {
          const unsigned int P  = polytope->index();
          const auto        &bp = boxes[P].get_boundary_points();
          for (unsigned int c = 0; c < dim; ++c)
            {
              F.bbox[(2 * P) * dim + c]     = bp.first[c];
              F.bbox[(2 * P + 1) * dim + c] = bp.second[c];
            }
          polytope->get_dof_indices(dofs);
          F.dof_offset[P] = static_cast<std::int32_t>(dofs[0]);

          // volume quadrature: real points + JxW of the concatenated sub-cell rules (:622-707)
          const auto &fev = ah.reinit(polytope);
          const auto &pts = fev.get_quadrature_points();
          for (unsigned int q = 0; q < pts.size(); ++q)
            {
              for (unsigned int c = 0; c < dim; ++c)
                F.vx[c].push_back(pts[q][c]);
              F.vq_w.push_back(fev.JxW(q));
            }
          F.vq_ptr.push_back(static_cast<std::int64_t>(F.vq_w.size()));

          for (unsigned int f = 0; f < polytope->n_faces(); ++f)
            {
              if (polytope->at_boundary(f))
                {
                  if (opt.zero_boundary)
                    continue;
                  F.push_face(ah.reinit(polytope, f), nullptr);
                  F.face_in.push_back(P);
                  F.face_out.push_back(-1);
                  F.sigma.push_back(opt.h_rule == 1 ? C : C / std::fabs(polytope->diameter()));
                }
              else
                {
                  const auto &neigh = polytope->neighbor(f);
                  const bool  owns  = opt.owner_by_index ? (polytope->index() < neigh->index()) :
                                                           (polytope->id() < neigh->id());
                  if (!owns)
                    continue;
                  const unsigned int nofn = polytope->neighbor_of_agglomerated_neighbor(f);
                  const auto        &ffs  = ah.reinit_interface(polytope, neigh, f, nofn);
                  F.push_face(ffs.first, &ffs.second);
                  F.face_in.push_back(P);
                  F.face_out.push_back(neigh->index());
                  double s = C / std::fabs(polytope->diameter());
                  if (opt.h_rule == 1)
                    s = C;
                  else if (opt.h_rule == 2)
                    s = C * std::max(1. / polytope->diameter(), 1. / neigh->diameter());
                  F.sigma.push_back(s);
                }
            }
        }
      // structure of arrays with the final strides
      for (unsigned int c = 0; c < dim; ++c)
        {
          F.vq_x.insert(F.vq_x.end(), F.vx[c].begin(), F.vx[c].end());
          F.fq_x.insert(F.fq_x.end(), F.fx[c].begin(), F.fx[c].end());
          F.fq_n.insert(F.fq_n.end(), F.fn[c].begin(), F.fn[c].end());
        }
      F.p.n_faces    = F.face_in.size();
      F.p.bbox       = F.bbox.data();
      F.p.dof_offset = F.dof_offset.data();
      F.p.vq_ptr     = F.vq_ptr.data();
      F.p.vq_x       = F.vq_x.data();
      F.p.vq_w       = F.vq_w.data();
      F.p.face_in    = F.face_in.data();
      F.p.face_out   = F.face_out.data();
      F.p.fq_ptr     = F.fq_ptr.data();
      F.p.fq_x       = F.fq_x.data();
      F.p.fq_n       = F.fq_n.data();
      F.p.fq_w       = F.fq_w.data();
      F.p.fq_w_out   = F.fq_w_out.data();
      F.p.face_sigma = F.sigma.data();
    }

    inline void
    run(const pdh_problem &p, double *values, const int device)
    {
      pdh_ctx *ctx = nullptr;
      AssertThrow(pdh_create(&ctx, device) == PDH_OK, ExcMessage(pdh_last_error(nullptr)));
      const int         rc  = pdh_assemble_sip(ctx, &p, values);
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
                     const SipOptions                &opt = SipOptions())
  {
    internal::Flattened<dim> F;
    internal::flatten(fe_dg, ah, opt, F);
    const SparsityPattern &sp = system_matrix.get_sparsity_pattern();
    F.rowptr.assign(1, 0);
    for (types::global_dof_index r = 0; r < sp.n_rows(); ++r)
      {
        for (unsigned int k = 0; k < sp.row_length(r); ++k)
          F.colind.push_back(static_cast<std::int32_t>(sp.column_number(r, k)));
        F.rowptr.push_back(static_cast<std::int64_t>(F.colind.size()));
      }
    F.p.rowptr     = F.rowptr.data();
    F.p.colind     = F.colind.data(); // verified against the DG block layout by pdh_set_problem
    F.p.diag_first = 1;
    // the value array of a SparseMatrix is contiguous in pattern order; global_entry(0) is its first element
    internal::run(F.p, &system_matrix.global_entry(0), opt.device);
  }


  // Trilinos matrix on ONE rank (local ids == global dof ids): Epetra's local CSR is plain ascending.
  // On several ranks Epetra numbers ghost columns after the owned ones, so the position of a neighbour's
  // block inside a row no longer follows the dof offsets: use pdh_set_problem_local per rank with a colind
  // translated to global dofs, or assemble into a serial pattern ("Distributed").
  template <int dim>
  void
  assemble_dg_matrix(TrilinosWrappers::SparseMatrix  &system_matrix,
                     const FiniteElement<dim>        &fe_dg,
                     const AgglomerationHandler<dim> &ah,
                     const SipOptions                &opt = SipOptions())
  {
    AssertThrow(Utilities::MPI::n_mpi_processes(system_matrix.get_mpi_communicator()) == 1,
                ExcNotImplemented());
    internal::Flattened<dim> F;
    internal::flatten(fe_dg, ah, opt, F);
    Epetra_CrsMatrix &M = const_cast<Epetra_CrsMatrix &>(system_matrix.trilinos_matrix());
    int    *rp = nullptr, *ci = nullptr;
    double *v = nullptr;
    AssertThrow(M.ExtractCrsDataPointers(rp, ci, v) == 0, ExcMessage("matrix storage is not optimised"));
    F.rowptr.assign(rp, rp + M.NumMyRows() + 1);
    F.colind.assign(ci, ci + rp[M.NumMyRows()]);
    F.p.rowptr     = F.rowptr.data();
    F.p.colind     = F.colind.data();
    F.p.diag_first = 0;
    internal::run(F.p, v, opt.device);
    system_matrix.compress(VectorOperation::add); // include/poly_utils.h:2194 (nothing to exchange on one rank)
  }
} // namespace PolyUtilsHIP

#endif // POLYDEAL_HIP_DEALII_ADAPTER_H
