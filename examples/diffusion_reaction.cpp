// diffusion_reaction.cpp — the flow of the reference's examples/diffusion_reaction.cc (BASELINE.json configs[3]) through the C ABI,
// with the MPI ranks of the reference played one after the other on one GPU:
//   main :840-868            degree 1 .. 4 of FE_DGQ (the HEX branch of the reference; degree 4 = 125 dofs per polytope: 64 x 64 tiles), c = 0.5
//   make_grid :372-396       unit cube, refined globally, cells partitioned over the ranks
//   setup_agglomerated_problem :402-417   METIS inside every rank, n_local_agglomerates each (-> regions grown over the cell graph:
//                            METIS is not available offline); whole agglomerates per rank (source/agglomeration_handler.cc:83-87)
//   assemble_system :420-696 QGauss(p+1), -Laplace u + c u = f, penalty 10 p^2 / h, Nitsche boundary with the exact solution,
//                            owner of a face: the side with the smaller id; TrilinosWrappers::SparseMatrix rows per rank (:448-466)
//   solve :699-735, output_results :739-835   L2 and H1-seminorm errors against u = exp(x y z), summed over the ranks
// Every "rank" describes ONLY its own polytopes and their ghost neighbours (pdh_problem::local = 1, Epetra column order: owned
// columns first, ghosts behind) and assembles ONLY its rows (owner computes rows: no matrix traffic between ranks); the rows are
// then put together for the host solver.  Usage: diffusion_reaction [n_refinements = 3] [n_ranks = 4] [n_local_agglomerates = 10]
#include "../polydeal_amd/csrc/host/polydeal_host.h"
#include "host_solver.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>

using namespace polydeal_hip;

int main(int argc, char **argv)
{
  constexpr int dim = 3;
  const int n_refinements = argc > 1 ? std::atoi(argv[1]) : 3;
  const int n_ranks = argc > 2 ? std::atoi(argv[2]) : 4;
  const int n_local_agglomerates = argc > 3 ? std::atoi(argv[3]) : 10;
  const double reaction_coefficient = 0.5; // :853
  auto u_exact = [](const double *x) { return std::exp(x[0] * x[1] * x[2]); }; // Solution::value :258
  auto grad_exact = [](const double *x, double *g) {                            // Solution::gradient :264
    const double s = std::exp(x[0] * x[1] * x[2]);
    g[0] = x[1] * x[2] * s, g[1] = x[0] * x[2] * s, g[2] = x[0] * x[1] * s;
  };
  auto rhs_f = [&](double x, double y, double z) { // RightHandSide::value_list :216-229
    return -std::exp(x * y * z) * ((x * y) * (x * y) + (x * z) * (x * z) + (y * z) * (y * z) - reaction_coefficient);
  };
  struct Row
  {
    unsigned degree, n_dofs;
    double l2, h1, assemble_s;
  };
  std::vector<Row> table;
  for (const unsigned degree : {1u, 2u, 3u, 4u})
    {
      const BackgroundGrid tria = BackgroundGrid::hyper_cube_refined(dim, 0., 1., n_refinements);
      AgglomerationHandler ah(tria);
      partition_into_grown_agglomerates(ah, n_ranks * n_local_agglomerates, 7);
      const FE_DGQ<dim> fe(degree); // :317
      ah.initialize_fe_values(degree + 1, degree + 1);
      ah.distribute_agglomerated_dofs(fe);
      const unsigned n = ah.n_dofs_per_cell(), N = ah.n_dofs(), nA = ah.n_agglomerates();
      const SipVariant variant = SipVariant::diffusion_reaction(fe); // penalty 10 p^2 (:366), id() < id() (:563), c = 0.5
      // the global matrix the host solver works on (ascending columns: Trilinos rows are sorted by column)
      std::vector<int64_t> rowptr;
      std::vector<int32_t> colind;
      ah.create_agglomeration_sparsity_pattern(rowptr, &colind, false);
      std::vector<double> values((size_t)rowptr.back(), 0.0), rhs(N, 0.0);
      // contiguous ranges of whole polytopes per rank, balanced by the non-zeros a rank writes
      std::vector<int> splits(1, 0);
      {
        std::vector<long long> w(nA, 0), cum(nA + 1, 0);
        for (unsigned P = 0; P < nA; ++P)
          w[ah.dof_offset_of(P) / n] = (rowptr[ah.dof_offset_of(P) + 1] - rowptr[ah.dof_offset_of(P)]) / n;
        for (unsigned a = 0; a < nA; ++a)
          cum[a + 1] = cum[a] + w[a];
        for (int r = 1; r < n_ranks; ++r)
          {
            unsigned a = (unsigned)splits.back() / n + 1;
            while (a < nA - (unsigned)(n_ranks - r) && cum[a] * n_ranks < cum[nA] * r)
              ++a;
            splits.push_back((int)(a * n));
          }
        splits.push_back((int)N);
      }
      double t_assemble = 0.0;
      std::vector<FlatProblem> F((size_t)n_ranks);
      std::vector<pdh_ctx *> ctx((size_t)n_ranks, nullptr);
      for (int r = 0; r < n_ranks; ++r)
        {
          const int r0 = splits[r], r1 = splits[r + 1];
          const auto start = std::chrono::high_resolution_clock::now();
          ah.flatten_local(variant, F[r], r0, r1, false, true, nullptr, &splits, true);
          if (pdh_create(&ctx[r], 0) != PDH_OK || pdh_set_problem_local(ctx[r], &F[r].c, r0, r1) != PDH_OK)
            {
              std::fprintf(stderr, "rank %d: %s\n", r, pdh_last_error(ctx[r]));
              return 1;
            }
          // f at the volume points, the Dirichlet datum (exact solution) at the face points of this rank's description
          const int64_t nq = F[r].vq_ptr.back(), nqf = F[r].fq_ptr.back();
          std::vector<double> f_vol((size_t)nq), g_bdry((size_t)nqf);
          for (int64_t q = 0; q < nq; ++q)
            f_vol[q] = rhs_f(F[r].vq_x[q], F[r].vq_x[nq + q], F[r].vq_x[2 * nq + q]);
          for (int64_t q = 0; q < nqf; ++q)
            {
              const double x[3] = {F[r].fq_x[q], F[r].fq_x[nqf + q], F[r].fq_x[2 * nqf + q]};
              g_bdry[q] = u_exact(x);
            }
          std::vector<double> local_values((size_t)(rowptr[r1] - rowptr[r0]));
          if (pdh_assemble(ctx[r], local_values.data()) != PDH_OK ||
              pdh_assemble_rhs(ctx[r], f_vol.data(), g_bdry.data(), rhs.data() + r0) != PDH_OK)
            {
              std::fprintf(stderr, "rank %d: %s\n", r, pdh_last_error(ctx[r]));
              return 1;
            }
          t_assemble += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - start).count();
          // the rank's rows arrive in Epetra's local column order (owned columns first, ghost columns behind; colind of the local
          // description holds LOCAL column ids): translate them - local polytope a owns local columns col_offset[a] .. + n and global
          // columns dof_offset[a] .. + n - and put every row back into ascending global columns
          std::vector<int32_t> loc2glob((size_t)F[r].c.n_agg * n, -1);
          for (int a = 0; a < F[r].c.n_agg; ++a)
            for (unsigned i = 0; i < n; ++i)
              loc2glob[(size_t)F[r].col_offset[a] + i] = F[r].dof_offset[a] + (int32_t)i;
          for (int row = r0; row < r1; ++row)
            {
              const int64_t lb = F[r].rowptr[row - r0], le = F[r].rowptr[row - r0 + 1];
              for (int64_t k = lb; k < le; ++k)
                {
                  const int32_t col = loc2glob[(size_t)F[r].colind[k]];
                  const int32_t *cb = colind.data() + rowptr[row], *ce = colind.data() + rowptr[row + 1];
                  const int32_t *it = std::lower_bound(cb, ce, col);
                  if (it == ce || *it != col)
                    {
                      std::fprintf(stderr, "rank %d: column %d of row %d is not in the global pattern\n", r, col, row);
                      return 1;
                    }
                  values[(size_t)(rowptr[row] + (it - cb))] = local_values[(size_t)k];
                }
            }
        }
      std::cout << "Degree " << degree << ": " << nA << " agglomerates on " << n_ranks << " ranks, " << N << " dofs" << std::endl;
      std::cout << "Time taken by assemble_system() (all ranks, set-up included): " << t_assemble << " seconds" << std::endl;
      std::vector<double> solution;
      const int its = example_solver::solve_cg(rowptr, colind, values, (int)n, rhs, solution);
      std::fprintf(stderr, "  (%d CG iterations)\n", its);
      // errors: every rank the squares over its own polytopes, summed like Utilities::MPI::sum (:736-745 of poly_utils.h)
      double l2sq = 0.0, h1sq = 0.0;
      for (int r = 0; r < n_ranks; ++r)
        {
          const std::vector<double> own(solution.begin() + splits[r], solution.begin() + splits[r + 1]);
          const auto e = PolyUtilsHIP::compute_global_error(ctx[r], F[r], own, u_exact, grad_exact);
          l2sq += e[0] * e[0], h1sq += e[1] * e[1];
          pdh_destroy(ctx[r]);
        }
      std::cout << "L2 error (exponential solution): " << std::sqrt(l2sq) << std::endl;
      std::cout << "Semi H1 error (exponential solution): " << std::sqrt(h1sq) << std::endl;
      table.push_back({degree, N, std::sqrt(l2sq), std::sqrt(h1sq), t_assemble});
    }
  std::cout << "degree\tn_dofs\tL2 error\tH1 error" << std::endl;
  for (const Row &r : table)
    std::cout << r.degree << "\t" << r.n_dofs << "\t" << r.l2 << "\t" << r.h1 << std::endl;
  // p-convergence on the smooth solution: every degree must gain at least a factor of 3 in both norms
  for (size_t k = 1; k < table.size(); ++k)
    if (!(table[k].l2 < table[k - 1].l2 / 3.0 && table[k].h1 < table[k - 1].h1 / 3.0))
      return 2;
  return 0;
}
