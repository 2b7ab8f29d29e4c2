// poisson.cpp — the flow of the reference's examples/poisson.cc with the matrix, the right-hand side, the evaluation of the
// solution and the error norms on the GPU through the C ABI:
//   main :1112-1144      p-convergence on the unstructured mesh meshes/t3.msh refined twice, 364 agglomerates, FE_AggloDGP(p)
//   make_grid :494-533, setup_agglomeration :536-655 (METIS -> regions grown over the cell graph: METIS is not available offline)
//   assemble_system :694-988  QGauss(p+1), penalty 10(p+1)(p+dim)/h of the lower-index polytope, Nitsche boundary, f = 2 pi^2 sin sin
//   solve :993-1000      SparseDirectUMFPACK -> conjugate gradients with the inverses of the diagonal blocks as preconditioner (host)
//   output_results :1004-1070  interpolate_to_fine_grid (values at the vertices of the sub-cells) and compute_global_error
//   run :1090-1106       prints "Time taken by assemble_system(): ... seconds"
// `poisson --bench [dim refine degree]` keeps the device-resident timing of BASELINE.json configs[1] (2-D, 4096 polytopes).
// Usage: poisson [path/to/t3.msh]
#include "../polydeal_amd/csrc/host/polydeal_host.h"
#include "host_solver.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace polydeal_hip;

namespace
{
std::string find_mesh(int argc, char **argv)
{
  if (argc > 1)
    return argv[1];
  for (const char *p : {"../tests/golden/t3.msh", "tests/golden/t3.msh", "../../meshes/t3.msh"})
    if (std::ifstream(p).good())
      return p;
  return "t3.msh";
}

int bench(int argc, char **argv); // device-resident timing (below)
} // namespace

int main(int argc, char **argv)
{
  if (argc > 1 && std::strcmp(argv[1], "--bench") == 0)
    return bench(argc - 1, argv + 1);
  constexpr int dim = 2;
  const std::string mesh = find_mesh(argc, argv);
  const double pi = M_PI;
  auto u_exact = [&](const double *x) { return std::sin(pi * x[0]) * std::sin(pi * x[1]); }; // SolutionType::product_sine (:95)
  auto grad_exact = [&](const double *x, double *g) {
    g[0] = pi * std::cos(pi * x[0]) * std::sin(pi * x[1]);
    g[1] = pi * std::sin(pi * x[0]) * std::cos(pi * x[1]);
  };
  struct Row
  {
    unsigned n_dofs;
    double l2, h1;
  };
  std::vector<Row> convergence_info;
  std::cout << "Testing p-convergence" << std::endl;
  for (const unsigned int fe_degree : {1u, 2u, 3u, 4u}) // :1119 (HEX)
    {
      std::cout << "Fe degree: " << fe_degree << std::endl;
      BackgroundGrid tria;
      try
        {
          tria = BackgroundGrid::read_msh(mesh); // GridType::unstructured (:497-512)
          tria.refine_global(2);
        }
      catch (const std::exception &e)
        {
          std::fprintf(stderr, "%s\n", e.what());
          return 1;
        }
      AgglomerationHandler ah(tria);
      partition_into_grown_agglomerates(ah, 364, 364); // PartitionerType::metis, n_subdomains = 364 (:1128-1131)
      const FE_AggloDGP<dim> dg_fe(fe_degree);         // :413
      ah.initialize_fe_values(fe_degree + 1, fe_degree + 1); // :702-709
      ah.distribute_agglomerated_dofs(dg_fe);
      std::vector<int64_t> rowptr;
      std::vector<int32_t> colind;
      ah.create_agglomeration_sparsity_pattern(rowptr, &colind, true);
      const unsigned n = ah.n_dofs_per_cell(), N = ah.n_dofs();

      // ---- assemble_system(): matrix and right-hand side on the GPU
      const auto start = std::chrono::high_resolution_clock::now();
      FlatProblem F;
      ah.flatten(SipVariant::poisson_example(dg_fe), F, true, false);
      pdh_ctx *ctx = nullptr;
      if (pdh_create(&ctx, 0) != PDH_OK || pdh_set_problem(ctx, &F.c) != PDH_OK)
        {
          std::fprintf(stderr, "%s\n", pdh_last_error(ctx));
          return 1;
        }
      std::vector<double> values((size_t)rowptr.back()), rhs(N), f_vol((size_t)F.vq_ptr.back()), g_bdry((size_t)F.fq_ptr.back());
      const int64_t nq = F.vq_ptr.back(), nqf = F.fq_ptr.back();
      for (int64_t q = 0; q < nq; ++q) // RightHandSide::value_list (:143-150): 2 pi^2 sin(pi x) sin(pi y)
        f_vol[q] = 2.0 * pi * pi * std::sin(pi * F.vq_x[q]) * std::sin(pi * F.vq_x[nq + q]);
      for (int64_t q = 0; q < nqf; ++q) // Dirichlet datum = the analytical solution on the boundary (:803-829)
        {
          const double x[2] = {F.fq_x[q], F.fq_x[nqf + q]};
          g_bdry[q] = u_exact(x);
        }
      if (pdh_assemble(ctx, values.data()) != PDH_OK || pdh_assemble_rhs(ctx, f_vol.data(), g_bdry.data(), rhs.data()) != PDH_OK)
        {
          std::fprintf(stderr, "%s\n", pdh_last_error(ctx));
          return 1;
        }
      const auto stop = std::chrono::high_resolution_clock::now();
      std::cout << "Time taken by assemble_system(): "
                << std::chrono::duration_cast<std::chrono::microseconds>(stop - start).count() / 1e6 << " seconds" << std::endl;

      // ---- solve()
      std::vector<double> solution;
      const int its = example_solver::solve_cg(rowptr, colind, values, (int)n, rhs, solution);
      std::fprintf(stderr, "  (%u dofs, %d CG iterations)\n", N, its);

      // ---- output_results(): interpolate_to_fine_grid = u_h at the vertices of every sub-cell (include/poly_utils.h:1196-1233)
      {
        std::vector<int64_t> pt_ptr(1, 0);
        std::vector<double> px, py;
        for (unsigned P = 0; P < ah.n_agglomerates(); ++P)
          {
            for (int cell : ah.get_agglomerate((int)P))
              for (int v = 0; v < 4; ++v)
                {
                  px.push_back(tria.vertex(cell, v)[0]);
                  py.push_back(tria.vertex(cell, v)[1]);
                }
            pt_ptr.push_back((int64_t)px.size());
          }
        const size_t np = px.size();
        std::vector<double> pts(2 * np), uh(np);
        std::copy(px.begin(), px.end(), pts.begin());
        std::copy(py.begin(), py.end(), pts.begin() + np);
        // (pt_ptr is indexed by the polytope numbers of the flattened description = polytope index)
        if (pdh_evaluate(ctx, solution.data(), pt_ptr.data(), pts.data(), uh.data(), nullptr) != PDH_OK)
          {
            std::fprintf(stderr, "%s\n", pdh_last_error(ctx));
            return 1;
          }
        double emax = 0.0;
        for (size_t q = 0; q < np; ++q)
          {
            const double x[2] = {pts[q], pts[np + q]};
            emax = std::max(emax, std::fabs(uh[q] - u_exact(x)));
          }
        std::cout << "interpolate_to_fine_grid: " << np << " nodal values, max nodal error " << emax << std::endl;
      }
      // compute_global_error (include/poly_utils.h:1647-1750): evaluation and weighted sums on the device
      const auto err = PolyUtilsHIP::compute_global_error(ctx, F, solution, u_exact, grad_exact);
      std::cout << "Error (L2): " << err[0] << std::endl;
      std::cout << "Error (H1): " << err[1] << std::endl;
      convergence_info.push_back({N, err[0], err[1]});
      pdh_destroy(ctx);
    }
  std::cout << "n_dofs\tL2 error\tH1 error" << std::endl; // ConvergenceInfo::print
  for (const Row &r : convergence_info)
    std::cout << r.n_dofs << "\t" << r.l2 << "\t" << r.h1 << std::endl;
  std::cout << std::endl;
  // p-convergence: every degree must gain at least a factor of 4 in L2 and 2.5 in H1 on this mesh
  for (size_t k = 1; k < convergence_info.size(); ++k)
    if (!(convergence_info[k].l2 < 0.25 * convergence_info[k - 1].l2 && convergence_info[k].h1 < 0.4 * convergence_info[k - 1].h1))
      return 2;
  return 0;
}

namespace
{
int bench(int argc, char **argv)
{
  const int dim = argc > 1 ? std::atoi(argv[1]) : 2;
  const int refine = argc > 2 ? std::atoi(argv[2]) : (dim == 2 ? 7 : 4);
  const int degree = argc > 3 ? std::atoi(argv[3]) : 2;
  const BackgroundGrid tria = BackgroundGrid::hyper_cube_refined(dim, 0., 1., refine);
  AgglomerationHandler ah(tria);
  define_block_agglomerates(ah, 2);
  FiniteElement dg_fe;
  dg_fe.dim = dim;
  dg_fe.degree = degree;
  dg_fe.basis = PDH_BASIS_AGGLODGP;
  ah.initialize_fe_values(degree + 1, degree + 1);
  ah.distribute_agglomerated_dofs(dg_fe);
  std::printf("N polytopes: %u\nDoFs per cell: %u\nN DoFs: %u\n", ah.n_agglomerates(), ah.n_dofs_per_cell(), ah.n_dofs());
  FlatProblem F;
  ah.flatten(SipVariant::poisson_example(dg_fe), F, true, false);
  pdh_ctx *ctx = nullptr;
  if (pdh_create(&ctx, 0) != PDH_OK || pdh_set_problem(ctx, &F.c) != PDH_OK)
    {
      std::fprintf(stderr, "%s\n", pdh_last_error(ctx));
      return 1;
    }
  std::vector<double> values((size_t)F.rowptr.back());
  pdh_assemble(ctx, values.data()); // warm-up
  const auto t0 = std::chrono::high_resolution_clock::now();
  const int reps = 20;
  for (int r = 0; r < reps; ++r)
    pdh_assemble_device(ctx);
  pdh_synchronize(ctx);
  const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
  std::printf("Time taken by assemble_system(): %.6f s (device-resident, mean of %d)\n", secs, reps);
  std::printf("Assembled DoF/s: %.4g\n", ah.n_dofs() / secs);
  // PolyUtils::compute_global_error with u_h = 0 and u = prod_c sin(2 pi x_c): the norms of u itself
  const double two_pi = 2.0 * M_PI;
  const auto err = PolyUtilsHIP::compute_global_error(
    ctx, F, std::vector<double>(ah.n_dofs(), 0.0),
    [&](const double *x) {
      double u = 1;
      for (int c = 0; c < dim; ++c)
        u *= std::sin(two_pi * x[c]);
      return u;
    },
    [&](const double *x, double *g) {
      for (int d = 0; d < dim; ++d)
        {
          g[d] = two_pi * std::cos(two_pi * x[d]);
          for (int c = 0; c < dim; ++c)
            if (c != d)
              g[d] *= std::sin(two_pi * x[c]);
        }
    });
  std::printf("compute_global_error(u_h=0): L2 %.10f (exact %.10f)  H1-semi %.10f (exact %.10f)\n", err[0], std::pow(0.5, 0.5 * dim), err[1],
              two_pi * std::sqrt((double)dim) * std::pow(0.5, 0.5 * dim));
  pdh_destroy(ctx);
  return 0;
}
} // namespace
