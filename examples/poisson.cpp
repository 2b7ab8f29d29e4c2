// poisson.cpp — the flow of the reference's examples/poisson.cc (setup_agglomeration :536-655,
// assemble_system :694-988, timing print :1099-1106) with the matrix assembled on the GPU.
// BASELINE.json configs[1]: 2-D unit square, 4096 agglomerates, FE_AggloDGP(2) (poisson.cc:413),
// QGauss(p+1) (:702-709), penalty 10(p+1)(p+dim)/h of the lower-index polytope (:476, 841, 900-901),
// Nitsche boundary.  METIS / R-tree agglomeration is replaced by 2x2 blocks of a 128x128 grid.
#include "../polydeal_amd/csrc/host/polydeal_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace polydeal_hip;

int main(int argc, char **argv)
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
  if (pdh_create(&ctx, 0) != PDH_OK)
    {
      std::fprintf(stderr, "%s\n", pdh_last_error(nullptr));
      return 1;
    }
  if (pdh_set_problem(ctx, &F.c) != PDH_OK)
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
  double sum = 0;
  for (double x : values)
    sum += x;
  std::printf("sum of entries (constants are in the kernel of the interior operator; Nitsche rows remain): %.12g\n", sum);
  // PolyUtils::compute_global_error (reference include/poly_utils.h:1647-1750) on the device evaluation path, with
  // u_h = 0 and u = prod_c sin(2 pi x_c): the norms of u itself, (1/2)^(dim/2) and 2 pi sqrt(dim) (1/2)^(dim/2)
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
  std::printf("compute_global_error(u_h=0): L2 %.10f (exact %.10f)  H1-semi %.10f (exact %.10f)\n", err[0],
              std::pow(0.5, 0.5 * dim), err[1], two_pi * std::sqrt((double)dim) * std::pow(0.5, 0.5 * dim));
  pdh_destroy(ctx);
  return 0;
}
