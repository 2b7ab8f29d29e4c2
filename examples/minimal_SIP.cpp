// minimal_SIP.cpp — the flow of the reference's examples/minimal_SIP.cc (make_grid :94-118, setup_agglomeration :122-139,
// assemble_system :143-365, perform_sanity_check :369-399, main :411-421) with the matrix assembled on the GPU through the C ABI.
// BASELINE.json configs[0]: 2-D unit square, the unstructured mesh meshes/t3.msh (91 quadrilaterals) refined 3 times, N
// agglomerates, FE_DGQ(1), QGauss(3), penalty 10*max(1/h_in,1/h_out), faces owned by the lower index(), boundary contributions
// zeroed (:230-262).  The reference partitions the cell graph with METIS; METIS is not available offline, so connected regions
// grown over the same graph stand in (partition_into_grown_agglomerates) - the identities the example prints,
//   v^T A v = 1, 2, ~1e-14   for v = x, x + y, 1,
// hold for ANY agglomeration, and the program prints them in the format of the reference's test of this flow
// (test/polydeal/poisson_sanity_check_03.output).  Usage: minimal_SIP [path/to/t3.msh] (default: tests/golden/t3.msh of this
// repository - the reference's data file).
#include "../polydeal_amd/csrc/host/polydeal_host.h"

#include <chrono>
#include <cstdio>
#include <iostream>

using namespace polydeal_hip;

static std::string find_mesh(int argc, char **argv)
{
  if (argc > 1)
    return argv[1];
  for (const char *p : {"../tests/golden/t3.msh", "tests/golden/t3.msh", "../../meshes/t3.msh"})
    if (std::ifstream(p).good())
      return p;
  return "t3.msh";
}

int main(int argc, char **argv)
{
  constexpr int dim = 2;
  const std::string mesh = find_mesh(argc, argv);
  int bad = 0;
  double secs = 0.0;
  for (const unsigned int n_subdomains : {50u, 100u, 120u, 300u, 400u, 800u}) // minimal_SIP.cc:414
    {
      BackgroundGrid tria;
      try
        {
          tria = BackgroundGrid::read_msh(mesh); // unstructured square [0,1]^2
          tria.refine_global(3);
        }
      catch (const std::exception &e)
        {
          std::fprintf(stderr, "%s\n", e.what());
          return 1;
        }
      AgglomerationHandler ah(tria);
      partition_into_grown_agglomerates(ah, (int)n_subdomains, /*seed*/ n_subdomains);
      std::cout << "N subdomains: " << ah.n_agglomerates() << std::endl;
      const FE_DGQ<dim> dg_fe(1);
      ah.initialize_fe_values(2 * dg_fe.degree + 1, 2 * dg_fe.degree + 1); // :151-157
      ah.distribute_agglomerated_dofs(dg_fe);
      std::vector<int64_t> rowptr;
      std::vector<int32_t> colind;
      ah.create_agglomeration_sparsity_pattern(rowptr, &colind, /*deal.II SparsityPattern layout*/ true);
      std::vector<double> values;
      const auto t0 = std::chrono::high_resolution_clock::now();
      try
        {
          PolyUtilsHIP::assemble_dg_matrix(values, dg_fe, ah, SipVariant::minimal_sip_example());
        }
      catch (const std::exception &e)
        {
          std::fprintf(stderr, "assembly failed: %s\n", e.what());
          return 1;
        }
      secs += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
      // VectorTools::interpolate on the box mapping: FE_DGQ(1) nodes are the corners of the bounding box (x fastest)
      const unsigned n = ah.n_dofs_per_cell();
      std::vector<double> vx(ah.n_dofs()), vxy(ah.n_dofs()), one(ah.n_dofs(), 1.0);
      for (unsigned P = 0; P < ah.n_agglomerates(); ++P)
        for (unsigned i = 0; i < n; ++i)
          {
            const double x = (i & 1) ? ah.bbox(P)[3] : ah.bbox(P)[0], y = (i & 2) ? ah.bbox(P)[4] : ah.bbox(P)[1];
            vx[ah.dof_offset_of(P) + i] = x;
            vxy[ah.dof_offset_of(P) + i] = x + y;
          }
      auto form = [&](const std::vector<double> &v) { // SparseMatrix::matrix_scalar_product
        double s = 0.0;
        for (unsigned r = 0; r < ah.n_dofs(); ++r)
          for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
            s += v[r] * values[k] * v[colind[k]];
        return s;
      };
      const double valuex = form(vx), valuexplusy = form(vxy), value_one = form(one);
      std::cout << "Test with f(x,y)=x:" << valuex << std::endl;
      std::cout << "Test with f(x,y)=x+y:" << valuexplusy << std::endl;
      std::cout << "Test with 1: " << value_one << std::endl;
      if (std::fabs(valuex - 1.0) > 1e-10 || std::fabs(valuexplusy - 2.0) > 1e-10 || std::fabs(value_one) > 1e-10)
        ++bad;
    }
  std::fprintf(stderr, "six assemblies (incl. set-up and transfers) in %.3f s\n", secs);
  return bad ? 2 : 0;
}
