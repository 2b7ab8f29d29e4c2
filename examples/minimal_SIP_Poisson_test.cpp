// minimal_SIP_Poisson_test.cpp — the reference's test/polydeal/minimal_SIP_Poisson.cc (:486-509) written against
// the host mirror and assembled on the GPU: the SIP matrix on 2x2 (2-D) / 2x2x2 (3-D) agglomerates of a uniformly
// refined cube must equal, entry by entry to 1e-13, the SIP matrix assembled with one polytope per coarse cell.
// FE_DGQ(1), penalty 20, h_f = 1, QGauss(3), faces owned by the lower index().  Prints "Ok" per dimension, i.e.
// the content of test/polydeal/minimal_SIP_Poisson.output.
#include "../polydeal_amd/csrc/host/polydeal_host.h"

#include <cstdio>
#include <map>

using namespace polydeal_hip;

static constexpr double entry_tol = 1e-13; // minimal_SIP_Poisson.cc:32

template <int dim>
std::map<std::pair<int, int>, double> run(const bool to_agglomerate)
{
  // make_grid (:111-140): hyper_cube(-1,1); 2-D: refine 2 (agglomerated) / 1; 3-D: refine 1 / 0
  const int refine = dim == 2 ? (to_agglomerate ? 2 : 1) : (to_agglomerate ? 1 : 0);
  const BackgroundGrid tria = BackgroundGrid::hyper_cube_refined(dim, -1., 1., refine);
  AgglomerationHandler ah(tria);
  if (to_agglomerate)
    {
      // setup_agglomeration (:148-216): the four coarse cells {0..3},{4..7},{8..11},{12..15}; 3-D: {0..7}
      const int per = 1 << dim;
      for (int b = 0; b < tria.n_active_cells() / per; ++b)
        {
          std::vector<int> cells;
          for (int c = 0; c < per; ++c)
            cells.push_back(b * per + c);
          ah.define_agglomerate(cells);
        }
    }
  else
    for (int c = 0; c < tria.n_active_cells(); ++c)
      ah.define_agglomerate({c});
  const FE_DGQ<dim> dg_fe(1);
  ah.initialize_fe_values(2 * dg_fe.degree + 1, 2 * dg_fe.degree + 1); // :253-260
  ah.distribute_agglomerated_dofs(dg_fe);
  std::vector<int64_t> rowptr;
  std::vector<int32_t> colind;
  ah.create_agglomeration_sparsity_pattern(rowptr, &colind, true);
  std::vector<double> values;
  PolyUtilsHIP::assemble_dg_matrix(values, dg_fe, ah, SipVariant::minimal_sip_test());
  std::map<std::pair<int, int>, double> A;
  for (unsigned r = 0; r < ah.n_dofs(); ++r)
    for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
      A[{(int)r, colind[k]}] = values[k];
  return A;
}

template <int dim>
bool test()
{
  const auto standard_matrix = run<dim>(false);
  const auto agglo_matrix = run<dim>(true);
  // Comparing entries of the two matrices (:494-506); el(i,j) of an entry outside the pattern is 0
  for (const auto &e : standard_matrix)
    {
      const auto it = agglo_matrix.find(e.first);
      if (std::fabs(e.second - (it == agglo_matrix.end() ? 0. : it->second)) >= entry_tol)
        return false;
    }
  for (const auto &e : agglo_matrix)
    {
      const auto it = standard_matrix.find(e.first);
      if (std::fabs(e.second - (it == standard_matrix.end() ? 0. : it->second)) >= entry_tol)
        return false;
    }
  std::printf("Ok\n");
  return true;
}

int main()
{
  // descriptor sanity of the templated FE classes (dofs per cell: (p+1)^dim and C(p+dim,dim), fe_agglodgp.cc:89-101)
  if (FE_DGQ<3>(3).n_dofs_per_cell() != 64 || FE_DGQ<2>(2).n_dofs_per_cell() != 9 || FE_AggloDGP<3>(3).n_dofs_per_cell() != 20 ||
      FE_AggloDGP<2>(2).n_dofs_per_cell() != 6 || FE_AggloDGP<3>(3).get_name() != "FE_AggloDGP<3>(3)")
    {
      std::printf("FE descriptors are wrong\n");
      return 3;
    }
  try
    {
      if (!test<2>() || !test<3>())
        {
          std::printf("Matrices are not equivalent up to machine precision.\n");
          return 1;
        }
    }
  catch (const std::exception &e)
    {
      std::fprintf(stderr, "%s\n", e.what());
      return 2;
    }
  return 0;
}
