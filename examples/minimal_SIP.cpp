// minimal_SIP.cpp — the flow of the reference's examples/minimal_SIP.cc (make_grid :94-118,
// setup_agglomeration :122-139, assemble_system :143-365) with the matrix assembled on the GPU through
// the C ABI.  BASELINE.json configs[0]: 2-D unit square, 64 agglomerates, FE_DGQ(1), QGauss(3), penalty
// 10*max(1/h_in,1/h_out), faces owned by the lower index(), boundary contributions zeroed (:230-262).
// The reference reads meshes/t3.msh and partitions it with METIS; neither is available offline, so a
// 64x64 Cartesian grid agglomerated into 8x8 blocks of 8x8 cells stands in.
#include "../polydeal_amd/csrc/host/polydeal_host.h"

#include <chrono>
#include <cstdio>

using namespace polydeal_hip;

int main()
{
  constexpr int dim = 2;
  const BackgroundGrid tria = BackgroundGrid::hyper_cube_refined(dim, 0., 1., 6); // 64 x 64 cells
  AgglomerationHandler ah(tria);
  define_block_agglomerates(ah, 8);                                               // 64 polytopes
  const FE_DGQ<dim> dg_fe(1);
  ah.initialize_fe_values(2 * dg_fe.degree + 1, 2 * dg_fe.degree + 1);            // minimal_SIP.cc:151-157
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
  const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();

  // v^T A v for v = nodal interpolant of x (FE_DGQ(1): vertices of the bbox): equals int |grad x|^2 = 1
  // (test/polydeal/poisson_sanity_check_01: boundary terms dropped, jumps vanish for continuous v)
  const unsigned n = ah.n_dofs_per_cell();
  std::vector<double> v(ah.n_dofs());
  for (unsigned P = 0; P < ah.n_agglomerates(); ++P)
    for (unsigned i = 0; i < n; ++i)
      v[ah.dof_offset_of(P) + i] = (i & 1) ? ah.bbox(P)[3] : ah.bbox(P)[0];
  double vAv = 0, fro = 0;
  for (unsigned r = 0; r < ah.n_dofs(); ++r)
    for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
      {
        vAv += v[r] * values[k] * v[colind[k]];
        fro += values[k] * values[k];
      }
  std::printf("N polytopes: %u\nDoFs per cell: %u\nN DoFs: %u\nnnz: %lld\n", ah.n_agglomerates(), n, ah.n_dofs(),
              (long long)rowptr.back());
  std::printf("Test with f(x,y)=x: %.12g\n||A||_F = %.12g\nassembled (incl. setup + transfers) in %.3f s\n", vAv,
              std::sqrt(fro), secs);
  return std::fabs(vAv - 1.0) < 1e-10 ? 0 : 2;
}
