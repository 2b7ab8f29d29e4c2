// ASan/UBSan smoke of the pure-host code paths (no GPU): grid, handler, connectivity, sparsity, flatten.
#include "../../polydeal_amd/csrc/host/polydeal_host.h"
#include <cstdio>
using namespace polydeal_hip;
int main()
{
  for (int dim = 2; dim <= 3; ++dim)
    {
      BackgroundGrid g = BackgroundGrid::hyper_cube_refined(dim, -1., 1., dim == 2 ? 3 : 2);
      g.distort(0.2, 3);
      AgglomerationHandler ah(g);
      // irregular: first a few hand-made agglomerates then singletons
      std::vector<std::vector<int>> groups = {{3, 6, 9}, {15, 36, 37}, {25, 19, 22}};
      std::vector<char> used(g.n_active_cells(), 0);
      for (auto &gr : groups)
        {
          std::vector<int> c;
          for (int x : gr)
            if (x < g.n_active_cells())
              {
                c.push_back(x);
                used[x] = 1;
              }
          std::sort(c.begin(), c.end());
          if (!c.empty())
            ah.define_agglomerate(c);
        }
      for (int c = 0; c < g.n_active_cells(); ++c)
        if (!used[c])
          ah.define_agglomerate({c});
      FiniteElement fe;
      fe.dim = dim;
      fe.degree = 2;
      fe.basis = PDH_BASIS_AGGLODGP;
      ah.initialize_fe_values(3, 3);
      ah.distribute_agglomerated_dofs(fe);
      FlatProblem F;
      ah.flatten(SipVariant::diffusion_reaction(fe), F, true, true);
      int64_t st[8];
      std::printf("dim %d: n_agg %d n_faces %d nnz %lld\n", dim, F.c.n_agg, F.c.n_faces, (long long)F.rowptr.back());
    }
  return 0;
}
