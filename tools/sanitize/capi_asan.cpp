// ASan smoke of pdh_check_problem (host-side validation/packing of the C ABI) on flattened problems.
#include "../../polydeal_amd/csrc/host/polydeal_host.h"
#include <cstdio>
using namespace polydeal_hip;
extern "C" int pdh_check_problem(const pdh_problem *, int32_t, int32_t, int64_t *);
extern "C" int pdh_check_rows(const pdh_problem *, int32_t, int32_t);
extern "C" int pdh_check_exchange(const pdh_problem *, int32_t, int32_t, int, int64_t *, int64_t *);
int main()
{
  for (int dim = 2; dim <= 3; ++dim)
    for (int basis = 0; basis < 2; ++basis)
      {
        BackgroundGrid g = BackgroundGrid::hyper_cube_refined(dim, 0., 1., dim == 2 ? 4 : 3);
        AgglomerationHandler ah(g);
        define_block_agglomerates(ah, 2);
        FiniteElement fe;
        fe.dim = dim;
        fe.degree = 3;
        fe.basis = basis;
        ah.initialize_fe_values(4, 4);
        ah.distribute_agglomerated_dofs(fe);
        FlatProblem F;
        ah.flatten(SipVariant::poisson_example(fe), F, true, true);
        int64_t st[8];
        const int n = fe.n_dofs_per_cell();
        const int nA = F.c.n_agg;
        int rc = pdh_check_problem(&F.c, 0, F.c.n_rows, st);
        int rc2 = pdh_check_problem(&F.c, (nA / 3) * n, (2 * nA / 3) * n, st);
        std::printf("dim %d basis %d: rc %d %d owned %lld items %lld\n", dim, basis, rc, rc2, (long long)st[0], (long long)st[1]);
        if (rc || rc2)
          return 1;
        // eligibility test of the row kernel (planes, tensor rules, per-slot records) on the whole problem and on a row range,
        // and a rank-local description with the exchange layout
        const int rr = pdh_check_rows(&F.c, 0, F.c.n_rows), rr2 = pdh_check_rows(&F.c, (nA / 3) * n, (2 * nA / 3) * n);
        std::printf("  row kernel applies: %d %d\n", rr, rr2);
        if (rr != (dim == 3 ? 1 : 0) || rr2 != rr)
          return 1;
        if (dim == 3)
          {
            FlatProblem L;
            std::vector<int> splits = {0, (nA / 2) * n, nA * n};
            ah.flatten_local(SipVariant::poisson_example(fe), L, 0, (nA / 2) * n, true, true, nullptr, &splits);
            int64_t sc[2], rcv[2];
            if (pdh_check_problem(&L.c, 0, (nA / 2) * n, st) || pdh_check_rows(&L.c, 0, (nA / 2) * n) != 1 ||
                pdh_check_exchange(&L.c, 0, (nA / 2) * n, 2, sc, rcv))
              return 1;
            std::printf("  local description: owned %lld, exchange send %lld recv %lld doubles\n", (long long)st[0], (long long)sc[1], (long long)rcv[1]);
          }
      }
  return 0;
}
