// host_solver.h — the host-side linear solver of the examples (the reference's callers hand the assembled matrix to
// SparseDirectUMFPACK / Trilinos, examples/poisson.cc:993-1000, examples/diffusion_reaction.cc:699-735; solvers are outside the
// hot path rebuilt here): conjugate gradients preconditioned with the inverses of the n x n diagonal blocks (one per polytope).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace example_solver
{
// y = A x for the CSR matrix of the agglomerated pattern
void vmult(const std::vector<int64_t> &rp, const std::vector<int32_t> &ci, const std::vector<double> &va, const std::vector<double> &x,
           std::vector<double> &y)
{
  for (size_t r = 0; r + 1 < rp.size(); ++r)
    {
      double s = 0.0;
      for (int64_t k = rp[r]; k < rp[r + 1]; ++k)
        s += va[k] * x[ci[k]];
      y[r] = s;
    }
}

// conjugate gradients, preconditioned with the inverses of the n x n diagonal blocks (one per polytope)
int solve_cg(const std::vector<int64_t> &rp, const std::vector<int32_t> &ci, const std::vector<double> &va, int n, const std::vector<double> &b,
             std::vector<double> &x)
{
  const size_t N = b.size(), nb = N / n;
  std::vector<double> Dinv(nb * n * n);
  for (size_t B = 0; B < nb; ++B)
    {
      std::vector<double> M((size_t)n * n, 0.0), I((size_t)n * n, 0.0);
      for (int i = 0; i < n; ++i)
        {
          const size_t r = B * n + i;
          for (int64_t k = rp[r]; k < rp[r + 1]; ++k)
            if ((size_t)ci[k] / n == B)
              M[(size_t)i * n + ci[k] % n] = va[k];
          I[(size_t)i * n + i] = 1.0;
        }
      for (int c = 0; c < n; ++c) // Gauss-Jordan with partial pivoting (the blocks are SPD)
        {
          int piv = c;
          for (int r = c + 1; r < n; ++r)
            if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)piv * n + c]))
              piv = r;
          for (int k = 0; k < n; ++k)
            {
              std::swap(M[(size_t)c * n + k], M[(size_t)piv * n + k]);
              std::swap(I[(size_t)c * n + k], I[(size_t)piv * n + k]);
            }
          const double d = 1.0 / M[(size_t)c * n + c];
          for (int k = 0; k < n; ++k)
            M[(size_t)c * n + k] *= d, I[(size_t)c * n + k] *= d;
          for (int r = 0; r < n; ++r)
            if (r != c)
              {
                const double f = M[(size_t)r * n + c];
                for (int k = 0; k < n; ++k)
                  M[(size_t)r * n + k] -= f * M[(size_t)c * n + k], I[(size_t)r * n + k] -= f * I[(size_t)c * n + k];
              }
        }
      std::copy(I.begin(), I.end(), Dinv.begin() + B * n * n);
    }
  auto prec = [&](const std::vector<double> &r, std::vector<double> &z) {
    for (size_t B = 0; B < nb; ++B)
      for (int i = 0; i < n; ++i)
        {
          double s = 0.0;
          for (int j = 0; j < n; ++j)
            s += Dinv[(B * n + i) * n + j] * r[B * n + j];
          z[B * n + i] = s;
        }
  };
  auto dot = [&](const std::vector<double> &a, const std::vector<double> &c) {
    double s = 0.0;
    for (size_t i = 0; i < N; ++i)
      s += a[i] * c[i];
    return s;
  };
  x.assign(N, 0.0);
  std::vector<double> r = b, z(N), p(N), q(N);
  prec(r, z);
  p = z;
  double rz = dot(r, z);
  const double stop = 1e-13 * std::sqrt(dot(b, b));
  int it = 0;
  for (; it < 20000 && std::sqrt(dot(r, r)) > stop; ++it)
    {
      vmult(rp, ci, va, p, q);
      const double alpha = rz / dot(p, q);
      for (size_t i = 0; i < N; ++i)
        x[i] += alpha * p[i], r[i] -= alpha * q[i];
      prec(r, z);
      const double rz1 = dot(r, z);
      for (size_t i = 0; i < N; ++i)
        p[i] = z[i] + (rz1 / rz) * p[i];
      rz = rz1;
    }
  return it;
}
} // namespace example_solver
