// pdh_basis.h — host-side tables of the two finite elements the path supports.
//
// FE_DGQ<dim>(p)     : tensor-product Lagrange basis on the p+1 Gauss-Lobatto points of [0,1],
//                      lexicographic numbering, x fastest  [deal.II convention, SURVEY.md 8(c) item 4].
// FE_AggloDGP<dim>(p): reference source/fe_agglodgp.cc:27-55 — PolynomialSpace<dim> over
//                      Polynomials::Legendre (L_k(x) = sqrt(2k+1) P_k(2x-1), L2-orthonormal on [0,1]);
//                      index order "for iz: for iy < n1d-iz: for ix < n1d-iy-iz" (x fastest);
//                      C(p+dim,dim) dofs (source/fe_agglodgp.cc:89-101).
//
// The kernels evaluate 1-D basis functions by Horner's rule from monomial coefficients in the CENTRED variable
// t = x - 1/2 (|t| <= 1/2 on the box: the coefficients of degree-7 Lagrange polynomials in x reach 1e4 and cost
// three to four digits in the cancellation; in t the evaluation stays at a few ulp).  The coefficients are
// generated here in long double.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace pdh
{
struct Basis1D
{
  int n1d = 0;
  std::vector<std::vector<long double>> coef; // coef[k][m]: phi_k(x) = sum_m coef[k][m] (x - 1/2)^m
};

inline std::vector<long double> gauss_lobatto_nodes(int p)
{
  std::vector<long double> x(p + 1);
  if (p == 0)
    {
      x[0] = 0.5L;
      return x;
    }
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i <= p; ++i)
    {
      long double t = -std::cos(pi * i / p); // Chebyshev-Gauss-Lobatto start
      if (i != 0 && i != p)
        for (int it = 0; it < 100; ++it)
          {
            // Newton on q(t) = P_p'(t): use (1-t^2) P_p' = p (P_{p-1} - t P_p)
            long double p0 = 1.0L, p1 = t;
            for (int k = 1; k < p; ++k)
              {
                const long double p2 = ((2 * k + 1) * t * p1 - k * p0) / (k + 1);
                p0 = p1;
                p1 = p2;
              }
            // p1 = P_p, p0 = P_{p-1}
            const long double dP = p * (p0 - t * p1) / (1 - t * t);
            const long double d2P = (2 * t * dP - p * (p + 1) * p1) / (1 - t * t);
            const long double dt = dP / d2P;
            t -= dt;
            if (std::fabs((double)dt) < 1e-19)
              break;
          }
      x[i] = 0.5L * (t + 1.0L);
    }
  x[0] = 0.0L;
  x[p] = 1.0L;
  return x;
}

inline Basis1D lagrange_basis(int p)
{
  Basis1D b;
  b.n1d = p + 1;
  const auto nodes = gauss_lobatto_nodes(p);
  b.coef.assign(p + 1, std::vector<long double>(p + 1, 0.0L));
  for (int k = 0; k <= p; ++k)
    {
      std::vector<long double> c(1, 1.0L);
      long double denom = 1.0L;
      for (int j = 0; j <= p; ++j)
        if (j != k)
          {
            std::vector<long double> d(c.size() + 1, 0.0L);
            for (size_t m = 0; m < c.size(); ++m)
              {
                d[m + 1] += c[m];
                d[m] -= (nodes[j] - 0.5L) * c[m];
              }
            c.swap(d);
            denom *= nodes[k] - nodes[j];
          }
      for (int m = 0; m <= p; ++m)
        b.coef[k][m] = c[m] / denom;
    }
  return b;
}

inline Basis1D legendre_basis(int p)
{
  Basis1D b;
  b.n1d = p + 1;
  std::vector<std::vector<long double>> P(p + 1, std::vector<long double>(p + 1, 0.0L));
  P[0][0] = 1.0L;
  if (p >= 1)
    P[1][1] = 2.0L; // P_1(2x-1) = 2t
  for (int k = 1; k < p; ++k)
    for (int m = 0; m <= p; ++m)
      {
        // (k+1) P_{k+1} = (2k+1) (2t) P_k - k P_{k-1}
        long double v = 0.0L;
        if (m > 0)
          v += (long double)(2 * k + 1) * 2.0L * P[k][m - 1];
        v -= (long double)k * P[k - 1][m];
        P[k + 1][m] = v / (k + 1);
      }
  b.coef.assign(p + 1, std::vector<long double>(p + 1, 0.0L));
  for (int k = 0; k <= p; ++k)
    {
      const long double s = std::sqrt((long double)(2 * k + 1));
      for (int m = 0; m <= p; ++m)
        b.coef[k][m] = s * P[k][m];
    }
  return b;
}

// multi-index of every dof, packed k0 | k1<<8 | k2<<16
inline std::vector<uint32_t> multi_indices(int dim, int p, int basis /*0 DGQ, 1 AggloDGP*/)
{
  std::vector<uint32_t> mi;
  const int n1d = p + 1;
  if (basis == 0)
    {
      const int n = (dim == 2) ? n1d * n1d : n1d * n1d * n1d;
      for (int i = 0; i < n; ++i)
        {
          const int i0 = i % n1d, i1 = (i / n1d) % n1d, i2 = i / (n1d * n1d);
          mi.push_back((uint32_t)i0 | ((uint32_t)i1 << 8) | ((uint32_t)i2 << 16));
        }
    }
  else if (dim == 2)
    {
      for (int iy = 0; iy < n1d; ++iy)
        for (int ix = 0; ix < n1d - iy; ++ix)
          mi.push_back((uint32_t)ix | ((uint32_t)iy << 8));
    }
  else
    {
      for (int iz = 0; iz < n1d; ++iz)
        for (int iy = 0; iy < n1d - iz; ++iy)
          for (int ix = 0; ix < n1d - iy - iz; ++ix)
            mi.push_back((uint32_t)ix | ((uint32_t)iy << 8) | ((uint32_t)iz << 16));
    }
  return mi;
}

// n-point Gauss-Legendre rule on [0,1] (Newton on P_n, long double)
inline void gauss_legendre01(int n, std::vector<long double> &x, std::vector<long double> &w)
{
  x.assign(n, 0.0L);
  w.assign(n, 0.0L);
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < n; ++i)
    {
      long double t = -std::cos(pi * (i + 0.75L) / (n + 0.5L)), dp = 1.0L;
      for (int it = 0; it < 100; ++it)
        {
          long double p0 = 1.0L, p1 = t;
          for (int k = 1; k < n; ++k)
            {
              const long double p2 = ((2 * k + 1) * t * p1 - k * p0) / (k + 1);
              p0 = p1;
              p1 = p2;
            }
          dp = n * (t * p1 - p0) / (t * t - 1.0L);
          const long double dt = p1 / dp;
          t -= dt;
          if (std::fabs((double)dt) < 1e-19)
            break;
        }
      x[i] = 0.5L * (t + 1.0L);
      w[i] = 1.0L / ((1.0L - t * t) * dp * dp); // = (2 / ((1-t^2) P_n'^2)) / 2 for the unit interval
    }
}

// Tables of the moment form (pdh_moment.h, layout MT<N1D>): expansions of the 1-D products B_k B_l, B'_k B'_l and
// (B_k B_l)' in the L2-orthonormal Legendre polynomials L_a(x) = sqrt(2a+1) P_a(2x-1), a <= 2p, followed by the
// data of the per-face rule (Gauss points, L_a(x_g) w_g, B_k(x_g), B'_k(x_g)).
inline std::vector<double> moment_tables(int p, int basis)
{
  const Basis1D b = (basis == 0) ? lagrange_basis(p) : legendre_basis(p);
  const int n1d = p + 1, NA = 2 * n1d - 1, NAP = NA + 1, NG = 2 * n1d, TAB = n1d * n1d * NAP;
  auto eval = [&](int k, long double x, long double &val, long double &der) {
    const long double t = x - 0.5L;
    val = b.coef[k][p];
    der = 0.0L;
    for (int m = p - 1; m >= 0; --m)
      {
        der = der * t + val;
        val = val * t + b.coef[k][m];
      }
  };
  auto leg = [&](int a, long double x) {
    const long double t = 2.0L * x - 1.0L;
    long double p0 = 1.0L, p1 = t;
    if (a == 0)
      return 1.0L;
    for (int k = 1; k < a; ++k)
      {
        const long double p2 = ((2 * k + 1) * t * p1 - k * p0) / (k + 1);
        p0 = p1;
        p1 = p2;
      }
    return std::sqrt((long double)(2 * a + 1)) * p1;
  };
  std::vector<double> out((size_t)3 * TAB + NG + NA * NG + 2 * n1d * NG, 0.0);
  std::vector<long double> qx, qw;
  gauss_legendre01(2 * n1d + 2, qx, qw); // exact to degree 4p+7
  for (int k = 0; k < n1d; ++k)
    for (int l = 0; l < n1d; ++l)
      for (int a = 0; a < NA; ++a)
        {
          long double e = 0.0L, d = 0.0L, f = 0.0L;
          for (size_t g = 0; g < qx.size(); ++g)
            {
              long double vk, dk, vl, dl;
              eval(k, qx[g], vk, dk);
              eval(l, qx[g], vl, dl);
              const long double la = leg(a, qx[g]) * qw[g];
              e += vk * vl * la;
              d += dk * dl * la;
              f += (dk * vl + vk * dl) * la;
            }
          out[(size_t)0 * TAB + (k * n1d + l) * NAP + a] = (double)e;
          out[(size_t)1 * TAB + (k * n1d + l) * NAP + a] = (double)d;
          out[(size_t)2 * TAB + (k * n1d + l) * NAP + a] = (double)f;
        }
  std::vector<long double> gx, gw;
  gauss_legendre01(NG, gx, gw);
  const size_t off_gx = (size_t)3 * TAB, off_gl = off_gx + NG, off_bv = off_gl + (size_t)NA * NG, off_bd = off_bv + (size_t)n1d * NG;
  for (int g = 0; g < NG; ++g)
    {
      out[off_gx + g] = (double)gx[g];
      for (int a = 0; a < NA; ++a)
        out[off_gl + (size_t)a * NG + g] = (double)(leg(a, gx[g]) * gw[g]);
      for (int k = 0; k < n1d; ++k)
        {
          long double v, d;
          eval(k, gx[g], v, d);
          out[off_bv + (size_t)k * NG + g] = (double)v;
          out[off_bd + (size_t)k * NG + g] = (double)d;
        }
    }
  return out;
}

inline int n_dofs_per_cell(int dim, int p, int basis)
{
  return (int)multi_indices(dim, p, basis).size();
}
} // namespace pdh
