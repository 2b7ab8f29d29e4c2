// polydeal_host.h — C++17 host-side mirror of the reference's operator surface for the SIP path.
//
// deal.II is not available where this builds, so this header provides just enough of the objects the
// reference's callers touch (examples/minimal_SIP.cc, examples/poisson.cc, include/poly_utils.h:2000-2195)
// for the path to be driven exactly like the reference drives it:
//
//   BackgroundGrid          ~ Triangulation<dim> built by GridGenerator::hyper_cube + refine_global /
//                             subdivided_hyper_cube (quad/hex cells, deal.II numbering conventions)
//   FiniteElement           ~ FE_DGQ<dim>(p) / FE_AggloDGP<dim>(p)        (include/fe_agglodgp.h:310-471)
//   AgglomerationHandler    ~ include/agglomeration_handler.h:203-452: define_agglomerate,
//                             distribute_agglomerated_dofs, initialize_fe_values,
//                             create_agglomeration_sparsity_pattern, polytope accessors
//                             (include/agglomeration_accessor.h:55-203)
//   PolyUtilsHIP::assemble_dg_matrix(values, fe, ah, variant) ~ PolyUtils::assemble_dg_matrix
//                             (include/poly_utils.h:2000-2195), calling the HIP kernels through the C ABI.
//
// The algorithms that matter for parity are restated from the reference and cite it:
//   * master/slave bookkeeping and bounding boxes      source/agglomeration_handler.cc:44-104, 476-491
//   * face enumeration with the global visited set     source/agglomeration_handler.cc:1253-1645
//   * DoF numbering (masters in ascending cell order)  source/agglomeration_handler.cc:326-379, 711-725
//   * DG sparsity with flux couplings                  source/agglomeration_handler.cc:910-1022
//   * concatenated sub-cell quadratures                source/agglomeration_handler.cc:622-707, 1146-1165
// This file is independent of oracle/ (which re-derives the same things in NumPy to cross-check it).
#pragma once

#include "../../../include/polydeal_hip.h"
#include "../pdh_basis.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <map>
#include <mutex>
#include <thread>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace polydeal_hip
{
constexpr int invalid_index = -1;

// ---------------------------------------------------------------------------------------------------
// 1-D Gauss-Legendre rule on [0,1]  (QGauss<1>(n))  [deal.II]
// ---------------------------------------------------------------------------------------------------
inline void qgauss_1d_compute(int n, std::vector<double> &x, std::vector<double> &w)
{
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < n; ++i)
    {
      long double t = -std::cos(pi * (i + 0.75L) / (n + 0.5L));
      long double dp = 1.0L;
      for (int it = 0; it < 100; ++it)
        {
          long double p0 = 1.0L, p1 = t;
          for (int k = 1; k < n; ++k)
            {
              const long double p2 = ((2 * k + 1) * t * p1 - k * p0) / (k + 1);
              p0 = p1;
              p1 = p2;
            }
          dp = n * (p0 - t * p1) / (1 - t * t);
          const long double dt = p1 / dp;
          t -= dt;
          if (std::fabs((double)dt) < 1e-19)
            break;
        }
      // recompute derivative at the converged root
      {
        long double p0 = 1.0L, p1 = t;
        for (int k = 1; k < n; ++k)
          {
            const long double p2 = ((2 * k + 1) * t * p1 - k * p0) / (k + 1);
            p0 = p1;
            p1 = p2;
          }
        dp = n * (p0 - t * p1) / (1 - t * t);
      }
      x[i] = (double)(0.5L * (t + 1.0L));
      w[i] = (double)(1.0L / ((1 - t * t) * dp * dp)); // = 2/((1-t^2)P'^2) / 2
    }
}

// cached per thread: the rule is requested once per sub-cell by the quadrature helpers below
inline void qgauss_1d(int n, std::vector<double> &x, std::vector<double> &w)
{
  thread_local std::map<int, std::pair<std::vector<double>, std::vector<double>>> cache;
  auto it = cache.find(n);
  if (it == cache.end())
    {
      std::vector<double> cx, cw;
      qgauss_1d_compute(n, cx, cw);
      it = cache.emplace(n, std::make_pair(cx, cw)).first;
    }
  x = it->second.first;
  w = it->second.second;
}

// ---------------------------------------------------------------------------------------------------
// Finite elements (descriptors; the basis itself lives in pdh_basis.h / the kernels)
// ---------------------------------------------------------------------------------------------------
struct FiniteElement
{
  int dim = 2, degree = 1, basis = PDH_BASIS_DGQ;
  int n_dofs_per_cell() const { return pdh::n_dofs_per_cell(dim, degree, basis); }
  std::string get_name() const
  {
    return std::string(basis == PDH_BASIS_DGQ ? "FE_DGQ<" : "FE_AggloDGP<") + std::to_string(dim) + ">(" +
           std::to_string(degree) + ")";
  }
};
// (the template parameter must not be called `dim`: inside the class that name finds the base-class member)
template <int spacedim>
struct FE_DGQ : FiniteElement
{
  explicit FE_DGQ(unsigned int p)
  {
    dim = spacedim;
    degree = (int)p;
    basis = PDH_BASIS_DGQ;
  }
};
template <int spacedim>
struct FE_AggloDGP : FiniteElement // reference include/fe_agglodgp.h:317
{
  explicit FE_AggloDGP(unsigned int p)
  {
    dim = spacedim;
    degree = (int)p;
    basis = PDH_BASIS_AGGLODGP;
  }
};

// ---------------------------------------------------------------------------------------------------
// Background grid with deal.II conventions [deal.II]: faces 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z,
// vertex v at ((v&1),(v>>1)&1,(v>>2)&1); hyper_cube + refine_global(k) numbers active cells in Morton
// order (child c at offset (c&1,(c>>1)&1,(c>>2)&1)); subdivided_hyper_cube numbers lexicographically.
// ---------------------------------------------------------------------------------------------------
class BackgroundGrid
{
public:
  int dim = 2, n_per_dir = 1; // n_per_dir: cells along x (= every direction unless built as a rectangle)
  std::array<int, 3> n_dir = {1, 1, 1};
  std::vector<std::array<int, 3>> cell_ijk;
  std::vector<int> lut;             // ijk -> cell
  std::vector<double> vertices;     // [cell][2^dim][dim]
  // Unstructured meshes (read_msh): explicit adjacency instead of the ijk lattice.  vertex_id numbers the vertices globally
  // (what makes two cells agree on the direction in which they walk a shared face), nbr_tab / nofn_tab are
  // cell->neighbor(f) and cell->neighbor_of_neighbor(f) of deal.II.
  bool unstructured = false;
  std::vector<int> vertex_id; // [cell][2^dim]
  std::vector<int> nbr_tab;   // [cell][2 dim], invalid_index on the boundary
  std::vector<int> nofn_tab;  // [cell][2 dim]

  static BackgroundGrid hyper_cube_refined(int dim, double lo, double hi, int n_refine)
  {
    return build(dim, 1 << n_refine, lo, hi, true);
  }
  static BackgroundGrid subdivided_hyper_cube(int dim, int n, double lo, double hi)
  {
    return build(dim, n, lo, hi, false);
  }
  // GridGenerator::subdivided_hyper_rectangle(tria, repetitions, p1, p2) [deal.II]: lexicographic cells
  static BackgroundGrid subdivided_hyper_rectangle(int dim, const int *repetitions, const double *lo, const double *hi)
  {
    if (dim != 2 && dim != 3)
      throw std::invalid_argument("BackgroundGrid: dim must be 2 or 3");
    std::array<int, 3> nd = {repetitions[0], repetitions[1], dim == 3 ? repetitions[2] : 1};
    for (int c = 0; c < dim; ++c)
      if (nd[c] < 1 || !(hi[c] > lo[c]))
        throw std::invalid_argument("BackgroundGrid: repetitions must be >= 1 and p2 > p1");
    const double l3[3] = {lo[0], lo[1], dim == 3 ? lo[2] : 0.0}, h3[3] = {hi[0], hi[1], dim == 3 ? hi[2] : 1.0};
    return build(dim, nd, l3, h3, false);
  }
  // GridIn<2>::read_msh [deal.II] for the gmsh 4.1 ASCII format, 4-node quadrilaterals only (what meshes/t3.msh of the
  // reference holds: examples/minimal_SIP.cc:94-118, test/polydeal/poisson_sanity_check_03.cc:103-113).  gmsh lists a quad
  // counter-clockwise; deal.II's vertex order is lexicographic (v at ((v&1),(v>>1)&1)), i.e. nodes (0,1,3,2).
  static BackgroundGrid read_msh(const std::string &path)
  {
    std::ifstream in(path);
    if (!in)
      throw std::invalid_argument("read_msh: cannot open " + path);
    std::string line;
    std::map<long long, std::array<double, 3>> nodes;
    std::vector<std::array<long long, 4>> quads;
    double version = 0.0;
    while (std::getline(in, line))
      {
        if (line.rfind("$MeshFormat", 0) == 0)
          {
            int ft, ds;
            in >> version >> ft >> ds;
            // (4.1 only: 4.0 interleaves tags and coordinates in $Nodes and orders the block headers differently)
            if (version < 4.1 - 1e-9 || version >= 4.2 - 1e-9 || ft != 0)
              throw std::invalid_argument("read_msh: only the ASCII format 4.1 is read");
          }
        else if (line.rfind("$Nodes", 0) == 0)
          {
            long long nblocks, nnodes, mn, mx;
            in >> nblocks >> nnodes >> mn >> mx;
            for (long long b = 0; b < nblocks; ++b)
              {
                int edim, etag, parametric;
                long long nb;
                in >> edim >> etag >> parametric >> nb;
                std::vector<long long> tags((size_t)nb);
                for (auto &t : tags)
                  in >> t;
                for (long long k = 0; k < nb; ++k)
                  {
                    std::array<double, 3> x;
                    in >> x[0] >> x[1] >> x[2];
                    for (int pc = 0; pc < (parametric ? edim : 0); ++pc)
                      {
                        double u;
                        in >> u;
                      }
                    nodes[tags[(size_t)k]] = x;
                  }
              }
          }
        else if (line.rfind("$Elements", 0) == 0)
          {
            long long nblocks, nel, mn, mx;
            in >> nblocks >> nel >> mn >> mx;
            for (long long b = 0; b < nblocks; ++b)
              {
                int edim, etag, etype;
                long long nb;
                in >> edim >> etag >> etype >> nb;
                // nodes per element of the types that can occur next to quads in a 2-D mesh file
                int npe = 0;
                switch (etype)
                  {
                  case 15: npe = 1; break; // point
                  case 1: npe = 2; break;  // line
                  case 2: npe = 3; break;  // triangle
                  case 3: npe = 4; break;  // quadrilateral
                  default: throw std::invalid_argument("read_msh: unsupported element type " + std::to_string(etype));
                  }
                for (long long k = 0; k < nb; ++k)
                  {
                    long long tag, nd[4] = {0, 0, 0, 0};
                    in >> tag;
                    for (int v = 0; v < npe; ++v)
                      in >> nd[v];
                    if (etype == 3)
                      quads.push_back({nd[0], nd[1], nd[2], nd[3]});
                    else if (etype == 2)
                      throw std::invalid_argument("read_msh: triangles are not supported (quadrilateral meshes only)");
                  }
              }
          }
        if (!in)
          throw std::invalid_argument("read_msh: malformed file " + path);
      }
    if (quads.empty())
      throw std::invalid_argument("read_msh: no quadrilaterals in " + path);
    BackgroundGrid g;
    g.dim = 2;
    g.unstructured = true;
    g.n_per_dir = 0;
    g.n_dir = {0, 0, 0};
    std::map<long long, int> vid; // node tag -> consecutive vertex number (ascending tags, like deal.II's vertex list)
    for (const auto &q : quads)
      for (long long t : q)
        if (!nodes.count(t))
          throw std::invalid_argument("read_msh: element refers to an unknown node");
        else
          vid.emplace(t, 0);
    int nvtx = 0;
    for (auto &kv : vid)
      kv.second = nvtx++;
    const size_t nc = quads.size();
    g.cell_ijk.assign(nc, {0, 0, 0});
    g.vertices.resize(nc * 4 * 2);
    g.vertex_id.resize(nc * 4);
    static const int order[4] = {0, 1, 3, 2};
    for (size_t c = 0; c < nc; ++c)
      {
        // orientation: counter-clockwise node lists have positive area; a clockwise one is mirrored (deal.II inverts such cells)
        double area2 = 0.0;
        for (int v = 0; v < 4; ++v)
          {
            const auto &a = nodes[quads[c][v]], &b = nodes[quads[c][(v + 1) % 4]];
            area2 += a[0] * b[1] - a[1] * b[0];
          }
        for (int v = 0; v < 4; ++v)
          {
            const long long tag = quads[c][area2 > 0 ? order[v] : order[v ^ 1]];
            g.vertex_id[c * 4 + v] = vid[tag];
            g.vertices[(c * 4 + v) * 2 + 0] = nodes[tag][0];
            g.vertices[(c * 4 + v) * 2 + 1] = nodes[tag][1];
          }
      }
    g.build_adjacency();
    return g;
  }
  // Triangulation::refine_global(times) [deal.II] for the unstructured 2-D mesh: every quadrilateral is split by its edge
  // midpoints and the mean of its vertices (FlatManifold on straight-sided cells); child c sits at ((c&1),(c>>1)) of the
  // parent and the children of cell i become cells 4i .. 4i+3 (all active cells are on one level, ordered by parent).
  void refine_global(int times)
  {
    if (!unstructured || dim != 2)
      throw std::logic_error("refine_global: implemented for meshes read by read_msh (structured grids are built at their final size)");
    for (int it = 0; it < times; ++it)
      {
        const size_t nc = (size_t)n_active_cells();
        int nvtx = 0;
        for (int v : vertex_id)
          nvtx = std::max(nvtx, v + 1);
        std::map<std::pair<int, int>, int> mid; // edge (sorted vertex pair) -> new vertex
        std::vector<double> nv_x;               // coordinates of the new vertices
        auto midpoint = [&](size_t c, int a, int b) {
          int va = vertex_id[c * 4 + a], vb = vertex_id[c * 4 + b];
          const std::pair<int, int> k = std::minmax(va, vb);
          auto f = mid.find(k);
          if (f != mid.end())
            return f->second;
          const int id = nvtx + (int)(nv_x.size() / 2);
          for (int d = 0; d < 2; ++d)
            nv_x.push_back(0.5 * (vertices[(c * 4 + a) * 2 + d] + vertices[(c * 4 + b) * 2 + d]));
          mid.emplace(k, id);
          return id;
        };
        std::vector<int> nid(nc * 16);
        std::vector<double> nx(nc * 16 * 2);
        auto coord = [&](size_t c, int id, int local, double *out) {
          if (local >= 0)
            {
              out[0] = vertices[(c * 4 + local) * 2], out[1] = vertices[(c * 4 + local) * 2 + 1];
              return;
            }
          const size_t k = (size_t)(id - nvtx);
          out[0] = nv_x[2 * k], out[1] = nv_x[2 * k + 1];
        };
        for (size_t c = 0; c < nc; ++c)
          {
            const int V[4] = {vertex_id[c * 4], vertex_id[c * 4 + 1], vertex_id[c * 4 + 2], vertex_id[c * 4 + 3]};
            const int EL = midpoint(c, 0, 2), ER = midpoint(c, 1, 3), EB = midpoint(c, 0, 1), ET = midpoint(c, 2, 3);
            const int C = nvtx + (int)(nv_x.size() / 2);
            for (int d = 0; d < 2; ++d)
              nv_x.push_back(0.25 * (vertices[(c * 4) * 2 + d] + vertices[(c * 4 + 1) * 2 + d] + vertices[(c * 4 + 2) * 2 + d] +
                                     vertices[(c * 4 + 3) * 2 + d]));
            const int ch[4][4] = {{V[0], EB, EL, C}, {EB, V[1], C, ER}, {EL, C, V[2], ET}, {C, ER, ET, V[3]}};
            const int loc[4][4] = {{0, -1, -1, -1}, {-1, 1, -1, -1}, {-1, -1, 2, -1}, {-1, -1, -1, 3}};
            for (int k = 0; k < 4; ++k)
              for (int v = 0; v < 4; ++v)
                {
                  nid[(c * 4 + k) * 4 + v] = ch[k][v];
                  coord(c, ch[k][v], loc[k][v], &nx[((c * 4 + k) * 4 + v) * 2]);
                }
          }
        vertex_id.swap(nid);
        vertices.swap(nx);
        cell_ijk.assign(nc * 4, {0, 0, 0});
        build_adjacency();
      }
  }
  int n_active_cells() const { return (int)cell_ijk.size(); }
  int n_faces_per_cell() const { return 2 * dim; }
  int nv() const { return 1 << dim; }
  const double *vertex(int cell, int v) const { return &vertices[((size_t)cell * nv() + v) * dim]; }
  int neighbor(int cell, int f) const
  {
    if (unstructured)
      return nbr_tab[(size_t)cell * n_faces_per_cell() + f];
    std::array<int, 3> ijk = cell_ijk[cell];
    const int ax = f / 2;
    ijk[ax] += (f & 1) ? 1 : -1;
    if (ijk[ax] < 0 || ijk[ax] >= n_dir[ax])
      return invalid_index;
    return lut[lin(ijk)];
  }
  // which face of neighbor(cell, f) is the shared one (cell->neighbor_of_neighbor(f) [deal.II])
  int neighbor_of_neighbor(int cell, int f) const { return unstructured ? nofn_tab[(size_t)cell * n_faces_per_cell() + f] : (f ^ 1); }
  int cell_at(int ix, int iy, int iz = 0) const
  {
    if (unstructured)
      throw std::logic_error("cell_at: the mesh has no ijk lattice");
    return lut[lin({ix, iy, iz})];
  }
  // Do the quadrature points of face f of `cell` run AGAINST the direction both cells sharing the face agree on (ascending
  // global vertex number)?  deal.II gets matching points on the two sides of a face from the face's own orientation
  // (line_orientation); on the ijk lattice both neighbours walk a face the same way and nothing has to be reversed.
  bool face_points_reversed(int cell, int f) const
  {
    if (!unstructured)
      return false;
    // 2-D faces: 0: v0->v2, 1: v1->v3, 2: v0->v1, 3: v2->v3 (the free unit coordinate increases from the first to the second)
    static const int fv[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};
    return vertex_id[(size_t)cell * 4 + fv[f][0]] > vertex_id[(size_t)cell * 4 + fv[f][1]];
  }

  // random interior-vertex jitter (stand-in for GridTools::distort_random, exact_solutions_dgp.cc:306)
  void distort(double factor, unsigned seed)
  {
    if (unstructured)
      throw std::logic_error("distort: implemented for lattice grids");
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    const double h = vertex(0, nv() - 1)[0] - vertex(0, 0)[0];
    size_t nvert = 1;
    for (int c = 0; c < dim; ++c)
      nvert *= (size_t)(n_dir[c] + 1);
    std::vector<double> jit(nvert * dim, 0.0);
    for (size_t v = 0; v < nvert; ++v)
      {
        size_t r = v;
        bool interior = true;
        for (int c = 0; c < dim; ++c)
          {
            const int i = (int)(r % (n_dir[c] + 1));
            r /= (n_dir[c] + 1);
            if (i == 0 || i == n_dir[c])
              interior = false;
          }
        for (int c = 0; c < dim; ++c)
          {
            const double u = U(rng);
            if (interior)
              jit[v * dim + c] = u * factor * h;
          }
      }
    for (int cell = 0; cell < n_active_cells(); ++cell)
      for (int v = 0; v < nv(); ++v)
        {
          size_t vid = 0, mul = 1;
          for (int c = 0; c < dim; ++c)
            {
              vid += mul * (size_t)(cell_ijk[cell][c] + ((v >> c) & 1));
              mul *= (size_t)(n_dir[c] + 1);
            }
          for (int c = 0; c < dim; ++c)
            vertices[((size_t)cell * nv() + v) * dim + c] += jit[vid * dim + c];
        }
  }

private:
  // neighbour tables of an unstructured 2-D mesh from shared edges
  void build_adjacency()
  {
    static const int fv[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};
    const size_t nc = (size_t)n_active_cells();
    nbr_tab.assign(nc * 4, invalid_index);
    nofn_tab.assign(nc * 4, invalid_index);
    std::map<std::pair<int, int>, std::pair<int, int>> first; // edge -> (cell, face) that met it first
    for (size_t c = 0; c < nc; ++c)
      for (int f = 0; f < 4; ++f)
        {
          const std::pair<int, int> k = std::minmax(vertex_id[c * 4 + fv[f][0]], vertex_id[c * 4 + fv[f][1]]);
          auto it = first.find(k);
          if (it == first.end())
            first.emplace(k, std::make_pair((int)c, f));
          else
            {
              const int oc = it->second.first, of = it->second.second;
              if (nbr_tab[(size_t)oc * 4 + of] != invalid_index)
                throw std::invalid_argument("mesh: an edge is shared by more than two cells");
              nbr_tab[c * 4 + f] = oc, nofn_tab[c * 4 + f] = of;
              nbr_tab[(size_t)oc * 4 + of] = (int)c, nofn_tab[(size_t)oc * 4 + of] = f;
            }
        }
  }
  int lin(const std::array<int, 3> &ijk) const
  {
    return ijk[0] + n_dir[0] * (ijk[1] + (dim == 3 ? n_dir[1] * ijk[2] : 0));
  }
  static BackgroundGrid build(int dim, int n, double lo, double hi, bool morton)
  {
    if (dim != 2 && dim != 3)
      throw std::invalid_argument("BackgroundGrid: dim must be 2 or 3");
    const double l3[3] = {lo, lo, lo}, h3[3] = {hi, hi, hi};
    return build(dim, std::array<int, 3>{n, n, dim == 3 ? n : 1}, l3, h3, morton);
  }
  static BackgroundGrid build(int dim, const std::array<int, 3> &nd, const double *lo, const double *hi, bool morton)
  {
    BackgroundGrid g;
    g.dim = dim;
    g.n_per_dir = nd[0];
    g.n_dir = nd;
    const int n = nd[0];
    const size_t nc = (size_t)nd[0] * nd[1] * (dim == 3 ? nd[2] : 1);
    g.cell_ijk.resize(nc);
    g.lut.assign(nc, 0);
    int levels = 0;
    while ((1 << levels) < n)
      ++levels;
    if (morton && ((1 << levels) != n || nd[1] != n || (dim == 3 && nd[2] != n)))
      throw std::invalid_argument("BackgroundGrid: Morton order needs a power-of-two cube");
    for (size_t l = 0; l < nc; ++l)
      {
        std::array<int, 3> ijk = {(int)(l % nd[0]), (int)((l / nd[0]) % nd[1]), (int)(dim == 3 ? l / ((size_t)nd[0] * nd[1]) : 0)};
        size_t idx = l;
        if (morton)
          {
            idx = 0;
            for (int lev = 0; lev < levels; ++lev)
              for (int c = 0; c < dim; ++c)
                idx |= (size_t)((ijk[c] >> lev) & 1) << (dim * lev + c);
          }
        g.cell_ijk[idx] = ijk;
        g.lut[l] = (int)idx;
      }
    g.vertices.resize(nc * (size_t)(1 << dim) * dim);
    for (size_t cell = 0; cell < nc; ++cell)
      for (int v = 0; v < (1 << dim); ++v)
        for (int c = 0; c < dim; ++c)
          g.vertices[(cell * (1 << dim) + v) * dim + c] = lo[c] + (g.cell_ijk[cell][c] + ((v >> c) & 1)) * ((hi[c] - lo[c]) / nd[c]);
    return g;
  }
};

// ---------------------------------------------------------------------------------------------------
// Q1 mapping of a cell / a face: what FEValues / FEFaceValues on FE_Nothing give the reference
// (source/agglomeration_handler.cc:224-235, 639-653, 1146-1165).
// ---------------------------------------------------------------------------------------------------
struct QPoints
{
  std::vector<double> x, n; // [npts][dim]
  std::vector<double> w;    // JxW
};

inline void q1_map(const BackgroundGrid &g, int cell, const double *xi, double *x, double J[3][3])
{
  const int dim = g.dim, nv = g.nv();
  for (int r = 0; r < dim; ++r)
    {
      x[r] = 0.0;
      for (int c = 0; c < dim; ++c)
        J[r][c] = 0.0;
    }
  for (int v = 0; v < nv; ++v)
    {
      double N = 1.0, dN[3] = {1.0, 1.0, 1.0};
      for (int c = 0; c < dim; ++c)
        {
          const int b = (v >> c) & 1;
          const double f = b ? xi[c] : 1.0 - xi[c];
          const double df = b ? 1.0 : -1.0;
          N *= f;
          for (int d = 0; d < dim; ++d)
            dN[d] *= (d == c) ? df : f;
        }
      const double *X = g.vertex(cell, v);
      for (int r = 0; r < dim; ++r)
        {
          x[r] += N * X[r];
          for (int c = 0; c < dim; ++c)
            J[r][c] += dN[c] * X[r];
        }
    }
}

// QGauss<dim>(nq) on a cell: real points + JxW, x fastest  [deal.II]
inline void cell_quadrature(const BackgroundGrid &g, int cell, int nq, QPoints &out)
{
  std::vector<double> x1, w1;
  qgauss_1d(nq, x1, w1);
  const int dim = g.dim;
  const int np = (dim == 2) ? nq * nq : nq * nq * nq;
  for (int q = 0; q < np; ++q)
    {
      const int i[3] = {q % nq, (q / nq) % nq, q / (nq * nq)};
      double xi[3], x[3], J[3][3], w = 1.0;
      for (int c = 0; c < dim; ++c)
        {
          xi[c] = x1[i[c]];
          w *= w1[i[c]];
        }
      q1_map(g, cell, xi, x, J);
      double det;
      if (dim == 2)
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      else
        det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
              J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      for (int c = 0; c < dim; ++c)
        out.x.push_back(x[c]);
      out.w.push_back(w * std::fabs(det));
    }
}

// QGauss<dim-1>(nqf) on face f of a cell: real points, JxW, outward unit normal.  Point order follows
// QProjector::project_to_face [deal.II]: 3-D faces 0/1 -> (s,t)=(y,z), 2/3 -> (z,x), 4/5 -> (x,y), s fastest.
inline void face_quadrature(const BackgroundGrid &g, int cell, int f, int nqf, QPoints &out)
{
  std::vector<double> x1, w1;
  qgauss_1d(nqf, x1, w1);
  const int dim = g.dim, ax = f / 2, side = f & 1;
  const int np = (dim == 2) ? nqf : nqf * nqf;
  const bool rev = g.face_points_reversed(cell, f);
  int tang[2];
  if (dim == 2)
    tang[0] = tang[1] = 1 - ax;
  else if (ax == 0)
    tang[0] = 1, tang[1] = 2;
  else if (ax == 1)
    tang[0] = 2, tang[1] = 0;
  else
    tang[0] = 0, tang[1] = 1;
  for (int q = 0; q < np; ++q)
    {
      double xi[3] = {0, 0, 0}, x[3], J[3][3], w;
      xi[ax] = side;
      if (dim == 2)
        {
          const int qq = rev ? nqf - 1 - q : q; // the direction both sides of the face agree on (face_points_reversed)
          xi[tang[0]] = x1[qq];
          w = w1[qq];
        }
      else
        {
          xi[tang[0]] = x1[q % nqf];
          xi[tang[1]] = x1[q / nqf];
          w = w1[q % nqf] * w1[q / nqf];
        }
      q1_map(g, cell, xi, x, J);
      double nr[3] = {0, 0, 0}, area;
      if (dim == 2)
        {
          const double t0 = J[0][tang[0]], t1 = J[1][tang[0]];
          area = std::sqrt(t0 * t0 + t1 * t1);
          nr[0] = t1;
          nr[1] = -t0;
        }
      else
        {
          const double a[3] = {J[0][tang[0]], J[1][tang[0]], J[2][tang[0]]};
          const double b[3] = {J[0][tang[1]], J[1][tang[1]], J[2][tang[1]]};
          nr[0] = a[1] * b[2] - a[2] * b[1];
          nr[1] = a[2] * b[0] - a[0] * b[2];
          nr[2] = a[0] * b[1] - a[1] * b[0];
          area = std::sqrt(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
        }
      double nn = 0.0, dot = 0.0;
      for (int c = 0; c < dim; ++c)
        nn += nr[c] * nr[c];
      nn = std::sqrt(nn);
      for (int c = 0; c < dim; ++c)
        {
          nr[c] /= nn;
          dot += nr[c] * J[c][ax] * (side ? 1.0 : -1.0);
        }
      const double sgn = dot < 0 ? -1.0 : 1.0;
      for (int c = 0; c < dim; ++c)
        {
          out.x.push_back(x[c]);
          out.n.push_back(sgn * nr[c]);
        }
      out.w.push_back(w * area);
    }
}

// ---------------------------------------------------------------------------------------------------
// Flattening helpers: the same arithmetic as cell_quadrature / face_quadrature, written straight into the
// structure-of-arrays of a pdh_problem at a known offset (no per-point allocation), so that the points of different
// polytopes / faces can be generated by several host threads.  PDH_HOST_THREADS overrides the thread count.
// ---------------------------------------------------------------------------------------------------
template <class F>
inline void parallel_for(size_t n, F &&fn)
{
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *e = std::getenv("PDH_HOST_THREADS"))
    nt = (unsigned)std::max(1, std::atoi(e));
  nt = std::max(1u, std::min<unsigned>(nt, 64u));
  if (nt == 1 || n < 256)
    {
      for (size_t i = 0; i < n; ++i)
        fn(i);
      return;
    }
  std::vector<std::thread> th;
  std::exception_ptr err;
  std::mutex mx;
  const size_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t)
    {
      const size_t b = (size_t)t * chunk, e = std::min(n, b + chunk);
      if (b >= e)
        break;
      th.emplace_back([&, b, e] {
        try
          {
            for (size_t i = b; i < e; ++i)
              fn(i);
          }
        catch (...)
          {
            std::lock_guard<std::mutex> lk(mx);
            err = std::current_exception();
          }
      });
    }
  for (auto &t : th)
    t.join();
  if (err)
    std::rethrow_exception(err);
}

// volume points of one cell -> X[c * stride + at + q], W[at + q]
inline void cell_quadrature_soa(const BackgroundGrid &g, int cell, int nq, const double *x1, const double *w1, double *X,
                                size_t stride, size_t at, double *W)
{
  const int dim = g.dim;
  const int np = (dim == 2) ? nq * nq : nq * nq * nq;
  for (int q = 0; q < np; ++q)
    {
      const int i[3] = {q % nq, (q / nq) % nq, q / (nq * nq)};
      double xi[3], x[3], J[3][3], w = 1.0;
      for (int c = 0; c < dim; ++c)
        {
          xi[c] = x1[i[c]];
          w *= w1[i[c]];
        }
      q1_map(g, cell, xi, x, J);
      double det;
      if (dim == 2)
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      else
        det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
              J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      for (int c = 0; c < dim; ++c)
        X[c * stride + at + q] = x[c];
      W[at + q] = w * std::fabs(det);
    }
}

// face points of (cell, f): X, N (may be null), W at offset `at`
inline void face_quadrature_soa(const BackgroundGrid &g, int cell, int f, int nqf, const double *x1, const double *w1, double *X,
                                double *N, size_t stride, size_t at, double *W)
{
  const int dim = g.dim, ax = f / 2, side = f & 1;
  const int np = (dim == 2) ? nqf : nqf * nqf;
  const bool rev = g.face_points_reversed(cell, f);
  int tang[2];
  if (dim == 2)
    tang[0] = tang[1] = 1 - ax;
  else if (ax == 0)
    tang[0] = 1, tang[1] = 2;
  else if (ax == 1)
    tang[0] = 2, tang[1] = 0;
  else
    tang[0] = 0, tang[1] = 1;
  for (int q = 0; q < np; ++q)
    {
      double xi[3] = {0, 0, 0}, x[3], J[3][3], w;
      xi[ax] = side;
      if (dim == 2)
        {
          const int qq = rev ? nqf - 1 - q : q; // the direction both sides of the face agree on (face_points_reversed)
          xi[tang[0]] = x1[qq];
          w = w1[qq];
        }
      else
        {
          xi[tang[0]] = x1[q % nqf];
          xi[tang[1]] = x1[q / nqf];
          w = w1[q % nqf] * w1[q / nqf];
        }
      q1_map(g, cell, xi, x, J);
      double nr[3] = {0, 0, 0}, area;
      if (dim == 2)
        {
          const double t0 = J[0][tang[0]], t1 = J[1][tang[0]];
          area = std::sqrt(t0 * t0 + t1 * t1);
          nr[0] = t1;
          nr[1] = -t0;
        }
      else
        {
          const double a[3] = {J[0][tang[0]], J[1][tang[0]], J[2][tang[0]]};
          const double b[3] = {J[0][tang[1]], J[1][tang[1]], J[2][tang[1]]};
          nr[0] = a[1] * b[2] - a[2] * b[1];
          nr[1] = a[2] * b[0] - a[0] * b[2];
          nr[2] = a[0] * b[1] - a[1] * b[0];
          area = std::sqrt(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
        }
      double nn = 0.0, dot = 0.0;
      for (int c = 0; c < dim; ++c)
        nn += nr[c] * nr[c];
      nn = std::sqrt(nn);
      for (int c = 0; c < dim; ++c)
        {
          nr[c] /= nn;
          dot += nr[c] * J[c][ax] * (side ? 1.0 : -1.0);
        }
      const double sgn = dot < 0 ? -1.0 : 1.0;
      if (X)
        for (int c = 0; c < dim; ++c)
          {
            X[c * stride + at + q] = x[c];
            N[c * stride + at + q] = sgn * nr[c];
          }
      W[at + q] = w * area;
    }
}

// ---------------------------------------------------------------------------------------------------
// SIP variants of the reference's callers (SURVEY.md 8(a), table "Variants")
// ---------------------------------------------------------------------------------------------------
struct SipVariant
{
  double penalty_constant = -1.0; // < 0: 10 (p+dim)(p+1)   (include/poly_utils.h:2018-2019)
  int owner_rule = 0;             // 0: id() < id() (poly_utils.h:2089); 1: index() < index() (poisson.cc:841)
  int h_rule = 0;                 // 0: C / diameter(owner); 1: C (h_f = 1); 2: C max(1/h_in, 1/h_out)
  int boundary = 0;               // 0: Nitsche; 1: zeroed (examples/minimal_SIP.cc:230-248)
  double reaction_c = 0.0;        // examples/diffusion_reaction.cc:495-501

  static SipVariant assemble_dg_matrix() { return SipVariant(); }
  static SipVariant poisson_example(const FiniteElement &fe) // examples/poisson.cc:476, 841, 900-901
  {
    SipVariant v;
    v.penalty_constant = 10.0 * (fe.degree + 1) * (fe.degree + fe.dim);
    v.owner_rule = 1;
    return v;
  }
  static SipVariant minimal_sip_test() // test/polydeal/minimal_SIP_Poisson.cc:101, 308
  {
    SipVariant v;
    v.penalty_constant = 20.0;
    v.owner_rule = 1;
    v.h_rule = 1;
    return v;
  }
  static SipVariant minimal_sip_example() // examples/minimal_SIP.cc:230-262
  {
    SipVariant v;
    v.penalty_constant = 10.0;
    v.owner_rule = 1;
    v.h_rule = 2;
    v.boundary = 1;
    return v;
  }
  static SipVariant diffusion_reaction(const FiniteElement &fe) // examples/diffusion_reaction.cc:366, 515, 563
  {
    SipVariant v;
    v.penalty_constant = 10.0 * fe.degree * fe.degree;
    v.reaction_c = 0.5;
    return v;
  }
};

// Flattened problem: owns the arrays a pdh_problem points to.
// std::vector whose resize() leaves new elements uninitialised: the point arrays (about 1 GB on the bench workload) are
// filled by all host threads right after they are sized; a serial zero-fill in between costs more than the fill.
template <class T>
struct UninitAllocator : std::allocator<T>
{
  template <class U>
  struct rebind
  {
    using other = UninitAllocator<U>;
  };
  template <class U, class... A>
  void construct(U *ptr, A &&...a)
  {
    if constexpr (sizeof...(A) == 0)
      ::new ((void *)ptr) U;
    else
      ::new ((void *)ptr) U(std::forward<A>(a)...);
  }
};
using PointArray = std::vector<double, UninitAllocator<double>>;

struct FlatProblem
{
  pdh_problem c{};
  std::vector<double> bbox, face_sigma;
  PointArray vq_x, vq_w, fq_x, fq_n, fq_w, fq_w_out;
  std::vector<int32_t> dof_offset, face_in, face_out, colind, col_offset, agg_rank;
  std::vector<int64_t> vq_ptr, fq_ptr, rowptr;
  // flatten_cartesian: the compact description of the points (include/polydeal_hip.h: pdh_cartesian_points) instead of the points
  pdh_cartesian_points cart{};
  std::vector<double> cart_box;
  std::vector<int32_t> cart_vq_cell, cart_fq_cell, cart_fq_face;
  bool cartesian = false;
  void bind()
  {
    if (cartesian)
      {
        cart.n_cells = (int32_t)(cart_box.size() / 6);
        cart.cell_box = cart_box.data();
        cart.vq_cell = cart_vq_cell.data();
        cart.fq_cell = cart_fq_cell.data();
        cart.fq_face = cart_fq_face.data();
      }
    c.bbox = bbox.data();
    c.dof_offset = dof_offset.data();
    c.vq_ptr = vq_ptr.data();
    c.vq_x = cartesian ? nullptr : vq_x.data();
    c.vq_w = cartesian ? nullptr : vq_w.data();
    c.face_in = face_in.data();
    c.face_out = face_out.data();
    c.fq_ptr = fq_ptr.data();
    c.fq_x = cartesian ? nullptr : fq_x.data();
    c.fq_n = cartesian ? nullptr : fq_n.data();
    c.fq_w = cartesian ? nullptr : fq_w.data();
    c.fq_w_out = (cartesian || fq_w_out.empty()) ? nullptr : fq_w_out.data();
    c.face_sigma = face_sigma.data();
    c.rowptr = rowptr.data();
    c.colind = colind.empty() ? nullptr : colind.data();
    c.col_offset = col_offset.empty() ? nullptr : col_offset.data();
    c.agg_rank = agg_rank.empty() ? nullptr : agg_rank.data();
  }
};

// ---------------------------------------------------------------------------------------------------
// AgglomerationHandler
// ---------------------------------------------------------------------------------------------------
class AgglomerationHandler
{
public:
  explicit AgglomerationHandler(const BackgroundGrid &grid)
    : tria(&grid)
    , master_of(grid.n_active_cells(), invalid_index)
  {}

  // source/agglomeration_handler.cc:44-104: cells[0] is the master, polytope index = call order
  int define_agglomerate(const std::vector<int> &cells)
  {
    if (cells.empty())
      throw std::invalid_argument("No cells to be agglomerated.");
    const int master = cells[0];
    const int poly = (int)master_cells.size();
    master_cells.push_back(master);
    master_of.at(master) = master;
    std::vector<int> slaves(cells.begin() + 1, cells.end());
    for (int s : slaves)
      master_of.at(s) = master;
    master2slaves[master] = slaves;
    master2polygon[master] = poly;
    // create_bounding_box (:476-491): box of all vertices of all cells
    const int dim = tria->dim;
    std::array<double, 6> bb;
    for (int c = 0; c < dim; ++c)
      {
        bb[c] = 1e300;
        bb[3 + c] = -1e300;
      }
    for (int cell : cells)
      for (int v = 0; v < tria->nv(); ++v)
        for (int c = 0; c < dim; ++c)
          {
            bb[c] = std::min(bb[c], tria->vertex(cell, v)[c]);
            bb[3 + c] = std::max(bb[3 + c], tria->vertex(cell, v)[c]);
          }
    bboxes.push_back(bb);
    connectivity_ready = false;
    return poly;
  }

  unsigned int n_agglomerates() const { return (unsigned int)master_cells.size(); }
  const BackgroundGrid &get_triangulation() const { return *tria; }
  const FiniteElement &get_fe() const { return fe; }
  unsigned int n_dofs_per_cell() const { return (unsigned int)fe.n_dofs_per_cell(); }
  unsigned int n_dofs() const { return n_dofs_; }
  int master_index(int P) const { return master_cells.at(P); }
  int cell_to_polytope_index(int cell) const { return master2polygon.at(master_of.at(cell)); }
  bool is_master_cell(int cell) const { return master_of.at(cell) == cell; }
  // value stored in master_slave_relationships (include/agglomeration_handler.h:688)
  int master_slave_value(int cell) const { return is_master_cell(cell) ? -1 : master_of.at(cell); }
  // slaves in insertion order, then the master (include/agglomeration_handler.h:1022-1032)
  std::vector<int> get_agglomerate(int P) const
  {
    const int m = master_cells.at(P);
    std::vector<int> a = master2slaves.at(m);
    a.push_back(m);
    return a;
  }
  const std::array<double, 6> &bbox(int P) const { return bboxes.at(P); }

  // source/agglomeration_handler.cc:210-236: number of Gauss points per direction
  void initialize_fe_values(int n_q_points_1d, int n_face_q_points_1d)
  {
    nq = n_q_points_1d;
    nqf = n_face_q_points_1d;
  }

  // source/agglomeration_handler.cc:326-379 (+711-725): masters get the FE, slaves FE_Nothing; dofs are
  // numbered consecutively over master cells in ascending active-cell order [deal.II]
  void distribute_agglomerated_dofs(const FiniteElement &fe_space)
  {
    for (int m : master_of)
      if (m == invalid_index)
        throw std::logic_error("every cell must belong to an agglomerate before distributing dofs");
    fe = fe_space;
    if (fe.dim != tria->dim)
      throw std::invalid_argument("finite element and triangulation dimensions differ");
    const int n = fe.n_dofs_per_cell();
    std::vector<int> order(master_cells.size());
    for (size_t i = 0; i < order.size(); ++i)
      order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return master_cells[a] < master_cells[b]; });
    dof_offset.assign(master_cells.size(), 0);
    for (size_t r = 0; r < order.size(); ++r)
      dof_offset[order[r]] = (int)r * n;
    n_dofs_ = (unsigned int)(n * master_cells.size());
    setup_connectivity_of_agglomeration();
  }

  // ---- accessor-level queries (include/agglomeration_accessor.h) ----------------------------------
  unsigned int n_faces(int P) const { return (unsigned int)face_nbr.at(P).size(); }                 // :324-331
  bool at_boundary(int P, unsigned f) const { return face_nbr.at(P).at(f) == invalid_index; }       // :736-772
  int neighbor(int P, unsigned f) const { return face_nbr.at(P).at(f); }                           // :335-422
  int neighbor_of_agglomerated_neighbor(int P, unsigned f) const                                    // :426-481
  {
    if (at_boundary(P, f))
      return invalid_index;
    const int Q = neighbor(P, f);
    for (unsigned fo = 0; fo < n_faces(Q); ++fo)
      if (!at_boundary(Q, fo) && neighbor(Q, fo) == P)
        return (int)fo;
    return invalid_index;
  }
  double diameter(int P) const // :582-601: diagonal of the bounding box
  {
    double s = 0;
    for (int c = 0; c < tria->dim; ++c)
      s += (bboxes[P][3 + c] - bboxes[P][c]) * (bboxes[P][3 + c] - bboxes[P][c]);
    return std::sqrt(s);
  }
  double volume(int P) const // :617-632
  {
    double v = 1;
    for (int c = 0; c < tria->dim; ++c)
      v *= bboxes[P][3 + c] - bboxes[P][c];
    return v;
  }
  void get_dof_indices(int P, std::vector<unsigned int> &idx) const // :534-558
  {
    idx.resize(n_dofs_per_cell());
    for (unsigned i = 0; i < idx.size(); ++i)
      idx[i] = (unsigned)dof_offset.at(P) + i;
  }
  int dof_offset_of(int P) const { return dof_offset.at(P); }
  // polytope_cache.interface[{id_in,id_out}] (include/agglomeration_handler.h:355-356); Q == P: boundary
  const std::vector<std::pair<int, int>> &get_interface(int P, int Q) const { return interface.at(key(P, Q)); }

  // ---- source/agglomeration_handler.cc:910-1022 -----------------------------------------------------
  // CSR of the DG pattern: own block + one block per valid neighbour.  diag_first = deal.II SparsityPattern
  // (diagonal entry first, then ascending) [deal.II]; otherwise plain ascending (DynamicSparsityPattern/Epetra).
  void create_agglomeration_sparsity_pattern(std::vector<int64_t> &rowptr, std::vector<int32_t> *colind,
                                             bool diag_first = true) const
  {
    const int n = fe.n_dofs_per_cell();
    rowptr.assign((size_t)n_dofs_ + 1, 0);
    std::vector<std::vector<int>> blocks(master_cells.size());
    for (size_t P = 0; P < master_cells.size(); ++P)
      {
        auto &b = blocks[P];
        b.push_back(dof_offset[P]);
        for (int Q : face_nbr[P])
          if (Q != invalid_index)
            b.push_back(dof_offset[Q]);
        std::sort(b.begin(), b.end());
        for (int i = 0; i < n; ++i)
          rowptr[(size_t)dof_offset[P] + i + 1] = (int64_t)b.size() * n;
      }
    for (size_t r = 0; r < n_dofs_; ++r)
      rowptr[r + 1] += rowptr[r];
    if (!colind)
      return;
    colind->resize((size_t)rowptr[n_dofs_]);
    for (size_t P = 0; P < master_cells.size(); ++P)
      for (int i = 0; i < n; ++i)
        {
          const int r = dof_offset[P] + i;
          int32_t *out = colind->data() + rowptr[r];
          if (diag_first)
            *out++ = r;
          for (int off : blocks[P])
            for (int j = 0; j < n; ++j)
              if (!(diag_first && off + j == r))
                *out++ = off + j;
        }
  }

  // ---- quadrature views used by tests ----------------------------------------------------------------
  // source/agglomeration_handler.cc:622-707: concatenation over get_agglomerate() order
  void agglomerated_quadrature(int P, QPoints &out) const
  {
    for (int cell : get_agglomerate(P))
      cell_quadrature(*tria, cell, nq, out);
  }
  // reinit_master, source/agglomeration_handler.cc:1129-1165
  void face_quadrature_of(int P, unsigned f, QPoints &out) const
  {
    const int Q = at_boundary(P, f) ? P : neighbor(P, f);
    for (const auto &cf : get_interface(P, Q))
      face_quadrature(*tria, cf.first, cf.second, nqf, out);
  }

  // ---- flattening for the HIP path (what the deal.II adapter does by walking the public API) ----------
  bool owns(const SipVariant &v, int P, int Q) const
  {
    return v.owner_rule == 1 ? (P < Q) : (master_cells[P] < master_cells[Q]); // index() vs id() (CellId order)
  }
  double sigma(const SipVariant &v, int P, int Q) const
  {
    const double C = v.penalty_constant >= 0 ? v.penalty_constant : 10.0 * (fe.degree + fe.dim) * (fe.degree + 1);
    if (v.h_rule == 1)
      return C;
    if (v.h_rule == 2 && Q != invalid_index)
      return C * std::max(1.0 / diameter(P), 1.0 / diameter(Q));
    return C / std::fabs(diameter(P));
  }

  void flatten(const SipVariant &var, FlatProblem &F, bool diag_first = true, bool with_colind = false) const
  {
    flatten_impl(var, F, 0, (int)n_dofs_, false, diag_first, with_colind, nullptr, nullptr, false);
  }
  // The same description WITHOUT the points: for agglomerates of Cartesian cells (axis-aligned boxes; 3-D) every group of points is
  // named by its cell (and local face) and generated on the device by pdh_set_problem_cartesian - the gather the reference times
  // (source/agglomeration_handler.cc:622-707, 1103-1243) leaves the host.  Throws if a cell is not a box.
  void flatten_cartesian(const SipVariant &var, FlatProblem &F, bool diag_first = true, bool with_colind = false) const
  {
    flatten_impl(var, F, 0, (int)n_dofs_, false, diag_first, with_colind, nullptr, nullptr, false, true);
  }
  void flatten_local_cartesian(const SipVariant &var, FlatProblem &F, int row_begin, int row_end, bool diag_first = true,
                               bool with_colind = false, std::vector<int> *local_of = nullptr,
                               const std::vector<int> *row_splits = nullptr, bool epetra_columns = false) const
  {
    flatten_impl(var, F, row_begin, row_end, true, diag_first, with_colind, local_of, row_splits, epetra_columns, true);
  }

  // Rank-local description (pdh_problem::local = 1) of the dof rows [row_begin,row_end): what one MPI rank of the
  // reference holds after setup_ghost_polytopes / the ghost exchanges (source/agglomeration_handler.cc:1026-1091,
  // 531-618) - its own polytopes plus the ghost polytopes across the partition boundary with their bounding boxes
  // and GLOBAL dof indices (recv_ghosted_bbox / recv_ghost_dofs, :1081-1090).  Local numbering: owned polytopes first
  // (polytope order), then the ghosts in order of first contact.  Volume quadrature for the owned ones only; every face
  // with an owned side, described from its owner side exactly as in flatten().  rowptr / colind cover the owned rows
  // only (colind holds global columns); `local_of` (optional) receives the global polytope index of every local one.
  // epetra_columns: col_offset = local column ids of an Epetra column map (owned columns first, ghosts behind in
  // ascending global order) and rows sorted by them - the layout of TrilinosWrappers::SparseMatrix; needs !diag_first.
  void flatten_local(const SipVariant &var, FlatProblem &F, int row_begin, int row_end, bool diag_first = true,
                     bool with_colind = false, std::vector<int> *local_of = nullptr,
                     const std::vector<int> *row_splits = nullptr, bool epetra_columns = false) const
  {
    flatten_impl(var, F, row_begin, row_end, true, diag_first, with_colind, local_of, row_splits, epetra_columns);
  }

  // Shared implementation.  Bookkeeping (which polytopes / faces, offsets of their points) runs serially; the quadrature
  // points themselves - 16.8 M volume and 12 M face-side points on BASELINE configs[2] - are generated by all host
  // threads straight into the structure-of-arrays of the description.
  void flatten_impl(const SipVariant &var, FlatProblem &F, int row_begin, int row_end, bool local, bool diag_first,
                    bool with_colind, std::vector<int> *local_of, const std::vector<int> *row_splits,
                    bool epetra_columns, bool cartesian = false) const
  {
    if (cartesian && tria->dim != 3)
      throw std::invalid_argument("flatten_cartesian: 3-D only");
    if (!connectivity_ready)
      throw std::logic_error("distribute_agglomerated_dofs must be called first");
    if (nq <= 0 || nqf <= 0)
      throw std::logic_error("initialize_fe_values must be called first");
    if (epetra_columns && diag_first)
      throw std::invalid_argument("Epetra column order needs the ascending row layout");
    const int dim = tria->dim, nA = (int)master_cells.size(), n = fe.n_dofs_per_cell();
    auto owned = [&](int P) { return dof_offset[P] >= row_begin && dof_offset[P] < row_end; };
    std::vector<int> loc(nA, -1), glob;
    for (int P = 0; P < nA; ++P)
      if (owned(P))
        {
          loc[P] = (int)glob.size();
          glob.push_back(P);
        }
    const int n_owned = (int)glob.size();
    if ((int64_t)n_owned * n != (int64_t)row_end - row_begin)
      throw std::invalid_argument("row range must consist of whole polytopes");
    for (int i = 0; i < n_owned; ++i)
      for (int Q : face_nbr[glob[i]])
        if (Q != invalid_index && loc[Q] < 0)
          {
            loc[Q] = (int)glob.size();
            glob.push_back(Q);
          }
    const int nL = (int)glob.size();
    F = FlatProblem();
    F.c.dim = dim;
    F.c.degree = fe.degree;
    F.c.basis = fe.basis;
    F.c.n_agg = nL;
    F.c.n_rows = (int32_t)n_dofs_;
    F.c.diag_first = diag_first ? 1 : 0;
    F.c.local = local ? 1 : 0;
    F.c.reaction_c = var.reaction_c;
    F.c.fq_tensor_n = nqf; // likewise QGauss<dim-1>(nqf) per sub-face
    F.c.vq_tensor_n = nq; // QGauss<dim>(nq) per sub-cell; the library verifies the tensor structure (fails on distorted cells)
    F.bbox.resize((size_t)nL * 2 * dim);
    F.dof_offset.resize(nL);
    const int64_t npc = (dim == 2) ? (int64_t)nq * nq : (int64_t)nq * nq * nq;       // points per cell
    const int64_t npf = (dim == 2) ? (int64_t)nqf : (int64_t)nqf * nqf;             // points per sub-face
    F.vq_ptr.assign(1, 0);
    for (int l = 0; l < nL; ++l)
      {
        const int P = glob[l];
        for (int c = 0; c < dim; ++c)
          {
            F.bbox[(size_t)l * 2 * dim + c] = bboxes[P][c];
            F.bbox[(size_t)l * 2 * dim + dim + c] = bboxes[P][3 + c];
          }
        F.dof_offset[l] = dof_offset[P];
        const int64_t cnt = l < n_owned ? npc * (int64_t)(master2slaves.at(master_cells[P]).size() + 1) : 0;
        F.vq_ptr.push_back(F.vq_ptr.back() + cnt);
      }
    const size_t nqt = (size_t)F.vq_ptr.back();
    F.cartesian = cartesian;
    if (cartesian)
      { // boxes of all cells (a cell that is not an axis-aligned box ends the attempt), then the cell of every group of volume points
        const int nc = tria->n_active_cells();
        F.cart_box.resize((size_t)nc * 6);
        std::vector<char> bad((size_t)nc, 0);
        parallel_for((size_t)nc, [&](size_t cell) {
          const double *v0 = tria->vertex((int)cell, 0), *v7 = tria->vertex((int)cell, 7);
          for (int c = 0; c < 3; ++c)
            {
              F.cart_box[cell * 6 + c] = v0[c];
              F.cart_box[cell * 6 + 3 + c] = v7[c];
            }
          for (int v = 0; v < 8; ++v)
            for (int c = 0; c < 3; ++c)
              if (tria->vertex((int)cell, v)[c] != (((v >> c) & 1) ? v7[c] : v0[c]))
                bad[cell] = 1;
        });
        for (char b : bad)
          if (b)
            throw std::invalid_argument("flatten_cartesian: a cell is not an axis-aligned box (use flatten)");
        F.cart_vq_cell.resize(nqt / (size_t)npc);
        for (int l = 0; l < n_owned; ++l)
          {
            size_t g = (size_t)(F.vq_ptr[l] / npc);
            for (int cell : get_agglomerate(glob[l]))
              F.cart_vq_cell[g++] = cell;
          }
        F.cart.nq = nq;
        F.cart.nqf = nqf;
      }
    else
      {
        F.vq_w.resize(nqt);
        F.vq_x.resize(nqt * dim);
      }
    std::vector<double> x1, w1, xf1, wf1;
    qgauss_1d_compute(nq, x1, w1);
    qgauss_1d_compute(nqf, xf1, wf1);
    if (!cartesian)
    parallel_for((size_t)n_owned, [&](size_t l) {
      size_t at = (size_t)F.vq_ptr[l];
      for (int cell : get_agglomerate(glob[l])) // slaves in insertion order, then the master (:622-707)
        {
          cell_quadrature_soa(*tria, cell, nq, x1.data(), w1.data(), F.vq_x.data(), nqt, at, F.vq_w.data());
          at += (size_t)npc;
        }
    });

    // faces: every face with an owned side, described from its owner side I (normal, JxW_0, sigma as the owner sees them)
    struct Job
    {
      int I, O;                       // global polytopes: owner side, other side (-1 boundary)
      const std::vector<std::pair<int, int>> *li, *lo; // (cell, face) lists of the two sides, matching order
    };
    std::vector<Job> jobs;
    F.fq_ptr.assign(1, 0);
    for (int i = 0; i < n_owned; ++i)
      {
        const int P = glob[i];
        for (unsigned f = 0; f < n_faces(P); ++f)
          {
            if (at_boundary(P, f))
              {
                if (var.boundary == 1)
                  continue;
                const auto &li = get_interface(P, P);
                jobs.push_back({P, -1, &li, nullptr});
                F.face_in.push_back(i);
                F.face_out.push_back(-1);
                F.face_sigma.push_back(sigma(var, P, invalid_index));
                F.fq_ptr.push_back(F.fq_ptr.back() + npf * (int64_t)li.size());
              }
            else
              {
                const int Q = neighbor(P, f);
                const bool p_owns = owns(var, P, Q);
                if (!p_owns && owned(Q))
                  continue; // listed when Q's faces are walked
                const int I = p_owns ? P : Q, O = p_owns ? Q : P;
                const auto &li = get_interface(I, O), &lo = get_interface(O, I);
                if (li.size() != lo.size())
                  throw std::logic_error("interface lists of the two sides differ in length");
                jobs.push_back({I, O, &li, &lo});
                F.face_in.push_back(loc[I]);
                F.face_out.push_back(loc[O]);
                F.face_sigma.push_back(sigma(var, I, O));
                F.fq_ptr.push_back(F.fq_ptr.back() + npf * (int64_t)li.size());
              }
          }
      }
    F.c.n_faces = (int32_t)F.face_in.size();
    const size_t nft = (size_t)F.fq_ptr.back();
    if (cartesian)
      { // (cell, local face) of every sub-face as its owner side I sees it
        F.cart_fq_cell.resize(nft / (size_t)npf);
        F.cart_fq_face.resize(nft / (size_t)npf);
        parallel_for(jobs.size(), [&](size_t j) {
          size_t s = (size_t)(F.fq_ptr[j] / npf);
          for (const auto &cf : *jobs[j].li)
            {
              F.cart_fq_cell[s] = cf.first;
              F.cart_fq_face[s++] = cf.second;
            }
        });
      }
    else
      {
        F.fq_w.resize(nft);
        F.fq_w_out.resize(nft);
        F.fq_x.resize(nft * dim);
        F.fq_n.resize(nft * dim);
      }
    if (!cartesian)
    parallel_for(jobs.size(), [&](size_t j) {
      const Job &J = jobs[j];
      size_t at = (size_t)F.fq_ptr[j];
      for (size_t k = 0; k < J.li->size(); ++k)
        {
          face_quadrature_soa(*tria, (*J.li)[k].first, (*J.li)[k].second, nqf, xf1.data(), wf1.data(), F.fq_x.data(),
                              F.fq_n.data(), nft, at, F.fq_w.data());
          if (J.lo) // side 1: only its JxW is used (poly_utils.h:1906-1922)
            face_quadrature_soa(*tria, (*J.lo)[k].first, (*J.lo)[k].second, nqf, xf1.data(), wf1.data(), nullptr, nullptr, nft,
                                at, F.fq_w_out.data());
          else
            for (int64_t q = 0; q < npf; ++q)
              F.fq_w_out[at + q] = F.fq_w[at + q]; // unused on the boundary
          at += (size_t)npf;
        }
    });

    // column numbering that orders a row
    std::vector<int> colnum(nL);
    for (int l = 0; l < nL; ++l)
      colnum[l] = F.dof_offset[l];
    if (epetra_columns)
      {
        std::vector<int> gh(glob.begin() + n_owned, glob.end());
        std::sort(gh.begin(), gh.end(), [&](int a, int b) { return dof_offset[a] < dof_offset[b]; });
        for (int l = 0; l < n_owned; ++l)
          colnum[l] = dof_offset[glob[l]] - row_begin;
        for (size_t g = 0; g < gh.size(); ++g)
          colnum[loc[gh[g]]] = (row_end - row_begin) + (int)g * n;
        F.col_offset.assign(colnum.begin(), colnum.end());
      }
    // pattern of the owned rows (create_agglomeration_sparsity_pattern, :910-1022; restricted to locally_owned_dofs, :933-935)
    const int nrow = row_end - row_begin;
    F.rowptr.assign((size_t)nrow + 1, 0);
    std::vector<std::vector<int>> blocks(n_owned);
    for (int i = 0; i < n_owned; ++i)
      {
        const int P = glob[i];
        auto &b = blocks[i];
        b.push_back(i);
        for (int Q : face_nbr[P])
          if (Q != invalid_index)
            b.push_back(loc[Q]);
        std::sort(b.begin(), b.end(), [&](int x, int y) { return colnum[x] < colnum[y]; });
        for (int r = 0; r < n; ++r)
          F.rowptr[(size_t)(dof_offset[P] - row_begin) + r + 1] = (int64_t)b.size() * n;
      }
    for (int r = 0; r < nrow; ++r)
      F.rowptr[r + 1] += F.rowptr[r];
    if (with_colind)
      {
        F.colind.resize((size_t)F.rowptr[nrow]);
        parallel_for((size_t)n_owned, [&](size_t i) {
          for (int r = 0; r < n; ++r)
            {
              const int row = dof_offset[glob[i]] + r;
              int32_t *out = F.colind.data() + F.rowptr[row - row_begin];
              const int own_col = colnum[i] + r;
              if (diag_first)
                *out++ = own_col;
              for (int l : blocks[i])
                for (int j = 0; j < n; ++j)
                  if (!(diag_first && colnum[l] + j == own_col))
                    *out++ = colnum[l] + j;
            }
        });
      }
    if (row_splits)
      { // owning rank of every polytope: row_splits[r] <= dof_offset < row_splits[r+1]
        F.agg_rank.resize(nL);
        for (int l = 0; l < nL; ++l)
          {
            const auto it = std::upper_bound(row_splits->begin(), row_splits->end(), F.dof_offset[l]);
            F.agg_rank[l] = (int)(it - row_splits->begin()) - 1;
          }
      }
    if (local_of)
      *local_of = glob;
    F.bind();
  }

private:
  static uint64_t key(int P, int Q) { return ((uint64_t)(uint32_t)P << 32) | (uint32_t)Q; }

  // source/agglomeration_handler.cc:495-527 + 1253-1645 (serial branches): see SURVEY.md Appendix A.1
  void setup_connectivity_of_agglomeration()
  {
    const BackgroundGrid &g = *tria;
    const int nP = (int)master_cells.size(), nf = g.n_faces_per_cell();
    face_nbr.assign(nP, {});
    interface.clear();
    std::vector<uint8_t> visited((size_t)g.n_active_cells() * nf, 0); // GLOBAL visited_cell_and_faces
    std::vector<int> seen_stamp(nP + 1, -1);                         // per-master visited_polygonal_neighbors
    for (int P = 0; P < nP; ++P)
      {
        for (int cell : get_agglomerate(P))
          for (int f = 0; f < nf; ++f)
            {
              const int nb = g.neighbor(cell, f);
              if (nb != invalid_index)
                {
                  if (master_of[nb] == master_of[cell])
                    continue; // are_cells_agglomerated (:1300-1301)
                  const int Q = master2polygon.at(master_of[nb]);
                  const int nof = g.neighbor_of_neighbor(cell, f);
                  if (seen_stamp[Q] != P)
                    { // first contact with polytope Q: new polytopal face (:1346-1367 / 1412-1433)
                      face_nbr[P].push_back(Q);
                      seen_stamp[Q] = P;
                    }
                  if (!visited[(size_t)cell * nf + f])
                    {
                      interface[key(P, Q)].emplace_back(cell, f); // (:1370-1382)
                      visited[(size_t)cell * nf + f] = 1;
                    }
                  if (!visited[(size_t)nb * nf + nof])
                    {
                      interface[key(Q, P)].emplace_back(nb, nof); // (:1385-1397)
                      visited[(size_t)nb * nf + nof] = 1;
                    }
                }
              else
                { // all domain-boundary sub-faces form ONE polytopal face (:1575-1613)
                  if (seen_stamp[nP] != P)
                    {
                      face_nbr[P].push_back(invalid_index);
                      seen_stamp[nP] = P;
                    }
                  if (!visited[(size_t)cell * nf + f])
                    {
                      interface[key(P, P)].emplace_back(cell, f);
                      visited[(size_t)cell * nf + f] = 1;
                    }
                }
            }
      }
    connectivity_ready = true;
  }

  const BackgroundGrid *tria;
  FiniteElement fe;
  std::vector<int> master_of;     // master cell of every cell
  std::vector<int> master_cells;  // master_cells_container (polytope order)
  std::unordered_map<int, std::vector<int>> master2slaves;
  std::unordered_map<int, int> master2polygon;
  std::vector<std::array<double, 6>> bboxes;
  std::vector<int> dof_offset;
  unsigned int n_dofs_ = 0;
  std::vector<std::vector<int>> face_nbr; // per polytope: neighbour polytope per local face, -1 = boundary
  std::unordered_map<uint64_t, std::vector<std::pair<int, int>>> interface;
  int nq = 0, nqf = 0;
  bool connectivity_ready = false;
};

// Block agglomeration of a structured grid: b^dim cells per polytope, cells in mesh order inside a block
// (master = lowest index, as PolyUtils::collect_cells_for_agglomeration yields: include/poly_utils.h:532-538),
// blocks enumerated lexicographically.  Stand-in for METIS / R-tree levels on structured grids
// (on which the reference's R-tree levels are exactly such blocks: test/polydeal/rtree_mesh.output).
inline void define_block_agglomerates(AgglomerationHandler &ah, int b)
{
  const BackgroundGrid &g = ah.get_triangulation();
  const int dim = g.dim;
  if (g.unstructured)
    throw std::invalid_argument("block agglomerates need a lattice grid (use define_grown_agglomerates)");
  if (b <= 0 || g.n_dir[0] % b || g.n_dir[1] % b || (dim == 3 && g.n_dir[2] % b))
    throw std::invalid_argument("block size must divide the number of cells per direction");
  const int nbx = g.n_dir[0] / b, nby = g.n_dir[1] / b, nbz = dim == 3 ? g.n_dir[2] / b : 1;
  const int nblocks = nbx * nby * nbz;
  const int ncb = (dim == 2) ? b * b : b * b * b;
  std::vector<int> cells(ncb);
  for (int B = 0; B < nblocks; ++B)
    {
      const int bi[3] = {B % nbx, (B / nbx) % nby, B / (nbx * nby)};
      for (int l = 0; l < ncb; ++l)
        {
          const int o[3] = {l % b, (l / b) % b, l / (b * b)};
          cells[l] = g.cell_at(bi[0] * b + o[0], bi[1] * b + o[1], dim == 3 ? bi[2] * b + o[2] : 0);
        }
      std::sort(cells.begin(), cells.end());
      ah.define_agglomerate(cells);
    }
}

// Connected agglomerates of about `cells_per_polytope` cells each, grown over the cell connectivity graph - the stand-in for
// PolyUtils::partition_locally_owned_regions (METIS_PartGraphKway on that graph: reference examples/poisson.cc:543-566,
// include/poly_utils.h:465-540; METIS is not available offline).  What matters for the assembly path is the KIND of polytope
// METIS produces - connected, irregular, of similar size, meeting a neighbour along several sub-faces that need not be
// coplanar - not the particular partition.  Seeds are spread by a stride over the cells, every region then takes one free
// neighbouring cell per round (breadth first, deterministic for a given seed); cells no region reached (enclosed pockets)
// join the region of a neighbour.  Cells of a polytope are passed in ascending order, so the master is the lowest one
// (PolyUtils::collect_cells_for_agglomeration: include/poly_utils.h:532-538).
// n_subdomains connected agglomerates (GridTools::partition_triangulation(n_subdomains, tria, metis) + one agglomerate per
// subdomain: reference examples/minimal_SIP.cc:107-139)
inline void partition_into_grown_agglomerates(AgglomerationHandler &ah, int n_subdomains, unsigned seed = 0)
{
  const BackgroundGrid &g = ah.get_triangulation();
  const int nc = g.n_active_cells(), nf = g.n_faces_per_cell();
  if (n_subdomains < 1 || n_subdomains > nc)
    throw std::invalid_argument("n_subdomains must be between 1 and the number of cells");
  const int np = n_subdomains;
  std::mt19937_64 rng(seed);
  std::vector<int> owner((size_t)nc, -1);
  std::vector<std::vector<int>> frontier((size_t)np);
  // seeds: one random cell out of every stride of nc / np consecutive cells (consecutive cells are close in space for both
  // the Morton and the lexicographic numbering)
  for (int k = 0; k < np; ++k)
    {
      const int64_t b0 = (int64_t)nc * k / np, b1 = (int64_t)nc * (k + 1) / np;
      const int c = (int)(b0 + (int64_t)(rng() % (uint64_t)std::max<int64_t>(1, b1 - b0)));
      owner[c] = k;
      frontier[k].push_back(c);
    }
  int64_t remaining = nc - np;
  std::vector<size_t> head((size_t)np, 0);
  while (remaining > 0)
    {
      bool progressed = false;
      for (int k = 0; k < np && remaining > 0; ++k)
        {
          // next free neighbour of the oldest frontier cell that still has one
          std::vector<int> &fr = frontier[k];
          while (head[k] < fr.size())
            {
              const int c = fr[head[k]];
              int pick = -1;
              const int f0 = (int)(rng() % (uint64_t)nf);
              for (int df = 0; df < nf && pick < 0; ++df)
                {
                  const int nb = g.neighbor(c, (f0 + df) % nf);
                  if (nb != invalid_index && owner[nb] < 0)
                    pick = nb;
                }
              if (pick < 0)
                {
                  ++head[k];
                  continue;
                }
              owner[pick] = k;
              fr.push_back(pick);
              --remaining;
              progressed = true;
              break;
            }
        }
      if (!progressed)
        break;
    }
  // pockets no region could reach: give them to a neighbour's region
  for (bool again = remaining > 0; again;)
    {
      again = false;
      bool assigned = false;
      for (int c = 0; c < nc; ++c)
        if (owner[c] < 0)
          {
            for (int f = 0; f < nf && owner[c] < 0; ++f)
              {
                const int nb = g.neighbor(c, f);
                if (nb != invalid_index && owner[nb] >= 0)
                  owner[c] = owner[nb];
              }
            if (owner[c] < 0)
              again = true;
            else
              assigned = true;
          }
      if (again && !assigned) // a connected component of the mesh that received no seed: no sweep will ever reach it
        throw std::invalid_argument("partition_into_grown_agglomerates: the mesh has a connected component without a seed "
                                    "(more components than regions?)");
    }
  std::vector<std::vector<int>> groups((size_t)np);
  for (int c = 0; c < nc; ++c)
    groups[(size_t)owner[c]].push_back(c); // ascending
  for (const auto &cells : groups)
    if (!cells.empty())
      ah.define_agglomerate(cells);
}
inline void define_grown_agglomerates(AgglomerationHandler &ah, int cells_per_polytope, unsigned seed = 0)
{
  if (cells_per_polytope < 1)
    throw std::invalid_argument("cells_per_polytope must be >= 1");
  partition_into_grown_agglomerates(ah, std::max(1, ah.get_triangulation().n_active_cells() / cells_per_polytope), seed);
}

// ---------------------------------------------------------------------------------------------------
// Drop-in for PolyUtils::assemble_dg_matrix (include/poly_utils.h:2000-2195): fills `values` (CSR value
// array of the pattern create_agglomeration_sparsity_pattern produces) on the GPU.  Throws on error, like
// the reference's AssertThrow; never falls back to a CPU path.
// ---------------------------------------------------------------------------------------------------
namespace PolyUtilsHIP
{
inline void assemble_dg_matrix(std::vector<double> &values, const FiniteElement &fe_dg, const AgglomerationHandler &ah,
                               const SipVariant &variant = SipVariant::assemble_dg_matrix(), bool diag_first = true,
                               int device = 0)
{
  if (fe_dg.basis != ah.get_fe().basis || fe_dg.degree != ah.get_fe().degree)
    throw std::invalid_argument("FE passed to assemble_dg_matrix differs from the handler's");
  FlatProblem F;
  ah.flatten(variant, F, diag_first, false);
  pdh_ctx *ctx = nullptr;
  if (pdh_create(&ctx, device) != PDH_OK)
    throw std::runtime_error(std::string("pdh_create: ") + pdh_last_error(nullptr));
  values.assign((size_t)F.rowptr.back(), 0.0);
  const int rc = pdh_assemble_sip(ctx, &F.c, values.data());
  const std::string msg = rc == PDH_OK ? "" : pdh_last_error(ctx);
  pdh_destroy(ctx);
  if (rc != PDH_OK)
    throw std::runtime_error("pdh_assemble_sip: " + msg);
}

// PolyUtils::compute_global_error (reference include/poly_utils.h:1647-1750) on a problem resident in `ctx`
// (set from the same FlatProblem): u_h, grad u_h at the polytopes' quadrature points and the weighted sums come from the
// device (pdh_global_error).  exact(const double *x) -> double; exact_grad(const double *x, double *g).
// Returns {L2 error, H1-seminorm error}; pass want_h1 = false to skip the gradient part (second entry 0).
template <class Exact, class ExactGrad>
inline std::array<double, 2> compute_global_error(pdh_ctx *ctx, const FlatProblem &F, const std::vector<double> &solution,
                                                  Exact exact, ExactGrad exact_grad, bool want_h1 = true)
{
  const int dim = F.c.dim;
  const int64_t N = F.vq_ptr.back();
  // the analytical solution is sampled here (the C ABI has no callbacks); u_h, grad u_h and the JxW-weighted sums of the squared
  // differences are formed by ONE kernel on the device (pdh_global_error)
  std::vector<double> eu((size_t)N), eg((size_t)N * dim, 0.0);
  for (int64_t q = 0; q < N; ++q)
    {
      double x[3] = {0, 0, 0}, g[3] = {0, 0, 0};
      for (int c = 0; c < dim; ++c)
        x[c] = F.vq_x[(size_t)c * N + q];
      eu[q] = exact(x);
      if (want_h1)
        {
          exact_grad(x, g);
          for (int c = 0; c < dim; ++c)
            eg[(size_t)c * N + q] = g[c];
        }
    }
  double sums[2] = {0.0, 0.0};
  if (pdh_global_error(ctx, solution.data(), F.vq_ptr.data(), F.vq_x.data(), F.vq_w.data(), eu.data(), eg.data(), sums) != PDH_OK)
    throw std::runtime_error(std::string("pdh_global_error: ") + pdh_last_error(ctx));
  return {std::sqrt(sums[0]), want_h1 ? std::sqrt(sums[1]) : 0.0};
}
} // namespace PolyUtilsHIP

namespace Utils
{
// Utils::fill_injection_matrix (reference include/utils.h:95-270): the injection from the coarse polytopal
// space into the fine one, row block of fine polytope F / column block of its parent C:
//   local_matrix(i, j) = phi^C_j( coarse_bbox.real_to_unit( fine_bbox.unit_to_real(support point i) ) ).
// The reference takes polytope->children() from the R-tree hierarchy; here the children of C are the fine
// polytopes whose cells lie in C (the two handlers must be nested over the same grid, checked).  Needs support
// points, i.e. FE_DGQ (FE_AggloDGP has none: reference include/fe_agglodgp.h:63).  The basis evaluation runs on
// the device (pdh_shape_values); output is CSR over fine rows x coarse columns, n entries per row.
inline void fill_injection_matrix(const AgglomerationHandler &coarse_ah, const AgglomerationHandler &fine_ah,
                                  std::vector<int64_t> &rowptr, std::vector<int32_t> &colind, std::vector<double> &values,
                                  int device = 0)
{
  const FiniteElement &fe = coarse_ah.get_fe();
  if (fe.basis != PDH_BASIS_DGQ || fine_ah.get_fe().basis != fe.basis || fine_ah.get_fe().degree != fe.degree)
    throw std::invalid_argument("fill_injection_matrix needs the same FE_DGQ space on both handlers");
  if (&coarse_ah.get_triangulation() != &fine_ah.get_triangulation())
    throw std::invalid_argument("both handlers must live on the same grid");
  if (!(coarse_ah.n_dofs() < fine_ah.n_dofs()))
    throw std::invalid_argument("the coarse space must be smaller than the fine one"); // utils.h:120
  const int dim = fe.dim, n = fe.n_dofs_per_cell(), n1d = fe.degree + 1;
  const int nC = (int)coarse_ah.n_agglomerates(), nF = (int)fine_ah.n_agglomerates();
  // parent of every fine polytope; all its cells must share it
  std::vector<int> parent(nF);
  std::vector<std::vector<int>> children(nC);
  for (int F = 0; F < nF; ++F)
    {
      parent[F] = coarse_ah.cell_to_polytope_index(fine_ah.master_index(F));
      for (int cell : fine_ah.get_agglomerate(F))
        if (coarse_ah.cell_to_polytope_index(cell) != parent[F])
          throw std::invalid_argument("fine polytopes are not nested in the coarse ones");
      children[parent[F]].push_back(F);
    }
  // unit support points: tensor Gauss-Lobatto nodes, lexicographic [deal.II FE_DGQ]
  const auto nodes = pdh::gauss_lobatto_nodes(fe.degree);
  const auto mi = pdh::multi_indices(dim, fe.degree, fe.basis);
  // real support points of the children, grouped by parent (utils.h:212-217)
  std::vector<int64_t> pt_ptr(nC + 1, 0);
  for (int C = 0; C < nC; ++C)
    pt_ptr[C + 1] = pt_ptr[C] + (int64_t)children[C].size() * n;
  const int64_t N = pt_ptr[nC];
  std::vector<double> pts((size_t)N * dim), bb((size_t)nC * 2 * dim);
  std::vector<int> row_of_point((size_t)N);
  for (int C = 0; C < nC; ++C)
    {
      for (int c = 0; c < dim; ++c)
        {
          bb[(size_t)C * 2 * dim + c] = coarse_ah.bbox(C)[c];
          bb[(size_t)C * 2 * dim + dim + c] = coarse_ah.bbox(C)[3 + c];
        }
      int64_t q = pt_ptr[C];
      for (int F : children[C])
        for (int i = 0; i < n; ++i, ++q)
          {
            for (int c = 0; c < dim; ++c)
              {
                const double lo = fine_ah.bbox(F)[c], hi = fine_ah.bbox(F)[3 + c];
                pts[(size_t)c * N + q] = lo + (double)nodes[(mi[i] >> (8 * c)) & 0xff] * (hi - lo); // unit_to_real
              }
            row_of_point[q] = fine_ah.dof_offset_of(F) + i;
          }
    }
  (void)n1d;
  std::vector<double> local((size_t)N * n);
  pdh_ctx *ctx = nullptr;
  if (pdh_create(&ctx, device) != PDH_OK)
    throw std::runtime_error(std::string("pdh_create: ") + pdh_last_error(nullptr));
  const int rc = pdh_shape_values(ctx, dim, fe.degree, fe.basis, nC, bb.data(), pt_ptr.data(), pts.data(), local.data());
  const std::string msg = rc == PDH_OK ? "" : pdh_last_error(ctx);
  pdh_destroy(ctx);
  if (rc != PDH_OK)
    throw std::runtime_error("pdh_shape_values: " + msg);
  // the sparsity of utils.h:166-185: every fine row couples with the n dofs of its parent
  const int64_t rows = fine_ah.n_dofs();
  rowptr.resize(rows + 1);
  for (int64_t r = 0; r <= rows; ++r)
    rowptr[r] = r * n;
  colind.resize((size_t)rows * n);
  values.resize((size_t)rows * n);
  for (int C = 0; C < nC; ++C)
    for (int64_t q = pt_ptr[C]; q < pt_ptr[C + 1]; ++q)
      for (int j = 0; j < n; ++j)
        {
          colind[(size_t)row_of_point[q] * n + j] = coarse_ah.dof_offset_of(C) + j;
          values[(size_t)row_of_point[q] * n + j] = local[(size_t)q * n + j];
        }
}
} // namespace Utils
} // namespace polydeal_hip
