// pdh_cartgen.hip — quadrature data of agglomerates of Cartesian cells generated ON THE DEVICE (pdh_set_problem_cartesian).
//
// The reference gathers the points of a polytope on the host, sub-cell by sub-cell, inside the span it times
// (source/agglomeration_handler.cc:622-707: real points and JxW of Quadrature<dim> under the fine mapping; :1146-1165: points, JxW
// and the outward normal of every sub-face).  On axis-aligned boxes with QGauss rules both are closed forms of the cell's box:
//   volume group (nq^3 points, x fastest):   x_d = lo_d + h_d xi[i_d],   JxW = h_0 h_1 h_2 w[i_0] w[i_1] w[i_2]
//   face group (nqf^2 points of local face 2 c + side, lower tangential axis fastest):
//                                            x_c = lo_c or hi_c,  x_t = lo_t + h_t xi[.],  n = -+ e_c,  JxW = h_ti h_tj w[.] w[.]
// One thread per point, structure-of-arrays output in the layout the caller's arrays would have had: 0.9 GB for the bench mesh are
// written at HBM speed instead of being built by the host (0.2 s) and pushed through PCIe (30 ms).
#include "pdh_kernels.h"

struct PdhRule
{
  double x[PDH_MAX_N1D], w[PDH_MAX_N1D]; // Gauss-Legendre nodes / weights on [0,1]
};

__global__ void __launch_bounds__(256) k_gen_volume(const PdhRule rule, const int nq, const double *__restrict__ box,
                                                    const int32_t *__restrict__ gcell, const int64_t n_points, double *__restrict__ vq_x,
                                                    const int64_t stride, double *__restrict__ vq_w)
{
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_points)
    return;
  const int m3 = nq * nq * nq;
  const int64_t g = q / m3;
  const int l = (int)(q - g * m3);
  const int i0 = l % nq, i1 = (l / nq) % nq, i2 = l / (nq * nq);
  const double *b = box + (int64_t)gcell[g] * 6;
  const double h0 = b[3] - b[0], h1 = b[4] - b[1], h2 = b[5] - b[2];
  vq_x[q] = b[0] + h0 * rule.x[i0];
  vq_x[stride + q] = b[1] + h1 * rule.x[i1];
  vq_x[2 * stride + q] = b[2] + h2 * rule.x[i2];
  vq_w[q] = (h0 * h1 * h2) * (rule.w[i0] * rule.w[i1] * rule.w[i2]);
}

__global__ void __launch_bounds__(256) k_gen_faces(const PdhRule rule, const int nqf, const double *__restrict__ box,
                                                   const int32_t *__restrict__ fq_cell, const int32_t *__restrict__ fq_face,
                                                   const int64_t n_points, double *__restrict__ fq_x, double *__restrict__ fq_n,
                                                   double *__restrict__ fq_w)
{
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_points)
    return;
  const int m2 = nqf * nqf;
  const int64_t s = q / m2;
  const int l = (int)(q - s * m2);
  const int ia = l % nqf, ib = l / nqf;
  const int f = fq_face[s], c = f >> 1, side = f & 1;
  const int ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2;
  const double *b = box + (int64_t)fq_cell[s] * 6;
  const double hi_ = b[3 + ti] - b[ti], hj_ = b[3 + tj] - b[tj];
  double x[3], n[3] = {0.0, 0.0, 0.0};
  x[c] = side ? b[3 + c] : b[c];
  x[ti] = b[ti] + hi_ * rule.x[ia];
  x[tj] = b[tj] + hj_ * rule.x[ib];
  n[c] = side ? 1.0 : -1.0;
  for (int d = 0; d < 3; ++d)
    {
      fq_x[d * n_points + q] = x[d];
      fq_n[d * n_points + q] = n[d];
    }
  fq_w[q] = (hi_ * hj_) * (rule.w[ia] * rule.w[ib]);
}

extern "C" hipError_t pdh_launch_gen_volume(int nq, const double *nodes, const double *weights, const double *d_box, const int32_t *d_gcell,
                                            int64_t n_points, double *vq_x, int64_t stride, double *vq_w, hipStream_t stream)
{
  if (n_points <= 0)
    return hipSuccess;
  PdhRule r;
  for (int i = 0; i < PDH_MAX_N1D; ++i)
    r.x[i] = i < nq ? nodes[i] : 0.0, r.w[i] = i < nq ? weights[i] : 0.0;
  const int64_t blocks = (n_points + 255) / 256;
  hipLaunchKernelGGL(k_gen_volume, dim3((unsigned)blocks), dim3(256), 0, stream, r, nq, d_box, d_gcell, n_points, vq_x, stride, vq_w);
  return hipGetLastError();
}

extern "C" hipError_t pdh_launch_gen_faces(int nqf, const double *nodes, const double *weights, const double *d_box, const int32_t *d_cell,
                                           const int32_t *d_face, int64_t n_points, double *fq_x, double *fq_n, double *fq_w,
                                           hipStream_t stream)
{
  if (n_points <= 0)
    return hipSuccess;
  PdhRule r;
  for (int i = 0; i < PDH_MAX_N1D; ++i)
    r.x[i] = i < nqf ? nodes[i] : 0.0, r.w[i] = i < nqf ? weights[i] : 0.0;
  const int64_t blocks = (n_points + 255) / 256;
  hipLaunchKernelGGL(k_gen_faces, dim3((unsigned)blocks), dim3(256), 0, stream, r, nqf, d_box, d_cell, d_face, n_points, fq_x, fq_n, fq_w);
  return hipGetLastError();
}
