// pdh_exchange.hip — receive side of the ghost-block exchange variant (PDH_EXCHANGE_GHOST, include/polydeal_hip.h).
//
// The reference lets the rank that owns side 0 of a face cut by the partition assemble all four interface blocks and
// ships M21 / M22 to the owner of those rows inside compress(VectorOperation::add) (include/poly_utils.h:1930-1992,
// 2134-2194).  Here the sender's assembly kernels have written M21 (one plain n x n block per cut face, [row of Q][col of P])
// and the M22 sums (one block per remote polytope, in the diagonal-block layout of a row of length n) into its send region;
// after the transport this kernel puts them in place: M21 blocks are stored (nothing else contributes to A[Q,P]), the M22
// sums are added to A[Q,Q], all contributions of one polytope by one wave in a fixed order (bit-reproducible, no atomics).
#include "pdh_kernels.h"

__global__ void __launch_bounds__(PDH_WAVE) k_ghost_apply(const PdhDev P, const double *__restrict__ recv, const int n_r21,
                                                          const int64_t *__restrict__ r21_src, const int64_t *__restrict__ r21_dst,
                                                          const int32_t *__restrict__ r21_rlen, const int n_r22,
                                                          const int64_t *__restrict__ r22_ptr, const int64_t *__restrict__ r22_src,
                                                          const int32_t *__restrict__ r22_slot)
{
  const int lane = threadIdx.x, item = blockIdx.x, n = P.n;
  if (lane >= n)
    return;
  if (item < n_r21)
    {
      const double *src = recv + r21_src[item] + lane;
      double *dst = P.values + r21_dst[item] + lane;
      const int64_t rlen = r21_rlen[item];
      for (int R = 0; R < n; ++R)
        dst[R * rlen] = src[(int64_t)R * n];
    }
  else if (item < n_r21 + n_r22)
    {
      const int k = item - n_r21, slot = r22_slot[k];
      const int64_t rlen = P.row_len[slot];
      // position `lane` of a sender row = the same position of the own block here (the diagonal entry stays in front)
      const int dest = (P.diag_first && lane == 0) ? 0 : P.diag_L[slot] + lane;
      double *dst = P.values + P.row_base[slot] + dest;
      const int64_t sb = r22_ptr[k], se = r22_ptr[k + 1];
      for (int R = 0; R < n; ++R)
        {
          double acc = dst[R * rlen];
          for (int64_t s = sb; s < se; ++s)
            acc += recv[r22_src[s] + (int64_t)R * n + lane];
          dst[R * rlen] = acc;
        }
    }
}

extern "C" hipError_t pdh_launch_ghost_apply(const PdhDev *P, const double *recv, int n_r21, const int64_t *r21_src,
                                             const int64_t *r21_dst, const int32_t *r21_rlen, int n_r22, const int64_t *r22_ptr,
                                             const int64_t *r22_src, const int32_t *r22_slot, hipStream_t stream)
{
  const int count = n_r21 + n_r22;
  if (count <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_ghost_apply, dim3((unsigned)count), dim3(PDH_WAVE), 0, stream, *P, recv, n_r21, r21_src, r21_dst, r21_rlen,
                     n_r22, r22_ptr, r22_src, r22_slot);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Validity signal for matrices too large to copy back (pdh_values_checksum): sum, sum of |.|, max |.| and the number of
// non-finite entries of the owned rows' values, in one pass over HBM.  For FE_DGQ the sum of all entries is 1^T A 1, which
// the SIP form fixes in closed form (sigma |dOmega| + c |Omega| for Nitsche boundaries) - bench.py checks it every run.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_checksum(const double *__restrict__ v, const int64_t n, double *__restrict__ out)
{
  double s = 0.0, sa = 0.0, mx = 0.0, bad = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    {
      const double x = v[i];
      if (isfinite(x))
        {
          s += x;
          sa += fabs(x);
          mx = fmax(mx, fabs(x));
        }
      else
        bad += 1.0;
    }
  for (int o = 32; o > 0; o >>= 1)
    {
      s += __shfl_down(s, o);
      sa += __shfl_down(sa, o);
      bad += __shfl_down(bad, o);
      mx = fmax(mx, __shfl_down(mx, o));
    }
  if ((threadIdx.x & 63) == 0)
    {
      atomicAdd(out + 0, s);
      atomicAdd(out + 1, sa);
      atomicAdd(out + 3, bad);
      // max through the integer order of non-negative doubles
      atomicMax(reinterpret_cast<unsigned long long *>(out + 2), (unsigned long long)__double_as_longlong(mx));
    }
}

extern "C" hipError_t pdh_launch_checksum(const double *values, int64_t n, double *d_out4, hipStream_t stream)
{
  hipError_t e = hipMemsetAsync(d_out4, 0, 4 * sizeof(double), stream);
  if (e != hipSuccess || n <= 0)
    return e;
  hipLaunchKernelGGL(k_checksum, dim3(2048), dim3(256), 0, stream, values, n, d_out4);
  return hipGetLastError();
}
