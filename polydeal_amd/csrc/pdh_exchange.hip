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

// ---------------------------------------------------------------------------------------------------------------------
// Set-up: own-side face points of every (polytope, face) run, written in HBM from the caller's face arrays (each face
// stored once).  One wave per run.  Weights / signs as the assembly kernels expect them (pdh_capi.cpp: Packed::pk_*):
//   boundary run: w_self = 2 JxW, sigma / 2, w_cross = 0; interior: w_self = JxW of the own side, w_cross = JxW of side 1,
//   normal = outward normal of the owning polytope (sign flipped when it is side 1 of the face).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_faces(const int dim, const int64_t nqf, const double *__restrict__ fq_x,
                                                    const double *__restrict__ fq_n, const double *__restrict__ fq_w,
                                                    const double *__restrict__ fq_w_out, const int64_t n_runs,
                                                    const int64_t *__restrict__ pk_at, const int64_t *__restrict__ pk_fq,
                                                    const int32_t *__restrict__ pk_cnt, const int32_t *__restrict__ pk_flags,
                                                    const double *__restrict__ pk_sig, const int64_t nap, double *__restrict__ ap_x,
                                                    double *__restrict__ ap_n, double *__restrict__ ap_wself,
                                                    double *__restrict__ ap_wcross, double *__restrict__ ap_sig)
{
  const int lane = threadIdx.x & (PDH_WAVE - 1);
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x / PDH_WAVE) + (threadIdx.x / PDH_WAVE);
  if (r >= n_runs)
    return;
  const int64_t at = pk_at[r], fq = pk_fq[r];
  const int cnt = pk_cnt[r], fl = pk_flags[r];
  const bool side0 = (fl & 1) != 0, bdry = (fl & 2) != 0;
  const double sig = pk_sig[r], sgn = side0 ? 1.0 : -1.0;
  for (int t = lane; t < cnt; t += PDH_WAVE)
    {
      const int64_t q = fq + t, a = at + t;
      const double w_in = fq_w[q], w_out = fq_w_out ? fq_w_out[q] : w_in;
      ap_wself[a] = bdry ? 2.0 * w_in : (side0 ? w_in : w_out);
      ap_wcross[a] = bdry ? 0.0 : w_out;
      ap_sig[a] = sig;
      for (int c = 0; c < dim; ++c)
        {
          ap_x[c * nap + a] = fq_x[c * nqf + q];
          ap_n[c * nap + a] = sgn * fq_n[c * nqf + q];
        }
    }
}

extern "C" hipError_t pdh_launch_pack_faces(int dim, int64_t nqf, const double *fq_x, const double *fq_n, const double *fq_w,
                                            const double *fq_w_out, int64_t n_runs, const int64_t *pk_at, const int64_t *pk_fq,
                                            const int32_t *pk_cnt, const int32_t *pk_flags, const double *pk_sig, int64_t nap,
                                            double *ap_x, double *ap_n, double *ap_wself, double *ap_wcross, double *ap_sig,
                                            hipStream_t stream)
{
  if (n_runs <= 0)
    return hipSuccess;
  const int per_block = 256 / PDH_WAVE;
  const int64_t blocks = (n_runs + per_block - 1) / per_block;
  hipLaunchKernelGGL(k_pack_faces, dim3((unsigned)blocks), dim3(256), 0, stream, dim, nqf, fq_x, fq_n, fq_w, fq_w_out, n_runs, pk_at,
                     pk_fq, pk_cnt, pk_flags, pk_sig, nap, ap_x, ap_n, ap_wself, ap_wcross, ap_sig);
  return hipGetLastError();
}
