// store_pattern.hip — write-bandwidth probe for the CSR row-segment store pattern of the coupling blocks.
// A matrix of NP polytopes x 64 rows x NB blocks x 64 doubles (diag-first style rows of NB*512 B) is written by
// one wave per (polytope, block): 64 row segments of 512 B at a stride of NB*512 B - the pattern of k_offdiag -
// or, for comparison, by one wave per 64 KB contiguous range.  Prints TB/s for both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_scatter(double *v, int nb, int shift, int rows_per_store)
{
  const int item = blockIdx.x; // (polytope, block)
  const int P = item / nb, blk = item % nb;
  const int lane = threadIdx.x;
  double *dst = v + ((size_t)P * 64) * nb * 64 + (size_t)blk * 64 + shift + lane;
  for (int r = 0; r < 64; ++r)
    dst[(size_t)r * nb * 64] = (double)(item + r);
}
// same pattern as k_scatter(+8B) but the nb waves of one polytope are 8 block ids apart: same XCD, close in time
__global__ void __launch_bounds__(64) k_scatter_xcd(double *v, int nb, int shift)
{
  const int b = blockIdx.x;
  // groups of 8*nb consecutive block ids hold 8 polytopes; inside a group, id = blk*8 + p8
  const int g = b / (8 * nb), w = b % (8 * nb);
  const int P = g * 8 + (w & 7), blk = w >> 3;
  const int lane = threadIdx.x;
  double *dst = v + ((size_t)P * 64) * nb * 64 + (size_t)blk * 64 + shift + lane;
  for (int r = 0; r < 64; ++r)
    dst[(size_t)r * nb * 64] = (double)(b + r);
}
// mode 0: nontemporal misaligned; 1: only the full-line middle (elements 15..62); 2: only head+tail (0..14, 63)
__global__ void __launch_bounds__(64) k_scatter_mode(double *v, int nb, int mode)
{
  const int item = blockIdx.x;
  const int P = item / nb, blk = item % nb;
  const int lane = threadIdx.x;
  double *dst = v + ((size_t)P * 64) * nb * 64 + (size_t)blk * 64 + 1 + lane;
  for (int r = 0; r < 64; ++r)
    {
      double *d = dst + (size_t)r * nb * 64;
      if (mode == 0)
        __builtin_nontemporal_store((double)(item + r), d);
      else if (mode == 1)
        {
          if (lane >= 15 && lane < 63)
            *d = (double)(item + r);
        }
      else
        {
          if (lane < 15 || lane == 63)
            *d = (double)(item + r);
        }
    }
}
__global__ void __launch_bounds__(64) k_contig(double *v, int nb)
{
  const int item = blockIdx.x;
  const int lane = threadIdx.x;
  double *dst = v + (size_t)item * 4096 + lane;
  for (int r = 0; r < 64; ++r)
    dst[(size_t)r * 64] = (double)(item + r);
}
// one workgroup of nb waves per polytope: wave w writes block w of every row -> whole rows written together
__global__ void __launch_bounds__(448) k_rowgroup(double *v, int nb, int shift)
{
  const int P = blockIdx.x, blk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *dst = v + ((size_t)P * 64) * nb * 64 + (size_t)blk * 64 + shift + lane;
  for (int r = 0; r < 64; ++r)
    dst[(size_t)r * nb * 64] = (double)(P + r);
}
int main()
{
  const int NP = 32768, NB = 7;
  const size_t n = (size_t)NP * 64 * NB * 64;
  double *v;
  CHECK(hipMalloc(&v, (n + 64) * sizeof(double)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto time = [&](const char *name, auto launch) {
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i)
      launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    std::printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, n * 8.0 / ms / 1e9);
  };
  time("scatter 512B rows (aligned)", [&] { hipLaunchKernelGGL(k_scatter, dim3(NP * NB), dim3(64), 0, 0, v, NB, 0, 1); });
  time("scatter 512B rows (+8B)", [&] { hipLaunchKernelGGL(k_scatter, dim3(NP * NB), dim3(64), 0, 0, v, NB, 1, 1); });
  time("+8B nontemporal", [&] { hipLaunchKernelGGL(k_scatter_mode, dim3(NP * NB), dim3(64), 0, 0, v, NB, 0); });
  time("+8B middle 48 only (x64/48)", [&] { hipLaunchKernelGGL(k_scatter_mode, dim3(NP * NB), dim3(64), 0, 0, v, NB, 1); });
  time("+8B head/tail 16 only", [&] { hipLaunchKernelGGL(k_scatter_mode, dim3(NP * NB), dim3(64), 0, 0, v, NB, 2); });
  time("contiguous 64KB per wave", [&] { hipLaunchKernelGGL(k_contig, dim3(NP * NB), dim3(64), 0, 0, v, NB); });
  time("row group (7 waves / polytope)", [&] { hipLaunchKernelGGL(k_rowgroup, dim3(NP), dim3(448), 0, 0, v, NB, 0); });
  time("row group +8B", [&] { hipLaunchKernelGGL(k_rowgroup, dim3(NP), dim3(448), 0, 0, v, NB, 1); });
  time("scatter +8B, same-XCD ordering", [&] { hipLaunchKernelGGL(k_scatter_xcd, dim3(NP * NB), dim3(64), 0, 0, v, NB, 1); });
  CHECK(hipMemset(v, 0, n * 8));
  return 0;
}
