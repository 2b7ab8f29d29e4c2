// rows_store_probe.hip — what the STORES of the row kernel (pdh_rows.h) cost on their own: the same persistent structure
// (W single-wave workgroups per CU with L bytes of LDS each, polytopes handed out by a device-wide counter), the same store
// instructions (buffer_store_dwordx2, scalar row offset + 8 * lane, 64 rows of one 512-byte piece after another, 7 pieces per
// polytope, row length 7 * 64 doubles), no arithmetic.  Answers: is 1.2 ms (6.1 TB/s, store_pattern.hip: one wave per piece,
// 229 376 waves) also what 2048 persistent waves reach, and how does it move with the number of waves per CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

// spin: busy VALU work per polytope between the store bursts (0: none), in units of 64 dependent FMAs
// order 0: piece-major (all 64 rows of piece 0, then piece 1, ... - the row kernel's P5); 1: row-major (the nb pieces of row 0,
// then row 1, ...: 3584 contiguous bytes at a time)
// AUX: cache-policy bits of the store (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
// g_pad: extra doubles between the regions of consecutive polytopes (the CSR layout has none: 64 rows x 448 values = 7 x 64 KB);
// g_rot: polytope k starts its pieces with row (g_rot k) mod 64 instead of row 0
__constant__ int g_pad, g_rot;
template <int AUX>
__global__ void __launch_bounds__(64, 2) k_rows_like(double *v, unsigned *sched, int n_poly, int nb, int spin, int order)
{
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  int slot = blockIdx.x;
  if (lane == 0)
    lds[0] = 1.0;
  double acc = lane;
  for (;;)
    {
      int nslot_v = 0;
      if (lane == 0)
        nslot_v = (int)gridDim.x + (int)__hip_atomic_fetch_add(sched, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int rlen = nb * 64;
      for (int s = 0; s < spin; ++s)
#pragma unroll
        for (int k = 0; k < 64; ++k)
          acc = acc * 1.0000001 + 1e-9;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(v + (size_t)slot * (64 * rlen + g_pad), 0, 64 * rlen * 8, 0x00020000);
      if (order == 3)
        { // piece-major, rows rotated by the polytope number
          const unsigned r0 = (unsigned)(g_rot * slot) & 63u;
          for (int b = 0; b < nb; ++b)
#pragma unroll 8
            for (int r = 0; r < 64; ++r)
              {
                const unsigned rr = (r0 + (unsigned)r) & 63u;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, acc + r), rs, lane * 8, 64u * 8u * (unsigned)b + rr * (unsigned)rlen * 8u, AUX);
              }
        }
      else if (order == 2)
        { // dwordx4: a store instruction writes 2 x 512 bytes - lanes 0..31 the piece of row r (16 bytes each), lanes 32..63 of row r + 32
          typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
          const unsigned lo = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * 32u * (unsigned)rlen * 8u;
          for (int b = 0; b < nb; ++b)
            {
              unsigned off = 64u * 8u * (unsigned)b;
#pragma unroll
              for (int r = 0; r < 32; ++r)
                {
                  u4_t d;
                  const u2_t a = __builtin_bit_cast(u2_t, acc + r), c = __builtin_bit_cast(u2_t, acc - r);
                  d.x = a.x, d.y = a.y, d.z = c.x, d.w = c.y;
                  __builtin_amdgcn_raw_buffer_store_b128(d, rs, lo, off, AUX);
                  off += (unsigned)rlen * 8u;
                }
            }
        }
      else if (order == 0)
        for (int b = 0; b < nb; ++b)
          {
            unsigned off = 64u * 8u * (unsigned)b;
#pragma unroll
            for (int r = 0; r < 64; ++r)
              {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, acc + r), rs, lane * 8, off, AUX);
                off += (unsigned)rlen * 8u;
              }
          }
      else
        {
          unsigned off = 0;
          for (int r = 0; r < 64; ++r)
#pragma unroll
            for (int b = 0; b < 7; ++b)
              {
                if (b < nb)
                  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, acc + r), rs, lane * 8, off, AUX);
                off += 512u;
              }
        }
      const int nslot = __builtin_amdgcn_readfirstlane(nslot_v);
      if (nslot >= n_poly)
        break;
      slot = nslot;
    }
  if (lane == 0)
    if (__hip_atomic_fetch_add(sched + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
      {
        __hip_atomic_store(sched, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sched + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
}

// SHARE waves of one workgroup write ONE polytope together (wave w the rows [64 w / SHARE, 64 (w + 1) / SHARE), piece after piece):
// the same number of waves per CU, 1 / SHARE as many concurrent store streams (regions written at a time).
// (APART = true: the control - the same workgroups and barriers, but every wave writes a polytope of its own, all 64 rows)
template <int AUX, int SHARE, bool APART = false>
__global__ void __launch_bounds__(64 * SHARE) k_rows_shared(double *v, unsigned *sched, int n_poly, int nb, int spin)
{
  extern __shared__ double lds[];
  __shared__ int next_slot;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int slot = blockIdx.x;
  double acc = lane;
  for (;;)
    {
      if (threadIdx.x == 0)
        next_slot = (int)gridDim.x + (int)__hip_atomic_fetch_add(sched, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int rlen = nb * 64;
      for (int s = 0; s < spin; ++s)
#pragma unroll
        for (int k = 0; k < 64; ++k)
          acc = acc * 1.0000001 + 1e-9;
      // (APART: slot counts groups of SHARE polytopes)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(v + (size_t)(APART ? slot * SHARE + w : slot) * 64 * rlen, 0, 64 * rlen * 8, 0x00020000);
      constexpr int RW = APART ? 64 : 64 / SHARE;
      for (int b = 0; b < nb; ++b)
        {
          unsigned off = 64u * 8u * (unsigned)b + (APART ? 0u : (unsigned)(w * RW) * (unsigned)rlen * 8u);
#pragma unroll
          for (int r = 0; r < RW; ++r)
            {
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, acc + r), rs, lane * 8, off, AUX);
              off += (unsigned)rlen * 8u;
            }
        }
      __syncthreads();
      const int nslot = next_slot;
      __syncthreads();
      if (nslot >= n_poly)
        break;
      slot = nslot;
    }
  if (threadIdx.x == 0)
    if (__hip_atomic_fetch_add(sched + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
      {
        __hip_atomic_store(sched, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sched + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
}

// The store pattern a workgroup-of-eight row kernel would have: a workgroup takes EIGHT polytopes at a time (wave w computes
// polytope 8 s + w); the six coupling blocks are written in six phases - in phase b every wave writes its eight rows
// (8 w .. 8 w + 7) of piece b of all eight polytopes, barrier - and each wave writes the 64 rows of the own block (piece 6 here)
// of its polytope alone.
template <int AUX>
__global__ void __launch_bounds__(512) k_rows_coop(double *v, unsigned *sched, int n_groups, int nb, int spin)
{
  extern __shared__ double lds[];
  __shared__ int next_slot;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int slot = blockIdx.x;
  double acc = lane;
  for (;;)
    {
      if (threadIdx.x == 0)
        next_slot = (int)gridDim.x + (int)__hip_atomic_fetch_add(sched, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int rlen = nb * 64;
      for (int s = 0; s < spin; ++s)
#pragma unroll
        for (int k = 0; k < 64; ++k)
          acc = acc * 1.0000001 + 1e-9;
      for (int b = 0; b < nb - 1; ++b)
        {
          for (int p = 0; p < 8; ++p)
            {
              const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(v + (size_t)(slot * 8 + p) * 64 * rlen, 0, 64 * rlen * 8, 0x00020000);
              unsigned off = 64u * 8u * (unsigned)b + (unsigned)(w * 8) * (unsigned)rlen * 8u;
#pragma unroll
              for (int r = 0; r < 8; ++r)
                {
                  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, acc + r), rs, lane * 8, off, AUX);
                  off += (unsigned)rlen * 8u;
                }
            }
          __syncthreads();
        }
      {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(v + (size_t)(slot * 8 + w) * 64 * rlen, 0, 64 * rlen * 8, 0x00020000);
        unsigned off = 64u * 8u * (unsigned)(nb - 1);
#pragma unroll
        for (int r = 0; r < 64; ++r)
          {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, acc + r), rs, lane * 8, off, AUX);
            off += (unsigned)rlen * 8u;
          }
      }
      __syncthreads();
      const int nslot = next_slot;
      __syncthreads();
      if (nslot >= n_groups)
        break;
      slot = nslot;
    }
  if (threadIdx.x == 0)
    if (__hip_atomic_fetch_add(sched + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
      {
        __hip_atomic_store(sched, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sched + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
}

int main()
{
  const int NP = 32768, NB = 7;
  const size_t n = (size_t)NP * 64 * NB * 64;
  double *v;
  unsigned *sched;
  // ROWS_PROBE_ALLOC=uncached | finegrained: the values in device memory of another kind (hipExtMallocWithFlags)
  const char *kind = getenv("ROWS_PROBE_ALLOC");
  if (kind && kind[0] == 'u')
    CHECK(hipExtMallocWithFlags((void **)&v, (n + 64) * sizeof(double), hipDeviceMallocUncached));
  else if (kind && kind[0] == 'f')
    CHECK(hipExtMallocWithFlags((void **)&v, (n + 64) * sizeof(double), hipDeviceMallocFinegrained));
  else
    CHECK(hipMalloc(&v, (n + 64 + (size_t)NP * 8300) * sizeof(double)));
  std::printf("allocation: %s\n", kind ? kind : "hipMalloc");
  CHECK(hipMalloc(&sched, 64));
  auto set_pad_rot = [&](int pad, int rot) {
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_pad), &pad, sizeof(int)));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_rot), &rot, sizeof(int)));
  };
  set_pad_rot(0, 0);
  CHECK(hipMemset(sched, 0, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](auto aux_, int order, int spin, int per_cu) {
    constexpr int AUX = decltype(aux_)::value;
    const size_t lds = per_cu <= 8 ? 20 * 1024 : (160 * 1024 / per_cu) / 1280 * 1280;
    const int grid = 256 * per_cu;
    auto launch = [&] { hipLaunchKernelGGL(k_rows_like<AUX>, dim3(grid), dim3(64), lds, 0, v, sched, NP, NB, spin, order); };
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
    std::printf("aux %2d  %s  spin %2d (x64 dependent FMA per polytope)  %2d waves/CU (LDS %6zu B): %.3f ms  %.2f TB/s\n", AUX,
                order == 3 ? "piece-major, rows rotated" : (order == 2 ? "piece-major, dwordx4 (2 rows)" : (order ? "row-major  " : "piece-major")), spin, per_cu, lds, ms, n * 8.0 / ms / 1e9);
  };
  auto run_apart = [&](auto share_, int spin, int waves_per_cu) {
    constexpr int SHARE = decltype(share_)::value;
    const int wg_per_cu = waves_per_cu / SHARE;
    const size_t lds = 64 * 1024 - 64;
    const int grid = 256 * wg_per_cu;
    auto launch = [&] { hipLaunchKernelGGL((k_rows_shared<18, SHARE, true>), dim3(grid), dim3(64 * SHARE), lds, 0, v, sched, NP / SHARE, NB, spin); };
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
    std::printf("aux 18  control: workgroups of %d waves in step, every wave a polytope of its own  %2d waves/CU: %.3f ms  %.2f TB/s\n", SHARE,
                waves_per_cu, ms, n * 8.0 / ms / 1e9);
  };
  auto run_shared = [&](auto share_, int spin, int waves_per_cu) {
    constexpr int SHARE = decltype(share_)::value;
    const int wg_per_cu = waves_per_cu / SHARE;
    const size_t lds = (size_t)20 * 1024 * SHARE <= 64 * 1024 ? (size_t)20 * 1024 * SHARE : 64 * 1024 - 64; // (<= 64 KB per workgroup)
    const int grid = 256 * wg_per_cu;
    auto launch = [&] { hipLaunchKernelGGL((k_rows_shared<18, SHARE>), dim3(grid), dim3(64 * SHARE), lds, 0, v, sched, NP, NB, spin); };
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
    std::printf("aux 18  %d waves share a polytope  spin %2d  %2d waves/CU = %d workgroups/CU = %d concurrent regions: %.3f ms  %.2f TB/s\n", SHARE,
                spin, waves_per_cu, wg_per_cu, grid, ms, n * 8.0 / ms / 1e9);
  };
  for (int per_cu : {8, 16})
    {
      run_shared(std::integral_constant<int, 1>{}, 16, per_cu);
      run_shared(std::integral_constant<int, 2>{}, 16, per_cu);
      run_shared(std::integral_constant<int, 4>{}, 16, per_cu);
      if (per_cu >= 8)
        run_shared(std::integral_constant<int, 8>{}, 16, per_cu);
    }
  for (int spin : {0, 16, 64})
    {
      auto launch = [&] { hipLaunchKernelGGL((k_rows_coop<18>), dim3(256), dim3(512), 160 * 1024 - 1024, 0, v, sched, NP / 8, NB, spin); };
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
      std::printf("aux 18  workgroup of eight: 8 polytopes at a time, coupling pieces by 8 rows per wave, own piece per wave  spin %2d: %.3f ms  %.2f TB/s\n", spin, ms,
                  n * 8.0 / ms / 1e9);
    }
  for (int pad : {0, 32, 512, 520, 8192 + 32})
    {
      set_pad_rot(pad, 0);
      std::printf("pad %d doubles between polytopes: ", pad);
      run(std::integral_constant<int, 18>{}, 0, 16, 8);
    }
  set_pad_rot(0, 0);
  for (int rot : {0, 1, 8, 9, 23})
    {
      set_pad_rot(0, rot);
      std::printf("rows rotated by %d x polytope number: ", rot);
      run(std::integral_constant<int, 18>{}, 3, 16, 8);
    }
  set_pad_rot(0, 0);
  run_apart(std::integral_constant<int, 8>{}, 16, 8);
  run_apart(std::integral_constant<int, 4>{}, 16, 8);
  for (int order : {0})
    for (int per_cu : {2, 8})
      {
        run(std::integral_constant<int, 0>{}, order, 16, per_cu);
        run(std::integral_constant<int, 18>{}, order, 16, per_cu);
      }
  return 0;
}
