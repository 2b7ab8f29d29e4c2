// pdh_capi.cpp — implementation of the C ABI declared in include/polydeal_hip.h.
//
// Host work done here is SETUP only (what the reference does once per mesh in
// AgglomerationHandler::distribute_agglomerated_dofs / setup_connectivity_of_agglomeration /
// create_agglomeration_sparsity_pattern, source/agglomeration_handler.cc:326-379, 495-527, 910-1022):
// validation, repacking of the face tables per owning polytope, block positions inside CSR rows, upload.
// The assembly itself (pdh_assemble_device) only launches the two HIP kernels of pdh_kernels.h.
#include "../../include/polydeal_hip.h"
#include "pdh_basis.h"
#include "pdh_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "pdh_combos.h"

#include <cstdlib>
#include <thread>

// setup-time loops over all quadrature points run on all host threads (PDH_HOST_THREADS overrides the count)
template <class F>
static void host_parallel_for(size_t n, F &&fn)
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
  const size_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t)
    {
      const size_t b = (size_t)t * chunk, e = std::min(n, b + chunk);
      if (b >= e)
        break;
      th.emplace_back([&fn, b, e] {
        for (size_t i = b; i < e; ++i)
          fn(i);
      });
    }
  for (auto &t : th)
    t.join();
}

extern "C" {
typedef hipError_t (*pdh_launch_fn)(int dim, int n1d, int nt, int lb, int which, const PdhDev *P, int count, size_t lds,
                                    hipStream_t stream);
#define PDH_DECL(g) hipError_t pdh_launch_g##g(int, int, int, int, int, const PdhDev *, int, size_t, hipStream_t);
PDH_DECL(0) PDH_DECL(1) PDH_DECL(2) PDH_DECL(3) PDH_DECL(4) PDH_DECL(5) PDH_DECL(6) PDH_DECL(7)
#undef PDH_DECL
}

extern "C" hipError_t pdh_launch_rhs(int dim, int n1d, const PdhDev *P, int count, const double *f_vol,
                                     const double *g_face, double *rhs, const int64_t *vq_src, const int64_t *ap_src,
                                     const int64_t *bd_rng, hipStream_t stream);

extern "C" hipError_t pdh_launch_eval(int dim, int n1d, int grad, const PdhDev *P, int count, const double *coef,
                                      const int64_t *pt_ptr, const double *pts, int64_t pts_stride, double *out_u,
                                      double *out_g, int by_agg, hipStream_t stream);

extern "C" hipError_t pdh_launch_shape(int dim, int n1d, const PdhDev *P, int n_boxes, const int64_t *pt_ptr,
                                       const double *pts, int64_t pts_stride, double *out, hipStream_t stream);

extern "C" hipError_t pdh_launch_moment(int n1d, int which, const PdhDev *P, const double *mtab, int count, hipStream_t stream);
#include "pdh_rows_tables.h"
#include "pdh_terms_tables.h"
extern "C" hipError_t pdh_launch_rows(const PdhDev *P, const PdhRows *R, const double *mtab, int count, hipStream_t stream);
extern "C" int pdh_rows_n_dofs(int n1d, int basis);
extern "C" hipError_t pdh_launch_terms(const PdhDev *P, const PdhTerms *T, int count, hipStream_t stream);
extern "C" int pdh_terms_has_kind(int n1d, int basis);
extern "C" hipError_t pdh_launch_tiled(int dim, int n1d, int which, const PdhDev *P, int count, hipStream_t stream);
extern "C" int pdh_tiled_has_kind(int dim, int n1d, int n);
extern "C" hipError_t pdh_launch_gen_volume(int nq, const double *nodes, const double *weights, const double *d_box, const int32_t *d_gcell,
                                            int64_t n_points, double *vq_x, int64_t stride, double *vq_w, hipStream_t stream);
extern "C" hipError_t pdh_launch_gen_faces(int nqf, const double *nodes, const double *weights, const double *d_box, const int32_t *d_cell,
                                           const int32_t *d_face, int64_t n_points, double *fq_x, double *fq_n, double *fq_w,
                                           hipStream_t stream);
extern "C" hipError_t pdh_launch_terms_gather(const PdhDev *P, const PdhTerms *T, double *out, int count, hipStream_t stream);
extern "C" int pdh_terms_task_doubles(int maxsf, int maxcell, int pm);
extern "C" int pdh_terms_lds_bytes(int n1d, int basis, int maxruns, int maxsf, int maxsi, int maxcell, int split, int task_pts);
extern "C" int pdh_rows_max_faces(void);
extern "C" int pdh_moment_table_doubles(int n1d);

static pdh_launch_fn g_launch[PDH_N_GROUPS] = {pdh_launch_g0, pdh_launch_g1, pdh_launch_g2, pdh_launch_g3,
                                               pdh_launch_g4, pdh_launch_g5, pdh_launch_g6, pdh_launch_g7};

// number of MFMA instructions one 4-point step of a product issues, from the kernels' own schedule
template <int NT, int LB>
static constexpr int sched_instr(bool sym)
{
  int c = 0;
  for (int a = 0; a < NT; ++a)
    for (int b = 0; b < NT; ++b)
      for (int r = 0; r < 4; ++r)
        if ((sym ? pdh::Sched<NT, LB>::sym_mask(a, b, r) : pdh::Sched<NT, LB>::full_mask(a, b, r)) != 0u)
          ++c;
  return c;
}
static int sched_instr_rt(int nt, int lb, bool sym)
{
  switch (nt * 4 + (lb - 1))
    {
#define PDH_C(NT, LB)                                                                              \
  case NT * 4 + (LB - 1):                                                                          \
    return sched_instr<NT, LB>(sym);
      PDH_C(1, 1) PDH_C(1, 2) PDH_C(1, 3) PDH_C(1, 4) PDH_C(2, 1) PDH_C(2, 2) PDH_C(2, 3) PDH_C(2, 4)
      PDH_C(3, 1) PDH_C(3, 2) PDH_C(3, 3) PDH_C(3, 4) PDH_C(4, 1) PDH_C(4, 2) PDH_C(4, 3) PDH_C(4, 4)
#undef PDH_C
    }
  return 0;
}

// translation-unit group holding the kernels of a combo, or -1 if that combo is not instantiated
static int combo_group(int dim, int n1d, int nt, int lb)
{
#define PDH_X(G, D, N, T, L)                                                                       \
  if (dim == D && n1d == N && nt == T && lb == L)                                                  \
    return G;
  PDH_COMBOS(PDH_X)
#undef PDH_X
  return -1;
}

struct pdh_ctx
{
  int device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr; // stream = the one in use (own_stream unless pdh_set_stream)
  std::string err;
  bool has_problem = false;
  std::vector<void *> allocs;
  PdhDev dev;
  int n_owned = 0, n_items = 0, NT = 0, LB = 0, group = -1;
  bool tiled = false; // n > 64 dofs per polytope: pdh_tiled.h instead of the kernels of `group`
  int64_t terms_merge[4] = {0, 0, 0, 0}; // term kernels: cells before / after merging, sub-faces before / after
  size_t lds_diag = 0, lds_off = 0;
  int64_t n_values = 0, n_vq = 0, n_ap = 0;
  // host-side maps from the caller's quadrature arrays to the packed device layout (for pdh_assemble_rhs)
  std::vector<int64_t> vq_src;                       // per owned slot: first volume point in the caller's arrays
  struct FaceRun { int64_t ap_begin, fq_begin; int32_t count; int32_t boundary; int32_t slot; };
  std::vector<FaceRun> face_runs;
  int64_t n_rows_owned = 0;
  int32_t n_agg_total = 0;
  // caller-order maps for the right-hand side (device): first caller volume point of every slot; caller face point of every
  // packed face point (-1: not on the boundary); sizes of the caller's point arrays
  const int64_t *d_vq_src = nullptr, *d_ap_src = nullptr;
  const int64_t *d_bd_rng = nullptr; // [n_owned][2] packed face points of every slot that lie on the boundary (one run)
  int64_t n_vq_caller = 0, n_fq_caller = 0;
  // grow-only device scratch for the host-pointer variants of rhs / evaluate / shape_values (no hipMalloc per call)
  struct Scratch { void *p = nullptr; size_t bytes = 0; };
  Scratch scratch[6];
  void *scratch_get(int i, size_t bytes)
  {
    Scratch &s = scratch[i];
    if (bytes > s.bytes)
      {
        if (s.p)
          (void)hipFree(s.p);
        s.p = nullptr;
        s.bytes = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&s.p, want) != hipSuccess)
          return nullptr;
        s.bytes = want;
      }
    return s.p;
  }
  // cached multi-index table of pdh_shape_values (per dim/degree/basis)
  int shape_key = -1;
  int32_t *d_shape_midx = nullptr;
  int64_t mfma_diag = 0, mfma_offdiag = 0; // MFMA instructions per launch
  // ghost-block exchange variant (pdh_set_exchange_mode); n_diag_slots = n_owned + pseudo slots of the outgoing M22 sums
  int exchange_mode = PDH_EXCHANGE_NONE;
  bool problem_ghost = false;
  int n_diag_slots = 0, n_r21 = 0, n_r22 = 0;
  int64_t n_send = 0, n_recv = 0;
  std::vector<int64_t> send_count, recv_count;
  const int64_t *d_r21_src = nullptr, *d_r21_dst = nullptr, *d_r22_ptr = nullptr, *d_r22_src = nullptr;
  const int32_t *d_r21_rlen = nullptr, *d_r22_slot = nullptr;
  // moment form (pdh_moment.h): available for 3-D bases of degree <= 3; `algorithm` = caller's choice
  int algorithm = PDH_ALG_AUTO;
  int basis = 0;
  double *d_mtab = nullptr;
  // row kernel (pdh_rows.h): available when every face of every owned polytope is an axis-aligned plane (FE_DGQ(3), 3-D)
  bool rows_ok = false;
  bool rows_auto = true; // AUTO takes the row kernel where it applies (degree 1 since 12 waves per CU are resident: 0.21 vs 0.24-0.30 ms)
  PdhRows rows;
  // term kernel (pdh_terms.h): the row kernel of the small elements on agglomerates of Cartesian cells with tensor rules - any
  // number of planes per neighbour; taken instead of the streamed kinds of pdh_rows.h wherever its tables fit the LDS budget
  bool terms_ok = false;
  PdhTerms terms;
  bool use_rows() const
  {
    return (rows_ok || terms_ok) && d_mtab && ((algorithm == PDH_ALG_AUTO && rows_auto) || algorithm == PDH_ALG_ROWS);
  }
  // which form each of the two launches uses: [0] diagonal blocks, [1] coupling blocks
  bool use_moment(int kind) const
  {
    if (!d_mtab || algorithm == PDH_ALG_DIRECT)
      return false;
    if (algorithm == PDH_ALG_MOMENT)
      return true;
    if (algorithm == PDH_ALG_ROWS)
      return false;
    // auto: where the moment form was measured faster than the MFMA contraction (profiles/README.md): FE_DGQ(3) both
    // kinds (8.5 -> 4.7 ms), FE_DGQ(2) the diagonal blocks only (BASELINE configs[3]: 9.7 -> 5.6 ms; its coupling blocks
    // 4.3 ms direct vs 6.4 ms moment)
    if (basis != PDH_BASIS_DGQ)
      return false;
    return dev.n1d == 4 || (dev.n1d == 3 && kind == 0);
  }
  // The two kernels of a step write disjoint values and have complementary bottlenecks (the diagonal items compute, the
  // coupling items mostly store): on large problems they run concurrently, the coupling kernel on stream2, forked from /
  // joined into `stream` by events so that the caller still sees one ordered stream.  Measured -4 % per step.
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int overlap = 1; // pdh_set_overlap
  // (two streams pay for their fork / join events only when the kernels run for a while: by the size of the matrix)
  static constexpr int64_t small_values = 16 << 20;
  bool overlapped() const { return overlap && stream2 && (int64_t)n_diag_slots + n_items >= 8192 && n_values >= small_values; }
  // Small problems are bound by the launches themselves (two kernels of a few microseconds each): the pair is captured
  // into a hipGraph once per (problem, algorithm, stream) and replayed with ONE launch.  graph_state: 0 none yet, 1 ready,
  // -1 capture failed on this problem (plain launches from then on).
  hipGraphExec_t graph_exec = nullptr;
  int graph_state = 0, graph_alg = -1;
  hipStream_t graph_stream = nullptr;
  void drop_graph()
  {
    if (graph_exec)
      (void)hipGraphExecDestroy(graph_exec);
    graph_exec = nullptr;
    graph_state = 0;
  }
  bool profiling = false;
  std::vector<hipEvent_t> events; // 4 per profiled launch: before / after the diagonal kernel, before / after the coupling kernel
  size_t ev_used = 0;
  hipEvent_t next_event()
  {
    if (ev_used == events.size())
      {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess)
          return nullptr;
        events.push_back(e);
      }
    return events[ev_used++];
  }
};

static thread_local std::string g_err_noctx;

static int fail(pdh_ctx *ctx, int code, const std::string &msg)
{
  if (ctx)
    ctx->err = msg;
  else
    g_err_noctx = msg;
  return code;
}

#define PDH_HIP(ctx, call)                                                                         \
  do                                                                                               \
    {                                                                                              \
      hipError_t e_ = (call);                                                                      \
      if (e_ != hipSuccess)                                                                        \
        return fail(ctx, PDH_EDEVICE, std::string(#call) + ": " + hipGetErrorString(e_));          \
    }                                                                                              \
  while (0)

static void free_problem(pdh_ctx *ctx)
{
  for (void *p : ctx->allocs)
    (void)hipFree(p);
  ctx->allocs.clear();
  ctx->drop_graph();
  ctx->has_problem = false;
  ctx->d_ap_src = nullptr;
  ctx->d_bd_rng = nullptr;
}

template <class T>
static int upload_n(pdh_ctx *ctx, const T *h, size_t count, const T **dptr)
{
  void *d = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  PDH_HIP(ctx, hipMalloc(&d, bytes));
  ctx->allocs.push_back(d);
  if (count)
    PDH_HIP(ctx, hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice));
  *dptr = static_cast<const T *>(d);
  return PDH_OK;
}
template <class V>
static int upload(pdh_ctx *ctx, const V &h, const typename V::value_type **dptr)
{
  return upload_n(ctx, h.data(), h.size(), dptr);
}

#ifndef PDH_SRC_HASH
#define PDH_SRC_HASH "unhashed"
#endif
// "polydeal_hip <version>+<hash of the library's sources and build flags> gfx950" (Makefile: SRC_HASH)
extern "C" const char *pdh_version(void) { return "polydeal_hip 0.4+" PDH_SRC_HASH " gfx950"; }

extern "C" const char *pdh_last_error(const pdh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err_noctx.c_str(); }

extern "C" int pdh_create(pdh_ctx **out, int device_id)
{
  if (!out)
    return fail(nullptr, PDH_EINVAL, "pdh_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, PDH_EDEVICE,
                std::string("pdh_create: no HIP device available (") + hipGetErrorString(e) +
                  "); this library has no CPU fallback");
  if (device_id < 0 || device_id >= ndev)
    return fail(nullptr, PDH_EINVAL, "pdh_create: device_id out of range");
  PDH_HIP(nullptr, hipSetDevice(device_id));
  pdh_ctx *ctx = new pdh_ctx;
  ctx->device = device_id;
  if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess)
    {
      delete ctx;
      return fail(nullptr, PDH_EDEVICE, "pdh_create: second stream / events");
    }
  if (hipStreamCreate(&ctx->own_stream) != hipSuccess)
    {
      delete ctx;
      return fail(nullptr, PDH_EDEVICE, "pdh_create: hipStreamCreate failed");
    }
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return PDH_OK;
}

extern "C" void pdh_destroy(pdh_ctx *ctx)
{
  if (!ctx)
    return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_problem(ctx);
  for (auto &sc : ctx->scratch)
    if (sc.p)
      (void)hipFree(sc.p);
  if (ctx->d_shape_midx)
    (void)hipFree(ctx->d_shape_midx);
  for (auto &ev : ctx->events)
    if (ev)
      (void)hipEventDestroy(ev);
  (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->stream2)
    (void)hipStreamDestroy(ctx->stream2);
  if (ctx->ev_fork)
    (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join)
    (void)hipEventDestroy(ctx->ev_join);
  delete ctx;
}

// ---------------------------------------------------------------------------------------------------
// Host-only part of set_problem: validation + repacking.  Kept separate so that it can be exercised on
// a machine without a GPU (tests call pdh_check_problem).
// ---------------------------------------------------------------------------------------------------
// std::vector whose resize() leaves the new elements uninitialised: the big point arrays are filled by all host threads
// right after they are sized, a serial zero-fill of 1.3 GB in between costs more than the fill itself
template <class T>
struct UninitAlloc : std::allocator<T>
{
  template <class U>
  struct rebind
  {
    using other = UninitAlloc<U>;
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
using dvec = std::vector<double, UninitAlloc<double>>;

struct Packed
{
  int n = 0, n1d = 0, NT = 0, LB = 0;
  bool tiled = false; // n > 64: blocks in 64 x 64 tiles (pdh_tiled.h)
  std::vector<int32_t> midx;
  PdhBasisTab tab;
  std::vector<int32_t> own_agg, own_row, row_len, diag_L, it_own, it_nbr, it_pcnt, it_pos, it_nbr_slot, it_pos_t;
  std::vector<int64_t> row_base, vq_ptr, ap_ptr, it_pbeg;
  dvec vq_x, vq_w;
  // Own-side face points are NOT built on the host: set_problem uploads the caller's face arrays as they are and a kernel
  // (pdh_exchange.hip: k_pack_faces) writes the per-polytope runs in HBM from these tables - one entry per run, owned slots
  // first, then the pseudo slots of the exchange variant: first packed point, first caller point, count, flags
  // (bit 0: the polytope is side 0 of the face, bit 1: boundary face), sigma of the face
  std::vector<int64_t> pk_at, pk_fq;
  std::vector<int32_t> pk_cnt, pk_flags;
  std::vector<double> pk_sig;
  int64_t n_ap = 0;
  // volume points of the owned slots: the caller's own arrays when the slots are its polytopes in its order (no copy),
  // else vq_x / vq_w above; [dim][vq_stride] and [n_vq]
  const double *vqx_h = nullptr, *vqw_h = nullptr;
  int64_t vq_stride_h = 0, n_vq = 0;
  std::vector<int64_t> vq_src, run_ap, run_fq;
  std::vector<int32_t> run_cnt, run_bdry;
  // per run (owned slots only, same order): owning slot, neighbour polytope (-1 boundary) and the ascending rank of the
  // neighbour's block in the slot's rows, penalty as stored per point - input of the row kernel's face table (pdh_rows.h)
  std::vector<int32_t> run_slot, run_nbr, run_blk, run_face;
  std::vector<double> run_sig;
  // pdh_set_problem_cartesian: the point arrays of `src` are NULL, the points are generated on the device from these
  const pdh_cartesian_points *cart = nullptr;
  // host view of a packed face point (what the kernel writes): run r of the owned slots, point q of the run
  const pdh_problem *src = nullptr;
  int64_t nqf_src = 0;
  double ap_x(int d, size_t r, int64_t q) const { return src->fq_x[d * nqf_src + pk_fq[r] + q]; }
  double ap_n(int d, size_t r, int64_t q) const { return ((pk_flags[r] & 1) ? 1.0 : -1.0) * src->fq_n[d * nqf_src + pk_fq[r] + q]; }
  double ap_wself(size_t r, int64_t q) const
  {
    const int64_t i = pk_fq[r] + q;
    if (pk_flags[r] & 2)
      return 2.0 * src->fq_w[i];
    return ((pk_flags[r] & 1) || !src->fq_w_out) ? src->fq_w[i] : src->fq_w_out[i];
  }
  double ap_wcross(size_t r, int64_t q) const
  {
    const int64_t i = pk_fq[r] + q;
    return (pk_flags[r] & 2) ? 0.0 : (src->fq_w_out ? src->fq_w_out[i] : src->fq_w[i]);
  }
  int64_t n_values = 0;
  int n_owned = 0; // own_agg / ap_ptr / ... may carry pseudo slots behind the owned ones (ghost-block exchange)
  // ghost-block exchange (PDH_EXCHANGE_GHOST): doubles per peer rank, and where the received blocks go
  std::vector<int64_t> send_count, recv_count;
  int64_t n_send = 0, n_recv = 0;
  std::vector<int32_t> r21_face, r21_rlen, r22_slot;
  std::vector<int64_t> r21_src, r21_dst, r22_ptr, r22_src;
};

static int pack_problem(pdh_ctx *ctx, const pdh_problem *p, int32_t row_begin, int32_t row_end, Packed &K, int exchange_mode = PDH_EXCHANGE_NONE,
                        const pdh_cartesian_points *cart = nullptr)
{
  if (!p)
    return fail(ctx, PDH_EINVAL, "problem is NULL");
  K.cart = cart;
  if (p->dim != 2 && p->dim != 3)
    return fail(ctx, PDH_EINVAL, "dim must be 2 or 3");
  if (p->basis != PDH_BASIS_DGQ && p->basis != PDH_BASIS_AGGLODGP)
    return fail(ctx, PDH_EINVAL, "unknown basis");
  if (p->degree < 0 || p->degree + 1 > PDH_MAX_N1D)
    return fail(ctx, PDH_EUNSUPPORTED, "degree must be in [0,7]");
  if (p->n_agg <= 0 || p->n_faces < 0)
    return fail(ctx, PDH_EINVAL, "n_agg must be positive and n_faces non-negative");
  if (!p->bbox || !p->dof_offset || !p->vq_ptr || (!cart && (!p->vq_x || !p->vq_w)) || !p->rowptr)
    return fail(ctx, PDH_EINVAL, "a required array is NULL");
  if (p->n_faces > 0 && (!p->face_in || !p->face_out || !p->fq_ptr || (!cart && (!p->fq_x || !p->fq_n || !p->fq_w)) || !p->face_sigma))
    return fail(ctx, PDH_EINVAL, "a required face array is NULL");
  if (cart)
    { // compact description of Cartesian cells: the groups of points must be whole rules on valid cells
      if (p->dim != 3 || cart->nq < 1 || cart->nq > PDH_MAX_N1D || cart->nqf < 1 || cart->nqf > PDH_MAX_N1D)
        return fail(ctx, PDH_EINVAL, "cartesian description: dim must be 3 and 1 <= nq, nqf <= 8");
      if (cart->n_cells <= 0 || !cart->cell_box || !cart->vq_cell || (p->n_faces > 0 && (!cart->fq_cell || !cart->fq_face)))
        return fail(ctx, PDH_EINVAL, "cartesian description: a required array is NULL");
      const int64_t m3 = (int64_t)cart->nq * cart->nq * cart->nq, m2 = (int64_t)cart->nqf * cart->nqf;
      for (int a = 0; a < p->n_agg; ++a)
        if ((p->vq_ptr[a + 1] - p->vq_ptr[a]) % m3)
          return fail(ctx, PDH_EINVAL, "cartesian description: the volume points of a polytope are not whole groups of nq^3");
      for (int f = 0; f < p->n_faces; ++f)
        if ((p->fq_ptr[f + 1] - p->fq_ptr[f]) % m2)
          return fail(ctx, PDH_EINVAL, "cartesian description: the points of a face are not whole groups of nqf^2");
      for (int64_t g = 0; g < p->vq_ptr[p->n_agg] / m3; ++g)
        if (cart->vq_cell[g] < 0 || cart->vq_cell[g] >= cart->n_cells)
          return fail(ctx, PDH_EINVAL, "cartesian description: vq_cell out of range");
      for (int64_t g = 0; g < (p->n_faces ? p->fq_ptr[p->n_faces] / m2 : 0); ++g)
        if (cart->fq_cell[g] < 0 || cart->fq_cell[g] >= cart->n_cells || cart->fq_face[g] < 0 || cart->fq_face[g] > 5)
          return fail(ctx, PDH_EINVAL, "cartesian description: fq_cell / fq_face out of range");
      for (int64_t c = 0; c < cart->n_cells; ++c)
        for (int d = 0; d < 3; ++d)
          if (!std::isfinite(cart->cell_box[c * 6 + d]) || !(cart->cell_box[c * 6 + 3 + d] > cart->cell_box[c * 6 + d]))
            return fail(ctx, PDH_EINVAL, "cartesian description: degenerate cell box");
    }
  const int dim = p->dim;
  const int n = pdh::n_dofs_per_cell(dim, p->degree, p->basis);
  // more than 64 dofs per polytope: blocks are computed in 64 x 64 tiles (pdh_tiled.h; 3-D, degree 4 .. 7)
  K.tiled = n > 64;
  if (K.tiled && !pdh_tiled_has_kind(p->dim, p->degree + 1, n))
    return fail(ctx, PDH_EUNSUPPORTED, "more than 64 dofs per polytope are supported in 3-D for degree 4 .. 7 only");
  const bool local = p->local != 0;
  if (!local && (int64_t)n * p->n_agg != p->n_rows)
    return fail(ctx, PDH_EINVAL, "n_rows != dofs_per_cell * n_agg (global description)");
  if (p->n_rows <= 0 || p->n_rows % n)
    return fail(ctx, PDH_EINVAL, "n_rows must be a positive multiple of dofs_per_cell");
  if (row_begin < 0 || row_end > p->n_rows || row_begin > row_end || row_begin % n || row_end % n)
    return fail(ctx, PDH_EINVAL, "owned row range must be aligned to whole polytopes");
  if (p->col_offset && p->diag_first)
    return fail(ctx, PDH_EINVAL, "col_offset (Epetra column order) requires diag_first = 0");
  // rowptr covers all rows (global description) or the owned rows only (rank-local description)
  const int64_t rp_shift = local ? (int64_t)row_begin : 0;
  if (p->rowptr[0] != 0)
    return fail(ctx, PDH_EINVAL, "rowptr[0] must be 0");
  K.n = n;
  K.n1d = p->degree + 1;
  const int T = (n + 3) / 4;
  K.NT = (T + 3) / 4;
  K.LB = T - 4 * (K.NT - 1);
  if (K.tiled)
    K.NT = K.LB = 4; // every tile is the full 64 x 64 product
  else if (combo_group(dim, K.n1d, K.NT, K.LB) < 0)
    return fail(ctx, PDH_EUNSUPPORTED, "no kernel instantiated for this (dim, basis, degree)");

  // basis tables
  const pdh::Basis1D b1 = (p->basis == PDH_BASIS_DGQ) ? pdh::lagrange_basis(p->degree) : pdh::legendre_basis(p->degree);
  std::memset(&K.tab, 0, sizeof(K.tab));
  for (int k = 0; k < K.n1d; ++k)
    for (int m = 0; m < K.n1d; ++m)
      K.tab.coef[k][m] = (double)b1.coef[k][m];
  const auto mi = pdh::multi_indices(dim, p->degree, p->basis);
  K.midx.assign(K.tiled ? (size_t)(n + 63) / 64 * 64 : (size_t)16 * K.NT, (int32_t)0xffffffffu);
  for (int i = 0; i < n; ++i)
    K.midx[i] = (int32_t)mi[i];

  const int nA = p->n_agg, nF = p->n_faces;
  // number that orders the blocks of a row: the global dof number, or the caller's column numbering (Epetra local ids)
  auto colnum = [&](int a) { return p->col_offset ? p->col_offset[a] : p->dof_offset[a]; };
  auto owned = [&](int a) { return p->dof_offset[a] >= row_begin && p->dof_offset[a] < row_end; };
  if (p->vq_ptr[0] != 0 || (nF > 0 && p->fq_ptr[0] != 0))
    return fail(ctx, PDH_EINVAL, "vq_ptr[0] and fq_ptr[0] must be 0");
  for (int a = 0; a < nA; ++a)
    {
      const int off = p->dof_offset[a];
      if (off < 0 || off % n || off + n > p->n_rows)
        return fail(ctx, PDH_EINVAL, "dof_offset must be a multiple of dofs_per_cell inside [0,n_rows)");
      if (p->col_offset && (p->col_offset[a] < 0 || p->col_offset[a] % n))
        return fail(ctx, PDH_EINVAL, "col_offset must be a non-negative multiple of dofs_per_cell");
      for (int c = 0; c < 2 * dim; ++c)
        if (!std::isfinite(p->bbox[(size_t)a * 2 * dim + c]))
          return fail(ctx, PDH_EINVAL, "bounding box is not finite");
      if (p->vq_ptr[a + 1] < p->vq_ptr[a])
        return fail(ctx, PDH_EINVAL, "vq_ptr must be non-decreasing");
      // (the weights themselves are checked below, by all host threads)
      for (int c = 0; c < dim; ++c)
        if (!(p->bbox[(size_t)a * 2 * dim + dim + c] > p->bbox[(size_t)a * 2 * dim + c]))
          return fail(ctx, PDH_EINVAL, "degenerate bounding box");
    }
  // faces per polytope (CSR by counting)
  std::vector<int64_t> fptr(nA + 1, 0);
  for (int f = 0; f < nF; ++f)
    {
      const int in = p->face_in[f], out = p->face_out[f];
      if (in < 0 || in >= nA || out < -1 || out >= nA || out == in)
        return fail(ctx, PDH_EINVAL, "face_in/face_out out of range");
      if (!std::isfinite(p->face_sigma[f]))
        return fail(ctx, PDH_EINVAL, "face_sigma is not finite");
      if (p->fq_ptr[f + 1] < p->fq_ptr[f])
        return fail(ctx, PDH_EINVAL, "fq_ptr must be non-decreasing");
      // (weights: checked below)
      ++fptr[in + 1];
      if (out >= 0)
        ++fptr[out + 1];
    }
  for (int a = 0; a < nA; ++a)
    fptr[a + 1] += fptr[a];
  std::vector<int32_t> flist(fptr[nA]);
  {
    std::vector<int64_t> cur(fptr.begin(), fptr.end() - 1);
    for (int f = 0; f < nF; ++f)
      {
        flist[cur[p->face_in[f]]++] = f;
        if (p->face_out[f] >= 0)
          flist[cur[p->face_out[f]]++] = f;
      }
  }

  const int64_t nq_tot = p->vq_ptr[nA];
  const int64_t nqf_tot = nF ? p->fq_ptr[nF] : 0;
  if (!cart) // (generated weights are products of positive box sides and Gauss weights)
  {
    // JxW must be non-negative (and not NaN): chunks of 64k points per task
    const size_t nchunk = (size_t)((nq_tot + nqf_tot) / 65536 + 1);
    std::vector<char> bad(nchunk, 0);
    host_parallel_for(nchunk, [&](size_t k) {
      const int64_t b = (int64_t)k * 65536, e = std::min<int64_t>(b + 65536, nq_tot + nqf_tot);
      for (int64_t i = b; i < e; ++i)
        {
          if (i < nq_tot)
            bad[k] |= !(p->vq_w[i] >= 0.0);
          else
            bad[k] |= !(p->fq_w[i - nq_tot] >= 0.0) || (p->fq_w_out && !(p->fq_w_out[i - nq_tot] >= 0.0));
        }
    });
    for (char b : bad)
      if (b)
        return fail(ctx, PDH_EINVAL, "quadrature weights (JxW) must be non-negative");
  }
  const int64_t val_base = p->rowptr[row_begin - rp_shift];
  K.n_values = p->rowptr[row_end - rp_shift] - val_base;

  // ---- ghost-block exchange variant (pdh_set_exchange_mode): which faces are cut by the partition ------------------
  // A face whose sides live on different ranks is assembled by the rank that owns side 0 (the caller lists every face from
  // its owner side: the reference's `id() < neighbor->id()` rule, include/poly_utils.h:2089, 2134-2190): that rank adds
  // M11, M12 to its own rows and ships M21 (one block per face) and M22 (summed per remote polytope) to the other rank.
  const bool ghost = exchange_mode == PDH_EXCHANGE_GHOST;
  if (ghost && K.tiled)
    return fail(ctx, PDH_EUNSUPPORTED, "more than 64 dofs per polytope run owner-computes-rows only (no ghost-block exchange)");
  int my_rank = -1, n_ranks = 1;
  if (ghost)
    {
      if (!p->agg_rank)
        return fail(ctx, PDH_EINVAL, "the ghost-block exchange needs agg_rank (owning rank of every polytope)");
      for (int a = 0; a < nA; ++a)
        {
          if (p->agg_rank[a] < 0)
            return fail(ctx, PDH_EINVAL, "agg_rank must be non-negative");
          n_ranks = std::max(n_ranks, p->agg_rank[a] + 1);
          if (owned(a))
            {
              if (my_rank >= 0 && p->agg_rank[a] != my_rank)
                return fail(ctx, PDH_EINVAL, "owned polytopes carry different ranks in agg_rank");
              my_rank = p->agg_rank[a];
            }
        }
      for (int a = 0; a < nA; ++a)
        if (!owned(a) && p->agg_rank[a] == my_rank)
          return fail(ctx, PDH_EINVAL, "a polytope outside the owned row range carries the owner's rank in agg_rank");
    }
  struct Cut { int peer, dof_in, dof_out, f; };
  std::vector<Cut> cut_send, cut_recv; // faces owned here with a remote side 1 / owned remotely with side 1 here
  if (ghost)
    for (int f = 0; f < nF; ++f)
      {
        const int in = p->face_in[f], out = p->face_out[f];
        if (out < 0)
          continue;
        if (owned(in) && !owned(out))
          cut_send.push_back({p->agg_rank[out], p->dof_offset[in], p->dof_offset[out], f});
        else if (!owned(in) && owned(out))
          cut_recv.push_back({p->agg_rank[in], p->dof_offset[in], p->dof_offset[out], f});
      }
  auto cut_less = [](const Cut &x, const Cut &y) {
    return x.peer != y.peer ? x.peer < y.peer : (x.dof_in != y.dof_in ? x.dof_in < y.dof_in : x.dof_out < y.dof_out);
  };
  std::sort(cut_send.begin(), cut_send.end(), cut_less);
  std::sort(cut_recv.begin(), cut_recv.end(), cut_less);
  // per peer: M21 blocks in the order above, then one M22 block per distinct side-1 polytope, ascending by its dof
  std::vector<int64_t> send_block_of_face(ghost ? nF : 0, -1); // block index (units of n^2 doubles) of a face's M21
  std::vector<std::pair<int, int>> send22; // (peer, remote polytope), sorted; block index follows
  std::vector<int64_t> send22_block;
  K.send_count.assign(n_ranks, 0);
  K.recv_count.assign(n_ranks, 0);
  {
    int64_t blk = 0;
    size_t i = 0;
    while (i < cut_send.size())
      {
        const int peer = cut_send[i].peer;
        const int64_t blk0 = blk;
        std::vector<std::pair<int, int>> outs; // (dof, polytope)
        for (; i < cut_send.size() && cut_send[i].peer == peer; ++i)
          {
            send_block_of_face[cut_send[i].f] = blk++;
            outs.emplace_back(cut_send[i].dof_out, p->face_out[cut_send[i].f]);
          }
        std::sort(outs.begin(), outs.end());
        outs.erase(std::unique(outs.begin(), outs.end()), outs.end());
        for (const auto &o : outs)
          {
            send22.emplace_back(peer, o.second);
            send22_block.push_back(blk++);
          }
        K.send_count[peer] = (blk - blk0) * (int64_t)n * n;
      }
    K.n_send = blk * (int64_t)n * n;
  }

  // owned polytopes in polytope order; in ghost mode followed by one pseudo slot per remote polytope that receives an M22
  K.vq_ptr.push_back(0);
  K.ap_ptr.push_back(0);
  struct Run { int a, f; }; // own-side points of face f seen from polytope a
  std::vector<Run> runs;
  std::vector<int64_t> run_slot_end; // runs.size() after every slot
  std::vector<std::pair<int32_t, int32_t>> blocks; // (column number, polytope)
  std::vector<int32_t> slot_of(nA, -1);
  // the points themselves are copied after this (serial) bookkeeping pass, by all host threads
  int64_t nap_run = 0, nvq_run = 0;
  std::vector<int64_t> run_at; // first packed point of every run
  auto append_run_points = [&](int a, int f) {
    runs.push_back({a, f});
    run_at.push_back(nap_run);
    nap_run += p->fq_ptr[f + 1] - p->fq_ptr[f];
  };
  for (int a = 0; a < nA; ++a)
    {
      const int off = p->dof_offset[a];
      if (off < row_begin || off >= row_end)
        continue;
      const int slot = (int)K.own_agg.size();
      slot_of[a] = slot;
      K.own_agg.push_back(a);
      K.own_row.push_back(off - row_begin);
      // coupled blocks, ascending by column number (reference :954-975)
      const int ocol = colnum(a);
      blocks.clear();
      blocks.emplace_back(ocol, a);
      for (int64_t t = fptr[a]; t < fptr[a + 1]; ++t)
        {
          const int f = flist[t];
          const int other = (p->face_in[f] == a) ? p->face_out[f] : p->face_in[f];
          if (other >= 0)
            blocks.emplace_back(colnum(other), other);
        }
      std::sort(blocks.begin(), blocks.end());
      for (size_t t = 1; t < blocks.size(); ++t)
        if (blocks[t].first == blocks[t - 1].first)
          return fail(ctx, PDH_EINVAL, "two faces couple the same pair of polytopes (faces must be merged per neighbour)");
      const int64_t *rp = p->rowptr + (off - rp_shift);
      const int64_t r0 = rp[0];
      const int64_t rl = rp[1] - r0;
      if (rl != (int64_t)blocks.size() * n)
        return fail(ctx, PDH_EINVAL, "row length does not match (1 + #neighbours) * dofs_per_cell for polytope " + std::to_string(a));
      for (int i = 1; i < n; ++i)
        if (rp[i + 1] - rp[i] != rl)
          return fail(ctx, PDH_EINVAL, "rows of one polytope must have equal length");
      K.row_base.push_back(r0 - val_base);
      K.row_len.push_back((int32_t)rl);
      int own_rank = 0;
      for (size_t t = 0; t < blocks.size(); ++t)
        if (blocks[t].second == a)
          own_rank = (int)t;
      K.diag_L.push_back(own_rank * n);
      if (p->colind)
        { // verify EVERY row of the polytope against the positions the kernels write to (each row carries its own
          // diagonal-first shift); O(nnz) host work, once per problem
          for (int i = 0; i < n; ++i)
            {
              const int32_t *ci = p->colind + r0 + (int64_t)i * rl;
              const int dcol = ocol + i;
              for (size_t t = 0; t < blocks.size(); ++t)
                for (int j = 0; j < n; ++j)
                  {
                    const int col = blocks[t].first + j;
                    int64_t pos = (int64_t)t * n + j;
                    if (p->diag_first)
                      pos = (col == dcol) ? 0 : (col < dcol ? pos + 1 : pos);
                    if (ci[pos] != col)
                      return fail(ctx, PDH_EINVAL, "colind does not have the DG block layout expected for polytope " +
                                                     std::to_string(a) + " (row " + std::to_string(off + i) + ")");
                  }
            }
        }
      // volume points
      K.vq_src.push_back(p->vq_ptr[a]);
      nvq_run += p->vq_ptr[a + 1] - p->vq_ptr[a];
      K.vq_ptr.push_back(nvq_run);
      // own-side face points + coupling items
      for (int64_t t = fptr[a]; t < fptr[a + 1]; ++t)
        {
          const int f = flist[t];
          const bool side0 = (p->face_in[f] == a);
          const int other = side0 ? p->face_out[f] : p->face_in[f];
          const int64_t qb = p->fq_ptr[f], qe = p->fq_ptr[f + 1];
          const bool cut = other >= 0 && !owned(other);
          if (ghost && cut && !side0)
            { // owned by the other rank: its M21 and M22 arrive through the exchange
              size_t rank = 0;
              for (size_t u = 0; u < blocks.size(); ++u)
                if (blocks[u].second == other)
                  rank = u;
              int pos = (int)rank * n;
              if (p->diag_first && colnum(other) < ocol)
                pos += 1;
              K.r21_face.push_back(f);
              K.r21_dst.push_back(K.row_base[slot] + pos);
              K.r21_rlen.push_back((int32_t)rl);
              continue;
            }
          K.run_ap.push_back(nap_run);
          K.run_fq.push_back(qb);
          K.run_cnt.push_back((int32_t)(qe - qb));
          K.run_bdry.push_back(other < 0 ? 1 : 0);
          K.run_slot.push_back(slot);
          K.run_face.push_back(f);
          K.run_nbr.push_back(other);
          {
            int brank = -1;
            for (size_t u = 0; u < blocks.size(); ++u)
              if (other >= 0 && blocks[u].second == other)
                brank = (int)u;
            K.run_blk.push_back(brank);
          }
          K.run_sig.push_back(other < 0 ? 0.5 * p->face_sigma[f] : p->face_sigma[f]);
          if (other >= 0)
            {
              const int ooff = colnum(other);
              const bool other_owned = owned(other);
              // one item per interior face: the owned side with the lower polytope id computes A[P,Q]
              // and also writes A[Q,P] = A[P,Q]^T when Q's rows are owned here too
              if (!other_owned || a < other)
                {
                  K.it_own.push_back(slot);
                  K.it_nbr.push_back(other);
                  K.it_pbeg.push_back(nap_run);
                  K.it_pcnt.push_back((int32_t)(qe - qb));
                  size_t rank = 0;
                  for (size_t u = 0; u < blocks.size(); ++u)
                    if (blocks[u].second == other)
                      rank = u;
                  int pos = (int)rank * n;
                  if (p->diag_first && ooff < ocol)
                    pos += 1;
                  K.it_pos.push_back(pos);
                  // polytope id for now (slot resolved below); ghost mode: -2 - f marks "M21 of face f goes to the send region"
                  K.it_nbr_slot.push_back(other_owned ? other : (ghost ? -2 - f : -1));
                  K.it_pos_t.push_back(0);
                }
            }
          append_run_points(a, f);
        }
      K.ap_ptr.push_back(nap_run);
      run_slot_end.push_back((int64_t)runs.size());
    }
  if ((int64_t)K.own_agg.size() * n != (int64_t)(row_end - row_begin))
    return fail(ctx, PDH_EINVAL, "dof_offset values do not tile the owned row range");
  K.n_owned = (int)K.own_agg.size();
  // pseudo slots: the M22 sums for remote polytopes (side 1 of cut faces owned here), computed by the diagonal-block kernel
  // from the points of those faces seen from side 1 and written as plain n x n blocks into the send region
  for (size_t j = 0; j < send22.size(); ++j)
    {
      const int q = send22[j].second;
      K.own_agg.push_back(q);
      K.own_row.push_back(0);
      K.row_base.push_back(K.n_values + send22_block[j] * (int64_t)n * n);
      K.row_len.push_back(n);
      K.diag_L.push_back(0);
      K.vq_ptr.push_back(nvq_run);
      for (int64_t t = fptr[q]; t < fptr[q + 1]; ++t)
        {
          const int f = flist[t];
          if (p->face_out[f] == q && owned(p->face_in[f]))
            append_run_points(q, f);
        }
      K.ap_ptr.push_back(nap_run);
      run_slot_end.push_back((int64_t)runs.size());
    }
  {
    // resolve the neighbour's owned slot and the position of P's block inside Q's rows
    for (size_t it = 0; it < K.it_own.size(); ++it)
      {
        const int q = K.it_nbr_slot[it];
        if (q == -1)
          continue;
        if (q <= -2)
          { // ghost mode: plain n x n block of the send region, addressed like a row range through a pseudo entry
            const int f = -2 - q;
            K.it_nbr_slot[it] = (int32_t)K.row_base.size();
            K.row_base.push_back(K.n_values + send_block_of_face[f] * (int64_t)n * n);
            K.row_len.push_back(n);
            K.it_pos_t[it] = 0;
            continue;
          }
        const int pa = K.own_agg[K.it_own[it]];
        const int poff = colnum(pa), qoff = colnum(q);
        // rank of P's block among Q's coupled blocks = number of Q's blocks with a smaller column number
        int rank = (qoff < poff) ? 1 : 0;
        for (int64_t t = fptr[q]; t < fptr[q + 1]; ++t)
          {
            const int f = flist[t];
            const int o = (p->face_in[f] == q) ? p->face_out[f] : p->face_in[f];
            if (o >= 0 && o != pa && colnum(o) < poff)
              ++rank;
          }
        int pos = rank * n;
        if (p->diag_first && poff < qoff)
          pos += 1;
        K.it_pos_t[it] = pos;
        K.it_nbr_slot[it] = slot_of[q];
      }
  }
  // receive side of the exchange: where the incoming blocks go
  if (ghost)
    {
      std::vector<int64_t> recv_block_of_face(nF, -1);
      std::vector<std::pair<std::pair<int, int>, int64_t>> recv22; // ((peer, own polytope), block)
      int64_t blk = 0;
      size_t i = 0;
      while (i < cut_recv.size())
        {
          const int peer = cut_recv[i].peer;
          const int64_t blk0 = blk;
          std::vector<std::pair<int, int>> outs;
          for (; i < cut_recv.size() && cut_recv[i].peer == peer; ++i)
            {
              recv_block_of_face[cut_recv[i].f] = blk++;
              outs.emplace_back(cut_recv[i].dof_out, p->face_out[cut_recv[i].f]);
            }
          std::sort(outs.begin(), outs.end());
          outs.erase(std::unique(outs.begin(), outs.end()), outs.end());
          for (const auto &o : outs)
            recv22.push_back({{peer, o.second}, blk++});
          K.recv_count[peer] = (blk - blk0) * (int64_t)n * n;
        }
      K.n_recv = blk * (int64_t)n * n;
      for (size_t k = 0; k < K.r21_face.size(); ++k)
        K.r21_src.push_back(recv_block_of_face[K.r21_face[k]] * (int64_t)n * n);
      // M22: grouped by destination polytope so that one wave adds all contributions of a polytope, in a fixed order
      std::sort(recv22.begin(), recv22.end(), [&](const auto &x, const auto &y) {
        return x.first.second != y.first.second ? x.first.second < y.first.second : x.first.first < y.first.first;
      });
      K.r22_ptr.push_back(0);
      for (size_t k = 0; k < recv22.size(); ++k)
        {
          const int a = recv22[k].first.second;
          if (k == 0 || recv22[k - 1].first.second != a)
            {
              if (k)
                K.r22_ptr.push_back((int64_t)K.r22_src.size());
              K.r22_slot.push_back(slot_of[a]);
            }
          K.r22_src.push_back(recv22[k].second * (int64_t)n * n);
        }
      K.r22_ptr.push_back((int64_t)K.r22_src.size());
      if (K.r22_slot.empty())
        K.r22_ptr.assign(1, 0);
    }

  // second pass: weights, coordinates and normals in SoA with the final strides - disjoint ranges, all host threads
  const int64_t nvq = nvq_run, nap = nap_run;
  bool vq_identity = nvq == nq_tot;
  for (int sl = 0; sl < K.n_owned && vq_identity; ++sl)
    vq_identity = K.vq_ptr[sl] == p->vq_ptr[K.own_agg[sl]];
  K.n_vq = nvq;
  if (cart)
    K.vqx_h = K.vqw_h = nullptr, K.vq_stride_h = nvq, vq_identity = true; // (generated on the device, slot by slot)
  else if (vq_identity)
    K.vqx_h = p->vq_x, K.vqw_h = p->vq_w, K.vq_stride_h = nq_tot;
  else
    {
      K.vq_w.resize((size_t)nvq);
      K.vq_x.resize((size_t)dim * nvq);
      K.vqx_h = K.vq_x.data(), K.vqw_h = K.vq_w.data(), K.vq_stride_h = nvq;
    }
  K.n_ap = nap;
  K.src = p;
  K.nqf_src = nqf_tot;
  if (!vq_identity)
  host_parallel_for((size_t)K.n_owned, [&](size_t sl) {
    const int a = K.own_agg[sl];
    int64_t vq = K.vq_ptr[sl];
    for (int64_t q = p->vq_ptr[a]; q < p->vq_ptr[a + 1]; ++q, ++vq)
      {
        K.vq_w[vq] = p->vq_w[q];
        for (int c = 0; c < dim; ++c)
          K.vq_x[c * nvq + vq] = p->vq_x[c * nq_tot + q];
      }
  });
  // tables of the device-side face repack (k_pack_faces): weights and signs as the kernels want them -
  //   boundary: w_self = 2 JxW, sigma / 2 (Nitsche boundary = interior self-block with these exact scalings), w_cross = 0;
  //   interior: w_self = JxW of the own side (M11 uses JxW_0, M22 JxW_1: poly_utils.h:1898, 1922), w_cross = JxW_1
  //   (M12, M21: poly_utils.h:1906, 1914), normal = outward normal of the owning polytope
  K.pk_at.resize(runs.size());
  K.pk_fq.resize(runs.size());
  K.pk_cnt.resize(runs.size());
  K.pk_flags.resize(runs.size());
  K.pk_sig.resize(runs.size());
  for (size_t ri = 0; ri < runs.size(); ++ri)
    {
      const Run &r = runs[ri];
      const bool side0 = (p->face_in[r.f] == r.a);
      const int other = side0 ? p->face_out[r.f] : p->face_in[r.f];
      K.pk_at[ri] = run_at[ri];
      K.pk_fq[ri] = p->fq_ptr[r.f];
      K.pk_cnt[ri] = (int32_t)(p->fq_ptr[r.f + 1] - p->fq_ptr[r.f]);
      K.pk_flags[ri] = (side0 ? 1 : 0) | (other < 0 ? 2 : 0);
      K.pk_sig[ri] = other < 0 ? 0.5 * p->face_sigma[r.f] : p->face_sigma[r.f];
    }
  return PDH_OK;
}


// ---------------------------------------------------------------------------------------------------
// Face tables of the row kernel (pdh_rows.h).  Eligible: 3-D FE_DGQ / FE_AggloDGP of degree 1 .. 3, no exchange variant, and
// every polytopal face of every owned polytope a union of pieces of axis-aligned planes (agglomerates of Cartesian cells).
// Block-shaped polytopes meet every neighbour along ONE plane; METIS-like agglomerates meet some along several ("staircase"
// faces): FE_DGQ(3) has an instantiation for those (RowsHost::multi), the other elements need one plane per neighbour and at
// most pdh_rows_max_faces() neighbours.  The test is made on the packed points themselves, so any description qualifies that
// has the geometry - there is no mesh-type flag.  Planarity is required to a few ulp: the kernel evaluates the bases at ONE
// plane coordinate per entry (the mean).
// ---------------------------------------------------------------------------------------------------
// Are the volume points of every owned polytope tensor-product rules of n^dim points on axis-aligned boxes (pdh_problem::
// vq_tensor_n)?  Checked on the packed points to a few ulp; dim = 3.
// Relative accuracy to expect of JxW (and of unit normals) that a caller computed from vertex coordinates of size |x| on
// cells of size h: eps |x| / h per factor (differences of rounded coordinates), taken from the polytope's box; between
// 1e-13 and 1e-12 (beyond that a deviation is treated as structure, not rounding).
static double geometry_rounding(const pdh_problem *p, int a)
{
  double t = 1e-13;
  for (int d = 0; d < p->dim; ++d)
    {
      const double lo_d = p->bbox[(size_t)a * 2 * p->dim + d], hi_d = p->bbox[(size_t)a * 2 * p->dim + p->dim + d];
      t = std::max(t, 64.0 * 2.2e-16 * std::max(std::fabs(lo_d), std::fabs(hi_d)) / (hi_d - lo_d));
    }
  return std::min(t, 1e-12);
}

static bool volume_rules_are_tensor(const pdh_problem *p, const Packed &K, int n)
{
  if (n <= 0 || n > 8 || p->dim != 3)
    return false;
  const int64_t m = (int64_t)n * n * n, nvq = K.vq_stride_h;
  const double *vq_x = K.vqx_h, *vq_w = K.vqw_h;
  std::vector<char> bad((size_t)K.n_owned, 0);
  host_parallel_for((size_t)K.n_owned, [&](size_t sl) {
    const int64_t b0 = K.vq_ptr[sl], e0 = K.vq_ptr[sl + 1];
    if ((e0 - b0) % m)
      {
        bad[sl] = 1;
        return;
      }
    const int a = K.own_agg[sl];
    const double wtol = geometry_rounding(p, a);
    for (int64_t b = b0; b < e0; b += m)
      {
        const double w000 = vq_w[b];
        if (!(w000 > 0.0))
          {
            bad[sl] = 1;
            return;
          }
        for (int k = 0; k < n; ++k)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
              {
                const int64_t q = b + i + (int64_t)n * (j + (int64_t)n * k);
                const int idx[3] = {i, j, k};
                const int64_t step[3] = {1, n, (int64_t)n * n};
                double wf = w000;
                for (int d = 0; d < 3; ++d)
                  {
                    const double X = vq_x[d * nvq + b + idx[d] * step[d]];
                    const double h = p->bbox[(size_t)a * 6 + 3 + d] - p->bbox[(size_t)a * 6 + d];
                    if (std::fabs(vq_x[d * nvq + q] - X) > 3e-15 * (std::fabs(X) + h))
                      {
                        bad[sl] = 1;
                        return;
                      }
                    wf *= vq_w[b + idx[d] * step[d]] / w000;
                  }
                if (std::fabs(vq_w[q] - wf) > wtol * wf)
                  {
                    bad[sl] = 1;
                    return;
                  }
              }
      }
  });
  for (char c : bad)
    if (c)
      return false;
  return true;
}

// Sub-face rules: groups of n^2 points of a run, tensor in the two tangential axes (ti < tj); returns for every run whether
// tj runs fastest (flag) - or false if some group is not a tensor rule.  dim = 3; axis[r] = normal axis of the run's plane(s)
// is not needed: the tangential axes are found per group from the first point's normal.
static bool face_rules_are_tensor(const pdh_problem *p, const Packed &K, int n, std::vector<signed char> &fast_j)
{
  if (n <= 0 || n > 8 || p->dim != 3)
    return false;
  const int m = n * n;
  const size_t nruns = K.run_ap.size();
  fast_j.assign(3 * nruns, -1); // per run and normal axis: does t_j run fastest? (-1: no group with that axis)
  std::vector<char> bad(nruns, 0);
  host_parallel_for(nruns, [&](size_t r) {
    const int cnt = K.run_cnt[r];
    if (cnt % m)
      {
        bad[r] = 1;
        return;
      }
    const int a = K.own_agg[K.run_slot[r]];
    const double wtol = geometry_rounding(p, a);
    for (int64_t b = 0; b < cnt; b += m) // b: first point of the group inside run r
      {
        int c = 0;
        for (int d = 0; d < 3; ++d)
          if (std::fabs(K.ap_n(d, r, b)) > 0.5)
            c = d;
        const int ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2;
        // which tangential coordinate changes between the first two points?
        const double h_i = p->bbox[(size_t)a * 6 + 3 + ti] - p->bbox[(size_t)a * 6 + ti];
        const bool i_moves = n > 1 && std::fabs(K.ap_x(ti, r, b + 1) - K.ap_x(ti, r, b)) > 1e-9 * h_i;
        const int f = (n == 1 || i_moves) ? 0 : 1;
        signed char &fast = fast_j[3 * r + c];
        if (fast < 0)
          fast = (signed char)f;
        else if (fast != f)
          {
            bad[r] = 1; // the kernel takes one orientation per plane of a run
            return;
          }
        const int64_t st_i = f == 0 ? 1 : n, st_j = f == 0 ? n : 1;
        for (int which = 0; which < 2; ++which)
          {
            auto w = [&](int64_t q) { return which ? K.ap_wcross(r, q) : K.ap_wself(r, q); };
            const double w00 = w(b);
            if (which && K.run_nbr[r] < 0)
              continue;
            if (!(w00 > 0.0))
              {
                bad[r] = 1;
                return;
              }
            for (int be = 0; be < n; ++be)
              for (int al = 0; al < n; ++al)
                {
                  const int64_t q = b + al * st_i + be * st_j;
                  const double Xi = K.ap_x(ti, r, b + al * st_i), Xj = K.ap_x(tj, r, b + be * st_j);
                  const double h_j = p->bbox[(size_t)a * 6 + 3 + tj] - p->bbox[(size_t)a * 6 + tj];
                  if (std::fabs(K.ap_x(ti, r, q) - Xi) > 3e-15 * (std::fabs(Xi) + h_i) ||
                      std::fabs(K.ap_x(tj, r, q) - Xj) > 3e-15 * (std::fabs(Xj) + h_j))
                    {
                      bad[r] = 1;
                      return;
                    }
                  const double wf = w(b + al * st_i) * (w(b + be * st_j) / w00);
                  if (std::fabs(w(q) - wf) > wtol * wf)
                    {
                      bad[r] = 1;
                      return;
                    }
                }
          }
      }
  });
  for (char c : bad)
    if (c)
      return false;
  return true;
}

// pdh_problem::vq_tensor_n / fq_tensor_n: > 0 a claim to verify, 0 find out (2 .. 8 points per direction are tried, largest
// first: a wrong candidate fails on the first group of points), < 0 do not look
template <class Check>
static int resolve_tensor_hint(int hint, Check &&holds)
{
  if (hint > 0)
    return holds(hint) ? hint : 0;
  if (hint < 0)
    return 0;
  for (int n = 8; n >= 2; --n)
    if (holds(n))
      return n;
  return 0;
}

struct RowsHost
{
  std::vector<int32_t> fr_ptr, fr_pcnt, fr_nbr, fr_axis, fr_blk, fr_flags;
  std::vector<int64_t> fr_pbeg;
  std::vector<double> fr_coord, fr_sigma, fr_nsign;
  std::vector<double> meta; // per-slot records of the kernel (pdh_rows.h: 12 + 12 maxe doubles each)
  int fq_tensor_n = 0; // verified (or detected) points per direction of the sub-face rules, 0: none
  // pdh_rows.h, MULTI instantiation: some neighbour is met along several planes, or a polytope has more interior plane
  // entries / entries than the block-shaped kernel provides for (6 / 16)
  bool multi = false;
  int maxe = 16, maxf = 6;
  int maxs = 0; // most sub-faces (groups of a tensor rule) of the interior entries of one polytope
  // every face point of every owned polytope has an axis-aligned normal and lies in the plane of its sub-face: established
  // before the element-specific limits of the kinds of pdh_rows.h are looked at (the term kernel, pdh_terms.h, needs no more)
  bool planar_ok = false;
  std::vector<signed char> fast_j; // per run and normal axis: does the second tangential axis run fastest in the sub-face rules?
};
static bool build_rows_tables(const pdh_problem *p, const Packed &K, RowsHost &R, std::string *why = nullptr)
{
  std::vector<signed char> &fast_j = R.fast_j;
  R.fq_tensor_n = resolve_tensor_hint(p->fq_tensor_n, [&](int n) { return face_rules_are_tensor(p, K, n, fast_j); });
  if (R.fq_tensor_n == 0)
    fast_j.clear();
  auto no = [&](const char *m) {
    if (why)
      *why = m;
    return false;
  };
  if (p->dim != 3 || p->degree < 1 || p->degree > 3 || K.n != pdh_rows_n_dofs(p->degree + 1, p->basis == PDH_BASIS_AGGLODGP ? 1 : 0))
    return no("not 3-D FE_DGQ / FE_AggloDGP of degree 1 .. 3");
  if ((int)K.own_agg.size() != K.n_owned) // pseudo slots of the exchange variant
    return no("exchange variant");
  const size_t nruns = K.run_ap.size();
  const int maxf = pdh_rows_max_faces();
  // Planes of every run.  An interior face must lie in one plane.  The boundary "face" of a polytope collects ALL its
  // domain-boundary sub-faces (reference source/agglomeration_handler.cc:1575-1613), up to three planes at a corner: it
  // becomes one entry per plane over the same point range, and the kernel masks the points of the other planes.
  struct Plane { int axis; double sign, coord; };
  std::vector<std::vector<Plane>> planes(nruns);
  std::vector<const char *> why_run(nruns, nullptr);
  host_parallel_for(nruns, [&](size_t r) {
    auto no = [&](const char *m) { why_run[r] = m; };
    {
      const int cnt = K.run_cnt[r];
      if (cnt <= 0)
        return no("empty face");
      const int a = K.own_agg[K.run_slot[r]];
      // (tangential components of a computed unit normal: of the order of the rounding of the geometry)
      const double ntol = geometry_rounding(p, a);
      std::vector<double> sum;
      std::vector<int> num;
      for (int q = 0; q < cnt; ++q)
        {
          int c = -1;
          for (int d = 0; d < 3; ++d)
            if (std::fabs(K.ap_n(d, r, q)) > 0.5)
              c = d;
          if (c < 0)
            return no("normal not axis-aligned");
          const double sg = K.ap_n(c, r, q) > 0 ? 1.0 : -1.0;
          for (int d = 0; d < 3; ++d)
            {
              const double nd = K.ap_n(d, r, q);
              if (d == c ? std::fabs(nd - sg) > ntol : std::fabs(nd) > ntol)
                return no("normal not axis-aligned");
            }
          const double x = K.ap_x(c, r, q);
          const double h = p->bbox[(size_t)a * 6 + 3 + c] - p->bbox[(size_t)a * 6 + c];
          size_t k = 0;
          for (; k < planes[r].size(); ++k)
            if (planes[r][k].axis == c && planes[r][k].sign == sg && std::fabs(planes[r][k].coord - x) <= 1e-9 * h)
              break;
          if (k == planes[r].size())
            {
              planes[r].push_back({c, sg, x});
              sum.push_back(0.0);
              num.push_back(0);
            }
          sum[k] += x;
          num[k] += 1;
        }
      // (an interior face may span several planes - one entry each, like the boundary run of a corner polytope: whether
      // the element's kernel can take that is decided below)
      // the kernel evaluates the bases at ONE coordinate per plane (the mean): the points must agree with it to a few ulp
      for (size_t k = 0; k < planes[r].size(); ++k)
        planes[r][k].coord = sum[k] / num[k];
      for (int q = 0; q < cnt; ++q)
        for (const Plane &pl : planes[r])
          {
            const double x = K.ap_x(pl.axis, r, q);
            const double h = p->bbox[(size_t)a * 6 + 3 + pl.axis] - p->bbox[(size_t)a * 6 + pl.axis];
            const bool mine = K.ap_n(pl.axis, r, q) * pl.sign > 0.5 && std::fabs(x - pl.coord) <= 1e-9 * h;
            if (mine && std::fabs(x - pl.coord) > 2e-15 * (std::fabs(pl.coord) + h))
              return no("face not planar");
          }
    }
  });
  for (size_t r = 0; r < nruns; ++r)
    if (why_run[r])
      return no(why_run[r]);
  R.planar_ok = true;
  // runs are stored slot by slot; order the faces of a slot: boundary first, then ascending block rank
  R.fr_ptr.assign(1, 0);
  size_t r = 0;
  for (int sl = 0; sl < K.n_owned; ++sl)
    {
      std::vector<size_t> idx;
      for (; r < nruns && K.run_slot[r] == sl; ++r)
        idx.push_back(r);
      std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return K.run_blk[x] < K.run_blk[y]; });
      int nf = 0;
      for (size_t t : idx)
        for (const Plane &pl : planes[t])
          {
            // A run that lies in several planes (the boundary run of a corner polytope; a staircase face towards one neighbour)
            // gives one entry per plane, and the kernel skips the sub-faces (groups of a tensor rule, else single points) whose
            // first point is not in the entry's plane.  The entry covers only the span from the first to the last group of ITS
            // plane - a run of s groups in k planes costs the sum of the spans, not k s, in lane tasks and moment sums - and is
            // flagged for masking only if a foreign group lies inside that span.
            const int64_t gsz = R.fq_tensor_n > 0 ? (int64_t)R.fq_tensor_n * R.fq_tensor_n : 1;
            int64_t sp_b = 0, sp_e = K.run_cnt[t];
            bool foreign_inside = false;
            if (planes[t].size() > 1)
              {
                const int a = K.own_agg[K.run_slot[t]];
                const double h = p->bbox[(size_t)a * 6 + 3 + pl.axis] - p->bbox[(size_t)a * 6 + pl.axis];
                auto mine = [&](int64_t q) {
                  return K.ap_n(pl.axis, t, q) * pl.sign > 0.5 && std::fabs(K.ap_x(pl.axis, t, q) - pl.coord) <= 1e-9 * h;
                };
                int64_t g_first = -1, g_last = -1;
                const int64_t ng = K.run_cnt[t] / gsz;
                for (int64_t g = 0; g < ng; ++g)
                  if (mine(g * gsz))
                    {
                      if (g_first < 0)
                        g_first = g;
                      g_last = g;
                    }
                // (a plane seen only on points that are not the first of their group would be a group straddling planes: not a
                // tensor rule on a rectangle, such runs fail face_rules_are_tensor; with gsz = 1 every point is its own group.
                // Should it happen all the same: the whole run, masked, as before)
                if (g_first < 0)
                  g_first = 0, g_last = ng - 1;
                for (int64_t g = g_first; g <= g_last; ++g)
                  foreign_inside = foreign_inside || !mine(g * gsz);
                sp_b = g_first * gsz, sp_e = (g_last + 1) * gsz;
                if (sp_e > K.run_cnt[t] || K.run_cnt[t] % gsz) // (a run that is not whole groups: keep all of it)
                  sp_b = 0, sp_e = K.run_cnt[t], foreign_inside = true;
              }
          // A boundary run with tensor sub-face rules is cut into pieces of at most 32 sub-faces: the kernel forms the moments
          // of a piece in one batch of 64 lane tasks (2 per sub-face), and a corner polytope of 4^3 cells already has 48
          // boundary sub-faces in its run.  Boundary pieces only add to the diagonal block's face tensors, so the cut
          // changes nothing but the order of summation.  (An interior face is one entry: its coupling moments are one set.)
          for (int64_t pc0 = sp_b, pcs = (K.run_nbr[t] < 0 && R.fq_tensor_n > 0) ? 32 * gsz : sp_e - sp_b; pc0 < sp_e; pc0 += pcs)
          {
            R.fr_pbeg.push_back(K.run_ap[t] + pc0);
            R.fr_pcnt.push_back((int32_t)std::min<int64_t>(pcs, sp_e - pc0));
            R.fr_nbr.push_back(K.run_nbr[t]);
            R.fr_axis.push_back(pl.axis);
            R.fr_blk.push_back(K.run_blk[t]);
            R.fr_flags.push_back((foreign_inside ? 1 : 0) | ((R.fq_tensor_n > 0 && fast_j[3 * t + pl.axis] == 1) ? 2 : 0));
            R.fr_coord.push_back(pl.coord);
            R.fr_sigma.push_back(K.run_sig[t]);
            R.fr_nsign.push_back(pl.sign);
            if (K.run_nbr[t] >= 0)
              {
                ++nf; // the LDS layout of the kernel limits the INTERIOR entries (coupling moments kept per entry)
                if (planes[t].size() > 1)
                  R.multi = true;
              }
          }
          }
      const int ne = (int)R.fr_pbeg.size() - R.fr_ptr.back();
      if (R.fq_tensor_n > 0)
        {
          int64_t nsub = 0;
          for (size_t f = (size_t)R.fr_ptr.back(); f < R.fr_pbeg.size(); ++f)
            if (R.fr_nbr[f] >= 0)
              nsub += R.fr_pcnt[f] / ((int64_t)R.fq_tensor_n * R.fq_tensor_n);
          R.maxs = std::max<int>(R.maxs, (int)nsub);
        }
      R.maxf = std::max(R.maxf, nf);
      R.maxe = std::max(R.maxe, ne);
      R.fr_ptr.push_back((int32_t)R.fr_pbeg.size());
    }
  if (r != nruns)
    return no("run bookkeeping");
  if (getenv("PDH_ROWS_VERBOSE"))
    {
      int64_t covered = 0, points = 0, masked = 0;
      for (size_t f = 0; f < R.fr_pcnt.size(); ++f)
        covered += R.fr_pcnt[f], masked += (R.fr_flags[f] & 1) ? 1 : 0;
      for (size_t t = 0; t < nruns; ++t)
        points += K.run_cnt[t];
      fprintf(stderr, "row kernel tables: %zu runs, %zu plane entries (%lld masked), %lld face points, %lld covered by the entries (%.2fx)\n",
              nruns, R.fr_pcnt.size(), (long long)masked, (long long)points, (long long)covered, (double)covered / (double)points);
    }
  if (R.maxf > maxf || R.maxe > 16)
    R.multi = true;
  if (R.multi)
    {
      // the MULTI instantiation exists for FE_DGQ(3); its records hold up to 48 entries (the face table of a polytope lives in
      // the lanes of the wave, twelve of which carry the header) and LDS provides one 512-byte slot per interior entry
      if (!(K.n1d == 4 && p->basis != PDH_BASIS_AGGLODGP))
        return no("a neighbour is met along several planes, or a polytope has more than 6 interior / 16 face entries: FE_DGQ(3) only");
      if (R.maxe > 48 || R.maxf > 40)
        return no("too many face entries on a polytope (48 plane entries, 40 of them interior)");
      R.maxe = (R.maxe + 3) / 4 * 4;
    }
  else
    R.maxe = 16, R.maxf = maxf;
  // per-slot records: header (number of entries, own box as lo / 1/h, row base / length / position of the own block,
  // volume point range) + one entry per face with the neighbour's box - everything the kernel needs about a polytope in
  // one contiguous block
  constexpr int HDR = 12, ENT = 12;
  const int MAXE = R.maxe, REC = HDR + MAXE * ENT;
  auto as_d = [](long long v) {
    double d;
    std::memcpy(&d, &v, sizeof(d));
    return d;
  };
  R.meta.assign((size_t)K.n_owned * REC, 0.0);
  for (int sl = 0; sl < K.n_owned; ++sl)
    {
      const int f0 = R.fr_ptr[sl], nf = R.fr_ptr[sl + 1] - f0;
      if (nf > MAXE)
        return no("too many face entries on a polytope");
      double *rec = R.meta.data() + (size_t)sl * REC;
      const int a = K.own_agg[sl];
      rec[0] = as_d(nf);
      for (int c = 0; c < 3; ++c)
        {
          rec[1 + c] = p->bbox[(size_t)a * 6 + c];
          rec[4 + c] = 1.0 / (p->bbox[(size_t)a * 6 + 3 + c] - p->bbox[(size_t)a * 6 + c]);
        }
      rec[7] = as_d(K.row_base[sl]);
      rec[8] = as_d(K.row_len[sl]);
      rec[9] = as_d(K.diag_L[sl]);
      rec[10] = as_d(K.vq_ptr[sl]);
      rec[11] = as_d(K.vq_ptr[sl + 1]);
      for (int e = 0; e < nf; ++e)
        {
          double *en = rec + HDR + e * ENT;
          const int f = f0 + e, nb = R.fr_nbr[f];
          en[0] = as_d(R.fr_pbeg[f]);
          en[1] = as_d((long long)(uint32_t)R.fr_pcnt[f] | ((long long)nb << 32));
          en[2] = as_d((long long)(R.fr_axis[f] & 0xff) | ((long long)(R.fr_flags[f] & 0xff) << 8) | ((long long)R.fr_blk[f] << 32));
          en[3] = R.fr_coord[f];
          en[4] = R.fr_sigma[f];
          en[5] = R.fr_nsign[f];
          for (int c = 0; c < 3; ++c)
            {
              en[6 + c] = nb >= 0 ? p->bbox[(size_t)nb * 6 + c] : 0.0;
              en[9 + c] = nb >= 0 ? 1.0 / (p->bbox[(size_t)nb * 6 + 3 + c] - p->bbox[(size_t)nb * 6 + c]) : 1.0;
            }
        }
    }
  return true;
}

// Second half of the eligibility test: structure of the volume rules (vq_n), and - every kind but FE_DGQ(3) has no
// general-point paths - tensor rules everywhere and no face entry with more sub-faces than the 64 lane tasks of a batch
// hold (pdh_rows.h, P2).
static bool rows_kind_applies(const pdh_problem *p, const Packed &K, const RowsHost &RH, int &vq_n, bool &tensor_only,
                              std::string *why = nullptr)
{
  vq_n = resolve_tensor_hint(p->vq_tensor_n, [&](int n) { return volume_rules_are_tensor(p, K, n); });
  bool ok = vq_n > 0 && RH.fq_tensor_n > 0;
  const int64_t m = (int64_t)RH.fq_tensor_n * RH.fq_tensor_n;
  for (size_t f = 0; ok && f < RH.fr_pcnt.size(); ++f)
    ok = RH.fr_pcnt[f] / m <= 32;
  tensor_only = ok;
  if (K.n1d == 4) // degree 3 has the general-point paths (pdh_rows.h: GENERAL)
    return true;
  if (!ok && why)
    *why = "this element takes the row kernel only with tensor-product rules on every sub-cell and sub-face";
  return ok;
}

// Tables of the term kernel (pdh_terms.h): per owned polytope one record (header + one entry per run = polytopal face, the
// boundary run first, then ascending block rank) and the list of its sub-faces (groups of a verified tensor rule), run by run:
// first own-side point, run, normal axis and sign, orientation of the rule.  Applies when build_rows_tables established planar
// axis-aligned faces (RowsHost::planar_ok) and both kinds of rule are verified tensor rules; any number of planes per neighbour.
struct TermsHost
{
  std::vector<double> meta;
  std::vector<int64_t> sf_pt;
  std::vector<int32_t> sf_info, sf_ivl, cell_ivl;
  int maxruns = 0, maxsf = 0, maxsi = 0, maxcell = 0, lds_bytes = 0, split = 0, task_pts = 0;
  int64_t n_cells_in = 0, n_cells_out = 0, n_sf_in = 0, n_sf_out = 0; // before / after merging (reporting)
};

// ---- merged cells and sub-faces of the term kernels --------------------------------------------------------------------------------
// A term is a product of three 1-D matrices, one per direction; the kernels sum the terms of all cells (sub-faces) of a polytope.
// Where cells form a TENSOR GRID - cell (i, j, k) has the i-th interval of the 1-D rules along x, the j-th along y, the k-th along z,
// and its weight factorises - the sum over a sub-grid of products is the product of the sums over the intervals:
//   sum_ijk A_i (x) B_j (x) C_k = (sum_i A_i) (x) (sum_j B_j) (x) (sum_k C_k),
// i.e. the sub-grid is ONE cell with composite 1-D rules of several intervals.  Block agglomerates (what an R-tree gives on a structured
// grid) are such grids as a whole; so are the sub-faces a polytope shares with one neighbour in one plane; METIS-like agglomerates
// contain pairs and quads of cells that are.  The cells (the sub-faces of a plane) are covered greedily by boxes (rectangles) of
// occupied grid slots - found on the data (first points of the rules, compared to rounding; weights checked for the factorisation),
// never assumed; what fits no larger box stays a box of one.  A composite rule has at most 8 points and TERMS_MI intervals (register
// slots of a lane task).
struct TermsMerged
{
  struct Cell
  {
    int32_t ivl[3][TERMS_MI];
  };
  struct Sf
  {
    int64_t pb;
    int c, pos, fj, ni, nj;
    int32_t ivl[2][TERMS_MI];
  };
  std::vector<Cell> cells;
  std::vector<Sf> sfs;      // run by run, in the order of the record's runs
  std::vector<int> run_off; // [runs + 1] into sfs
  int nsf = 0, nsi = 0, ivl_c = 1, ivl_f = 1; // sub-faces, interior sub-faces, most intervals of a cell / sub-face rule
  int64_t n_in = 0;                           // cells + sub-faces as given
  size_t run_size(size_t e) const { return (size_t)(run_off[e + 1] - run_off[e]); }
  const Sf *run_begin(size_t e) const { return sfs.data() + run_off[e]; }
};
namespace
{
struct TermsSfGeom // a sub-face as the merge sees it: normal axis, side, orientation of its rule, plane and first tangential coordinates
{
  int c, pos, fj;
  double z, xi, xj;
};
// indices of the values of v in the sorted list of its distinct values (equal within tol); returns the number of distinct values
int cluster_1d(const std::vector<double> &v, double tol, std::vector<int> &idx)
{
  static thread_local std::vector<size_t> o; // (scratch: this runs once per polytope and plane on every host thread)
  o.resize(v.size());
  for (size_t i = 0; i < o.size(); ++i)
    o[i] = i;
  std::sort(o.begin(), o.end(), [&](size_t a, size_t b) { return v[a] < v[b]; });
  idx.assign(v.size(), 0);
  int n = 0;
  for (size_t k = 0; k < o.size(); ++k)
    {
      if (k > 0 && v[o[k]] - v[o[k - 1]] > tol)
        ++n;
      idx[o[k]] = n;
    }
  return v.empty() ? 0 : n + 1;
}
} // namespace

static void merge_terms_of_slot(const pdh_problem *p, const Packed &K, const RowsHost &RH, int tn, int fn, size_t sl,
                                const std::vector<size_t> &order, bool enabled, TermsMerged &M)
{
  const int a = K.own_agg[sl];
  const int64_t m3 = (int64_t)tn * tn * tn, gsz = (int64_t)fn * fn;
  const int ncell = (int)((K.vq_ptr[sl + 1] - K.vq_ptr[sl]) / m3);
  double hbox[3];
  for (int d = 0; d < 3; ++d)
    hbox[d] = p->bbox[(size_t)a * 6 + 3 + d] - p->bbox[(size_t)a * 6 + d];
  const double wtol = 8.0 * geometry_rounding(p, a);
  const bool cart = K.cart != nullptr;
  auto single_cells = [&] {
    M.cells.resize((size_t)ncell);
    for (int u = 0; u < ncell; ++u)
      for (int d = 0; d < 3; ++d)
        for (int i = 0; i < TERMS_MI; ++i)
          M.cells[(size_t)u].ivl[d][i] = i == 0 ? u : -1;
  };
  // ---------------- cells: a greedy cover by boxes of at most mc intervals per direction.  The cells are placed on the tensor grid of
  // their clustered first points; from every cell not yet taken, in index order, the largest box of taken-free, occupied slots whose
  // cells share their 1-D rules per interval (and whose weights factorise) becomes one cell with composite rules.  A polytope that is a
  // full grid is cut into sub-grids of mc intervals exactly as a chunking would; METIS-like agglomerates keep what pairs and quads they have.
  const int mc = tn >= 1 && tn <= 4 ? std::min(TERMS_MI, 8 / tn) : 1;
  bool done = false;
  if (enabled && mc > 1 && ncell > 1)
    {
      auto cellbox = [&](int u) { return K.cart->cell_box + (size_t)K.cart->vq_cell[p->vq_ptr[a] / m3 + u] * 6; };
      const int64_t b0 = K.vq_ptr[sl], st = K.vq_stride_h;
      const int64_t step[3] = {1, tn, (int64_t)tn * tn};
      static thread_local std::vector<double> key[3];
      static thread_local std::vector<int> idx[3];
      int nd[3];
      for (int d = 0; d < 3; ++d)
        {
          key[d].resize((size_t)ncell);
          for (int u = 0; u < ncell; ++u)
            key[d][(size_t)u] = cart ? cellbox(u)[d] : K.vqx_h[d * st + b0 + u * m3];
          nd[d] = cluster_1d(key[d], 1e-9 * hbox[d], idx[d]);
        }
      const int64_t nslot = (int64_t)nd[0] * nd[1] * nd[2];
      bool ok = nslot <= 4096;
      static thread_local std::vector<int> grid;
      static thread_local std::vector<char> taken;
      if (ok)
        {
          grid.assign((size_t)nslot, -1);
          for (int u = 0; u < ncell && ok; ++u)
            {
              int &g = grid[(size_t)(idx[0][u] + nd[0] * (idx[1][u] + nd[1] * idx[2][u]))];
              ok = g < 0; // (two cells on one slot: no grid)
              g = u;
            }
        }
      auto at = [&](int i, int j, int k) { return grid[(size_t)(i + nd[0] * (j + nd[1] * k))]; };
      // does cell u carry, along every direction, the 1-D rule of the box's reference cell of its interval, and does its weight factorise?
      auto fits = [&](int u, const int *o) {
        if (cart)
          { // same interval = same extent along the direction
            for (int d = 0; d < 3; ++d)
              {
                int r[3] = {o[0], o[1], o[2]};
                r[d] = idx[d][u];
                const double *bu = cellbox(u), *br = cellbox(at(r[0], r[1], r[2]));
                if (!(std::fabs(bu[d] - br[d]) <= 1e-12 * hbox[d] && std::fabs(bu[3 + d] - br[3 + d]) <= 1e-12 * hbox[d]))
                  return false;
              }
            return true;
          }
        const int64_t bu = b0 + u * m3;
        const double wu = K.vqw_h[bu];
        double wf = 1.0;
        const double w0 = K.vqw_h[b0 + at(o[0], o[1], o[2]) * m3];
        for (int d = 0; d < 3; ++d)
          {
            int r[3] = {o[0], o[1], o[2]};
            r[d] = idx[d][u];
            const int64_t br = b0 + at(r[0], r[1], r[2]) * m3;
            const double wr = K.vqw_h[br];
            wf *= wr / w0;
            for (int i = 0; i < tn; ++i)
              {
                const double X = K.vqx_h[d * st + br + i * step[d]];
                if (!(std::fabs(K.vqx_h[d * st + bu + i * step[d]] - X) <= 3e-15 * (std::fabs(X) + hbox[d]) &&
                      std::fabs(K.vqw_h[bu + i * step[d]] / wu - K.vqw_h[br + i * step[d]] / wr) <= wtol * (K.vqw_h[br + i * step[d]] / wr)))
                  return false;
              }
          }
        return std::fabs(wu - w0 * wf) <= wtol * wu;
      };
      auto box_ok = [&](const int *o, const int *sz) {
        for (int d = 0; d < 3; ++d)
          if (o[d] + sz[d] > nd[d])
            return false;
        for (int k = 0; k < sz[2]; ++k)
          for (int j = 0; j < sz[1]; ++j)
            for (int i = 0; i < sz[0]; ++i)
              {
                const int u = at(o[0] + i, o[1] + j, o[2] + k);
                if (u < 0 || taken[(size_t)u])
                  return false;
              }
        for (int k = 0; k < sz[2]; ++k)
          for (int j = 0; j < sz[1]; ++j)
            for (int i = 0; i < sz[0]; ++i)
              if (!fits(at(o[0] + i, o[1] + j, o[2] + k), o))
                return false;
        return true;
      };
      if (ok)
        {
          taken.assign((size_t)ncell, 0);
          for (int k0 = 0; k0 < nd[2]; ++k0)
            for (int j0 = 0; j0 < nd[1]; ++j0)
              for (int i0 = 0; i0 < nd[0]; ++i0)
                {
                  const int u0 = at(i0, j0, k0);
                  if (u0 < 0 || taken[(size_t)u0])
                    continue;
                  const int o[3] = {i0, j0, k0};
                  int best[3] = {1, 1, 1}, bestv = 1;
                  for (int c = mc; c >= 1; --c) // the largest box anchored here (volume; ties: the first found)
                    for (int b = mc; b >= 1; --b)
                      for (int aa = mc; aa >= 1; --aa)
                        {
                          const int sz[3] = {aa, b, c};
                          if (aa * b * c > bestv && box_ok(o, sz))
                            best[0] = aa, best[1] = b, best[2] = c, bestv = aa * b * c;
                        }
                  TermsMerged::Cell cm;
                  for (int d = 0; d < 3; ++d)
                    for (int t = 0; t < TERMS_MI; ++t)
                      {
                        int r[3] = {o[0], o[1], o[2]};
                        r[d] = o[d] + t;
                        cm.ivl[d][t] = t < best[d] ? at(r[0], r[1], r[2]) : -1;
                      }
                  for (int k = 0; k < best[2]; ++k)
                    for (int j = 0; j < best[1]; ++j)
                      for (int i = 0; i < best[0]; ++i)
                        taken[(size_t)at(i0 + i, j0 + j, k0 + k)] = 1;
                  M.cells.push_back(cm);
                }
          done = true;
        }
    }
  if (!done)
    single_cells();
  // ---------------- sub-faces, run by run
  const int mf = fn >= 1 && fn <= 4 ? std::min(TERMS_MI, 8 / fn) : 1;
  M.run_off.assign(order.size() + 1, 0);
  {
    size_t tot = 0;
    for (size_t t : order)
      tot += (size_t)(K.run_cnt[t] / gsz);
    M.sfs.reserve(tot);
  }
  auto &out = M.sfs;
  for (size_t e = 0; e < order.size(); M.run_off[e + 1] = (int)M.sfs.size(), ++e)
    {
      const size_t t = order[e];
      const int ns = (int)(K.run_cnt[t] / gsz);
      static thread_local std::vector<TermsSfGeom> gs;
      using G = TermsSfGeom;
      gs.resize((size_t)ns);
      for (int g = 0; g < ns; ++g)
        {
          G &s = gs[(size_t)g];
          if (cart)
            {
              const int64_t grp = K.pk_fq[t] / gsz + g;
              const int lf = K.cart->fq_face[grp];
              const double *bx = K.cart->cell_box + (size_t)K.cart->fq_cell[grp] * 6;
              s.c = lf >> 1;
              s.pos = ((lf & 1) != 0) == ((K.pk_flags[t] & 1) != 0) ? 1 : 0;
              s.z = bx[(lf & 1) ? 3 + s.c : s.c];
              const int ti = s.c == 0 ? 1 : 0, tj = s.c == 2 ? 1 : 2;
              s.xi = bx[ti], s.xj = bx[tj];
            }
          else
            {
              s.c = 0;
              for (int d = 0; d < 3; ++d)
                if (std::fabs(K.ap_n(d, t, g * gsz)) > 0.5)
                  s.c = d;
              s.pos = K.ap_n(s.c, t, g * gsz) > 0 ? 1 : 0;
              const int ti = s.c == 0 ? 1 : 0, tj = s.c == 2 ? 1 : 2;
              s.z = K.ap_x(s.c, t, g * gsz);
              s.xi = K.ap_x(ti, t, g * gsz), s.xj = K.ap_x(tj, t, g * gsz);
            }
          s.fj = RH.fast_j[3 * t + s.c] == 1 ? 1 : 0;
        }
      auto single = [&](int g) {
        TermsMerged::Sf f;
        f.pb = K.run_ap[t] + g * gsz;
        f.c = gs[(size_t)g].c, f.pos = gs[(size_t)g].pos, f.fj = gs[(size_t)g].fj, f.ni = f.nj = 1;
        for (int d = 0; d < 2; ++d)
          for (int i = 0; i < TERMS_MI; ++i)
            f.ivl[d][i] = 0;
        out.push_back(f);
      };
      if (!enabled || mf <= 1 || ns <= 1)
        {
          for (int g = 0; g < ns; ++g)
            single(g);
          continue;
        }
      // planes of the run: (axis, side, coordinate)
      static thread_local std::vector<char> used;
      used.assign((size_t)ns, 0);
      static thread_local std::vector<int> mem, ii, jj, fgrid;
      static thread_local std::vector<double> ki, kj;
      for (int g0 = 0; g0 < ns; ++g0)
        {
          if (used[(size_t)g0])
            continue;
          const G &r0 = gs[(size_t)g0];
          mem.clear();
          for (int g = g0; g < ns; ++g)
            if (!used[(size_t)g] && gs[(size_t)g].c == r0.c && gs[(size_t)g].pos == r0.pos && gs[(size_t)g].fj == r0.fj &&
                std::fabs(gs[(size_t)g].z - r0.z) <= 1e-9 * hbox[r0.c])
              {
                mem.push_back(g);
                used[(size_t)g] = 1;
              }
          const int c = r0.c, ti = c == 0 ? 1 : 0, tj = c == 2 ? 1 : 2;
          if (mem.size() <= 1)
            { // (a plane with one sub-face: nothing to merge)
              for (int g : mem)
                single(g);
              continue;
            }
          ki.resize(mem.size()), kj.resize(mem.size());
          for (size_t m = 0; m < mem.size(); ++m)
            ki[m] = gs[(size_t)mem[m]].xi, kj[m] = gs[(size_t)mem[m]].xj;
          const int ni = cluster_1d(ki, 1e-9 * hbox[ti], ii), nj = cluster_1d(kj, 1e-9 * hbox[tj], jj);
          // greedy cover of the plane's sub-faces by rectangles of at most mf x mf intervals (as for the cells above)
          bool ok = (int64_t)ni * nj <= 4096;
          if (ok)
            {
              fgrid.assign((size_t)ni * nj, -1);
              for (size_t m = 0; m < mem.size() && ok; ++m)
                {
                  int &gg = fgrid[(size_t)(ii[m] + ni * jj[m])];
                  ok = gg < 0;
                  gg = (int)m;
                }
            }
          if (!ok)
            {
              for (int g : mem)
                single(g);
              continue;
            }
          auto atm = [&](int i, int j) { return fgrid[(size_t)(i + ni * j)]; };        // member index or -1
          auto at = [&](int i, int j) { return mem[(size_t)fgrid[(size_t)(i + ni * j)]]; }; // sub-face of the run
          static thread_local std::vector<char> ftaken;
          ftaken.assign(mem.size(), 0);
          auto fits = [&](int m, const int *o) {
            const int g = mem[(size_t)m];
            if (cart)
              { // same interval = same extent along the tangential direction (the side-0 cell's box carries the sub-face)
                const int64_t grp = K.pk_fq[t] / gsz;
                const double *bu = K.cart->cell_box + (size_t)K.cart->fq_cell[grp + g] * 6;
                const double *bi = K.cart->cell_box + (size_t)K.cart->fq_cell[grp + at(ii[(size_t)m], o[1])] * 6;
                const double *bj = K.cart->cell_box + (size_t)K.cart->fq_cell[grp + at(o[0], jj[(size_t)m])] * 6;
                return std::fabs(bu[ti] - bi[ti]) <= 1e-12 * hbox[ti] && std::fabs(bu[3 + ti] - bi[3 + ti]) <= 1e-12 * hbox[ti] &&
                       std::fabs(bu[tj] - bj[tj]) <= 1e-12 * hbox[tj] && std::fabs(bu[3 + tj] - bj[3 + tj]) <= 1e-12 * hbox[tj];
              }
            const int fj = r0.fj;
            const int64_t st_i = fj ? fn : 1, st_j = fj ? 1 : fn;
            const int gi = at(ii[(size_t)m], o[1]), gj = at(o[0], jj[(size_t)m]), go = at(o[0], o[1]);
            for (int which = 0; which < 2; ++which)
              { // own-side and cross weights (the latter zero on the boundary)
                auto W = [&](int gq, int64_t q) { return which ? K.ap_wcross(t, gq * gsz + q) : K.ap_wself(t, gq * gsz + q); };
                const double wu = W(g, 0), wi = W(gi, 0), wj = W(gj, 0), wo = W(go, 0);
                if (which && wu == 0.0 && wo == 0.0)
                  continue;
                if (!(wu > 0.0 && wo > 0.0 && std::fabs(wu - wi * wj / wo) <= wtol * wu))
                  return false;
                for (int al = 0; al < fn; ++al)
                  {
                    const double Xi = K.ap_x(ti, t, gi * gsz + al * st_i), Xj = K.ap_x(tj, t, gj * gsz + al * st_j);
                    if (!(std::fabs(K.ap_x(ti, t, g * gsz + al * st_i) - Xi) <= 3e-15 * (std::fabs(Xi) + hbox[ti]) &&
                          std::fabs(K.ap_x(tj, t, g * gsz + al * st_j) - Xj) <= 3e-15 * (std::fabs(Xj) + hbox[tj]) &&
                          std::fabs(W(g, al * st_i) / wu - W(gi, al * st_i) / wi) <= wtol * (W(gi, al * st_i) / wi) &&
                          std::fabs(W(g, al * st_j) / wu - W(gj, al * st_j) / wj) <= wtol * (W(gj, al * st_j) / wj)))
                      return false;
                  }
              }
            return true;
          };
          auto rect_ok = [&](const int *o, int si, int sj) {
            if (o[0] + si > ni || o[1] + sj > nj)
              return false;
            for (int j = 0; j < sj; ++j)
              for (int i = 0; i < si; ++i)
                {
                  const int m = atm(o[0] + i, o[1] + j);
                  if (m < 0 || ftaken[(size_t)m])
                    return false;
                }
            for (int j = 0; j < sj; ++j)
              for (int i = 0; i < si; ++i)
                if (!fits(atm(o[0] + i, o[1] + j), o))
                  return false;
            return true;
          };
          for (int j0 = 0; j0 < nj; ++j0)
            for (int i0 = 0; i0 < ni; ++i0)
              {
                const int m0 = atm(i0, j0);
                if (m0 < 0 || ftaken[(size_t)m0])
                  continue;
                const int o[2] = {i0, j0};
                int bi = 1, bj = 1;
                for (int sj = mf; sj >= 1; --sj)
                  for (int si = mf; si >= 1; --si)
                    if (si * sj > bi * bj && rect_ok(o, si, sj))
                      bi = si, bj = sj;
                TermsMerged::Sf f;
                const int go = at(i0, j0);
                f.pb = K.run_ap[t] + go * gsz;
                f.c = c, f.pos = r0.pos, f.fj = r0.fj;
                f.ni = bi, f.nj = bj;
                for (int q = 0; q < TERMS_MI; ++q)
                  {
                    f.ivl[0][q] = q < bi ? (int32_t)((at(i0 + q, j0) - go) * gsz) : 0;
                    f.ivl[1][q] = q < bj ? (int32_t)((at(i0, j0 + q) - go) * gsz) : 0;
                  }
                for (int j = 0; j < bj; ++j)
                  for (int i = 0; i < bi; ++i)
                    ftaken[(size_t)atm(i0 + i, j0 + j)] = 1;
                out.push_back(f);
              }
        }
    }
  // summary (build_terms_tables reduces these instead of walking the lists again)
  M.n_in = ncell;
  for (size_t e = 0; e < order.size(); ++e)
    {
      M.n_in += K.run_cnt[order[e]] / gsz;
      M.nsf += (int)M.run_size(e);
      if (K.run_nbr[order[e]] >= 0)
        M.nsi += (int)M.run_size(e);
    }
  for (const auto &f : M.sfs)
    M.ivl_f = std::max(M.ivl_f, std::max(f.ni, f.nj));
  for (const auto &c : M.cells)
    for (int d = 0; d < 3; ++d)
      {
        int k = 0;
        for (int i = 0; i < TERMS_MI; ++i)
          k += c.ivl[d][i] >= 0 ? 1 : 0;
        M.ivl_c = std::max(M.ivl_c, k);
      }
}
static constexpr int PDH_TERMS_LDS_CAP = 40 * 1024; // bytes per workgroup: four resident waves per CU at least
static bool build_terms_tables(const pdh_problem *p, const Packed &K, const RowsHost &RH, int vq_n, TermsHost &T, std::string *why = nullptr)
{
  auto no = [&](const char *m) {
    if (why)
      *why = m;
    return false;
  };
  const int basis = p->basis == PDH_BASIS_AGGLODGP ? 1 : 0;
  if (p->dim != 3 || !pdh_terms_has_kind(K.n1d, basis))
    return no("term kernel: 3-D FE_DGQ(1,2) / FE_AggloDGP(1..3) only");
  if (!RH.planar_ok || RH.fq_tensor_n <= 0 || vq_n <= 0 || (int)K.own_agg.size() != K.n_owned)
    return no("term kernel: needs axis-aligned planar faces and tensor-product rules on every sub-cell and sub-face");
  const int fn = RH.fq_tensor_n;
  const int64_t gsz = (int64_t)fn * fn, m3 = (int64_t)vq_n * vq_n * vq_n;
  const size_t nruns = K.run_ap.size();
  // runs of every slot in the order of the records
  std::vector<std::vector<size_t>> order((size_t)K.n_owned);
  {
    size_t r = 0;
    for (int sl = 0; sl < K.n_owned; ++sl)
      {
        auto &idx = order[sl];
        for (; r < nruns && K.run_slot[r] == sl; ++r)
          idx.push_back(r);
        std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return K.run_blk[x] < K.run_blk[y]; });
        int nb = 0;
        for (size_t t : idx)
          {
            if (K.run_cnt[t] % gsz)
              return no("term kernel: a face is not made of whole sub-face rules");
            if (K.run_nbr[t] < 0)
              ++nb;
          }
        if (nb > 1)
          return no("term kernel: more than one boundary run on a polytope");
        const int64_t nq = K.vq_ptr[sl + 1] - K.vq_ptr[sl];
        if (nq % m3 || nq / m3 > 65535 || idx.size() > 250)
          return no("term kernel: too many cells or faces on a polytope");
        T.maxruns = std::max<int>(T.maxruns, (int)idx.size());
      }
    if (r != nruns)
      return no("run bookkeeping");
  }
  T.maxruns = std::max(T.maxruns, 1);
  // cells and sub-faces the kernel sums over: merged where they form tensor grids (merge_terms_of_slot); PDH_TERMS_MERGE=0: as given
  const char *me = getenv("PDH_TERMS_MERGE");
  const bool merge = !(me && me[0] == '0');
  const bool trace = getenv("PDH_TRACE_SETUP") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_prev = tnow();
  auto tlap = [&](const char *what) {
    if (trace)
      fprintf(stderr, "[build_terms_tables] %-28s %6.1f ms\n", what, std::chrono::duration<double, std::milli>(tnow() - t_prev).count());
    t_prev = tnow();
  };
  tlap("run order");
  std::vector<TermsMerged> MG((size_t)K.n_owned);
  host_parallel_for((size_t)K.n_owned, [&](size_t sl) { merge_terms_of_slot(p, K, RH, vq_n, fn, sl, order[sl], merge, MG[sl]); });
  if (merge && !(me && me[0] == '2'))
    { // composite rules cost every lane task of the problem 8 instead of 4 register slots: taken when they remove at least a third of
      // what is summed over - block agglomerates lose 7 / 8 of their cells and 3 / 4 of their sub-faces, METIS-like ones 44 % of their
      // cells but only 17 % of their sub-faces (23 % together) and ran 2-10 % slower merged (profiles/r04_terms_merge.txt).
      // PDH_TERMS_MERGE=2 merges whatever can be merged.
      int64_t n_in = 0, n_out = 0;
      for (const auto &m : MG)
        n_in += m.n_in, n_out += (int64_t)m.cells.size() + m.nsf;
      if (3 * n_out > 2 * n_in)
        {
          for (auto &m : MG)
            m = TermsMerged();
          host_parallel_for((size_t)K.n_owned, [&](size_t sl) { merge_terms_of_slot(p, K, RH, vq_n, fn, sl, order[sl], false, MG[sl]); });
        }
    }
  tlap("merge");
  int ivl_c = 1, ivl_f = 1;
  for (int sl = 0; sl < K.n_owned; ++sl)
    {
      const TermsMerged &M = MG[(size_t)sl];
      ivl_c = std::max(ivl_c, M.ivl_c), ivl_f = std::max(ivl_f, M.ivl_f);
      T.n_sf_out += M.nsf;
      T.n_sf_in += M.n_in - (K.vq_ptr[sl + 1] - K.vq_ptr[sl]) / m3;
      T.n_cells_in += (K.vq_ptr[sl + 1] - K.vq_ptr[sl]) / m3;
      T.n_cells_out += (int64_t)M.cells.size();
      if (M.nsf > 65535)
        return no("term kernel: too many sub-faces on a polytope");
      T.maxsf = std::max(T.maxsf, M.nsf);
      T.maxsi = std::max(T.maxsi, M.nsi);
      T.maxcell = std::max<int>(T.maxcell, (int)M.cells.size());
    }
  T.task_pts = std::max(vq_n * ivl_c, fn * ivl_f);
  tlap("maxima");
  {
    // one pass or two (pdh_terms.h: SPLIT): whichever lets more single-wave workgroups stay resident on a CU - by LDS (160 KB in
    // granules of 1280 bytes), capped by the 12 waves the kernels' registers allow; a tie goes to the single pass (fewer
    // instructions), and so does FE_DGQ(2): with 27 functions its phases are bound by VALU issue rather than by latency, and the
    // second evaluation of the bases costs more than three more waves give (grown agglomerates of the bench cells: 0.79 ms in one
    // pass at 6 waves, 0.83 in two at 9; FE_AggloDGP(3) 0.71 -> 0.59, FE_AggloDGP(2) 0.29 -> 0.24: profiles/r04_terms_split.txt)
    const int one = pdh_terms_lds_bytes(K.n1d, basis, T.maxruns, T.maxsf, T.maxsi, T.maxcell, 0, T.task_pts);
    const int two = pdh_terms_lds_bytes(K.n1d, basis, T.maxruns, T.maxsf, T.maxsi, T.maxcell, 1, T.task_pts);
    auto waves = [](int bytes) { return bytes <= 0 ? 0 : std::min(12, (int)(160 * 1024 / (((int64_t)bytes + 1279) / 1280 * 1280))); };
    const char *fs = getenv("PDH_TERMS_SPLIT"); // (diagnostics: 0 / 1 forces the form)
    T.split = fs ? (fs[0] == '1') : ((waves(two) > waves(one) && K.n <= 20) ? 1 : 0);
    if (pdh_terms_has_kind(K.n1d, basis) == 2)
      T.split = 0; // (workgroup kernel: no such form)
    T.lds_bytes = T.split ? two : one;
  }
  if (T.lds_bytes <= 0 || T.lds_bytes > PDH_TERMS_LDS_CAP)
    return no("term kernel: the tables of the largest polytope do not fit its LDS budget (moment-based kinds take over)");
  constexpr int HDR = 12, ENT = 10;
  const int REC = HDR + T.maxruns * ENT;
  auto as_d = [](long long v) {
    double d;
    std::memcpy(&d, &v, sizeof(d));
    return d;
  };
  // the sub-faces (cells) of a polytope stand at a fixed stride (maxsf, maxcell): the kernel requests them together with the record
  T.maxsf = std::max(T.maxsf, 1);
  T.maxcell = std::max(T.maxcell, 1);
  T.sf_pt.assign((size_t)K.n_owned * T.maxsf, 0);
  T.sf_info.assign((size_t)K.n_owned * T.maxsf, 0);
  T.sf_ivl.assign((size_t)K.n_owned * T.maxsf * 2 * TERMS_MI, 0);
  T.cell_ivl.assign((size_t)K.n_owned * T.maxcell * 3 * TERMS_MI, -1);
  T.meta.assign((size_t)K.n_owned * REC, 0.0);
  std::vector<char> bad((size_t)K.n_owned, 0);
  host_parallel_for((size_t)K.n_owned, [&](size_t sl) {
    double *rec = T.meta.data() + sl * REC;
    const int a = K.own_agg[sl];
    const auto &idx = order[sl];
    const TermsMerged &M = MG[sl];
    int64_t at = (int64_t)sl * T.maxsf;
    const int64_t at0 = at;
    int nsfb = 0, e = 0;
    for (size_t t : idx)
      {
        const TermsMerged::Sf *fs = M.run_begin((size_t)e);
        const int ns = (int)M.run_size((size_t)e), nb = K.run_nbr[t];
        double *en = rec + HDR + e * ENT;
        en[0] = as_d((long long)(uint32_t)(at - at0) | ((long long)ns << 32));
        en[1] = as_d((long long)K.run_blk[t]);
        en[2] = K.run_sig[t];
        for (int c = 0; c < 3; ++c)
          {
            en[3 + c] = nb >= 0 ? p->bbox[(size_t)nb * 6 + c] : 0.0;
            en[6 + c] = nb >= 0 ? 1.0 / (p->bbox[(size_t)nb * 6 + 3 + c] - p->bbox[(size_t)nb * 6 + c]) : 1.0;
          }
        if (nb < 0)
          {
            nsfb += ns;
            if (e != 0)
              bad[sl] = 1; // (the kernel takes the boundary run to be run 0)
          }
        for (int q = 0; q < ns; ++q)
          {
            const TermsMerged::Sf &f = fs[q];
            T.sf_pt[(size_t)at] = f.pb;
            T.sf_info[(size_t)at] = e | (f.c << 8) | (f.pos << 10) | (f.fj << 11) | (f.ni << 12) | (f.nj << 15);
            for (int d = 0; d < 2; ++d)
              for (int i = 0; i < TERMS_MI; ++i)
                T.sf_ivl[((size_t)at * 2 + d) * TERMS_MI + i] = f.ivl[d][i];
            ++at;
          }
        ++e;
      }
    for (size_t u = 0; u < M.cells.size(); ++u)
      for (int d = 0; d < 3; ++d)
        for (int i = 0; i < TERMS_MI; ++i)
          T.cell_ivl[((sl * T.maxcell + u) * 3 + d) * TERMS_MI + i] = M.cells[u].ivl[d][i];
    rec[0] = as_d((long long)idx.size() | ((long long)M.cells.size() << 16) | ((long long)nsfb << 32));
    for (int c = 0; c < 3; ++c)
      {
        rec[1 + c] = p->bbox[(size_t)a * 6 + c];
        rec[4 + c] = 1.0 / (p->bbox[(size_t)a * 6 + 3 + c] - p->bbox[(size_t)a * 6 + c]);
      }
    rec[7] = as_d(K.row_base[sl]);
    rec[8] = as_d(K.row_len[sl]);
    rec[9] = as_d(K.diag_L[sl]);
    rec[10] = as_d(K.vq_ptr[sl]);
    rec[11] = as_d((long long)(at - at0));
  });
  tlap("tables");
  for (char c : bad)
    if (c)
      return no("term kernel: run order");
  return true;
}

// Host-only: 1 if the row kernel (PDH_ALG_ROWS) applies to this description and row range, 0 if not (pdh_last_error(NULL)
// says why), < 0 on an invalid description.
extern "C" int pdh_check_rows(const pdh_problem *p, int32_t row_begin, int32_t row_end)
{
  Packed K;
  g_err_noctx.clear();
  const int rc = pack_problem(nullptr, p, row_begin, row_end, K);
  if (rc != PDH_OK)
    return rc;
  RowsHost R;
  std::string why;
  int vq_n = 0;
  bool tensor_only = false;
  if (build_rows_tables(p, K, R, &why) && rows_kind_applies(p, K, R, vq_n, tensor_only, &why))
    return 1;
  if (R.planar_ok && R.fq_tensor_n > 0)
    { // the term kernel (pdh_terms.h) takes the small elements on any agglomerate of Cartesian cells with tensor rules
      std::string why_t;
      TermsHost TH;
      const int vn = resolve_tensor_hint(p->vq_tensor_n, [&](int n) { return volume_rules_are_tensor(p, K, n); });
      if (vn > 0 && build_terms_tables(p, K, R, vn, TH, &why_t))
        return 1;
      if (!why_t.empty())
        why += "; " + why_t;
    }
  g_err_noctx = why;
  return 0;
}


// Host-only: 1 if the term kernels (pdh_terms.h / pdh_terms_wg.h) apply to this description and row range, 0 if not
// (pdh_last_error(NULL) says why), < 0 on an invalid description.  stats4 (may be NULL): most runs / sub-faces / interior
// sub-faces / cells of one owned polytope... and the LDS bytes of a workgroup in stats4[4].
// Term kernels of the resident problem: cells before / after merging, sub-faces before / after (pdh_terms_tables.h); zeros if another
// kernel serves the problem.
extern "C" int pdh_terms_merge_stats(pdh_ctx *ctx, int64_t *out4)
{
  if (!ctx || !out4)
    return fail(ctx, PDH_EINVAL, "ctx and out4 are required");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_terms_merge_stats called before pdh_set_problem");
  for (int i = 0; i < 4; ++i)
    out4[i] = ctx->terms_ok ? ctx->terms_merge[i] : 0;
  return PDH_OK;
}

extern "C" int pdh_check_terms(const pdh_problem *p, int32_t row_begin, int32_t row_end, int64_t *stats5)
{
  Packed K;
  g_err_noctx.clear();
  const int rc = pack_problem(nullptr, p, row_begin, row_end, K);
  if (rc != PDH_OK)
    return rc;
  RowsHost R;
  std::string why;
  (void)build_rows_tables(p, K, R, &why);
  if (!R.planar_ok)
    {
      g_err_noctx = why;
      return 0;
    }
  const int vn = resolve_tensor_hint(p->vq_tensor_n, [&](int n) { return volume_rules_are_tensor(p, K, n); });
  TermsHost TH;
  std::string why_t;
  const bool ok = build_terms_tables(p, K, R, vn, TH, &why_t);
  if (stats5)
    {
      stats5[0] = TH.maxruns, stats5[1] = TH.maxsf, stats5[2] = TH.maxsi, stats5[3] = TH.maxcell, stats5[4] = TH.lds_bytes;
    }
  if (!ok)
    g_err_noctx = why_t;
  return ok ? 1 : 0;
}

// Host-only validation (no GPU needed): runs exactly the checks of pdh_set_problem.
extern "C" int pdh_check_problem(const pdh_problem *p, int32_t row_begin, int32_t row_end, int64_t *stats)
{
  Packed K;
  g_err_noctx.clear();
  const int rc = pack_problem(nullptr, p, row_begin, row_end, K);
  if (rc == PDH_OK && stats)
    {
      stats[0] = (int64_t)K.n_owned;
      stats[1] = (int64_t)K.it_own.size();
      stats[2] = K.n_vq;
      stats[3] = K.n_ap;
      stats[4] = K.n_values;
      stats[5] = K.n;
      stats[6] = (int64_t)pdh::lds_bytes_diag(p->dim, K.n1d, K.NT);
      stats[7] = (int64_t)pdh::lds_bytes_offdiag(p->dim, K.n1d, K.NT);
    }
  return rc;
}

// Host-only: per-peer sizes of the ghost-block exchange of a description (what pdh_exchange_layout reports after
// pdh_set_problem_local in PDH_EXCHANGE_GHOST mode) - lets the multi-rank logic be checked on machines without a GPU.
extern "C" int pdh_check_exchange(const pdh_problem *p, int32_t row_begin, int32_t row_end, int n_ranks, int64_t *send_count,
                                  int64_t *recv_count)
{
  Packed K;
  g_err_noctx.clear();
  const int rc = pack_problem(nullptr, p, row_begin, row_end, K, PDH_EXCHANGE_GHOST);
  if (rc != PDH_OK)
    return rc;
  if (n_ranks < (int)K.send_count.size() || !send_count || !recv_count)
    return fail(nullptr, PDH_EINVAL, "n_ranks is smaller than the number of ranks in agg_rank, or an output is NULL");
  for (int r = 0; r < n_ranks; ++r)
    {
      send_count[r] = r < (int)K.send_count.size() ? K.send_count[r] : 0;
      recv_count[r] = r < (int)K.recv_count.size() ? K.recv_count[r] : 0;
    }
  return PDH_OK;
}

// Own-side face points of every slot (PdhDev::ap_*), built in HBM: the caller's face arrays go up as they are (each face
// once), k_pack_faces writes one SoA run per (polytope, face) with the sign of the normal, the weights and sigma resolved
// (tables Packed::pk_*).  The staging copies are released before this returns.
extern "C" hipError_t pdh_launch_pack_faces(int dim, int64_t nqf, const double *fq_x, const double *fq_n, const double *fq_w,
                                            const double *fq_w_out, int64_t n_runs, const int64_t *pk_at, const int64_t *pk_fq,
                                            const int32_t *pk_cnt, const int32_t *pk_flags, const double *pk_sig, int64_t nap,
                                            double *ap_x, double *ap_n, double *ap_wself, double *ap_wcross, double *ap_sig,
                                            hipStream_t stream);
static int pack_faces_on_device(pdh_ctx *ctx, const pdh_problem *p, const Packed &K, PdhDev &D)
{
  const int dim = p->dim;
  const int64_t nap = K.n_ap, nqf = K.nqf_src, nruns = (int64_t)K.pk_at.size();
  double *out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const size_t out_n[5] = {(size_t)dim * nap, (size_t)dim * nap, (size_t)nap, (size_t)nap, (size_t)nap};
  for (int k = 0; k < 5; ++k)
    {
      void *d = nullptr;
      PDH_HIP(ctx, hipMalloc(&d, std::max<size_t>(out_n[k], 1) * sizeof(double)));
      ctx->allocs.push_back(d);
      out[k] = static_cast<double *>(d);
    }
  D.ap_x = out[0], D.ap_n = out[1], D.ap_wself = out[2], D.ap_wcross = out[3], D.ap_sig = out[4];
  if (nruns == 0 || nap == 0)
    return PDH_OK;
  std::vector<void *> tmp;
  auto stage = [&](const void *h, size_t bytes, const void **dptr) -> hipError_t {
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, std::max<size_t>(bytes, 8));
    if (e != hipSuccess)
      return e;
    tmp.push_back(d);
    *dptr = d;
    return bytes ? hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  const void *d_x = nullptr, *d_n = nullptr, *d_w = nullptr, *d_wo = nullptr, *d_at = nullptr, *d_fq = nullptr, *d_cnt = nullptr,
             *d_fl = nullptr, *d_sg = nullptr;
  hipError_t e = hipSuccess;
  if (K.cart)
    { // the caller-order face arrays are generated here from (cell, local face) of every sub-face (pdh_cartgen.hip); JxW of side 1
      // equals JxW of side 0 on a conforming Cartesian grid (d_wo stays NULL)
      const pdh_cartesian_points *cp = K.cart;
      const int64_t nsf = nqf / ((int64_t)cp->nqf * cp->nqf);
      std::vector<long double> gx, gw;
      pdh::gauss_legendre01(cp->nqf, gx, gw);
      double nodes[PDH_MAX_N1D] = {0}, weights[PDH_MAX_N1D] = {0};
      for (int i = 0; i < cp->nqf; ++i)
        nodes[i] = (double)gx[i], weights[i] = (double)gw[i];
      const void *d_box = nullptr, *d_cell = nullptr, *d_face = nullptr;
      auto alloc = [&](size_t bytes, const void **dptr) -> hipError_t {
        void *d = nullptr;
        hipError_t e_ = hipMalloc(&d, std::max<size_t>(bytes, 8));
        if (e_ == hipSuccess)
          tmp.push_back(d), *dptr = d;
        return e_;
      };
      e = stage(cp->cell_box, (size_t)cp->n_cells * 6 * sizeof(double), &d_box);
      if (e == hipSuccess)
        e = stage(cp->fq_cell, (size_t)nsf * sizeof(int32_t), &d_cell);
      if (e == hipSuccess)
        e = stage(cp->fq_face, (size_t)nsf * sizeof(int32_t), &d_face);
      if (e == hipSuccess)
        e = alloc((size_t)dim * nqf * sizeof(double), &d_x);
      if (e == hipSuccess)
        e = alloc((size_t)dim * nqf * sizeof(double), &d_n);
      if (e == hipSuccess)
        e = alloc((size_t)nqf * sizeof(double), &d_w);
      if (e == hipSuccess)
        e = pdh_launch_gen_faces(cp->nqf, nodes, weights, (const double *)d_box, (const int32_t *)d_cell, (const int32_t *)d_face, nqf,
                                 (double *)d_x, (double *)d_n, (double *)d_w, ctx->stream);
    }
  else
    {
      e = stage(p->fq_x, (size_t)dim * nqf * sizeof(double), &d_x);
      if (e == hipSuccess)
        e = stage(p->fq_n, (size_t)dim * nqf * sizeof(double), &d_n);
      if (e == hipSuccess)
        e = stage(p->fq_w, (size_t)nqf * sizeof(double), &d_w);
      if (e == hipSuccess && p->fq_w_out)
        e = stage(p->fq_w_out, (size_t)nqf * sizeof(double), &d_wo);
    }
  if (e == hipSuccess)
    e = stage(K.pk_at.data(), K.pk_at.size() * sizeof(int64_t), &d_at);
  if (e == hipSuccess)
    e = stage(K.pk_fq.data(), K.pk_fq.size() * sizeof(int64_t), &d_fq);
  if (e == hipSuccess)
    e = stage(K.pk_cnt.data(), K.pk_cnt.size() * sizeof(int32_t), &d_cnt);
  if (e == hipSuccess)
    e = stage(K.pk_flags.data(), K.pk_flags.size() * sizeof(int32_t), &d_fl);
  if (e == hipSuccess)
    e = stage(K.pk_sig.data(), K.pk_sig.size() * sizeof(double), &d_sg);
  if (e == hipSuccess)
    e = pdh_launch_pack_faces(dim, nqf, (const double *)d_x, (const double *)d_n, (const double *)d_w, (const double *)d_wo, nruns,
                              (const int64_t *)d_at, (const int64_t *)d_fq, (const int32_t *)d_cnt, (const int32_t *)d_fl,
                              (const double *)d_sg, nap, out[0], out[1], out[2], out[3], out[4], ctx->stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize(ctx->stream);
  for (void *d : tmp)
    (void)hipFree(d);
  if (e != hipSuccess)
    return fail(ctx, PDH_EDEVICE, std::string("face repack: ") + hipGetErrorString(e));
  return PDH_OK;
}

static int set_problem_impl(pdh_ctx *ctx, const pdh_problem *p, int32_t row_begin, int32_t row_end, const pdh_cartesian_points *cart);

extern "C" int pdh_set_problem_local(pdh_ctx *ctx, const pdh_problem *p, int32_t row_begin, int32_t row_end)
{
  return set_problem_impl(ctx, p, row_begin, row_end, nullptr);
}

extern "C" int pdh_set_problem_cartesian(pdh_ctx *ctx, const pdh_problem *p, const pdh_cartesian_points *points, int32_t row_begin,
                                         int32_t row_end)
{
  if (!points)
    return fail(ctx, PDH_EINVAL, "points is NULL");
  return set_problem_impl(ctx, p, row_begin, row_end, points);
}

static int set_problem_impl(pdh_ctx *ctx, const pdh_problem *p, int32_t row_begin, int32_t row_end, const pdh_cartesian_points *cart)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (cart && ctx->exchange_mode == PDH_EXCHANGE_GHOST)
    return fail(ctx, PDH_EUNSUPPORTED, "the cartesian description runs owner-computes-rows only (no ghost-block exchange)");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_problem(ctx);
  // PDH_TRACE_SETUP=1 (diagnostics): wall time of the phases of this call on stderr
  static const bool trace = getenv("PDH_TRACE_SETUP") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!trace)
      return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[pdh_set_problem] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  lap("wait for the stream, free the old problem");
  std::unique_ptr<Packed> K_owner(new Packed);
  Packed &K = *K_owner;
  int rc = pack_problem(ctx, p, row_begin, row_end, K, ctx->exchange_mode, cart);
  if (rc != PDH_OK)
    return rc;
  lap("validate + repack (host)");

  PdhDev &D = ctx->dev;
  std::memset(&D, 0, sizeof(D));
  D.dim = p->dim;
  D.n = K.n;
  D.n1d = K.n1d;
  D.diag_first = p->diag_first ? 1 : 0;
  D.reaction_c = p->reaction_c;
  D.tab = K.tab;
  std::vector<double> bbox(p->bbox, p->bbox + (size_t)p->n_agg * 2 * p->dim);
#define PDH_UP(vec, field)                                                                         \
  if ((rc = upload(ctx, vec, &D.field)) != PDH_OK)                                                 \
    {                                                                                              \
      free_problem(ctx);                                                                           \
      return rc;                                                                                   \
    }
  PDH_UP(bbox, bbox)
  PDH_UP(K.midx, midx)
  PDH_UP(K.vq_ptr, vq_ptr)
  if (cart)
    { // volume points: generated slot by slot from the cells' boxes (pdh_cartgen.hip)
      const int64_t m3 = (int64_t)cart->nq * cart->nq * cart->nq, ngroups = K.n_vq / m3;
      std::vector<int32_t> gcell((size_t)std::max<int64_t>(ngroups, 1));
      for (int sl = 0; sl < K.n_owned; ++sl)
        {
          const int64_t g0 = K.vq_ptr[sl] / m3, g1 = K.vq_ptr[sl + 1] / m3, src = K.vq_src[sl] / m3;
          for (int64_t g = g0; g < g1; ++g)
            gcell[(size_t)g] = cart->vq_cell[src + (g - g0)];
        }
      std::vector<long double> gx, gw;
      pdh::gauss_legendre01(cart->nq, gx, gw);
      double nodes[PDH_MAX_N1D] = {0}, weights[PDH_MAX_N1D] = {0};
      for (int i = 0; i < cart->nq; ++i)
        nodes[i] = (double)gx[i], weights[i] = (double)gw[i];
      void *dx = nullptr, *dw = nullptr, *dbox = nullptr, *dgc = nullptr;
      hipError_t eg = hipMalloc(&dx, std::max<size_t>((size_t)3 * K.n_vq, 1) * sizeof(double));
      if (eg == hipSuccess)
        {
          ctx->allocs.push_back(dx);
          eg = hipMalloc(&dw, std::max<size_t>((size_t)K.n_vq, 1) * sizeof(double));
        }
      if (eg == hipSuccess)
        {
          ctx->allocs.push_back(dw);
          eg = hipMalloc(&dbox, (size_t)cart->n_cells * 6 * sizeof(double));
        }
      if (eg == hipSuccess)
        eg = hipMalloc(&dgc, gcell.size() * sizeof(int32_t));
      if (eg == hipSuccess)
        eg = hipMemcpy(dbox, cart->cell_box, (size_t)cart->n_cells * 6 * sizeof(double), hipMemcpyHostToDevice);
      if (eg == hipSuccess)
        eg = hipMemcpy(dgc, gcell.data(), gcell.size() * sizeof(int32_t), hipMemcpyHostToDevice);
      if (eg == hipSuccess)
        eg = pdh_launch_gen_volume(cart->nq, nodes, weights, (const double *)dbox, (const int32_t *)dgc, K.n_vq, (double *)dx, K.n_vq,
                                   (double *)dw, ctx->stream);
      if (eg == hipSuccess)
        eg = hipStreamSynchronize(ctx->stream);
      if (dbox)
        (void)hipFree(dbox);
      if (dgc)
        (void)hipFree(dgc);
      if (eg != hipSuccess)
        {
          free_problem(ctx);
          return fail(ctx, PDH_EDEVICE, std::string("cartesian description: ") + hipGetErrorString(eg));
        }
      D.vq_x = static_cast<const double *>(dx);
      D.vq_w = static_cast<const double *>(dw);
    }
  else if ((rc = upload_n(ctx, K.vqx_h, (size_t)p->dim * K.vq_stride_h, &D.vq_x)) != PDH_OK ||
           (rc = upload_n(ctx, K.vqw_h, (size_t)K.n_vq, &D.vq_w)) != PDH_OK)
    {
      free_problem(ctx);
      return rc;
    }
  PDH_UP(K.ap_ptr, ap_ptr)
  if ((rc = pack_faces_on_device(ctx, p, K, D)) != PDH_OK)
    {
      free_problem(ctx);
      return rc;
    }
  PDH_UP(K.own_agg, own_agg)
  PDH_UP(K.own_row, own_row)
  PDH_UP(K.row_base, row_base)
  PDH_UP(K.row_len, row_len)
  PDH_UP(K.diag_L, diag_L)
  PDH_UP(K.it_own, it_own)
  PDH_UP(K.it_nbr, it_nbr)
  PDH_UP(K.it_pbeg, it_pbeg)
  PDH_UP(K.it_pcnt, it_pcnt)
  PDH_UP(K.it_pos, it_pos)
  PDH_UP(K.it_nbr_slot, it_nbr_slot)
  PDH_UP(K.it_pos_t, it_pos_t)
#undef PDH_UP
  lap("upload");
  ctx->problem_ghost = ctx->exchange_mode == PDH_EXCHANGE_GHOST;
  ctx->n_send = K.n_send;
  ctx->n_recv = K.n_recv;
  ctx->send_count = K.send_count;
  ctx->recv_count = K.recv_count;
  ctx->n_r21 = (int)K.r21_src.size();
  ctx->n_r22 = (int)K.r22_slot.size();
  if (ctx->problem_ghost)
    {
      if ((rc = upload(ctx, K.r21_src, &ctx->d_r21_src)) != PDH_OK || (rc = upload(ctx, K.r21_dst, &ctx->d_r21_dst)) != PDH_OK ||
          (rc = upload(ctx, K.r21_rlen, &ctx->d_r21_rlen)) != PDH_OK || (rc = upload(ctx, K.r22_ptr, &ctx->d_r22_ptr)) != PDH_OK ||
          (rc = upload(ctx, K.r22_src, &ctx->d_r22_src)) != PDH_OK || (rc = upload(ctx, K.r22_slot, &ctx->d_r22_slot)) != PDH_OK)
        {
          free_problem(ctx);
          return rc;
        }
    }
  D.vq_stride = K.vq_stride_h;
  D.ap_stride = K.n_ap;
  void *dv = nullptr;
  // the send region of the ghost-block exchange sits behind the values: the kernels address it like more rows
  hipError_t e = hipMalloc(&dv, std::max<int64_t>(K.n_values + K.n_send, 1) * sizeof(double));
  if (e != hipSuccess)
    {
      free_problem(ctx);
      return fail(ctx, PDH_EDEVICE, std::string("hipMalloc(values): ") + hipGetErrorString(e));
    }
  ctx->allocs.push_back(dv);
  D.values = static_cast<double *>(dv);
  ctx->n_values = K.n_values;
  ctx->n_owned = K.n_owned;
  ctx->n_diag_slots = (int)K.own_agg.size();
  ctx->n_items = (int)K.it_own.size();
  ctx->n_vq = K.n_vq;
  ctx->n_ap = K.n_ap;
  ctx->NT = K.NT;
  ctx->LB = K.LB;
  ctx->tiled = K.tiled;
  ctx->group = K.tiled ? -1 : combo_group(p->dim, K.n1d, K.NT, K.LB);
  ctx->lds_diag = pdh::lds_bytes_diag(p->dim, K.n1d, K.NT);
  ctx->lds_off = pdh::lds_bytes_offdiag(p->dim, K.n1d, K.NT);
  lap("values allocation");
  ctx->vq_src = K.vq_src;
  {
    // (the per-point map of the packed boundary points to the caller's face points - 8 bytes per packed face point - is
    // needed by the right-hand side only: built and uploaded at its first call, ensure_ap_src)
    ctx->d_ap_src = nullptr;
    if ((rc = upload(ctx, K.vq_src, &ctx->d_vq_src)) != PDH_OK)
      {
        free_problem(ctx);
        return rc;
      }
    ctx->n_vq_caller = p->vq_ptr[p->n_agg];
    ctx->n_fq_caller = p->n_faces ? p->fq_ptr[p->n_faces] : 0;
  }
  ctx->face_runs.clear();
  ctx->face_runs.reserve(K.run_ap.size());
  for (size_t r = 0; r < K.run_ap.size(); ++r)
    ctx->face_runs.push_back({K.run_ap[r], K.run_fq[r], K.run_cnt[r], K.run_bdry[r], K.run_slot[r]});
  lap("caller-order maps");
  ctx->n_rows_owned = (int64_t)K.n_owned * K.n;
  ctx->n_agg_total = p->n_agg;
  {
    // executed work: k-steps of 4 points per chunk (64 points in k_diag for NT >= 3, else 32; 32 in k_offdiag)
    const int64_t i_sym = sched_instr_rt(K.NT, K.LB, true), i_full = sched_instr_rt(K.NT, K.LB, false);
    const int ch_d = (K.NT >= 3) ? 64 : 32, ch_o = 32;
    auto ksteps = [](int64_t npts, int ch) {
      int64_t s = (npts / ch) * (ch / 4);
      const int64_t rem = npts % ch;
      return s + (rem + 3) / 4;
    };
    int64_t kv = 0, kf = 0, ko = 0;
    for (size_t sl = 0; sl < K.own_agg.size(); ++sl)
      {
        kv += ksteps(K.vq_ptr[sl + 1] - K.vq_ptr[sl], ch_d);
        kf += ksteps(K.ap_ptr[sl + 1] - K.ap_ptr[sl], ch_d);
      }
    for (size_t it = 0; it < K.it_pcnt.size(); ++it)
      ko += ksteps(K.it_pcnt[it], ch_o);
    ctx->mfma_diag = kv * (p->dim + (p->reaction_c != 0.0 ? 1 : 0)) * i_sym + kf * 2 * i_sym;
    ctx->mfma_offdiag = ko * 2 * i_full;
    if (K.tiled)
      { // tiles ti < tj of the own block and all tiles of a coupling block are full 64 x 64 products (64 instructions per k-step), the
        // tiles ti == tj symmetric ones (the schedule of a full n = 64 block)
        const int64_t nt = (K.n + 63) / 64, i64 = sched_instr_rt(4, 4, true);
        ctx->mfma_diag = (kv * (p->dim + (p->reaction_c != 0.0 ? 1 : 0)) + kf * 2) * (64 * (nt * (nt - 1) / 2) + i64 * nt);
        ctx->mfma_offdiag = ko * 2 * 64 * nt * nt;
      }
  }
  ctx->basis = p->basis;
  ctx->d_mtab = nullptr;
  if (p->dim == 3 && K.n1d >= 2 && K.n1d <= 4)
    {
      const std::vector<double> mt = pdh::moment_tables(p->degree, p->basis);
      void *dm = nullptr;
      hipError_t em = (int)mt.size() == pdh_moment_table_doubles(K.n1d) ? hipMalloc(&dm, mt.size() * sizeof(double)) : hipErrorInvalidValue;
      if (em == hipSuccess)
        {
          ctx->allocs.push_back(dm);
          em = hipMemcpy(dm, mt.data(), mt.size() * sizeof(double), hipMemcpyHostToDevice);
        }
      if (em != hipSuccess)
        {
          free_problem(ctx);
          return fail(ctx, PDH_EDEVICE, std::string("moment tables: ") + hipGetErrorString(em));
        }
      ctx->d_mtab = static_cast<double *>(dm);
    }
  lap("values + tables");
  ctx->rows_ok = false;
  ctx->terms_ok = false;
  if (ctx->d_mtab && !ctx->problem_ghost)
    {
      RowsHost RH;
      // (cartesian description: planar axis-aligned faces and tensor rules hold by construction - and there are no host copies of
      // the points to look at: the kinds of pdh_rows.h, whose tables are made from the points, are not offered)
      const bool rows_built = cart ? false : build_rows_tables(p, K, RH);
      int vq_n_terms = -1; // (not looked at yet)
      if (cart)
        {
          RH.planar_ok = true;
          RH.fq_tensor_n = cart->nqf;
          RH.fast_j.assign(3 * K.run_ap.size(), 0); // (the generator runs the lower tangential axis fastest)
          vq_n_terms = cart->nq;
        }
      if (rows_built)
        {
          lap("row kernel: planes + records");
          PdhRows &R = ctx->rows;
          if ((rc = upload(ctx, RH.fr_ptr, &R.fr_ptr)) != PDH_OK || (rc = upload(ctx, RH.fr_pbeg, &R.fr_pbeg)) != PDH_OK ||
              (rc = upload(ctx, RH.fr_pcnt, &R.fr_pcnt)) != PDH_OK || (rc = upload(ctx, RH.fr_nbr, &R.fr_nbr)) != PDH_OK ||
              (rc = upload(ctx, RH.fr_axis, &R.fr_axis)) != PDH_OK || (rc = upload(ctx, RH.fr_blk, &R.fr_blk)) != PDH_OK ||
              (rc = upload(ctx, RH.fr_flags, &R.fr_flags)) != PDH_OK ||
              (rc = upload(ctx, RH.fr_coord, &R.fr_coord)) != PDH_OK || (rc = upload(ctx, RH.fr_sigma, &R.fr_sigma)) != PDH_OK ||
              (rc = upload(ctx, RH.fr_nsign, &R.fr_nsign)) != PDH_OK || (rc = upload(ctx, RH.meta, &R.meta)) != PDH_OK)
            {
              free_problem(ctx);
              return rc;
            }
          {
            void *dq = nullptr;
            if (hipMalloc(&dq, 64) != hipSuccess)
              {
                free_problem(ctx);
                return fail(ctx, PDH_EDEVICE, "row kernel: out of device memory");
              }
            ctx->allocs.push_back(dq);
            if (hipMemset(dq, 0, 64) != hipSuccess)
              {
                free_problem(ctx);
                return fail(ctx, PDH_EDEVICE, "row kernel: hipMemset of the work counter failed");
              }
            R.sched = static_cast<unsigned int *>(dq);
          }
          {
            void *ds = nullptr;
            const size_t nb = (size_t)std::max(K.n_owned, 1) * 16 * sizeof(long long);
            if (hipMalloc(&ds, nb) == hipSuccess)
              {
                ctx->allocs.push_back(ds);
                (void)hipMemset(ds, 0, nb);
                R.stamps = static_cast<long long *>(ds);
              }
            else
              R.stamps = nullptr;
          }
          R.m2c_scratch = nullptr;
          R.scratch_waves = 0;
          R.scratch_stride = 0;
          if (RH.multi)
            {
              // MULTI instantiation: the coupling moments of a polytope's interior entries (8 x 8 doubles each, up to 40 of
              // them) are parked between P2 and P5 in a per-wave row of this buffer instead of LDS (pdh_rows.h) - 8 waves per
              // CU at most (256 VGPRs), a few tens of MB that stay in L2 / the memory-side cache
              int cus = 256;
              (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
              const int waves = cus * 8;
              void *dm = nullptr;
              // (with tensor sub-face rules the row holds 16 + 16 factors per interior sub-face instead, pdh_rows.h: FACT)
              const size_t stride = std::max<size_t>((size_t)RH.maxf * 64, (size_t)RH.maxs * 32 + 64);
              R.scratch_stride = (int64_t)stride;
              if (hipMalloc(&dm, (size_t)waves * stride * sizeof(double)) != hipSuccess)
                {
                  free_problem(ctx);
                  return fail(ctx, PDH_EDEVICE, "row kernel: out of device memory");
                }
              ctx->allocs.push_back(dm);
              R.m2c_scratch = static_cast<double *>(dm);
              R.scratch_waves = waves;
            }
          lap("row kernel: upload");
          int vq_n = 0;
          bool tensor_only = false;
          const bool ok = rows_kind_applies(p, K, RH, vq_n, tensor_only);
          R.tensor_only = tensor_only ? 1 : 0;
          R.multi = RH.multi ? 1 : 0;
          R.maxe = RH.maxe;
          R.maxf = RH.maxf;
          R.vq_tensor_n = vq_n;
          R.fq_tensor_n = RH.fq_tensor_n;
          ctx->rows_ok = ok;
          ctx->rows_auto = true;
          vq_n_terms = vq_n;
          lap("row kernel: volume rule check");
        }
      // term kernel (pdh_terms.h): the small elements on agglomerates of Cartesian cells with tensor rules, any number of planes
      // per neighbour.  PDH_TERMS=0 (diagnostics) keeps the kinds of pdh_rows.h.
      const char *terms_e = getenv("PDH_TERMS"); // (read per call: the tests compare both kernels in one process)
      const bool terms_env = !(terms_e && terms_e[0] == '0');
      // (FE_DGQ(3) has the workgroup-per-polytope form of the term kernel, pdh_terms_wg.h: the default where it applies since the records
      // of 1-D rules and the merged cells - 1.28-1.35 ms on the bench mesh where pdh_rows.h takes 1.59-1.66, never slower on the other
      // shapes tried, profiles/r04_wg_forms.txt; PDH_TERMS_DGQ3=0 keeps pdh_rows.h for that element)
      const int terms_kind = pdh_terms_has_kind(K.n1d, p->basis == PDH_BASIS_AGGLODGP ? 1 : 0);
      const char *terms_q3 = getenv("PDH_TERMS_DGQ3");
      if ((terms_env || cart) && RH.planar_ok && RH.fq_tensor_n > 0 &&
          (terms_kind == 1 || (terms_kind == 2 && (cart || !(terms_q3 && terms_q3[0] == '0')))))
        {
          if (vq_n_terms < 0)
            vq_n_terms = resolve_tensor_hint(p->vq_tensor_n, [&](int n) { return volume_rules_are_tensor(p, K, n); });
          TermsHost TH;
          if (vq_n_terms > 0 && build_terms_tables(p, K, RH, vq_n_terms, TH))
            {
              lap("term kernel: tables (host)");
              PdhTerms &T = ctx->terms;
              if ((rc = upload(ctx, TH.meta, &T.meta)) != PDH_OK || (rc = upload(ctx, TH.sf_pt, &T.sf_pt)) != PDH_OK ||
                  (rc = upload(ctx, TH.sf_info, &T.sf_info)) != PDH_OK || (rc = upload(ctx, TH.sf_ivl, &T.sf_ivl)) != PDH_OK ||
                  (rc = upload(ctx, TH.cell_ivl, &T.cell_ivl)) != PDH_OK)
                {
                  free_problem(ctx);
                  return rc;
                }
              lap("term kernel: upload");
              T.maxruns = TH.maxruns, T.maxsf = TH.maxsf, T.maxsi = TH.maxsi, T.maxcell = TH.maxcell;
              T.vq_tensor_n = vq_n_terms, T.fq_tensor_n = RH.fq_tensor_n;
              T.lds_bytes = TH.lds_bytes;
              T.split = TH.split;
              T.task_pts = TH.task_pts;
              {
                // the 1-D rules the kernels read, gathered on the device from the point arrays (zero-filled: slots behind a rule)
                T.tpm = TH.task_pts > 4 ? 8 : 4;
                T.tstride = pdh_terms_task_doubles(T.maxsf, T.maxcell, T.tpm);
                void *dt = nullptr;
                const size_t nb = (size_t)std::max(K.n_owned, 1) * T.tstride * sizeof(double);
                hipError_t eg = hipMalloc(&dt, nb);
                if (eg == hipSuccess)
                  {
                    ctx->allocs.push_back(dt);
                    eg = hipMemsetAsync(dt, 0, nb, ctx->stream);
                  }
                if (eg == hipSuccess)
                  {
                    T.tdata = static_cast<const double *>(dt);
                    eg = pdh_launch_terms_gather(&ctx->dev, &T, static_cast<double *>(dt), K.n_owned, ctx->stream);
                  }
                if (eg == hipSuccess)
                  eg = hipStreamSynchronize(ctx->stream);
                if (eg != hipSuccess)
                  {
                    free_problem(ctx);
                    return fail(ctx, PDH_EDEVICE, std::string("term kernel: records of 1-D rules: ") + hipGetErrorString(eg));
                  }
              }
              ctx->terms_merge[0] = TH.n_cells_in, ctx->terms_merge[1] = TH.n_cells_out;
              ctx->terms_merge[2] = TH.n_sf_in, ctx->terms_merge[3] = TH.n_sf_out;
              {
                void *ds = nullptr;
                const size_t nb = (size_t)std::max(K.n_owned, 1) * 16 * sizeof(long long);
                T.stamps = nullptr;
                if (hipMalloc(&ds, nb) == hipSuccess)
                  {
                    ctx->allocs.push_back(ds);
                    (void)hipMemset(ds, 0, nb);
                    T.stamps = static_cast<long long *>(ds);
                  }
              }
              ctx->terms_ok = true;
              ctx->rows_auto = true;
              lap("term kernel: records of 1-D rules (device)");
            }
        }
    }
  if (cart && !ctx->terms_ok)
    {
      free_problem(ctx);
      return fail(ctx, PDH_EUNSUPPORTED, "cartesian description: the term kernels do not apply (a polytope's tables exceed their LDS budget, "
                                         "or the element has none): describe the problem with its points (pdh_set_problem)");
    }
  ctx->has_problem = true;
  ctx->ev_used = 0;
  K_owner.reset();
  lap("release host staging");
  return PDH_OK;
}

extern "C" int pdh_set_algorithm(pdh_ctx *ctx, int algorithm)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (algorithm != PDH_ALG_AUTO && algorithm != PDH_ALG_DIRECT && algorithm != PDH_ALG_MOMENT && algorithm != PDH_ALG_ROWS)
    return fail(ctx, PDH_EINVAL, "algorithm must be PDH_ALG_AUTO, PDH_ALG_DIRECT, PDH_ALG_MOMENT or PDH_ALG_ROWS");
  ctx->algorithm = algorithm;
  return PDH_OK;
}

extern "C" int pdh_algorithm_in_use(pdh_ctx *ctx)
{
  if (!ctx || !ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  if (ctx->use_rows())
    return PDH_ALG_ROWS;
  const bool d = ctx->use_moment(0), o = ctx->use_moment(1);
  return d && o ? PDH_ALG_MOMENT : (d || o ? PDH_ALG_MIXED : PDH_ALG_DIRECT);
}

extern "C" int pdh_rows_kernel_in_use(pdh_ctx *ctx)
{
  if (!ctx || !ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  if (!ctx->use_rows())
    return PDH_ROWS_NONE;
  if (ctx->terms_ok)
    return PDH_ROWS_TERMS;
  if (ctx->rows.multi)
    return PDH_ROWS_MULTI;
  return ctx->dev.n == 64 ? PDH_ROWS_PIECES : PDH_ROWS_STREAMED;
}

extern "C" int pdh_set_problem(pdh_ctx *ctx, const pdh_problem *p)
{
  if (!p)
    return fail(ctx, PDH_EINVAL, "problem is NULL");
  return pdh_set_problem_local(ctx, p, 0, p->n_rows);
}

extern "C" int pdh_assemble_device(pdh_ctx *ctx)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_assemble_device called before pdh_set_problem");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  pdh_launch_fn fn = ctx->tiled ? nullptr : g_launch[ctx->group];
  const int dim = ctx->dev.dim, n1d = ctx->dev.n1d, nt = ctx->NT, lb = ctx->LB;
  if (ctx->algorithm == PDH_ALG_MOMENT && !ctx->d_mtab)
    return fail(ctx, PDH_EUNSUPPORTED, "the moment form exists for 3-D bases of degree 1..3 only");
  if (ctx->algorithm == PDH_ALG_ROWS && !ctx->use_rows())
    return fail(ctx, PDH_EUNSUPPORTED, "the row kernel does not apply to the resident problem (3-D FE_DGQ / FE_AggloDGP of degree 1 .. 3 on polytopes whose faces are unions of axis-aligned planes, no exchange variant; pdh_check_rows says why)");
  if (ctx->use_rows())
    { // one launch writes everything; reported as kernel 0, kernel 1 takes no time
      hipEvent_t r0 = nullptr, r1 = nullptr, z0 = nullptr, z1 = nullptr;
      if (ctx->profiling)
        {
          r0 = ctx->next_event();
          r1 = ctx->next_event();
          z0 = ctx->next_event();
          z1 = ctx->next_event();
          if (!r0 || !r1 || !z0 || !z1)
            return fail(ctx, PDH_EDEVICE, "hipEventCreate failed");
          PDH_HIP(ctx, hipEventRecord(r0, ctx->stream));
        }
      if (ctx->terms_ok)
        PDH_HIP(ctx, pdh_launch_terms(&ctx->dev, &ctx->terms, ctx->n_owned, ctx->stream));
      else
        PDH_HIP(ctx, pdh_launch_rows(&ctx->dev, &ctx->rows, ctx->d_mtab, ctx->n_owned, ctx->stream));
      if (ctx->profiling)
        {
          PDH_HIP(ctx, hipEventRecord(r1, ctx->stream));
          PDH_HIP(ctx, hipEventRecord(z0, ctx->stream));
          PDH_HIP(ctx, hipEventRecord(z1, ctx->stream));
        }
      return PDH_OK;
    }
  hipEvent_t e0 = nullptr, e1 = nullptr, f0 = nullptr, f1 = nullptr;
  if (ctx->profiling)
    {
      e0 = ctx->next_event();
      e1 = ctx->next_event();
      f0 = ctx->next_event();
      f1 = ctx->next_event();
      if (!e0 || !e1 || !f0 || !f1)
        return fail(ctx, PDH_EDEVICE, "hipEventCreate failed");
    }
  const bool ov = ctx->overlapped();
  hipStream_t sd = ctx->stream, so = ov ? ctx->stream2 : ctx->stream;
  // launch-bound sizes: replay the captured pair (see pdh_ctx::graph_exec)
  const bool graphable = !ctx->profiling && !ov && ctx->n_values < pdh_ctx::small_values;
  if (graphable && ctx->graph_state == 1 && (ctx->graph_alg != ctx->algorithm || ctx->graph_stream != ctx->stream))
    ctx->drop_graph();
  if (graphable && ctx->graph_state == 1)
    {
      PDH_HIP(ctx, hipGraphLaunch(ctx->graph_exec, ctx->stream));
      return PDH_OK;
    }
  bool capturing = false;
  if (graphable && ctx->graph_state == 0)
    {
      if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess)
        capturing = true;
      else
        {
          (void)hipGetLastError();
          ctx->graph_state = -1;
        }
    }
  auto end_capture = [&](bool ok) {
    if (!capturing)
      return;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (ok && e == hipSuccess && g && hipGraphInstantiate(&ctx->graph_exec, g, nullptr, nullptr, 0) == hipSuccess)
      {
        ctx->graph_state = 1;
        ctx->graph_alg = ctx->algorithm;
        ctx->graph_stream = ctx->stream;
      }
    else
      {
        (void)hipGetLastError();
        ctx->graph_exec = nullptr;
        ctx->graph_state = -1;
      }
    if (g)
      (void)hipGraphDestroy(g);
  };
  if (ov)
    {
      PDH_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
      PDH_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
  // diagonal blocks
  if (ctx->profiling)
    PDH_HIP(ctx, hipEventRecord(e0, sd));
  {
    const hipError_t le = ctx->use_moment(0) ? pdh_launch_moment(n1d, 0, &ctx->dev, ctx->d_mtab, ctx->n_diag_slots, sd)
                          : ctx->tiled     ? pdh_launch_tiled(dim, n1d, ctx->dev.reaction_c != 0.0 ? 2 : 0, &ctx->dev, ctx->n_diag_slots, sd)
                                           : fn(dim, n1d, nt, lb, ctx->dev.reaction_c != 0.0 ? 2 : 0, &ctx->dev, ctx->n_diag_slots,
                                                ctx->lds_diag, sd);
    if (le != hipSuccess)
      {
        end_capture(false); // (a stream must not be left in capture mode)
        return fail(ctx, PDH_EDEVICE, std::string("diagonal-block kernel: ") + hipGetErrorString(le));
      }
  }
  if (ctx->profiling)
    PDH_HIP(ctx, hipEventRecord(e1, sd));
  // coupling blocks
  if (ctx->profiling)
    PDH_HIP(ctx, hipEventRecord(f0, so));
  {
    const hipError_t le = ctx->use_moment(1) ? pdh_launch_moment(n1d, 1, &ctx->dev, ctx->d_mtab, ctx->n_items, so)
                          : ctx->tiled     ? pdh_launch_tiled(dim, n1d, 1, &ctx->dev, ctx->n_items, so)
                                           : fn(dim, n1d, nt, lb, 1, &ctx->dev, ctx->n_items, ctx->lds_off, so);
    if (le != hipSuccess)
      {
        end_capture(false);
        return fail(ctx, PDH_EDEVICE, std::string("coupling-block kernel: ") + hipGetErrorString(le));
      }
  }
  if (ctx->profiling)
    PDH_HIP(ctx, hipEventRecord(f1, so));
  if (ov)
    {
      PDH_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
      PDH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
  if (capturing)
    { // nothing ran yet: the launches above were recorded.  Replay them now - or, if the graph could not be built, launch plainly
      end_capture(true);
      if (ctx->graph_state == 1)
        PDH_HIP(ctx, hipGraphLaunch(ctx->graph_exec, ctx->stream));
      else
        return pdh_assemble_device(ctx);
    }
  return PDH_OK;
}

extern "C" hipError_t pdh_launch_ghost_apply(const PdhDev *P, const double *recv, int n_r21, const int64_t *r21_src,
                                             const int64_t *r21_dst, const int32_t *r21_rlen, int n_r22, const int64_t *r22_ptr,
                                             const int64_t *r22_src, const int32_t *r22_slot, hipStream_t stream);

extern "C" int pdh_set_exchange_mode(pdh_ctx *ctx, int mode)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (mode != PDH_EXCHANGE_NONE && mode != PDH_EXCHANGE_GHOST)
    return fail(ctx, PDH_EINVAL, "mode must be PDH_EXCHANGE_NONE or PDH_EXCHANGE_GHOST");
  ctx->exchange_mode = mode; // takes effect at the next pdh_set_problem*
  return PDH_OK;
}

extern "C" int pdh_exchange_layout(pdh_ctx *ctx, int n_ranks, int64_t *send_count, int64_t *recv_count)
{
  if (!ctx || !ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  if (!ctx->problem_ghost)
    return fail(ctx, PDH_ESTATE, "the resident problem was not set in PDH_EXCHANGE_GHOST mode");
  if (n_ranks < (int)ctx->send_count.size() || !send_count || !recv_count)
    return fail(ctx, PDH_EINVAL, "n_ranks is smaller than the number of ranks in agg_rank, or an output is NULL");
  for (int r = 0; r < n_ranks; ++r)
    {
      send_count[r] = r < (int)ctx->send_count.size() ? ctx->send_count[r] : 0;
      recv_count[r] = r < (int)ctx->recv_count.size() ? ctx->recv_count[r] : 0;
    }
  return PDH_OK;
}

extern "C" int pdh_exchange_get_send(pdh_ctx *ctx, double *d_send)
{
  if (!ctx || !ctx->has_problem || !ctx->problem_ghost)
    return fail(ctx, PDH_ESTATE, "no problem resident in PDH_EXCHANGE_GHOST mode");
  if (ctx->n_send == 0)
    return PDH_OK;
  if (!d_send)
    return fail(ctx, PDH_EINVAL, "d_send is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipMemcpyAsync(d_send, ctx->dev.values + ctx->n_values, ctx->n_send * sizeof(double), hipMemcpyDeviceToDevice,
                              ctx->stream));
  return PDH_OK;
}

extern "C" int pdh_exchange_apply(pdh_ctx *ctx, const double *d_recv)
{
  if (!ctx || !ctx->has_problem || !ctx->problem_ghost)
    return fail(ctx, PDH_ESTATE, "no problem resident in PDH_EXCHANGE_GHOST mode");
  if (ctx->n_recv == 0)
    return PDH_OK;
  if (!d_recv)
    return fail(ctx, PDH_EINVAL, "d_recv is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, pdh_launch_ghost_apply(&ctx->dev, d_recv, ctx->n_r21, ctx->d_r21_src, ctx->d_r21_dst, ctx->d_r21_rlen, ctx->n_r22,
                                      ctx->d_r22_ptr, ctx->d_r22_src, ctx->d_r22_slot, ctx->stream));
  return PDH_OK;
}

// Run the kernels on a stream of the caller (e.g. the framework's current stream, so that collectives issued there are
// ordered with the assembly without host synchronisation).  NULL restores the context's own stream.
extern "C" int pdh_set_stream(pdh_ctx *ctx, void *stream)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
  return PDH_OK;
}

// Diagnostic (builds with -DPDHR_STAMP only): s_memtime stamps of the row kernel's phase boundaries and in-phase sums, [n_owned][16].
extern "C" int pdh_debug_rows_stamps(pdh_ctx *ctx, long long *out)
{
  const long long *src = !ctx || !ctx->has_problem ? nullptr : (ctx->terms_ok ? ctx->terms.stamps : (ctx->rows_ok ? ctx->rows.stamps : nullptr));
  if (!src || !out)
    return fail(ctx, PDH_ESTATE, "no row-kernel problem resident");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PDH_HIP(ctx, hipMemcpy(out, src, (size_t)ctx->n_owned * 16 * sizeof(long long), hipMemcpyDeviceToHost));
  return PDH_OK;
}

extern "C" int pdh_set_overlap(pdh_ctx *ctx, int enabled)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  ctx->overlap = enabled != 0;
  return PDH_OK;
}

extern "C" int pdh_synchronize(pdh_ctx *ctx)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PDH_OK;
}

extern "C" void *pdh_stream(pdh_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int pdh_assemble(pdh_ctx *ctx, double *values)
{
  if (!values)
    return fail(ctx, PDH_EINVAL, "values is NULL");
  int rc = pdh_assemble_device(ctx);
  if (rc != PDH_OK)
    return rc;
  PDH_HIP(ctx, hipMemcpyAsync(values, ctx->dev.values, ctx->n_values * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PDH_OK;
}

// Copy of the CSR values as they stand in HBM (after pdh_assemble_device / pdh_exchange_apply), without re-assembling.
extern "C" int pdh_copy_values(pdh_ctx *ctx, double *values)
{
  if (!ctx || !ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  if (!values)
    return fail(ctx, PDH_EINVAL, "values is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipMemcpyAsync(values, ctx->dev.values, ctx->n_values * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PDH_OK;
}

extern "C" hipError_t pdh_launch_checksum(const double *values, int64_t n, double *d_out4, hipStream_t stream);

// sum, sum of |.|, max |.| and number of non-finite entries of the owned rows' values as they stand in HBM
extern "C" int pdh_values_checksum(pdh_ctx *ctx, double *out4)
{
  if (!ctx || !ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  if (!out4)
    return fail(ctx, PDH_EINVAL, "out4 is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  double *d = nullptr;
  PDH_HIP(ctx, hipMalloc((void **)&d, 4 * sizeof(double)));
  hipError_t e = pdh_launch_checksum(ctx->dev.values, ctx->n_values, d, ctx->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(out4, d, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess)
    return fail(ctx, PDH_EDEVICE, std::string("pdh_values_checksum: ") + hipGetErrorString(e));
  return PDH_OK;
}

extern "C" int pdh_assemble_sip_local(pdh_ctx *ctx, const pdh_problem *p, int32_t row_begin, int32_t row_end, double *values)
{
  int rc = pdh_set_problem_local(ctx, p, row_begin, row_end);
  if (rc != PDH_OK)
    return rc;
  return pdh_assemble(ctx, values);
}

extern "C" int pdh_assemble_sip(pdh_ctx *ctx, const pdh_problem *p, double *values)
{
  if (!p)
    return fail(ctx, PDH_EINVAL, "problem is NULL");
  return pdh_assemble_sip_local(ctx, p, 0, p->n_rows, values);
}

// ---- right-hand side -------------------------------------------------------------------------------------------------
// packed boundary point -> the caller's face point (-1: interior), for the Nitsche datum; once per problem
static int ensure_ap_src(pdh_ctx *ctx)
{
  if (ctx->d_ap_src)
    return PDH_OK;
  std::vector<int64_t> ap_src((size_t)std::max<int64_t>(ctx->n_ap, 1), -1);
  host_parallel_for(ctx->face_runs.size(), [&](size_t r) {
    const auto &fr = ctx->face_runs[r];
    if (fr.boundary)
      for (int32_t t = 0; t < fr.count; ++t)
        ap_src[fr.ap_begin + t] = fr.fq_begin + t;
  });
  // the boundary points of a slot are one contiguous run (all boundary sub-faces form ONE polytopal face, reference
  // source/agglomeration_handler.cc:1575-1613): the kernel visits only that range
  std::vector<int64_t> bd((size_t)std::max(ctx->n_owned, 1) * 2, 0);
  for (const auto &fr : ctx->face_runs)
    if (fr.boundary && fr.slot >= 0 && fr.slot < ctx->n_owned)
      {
        int64_t &b = bd[(size_t)fr.slot * 2], &e = bd[(size_t)fr.slot * 2 + 1];
        if (e == b)
          b = fr.ap_begin, e = fr.ap_begin + fr.count;
        else
          b = std::min(b, fr.ap_begin), e = std::max(e, fr.ap_begin + fr.count); // (several runs: their hull; interior points in between carry no datum)
      }
  int rc = upload(ctx, bd, &ctx->d_bd_rng);
  if (rc != PDH_OK)
    return rc;
  return upload(ctx, ap_src, &ctx->d_ap_src);
}

extern "C" int pdh_assemble_rhs_device(pdh_ctx *ctx, const double *d_f_vol, const double *d_g_bdry, double *d_rhs)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_assemble_rhs called before pdh_set_problem");
  if (!d_rhs)
    return fail(ctx, PDH_EINVAL, "rhs is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  const int rc_map = ensure_ap_src(ctx);
  if (rc_map != PDH_OK)
    return rc_map;
  PDH_HIP(ctx, pdh_launch_rhs(ctx->dev.dim, ctx->dev.n1d, &ctx->dev, ctx->n_owned, d_f_vol, d_g_bdry, d_rhs, ctx->d_vq_src,
                              ctx->d_ap_src, ctx->d_bd_rng, ctx->stream));
  return PDH_OK;
}

extern "C" int pdh_assemble_rhs(pdh_ctx *ctx, const double *f_vol, const double *g_bdry, double *rhs)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_assemble_rhs called before pdh_set_problem");
  if (!rhs)
    return fail(ctx, PDH_EINVAL, "rhs is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  // the caller's samples go up as they are (caller order; the kernel indexes them through the maps made at set_problem)
  double *d_f = nullptr, *d_g = nullptr;
  double *d_rhs = static_cast<double *>(ctx->scratch_get(2, std::max<int64_t>(ctx->n_rows_owned, 1) * sizeof(double)));
  if (f_vol)
    d_f = static_cast<double *>(ctx->scratch_get(0, std::max<int64_t>(ctx->n_vq_caller, 1) * sizeof(double)));
  if (g_bdry)
    d_g = static_cast<double *>(ctx->scratch_get(1, std::max<int64_t>(ctx->n_fq_caller, 1) * sizeof(double)));
  if (!d_rhs || (f_vol && !d_f) || (g_bdry && !d_g))
    return fail(ctx, PDH_EDEVICE, "pdh_assemble_rhs: out of device memory");
  if (f_vol)
    PDH_HIP(ctx, hipMemcpyAsync(d_f, f_vol, ctx->n_vq_caller * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (g_bdry)
    PDH_HIP(ctx, hipMemcpyAsync(d_g, g_bdry, ctx->n_fq_caller * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const int rc = pdh_assemble_rhs_device(ctx, d_f, d_g, d_rhs);
  if (rc != PDH_OK)
    return rc;
  PDH_HIP(ctx, hipMemcpyAsync(rhs, d_rhs, ctx->n_rows_owned * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PDH_OK;
}

// ---- evaluation ------------------------------------------------------------------------------------------------------
extern "C" int pdh_evaluate_device(pdh_ctx *ctx, const double *d_solution, const int64_t *d_pt_ptr, const double *d_pts,
                                   int64_t n_points, double *d_u, double *d_grad)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_evaluate called before pdh_set_problem");
  if (!d_solution || !d_pt_ptr || !d_pts || !d_u || n_points < 0)
    return fail(ctx, PDH_EINVAL, "solution, pt_ptr, pts and u are required");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, pdh_launch_eval(ctx->dev.dim, ctx->dev.n1d, d_grad ? 1 : 0, &ctx->dev, ctx->n_owned, d_solution, d_pt_ptr, d_pts,
                               n_points, d_u, d_grad, 1, ctx->stream));
  return PDH_OK;
}

extern "C" int pdh_evaluate(pdh_ctx *ctx, const double *solution, const int64_t *pt_ptr, const double *pts, double *u,
                            double *grad)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_evaluate called before pdh_set_problem");
  if (!solution || !pt_ptr || !pts || !u)
    return fail(ctx, PDH_EINVAL, "solution, pt_ptr, pts and u are required");
  const int nA = ctx->n_agg_total, dim = ctx->dev.dim;
  if (pt_ptr[0] != 0)
    return fail(ctx, PDH_EINVAL, "pt_ptr[0] must be 0");
  for (int a = 0; a < nA; ++a)
    if (pt_ptr[a + 1] < pt_ptr[a])
      return fail(ctx, PDH_EINVAL, "pt_ptr must be non-decreasing");
  const int64_t N = pt_ptr[nA];
  if (N == 0)
    return PDH_OK;
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  double *d_sol = static_cast<double *>(ctx->scratch_get(0, std::max<int64_t>(ctx->n_rows_owned, 1) * sizeof(double)));
  double *d_pts = static_cast<double *>(ctx->scratch_get(1, (size_t)N * dim * sizeof(double)));
  int64_t *d_ptr = static_cast<int64_t *>(ctx->scratch_get(2, ((size_t)nA + 1) * sizeof(int64_t)));
  double *d_u = static_cast<double *>(ctx->scratch_get(3, (size_t)N * sizeof(double)));
  double *d_g = grad ? static_cast<double *>(ctx->scratch_get(4, (size_t)N * dim * sizeof(double))) : nullptr;
  if (!d_sol || !d_pts || !d_ptr || !d_u || (grad && !d_g))
    return fail(ctx, PDH_EDEVICE, "pdh_evaluate: out of device memory");
  PDH_HIP(ctx, hipMemcpyAsync(d_sol, solution, ctx->n_rows_owned * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_pts, pts, (size_t)N * dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_ptr, pt_ptr, ((size_t)nA + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  const int rc = pdh_evaluate_device(ctx, d_sol, d_ptr, d_pts, N, d_u, d_g);
  if (rc != PDH_OK)
    return rc;
  // only the points of polytopes owned here are produced; the others are left untouched in the caller's arrays
  std::vector<double> hu((size_t)N), hg(grad ? (size_t)N * dim : 0);
  PDH_HIP(ctx, hipMemcpyAsync(hu.data(), d_u, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (grad)
    PDH_HIP(ctx, hipMemcpyAsync(hg.data(), d_g, (size_t)N * dim * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  std::vector<int32_t> own((size_t)ctx->n_owned);
  PDH_HIP(ctx, hipMemcpyAsync(own.data(), ctx->dev.own_agg, own.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int a : own)
    for (int64_t q = pt_ptr[a]; q < pt_ptr[a + 1]; ++q)
      {
        u[q] = hu[q];
        if (grad)
          for (int c = 0; c < dim; ++c)
            grad[(size_t)c * N + q] = hg[(size_t)c * N + q];
      }
  return PDH_OK;
}

// ---- PolyUtils::compute_global_error fused on the device (reference include/poly_utils.h:1647-1750) -----------------------
extern "C" hipError_t pdh_launch_eval_err(int dim, int n1d, const PdhDev *P, int count, const double *coef, const int64_t *pt_ptr,
                                          const double *pts, int64_t pts_stride, const double *w, const double *exact_u,
                                          const double *exact_g, double *err, hipStream_t stream);
extern "C" int pdh_global_error_device(pdh_ctx *ctx, const double *d_solution, const int64_t *d_pt_ptr, const double *d_pts,
                                       int64_t n_points, const double *d_w, const double *d_exact_u, const double *d_exact_grad,
                                       double *sums)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_global_error called before pdh_set_problem");
  if (!d_solution || !d_pt_ptr || !d_pts || !d_w || !d_exact_u || !d_exact_grad || !sums || n_points < 0)
    return fail(ctx, PDH_EINVAL, "solution, pt_ptr, pts, w, exact_u, exact_grad and sums are required");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  sums[0] = sums[1] = 0.0;
  if (ctx->n_owned == 0)
    return PDH_OK;
  double *d_err = static_cast<double *>(ctx->scratch_get(5, (size_t)ctx->n_owned * 2 * sizeof(double)));
  if (!d_err)
    return fail(ctx, PDH_EDEVICE, "pdh_global_error: out of device memory");
  PDH_HIP(ctx, pdh_launch_eval_err(ctx->dev.dim, ctx->dev.n1d, &ctx->dev, ctx->n_owned, d_solution, d_pt_ptr, d_pts, n_points, d_w,
                                   d_exact_u, d_exact_grad, d_err, ctx->stream));
  // 16 bytes per polytope come back; they are added in slot order (the result does not depend on the launch)
  std::vector<double> h((size_t)ctx->n_owned * 2);
  PDH_HIP(ctx, hipMemcpyAsync(h.data(), d_err, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int sl = 0; sl < ctx->n_owned; ++sl)
    {
      sums[0] += h[2 * (size_t)sl];
      sums[1] += h[2 * (size_t)sl + 1];
    }
  return PDH_OK;
}

extern "C" int pdh_global_error(pdh_ctx *ctx, const double *solution, const int64_t *pt_ptr, const double *pts, const double *w,
                                const double *exact_u, const double *exact_grad, double *sums)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (!ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "pdh_global_error called before pdh_set_problem");
  if (!solution || !pt_ptr || !pts || !w || !exact_u || !exact_grad || !sums)
    return fail(ctx, PDH_EINVAL, "solution, pt_ptr, pts, w, exact_u, exact_grad and sums are required");
  const int nA = ctx->n_agg_total, dim = ctx->dev.dim;
  if (pt_ptr[0] != 0)
    return fail(ctx, PDH_EINVAL, "pt_ptr[0] must be 0");
  for (int a = 0; a < nA; ++a)
    if (pt_ptr[a + 1] < pt_ptr[a])
      return fail(ctx, PDH_EINVAL, "pt_ptr must be non-decreasing");
  const int64_t N = pt_ptr[nA];
  sums[0] = sums[1] = 0.0;
  if (N == 0)
    return PDH_OK;
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  double *d_sol = static_cast<double *>(ctx->scratch_get(0, std::max<int64_t>(ctx->n_rows_owned, 1) * sizeof(double)));
  double *d_pts = static_cast<double *>(ctx->scratch_get(1, (size_t)N * dim * sizeof(double)));
  int64_t *d_ptr = static_cast<int64_t *>(ctx->scratch_get(2, ((size_t)nA + 1) * sizeof(int64_t)));
  double *d_eu = static_cast<double *>(ctx->scratch_get(3, (size_t)N * 2 * sizeof(double))); // exact_u | w
  double *d_eg = static_cast<double *>(ctx->scratch_get(4, (size_t)N * dim * sizeof(double)));
  if (!d_sol || !d_pts || !d_ptr || !d_eu || !d_eg)
    return fail(ctx, PDH_EDEVICE, "pdh_global_error: out of device memory");
  PDH_HIP(ctx, hipMemcpyAsync(d_sol, solution, ctx->n_rows_owned * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_pts, pts, (size_t)N * dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_ptr, pt_ptr, ((size_t)nA + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_eu, exact_u, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_eu + N, w, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_eg, exact_grad, (size_t)N * dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return pdh_global_error_device(ctx, d_sol, d_ptr, d_pts, N, d_eu + N, d_eu, d_eg, sums);
}

// ---- basis values on boxes (injection matrices) -----------------------------------------------------------------------
extern "C" int pdh_shape_values_device(pdh_ctx *ctx, int dim, int degree, int basis, int n_boxes, const double *d_bbox,
                                       const int64_t *d_pt_ptr, const double *d_pts, int64_t n_points, double *d_values)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (dim < 2 || dim > 3 || degree < 0 || (basis != PDH_BASIS_DGQ && basis != PDH_BASIS_AGGLODGP))
    return fail(ctx, PDH_EINVAL, "dim must be 2 or 3, degree >= 0, basis DGQ or AGGLODGP");
  const int n = pdh::n_dofs_per_cell(dim, degree, basis);
  const int n1d = degree + 1;
  if (n1d > 8 || (dim == 2 && n > 64))
    return fail(ctx, PDH_EUNSUPPORTED, "no kernel instantiated for this (dim, basis, degree)");
  if (n_boxes <= 0 || n_points <= 0)
    return PDH_OK;
  if (!d_bbox || !d_pt_ptr || !d_pts || !d_values)
    return fail(ctx, PDH_EINVAL, "bbox, pt_ptr, pts and values are required");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  const int key = (dim * 16 + degree) * 2 + basis;
  if (ctx->shape_key != key)
    {
      const auto mi = pdh::multi_indices(dim, degree, basis);
      std::vector<int32_t> midx(512, (int32_t)0xffffffffu); // (n <= 8^3)
      for (int i = 0; i < n; ++i)
        midx[i] = (int32_t)mi[i];
      if (!ctx->d_shape_midx)
        PDH_HIP(ctx, hipMalloc((void **)&ctx->d_shape_midx, 512 * sizeof(int32_t)));
      PDH_HIP(ctx, hipMemcpy(ctx->d_shape_midx, midx.data(), 512 * sizeof(int32_t), hipMemcpyHostToDevice));
      ctx->shape_key = key;
    }
  PdhDev D;
  std::memset(&D, 0, sizeof(D));
  D.dim = dim;
  D.n = n;
  D.n1d = n1d;
  const pdh::Basis1D b1 = (basis == PDH_BASIS_DGQ) ? pdh::lagrange_basis(degree) : pdh::legendre_basis(degree);
  for (int k = 0; k < n1d; ++k)
    for (int m = 0; m < n1d; ++m)
      D.tab.coef[k][m] = (double)b1.coef[k][m];
  D.bbox = d_bbox;
  D.midx = ctx->d_shape_midx;
  PDH_HIP(ctx, pdh_launch_shape(dim, n1d, &D, n_boxes, d_pt_ptr, d_pts, n_points, d_values, ctx->stream));
  return PDH_OK;
}

extern "C" int pdh_shape_values(pdh_ctx *ctx, int dim, int degree, int basis, int n_boxes, const double *bbox,
                                const int64_t *pt_ptr, const double *pts, double *values)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  if (dim < 2 || dim > 3 || degree < 0 || (basis != PDH_BASIS_DGQ && basis != PDH_BASIS_AGGLODGP))
    return fail(ctx, PDH_EINVAL, "dim must be 2 or 3, degree >= 0, basis DGQ or AGGLODGP");
  if (n_boxes < 0 || (n_boxes > 0 && (!bbox || !pt_ptr || !pts || !values)))
    return fail(ctx, PDH_EINVAL, "bbox, pt_ptr, pts and values are required");
  const int n = pdh::n_dofs_per_cell(dim, degree, basis);
  if (degree + 1 > 8 || (dim == 2 && n > 64))
    return fail(ctx, PDH_EUNSUPPORTED, "no kernel instantiated for this (dim, basis, degree)");
  if (n_boxes == 0)
    return PDH_OK;
  for (int b = 0; b < n_boxes; ++b)
    {
      if (pt_ptr[b + 1] < pt_ptr[b])
        return fail(ctx, PDH_EINVAL, "pt_ptr must be non-decreasing");
      for (int c = 0; c < dim; ++c)
        if (!(bbox[(size_t)b * 2 * dim + dim + c] > bbox[(size_t)b * 2 * dim + c]))
          return fail(ctx, PDH_EINVAL, "degenerate bounding box");
    }
  if (pt_ptr[0] != 0)
    return fail(ctx, PDH_EINVAL, "pt_ptr[0] must be 0");
  const int64_t N = pt_ptr[n_boxes];
  if (N == 0)
    return PDH_OK;
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  double *d_bbox = static_cast<double *>(ctx->scratch_get(0, (size_t)n_boxes * 2 * dim * sizeof(double)));
  double *d_pts = static_cast<double *>(ctx->scratch_get(1, (size_t)N * dim * sizeof(double)));
  int64_t *d_ptr = static_cast<int64_t *>(ctx->scratch_get(2, ((size_t)n_boxes + 1) * sizeof(int64_t)));
  double *d_out = static_cast<double *>(ctx->scratch_get(3, (size_t)N * n * sizeof(double)));
  if (!d_bbox || !d_pts || !d_ptr || !d_out)
    return fail(ctx, PDH_EDEVICE, "pdh_shape_values: out of device memory");
  PDH_HIP(ctx, hipMemcpyAsync(d_bbox, bbox, (size_t)n_boxes * 2 * dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_pts, pts, (size_t)N * dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  PDH_HIP(ctx, hipMemcpyAsync(d_ptr, pt_ptr, ((size_t)n_boxes + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  const int rc = pdh_shape_values_device(ctx, dim, degree, basis, n_boxes, d_bbox, d_ptr, d_pts, N, d_out);
  if (rc != PDH_OK)
    return rc;
  PDH_HIP(ctx, hipMemcpyAsync(values, d_out, (size_t)N * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PDH_OK;
}

extern "C" int pdh_device_values(pdh_ctx *ctx, double **device_ptr, int64_t *n_values)
{
  if (!ctx || !ctx->has_problem)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  if (device_ptr)
    *device_ptr = ctx->dev.values;
  if (n_values)
    *n_values = ctx->n_values;
  return PDH_OK;
}

extern "C" int pdh_set_profiling(pdh_ctx *ctx, int enabled)
{
  if (!ctx)
    return fail(nullptr, PDH_EINVAL, "ctx is NULL");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->profiling = enabled != 0;
  ctx->ev_used = 0;
  return PDH_OK;
}

extern "C" int pdh_kernel_times_ms(pdh_ctx *ctx, float *ms, int *n_launches)
{
  if (!ctx || !ms)
    return fail(ctx, PDH_EINVAL, "ctx or ms is NULL");
  const size_t nl = ctx->ev_used / 4;
  if (nl == 0)
    return fail(ctx, PDH_ESTATE, "no profiled launch recorded (pdh_set_profiling(1) then pdh_assemble_device)");
  PDH_HIP(ctx, hipSetDevice(ctx->device));
  PDH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double sum[PDH_N_KERNELS] = {0.0, 0.0};
  for (size_t l = 0; l < nl; ++l)
    for (int k = 0; k < PDH_N_KERNELS; ++k)
      {
        float t = 0.f;
        PDH_HIP(ctx, hipEventElapsedTime(&t, ctx->events[4 * l + 2 * k], ctx->events[4 * l + 2 * k + 1]));
        sum[k] += t;
      }
  for (int k = 0; k < PDH_N_KERNELS; ++k)
    ms[k] = (float)(sum[k] / (double)nl);
  if (n_launches)
    *n_launches = (int)nl;
  return PDH_OK;
}

extern "C" int pdh_kernel_work(pdh_ctx *ctx, int64_t *mfma_instr)
{
  if (!ctx || !ctx->has_problem || !mfma_instr)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  mfma_instr[0] = (ctx->use_moment(0) || ctx->use_rows()) ? 0 : ctx->mfma_diag; // counted for the direct form only
  mfma_instr[1] = (ctx->use_moment(1) || ctx->use_rows()) ? 0 : ctx->mfma_offdiag;
  return PDH_OK;
}

extern "C" int pdh_problem_stats(pdh_ctx *ctx, int64_t *stats)
{
  if (!ctx || !ctx->has_problem || !stats)
    return fail(ctx, PDH_ESTATE, "no problem resident");
  stats[0] = ctx->n_owned;
  stats[1] = ctx->n_items;
  stats[2] = ctx->n_vq;
  stats[3] = ctx->n_ap;
  stats[4] = ctx->n_values;
  stats[5] = ctx->dev.n;
  stats[6] = (int64_t)ctx->lds_diag;
  stats[7] = (int64_t)ctx->lds_off;
  return PDH_OK;
}
