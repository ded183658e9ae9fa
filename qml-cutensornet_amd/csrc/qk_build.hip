// qk_build.hip -- device MPS builder: the input producer of the Gram path on the GPU.
// Replaces simulate(libhandle, circ, SimulationAlgorithm.MPSxGate, config) of the reference
// (gpu_backend/kernel_state_ansatz.py:141-144, 221, 263; truncation criterion as KernelPkg.jl:68) for the ansatz gate
// program (H, Rz, XXPhase, SWAP on adjacent qubits: qml-cutensornet_amd/ansatz.py).  Same algorithm as the host
// builder (csrc/qk_builder.cpp, mps.py:_simulate) -- orthogonality centre carried along, one SVD per two-qubit gate,
// fewest singular values whose weight keeps the fidelity -- restated for the device:
//   * all data points share ONE gate structure and differ only in the angles, so the whole list is one persistent
//     launch: a workgroup pulls a state index from a device counter and runs that state's complete gate program;
//   * the only dense factorisation is a one-sided (Hestenes) Jacobi sweep over column pairs, GL = 8 lanes per pair and
//     BT / GL = 32 pairs per step in round-robin order (256-thread workgroups): it serves as the SVD of a gate's theta
//     matrix and, with the same code, as the rank-revealing orthogonalisation of a centre move (M = (W/s)(s V^H)
//     instead of QR).  When A and V fit they are factorised in LDS (odd leading dimension: conflict-free for the 16-byte
//     elements), otherwise from the L2-resident workspace;
//   * site tensors live in a per-workgroup arena (fixed slots of 2*cap^2 complex), theta / V / temporaries in a
//     per-workgroup workspace -- L2-resident at the bonds of the reference's workloads; finished states are packed
//     into one heap (atomic bump) and described by dims / offsets / fidelity arrays.
// Where it stands: wins over the 16-core host pool at bonds <= 33 with hundreds of states, loses
// beyond bond ~64 (a block Jacobi on the matrix cores is the missing piece); build_kernel_matrix uses it only where no
// state can outgrow its bond cap (QK_BUILDER=auto).
#include "qk_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
typedef double cd __attribute__((ext_vector_type(2)));  // complex128 as (re, im); a native vector so that LDS-typed pointers work

constexpr int MAX_SWEEPS = 40;

enum { OP_H = 0, OP_RZ = 1, OP_XX = 2, OP_SWAP = 3 };  // ansatz.py

enum { ERR_BOND = 1, ERR_HEAP = 2, ERR_SWEEPS = 4, ERR_GATE = 8 };

struct BuildArgs {
  int n_states, n_qubits, n_ops, cap;
  const int8_t* op;
  const int32_t* q0;
  const double* alpha;  // [n_states][n_ops] half-turns
  double budget, zero;
  cd* arena;  // per workgroup: n_qubits slots of 2 cap^2
  cd* work;   // per workgroup: 4 buffers of (2 cap + 32)^2
  int block;  // the preconditioned block Jacobi for factorisations beyond the LDS working set (QK_BUILD_BLOCK=0: scalar kernel)
  cd* heap;
  unsigned long long heap_cap;
  unsigned long long* heap_top;
  int32_t* dims_out;    // [n_states][n_qubits + 1]
  double* fid_out;      // [n_states]
  double* secs_out;     // [n_states] seconds of workgroup time the state took (device clock)
  long long* offs_out;  // [n_states] complex elements into heap
  unsigned long long* counter;
  int* error;     // [0] error bits, [1..4] Jacobi statistics: factorisations, sweeps, most sweeps, unconverged
  int jl_offset;  // doubles from the start of the dynamic LDS to the Jacobi working set
  int jl_elems;   // complex elements it holds
  int partial;    // a state that outgrows cap is dropped (fidelity -1) instead of failing the call
  int truncate;   // bonds are cut at cap (the chi of pytket-cutensornet's Config) instead
  const int32_t* order;  // queue position -> state index: states expected to be expensive first
};


// Shared scalars of a workgroup (one instance in LDS).
struct WgShared {
  int flag, keep, state, pad;
  double frac, nrm;
  unsigned long long off;
  unsigned long long worst;  // bits of the largest squared relative inner product rotated in the current sweep
};


// Built states -> the Gram engine's set image: site (s, k) of the heap ([l][2][r] complex, interleaved) becomes two planes
// [pad16(l)][2][pad16(r)] (re, im) in a zero-initialised allocation (the layout of qk_pack_state, qkgram.hip).
__global__ __launch_bounds__(256) void qk_pack_built_kernel(const cd* __restrict__ heap, const long long* __restrict__ src_offs,
                                                            const long long* __restrict__ dst_offs, const int32_t* __restrict__ dims_true,
                                                            const int32_t* __restrict__ dims_pad, int n_sites, double* __restrict__ data) {
  const int s = blockIdx.x / n_sites, k = blockIdx.x - s * n_sites;
  const int cl = dims_true[(long)s * (n_sites + 1) + k], cr = dims_true[(long)s * (n_sites + 1) + k + 1];
  const int pl = dims_pad[(long)s * (n_sites + 1) + k], pr = dims_pad[(long)s * (n_sites + 1) + k + 1];
  const cd* src = heap + src_offs[blockIdx.x];
  double* re = data + dst_offs[blockIdx.x];
  double* im = re + (long)pl * 2 * pr;
  for (int e = threadIdx.x; e < cl * 2 * cr; e += 256) {
    const int row = e / cr, c = e - row * cr;
    const cd v = src[e];
    re[(long)row * pr + c] = v.x;
    im[(long)row * pr + c] = v.y;
  }
}


}  // namespace

#define QKB_NS qkb256
#define QK_BUILD_BT 256
#include "qk_build_kernels.h"
#undef QKB_NS
#undef QK_BUILD_BT
#define QKB_NS qkb512
#define QK_BUILD_BT 512
#include "qk_build_kernels.h"
#undef QKB_NS
#undef QK_BUILD_BT

struct qk_built {
  qk_ctx* ctx = nullptr;
  int n_states = 0, n_qubits = 0;
  cd* heap = nullptr;
  std::vector<int32_t> dims;
  std::vector<double> fidelity;
  std::vector<double> secs;  // workgroup time per state
  std::vector<int64_t> offsets;
  int64_t total = 0;
  double kernel_ms = 0;
};

extern "C" int qk_build_mps(qk_ctx* c, int32_t n_states, int32_t n_qubits, int32_t n_ops, const int8_t* op, const int32_t* q0,
                            const double* alpha, double trunc_budget, double value_of_zero, int32_t max_bond, uint32_t flags, qk_built** out) {
  if (!c || !op || !q0 || !alpha || !out) return qk_fail(QK_EINVAL, "qk_build_mps: null argument");
  if (n_states <= 0 || n_qubits <= 0 || n_ops < 0) return qk_fail(QK_EINVAL, "qk_build_mps: empty problem (%d states, %d qubits, %d gates)", n_states, n_qubits, n_ops);
  if (max_bond < 2 || max_bond > 1024) return qk_fail(QK_EINVAL, "qk_build_mps: max_bond %d outside 2..1024", max_bond);
  *out = nullptr;
  QkRangeGuard range_("qk:build");
  HIP_TRY(hipSetDevice(c->device));
  const int cap = max_bond;
  size_t lds_meta = (size_t)2 * cap * sizeof(double) + (size_t)2 * cap * sizeof(int) + (size_t)(n_qubits + 1) * sizeof(int);
  lds_meta = (lds_meta + 15) / 16 * 16;
  if (lds_meta > 24 * 1024) return qk_fail(QK_EINVAL, "qk_build_mps: %d qubits at max_bond %d need %zu bytes of LDS", n_qubits, cap, lds_meta);
  // Two workgroups per CU with 76 KiB of LDS each (what the bookkeeping leaves is the Jacobi working set: A and V of a
  // factorisation up to ~(p + q) q = 4500 complex numbers, e.g. 74 x 37; larger ones run from L2) -- or, when the caller
  // bounds the bonds by 32, four with 38 KiB and half the registers each: more latency hiding for small factorisations
  // (cfg5-shaped: 8.0 instead of 10.4 s), worse as soon as many of them spill to the L2 path.  QK_BUILD_WGS=2|4 overrides.
  // Workgroup shape: 256 threads, four workgroups per CU (bonds <= 32) or two -- or, for bonds beyond 64, 512 threads and ONE
  // workgroup per CU with 152 KiB of LDS (qk_build_kernels.h): a heterogeneous data set ends with its few heaviest states, whose
  // block factorisations run one visit per wavefront.  QK_BUILD_WGS=1|2|4 overrides.
  int wgs_variant = (cap <= 32) ? 4 : (cap <= 64 ? 2 : 1);
  // a share with a workgroup slot per state even in the 512-thread shape (<= one state per CU) ends with its slowest state either
  // way, and a state is built faster by 512 threads with the whole CU's LDS: 96 states of 100 qubits x 10 layers cut at bond 64,
  // 60.7 s in the 256-thread shape, 46.8 s in this one (profiles/r03/builder_capped_cfg5_gamma0.5_chi64.txt)
  if (wgs_variant == 2 && n_states <= c->num_cus) wgs_variant = 1;
  if (const char* v = std::getenv("QK_BUILD_WGS")) wgs_variant = (std::atoi(v) >= 4) ? 4 : (std::atoi(v) <= 1 ? 1 : 2);
  const int bt = wgs_variant == 1 ? 512 : 256;
  size_t lds_total = (wgs_variant == 4 ? 38 : wgs_variant == 2 ? 76 : 152) * 1024;
  if (const char* v = std::getenv("QK_BUILD_LDS_KB")) lds_total = (size_t)std::max(32, std::min(wgs_variant == 4 ? 38 : 156, std::atoi(v))) * 1024;
  const int jl_elems = (int)((lds_total - lds_meta) / sizeof(cd));
  const size_t lds = lds_total;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qkb256::qk_build_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qkb256::qk_build_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 38 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qkb512::qk_build_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  const int wgs_per_cu = std::min(wgs_variant, (int)((160 * 1024) / (lds_total + 1024)));  // + the static LDS of the kernel
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t wslot = (size_t)(2 * cap + 32) * (2 * cap + 32);
  const size_t wtab = (((size_t)(2 * cap + 32) / 8) * ((size_t)(2 * cap + 32) / 8) + 3) / 4;
  const size_t per_wg = ((size_t)n_qubits * 2 * cap * cap + 4 * wslot + wtab) * sizeof(cd);
  long long grid = std::min<long long>(n_states, (long long)wgs_per_cu * c->num_cus);
  grid = std::min<long long>(grid, (long long)(0.35 * (double)free_b / (double)per_wg));
  if (grid < 1) return qk_fail(QK_EDEVICE, "qk_build_mps: not enough device memory for one workgroup's arena (%zu bytes)", per_wg);
  // heap: every finished state, packed; bounded by the arena size of all states and by the free memory
  // (worst case = every bond at the cap; real data sets need a few per cent of that, and allocating -- and freeing -- a hundred GB
  // costs seconds: a twelfth of the free memory unless QK_BUILD_HEAP_GB says otherwise; a heap that turns out too small fails loudly)
  const double heap_want = (double)n_states * (double)n_qubits * 2.0 * cap * cap;
  double heap_lim = 0.08 * (double)free_b / (double)sizeof(cd);
  if (const char* v = std::getenv("QK_BUILD_HEAP_GB")) heap_lim = std::min(0.45 * (double)free_b, std::atof(v) * 1073741824.0) / (double)sizeof(cd);
  const size_t heap_cap = (size_t)std::max(1024.0, std::min(heap_want, heap_lim));
  const double t_host0 = (double)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-6;
  cd *arena = nullptr, *work = nullptr, *heap = nullptr;
  int8_t* d_op = nullptr;
  int32_t* d_q0 = nullptr;
  double *d_alpha = nullptr, *d_fid = nullptr, *d_secs = nullptr;
  int32_t* d_dims = nullptr;
  long long* d_offs = nullptr;
  unsigned long long* d_ctr = nullptr;  // [0] state counter, [1] heap top
  int* d_err = nullptr;
  int32_t* d_order = nullptr;
  auto release = [&]() {
    (void)hipFree(d_op), (void)hipFree(d_q0), (void)hipFree(d_alpha), (void)hipFree(d_fid), (void)hipFree(d_secs);
    (void)hipFree(d_dims), (void)hipFree(d_offs), (void)hipFree(d_ctr), (void)hipFree(d_err), (void)hipFree(d_order);
  };
#define BUILD_TRY(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      release();                                                                                          \
      (void)hipFree(heap);                                                                                \
      return qk_fail(QK_EDEVICE, "qk_build_mps: %s failed: %s", #expr, hipGetErrorString(e_));            \
    }                                                                                                     \
  } while (0)
  // the per-workgroup arena and workspace (tens of GB at large bond caps: allocating and releasing them costs seconds) stay with
  // the context between calls -- build_kernel_matrix builds the X and the Y share one after the other -- and go with it
  {
    const size_t arena_b = (size_t)grid * n_qubits * 2 * cap * cap * sizeof(cd), work_b = (size_t)grid * (4 * wslot + wtab) * sizeof(cd);
    if (c->build_arena_bytes < arena_b) {
      if (c->build_arena) (void)hipFree(c->build_arena);
      c->build_arena = nullptr, c->build_arena_bytes = 0;
      BUILD_TRY(hipMalloc(&c->build_arena, arena_b));
      c->build_arena_bytes = arena_b;
    }
    if (c->build_work_bytes < work_b) {
      if (c->build_work) (void)hipFree(c->build_work);
      c->build_work = nullptr, c->build_work_bytes = 0;
      BUILD_TRY(hipMalloc(&c->build_work, work_b));
      c->build_work_bytes = work_b;
    }
    arena = static_cast<cd*>(c->build_arena), work = static_cast<cd*>(c->build_work);
  }
  BUILD_TRY(hipMalloc(&heap, heap_cap * sizeof(cd)));
  BUILD_TRY(hipMalloc(&d_op, std::max(1, n_ops)));
  BUILD_TRY(hipMalloc(&d_q0, (size_t)std::max(1, n_ops) * sizeof(int32_t)));
  BUILD_TRY(hipMalloc(&d_alpha, (size_t)n_states * std::max(1, n_ops) * sizeof(double)));
  BUILD_TRY(hipMalloc(&d_fid, (size_t)n_states * sizeof(double)));
  BUILD_TRY(hipMalloc(&d_secs, (size_t)n_states * sizeof(double)));
  BUILD_TRY(hipMalloc(&d_dims, (size_t)n_states * (n_qubits + 1) * sizeof(int32_t)));
  BUILD_TRY(hipMalloc(&d_offs, (size_t)n_states * sizeof(long long)));
  BUILD_TRY(hipMalloc(&d_ctr, 2 * sizeof(unsigned long long)));
  BUILD_TRY(hipMalloc(&d_err, 32 * sizeof(int)));
  if (n_ops > 0) {
    BUILD_TRY(hipMemcpyAsync(d_op, op, n_ops, hipMemcpyHostToDevice, c->stream));
    BUILD_TRY(hipMemcpyAsync(d_q0, q0, (size_t)n_ops * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    BUILD_TRY(hipMemcpyAsync(d_alpha, alpha, (size_t)n_states * n_ops * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  BUILD_TRY(hipMemsetAsync(d_ctr, 0, 2 * sizeof(unsigned long long), c->stream));
  BUILD_TRY(hipMemsetAsync(d_err, 0, 32 * sizeof(int), c->stream));
  // Queue order: longest expected first.  The cost of a state grows with its bonds, and those with the entangling power
  // of its XXPhase gates, sin^2(pi alpha) summed over the gates -- a cheap proxy that keeps the tail of the launch short.
  std::vector<int32_t> order(n_states);
  {
    std::vector<double> proxy(n_states, 0.0);
    for (int s = 0; s < n_states; ++s)
      for (int i = 0; i < n_ops; ++i)
        if (op[i] == OP_XX) {
          const double sn = std::sin(M_PI * alpha[(size_t)s * n_ops + i]);
          proxy[s] += sn * sn;
        }
    for (int s = 0; s < n_states; ++s) order[s] = s;
    if (!std::getenv("QK_BUILD_NO_ORDER")) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return proxy[x] > proxy[y]; });
  }
  BUILD_TRY(hipMalloc(&d_order, (size_t)n_states * sizeof(int32_t)));
  BUILD_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)n_states * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  BuildArgs a;
  a.order = d_order;
  a.n_states = n_states, a.n_qubits = n_qubits, a.n_ops = n_ops, a.cap = cap;
  a.op = d_op, a.q0 = d_q0, a.alpha = d_alpha;
  a.budget = trunc_budget, a.zero = value_of_zero;
  a.arena = arena, a.work = work, a.heap = heap, a.heap_cap = heap_cap, a.heap_top = d_ctr + 1;
  a.dims_out = d_dims, a.fid_out = d_fid, a.secs_out = d_secs, a.offs_out = d_offs, a.counter = d_ctr, a.error = d_err;
  a.jl_offset = (int)(lds_meta / sizeof(double)), a.jl_elems = jl_elems;
  a.partial = (flags & QK_BUILD_PARTIAL) ? 1 : 0;
  a.truncate = (flags & QK_BUILD_TRUNCATE) ? 1 : 0;
  a.block = 1;
  if (const char* v = std::getenv("QK_BUILD_BLOCK")) a.block = std::atoi(v) != 0;
  BUILD_TRY(hipEventRecord(c->ev0, c->stream));
  if (wgs_variant == 4) qkb256::qk_build_kernel<4><<<dim3((unsigned)grid), dim3(bt), lds, c->stream>>>(a);
  else if (wgs_variant == 2) qkb256::qk_build_kernel<2><<<dim3((unsigned)grid), dim3(bt), lds, c->stream>>>(a);
  else qkb512::qk_build_kernel<1><<<dim3((unsigned)grid), dim3(bt), lds, c->stream>>>(a);
  BUILD_TRY(hipGetLastError());
  BUILD_TRY(hipEventRecord(c->ev1, c->stream));
  qk_built* b = new qk_built;
  b->ctx = c, b->n_states = n_states, b->n_qubits = n_qubits;
  b->dims.resize((size_t)n_states * (n_qubits + 1));
  b->fidelity.resize(n_states);
  b->offsets.resize(n_states);
  std::vector<long long> offs(n_states);
  int errv[32] = {0};
  unsigned long long ctr[2] = {0, 0};
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(b->dims.data(), d_dims, b->dims.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(b->fidelity.data(), d_fid, (size_t)n_states * sizeof(double), hipMemcpyDeviceToHost);
  b->secs.resize(n_states);
  if (e == hipSuccess) e = hipMemcpy(b->secs.data(), d_secs, (size_t)n_states * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(offs.data(), d_offs, (size_t)n_states * sizeof(long long), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(errv, d_err, sizeof(errv), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(ctr, d_ctr, sizeof(ctr), hipMemcpyDeviceToHost);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
  release();
  if (std::getenv("QK_BUILD_DEBUG")) {
    const double t_host1 = (double)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-6;
    std::fprintf(stderr, "[qk_build_mps] host wall %.2f s for a %.2f s launch (arena %.1f GB, heap %.1f GB: allocation, upload, download, release)\n", t_host1 - t_host0, ms / 1e3,
                 (double)grid * (double)per_wg / 1e9, (double)heap_cap * sizeof(cd) / 1e9);
  }
  if (e != hipSuccess) {
    (void)hipFree(heap);
    delete b;
    return qk_fail(QK_EDEVICE, "qk_build_mps: %s", hipGetErrorString(e));
  }
  const int err = errv[0];
  if (std::getenv("QK_BUILD_DEBUG"))
    std::fprintf(stderr, "[qk_build_mps] %d states, grid %lld x %d threads, %.1f ms; Jacobi: %d factorisations, %.2f sweeps on average, %d at most, %d unconverged, %d in LDS / %d from L2; error bits %d\n",
                 n_states, grid, bt, ms, errv[1], errv[1] ? (double)errv[2] / errv[1] : 0.0, errv[3], errv[4], errv[5], errv[6], err);
  if (std::getenv("QK_BUILD_DEBUG") && errv[7]) {
    unsigned long long tk[5];
    for (int i = 0; i < 5; ++i) std::memcpy(&tk[i], errv + 14 + 2 * i, 8);
    std::fprintf(stderr, "[qk_build_mps] %d centre moves of large sites by Gram-Schmidt twice (no sweeps)\n", errv[24]);
    std::fprintf(stderr, "[qk_build_mps] %d preconditioned block factorisations: %.3f s of workgroup time (sort+copy %.1f %%, Gram-Schmidt %.1f %%, sweeps %.1f %%, V and W = A V %.1f %%)\n", errv[7],
                 (double)tk[0] / 1e8, 100.0 * tk[1] / std::max(1ull, tk[0]), 100.0 * tk[2] / std::max(1ull, tk[0]), 100.0 * tk[3] / std::max(1ull, tk[0]), 100.0 * tk[4] / std::max(1ull, tk[0]));
  }
  if (std::getenv("QK_BUILD_DEBUG")) {
    std::vector<double> t(b->secs);
    std::sort(t.begin(), t.end());
    double sum = 0;
    for (double x : t) sum += x;
    std::fprintf(stderr, "[qk_build_mps] workgroup time per state: median %.3f s, 90 %% %.3f s, the three longest %.3f %.3f %.3f s; sum %.1f s = %.2f s per workgroup slot\n", t[t.size() / 2], t[(size_t)(0.9 * (t.size() - 1))],
                 t[t.size() >= 3 ? t.size() - 3 : 0], t[t.size() >= 2 ? t.size() - 2 : 0], t.back(), sum, sum / (double)grid);
  }
  if (std::getenv("QK_BUILD_DEBUG")) {
    unsigned long long ticks = 0, steps = 0;
    std::memcpy(&ticks, errv + 8, 8), std::memcpy(&steps, errv + 10, 8);
    unsigned long long busy = 0;
    std::memcpy(&busy, errv + 12, 8);
    std::fprintf(stderr, "[qk_build_mps] workgroups busy %.1f %% of the launch (%.3f s of workgroup time per state); sweeps are %.1f %% of the busy time (%.2f us per step, %llu steps)\n",
                 100.0 * (double)busy / 1e8 / ((double)grid * ms / 1e3), (double)busy / 1e8 / n_states, 100.0 * (double)ticks / (double)std::max(1ull, busy),
                 steps ? (double)ticks / 100.0 / (double)steps : 0.0, steps);
  }
  if (err) {
    (void)hipFree(heap);
    delete b;
    if (err & ERR_GATE) return qk_fail(QK_EINVAL, "qk_build_mps: gate on a qubit outside the register");
    if (err & ERR_BOND) return qk_fail(QK_EINVAL, "qk_build_mps: a bond grew beyond max_bond = %d", cap);
    if (err & ERR_HEAP) return qk_fail(QK_EDEVICE, "qk_build_mps: the packed states need %llu complex numbers, the heap holds %zu", ctr[1], heap_cap);
    return qk_fail(QK_EDEVICE, "qk_build_mps: a Jacobi factorisation did not converge in %d sweeps", MAX_SWEEPS);
  }
  for (int s = 0; s < n_states; ++s) b->offsets[s] = offs[s];
  b->heap = heap;
  b->total = (int64_t)ctr[1];
  b->kernel_ms = ms;
  *out = b;
  return QK_OK;
}

extern "C" int qk_built_info(const qk_built* b, int32_t* dims, double* fidelity, int64_t* offsets, int64_t* total_complex, double* kernel_ms) {
  if (!b) return qk_fail(QK_EINVAL, "qk_built_info: null handle");
  if (dims) std::copy(b->dims.begin(), b->dims.end(), dims);
  if (fidelity) std::copy(b->fidelity.begin(), b->fidelity.end(), fidelity);
  if (offsets) std::copy(b->offsets.begin(), b->offsets.end(), offsets);
  if (total_complex) *total_complex = b->total;
  if (kernel_ms) *kernel_ms = b->kernel_ms;
  return QK_OK;
}

extern "C" int qk_built_download(const qk_built* b, double* host) {
  if (!b || !host) return qk_fail(QK_EINVAL, "qk_built_download: null argument");
  HIP_TRY(hipSetDevice(b->ctx->device));
  HIP_TRY(hipMemcpy(host, b->heap, (size_t)b->total * sizeof(cd), hipMemcpyDeviceToHost));
  return QK_OK;
}

extern "C" int qk_mps_set_from_built(qk_ctx* c, const qk_built* b, qk_mps_set** out) {
  if (!c || !b || !out) return qk_fail(QK_EINVAL, "qk_mps_set_from_built: null argument");
  if (b->ctx != c) return qk_fail(QK_EINVAL, "qk_mps_set_from_built: the states were built in another context");
  for (int s = 0; s < b->n_states; ++s)
    if (b->fidelity[s] < 0) return qk_fail(QK_EINVAL, "qk_mps_set_from_built: state %d outgrew max_bond and was dropped (QK_BUILD_PARTIAL)", s);
  HIP_TRY(hipSetDevice(c->device));
  const int ns = b->n_states, n = b->n_qubits, stride = n + 1;
  auto pad16 = [](int x) { return (x + 15) / 16 * 16; };
  std::vector<int32_t> pad((size_t)ns * stride);
  std::vector<long long> src((size_t)ns * n), dst((size_t)ns * n);
  long long total = 0;
  int max_pad = 0;
  for (int s = 0; s < ns; ++s) {
    long long pos = b->offsets[s];
    for (int k = 0; k <= n; ++k) {
      pad[(size_t)s * stride + k] = pad16(b->dims[(size_t)s * stride + k]);
      max_pad = std::max(max_pad, pad[(size_t)s * stride + k]);
    }
    for (int k = 0; k < n; ++k) {
      src[(size_t)s * n + k] = pos;
      dst[(size_t)s * n + k] = total;
      pos += 2ll * b->dims[(size_t)s * stride + k] * b->dims[(size_t)s * stride + k + 1];
      total += 2ll * pad[(size_t)s * stride + k] * 2 * pad[(size_t)s * stride + k + 1];
    }
  }
  qk_mps_set* m = new qk_mps_set;
  m->ctx = c, m->n_states = ns, m->n_sites = n, m->max_pad = max_pad;
  m->dims_true = b->dims;
  m->bytes = total * (long long)sizeof(double);
  long long* d_src = nullptr;
  hipError_t e = hipMalloc(&m->d_data, (size_t)m->bytes);
  if (e == hipSuccess) e = hipMemsetAsync(m->d_data, 0, (size_t)m->bytes, c->stream);
  if (e == hipSuccess) e = hipMalloc(&m->d_dims, pad.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_true, pad.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_offs, dst.size() * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&d_src, src.size() * sizeof(long long));
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_dims, pad.data(), pad.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_true, b->dims.data(), pad.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_offs, dst.data(), dst.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_src, src.data(), src.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    qk_pack_built_kernel<<<dim3((unsigned)(ns * n)), dim3(256), 0, c->stream>>>(b->heap, d_src, reinterpret_cast<const long long*>(m->d_offs), m->d_true,
                                                                              m->d_dims, n, m->d_data);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_src);
  if (e != hipSuccess) {
    (void)hipFree(m->d_data), (void)hipFree(m->d_dims), (void)hipFree(m->d_true), (void)hipFree(m->d_offs);
    delete m;
    return qk_fail(QK_EDEVICE, "qk_mps_set_from_built: %s", hipGetErrorString(e));
  }
  *out = m;
  return QK_OK;
}

extern "C" int qk_built_destroy(qk_built* b) {
  if (!b) return QK_OK;
  if (b->heap) (void)hipFree(b->heap);
  delete b;
  return QK_OK;
}

extern "C" int qk_debug_jacobi_precond(qk_ctx* c, int32_t p, int32_t q, double* a_inout, double* v_out, double* sig_out, int32_t* ord_out, int32_t* stats_out) {
  if (!c || !a_inout || !v_out || !sig_out || !ord_out) return qk_fail(QK_EINVAL, "qk_debug_jacobi_precond: null argument");
  if (p < 1 || q < 16 || q > 1024 || p > 64 * qkb256::MGS_R) return qk_fail(QK_EINVAL, "qk_debug_jacobi_precond: bad shape %d x %d (16 <= q <= 1024, p <= %d)", p, q, 64 * qkb256::MGS_R);
  HIP_TRY(hipSetDevice(c->device));
  struct Bufs {
    cd *dA = nullptr, *dV = nullptr, *dS = nullptr, *dL = nullptr;
    double* dSig = nullptr;
    int *dO = nullptr, *dE = nullptr, *dC = nullptr;
    ~Bufs() { (void)hipFree(dA), (void)hipFree(dV), (void)hipFree(dS), (void)hipFree(dL), (void)hipFree(dSig), (void)hipFree(dO), (void)hipFree(dE), (void)hipFree(dC); }
  } b;
  const size_t lrows = (size_t)(q + 31) / 32 * 32, qpad = (size_t)(q + 15) / 16 * 16;
  HIP_TRY(hipMalloc(&b.dA, (size_t)p * q * sizeof(cd)));
  HIP_TRY(hipMalloc(&b.dV, (size_t)q * q * sizeof(cd)));
  HIP_TRY(hipMalloc(&b.dS, (size_t)p * q * sizeof(cd)));
  HIP_TRY(hipMalloc(&b.dL, lrows * qpad * sizeof(cd)));
  HIP_TRY(hipMalloc(&b.dSig, (size_t)q * sizeof(double)));
  HIP_TRY(hipMalloc(&b.dO, (size_t)q * sizeof(int)));
  HIP_TRY(hipMalloc(&b.dE, 32 * sizeof(int)));
  HIP_TRY(hipMalloc(&b.dC, (qpad / 8) * (qpad / 8) * sizeof(int)));
  HIP_TRY(hipMemset(b.dE, 0, 32 * sizeof(int)));
  HIP_TRY(hipMemcpy(b.dA, a_inout, (size_t)p * q * sizeof(cd), hipMemcpyHostToDevice));
  const size_t lds_head = (size_t)(((q * 12 + 15) / 16) * 2 + 2) * sizeof(double);
  const bool wide = std::getenv("QK_BUILD_WGS") && std::atoi(std::getenv("QK_BUILD_WGS")) <= 1;  // the 512-thread variant of the builder
  const size_t lds = (wide ? 152 : 72) * 1024;
  const int lds_elems = (int)((lds - lds_head) / sizeof(cd));
  const size_t need = lds_head + (size_t)(wide ? qkb512::NWV * qkb512::BLK_LDS : qkb256::NWV * qkb256::BLK_LDS) * sizeof(cd);
  if (need > lds) return qk_fail(QK_EINVAL, "qk_debug_jacobi_precond: q = %d needs %zu bytes of LDS", q, need);
  int* const chk = std::getenv("QK_BUILD_NO_SKIP") ? nullptr : b.dC;
  if (wide) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qkb512::qk_jacobi_precond_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    qkb512::qk_jacobi_precond_kernel<<<dim3(1), dim3(512), lds, c->stream>>>(b.dA, p, q, b.dV, b.dSig, b.dO, b.dE, b.dS, b.dL, lds_elems, chk);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qkb256::qk_jacobi_precond_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024));
    qkb256::qk_jacobi_precond_kernel<<<dim3(1), dim3(256), lds, c->stream>>>(b.dA, p, q, b.dV, b.dSig, b.dO, b.dE, b.dS, b.dL, lds_elems, chk);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(a_inout, b.dA, (size_t)p * q * sizeof(cd), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(v_out, b.dV, (size_t)q * q * sizeof(cd), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sig_out, b.dSig, (size_t)q * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ord_out, b.dO, (size_t)q * sizeof(int), hipMemcpyDeviceToHost));
  int errv[32] = {0};
  HIP_TRY(hipMemcpy(errv, b.dE, sizeof errv, hipMemcpyDeviceToHost));
  if (stats_out) {  // [0] sweeps, then 100 MHz ticks (low words): [1] all, [2] sort + copy, [3] Gram-Schmidt, [4] sweeps, [5] V and W = A V
    stats_out[0] = errv[2];
    stats_out[1] = errv[14], stats_out[2] = errv[16], stats_out[3] = errv[18], stats_out[4] = errv[20], stats_out[5] = errv[22];
  }
  if (errv[0]) return qk_fail(QK_EDEVICE, "qk_debug_jacobi_precond: no convergence in %d sweeps", MAX_SWEEPS);
  return QK_OK;
}

extern "C" int qk_debug_jacobi(qk_ctx* c, int32_t p, int32_t q, double* a_inout, double* v_out, double* sig_out, int32_t* ord_out) {
  if (!c || !a_inout || !v_out || !sig_out || !ord_out) return qk_fail(QK_EINVAL, "qk_debug_jacobi: null argument");
  if (p < 1 || q < 1 || q > 2048) return qk_fail(QK_EINVAL, "qk_debug_jacobi: bad shape %d x %d", p, q);
  HIP_TRY(hipSetDevice(c->device));
  cd *dA = nullptr, *dV = nullptr;
  double* dS = nullptr;
  int *dO = nullptr, *dE = nullptr;
  HIP_TRY(hipMalloc(&dA, (size_t)p * q * sizeof(cd)));
  HIP_TRY(hipMalloc(&dV, (size_t)q * q * sizeof(cd)));
  HIP_TRY(hipMalloc(&dS, (size_t)q * sizeof(double)));
  HIP_TRY(hipMalloc(&dO, (size_t)q * sizeof(int)));
  HIP_TRY(hipMalloc(&dE, 32 * sizeof(int)));
  HIP_TRY(hipMemset(dE, 0, 32 * sizeof(int)));
  HIP_TRY(hipMemcpy(dA, a_inout, (size_t)p * q * sizeof(cd), hipMemcpyHostToDevice));
  qkb256::qk_jacobi_kernel<<<dim3(1), dim3(256), (size_t)q * (sizeof(double) + sizeof(int)) + 16, c->stream>>>(dA, p, q, dV, dS, dO, dE);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(a_inout, dA, (size_t)p * q * sizeof(cd), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(v_out, dV, (size_t)q * q * sizeof(cd), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sig_out, dS, (size_t)q * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ord_out, dO, (size_t)q * sizeof(int), hipMemcpyDeviceToHost));
  int err = 0;
  HIP_TRY(hipMemcpy(&err, dE, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dA), (void)hipFree(dV), (void)hipFree(dS), (void)hipFree(dO), (void)hipFree(dE);
  if (err) return qk_fail(QK_EDEVICE, "qk_debug_jacobi: no convergence in %d sweeps", MAX_SWEEPS);
  return QK_OK;
}
