// qk_device.h -- device-side definitions shared by every sweep kernel (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

static constexpr int TILE = 16;  // M/N granule of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct SweepArgs {
  const double* xdata;
  const int32_t* xdims;   // padded bonds
  const int32_t* xtrue;   // true bonds
  const int64_t* xoffs;
  const double* ydata;
  const int32_t* ydims;
  const int32_t* ytrue;
  const int64_t* yoffs;
  int n_sites;
  const int32_t* pairs;
  long long npairs;
  const int32_t* groups;  // (first pair, count) per group
  long long ngroups;
  double* values;
  double* z;
  double* scratch;
  long long x_plane;  // doubles per X plane
  long long t_plane;  // doubles per T plane
  unsigned long long* counter;  // work-queue heads: head of queue s at counter[QK_QSTRIDE * s] (kernels without XCD queues use counter[0] only)
  int nq;                       // queues of this launch: 8 (one per XCD) or 16 (two classes of pairs x 8); <= 1: one list [0, npairs)
  long long qstart[17];         // queue s holds the pairs [qstart[s], qstart[s + 1]) of this launch's list
  // EDGE BLOCKS (site-fused sweep; edge_k = 0: none): the first and the last edge_k sites of every state contracted into one matrix
  // each -- left block L[s][a] (s = the 2^edge_k configurations of the first edge_k physical legs, a = bond edge_k), right block
  // R[s][a] (the last edge_k legs, bond n - edge_k) --, interleaved complex, row-major with the padded bond as leading dimension;
  // a pair's environment then STARTS as X = Ly^T conj(Lx) (one product with K = 2^edge_k instead of edge_k sites of the chain)
  // and the overlap ENDS as sum X . (Ry^T conj(Rx)).  xedge_offs / yedge_offs: [n_states][2] element offsets of (L, R).
  const double* xedge;
  const long long* xedge_offs;
  const double* yedge;
  const long long* yedge_offs;
  int edge_k;
  // MERGED STEPS (site-fused sweep; merge_steps = 0: none): xmg / ymg hold, per state, the chain's sites [edge_k, n - edge_k) contracted
  // in twos -- tensors [l][4][r], physical index 2 p1 + p2, interleaved complex, offsets (doubles) in x/ymg_offs[state][merge_steps],
  // merge_steps = (n - 2 edge_k) / 2.  A workgroup decides per pair and per step whether it walks the merged tensor (half the barriers,
  // set-ups and load latencies for the same matrix work when the bonds are level) or the two plain sites (qk_fused.h: qkf_step_table).
  const double* xmg;
  const double* ymg;
  const int64_t* xmg_offs;
  const int64_t* ymg_offs;
  int merge_steps;
  int turn_ints;  // the DET forms of the site-fused kernels: turn counters per set = blocks of b' x blocks of a' of the largest site (qk_fused.h: qkf_turn_add)
  unsigned long long* tail;     // per-launch device clocks (s_memrealtime, 100 MHz): [0] first workgroup start, [1] first workgroup exit, [4] last workgroup exit
  unsigned long long* err;      // one word per sweep (both launches): set by a DET kernel whose ordered accumulation ran out of patience (qk_get_stats fails the call)
  unsigned long long* prof;  // diagnostic build only: cycle sums per section (see QK_VARIANT=9)
  // GANG START (QK_GANG=1; gang_n = workgroups of this launch per XCD, 0 = off): the workgroups of an XCD begin their next pairs together -- they pull
  // neighbouring pairs of one plan tile, which share their x and y states, and only workgroups that walk those states at the same pace find each
  // other's fragments in the XCD's L2.  gang[QK_QSTRIDE * xcc]: [0] arrivals, [1] 'stop waiting' (a workgroup ran out of pairs, or a wait ran out).
  unsigned long long* gang;
  int gang_n;
  int debug_flags;           // timing experiments only (QK_DEBUG_FLAGS): bit 0 = skip epilogue stores, bit 1 = skip steady-state fetch/stash, bit 2 = skip MFMAs, bit 3 = skip steady-state barriers (all give WRONG results)
  int prio_mode;             // 0: none; 1: second half of the grid at s_setprio 1; 2: odd blocks at s_setprio 1
};

static constexpr int QK_QSTRIDE = 16;  // queue heads 128 bytes apart
static constexpr int QK_NQ_MAX = 16;

// The XCD this wave runs on (0..7): blocks are dealt round-robin over the 8 XCDs, each with its own 4 MiB L2.
__device__ __forceinline__ int qk_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }  // hwreg(HW_REG_XCC_ID, 0, 4)

// Next pair of this launch for a workgroup on XCD `xcc` (called by ONE lane): the plan lists the pairs as 8 queues per class
// (qk_plan_create: tiles of pairs that share their x and y states, dealt to the queues), a workgroup drains the queue of its
// own XCD first -- so that the workgroups that share an L2 stream the same few states -- and then steals from the others.
// Returns -1 when every queue is empty.
__device__ __forceinline__ long long qk_pull(const SweepArgs& g, const int xcc) {
  if (g.nq <= 1) {
    const long long p = (long long)atomicAdd(g.counter, 1ull);
    return p < g.npairs ? p : -1;
  }
  for (int s = 0; s < g.nq; ++s) {
    const int qd = ((xcc + s) & 7) + (s & 8);
    const long long n = g.qstart[qd + 1] - g.qstart[qd];
    if (n <= 0) continue;
    const long long t = (long long)atomicAdd(g.counter + QK_QSTRIDE * qd, 1ull);
    if (t < n) return g.qstart[qd] + t;
  }
  return -1;
}

// Gang start (SweepArgs.gang): called by ONE lane of a workgroup behind every qk_pull.  `round` counts this workgroup's pairs.  Never a hang: a wait is
// bounded (about a millisecond), and the first wait that runs out -- or the first workgroup without a pair -- switches the waiting off for the whole gang.
__device__ __forceinline__ void qk_gang_sync(const SweepArgs& g, const int xcc, const bool have_pair, int& round) {
  if (g.gang_n <= 1) return;
  unsigned long long* const a = g.gang + QK_QSTRIDE * xcc;
  if (!have_pair) {
    __hip_atomic_store(a + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const unsigned long long target = (unsigned long long)g.gang_n * (unsigned long long)(++round);
  __hip_atomic_fetch_add(a, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int spins = 0;; ++spins) {
    if (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
    if (__hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    if (spins > 4000) {
      __hip_atomic_store(a + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

// Device clocks of a launch for the tail accounting (one lane per workgroup): when the first workgroup started, when the first
// one ran out of work and when the last one did -- (last - first exit) / (last exit - first start) is the share of the launch
// during which the chip was draining.
__device__ __forceinline__ void qk_tail_start(const SweepArgs& g) {
  if (g.tail) atomicMin(g.tail, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void qk_tail_exit(const SweepArgs& g) {
  if (g.tail) {
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    atomicMin(g.tail + 1, t);
    atomicMax(g.tail + 4, t);
  }
}

// Workgroup barrier that publishes LDS writes only: it does NOT drain outstanding global loads or LDS-DMAs
// (a __syncthreads() would wait vmcnt(0) and cancel the prefetch that is meant to stay in flight).
__device__ __forceinline__ void qk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
