// qkgram.hip -- MI355X (gfx950 / CDNA4) engine for the quantum-kernel Gram hot path.
//
// What it replaces (reference = mmetcalf14/qml-cutensornet, G = gpu_backend/kernel_state_ansatz.py):
//   G:372-400  the Python double loop calling  x_mps.vdot(y_mps)  once per Gram entry
//   G:380      MPS.vdot -> one cuTensorNet contraction + a device->host sync per entry
// by ONE persistent kernel launch per Gram share: every workgroup pulls (x_i, y_j) pairs
// from a device-side queue and carries the whole transfer-matrix sweep
//     X_0 = 1,   X_{k+1}[r,R] = sum_{L,l,p} X_k[l,L] * conj(A_k[L,p,R]) * B_k[l,p,r]
// on chip/L2 as a chain of complex GEMMs on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64), and finally writes |<x|y>|^2.
//
//
// Files: this one = C ABI (include/qkgram.h), packing, planner, small kernels, launches;  qk_fused.h = the site-fused
// sweep (the fp64 hot path) and the 2 x 2-tile one-wave sweep (fp64, bonds <= 32);  qk_ring.h = the ring sweep (complex64,
// very large bonds), the LDS-resident small-bond sweep and the one-tile one-wave sweep;  qk_build.hip = the device MPS builder.  lab/qk_lab.hip (experimental / diagnostic kernels) is NOT part of
// libqkgram.so: it is linked only into libqklab.so (-DQK_LAB), which tools/ load for A/B measurements.
// Written for gfx950 only: 64-lane wavefronts, 160 KiB LDS per CU, no portability layer.
#include "qk_host.h"
#include "qk_ring.h"
#include "qk_fused.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

// ----------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------
static thread_local std::string g_err;

int qk_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define fail qk_fail

extern "C" const char* qk_last_error(void) { return g_err.c_str(); }

// ----------------------------------------------------------------------------------------
// roctx ranges around the phases of the path (build / upload / sweep / all-gather / scatter: the reference's timing sites
// G:379-381 and its profiling keys), so that ONE `rocprofv3 --marker-trace --kernel-trace` run yields the phase table.
// The marker library is resolved at first use -- only when a profiler is attached (rocprofv3 preloads its tool library)
// or QK_ROCTX=1 -- and the calls are no-ops otherwise.
// ----------------------------------------------------------------------------------------
#include <dlfcn.h>
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
Roctx& roctx() {
  static Roctx r = [] {
    Roctx x;
    const char* e = std::getenv("QK_ROCTX");
    const char* pre = std::getenv("LD_PRELOAD");
    const bool attached = std::getenv("ROCP_TOOL_LIBRARIES") || (pre && std::strstr(pre, "rocprofiler"));
    if (e ? std::atoi(e) == 0 : !attached) return x;
    void* h = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return x;
    x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!x.push || !x.pop) x.push = nullptr, x.pop = nullptr;
    return x;
  }();
  return r;
}
}  // namespace
extern "C" int qk_range_push(const char* name) {
  Roctx& r = roctx();
  return (r.push && name) ? r.push(name) : 0;
}
extern "C" int qk_range_pop(void) {
  Roctx& r = roctx();
  return r.pop ? r.pop() : 0;
}

static inline int pad16(int x) { return (x + TILE - 1) / TILE * TILE; }

// a device allocation that is released on every exit path (HIP_TRY returns early)
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

// ----------------------------------------------------------------------------------------
// host: packing one MPS into the padded planar device image
// ----------------------------------------------------------------------------------------
extern "C" int64_t qk_pack_state_size(int32_t n_sites, const int32_t* bond_dims) {
  int64_t tot = 0;
  for (int k = 0; k < n_sites; ++k) tot += 2ll * pad16(bond_dims[k]) * 2 * pad16(bond_dims[k + 1]);
  return tot;
}

extern "C" int qk_pack_state(int32_t n_sites, const int32_t* bond_dims, const double* const* site_tensors,
                             int32_t layout, double* out, int64_t* site_offsets) {
  if (n_sites <= 0 || !bond_dims || !site_tensors || !out) return fail(QK_EINVAL, "qk_pack_state: null argument");
  if (layout != QK_LAYOUT_LPR && layout != QK_LAYOUT_LRP) return fail(QK_EINVAL, "qk_pack_state: unknown layout %d", layout);
  if (bond_dims[0] != 1 || bond_dims[n_sites] != 1) return fail(QK_EINVAL, "qk_pack_state: boundary bonds must be 1");
  int64_t off = 0;
  for (int k = 0; k < n_sites; ++k) {
    const int cl = bond_dims[k], cr = bond_dims[k + 1];
    if (cl <= 0 || cr <= 0) return fail(QK_EINVAL, "qk_pack_state: non-positive bond at site %d", k);
    const int pl = pad16(cl), pr = pad16(cr);
    const int64_t plane = (int64_t)pl * 2 * pr;
    double* re = out + off;
    double* im = re + plane;
    std::memset(re, 0, sizeof(double) * 2 * plane);
    const double* src = site_tensors[k];
    if (!src) return fail(QK_EINVAL, "qk_pack_state: null tensor at site %d", k);
    for (int l = 0; l < cl; ++l)
      for (int p = 0; p < 2; ++p)
        for (int r = 0; r < cr; ++r) {
          const int64_t s = (layout == QK_LAYOUT_LPR) ? (((int64_t)l * 2 + p) * cr + r) : (((int64_t)l * cr + r) * 2 + p);
          const int64_t d = ((int64_t)l * 2 + p) * pr + r;
          re[d] = src[2 * s];
          im[d] = src[2 * s + 1];
        }
    if (site_offsets) site_offsets[k] = off;
    off += 2 * plane;
  }
  return QK_OK;
}

// ----------------------------------------------------------------------------------------
// host: work model and planner
// ----------------------------------------------------------------------------------------
// Algorithmic flops of one overlap: 8 real flops per complex
// multiply-add, cheaper association per site.  Padded: what this engine executes.
#ifndef QKF_XCAP_ONE_V
#define QKF_XCAP_ONE_V 8192
#endif
static constexpr int QKF_XCAP_ONE = QKF_XCAP_ONE_V, QKF_XCAP_TWO = 4608;  // elements of the fused sweep's LDS X buffer with one / two workgroups per CU
// shapes of the two instantiations: waves per workgroup, T slots per wave, waves per SIMD (experiment builds override them)
#ifndef QKF_ONE_NW
#define QKF_ONE_NW 12  // three waves per SIMD at 168 VGPRs: 452 against 482 ms for 8 waves x 4 slots on the headline set (16 x 1: 455)
#define QKF_ONE_S 2
#define QKF_ONE_WPS 3
#endif
#ifndef QKF_TWO_NW
#define QKF_TWO_NW 8  // four waves per SIMD at 128 VGPRs, one slot: 16.4 against 18.2 ms for 4 waves x 4 slots on the 40-qubit x 4-layer set
#define QKF_TWO_S 1
#define QKF_TWO_WPS 4
#endif
#define QKF_KERNEL_ONE qk_sweep_fused_kernel<QKF_ONE_NW, QKF_ONE_S, QKF_XCAP_ONE, QKF_ONE_WPS>
#define QKF_KERNEL_TWO qk_sweep_fused_kernel<QKF_TWO_NW, QKF_TWO_S, QKF_XCAP_TWO, QKF_TWO_WPS>
#ifndef QKF_DUAL_NW
#define QKF_DUAL_NW 12
#define QKF_DUAL_WPS 3
#endif
#define QKF_KERNEL_DUAL qk_sweep_fused_dual_kernel<QKF_DUAL_NW, QKF_XCAP_ONE, QKF_DUAL_WPS>
// The narrow site size of the pair classes (QK_PLAN_FIT, elements of X): when a set holds a substantial share of LARGE pairs, only
// pairs whose work sits in sites of at most this many elements go to the two-workgroup shape -- the 12-wave dual shape is the
// better one from about 4 x 4 tiles per site on (uniform chains: bond 48 39.8 against 42.4 ms for the two-workgroup shape, bond 64
// 99.8 against 86.6 ms).  60 qubits x 6 layers, whole sweep: 4608 (every site that fits the smaller buffer) 377.0 ms, 3584 365.9,
// 3072 365.6, 2560 and below (one launch of the dual shape) 368.1.  A set without large pairs (40 qubits x 4 layers) stays on the
// two-workgroup shape as a whole: 12.65 ms against 13.1-13.2 when split at the narrow size.
static double g_plan_fit = 3072;  // (set by qk_plan_create)
static void pair_work(int n, const int32_t* a, const int32_t* b, double* flops, double* padded, double* bytes, double* fit_two = nullptr, double* fit_narrow = nullptr) {
  double f = 0, fp = 0, by = 0, ft = 0, fn = 0;
  for (int k = 0; k < n; ++k) {
    const double a0 = a[k], a1 = a[k + 1], b0 = b[k], b1 = b[k + 1];
    const double f1 = a0 * b0 * 2 * b1 + 2 * a0 * a1 * b1;
    const double f2 = a0 * b0 * 2 * a1 + 2 * b0 * a1 * b1;
    f += 8 * std::min(f1, f2);
    const double A0 = pad16(a[k]), A1 = pad16(a[k + 1]), B0 = pad16(b[k]), B1 = pad16(b[k + 1]);
    fp += 8 * (A0 * B0 * 2 * B1 + 2 * A0 * A1 * B1);
    if (A0 * B0 <= QKF_XCAP_TWO && A1 * B1 <= QKF_XCAP_TWO) ft += 8 * (A0 * B0 * 2 * B1 + 2 * A0 * A1 * B1);  // X and X' of this site fit the smaller buffer
    if (A0 * B0 <= g_plan_fit && A1 * B1 <= g_plan_fit) fn += 8 * (A0 * B0 * 2 * B1 + 2 * A0 * A1 * B1);      // ... with room to spare (see g_plan_fit)
    by += 16.0 * 2 * (a0 * a1 + b0 * b1);
  }
  if (fit_two) *fit_two = ft;
  if (fit_narrow) *fit_narrow = fn;
  *flops = f;
  *padded = fp;
  *bytes = by + 8;
}

// Matrix instructions the site-fused sweep issues for the pair (x = a, y = b): per site (a^/16)(b'^/16) tiles of T, each
// ceil(b/4) k-steps in phase 1 and (a'^/16) column blocks x ceil(a/4)-bounded k-steps in phase 2 (x 3 for the 3M product,
// x 2 for p).  The ring sweep's padded flops follow the same asymmetry, so one model serves both.
static double fused_cost(int n, const int32_t* a, const int32_t* b) {
  double c = 0;
  for (int k = 0; k < n; ++k) {
    const double A0 = pad16(a[k]) / 16, A1 = pad16(a[k + 1]) / 16, B1 = pad16(b[k + 1]) / 16;
    const double kb = (b[k] + 3) / 4, ka = (a[k] + 3) / 4;
    c += 6 * A0 * B1 * kb + 6 * B1 * A1 * ka;
  }
  return c;
}

// How many sites at either end of the chain the site-fused sweep should take from edge blocks (qk_fused.h: qkf_edge_prefix /
// qkf_edge_suffix) instead of walking them: the k that minimises, over a sample of this rank's pairs, the matrix instructions of
// the sweep plus a fixed cost per site walked (two barriers, set-up, load latencies: 375 instructions' worth = 2.5 us of a 12-wave
// workgroup, measured on uniform small chains, tools/site_overhead.py).  A block product costs tiles(a_k) tiles(b_k) 2^k / 4 x 3
// instructions x 2 (its tiles stream both blocks from L2).  Calibrated on the two headline sets with merged steps in the middle of
// the chain: 60 qubits x 6 layers, k = 6 / 7 / 8 / 9 measured 403.8 / 393.8 / 396.2 / 401.8 ms (model, relative to k = 8: 1.008 /
// 0.997 / 1 / 1.036); 40 qubits x 4 layers, k = 5 .. 9: 13.23 / 13.11 / 12.80 / 12.85 / 13.83 ms (model 1.054 / 1.021 / 0.995 / 1 /
// 1.087).  Before the merged steps (every site of the middle walked singly) k = 8 was the minimum, at factor 1.5; the chain's cost is fused_cost's.  QK_EDGE=0 disables, QK_EDGE=k forces.  This is a contraction order chosen on the
// host (north star; reference call site G:380): while the bonds still grow like 2^k, the ends of the two states are cheaper to
// contract across their physical legs than along the chain.
static int choose_edge_k(const int n_sites, const int32_t* x_dims, const int32_t* y_dims, const std::vector<int32_t>& pairs) {
  constexpr int KMIN = 4, KMAX = 9;  // K = 2^k >= 16 (four k-steps in flight); 2^9 rows per block at most
  if (const char* e = std::getenv("QK_EDGE")) {
    const int v = std::atoi(e);
    if (v <= 0) return 0;
    return (v >= KMIN && v <= KMAX && n_sites >= 2 * v + 2) ? v : 0;
  }
  const int64_t np = (int64_t)pairs.size() / 2;
  if (np == 0 || n_sites < 2 * KMIN + 2) return 0;
  const int stride = n_sites + 1;
  const int64_t step = std::max<int64_t>(1, np / 512);
  const double over = 375.0;
  auto t16 = [](const int v) { return (double)((v + 15) / 16); };
  std::vector<double> total((size_t)KMAX + 1, 0.0);
  for (int64_t t = 0; t < np; t += step) {
    const int32_t* a = x_dims + (int64_t)pairs[2 * t] * stride;
    const int32_t* b = y_dims + (int64_t)pairs[2 * t + 1] * stride;
    std::vector<double> site((size_t)n_sites);
    for (int k = 0; k < n_sites; ++k)
      site[(size_t)k] = 6 * t16(a[k]) * t16(b[k + 1]) * ((b[k] + 3) / 4) + 6 * t16(b[k + 1]) * t16(a[k + 1]) * ((a[k] + 3) / 4) + over;
    double chain = 0;
    for (double v : site) chain += v;
    total[0] += chain;
    double ends = 0;  // cost of the sites the edges replace
    for (int k = 1; k <= KMAX && n_sites >= 2 * k + 2; ++k) {
      ends += site[(size_t)k - 1] + site[(size_t)(n_sites - k)];
      if (k < KMIN) continue;
      const double blocks = 2.0 * 3.0 * ((1 << k) / 4) * (t16(a[k]) * t16(b[k]) + t16(a[n_sites - k]) * t16(b[n_sites - k])) + 2 * over;
      total[(size_t)k] += chain - ends + blocks;
    }
  }
  int best = 0;
  for (int k = KMIN; k <= KMAX && n_sites >= 2 * k + 2; ++k)
    if (total[(size_t)k] > 0 && total[(size_t)k] < total[(size_t)best] * 0.995) best = k;
  return best;
}

// ----------------------------------------------------------------------------------------
// The tiled plan: XCD-aware work queues (default; QK_PLAN_XCD=0 or an explicit locality `block` selects the flat list).
// An MI355X has 8 XCDs with a private 4 MiB L2 each, and blocks are dealt to them round-robin.  With one cost-ordered list
// the 32 (or 64) workgroups that share an L2 stream 64 unrelated states through it (measured hit rate 39 %, 2.5-3 x the
// algorithmic bytes at the L2 <-> fabric boundary).  Here the states are sorted by weight, the Gram is cut into tiles of T x T
// pairs in that order -- pairs of a tile share their T x states and T y states and cost about the same, so the workgroups
// sweeping a tile walk the chain at a similar pace --, the tiles are dealt (heaviest first, to the least loaded) to the
// ranks and, per class of pairs, to 8 queues; a workgroup drains the queue of its own XCD and then steals (qk_pull).
// ----------------------------------------------------------------------------------------
static void plan_tiled(qk_plan* p, const int n_sites, const int nx, const int32_t* x_dims, const int ny, const int32_t* y_dims, const bool sym, const bool orient, const int world,
                       const int rank, const int T) {
  const int stride = n_sites + 1;
  auto order_of = [&](const int n, const int32_t* dims) {
    std::vector<double> w((size_t)n);
    for (int s = 0; s < n; ++s) {
      const int32_t* d = dims + (int64_t)s * stride;
      double acc = 0;
      for (int k = 0; k < n_sites; ++k) {
        const double A0 = pad16(d[k]), A1 = pad16(d[k + 1]);
        acc += A0 * A1 * (A0 + A1);
      }
      w[(size_t)s] = acc;
    }
    std::vector<int> o((size_t)n);
    std::iota(o.begin(), o.end(), 0);
    std::stable_sort(o.begin(), o.end(), [&](const int u, const int v) { return w[(size_t)u] > w[(size_t)v]; });
    return o;
  };
  const std::vector<int> ox = order_of(nx, x_dims), oy = sym ? ox : order_of(ny, y_dims);
  struct Item {
    int32_t i, j;
    double f, fp, by, ft, fn;
  };
  struct Tile {
    int64_t start, count;
    double cost;
  };
  std::vector<Item> items;
  std::vector<Tile> tiles;
  const int nbx = (nx + T - 1) / T, nby = (ny + T - 1) / T;
  for (int bj = 0; bj < nby; ++bj)
    for (int bi = 0; bi < nbx; ++bi) {
      if (sym && bi > bj) continue;
      Tile t{(int64_t)items.size(), 0, 0.0};
      for (int v = bj * T; v < std::min(ny, (bj + 1) * T); ++v)
        for (int u = bi * T; u < std::min(nx, (bi + 1) * T); ++u) {
          if (sym && u > v) continue;  // positions in the weight order: every unordered pair once
          int xi = ox[(size_t)u], yj = oy[(size_t)v];
          if (sym && !orient && xi > yj) std::swap(xi, yj);  // the plain symmetric list names a pair as i <= j
          if (orient && xi != yj && fused_cost(n_sites, x_dims + (int64_t)yj * stride, y_dims + (int64_t)xi * stride) < fused_cost(n_sites, x_dims + (int64_t)xi * stride, y_dims + (int64_t)yj * stride))
            std::swap(xi, yj);
          Item it{xi, yj, 0, 0, 0, 0};
          pair_work(n_sites, x_dims + (int64_t)xi * stride, y_dims + (int64_t)yj * stride, &it.f, &it.fp, &it.by, &it.ft, &it.fn);
          items.push_back(it);
          t.cost += it.fp;
        }
      t.count = (int64_t)items.size() - t.start;
      if (t.count) tiles.push_back(t);
    }
  std::vector<int> by_cost(tiles.size());
  std::iota(by_cost.begin(), by_cost.end(), 0);
  std::stable_sort(by_cost.begin(), by_cost.end(), [&](const int u, const int v) { return tiles[(size_t)u].cost > tiles[(size_t)v].cost; });
  if (world > 1) {
    // several ranks: the lightest tiles (the last 3 % of the work) are dealt pair by pair, so that the shares end level to a
    // pair's cost instead of a tile's (cut into one-pair tiles here; they keep their place behind the whole tiles)
    double total = 0, acc = 0;
    for (const Tile& t : tiles) total += t.cost;
    std::vector<Tile> cut;
    std::vector<int> order;
    for (const int t : by_cost) {
      acc += tiles[(size_t)t].cost;
      if (acc <= 0.97 * total || tiles[(size_t)t].count == 1) {
        order.push_back((int)cut.size());
        cut.push_back(tiles[(size_t)t]);
      } else {
        std::vector<int64_t> q((size_t)tiles[(size_t)t].count);
        std::iota(q.begin(), q.end(), tiles[(size_t)t].start);
        std::stable_sort(q.begin(), q.end(), [&](const int64_t u, const int64_t v) { return items[(size_t)u].fp > items[(size_t)v].fp; });
        for (const int64_t e : q) {
          order.push_back((int)cut.size());
          cut.push_back(Tile{e, 1, items[(size_t)e].fp});
        }
      }
    }
    tiles.swap(cut), by_cost.swap(order);
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](const int u, const int v) { return tiles[(size_t)u].cost > tiles[(size_t)v].cost; });
  }
  // tiles to ranks: heaviest first, each to the least loaded rank
  std::vector<double> load((size_t)world, 0.0);
  std::vector<int64_t> per_rank((size_t)world, 0);
  std::vector<int> mine;
  for (const int t : by_cost) {
    const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    load[(size_t)r] += tiles[(size_t)t].cost, per_rank[(size_t)r] += tiles[(size_t)t].count;
    if (r == rank) mine.push_back(t);
  }
  // classes of this rank's pairs (see qk_plan_create): class 1 = nearly all of the work fits the fused sweep's smaller LDS buffer
  double split = 0.75;
  if (const char* e = std::getenv("QK_PLAN_SPLIT")) split = std::atof(e);
  double flops = 0, padded = 0, bytes = 0, fit_two = 0, fit_narrow = 0, small_work = 0;
  for (const int t : mine)
    for (int64_t q = tiles[(size_t)t].start; q < tiles[(size_t)t].start + tiles[(size_t)t].count; ++q) {
      const Item& it = items[(size_t)q];
      flops += it.f, padded += it.fp, bytes += it.by, fit_two += it.ft, fit_narrow += it.fn;
      if (it.fp > 0 && it.ft >= split * it.fp) small_work += it.fp;
    }
  // A MIXED set -- some states with every bond <= 32 next to larger ones -- keeps its small-small pairs on the one-wave sweep
  // (2 x 2 register tiles, 2-3 x faster per such pair than the multi-wave kernels): they form the second run instead, swept by
  // qk_sweep_wave2_kernel right behind the fused launch.  (A set whose bonds are all <= 32 runs that kernel anyway.)
  auto max_pad_of = [&](const int n, const int32_t* dims) {
    std::vector<int> mp((size_t)n, 0);
    for (int s_ = 0; s_ < n; ++s_)
      for (int k = 0; k <= n_sites; ++k) mp[(size_t)s_] = std::max(mp[(size_t)s_], pad16(dims[(int64_t)s_ * stride + k]));
    return mp;
  };
  const std::vector<int> mpx = max_pad_of(nx, x_dims), mpy = sym ? mpx : max_pad_of(ny, y_dims);
  const bool any_large = *std::max_element(mpx.begin(), mpx.end()) > 32 || *std::max_element(mpy.begin(), mpy.end()) > 32;
  auto small_pair = [&](const Item& it) { return mpx[(size_t)it.i] <= 32 && mpy[(size_t)it.j] <= 32; };
  int64_t n_small_pairs = 0, n_mine = 0;
  for (const int t : mine)
    for (int64_t q = tiles[(size_t)t].start; q < tiles[(size_t)t].start + tiles[(size_t)t].count; ++q) n_small_pairs += small_pair(items[(size_t)q]) ? 1 : 0, ++n_mine;
  const bool mixed = any_large && n_small_pairs >= std::max<int64_t>(64, n_mine / 50) && n_small_pairs < n_mine && !std::getenv("QK_PLAN_NO_MIXED");
  p->second_wave2 = mixed;
  const bool two_classes = mixed || !(small_work < 0.05 * padded || small_work > 0.95 * padded);
  // with a quarter or more of the work in large pairs the second class is cut at the narrow site size (g_plan_fit)
  const bool narrow = !mixed && two_classes && small_work < 0.75 * padded;
  auto cls_of = [&](const Item& it) {
    return mixed ? (small_pair(it) ? 1 : 0) : ((two_classes && it.fp > 0 && (narrow ? it.fn : it.ft) >= split * it.fp) ? 1 : 0);
  };
  p->pairs.clear(), p->groups.clear();
  p->second = qk_stats{};
  p->nq = QK_NQ_MAX;
  for (int c = 0; c < 2; ++c) {
    // this class's share of each tile, tiles to the 8 queues heaviest first / least loaded
    std::vector<std::pair<double, int>> part;  // (class-c cost of the tile, tile)
    for (const int t : mine) {
      double cc = 0;
      for (int64_t q = tiles[(size_t)t].start; q < tiles[(size_t)t].start + tiles[(size_t)t].count; ++q)
        if (cls_of(items[(size_t)q]) == c) cc += items[(size_t)q].fp;
      if (cc > 0) part.push_back({cc, t});
    }
    std::stable_sort(part.begin(), part.end(), [](const std::pair<double, int>& u, const std::pair<double, int>& v) { return u.first > v.first; });
    std::vector<double> ql(8, 0.0);
    std::vector<std::vector<int>> queue(8);
    for (const auto& pt : part) {
      const int qd = (int)(std::min_element(ql.begin(), ql.end()) - ql.begin());
      ql[(size_t)qd] += pt.first;
      queue[(size_t)qd].push_back(pt.second);
    }
    for (int qd = 0; qd < 8; ++qd) {
      p->qstart[8 * c + qd] = (int64_t)p->pairs.size() / 2;
      for (const int t : queue[(size_t)qd])
        for (int64_t q = tiles[(size_t)t].start; q < tiles[(size_t)t].start + tiles[(size_t)t].count; ++q) {
          const Item& it = items[(size_t)q];
          if (cls_of(it) != c) continue;
          p->groups.push_back((int32_t)(p->pairs.size() / 2));
          p->groups.push_back(1);
          p->pairs.push_back(it.i);
          p->pairs.push_back(it.j);
          if (c == 1) p->second.pairs += 1, p->second.flops += it.f, p->second.padded_flops += it.fp, p->second.bytes += it.by;
        }
    }
  }
  const int64_t np = (int64_t)p->pairs.size() / 2;
  p->qstart[16] = np;
  p->group = 1;
  p->n_first = p->qstart[8];  // == np when there is one class
  p->total_pairs = (int64_t)items.size();
  p->max_per_rank = *std::max_element(per_rank.begin(), per_rank.end());
  p->stats.pairs = np;
  p->stats.flops = flops, p->stats.padded_flops = padded, p->stats.bytes = bytes;
  p->fit_two = padded > 0 ? fit_two / padded : 1.0;
  p->fit_narrow = padded > 0 ? fit_narrow / padded : 1.0;
  p->edge_k = choose_edge_k(n_sites, x_dims, y_dims, p->pairs);
}

extern "C" int qk_plan_create(int32_t n_sites, int32_t nx, const int32_t* x_dims, int32_t ny, const int32_t* y_dims,
                              uint32_t flags, int32_t world_size, int32_t rank, int32_t block, qk_plan** out) {
  if (!out || !x_dims || n_sites <= 0 || nx <= 0) return fail(QK_EINVAL, "qk_plan_create: bad argument");
  const bool sym = (flags & QK_PLAN_SYMMETRIC) != 0;
  const bool orient = sym && (flags & QK_PLAN_ORIENT) != 0 && !(flags & QK_PLAN_QUADS);
  if (sym) {
    y_dims = x_dims;
    ny = nx;
  } else if (!y_dims || ny <= 0)
    return fail(QK_EINVAL, "qk_plan_create: y_dims required unless symmetric");
  if (world_size <= 0 || rank < 0 || rank >= world_size) return fail(QK_EINVAL, "qk_plan_create: bad rank %d/%d", rank, world_size);
  {
    const char* e = std::getenv("QK_PLAN_FIT");
    g_plan_fit = e ? std::atof(e) : 3072.0;
  }
  const int block_arg = block;
  if (block <= 0) block = std::max(nx, ny);  // flat list (QK_PLAN_XCD=0): the whole pair list in cost order
  qk_plan* p = new (std::nothrow) qk_plan;
  if (!p) return fail(QK_ENOMEM, "qk_plan_create: out of memory");
  p->n_sites = n_sites, p->nx = nx, p->ny = ny, p->symmetric = sym, p->world = world_size, p->rank = rank;

  struct Item {
    int32_t i, j;
    float cost;
    int32_t tile = 0;
    int32_t cls = 0;  // 1: nearly all of the pair's work sits in sites that fit the fused sweep's smaller LDS buffer
  };
  if (flags & QK_PLAN_QUADS) {
    // 2x2 blocks of pairs {i1, i2} x {j1, j2} (duos of consecutive states; the last duo of an odd set names its state
    // twice).  A symmetric Gram takes the duo pairs u <= v; its diagonal blocks then hold one mirrored pair (i > j) that
    // is computed redundantly.  Blocks are ordered by decreasing cost and dealt in serpentine order like pairs.
    p->quad = true;
    const int stride = n_sites + 1;
    const int nxd = (nx + 1) / 2, nyd = (ny + 1) / 2;
    struct Quad {
      int32_t i1, i2, j1, j2;
      double cost;
    };
    std::vector<Quad> quads;
    auto work = [&](int i, int j, double* f, double* fp, double* by) { pair_work(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride, f, fp, by); };
    for (int v = 0; v < nyd; ++v)
      for (int u = 0; u < nxd; ++u) {
        if (sym && u > v) continue;
        Quad q{2 * u, std::min(2 * u + 1, nx - 1), 2 * v, std::min(2 * v + 1, ny - 1), 0.0};
        const int32_t is[2] = {q.i1, q.i2}, js[2] = {q.j1, q.j2};
        for (int b = 0; b < 2; ++b)
          for (int a = 0; a < 2; ++a) {
            double f, fp, by;
            work(is[a], js[b], &f, &fp, &by);
            q.cost += fp;
          }
        quads.push_back(q);
      }
    std::stable_sort(quads.begin(), quads.end(), [](const Quad& a, const Quad& b) { return a.cost > b.cost; });
    std::vector<int64_t> per_rank(world_size, 0);
    double flops = 0, padded = 0, bytes = 0;
    int64_t t = 0;
    for (const Quad& q : quads) {
      const int64_t u = t % (2 * (int64_t)world_size);
      const int r = (int)(u < world_size ? u : 2 * (int64_t)world_size - 1 - u);
      per_rank[r] += 4;
      if (r == rank) {
        const int32_t is[2] = {q.i1, q.i2}, js[2] = {q.j1, q.j2};
        for (int b = 0; b < 2; ++b)
          for (int a = 0; a < 2; ++a) {
            p->pairs.push_back(is[a]);
            p->pairs.push_back(js[b]);
            const bool redundant = (a == 1 && q.i2 == q.i1) || (b == 1 && q.j2 == q.j1) || (sym && is[a] > js[b]);
            if (!redundant) {
              double f, fp, by;
              work(is[a], js[b], &f, &fp, &by);
              flops += f, padded += fp, bytes += by;
            }
          }
      }
      ++t;
    }
    p->groups = {0, 0};
    p->n_first = (int64_t)p->pairs.size() / 2;
    p->total_pairs = 4 * t;
    p->max_per_rank = *std::max_element(per_rank.begin(), per_rank.end());
    p->stats.pairs = (int64_t)p->pairs.size() / 2;
    p->stats.flops = flops, p->stats.padded_flops = padded, p->stats.bytes = bytes;
    *out = p;
    return QK_OK;
  }
  {
    const char* e = std::getenv("QK_PLAN_XCD");
    if (block_arg <= 0 && !(e && std::atoi(e) == 0)) {
      int T = 8;
      if (const char* te = std::getenv("QK_PLAN_TILE")) T = std::max(1, std::min(64, std::atoi(te)));
      plan_tiled(p, n_sites, nx, x_dims, ny, y_dims, sym, orient, world_size, rank, T);
      *out = p;
      return QK_OK;
    }
  }
  std::vector<Item> tile;
  const int stride = n_sites + 1;
  int64_t t = 0;  // running index in the global order
  std::vector<int64_t> per_rank(world_size, 0);
  std::vector<int32_t> tile_of;  // locality tile of each pair of this rank
  double flops = 0, padded = 0, bytes = 0, fit_two = 0, fit_narrow = 0;
  const int nbx = (nx + block - 1) / block, nby = (ny + block - 1) / block;
  for (int bj = 0; bj < nby; ++bj)
    for (int bi = 0; bi < nbx; ++bi) {
      if (sym && bi > bj) continue;
      tile.clear();
      for (int j = bj * block; j < std::min(ny, (bj + 1) * block); ++j)
        for (int i = bi * block; i < std::min(nx, (bi + 1) * block); ++i) {
          if (sym && i > j) continue;
          double f, fp, by;
          int xi = i, yj = j;  // the cheaper order of contraction: which state plays Y (QK_PLAN_ORIENT)
          if (orient && i != j && fused_cost(n_sites, x_dims + (int64_t)j * stride, y_dims + (int64_t)i * stride) < fused_cost(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride))
            xi = j, yj = i;
          pair_work(n_sites, x_dims + (int64_t)xi * stride, y_dims + (int64_t)yj * stride, &f, &fp, &by);
          tile.push_back({xi, yj, (float)fp});
        }
      std::stable_sort(tile.begin(), tile.end(), [](const Item& u, const Item& v) { return u.cost > v.cost; });
      for (const Item& it : tile) {
        // serpentine deal (0..W-1, W-1..0, ...): in a cost-sorted run plain round-robin would hand rank 0 the
        // heaviest pair of every W (13 % more flops than rank W-1 on cfg4 at W = 8)
        const int64_t u = t % (2 * (int64_t)world_size);
        const int r = (int)(u < world_size ? u : 2 * (int64_t)world_size - 1 - u);
        ++per_rank[r];
        if (r == rank) {
          p->pairs.push_back(it.i);
          p->pairs.push_back(it.j);
          tile_of.push_back(bj * nbx + bi);
          double f, fp, by, ft, fn;
          pair_work(n_sites, x_dims + (int64_t)it.i * stride, y_dims + (int64_t)it.j * stride, &f, &fp, &by, &ft, &fn);
          flops += f, padded += fp, bytes += by, fit_two += ft, fit_narrow += fn;
        }
        ++t;
      }
    }
  // Regroup this rank's share: pairs that share the x state are made contiguous and cut into
  // groups of at most QK_GROUP (default 4) pairs -- one workgroup sweeps a group in lockstep so
  // that A_i is read once per group and the per-phase latencies are shared.  Groups are then
  // ordered by decreasing cost (longest first for the device-side queue) -- inside their locality tile when the plan has
  // tiles (`block`): the queue then walks the Gram tile by tile.
  {
    int G = 4;
    if (const char* e = std::getenv("QK_GROUP")) G = std::max(1, std::min(4, std::atoi(e)));
    p->group = G;
    const int64_t np = (int64_t)p->pairs.size() / 2;
    std::vector<Item> mine((size_t)np);
    // Two classes of pairs: a set of states of very different entanglement (the 60-qubit x 6-layer set: largest bond 40 ... 248,
    // median 78) holds pairs that are best swept by one 12-wave workgroup per CU next to pairs whose sites all fit the
    // smaller LDS buffer and are best swept two workgroups per CU.  Class-1 pairs (>= QK_PLAN_SPLIT, default 0.75, of their
    // padded work fits the smaller buffer) are listed behind the others; qk_gram_values may sweep the two runs with the two
    // shapes of the site-fused kernel.
    double split = 0.75;
    if (const char* e = std::getenv("QK_PLAN_SPLIT")) split = std::atof(e);
    double small_work = 0;
    for (int64_t q = 0; q < np; ++q) {
      double f, fp, by, ft;
      const int i = p->pairs[2 * q], j = p->pairs[2 * q + 1];
      pair_work(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride, &f, &fp, &by, &ft);
      const int cls = (fp > 0 && ft >= split * fp) ? 1 : 0;
      if (cls) small_work += fp, p->second.pairs += 1, p->second.flops += f, p->second.padded_flops += fp, p->second.bytes += by;
      mine[(size_t)q] = {i, j, (float)fp, tile_of[(size_t)q], cls};
    }
    if (small_work < 0.05 * padded || small_work > 0.95 * padded) {  // (nearly) one class: no split
      for (Item& it : mine) it.cls = 0;
      p->second = qk_stats{};
    }
    std::stable_sort(mine.begin(), mine.end(), [](const Item& u, const Item& v) {
      return u.cls != v.cls ? u.cls < v.cls : u.tile != v.tile ? u.tile < v.tile : u.i != v.i ? u.i < v.i : u.cost > v.cost;
    });
    struct Grp {
      int64_t start;
      int count;
      double cost;
      int32_t tile, cls;
    };
    std::vector<Grp> grp;
    for (int64_t q = 0; q < np;) {
      int c = 1;
      double cost = mine[(size_t)q].cost;
      while (c < G && q + c < np && mine[(size_t)(q + c)].i == mine[(size_t)q].i && mine[(size_t)(q + c)].tile == mine[(size_t)q].tile && mine[(size_t)(q + c)].cls == mine[(size_t)q].cls)
        cost += mine[(size_t)(q + c)].cost, ++c;
      grp.push_back({q, c, cost, mine[(size_t)q].tile, mine[(size_t)q].cls});
      q += c;
    }
    std::stable_sort(grp.begin(), grp.end(), [](const Grp& u, const Grp& v) { return u.cls != v.cls ? u.cls < v.cls : u.tile != v.tile ? u.tile < v.tile : u.cost > v.cost; });
    p->pairs.clear();
    p->n_first = np;
    for (const Grp& gq : grp) {
      if (gq.cls == 1 && p->n_first == np) p->n_first = (int64_t)p->pairs.size() / 2;  // where the class-1 run starts
      p->groups.push_back((int32_t)(p->pairs.size() / 2));
      p->groups.push_back(gq.count);
      for (int c = 0; c < gq.count; ++c) {
        p->pairs.push_back(mine[(size_t)(gq.start + c)].i);
        p->pairs.push_back(mine[(size_t)(gq.start + c)].j);
      }
    }
  }
  p->total_pairs = t;
  p->max_per_rank = *std::max_element(per_rank.begin(), per_rank.end());
  p->stats.pairs = (int64_t)p->pairs.size() / 2;
  p->stats.flops = flops, p->stats.padded_flops = padded, p->stats.bytes = bytes;
  p->fit_two = padded > 0 ? fit_two / padded : 1.0;
  p->fit_narrow = padded > 0 ? fit_narrow / padded : 1.0;
  p->nq = 1;  // the flat list: one queue per launch
  p->edge_k = choose_edge_k(n_sites, x_dims, y_dims, p->pairs);
  *out = p;
  return QK_OK;
}

extern "C" int qk_plan_destroy(qk_plan* plan) {
  if (!plan) return QK_OK;
  if (plan->d_pairs) (void)hipFree(plan->d_pairs);
  if (plan->d_groups) (void)hipFree(plan->d_groups);
  delete plan;
  return QK_OK;
}
extern "C" int64_t qk_plan_num_pairs(const qk_plan* p) { return p ? (int64_t)p->pairs.size() / 2 : 0; }
extern "C" int64_t qk_plan_total_pairs(const qk_plan* p) { return p ? p->total_pairs : 0; }
extern "C" int64_t qk_plan_max_pairs_per_rank(const qk_plan* p) { return p ? p->max_per_rank : 0; }
extern "C" const int32_t* qk_plan_pairs(const qk_plan* p) { return p ? p->pairs.data() : nullptr; }
extern "C" int64_t qk_plan_first_run(const qk_plan* p) { return p ? (p->n_first > 0 ? p->n_first : (int64_t)p->pairs.size() / 2) : 0; }
extern "C" int32_t qk_plan_edge_sites(const qk_plan* p) { return p ? p->edge_k : 0; }
extern "C" int qk_plan_queues(const qk_plan* p, int64_t* qstart) {
  if (!p) return 0;
  if (qstart)
    for (int s = 0; s <= QK_NQ_MAX; ++s) qstart[s] = p->nq > 1 ? p->qstart[s] : (s == 0 ? 0 : (int64_t)p->pairs.size() / 2);
  return p->nq;
}
extern "C" int qk_plan_stats(const qk_plan* p, qk_stats* out) {
  if (!p || !out) return fail(QK_EINVAL, "qk_plan_stats: null argument");
  *out = p->stats;
  return QK_OK;
}

// ----------------------------------------------------------------------------------------
// small kernels of the product path (the sweep kernels are in qk_ring.h)
// ----------------------------------------------------------------------------------------
__global__ void qk_convert_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, const long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) dst[e] = (float)src[e];
}

__global__ void qk_scatter_kernel(const int32_t* __restrict__ pairs, const double* __restrict__ vals, long long n,
                                  double* __restrict__ K, long long ld, int mirror) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = pairs[2 * t], j = pairs[2 * t + 1];
  if (i < 0) return;  // padding entry of an all-gathered list
  const double v = vals[t];
  K[(long long)j * ld + i] = v;
  if (mirror) K[(long long)i * ld + j] = v;
}

// Edge blocks of a set (SweepArgs.edge_k; qk_fused.h): per state the first k sites contracted into L[s][a] (s = the configuration of
// the first k physical legs, row index built as 2 s + p site by site; a = bond k, padded) and the last k sites into R[s][a] (a = bond
// n - k).  One workgroup per (state, side) at a time; a step multiplies the block so far by one site tensor (split planes of the
// set image), ping-pong between two scratch buffers of the workgroup, the last step writes the destination.  Interleaved complex.
typedef double qk_v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void qk_edge_kernel(const double* __restrict__ data, const int32_t* __restrict__ dims, const int64_t* __restrict__ offs, const int n_sites, const int k,
                                                      const long long n_states, qk_v2d* __restrict__ edge, const long long* __restrict__ edge_offs, qk_v2d* __restrict__ tmp,
                                                      const long long tmp_elems) {
  qk_v2d* const t0 = tmp + (long long)blockIdx.x * 2 * tmp_elems;
  qk_v2d* const t1 = t0 + tmp_elems;
  const int n1 = n_sites + 1;
  for (long long t = blockIdx.x; t < 2 * n_states; t += gridDim.x) {
    const long long st = t >> 1;
    const int right = (int)(t & 1);
    const int32_t* d = dims + st * n1;
    qk_v2d* const dst = edge + edge_offs[t];
    // the block before the first step: one row, 1 at [0][0] (the boundary bond is 1, padded to 16)
    for (int e = threadIdx.x; e < 16; e += blockDim.x) t0[e] = (qk_v2d){e == 0 ? 1.0 : 0.0, 0.0};
    __syncthreads();
    const qk_v2d* in = t0;
    for (int jj = 0; jj < k; ++jj) {
      const int site = right ? n_sites - 1 - jj : jj;
      const int lp = d[site], rp = d[site + 1];  // padded bonds of the site tensor [lp][2][rp]
      const double* re = data + offs[st * n_sites + site];
      const double* im = re + (long long)lp * 2 * rp;
      const int ld_in = right ? rp : lp, ld_out = right ? lp : rp, rows_in = 1 << jj;
      qk_v2d* const out = (jj + 1 == k) ? dst : (in == t0 ? t1 : t0);
      for (int e = threadIdx.x; e < 2 * rows_in * ld_out; e += blockDim.x) {
        const int row = e / ld_out, c = e - row * ld_out, s_ = row >> 1, pp = row & 1;
        double ar = 0, ai = 0;
        for (int l = 0; l < ld_in; ++l) {
          const qk_v2d x = in[(long long)s_ * ld_in + l];
          // left: A[l][pp][c];  right: A[c][pp][l]
          const long long idx = right ? ((long long)(c * 2 + pp) * rp + l) : ((long long)(l * 2 + pp) * rp + c);
          const double tr = re[idx], ti = im[idx];
          ar += x.x * tr - x.y * ti, ai += x.x * ti + x.y * tr;
        }
        out[e] = (qk_v2d){ar, ai};
      }
      __syncthreads();
      in = out;
    }
  }
}

// Merged image of a set (SweepArgs.merge_steps; qk_fused.h): step t of state st = the chain's sites s = k + 2 t and s + 1 contracted over
// the bond between them, M[l][2 p1 + p2][r] = sum_m A_s[l][p1][m] A_{s+1}[m][p2][r] (padded bonds; the padding of the planes is zero,
// so is M's).  One workgroup per step at a time, a thread per (l, r) with the four physical combinations in registers; reads the
// split planes, writes interleaved complex.  Once per set: not part of a sweep.
__global__ __launch_bounds__(256) void qk_merge_kernel(const double* __restrict__ data, const int32_t* __restrict__ dims, const int64_t* __restrict__ offs, const int n_sites, const int k,
                                                       const int steps, const long long n_states, qk_v2d* __restrict__ out, const int64_t* __restrict__ out_offs) {
  const int n1 = n_sites + 1;
  for (long long t = blockIdx.x; t < n_states * steps; t += gridDim.x) {
    const long long st = t / steps;
    const int step = (int)(t - st * steps), s_ = k + 2 * step;
    const int32_t* d = dims + st * n1;
    qk_v2d* const dst = out + (out_offs[t] >> 1);
    const int lp = d[s_], mp = d[s_ + 1], rp = d[s_ + 2];
    const double* re1 = data + offs[st * n_sites + s_];
    const double* im1 = re1 + (long long)lp * 2 * mp;
    const double* re2 = data + offs[st * n_sites + s_ + 1];
    const double* im2 = re2 + (long long)mp * 2 * rp;
    for (int e = threadIdx.x; e < lp * rp; e += blockDim.x) {
      const int l = e / rp, r = e - l * rp;
      double ar[4] = {0, 0, 0, 0}, ai[4] = {0, 0, 0, 0};
      const double* a0r = re1 + (long long)(2 * l) * mp;  // A_s[l][0][.], [l][1][.] follows at + mp
      const double* a0i = im1 + (long long)(2 * l) * mp;
      for (int m = 0; m < mp; ++m) {
        const double xr[2] = {a0r[m], a0r[mp + m]}, xi[2] = {a0i[m], a0i[mp + m]};
        const long long o2 = (long long)(2 * m) * rp + r;  // A_{s+1}[m][0][r], [m][1][r] at + rp
        const double yr[2] = {re2[o2], re2[o2 + rp]}, yi[2] = {im2[o2], im2[o2 + rp]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int p1 = c >> 1, p2 = c & 1;
          ar[c] += xr[p1] * yr[p2] - xi[p1] * yi[p2];
          ai[c] += xr[p1] * yi[p2] + xi[p1] * yr[p2];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[(long long)(4 * l + c) * rp + r] = (qk_v2d){ar[c], ai[c]};
    }
  }
}

#ifdef QK_LAB
__global__ void qk_lab_fold_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst, const long long n, const long long win) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = (src[i] % win) & ~1ll;
}
#endif
// self-test: C[16x16] = sum_{k<16} P[k][m] * Q[k][n] with the fragment maps used above
__global__ void qk_selftest_f32_kernel(const float* __restrict__ P, const float* __restrict__ Q, float* __restrict__ C) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  v4f acc = {0, 0, 0, 0};
  for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(P[(4 * ks + q) * 16 + j], Q[(4 * ks + q) * 16 + j], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * q + r) * 16 + j] = acc[r];
}

__global__ void qk_selftest_kernel(const double* __restrict__ P, const double* __restrict__ Q, double* __restrict__ C) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  v4d acc = {0, 0, 0, 0};
  for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(4 * ks + q) * 16 + j], Q[(4 * ks + q) * 16 + j], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(q + 4 * r) * 16 + j] = acc[r];
}

// ----------------------------------------------------------------------------------------
// host API
// ----------------------------------------------------------------------------------------
extern "C" int qk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int ctx_init(qk_ctx* c, int device_id, int num_cus);
static void free_merged(qk_mps_set* m);

extern "C" int qk_ctx_create(int device_id, qk_ctx** out) {
  if (!out) return fail(QK_EINVAL, "qk_ctx_create: null out");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(QK_EDEVICE, "qk_ctx_create: no HIP device available (%s); this engine has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(QK_EINVAL, "qk_ctx_create: device %d out of range [0,%d)", device_id, n);
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(QK_EDEVICE, "qk_ctx_create: device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
  qk_ctx* c = new (std::nothrow) qk_ctx;
  if (!c) return fail(QK_ENOMEM, "qk_ctx_create: out of memory");
  const int rc_init = ctx_init(c, device_id, prop.multiProcessorCount);
  if (rc_init != QK_OK) {
    qk_ctx_destroy(c);  // releases whatever was created before the failure
    return rc_init;
  }
  *out = c;
  return QK_OK;
}

static int ctx_init(qk_ctx* c, int device_id, int num_cus) {
  c->device = device_id;
  c->num_cus = num_cus;
  HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  HIP_TRY(hipEventCreate(&c->ev_mid));
  HIP_TRY(hipMalloc(&c->counter, (QK_NQ_MAX * QK_QSTRIDE + 8) * sizeof(unsigned long long)));  // queue heads (8 per launch of a split sweep), tail clocks
  HIP_TRY(hipMalloc(&c->prof, 8 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(c->prof, 0, 8 * sizeof(unsigned long long)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_ring_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_ring_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_small_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_small_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(QKF_KERNEL_ONE), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(QKF_KERNEL_TWO), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(QKF_KERNEL_DUAL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef QK_LAB  // libqklab.so only: the experimental kernels of qk_lab.hip, selectable with QK_VARIANT
  {
    const int rc = qk_lab_init(c);
    if (rc != QK_OK) return rc;
  }
  if (const char* v = std::getenv("QK_VARIANT")) c->variant = std::atoi(v);
#endif
  if (const char* v = std::getenv("QK_SMALL")) c->small_path = std::atoi(v) != 0;
  if (const char* v = std::getenv("QK_WAVE")) c->wave_path = std::atoi(v) != 0;
  if (const char* v = std::getenv("QK_WAVE2")) c->wave2_path = std::atoi(v) != 0, c->wave2_ring = std::atoi(v) != 2;
  if (const char* v = std::getenv("QK_FUSED")) c->fused_path = std::atoi(v);
  if (const char* v = std::getenv("QK_MERGE")) c->merge_sites = std::atoi(v) != 0;
  // QK_DETERMINISTIC=1: bit-reproducible Grams.  The site-fused sweep sums the tiles of a column with LDS atomics in arrival
  // order (two launches on the same inputs differ in the last bits, <= 9e-16); the ring sweep, the small-bond sweep and the
  // one-wave sweeps add in a fixed order.  So the fused sweep is taken out of the selection (sets with a bond > 32 run the ring
  // sweep: 518 instead of 412 ms on the headline set) and the order in which workgroups pull pairs no longer matters.
  if (const char* v = std::getenv("QK_DETERMINISTIC"))
    if (std::atoi(v) != 0) c->fused_path = 0, c->deterministic = true;
  if (const char* v = std::getenv("QK_FUSED_SPLIT")) c->fused_split = std::atoi(v) != 0;
  if (const char* v = std::getenv("QK_FUSED_WGS")) c->fused_wgs = std::max(0, std::min(2, std::atoi(v)));
  if (const char* v = std::getenv("QK_WGS_PER_CU")) c->wgs_per_cu = std::max(1, std::min(4, std::atoi(v)));
  return QK_OK;
}

extern "C" int qk_ctx_destroy(qk_ctx* c) {
  if (!c) return QK_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->build_arena) (void)hipFree(c->build_arena);
  if (c->build_work) (void)hipFree(c->build_work);
  if (c->counter) (void)hipFree(c->counter);
  if (c->prof) (void)hipFree(c->prof);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return QK_OK;
}

extern "C" int qk_ctx_trim(qk_ctx* c) {
  if (!c) return fail(QK_EINVAL, "qk_ctx_trim: null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->build_arena) (void)hipFree(c->build_arena);
  if (c->build_work) (void)hipFree(c->build_work);
  c->scratch = nullptr, c->scratch_bytes = 0;
  c->build_arena = nullptr, c->build_arena_bytes = 0;
  c->build_work = nullptr, c->build_work_bytes = 0;
  return QK_OK;
}

extern "C" int qk_ctx_set_stream(qk_ctx* c, void* s) {
  if (!c) return fail(QK_EINVAL, "qk_ctx_set_stream: null context");
  c->stream = reinterpret_cast<hipStream_t>(s);  // NULL = HIP's null stream
  return QK_OK;
}

extern "C" int qk_ctx_use_own_stream(qk_ctx* c) {
  if (!c) return fail(QK_EINVAL, "qk_ctx_use_own_stream: null context");
  c->stream = c->own_stream;
  return QK_OK;
}

extern "C" int qk_ctx_synchronize(qk_ctx* c) {
  if (!c) return fail(QK_EINVAL, "qk_ctx_synchronize: null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QK_OK;
}

extern "C" int qk_mps_set_create(qk_ctx* c, int32_t n_states, int32_t n_sites, const int32_t* bond_dims,
                                 const double* const* site_tensors, int32_t layout, qk_mps_set** out) {
  if (!c || !out || !bond_dims || !site_tensors) return fail(QK_EINVAL, "qk_mps_set_create: null argument");
  if (n_states <= 0 || n_sites <= 0) return fail(QK_EINVAL, "qk_mps_set_create: empty set (%d states, %d sites)", n_states, n_sites);
  QkRangeGuard range_("qk:upload");
  HIP_TRY(hipSetDevice(c->device));
  const int stride = n_sites + 1;
  std::vector<int32_t> pad((size_t)n_states * stride);
  std::vector<int64_t> offs((size_t)n_states * n_sites);
  std::vector<int64_t> state_off(n_states + 1, 0);
  int max_pad = 0;
  for (int s = 0; s < n_states; ++s) {
    const int32_t* d = bond_dims + (size_t)s * stride;
    if (d[0] != 1 || d[n_sites] != 1) return fail(QK_EINVAL, "qk_mps_set_create: state %d: boundary bonds must be 1", s);
    for (int k = 0; k <= n_sites; ++k) {
      if (d[k] <= 0) return fail(QK_EINVAL, "qk_mps_set_create: state %d: non-positive bond %d", s, k);
      pad[(size_t)s * stride + k] = pad16(d[k]);
      max_pad = std::max(max_pad, pad16(d[k]));
    }
    state_off[s + 1] = state_off[s] + qk_pack_state_size(n_sites, d);
  }
  qk_mps_set* m = new (std::nothrow) qk_mps_set;
  if (!m) return fail(QK_ENOMEM, "qk_mps_set_create: out of memory");
  m->ctx = c, m->n_states = n_states, m->n_sites = n_sites, m->max_pad = max_pad;
  m->dims_true.assign(bond_dims, bond_dims + (size_t)n_states * stride);
  const int64_t total = state_off[n_states];
  m->bytes = total * (int64_t)sizeof(double);
  hipError_t e = hipMalloc(&m->d_data, (size_t)m->bytes);
  if (e != hipSuccess) {
    delete m;
    return fail(QK_EDEVICE, "qk_mps_set_create: hipMalloc of %lld bytes failed: %s", (long long)m->bytes, hipGetErrorString(e));
  }
  std::vector<double> stage;
  std::vector<int64_t> so(n_sites);
  for (int s = 0; s < n_states; ++s) {
    const int64_t sz = state_off[s + 1] - state_off[s];
    stage.resize((size_t)sz);
    int rc = qk_pack_state(n_sites, bond_dims + (size_t)s * stride, site_tensors + (size_t)s * n_sites, layout, stage.data(), so.data());
    if (rc != QK_OK) {
      (void)hipFree(m->d_data);
      delete m;
      return rc;
    }
    for (int k = 0; k < n_sites; ++k) offs[(size_t)s * n_sites + k] = state_off[s] + so[k];
    e = hipMemcpy(m->d_data + state_off[s], stage.data(), (size_t)sz * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(m->d_data);
      delete m;
      return fail(QK_EDEVICE, "qk_mps_set_create: upload failed: %s", hipGetErrorString(e));
    }
  }
  e = hipMalloc(&m->d_dims, pad.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_offs, offs.size() * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_true, pad.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(m->d_dims, pad.data(), pad.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_true, bond_dims, pad.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_offs, offs.data(), offs.size() * sizeof(int64_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    qk_mps_set_destroy(m);
    return fail(QK_EDEVICE, "qk_mps_set_create: table upload failed: %s", hipGetErrorString(e));
  }
  *out = m;
  return QK_OK;
}

extern "C" int qk_mps_set_destroy(qk_mps_set* m) {
  if (!m) return QK_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->d_data) (void)hipFree(m->d_data);
  if (m->d_il) (void)hipFree(m->d_il);
  if (m->d_edge) (void)hipFree(m->d_edge);
  if (m->d_edge_offs) (void)hipFree(m->d_edge_offs);
  free_merged(m);
  if (m->d_dims) (void)hipFree(m->d_dims);
  if (m->d_true) (void)hipFree(m->d_true);
  if (m->d_offs) (void)hipFree(m->d_offs);
  delete m;
  return QK_OK;
}

extern "C" int qk_mps_set_info(const qk_mps_set* m, int32_t* n_states, int32_t* n_sites, int32_t* max_padded_bond, int64_t* device_bytes) {
  if (!m) return fail(QK_EINVAL, "qk_mps_set_info: null set");
  if (n_states) *n_states = m->n_states;
  if (n_sites) *n_sites = m->n_sites;
  if (max_padded_bond) *max_padded_bond = m->max_pad;
  if (device_bytes) *device_bytes = m->bytes * (m->d_il ? 2 : 1) + m->edge_bytes + m->mg_bytes;  // the interleaved twin the fused / wave2 sweeps make on first use counts
  return QK_OK;
}

extern "C" int qk_mps_set_precision(const qk_mps_set* m) { return m ? m->precision : 0; }

extern "C" int qk_mps_set_image(const qk_mps_set* m, int64_t* n_doubles, const double** planes_dev, int32_t* dims_true, int64_t* offsets) {
  if (!m) return fail(QK_EINVAL, "qk_mps_set_image: null set");
  if (m->precision != 64) return fail(QK_EINVAL, "qk_mps_set_image: only fp64 sets are exchanged");
  if (n_doubles) *n_doubles = m->bytes / (int64_t)sizeof(double);
  if (planes_dev) *planes_dev = m->d_data;
  if (dims_true) std::memcpy(dims_true, m->dims_true.data(), m->dims_true.size() * sizeof(int32_t));
  if (offsets) {
    HIP_TRY(hipSetDevice(m->ctx->device));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    HIP_TRY(hipMemcpy(offsets, m->d_offs, (size_t)m->n_states * m->n_sites * sizeof(int64_t), hipMemcpyDeviceToHost));
  }
  return QK_OK;
}

extern "C" int qk_mps_set_copy_image(const qk_mps_set* m, double* dst, int64_t n_doubles) {
  if (!m || !dst) return fail(QK_EINVAL, "qk_mps_set_copy_image: null argument");
  if (m->precision != 64) return fail(QK_EINVAL, "qk_mps_set_copy_image: only fp64 sets are exchanged");
  if (n_doubles * (int64_t)sizeof(double) < m->bytes) return fail(QK_EINVAL, "qk_mps_set_copy_image: destination holds %lld doubles, the image has %lld", (long long)n_doubles, (long long)(m->bytes / 8));
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipMemcpyAsync(dst, m->d_data, (size_t)m->bytes, hipMemcpyDefault, m->ctx->stream));
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  return QK_OK;
}

extern "C" int qk_mps_set_from_packed(qk_ctx* c, int32_t n_states, int32_t n_sites, const int32_t* dims_true, const int64_t* offsets,
                                      const double* planes_dev, int64_t n_doubles, qk_mps_set** out) {
  if (!c || !out || !dims_true || !offsets || !planes_dev) return fail(QK_EINVAL, "qk_mps_set_from_packed: null argument");
  if (n_states <= 0 || n_sites <= 0 || n_doubles <= 0) return fail(QK_EINVAL, "qk_mps_set_from_packed: empty set");
  const int stride = n_sites + 1;
  std::vector<int32_t> pad((size_t)n_states * stride);
  int max_pad = 0;
  for (int s = 0; s < n_states; ++s) {
    const int32_t* d = dims_true + (size_t)s * stride;
    if (d[0] != 1 || d[n_sites] != 1) return fail(QK_EINVAL, "qk_mps_set_from_packed: state %d: boundary bonds must be 1", s);
    for (int k = 0; k <= n_sites; ++k) {
      if (d[k] <= 0) return fail(QK_EINVAL, "qk_mps_set_from_packed: state %d: non-positive bond %d", s, k);
      pad[(size_t)s * stride + k] = pad16(d[k]);
      max_pad = std::max(max_pad, pad16(d[k]));
    }
    for (int k = 0; k < n_sites; ++k) {  // every tensor must lie inside the buffer, 16-byte aligned
      const int64_t off = offsets[(size_t)s * n_sites + k], sz = 2ll * pad[(size_t)s * stride + k] * 2 * pad[(size_t)s * stride + k + 1];
      if (off < 0 || (off & 1) || off + sz > n_doubles) return fail(QK_EINVAL, "qk_mps_set_from_packed: state %d site %d: tensor [%lld, %lld) outside the %lld-double image", s, k, (long long)off, (long long)(off + sz), (long long)n_doubles);
    }
  }
  HIP_TRY(hipSetDevice(c->device));
  qk_mps_set* m = new (std::nothrow) qk_mps_set;
  if (!m) return fail(QK_ENOMEM, "qk_mps_set_from_packed: out of memory");
  m->ctx = c, m->n_states = n_states, m->n_sites = n_sites, m->max_pad = max_pad;
  m->dims_true.assign(dims_true, dims_true + (size_t)n_states * stride);
  m->bytes = n_doubles * (int64_t)sizeof(double);
  hipError_t e = hipMalloc(&m->d_data, (size_t)m->bytes);
  if (e == hipSuccess) e = hipMalloc(&m->d_dims, pad.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_true, pad.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_offs, (size_t)n_states * n_sites * sizeof(int64_t));
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_data, planes_dev, (size_t)m->bytes, hipMemcpyDefault, c->stream);  // device or host source
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_dims, pad.data(), pad.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_true, dims_true, pad.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_offs, offsets, (size_t)n_states * n_sites * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    qk_mps_set_destroy(m);
    return fail(QK_EDEVICE, "qk_mps_set_from_packed: %s", hipGetErrorString(e));
  }
  *out = m;
  return QK_OK;
}

extern "C" int qk_mps_set_to_f32(qk_ctx* c, const qk_mps_set* src, qk_mps_set** out) {
  if (!c || !src || !out) return fail(QK_EINVAL, "qk_mps_set_to_f32: null argument");
  if (src->ctx != c) return fail(QK_EINVAL, "qk_mps_set_to_f32: the set belongs to another context");
  if (src->precision != 64) return fail(QK_EINVAL, "qk_mps_set_to_f32: the source set is not fp64");
  HIP_TRY(hipSetDevice(c->device));
  qk_mps_set* m = new (std::nothrow) qk_mps_set;
  if (!m) return fail(QK_ENOMEM, "qk_mps_set_to_f32: out of memory");
  m->ctx = c, m->n_states = src->n_states, m->n_sites = src->n_sites, m->max_pad = src->max_pad, m->precision = 32;
  m->dims_true = src->dims_true;
  const long long n = src->bytes / (long long)sizeof(double);
  m->bytes = n * (long long)sizeof(float);
  const size_t nd = (size_t)src->n_states * (src->n_sites + 1), no = (size_t)src->n_states * src->n_sites;
  hipError_t e = hipMalloc(&m->d_data, (size_t)m->bytes);
  if (e == hipSuccess) e = hipMalloc(&m->d_dims, nd * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_true, nd * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_offs, no * sizeof(int64_t));
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_dims, src->d_dims, nd * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_true, src->d_true, nd * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(m->d_offs, src->d_offs, no * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) {
    qk_convert_f32_kernel<<<dim3(4 * c->num_cus), dim3(256), 0, c->stream>>>(src->d_data, reinterpret_cast<float*>(m->d_data), n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    qk_mps_set_destroy(m);
    return fail(QK_EDEVICE, "qk_mps_set_to_f32: %s", hipGetErrorString(e));
  }
  *out = m;
  return QK_OK;
}

// the interleaved complex128 image of a set, made once on the device from the split planes
static int ensure_interleaved(qk_ctx* c, qk_mps_set* m) {
  if (m->d_il) return QK_OK;
  DevBuf il;  // published only after the conversion has been launched without error (released otherwise)
  HIP_TRY(il.alloc((size_t)m->bytes));
  const long long nt = (long long)m->n_states * m->n_sites;
  const dim3 grid((unsigned)std::min<long long>(nt, 64ll * c->num_cus));
  if (m->precision == 64) qk_interleave_kernel<double><<<grid, dim3(256), 0, c->stream>>>(m->d_data, il.as<double>(), m->d_dims, m->d_offs, m->n_sites, nt);
  else  // complex64 image of an fp32 set (same offsets, counted in floats)
    qk_interleave_kernel<float><<<grid, dim3(256), 0, c->stream>>>(reinterpret_cast<const float*>(m->d_data), il.as<float>(), m->d_dims, m->d_offs, m->n_sites, nt);
  HIP_TRY(hipGetLastError());
  m->d_il = il.as<double>();
  il.p = nullptr;
  return QK_OK;
}

// the edge blocks of a set for `k` sites at either end (made once per set and k; fp64 sets)
static int ensure_edges(qk_ctx* c, qk_mps_set* m, const int k) {
  if (k <= 0 || (m->d_edge && m->edge_k == k)) return QK_OK;
  if (m->d_edge) (void)hipFree(m->d_edge);
  if (m->d_edge_offs) (void)hipFree(m->d_edge_offs);
  m->d_edge = nullptr, m->d_edge_offs = nullptr, m->edge_k = 0, m->edge_bytes = 0;
  const int n = m->n_sites, stride = n + 1;
  std::vector<long long> offs((size_t)m->n_states * 2);
  long long total = 0;
  int maxld = 16;
  for (int s_ = 0; s_ < m->n_states; ++s_) {
    const int32_t* d = m->dims_true.data() + (size_t)s_ * stride;
    offs[(size_t)2 * s_] = total;
    total += (long long)(1 << k) * pad16(d[k]);
    offs[(size_t)2 * s_ + 1] = total;
    total += (long long)(1 << k) * pad16(d[n - k]);
    for (int jj = 0; jj <= k; ++jj) maxld = std::max(maxld, std::max(pad16(d[jj]), pad16(d[n - jj])));
  }
  const long long tmp_elems = (long long)(1 << (k - 1)) * maxld;  // the largest intermediate block
  const int grid = (int)std::min<long long>(2ll * m->n_states, 4ll * c->num_cus);
  DevBuf eb, ob, tb;
  HIP_TRY(eb.alloc((size_t)total * 2 * sizeof(double)));
  HIP_TRY(ob.alloc(offs.size() * sizeof(long long)));
  HIP_TRY(tb.alloc((size_t)grid * 2 * tmp_elems * 2 * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(ob.p, offs.data(), offs.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
  qk_edge_kernel<<<dim3(grid), dim3(256), 0, c->stream>>>(m->d_data, m->d_dims, m->d_offs, n, k, m->n_states, eb.as<qk_v2d>(), ob.as<long long>(), tb.as<qk_v2d>(), tmp_elems);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));  // (the scratch goes out of scope; once per set)
  m->d_edge = eb.as<double>(), m->d_edge_offs = ob.as<long long>();
  eb.p = nullptr, ob.p = nullptr;
  m->edge_k = k, m->edge_bytes = total * 2 * (long long)sizeof(double);
  return QK_OK;
}

static void free_merged(qk_mps_set* m) {
  if (m->d_mg) (void)hipFree(m->d_mg);
  if (m->d_mg_offs) (void)hipFree(m->d_mg_offs);
  m->d_mg = nullptr, m->d_mg_offs = nullptr, m->mg_k = -1, m->mg_steps = 0, m->mg_bytes = 0;
}

// the merged image of a set for a chain that starts `k` sites in (made once per set and k; fp64 sets)
static int ensure_merged(qk_ctx* c, qk_mps_set* m, const int k) {
  if (m->d_mg && m->mg_k == k) return QK_OK;
  free_merged(m);
  const int n = m->n_sites, stride = n + 1, steps = (n - 2 * k) / 2;
  if (steps < 1) return fail(QK_EINVAL, "ensure_merged: a chain of %d sites", n - 2 * k);
  std::vector<int64_t> mo((size_t)m->n_states * steps);
  long long total = 0;  // complex elements
  for (int s_ = 0; s_ < m->n_states; ++s_) {
    const int32_t* d = m->dims_true.data() + (size_t)s_ * stride;
    for (int t = 0; t < steps; ++t) {
      mo[(size_t)s_ * steps + t] = 2 * total;
      total += 4ll * pad16(d[k + 2 * t]) * pad16(d[k + 2 * t + 2]);
    }
  }
  DevBuf ib, ob;
  HIP_TRY(ib.alloc((size_t)total * 2 * sizeof(double)));
  HIP_TRY(ob.alloc(mo.size() * sizeof(int64_t)));
  HIP_TRY(hipMemcpyAsync(ob.p, mo.data(), mo.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  const long long tasks = (long long)m->n_states * steps;
  qk_merge_kernel<<<dim3((unsigned)std::min<long long>(tasks, 16ll * c->num_cus)), dim3(256), 0, c->stream>>>(m->d_data, m->d_dims, m->d_offs, n, k, steps, m->n_states, ib.as<qk_v2d>(),
                                                                                                                  ob.as<int64_t>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));  // (the host table goes out of scope; once per set)
  m->d_mg = ib.as<double>(), m->d_mg_offs = ob.as<int64_t>();
  ib.p = ob.p = nullptr;
  m->mg_k = k, m->mg_steps = steps, m->mg_bytes = total * 2 * (long long)sizeof(double);
  return QK_OK;
}

static int ensure_plan_uploaded(qk_ctx* c, qk_plan* p) {
  if (p->d_pairs && p->up_ctx == c) return QK_OK;
  if (p->d_pairs) {
    (void)hipFree(p->d_pairs);
    p->d_pairs = nullptr;
  }
  if (p->d_groups) {
    (void)hipFree(p->d_groups);
    p->d_groups = nullptr;
  }
  if (p->pairs.empty()) return QK_OK;
  HIP_TRY(hipMalloc(&p->d_pairs, p->pairs.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(p->d_pairs, p->pairs.data(), p->pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&p->d_groups, p->groups.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(p->d_groups, p->groups.data(), p->groups.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  p->up_ctx = c;
  return QK_OK;
}

extern "C" int qk_gram_values(qk_ctx* c, const qk_mps_set* xs, const qk_mps_set* ys, const qk_plan* plan_c, double* values_dev, double* z_dev) {
  if (!c || !xs || !plan_c || !values_dev) return fail(QK_EINVAL, "qk_gram_values: null argument");
  qk_plan* plan = const_cast<qk_plan*>(plan_c);
  if (!ys) ys = xs;
  if (xs->ctx != c || ys->ctx != c) return fail(QK_EINVAL, "qk_gram_values: sets belong to another context");
  if (xs->n_sites != ys->n_sites || xs->n_sites != plan->n_sites) return fail(QK_EINVAL, "qk_gram_values: site counts differ (%d, %d, plan %d)", xs->n_sites, ys->n_sites, plan->n_sites);
  if (plan->nx != xs->n_states || plan->ny != ys->n_states) return fail(QK_EINVAL, "qk_gram_values: plan is for %dx%d states, sets hold %dx%d", plan->nx, plan->ny, xs->n_states, ys->n_states);
  QkRangeGuard range_("qk:sweep");
  HIP_TRY(hipSetDevice(c->device));
  const long long np = (long long)plan->pairs.size() / 2;
  c->last = plan->stats;
  c->split_pending = false;
  c->last.max_bond = std::max(xs->max_pad, ys->max_pad);
  c->last.kernel_ms = 0;
  c->last.grid = 0;
  c->last.kernel = QK_KERNEL_NONE;
  c->last.precision = xs->precision;
  if (np == 0) return QK_OK;
  int rc = ensure_plan_uploaded(c, plan);
  if (rc != QK_OK) return rc;

  if (xs->precision != ys->precision) return fail(QK_EINVAL, "qk_gram_values: the two sets differ in precision (fp%d, fp%d)", xs->precision, ys->precision);
  const bool f32 = (xs->precision == 32);
  if (f32) c->last.bytes *= 0.5;  // complex64 planes
  const bool quad = plan->quad;
  const bool grouped = (c->variant == 14) && !f32 && !quad;
  const bool duo = (c->variant == 16) && !f32 && !quad;
  const long long members = grouped ? GMAX : 1;        // pairs stacked in one X/T buffer (the quad kernel doubles the planes itself)
  const long long chains = quad ? 4 : (duo ? 2 : 1);  // X/T buffer sets per workgroup (quad: 2 stacked sets = 4 single ones)
  const long long x_plane = members * xs->max_pad * ys->max_pad;
  const long long t_plane = 2 * x_plane;
  const long long units = quad ? np / 4 : grouped ? (long long)plan->groups.size() / 2 : (duo ? (np + 1) / 2 : np);
  const int max_pad = std::max(xs->max_pad, ys->max_pad);
  // the site-fused sweep (qk_fused.h), fp64.  Two shapes: one 12-wave workgroup per CU (three waves per SIMD, two T slots
  // each) with an 8192-element X buffer, or two 8-wave workgroups (four waves per SIMD, one slot) with 4608 elements each.  The second workgroup fills the first one's barriers and per-site set-up
  // (+24 % on the 40-qubit x 4-layer set), but every site that does not fit the smaller buffer runs in strips from a global
  // X: on the 60-qubit x 6-layer headline set (57 % of the work fits) the two shapes are within 2 % in time while the
  // smaller buffer moves 3.2 instead of 1.9 TB through the fabric -- so two workgroups only when >= 75 % of the padded
  // work fits.  A 16-row strip of X' must fit the buffer: bonds <= XCAP / 16.
  const size_t lds_meta = 32 + (size_t)xs->n_sites * (48 + 16);  // queue slot, the overlap's accumulator, per-site records and tensor offsets
  const bool fused_ok = c->variant == 20 && !f32 && !quad && c->fused_path != 0 && max_pad > (c->fused_path >= 2 ? 16 : 32);
  const bool can_one = max_pad <= QKF_XCAP_ONE / TILE && (size_t)QKF_XCAP_ONE * 16 + lds_meta <= 160 * 1024;
  const bool can_two = max_pad <= QKF_XCAP_TWO / TILE && (size_t)QKF_XCAP_TWO * 16 + lds_meta <= 80 * 1024;
  const bool fused = fused_ok && (can_one || can_two);
  // two runs of pairs, two shapes (see qk_plan_create): only when the launch is free to choose its shape
  const bool two_runs = fused && can_one && can_two && c->fused_wgs == 0 && c->fused_split && !plan->second_wave2 && plan->n_first > 0 && plan->n_first < np;
  // a mixed set: the plan's second run holds the pairs of two small states (every bond <= 32) for the one-wave sweep
  const bool mixed = fused && plan->second_wave2 && c->wave2_path && c->wave2_ring && plan->n_first > 0 && plan->n_first < np;
  // One class of pairs: the two-workgroup shape when the work sits in sites that fit its buffer AND most of it in sites of at most
  // the narrow size -- from about 4 x 4 tiles per site on the 12-wave dual shape is the faster one although the site would still fit
  // (uniform chains of bond 64, i.e. what a bond cap of 64 produces: dual against two workgroups measured in tools/uniform_ab.py)
  const bool fused_two = fused && can_two && !two_runs && (!can_one || c->fused_wgs == 2 || (c->fused_wgs == 0 && plan->fit_two >= 0.75 && plan->fit_narrow >= 0.5));
  const size_t lds_fused = (size_t)(fused_two ? QKF_XCAP_TWO : QKF_XCAP_ONE) * 16 + lds_meta;
  const int grid = (int)std::min<long long>(units, (long long)(fused ? (fused_two ? 2 : 1) : c->wgs_per_cu) * c->num_cus);
  const char* dual_env = std::getenv("QK_FUSED_DUAL");
  // the 12-wave shape comes in two forms; the dual one (pairs of tiles per wave) is the default (QK_FUSED_DUAL=0: single tiles)
  const bool dual = fused && !fused_two && (dual_env ? std::atoi(dual_env) != 0 : true);
  const bool split = two_runs;
  const size_t need = (size_t)(split ? 2 * c->num_cus : grid) * (size_t)chains * 2 * (size_t)(x_plane + t_plane) * sizeof(double);
  if (need > c->scratch_bytes) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->scratch) HIP_TRY(hipFree(c->scratch));
    c->scratch = nullptr, c->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&c->scratch, need));
    c->scratch_bytes = need;
  }
  SweepArgs a;
  a.xdata = xs->d_data, a.xdims = xs->d_dims, a.xtrue = xs->d_true, a.xoffs = xs->d_offs;
  a.ydata = ys->d_data, a.ydims = ys->d_dims, a.ytrue = ys->d_true, a.yoffs = ys->d_offs;
  a.n_sites = xs->n_sites;
  a.pairs = plan->d_pairs, a.npairs = np;
  a.groups = plan->d_groups, a.ngroups = (long long)plan->groups.size() / 2;
  a.values = values_dev, a.z = z_dev;
  a.scratch = c->scratch, a.x_plane = x_plane, a.t_plane = t_plane;
  a.xedge = a.yedge = nullptr, a.xedge_offs = a.yedge_offs = nullptr, a.edge_k = 0;
  a.xmg = a.ymg = nullptr, a.xmg_offs = a.ymg_offs = nullptr, a.merge_steps = 0;
  a.counter = c->counter;
  a.nq = 1;  // kernels with XCD queues (site-fused, wave2) get the plan's queues below
  for (int s_ = 0; s_ <= QK_NQ_MAX; ++s_) a.qstart[s_] = plan->nq > 1 ? plan->qstart[s_] : (s_ == 0 ? 0 : np);
  // device clocks for the tail accounting, behind the queue heads: launch 1 uses [0] [1] [4], launch 2 [2] [3] [6]
  unsigned long long* const tail = c->counter + QK_NQ_MAX * QK_QSTRIDE;
  a.tail = tail;
  a.prof = c->prof;
  a.debug_flags = 0, a.prio_mode = 0;
#ifdef QK_LAB  // timing experiments of the lab kernels (they give wrong results by construction): libqklab.so only
  if (const char* v = std::getenv("QK_DEBUG_FLAGS")) a.debug_flags = std::atoi(v);
  if (const char* v = std::getenv("QK_PRIO")) a.prio_mode = std::atoi(v);
#endif
  HIP_TRY(hipMemsetAsync(c->counter, 0, (QK_NQ_MAX * QK_QSTRIDE + 8) * sizeof(unsigned long long), c->stream));
  HIP_TRY(hipMemsetAsync(tail, 0xFF, 4 * sizeof(unsigned long long), c->stream));  // the four minima
  c->last.queues = 1, c->last.tail_frac = c->last.second_tail_frac = 0;
  c->tail_pending = false;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  int launched_grid = grid;
  // per-pair site metadata in LDS behind the three ring slots: 4 (n+1) ints + 2 n int64 (+ alignment)
  const size_t lds_ring = 3 * 16 * 1024 + 16 + (size_t)(4 * (xs->n_sites + 1) + 2) * sizeof(int) + (size_t)2 * xs->n_sites * sizeof(long long);
  if (lds_ring > 80 * 1024) return fail(QK_EINVAL, "qk_gram_values: %d sites need %zu bytes of LDS per workgroup (limit 80 KiB for 2 workgroups per CU)", xs->n_sites, lds_ring);
  const size_t esz = f32 ? sizeof(float) : sizeof(double);
  const size_t lds_small = (size_t)(3 * 2 * (64 / esz) * 64 + 6 * 32 * 32) * esz + 16 + (size_t)(4 * (xs->n_sites + 1) + 2) * sizeof(int) + (size_t)2 * xs->n_sites * sizeof(long long);
  if (quad) {  // 2x2 blocks of pairs per workgroup (QK_PLAN_QUADS plans): an experimental kernel of the lab library
#ifdef QK_LAB
    const int rc_quad = qk_lab_launch_quad(c, a, grid, xs->n_sites, f32);
    if (rc_quad != QK_OK) return rc_quad;
    c->last.kernel = QK_KERNEL_LAB;
#else
    return fail(QK_EINVAL, "qk_gram_values: QK_PLAN_QUADS plans are swept by an experimental kernel that only libqklab.so contains");
#endif
  } else if (c->variant == 20 && c->wave_path && !f32 && std::max(xs->max_pad, ys->max_pad) <= 16) {
    // every bond <= 16: a pair lives in the registers of one wavefront (qk_sweep_wave_kernel); 16 waves per CU
    const int wgrid = (int)std::min<long long>(np, 16ll * c->num_cus);
    qk_sweep_wave_kernel<0><<<dim3(wgrid), dim3(64), 0, c->stream>>>(a);
    launched_grid = wgrid, c->last.kernel = QK_KERNEL_WAVE;
  } else if (!fused && c->variant == 20 && c->wave2_path && (!f32 || c->wave2_ring) && std::max(xs->max_pad, ys->max_pad) <= 32) {
    // every bond <= 32, fp64: a pair lives in the registers of one wavefront as 2 x 2 tiles (qk_sweep_wave2_kernel); 8 waves per CU
    for (const qk_mps_set* m : {xs, ys}) {
      const int rc_il = ensure_interleaved(c, const_cast<qk_mps_set*>(m));
      if (rc_il != QK_OK) return rc_il;
    }
    a.xdata = xs->d_il, a.ydata = ys->d_il;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));  // the conversion above is not part of the sweep
    const int wgrid = (int)std::min<long long>(np, 8ll * c->num_cus);
    a.nq = plan->nq, c->last.queues = plan->nq > 1 ? 8 : 1, c->tail_pending = true;
    if (f32) qk_sweep_wave2_kernel<3, float><<<dim3(wgrid), dim3(64), 0, c->stream>>>(a);  // complex64 storage, fp64 arithmetic
    else if (c->wave2_ring) qk_sweep_wave2_kernel<3, double><<<dim3(wgrid), dim3(64), 0, c->stream>>>(a);
    else qk_sweep_wave2_kernel<0, double><<<dim3(wgrid), dim3(64), 0, c->stream>>>(a);
    launched_grid = wgrid, c->last.kernel = (f32 || c->wave2_ring) ? QK_KERNEL_WAVE2 : QK_KERNEL_WAVE2_PLAIN;
  } else if (!fused && c->variant == 20 && c->small_path && std::max(xs->max_pad, ys->max_pad) <= 32 && lds_small <= 80 * 1024) {
    // every bond <= 32: X and T stay in LDS, only the site tensors stream (qk_sweep_small_kernel); chains too long for
    // its LDS budget (several hundred sites) take the ring kernel below
    if (f32) qk_sweep_small_kernel<float><<<dim3(grid), dim3(512), lds_small, c->stream>>>(a);
    else qk_sweep_small_kernel<double><<<dim3(grid), dim3(512), lds_small, c->stream>>>(a);
    c->last.kernel = QK_KERNEL_SMALL;
  } else if (fused) {
    // X in LDS, T in registers, site tensors read straight into MFMA fragments from the interleaved image
    for (const qk_mps_set* m : {xs, ys}) {
      const int rc_il = ensure_interleaved(c, const_cast<qk_mps_set*>(m));
      if (rc_il != QK_OK) return rc_il;
    }
    a.xdata = xs->d_il, a.ydata = ys->d_il;
    if (plan->edge_k > 0) {  // the ends of the chain from the sets' edge blocks (chosen by the planner)
      for (const qk_mps_set* m : {xs, ys}) {
        const int rc_e = ensure_edges(c, const_cast<qk_mps_set*>(m), plan->edge_k);
        if (rc_e != QK_OK) return rc_e;
      }
      a.xedge = xs->d_edge, a.xedge_offs = xs->d_edge_offs, a.yedge = ys->d_edge, a.yedge_offs = ys->d_edge_offs, a.edge_k = plan->edge_k;
    }
    if (c->merge_sites && xs->n_sites - 2 * plan->edge_k >= 2) {  // the chain's sites contracted in twos: a workgroup picks per pair and step
      for (const qk_mps_set* m : {xs, ys}) {
        const int rc_m = ensure_merged(c, const_cast<qk_mps_set*>(m), plan->edge_k);
        if (rc_m != QK_OK) return rc_m;
      }
      a.xmg = xs->d_mg, a.xmg_offs = xs->d_mg_offs, a.ymg = ys->d_mg, a.ymg_offs = ys->d_mg_offs, a.merge_steps = xs->mg_steps;
    }
#ifdef QK_LAB  // TIMING EXPERIMENT (wrong results): every tensor read from the first MiB of its image -- what would perfect L2 hits buy?
    if (const char* v = std::getenv("QK_DEBUG_ALIAS")) {
      const long long win = std::atoll(v);  // window in doubles (e.g. 131072 = 1 MiB)
      if (win > 0) {
        auto fold = [&](const int64_t* src, const long long n) -> const int64_t* {
          int64_t* dst = nullptr;
          if (hipMalloc(&dst, (size_t)n * sizeof(int64_t)) != hipSuccess) return src;  // (leaked: experiment)
          qk_lab_fold_kernel<<<dim3(256), dim3(256), 0, c->stream>>>(src, dst, n, win);
          return dst;
        };
        a.xoffs = fold(a.xoffs, (long long)xs->n_states * xs->n_sites), a.yoffs = fold(a.yoffs, (long long)ys->n_states * ys->n_sites);
        if (a.merge_steps > 0)
          a.xmg_offs = fold(a.xmg_offs, (long long)xs->n_states * xs->mg_steps), a.ymg_offs = fold(a.ymg_offs, (long long)ys->n_states * ys->mg_steps);
      }
    }
#endif
    a.x_plane = (long long)xs->max_pad * ys->max_pad;  // complex elements per global X buffer (two per workgroup)
    HIP_TRY(hipEventRecord(c->ev0, c->stream));        // the conversion above is not part of the sweep
    a.nq = plan->nq, c->last.queues = plan->nq > 1 ? 8 : 1, c->tail_pending = true;
    // the dual form (pairs of tiles per wave: half the A and X fragments per matrix instruction) against single tiles, same box:
    // uniform bonds 48 / 64 / 96 / 128 / 256: +2 / +4 / +7 / +12 / +19 %; first run of the headline set's split sweep: 255 against 264 ms
    if (mixed) {
      SweepArgs a1 = a, a2 = a;
      a1.npairs = plan->n_first;
      a2.pairs = a.pairs + 2 * plan->n_first, a2.npairs = np - plan->n_first;
      a2.values = a.values + plan->n_first, a2.z = a.z ? a.z + 2 * plan->n_first : nullptr;
      a2.counter = c->counter + 8 * QK_QSTRIDE, a2.tail = tail + 2;
      if (plan->nq > 1) {
        a1.nq = a2.nq = 8;
        for (int s_ = 0; s_ <= 8; ++s_) a2.qstart[s_] = plan->qstart[8 + s_] - plan->n_first;
      }
      const unsigned g1 = (unsigned)std::min<long long>(a1.npairs, (long long)(fused_two ? 2 : 1) * c->num_cus);
      if (fused_two) QKF_KERNEL_TWO<<<dim3(g1), dim3(64 * QKF_TWO_NW), lds_fused, c->stream>>>(a1);
      else if (dual) QKF_KERNEL_DUAL<<<dim3(g1), dim3(64 * QKF_DUAL_NW), lds_fused, c->stream>>>(a1);
      else QKF_KERNEL_ONE<<<dim3(g1), dim3(64 * QKF_ONE_NW), lds_fused, c->stream>>>(a1);
      HIP_TRY(hipEventRecord(c->ev_mid, c->stream));
      qk_sweep_wave2_kernel<3, double><<<dim3((unsigned)std::min<long long>(a2.npairs, 8ll * c->num_cus)), dim3(64), 0, c->stream>>>(a2);
      c->last.second_pairs = plan->second.pairs, c->last.second_flops = plan->second.flops, c->last.second_padded_flops = plan->second.padded_flops;
      c->last.second_bytes = plan->second.bytes, c->last.second_kernel = QK_KERNEL_WAVE2;
      c->split_pending = true;
    } else if (fused_two) QKF_KERNEL_TWO<<<dim3(grid), dim3(64 * QKF_TWO_NW), lds_fused, c->stream>>>(a);
    else if (dual && !split) QKF_KERNEL_DUAL<<<dim3(grid), dim3(64 * QKF_DUAL_NW), lds_fused, c->stream>>>(a);
    else if (split) {
      // the plan lists the pairs whose sites fit the smaller LDS buffer behind the others: one 12-wave workgroup per CU for
      // the first run, two 8-wave workgroups per CU for the second, back to back on the stream
      SweepArgs a1 = a, a2 = a;
      a1.npairs = plan->n_first;
      a2.pairs = a.pairs + 2 * plan->n_first, a2.npairs = np - plan->n_first;
      a2.values = a.values + plan->n_first, a2.z = a.z ? a.z + 2 * plan->n_first : nullptr;
      a2.counter = c->counter + 8 * QK_QSTRIDE, a2.tail = tail + 2;
      if (plan->nq > 1) {  // 8 queues per run
        a1.nq = a2.nq = 8;
        for (int s_ = 0; s_ <= 8; ++s_) a2.qstart[s_] = plan->qstart[8 + s_] - plan->n_first;
      }
      if (dual) QKF_KERNEL_DUAL<<<dim3((unsigned)std::min<long long>(a1.npairs, c->num_cus)), dim3(64 * QKF_DUAL_NW), lds_fused, c->stream>>>(a1);
      else QKF_KERNEL_ONE<<<dim3((unsigned)std::min<long long>(a1.npairs, c->num_cus)), dim3(64 * QKF_ONE_NW), lds_fused, c->stream>>>(a1);
      HIP_TRY(hipEventRecord(c->ev_mid, c->stream));
      QKF_KERNEL_TWO<<<dim3((unsigned)std::min<long long>(a2.npairs, 2ll * c->num_cus)), dim3(64 * QKF_TWO_NW), (size_t)QKF_XCAP_TWO * 16 + lds_meta, c->stream>>>(a2);
      c->last.second_pairs = plan->second.pairs, c->last.second_flops = plan->second.flops, c->last.second_padded_flops = plan->second.padded_flops;
      c->last.second_bytes = plan->second.bytes, c->last.second_kernel = QK_KERNEL_FUSED2;
      c->split_pending = true;
    } else QKF_KERNEL_ONE<<<dim3(grid), dim3(64 * QKF_ONE_NW), lds_fused, c->stream>>>(a);
    c->last.kernel = fused_two ? QK_KERNEL_FUSED2 : dual ? QK_KERNEL_FUSED_DUAL : QK_KERNEL_FUSED1;
  } else if (f32) {  // complex64 sweep: the ring kernel on fp32 planes; QK_VARIANT does not apply
    qk_sweep_ring_kernel<float><<<dim3(grid), dim3(512), lds_ring, c->stream>>>(a);
    c->last.kernel = QK_KERNEL_RING;
  } else if (c->variant == 20) {  // the ring sweep: LDS-DMA staging ring (K-tile 8, three slots) + 3M complex product
    qk_sweep_ring_kernel<double><<<dim3(grid), dim3(512), lds_ring, c->stream>>>(a);
    c->last.kernel = QK_KERNEL_RING;
  } else {  // experimental / diagnostic kernels (qk_lab.hip, libqklab.so only)
#ifdef QK_LAB
    const int rc_lab = qk_lab_launch(c, c->variant, a, grid, xs->n_sites);
    if (rc_lab != QK_OK) return rc_lab;
    c->last.kernel = QK_KERNEL_LAB;
#else
    return fail(QK_EINVAL, "qk_gram_values: no kernel for this call");
#endif
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  c->ev_pending = true;
  c->last.grid = launched_grid;
  return QK_OK;
}

extern "C" int qk_gram_values_host(qk_ctx* c, const qk_mps_set* xs, const qk_mps_set* ys, const qk_plan* plan, double* values_host, double* z_host) {
  if (!c || !plan || !values_host) return fail(QK_EINVAL, "qk_gram_values_host: null argument");
  const int64_t np = qk_plan_num_pairs(plan);
  if (np == 0) return QK_OK;
  HIP_TRY(hipSetDevice(c->device));
  DevBuf vals, z;  // released on every exit path
  HIP_TRY(vals.alloc((size_t)np * sizeof(double)));
  if (z_host) HIP_TRY(z.alloc((size_t)np * 2 * sizeof(double)));
  const int rc = qk_gram_values(c, xs, ys, plan, vals.as<double>(), z.as<double>());
  if (rc != QK_OK) return rc;
  HIP_TRY(hipMemcpyAsync(values_host, vals.p, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (z_host) HIP_TRY(hipMemcpyAsync(z_host, z.p, (size_t)np * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QK_OK;
}

extern "C" int qk_scatter(qk_ctx* c, const int32_t* pairs_dev, const double* values_dev, int64_t n, double* k_dev, int64_t ld, int32_t mirror) {
  if (!c || !pairs_dev || !values_dev || !k_dev) return fail(QK_EINVAL, "qk_scatter: null argument");
  if (n <= 0) return QK_OK;
  QkRangeGuard range_("qk:scatter");
  HIP_TRY(hipSetDevice(c->device));
  const int bs = 256;
  hipLaunchKernelGGL(qk_scatter_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, c->stream, pairs_dev, values_dev, (long long)n, k_dev, (long long)ld, (int)mirror);
  HIP_TRY(hipGetLastError());
  return QK_OK;
}

extern "C" const char* qk_kernel_name(int32_t kernel, int32_t precision) {
  const bool f32 = precision == 32;
  switch (kernel) {
    case QK_KERNEL_WAVE: return "qk_sweep_wave_kernel<0>";
    case QK_KERNEL_SMALL: return f32 ? "qk_sweep_small_kernel<float>" : "qk_sweep_small_kernel<double>";
    case QK_KERNEL_FUSED1: return "qk_sweep_fused_kernel<12, 2, 8192, 3>";
    case QK_KERNEL_FUSED2: return "qk_sweep_fused_kernel<8, 1, 4608, 4>";
    case QK_KERNEL_RING: return f32 ? "qk_sweep_ring_kernel<float>" : "qk_sweep_ring_kernel<double>";
    case QK_KERNEL_WAVE2: return precision == 32 ? "qk_sweep_wave2_kernel<3, float>" : "qk_sweep_wave2_kernel<3, double>";
    case QK_KERNEL_WAVE2_PLAIN: return "qk_sweep_wave2_kernel<0, double>";
    case QK_KERNEL_FUSED_DUAL: return "qk_sweep_fused_dual_kernel<12, 8192, 3>";
    case QK_KERNEL_LAB: return "(lab kernel)";
    default: return "(none)";
  }
}
#ifndef QKF_EXPERIMENT
static_assert(QKF_ONE_NW == 12 && QKF_ONE_S == 2 && QKF_ONE_WPS == 3 && QKF_TWO_NW == 8 && QKF_TWO_S == 1 && QKF_TWO_WPS == 4 && QKF_XCAP_ONE == 8192 && QKF_XCAP_TWO == 4608,
              "qk_kernel_name spells the fused sweep's template arguments");
#endif

extern "C" int qk_get_stats(qk_ctx* c, qk_stats* out) {
  if (!c || !out) return fail(QK_EINVAL, "qk_get_stats: null argument");
  if (c->ev_pending) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last.kernel_ms = ms;
    if (c->tail_pending) {  // device clocks of the launch(es): share of the duration during which the chip was draining
      unsigned long long t[8];
      HIP_TRY(hipMemcpy(t, c->counter + QK_NQ_MAX * QK_QSTRIDE, sizeof t, hipMemcpyDeviceToHost));
      auto frac = [](const unsigned long long start, const unsigned long long first_exit, const unsigned long long last_exit) {
        return (last_exit > start && first_exit <= last_exit && first_exit >= start) ? (double)(last_exit - first_exit) / (double)(last_exit - start) : 0.0;
      };
      c->last.tail_frac = frac(t[0], t[1], t[4]);
      if (c->split_pending) c->last.second_tail_frac = frac(t[2], t[3], t[6]);
      c->tail_pending = false;
    }
    if (c->split_pending) {
      HIP_TRY(hipEventElapsedTime(&ms, c->ev_mid, c->ev1));
      c->last.second_ms = ms;
      c->split_pending = false;
    }
    c->ev_pending = false;
  }
  *out = c->last;
  return QK_OK;
}

// dims table of a set as the planner wants it
static int plan_for_sets(const qk_mps_set* xs, const qk_mps_set* ys, qk_plan** plan) {
  const bool sym = (ys == nullptr || ys == xs);
  return qk_plan_create(xs->n_sites, xs->n_states, xs->dims_true.data(), sym ? xs->n_states : ys->n_states,
                        sym ? nullptr : ys->dims_true.data(), sym ? (QK_PLAN_SYMMETRIC | QK_PLAN_ORIENT) : 0u, 1, 0, 0, plan);
}

struct PlanGuard {  // a plan owned by one call
  qk_plan* p = nullptr;
  ~PlanGuard() { qk_plan_destroy(p); }
};

extern "C" int qk_gram_host(qk_ctx* c, const qk_mps_set* xs, const qk_mps_set* ys, double* out, int64_t ld) {
  if (!c || !xs || !out) return fail(QK_EINVAL, "qk_gram_host: null argument");
  const bool sym = (ys == nullptr || ys == xs);
  const int nx = xs->n_states, ny = sym ? nx : ys->n_states;
  if (ld < nx) return fail(QK_EINVAL, "qk_gram_host: ld %lld < %d columns", (long long)ld, nx);
  PlanGuard plan;
  int rc = plan_for_sets(xs, ys, &plan.p);
  if (rc != QK_OK) return rc;
  const int64_t np = qk_plan_num_pairs(plan.p);
  DevBuf vals, k;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(vals.alloc((size_t)np * sizeof(double)));
  HIP_TRY(k.alloc((size_t)ny * nx * sizeof(double)));
  HIP_TRY(hipMemsetAsync(k.p, 0, (size_t)ny * nx * sizeof(double), c->stream));
  rc = qk_gram_values(c, xs, ys, plan.p, vals.as<double>(), nullptr);
  if (rc == QK_OK) rc = qk_scatter(c, plan.p->d_pairs, vals.as<double>(), np, k.as<double>(), nx, sym ? 1 : 0);
  if (rc != QK_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld * sizeof(double), k.p, (size_t)nx * sizeof(double), (size_t)nx * sizeof(double), (size_t)ny, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QK_OK;
}

extern "C" int qk_overlaps_host(qk_ctx* c, const qk_mps_set* xs, const qk_mps_set* ys, double* out) {
  if (!c || !xs || !out) return fail(QK_EINVAL, "qk_overlaps_host: null argument");
  if (!ys) ys = xs;
  const int nx = xs->n_states, ny = ys->n_states;
  PlanGuard plan;  // all ny*nx pairs (no symmetry: z[i][j] = conj z[j][i] is left to the caller)
  int rc = qk_plan_create(xs->n_sites, nx, xs->dims_true.data(), ny, ys->dims_true.data(), 0u, 1, 0, 16, &plan.p);
  if (rc != QK_OK) return rc;
  const int64_t np = qk_plan_num_pairs(plan.p);
  DevBuf vals, zd;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(vals.alloc((size_t)np * sizeof(double)));
  HIP_TRY(zd.alloc((size_t)np * 2 * sizeof(double)));
  rc = qk_gram_values(c, xs, ys, plan.p, vals.as<double>(), zd.as<double>());
  if (rc != QK_OK) return rc;
  std::vector<double> z((size_t)np * 2);
  HIP_TRY(hipMemcpyAsync(z.data(), zd.p, z.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const int32_t* pr = qk_plan_pairs(plan.p);
  for (int64_t t = 0; t < np; ++t) {
    const int64_t o = ((int64_t)pr[2 * t + 1] * nx + pr[2 * t]) * 2;
    out[o] = z[2 * t], out[o + 1] = z[2 * t + 1];
  }
  return QK_OK;
}

extern "C" int qk_selftest_mfma(qk_ctx* c) {
  if (!c) return fail(QK_EINVAL, "qk_selftest_mfma: null context");
  HIP_TRY(hipSetDevice(c->device));
  double hp[256], hq[256], hc[256], ref[256];
  for (int k = 0; k < 16; ++k)
    for (int m = 0; m < 16; ++m) {
      hp[k * 16 + m] = 1.0 + 0.25 * k - 0.5 * m + 0.125 * ((k * 7 + m * 3) % 5);  // asymmetric on purpose
      hq[k * 16 + m] = -2.0 + 0.5 * k + 0.75 * m - 0.25 * ((k * 5 + m * 11) % 7);
    }
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double s = 0;
      for (int k = 0; k < 16; ++k) s += hp[k * 16 + m] * hq[k * 16 + n];
      ref[m * 16 + n] = s;
    }
  DevBuf bp, bq, bc;
  HIP_TRY(bp.alloc(sizeof hp));
  HIP_TRY(bq.alloc(sizeof hq));
  HIP_TRY(bc.alloc(sizeof hc));
  double *dp = bp.as<double>(), *dq = bq.as<double>(), *dc = bc.as<double>();
  HIP_TRY(hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dq, hq, sizeof hq, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qk_selftest_kernel, dim3(1), dim3(64), 0, c->stream, dp, dq, dc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int e = 0; e < 256; ++e) worst = std::max(worst, std::fabs(hc[e] - ref[e]));
  if (worst > 1e-9) return fail(QK_EDEVICE, "qk_selftest_mfma: f64 MFMA fragment map mismatch (max abs error %.3g)", worst);
  // the same product through v_mfma_f32_16x16x4_f32 (operands are exact in fp32; sums of 16 such products too)
  float fp[256], fq[256], fc[256];
  for (int e = 0; e < 256; ++e) fp[e] = (float)hp[e], fq[e] = (float)hq[e];
  DevBuf be, bf, bg;
  HIP_TRY(be.alloc(sizeof fp));
  HIP_TRY(bf.alloc(sizeof fq));
  HIP_TRY(bg.alloc(sizeof fc));
  float *ep = be.as<float>(), *eq = bf.as<float>(), *ec = bg.as<float>();
  HIP_TRY(hipMemcpy(ep, fp, sizeof fp, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(eq, fq, sizeof fq, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qk_selftest_f32_kernel, dim3(1), dim3(64), 0, c->stream, ep, eq, ec);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(fc, ec, sizeof fc, hipMemcpyDeviceToHost));
  worst = 0;
  for (int e = 0; e < 256; ++e) worst = std::max(worst, std::fabs((double)fc[e] - ref[e]));
  if (worst > 1e-3) return fail(QK_EDEVICE, "qk_selftest_mfma: f32 MFMA fragment map mismatch (max abs error %.3g)", worst);
  return QK_OK;
}

