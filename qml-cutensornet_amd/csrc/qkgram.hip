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
// Written for gfx950 only: 64-lane wavefronts, 160 KiB LDS per CU, no portability layer.
#include "../../include/qkgram.h"

#include <hip/hip_runtime.h>
#include <type_traits>

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

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(QK_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

extern "C" const char* qk_last_error(void) { return g_err.c_str(); }

// ----------------------------------------------------------------------------------------
// shapes
// ----------------------------------------------------------------------------------------
static constexpr int TILE = 16;  // M/N granule of v_mfma_f64_16x16x4_f64
static inline int pad16(int x) { return (x + TILE - 1) / TILE * TILE; }

struct qk_ctx {
  int device = 0;
  int num_cus = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_pending = false;
  double* scratch = nullptr;
  size_t scratch_bytes = 0;
  unsigned long long* counter = nullptr;
  unsigned long long* prof = nullptr;  // 8 cycle sums of the diagnostic variant
  int variant = 20;    // sweep kernel variant (QK_VARIANT): 20 = shipped (ring sweep: LDS-DMA ring + 3M product); 17 = lean register-staged sweep;
                       // 0, 2, 12, 13, 14, 16, 21, 23 = other kernels kept for A/B; 9, 19 = instrumented
  int wgs_per_cu = 2;  // resident workgroups per CU (QK_WGS_PER_CU)
  qk_stats last{};
};

struct qk_mps_set {
  qk_ctx* ctx = nullptr;
  int n_states = 0, n_sites = 0, max_pad = 0;
  int precision = 64;         // bits of a real: 64 (complex128 planes) or 32 (complex64 planes, same element offsets)
  double* d_data = nullptr;   // the planes; floats when precision == 32
  int32_t* d_dims = nullptr;  // padded bonds [n_states][n_sites+1]
  int32_t* d_true = nullptr;  // true bonds   [n_states][n_sites+1]
  int64_t* d_offs = nullptr;  // re-plane offsets (doubles) [n_states][n_sites]
  std::vector<int32_t> dims_true;
  int64_t bytes = 0;
};

struct qk_plan {
  int n_sites = 0, nx = 0, ny = 0;
  bool symmetric = false;
  int world = 1, rank = 0;
  int64_t total_pairs = 0, max_per_rank = 0;
  std::vector<int32_t> pairs;   // this rank, (i, j) interleaved
  std::vector<int32_t> groups;  // (first pair, count): runs of <= group pairs that share the x state
  int group = 1;
  qk_stats stats{};
  // lazily uploaded copy
  qk_ctx* up_ctx = nullptr;
  int32_t* d_pairs = nullptr;
  int32_t* d_groups = nullptr;
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
static void pair_work(int n, const int32_t* a, const int32_t* b, double* flops, double* padded, double* bytes) {
  double f = 0, fp = 0, by = 0;
  for (int k = 0; k < n; ++k) {
    const double a0 = a[k], a1 = a[k + 1], b0 = b[k], b1 = b[k + 1];
    const double f1 = a0 * b0 * 2 * b1 + 2 * a0 * a1 * b1;
    const double f2 = a0 * b0 * 2 * a1 + 2 * b0 * a1 * b1;
    f += 8 * std::min(f1, f2);
    const double A0 = pad16(a[k]), A1 = pad16(a[k + 1]), B0 = pad16(b[k]), B1 = pad16(b[k + 1]);
    fp += 8 * (A0 * B0 * 2 * B1 + 2 * A0 * A1 * B1);
    by += 16.0 * 2 * (a0 * a1 + b0 * b1);
  }
  *flops = f;
  *padded = fp;
  *bytes = by + 8;
}

extern "C" int qk_plan_create(int32_t n_sites, int32_t nx, const int32_t* x_dims, int32_t ny, const int32_t* y_dims,
                              uint32_t flags, int32_t world_size, int32_t rank, int32_t block, qk_plan** out) {
  if (!out || !x_dims || n_sites <= 0 || nx <= 0) return fail(QK_EINVAL, "qk_plan_create: bad argument");
  const bool sym = (flags & QK_PLAN_SYMMETRIC) != 0;
  if (sym) {
    y_dims = x_dims;
    ny = nx;
  } else if (!y_dims || ny <= 0)
    return fail(QK_EINVAL, "qk_plan_create: y_dims required unless symmetric");
  if (world_size <= 0 || rank < 0 || rank >= world_size) return fail(QK_EINVAL, "qk_plan_create: bad rank %d/%d", rank, world_size);
  if (block <= 0) block = std::max(nx, ny);  // one tile: the whole pair list in cost order (locality blocks measured no gain)
  qk_plan* p = new (std::nothrow) qk_plan;
  if (!p) return fail(QK_ENOMEM, "qk_plan_create: out of memory");
  p->n_sites = n_sites, p->nx = nx, p->ny = ny, p->symmetric = sym, p->world = world_size, p->rank = rank;

  struct Item {
    int32_t i, j;
    float cost;
  };
  std::vector<Item> tile;
  const int stride = n_sites + 1;
  int64_t t = 0;  // running index in the global order
  std::vector<int64_t> per_rank(world_size, 0);
  double flops = 0, padded = 0, bytes = 0;
  const int nbx = (nx + block - 1) / block, nby = (ny + block - 1) / block;
  for (int bj = 0; bj < nby; ++bj)
    for (int bi = 0; bi < nbx; ++bi) {
      if (sym && bi > bj) continue;
      tile.clear();
      for (int j = bj * block; j < std::min(ny, (bj + 1) * block); ++j)
        for (int i = bi * block; i < std::min(nx, (bi + 1) * block); ++i) {
          if (sym && i > j) continue;
          double f, fp, by;
          pair_work(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride, &f, &fp, &by);
          tile.push_back({i, j, (float)fp});
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
          double f, fp, by;
          pair_work(n_sites, x_dims + (int64_t)it.i * stride, y_dims + (int64_t)it.j * stride, &f, &fp, &by);
          flops += f, padded += fp, bytes += by;
        }
        ++t;
      }
    }
  // Regroup this rank's share: pairs that share the x state are made contiguous and cut into
  // groups of at most QK_GROUP (default 4) pairs -- one workgroup sweeps a group in lockstep so
  // that A_i is read once per group and the per-phase latencies are shared.  Groups are then
  // ordered by decreasing cost (longest first for the device-side queue).
  {
    int G = 4;
    if (const char* e = std::getenv("QK_GROUP")) G = std::max(1, std::min(4, std::atoi(e)));
    p->group = G;
    const int64_t np = (int64_t)p->pairs.size() / 2;
    std::vector<Item> mine((size_t)np);
    for (int64_t q = 0; q < np; ++q) {
      double f, fp, by;
      const int i = p->pairs[2 * q], j = p->pairs[2 * q + 1];
      pair_work(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride, &f, &fp, &by);
      mine[(size_t)q] = {i, j, (float)fp};
    }
    std::stable_sort(mine.begin(), mine.end(), [](const Item& u, const Item& v) { return u.i != v.i ? u.i < v.i : u.cost > v.cost; });
    struct Grp {
      int64_t start;
      int count;
      double cost;
    };
    std::vector<Grp> grp;
    for (int64_t q = 0; q < np;) {
      int c = 1;
      double cost = mine[(size_t)q].cost;
      while (c < G && q + c < np && mine[(size_t)(q + c)].i == mine[(size_t)q].i) cost += mine[(size_t)(q + c)].cost, ++c;
      grp.push_back({q, c, cost});
      q += c;
    }
    std::stable_sort(grp.begin(), grp.end(), [](const Grp& u, const Grp& v) { return u.cost > v.cost; });
    p->pairs.clear();
    for (const Grp& gq : grp) {
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
extern "C" int qk_plan_stats(const qk_plan* p, qk_stats* out) {
  if (!p || !out) return fail(QK_EINVAL, "qk_plan_stats: null argument");
  *out = p->stats;
  return QK_OK;
}

// ----------------------------------------------------------------------------------------
// device code
// ----------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
// Staging geometry of the complex GEMM: a workgroup (4 waves) produces one 64x64 complex
// output block per pass; operands are staged k-major through LDS in K-tiles of 16 rows,
// 4 planes (A re/im, B re/im) of [16][64] doubles, double-buffered = 64 KiB.
static constexpr int WG_THREADS = 256;
static constexpr int PASS = 64;
static constexpr int KT = 16;
static constexpr int PLANE = KT * PASS;         // doubles per staged plane
static constexpr int STAGE = 4 * PLANE;         // doubles per buffer
static constexpr int LDS_DOUBLES = 2 * STAGE;   // double-buffered
static constexpr size_t LDS_BYTES = LDS_DOUBLES * sizeof(double) + 16;  // + pair slot

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
  unsigned long long* counter;
  unsigned long long* prof;  // diagnostic build only: cycle sums per section (see QK_VARIANT=9)
  int debug_flags;           // timing experiments only (QK_DEBUG_FLAGS): bit 0 = skip epilogue stores, bit 1 = skip steady-state fetch/stash, bit 2 = skip MFMAs, bit 3 = skip steady-state barriers (all give WRONG results)
  int prio_mode;             // 0: none; 1: second half of the grid at s_setprio 1; 2: odd blocks at s_setprio 1
};

// C[M x N] = sum_k Aop[k][m] * Bop[k][n]   (complex, split planes; CONJB conjugates Bop)
// Aop, Bop are "k-major": row k holds the M (resp. N) entries contiguously.  M, N, K are
// multiples of 16.  All 256 threads of the workgroup call this together.
//
// MFMA fragment maps (v_mfma_f64_16x16x4_f64; lane = 16*q + j):
//   A operand: lane holds Aop_tile[i = j][k = q]  -> staged element [4*ks + q][16*tm + j]
//   B operand: lane holds Bop_tile[k = q][n = j]  -> staged element [4*ks + q][16*tn + j]
//   C/D:       register r of the lane is C_tile[row = q + 4 r][col = j]
template <bool CONJB>
__device__ __forceinline__ void zgemm_kmajor(double* __restrict__ Cre, double* __restrict__ Cim, const int ldc,
                                             const double* __restrict__ Are, const double* __restrict__ Aim, const int lda,
                                             const double* __restrict__ Bre, const double* __restrict__ Bim, const int ldb,
                                             const int M, const int N, const int K, double* __restrict__ lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform
  const int j = lane & 15, q = lane >> 4;
  // staging role of this thread: two rows (srow, srow + 8), one 16-byte column unit
  const int srow = tid >> 5;      // 0..7
  const int scol = (tid & 31) * 2;  // 0..62
  const int nk = K / KT;

  for (int n0 = 0; n0 < N; n0 += PASS)
    for (int m0 = 0; m0 < M; m0 += PASS) {
      const int mt = min(PASS / TILE, (M - m0) / TILE);
      const int nt = min(PASS / TILE, (N - n0) / TILE);
      const int vt = mt * nt;
      const bool ldA = scol < mt * TILE, ldB = scol < nt * TILE;

      v4d cre[4], cim[4];
      int tm[4], tn[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        cre[s] = (v4d){0, 0, 0, 0};
        cim[s] = (v4d){0, 0, 0, 0};
        const int t = wave + 4 * s;
        tm[s] = (t < vt) ? (t % mt) : -1;
        tn[s] = (t < vt) ? (t / mt) : 0;
      }

      // staging registers: [re/im plane][row half] of the A and B operands
      double2 a0, a1, a2, a3, b0, b1, b2, b3;
      a0 = a1 = a2 = a3 = b0 = b1 = b2 = b3 = make_double2(0.0, 0.0);
#define QK_FETCH(kt_)                                                        \
  do {                                                                       \
    const long long k0_ = (long long)(kt_)*KT + srow;                        \
    if (ldA) {                                                               \
      const long long o0_ = k0_ * lda + m0 + scol, o1_ = o0_ + 8ll * lda;    \
      a0 = *reinterpret_cast<const double2*>(Are + o0_);                     \
      a1 = *reinterpret_cast<const double2*>(Are + o1_);                     \
      a2 = *reinterpret_cast<const double2*>(Aim + o0_);                     \
      a3 = *reinterpret_cast<const double2*>(Aim + o1_);                     \
    }                                                                        \
    if (ldB) {                                                               \
      const long long o0_ = k0_ * ldb + n0 + scol, o1_ = o0_ + 8ll * ldb;    \
      b0 = *reinterpret_cast<const double2*>(Bre + o0_);                     \
      b1 = *reinterpret_cast<const double2*>(Bre + o1_);                     \
      b2 = *reinterpret_cast<const double2*>(Bim + o0_);                     \
      b3 = *reinterpret_cast<const double2*>(Bim + o1_);                     \
    }                                                                        \
  } while (0)
#define QK_STASH(buf_)                                                       \
  do {                                                                       \
    double* base_ = lds + (buf_)*STAGE;                                      \
    const int o0_ = srow * PASS + scol, o1_ = o0_ + 8 * PASS;                \
    if (ldA) {                                                               \
      *reinterpret_cast<double2*>(base_ + 0 * PLANE + o0_) = a0;             \
      *reinterpret_cast<double2*>(base_ + 0 * PLANE + o1_) = a1;             \
      *reinterpret_cast<double2*>(base_ + 1 * PLANE + o0_) = a2;             \
      *reinterpret_cast<double2*>(base_ + 1 * PLANE + o1_) = a3;             \
    }                                                                        \
    if (ldB) {                                                               \
      *reinterpret_cast<double2*>(base_ + 2 * PLANE + o0_) = b0;             \
      *reinterpret_cast<double2*>(base_ + 2 * PLANE + o1_) = b1;             \
      *reinterpret_cast<double2*>(base_ + 3 * PLANE + o0_) = b2;             \
      *reinterpret_cast<double2*>(base_ + 3 * PLANE + o1_) = b3;             \
    }                                                                        \
  } while (0)

      QK_FETCH(0);
      QK_STASH(0);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) QK_FETCH(kt + 1);
        const double* base = lds + (kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (tm[s] >= 0) {
            const double* pa = base + q * PASS + tm[s] * TILE + j;
            const double* pb = base + 2 * PLANE + q * PASS + tn[s] * TILE + j;
#pragma unroll
            for (int ks = 0; ks < KT / 4; ++ks) {
              const double ar = pa[ks * 4 * PASS];
              const double ai = pa[PLANE + ks * 4 * PASS];
              const double br = pb[ks * 4 * PASS];
              double bi = pb[PLANE + ks * 4 * PASS];
              if (CONJB) bi = -bi;
              cre[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cre[s], 0, 0, 0);
              cim[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, cim[s], 0, 0, 0);
              cre[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cre[s], 0, 0, 0);
              cim[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, cim[s], 0, 0, 0);
            }
          }
        }
        if (kt + 1 < nk) QK_STASH((kt + 1) & 1);
        __syncthreads();
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (tm[s] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long long o = (long long)(m0 + tm[s] * TILE + q + 4 * r) * ldc + n0 + tn[s] * TILE + j;
            Cre[o] = cre[s][r];
            Cim[o] = cim[s][r];
          }
        }
      }
    }
#undef QK_FETCH
#undef QK_STASH
  // make this phase's output visible to the whole workgroup before the next phase reads it
  __syncthreads();
}

// One persistent workgroup = one (x_i, y_j) overlap at a time, pulled from a global queue.
//   X  [b x a]       environment, stored k-major for phase 1: X[l][L]        (scratch, L2-resident)
//   T  [a x 2b']     T[L][(p,r)] = sum_l X[l][L] B[l][(p,r)]                 (phase 1)
//   X' [b' x a']     X'[r][R]   = sum_{(L,p)} T[(L,p)][r] conj(A[(L,p)][R])  (phase 2; T re-read as a [2a x b'] k-major matrix)
__global__ __launch_bounds__(WG_THREADS, 2) void qk_sweep_kernel(const SweepArgs g) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  long long* slot = reinterpret_cast<long long*>(lds + LDS_DOUBLES);

  double* Xre = g.scratch + (long long)blockIdx.x * 2 * (g.x_plane + g.t_plane);
  double* Xim = Xre + g.x_plane;
  double* Tre = Xim + g.x_plane;
  double* Tim = Tre + g.t_plane;
  const int tid = threadIdx.x;

  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p >= g.npairs) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    const int32_t* xd = g.xdims + (long long)xi * (g.n_sites + 1);
    const int32_t* yd = g.ydims + (long long)yj * (g.n_sites + 1);
    const int64_t* xo = g.xoffs + (long long)xi * g.n_sites;
    const int64_t* yo = g.yoffs + (long long)yj * g.n_sites;

    // X_0 = 1 (1x1) in a zero 16x16 block
    {
      const int a = xd[0], b = yd[0];
      for (int e = tid; e < a * b; e += WG_THREADS) {
        Xre[e] = (e == 0) ? 1.0 : 0.0;
        Xim[e] = 0.0;
      }
      __syncthreads();
    }
    for (int k = 0; k < g.n_sites; ++k) {
      const int a = xd[k], a2 = xd[k + 1], b = yd[k], b2 = yd[k + 1];
      const double* Are = g.xdata + xo[k];
      const double* Aim = Are + (long long)a * 2 * a2;
      const double* Bre = g.ydata + yo[k];
      const double* Bim = Bre + (long long)b * 2 * b2;
      // phase 1: T[a x 2 b2] = X^T B      (A-operand X: K = b rows of a; B-operand B: K = b rows of 2 b2)
      zgemm_kmajor<false>(Tre, Tim, 2 * b2, Xre, Xim, a, Bre, Bim, 2 * b2, a, 2 * b2, b, lds);
      // phase 2: X'[b2 x a2] = T^T conj(A) (A-operand T as [2a][b2]; B-operand A as [2a][a2])
      zgemm_kmajor<true>(Xre, Xim, a2, Tre, Tim, b2, Are, Aim, a2, b2, a2, 2 * a, lds);
    }
    if (tid == 0) {
      const double re = Xre[0], im = Xim[0];
      g.values[p] = re * re + im * im;
      if (g.z) {
        g.z[2 * p] = re;
        g.z[2 * p + 1] = im;
      }
    }
    __syncthreads();
  }
}


// ----------------------------------------------------------------------------------------
// v2: flat software pipeline.  The (pass, K-tile) iteration space of one GEMM is a single
// sequence of steps; the operands of step s+1 are fetched from global memory while step s is
// multiplied, ACROSS pass boundaries, so only the first step of a phase exposes memory latency.
// Output block per pass: 64 x PN complex (PN = 64 or 128), K-tiles of KTL rows, double-buffered.
// K is walked in units of 4 (the MFMA k extent) up to the TRUE contraction length: rows beyond
// it are zero padding and are skipped.
// ----------------------------------------------------------------------------------------
// In-kernel cycle stamp for the DIAGNOSTIC variant only (never in the timed kernels): s_memtime
// with its own lgkmcnt(0), fenced against instruction motion.
__device__ __forceinline__ long long qk_stamp() {
  long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define QK_T(slot_, ...)                   \
  do {                                     \
    if (PROF) {                            \
      const long long t0_ = qk_stamp();    \
      __VA_ARGS__;                         \
      pc[slot_] += qk_stamp() - t0_;       \
    } else {                               \
      __VA_ARGS__;                         \
    }                                      \
  } while (0)

template <int PN, int KTL, int NW = 4, int PM_ = 64>
struct GemmCfg {
  static constexpr int PM = PM_;
  static constexpr int WGT = 64 * NW;  // threads per workgroup
  static constexpr int A_PLANE = KTL * PM;
  static constexpr int B_PLANE = KTL * PN;
  static constexpr int STAGE_D = 2 * A_PLANE + 2 * B_PLANE;  // doubles per buffer
  static constexpr int LDS_D = 2 * STAGE_D;
  static constexpr size_t LDS_B = (size_t)LDS_D * sizeof(double) + 16;
  static constexpr int UA = (A_PLANE / 2) / WGT;  // 16-byte units per thread per A plane
  static constexpr int UB = (B_PLANE / 2) / WGT;
  static constexpr int MAXT = (PM / TILE) * (PN / TILE) / NW;  // output tiles per wave
  static_assert(UA >= 1 && UB >= 1, "staging tile too small for the workgroup");
};


// Multiply one staged K-tile into this wave's accumulator tiles (the first `cnt` are valid).
// Full K-tiles (the common case) run a software pipeline over "groups" of 2 k-steps: the LDS
// fragment reads of group g+1 are issued before the 8 MFMAs of group g, so that the LDS latency
// hides under 512 cycles of matrix work instead of stalling the wave at every group.
template <bool CONJB, int PM, int PN, int A_PLANE, int B_PLANE, int KSTEPS, int MAXT, bool FULLK, bool PIPE = true>
__device__ __forceinline__ void mma_ktile(v4d (&cre)[MAXT], v4d (&cim)[MAXT], const int (&tm)[MAXT], const int (&tn)[MAXT],
                                          const double* __restrict__ base, const int q, const int j, const int cnt, const int ksteps) {
  if constexpr (PIPE && FULLK && (KSTEPS % 2 == 0)) {
    constexpr int GPT = KSTEPS / 2;       // groups per tile
    constexpr int NG = MAXT * GPT;        // groups per K-tile
    double far[2][2], fai[2][2], fbr[2][2], fbi[2][2];  // [buffer][k-step in group]
    auto load = [&](int g, int buf) __attribute__((always_inline)) {
      const int e = g / GPT, k0 = (g % GPT) * 2;
      const double* pa = base + (q + 4 * k0) * PM + tm[e] * TILE + j;
      const double* pb = base + 2 * A_PLANE + (q + 4 * k0) * PN + tn[e] * TILE + j;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        far[buf][h] = pa[h * 4 * PM];
        fai[buf][h] = pa[A_PLANE + h * 4 * PM];
        fbr[buf][h] = pb[h * 4 * PN];
        fbi[buf][h] = CONJB ? -pb[B_PLANE + h * 4 * PN] : pb[B_PLANE + h * 4 * PN];
      }
    };
    if (cnt > 0) load(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int e = g / GPT;
      if (e < cnt) {
        if (g + 1 < NG && (g + 1) / GPT < cnt) load(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double ar = far[g & 1][h], ai = fai[g & 1][h], br = fbr[g & 1][h], bi = fbi[g & 1][h];
          cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cre[e], 0, 0, 0);
          cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, cim[e], 0, 0, 0);
          cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cre[e], 0, 0, 0);
          cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, cim[e], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < MAXT; ++e) {
      if (e < cnt) {
        const double* pa = base + q * PM + tm[e] * TILE + j;
        const double* pb = base + 2 * A_PLANE + q * PN + tn[e] * TILE + j;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          if (FULLK || ks < ksteps) {
            const double ar = pa[ks * 4 * PM];
            const double ai = pa[A_PLANE + ks * 4 * PM];
            const double br = pb[ks * 4 * PN];
            double bi = pb[B_PLANE + ks * 4 * PN];
            if (CONJB) bi = -bi;
            cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cre[e], 0, 0, 0);
            cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, cim[e], 0, 0, 0);
            cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cre[e], 0, 0, 0);
            cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, cim[e], 0, 0, 0);
          }
        }
      }
    }
  }
}

template <bool CONJB, int PN, int KTL, bool PROF, int NW = 4, int PMT = 64>
__device__ __forceinline__ void zgemm_flat(double* __restrict__ Cre, double* __restrict__ Cim, const int ldc,
                                           const double* __restrict__ Are, const double* __restrict__ Aim, const int lda,
                                           const double* __restrict__ Bre, const double* __restrict__ Bim, const int ldb,
                                           const int M, const int N, const int Ktrue, double* __restrict__ lds, long long (&pc)[8]) {
  using G = GemmCfg<PN, KTL, NW, PMT>;
  constexpr int PM = G::PM;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;

  const int npm = (M + PM - 1) / PM;
  const int npn = (N + PN - 1) / PN;
  const int nk = (Ktrue + KTL - 1) / KTL;
  const int k4 = (Ktrue + 3) >> 2;  // MFMA k-steps in total
  const int total = npm * npn * nk;

  double2 ra[2 * G::UA], rb[2 * G::UB];
#pragma unroll
  for (int i = 0; i < 2 * G::UA; ++i) ra[i] = make_double2(0.0, 0.0);
#pragma unroll
  for (int i = 0; i < 2 * G::UB; ++i) rb[i] = make_double2(0.0, 0.0);

  // position of the step being FETCHED
  int f_kt = 0, f_pm = 0, f_pn = 0;
  auto fetch = [&]() __attribute__((always_inline)) {
    const int m0 = f_pm * PM, n0 = f_pn * PN;
    const int mcols = min(PM, M - m0), ncols = min(PN, N - n0);
    const long long krow = (long long)f_kt * KTL;
#pragma unroll
    for (int i = 0; i < G::UA; ++i) {
      const int u = tid + G::WGT * i;
      const int row = u / (PM / 2), col = (u % (PM / 2)) * 2;
      if (col < mcols) {
        const long long o = (krow + row) * lda + m0 + col;
        ra[2 * i] = *reinterpret_cast<const double2*>(Are + o);
        ra[2 * i + 1] = *reinterpret_cast<const double2*>(Aim + o);
      }
    }
#pragma unroll
    for (int i = 0; i < G::UB; ++i) {
      const int u = tid + G::WGT * i;
      const int row = u / (PN / 2), col = (u % (PN / 2)) * 2;
      if (col < ncols) {
        const long long o = (krow + row) * ldb + n0 + col;
        rb[2 * i] = *reinterpret_cast<const double2*>(Bre + o);
        rb[2 * i + 1] = *reinterpret_cast<const double2*>(Bim + o);
      }
    }
    if (++f_kt == nk) {
      f_kt = 0;
      if (++f_pm == npm) f_pm = 0, ++f_pn;
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
    double* base = lds + buf * G::STAGE_D;
#pragma unroll
    for (int i = 0; i < G::UA; ++i) {
      const int u = tid + G::WGT * i;
      const int o = (u / (PM / 2)) * PM + (u % (PM / 2)) * 2;
      *reinterpret_cast<double2*>(base + o) = ra[2 * i];
      *reinterpret_cast<double2*>(base + G::A_PLANE + o) = ra[2 * i + 1];
    }
#pragma unroll
    for (int i = 0; i < G::UB; ++i) {
      const int u = tid + G::WGT * i;
      const int o = (u / (PN / 2)) * PN + (u % (PN / 2)) * 2;
      *reinterpret_cast<double2*>(base + 2 * G::A_PLANE + o) = rb[2 * i];
      *reinterpret_cast<double2*>(base + 2 * G::A_PLANE + G::B_PLANE + o) = rb[2 * i + 1];
    }
  };

  v4d cre[G::MAXT], cim[G::MAXT];
  int tm[G::MAXT], tn[G::MAXT];
  int cnt = 0;                       // valid output tiles of this wave in the current pass
  int c_kt = 0, c_pm = 0, c_pn = 0;  // position of the step being COMPUTED

  QK_T(5, { fetch(); stash(0); __syncthreads(); });
  for (int s = 0; s < total; ++s) {
    QK_T(0, { if (s + 1 < total) fetch(); });
    const int m0 = c_pm * PM, n0 = c_pn * PN;
    if (c_kt == 0) {
      const int mt = min(PM / TILE, (M - m0) / TILE);
      const int nt = min(PN / TILE, (N - n0) / TILE);
      const int vt = mt * nt;
      cnt = (vt > wave) ? (vt - wave + NW - 1) / NW : 0;  // tiles t = wave + NW e < vt
#pragma unroll
      for (int e = 0; e < G::MAXT; ++e) {
        cre[e] = (v4d){0, 0, 0, 0};
        cim[e] = (v4d){0, 0, 0, 0};
        const int t = min(wave + NW * e, vt - 1);  // clamp: entries e >= cnt are never used
        tm[e] = t % mt;
        tn[e] = t / mt;
      }
    }
    const double* base = lds + (s & 1) * G::STAGE_D;
    const int ksteps = min(KTL / 4, k4 - c_kt * (KTL / 4));
    QK_T(1, {
      if (ksteps == KTL / 4)
        mma_ktile<CONJB, PM, PN, G::A_PLANE, G::B_PLANE, KTL / 4, G::MAXT, true>(cre, cim, tm, tn, base, q, j, cnt, ksteps);
      else
        mma_ktile<CONJB, PM, PN, G::A_PLANE, G::B_PLANE, KTL / 4, G::MAXT, false>(cre, cim, tm, tn, base, q, j, cnt, ksteps);
    });
    const long long te_ = PROF ? qk_stamp() : 0;
    if (c_kt == nk - 1) {
#pragma unroll
      for (int e = 0; e < G::MAXT; ++e) {
        if (e < cnt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long long o = (long long)(m0 + tm[e] * TILE + q + 4 * r) * ldc + n0 + tn[e] * TILE + j;
            Cre[o] = cre[e][r];
            Cim[o] = cim[e][r];
          }
        }
      }
    }
    if (PROF) pc[2] += qk_stamp() - te_;
    if (++c_kt == nk) {
      c_kt = 0;
      if (++c_pm == npm) c_pm = 0, ++c_pn;
    }
    QK_T(3, { if (s + 1 < total) stash((s + 1) & 1); });
    QK_T(4, { __syncthreads(); });
  }
  // make this phase's output visible to the whole workgroup before the next phase reads it
  QK_T(6, { __syncthreads(); });
}

// ----------------------------------------------------------------------------------------
// v3: the flat pipeline with a TWO-step-deep register prefetch.  Tile t is fetched from global
// memory at the start of step t-2 and written to LDS at the end of step t-1, so every load has
// two full MFMA blocks (~3-7 us) to land.  Two staging register sets alternate by tile parity;
// the steady-state loop is unrolled by two with unconditional fetches so that the compiler's
// vmcnt bookkeeping stays exact (a conditional fetch would force vmcnt(0) at the stash).
// ----------------------------------------------------------------------------------------
// Workgroup barrier that publishes LDS writes only: it does NOT drain outstanding global loads
// (a __syncthreads() would wait vmcnt(0) and cancel the prefetch that is meant to stay in flight).
__device__ __forceinline__ void qk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool CONJB, int PN, int KTL, int NW, int PMT, bool PROF = false>
__device__ __forceinline__ void zgemm_deep(double* __restrict__ Cre, double* __restrict__ Cim, const int ldc,
                                           const double* __restrict__ Are, const double* __restrict__ Aim, const int lda,
                                           const double* __restrict__ Bre, const double* __restrict__ Bim, const int ldb,
                                           const int M, const int N, const int Ktrue, double* __restrict__ lds, long long (&pc)[8], const int dbg = 0) {
  using G = GemmCfg<PN, KTL, NW, PMT>;
  constexpr int PM = G::PM;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int npm = (M + PM - 1) / PM;
  const int npn = (N + PN - 1) / PN;
  const int nk = (Ktrue + KTL - 1) / KTL;
  const int k4 = (Ktrue + 3) >> 2;
  const int total = npm * npn * nk;

  double2 ra0[2 * G::UA], rb0[2 * G::UB], ra1[2 * G::UA], rb1[2 * G::UB];
#pragma unroll
  for (int i = 0; i < 2 * G::UA; ++i) ra0[i] = ra1[i] = make_double2(0.0, 0.0);
#pragma unroll
  for (int i = 0; i < 2 * G::UB; ++i) rb0[i] = rb1[i] = make_double2(0.0, 0.0);

  int f_kt = 0, f_pm = 0, f_pn = 0;
  // Per-thread staging coordinates are fixed for the whole GEMM: row offset (in elements) and
  // column of each 16-byte unit.  Per step only a wave-uniform base (SGPR pair) changes, so a load
  // costs a clamp, an add and the instruction itself instead of 64-bit per-lane address math.
  unsigned rowoffA[G::UA], rowoffB[G::UB];
  int colA[G::UA], colB[G::UB];
#pragma unroll
  for (int i = 0; i < G::UA; ++i) {
    const int u = tid + G::WGT * i;
    rowoffA[i] = (unsigned)((u / (PM / 2)) * lda);
    colA[i] = (u % (PM / 2)) * 2;
  }
#pragma unroll
  for (int i = 0; i < G::UB; ++i) {
    const int u = tid + G::WGT * i;
    rowoffB[i] = (unsigned)((u / (PN / 2)) * ldb);
    colB[i] = (u % (PN / 2)) * 2;
  }
#define QK_FETCH_SET(RA, RB)                                                      \
  do {                                                                            \
    const int m0_ = f_pm * PM, n0_ = f_pn * PN;                                   \
    const int mcols_ = min(PM, M - m0_), ncols_ = min(PN, N - n0_);               \
    const long long ka_ = (long long)f_kt * KTL * lda + m0_;                      \
    const long long kb_ = (long long)f_kt * KTL * ldb + n0_;                      \
    const double* are_ = Are + ka_;                                               \
    const double* aim_ = Aim + ka_;                                               \
    const double* bre_ = Bre + kb_;                                               \
    const double* bim_ = Bim + kb_;                                               \
    _Pragma("unroll") for (int i = 0; i < G::UA; ++i) {                           \
      const unsigned o = rowoffA[i] + (unsigned)min(colA[i], mcols_ - 2);         \
      RA[2 * i] = *reinterpret_cast<const double2*>(are_ + o);                    \
      RA[2 * i + 1] = *reinterpret_cast<const double2*>(aim_ + o);                \
    }                                                                             \
    _Pragma("unroll") for (int i = 0; i < G::UB; ++i) {                           \
      const unsigned o = rowoffB[i] + (unsigned)min(colB[i], ncols_ - 2);         \
      RB[2 * i] = *reinterpret_cast<const double2*>(bre_ + o);                    \
      RB[2 * i + 1] = *reinterpret_cast<const double2*>(bim_ + o);                \
    }                                                                             \
    if (++f_kt == nk) {                                                           \
      f_kt = 0;                                                                   \
      if (++f_pm == npm) f_pm = 0, ++f_pn;                                        \
    }                                                                             \
  } while (0)
#define QK_STASH_SET(BUF, RA, RB)                                                 \
  do {                                                                            \
    double* base_ = lds + (BUF)*G::STAGE_D;                                       \
    _Pragma("unroll") for (int i = 0; i < G::UA; ++i) {                           \
      const int u = tid + G::WGT * i;                                             \
      const int o = (u / (PM / 2)) * PM + (u % (PM / 2)) * 2;                     \
      *reinterpret_cast<double2*>(base_ + o) = RA[2 * i];                         \
      *reinterpret_cast<double2*>(base_ + G::A_PLANE + o) = RA[2 * i + 1];        \
    }                                                                             \
    _Pragma("unroll") for (int i = 0; i < G::UB; ++i) {                           \
      const int u = tid + G::WGT * i;                                             \
      const int o = (u / (PN / 2)) * PN + (u % (PN / 2)) * 2;                     \
      *reinterpret_cast<double2*>(base_ + 2 * G::A_PLANE + o) = RB[2 * i];        \
      *reinterpret_cast<double2*>(base_ + 2 * G::A_PLANE + G::B_PLANE + o) = RB[2 * i + 1]; \
    }                                                                             \
  } while (0)

  v4d cre[G::MAXT], cim[G::MAXT];
  int tm[G::MAXT], tn[G::MAXT];
  int cnt = 0;
  int c_kt = 0, c_pm = 0, c_pn = 0;
  auto compute_step = [&](int buf) __attribute__((always_inline)) {
    const int m0 = c_pm * PM, n0 = c_pn * PN;
    if (c_kt == 0) {
      const int mt = min(PM / TILE, (M - m0) / TILE);
      const int nt = min(PN / TILE, (N - n0) / TILE);
      const int vt = mt * nt;
      cnt = (vt > wave) ? (vt - wave + NW - 1) / NW : 0;
#pragma unroll
      for (int e = 0; e < G::MAXT; ++e) {
        cre[e] = (v4d){0, 0, 0, 0};
        cim[e] = (v4d){0, 0, 0, 0};
        const int t = min(wave + NW * e, vt - 1);
        tm[e] = t % mt;
        tn[e] = t / mt;
      }
    }
    const double* base = lds + buf * G::STAGE_D;
    const int ksteps = min(KTL / 4, k4 - c_kt * (KTL / 4));
    if (!(dbg & 4)) {
      if (ksteps == KTL / 4)
        mma_ktile<CONJB, PM, PN, G::A_PLANE, G::B_PLANE, KTL / 4, G::MAXT, true, true>(cre, cim, tm, tn, base, q, j, cnt, ksteps);
      else
        mma_ktile<CONJB, PM, PN, G::A_PLANE, G::B_PLANE, KTL / 4, G::MAXT, false, true>(cre, cim, tm, tn, base, q, j, cnt, ksteps);
    }
    if (c_kt == nk - 1 && !(dbg & 1)) {
#pragma unroll
      for (int e = 0; e < G::MAXT; ++e) {
        if (e < cnt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long long o = (long long)(m0 + tm[e] * TILE + q + 4 * r) * ldc + n0 + tn[e] * TILE + j;
            Cre[o] = cre[e][r];
            Cim[o] = cim[e][r];
          }
        }
      }
    }
    if (++c_kt == nk) {
      c_kt = 0;
      if (++c_pm == npm) c_pm = 0, ++c_pn;
    }
  };

  // prologue: tiles 0 and 1 in flight, tile 0 published
  QK_T(5, {
    QK_FETCH_SET(ra0, rb0);
    if (total > 1) QK_FETCH_SET(ra1, rb1);
    QK_STASH_SET(0, ra0, rb0);
    qk_lds_barrier();
  });
  int s = 0;
  while (s + 3 < total) {  // tiles s+2 and s+3 exist: both fetches unconditional
    if (!(dbg & 2)) QK_T(0, QK_FETCH_SET(ra0, rb0));     // tile s+2
    QK_T(1, compute_step(0));            // tile s     (s is even here)
    if (!(dbg & 2)) QK_T(3, QK_STASH_SET(1, ra1, rb1));  // tile s+1, fetched two steps ago
    if (!(dbg & 8)) QK_T(4, qk_lds_barrier());
    if (!(dbg & 2)) QK_T(0, QK_FETCH_SET(ra1, rb1));     // tile s+3
    QK_T(1, compute_step(1));            // tile s+1
    if (!(dbg & 2)) QK_T(3, QK_STASH_SET(0, ra0, rb0));  // tile s+2
    if (!(dbg & 8)) QK_T(4, qk_lds_barrier());
    s += 2;
  }
  for (; s < total; ++s) {  // tail (at most 3 steps); s keeps its parity convention
    const bool even = (s & 1) == 0;
    QK_T(0, {
      if (s + 2 < total) {
        if (even) QK_FETCH_SET(ra0, rb0); else QK_FETCH_SET(ra1, rb1);
      }
    });
    QK_T(1, compute_step(s & 1));
    QK_T(3, {
      if (s + 1 < total) {
        if (even) QK_STASH_SET(1, ra1, rb1); else QK_STASH_SET(0, ra0, rb0);
      }
    });
    QK_T(4, qk_lds_barrier());
  }
#undef QK_FETCH_SET
#undef QK_STASH_SET
  QK_T(6, __syncthreads());
}

template <int PN, int KTL, int OCC, int NW, int PMT, bool PROF = false>
__global__ __launch_bounds__(64 * NW, OCC) void qk_sweep_deep_kernel(const SweepArgs g) {
  using G = GemmCfg<PN, KTL, NW, PMT>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  long long* slot = reinterpret_cast<long long*>(lds + G::LDS_D);
  double* Xre = g.scratch + (long long)blockIdx.x * 2 * (g.x_plane + g.t_plane);
  double* Xim = Xre + g.x_plane;
  double* Tre = Xim + g.x_plane;
  double* Tim = Tre + g.t_plane;
  const int tid = threadIdx.x;
  // Static priority for one of the two workgroups that share a CU: it wins the matrix pipe, finishes its
  // MFMA phase first and does its fetch/stash/barrier while the other one computes (they alternate
  // instead of falling into lock-step).  Which blocks share a CU is not defined; both guesses are offered.
  if ((g.prio_mode == 1 && blockIdx.x >= gridDim.x / 2) || (g.prio_mode == 2 && (blockIdx.x & 1))) __builtin_amdgcn_s_setprio(1);
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = PROF ? qk_stamp() : 0;
  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p >= g.npairs) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    // Stage the pair's per-site metadata in LDS once (one coalesced pass) instead of chasing it
    // through global memory at every site: [xd | yd | xt | yt] (n+1 ints each) then [xo | yo] (n int64).
    const int n1 = g.n_sites + 1;
    int* m_xd = reinterpret_cast<int*>(slot + 2);
    int* m_yd = m_xd + n1;
    int* m_xt = m_yd + n1;
    int* m_yt = m_xt + n1;
    long long* m_xo = reinterpret_cast<long long*>(m_xd + 4 * n1 + (4 * n1 & 1));
    long long* m_yo = m_xo + g.n_sites;
    for (int e = tid; e < n1; e += 64 * NW) {
      m_xd[e] = g.xdims[(long long)xi * n1 + e];
      m_yd[e] = g.ydims[(long long)yj * n1 + e];
      m_xt[e] = g.xtrue[(long long)xi * n1 + e];
      m_yt[e] = g.ytrue[(long long)yj * n1 + e];
      if (e < g.n_sites) {
        m_xo[e] = g.xoffs[(long long)xi * g.n_sites + e];
        m_yo[e] = g.yoffs[(long long)yj * g.n_sites + e];
      }
    }
    __syncthreads();
    auto ldi = [&](const int* q_) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*q_); };
    auto ldl = [&](const long long* q_) __attribute__((always_inline)) {
      const long long v = *q_;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
      return (long long)(((unsigned long long)hi << 32) | lo);
    };
    {
      const int a = ldi(m_xd), b = ldi(m_yd);
      for (int e = tid; e < a * b; e += 64 * NW) {
        Xre[e] = (e == 0) ? 1.0 : 0.0;
        Xim[e] = 0.0;
      }
      __syncthreads();
    }
    for (int k = 0; k < g.n_sites; ++k) {
      const int a = ldi(m_xd + k), a2 = ldi(m_xd + k + 1), b = ldi(m_yd + k), b2 = ldi(m_yd + k + 1);
      const double* Are = g.xdata + ldl(m_xo + k);
      const double* Aim = Are + (long long)a * 2 * a2;
      const double* Bre = g.ydata + ldl(m_yo + k);
      const double* Bim = Bre + (long long)b * 2 * b2;
      zgemm_deep<false, PN, KTL, NW, PMT, PROF>(Tre, Tim, 2 * b2, Xre, Xim, a, Bre, Bim, 2 * b2, a, 2 * b2, ldi(m_yt + k), lds, pc, g.debug_flags);
      zgemm_deep<true, PN, KTL, NW, PMT, PROF>(Xre, Xim, a2, Tre, Tim, b2, Are, Aim, a2, b2, a2, 2 * ldi(m_xt + k), lds, pc, g.debug_flags);
    }
    if (tid == 0) {
      const double re = Xre[0], im = Xim[0];
      g.values[p] = re * re + im * im;
      if (g.z) {
        g.z[2 * p] = re;
        g.z[2 * p + 1] = im;
      }
    }
    __syncthreads();
  }
  if (PROF && g.prof && (tid & 63) == 0) {
    pc[7] = qk_stamp() - t_begin;
#pragma unroll
    for (int c = 0; c < 8; ++c) atomicAdd(g.prof + c, (unsigned long long)pc[c]);
  }
}

// ----------------------------------------------------------------------------------------
// v6: lean steady state.  Same pipeline as zgemm_deep (flat step sequence, two-step-deep register
// prefetch, raw LDS barrier), but everything that depends only on the pass is computed once per pass:
// running operand pointers (+= one K-tile per step), clamped per-thread load offsets, LDS fragment
// offsets of the wave's tiles and their output addresses.  A steady-state step is then 4 loads,
// the MFMA block, 4 LDS stores, one barrier and a handful of scalar adds.  Written for the 8-wave,
// 64x64, K-tile-16 configuration (one 16-byte staging unit per thread and operand plane).
// ----------------------------------------------------------------------------------------
template <bool CONJB, bool FULLK>
__device__ __forceinline__ void mma_lean(v4d (&cre)[2], v4d (&cim)[2], const int (&la)[2], const int (&lb)[2],
                                         const double* __restrict__ base, const int cnt, const int ksteps) {
  constexpr int PMN = 64, APL = 16 * 64, BPL = 16 * 64;  // staged planes: A re | A im | B re | B im
  if constexpr (FULLK) {
    double far[2][2], fai[2][2], fbr[2][2], fbi[2][2];
    auto load = [&](int g, int buf) __attribute__((always_inline)) {
      const int e = g >> 1, k0 = (g & 1) * 2;
      const double* pa = base + la[e] + 4 * k0 * PMN;
      const double* pb = base + lb[e] + 4 * k0 * PMN;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        far[buf][h] = pa[h * 4 * PMN];
        fai[buf][h] = pa[APL + h * 4 * PMN];
        fbr[buf][h] = pb[h * 4 * PMN];
        fbi[buf][h] = CONJB ? -pb[BPL + h * 4 * PMN] : pb[BPL + h * 4 * PMN];
      }
    };
    if (cnt > 0) load(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int e = g >> 1;
      if (e < cnt) {
        if (g + 1 < 4 && ((g + 1) >> 1) < cnt) load(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double ar = far[g & 1][h], ai = fai[g & 1][h], br = fbr[g & 1][h], bi = fbi[g & 1][h];
          cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cre[e], 0, 0, 0);
          cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, cim[e], 0, 0, 0);
          cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cre[e], 0, 0, 0);
          cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, cim[e], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (e < cnt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < ksteps) {
            const double* pa = base + la[e] + 4 * ks * PMN;
            const double* pb = base + lb[e] + 4 * ks * PMN;
            const double ar = pa[0], ai = pa[APL], br = pb[0];
            double bi = pb[BPL];
            if (CONJB) bi = -bi;
            cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cre[e], 0, 0, 0);
            cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, cim[e], 0, 0, 0);
            cre[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cre[e], 0, 0, 0);
            cim[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, cim[e], 0, 0, 0);
          }
        }
      }
    }
  }
}

template <bool CONJB>
__device__ __forceinline__ void zgemm_lean(double* __restrict__ Cre, double* __restrict__ Cim, const int ldc,
                                           const double* __restrict__ Are, const double* __restrict__ Aim, const int lda,
                                           const double* __restrict__ Bre, const double* __restrict__ Bim, const int ldb,
                                           const int M, const int N, const int Ktrue, double* __restrict__ lds) {
  constexpr int PM = 64, PN = 64, KTL = 16, NW = 8;
  constexpr int APL = KTL * PM, STAGE_D = 4 * APL;  // doubles per plane / per buffer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int npm = (M + PM - 1) / PM, npn = (N + PN - 1) / PN;
  const int nk = (Ktrue + KTL - 1) / KTL;
  const int ks_last = ((Ktrue + 3) >> 2) - (nk - 1) * (KTL / 4);  // k-steps of the last K-tile (1..4)
  const int total = npm * npn * nk;
  const long long sA = (long long)KTL * lda, sB = (long long)KTL * ldb;

  // per-thread staging role: one 16-byte unit per plane; LDS position is simply 2*tid
  const int srow = tid >> 5, scol = (tid & 31) * 2;
  const unsigned rA = (unsigned)(srow * lda), rB = (unsigned)(srow * ldb);
  double* st0 = lds + 2 * tid;             // buffer 0
  double* st1 = st0 + STAGE_D;             // buffer 1

  // ---- fetch-side pass state
  int f_pm = 0, f_pn = 0, f_left = nk;
  const double *fa_re = Are, *fa_im = Aim, *fb_re = Bre, *fb_im = Bim;
  unsigned offA = rA + (unsigned)min(scol, min(PM, M) - 2), offB = rB + (unsigned)min(scol, min(PN, N) - 2);
  auto fetch_next_pass = [&]() __attribute__((always_inline)) {
    if (++f_pm == npm) f_pm = 0, ++f_pn;
    const int m0 = f_pm * PM, n0 = f_pn * PN;
    fa_re = Are + m0, fa_im = Aim + m0, fb_re = Bre + n0, fb_im = Bim + n0;
    offA = rA + (unsigned)min(scol, min(PM, M - m0) - 2);
    offB = rB + (unsigned)min(scol, min(PN, N - n0) - 2);
    f_left = nk;
  };
  double2 a0r, a0i, b0r, b0i, a1r, a1i, b1r, b1i;  // two staging register sets (always loaded before they are stashed)
#define QK_LFETCH(AR, AI, BR, BI)                              \
  do {                                                         \
    AR = *reinterpret_cast<const double2*>(fa_re + offA);      \
    AI = *reinterpret_cast<const double2*>(fa_im + offA);      \
    BR = *reinterpret_cast<const double2*>(fb_re + offB);      \
    BI = *reinterpret_cast<const double2*>(fb_im + offB);      \
    fa_re += sA, fa_im += sA, fb_re += sB, fb_im += sB;        \
    if (--f_left == 0) fetch_next_pass();                      \
  } while (0)
#define QK_LSTASH(ST, AR, AI, BR, BI)                          \
  do {                                                         \
    *reinterpret_cast<double2*>(ST) = AR;                      \
    *reinterpret_cast<double2*>(ST + APL) = AI;                \
    *reinterpret_cast<double2*>(ST + 2 * APL) = BR;            \
    *reinterpret_cast<double2*>(ST + 3 * APL) = BI;            \
  } while (0)

  // ---- compute-side pass state
  int c_pm = 0, c_pn = 0, c_left = nk, cnt = 0;
  int la[2], lb[2];
  long long co[2];
  v4d cre[2], cim[2];
  auto compute_pass_setup = [&]() __attribute__((always_inline)) {
    const int m0 = c_pm * PM, n0 = c_pn * PN;
    const int mt = min(PM / TILE, (M - m0) / TILE), nt = min(PN / TILE, (N - n0) / TILE);
    const int vt = mt * nt;
    cnt = (vt > wave) ? (vt - wave + NW - 1) / NW : 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int t = min(wave + NW * e, vt - 1);
      const int tm = t % mt, tn = t / mt;
      la[e] = q * PM + tm * TILE + j;
      lb[e] = 2 * APL + q * PN + tn * TILE + j;
      co[e] = (long long)(m0 + tm * TILE + q) * ldc + n0 + tn * TILE + j;
      cre[e] = (v4d){0, 0, 0, 0};
      cim[e] = (v4d){0, 0, 0, 0};
    }
    c_left = nk;
  };
  compute_pass_setup();
  const long long crow = 4ll * ldc;
  auto step = [&](const double* base) __attribute__((always_inline)) {
    if (c_left > 1 || ks_last == KTL / 4) {
      mma_lean<CONJB, true>(cre, cim, la, lb, base, cnt, KTL / 4);
    } else {
      mma_lean<CONJB, false>(cre, cim, la, lb, base, cnt, ks_last);
    }
    if (--c_left == 0) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e < cnt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Cre[co[e] + r * crow] = cre[e][r];
            Cim[co[e] + r * crow] = cim[e][r];
          }
        }
      }
      if (++c_pm == npm) c_pm = 0, ++c_pn;
      if (c_pn < npn) compute_pass_setup();
    }
  };

  QK_LFETCH(a0r, a0i, b0r, b0i);
  if (total > 1) QK_LFETCH(a1r, a1i, b1r, b1i);
  QK_LSTASH(st0, a0r, a0i, b0r, b0i);
  qk_lds_barrier();
  int s = 0;
  while (s + 3 < total) {
    QK_LFETCH(a0r, a0i, b0r, b0i);        // tile s+2
    step(lds);                            // tile s (buffer 0)
    QK_LSTASH(st1, a1r, a1i, b1r, b1i);   // tile s+1
    qk_lds_barrier();
    QK_LFETCH(a1r, a1i, b1r, b1i);        // tile s+3
    step(lds + STAGE_D);                  // tile s+1 (buffer 1)
    QK_LSTASH(st0, a0r, a0i, b0r, b0i);   // tile s+2
    qk_lds_barrier();
    s += 2;
  }
  for (; s < total; ++s) {
    const bool even = (s & 1) == 0;
    if (s + 2 < total) {
      if (even) QK_LFETCH(a0r, a0i, b0r, b0i); else QK_LFETCH(a1r, a1i, b1r, b1i);
    }
    step(even ? lds : lds + STAGE_D);
    if (s + 1 < total) {
      if (even) QK_LSTASH(st1, a1r, a1i, b1r, b1i); else QK_LSTASH(st0, a0r, a0i, b0r, b0i);
    }
    qk_lds_barrier();
  }
#undef QK_LFETCH
#undef QK_LSTASH
  __syncthreads();
}

// ----------------------------------------------------------------------------------------
// v8: ring GEMM.  Same 64x64 pass / 8-wave / 2-tiles-per-wave decomposition as zgemm_lean, but
//   * staging is LDS-DMA (global_load_lds, 16 B per lane) into a ring of three K-tile-8 slots, two
//     K-tiles in flight across the raw barrier, retired by a counted s_waitcnt vmcnt -- no staging
//     registers and no ds_write pass;
//   * the registers this frees hold a third accumulator per tile, so the complex product is the 3M form
//       P1 += ar*br, P2 += ai*bi, P3 += (ar+ai)*(br+sbi),  sbi = +bi (plain) | -bi (conjugated B)
//       re = P1 - P2 | P1 + P2,   im = P3 - P1 - P2 | P3 - P1 + P2
//     three MFMAs per complex k-step instead of four (operand sums: two v_add_f64 on the fragments).
// Staging roles: waves 0-3 bring the re planes, waves 4-7 the im planes; wave w covers K rows
// 2(w&3), 2(w&3)+1 of both operands (one 1-KiB wave-linear piece of the A plane and one of the B plane).
// ----------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void qk_wait_const() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void qk_wait_vm(const int n) {  // wave-uniform n; values above 24 wait for everything
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Scalar traits of the ring GEMM: accumulator vector, MFMA and where an accumulator register lands in the tile.
typedef float v4f __attribute__((ext_vector_type(4)));
template <typename T>
struct QkScalar;
template <>
struct QkScalar<double> {
  using v4 = v4d;
  static constexpr int ROW_Q = 1, ROW_R = 4;  // v_mfma_f64_16x16x4_f64: register r of lane (q, j) = C[q + 4r][j]
  static __device__ __forceinline__ v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};
template <>
struct QkScalar<float> {
  using v4 = v4f;
  static constexpr int ROW_Q = 4, ROW_R = 1;  // v_mfma_f32_16x16x4_f32: register r of lane (q, j) = C[4q + r][j]
  static __device__ __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// One complex k-step of one tile.  M3: 3M form (three independent accumulators c1 = P1, c2 = P2, c3 = P3);
// otherwise the plain four-product form (c1 = re, c2 = im, c3 unused).
template <bool CONJB, bool M3, typename T>
__device__ __forceinline__ void mma3_kstep(typename QkScalar<T>::v4& c1, typename QkScalar<T>::v4& c2, typename QkScalar<T>::v4& c3,
                                           const T ar, const T ai, const T br, const T bi) {
  using S = QkScalar<T>;
  if constexpr (M3) {
    const T sa = ar + ai, sb = CONJB ? br - bi : br + bi;
    c1 = S::mfma(ar, br, c1);
    c2 = S::mfma(ai, bi, c2);
    c3 = S::mfma(sa, sb, c3);
  } else {
    const T sbi = CONJB ? -bi : bi;
    c1 = S::mfma(ar, br, c1);
    c2 = S::mfma(ar, sbi, c2);
    c1 = S::mfma(-ai, sbi, c1);
    c2 = S::mfma(ai, br, c2);
  }
}

template <bool CONJB, int CNT, bool FULLK, bool M3, int KTL, int PN = 64, typename T = double>
__device__ __forceinline__ void mma_ring3(typename QkScalar<T>::v4 (&c1)[2], typename QkScalar<T>::v4 (&c2)[2], typename QkScalar<T>::v4 (&c3)[2],
                                          const int (&la)[2], const int (&lb)[2], const T* __restrict__ base, const int ksteps) {
  constexpr int PM = 64, APL = KTL * PM, BPL = KTL * PN, KS = KTL / 4;  // staged planes: A re | A im | B re | B im
  if constexpr (CNT == 0) return;
  if constexpr (FULLK) {
    T far[2], fai[2], fbr[2], fbi[2];
    auto load = [&](int g, int buf) __attribute__((always_inline)) {
      const int e = g / KS, ks = g % KS;
      const T* pa = base + la[e] + 4 * ks * PM;
      const T* pb = base + lb[e] + 4 * ks * PN;
      far[buf] = pa[0];
      fai[buf] = pa[APL];
      fbr[buf] = pb[0];
      fbi[buf] = pb[BPL];
    };
    load(0, 0);
#pragma unroll
    for (int g = 0; g < KS * CNT; ++g) {
      const int e = g / KS;
      if (g + 1 < KS * CNT) load(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma3_kstep<CONJB, M3, T>(c1[e], c2[e], c3[e], far[g & 1], fai[g & 1], fbr[g & 1], fbi[g & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {  // the last K-tile of a K range that does not fill it: ksteps in 1..KS-1
#pragma unroll
    for (int e = 0; e < CNT; ++e) {
#pragma unroll
      for (int ks = 0; ks < KS - 1; ++ks) {
        if (ks < ksteps) {
          const T* pa = base + la[e] + 4 * ks * PM;
          const T* pb = base + lb[e] + 4 * ks * PN;
          mma3_kstep<CONJB, M3, T>(c1[e], c2[e], c3[e], pa[0], pa[APL], pb[0], pb[BPL]);
        }
      }
    }
  }
}

template <bool CONJB, int KTL, int NSLOT, bool M3, int NW = 8, int PN = 64, typename T = double>
__device__ __forceinline__ void zgemm_ring3(T* __restrict__ Cre, T* __restrict__ Cim, const int ldc,
                                            const T* __restrict__ Are, const T* __restrict__ Aim, const int lda,
                                            const T* __restrict__ Bre, const T* __restrict__ Bim, const int ldb,
                                            const int M, const int N, const int Ktrue, T* __restrict__ lds) {
  using S = QkScalar<T>;
  using V4 = typename S::v4;
  constexpr int EPL = 16 / (int)sizeof(T);            // elements per lane and LDS-DMA (2 doubles / 4 floats)
  constexpr int CHUNK = 1024 / (int)sizeof(T);        // elements per 1-KiB wave-linear piece
  constexpr int RPC = CHUNK / 64;                     // K rows of a 64-wide plane per piece (2 / 4)
  constexpr bool SPLIT = (KTL == 4 * RPC);            // planes of four pieces: waves 0-3 take re, waves 4-7 im
  static_assert((NW == 8 && PN == 64 && (KTL == 4 * RPC || KTL == 8 * RPC)) || (NW == 4 && PN == 32 && KTL == 8 && sizeof(T) == 8), "supported shapes");
  constexpr int PM = 64;
  constexpr int APL = KTL * PM, BPL = KTL * PN, SLOT_D = 2 * APL + 2 * BPL;  // doubles per plane / per ring slot
  constexpr int DEPTH = NSLOT - 1;                 // K-tiles in flight ahead of the one being multiplied
  constexpr int LPS = (NW == 4) ? 3 : (SPLIT ? 2 : 4);  // LDS-DMA instructions per wave and K-tile
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int npm = (M + PM - 1) / PM, npn = (N + PN - 1) / PN;
  const int nk = (Ktrue + KTL - 1) / KTL;
  const int ks_last = ((Ktrue + 3) >> 2) - (nk - 1) * (KTL / 4);  // k-steps of the last K-tile (1..KTL/4)
  const int total = npm * npn * nk;
  const long long sA = (long long)KTL * lda, sB = (long long)KTL * ldb;

  // ---- staging role of this wave / lane: 1-KiB wave-linear pieces of the staged planes.
  // 8 waves, K-tile 8: waves 0-3 bring the re planes, waves 4-7 the im planes (K rows 2(w&3), +1 of A and of B).
  // 8 waves, K-tile 16: every wave brings rows 2w, 2w+1 of all four planes.
  // 4 waves (64x32 pass), K-tile 8: wave w brings rows 2w, 2w+1 of A re and A im, and one of the four 1-KiB pieces
  // of the B planes (plane w>>1, K rows 4(w&1) .. +3; a B row is 32 doubles).
  const int w3 = (NW == 8 && SPLIT) ? (wave & 3) : wave;
  const int pl = (NW == 8 && SPLIT) ? (wave >> 2) : 0;
  const long long a_im = Aim - Are, b_im = Bim - Bre;  // plane strides of the operands
  const T* const Asrc = pl ? Aim : Are;
  const T* const Bsrc = (NW == 4) ? ((wave >> 1) ? Bim : Bre) : (pl ? Bim : Bre);
  constexpr int LPR = 64 / RPC;                        // lanes per 64-wide K row of a piece
  const int srow = RPC * w3 + lane / LPR, scol = (lane % LPR) * EPL;
  const int srowB = (NW == 4) ? 4 * (wave & 1) + (lane >> 4) : srow;
  const int scolB = (NW == 4) ? (lane & 15) * 2 : scol;
  const unsigned rA = (unsigned)(srow * lda), rB = (unsigned)(srowB * ldb);
  T* const dA = lds + pl * APL + w3 * CHUNK;           // slot 0 destinations (wave-uniform)
  T* const dB = (NW == 4) ? lds + 2 * APL + (wave >> 1) * BPL + (wave & 1) * CHUNK : lds + 2 * APL + pl * BPL + w3 * CHUNK;

  // ---- fetch-side pass state (runs DEPTH K-tiles ahead of the compute side, across pass boundaries)
  int f_pm = 0, f_pn = 0, f_left = nk, f_slot = 0;
  const T *fa = Asrc, *fb = Bsrc;
  unsigned offA = rA + (unsigned)min(scol, min(PM, M) - EPL), offB = rB + (unsigned)min(scolB, min(PN, N) - EPL);
  auto fetch = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds(fa + offA, (lds_ptr_t)(dA + f_slot), 16, 0, 0);
    if constexpr (!SPLIT || NW == 4) __builtin_amdgcn_global_load_lds(fa + a_im + offA, (lds_ptr_t)(dA + APL + f_slot), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(fb + offB, (lds_ptr_t)(dB + f_slot), 16, 0, 0);
    if constexpr (!SPLIT && NW == 8) __builtin_amdgcn_global_load_lds(fb + b_im + offB, (lds_ptr_t)(dB + BPL + f_slot), 16, 0, 0);
    fa += sA, fb += sB;
    f_slot = (f_slot == (NSLOT - 1) * SLOT_D) ? 0 : f_slot + SLOT_D;
    if (--f_left == 0) {
      if (++f_pm == npm) f_pm = 0, ++f_pn;
      const int m0 = f_pm * PM, n0 = f_pn * PN;
      fa = Asrc + m0, fb = Bsrc + n0;
      offA = rA + (unsigned)min(scol, min(PM, M - m0) - EPL);
      offB = rB + (unsigned)min(scolB, min(PN, N - n0) - EPL);
      f_left = nk;
    }
  };

  // ---- step counters shared by all passes
  int s = 0, c_slot = 0, pend = 0;
  const int crow = S::ROW_R * ldc;

  // One pass = nk steps on one 64x64 output tile, specialised on the number of tiles this wave owns so that
  // the MFMA block is branch-free and the accumulators live only inside the pass.
  auto run_pass = [&](auto cnt_tag, const int m0, const int n0, const int mt, const int vt) __attribute__((always_inline)) {
    constexpr int CNT = decltype(cnt_tag)::value;
    int la[2], lb[2], co[2];
    V4 c1[2], c2[2], c3[2];
    const int inv = (mt == 1) ? 32 : (mt == 2) ? 16 : (mt == 3) ? 11 : 8;  // t / mt == (t * inv) >> 5 for t < 16
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int t = min(wave + NW * e, vt - 1);
      const int tn = (t * inv) >> 5, tm = t - tn * mt;
      la[e] = q * PM + tm * TILE + j;
      lb[e] = 2 * APL + q * PN + tn * TILE + j;
      co[e] = (m0 + tm * TILE + S::ROW_Q * q) * ldc + n0 + tn * TILE + j;
      c1[e] = (V4){0, 0, 0, 0};
      c2[e] = (V4){0, 0, 0, 0};
      c3[e] = (V4){0, 0, 0, 0};
    }
    for (int kt = 0; kt < nk; ++kt, ++s) {  // (ring)
      if (s + DEPTH < total) fetch();                    // K-tile s+DEPTH -> the slot read in step s-1
      const T* base = lds + c_slot;
      if (kt + 1 < nk || ks_last == KTL / 4) mma_ring3<CONJB, CNT, true, M3, KTL, PN, T>(c1, c2, c3, la, lb, base, KTL / 4);
      else mma_ring3<CONJB, CNT, false, M3, KTL, PN, T>(c1, c2, c3, la, lb, base, ks_last);
      if (s + 1 < total) {
        // K-tile s+1 must have landed; everything issued after it may stay in flight: the younger K-tiles and,
        // when it was issued before the previous step's epilogue (DEPTH >= 2), that epilogue's stores
        const int young = LPS * min(DEPTH - 1, total - 2 - s);
        if (DEPTH == 1 || pend == 0) {  // the common case first: the generic switch costs a branch tree per step
          if (young == LPS * (DEPTH - 1)) qk_wait_const<LPS * (DEPTH - 1)>();
          else qk_wait_vm(young);
        } else {
          qk_wait_vm(young + pend);
        }
        qk_lds_barrier();
      }
      pend = 0;
      c_slot = (c_slot == (NSLOT - 1) * SLOT_D) ? 0 : c_slot + SLOT_D;
    }
    if constexpr (CNT > 0) {
#pragma unroll
      for (int e = 0; e < CNT; ++e) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (M3) {
            const T p1 = c1[e][r], p2 = c2[e][r], p3 = c3[e][r];
            Cre[co[e] + r * crow] = CONJB ? p1 + p2 : p1 - p2;
            Cim[co[e] + r * crow] = CONJB ? (p3 - p1) + p2 : (p3 - p1) - p2;
          } else {
            Cre[co[e] + r * crow] = c1[e][r];
            Cim[co[e] + r * crow] = c2[e][r];
          }
        }
      }
      pend = 8 * CNT;
    }
  };

#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    if (i < total) fetch();
  qk_wait_vm(LPS * (min(DEPTH, total) - 1));
  qk_lds_barrier();
  for (int pn = 0; pn < npn; ++pn) {
    for (int pm = 0; pm < npm; ++pm) {
      const int m0 = pm * PM, n0 = pn * PN;
      const int mt = min(PM / TILE, (M - m0) / TILE), nt = min(PN / TILE, (N - n0) / TILE);
      const int vt = mt * nt;
      const int cnt = (vt > wave) ? (vt - wave + NW - 1) / NW : 0;
      if (cnt == 2) run_pass(std::integral_constant<int, 2>{}, m0, n0, mt, vt);
      else if (cnt == 1) run_pass(std::integral_constant<int, 1>{}, m0, n0, mt, vt);
      else run_pass(std::integral_constant<int, 0>{}, m0, n0, mt, vt);
    }
  }
  __syncthreads();
}

// the deep kernel's pair loop around the lean GEMM
// MODE 0: lean GEMM (register staging).  MODE 1 (shipped) / 2: ring GEMM, K-tile 8, 3 / 4 slots, 3M product.
// MODE 4: ring GEMM, K-tile 16, 2 slots (one K-tile in flight), 3M product.  (MODE 3 = 3 slots + four-product MFMA, for a
// three-workgroups-per-CU build, is not instantiated: at 80 VGPRs it spills into the K loop.)
//         MODE 5: ring GEMM on 4-wave workgroups (64x32 pass, K-tile 8, 3 slots, 3M), four workgroups per CU.
template <int OCC, int MODE = 0>
__global__ __launch_bounds__(MODE == 5 ? 256 : 512, OCC) void qk_sweep_lean_kernel(const SweepArgs g) {
  using G = GemmCfg<64, 16, 8, 64>;
  constexpr int NW = (MODE == 5) ? 4 : 8;
  constexpr bool PROF = false;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int STAGE_DOUBLES = (MODE == 5) ? 3 * 1536 : (MODE == 1 || MODE == 3) ? 3 * 2048 : G::LDS_D;  // ring slots; lean / 4-slot ring: 64 KiB
  long long* slot = reinterpret_cast<long long*>(lds + STAGE_DOUBLES);       // then the pair slot and the per-site metadata
  double* Xre = g.scratch + (long long)blockIdx.x * 2 * (g.x_plane + g.t_plane);
  double* Xim = Xre + g.x_plane;
  double* Tre = Xim + g.x_plane;
  double* Tim = Tre + g.t_plane;
  const int tid = threadIdx.x;
  // Static priority for one of the two workgroups that share a CU: it wins the matrix pipe, finishes its
  // MFMA phase first and does its fetch/stash/barrier while the other one computes (they alternate
  // instead of falling into lock-step).  Which blocks share a CU is not defined; both guesses are offered.
  if ((g.prio_mode == 1 && blockIdx.x >= gridDim.x / 2) || (g.prio_mode == 2 && (blockIdx.x & 1))) __builtin_amdgcn_s_setprio(1);
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = PROF ? qk_stamp() : 0;
  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p >= g.npairs) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    // Stage the pair's per-site metadata in LDS once (one coalesced pass) instead of chasing it
    // through global memory at every site: [xd | yd | xt | yt] (n+1 ints each) then [xo | yo] (n int64).
    const int n1 = g.n_sites + 1;
    int* m_xd = reinterpret_cast<int*>(slot + 2);
    int* m_yd = m_xd + n1;
    int* m_xt = m_yd + n1;
    int* m_yt = m_xt + n1;
    long long* m_xo = reinterpret_cast<long long*>(m_xd + 4 * n1 + (4 * n1 & 1));
    long long* m_yo = m_xo + g.n_sites;
    for (int e = tid; e < n1; e += 64 * NW) {
      m_xd[e] = g.xdims[(long long)xi * n1 + e];
      m_yd[e] = g.ydims[(long long)yj * n1 + e];
      m_xt[e] = g.xtrue[(long long)xi * n1 + e];
      m_yt[e] = g.ytrue[(long long)yj * n1 + e];
      if (e < g.n_sites) {
        m_xo[e] = g.xoffs[(long long)xi * g.n_sites + e];
        m_yo[e] = g.yoffs[(long long)yj * g.n_sites + e];
      }
    }
    __syncthreads();
    auto ldi = [&](const int* q_) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*q_); };
    auto ldl = [&](const long long* q_) __attribute__((always_inline)) {
      const long long v = *q_;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
      return (long long)(((unsigned long long)hi << 32) | lo);
    };
    {
      const int a = ldi(m_xd), b = ldi(m_yd);
      for (int e = tid; e < a * b; e += 64 * NW) {
        Xre[e] = (e == 0) ? 1.0 : 0.0;
        Xim[e] = 0.0;
      }
      __syncthreads();
    }
    for (int k = 0; k < g.n_sites; ++k) {
      const int a = ldi(m_xd + k), a2 = ldi(m_xd + k + 1), b = ldi(m_yd + k), b2 = ldi(m_yd + k + 1);
      const double* Are = g.xdata + ldl(m_xo + k);
      const double* Aim = Are + (long long)a * 2 * a2;
      const double* Bre = g.ydata + ldl(m_yo + k);
      const double* Bim = Bre + (long long)b * 2 * b2;
      if constexpr (MODE != 0) {
        constexpr int KTL = (MODE == 4) ? 16 : 8;
        constexpr int NSLOT = (MODE == 2) ? 4 : (MODE == 4) ? 2 : 3;
        constexpr bool M3 = (MODE != 3);
        constexpr int PN = (MODE == 5) ? 32 : 64;
        zgemm_ring3<false, KTL, NSLOT, M3, NW, PN>(Tre, Tim, 2 * b2, Xre, Xim, a, Bre, Bim, 2 * b2, a, 2 * b2, ldi(m_yt + k), lds);
        zgemm_ring3<true, KTL, NSLOT, M3, NW, PN>(Xre, Xim, a2, Tre, Tim, b2, Are, Aim, a2, b2, a2, 2 * ldi(m_xt + k), lds);
      } else {
        zgemm_lean<false>(Tre, Tim, 2 * b2, Xre, Xim, a, Bre, Bim, 2 * b2, a, 2 * b2, ldi(m_yt + k), lds);
        zgemm_lean<true>(Xre, Xim, a2, Tre, Tim, b2, Are, Aim, a2, b2, a2, 2 * ldi(m_xt + k), lds);
      }
    }
    if (tid == 0) {
      const double re = Xre[0], im = Xim[0];
      g.values[p] = re * re + im * im;
      if (g.z) {
        g.z[2 * p] = re;
        g.z[2 * p + 1] = im;
      }
    }
    __syncthreads();
  }
  if (PROF && g.prof && (tid & 63) == 0) {
    pc[7] = qk_stamp() - t_begin;
#pragma unroll
    for (int c = 0; c < 8; ++c) atomicAdd(g.prof + c, (unsigned long long)pc[c]);
  }
}


// ----------------------------------------------------------------------------------------
// The ring sweep as its own kernel, templated on the scalar type:
//   T = double: the same code path as qk_sweep_lean_kernel<4, 1> (K-tile 8);
//   T = float : complex64 sweep on v_mfma_f32_16x16x4_f32 (K-tile 16: the same 16-KiB slots, pieces and roles).
// The MPS set is read as T planes with the SAME element offsets as the fp64 image (qk_mps_set_to_f32 converts
// element by element), the X/T scratch holds T, the outputs are doubles.
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512, 4) void qk_sweep_ring_kernel(const SweepArgs g) {
  constexpr int NW = 8;
  constexpr int KTL = 32 / (int)sizeof(T) * 2;  // 8 rows of doubles, 16 rows of floats: 16-KiB slots either way
  constexpr int SLOT_BYTES = 16 * 1024, NSLOT = 3;
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  T* lds = reinterpret_cast<T*>(lds_raw);
  long long* slot = reinterpret_cast<long long*>(reinterpret_cast<char*>(lds_raw) + NSLOT * SLOT_BYTES);
  const T* xdata = reinterpret_cast<const T*>(g.xdata);
  const T* ydata = reinterpret_cast<const T*>(g.ydata);
  T* Xre = reinterpret_cast<T*>(g.scratch) + (long long)blockIdx.x * 2 * (g.x_plane + g.t_plane);
  T* Xim = Xre + g.x_plane;
  T* Tre = Xim + g.x_plane;
  T* Tim = Tre + g.t_plane;
  const int tid = threadIdx.x;
  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p >= g.npairs) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    // per-site metadata of the pair, staged once: [xd | yd | xt | yt] (n+1 ints each) then [xo | yo] (n int64)
    const int n1 = g.n_sites + 1;
    int* m_xd = reinterpret_cast<int*>(slot + 2);
    int* m_yd = m_xd + n1;
    int* m_xt = m_yd + n1;
    int* m_yt = m_xt + n1;
    long long* m_xo = reinterpret_cast<long long*>(m_xd + 4 * n1 + (4 * n1 & 1));
    long long* m_yo = m_xo + g.n_sites;
    for (int e = tid; e < n1; e += 64 * NW) {
      m_xd[e] = g.xdims[(long long)xi * n1 + e];
      m_yd[e] = g.ydims[(long long)yj * n1 + e];
      m_xt[e] = g.xtrue[(long long)xi * n1 + e];
      m_yt[e] = g.ytrue[(long long)yj * n1 + e];
      if (e < g.n_sites) {
        m_xo[e] = g.xoffs[(long long)xi * g.n_sites + e];
        m_yo[e] = g.yoffs[(long long)yj * g.n_sites + e];
      }
    }
    __syncthreads();
    auto ldi = [&](const int* q_) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*q_); };
    auto ldl = [&](const long long* q_) __attribute__((always_inline)) {
      const long long v = *q_;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
      return (long long)(((unsigned long long)hi << 32) | lo);
    };
    {
      const int a = ldi(m_xd), b = ldi(m_yd);
      for (int e = tid; e < a * b; e += 64 * NW) {
        Xre[e] = (e == 0) ? (T)1 : (T)0;
        Xim[e] = (T)0;
      }
      __syncthreads();
    }
    for (int k = 0; k < g.n_sites; ++k) {
      const int a = ldi(m_xd + k), a2 = ldi(m_xd + k + 1), b = ldi(m_yd + k), b2 = ldi(m_yd + k + 1);
      const T* Are = xdata + ldl(m_xo + k);
      const T* Aim = Are + (long long)a * 2 * a2;
      const T* Bre = ydata + ldl(m_yo + k);
      const T* Bim = Bre + (long long)b * 2 * b2;
      zgemm_ring3<false, KTL, NSLOT, true, NW, 64>(Tre, Tim, 2 * b2, Xre, Xim, a, Bre, Bim, 2 * b2, a, 2 * b2, ldi(m_yt + k), lds);
      zgemm_ring3<true, KTL, NSLOT, true, NW, 64>(Xre, Xim, a2, Tre, Tim, b2, Are, Aim, a2, b2, a2, 2 * ldi(m_xt + k), lds);
    }
    if (tid == 0) {
      const double re = (double)Xre[0], im = (double)Xim[0];
      g.values[p] = re * re + im * im;
      if (g.z) {
        g.z[2 * p] = re;
        g.z[2 * p + 1] = im;
      }
    }
    __syncthreads();
  }
}

__global__ void qk_convert_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, const long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) dst[e] = (float)src[e];
}

// ----------------------------------------------------------------------------------------
// v4: group sweep.  One workgroup carries up to GMAX pairs that share the x state through the
// sweep in lockstep.  Per site:
//   phase 1  = a STREAM of cnt independent GEMMs  T_g[a x 2b'_g] = X_g^T B_g, whose two column halves
//              (physical index p) are written into one stacked matrix T_all[(L,p)][sum_g b'_g];
//   phase 2  = ONE GEMM  X'_all[sum_g b'_g x a'] = T_all^T conj(A_k)  (A_k read once per group).
// zgemm_stream runs the two-step-deep prefetch pipeline of zgemm_deep over a list of GEMM
// descriptors without draining between them, so the fixed per-phase latencies (prologue, barriers,
// store->load round trip) are paid once per GROUP-phase while the MFMA work grows with the group.
// ----------------------------------------------------------------------------------------
static constexpr int GMAX = 4;

struct GemmDesc {  // lives in LDS; planes: im = re + plane
  double* Cre;
  const double* Are;
  const double* Bre;
  long long c_plane, a_plane, b_plane;
  long long c_jump;  // output columns >= n_half land c_jump elements further (second physical index of T_all)
  int ldc, lda, ldb, M, N, Ktrue, n_half;
  int conjb;  // conjugate the B operand (the x-state tensor in X' = T^T conj(A))
};

__device__ __forceinline__ long long qk_uniform_ll(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int qk_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int PN, int KTL, int NW, int PMT>
__device__ __forceinline__ void zgemm_stream(const GemmDesc* __restrict__ descs, const int count, double* __restrict__ lds, const bool fence, const int dbg = 0) {
  using G = GemmCfg<PN, KTL, NW, PMT>;
  constexpr int PM = G::PM;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;

  // total number of (gemm, pass, K-tile) steps
  int total = 0;
  for (int g = 0; g < count; ++g) {
    const int M = qk_uniform_i(descs[g].M), N = qk_uniform_i(descs[g].N), K = qk_uniform_i(descs[g].Ktrue);
    total += ((M + PM - 1) / PM) * ((N + PN - 1) / PN) * ((K + KTL - 1) / KTL);
  }

  // ---- fetch-side iterator (runs two steps ahead)
  int f_g = 0, f_kt = 0, f_pm = 0, f_pn = 0;
  const double *fAre, *fAim, *fBre, *fBim;
  int f_lda, f_ldb, f_M, f_N, f_nk, f_npm, f_npn;
  double f_sgn = 1.0, sgn0 = 1.0, sgn1 = 1.0;
  bool f_new = false;  // the next fetch is the first tile of a GEMM other than the stream's first  // sign of the staged B imaginary plane (conjugation), per register set
  unsigned rowoffA[G::UA], rowoffB[G::UB];
  int colA[G::UA], colB[G::UB];
#pragma unroll
  for (int i = 0; i < G::UA; ++i) colA[i] = ((tid + G::WGT * i) % (PM / 2)) * 2;
#pragma unroll
  for (int i = 0; i < G::UB; ++i) colB[i] = ((tid + G::WGT * i) % (PN / 2)) * 2;
  auto load_fetch_desc = [&](int g) __attribute__((always_inline)) {
    const GemmDesc* d = descs + g;
    fAre = reinterpret_cast<const double*>(qk_uniform_ll(reinterpret_cast<long long>(d->Are)));
    fAim = fAre + qk_uniform_ll(d->a_plane);
    fBre = reinterpret_cast<const double*>(qk_uniform_ll(reinterpret_cast<long long>(d->Bre)));
    fBim = fBre + qk_uniform_ll(d->b_plane);
    f_lda = qk_uniform_i(d->lda), f_ldb = qk_uniform_i(d->ldb);
    f_M = qk_uniform_i(d->M), f_N = qk_uniform_i(d->N);
    f_sgn = qk_uniform_i(d->conjb) ? -1.0 : 1.0;
    f_nk = (qk_uniform_i(d->Ktrue) + KTL - 1) / KTL;
    f_npm = (f_M + PM - 1) / PM, f_npn = (f_N + PN - 1) / PN;
#pragma unroll
    for (int i = 0; i < G::UA; ++i) rowoffA[i] = (unsigned)(((tid + G::WGT * i) / (PM / 2)) * f_lda);
#pragma unroll
    for (int i = 0; i < G::UB; ++i) rowoffB[i] = (unsigned)(((tid + G::WGT * i) / (PN / 2)) * f_ldb);
  };
  load_fetch_desc(0);

  double2 ra0[2 * G::UA], rb0[2 * G::UB], ra1[2 * G::UA], rb1[2 * G::UB];
#pragma unroll
  for (int i = 0; i < 2 * G::UA; ++i) ra0[i] = ra1[i] = make_double2(0.0, 0.0);
#pragma unroll
  for (int i = 0; i < 2 * G::UB; ++i) rb0[i] = rb1[i] = make_double2(0.0, 0.0);

#define QK_FETCH_SET(RA, RB, SG)                                                  \
  do {                                                                            \
    const int m0_ = f_pm * PM, n0_ = f_pn * PN;                                   \
    const int mcols_ = min(PM, f_M - m0_), ncols_ = min(PN, f_N - n0_);           \
    if (fence && f_new) { /* the producer of this GEMM's input finished >= 1 GEMM ago: drain its stores now */ \
      __syncthreads();                                                            \
      f_new = false;                                                              \
    }                                                                             \
    SG = f_sgn;                                                                   \
    const long long ka_ = (long long)f_kt * KTL * f_lda + m0_;                    \
    const long long kb_ = (long long)f_kt * KTL * f_ldb + n0_;                    \
    const double* are_ = fAre + ka_;                                              \
    const double* aim_ = fAim + ka_;                                              \
    const double* bre_ = fBre + kb_;                                              \
    const double* bim_ = fBim + kb_;                                              \
    _Pragma("unroll") for (int i = 0; i < G::UA; ++i) {                           \
      const unsigned o = rowoffA[i] + (unsigned)min(colA[i], mcols_ - 2);         \
      RA[2 * i] = *reinterpret_cast<const double2*>(are_ + o);                    \
      RA[2 * i + 1] = *reinterpret_cast<const double2*>(aim_ + o);                \
    }                                                                             \
    _Pragma("unroll") for (int i = 0; i < G::UB; ++i) {                           \
      const unsigned o = rowoffB[i] + (unsigned)min(colB[i], ncols_ - 2);         \
      RB[2 * i] = *reinterpret_cast<const double2*>(bre_ + o);                    \
      RB[2 * i + 1] = *reinterpret_cast<const double2*>(bim_ + o);                \
    }                                                                             \
    if (++f_kt == f_nk) {                                                         \
      f_kt = 0;                                                                   \
      if (++f_pm == f_npm) {                                                      \
        f_pm = 0;                                                                 \
        if (++f_pn == f_npn) {                                                    \
          f_pn = 0;                                                               \
          if (++f_g < count) {                                                    \
            load_fetch_desc(f_g);                                                 \
            f_new = true;                                                         \
          }                                                                       \
        }                                                                         \
      }                                                                           \
    }                                                                             \
  } while (0)
#define QK_STASH_SET(BUF, RA, RB, SG)                                             \
  do {                                                                            \
    double* base_ = lds + (BUF)*G::STAGE_D;                                       \
    _Pragma("unroll") for (int i = 0; i < G::UA; ++i) {                           \
      const int u = tid + G::WGT * i;                                             \
      const int o = (u / (PM / 2)) * PM + (u % (PM / 2)) * 2;                     \
      *reinterpret_cast<double2*>(base_ + o) = RA[2 * i];                         \
      *reinterpret_cast<double2*>(base_ + G::A_PLANE + o) = RA[2 * i + 1];        \
    }                                                                             \
    _Pragma("unroll") for (int i = 0; i < G::UB; ++i) {                           \
      const int u = tid + G::WGT * i;                                             \
      const int o = (u / (PN / 2)) * PN + (u % (PN / 2)) * 2;                     \
      *reinterpret_cast<double2*>(base_ + 2 * G::A_PLANE + o) = RB[2 * i];        \
      *reinterpret_cast<double2*>(base_ + 2 * G::A_PLANE + G::B_PLANE + o) = make_double2(SG * RB[2 * i + 1].x, SG * RB[2 * i + 1].y); \
    }                                                                             \
  } while (0)

  // ---- compute-side iterator
  int c_g = 0, c_kt = 0, c_pm = 0, c_pn = 0;
  double *cCre, *cCim;
  int c_ldc, c_M, c_N, c_nk, c_k4, c_npm, c_npn, c_nhalf;
  long long c_jump;
  auto load_compute_desc = [&](int g) __attribute__((always_inline)) {
    const GemmDesc* d = descs + g;
    cCre = reinterpret_cast<double*>(qk_uniform_ll(reinterpret_cast<long long>(d->Cre)));
    cCim = cCre + qk_uniform_ll(d->c_plane);
    c_ldc = qk_uniform_i(d->ldc);
    c_nhalf = qk_uniform_i(d->n_half);
    c_jump = qk_uniform_ll(d->c_jump);
    c_M = qk_uniform_i(d->M), c_N = qk_uniform_i(d->N);
    const int K = qk_uniform_i(d->Ktrue);
    c_nk = (K + KTL - 1) / KTL, c_k4 = (K + 3) >> 2;
    c_npm = (c_M + PM - 1) / PM, c_npn = (c_N + PN - 1) / PN;
  };
  load_compute_desc(0);

  v4d cre[G::MAXT], cim[G::MAXT];
  int tm[G::MAXT], tn[G::MAXT];
  int cnt = 0;
  bool crossed = false;  // the step just computed was the last one of its GEMM
  auto compute_step = [&](int buf) __attribute__((always_inline)) {
    crossed = false;
    const int m0 = c_pm * PM, n0 = c_pn * PN;
    if (c_kt == 0) {
      const int mt = min(PM / TILE, (c_M - m0) / TILE);
      const int nt = min(PN / TILE, (c_N - n0) / TILE);
      const int vt = mt * nt;
      cnt = (vt > wave) ? (vt - wave + NW - 1) / NW : 0;
#pragma unroll
      for (int e = 0; e < G::MAXT; ++e) {
        cre[e] = (v4d){0, 0, 0, 0};
        cim[e] = (v4d){0, 0, 0, 0};
        const int t = min(wave + NW * e, vt - 1);
        tm[e] = t % mt;
        tn[e] = t / mt;
      }
    }
    const double* base = lds + buf * G::STAGE_D;
    const int ksteps = min(KTL / 4, c_k4 - c_kt * (KTL / 4));
    if (!(dbg & 4)) {
      if (ksteps == KTL / 4)
        mma_ktile<false, PM, PN, G::A_PLANE, G::B_PLANE, KTL / 4, G::MAXT, true, true>(cre, cim, tm, tn, base, q, j, cnt, ksteps);
      else
        mma_ktile<false, PM, PN, G::A_PLANE, G::B_PLANE, KTL / 4, G::MAXT, false, true>(cre, cim, tm, tn, base, q, j, cnt, ksteps);
    }
    if (c_kt == c_nk - 1 && !(dbg & 1)) {
#pragma unroll
      for (int e = 0; e < G::MAXT; ++e) {
        if (e < cnt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col0 = n0 + tn[e] * TILE;  // tiles never straddle n_half (a multiple of 16)
            const long long o = (long long)(m0 + tm[e] * TILE + q + 4 * r) * c_ldc + col0 + j + (col0 >= c_nhalf ? c_jump : 0);
            cCre[o] = cre[e][r];
            cCim[o] = cim[e][r];
          }
        }
      }
    }
    if (++c_kt == c_nk) {
      c_kt = 0;
      if (++c_pm == c_npm) {
        c_pm = 0;
        if (++c_pn == c_npn) {
          c_pn = 0;
          crossed = true;
          if (++c_g < count) load_compute_desc(c_g);
        }
      }
    }
  };

  // In an interleaved stream (fence = true) the consumer of a GEMM's output is the GEMM after the
  // next one.  The full fence that makes that output visible is taken on the FETCH side, right
  // before the consumer's first tile is requested -- by then the producer's stores have had a whole
  // GEMM to drain -- and never right after the producer's epilogue.
#define QK_STEP_BARRIER() qk_lds_barrier()
  QK_FETCH_SET(ra0, rb0, sgn0);
  if (total > 1) QK_FETCH_SET(ra1, rb1, sgn1);
  QK_STASH_SET(0, ra0, rb0, sgn0);
  qk_lds_barrier();
  int s = 0;
  while (s + 3 < total) {
    if (!(dbg & 2)) QK_FETCH_SET(ra0, rb0, sgn0);
    compute_step(0);
    if (!(dbg & 2)) QK_STASH_SET(1, ra1, rb1, sgn1);
    if (!(dbg & 8)) QK_STEP_BARRIER();
    if (!(dbg & 2)) QK_FETCH_SET(ra1, rb1, sgn1);
    compute_step(1);
    if (!(dbg & 2)) QK_STASH_SET(0, ra0, rb0, sgn0);
    if (!(dbg & 8)) QK_STEP_BARRIER();
    s += 2;
  }
  for (; s < total; ++s) {
    const bool even = (s & 1) == 0;
    if (s + 2 < total) {
      if (even) QK_FETCH_SET(ra0, rb0, sgn0); else QK_FETCH_SET(ra1, rb1, sgn1);
    }
    compute_step(s & 1);
    if (s + 1 < total) {
      if (even) QK_STASH_SET(1, ra1, rb1, sgn1); else QK_STASH_SET(0, ra0, rb0, sgn0);
    }
    QK_STEP_BARRIER();
  }
#undef QK_STEP_BARRIER
#undef QK_FETCH_SET
#undef QK_STASH_SET
  __syncthreads();
}

template <int PN, int KTL, int OCC, int NW, int PMT>
__global__ __launch_bounds__(64 * NW, OCC) void qk_sweep_group_kernel(const SweepArgs g) {
  using G = GemmCfg<PN, KTL, NW, PMT>;
  constexpr int T = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  long long* slot = reinterpret_cast<long long*>(lds + G::LDS_D);
  const int n = g.n_sites, n1 = n + 1;
  // LDS after the staging buffers: slot | descriptors | x meta | y meta (per group member)
  GemmDesc* desc = reinterpret_cast<GemmDesc*>(slot + 2);
  long long* m_xo = reinterpret_cast<long long*>(desc + GMAX + 1);
  long long* m_yo = m_xo + n;            // [GMAX][n]
  int* m_xd = reinterpret_cast<int*>(m_yo + GMAX * n);
  int* m_xt = m_xd + n1;
  int* m_yd = m_xt + n1;                 // [GMAX][n1]
  int* m_yt = m_yd + GMAX * n1;          // [GMAX][n1]

  double* Xre = g.scratch + (long long)blockIdx.x * 2 * (g.x_plane + g.t_plane);
  double* Xim = Xre + g.x_plane;
  double* Tre = Xim + g.x_plane;
  double* Tim = Tre + g.t_plane;
  const int tid = threadIdx.x;

  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long gi = *slot;
    __syncthreads();
    if (gi >= g.ngroups) break;
    const long long first = g.groups[2 * gi];
    const int cnt = g.groups[2 * gi + 1];
    const int xi = g.pairs[2 * first];
    for (int e = tid; e < n1; e += T) {
      m_xd[e] = g.xdims[(long long)xi * n1 + e];
      m_xt[e] = g.xtrue[(long long)xi * n1 + e];
      if (e < n) m_xo[e] = g.xoffs[(long long)xi * n + e];
    }
    for (int e = tid; e < cnt * n1; e += T) {
      const int gg = e / n1, k = e - gg * n1;
      const int yj = g.pairs[2 * (first + gg) + 1];
      m_yd[gg * n1 + k] = g.ydims[(long long)yj * n1 + k];
      m_yt[gg * n1 + k] = g.ytrue[(long long)yj * n1 + k];
      if (k < n) m_yo[gg * n + k] = g.yoffs[(long long)yj * n + k];
    }
    __syncthreads();
    // X_all at site 0: one 16x16 block per member, X[0][0] = 1
    {
      const int a = qk_uniform_i(m_xd[0]);
      int rows = 0;
      for (int gg = 0; gg < cnt; ++gg) rows += qk_uniform_i(m_yd[gg * n1]);
      for (int e = tid; e < rows * a; e += T) {
        const int r = e / a, c = e - r * a;
        Xre[e] = (c == 0 && (r % TILE) == 0) ? 1.0 : 0.0;  // every member starts from a 1x1 bond padded to 16
        Xim[e] = 0.0;
      }
      __syncthreads();
    }
    for (int k = 0; k < n; ++k) {
      if (tid == 0) {
        const int a = m_xd[k], a2 = m_xd[k + 1], at = m_xt[k];
        int SB2 = 0;
        for (int gg = 0; gg < cnt; ++gg) SB2 += m_yd[gg * n1 + k + 1];
        int rowoff = 0, coff = 0;
        for (int gg = 0; gg < cnt; ++gg) {
          const int b = m_yd[gg * n1 + k], b2 = m_yd[gg * n1 + k + 1], bt = m_yt[gg * n1 + k];
          {
            GemmDesc& d = desc[gg];
            d.Cre = Tre + coff;              // T_all[(L,p)][coff + r]: row (2L+p) of a [2a][SB2] matrix
            d.c_plane = Tim - Tre;
            d.ldc = 2 * SB2;                 // consecutive L are 2 rows of T_all apart
            d.n_half = b2;                   // columns n >= b2 belong to p = 1 ...
            d.c_jump = (long long)SB2 - b2;  // ... and start one T_all row further
            d.Are = Xre + (long long)rowoff * a;
            d.a_plane = Xim - Xre;
            d.lda = a;
            d.Bre = g.ydata + m_yo[gg * n + k];
            d.b_plane = (long long)b * 2 * b2;
            d.ldb = 2 * b2;
            d.M = a, d.N = 2 * b2, d.Ktrue = bt;
            d.conjb = 0;
          }
          rowoff += b, coff += b2;
        }
        GemmDesc& d = desc[cnt];
        d.n_half = a2, d.c_jump = 0, d.conjb = 1;
        d.Cre = Xre, d.c_plane = Xim - Xre, d.ldc = a2;
        d.Are = Tre, d.a_plane = Tim - Tre, d.lda = SB2;
        d.Bre = g.xdata + m_xo[k], d.b_plane = (long long)a * 2 * a2, d.ldb = a2;
        d.M = SB2, d.N = a2, d.Ktrue = 2 * at;
      }
      __syncthreads();
      zgemm_stream<PN, KTL, NW, PMT>(desc, cnt, lds, false, g.debug_flags);
      zgemm_stream<PN, KTL, NW, PMT>(desc + cnt, 1, lds, false, g.debug_flags);
    }
    if (tid < cnt) {
      // final environment of member `tid`: a 16x16 block at row offset sum of the earlier members' last bonds
      int rowoff = 0;
      for (int gg = 0; gg < tid; ++gg) rowoff += m_yd[gg * n1 + n];
      const long long o = (long long)rowoff * m_xd[n];
      const double re = Xre[o], im = Xim[o];
      g.values[first + tid] = re * re + im * im;
      if (g.z) {
        g.z[2 * (first + tid)] = re;
        g.z[2 * (first + tid) + 1] = im;
      }
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------
// v5: duo sweep.  One workgroup carries TWO independent pairs (chains A and B) through the sweep
// and interleaves their phases in one GEMM stream  [A.p1, B.p1, A.p2, B.p2]  per site.  A phase's
// consumer is the GEMM after the next one, so its store -> load round trip and the consumer's first
// tile fetch are hidden behind the other chain's GEMM instead of stalling the workgroup (the
// per-phase prologue + store drain measured ~6 us x 120 phases per pair on the single-chain kernel).
// Sites where some GEMM has fewer than two steps (chain ends) fall back to one GEMM at a time.
// ----------------------------------------------------------------------------------------
template <int PN, int KTL, int OCC, int NW, int PMT>
__global__ __launch_bounds__(64 * NW, OCC) void qk_sweep_duo_kernel(const SweepArgs g) {
  using G = GemmCfg<PN, KTL, NW, PMT>;
  constexpr int T = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  long long* slot = reinterpret_cast<long long*>(lds + G::LDS_D);
  const int n = g.n_sites, n1 = n + 1;
  GemmDesc* desc = reinterpret_cast<GemmDesc*>(slot + 2);            // [4]
  int* flags = reinterpret_cast<int*>(desc + 4);                    // [2]: interleave ok, pad
  long long* m_xo = reinterpret_cast<long long*>(flags + 2);        // [2][n]
  long long* m_yo = m_xo + 2 * n;                                   // [2][n]
  int* m_xd = reinterpret_cast<int*>(m_yo + 2 * n);                 // [2][n1] each below
  int* m_yd = m_xd + 2 * n1;
  int* m_xt = m_yd + 2 * n1;
  int* m_yt = m_xt + 2 * n1;

  const long long chain_stride = 2 * (g.x_plane + g.t_plane);
  double* base = g.scratch + (long long)blockIdx.x * 2 * chain_stride;
  const int tid = threadIdx.x;

  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long gi = *slot;
    __syncthreads();
    const long long p0 = 2 * gi;
    if (p0 >= g.npairs) break;
    const int nch = (p0 + 1 < g.npairs) ? 2 : 1;
    for (int e = tid; e < nch * n1; e += T) {
      const int c = e / n1, k = e - c * n1;
      const int xi = g.pairs[2 * (p0 + c)], yj = g.pairs[2 * (p0 + c) + 1];
      m_xd[c * n1 + k] = g.xdims[(long long)xi * n1 + k];
      m_yd[c * n1 + k] = g.ydims[(long long)yj * n1 + k];
      m_xt[c * n1 + k] = g.xtrue[(long long)xi * n1 + k];
      m_yt[c * n1 + k] = g.ytrue[(long long)yj * n1 + k];
      if (k < n) {
        m_xo[c * n + k] = g.xoffs[(long long)xi * n + k];
        m_yo[c * n + k] = g.yoffs[(long long)yj * n + k];
      }
    }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      double* Xre = base + c * chain_stride;
      double* Xim = Xre + g.x_plane;
      const int ab = qk_uniform_i(m_xd[c * n1]) * qk_uniform_i(m_yd[c * n1]);
      for (int e = tid; e < ab; e += T) {
        Xre[e] = (e == 0) ? 1.0 : 0.0;
        Xim[e] = 0.0;
      }
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      if (tid < nch) {
        const int c = tid;
        double* Xre = base + c * chain_stride;
        double* Xim = Xre + g.x_plane;
        double* Tre = Xim + g.x_plane;
        double* Tim = Tre + g.t_plane;
        const int a = m_xd[c * n1 + k], a2 = m_xd[c * n1 + k + 1], b = m_yd[c * n1 + k], b2 = m_yd[c * n1 + k + 1];
        GemmDesc& d1 = desc[c];          // phase 1: T[a x 2 b2] = X^T B
        d1.Cre = Tre, d1.c_plane = Tim - Tre, d1.ldc = 2 * b2, d1.n_half = 2 * b2, d1.c_jump = 0;
        d1.Are = Xre, d1.a_plane = Xim - Xre, d1.lda = a;
        d1.Bre = g.ydata + m_yo[c * n + k], d1.b_plane = (long long)b * 2 * b2, d1.ldb = 2 * b2;
        d1.M = a, d1.N = 2 * b2, d1.Ktrue = m_yt[c * n1 + k], d1.conjb = 0;
        GemmDesc& d2 = desc[nch + c];    // phase 2: X'[b2 x a2] = T^T conj(A)
        d2.Cre = Xre, d2.c_plane = Xim - Xre, d2.ldc = a2, d2.n_half = a2, d2.c_jump = 0;
        d2.Are = Tre, d2.a_plane = Tim - Tre, d2.lda = b2;
        d2.Bre = g.xdata + m_xo[c * n + k], d2.b_plane = (long long)a * 2 * a2, d2.ldb = a2;
        d2.M = b2, d2.N = a2, d2.Ktrue = 2 * m_xt[c * n1 + k], d2.conjb = 1;
      }
      __syncthreads();
      bool inter = (nch == 2);
      if (inter) {  // every GEMM of the interleaved stream needs at least two steps (see zgemm_stream)
        for (int i = 0; i < 4; ++i) {
          const int M = qk_uniform_i(desc[i].M), N = qk_uniform_i(desc[i].N), K = qk_uniform_i(desc[i].Ktrue);
          const int steps = ((M + G::PM - 1) / G::PM) * ((N + PN - 1) / PN) * ((K + KTL - 1) / KTL);
          inter = inter && steps >= 2;
        }
      }
      if (inter) {
        zgemm_stream<PN, KTL, NW, PMT>(desc, 4, lds, true, g.debug_flags);
      } else {
        for (int i = 0; i < 2 * nch; ++i) zgemm_stream<PN, KTL, NW, PMT>(desc + i, 1, lds, false, g.debug_flags);
      }
    }
    if (tid < nch) {
      const double* Xre = base + tid * chain_stride;
      const double re = Xre[0], im = Xre[g.x_plane];
      g.values[p0 + tid] = re * re + im * im;
      if (g.z) {
        g.z[2 * (p0 + tid)] = re;
        g.z[2 * (p0 + tid) + 1] = im;
      }
    }
    __syncthreads();
  }
}

template <int PN, int KTL, bool PROF = false, int NW = 4, int PMT = 64>
__global__ __launch_bounds__(64 * NW, 2) void qk_sweep_flat_kernel(const SweepArgs g) {
  using G = GemmCfg<PN, KTL, NW, PMT>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  long long* slot = reinterpret_cast<long long*>(lds + G::LDS_D);

  double* Xre = g.scratch + (long long)blockIdx.x * 2 * (g.x_plane + g.t_plane);
  double* Xim = Xre + g.x_plane;
  double* Tre = Xim + g.x_plane;
  double* Tim = Tre + g.t_plane;
  const int tid = threadIdx.x;
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = PROF ? qk_stamp() : 0;

  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p >= g.npairs) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    const int32_t* xd = g.xdims + (long long)xi * (g.n_sites + 1);
    const int32_t* yd = g.ydims + (long long)yj * (g.n_sites + 1);
    const int32_t* xt = g.xtrue + (long long)xi * (g.n_sites + 1);
    const int32_t* yt = g.ytrue + (long long)yj * (g.n_sites + 1);
    const int64_t* xo = g.xoffs + (long long)xi * g.n_sites;
    const int64_t* yo = g.yoffs + (long long)yj * g.n_sites;
    {
      const int a = xd[0], b = yd[0];
      for (int e = tid; e < a * b; e += 64 * NW) {
        Xre[e] = (e == 0) ? 1.0 : 0.0;
        Xim[e] = 0.0;
      }
      __syncthreads();
    }
    for (int k = 0; k < g.n_sites; ++k) {
      const int a = xd[k], a2 = xd[k + 1], b = yd[k], b2 = yd[k + 1];
      const double* Are = g.xdata + xo[k];
      const double* Aim = Are + (long long)a * 2 * a2;
      const double* Bre = g.ydata + yo[k];
      const double* Bim = Bre + (long long)b * 2 * b2;
      // phase 1: T[a x 2b2] = X^T B, contraction over the TRUE bond b_k of y
      zgemm_flat<false, PN, KTL, PROF, NW, PMT>(Tre, Tim, 2 * b2, Xre, Xim, a, Bre, Bim, 2 * b2, a, 2 * b2, yt[k], lds, pc);
      // phase 2: X'[b2 x a2] = T^T conj(A), contraction over the 2 * a_k true rows (L, p)
      zgemm_flat<true, PN, KTL, PROF, NW, PMT>(Xre, Xim, a2, Tre, Tim, b2, Are, Aim, a2, b2, a2, 2 * xt[k], lds, pc);
    }
    if (tid == 0) {
      const double re = Xre[0], im = Xim[0];
      g.values[p] = re * re + im * im;
      if (g.z) {
        g.z[2 * p] = re;
        g.z[2 * p + 1] = im;
      }
    }
    __syncthreads();
  }
  if (PROF && g.prof && (tid & 63) == 0) {
    pc[7] = qk_stamp() - t_begin;  // wave lifetime
#pragma unroll
    for (int c = 0; c < 8; ++c) atomicAdd(g.prof + c, (unsigned long long)pc[c]);
  }
}

// Diagnostic micro-kernel: the MFMA block alone (LDS fragments -> MFMAs), then with the other
// per-step ingredients of the sweep added back one at a time (FLAGS bit 0: workgroup barrier per
// step, bit 1: LDS stash of a staged tile, bit 2: global fetch of the next tile from an L2-resident
// buffer, bit 3: two-step-deep fetch like zgemm_deep).  Measures what each ingredient costs.
template <int NW, int FLAGS>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 4 : 2)) void qk_mma_bench_kernel(int reps, const double* __restrict__ src, double* out, double* cbuf, int epi_every) {
  using G = GemmCfg<64, 16, NW, 64>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  for (int e = tid; e < G::LDS_D; e += 64 * NW) lds[e] = 1e-3 * (double)((e * 7 + 3) % 11);
  __syncthreads();
  v4d cre[G::MAXT], cim[G::MAXT];
  int tm[G::MAXT], tn[G::MAXT];
#pragma unroll
  for (int e = 0; e < G::MAXT; ++e) {
    cre[e] = (v4d){0, 0, 0, 0};
    cim[e] = (v4d){0, 0, 0, 0};
    const int t = wave + NW * e;
    tm[e] = t % 4;
    tn[e] = t / 4;
  }
  constexpr int U = G::UA + G::UB;  // 16-byte units per thread per plane pair
  double2 r0[2 * U], r1[2 * U];
#pragma unroll
  for (int i = 0; i < 2 * U; ++i) r0[i] = r1[i] = make_double2(1e-3, 2e-3);
  // FLAGS bit 4: stream unique data from a large HBM-resident buffer (one 32 KiB tile per step and
  // workgroup, wrapping inside a 256 MiB-per-64-workgroups region) instead of an L2-resident window
  const bool big = (FLAGS & 16) != 0;
  const double* base_src = big ? src + (size_t)(blockIdx.x % 512) * (size_t)(1 << 18) : src + (size_t)(blockIdx.x % 64) * 8192;
  auto fetch = [&](double2 (&r)[2 * U], int step) __attribute__((always_inline)) {
    const size_t tile = big ? (size_t)(step & 63) * 4096 : (size_t)((step & 3) * 2048);
#pragma unroll
    for (int i = 0; i < 2 * U; ++i) r[i] = *reinterpret_cast<const double2*>(base_src + tile + (size_t)((i * 64 * NW + tid) * 2) % (big ? 4096 : 8192));
  };
  auto stash = [&](const double2 (&r)[2 * U], int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2 * U; ++i) *reinterpret_cast<double2*>(lds + buf * G::STAGE_D + (i * 64 * NW + tid) * 2) = r[i];
  };
  if (FLAGS & 4) {
    fetch(r0, 0);
    if (FLAGS & 8) fetch(r1, 1);
  }
  for (int r = 0; r < reps; r += 2) {
    // even step
    if ((FLAGS & 4) && (FLAGS & 8)) fetch(r0, r + 2);
    mma_ktile<false, 64, 64, G::A_PLANE, G::B_PLANE, 4, G::MAXT, true, true>(cre, cim, tm, tn, lds, q, j, G::MAXT, 4);
    if (FLAGS & 2) stash((FLAGS & 8) ? r1 : r0, 1);
    if ((FLAGS & 4) && !(FLAGS & 8)) fetch(r0, r + 1);
    if (FLAGS & 1) qk_lds_barrier();
    if ((FLAGS & 32) && ((r / 2) % epi_every) == epi_every - 1) {  // FLAGS bit 5: the sweep's per-pass epilogue
      double* cw = cbuf + (size_t)blockIdx.x * 2 * 64 * 64;       // one 64x64 complex block per workgroup
#pragma unroll
      for (int e = 0; e < G::MAXT; ++e) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const size_t o = (size_t)(tm[e] * TILE + q + 4 * rr) * 64 + tn[e] * TILE + j;
          cw[o] = cre[e][rr];
          cw[64 * 64 + o] = cim[e][rr];
        }
        cre[e] = (v4d){0, 0, 0, 0};
        cim[e] = (v4d){0, 0, 0, 0};
      }
    }
    // odd step
    if ((FLAGS & 4) && (FLAGS & 8)) fetch(r1, r + 3);
    mma_ktile<false, 64, 64, G::A_PLANE, G::B_PLANE, 4, G::MAXT, true, true>(cre, cim, tm, tn, lds + G::STAGE_D, q, j, G::MAXT, 4);
    if (FLAGS & 2) stash(r0, 0);
    if ((FLAGS & 4) && !(FLAGS & 8)) fetch(r0, r + 2);
    if (FLAGS & 1) qk_lds_barrier();
  }
  double acc = 0;
#pragma unroll
  for (int e = 0; e < G::MAXT; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc += cre[e][r] + cim[e][r];
  out[(size_t)blockIdx.x * 64 * NW + tid] = acc + r0[0].x + r1[0].x;
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
  c->device = device_id;
  c->num_cus = prop.multiProcessorCount;
  HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  HIP_TRY(hipMalloc(&c->counter, sizeof(unsigned long long)));
  HIP_TRY(hipMalloc(&c->prof, 8 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(c->prof, 0, 8 * sizeof(unsigned long long)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_flat_kernel<64, 16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_flat_kernel<64, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_deep_kernel<64, 16, 2, 4, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_deep_kernel<64, 16, 4, 8, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_deep_kernel<64, 16, 4, 8, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_group_kernel<64, 16, 4, 8, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_duo_kernel<64, 16, 4, 8, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_ring_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 23>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 31>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<4, 63>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 23>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 31>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_mma_bench_kernel<8, 63>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16, 4, 64>::LDS_B));
  if (const char* v = std::getenv("QK_VARIANT")) c->variant = std::atoi(v);
  if (const char* v = std::getenv("QK_WGS_PER_CU")) c->wgs_per_cu = std::max(1, std::min(4, std::atoi(v)));
  *out = c;
  return QK_OK;
}

extern "C" int qk_ctx_destroy(qk_ctx* c) {
  if (!c) return QK_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->counter) (void)hipFree(c->counter);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
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
  HIP_TRY(hipMalloc(&m->d_dims, pad.size() * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&m->d_offs, offs.size() * sizeof(int64_t)));
  HIP_TRY(hipMemcpy(m->d_dims, pad.data(), pad.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&m->d_true, pad.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(m->d_true, bond_dims, pad.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->d_offs, offs.data(), offs.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  *out = m;
  return QK_OK;
}

extern "C" int qk_mps_set_destroy(qk_mps_set* m) {
  if (!m) return QK_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->d_data) (void)hipFree(m->d_data);
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
  if (device_bytes) *device_bytes = m->bytes;
  return QK_OK;
}

extern "C" int qk_mps_set_precision(const qk_mps_set* m) { return m ? m->precision : 0; }

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
  HIP_TRY(hipSetDevice(c->device));
  const long long np = (long long)plan->pairs.size() / 2;
  c->last = plan->stats;
  c->last.max_bond = std::max(xs->max_pad, ys->max_pad);
  c->last.kernel_ms = 0;
  c->last.grid = 0;
  if (np == 0) return QK_OK;
  int rc = ensure_plan_uploaded(c, plan);
  if (rc != QK_OK) return rc;

  if (xs->precision != ys->precision) return fail(QK_EINVAL, "qk_gram_values: the two sets differ in precision (fp%d, fp%d)", xs->precision, ys->precision);
  const bool f32 = (xs->precision == 32);
  if (f32) c->last.bytes *= 0.5;  // complex64 planes
  const bool grouped = (c->variant == 14) && !f32;
  const bool duo = (c->variant == 16) && !f32;
  const long long members = grouped ? GMAX : 1;  // pairs stacked in one X/T buffer
  const long long chains = duo ? 2 : 1;          // independent X/T buffer sets per workgroup
  const long long x_plane = members * xs->max_pad * ys->max_pad;
  const long long t_plane = 2 * x_plane;
  const long long units = grouped ? (long long)plan->groups.size() / 2 : (duo ? (np + 1) / 2 : np);
  const int grid = (int)std::min<long long>(units, (long long)c->wgs_per_cu * c->num_cus);
  const size_t need = (size_t)grid * (size_t)chains * 2 * (size_t)(x_plane + t_plane) * sizeof(double);
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
  a.counter = c->counter;
  a.prof = c->prof;
  a.debug_flags = 0;
  if (const char* v = std::getenv("QK_DEBUG_FLAGS")) a.debug_flags = std::atoi(v);
  a.prio_mode = 0;
  if (const char* v = std::getenv("QK_PRIO")) a.prio_mode = std::atoi(v);
  HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  constexpr size_t lds_b = GemmCfg<64, 16>::LDS_B;
  // deep kernels also keep the pair's per-site metadata in LDS: 4 (n+1) ints + 2 n int64 (+ alignment)
  const size_t lds_deep = lds_b + 16 + (size_t)(4 * (xs->n_sites + 1) + 2) * sizeof(int) + (size_t)2 * xs->n_sites * sizeof(long long);
  if (lds_deep > 80 * 1024) return fail(QK_EINVAL, "qk_gram_values: %d sites need %zu bytes of LDS per workgroup (limit 80 KiB for 2 workgroups per CU)", xs->n_sites, lds_deep);
  if (f32) {  // complex64 sweep: the ring kernel on fp32 planes; QK_VARIANT does not apply
    qk_sweep_ring_kernel<float><<<dim3(grid), dim3(512), lds_deep - 16 * 1024, c->stream>>>(a);
  } else
  switch (c->variant) {
    case 0:  // v1: per-pass pipeline, 4 waves
      qk_sweep_kernel<<<dim3(grid), dim3(WG_THREADS), LDS_BYTES, c->stream>>>(a);
      break;
    case 2:  // flat pipeline, 4 waves, one-step prefetch
      qk_sweep_flat_kernel<64, 16, false><<<dim3(grid), dim3(WG_THREADS), lds_b, c->stream>>>(a);
      break;
    case 9:  // diagnostic: instrumented flat pipeline (qk_debug_profile)
      HIP_TRY(hipMemsetAsync(c->prof, 0, 8 * sizeof(unsigned long long), c->stream));
      qk_sweep_flat_kernel<64, 16, true><<<dim3(grid), dim3(WG_THREADS), lds_b, c->stream>>>(a);
      break;
    case 14: {  // group sweep: up to GMAX pairs sharing the x state per workgroup
      const int ns = xs->n_sites;
      const size_t lds_group = lds_b + 16 + (GMAX + 1) * sizeof(GemmDesc) + (size_t)(1 + GMAX) * ns * sizeof(long long) + (size_t)(2 + 2 * GMAX) * (ns + 1) * sizeof(int);
      if (lds_group > 80 * 1024) return fail(QK_EINVAL, "qk_gram_values: %d sites need %zu bytes of LDS per workgroup", ns, lds_group);
      qk_sweep_group_kernel<64, 16, 4, 8, 64><<<dim3(grid), dim3(512), lds_group, c->stream>>>(a);
      break;
    }
    case 16: {  // duo sweep: two independent pairs per workgroup, phases interleaved in one stream
      const int ns = xs->n_sites;
      const size_t lds_duo = lds_b + 16 + 4 * sizeof(GemmDesc) + 8 + (size_t)4 * ns * sizeof(long long) + (size_t)8 * (ns + 1) * sizeof(int);
      if (lds_duo > 80 * 1024) return fail(QK_EINVAL, "qk_gram_values: %d sites need %zu bytes of LDS per workgroup", ns, lds_duo);
      qk_sweep_duo_kernel<64, 16, 4, 8, 64><<<dim3(grid), dim3(512), lds_duo, c->stream>>>(a);
      break;
    }
    case 19:  // diagnostic: instrumented shipped kernel
      HIP_TRY(hipMemsetAsync(c->prof, 0, 8 * sizeof(unsigned long long), c->stream));
      qk_sweep_deep_kernel<64, 16, 4, 8, 64, true><<<dim3(grid), dim3(512), lds_deep, c->stream>>>(a);
      break;
    case 12:  // two-step-deep prefetch, 4 waves
      qk_sweep_deep_kernel<64, 16, 2, 4, 64><<<dim3(grid), dim3(256), lds_deep, c->stream>>>(a);
      break;
    case 13:  // two-step-deep prefetch, 8 waves (2 tiles per wave, 16 waves per CU)
      qk_sweep_deep_kernel<64, 16, 4, 8, 64><<<dim3(grid), dim3(512), lds_deep, c->stream>>>(a);
      break;
    case 17:  // lean register-staged sweep (two-step-deep register prefetch, four-product complex MFMA)
      qk_sweep_lean_kernel<4><<<dim3(grid), dim3(512), lds_deep, c->stream>>>(a);
      break;
    case 21:  // ring kernel with four slots (three K-tiles in flight)
      qk_sweep_lean_kernel<4, 2><<<dim3(grid), dim3(512), lds_deep, c->stream>>>(a);
      break;
    case 24:  // ring kernel on 4-wave workgroups (64x32 pass), four workgroups per CU (QK_WGS_PER_CU=4)
      qk_sweep_lean_kernel<4, 5><<<dim3(grid), dim3(256), lds_deep - 28 * 1024, c->stream>>>(a);
      break;
    case 23:  // ring kernel, K-tile 16, two slots
      qk_sweep_lean_kernel<4, 4><<<dim3(grid), dim3(512), lds_deep, c->stream>>>(a);
      break;
    default:  // 20: the shipped kernel -- ring sweep: LDS-DMA staging ring (K-tile 8, three slots) + 3M complex product
      qk_sweep_lean_kernel<4, 1><<<dim3(grid), dim3(512), lds_deep - 16 * 1024, c->stream>>>(a);
      break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  c->ev_pending = true;
  c->last.grid = grid;
  return QK_OK;
}

extern "C" int qk_gram_values_host(qk_ctx* c, const qk_mps_set* xs, const qk_mps_set* ys, const qk_plan* plan, double* values_host, double* z_host) {
  if (!c || !plan || !values_host) return fail(QK_EINVAL, "qk_gram_values_host: null argument");
  const int64_t np = qk_plan_num_pairs(plan);
  if (np == 0) return QK_OK;
  HIP_TRY(hipSetDevice(c->device));
  double *d_vals = nullptr, *d_z = nullptr;
  HIP_TRY(hipMalloc(&d_vals, (size_t)np * sizeof(double)));
  if (z_host) HIP_TRY(hipMalloc(&d_z, (size_t)np * 2 * sizeof(double)));
  int rc = qk_gram_values(c, xs, ys, plan, d_vals, d_z);
  if (rc == QK_OK) {
    hipError_t e = hipMemcpyAsync(values_host, d_vals, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && z_host) e = hipMemcpyAsync(z_host, d_z, (size_t)np * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(QK_EDEVICE, "qk_gram_values_host: copy back failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_vals);
  if (d_z) (void)hipFree(d_z);
  return rc;
}

extern "C" int qk_scatter(qk_ctx* c, const int32_t* pairs_dev, const double* values_dev, int64_t n, double* k_dev, int64_t ld, int32_t mirror) {
  if (!c || !pairs_dev || !values_dev || !k_dev) return fail(QK_EINVAL, "qk_scatter: null argument");
  if (n <= 0) return QK_OK;
  HIP_TRY(hipSetDevice(c->device));
  const int bs = 256;
  hipLaunchKernelGGL(qk_scatter_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, c->stream, pairs_dev, values_dev, (long long)n, k_dev, (long long)ld, (int)mirror);
  HIP_TRY(hipGetLastError());
  return QK_OK;
}

extern "C" int qk_get_stats(qk_ctx* c, qk_stats* out) {
  if (!c || !out) return fail(QK_EINVAL, "qk_get_stats: null argument");
  if (c->ev_pending) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last.kernel_ms = ms;
    c->ev_pending = false;
  }
  *out = c->last;
  return QK_OK;
}

// dims table of a set as the planner wants it
static int plan_for_sets(const qk_mps_set* xs, const qk_mps_set* ys, qk_plan** plan) {
  const bool sym = (ys == nullptr || ys == xs);
  return qk_plan_create(xs->n_sites, xs->n_states, xs->dims_true.data(), sym ? xs->n_states : ys->n_states,
                        sym ? nullptr : ys->dims_true.data(), sym ? QK_PLAN_SYMMETRIC : 0u, 1, 0, 0, plan);
}

extern "C" int qk_gram_host(qk_ctx* c, const qk_mps_set* xs, const qk_mps_set* ys, double* out, int64_t ld) {
  if (!c || !xs || !out) return fail(QK_EINVAL, "qk_gram_host: null argument");
  const bool sym = (ys == nullptr || ys == xs);
  const int nx = xs->n_states, ny = sym ? nx : ys->n_states;
  if (ld < nx) return fail(QK_EINVAL, "qk_gram_host: ld %lld < %d columns", (long long)ld, nx);
  qk_plan* plan = nullptr;
  int rc = plan_for_sets(xs, ys, &plan);
  if (rc != QK_OK) return rc;
  const int64_t np = qk_plan_num_pairs(plan);
  double *d_vals = nullptr, *d_k = nullptr;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc(&d_vals, (size_t)np * sizeof(double)));
  HIP_TRY(hipMalloc(&d_k, (size_t)ny * nx * sizeof(double)));
  HIP_TRY(hipMemsetAsync(d_k, 0, (size_t)ny * nx * sizeof(double), c->stream));
  rc = qk_gram_values(c, xs, ys, plan, d_vals, nullptr);
  if (rc == QK_OK) rc = qk_scatter(c, plan->d_pairs, d_vals, np, d_k, nx, sym ? 1 : 0);
  if (rc == QK_OK) {
    hipError_t e = hipMemcpy2DAsync(out, (size_t)ld * sizeof(double), d_k, (size_t)nx * sizeof(double), (size_t)nx * sizeof(double), (size_t)ny, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(QK_EDEVICE, "qk_gram_host: copy back failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_vals);
  (void)hipFree(d_k);
  qk_plan_destroy(plan);
  return rc;
}

extern "C" int qk_overlaps_host(qk_ctx* c, const qk_mps_set* xs, const qk_mps_set* ys, double* out) {
  if (!c || !xs || !out) return fail(QK_EINVAL, "qk_overlaps_host: null argument");
  if (!ys) ys = xs;
  const int nx = xs->n_states, ny = ys->n_states;
  qk_plan* plan = nullptr;  // all ny*nx pairs (no symmetry: z[i][j] = conj z[j][i] is left to the caller)
  int rc = qk_plan_create(xs->n_sites, nx, xs->dims_true.data(), ny, ys->dims_true.data(), 0u, 1, 0, 16, &plan);
  if (rc != QK_OK) return rc;
  const int64_t np = qk_plan_num_pairs(plan);
  double *d_vals = nullptr, *d_z = nullptr;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc(&d_vals, (size_t)np * sizeof(double)));
  HIP_TRY(hipMalloc(&d_z, (size_t)np * 2 * sizeof(double)));
  rc = qk_gram_values(c, xs, ys, plan, d_vals, d_z);
  std::vector<double> z((size_t)np * 2);
  if (rc == QK_OK) {
    hipError_t e = hipMemcpyAsync(z.data(), d_z, z.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(QK_EDEVICE, "qk_overlaps_host: copy back failed: %s", hipGetErrorString(e));
  }
  if (rc == QK_OK) {
    const int32_t* pr = qk_plan_pairs(plan);
    for (int64_t t = 0; t < np; ++t) {
      const int64_t o = ((int64_t)pr[2 * t + 1] * nx + pr[2 * t]) * 2;
      out[o] = z[2 * t], out[o + 1] = z[2 * t + 1];
    }
  }
  (void)hipFree(d_vals);
  (void)hipFree(d_z);
  qk_plan_destroy(plan);
  return rc;
}

extern "C" int qk_debug_profile(qk_ctx* c, unsigned long long* out8) {
  if (!c || !out8) return fail(QK_EINVAL, "qk_debug_profile: null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(out8, c->prof, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return QK_OK;
}

// Diagnostic: TFLOP/s of the MFMA block with per-step ingredients added back (see qk_mma_bench_kernel).
// which = 8 * (waves == 8) + flags-index, flags-index in {0: bare, 1: +barrier, 2: +barrier+stash, 3: +barrier+stash+fetch, 4: + deep fetch}
template <int NW>
static int run_mma_bench(qk_ctx* c, int fi, int grid, int reps, const double* src, double* out, size_t lds, double* cbuf, int epi) {
  switch (fi) {
    case 0: qk_mma_bench_kernel<NW, 0><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
    case 1: qk_mma_bench_kernel<NW, 1><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
    case 2: qk_mma_bench_kernel<NW, 3><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
    case 3: qk_mma_bench_kernel<NW, 7><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
    case 4: qk_mma_bench_kernel<NW, 15><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
    case 5: qk_mma_bench_kernel<NW, 23><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
    case 6: qk_mma_bench_kernel<NW, 31><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
    default: qk_mma_bench_kernel<NW, 63><<<dim3(grid), dim3(64 * NW), lds, c->stream>>>(reps, src, out, cbuf, epi); break;
  }
  return 0;
}

extern "C" int qk_debug_mma_bench(qk_ctx* c, int which, int wgs_per_cu, int reps, double* tflops) {
  if (!c || !tflops) return fail(QK_EINVAL, "qk_debug_mma_bench: null argument");
  HIP_TRY(hipSetDevice(c->device));
  const int nw = (which & 8) ? 8 : 4;
  const int fi = which & 7;
  const int epi = std::max(1, which >> 4);  // epilogue every `epi` step pairs (which = 16*epi + 8*(8 waves) + flags index)
  const int grid = c->num_cus * wgs_per_cu;
  double *out = nullptr, *src = nullptr;
  HIP_TRY(hipMalloc(&out, (size_t)grid * 64 * nw * sizeof(double)));
  const size_t src_bytes = (size_t)512 * (1 << 18) * sizeof(double) + 65536;  // 1 GiB: 2 MiB per workgroup slot
  HIP_TRY(hipMalloc(&src, src_bytes));
  HIP_TRY(hipMemset(src, 0, src_bytes));
  double* cbuf = nullptr;
  HIP_TRY(hipMalloc(&cbuf, (size_t)grid * 2 * 64 * 64 * sizeof(double)));
  const size_t lds = GemmCfg<64, 16, 4, 64>::LDS_B;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) {
    HIP_TRY(hipEventRecord(e0, c->stream));
    if (nw == 8) run_mma_bench<8>(c, fi, grid, reps, src, out, lds, cbuf, epi);
    else run_mma_bench<4>(c, fi, grid, reps, src, out, lds, cbuf, epi);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
  }
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *tflops = (double)grid * reps * 16.0 * 4 * 4 * 2048 / (ms * 1e-3) / 1e12;  // 16 tiles x 4 k-steps x 4 MFMAs x 2048 flop per step
  (void)hipFree(out), (void)hipFree(src), (void)hipFree(cbuf);
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
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
  double *dp, *dq, *dc;
  HIP_TRY(hipMalloc(&dp, sizeof hp));
  HIP_TRY(hipMalloc(&dq, sizeof hq));
  HIP_TRY(hipMalloc(&dc, sizeof hc));
  HIP_TRY(hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dq, hq, sizeof hq, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qk_selftest_kernel, dim3(1), dim3(64), 0, c->stream, dp, dq, dc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost));
  (void)hipFree(dp), (void)hipFree(dq), (void)hipFree(dc);
  double worst = 0;
  for (int e = 0; e < 256; ++e) worst = std::max(worst, std::fabs(hc[e] - ref[e]));
  if (worst > 1e-9) return fail(QK_EDEVICE, "qk_selftest_mfma: f64 MFMA fragment map mismatch (max abs error %.3g)", worst);
  // the same product through v_mfma_f32_16x16x4_f32 (operands are exact in fp32; sums of 16 such products too)
  float fp[256], fq[256], fc[256];
  for (int e = 0; e < 256; ++e) fp[e] = (float)hp[e], fq[e] = (float)hq[e];
  float *ep, *eq, *ec;
  HIP_TRY(hipMalloc(&ep, sizeof fp));
  HIP_TRY(hipMalloc(&eq, sizeof fq));
  HIP_TRY(hipMalloc(&ec, sizeof fc));
  HIP_TRY(hipMemcpy(ep, fp, sizeof fp, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(eq, fq, sizeof fq, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qk_selftest_f32_kernel, dim3(1), dim3(64), 0, c->stream, ep, eq, ec);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(fc, ec, sizeof fc, hipMemcpyDeviceToHost));
  (void)hipFree(ep), (void)hipFree(eq), (void)hipFree(ec);
  worst = 0;
  for (int e = 0; e < 256; ++e) worst = std::max(worst, std::fabs((double)fc[e] - ref[e]));
  if (worst > 1e-3) return fail(QK_EDEVICE, "qk_selftest_mfma: f32 MFMA fragment map mismatch (max abs error %.3g)", worst);
  return QK_OK;
}
