// qk_lab.hip -- experimental and diagnostic kernels of the Gram engine, kept selectable (QK_VARIANT) because
// DESIGN.md's investigation quotes measurements of each of them.  Nothing here is on the shipped path and nothing here
// is in libqkgram.so: this file is linked only into lab/libqklab.so (built with -DQK_LAB, loaded by lab/tools), which exports
// the same C ABI plus the entry points of qk_lab.h.
#include "qk_host.h"
#include "qk_lab.h"
#include "qk_ring.h"

#include <algorithm>
#include <cstdlib>

#define fail qk_fail

// ----------------------------------------------------------------------------------------
// device code
// ----------------------------------------------------------------------------------------
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

// the deep kernel's pair loop around the lean GEMM
// MODE 0: lean GEMM (register staging).  MODE 2: ring GEMM, K-tile 8, 4 slots, 3M product (3 slots = the shipped qk_sweep_ring_kernel<double>).
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
// v4: group sweep.  One workgroup carries up to GMAX pairs that share the x state through the
// sweep in lockstep.  Per site:
//   phase 1  = a STREAM of cnt independent GEMMs  T_g[a x 2b'_g] = X_g^T B_g, whose two column halves
//              (physical index p) are written into one stacked matrix T_all[(L,p)][sum_g b'_g];
//   phase 2  = ONE GEMM  X'_all[sum_g b'_g x a'] = T_all^T conj(A_k)  (A_k read once per group).
// zgemm_stream runs the two-step-deep prefetch pipeline of zgemm_deep over a list of GEMM
// descriptors without draining between them, so the fixed per-phase latencies (prologue, barriers,
// store->load round trip) are paid once per GROUP-phase while the MFMA work grows with the group.
// ----------------------------------------------------------------------------------------

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


// ----------------------------------------------------------------------------------------
// Quad sweep: one workgroup carries the 2x2 block of pairs {x1, x2} x {y1, y2} through the chain in lockstep.
// Per y_j the two environments sit side by side in one k-major matrix  XB_j = [X_1j | X_2j]  ([b_j][a_1 + a_2]), so
//   phase 1:  TB_j [(a_1 + a_2) x 2b'_j] = XB_j^T . B_kj           ONE GEMM per y_j: the site tensor of y_j is read
//                                                                   once for two pairs and M is doubled;
//   phase 2:  X'_ij [b'_j x a'_i] = T_ij^T . conj(A_ki)            per (i, j), T_ij = the a_i rows of TB_j, written into
//                                                                   XB'_j at column offset (i = 2 ? a'_1 : 0);  the two
//                                                                   reads of A_ki follow each other (second one from L2).
// Six GEMM calls per site for four pairs instead of eight, the same ring GEMM.  MEASURED (cfg4): 539 vs 526 ms for the
// pair kernel, fabric reads 2.16 vs 2.22 TB -- the stacked GEMM has two M passes and re-reads the B panel for each, so
// the shared site tensor is not read less; kept as an option (QK_PLAN_QUADS plans) because it is correct and tested.  pairs[4q .. 4q+3] = (x1,y1), (x2,y1), (x1,y2), (x2,y2); a duo may name the same state twice (odd set
// sizes): the duplicate is computed redundantly.  Scratch per workgroup: 2 x (XB_j planes + TB_j planes).
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512, 4) void qk_sweep_quad_kernel(const SweepArgs g) {
  constexpr int NW = 8;
  constexpr int KTL = 32 / (int)sizeof(T) * 2;
  constexpr int SLOT_BYTES = 16 * 1024, NSLOT = 3;
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  T* lds = reinterpret_cast<T*>(lds_raw);
  long long* slot = reinterpret_cast<long long*>(reinterpret_cast<char*>(lds_raw) + NSLOT * SLOT_BYTES);
  const T* xdata = reinterpret_cast<const T*>(g.xdata);
  const T* ydata = reinterpret_cast<const T*>(g.ydata);
  const long long xp = 2 * g.x_plane, tp = 2 * g.t_plane;  // planes of the stacked buffers
  T* const base = reinterpret_cast<T*>(g.scratch) + (long long)blockIdx.x * 4 * (xp + tp);
  const long long jstride = 2 * (xp + tp);  // XB_j = base + j * jstride
  const int tid = threadIdx.x;
  const long long nquads = g.npairs / 4;
  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long qd = *slot;
    __syncthreads();
    if (qd >= nquads) break;
    const int x1 = g.pairs[8 * qd], y1 = g.pairs[8 * qd + 1], x2 = g.pairs[8 * qd + 2], y2 = g.pairs[8 * qd + 5];
    // per-site metadata of the four states: [xd1 | xd2 | yd1 | yd2 | xt1 | xt2 | yt1 | yt2] (n+1 ints each), then
    // [xo1 | xo2 | yo1 | yo2] (n int64 each)
    const int n1 = g.n_sites + 1, ns = g.n_sites;
    int* md = reinterpret_cast<int*>(slot + 2);
    long long* mo = reinterpret_cast<long long*>(md + 8 * n1 + (8 * n1 & 1));
    for (int e = tid; e < n1; e += 64 * NW) {
      md[e] = g.xdims[(long long)x1 * n1 + e];
      md[n1 + e] = g.xdims[(long long)x2 * n1 + e];
      md[2 * n1 + e] = g.ydims[(long long)y1 * n1 + e];
      md[3 * n1 + e] = g.ydims[(long long)y2 * n1 + e];
      md[4 * n1 + e] = g.xtrue[(long long)x1 * n1 + e];
      md[5 * n1 + e] = g.xtrue[(long long)x2 * n1 + e];
      md[6 * n1 + e] = g.ytrue[(long long)y1 * n1 + e];
      md[7 * n1 + e] = g.ytrue[(long long)y2 * n1 + e];
      if (e < ns) {
        mo[e] = g.xoffs[(long long)x1 * ns + e];
        mo[ns + e] = g.xoffs[(long long)x2 * ns + e];
        mo[2 * ns + e] = g.yoffs[(long long)y1 * ns + e];
        mo[3 * ns + e] = g.yoffs[(long long)y2 * ns + e];
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
    {  // XB_j at site 0: [b_j = 16 rows][a_1 + a_2 = 32 columns], X_1j[0][0] = X_2j[0][0] = 1
      const int a1 = ldi(md), A2 = a1 + ldi(md + n1);
      for (int j = 0; j < 2; ++j) {
        const int b = ldi(md + (2 + j) * n1);
        T* Xre = base + j * jstride;
        T* Xim = Xre + xp;
        for (int e = tid; e < b * A2; e += 64 * NW) {
          Xre[e] = (e == 0 || e == a1) ? (T)1 : (T)0;
          Xim[e] = (T)0;
        }
      }
      __syncthreads();
    }
    for (int k = 0; k < ns; ++k) {
      const int a1 = ldi(md + k), a2 = ldi(md + n1 + k), a1n = ldi(md + k + 1), a2n = ldi(md + n1 + k + 1);
      const int A2 = a1 + a2, A2n = a1n + a2n;
      for (int j = 0; j < 2; ++j) {  // phase 1: TB_j = XB_j^T B_j, contraction over the TRUE bond of y_j
        const int b = ldi(md + (2 + j) * n1 + k), bn = ldi(md + (2 + j) * n1 + k + 1);
        const T* Bre = ydata + ldl(mo + (2 + j) * ns + k);
        const T* Bim = Bre + (long long)b * 2 * bn;
        T* Xre = base + j * jstride;
        T* Xim = Xre + xp;
        T* Tre = Xim + xp;
        T* Tim = Tre + tp;
        zgemm_ring3<false, KTL, NSLOT, true, NW, 64, T, 1>(Tre, Tim, 2 * bn, Xre, Xim, A2, Bre, Bim, 2 * bn, A2, 2 * bn, ldi(md + (6 + j) * n1 + k), lds);
      }
      for (int jj = 0; jj < 4; ++jj) {  // phase 2: X'_ij = T_ij^T conj(A_i), contraction over the 2 a_i TRUE rows (a, p)
        const int i = jj >> 1, j = jj & 1;  // order (i, j) = (1,1) (1,2) (2,1) (2,2): the two reads of A_i follow each other
        const int bn = ldi(md + (2 + j) * n1 + k + 1);
        T* Xre = base + j * jstride;
        T* Xim = Xre + xp;
        T* Tre = Xim + xp;
        T* Tim = Tre + tp;
        const int a = i ? a2 : a1, an = i ? a2n : a1n;
        const T* Are = xdata + ldl(mo + i * ns + k);
        const T* Aim = Are + (long long)a * 2 * an;
        const long long trow = i ? (long long)a1 * 2 * bn : 0;  // T_2j starts after the a_1 rows of TB_j
        const int ccol = i ? a1n : 0;
        zgemm_ring3<true, KTL, NSLOT, true, NW, 64, T, 1>(Xre + ccol, Xim + ccol, A2n, Tre + trow, Tim + trow, bn, Are, Aim, an, bn, an, 2 * ldi(md + (4 + i) * n1 + k), lds);
      }
    }
    if (tid < 4) {  // XB_j is [16][32] now: z_1j at [0][0], z_2j at [0][a_1 = 16]
      const int j = tid >> 1, i = tid & 1;
      const int a1n = md[ns];
      const T* Xre = base + j * jstride;
      const double re = (double)Xre[i ? a1n : 0], im = (double)Xre[xp + (i ? a1n : 0)];
      const long long p = 4 * qd + 2 * j + i;
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
// host side of the lab
// ----------------------------------------------------------------------------------------
int qk_lab_init(qk_ctx* c) {
  (void)c;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_quad_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_quad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_flat_kernel<64, 16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_flat_kernel<64, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<64, 16>::LDS_B));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_deep_kernel<64, 16, 2, 4, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_deep_kernel<64, 16, 4, 8, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_deep_kernel<64, 16, 4, 8, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_group_kernel<64, 16, 4, 8, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_duo_kernel<64, 16, 4, 8, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_sweep_lean_kernel<4, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
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
  return QK_OK;
}

int qk_lab_launch_quad(qk_ctx* c, const SweepArgs& a, int grid, int n_sites, bool f32) {
  const size_t lds_quad = 3 * 16 * 1024 + 16 + (size_t)(8 * (n_sites + 1) + 2) * sizeof(int) + (size_t)4 * n_sites * sizeof(long long);
  if (lds_quad > 80 * 1024) return fail(QK_EINVAL, "qk_gram_values: %d sites need %zu bytes of LDS per workgroup (limit 80 KiB for 2 workgroups per CU)", n_sites, lds_quad);
  if (f32) qk_sweep_quad_kernel<float><<<dim3(grid), dim3(512), lds_quad, c->stream>>>(a);
  else qk_sweep_quad_kernel<double><<<dim3(grid), dim3(512), lds_quad, c->stream>>>(a);
  return QK_OK;
}

int qk_lab_launch(qk_ctx* c, int variant, const SweepArgs& a, int grid, int n_sites) {
  constexpr size_t lds_b = GemmCfg<64, 16>::LDS_B;
  // deep kernels also keep the pair's per-site metadata in LDS: 4 (n+1) ints + 2 n int64 (+ alignment)
  const size_t lds_deep = lds_b + 16 + (size_t)(4 * (n_sites + 1) + 2) * sizeof(int) + (size_t)2 * n_sites * sizeof(long long);
  if (lds_deep > 80 * 1024) return fail(QK_EINVAL, "qk_gram_values: %d sites need %zu bytes of LDS per workgroup (limit 80 KiB for 2 workgroups per CU)", n_sites, lds_deep);
  switch (variant) {
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
      const int ns = n_sites;
      const size_t lds_group = lds_b + 16 + (GMAX + 1) * sizeof(GemmDesc) + (size_t)(1 + GMAX) * ns * sizeof(long long) + (size_t)(2 + 2 * GMAX) * (ns + 1) * sizeof(int);
      if (lds_group > 80 * 1024) return fail(QK_EINVAL, "qk_gram_values: %d sites need %zu bytes of LDS per workgroup", ns, lds_group);
      qk_sweep_group_kernel<64, 16, 4, 8, 64><<<dim3(grid), dim3(512), lds_group, c->stream>>>(a);
      break;
    }
    case 16: {  // duo sweep: two independent pairs per workgroup, phases interleaved in one stream
      const int ns = n_sites;
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
    default:
      return fail(QK_EINVAL, "QK_VARIANT=%d is not a kernel of this build", variant);
  }
  return QK_OK;
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

