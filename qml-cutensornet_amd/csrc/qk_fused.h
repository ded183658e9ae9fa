// qk_fused.h -- the site-fused sweep (qk_sweep_fused_kernel): the shipped fp64 hot path for sets with bonds > 16.
//
// One overlap <x|y> (reference: MPS.vdot, gpu_backend/kernel_state_ansatz.py:380; KernelPkg.jl:106) is the chain
//     X_0 = 1,   T[a, p, b'] = sum_b X[b, a] B_k[b, p, b'],   X'[b', a'] = sum_{a, p} T[a, p, b'] conj(A_k[a, p, a'])
// The ring sweep (qk_ring.h) runs it as two GEMMs per site with X and T in an L2-resident scratch: 40 % of its fabric
// traffic is the T round trip and every GEMM starts with a write->read stall.  Here T never exists in memory:
//
//   * ONE 8-wave workgroup per CU (256 VGPRs per wave, the whole 160 KiB LDS) carries a pair; X lives in LDS
//     (complex128 interleaved, [b][a] row-major = "k-major" for the next site) whenever b^ a^ <= XCAP (8192: 98.7 % of
//     the sites and 88 % of the matrix work of the 60-qubit x 6-layer headline workload);
//   * the unit of work is an ITEM (ta, tb) = one 16-row block of a times one 16-column block of b'.  Phase 1: the wave
//     that owns the item computes the two tiles T[ta, p = 0/1, tb] (K = b) and KEEPS them in registers -- the C/D
//     layout of v_mfma_f64_16x16x4_f64 (register r of lane (q, j) = C[q + 4r][j]) is the A-operand layout of a k-major
//     operand with k-step r, so the tiles are fed straight back as the A operand of phase 2:
//     X'[tb, tn] += sum_p T_p^T conj(A_k[ta rows, p, tn cols]) for every column block tn of a'.  The sum over ta (other
//     waves' items) is taken in LDS with ds_add_f64;
//   * site tensors are read straight from the set image into B-operand fragments: one 16-byte load per lane and k-step
//     (complex128 interleaved image, rows of 16 elements = 256 contiguous bytes), prefetched one group of four k-steps
//     ahead.  No staging ring, no per-K-tile barrier: a wave runs its items autonomously and the workgroup meets at
//     three barriers per site (X read / X' zeroed / X' complete).  The fragments that several items share are
//     re-read through L1/L2, never through the fabric;
//   * K is walked in units of 4 up to the TRUE bond; the complex product is the 3M form of the ring kernel;
//   * sites too large for LDS run in STRIPS: X is read from a per-workgroup global buffer (A-operand fragments loaded
//     like the site tensors), X' is accumulated strip by strip (a block of b' rows at a time, sized so that the
//     strip's items fit the T registers and its rows the LDS) and written back to the other global buffer.
// fp64 only: the f32 MFMA's C layout (C[4q + r][j]) is not an operand layout (the complex64 sweep stays on qk_ring.h).
#pragma once
#include "qk_device.h"

typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_v2d;

#ifndef QKF_NW_
#define QKF_NW_ 8
#define QKF_SLOTS_ 2
#endif
#ifndef QKF_XCAP_
#define QKF_XCAP_ 8192
#endif
static constexpr int QKF_NW = QKF_NW_;        // waves per workgroup (two per SIMD, 256 registers each)
static constexpr int QKF_XCAP = QKF_XCAP_;   // complex elements of the LDS X buffer (128 KiB)
static constexpr int QKF_SLOTS = QKF_SLOTS_;     // T slots (items) per wave and phase: 16 VGPRs per tile, two tiles per item

// one complex k-step, 3M form: (ar + i ai) * (br + i s bi), s = +1 | -1 (CONJB)
template <bool CONJB>
__device__ __forceinline__ void qkf_kstep(v4d& p1, v4d& p2, v4d& p3, const double ar, const double ai, const double br, const double bi) {
  const double sa = ar + ai, sb = CONJB ? br - bi : br + bi;
  p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, p1, 0, 0, 0);
  p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, p2, 0, 0, 0);
  p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, sb, p3, 0, 0, 0);
}

struct QkfTile {
  v4d re, im;
};

// 16-byte fragment load from a wave-uniform base and a 32-bit lane offset (in elements): the addressing form
// global_load_dwordx4 v, v_off, s[base:base+1] -- one VGPR of address state per stream
__device__ __forceinline__ v2d qkf_ldg(const v2d* __restrict__ base, const unsigned off) {
  return *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(base) + (size_t)(off * 16u));
}
__device__ __forceinline__ v2d qkf_ldx(const v2d* __restrict__ base, const unsigned off) { return qkf_ldg(base, off); }
__device__ __forceinline__ v2d qkf_ldx(const lds_v2d* base, const unsigned off) { return base[off]; }

// Phase 1, one tile: T[ta, p, tb] = sum_{l < 4 nks} X[l][16 ta + .] * B[l][p][16 tb + .].
//   B operand: element (bp + boff + i * bstep) for k-step i (bp uniform, boff = this lane's element of k-step 0);
//   A operand: X element (xp + xoff + i * xstep), X in LDS or in the global buffer.
// Four k-steps of fragments are in flight: the registers of a k-step are reloaded for k-step + 4 right after its
// MFMAs (sched_barrier keeps that order).  Loads are unconditional (rows up to the padded bond exist and are zero;
// past the last group the same rows are read again), MFMAs are issued only for the k-steps below the true bond.
template <typename XPtr>
__device__ __forceinline__ void qkf_p1_tile(v4d& p1, v4d& p2, v4d& p3, const v2d* __restrict__ bp, unsigned boff, const int bstep, XPtr xp, unsigned xoff, const int xstep, const int nks) {
  p1 = p2 = p3 = (v4d){0, 0, 0, 0};
  v2d fb[4], fx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fb[i] = qkf_ldg(bp + i * bstep, boff), fx[i] = qkf_ldx(xp + i * xstep, xoff);
  const int nfull = nks >> 2, tail = nks & 3, ng = (nks + 3) >> 2;
#pragma unroll 1
  for (int gq = 0; gq < nfull; ++gq) {
    const int adv = (gq + 1 < ng) ? 4 : 0;
    boff += adv * bstep, xoff += adv * xstep;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fb[i].x, fb[i].y);
      fb[i] = qkf_ldg(bp + i * bstep, boff), fx[i] = qkf_ldx(xp + i * xstep, xoff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (i < tail) qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fb[i].x, fb[i].y);
}

// Phase 2, one item: X'[tb rows, tn cols] += sum_p T_p^T conj(A[16 ta + ., p, 16 tn + .]) for every tn, accumulated
// into the LDS image `xo` (row stride a2, this item's 16 rows start at xo).  A operand element of k-step i, block p,
// column block tn: ap + aoff + (p + 8 i) a2 + 16 tn, aoff = this lane's ((16 ta + q) * 2) * a2 + j.  FULL: all four k-steps
// of this ta block lie below the true bond (every block but the last one of a ragged bond); otherwise kmax of them do.
template <bool FULL>
__device__ __forceinline__ void qkf_p2_item(const QkfTile& t0, const QkfTile& t1, const v2d* __restrict__ ap, unsigned aoff, const int a2, const int nn, const int kmax,
                                            lds_v2d* xo, const int q, const int j) {
  v2d fa[4];
  const int kstep = 8 * a2;  // elements per k-step: 4 rows of a, 2 p each
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = qkf_ldg(ap + i * kstep, aoff);
  __attribute__((address_space(3))) double* d = (__attribute__((address_space(3))) double*)(xo + q * a2 + j);
#pragma unroll 1
  for (int tn = 0; tn < nn; ++tn) {
    v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // p = 0; the registers of a k-step are reloaded for p = 1 right after its MFMAs
      if (FULL || i < kmax) qkf_kstep<true>(p1, p2, p3, t0.re[i], t0.im[i], fa[i].x, fa[i].y);
      fa[i] = qkf_ldg(ap + a2 + i * kstep, aoff);
      __builtin_amdgcn_sched_barrier(0);
    }
    aoff += (tn + 1 < nn) ? TILE : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // p = 1; reloaded for p = 0 of the next tn
      if (FULL || i < kmax) qkf_kstep<true>(p1, p2, p3, t1.re[i], t1.im[i], fa[i].x, fa[i].y);
      fa[i] = qkf_ldg(ap + i * kstep, aoff);
      __builtin_amdgcn_sched_barrier(0);
    }
    const v4d re = p1 + p2, im = p3 - p1 + p2;
#ifndef QKF_EXP_NOADD
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __hip_atomic_fetch_add(d + (long)r * 8 * a2, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d + (long)r * 8 * a2 + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#else
    if (re[0] + im[1] + re[2] + im[3] == 1.2345e300) d[0] = 1.0;
#endif
    d += 2 * TILE;
  }
}

// The T tiles of a wave live in registers, S slots of two tiles.  The slot loops are NOT unrolled (unrolled, every slot
// drags ~50 VGPRs of hoisted address state through the whole sweep) and the working slot is always T[S-1]: when a
// phase has more items than waves the array is rotated by one slot before each item (phase 1 makes all S turns, so
// that item s ends in slot s; phase 2 turns once per item, which brings item s to slot S-1).
template <int S>
__device__ __forceinline__ void qkf_rotate(QkfTile (&T)[S][2]) {
  const QkfTile t0 = T[0][0], t1 = T[0][1];
#pragma unroll
  for (int e = 0; e + 1 < S; ++e) T[e][0] = T[e + 1][0], T[e][1] = T[e + 1][1];
  T[S - 1][0] = t0, T[S - 1][1] = t1;
}

template <int NW, int S>  // waves per workgroup, T slots per wave: a phase holds up to NW * S items
__global__ __launch_bounds__(64 * NW, 2) void qk_sweep_fused_kernel(const SweepArgs g) {
  constexpr int XCAP = QKF_XCAP, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  lds_v2d* const XL = (lds_v2d*)lds_raw;  // (a C-style cast: the generic -> LDS address-space cast)
  long long* const slot = reinterpret_cast<long long*>(lds_raw + 2 * XCAP);
  const v2d* const xdata = reinterpret_cast<const v2d*>(g.xdata);  // interleaved complex128 images
  const v2d* const ydata = reinterpret_cast<const v2d*>(g.ydata);
  v2d* const G0 = reinterpret_cast<v2d*>(g.scratch) + (long long)blockIdx.x * 2 * g.x_plane;  // two global X buffers of x_plane complex
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int ns = g.n_sites, n1 = ns + 1;
  int* const m_xd = reinterpret_cast<int*>(slot + 2);
  int* const m_yd = m_xd + n1;
  int* const m_xt = m_yd + n1;
  int* const m_yt = m_xt + n1;
  long long* const m_xo = reinterpret_cast<long long*>(m_xd + 4 * n1 + (4 * n1 & 1));
  long long* const m_yo = m_xo + ns;
  auto ldi = [&](const int* p_) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*p_); };
  auto ldl = [&](const long long* p_) __attribute__((always_inline)) {
    const long long v = *p_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
#ifndef QKF_TOUCH_DIST
#define QKF_TOUCH_DIST 2
#endif
  // Touch-ahead: every pair streams its two states from HBM exactly once, and a wave keeps only four fragment loads in
  // flight -- far too few bytes to cover an HBM miss.  So the waves that finish a site early (fewer items than the
  // others, or none) read one word of every 128-byte line of the tensors of site k + QKF_TOUCH_DIST while they would
  // otherwise wait at the site's last barrier: 8 KiB in flight per load instruction, and the fragments of that site
  // are then L2 / Infinity-Cache hits.  The words are folded into `sink`, which is never stored.
  unsigned sink = 0;
  auto touch = [&](const int kk, const int items_here) __attribute__((always_inline)) {
    if (kk >= ns) return;
    const int first = (items_here >= NW) ? items_here % NW : items_here;  // waves first .. NW-1 carry the lighter load
    if (wave < first) return;
    const unsigned stride = (unsigned)(NW - first) * 64u * 32u, o0 = ((unsigned)(wave - first) * 64u + lane) * 32u;
    const unsigned* An = reinterpret_cast<const unsigned*>(xdata + (ldl(m_xo + kk) >> 1));
    const unsigned* Bn = reinterpret_cast<const unsigned*>(ydata + (ldl(m_yo + kk) >> 1));
    const unsigned nA = (unsigned)(ldi(m_xd + kk) * 2 * ldi(m_xd + kk + 1)) * 4u, nB = (unsigned)(ldi(m_yd + kk) * 2 * ldi(m_yd + kk + 1)) * 4u;  // dwords
    for (unsigned o = o0; o < nA; o += 4 * stride) {
      const unsigned u0 = An[o], u1 = (o + stride < nA) ? An[o + stride] : 0u, u2 = (o + 2 * stride < nA) ? An[o + 2 * stride] : 0u, u3 = (o + 3 * stride < nA) ? An[o + 3 * stride] : 0u;
      sink ^= u0 ^ u1 ^ u2 ^ u3;
    }
    for (unsigned o = o0; o < nB; o += 4 * stride) {
      const unsigned u0 = Bn[o], u1 = (o + stride < nB) ? Bn[o + stride] : 0u, u2 = (o + 2 * stride < nB) ? Bn[o + 2 * stride] : 0u, u3 = (o + 3 * stride < nB) ? Bn[o + 3 * stride] : 0u;
      sink ^= u0 ^ u1 ^ u2 ^ u3;
    }
  };
#ifdef QKF_PROF  // experiment builds only: cycle sums per section of a wave's life (tools/fused_sections.py)
  unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
  const unsigned long long pt0 = pt;
#define QKF_STAMP(i)                                          \
  do {                                                        \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    pf[i] += now_ - pt;                                       \
    pt = now_;                                                \
  } while (0)
#else
#define QKF_STAMP(i)
#endif
  for (;;) {
    if (tid == 0) *slot = (long long)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p >= g.npairs) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    for (int e = tid; e < n1; e += NT) {
      m_xd[e] = g.xdims[(long long)xi * n1 + e];
      m_yd[e] = g.ydims[(long long)yj * n1 + e];
      m_xt[e] = g.xtrue[(long long)xi * n1 + e];
      m_yt[e] = g.ytrue[(long long)yj * n1 + e];
      if (e < ns) {
        m_xo[e] = g.xoffs[(long long)xi * ns + e];
        m_yo[e] = g.yoffs[(long long)yj * ns + e];
      }
    }
    for (int e = tid; e < TILE * TILE; e += NT) XL[e] = (v2d){e == 0 ? 1.0 : 0.0, 0.0};  // X_0 = 1 in a 16 x 16 block
    __syncthreads();
    QKF_STAMP(0);  // pair set-up
    bool xg = false;  // where X lives: LDS (row stride a) or the global buffer G0 + cur * x_plane
    int cur = 0;
    QkfTile T[S][2];
    for (int k = 0; k < ns; ++k) {
      const int a = ldi(m_xd + k), a2 = ldi(m_xd + k + 1), b = ldi(m_yd + k), b2 = ldi(m_yd + k + 1);
      const int at = ldi(m_xt + k), bt = ldi(m_yt + k);
      const int mt = a / TILE, nt = b2 / TILE, nn = a2 / TILE;
      const int nks = (bt + 3) >> 2;
      const v2d* const Ak = xdata + (ldl(m_xo + k) >> 1);  // [a][2][a2]
      const v2d* const Bk = ydata + (ldl(m_yo + k) >> 1);  // [b][2][b2]
      const bool small = a * b <= XCAP && a2 * b2 <= XCAP && mt * nt <= NW * S;
      v2d* const Gc = G0 + (long long)cur * g.x_plane;
      v2d* const Gn = G0 + (long long)(cur ^ 1) * g.x_plane;
      // ---- where X has to be for this site
      if (small && xg) {  // global -> LDS
        for (int e = tid; e < a * b; e += NT) XL[e] = Gc[e];
        __syncthreads();
        xg = false;
      } else if (!small && !xg) {  // LDS -> global (the LDS is needed for the strips of X')
        for (int e = tid; e < a * b; e += NT) Gc[e] = XL[e];
        __syncthreads();
        xg = true;
      }
      QKF_STAMP(1);  // X moved between LDS and the global buffer
      // strip width (blocks of b'): the strip's rows of X' must fit the LDS; its items run in rounds of NW * S (the T registers)
#ifdef QKF_OLD_W
      const int W = small ? nt : max(1, min(nt, min((NW * S) / mt, XCAP / (TILE * a2))));
#else
      const int W = small ? nt : max(1, min(nt, XCAP / (TILE * a2)));
#endif
      for (int s0 = 0; s0 < nt; s0 += W) {
        const int w = min(W, nt - s0), items = mt * w;
        if (!small) {  // zero this strip's X' rows (the small path zeroes after phase 1: X is still being read)
          for (int e = tid; e < w * TILE * a2; e += NT) XL[e] = (v2d){0.0, 0.0};
          qk_lds_barrier();
        }
        QKF_STAMP(6);  // strip zeroing, touch-ahead
        for (int r0 = 0; r0 < items; r0 += NW * S) {  // (one round on the small path)
          // ---- phase 1: T tiles of this wave's items
          const bool multi = items - r0 > NW;  // more than one slot in use
          auto phase1 = [&](auto xbase) __attribute__((always_inline)) {
#pragma unroll 1
            for (int s = 0; s < (multi ? S : 1); ++s) {
              if (multi) qkf_rotate<S>(T);
              const int it = r0 + wave + NW * s;
              if (it < items) {
                const int tbl = it / mt, ta = it - tbl * mt;
                const unsigned boff = (unsigned)((q * 2) * b2 + (s0 + tbl) * TILE + j), xoff = (unsigned)(q * a + ta * TILE + j);
#pragma unroll 1
                for (int pp = 0; pp < 2; ++pp) {
                  v4d p1, p2, p3;
                  qkf_p1_tile(p1, p2, p3, Bk + pp * b2, boff, 8 * b2, xbase, xoff, 4 * a, nks);
                  if (pp == 0) T[S - 1][0].re = p1 - p2, T[S - 1][0].im = p3 - p1 - p2;
                  else T[S - 1][1].re = p1 - p2, T[S - 1][1].im = p3 - p1 - p2;
                }
              }
            }
          };
          if (xg) phase1((const v2d*)Gc);
          else phase1((const lds_v2d*)XL);
          QKF_STAMP(2);  // phase 1
          if (small) {
            qk_lds_barrier();  // every wave has read X: it becomes X'
            for (int e = tid; e < b2 * a2; e += NT) XL[e] = (v2d){0.0, 0.0};
            qk_lds_barrier();
          }
          QKF_STAMP(3);  // wait for the other waves' phase 1, zero X'
          // ---- phase 2: accumulate the items' contributions to X'[strip rows]
#pragma unroll 1
          for (int s = 0; s < S; ++s) {
            const int it = r0 + wave + NW * s;
            if (it >= items) break;
            if (multi) qkf_rotate<S>(T);
            const int tbl = it / mt, ta = it - tbl * mt;
            const int kmax = min(4, (at - ta * TILE + 3) >> 2);
            const unsigned aoff = (unsigned)(((ta * TILE + q) * 2) * a2 + j);
            if (kmax == 4) qkf_p2_item<true>(T[S - 1][0], T[S - 1][1], Ak, aoff, a2, nn, 4, XL + tbl * TILE * a2, q, j);
            else qkf_p2_item<false>(T[S - 1][0], T[S - 1][1], Ak, aoff, a2, nn, kmax, XL + tbl * TILE * a2, q, j);
          }
          QKF_STAMP(4);  // phase 2
        }
#ifdef QKF_TOUCH
        if (s0 + W >= nt) touch(k + QKF_TOUCH_DIST, mt * nt);
#endif
        QKF_STAMP(6);
        qk_lds_barrier();  // the strip of X' is complete
        QKF_STAMP(5);  // wait for the other waves' phase 2
        if (!small && nt > W) {  // several strips: this one goes to the other global buffer
          for (int e = tid; e < w * TILE * a2; e += NT) Gn[(long long)s0 * TILE * a2 + e] = XL[e];
          __syncthreads();
          QKF_STAMP(1);
        }
      }
      if (!small) {
        if (nt > W) cur ^= 1;  // X' was written strip by strip to Gn
        else xg = false;       // a single strip: X' is complete in LDS
      }
    }
    if (tid == 0) {
      const v2d zz = xg ? G0[(long long)cur * g.x_plane] : (v2d)XL[0];
      g.values[p] = zz.x * zz.x + zz.y * zz.y;
      if (g.z) {
        g.z[2 * p] = zz.x;
        g.z[2 * p + 1] = zz.y;
      }
    }
    __syncthreads();
  }
  if (sink == 0x5a5a5a5au && g.npairs < 0) g.values[0] = 0.0;  // never true: keeps the touch-ahead loads alive
#ifdef QKF_PROF
  if (lane == 0) {
    pf[7] = __builtin_amdgcn_s_memtime() - pt0;
    for (int i = 0; i < 8; ++i) atomicAdd(g.prof + i, pf[i]);
  }
#endif
}

// split planes (re | im) of a set image -> interleaved complex128, same offsets; one workgroup per (state, site)
__global__ void qk_interleave_kernel(const double* __restrict__ src, double* __restrict__ dst, const int32_t* __restrict__ dims, const int64_t* __restrict__ offs,
                                     const int n_sites, const long long n_tensors) {
  for (long long t = blockIdx.x; t < n_tensors; t += gridDim.x) {
    const long long s = t / n_sites;
    const int k = (int)(t - s * n_sites);
    const long long plane = (long long)dims[s * (n_sites + 1) + k] * 2 * dims[s * (n_sites + 1) + k + 1];
    const double* re = src + offs[t];
    const double* im = re + plane;
    v2d* d = reinterpret_cast<v2d*>(dst + offs[t]);
    for (long long e = threadIdx.x; e < plane; e += blockDim.x) d[e] = (v2d){re[e], im[e]};
  }
}
