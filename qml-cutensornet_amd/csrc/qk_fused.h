// qk_fused.h -- the site-fused sweep (qk_sweep_fused_kernel): the shipped fp64 hot path for sets with a bond > 32.
//
// One overlap <x|y> (reference: MPS.vdot, gpu_backend/kernel_state_ansatz.py:380; KernelPkg.jl:106) is the chain
//     X_0 = 1,   T[a, p, b'] = sum_b X[b, a] B_k[b, p, b'],   X'[b', a'] = sum_{a, p} T[a, p, b'] conj(A_k[a, p, a'])
// The ring sweep (qk_ring.h) runs it as two GEMMs per site with X and T in an L2-resident scratch: 40 % of its fabric
// traffic is the T round trip and every GEMM starts with a write->read stall.  Here T never exists in memory:
//
//   * X lives in LDS (complex128 interleaved, [b][a] row-major = "k-major" for the next site) whenever it fits the
//     workgroup's X buffer (98.7 % of the sites and 88 % of the matrix work of the 60-qubit x 6-layer headline workload
//     at 8192 elements).  When X and X' fit side by side, X' is built at the other end of the buffer (zeroed while the
//     previous site's result is consumed); otherwise it overwrites X after a barrier;
//   * the unit of work is an ITEM (ta, tb, p): one 16 x 16 tile T[ta, p, tb], dealt round-robin to the waves.
//     Phase 1: the wave computes its tile (K = b) and KEEPS it in registers -- the C/D layout of v_mfma_f64_16x16x4_f64
//     (register r of lane (q, j) = C[q + 4r][j]) is the A-operand layout of a k-major operand with k-step r, so the
//     tile is fed straight back as the A operand of phase 2:
//         X'[tb rows, tn cols] += T^T conj(A_k[ta rows, p, tn cols])     for every column block tn of a',
//     added into LDS with ds_add_f64 (the sum over ta and p is taken by the LDS);
//   * site tensors are read straight from the set image into B-operand fragments: one 16-byte load per lane and k-step
//     (complex128 interleaved image, rows of 16 elements = 256 contiguous bytes), four k-steps in flight per wave.  The
//     stream never drains: the last group of a tile loads the first group of the wave's next tile, of its first
//     phase-2 group, or of its first tile of the NEXT site, across the barriers.  No staging ring, no per-K-tile
//     barrier: a wave runs its items autonomously and the workgroup meets at two or three barriers per site.  The
//     fragments that several items share are re-read through the caches (which catch a fifth to a quarter of them: the rest comes
//     back over the fabric -- DESIGN.md, cache counters);
//   * K is walked in units of 4 up to the TRUE bond; the complex product is the 3M form of the ring kernel;
//   * sites too large for the LDS run in STRIPS: X is read from a per-workgroup global buffer (A-operand fragments
//     loaded like the site tensors), X' is accumulated a block of b' rows at a time (as many as fit the LDS), items in
//     rounds of (waves x slots), and written to the other global buffer;
//   * the chain is walked in STEPS: the first and last k sites of both states come as edge blocks (one product each), a step in
//     between is one site or two neighbouring sites contracted into one tensor of physical dimension 4 (merged image of the set),
//     whichever costs less for the pair at hand; per-step control is a 48-byte record, the step table of the pair, built by all
//     threads at pair set-up (qkf_step_table).
// Two launch shapes (chosen per launch -- or per run of pairs of a split plan -- in qkgram.hip): one 12-wave workgroup per CU
// (three waves per SIMD at 168 VGPRs, two tiles per wave) with an 8192-element X buffer -- by default in its DUAL form
// (qk_sweep_fused_dual_kernel below: the two tiles of a wave share the rows of X and of A, so every A and X fragment feeds
// two tiles), with independent single tiles in this kernel (QK_FUSED_DUAL=0) -- or two 8-wave workgroups (128 VGPRs, one tile
// per wave) with 4608 elements each -- the second workgroup fills the first one's barriers and per-site set-up, which pays
// while most of the work sits in sites that fit the smaller buffer.  A wave issues in order, so its tails, set-up and load waits
// stall its own matrix stream: the more waves share a SIMD, the better the matrix pipe is fed (8 waves x 4 slots: 482 ms
// on the headline set, 12 x 2: 452 ms, 16 x 1: 455 ms with a few spilled registers).
// Both kernels come in a DET form (template parameter; QK_DETERMINISTIC=1): the contributions to a block of X' are then added in a fixed
// order (qkf_turn_add) and a Gram is bit-reproducible, at 1.03 x the time on the headline set.
// fp64 only: the f32 MFMA's C layout (C[4q + r][j]) is not an operand layout (the complex64 sweep stays on qk_ring.h).
#pragma once
#include "qk_device.h"

// Wave priority: a wave that is NOT in a matrix block (tile tails with their adds and LDS atomics, item set-up, barriers)
// runs at raised priority, so that it gets the issue slots ahead of the other wave's matrix stream and is back at its own
// MFMAs sooner: 481.4 against 490.5 ms on the 60-qubit x 6-layer set, same box, twice (the opposite rule: 489.6).
#define QKF_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#define QKF_PRIO_HI() __builtin_amdgcn_s_setprio(2)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_v2d;
// X (b rows x a columns, LDS or the global buffers) is kept in PANELS of 16 columns: element (row, col) sits at
//     (col / 16) * (rows * 16) + row * 16 + col % 16
// so that everything a lane adds to its address inside the loops is a constant: a k-step of phase 1 (4 rows) is 64 elements, the four rows
// q + 4 r of a tile in phase 2 are 64 elements apart, the second tile of a pair (16 rows down) 256 -- immediate offsets of the LDS and
// global instructions instead of vector additions, which on gfx950 take time from the matrix pipe (lab/tools/mfma_rate.hip).  A row-major
// X (stride a) costs four additions per group of k-steps in phase 1 and eight per column block in phase 2.
static constexpr int QKF_XSTEP = 4 * TILE;        // elements between two k-steps of X (phase 1) and between the rows q + 4 r of a tile (phase 2)
static constexpr int QKF_XBLOCK = TILE * TILE;    // elements of a block of 16 rows of a panel

// Two shapes are shipped (qkgram.hip picks one per launch from the plan's work profile):
//   <12 waves, 2 slots, 8192-element X buffer, 3 waves per SIMD>: one workgroup per CU -- large bonds (more sites stay LDS-resident)
//   <8 waves, 1 slot, 4608-element X buffer, 4 waves per SIMD>: two workgroups per CU -- small / medium bonds (the second workgroup
//                                              fills the first one's barriers and per-site set-up)

// experiment switches (lab builds only: -DQKF_EXPERIMENT -D...; all default to the shipped code)
#ifndef QKF_LDS3M
#define QKF_LDS3M 0
#endif
#ifndef QKF_P2_PROBE
#define QKF_P2_PROBE 0
#endif
// ABLATION builds of the dual kernel (timing only, WRONG results; lab/tools/r04_run18.sh): bit 0 = no operand sums inside the matrix loops, bit 1 = no global
// loads inside them, bit 2 = no s_barrier in the step loop, bit 3 = no LDS reads of X inside the loops of phase 1; one-wave sweep (qk_sweep_wave2_kernel): bit 4 = no LDS-DMA, bit 5 = no LDS reads of the fragments, bit 6 = no additions behind a T tile; one-tile kernel (qk_sweep_fused_kernel): bits 0-3 as in the dual kernel, bit 7 = no additions behind a block of phase 2, bit 8 = no LDS adds there
#ifndef QKF_ABL
#define QKF_ABL 0
#endif
#define QKF_STEP_BARRIER()                            \
  do {                                                \
    if (QKF_ABL & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    else qk_lds_barrier();                            \
  } while (0)
#define QKF_SUM(a, b) ((QKF_ABL & 1) ? (a) : (a) + (b))
#define QKF_DIF(a, b) ((QKF_ABL & 1) ? (b) : (a) - (b))
// one complex k-step, 3M form: (ar + i ai) * (br + i s bi), s = +1 | -1 (CONJB)
template <bool CONJB>
__device__ __forceinline__ void qkf_kstep(v4d& p1, v4d& p2, v4d& p3, const double ar, const double ai, const double br, const double bi) {
  const double sa = QKF_SUM(ar, ai), sb = CONJB ? QKF_DIF(br, bi) : QKF_SUM(br, bi);
  p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, p1, 0, 0, 0);
  p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, p2, 0, 0, 0);
  p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, sb, p3, 0, 0, 0);
}

struct QkfTile {  // a 16 x 16 complex tile in the C/D register layout: 16 VGPRs
  v4d re, im;
};

// 16-byte fragment load from a wave-uniform base and a 32-bit lane offset (in elements): the addressing form
// global_load_dwordx4 v, v_off, s[base:base+1] -- one VGPR of address state per stream
__device__ __forceinline__ v2d qkf_ldg(const v2d* __restrict__ base, const unsigned off) {
  return *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(base) + (size_t)(off * 16u));
}
// experiment builds (-DQKF_EXPERIMENT -DQKF_NT_A=1 / -DQKF_NT_B=1): the phase-2 (A) / phase-1 (B) fragment stream with non-temporal loads
#ifndef QKF_NT_A
#define QKF_NT_A 0
#endif
#ifndef QKF_NT_B
#define QKF_NT_B 0
#endif
__device__ __forceinline__ v2d qkf_ldg_nt(const v2d* __restrict__ base, const unsigned off) {
  return __builtin_nontemporal_load(reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(base) + (size_t)(off * 16u)));
}
__device__ __forceinline__ v2d qkf_ldg_a(const v2d* __restrict__ base, const unsigned off) { return QKF_NT_A ? qkf_ldg_nt(base, off) : qkf_ldg(base, off); }
__device__ __forceinline__ v2d qkf_ldg_b(const v2d* __restrict__ base, const unsigned off) { return QKF_NT_B ? qkf_ldg_nt(base, off) : qkf_ldg(base, off); }
__device__ __forceinline__ v2d qkf_ldx(const v2d* __restrict__ base, const unsigned off) { return qkf_ldg(base, off); }
__device__ __forceinline__ v2d qkf_ldx(const lds_v2d* base, const unsigned off) { return base[off]; }

// A fragment stream: k-step i of the current group is element base + off + i * step (base wave-uniform, off this lane's)
struct QkfStream {
  const v2d* base;
  unsigned off;
  int step;
};
__device__ __forceinline__ void qkf_load4(v2d (&fr)[4], const QkfStream& st) {
#pragma unroll
  for (int i = 0; i < 4; ++i) fr[i] = qkf_ldg(st.base + i * st.step, st.off);
}

// Phase 1, one tile (= one item): T[ta, p, tb] = sum_{l < 4 nks} X[l][16 ta + .] * B[l][p][16 tb + .].
//   B operand: stream `cur` (step = one k-step = 4 rows of b); its first group is already in `fr` when `primed`;
//   A operand: X element (xp + xoff + 64 i) of the panel of columns 16 ta .. (k-step i: rows 4 i + q), X in LDS or in the global buffer.
// Four k-steps of fragments are in flight: the registers of a k-step are reloaded for k-step + 4 right after its
// MFMAs (sched_barrier keeps that order); in the LAST group they are reloaded with the first group of stream `nxt`
// -- the wave's next tile, or its first phase-2 group -- so the stream never drains between tiles or across the
// barriers.  Loads are unconditional (rows up to the padded bond exist and are zero), MFMAs are issued only for the
// k-steps below the true bond.
template <typename XPtr>
__device__ __forceinline__ void qkf_p1_tile(QkfTile& t, v2d (&fr)[4], const bool primed, QkfStream cur, XPtr xp, unsigned xoff, const int nks, const QkfStream nxt) {
  v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
  v2d fx[4];
  if (!primed) qkf_load4(fr, cur);
#pragma unroll
  for (int i = 0; i < 4; ++i) fx[i] = qkf_ldx(xp + i * QKF_XSTEP, xoff);
  const int ng = (nks + 3) >> 2, last = nks - 4 * (ng - 1);  // k-steps of the last group: 1..4
  QKF_PRIO_LO();
  int gq = 0;
  // the first k-step of a chain of more than four k-steps STARTS the accumulators (literal zero as the C operand): no register moves to zero
  // them -- on gfx950 every vector instruction takes time from the matrix pipe (lab/tools/mfma_rate.hip)
  if (ng >= 2) {
    gq = 1;
    cur.off += 4 * cur.step, xoff += 4 * QKF_XSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 0) {
        const v4d z = {0, 0, 0, 0};
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fx[0].x, fr[0].x, z, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fx[0].y, fr[0].y, z, 0, 0, 0);
        p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(QKF_SUM(fx[0].x, fx[0].y), QKF_SUM(fr[0].x, fr[0].y), z, 0, 0, 0);
      } else {
        qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fr[i].x, fr[i].y);
      }
      if (!(QKF_ABL & 2)) fr[i] = qkf_ldg_b(cur.base + i * cur.step, cur.off);
      if (!(QKF_ABL & 8)) fx[i] = qkf_ldx(xp + i * QKF_XSTEP, xoff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll 1
  for (; gq + 1 < ng; ++gq) {
    cur.off += 4 * cur.step, xoff += 4 * QKF_XSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fr[i].x, fr[i].y);
      if (!(QKF_ABL & 2)) fr[i] = qkf_ldg_b(cur.base + i * cur.step, cur.off);
      if (!(QKF_ABL & 8)) fx[i] = qkf_ldx(xp + i * QKF_XSTEP, xoff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < last) qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fr[i].x, fr[i].y);
    if (!(QKF_ABL & 2)) fr[i] = qkf_ldg(nxt.base + i * nxt.step, nxt.off);
    __builtin_amdgcn_sched_barrier(0);
  }
  QKF_PRIO_HI();
  t.re = p1 - p2;
  t.im = p3 - p1 - p2;
}

// ORDERED accumulation (the DET forms of the kernels: bit-reproducible sweeps).  The contributions to one 16 x 16 block of X' -- one per
// (ta, p) -- are added in the order of their index i = pd ta + p instead of in arrival order: the block has a turn counter in LDS, the
// contribution of index 0 STORES (so X' needs no zeroing pass, and an LDS-resident site with X and X' side by side needs no barrier between
// its phases), index i waits until the counter reads i, adds and passes the turn on.  Waits always go to a lower index, and a wave takes
// its units in increasing order, so the chain cannot lock; waves that keep pace wait a few cycles per block (the LDS unit serialises adds
// to one block anyway).  The counters of a step are zeroed during the step before (two sets, alternating).
typedef __attribute__((address_space(3))) int lds_int;
struct QkfTurn {
  lds_int* at;                 // the turn counters of this contribution's block of rows, one per column block tn
  int idx;                     // its place in the order
  lds_int* broken;             // sticky flag of the workgroup: a wait ran out (a bug, never seen) -- every later wait is skipped, the launch ends
  unsigned long long* gerr;    // ... and is reported through this word (qk_get_stats fails the call)
};
__device__ __forceinline__ void qkf_turn_add(__attribute__((address_space(3))) double* const d, const long rs, const v4d& re, const v4d& im, const QkfTurn& t, const int tn) {
  lds_int* const turn = t.at + tn;
  if (t.idx > 0) {
    int spins = 0;
#ifdef QKF_FIRST_STORE  // experiment: only the FIRST contribution is ordered (it stores: no zeroing, no barrier between the phases); the others add in arrival order
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) {
#else
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != t.idx) {
#endif
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 20) || __builtin_amdgcn_readfirstlane(__hip_atomic_load(t.broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0) {  // never a hung GPU
        __hip_atomic_store(t.broken, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t.gerr) *t.gerr = 1ull;
        break;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __hip_atomic_fetch_add(d + r * rs, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d + r * rs + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) *(lds_v2d*)(d + r * rs) = (v2d){re[r], im[r]};
  }
#ifdef QKF_FIRST_STORE
  if (t.idx > 0) return;
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's adds have left the LDS queue before the turn moves on
  __hip_atomic_store(turn, t.idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Phase 2, one item: X'[tb rows, tn cols] += T^T conj(A[16 ta + ., p, 16 tn + .]) for every column block tn, added into
// the LDS image `xo` (row stride a2, this item's 16 rows start at xo) with ds_add_f64.  A operand: stream `cur` (group
// = column block tn: off advances by 16 per group; step = one k-step = 4 rows of a).  FULL: all four k-steps of this ta
// block lie below the true bond (every block but the last one of a ragged bond); otherwise kmax of them do.  The last
// group reloads the registers with the first group of `nxt` (the wave's next item, or its first tile of the next site).
template <bool FULL, bool DET = false>
__device__ __forceinline__ void qkf_p2_item(const QkfTile& t, v2d (&fr)[4], const bool primed, QkfStream cur, const int ps, const int nn, const int kmax, lds_v2d* xo, const int q,
                                            const int j, const QkfStream nxt, const QkfTurn turn = QkfTurn{nullptr, 0, nullptr, nullptr}) {
  if (!primed) qkf_load4(fr, cur);
  __attribute__((address_space(3))) double* d = (__attribute__((address_space(3))) double*)(xo + q * TILE + j);
#pragma unroll 1
  for (int tn = 0; tn < nn; ++tn) {
    v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
    const bool fin = tn + 1 == nn;
    const v2d* const rb = fin ? nxt.base : cur.base;
    const int rs = fin ? nxt.step : cur.step;
    cur.off = fin ? nxt.off : cur.off + TILE;
    QKF_PRIO_LO();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (FULL || i < kmax) qkf_kstep<true>(p1, p2, p3, t.re[i], t.im[i], fr[i].x, fr[i].y);
      if (!(QKF_ABL & 2)) fr[i] = qkf_ldg_a(rb + i * rs, cur.off);
      __builtin_amdgcn_sched_barrier(0);
    }
    QKF_PRIO_HI();
    const v4d re = (QKF_ABL & 128) ? p1 : p1 + p2, im = (QKF_ABL & 128) ? p3 : p3 - p1 + p2;
    if constexpr (DET) qkf_turn_add(d, (long)2 * QKF_XSTEP, re, im, turn, tn);
    else if (QKF_ABL & 256) {
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(re[r]), "v"(im[r]));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        __hip_atomic_fetch_add(d + r * 2 * QKF_XSTEP, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(d + r * 2 * QKF_XSTEP + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    d += 2 * ps;  // the next panel
  }
}

// The T tiles of a wave live in registers, S slots.  The slot loops are NOT unrolled (unrolled, every slot drags ~50
// VGPRs of hoisted address state through the whole sweep) and the working slot is always T[S-1]: when a round gives
// the waves up to L > 1 items each, the last L slots are turned by one before each item (phase 1 makes all L turns, so
// that item s ends in slot S-L+s; phase 2 turns once per item, which brings item s to slot S-1).  S <= 4.
template <int S>
__device__ __forceinline__ void qkf_rotate(QkfTile (&T)[S], const int L) {  // turn the last L <= S slots by one: T[S-L] goes to T[S-1]
  if constexpr (S >= 2) {
    if (L == 2) {
      const QkfTile t = T[S - 2];
      T[S - 2] = T[S - 1], T[S - 1] = t;
    }
  }
  if constexpr (S >= 3) {
    if (L == 3) {
      const QkfTile t = T[S - 3];
      T[S - 3] = T[S - 2], T[S - 2] = T[S - 1], T[S - 1] = t;
    }
  }
  if constexpr (S >= 4) {
    if (L >= 4) {
      const QkfTile t0 = T[0];
#pragma unroll
      for (int e = 0; e + 1 < S; ++e) T[e] = T[e + 1];
      T[S - 1] = t0;
    }
  }
}

// ----------------------------------------------------------------------------------------
// Edge blocks (SweepArgs.edge_k): the contraction order chosen on the host for the ENDS of the chain.  While the bonds still grow
// like 2^k, contracting the first k sites of each state into one matrix L[s][a] and taking X = Ly^T conj(Lx) -- one product with
// K = 2^k -- is cheaper than k sites of the chain, and it saves their per-site costs (two barriers, set-up, load latencies: 3-5 us
// each, as much as the matrix work of a 64 x 64 site); likewise at the right end, where the overlap is sum_{b,a} X[b][a] R[b][a]
// with R = Ry^T conj(Rx).  On the 60-qubit x 6-layer set the model picks k = 8: 16 of 60 sites go.
// ----------------------------------------------------------------------------------------
// One 16 x 16 tile of  Ay^T conj(Ax)  (K = ks4 k-steps of 4 rows): Ay / Ax point at this lane's element of k-step 0 (row q, column
// 16 t + j), ldy / ldx = elements per row.
__device__ __forceinline__ void qkf_edge_tile(QkfTile& t, const v2d* __restrict__ Ay, const int ldy, const v2d* __restrict__ Ax, const int ldx, const int ks4) {
  v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
  v2d fy[4], fx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fy[i] = Ay[(long)(4 * i) * ldy], fx[i] = Ax[(long)(4 * i) * ldx];  // (ks4 >= 4: K = 2^k >= 16)
  for (int s0 = 0; s0 < ks4; s0 += 4) {
    const bool more = s0 + 4 < ks4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qkf_kstep<true>(p1, p2, p3, fy[i].x, fy[i].y, fx[i].x, fx[i].y);
      if (more) fy[i] = Ay[(long)(4 * (s0 + 4 + i)) * ldy], fx[i] = Ax[(long)(4 * (s0 + 4 + i)) * ldx];
    }
  }
  t.re = p1 + p2, t.im = p3 - p1 + p2;
}

// The environment behind the left edge: X[b][a] = sum_s Ly[s][b] conj(Lx[s][a]), b x a (padded bonds of site edge_k), written
// row-major with stride a to `xo` (LDS or the global X buffer); tiles dealt round-robin to the NW wavefronts.
template <int NW, typename XOut>
__device__ __forceinline__ void qkf_edge_prefix(const SweepArgs& g, const int xi, const int yj, const int a, const int b, XOut xo, const int wave, const int q, const int j) {
  const v2d* const Lx = reinterpret_cast<const v2d*>(g.xedge) + g.xedge_offs[2 * (long long)xi];
  const v2d* const Ly = reinterpret_cast<const v2d*>(g.yedge) + g.yedge_offs[2 * (long long)yj];
  const int mt = a / TILE, nt = b / TILE, ks4 = (1 << g.edge_k) >> 2;
  for (int t = wave; t < mt * nt; t += NW) {
    const int tb = t / mt, ta = t - tb * mt;
    QkfTile T;
    qkf_edge_tile(T, Ly + (long)q * b + tb * TILE + j, b, Lx + (long)q * a + ta * TILE + j, a, ks4);
#pragma unroll
    for (int r = 0; r < 4; ++r) xo[ta * (b * TILE) + (tb * TILE + q + 4 * r) * TILE + j] = (v2d){T.re[r], T.im[r]};  // (panels of 16 columns)
  }
}

// The overlap at the right edge: z = sum_{b,a} X[b][a] R[b][a], R = Ry^T conj(Rx); X row-major with stride a at `xin`.  Every
// wavefront leaves its tiles' share in its own two doubles of `zacc` (LDS, [NW][2]); the caller adds them up in wavefront order behind a barrier.
template <int NW, typename XIn>
__device__ __forceinline__ void qkf_edge_suffix(const SweepArgs& g, const int xi, const int yj, const int a, const int b, XIn xin, __attribute__((address_space(3))) double* zacc, const int wave,
                                                const int q, const int j) {
  const v2d* const Rx = reinterpret_cast<const v2d*>(g.xedge) + g.xedge_offs[2 * (long long)xi + 1];
  const v2d* const Ry = reinterpret_cast<const v2d*>(g.yedge) + g.yedge_offs[2 * (long long)yj + 1];
  const int mt = a / TILE, nt = b / TILE, ks4 = (1 << g.edge_k) >> 2;
  double zr = 0, zi = 0;
  for (int t = wave; t < mt * nt; t += NW) {
    const int tb = t / mt, ta = t - tb * mt;
    QkfTile T;
    qkf_edge_tile(T, Ry + (long)q * b + tb * TILE + j, b, Rx + (long)q * a + ta * TILE + j, a, ks4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const v2d x = xin[ta * (b * TILE) + (tb * TILE + q + 4 * r) * TILE + j];
      zr += x.x * T.re[r] - x.y * T.im[r], zi += x.x * T.im[r] + x.y * T.re[r];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) zr += __shfl_xor(zr, o), zi += __shfl_xor(zi, o);
  if ((q | j) == 0) zacc[2 * wave] = zr, zacc[2 * wave + 1] = zi;  // one slot per wavefront: the caller adds them up in wavefront order (reproducible)
}

// What a site needs: bonds, tile counts, where its X / X' live and how its items are cut into strips.
struct QkfSite {
  int a, a2, b, b2, at, nks, mt, nt, nn, W, inv, pd, ps, next;  // next = the table entry of the step after this one; pd = physical dimension of the step (2, or 4 for two merged sites), ps = log2 pd; inv = ceil(2^20 / mt): u / mt == (u * inv) >> 20 for u < 2048, mt <= 32 (checked exhaustively)
  bool small;
  const v2d *Ak, *Bk;
};
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i lds_v4i;

// The step table of a pair, built by all threads at pair set-up: per step 12 ints (48 bytes)
//   [0] a  [1] a2  [2] b  [3] b2  [4] true a  [5] k-steps of b  [6] W  [7] small  [8] ceil(2^20 / mt)  [9] log2 pd  [10] next step  [11] -
// and the addresses of the step's two tensors ([a][pd][a2] of x, [b][pd][b2] of y) in a second table.  A step is one site (pd = 2,
// entry = the site's index, next = + 1) or, with a merged image (SweepArgs.merge_steps), two sites contracted into one tensor of
// physical dimension 4 (entry = the first site's index, next = + 2).  The merged step does the same matrix work as its two sites when
// the bond between them is as large as the bonds around them, with half the barriers, set-ups and stream turn-arounds; it is taken
// unless it would push an LDS-resident pair of sites out of the LDS (its 4 mt nt items no longer fit one round, and X and X' do not
// fit side by side) or cost more padded work than the two sites (a dip of the bond between them).  `one_round(pd * mt, nt)`:
// the kernel's test that a single round holds all items of a step.
template <int XCAP, int NT, typename OneRound>
__device__ __forceinline__ void qkf_step_table(const SweepArgs& g, const int xi, const int yj, lds_v4i* const rec, long long* const m_off, const int tid, const OneRound one_round) {
  const int ns = g.n_sites, n1 = ns + 1;
  const int32_t* const xd = g.xdims + (long long)xi * n1;
  const int32_t* const yd = g.ydims + (long long)yj * n1;
  const int32_t* const xt = g.xtrue + (long long)xi * n1;
  const int32_t* const yt = g.ytrue + (long long)yj * n1;
  const v2d* const xdata = reinterpret_cast<const v2d*>(g.xdata);
  const v2d* const ydata = reinterpret_cast<const v2d*>(g.ydata);
  auto is_small = [&](const int a, const int a2, const int b, const int b2, const int pd) __attribute__((always_inline)) {
    return a * b <= XCAP && a2 * b2 <= XCAP && (one_round(pd * (a / TILE), b2 / TILE) || a * b + a2 * b2 <= XCAP);
  };
  auto put = [&](const int e, const int a, const int a2, const int b, const int b2, const int ps, const int next, const bool small, const v2d* const A, const v2d* const B)
                 __attribute__((always_inline)) {
    const int mt = a / TILE, nt = b2 / TILE;
    const int W = small ? nt : max(1, min(nt, XCAP / (TILE * a2)));  // X' in strips of W blocks of b' (the strip's rows must fit the LDS)
    rec[3 * e] = (v4i){a, a2, b, b2};
    rec[3 * e + 1] = (v4i){xt[e], (yt[e] + 3) >> 2, W, small ? 1 : 0};
    rec[3 * e + 2] = (v4i){((1 << 20) + mt - 1) / mt, ps, next, 0};
    // (as element offsets from the sets' plain images, whatever buffer the tensor lives in: the kernels add them to their pointer
    //  ARGUMENTS, which keeps the fragment loads global_load -- an address that comes out of the table as an integer makes them
    //  flat_load, whose waits cover the LDS counter as well)
    m_off[2 * e] = (long long)(((const char*)A - (const char*)xdata) / 16);
    m_off[2 * e + 1] = (long long)(((const char*)B - (const char*)ydata) / 16);
  };
  auto plain = [&](const int e, const int next) __attribute__((always_inline)) {
    const int a = xd[e], a2 = xd[e + 1], b = yd[e], b2 = yd[e + 1];
    put(e, a, a2, b, b2, 1, next, is_small(a, a2, b, b2, 2), xdata + (g.xoffs[(long long)xi * ns + e] >> 1), ydata + (g.yoffs[(long long)yj * ns + e] >> 1));
  };
  if (g.merge_steps == 0) {
    for (int e = tid; e < ns; e += NT) plain(e, e + 1);
    return;
  }
  const int ek = g.edge_k, k_hi = ns - ek;
  for (int t = tid; 2 * t < k_hi - ek; t += NT) {
    const int e = ek + 2 * t;
    if (e + 1 >= k_hi) {  // the single last site of an odd chain
      plain(e, e + 1);
      continue;
    }
    const int a = xd[e], am = xd[e + 1], a2 = xd[e + 2], b = yd[e], bm = yd[e + 1], b2 = yd[e + 2];
    const bool sm = is_small(a, a2, b, b2, 4);
    const long long work_m = 4ll * a * b2 * (b + a2), work_p = 2ll * a * bm * (b + am) + 2ll * am * b2 * (bm + a2);
    // (measured on the two headline sets, same box: allowing 1/8 or 1/4 more padded work per merged step loses what the saved barriers
    // gain -- 397.4 / 400.0 against 395.7 ms; merging LDS-resident steps only when that saves 6 % of the work: 406 ms)
    const bool merged = (sm || (!is_small(a, am, b, bm, 2) && !is_small(am, a2, bm, b2, 2))) && work_m <= work_p;
    plain(e + 1, e + 2);
    if (merged)
      put(e, a, a2, b, b2, 2, e + 2, sm, reinterpret_cast<const v2d*>(g.xmg) + (g.xmg_offs[(long long)xi * g.merge_steps + t] >> 1),
          reinterpret_cast<const v2d*>(g.ymg) + (g.ymg_offs[(long long)yj * g.merge_steps + t] >> 1));
    else plain(e, e + 1);
  }
}

// experiment builds only (-DQKF_PROF): cycle sums per section of a wave's life, added up over all waves into SweepArgs.prof[0..7]
// (lab/tools/fused_sections.py; printed by qk_get_stats of such a build)
#ifdef QKF_PROF
#define QKF_PROF_DECL() \
  unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime(); \
  const unsigned long long pt0 = pt
#define QKF_STAMP(i)                                              \
  do {                                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    pf[i] += now_ - pt;                                           \
    pt = now_;                                                    \
  } while (0)
#define QKF_PROF_FLUSH()                                             \
  do {                                                               \
    if (lane == 0) {                                                 \
      pf[7] = __builtin_amdgcn_s_memtime() - pt0;                    \
      for (int i_ = 0; i_ < 8; ++i_) atomicAdd(g.prof + i_, pf[i_]); \
    }                                                                \
  } while (0)
#else
#define QKF_PROF_DECL()
#define QKF_STAMP(i)
#define QKF_PROF_FLUSH()
#endif

template <int NW, int S, int XCAP, int WPS, bool DET = false>  // waves per workgroup; T slots per wave (a round holds NW * S items); elements of the LDS X buffer; waves per SIMD (register budget); ordered accumulation (bit-reproducible)
__global__ __launch_bounds__(64 * NW, WPS) void qk_sweep_fused_kernel(const SweepArgs g) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  lds_v2d* const XL = (lds_v2d*)lds_raw;  // (a C-style cast: the generic -> LDS address-space cast)
  long long* const slot = reinterpret_cast<long long*>(lds_raw + 2 * XCAP);
  const v2d* const xdata = reinterpret_cast<const v2d*>(g.xdata);  // interleaved complex128 images
  const v2d* const ydata = reinterpret_cast<const v2d*>(g.ydata);
  v2d* const G0 = reinterpret_cast<v2d*>(g.scratch) + (long long)blockIdx.x * 2 * g.x_plane;  // two global X buffers of x_plane complex
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int ns = g.n_sites;
  // the step table of the current pair (qkf_step_table): 12 ints per step, and the addresses of the step's two tensors in a second table
  __attribute__((address_space(3))) double* const zacc = (__attribute__((address_space(3))) double*)(slot + 2);  // the overlap's partial sums (edge blocks): [16 wavefronts][2]
  lds_v4i* const rec = (lds_v4i*)(slot + 2 + 32);
  long long* const m_off = reinterpret_cast<long long*>(slot + 2 + 32) + 6 * (long long)ns;  // [ns][2]: A_k, B_k
  lds_int* const tbroken = (lds_int*)(m_off + 2 * (long long)ns);  // DET: the workgroup's sticky 'a wait ran out' flag, then
  lds_int* const turn0 = tbroken + 2;                              // two sets of g.turn_ints turn counters (qkf_turn_add)
  auto rfl = [&](const int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  auto ldl = [&](const long long* p_) __attribute__((always_inline)) {
    const long long v = *p_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  auto site = [&](const int k) __attribute__((always_inline)) {
    const v4i r0 = rec[3 * k], r1 = rec[3 * k + 1], r2 = rec[3 * k + 2];  // every lane reads the same 48 bytes
    QkfSite s;
    s.a = rfl(r0.x), s.a2 = rfl(r0.y), s.b = rfl(r0.z), s.b2 = rfl(r0.w);
    s.at = rfl(r1.x), s.nks = rfl(r1.y), s.W = rfl(r1.z), s.small = rfl(r1.w) != 0;
    s.inv = rfl(r2.x);
    s.ps = rfl(r2.y), s.pd = 1 << s.ps, s.next = rfl(r2.z);
    s.mt = s.a / TILE, s.nt = s.b2 / TILE, s.nn = s.a2 / TILE;
    s.Ak = xdata + ldl(m_off + 2 * k);      // [a][pd][a2]
    s.Bk = ydata + ldl(m_off + 2 * k + 1);  // [b][pd][b2]
    return s;
  };
  // the streams of item `it` of a strip starting at block s0 (it = 2 (tbl * mt + ta) + p)
  auto b_stream = [&](const QkfSite& s, const int s0, const int it) __attribute__((always_inline)) {
    const int pp = it & (s.pd - 1), u = it >> s.ps, tbl = (u * s.inv) >> 20;
    return QkfStream{s.Bk + pp * s.b2, (unsigned)((q * s.pd) * s.b2 + (s0 + tbl) * TILE + j), 4 * s.pd * s.b2};
  };
  auto a_stream = [&](const QkfSite& s, const int it) __attribute__((always_inline)) {
    const int pp = it & (s.pd - 1), u = it >> s.ps, tbl = (u * s.inv) >> 20, ta = u - tbl * s.mt;
    return QkfStream{s.Ak + pp * s.a2, (unsigned)(((ta * TILE + q) * s.pd) * s.a2 + j), 4 * s.pd * s.a2};
  };
  QKF_PROF_DECL();
  const int xcc = qk_xcc_id();
  int gang_round = 0;  // (gang start: this workgroup's pairs so far)
  if (tid == 0) qk_tail_start(g);
  for (;;) {
    if (tid == 0) {
      const long long pp_ = qk_pull(g, xcc);
      qk_gang_sync(g, xcc, pp_ >= 0, gang_round);
      *slot = pp_;
    }  // this XCD's queue first: the workgroups that share an L2 stream the same few states
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p < 0) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    // an item is (ta, tb, p): pd mt nt of them.  LDS-resident step: X and X' fit the buffer and either ONE round holds all items
    // (X' may then overwrite X) or X and X' fit side by side (any number of rounds: X stays intact);
    // otherwise X' is built in strips of W blocks of b' (the strip's rows must fit the LDS), items in rounds of NW * S
    qkf_step_table<XCAP, NT>(g, xi, yj, rec, m_off, tid, [](const int pmt, const int nt) { return pmt * nt <= NW * S; });
    const int ek = g.edge_k, k_hi = ns - ek;  // the chain runs over the sites [ek, k_hi): the ends are in the edge blocks
    const bool edges = ek > 0;
    if (!edges)
      for (int e = tid; e < TILE * TILE; e += NT) XL[e] = (v2d){e == 0 ? 1.0 : 0.0, 0.0};  // X_0 = 1 in a 16 x 16 block
    if constexpr (DET)
      for (int e = tid; e < 2 * g.turn_ints + 2; e += NT) tbroken[e] = 0;
    int tsel = 0;  // DET: the set of turn counters in use (the other one is zeroed meanwhile for the next strip or step)
    __syncthreads();
    bool xg = false;  // where X lives: LDS (at element xb, row stride a) or the global buffer G0 + cur * x_plane
    int cur = 0, xb = 0;
    if (edges) {  // X behind the left edge: one product of the two left blocks
      const v4i r0 = rec[3 * ek];
      const int a_e = rfl(r0.x), b_e = rfl(r0.z);
      if (a_e * b_e <= XCAP) qkf_edge_prefix<NW>(g, xi, yj, a_e, b_e, XL, wave, q, j);
      else qkf_edge_prefix<NW>(g, xi, yj, a_e, b_e, G0, wave, q, j), xg = true;
      __syncthreads();
    }
    QKF_STAMP(0);  // pair set-up
    QkfTile T[S];
    v2d fr[4];            // the fragment registers of the wave's global stream: they carry its next group across tiles and barriers
    bool primed = false;  // fr holds the first group of the wave's next tile
    QkfSite sn = site(ek);
    for (int k = ek; k < k_hi;) {
      const QkfSite sc = sn;
      k = sc.next;  // (from here on: the entry of the NEXT step)
      if (k < k_hi) sn = site(k);
      const int a = sc.a, a2 = sc.a2, b = sc.b, mt = sc.mt, nt = sc.nt, W = sc.W;
      const bool small = sc.small;
      v2d* const Gc = G0 + (long long)cur * g.x_plane;
      v2d* const Gn = G0 + (long long)(cur ^ 1) * g.x_plane;
      // ---- where X has to be for this site
      if (small && xg) {  // global -> LDS
        for (int e = tid; e < a * b; e += NT) XL[e] = Gc[e];
        __syncthreads();
        xg = false, xb = 0;
      } else if (!small && !xg) {  // LDS -> global (the LDS is needed for the strips of X')
        for (int e = tid; e < a * b; e += NT) Gc[e] = XL[xb + e];
        __syncthreads();
        xg = true;
      }
      // LDS-resident site: when X and X' fit the buffer side by side, X' goes to the other end and is zeroed right here
      // (that region held the X of the previous site, dead since its last barrier): one barrier between the phases
      // instead of barrier - zero - barrier.  Otherwise X' overwrites X from element 0.
      const int n_out = sc.b2 * a2;
      const bool pingpong = small && a * b + n_out <= XCAP;
      const int ob = !small ? 0 : pingpong ? (xb == 0 ? XCAP - n_out : 0) : 0;  // where X' (or the strip of X') is built
      if (pingpong && !DET)
        for (int e = tid; e < n_out; e += NT) XL[ob + e] = (v2d){0.0, 0.0};
      QKF_STAMP(1);  // X moved between LDS and the global buffer
      for (int s0 = 0; s0 < nt; s0 += W) {
        const int w = min(W, nt - s0), items = sc.pd * mt * w;
        lds_int* const tcur = turn0 + tsel * g.turn_ints;
        if constexpr (DET) {  // (ordered accumulation: the first contribution to a block stores, nothing is zeroed but the next set of counters)
          for (int e = tid; e < g.turn_ints; e += NT) turn0[(tsel ^ 1) * g.turn_ints + e] = 0;
        } else if (!small) {  // zero this strip's X' rows (the LDS-resident path zeroes after phase 1: X is still being read)
          for (int e = tid; e < w * TILE * a2; e += NT) XL[e] = (v2d){0.0, 0.0};
          QKF_STEP_BARRIER();
        }
        QKF_STAMP(6);  // strip zeroing
        for (int r0 = 0; r0 < items; r0 += NW * S) {  // (LDS-resident sites: one round, or several when X and X' sit side by side)
          const int L = min(S, (items - r0 + NW - 1) / NW);  // slots in use this round (the same for every wave)
          const bool multi = L > 1;
          const int it0 = r0 + wave;          // this wave's items: it0, it0 + NW, ...
          // ---- phase 1: the T tile of each of this wave's items
          auto phase1 = [&](auto xbase) __attribute__((always_inline)) {
#pragma unroll 1
            for (int s = 0; s < L; ++s) {
              if (multi) qkf_rotate<S>(T, L);
              const int it = it0 + NW * s;
              if (it < items) {
                const int u = it >> sc.ps, tbl = (u * sc.inv) >> 20, ta = u - tbl * mt;
                const bool more = s + 1 < L && it + NW < items;  // another tile follows in this phase; else phase 2 starts with item it0
                const QkfStream nxt = more ? b_stream(sc, s0, it + NW) : a_stream(sc, it0);
                qkf_p1_tile(T[S - 1], fr, primed, b_stream(sc, s0, it), xbase, (unsigned)(ta * (b * TILE) + q * TILE + j), sc.nks, nxt);
                primed = true;
              }
            }
          };
          if (xg) phase1((const v2d*)Gc);
          else phase1((const lds_v2d*)(XL + xb));
          QKF_STAMP(2);  // phase 1
          if (small && r0 == 0 && !(DET && pingpong)) {
            QKF_STEP_BARRIER();  // ping-pong: X' is zero everywhere; in place (one round): every wave has read X, it becomes X'
            if (!pingpong && !DET) {
              for (int e = tid; e < n_out; e += NT) XL[e] = (v2d){0.0, 0.0};
              QKF_STEP_BARRIER();
            }
          }
          QKF_STAMP(3);  // wait for the other waves' phase 1, zero X'
          // ---- phase 2: add the items' contributions to X'[strip rows]
          const bool last_round = s0 + W >= nt && r0 + NW * S >= items;
#pragma unroll 1
          for (int s = 0; s < S; ++s) {
            const int it = it0 + NW * s;
            if (it >= items) break;
            if (multi) qkf_rotate<S>(T, L);
            const int u = it >> sc.ps, tbl = (u * sc.inv) >> 20, ta = u - tbl * mt;
            const int kmax = min(4, (sc.at - ta * TILE + 3) >> 2);
            const bool more = s + 1 < S && it + NW < items;
            // after the wave's last item of the site: its first tile of the next site (strip 0, round 0), if it has one
            const bool chain = !more && last_round && k < k_hi && wave < sn.pd * sn.mt * min(sn.W, sn.nt);
            const QkfStream nxt = more ? a_stream(sc, it + NW) : chain ? b_stream(sn, 0, wave) : a_stream(sc, it);
            const int tix = sc.pd * ta + (it & (sc.pd - 1));  // this contribution's place in the order of its block of rows
            const QkfTurn turn{tcur + tbl * sc.nn, tix, tbroken, g.err};
            if (kmax == 4) qkf_p2_item<true, DET>(T[S - 1], fr, primed, a_stream(sc, it), w * QKF_XBLOCK, sc.nn, 4, XL + ob + tbl * QKF_XBLOCK, q, j, nxt, turn);
            else qkf_p2_item<false, DET>(T[S - 1], fr, primed, a_stream(sc, it), w * QKF_XBLOCK, sc.nn, kmax, XL + ob + tbl * QKF_XBLOCK, q, j, nxt, turn);
            primed = more || chain;
          }
          QKF_STAMP(4);  // phase 2
        }
        QKF_STEP_BARRIER();  // the strip of X' is complete
        tsel ^= 1;
        QKF_STAMP(5);      // wait for the other waves' phase 2
        if (!small && nt > W) {  // several strips: this one goes to the other global buffer
          for (int tn = 0; tn < sc.nn; ++tn)  // (panel by panel: the strip's rows of a panel are contiguous in both buffers)
            for (int e = tid; e < w * QKF_XBLOCK; e += NT) Gn[(long long)tn * (sc.b2 * TILE) + s0 * QKF_XBLOCK + e] = XL[tn * w * QKF_XBLOCK + e];
          __syncthreads();
          QKF_STAMP(1);
        }
      }
      if (!small) {
        if (nt > W) cur ^= 1;      // X' was written strip by strip to Gn
        else xg = false, xb = 0;   // a single strip: X' is complete in LDS
      } else {
        xb = ob;
      }
    }
    if (edges) {  // the overlap: X against the product of the two right blocks
      const v4i r0 = rec[3 * (k_hi - 1)];
      const int a_e = rfl(r0.y), b_e = rfl(r0.w);
      if (xg) qkf_edge_suffix<NW>(g, xi, yj, a_e, b_e, (const v2d*)(G0 + (long long)cur * g.x_plane), zacc, wave, q, j);
      else qkf_edge_suffix<NW>(g, xi, yj, a_e, b_e, (const lds_v2d*)(XL + xb), zacc, wave, q, j);
      __syncthreads();
    }
    if (tid == 0) {
      v2d zz = {0.0, 0.0};
      if (edges)
        for (int w_ = 0; w_ < NW; ++w_) zz.x += zacc[2 * w_], zz.y += zacc[2 * w_ + 1];
      else zz = xg ? G0[(long long)cur * g.x_plane] : (v2d)XL[xb];
      g.values[p] = zz.x * zz.x + zz.y * zz.y;
      if (g.z) {
        g.z[2 * p] = zz.x;
        g.z[2 * p + 1] = zz.y;
      }
    }
    __syncthreads();
  }
  if (tid == 0) qk_tail_exit(g);
  QKF_PROF_FLUSH();
}

// ----------------------------------------------------------------------------------------
// The DUAL form of the site-fused sweep: the unit of work is a pair of tiles T[ta, p, tb0], T[ta, p, tb0 + 1] (same rows
// of X and of A, neighbouring column blocks of B).  Phase 1 reads each X fragment once for both tiles; phase 2 reads each
// fragment of A once and feeds it to both tiles: half the A fragments, half the X fragments and half the per-item set-up
// of the single-tile form for the same matrix work.  One pair per wave and round, so there are no slots to rotate.
// ----------------------------------------------------------------------------------------
template <bool HAS1, typename XPtr>
__device__ __forceinline__ void qkf_p1_dual(QkfTile& t0, QkfTile& t1, v2d (&fr)[4], v2d (&fs)[4], const bool primed, QkfStream cur, XPtr xp, unsigned xoff, const int nks,
                                            const QkfStream nxt) {
  v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};
  v2d fx[4];
  if (!primed) qkf_load4(fr, cur);
  if (HAS1) {  // (the second column block is not carried across units: 16 registers less through phase 2)
#pragma unroll
    for (int i = 0; i < 4; ++i) fs[i] = qkf_ldg(cur.base + i * cur.step, cur.off + TILE);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) fx[i] = qkf_ldx(xp + i * QKF_XSTEP, xoff);
  const int ng = (nks + 3) >> 2, last = nks - 4 * (ng - 1);  // k-steps of the last group: 1..4
  QKF_PRIO_LO();
  int gq = 0;
  // (the first k-step of a chain of more than four k-steps STARTS the accumulators -- literal zero as the C operand --: no register moves to zero them)
  if (ng >= 2) {
    gq = 1;
    cur.off += 4 * cur.step, xoff += 4 * QKF_XSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 0) {
        const v4d z = {0, 0, 0, 0};
        const double sa = QKF_SUM(fx[0].x, fx[0].y);
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fx[0].x, fr[0].x, z, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fx[0].y, fr[0].y, z, 0, 0, 0);
        p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, QKF_SUM(fr[0].x, fr[0].y), z, 0, 0, 0);
        if (HAS1) {
          r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fx[0].x, fs[0].x, z, 0, 0, 0);
          r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fx[0].y, fs[0].y, z, 0, 0, 0);
          r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, QKF_SUM(fs[0].x, fs[0].y), z, 0, 0, 0);
        }
      } else {
        qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fr[i].x, fr[i].y);
        if (HAS1) qkf_kstep<false>(r1, r2, r3, fx[i].x, fx[i].y, fs[i].x, fs[i].y);
      }
      if (!(QKF_ABL & 2)) fr[i] = qkf_ldg_b(cur.base + i * cur.step, cur.off);
      if (HAS1 && !(QKF_ABL & 2)) fs[i] = qkf_ldg_b(cur.base + i * cur.step, cur.off + TILE);
      if (!(QKF_ABL & 8)) fx[i] = qkf_ldx(xp + i * QKF_XSTEP, xoff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll 1
  for (; gq + 1 < ng; ++gq) {
    cur.off += 4 * cur.step, xoff += 4 * QKF_XSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fr[i].x, fr[i].y);
      if (HAS1) qkf_kstep<false>(r1, r2, r3, fx[i].x, fx[i].y, fs[i].x, fs[i].y);
      if (!(QKF_ABL & 2)) fr[i] = qkf_ldg_b(cur.base + i * cur.step, cur.off);
      if (HAS1 && !(QKF_ABL & 2)) fs[i] = qkf_ldg_b(cur.base + i * cur.step, cur.off + TILE);
      if (!(QKF_ABL & 8)) fx[i] = qkf_ldx(xp + i * QKF_XSTEP, xoff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < last) {
      qkf_kstep<false>(p1, p2, p3, fx[i].x, fx[i].y, fr[i].x, fr[i].y);
      if (HAS1) qkf_kstep<false>(r1, r2, r3, fx[i].x, fx[i].y, fs[i].x, fs[i].y);
    }
    if (!(QKF_ABL & 2)) fr[i] = qkf_ldg(nxt.base + i * nxt.step, nxt.off);  // the first group of this pair's phase 2 (one stream: A)
    __builtin_amdgcn_sched_barrier(0);
  }
  QKF_PRIO_HI();
  t0.re = p1 - p2, t0.im = p3 - p1 - p2;
  if (HAS1) t1.re = r1 - r2, t1.im = r3 - r1 - r2;
}

// Phase 2 of a pair of tiles: X'[tb0 rows | tb0 + 1 rows, tn cols] += T0^T | T1^T conj(A[16 ta + ., p, 16 tn + .]).  `nxt` is the
// wave's next phase-1 stream (NXT_P1: both column blocks are loaded, the second one at + n1 elements) or a dummy.
// On gfx950 a vector-ALU instruction is not free beside the fp64 matrix instructions: whichever wave of the SIMD issues it, it takes 6 - 9
// cycles that the matrix pipe then stands still (lab/tools/mfma_rate.hip: 64.6 cycles per v_mfma_f64_16x16x4_f64 alone, + 6.6 per 32-bit and
// + 9.2 per 64-bit instruction put between them at two waves per SIMD).  So the loop over the column blocks is written for FEW instructions:
// the sums re + im of the T tiles' k-steps are taken once per unit, not once per block; the loop runs over the blocks that are followed by
// another block of this stream -- the four k-steps' addresses are then a lane offset on four wave-uniform bases that do not change -- and the
// last block, which reloads the registers with the first group of `nxt`, stands behind it; the product's second 3M form saves a third of
// the additions behind a block.  cfg4, one box, step by step: 374.4 -> 370.6 (this loop) -> 368.0 (3M form) -> 364.3 (phase 1 started by its
// first k-step) -> 363.9 ms (the same in the one-tile kernel).
typedef __attribute__((address_space(3))) double lds_double;
// one column block: the matrix instructions of the k-steps in `fr`, each followed by the reload of its registers from (b_i, off), then the
// results added to X' at d (and d1 for the second tile)
template <bool FULL, bool HAS1>
__device__ __forceinline__ void qkf_p2_block(const QkfTile& t0, const QkfTile& t1, const v4d& s0, const v4d& s1, v2d (&fr)[4], const int kmax, const v2d* const b0, const v2d* const b1,
                                             const v2d* const b2, const v2d* const b3, const unsigned off, lds_double* const d, lds_double* const d1, const long rs) {
  v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};
  QKF_PRIO_LO();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (FULL || i < kmax) {
      // the 3M product in the form  k1 = (ar + ai) br,  k2 = ar (br + bi),  k3 = ai (br - bi):  re = k1 - k3,  im = k1 - k2 -- two additions per
      // element behind the block instead of three (and two per fragment, shared by both tiles, instead of one)
#if QKF_LDS3M  // (the LDS takes the product's last two additions: p2 = - k2, p3 = - k3 -- the signs ride on the operand sums as source modifiers)
      const double sp = -fr[i].x - fr[i].y, sm = fr[i].y - fr[i].x;
#else
      const double sp = QKF_SUM(fr[i].x, fr[i].y), sm = QKF_DIF(fr[i].x, fr[i].y);
#endif
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s0[i], fr[i].x, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.re[i], sp, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.im[i], sm, p3, 0, 0, 0);
      if (HAS1) {
        r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s1[i], fr[i].x, r1, 0, 0, 0);
        r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t1.re[i], sp, r2, 0, 0, 0);
        r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(t1.im[i], sm, r3, 0, 0, 0);
      }
    }
    if (!(QKF_ABL & 2)) fr[i] = qkf_ldg_a(i == 0 ? b0 : i == 1 ? b1 : i == 2 ? b2 : b3, off);
    __builtin_amdgcn_sched_barrier(0);
  }
  QKF_PRIO_HI();
#if QKF_LDS3M
  // re = k1 - k3 and im = k1 - k2 are not formed in registers: the three accumulators go to X' as they are (k1 to both parts), 16 LDS adds per
  // tile in place of 8 vector additions + 8 LDS adds -- an LDS instruction costs the matrix pipe 1.8 cycles, a v_add_f64 9.2
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    __hip_atomic_fetch_add(d + r * rs, p1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(d + r * rs + 1, p1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(d + r * rs, p3[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(d + r * rs + 1, p2[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (HAS1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __hip_atomic_fetch_add(d1 + r * rs, r1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d1 + r * rs + 1, r1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d1 + r * rs, r3[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d1 + r * rs + 1, r2[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
#elif QKF_P2_PROBE  // TIMING PROBES (wrong results): 1 = the adds at conflict-free addresses (a lane's re and im 2 KiB apart, lanes 8 bytes apart), 2 = plain stores instead
                    // of adds, 3 = no LDS instruction at all, 4 = neither the additions nor the LDS instructions
  {
    const v4d re = p1 - p3, im = p1 - p2, re1 = r1 - r3, im1 = r1 - r2;
    lds_double* const e = d - (threadIdx.x & 63);  // the block's base + q * 16 + j (d is 2 (q * 16 + j) doubles into the block)
    lds_double* const e1 = e + 2 * QKF_XBLOCK;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (QKF_P2_PROBE == 1) {
        __hip_atomic_fetch_add(e + r * 64, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(e + r * 64 + 256, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (HAS1) {
          __hip_atomic_fetch_add(e1 + r * 64, re1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(e1 + r * 64 + 256, im1[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else if (QKF_P2_PROBE == 2) {
        *(lds_v2d*)(d + r * rs) = (v2d){re[r], im[r]};
        if (HAS1) *(lds_v2d*)(d1 + r * rs) = (v2d){re1[r], im1[r]};
      } else if (QKF_P2_PROBE == 3) {
        asm volatile("" ::"v"(re[r]), "v"(im[r]), "v"(re1[r]), "v"(im1[r]));
      } else {
        asm volatile("" ::"v"(p1[r]), "v"(p2[r]), "v"(p3[r]), "v"(r1[r]), "v"(r2[r]), "v"(r3[r]));
      }
    }
  }
#else
  {
    const v4d re = p1 - p3, im = p1 - p2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __hip_atomic_fetch_add(d + r * rs, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d + r * rs + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (HAS1) {
    const v4d re = r1 - r3, im = r1 - r2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __hip_atomic_fetch_add(d1 + r * rs, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d1 + r * rs + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
#endif
}
// the loop of the DET forms (ordered accumulation): one body for every column block, the last one selecting `nxt`, the product in its first 3M
// form (measured: the split loop of the plain form makes the ordered form 3.5 % slower)
template <bool FULL, bool HAS1, bool DET>
__device__ __forceinline__ void qkf_p2_dual_turn(const QkfTile& t0, const QkfTile& t1, v2d (&fr)[4], v2d (&fs)[4], QkfStream cur, const int ps, const int nn, const int kmax, lds_v2d* xo, const int q,
                                            const int j, const QkfStream nxt, const bool nxt_p1, const unsigned n1, const QkfTurn turn = QkfTurn{nullptr, 0, nullptr, nullptr}) {
  __attribute__((address_space(3))) double* d = (__attribute__((address_space(3))) double*)(xo + q * TILE + j);
#pragma unroll 1
  for (int tn = 0; tn < nn; ++tn) {
    v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};
    const bool fin = tn + 1 == nn;
    const v2d* const rb = fin ? nxt.base : cur.base;
    const int rs = fin ? nxt.step : cur.step;
    cur.off = fin ? nxt.off : cur.off + TILE;
    QKF_PRIO_LO();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (FULL || i < kmax) {
        qkf_kstep<true>(p1, p2, p3, t0.re[i], t0.im[i], fr[i].x, fr[i].y);
        if (HAS1) qkf_kstep<true>(r1, r2, r3, t1.re[i], t1.im[i], fr[i].x, fr[i].y);
      }
      fr[i] = qkf_ldg_a(rb + i * rs, cur.off);
      __builtin_amdgcn_sched_barrier(0);
    }
    QKF_PRIO_HI();
    {
      const v4d re = p1 + p2, im = p3 - p1 + p2;
      if constexpr (DET) qkf_turn_add(d, (long)2 * QKF_XSTEP, re, im, turn, tn);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __hip_atomic_fetch_add(d + r * 2 * QKF_XSTEP, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(d + r * 2 * QKF_XSTEP + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    if (HAS1) {
      const v4d re = r1 + r2, im = r3 - r1 + r2;
      __attribute__((address_space(3))) double* const d1 = d + 2 * QKF_XBLOCK;  // 16 rows further down
      if constexpr (DET) qkf_turn_add(d1, (long)2 * QKF_XSTEP, re, im, QkfTurn{turn.at + nn, turn.idx, turn.broken, turn.gerr}, tn);  // (the counters of the next block of rows follow)
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __hip_atomic_fetch_add(d1 + r * 2 * QKF_XSTEP, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(d1 + r * 2 * QKF_XSTEP + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    d += 2 * ps;  // the next panel
  }
}

template <bool FULL, bool HAS1, bool DET = false>
__device__ __forceinline__ void qkf_p2_dual(const QkfTile& t0, const QkfTile& t1, v2d (&fr)[4], v2d (&fs)[4], QkfStream cur, const int ps, const int nn, const int kmax, lds_v2d* xo, const int q,
                                            const int j, const QkfStream nxt, const bool nxt_p1, const unsigned n1, const QkfTurn turn = QkfTurn{nullptr, 0, nullptr, nullptr}) {
  if constexpr (DET) {
    qkf_p2_dual_turn<FULL, HAS1, DET>(t0, t1, fr, fs, cur, ps, nn, kmax, xo, q, j, nxt, nxt_p1, n1, turn);
    return;
  }
  lds_double* d = (lds_double*)(xo + q * TILE + j);
  lds_double* d1 = d + 2 * QKF_XBLOCK;  // the second tile's rows: 16 rows further down
  constexpr long rs = 2 * QKF_XSTEP;
  const v4d s0 = t0.re + t0.im, s1 = HAS1 ? t1.re + t1.im : s0;
  const v2d *const c0 = cur.base, *const c1 = cur.base + cur.step, *const c2 = cur.base + 2 * cur.step, *const c3 = cur.base + 3 * cur.step;
  unsigned off = cur.off;
#pragma unroll 1
  for (int tn = 0; tn + 1 < nn; ++tn) {
    off += TILE;
    qkf_p2_block<FULL, HAS1>(t0, t1, s0, s1, fr, kmax, c0, c1, c2, c3, off, d, d1, rs);
    d += 2 * ps, d1 += 2 * ps;  // the next panel
  }
  qkf_p2_block<FULL, HAS1>(t0, t1, s0, s1, fr, kmax, nxt.base, nxt.base + nxt.step, nxt.base + 2 * nxt.step, nxt.base + 3 * nxt.step, nxt.off, d, d1, rs);
}

template <int NW, int XCAP, int WPS, bool DET = false>  // waves per workgroup (a round holds NW pairs of tiles); elements of the LDS X buffer; waves per SIMD (register budget); ordered accumulation (bit-reproducible)
__global__ __launch_bounds__(64 * NW, WPS) void qk_sweep_fused_dual_kernel(const SweepArgs g) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  lds_v2d* const XL = (lds_v2d*)lds_raw;
  long long* const slot = reinterpret_cast<long long*>(lds_raw + 2 * XCAP);
  const v2d* const xdata = reinterpret_cast<const v2d*>(g.xdata);
  const v2d* const ydata = reinterpret_cast<const v2d*>(g.ydata);
  v2d* const G0 = reinterpret_cast<v2d*>(g.scratch) + (long long)blockIdx.x * 2 * g.x_plane;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int ns = g.n_sites;
  __attribute__((address_space(3))) double* const zacc = (__attribute__((address_space(3))) double*)(slot + 2);  // [16 wavefronts][2]
  lds_v4i* const rec = (lds_v4i*)(slot + 2 + 32);  // per-site records as in qk_sweep_fused_kernel
  long long* const m_off = reinterpret_cast<long long*>(slot + 2 + 32) + 6 * (long long)ns;
  lds_int* const tbroken = (lds_int*)(m_off + 2 * (long long)ns);  // DET: the workgroup's sticky 'a wait ran out' flag, then
  lds_int* const turn0 = tbroken + 2;                              // two sets of g.turn_ints turn counters (qkf_turn_add)
  auto rfl = [&](const int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  auto ldl = [&](const long long* p_) __attribute__((always_inline)) {
    const long long v = *p_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  auto site = [&](const int k) __attribute__((always_inline)) {
    const v4i r0 = rec[3 * k], r1 = rec[3 * k + 1], r2 = rec[3 * k + 2];
    QkfSite s;
    s.a = rfl(r0.x), s.a2 = rfl(r0.y), s.b = rfl(r0.z), s.b2 = rfl(r0.w);
    s.at = rfl(r1.x), s.nks = rfl(r1.y), s.W = rfl(r1.z), s.small = rfl(r1.w) != 0;
    s.inv = rfl(r2.x);
    s.ps = rfl(r2.y), s.pd = 1 << s.ps, s.next = rfl(r2.z);
    s.mt = s.a / TILE, s.nt = s.b2 / TILE, s.nn = s.a2 / TILE;
    s.Ak = xdata + ldl(m_off + 2 * k);
    s.Bk = ydata + ldl(m_off + 2 * k + 1);
    return s;
  };
  // the streams of pair `v` of a strip starting at block s0 (v = 2 (tp * mt + ta) + p; column blocks s0 + 2 tp, + 1)
  auto b_stream = [&](const QkfSite& s, const int s0, const int v, const int half) __attribute__((always_inline)) {
    const int pp = v & (s.pd - 1), u = v >> s.ps, tp = (u * s.inv) >> 20;
    return QkfStream{s.Bk + pp * s.b2, (unsigned)((q * s.pd) * s.b2 + (s0 + 2 * tp + half) * TILE + j), 4 * s.pd * s.b2};
  };
  // What this wave does in the round that starts at unit r0 of a strip of w blocks with `units` pairs of tiles: pair r0 + wave
  // -- or, when the units left fill at most half of the waves, ONE tile of a pair (column block 2 tp or 2 tp + 1):
  // the last round of a site then takes half as long (a site of 4 x 4 tiles has 16 pairs: 12 + 4, i.e. 12 pairs + 8 tiles).
  struct Unit {
    bool mine, has1;
    int v, half;
  };
  auto unit_of = [&](const QkfSite& s, const int w, const int units, const int r0) __attribute__((always_inline)) {
    Unit un;
    const int left = units - r0;
    const bool halves = 2 * left <= NW;
    // (first tiles to waves 0 .. left - 1, second tiles to the next `left` waves: consecutive waves sit on different SIMDs)
    un.half = (halves && wave >= left) ? 1 : 0;
    un.v = r0 + wave - (un.half ? left : 0);
    const int tp = ((un.v >> s.ps) * s.inv) >> 20;
    const bool second = 2 * tp + 1 < w;  // the pair has a second tile
    un.mine = un.v < units && (un.half == 0 || second);
    un.has1 = !halves && second;
    return un;
  };
  auto a_stream = [&](const QkfSite& s, const int v) __attribute__((always_inline)) {
    const int pp = v & (s.pd - 1), u = v >> s.ps, tp = (u * s.inv) >> 20, ta = u - tp * s.mt;
    return QkfStream{s.Ak + pp * s.a2, (unsigned)(((ta * TILE + q) * s.pd) * s.a2 + j), 4 * s.pd * s.a2};
  };
  QKF_PROF_DECL();
  const int xcc = qk_xcc_id();
  int gang_round = 0;  // (gang start: this workgroup's pairs so far)
  if (tid == 0) qk_tail_start(g);
  for (;;) {
    if (tid == 0) {
      const long long pp_ = qk_pull(g, xcc);
      qk_gang_sync(g, xcc, pp_ >= 0, gang_round);
      *slot = pp_;
    }  // this XCD's queue first: the workgroups that share an L2 stream the same few states
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p < 0) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    // a unit is (ta, tp, p): pd mt ceil(nt / 2) pairs of tiles.  LDS-resident step: X and X' fit the buffer and ONE round holds
    // all pairs, or they fit side by side; otherwise X' is built in strips of W blocks of b', pairs in rounds of NW
    qkf_step_table<XCAP, NT>(g, xi, yj, rec, m_off, tid, [](const int pmt, const int nt) { return pmt * ((nt + 1) / 2) <= NW; });
    const int ek = g.edge_k, k_hi = ns - ek;  // the chain runs over the sites [ek, k_hi): the ends are in the edge blocks
    const bool edges = ek > 0;
    if (!edges)
      for (int e = tid; e < TILE * TILE; e += NT) XL[e] = (v2d){e == 0 ? 1.0 : 0.0, 0.0};
    if constexpr (DET)
      for (int e = tid; e < 2 * g.turn_ints + 2; e += NT) tbroken[e] = 0;
    int tsel = 0;  // DET: the set of turn counters in use (the other one is zeroed meanwhile for the next strip or step)
    __syncthreads();
    bool xg = false;
    int cur = 0, xb = 0;
    if (edges) {  // X behind the left edge: one product of the two left blocks
      const v4i r0 = rec[3 * ek];
      const int a_e = rfl(r0.x), b_e = rfl(r0.z);
      if (a_e * b_e <= XCAP) qkf_edge_prefix<NW>(g, xi, yj, a_e, b_e, XL, wave, q, j);
      else qkf_edge_prefix<NW>(g, xi, yj, a_e, b_e, G0, wave, q, j), xg = true;
      __syncthreads();
    }
    QKF_STAMP(0);  // pair set-up (queue, step table, left edge)
    QkfTile T0, T1;
    v2d fr[4], fs[4];     // the fragment registers of the wave's global streams (fs: the second column block of phase 1)
    bool primed = false;  // fr / fs hold the first group of the wave's next pair of tiles
    QkfSite sn = site(ek);
    for (int k = ek; k < k_hi;) {
      const QkfSite sc = sn;
      k = sc.next;  // (from here on: the entry of the NEXT step)
      if (k < k_hi) sn = site(k);
      const int a = sc.a, a2 = sc.a2, b = sc.b, mt = sc.mt, nt = sc.nt, W = sc.W;
      const bool small = sc.small;
      v2d* const Gc = G0 + (long long)cur * g.x_plane;
      v2d* const Gn = G0 + (long long)(cur ^ 1) * g.x_plane;
      if (small && xg) {
        for (int e = tid; e < a * b; e += NT) XL[e] = Gc[e];
        __syncthreads();
        xg = false, xb = 0;
      } else if (!small && !xg) {
        for (int e = tid; e < a * b; e += NT) Gc[e] = XL[xb + e];
        __syncthreads();
        xg = true;
      }
      const int n_out = sc.b2 * a2;
      const bool pingpong = small && a * b + n_out <= XCAP;
      const int ob = !small ? 0 : pingpong ? (xb == 0 ? XCAP - n_out : 0) : 0;
      if (pingpong && !DET)
        for (int e = tid; e < n_out; e += NT) XL[ob + e] = (v2d){0.0, 0.0};
      QKF_STAMP(1);  // step set-up: record decode, X moved between LDS and the global buffer, X' zeroed (ping-pong)
      for (int s0 = 0; s0 < nt; s0 += W) {
        const int w = min(W, nt - s0), units = sc.pd * mt * ((w + 1) >> 1);
        lds_int* const tcur = turn0 + tsel * g.turn_ints;
        if constexpr (DET) {  // (ordered accumulation: the first contribution to a block stores, nothing is zeroed but the next set of counters)
          for (int e = tid; e < g.turn_ints; e += NT) turn0[(tsel ^ 1) * g.turn_ints + e] = 0;
        } else if (!small) {
          for (int e = tid; e < w * TILE * a2; e += NT) XL[e] = (v2d){0.0, 0.0};
          QKF_STEP_BARRIER();
        }
        for (int r0 = 0; r0 < units; r0 += NW) {  // (LDS-resident sites: one round, or several when X and X' sit side by side)
          const Unit un = unit_of(sc, w, units, r0);
          const int v = un.v;
          const bool mine = un.mine, has1 = un.has1;
          const int u = v >> sc.ps, tp = (u * sc.inv) >> 20, ta = u - tp * mt;
          if (mine) {
            const QkfStream bs = b_stream(sc, s0, v, un.half), as = a_stream(sc, v);
            const unsigned xoff = (unsigned)(ta * (b * TILE) + q * TILE + j);
            if (xg) {
              if (has1) qkf_p1_dual<true>(T0, T1, fr, fs, primed, bs, (const v2d*)Gc, xoff, sc.nks, as);
              else qkf_p1_dual<false>(T0, T1, fr, fs, primed, bs, (const v2d*)Gc, xoff, sc.nks, as);
            } else {
              if (has1) qkf_p1_dual<true>(T0, T1, fr, fs, primed, bs, (const lds_v2d*)(XL + xb), xoff, sc.nks, as);
              else qkf_p1_dual<false>(T0, T1, fr, fs, primed, bs, (const lds_v2d*)(XL + xb), xoff, sc.nks, as);
            }
          }
          QKF_STAMP(2);  // phase 1
          if (small && r0 == 0 && !(DET && pingpong)) {
            QKF_STEP_BARRIER();  // ping-pong: X' is zero everywhere; in place (one round): every wave has read X, it becomes X'
            if (!pingpong && !DET) {
              for (int e = tid; e < n_out; e += NT) XL[e] = (v2d){0.0, 0.0};
              QKF_STEP_BARRIER();
            }
          }
          QKF_STAMP(3);  // wait for the other waves' phase 1 (LDS-resident steps), zero X'
          if (mine) {
            const int kmax = min(4, (sc.at - ta * TILE + 3) >> 2);
            // what the wave does next: its unit of the next round of this strip, of the first round of the next strip, or of
            // the next site (strip 0, round 0) -- if it has one there
            QkfStream nxt = a_stream(sc, v);
            bool np1 = false;
            if (r0 + NW < units) {
              const Unit nu = unit_of(sc, w, units, r0 + NW);
              if (nu.mine) nxt = b_stream(sc, s0, nu.v, nu.half), np1 = true;
            } else if (s0 + W < nt) {
              const int w2 = min(W, nt - s0 - W);
              const Unit nu = unit_of(sc, w2, sc.pd * mt * ((w2 + 1) >> 1), 0);
              if (nu.mine) nxt = b_stream(sc, s0 + W, nu.v, nu.half), np1 = true;
            } else if (k < k_hi) {
              const int w2 = min(sn.W, sn.nt);
              const Unit nu = unit_of(sn, w2, sn.pd * sn.mt * ((w2 + 1) >> 1), 0);
              if (nu.mine) nxt = b_stream(sn, 0, nu.v, nu.half), np1 = true;
            }
            const unsigned nd1 = 0;
            lds_v2d* const xo = XL + ob + (2 * tp + un.half) * QKF_XBLOCK;
            // the turn counters of this block of rows and this contribution's place in their order
            const QkfTurn tp_turn{tcur + (2 * tp + un.half) * sc.nn, sc.pd * ta + (v & (sc.pd - 1)), tbroken, g.err};
            if (has1) {
              if (kmax == 4) qkf_p2_dual<true, true, DET>(T0, T1, fr, fs, a_stream(sc, v), w * QKF_XBLOCK, sc.nn, 4, xo, q, j, nxt, np1, nd1, tp_turn);
              else qkf_p2_dual<false, true, DET>(T0, T1, fr, fs, a_stream(sc, v), w * QKF_XBLOCK, sc.nn, kmax, xo, q, j, nxt, np1, nd1, tp_turn);
            } else {
              if (kmax == 4) qkf_p2_dual<true, false, DET>(T0, T1, fr, fs, a_stream(sc, v), w * QKF_XBLOCK, sc.nn, 4, xo, q, j, nxt, np1, nd1, tp_turn);
              else qkf_p2_dual<false, false, DET>(T0, T1, fr, fs, a_stream(sc, v), w * QKF_XBLOCK, sc.nn, kmax, xo, q, j, nxt, np1, nd1, tp_turn);
            }
            primed = np1;
          }
          QKF_STAMP(4);  // phase 2
        }
        QKF_STEP_BARRIER();  // the strip of X' is complete
        tsel ^= 1;
        QKF_STAMP(5);  // wait for the other waves' phase 2
        if (!small && nt > W) {
          for (int tn = 0; tn < sc.nn; ++tn)  // (panel by panel: the strip's rows of a panel are contiguous in both buffers)
            for (int e = tid; e < w * QKF_XBLOCK; e += NT) Gn[(long long)tn * (sc.b2 * TILE) + s0 * QKF_XBLOCK + e] = XL[tn * w * QKF_XBLOCK + e];
          __syncthreads();
        }
      }
      if (!small) {
        if (nt > W) cur ^= 1;
        else xg = false, xb = 0;
      } else {
        xb = ob;
      }
    }
    if (edges) {  // the overlap: X against the product of the two right blocks
      const v4i r0 = rec[3 * (k_hi - 1)];
      const int a_e = rfl(r0.y), b_e = rfl(r0.w);
      if (xg) qkf_edge_suffix<NW>(g, xi, yj, a_e, b_e, (const v2d*)(G0 + (long long)cur * g.x_plane), zacc, wave, q, j);
      else qkf_edge_suffix<NW>(g, xi, yj, a_e, b_e, (const lds_v2d*)(XL + xb), zacc, wave, q, j);
      __syncthreads();
    }
    if (tid == 0) {
      v2d zz = {0.0, 0.0};
      if (edges)
        for (int w_ = 0; w_ < NW; ++w_) zz.x += zacc[2 * w_], zz.y += zacc[2 * w_ + 1];
      else zz = xg ? G0[(long long)cur * g.x_plane] : (v2d)XL[xb];
      g.values[p] = zz.x * zz.x + zz.y * zz.y;
      if (g.z) {
        g.z[2 * p] = zz.x;
        g.z[2 * p + 1] = zz.y;
      }
    }
    __syncthreads();
    QKF_STAMP(6);  // right edge, result
  }
  if (tid == 0) qk_tail_exit(g);
  QKF_PROF_FLUSH();
}

// split planes (re | im) of a set image -> interleaved complex (complex128 for double, complex64 for float), same offsets;
// one workgroup per (state, site)
template <typename T>
__global__ void qk_interleave_kernel(const T* __restrict__ src, T* __restrict__ dst, const int32_t* __restrict__ dims, const int64_t* __restrict__ offs,
                                     const int n_sites, const long long n_tensors) {
  typedef T v2 __attribute__((ext_vector_type(2)));
  for (long long t = blockIdx.x; t < n_tensors; t += gridDim.x) {
    const long long s = t / n_sites;
    const int k = (int)(t - s * n_sites);
    const long long plane = (long long)dims[s * (n_sites + 1) + k] * 2 * dims[s * (n_sites + 1) + k + 1];
    const T* re = src + offs[t];
    const T* im = re + plane;
    v2* d = reinterpret_cast<v2*>(dst + offs[t]);
    for (long long e = threadIdx.x; e < plane; e += blockDim.x) d[e] = (v2){re[e], im[e]};
  }
}

// ----------------------------------------------------------------------------------------
// Wave sweep for bonds <= 32 (fp64): ONE pair per wavefront, the whole chain in registers -- no barrier, no atomics, no
// scratch.  The generalisation of qk_sweep_wave_kernel (bonds <= 16, qk_ring.h) to 2 x 2 tiles: X is held as up to four
// A-operand tiles XA[tk][ta] (tk: block of b, ta: block of a).  Per site, for every block tb of b':
//     for each (ta, p):  T = sum_tk XA[tk][ta]^T B_k[tk rows, p, tb cols]          (one tile, 16 VGPRs, never stored)
//                        N[tn] += T^T conj(A_k[ta rows, p, tn cols])   for tn = 0, 1 (raw 3M accumulators)
//     XN[tb][tn] = combine(N[tn])                                                  (C layout = next site's XA[tk = tb][ta = tn])
// This is the regime of the 100-qubit x 10-layer config at gamma = 0.1 (bonds <= 27) and of the reference's own runs at
// gamma <= 0.5: most sites have one or two blocks per bond, so a multi-wave workgroup would leave most of its waves idle.
// What bounds it is the stream of site tensors (8 waves per CU, each reading its own pair): with plain loads right before
// use a wave has 4 KiB in flight and the launch sits at the latency of HBM (219 ms on that config).  So the fragments reach
// the wave through a private 12-KiB LDS ring filled by LDS-DMA two k-step groups ahead of the matrix instructions, the per-pair
// tables come through the scalar cache, and only the k-steps below the TRUE bond are fetched (191 -> 170 ms: the rows of zero
// padding are a third of the image at bonds around 20).
// ----------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void qkw_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NG > 0: the k-step groups (4 fragments of 1 KiB) reach the wave through a private LDS ring of NG groups filled by LDS-DMA,
// NG - 1 .. NG groups ahead of the matrix instructions (the fetch side walks the same loop nest as a small scalar state
// machine, across site boundaries).  NG = 0: plain loads into registers right before use (4 KiB in flight per wave).
// SRC = float: the set image is complex64 (interleaved), the arithmetic stays fp64 -- the sweep is bound by the bytes it
// streams, so single-precision STORAGE halves its time while the only error left is the rounding of the inputs.  A group
// is then 16 rows x 16 columns x 8 bytes = 2 KiB, fetched as one or two 1-KiB pieces (8 rows each, 16 bytes per lane).
template <int NG, typename SRC>
__global__ __launch_bounds__(64, 2) void qk_sweep_wave2_kernel(const SweepArgs g) {
  constexpr bool F32 = sizeof(SRC) == 4;
  static_assert(!F32 || NG > 0, "complex64 images go through the LDS ring");
  constexpr int ES = F32 ? 8 : 16;  // bytes of a complex element of the image (addresses below are byte addresses: pointers to a
                                    // vector of the template's scalar make hipcc drop the host-side stub of the kernel without a word)
  constexpr int GROUP = F32 ? 128 : 256;  // v2d-sized (16-byte) units of LDS per group
  __shared__ long long slot;
  __shared__ v2d ring[NG > 0 ? NG * GROUP : 1];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int lane = threadIdx.x, j = lane & 15, q = lane >> 4;
  // this lane's byte address in slot 0: fragment 0 (k-step 0) of the group
  const unsigned ring_lds = (unsigned)(uintptr_t)(lds_ptr_t)ring + (F32 ? (unsigned)(q * 128 + j * 8) : (unsigned)lane * 16u);
  const int ns = g.n_sites, n1 = ns + 1;
  const char* const xdata = reinterpret_cast<const char*>(g.xdata);
  const char* const ydata = reinterpret_cast<const char*>(g.ydata);
  auto uni = [](const int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  auto unil = [](const long long v) __attribute__((always_inline)) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  const int xcc = qk_xcc_id();
  if (lane == 0) qk_tail_start(g);
  for (;;) {
    if (lane == 0) slot = qk_pull(g, xcc);
    __syncthreads();
    const long long p = unil(slot);
    __syncthreads();
    if (p < 0) break;
    // per-pair tables through the scalar cache (wave-uniform addresses in the constant address space): a vector load
    // per site and table would put its whole latency in front of the site's first tensor load
    typedef const __attribute__((address_space(4))) int* sint_p;
    typedef const __attribute__((address_space(4))) int64_t* slong_p;
    const int xi = ((sint_p)g.pairs)[2 * p], yj = ((sint_p)g.pairs)[2 * p + 1];
    const sint_p xd = (sint_p)(g.xdims + (long long)xi * n1);
    const sint_p yd = (sint_p)(g.ydims + (long long)yj * n1);
    const sint_p xt = (sint_p)(g.xtrue + (long long)xi * n1);
    const sint_p yt = (sint_p)(g.ytrue + (long long)yj * n1);
    const slong_p xo = (slong_p)(g.xoffs + (long long)xi * ns);
    const slong_p yo = (slong_p)(g.yoffs + (long long)yj * ns);
    QkfTile XA[2][2], XN[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) XA[u][v].re = XA[u][v].im = XN[u][v].re = XN[u][v].im = (v4d){0, 0, 0, 0};
    XA[0][0].re[0] = (lane == 0) ? 1.0 : 0.0;  // X_0 = 1: A-operand element [k = 0][m = 0]
    // ---- fetch side (NG > 0): group order = for tb, ta, p: { P1 groups tk < kb, P2 groups tn < nn }, site after site.
    // Entering a site, lane g computes the descriptor of the site's group g (<= 32 groups): element offset in its tensor,
    // k-steps below the true bond, which tensor -- a dozen vector instructions for the whole site; fetching a group is then a
    // v_readlane and the LDS-DMAs, with no branch but "site finished".  (The first version walked the loop nest as nested
    // scalar ifs and picked the wait with a switch: ~250 scalar instructions and ~30 branches between two groups of 3-12
    // MFMAs -- the wave was bound by its own control code.)
    int f_k = 0, f_i = 0, f_n = 0, f_slot = 0, c_slot = 0;
    int f_a2 = 0, f_b2 = 0;
    const char* f_A = xdata;
    const char* f_B = ydata;
    int desc = 0;
    auto enter = [&]() __attribute__((always_inline)) {  // descriptors of site f_k
      // (pinned to scalar registers: left alone, the compiler folds the two true-bond loads and the per-lane select below into
      //  ONE per-lane vector load -- and the s_waitcnt vmcnt(0) behind it drains the LDS-DMAs in flight at every site)
      const int fa = xd[f_k], fb = yd[f_k], fat = __builtin_amdgcn_readfirstlane(xt[f_k]), fbt = __builtin_amdgcn_readfirstlane(yt[f_k]);
      f_a2 = xd[f_k + 1], f_b2 = yd[f_k + 1];
      f_A = xdata + (xo[f_k] >> 1) * ES, f_B = ydata + (yo[f_k] >> 1) * ES;
      const int kb = fb >> 4, nn = f_a2 >> 4, mt = fa >> 4, nt = f_b2 >> 4, gpb = kb + nn;
      const int inv = (gpb == 2) ? 32768 : (gpb == 3) ? 21846 : 16384;  // g / gpb == (g * inv) >> 16 for g < 64
      const int gi = lane, blk = (gi * inv) >> 16, h = gi - blk * gpb, pp = blk & 1, r = blk >> 1;
      const int ta = (mt == 2) ? (r & 1) : 0, tb = (mt == 2) ? (r >> 1) : r;
      const bool is_b = h < kb;
      const int off = is_b ? ((h * TILE) * 2 + pp) * f_b2 + tb * TILE : ((ta * TILE) * 2 + pp) * f_a2 + (h - kb) * TILE;
      const int cnt = is_b ? min(4, (fbt - h * TILE + 3) >> 2) : min(4, (fat - ta * TILE + 3) >> 2);
      desc = off | (cnt << 16) | ((is_b ? 1 : 0) << 20);
      f_n = nt * mt * 2 * gpb, f_i = 0;
    };
    // fetch the next group: its k-steps below the true bond (the rows above are zero padding: a third of the image at bonds
    // around 20); the pieces above are asked for again at the address of the last one needed -- an L1 hit, no fabric bytes --
    // so that every group is the same number of LDS-DMAs and the consumer's wait is a constant.
    auto issue = [&]() __attribute__((always_inline)) {
      if (f_k >= ns) return;
      const int d = __builtin_amdgcn_readlane(desc, f_i);
      const int off = d & 0xffff, cnt = (d >> 16) & 7;
      const bool is_b = (d >> 20) != 0;
      const int ld = is_b ? f_b2 : f_a2;
      const char* const base = (is_b ? f_B : f_A) + (long)off * ES;
      if constexpr (F32) {
        // piece p = rows 8 p .. 8 p + 7 (two k-steps): lane L brings the 16 bytes of columns 2 (L & 7), + 1 of row 8 p + (L >> 3)
        const char* const src = base + (((lane >> 3) * 2) * ld + (lane & 7) * 2) * ES;
        const int last = (cnt - 1) >> 1;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) if (!(QKF_ABL & 16)) __builtin_amdgcn_global_load_lds(src + (min(pc, last) * 16 * ld) * ES, (lds_ptr_t)(ring + f_slot * GROUP + pc * 64), 16, 0, 0);
      } else {
        const char* const src = base + ((q * 2) * ld + j) * ES;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (!(QKF_ABL & 16)) __builtin_amdgcn_global_load_lds(src + (min(i, cnt - 1) * 8 * ld) * ES, (lds_ptr_t)(ring + (f_slot * 4 + i) * 64), 16, 0, 0);
      }
      f_slot = (f_slot == NG - 1) ? 0 : f_slot + 1;
      if (++f_i == f_n) {
        if (++f_k < ns) enter();
      }
    };
    // the next group of the order above, as four fragments in registers (those above the true bond hold copies and are not used)
    auto take = [&](v2d(&f)[4]) __attribute__((always_inline)) {
      issue();
      if (f_k >= ns) qkw_wait_vmcnt<0>();  // the last groups of a chain: nothing more is fetched behind them
      else qkw_wait_vmcnt<(NG > 0 ? (F32 ? 2 : 4) * (NG - 1) : 0)>();
      // (read by hand: the compiler puts vmcnt(0) in front of every LDS read it can see next to an LDS-DMA)
      const unsigned at = ring_lds + (unsigned)c_slot * (unsigned)(GROUP * 16);
      if constexpr (F32) {  // k-step i of lane (q, j): row q + 4 i, column j of the 16 x 16 image (128-byte rows)
        double h[4];  // (two floats each)
        asm volatile("ds_read_b64 %0, %1" : "=v"(h[0]) : "v"(at));
        asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(h[1]) : "v"(at));
        asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(h[2]) : "v"(at));
        asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(h[3]) : "v"(at));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3])::"memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long long bits = __double_as_longlong(h[i]);
          f[i] = (v2d){(double)__int_as_float((int)bits), (double)__int_as_float((int)(bits >> 32))};
        }
      } else if (QKF_ABL & 32) {
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(at));
      } else {
        asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(at));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(f[1]) : "v"(at));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f[2]) : "v"(at));
        asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(f[3]) : "v"(at));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
      }
      c_slot = (c_slot == NG - 1) ? 0 : c_slot + 1;
    };
    if constexpr (NG > 0) {
      enter();
#pragma unroll
      for (int i = 0; i < NG - 1; ++i) issue();
    }
    // the tables of site k + 1 are fetched while site k is swept
    int a_nx = xd[0], b_nx = yd[0], a2_nx = xd[1], b2_nx = yd[1], at_nx = xt[0], bt_nx = yt[0];
    long long xo_nx = xo[0], yo_nx = yo[0];
    for (int k = 0; k < ns; ++k) {
      const int a = a_nx, a2 = a2_nx, b = b_nx, b2 = b2_nx, at = at_nx, bt = bt_nx;
      const v2d* const Ak = reinterpret_cast<const v2d*>(xdata) + (xo_nx >> 1);  // (the plain-load path: complex128 images only)
      const v2d* const Bk = reinterpret_cast<const v2d*>(ydata) + (yo_nx >> 1);
      {
        const int k1 = min(k + 1, ns - 1);
        a_nx = a2, b_nx = b2, a2_nx = xd[k1 + 1], b2_nx = yd[k1 + 1], at_nx = xt[k1], bt_nx = yt[k1];
        xo_nx = xo[k1], yo_nx = yo[k1];
      }
      const int mt = a >> 4, kb = b >> 4, nn = a2 >> 4, nt = b2 >> 4;  // blocks of a, b, a', b' (1 or 2 each)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        if (tb < nt) {
          v4d n1a[2], n2a[2], n3a[2];  // raw accumulators of X'[tb][tn]
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) n1a[tn] = n2a[tn] = n3a[tn] = (v4d){0, 0, 0, 0};
#pragma unroll
          for (int ta = 0; ta < 2; ++ta) {
            if (ta < mt) {
              const int ka = min(4, (at - ta * TILE + 3) >> 2);  // k-steps of this block of a below the true bond
#pragma unroll
              for (int pp = 0; pp < 2; ++pp) {
                // ---- T = sum_tk XA[tk][ta]^T B[tk rows, pp, tb cols]
                v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
#pragma unroll
                for (int tk = 0; tk < 2; ++tk) {
                  if (tk < kb) {
                    const int kk = min(4, (bt - tk * TILE + 3) >> 2);
                    const v2d* const bp = Bk + ((tk * TILE + q) * 2 + pp) * b2 + tb * TILE + j;
                    v2d f[4];
                    if constexpr (NG > 0) take(f);
                    else {
#pragma unroll
                      for (int i = 0; i < 4; ++i) f[i] = bp[i * 8 * b2];  // (rows up to the padded bond exist and are zero)
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                      if (i < kk) qkf_kstep<false>(p1, p2, p3, XA[tk][ta].re[i], XA[tk][ta].im[i], f[i].x, f[i].y);
                  }
                }
                QkfTile t;
                if (QKF_ABL & 64) t.re = p1, t.im = p3;
                else t.re = p1 - p2, t.im = p3 - p1 - p2;
                // ---- N[tn] += T^T conj(A[ta rows, pp, tn cols])
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                  if (tn < nn) {
                    const v2d* const ap = Ak + ((ta * TILE + q) * 2 + pp) * a2 + tn * TILE + j;
                    v2d f[4];
                    if constexpr (NG > 0) take(f);
                    else {
#pragma unroll
                      for (int i = 0; i < 4; ++i) f[i] = ap[i * 8 * a2];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                      if (i < ka) qkf_kstep<true>(n1a[tn], n2a[tn], n3a[tn], t.re[i], t.im[i], f[i].x, f[i].y);
                  }
                }
              }
            }
          }
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) XN[tb][tn].re = n1a[tn] + n2a[tn], XN[tb][tn].im = n3a[tn] - n1a[tn] + n2a[tn];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) XA[u][v] = XN[u][v];
    }
    {
      // z = X_n[0][0] sits in lane 0; broadcast and stored by every lane (see qk_sweep_wave_kernel for why)
      const double re = __longlong_as_double(unil(__double_as_longlong(XA[0][0].re[0])));
      const double im = __longlong_as_double(unil(__double_as_longlong(XA[0][0].im[0])));
      g.values[p] = re * re + im * im;
      if (g.z) {
        g.z[2 * p] = re;
        g.z[2 * p + 1] = im;
      }
    }
  }
  if (lane == 0) qk_tail_exit(g);
}
