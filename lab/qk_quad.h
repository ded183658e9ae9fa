// qk_quad.h -- the QUAD form of the site-fused sweep (qk_sweep_fused_quad_kernel): 2 x 2 tiles per wave.
//
// Same chain, same tables, same LDS image of X as qk_fused.h (read that header first); what changes is the unit of work of a wave:
// the four tiles  T[ta0 | ta0 + 1,  p,  tb0 | tb0 + 1]  -- two neighbouring row blocks of a, two neighbouring column blocks of b'.
//   Phase 1: a k-step reads two fragments of X (LDS) and two of B (global) and issues 12 matrix instructions: half the global loads
//            and half the B-side operand sums per matrix instruction of the dual form (one fragment of X, two of B per 6);
//   Phase 2: the two row blocks of a are consecutive K blocks of the SAME products  X'[tb rows, tn cols] += T^T conj(A):  a column block
//            is 8 k-steps into 6 accumulators, so the additions behind a block and the LDS adds come once per 48 matrix instructions
//            (dual form: once per 24).
// Why: the ablations of the dual kernel on the 60-qubit x 6-layer set (profiles/r04/ablations.txt) price the global loads inside
// the loops at 11.7 % of its time, the operand sums at 5.3 %, the tails of phase 2 at 3.6 % -- all three are per-fragment or per-block
// costs, and a 2 x 2 unit has a third fewer fragments and half the blocks per matrix instruction.  It needs ~200 VGPRs: 8 waves per
// workgroup, two per SIMD (the dual form as 8 waves ties its 12-wave shape: lab/NOTES_r04.md).
// STATUS: a lab kernel (built only with -DQKF_QUAD=1, lab/libqkgram_quad.so).  Parity-green (fuzz against the oracle 2.7e-15), and 2 % SLOWER than
// the shipped dual form on cfg4 (365.1 against 357.9 ms, same box; the dual form as 8 waves: 361.4): whole quads gain 7 - 13 ms over the same kernel
// dealing column pairs only, but that kernel is 10 - 22 ms behind the 8-wave dual form it ought to equal (lab/NOTES_r04.md).
// A round deals whole quads while at least half of the waves get one; the remainder of a strip is cut into column pairs (one row block
// each: two waves per quad) or single tiles (four waves per quad), so that the last round of a site is as short as its tiles allow.
// fp64, arrival-order accumulation only (QK_DETERMINISTIC=1 stays on the DET forms of qk_fused.h).
#pragma once
// (included by csrc/qkgram.hip behind qk_fused.h, whose streams, tiles, tables and edge products it uses)

// ceil(2^20 / d) for d = 0 .. 16 (0 for d = 0): wave-uniform index, so the look-up is a scalar load
__device__ __forceinline__ int qkq_qinv(const int d) {
  static constexpr int T[17] = {0, 1048576, 524288, 349526, 262144, 209716, 174763, 149797, 131072, 116509, 104858, 95326, 87382, 80660, 74899, 69906, 65536};
  return T[d];
}

struct QkqUnit {
  bool mine, a1, b1;  // this wave has work; the unit has a second row block / a second column block
  int ta, tb, p;      // first row block of a, first column block (strip-local), physical index
};

// Phase 1 of a quad.  B operand: stream `cur` (column block tb; tb + 1 is TILE elements further); A operand: X elements (xp + x0 + 64 i)
// for row block ta and (xp + x1 + 64 i) for ta + 1 (panels of 16 columns, see qk_fused.h).  fr / fs carry the first group of the two B
// column blocks in, and the first groups of the unit's two phase-2 streams (n0: row block ta, n1: ta + 1) out.
template <bool A1, bool B1, typename XPtr>
__device__ __forceinline__ void qkq_p1(QkfTile (&T)[2][2], v2d (&fr)[4], v2d (&fs)[4], const bool primed, QkfStream cur, XPtr xp, unsigned x0, unsigned x1, const int nks,
                                       const QkfStream n0, const unsigned n1off) {
  v4d P[2][2][3];
  v2d fx[4], fy[4];
  if (!primed) qkf_load4(fr, cur);
  if (B1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fs[i] = qkf_ldg(cur.base + i * cur.step, cur.off + TILE);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fx[i] = qkf_ldx(xp + i * QKF_XSTEP, x0);
    if (A1) fy[i] = qkf_ldx(xp + i * QKF_XSTEP, x1);
  }
  const int ng = (nks + 3) >> 2, last = nks - 4 * (ng - 1);  // k-steps of the last group: 1..4
  auto kstep = [&](const int i, const bool first) __attribute__((always_inline)) {
    const double sb0 = fr[i].x + fr[i].y, sb1 = B1 ? fs[i].x + fs[i].y : 0.0;
    const double sa0 = fx[i].x + fx[i].y, sa1 = A1 ? fy[i].x + fy[i].y : 0.0;
    const v4d z = {0, 0, 0, 0};
    auto mm = [&](v4d(&acc)[3], const double ar, const double ai, const double sa, const double br, const double bi, const double sb) __attribute__((always_inline)) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, first ? z : acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, first ? z : acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, sb, first ? z : acc[2], 0, 0, 0);
    };
    mm(P[0][0], fx[i].x, fx[i].y, sa0, fr[i].x, fr[i].y, sb0);
    if (B1) mm(P[0][1], fx[i].x, fx[i].y, sa0, fs[i].x, fs[i].y, sb1);
    if (A1) mm(P[1][0], fy[i].x, fy[i].y, sa1, fr[i].x, fr[i].y, sb0);
    if (A1 && B1) mm(P[1][1], fy[i].x, fy[i].y, sa1, fs[i].x, fs[i].y, sb1);
  };
  auto reload = [&](const int i) __attribute__((always_inline)) {
    fr[i] = qkf_ldg(cur.base + i * cur.step, cur.off);
    if (B1) fs[i] = qkf_ldg(cur.base + i * cur.step, cur.off + TILE);
    fx[i] = qkf_ldx(xp + i * QKF_XSTEP, x0);
    if (A1) fy[i] = qkf_ldx(xp + i * QKF_XSTEP, x1);
  };
  QKF_PRIO_LO();
  int gq = 0;
  if (ng >= 2) {  // the first k-step of a chain of more than four STARTS the accumulators (literal zero as the C operand)
    gq = 1;
    cur.off += 4 * cur.step, x0 += 4 * QKF_XSTEP, x1 += 4 * QKF_XSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kstep(i, i == 0);
      reload(i);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int e = 0; e < 3; ++e) P[u][v][e] = (v4d){0, 0, 0, 0};
  }
#pragma unroll 1
  for (; gq + 1 < ng; ++gq) {
    cur.off += 4 * cur.step, x0 += 4 * QKF_XSTEP, x1 += 4 * QKF_XSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kstep(i, false);
      reload(i);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < last) kstep(i, false);
    fr[i] = qkf_ldg(n0.base + i * n0.step, n0.off);  // the first groups of this unit's phase 2: row block ta ...
    if (A1) fs[i] = qkf_ldg(n0.base + i * n0.step, n1off);  // ... and ta + 1
    __builtin_amdgcn_sched_barrier(0);
  }
  QKF_PRIO_HI();
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
      if ((u == 0 || A1) && (v == 0 || B1)) T[u][v].re = P[u][v][0] - P[u][v][1], T[u][v].im = P[u][v][2] - P[u][v][0] - P[u][v][1];
}

// One column block tn of phase 2 of a quad:  X'[tb rows | tb + 1 rows, tn cols] += sum over the unit's row blocks of  T^T conj(A[rows, p, 16 tn + .]).
// fr / fs hold the four k-steps of row block ta / ta + 1; each is reloaded right behind its matrix instructions from (b_i, o0) / (b_i, o1)
// -- the next column block of the same streams, or (LAST) fr alone from the wave's next phase-1 stream.  The row block that may be ragged
// is the unit's last one (kmax of its k-steps lie below the true bond).  3M product in the form of qkf_p2_block.
template <bool FULL, bool A1, bool B1, bool LAST>
__device__ __forceinline__ void qkq_p2_block(const QkfTile (&T)[2][2], const v4d (&S)[2][2], v2d (&fr)[4], v2d (&fs)[4], const int kmax, const v2d* const b0, const v2d* const b1,
                                             const v2d* const b2, const v2d* const b3, const unsigned o0, const unsigned o1, lds_double* const d, lds_double* const d1, const long rs) {
  v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};
  QKF_PRIO_LO();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (A1 || FULL || i < kmax) {
      const double sp = fr[i].x + fr[i].y, sm = fr[i].x - fr[i].y;
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0][0][i], fr[i].x, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[0][0].re[i], sp, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[0][0].im[i], sm, p3, 0, 0, 0);
      if (B1) {
        r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0][1][i], fr[i].x, r1, 0, 0, 0);
        r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[0][1].re[i], sp, r2, 0, 0, 0);
        r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[0][1].im[i], sm, r3, 0, 0, 0);
      }
    }
    fr[i] = qkf_ldg(i == 0 ? b0 : i == 1 ? b1 : i == 2 ? b2 : b3, o0);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (A1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (FULL || i < kmax) {
        const double sp = fs[i].x + fs[i].y, sm = fs[i].x - fs[i].y;
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1][0][i], fs[i].x, p1, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[1][0].re[i], sp, p2, 0, 0, 0);
        p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[1][0].im[i], sm, p3, 0, 0, 0);
        if (B1) {
          r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1][1][i], fs[i].x, r1, 0, 0, 0);
          r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[1][1].re[i], sp, r2, 0, 0, 0);
          r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[1][1].im[i], sm, r3, 0, 0, 0);
        }
      }
      if (!LAST) fs[i] = qkf_ldg(i == 0 ? b0 : i == 1 ? b1 : i == 2 ? b2 : b3, o1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  QKF_PRIO_HI();
  {
    const v4d re = p1 - p3, im = p1 - p2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __hip_atomic_fetch_add(d + r * rs, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d + r * rs + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (B1) {
    const v4d re = r1 - r3, im = r1 - r2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __hip_atomic_fetch_add(d1 + r * rs, re[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(d1 + r * rs + 1, im[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// Phase 2 of a quad: all column blocks of a'.  `cur` = the stream of row block ta (ta + 1: a1off elements further), `nxt` = the wave's next
// phase-1 stream (or a dummy: its own stream again), `ps` = elements per panel of X' (the strip's rows x 16), xo = the unit's first 16 rows of X'.
template <bool FULL, bool A1, bool B1>
__device__ __forceinline__ void qkq_p2(const QkfTile (&T)[2][2], v2d (&fr)[4], v2d (&fs)[4], const QkfStream cur, const unsigned a1off, const int ps, const int nn, const int kmax, lds_v2d* xo,
                                       const int q, const int j, const QkfStream nxt) {
  lds_double* d = (lds_double*)(xo + q * TILE + j);
  lds_double* d1 = d + 2 * QKF_XBLOCK;  // the second column block's rows of X': 16 rows further down
  constexpr long rs = 2 * QKF_XSTEP;
  v4d S[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) S[u][v] = ((u == 0 || A1) && (v == 0 || B1)) ? T[u][v].re + T[u][v].im : (v4d){0, 0, 0, 0};
  const v2d *const c0 = cur.base, *const c1 = cur.base + cur.step, *const c2 = cur.base + 2 * cur.step, *const c3 = cur.base + 3 * cur.step;
  unsigned o0 = cur.off, o1 = cur.off + a1off;
#pragma unroll 1
  for (int tn = 0; tn + 1 < nn; ++tn) {
    o0 += TILE, o1 += TILE;
    qkq_p2_block<FULL, A1, B1, false>(T, S, fr, fs, kmax, c0, c1, c2, c3, o0, o1, d, d1, rs);
    d += 2 * ps, d1 += 2 * ps;  // the next panel
  }
  qkq_p2_block<FULL, A1, B1, true>(T, S, fr, fs, kmax, nxt.base, nxt.base + nxt.step, nxt.base + 2 * nxt.step, nxt.base + 3 * nxt.step, nxt.off, 0u, d, d1, rs);
}

template <int NW, int XCAP, int WPS>  // waves per workgroup (a round holds NW quads); elements of the LDS X buffer; waves per SIMD (register budget)
__global__ __launch_bounds__(64 * NW, WPS) void qk_sweep_fused_quad_kernel(const SweepArgs g) {
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
  // The PIECES of a strip of w column blocks, in the order they are dealt (piece r0 + wave to wave `wave` in the round that starts at r0):
  //   1. whole quads -- as many as fill complete rounds (kq = a multiple of NW): every wave is busy for four tiles, nobody waits;
  //   2. column pairs (1 x 2 tiles): the two row blocks of each quad left over, then the pairs of the odd last row block;
  //      row pairs (2 x 1 tiles) of the odd last column block;
  //   3. the corner tiles (odd mt and odd w).
  // Quads are numbered  vq = pd (qb qa_f + qa) + p  (qa < qa_f = mt / 2 pairs of row blocks, qb < qb_f = w / 2 pairs of column blocks).
  // In the LAST round, when the pieces left fill at most half of the waves, each two-tile piece is cut into its tiles (wave / left = which).
  // (Measured against this order, cfg4: the pieces dealt column pair by column pair, quads and the odd row block's pairs mixed in a round,
  //  so that the waves of a round share their columns of B as in the dual form: 378 against 368 ms.)
  struct Pieces {
    int qa_f, qb_f, ra, rb, nA, kq, n2a, nB, nC, nD, L, qinv;
  };
  auto pieces_of = [&](const QkfSite& s, const int w) __attribute__((always_inline)) {
    Pieces pc;
    pc.qa_f = s.mt >> 1, pc.qb_f = w >> 1, pc.ra = s.mt & 1, pc.rb = w & 1;
    pc.nA = s.pd * pc.qa_f * pc.qb_f;
#if defined(QKF_QUAD_MODE) && QKF_QUAD_MODE == 0  // experiment: no whole quads at all (the kernel as an 8-wave dual form)
    pc.kq = 0;
#else
    pc.kq = pc.nA - pc.nA % NW;
#endif
    pc.n2a = 2 * (pc.nA - pc.kq);
    pc.nB = s.pd * pc.ra * pc.qb_f, pc.nC = s.pd * pc.qa_f * pc.rb, pc.nD = s.pd * pc.ra * pc.rb;
    pc.L = pc.kq + pc.n2a + pc.nB + pc.nC + pc.nD;
    pc.qinv = qkq_qinv(pc.qa_f);  // u / qa_f == (u * qinv) >> 20 (exact for u < 2048, qa_f <= 16); from a table: an integer division here costs ~40 instructions per call
    return pc;
  };
  auto unit_of = [&](const QkfSite& s, const int w, const Pieces& pc, const int r0) __attribute__((always_inline)) {
    QkqUnit un;
    const int left = pc.L - r0;
    const bool cut = 2 * left <= NW;  // (then this is the last round)
    const int sub = cut && wave >= left ? 1 : 0;
    int i = r0 + wave - sub * left;
    un.mine = cut ? wave < 2 * left : i < pc.L;
    un.a1 = un.b1 = false, un.ta = un.tb = un.p = 0;
    auto quad = [&](const int vq) __attribute__((always_inline)) {
      const int u = vq >> s.ps, qb = (u * pc.qinv) >> 20, qa = u - qb * pc.qa_f;
      un.p = vq & (s.pd - 1), un.ta = 2 * qa, un.tb = 2 * qb;
    };
    if (i < pc.kq) {  // a whole quad (never cut: kq fills complete rounds)
      quad(i), un.a1 = un.b1 = true;
    } else if ((i -= pc.kq) < pc.n2a) {  // one row block of a left-over quad
      quad(pc.kq + (i >> 1)), un.ta += i & 1, un.b1 = true;
    } else if ((i -= pc.n2a) < pc.nB) {  // the odd last row block: pairs of column blocks
      un.p = i & (s.pd - 1), un.ta = s.mt - 1, un.tb = 2 * (i >> s.ps), un.b1 = true;
    } else if ((i -= pc.nB) < pc.nC) {  // the odd last column block: pairs of row blocks
      un.p = i & (s.pd - 1), un.ta = 2 * (i >> s.ps), un.tb = w - 1, un.a1 = true;
    } else {  // the corner
      i -= pc.nC;
      un.p = i & (s.pd - 1), un.ta = s.mt - 1, un.tb = w - 1;
    }
    if (cut) {  // the last round, with waves to spare: tile `sub` of a two-tile piece (a single tile goes to sub 0 alone)
      if (un.b1) un.tb += sub, un.b1 = false;
      else if (un.a1) un.ta += sub, un.a1 = false;
      else un.mine = un.mine && sub == 0;
    }
    return un;
  };
  auto b_stream = [&](const QkfSite& s, const int s0, const QkqUnit& un) __attribute__((always_inline)) {
    return QkfStream{s.Bk + un.p * s.b2, (unsigned)((q * s.pd) * s.b2 + (s0 + un.tb) * TILE + j), 4 * s.pd * s.b2};
  };
  auto a_stream = [&](const QkfSite& s, const QkqUnit& un) __attribute__((always_inline)) {
    return QkfStream{s.Ak + un.p * s.a2, (unsigned)(((un.ta * TILE + q) * s.pd) * s.a2 + j), 4 * s.pd * s.a2};
  };
  const int xcc = qk_xcc_id();
  if (tid == 0) qk_tail_start(g);
  for (;;) {
    if (tid == 0) *slot = qk_pull(g, xcc);
    __syncthreads();
    const long long p = *slot;
    __syncthreads();
    if (p < 0) break;
    const int xi = g.pairs[2 * p], yj = g.pairs[2 * p + 1];
    // LDS-resident step: X and X' fit the buffer and ONE round holds all quads (X' may then overwrite X), or they fit side by side;
    // otherwise X' is built in strips of W blocks of b'.  (one round: no whole quad is dealt then, and the pieces are at most pd mt ceil(nt / 2))
    qkf_step_table<XCAP, NT>(g, xi, yj, rec, m_off, tid, [](const int pmt, const int nt) { return pmt * ((nt + 1) / 2) <= NW; });
    const int ek = g.edge_k, k_hi = ns - ek;
    const bool edges = ek > 0;
    if (!edges)
      for (int e = tid; e < TILE * TILE; e += NT) XL[e] = (v2d){e == 0 ? 1.0 : 0.0, 0.0};
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
    v2d fr[4], fs[4];
    bool primed = false;  // fr holds the first group of the wave's next phase-1 stream
    QkfSite sn = site(ek);
    for (int k = ek; k < k_hi;) {
      const QkfSite sc = sn;
      k = sc.next;  // (from here on: the entry of the NEXT step)
      if (k < k_hi) sn = site(k);
      const int a = sc.a, a2 = sc.a2, b = sc.b, nt = sc.nt, W = sc.W;
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
      if (pingpong)
        for (int e = tid; e < n_out; e += NT) XL[ob + e] = (v2d){0.0, 0.0};
      for (int s0 = 0; s0 < nt; s0 += W) {
        const int w = min(W, nt - s0);
        const Pieces pc = pieces_of(sc, w);
        const int quads = pc.L;
        if (!small) {
          for (int e = tid; e < w * TILE * a2; e += NT) XL[e] = (v2d){0.0, 0.0};
          qk_lds_barrier();
        }
        for (int r0 = 0; r0 < quads; r0 += NW) {  // (LDS-resident steps: one round, or several when X and X' sit side by side)
          const QkqUnit un = unit_of(sc, w, pc, r0);
          const QkfStream as = a_stream(sc, un);
          const unsigned a1off = (unsigned)(TILE * sc.pd * a2);  // the stream of row block ta + 1: 16 rows of [pd][a2] further
          // between the phases of the first round of an LDS-resident step (every wave of the workgroup passes here, with or without a unit)
          auto between = [&]() __attribute__((always_inline)) {
            if (small && r0 == 0) {
              qk_lds_barrier();  // ping-pong: X' is zero everywhere; in place (one round): every wave has read X, it becomes X'
              if (!pingpong) {
                for (int e = tid; e < n_out; e += NT) XL[e] = (v2d){0.0, 0.0};
                qk_lds_barrier();
              }
            }
          };
          if (un.mine) {
            QkfTile T[2][2];  // (declared here: the tiles are dead between rounds)
            const QkfStream bs = b_stream(sc, s0, un);
            const unsigned x0 = (unsigned)(un.ta * (b * TILE) + q * TILE + j), x1 = x0 + (unsigned)(b * TILE);
            auto go = [&](auto xbase) __attribute__((always_inline)) {
              if (un.a1) {
                if (un.b1) qkq_p1<true, true>(T, fr, fs, primed, bs, xbase, x0, x1, sc.nks, as, as.off + a1off);
                else qkq_p1<true, false>(T, fr, fs, primed, bs, xbase, x0, x1, sc.nks, as, as.off + a1off);
              } else {
                if (un.b1) qkq_p1<false, true>(T, fr, fs, primed, bs, xbase, x0, x1, sc.nks, as, as.off + a1off);
                else qkq_p1<false, false>(T, fr, fs, primed, bs, xbase, x0, x1, sc.nks, as, as.off + a1off);
              }
            };
            if (xg) go((const v2d*)Gc);
            else go((const lds_v2d*)(XL + xb));
            between();
            // the row block that may be ragged is the unit's last one
            const int kmax = min(4, (sc.at - (un.ta + (un.a1 ? 1 : 0)) * TILE + 3) >> 2);
            // what the wave does next: its unit of the next round of this strip, of the first round of the next strip, or of the next
            // step (strip 0, round 0) -- if it has one there
            QkfStream nxt = as;
            bool np1 = false;
            if (r0 + NW < quads) {
              const QkqUnit nu = unit_of(sc, w, pc, r0 + NW);
              if (nu.mine) nxt = b_stream(sc, s0, nu), np1 = true;
            } else if (s0 + W < nt) {
              const int w2 = min(W, nt - s0 - W);
              const QkqUnit nu = unit_of(sc, w2, pieces_of(sc, w2), 0);
              if (nu.mine) nxt = b_stream(sc, s0 + W, nu), np1 = true;
            } else if (k < k_hi) {
              const int w2 = min(sn.W, sn.nt);
              const QkqUnit nu = unit_of(sn, w2, pieces_of(sn, w2), 0);
              if (nu.mine) nxt = b_stream(sn, 0, nu), np1 = true;
            }
            lds_v2d* const xo = XL + ob + un.tb * QKF_XBLOCK;
            const int ps = w * QKF_XBLOCK;
            // (FULL: every k-step of the unit's last row block lies below the true bond -- the common case gets matrix blocks without a branch;
            //  the ragged case is one instantiation per shape with the k-steps tested at run time)
            auto p2 = [&](auto a1_, auto b1_) __attribute__((always_inline)) {
              constexpr bool A1_ = decltype(a1_)::value, B1_ = decltype(b1_)::value;
              if (kmax == 4) qkq_p2<true, A1_, B1_>(T, fr, fs, as, a1off, ps, sc.nn, 4, xo, q, j, nxt);
              else qkq_p2<false, A1_, B1_>(T, fr, fs, as, a1off, ps, sc.nn, kmax, xo, q, j, nxt);
            };
            if (un.a1) {
              if (un.b1) p2(std::true_type{}, std::true_type{});
              else p2(std::true_type{}, std::false_type{});
            } else {
              if (un.b1) p2(std::false_type{}, std::true_type{});
              else p2(std::false_type{}, std::false_type{});
            }
            primed = np1;
          } else {
            between();
          }
        }
        qk_lds_barrier();  // the strip of X' is complete
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
  }
  if (tid == 0) qk_tail_exit(g);
}
