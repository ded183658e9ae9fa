// qk_build_kernels.h -- the device code of the MPS builder (csrc/qk_build.hip), compiled once per workgroup size:
//   namespace qkb256: 256 threads (4 wavefronts), two or four workgroups per CU -- small and medium bonds, hundreds of states in flight;
//   namespace qkb512: 512 threads (8 wavefronts), ONE workgroup per CU with 152 KiB of LDS -- large bonds: the block factorisation
//                     runs one visit per wavefront, so eight of them are in flight and two wavefronts share a SIMD (one's rotations
//                     in LDS overlap the other's matrix instructions); the launch of a heterogeneous data set ends with its few
//                     heaviest states, and their time falls with the wavefronts they are given.
// QKB_NS and QK_BUILD_BT are defined by the including file.
namespace QKB_NS {
constexpr int BT = QK_BUILD_BT;  // threads per workgroup
#ifndef QK_BUILD_GL
#define QK_BUILD_GL 8
#endif
constexpr int GL = QK_BUILD_GL;  // lanes that share one column pair
constexpr int NG = BT / GL;   // column pairs per step

__device__ __forceinline__ cd cmul(const cd a, const cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd cfma(const cd a, const cd b, const cd c) { return cd{c.x + a.x * b.x - a.y * b.y, c.y + a.x * b.y + a.y * b.x}; }

// Sum over the GL lanes of a pair group with DPP lane permutations (quad swaps, then half-row / row mirrors): a handful of
// VALU moves instead of the LDS round trip of a ds_bpermute per 32-bit half and stage.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group_sum(double v) {
  static_assert(GL == 8 || GL == 16, "group_sum is written for 8 or 16 lanes per pair");
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_f64<0x141>(v);  // row_half_mirror: the other quad of the 8
  if (GL == 16) v += dpp_f64<0x140>(v);  // row_mirror: the other half of the 16
  return v;
}

// One-sided Jacobi.  A is p x q, element (i, j) at A[i * rs + j * cs].  On return A <- A V with mutually orthogonal columns and V
// (q x q, element (i, c) at V[i * vrs + c * vcs]) holds the accumulated unitary; sig[j] = |column j|, ord = column indices by decreasing sig.
typedef __attribute__((address_space(3))) cd* lds_cd_ptr;  // LDS-typed: ds_read/ds_write instead of flat accesses
// R > 0: every lane keeps its rows of the two columns in registers (p, q <= R * GL): all loads of a step are issued at
// once and the rotation does not read the columns a second time; R = 0 is the general loop.
template <typename P, int R = 0>  // P = cd* (L2-resident workspace) or lds_cd_ptr
__device__ void jacobi_orth(P A, const long rs, const long cs, const int p, const int q, P V, const int vrs, const int vcs,
                            double* sig, int* ord, WgShared* sh, int* error, const bool init_v = true) {
  const int tid = threadIdx.x, gl = tid % GL;
#ifdef QK_BUILD_SPREAD  // consecutive pairs go to different wavefronts
  const int grp = ((tid % 64) / GL) * (BT / 64) + tid / 64;
#else
  const int grp = tid / GL;
#endif
  if (init_v)
    for (int e = tid; e < q * q; e += BT) V[(e / q) * vrs + (e % q) * vcs] = cd{(e / q == e % q) ? 1.0 : 0.0, 0.0};
  const double tol2 = 1e-29 * (double)max(p, 10);  // (1e-14 sqrt(p / 10))^2: a decade above the rounding floor of a length-p inner product
  for (int jc = grp; jc < q; jc += NG) {  // squared Frobenius norm (sets the absolute floor of the rotation test)
    double al = 0;
    for (int i = gl; i < p; i += GL) {
      const cd x = A[i * rs + jc * cs];
      al += x.x * x.x + x.y * x.y;
    }
    al = group_sum(al);
    sig[jc] = al;
  }
  __syncthreads();
  double frob = 0;
  for (int jc = 0; jc < q; ++jc) frob += sig[jc];
  __syncthreads();
  if (q >= 2) {
    const int qe = q + (q & 1), half = qe / 2, nr = qe - 1;
    int sweep = 0;
    bool done = false;
    const long long t_begin = wall_clock64();
    for (; sweep < MAX_SWEEPS; ++sweep) {
      if (tid == 0) sh->flag = 0, sh->worst = 0ull;
      __syncthreads();
      for (int r = 0; r < nr; ++r) {
        for (int k = grp; k < half; k += NG) {
          int c1 = r + k, c2 = r - k;  // round-robin tournament: (nr, r) and ((r + k) mod nr, (r - k) mod nr), k = 1..half-1
          if (c1 >= nr) c1 -= nr;
          if (c2 < 0) c2 += nr;
          if (k == 0) c1 = nr, c2 = r;
          if (c1 < q && c2 < q) {
            if (c1 > c2) {
              const int t_ = c1;
              c1 = c2, c2 = t_;
            }
            P a1 = A + c1 * cs;
            P a2 = A + c2 * cs;
            double al = 0, be = 0, gr = 0, gi = 0;
            cd xa[R > 0 ? R : 1], ya[R > 0 ? R : 1];
            if constexpr (R > 0) {
#pragma unroll
              for (int u = 0; u < R; ++u) {
                const int i = gl + u * GL;
                const bool in = i < p;
                xa[u] = in ? a1[i * rs] : cd{0.0, 0.0};
                ya[u] = in ? a2[i * rs] : cd{0.0, 0.0};
              }
#pragma unroll
              for (int u = 0; u < R; ++u) {
                const cd x = xa[u], y = ya[u];
                al += x.x * x.x + x.y * x.y;
                be += y.x * y.x + y.y * y.y;
                gr += x.x * y.x + x.y * y.y;
                gi += x.x * y.y - x.y * y.x;
              }
            } else {
              for (int i = gl; i < p; i += GL) {
                const cd x = a1[i * rs], y = a2[i * rs];
                al += x.x * x.x + x.y * x.y;
                be += y.x * y.x + y.y * y.y;
                gr += x.x * y.x + x.y * y.y;  // conj(x) * y
                gi += x.x * y.y - x.y * y.x;
              }
            }
            al = group_sum(al), be = group_sum(be), gr = group_sum(gr), gi = group_sum(gi);
            const double g2 = gr * gr + gi * gi;
            // rotate when |<a1, a2>| > tol |a1| max(|a2|, 0.03 |A|_F), a1 the longer column: relative orthogonality for
            // the columns that carry weight, the absolute accuracy of a LAPACK SVD (eps |A|) for the short ones -- whose
            // directions are rounding noise of the products that made A and would never settle under the relative test
            const double scale2 = fmax(al, be) * fmax(fmin(al, be), 1e-3 * frob);
            if (g2 > tol2 * scale2) {
              if (gl == 0) atomicMax(&sh->worst, (unsigned long long)__double_as_longlong(g2 / scale2));
              P v1 = V + c1 * vcs;
              P v2 = V + c2 * vcs;
              cd xv[R > 0 ? R : 1], yv[R > 0 ? R : 1];
              if constexpr (R > 0) {  // the V rows travel while the rotation is being worked out
#pragma unroll
                for (int u = 0; u < R; ++u) {
                  const int i = gl + u * GL;
                  const bool in = i < q;
                  xv[u] = in ? v1[i * vrs] : cd{0.0, 0.0};
                  yv[u] = in ? v2[i * vrs] : cd{0.0, 0.0};
                }
              }
              const double iga = rsqrt(g2);                  // 1 / |<a1, a2>|
              const double zeta = 0.5 * (be - al) * iga;
              const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
              const double c = rsqrt(1.0 + t * t), s = c * t;
              const double phr = gr * iga, phi = gi * iga;  // e^{i phi}
              const cd s1 = cd{-s * phr, s * phi};        // -s conj(ph)
              const cd s2 = cd{s * phr, s * phi};         //  s ph
              if constexpr (R > 0) {
#pragma unroll
                for (int u = 0; u < R; ++u) {
                  const int i = gl + u * GL;
                  if (i < p) {
                    a1[i * rs] = cfma(s1, ya[u], cd{c * xa[u].x, c * xa[u].y});
                    a2[i * rs] = cfma(s2, xa[u], cd{c * ya[u].x, c * ya[u].y});
                  }
                }
#pragma unroll
                for (int u = 0; u < R; ++u) {
                  const int i = gl + u * GL;
                  if (i < q) {
                    v1[i * vrs] = cfma(s1, yv[u], cd{c * xv[u].x, c * xv[u].y});
                    v2[i * vrs] = cfma(s2, xv[u], cd{c * yv[u].x, c * yv[u].y});
                  }
                }
              } else {
                for (int i = gl; i < p; i += GL) {
                  const cd x = a1[i * rs], y = a2[i * rs];
                  a1[i * rs] = cfma(s1, y, cd{c * x.x, c * x.y});
                  a2[i * rs] = cfma(s2, x, cd{c * y.x, c * y.y});
                }
                for (int i = gl; i < q; i += GL) {
                  const cd x = v1[i * vrs], y = v2[i * vrs];
                  v1[i * vrs] = cfma(s1, y, cd{c * x.x, c * x.y});
                  v2[i * vrs] = cfma(s2, x, cd{c * y.x, c * y.y});
                }
              }
              sh->flag = 1;
            }
          }
        }
        __syncthreads();
      }
      const int f = sh->flag;
      const double worst = __longlong_as_double((long long)sh->worst);
      __syncthreads();
      // done when nothing was rotated -- or only pairs that were already orthogonal to 1e-10: a rotation leaves a residue
      // of the order of the square of what it removed, far below the test, so the checking sweep can be skipped
      if (!f || worst <= 1e-20) {
        done = true;
        ++sweep;
        break;
      }
    }
    if (tid == 0) {
      // out of sweeps: a residue below 1e-10 (relative) is harmless for the truncation and the canonical form (it
      // perturbs singular values by 1e-20); anything larger is reported
      if (!done) {
        atomicAdd(error + 4, 1);
        if (__longlong_as_double((long long)sh->worst) > 1e-20) atomicOr(error, ERR_SWEEPS);
      }
      atomicAdd(error + 1, 1);              // statistics: factorisations, sweeps, most sweeps of one factorisation
      atomicAdd(error + 2, min(sweep, MAX_SWEEPS));
      atomicMax(error + 3, min(sweep, MAX_SWEEPS));
      atomicAdd(reinterpret_cast<unsigned long long*>(error + 8), (unsigned long long)(wall_clock64() - t_begin));  // 100 MHz ticks in sweeps
      atomicAdd(reinterpret_cast<unsigned long long*>(error + 10), (unsigned long long)(min(sweep, MAX_SWEEPS) * nr));  // steps
    }
  }
  for (int jc = grp; jc < q; jc += NG) {
    double al = 0;
    for (int i = gl; i < p; i += GL) {
      const cd x = A[i * rs + jc * cs];
      al += x.x * x.x + x.y * x.y;
    }
    al = group_sum(al);
    sig[jc] = sqrt(al);
  }
  __syncthreads();
  for (int jc = tid; jc < q; jc += BT) {
    const double v = sig[jc];
    int rank = 0;
    for (int i = 0; i < q; ++i) {
      const double u = sig[i];
      rank += (u > v) || (u == v && i < jc);
    }
    ord[rank] = jc;
  }
  __syncthreads();
}

// C[M x N] (row-major, ld N) = sum_k A(i, k) B(k, j); A(i, k) at A[i * ars + k * acs], B(k, j) at B[k * brs + j * bcs]
// Register-blocked: a thread owns a 4 x 4 block of C (16 independent accumulators, 8 operand loads per 16 products; the plain
// one-output-per-thread loop was latency-bound and took 85 % of the build time at bonds of 100).  Optional operand maps:
// row i of A is taken from source row amap[i] (conjugated if CONJA) and the result row scaled by rscale[amap[i]]; likewise
// column j of B from bmap[j] (conjugated if CONJB), result column scaled by cscale[bmap[j]] -- that is how the centre moves
// multiply by R = diag(s) V^H with the columns of V in sorted order.
template <bool CONJA, bool CONJB>
__device__ void wg_gemm(cd* __restrict__ C, const int M, const int N, const int K, const cd* __restrict__ A, const long ars, const long acs,
                        const cd* __restrict__ B, const long brs, const long bcs, const int* amap = nullptr, const double* rscale = nullptr,
                        const int* bmap = nullptr, const double* cscale = nullptr) {
  const int tn = (N + 3) / 4, tiles = ((M + 3) / 4) * tn;
  for (int t = threadIdx.x; t < tiles; t += BT) {
    const int ti = t / tn, tj = t - ti * tn;
    const cd* pa[4];
    const cd* pb[4];
    double ra[4], cb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = min(4 * ti + a, M - 1), si = amap ? amap[i] : i;
      pa[a] = A + si * ars;
      ra[a] = rscale ? rscale[si] : 1.0;
      const int jn = min(4 * tj + a, N - 1), sj = bmap ? bmap[jn] : jn;
      pb[a] = B + sj * bcs;
      cb[a] = cscale ? cscale[sj] : 1.0;
    }
    cd acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = cd{0.0, 0.0};
    for (int k = 0; k < K; ++k) {
      cd av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        av[a] = pa[a][k * acs];
        bv[a] = pb[a][k * brs];
        if (CONJA) av[a].y = -av[a].y;
        if (CONJB) bv[a].y = -bv[a].y;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = cfma(av[a], bv[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (4 * ti + a < M && 4 * tj + b < N) {
          const double f = ra[a] * cb[b];
          C[(long)(4 * ti + a) * N + 4 * tj + b] = cd{acc[a][b].x * f, acc[a][b].y * f};
        }
  }
  __syncthreads();
}

// The same product on the f64 matrix cores for the large ones (theta = A_q A_{q+1}, W = A V, R x neighbour): C[M x N] (row-major,
// ld N) = A B with A(i, k) at A[i * ars + k * acs], B(k, j) at B[k * brs + j * bcs].  One 16 x 16 tile of C per wavefront at a time,
// 3M complex product on v_mfma_f64_16x16x4_f64, operands read straight into fragments (lane (q4, j) of k-step s: A(16 tm + j,
// 4 s + q4) and B(4 s + q4, 16 tn + j); out-of-range elements read as zero).
typedef double v4d_g __attribute__((ext_vector_type(4)));
__device__ void wg_gemm_mfma(cd* __restrict__ C, const int M, const int N, const int K, const cd* __restrict__ A, const long ars, const long acs, const cd* __restrict__ B, const long brs,
                             const long bcs) {
  const int lane = threadIdx.x & 63, q4 = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tn_n = (N + 15) / 16, tiles = ((M + 15) / 16) * tn_n, nks = (K + 3) / 4;
  for (int t = wave; t < tiles; t += BT / 64) {
    const int tm = t / tn_n, tn = t - tm * tn_n;
    const int ia = 16 * tm + j, jb = 16 * tn + j;
    const bool va = ia < M, vb = jb < N;
    const cd* pa = A + (long)min(ia, M - 1) * ars + (long)q4 * acs;
    const cd* pb = B + (long)q4 * brs + (long)min(jb, N - 1) * bcs;
    v4d_g p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
    constexpr int PF = 4;
    cd xa[PF], xb[PF];
    auto ld = [&](const int s, cd& a, cd& b) __attribute__((always_inline)) {
      const int k = 4 * s + q4;
      const bool vk = k < K;
      a = (va && vk) ? pa[(long)(4 * s) * acs] : cd{0.0, 0.0};
      b = (vb && vk) ? pb[(long)(4 * s) * brs] : cd{0.0, 0.0};
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (i < nks) ld(i, xa[i], xb[i]);
      else xa[i] = cd{0.0, 0.0}, xb[i] = cd{0.0, 0.0};
    }
    for (int s0 = 0; s0 < nks; s0 += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const cd a = xa[i], b = xb[i];
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, p1, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, p2, 0, 0, 0);
        p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, b.x + b.y, p3, 0, 0, 0);
        if (s0 + PF + i < nks) ld(s0 + PF + i, xa[i], xb[i]);
        else xa[i] = cd{0.0, 0.0}, xb[i] = cd{0.0, 0.0};
      }
    }
    const v4d_g re = p1 - p2, im = p3 - p1 - p2;
    if (jb < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * tm + q4 + 4 * r;
        if (i < M) C[(long)i * N + jb] = cd{re[r], im[r]};
      }
    }
  }
  __syncthreads();
}

__device__ void wg_copy(cd* __restrict__ dst, const cd* __restrict__ src, const long n) {
  for (long e = threadIdx.x; e < n; e += BT) dst[e] = src[e];
  __syncthreads();
}

// ----------------------------------------------------------------------------------------
// BLOCK one-sided Jacobi on the f64 matrix cores: the factorisation for matrices beyond the LDS working set (bonds from
// ~40 to the cap).  The scalar kernel above spends one latency-bound step per round of column pairs (q - 1 rounds per
// sweep, every column re-read from L2 for two flops per byte); here the columns are kept in blocks of NBC = 8
// ([row][8] tiles, 128 contiguous bytes per row), a VISIT of a block pair works on their 16-column panel P and is
// BLAS-3 on v_mfma_f64_16x16x4_f64:
//     G = P^H P         one 16 x 16 complex tile, K = rows: 3 matrix instructions per 4 rows
//                       (Re = Pr^T Pr + Pi^T Pi, Im = S - S^T with S = Pr^T Pi: the transpose is taken in LDS)
//     J                 the rotations of the scalar kernel -- same formula, same test, taken from G's entries --
//                       applied to G from both sides and accumulated in J (16 x 16, LDS, one wavefront): all 120 pairs
//                       of the panel in a sweep's first round, the 64 cross pairs afterwards, so that every column pair
//                       of the matrix meets exactly once per sweep (a cyclic ordering of the scalar method)
//     P <- P J, V <- V J    3M complex products, 12 matrix instructions per 16 rows
// One wavefront per visit, block pairs in round-robin order, NWV visits at a time, one workgroup barrier per round.
// oracle/jacobi_model.py: jacobi_block restates it sequentially (tests/test_jacobi_model.py: against LAPACK).
// ----------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NBC = 8;         // columns per block
constexpr int NWV = BT / 64;   // wavefronts per workgroup = visits in flight
constexpr int GLD = 17;        // leading dimension (complex) of the 16 x 16 LDS matrices: odd, so that a column is conflict-free too
constexpr int BLK_LDS = 2 * 16 * GLD;  // complex elements of LDS per wavefront: G and J

__device__ __forceinline__ void wave_lds_sync() {  // orders the LDS accesses of ONE wavefront (its lanes exchange data through LDS)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int pad_to(const int x, const int m) { return (x + m - 1) / m * m; }

// The rotations of one visit on G (16 x 16 Hermitian, LDS), accumulated in Jm: 8 disjoint column pairs per step.  The 64 lanes
// are the 8 x 8 grid of pair combinations: lane (P, Q) owns the 2 x 2 block G[rows of pair P][columns of pair Q] and replaces it
// by R_P^H G[P, Q] R_Q -- both rotations worked out from the diagonal blocks G[P, P], G[Q, Q], so a step is ONE read phase and
// one write phase; the columns of J take R_Q the same way (two (row, pair) items per lane).
// Returns whether any pair was rotated; `worst` collects the largest squared relative inner product met.
struct BlkRot {
  double c;
  cd s1, s2;  // a1' = c a1 + s1 a2, a2' = s2 a1 + c a2
  bool rot;
};
__device__ __forceinline__ BlkRot blk_rotation(const double al, const double be, const cd g, const double tol2, const double floor2, double& worst) {
  BlkRot r{1.0, cd{0.0, 0.0}, cd{0.0, 0.0}, false};
  const double g2 = g.x * g.x + g.y * g.y;
  const double scale2 = fmax(al, be) * fmax(fmin(al, be), floor2);
  r.rot = g2 > tol2 * scale2;
  if (r.rot) {
    worst = fmax(worst, g2 / scale2);
    const double iga = rsqrt(g2);
    const double zeta = 0.5 * (be - al) * iga;
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    r.c = rsqrt(1.0 + t * t);
    const double sn = r.c * t, phr = g.x * iga, phi = g.y * iga;
    r.s1 = cd{-sn * phr, sn * phi};  // -s conj(ph)
    r.s2 = cd{sn * phr, sn * phi};   //  s ph
  }
  return r;
}
template <bool FULL>
__device__ __forceinline__ bool blk_rotations(lds_cd_ptr G, lds_cd_ptr Jm, const int lane, const double tol2, const double floor2, double& worst) {
  const int pq = lane & 7, pp = lane >> 3;  // this lane's block: rows of pair pp, columns of pair pq
  for (int e = lane; e < 256; e += 64) Jm[(e >> 4) * GLD + (e & 15)] = cd{(e >> 4) == (e & 15) ? 1.0 : 0.0, 0.0};
  wave_lds_sync();
  bool any = false;
  constexpr int STEPS = FULL ? 15 : 8;
  auto pair_of = [&](const int st, const int k, int& c1, int& c2) __attribute__((always_inline)) {
    if (FULL) {  // round-robin tournament over the 16 columns
      c1 = st + k, c2 = st - k;
      if (c1 >= 15) c1 -= 15;
      if (c2 < 0) c2 += 15;
      if (k == 0) c1 = 15, c2 = st;
      if (c1 > c2) {
        const int t_ = c1;
        c1 = c2, c2 = t_;
      }
    } else {  // cross pairs: column k of the first block with column (k + st) mod 8 of the second
      c1 = k, c2 = 8 + ((k + st) & 7);
    }
  };
  for (int st = 0; st < STEPS; ++st) {
    int r1, r2, c1, c2;
    pair_of(st, pp, r1, r2);
    pair_of(st, pq, c1, c2);
    double wdummy = 0.0;
    const BlkRot rp = blk_rotation(G[r1 * GLD + r1].x, G[r2 * GLD + r2].x, G[r1 * GLD + r2], tol2, floor2, wdummy);
    const BlkRot rq = blk_rotation(G[c1 * GLD + c1].x, G[c2 * GLD + c2].x, G[c1 * GLD + c2], tol2, floor2, worst);
    const cd b00 = G[r1 * GLD + c1], b01 = G[r1 * GLD + c2], b10 = G[r2 * GLD + c1], b11 = G[r2 * GLD + c2];
    // the columns of J that pair pq rotates: rows pp and pp + 8
    const cd j0a = Jm[pp * GLD + c1], j0b = Jm[pp * GLD + c2], j1a = Jm[(pp + 8) * GLD + c1], j1b = Jm[(pp + 8) * GLD + c2];
    any |= __builtin_amdgcn_ballot_w64(rq.rot) != 0ull;
    wave_lds_sync();  // every lane has read before any entry changes
    if (rp.rot || rq.rot) {
      // T = B R_Q, B' = R_P^H T
      const cd t00 = cfma(rq.s1, b01, cd{rq.c * b00.x, rq.c * b00.y}), t01 = cfma(rq.s2, b00, cd{rq.c * b01.x, rq.c * b01.y});
      const cd t10 = cfma(rq.s1, b11, cd{rq.c * b10.x, rq.c * b10.y}), t11 = cfma(rq.s2, b10, cd{rq.c * b11.x, rq.c * b11.y});
      const cd s1c = cd{rp.s1.x, -rp.s1.y}, s2c = cd{rp.s2.x, -rp.s2.y};
      G[r1 * GLD + c1] = cfma(s1c, t10, cd{rp.c * t00.x, rp.c * t00.y});
      G[r1 * GLD + c2] = cfma(s1c, t11, cd{rp.c * t01.x, rp.c * t01.y});
      G[r2 * GLD + c1] = cfma(s2c, t00, cd{rp.c * t10.x, rp.c * t10.y});
      G[r2 * GLD + c2] = cfma(s2c, t01, cd{rp.c * t11.x, rp.c * t11.y});
    }
    if (rq.rot) {
      Jm[pp * GLD + c1] = cfma(rq.s1, j0b, cd{rq.c * j0a.x, rq.c * j0a.y});
      Jm[pp * GLD + c2] = cfma(rq.s2, j0a, cd{rq.c * j0b.x, rq.c * j0b.y});
      Jm[(pp + 8) * GLD + c1] = cfma(rq.s1, j1b, cd{rq.c * j1a.x, rq.c * j1a.y});
      Jm[(pp + 8) * GLD + c2] = cfma(rq.s2, j1a, cd{rq.c * j1b.x, rq.c * j1b.y});
    }
    wave_lds_sync();
  }
  return any;
}

// P <- P J for one 16-column panel (blocks at P1 / P2, rows x [8] complex each), `rows` a multiple of 32.  3M product.
__device__ __forceinline__ void blk_apply(cd* __restrict__ P1, cd* __restrict__ P2, const int rows, const double (&Jr)[4], const double (&Ji)[4], const double (&Js)[4], const int q4,
                                          const int j) {
  const cd* const src = (q4 < 2 ? P1 : P2) + (long)j * NBC + 4 * (q4 & 1);  // row 16 t + j, panel columns 4 q4 .. 4 q4 + 3
  cd* const dst = (j < 8 ? P1 : P2) + (long)q4 * NBC + (j & 7);               // row 16 t + q4 + 4 r, panel column j
  const int nt = rows / 16;
  cd xa[4], xb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) xa[s] = src[s];
  auto tile = [&](const cd(&x)[4], const int t) __attribute__((always_inline)) {
    v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s].x, Jr[s], p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s].y, Ji[s], p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s].x + x[s].y, Js[s], p3, 0, 0, 0);
    }
    const v4d re = p1 - p2, im = p3 - p1 - p2;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(long)(16 * t + 4 * r) * NBC] = cd{re[r], im[r]};
  };
  for (int t = 0; t < nt; t += 2) {
#pragma unroll
    for (int s = 0; s < 4; ++s) xb[s] = src[(long)(16 * (t + 1)) * NBC + s];
    tile(xa, t);
    if (t + 2 < nt) {
#pragma unroll
      for (int s = 0; s < 4; ++s) xa[s] = src[(long)(16 * (t + 2)) * NBC + s];
    }
    tile(xb, t + 1);
  }
}

// One visit of the block pair (b1, b2) by one wavefront.  Returns whether the panel was rotated.
template <bool FULL>
__device__ __forceinline__ bool blk_visit(cd* __restrict__ AB, const long astride, const int arows, cd* __restrict__ VB, const long vstride, const int vrows, const int b1, const int b2, lds_cd_ptr G,
                                          lds_cd_ptr Jm, const int lane, const double tol2, const double floor2, double& worst) {
  const int q4 = lane >> 4, j = lane & 15;
  // ---- G = P^H P
  {
    const cd* const base = AB + (long)(j < 8 ? b1 : b2) * astride + (long)q4 * NBC + (j & 7);  // row 4 s + q4, panel column j
    constexpr int PF = 8;
    cd x[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) x[i] = base[(long)(4 * i) * NBC];
    v4d p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
    const int nks = arows / 4;  // a multiple of PF (rows padded to 32)
    for (int s0 = 0; s0 < nks; s0 += PF) {
      const bool more = s0 + PF < nks;
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i].x, x[i].x, p1, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i].y, x[i].y, p2, 0, 0, 0);
        p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i].x, x[i].y, p3, 0, 0, 0);
        if (more) x[i] = base[(long)(4 * (s0 + PF + i)) * NBC];
      }
    }
    double* const Td = (double*)Jm;  // S = Pr^T Pi, parked in the J region until its transpose has been taken
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = q4 + 4 * r;
      G[i * GLD + j] = cd{p1[r] + p2[r], 0.0};
      ((__attribute__((address_space(3))) double*)Td)[i * (2 * GLD) + j] = p3[r];
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = q4 + 4 * r;
      const __attribute__((address_space(3))) double* T = (__attribute__((address_space(3))) double*)Td;
      const double im = T[i * (2 * GLD) + j] - T[j * (2 * GLD) + i];
      const cd gv = G[i * GLD + j];
      G[i * GLD + j] = cd{gv.x, im};
    }
    wave_lds_sync();
  }
  // ---- the rotations
  const bool any = blk_rotations<FULL>(G, Jm, lane, tol2, floor2, worst);
  if (!any) return false;
  // ---- P <- P J, V <- V J
  double Jr[4], Ji[4], Js[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const cd v = Jm[(4 * q4 + s) * GLD + j];
    Jr[s] = v.x, Ji[s] = v.y, Js[s] = v.x + v.y;
  }
  blk_apply(AB + (long)b1 * astride, AB + (long)b2 * astride, arows, Jr, Ji, Js, q4, j);
  if (VB) blk_apply(VB + (long)b1 * vstride, VB + (long)b2 * vstride, vrows, Jr, Ji, Js, q4, j);
  return true;
}

// Sweeps of the block Jacobi over the first `ncols` (a multiple of 16) columns of AB (blocks of `astride` complex, `arows` rows, a
// multiple of 32); VB (may be null: rotations are then not accumulated) likewise.  Returns the number of sweeps through sh->keep
// (workgroup-uniform) and whether it converged through sh->flag.
// CLEAN PAIRS ARE SKIPPED: `chk` (global, nb x nb ints, may be null) remembers for every block pair the round in which a visit
// found nothing to rotate, mod_s the round in which a block's columns last changed; a pair verified clean after both of its
// blocks' last change is still clean -- in the last sweeps that is most of them, and the visit's Gram product is saved too.
constexpr int BLK_NB_MAX = 160;  // blocks of 8 columns: bonds up to 640 on the factorised side
__device__ void blk_sweeps(cd* AB, const long astride, const int arows, cd* VB, const long vstride, const int vrows, const int ncols, const double tol2, const double floor2, WgShared* sh, cd* lds,
                           int* error, int* chk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = ncols / NBC, nr = nb - 1, half = nb / 2;  // nb is even
  lds_cd_ptr G = (lds_cd_ptr)(lds + (long)wave * BLK_LDS);
  lds_cd_ptr Jm = G + 16 * GLD;
  __shared__ int mod_s[BLK_NB_MAX];
  if (nb > BLK_NB_MAX) chk = nullptr;
  if (chk) {
    for (int e = tid; e < nb * nb; e += BT) chk[e] = -1;
    for (int e = tid; e < nb; e += BT) mod_s[e] = 0;
  }
  int sweep = 0;
  bool done = false;
  for (; sweep < MAX_SWEEPS; ++sweep) {
    if (tid == 0) sh->flag = 0, sh->worst = 0ull;
    __syncthreads();
    for (int r = 0; r < nr; ++r) {
      bool rotated = false;
      double worst = 0.0;
      const int now = sweep * nr + r + 1;
      for (int k = wave; k < half; k += NWV) {
        int b1 = r + k, b2 = r - k;
        if (b1 >= nr) b1 -= nr;
        if (b2 < 0) b2 += nr;
        if (k == 0) b1 = nr, b2 = r;
        if (b1 > b2) {
          const int t_ = b1;
          b1 = b2, b2 = t_;
        }
        if (chk && __builtin_amdgcn_readfirstlane(chk[b1 * nb + b2]) >= max(mod_s[b1], mod_s[b2])) continue;  // verified clean since both blocks last changed
        bool rot;
        if (r == 0) rot = blk_visit<true>(AB, astride, arows, VB, vstride, vrows, b1, b2, G, Jm, lane, tol2, floor2, worst);
        else rot = blk_visit<false>(AB, astride, arows, VB, vstride, vrows, b1, b2, G, Jm, lane, tol2, floor2, worst);
        rotated |= rot;
        if (chk && lane == 0) {
          if (rot) mod_s[b1] = now, mod_s[b2] = now;  // (a block belongs to one pair per round: no other wavefront reads these two entries this round)
          else chk[b1 * nb + b2] = now;
        }
      }
      if (rotated) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o));
        if (lane == 0) {
          sh->flag = 1;
          atomicMax(&sh->worst, (unsigned long long)__double_as_longlong(worst));
        }
      }
      __syncthreads();  // the round's panels are complete (global stores drained) before the next round pairs the blocks anew
    }
    const int f = sh->flag;
    const double worst_all = __longlong_as_double((long long)sh->worst);
    __syncthreads();
    if (!f || worst_all <= 1e-20) {
      done = true;
      ++sweep;
      break;
    }
  }
  if (tid == 0) {
    if (!done) {
      atomicAdd(error + 4, 1);
      if (__longlong_as_double((long long)sh->worst) > 1e-20) atomicOr(error, ERR_SWEEPS);
    }
    atomicAdd(error + 1, 1);
    atomicAdd(error + 2, min(sweep, MAX_SWEEPS));
    atomicMax(error + 3, min(sweep, MAX_SWEEPS));
    atomicAdd(error + 7, 1);  // block factorisations
  }
  __syncthreads();
}

// Sum over the 64 lanes of a wavefront (complex)
__device__ __forceinline__ cd wave_sum(cd v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v.x += __shfl_xor(v.x, o), v.y += __shfl_xor(v.y, o);
  return v;
}

constexpr int MGS_NB = 8;   // columns of a Gram-Schmidt panel at most
constexpr int MGS_R = 16;   // rows of a column per lane at most (64 lanes): columns up to 1024 rows
constexpr double PRECOND_CUT = 1e-22;  // rows of R below 1e-11 ||A||_F: three orders of magnitude below the smallest singular value the truncation keeps
// Sum over the 64 lanes of a wavefront, result in every lane: DPP permutations inside the rows of 16 lanes, then the four row
// sums through v_readlane (a dozen VALU instructions instead of six LDS round trips per 32-bit half).
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);   // lane ^ 1
  v += dpp_f64<0x4E>(v);   // lane ^ 2
  v += dpp_f64<0x141>(v);  // the other quad of the 8
  v += dpp_f64<0x140>(v);  // the other half of the 16
  auto row = [&](const int l) __attribute__((always_inline)) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  };
  return (row(0) + row(16)) + (row(32) + row(48));
}

// R of S = Q R (S: p x q column-major, overwritten) by modified Gram-Schmidt, a panel of up to `nbp` columns at a time; L = R^H goes
// into the block layout LB (column k of L in block k / 8, rows of 8 complex).  The panel lives in LDS (`lds`: nbp * p complex): it is
// orthogonalised there and then projected out of every trailing column -- one column per wavefront at a time, the column in
// registers (RR rows per lane), read from and written to memory once per panel.  Q is never needed, so the panel is not written back.
// put_l(j, k, re, im) receives r_kj (k <= j); KEEP_Q: the orthonormalised panel is written back, so that S ends as Q.
template <int RR, bool KEEP_Q, typename PUT>
__device__ void mgs_panels(cd* S, const int p, const int q, const int nbp, const PUT put_l, cd* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  lds_cd_ptr QP = (lds_cd_ptr)lds;  // QP[kk * p + i]
  __shared__ double nrm_s[MGS_NB];
  for (int k0 = 0; k0 < q; k0 += nbp) {
    const int kn = min(nbp, q - k0);
    for (int e = tid; e < kn * p; e += BT) QP[e] = S[(long)k0 * p + e];
    __syncthreads();
    // inside the panel: column kk is projected out of the later ones (one per wavefront); normalisation is folded in
    for (int kk = 0; kk < kn; ++kk) {
      double n2 = 0;
      for (int i = lane; i < p; i += 64) {
        const cd x = QP[kk * p + i];
        n2 += x.x * x.x + x.y * x.y;
      }
      n2 = wave_sum_f64(n2);
      const double nrm = sqrt(n2), inv = nrm > 1e-300 ? 1.0 / nrm : 0.0;
      if (tid == 0) {
        nrm_s[kk] = inv;
        put_l(k0 + kk, k0 + kk, nrm, 0.0);
      }
      for (int jj = kk + 1 + wave; jj < kn; jj += NWV) {
        double dr = 0, di = 0;
        for (int i = lane; i < p; i += 64) {
          const cd x = QP[kk * p + i], y = QP[jj * p + i];
          dr += x.x * y.x + x.y * y.y, di += x.x * y.y - x.y * y.x;  // conj(c_kk) c_jj
        }
        dr = wave_sum_f64(dr) * inv, di = wave_sum_f64(di) * inv;  // r_kj = conj(q_kk) c_jj
        const double fr = dr * inv, fi = di * inv;
        for (int i = lane; i < p; i += 64) {
          const cd x = QP[kk * p + i], y = QP[jj * p + i];
          QP[jj * p + i] = cd{y.x - (fr * x.x - fi * x.y), y.y - (fr * x.y + fi * x.x)};
        }
        if (lane == 0) put_l(k0 + jj, k0 + kk, dr, di);
      }
      __syncthreads();
    }
    for (int kk = wave; kk < kn; kk += NWV) {
      const double inv = nrm_s[kk];
      for (int i = lane; i < p; i += 64) {
        const cd x = QP[kk * p + i];
        const cd y = cd{x.x * inv, x.y * inv};
        QP[kk * p + i] = y;
        if (KEEP_Q) S[(long)(k0 + kk) * p + i] = y;
      }
    }
    __syncthreads();
    // the trailing columns: NC = 2 per wavefront at a time where the registers allow (columns up to 512 rows): one read of a panel
    // vector serves both and their two reduction chains overlap; the last one of an odd count goes alone
    constexpr int NC = RR <= 8 ? 2 : 1;
    for (int j0 = k0 + kn + NC * wave; j0 < q; j0 += NC * NWV) {
      cd y[NC][RR];
      bool have[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        have[c] = j0 + c < q;
        const cd* const cj = S + (long)(have[c] ? j0 + c : j0) * p;
#pragma unroll
        for (int u = 0; u < RR; ++u) {
          const int i = lane + 64 * u;
          y[c][u] = (have[c] && i < p) ? cj[i] : cd{0.0, 0.0};
        }
      }
      for (int kk = 0; kk < kn; ++kk) {
        double dr[NC], di[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) dr[c] = 0, di[c] = 0;
#pragma unroll
        for (int u = 0; u < RR; ++u) {
          const int i = lane + 64 * u;
          const cd x = i < p ? (cd)QP[kk * p + i] : cd{0.0, 0.0};
#pragma unroll
          for (int c = 0; c < NC; ++c) dr[c] += x.x * y[c][u].x + x.y * y[c][u].y, di[c] += x.x * y[c][u].y - x.y * y[c][u].x;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) dr[c] = wave_sum_f64(dr[c]), di[c] = wave_sum_f64(di[c]);
#pragma unroll
        for (int u = 0; u < RR; ++u) {
          const int i = lane + 64 * u;
          const cd x = i < p ? (cd)QP[kk * p + i] : cd{0.0, 0.0};
#pragma unroll
          for (int c = 0; c < NC; ++c) y[c][u] = cd{y[c][u].x - (dr[c] * x.x - di[c] * x.y), y[c][u].y - (dr[c] * x.y + di[c] * x.x)};
        }
        if (lane == 0) {
#pragma unroll
          for (int c = 0; c < NC; ++c)
            if (have[c]) put_l(j0 + c, k0 + kk, dr[c], di[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (!have[c]) continue;
        cd* const cj = S + (long)(j0 + c) * p;
#pragma unroll
        for (int u = 0; u < RR; ++u) {
          const int i = lane + 64 * u;
          if (i < p) cj[i] = y[c][u];
        }
      }
    }
    __syncthreads();
  }
}

// The PRECONDITIONED factorisation (Drmac / Veselic), same contract as jacobi_orth / jacobi_auto: A <- W = A V, V row-major q x q,
// sig = column norms of W, ord = columns by decreasing norm -- except that only the first r columns (the numerical rank) are
// produced, the others are zero with sig = 0 (callers use the columns whose sig counts).
// A gate's theta is a GRADED matrix -- singular values falling by twenty orders of magnitude --, on which plain one-sided Jacobi
// needs 13-17 sweeps (it peels about a decade and a half per sweep; oracle/jacobi_model.py).  So:
//   1. the columns are sorted by decreasing norm (S: column-major copy, p x q);
//   2. R of S = Q R by modified Gram-Schmidt, a panel of MGS_NB columns at a time: a trailing column is read once per panel,
//      kept in registers by one wavefront while the panel's vectors are projected out, and written once; Q is never needed.
//      L = R^H is written straight into the block layout (q rows, column k in block k / 8);
//   3. the columns of L (rows of R) with squared norm below CUT ||A||_F^2 are dropped: six orders of magnitude below the
//      truncation budget (they would perturb the smallest kept singular value by 1e-6 of itself), and what is left has full
//      numerical rank, so the rotation test below is purely relative;
//   4. block Jacobi on L (q x r): 6-8 sweeps -- L's columns are nearly orthogonal already -- WITHOUT accumulating rotations;
//   5. L V_L = U_L Sigma: the right singular vectors of A are V = (normalised columns of L V_L), rows back in A's column order,
//      and W = A V (one product on the original A).
// S: p x q complex scratch (column-major), LB: pad32(q) x pad16(q) complex (block layout), lds: NWV * BLK_LDS complex.
template <int MINWG>  // (instantiated for the two-workgroups-per-CU kernel only: 256 VGPRs, 76 KiB of LDS)
__device__ __noinline__ void jacobi_precond(cd* A, const long rs, const long cs, const int p, const int q, cd* V, double* sig, int* ord, WgShared* sh, int* error, cd* lds, const int lds_elems, cd* S, cd* LB, int* chk) {
  const int tid = threadIdx.x, gl = tid % GL, grp = tid / GL, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrows = pad_to(q, 32), qpad = pad_to(q, 16);
  const long lstride = (long)lrows * NBC;
  const long long t_begin = wall_clock64();
  // ---- 1. column norms and their order; S = the sorted columns
  for (int jc = grp; jc < q; jc += NG) {
    double al = 0;
    for (int i = gl; i < p; i += GL) {
      const cd x = A[i * rs + jc * cs];
      al += x.x * x.x + x.y * x.y;
    }
    al = group_sum(al);
    sig[jc] = al;
  }
  for (long e = tid; e < (long)lrows * qpad; e += BT) LB[e] = cd{0.0, 0.0};
  __syncthreads();
  double frob = 0;
  for (int jc = 0; jc < q; ++jc) frob += sig[jc];
  for (int jc = tid; jc < q; jc += BT) {
    const double v = sig[jc];
    int rank = 0;
    for (int i = 0; i < q; ++i) {
      const double u = sig[i];
      rank += (u > v) || (u == v && i < jc);
    }
    ord[jc] = rank;  // column jc of A is column `rank` of S
  }
  __syncthreads();
  for (long e = tid; e < (long)p * q; e += BT) {
    const int i = (int)(e / q), jc = (int)(e - (long)i * q);
    S[(long)ord[jc] * p + i] = A[i * rs + jc * cs];
  }
  __syncthreads();
  const long long t_sorted = wall_clock64();
  // ---- 2. panel MGS: L[j][k] = conj(r_kj)
  {
    const int nbp = max(1, min(MGS_NB, lds_elems / max(p, 1)));  // columns of a panel: what the LDS working set holds
    auto put_l = [=](const int j, const int k, const double re, const double im) __attribute__((always_inline)) { LB[(long)(k >> 3) * lstride + (long)j * NBC + (k & 7)] = cd{re, -im}; };
    if (p <= 128) mgs_panels<2, false>(S, p, q, nbp, put_l, lds);
    else if (p <= 256) mgs_panels<4, false>(S, p, q, nbp, put_l, lds);
    else if (p <= 512) mgs_panels<8, false>(S, p, q, nbp, put_l, lds);
    else if (p <= 768) mgs_panels<12, false>(S, p, q, nbp, put_l, lds);
    else mgs_panels<16, false>(S, p, q, nbp, put_l, lds);
  }
  const long long t_mgs = wall_clock64();
  // ---- 3. numerical rank: the last column of L (row of R) that carries weight
  for (int c = grp; c < q; c += NG) {
    double al = 0;
    const cd* col = LB + (long)(c >> 3) * lstride + (c & 7);
    for (int i = gl; i < q; i += GL) {
      const cd x = col[(long)i * NBC];
      al += x.x * x.x + x.y * x.y;
    }
    al = group_sum(al);
    sig[c] = al;
  }
  __syncthreads();
  if (tid == 0) {
    int rk = 1;
    for (int c = 0; c < q; ++c)
      if (sig[c] > PRECOND_CUT * frob) rk = c + 1;
    sh->keep = rk;
  }
  __syncthreads();
  const int rk = sh->keep, rpad = pad_to(rk, 16);
  __syncthreads();
  // (columns rk .. rpad - 1 of L take part in the sweeps as they are: at most 15 columns below the cut)
  // ---- 4. block Jacobi on L, rotations not accumulated
  if (rpad >= 16) blk_sweeps(LB, lstride, lrows, nullptr, 0, 0, rpad, 1e-29 * (double)max(q, 10), PRECOND_CUT * frob, sh, lds, error, chk);
  const long long t_sweeps = wall_clock64();
  // ---- 5. sig, V = normalised columns of L V_L with the rows back in A's column order, W = A V
  for (int c = grp; c < q; c += NG) {
    double al = 0;
    if (c < rk) {
      const cd* col = LB + (long)(c >> 3) * lstride + (c & 7);
      for (int i = gl; i < q; i += GL) {
        const cd x = col[(long)i * NBC];
        al += x.x * x.x + x.y * x.y;
      }
      al = group_sum(al);
    }
    sig[c] = sqrt(al);
  }
  __syncthreads();
  for (long e = tid; e < (long)q * q; e += BT) {
    const int i = (int)(e / q), c = (int)(e - (long)i * q);
    cd v = cd{0.0, 0.0};
    if (c < rk && sig[c] > 0.0) {
      const cd x = LB[(long)(c >> 3) * lstride + (long)ord[i] * NBC + (c & 7)];
      const double inv = 1.0 / sig[c];
      v = cd{x.x * inv, x.y * inv};
    }
    V[e] = v;
  }
  __syncthreads();
  wg_gemm_mfma(S, p, rk, q, A, rs, cs, V, q, 1);  // W[p x rk] (row-major in S) = A V
  for (long e = tid; e < (long)p * q; e += BT) {
    const int i = (int)(e / q), c = (int)(e - (long)i * q);
    A[i * rs + c * cs] = c < rk ? S[(long)i * rk + c] : cd{0.0, 0.0};
  }
  __syncthreads();
  for (int jc = tid; jc < q; jc += BT) {
    const double v = sig[jc];
    int rank = 0;
    for (int i = 0; i < q; ++i) {
      const double u = sig[i];
      rank += (u > v) || (u == v && i < jc);
    }
    ord[rank] = jc;
  }
  if (tid == 0) {  // 100 MHz ticks: all of it; sorting + copy, Gram-Schmidt, sweeps, V and W = A V
    const long long t_end = wall_clock64();
    atomicAdd(reinterpret_cast<unsigned long long*>(error + 14), (unsigned long long)(t_end - t_begin));
    atomicAdd(reinterpret_cast<unsigned long long*>(error + 16), (unsigned long long)(t_sorted - t_begin));
    atomicAdd(reinterpret_cast<unsigned long long*>(error + 18), (unsigned long long)(t_mgs - t_sorted));
    atomicAdd(reinterpret_cast<unsigned long long*>(error + 20), (unsigned long long)(t_sweeps - t_mgs));
    atomicAdd(reinterpret_cast<unsigned long long*>(error + 22), (unsigned long long)(t_end - t_sweeps));
  }
  __syncthreads();
}

// The same factorisation with the working set in LDS when it fits (A and V side by side, odd leading dimension so
// that the 16 lanes of a pair hit 16 different banks): a step is then a few hundred cycles instead of a store-drain +
// L2 round trip.  Results are copied back to the global A (same strides) and to V (row-major, ld q).
// (Inlined into the kernel so that the kernel's register budget -- MINWG workgroups per CU -- governs it; the variant that
// keeps 8 rows per lane in registers exists only at 2 workgroups per CU.)
// QR for the centre moves of large sites: A (p x q, element (i, j) at A[i * rs + j * cs]) = Q R with Q orthonormal to working
// precision -- Gram-Schmidt TWICE ("twice is enough": the second pass removes what the first, on a centre tensor whose singular
// values span many orders of magnitude, leaves of the earlier columns in the later ones) -- and R = R2 R1.  No sweeps at all: a
// centre move needs an isometry and a triangle, not singular values; the bond does not change (what the host builder's QR does).
// Out: S = Q (column-major, p x q); R1, R2 (row-major q x q, upper triangles; the strict lower parts are zeroed).
__device__ void mgs2_qr(const cd* A, const long rs, const long cs, const int p, const int q, cd* S, cd* R1, cd* R2, cd* lds, const int lds_elems) {
  const int tid = threadIdx.x;
  for (long e = tid; e < (long)p * q; e += BT) {
    const int i = (int)(e / q), jc = (int)(e - (long)i * q);
    S[(long)jc * p + i] = A[i * rs + jc * cs];
  }
  for (long e = tid; e < (long)q * q; e += BT) R1[e] = cd{0.0, 0.0}, R2[e] = cd{0.0, 0.0};
  __syncthreads();
  const int nbp = max(1, min(MGS_NB, lds_elems / max(p, 1)));
  for (int pass = 0; pass < 2; ++pass) {
    cd* const R = pass == 0 ? R1 : R2;
    auto put = [=](const int j, const int k, const double re, const double im) __attribute__((always_inline)) { R[(long)k * q + j] = cd{re, im}; };
    if (p <= 128) mgs_panels<2, true>(S, p, q, nbp, put, lds);
    else if (p <= 256) mgs_panels<4, true>(S, p, q, nbp, put, lds);
    else if (p <= 512) mgs_panels<8, true>(S, p, q, nbp, put, lds);
    else if (p <= 768) mgs_panels<12, true>(S, p, q, nbp, put, lds);
    else mgs_panels<16, true>(S, p, q, nbp, put, lds);
  }
}

constexpr int g_precond_from = 48;  // columns from which a factorisation takes the preconditioned block path
template <int MINWG>
__device__ __forceinline__ void jacobi_auto(cd* A, const long rs, const long cs, const int p, const int q, cd* V, double* sig, int* ord, WgShared* sh,
                                            int* error, cd* lds, const int lds_elems, cd* scratch, cd* lbuf, const long lbuf_elems) {
  const int ld = q | 1;
  // from 48 columns on the preconditioned block factorisation is the faster one even where the scalar one would fit the LDS
  // (a graded 78 x 66 theta: 0.9 against 1.8 ms; 64 x 48: equal)
  const bool blocked = MINWG <= 2 && lbuf && q >= g_precond_from && p <= 64 * MGS_R && lds_elems >= NWV * BLK_LDS;
  if (threadIdx.x == 0) atomicAdd(error + ((!blocked && (long)(p + q) * ld <= lds_elems) ? 5 : 6), 1);  // statistics: scalar in LDS / the rest
  if (blocked) {
    if constexpr (MINWG <= 2) jacobi_precond<MINWG>(A, rs, cs, p, q, V, sig, ord, sh, error, lds, lds_elems, scratch, lbuf, reinterpret_cast<int*>(lbuf + lbuf_elems));
  } else if ((long)(p + q) * ld <= lds_elems) {
    cd* LA = lds;
    cd* LV = lds + (long)p * ld;
#ifndef QK_BUILD_NO_SORT  // de Rijk: start from the columns in order of decreasing norm (V starts as that permutation)
    {
      const int grp = threadIdx.x / GL, gl = threadIdx.x % GL;
      for (int jc = grp; jc < q; jc += NG) {
        double al = 0;
        for (int i = gl; i < p; i += GL) {
          const cd x = A[i * rs + jc * cs];
          al += x.x * x.x + x.y * x.y;
        }
        al = group_sum(al);
        sig[jc] = al;
      }
      __syncthreads();
      for (int jc = threadIdx.x; jc < q; jc += BT) {
        const double v = sig[jc];
        int rank = 0;
        for (int i = 0; i < q; ++i) {
          const double u = sig[i];
          rank += (u > v) || (u == v && i < jc);
        }
        ord[jc] = rank;  // column jc goes to position rank
      }
      __syncthreads();
      for (int e = threadIdx.x; e < p * q; e += BT) {
        const int i = e / q, jc = e - i * q;
        LA[i * ld + ord[jc]] = A[i * rs + jc * cs];
      }
      for (int e = threadIdx.x; e < q * q; e += BT) {
        const int i = e / q, jc = e - i * q;
        LV[i * ld + jc] = cd{(ord[i] == jc) ? 1.0 : 0.0, 0.0};
      }
      __syncthreads();
    }
    {
      const lds_cd_ptr la = (lds_cd_ptr)LA, lv = (lds_cd_ptr)LV;
      const int rows = (max(p, q) + GL - 1) / GL;  // rows of a column per lane
#ifndef QK_BUILD_NO_REGS
      if (rows <= 2) jacobi_orth<lds_cd_ptr, 2>(la, ld, 1, p, q, lv, ld, 1, sig, ord, sh, error, false);
      else if (rows <= 4) jacobi_orth<lds_cd_ptr, 4>(la, ld, 1, p, q, lv, ld, 1, sig, ord, sh, error, false);
      else if (MINWG <= 2 && rows <= 8) jacobi_orth<lds_cd_ptr, (MINWG <= 2 ? 8 : 4)>(la, ld, 1, p, q, lv, ld, 1, sig, ord, sh, error, false);
      else
#endif
        jacobi_orth<lds_cd_ptr, 0>(la, ld, 1, p, q, lv, ld, 1, sig, ord, sh, error, false);
    }
#else
    for (int e = threadIdx.x; e < p * q; e += BT) {
      const int i = e / q, jc = e - i * q;
      LA[i * ld + jc] = A[i * rs + jc * cs];
    }
    __syncthreads();
    jacobi_orth((lds_cd_ptr)LA, ld, 1, p, q, (lds_cd_ptr)LV, ld, 1, sig, ord, sh, error);
#endif
    for (int e = threadIdx.x; e < p * q; e += BT) {
      const int i = e / q, jc = e - i * q;
      A[i * rs + jc * cs] = LA[i * ld + jc];
    }
    for (int e = threadIdx.x; e < q * q; e += BT) {
      const int i = e / q, jc = e - i * q;
      V[e] = LV[i * ld + jc];
    }
    __syncthreads();
  } else {
    // from the L2-resident workspace: columns contiguous (a pair's 8 lanes read whole cache lines), for A through a
    // column-major copy in `scratch` when its columns are strided, for V by accumulating V^T and transposing at the end
#ifndef QK_BUILD_NO_COLMAJOR
    cd* S = A;
    long srs = rs, scs = cs;
    if (rs != 1) {
      for (int e = threadIdx.x; e < p * q; e += BT) {
        const int i = e / q, jc = e - i * q;
        scratch[(long)jc * p + i] = A[i * rs + jc * cs];
      }
      __syncthreads();
      S = scratch, srs = 1, scs = p;
    }
    jacobi_orth(S, srs, scs, p, q, V, 1, q, sig, ord, sh, error);
    if (rs != 1) {
      for (int e = threadIdx.x; e < p * q; e += BT) {
        const int i = e / q, jc = e - i * q;
        A[i * rs + jc * cs] = scratch[(long)jc * p + i];
      }
    }
    for (int e = threadIdx.x; e < q * q; e += BT) {  // V^T -> V in place
      const int i = e / q, jc = e - i * q;
      if (i < jc) {
        const cd a = V[i * q + jc], b = V[jc * q + i];
        V[i * q + jc] = b;
        V[jc * q + i] = a;
      }
    }
    __syncthreads();
#else
    jacobi_orth(A, rs, cs, p, q, V, q, 1, sig, ord, sh, error);
#endif
  }
}

// thread 0: how many leading (sorted) singular values survive (qk_builder.cpp: kept(), mps.py:_kept); results in sh
__device__ void wg_kept(const double* sig, const int* ord, const int n, const double budget, const double zero, WgShared* sh, const int chi = 0) {
  if (threadIdx.x == 0) {
    double total = 0;
    for (int i = 0; i < n; ++i) total += sig[i] * sig[i];
    int keep = 0;
    for (int i = 0; i < n; ++i) keep += (sig[i] > zero);
    keep = max(keep, 1);
    double tail = 0;
    int drop = 0;
    for (int i = keep - 1; i >= 0; --i) {
      const double s = sig[ord[i]];
      tail += s * s;
      if (tail <= budget * total) ++drop;
      else break;
    }
    keep = max(keep - drop, 1);
    if (chi > 0) keep = min(keep, chi);  // the bond cap (QK_BUILD_TRUNCATE): its cost goes into the fidelity below
    double w = 0;
    for (int i = 0; i < keep; ++i) {
      const double s = sig[ord[i]];
      w += s * s;
    }
    sh->keep = keep;
    sh->frac = (total > 0) ? w / total : 1.0;
    sh->nrm = sqrt(w);
  }
  __syncthreads();
}

template <int MINWG>  // resident workgroups per CU the register budget is cut for: 256 threads: 2 (76 KiB of LDS each) or 4 (38 KiB); 512 threads: 1 (152 KiB)
__global__ __launch_bounds__(BT, MINWG) void qk_build_kernel(const BuildArgs g) {
  extern __shared__ double sh_raw[];
  const int n = g.n_qubits, cap = g.cap, tid = threadIdx.x;
  double* sig = sh_raw;                                   // [2 cap]
  int* ord = reinterpret_cast<int*>(sig + 2 * cap);      // [2 cap]
  int* dims = ord + 2 * cap;                              // [n + 1]
  cd* const jl = reinterpret_cast<cd*>(sh_raw + g.jl_offset);  // LDS working set of the Jacobi factorisations
  __shared__ WgShared sh;
  const long slot = 2L * cap * cap, wslot = (long)(2 * cap + 32) * (2 * cap + 32);  // (the block layouts pad rows to 32, columns to 16)
  cd* const sites = g.arena + (long)blockIdx.x * n * slot;
  const long wtab = (((long)(2 * cap + 32) / NBC) * ((long)(2 * cap + 32) / NBC) + 3) / 4;  // the clean-pair table of the block sweeps (ints, in units of a complex)
  cd* const TH = g.work + (long)blockIdx.x * (4 * wslot + wtab);
  cd* const VV = TH + wslot;
  cd* const TMP = VV + wslot;
  cd* const LBUF = g.block ? TMP + wslot : nullptr;  // L = R^H of the preconditioned factorisation (null: QK_BUILD_BLOCK=0, the scalar kernel everywhere)
  const double sqrt_half = 0.7071067811865476;
  const long long wg_begin = wall_clock64();
  for (;;) {
    if (tid == 0) sh.state = (int)atomicAdd(g.counter, 1ull);
    __syncthreads();
    const int slot_no = sh.state;
    __syncthreads();
    const int st = (slot_no < g.n_states) ? g.order[slot_no] : slot_no;
    if (st >= g.n_states) {
      if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(g.error + 12), (unsigned long long)(wall_clock64() - wg_begin));  // busy ticks
      break;
    }
    for (int k = tid; k <= n; k += BT) dims[k] = 1;
    for (int k = tid; k < n; k += BT) {
      sites[k * slot] = cd{1.0, 0.0};
      sites[k * slot + 1] = cd{0.0, 0.0};
    }
    __syncthreads();
    const long long st_begin = wall_clock64();
    const double* alpha = g.alpha + (long)st * g.n_ops;
    double fidelity = 1.0;
    int centre = 0;
    bool outgrown = false;
    for (int i = 0; i < g.n_ops && !outgrown; ++i) {
      const int o = g.op[i], q = g.q0[i];
      if (q < 0 || q >= n || (o >= OP_XX && q + 1 >= n) || o < 0 || o > OP_SWAP) {
        if (tid == 0) atomicOr(g.error, ERR_GATE);
        continue;
      }
      if (o == OP_H || o == OP_RZ) {
        cd* t = sites + q * slot;
        const int l = dims[q], r = dims[q + 1];
        const double th = 0.5 * M_PI * alpha[i];
        const cd ph = cd{cos(th), sin(th)};
        for (int e = tid; e < l * r; e += BT) {
          const int a = e / r, c = e - a * r;
          const cd t0 = t[(a * 2) * r + c], t1 = t[(a * 2 + 1) * r + c];
          if (o == OP_H) {
            t[(a * 2) * r + c] = cd{(t0.x + t1.x) * sqrt_half, (t0.y + t1.y) * sqrt_half};
            t[(a * 2 + 1) * r + c] = cd{(t0.x - t1.x) * sqrt_half, (t0.y - t1.y) * sqrt_half};
          } else {
            t[(a * 2) * r + c] = cmul(t0, cd{ph.x, -ph.y});
            t[(a * 2 + 1) * r + c] = cmul(t1, ph);
          }
        }
        __syncthreads();
        continue;
      }
      // ---- two-qubit gate on (q, q+1): bring the orthogonality centre onto the pair
      while (centre < q) {  // t = (W/s), next <- (s V^H) next
        cd* t = sites + centre * slot;
        cd* u = sites + (centre + 1) * slot;
        const int l = dims[centre], r = dims[centre + 1], r2 = dims[centre + 2];
        const int m = 2 * l;
        if (MINWG <= 2 && LBUF && r >= g_precond_from && r <= m && m <= 64 * MGS_R) {
          // a large site: t = Q R by Gram-Schmidt twice, u <- R2 (R1 u); the bond keeps its size
          mgs2_qr(t, r, 1, m, r, TMP, LBUF, VV, jl, g.jl_elems);
          if (tid == 0) atomicAdd(g.error + 24, 1);
          wg_gemm_mfma(TH, r, 2 * r2, r, LBUF, r, 1, u, 2 * r2, 1);
          wg_gemm_mfma(u, r, 2 * r2, r, VV, r, 1, TH, 2 * r2, 1);
          for (long e = tid; e < (long)m * r; e += BT) {
            const int row = (int)(e / r), c = (int)(e - (long)row * r);
            t[e] = TMP[(long)c * m + row];
          }
          __syncthreads();
          ++centre;
          continue;
        }
        jacobi_auto<MINWG>(t, r, 1, m, r, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP, LBUF, wslot);
        if (tid == 0) {
          int k = 0;
          const double smax = sig[ord[0]];
          for (int jj = 0; jj < r; ++jj) k += (sig[ord[jj]] > 1e-15 * smax);
          sh.keep = max(k, 1);
        }
        __syncthreads();
        const int k = sh.keep;
        for (int e = tid; e < m * k; e += BT) {  // Q[row][jj] = W[row][ord jj] / s
          const int row = e / k, jj = e - row * k;
          const int c = ord[jj];
          const double s = sig[c];
          const cd w = t[row * r + c];
          TMP[e] = (s > 0) ? cd{w.x / s, w.y / s} : cd{0.0, 0.0};
        }
        // R[jj][c] = s_jj conj(V[c][ord jj]); u'[jj][x] = sum_c R[jj][c] u[c][x]
        __syncthreads();
        wg_gemm<true, false>(TH, k, 2 * r2, r, VV, 1, r, u, 2 * r2, 1, ord, sig);  // A(jj, c) = conj(V[c][ord jj]), row scale s
        wg_copy(t, TMP, (long)m * k);
        wg_copy(u, TH, (long)k * 2 * r2);
        if (tid == 0) dims[centre + 1] = k;
        __syncthreads();
        ++centre;
      }
      while (centre > q + 1) {  // t^T = (W/s)(s V^H): t <- (W/s)^T, previous <- previous (s V^H)^T
        cd* t = sites + centre * slot;
        cd* d = sites + (centre - 1) * slot;
        const int l = dims[centre], r = dims[centre + 1], l0 = dims[centre - 1];
        const int w = 2 * r;
        // A(i = (p, c), j = a) = t[a][i]: rs = 1, cs = w
        if (MINWG <= 2 && LBUF && l >= g_precond_from && l <= w && w <= 64 * MGS_R) {
          // a large site: t^T = Q R by Gram-Schmidt twice: t <- Q^T, d <- d R^T = (d R1^T) R2^T ... in the order R = R2 R1: d R^T = (d R1^T) R2^T
          mgs2_qr(t, 1, w, w, l, TMP, LBUF, VV, jl, g.jl_elems);
          if (tid == 0) atomicAdd(g.error + 24, 1);
          wg_gemm_mfma(TH, 2 * l0, l, l, d, l, 1, LBUF, 1, l);  // B(a, jj) = R1[jj][a]
          wg_gemm_mfma(d, 2 * l0, l, l, TH, l, 1, VV, 1, l);    // B(a, jj) = R2[jj][a]
          for (long e = tid; e < (long)l * w; e += BT) t[e] = TMP[e];     // t'[jj][i] = Q(i, jj): Q's column jj, contiguous in S
          __syncthreads();
          --centre;
          continue;
        }
        jacobi_auto<MINWG>(t, 1, w, w, l, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP, LBUF, wslot);
        if (tid == 0) {
          int k = 0;
          const double smax = sig[ord[0]];
          for (int jj = 0; jj < l; ++jj) k += (sig[ord[jj]] > 1e-15 * smax);
          sh.keep = max(k, 1);
        }
        __syncthreads();
        const int k = sh.keep;
        for (int e = tid; e < k * w; e += BT) {  // t'[jj][i] = W(i, ord jj) / s = t[ord jj][i] / s
          const int jj = e / w, ii = e - jj * w;
          const int c = ord[jj];
          const double s = sig[c];
          const cd x = t[c * w + ii];
          TMP[e] = (s > 0) ? cd{x.x / s, x.y / s} : cd{0.0, 0.0};
        }
        // t^T = Q R with R[jj][a] = s_jj conj(V[a][ord jj]);  d'[i][jj] = sum_a d[i][a] R[jj][a]
        __syncthreads();
        wg_gemm<false, true>(TH, 2 * l0, k, l, d, l, 1, VV, l, 1, nullptr, nullptr, ord, sig);  // B(a, jj) = conj(V[a][ord jj]), column scale s
        wg_copy(t, TMP, (long)k * w);
        wg_copy(d, TH, (long)2 * l0 * k);
        if (tid == 0) dims[centre] = k;
        __syncthreads();
        --centre;
      }
      cd* a0 = sites + q * slot;
      cd* a1 = sites + (q + 1) * slot;
      const int l = dims[q], mid = dims[q + 1], r = dims[q + 2];
      const int m = 2 * l, nn = 2 * r;
      if (mid >= 32) wg_gemm_mfma(TH, m, nn, mid, a0, mid, 1, a1, nn, 1);  // theta[(a,p)][(p',c)]: on the matrix cores from bond 32 on
      else wg_gemm<false, false>(TH, m, nn, mid, a0, mid, 1, a1, nn, 1);
      {
        const double th = 0.5 * M_PI * alpha[i];
        const double cs = cos(th), sn = sin(th);
        for (int e = tid; e < l * r; e += BT) {
          const int a = e / r, c = e - a * r;
          cd* p00 = TH + (long)(a * 2) * nn + c;
          cd* p01 = p00 + r;
          cd* p10 = TH + (long)(a * 2 + 1) * nn + c;
          cd* p11 = p10 + r;
          const cd t00 = *p00, t01 = *p01, t10 = *p10, t11 = *p11;
          if (o == OP_SWAP) {
            *p01 = t10;
            *p10 = t01;
          } else {  // XXPhase: cos(th) 1 - i sin(th) X(x)X ;  -i sn * (x + i y) = sn y - i sn x
            *p00 = cd{cs * t00.x + sn * t11.y, cs * t00.y - sn * t11.x};
            *p01 = cd{cs * t01.x + sn * t10.y, cs * t01.y - sn * t10.x};
            *p10 = cd{cs * t10.x + sn * t01.y, cs * t10.y - sn * t01.x};
            *p11 = cd{cs * t11.x + sn * t00.y, cs * t11.y - sn * t00.x};
          }
        }
        __syncthreads();
      }
      // ---- SVD of theta[m x nn] by one-sided Jacobi on its smaller side
      const bool cols = (nn <= m);
      const int qd = cols ? nn : m;
      if (cols) jacobi_auto<MINWG>(TH, nn, 1, m, nn, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP, LBUF, wslot);
      else jacobi_auto<MINWG>(TH, 1, nn, nn, m, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP, LBUF, wslot);
      wg_kept(sig, ord, qd, g.budget, g.zero, &sh, g.truncate ? cap : 0);
      int keep = sh.keep;
      fidelity *= sh.frac;
      const double nrm = sh.nrm;
      if (keep > cap) {
        outgrown = true;  // the program of this state stops after this gate
        if (tid == 0 && !g.partial) atomicOr(g.error, ERR_BOND);
        keep = cap;
      }
      int nxt = q;
      for (int j2 = i + 1; j2 < g.n_ops; ++j2)
        if (g.op[j2] >= OP_XX) {
          nxt = g.q0[j2];
          break;
        }
      const bool centre_right = (nxt >= q + 1) || (nxt == q);
      // theta = U S Vh.  cols: U = W/s, Vh = V^H.  rows (theta^T = W V^H): U = conj(V), Vh[j][c] = W(c, j)/s = TH[j][c]/s.
      for (int e = tid; e < m * keep; e += BT) {
        const int row = e / keep, jj = e - row * keep;
        const int c = ord[jj];
        const double s = sig[c];
        cd v;
        if (cols) {
          const cd w = TH[(long)row * nn + c];
          v = (s > 0) ? cd{w.x / s, w.y / s} : cd{0.0, 0.0};
        } else {
          const cd w = VV[row * m + c];
          v = cd{w.x, -w.y};
        }
        const double f = centre_right ? 1.0 : s / nrm;
        a0[e] = cd{v.x * f, v.y * f};
      }
      for (int e = tid; e < keep * nn; e += BT) {
        const int jj = e / nn, col = e - jj * nn;
        const int c = ord[jj];
        const double s = sig[c];
        cd v;
        if (cols) {
          const cd w = VV[col * nn + c];
          v = cd{w.x, -w.y};
        } else {
          const cd w = TH[(long)c * nn + col];
          v = (s > 0) ? cd{w.x / s, w.y / s} : cd{0.0, 0.0};
        }
        const double f = centre_right ? s / nrm : 1.0;
        a1[e] = cd{v.x * f, v.y * f};
      }
      if (tid == 0) dims[q + 1] = keep;
      __syncthreads();
      centre = centre_right ? q + 1 : q;
    }
    // ---- pack the finished state into the heap
    if (tid == 0) {
      unsigned long long total = 0;
      if (outgrown) fidelity = -1.0;  // marks a dropped state; it gets no tensors
      else
        for (int k = 0; k < n; ++k) total += 2ull * dims[k] * dims[k + 1];
      const unsigned long long off = atomicAdd(g.heap_top, total);
      sh.off = off;
      sh.flag = (off + total <= g.heap_cap);
      if (!sh.flag) atomicOr(g.error, ERR_HEAP);
      g.offs_out[st] = (long long)off;
      g.fid_out[st] = fidelity;
      g.secs_out[st] = (double)(wall_clock64() - st_begin) * 1e-8;
    }
    __syncthreads();
    for (int k = tid; k <= n; k += BT) g.dims_out[(long)st * (n + 1) + k] = dims[k];
    if (sh.flag && !outgrown) {
      unsigned long long pos = sh.off;
      for (int k = 0; k < n; ++k) {
        const long cnt = 2L * dims[k] * dims[k + 1];
        const cd* src = sites + k * slot;
        for (long e = tid; e < cnt; e += BT) g.heap[pos + e] = src[e];
        pos += cnt;
      }
    }
    __syncthreads();
  }
}

// ---- debug: one Jacobi factorisation of a host matrix (tests the primitive on its own)
__global__ __launch_bounds__(BT) void qk_jacobi_kernel(cd* A, int p, int q, cd* V, double* sig_out, int* ord_out, int* error) {
  extern __shared__ double sh_raw[];
  double* sig = sh_raw;
  int* ord = reinterpret_cast<int*>(sig + q);
  __shared__ WgShared sh;
  jacobi_orth(A, q, 1, p, q, V, q, 1, sig, ord, &sh, error);
  for (int e = threadIdx.x; e < q; e += BT) {
    sig_out[e] = sig[e];
    ord_out[e] = ord[e];
  }
}
// ---- debug: the preconditioned block factorisation of one host matrix
__global__ __launch_bounds__(BT, BT > 256 ? 1 : 2) void qk_jacobi_precond_kernel(cd* A, int p, int q, cd* V, double* sig_out, int* ord_out, int* error, cd* S, cd* LB, int lds_elems, int* chk) {
  extern __shared__ double sh_raw[];
  double* sig = sh_raw;
  int* ord = reinterpret_cast<int*>(sig + q);
  cd* lds = reinterpret_cast<cd*>(sh_raw + ((q * 12 + 15) / 16) * 2 + 2);  // behind sig / ord, 16-byte aligned
  __shared__ WgShared sh;
  jacobi_precond<(BT > 256 ? 1 : 2)>(A, q, 1, p, q, V, sig, ord, &sh, error, lds, lds_elems, S, LB, chk);
  for (int e = threadIdx.x; e < q; e += BT) {
    sig_out[e] = sig[e];
    ord_out[e] = ord[e];
  }
}
}  // namespace QKB_NS
