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
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
typedef double cd __attribute__((ext_vector_type(2)));  // complex128 as (re, im); a native vector so that LDS-typed pointers work
#ifndef QK_BUILD_BT
#define QK_BUILD_BT 256
#endif
constexpr int BT = QK_BUILD_BT;  // threads per workgroup (4 wavefronts, one per SIMD; 512 x 16 lanes per pair measured 1.3x slower)
#ifndef QK_BUILD_GL
#define QK_BUILD_GL 8
#endif
constexpr int GL = QK_BUILD_GL;  // lanes that share one column pair
constexpr int NG = BT / GL;   // column pairs per step
constexpr int MAX_SWEEPS = 40;
enum { OP_H = 0, OP_RZ = 1, OP_XX = 2, OP_SWAP = 3 };  // ansatz.py
enum { ERR_BOND = 1, ERR_HEAP = 2, ERR_SWEEPS = 4, ERR_GATE = 8 };

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

struct BuildArgs {
  int n_states, n_qubits, n_ops, cap;
  const int8_t* op;
  const int32_t* q0;
  const double* alpha;  // [n_states][n_ops] half-turns
  double budget, zero;
  cd* arena;  // per workgroup: n_qubits slots of 2 cap^2
  cd* work;   // per workgroup: 3 buffers of 4 cap^2
  cd* heap;
  unsigned long long heap_cap;
  unsigned long long* heap_top;
  int32_t* dims_out;    // [n_states][n_qubits + 1]
  double* fid_out;      // [n_states]
  long long* offs_out;  // [n_states] complex elements into heap
  unsigned long long* counter;
  int* error;     // [0] error bits, [1..4] Jacobi statistics: factorisations, sweeps, most sweeps, unconverged
  int jl_offset;  // doubles from the start of the dynamic LDS to the Jacobi working set
  int jl_elems;   // complex elements it holds
  int partial;    // a state that outgrows cap is dropped (fidelity -1) instead of failing the call
  const int32_t* order;  // queue position -> state index: states expected to be expensive first
};

// Shared scalars of a workgroup (one instance in LDS).
struct WgShared {
  int flag, keep, state, pad;
  double frac, nrm;
  unsigned long long off;
  unsigned long long worst;  // bits of the largest squared relative inner product rotated in the current sweep
};

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

// The same factorisation with the working set in LDS when it fits (A and V side by side, odd leading dimension so
// that the 16 lanes of a pair hit 16 different banks): a step is then a few hundred cycles instead of a store-drain +
// L2 round trip.  Results are copied back to the global A (same strides) and to V (row-major, ld q).
// (Inlined into the kernel so that the kernel's register budget -- MINWG workgroups per CU -- governs it; the variant that
// keeps 8 rows per lane in registers exists only at 2 workgroups per CU.)
template <int MINWG>
__device__ __forceinline__ void jacobi_auto(cd* A, const long rs, const long cs, const int p, const int q, cd* V, double* sig, int* ord, WgShared* sh,
                                            int* error, cd* lds, const int lds_elems, cd* scratch) {
  const int ld = q | 1;
  if (threadIdx.x == 0) atomicAdd(error + (((long)(p + q) * ld <= lds_elems) ? 5 : 6), 1);  // statistics: LDS-resident / L2-resident
  if ((long)(p + q) * ld <= lds_elems) {
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

__device__ void wg_copy(cd* __restrict__ dst, const cd* __restrict__ src, const long n) {
  for (long e = threadIdx.x; e < n; e += BT) dst[e] = src[e];
  __syncthreads();
}

// thread 0: how many leading (sorted) singular values survive (qk_builder.cpp: kept(), mps.py:_kept); results in sh
__device__ void wg_kept(const double* sig, const int* ord, const int n, const double budget, const double zero, WgShared* sh) {
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

template <int MINWG>  // resident workgroups per CU the register budget is cut for: 2 (76 KiB of LDS each) or 4 (38 KiB)
__global__ __launch_bounds__(BT, MINWG) void qk_build_kernel(const BuildArgs g) {
  extern __shared__ double sh_raw[];
  const int n = g.n_qubits, cap = g.cap, tid = threadIdx.x;
  double* sig = sh_raw;                                   // [2 cap]
  int* ord = reinterpret_cast<int*>(sig + 2 * cap);      // [2 cap]
  int* dims = ord + 2 * cap;                              // [n + 1]
  cd* const jl = reinterpret_cast<cd*>(sh_raw + g.jl_offset);  // LDS working set of the Jacobi factorisations
  __shared__ WgShared sh;
  const long slot = 2L * cap * cap, wslot = 4L * cap * cap;
  cd* const sites = g.arena + (long)blockIdx.x * n * slot;
  cd* const TH = g.work + (long)blockIdx.x * 3 * wslot;
  cd* const VV = TH + wslot;
  cd* const TMP = VV + wslot;
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
        jacobi_auto<MINWG>(t, r, 1, m, r, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP);
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
        jacobi_auto<MINWG>(t, 1, w, w, l, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP);
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
      wg_gemm<false, false>(TH, m, nn, mid, a0, mid, 1, a1, nn, 1);  // theta[(a,p)][(p',c)]
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
      if (cols) jacobi_auto<MINWG>(TH, nn, 1, m, nn, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP);
      else jacobi_auto<MINWG>(TH, 1, nn, nn, m, VV, sig, ord, &sh, g.error, jl, g.jl_elems, TMP);
      wg_kept(sig, ord, qd, g.budget, g.zero, &sh);
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
}  // namespace

struct qk_built {
  qk_ctx* ctx = nullptr;
  int n_states = 0, n_qubits = 0;
  cd* heap = nullptr;
  std::vector<int32_t> dims;
  std::vector<double> fidelity;
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
  int wgs_variant = (cap <= 32) ? 4 : 2;
  if (const char* v = std::getenv("QK_BUILD_WGS")) wgs_variant = (std::atoi(v) >= 4) ? 4 : 2;
  size_t lds_total = (wgs_variant == 4 ? 38 : 76) * 1024;
  if (const char* v = std::getenv("QK_BUILD_LDS_KB")) lds_total = (size_t)std::max(32, std::min(wgs_variant == 4 ? 38 : 156, std::atoi(v))) * 1024;
  const int jl_elems = (int)((lds_total - lds_meta) / sizeof(cd));
  const size_t lds = lds_total;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_build_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(qk_build_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 38 * 1024));
  const int wgs_per_cu = std::min(wgs_variant, (int)((160 * 1024) / (lds_total + 1024)));  // + the static LDS of the kernel
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t per_wg = ((size_t)n_qubits * 2 * cap * cap + (size_t)3 * 4 * cap * cap) * sizeof(cd);
  long long grid = std::min<long long>(n_states, (long long)wgs_per_cu * c->num_cus);
  grid = std::min<long long>(grid, (long long)(0.35 * (double)free_b / (double)per_wg));
  if (grid < 1) return qk_fail(QK_EDEVICE, "qk_build_mps: not enough device memory for one workgroup's arena (%zu bytes)", per_wg);
  // heap: every finished state, packed; bounded by the arena size of all states and by the free memory
  const double heap_want = (double)n_states * (double)n_qubits * 2.0 * cap * cap;
  const size_t heap_cap = (size_t)std::min(heap_want, 0.45 * (double)free_b / (double)sizeof(cd));
  cd *arena = nullptr, *work = nullptr, *heap = nullptr;
  int8_t* d_op = nullptr;
  int32_t* d_q0 = nullptr;
  double *d_alpha = nullptr, *d_fid = nullptr;
  int32_t* d_dims = nullptr;
  long long* d_offs = nullptr;
  unsigned long long* d_ctr = nullptr;  // [0] state counter, [1] heap top
  int* d_err = nullptr;
  int32_t* d_order = nullptr;
  auto release = [&]() {
    (void)hipFree(arena), (void)hipFree(work), (void)hipFree(d_op), (void)hipFree(d_q0), (void)hipFree(d_alpha), (void)hipFree(d_fid);
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
  BUILD_TRY(hipMalloc(&arena, (size_t)grid * n_qubits * 2 * cap * cap * sizeof(cd)));
  BUILD_TRY(hipMalloc(&work, (size_t)grid * 3 * 4 * cap * cap * sizeof(cd)));
  BUILD_TRY(hipMalloc(&heap, heap_cap * sizeof(cd)));
  BUILD_TRY(hipMalloc(&d_op, std::max(1, n_ops)));
  BUILD_TRY(hipMalloc(&d_q0, (size_t)std::max(1, n_ops) * sizeof(int32_t)));
  BUILD_TRY(hipMalloc(&d_alpha, (size_t)n_states * std::max(1, n_ops) * sizeof(double)));
  BUILD_TRY(hipMalloc(&d_fid, (size_t)n_states * sizeof(double)));
  BUILD_TRY(hipMalloc(&d_dims, (size_t)n_states * (n_qubits + 1) * sizeof(int32_t)));
  BUILD_TRY(hipMalloc(&d_offs, (size_t)n_states * sizeof(long long)));
  BUILD_TRY(hipMalloc(&d_ctr, 2 * sizeof(unsigned long long)));
  BUILD_TRY(hipMalloc(&d_err, 16 * sizeof(int)));
  if (n_ops > 0) {
    BUILD_TRY(hipMemcpyAsync(d_op, op, n_ops, hipMemcpyHostToDevice, c->stream));
    BUILD_TRY(hipMemcpyAsync(d_q0, q0, (size_t)n_ops * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    BUILD_TRY(hipMemcpyAsync(d_alpha, alpha, (size_t)n_states * n_ops * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  BUILD_TRY(hipMemsetAsync(d_ctr, 0, 2 * sizeof(unsigned long long), c->stream));
  BUILD_TRY(hipMemsetAsync(d_err, 0, 16 * sizeof(int), c->stream));
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
  a.dims_out = d_dims, a.fid_out = d_fid, a.offs_out = d_offs, a.counter = d_ctr, a.error = d_err;
  a.jl_offset = (int)(lds_meta / sizeof(double)), a.jl_elems = jl_elems;
  a.partial = (flags & QK_BUILD_PARTIAL) ? 1 : 0;
  BUILD_TRY(hipEventRecord(c->ev0, c->stream));
  if (wgs_variant == 4) qk_build_kernel<4><<<dim3((unsigned)grid), dim3(BT), lds, c->stream>>>(a);
  else qk_build_kernel<2><<<dim3((unsigned)grid), dim3(BT), lds, c->stream>>>(a);
  BUILD_TRY(hipGetLastError());
  BUILD_TRY(hipEventRecord(c->ev1, c->stream));
  qk_built* b = new qk_built;
  b->ctx = c, b->n_states = n_states, b->n_qubits = n_qubits;
  b->dims.resize((size_t)n_states * (n_qubits + 1));
  b->fidelity.resize(n_states);
  b->offsets.resize(n_states);
  std::vector<long long> offs(n_states);
  int errv[16] = {0};
  unsigned long long ctr[2] = {0, 0};
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(b->dims.data(), d_dims, b->dims.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(b->fidelity.data(), d_fid, (size_t)n_states * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(offs.data(), d_offs, (size_t)n_states * sizeof(long long), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(errv, d_err, sizeof(errv), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(ctr, d_ctr, sizeof(ctr), hipMemcpyDeviceToHost);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
  release();
  if (e != hipSuccess) {
    (void)hipFree(heap);
    delete b;
    return qk_fail(QK_EDEVICE, "qk_build_mps: %s", hipGetErrorString(e));
  }
  const int err = errv[0];
  if (std::getenv("QK_BUILD_DEBUG"))
    std::fprintf(stderr, "[qk_build_mps] %d states, grid %lld, %.1f ms; Jacobi: %d factorisations, %.2f sweeps on average, %d at most, %d unconverged, %d in LDS / %d from L2; error bits %d\n",
                 n_states, grid, ms, errv[1], errv[1] ? (double)errv[2] / errv[1] : 0.0, errv[3], errv[4], errv[5], errv[6], err);
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
  HIP_TRY(hipMalloc(&dE, 16 * sizeof(int)));
  HIP_TRY(hipMemset(dE, 0, 16 * sizeof(int)));
  HIP_TRY(hipMemcpy(dA, a_inout, (size_t)p * q * sizeof(cd), hipMemcpyHostToDevice));
  qk_jacobi_kernel<<<dim3(1), dim3(BT), (size_t)q * (sizeof(double) + sizeof(int)) + 16, c->stream>>>(dA, p, q, dV, dS, dO, dE);
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
