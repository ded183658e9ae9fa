/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED by the reference.
 *
 * The same CPU hot loop as overlap_ref.c
 *     tile[i, j] = abs(inner(y_mps[i], x_mps[j]))^2      (KernelPkg/src/KernelPkg.jl:101-109)
 * with every site's two contractions done by BLAS zgemm -- which is what ITensors' `inner` (KernelPkg.jl:106)
 * runs on (NDTensors contracts through BLAS).  This is the FAIR CPU baseline of bench.py ("kind": "port",
 * "blas": the OpenBLAS that scipy ships, resolved at run time with dlopen); overlap_ref.c is the hand-written
 * loop kept as a second opinion for the tests.  One pair per OpenMP thread, single-threaded BLAS inside.
 *
 *   E_0 = 1,  T[L, (p, r)] = sum_l E[L, l] B[l, (p, r)],   E'[R, r] = sum_{(L, p)} conj(A[(L, p), R]) T[(L, p), r]
 *
 * Tensors: complex128, C order [chi_l][2][chi_r], re/im interleaved.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef void (*zgemm_t)(const char*, const char*, const int*, const int*, const int*, const double*, const double*, const int*, const double*,
                        const int*, const double*, double*, const int*);
static zgemm_t p_zgemm = 0;
typedef void (*set_threads_t)(int);

/* resolve zgemm from the OpenBLAS at `path` (scipy's: symbols carry a scipy_ prefix); 0 on success */
int qkob_init(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return -1;
  p_zgemm = (zgemm_t)dlsym(h, "scipy_zgemm_");
  if (!p_zgemm) p_zgemm = (zgemm_t)dlsym(h, "zgemm_");
  set_threads_t st = (set_threads_t)dlsym(h, "scipy_openblas_set_num_threads");
  if (!st) st = (set_threads_t)dlsym(h, "openblas_set_num_threads");
  if (st) st(1); /* parallelism is over pairs */
  return p_zgemm ? 0 : -2;
}

/* row-major C[M x N] = op(A) B, computed as the column-major product C^T = B^T op(A)^T */
static void sweep(int n_sites, const int32_t* a, const double* const* A, const int32_t* b, const double* const* B, double* E, double* T, double* F,
                  double* out) {
  const double one[2] = {1.0, 0.0}, zero[2] = {0.0, 0.0};
  E[0] = 1.0, E[1] = 0.0;
  for (int k = 0; k < n_sites; ++k) {
    const int a0 = a[k], a1 = a[k + 1], b0 = b[k], b1 = b[k + 1];
    const int n1 = 2 * b1, k2 = 2 * a0;
    /* T[a0 x 2 b1] = E[a0 x b0] B[b0 x 2 b1] */
    p_zgemm("N", "N", &n1, &a0, &b0, one, B[k], &n1, E, &b0, zero, T, &n1);
    /* E'[a1 x b1] = A_mat^H T_mat,  A_mat = A[(a0, p) x a1],  T_mat = T[(a0, p) x b1] */
    p_zgemm("N", "C", &b1, &a1, &k2, one, T, &b1, A[k], &a1, zero, F, &b1);
    double* t = E;
    E = F, F = t;
  }
  out[0] = E[0], out[1] = E[1];
}

/* Same interface as qko_gram_pairs (overlap_ref.c).  Returns the number of threads used, < 0 on failure. */
int qkob_gram_pairs(int n_sites, int nx, const int32_t* xdims, const double* const* xt, int ny, const int32_t* ydims, const double* const* yt,
                    int64_t npairs, const int32_t* pairs, double* values, double* z, int threads) {
  if (!p_zgemm) return -3;
  if (!ydims) ydims = xdims, yt = xt, ny = nx;
  int amax = 1, bmax = 1;
  for (int64_t i = 0; i < (int64_t)nx * (n_sites + 1); ++i) amax = xdims[i] > amax ? xdims[i] : amax;
  for (int64_t i = 0; i < (int64_t)ny * (n_sites + 1); ++i) bmax = ydims[i] > bmax ? ydims[i] : bmax;
  int used = 1, fail = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    double* E = (double*)malloc(sizeof(double) * 2 * (size_t)amax * bmax);
    double* F = (double*)malloc(sizeof(double) * 2 * (size_t)amax * bmax);
    double* T = (double*)malloc(sizeof(double) * 4 * (size_t)amax * bmax);
    if (!E || !F || !T) {
#pragma omp atomic write
      fail = 1;
    } else {
#pragma omp for schedule(dynamic, 4)
      for (int64_t t = 0; t < npairs; ++t) {
        const int i = pairs[2 * t], j = pairs[2 * t + 1];
        double zz[2];
        sweep(n_sites, xdims + (int64_t)i * (n_sites + 1), xt + (int64_t)i * n_sites, ydims + (int64_t)j * (n_sites + 1), yt + (int64_t)j * n_sites, E, T, F, zz);
        values[t] = zz[0] * zz[0] + zz[1] * zz[1];
        if (z) z[2 * t] = zz[0], z[2 * t + 1] = zz[1];
      }
    }
    free(E), free(F), free(T);
  }
  return fail ? -4 : used;
}
