/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED by the reference.
 *
 * Plain-C restatement of the reference's CPU hot loop
 *     tile[i, j] = abs(inner(y_mps[i], x_mps[j]))^2      (KernelPkg/src/KernelPkg.jl:101-109)
 * which is also what the GPU backend computes per entry
 *     overlap = x_mps.vdot(y_mps); entry = (overlap*conj(overlap)).real
 *                                                         (gpu_backend/kernel_state_ansatz.py:380-383)
 * `inner`/`vdot` is the site-by-site transfer-matrix sweep
 *     E_0 = 1,  E_{k+1}[R,r] = sum_{L,l,p} E_k[L,l] conj(A_k[L,p,R]) B_k[l,p,r].
 * Used (a) by tests as a second CPU opinion next to oracle/restatement.py and (b) by
 * bench.py's `cpu_baseline` leg, timed on the host cores ("kind": "port").
 * It is never called by the product.
 *
 * Tensors: complex128, C order [chi_l][2][chi_r], re/im interleaved.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  double *re, *im; /* split planes of one site tensor, [chi_l][2][chi_r] */
} site_t;

static void split_planes(const double* z, int64_t n, double* re, double* im) {
  for (int64_t i = 0; i < n; ++i) {
    re[i] = z[2 * i];
    im[i] = z[2 * i + 1];
  }
}

/* <x|y> for split-plane tensors; work arrays sized for the largest bonds */
static void sweep(int n_sites, const int32_t* a, const site_t* A, const int32_t* b, const site_t* B, double* Ere,
                  double* Eim, double* Tre, double* Tim, double* Fre, double* Fim, double* out) {
  Ere[0] = 1.0;
  Eim[0] = 0.0;
  for (int k = 0; k < n_sites; ++k) {
    const int a0 = a[k], a1 = a[k + 1], b0 = b[k], b1 = b[k + 1];
    const int w = 2 * b1;
    /* T[L][(p,r)] = sum_l E[L][l] B[l][(p,r)] */
    memset(Tre, 0, sizeof(double) * (size_t)a0 * w);
    memset(Tim, 0, sizeof(double) * (size_t)a0 * w);
    for (int L = 0; L < a0; ++L) {
      double* tr = Tre + (size_t)L * w;
      double* ti = Tim + (size_t)L * w;
      for (int l = 0; l < b0; ++l) {
        const double er = Ere[(size_t)L * b0 + l], ei = Eim[(size_t)L * b0 + l];
        const double* br = B[k].re + (size_t)l * w;
        const double* bi = B[k].im + (size_t)l * w;
        for (int c = 0; c < w; ++c) {
          tr[c] += er * br[c] - ei * bi[c];
          ti[c] += er * bi[c] + ei * br[c];
        }
      }
    }
    /* E'[R][r] = sum_{L,p} conj(A[L][p][R]) T[L][p][r] */
    memset(Fre, 0, sizeof(double) * (size_t)a1 * b1);
    memset(Fim, 0, sizeof(double) * (size_t)a1 * b1);
    for (int Lp = 0; Lp < 2 * a0; ++Lp) {
      const double* tr = Tre + (size_t)Lp * b1;
      const double* ti = Tim + (size_t)Lp * b1;
      const double* ar = A[k].re + (size_t)Lp * a1;
      const double* ai = A[k].im + (size_t)Lp * a1;
      for (int R = 0; R < a1; ++R) {
        const double cr = ar[R], ci = -ai[R];
        double* fr = Fre + (size_t)R * b1;
        double* fi = Fim + (size_t)R * b1;
        for (int r = 0; r < b1; ++r) {
          fr[r] += cr * tr[r] - ci * ti[r];
          fi[r] += cr * ti[r] + ci * tr[r];
        }
      }
    }
    double* s;
    s = Ere, Ere = Fre, Fre = s;
    s = Eim, Eim = Fim, Fim = s;
  }
  out[0] = Ere[0];
  out[1] = Eim[0];
}

typedef struct {
  int n_states, n_sites, max_bond;
  const int32_t* dims;
  site_t* sites; /* [n_states][n_sites] */
  double* pool;
} set_t;

static int make_set(set_t* s, int n_states, int n_sites, const int32_t* dims, const double* const* tensors) {
  s->n_states = n_states, s->n_sites = n_sites, s->dims = dims, s->max_bond = 1;
  int64_t tot = 0;
  for (int i = 0; i < n_states; ++i)
    for (int k = 0; k < n_sites; ++k) {
      const int32_t* d = dims + (size_t)i * (n_sites + 1);
      tot += (int64_t)d[k] * 2 * d[k + 1];
      if (d[k] > s->max_bond) s->max_bond = d[k];
    }
  s->pool = (double*)malloc(sizeof(double) * 2 * (size_t)tot);
  s->sites = (site_t*)malloc(sizeof(site_t) * (size_t)n_states * n_sites);
  if (!s->pool || !s->sites) return -1;
  double* p = s->pool;
  for (int i = 0; i < n_states; ++i)
    for (int k = 0; k < n_sites; ++k) {
      const int32_t* d = dims + (size_t)i * (n_sites + 1);
      const int64_t n = (int64_t)d[k] * 2 * d[k + 1];
      site_t* st = &s->sites[(size_t)i * n_sites + k];
      st->re = p, st->im = p + n;
      split_planes(tensors[(size_t)i * n_sites + k], n, st->re, st->im);
      p += 2 * n;
    }
  return 0;
}

static void free_set(set_t* s) {
  free(s->pool);
  free(s->sites);
}

/* values[t] = |<x_i|y_j>|^2 and z[2t..2t+1] = <x_i|y_j> for pairs[t] = (i, j).
 * ytensors == NULL means Y is X.  Returns the number of threads used (<0 on error). */
int qko_gram_pairs(int n_sites, int nx, const int32_t* xdims, const double* const* xtensors, int ny,
                   const int32_t* ydims, const double* const* ytensors, int64_t npairs, const int32_t* pairs,
                   double* values, double* z, int nthreads) {
  set_t X, Y;
  if (make_set(&X, nx, n_sites, xdims, xtensors)) return -1;
  const int sym = (ytensors == NULL);
  if (sym)
    Y = X;
  else if (make_set(&Y, ny, n_sites, ydims, ytensors))
    return -1;
  const size_t m = (size_t)(X.max_bond > Y.max_bond ? X.max_bond : Y.max_bond);
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    double* w = (double*)malloc(sizeof(double) * (4 * m * m + 2 * m * 2 * m));
    double *Ere = w, *Eim = w + m * m, *Fre = w + 2 * m * m, *Fim = w + 3 * m * m, *Tre = w + 4 * m * m, *Tim = Tre + 2 * m * m;
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
    for (int64_t t = 0; t < npairs; ++t) {
      const int i = pairs[2 * t], j = pairs[2 * t + 1];
      double o[2];
      sweep(n_sites, xdims + (size_t)i * (n_sites + 1), X.sites + (size_t)i * n_sites, (sym ? xdims : ydims) + (size_t)j * (n_sites + 1),
            Y.sites + (size_t)j * n_sites, Ere, Eim, Tre, Tim, Fre, Fim, o);
      values[t] = o[0] * o[0] + o[1] * o[1];
      if (z) z[2 * t] = o[0], z[2 * t + 1] = o[1];
    }
    free(w);
  }
  free_set(&X);
  if (!sym) free_set(&Y);
  return used;
}
