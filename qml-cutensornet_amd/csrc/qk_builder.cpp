// qk_builder.cpp -- native host builder of the ansatz MPS: the same algorithm as
// qml-cutensornet_amd/mps.py:_simulate (one LAPACK gesdd per two-qubit gate, QR centre moves, ITensors-style cutoff),
// without the Python interpreter between the 10^3 ... 10^4 small LAPACK calls of a circuit.  Plain C++ (g++), no HIP:
// built as libqkbuilder.so next to libqkgram.so.  LAPACK/BLAS come from the OpenBLAS that scipy already ships
// (symbols scipy_zgesdd_, scipy_zgeqrf_, scipy_zungqr_, scipy_zgemm_), resolved at run time with dlopen: the Python
// side passes the library's path (qkb_init).  Reference semantics: gpu_backend/kernel_state_ansatz.py:141-144, 221
// (MPSxGate, truncation_fidelity) and KernelPkg.jl:68 (cutoff).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using cd = std::complex<double>;

namespace {
thread_local std::string g_err;

// Fortran LAPACK / BLAS entry points (LP64 integers)
using zgesdd_t = void (*)(const char*, const int*, const int*, cd*, const int*, double*, cd*, const int*, cd*, const int*, cd*, const int*,
                          double*, int*, int*);
using zgeqrf_t = void (*)(const int*, const int*, cd*, const int*, cd*, cd*, const int*, int*);
using zungqr_t = void (*)(const int*, const int*, const int*, cd*, const int*, const cd*, cd*, const int*, int*);
using zgemm_t = void (*)(const char*, const char*, const int*, const int*, const int*, const cd*, const cd*, const int*, const cd*, const int*,
                         const cd*, cd*, const int*);
zgesdd_t p_zgesdd = nullptr;
zgeqrf_t p_zgeqrf = nullptr;
zungqr_t p_zungqr = nullptr;
zgemm_t p_zgemm = nullptr;

enum { OP_H = 0, OP_RZ = 1, OP_XX = 2, OP_SWAP = 3 };  // ansatz.py

// row-major C[m x n] = A[m x k] * B[k x n]  (as column-major C^T = B^T A^T)
void gemm_rm(int m, int n, int k, const cd* A, const cd* B, cd* C) {
  const cd one(1.0, 0.0), zero(0.0, 0.0);
  p_zgemm("N", "N", &n, &m, &k, &one, B, &n, A, &k, &zero, C, &n);
}

struct Tensor {  // [l][2][r], row-major
  int l = 1, r = 1;
  std::vector<cd> v;
  cd& at(int a, int p, int c) { return v[((size_t)a * 2 + p) * r + c]; }
};

// economic QR of a row-major M[m x n]: Q[m x k], R[k x n], k = min(m, n)  (geqrf + ungqr, as scipy.linalg.qr(mode="economic"))
int qr_economic(int m, int n, const cd* M, std::vector<cd>& Q, std::vector<cd>& R, std::vector<cd>& work, std::vector<cd>& cm) {
  const int k = std::min(m, n);
  cm.resize((size_t)m * n);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) cm[(size_t)j * m + i] = M[(size_t)i * n + j];  // to column-major
  std::vector<cd> tau(k);
  int info = 0, lwork = -1;
  cd wq;
  p_zgeqrf(&m, &n, cm.data(), &m, tau.data(), &wq, &lwork, &info);
  lwork = std::max(1, (int)wq.real());
  if ((int)work.size() < lwork) work.resize(lwork);
  p_zgeqrf(&m, &n, cm.data(), &m, tau.data(), work.data(), &lwork, &info);
  if (info != 0) return info;
  R.assign((size_t)k * n, cd(0, 0));
  for (int i = 0; i < k; ++i)
    for (int j = i; j < n; ++j) R[(size_t)i * n + j] = cm[(size_t)j * m + i];
  lwork = -1;
  p_zungqr(&m, &k, &k, cm.data(), &m, tau.data(), &wq, &lwork, &info);
  lwork = std::max(1, (int)wq.real());
  if ((int)work.size() < lwork) work.resize(lwork);
  p_zungqr(&m, &k, &k, cm.data(), &m, tau.data(), work.data(), &lwork, &info);
  if (info != 0) return info;
  Q.resize((size_t)m * k);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < k; ++j) Q[(size_t)i * k + j] = cm[(size_t)j * m + i];
  return 0;
}

// how many leading singular values survive, and the kept fraction of the weight (mps.py:_kept)
int kept(const std::vector<double>& s, double budget, double zero, double* frac) {
  const int n = (int)s.size();
  double total = 0;
  for (double x : s) total += x * x;
  int keep = 0;
  for (double x : s) keep += (x > zero);
  keep = std::max(keep, 1);
  double tail = 0;
  int drop = 0;
  for (int i = keep - 1; i >= 0; --i) {  // cumulative weight from the small end; count entries <= budget * total
    tail += s[i] * s[i];
    if (tail <= budget * total) ++drop;
    else break;
  }
  keep = std::max(keep - drop, 1);
  double w = 0;
  for (int i = 0; i < keep; ++i) w += s[i] * s[i];
  *frac = w / total;
  (void)n;
  return keep;
}
}  // namespace

extern "C" {

const char* qkb_last_error(void) { return g_err.c_str(); }

// Resolve LAPACK / BLAS from the OpenBLAS shipped with scipy.  Returns 0 on success.
int qkb_init(const char* openblas_path) {
  if (p_zgesdd) return 0;
  void* h = dlopen(openblas_path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    g_err = std::string("dlopen failed: ") + dlerror();
    return -1;
  }
  auto sym = [&](const char* a, const char* b) -> void* {
    void* s = dlsym(h, a);
    return s ? s : dlsym(h, b);
  };
  p_zgesdd = (zgesdd_t)sym("scipy_zgesdd_", "zgesdd_");
  p_zgeqrf = (zgeqrf_t)sym("scipy_zgeqrf_", "zgeqrf_");
  p_zungqr = (zungqr_t)sym("scipy_zungqr_", "zungqr_");
  p_zgemm = (zgemm_t)sym("scipy_zgemm_", "zgemm_");
  if (!p_zgesdd || !p_zgeqrf || !p_zungqr || !p_zgemm) {
    p_zgesdd = nullptr;
    g_err = "the library does not export zgesdd_/zgeqrf_/zungqr_/zgemm_";
    return -2;
  }
  return 0;
}

// MPS of circuit |0...0>.  op/q0/alpha: the bound gate program (ansatz.py: BoundCircuit).  On success fills
// dims_out[n_qubits + 1] and *tensors_out: ONE malloc'd block holding the site tensors back to back, complex128
// [l][2][r] row-major (free it with qkb_free), and *fidelity.  Returns 0, or a negative code with qkb_last_error().
int qkb_simulate_chi(int32_t n_qubits, int32_t n_ops, const int8_t* op, const int32_t* q0, const double* alpha, double trunc_budget,
                     double value_of_zero, int32_t max_bond, int32_t* dims_out, double** tensors_out, int64_t* n_complex_out, double* fidelity_out);
int qkb_simulate(int32_t n_qubits, int32_t n_ops, const int8_t* op, const int32_t* q0, const double* alpha, double trunc_budget,
                 double value_of_zero, int32_t* dims_out, double** tensors_out, int64_t* n_complex_out, double* fidelity_out) {
  return qkb_simulate_chi(n_qubits, n_ops, op, q0, alpha, trunc_budget, value_of_zero, 0, dims_out, tensors_out, n_complex_out, fidelity_out);
}
// max_bond > 0: at most that many singular values survive a gate (the chi of pytket-cutensornet's Config, reference
// gpu_backend/kernel_state_ansatz.py:141-144); the lost weight goes into the fidelity like any other truncation.
int qkb_simulate_chi(int32_t n_qubits, int32_t n_ops, const int8_t* op, const int32_t* q0, const double* alpha, double trunc_budget,
                     double value_of_zero, int32_t max_bond, int32_t* dims_out, double** tensors_out, int64_t* n_complex_out, double* fidelity_out) {
  if (!p_zgesdd) {
    g_err = "qkb_init has not been called";
    return -1;
  }
  const int n = n_qubits;
  std::vector<Tensor> A(n);
  for (auto& t : A) t.v = {cd(1, 0), cd(0, 0)};
  std::vector<int> two_q_pos;
  for (int i = 0; i < n_ops; ++i)
    if (op[i] == OP_XX || op[i] == OP_SWAP) two_q_pos.push_back(q0[i]);
  double fidelity = 1.0;
  int centre = 0;  // sites < centre are left-orthonormal, sites > centre right-orthonormal
  size_t g2 = 0;
  std::vector<cd> Q, R, work, cm, theta, U, VT, tmp;
  std::vector<double> S, rwork;
  std::vector<int> iwork;
  const double sqrt_half = 0.7071067811865476;

  for (int i = 0; i < n_ops; ++i) {
    const int o = op[i], q = q0[i];
    if (q < 0 || q >= n || ((o == OP_XX || o == OP_SWAP) && q + 1 >= n)) {
      g_err = "gate on a qubit outside the register";
      return -3;
    }
    if (o == OP_H) {
      Tensor& t = A[q];
      for (int a = 0; a < t.l; ++a)
        for (int c = 0; c < t.r; ++c) {
          const cd t0 = t.at(a, 0, c), t1 = t.at(a, 1, c);
          t.at(a, 0, c) = (t0 + t1) * sqrt_half;
          t.at(a, 1, c) = (t0 - t1) * sqrt_half;
        }
      continue;
    }
    if (o == OP_RZ) {
      const double th = 0.5 * M_PI * alpha[i];
      const cd ph(std::cos(th), std::sin(th));
      Tensor& t = A[q];
      for (int a = 0; a < t.l; ++a)
        for (int c = 0; c < t.r; ++c) {
          t.at(a, 0, c) *= std::conj(ph);
          t.at(a, 1, c) *= ph;
        }
      continue;
    }
    // ---- two-qubit gate on (q, q+1): bring the orthogonality centre onto the pair
    while (centre < q) {
      Tensor& t = A[centre];
      const int m = t.l * 2, nn = t.r;
      if (qr_economic(m, nn, t.v.data(), Q, R, work, cm) != 0) {
        g_err = "zgeqrf/zungqr failed";
        return -4;
      }
      const int k = std::min(m, nn);
      Tensor& u = A[centre + 1];  // u <- R u : [k][2 r2] = R[k x nn] * u[nn x 2 r2]
      tmp.resize((size_t)k * 2 * u.r);
      gemm_rm(k, 2 * u.r, nn, R.data(), u.v.data(), tmp.data());
      u.v = tmp;
      u.l = k;
      t.v = Q;
      t.r = k;
      ++centre;
    }
    while (centre > q + 1) {
      Tensor& t = A[centre];  // [l][2 r] -> QR of its transpose [2r x l]
      const int l = t.l, w = 2 * t.r;
      cm.resize(0);
      std::vector<cd> Tt((size_t)w * l);
      for (int a = 0; a < l; ++a)
        for (int c = 0; c < w; ++c) Tt[(size_t)c * l + a] = t.v[(size_t)a * w + c];
      if (qr_economic(w, l, Tt.data(), Q, R, work, cm) != 0) {
        g_err = "zgeqrf/zungqr failed";
        return -4;
      }
      const int k = std::min(w, l);
      Tensor& d = A[centre - 1];  // d <- d * R^T : new[i,p,j] = sum_l d[i,p,l] R[j,l]
      std::vector<cd> Rt((size_t)l * k);
      for (int jj = 0; jj < k; ++jj)
        for (int a = 0; a < l; ++a) Rt[(size_t)a * k + jj] = R[(size_t)jj * l + a];
      tmp.resize((size_t)d.l * 2 * k);
      gemm_rm(d.l * 2, k, l, d.v.data(), Rt.data(), tmp.data());
      d.v = tmp;
      d.r = k;
      t.v.resize((size_t)k * w);  // t <- Q^T : [k][2 r]
      for (int jj = 0; jj < k; ++jj)
        for (int c = 0; c < w; ++c) t.v[(size_t)jj * w + c] = Q[(size_t)c * k + jj];
      t.l = k;
      --centre;
    }
    Tensor& a0 = A[q];
    Tensor& a1 = A[q + 1];
    const int l = a0.l, mid = a0.r, r = a1.r;
    const int m = 2 * l, nn = 2 * r;
    theta.resize((size_t)m * nn);  // theta[(a,p)][(p',c)]
    gemm_rm(m, nn, mid, a0.v.data(), a1.v.data(), theta.data());
    auto TH = [&](int a, int p, int pp, int c) -> cd& { return theta[((size_t)(a * 2 + p)) * nn + (size_t)pp * r + c]; };
    if (o == OP_SWAP) {
      for (int a = 0; a < l; ++a)
        for (int c = 0; c < r; ++c) std::swap(TH(a, 0, 1, c), TH(a, 1, 0, c));
    } else {  // XXPhase: cos(th) 1 - i sin(th) X(x)X
      const double th = 0.5 * M_PI * alpha[i];
      const double cs = std::cos(th), sn = std::sin(th);
      const cd mis(0.0, -sn);
      for (int a = 0; a < l; ++a)
        for (int c = 0; c < r; ++c) {
          const cd t00 = TH(a, 0, 0, c), t01 = TH(a, 0, 1, c), t10 = TH(a, 1, 0, c), t11 = TH(a, 1, 1, c);
          TH(a, 0, 0, c) = cs * t00 + mis * t11;
          TH(a, 0, 1, c) = cs * t01 + mis * t10;
          TH(a, 1, 0, c) = cs * t10 + mis * t01;
          TH(a, 1, 1, c) = cs * t11 + mis * t00;
        }
    }
    // ---- SVD (gesdd, jobz = 'S') of the row-major theta[m x nn]: hand LAPACK the column-major view theta^T [nn x m];
    // theta^T = U' S V'^H  =>  theta = conj(V') S U'^T, i.e. U = conj(V'), Vh = U'^T.  Column-major U' [nn x k] read as
    // row-major is U'^T = Vh [k x nn]; column-major V'^H [k x m] read as row-major is (V'^H)^T = conj(V') = U ... [m x k].
    const int k = std::min(m, nn);
    S.resize(k);
    U.resize((size_t)nn * k);   // LAPACK "U" of theta^T, column-major [nn x k]  == row-major Vh [k x nn]
    VT.resize((size_t)k * m);   // LAPACK "VT" of theta^T, column-major [k x m]  == row-major [m x k] = U of theta
    int info = 0, lwork = -1;
    cd wq;
    const int mnmin = k, mnmax = std::max(m, nn);
    rwork.resize((size_t)std::max(5 * mnmin * mnmin + 5 * mnmin, 2 * mnmax * mnmin + 2 * mnmin * mnmin + mnmin) + 16);
    iwork.resize((size_t)8 * mnmin);
    p_zgesdd("S", &nn, &m, theta.data(), &nn, S.data(), U.data(), &nn, VT.data(), &k, &wq, &lwork, rwork.data(), iwork.data(), &info);
    lwork = std::max(1, (int)wq.real());
    if ((int)work.size() < lwork) work.resize(lwork);
    p_zgesdd("S", &nn, &m, theta.data(), &nn, S.data(), U.data(), &nn, VT.data(), &k, work.data(), &lwork, rwork.data(), iwork.data(), &info);
    if (info != 0) {
      g_err = "zgesdd failed";
      return -5;
    }
    double frac = 1.0;
    int keep = kept(S, trunc_budget, value_of_zero, &frac);
    if (max_bond > 0 && keep > max_bond) {  // the chi cap
      keep = max_bond;
      double tot = 0, w = 0;
      for (double x : S) tot += x * x;
      for (int j = 0; j < keep; ++j) w += S[j] * S[j];
      frac = w / tot;
    }
    fidelity *= frac;
    double nrm = 0;
    for (int j = 0; j < keep; ++j) nrm += S[j] * S[j];
    nrm = std::sqrt(nrm);
    ++g2;
    const int nxt = g2 < two_q_pos.size() ? two_q_pos[g2] : q;
    const bool centre_right = (nxt >= q + 1) || (nxt == q);
    // U of theta: element [row][j] = VT_colmajor[j + row * k] (VT is [k x m] column-major: (j, row) at j + row * k)
    a0.v.resize((size_t)m * keep);
    for (int row = 0; row < m; ++row)
      for (int j = 0; j < keep; ++j) a0.v[(size_t)row * keep + j] = VT[(size_t)row * k + j] * (centre_right ? 1.0 : S[j] / nrm);
    a0.r = keep;
    // Vh of theta: element [j][col] = U_colmajor[col + j * nn]
    a1.v.resize((size_t)keep * nn);
    for (int j = 0; j < keep; ++j)
      for (int col = 0; col < nn; ++col) a1.v[(size_t)j * nn + col] = U[(size_t)j * nn + col] * (centre_right ? S[j] / nrm : 1.0);
    a1.l = keep;
    centre = centre_right ? q + 1 : q;
  }
  // ---- hand the tensors over as one block
  int64_t total = 0;
  dims_out[0] = 1;
  for (int k2 = 0; k2 < n; ++k2) {
    dims_out[k2 + 1] = A[k2].r;
    total += (int64_t)A[k2].v.size();
  }
  cd* block = (cd*)std::malloc((size_t)total * sizeof(cd));
  if (!block) {
    g_err = "out of memory";
    return -6;
  }
  int64_t pos = 0;
  for (int k2 = 0; k2 < n; ++k2) {
    std::memcpy(block + pos, A[k2].v.data(), A[k2].v.size() * sizeof(cd));
    pos += (int64_t)A[k2].v.size();
  }
  *tensors_out = reinterpret_cast<double*>(block);
  *n_complex_out = total;
  *fidelity_out = fidelity;
  return 0;
}

void qkb_free(double* p) { std::free(p); }

}  // extern "C"
