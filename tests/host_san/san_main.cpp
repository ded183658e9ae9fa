// Host-only sanitizer driver (tests/test_host_sanitizers.py builds it with -fsanitize=address,undefined and runs it in the CPU
// tier): the planner (qk_planner.cpp) on ragged random bond tables in every mode, and -- when the test hands over an OpenBLAS and
// a bound gate program -- the native host MPS builder (qk_builder.cpp).  Test infrastructure: nothing of the product links it.
#include "../../qml-cutensornet_amd/csrc/qk_plan.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <vector>

int qk_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  return code;
}
extern "C" int qk_plan_destroy(qk_plan* p) {  // (the product's version also frees the plan's device copy)
  delete p;
  return QK_OK;
}

extern "C" {
const char* qkb_last_error(void);
int qkb_init(const char* openblas_path);
int qkb_simulate_chi(int32_t n_qubits, int32_t n_ops, const int8_t* op, const int32_t* q0, const double* alpha, double trunc_budget, double value_of_zero, int32_t max_bond,
                     int32_t* dims_out, double** tensors_out, int64_t* n_complex_out, double* fidelity_out);
void qkb_free(double* p);
}

static std::vector<int32_t> table(std::mt19937& g, int n_states, int n_sites, int max_bond) {
  std::vector<int32_t> d((size_t)n_states * (n_sites + 1), 1);
  std::lognormal_distribution<double> ln(3.2, 0.9);
  for (int s = 0; s < n_states; ++s)
    for (int k = 1; k < n_sites; ++k) d[(size_t)s * (n_sites + 1) + k] = std::max(1, std::min(max_bond, (int)ln(g)));
  return d;
}

static int check_cover(const std::vector<qk_plan*>& plans, int nx, int ny, bool sym, bool quads) {
  std::set<std::pair<int, int>> seen;
  for (const qk_plan* p : plans) {
    const int64_t np = qk_plan_num_pairs(p);
    const int32_t* pr = qk_plan_pairs(p);
    int64_t qs[17];
    const int nq = qk_plan_queues(p, qs);
    if (qs[0] != 0 || qs[16] != np || nq < 1) return 1;
    for (int64_t t = 0; t < np; ++t) {
      int i = pr[2 * t], j = pr[2 * t + 1];
      if (i < 0 || i >= nx || j < 0 || j >= ny) return 2;
      if (sym && i > j) std::swap(i, j);
      if (!seen.insert({i, j}).second && !quads) return 3;  // (quad plans repeat a state at odd ends and mirror a pair on diagonal blocks)
    }
  }
  const size_t want = sym ? (size_t)nx * (nx + 1) / 2 : (size_t)nx * ny;
  return seen.size() == want ? 0 : 4;
}

int main(int argc, char** argv) {
  std::mt19937 g(7);
  int runs = 0;
  for (const int world : {1, 3, 8})
    for (const bool sym : {true, false})
      for (const int mode : {0, 1, 2}) {  // tiled queues, flat list (block), quads
        const int nx = 37 + 5 * world, ny = sym ? nx : 23, n_sites = 11 + world;
        const std::vector<int32_t> xd = table(g, nx, n_sites, 250), yd = table(g, ny, n_sites, 90);
        const uint32_t flags = (sym ? QK_PLAN_SYMMETRIC | (mode == 2 ? 0u : QK_PLAN_ORIENT) : 0u) | (mode == 2 ? QK_PLAN_QUADS : 0u);
        std::vector<qk_plan*> plans((size_t)world, nullptr);
        for (int r = 0; r < world; ++r)
          if (qk_plan_create(n_sites, nx, xd.data(), ny, sym ? nullptr : yd.data(), flags, world, r, mode == 1 ? 16 : 0, &plans[(size_t)r]) != QK_OK) return 10;
        const int rc = check_cover(plans, nx, ny, sym, mode == 2);
        if (rc) {
          fprintf(stderr, "cover check %d failed: world %d sym %d mode %d\n", rc, world, (int)sym, mode);
          return 11;
        }
        if (mode == 0) {  // one cost pass for all ranks == rank by rank
          std::vector<qk_plan*> all((size_t)world, nullptr);
          if (qk_plan_create_all(n_sites, nx, xd.data(), ny, sym ? nullptr : yd.data(), flags, world, all.data()) != QK_OK) return 12;
          for (int r = 0; r < world; ++r) {
            if (all[(size_t)r]->pairs != plans[(size_t)r]->pairs || all[(size_t)r]->stats.flops != plans[(size_t)r]->stats.flops) return 13;
            qk_plan_destroy(all[(size_t)r]);
          }
        }
        for (qk_plan* p : plans) qk_plan_destroy(p);
        ++runs;
      }
  // argument errors must come back as codes, not as crashes
  qk_plan* bad = nullptr;
  const std::vector<int32_t> xd = table(g, 4, 5, 30);
  if (qk_plan_create(5, 4, xd.data(), 0, nullptr, 0u, 1, 0, 0, &bad) == QK_OK) return 20;
  if (qk_plan_create(5, 4, xd.data(), 4, nullptr, QK_PLAN_SYMMETRIC, 2, 2, 0, &bad) == QK_OK) return 21;
  printf("planner: %d plans ok\n", runs);
  if (argc >= 3) {  // the native host builder on a bound gate program dumped by the test: int32 n, int32 n_ops, then op[int8], q0[int32], alpha[f64]
    if (qkb_init(argv[1]) != 0) {
      fprintf(stderr, "qkb_init: %s\n", qkb_last_error());
      return 30;
    }
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 31;
    int32_t hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 32;
    const int n = hdr[0], n_ops = hdr[1];
    std::vector<int8_t> op((size_t)n_ops);
    std::vector<int32_t> q0((size_t)n_ops);
    std::vector<double> alpha((size_t)n_ops);
    if (fread(op.data(), 1, op.size(), f) != op.size() || fread(q0.data(), 4, q0.size(), f) != q0.size() || fread(alpha.data(), 8, alpha.size(), f) != alpha.size()) return 33;
    fclose(f);
    for (const int cap : {0, 4}) {
      std::vector<int32_t> dims((size_t)n + 1);
      double* block = nullptr;
      int64_t nc = 0;
      double fid = 0;
      if (qkb_simulate_chi(n, n_ops, op.data(), q0.data(), alpha.data(), 1e-16, 1e-16, cap, dims.data(), &block, &nc, &fid) != 0) {
        fprintf(stderr, "qkb_simulate_chi: %s\n", qkb_last_error());
        return 34;
      }
      int64_t want = 0;
      for (int k = 0; k < n; ++k) want += (int64_t)dims[(size_t)k] * 2 * dims[(size_t)k + 1];
      if (want != nc || dims[0] != 1 || dims[(size_t)n] != 1 || !(fid > 0.0 && fid <= 1.0 + 1e-12)) return 35;
      printf("builder: %d qubits, %d gates, cap %d: max bond %d, fidelity %.17g\n", n, n_ops, cap, *std::max_element(dims.begin(), dims.end()), fid);
      qkb_free(block);
    }
  }
  return 0;
}
