// qk_plan.h -- the host-side plan of a Gram share (struct qk_plan) and the constants the planner shares with the launches.
// Plain C++: no HIP type appears here, so the planner (qk_planner.cpp) also builds as a host-only translation unit
// (tests/host_san: -fsanitize=address,undefined).
#pragma once
#include "../../include/qkgram.h"

#include <cstdint>
#include <vector>

int qk_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));  // sets qk_last_error(), returns code

#ifndef QKF_XCAP_ONE_V
#define QKF_XCAP_ONE_V 8192
#endif
#ifndef QKF_XCAP_TWO_V
#define QKF_XCAP_TWO_V 4608
#endif
#ifndef QKF_TWO_WGS
#define QKF_TWO_WGS 2  // workgroups per CU of the small-site shape (experiment builds: 3 with a 3072-element buffer)
#endif
static constexpr int QKF_XCAP_ONE = QKF_XCAP_ONE_V, QKF_XCAP_TWO = QKF_XCAP_TWO_V;  // elements of the fused sweep's LDS X buffer with one / two workgroups per CU
static constexpr int QK_TILE = 16;                                        // M/N granule of v_mfma_f64_16x16x4_f64
static inline int qk_pad16(int x) { return (x + QK_TILE - 1) / QK_TILE * QK_TILE; }

struct qk_ctx;

struct qk_plan {
  int n_sites = 0, nx = 0, ny = 0;
  bool symmetric = false;
  bool quad = false;  // pairs come in 2x2 blocks (QK_PLAN_QUADS): [4q..4q+3] = (i1,j1), (i2,j1), (i1,j2), (i2,j2)
  int world = 1, rank = 0;
  int64_t total_pairs = 0, max_per_rank = 0;
  std::vector<int32_t> pairs;   // this rank, (i, j) interleaved
  std::vector<int32_t> groups;  // (first pair, count): runs of <= group pairs that share the x state
  int group = 1;
  qk_stats stats{};
  qk_stats second{};       // pairs / flops / padded_flops / bytes of the class-1 run [n_first, end)
  int64_t n_first = 0;     // pairs [n_first, end) are the class whose sites fit the fused sweep's smaller LDS buffer (== number of pairs: no split)
  int nq = 1;                 // device work queues: 1 = one list; 16 = two classes of pairs x 8 XCD queues (the second class may be empty)
  int64_t qstart[17] = {0};   // queue s = pairs [qstart[s], qstart[s + 1]) of this rank's list; queues 8..15 = the class-1 run
  int edge_k = 0;             // sites at either end of the chain that the fused sweep takes from the sets' edge blocks (0: none)
  bool second_wave2 = false;  // the second run holds the pairs of two states whose bonds are all <= 32: swept by the one-wave kernel (mixed sets)
  double fit_two = 1.0;  // share of this rank's padded work in sites whose X and X' fit the fused sweep's smaller LDS buffer
  double fit_narrow = 1.0;  // ... in sites of at most the narrow size (QK_PLAN_FIT: where the two-workgroup shape still beats the 12-wave dual one)
  double tile_reuse_bytes = 0;  // bytes of this rank's share if every state were read once per plan tile it takes part in (the tile-reuse lower bound)
  double plan_ms = 0;           // host time qk_plan_create spent on this plan
  int plan_threads = 1;         // host threads it used
  // lazily uploaded copy
  qk_ctx* up_ctx = nullptr;
  int32_t* d_pairs = nullptr;
  int32_t* d_groups = nullptr;
};
