// qk_host.h -- host-side state shared by the translation units:
//   qkgram.hip    the C ABI, the planner and the shipped sweep kernels      } libqkgram.so
//   qk_build.hip  the device MPS builder                                    }
//   qk_lab.hip    experimental / diagnostic kernels for A/B measurements: only in lab/libqklab.so (-DQK_LAB, lab/tools)
#pragma once
#include "../../include/qkgram.h"
#include "qk_plan.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

struct QkRangeGuard {  // a roctx range (qk_range_push / qk_range_pop) that closes on every exit path
  explicit QkRangeGuard(const char* n) { qk_range_push(n); }
  ~QkRangeGuard() { qk_range_pop(); }
};

#define HIP_TRY(expr)                                                                                 \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) return qk_fail(QK_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));  \
  } while (0)

static constexpr int GMAX = 4;  // pairs per group of the group-sweep lab kernel (sizes its X/T scratch)

struct qk_ctx {
  int device = 0;
  int num_cus = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool split_pending = false;  // the last sweep was two launches: second_ms is still to be read from the events
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;  // ev_mid: between the two launches of a split sweep
  hipEvent_t ev_d = nullptr;  // at the start of qk_gram_values: what runs between it and ev0 are the kernels that make a set's derived images (first Gram of a set)
  bool ev_pending = false;
  double* scratch = nullptr;
  size_t scratch_bytes = 0;
  unsigned long long* counter = nullptr;  // work-queue heads (QK_NQ_MAX of them, QK_QSTRIDE apart) + 2 x 4 tail clocks behind them
  bool tail_pending = false;
  bool merge_sites = true;  // QK_MERGE (0 disables): the site-fused sweep walks the chain in merged steps of two sites (qk_device.h: SweepArgs.merge_steps)
  unsigned long long* prof = nullptr;  // 8 cycle sums of the diagnostic variant
  int variant = 20;    // 20 = the shipped kernels.  Anything else exists only in libqklab.so (QK_VARIANT there: 17 = lean register-staged
                       // sweep; 0, 2, 12, 13, 14, 16, 21, 23 = other kernels kept for A/B; 9, 19 = instrumented)
  int wgs_per_cu = 2;  // resident workgroups per CU (QK_WGS_PER_CU)
  bool wave_path = true;   // fp64 sets whose bonds are all <= 16 use the one-wave-per-pair register sweep (QK_WAVE=0 opts out)
  bool wave2_ring = true;  // ... with its k-step groups prefetched through a per-wave LDS ring (QK_WAVE2=2: plain loads)
  bool wave2_path = true;  // fp64 sets whose bonds are all <= 32 use the one-wave-per-pair sweep with 2 x 2 register tiles (QK_WAVE2=0 opts out)
  bool small_path = true;  // sets whose bonds are all <= 32 use the LDS-resident small-bond sweep (QK_SMALL=0 opts out)
  int fused_split = 1;  // sweep the plan's two runs of pairs with the two shapes of the site-fused kernel: 1 = when the share is long enough for two launches (default), 2 = always, 0 = one shape (QK_FUSED_SPLIT)
  int fused_wgs = 0;       // workgroups per CU of the site-fused sweep: 0 = chosen per launch from the plan, 1 / 2 forced (QK_FUSED_WGS)
  bool deterministic = false;  // QK_DETERMINISTIC=1: only kernels that add in a fixed order (no LDS atomics)
  int fused_path = 1;      // fp64 sets with a bond > 32 use the site-fused sweep (QK_FUSED=0: ring sweep instead; 2: also for bonds 17..32)
  qk_stats last{};
  // the device MPS builder's per-workgroup arena and workspace, kept between calls (qk_build.hip)
  void* build_arena = nullptr;
  size_t build_arena_bytes = 0;
  void* build_work = nullptr;
  size_t build_work_bytes = 0;
  // scratch of the workgroups that make a set's edge blocks (qk_edge_kernel), kept between calls
  void* derive_tmp = nullptr;
  size_t derive_tmp_bytes = 0;
};

uint64_t qk_next_uid();

struct qk_mps_set {
  qk_ctx* ctx = nullptr;
  uint64_t uid = qk_next_uid();  // unique per set of this process: caches keyed on a set's address also compare this (a freed address may come back)
  int n_states = 0, n_sites = 0, max_pad = 0;
  int precision = 64;         // bits of a real: 64 (complex128 planes) or 32 (complex64 planes, same element offsets)
  double* d_data = nullptr;   // the planes; floats when precision == 32
  double* d_il = nullptr;     // fp64 only, made on first use by the site-fused sweep: the same image with re/im interleaved (complex128), same offsets
  int32_t* d_dims = nullptr;  // padded bonds [n_states][n_sites+1]
  int32_t* d_true = nullptr;  // true bonds   [n_states][n_sites+1]
  int64_t* d_offs = nullptr;  // re-plane offsets (doubles) [n_states][n_sites]
  std::vector<int32_t> dims_true;
  int64_t bytes = 0;
  // edge blocks of the site-fused sweep (made on first use for the plan's edge_k; qk_device.h: SweepArgs.edge_k)
  double* d_edge = nullptr;        // interleaved complex: per state the left block [2^k][pad(chi_k)], then the right block [2^k][pad(chi_{n-k})]
  long long* d_edge_offs = nullptr;  // [n_states][2] element offsets
  int edge_k = 0;
  int64_t edge_bytes = 0;
  // merged image of the site-fused sweep (SweepArgs.merge_steps; made on first use for the plan's edge_k): the chain's sites [k, n - k)
  // contracted in twos, interleaved complex [l][4][r]
  double* d_mg = nullptr;
  int64_t* d_mg_offs = nullptr;  // offsets in doubles [n_states][mg_steps]
  long long* d_mg_units = nullptr;  // first 16 x 16 block of each (state, step) in the numbering of qk_merge_kernel's units [n_states * mg_steps + 1]
  std::vector<long long> h_edge_offs, h_mg_units;  // host staging of the tables above (they outlive the asynchronous copies)
  std::vector<int64_t> h_mg_offs;
  int mg_k = -1, mg_steps = 0;
  int64_t mg_bytes = 0;
};

struct SweepArgs;
// qk_lab.hip: raise the LDS limit of the lab kernels; launch lab variant `variant` (returns QK_EINVAL if it is not one)
int qk_lab_init(qk_ctx* c);
int qk_lab_launch(qk_ctx* c, int variant, const SweepArgs& a, int grid, int n_sites);
int qk_lab_launch_quad(qk_ctx* c, const SweepArgs& a, int grid, int n_sites, bool f32);  // QK_PLAN_QUADS plans
