// qk_device.h -- device-side definitions shared by every sweep kernel (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

static constexpr int TILE = 16;  // M/N granule of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct SweepArgs {
  const double* xdata;
  const int32_t* xdims;   // padded bonds
  const int32_t* xtrue;   // true bonds
  const int64_t* xoffs;
  const double* ydata;
  const int32_t* ydims;
  const int32_t* ytrue;
  const int64_t* yoffs;
  int n_sites;
  const int32_t* pairs;
  long long npairs;
  const int32_t* groups;  // (first pair, count) per group
  long long ngroups;
  double* values;
  double* z;
  double* scratch;
  long long x_plane;  // doubles per X plane
  long long t_plane;  // doubles per T plane
  unsigned long long* counter;
  unsigned long long* prof;  // diagnostic build only: cycle sums per section (see QK_VARIANT=9)
  int debug_flags;           // timing experiments only (QK_DEBUG_FLAGS): bit 0 = skip epilogue stores, bit 1 = skip steady-state fetch/stash, bit 2 = skip MFMAs, bit 3 = skip steady-state barriers (all give WRONG results)
  int prio_mode;             // 0: none; 1: second half of the grid at s_setprio 1; 2: odd blocks at s_setprio 1
};

// Workgroup barrier that publishes LDS writes only: it does NOT drain outstanding global loads or LDS-DMAs
// (a __syncthreads() would wait vmcnt(0) and cancel the prefetch that is meant to stay in flight).
__device__ __forceinline__ void qk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
