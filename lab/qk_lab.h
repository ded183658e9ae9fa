/* qk_lab.h -- entry points that exist ONLY in the lab library libqklab.so (qkgram.hip + qk_build.hip + qk_lab.hip built
 * with -DQK_LAB; tools/ load it for A/B measurements).  The shipped libqkgram.so has none of this: no QK_VARIANT, no
 * QK_DEBUG_FLAGS / QK_PRIO (switches of the experimental kernels that give wrong results by construction). */
#ifndef QK_LAB_H
#define QK_LAB_H
#include "../../include/qkgram.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Cycle sums of an instrumented sweep kernel's sections (QK_VARIANT=9 / 19, or a -DQKF_PROF build of the fused sweep). */
int qk_debug_profile(qk_ctx* ctx, unsigned long long* out8);
/* TFLOP/s of the MFMA block with the ring sweep's per-step ingredients added back one at a time.
 * which = 8*(8-wave workgroup) + {0 bare, 1 +barrier, 2 +stash, 3 +global fetch, 4 +deep fetch}. */
int qk_debug_mma_bench(qk_ctx* ctx, int which, int wgs_per_cu, int reps, double* tflops);
#ifdef __cplusplus
}
#endif
#endif
