// lab/tools/mfma_rate.hip -- how fast are the two fp64 matrix instructions of gfx950, the fp64 vector FMA, and the two together?
//   hipcc --offload-arch=gfx950 -O3 -o lab/mfma_rate lab/tools/mfma_rate.hip && lab/mfma_rate
// One wavefront per SIMD (256 blocks x 256 threads) or two (512 threads); every wavefront issues N instructions of the kind with
// four independent accumulators; rate = instructions per SIMD and second, cycles per instruction at the 2.4 GHz peak clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void rate_kernel(const int n, double* out, const double seed) {
  const double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  double f0 = a, f1 = b, f2 = a + b, f3 = a - b, f4 = a, f5 = b, f6 = a, f7 = b;
  for (int i = 0; i < n; ++i) {
    if (KIND == 0 || KIND == 3) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
    }
    if (KIND == 1 || KIND == 4) {
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, d3, 0, 0, 0);
    }
    if (KIND == 2 || KIND == 3 || KIND == 4) {  // 8 independent vector FMAs
      f0 = __builtin_fma(f0, a, b), f1 = __builtin_fma(f1, a, b), f2 = __builtin_fma(f2, a, b), f3 = __builtin_fma(f3, a, b);
      f4 = __builtin_fma(f4, a, b), f5 = __builtin_fma(f5, a, b), f6 = __builtin_fma(f6, a, b), f7 = __builtin_fma(f7, a, b);
    }
    if (KIND == 5) {  // 16 x 16 x 4 and 4 x 4 x 4 together
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, d1, 0, 0, 0);
    }
  }
  const v4d c = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int KIND>
static void run(const char* name, const int threads, const double per_iter_mfma, const double per_iter_fma, const double flop_per_mfma) {
  const int blocks = 256, n = 200000;
  double* out;
  if (hipMalloc(&out, sizeof(double) * blocks * threads) != hipSuccess) exit(1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  rate_kernel<KIND><<<blocks, threads>>>(1000, out, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate_kernel<KIND><<<blocks, threads>>>(n, out, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = threads / 256.0, s = ms * 1e-3;
  const double mf = per_iter_mfma * n * waves_per_simd / s, fm = per_iter_fma * n * waves_per_simd / s;  // per SIMD and second
  printf("%-44s %d wave(s)/SIMD  %8.2f ms", name, threads / 256, ms);
  if (per_iter_mfma > 0) printf("  matrix: %6.1f cycles per instruction at 2.4 GHz, %6.2f TFLOP/s on 1024 SIMDs", 2.4e9 / mf, mf * flop_per_mfma * 1024 / 1e12);
  if (per_iter_fma > 0) printf("  vector FMA: %5.2f cycles per instruction, %6.2f TFLOP/s", 2.4e9 / fm, fm * 128 * 1024 / 1e12);
  printf("\n");
  hipFree(out);
}


// one matrix instruction followed by TWO instructions of a class (inline assembly: the compiler neither drops nor moves them): which
// classes take time from the matrix pipe?
template <int CLS>
__global__ void mix_kernel(const int n, double* out, const double seed, double* gbuf) {
  extern __shared__ double lds[];
  const double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double x0 = a, x1 = b;
  float y0 = (float)a, y1 = (float)b;
  int z0 = threadIdx.x, z1 = 1;
  __attribute__((address_space(3))) double* lp = (__attribute__((address_space(3))) double*)lds + threadIdx.x;
  const double* gp = gbuf + threadIdx.x * 2;
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d g0 = {0, 0}, g1 = {0, 0};
#define TWO(A0, A1)           \
  asm volatile(A0 : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1), "+v"(z0), "+v"(z1), "+v"(g0), "+v"(g1) : "v"(a), "v"(b), "v"(lp), "v"(gp) : "memory"); \
  asm volatile(A1 : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1), "+v"(z0), "+v"(z1), "+v"(g0), "+v"(g1) : "v"(a), "v"(b), "v"(lp), "v"(gp) : "memory")
  for (int i = 0; i < n; ++i) {
#define ONE(cx, ax, bx)                                        \
  cx = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bx, cx, 0, 0, 0); \
  if (CLS == 1) { TWO("v_add_f64 %0, %0, %8", "v_add_f64 %1, %1, %9"); }          \
  if (CLS == 2) { TWO("v_mul_f64 %0, %0, %8", "v_mul_f64 %1, %1, %9"); }          \
  if (CLS == 3) { TWO("v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %1, %1, %9, %8"); }  \
  if (CLS == 4) { TWO("v_add_f32 %2, %2, %2", "v_add_f32 %3, %3, %3"); }          \
  if (CLS == 5) { TWO("v_fma_f32 %2, %2, %2, %3", "v_fma_f32 %3, %3, %3, %2"); }  \
  if (CLS == 6) { TWO("v_add_u32 %4, %4, %5", "v_add_u32 %5, %5, %4"); }          \
  if (CLS == 7) { TWO("v_mov_b32 %4, %5", "v_mov_b32 %5, %4"); }                  \
  if (CLS == 8) { TWO("ds_add_f64 %10, %8", "ds_add_f64 %10, %9 offset:8192"); }  \
  if (CLS == 9) { TWO("global_load_dwordx4 %6, %11, off", "global_load_dwordx4 %7, %11, off offset:2048"); } \
  if (CLS == 10) { TWO("v_xor_b32 %4, %4, %5", "v_lshlrev_b32 %5, 1, %5"); }      \
  if (CLS == 11) { TWO("v_mov_b64 %0, %1", "v_mov_b64 %1, %0"); }
    ONE(c0, a, b)
    ONE(c1, b, a)
    ONE(c2, a, a)
    ONE(c3, b, b)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const v4d c = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + x0 + x1 + y0 + y1 + z0 + z1 + g0.x + g1.y;
}
template <int CLS>
static void mix(const char* name, const int threads) {
  const int blocks = 256, n = 100000;
  double *out, *gbuf;
  if (hipMalloc(&out, sizeof(double) * blocks * threads) != hipSuccess || hipMalloc(&gbuf, 1 << 20) != hipSuccess) exit(1);
  hipMemset(gbuf, 0, 1 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  mix_kernel<CLS><<<blocks, threads, 32768>>>(1000, out, 1.0, gbuf);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mix_kernel<CLS><<<blocks, threads, 32768>>>(n, out, 1.0, gbuf);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = 4.0 * n * (threads / 256.0) / (ms * 1e-3);
  printf("16x16x4 + 2 x %-22s %d wave(s)/SIMD  %8.2f ms  %6.1f cycles per matrix instruction (%+5.1f against the bare one: %4.1f per added instruction)\n", name, threads / 256, ms, 2.4e9 / per_simd,
         2.4e9 / per_simd - 64.6, (2.4e9 / per_simd - 64.6) / 2);
  hipFree(out), hipFree(gbuf);
}

int main() {
  for (const int threads : {256, 512}) {
    run<0>("v_mfma_f64_16x16x4_f64", threads, 4, 0, 2048);
    run<1>("v_mfma_f64_4x4x4_4b_f64", threads, 4, 0, 512);
    run<2>("v_fma_f64", threads, 0, 8, 0);
    run<3>("v_mfma_f64_16x16x4_f64 + 2 v_fma_f64 each", threads, 4, 8, 2048);
    run<4>("v_mfma_f64_4x4x4_4b_f64 + 2 v_fma_f64 each", threads, 4, 8, 512);
    run<5>("16x16x4 and 4x4x4 alternating", threads, 4, 0, 1280);
  }
  for (const int threads : {256, 512}) {
    mix<0>("nothing", threads);
    mix<1>("v_add_f64", threads);
    mix<2>("v_mul_f64", threads);
    mix<3>("v_fma_f64", threads);
    mix<11>("v_mov_b64", threads);
    mix<4>("v_add_f32", threads);
    mix<5>("v_fma_f32", threads);
    mix<6>("v_add_u32", threads);
    mix<7>("v_mov_b32", threads);
    mix<10>("v_xor_b32 / v_lshlrev_b32", threads);
    mix<8>("ds_add_f64", threads);
    mix<9>("global_load_dwordx4", threads);
  }
  return 0;
}
