// fetch_calib.hip -- known-bytes microbenchmark for the rocprofv3 FETCH_SIZE counter on gfx950.
// MI355X_MICROARCH.md: "FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane)".  The sweep
// kernels do not read like that: the site-fused sweep loads MFMA fragments as 4 rows x 256 contiguous bytes per wave instruction
// (row stride = a whole tensor row), the one-wave sweep brings the same 256-byte row segments by LDS-DMA.  Each kernel below reads
// a buffer far larger than the 256 MiB Infinity Cache EXACTLY ONCE with one of those patterns, so bytes(known) / FETCH_SIZE is
// the correction that pattern needs.   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o pmc -- /tmp/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(e)                                                                     \
  do {                                                                               \
    hipError_t r_ = (e);                                                             \
    if (r_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s failed: %s\n", #e, hipGetErrorString(r_));            \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

// (a) the guide's case: consecutive lanes read consecutive 16-byte words, 1 KiB per wave instruction
__global__ __launch_bounds__(256) void calib_wide_coalesced(const v2d* __restrict__ src, double* __restrict__ sink, const long n16) {
  v2d acc = {0, 0};
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (long)gridDim.x * blockDim.x) acc += src[e];
  if (acc.x == 12345.678) sink[0] = acc.y;
}

// (b) fragment loads of the site-fused sweep: lane (q, j) of k-step s reads the 16 bytes of element [4 s + q][j] of a row-major
// matrix with `ld` complex elements per row: 4 segments of 256 contiguous bytes per wave instruction, `ld * 16` bytes apart.
// Every wave walks its own 16-column strip of the matrix top to bottom, so each byte of the matrix is read exactly once.
__global__ __launch_bounds__(256) void calib_fragment_rows(const v2d* __restrict__ src, double* __restrict__ sink, const int rows, const int ld) {
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const long strip = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);  // 16-column strips
  const long nstrips = ld / 16;
  v2d acc = {0, 0};
  for (long st = strip; st < nstrips; st += (long)gridDim.x * (blockDim.x / 64)) {
    const v2d* base = src + st * 16 + j;
#pragma unroll 4
    for (int s = 0; s < rows / 4; ++s) acc += base[(long)(4 * s + q) * ld];
  }
  if (acc.x == 12345.678) sink[0] = acc.y;
}

// (c) the same 256-byte row segments brought by LDS-DMA (global_load_lds_dwordx4), as the one-wave sweep does
__global__ __launch_bounds__(64) void calib_lds_dma_rows(const v2d* __restrict__ src, double* __restrict__ sink, const int rows, const int ld) {
  __shared__ v2d ring[4 * 64];
  const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
  const long nstrips = ld / 16;
  double acc = 0;
  for (long st = blockIdx.x; st < nstrips; st += gridDim.x) {
    const char* base = reinterpret_cast<const char*>(src + st * 16 + j);
    for (int s0 = 0; s0 < rows / 4; s0 += 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(base + (long)(4 * (s0 + i) + q) * ld * 16, (lds_ptr_t)(ring + i * 64), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += ring[lane].x + ring[64 + lane].y + ring[128 + lane].x + ring[192 + lane].y;
    }
  }
  if (acc == 12345.678) sink[0] = acc;
}

int main() {
  const long bytes = 4l << 30;  // 4 GiB: 16 x the Infinity Cache
  v2d* buf = nullptr;
  double* sink = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 0, bytes));
  const long n16 = bytes / 16;
  for (int rep = 0; rep < 2; ++rep) {
    calib_wide_coalesced<<<dim3(2048), dim3(256)>>>(buf, sink, n16);
    for (int ld : {64, 128, 256, 1024}) {  // complex elements per matrix row: tensors with right bonds of 32 / 64 / 128 / 512 (x 2 for p)
      const int rows = (int)(n16 / ld);
      calib_fragment_rows<<<dim3(1024), dim3(256)>>>(buf, sink, rows / 4 * 4, ld);
      calib_lds_dma_rows<<<dim3(4096), dim3(64)>>>(buf, sink, rows / 16 * 16, ld);
    }
    CHECK(hipDeviceSynchronize());
  }
  std::printf("bytes per launch (known): %ld; launch order per repetition: wide, then for ld = 64, 128, 256, 1024: fragment rows, LDS-DMA rows\n", bytes);
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}
