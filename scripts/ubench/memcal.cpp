// memcal.cpp -- known-size streaming copies used to CALIBRATE the rocprofv3 FETCH_SIZE /
// WRITE_SIZE counters on gfx950 for the access widths the gar kernels use
// (MI355X_MICROARCH.md, section HBM: FETCH_SIZE reports 1/2 of the bytes of a 16 B/lane
// streaming read; other widths and WRITE_SIZE are uncalibrated).  Each kernel moves exactly
// `bytes` from src to dst (1 GiB by default, far beyond the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));
__global__ void memcal_copy_b64(const double *src, double *dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
__global__ void memcal_copy_b128(const double2_t *src, double2_t *dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
// 8 B per lane in 32 B segments (the stride pattern of the MFMA-operand loads: 4 lanes contiguous,
// groups 288 B apart)
__global__ void memcal_read_seg32(const double *src, double *dst, size_t n) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double acc = 0;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t base = wave * 576; base + 576 <= n; base += nw * 576) // 16 columns x 36 rows
    for (int s = 0; s < 9; ++s) acc += src[base + li * 36 + 4 * s + lk];
  if (acc == 1.2345) dst[0] = acc;
}
int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : (size_t)1 << 30);
  double *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
  const size_t n = bytes / 8;
  hipLaunchKernelGGL(memcal_copy_b64, dim3(4096), dim3(256), 0, 0, a, b, n);
  hipLaunchKernelGGL(memcal_copy_b128, dim3(4096), dim3(256), 0, 0, (const double2_t *)a, (double2_t *)b, n / 2);
  hipLaunchKernelGGL(memcal_read_seg32, dim3(4096), dim3(256), 0, 0, a, b, n - (n % 576));
  CHECK(hipDeviceSynchronize());
  printf("memcal: each kernel moved %zu bytes (read) / %zu (written; seg32: 0)\n", bytes, bytes);
  return 0;
}
