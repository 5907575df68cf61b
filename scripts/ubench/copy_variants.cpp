// What does this pool's HBM give a plain device-to-device copy?  (MI355X_MICROARCH.md quotes ~6.3 TB/s achievable.)
// Variants of a 16 B/lane copy of 28.5 GB -> 28.5 GB (the byte count of one backward launch of the bench):
// grid-stride with 1 / 4 / 8 independent loads in flight per lane, with and without nontemporal hints, at several
// grid sizes, and hipMemcpyAsync D2D for reference.  Build: hipcc -O3 --offload-arch=gfx950 copy_variants.cpp -o copy_variants
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT> __global__ void __launch_bounds__(256) copyk(const d2 *__restrict__ s, d2 *__restrict__ d, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = NT ? __builtin_nontemporal_load(&s[i + u * stride]) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], &d[i + u * stride]);
      else d[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride)
    d[i] = s[i];
}
template <int U, bool NT> double run(const d2 *s, d2 *d, long long n, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((copyk<U, NT>), dim3(blocks), dim3(256), 0, 0, s, d, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}
int main() {
  const long long bytes = 28490000000ll / 2 * 2; // per direction
  const long long n = bytes / 16;
  d2 *s, *d;
  if (hipMalloc(&s, n * 16) != hipSuccess || hipMalloc(&d, n * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(s, 1, n * 16);
  hipDeviceSynchronize();
  const double tot = 2.0 * n * 16;
  for (int blocks : {1024, 2048, 4096, 16384, 65536}) {
    double a = run<1, false>(s, d, n, blocks), b = run<4, false>(s, d, n, blocks), c = run<8, false>(s, d, n, blocks),
           e = run<4, true>(s, d, n, blocks);
    printf("blocks %6d: U=1 %.2f ms %.2f TB/s | U=4 %.2f ms %.2f TB/s | U=8 %.2f ms %.2f TB/s | U=4 nontemporal %.2f ms %.2f TB/s\n", blocks,
           a, tot / a / 1e9, b, tot / b / 1e9, c, tot / c / 1e9, e, tot / e / 1e9);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0, 0); hipMemcpyAsync(d, s, n * 16, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  printf("hipMemcpyAsync D2D: %.2f ms %.2f TB/s\n", best, tot / best / 1e9);
  return 0;
}
