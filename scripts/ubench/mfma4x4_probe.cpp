// v_mfma_f64_4x4x4_4b_f64: issue cost and lane layout (A=I style probes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void timing(double *out, long long *cyc) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  long long t0 = clock64();
  for (int i = 0; i < 4096; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc3, 0, 0, 0);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}
// layout probe: lane l supplies a = A-value, b = B-value; D[l] returned
__global__ void probe(const double *A, const double *B, double *D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
int main() {
  double *out; long long *cyc; CHECK(hipMalloc(&out, 8 * 64 * 256)); CHECK(hipMalloc(&cyc, 8 * 256));
  hipLaunchKernelGGL(timing, dim3(256), dim3(64), 0, 0, out, cyc); CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(timing, dim3(256), dim3(64), 0, 0, out, cyc); CHECK(hipDeviceSynchronize());
  long long c[256]; CHECK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
  printf("v_mfma_f64_4x4x4_4b: %.1f cycles per instruction (4 independent accumulators)\n", c[0] / (4096.0 * 4));
  // layout: for each source lane la (A one-hot = 1 at la) and lb (B one-hot), find D lanes that become 1
  double *dA, *dB, *dD; CHECK(hipMalloc(&dA, 512)); CHECK(hipMalloc(&dB, 512)); CHECK(hipMalloc(&dD, 512));
  double hA[64], hB[64], hD[64];
  // for A lane la: set B all ones -> D lanes with nonzero tell (block,row) of la ; count tells k sharing
  for (int la = 0; la < 64; la += 1) {
    for (int i = 0; i < 64; ++i) { hA[i] = (i == la); hB[i] = 1.0 + i; }
    CHECK(hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD); CHECK(hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost));
    if (la < 20 || la % 16 == 0) {
      printf("A lane %2d -> D:", la);
      for (int i = 0; i < 64; ++i) if (hD[i] != 0.0) printf(" [%d]=B%d", i, (int)hD[i] - 1);
      printf("\n");
    }
  }
  return 0;
}
