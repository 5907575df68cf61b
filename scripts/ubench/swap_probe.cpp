// Which rows / halves do v_permlane16_swap / v_permlane32_swap exchange?  Prints, per 16-lane row, the row each
// of the two results came from.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
  const unsigned lane = threadIdx.x, a = 100 + lane, b = 200 + lane; // vdst = a, src0 = b
  uint2v r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  uint2v r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[lane] = r16[0]; o[64 + lane] = r16[1]; o[128 + lane] = r32[0]; o[192 + lane] = r32[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *nm[] = {"swap16 [0] (vdst)", "swap16 [1] (src0)", "swap32 [0] (vdst)", "swap32 [1] (src0)"};
  for (int q = 0; q < 4; ++q) {
    printf("%s:", nm[q]);
    for (int row = 0; row < 4; ++row) printf("  row%d<-%s.row%u", row, h[64 * q + 16 * row] >= 200 ? "src0" : "vdst", (h[64 * q + 16 * row] % 100) / 16);
    printf("\n");
  }
  return 0;
}
