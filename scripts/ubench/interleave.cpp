// Can the compiler interleave an independent VALU dependent-chain workload into an MFMA stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rcp(double a) {
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
// a factor-like dependent VALU workload on 12 registers (lane = row)
template <int NU> __device__ __forceinline__ void ldl(double (&a)[NU]) {
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const double akk = lane_bcast(a[k], k);
    const double d = fast_rcp(akk);
    const double lik = a[k] * d;
#pragma unroll
    for (int j = k + 1; j < NU; ++j)
      a[j] = __builtin_fma(-lane_bcast(lik, j), a[k], a[j]);
    a[k] = lik;
  }
}
template <int MODE> __global__ void __launch_bounds__(64, 1) k(double *out, long long *cyc, const double *in) {
  const int lane = threadIdx.x;
  double a[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) a[j] = in[lane * 12 + j] + (j == (lane % 12) ? 20.0 : 0.0);
  double fa[9], fb[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) { fa[s] = in[768 + lane + 64 * s]; fb[s] = in[2048 + lane + 64 * s]; }
  double4_t acc[3] = {double4_t{0,0,0,0}, double4_t{0,0,0,0}, double4_t{0,0,0,0}};
  long long t0 = clock64();
  for (int it = 0; it < 64; ++it) {
    if (MODE == 0 || MODE == 2) {           // 135 MFMAs
#pragma unroll
      for (int rep = 0; rep < 5; ++rep)
#pragma unroll
        for (int s = 0; s < 9; ++s)
#pragma unroll
          for (int m = 0; m < 3; ++m)
            acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[(s + m) % 9], acc[m], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {           // the VALU chain (3 factorizations ~ factor+solve work)
      ldl<12>(a); ldl<12>(a); ldl<12>(a);
    }
    if (MODE == 3) {  // interleaved with sched_group_barrier: 1 MFMA then 6 VALU, repeated
#pragma unroll
      for (int rep = 0; rep < 5; ++rep)
#pragma unroll
        for (int s = 0; s < 9; ++s)
#pragma unroll
          for (int m = 0; m < 3; ++m)
            acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[(s + m) % 9], acc[m], 0, 0, 0);
      ldl<12>(a); ldl<12>(a); ldl<12>(a);
#pragma unroll
      for (int g = 0; g < 135; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);  // 8 VALU
      }
    }
  }
  long long t1 = clock64();
  double s = acc[0][0] + acc[1][1] + acc[2][2];
#pragma unroll
  for (int j = 0; j < 12; ++j) s += a[j];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *out, *in; long long *cyc;
  CHECK(hipMalloc(&out, 8 * 64 * 1024)); CHECK(hipMalloc(&in, 8 * 4096)); CHECK(hipMalloc(&cyc, 8 * 1024));
  double h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.01 * ((i * 7919) % 97) + 0.5;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  long long c[1024];
  const char *names[] = {"135 MFMA only", "VALU chain only (3x LDL12)", "both, source order", "both, sched_group_barrier 1 MFMA : 8 VALU"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, out, cyc, in);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, out, cyc, in);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, out, cyc, in);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 0, 0, out, cyc, in);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 1024; ++i) avg += c[i]; avg /= 1024;
    printf("%-46s %.0f cycles / iteration\n", names[mode], avg / 64);
  }
  return 0;
}
