// What hides in the shadow of a 64-cycle v_mfma_f64_16x16x4 of the SAME wave (one wave per SIMD)?
//  (a) a latency-bound VALU chain (the DPP-broadcast LDL^T of gar_wave2.hpp), one chain instruction per MFMA;
//  (b) independent memory instructions (global_load / global_store / ds_write / ds_read).
// Source order is pinned with sched_barrier(0) so that what is measured is what is written.
// hipcc -O3 --offload-arch=gfx950 -o overlap overlap.cpp && ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)
template <int N> __device__ __forceinline__ double rb(double v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + N, 0xF, 0xF, true); }
__device__ __forceinline__ double row_bcast(double v, int n) {
  switch (n) {
  case 0: return rb<0>(v); case 1: return rb<1>(v); case 2: return rb<2>(v); case 3: return rb<3>(v);
  case 4: return rb<4>(v); case 5: return rb<5>(v); case 6: return rb<6>(v); case 7: return rb<7>(v);
  case 8: return rb<8>(v); case 9: return rb<9>(v); case 10: return rb<10>(v); default: return rb<11>(v);
  }
}
struct Mf {
  double fa[9], fb[9];
  double4_t acc[3];
  int n = 0;
  __device__ __forceinline__ void one() { // the next MFMA of the stream
    const int s = n % 9, m = n % 3;
    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[(s + m) % 9], acc[m], 0, 0, 0);
    ++n;
  }
};
// MODE 0: 96 MFMAs; 1: LDL chain alone; 2: MFMAs then chain; 3: chain with one MFMA after each critical instruction
// MODE 4: 96 MFMAs then 48 loads + 48 ds_write + 24 stores; 5: the same memory instructions one per MFMA shadow
template <int MODE> __global__ void __launch_bounds__(64, 1) k(double *out, long long *cyc, const double *in, double *scratch) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  double a[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) a[j] = in[(lane % 12) * 12 + j] + (j == (lane % 12) ? 20.0 : 0.0);
  Mf M;
#pragma unroll
  for (int s = 0; s < 9; ++s) { M.fa[s] = in[768 + lane + 64 * s]; M.fb[s] = in[2048 + lane + 64 * s]; }
  M.acc[0] = M.acc[1] = M.acc[2] = double4_t{0, 0, 0, 0};
  const double *gsrc = in + lane;
  double *gdst = scratch + (size_t)blockIdx.x * 64 * 64 + lane;
  double ld[48];
  double sum = 0.0;
  long long t0 = clock64();
  for (int it = 0; it < 32; ++it) {
    M.n = 0;
    if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
      for (int g = 0; g < 96; ++g) M.one();
    }
    if (MODE == 1 || MODE == 2 || MODE == 3) {
      SB;
#pragma unroll
      for (int kcol = 0; kcol < 12; ++kcol) {
        const double akk = row_bcast(a[kcol], kcol);
        if (MODE == 3) { SB; M.one(); SB; }
        double r = __builtin_amdgcn_rcp(akk);
        if (MODE == 3) { SB; M.one(); SB; }
        double e = __builtin_fma(-akk, r, 1.0);
        if (MODE == 3) { SB; M.one(); SB; }
        r = __builtin_fma(r, e, r);
        if (MODE == 3) { SB; M.one(); SB; }
        e = __builtin_fma(-akk, r, 1.0);
        if (MODE == 3) { SB; M.one(); SB; }
        r = __builtin_fma(r, e, r);
        if (MODE == 3) { SB; M.one(); SB; }
        const double nl = a[kcol] * -r;
        if (MODE == 3) { SB; M.one(); SB; }
#pragma unroll
        for (int j = kcol + 1; j < 12; ++j)
          a[j] = __builtin_fma(row_bcast(nl, j), a[kcol], a[j]);
        a[kcol] = nl;
        if (MODE == 3) { SB; M.one(); SB; }
      }
      SB;
    }
    if (MODE == 4) {
      SB;
#pragma unroll
      for (int g = 0; g < 48; ++g) ld[g] = gsrc[64 * (g + it)];
#pragma unroll
      for (int g = 0; g < 48; ++g) lds[64 * g + lane] = M.fa[g % 9];
#pragma unroll
      for (int g = 0; g < 24; ++g) gdst[64 * g] = M.fb[g % 9];
      SB;
#pragma unroll
      for (int g = 0; g < 48; ++g) sum += ld[g];
    }
    if (MODE == 5) {
#pragma unroll
      for (int g = 0; g < 96; ++g) {
        M.one();
        SB;
        if (g < 48) ld[g] = gsrc[64 * (g + it)];
        else lds[64 * (g - 48) + lane] = M.fa[g % 9];
        if (g % 4 == 0) gdst[64 * (g / 4)] = M.fb[(g / 4) % 9];
        SB;
      }
#pragma unroll
      for (int g = 0; g < 48; ++g) sum += ld[g];
    }
  }
  long long t1 = clock64();
  double s = M.acc[0][0] + M.acc[1][1] + M.acc[2][2] + sum + lds[lane];
#pragma unroll
  for (int j = 0; j < 12; ++j) s += a[j];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *out, *in, *scratch; long long *cyc;
  CHECK(hipMalloc(&out, 8 * 64 * 1024)); CHECK(hipMalloc(&in, 8 * 65536)); CHECK(hipMalloc(&cyc, 8 * 1024));
  CHECK(hipMalloc(&scratch, 8ull * 1024 * 64 * 64));
  static double h[65536]; for (int i = 0; i < 65536; ++i) h[i] = 0.01 * ((i * 7919) % 97) + 0.5;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  static long long c[1024];
  const char *names[] = {"96 MFMA", "DPP LDL^T chain (12 columns) alone", "96 MFMA, then the chain", "chain, one MFMA after each critical instruction (96)",
                         "96 MFMA, then 48 global_load + 48 ds_write + 24 global_store", "the same memory instructions, one per MFMA shadow"};
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      const size_t sh = 64 * 48 * 8 + 512;
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), sh, 0, out, cyc, in, scratch);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), sh, 0, out, cyc, in, scratch);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), sh, 0, out, cyc, in, scratch);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), sh, 0, out, cyc, in, scratch);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1024), dim3(64), sh, 0, out, cyc, in, scratch);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(1024), dim3(64), sh, 0, out, cyc, in, scratch);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 1024; ++i) avg += c[i]; avg /= 1024;
    printf("%-70s %.0f cycles / iteration\n", names[mode], avg / 32);
  }
  return 0;
}
