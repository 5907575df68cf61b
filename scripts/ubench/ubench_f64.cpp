// ubench_f64.cpp -- fp64 issue-rate microbenchmarks for gfx950 (MI355X): how fast does
// v_mfma_f64_16x16x4_f64 issue, what is its dependent latency, and does it overlap with
// v_fma_f64 / LDS traffic of the same or a co-resident wave?  Feeds DESIGN.md's kernel budget.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_f64 ubench_f64.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args { double *out; long long *cyc; int iters; int mode; };

__device__ __forceinline__ double4_t mfma(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// role: 0 = idle, 1 = mfma NACC independent accumulators, 2 = v_fma_f64 x8 chains,
// 3 = mfma + interleaved fma in same wave, 4 = ds_read_b64 stream, 5 = readlane-broadcast fma
template <int NACC> __device__ __forceinline__ double run_mfma(int iters, double a, double b) {
  double4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = mfma(a, b, acc[i]);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  return s;
}

__device__ __forceinline__ double run_fma(int iters, double a, double b) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  return s;
}

template <int NF> __device__ __forceinline__ double run_mixed(int iters, double a, double b) {
  double4_t acc[2] = {double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}};
  double x[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      acc[m] = mfma(a, b, acc[m]);
#pragma unroll
      for (int i = 0; i < NF; ++i) x[i] = __builtin_fma(x[i], b, a);
    }
  }
  double s = acc[0][0] + acc[1][1];
#pragma unroll
  for (int i = 0; i < NF; ++i) s += x[i];
  return s;
}

__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double run_bcast(int iters, double a, double b) {
  double x[4] = {a, a + 1, a + 2, a + 3};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i & 3] = __builtin_fma(lane_bcast(x[(i + 1) & 3], i), b, x[i & 3]);
  }
  return x[0] + x[1] + x[2] + x[3];
}

// MFMA with both operands freshly read from LDS each time (the real kernel's pattern)
template <int NACC> __device__ __forceinline__ double run_mfma_lds(int iters, const double *lds, int lane) {
  double4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  const int li = lane & 15, lk = lane >> 4;
  for (int it = 0; it < iters; ++it) {
    const double *p = lds + ((it & 7) * 4 + lk);
    const double bq = p[li * 38];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const double aq = p[(16 * i + li) * 38 + 600];
      acc[i] = mfma(aq, bq, acc[i]);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  return s;
}

__device__ __forceinline__ double run_int(int iters, int lane) {
  unsigned x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = lane + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = x[i] * 1664525u + 1013904223u; // v_mad_u32 / mul_lo+add
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= x[i];
  return (double)s;
}
__device__ __forceinline__ double run_f32(int iters, int lane) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = lane + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 0.999f, 1.0f);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  return (double)s;
}
// alternating phases like the one-wave-per-problem sweep: a burst of NB register-operand MFMAs, then a
// latency-bound serial phase (readlane broadcast + fma chain), optionally at raised priority
template <int PRIO> __device__ __forceinline__ double run_alt(int iters, double a, double b) {
  double4_t acc[3] = {double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}};
  double x[4] = {a, a + 1, a + 2, a + 3};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 30; ++m) acc[m % 3] = mfma(a + x[0], b, acc[m % 3]);
    if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int rep = 0; rep < 6; ++rep) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i & 3] = __builtin_fma(lane_bcast(x[(i + 1) & 3], i), b, x[i & 3]);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  }
  return x[0] + x[1] + x[2] + x[3] + acc[0][0] + acc[1][0] + acc[2][0];
}

__global__ void __launch_bounds__(512) kern(Args A, int roleLo, int roleHi) {
  __shared__ double lds[8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 1e-3 * i;
  __syncthreads();
  const int role = wave < 4 ? roleLo : roleHi;
  const double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
  long long t0 = clock64();
  double r = 0;
  switch (role) {
  case 1: r = run_mfma<1>(A.iters * 4, a, b); break;
  case 2: r = run_mfma<2>(A.iters * 2, a, b); break;
  case 3: r = run_mfma<4>(A.iters, a, b); break;
  case 4: r = run_fma(A.iters, a, b); break;              // 8 fma / iter
  case 5: r = run_mixed<4>(A.iters, a, b); break;         // per iter: 2 mfma + 8 fma
  case 6: r = run_mixed<8>(A.iters, a, b); break;         // per iter: 2 mfma + 16 fma
  case 7: r = run_mixed<12>(A.iters, a, b); break;        // per iter: 2 mfma + 24 fma
  case 8: r = run_bcast(A.iters, a, b); break;            // 16 (2 readlane + fma) / iter
  case 9: r = run_mfma_lds<3>(A.iters, lds, lane); break; // 3 mfma + 4 ds_read_b64 / iter
  case 10: r = run_mfma_lds<1>(A.iters * 3, lds, lane); break;
  case 11: __builtin_amdgcn_s_setprio(3); r = run_fma(A.iters, a, b); break;
  case 12: r = run_int(A.iters, lane); break;
  case 13: r = run_f32(A.iters, lane); break;
  case 14: r = run_alt<0>(A.iters / 16, a, b); break;   // per iter: 30 mfma + 96 bcast-fma
  case 15: r = run_alt<1>(A.iters / 16, a, b); break;
  default: break;
  }
  long long t1 = clock64();
  if (lane == 0) A.cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (r == 12345.678) A.out[0] = r;
}

int main() {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, dev));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double *out; long long *cyc;
  const int nblk = 256;
  CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, sizeof(long long) * nblk * 8));
  std::vector<long long> h(nblk * 8);
  struct Cfg { const char *name; int threads, lo, hi; double mfma_per_iter_lo, mfma_per_iter_hi; };
  Cfg cfgs[] = {
      {"mfma x1 chain (dependent), 1 wave/SIMD", 256, 1, 0, 4, 0},
      {"mfma x2 acc, 1 wave/SIMD", 256, 2, 0, 4, 0},
      {"mfma x4 acc, 1 wave/SIMD", 256, 3, 0, 4, 0},
      {"mfma x4 acc, 2 waves/SIMD", 512, 3, 3, 4, 4},
      {"mfma x1 chain, 2 waves/SIMD", 512, 1, 1, 4, 4},
      {"fma f64 x8, 1 wave/SIMD", 256, 4, 0, 0, 0},
      {"fma f64 x8, 2 waves/SIMD", 512, 4, 4, 0, 0},
      {"mfma x4 (lo) || fma (hi) co-resident", 512, 3, 4, 4, 0},
      {"mfma x1 (lo) || fma (hi) co-resident", 512, 1, 4, 4, 0},
      {"same wave: 2 mfma + 8 fma /iter", 256, 5, 0, 2, 0},
      {"same wave: 2 mfma + 16 fma /iter", 256, 6, 0, 2, 0},
      {"same wave: 2 mfma + 24 fma /iter", 256, 7, 0, 2, 0},
      {"readlane-bcast fma: 16/iter", 256, 8, 0, 0, 0},
      {"mfma x3 + 4 ds_read_b64 /iter (LDS operands)", 256, 9, 0, 3, 0},
      {"mfma x1 chain + 2 ds_read_b64 each (LDS operands)", 256, 10, 0, 3, 0},
      {"mfma x3 LDS operands, 2 waves/SIMD", 512, 9, 9, 3, 3},
      {"mfma x3 LDS (lo) || readlane-fma (hi)", 512, 9, 8, 3, 0},
      {"mfma x4 (lo) || fma prio3 (hi)", 512, 3, 11, 4, 0},
      {"int valu x8, 1 wave/SIMD", 256, 12, 0, 0, 0},
      {"mfma x4 (lo) || int valu (hi)", 512, 3, 12, 4, 0},
      {"f32 fma x8, 1 wave/SIMD", 256, 13, 0, 0, 0},
      {"mfma x4 (lo) || f32 fma (hi)", 512, 3, 13, 4, 0},
      {"alt 30mfma+96bcastfma, 1 wave/SIMD (iters/16)", 256, 14, 0, 0, 0},
      {"alt, 2 waves/SIMD no prio", 512, 14, 14, 0, 0},
      {"alt, 2 waves/SIMD prio3 in serial phase", 512, 15, 15, 0, 0},
      {"alt, 1 wave/SIMD prio variant", 256, 15, 0, 0, 0},
  };
  const int iters = 4096;
  for (const Cfg &c : cfgs) {
    Args A{out, cyc, iters, 0};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(c.threads), 0, 0, A, c.lo, c.hi); // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(c.threads), 0, 0, A, c.lo, c.hi);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * nblk * 8, hipMemcpyDeviceToHost));
    double lo = 0, hi = 0;
    for (int b = 0; b < nblk; ++b) { lo += h[b * 8 + 0]; hi += h[b * 8 + 4]; }
    lo /= nblk; hi /= nblk;
    printf("%-52s wall %.3f ms | wave0 %.0f cyc (%.1f cyc/iter) wave4 %.0f cyc (%.1f/iter) | eff clock %.2f GHz",
           c.name, ms, lo, lo / iters, c.threads > 256 ? hi : 0.0, c.threads > 256 ? hi / iters : 0.0,
           (lo > hi ? lo : hi) / (ms * 1e6));
    if (c.mfma_per_iter_lo > 0) printf(" | %.1f cyc/mfma(lo)", lo / (iters * c.mfma_per_iter_lo));
    printf("\n");
  }
  return 0;
}
