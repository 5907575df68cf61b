// Does the PLACEMENT of the per-problem records matter to a one-wave-per-problem sweep?  4096 waves, each
// streaming through 256 records of 24 864 B (the north-star factor record), one record in flight per wave
// (as the forward sweep: the next record is requested, the current one consumed):
//   problem-major: record t of problem b at b * (256 * rec) + t * rec   (what the library does: a problem is
//                  one contiguous block; a wave walks its own 6.4 MB, 4096 different regions are live)
//   stage-major:   record t of problem b at t * (4096 * rec) + b * rec  (all waves inside the same 100 MB)
// hipcc -O3 --offload-arch=gfx950 -o stride_stream stride_stream.cpp && ./stride_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int REC = 24864 / 16; // 16-byte pieces per record (1554)
constexpr int PER_LANE = (REC + 63) / 64; // 25
template <int MODE> __global__ void __launch_bounds__(64) sweep(const double2_t *base, double *out, int batch, int nrec) {
  const int b = blockIdx.x, lane = threadIdx.x;
  double2_t cur[PER_LANE], nxt[PER_LANE];
  auto rec_ptr = [&](int t) { return base + (MODE == 0 ? ((size_t)b * nrec + t) * REC : ((size_t)t * batch + b) * REC); };
  auto load = [&](const double2_t *p, double2_t (&r)[PER_LANE]) {
#pragma unroll
    for (int q = 0; q < PER_LANE; ++q) { const int e = 64 * q + lane; r[q] = p[e < REC ? e : REC - 1]; }
  };
  load(rec_ptr(0), cur);
  double acc = 0.0;
  for (int t = 0; t < nrec; ++t) {
    if (t + 1 < nrec) load(rec_ptr(t + 1), nxt);
#pragma unroll
    for (int q = 0; q < PER_LANE; ++q) acc += cur[q].x * 1.0000001 + cur[q].y;
    // a serial dependence per record, as the state recursion has: ~70 dependent FMAs
#pragma unroll 8
    for (int i = 0; i < 72; ++i) acc = __builtin_fma(acc, 0.999999, 1e-9);
#pragma unroll
    for (int q = 0; q < PER_LANE; ++q) cur[q] = nxt[q];
  }
  out[(size_t)b * 64 + lane] = acc;
}
// The backward sweep's mix: per stage 29 472 B read (the knot) and 24 864 B written (the factor record), one knot in
// flight per wave.
constexpr int KNOT = 29472 / 16, KPL = (KNOT + 63) / 64; // 1842 pieces, 29 per lane
template <int NT> __global__ void __launch_bounds__(64) sweep_rw(const double2_t *in, double2_t *outrec, double *out, int nrec) {
  const int b = blockIdx.x, lane = threadIdx.x;
  double2_t cur[KPL], nxt[KPL];
  auto load = [&](const double2_t *p, double2_t (&r)[KPL]) {
#pragma unroll
    for (int q = 0; q < KPL; ++q) { const int e = 64 * q + lane; const double2_t *a = &p[e < KNOT ? e : KNOT - 1]; r[q] = (NT & 1) ? __builtin_nontemporal_load(a) : *a; }
  };
  const double2_t *pin = in + (size_t)b * nrec * KNOT;
  double2_t *pout = outrec + (size_t)b * nrec * REC;
  load(pin + (size_t)(nrec - 1) * KNOT, cur);
  double acc = 0.0;
  for (int t = nrec - 1; t >= 0; --t) {
    if (t > 0) load(pin + (size_t)(t - 1) * KNOT, nxt);
#pragma unroll
    for (int q = 0; q < KPL; ++q) acc += cur[q].x * 1.0000001 + cur[q].y;
#pragma unroll 8
    for (int i = 0; i < 72; ++i) acc = __builtin_fma(acc, 0.999999, 1e-9);
#pragma unroll
    for (int q = 0; q < PER_LANE; ++q) { const int e = 64 * q + lane; if (e < REC) { double2_t v = double2_t{acc, cur[q].x}; if (NT & 2) __builtin_nontemporal_store(v, &pout[(size_t)t * REC + e]); else pout[(size_t)t * REC + e] = v; } }
#pragma unroll
    for (int q = 0; q < KPL; ++q) cur[q] = nxt[q];
  }
  out[(size_t)b * 64 + lane] = acc;
}
int main() {
  const int batch = 4096, nrec = 256;
  const size_t bytes = (size_t)batch * nrec * REC * 16;
  double2_t *buf; double *out;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, (size_t)batch * 64 * 8));
  CHECK(hipMemset(buf, 0, bytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(sweep<0>, dim3(batch), dim3(64), 0, 0, buf, out, batch, nrec);
      else hipLaunchKernelGGL(sweep<1>, dim3(batch), dim3(64), 0, 0, buf, out, batch, nrec);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-14s rep %d: %7.3f ms  %6.2f TB/s\n", mode == 0 ? "problem-major" : "stage-major", rep, ms, bytes / (ms * 1e-3) / 1e12);
    }
  {
    double2_t *kin; CHECK(hipMalloc(&kin, (size_t)batch * nrec * KNOT * 16)); CHECK(hipMemset(kin, 0, (size_t)batch * nrec * KNOT * 16));
    const double tot = (double)batch * nrec * (KNOT + REC) * 16;
    for (int nt = 0; nt < 4; ++nt)
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      if (nt == 0) hipLaunchKernelGGL(sweep_rw<0>, dim3(batch), dim3(64), 0, 0, kin, buf, out, nrec);
      if (nt == 1) hipLaunchKernelGGL(sweep_rw<1>, dim3(batch), dim3(64), 0, 0, kin, buf, out, nrec);
      if (nt == 2) hipLaunchKernelGGL(sweep_rw<2>, dim3(batch), dim3(64), 0, 0, kin, buf, out, nrec);
      if (nt == 3) hipLaunchKernelGGL(sweep_rw<3>, dim3(batch), dim3(64), 0, 0, kin, buf, out, nrec);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-14s nt-load %d nt-store %d rep %d: %7.3f ms  %6.2f TB/s  (29 472 B read + 24 864 B written per stage: the backward sweep's bytes)\n", "read+write", nt & 1, (nt >> 1) & 1, rep, ms, tot / (ms * 1e-3) / 1e12);
    }
  }
  { // the same read+write stream at ONE wave per SIMD (what the backward kernel's registers allow): 40 KB of
    // dynamic LDS per workgroup -> four workgroups per CU
    double2_t *kin; CHECK(hipMalloc(&kin, (size_t)batch * nrec * KNOT * 16)); CHECK(hipMemset(kin, 0, (size_t)batch * nrec * KNOT * 16));
    const double tot = (double)batch * nrec * (KNOT + REC) * 16;
    CHECK(hipFuncSetAttribute((const void *)sweep_rw<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(sweep_rw<0>, dim3(batch), dim3(64), 40 * 1024, 0, kin, buf, out, nrec);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-14s 1 wave/SIMD rep %d: %7.3f ms  %6.2f TB/s\n", "read+write", rep, ms, tot / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
