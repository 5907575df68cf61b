// TEST-ONLY emulation of the slice of the HIP runtime and of the gfx950 wave64
// intrinsics that aligator_amd/csrc uses.  It lets the *unmodified* kernel
// sources run on host threads (one OS thread per lane, std::barrier for
// __syncthreads and for the lock-step of cross-lane instructions) so that
// index maps, LDS plans and barrier placement can be checked -- also under
// ThreadSanitizer -- without spending GPU minutes.  It is never built into,
// loaded by, or shipped with the product (aligator_amd/); the product has no
// CPU path and fails loudly without a HIP device.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef struct emu_stream_t *hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipEventDisableTiming = 2;
constexpr hipError_t hipErrorPeerAccessAlreadyEnabled = 704;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };

namespace emu {
struct WaveCtx {
  std::barrier<> bar{64};
  double a[64], b[64];
};
struct BlockCtx {
  std::barrier<> bar;
  std::vector<std::unique_ptr<WaveCtx>> waves;
  explicit BlockCtx(int nthr) : bar(nthr) {
    for (int i = 0; i < (nthr + 63) / 64; ++i)
      waves.emplace_back(new WaveCtx());
  }
};
extern thread_local BlockCtx *block;
extern int device_count; // tests flip this to check the "no GPU" error path
} // namespace emu

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

inline void __syncthreads() { emu::block->bar.arrive_and_wait(); }

typedef double emu_double4 __attribute__((ext_vector_type(4)));
// v_mfma_f64_16x16x4_f64: A[i][k] in lane i+16k, B[k][j] in lane j+16k,
// D[row=(l>>4)+4r][col=l&15] in lane l register r (cdna_hip_programming.md sec. 3)
inline emu_double4 emu_mfma_f64_16x16x4(double a, double b, emu_double4 c, int, int, int) {
  const int lane = threadIdx.x & 63;
  emu::WaveCtx &W = *emu::block->waves[threadIdx.x >> 6];
  W.a[lane] = a;
  W.b[lane] = b;
  W.bar.arrive_and_wait();
  emu_double4 d;
  const int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = (lane >> 4) + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; ++k)
      acc = std::fma(W.a[row + 16 * k], W.b[col + 16 * k], acc);
    d[r] = acc;
  }
  W.bar.arrive_and_wait();
  return d;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64 emu_mfma_f64_16x16x4
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products (blocks b).  Lane maps measured on
// MI355X (scripts/ubench/mfma4x4_probe.cpp): A_b[i][k] in lane 16k+4b+i, B_b[k][j] in lane
// 16k+4b+j, D_b[i][j] in lane 16i+4b+j (one double per lane).
inline double emu_mfma_f64_4x4x4(double a, double b, double c, int, int, int) {
  const int lane = threadIdx.x & 63;
  emu::WaveCtx &W = *emu::block->waves[threadIdx.x >> 6];
  W.a[lane] = a;
  W.b[lane] = b;
  W.bar.arrive_and_wait();
  const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
  double acc = c;
  for (int k = 0; k < 4; ++k)
    acc = std::fma(W.a[16 * k + 4 * blk + i], W.b[16 * k + 4 * blk + j], acc);
  W.bar.arrive_and_wait();
  return acc;
}
#define __builtin_amdgcn_mfma_f64_4x4x4f64 emu_mfma_f64_4x4x4

// ---- wave-level exchange (lock-step via the wave barrier) ----------------------
namespace emu {
inline WaveCtx &wave() { return *block->waves[threadIdx.x >> 6]; }
} // namespace emu
inline double __shfl(double v, int src) {
  emu::WaveCtx &W = emu::wave();
  const int lane = threadIdx.x & 63;
  W.a[lane] = v;
  W.bar.arrive_and_wait();
  const double r = W.a[src & 63];
  W.bar.arrive_and_wait();
  return r;
}
inline double __shfl_xor(double v, int mask) { return __shfl(v, (int)((threadIdx.x & 63) ^ mask)); }
inline int __shfl_xor(int v, int mask) { return (int)__shfl((double)v, (int)((threadIdx.x & 63) ^ mask)); }
inline int __builtin_amdgcn_readlane(int v, int src) {
  emu::WaveCtx &W = emu::wave();
  const int lane = threadIdx.x & 63;
  W.b[lane] = (double)v; // exact for 32-bit ints
  W.bar.arrive_and_wait();
  const int r = (int)W.b[src & 63];
  W.bar.arrive_and_wait();
  return r;
}
inline int __builtin_amdgcn_readfirstlane(int v) { // exec is full wherever the kernels call it
  return __builtin_amdgcn_readlane(v, 0);
}
inline unsigned long long __ballot(int pred) {
  emu::WaveCtx &W = emu::wave();
  const int lane = threadIdx.x & 63;
  W.a[lane] = pred ? 1.0 : 0.0;
  W.bar.arrive_and_wait();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (W.a[l] != 0.0)
      m |= 1ull << l;
  W.bar.arrive_and_wait();
  return m;
}
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline long long __double_as_longlong(double d) { long long b; std::memcpy(&b, &d, 8); return b; }
inline double __longlong_as_double(long long b) { double d; std::memcpy(&d, &b, 8); return d; }
// DPP moves used by wave_max_f64 (gar_device.hpp): row_shr:n (0x110+n), row_bcast:15 (0x142),
// row_bcast:31 (0x143); a lane without a valid source, or masked off, keeps `old`.
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool) {
  emu::WaveCtx &W = emu::wave();
  const int lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15;
  W.b[lane] = (double)src; // exact for 32-bit ints
  W.bar.arrive_and_wait();
  int from = -1;
  if (ctrl > 0x110 && ctrl <= 0x11f) {
    if (l16 >= ctrl - 0x110)
      from = lane - (ctrl - 0x110);
  } else if (ctrl == 0x142) {
    if (row > 0)
      from = (row - 1) * 16 + 15;
  } else if (ctrl == 0x143) {
    if (row >= 2)
      from = 31;
  }
  const bool en = ((row_mask >> row) & 1) && ((bank_mask >> (l16 >> 2)) & 1);
  const int r = (from >= 0 && en) ? (int)W.b[from] : old;
  W.bar.arrive_and_wait();
  return r;
}
// v_mov_b64_dpp row_newbcast:N (ctrl 0x150 + N): every lane reads lane N of its own 16-lane row
inline double __builtin_amdgcn_mov_dpp(double v, int ctrl, int row_mask, int bank_mask, bool) {
  emu::WaveCtx &W = emu::wave();
  const int lane = threadIdx.x & 63;
  W.a[lane] = v;
  W.bar.arrive_and_wait();
  double r = v;
  if (ctrl >= 0x150 && ctrl <= 0x15f && ((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane & 15) >> 2)) & 1))
    r = W.a[(lane & 48) | (ctrl - 0x150)];
  W.bar.arrive_and_wait();
  return r;
}
// v_permlane16_swap / v_permlane32_swap (gfx950), semantics probed on the hardware (scripts/ubench/swap_probe.cpp):
// vdst's odd rows (upper half) are exchanged with src0's even rows (lower half).  Result [0] = new vdst, [1] = new src0.
typedef unsigned emu_uint2 __attribute__((ext_vector_type(2)));
inline emu_uint2 emu_permlane_swap(unsigned vdst, unsigned src0, int span) { // span 16: rows, 32: halves
  emu::WaveCtx &W = emu::wave();
  const int lane = threadIdx.x & 63;
  W.a[lane] = (double)vdst; // exact for 32-bit values
  W.b[lane] = (double)src0;
  W.bar.arrive_and_wait();
  const bool upper = (lane & span) != 0;
  emu_uint2 r;
  r[0] = upper ? (unsigned)W.b[lane - span] : vdst;
  r[1] = upper ? src0 : (unsigned)W.a[lane + span];
  W.bar.arrive_and_wait();
  return r;
}
inline emu_uint2 __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned src0, bool, bool) { return emu_permlane_swap(vdst, src0, 16); }
inline emu_uint2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src0, bool, bool) { return emu_permlane_swap(vdst, src0, 32); }
inline unsigned long long __ballot(int pred);
inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return __ballot(p ? 1 : 0); }
inline int __double2hiint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b >> 32); }
inline int __double2loint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b & 0xffffffffll); }
inline double __hiloint2double(int hi, int lo) {
  long long b = ((long long)hi << 32) | (unsigned int)lo;
  double d; std::memcpy(&d, &b, 8); return d;
}
inline void __builtin_amdgcn_wave_barrier() {
  emu::wave().bar.arrive_and_wait();
}
#define __builtin_amdgcn_fence(order, ...) std::atomic_thread_fence(std::memory_order_seq_cst)
inline void __builtin_amdgcn_s_barrier() { emu::block->bar.arrive_and_wait(); }
inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }

inline long long clock64() { return 0; }
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
// global_load_lds_dwordx{1,4}: every lane copies `size` bytes from ITS global address to the wave-uniform LDS
// address + lane * size (the copy completes at once here; the kernels' s_waitcnt arithmetic is not exercised)
inline void __builtin_amdgcn_global_load_lds(const void *gsrc, void *lds_wave, int size, int offset, int /*aux*/) {
  const int lane = threadIdx.x & 63;
  std::memcpy((char *)lds_wave + offset + (size_t)lane * size, (const char *)gsrc + offset, (size_t)size);
}
inline double __builtin_amdgcn_rcp(double a) { return 1.0 / a; }
inline int atomicOr(int *p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

// ---- runtime -------------------------------------------------------------------
inline hipError_t hipGetDeviceCount(int *n) { *n = emu::device_count; return hipSuccess; }
inline int &emu_current_device() { static thread_local int d = 0; return d; }
inline hipError_t hipSetDevice(int d) { emu_current_device() = d; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = emu_current_device(); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 0; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return 0; }
inline long long &emu_memcpy_async_calls() { static long long n = 0; return n; } // (tests count the copies a call sequence issues)
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { ++emu_memcpy_async_calls(); std::memmove(d, s, n); return 0; }
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
inline hipError_t hipMemset2DAsync(void *d, size_t pitch, int v, size_t w, size_t h, hipStream_t) {
  for (size_t r = 0; r < h; ++r) std::memset((char *)d + r * pitch, v, w);
  return 0;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return 0; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
typedef struct emu_event_t *hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)std::malloc(1); return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return 0; }
// (launches run to completion before they return: every event is complete when it is waited for)
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { std::memmove(d, s, n); return 0; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = std::getenv("GAR_EMU_NO_PEER") ? 0 : 1; return 0; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "emu error"; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return 0; }

template <class Kernel, class... Args>
void hipLaunchKernelGGL(Kernel kernel, dim3 grid, dim3 block, size_t /*lds*/, hipStream_t,
                        Args... args) {
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        emu::BlockCtx ctx((int)block.x);
        std::vector<std::thread> thr;
        thr.reserve(block.x);
        for (unsigned t = 0; t < block.x; ++t)
          thr.emplace_back([&, t]() {
            emu::block = &ctx;
            threadIdx = dim3(t, 0, 0);
            blockIdx = dim3(bx, by, bz);
            blockDim = block;
            gridDim = grid;
            kernel(args...);
          });
        for (auto &th : thr)
          th.join();
      }
}
