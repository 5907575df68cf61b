// TEST-ONLY: storage behind tests/emu/include/hip/hip_runtime.h
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace emu {
thread_local BlockCtx *block = nullptr;
int device_count = 1;
} // namespace emu
namespace gar {
alignas(16) double gar_smem[160 * 1024 / 8]; // the one dynamic-LDS region (blocks run one at a time)
}
extern "C" void emu_set_device_count(int n) { emu::device_count = n; }
extern "C" long long emu_memcpy_async_count(void) { return emu_memcpy_async_calls(); }
