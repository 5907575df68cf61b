// gar_wave_sweep.cpp -- the translation unit of the HEADLINE kernels: gar_backward_wave<NX, NU> (one wave per
// problem, serial in time, unconstrained; gar_wave.hpp + gar_wave2.hpp) for the six specialised shapes, and nothing
// else.  A translation unit of its own so that code-generation switches that pay on this kernel only can be applied
// to it alone (the Makefile's SWEEP_FLAGS): `-mllvm -amdgpu-mfma-vgpr-form=1` -- every MFMA takes and returns VGPRs
// instead of accumulating in AGPRs, which removes most of the v_accvgpr copies of the stage (318 per stage, DESIGN
// 5.1) -- measured on the whole library in round 3: backward -2.5 %, forward +10 % (not adopted then: the switch is
// per translation unit; and in round 4 the same switch crashes the compiler on gar_backward_wave<56,24>).
// gar_hip.cpp declares these instantiations `extern template` and takes their addresses; the kernels' device code
// lives in this object's code object.
#include "gar_wave.hpp"

namespace gar {
#define GAR_SWEEP_INSTANCE(NX, NU)                                                                                      \
  template __global__ void gar_backward_wave<NX, NU, 0>(MfmaParams, int);                                              \
  template __global__ void gar_backward_wave_half<NX, NU>(MfmaParams, int); // (the pipelined schedule's launches)
GAR_SWEEP_SHAPES(GAR_SWEEP_INSTANCE)
#undef GAR_SWEEP_INSTANCE
} // namespace gar
