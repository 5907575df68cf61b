// gar_wave_wide.cpp -- the translation unit of gar_backward_wave<56, 24>: the ONE-wave-per-problem sweep of the wide
// shape (GAR_HIP_WIDE=single; the default on this shape is the two-wave kernel of gar_wave_pair.hpp).  The kernel needs
// all 512 registers and 1.3 KB of scratch per lane, and its code generation -- which spill goes where -- followed every
// unrelated change of the big translation unit it used to live in (inlining decisions are made per translation unit):
// in round 6 a change to the two-wave kernel alone made THIS kernel's results drift to 6e-8 of the oracle's (its code
// differed in 8 983 disassembly lines, its sources in none; GPU tests test_config4_talos_lq_shape and the randomised
// soak caught it).  In a translation unit of its own it compiles the same way whatever happens next door.
#include "gar_wave.hpp"

namespace gar {
template __global__ void gar_backward_wave<56, 24, 0>(MfmaParams, int);
} // namespace gar
