// gar_cstr_seg.cpp -- the translation unit of the constrained segment legs (gar_cstr_seg.hpp: leg mode on problems with
// coupled constraints, on the serial constrained chain's stage kernels + a parameter recursion).  A translation unit of
// its own: see the header (and gar_wave_wide.cpp for why register-critical kernels do not share one).
#include "gar_cstr_seg.hpp"

namespace gar {

template <int NX, int NU, int NC> static void cseg_fill(CsegKernels *k) {
  k->backward[0] = gar_cseg_backward<NX, NU, NC, 0>;
  k->backward[1] = gar_cseg_backward<NX, NU, NC, 1>;
  k->backward[2] = gar_cseg_backward<NX, NU, NC, 2>;
  k->leg_end = gar_cseg_leg_end<NX, NU, NC>;
  k->leg_end_lds_doubles = cseg_leg_end_lds_doubles<NX, NU, NC>();
  k->chain = gar_cseg_param_chain<NX, NU, NC>;
  k->stage = gar_cseg_param_stage<NX, NU, NC>;
  k->forward = gar_cseg_forward<NX, NU, NC>;
  k->backward_lds_doubles = WaveCfg<NX, NU, NC>::total;
  k->chain_lds_doubles = cseg_chain_lds_doubles<NX>();
  k->stage_lds_doubles = cseg_stage_lds_doubles<NX, NU, NC>();
  k->chain_threads = GAR_CSEG_CHAIN_THREADS;
  k->stage_threads = GAR_CSEG_STAGE_THREADS;
  k->rec = CsegCfg<NX, NU, NC>::rec;
  k->scratch_doubles = [](int horizon, int num_legs) { return CsegCfg<NX, NU, NC>::doubles(horizon, num_legs); };
}

bool cseg_bind(int nx, int nu, int nc, CsegKernels *out) {
  if (nx == 36 && nu == 12 && nc == 32) cseg_fill<36, 12, 32>(out);
  else if (nx == 16 && nu == 8 && nc == 8) cseg_fill<16, 8, 8>(out);
  else if (nx == 8 && nu == 4 && nc == 4) cseg_fill<8, 4, 4>(out);
  else return false;
  return true;
}

} // namespace gar
