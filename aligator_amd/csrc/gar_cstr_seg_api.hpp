// gar_cstr_seg_api.hpp -- what gar_hip.cpp sees of the constrained segment legs (gar_cstr_seg.hpp; the kernels are
// instantiated in gar_cstr_seg.cpp, a translation unit of their own): the parameter block of the parameter kernels and
// the binding of a shape to its kernels.
#pragma once
#include "gar_mfma.hpp"

namespace gar {

struct CsegParams {
  const gar_stage_meta *meta; // the solver's layout: caller-visible records (nth = nx on non-final legs)
  const double *prob;         // the caller's knots (Q, R as packed lower triangles: gar_layout.h)
  const double *fac2;         // scratch records written by the plain part (CsegCfg)
  double *fac;                // caller-visible records
  int *status;
  const int *only;            // per problem: 1 = this family's
  long long prob_stride, fac_stride, fac2_stride;
  long long in_off0, in_rec;
  int horizon, num_legs, leg_begin, local_legs;
  double mueq;
};

constexpr int kCsegReenter = 1, kCsegSingle = 2; // flags of the plain kernels (gar_cstr_seg.hpp)
struct CsegFwdParams {
  const gar_stage_meta *meta; // caller-visible layout
  const double *fac;          // caller-visible records (row-major fb / fth, full Vxx)
  double *sol;
  const double *csol;         // condensed solution [problem][2 * legs][nxb]
  const int *only;
  long long fac_stride, sol_stride;
  int horizon, num_legs, leg_begin, nxb, nc0;
};

struct CsegKernels {
  void (*backward[3])(MfmaParams, int, int, const int *, int) = {nullptr, nullptr, nullptr}; // the chain: decoupled, coupled, LDS Bunch-Kaufman
  void (*leg_end)(MfmaParams, int, int, const int *) = nullptr; // the leg-end stage of every non-final leg, a workgroup each
  void (*chain)(CsegParams) = nullptr;
  void (*stage)(CsegParams) = nullptr;
  void (*forward)(CsegFwdParams) = nullptr; // the roll-out, one wave per (leg, problem)
  int backward_lds_doubles = 0, leg_end_lds_doubles = 0, chain_lds_doubles = 0, stage_lds_doubles = 0, chain_threads = 0, stage_threads = 0;
  long long rec = 0;               // pitch of the scratch records (the terminal knot's at horizon * rec)
  long long (*scratch_doubles)(int horizon, int num_legs) = nullptr; // per problem
};

// the kernels of shape (nx, nu, nc), if the family has them
bool cseg_bind(int nx, int nu, int nc, CsegKernels *out);

} // namespace gar
