// gar_select.hpp -- which kernel family serves a problem: the bind_* tables, select_kernel / select_leg_kernel, the padding decision (choose_padding) and configure().
// Part of the ONE translation unit gar_hip.cpp (included in place: it uses the solver struct and the helpers defined
// above its include line); split out for readability only.
#pragma once

// ---- specialised kernel dispatch ---------------------------------------------
template <int NX, int NU> void bind_mfma(gar_hip_solver *s) {
  s->mfma_kernel = gar::gar_backward_mfma<NX, NU>;
  s->mfma_fwd_kernel = gar::gar_forward_mfma<NX, NU>;
  s->mfma_fwd_lds_bytes = GAR_VXX_PACKED ? sizeof(double) * (size_t)gar_sym_packed_doubles(NX) : 0;
  s->fb_t2 = true;
  s->mfma_lds_doubles = gar::MfmaCfg<NX, NU>::total;
  s->kernel_name = "mfma<" + std::to_string(NX) + "," + std::to_string(NU) + ">";
  // Two backward kernels: one wave per problem (throughput: every SIMD runs its own problem) and
  // one 4-wave workgroup per problem (latency: a problem gets a whole CU; measured 2.2 ms vs
  // 3.0 ms per sweep while there are no more problems than CUs).  GAR_HIP_BACKWARD=wave|wg4
  // overrides the choice.
  const char *bw = gar_option("GAR_HIP_BACKWARD");
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device);
  const bool want_wave = bw ? std::string(bw) != "wg4" : s->batch > cus;
  // GAR_HIP_BACKWARD=pair: two waves per problem, the tile columns split between them
  // (gar_wave_pair.hpp; <= 256 registers per wave, so two waves share a SIMD)
  constexpr bool can_pair = (NX % 16) != 0 && ((NX >> 4) >= (gar::WaveCfg<NX, NU>::TW / 2)) && (gar::WaveCfg<NX, NU>::TW / 2) >= 1;
  if (bw && std::string(bw) == "pair" && can_pair) {
    if constexpr (can_pair) {
      s->wave_kernel = gar::gar_backward_pair<NX, NU>;
      s->wave_fused_init = false;
      s->wave_lds_doubles = gar::PairCfg<NX, NU>::total;
      s->wave_block_threads = 128;
      s->waves_per_block = 1;
      s->kernel_name = "pair<" + std::to_string(NX) + "," + std::to_string(NU) + ">";
    }
  } else if (want_wave) {
    s->wave_kernel = gar::gar_backward_wave<NX, NU>;
    // the initial stage is fused into the sweep when its packed kkt0 fits beside V in a
    // quarter of the CU's LDS (four waves per CU)
    const int with_init = gar::WaveCfg<NX, NU>::total_with_init(s->nc0);
    s->wave_fused_init = (size_t)with_init * sizeof(double) <= 40 * 1024 && s->nth0 == 0;
    // (F-DMA: the next knot's [A | B] image lies behind `total`, under the fused initial stage's kkt0)
    const int sweep = GAR_F_DMA ? gar::WaveCfg<NX, NU>::total_fdma : gar::WaveCfg<NX, NU>::total;
    s->wave_lds_doubles = s->wave_fused_init ? std::max(with_init, sweep) : sweep;
    s->waves_per_block = 1; // one 64-thread workgroup per problem (constant LDS base)
    s->kernel_name = "wave<" + std::to_string(NX) + "," + std::to_string(NU) + ">";
    s->qr_packed = GAR_QR_PACKED != 0; // the sweep reads only the lower triangles of Q and R (gar_layout.h)
    // the pipelined sweep's roll-out (gar_hip_set_pipeline): reads the packed Vxx records
    if (GAR_VXX_PACKED) {
      s->lean_fwd_kernel = gar::gar_forward_lean<NX, NU>;
      s->wave_half_kernel = gar::gar_backward_wave_half<NX, NU>;
      s->lean_fwd_used = gar::LeanFwdCfg<NX, NU>::USED; // (what it uses; pipe_plan decides what it asks for)
      s->wave_lds_doubles_small = sweep;
    }
  }
}

// Wide shapes (nx + nu > 64, e.g. the Talos walk's (56, 22) padded to (56, 24)): the one-wave-per-problem
// backward sweep only (gar_wave2.hpp; fb ROW-major = the generic record layout), the initial stage
// and the forward sweep on the generic kernels.
template <int NX, int NU> void bind_wide(gar_hip_solver *s) {
  // two waves per problem (the tile columns split between them) unless GAR_HIP_WIDE=single
  const char *w = gar_option("GAR_HIP_WIDE");
  const bool pair = !(w && std::string(w) == "single");
  const bool generic_fwd = w && std::string(w) == "generic-forward";
  if (!generic_fwd)
    s->mfma_fwd_kernel = gar::gar_forward_wide<NX, NU>; // row-major fb: fb_t2 stays false
  s->wave_fused_init = false;
  s->waves_per_block = 1;
  s->fb_t2 = false;
  if (pair) {
    // packed records (gar_wave_pair.hpp, GAR_PAIR_PACKED): lower triangles of Q, R in the knots, of Vxx in the factors
    // -- with the roll-out that reads them (the any-dimension roll-out reads the flag from its parameters)
    // (the any-dimension roll-out reads full Vxx blocks: GAR_HIP_WIDE=generic-forward keeps the full records)
    constexpr bool PKD = GAR_PAIR_PACKED && GAR_QR_PACKED && GAR_VXX_PACKED && (NX % 4 == 0);
    if (PKD && !generic_fwd) {
      s->wave_kernel = gar::gar_backward_pair<NX, NU, PKD>;
      s->qr_packed = true;
      s->wide_vxx_packed = true;
      s->mfma_fwd_kernel = gar::gar_forward_wide<NX, NU, true>;
      s->mfma_fwd_lds_bytes = sizeof(double) * (size_t)gar_sym_packed_doubles(NX);
    } else {
      s->wave_kernel = gar::gar_backward_pair<NX, NU, false>;
    }
    s->wave_lds_doubles = gar::PairCfg<NX, NU>::total;
    s->wave_block_threads = 128;
    s->kernel_name = "pair<" + std::to_string(NX) + "," + std::to_string(NU) + ">";
  } else {
    s->wave_kernel = gar::gar_backward_wave<NX, NU>;
    s->wave_lds_doubles = gar::WaveCfg<NX, NU>::total;
    s->kernel_name = "wave<" + std::to_string(NX) + "," + std::to_string(NU) + ">";
  }
}

template <int NX, int NU> void bind_leg(gar_hip_solver *s) {
  s->fb_t2 = true;
  // two waves per leg (plain part / parameter part) unless GAR_HIP_LEG_WAVES=1
  const char *lw = gar_option("GAR_HIP_LEG_WAVES");
  s->leg_waves = (lw && std::string(lw) == "1") ? 1 : 2;
  s->leg_bwd_kernel = s->leg_waves == 2 ? gar::gar_backward_wave_leg2<NX, NU> : gar::gar_backward_wave_leg<NX, NU>;
  s->leg_tuple_kernel = gar::gar_leg_tuples<NX, NU>;
  s->leg_fwd_kernel = gar::gar_forward_wave_leg<NX, NU>;
  s->leg_collapse_kernel = gar::gar_collapse_feedback_t2<NX, NU>;
  s->leg_lds_doubles = s->leg_waves == 2 ? gar::WaveCfg<NX, NU>::leg2_total : gar::WaveCfg<NX, NU>::leg_total;
  const char *ck = gar_option("GAR_HIP_CONDENSED");
  if (!(ck && std::string(ck) == "generic")) {
    const int lds = 4 * NX * NX + 16 * NX + NX + NX + (NX & 1) + (NX + 16) / 2 + 2 +
                    2 * (2 * s->num_legs) * NX + 2;
    if ((size_t)lds * sizeof(double) <= 160 * 1024) {
      s->cond_wave_kernel = gar::gar_condensed_wave<NX>;
      s->cond_wave_lds_doubles = lds;
    }
    if (!(ck && std::string(ck) == "chain")) {
      s->cyc_setup_kernel = gar::gar_cyclic_setup<NX>;
      s->cyc_reduce_kernel = gar::gar_cyclic_reduce<NX>;
      s->cyc_top_kernel = gar::gar_cyclic_top<NX>;
      s->cyc_backlevel_kernel = gar::gar_cyclic_backlevel<NX>;
      s->cyc_recover_kernel = gar::gar_cyclic_recover<NX>;
      s->cyc_lds_doubles = gar::CyclicLds<NX>::total;
      s->cyc_block_doubles = NX * NX;
    }
  }
  s->kernel_name = "wave_leg<" + std::to_string(NX) + "," + std::to_string(NU) + ">";
}

// the wide shape in leg mode: segment legs (gar_leg_seg.hpp) on the two-wave stage kernel
template <int NX, int NU> void bind_seg_leg(gar_hip_solver *s) {
  s->seg_bwd_kernel = gar::gar_backward_pair_leg<NX, NU>;
  s->seg_fwd_kernel = gar::gar_forward_wide_leg<NX, NU>;
  s->seg_lds_doubles = gar::PairCfg<NX, NU>::total;
  s->fb_t2 = false; // row-major fb: the generic roll-out, condensed solve and collapse serve the family
  s->kernel_name = "pair_leg<" + std::to_string(NX) + "," + std::to_string(NU) + ">";
}

// leg mode: uniform unconstrained problem whose every leg holds at least two knots
void select_leg_kernel(gar_hip_solver *s) {
  const int N = s->horizon;
  const char *lk = gar_option("GAR_HIP_LEGS");
  if (lk && std::string(lk) == "generic")
    return;
  if (N < 1 || s->nxb != s->dims5[0])
    return;
  const int nx = s->dims5[0], nu = s->dims5[1];
  bool any_nc = false;
  for (int t = 0; t <= N; ++t) {
    const int32_t *d = &s->dims5[5 * t];
    if (d[0] != nx || d[1] != (t < N ? nu : 0) || d[3] != nx || d[4] != 0)
      return;
    any_nc |= d[2] != 0;
  }
  // constrained knots: folded onto the unconstrained family (gar_fold.hpp); the generic leg kernels are the
  // fallback for problems with D != 0, so they must fit a CU's LDS
  const char *fe = gar_option("GAR_HIP_FOLD");
  if (any_nc && (!s->lds_error.empty() || (fe && fe[0] == '0')))
    return;
  for (int i = 0; i < s->num_legs; ++i) {
    int i0, i1;
    gar_get_work(N, i, s->num_legs, &i0, &i1);
    if (i1 - i0 < (i + 1 < s->num_legs ? 2 : 1))
      return;
  }
  if (nx == 36 && nu == 12) bind_leg<36, 12>(s);
  else if (nx == 32 && nu == 12) bind_leg<32, 12>(s);
  else if (nx == 16 && nu == 8) bind_leg<16, 8>(s);
  else if (nx == 12 && nu == 8) bind_leg<12, 8>(s);
  else if (nx == 12 && nu == 4) bind_leg<12, 4>(s);
  else if (nx == 8 && nu == 4) bind_leg<8, 4>(s);
  else if (nx == 56 && nu == 24 && !any_nc) {
    const char *sg = gar_option("GAR_HIP_SEG_LEGS");
    if (!(sg && sg[0] == '0') && (size_t)gar::leg_stage_lds_doubles(56, 24) * sizeof(double) <= 160 * 1024)
      bind_seg_leg<56, 24>(s);
  }
  s->fold = any_nc && s->leg_bwd_kernel != nullptr;
}

// uniform problems with NC constraints on every knot: the one-wave-per-problem kernels with the
// reduced KKT system factorised by the wave-scope Bunch-Kaufman (gar_wave.hpp, NC > 0)
template <int NX, int NU, int NC> void bind_cstr(gar_hip_solver *s) {
  s->wave_kernel = gar::gar_backward_wave<NX, NU, NC>;
  s->wave_coupled_kernel = gar::gar_backward_wave_coupled<NX, NU, NC>;
  s->wave_bk_kernel = gar::gar_backward_wave_bk<NX, NU, NC>;
  s->mfma_fwd_kernel = gar::gar_forward_mfma<NX, NU, NC>;
  s->mfma_fwd_lds_bytes = GAR_VXX_PACKED ? sizeof(double) * (size_t)gar_sym_packed_doubles(NX) : 0;
  s->fb_t2 = true;
  const int with_init = gar::WaveCfg<NX, NU, NC>::total_with_init(s->nc0);
  s->wave_fused_init = (size_t)with_init * sizeof(double) <= 64 * 1024 && s->nth0 == 0;
  s->wave_lds_doubles = s->wave_fused_init ? with_init : gar::WaveCfg<NX, NU, NC>::total;
  s->waves_per_block = 1;
  s->kernel_name = "wave<" + std::to_string(NX) + "," + std::to_string(NU) + "," + std::to_string(NC) + ">";
  s->qr_packed = GAR_QR_PACKED != 0; // the chain's three kernels read only the lower triangles of Q and R (gar_layout.h)
}

void select_kernel(gar_hip_solver *s) {
  s->fold = false;
  s->cseg_on = false;
  s->seg_bwd_kernel = nullptr;
  s->seg_fwd_kernel = nullptr;
  s->leg_bwd_kernel = nullptr;
  s->leg_tuple_kernel = nullptr;
  s->leg_fwd_kernel = nullptr;
  s->leg_collapse_kernel = nullptr;
  s->cond_wave_kernel = nullptr;
  s->cyc_setup_kernel = nullptr;
  s->cyc_reduce_kernel = nullptr;
  s->cyc_top_kernel = nullptr;
  s->cyc_backlevel_kernel = nullptr;
  s->cyc_recover_kernel = nullptr;
  s->mfma_kernel = nullptr;
  s->mfma_fwd_kernel = nullptr;
  s->mfma_fwd_lds_bytes = 0;
  s->wave_kernel = nullptr;
  s->wave_coupled_kernel = nullptr;
  s->wave_bk_kernel = nullptr;
  // (the pipelined sweep's kernels belong to the family bound below: a rebuild for other dimensions must not keep
  // launching the old shape's half-batch kernels over the new records)
  s->lean_fwd_kernel = nullptr;
  s->wave_half_kernel = nullptr;
  s->lean_fwd_used = 0;
  s->lean_fwd_lds_bytes = 0;
  s->wave_lds_doubles_small = 0;
  s->wave_fused_init = false;
  s->wave_block_threads = 64;
  s->fb_t2 = false;
  s->vxx_packed = false;
  s->wide_vxx_packed = false;
  s->qr_packed = false;
  {
    const char *ik = gar_option("GAR_HIP_INIT");
    s->init_closed = !(ik && std::string(ik) == "bk");
  }
  s->kernel_name = "generic";
  if (s->dense) {
    s->kernel_name = "dense";
    return;
  }
  const char *force = gar_option("GAR_HIP_FORCE_GENERIC");
  if (force && force[0] == '1')
    return;
  const int N = s->horizon;
  if (s->num_legs > 1) {
    select_leg_kernel(s);
    return;
  }
  if (N < 1)
    return;
  const gar_stage_meta &m0 = s->meta[0];
  if (m0.nth != 0 || m0.nx2 != m0.nx)
    return;
  for (int t = 1; t < N; ++t) {
    const gar_stage_meta &m = s->meta[t];
    if (m.nx != m0.nx || m.nu != m0.nu || m.nc != m0.nc || m.nth != 0 || m.nx2 != m0.nx ||
        m.in_off - s->meta[t - 1].in_off != s->meta[1 < N ? 1 : 0].in_off - m0.in_off)
      return;
  }
  // (the terminal knot's factor record is addressed through compile-time offsets that assume nx2 = nx rows of
  // [yff | Aff] in it; the nx2 = 0 terminal knot SolverProxDDP builds arrives here normalised to nx2 = nx --
  // normalise_terminal -- anything else is the any-dimension kernels')
  const gar_stage_meta &mt = s->meta[N];
  if (mt.nx != m0.nx || mt.nu != 0 || mt.nc != m0.nc || mt.nth != 0 || mt.nx2 != m0.nx)
    return;
  const int nx = m0.nx, nu = m0.nu;
  if (m0.nc != 0) { // every knot constrained (the reference's bench/gar-riccati.cpp shape)
    const int nc = m0.nc;
    if (nx == 36 && nu == 12 && nc == 32) bind_cstr<36, 12, 32>(s);
    else if (nx == 16 && nu == 8 && nc == 8) bind_cstr<16, 8, 8>(s);
    else if (nx == 8 && nu == 4 && nc == 4) bind_cstr<8, 4, 4>(s);
    s->vxx_packed = GAR_VXX_PACKED && s->fb_t2; // (serial one-wave family: gar_layout.h)
    return;
  }
  if (nx == 36 && nu == 12) bind_mfma<36, 12>(s);
  else if (nx == 32 && nu == 12) bind_mfma<32, 12>(s);
  else if (nx == 16 && nu == 8) bind_mfma<16, 8>(s);
  else if (nx == 12 && nu == 8) bind_mfma<12, 8>(s);
  else if (nx == 12 && nu == 4) bind_mfma<12, 4>(s);
  else if (nx == 8 && nu == 4) bind_mfma<8, 4>(s);
  else if (nx == 56 && nu == 24) bind_wide<56, 24>(s);
  // the serial one-wave family keeps the lower triangle of Vxx, packed (gar_layout.h); round 6: the two-wave wide family too
  s->vxx_packed = GAR_VXX_PACKED && (s->fb_t2 || s->wide_vxx_packed);
}

// (nx, nu) shapes with kernels of their own (bind_mfma / bind_leg / bind_wide / bind_seg_leg above)
struct SpecShape { int nx, nu; bool serial_only; };
constexpr SpecShape kSpecialised[] = {{36, 12, false}, {32, 12, false}, {16, 8, false}, {12, 8, false},
                                      {12, 4, false}, {8, 4, false},   {56, 24, false}};

// Decide the device dimensions from the caller's (see gar_hip_solver::padded).  GAR_HIP_PAD=0: never pad.
void choose_padding(gar_hip_solver *s) {
  s->dims5 = s->user_dims5;
  s->nc0 = s->user_nc0;
  s->padded = false;
  s->unx = s->unu = s->pnx = s->pnu = 0;
  const char *pe = gar_option("GAR_HIP_PAD");
  const int N = s->horizon;
  if (s->dense || (pe && pe[0] == '0') || N < 1)
    return;
  const int32_t *d0 = &s->user_dims5[0];
  const int nx = d0[0], nu = d0[1];
  if (nu == 0 || nx <= 0)
    return;
  for (int t = 0; t <= N; ++t) {
    const int32_t *d = &s->user_dims5[5 * t];
    if (d[0] != nx || d[1] != (t < N ? nu : 0) || d[2] != 0 || d[3] != nx || d[4] != 0)
      return;
  }
  long best = -1;
  int bx = 0, bu = 0;
  for (const SpecShape &sh : kSpecialised) {
    if (s->num_legs > 1 && sh.serial_only)
      continue;
    if (sh.nx == nx && sh.nu == nu)
      return; // the shape has its own kernels
    if (sh.nx >= nx && sh.nu >= nu) {
      const long cost = (long)sh.nx * (sh.nx + sh.nu);
      if (best < 0 || cost < best)
        best = cost, bx = sh.nx, bu = sh.nu;
    }
  }
  if (best < 0)
    return;
  s->padded = true;
  s->unx = nx;
  s->unu = nu;
  s->pnx = bx;
  s->pnu = bu;
  for (int t = 0; t <= N; ++t) {
    int32_t *d = &s->dims5[5 * t];
    d[0] = d[3] = bx;
    if (t < N)
      d[1] = bu;
  }
  s->nc0 = s->user_nc0 + (bx - nx); // the dummy states are pinned by extra rows of the initial constraint
}

int configure_padded_or_not(gar_hip_solver *s);

// Everything that can be decided and validated WITHOUT touching device memory: padding, both layouts, the LDS
// plan, leg-mode geometry, the kernel family.  create and cycle_append (on a trial object) share it.
int configure(gar_hip_solver *s) {
  choose_padding(s);
  for (;;) {
    const int rc = configure_padded_or_not(s);
    // Padded onto a specialised shape, but no kernel of that family binds (leg mode with a leg of fewer than two
    // knots, GAR_HIP_SEG_LEGS=0, parameter LDS beyond a CU, ...): the any-dimension kernels would then sweep the
    // PADDED shape -- several times the work and LDS of the caller's own, possibly beyond what fits.  Redo the
    // configuration on the caller's dimensions.
    if (s->padded && (rc != GAR_HIP_OK || !(s->wave_kernel || s->mfma_kernel || s->leg_bwd_kernel || s->seg_bwd_kernel))) {
      s->dims5 = s->user_dims5;
      s->nc0 = s->user_nc0;
      s->padded = false;
      s->unx = s->unu = s->pnx = s->pnu = 0;
      continue;
    }
    return rc;
  }
}

int configure_padded_or_not(gar_hip_solver *s) {
  if (int rc = build_layout(s))
    return rc;
  if (int rc = plan_lds(s))
    return rc;
  delete s->ulay;
  s->ulay = nullptr;
  if (s->padded) {
    gar_hip_solver *u = new gar_hip_solver();
    u->horizon = s->horizon;
    u->batch = s->batch;
    u->num_legs = s->num_legs;
    u->dense = s->dense;
    u->nc0 = s->user_nc0;
    u->dims5 = s->user_dims5;
    s->ulay = u;
    if (int rc = build_layout(u))
      return rc;
  }
  if (s->num_legs > 1) {
    const int J = s->num_legs, W = s->world;
    if (W < 1 || W > J || s->rank < 0 || s->rank >= W)
      return fail(GAR_HIP_ERR_ARG, "horizon sharding needs 1 <= ranks <= num_legs");
    s->leg_begin = (int)((long long)s->rank * J / W);
    s->leg_end = (int)((long long)(s->rank + 1) * J / W);
    s->legs_per_rank = (J + W - 1) / W; // chunk pitch of the gathered tuples (gar_generic.hpp, cond_tuple)
    int nxb = 0;
    for (const auto &m : s->meta)
      nxb = std::max(nxb, std::max(m.nx, m.nx2));
    if (s->nc0 > nxb)
      return fail(GAR_HIP_ERR_UNSUPPORTED, "leg mode needs nc0 <= nx");
    s->nxb = nxb;
    s->tuple_doubles = 3 * (int64_t)nxb * nxb + 2 * nxb;
  }
  select_kernel(s);
  if (!s->lds_error.empty() && !(s->wave_kernel || s->mfma_kernel || s->leg_bwd_kernel || s->seg_bwd_kernel))
    return fail(GAR_HIP_ERR_UNSUPPORTED, s->lds_error);
  delete s->flay;
  s->flay = nullptr;
  if (s->seg_bwd_kernel) { // scratch records of the plain kernels: the same knots, serial (nth = 0) layout
    gar_hip_solver *f = new gar_hip_solver();
    f->horizon = s->horizon;
    f->batch = s->batch;
    f->num_legs = 1;
    f->nc0 = s->nc0;
    f->dims5 = s->dims5;
    s->flay = f;
    if (int rc = build_layout(f))
      return rc;
  }
  if (s->fold) {
    gar_hip_solver *f = new gar_hip_solver();
    f->horizon = s->horizon;
    f->batch = s->batch;
    f->num_legs = s->num_legs;
    f->nc0 = s->nc0;
    f->dims5 = s->dims5;
    for (int t = 0; t <= s->horizon; ++t)
      f->dims5[5 * (size_t)t + 2] = 0;
    s->flay = f;
    if (int rc = build_layout(f))
      return rc;
    s->kernel_name += "+fold";
    // problems with D != 0: the constrained segment legs (gar_cstr_seg.hpp) where every knot carries the same number of
    // constraints and the shape has the serial constrained chain; their scratch records live in the flagged problem's
    // slice of the wave-leg family's factor buffer
    s->cseg_on = false;
    const char *cs = gar_option("GAR_HIP_CSTR_SEG_LEGS");
    const int nx = s->dims5[0], nu = s->dims5[1], nc = s->dims5[2];
    bool uniform_nc = nc > 0 && GAR_QR_PACKED != 0 && GAR_VXX_PACKED != 0;
    for (int t = 0; t <= s->horizon; ++t)
      uniform_nc &= s->dims5[5 * (size_t)t + 2] == nc;
    if (uniform_nc && !(cs && cs[0] == '0') && gar::cseg_bind(nx, nu, nc, &s->cseg) &&
        s->cseg.scratch_doubles(s->horizon, s->num_legs) <= f->fac_doubles) {
      s->cseg_on = true;
      s->qr_packed = true; // the chain's kernels read only the lower triangles of Q and R (gar_layout.h); the fold unpacks
      s->kernel_name += "|wave_seg<" + std::to_string(nx) + "," + std::to_string(nu) + "," + std::to_string(nc) + ">";
    }
  }
  // (see gar_hip_backward_legs_async)
  s->mu_divides = s->fold;
  for (int t = 0; t <= s->horizon; ++t) {
    const int32_t *d = &s->dims5[5 * (size_t)t];
    s->mu_divides |= d[2] > 0 && (d[1] == 0 || s->wave_kernel != nullptr);
  }
  return GAR_HIP_OK;
}

