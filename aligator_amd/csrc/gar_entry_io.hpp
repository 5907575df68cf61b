// gar_entry_io.hpp -- the caller-facing entry points whose records differ from the device's under padding (gar_hip_solver::padded) or for a normalised terminal knot -- upload_stage, set_init, packed up / download, the getters -- the device-layout getters and the debug counters (inside extern "C").
// Part of the ONE translation unit gar_hip.cpp (included in place: it uses the solver struct and the helpers defined
// above its include line); split out for readability only.
#pragma once

// ---- caller-facing entry points whose records differ under padding (gar_hip_solver::padded) --------------------
int gar_hip_upload_stage(gar_hip_solver *s, int b, int t, const double *Q, const double *S, const double *R,
                         const double *q, const double *r, const double *A, const double *B, const double *f,
                         const double *C, const double *D, const double *d, const double *Gth, const double *Gx,
                         const double *Gu, const double *Gv, const double *gamma) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, t))
    return rc;
  GAR_MULTI(s, gar_hip_upload_stage(multi_owner(s, t), b, t, Q, S, R, q, r, A, B, f, C, D, d, Gth, Gx, Gu, Gv, gamma));
  return upload_stage_impl(s, b, t, Q, S, R, q, r, A, B, f, C, D, d, Gth, Gx, Gu, Gv, gamma);
}

static int upload_stage_impl(gar_hip_solver *s, int b, int t, const double *Q, const double *S, const double *R,
                             const double *q, const double *r, const double *A, const double *B, const double *f,
                             const double *C, const double *D, const double *d, const double *Gth, const double *Gx,
                             const double *Gu, const double *Gv, const double *gamma) {
  if (s->term_grown && t == s->horizon) // the caller's terminal A (0 x nx) and f are empty: zeros in the record
    A = f = s->term_zeros.data();
  if (!s->padded)
    return upload_stage_dev(s, b, t, Q, S, R, q, r, A, B, f, C, D, d, Gth, Gx, Gu, Gv, gamma);
  const gar_stage_meta &m = s->meta[t];
  const int nx = s->unx, nu = m.nu > 0 ? s->unu : 0, NX = m.nx, NU = m.nu;
  if (!Q || !q || !A || !f || (nu > 0 && (!S || !R || !r || !B)))
    return fail(GAR_HIP_ERR_ARG, "gar_hip_upload_stage: null block");
  if (s->staged) {
    // straight into the pinned staging record, one pass: real rows / columns copied column by column, the dummy
    // ones written beside them (no intermediate padded copy of the blocks); then the knot is one dirty range
    const gar_knot_offsets o = gar_knot_layout(NX, NU, 0, NX, 0);
    double *rec = s->h_prob + (int64_t)b * s->prob_doubles + m.in_off;
    const bool nt = s->stage_nt;
    auto put = [rec, nt](int64_t off, const double *src, int r, int c, int R, int C, double diag) {
      double *dst = rec + off;
      if (r == R) { // same column pitch (e.g. (56, 22) -> (56, 24): only controls are added): the real columns in one go
        stage_copy(dst, src, (size_t)r * (size_t)c, nt);
      } else {
        for (int j = 0; j < c; ++j) {
          std::memcpy(dst + (size_t)j * R, src + (size_t)j * r, sizeof(double) * (size_t)r);
          std::memset(dst + (size_t)j * R + r, 0, sizeof(double) * (size_t)(R - r));
        }
      }
      if (C > c)
        std::memset(dst + (size_t)c * R, 0, sizeof(double) * (size_t)R * (size_t)(C - c));
      if (diag != 0.0)
        for (int i = std::min(r, c); i < std::min(R, C); ++i)
          dst[(size_t)i * R + i] = diag;
    };
    if (s->qr_packed && t < s->horizon) {
      pack_lower(rec + o.Q, Q, nx, NX, 1.0);
      pack_lower(rec + o.R, R, nu, NU, 1.0);
    } else {
      put(o.Q, Q, nx, nx, NX, NX, 1.0);
      put(o.R, R, nu, nu, NU, NU, 1.0);
    }
    put(o.S, S, nx, nu, NX, NU, 0.0);
    put(o.q, q, nx, 1, NX, 1, 0.0);
    put(o.r, r, nu, 1, NU, 1, 0.0);
    put(o.A, A, nx, nx, NX, NX, 0.0);
    put(o.B, B, nx, nu, NX, NU, 0.0);
    put(o.f, f, nx, 1, NX, 1, 0.0);
    mark_dirty(s, b, m.in_off, m.in_off + gar_knot_doubles(NX, NU, 0, NX, 0));
    return flush_if_grown(s, b);
  }
  thread_local std::vector<double> bQ, bS, bR, bq, br, bA, bB, bf;
  return upload_stage_dev(s, b, t, padded_block(bQ, Q, nx, nx, NX, NX, 1.0), padded_block(bS, S, nx, nu, NX, NU, 0.0),
                          padded_block(bR, R, nu, nu, NU, NU, 1.0), padded_block(bq, q, nx, 1, NX, 1, 0.0),
                          padded_block(br, r, nu, 1, NU, 1, 0.0), padded_block(bA, A, nx, nx, NX, NX, 0.0),
                          padded_block(bB, B, nx, nu, NX, NU, 0.0), padded_block(bf, f, nx, 1, NX, 1, 0.0), nullptr,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int gar_hip_set_init(gar_hip_solver *s, int b, const double *G0, const double *g0) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  GAR_MULTI(s, multi_set_init(s, b, G0, g0));
  if (!s->padded)
    return set_init_dev(s, b, G0, g0);
  if (s->user_nc0 > 0 && (!G0 || !g0))
    return fail(GAR_HIP_ERR_ARG, "gar_hip_set_init: null block");
  // [G0 0; 0 -I], [g0; 0]: the dummy states start (and stay) at zero
  const int nc0u = s->user_nc0, nc0 = s->nc0, nx = s->unx, NX = s->pnx;
  thread_local std::vector<double> G, g;
  G.assign((size_t)nc0 * NX, 0.0);
  g.assign((size_t)nc0, 0.0);
  for (int j = 0; j < nx; ++j)
    for (int i = 0; i < nc0u; ++i)
      G[(size_t)j * nc0 + i] = G0[(size_t)j * nc0u + i];
  for (int i = 0; i < NX - nx; ++i)
    G[(size_t)(nx + i) * nc0 + nc0u + i] = -1.0;
  for (int i = 0; i < nc0u; ++i)
    g[i] = g0[i];
  return set_init_dev(s, b, G.data(), g.data());
}

int gar_hip_set_condensed_backward_ok(gar_hip_solver *s, double omega) {
  if (!s || !(omega >= 0.0))
    return fail(GAR_HIP_ERR_ARG, "bad backward-error bound");
  s->cond_backward_ok = omega;
  GAR_MULTI(s, multi_all(s, [&](gar_hip_solver *q) { return gar_hip_set_condensed_backward_ok(q, omega); }));
  return GAR_HIP_OK;
}

#ifdef GAR_CTRACE
extern "C" int gar_hip_debug_ctrace(long long *out) {
  long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gar::g_ctrace), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(gar::g_ctrace), z, sizeof(z)) != hipSuccess) return 2;
  return 0;
}
extern "C" int gar_hip_debug_ptrace(long long *out) {
  long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gar::g_ptrace), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(gar::g_ptrace), z, sizeof(z)) != hipSuccess) return 2;
  return 0;
}
extern "C" int gar_hip_debug_crtrace(long long *out) {
  long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gar::g_crtrace), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(gar::g_crtrace), z, sizeof(z)) != hipSuccess) return 2;
  return 0;
}
#endif
int gar_hip_condensed_resolved(gar_hip_solver *s, int b, int *out) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  GAR_MULTI(s, gar_hip_condensed_resolved(s->multi->subs[0], b, out));
  if (s->num_legs < 2 || !out)
    return fail(GAR_HIP_ERR_ARG, "condensed info needs leg mode");
  const int64_t nblk = 2 * s->num_legs, bs = (int64_t)s->nxb * s->nxb;
  const double *info = s->d_cscratch + (int64_t)b * s->cscratch_doubles + 4 * nblk * bs + 4 * nblk * s->nxb;
  double v = 0.0;
  if (int rc = d2h(s, &v, info + 3, 1))
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  *out = v != 0.0 ? 1 : 0;
  return GAR_HIP_OK;
}

int gar_hip_condensed_backward_error(gar_hip_solver *s, int b, double *out) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  GAR_MULTI(s, gar_hip_condensed_backward_error(s->multi->subs[0], b, out));
  if (s->num_legs < 2 || !out)
    return fail(GAR_HIP_ERR_ARG, "condensed info needs leg mode");
  const int64_t nblk = 2 * s->num_legs, bs = (int64_t)s->nxb * s->nxb;
  const double *info = s->d_cscratch + (int64_t)b * s->cscratch_doubles + 4 * nblk * bs + 4 * nblk * s->nxb;
  double v[3] = {0.0, 0.0, 0.0};
  if (int rc = d2h(s, v, info, 3))
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  *out = v[2] > 0.0 ? v[0] / v[2] : 0.0;
  return GAR_HIP_OK;
}

int gar_hip_get_solution(gar_hip_solver *s, int b, double *xs, double *us, double *vs, double *lbdas) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  GAR_MULTI(s, multi_get_solution(s, b, xs, us, vs, lbdas));
  if (!s->padded)
    return get_solution_dev(s, b, xs, us, vs, lbdas);
  std::vector<double> rec((size_t)s->sol_doubles);
  HIP_TRY(hipMemcpyAsync(rec.data(), s->d_sol + (int64_t)b * s->sol_doubles, sizeof(double) * rec.size(),
                         hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  strip_solution(s, rec.data(), xs, us, vs, lbdas);
  return GAR_HIP_OK;
}

int gar_hip_get_gains(gar_hip_solver *s, int b, int t, double *ff, double *fb, double *fth) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, t))
    return rc;
  GAR_MULTI(s, gar_hip_get_gains(multi_owner(s, t), b, t, ff, fb, fth));
  if (s->term_grown && t == s->horizon) {
    // the caller's terminal knot has nx2 = 0: its ff / fb / fth hold the nc rows [zff | Z | Zth] alone (they lead
    // the record's rows; a padded solver has nc = 0: nothing to hand back)
    const gar_stage_meta &m = s->meta[t];
    const int nc = m.nc, NR = m.nu + m.nc + m.nx2, NX = m.nx, NT = m.nth;
    if (nc == 0 || s->padded)
      return GAR_HIP_OK;
    std::vector<double> F((size_t)NR), Fb((size_t)NR * NX), Ft((size_t)NR * std::max(NT, 1));
    if (int rc = get_gains_dev(s, b, t, F.data(), Fb.data(), NT > 0 ? Ft.data() : nullptr))
      return rc;
    if (ff)
      std::copy(F.begin(), F.begin() + nc, ff);
    if (fb)
      std::copy(Fb.begin(), Fb.begin() + (size_t)nc * NX, fb);
    if (fth && NT > 0)
      std::copy(Ft.begin(), Ft.begin() + (size_t)nc * NT, fth);
    return GAR_HIP_OK;
  }
  if (!s->padded)
    return get_gains_dev(s, b, t, ff, fb, fth);
  const gar_stage_meta &m = s->meta[t];
  const int NR = m.nu + m.nx2, NX = m.nx, NT = m.nth;
  const int nx = s->unx, nu = m.nu > 0 ? s->unu : 0, nr = nu + nx, nt = NT > 0 ? nx : 0;
  std::vector<double> F((size_t)NR), Fb((size_t)NR * NX), Ft((size_t)NR * std::max(NT, 1));
  if (int rc = get_gains_dev(s, b, t, F.data(), Fb.data(), NT > 0 ? Ft.data() : nullptr))
    return rc;
  for (int r = 0; r < nr; ++r) {
    const int rd = gain_row(s, r, m.nu);
    if (ff)
      ff[r] = F[(size_t)rd];
    if (fb)
      for (int j = 0; j < nx; ++j)
        fb[(size_t)r * nx + j] = Fb[(size_t)rd * NX + j];
    if (fth)
      for (int j = 0; j < nt; ++j)
        fth[(size_t)r * nt + j] = Ft[(size_t)rd * NT + j];
  }
  return GAR_HIP_OK;
}

int gar_hip_get_value(gar_hip_solver *s, int b, int t, double *Vxx, double *vx, double *Vxt, double *Vtt,
                      double *vt) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, t))
    return rc;
  GAR_MULTI(s, gar_hip_get_value(multi_owner(s, t), b, t, Vxx, vx, Vxt, Vtt, vt));
  if (!s->padded)
    return get_value_dev(s, b, t, Vxx, vx, Vxt, Vtt, vt);
  const gar_stage_meta &m = s->meta[t];
  const int NX = m.nx, NT = m.nth, nx = s->unx, nt = NT > 0 ? nx : 0;
  std::vector<double> V((size_t)NX * NX), v((size_t)NX), Xt((size_t)NX * std::max(NT, 1)),
      Tt((size_t)std::max(NT, 1) * std::max(NT, 1)), tv((size_t)std::max(NT, 1));
  if (int rc = get_value_dev(s, b, t, V.data(), v.data(), Xt.data(), Tt.data(), tv.data()))
    return rc;
  for (int j = 0; j < nx; ++j) {
    if (Vxx)
      std::memcpy(Vxx + (size_t)j * nx, &V[(size_t)j * NX], sizeof(double) * (size_t)nx);
    if (vx)
      vx[j] = v[(size_t)j];
  }
  for (int j = 0; j < nt; ++j) {
    if (Vxt)
      std::memcpy(Vxt + (size_t)j * nx, &Xt[(size_t)j * NX], sizeof(double) * (size_t)nx);
    if (Vtt)
      std::memcpy(Vtt + (size_t)j * nt, &Tt[(size_t)j * NT], sizeof(double) * (size_t)nt);
    if (vt)
      vt[j] = tv[(size_t)j];
  }
  return GAR_HIP_OK;
}

int gar_hip_get_kkt(gar_hip_solver *s, int b, int t, double mueq, double *out) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, t))
    return rc;
  GAR_MULTI(s, gar_hip_get_kkt(multi_owner(s, t), b, t, mueq, out));
  if (!s->padded)
    return get_kkt_dev(s, b, t, mueq, out);
  if (!out)
    return fail(GAR_HIP_ERR_ARG, "null output");
  const gar_stage_meta &m = s->meta[t];
  const int NU = m.nu, nu = NU > 0 ? s->unu : 0;
  if (nu == 0)
    return GAR_HIP_OK;
  std::vector<double> K((size_t)NU * NU);
  if (int rc = get_kkt_dev(s, b, t, mueq, K.data()))
    return rc;
  for (int j = 0; j < nu; ++j)
    std::memcpy(out + (size_t)j * nu, &K[(size_t)j * NU], sizeof(double) * (size_t)nu);
  return GAR_HIP_OK;
}

int gar_hip_get_initial(gar_hip_solver *s, int b, double *kkt0_ff, double *kkt0_fth, double *thGrad,
                        double *thHess) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  GAR_MULTI(s, gar_hip_get_initial(s->multi->subs[0], b, kkt0_ff, kkt0_fth, thGrad, thHess));
  if (!s->padded)
    return get_initial_dev(s, b, kkt0_ff, kkt0_fth, thGrad, thHess);
  const int n0 = s->n0, NT = s->nth0, NX = s->pnx, nx = s->unx, nt = NT > 0 ? nx : 0, n0u = nx + s->user_nc0;
  std::vector<double> F((size_t)n0), Ft((size_t)n0 * std::max(NT, 1)), g((size_t)std::max(NT, 1)),
      H((size_t)std::max(NT, 1) * std::max(NT, 1));
  if (int rc = get_initial_dev(s, b, F.data(), Ft.data(), g.data(), H.data()))
    return rc;
  for (int r = 0; r < n0u; ++r) { // kkt0.ff = [x0; lbd0]: the real entries of each part
    const int rd = r < nx ? r : r - nx + NX;
    if (kkt0_ff)
      kkt0_ff[r] = F[(size_t)rd];
    if (kkt0_fth)
      for (int j = 0; j < nt; ++j)
        kkt0_fth[(size_t)r * nt + j] = Ft[(size_t)rd * NT + j];
  }
  for (int j = 0; j < nt; ++j) {
    if (thGrad)
      thGrad[j] = g[(size_t)j];
    if (thHess)
      std::memcpy(thHess + (size_t)j * nt, &H[(size_t)j * NT], sizeof(double) * (size_t)nt);
  }
  return GAR_HIP_OK;
}

/* ---- the device side of a (possibly padded) solver, for device-resident producers and consumers ---------------- */
int gar_hip_device_stage_layout(const gar_hip_solver *s, int t, int64_t out[11]) {
  if (int rc = check_bt(s, 0, t))
    return rc;
  const gar_stage_meta &m = s->meta[t];
  out[0] = m.nx; out[1] = m.nu; out[2] = m.nc; out[3] = m.nx2; out[4] = s->dims5[5 * (size_t)t + 4];
  out[5] = m.in_off; out[6] = m.fac_off; out[7] = m.x_off; out[8] = m.u_off; out[9] = m.v_off; out[10] = m.l_off;
  return GAR_HIP_OK;
}

gar_hip_solver *gar_hip_multi_create(int ndev, const int *dev_ids, int horizon, const int32_t *dims5, int nc0, int batch,
                                     int num_legs) {
  return multi_create(ndev, dev_ids, horizon, dims5, nc0, batch, num_legs);
}

int gar_hip_num_devices(const gar_hip_solver *s) { return !s ? 0 : (s->multi ? (int)s->multi->subs.size() : 1); }

int gar_hip_stage_device(const gar_hip_solver *s, int t) {
  if (int rc = check_bt(s, 0, t))
    return rc;
  return s->multi ? s->multi->subs[(size_t)s->multi->owner[(size_t)t]]->device : s->device;
}

const char *gar_hip_multi_exchange_name(const gar_hip_solver *s) {
  return (s && s->multi) ? (s->multi->pull ? "pull" : "copy") : "";
}

long long gar_hip_debug_alloc_count(void) { return g_alloc_count.load(std::memory_order_relaxed); }

int gar_hip_device_sizes(const gar_hip_solver *s, int64_t out[8]) {
  if (!s || !out)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  out[0] = s->prob_doubles; out[1] = s->fac_doubles; out[2] = s->sol_doubles; out[3] = s->nc0;
  out[4] = s->G0_off; out[5] = s->g0_off; out[6] = s->padded ? 1 : 0; out[7] = s->init_doubles;
  return GAR_HIP_OK;
}

int gar_hip_packed_stage_dims(const gar_hip_solver *s, int t, int32_t out[5]) {
  if (!s || !out || t < 0 || t > s->horizon)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_packed_stage_dims: bad argument");
  std::copy(&s->user_dims5[5 * (size_t)t], &s->user_dims5[5 * (size_t)t] + 5, out);
  return GAR_HIP_OK;
}

int gar_hip_device_record_format(const gar_hip_solver *s) {
  if (!s)
    return 0;
  return (s->qr_packed ? GAR_HIP_FMT_QR_PACKED : 0) | (s->vxx_packed ? GAR_HIP_FMT_VXX_PACKED : 0) |
         (s->fb_t2 ? GAR_HIP_FMT_FB_T2 : 0);
}

