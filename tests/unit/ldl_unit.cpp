// TEST-ONLY: the workgroup-level L D L^T building blocks of aligator_amd/csrc/gar_device.hpp behind a C entry
// point, so that tests can drive them on a matrix of their choosing (definite, indefinite, with 2x2 pivots).
// Built two ways from this one file: for the wave emulator (tests/emu, host threads) and with hipcc for gfx950.
#include "gar_generic.hpp" // gar_smem, gar_device.hpp

namespace {
struct UnitParams {
  double *A;  // n x n column-major: in the matrix (lower triangle read), out the factorised block
  double *X;  // n x ncols column-major: in the right-hand sides, out the solution
  double *sub;
  int *piv;
  int *info;  // [0] what the definite factorisation returned (-1: not run), [1] what Bunch-Kaufman returned (-1: not run),
              // [2] cycles of the factorisation, [3] cycles of the solve (0 on the emulator)
  int n, ncols, definite_first, x_rowmajor; // X: n x ncols column-major, or row-major when x_rowmajor
  int blocked;                              // 1: wg_bk_factor_blocked, 2: the same on the PACKED lower triangle
};

__global__ void __launch_bounds__(1024) ldl_unit_kernel(UnitParams P) {
  using namespace gar;
  const WG w = wg_self();
  const int n = P.n, nc = P.ncols;
  const bool packed = P.blocked == 2; // the matrix lives in LDS as its packed lower triangle (column j: rows j .. n-1)
  const int asz = packed ? n * (n + 1) / 2 : n * n;
  double *a = gar_smem, *x = a + ((asz + 1) & ~1), *wk = x + ((n * nc + 1) & ~1), *sub = wk + GAR_LDL_PANEL * n + 2;
  int *piv = (int *)(sub + n + (n & 1)), *ctrl = piv + n + 8;
  for (int e = w.tid; e < n * n; e += w.nthr) {
    const int j = e / n, i = e - j * n;
    if (!packed)
      a[e] = P.A[e];
    else if (i >= j)
      a[bk_idx<GAR_PACKED_LOWER>(i, j, n)] = P.A[e];
  }
  for (int e = w.tid; e < n * nc; e += w.nthr)
    x[e] = P.X[e];
  __syncthreads();
  int r_def = -1, r_bk = -1;
  const long long c0 = clock64();
  if (P.definite_first && !packed) {
    r_def = wg_ldl_definite_factor(w, n, a, n, sub, piv, wk, ctrl);
    if (r_def) {
      for (int e = w.tid; e < n * n; e += w.nthr)
        a[e] = P.A[e];
      __syncthreads();
    }
  }
  if (r_def != 0 && packed)
    r_bk = wg_bk_factor_blocked<GAR_PACKED_LOWER>(w, n, a, n, sub, piv, ctrl, wk);
  else if (r_def != 0 && P.blocked == 1)
    r_bk = wg_bk_factor_blocked(w, n, a, n, sub, piv, ctrl, wk);
  else if (r_def != 0)
    r_bk = wg_bk_factor(w, n, a, n, sub, piv, ctrl);
  const long long c1 = clock64();
  if (packed)
    wg_bk_solve<GAR_PACKED_LOWER>(w, n, a, n, sub, piv, x, P.x_rowmajor ? nc : 1, P.x_rowmajor ? 1 : n, nc);
  else if (P.x_rowmajor)
    wg_bk_solve(w, n, a, n, sub, piv, x, nc, 1, nc);
  else
    wg_bk_solve(w, n, a, n, sub, piv, x, 1, n, nc);
  const long long c2 = clock64();
  for (int e = w.tid; e < n * n; e += w.nthr) {
    const int j = e / n, i = e - j * n;
    if (!packed)
      P.A[e] = a[e];
    else if (i >= j)
      P.A[e] = a[bk_idx<GAR_PACKED_LOWER>(i, j, n)];
  }
  for (int e = w.tid; e < n * nc; e += w.nthr)
    P.X[e] = x[e];
  for (int e = w.tid; e < n; e += w.nthr) {
    P.sub[e] = sub[e];
    P.piv[e] = piv[e];
  }
  if (w.tid == 0) {
    P.info[0] = r_def;
    P.info[1] = r_bk;
    P.info[2] = (int)(c1 - c0);
    P.info[3] = (int)(c2 - c1);
  }
}
} // namespace

// host pointers in, host pointers out; returns 0 or a HIP error code
extern "C" int gar_ldl_unit(int n, int ncols, int definite_first, int x_rowmajor, int blocked, int threads, double *A, double *X, double *sub, int *piv,
                            int *info) {
  UnitParams P{};
  P.n = n;
  P.ncols = ncols;
  P.definite_first = definite_first;
  P.x_rowmajor = x_rowmajor;
  P.blocked = blocked;
  const size_t bA = sizeof(double) * n * n, bX = sizeof(double) * n * ncols;
#define TRY(e)                                                                                                         \
  do {                                                                                                                 \
    hipError_t err_ = (e);                                                                                             \
    if (err_ != hipSuccess)                                                                                            \
      return (int)err_ ? (int)err_ : -1;                                                                               \
  } while (0)
  TRY(hipMalloc((void **)&P.A, bA));
  TRY(hipMalloc((void **)&P.X, bX ? bX : 8));
  TRY(hipMalloc((void **)&P.sub, sizeof(double) * n));
  TRY(hipMalloc((void **)&P.piv, sizeof(int) * n));
  TRY(hipMalloc((void **)&P.info, sizeof(int) * 4));
  TRY(hipMemcpy(P.A, A, bA, hipMemcpyHostToDevice));
  TRY(hipMemcpy(P.X, X, bX, hipMemcpyHostToDevice));
  const size_t lds = sizeof(double) * (size_t)(n * n + n * ncols + (GAR_LDL_PANEL + 2) * n + 64) + sizeof(int) * (size_t)(n + 48);
  TRY(hipFuncSetAttribute((const void *)ldl_unit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ldl_unit_kernel, dim3(1), dim3(threads > 0 ? threads : 256), lds, (hipStream_t) nullptr, P);
  TRY(hipGetLastError());
  TRY(hipStreamSynchronize((hipStream_t) nullptr));
  TRY(hipMemcpy(A, P.A, bA, hipMemcpyDeviceToHost));
  TRY(hipMemcpy(X, P.X, bX, hipMemcpyDeviceToHost));
  TRY(hipMemcpy(sub, P.sub, sizeof(double) * n, hipMemcpyDeviceToHost));
  TRY(hipMemcpy(piv, P.piv, sizeof(int) * n, hipMemcpyDeviceToHost));
  TRY(hipMemcpy(info, P.info, sizeof(int) * 4, hipMemcpyDeviceToHost));
  hipFree(P.A);
  hipFree(P.X);
  hipFree(P.sub);
  hipFree(P.piv);
  hipFree(P.info);
  return 0;
#undef TRY
}
