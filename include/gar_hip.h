/*
 * gar_hip.h -- C ABI of the MI355X-native gar Riccati/LQR backend.
 *
 * This is the drop-in boundary: a plain C interface (extern "C", raw pointers
 * and sizes, no C++/torch types, no exceptions across it) exporting exactly
 * what a `gar::RiccatiSolverBase<double>` subclass needs to serve the six
 * calls SolverProxDDP makes on `linear_solver_`
 * (/root/reference/include/aligator/solvers/proxddp/solver-proxddp.hxx:208,
 * 608-611, 619, 624-625, 631-632).  INTEGRATION.md shows the reference-side
 * binding (a `HipRiccatiSolver : gar::RiccatiSolverBase<double>` of ~80 lines).
 *
 * Reference interface replaced, entry point by entry point:
 *   gar_hip_solver_create        ProximalRiccatiSolver::ProximalRiccatiSolver
 *                                (gar/proximal-riccati.hxx:13-31) and
 *                                ParallelRiccatiSolver ctor + initialize()
 *                                (gar/parallel-solver.hxx:32-82, 261-287)
 *   gar_hip_solver_create_dense  RiccatiSolverDense::RiccatiSolverDense
 *                                (gar/dense-riccati.hxx:12-46): the stage-dense solver over
 *                                DenseKernel (gar/dense-kernel.hpp:55-209); every other entry
 *                                point serves it unchanged, with nu+nc+2*nx2 gain rows
 *   gar_hip_upload_stage         reads of LqrKnotTpl blocks (gar/lqr-problem.hpp:45-56);
 *                                the reference re-reads the caller's problem on every
 *                                backward() (proximal-riccati.hxx:37)
 *   gar_hip_set_init             LqrProblemTpl::G0, g0 (gar/lqr-problem.hpp:111-112)
 *   gar_hip_backward             RiccatiSolverBase::backward (gar/riccati-base.hpp:20):
 *                                ProximalRiccatiSolver::backward (proximal-riccati.hxx:34-62),
 *                                ParallelRiccatiSolver::backward (parallel-solver.hxx:132-206)
 *   gar_hip_forward              RiccatiSolverBase::forward (riccati-base.hpp:22-25):
 *                                proximal-riccati.hxx:65-77, parallel-solver.hxx:209-243
 *   gar_hip_collapse_feedback    RiccatiSolverBase::collapseFeedback (riccati-base.hpp:33,
 *                                parallel-solver.hpp:41-51)
 *   gar_hip_get_gains            getFeedforward / getFeedback (riccati-base.hpp:34-35)
 *   gar_hip_get_value / _initial public StageFactor::vm, kkt0, thGrad, thHess
 *                                (riccati-kernel.hpp:33-39,113-123; proximal-riccati.hpp:40-43)
 *   gar_hip_cycle_append         RiccatiSolverBase::cycleAppend (riccati-base.hpp:28,
 *                                proximal-riccati.hxx:79-86, parallel-solver.hxx:246-258)
 *
 * New relative to the reference (its own axis for reaching the HBM roofline,
 * SURVEY.md section 2c): a solver owns `batch` independent problems of identical
 * dimensions and sweeps them in one launch.
 *
 * Conventions: all host matrices are column-major float64 unless stated
 * (fb / fth are ROW-major like StageFactor::fb, riccati-kernel.hpp:96-98).
 * Return value: 0 = ok, <0 = error (gar_hip_last_error()).  Calls are
 * synchronous on return unless suffixed _async.  One solver <-> one host
 * thread at a time.  The library owns all device memory; the caller owns all
 * host buffers.  No device allocation happens inside backward/forward (the
 * reference's ALIGATOR_NOMALLOC_SCOPED contract, proximal-riccati.hxx:35).
 */
#ifndef GAR_HIP_H
#define GAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gar_hip_solver gar_hip_solver;

#define GAR_HIP_OK 0
#define GAR_HIP_ERR_ARG (-1)     /* bad argument / index out of range         */
#define GAR_HIP_ERR_DEVICE (-2)  /* HIP runtime error                          */
#define GAR_HIP_ERR_UNSUPPORTED (-3) /* dims exceed what fits one CU's LDS     */
#define GAR_HIP_ERR_FACTOR (-4)  /* a stage LDL^T hit an exactly-zero pivot
                                    column (riccati-kernel.hxx:239-241)       */

/* ---- library ------------------------------------------------------------ */
const char *gar_hip_version(void);
const char *gar_hip_last_error(void);
int gar_hip_device_count(void);
/* Device and pinned-host allocations the library has made so far in this process.  The reference runs backward /
 * forward under ALIGATOR_NOMALLOC_SCOPED (gar/proximal-riccati.hxx:35, tests/nomalloc.cpp); the same contract here:
 * the count does not move across gar_hip_backward / gar_hip_forward (tests/test_nomalloc.py). */
long long gar_hip_debug_alloc_count(void);
/* Measurement aid, no reference counterpart (bench.py's roofline.stream_ceiling): milliseconds (best of `reps`)
 * that `batch` one-wave-per-problem streams need for the BYTES of a serial-in-time backward sweep and nothing
 * else -- per stage in_bytes read (one knot ahead in flight), out_bytes written, a 72-FMA dependent chain -- i.e.
 * what this GPU's HBM sustains for the sweep's read/write mix and walk.  Allocates and frees its own buffers
 * (batch * horizon * (in + out) bytes); in_bytes <= 32 KiB, out_bytes <= 28 KiB.  Negative on error.
 * reps < 0: the same walk with TWO knots requested ahead of the one being consumed, |reps| repetitions. */
double gar_hip_stream_ceiling_ms(int device, int batch, int horizon, int64_t in_bytes_per_stage,
                                 int64_t out_bytes_per_stage, int reps);

/* Measurement aid beside it (roofline.stream_ceiling.plain_copy): milliseconds (best of `reps`) of a plain
 * grid-stride 16-byte-per-lane copy that moves `bytes_moved` bytes in total (half read, half written) -- the
 * kernel the guide's "achievable" HBM figure describes.  Allocates and frees bytes_moved bytes.  Negative on error. */
double gar_hip_copy_ceiling_ms(int device, int64_t bytes_moved, int reps);

/* ---- layout queries (pure host arithmetic, no GPU needed) ---------------- */
/* doubles in the packed record of one knot / one factor (csrc/gar_layout.h) */
int64_t gar_hip_knot_doubles(const int32_t dims5[5]);
int64_t gar_hip_factor_doubles(const int32_t dims5[5]);

/* ---- lifetime ------------------------------------------------------------ */
/* dims5: (horizon+1) x {nx,nu,nc,nx2,nth} exactly as LqrKnotTpl's fields -- the CALLER's dimensions: what
 *   `knot.nx, knot.nu, ...` say (ProximalRiccatiSolver's constructor reads them off the problem,
 *   proximal-riccati.hxx:24-27).  Every entry point below that takes or returns a block, a packed problem, a
 *   solution or a gain speaks these dimensions.
 * Kernel selection and padding happen behind this call: a uniform unconstrained, unparameterised problem whose
 *   (nx, nu) has no kernel family of its own -- e.g. the Talos walk's (56, 22), BASELINE configs[2]'s (12, 6),
 *   configs[0]'s (4, 2) -- runs on the smallest specialised shape that holds it ((56, 24), (12, 8), (8, 4)), with
 *   dummy controls / states that solve to exactly zero and are stripped from every result; gar_hip_kernel_name
 *   reports the family that runs.  The device records then have the padded dimensions: device-resident producers
 *   and consumers (gar_hip_device_*) ask gar_hip_device_stage_layout.  GAR_HIP_PAD=0 in the environment disables it.
 * A terminal knot given with nx2 = 0 (as SolverProxDDP builds it, solvers/proxddp/workspace.hxx:54-55; tests/gar/ gives
 *   it nx2 = nx) is kept as nx2 = nx -- the uniform record every kernel family addresses -- with zeros for its A, f,
 *   which nothing of the algorithm reads (riccati-kernel.hxx:130-193): gar_hip_packed_stage_dims.
 * num_legs = 1: serial-in-time ProximalRiccatiSolver semantics.
 * num_legs >= 2: ParallelRiccatiSolver(problem, num_legs) semantics; like the
 *   reference, every knot of a non-final leg is (implicitly) re-parameterised
 *   with nth = nx2 of the leg's last knot.
 * Horizon sharding over ranks (one rank per GPU): _ranked -- this solver owns legs [rank J / W, (rank+1) J / W) of
 *   J = num_legs over W = world ranks, any 1 <= W <= J (the reference takes any thread count >= 2,
 *   parallel-solver.hxx:42-46); _sharded is the even split [leg_begin, leg_end), kept for callers that count legs. */
gar_hip_solver *gar_hip_solver_create(int device, int horizon, const int32_t *dims5,
                                      int nc0, int batch, int num_legs);
gar_hip_solver *gar_hip_solver_create_ranked(int device, int horizon, const int32_t *dims5, int nc0, int batch,
                                             int num_legs, int rank, int world);
gar_hip_solver *gar_hip_solver_create_sharded(int device, int horizon,
                                              const int32_t *dims5, int nc0, int batch,
                                              int num_legs, int leg_begin, int leg_end);
/* ONE process, SEVERAL devices: ParallelRiccatiSolver(problem, num_legs) semantics with the legs split over the `ndev`
 * devices dev_ids[0..ndev) -- device r owns legs [r J / ndev, (r+1) J / ndev), get_work over devices
 * (gar/parallel-solver.hxx:23-28) -- behind the SAME handle type, so that the one RiccatiSolverBase object
 * SolverProxDDP holds (solvers/proxddp/solver-proxddp.hpp:56,181) shards its horizon over the GPUs of the node
 * without a process per GPU.  The boundary exchange the reference performs as the barrier closing its OpenMP region
 * (parallel-solver.hxx:150-169) happens INSIDE gar_hip_backward: every device sweeps its legs on its own stream,
 * then reads the other devices' boundary tuples (3 nx^2 + 2 nx doubles per leg) -- one gather kernel over
 * peer-mapped buffers (xGMI) when every pair of devices has peer access, hipMemcpyPeerAsync otherwise
 * (GAR_HIP_MULTI_EXCHANGE=copy forces it; gar_hip_multi_exchange_name tells) -- ordered by HIP events, no host
 * synchronisation, and solves the condensed system redundantly.  Every entry point of this header then serves the
 * handle: uploads and per-stage getters go to the stage's device, G0 / g0 and settings to all, the solution and the
 * bulk read-back are merged (each device copies only its own stages, all devices at once).  Entry points that take
 * or return DEVICE pointers (gar_hip_device_*, gar_hip_upload_packed_device, gar_hip_update_lq_subproblem_device,
 * gar_hip_set_stream) answer GAR_HIP_ERR_UNSUPPORTED / NULL.  dev_ids may name the same device more than once
 * (several ranked solvers sharing one GPU: how the path is tested on a one-GPU box).  ndev == 1 returns the plain
 * solver of gar_hip_solver_create (num_legs == 1: serial in time); ndev >= 2 needs num_legs >= ndev.  ndev <= 16. */
gar_hip_solver *gar_hip_multi_create(int ndev, const int *dev_ids, int horizon, const int32_t *dims5, int nc0,
                                     int batch, int num_legs);
/* devices behind the handle (1 for every other constructor); the device that holds stage t; "pull" / "copy" / "" */
int gar_hip_num_devices(const gar_hip_solver *s);
int gar_hip_stage_device(const gar_hip_solver *s, int t);
const char *gar_hip_multi_exchange_name(const gar_hip_solver *s);
/* RiccatiSolverDense(problem): serial in time, any dimensions.  backward() factorises the whole
 * (nu+nc+2*nx2)^2 stage matrix (gar/dense-kernel.hpp:100-172); gar_hip_get_gains then returns
 * ff (nu+nc+2*nx2) and fb (nu+nc+2*nx2, nx) row-major = block rows [K; Z; L; Y] as
 * getFeedforward/getFeedback do (dense-riccati.hpp:49-50); gar_hip_get_value returns Pxx, px, Pxt,
 * Ptt, pt (dense-riccati.hpp:32-36). */
gar_hip_solver *gar_hip_solver_create_dense(int device, int horizon, const int32_t *dims5,
                                            int nc0, int batch);
void gar_hip_solver_destroy(gar_hip_solver *s);

/* Launch every kernel of this solver on `hip_stream` (a hipStream_t); NULL
 * selects the solver's own stream. */
int gar_hip_set_stream(gar_hip_solver *s, void *hip_stream);
int gar_hip_sync(gar_hip_solver *s);

/* ---- sizes (the caller's records: laid out by the caller's dimensions, csrc/gar_layout.h) ------------------ */
int64_t gar_hip_problem_doubles(const gar_hip_solver *s);  /* one packed problem (gar_hip_upload_packed)  */
int64_t gar_hip_factors_doubles(const gar_hip_solver *s);  /* one problem's factor records ON THE DEVICE */
int64_t gar_hip_solution_doubles(const gar_hip_solver *s); /* one problem's xs|us|vs|lbdas */
int gar_hip_batch(const gar_hip_solver *s);
int gar_hip_horizon(const gar_hip_solver *s);
/* which kernel family serves this solver: "generic", "dense", or a specialised family such as "wave<36,12>",
 * "mfma<36,12>", "wave<36,12,32>", "pair<56,24>", "wave_leg<12,8>" -- under padding the PADDED family */
const char *gar_hip_kernel_name(const gar_hip_solver *s);
/* How many legs to ask for when ONE problem is to be solved in parallel-in-time mode on one device (the reference has
 * no counterpart: its caller passes num_threads, parallel-solver.hxx:23-28).  A measured table, not a model: the Newton
 * iteration through gar_hip_backward_blocks at N = 256 on one MI355X (profiles/r06_seam_leg_counts.log) is fastest
 * with horizon / 4 legs on the families of one wave per leg (nx <= 36: 721-729 us with 64 legs, 736-740 with 32,
 * 807-812 with 16) and with horizon / 8 on the wider ones ((56, 22): 1 909-1 942 us with 32, 1 916-1 924 with 64,
 * 2 067 with 128): a leg is a sequential chain over its stages, a level of the condensed solve costs two to three
 * stages.  Returns a value in [2, horizon + 1] (every leg needs a stage); 1 for horizon < 1. */
int gar_hip_suggest_num_legs(int horizon, int nx, int nu);
/* offsets (doubles) of stage t inside one packed problem / one solution record of the caller:
 * out[0]=knot record, out[1]=factor record (device), out[2..5]= x,u,v,lbda offsets */
int gar_hip_stage_offsets(const gar_hip_solver *s, int t, int64_t out[6]);
/* offset (doubles) of G0 and g0 inside one packed problem */
int gar_hip_init_offsets(const gar_hip_solver *s, int64_t out[2]);

/* ---- problem upload (host -> HBM) ---------------------------------------- */
/* One knot of problem b from LqrKnotTpl's separately allocated blocks.  Any of
 * the parametric pointers may be NULL when nth == 0; C,D,d may be NULL when
 * nc == 0. */
int gar_hip_upload_stage(gar_hip_solver *s, int b, int t, const double *Q,
                         const double *S, const double *R, const double *q,
                         const double *r, const double *A, const double *B,
                         const double *f, const double *C, const double *D,
                         const double *d, const double *Gth, const double *Gx,
                         const double *Gu, const double *Gv, const double *gamma);
int gar_hip_set_init(gar_hip_solver *s, int b, const double *G0, const double *g0);
/* nb already-packed problems (gar_hip_problem_doubles() each, the caller's dimensions) starting at b0 */
int gar_hip_upload_packed(gar_hip_solver *s, int b0, int nb, const double *packed);
/* same, but `packed_dev` is a DEVICE pointer to DEVICE records (gar_hip_device_sizes()[0] doubles each) */
int gar_hip_upload_packed_device(gar_hip_solver *s, int b0, int nb, const double *packed_dev);
/* Flush the pinned staging area filled by upload_stage/set_init to the GPU.
 * backward() calls it implicitly when staging is dirty. */
int gar_hip_commit(gar_hip_solver *s);
/* Device pointer to the packed problems (batch x problem_doubles) so that
 * device-resident producers (the updateLQSubproblem replacement, SURVEY 8f1, or
 * a synthetic generator) can write knots in place. */
double *gar_hip_device_problems(gar_hip_solver *s);
/* (factors: the solver's own device format -- csrc/gar_layout.h.  The specialised families keep fb / fth in the fbT2
 * order and, the serial one-wave family, the symmetric Vxx as its packed lower triangle (gar_sym_index): read them
 * through gar_hip_get_gains / gar_hip_get_value / gar_hip_fetch_results, which convert.) */
double *gar_hip_device_factors(gar_hip_solver *s);
double *gar_hip_device_solutions(gar_hip_solver *s);
/* The DEVICE side of the records behind those pointers (and behind gar_hip_upload_packed_device): identical to the
 * caller's unless the solver is padded.
 *   gar_hip_device_stage_layout  out[0..4] = nx,nu,nc,nx2,nth of stage t's device records, out[5] = knot record
 *                                offset, out[6] = factor record offset, out[7..10] = x,u,v,lbda offsets
 *   gar_hip_device_sizes         out[0..2] = doubles of one device problem / factor set / solution record,
 *                                out[3] = rows of the device G0 (nc0 + dummy states), out[4..5] = offsets of G0, g0,
 *                                out[6] = 1 if the solver is padded, 0 otherwise (a boolean),
 *                                out[7] = doubles of one initial-stage record
 *   gar_hip_device_record_format the FORMAT of the device records, bit flags (csrc/gar_layout.h):
 *       GAR_HIP_FMT_QR_PACKED   the knots t < horizon keep Q and R as their LOWER TRIANGLES, packed column after
 *                               column (LAPACK "L" order) in the first n (n + 1) / 2 doubles of the Q / R block, the
 *                               rest of the block unused (the headline one-wave sweep reads 19 % fewer bytes per
 *                               knot).  Every host entry point converts; an in-place DEVICE producer must write
 *                               this format -- and says which it wrote through gar_hip_upload_packed_device_fmt,
 *                               which refuses a mismatch instead of sweeping misread blocks.  The format follows
 *                               the kernel family (batch vs. number of CUs, GAR_HIP_BACKWARD): ask, do not assume.
 *       GAR_HIP_FMT_VXX_PACKED  factor records: Vxx as its packed lower triangle (gar_sym_index)
 *       GAR_HIP_FMT_FB_T2       factor records: fb / fth in the fbT2 order */
/* The dimensions (nx, nu, nc, nx2, nth) of stage t in the CALLER-FACING packed records (gar_hip_upload_packed /
 * gar_hip_download_packed; block order of csrc/gar_layout.h): the caller's own, with one exception -- a terminal knot
 * given with nx2 = 0, as SolverProxDDP builds it (solvers/proxddp/workspace.hxx:54-55), is kept as nx2 = nx (the
 * uniform record every kernel family addresses; its A and f, which nothing of the algorithm reads, are zeros;
 * gar_hip_upload_stage ignores the two pointers for that knot and gar_hip_get_gains hands back the caller's rows). */
int gar_hip_packed_stage_dims(const gar_hip_solver *s, int t, int32_t out[5]);
#define GAR_HIP_FMT_QR_PACKED 1
#define GAR_HIP_FMT_VXX_PACKED 2
#define GAR_HIP_FMT_FB_T2 4
int gar_hip_device_stage_layout(const gar_hip_solver *s, int t, int64_t out[11]);
int gar_hip_device_sizes(const gar_hip_solver *s, int64_t out[8]);
int gar_hip_device_record_format(const gar_hip_solver *s);
/* gar_hip_upload_packed_device with the producer's statement of the format it wrote (GAR_HIP_FMT_* flags; only
 * GAR_HIP_FMT_QR_PACKED concerns knot records): GAR_HIP_ERR_ARG when it is not the solver's. */
int gar_hip_upload_packed_device_fmt(gar_hip_solver *s, int b0, int nb, const double *packed_dev, int record_format);

/* nb packed problems (gar_hip_problem_doubles() each) back to the host (diagnostics, tests) */
int gar_hip_download_packed(gar_hip_solver *s, int b0, int nb, double *packed);

/* ---- device-resident LQ assembly (next row either side of the path) ----------- */
/* Replaces SolverProxDDPTpl::updateLQSubproblem (solvers/proxddp/solver-proxddp.hxx:734-805):
 * `deriv_dev` is a DEVICE buffer of batch x gar_hip_deriv_doubles() doubles holding, per problem,
 * the header G0 | g0 | init Hxx and one derivative record per stage (csrc/gar_layout.h:
 * Lxx Lxu Luu Lx Lu Jx Ju slack Cx Cu Lv | Hxx Hxu Huu | lx_corr lu_corr); the kernel writes the
 * knot records in place (Q,R get `preg` on the diagonal, the dynamics Hessians are added when
 * hess_exact != 0, stage 0 gets the initial condition's Hessian).  Asynchronous on the solver's
 * stream: follow with gar_hip_backward_async.  out[0] of gar_hip_deriv_offsets = offset of stage
 * t's record, out[1..3] = offsets of G0, g0, init Hxx inside one problem's derivative buffer.
 * The derivative records ALWAYS speak the caller's dimensions: on a padded solver ((56, 22) -> (56, 24), (12, 6) ->
 * (12, 8), (4, 2) -> (8, 4), ...) the kernel scatters the real rows / columns into the padded knots and writes the
 * dummy ones (R = I, S = 0, B = 0, r = 0; Q = I, A = 0, f = 0; [0 -I] x0 = 0), as gar_hip_upload_stage does.
 * Not available on a multi-device solver (GAR_HIP_ERR_UNSUPPORTED). */
int64_t gar_hip_deriv_doubles(const gar_hip_solver *s);
int gar_hip_deriv_offsets(const gar_hip_solver *s, int t, int64_t out[4]);
int gar_hip_update_lq_subproblem_device(gar_hip_solver *s, const double *deriv_dev, double preg,
                                        int hess_exact);

/* ---- the sweep ------------------------------------------------------------ */
/* backward(mueq): returns 0, or GAR_HIP_ERR_FACTOR if any stage factorisation
 * of any problem failed (the reference throws).  mueq = 0 (|mueq| < 1e-290, NaN) on a problem with a knot whose
 * solve divides by it -- a constrained knot without controls, Z = C / mu (riccati-kernel.hxx:146-149); the
 * specialised constrained families -- is GAR_HIP_ERR_FACTOR before anything is launched (the reference: infinities,
 * or its "failed stage" exception on the singular [Rhat 0; 0 0]). */
int gar_hip_backward(gar_hip_solver *s, double mueq);
int gar_hip_backward_async(gar_hip_solver *s, double mueq);
/* forward: theta (host, ntheta doubles per problem, batch-major) or NULL.
 * Results stay in HBM; fetch with gar_hip_get_solution. */
int gar_hip_forward(gar_hip_solver *s, const double *theta);
int gar_hip_forward_async(gar_hip_solver *s, const double *theta_device);
/* The binding's backward(mueq) in ONE call: the caller's whole problem (batch = 1) -- blocks[16 t + k], k = 0..15 in
 * the argument order of gar_hip_upload_stage (Q S R q r A B f C D d Gth Gx Gu Gv gamma; same NULL rules), G0, g0 --
 * re-read as the reference does on every backward (proximal-riccati.hxx:37), then the sweep: gar_hip_upload_stage x
 * (N+1) + gar_hip_set_init + gar_hip_backward behind one crossing of the ABI.
 * On a problem without parameter (nth = 0, or leg mode, where theta has no say) on one device the roll-out, the
 * solution's copy into the pinned result buffer (by the kernel's own stores: the copy engine carries the gains) and
 * the gains' read-back (in leg mode already behind the leg sweeps, under the condensed solve) are enqueued right
 * behind the sweep, BEFORE the host waits for the status word: the device
 * runs them back to back, and the gar_hip_forward / gar_hip_prefetch_gains / gar_hip_fetch_results(solution) calls
 * that follow find their work done (same results; the next backward, gar_hip_collapse_feedback or
 * gar_hip_cycle_append ends that state).  GAR_HIP_EAGER=0 switches it off. */
int gar_hip_backward_blocks(gar_hip_solver *s, const double *const *blocks, const double *G0, const double *g0,
                            double mueq);
/* number of problems whose backward reported a failed factorisation */
int gar_hip_num_failed(gar_hip_solver *s);

/* Diagnostics of the last backward (specialised kernel families; no reference counterpart): the
 * register LDL^T of Rhat is what Bunch-Kaufman does whenever its first test |a_kk| >= alpha*colmax
 * holds at every column (bunchkaufman.hpp:61); out[0] = stages (summed over the batch) where it did
 * not and the complete rule was evaluated, out[1] = those of them where Bunch-Kaufman really
 * interchanges or takes a 2x2 pivot (generic device Bunch-Kaufman, exactly the reference's). */
int gar_hip_slow_path_stages(gar_hip_solver *s, int64_t out[2]);
/* Constrained wave kernels (nc > 0 on every knot): a knot with D = 0 -- the reference's own generator,
 * tests/gar/test_util.cpp:42-43 -- on whose Rhat Bunch-Kaufman keeps the natural order is the
 * unconstrained stage plus [zff | Z] = [d | C] / mu (riccati-kernel.hxx:232-262 on a block-diagonal KKT
 * matrix).  Of the last backward, summed over the batch: out[0] = stages run as the coupled stage (register
 * LDL^T of the (nu+nc) x (nu+nc) reduced KKT matrix, Bunch-Kaufman keeping its natural order), out[1] =
 * stages run with the LDS Bunch-Kaufman (interchanges / 2x2 pivots).  A problem stays on the kernel it
 * reached until that one meets a knot it does not serve. */
int gar_hip_constrained_bk_stages(gar_hip_solver *s, int64_t out[2]);

/* ---- horizon sharding (leg mode, one rank per GPU) ------------------------ */
/* doubles per leg in the boundary tuple (Vxx | Vxt | Vtt | vx | vt of the leg's
 * first stage): 3 nx^2 + 2 nx */
int64_t gar_hip_boundary_doubles(const gar_hip_solver *s);
/* device buffer holding this rank's tuples: [batch][local legs][tuple] */
double *gar_hip_device_boundary_local(gar_hip_solver *s);
/* device buffer the condensed solve reads: [num_ranks chunks as gathered]; on a
 * single-GPU solver it aliases the local buffer */
double *gar_hip_device_boundary_all(gar_hip_solver *s);
int gar_hip_backward_legs_async(gar_hip_solver *s, double mueq);   /* parallel-solver.hxx:150-164 */
int gar_hip_condensed_solve_async(gar_hip_solver *s);              /* :169-202 */
int gar_hip_forward_legs_async(gar_hip_solver *s);                 /* :209-243 */
/* refinement controls (parallel-solver.hpp:92-94) */
int gar_hip_set_refinement(gar_hip_solver *s, double condensed_threshold, int max_steps);
/* outcome of the last condensed solve of problem b: out[0] = infinity norm of the last residual
 * evaluated (the quantity parallel-solver.hxx:191 tests), out[1] = refinement steps taken */
int gar_hip_condensed_info(gar_hip_solver *s, int b, double out[2]);
/* Block cyclic reduction of the condensed system (specialised leg families): the backward error
 * omega = max_i |r_i| / max_i (|rhs_i| + sum_j |K_ij| |s_j|) of its solution.  The solve stands when the reference's
 * absolute residual threshold is met OR omega <= 1e-13: with value functions of order 1/mu (constrained knots) the
 * absolute threshold (parallel-solver.hpp:92) is out of reach of any fp64 solver -- the reference then spends its
 * maxRefinementSteps without effect (parallel-solver.hxx:184-202).  0 when the elimination chain solved instead. */
int gar_hip_condensed_backward_error(gar_hip_solver *s, int b, double *out);
/* The condensed system is first solved by a fast elimination order -- block cyclic reduction (specialised leg
 * families) or, on the any-dimension path, with the leg states eliminated leg-parallel and the chain run on the
 * J remaining blocks -- and checked by its residual; *out = 1 when the last solve of problem b missed the check and
 * was redone in the reference's order (block-tridiagonal.hpp:82-138 with refinement), 0 when the fast result stood. */
int gar_hip_condensed_resolved(gar_hip_solver *s, int b, int *out);
/* which fast solver the condensed system of this (leg-mode) solver goes through before the gated chain:
 *   "cyclic"         block cyclic reduction, one wave per block (specialised leg families);
 *   "chain"          the wave-scope elimination chain in the reference's order (GAR_HIP_CONDENSED=chain);
 *   "reduced+cyclic" any-dimension path: leg states eliminated leg-parallel, the J remaining blocks by block cyclic
 *                    reduction on a workgroup per block and level (log2 J dependent steps; from 4 legs on;
 *                    GAR_HIP_CONDENSED_CR=0 switches it off, =<k> moves the threshold to k legs);
 *   "reduced+chain"  ... the J remaining blocks by the one-workgroup chain;
 *   "generic-chain"  the 2 J blocks in the reference's order (block-tridiagonal.hpp:82-138) only.
 * "" on a solver without legs. */
const char *gar_hip_condensed_solver_name(const gar_hip_solver *s);
/* the omega bound above (default 1e-13; 1e-12 until round 4: a soak draw with omega = 9.4e-13 kept multipliers 150 x
 * farther from LAPACK than any CPU solver, the forward error being cond * omega); 0: only the reference's absolute threshold counts */
int gar_hip_set_condensed_backward_ok(gar_hip_solver *s, double omega);

/* ---- results (HBM -> host) ------------------------------------------------ */
/* packed solution of problem b: xs | us | vs | lbdas (per-stage offsets from
 * gar_hip_stage_offsets); any pointer may be NULL */
int gar_hip_get_solution(gar_hip_solver *s, int b, double *xs, double *us,
                         double *vs, double *lbdas);
/* ff (nu+nc+nx2), fb ((nu+nc+nx2) x nx ROW-major), fth (.. x nth ROW-major) */
int gar_hip_get_gains(gar_hip_solver *s, int b, int t, double *ff, double *fb,
                      double *fth);
/* Bulk read-back for the caller's per-iteration loop (SolverProxDDPTpl::computeDirection copies
 * getFeedforward(i) / getFeedback(i) of EVERY stage, solver-proxddp.hxx:620-632): the gains of all
 * stages of problem b as two dense arrays, ff_all = [ff_0 | ff_1 | ...] and fb_all = [fb_0 | fb_1 |
 * ...], every fb_t ROW-major like StageFactor::fb.  One device-side gather (the specialised kernel
 * families keep fb in a device order), ONE device-to-host copy, ONE synchronisation -- instead of
 * 2-3 copies and a synchronisation per stage through gar_hip_get_gains.
 *   gar_hip_gains_doubles   out[0] = doubles in ff_all, out[1] = doubles in fb_all
 *   gar_hip_gains_offsets   out[0], out[1] = offset of stage t inside ff_all / fb_all
 *   gar_hip_fetch_results   what = 1: the solution xs|us|vs|lbdas, 2: the gains, 3: both -- into a
 *                           pinned host buffer the library owns; synchronous on return
 *   gar_hip_host_results    that buffer; offs[0..2] = offsets (doubles) of the solution record, of
 *                           ff_all and of fb_all inside it.  The subclass maps its getFeedforward /
 *                           getFeedback views straight onto it ("views into solver-owned host
 *                           memory, valid until the next backward", as in the reference).
 *   gar_hip_get_gains_all   fetch + copy into caller-owned arrays (either may be NULL)
 *   gar_hip_prefetch_gains  optional, right after gar_hip_backward: starts the gather and the copy of problem b's
 *                           gains on a second stream, so that they overlap the forward sweep and the solution
 *                           read-back; the next gar_hip_fetch_results(s, b, 2) then only waits for them (and fetches
 *                           stage 0 again if gar_hip_collapse_feedback ran in between).  The host buffer must not be
 *                           read for gains before that fetch.  A no-op on multi-device and folded solvers. */
int gar_hip_prefetch_gains(gar_hip_solver *s, int b);
int gar_hip_gains_doubles(const gar_hip_solver *s, int64_t out[2]);
int gar_hip_gains_offsets(const gar_hip_solver *s, int t, int64_t out[2]);
int gar_hip_fetch_results(gar_hip_solver *s, int b, int what);
const double *gar_hip_host_results(gar_hip_solver *s, int64_t offs[3]);
int gar_hip_get_gains_all(gar_hip_solver *s, int b, double *ff_all, double *fb_all);
/* StageFactor::kktMat of stage t (gar/riccati-kernel.hpp:30-102; Python: datas[t].kktMat,
 * bindings/python/src/gar/expose-prox-riccati.cpp:30-31): the reduced KKT matrix [Rhat D^T; D -mu I],
 * (nu+nc) x (nu+nc) column-major, Rhat = R + B^T Vxx' B.  The sweeps do not keep it; it is formed on the
 * device, on request, from the knot and stage t+1's Vxx as the last backward left them (pass that sweep's
 * mueq).  Not available on the stage-dense solver. */
int gar_hip_get_kkt(gar_hip_solver *s, int b, int t, double mueq, double *out);
int gar_hip_get_value(gar_hip_solver *s, int b, int t, double *Vxx, double *vx,
                      double *Vxt, double *Vtt, double *vt);
/* kkt0.ff (nx0+nc0), kkt0.fth ((nx0+nc0) x nth ROW-major), thGrad, thHess */
int gar_hip_get_initial(gar_hip_solver *s, int b, double *kkt0_ff,
                        double *kkt0_fth, double *thGrad, double *thHess);
/* collapseFeedback (riccati-base.hpp:33; no-op except in leg mode): enqueued on the solver's stream, not waited
 * for -- every getter and fetch of this library is ordered behind it; a caller reading gar_hip_device_factors() from
 * another stream synchronises with gar_hip_sync first. */
int gar_hip_collapse_feedback(gar_hip_solver *s);
/* Debug aid (no reference counterpart): with enable != 0 the specialised backward
 * kernel stamps s_memtime at its phase boundaries for one stage of problem 0,
 * 16 marks per wave; `out` (may be NULL) receives the last 4 x 16 stamps. */
int gar_hip_debug_trace(gar_hip_solver *s, int enable, long long out[64]);
/* Behaviour switches.  The reference exposes its knobs as struct fields (parallel-solver.hpp:92-94: those two have
 * entry points of their own here, gar_hip_set_refinement / gar_hip_set_condensed_backward_ok); what chooses between
 * this library's kernel families and measured alternatives is a set of named switches, each of which can be given in
 * the ENVIRONMENT (GAR_HIP_<NAME>) or through this call, which takes precedence (process-wide; value NULL: back to the
 * environment; `name` with or without the GAR_HIP_ prefix).  Read when a solver is CREATED unless noted (a change
 * does not reach solvers that exist already); thread-safe.
 *   BACKWARD = wave | wg4 | pair   serial unconstrained family: one wave per problem (default when batch > #CUs),
 *                                  one 4-wave workgroup per problem (default otherwise), two waves per problem
 *   WIDE = single | generic-forward   the (56, 24) family: one wave per problem / the any-dimension roll-out
 *   FORCE_GENERIC = 1              the any-dimension kernels whatever the shape
 *   PAD = 0                        never pad a shape onto a specialised family
 *   INIT = bk                      always factorise kkt0 (no closed form for G0 = +-I)
 *   SPD_ACCEPT = 0                 (per launch) the reference's pivot rule literally in the headline stage: no
 *                                  acceptance of an unpivoted positive definite R-hat (deviation 7)
 *   LEGS = generic | LEG_WAVES = 1 | SEG_LEGS = 0 | FOLD = 0     leg mode: the any-dimension leg kernels / one wave
 *                                  per leg / no segment legs on the wide shape / constrained knots not folded
 *   CSTR_SEG_LEGS = 0              leg mode, problems with D != 0 on the any-dimension leg kernels instead of the
 *                                  constrained segment legs (csrc/gar_cstr_seg.hpp; (36,12,32), (16,8,8), (8,4,4));
 *                                  CSTR_SEG_LEG_END = 0 (per launch): their leg ends through the stage chain (two rounds)
 *                                  instead of the leg-end kernel; CSTR_SEG_FORWARD = generic: the any-dimension roll-out
 *   CONDENSED = generic | chain, CONDENSED_REDUCED = 0, CONDENSED_CR = 0 | <k>   the condensed solve: elimination
 *                                  chain instead of block cyclic reduction (specialised / any-dimension paths)
 *   STAGE_NT = 0 | 1, EAGER = 0    host staging with / without non-temporal stores; gar_hip_backward_blocks without
 *                                  the roll-out enqueued behind the sweep (per call)
 *   MULTI_EXCHANGE = copy          gar_hip_multi_create: hipMemcpyPeerAsync instead of the peer-mapped gather
 *   PIPELINE = auto | 0 | 2        the schedule a NEW solver starts with (default auto: gar_hip_set_pipeline(s, -1))
 *   PIPE_PRIORITY = 0 | 1          gar_hip_set_pipeline: plain half streams / one at high priority
 *   FORWARD = lean                 (per launch) the LDS-DMA roll-out of the pipelined schedule in the plain one too
 * GAR_HIP_ERR_ARG for a name that is none of these. */
int gar_hip_set_option(const char *name, const char *value);
const char *gar_hip_get_option(const char *name);
/* The pipelined sweep (no reference counterpart: the reference's batch axis is a caller's OpenMP loop over
 * solvers, bench/gar-riccati.cpp:42-51).  halves = 2 cuts the batch in two halves with a stream each, owned by the
 * library: gar_hip_backward_async / gar_hip_forward_async then enqueue on those streams, the backward sweeps of
 * the two halves alternating and the forward sweep of a half running BESIDE the backward sweep of the other half
 * -- of this call pair or of the next one -- on every SIMD (csrc/gar_forward_lean.hpp).  Same results, bit for
 * bit.  Ordering contract: work the caller enqueued on the solver's stream before the call is waited for; EVERY
 * other entry point of this library (gar_hip_sync, the getters, uploads, gar_hip_device_problems / _solutions, ...)
 * first orders the solver's stream behind the half streams, so code that touches the records only through
 * this header needs no change; a caller that enqueues its own kernels on the records calls gar_hip_sync (or any
 * getter) first.  halves = 0 / 1: off.  GAR_HIP_ERR_UNSUPPORTED unless the solver runs the serial
 * one-wave-per-problem family (batch > number of CUs, nc = nth = 0, at least 2 problems).
 * halves = -1: the LIBRARY chooses (never an error) -- two halves iff the solver is eligible and each half fills every
 * SIMD of the device with a backward wave (batch >= 8 x #CUs: 2 048 on an MI355X), plain otherwise.  This is what a
 * new solver starts with (round 6; switch PIPELINE above), and what a cycleAppend that rebuilds the solver for other
 * dimensions re-evaluates; gar_hip_pipeline tells which schedule is in force. */
int gar_hip_set_pipeline(gar_hip_solver *s, int halves);
int gar_hip_pipeline(const gar_hip_solver *s);
/* Measurement aid (no reference counterpart): with enable != 0 the library brackets the
 * backward sweep kernel, the initial-stage kernel and the forward sweep kernel with HIP events
 * on the launch stream; gar_hip_last_kernel_ms returns the durations (ms) of the last
 * backward/forward pair as out[0..2].  Specialised kernel family only.  Pipelined sweeps: per HALF-batch
 * launch (mean of the two halves), out[1] = 0 (the initial stage rides inside the sweep). */
int gar_hip_set_timing(gar_hip_solver *s, int enable);
int gar_hip_last_kernel_ms(gar_hip_solver *s, double out[3]);
/* MPC cycling: drop knot 0, shift left, last-but-one knot gets dims5_new.
 * Uniform serial problems (every stage of the same dimensions as the new knot): a RING -- no record
 * of the problem or of the factors is copied, the stream is not synchronised; gar_hip_stage_offsets
 * reports where each (logical) stage now lives.  Otherwise (dimensions change, leg mode: "just
 * reinitialise everything", parallel-solver.hxx:246-258) the layout and the buffers are rebuilt:
 * the new configuration (padding, layout, LDS plan, kernel family) is validated on a trial object first -- on
 * error the solver is unchanged, ring position included -- device pointers fetched earlier become invalid,
 * resident problem data is not carried over.  dims5_new: the caller's dimensions of the new knot. */
int gar_hip_cycle_append(gar_hip_solver *s, const int32_t dims5_new[5]);

#ifdef __cplusplus
}
#endif
#endif
