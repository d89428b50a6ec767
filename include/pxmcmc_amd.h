/*
 * pxmcmc_amd.h -- C-ABI of the MI355X (gfx950) implementation of pxmcmc's
 * proximal-Langevin hot path (MYULA / PxMALA iteration).
 *
 * The reference (auggiemarignier/pxmcmc v1.0.1) has no FFI of its own: its hot
 * path is a duck-typed Python protocol whose O(L^3) work is
 * done by the pyssht / pys2let wheels.  These entry points are what a binding
 * for that path replaces; each cites the reference interface (file:line in
 * /root/reference) it stands in for.  Conventions:
 *
 *   - every function returns 0 on success, <0 on error (pxm_last_error() gives text);
 *   - device buffers are caller-owned; nothing is allocated after plan creation;
 *   - all device work is enqueued on the caller-supplied HIP stream (hipStream_t
 *     passed as void*); plans are not shared across threads or devices;
 *   - all mutable state lives in a plan (workspace, carried rings, Philox iteration counter, profiler); the
 *     only process-wide objects are the read-only per-device table cache (pxm_tables_trim), the borrowed side
 *     streams and the graveyard of deferred frees below;
 *   - arrays carry a leading chain-batch dimension C (independent chains); inside
 *     a chain the reference's own 1-D orders are kept: harmonic index el^2+el+m,
 *     MW images theta-major (L, 2L-1) C-order, wavelet coefficient vectors
 *     [scaling | j=J_min | ... | j=J_max] each block theta-major
 *     (pxmcmc/utils.py:11-22,49-51);
 *   - complex128 values are interleaved (re, im) doubles.
 */
#ifndef PXMCMC_AMD_H
#define PXMCMC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pxm_sht_plan_s* pxm_sht_plan_t;
typedef struct pxm_wav_plan_s* pxm_wav_plan_t;
typedef void* pxm_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------ */
int pxm_version(void);
/* Precision of the Box-Muller step of the Philox noise stream (pxmcmc/mcmc.py:193-195 draws fp64 randn).  Every
 * entry point that can draw noise takes the flag PXM_NOISE_F64, OR-ed into its `mode` (pxm_wav_*_step), `noise_complex`
 * (pxm_myula_step*, pxm_chain_step*, pxm_pxmala_propose) or `dtype` (pxm_randn) argument:
 *   absent   Box-Muller on the f32 transcendental units (v_log_f32 / v_sin_f32 / v_cos_f32 with the exact fp64 exponent:
 *            deviates ~1e-6 relative, tail to 8.5 sigma) -- the default, pxm_noise_bits() == 32;
 *   present  log / sqrt / sincos evaluated in double precision (branch-free polynomials, csrc/philox.h): deviates equal
 *            numpy's float64 evaluation of the same formulae to ~1e-15.
 * Both read the same Philox4x32-10 counter stream and the same uniforms: the two streams agree to ~1e-6. */
#ifdef PXM_NOISE_F64
#error "the -DPXM_NOISE_F64 build switch was removed: pass the flag PXM_NOISE_F64 per call (one library serves both precisions)"
#endif
#define PXM_NOISE_F64 16
int pxm_noise_bits(void);
/* Host-only (no GPU) check that every global address a launch of the plans' GEMM task lists and DFT groups can form
 * -- including the clamped / aliased loads whose values are discarded -- lies inside its buffer.  Builds the plans
 * named by `what` (1: SHT plan (L, spin); 2: wavelet plan (L, B, J_min) + Gram lists; 4: + weak-lensing lists) in
 * dry-run mode and returns the number of address ranges verified, < 0 on a violation (pxm_last_error names the task).
 * The same check runs at every real plan creation.  The dry-run switch is per thread (plans created concurrently on
 * other threads are real); not to be called during a stream capture. */
int64_t pxm_host_check_address_ranges(int L, double B, int J_min, int spin, int max_chains, int what);
const char* pxm_last_error(void);
/* number of visible HIP devices (0 when none; never fails) */
int pxm_device_count(void);

/* ---- teardown during stream capture ---------------------------------------------------------
 * hipFree is illegal while a stream capture is in progress, and the host language may tear a plan down at any
 * moment (Python's garbage collector).  The destroy calls therefore never free directly: device memory goes to
 * a graveyard that is emptied at safe points (plan creation, pxm_capture_end, a destroy outside any capture).
 * Bracket a capture with pxm_capture_begin / pxm_capture_end; a capture started elsewhere is also recognised
 * as soon as one entry point has been called on its stream.  pxm_deferred_pending: entries still queued. */
int pxm_capture_begin(void);
int pxm_capture_end(void);
int pxm_deferred_pending(void);
/* The Wigner ring tables are cached per device and shared by plans; this frees every cache entry no live plan
 * holds (returns the MiB released). */
int pxm_tables_trim(void);

/* ---- host-side setup helpers (no GPU needed) ------------------------------ */
/* pys2let.pys2let_j_max(B, L, J_min)            (pxmcmc/transforms.py:75) */
int pxm_j_max(int L, double B);
/* multiresolution bandlimits [scaling, j=J_min..J_max] (pxmcmc/utils.py:116-125);
 * writes at most cap entries, returns the count or <0 */
int pxm_wav_bandlimits(int L, double B, int J_min, int* bl_out, int cap);
/* number of wavelet+scaling coefficients  (pxmcmc/transforms.py:156-166) */
int64_t pxm_wav_ncoefs(int L, double B, int J_min, int64_t* nscal_out);
/* axisymmetric tiling: kappa0[L], kappa[(J_max+1)*L]  (pys2let.wavelet_tiling,
 * pxmcmc/utils.py:117; prior.py:121,132) */
int pxm_tiling_axisym(int L, double B, int J_min, double* kappa0, double* kappa);
/* MW quadrature weight per ring q[L] (pxmcmc/utils.py:262-283: mw_map_weights = outer(q, 1)) */
int pxm_mw_ring_weights(int L, double* q);
/* dense per-m ring tables for tests (small L): Binv[t*L+el] = (-1)^s N_el d^el_{m,-s}(theta_t),
 * Afwd[el*L+t] = exact-quadrature forward matrix; either pointer may be NULL */
int pxm_host_sht_tables(int L, int spin, int m, double* Binv, double* Afwd);
/* the same Binv[t*L+el] as the table-free recursion kernels generate it (csrc/rec_core.h: three-term recursion in el in
 * double precision on a per-ring scaled state; host emulation, operation for operation): what pyssht.inverse /
 * inverse_adjoint (pxmcmc/measurements.py:225,237) multiply by when a plan takes the recursion path */
int pxm_host_rec_table(int L, int spin, int m, double* Brec);
/* tables of the exact-length phi-DFT unit for ring length 511 = 7 x 73 (csrc/dft_pfa.h; the phi stage of
 * pys2let.synthesis_wav2px / synthesis_adjoint_px2wav at bandlimit 256, pxmcmc/transforms.py:126,138): idx[(64 + 80) * 8]
 * uint16 gather / scatter offsets, b2[8 * 9 * 2] the spectrum of Rader's filter; for the CPU test against
 * scripts/dev/proto_pfa511.py */
int pxm_host_pfa511_tables(uint16_t* idx, double* b2);

/* ---- spin spherical-harmonic transforms on the MW grid ---------------------- */
/* replaces pyssht.forward / inverse / inverse_adjoint / forward_adjoint
 * (pxmcmc/measurements.py:223,225,237,239).  flm: [C][L*L] c128, f: [C][L*(2L-1)] c128. */
int pxm_sht_plan_create(int L, int spin, int max_chains, unsigned flags, pxm_sht_plan_t* plan);
int pxm_sht_plan_destroy(pxm_sht_plan_t plan);
/* Non-zero when the plan's inverse / inverse_adjoint (pyssht.inverse / inverse_adjoint, pxmcmc/measurements.py:225,237) run the
 * table-free ring stage -- Wigner rows by three-term recursion in el on the vector pipe, csrc/sht_rec.hip -- instead of the
 * ring-table GEMM: 16 * (ring blocks per wavefront) + (complex columns per order).  Chosen at plan creation: few-column
 * plans at large L; PXM_REC=1 / 0 forces it on (where the column count allows) / off. */
int pxm_sht_uses_recursion(pxm_sht_plan_t plan);
/* self-test of the wavefront transpose-reduce of the ring -> el recursion kernel: out128[lane] = the sum the lane holds for
 * 16 known per-lane values, out128[64 + lane] = the id of the value it claims to hold (test aid) */
int pxm_rec_reduce_selftest(double* out128);
int pxm_sht_inverse(pxm_sht_plan_t plan, const void* flm, void* f, int C, pxm_stream_t stream);
int pxm_sht_forward(pxm_sht_plan_t plan, const void* f, void* flm, int C, pxm_stream_t stream);
int pxm_sht_inverse_adjoint(pxm_sht_plan_t plan, const void* f, void* flm, int C, pxm_stream_t stream);
int pxm_sht_forward_adjoint(pxm_sht_plan_t plan, const void* flm, void* f, int C, pxm_stream_t stream);
/* bytes of Legendre/Wigner ring table one transform launch streams (roofline accounting) */
int64_t pxm_sht_table_bytes(pxm_sht_plan_t plan, int op /*0 inv,1 fwd,2 inv_adj,3 fwd_adj*/);

/* ---- scale-discretised wavelet transform (N=1, spin 0, upsample=0) -------------- */
/* replaces pys2let.synthesis_wav2px / synthesis_adjoint_px2wav / analysis_px2wav /
 * analysis_adjoint_wav2px (pxmcmc/transforms.py:95-98).  X: [C][ncoefs] c128, f: [C][L*(2L-1)] c128. */
int pxm_wav_plan_create(int L, double B, int J_min, int max_chains, unsigned flags, pxm_wav_plan_t* plan);
int pxm_wav_plan_destroy(pxm_wav_plan_t plan);
int pxm_wav_synthesis(pxm_wav_plan_t plan, const void* X, void* f, int C, pxm_stream_t stream);
int pxm_wav_synthesis_adjoint(pxm_wav_plan_t plan, const void* f, void* X, int C, pxm_stream_t stream);
int pxm_wav_analysis(pxm_wav_plan_t plan, const void* f, void* X, int C, pxm_stream_t stream);
int pxm_wav_analysis_adjoint(pxm_wav_plan_t plan, const void* X, void* f, int C, pxm_stream_t stream);
int64_t pxm_wav_table_bytes(pxm_wav_plan_t plan, int op /*0 synthesis,1 synthesis_adjoint*/);

/* Device-resident Philox iteration counter OF ONE PLAN (HIP-graph replay of the MYULA step): when registered,
 * the plan's fused steps use iteration = iter + *counter, read on the device at execution time, so a captured
 * graph draws fresh noise at every replay.  Two plans (two samplers) in one process never share a counter.
 * pxm_wav_iter_counter_add enqueues "*counter += inc" on the stream.  A plan holds ONE live counter: registering
 * a different one while another is live is an error (two stepping engines on one plan would redirect each other's
 * noise stream); NULL unregisters unconditionally, pxm_wav_release_iter_counter only if `counter_dev` is still the
 * registered one (the owner's teardown call: never drops somebody else's counter). */
int pxm_wav_set_iter_counter(pxm_wav_plan_t plan, uint64_t* counter_dev);
int pxm_wav_release_iter_counter(pxm_wav_plan_t plan, const uint64_t* counter_dev);
int pxm_wav_iter_counter_add(pxm_wav_plan_t plan, uint64_t inc, pxm_stream_t stream);
/* Device status of a plan.  Two kernels of this library wait on each other with BOUNDED spins instead of barriers: the
 * wave pairs of the fused phi-DFT kernels (csrc/dft5.hip, d5_pair_sync: an LDS counter per pair) and, with PXM_FLOW=1
 * (experimental; read when a plan's Gram lists are built), the forward-adjoint tasks of the dataflow GEMM launch (per-
 * order counters).  A wait that expires does not hang the GPU -- the kernel runs on with data its partner has not
 * written -- and ORs a bit into the plan's status word; the results of that launch are invalid.  The reference fails
 * loudly on bad state (pxmcmc/mcmc.py:104-109); so does the sampler here: it reads the word wherever it already
 * synchronises (saved samples, progress prints, end of run) and raises.
 *   pxm_wav_status / pxm_sht_status : bit mask since the last clear, 0 = every wait was satisfied; `clear` != 0 resets
 *                                     it after reading.  Synchronises the stream.  < 0: error.
 *   pxm_wav_flow_status             : the PXM_STATUS_FLOW_WAIT bit as 0 / 1 (round-3 interface).
 *   pxm_wav_flow_enabled            : 1 when this plan's ring-space step takes the dataflow launch (known after
 *                                     pxm_wav_ring_set_data), 0 when it runs the two ordinary launches.
 * PXM_DEBUG_PAIR_SYNC_LIMIT=<n> (read at plan creation) sets the bound of the pair wait; 0 forces every wait to
 * expire -- the test of this report path. */
#define PXM_STATUS_FLOW_WAIT 1
#define PXM_STATUS_PAIR_SYNC 2
int pxm_wav_status(pxm_wav_plan_t plan, int clear, pxm_stream_t stream);
int pxm_sht_status(pxm_sht_plan_t plan, int clear, pxm_stream_t stream);
int pxm_wav_flow_status(pxm_wav_plan_t plan, pxm_stream_t stream);
int pxm_wav_flow_enabled(pxm_wav_plan_t plan);
/* scales whose 511-point rings the fused rings -> X' -> rings launch takes through the exact-length phi-DFT unit (0: Bluestein) */
int pxm_wav_exact_dft_scales(pxm_wav_plan_t plan);

/* Live kernel timing of one plan (bench.py roofline leg).  pxm_wav_profile_enable(plan, n) with n > 0 creates
 * n event pairs per kernel class; while enabled every SHT ring-GEMM launch and every grouped phi-DFT launch of
 * this plan is bracketed by a pair on the stream it is launched on (kernel start / stop, as rocprofv3 reports
 * them).  The read calls synchronise the events and return the summed kernel time (ms), the number of
 * launches, the algorithmic bytes moved and the MFMA flops (any pointer may be NULL), then reset.
 * n = 0 disables and releases the events. */
int pxm_wav_profile_enable(pxm_wav_plan_t plan, int max_launches);
int pxm_wav_profile_read(pxm_wav_plan_t plan, double* gemm_ms, int64_t* gemm_launches, double* gemm_alg_bytes,
                         double* gemm_flops);
int pxm_wav_profile_read_dft(pxm_wav_plan_t plan, double* dft_ms, int64_t* dft_launches, double* dft_alg_bytes);
/* per-launch form of pxm_wav_profile_read (instead of it): kernel time (ms), algorithmic bytes and (optional, may be
 * NULL) workgroup count of each of the first `cap` ring-GEMM launches, in launch order -- the workgroup count is the
 * key a rocprofv3 record of the same launch carries; *launches = how many were bracketed; then resets */
int pxm_wav_profile_read_launches(pxm_wav_plan_t plan, double* launch_ms, double* launch_alg_bytes,
                                  int32_t* launch_workgroups, int64_t cap, int64_t* launches);
/* test aid: number of non-finite doubles in the plan's workspace (ring / harmonic arrays incl. the padding
 * chains' columns); synchronises the stream */
int64_t pxm_wav_workspace_nonfinite(pxm_wav_plan_t plan, pxm_stream_t stream);

/* Fused MYULA half-steps (pxmcmc/mcmc.py:158-161 with forward.py:66-72, prior.py:49-50):
 *   pxm_wav_gradg_step: X_out = (1-d/l) X + (d/l) soft(X,T) - d * S^H( invcov .* (preds - data) ) + sqrt(2 d) w
 * i.e. calc_gradg + proxf + chain_step in one pass over the coefficient vector, with the
 * residual folded into the transform's input read and the update into its output write.
 * data/invcov: [P] shared by all chains (invcov complex iff invcov_complex); T: [N] or NULL
 * (then T_scalar); noise: [C][N] injected N(0,1) or NULL for the Philox stream keyed
 * (seed, chain0 + c, iter).  mode (how a complex128 slot of the state is read):
 *   PXM_MODE_REAL_NOISE 0  complex state, real noise (float64 [C][N] when injected): params.complex = False
 *   PXM_MODE_CPLX_NOISE 1  complex state, complex noise (c128 [C][N]):                params.complex = True
 *   PXM_MODE_REAL_PAIRS 2  real data and real state: slot c carries the two REAL chains 2c (real part) and
 *                          2c+1 (imaginary part) through the complex-linear transforms; soft threshold and
 *                          noise are applied per component (injected noise: float64 [2C][N]; Philox keys
 *                          chain0 + 2c and chain0 + 2c + 1); data must then be passed as d + i d.
 * | PXM_NOISE_F64: the Philox stream's Box-Muller step in double precision (see pxm_noise_bits above). */
#define PXM_MODE_REAL_NOISE 0
#define PXM_MODE_CPLX_NOISE 1
#define PXM_MODE_REAL_PAIRS 2
int pxm_wav_gradg_step(pxm_wav_plan_t plan, const void* X, const void* preds, const void* data,
                       const void* invcov, int invcov_complex, const double* T, double T_scalar,
                       double delta, double lmda, const void* noise, int mode,
                       uint64_t seed, uint64_t chain0, uint64_t iter, void* X_out, int C,
                       pxm_stream_t stream);

/* The whole loop body pxmcmc/mcmc.py:158-161 for the identity measurement and a DIAGONAL (per-pixel) inverse
 * covariance: pxm_wav_gradg_step followed by pxm_wav_synthesis of the new state, fused.  The rings of the
 * residual invcov .* (preds - data) (pxmcmc/forward.py:66-69) are carried inside the plan between calls:
 *   pxm_wav_image_init : residual rings of the start state's preds                      (start of a run)
 *   pxm_wav_image_step : X_out = MYULA update of X (as pxm_wav_gradg_step); preds_out = forward(X_out);
 *                        residual rings <- those of preds_out.  Any other call on the plan invalidates the
 *                        carried rings (call pxm_wav_image_init again). */
int pxm_wav_image_init(pxm_wav_plan_t plan, const void* preds, const void* data, const void* invcov,
                       int invcov_complex, int C, pxm_stream_t stream);
int pxm_wav_image_step(pxm_wav_plan_t plan, const void* X, const void* data, const void* invcov,
                       int invcov_complex, const double* T, double T_scalar, double delta, double lmda,
                       const void* noise, int mode, uint64_t seed, uint64_t chain0, uint64_t iter,
                       void* X_out, void* preds_out, int C, pxm_stream_t stream);

/* Ring-space MYULA iteration: identity measurement + UNIFORM inverse covariance w (complex scalar), i.e.
 * ForwardOperator(data, scalar sig_d, "synthesis", SphericalWaveletTransform, Identity).  Between
 * forward() and calc_gradg() the reference forms the image-space residual w (preds - data)
 * (pxmcmc/forward.py:63-72).  On every ring DFT o iDFT = (2L-1) I, so DFT(residual) =
 * w ((2L-1) G - DFT(data)) with G the rings of S X: the L-level iDFT / DFT pair is never executed and
 * the image `preds` is produced only on demand (saved iterations).  Results equal pxm_wav_gradg_step +
 * pxm_wav_synthesis to round-off.  The rings of the current state are carried inside the plan:
 *   pxm_wav_ring_set_data : rings of the data image (once per data set)
 *   pxm_wav_ring_init     : rings <- S X                           (start of a run)
 *   pxm_wav_ring_step     : X_out = MYULA update of X (as pxm_wav_gradg_step); rings <- S X_out.
 *                           The plan's iteration counter, if registered, is advanced by 1 at the START of the step
 *                           (the step's Philox iteration = iter + counter after the increment).
 *   pxm_wav_ring_preds    : preds = forward(X) of the carried state, [C][L(2L-1)] */
int pxm_wav_ring_set_data(pxm_wav_plan_t plan, const void* data, pxm_stream_t stream);
int pxm_wav_ring_init(pxm_wav_plan_t plan, const void* X, int C, pxm_stream_t stream);
int pxm_wav_ring_step(pxm_wav_plan_t plan, const void* X, double w_re, double w_im, const double* T,
                      double T_scalar, double delta, double lmda, const void* noise, int mode,
                      uint64_t seed, uint64_t chain0, uint64_t iter, void* X_out, int C, pxm_stream_t stream);
int pxm_wav_ring_preds(pxm_wav_plan_t plan, void* preds, int C, pxm_stream_t stream);

/* Weak-lensing measurement fused with the wavelet synthesis (BASELINE config 5; pxmcmc/forward.py:63-72 with
 * transform = SphericalWaveletTransform (transforms.py:114-139) and measurement = WeakLensing
 * (measurements.py:209-304)).  Between the synthesis and the measurement the reference runs an inverse SHT and
 * a forward SHT of the same band-limited field at the same bandlimit -- the identity on harmonic coefficients
 * (exact MW quadrature) -- so the harmonic kernel k_l is applied to the synthesised coefficients directly:
 *   pxm_wav_wl_attach  : spin-2 tables; pix2data [L(2L-1)] maps a pixel to its index in the masked data vector
 *                        (< 0 = masked; NULL = no mask, ndata = L(2L-1)); weight [ndata] = WeakLensing.inv_cov or
 *                        NULL.  Both stay caller-owned and must outlive the plan's use of them.
 *   pxm_wav_wl_forward : gamma [C][ndata] = WeakLensing.forward(transform.inverse(X))
 *   pxm_wav_wl_adjoint : X_out [C][ncoefs] = transform.inverse_adjoint(WeakLensing.adjoint(g)), g = gamma, or the
 *                        residual invcov .* (gamma - data) when data / invcov ([ndata]) are given (calc_gradg). */
int pxm_wav_wl_attach(pxm_wav_plan_t plan, const int32_t* pix2data, const double* weight, int64_t ndata);
/* non-zero (as pxm_sht_uses_recursion) when the attached spin-2 stage of pxm_wav_wl_forward / _adjoint -- pyssht.inverse /
 * inverse_adjoint with Spin=2, pxmcmc/measurements.py:225,237 -- runs the table-free recursion kernels */
int pxm_wav_wl_uses_recursion(pxm_wav_plan_t plan);
int pxm_wav_wl_forward(pxm_wav_plan_t plan, const void* X, void* gamma, int C, pxm_stream_t stream);
int pxm_wav_wl_adjoint(pxm_wav_plan_t plan, const void* gamma, const void* data, const void* invcov,
                       int invcov_complex, void* X_out, int C, pxm_stream_t stream);

/* ---- elementwise / reductions ---------------------------------------------------- */
/* dtype: 0 = float64, 1 = complex128.  n = elements per chain. */
/* utils.soft (pxmcmc/utils.py:55-67,84-88): T vector [n] (shared by chains) or NULL -> T_scalar */
int pxm_soft(const void* X, const double* T, double T_scalar, void* out, int64_t n, int C, int dtype,
             pxm_stream_t stream);
/* ForwardOperator._gradg_analysis residual (pxmcmc/forward.py:66-69) with a diagonal invcov:
 * out = invcov .* (preds - data); data, invcov are [n] shared by all chains */
int pxm_residual_grad(const void* preds, const void* data, const void* invcov, int invcov_complex,
                      void* out, int64_t n, int C, int dtype, pxm_stream_t stream);
/* MYULA.chain_step fused with L1 prox (pxmcmc/mcmc.py:185-201 + prior.py:49-50).
 * delta_dev: per-chain step sizes [C] on the device or NULL -> delta. */
int pxm_myula_step(const void* X, const void* gradg, const double* T, double T_scalar,
                   const double* delta_dev, double delta, double lmda, const void* noise,
                   int noise_complex, uint64_t seed, uint64_t chain0, uint64_t iter, void* X_out,
                   int64_t n, int C, int dtype, pxm_stream_t stream);
/* chain_step with proxf given (PxMALA keeps proxf of the current state, mcmc.py:231) */
int pxm_chain_step(const void* X, const void* proxf, const void* gradg, const double* delta_dev,
                   double delta, double lmda, const void* noise, int noise_complex, uint64_t seed,
                   uint64_t chain0, uint64_t iter, void* X_out, int64_t n, int C, int dtype,
                   pxm_stream_t stream);
/* The same two steps with a caller-owned device iteration counter: the Philox iteration is iter + *iter_dev, read
 * when the kernel runs (NULL: iter alone), so a HIP graph of an iteration replays with fresh noise once the
 * captured sequence ends with pxm_counter_add.  (MYULA's generic-operator stepping engine, pxmcmc/mcmc.py:157-164.) */
int pxm_myula_step_it(const void* X, const void* gradg, const double* T, double T_scalar,
                      const double* delta_dev, double delta, double lmda, const void* noise,
                      int noise_complex, uint64_t seed, uint64_t chain0, uint64_t iter,
                      const uint64_t* iter_dev, void* X_out, int64_t n, int C, int dtype,
                      pxm_stream_t stream);
int pxm_chain_step_it(const void* X, const void* proxf, const void* gradg, const double* delta_dev,
                      double delta, double lmda, const void* noise, int noise_complex, uint64_t seed,
                      uint64_t chain0, uint64_t iter, const uint64_t* iter_dev, void* X_out, int64_t n,
                      int C, int dtype, pxm_stream_t stream);
/* N(0,1) draws of the Philox4x32-10 stream keyed (seed, chain0+c, iter): out [C][n] (f64 or c128) */
int pxm_randn(void* out, int64_t n, int C, int dtype, uint64_t seed, uint64_t chain0, uint64_t iter,
              pxm_stream_t stream);
/* The Box-Muller step of that stream on GIVEN uniforms u1, u2 in (0, 1] (device arrays [n]): z0 = r cos(2 pi u2),
 * z1 = r sin(2 pi u2), r = sqrt(-2 ln u1); f64 != 0 selects the double-precision evaluation.  Test aid: the edge cases
 * (u1 rounding to 1, the smallest u1, quadrant boundaries of u2) have probability ~2^-53 under Philox. */
int pxm_box_muller(const double* u1, const double* u2, double* z0, double* z1, int64_t n, int f64, pxm_stream_t stream);
/* The reductions below are deterministic two-stage sums; `scratch` is a caller-owned device buffer of
 * pxm_reduce_scratch_doubles(C) doubles (no library-owned buffer is shared between calls or streams). */
int64_t pxm_reduce_scratch_doubles(int C);
/* L1.prior / S2_Wavelets_L1.prior (pxmcmc/prior.py:28-35,83-84): out[c] = sum_i |w_i X_ci| */
int pxm_reduce_l1(const void* X, const double* w, double* out, double* scratch, int64_t n, int C, int dtype,
                  pxm_stream_t stream);
/* logpi's L2 = vdot(d, invcov d), d = data - preds (pxmcmc/mcmc.py:78-79): out[c] = (re, im) */
int pxm_reduce_l2(const void* preds, const void* data, const void* invcov, int invcov_complex,
                  double* out, double* scratch, int64_t n, int C, int dtype, pxm_stream_t stream);
/* np.vdot(a, b) per chain: out[c] = sum conj(a) b as (re, im) -- logpi's L2 = vdot(d, invcov @ d) when invcov is a
 * full (sparse) matrix applied with pxm_csr_matvec (pxmcmc/forward.py:75-78, mcmc.py:78-79) */
int pxm_reduce_vdot(const void* a, const void* b, double* out, double* scratch, int64_t n, int C, int dtype,
                    pxm_stream_t stream);
/* uncertainty.credible_interval_range (pxmcmc/uncertainty.py:7-16) of a chain resident on the device: out[j] = Q(1 - alpha/2) -
 * Q(alpha/2) over the nsamples rows of column j of chain[nsamples][ld >= nparams] (float64), numpy.quantile's default "linear"
 * method reproduced (exact order statistics by radix select + numpy's lerp) */
int pxm_quantile_range(const double* chain, int64_t nsamples, int64_t nparams, int64_t ld, double alpha, double* out,
                       pxm_stream_t stream);
/* PxMALA.calc_logtransition, literal (pxmcmc/mcmc.py:281-289): out[c] = (re, im) */
int pxm_logtransition(const void* X1, const void* X2, const void* proxf, const void* gradg,
                      const double* delta_dev, double delta, double lmda, double* out, double* scratch,
                      int64_t n, int C, int dtype, pxm_stream_t stream);
/* Metropolis accept + state swap + delta adaptation for every chain (pxmcmc/mcmc.py:244-260,
 * 277-279).  logalpha_terms: [C][4] = (logtrans_pc, logpi_p, logtrans_cp, logpi_c) real parts.
 * For accepted chains copies prop -> curr for each of nbuf (buffer pairs, sizes in elements).
 * u: injected uniforms [C] or NULL (Philox).  accept_out [C] int32; delta_dev updated when tune. */
int pxm_pxmala_accept(const double* logalpha_terms, const double* u, uint64_t seed, uint64_t chain0,
                      uint64_t iter, int32_t* accept_out, double* delta_dev, int tune, double lmda,
                      int64_t it_index, int C, pxm_stream_t stream);
/* One PxMALA iteration with every per-iteration quantity on the device (pxmcmc/mcmc.py:230-260).
 *   pxm_pxmala_propose : X' = chain_step(X, proxf, gradg) with per-chain delta_dev [C]; proxf' = soft(X', T);
 *                        logtrans_out[c] = calc_logtransition(X, X', proxf, gradg) as (re, im);
 *                        prior_out[c] = sum |w X'| (w = prior_weights [n] or NULL) -- ONE pass over the state.
 *                        iter_dev: optional caller-owned device counter added to iter (HIP-graph replay).
 *                        scratch: 4 * pxm_reduce_scratch_doubles(C) doubles.  logtrans_out == prior_out == NULL: the
 *                        totals are DEFERRED -- the per-slice sums stay in `scratch` for pxm_pxmala_finish.
 *                        proxf == proxf_prop == NULL: the prox arrays are neither read nor written -- proxf = soft(X, T)
 *                        (the stock L1 prox, pxmcmc/prior.py:49-50) is formed in the kernel, and pxm_pxmala_finish forms
 *                        soft(X', T) the same way: 40 % fewer bytes in the pass and one array less in the conditional copy.
 *   pxm_pxmala_accept2 : logalpha = Re(logtrans_pc + logpi' - logtrans_cp - logpi), logpi' = -mu prior' - L2';
 *                        accept iff log(u) < logalpha (u injected [C] or the Philox uniform of (seed, chain, iteration));
 *                        accepted chains take (logpi', L2', prior') into their state scalars (logpi_c, L2_c as (re, im),
 *                        prior_c); delta_dev adapted when tune (:277-279) with the iteration number iter + *iter_dev;
 *                        acc_trace / delta_trace: optional [chunk][C] ring buffers written at row iteration % chunk.
 *   pxm_pxmala_finish  : everything between the proposal's gradient and the conditional copy in TWO launches instead of
 *                        seven: (i) ONE grid with the slices of the reverse transition sum of
 *                        calc_logtransition(X', X, proxf', gradg') [n, dtype; proxf_prop == NULL: proxf' = soft(X', T)
 *                        with T [n] or T_scalar] and of L2' = vdot(d, invcov d),
 *                        d = data - preds' [n_data, data_dtype; invcov as for pxm_reduce_l2]; (ii) ONE workgroup that
 *                        totals those and the deferred sums of pxm_pxmala_propose (`propose_scratch`: that call's scratch) in
 *                        the order of the separate reductions, stores logtrans_pc / logtrans_cp / L2' as (re, im) [C] and
 *                        prior' [C] for observers, and runs the test of pxm_pxmala_accept2.  bump_counter: optional
 *                        device counter advanced by one AFTER every chain has read iter_dev (replaces pxm_counter_add
 *                        in a captured iteration).  scratch: 2 * pxm_reduce_scratch_doubles(C) doubles.
 *   pxm_select_copy_many : up to 4 arrays per call, dst_a[c] = src_a[c] for accepted chains.
 *   pxm_counter_add    : *counter += inc on the stream (after every reader of the iteration). */
int pxm_pxmala_propose(const void* X, const void* proxf, const void* gradg, const double* T, double T_scalar,
                       const double* prior_weights, const double* delta_dev, double lmda, const void* noise,
                       int noise_complex, uint64_t seed, uint64_t chain0, uint64_t iter, const uint64_t* iter_dev,
                       void* X_prop, void* proxf_prop, double* logtrans_out, double* prior_out, double* scratch,
                       int64_t n, int C, int dtype, pxm_stream_t stream);
int pxm_pxmala_accept2(const double* logtrans_pc, const double* logtrans_cp, const double* prior_p, const double* L2_p,
                       double mu, double* logpi_c, double* L2_c, double* prior_c, const double* u, uint64_t seed,
                       uint64_t chain0, uint64_t iter, const uint64_t* iter_dev, int32_t* accept_out, double* delta_dev,
                       int tune, double lmda, int32_t* acc_trace, double* delta_trace, int chunk, int C,
                       pxm_stream_t stream);
int pxm_pxmala_finish(const void* X_prop, const void* X_curr, const void* proxf_prop, const double* T, double T_scalar,
                      const void* gradg_prop, int64_t n, int dtype, const void* preds_prop, const void* data, const void* invcov, int invcov_complex,
                      int64_t n_data, int data_dtype, const double* propose_scratch, double mu, double lmda, double* logpi_c,
                      double* L2_c, double* prior_c, const double* u, uint64_t seed, uint64_t chain0, uint64_t iter,
                      const uint64_t* iter_dev, int32_t* accept_out, double* delta_dev, int tune, int32_t* acc_trace,
                      double* delta_trace, int chunk, double* logtrans_pc_out, double* logtrans_cp_out, double* prior_p_out,
                      double* L2_p_out, double* scratch, uint64_t* bump_counter, int C, pxm_stream_t stream);
int pxm_select_copy_many(const int32_t* flag, int narrays, const void* const* src, void* const* dst, const int64_t* n,
                         const int* esize, int C, pxm_stream_t stream);
int pxm_counter_add(uint64_t* counter_dev, uint64_t inc, pxm_stream_t stream);
/* per-chain conditional copy: dst[c] = src[c] where flag[c] != 0; n elements of esize bytes */
int pxm_select_copy(const int32_t* flag, const void* src, void* dst, int64_t n, int esize, int C,
                    pxm_stream_t stream);

/* ---- weak-lensing measurement helpers (pxmcmc/measurements.py:151-171, 242-304) --------- */
/* out = flm .* kernel with entries [0,4) zeroed: harmonic_mapping (:162-171). kernel: [L*L] */
int pxm_wl_harmonic_mapping(const void* flm, const double* kernel, void* out, int64_t n, int C,
                            pxm_stream_t stream);
/* gather unmasked pixels and weight: out[c][k] = f[c][idx[k]] * w[k]   (mask_forward + cov_weight) */
int pxm_wl_mask_gather(const void* f, const int64_t* idx, const double* w, void* out, int64_t npix,
                       int64_t ndata, int C, pxm_stream_t stream);
/* weight and scatter into zeros: f[c][idx[k]] = g[c][k] * w[k]          (cov_weight + mask_adjoint) */
int pxm_wl_mask_scatter(const void* g, const int64_t* idx, const double* w, void* f, int64_t npix,
                        int64_t ndata, int C, pxm_stream_t stream);

/* ---- sparse path-integral measurement (pxmcmc/measurements.py:59-83) ---------------------- */
/* y[c][row] = sum_k vals[k] x[c][indices[k]], k in [indptr[row], indptr[row+1]): PathIntegral.forward with
 * the CSR of path_matrix, PathIntegral.adjoint with the CSR of path_matrix.getH().  vals: float64, or
 * complex128 iff vals_complex; x: [C][ncols], y: [C][nrows], float64 (dtype 0) or complex128 (dtype 1). */
int pxm_csr_matvec(const int64_t* indptr, const int32_t* indices, const void* vals, int vals_complex,
                   int64_t nrows, int64_t ncols, const void* x, void* y, int C, int dtype, pxm_stream_t stream);
/* The same product for a chain batch, through a caller-owned scratch of ncols * C elements of x's type: the operand is
 * first copied chain-minor ([ncols][C]) so that a gathered non-zero reads its C chains from one contiguous segment
 * instead of C cache lines.  Identical sums in identical order (bit-equal results); scratch NULL = pxm_csr_matvec. */
int pxm_csr_matvec_batched(const int64_t* indptr, const int32_t* indices, const void* vals, int vals_complex,
                           int64_t nrows, int64_t ncols, const void* x, void* y, int C, int dtype, void* scratch,
                           pxm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
