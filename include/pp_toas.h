/*
 * pp_toas.h -- C ABI of the MI355X wideband-TOA fit engine (libpptoas_hip.so).
 *
 * The reference (pennucci/PulsePortraiture) is pure Python with no FFI of its
 * own; the drop-in boundary is therefore the Python call signature, and this
 * header is what the Python shims in pulseportraiture_amd/ bind with ctypes
 * (INTEGRATION.md shows the stub a maintainer would add to the reference).
 * Each entry point names the reference interface it replaces (file:line under
 * the reference tree).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types;
 *   - every function returns an int status: PP_OK (0) or a negative PP_E*
 *     code, with a human-readable message available from pp_last_error();
 *     no exception crosses the boundary;
 *   - per-subint numerical status goes in return_code[] (PP_RC_*), the fit
 *     itself never fails the call (reference: pptoaslib.py:1016-1033);
 *   - the caller owns every host buffer; the library copies what it needs and
 *     keeps no host pointer after return.  Device scratch lives in the opaque
 *     context.  One context per (host thread, GPU); calls on one context are
 *     serialised by the caller.
 *   - portraits are C-contiguous [nsub][nchan][nbin]; nbin is a power of two
 *     in [32, 8192] (tuned plans) or any even number in [8, 4096] (every entry
 *     point, by the chirp-z route: the reference's numpy.fft takes every nbin).
 *   - a subint's outputs are a function of that subint's inputs alone: whatever
 *     else is in the batch, however it is cut into sub-batches, shards or ranks,
 *     the same bits come back (the reference fits subints in a plain loop,
 *     pptoas.py:344-489).
 */
#ifndef PP_TOAS_H
#define PP_TOAS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_ABI_VERSION 5

/* status codes */
#define PP_OK 0
#define PP_EINVAL (-1)   /* bad argument */
#define PP_EHIP (-2)     /* HIP runtime error */
#define PP_ENOMEM (-3)   /* device memory */
#define PP_ESTATE (-4)   /* call order / missing model */
#define PP_ENOTSUP (-5)  /* the request is valid but this shape / option set has no device path
                            (pp_fit_in.ref_seed: the caller then forms the guess with
                            pp_reference_phase_seed and fits in a second call) */

/* sample types of portraits */
#define PP_F64 0
#define PP_F32 1

/* per-subint return codes (return_code[]); 2 is what the reference's
 * trust-ncg normally reports with gtol=-1 (pptoaslib.py:1002) */
#define PP_RC_GRAD 0      /* gradient below tolerance */
#define PP_RC_MAXITER 1   /* iteration limit */
#define PP_RC_STALL 2     /* no predicted reduction left: converged to rounding */
#define PP_RC_NAN 3       /* NaN / singular objective */

/* minimiser (pp_fit_in.method; the reference's `method` argument,
 * pptoaslib.py:932, 993-1010).  The reference's answer for GM and scattering
 * fits is where SciPy's trust-ncg stops with gtol = -1 ("no predicted
 * reduction" in floating point, after truncated conjugate-gradient steps), up
 * to ~1.5e-9 rot short of the optimum: PP_METHOD_TRUST_NCG walks that very
 * iteration (scipy/optimize/_trustregion.py, _trustregion_ncg.py) on the device
 * and stops where it stops; PP_METHOD_NEWTON converges to the rounding of the
 * objective in fewer evaluations (what 'Newton-CG' and 'TNC' also aim at). */
#define PP_METHOD_TRUST_NCG 0
#define PP_METHOD_NEWTON 1

typedef struct pp_ctx pp_ctx;

/* Streams.  A context launches everything on its own non-blocking HIP stream
 * (pp_stream) and every entry point returns after synchronising it, so outputs --
 * host or device -- are complete on return.  Device pointers passed IN (portraits,
 * errs, masks, output tensors) must be complete as well: if they were produced on
 * another stream (e.g. an asynchronous copy or kernel of the caller's framework),
 * synchronise that stream, or make pp_stream wait on an event of it, before the
 * call.  The Python binding does the former for torch tensors. */

/* ---- context ---------------------------------------------------------- */
int pp_abi_version(void);
const char* pp_last_error(void);
int pp_create(int device_id, pp_ctx** out);
int pp_destroy(pp_ctx* ctx);
/* block until everything queued on the context's stream has finished.  A batch enqueued with pp_fit_enqueue whose
 * solve and post-fit stage were still waiting for a next batch to carry them (option "fuse_tail") gets them queued
 * first, by the stand-alone kernels: after pp_synchronize the DEVICE-resident outputs of every enqueued batch
 * (records_dev, per-channel arrays with chan_on_device) are written; host outputs are complete after pp_fit_collect. */
int pp_synchronize(pp_ctx* ctx);
/* the hipStream_t the context launches on (as void*) */
void* pp_stream(pp_ctx* ctx);

/* Options (name, value):
 *   "harm_eps"     relative |m_nk| threshold below which the model's trailing
 *                  harmonics are dropped from the cross-spectrum (0 = keep all;
 *                  default 2^-50, DESIGN.md "Harmonic truncation")
 *   "max_iter"     trust-region iteration limit (default 64; reference: 1000)
 *   "profile"      1 = record HIP events around every kernel (pp_kernel_times)
 *   "check_every"  iterations between host checks of the active count
 *   "lagged_check" 1 (default) = evaluation loops without the scattering model read that count
 *                  one iteration behind, so the GPU never waits for the host's look at it
 *                  (one extra, empty iteration is queued at the end); 0 = synchronous checks
 *   "x_pad"        pad (elements) of the rows of a stored cross-spectrum (default 0: measured
 *                  neutral)
 *   "max_work_bytes"  cap on device scratch per call (larger batches are split)
 *   "taylor"       1 (default) = fits without scattering first try the
 *                  per-channel Taylor-model solve; 0 = always iterate
 *   "taylor_recentre"  fits of phase / DM whose Taylor model fails its certificate (poor
 *                  guesses) are expanded again about the tentative answer -- one more pass
 *                  over those subints' data -- before falling back to evaluations over a
 *                  stored cross-spectrum: how many times (default 1, 0 = never)
 *   "seed_chan_stride"  device phase seed (seed_ns > 0) of fits without scattering:
 *                  the seed is formed in a pilot pass over every n-th channel (default
 *                  16; 1 = from all channels, with the cross-spectrum stored), then the
 *                  usual single pass runs at the seeded phase
 *   "seed_min_snr" pilot seeds whose correlation peak stands less than this many rms
 *                  above the mean of the grid (default 8) are redone from all channels
 *   "seed_ndm", "seed_dm_step"  the coarse seed grid is (phi, DM): seed_ns phases x
 *                  seed_ndm trial DMs spaced seed_dm_step [pc cm^-3] about the guessed DM
 *                  (centred; an odd count keeps the guess itself); the DM of the highest
 *                  correlation peak, refined by the parabola through it and its
 *                  neighbours, seeds the DM, and the phase is seeded at it.  Default 1 trial:
 *                  the reference's seed trusts the header DM (pptoas.py:421-457)
 *   "paired_split" 1 (default) = 2048-bin rows whose template keeps fewer than 512
 *                  harmonics take the transform kernel that does the last FFT stage
 *                  and the even/odd split in registers; 0 = the generic kernel
 *   "one_exchange" 1 (default) = 2048- and 1024-bin rows fitted without scattering (noise
 *                  given or measured) take the transform kernels whose FFT crosses the LDS
 *                  once (first exchange by lane swaps) and whose split reads only the partner
 *                  harmonics -- k_xspec_q1024 / k_xspec_qf<1024> / k_xspec_qf<512> --, and the
 *                  single-pass reference seed (pp_seed_ref) exists; 0 = the kernels above
 *   "scat_model"   scattering fits: 1 (default) = once the trust-ncg iteration is
 *                  predicted to stay within its range, ONE more pass over the
 *                  cross-spectrum leaves a degree-8 polynomial model of every channel's
 *                  sums in (phi_n, tau_n) and the remaining evaluations are made on it,
 *                  each with a truncation certificate (DESIGN.md "Scattering fits");
 *                  2 = also for method 'newton' (whose few closing iterations do not
 *                  repay the pass); 0 = every evaluation is a pass over the cross-spectrum
 *   "fuse_scat"    1 (default) = scattering fits of 2048-bin portraits (template cut below 512
 *                  harmonics, noise given) take the transform that stores the cross-spectrum AND
 *                  forms the first evaluation's sums while it is in registers (one pass over the
 *                  stored cross-spectrum fewer); 0 = the general transform + an evaluation pass
 *   "x_f32"        1 = scattering fits keep their stored cross-spectrum X_nk = d_nk m_nk* as
 *                  pairs of floats (half the bytes of every evaluation pass, all arithmetic f64;
 *                  without the closing model).  Off by default: it buys 8 % on configs[3] and
 *                  costs chi2 its 1e-10 agreement with the reference
 *   "scat_model_tol"  predicted relative truncation below which that pass is asked
 *                  for (default 1e-10; the certificate guards the result either way)
 *   "fps_finish"   pp_fit_phase_shift_batch after its brute grid: 0 (default) = Newton to
 *                  the exact local optimum; 1 = what scipy.optimize.brute does by default
 *                  and the reference therefore returns (pplib.py:2085): the Nelder-Mead
 *                  simplex to xtol = ftol = 1e-4, operation by operation
 *   "skip_masked"  1 (default) = channels a subint's chan_mask removes are not transformed at
 *                  all: the transform walks a compact list of the (subint, channel) rows in use,
 *                  as the reference slices the good channels away before its fit
 *                  (pptoas.py:384-397); 0 = every row is transformed and masked ones get weight 0
 *   "eager_flush"  1 (default) = the stream is queried once the transform has been queued, which makes the
 *                  runtime hand what is queued to the GPU at once instead of with the next blocking call
 *   "solve_threads"  one-pass flow: threads per subint of the solve on the Taylor model (0 = default: 64 for a
 *                  band of up to 512 channels, 128 up to 1024, 256 beyond; 64 / 128 / 256 / 512 force it).  The
 *                  channel sums are taken in a different order for each: results agree to rounding, not bitwise
 *   "solve_prefetch"  that solve at more than 2048 channels: 0 (default) = each Taylor row fetched when its turn comes
 *                  (the batch's rows exceed the Infinity Cache there: the less a CU keeps in flight, the less it evicts),
 *                  1 = rows prefetched two ahead as for narrower bands
 *   "solve_cache"  ... channels whose weight, phase geometry and template power that solve keeps in LDS
 *                  (-1 = default: all, up to 8 per thread; 0 = formed again on every evaluation)
 *   "finalize_regs"  post-fit stage of fits without scattering: 1 (default) = each thread holds its (up to 8)
 *                  channels' numbers in registers over the stage's passes, 64 ... 512 threads per subint by band
 *                  width; 0 = the pass-by-pass kernel every other fit uses; 64 / 128 / 256 / 512 force a width
 *   "copy_kernels"  1 (default) = the packed block of small inputs and the packed per-subint outputs cross PCIe by a
 *                  kernel that reads / writes the pinned staging block directly; 0 = by hipMemcpyAsync (a copy
 *                  command between two kernels hands the stream to the copy engine and back: ~0.05 ms per batch)
 *   "fuse_tail"    pp_fit_enqueue, one-pass fits of 2048-bin portraits: 1 (default) = the batch's solve on the Taylor model and
 *                  its post-fit stage are NOT queued behind its transform; the transform of the next enqueued batch works
 *                  them off, one subint per ticket, between its own rows (one wave per ticket walking the waves of the
 *                  stand-alone kernels in turn: bitwise their results), and the batch's outputs and event follow that
 *                  transform on the stream.  The reference-seed flow (pp_seed_ref, what get_TOAs runs by default) is part
 *                  of this since round 6: its pass (k_xspec_qr1024) carries tickets, and ITS tail -- the guess's spectrum
 *                  from the pass's chunk partials, fit_phase_shift with SciPy's simplex, the start points, then solve and
 *                  post-fit stage -- is a ticket of the next pass (+2.7 % fits/s, profiles/r06_refseed_tail_ab.txt).  If no
 *                  batch follows (pp_fit_collect or pp_synchronize comes first), the next batch cannot carry them (another
 *                  row length, scattering, a device seed) or is much smaller than the one it would carry (fewer than half
 *                  as many waves as tickets), the stand-alone kernels are queued then.  Keep THREE batches enqueued to hide
 *                  it all (a batch completes one transform later).  +0.6 ... 2.9 % fits/s (profiles/r05_fuse_tail_ab.txt);
 *                  0 = solve and post-fit stage queued at once
 *   "tail_virtual" experiments: 1 = the stand-alone solve / post-fit kernels themselves in their one-wave form
 *   "check_from"   evaluation loop: the first iteration after which the host looks at the count of unfinished subints
 *                  (default 2; trust-ncg scattering fits with the closing model: 5 at least -- none is done before)
 *   "refseed_stride"  pp_seed_ref: channel stride of the pilot pass on wide bands (0 = default 64; at least 32 pilot
 *                  channels are kept).  128 measured +0.4 % at 4096 channels: not taken, a thinner pilot is a weaker one
 *   "overlap_post"  pp_fit_enqueue: 1 = the solve and post-fit stage of a deferred batch are queued on a second,
 *                  higher-priority stream of the context behind an event of its transform, with a work-buffer set
 *                  of their own, so that they may run beside the NEXT batch's transform; 0 (default) = one stream.
 *                  Same results either way.  Measured neutral (profiles/r05_overlap_ab.txt): the persistent transform
 *                  holds every wave slot, the two kernels alternate instead of co-residing
 *   "coarse_newton"  1 (default) = scattering fits with PP_METHOD_NEWTON first iterate on every 16th channel
 *                  (each evaluation reads a sixteenth of the stored cross-spectrum) and start the
 *                  full-channel iteration from that answer -- the optimum does not depend on the path --;
 *                  0 = every evaluation over all channels.  nfeval includes the coarse evaluations, npass
 *                  counts full passes.  (PP_METHOD_TRUST_NCG retraces SciPy's iterates and is not affected.)
 *   "nfev_shadow"  one-pass flow, method trust-ncg: how SciPy's one-point cache is mirrored when nfeval is
 *                  counted.  0: proposals are compared as displacements from the expansion point
 *                  (resolution 1e-21: the closing proposal p = -H^-1 g is always a new point and counts);
 *                  1: on the absolute iterate fl(x + p) as SciPy forms it; 2: as 1, with the model evaluated
 *                  at the rounded point; -1 (default): 1 for one-parameter fits, 0 otherwise -- the rule
 *                  that agrees with the reference's count most often, family by family (3000 random fits,
 *                  profiles/r04_parity_sweep.txt; the closing p is rounding noise of whoever computes it, so
 *                  the last unit of nfeval is a coin toss of the reference's own arithmetic)
 *   "moments_in_xspec"  1 (default) = the Taylor moments are accumulated inside
 *                  the transform kernel and no cross-spectrum is stored; 0 = store
 *                  the cross-spectrum and take the moments in a second pass
 */
int pp_set_option(pp_ctx* ctx, const char* name, double value);
/* the current value of an option (a caller that changes one for a call -- max_iter = 0 to
 * evaluate only -- reads it first and puts it back) */
int pp_get_option(pp_ctx* ctx, const char* name, double* value);

/* ---- model portraits ---------------------------------------------------- */
/* Upload an nchan x nbin template into slot `slot` (0..PP_MAX_SLOTS-1), rFFT it
 * on the device, zero the DC harmonic and keep sum_k |m_nk|^2 per channel.
 * Replaces the model half of pptoaslib.py:978-979 (and pplib.py:2123-2124).
 * `on_device` != 0 means `portrait` is a device pointer on this GPU. */
#define PP_MAX_SLOTS 64
int pp_model_set(pp_ctx* ctx, int slot, const void* portrait, int dtype,
                 int on_device, int nchan, int nbin);
/* number of harmonics (of nbin/2) kept for this slot after truncation */
int pp_model_nharm(pp_ctx* ctx, int slot);
/* DC harmonic of every channel of the slot's template, dc[nchan] (host): nbin x the
 * profile mean the flux estimate of get_TOAs uses (pptoas.py:554-575) */
int pp_model_dc(pp_ctx* ctx, int slot, double* dc);

/* Gaussian-component template portraits synthesised on the device (.gmodel
 * files: read_model / gen_gaussian_portrait / gaussian_profile / evolve_parameter,
 * pplib.py:2867-2953, 853-930, 770-825, 996-1046).  code[3]: '0' = power-law,
 * '1' = linear evolution of (loc, wid, amp); comps[ngauss][6] = loc, m_loc, wid,
 * m_wid, amp, m_amp at nu_ref; dc = baseline; tau_rot = scattering time at nu_ref
 * in rotations (TAU / P, 0 = none), scaled as (nu/nu_ref)^alpha and applied in the
 * Fourier domain (pplib.py:915-922).  pp_gaussian_portrait writes the
 * [nchan][nbin] f64 portrait (host or device pointer); pp_model_set_gaussian
 * loads it straight into a model slot (no host copy of the portrait at all). */
int pp_gaussian_portrait(pp_ctx* ctx, int nchan, int nbin, const double* freqs,
                         const char* code, double nu_ref, double dc, double tau_rot,
                         double alpha, int ngauss, const double* comps,
                         double* portrait, int out_on_device);
int pp_model_set_gaussian(pp_ctx* ctx, int slot, int nchan, int nbin,
                          const double* freqs, const char* code, double nu_ref,
                          double dc, double tau_rot, double alpha, int ngauss,
                          const double* comps);

/* Spline (PCA + B-spline) template portraits synthesised on the device (.spl files:
 * read_spline_model / gen_spline_portrait, pplib.py:2955-2987, 932-956): row n =
 * basis[0] + sum_c splev(freqs[n]; t, coefs[c], degree) * basis[1 + c], where basis
 * [ncomp + 1][nbin] holds the mean profile and the eigenvectors (resampled to nbin
 * on the host when the model's own resolution differs), t[nknots] the knots and
 * coefs[ncomp][nknots] the B-spline coefficients of scipy's `tck` (FITPACK layout).
 * The curve is evaluated like scipy.interpolate.splev(..., ext=0).  pp_spline_portrait
 * writes the [nchan][nbin] f64 portrait (host or device pointer);
 * pp_model_set_spline loads it straight into a model slot. */
int pp_spline_portrait(pp_ctx* ctx, int nchan, int nbin, const double* freqs, int ncomp,
                       const double* basis, int nknots, const double* t,
                       const double* coefs, int degree, double* portrait,
                       int out_on_device);
int pp_model_set_spline(pp_ctx* ctx, int slot, int nchan, int nbin, const double* freqs,
                        int ncomp, const double* basis, int nknots, const double* t,
                        const double* coefs, int degree);

/* Instrumental response applied to the template resident in `slot`, in the Fourier
 * domain on the device (instrumental_response_port_FT, pptoaslib.py:145-179, as
 * get_TOAs(add_instrumental_response=True) uses it, pptoas.py:388-394):
 * m_nk <- m_nk * rconst[k] * sinc(k smear_wid[n]).  rconst: host [nbin/2 + 1]
 * interleaved (re, im), the product of the constant responses, or NULL; smear_wid:
 * host [nchan], the dispersive smearing width of every channel in rotations
 * (8.3e-6 chan_bw / nu_GHz^3 / P as the reference forms it; 0 = none), or NULL.  The
 * harmonic truncation of the slot is re-derived. */
int pp_model_apply_response(pp_ctx* ctx, int slot, const double* rconst,
                            const double* smear_wid);

/* ---- the batched fit ----------------------------------------------------- */
/* The reference's own initial phase guess formed INSIDE the fit, from the same single pass over
 * the portraits (pp_fit_in.ref_seed; pptoas.py:421-457):
 *   rot_prof_i = np.average(rotate_data(port_i, 0.0, DM_i, P_i, freqs_i, nu_mean_i), axis=0,
 *                           weights=weights_i)                    (DM_i = init_params[i][1])
 *   phi_i      = fit_phase_shift(rot_prof_i, model_prof_i, Ns=Ns, bounds=(lo, hi)).phase
 *   init_params[i][0] <- phase_transform(phi_i, DM_i, nu_mean_i, nu_fits[i][0], P_i, mod=True)
 * and the fit then starts from there (init_params[.][0] as given is ignored).  The transform
 * kernel takes the per-channel Taylor model about a provisional phase (a pilot pass over every
 * 16th channel) together with the rotated channel sums; the iteration starts off-centre, at the
 * reference's guess.  A scattering fit iterates over the stored cross-spectrum instead: the
 * transform that stores it takes the rotated channel sums as well (rotation by the DM guess
 * alone, no pilot) and the iteration starts AT the reference's guess.  Available for 2048-bin
 * portraits whose template keeps fewer than 512 harmonics, nchan a multiple of 32 (and >= 256
 * without scattering), errs given, GM guesses 0: otherwise pp_fit_portrait_batch returns
 * PP_ENOTSUP and nothing has been done. */
typedef struct {
    const double* weights;       /* [nsub][nchan] weights of the channel mean (0 = channel not used);
                                    host, or device when pp_fit_in.aux_on_device; NULL = all 1 */
    const double* model_profs;   /* host: [nsub][nbin], or [nbin] when model_prof_stride == 0 */
    int64_t model_prof_stride;   /* 0 or nbin */
    const double* nu_mean;       /* host [nsub]: mean frequency of the channels used */
    double lo, hi;               /* bounds of the brute grid (-0.5, 0.5) */
    int32_t Ns;                  /* its size (100) */
    int32_t finish;              /* 1 = SciPy brute's simplex finish (what the reference returns), 0 = Newton */
    double* seed_phase;          /* host [nsub] out, or NULL: the phase guesses formed (at nu_fit) */
} pp_seed_ref;

typedef struct {
    int32_t nsub, nchan, nbin;
    const void* data;          /* [nsub][nchan][nbin] */
    int32_t data_dtype;        /* PP_F64 | PP_F32 */
    int32_t data_on_device;    /* data is a device pointer */
    const int32_t* model_slot; /* [nsub] or NULL (= slot 0 for all) */
    const double* freqs;       /* [nsub][nchan] or [nchan] if freqs_stride==0 */
    int64_t freqs_stride;      /* 0 or nchan */
    const double* errs;        /* [nsub][nchan] time-domain sigma, or NULL:
                                  measured per channel as get_noise_PS
                                  (pplib.py:2227-2247) */
    const uint8_t* chan_mask;  /* [nsub][nchan], 1 = fit this channel, or NULL */
    int32_t aux_on_device;     /* errs and chan_mask are device pointers */
    const double* P;           /* [nsub] spin period [s] */
    const double* init_params; /* [nsub][5] phi, DM, GM, tau|log10 tau, alpha */
    const double* nu_fits;     /* [nsub][3] or NULL; NaN = mean(freqs) */
    const double* nu_outs;     /* [nsub][3] or NULL; NaN = zero-covariance */
    int32_t fit_flags[5];
    int32_t log10_tau;
    int32_t option;            /* get_nu_zeros option (pptoaslib.py:734) */
    int32_t is_toa;
    int32_t method;            /* PP_METHOD_* */
    int32_t seed_ns;           /* > 0: ignore init_params[.][0] and seed the phase on
                                  the device: seed_ns-point grid over [-0.5, 0.5] of
                                  the channel-summed cross-correlation at the guessed
                                  DM/GM/tau, polished to its maximum (the role of
                                  pptoas.py:421-457, fit_phase_shift with Ns=100) */
    const pp_seed_ref* ref_seed; /* NULL, or: form the reference's own phase guess in the pass (above) */
} pp_fit_in;

typedef struct {
    double* params;       /* [nsub][5]  phi(nu_out), DM, GM, tau(nu_out), alpha */
    double* param_errs;   /* [nsub][5]  0 where not fitted */
    double* nu_refs;      /* [nsub][3]  nu_DM, nu_GM, nu_tau of the outputs */
    double* cov;          /* [nsub][5][5] covariance, zero rows/cols if unfit */
    double* chi2;         /* [nsub] */
    double* red_chi2;     /* [nsub] */
    double* snr;          /* [nsub] */
    int32_t* nfeval;      /* [nsub] objective evaluations as the reference reports them: SciPy's
                             `nfev` (pptoaslib.py:1017 `nfeval = results.nfev`) -- the initial point,
                             every proposal that is not the point evaluated last (SciPy's one-point
                             cache), and the proposal SciPy evaluates before it tests the predicted
                             reduction and stops.  Most of them cost no pass over the data here. */
    int32_t* return_code; /* [nsub] PP_RC_* */
    int32_t chan_on_device; /* the three per-channel outputs are device pointers */
    double* scales;       /* [nsub][nchan] or NULL */
    double* scale_errs;   /* [nsub][nchan] or NULL */
    double* channel_snrs; /* [nsub][nchan] or NULL */
    double* obj_f;        /* [nsub]      objective at init_params, or NULL */
    double* obj_grad;     /* [nsub][5]   its gradient, or NULL */
    double* obj_hess;     /* [nsub][25]  its Hessian, or NULL */
    double* duration;     /* [1] seconds of device time for the whole call */
    int32_t* npass;       /* [nsub] or NULL: how many of those evaluations were passes over the
                             portraits or the stored cross-spectrum (1 for a fit solved on the
                             Taylor model of its single pass) */
    double* records_dev;  /* DEVICE pointer [nsub][PP_RECORD_WIDTH] or NULL: one fixed-size
                             TOA record per subint left in HBM -- phi, DM, GM, tau, alpha,
                             their five errors, nu_DM, nu_GM, nu_tau, chi2, red_chi2, snr,
                             nfeval, return_code -- what the ranks of a multi-GPU job
                             gather (one RCCL gather at the end of the job) */
} pp_fit_out;
#define PP_RECORD_WIDTH 18

/* Fit every subint of the batch: rFFT, cross-spectrum, trust-region Newton
 * solve, zero-covariance frequencies, errors, S/N, chi2.  One call replaces
 * nsub calls of fit_portrait_full (pptoaslib.py:928-1096), i.e. the body of
 * the per-subint loop of GetTOAs.get_TOAs (pptoas.py:344-489).
 * max_iter == 0 only evaluates the objective at init_params (obj_* outputs). */
int pp_fit_portrait_batch(pp_ctx* ctx, const pp_fit_in* in, pp_fit_out* out);

/* Asynchronous form.  pp_fit_submit copies the two argument blocks, starts
 * the batch on a worker thread of the context and returns at once; pp_fit_wait blocks
 * until it has finished and returns what pp_fit_portrait_batch would have returned (the
 * message of a failure is then available from pp_last_error on the waiting thread);
 * pp_fit_poll returns 1 once the batch is complete, 0 while it runs.  Every buffer the
 * argument blocks point to must stay valid and untouched until pp_fit_wait returns; one
 * batch per context may be in flight, and no other call may be made on that context
 * meanwhile (PP_ESTATE on a second submit).  Batches submitted on two contexts of one
 * GPU overlap: one context's host-to-device copies run beside the other's kernels.
 * The reference has no counterpart: its loop over subints is serial
 * (pptoas.py:344-489). */
int pp_fit_submit(pp_ctx* ctx, const pp_fit_in* in, pp_fit_out* out);
int pp_fit_poll(pp_ctx* ctx);
int pp_fit_wait(pp_ctx* ctx);

/* Stream-ordered batches (one context, one stream, no extra thread).  pp_fit_enqueue copies the two
 * argument blocks and queues the WHOLE batch on the context's stream -- small inputs, every kernel,
 * the outputs on their way to a pinned staging block -- and returns without waiting; pp_fit_collect
 * completes the OLDEST enqueued batch (waits for it, fills the caller's output arrays) and returns what
 * pp_fit_portrait_batch would have returned.  Up to THREE batches may be pending (each has its own
 * staging blocks and its own set of the device work buffers its solve and post-fit stage use), so a
 * caller that enqueues batches k + 1, k + 2 before it collects batch k keeps the GPU busy while the host
 * marshals.  One-pass fits of 2048-bin portraits leave their solve and post-fit stage to the transform of
 * the NEXT enqueued batch (option "fuse_tail", default on): batch k is then complete one transform later,
 * or when pp_fit_collect finds it the youngest and queues them itself.  Same results either way, to the
 * bit, as pp_fit_portrait_batch.  Batches whose flow needs a host decision in its middle (scattering fits,
 * device seeds, sub-batching) simply run to their end inside pp_fit_enqueue; the one-pass flow, with
 * the caller's guesses or with the reference's own (pp_seed_ref; phase / DM / GM fits), is deferred whole; a one-pass
 * batch in which some subint fails its certificate (poor guesses) is fitted again by the general
 * flow inside pp_fit_collect.  Every buffer the argument blocks point to (pp_seed_ref included) must
 * stay valid and untouched until the batch has been collected; no other fit call may be made on the
 * context while batches are pending (PP_ESTATE).  pp_fit_pending returns how many are.
 * The reference has no counterpart: its loop over subints is serial (pptoas.py:344-489). */
int pp_fit_enqueue(pp_ctx* ctx, const pp_fit_in* in, pp_fit_out* out);
int pp_fit_collect(pp_ctx* ctx);
int pp_fit_pending(pp_ctx* ctx);

/* ---- building blocks exported for parity tests --------------------------- */
/* rFFT of nrows real rows of length nbin (host pointers); out holds
 * nrows*(nbin/2+1) interleaved (re,im) doubles.  numpy.fft.rfft of
 * pptoaslib.py:976-978. */
int pp_rfft_rows(pp_ctx* ctx, const void* rows, int dtype, int nrows, int nbin,
                 double* out);

/* 1-D FFTFIT of nprof (data, model) profile pairs: 7 doubles per pair
 * (phase, phase_err, scale, scale_err, snr, red_chi2, duration).
 * fit_phase_shift (pplib.py:2054-2100).  noise[i] < 0 or NaN = measure it. */
int pp_fit_phase_shift_batch(pp_ctx* ctx, const double* data,
                             const double* model, const double* noise,
                             int nprof, int nbin, double lo, double hi, int Ns,
                             double* out7);

/* The reference's initial phase guess of nsub subints (pptoas.py:421-457), data side
 * fused into one read of the portraits:
 *   rot_prof_i = np.average(rotate_data(port_i, phi_i, DM_i, P_i, freqs_i, nu_DM),
 *                           axis=0, weights=weights_i)      (par3 = phi, DM, GM per subint)
 *   out7[i]    = fit_phase_shift(rot_prof_i, model_profs_i, Ns=Ns, bounds=(lo, hi))
 * with the finish selected by option "fps_finish".  Channels of zero weight are not
 * read.  weights [nsub][nchan] and model_profs [nsub][nbin] are host arrays; `src` is
 * a host or device pointer (`on_device`), dtype PP_F64 / PP_F32. */
int pp_reference_phase_seed(pp_ctx* ctx, const void* src, int dtype, int on_device,
                            int nsub, int nchan, int nbin, const double* freqs,
                            int64_t freqs_stride, const double* P, const double* par3,
                            double nu_DM, double nu_GM, const double* weights,
                            const double* model_profs, double lo, double hi, int Ns,
                            double* out7);

/* Fourier rotation / (de)dispersion of portraits: dst[i][n] = irfft(rfft(src[i][n])
 * e^{2 pi i k phi_in}), phi_in = par[i][0] + Dconst par[i][1] (nu_n^-2 - nu_DM^-2)/P_i
 * + Dconst^2 par[i][2] (nu_n^-4 - nu_GM^-4)/P_i.  rotate_data (pplib.py:2338-2426),
 * rotate_portrait (:2428-2460), rotate_portrait_full (pptoaslib.py:52-81).
 * src/dst: [nsub][nchan][nbin] of `dtype`, both host or both device pointers
 * (dst may equal src on the device); nu_DM / nu_GM may be INFINITY. */
int pp_rotate_portraits(pp_ctx* ctx, const void* src, void* dst, int dtype,
                        int on_device, int nsub, int nchan, int nbin,
                        const double* freqs, int64_t freqs_stride, const double* P,
                        const double* par3, double nu_DM, double nu_GM);

/* ---- synthetic portraits generated on the device ------------------------- */
/* ppalign's accumulation (ppalign.py:199-206): aligned[n][:] = sum_i w[i][n] *
 * rotate_data(src[i][n], phase_i, DM_i, P_i, freqs, nu_ref_i) and total_weights[n] =
 * sum_i w[i][n]; rows with w = 0 (or NaN) are skipped.  par3[i] = {phase, DM, nu_ref}
 * (nu_ref may be INFINITY).  src: [nsub][nchan][nbin] of `dtype`, host or device;
 * aligned [nchan][nbin] and total_weights [nchan] are host arrays, overwritten. */
int pp_align_accumulate(pp_ctx* ctx, const void* src, int dtype, int on_device,
                        int nsub, int nchan, int nbin, const double* freqs,
                        int64_t freqs_stride, const double* P, const double* par3,
                        const double* weights, double* aligned, double* total_weights);

/* Per-channel reduced chi^2 of fitted subints in the time domain, as
 * get_channels_to_zap forms it (pptoas.py:1239-1245 via show_fit :1394-1404 and
 * get_red_chi2 pplib.py:727-750): sum over bins of (data rotated by the fitted
 * phi, DM, GM  -  scales[n] x template, scattered by tau, alpha when tau != 0)^2
 * / errs[n]^2 / (nbin - 2).  params5[i] = {phi, DM, GM, tau [rot, linear], alpha}
 * at nu_refs3[i]; the template is the model slot of each subint (null: slot 0);
 * red_chi2: [nsub][nchan] host array. */
int pp_channel_red_chi2(pp_ctx* ctx, const void* src, int dtype, int on_device,
                        int nsub, int nchan, int nbin, const int32_t* model_slot,
                        const double* freqs, int64_t freqs_stride, const double* P,
                        const double* params5, const double* nu_refs3,
                        const double* scales, const double* errs, double* red_chi2);

/* Fill dst[nsub][nchan][nbin] (device pointer, dtype) with
 *   gains[i][n] * rotate(model slot, -phi_i, -DM_i, -GM_i) + N(0, sigma)
 * using a counter-based RNG keyed on (seed, first_subint + i, channel, bin).
 * inj is host [nsub][3] (phi, DM, GM injected, reference frequency infinity); gains is host
 * [nsub][nchan] -- per-channel amplitudes, e.g. the scintillation pattern of add_scintillation
 * (pplib.py:1146-1174) -- or NULL (= 1 everywhere).  The synthetic inputs of SURVEY 8(d). */
int pp_synth_portraits(pp_ctx* ctx, int slot, void* dst, int dtype, int nsub,
                       const double* freqs, const double* P, const double* inj,
                       const double* gains, double sigma, uint64_t seed, int64_t first_subint);

/* ---- measurement --------------------------------------------------------- */
/* Accumulated HIP-event time per kernel family since the last reset (only
 * while option "profile" = 1): names[i] points to a static string.
 * Returns the number of families written (<= cap). */
int pp_kernel_times(pp_ctx* ctx, int cap, const char** names, double* seconds,
                    int64_t* launches);
int pp_kernel_times_reset(pp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PP_TOAS_H */
