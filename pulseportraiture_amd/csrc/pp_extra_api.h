// host entry points for pp_extra.h (included at the end of pp_toas.hip)

// Host-resident inputs larger than the work-memory budget go through the device in
// runs of whole subints (every operation here is independent per subint, or a sum
// over them): how many subints of `per_sub` device bytes fit
static int aux_chunk_cap(pp_ctx* c, double per_sub, int nsub) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return nsub;
    const double budget = std::min(c->max_work_bytes, 0.85 * ((double)free_b + (double)c->data.cap + (double)c->X.cap));
    return (int)std::min<double>(nsub, std::max(1.0, std::floor(budget / std::max(per_sub, 1.0))));
}

extern "C" int pp_fit_phase_shift_batch(pp_ctx* c, const double* data, const double* model, const double* noise,
                                        int nprof, int nbin, double lo, double hi, int Ns, double* out7) {
    if (int busy_ = ctx_busy(c, "pp_fit_phase_shift_batch")) return busy_;
    if (!c || !data || !model || !out7) return fail(PP_EINVAL, "pp_fit_phase_shift_batch: null argument");
    if (!nbin_any_ok(nbin)) return nbin_refuse("pp_fit_phase_shift_batch", nbin);
    const bool anyb = !nbin_ok(nbin);
    if (nprof < 1) return fail(PP_EINVAL, "pp_fit_phase_shift_batch: bad shape %d x %d", nprof, nbin);
    if (Ns < 1) return fail(PP_EINVAL, "pp_fit_phase_shift_batch: Ns %d", Ns);
    HIP_TRY(hipSetDevice(c->device));
    const int M = nbin / 2;
    int rc;
    {
        const int cap = aux_chunk_cap(c, 2.0 * nbin * 8 + (2.0 * (M + 1) + M) * 16 + 64, nprof);
        if (nprof > cap) {
            for (int p0 = 0; p0 < nprof; p0 += cap) {
                const int n = std::min(cap, nprof - p0);
                if ((rc = pp_fit_phase_shift_batch(c, data + (size_t)p0 * nbin, model + (size_t)p0 * nbin,
                                                   noise ? noise + p0 : nullptr, n, nbin, lo, hi, Ns, out7 + (size_t)p0 * 7)))
                    return rc;
            }
            return PP_OK;
        }
    }
    // interleave rows: data_i, model_i
    const size_t rowb = (size_t)nbin * 8;
    if ((rc = c->data.reserve(2 * (size_t)nprof * rowb))) return rc;
    HIP_TRY(hipMemcpy2DAsync(c->data.p, 2 * rowb, data, rowb, rowb, nprof, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpy2DAsync((char*)c->data.p + rowb, 2 * rowb, model, rowb, rowb, nprof, hipMemcpyHostToDevice, c->stream));
    if ((rc = c->X.reserve((size_t)nprof * (2 * (size_t)(M + 1) + M) * sizeof(cplx)))) return rc;
    if ((rc = c->o_params.reserve((size_t)nprof * 56))) return rc;
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, nbin, &tw))) return rc;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    cplx* spec = c->X.as<cplx>();
    cplx* xwork = spec + 2 * (size_t)nprof * (M + 1);
    {
        Prof pr(c, KF_FPS);
        if (anyb) {
            // (row lengths without a tuned plan: the same spectra by the chirp-z path, pp_anybin.h)
            XspecArgs xa;
            memset(&xa, 0, sizeof xa);
            xa.data = c->data.p; xa.nsub = 1; xa.nchan = 2 * nprof; xa.nchan_full = 2 * nprof; xa.cstep = 1;
            if ((rc = launch_any(c, xa, nbin, ((M + 63) / 64) * 64, PP_F64, -1, false, spec, nullptr))) return rc;
        } else {
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            hipLaunchKernelGGL((k_rfft_rows<MM, double>), dim3(fft_grid(T, 2 * nprof)), dim3(T), 0, c->stream,
                               (const void*)c->data.p, spec, tw, 2 * nprof);
        });
        }
        const double* dnoise = nullptr;
        if (noise) {
            if ((rc = upload(c, c->errs, noise, (size_t)nprof * 8))) return rc;
            dnoise = c->errs.as<double>();
        }
        FpsArgs fa{spec, dnoise, c->o_params.as<double>(), lo, hi, Ns, M, nprof, c->fps_finish, nullptr, M + 1};
        hipLaunchKernelGGL(k_fps, dim3(nprof), dim3(256), 0, c->stream, fa, xwork);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipMemcpyAsync(out7, c->o_params.p, (size_t)nprof * 56, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    for (int i = 0; i < nprof; ++i) out7[(size_t)i * 7 + 6] = 1e-3 * ms / nprof;
    return PP_OK;
}

// ---- the reference's initial phase guess, data side fused --------------------
extern "C" int pp_reference_phase_seed(pp_ctx* c, const void* src, int dtype, int on_device, int nsub, int nchan,
                                       int nbin, const double* freqs, int64_t freqs_stride, const double* P,
                                       const double* par3, double nu_DM, double nu_GM, const double* weights,
                                       const double* model_profs, double lo, double hi, int Ns, double* out7) {
    if (int busy_ = ctx_busy(c, "pp_reference_phase_seed")) return busy_;
    if (!c || !src || !freqs || !P || !par3 || !weights || !model_profs || !out7)
        return fail(PP_EINVAL, "pp_reference_phase_seed: null argument");
    if (!nbin_any_ok(nbin)) return nbin_refuse("pp_reference_phase_seed", nbin);
    const bool anyb = !nbin_ok(nbin);
    if (nsub < 1 || nchan < 1 || Ns < 1) return fail(PP_EINVAL, "pp_reference_phase_seed: bad shape");
    if (dtype != PP_F64 && dtype != PP_F32) return fail(PP_EINVAL, "pp_reference_phase_seed: dtype %d", dtype);
    if (freqs_stride != 0 && freqs_stride != nchan) return fail(PP_EINVAL, "freqs_stride must be 0 or nchan");
    HIP_TRY(hipSetDevice(c->device));
    const int M = nbin / 2;
    const size_t esz = dtype == PP_F64 ? 8 : 4;
    const size_t sub_b = (size_t)nchan * nbin * esz;
    int rc;
    if (!on_device) {
        const int cap = aux_chunk_cap(c, (double)sub_b + 64.0 * nchan + 64.0 * nbin, nsub);
        if (nsub > cap) {
            for (int s0 = 0; s0 < nsub; s0 += cap) {
                const int n = std::min(cap, nsub - s0);
                if ((rc = pp_reference_phase_seed(c, (const char*)src + s0 * sub_b, dtype, 0, n, nchan, nbin,
                                                  freqs + (freqs_stride ? (size_t)s0 * nchan : 0), freqs_stride, P + s0,
                                                  par3 + (size_t)s0 * 3, nu_DM, nu_GM, weights + (size_t)s0 * nchan,
                                                  model_profs + (size_t)s0 * nbin, lo, hi, Ns, out7 + (size_t)s0 * 7)))
                    return rc;
            }
            return PP_OK;
        }
    }
    const void* dsrc = src;
    if (!on_device) {
        if ((rc = c->data.reserve((size_t)nsub * sub_b))) return rc;
        HIP_TRY(hipMemcpyAsync(c->data.p, src, (size_t)nsub * sub_b, hipMemcpyHostToDevice, c->stream));
        dsrc = c->data.p;
    }
    if ((rc = upload(c, c->freqs, freqs, (size_t)(freqs_stride ? (size_t)nsub * nchan : nchan) * 8))) return rc;
    if ((rc = upload(c, c->P, P, (size_t)nsub * 8))) return rc;
    if ((rc = upload(c, c->x0, par3, (size_t)nsub * 24))) return rc;
    if ((rc = upload(c, c->wts, weights, (size_t)nsub * nchan * 8))) return rc;
    if ((rc = upload(c, c->errs, model_profs, (size_t)nsub * nbin * 8))) return rc;
    // runs of channels per subint: one partial spectrum per run, the runs added in a fixed order.  The run
    // length is a function of the band alone (an eighth of it, 16 ... 256 channels), never of the number of
    // subints in the call: the reference's guess for a subint does not depend on its neighbours (pptoas.py:421-457)
    const int cpr = std::max(16, std::min(256, (((nchan + 7) / 8) + 15) / 16 * 16));
    int nrun = (nchan + cpr - 1) / cpr;
    const size_t H = (size_t)M + 1;
    // (general row lengths: the harmonics of every row are written out first -- nrun = nchan slots of H)
    if (anyb) nrun = nchan;
    // X: [nsub][nrun][H] partial spectra | [nsub][H] data spectra | [nsub][H] model spectra | [nsub][M] k_fps work
    if ((rc = c->X.reserve(((size_t)nsub * nrun * H + 2 * (size_t)nsub * H + (size_t)nsub * M) * sizeof(cplx)))) return rc;
    if ((rc = c->sdraw.reserve((size_t)nsub * nrun * 8))) return rc;
    if ((rc = c->o_params.reserve((size_t)nsub * 56))) return rc;
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, nbin, &tw))) return rc;
    cplx* part = c->X.as<cplx>();
    cplx* dspec = part + (size_t)nsub * nrun * H;
    cplx* mspec = dspec + (size_t)nsub * H;
    cplx* xwork = mspec + (size_t)nsub * H;
    RotMeanArgs ra{dsrc, c->freqs.as<double>(), (long long)freqs_stride, c->P.as<double>(), c->x0.as<double>(),
                   c->wts.as<double>(), tw, std::isinf(nu_DM) ? 0.0 : 1.0 / (nu_DM * nu_DM),
                   std::isinf(nu_GM) ? 0.0 : 1.0 / (nu_GM * nu_GM * nu_GM * nu_GM), part, c->sdraw.as<double>(),
                   nsub, nchan, nrun, cpr};
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (anyb) {
        // row lengths without a tuned plan (pp_anybin.h): every row's harmonics by the chirp-z path, then
        // the weighted, rotated channel sum per harmonic, and the template profiles' spectra the same way
        Prof pr(c, KF_FPS);
        XspecArgs xa;
        memset(&xa, 0, sizeof xa);
        xa.data = dsrc; xa.nsub = nsub; xa.nchan = nchan; xa.nchan_full = nchan; xa.cstep = 1;
        const int Mp = ((M + 63) / 64) * 64;
        if ((rc = launch_any(c, xa, nbin, Mp, dtype, -1, false, part, nullptr))) return rc;
        hipLaunchKernelGGL(k_rot_mean_harm, dim3((unsigned)((M + 1 + 255) / 256), nsub), dim3(256), 0, c->stream,
                           (const cplx*)part, ra, M, dspec);
        memset(&xa, 0, sizeof xa);
        xa.data = c->errs.p; xa.nsub = 1; xa.nchan = nsub; xa.nchan_full = nsub; xa.cstep = 1;
        if ((rc = launch_any(c, xa, nbin, Mp, PP_F64, -1, false, mspec, nullptr))) return rc;
        FpsArgs fa{dspec, nullptr, c->o_params.as<double>(), lo, hi, Ns, M, nsub, c->fps_finish, mspec, M + 1};
        hipLaunchKernelGGL(k_fps, dim3(nsub), dim3(256), 0, c->stream, fa, xwork);
    } else {
        Prof pr(c, KF_FPS);
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            if (MM == 1024 && c->one_exchange) {
                // 2048-bin rows: the one-exchange transform (k_rot_mean_q1024)
                if (dtype == PP_F64) hipLaunchKernelGGL((k_rot_mean_q1024<double>), dim3(nsub * nrun), dim3(64), 0, c->stream, ra);
                else hipLaunchKernelGGL((k_rot_mean_q1024<float>), dim3(nsub * nrun), dim3(64), 0, c->stream, ra);
            } else if (dtype == PP_F64) hipLaunchKernelGGL((k_rot_mean<MM, double>), dim3(nsub * nrun), dim3(T), 0, c->stream, ra);
            else hipLaunchKernelGGL((k_rot_mean<MM, float>), dim3(nsub * nrun), dim3(T), 0, c->stream, ra);
            hipLaunchKernelGGL(k_rot_mean_finish, dim3((M + 1 + 255) / 256, nsub), dim3(256), 0, c->stream,
                               (const cplx*)part, (const double*)c->sdraw.as<double>(), nsub, nrun, M, dspec);
            hipLaunchKernelGGL((k_rfft_rows<MM, double>), dim3(fft_grid(T, nsub)), dim3(T), 0, c->stream,
                               (const void*)c->errs.p, mspec, tw, nsub);
        });
        FpsArgs fa{dspec, nullptr, c->o_params.as<double>(), lo, hi, Ns, M, nsub, c->fps_finish, mspec, M + 1};
        hipLaunchKernelGGL(k_fps, dim3(nsub), dim3(256), 0, c->stream, fa, xwork);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipMemcpyAsync(out7, c->o_params.p, (size_t)nsub * 56, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    for (int i = 0; i < nsub; ++i) out7[(size_t)i * 7 + 6] = 1e-3 * ms / nsub;
    return PP_OK;
}

// ---- general row lengths (pp_anybin.h): harmonics out, harmonics back ---------------------
// the Bluestein tables of a row length as the kernels take them
static int any_args(pp_ctx* c, int nbin, AnyArgs* g, int* L) {
    pp_ctx::AnyPlan* pl = nullptr;
    int rc;
    if ((rc = get_any_plan(c, nbin, &pl))) return rc;
    const cplx *twL = nullptr, *twB = nullptr;
    if ((rc = get_twiddles(c, 2 * pl->L, &twL))) return rc;
    if ((rc = get_twiddles(c, nbin, &twB))) return rc;
    const int M = nbin / 2;
    *g = AnyArgs{nbin, M, ((M + 63) / 64) * 64, pl->chirp.as<cplx>(), pl->bft.as<cplx>(), twL, twB, 0, 0, nullptr, nullptr};
    *L = pl->L;
    return PP_OK;
}
// harmonics 0..M of nrows_sub x nrows_chan rows of `rows` (row-major [nsub][nchan][nbin]) into hout, CHANNEL-major
// (hout row = n nsub + i), by k_any
static int harmonics_any(pp_ctx* c, const void* rows, int dtype, int nsub, int nchan, int nbin, cplx* hout) {
    XspecArgs xa;
    memset(&xa, 0, sizeof xa);
    xa.data = rows; xa.nsub = nsub; xa.nchan = nchan; xa.nchan_full = nchan; xa.cstep = 1;
    return launch_any(c, xa, nbin, ((nbin / 2 + 63) / 64) * 64, dtype, -1, false, hout, nullptr);
}
// numpy.fft.irfft of nrows rows of M + 1 harmonics -> out[nrows][nbin] (device pointers)
static int irfft_any(pp_ctx* c, int nbin, const cplx* harm, long long nrows, double* out) {
    AnyArgs g;
    int L = 0, rc;
    if ((rc = any_args(c, nbin, &g, &L))) return rc;
    const int grid = (int)std::max(1LL, std::min(nrows, 2048LL));
    switch (L) {
        case 64: hipLaunchKernelGGL((k_irfft_any<64>), dim3(grid), dim3(FftPlan<64>::T), 0, c->stream, harm, g, nrows, out); break;
        case 256: hipLaunchKernelGGL((k_irfft_any<256>), dim3(grid), dim3(FftPlan<256>::T), 0, c->stream, harm, g, nrows, out); break;
        case 1024: hipLaunchKernelGGL((k_irfft_any<1024>), dim3(grid), dim3(FftPlan<1024>::T), 0, c->stream, harm, g, nrows, out); break;
        case 4096: hipLaunchKernelGGL((k_irfft_any<4096>), dim3(grid), dim3(FftPlan<4096>::T), 0, c->stream, harm, g, nrows, out); break;
        default: return fail(PP_EINVAL, "no transform of %d points", L);
    }
    HIP_TRY(hipGetLastError());
    return PP_OK;
}

extern "C" int pp_synth_portraits(pp_ctx* c, int slot, void* dst, int dtype, int nsub, const double* freqs,
                                  const double* P, const double* inj, const double* gains, double sigma,
                                  uint64_t seed, int64_t first_subint) {
    if (int busy_ = ctx_busy(c, "pp_synth_portraits")) return busy_;
    if (!c || !dst || !freqs || !P || !inj) return fail(PP_EINVAL, "pp_synth_portraits: null argument");
    if (slot < 0 || slot >= PP_MAX_SLOTS || !c->slots[slot].set) return fail(PP_ESTATE, "pp_synth_portraits: slot %d not set", slot);
    if (dtype != PP_F64 && dtype != PP_F32) return fail(PP_EINVAL, "pp_synth_portraits: dtype %d", dtype);
    if (nsub < 1) return fail(PP_EINVAL, "pp_synth_portraits: nsub %d", nsub);
    HIP_TRY(hipSetDevice(c->device));
    ModelSlot& s = c->slots[slot];
    const int C = s.nchan, B = s.nbin, M = B / 2;
    int rc;
    if ((rc = upload(c, c->freqs, freqs, (size_t)C * 8))) return rc;
    if ((rc = upload(c, c->P, P, (size_t)nsub * 8))) return rc;
    if ((rc = upload(c, c->x0, inj, (size_t)nsub * 24))) return rc;
    if (gains) if ((rc = upload(c, c->errs, gains, (size_t)nsub * C * 8))) return rc;
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, B, &tw))) return rc;
    SynthArgs a{s.mft.as<cplx>(), s.mdc.as<double>(), dst, c->freqs.as<double>(), c->P.as<double>(),
                c->x0.as<double>(), tw, sigma, seed, first_subint, nsub, C,
                gains ? c->errs.as<double>() : (const double*)nullptr};
    if (!nbin_ok(B)) {
        // a row length without a tuned plan: the rotated template's harmonics back by the chirp-z route
        Prof pr(c, KF_SYNTH);
        AnyArgs g;
        int L = 0;
        if ((rc = any_args(c, B, &g, &L))) return rc;
        const int grid = (int)std::max(1LL, std::min((long long)nsub * C, 2048LL));
#define PP_SYN_ANY(LL)                                                                                          \
    do {                                                                                                        \
        if (dtype == PP_F64) hipLaunchKernelGGL((k_synth_any<LL, double>), dim3(grid), dim3(FftPlan<LL>::T), 0, c->stream, a, g); \
        else hipLaunchKernelGGL((k_synth_any<LL, float>), dim3(grid), dim3(FftPlan<LL>::T), 0, c->stream, a, g); \
    } while (0)
        switch (L) {
            case 64: PP_SYN_ANY(64); break;
            case 256: PP_SYN_ANY(256); break;
            case 1024: PP_SYN_ANY(1024); break;
            case 4096: PP_SYN_ANY(4096); break;
            default: return fail(PP_EINVAL, "no transform of %d points", L);
        }
#undef PP_SYN_ANY
    } else {
        Prof pr(c, KF_SYNTH);
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            const int grid = fft_grid(T, (long long)nsub * C);
            if (dtype == PP_F64) hipLaunchKernelGGL((k_synth<MM, double>), dim3(grid), dim3(T), 0, c->stream, a);
            else hipLaunchKernelGGL((k_synth<MM, float>), dim3(grid), dim3(T), 0, c->stream, a);
        });
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

extern "C" int pp_rotate_portraits(pp_ctx* c, const void* src, void* dst, int dtype, int on_device, int nsub,
                                   int nchan, int nbin, const double* freqs, int64_t freqs_stride,
                                   const double* P, const double* par3, double nu_DM, double nu_GM) {
    if (int busy_ = ctx_busy(c, "pp_rotate_portraits")) return busy_;
    if (!c || !src || !dst || !freqs || !P || !par3) return fail(PP_EINVAL, "pp_rotate_portraits: null argument");
    if (!nbin_any_ok(nbin)) return nbin_refuse("pp_rotate_portraits", nbin);
    if (nsub < 1 || nchan < 1) return fail(PP_EINVAL, "pp_rotate_portraits: bad shape");
    if (dtype != PP_F64 && dtype != PP_F32) return fail(PP_EINVAL, "pp_rotate_portraits: dtype %d", dtype);
    if (freqs_stride != 0 && freqs_stride != nchan) return fail(PP_EINVAL, "freqs_stride must be 0 or nchan");
    HIP_TRY(hipSetDevice(c->device));
    const int M = nbin / 2;
    const size_t esz = dtype == PP_F64 ? 8 : 4;
    const size_t bytes = (size_t)nsub * nchan * nbin * esz;
    int rc;
    if (!on_device) {
        const size_t sub_b = (size_t)nchan * nbin * esz;
        const int cap = aux_chunk_cap(c, (double)sub_b + 64.0 * nchan, nsub);
        if (nsub > cap) {
            for (int s0 = 0; s0 < nsub; s0 += cap) {
                const int n = std::min(cap, nsub - s0);
                if ((rc = pp_rotate_portraits(c, (const char*)src + s0 * sub_b, (char*)dst + s0 * sub_b, dtype, 0, n, nchan, nbin,
                                              freqs + (freqs_stride ? (size_t)s0 * nchan : 0), freqs_stride, P + s0,
                                              par3 + (size_t)s0 * 3, nu_DM, nu_GM)))
                    return rc;
            }
            return PP_OK;
        }
    }
    const void* dsrc = src;
    void* ddst = dst;
    if (!on_device) {
        if ((rc = c->data.reserve(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(c->data.p, src, bytes, hipMemcpyHostToDevice, c->stream));
        dsrc = c->data.p; ddst = c->data.p;
    }
    if ((rc = upload(c, c->freqs, freqs, (size_t)(freqs_stride ? (size_t)nsub * nchan : nchan) * 8))) return rc;
    if ((rc = upload(c, c->P, P, (size_t)nsub * 8))) return rc;
    if ((rc = upload(c, c->x0, par3, (size_t)nsub * 24))) return rc;
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, nbin, &tw))) return rc;
    RotateArgs a{dsrc, ddst, c->freqs.as<double>(), (long long)freqs_stride, c->P.as<double>(), c->x0.as<double>(), tw,
                 std::isinf(nu_DM) ? 0.0 : 1.0 / (nu_DM * nu_DM),
                 std::isinf(nu_GM) ? 0.0 : 1.0 / (nu_GM * nu_GM * nu_GM * nu_GM), nsub, nchan};
    if (!nbin_ok(nbin)) {
        // a row length without a tuned plan: the chirp-z route, forward and back (pp_anybin.h)
        Prof pr(c, KF_SYNTH);
        pp_ctx::AnyPlan* pl = nullptr;
        if ((rc = get_any_plan(c, nbin, &pl))) return rc;
        const cplx* twL = nullptr;
        if ((rc = get_twiddles(c, 2 * pl->L, &twL))) return rc;
        AnyArgs g{nbin, M, ((M + 63) / 64) * 64, pl->chirp.as<cplx>(), pl->bft.as<cplx>(), twL, tw, 0, 0, nullptr, nullptr};
        const int grid = (int)std::max(1LL, std::min((long long)nsub * nchan, 2048LL));
#define PP_ROT_ANY(LL)                                                                                          \
    do {                                                                                                        \
        if (dtype == PP_F64) hipLaunchKernelGGL((k_rotate_any<LL, double>), dim3(grid), dim3(FftPlan<LL>::T), 0, c->stream, a, g); \
        else hipLaunchKernelGGL((k_rotate_any<LL, float>), dim3(grid), dim3(FftPlan<LL>::T), 0, c->stream, a, g); \
    } while (0)
        switch (pl->L) {
            case 64: PP_ROT_ANY(64); break;
            case 256: PP_ROT_ANY(256); break;
            case 1024: PP_ROT_ANY(1024); break;
            case 4096: PP_ROT_ANY(4096); break;
            default: return fail(PP_EINVAL, "no transform of %d points", pl->L);
        }
#undef PP_ROT_ANY
    } else {
        Prof pr(c, KF_SYNTH);
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            const int grid = fft_grid(T, (long long)nsub * nchan);
            if (dtype == PP_F64) hipLaunchKernelGGL((k_rotate<MM, double>), dim3(grid), dim3(T), 0, c->stream, a);
            else hipLaunchKernelGGL((k_rotate<MM, float>), dim3(grid), dim3(T), 0, c->stream, a);
        });
    }
    HIP_TRY(hipGetLastError());
    if (!on_device) HIP_TRY(hipMemcpyAsync(dst, c->data.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

// ---- ppalign accumulation -------------------------------------------------
extern "C" int pp_align_accumulate(pp_ctx* c, const void* src, int dtype, int on_device, int nsub, int nchan,
                                   int nbin, const double* freqs, int64_t freqs_stride, const double* P,
                                   const double* par3, const double* weights, double* aligned,
                                   double* total_weights) {
    if (int busy_ = ctx_busy(c, "pp_align_accumulate")) return busy_;
    if (!c || !src || !freqs || !P || !par3 || !weights || !aligned || !total_weights)
        return fail(PP_EINVAL, "pp_align_accumulate: null argument");
    if (!nbin_any_ok(nbin)) return nbin_refuse("pp_align_accumulate", nbin);
    if (nsub < 1 || nchan < 1) return fail(PP_EINVAL, "pp_align_accumulate: bad shape");
    if (dtype != PP_F64 && dtype != PP_F32) return fail(PP_EINVAL, "pp_align_accumulate: dtype %d", dtype);
    if (freqs_stride != 0 && freqs_stride != nchan) return fail(PP_EINVAL, "freqs_stride must be 0 or nchan");
    HIP_TRY(hipSetDevice(c->device));
    const int M = nbin / 2;
    const size_t esz = dtype == PP_F64 ? 8 : 4;
    const size_t bytes = (size_t)nsub * nchan * nbin * esz;
    int rc;
    if (!on_device) {
        const size_t sub_b = (size_t)nchan * nbin * esz;
        const int cap = aux_chunk_cap(c, (double)sub_b + 64.0 * nchan, nsub);
        if (nsub > cap) {
            // the sums of the runs are added on the host
            std::vector<double> part((size_t)nchan * nbin), wpart((size_t)nchan);
            std::fill(aligned, aligned + (size_t)nchan * nbin, 0.0);
            std::fill(total_weights, total_weights + nchan, 0.0);
            for (int s0 = 0; s0 < nsub; s0 += cap) {
                const int n = std::min(cap, nsub - s0);
                if ((rc = pp_align_accumulate(c, (const char*)src + s0 * sub_b, dtype, 0, n, nchan, nbin,
                                              freqs + (freqs_stride ? (size_t)s0 * nchan : 0), freqs_stride, P + s0,
                                              par3 + (size_t)s0 * 3, weights + (size_t)s0 * nchan, part.data(), wpart.data())))
                    return rc;
                for (size_t j = 0; j < part.size(); ++j) aligned[j] += part[j];
                for (int j = 0; j < nchan; ++j) total_weights[j] += wpart[j];
            }
            return PP_OK;
        }
    }
    const void* dsrc = src;
    if (!on_device) {
        if ((rc = c->data.reserve(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(c->data.p, src, bytes, hipMemcpyHostToDevice, c->stream));
        dsrc = c->data.p;
    }
    if ((rc = upload(c, c->freqs, freqs, (size_t)(freqs_stride ? (size_t)nsub * nchan : nchan) * 8))) return rc;
    if ((rc = upload(c, c->P, P, (size_t)nsub * 8))) return rc;
    if ((rc = upload(c, c->x0, par3, (size_t)nsub * 24))) return rc;
    if ((rc = upload(c, c->wts, weights, (size_t)nsub * nchan * 8))) return rc;
    if ((rc = c->X.reserve((size_t)nchan * nbin * 8))) return rc;          // aligned portrait
    if ((rc = c->sdraw.reserve((size_t)nchan * 8))) return rc;             // total weights
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, nbin, &tw))) return rc;
    AlignArgs a{dsrc, c->freqs.as<double>(), (long long)freqs_stride, c->P.as<double>(), c->x0.as<double>(),
                c->wts.as<double>(), tw, c->X.as<double>(), c->sdraw.as<double>(), nsub, nchan};
    if (!nbin_ok(nbin)) {
        // a row length without a tuned plan (pp_anybin.h): harmonics of every row by the chirp-z route, the weighted,
        // rotated sum per (channel, harmonic) over the subints in index order, ONE inverse transform per channel.
        // Subints go through in chunks whose harmonics fit a 2 GB scratch buffer.
        Prof pr(c, KF_SYNTH);
        const size_t H = (size_t)M + 1;
        const int cs = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsub, ((size_t)2 << 30) / ((size_t)nchan * H * sizeof(cplx))));
        if ((rc = c->seedbuf.reserve(((size_t)cs * nchan * H + (size_t)nchan * H) * sizeof(cplx)))) return rc;
        cplx* hout = c->seedbuf.as<cplx>();
        cplx* spec = hout + (size_t)cs * nchan * H;
        for (int s0 = 0; s0 < nsub; s0 += cs) {
            const int ns = std::min(cs, nsub - s0);
            if ((rc = harmonics_any(c, (const char*)dsrc + (size_t)s0 * nchan * nbin * esz, dtype, ns, nchan, nbin, hout))) return rc;
            hipLaunchKernelGGL(k_align_harm, dim3((unsigned)((H + 255) / 256), nchan), dim3(256), 0, c->stream,
                               (const cplx*)hout, a, s0, ns, M, spec, c->sdraw.as<double>(), s0 == 0 ? 1 : 0);
            HIP_TRY(hipGetLastError());
        }
        if ((rc = irfft_any(c, nbin, spec, nchan, c->X.as<double>()))) return rc;
    } else {
        Prof pr(c, KF_SYNTH);
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            if (dtype == PP_F64) hipLaunchKernelGGL((k_align_accum<MM, double>), dim3(nchan), dim3(T), 0, c->stream, a);
            else hipLaunchKernelGGL((k_align_accum<MM, float>), dim3(nchan), dim3(T), 0, c->stream, a);
        });
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(aligned, c->X.p, (size_t)nchan * nbin * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(total_weights, c->sdraw.p, (size_t)nchan * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

// ---- per-channel reduced chi^2 of fitted subints ---------------------------
extern "C" int pp_channel_red_chi2(pp_ctx* c, const void* src, int dtype, int on_device, int nsub, int nchan,
                                   int nbin, const int32_t* model_slot, const double* freqs,
                                   int64_t freqs_stride, const double* P, const double* params5,
                                   const double* nu_refs3, const double* scales, const double* errs,
                                   double* red_chi2) {
    if (int busy_ = ctx_busy(c, "pp_channel_red_chi2")) return busy_;
    if (!c || !src || !freqs || !P || !params5 || !nu_refs3 || !scales || !errs || !red_chi2)
        return fail(PP_EINVAL, "pp_channel_red_chi2: null argument");
    if (!nbin_any_ok(nbin)) return nbin_refuse("pp_channel_red_chi2", nbin);
    if (nsub < 1 || nchan < 1) return fail(PP_EINVAL, "pp_channel_red_chi2: bad shape");
    if (dtype != PP_F64 && dtype != PP_F32) return fail(PP_EINVAL, "pp_channel_red_chi2: dtype %d", dtype);
    if (freqs_stride != 0 && freqs_stride != nchan) return fail(PP_EINVAL, "freqs_stride must be 0 or nchan");
    HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < nsub; ++i) {
        const int sl = model_slot ? model_slot[i] : 0;
        if (sl < 0 || sl >= PP_MAX_SLOTS || !c->slots[sl].set || c->slots[sl].nchan != nchan ||
            c->slots[sl].nbin != nbin)
            return fail(PP_EINVAL, "pp_channel_red_chi2: model slot %d is not a %d x %d template", sl, nchan, nbin);
    }
    const int M = nbin / 2;
    const size_t esz = dtype == PP_F64 ? 8 : 4;
    const size_t bytes = (size_t)nsub * nchan * nbin * esz;
    const size_t nc = (size_t)nsub * nchan;
    int rc;
    if (!on_device) {
        const size_t sub_b = (size_t)nchan * nbin * esz;
        const int cap = aux_chunk_cap(c, (double)sub_b + 64.0 * nchan, nsub);
        if (nsub > cap) {
            for (int s0 = 0; s0 < nsub; s0 += cap) {
                const int n = std::min(cap, nsub - s0);
                const size_t o = (size_t)s0 * nchan;
                if ((rc = pp_channel_red_chi2(c, (const char*)src + s0 * sub_b, dtype, 0, n, nchan, nbin,
                                              model_slot ? model_slot + s0 : nullptr, freqs + (freqs_stride ? o : 0),
                                              freqs_stride, P + s0, params5 + (size_t)s0 * 5, nu_refs3 + (size_t)s0 * 3,
                                              scales + o, errs + o, red_chi2 + o)))
                    return rc;
            }
            return PP_OK;
        }
    }
    const void* dsrc = src;
    if (!on_device) {
        if ((rc = c->data.reserve(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(c->data.p, src, bytes, hipMemcpyHostToDevice, c->stream));
        dsrc = c->data.p;
    }
    if ((rc = upload(c, c->freqs, freqs, (size_t)(freqs_stride ? nc : (size_t)nchan) * 8))) return rc;
    if ((rc = upload(c, c->P, P, (size_t)nsub * 8))) return rc;
    if ((rc = upload(c, c->x0, params5, (size_t)nsub * 40))) return rc;
    if ((rc = upload(c, c->nufit, nu_refs3, (size_t)nsub * 24))) return rc;
    if ((rc = upload(c, c->wts, scales, nc * 8))) return rc;
    if ((rc = upload(c, c->errs, errs, nc * 8))) return rc;
    if (model_slot) if ((rc = upload(c, c->slot, model_slot, (size_t)nsub * 4))) return rc;
    if ((rc = c->sdraw.reserve(nc * 8))) return rc;
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, nbin, &tw))) return rc;
    ChanChi2Args a{dsrc, (const cplx* const*)c->mft_table.p, (const double* const*)c->mdc_table.p,
                   model_slot ? c->slot.as<int>() : nullptr, c->freqs.as<double>(), (long long)freqs_stride,
                   c->P.as<double>(), c->x0.as<double>(), c->nufit.as<double>(), c->wts.as<double>(),
                   c->errs.as<double>(), tw, c->sdraw.as<double>(), nsub, nchan};
    if (!nbin_ok(nbin)) {
        // a row length without a tuned plan: the data rows' harmonics by the chirp-z route, then Parseval on the
        // residual spectrum exactly as k_chan_chi2 forms it (the slot's spectrum rows are pitched to Mp)
        Prof pr(c, KF_FINAL);
        const size_t H = (size_t)M + 1;
        const int Mp = c->slots[model_slot ? model_slot[0] : 0].Mp;
        const int cs = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsub, ((size_t)2 << 30) / ((size_t)nchan * H * sizeof(cplx))));
        if ((rc = c->seedbuf.reserve((size_t)cs * nchan * H * sizeof(cplx)))) return rc;
        cplx* hout = c->seedbuf.as<cplx>();
        for (int s0 = 0; s0 < nsub; s0 += cs) {
            const int ns = std::min(cs, nsub - s0);
            if ((rc = harmonics_any(c, (const char*)dsrc + (size_t)s0 * nchan * nbin * esz, dtype, ns, nchan, nbin, hout))) return rc;
            const int grid = (int)std::max(1LL, std::min((long long)ns * nchan, 4096LL));
            hipLaunchKernelGGL(k_chan_chi2_harm, dim3(grid), dim3(256), 0, c->stream, (const cplx*)hout, a, s0, ns, M, Mp);
            HIP_TRY(hipGetLastError());
        }
    } else {
        Prof pr(c, KF_FINAL);
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            const int grid = fft_grid(T, (long long)nc);
            if (dtype == PP_F64) hipLaunchKernelGGL((k_chan_chi2<MM, double>), dim3(grid), dim3(T), 0, c->stream, a);
            else hipLaunchKernelGGL((k_chan_chi2<MM, float>), dim3(grid), dim3(T), 0, c->stream, a);
        });
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(red_chi2, c->sdraw.p, nc * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

// ---- Gaussian-component templates on the device ------------------------------
static int gauss_generate(pp_ctx* c, int nchan, int nbin, const double* freqs, const char* code, double nu_ref,
                          double dc, double tau_rot, double alpha, int ngauss, const double* comps,
                          double* dev_out) {
    if (!freqs || !code || !comps) return fail(PP_EINVAL, "gaussian portrait: null argument");
    if (!nbin_any_ok(nbin)) return nbin_refuse("gaussian portrait", nbin);
    if (nchan < 1) return fail(PP_EINVAL, "gaussian portrait: bad shape");
    if (ngauss < 1 || ngauss > PP_MAX_GAUSS) return fail(PP_EINVAL, "gaussian portrait: 1..%d components", PP_MAX_GAUSS);
    for (int j = 0; j < 3; ++j)
        if (code[j] != '0' && code[j] != '1') return fail(PP_EINVAL, "gaussian portrait: model code '%.3s'", code);
    int rc;
    if ((rc = upload(c, c->freqs, freqs, (size_t)nchan * 8))) return rc;
    if ((rc = upload(c, c->misc, comps, (size_t)ngauss * 48))) return rc;
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, nbin, &tw))) return rc;
    GaussArgs a{c->freqs.as<double>(), c->misc.as<double>(), tw, dev_out, nu_ref, dc, tau_rot, alpha, nchan, ngauss,
                code[0] - '0', code[1] - '0', code[2] - '0'};
    const int M = nbin / 2;
    if (!nbin_ok(nbin)) {
        // a row length without a tuned plan: the rows need no transform; a scattered model's filter
        // 1 / (1 + 2 pi i k tau_n) goes through the harmonics (chirp-z route there and back)
        Prof pr(c, KF_MODEL);
        hipLaunchKernelGGL(k_gauss_rows, dim3(nchan), dim3(256), 0, c->stream, a, nbin);
        HIP_TRY(hipGetLastError());
        if (tau_rot != 0.0) {
            const size_t H = (size_t)M + 1;
            if ((rc = c->seedbuf.reserve((size_t)nchan * H * sizeof(cplx)))) return rc;
            cplx* harm = c->seedbuf.as<cplx>();
            if ((rc = harmonics_any(c, dev_out, PP_F64, 1, nchan, nbin, harm))) return rc;
            hipLaunchKernelGGL(k_scatter_harm, dim3((unsigned)((H + 255) / 256), nchan), dim3(256), 0, c->stream, harm,
                               (const double*)c->freqs.as<double>(), nu_ref, tau_rot, alpha, M);
            HIP_TRY(hipGetLastError());
            if ((rc = irfft_any(c, nbin, harm, nchan, dev_out))) return rc;
        }
        return PP_OK;
    }
    {
        Prof pr(c, KF_MODEL);
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            hipLaunchKernelGGL((k_gauss_portrait<MM>), dim3(nchan), dim3(T), 0, c->stream, a);
        });
    }
    HIP_TRY(hipGetLastError());
    return PP_OK;
}

extern "C" int pp_gaussian_portrait(pp_ctx* c, int nchan, int nbin, const double* freqs, const char* code,
                                    double nu_ref, double dc, double tau_rot, double alpha, int ngauss,
                                    const double* comps, double* portrait, int out_on_device) {
    if (int busy_ = ctx_busy(c, "pp_gaussian_portrait")) return busy_;
    if (!c || !portrait) return fail(PP_EINVAL, "pp_gaussian_portrait: null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    double* dout = portrait;
    if (!out_on_device) {
        if ((rc = c->data.reserve((size_t)nchan * nbin * 8))) return rc;
        dout = c->data.as<double>();
    }
    if ((rc = gauss_generate(c, nchan, nbin, freqs, code, nu_ref, dc, tau_rot, alpha, ngauss, comps, dout))) return rc;
    if (!out_on_device)
        HIP_TRY(hipMemcpyAsync(portrait, dout, (size_t)nchan * nbin * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

extern "C" int pp_model_set_gaussian(pp_ctx* c, int slot, int nchan, int nbin, const double* freqs,
                                     const char* code, double nu_ref, double dc, double tau_rot, double alpha,
                                     int ngauss, const double* comps) {
    if (int busy_ = ctx_busy(c, "pp_model_set_gaussian")) return busy_;
    if (!c) return fail(PP_EINVAL, "pp_model_set_gaussian: null context");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    // scratch for the portrait (general row lengths: pp_model_set leaves the rows' harmonics in c->X)
    DevBuf& scratch = nbin_ok(nbin) ? c->X : c->data;
    if ((rc = scratch.reserve((size_t)nchan * nbin * 8))) return rc;
    if ((rc = gauss_generate(c, nchan, nbin, freqs, code, nu_ref, dc, tau_rot, alpha, ngauss, comps,
                             scratch.as<double>())))
        return rc;
    return pp_model_set(c, slot, scratch.p, PP_F64, 1, nchan, nbin);
}

// ---- spline (PCA + B-spline) templates on the device --------------------------
static int spline_generate(pp_ctx* c, int nchan, int nbin, const double* freqs, int ncomp, const double* basis,
                           int nknots, const double* t, const double* coefs, int degree, double* dev_out) {
    if (!freqs || !basis || (ncomp > 0 && (!t || !coefs))) return fail(PP_EINVAL, "spline portrait: null argument");
    if (nbin < 2 || nchan < 1) return fail(PP_EINVAL, "spline portrait: bad shape");
    if (ncomp < 0 || ncomp > PP_MAX_SPLINE_COMP) return fail(PP_EINVAL, "spline portrait: 0..%d components", PP_MAX_SPLINE_COMP);
    if (ncomp > 0 && (degree < 1 || degree > PP_MAX_SPLINE_DEG || nknots < 2 * (degree + 1)))
        return fail(PP_EINVAL, "spline portrait: degree %d with %d knots", degree, nknots);
    int rc;
    if ((rc = upload(c, c->freqs, freqs, (size_t)nchan * 8))) return rc;
    const size_t nb = (size_t)(ncomp + 1) * nbin, nt = (size_t)std::max(nknots, 1), ncf = (size_t)std::max(ncomp, 1) * nt;
    if ((rc = c->seedbuf.reserve((nb + nt + ncf) * 8))) return rc;
    double* dbasis = c->seedbuf.as<double>();
    double* dt = dbasis + nb;
    double* dc = dt + nt;
    HIP_TRY(hipMemcpyAsync(dbasis, basis, nb * 8, hipMemcpyHostToDevice, c->stream));
    if (ncomp > 0) {
        HIP_TRY(hipMemcpyAsync(dt, t, (size_t)nknots * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(dc, coefs, (size_t)ncomp * nknots * 8, hipMemcpyHostToDevice, c->stream));
    }
    SplineArgs a{c->freqs.as<double>(), dbasis, dt, dc, dev_out, nchan, nbin, ncomp, nknots, degree};
    {
        Prof pr(c, KF_MODEL);
        hipLaunchKernelGGL(k_spline_portrait, dim3(nchan), dim3(256), 0, c->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return PP_OK;
}

extern "C" int pp_spline_portrait(pp_ctx* c, int nchan, int nbin, const double* freqs, int ncomp,
                                  const double* basis, int nknots, const double* t, const double* coefs,
                                  int degree, double* portrait, int out_on_device) {
    if (int busy_ = ctx_busy(c, "pp_spline_portrait")) return busy_;
    if (!c || !portrait) return fail(PP_EINVAL, "pp_spline_portrait: null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    double* dout = portrait;
    if (!out_on_device) {
        if ((rc = c->data.reserve((size_t)nchan * nbin * 8))) return rc;
        dout = c->data.as<double>();
    }
    if ((rc = spline_generate(c, nchan, nbin, freqs, ncomp, basis, nknots, t, coefs, degree, dout))) return rc;
    if (!out_on_device)
        HIP_TRY(hipMemcpyAsync(portrait, dout, (size_t)nchan * nbin * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

extern "C" int pp_model_set_spline(pp_ctx* c, int slot, int nchan, int nbin, const double* freqs, int ncomp,
                                   const double* basis, int nknots, const double* t, const double* coefs,
                                   int degree) {
    if (int busy_ = ctx_busy(c, "pp_model_set_spline")) return busy_;
    if (!c) return fail(PP_EINVAL, "pp_model_set_spline: null context");
    if (!nbin_any_ok(nbin)) return nbin_refuse("pp_model_set_spline", nbin);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    DevBuf& scratch = nbin_ok(nbin) ? c->X : c->data;      // (as pp_model_set_gaussian)
    if ((rc = scratch.reserve((size_t)nchan * nbin * 8))) return rc;
    if ((rc = spline_generate(c, nchan, nbin, freqs, ncomp, basis, nknots, t, coefs, degree, scratch.as<double>())))
        return rc;
    return pp_model_set(c, slot, scratch.p, PP_F64, 1, nchan, nbin);
}

// ---- instrumental response applied to a resident template ---------------------
extern "C" int pp_model_apply_response(pp_ctx* c, int slot, const double* rconst, const double* smear_wid) {
    if (int busy_ = ctx_busy(c, "pp_model_apply_response")) return busy_;
    if (!c || slot < 0 || slot >= PP_MAX_SLOTS || !c->slots[slot].set)
        return fail(PP_ESTATE, "pp_model_apply_response: slot not set");
    if (!rconst && !smear_wid) return PP_OK;
    HIP_TRY(hipSetDevice(c->device));
    ModelSlot& s = c->slots[slot];
    const int M = s.nbin / 2;
    int rc;
    const cplx* drc = nullptr;
    const double* dwid = nullptr;
    if (rconst) {
        if ((rc = upload(c, c->seedbuf, rconst, (size_t)(M + 1) * 16))) return rc;
        drc = c->seedbuf.as<cplx>();
    }
    if (smear_wid) {
        if ((rc = upload(c, c->errs, smear_wid, (size_t)s.nchan * 8))) return rc;
        dwid = c->errs.as<double>();
    }
    {
        Prof pr(c, KF_MODEL);
        hipLaunchKernelGGL(k_model_response, dim3(s.nchan), dim3(256), 0, c->stream, s.mft.as<cplx>(),
                           s.msq.as<double>(), s.msum.as<double>(), s.mmax.as<double>(), s.mdc.as<double>(), drc,
                           dwid, s.nchan, M, s.Mp);
    }
    HIP_TRY(hipGetLastError());
    return model_publish(c, slot);
}
