// host entry points for pp_extra.h (included at the end of pp_toas.hip)
extern "C" int pp_fit_phase_shift_batch(pp_ctx* c, const double* data, const double* model, const double* noise,
                                        int nprof, int nbin, double lo, double hi, int Ns, double* out7) {
    return fail(PP_ESTATE, "pp_fit_phase_shift_batch: not built yet");
}
extern "C" int pp_synth_portraits(pp_ctx* c, int slot, void* dst, int dtype, int nsub, const double* freqs,
                                  const double* P, const double* inj, double sigma, uint64_t seed,
                                  int64_t first_subint) {
    return fail(PP_ESTATE, "pp_synth_portraits: not built yet");
}
