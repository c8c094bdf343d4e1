// Drives the HOST side of libpptoas_hip.so under ThreadSanitizer against the no-op HIP stub (hip_stub.cpp): two
// contexts, each with pp_fit_submit / pp_fit_poll / pp_fit_wait batches in flight on its worker thread while the
// other context is driven from a second caller thread, then the stream-ordered form -- pp_fit_enqueue three deep,
// pp_fit_collect, plain and reference-seed batches alternating (deferred tails carried or flushed), pp_synchronize in
// between -- and the refusals (a fit call while batches are pending).  Kernels do not run: the numbers are zeros and
// nothing is checked but return codes and what the sanitizer sees.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/pp_toas.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != PP_OK) { fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, pp_last_error()); exit(1); } } while (0)

struct Batch {
    int ns, C, B;
    std::vector<double> data, freqs, errs, P, x0, nufit, numean, mprof;
    std::vector<double> params, perr, nu, cov, chi2, rchi2, snr, seedph;
    std::vector<int32_t> nfev, rcode, npass;
    pp_seed_ref rs;
    pp_fit_in in;
    pp_fit_out out;
    Batch(int ns_, int C_, int B_, bool refseed) : ns(ns_), C(C_), B(B_) {
        data.assign((size_t)ns * C * B, 0.5); freqs.resize(C); errs.assign((size_t)ns * C, 0.05); P.assign(ns, 0.003);
        x0.assign((size_t)ns * 5, 0.0); nufit.assign((size_t)ns * 3, 1500.0); numean.assign(ns, 1500.0); mprof.assign(B, 1.0);
        for (int n = 0; n < C; ++n) freqs[n] = 1100.0 + 800.0 * (n + 0.5) / C;
        params.resize((size_t)ns * 5); perr.resize((size_t)ns * 5); nu.resize((size_t)ns * 3); cov.resize((size_t)ns * 25);
        chi2.resize(ns); rchi2.resize(ns); snr.resize(ns); seedph.resize(ns); nfev.resize(ns); rcode.resize(ns); npass.resize(ns);
        memset(&in, 0, sizeof in); memset(&out, 0, sizeof out); memset(&rs, 0, sizeof rs);
        in.nsub = ns; in.nchan = C; in.nbin = B; in.data = data.data(); in.data_dtype = PP_F64; in.freqs = freqs.data();
        in.errs = errs.data(); in.P = P.data(); in.init_params = x0.data(); in.nu_fits = nufit.data();
        in.fit_flags[0] = in.fit_flags[1] = 1; in.is_toa = 1; in.method = PP_METHOD_TRUST_NCG;
        if (refseed) {
            rs.model_profs = mprof.data(); rs.nu_mean = numean.data(); rs.lo = -0.5; rs.hi = 0.5; rs.Ns = 100; rs.finish = 1;
            rs.seed_phase = seedph.data();
            in.ref_seed = &rs;
        }
        out.params = params.data(); out.param_errs = perr.data(); out.nu_refs = nu.data(); out.cov = cov.data();
        out.chi2 = chi2.data(); out.red_chi2 = rchi2.data(); out.snr = snr.data(); out.nfeval = nfev.data();
        out.return_code = rcode.data(); out.npass = npass.data();
    }
};

static pp_ctx* make_ctx(int C, int B) {
    pp_ctx* c = nullptr;
    CHECK(pp_create(0, &c));
    std::vector<double> model((size_t)C * B, 0.0);
    for (int n = 0; n < C; ++n) for (int b = 0; b < B; ++b) model[(size_t)n * B + b] = (b % 7) * 0.1 + n * 1e-3;
    CHECK(pp_model_set(c, 0, model.data(), PP_F64, 0, C, B));
    return c;
}

static void submit_loop(pp_ctx* c, int rounds, int C, int B) {
    for (int r = 0; r < rounds; ++r) {
        Batch b(6 + r % 3, C, B, false);
        CHECK(pp_fit_submit(c, &b.in, &b.out));
        int polls = 0;
        while (pp_fit_poll(c) == 0) { ++polls; std::this_thread::yield(); }
        CHECK(pp_fit_wait(c));
        double v = 0;
        CHECK(pp_get_option(c, "max_iter", &v));
        (void)polls;
    }
}

static void enqueue_loop(pp_ctx* c, int rounds, int C, int B) {
    std::vector<Batch*> live;
    int refused = 0;
    for (int r = 0; r < rounds; ++r) {
        Batch* b = new Batch(5 + r % 4, C, B, (r % 3) == 1);
        CHECK(pp_fit_enqueue(c, &b->in, &b->out));
        live.push_back(b);
        if (r % 5 == 4) CHECK(pp_synchronize(c));          // (flushes a tail nobody carries yet)
        if (pp_fit_pending(c) > 0) {
            Batch probe(2, C, B, false);
            if (pp_fit_portrait_batch(c, &probe.in, &probe.out) == PP_ESTATE) ++refused;     // must refuse while batches are pending
        }
        if (live.size() >= 3) { CHECK(pp_fit_collect(c)); delete live.front(); live.erase(live.begin()); }
    }
    while (!live.empty()) { CHECK(pp_fit_collect(c)); delete live.front(); live.erase(live.begin()); }
    if (refused == 0) { fprintf(stderr, "a synchronous fit was accepted while batches were pending\n"); exit(1); }
}

int main() {
    const int C = 256, B = 2048;
    pp_ctx* a = make_ctx(C, B);
    pp_ctx* b = make_ctx(C, B);
    // two contexts, two caller threads: worker threads of pp_fit_submit on both, then the queue on both
    std::thread t1([&] { submit_loop(a, 12, C, B); enqueue_loop(a, 20, C, B); });
    std::thread t2([&] { enqueue_loop(b, 20, C, B); submit_loop(b, 12, C, B); });
    t1.join(); t2.join();
    // a synchronous batch of another row length, and one that needs sub-batches
    { Batch s(3, C, B, false); CHECK(pp_fit_portrait_batch(a, &s.in, &s.out)); }
    CHECK(pp_set_option(b, "max_work_bytes", 64e6));
    { Batch s(40, C, B, false); CHECK(pp_fit_portrait_batch(b, &s.in, &s.out)); }
    CHECK(pp_destroy(a));
    CHECK(pp_destroy(b));
    printf("tsan_driver: done\n");
    return 0;
}
