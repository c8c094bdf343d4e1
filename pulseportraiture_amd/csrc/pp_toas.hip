// libpptoas_hip.so -- host side of the C ABI declared in include/pp_toas.h.
// Pure HIP runtime (no torch, no vendor FFT/BLAS); gfx950 only.
#include "../../include/pp_toas.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <deque>
#include <cstdarg>
#include <cstdio>
#include <climits>
#include <cstdint>
#include <cstring>
#include <map>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "pp_kernels.h"
#include "pp_extra.h"
#include "pp_xspec1024r.h"
#include "pp_tail.h"
#include "pp_anybin.h"

using namespace pp;

// --------------------------------------------------------------------------
// errors
// --------------------------------------------------------------------------
static thread_local std::string g_err;
// the context whose submitted batch this thread is running (pp_fit_submit's worker), or nullptr: set by the worker
// itself -- comparing thread ids with pp_ctx::job raced with the submitting thread's assignment of that very member
// (found by ThreadSanitizer, tools/sanitize: the worker could refuse its own batch with PP_ESTATE)
struct pp_ctx;
static thread_local const pp_ctx* t_worker_of = nullptr;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

struct pp_ctx;
static int ctx_busy(pp_ctx* c, const char* who);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? PP_ENOMEM : PP_EHIP, "%s: %s (%s:%d)",   \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

// --------------------------------------------------------------------------
// context
// --------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return PP_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(PP_ENOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        }
        cap = bytes;
        return PP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct ModelSlot {
    bool set = false;
    int nchan = 0, nbin = 0, Kt = 0;
    int Mp = 0;        // pitch of the spectrum rows: nbin / 2, rounded up to 64 for row lengths that are no power of two
    DevBuf mft, msum, mmax, mdc, kt, msq;
};

// the small input block and the packed outputs cross PCIe by a kernel that reads / writes the pinned staging
// block directly (option copy_kernels): a copy command between two kernels of a stream costs the stream a hand-over
// to the copy engine and back, ~15 us each -- a tenth of a 512 x 1024 step
__global__ void k_copy_words(unsigned long long* dst, const unsigned long long* src, size_t n) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) dst[j] = src[j];
}
static int staged_copy(pp_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st = nullptr);
__global__ void k_publish_int(int* host_dst, const int* dev_src) { *host_dst = *dev_src; }
static int publish_int(pp_ctx* c, int* host_dst, const int* dev_src);
enum KernelFamily { KF_MODEL = 0, KF_XSPEC, KF_PREP, KF_SEED, KF_ACCUM, KF_EVAL, KF_TAYLOR, KF_STEP, KF_FINAL, KF_SYNTH, KF_FPS, KF_SCATMODEL, KF_COUNT };
static const char* kFamilyNames[KF_COUNT] = {"model_fft", "xspec", "prep", "seed", "accum", "eval", "taylor_solve", "step", "finalize",
                                            "synth", "fit_phase_shift", "scat_model"};

#define PP_NSTAGE 3     // enqueued batches that may be pending at once (staging blocks, work-buffer sets)
struct pp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<int, DevBuf> twiddles;   // by nbin
    struct AnyPlan { int L = 0; DevBuf chirp, bft; };
    std::map<int, AnyPlan> anyplans;  // Bluestein tables of row lengths that are no power of two (pp_anybin.h)
    ModelSlot slots[PP_MAX_SLOTS];
    DevBuf mft_table, msum_table, kt_table, mdc_table, msq_table;   // device arrays of slot base pointers
    // work buffers
    DevBuf data, X, sdraw, noise, wts, freqs, errs, mask, P, x0, nufit, nuout, slot, state, csum, partial;
    std::map<const void*, int> occ_cache;   // resident workgroups per CU, by kernel
    DevBuf ticket;                          // k_xspec's chunk counter (RowWalk); never reset,
    unsigned ticket_base = 0;               // ... its value before the next launch (wraps)
    int ncu = 0;                            // compute units of the device
    DevBuf mwords;   // rows in use of a masked batch, one word per chunk (k_mask_words), both row orders
    DevBuf refbuf;   // reference-seed flow: partial channel sums, spectra, profiles, start points
    // Device work buffers of a batch's SOLVE and POST-FIT stage, two sets: a deferred batch (pp_fit_enqueue) runs those
    // stages on a second stream (`stream2`) while the next batch's transform already runs on `stream` -- the next batch
    // writes the other set.  (inpack: the per-batch small inputs -- freqs, P, x0, nu_fit, nu_out, slot -- one H2D copy.)
    // Buffers only the transform stage touches (data, X, errs, mask, the pilot's seed buffers) stay single:
    // transforms are serialised on `stream`.  Synchronous flows use one set and one stream.
    struct WorkSet {
        DevBuf inpack, sdraw, noise, wts, state, csum, partial, tay, mdl, ph0, misc, act, o_pack, o_f0, o_g0, o_H0,
            o_scales, o_serrs, o_csnr, refbuf,
            mwords;      // rows in use of a masked batch, one word per chunk (k_mask_words), both row orders: the reference-seed
                         // flow's finish reads them in the solve stage
        hipEvent_t xdone = nullptr;      // the batch's transform stage has been queued up to here
        void release() {
            DevBuf* b[] = {&inpack, &sdraw, &noise, &wts, &state, &csum, &partial, &tay, &mdl, &ph0, &misc, &act, &o_pack,
                           &o_f0, &o_g0, &o_H0, &o_scales, &o_serrs, &o_csnr, &refbuf, &mwords};
            for (DevBuf* q : b) q->release();
            if (xdone) (void)hipEventDestroy(xdone);
            xdone = nullptr;
        }
    };
    WorkSet work[PP_NSTAGE];
    hipStream_t stream2 = nullptr;   // solve + post-fit stage of deferred batches (higher priority than `stream`)
    hipStream_t last_post = nullptr; // the stream the last fit_chunk queued its post-fit stage on
    int overlap_post = 0;            // deferred batches: 1 = solve / post-fit stage on stream2, beside the next transform.
                                     // Measured (profiles/r05_overlap_ab.txt, three alternations on one box): no gain --
                                     // headline 67.8-68.0 k fits/s either way, configs[1] 735-754 k either way, configs[2]
                                     // -0.5 ... -2.8 %.  The persistent transform of the next batch holds every wave slot
                                     // (two waves of 250 VGPRs per SIMD) before the solve's workgroups (four waves + 64 KB
                                     // of LDS on one CU) are dispatched, stream priority notwithstanding: the solve's event
                                     // span grows from 0.46 to 12-14 ms, i.e. it runs when that transform's workgroups
                                     // retire.  And a wave slot given to the solve is one the transform does not have:
                                     // with registers as the binding resource the two do not co-reside, they alternate.
                                     // Kept as an option: the work-set / two-stream plumbing is what a fused tail would use.
    DevBuf inpack;   // (aux entry points)
    // pinned host staging of the small inputs / the packed outputs of a batch: two sets, so that a
    // deferred batch (pp_fit_enqueue) keeps its own while the next one is being queued
    struct Stage { void* in_host = nullptr; size_t in_cap = 0; void* o_host = nullptr; size_t o_cap = 0;
                   hipEvent_t t0 = nullptr, done = nullptr; };
    Stage stage[PP_NSTAGE];
    int cur_stage = 0;
    // The tail of the youngest enqueued batch -- its solve on the Taylor model, its post-fit stage and the copies of its
    // outputs -- when it has NOT been queued yet (option fuse_tail): the next enqueued batch's transform works it off
    // as tickets (tail_work in pp_kernels.h), or, if none comes / the next one cannot carry it, flush_tail() queues
    // the stand-alone kernels.
    struct PendingTail {
        bool valid = false;
        int stage = 0;                   // staging block / work set of the batch it belongs to
        FitArgs fa;                      // arguments of its solve and post-fit stage
        int ns = 0, C = 0, solve_nt = 0, solve_pf0 = 0, fin_nt = 0;
        size_t solve_lds = 0;
        pp_fit_out out; int s0 = 0; bool chan_dev = false; size_t copy_bytes = 0;
        RefTailArgs rs;                  // reference-seed flow: the guess's finish, fit and start points belong to the tail too
        cplx* rs_dspec = nullptr;        // ... by the stand-alone kernels (flush_tail): the spectrum's array and the fit's
        cplx* rs_xwork = nullptr;        //     work buffer
    } ptail;
    DevBuf tailbuf[PP_NSTAGE];           // the TailArgs a carrying transform reads
    void* tail_host[PP_NSTAGE] = {nullptr, nullptr, nullptr};   // ... and their pinned source
    int fuse_tail = 1;          // enqueued one-pass batches of 2048-bin rows: 1 (default) = the solve + post-fit stage are
                                // not queued behind the transform but worked off as tickets by the NEXT enqueued batch's
                                // transform (tail_work in pp_kernels.h; bitwise the same results); 0 = queued at once
    // pp_fit_enqueue / pp_fit_collect: batches queued on the stream and not yet collected (oldest first)
    struct Deferred { pp_fit_in in; pp_fit_out out; int stage; bool queued; int rc; std::string err; size_t span_end = 0; };
    std::deque<Deferred> pending;
    DevBuf o_pack;   // per-subint scalar outputs, one allocation -> one D2H copy
    DevBuf o_params, o_errs, o_nu, o_cov, o_chi2, o_rchi2, o_snr, o_nfev, o_rc, o_scales, o_serrs, o_csnr,
        o_f0, o_g0, o_H0, misc, seedbuf, tay, ph0, act, seedq, xbase, mdl;
    int* nactive_h = nullptr;   // pinned
    // options
    double harm_eps = 8.8817841970012523e-16;  // 2^-50
    int max_iter = 64;
    int profile = 0;
    int check_every = 1;
    int check_from = 2;         // evaluation loop: the first iteration after which the host looks at the count of unfinished subints
    int use_taylor = 1;
    int moments_in_xspec = 1;   // fold the Taylor moments into k_xspec (mode 2) when it applies
    int scat_model = 1;         // scattering fits: closing iterations on the per-channel model (pp_scatmodel.h)
    double scat_model_tol = 1e-10;
    int scat_model_bet = 1;
    int x_pad = 0;              // pad of the stored cross-spectrum's rows (elements); measured neutral
    int fuse_scat = 1;          // scattering fits: first evaluation inside the transform (k_xspec_qs1024) where it applies
    int x_f32 = 0;              // 1 = scattering fits store the cross-spectrum as float pairs (measured: not worth it)
    int taylor_recentre = 1;    // one-pass flow: re-expansions about the tentative answer when the certificate fails
                                // (0 = none; a second one rarely rescues what the first did not)
    int debug_poison = 0;       // fill the work buffers with NaN bit patterns before every batch (finds unwritten reads)
    int fps_finish = 0;         // pp_fit_phase_shift_batch: 0 = Newton polish, 1 = SciPy brute's simplex finish
    int paired_split = 1;       // 2048-bin rows: last FFT stage + split in registers (k_xspec_p1024)
    int one_exchange = 1;       // 2048-bin rows, mode 2, noise given: one-exchange FFT (k_xspec_q1024)
    int seed_chan_stride = 16;  // device phase seed: pilot pass over every n-th channel (1 = all channels)
    int tail_virtual = 0;       // experiments: solve (and post-fit stage) by ONE real wave per subint that walks the
                                // waves of the multi-wave kernels in turn (bitwise the same results)
    int refseed_stride = 0;     // reference-seed flow: its pilot's channel stride on wide bands (0 = 64; >= 32 pilot channels kept)
    double seed_min_snr = 8.0;  // pilot seeds below this peak significance are redone from all channels
    int seed_ndm = 1;           // DM trials of the coarse (phi, DM) seed grid (1 = phase only, at the guessed DM)
    double seed_dm_step = 0.0;  // their spacing [pc cm^-3]
    double max_work_bytes = 96e9;
    bool solve_lds_attr = false;
    int solve_cache = -1;       // k_taylor_solve: channels whose weight / geometry / template power stay in LDS (-1: all, up to 4096)
    int solve_prefetch = 0;     // k_taylor_solve at more than 2048 channels: 0 = rows fetched in turn (default), 1 = prefetched as for narrow bands
    int solve_threads = 0;      // k_taylor_solve: threads per subint (0: by band width; 64 / 128 / 256 / 512 for A/B)
    int copy_kernels = 1;       // the staged input / output blocks are moved by a kernel instead of a copy command
    int finalize_regs = 1;      // post-fit stage with the channel's numbers held in registers (fits without scattering,
                                // <= 4096 channels); 0: the pass-by-pass kernel (A/B)
    int eager_flush = 1;        // poke the stream once the transform is queued (hipStreamQuery), so that the GPU starts
                                // while the host is still queueing the rest of the batch
    int coarse_newton = 1;      // Newton solver, scattering fits: iterate on every 16th channel first
    int nfev_shadow = -1;       // (-1: by family, measured: profiles/README.md round 4 -- see pp_set_option's table in include/pp_toas.h)
    int skip_masked = 1;        // channels masked out of a subint are not transformed at all (compact row list)
    // profiling
    struct Span { int fam; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;       // recycled profiling events
    std::map<int, double> known_ok;        // largest work-memory demand a batch of each FLOW has already been granted
                                           // (key: scattering | host data | dtype | reference seed | device seed --
                                           // flows hold different buffer sets, and free memory may have shrunk)
    int max_lds_bytes = 0;                 // LDS a workgroup of this device may use (sharedMemPerBlock)
    double fam_sec[KF_COUNT] = {0};
    long long fam_n[KF_COUNT] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evq[2] = {nullptr, nullptr};   // the evaluation loop's lagged checks of the active count
    int lagged_check = 1;       // evaluation loop: read the active count one iteration behind (no idle GPU while the host looks)
    // pp_fit_submit / pp_fit_wait: one fit in flight on a worker thread of the context
    std::thread job;
    bool job_active = false;
    std::atomic<int> job_done{0};
    int job_rc = 0;
    std::string job_err;
    pp_fit_in job_in;
    pp_fit_out job_out;
};

// (events are recycled: creating and destroying ten of them per batch cost more than the solve
// of a 512 x 1024 batch's launch)
static hipEvent_t prof_event(pp_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Prof {
    pp_ctx* c; int fam; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    Prof(pp_ctx* c_, int fam_, hipStream_t st_ = nullptr) : c(c_), fam(fam_), st(st_ ? st_ : c_->stream) {
        if (c->profile) {
            a = prof_event(c); b = prof_event(c);
            (void)hipEventRecord(a, st);
        }
    }
    ~Prof() {
        if (c->profile) {
            (void)hipEventRecord(b, st);
            c->spans.push_back({fam, a, b});
        }
    }
};

// (upto: the first `upto` spans only -- those of a collected batch while later batches are still queued, so that
// a long pipelined run recycles its events instead of creating two per kernel span of every batch)
static void resolve_spans(pp_ctx* c, size_t upto = (size_t)-1) {
    upto = std::min(upto, c->spans.size());
    for (size_t q = 0; q < upto; ++q) {
        auto& s = c->spans[q];
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->fam_sec[s.fam] += 1e-3 * ms;
            c->fam_n[s.fam] += 1;
        }
        c->ev_pool.push_back(s.a); c->ev_pool.push_back(s.b);
    }
    c->spans.erase(c->spans.begin(), c->spans.begin() + upto);
    // (batches still queued keep counting from the new front)
    for (auto& q : c->pending) q.span_end = q.span_end > upto ? q.span_end - upto : 0;
}

// a submitted (pp_fit_submit) or enqueued (pp_fit_enqueue) batch owns the context's work buffers, stream
// and counters until it has been waited for / collected: every other entry point that uses them refuses
static int ctx_busy(pp_ctx* c, const char* who) {
    if (!c) return PP_OK;           // (the entry point reports the null context itself)
    if (c->job_active && t_worker_of != c)
        return fail(PP_ESTATE, "%s: a submitted fit is pending on this context (pp_fit_wait first)", who);
    if (!c->pending.empty())
        return fail(PP_ESTATE, "%s: %zu enqueued batch(es) not collected yet (pp_fit_collect first)", who, c->pending.size());
    return PP_OK;
}

extern "C" int pp_abi_version(void) { return PP_ABI_VERSION; }
extern "C" const char* pp_last_error(void) { return g_err.c_str(); }

static int ctx_init(pp_ctx* c) {
    // PP_CU_EXCLUDE=n (experiments): the context's stream may not use the last n compute units
    // (hipExtStreamCreateWithCUMask) and the persistent grids are sized for the rest -- measures what
    // the transform loses when a few CUs are set aside for other work (profiles/README.md, round 4)
    const char* ex = getenv("PP_CU_EXCLUDE");
    const int nex = ex ? atoi(ex) : 0;
    if (nex > 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int i = 0; i < ncu - nex; ++i) mask[i / 32] |= 1u << (i % 32);
        HIP_TRY(hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()));
        c->ncu = ncu - nex;
    } else
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        // the stream of the solve / post-fit stage of deferred batches: OLDER work, dispatched first where the
        // two streams compete (numerically lower = higher priority)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        }
        for (auto& w : c->work) HIP_TRY(hipEventCreateWithFlags(&w.xdone, hipEventDisableTiming));
    }
    HIP_TRY(hipHostMalloc((void**)&c->nactive_h, sizeof(int) * 4, hipHostMallocDefault));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    for (auto& e : c->evq) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& sg : c->stage) { HIP_TRY(hipEventCreate(&sg.t0)); HIP_TRY(hipEventCreate(&sg.done)); }
    DevBuf* tables[] = {&c->mft_table, &c->msum_table, &c->kt_table, &c->mdc_table, &c->msq_table};
    for (DevBuf* t : tables) {
        int rc = t->reserve(sizeof(void*) * PP_MAX_SLOTS);
        if (rc) return rc;
        HIP_TRY(hipMemset(t->p, 0, sizeof(void*) * PP_MAX_SLOTS));
    }
    int rc = c->ticket.reserve(64);
    if (rc) return rc;
    HIP_TRY(hipMemset(c->ticket.p, 0, 64));
    return PP_OK;
}

extern "C" int pp_destroy(pp_ctx* c);

extern "C" int pp_create(int device_id, pp_ctx** out) {
    if (!out) return fail(PP_EINVAL, "pp_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(PP_EINVAL, "pp_create: device %d of %d", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    pp_ctx* c = new pp_ctx();
    c->device = device_id;
    const int rc = ctx_init(c);
    if (rc) {
        // release whatever the half-built context already holds (the error
        // string of the failing call stays in place)
        const std::string keep = g_err;
        (void)pp_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return PP_OK;
}

extern "C" int pp_destroy(pp_ctx* c) {
    if (!c) return PP_OK;
    if (c->job_active) { c->job.join(); c->job_active = false; }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    resolve_spans(c);
    for (auto& w : c->work) w.release();
    for (auto& b : c->tailbuf) b.release();
    for (void*& h : c->tail_host) { if (h) (void)hipHostFree(h); h = nullptr; }
    for (auto& kv : c->twiddles) kv.second.release();
    for (auto& kv : c->anyplans) { kv.second.chirp.release(); kv.second.bft.release(); }
    for (auto& s : c->slots) { s.mft.release(); s.msum.release(); s.mmax.release(); s.mdc.release(); s.kt.release(); s.msq.release(); }
    for (auto& sg : c->stage) {
        if (sg.o_host) (void)hipHostFree(sg.o_host);
        if (sg.in_host) (void)hipHostFree(sg.in_host);
        if (sg.t0) (void)hipEventDestroy(sg.t0);
        if (sg.done) (void)hipEventDestroy(sg.done);
    }
    c->inpack.release();
    c->refbuf.release();
    c->mwords.release();
    DevBuf* bufs[] = {&c->ticket, &c->o_pack, &c->mft_table, &c->msum_table, &c->kt_table, &c->mdc_table, &c->msq_table, &c->data, &c->X, &c->sdraw, &c->noise, &c->wts, &c->freqs,
                      &c->errs, &c->mask, &c->P, &c->x0, &c->nufit, &c->nuout, &c->slot, &c->state, &c->csum,
                      &c->partial, &c->o_params, &c->o_errs, &c->o_nu, &c->o_cov, &c->o_chi2, &c->o_rchi2,
                      &c->o_snr, &c->o_nfev, &c->o_rc, &c->o_scales, &c->o_serrs, &c->o_csnr, &c->o_f0, &c->o_g0,
                      &c->o_H0, &c->misc, &c->seedbuf, &c->tay, &c->ph0, &c->act, &c->seedq, &c->xbase};
    for (DevBuf* b : bufs) b->release();
    if (c->nactive_h) (void)hipHostFree(c->nactive_h);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    c->ev_pool.clear();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto& e : c->evq) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;
    return PP_OK;
}

static int flush_tail(pp_ctx* c);
extern "C" int pp_synchronize(pp_ctx* c) {
    if (!c) return fail(PP_EINVAL, "null context");
    // (the youngest enqueued batch's tail may still be unqueued -- option fuse_tail --: a caller that synchronises
    // expects the device-resident outputs of everything it has enqueued, so it goes out by the stand-alone kernels)
    if (int rc = flush_tail(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    return PP_OK;
}

extern "C" void* pp_stream(pp_ctx* c) { return c ? (void*)c->stream : nullptr; }

// one table for pp_set_option / pp_get_option: name, kind ('d' double, 'i' int), lower clamp of ints
namespace {
struct OptRef { const char* name; char kind; void* p; int imin; };
}
static bool option_ref(pp_ctx* c, const std::string& n, OptRef* out) {
    const OptRef tab[] = {
        {"harm_eps", 'd', &c->harm_eps, 0}, {"max_iter", 'i', &c->max_iter, INT32_MIN},
        {"profile", 'i', &c->profile, INT32_MIN}, {"check_every", 'i', &c->check_every, 1}, {"check_from", 'i', &c->check_from, 2},
        {"lagged_check", 'i', &c->lagged_check, INT32_MIN}, {"max_work_bytes", 'd', &c->max_work_bytes, 0},
        {"taylor", 'i', &c->use_taylor, INT32_MIN}, {"moments_in_xspec", 'i', &c->moments_in_xspec, INT32_MIN},
        {"paired_split", 'i', &c->paired_split, INT32_MIN}, {"one_exchange", 'i', &c->one_exchange, INT32_MIN},
        {"scat_model", 'i', &c->scat_model, INT32_MIN}, {"scat_model_tol", 'd', &c->scat_model_tol, 0},
        {"scat_model_bet", 'i', &c->scat_model_bet, INT32_MIN}, {"x_f32", 'i', &c->x_f32, INT32_MIN},
        {"fuse_scat", 'i', &c->fuse_scat, INT32_MIN}, {"x_pad", 'i', &c->x_pad, INT32_MIN},
        {"fps_finish", 'i', &c->fps_finish, INT32_MIN}, {"debug_poison", 'i', &c->debug_poison, INT32_MIN},
        {"taylor_recentre", 'i', &c->taylor_recentre, INT32_MIN}, {"seed_chan_stride", 'i', &c->seed_chan_stride, 1},
        {"seed_min_snr", 'd', &c->seed_min_snr, 0}, {"seed_ndm", 'i', &c->seed_ndm, 1},
        {"seed_dm_step", 'd', &c->seed_dm_step, 0}, {"skip_masked", 'i', &c->skip_masked, INT32_MIN},
        {"nfev_shadow", 'i', &c->nfev_shadow, INT32_MIN}, {"coarse_newton", 'i', &c->coarse_newton, INT32_MIN},
        {"eager_flush", 'i', &c->eager_flush, INT32_MIN},
        {"finalize_regs", 'i', &c->finalize_regs, INT32_MIN}, {"solve_cache", 'i', &c->solve_cache, -1},
        {"solve_threads", 'i', &c->solve_threads, 0}, {"copy_kernels", 'i', &c->copy_kernels, INT32_MIN},
        {"solve_prefetch", 'i', &c->solve_prefetch, INT32_MIN}, {"overlap_post", 'i', &c->overlap_post, INT32_MIN},
        {"refseed_stride", 'i', &c->refseed_stride, 0}, {"tail_virtual", 'i', &c->tail_virtual, INT32_MIN},
        {"fuse_tail", 'i', &c->fuse_tail, INT32_MIN},
    };
    for (const OptRef& o : tab)
        if (n == o.name) { *out = o; return true; }
    return false;
}

extern "C" int pp_set_option(pp_ctx* c, const char* name, double value) {
    if (!c || !name) return fail(PP_EINVAL, "pp_set_option: null argument");
    OptRef o;
    if (!option_ref(c, name, &o)) return fail(PP_EINVAL, "pp_set_option: unknown option '%s'", name);
    if (o.kind == 'd') *static_cast<double*>(o.p) = value;
    else *static_cast<int*>(o.p) = std::max(o.imin, (int)value);
    return PP_OK;
}

extern "C" int pp_get_option(pp_ctx* c, const char* name, double* value) {
    if (!c || !name || !value) return fail(PP_EINVAL, "pp_get_option: null argument");
    OptRef o;
    if (!option_ref(c, name, &o)) return fail(PP_EINVAL, "pp_get_option: unknown option '%s'", name);
    *value = o.kind == 'd' ? *static_cast<double*>(o.p) : (double)*static_cast<int*>(o.p);
    return PP_OK;
}

extern "C" int pp_kernel_times(pp_ctx* c, int cap, const char** names, double* seconds, int64_t* launches) {
    if (!c) return fail(PP_EINVAL, "null context");
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream2);
    resolve_spans(c);
    int n = std::min(cap, (int)KF_COUNT);
    for (int i = 0; i < n; ++i) {
        if (names) names[i] = kFamilyNames[i];
        if (seconds) seconds[i] = c->fam_sec[i];
        if (launches) launches[i] = c->fam_n[i];
    }
    return n;
}

extern "C" int pp_kernel_times_reset(pp_ctx* c) {
    if (!c) return fail(PP_EINVAL, "null context");
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream2);
    resolve_spans(c);
    for (int i = 0; i < KF_COUNT; ++i) { c->fam_sec[i] = 0; c->fam_n[i] = 0; }
    return PP_OK;
}

// --------------------------------------------------------------------------
// twiddles: W_B^k = exp(-2 pi i k / B), k = 0..B/2, correctly rounded
// --------------------------------------------------------------------------
static int get_twiddles(pp_ctx* c, int nbin, const cplx** out) {
    auto it = c->twiddles.find(nbin);
    if (it != c->twiddles.end()) { *out = it->second.as<cplx>(); return PP_OK; }
    const int M = nbin / 2;
    std::vector<double> h(2 * (size_t)(M + 1));
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k <= M; ++k) {
        // octant symmetry keeps the argument in [0, pi/4] for full accuracy
        long double ang = two_pi * (long double)k / (long double)nbin;
        h[2 * k] = (double)cosl(ang);
        h[2 * k + 1] = (double)(-sinl(ang));
    }
    h[0] = 1.0; h[1] = 0.0;
    if (M % 2 == 0) { h[2 * (M / 2)] = 0.0; h[2 * (M / 2) + 1] = -1.0; }
    h[2 * M] = -1.0; h[2 * M + 1] = 0.0;
    DevBuf b;
    int rc = b.reserve(h.size() * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemcpy(b.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    c->twiddles[nbin] = b;
    *out = b.as<cplx>();
    return PP_OK;
}

// row lengths with a tuned plan (every entry point), and every row length the fit itself takes:
// the reference's numpy.fft.rfft accepts any (pptoaslib.py:976-979); even lengths up to 4096 that
// are no power of two go through pp_anybin.h
static int staged_copy(pp_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    if (!st) st = c->stream;
    if (c->copy_kernels && bytes % 8 == 0 && bytes <= ((size_t)64 << 20)) {
        const size_t n = bytes / 8;
        const unsigned nb = (unsigned)std::min<size_t>((n + 255) / 256, 256);
        hipLaunchKernelGGL(k_copy_words, dim3(nb ? nb : 1), dim3(256), 0, st, (unsigned long long*)dst,
                           (const unsigned long long*)src, n);
        return hipGetLastError() == hipSuccess ? PP_OK : PP_EHIP;
    }
    return hipMemcpyAsync(dst, src, bytes, kind, st) == hipSuccess ? PP_OK : PP_EHIP;
}
// one int of device state into the pinned word the host looks at an iteration later (the lagged check of the
// evaluation loop: the count of unfinished subints); where the host waits at once a copy command measured faster
static int publish_int(pp_ctx* c, int* host_dst, const int* dev_src) {
    if (c->copy_kernels) {
        hipLaunchKernelGGL(k_publish_int, dim3(1), dim3(1), 0, c->stream, host_dst, dev_src);
        return hipGetLastError() == hipSuccess ? PP_OK : PP_EHIP;
    }
    return hipMemcpyAsync(host_dst, dev_src, sizeof(int), hipMemcpyDeviceToHost, c->stream) == hipSuccess ? PP_OK : PP_EHIP;
}
static bool nbin_ok(int nbin) { return nbin >= 32 && nbin <= 8192 && (nbin & (nbin - 1)) == 0; }
static bool nbin_any_ok(int nbin) { return nbin_ok(nbin) || (nbin >= 8 && nbin <= 4096 && nbin % 2 == 0); }
static int fail(int code, const char* fmt, ...);
// what an entry point answers for a row length nobody takes (round 5: every entry point takes the general even
// lengths the fit takes)
static int nbin_refuse(const char* who, int nbin) {
    return fail(PP_EINVAL, "%s: nbin %d must be a power of two in [32, 8192] or an even number in [8, 4096]", who, nbin);
}

// Bluestein tables of a row length B = 2 M that is no power of two
static int get_twiddles(pp_ctx* c, int nbin, const cplx** out);
static int get_any_plan(pp_ctx* c, int nbin, pp_ctx::AnyPlan** out) {
    auto it = c->anyplans.find(nbin);
    if (it != c->anyplans.end()) { *out = &it->second; return PP_OK; }
    const int M = nbin / 2;
    int L = 16;
    while (L < 2 * M - 1) L <<= 1;
    // (four transform sizes are instantiated: the next one up serves)
    const int sizes[4] = {64, 256, 1024, 4096};
    for (int sz : sizes) if (L <= sz) { L = sz; break; }
    const long double pi = 3.141592653589793238462643383279502884L;
    std::vector<double> w(2 * (size_t)M);
    std::vector<std::complex<long double>> b((size_t)L, std::complex<long double>(0.0L, 0.0L));
    for (int j = 0; j < M; ++j) {
        const long long q = ((long long)j * j) % (2LL * M);        // j^2 mod 2 M, exactly
        const long double ang = pi * (long double)q / (long double)M;
        const long double cs = cosl(ang), sn = sinl(ang);
        w[2 * (size_t)j] = (double)cs; w[2 * (size_t)j + 1] = (double)(-sn);           // w_j = exp(-i pi j^2 / M)
        b[j] = std::complex<long double>(cs, sn);                                       // conj(w_j)
        if (j) b[L - j] = b[j];
    }
    // transform of the wrapped chirp, radix 2 in extended precision (once per row length)
    {
        for (int i = 1, j = 0; i < L; ++i) {
            int bit = L >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(b[i], b[j]);
        }
        for (int len = 2; len <= L; len <<= 1) {
            for (int i = 0; i < L; i += len)
                for (int k = 0; k < len / 2; ++k) {
                    const long double ang = -2.0L * pi * (long double)k / (long double)len;
                    const std::complex<long double> tw(cosl(ang), sinl(ang));
                    const std::complex<long double> u = b[i + k], v = b[i + k + len / 2] * tw;
                    b[i + k] = u + v; b[i + k + len / 2] = u - v;
                }
        }
    }
    std::vector<double> bf(2 * (size_t)L);
    for (int k = 0; k < L; ++k) { bf[2 * (size_t)k] = (double)b[k].real(); bf[2 * (size_t)k + 1] = (double)b[k].imag(); }
    pp_ctx::AnyPlan pl;
    pl.L = L;
    int rc;
    if ((rc = pl.chirp.reserve(w.size() * 8))) return rc;
    if ((rc = pl.bft.reserve(bf.size() * 8))) return rc;
    HIP_TRY(hipMemcpy(pl.chirp.p, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pl.bft.p, bf.data(), bf.size() * 8, hipMemcpyHostToDevice));
    c->anyplans[nbin] = pl;
    *out = &c->anyplans[nbin];
    return PP_OK;
}

static int fft_grid(int T, long long nrows) {
    // persistent workgroups: enough to fill 256 CUs at the LDS-limited residency
    long long g = 256LL * (T <= 128 ? 8 : (T == 256 ? 4 : 2));
    return (int)std::max(1LL, std::min(nrows, g));
}

// one launch of the general-length transform (pp_anybin.h): mode -1 = harmonics to hout, 0..3 = k_xspec's modes
static int launch_any(pp_ctx* c, const XspecArgs& xa, int nbin, int Mp, int dtype, int mode, bool tail, cplx* hout,
                      const unsigned char* mask) {
    pp_ctx::AnyPlan* pl = nullptr;
    int rc;
    if ((rc = get_any_plan(c, nbin, &pl))) return rc;
    const cplx *twL = nullptr, *twB = nullptr;
    if ((rc = get_twiddles(c, 2 * pl->L, &twL))) return rc;
    if ((rc = get_twiddles(c, nbin, &twB))) return rc;
    AnyArgs g{nbin, nbin / 2, Mp, pl->chirp.as<cplx>(), pl->bft.as<cplx>(), twL, twB, mode, tail ? 1 : 0, hout, mask};
    const long long nrows = (long long)xa.nsub * xa.nchan;
    const int grid = (int)std::max(1LL, std::min(nrows, 2048LL));
#define PP_ANY(LL)                                                                                            \
    do {                                                                                                      \
        if (dtype == PP_F64) hipLaunchKernelGGL((k_any<LL, double>), dim3(grid), dim3(FftPlan<LL>::T), 0, c->stream, xa, g); \
        else hipLaunchKernelGGL((k_any<LL, float>), dim3(grid), dim3(FftPlan<LL>::T), 0, c->stream, xa, g);   \
    } while (0)
    switch (pl->L) {
        case 64: PP_ANY(64); break;
        case 256: PP_ANY(256); break;
        case 1024: PP_ANY(1024); break;
        case 4096: PP_ANY(4096); break;
        default: return fail(PP_EINVAL, "no transform of %d points", pl->L);
    }
#undef PP_ANY
    HIP_TRY(hipGetLastError());
    return PP_OK;
}

// dispatch a templated kernel on (M, dtype)
#define PP_DISPATCH_M(M_, ...)         \
    switch (M_) {                      \
        case 16: { constexpr int MM = 16; __VA_ARGS__; break; }     \
        case 32: { constexpr int MM = 32; __VA_ARGS__; break; }     \
        case 64: { constexpr int MM = 64; __VA_ARGS__; break; }     \
        case 128: { constexpr int MM = 128; __VA_ARGS__; break; }   \
        case 256: { constexpr int MM = 256; __VA_ARGS__; break; }   \
        case 512: { constexpr int MM = 512; __VA_ARGS__; break; }   \
        case 1024: { constexpr int MM = 1024; __VA_ARGS__; break; } \
        case 2048: { constexpr int MM = 2048; __VA_ARGS__; break; } \
        case 4096: { constexpr int MM = 4096; __VA_ARGS__; break; } \
        default: return fail(PP_EINVAL, "unsupported nbin %d", 2 * (M_)); \
    }

// --------------------------------------------------------------------------
// model
// --------------------------------------------------------------------------
// harmonic truncation of a slot whose spectrum is in place (per-channel kept
// harmonics kt[n] and their maximum) and its entry in the device pointer tables
static int model_publish(pp_ctx* c, int slot) {
    ModelSlot& s = c->slots[slot];
    const int nchan = s.nchan, M = s.Mp;      // (rows pitched to Mp, zeros beyond nbin / 2)
    int rc;
    int Kt = M;
    {
        if ((rc = c->misc.reserve(256))) return rc;
        HIP_TRY(hipMemsetAsync(c->misc.p, 0, sizeof(int), c->stream));
        // eps = 0 keeps everything: threshold -1 makes every harmonic "significant"
        const double eps2 = c->harm_eps > 0.0 ? c->harm_eps * c->harm_eps : -1.0;
        hipLaunchKernelGGL(k_model_kcut, dim3(nchan), dim3(64), 0, c->stream, s.mft.as<cplx>(), s.mmax.as<double>(),
                           nchan, M, eps2, s.kt.as<int>(), c->misc.as<int>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&Kt, c->misc.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        Kt = std::min(M, std::max(64, Kt));
    }
    s.set = true; s.Kt = Kt;
    void* ptrs[5] = {s.mft.p, s.msum.p, s.kt.p, s.mdc.p, s.msq.p};
    DevBuf* tables[5] = {&c->mft_table, &c->msum_table, &c->kt_table, &c->mdc_table, &c->msq_table};
    for (int j = 0; j < 5; ++j)
        HIP_TRY(hipMemcpy((char*)tables[j]->p + sizeof(void*) * slot, &ptrs[j], sizeof(void*), hipMemcpyHostToDevice));
    return PP_OK;
}

extern "C" int pp_model_set(pp_ctx* c, int slot, const void* portrait, int dtype, int on_device, int nchan,
                            int nbin) {
    if (int busy_ = ctx_busy(c, "pp_model_set")) return busy_;
    if (!c || !portrait) return fail(PP_EINVAL, "pp_model_set: null argument");
    if (slot < 0 || slot >= PP_MAX_SLOTS) return fail(PP_EINVAL, "pp_model_set: slot %d", slot);
    if (!nbin_any_ok(nbin))
        return fail(PP_EINVAL, "pp_model_set: nbin %d must be a power of two in [32, 8192] or even in [8, 4096]", nbin);
    if (nchan < 1) return fail(PP_EINVAL, "pp_model_set: nchan %d", nchan);
    if (dtype != PP_F64 && dtype != PP_F32) return fail(PP_EINVAL, "pp_model_set: dtype %d", dtype);
    HIP_TRY(hipSetDevice(c->device));
    const bool anyb = !nbin_ok(nbin);
    const int M = nbin / 2, Mp = anyb ? ((M + 63) / 64) * 64 : M;
    const size_t esz = dtype == PP_F64 ? 8 : 4;
    ModelSlot& s = c->slots[slot];
    int rc;
    if ((rc = s.mft.reserve((size_t)nchan * Mp * sizeof(cplx)))) return rc;
    if ((rc = s.msum.reserve((size_t)nchan * sizeof(double)))) return rc;
    if ((rc = s.mmax.reserve((size_t)nchan * sizeof(double)))) return rc;
    if ((rc = s.mdc.reserve((size_t)nchan * sizeof(double)))) return rc;
    if ((rc = s.kt.reserve((size_t)nchan * sizeof(int)))) return rc;
    if ((rc = s.msq.reserve((size_t)nchan * Mp * sizeof(double)))) return rc;
    const void* dport = portrait;
    if (!on_device) {
        if ((rc = c->data.reserve((size_t)nchan * nbin * esz))) return rc;
        HIP_TRY(hipMemcpyAsync(c->data.p, portrait, (size_t)nchan * nbin * esz, hipMemcpyHostToDevice, c->stream));
        dport = c->data.p;
    }
    if (anyb) {
        // general row length: harmonics by the Bluestein path, then the slot's padded rows
        if ((rc = c->X.reserve((size_t)nchan * (M + 1) * sizeof(cplx)))) return rc;
        XspecArgs xa;
        memset(&xa, 0, sizeof xa);
        xa.data = dport; xa.nsub = 1; xa.nchan = nchan; xa.nchan_full = nchan; xa.cstep = 1;
        Prof pr(c, KF_MODEL);
        if ((rc = launch_any(c, xa, nbin, Mp, dtype, -1, false, c->X.as<cplx>(), nullptr))) return rc;
        hipLaunchKernelGGL(k_model_from_harm, dim3(nchan), dim3(64), 0, c->stream, (const cplx*)c->X.p, M, Mp,
                           s.mft.as<cplx>(), s.msq.as<double>(), s.msum.as<double>(), s.mmax.as<double>(), s.mdc.as<double>());
        HIP_TRY(hipGetLastError());
        s.nchan = nchan; s.nbin = nbin; s.Mp = Mp;
        return model_publish(c, slot);
    }
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, nbin, &tw))) return rc;
    ModelFftArgs a{dport, s.mft.as<cplx>(), s.msum.as<double>(), s.mmax.as<double>(), s.mdc.as<double>(),
                   s.msq.as<double>(), tw, nchan};
    {
        Prof pr(c, KF_MODEL);
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            if (dtype == PP_F64) hipLaunchKernelGGL((k_model_fft<MM, double>), dim3(fft_grid(T, nchan)), dim3(T), 0, c->stream, a);
            else hipLaunchKernelGGL((k_model_fft<MM, float>), dim3(fft_grid(T, nchan)), dim3(T), 0, c->stream, a);
        });
    }
    HIP_TRY(hipGetLastError());
    s.nchan = nchan; s.nbin = nbin; s.Mp = M;
    return model_publish(c, slot);
}

extern "C" int pp_model_dc(pp_ctx* c, int slot, double* dc) {
    if (!c || !dc || slot < 0 || slot >= PP_MAX_SLOTS || !c->slots[slot].set) return fail(PP_ESTATE, "slot not set");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dc, c->slots[slot].mdc.p, (size_t)c->slots[slot].nchan * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

extern "C" int pp_model_nharm(pp_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= PP_MAX_SLOTS || !c->slots[slot].set) return fail(PP_ESTATE, "slot not set");
    return c->slots[slot].Kt;
}

// --------------------------------------------------------------------------
// rFFT parity hook
// --------------------------------------------------------------------------
extern "C" int pp_rfft_rows(pp_ctx* c, const void* rows, int dtype, int nrows, int nbin, double* out) {
    if (int busy_ = ctx_busy(c, "pp_rfft_rows")) return busy_;
    if (!c || !rows || !out) return fail(PP_EINVAL, "pp_rfft_rows: null argument");
    if (!nbin_any_ok(nbin) || nrows < 1) return fail(PP_EINVAL, "pp_rfft_rows: bad shape %d x %d", nrows, nbin);
    HIP_TRY(hipSetDevice(c->device));
    const int M = nbin / 2;
    const size_t esz = dtype == PP_F64 ? 8 : 4;
    int rc;
    if ((rc = c->data.reserve((size_t)nrows * nbin * esz))) return rc;
    if ((rc = c->X.reserve((size_t)nrows * (M + 1) * sizeof(cplx)))) return rc;
    HIP_TRY(hipMemcpyAsync(c->data.p, rows, (size_t)nrows * nbin * esz, hipMemcpyHostToDevice, c->stream));
    if (!nbin_ok(nbin)) {
        XspecArgs xa;
        memset(&xa, 0, sizeof xa);
        xa.data = c->data.p; xa.nsub = 1; xa.nchan = nrows; xa.nchan_full = nrows; xa.cstep = 1;
        if ((rc = launch_any(c, xa, nbin, ((M + 63) / 64) * 64, dtype, -1, false, c->X.as<cplx>(), nullptr))) return rc;
    } else {
        const cplx* tw = nullptr;
        if ((rc = get_twiddles(c, nbin, &tw))) return rc;
        PP_DISPATCH_M(M, {
            const int T = FftPlan<MM>::T;
            if (dtype == PP_F64) hipLaunchKernelGGL((k_rfft_rows<MM, double>), dim3(fft_grid(T, nrows)), dim3(T), 0, c->stream, (const void*)c->data.p, c->X.as<cplx>(), tw, nrows);
            else hipLaunchKernelGGL((k_rfft_rows<MM, float>), dim3(fft_grid(T, nrows)), dim3(T), 0, c->stream, (const void*)c->data.p, c->X.as<cplx>(), tw, nrows);
        });
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(out, c->X.p, (size_t)nrows * (M + 1) * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PP_OK;
}

// --------------------------------------------------------------------------
// the batched fit
// --------------------------------------------------------------------------
static int upload(pp_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = b.reserve(bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return PP_OK;
}

// persistent grid of exactly the resident capacity of this instantiation (its
// register / LDS footprint decides how many workgroups a CU holds): a larger grid
// would run its surplus workgroups in a second, mostly idle round
template <typename K>
static int resident_grid(pp_ctx* c, K kernel, int T, long long nrows, int fallback) {
    const void* key = reinterpret_cast<const void*>(kernel);
    auto it = c->occ_cache.find(key);
    int per_cu = 0;
    if (it != c->occ_cache.end()) per_cu = it->second;
    else {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, T, 0) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 0;
        }
        c->occ_cache[key] = per_cu;
    }
    if (!c->ncu) {
        hipDeviceProp_t prop;
        c->ncu = (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
                     ? prop.multiProcessorCount : 256;
    }
    long long g = per_cu > 0 ? (long long)per_cu * c->ncu : (long long)fallback;
    // PP_GRID_SCALE=f (experiments): the persistent grids at a fraction f of their residency -- what the transform
    // loses when wave slots are left free for other work (profiles/README.md, round 5)
    static const double scale = [] { const char* e = getenv("PP_GRID_SCALE"); return e ? atof(e) : 1.0; }();
    if (scale > 0.0 && scale != 1.0) g = std::max(1LL, (long long)std::llround((double)g * scale));
    return (int)std::max(1LL, std::min(nrows, g));
}

template <int MM, typename TIN>
static void launch_xspec(pp_ctx* c, const XspecArgs& xa_in, bool tail, int mode) {
    const int T = FftPlan<MM>::T;
    const long long nrows = (long long)xa_in.nsub * xa_in.nchan;
    // one ticket per chunk of rows is drawn by the launch (RowWalk)
    XspecArgs xa = xa_in;
    xa.ticket = c->ticket.as<unsigned>();
    xa.ticket_base = c->ticket_base;
    if (T == 64) c->ticket_base += (unsigned)((nrows + PP_ROW_CHUNK - 1) / PP_ROW_CHUNK);
    const dim3 blk(T);
    if constexpr (MM == 1024) {
        // 2048-bin rows, Taylor sums only (noise given or measured): the one-exchange transform
        // (pp_xspec1024q.h) -- a third of the LDS traffic of the general kernel
        if (c->one_exchange && mode == 2 && 2 * xa.Kt < MM) {
            if (tail) {
                const dim3 grid(resident_grid(c, k_xspec_q1024<TIN, true>, T, nrows, fft_grid(T, nrows)));
                hipLaunchKernelGGL((k_xspec_q1024<TIN, true>), grid, blk, 0, c->stream, xa);
            } else {
                const dim3 grid(resident_grid(c, k_xspec_q1024<TIN, false>, T, nrows, fft_grid(T, nrows)));
                hipLaunchKernelGGL((k_xspec_q1024<TIN, false>), grid, blk, 0, c->stream, xa);
            }
            return;
        }
    }
    if constexpr (MM == 1024) {
        // ... and with a template that keeps 512 harmonics or more (mode 3)
        if (c->one_exchange && mode == 3) {
            if (tail) {
                const dim3 grid(resident_grid(c, k_xspec_qf<1024, TIN, true>, T, nrows, fft_grid(T, nrows)));
                hipLaunchKernelGGL((k_xspec_qf<1024, TIN, true>), grid, blk, 0, c->stream, xa);
            } else {
                const dim3 grid(resident_grid(c, k_xspec_qf<1024, TIN, false>, T, nrows, fft_grid(T, nrows)));
                hipLaunchKernelGGL((k_xspec_qf<1024, TIN, false>), grid, blk, 0, c->stream, xa);
            }
            return;
        }
    }
    if constexpr (MM == 512) {
        // 1024-bin rows, Taylor sums only: the one-exchange transform of pp_fftq.h (plan 8.4.2.8),
        // whatever the template keeps
        if (c->one_exchange && (mode == 2 || mode == 3)) {
            if (tail) {
                const dim3 grid(resident_grid(c, k_xspec_qf<512, TIN, true>, T, nrows, fft_grid(T, nrows)));
                hipLaunchKernelGGL((k_xspec_qf<512, TIN, true>), grid, blk, 0, c->stream, xa);
            } else {
                const dim3 grid(resident_grid(c, k_xspec_qf<512, TIN, false>, T, nrows, fft_grid(T, nrows)));
                hipLaunchKernelGGL((k_xspec_qf<512, TIN, false>), grid, blk, 0, c->stream, xa);
            }
            return;
        }
    }
    if constexpr (MM == 1024 && sizeof(TIN) == 4) {
        // 2048-bin rows whose template keeps fewer than 512 harmonics: last stage
        // and split in registers (pp_xspec1024.h)
        // (f32 portraits only: with f64 rows the 64 prefetch registers on top of the
        // 16 held outputs push the kernel over 256 VGPRs -- it spills and loses)
        if (c->paired_split && 2 * xa.Kt < MM && mode <= 2 && !xa.act && xa.cstep == 1) {
#define PP_XP(TL, MD)                                                                                  \
    do {                                                                                               \
        const dim3 grid(resident_grid(c, k_xspec_p1024<TIN, TL, MD>, T, nrows, fft_grid(T, nrows)));   \
        hipLaunchKernelGGL((k_xspec_p1024<TIN, TL, MD>), grid, blk, 0, c->stream, xa);                 \
    } while (0)
            if (tail) { if (mode == 2) PP_XP(true, 2); else if (mode == 1) PP_XP(true, 1); else PP_XP(true, 0); }
            else { if (mode == 2) PP_XP(false, 2); else if (mode == 1) PP_XP(false, 1); else PP_XP(false, 0); }
#undef PP_XP
            return;
        }
    }
#define PP_XS(TL, MD)                                                                                  \
    do {                                                                                               \
        const dim3 grid(resident_grid(c, k_xspec<MM, TIN, TL, MD>, T, nrows, fft_grid(T, nrows)));     \
        hipLaunchKernelGGL((k_xspec<MM, TIN, TL, MD>), grid, blk, 0, c->stream, xa);                   \
    } while (0)
    if (tail) {
        if (mode == 3) PP_XS(true, 3); else if (mode == 2) PP_XS(true, 2); else if (mode == 1) PP_XS(true, 1); else PP_XS(true, 0);
    } else {
        if (mode == 3) PP_XS(false, 3); else if (mode == 2) PP_XS(false, 2); else if (mode == 1) PP_XS(false, 1); else PP_XS(false, 0);
    }
#undef PP_XS
}

// the packed per-subint outputs of a batch, from its host staging block to the caller's arrays
static void unpack_stage(const void* o_host, pp_fit_out* out, int s0, int ns) {
    const double* h = reinterpret_cast<const double*>(o_host);
    memcpy(out->params + (size_t)s0 * 5, h, (size_t)ns * 40);
    memcpy(out->param_errs + (size_t)s0 * 5, h + (size_t)ns * 5, (size_t)ns * 40);
    memcpy(out->nu_refs + (size_t)s0 * 3, h + (size_t)ns * 10, (size_t)ns * 24);
    memcpy(out->cov + (size_t)s0 * 25, h + (size_t)ns * 13, (size_t)ns * 200);
    memcpy(out->chi2 + s0, h + (size_t)ns * 38, (size_t)ns * 8);
    memcpy(out->red_chi2 + s0, h + (size_t)ns * 39, (size_t)ns * 8);
    memcpy(out->snr + s0, h + (size_t)ns * 40, (size_t)ns * 8);
    const int32_t* hi = reinterpret_cast<const int32_t*>(h + (size_t)ns * 41);
    memcpy(out->nfeval + s0, hi, (size_t)ns * 4);
    memcpy(out->return_code + s0, hi + ns, (size_t)ns * 4);
    if (out->npass) memcpy(out->npass + s0, hi + 2 * (size_t)ns, (size_t)ns * 4);
}
// the reference-seed flow's phase guesses ride in the same pinned block, behind the packed outputs (a copy
// straight into the caller's pageable array would make the enqueueing call wait for the whole batch)
static size_t stage_seed_offset(int ns) { return (((size_t)ns * 340 + 8) + 7) & ~(size_t)7; }
static void unpack_seed_phases(const void* o_host, const pp_fit_in* in, int s0, int ns) {
    if (in->ref_seed && in->ref_seed->seed_phase)
        memcpy(in->ref_seed->seed_phase + s0, reinterpret_cast<const char*>(o_host) + stage_seed_offset(ns), (size_t)ns * 8);
}
// subints the post-fit stage found unfinished (as k_finalize saw it)
static int unfinished_in_stage(const void* o_host, int ns) {
    return reinterpret_cast<const int32_t*>(reinterpret_cast<const double*>(o_host) + (size_t)ns * 41)[3 * (size_t)ns];
}

// the solve on the Taylor model: NT threads per subint by band width (or option solve_threads), rows fetched in turn
// for bands wider than 2048 channels.  `fa.solve_cache` may be lowered (dynamic LDS refused by the runtime).
static int solve_threads_for(pp_ctx* c, int C) { return c->solve_threads > 0 ? c->solve_threads : (C <= 512 ? 64 : C <= 1024 ? 128 : 256); }
static bool solve_rows_in_turn(pp_ctx* c, int C, int solve_nt) { return solve_nt == 256 && C > 2048 && c->solve_prefetch <= 0; }
static void solve_launch(pp_ctx* c, FitArgs& fa, int ns, int C, int solve_nt, hipStream_t sp) {
    size_t lds = (size_t)fa.solve_cache * 32;
    if (lds > 48 * 1024 && !c->solve_lds_attr) {     // (dynamic LDS beyond the default cap: said once)
        const void* fns[9] = {(const void*)k_taylor_solve<64>, (const void*)k_taylor_solve<128>, (const void*)k_taylor_solve<256>,
                              (const void*)k_taylor_solve<256, 0>, (const void*)k_taylor_solve<512>,
                              (const void*)k_taylor_solve_v<128>, (const void*)k_taylor_solve_v<256>,
                              (const void*)k_taylor_solve_v<256, 0>, (const void*)k_taylor_solve_v<512>};
        bool ok = true;
        for (const void* fn : fns)
            ok = (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_SOLVE_CACHE_MAX * 32) == hipSuccess) && ok;
        if (!ok) {
            // the runtime refuses that much dynamic LDS: the solve runs without the cache
            (void)hipGetLastError();
            c->solve_cache = 0;
        }
        c->solve_lds_attr = true;
    }
    if (lds > 48 * 1024 && c->solve_cache == 0) { fa.solve_cache = 0; lds = 0; }
    const bool in_turn = solve_rows_in_turn(c, C, solve_nt);
    if (c->tail_virtual && solve_nt > 64) {
        // (experiments: ONE real wave per subint walks the waves of the kernel below in turn -- same bits)
        if (solve_nt == 128) hipLaunchKernelGGL(k_taylor_solve_v<128>, dim3(ns), dim3(64), lds, sp, fa);
        else if (solve_nt == 512) hipLaunchKernelGGL(k_taylor_solve_v<512>, dim3(ns), dim3(64), lds, sp, fa);
        else if (in_turn) hipLaunchKernelGGL((k_taylor_solve_v<256, 0>), dim3(ns), dim3(64), lds, sp, fa);
        else hipLaunchKernelGGL(k_taylor_solve_v<256>, dim3(ns), dim3(64), lds, sp, fa);
        return;
    }
    if (solve_nt == 64) hipLaunchKernelGGL(k_taylor_solve<64>, dim3(ns), dim3(64), lds, sp, fa);
    else if (solve_nt == 128) hipLaunchKernelGGL(k_taylor_solve<128>, dim3(ns), dim3(128), lds, sp, fa);
    else if (solve_nt == 512) hipLaunchKernelGGL(k_taylor_solve<512>, dim3(ns), dim3(512), lds, sp, fa);
    else if (in_turn) hipLaunchKernelGGL((k_taylor_solve<256, 0>), dim3(ns), dim3(256), lds, sp, fa);
    else hipLaunchKernelGGL(k_taylor_solve<256>, dim3(ns), dim3(256), lds, sp, fa);
}
// the post-fit stage: phase / DM / GM fits of up to 4096 channels hold a channel's numbers in registers over the
// passes, as few waves per subint as hold the band at 8 channels per thread; everything else pass by pass.
// Returns the width taken (64 ... 512; 0 = the pass-by-pass kernel of 256 threads).
static int finalize_threads_for(pp_ctx* c, const FitArgs& ff, int C) {
    return (ff.ncs != 3 || C > 4096 || !c->finalize_regs) ? 0
           : c->finalize_regs > 1 ? c->finalize_regs : C <= 512 ? 64 : C <= 1024 ? 128 : C <= 2048 ? 256 : 512;
}
static void finalize_launch(pp_ctx* c, const FitArgs& ff, int ns, int C, hipStream_t sp) {
    const int fnt = finalize_threads_for(c, ff, C);
    if (c->tail_virtual && fnt > 64 && C <= 8 * fnt) {
        // (experiments: ONE real wave per subint walks the waves of the kernel below in turn -- same bits)
        if (fnt == 128) hipLaunchKernelGGL((k_finalize_v<128>), dim3(ns), dim3(64), 0, sp, ff);
        else if (fnt == 256) hipLaunchKernelGGL((k_finalize_v<256>), dim3(ns), dim3(64), 0, sp, ff);
        else hipLaunchKernelGGL((k_finalize_v<512>), dim3(ns), dim3(64), 0, sp, ff);
    } else
    if (fnt == 64 && C <= 512) hipLaunchKernelGGL((k_finalize<8, 64>), dim3(ns), dim3(64), 0, sp, ff);
    else if (fnt == 128 && C <= 1024) hipLaunchKernelGGL((k_finalize<8, 128>), dim3(ns), dim3(128), 0, sp, ff);
    else if (fnt == 256 && C <= 2048) hipLaunchKernelGGL((k_finalize<8, 256>), dim3(ns), dim3(256), 0, sp, ff);
    else if (fnt == 512) hipLaunchKernelGGL((k_finalize<8, 512>), dim3(ns), dim3(512), 0, sp, ff);
    else hipLaunchKernelGGL((k_finalize<0, 256>), dim3(ns), dim3(256), 0, sp, ff);
}
// which body tail_work runs for the post-fit stage: the width the stand-alone launch above takes
static int finalize_width_taken(pp_ctx* c, const FitArgs& ff, int C) {
    const int fnt = finalize_threads_for(c, ff, C);
    if (fnt == 64 && C <= 512) return 64;
    if (fnt == 128 && C <= 1024) return 128;
    if (fnt == 256 && C <= 2048) return 256;
    if (fnt == 512) return 512;
    return 0;
}
// what follows the post-fit stage of a batch: its packed outputs on their way to the staging block (and the
// per-channel / objective arrays to the caller's host arrays, when asked for)
static int queue_outputs(pp_ctx* c, int stage, const pp_fit_out* out, int s0, int ns, int C, bool chan_dev, size_t copy_bytes,
                         hipStream_t sp) {
    pp_ctx::Stage& sg = c->stage[stage];
    pp_ctx::WorkSet& W = c->work[stage];
    const size_t nc = (size_t)ns * C;
    int rc;
#define PP_D2H(dst, buf, off, bytes) \
    if (dst) HIP_TRY(hipMemcpyAsync((char*)(dst) + (off), (buf).p, (bytes), hipMemcpyDeviceToHost, sp))
    if ((rc = staged_copy(c, sg.o_host, W.o_pack.p, copy_bytes, hipMemcpyDeviceToHost, sp))) return fail(rc, "output block copy failed");
    if (!chan_dev) {
        PP_D2H(out->scales, W.o_scales, (size_t)s0 * C * 8, nc * 8);
        PP_D2H(out->scale_errs, W.o_serrs, (size_t)s0 * C * 8, nc * 8);
        PP_D2H(out->channel_snrs, W.o_csnr, (size_t)s0 * C * 8, nc * 8);
    }
    PP_D2H(out->obj_f, W.o_f0, (size_t)s0 * 8, (size_t)ns * 8);
    PP_D2H(out->obj_grad, W.o_g0, (size_t)s0 * 40, (size_t)ns * 40);
    PP_D2H(out->obj_hess, W.o_H0, (size_t)s0 * 200, (size_t)ns * 200);
#undef PP_D2H
    return PP_OK;
}
// the pending tail by the stand-alone kernels (nobody carried it): solve, post-fit stage, outputs, the batch's event
static int flush_tail_queue(pp_ctx* c, pp_ctx::PendingTail& t) {
    int rc;
    if (t.rs.on) {
        // the reference's guess from the pass's chunk partials: k_refseed_finish -> k_fps -> k_refseed_start
        constexpr int M = 1024;
        const size_t H = (size_t)M + 1;
        Prof pr(c, KF_FPS);
        hipLaunchKernelGGL(k_refseed_finish, dim3((unsigned)((H + 255) / 256), t.ns), dim3(256), 0, c->stream,
                           t.rs.part, t.rs.ncc, t.rs.delta, t.rs.wsum, t.ns, t.rs_dspec, t.rs.mws);
        FpsArgs f = t.rs.fps;
        f.spec = t.rs_dspec; f.specm = t.rs.mspec; f.mstride = t.rs.mstride;
        hipLaunchKernelGGL(k_fps, dim3(t.ns), dim3(256), 0, c->stream, f, t.rs_xwork);
        hipLaunchKernelGGL(k_refseed_start, dim3((t.ns + 63) / 64), dim3(64), 0, c->stream, (const double*)t.rs.fps.out7, t.ns,
                           t.rs.xs, t.rs.seed_phase);
        HIP_TRY(hipGetLastError());
    }
    {
        Prof pr(c, KF_TAYLOR);
        solve_launch(c, t.fa, t.ns, t.C, t.solve_nt, c->stream);
    }
    {
        Prof pr(c, KF_FINAL);
        finalize_launch(c, t.fa, t.ns, t.C, c->stream);
    }
    HIP_TRY(hipGetLastError());
    if ((rc = queue_outputs(c, t.stage, &t.out, t.s0, t.ns, t.C, t.chan_dev, t.copy_bytes, c->stream))) return rc;
    HIP_TRY(hipEventRecord(c->stage[t.stage].done, c->stream));
    if (c->eager_flush) (void)hipStreamQuery(c->stream);
    return PP_OK;
}
// (a tail that could not be queued: the batch it belongs to must not be collected as if it had run -- its `done`
// event would be a stale one and its staging block another batch's numbers)
static void fail_pending_stage(pp_ctx* c, int stage, int rc);
static int flush_tail(pp_ctx* c) {
    pp_ctx::PendingTail& t = c->ptail;
    if (!t.valid) return PP_OK;
    const int rc = flush_tail_queue(c, t);
    t.valid = false;
    if (rc) fail_pending_stage(c, t.stage, rc);
    return rc;
}

static void fail_pending_stage(pp_ctx* c, int stage, int rc) {
    for (auto& d : c->pending)
        if (d.queued && d.stage == stage) { d.queued = false; d.rc = rc; d.err = g_err; }
}

// `deferred` (pp_fit_enqueue): when non-null and the batch takes the one-pass flow without a host
// decision in its middle, everything is queued -- outputs on their way to the staging block included --
// and the call returns WITHOUT waiting (*deferred = true); pp_fit_collect finishes it.
static int fit_chunk(pp_ctx* c, const pp_fit_in* in, pp_fit_out* out, int s0, int ns, int Kt, bool scat,
                     const std::vector<double>& nufit_h, const std::vector<double>& nuout_h, bool* deferred = nullptr) {
    pp_ctx::Stage& sg = c->stage[c->cur_stage];
    pp_ctx::WorkSet& W = c->work[c->cur_stage];
    const pp_seed_ref* rs = in->ref_seed;        // (applicability was checked by the caller)
    const bool refseed = (rs != nullptr);
    const int seed_ns = refseed ? rs->Ns : in->seed_ns;
    const int C = in->nchan, B = in->nbin;
    // row lengths without a tuned plan (pp_anybin.h): the slot's spectrum rows are pitched to Mp
    const bool anyb = !nbin_ok(B);
    const int Mp_any = c->slots[in->model_slot ? in->model_slot[s0] : 0].Mp;
    const int M = anyb ? Mp_any : B / 2;
    const size_t esz = in->data_dtype == PP_F64 ? 8 : 4;
    int rc;
    const cplx* tw = nullptr;
    if ((rc = get_twiddles(c, B, &tw))) return rc;
    // ---- inputs ----
    const void* ddata;
    const size_t sub_elems = (size_t)C * B;
    if (in->data_on_device) {
        ddata = (const char*)in->data + (size_t)s0 * sub_elems * esz;
    } else {
        if ((rc = upload(c, c->data, (const char*)in->data + (size_t)s0 * sub_elems * esz, (size_t)ns * sub_elems * esz))) return rc;
        ddata = c->data.p;
    }
    const size_t nc = (size_t)ns * C;
    // the small inputs travel in ONE copy from a pinned staging block (seven separate pageable
    // copies cost more than the solve of a 512 x 1024 batch): freqs | P | x0 | nu_fit | nu_out | slot
    const size_t nfreq = in->freqs_stride ? nc : (size_t)C;
    // (reference-seed flow: + the model profile(s), nu_mean and the start points' host-formed part)
    const size_t rs_nprof = refseed ? (rs->model_prof_stride ? (size_t)ns : 1) : 0;
    const size_t rs_doubles = refseed ? rs_nprof * B + (size_t)ns * (1 + 5) : 0;
    const size_t in_doubles = nfreq + (size_t)ns * (1 + 5 + 3 + 3) + rs_doubles;
    const size_t in_bytes = (in_doubles * 8 + (size_t)ns * 4 + 7) & ~(size_t)7;      // (whole 8-byte words: staged_copy)
    if ((rc = W.inpack.reserve(in_bytes))) return rc;
    if (sg.in_cap < in_bytes) {
        if (sg.in_host) (void)hipHostFree(sg.in_host);
        sg.in_host = nullptr; sg.in_cap = 0;
        HIP_TRY(hipHostMalloc(&sg.in_host, in_bytes, hipHostMallocDefault));
        sg.in_cap = in_bytes;
    }
    double* const d_freqs = W.inpack.as<double>();
    double* const d_P = d_freqs + nfreq;
    double* const d_x0 = d_P + ns;
    double* const d_nufit = d_x0 + (size_t)ns * 5;
    double* const d_nuout = d_nufit + (size_t)ns * 3;
    double* const d_rs_mprof = d_nuout + (size_t)ns * 3;
    double* const d_rs_numean = d_rs_mprof + rs_nprof * B;
    double* const d_rs_xs = d_rs_numean + (refseed ? ns : 0);
    int* const d_slot = reinterpret_cast<int*>(d_nuout + (size_t)ns * 3 + rs_doubles);
    {
        double* h = reinterpret_cast<double*>(sg.in_host);
        memcpy(h, in->freqs + (in->freqs_stride ? (size_t)s0 * C : 0), nfreq * 8); h += nfreq;
        memcpy(h, in->P + s0, (size_t)ns * 8); h += ns;
        memcpy(h, in->init_params + (size_t)s0 * 5, (size_t)ns * 40);
        // (reference seed of a scattering fit: the pass rotates by the DM guess alone -- phase 0 --
        // and the phases the reference's guess gives are written before the state is set)
        if (refseed && scat) for (int i = 0; i < ns; ++i) h[(size_t)i * 5] = 0.0;
        h += (size_t)ns * 5;
        memcpy(h, nufit_h.data() + (size_t)s0 * 3, (size_t)ns * 24); h += (size_t)ns * 3;
        memcpy(h, nuout_h.data() + (size_t)s0 * 3, (size_t)ns * 24); h += (size_t)ns * 3;
        if (refseed) {
            memcpy(h, rs->model_profs + (rs->model_prof_stride ? (size_t)s0 * B : 0), rs_nprof * B * 8); h += rs_nprof * B;
            memcpy(h, rs->nu_mean + s0, (size_t)ns * 8); h += ns;
            // phase_transform(phi, DM, nu_mean, nu_fit, P, mod=True) (pplib.py:2592-2616): the term it adds
            // depends on the inputs alone -- formed here in NumPy's order of operations with libm's pow, as the
            // reference forms it -- so the device only adds it to its fit_phase_shift result and wraps: no
            // host round trip between the pass and the iteration.  The other parameters start as given.
            for (int i = 0; i < ns; ++i) {
                const double* x0i = in->init_params + (size_t)(s0 + i) * 5;
                const double P = in->P[s0 + i], nu1 = rs->nu_mean[s0 + i], nu2 = nufit_h[(size_t)(s0 + i) * 3];
                h[(size_t)i * 5] = PP_DCONST * x0i[1] * pow(P, -1.0) * (pow(nu2, -2.0) - pow(nu1, -2.0));
                for (int j = 1; j < 5; ++j) h[(size_t)i * 5 + j] = x0i[j];
            }
            h += (size_t)ns * 5;
        }
        if (in->model_slot) memcpy(h, in->model_slot + s0, (size_t)ns * 4);
        if ((rc = staged_copy(c, W.inpack.p, sg.in_host, in_bytes, hipMemcpyHostToDevice))) return fail(rc, "input block copy failed");
    }
    const double* d_errs = nullptr;
    const unsigned char* d_mask = nullptr;
    if (in->aux_on_device) {
        if (in->errs) d_errs = in->errs + (size_t)s0 * C;
        if (in->chan_mask) d_mask = in->chan_mask + (size_t)s0 * C;
    } else {
        if (in->errs) { if ((rc = upload(c, c->errs, in->errs + (size_t)s0 * C, nc * 8))) return rc; d_errs = c->errs.as<double>(); }
        if (in->chan_mask) { if ((rc = upload(c, c->mask, in->chan_mask + (size_t)s0 * C, nc))) return rc; d_mask = c->mask.as<unsigned char>(); }
    }
    // rows in use: channels the mask removes from a subint are not transformed at all (RowWalk)
    const unsigned* mw_main = nullptr;
    const unsigned* mw_sub = nullptr;
    if (d_mask && c->skip_masked) {
        const size_t nw_main = (nc + 31) / 32 + 1, nw_sub = (C % 32 == 0) ? nc / 32 : 0;
        if ((rc = W.mwords.reserve((nw_main + nw_sub) * sizeof(unsigned)))) return rc;
        HIP_TRY(hipMemsetAsync(W.mwords.p, 0, (nw_main + nw_sub) * sizeof(unsigned), c->stream));
        unsigned* wm = W.mwords.as<unsigned>();
        unsigned* ws = nw_sub ? wm + nw_main : nullptr;
        hipLaunchKernelGGL(k_mask_words, dim3((unsigned)((C + 255) / 256), (unsigned)((ns + 31) / 32)), dim3(256), 0, c->stream,
                           d_mask, ns, C, wm, ws);
        HIP_TRY(hipGetLastError());
        mw_main = wm; mw_sub = ws;
    }
    // ---- work ----
    // no scattering: one pass over the data that leaves a Taylor model of every
    // channel + a solve on it replace the evaluation loop (fallback: the loop below,
    // on the subints that need it); otherwise evaluate as usual
    const bool taylor = !scat && c->max_iter > 0 && c->use_taylor;
    const bool seeded = seed_ns > 0;
    // Phase seed.  The Taylor flow wants the phase BEFORE its single pass, so the seed
    // comes from a pilot pass over every cstep-th channel (1/cstep of the rows and of
    // the bytes), certified by the significance of its correlation peak; subints whose
    // pilot seed is not convincing are seeded from all their channels.  Scattering
    // fits store the whole cross-spectrum anyway and seed from it.
    // (reference-seed flow: the pilot only supplies the expansion point of the Taylor model, which the
    // certificate guards -- wide bands take every 64th channel, a quarter of the pilot's rows)
    const int rstride = c->refseed_stride > 0 ? c->refseed_stride : 64;
    const int cstep = (refseed && C / rstride >= 32) ? std::max(rstride, c->seed_chan_stride) : std::max(1, c->seed_chan_stride);
    const bool pilot = seeded && taylor && c->moments_in_xspec && cstep > 1 && C / cstep >= 16;
    const bool seed_full = seeded && !pilot;
    // scattering fits of 2048-bin rows (template cut 2 Kt < M): the transform built on the
    // one-exchange FFT stores the cross-spectrum and takes the first evaluation's nine sums
    // while X is in registers (k_xspec_qs1024) -- one pass over the stored cross-spectrum fewer
    // Newton solver on a scattering fit: the answer does not depend on the path, so the iteration is first
    // run on every 16th channel -- a sixteenth of every evaluation pass over the stored cross-spectrum --
    // and the full-channel iteration starts from there: two or three full passes instead of five or six
    constexpr int kCoarseStep = 16;
    const bool coarse = scat && in->method == PP_METHOD_NEWTON && !seeded && !refseed && c->max_iter > 0 &&
                        c->coarse_newton && C / kCoarseStep >= 32 && !(c->x_f32 > 0);
    const bool fuse_scat = scat && !seeded && !coarse && c->max_iter > 0 && c->one_exchange && c->fuse_scat && !anyb && B == 2048 &&
                           2 * Kt < M && !(c->x_f32 > 0) && in->errs != nullptr;
    const bool fuse = (!scat && !taylor && !seeded) || fuse_scat;   // first evaluation folded into the transform
    // k_xspec mode: 2/3 = Taylor model only, no cross-spectrum stored;
    // 2 while every thread owns single harmonics (2 Kt < M), else 3 (pairs k, M-k)
    const bool xmom = taylor && c->moments_in_xspec && !seed_full;
    const int xmode = xmom ? (2 * Kt < M ? 2 : 3) : (fuse ? 1 : 0);
    const bool xstore = (xmode < 2);
    // a batch of the one-pass flow without a host decision in its middle may be left queued (pp_fit_enqueue); its
    // solve and post-fit stage then go to the context's second stream, behind an event of the transform, so that
    // they run BESIDE the next batch's transform (queued on `stream` right behind this one's) instead of in front of
    // it: the solve re-reads the Taylor rows at the HBM roofline with the SIMDs idle, the transform is bound by
    // instruction issue with bandwidth to spare -- and the persistent transform draws its rows by ticket, so its
    // workgroups may start as the solve's retire
    const bool defer_ok = deferred && taylor && !seed_full && (!pilot || refseed);
    const hipStream_t sp = (defer_ok && c->overlap_post && c->stream2) ? c->stream2 : c->stream;
    c->last_post = sp;
    const int ncs = scat ? PP_NCS : 3;
    // Channel chunks of the kernels that sum over channels (evaluators, seed, moments).  The run length is a
    // function of the BAND alone -- never of how many subints share the launch -- so that the partial sums of a
    // subint are formed and added in one order whatever else is in the batch: a subint's answer is a function of
    // that subint alone, as in the reference's loop (pptoas.py:344-489).  (Until round 4 the number of chunks
    // grew as the batch shrank, to fill the chip with a single subint: the rounding of f then depended on the
    // batch, and through SciPy's 1-ulp exit tests so did ~1e-9 rot of some answers.)
    auto chunking = [&](int nch, int& nchunk_, int& cpc_) {
        cpc_ = nch >= PP_CHUNK_CHANNELS ? PP_CHUNK_CHANNELS : ((nch + 15) / 16) * 16;
        nchunk_ = (nch + cpc_ - 1) / cpc_;
    };
    int nchunk, cpc;
    chunking(C, nchunk, cpc);
    // pitch of the stored cross-spectrum's rows: Kt harmonics + an optional pad (option x_pad, elements).
    // Kt x 16 B is a multiple of 1 KB and the evaluators stream 32 rows per workgroup at the same
    // pace, which looked like a recipe for memory-channel camping: measured, pads of 8 / 16 / 48
    // elements change nothing (profiles/README.md, round 3) -- the default stays 0
    const size_t Xs = (size_t)Kt + (size_t)std::max(0, c->x_pad);
    if (xstore) if ((rc = c->X.reserve(nc * Xs * sizeof(cplx)))) return rc;
    if ((rc = W.sdraw.reserve(nc * 8))) return rc;
    if ((rc = W.noise.reserve(nc * 8))) return rc;
    if ((rc = W.wts.reserve(nc * 8))) return rc;
    if ((rc = W.state.reserve((size_t)ns * sizeof(SubState)))) return rc;
    if ((rc = W.csum.reserve(2 * nc * ncs * 8))) return rc;
    if ((rc = W.partial.reserve((size_t)ns * nchunk * PP_NACC * 8))) return rc;
    if (taylor) if ((rc = W.tay.reserve(((nc + 63) / 64) * 64 * PP_TSTRIDE * 8))) return rc;   // (whole blocks of 64 rows: tay_idx)
    // (SciPy's trust-ncg spends ~8 of its ~15 evaluations inside the model's range;
    // the Newton iteration only 2-3 of 6, less than the model pass costs)
    const bool smodel = scat && c->max_iter > 0 && (c->scat_model >= 2 || (c->scat_model == 1 && in->method == PP_METHOD_TRUST_NCG));
    if (smodel) if ((rc = W.mdl.reserve(nc * PP_MROW * 8))) return rc;
    // Option x_f32 (off by default): the stored cross-spectrum of a scattering fit kept as float
    // pairs, half the bytes of every evaluation pass.  Measured on configs[3] with the Newton
    // solver (profiles/README.md, round 3, with the first evaluator, which was not HBM-bound: the
    // six passes went from 7.92 to 7.28 ms only; k_eval_scat is, so the gain would be larger now)
    // -- but chi2 loses its 1e-10 agreement with the reference (6e-8 of every |X_nk|
    // moves f by ~1e-6 of itself; the optimum by ~1e-11 rot).  Kept for experiments.
    const bool xf32 = scat && !seeded && !smodel && c->max_iter > 0 && c->x_f32 > 0;
    const bool want_ph0 = xmode != 0 || fuse_scat || refseed;
    if (want_ph0) if ((rc = W.ph0.reserve(nc * 8 * (fuse_scat ? 2 : 1)))) return rc;
    if ((rc = W.misc.reserve(256))) return rc;
    if ((rc = W.act.reserve((size_t)ns * 4))) return rc;
    // per-subint scalar outputs: blocks of one allocation (params 5, errs 5, nu 3,
    // cov 25, chi2, red_chi2, snr doubles; nfeval, return_code, npass ints) = 340 B / subint
    const size_t o_bytes = ((size_t)ns * 340 + 8 + 7) & ~(size_t)7;       // (+ the count of unfinished subints; whole words)
    const size_t o_stage = stage_seed_offset(ns) + (size_t)ns * 8;   // (+ the reference-seed flow's phase guesses)
    if ((rc = W.o_pack.reserve(o_stage))) return rc;      // (the phase guesses of the reference-seed flow behind the pack)
    if (sg.o_cap < o_stage) {
        if (sg.o_host) (void)hipHostFree(sg.o_host);
        sg.o_host = nullptr; sg.o_cap = 0;
        HIP_TRY(hipHostMalloc(&sg.o_host, o_stage, hipHostMallocDefault));
        sg.o_cap = o_stage;
    }
    double* const o_base = W.o_pack.as<double>();
    if ((rc = W.o_f0.reserve((size_t)ns * 8))) return rc;
    if ((rc = W.o_g0.reserve((size_t)ns * 40))) return rc;
    if ((rc = W.o_H0.reserve((size_t)ns * 200))) return rc;
    const bool chan_dev = out->chan_on_device != 0;
    if (!chan_dev) {
        if (out->scales) if ((rc = W.o_scales.reserve(nc * 8))) return rc;
        if (out->scale_errs) if ((rc = W.o_serrs.reserve(nc * 8))) return rc;
        if (out->channel_snrs) if ((rc = W.o_csnr.reserve(nc * 8))) return rc;
    }

    if (c->debug_poison) {
        // every buffer a kernel of this batch reads must have been written by one: stale
        // contents become NaNs that surface as failed certificates / non-finite results
        // (bit mask: 1 tay, 2 sdraw, 4 noise, 8 wts, 16 csum, 32 ph0, 64 X, 128 mdl)
        const int pz = c->debug_poison;
        if ((pz & 1) && taylor) HIP_TRY(hipMemsetAsync(W.tay.p, 0xFF, ((nc + 63) / 64) * 64 * PP_TSTRIDE * 8, c->stream));
        if (pz & 2) HIP_TRY(hipMemsetAsync(W.sdraw.p, 0xFF, nc * 8, c->stream));
        if (pz & 4) HIP_TRY(hipMemsetAsync(W.noise.p, 0xFF, nc * 8, c->stream));
        if (pz & 8) HIP_TRY(hipMemsetAsync(W.wts.p, 0xFF, nc * 8, c->stream));
        if (pz & 16) HIP_TRY(hipMemsetAsync(W.csum.p, 0xFF, 2 * nc * ncs * 8, c->stream));
        if ((pz & 32) && want_ph0) HIP_TRY(hipMemsetAsync(W.ph0.p, 0xFF, nc * 8, c->stream));
        if ((pz & 64) && xstore) HIP_TRY(hipMemsetAsync(c->X.p, 0xFF, nc * Xs * sizeof(cplx), c->stream));
        if ((pz & 128) && smodel) HIP_TRY(hipMemsetAsync(W.mdl.p, 0xFF, nc * PP_MROW * 8, c->stream));
    }
    // ---- argument blocks ----
    const bool tail = (in->errs == nullptr);
    XspecArgs xa;
    memset(&xa, 0, sizeof xa);
    xa.data = ddata; xa.mft = (const cplx* const*)c->mft_table.p;
    xa.mft0 = c->slots[0].mft.as<cplx>();
    xa.ktab = (const int* const*)c->kt_table.p;
    xa.kt0 = c->slots[0].kt.as<int>();
    xa.slot = in->model_slot ? d_slot : nullptr;
    xa.X = c->X.as<cplx>(); xa.sdraw = W.sdraw.as<double>(); xa.noise = W.noise.as<double>();
    xa.twB = tw; xa.nsub = ns; xa.nchan = C; xa.Kt = Kt; xa.Xs = (int)Xs;
    xa.x0 = d_x0; xa.P = d_P; xa.nu_fit = d_nufit;
    xa.freqs = d_freqs; xa.freqs_stride = in->freqs_stride ? C : 0;
    xa.csum0 = W.csum.as<double>();
    xa.tay = W.tay.as<double>();
    xa.ph0 = W.ph0.as<double>();
    xa.act = nullptr; xa.cstep = 1; xa.coff = 0; xa.nchan_full = C;
    xa.x_f32 = xf32 ? 1 : 0;
    FitArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.nsub = ns; fa.nchan = C; fa.nbin = B; fa.M = M; fa.Kt = Kt; fa.Xs = (int)Xs;
    for (int j = 0; j < 5; ++j) fa.flags[j] = in->fit_flags[j] ? 1 : 0;
    fa.log10_tau = in->log10_tau ? 1 : 0; fa.option = in->option; fa.is_toa = in->is_toa ? 1 : 0;
    fa.max_iter = c->max_iter; fa.scat = scat ? 1 : 0;
    fa.method = in->method;
    fa.X = c->X.as<cplx>();
    fa.mft = (const cplx* const*)c->mft_table.p;
    fa.msq = (const double* const*)c->msq_table.p;
    fa.msum = (const double* const*)c->msum_table.p;
    fa.ktab = (const int* const*)c->kt_table.p;
    fa.slot = in->model_slot ? d_slot : nullptr;
    fa.freqs = d_freqs; fa.freqs_stride = in->freqs_stride ? C : 0;
    fa.wts = W.wts.as<double>(); fa.sdraw = W.sdraw.as<double>();
    fa.P = d_P; fa.nu_fit = d_nufit; fa.nu_out = d_nuout;
    fa.x0 = d_x0; fa.st = W.state.as<SubState>();
    fa.csum = W.csum.as<double>(); fa.ncs = ncs;
    fa.tay = W.tay.as<double>();
    fa.partial = W.partial.as<double>(); fa.nchunk = nchunk; fa.cpc = cpc;
    fa.nactive = W.misc.as<int>();
    fa.o_params = o_base; fa.o_errs = o_base + (size_t)ns * 5; fa.o_nu = o_base + (size_t)ns * 10;
    fa.o_cov = o_base + (size_t)ns * 13; fa.o_chi2 = o_base + (size_t)ns * 38; fa.o_rchi2 = o_base + (size_t)ns * 39;
    fa.o_snr = o_base + (size_t)ns * 40;
    fa.o_nfev = reinterpret_cast<int*>(o_base + (size_t)ns * 41); fa.o_rc = fa.o_nfev + ns; fa.o_npass = fa.o_rc + ns;
    if (chan_dev) {
        fa.o_scales = out->scales ? out->scales + (size_t)s0 * C : nullptr;
        fa.o_scale_errs = out->scale_errs ? out->scale_errs + (size_t)s0 * C : nullptr;
        fa.o_csnr = out->channel_snrs ? out->channel_snrs + (size_t)s0 * C : nullptr;
    } else {
        fa.o_scales = out->scales ? W.o_scales.as<double>() : nullptr;
        fa.o_scale_errs = out->scale_errs ? W.o_serrs.as<double>() : nullptr;
        fa.o_csnr = out->channel_snrs ? W.o_csnr.as<double>() : nullptr;
    }
    fa.o_f0 = W.o_f0.as<double>(); fa.o_g0 = W.o_g0.as<double>(); fa.o_H0 = W.o_H0.as<double>();
    fa.o_rec = out->records_dev ? out->records_dev + (size_t)s0 * PP_RECORD_WIDTH : nullptr;
    fa.act = nullptr; fa.nact = ns; fa.nchan_x = C; fa.cstep = 1; fa.coff = 0;
    fa.mdl = W.mdl.as<double>(); fa.use_model = smodel ? 1 : 0; fa.model_tol = c->scat_model_tol; fa.model_bet = c->scat_model_bet;
    // (a GM fit walked the SciPy way ends where its path ends: it keeps the exact path; the
    // Newton solver converges to the optimum from anywhere)
    fa.recentre = (taylor && xmom && (!in->fit_flags[2] || in->method == PP_METHOD_NEWTON))
                      ? std::max(0, c->taylor_recentre) : 0;
    fa.x0w = d_x0;
    fa.x_f32 = xf32 ? 1 : 0;
    fa.nfev_shadow = c->nfev_shadow;
    // (a band of up to 512 channels: one wave per subint; up to 1024: two; wider: four -- eight measured slower at 4096)
    const int solve_nt = solve_threads_for(c, C);
    // (LDS: 32 B per cached channel, 512 channels per wave of the block keep the CU's eight waves within 128 KB)
    fa.solve_cache = std::min(C, c->solve_cache >= 0 ? c->solve_cache : std::min(PP_SOLVE_CACHE_MAX, solve_nt * 8));
    {
        // what the device grants a workgroup (160 KB on gfx950; 64 KB on older parts) less the kernel's static scratch
        // bounds the cache: a smaller cache only means more channels' invariants formed again per evaluation
        if (!c->max_lds_bytes) {
            hipDeviceProp_t prop;
            c->max_lds_bytes = (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.sharedMemPerBlock > 0)
                                   ? (int)std::min<size_t>(prop.sharedMemPerBlock, (size_t)1 << 30) : 64 * 1024;
        }
        const int room = std::max(0, c->max_lds_bytes - 16 * 1024) / 32;
        fa.solve_cache = std::min(fa.solve_cache, room);
    }
    auto launch_taylor_solve = [&]() { solve_launch(c, fa, ns, C, solve_nt, sp); };

    auto run_xspec = [&](const XspecArgs& x, int mode) -> int {
        Prof pr(c, KF_XSPEC);
        if (anyb) return launch_any(c, x, B, M, in->data_dtype, mode, tail, nullptr, c->skip_masked ? d_mask : nullptr);
        PP_DISPATCH_M(M, {
            if (in->data_dtype == PP_F64) launch_xspec<MM, double>(c, x, tail, mode);
            else launch_xspec<MM, float>(c, x, tail, mode);
        });
        HIP_TRY(hipGetLastError());
        return PP_OK;
    };
    auto run_prep = [&]() -> int {
        Prof pr(c, KF_PREP);
        hipLaunchKernelGGL(k_prep, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, ns, C, B,
                           d_errs, W.noise.as<double>(), d_mask, W.wts.as<double>());
        HIP_TRY(hipGetLastError());
        return PP_OK;
    };
    // the coarse seed uses the lowest 64 * PP_SEED_KPT = 1024 harmonics at most
    // (templates that keep more -- nbin 4096 / 8192 with power out to Nyquist --
    // lose nothing a 100-point grid could resolve)
    const int Ks = std::min(Kt, 64 * PP_SEED_KPT);
    // seed the subints f lists (f.act / f.nact, channels f.coff + nn f.cstep) from the
    // cross-spectrum in f.X; seedq (optional) receives the peak significance
    // The coarse grid is (phi, DM): seed_ns phases x seed_ndm trial DMs spaced
    // seed_dm_step about the guess (1 trial = the reference's behaviour, whose seed
    // trusts the header DM, pptoas.py:421-457); the best correlation peak wins.
    const int ndm = (c->seed_ndm > 1 && c->seed_dm_step > 0.0) ? c->seed_ndm : 1;
    auto run_seed = [&](const FitArgs& f, double* seedq) -> int {
        if ((rc = c->seedbuf.reserve(((size_t)f.nact * f.nchunk + f.nact) * Ks * sizeof(cplx)))) return rc;
        cplx* ypart = c->seedbuf.as<cplx>();
        cplx* ywork = ypart + (size_t)f.nact * f.nchunk * Ks;
        Prof pr(c, KF_SEED);
        const double* xbase = d_x0;
        if (ndm > 1) {
            // trial DMs: peak heights only, then the refined DM of every subint, then
            // the seed proper at that DM
            if ((rc = c->xbase.reserve((size_t)ns * 80 + (size_t)ns * ndm * 8))) return rc;
            double* xb = c->xbase.as<double>();          // [ns][5] guesses as given
            double* xr = xb + (size_t)ns * 5;            // [ns][5] with the chosen DM
            double* pk = xr + (size_t)ns * 5;            // [ns][ndm]
            HIP_TRY(hipMemcpyAsync(xb, d_x0, (size_t)ns * 40, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(xr, d_x0, (size_t)ns * 40, hipMemcpyDeviceToDevice, c->stream));
            for (int t = 0; t < ndm; ++t) {
                const double off = (t - (ndm - 1) / 2) * c->seed_dm_step;
                hipLaunchKernelGGL(k_seed_accum, dim3(f.nact, f.nchunk), dim3(256), 0, c->stream, f, ypart, Ks,
                                   (const double*)xb, off);
                hipLaunchKernelGGL(k_seed_fit, dim3(f.nact), dim3(256), 0, c->stream, f, (const cplx*)ypart, ywork,
                                   d_x0, seed_ns, Ks, (double*)nullptr, (const double*)xb, off,
                                   pk, t, ndm);
            }
            hipLaunchKernelGGL(k_seed_dm_pick, dim3((f.nact + 63) / 64), dim3(64), 0, c->stream, f.act, f.nact,
                               (const double*)pk, ndm, c->seed_dm_step, (const double*)xb, xr);
            xbase = xr;
        }
        hipLaunchKernelGGL(k_seed_accum, dim3(f.nact, f.nchunk), dim3(256), 0, c->stream, f, ypart, Ks, xbase, 0.0);
        hipLaunchKernelGGL(k_seed_fit, dim3(f.nact), dim3(256), 0, c->stream, f, (const cplx*)ypart, ywork,
                           d_x0, seed_ns, Ks, seedq, xbase, 0.0, (double*)nullptr, 0, 1);
        HIP_TRY(hipGetLastError());
        return PP_OK;
    };
    // list the subints that need more work into W.act; returns their number
    auto list_active = [&](const double* seedq, double qmin, int* count) -> int {
        hipLaunchKernelGGL(k_list_active, dim3(1), dim3(256), 0, c->stream, (const SubState*)W.state.p, seedq, qmin,
                           ns, W.act.as<int>(), W.misc.as<int>() + 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->nactive_h + 1, W.misc.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        *count = c->nactive_h[1];
        return PP_OK;
    };
    // transform every channel of the listed subints again with the cross-spectrum
    // stored (compact in the list index), and point the evaluators at the list
    auto store_x_for_list = [&](int count) -> int {
        if ((rc = c->X.reserve((size_t)count * C * Xs * sizeof(cplx)))) return rc;
        XspecArgs xl = xa;
        xl.X = c->X.as<cplx>(); xl.act = W.act.as<int>(); xl.nsub = count;
        if ((rc = run_xspec(xl, 0))) return rc;
        fa.X = c->X.as<cplx>(); fa.act = W.act.as<int>(); fa.nact = count;
        chunking(C, fa.nchunk, fa.cpc);
        if ((rc = W.partial.reserve((size_t)ns * fa.nchunk * PP_NACC * 8))) return rc;
        fa.partial = W.partial.as<double>();
        return PP_OK;
    };

    // (a pending tail of the previous enqueued batch that this batch's transform will not carry -- any flow but the
    // plain one-pass one -- goes out by the stand-alone kernels now)
    if (c->ptail.valid && ((refseed && scat) || seed_full || fuse_scat || !xmom || anyb || !c->one_exchange || M != 1024 || !deferred))
        if ((rc = flush_tail(c))) return rc;
    // The previous enqueued batch's tail, still unqueued: this batch's transform works it off as tickets if it is one of
    // the kernels that can (k_xspec_q1024 / k_xspec_qf<1024> / k_xspec_qr1024: 2048-bin rows, Taylor sums only) -- and
    // if its grid is wide enough for the tickets: a small batch behind a large one would hand each of its few waves
    // many tickets in a row, ~1 ms each, where the stand-alone kernels take 0.7 ms for all of them.
    auto carrier_block = [&](XspecArgs& x) -> int {
        pp_ctx::PendingTail& pt = c->ptail;
        const int st_i = c->cur_stage;
        const size_t tb = (sizeof(TailArgs) + 7) & ~(size_t)7;
        if ((rc = c->tailbuf[st_i].reserve(tb))) return rc;
        if (!c->tail_host[st_i]) HIP_TRY(hipHostMalloc(&c->tail_host[st_i], tb, hipHostMallocDefault));
        TailArgs* th = reinterpret_cast<TailArgs*>(c->tail_host[st_i]);
        memset(th, 0, tb);
        th->fa = pt.fa; th->ticket = 0; th->done = 0; th->nsub = pt.ns;
        th->fa.solve_cache = std::min(pt.C, (int)PP_TAIL_CACHE);
        th->fa.tail_fused = 1;
        th->solve_nt = pt.solve_nt; th->solve_pf = pt.solve_pf0 ? 0 : PP_SOLVE_PF; th->fin_nt = pt.fin_nt;
        th->rs = pt.rs;
        if ((rc = staged_copy(c, c->tailbuf[st_i].p, th, tb, hipMemcpyHostToDevice))) return fail(rc, "tail block copy failed");
        x.tail = c->tailbuf[st_i].as<TailArgs>();
        x.tail_nsub = pt.ns;
        return PP_OK;
    };
    // ... and behind the carrying transform: the carried batch's outputs and its event
    auto carrier_done = [&]() -> int {
        pp_ctx::PendingTail& pt = c->ptail;
        int r2 = queue_outputs(c, pt.stage, &pt.out, pt.s0, pt.ns, pt.C, pt.chan_dev, pt.copy_bytes, c->stream);
        if (!r2 && hipEventRecord(c->stage[pt.stage].done, c->stream) != hipSuccess) r2 = fail(PP_EHIP, "hipEventRecord failed");
        pt.valid = false;
        if (r2) fail_pending_stage(c, pt.stage, r2);
        return r2;
    };
    const bool wide_enough = c->ptail.valid && (long long)std::min<long long>((long long)ns * C, 4096) * 2 >= (long long)c->ptail.ns;
    if (c->ptail.valid && !wide_enough)
        if ((rc = flush_tail(c))) return rc;
    // a batch of the one-pass flow whose own tail stays unqueued for the next batch's transform (option fuse_tail)
    const bool tail_pending = defer_ok && c->fuse_tail && xmom && sp == c->stream && !anyb && M == 1024 && c->one_exchange;
    // ---- phase seed from a pilot pass ----
    if (pilot) {
        const int Cp = (C + cstep - 1) / cstep;
        if ((rc = c->X.reserve((size_t)ns * Cp * Xs * sizeof(cplx)))) return rc;
        if ((rc = c->seedq.reserve((size_t)ns * 8))) return rc;
        XspecArgs xp = xa;
        xp.X = c->X.as<cplx>(); xp.nchan = Cp; xp.cstep = cstep;
        if ((rc = run_xspec(xp, 0))) return rc;
        if ((rc = run_prep())) return rc;      // (rows not transformed yet have no measured noise: unused here)
        FitArgs fp = fa;
        fp.X = c->X.as<cplx>(); fp.nchan_x = Cp; fp.cstep = cstep;
        chunking(Cp, fp.nchunk, fp.cpc);
        if ((rc = run_seed(fp, c->seedq.as<double>()))) return rc;
        // (reference-seed flow: the pilot's phase is only the expansion point of the Taylor model, which the
        // certificate guards -- a weak pilot costs its subint a second expansion, not the batch a host round trip)
        int nweak = 0;
        if (!refseed) if ((rc = list_active(c->seedq.as<double>(), c->seed_min_snr, &nweak))) return rc;
        if (nweak > 0) {
            // not convincing on a subset: seed these from all their channels
            if ((rc = store_x_for_list(nweak))) return rc;
            if ((rc = run_prep())) return rc;
            if ((rc = run_seed(fa, nullptr))) return rc;
            fa.act = nullptr; fa.nact = ns; fa.nchunk = nchunk; fa.cpc = cpc;
        }
    }

    // ---- rFFT + cross-spectrum (or the Taylor model) of every row ----
    // one launch in front of it: phi_n at the initial parameters, the weights when the noise is
    // given, the solver state (a full seed rewrites x0 after the transform: state set then)
    const bool wts_early = (d_errs != nullptr);
    {
        Prof pr(c, KF_PREP);
        hipLaunchKernelGGL(k_setup, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, fa,
                           wts_early ? d_errs : (const double*)nullptr, d_mask, W.wts.as<double>(),
                           want_ph0 ? W.ph0.as<double>() : (double*)nullptr,
                           fuse_scat ? W.ph0.as<double>() + nc : (double*)nullptr, seed_full ? 0 : 1);
    }
    // ---- reference-seed flow: Taylor model about the pilot's phase + the rotated channel sums in
    // one pass, then the reference's fit_phase_shift on the channel mean, then the start points
    double* d_seedph = nullptr;      // [ns] the phase guesses the reference-seed flow formed (fetched with the outputs)
    // hand-over from the transform stage (on `stream`) to the solve / post-fit stage (on `sp`), once per batch
    bool chained = false;
    auto chain_post = [&]() -> int {
        if (sp == c->stream || chained) return PP_OK;
        HIP_TRY(hipEventRecord(W.xdone, c->stream));
        HIP_TRY(hipStreamWaitEvent(sp, W.xdone, 0));
        if (c->eager_flush) (void)hipStreamQuery(c->stream);
        chained = true;
        return PP_OK;
    };
    RefTailArgs rs_tail;             // this batch's own reference-seed tail, when it stays unqueued (tail_pending)
    memset(&rs_tail, 0, sizeof rs_tail);
    cplx* rs_tail_dspec = nullptr;
    cplx* rs_tail_xwork = nullptr;
    auto run_refseed_pass = [&]() -> int {
        const int ncc = C / PP_ROW_CHUNK;
        const size_t H = (size_t)M + 1;
        const size_t nprof = rs->model_prof_stride ? (size_t)ns : 1;
        const bool w_host = (rs->weights && !in->aux_on_device) || d_mask;     // (room for the masked weights)
        const size_t n_part = (size_t)ns * ncc * RS_NACC * 64;
        const size_t n_cplx = n_part + (size_t)ns * H + nprof * H + (size_t)ns * M;
        const size_t n_dbl = (size_t)ns * (1 + 1 + 7) + (w_host ? nc : 0);
        if ((rc = W.refbuf.reserve(n_cplx * sizeof(cplx) + n_dbl * 8))) return rc;
        cplx* part = W.refbuf.as<cplx>();
        cplx* dspec = part + n_part;
        cplx* mspec = dspec + (size_t)ns * H;
        cplx* xwork = mspec + nprof * H;
        // (the model profile(s), nu_mean and the host-formed part of the start points came with the batch's one
        // input block: no copy command of their own)
        double* mprof = d_rs_mprof;
        double* d_numean = d_rs_numean;
        double* d_xs = d_rs_xs;
        double* d_delta = reinterpret_cast<double*>(xwork + (size_t)ns * M);
        double* d_wsum = d_delta + ns;
        double* d_out7 = d_wsum + ns;
        double* d_sph = reinterpret_cast<double*>(reinterpret_cast<char*>(W.o_pack.p) + stage_seed_offset(ns));   // (leaves with the outputs)
        double* d_wh = d_out7 + (size_t)ns * 7;
        const double* d_w = nullptr;
        if (rs->weights && !in->aux_on_device) {
            HIP_TRY(hipMemcpyAsync(d_wh, rs->weights + (size_t)s0 * C, nc * 8, hipMemcpyHostToDevice, c->stream));
            d_w = d_wh;
        } else if (rs->weights) d_w = rs->weights + (size_t)s0 * C;
        if (d_mask) {
            // the channel mean is taken over the channels the fit uses
            hipLaunchKernelGGL(k_refseed_weights, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, d_w, d_mask,
                               (long long)nc, d_wh);
            d_w = d_wh;
        }
        const long long nrows = (long long)ns * C;
        XspecArgs x = xa;
        x.ticket = c->ticket.as<unsigned>();
        x.ticket_base = c->ticket_base;
        x.mwords = mw_sub;
        c->ticket_base += (unsigned)((nrows + PP_ROW_CHUNK - 1) / PP_ROW_CHUNK);
        RefSeedArgs ra{d_w, part, ncc};
        // what the guess needs beside the pass's channel sums does not depend on the pass: the template profile's
        // spectrum, Delta_i and the summed weights -- queued in front of it (the expansion points are the pilot's)
        {
            Prof pr(c, KF_PREP);
            hipLaunchKernelGGL((k_rfft_rows<1024, double>), dim3(fft_grid(64, (long long)nprof)), dim3(64), 0, c->stream,
                               (const void*)mprof, mspec, tw, (int)nprof);
            hipLaunchKernelGGL(k_refseed_prep, dim3(ns), dim3(256), 0, c->stream, (const double*)d_x0, (const double*)d_P,
                               (const double*)d_nufit, (const double*)d_numean, d_w, C, d_delta, d_wsum);
        }
        HIP_TRY(hipGetLastError());
        // (the previous batch's tail rides in this pass if it can: see carrier_block)
        const bool carrier = c->ptail.valid && !scat && deferred != nullptr;
        if (carrier) { if ((rc = carrier_block(x))) return rc; }
        else if (c->ptail.valid) { if ((rc = flush_tail(c))) return rc; }
        {
            Prof pr(c, KF_XSPEC);
#define PP_QR(TIN, ST)                                                                                     \
    do {                                                                                                   \
        const dim3 grid(resident_grid(c, k_xspec_qr1024<TIN, ST>, 64, nrows, fft_grid(64, nrows)));        \
        hipLaunchKernelGGL((k_xspec_qr1024<TIN, ST>), grid, dim3(64), 0, c->stream, x, ra);               \
    } while (0)
            if (in->data_dtype == PP_F64) { if (scat) PP_QR(double, true); else PP_QR(double, false); }
            else { if (scat) PP_QR(float, true); else PP_QR(float, false); }
#undef PP_QR
        }
        HIP_TRY(hipGetLastError());
        if (carrier) if ((rc = carrier_done())) return rc;
        FpsArgs f{dspec, nullptr, d_out7, rs->lo, rs->hi, rs->Ns, M, ns, rs->finish, mspec,
                  rs->model_prof_stride ? (int)H : 0};
        d_seedph = d_sph;
        if (tail_pending && !scat) {
            // this batch's own guess -- the spectrum from the chunk partials, the reference's fit_phase_shift, the
            // start points -- waits with its solve and post-fit stage for the next batch's transform (or flush_tail)
            memset(&rs_tail, 0, sizeof rs_tail);
            rs_tail.on = 1; rs_tail.ncc = ncc; rs_tail.part = part; rs_tail.delta = d_delta; rs_tail.wsum = d_wsum;
            rs_tail.mws = mw_sub; rs_tail.mspec = mspec; rs_tail.mstride = rs->model_prof_stride ? (int)H : 0;
            rs_tail.fps = f; rs_tail.fps.spec = nullptr; rs_tail.fps.specm = nullptr;
            rs_tail.xs = d_xs; rs_tail.seed_phase = d_sph;
            rs_tail_dspec = dspec; rs_tail_xwork = xwork;
            fa.xstart = d_xs;
            return PP_OK;
        }
        // (what follows the pass -- the channel mean's spectrum, the reference's fit_phase_shift, the start points --
        // belongs to the solve stage: beside the next batch's transform when the batch is deferred.  The mask words
        // the finish reads are the transform stage's buffer: a deferred batch has its own copy)
        if ((rc = chain_post())) return rc;
        {
            Prof pr(c, KF_FPS, sp);
            hipLaunchKernelGGL(k_refseed_finish, dim3((unsigned)((H + 255) / 256), ns), dim3(256), 0, sp,
                               (const cplx*)part, ncc, (const double*)d_delta, (const double*)d_wsum, ns, dspec, mw_sub);
            hipLaunchKernelGGL(k_fps, dim3(ns), dim3(256), 0, sp, f, xwork);
        }
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_refseed_start, dim3((ns + 63) / 64), dim3(64), 0, sp, (const double*)d_out7, ns, d_xs, d_sph);
        HIP_TRY(hipGetLastError());
        if (scat) {
            // stored cross-spectrum: the iteration starts AT the reference's guess
            HIP_TRY(hipMemcpyAsync(d_x0, d_xs, (size_t)ns * 40, hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(k_init_state, dim3((ns + 63) / 64), dim3(64), 0, c->stream, fa);
            HIP_TRY(hipGetLastError());
        } else fa.xstart = d_xs;       // Taylor model about the pilot's phase: the walk starts off-centre
        return PP_OK;
    };
    if (refseed) {
        if ((rc = run_refseed_pass())) return rc;
    } else if (seed_full) {
        // the seed needs the cross-spectrum at a phase not known yet: store it, seed,
        // then take the Taylor moments (or iterate) in a second pass over it
        XspecArgs xm = xa;
        xm.mwords = mw_main;
        if ((rc = run_xspec(xm, 0))) return rc;
        if (!wts_early) if ((rc = run_prep())) return rc;
        if ((rc = run_seed(fa, nullptr))) return rc;
        hipLaunchKernelGGL(k_init_state, dim3((ns + 63) / 64), dim3(64), 0, c->stream, fa);
    } else if (fuse_scat) {
        const long long nrows = (long long)ns * C;
        XspecArgs x = xa;
        x.ticket = c->ticket.as<unsigned>();
        x.ticket_base = c->ticket_base;
        x.mwords = mw_main;
        c->ticket_base += (unsigned)((nrows + PP_ROW_CHUNK - 1) / PP_ROW_CHUNK);
        {
            Prof pr(c, KF_XSPEC);
            if (in->data_dtype == PP_F64) {
                const dim3 grid(resident_grid(c, k_xspec_qs1024<double>, 64, nrows, fft_grid(64, nrows)));
                hipLaunchKernelGGL((k_xspec_qs1024<double>), grid, dim3(64), 0, c->stream, x,
                                   (const double*)(W.ph0.as<double>() + nc), W.csum.as<double>());
            } else {
                const dim3 grid(resident_grid(c, k_xspec_qs1024<float>, 64, nrows, fft_grid(64, nrows)));
                hipLaunchKernelGGL((k_xspec_qs1024<float>), grid, dim3(64), 0, c->stream, x,
                                   (const double*)(W.ph0.as<double>() + nc), W.csum.as<double>());
            }
        }
        if (!wts_early) if ((rc = run_prep())) return rc;
    } else {
        XspecArgs xm = xa;
        xm.mwords = mw_main;
        // (the previous enqueued batch's tail rides in this transform if it can: see carrier_block)
        const bool carrier = c->ptail.valid && xmom && !anyb && c->one_exchange && M == 1024 && deferred != nullptr;
        if (carrier) { if ((rc = carrier_block(xm))) return rc; }
        else if (c->ptail.valid) { if ((rc = flush_tail(c))) return rc; }
        if ((rc = run_xspec(xm, xmode))) return rc;
        if (carrier) if ((rc = carrier_done())) return rc;
        if (!wts_early) if ((rc = run_prep())) return rc;
    }
    HIP_TRY(hipGetLastError());
    if (c->eager_flush) (void)hipStreamQuery(c->stream);
    // ---- post-fit stage + every output in one round trip ----
    auto finalize_and_fetch = [&](bool wait = true) -> int {
        FitArgs ff = fa;
        ff.act = nullptr; ff.nact = ns;
        {
            Prof pr(c, KF_FINAL, sp);
            finalize_launch(c, ff, ns, C, sp);
        }
        HIP_TRY(hipGetLastError());
        if ((rc = queue_outputs(c, c->cur_stage, out, s0, ns, C, chan_dev, d_seedph ? o_stage : o_bytes, sp))) return rc;
        if (c->eager_flush && sp != c->stream) (void)hipStreamQuery(sp);
        if (wait) HIP_TRY(hipStreamSynchronize(sp));
        return PP_OK;
    };
    auto unpack_outputs = [&]() { unpack_stage(sg.o_host, out, s0, ns); if (d_seedph) unpack_seed_phases(sg.o_host, in, s0, ns); };
    auto unfinished = [&]() -> int { return unfinished_in_stage(sg.o_host, ns); };
    bool all_done = false;
    if (taylor) {
        if (xstore) {
            Prof pr(c, KF_EVAL);
            hipLaunchKernelGGL(k_eval_moments, dim3(ns, nchunk), dim3(256), 0, c->stream, fa);
        }
        if (tail_pending) {
            // nothing of the tail is queued: the next enqueued batch's transform works it off (or flush_tail does)
            pp_ctx::PendingTail& pt = c->ptail;
            pt.valid = true; pt.stage = c->cur_stage;
            pt.fa = fa; pt.fa.act = nullptr; pt.fa.nact = ns;
            pt.ns = ns; pt.C = C; pt.solve_nt = solve_nt; pt.solve_pf0 = solve_rows_in_turn(c, C, solve_nt) ? 1 : 0;
            pt.fin_nt = finalize_width_taken(c, pt.fa, C);
            pt.out = *out; pt.s0 = s0; pt.chan_dev = chan_dev; pt.copy_bytes = d_seedph ? o_stage : o_bytes;
            pt.rs = rs_tail; pt.rs_dspec = rs_tail_dspec; pt.rs_xwork = rs_tail_xwork;
            *deferred = true;
            return PP_OK;
        }
        if ((rc = chain_post())) return rc;
        {
            Prof pr(c, KF_TAYLOR, sp);
            // (rows of the Taylor model in registers where the channel count allows)
            launch_taylor_solve();
        }
        HIP_TRY(hipGetLastError());
        // The solve certifies nearly every subint of nearly every batch: the post-fit stage is
        // launched straight behind it and the count of unfinished subints comes back with the
        // outputs -- one host round trip per batch instead of two.  (When some are left, what
        // the post-fit stage wrote for them is overwritten below.)
        if (defer_ok) {
            // nothing left for the host to decide before the outputs are on their way: pp_fit_collect
            // looks at the count of unfinished subints (and fits the batch again, synchronously, in the
            // rare case that some are left -- their guesses were poor)
            if ((rc = finalize_and_fetch(false))) return rc;
            *deferred = true;
            return PP_OK;
        }
        if ((rc = finalize_and_fetch())) return rc;
        all_done = (unfinished() <= 0);
        if (all_done) { unpack_outputs(); return PP_OK; }
        // (reference-seed flow: a subint whose walk left the model taken about the pilot's phase is
        // expanded again about the reference's guess itself -- the ordinary flow from there)
        const int nrep = std::max(fa.recentre, refseed ? 1 : 0);
        for (int rep = 0; rep < nrep && !all_done; ++rep) {
            // some subints failed the certificate (poor guesses): k_taylor_solve moved
            // their expansion points to its tentative answers -- take the Taylor model of
            // THOSE again (one more pass over their rows, nothing stored) and solve again
            int nleft = 0;
            if ((rc = list_active(nullptr, 0.0, &nleft))) return rc;
            {
                Prof pr(c, KF_PREP);
                hipLaunchKernelGGL(k_phase0, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, ns, C,
                                   xa.x0, xa.P, xa.nu_fit, xa.freqs, xa.freqs_stride, W.ph0.as<double>());
            }
            XspecArgs xl = xa;
            xl.act = W.act.as<int>(); xl.nsub = nleft;
            if ((rc = run_xspec(xl, xmode))) return rc;
            {
                Prof pr(c, KF_TAYLOR);
                launch_taylor_solve();
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(c->nactive_h, fa.nactive, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            all_done = (c->nactive_h[0] <= 0);
        }
        if (!all_done && !xstore) {
            // still not certified: they need evaluations over the cross-spectrum, which
            // was not stored -- transform THOSE again, keeping it
            int nleft = 0;
            if ((rc = list_active(nullptr, 0.0, &nleft))) return rc;
            if ((rc = store_x_for_list(nleft))) return rc;
        }
    }
    if (coarse && !all_done) {
        FitArgs fs = fa;
        fs.cstep = kCoarseStep; fs.coff = 0; fs.nchan_x = (C + kCoarseStep - 1) / kCoarseStep; fs.x_full = 1;
        fs.use_model = 0;
        chunking(fs.nchan_x, fs.nchunk, fs.cpc);
        // (a fixed number of iterations, no host check: subints that are done cost their kernels nothing)
        for (int it = 0; it < std::min(12, c->max_iter + 1); ++it) {
            { Prof pr(c, KF_EVAL);
              hipLaunchKernelGGL((k_eval_scat<8, false>), dim3(fs.nact, fs.nchunk), dim3(256), 0, c->stream, fs); }
            { Prof pr(c, KF_STEP);
              hipLaunchKernelGGL(k_step, dim3(fs.nact), dim3(64), 0, c->stream, fs); }
        }
        hipLaunchKernelGGL(k_adopt_coarse, dim3((ns + 63) / 64), dim3(64), 0, c->stream, fa);
        HIP_TRY(hipGetLastError());
    }
    // ---- trust-region iterations: evaluation + step, until every subint is done
    const int max_evals = all_done ? 0 : std::max(1, c->max_iter + 1);
    // (the first look at the count of unfinished subints: SciPy's trust-ncg needs ~15 evaluations for a scattering
    // fit and the model takes over after ~6 passes -- no subint is done before iteration 5, and every look before
    // that drains the queue for nothing: +0.8 ... 1.2 % on configs[3], profiles/r05_small_ab.txt)
    const int check_from = smodel ? std::max(c->check_from, 5) : c->check_from;
    int pending = -1;           // slot of the lagged check in flight
    for (int it = 0; it < max_evals; ++it) {
        if (it == 0 && fuse) {
            Prof pr(c, KF_ACCUM);
            hipLaunchKernelGGL(k_accum, dim3(fa.nact, fa.nchunk), dim3(256), 0, c->stream, fa);
        } else {
            Prof pr(c, KF_EVAL);
            const dim3 eg(fa.nact, fa.nchunk);
            if (scat && fa.x_f32) hipLaunchKernelGGL((k_eval_scat<8, true>), eg, dim3(256), 0, c->stream, fa);
            else if (scat) hipLaunchKernelGGL((k_eval_scat<8, false>), eg, dim3(256), 0, c->stream, fa);
            else hipLaunchKernelGGL(k_eval_fast, dim3(fa.nact, fa.nchunk), dim3(256), 0, c->stream, fa);
        }
        if (smodel && it >= 2) {
            // subints whose last proposal asked for it: this evaluation is the model
            // pass, and the rest of their iterations run on the model
            Prof pr(c, KF_SCATMODEL);
            hipLaunchKernelGGL(k_scat_model, dim3(fa.nact, fa.nchunk), dim3(256), 0, c->stream, fa);
            hipLaunchKernelGGL(k_scat_model_solve, dim3(fa.nact), dim3(256), 0, c->stream, fa);
        }
        {
            Prof pr(c, KF_STEP);
            hipLaunchKernelGGL(k_step, dim3(fa.nact), dim3(64), 0, c->stream, fa);
        }
        HIP_TRY(hipGetLastError());
        if (c->lagged_check && !smodel) {
            // (measured on configs[3]: +5 % with the Newton solver; with the model pass in the loop
            // -0.8 %, so that flow keeps the synchronous check.)  The host looks at the count of unfinished subints ONE iteration behind: the count as
            // it stood after iteration it - 1 arrives while the GPU works on iteration it, so the
            // queue never runs dry while the host waits and launches (a synchronous check costs a
            // drain and a relaunch, ~50-80 us, per iteration).  The price: when everything has
            // finished, one more iteration has been queued -- kernels that find every subint done.
            if (pending >= 0) {
                HIP_TRY(hipEventSynchronize(c->evq[pending]));
                if (c->nactive_h[2 + pending] <= 0) break;
                pending = -1;
            }
            if (it >= check_from && ((it - check_from) % c->check_every) == 0) {
                const int slot = it & 1;
                if ((rc = publish_int(c, c->nactive_h + 2 + slot, fa.nactive))) return fail(rc, "count copy failed");
                HIP_TRY(hipEventRecord(c->evq[slot], c->stream));
                pending = slot;
            }
        } else if (it >= check_from && ((it - check_from) % c->check_every) == 0) {
            // (a copy command here: followed at once by a wait, it measured faster than the publishing kernel)
            HIP_TRY(hipMemcpyAsync(c->nactive_h, fa.nactive, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->nactive_h[0] <= 0) break;
        }
    }
    if ((rc = finalize_and_fetch())) return rc;
    unpack_outputs();
    return PP_OK;
}

// what a batch needs before its first sub-batch: validation, the models it uses, whether the scattering
// path is needed, default reference frequencies, the sub-batch size the work-memory budget allows
struct BatchPlan {
    int Kt = 0, cap = 0, flow_key = 0;
    bool scat = false;
    double per_sub = 0.0;
    std::vector<double> nufit, nuout;
};

static int plan_batch(pp_ctx* c, const pp_fit_in* in, pp_fit_out* out, BatchPlan* bp) {
    if (in->nsub < 1 || in->nchan < 1) return fail(PP_EINVAL, "bad batch shape %d x %d", in->nsub, in->nchan);
    if (!nbin_any_ok(in->nbin))
        return fail(PP_EINVAL, "nbin %d must be a power of two in [32, 8192] or an even number in [8, 4096]", in->nbin);
    if (!in->data || !in->freqs || !in->P || !in->init_params) return fail(PP_EINVAL, "missing input array");
    if (in->data_dtype != PP_F64 && in->data_dtype != PP_F32) return fail(PP_EINVAL, "data_dtype %d", in->data_dtype);
    if (in->freqs_stride != 0 && in->freqs_stride != in->nchan) return fail(PP_EINVAL, "freqs_stride must be 0 or nchan");
    if (in->method != PP_METHOD_TRUST_NCG && in->method != PP_METHOD_NEWTON) return fail(PP_EINVAL, "method %d", in->method);
    if (!out->params || !out->param_errs || !out->nu_refs || !out->cov || !out->chi2 || !out->red_chi2 ||
        !out->snr || !out->nfeval || !out->return_code)
        return fail(PP_EINVAL, "missing output array");
    HIP_TRY(hipSetDevice(c->device));
    const int N = in->nsub, C = in->nchan, B = in->nbin;
    // models used by this batch
    int Kt = 0;
    for (int i = 0; i < N; ++i) {
        const int sl = in->model_slot ? in->model_slot[i] : 0;
        if (sl < 0 || sl >= PP_MAX_SLOTS || !c->slots[sl].set) return fail(PP_ESTATE, "subint %d: model slot %d not set", i, sl);
        if (c->slots[sl].nchan != C || c->slots[sl].nbin != B)
            return fail(PP_EINVAL, "subint %d: model slot %d is %dx%d, data is %dx%d", i, sl, c->slots[sl].nchan, c->slots[sl].nbin, C, B);
        Kt = std::max(Kt, c->slots[sl].Kt);
        if (!in->model_slot) break;
    }
    // scattering path needed? (reference builds B_nk for any tau != 0)
    bool scat = in->fit_flags[3] || in->fit_flags[4] || in->log10_tau;
    if (!scat)
        for (int i = 0; i < N; ++i) if (in->init_params[(size_t)i * 5 + 3] != 0.0) { scat = true; break; }
    if (in->ref_seed) {
        const pp_seed_ref* rs = in->ref_seed;
        const int cstep = std::max(1, c->seed_chan_stride);
        // (no scattering: Taylor model about the pilot seed's phase; scattering: cross-spectrum stored)
        const bool path = scat ? true : (c->use_taylor && c->moments_in_xspec && cstep > 1 && C / cstep >= 16);
        bool ok = path && c->max_iter > 0 && c->one_exchange &&
                  B == 2048 && 2 * Kt < B / 2 && C % PP_ROW_CHUNK == 0 &&
                  in->errs && in->seed_ns == 0 && rs->model_profs && rs->nu_mean && rs->Ns >= 1 &&
                  (rs->model_prof_stride == 0 || rs->model_prof_stride == B);
        for (int i = 0; ok && i < N; ++i) ok = (in->init_params[(size_t)i * 5 + 2] == 0.0);
        if (!ok)
            return fail(PP_ENOTSUP, "ref_seed: no single-pass path for this batch (needs 2048-bin portraits, "
                                    "a template that keeps < 512 harmonics, nchan a multiple of %d (and >= %d without "
                                    "scattering), errs given, GM guesses 0)", PP_ROW_CHUNK, 16 * cstep);
    }
    // default reference frequencies: mean of the (unmasked) channel frequencies
    bp->nufit.assign((size_t)N * 3, 0.0);
    bp->nuout.assign((size_t)N * 3, 0.0);
    // the masked mean needs the mask on the host: a device-resident mask is
    // copied back once, and only if some reference frequency was left to default
    std::vector<uint8_t> mask_h;
    const uint8_t* mask_host = in->chan_mask;
    if (in->chan_mask && in->aux_on_device) {
        bool need = (in->nu_fits == nullptr);
        for (size_t j = 0; !need && j < (size_t)N * 3; ++j) need = std::isnan(in->nu_fits[j]);
        mask_host = nullptr;
        if (need) {
            mask_h.resize((size_t)N * C);
            HIP_TRY(hipMemcpy(mask_h.data(), in->chan_mask, mask_h.size(), hipMemcpyDeviceToHost));
            mask_host = mask_h.data();
        }
    }
    for (int i = 0; i < N; ++i) {
        double mean = NAN;
        for (int j = 0; j < 3; ++j) {
            double v = in->nu_fits ? in->nu_fits[(size_t)i * 3 + j] : NAN;
            if (std::isnan(v)) {
                if (std::isnan(mean)) {
                    const double* f = in->freqs + (in->freqs_stride ? (size_t)i * C : 0);
                    const uint8_t* m = mask_host ? mask_host + (size_t)i * C : nullptr;
                    double s = 0.0; long long cnt = 0;
                    for (int n = 0; n < C; ++n) if (!m || m[n]) { s += f[n]; ++cnt; }
                    mean = cnt ? s / (double)cnt : NAN;
                }
                v = mean;
            }
            bp->nufit[(size_t)i * 3 + j] = v;
            bp->nuout[(size_t)i * 3 + j] = in->nu_outs ? in->nu_outs[(size_t)i * 3 + j] : NAN;
        }
    }
    // sub-batches sized to the work-memory budget
    size_t free_b = 0, total_b = 0;
    const double per_sub = (double)C * (Kt + std::max(0, c->x_pad)) * 16.0 + (in->data_on_device ? 0.0 : (double)C * B * (in->data_dtype == PP_F64 ? 8 : 4)) +
                           (double)C * (8.0 * 12 + 2 * 9 * 8.0) + 4096.0 +
                           ((scat && c->scat_model) ? (double)C * PP_MROW * 8.0 : 0.0);
    // (a batch no larger than one that already ran under this budget needs no look at the free
    // memory: hipMemGetInfo costs as much as the post-fit stage of a 512 x 1024 batch)
    int cap = N;
    const int flow_key = (scat ? 1 : 0) | (in->data_on_device ? 2 : 0) | (in->data_dtype == PP_F64 ? 4 : 0) |
                         (in->ref_seed ? 8 : 0) | (in->seed_ns > 0 ? 16 : 0) | (in->method == PP_METHOD_NEWTON ? 32 : 0);
    bp->flow_key = flow_key;
    const auto known = c->known_ok.find(flow_key);
    if (per_sub * N > std::min(c->max_work_bytes, known == c->known_ok.end() ? 0.0 : known->second)) {
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const double budget = std::min(c->max_work_bytes, 0.85 * ((double)free_b + (double)c->X.cap + (double)c->data.cap + (double)c->csum.cap));
        cap = (int)std::max(1.0, std::floor(budget / per_sub));
    }
    cap = std::min(std::min(cap, N), 65535);   // (subints index the grid's y dimension)
    cap = std::max(1, std::min(cap, (int)(2147483647LL / C)));   // (the transforms count rows in 32 bits)
    bp->Kt = Kt; bp->scat = scat; bp->cap = cap; bp->per_sub = per_sub;
    return PP_OK;
}

// every sub-batch in turn, synchronously; device time of the whole into out->duration
static int run_batch_sync(pp_ctx* c, const pp_fit_in* in, pp_fit_out* out, const BatchPlan& bp) {
    const int N = in->nsub;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (int s0 = 0; s0 < N; s0 += bp.cap) {
        const int ns = std::min(bp.cap, N - s0);
        int rc = fit_chunk(c, in, out, s0, ns, bp.Kt, bp.scat, bp.nufit, bp.nuout);
        if (rc) return rc;
    }
    c->known_ok[bp.flow_key] = std::max(c->known_ok[bp.flow_key], bp.per_sub * std::min(bp.cap, N));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    if (out->duration) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        out->duration[0] = 1e-3 * ms;
    }
    if (c->profile) resolve_spans(c);
    return PP_OK;
}

extern "C" int pp_fit_portrait_batch(pp_ctx* c, const pp_fit_in* in, pp_fit_out* out) {
    if (!c || !in || !out) return fail(PP_EINVAL, "pp_fit_portrait_batch: null argument");
    // a submitted batch owns the context (work buffers, stream, counters) until pp_fit_wait
    if (c->job_active && t_worker_of != c)
        return fail(PP_ESTATE, "pp_fit_portrait_batch: a submitted fit is pending on this context (pp_fit_wait first)");
    if (!c->pending.empty())
        return fail(PP_ESTATE, "pp_fit_portrait_batch: %zu enqueued batch(es) not collected yet (pp_fit_collect first)", c->pending.size());
    BatchPlan bp;
    int rc = plan_batch(c, in, out, &bp);
    if (rc) return rc;
    return run_batch_sync(c, in, out, bp);
}

// --------------------------------------------------------------------------
// stream-ordered batches: pp_fit_enqueue queues a whole batch -- inputs, kernels, outputs on their
// way to a pinned staging block -- and returns without waiting; pp_fit_collect completes the oldest.
// PP_NSTAGE may be pending: while the GPU works on one batch the host marshals and queues the next, so
// the stream never runs dry between batches.  One stream; a staging block and a set of work buffers per
// pending batch, because a batch's solve and post-fit stage may run INSIDE the next batch's transform
// (fuse_tail: pp_ctx::ptail, tail_work) and its outputs leave behind that transform.
// --------------------------------------------------------------------------
extern "C" int pp_fit_enqueue(pp_ctx* c, const pp_fit_in* in, pp_fit_out* out) {
    if (!c || !in || !out) return fail(PP_EINVAL, "pp_fit_enqueue: null argument");
    if (c->job_active) return fail(PP_ESTATE, "pp_fit_enqueue: a submitted fit is pending on this context (pp_fit_wait first)");
    if (c->pending.size() >= PP_NSTAGE) return fail(PP_ESTATE, "pp_fit_enqueue: %d batches are pending (pp_fit_collect first)", PP_NSTAGE);
    BatchPlan bp;
    int rc = plan_batch(c, in, out, &bp);
    if (rc) return rc;
    pp_ctx::Deferred d;
    d.in = *in; d.out = *out; d.queued = false; d.rc = PP_OK;
    d.stage = c->pending.empty() ? c->cur_stage : (c->pending.back().stage + 1) % PP_NSTAGE;
    c->cur_stage = d.stage;
    pp_ctx::Stage& sg = c->stage[d.stage];
    if (bp.cap >= in->nsub) {
        HIP_TRY(hipEventRecord(sg.t0, c->stream));
        bool deferred = false;
        rc = fit_chunk(c, in, out, 0, in->nsub, bp.Kt, bp.scat, bp.nufit, bp.nuout, &deferred);
        if (rc) return rc;
        c->known_ok[bp.flow_key] = std::max(c->known_ok[bp.flow_key], bp.per_sub * in->nsub);
        // (a deferred batch ends on the stream of its post-fit stage, which waited for its transform; one whose tail
        // is still unqueued -- option fuse_tail -- gets its event when the tail has been queued: by the next batch's
        // transform, or by flush_tail)
        if (!(deferred && c->ptail.valid && c->ptail.stage == d.stage))
            HIP_TRY(hipEventRecord(sg.done, (deferred && c->last_post) ? c->last_post : c->stream));
        d.queued = deferred;
        d.span_end = c->spans.size();
        if (!deferred) {
            // (a flow with host decisions in its middle: it has run to its end)
            HIP_TRY(hipEventSynchronize(sg.done));
            if (out->duration) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, sg.t0, sg.done));
                out->duration[0] = 1e-3 * ms;
            }
            if (c->profile) resolve_spans(c);
            d.span_end = 0;
        }
    } else {
        // (more than the work-memory budget holds at once: sub-batches, synchronously -- only with
        // nothing else pending, the sub-batches reuse the staging blocks)
        if (!c->pending.empty()) return fail(PP_ESTATE, "pp_fit_enqueue: a batch that needs sub-batches cannot follow a pending one");
        if ((rc = flush_tail(c))) return rc;
        if ((rc = run_batch_sync(c, in, out, bp))) return rc;
    }
    c->pending.push_back(d);
    return PP_OK;
}

extern "C" int pp_fit_pending(pp_ctx* c) {
    if (!c) return fail(PP_EINVAL, "null context");
    return (int)c->pending.size();
}

extern "C" int pp_fit_collect(pp_ctx* c) {
    if (!c) return fail(PP_EINVAL, "null context");
    if (c->pending.empty()) return fail(PP_ESTATE, "pp_fit_collect: nothing enqueued");
    pp_ctx::Deferred d = c->pending.front();
    c->pending.pop_front();
    if (!d.queued) { if (d.rc) g_err = d.err; return d.rc; }
    HIP_TRY(hipSetDevice(c->device));
    pp_ctx::Stage& sg = c->stage[d.stage];
    // (its tail was waiting for a next batch that did not come: the stand-alone kernels)
    if (c->ptail.valid && c->ptail.stage == d.stage)
        if (int rcf = flush_tail(c)) return rcf;
    HIP_TRY(hipEventSynchronize(sg.done));
    const int ns = d.in.nsub;
    if (unfinished_in_stage(sg.o_host, ns) <= 0) {
        unpack_stage(sg.o_host, &d.out, 0, ns);
        unpack_seed_phases(sg.o_host, &d.in, 0, ns);
        if (d.out.duration) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, sg.t0, sg.done));
            d.out.duration[0] = 1e-3 * ms;
        }
        if (c->profile) resolve_spans(c, d.span_end);
        return PP_OK;
    }
    // some subints left the one-pass flow (poor guesses): the batch is fitted again, synchronously, by
    // the general flow -- behind whatever has been queued since; its inputs are still the caller's
    c->cur_stage = d.stage;
    BatchPlan bp;
    int rc = plan_batch(c, &d.in, &d.out, &bp);
    if (rc) return rc;
    // (run_batch_sync refuses nothing here: the check for pending batches is pp_fit_portrait_batch's)
    return run_batch_sync(c, &d.in, &d.out, bp);
}

// --------------------------------------------------------------------------
// asynchronous form: the batch runs on a worker thread of the context (host-to-device
// copies, kernels and the few host checks of the iteration included), the caller's
// thread is free meanwhile -- to read the next archive, or to drive another context
// whose copies and kernels then overlap with this one's
// --------------------------------------------------------------------------
extern "C" int pp_fit_submit(pp_ctx* c, const pp_fit_in* in, pp_fit_out* out) {
    if (!c || !in || !out) return fail(PP_EINVAL, "pp_fit_submit: null argument");
    if (c->job_active) return fail(PP_ESTATE, "pp_fit_submit: a submitted fit is pending (pp_fit_wait first)");
    c->job_in = *in;
    c->job_out = *out;
    c->job_done.store(0);
    c->job_active = true;
    c->job = std::thread([c]() {
        t_worker_of = c;
        c->job_rc = pp_fit_portrait_batch(c, &c->job_in, &c->job_out);
        c->job_err = g_err;              // (the worker's thread-local message)
        c->job_done.store(1);
    });
    return PP_OK;
}

extern "C" int pp_fit_poll(pp_ctx* c) {
    if (!c) return fail(PP_EINVAL, "null context");
    if (!c->job_active) return fail(PP_ESTATE, "pp_fit_poll: nothing submitted");
    return c->job_done.load() ? 1 : 0;
}

extern "C" int pp_fit_wait(pp_ctx* c) {
    if (!c) return fail(PP_EINVAL, "null context");
    if (!c->job_active) return fail(PP_ESTATE, "pp_fit_wait: nothing submitted");
    c->job.join();
    c->job_active = false;
    if (c->job_rc) g_err = c->job_err;
    return c->job_rc;
}

#include "pp_extra_api.h"
