// The tail of an enqueued batch -- what follows its transform: the solve on the Taylor model and the post-fit stage,
// and for the reference-seed flow (get_TOAs' default) the reference's phase guess in front of them -- as TICKETS that
// the waves of the NEXT batch's transform work off between their own rows (tail_work; declared at the top of
// pp_kernels.h, called from k_xspec_q1024 / k_xspec_qf<1024> / k_xspec_qr1024).
#pragma once
#include "pp_kernels.h"
#include "pp_extra.h"
#include "pp_xspec1024r.h"

namespace pp {

// --------------------------------------------------------------------------
// The tail of a batch inside the NEXT batch's transform (round 5).  The solve on the Taylor model and the post-fit
// stage of batch k need every row of batch k, so they cannot start before its transform ends -- and behind it they
// are 0.7 ms of a 14.7 ms step during which the f64 pipes idle (the solve re-reads the Taylor rows at the HBM
// roofline) --; beside the persistent transform of batch k + 1 no other kernel finds a wave slot
// (profiles/r05_overlap_ab.txt).  So the transform's own waves do the work: a subint of batch k is a TICKET, every
// wave of batch k + 1's transform draws one before its first row and the waves that run out of rows draw the rest.
// A wave is out of the transform for the ~1 ms its ticket takes (of ~14), which costs the transform far less than its
// share of the waves (half the waves keep 81 % of the rate, profiles/r05_grid_scale.txt).  One wave does what four
// (solve) and eight (post-fit) do in the stand-alone kernels by walking their waves in turn (NVW = NT / 64): the same
// bits, so an enqueued batch returns what a synchronous call returns.
// LDS: the transform's own image, free before the first row and after the last.
// --------------------------------------------------------------------------
// The reference-seed flow's part of a tail (get_TOAs' default, pptoas.py:421-457): what turns the chunk partials its
// pass left (k_xspec_qr1024) into the reference's phase guess and the start point of SciPy's walk -- the work of
// k_refseed_finish, k_fps and k_refseed_start, per subint.
struct RefTailArgs {
    int on;                   // 0: a plain tail (solve + post-fit stage only)
    int ncc;                  // channel blocks per subint
    const cplx* part;         // [nsub][ncc][RS_NACC][64] the pass's partial channel sums
    const double* delta;      // [nsub] Taylor phase minus the reference's rotation phase (k_refseed_prep)
    const double* wsum;       // [nsub] summed weights of the channel mean
    const unsigned* mws;      // rows-in-use words of the pass, or nullptr
    const cplx* mspec;        // the template profile's spectrum: [M + 1] for all subints, or one per subint
    int mstride;              // M + 1, or 0
    FpsArgs fps;              // lo, hi, Ns, finish, out7 (spec / specm unused: the data spectrum comes from `part`)
    double* xs;               // [nsub][5] start points: the phase becomes wrap(fit_phase_shift's phase + K_i)
    double* seed_phase;       // [nsub] leaves with the outputs
};
struct TailArgs {
    FitArgs fa;               // of the batch whose tail this is (solve_cache <= PP_TAIL_CACHE, tail_fused = 1)
    unsigned ticket, done;    // next subint to hand out; subints finished
    int nsub;
    int solve_nt, solve_pf;   // the widths the stand-alone kernels would be launched with
    int fin_nt;               // 64 (<8, 64>), 128 / 256 / 512 (<8, NT>), 0 (<0, 256>)
    RefTailArgs rs;
};

// the reference's phase guess of subint i by ONE wave: k_refseed_finish's harmonics feed k_fps's fit directly (no
// spectrum in global memory), the fit's 256 threads are walked as four virtual waves (fps_body<4>: bitwise k_fps),
// k_refseed_start's arithmetic closes it.  LDS: [0, 16) block sums, [128, 128 + 2 * 1024) the cross-spectrum.
__device__ __forceinline__ void refseed_ticket(const RefTailArgs& r, const int nsub, const int i, const int tid, double* lds) {
    constexpr int M = 1024;
    static_assert(128 + 2 * M <= PP_TAIL_LDS_DOUBLES, "the fit's cross-spectrum in the carrying kernel's image");
    cplx* X = reinterpret_cast<cplx*>(lds + 128);
#ifdef PP_TICKET_TIMING
    const long long tr0 = wall_clock64();
#endif
    // ---- rot_prof's spectrum from the chunk partials (k_refseed_finish's arithmetic, harmonic by harmonic: the same
    // sums in the same order), laid out for ONE wave: lane l takes partial-lane l of all twelve slots, so that every
    // chunk is twelve coalesced 1 KB loads with nothing depending on them but twelve running sums -- the harmonic-by-
    // harmonic walk of refseed_spec_value is 128 dependent round trips to HBM per harmonic, a millisecond per ticket
    // beside a transform that keeps the memory system busy.  d_k goes to X[k - 1], where the fit reads it (and
    // overwrites it with the cross-spectrum, same index).
    {
        const double delta_i = r.delta[i], wsum_i = r.wsum[i];
        const double inv = wsum_i > 0.0 ? 1.0 / wsum_i : 0.0;
        cplx s[RS_NACC];
#pragma unroll
        for (int q = 0; q < RS_NACC; ++q) s[q] = make_double2(0.0, 0.0);
        const cplx* base = r.part + (size_t)i * r.ncc * RS_NACC * 64 + tid;
#pragma unroll 2
        for (int cc = 0; cc < r.ncc; ++cc) {
            if (r.mws && r.mws[(size_t)cc * nsub + i] == 0u) continue;
            cplx v[RS_NACC];
#pragma unroll
            for (int q = 0; q < RS_NACC; ++q)
                v[q] = (q < RS_NREG || tid == 0) ? base[((size_t)cc * RS_NACC + q) * 64] : make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < RS_NACC; ++q) { s[q].x += v[q].x; s[q].y += v[q].y; }
        }
        const int lam = fftq_lambda(tid);          // (fftq_lane_of(lam) == tid: this lane's partials are harmonics lam + 64 kd)
        auto put = [&](const int k, cplx t) __attribute__((always_inline)) {
            const cplx rot = unit_phasor((double)k, -delta_i);
            t = cmul(t, rot);
            t.x *= inv; t.y *= inv;
            if (k == M) t.y = 0.0;
            X[k - 1] = t;
        };
#pragma unroll
        for (int q = 0; q < 7; ++q) put(lam != 0 ? lam + 64 * q : 64 * (q + 1), s[q]);
#pragma unroll
        for (int q = 7; q < RS_NREG; ++q) put(lam + 64 * (12 + q - 7), s[q]);
        if (lam == 0) put(M, s[RS_NREG]);
        // (the harmonics the pass does not accumulate: 448 < k < 768)
#pragma unroll
        for (int kd = 7; kd < 12; ++kd) {
            const int k = lam + 64 * kd;
            if (k > 448) X[k - 1] = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();
#ifdef PP_TICKET_TIMING
    if (tid == 0 && (i % 128) == 5) printf("ticket %4d: partial sums + spectrum %.1f us\n", i, 0.01 * (double)(wall_clock64() - tr0));
#endif
    const cplx* m = r.mspec + (size_t)i * r.mstride;
    auto dv = [&](const int k) __attribute__((always_inline)) { return X[k - 1]; };
    fps_body<4>(r.fps, i, tid, M, dv, m, X, lds, nullptr, nullptr);
    __syncthreads();             // (out7[i] written by lane 0, read by lane 0: ordered anyway; the LDS is free again)
    if (tid == 0) refseed_start_one(r.fps.out7, i, r.xs, r.seed_phase);
    __syncthreads();
}

__device__ __noinline__ void tail_work(const TailArgs* t, double* lds, int nlds, int tid, int max_tickets) {
    TailArgs* tw = const_cast<TailArgs*>(t);
    const int nsub = t->nsub;
    for (int round = 0; round < max_tickets; ++round) {
        unsigned tk = 0;
        if (tid == 0) tk = atomicAdd(&tw->ticket, 1u);
        tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
        if (tk >= (unsigned)nsub) return;
        // (read where it lies, in device memory: a private copy of the ~600-byte block lives in scratch memory and
        // every use of a field becomes a scratch load; the host has set solve_cache for PP_TAIL_LDS_DOUBLES and tail_fused)
        const FitArgs& a = t->fa;
        const int i = (int)tk;
        // [0, 528): block sums (PP_BSUM_DOUBLES(8, 31) = 512 the larger);  [528, 536): one broadcast value;  the rest:
        // the solve's cache of channel invariants (4 doubles a channel; results do not depend on its size)
        double* scratch = lds;
        double* sh = lds + 528;
        double* inv = lds + 536;
        (void)nlds;
#ifdef PP_TICKET_TIMING
        const long long tk0 = wall_clock64();
#endif
        if (t->rs.on) refseed_ticket(t->rs, nsub, i, tid, lds);
#ifdef PP_TICKET_TIMING
        const long long tk1 = wall_clock64();
#endif
        const int snt = t->solve_nt, spf = t->solve_pf, fnt = t->fin_nt;
        if (snt == 64) taylor_solve_body<64, PP_SOLVE_PF, 1>(a, i, tid, scratch, inv);
        else if (snt == 128) taylor_solve_body<128, PP_SOLVE_PF, 2>(a, i, tid, scratch, inv);
        else if (snt == 512) taylor_solve_body<512, PP_SOLVE_PF, 8>(a, i, tid, scratch, inv);
        else if (spf == 0) taylor_solve_body<256, 0, 4>(a, i, tid, scratch, inv);
        else taylor_solve_body<256, PP_SOLVE_PF, 4>(a, i, tid, scratch, inv);
        __syncthreads();             // (one wave: orders its LDS and global writes before the post-fit stage reads them)
#ifdef PP_TICKET_TIMING
        const long long tk2 = wall_clock64();
#endif
        if (fnt == 64) finalize_body<8, 64, 1>(a, i, tid, scratch, sh);
        else if (fnt == 128) finalize_body<0, 128, 2>(a, i, tid, scratch, sh);
        else if (fnt == 512) finalize_body<0, 512, 8>(a, i, tid, scratch, sh);
        else finalize_body<0, 256, 4>(a, i, tid, scratch, sh);
        __syncthreads();
#ifdef PP_TICKET_TIMING
        if (tid == 0 && (i % 128) == 5)
            printf("ticket %4d (block %5d): guess %.1f us  solve %.1f us  post-fit %.1f us\n", i, (int)blockIdx.x, 0.01 * (double)(tk1 - tk0),
                   0.01 * (double)(tk2 - tk1), 0.01 * (double)(wall_clock64() - tk2));
#endif
        // the last ticket to finish publishes the count of unfinished subints (the stand-alone post-fit kernel
        // runs after the whole solve kernel: its subint 0 does it)
        if (tid == 0) {
            __threadfence();
            const unsigned d = atomicAdd(&tw->done, 1u);
            if (d + 1u == (unsigned)nsub) {
                __threadfence();
                a.o_npass[a.nsub] = atomicAdd(a.nactive, 0);
            }
        }
    }
}

}  // namespace pp
