// HIP kernels of the wideband-TOA fit (gfx950).  Included by pp_toas.hip.
//
//   k_model_fft   rFFT of template rows -> m_nk (k = 1..M), sum_k |m_nk|^2
//   k_model_kcut  last harmonic whose |m_nk| exceeds eps * max_k |m_nk|
//   k_phase0      phi_n of every (subint, channel) at the initial parameters
//   k_xspec       rFFT of data rows, X_nk = d_nk conj(m_nk), sum_k |d_nk|^2,
//                 power-spectrum noise estimate        (pptoaslib.py:976-985);
//                 modes 2/3: the per-channel Taylor model of C_n instead of X
//   k_eval_moments / k_taylor_solve   the same Taylor model from a stored X, and
//                 the certified Newton solve on it     (replaces :1001-1014 when
//                 there is no scattering)
//   k_prep        per-channel weights 1/sigma_F^2, S_d   (pptoaslib.py:980-985)
//   k_eval        chi^2 surface evaluator: per-channel Fourier sums by phasor
//                 recurrence, reduced to f, grad f, Hess f (pptoaslib.py:525-643)
//   k_step        trust-region Newton step per subint    (pptoaslib.py:1001-1014)
//   k_finalize    zero-covariance frequencies, output transform, covariance
//                 with amplitudes, S/N, chi^2            (pptoaslib.py:1040-1096)
#pragma once
#include "pp_fft.h"

namespace pp {

// --------------------------------------------------------------------------
// argument blocks (passed by value)
// --------------------------------------------------------------------------
struct ModelFftArgs {
    const void* port;   // [nchan][B]
    cplx* mft;          // [nchan][M]   harmonics 1..M
    double* msum;       // [nchan] sum_k |m|^2
    double* mmax;       // [nchan] max_k |m|^2
    double* mdc;        // [nchan] DC harmonic (kept for the synthetic generator only)
    double* msq;        // [nchan][M]   |m_nk|^2 (what the scattering sums S_n(tau) read)
    const cplx* twB;
    int nchan;
};

// The solve and post-fit stage of the PREVIOUS batch, worked off by the waves of this batch's transform (round 5;
// defined at the end of this file, behind the bodies it runs): `tail` of XspecArgs, or nullptr.
struct TailArgs;
#define PP_TAIL_LDS_DOUBLES 2176     // LDS of the carrying kernels (k_xspec_q1024, k_xspec_qf<1024>: 1088 complex), in doubles
#define PP_TAIL_CACHE ((PP_TAIL_LDS_DOUBLES - 536) / 4)     // channels whose invariants the in-kernel solve keeps there
__device__ void tail_work(const TailArgs* t, double* lds, int nlds, int tid, int max_tickets);

struct XspecArgs {
    const void* data;         // [nsub][nchan][B]
    const cplx* const* mft;   // [nslot] device table of model FT base pointers
    const cplx* mft0;         // slot 0's base (no pointer chase when slot == nullptr)
    const int* const* ktab;   // [nslot] per-channel kept-harmonic counts, or nullptr (= Kt)
    const int* kt0;           // slot 0's
    const int* slot;          // [nsub] or nullptr
    cplx* X;                  // [nsub][nchan][Xs], harmonics 1..Kt in the first Kt places of a row
    double* sdraw;            // [nsub][nchan] sum_{k>=1} |d|^2
    double* noise;            // [nsub][nchan] get_noise_PS estimate
    const cplx* twB;
    int nsub, nchan, Kt;
    int Xs;                   // pitch of X's rows (elements): Kt + the optional pad
    // fused first evaluation (phase/DM/GM model at the initial parameters)
    const double* x0;         // [nsub][5]
    const double* P;          // [nsub]
    const double* nu_fit;     // [nsub][3]
    const double* freqs; long long freqs_stride;
    double* csum0;            // [nsub][nchan][3]: A0 A1 A2 at x0 (csum buffer 0)
    double* tay;              // [nsub][nchan][PP_TSTRIDE]: Taylor model at x0 (MODE 2)
    const double* ph0;        // [nsub][nchan]: phi_n at x0 (k_phase0), MODE 1 and 2
    // Subsets (pilot pass over every cstep-th channel; re-transform of the subints
    // that need evaluations): `nsub` x `nchan` is the set of rows PROCESSED, row
    // (nn, j) being channel coff + nn*cstep of subint act[j] (act == nullptr: j).
    // X is indexed by the compact (j, nn); every other array by the true
    // (subint, channel) with row pitch nchan_full.
    const int* act;
    int cstep, coff, nchan_full;
    // dynamic dealing of row chunks (RowWalk): a device counter that only grows, and
    // its value when this launch started
    unsigned* ticket;
    unsigned ticket_base;
    int x_f32;                // the cross-spectrum is stored as pairs of floats (8 B per harmonic)
    // rows in use (RowWalk): one word per chunk of PP_ROW_CHUNK rows in the kernel's own row order,
    // or nullptr = every row.  Only with act == nullptr, cstep == 1 (the main pass over a batch).
    const unsigned* mwords;
    // one-pass flow, enqueued batches: the previous batch's solve + post-fit stage as tickets (one subint each) the
    // waves of this launch draw -- one before their first row, the rest after their last -- or nullptr
    const TailArgs* tail;
    int tail_nsub;            // ... how many tickets it holds
};

// one harmonic of the stored cross-spectrum (row pitch Xs elements of 16 or 8 bytes)
__device__ __forceinline__ void store_x(const XspecArgs& a, size_t row, int k, const cplx& x) {
    if (a.x_f32) reinterpret_cast<float2*>(a.X)[row * a.Xs + (k - 1)] = make_float2((float)x.x, (float)x.y);
    else a.X[row * a.Xs + (k - 1)] = x;
}

struct FitArgs {
    int nsub, nchan, nbin, M, Kt;
    int Xs;                   // pitch of X's rows (elements)
    int flags[5];
    int log10_tau, option, is_toa, max_iter, scat;
    int method;               // PP_METHOD_*: 0 = SciPy trust-ncg, step for step; 1 = Newton to rounding
    const cplx* X;
    const cplx* const* mft;
    const double* const* msq;   // [nslot] |m_nk|^2 tables
    const double* const* msum;
    const int* const* ktab;   // per-slot per-channel kept harmonics, nullptr = Kt everywhere
    const int* slot;
    const double* freqs; long long freqs_stride;
    const double* wts;        // [nsub][nchan] 1/(sigma^2 B/2), 0 = masked
    const double* sdraw;
    const double* P;          // [nsub]
    const double* nu_fit;     // [nsub][3]
    const double* nu_out;     // [nsub][3] NaN = zero-covariance
    const double* x0;         // [nsub][5]
    SubState* st;
    double* csum;             // [2][nsub][nchan][ncs]
    int ncs;                  // 3 (no scattering) or 9
    double* tay;              // [nsub][nchan][PP_TSTRIDE] Taylor model about x0 (no scattering)
    double* partial;          // [nsub][nchunk][PP_NACC]
    int nchunk, cpc;          // channels per chunk
    int* nactive;
    // outputs (device)
    double* o_params; double* o_errs; double* o_nu; double* o_cov;
    double* o_chi2; double* o_rchi2; double* o_snr; int* o_nfev; int* o_rc; int* o_npass;
    double* o_scales; double* o_scale_errs; double* o_csnr;
    double* o_f0; double* o_g0; double* o_H0;
    double* o_rec;            // [nsub][PP_RECORD_WIDTH] TOA records left on the device, or nullptr
    // subsets for the kernels that read X (see XspecArgs): blockIdx.y = j walks
    // act[0..nact), channels coff + nn*cstep for nn < nchan_x; X is compact in (j, nn)
    const int* act;
    int nact, nchan_x, cstep, coff;
    // scattering fits: per-channel model of the closing iterations (pp_scatmodel.h)
    double* mdl;              // [nsub][PP_MROW][nchan]
    int use_model;
    double model_tol;         // predicted relative truncation below which the model pass is asked for
    int model_bet;            // criterion (b): ask one evaluation earlier when the proposal is the full Newton step
    // one-pass flow: a subint whose certificate fails is expanded again about its tentative answer,
    // up to `recentre` times
    int recentre;
    double* x0w;              // [nsub][5] the expansion points k_phase0 reads (writable view of x0)
    int x_f32;                // the cross-spectrum is stored as pairs of floats (k_eval_scat<., true> only)
    // reference-seed flow (pp_xspec1024r.h): the Taylor model was taken about x0 (the pilot seed's
    // phase), the iteration starts from xstart (the reference's own guess, known only after the pass)
    const double* xstart;     // [nsub][5] or nullptr (= start at the expansion point)
    int nfev_shadow;          // one-pass flow: SciPy's one-point cache compared on the absolute iterate fl(x + p)
    int solve_cache;          // k_taylor_solve: channels whose weight / geometry / template power are kept in LDS (32 B each)
    int tail_fused;           // the post-fit stage runs as tickets inside the next transform (tail_work): the count of
                              // unfinished subints is published by the LAST ticket, not by subint 0's
    int x_full;               // the channel subset (coff, cstep, nchan_x) is evaluated over a cross-spectrum stored for
                              // ALL channels: X rows are addressed by the true channel (k_eval_scat)
};

__device__ __forceinline__ int sub_of(const int* act, int j) { return act ? act[j] : j; }

// a pointer every lane holds the same value of, moved to scalar registers -- whatever load produced it is waited for HERE
template <typename P>
__device__ __forceinline__ const P* uniform_ptr(const P* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return as_global(reinterpret_cast<const P*>(((unsigned long long)hi << 32) | (unsigned long long)lo));
}

// The template row and the template's cut for (subint ia, channel n -> true channel ne), looked up when the CHANNEL
// changes -- once per chunk of rows; per row only with per-subint templates (a.slot) -- and waited for in here, the
// results leaving in scalar registers.  Round 5: the CU's vector-memory path returns in order ACROSS its waves, so a
// 4-byte L1 hit issued at the top of a row and read at once (the cut; and the lookups were written as selects between
// a loaded and a scalar pointer, which the compiler can only wait for with vmcnt(0)) comes back behind whatever the
// other rows' prefetches have in flight: a memory latency per wait, two waits per row.  With the lookups inside this
// branch a row that does not take it waits for nothing but its own data (profiles/r05_quiet_row_top_ab.txt: +3.5 %).
// Returns true when it looked.
#ifndef PP_STICKY_LOOKUP
#define PP_STICKY_LOOKUP 1
#endif
__device__ __forceinline__ bool channel_lookup(const XspecArgs& a, int ia, int n, int ne, int M, int& n_held,
                                               const cplx*& mrow, int& ktn) {
    if (PP_STICKY_LOOKUP && !a.slot && n == n_held) return false;
    const cplx* m = as_global(a.slot ? a.mft[a.slot[ia]] : a.mft0) + (size_t)ne * M;
    const int k = a.ktab ? as_global(a.slot ? a.ktab[a.slot[ia]] : a.kt0)[ne] : a.Kt;
    mrow = uniform_ptr(m);
    ktn = __builtin_amdgcn_readfirstlane(k);
    n_held = n;
    return true;
}

// --------------------------------------------------------------------------
// model rFFT
// --------------------------------------------------------------------------
template <int M, typename Tin>
__global__ __launch_bounds__(FftPlan<M>::T) void k_model_fft(ModelFftArgs a) {
    constexpr int T = FftPlan<M>::T;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    __shared__ double red[2 * (T / 64) + 2];
    const int tid = threadIdx.x;
    for (int n = blockIdx.x; n < a.nchan; n += gridDim.x) {
        const Tin* grow = reinterpret_cast<const Tin*>(a.port) + (size_t)n * (2 * M);
        fft_row<M, Tin>(lds, grow, a.twB, tid);
        double s = 0.0, mx = 0.0;
        for (int k = 1 + tid; k <= M; k += T) {
            const cplx d = rfft_harmonic<M>(lds, a.twB, k);
            a.mft[(size_t)n * M + (k - 1)] = d;
            const double p = cnorm(d);
            a.msq[(size_t)n * M + (k - 1)] = p;
            s += p;
            mx = fmax(mx, p);
        }
        s = group_sum<64>(s);
        mx = group_max<64>(mx);
        if (T > 64) {
            if ((tid & 63) == 0) { red[2 * (tid >> 6)] = s; red[2 * (tid >> 6) + 1] = mx; }
            __syncthreads();
            if (tid == 0) {
                s = 0.0; mx = 0.0;
                for (int w = 0; w < T / 64; ++w) { s += red[2 * w]; mx = fmax(mx, red[2 * w + 1]); }
            }
        }
        if (tid == 0) {
            a.msum[n] = s; a.mmax[n] = mx;
            const cplx z0 = lds[0];
            a.mdc[n] = z0.x + z0.y;
        }
        __syncthreads();
    }
}

// per channel: last harmonic k (1-based) with |m_nk|^2 > eps2 * max_k |m_nk|^2,
// rounded up to a multiple of 64 (>= 64, <= M) -> kt[n]; kmax = max_n kt[n]
__global__ void k_model_kcut(const cplx* mft, const double* mmax, int /*nchan*/, int M, double eps2,
                             int* kt, int* kmax) {
    const int n = blockIdx.x;
    const double thr = eps2 * mmax[n];
    int last = 0;
    for (int k = 1 + threadIdx.x; k <= M; k += blockDim.x)
        if (cnorm(mft[(size_t)n * M + k - 1]) > thr) last = k;
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if ((threadIdx.x & 63) == 0) {
        const int v = min(M, max(64, ((last + 63) / 64) * 64));
        kt[n] = v;
        atomicMax(kmax, v);
    }
}

// --------------------------------------------------------------------------
// data rFFT + cross-spectrum.  Rows are ordered channel-major (row = n*nsub+i)
// and every workgroup of the persistent grid takes a contiguous run of them, so
// its template row changes once per nsub rows (modes 2/3 keep it in registers;
// modes 0/1 re-read it from L2).  The next row's samples are prefetched into
// registers while the current row is transformed, so each resident workgroup
// always has one row of HBM loads in flight.  Only the harmonics the (truncated)
// template keeps go through the even/odd split; S_d needs no split at all because
//   sum_{k=1}^{M-1} |d_k|^2 = sum_{k=1}^{M-1} |Z_k|^2   (Z = packed complex FFT),
// and d_M = Re Z_0 - Im Z_0.
// TAIL: also measure the noise from the top quarter of the power spectrum
// (errs == NULL).
// --------------------------------------------------------------------------
// Rows are dealt to the persistent grid in chunks of PP_ROW_CHUNK consecutive rows
// (= subints of one channel): workgroup b starts with chunk b and draws every
// further chunk from a device counter.  With equal shares per workgroup the two
// waves of a SIMD do not finish together -- the SIMD issues its older wave first,
// so that wave ran through its share in 13.9 ms and left the younger one alone for
// the last 2.7 ms of 16.6 (tools/dev_xspec_stamps.py: ends bimodal, mean / max
// 0.87-0.90), and the XCDs hold clocks 5 % apart.  Drawn on demand, every wave works
// until the rows run out.  The ticket is drawn (lane 0, one returning atomic) at the
// first row of a chunk and read where the last row of the chunk queues its prefetch:
// everything older than that prefetch has landed by then, the atomic included.  The
// counter is never reset: a launch consumes exactly one ticket per chunk, which the
// host adds to the base it passes to the next launch (32 bits, wrapping: only the
// difference is used).  The template row is re-read (L2) once per chunk.
#ifndef PP_ROW_CHUNK
#define PP_ROW_CHUNK 32
#endif
// DYN = false (workgroups of several waves: the ticket would have to cross waves):
// chunk c goes to workgroup c mod G; the tickets are then not drawn at all.
// Rows in use (XspecArgs::mwords): one 32-bit word per chunk, bit b set = row 32 c + b is to be
// transformed (k_mask_words builds them from the batch's chan_mask; nullptr = every row, i.e. full
// words).  The walk visits the set bits of a chunk's word only -- a channel the mask removes from a
// subint is neither read nor transformed, as the reference slices the good channels away before
// its fit (pptoas.py:384-397).  All of it is scalar work on a handful of 32-bit registers (the
// transform kernels have none to spare: rows are counted in 32 bits, nrows < 2^31): the word of the
// NEXT chunk is fetched (one scalar load) while the current one is walked, as soon as the ticket
// that names it has arrived; an empty chunk costs its visitor one more (synchronous) ticket.  Every
// in-range chunk draws exactly one ticket.
template <bool DYN>
struct RowWalk {
    unsigned row, row_nx;      // this row, the row after it
    unsigned bits;             // rows of this chunk still to come (after `row`)
    unsigned wnx, cnx;         // masked walk: the next chunk's word and index, once fetched
    unsigned tick;             // lane 0: the ticket; kept per-lane (not uniform) so that it stays
                               // in a VGPR and nothing waits for the atomic before next() reads it
    // (32-bit flags, not bool: byte-sized members of this struct ended up in scratch memory, and a
    // scratch load queues behind the prefetched row like every other vector-memory access)
    unsigned more, more_nx;
    unsigned fresh, have_wnx;  // `row` is the first row visited of its chunk; (wnx, cnx) are valid
    static __device__ __forceinline__ unsigned word_of(const unsigned* mw, unsigned c, unsigned nrows) {
        // (32-bit arithmetic throughout -- there is no scalar 64-bit ordered compare, and a flag made by
        // the vector ALU lives in a vector register: c < 2^27 because nrows < 2^31 and the grid is small)
        const unsigned r0 = c * PP_ROW_CHUNK;
        if (r0 >= nrows) return 0u;
        if (mw) return as_global(mw)[c];
        const unsigned rem = nrows - r0;
        return rem >= 32u ? 0xffffffffu : ((1u << rem) - 1u);
    }
    // the chunk a visitor of chunk c moves on to (DYN: one synchronous ticket)
    static __device__ __forceinline__ unsigned chunk_after(unsigned c, unsigned* ticket, unsigned base) {
        if (DYN) {
            unsigned t = 0;
            if (threadIdx.x == 0) t = atomicAdd(ticket, 1u);
            return gridDim.x + (__builtin_amdgcn_readfirstlane(t) - base);
        }
        return c + gridDim.x;
    }
    __device__ __forceinline__ void enter(unsigned c, unsigned w, unsigned nrows) {
        more_nx = (c * PP_ROW_CHUNK < nrows) ? 1u : 0u;
        row_nx = c * PP_ROW_CHUNK + (w ? (unsigned)__builtin_ctz(w) : 0u);
        bits = w & (w - 1u);
    }
    __device__ __forceinline__ void start(long long nrows_, const unsigned* mw = nullptr, unsigned* ticket = nullptr,
                                          unsigned base = 0) {
        static_assert(PP_ROW_CHUNK == 32, "one 32-bit word per chunk");
        const unsigned nrows = (unsigned)nrows_;
        wnx = 0; cnx = 0; have_wnx = 0u;
        unsigned c = blockIdx.x;
        unsigned w = word_of(mw, c, nrows);
        // the first chunk with a row in use (an empty one still owes its ticket)
        while (w == 0u && c * PP_ROW_CHUNK < nrows) {
            c = chunk_after(c, ticket, base);
            w = word_of(mw, c, nrows);
        }
        enter(c, __builtin_amdgcn_readfirstlane(w), nrows);
        row = row_nx; more = more_nx;
        tick = threadIdx.x;
        fresh = 1u;
    }
    // top of a row: draw the ticket of the chunk after this one
    __device__ __forceinline__ void draw(unsigned* ticket) {
        if (DYN && fresh && threadIdx.x == 0) tick = atomicAdd(ticket, 1u);
    }
    // masked walk, top of a row that is not the first of its chunk: the ticket has long arrived --
    // fetch the word of the chunk it names, to be looked at when this chunk runs out
    __device__ __forceinline__ void peek(long long nrows_, unsigned base, const unsigned* mw) {
        if (mw && !fresh && !have_wnx) {
            cnx = DYN ? gridDim.x + (__builtin_amdgcn_readfirstlane(tick) - base) : (row >> 5) + gridDim.x;
            wnx = word_of(mw, cnx, (unsigned)nrows_);
            have_wnx = 1u;
        }
    }
    // (subint, channel) of the row after this one: the next row in use of the chunk, or the first
    // of the chunk the ticket names (one division per chunk, all scalar)
    __device__ __forceinline__ void next(int i, int n, int& i_nx, int& n_nx, long long nrows_, int nsub,
                                         unsigned base, unsigned* ticket = nullptr, const unsigned* mw = nullptr) {
        const unsigned nrows = (unsigned)nrows_;
        if (bits) {
            const unsigned b = (unsigned)__builtin_ctz(bits);
            bits &= bits - 1u;
            row_nx = (row & ~(unsigned)(PP_ROW_CHUNK - 1)) + b;
            i_nx = i + (int)(row_nx - row); n_nx = n;
            while (i_nx >= nsub) { i_nx -= nsub; ++n_nx; }
            more_nx = 1u;
            return;
        }
        // this chunk is done: the one its ticket names, or the first after it with a row in use
        unsigned c, w;
        // (have_wnx is cleared HERE and not in enter(): two stores of constants to different members
        // on joining paths are merged by the compiler into one store through a selected pointer, which
        // keeps the whole walk in scratch memory)
        if (have_wnx) { c = cnx; w = wnx; have_wnx = 0u; }
        else {
            c = DYN ? gridDim.x + (__builtin_amdgcn_readfirstlane(tick) - base) : (row >> 5) + gridDim.x;
            w = word_of(mw, c, nrows);
        }
        while (w == 0u && c * PP_ROW_CHUNK < nrows) {
            c = chunk_after(c, ticket, base);
            w = word_of(mw, c, nrows);
        }
        enter(c, __builtin_amdgcn_readfirstlane(w), nrows);
        if (more_nx) {
            n_nx = __builtin_amdgcn_readfirstlane((int)(row_nx / (unsigned)nsub));
            i_nx = __builtin_amdgcn_readfirstlane((int)(row_nx % (unsigned)nsub));
        }
    }
    // the same for a caller that reads its indices off the row number itself
    __device__ __forceinline__ void next_row(long long nrows_, unsigned base, unsigned* ticket, const unsigned* mw) {
        const unsigned nrows = (unsigned)nrows_;
        if (bits) {
            const unsigned b = (unsigned)__builtin_ctz(bits);
            bits &= bits - 1u;
            row_nx = (row & ~(unsigned)(PP_ROW_CHUNK - 1)) + b;
            more_nx = 1u;
            return;
        }
        unsigned c = DYN ? gridDim.x + (__builtin_amdgcn_readfirstlane(tick) - base) : (row >> 5) + gridDim.x;
        unsigned w = word_of(mw, c, nrows);
        while (w == 0u && c * PP_ROW_CHUNK < nrows) {
            c = chunk_after(c, ticket, base);
            w = word_of(mw, c, nrows);
        }
        enter(c, __builtin_amdgcn_readfirstlane(w), nrows);
    }
    __device__ __forceinline__ void advance() {
        // (a row is the first of its chunk when the walk has just changed chunks)
        fresh = (((row_nx ^ row) >> 5) != 0u) ? 1u : 0u;
        row = row_nx; more = more_nx;
    }
};

#ifndef PP_SPLIT_U
#define PP_SPLIT_U 4          // harmonics per thread processed together in the split loop
#endif
#ifndef PP_OPAQUE_ROW
#define PP_OPAQUE_ROW 2         // 0 never, 1 always, 2 only in MODE 2 (register-bound)
#endif
// MODE 0: store X.  MODE 1: store X and the sums A0, A1, A2 at the initial
// parameters.  MODE 2: store NO cross-spectrum, only the Taylor model of every
// channel about the initial parameters (A_0..A_PP_TJ + remainder coefficient,
// see k_eval_moments): the fit then needs no further pass over the data.
// MODE 2 requires 2 Kt < M (each thread then owns harmonics k only); MODE 3 is
// the same for any Kt <= M: harmonic M-k is formed with k from the same two
// transform outputs.
template <int M, typename Tin, bool TAIL, int MODE>
__global__ __launch_bounds__(FftPlan<M>::T, (FftPlan<M>::T >= 256 ? 1 : 2)) void k_xspec(XspecArgs a) {
    constexpr bool FUSE = (MODE != 0);
    constexpr bool M2 = (MODE == 2 || MODE == 3), PAIR = (MODE == 3);
    constexpr int T = FftPlan<M>::T, R1 = FftPlan<M>::R1, PER1 = FftPlan<M>::PER1;
    constexpr int PL = FftPlan<M>::PADLOG;
    constexpr int NW = T / 64;
    typedef typename RawOf<Tin>::type Raw;
    // the image doubles as scratch of the MODE 2 reduction (one region per wave)
    // one wave per row: S_d and the noise tail ride along with the 12 Taylor sums in
    // the one reduction through LDS (no chain of 6 dependent lane exchanges each)
    constexpr bool RIDE = M2 && NW == 1;
    constexpr int NRED = RIDE ? PP_TSTRIDE + (TAIL ? 2 : 1) : PP_TSTRIDE;
    static_assert(NRED <= 16, "wave_reduce_lds takes 16 values");
    constexpr int WRED = PP_WRED_DOUBLES(NRED) / 2;   // in cplx
    constexpr int LDSN = (M2 && NW * WRED > FftPlan<M>::LDS_ELEMS) ? NW * WRED : FftPlan<M>::LDS_ELEMS;
    __shared__ cplx lds[LDSN];
    __shared__ double red[(M2 ? 16 : 5) * NW + 4];
    int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    const int H = M + 1;
    const int kc = (int)(0.75 * H);   // get_noise_PS: int((1 - 1/4) * len(pows))
    Raw cur[PER1][R1];
    RowTwiddles<M> tw;
    load_row_twiddles<M>(tw, a.twB, tid);
    // W_B^(tid+1) and W_B^T: split twiddles by recurrence (no loads in the loop)
    cplx wb0 = a.twB[min(tid + 1, M)];
    const cplx wbT = a.twB[min(T, M)];
    // Each block takes a contiguous run of rows in (channel, subint) order: the
    // channel -- hence the template row -- changes once per nsub rows, and the
    // (subint, channel) indices advance without divisions.
    RowWalk<(NW == 1)> rw;
    rw.start(nrows, a.mwords, a.ticket, a.ticket_base);
    long long row = rw.row;
    int n = 0, i = 0;
    if (rw.more) {
        n = __builtin_amdgcn_readfirstlane((int)(row / a.nsub));
        i = __builtin_amdgcn_readfirstlane((int)(row % a.nsub));
        const size_t rc = (size_t)sub_of(a.act, i) * a.nchan_full + (a.coff + n * a.cstep);
        stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rc * (2 * M), tid);
    }
    // MODE 2 keeps this thread's harmonics of the template row in registers; they
    // are reloaded only when the row changes, so the split issues no vector loads
    // (which would have to wait behind the prefetch of the next data row)
    constexpr int KPT = (M / 2 + T - 1) / T;
    cplx mv2[KPT];
    // MODE 3: the partners' template values m_{M-k} too, where registers allow
    constexpr bool CRES = PAIR && M != 1024;
    cplx mc2[CRES ? KPT : 1];
    const cplx* mheld = nullptr;
    const cplx* mrow = nullptr;    // the template row and cut of the channel in hand (channel_lookup)
    int n_held = -1, ktn = 0;
    int i_nx = i, n_nx = n;
    for (; rw.more; rw.advance(), row = rw.row, i = i_nx, n = n_nx) {
        rw.draw(a.ticket);
        rw.peek(nrows, a.ticket_base, a.mwords);
        // Everything derived from the thread index and the twiddles is invariant
        // over this loop, and the compiler hoists all of it (LDS addresses of every
        // stage, twiddle powers: ~50 VGPRs held across the whole row).  Recomputing
        // them per row frees those registers; it only pays where that buys occupancy.
        if (PP_OPAQUE_ROW == 1 || (PP_OPAQUE_ROW == 2 && M2)) {
            asm volatile("" : "+v"(tid));
            opaque_twiddles<M>(tw);
            asm volatile("" : "+v"(wb0.x), "+v"(wb0.y));   // or all of wb0 wbT^j are hoisted
        }
        const int ia = sub_of(a.act, i), ne = a.coff + n * a.cstep;   // true subint, channel
        const size_t rc = (size_t)ia * a.nchan_full + ne;
        const size_t rx = (size_t)i * a.nchan + n;                    // compact row of X
        // Issue, BEFORE anything waits, every load of this row whose result is
        // needed late: vector-memory results return in order, so these must be
        // older than the prefetch of the next row or consuming them would drain it.
        // (template row; harmonics this channel's template keeps, a multiple of 64)
        const bool looked = channel_lookup(a, ia, n, ne, M, n_held, mrow, ktn);
        if (M2 && looked && mrow != mheld) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const int k = tid + 1 + j * T;
                // Unconditional loads from a clamped index, and no arithmetic on them
                // here: the row is re-read once per chunk of rows, loads under a branch
                // are waited for one by one, and anything computed from them now would
                // wait in front of the transform.  The split forms 2 d_k (its factors
                // 1/2 are applied to the 12 sums at the end -- exact) and skips the
                // harmonics beyond ktn itself.
                mv2[j] = mrow[min(k, M) - 1];
                if (CRES) mc2[j] = mrow[max(M - k, 1) - 1];
            }
            mheld = mrow;
        }
        // phi_n at the initial parameters (k_phase0); loaded here, before the
        // prefetch of the next row is queued
        double phin = 0.0;
        if (FUSE) phin = a.ph0[rc];
        {
            cplx v[PER1][R1];
#pragma unroll
            for (int ii = 0; ii < PER1; ++ii)
#pragma unroll
                for (int k = 0; k < R1; ++k) v[ii][k] = to_cplx(cur[ii][k]);
            // The next row's HBM loads are queued as soon as this row's registers are
            // dead -- after the first quarter of the first stage, not after its twiddles
            // and stores: a wave has one row in flight, and whatever part of a row's time
            // it spends with nothing in flight the memory system idles for
            // (unconditional -- the last row of the run fetches itself again: a prefetch
            // under a branch makes the compiler drain the whole queue, vmcnt(0), before
            // every use of an earlier load, because on the path without it no younger
            // loads exist)
            // (the row after this one is decided outside the lambda: see k_xspec_q1024)
            rw.next(i, n, i_nx, n_nx, nrows, a.nsub, a.ticket_base, a.ticket, a.mwords);
            const size_t rn = rw.more_nx
                ? (size_t)sub_of(a.act, i_nx) * a.nchan_full + (a.coff + n_nx * a.cstep) : rc;
            const Tin* const nxrow = reinterpret_cast<const Tin*>(a.data) + rn * (2 * M);
            auto prefetch = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                stage_load_global<M, T, R1>(cur, nxrow, tid);
                __builtin_amdgcn_sched_barrier(0);
            };
            fft_first_stage<M, M2>(lds, v, tw, tid, prefetch);
        }
        // ---- S_d comes out of the last stage's registers ----
        double sd = 0.0, tail = 0.0;
        fft_later_stages<M>(lds, tw, tid, &sd);
        __builtin_amdgcn_sched_barrier(0);
        if (TAIL) {
            for (int k = kc + tid; k <= M; k += T) tail += cnorm(rfft_harmonic<M>(lds, a.twB, k));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- cross-spectrum (and the first evaluation's sums) ----
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        cplx e = make_double2(1.0, 0.0), wst = make_double2(1.0, 0.0);
        auto setup_phasors = [&]() {
            // e^{2 pi i (tid+1) phi}: one sincos per lane (k = lane+1); lane 63
            // holds e^{2 pi i 64 phi}, whose powers give the wave offset and the
            // per-iteration step e^{2 pi i T phi}
            const cplx el = unit_phasor<true>((double)((tid & 63) + 1), phin);
            const cplx w64 = make_double2(__shfl(el.x, 63, 64), __shfl(el.y, 63, 64));
            e = el;
            for (int q = 0; q < (tid >> 6); ++q) e = cmul(e, w64);
            wst = w64;
#pragma unroll
            for (int q = 1; q < NW; q <<= 1) wst = cmul(wst, wst);
        };
        if (MODE == 1) setup_phasors();
        cplx wb = wb0;
        double tm[PP_TSTRIDE];
        if (M2) {
            // Split and Taylor sums in one sweep over this thread's harmonics.  T is
            // a multiple of the padding period: the padded slots of k + jT and of
            // M - k - jT are affine in j (constant LDS offsets).
            static_assert(T % (1 << PL) == 0, "padding period must divide the block size");
            static_assert(PP_TJ == 10, "power ladder written for order 10");
            constexpr int JS = T + (T >> PL);
            const cplx* pk = lds + lds_pad<PL>(tid + 1);
            const cplx* pc = lds + lds_pad<PL>(M - 1 - tid);
            setup_phasors();
            // One wave per row and a template cut that is a multiple of 64 (k_model_kcut):
            // a slot of 64 harmonics is kept or dropped as a whole -- a scalar branch, no
            // per-lane compare / exec mask --, slot 0 is always kept and starts the sums
            // (no zeroing), and kappa_k = 2 pi (lane + 1) + 2 pi 64 j takes one addition
            // of a scalar constant instead of a conversion and a product.
            constexpr bool UNI = (T == 64) && (M >= 64) && !PAIR;
            const int ktu = UNI ? __builtin_amdgcn_readfirstlane(ktn) : 0;
            const double kap0 = PP_TWO_PI * (double)(tid + 1);
            if (!UNI) {
#pragma unroll
                for (int j = 0; j < PP_TSTRIDE; ++j) tm[j] = 0.0;
            }
            // kappa^2, ^4 .. ^10 once per harmonic; every sum is then one FMA
            // (first: the sums start from this harmonic)
            auto taylor_sums = [&](const cplx& x, const cplx& z, double kap, bool first) {
                const double p2 = kap * kap, p4 = p2 * p2, p6 = p4 * p2, p8 = p4 * p4, p10 = p8 * p2;
                const double ui = z.y * kap;
                const double ax = fabs(x.x) + fabs(x.y);
                if (first) {
                    tm[0] = z.x;
                    tm[1] = ui;
                    tm[2] = p2 * z.x;
                    tm[3] = p2 * ui;
                    tm[4] = p4 * z.x;
                    tm[5] = p4 * ui;
                    tm[6] = p6 * z.x;
                    tm[7] = p6 * ui;
                    tm[8] = p8 * z.x;
                    tm[9] = p8 * ui;
                    tm[10] = p10 * z.x;
                    tm[11] = (p10 * kap) * ax;
                    return;
                }
                tm[0] += z.x;
                tm[1] += ui;
                tm[2] = fma(p2, z.x, tm[2]);
                tm[3] = fma(p2, ui, tm[3]);
                tm[4] = fma(p4, z.x, tm[4]);
                tm[5] = fma(p4, ui, tm[5]);
                tm[6] = fma(p6, z.x, tm[6]);
                tm[7] = fma(p6, ui, tm[7]);
                tm[8] = fma(p8, z.x, tm[8]);
                tm[9] = fma(p8, ui, tm[9]);
                tm[10] = fma(p10, z.x, tm[10]);
                tm[11] = fma(p10 * kap, ax, tm[11]);
            };
            cplx eM = wst;     // e^{2 pi i M phi}: the partner of e_k is eM conj(e_k)
            if (PAIR) {
                if constexpr (M < T) {
                    // one wave, lane l holds e^{2 pi i (l+1) phi}
                    eM = make_double2(__shfl(e.x, M - 1, 64), __shfl(e.y, M - 1, 64));
                } else {
#pragma unroll
                    for (int q = T; q < M; q <<= 1) eM = cmul(eM, eM);
                }
            }
            // The two transform outputs of slot j + 1 are read (unconditionally: the
            // addresses stay inside the image) before slot j is worked on, so that the
            // LDS round trip of every slot but the first hides under arithmetic.
            // (Not with the noise tail or the paired harmonics: the 8 registers of the
            // look-ahead spill there, and those slots are read in place.)
            constexpr bool AHEAD = !TAIL && !PAIR;
            // (rows shorter than two harmonics per lane: lanes beyond M/2 read slot 0 of lane 0)
            constexpr bool INSIDE = (KPT * T <= M);
            const cplx* pk0 = (INSIDE || tid + 1 <= M / 2) ? pk : lds + lds_pad<PL>(1);
            const cplx* pc0 = (INSIDE || tid + 1 <= M / 2) ? pc : lds + lds_pad<PL>(M - 1);
            cplx zk_nx = pk0[0], zc_nx = pc0[0];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const int k = tid + 1 + j * T;
                const cplx zk = (!AHEAD && j > 0) ? pk0[j * JS] : zk_nx;
                cplx zc = (!AHEAD && j > 0) ? pc0[-j * JS] : zc_nx;
                if (j + 1 < KPT) {
                    if (AHEAD) { zk_nx = pk0[(j + 1) * JS]; zc_nx = pc0[-(j + 1) * JS]; }
                }
                // (pairs: k runs to M/2 only -- the upper half comes as partners;
                // matters when M/2 is not a multiple of the block size)
                const bool keep = UNI ? (j == 0 || j * T < ktu) : (k <= ktn && (!PAIR || 2 * k <= M));
                if (keep) {
                    zc.y = -zc.y;
                    const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
                    const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
                    const cplx wo = cmul(wb, O);
                    // 2 d_k = E - i W^k O
                    const cplx x = cmulc(make_double2(E.x + wo.y, E.y - wo.x), mv2[j]);
                    const double kap = !UNI ? PP_TWO_PI * (double)k
                                            : (j == 0 ? kap0 : kap0 + kconst<true>(PP_TWO_PI * (double)(j * T)));
                    taylor_sums(x, cmul(x, e), kap, UNI && j == 0);
                    if (PAIR) {
                        // 2 d_{M-k} = conj(E) - i conj(W^k O)   (W^{M-k} = -conj W^k)
                        const int kp = M - k;
                        if (kp <= ktn && kp != k) {
                            cplx mp;
                            if (CRES) mp = mc2[j];
                            else mp = mrow[kp - 1];
                            const cplx xp = cmulc(make_double2(E.x - wo.y, -E.y - wo.x), mp);
                            taylor_sums(xp, cmul(xp, cmulc(eM, e)), PP_TWO_PI * (double)kp, false);
                        }
                    }
                }
                wb = cmul(wb, wbT);
                e = cmul(e, wst);
            }
            if (PAIR && ktn == M && tid == 0) {
                // Nyquist harmonic: d_M = Re Z_0 - Im Z_0
                const cplx z0 = lds[0], mM = mrow[M - 1];
                const double dM = 2.0 * (z0.x - z0.y);     // (the sums are halved at the end)
                const cplx x = make_double2(dM * mM.x, -dM * mM.y);
                taylor_sums(x, cmul(x, eM), PP_TWO_PI * (double)M, false);
            }
        }
        for (int kb = 1 + tid; !M2 && kb <= ktn; kb += PP_SPLIT_U * T) {
            cplx mv[PP_SPLIT_U];   // independent model loads in flight per chunk
#pragma unroll
            for (int j = 0; j < PP_SPLIT_U; ++j) {
                const int k = kb + j * T;
                mv[j] = (k <= ktn) ? mrow[k - 1] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < PP_SPLIT_U; ++j) {
                const int k = kb + j * T;
                if (k <= ktn) {
                    const cplx d = rfft_harmonic_w<M>(lds, wb, k);
                    const cplx x = cmulc(d, mv[j]);
                    store_x(a, rx, k, x);
                    if (MODE == 1) {
                        const cplx z = cmul(x, e);
                        const double kk = (double)k;
                        s0 += z.x;
                        s1 = fma(kk, z.y, s1);
                        s2 = fma(kk * kk, z.x, s2);
                    }
                }
                wb = cmul(wb, wbT);
                if (MODE == 1) e = cmul(e, wst);
            }
        }
        if (!RIDE) {
            sd = group_sum<64>(sd);
            if (TAIL) tail = group_sum<64>(tail);
        }
        if (MODE == 1) { s0 = group_sum<64>(s0); s1 = group_sum<64>(s1); s2 = group_sum<64>(s2); }
        double tv = 0.0;
        if (RIDE) {
            double tr[NRED];
#pragma unroll
            for (int j = 0; j < PP_TSTRIDE; ++j) tr[j] = tm[j];
            tr[PP_TSTRIDE] = sd;
            if (TAIL) tr[NRED - 1] = tail;
            tv = wave_reduce_lds(tr, tid, reinterpret_cast<double*>(lds));
        } else if (M2) {
            if (NW > 1) lds_sync<T>();   // other waves may still read the transform
            tv = wave_reduce_lds(tm, tid & 63, reinterpret_cast<double*>(lds + (tid >> 6) * WRED));
        }
        if (NW > 1) {
            if (M2) {
                if (((tid & 63) & 3) == 0) red[16 * (tid >> 6) + wave_reduce16_index(tid & 63)] = tv;
                lds_sync<T>();
                if (tid < 64) {
                    tv = 0.0;
                    for (int w = 0; w < NW; ++w) tv += red[16 * w + wave_reduce16_index(tid)];
                }
                lds_sync<T>();
            }
            if ((tid & 63) == 0) {
                double* r = red + 5 * (tid >> 6);
                r[0] = sd; r[1] = tail; r[2] = s0; r[3] = s1; r[4] = s2;
            }
            lds_sync<T>();
            if (tid == 0) {
                sd = tail = s0 = s1 = s2 = 0.0;
                for (int w = 0; w < NW; ++w) {
                    sd += red[5 * w]; tail += red[5 * w + 1];
                    s0 += red[5 * w + 2]; s1 += red[5 * w + 3]; s2 += red[5 * w + 4];
                }
            }
        }
        if (M2 && tid < 64 && (tid & 3) == 0) {
            const int q = wave_reduce16_index(tid);
            if (q < PP_TSTRIDE) {
                // Re(i^q z): +Re, -Im, -Re, +Im, ...
                // (x 1/2: the template values were used unhalved against 2 d_k)
                tv *= 0.5;
                a.tay[tay_idx(rc, q)] = (q <= PP_TJ && ((q & 3) == 1 || (q & 3) == 2)) ? -tv : tv;
            }
        }
        if (RIDE) {
            // lane quads 12 and 13 hold the totals of S_d and of the tail
            if (tid == 4 * PP_TSTRIDE) a.sdraw[rc] = tv;
            if (TAIL && tid == 4 * (PP_TSTRIDE + 1)) a.noise[rc] = sqrt(tv / (2.0 * M) / (double)(H - kc));
        } else if (tid == 0) {
            a.sdraw[rc] = sd;
            if (TAIL) a.noise[rc] = sqrt(tail / (2.0 * M) / (double)(H - kc));
            if (MODE == 1) {
                double* co = a.csum0 + rc * 3;
                co[0] = s0;
                co[1] = -PP_TWO_PI * s1;
                co[2] = -PP_TWO_PI * PP_TWO_PI * s2;
            }
        }
        lds_sync<T>();
    }
}

}  // namespace pp
#include "pp_xspec1024.h"
#include "pp_xspec1024q.h"
#include "pp_xspec1024s.h"
namespace pp {

// plain rFFT of rows (parity hook): out[row][0..M] complex
template <int M, typename Tin>
__global__ __launch_bounds__(FftPlan<M>::T) void k_rfft_rows(const void* in, cplx* out, const cplx* twB,
                                                             int nrows) {
    constexpr int T = FftPlan<M>::T;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    const int tid = threadIdx.x;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const Tin* grow = reinterpret_cast<const Tin*>(in) + (size_t)r * (2 * M);
        fft_row<M, Tin>(lds, grow, twB, tid);
        for (int k = 1 + tid; k <= M; k += T) out[(size_t)r * (M + 1) + k] = rfft_harmonic<M>(lds, twB, k);
        if (tid == 0) {
            const cplx z0 = lds[0];
            out[(size_t)r * (M + 1)] = make_double2(z0.x + z0.y, 0.0);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// weights, solver state
// --------------------------------------------------------------------------
// wts = mask / (errs^2 * B/2); errs == nullptr -> measured noise
__global__ void k_prep(int nsub, int nchan, int nbin, const double* errs, const double* noise,
                       const unsigned char* mask, double* wts) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nsub * nchan) return;
    const double e = errs ? errs[idx] : noise[idx];
    const double w = 1.0 / (e * e * (0.5 * nbin));
    wts[idx] = (mask && !mask[idx]) ? 0.0 : w;
}

__device__ inline void init_state(const FitArgs& a, int i) {
    SubState& s = a.st[i];
    for (int j = 0; j < 5; ++j) { s.x[j] = a.x0[i * 5 + j]; s.xe[j] = s.x[j]; s.g[j] = 0.0; }
    for (int j = 0; j < 25; ++j) s.H[j] = 0.0;
    s.f = 0.0;
    s.radius = 1.0;          // scipy initial_trust_radius
    s.pred_red = 0.0;
    s.hits_boundary = 0;
    s.iter = 0; s.nfev = 0; s.npass = 0; s.status = PP_RC_MAXITER; s.done = 0; s.cur = 1; s.fresh = 1;
    s.model = 0; s.geo[0] = s.geo[1] = s.geo[2] = s.geo[3] = 0.0;
    s.recentred = 0; s.nmodel = 0;
    for (int j = 0; j < 5; ++j) s.xl[j] = NAN;          // (no point evaluated yet)
    s.fl = NAN;
    for (int j = 0; j < 5; ++j) s.xprev[j] = s.x[j];
    if (i == 0) *a.nactive = a.nsub;
}

__global__ void k_init_state(FitArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.nsub) init_state(a, i);
}

// Newton solver, scattering fits: the iteration has first been run on every 16th channel (a
// sixteenth of every evaluation pass); the full-channel iteration now starts from that answer.
// Subints whose coarse solve did not end on its normal exit keep their initial parameters.
__global__ void k_adopt_coarse(FitArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nsub) return;
    SubState& s = a.st[i];
    bool ok = (s.done && s.status == PP_RC_STALL);
    for (int j = 0; j < 5; ++j) ok = ok && isfinite(s.x[j]);
    const int nf = s.nfev;
    if (ok) for (int j = 0; j < 5; ++j) a.x0w[i * 5 + j] = s.x[j];
    init_state(a, i);
    a.st[i].nfev = nf;          // (the coarse evaluations stay counted; npass counts full passes only)
}

// per-channel geometry shared by evaluator and finaliser
struct ChanGeom {
    double p1, p2;               // d phi_n / d DM, / d GM
    double taun, q1, q2, q11, q12, q22, lnf;
};

__device__ __forceinline__ void chan_geom(double nu, double P, double nuDM, double nuGM, double nutau,
                                          double tau, double alpha, int log10_tau, bool scat_on,
                                          ChanGeom& c) {
    const double a2 = 1.0 / (nu * nu);
    const double iDM = (nuDM == INFINITY) ? 0.0 : 1.0 / (nuDM * nuDM);
    const double iGM = (nuGM == INFINITY) ? 0.0 : 1.0 / (nuGM * nuGM * nuGM * nuGM);
    c.p1 = PP_DCONST * (a2 - iDM) / P;
    c.p2 = PP_DCONST * PP_DCONST * (a2 * a2 - iGM) / P;
    if (!scat_on) {
        // tau = 0: every scattering quantity vanishes (log / pow skipped)
        c.lnf = c.taun = c.q1 = c.q2 = c.q11 = c.q12 = c.q22 = 0.0;
        return;
    }
    const double r = nu / nutau;
    c.lnf = log(r);
    c.taun = tau * pow(r, alpha);
    if (!log10_tau) {
        c.q1 = scat_on ? c.taun / tau : 0.0;
        c.q2 = c.lnf * c.taun;
        c.q11 = 0.0;
        c.q12 = scat_on ? c.q2 / tau : 0.0;
    } else {
        c.q1 = PP_LN10 * c.taun;
        c.q2 = c.lnf * c.taun;
        c.q11 = PP_LN10 * c.q1;
        c.q12 = PP_LN10 * c.q2;
    }
    c.q22 = c.lnf * c.q2;
}

// phase-model part only (no scattering): d phi_n / d DM, d phi_n / d GM
__device__ __forceinline__ void phase_geom(double nu, double P, double nuDM, double nuGM, double& p1,
                                           double& p2) {
    const double a2 = 1.0 / (nu * nu);
    const double iDM = (nuDM == INFINITY) ? 0.0 : 1.0 / (nuDM * nuDM);
    const double iGM = (nuGM == INFINITY) ? 0.0 : 1.0 / (nuGM * nuGM * nuGM * nuGM);
    p1 = PP_DCONST * (a2 - iDM) / P;
    p2 = PP_DCONST * PP_DCONST * (a2 * a2 - iGM) / P;
}

// phi_n of every (subint, channel) at the initial parameters: the expansion
// point of the Taylor model k_xspec accumulates (reference order of operations,
// pptoaslib.py:181-198)
__global__ __launch_bounds__(256) void k_phase0(int nsub, int nchan, const double* x0, const double* P,
                                                const double* nu_fit, const double* freqs, int freqs_stride,
                                                double* ph0) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)nsub * nchan) return;
    const int i = (int)(idx / nchan), n = (int)(idx % nchan);
    double p1, p2;
    phase_geom(freqs[(size_t)i * freqs_stride + n], P[i], nu_fit[i * 3], nu_fit[i * 3 + 1], p1, p2);
    ph0[idx] = x0[i * 5] + x0[i * 5 + 1] * p1 + x0[i * 5 + 2] * p2;
}

// Everything a batch needs before its transform, in one launch: phi_n at the initial
// parameters (ph0 != nullptr: k_phase0), the weights when the noise is given (errs != nullptr:
// k_prep; measured noise only exists after the transform) and the solver state (do_init:
// k_init_state).
__global__ __launch_bounds__(256) void k_setup(FitArgs a, const double* errs, const unsigned char* mask,
                                               double* wts, double* ph0, double* tau0, int do_init) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.nsub * a.nchan) return;
    const int i = (int)(idx / a.nchan), n = (int)(idx % a.nchan);
    if (ph0) {
        double p1, p2;
        phase_geom(a.freqs[(size_t)i * a.freqs_stride + n], a.P[i], a.nu_fit[i * 3], a.nu_fit[i * 3 + 1], p1, p2);
        ph0[idx] = a.x0[i * 5] + a.x0[i * 5 + 1] * p1 + a.x0[i * 5 + 2] * p2;
    }
    if (tau0) {
        // tau_n at the initial parameters (scattering fits whose first evaluation rides in the transform)
        const double tau = a.log10_tau ? pow(10.0, a.x0[i * 5 + 3]) : a.x0[i * 5 + 3];
        ChanGeom cg;
        chan_geom(a.freqs[(size_t)i * a.freqs_stride + n], a.P[i], a.nu_fit[i * 3], a.nu_fit[i * 3 + 1], a.nu_fit[i * 3 + 2],
                  tau, a.x0[i * 5 + 4], a.log10_tau, tau != 0.0, cg);
        tau0[idx] = cg.taun;
    }
    if (errs) {
        const double e = errs[idx];
        const double w = 1.0 / (e * e * (0.5 * a.nbin));
        wts[idx] = (mask && !mask[idx]) ? 0.0 : w;
    }
    if (do_init && idx < a.nsub) init_state(a, (int)idx);
}

// The rows in use of a batch as RowWalk wants them: one 32-bit word per chunk of PP_ROW_CHUNK
// rows, in two row orders -- `wmain`: row = n nsub + i (the transform kernels' channel-major
// order); `wsub` (optional): row = (cb nsub + i) 32 + b for channel 32 cb + b of subint i
// (k_xspec_qr1024's chunks of 32 channels of one subint: the mask's own layout).  A block takes
// 32 subints x 256 channels of the mask through LDS (coalesced reads) and every thread assembles
// the 32 subints of its channel; words that straddle two channels (nsub not a multiple of 32) are
// merged with atomicOr, so both arrays must be zero on entry.  grid = (ceil(nchan / 256), ceil(nsub / 32)).
__global__ __launch_bounds__(256) void k_mask_words(const unsigned char* mask, int nsub, int nchan, unsigned* wmain,
                                                    unsigned* wsub) {
    __shared__ unsigned char tile[32][256 + 4];
    const int tid = threadIdx.x, n0 = blockIdx.x * 256, i0 = blockIdx.y * 32;
    const int n = n0 + tid;
    for (int b = 0; b < 32; ++b) {
        const int i = i0 + b;
        tile[b][tid] = (i < nsub && n < nchan) ? mask[(size_t)i * nchan + n] : (unsigned char)0;
    }
    __syncthreads();
    if (n < nchan) {
        unsigned w = 0;
        for (int b = 0; b < 32; ++b) w |= (tile[b][tid] ? 1u : 0u) << b;
        const long long r0 = (long long)n * nsub + i0;      // first row of these 32
        const int sh = (int)(r0 & 31);
        if (w) {
            if (sh == 0) atomicOr(&wmain[r0 >> 5], w);
            else {
                atomicOr(&wmain[r0 >> 5], w << sh);
                if (w >> (32 - sh)) atomicOr(&wmain[(r0 >> 5) + 1], w >> (32 - sh));
            }
        }
    }
    if (wsub && tid < 32 * 8) {
        // (subint i0 + tid / 8, channel block (n0 / 32) + tid % 8): 32 consecutive channels of one subint
        const int b = tid >> 3, q = tid & 7, i = i0 + b, cb = (n0 >> 5) + q;
        if (i < nsub && cb * 32 < nchan) {
            unsigned w = 0;
            for (int k = 0; k < 32; ++k) w |= (tile[b][q * 32 + k] ? 1u : 0u) << k;
            wsub[(size_t)cb * nsub + i] = w;
        }
    }
}

// local (phi_n, tau_n) derivatives of F_n = -C^2/S from weighted sums
struct Local {
    double F, Gp, Gt, Lpp, Lpt, Ltt;
};
__device__ __forceinline__ Local local_terms(const double* s /*9 raw sums*/, double w) {
    const double A0 = s[0], A1 = s[1], A2 = s[2], T1 = s[3], T2 = s[4], A1T = s[5];
    const double S0 = s[6], S1 = s[7], S2 = s[8];
    const double iS = 1.0 / S0;
    const double r = A0 * iS;          // = C/S (weights cancel)
    Local L;
    L.F = -w * A0 * r;
    L.Gp = -2.0 * w * r * A1;
    L.Gt = -w * (2.0 * r * T1 - r * r * S1);
    L.Lpp = -2.0 * w * (A1 * A1 * iS + r * A2);
    L.Lpt = -2.0 * w * (r * A1T + A1 * T1 * iS - r * A1 * S1 * iS);
    L.Ltt = -2.0 * w * (r * T2 - 0.5 * r * r * S2 + T1 * T1 * iS + r * r * S1 * S1 * iS -
                        2.0 * r * T1 * S1 * iS);
    return L;
}

// accumulate one channel's local terms into the 21 per-subint sums
__device__ __forceinline__ void accumulate_channel(const Local& L, const ChanGeom& cg, double (&c)[PP_NACC]) {
    const double p1 = cg.p1, p2 = cg.p2;
    c[0] = L.F;
    c[1] = L.Gp; c[2] = L.Gp * p1; c[3] = L.Gp * p2;
    c[4] = L.Gt * cg.q1; c[5] = L.Gt * cg.q2;
    c[6] = L.Lpp; c[7] = L.Lpp * p1; c[8] = L.Lpp * p2;
    c[9] = L.Lpt * cg.q1; c[10] = L.Lpt * cg.q2;
    c[11] = L.Lpp * p1 * p1; c[12] = L.Lpp * p1 * p2;
    c[13] = L.Lpt * p1 * cg.q1; c[14] = L.Lpt * p1 * cg.q2;
    c[15] = L.Lpp * p2 * p2;
    c[16] = L.Lpt * p2 * cg.q1; c[17] = L.Lpt * p2 * cg.q2;
    c[18] = L.Ltt * cg.q1 * cg.q1 + L.Gt * cg.q11;
    c[19] = L.Ltt * cg.q1 * cg.q2 + L.Gt * cg.q12;
    c[20] = L.Ltt * cg.q2 * cg.q2 + L.Gt * cg.q22;
}

// --------------------------------------------------------------------------
// chi^2 evaluators: k_eval_fast below (no scattering) and k_eval_scat (pp_evalscat.h).
// --------------------------------------------------------------------------
#ifndef PP_FAST_RECIP
#define PP_FAST_RECIP 1       // |B_nk|^2 = 1/(1 + u^2) by rcp + 2 Newton steps (~1 ulp)
#endif

// First evaluation when k_xspec already produced the per-channel sums (FUSE):
// only the O(nchan) chain rule + reduction remains.  grid = (nchunk, nsub).
__global__ __launch_bounds__(256) void k_accum(FitArgs a) {
    const int i = sub_of(a.act, blockIdx.x), chunk = blockIdx.y, tid = threadIdx.x;
    SubState& st = a.st[i];
    if (st.done) return;
    __shared__ double scratch[4 * PP_NACC];
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const double* msum = as_global(a.msum[a.slot ? a.slot[i] : 0]);
    const int trial = 1 - st.cur;
    const double* csum = a.csum + ((size_t)trial * a.nsub + i) * a.nchan * a.ncs;
    double acc[PP_NACC];
#pragma unroll
    for (int j = 0; j < PP_NACC; ++j) acc[j] = 0.0;
    const int n0 = chunk * a.cpc, n1 = min(n0 + a.cpc, a.nchan_x);
    if (a.scat) {
        // the nine sums of a scattering fit (k_xspec_qs1024): the whole chain rule of k_eval_scat
        const double tau = a.log10_tau ? pow(10.0, st.xe[3]) : st.xe[3];
        const double nutau = a.nu_fit[i * 3 + 2], alpha = st.xe[4];
        for (int nn = n0 + tid; nn < n1; nn += 256) {
            const int n = a.coff + nn * a.cstep;
            const double w = wts[n];
            if (w == 0.0) continue;
            double cs[PP_NCS];
#pragma unroll
            for (int j = 0; j < PP_NCS; ++j) cs[j] = csum[(size_t)n * PP_NCS + j];
            ChanGeom cg;
            chan_geom(freqs[n], P, nuDM, nuGM, nutau, tau, alpha, a.log10_tau, tau != 0.0, cg);
            const Local L = local_terms(cs, w);
            double c[PP_NACC];
            accumulate_channel(L, cg, c);
#pragma unroll
            for (int j = 0; j < PP_NACC; ++j) acc[j] += c[j];
        }
    } else
    for (int nn = n0 + tid; nn < n1; nn += 256) {
        const int n = a.coff + nn * a.cstep;
        const double w = wts[n];
        if (w == 0.0) continue;
        double cs[PP_NCS];
        cs[0] = csum[(size_t)n * 3]; cs[1] = csum[(size_t)n * 3 + 1]; cs[2] = csum[(size_t)n * 3 + 2];
        cs[3] = cs[4] = cs[5] = 0.0; cs[6] = msum[n]; cs[7] = cs[8] = 0.0;
        double p1, p2;
        phase_geom(freqs[n], P, nuDM, nuGM, p1, p2);
        const double r = cs[0] / cs[6];
        const double F = -w * cs[0] * r, Gp = -2.0 * w * r * cs[1];
        const double Lpp = -2.0 * w * (cs[1] * cs[1] / cs[6] + r * cs[2]);
        acc[0] += F;
        acc[1] += Gp; acc[2] += Gp * p1; acc[3] += Gp * p2;
        acc[6] += Lpp; acc[7] += Lpp * p1; acc[8] += Lpp * p2;
        acc[11] += Lpp * p1 * p1; acc[12] += Lpp * p1 * p2; acc[15] += Lpp * p2 * p2;
    }
    block_sum<PP_NACC>(acc, scratch);
#pragma unroll
    for (int j = 0; j < PP_NACC; ++j)
        if (tid == j) a.partial[((size_t)i * a.nchunk + chunk) * PP_NACC + j] = acc[j];
}

// chi^2 evaluator without scattering (tau_n = 0): only A0, A1, A2 are needed
// and S_n is the constant sum_k |m_nk|^2.  16 lanes per channel, four
// independent 16-byte loads in flight per lane per iteration (Kt is a multiple
// of 64), phasors advanced by e^{2 pi i 64 phi_n}; each lane of a group keeps
// only two of the 21 per-subint accumulators.
__global__ __launch_bounds__(256) void k_eval_fast(FitArgs a) {
    constexpr int LPC = 16;
    const int jx = blockIdx.x, i = sub_of(a.act, jx), chunk = blockIdx.y;
    SubState& st = a.st[i];
    if (st.done) return;
    __shared__ double red[(256 / LPC) * PP_NACC];
    const int tid = threadIdx.x, g = tid / LPC, l = tid % LPC;
    const double phi = st.xe[0], DM = st.xe[1], GM = st.xe[2];
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const double* msum = as_global(a.msum[a.slot ? a.slot[i] : 0]);
    const int* ktv = a.ktab ? as_global(a.ktab[a.slot ? a.slot[i] : 0]) : nullptr;
    const int trial = 1 - st.cur;
    double* csum = a.csum + ((size_t)trial * a.nsub + i) * a.nchan * a.ncs;
    double accA = 0.0;
    const int n0 = chunk * a.cpc, n1 = min(n0 + a.cpc, a.nchan_x);
    const int src = ((tid & 63) & ~(LPC - 1)) | (LPC - 1);
    for (int nn = n0 + g; nn < n1; nn += 256 / LPC) {
        const int n = a.coff + nn * a.cstep;
        const double w = wts[n];
        double p1, p2;
        phase_geom(freqs[n], P, nuDM, nuGM, p1, p2);
        const double phin = phi + DM * p1 + GM * p2;
        cplx e0 = unit_phasor((double)(l + 1), phin);
        const cplx w1 = make_double2(__shfl(e0.x, src, 64), __shfl(e0.y, src, 64));
        const cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
        const cplx* xrow = a.X + ((size_t)jx * a.nchan_x + nn) * a.Xs;
        double s0a = 0, s1a = 0, s2a = 0, s0b = 0, s1b = 0, s2b = 0;
        double k = (double)(l + 1);
        const int ktn = ktv ? ktv[n] : a.Kt;
        if (w != 0.0) {
#pragma unroll 1
            for (int j = l; j < ktn; j += 4 * LPC) {
                // ktn is a multiple of 64 except for nbin < 128 (ktn = M = 16 or 32)
                const cplx zero = make_double2(0.0, 0.0);
                const cplx x0 = xrow[j];
                const cplx x1 = (j + LPC < ktn) ? xrow[j + LPC] : zero;
                const cplx x2 = (j + 2 * LPC < ktn) ? xrow[j + 2 * LPC] : zero;
                const cplx x3 = (j + 3 * LPC < ktn) ? xrow[j + 3 * LPC] : zero;
                const cplx e1 = cmul(e0, w1), e2 = cmul(e0, w2), e3 = cmul(e0, w3);
                const cplx z0 = cmul(x0, e0), z1 = cmul(x1, e1), z2 = cmul(x2, e2), z3 = cmul(x3, e3);
                const double k1 = k + LPC, k2 = k + 2 * LPC, k3 = k + 3 * LPC;
                s0a += z0.x; s1a = fma(k, z0.y, s1a); s2a = fma(k * k, z0.x, s2a);
                s0b += z1.x; s1b = fma(k1, z1.y, s1b); s2b = fma(k1 * k1, z1.x, s2b);
                s0a += z2.x; s1a = fma(k2, z2.y, s1a); s2a = fma(k2 * k2, z2.x, s2a);
                s0b += z3.x; s1b = fma(k3, z3.y, s1b); s2b = fma(k3 * k3, z3.x, s2b);
                e0 = cmul(e0, w4);
                k += 4.0 * LPC;
            }
        }
        const double A0 = group_sum<LPC>(s0a + s0b);
        const double A1 = -PP_TWO_PI * group_sum<LPC>(s1a + s1b);
        const double A2 = -PP_TWO_PI * PP_TWO_PI * group_sum<LPC>(s2a + s2b);
        if (l < 3) csum[(size_t)n * 3 + l] = (l == 0) ? A0 : (l == 1 ? A1 : A2);
        if (w != 0.0) {
            const double S0 = msum[n], r = A0 / S0;
            const double F = -w * A0 * r, Gp = -2.0 * w * r * A1;
            const double Lpp = -2.0 * w * (A1 * A1 / S0 + r * A2);
            // the 10 non-zero accumulators, one per lane: f, g[0..2], H phi-block
            double ca = F;
            ca = (l == 1) ? Gp : ca;
            ca = (l == 2) ? Gp * p1 : ca;
            ca = (l == 3) ? Gp * p2 : ca;
            ca = (l == 4) ? Lpp : ca;
            ca = (l == 5) ? Lpp * p1 : ca;
            ca = (l == 6) ? Lpp * p2 : ca;
            ca = (l == 7) ? Lpp * p1 * p1 : ca;
            ca = (l == 8) ? Lpp * p1 * p2 : ca;
            ca = (l == 9) ? Lpp * p2 * p2 : ca;
            accA += ca;
        }
    }
    // lane -> accumulator slot of PP_NACC
    {
        const int slot = (l < 4) ? l : (l == 4 ? 6 : (l == 5 ? 7 : (l == 6 ? 8 : (l == 7 ? 11 : (l == 8 ? 12 : 15)))));
        if (l >= 10) { /* no accumulator */ }
        else red[g * PP_NACC + slot] = accA;
        // zero the scattering slots once per group
        if (l >= 10) {
            const int z[6] = {4, 5, 9, 10, 13, 14};
            red[g * PP_NACC + z[l - 10]] = 0.0;
        }
        if (l < 5) red[g * PP_NACC + 16 + l] = 0.0;
    }
    __syncthreads();
    if (tid < PP_NACC) {
        double s = 0.0;
        for (int gg = 0; gg < 256 / LPC; ++gg) s += red[gg * PP_NACC + tid];
        a.partial[((size_t)i * a.nchunk + chunk) * PP_NACC + tid] = s;
    }
}

// --------------------------------------------------------------------------
// small dense linear algebra on the fit subspace (n <= 5), one thread
// --------------------------------------------------------------------------
template <int n>
__device__ inline bool chol_solve(const double* A, const double* b, double* x) {
    double Lm[25];
    #pragma unroll
    for (int i = 0; i < n; ++i)
        #pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i * n + j];
            #pragma unroll
            for (int k = 0; k < j; ++k) s -= Lm[i * n + k] * Lm[j * n + k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                Lm[i * n + i] = sqrt(s);
            } else Lm[i * n + j] = s / Lm[j * n + j];
        }
    double y[5];
    #pragma unroll
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        #pragma unroll
        for (int k = 0; k < i; ++k) s -= Lm[i * n + k] * y[k];
        y[i] = s / Lm[i * n + i];
    }
    #pragma unroll
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        #pragma unroll
        for (int k = i + 1; k < n; ++k) s -= Lm[k * n + i] * x[k];
        x[i] = s / Lm[i * n + i];
    }
    return true;
}

// in-place Gauss-Jordan inverse with partial pivoting; false if singular
template <int n>
__device__ inline bool mat_inverse(double* A) {
    double inv[25];
    #pragma unroll
    for (int i = 0; i < n; ++i)
        #pragma unroll
        for (int j = 0; j < n; ++j) inv[i * n + j] = (i == j) ? 1.0 : 0.0;
    #pragma unroll
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(A[c * n + c]);
        #pragma unroll
        for (int r = c + 1; r < n; ++r)
            if (fabs(A[r * n + c]) > best) { best = fabs(A[r * n + c]); piv = r; }
        if (!(best > 0.0) || !isfinite(best)) return false;
        // (rows c and piv change places; written over the candidate rows so that every index is a constant)
        #pragma unroll
        for (int r = c + 1; r < n; ++r)
            if (piv == r) {
                #pragma unroll
                for (int j = 0; j < n; ++j) {
                    double t = A[c * n + j]; A[c * n + j] = A[r * n + j]; A[r * n + j] = t;
                    t = inv[c * n + j]; inv[c * n + j] = inv[r * n + j]; inv[r * n + j] = t;
                }
            }
        const double d = 1.0 / A[c * n + c];
        #pragma unroll
        for (int j = 0; j < n; ++j) { A[c * n + j] *= d; inv[c * n + j] *= d; }
        #pragma unroll
        for (int r = 0; r < n; ++r)
            if (r != c) {
                const double f = A[r * n + c];
                if (f != 0.0)
                    #pragma unroll
                    for (int j = 0; j < n; ++j) { A[r * n + j] -= f * A[c * n + j]; inv[r * n + j] -= f * inv[c * n + j]; }
            }
    }
    #pragma unroll
    for (int j = 0; j < n * n; ++j) A[j] = inv[j];
    return true;
}

template <int n>
__device__ inline double vdot(const double* a, const double* b) {
    double s = 0.0;
    #pragma unroll
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

// trust-region subproblem: Newton step if H is positive definite and the step
// is inside the region, else Steihaug conjugate gradients to the boundary
// (scipy/optimize/_trustregion_ncg.py is what the reference drives).
template <int n>
__device__ inline void tr_subproblem(const double* g, const double* H, double radius, double* p,
                                     int* hits) {
    *hits = 0;
    double mg[5];
    #pragma unroll
    for (int i = 0; i < n; ++i) mg[i] = -g[i];
    if (chol_solve<n>(H, mg, p)) {
        if (sqrt(vdot<n>(p, p)) < radius) return;
    }
    double z[5] = {0, 0, 0, 0, 0}, r[5], d[5], Bd[5];
    #pragma unroll
    for (int i = 0; i < n; ++i) { r[i] = g[i]; d[i] = -g[i]; }
    const double gnorm = sqrt(vdot<n>(g, g));
    const double tol = 1e-14 * gnorm;
    for (int it = 0; it < 4 * n + 4; ++it) {
        #pragma unroll
        for (int i = 0; i < n; ++i) Bd[i] = vdot<n>(H + i * n, d);
        const double dBd = vdot<n>(d, Bd);
        const double dd = vdot<n>(d, d), zd = vdot<n>(z, d), zz = vdot<n>(z, z);
        // intersections of z + t d with the boundary
        const double disc = sqrt(fmax(zd * zd - dd * (zz - radius * radius), 0.0));
        const double ta = (-zd - disc) / dd, tb = (-zd + disc) / dd;
        if (dBd <= 0.0) {
            double pa[5], pb[5], Hp[5];
            #pragma unroll
            for (int i = 0; i < n; ++i) { pa[i] = z[i] + ta * d[i]; pb[i] = z[i] + tb * d[i]; }
            #pragma unroll
            for (int i = 0; i < n; ++i) Hp[i] = vdot<n>(H + i * n, pa);
            const double ma = vdot<n>(g, pa) + 0.5 * vdot<n>(pa, Hp);
            #pragma unroll
            for (int i = 0; i < n; ++i) Hp[i] = vdot<n>(H + i * n, pb);
            const double mb = vdot<n>(g, pb) + 0.5 * vdot<n>(pb, Hp);
            #pragma unroll
            for (int i = 0; i < n; ++i) p[i] = (ma < mb) ? pa[i] : pb[i];
            *hits = 1;
            return;
        }
        const double rr = vdot<n>(r, r), al = rr / dBd;
        double zn[5];
        #pragma unroll
        for (int i = 0; i < n; ++i) zn[i] = z[i] + al * d[i];
        if (sqrt(vdot<n>(zn, zn)) >= radius) {
            #pragma unroll
            for (int i = 0; i < n; ++i) p[i] = z[i] + tb * d[i];
            *hits = 1;
            return;
        }
        double rn2 = 0.0;
        #pragma unroll
        for (int i = 0; i < n; ++i) { r[i] += al * Bd[i]; rn2 += r[i] * r[i]; }
        #pragma unroll
        for (int i = 0; i < n; ++i) z[i] = zn[i];
        if (sqrt(rn2) <= tol) break;
        const double be = rn2 / rr;
        #pragma unroll
        for (int i = 0; i < n; ++i) d[i] = -r[i] + be * d[i];
    }
    #pragma unroll
    for (int i = 0; i < n; ++i) p[i] = z[i];
}

// --------------------------------------------------------------------------
// SciPy's trust-ncg, operation by operation (scipy/optimize/_trustregion_ncg.py
// CGSteihaugSubproblem.solve and _trustregion.py _minimize_trust_region): what
// the reference drives with gtol = -1 (pptoaslib.py:1001-1014).  Its exit --
// "predicted reduction <= 0" in floating point -- and its TRUNCATED conjugate-
// gradient steps (residual tolerance min(1/2, sqrt|g|) |g|) decide where the
// reference's answer lands (up to ~1.5e-9 rot short of the optimum for GM and
// scattering fits); following the same iterates reproduces that answer, not
// merely the optimum.  Unfitted parameters have zero gradient / Hessian rows in
// the reference, so working on the fit subspace gives the same numbers.
// --------------------------------------------------------------------------
template <int n>
__device__ inline double tr_model_value(double f, const double* g, const double* H, const double* p) {
    double Hp[5];
    #pragma unroll
    for (int i = 0; i < n; ++i) Hp[i] = vdot<n>(H + i * n, p);
    return f + vdot<n>(g, p) + 0.5 * vdot<n>(p, Hp);
}

template <int n>
__device__ inline void tr_boundaries(const double* z, const double* d, double radius, double* ta,
                                     double* tb) {
    const double a = vdot<n>(d, d), b = 2.0 * vdot<n>(z, d), c = vdot<n>(z, z) - radius * radius;
    const double sq = sqrt(b * b - 4.0 * a * c);
    const double aux = b + copysign(sq, b);
    const double t1 = -aux / (2.0 * a), t2 = -2.0 * c / aux;
    *ta = fmin(t1, t2); *tb = fmax(t1, t2);
}

template <int n>
__device__ inline void tr_cg_steihaug_scipy(double f, const double* g, const double* H, double radius,
                                            double* p, int* hits) {
    *hits = 0;
    #pragma unroll
    for (int i = 0; i < n; ++i) p[i] = 0.0;
    const double gmag = sqrt(vdot<n>(g, g));
    const double tol = fmin(0.5, sqrt(gmag)) * gmag;
    if (gmag < tol) return;
    double z[5] = {0, 0, 0, 0, 0}, r[5], d[5], Bd[5];
    #pragma unroll
    for (int i = 0; i < n; ++i) { r[i] = g[i]; d[i] = -g[i]; }
    for (int it = 0; it < 64; ++it) {
        #pragma unroll
        for (int i = 0; i < n; ++i) Bd[i] = vdot<n>(H + i * n, d);
        const double dBd = vdot<n>(d, Bd);
        if (dBd <= 0.0) {
            double ta, tb, pa[5], pb[5];
            tr_boundaries<n>(z, d, radius, &ta, &tb);
            #pragma unroll
            for (int i = 0; i < n; ++i) { pa[i] = z[i] + ta * d[i]; pb[i] = z[i] + tb * d[i]; }
            const bool first = tr_model_value<n>(f, g, H, pa) < tr_model_value<n>(f, g, H, pb);
            #pragma unroll
            for (int i = 0; i < n; ++i) p[i] = first ? pa[i] : pb[i];
            *hits = 1;
            return;
        }
        const double rsq = vdot<n>(r, r), alpha = rsq / dBd;
        double zn[5];
        #pragma unroll
        for (int i = 0; i < n; ++i) zn[i] = z[i] + alpha * d[i];
        if (sqrt(vdot<n>(zn, zn)) >= radius) {
            double ta, tb;
            tr_boundaries<n>(z, d, radius, &ta, &tb);
            #pragma unroll
            for (int i = 0; i < n; ++i) p[i] = z[i] + tb * d[i];
            *hits = 1;
            return;
        }
        double rn[5];
        #pragma unroll
        for (int i = 0; i < n; ++i) rn[i] = r[i] + alpha * Bd[i];
        const double rnsq = vdot<n>(rn, rn);
        if (sqrt(rnsq) < tol || !(rnsq == rnsq)) {
            #pragma unroll
            for (int i = 0; i < n; ++i) p[i] = zn[i];
            return;
        }
        const double beta = rnsq / rsq;
        #pragma unroll
        for (int i = 0; i < n; ++i) { d[i] = -rn[i] + beta * d[i]; z[i] = zn[i]; r[i] = rn[i]; }
    }
    #pragma unroll
    for (int i = 0; i < n; ++i) p[i] = z[i];
}

// the same, for a subspace dimension only known at run time
#define PP_FOR_N(n_, EXPR)            \
    switch (n_) {                     \
        case 1: { constexpr int N_ = 1; EXPR; } break; \
        case 2: { constexpr int N_ = 2; EXPR; } break; \
        case 3: { constexpr int N_ = 3; EXPR; } break; \
        case 4: { constexpr int N_ = 4; EXPR; } break; \
        default: { constexpr int N_ = 5; EXPR; } break; \
    }
__device__ inline bool chol_solve(int n, const double* A, const double* b, double* x) {
    bool ok = false;
    PP_FOR_N(n, ok = chol_solve<N_>(A, b, x));
    return ok;
}
__device__ inline bool mat_inverse(int n, double* A) {
    bool ok = false;
    PP_FOR_N(n, ok = mat_inverse<N_>(A));
    return ok;
}
__device__ inline double vdot(int n, const double* a, const double* b) {
    double v = 0.0;
    PP_FOR_N(n, v = vdot<N_>(a, b));
    return v;
}
__device__ inline double tr_model_value(int n, double f, const double* g, const double* H, const double* p) {
    double v = 0.0;
    PP_FOR_N(n, v = tr_model_value<N_>(f, g, H, p));
    return v;
}
__device__ inline void tr_cg_steihaug_scipy(int n, double f, const double* g, const double* H, double radius,
                                            double* p, int* hits) {
    PP_FOR_N(n, tr_cg_steihaug_scipy<N_>(f, g, H, radius, p, hits));
}

// one decision of SciPy's loop after the proposal x + p was evaluated (f_new):
// radius update and acceptance (eta = 0.15, max radius 1000)
__device__ inline bool tr_scipy_accept(double f, double f_new, double pred, int hits, bool finite,
                                       double* radius) {
    const double rho = finite ? (f - f_new) / pred : -1.0;
    if (rho < 0.25) *radius *= 0.25;
    else if (rho > 0.75 && hits) *radius = fmin(2.0 * *radius, 1000.0);
    return rho > 0.15;
}

// The (phi, DM, GM) solvers' proposals on the fit subspace WITHOUT a dynamically indexed array (those
// live in scratch memory: every access of the serial walk a round trip): ix[k] = position of the k-th
// fitted parameter, N = their number (compile time, PP_FOR_N3); the proposal comes back scattered to
// the three positions (0 where not fitted).  Same operations in the same order as the pointer versions.
#define PP_FOR_N3(n_, EXPR)           \
    switch (n_) {                     \
        case 1: { constexpr int N_ = 1; EXPR; } break; \
        case 2: { constexpr int N_ = 2; EXPR; } break; \
        default: { constexpr int N_ = 3; EXPR; } break; \
    }
// (values, not addresses: given an array the compiler turns the selection into an indexed load from a scratch copy)
__device__ __forceinline__ double sel3(double v0, double v1, double v2, int k) { return k == 0 ? v0 : (k == 1 ? v1 : v2); }
template <int N>
__device__ __forceinline__ void gather_sub3(const double (&g)[3], const double (&H)[9], const int (&ix)[3], double (&gs)[N],
                                            double (&Hs)[N * N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const int k = ix[r];
        gs[r] = sel3(g[0], g[1], g[2], k);
        const double r0 = sel3(H[0], H[3], H[6], k), r1 = sel3(H[1], H[4], H[7], k), r2 = sel3(H[2], H[5], H[8], k);   // row ix[r] of H
#pragma unroll
        for (int c = 0; c < N; ++c) Hs[r * N + c] = sel3(r0, r1, r2, ix[c]);
    }
}
template <int N>
__device__ __forceinline__ void scatter_sub3(const double (&p)[N], const int (&ix)[3], double (&p3)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        p3[j] = 0.0;
#pragma unroll
        for (int r = 0; r < N; ++r) if (ix[r] == j) p3[j] = p[r];
    }
}
template <int N>
__device__ __forceinline__ void tr_propose_scipy3(double f, const double (&g)[3], const double (&H)[9], const int (&ix)[3],
                                                  double radius, double (&p3)[3], int& hits, double& pred) {
    double gs[N], Hs[N * N], p[N];
    gather_sub3<N>(g, H, ix, gs, Hs);
    tr_cg_steihaug_scipy<N>(f, gs, Hs, radius, p, &hits);
    pred = f - tr_model_value<N>(f, gs, Hs, p);
    scatter_sub3<N>(p, ix, p3);
}
template <int N>
__device__ __forceinline__ bool newton_propose3(const double (&g)[3], const double (&H)[9], const int (&ix)[3], double (&p3)[3],
                                                double& pred) {
    double gs[N], Hs[N * N], p[N], mg[N], Hp[N];
    gather_sub3<N>(g, H, ix, gs, Hs);
#pragma unroll
    for (int r = 0; r < N; ++r) mg[r] = -gs[r];
    if (!chol_solve<N>(Hs, mg, p)) return false;
#pragma unroll
    for (int r = 0; r < N; ++r) Hp[r] = vdot<N>(Hs + r * N, p);
    pred = -(vdot<N>(gs, p) + 0.5 * vdot<N>(p, Hp));
    scatter_sub3<N>(p, ix, p3);
    return true;
}

// --------------------------------------------------------------------------
// One pass over X that makes further passes unnecessary (no scattering): the
// per-channel cross-correlation C_n(phi_n + d) = Re sum_k X_nk e^{2 pi i k (phi_n + d)}
// is an entire function of d, so its derivatives at the initial point
//   A_j = Re sum_k (2 pi i k)^j X_nk e^{2 pi i k phi_n},  j = 0..PP_TJ
// give C_n, C_n', C_n'' anywhere nearby, with the rigorous remainder
//   |R| <= Bn |d|^(PP_TJ+1-m) / (PP_TJ+1-m)!,  Bn = sum_k (|Re X|+|Im X|) (2 pi k)^(PP_TJ+1)
// for the m-th derivative.  k_taylor_solve then runs the whole Newton solve on
// these 12 numbers per channel and certifies the truncation error; subints that
// move too far for the certificate fall back to evaluations over X.
// Same work layout as k_eval_fast.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_eval_moments(FitArgs a) {
    constexpr int LPC = 16;
    const int jx = blockIdx.x, i = sub_of(a.act, jx), chunk = blockIdx.y;
    const SubState& st = a.st[i];
    const int tid = threadIdx.x, g = tid / LPC, l = tid % LPC;
    const double phi = st.xe[0], DM = st.xe[1], GM = st.xe[2];
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const int* ktv = a.ktab ? as_global(a.ktab[a.slot ? a.slot[i] : 0]) : nullptr;
    double* tay = a.tay;
    const size_t row0 = (size_t)i * a.nchan;
    const int n0 = chunk * a.cpc, n1 = min(n0 + a.cpc, a.nchan_x);
    const int src = ((tid & 63) & ~(LPC - 1)) | (LPC - 1);
    for (int nn = n0 + g; nn < n1; nn += 256 / LPC) {
        const int n = a.coff + nn * a.cstep;
        const double w = wts[n];
        double p1, p2;
        phase_geom(freqs[n], P, nuDM, nuGM, p1, p2);
        const double phin = phi + DM * p1 + GM * p2;
        cplx e0 = unit_phasor((double)(l + 1), phin);
        const cplx w1 = make_double2(__shfl(e0.x, src, 64), __shfl(e0.y, src, 64));
        const cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
        const cplx* xrow = a.X + ((size_t)jx * a.nchan_x + nn) * a.Xs;
        const int ktn = ktv ? ktv[n] : a.Kt;
        double s[PP_TSTRIDE];
#pragma unroll
        for (int j = 0; j < PP_TSTRIDE; ++j) s[j] = 0.0;
        double k = (double)(l + 1);
        if (w != 0.0) {
#pragma unroll 1
            for (int jj = l; jj < ktn; jj += 4 * LPC) {
                // four independent 16-byte loads in flight per lane
                const cplx zero = make_double2(0.0, 0.0);
                cplx xv[4];
                xv[0] = xrow[jj];
                xv[1] = (jj + LPC < ktn) ? xrow[jj + LPC] : zero;
                xv[2] = (jj + 2 * LPC < ktn) ? xrow[jj + 2 * LPC] : zero;
                xv[3] = (jj + 3 * LPC < ktn) ? xrow[jj + 3 * LPC] : zero;
                cplx ev[4];
                ev[0] = e0; ev[1] = cmul(e0, w1); ev[2] = cmul(e0, w2); ev[3] = cmul(e0, w3);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const cplx z = cmul(xv[q], ev[q]);
                    const double kap = PP_TWO_PI * (k + (double)(q * LPC));
                    const double kap2 = kap * kap;
                    double ur = z.x, ui = z.y * kap;     // even / odd orders advance by kap^2
#pragma unroll
                    for (int j = 0; j <= PP_TJ; j += 2) {
                        s[j] += ur;
                        ur *= kap2;
                        if (j + 1 <= PP_TJ) { s[j + 1] += ui; ui *= kap2; }
                    }
                    // remainder coefficient: (2 pi k)^(PP_TJ+1) (|Re X| + |Im X|)
                    static_assert(PP_TJ % 2 == 0, "PP_TJ must be even");
                    double pw = kap;
#pragma unroll
                    for (int j = 0; j < PP_TJ / 2; ++j) pw *= kap2;
                    s[PP_TJ + 1] = fma(pw, fabs(xv[q].x) + fabs(xv[q].y), s[PP_TJ + 1]);
                }
                e0 = cmul(e0, w4);
                k += 4.0 * LPC;
            }
        }
        // Re(i^j z): +Re, -Im, -Re, +Im, ...
#pragma unroll
        for (int j = 0; j < PP_TSTRIDE; ++j) {
            double v = group_sum<LPC>(s[j]);
            if (j <= PP_TJ && ((j & 3) == 1 || (j & 3) == 2)) v = -v;
            if (l == (j % LPC)) tay[tay_idx(row0 + n, j)] = v;
        }
    }
}

// Horner evaluation of the shifted sums A0', A1', A2' from the Taylor model
// (the channel's 12 doubles are fetched as six 16-byte loads: rows are 96 B apart)
__device__ __forceinline__ void taylor_load(const double* tay, size_t row, double (&t)[PP_TSTRIDE]) {
    static_assert(PP_TSTRIDE % 2 == 0, "rows of the Taylor model are read in pairs");
    const double* tg = tay + tay_idx(row, 0);
#pragma unroll
    for (int j = 0; j < PP_TSTRIDE / 2; ++j) {
        const double2 v = *reinterpret_cast<const double2*>(tg + j * PP_TAY_PAIR_STRIDE);
        t[2 * j] = v.x; t[2 * j + 1] = v.y;
    }
}
__device__ __forceinline__ void taylor_shift_reg(const double (&t)[PP_TSTRIDE], double d, double& A0, double& A1,
                                                 double& A2);
__device__ __forceinline__ void taylor_shift_reg(const double (&t)[PP_TSTRIDE], double d, double& A0, double& A1,
                                                 double& A2) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int j = PP_TJ; j >= 0; --j) {
        a0 = fma(a0, d * (1.0 / (double)(j + 1)), t[j]);
        if (j >= 1) a1 = fma(a1, d * (1.0 / (double)j), t[j]);
        if (j >= 2) a2 = fma(a2, d * (1.0 / (double)(j - 1)), t[j]);
    }
    // a0 = sum_j t[j] d^j/j!  (Horner with the 1/(j+1) factors);  a1, a2 the same
    // series started at t[1], t[2]
    A0 = a0; A1 = a1; A2 = a2;
}

// Solve on the Taylor model, one block of NT threads per subint.  method 0 walks
// SciPy's trust-ncg iteration on the model (every evaluation is O(nchan), no pass
// over the data), method 1 is plain Newton to the rounding of f.
// NT: the walk is a chain of ~20 serial decisions between evaluations of ~C/NT channels per thread --
// a narrow band is solved by ONE wave per subint (no barrier, four times the subints in flight per CU),
// a wide one by up to eight (8 channels per thread; the CU's eight waves then belong to ONE subint, and the
// rows the subints in flight re-read on every evaluation -- 393 KB each at 4096 channels -- stay inside the
// Infinity Cache).  (Rows held in registers over the evaluations: measured slower in round 3, profiles/README.md.)
#define PP_SOLVE_CACHE_MAX 4096   // channels whose invariants k_taylor_solve can keep in LDS (32 B each: 512 per wave of the block)
#ifndef PP_SOLVE_PF
#define PP_SOLVE_PF 2         // Taylor rows a thread keeps on their way (24 registers each)
#endif
#ifndef PP_TAYLOR_WAVES
#define PP_TAYLOR_WAVES 2     // waves per SIMD the kernel is compiled for (register cap 512 / n)
#endif
// The body serves two kernels.  NVW = 1: NT real threads, one subint per workgroup (k_taylor_solve).  NVW = NT / 64:
// ONE real wave walks the NT / 64 waves of that kernel in turn -- lane l plays threads l, l + 64, ... -- and forms
// every block-wide sum from the same per-wave totals in the same order (vblock_sum): bitwise the results of the
// NT-thread kernel from a workgroup of 64 threads (k_taylor_solve_v; what a transform wave could run between rows).
// `tid`: the real thread (0 .. NT / NVW - 1).  scratch: PP_BSUM_DOUBLES(NT / 64, 10) doubles of LDS; inv_lds:
// 4 a.solve_cache doubles of LDS.
template <int NV, int NVMAX, int NT, int NVW, typename F>
__device__ __forceinline__ void vblock_sum(double (&out)[NV], double* scratch, int& flip, double* mx, int tid, F&& accumulate) {
    static_assert(NV <= NVMAX, "scratch sized for NVMAX values");
    constexpr int NWV = NT / 64, RT = NT / NVW;      // virtual waves, real threads
    const int lane = tid & 63, rw = tid >> 6;
    double* buf = scratch + flip * NWV * (NVMAX + 1);
    flip ^= 1;
    // (`out` is the accumulator of the wave in hand, as block_sum_t's argument was: no second set of NV registers)
    auto one_wave = [&](const int vw) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) out[j] = 0.0;
        double m = 0.0;
        accumulate(lane + 64 * vw, out, m);
        wave_totals_to<NV, 0>(out, lane, buf + vw * (NV + 1));
        if (mx) {
            m = group_max<64>(m);
            if (lane == 0) buf[vw * (NV + 1) + NV] = m;
        }
    };
    if constexpr (NVW == 1) one_wave(rw);
    else {
#pragma unroll 1
        for (int k = 0; k < NVW; ++k) one_wave(rw + (RT / 64) * k);   // the virtual wave this real wave plays now
    }
    __syncthreads();
    // (the waves' totals are added one wave at a time, as block_sum_t's run-time loop does: with the wave loop
    // unrolled too, 8 x 31 LDS reads are in flight at once and the post-fit kernels spill hundreds of registers)
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = 0.0;
#pragma unroll 1
    for (int w = 0; w < NWV; ++w) {
#pragma unroll
        for (int i = 0; i < NV; ++i) out[i] += buf[w * (NV + 1) + i];
    }
    if (mx) {
        double m = buf[NV];
#pragma unroll 1
        for (int w = 1; w < NWV; ++w) m = fmax(m, buf[w * (NV + 1) + NV]);
        *mx = m;
    }
}

template <int NT, int PF, int NVW>
__device__ __forceinline__ void taylor_solve_body(const FitArgs& a, const int i, const int tid, double* scratch, double* inv_lds) {
    constexpr int RT = NT / NVW;         // real threads
    static_assert(NT % 64 == 0 && RT % 64 == 0 && NT % RT == 0, "whole waves");
    SubState& st = a.st[i];
    if (st.done) return;                 // (second launch, after a re-expansion of the others)
    int flip = 0;
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const double* msum = as_global(a.msum[a.slot ? a.slot[i] : 0]);
    const double* tay = a.tay;
    const size_t row0 = (size_t)i * a.nchan;
    const int* fl = a.flags;
    int ix[3] = {0, 0, 0}, nf = 0;       // positions of the fitted parameters (statically indexed: no scratch)
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (fl[j]) {
            if (nf == 0) ix[0] = j; else if (nf == 1) ix[1] = j; else ix[2] = j;
            ++nf;
        }
    // what an evaluation needs of a channel beside its Taylor row -- weight, phase geometry (five
    // divisions), template power -- is the same in every evaluation of the solve: formed once, kept in
    // LDS (dynamic: 32 B x a.solve_cache channels; channels beyond, if any, are formed again on every evaluation)
    const int ncache = a.solve_cache;
    double *inv_w = inv_lds, *inv_p1 = inv_lds + ncache, *inv_p2 = inv_lds + 2 * ncache, *inv_S = inv_lds + 3 * ncache;
    {
        const int nc = min(a.nchan, ncache);
#pragma unroll 4
        for (int n = tid; n < nc; n += RT) {
            double p1, p2;
            phase_geom(freqs[n], P, nuDM, nuGM, p1, p2);
            inv_w[n] = wts[n]; inv_p1[n] = p1; inv_p2[n] = p2; inv_S[n] = msum[n];
        }
        __syncthreads();
    }
    auto chan_inv = [&](int n, double& w, double& p1, double& p2, double& S0) {
        if (n < ncache) { w = inv_w[n]; p1 = inv_p1[n]; p2 = inv_p2[n]; S0 = inv_S[n]; }
        else { w = wts[n]; phase_geom(freqs[n], P, nuDM, nuGM, p1, p2); S0 = msum[n]; }
    };
    // the channels of (virtual) thread vt: body(n, row of the Taylor model)
    auto for_channels = [&](const int vt, auto&& body) {
        // (the next row is on its way while this one is worked on: the loop is a chain of
        // memory latencies otherwise, two waves per SIMD hide none of it)
        // (PF rows ahead.  Narrow bands -- the subints' rows together fit the Infinity Cache, the evaluation is a chain of
        // latencies: two ahead.  4096 channels -- 403 MB of rows per 1024 subints, re-read by every evaluation: the more
        // a CU keeps in flight, the more of the others' rows it pushes out; PF = 0, each row fetched when its turn
        // comes, measured fastest there: 0.45 ms against 0.47 / 0.48 / 0.485 for 1 / 2 / 3 ahead)
        if constexpr (PF == 0) {
            for (int n = vt; n < a.nchan; n += NT) {
                double t[PP_TSTRIDE];
                taylor_load(tay, row0 + n, t);
                body(n, t);
            }
        } else {
            double buf[PF ? PF : 1][PP_TSTRIDE];
#pragma unroll
            for (int d = 0; d < PF; ++d)
                if (vt + d * NT < a.nchan) taylor_load(tay, row0 + vt + d * NT, buf[d]);
            for (int n0 = vt; n0 < a.nchan; n0 += PF * NT) {
#pragma unroll
                for (int d = 0; d < PF; ++d) {
                    const int n = n0 + d * NT;
                    if (n < a.nchan) {
                        double t[PP_TSTRIDE];
#pragma unroll
                        for (int j = 0; j < PP_TSTRIDE; ++j) t[j] = buf[d][j];
                        if (n + PF * NT < a.nchan) taylor_load(tay, row0 + n + PF * NT, buf[d]);
                        body(n, t);
                    }
                }
            }
        }
    };
    // f, g, H of the model at displacement dx from x0 (identical in every thread);
    // returns the largest per-channel phase displacement
#ifdef PP_SOLVE_CLOCKS
    long long ck_loop = 0, ck_red = 0, ck_start = clock64(), ck_n = 0;
#endif
    // ... and at the expansion point itself (the first evaluation of the ordinary flow: d = 0 in every channel,
    // where Horner's rule returns A0, A1, A2 = t[0], t[1], t[2] exactly) only the first two coefficient pairs
    // of a row are read: 32 of its 96 bytes (which saves HBM traffic with the blocked row layout only)
    auto for_channels_at_origin = [&](const int vt, auto&& body) {
        constexpr int U = 4;
        for (int n0 = vt; n0 < a.nchan; n0 += U * NT) {
            double2 h[U][2];
#pragma unroll
            for (int d = 0; d < U; ++d)
                if (n0 + d * NT < a.nchan) {
                    const double* tg = tay + tay_idx(row0 + n0 + d * NT, 0);
                    h[d][0] = *reinterpret_cast<const double2*>(tg);
                    h[d][1] = *reinterpret_cast<const double2*>(tg + PP_TAY_PAIR_STRIDE);
                }
#pragma unroll
            for (int d = 0; d < U; ++d)
                if (n0 + d * NT < a.nchan) body(n0 + d * NT, h[d][0].x, h[d][0].y, h[d][1].x);
        }
    };
    auto evalm = [&](const double* dx, double& f, double* g, double* H, bool at_origin = false) -> double {
#ifdef PP_SOLVE_CLOCKS
        const long long ck0 = clock64();
#endif
        double acc[10];
        double dmax = 0.0;
        // (a channel out of the fit adds zeros instead of branching around the work: the two channels a thread
        // has in hand then interleave -- with one wave per SIMD a dependent chain costs its full latency)
        vblock_sum<10, 10, NT, NVW>(acc, scratch, flip, &dmax, tid, [&](const int vt, double (&ac)[10], double& dm) __attribute__((always_inline)) {
            auto add = [&](double w, double p1, double p2, double S0, double A0, double A1, double A2) {
                const bool in = (w != 0.0);
                const double r = A0 / S0;
                const double F = in ? -w * A0 * r : 0.0, Gp = in ? -2.0 * w * r * A1 : 0.0;
                const double Lpp = in ? -2.0 * w * (A1 * A1 / S0 + r * A2) : 0.0;
                ac[0] += F;
                ac[1] += Gp; ac[2] += Gp * p1; ac[3] += Gp * p2;
                ac[4] += Lpp; ac[5] += Lpp * p1; ac[6] += Lpp * p2;
                ac[7] += Lpp * p1 * p1; ac[8] += Lpp * p1 * p2; ac[9] += Lpp * p2 * p2;
            };
            if (at_origin)
                for_channels_at_origin(vt, [&](int n, double A0, double A1, double A2) {
                    double w, p1, p2, S0;
                    chan_inv(n, w, p1, p2, S0);
                    add(w, p1, p2, S0, A0, A1, A2);
                });
            else
                for_channels(vt, [&](int n, const double (&t)[PP_TSTRIDE]) {
                    double w, p1, p2, S0;
                    chan_inv(n, w, p1, p2, S0);
                    const double d = dx[0] + dx[1] * p1 + dx[2] * p2;
                    dm = fmax(dm, (w != 0.0) ? fabs(d) : 0.0);
                    double A0, A1, A2;
                    taylor_shift_reg(t, d, A0, A1, A2);
                    add(w, p1, p2, S0, A0, A1, A2);
                });
        });
#ifdef PP_SOLVE_CLOCKS
        const long long ck1 = clock64();
#endif
#ifdef PP_SOLVE_CLOCKS
        ck_loop += ck1 - ck0; ck_red += clock64() - ck1; ++ck_n;
#endif
        f = acc[0];
        // (phi, DM, GM) block only: 3 + 9 numbers per thread instead of 5 + 25
        g[0] = fl[0] ? acc[1] : 0.0; g[1] = fl[1] ? acc[2] : 0.0; g[2] = fl[2] ? acc[3] : 0.0;
        const double hh[3][3] = {{acc[4], acc[5], acc[6]}, {acc[5], acc[7], acc[8]}, {acc[6], acc[8], acc[9]}};
#pragma unroll
        for (int r_ = 0; r_ < 3; ++r_)
#pragma unroll
            for (int c_ = 0; c_ < 3; ++c_) H[r_ * 3 + c_] = (fl[r_] && fl[c_]) ? hh[r_][c_] : 0.0;
        return dmax;
    };
    double dx[3] = {0.0, 0.0, 0.0};      // accepted displacement from x0 in (phi, DM, GM)
    // (the reference-seed flow starts the iteration off centre: at the reference's own guess)
    const bool off_centre = (a.xstart != nullptr) && st.recentred == 0;
    if (off_centre)
#pragma unroll
        for (int j = 0; j < 3; ++j) dx[j] = fl[j] ? a.xstart[i * 5 + j] - st.xe[j] : 0.0;
    const double dx0[3] = {dx[0], dx[1], dx[2]};
    double f, g[3], H[9];
    bool ok = true;
    int it = 0, nfev = 1;                // objective evaluations as SciPy's nfev counts them
    double dpath = evalm(dx, f, g, H, !off_centre);   // (0 at the expansion point)
    if (tid == 0 && st.recentred == 0) {     // (objective hooks: at init_params only)
        st.f0 = f;
#pragma unroll
        for (int j = 0; j < 5; ++j) st.g0[j] = j < 3 ? g[j] : 0.0;
#pragma unroll
        for (int r_ = 0; r_ < 5; ++r_)
#pragma unroll
            for (int c_ = 0; c_ < 5; ++c_) st.H0[r_ * 5 + c_] = (r_ < 3 && c_ < 3) ? H[r_ * 3 + c_] : 0.0;
    }
    if (!isfinite(f)) ok = false;
    if (ok && a.method == 0) {
        double radius = 1.0;
        // SciPy's ScalarFunction keeps the point it evaluated last: a proposal that IS that
        // point (a rejected step proposed again under a smaller radius, ~15 times in a row
        // at the end of the iteration) is answered from the cache and not counted in nfev.
        // SciPy iterates on ABSOLUTE parameters, x <- fl(x + p), and compares the proposal, bit for
        // bit, with the point its ScalarFunction evaluated last.  The walk keeps that absolute iterate
        // (xa) beside the displacement from the expansion point (dx) the model is evaluated at, so the
        // cache decision -- including whether the closing proposal p = -H^-1 g ~ 1e-15 still changes
        // x and therefore counts as an evaluation (pptoaslib.py:1017 `nfeval = results.nfev`) -- is
        // taken on the reference's own numbers.  (The model itself is evaluated at the unrounded
        // displacement: half an ulp of DM is worth ~1e-15 rot, the same size as the rounding of the
        // reference's own phases at these DMs.)
        double xa[3], xla[3], xld[3] = {dx0[0], dx0[1], dx0[2]};
#pragma unroll
        for (int j = 0; j < 3; ++j) { xa[j] = off_centre ? (fl[j] ? a.xstart[i * 5 + j] : st.xe[j]) : st.xe[j]; xla[j] = xa[j]; }
        double f2 = f, g2[3] = {g[0], g[1], g[2]}, H2[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) H2[j] = H[j];
        for (;;) {
            double p3[3], pred = 0.0;
            int hits = 0;
            PP_FOR_N3(nf, (tr_propose_scipy3<N_>(f, g, H, ix, radius, p3, hits, pred)));
            double xta[3], xt[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { xta[j] = fl[j] ? xa[j] + p3[j] : xa[j]; xt[j] = fl[j] ? dx[j] + p3[j] : dx[j]; }
            // (nfev_shadow = 2: the model is evaluated AT the rounded point fl(x + p), as SciPy evaluates its
            // objective there; xta - x0 is exact, the two are neighbours)
            if (a.nfev_shadow >= 2)
#pragma unroll
                for (int j = 0; j < 3; ++j) xt[j] = xta[j] - st.xe[j];
            // (otherwise the comparison is made on the displacements, which resolve 1e-21 -- the closing
            // proposal is then always a new point and is counted.  Which rule lands on the reference's count
            // more often was measured on 3000 random fits, profiles/r04_parity_sweep.txt: the absolute
            // iterate for one-parameter fits, 81 against 74 %; the displacements for every other family,
            // 94-100 against 82-91 % -- the closing p is each implementation's own rounding noise, so the
            // last unit of nfeval is not reproducible in general)
            const bool shadow = a.nfev_shadow > 0 || (a.nfev_shadow < 0 && nf == 1);
            const bool cached = shadow ? (xta[0] == xla[0] && xta[1] == xla[1] && xta[2] == xla[2])
                                              : (xt[0] == xld[0] && xt[1] == xld[1] && xt[2] == xld[2]);
            if (!(pred > 0.0)) {
                // SciPy's status 2, the reference's normal exit -- taken AFTER it has evaluated
                // the proposal (scipy/optimize/_trustregion.py: m_proposed.fun comes before the
                // test), so that evaluation is in the reference's nfeval
                if (!cached) ++nfev;
                break;
            }
            if (!cached) {
                dpath = fmax(dpath, evalm(xt, f2, g2, H2));
#pragma unroll
                for (int j = 0; j < 3; ++j) { xla[j] = xta[j]; xld[j] = xt[j]; }
                ++nfev;
            }
            bool finite = isfinite(f2);
#pragma unroll
            for (int j = 0; j < 3; ++j) finite = finite && isfinite(g2[j]);
#if defined(PP_TAYLOR_TRACE) && PP_TAYLOR_TRACE >= 3
            if (tid == 0)
                printf("tay it %2d f %.17g f_new %.17g actual %.3e pred %.3e rho %.3f radius %.3e hits %d cached %d\n", it, f, f2,
                       f - f2, pred, (f - f2) / pred, radius, hits, (int)cached);
#endif
            if (tr_scipy_accept(f, f2, pred, hits, finite, &radius)) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { dx[j] = xt[j]; xa[j] = xta[j]; }
                f = f2;
#pragma unroll
                for (int j = 0; j < 3; ++j) g[j] = g2[j];
#pragma unroll
                for (int j = 0; j < 9; ++j) H[j] = H2[j];
            }
            ++it;
            if (it >= a.max_iter || !(radius > 1e-300)) {
#ifdef PP_TAYLOR_TRACE
                if (tid == 0) printf("taylor %d: iteration limit it %d radius %.3e\n", i, it, radius);
#endif
                ok = false; break;
            }
        }
    } else if (ok) {
        double fprev = INFINITY;
        for (it = 0; it < 24; ++it) {
            if (it > 0) { dpath = fmax(dpath, evalm(dx, f, g, H)); ++nfev; }
            if (!isfinite(f)) { ok = false; break; }
            // Newton step on the fit subspace (every thread computes the same)
            double p3[3], pred = 0.0;
            bool solved = false;
            PP_FOR_N3(nf, (solved = newton_propose3<N_>(g, H, ix, p3, pred)));
            if (!solved) { ok = false; break; }
            if (f > fprev + 1e-9 * fabs(fprev)) { ok = false; break; }   // not descending: leave it to the trust region
            fprev = f;
#pragma unroll
            for (int j = 0; j < 3; ++j) if (fl[j]) dx[j] += p3[j];
            // converged to the rounding of f: that was the last step
            if (!(pred > 64.0 * 2.220446049250313e-16 * fabs(f))) break;
        }
        if (it >= 24) ok = false;
        // (g, H belong to the point before the last step: the certificate only
        // needs the curvature scale; the published sums are taken at x0 + dx)
    }
    // ---- certificate (truncation error of the gradient, in parameter units) and the
    // sums of the accepted point x0 + dx for the post-fit stage, in ONE pass over the
    // model (they go to the buffer that only becomes current if the certificate holds)
#ifdef PP_SOLVE_CLOCKS
    const long long ck_cert = clock64();
#endif
    const int buf = 1 - st.cur;
    if (ok) {
        double* csum = a.csum + ((size_t)buf * a.nsub + i) * a.nchan * a.ncs;
        double ev[4];                          // err bounds g_phi, g_DM, g_GM; f at the accepted point
        double jf = 1.0;
        for (int j = 2; j <= PP_TJ; ++j) jf *= (double)j;   // PP_TJ!
        vblock_sum<4, 10, NT, NVW>(ev, scratch, flip, nullptr, tid, [&](const int vt, double (&e4)[4], double&) __attribute__((always_inline)) {
            for_channels(vt, [&](int n, const double (&t)[PP_TSTRIDE]) {
                double w, p1, p2, S0;
                chan_inv(n, w, p1, p2, S0);
                const double d = dx[0] + dx[1] * p1 + dx[2] * p2;
                double A0, A1, A2;
                taylor_shift_reg(t, d, A0, A1, A2);
                csum[(size_t)n * 3] = A0; csum[(size_t)n * 3 + 1] = A1; csum[(size_t)n * 3 + 2] = A2;
                if (w == 0.0) return;
                e4[3] += -w * A0 * A0 / S0;
                // remainder of A1's series (one derivative): Bn |d|^PP_TJ / PP_TJ!
                const double d2 = d * d, d4 = d2 * d2, d10 = d4 * d4 * d2;
                static_assert(PP_TJ == 10, "remainder power written for order 10");
                const double e1 = t[PP_TJ + 1] * d10 / jf;
                const double r = fabs(t[0] / S0) + 1e-300;
                const double ge = 2.0 * w * r * e1 * 1.5;   // + remainder through A0 (smaller by |d|/PP_TJ)
                e4[0] += ge; e4[1] += ge * fabs(p1); e4[2] += ge * fabs(p2);
            });
        });
        // position error <= gradient error / curvature, per fitted parameter
        // (1e-11 pc cm^-3 of DM is worth ~1e-11 rot of phase at the band edge: two
        // decades inside the parity bars)
        const double tol[3] = {1e-13, 1e-11, 1e-8};   // turns, pc cm^-3, GM units
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (fl[j]) {
                const double hjj = fabs(H[j * 3 + j]);
#if defined(PP_TAYLOR_TRACE) && PP_TAYLOR_TRACE >= 2
                if (tid == 0 && j == 0) printf("tmargin %d %.4e %.4e\n", i, ev[j] / (tol[j] * hjj), dpath);
#endif
                if (!(ev[j] <= tol[j] * hjj)) {
#ifdef PP_TAYLOR_TRACE
                    if (tid == 0) printf("taylor %d: certificate %d: err %.3e > tol %.3e (dpath %.3e, it %d) x0 %.9f %.9f dx %.3e %.3e f %.10e\n", i, j, ev[j], tol[j] * hjj, dpath, it, st.xe[0], st.xe[1], dx[0], dx[1], f);
#endif
                    ok = false;
                }
            }
        // every point the iteration visited must lie inside the model's range
        if (!(dpath < 0.02)) {
#ifdef PP_TAYLOR_TRACE
            if (tid == 0) printf("taylor %d: path %.3e left the model's range (it %d)\n", i, dpath, it);
#endif
            ok = false;
        }
        if (ok && tid == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) st.x[j] = st.xe[j] + dx[j];
            st.x[3] = st.xe[3]; st.x[4] = st.xe[4];
            st.f = ev[3];
#pragma unroll
            for (int j = 0; j < 5; ++j) st.g[j] = j < 3 ? g[j] : 0.0;
#pragma unroll
            for (int r_ = 0; r_ < 5; ++r_)
#pragma unroll
                for (int c_ = 0; c_ < 5; ++c_) st.H[r_ * 5 + c_] = (r_ < 3 && c_ < 3) ? H[r_ * 3 + c_] : 0.0;
            st.cur = buf; st.nfev += nfev; st.npass = 1 + st.recentred; st.iter = it; st.status = PP_RC_STALL; st.done = 1; st.fresh = 0;
            atomicSub(a.nactive, 1);
        }
    }
#ifdef PP_SOLVE_CLOCKS
    if (tid == 0 && (i % 200) == 0) {
        const long long e = clock64();
        printf("solve %d: total %lld loop %lld red %lld cert %lld evals %lld -> serial %lld\n", i, e - ck_start, ck_loop, ck_red,
               e - ck_cert, ck_n, (e - ck_start) - ck_loop - ck_red - (e - ck_cert));
    }
#endif
    // Not ok.  The tentative answer x0 + dx is usually still far better than x0 (the
    // harmonics that carry the power are within the model's reach long after the
    // highest ones have left it): expand again about it -- one more pass over this
    // subint's data, at most a.recentre times -- before falling back to evaluations over
    // the cross-spectrum
    // (which start from st.xe, fresh = 1 as k_init_state left it).  Phase / DM fits only:
    // they end at the optimum whatever the path; a GM fit's exit point depends on it.
    if (!ok && tid == 0) {
        // (only from inside the model's nominal range: farther out the tentative answer may
        // belong to another maximum of the correlation altogether)
        bool fin = isfinite(dpath) && dpath < 0.02;
#pragma unroll
        for (int j = 0; j < 3; ++j) fin = fin && isfinite(dx[j]);
        if (off_centre) {
            // reference-seed flow: the model about the pilot's phase did not carry the walk from
            // the reference's guess -- expand again about that guess itself (one more pass over
            // this subint's rows), which is the ordinary one-pass flow from there on: the walk
            // restarts at its own expansion point, whatever the fit family
#pragma unroll
            for (int j = 0; j < 5; ++j) { st.xe[j] = a.xstart[i * 5 + j]; st.x[j] = st.xe[j]; a.x0w[i * 5 + j] = st.xe[j]; }
            st.recentred = 1;
        } else if (st.recentred < a.recentre && fin) {
            st.nfev += nfev;             // (the evaluations of the first expansion stay counted)
#pragma unroll
            for (int j = 0; j < 3; ++j) st.xe[j] += dx[j];
#pragma unroll
            for (int j = 0; j < 5; ++j) { st.x[j] = st.xe[j]; a.x0w[i * 5 + j] = st.xe[j]; }
            st.recentred += 1;
        } else { st.recentred = a.recentre + 1; st.nfev = 0; }   // (no further expansion: the evaluation
                                                                  // loop starts over from st.xe and counts its own)
        st.fresh = 1;
    }
}

template <int NT, int PF = PP_SOLVE_PF>
__global__ __launch_bounds__(NT, PP_TAYLOR_WAVES) void k_taylor_solve(FitArgs a) {
    __shared__ double scratch[PP_BSUM_DOUBLES(NT / 64, 10)];
    extern __shared__ double inv_lds[];
    taylor_solve_body<NT, PF, 1>(a, blockIdx.x, threadIdx.x, scratch, inv_lds);
}
// one real wave per subint walking the NT / 64 waves of k_taylor_solve<NT, PF> in turn: bitwise its results
template <int NT, int PF = PP_SOLVE_PF>
__global__ __launch_bounds__(64, PP_TAYLOR_WAVES) void k_taylor_solve_v(FitArgs a) {
    __shared__ double scratch[PP_BSUM_DOUBLES(NT / 64, 10)];
    extern __shared__ double inv_lds[];
    taylor_solve_body<NT, PF, NT / 64>(a, blockIdx.x, threadIdx.x, scratch, inv_lds);
}

// unpack the 21 accumulators into g[5], H[25] with the fit flags applied
// (pptoaslib.py:573, 629-630)
__device__ inline void unpack_acc(const double* acc, const int* flags, double& f, double* g, double* H) {
    f = acc[0];
    for (int j = 0; j < 5; ++j) g[j] = flags[j] ? acc[1 + j] : 0.0;
    int c = 6;
    for (int i = 0; i < 5; ++i)
        for (int j = i; j < 5; ++j) {
            const double v = (flags[i] && flags[j]) ? acc[c] : 0.0;
            H[i * 5 + j] = v;
            H[j * 5 + i] = v;
            ++c;
        }
}

// the next proposal from the accepted point's g, H on the fit subspace (dimension n,
// parameters idx[0..n)); returns true when the iteration ends here
template <int n>
__device__ inline bool step_propose(const FitArgs& a, SubState& s, const int* idx) {
    bool done = false;
        double gs[n], Hs[n * n], p[n];
        for (int r = 0; r < n; ++r) {
            gs[r] = s.g[idx[r]];
            for (int c = 0; c < n; ++c) Hs[r * n + c] = s.H[idx[r] * 5 + idx[c]];
        }
        int hits = 0;
        if (a.method == 0) {
            // the reference's own iteration (SciPy trust-ncg): truncated CG step, and
            // the exit where the model predicts no reduction in floating point
            tr_cg_steihaug_scipy<n>(s.f, gs, Hs, s.radius, p, &hits);
            const double pred = s.f - tr_model_value<n>(s.f, gs, Hs, p);
            if (!(pred > 0.0)) {
                // SciPy evaluates the proposal before it tests the predicted reduction
                // (_trustregion.py: m_proposed.fun, then `if predicted_reduction <= 0`), so the
                // reference's nfeval holds one more evaluation -- unless the proposal is the
                // point its ScalarFunction evaluated last.  No pass is spent on it here.
                bool cached = true;
                for (int j = 0; j < 5; ++j) {
                    double xp = s.x[j];
                    for (int r = 0; r < n; ++r) if (idx[r] == j) xp = s.x[j] + p[r];
                    cached = cached && (xp == s.xl[j]);
                }
                if (!cached) s.nfev += 1;
                s.status = PP_RC_STALL; done = true;
            } else {
                for (int j = 0; j < 5; ++j) s.xe[j] = s.x[j];
                for (int r = 0; r < n; ++r) s.xe[idx[r]] = s.x[idx[r]] + p[r];
                s.pred_red = pred;
                s.hits_boundary = hits;
            }
        } else {
        tr_subproblem<n>(gs, Hs, s.radius, p, &hits);
        double Hp[n];
        for (int r = 0; r < n; ++r) Hp[r] = vdot<n>(Hs + r * n, p);
        const double pred = -(vdot<n>(gs, p) + 0.5 * vdot<n>(p, Hp));
        // scipy: predicted_reduction <= 0 -> status 2 (the reference's normal exit)
        const double fpred = s.f - pred;     // what scipy compares: m(p) vs m(0)
        const double noise = 2.220446049250313e-16 * fabs(s.f);
        if (!(pred > 0.0) || !(fpred < s.f) || pred <= 64.0 * noise) {
            // The predicted reduction is below the rounding noise of f itself, so
            // the ratio test can no longer see it (scipy stops here with status 2,
            // the reference's normal exit, up to ~1e-9 rot short of the optimum).
            // Finish with the full Newton step when that step too is worth no more
            // than noise in f -- this close the quadratic model is exact to working
            // precision -- which lands at least as close to the optimum.
            double pn[n], Hpn[n];
            int hn = 0;
            tr_subproblem<n>(gs, Hs, 1e150, pn, &hn);
            for (int r = 0; r < n; ++r) Hpn[r] = vdot<n>(Hs + r * n, pn);
            const double predn = -(vdot<n>(gs, pn) + 0.5 * vdot<n>(pn, Hpn));
            if (!hn && predn >= 0.0 && predn <= 4096.0 * noise && s.iter + 1 < a.max_iter) {
                for (int j = 0; j < 5; ++j) s.xe[j] = s.x[j];
                for (int r = 0; r < n; ++r) s.xe[idx[r]] = s.x[idx[r]] + pn[r];
                s.fresh = 2;
            } else {
                s.status = PP_RC_STALL; done = true;
            }
        } else {
            for (int j = 0; j < 5; ++j) s.xe[j] = s.x[j];
            for (int r = 0; r < n; ++r) s.xe[idx[r]] = s.x[idx[r]] + p[r];
            s.pred_red = pred;
            s.hits_boundary = hits;
        }
        }
    return done;
}

// One decision of the trust-region loop for one subint, given the objective, gradient
// and Hessian at s.xe (just evaluated: counted; or the cached values of that point):
// the ratio test on the pending proposal (or the bookkeeping of an initial / closing
// evaluation), then the next proposal into s.xe.
// Returns true when the subint is finished (s.status set).  One thread.
__device__ inline bool step_decide(const FitArgs& a, SubState& s, double f, const double* g, const double* H, bool counted,
                                   bool is_pass) {
    bool finite = isfinite(f);
    for (int j = 0; j < 5; ++j) finite = finite && isfinite(g[j]);
    for (int j = 0; j < 25; ++j) finite = finite && isfinite(H[j]);
    const bool first = (s.fresh == 1), closing = (s.fresh == 2);
    s.fresh = 0;
    if (counted) s.nfev += 1;
    if (is_pass) s.npass += 1;
    bool done = false;
    if (closing) {
        // evaluation at the point of the final Newton step: the post-fit stage
        // (zero-covariance frequencies, errors, scales) uses sums taken AT the
        // returned parameters, as the reference does
        if (finite) {
            for (int j = 0; j < 5; ++j) { s.x[j] = s.xe[j]; s.g[j] = g[j]; }
            for (int j = 0; j < 25; ++j) s.H[j] = H[j];
            s.f = f;
            s.cur = 1 - s.cur;
        }
        s.status = PP_RC_STALL; done = true;
    } else if (first) {
        s.f = f;
        for (int j = 0; j < 5; ++j) s.g[j] = g[j];
        for (int j = 0; j < 25; ++j) s.H[j] = H[j];
        if (s.recentred == 0) {          // (objective hooks: at init_params only)
            s.f0 = f;
            for (int j = 0; j < 5; ++j) s.g0[j] = g[j];
            for (int j = 0; j < 25; ++j) s.H0[j] = H[j];
        }
        s.cur = 1 - s.cur;
        if (!finite) { s.status = PP_RC_NAN; done = true; }
        if (a.max_iter <= 0) { s.status = PP_RC_MAXITER; done = true; }
    } else {
        const double actual = s.f - f;
        const double rho = finite ? actual / s.pred_red : -1.0;
#ifdef PP_STEP_TRACE
        if (&s == &a.st[PP_STEP_TRACE])
            printf("dev it %2d f %.17g f_new %.17g actual %.3e pred %.3e rho %.3f radius %.3e hits %d\n", s.iter, s.f, f,
                   actual, s.pred_red, rho, s.radius, s.hits_boundary);
#endif
        if (rho < 0.25) s.radius *= 0.25;
        else if (rho > 0.75 && s.hits_boundary) s.radius = fmin(2.0 * s.radius, 1000.0);
        if (rho > 0.15) {
            for (int j = 0; j < 5; ++j) { s.xprev[j] = s.x[j]; s.x[j] = s.xe[j]; s.g[j] = g[j]; }
            for (int j = 0; j < 25; ++j) s.H[j] = H[j];
            s.f = f;
            s.cur = 1 - s.cur;
        }
        s.iter += 1;
        if (s.iter >= a.max_iter) { s.status = PP_RC_MAXITER; done = true; }
        if (!(s.radius > 1e-300)) { s.status = PP_RC_NAN; done = true; }
    }
    if (!done) {
        int idx[5], n = 0;
        for (int j = 0; j < 5; ++j) if (a.flags[j]) idx[n++] = j;
        PP_FOR_N(n, done = step_propose<N_>(a, s, idx));
    }
    return done;
}

// The evaluation at s.xe has arrived: decide, propose -- and keep deciding while the
// next proposal is the very point evaluated last.  SciPy's ScalarFunction hands back
// its cached values for that point without calling the objective (the reference's
// nfeval does not count it either).  It matters in the iteration's tail: once the
// optimum is reached to the last bit of f, a step whose actual reduction rounds to
// <= 0 is rejected, the radius shrinks by 4, and the SAME step is proposed again --
// ~15 times, until the radius is smaller than the step -- without a single new
// evaluation.
__device__ inline bool step_logic(const FitArgs& a, SubState& s, double f, const double* g, const double* H, bool is_pass) {
    for (int j = 0; j < 5; ++j) { s.xl[j] = s.xe[j]; s.gl[j] = g[j]; }
    for (int j = 0; j < 25; ++j) s.Hl[j] = H[j];
    s.fl = f;
    bool done = step_decide(a, s, f, g, H, true, is_pass);
    while (!done && s.fresh == 0) {
        bool same = true;
        for (int j = 0; j < 5; ++j) same = same && (s.xe[j] == s.xl[j]);
        if (!same) break;
        done = step_decide(a, s, s.fl, s.gl, s.Hl, false, false);
    }
    return done;
}

#include "pp_evalscat.h"
#include "pp_scatmodel.h"

// one trust-region iteration per subint (64 threads, lane 0 decides)
__global__ __launch_bounds__(64) void k_step(FitArgs a) {
    const int i = sub_of(a.act, blockIdx.x), tid = threadIdx.x;
    SubState& s = a.st[i];
    if (s.done || s.model == 1) return;      // (model == 1: k_scat_model_solve owns this evaluation)
    if (s.model >= 4) {
        // the model pass of this iteration was abandoned: the evaluation it stood for is
        // still to be made (next iteration, over the cross-spectrum)
        if (tid == 0) s.model = (s.model == 4) ? 0 : 3;
        return;
    }
    __shared__ double acc[PP_NACC];
    if (tid < PP_NACC) {
        double v = 0.0;
        for (int c = 0; c < a.nchunk; ++c) v += a.partial[((size_t)i * a.nchunk + c) * PP_NACC + tid];
        acc[tid] = v;
    }
    __syncthreads();
    if (s.fresh == 1 && a.scat && a.use_model) scat_model_geometry(a, i, s);   // (all 64 lanes)
    if (tid != 0) return;
    double f, g[5], H[25];
    unpack_acc(acc, a.flags, f, g, H);
    const bool done = step_logic(a, s, f, g, H, true);
    if (done) {
        s.done = 1;
        atomicSub(a.nactive, 1);
    } else if (a.scat && a.use_model && s.model == 0) {
        scat_model_request(a, s);
    }
}

// --------------------------------------------------------------------------
// real roots of a real polynomial (degree <= 6) by Aberth-Ehrlich iteration;
// returns the positive real root (after an optional sqrt) closest to target,
// NaN if none (reference: np.roots + selection, pptoaslib.py:791-794, 859-863)
// --------------------------------------------------------------------------
__device__ inline double pick_poly_root(const double* cin, int deg, double target, bool take_sqrt) {
    // strip leading zeros
    double c[7];
    int n = deg, off = 0;
    while (n > 0 && cin[off] == 0.0) { ++off; --n; }
    while (n > 0 && cin[off + n] == 0.0) --n;   // roots at zero are never selected
    for (int j = 0; j <= n; ++j) c[j] = cin[off + j] / cin[off];
    if (n <= 0) return NAN;
    // Fujiwara's bound on the root moduli, 2 max_j |c_j|^(1/j): within a factor of
    // two of the largest root however the coefficients are scaled (the nu_zero
    // polynomials in nu^2 have coefficient ratios of 1e17..1e31; Cauchy's bound
    // 1 + max|c_j| would start the iteration 25 decades away)
    double rad = 0.0;
    for (int j = 1; j <= n; ++j) rad = fmax(rad, pow(fabs(c[j]), 1.0 / (double)j));
    rad *= 2.0;
    double zr[6], zi[6];
    for (int j = 0; j < n; ++j) {
        double s, co;
        sincos(PP_TWO_PI * j / n + 0.4, &s, &co);
        zr[j] = 0.5 * rad * co; zi[j] = 0.5 * rad * s;
    }
    // (stops a few ulp short: every selected root is polished by Newton below, and
    // a step that hovers at the rounding level would otherwise never meet a tighter
    // test; roots that are zero to rounding are measured against the bound instead)
    for (int it = 0; it < 200; ++it) {
        double maxstep = 0.0;
        for (int j = 0; j < n; ++j) {
            // p(z), p'(z) by Horner
            double pr = 1.0, pi = 0.0, dr = 0.0, di = 0.0;
            for (int m = 1; m <= n; ++m) {
                const double ndr = dr * zr[j] - di * zi[j] + pr, ndi = dr * zi[j] + di * zr[j] + pi;
                dr = ndr; di = ndi;
                const double npr = pr * zr[j] - pi * zi[j] + c[m], npi = pr * zi[j] + pi * zr[j];
                pr = npr; pi = npi;
            }
            const double dn = dr * dr + di * di;
            if (dn == 0.0) continue;
            // w = p/p'
            double wr = (pr * dr + pi * di) / dn, wi = (pi * dr - pr * di) / dn;
            double sr = 0.0, si = 0.0;
            for (int m = 0; m < n; ++m)
                if (m != j) {
                    const double er = zr[j] - zr[m], ei = zi[j] - zi[m], en = er * er + ei * ei;
                    if (en > 0.0) { sr += er / en; si -= ei / en; }
                }
            // step = w / (1 - w*s)
            const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr), qn = qr * qr + qi * qi;
            const double stx = (wr * qr + wi * qi) / qn, sty = (wi * qr - wr * qi) / qn;
            zr[j] -= stx; zi[j] -= sty;
            maxstep = fmax(maxstep, (fabs(stx) + fabs(sty)) / (fabs(zr[j]) + fabs(zi[j]) + 1e-10 * rad));
        }
        if (maxstep < 8e-16) break;
    }
    double best = NAN, bestd = INFINITY;
    for (int j = 0; j < n; ++j) {
        if (fabs(zi[j]) > 1e-9 * fabs(zr[j]) || !(zr[j] > 0.0)) continue;
        // polish the real root with Newton on the real polynomial
        double x = zr[j];
        for (int it = 0; it < 4; ++it) {
            double pv = 1.0, dv = 0.0;
            for (int m = 1; m <= n; ++m) { dv = dv * x + pv; pv = pv * x + c[m]; }
            if (dv != 0.0) x -= pv / dv;
        }
        if (!(x > 0.0)) continue;
        const double v = take_sqrt ? sqrt(x) : x;
        if (fabs(target - v) < bestd) { bestd = fabs(target - v); best = v; }
    }
    return best;
}

// --------------------------------------------------------------------------
// post-fit stage: one 256-thread block per subint
// --------------------------------------------------------------------------
__device__ __forceinline__ double py_wrap_half(double x) {
    // reference pptoaslib.py:1056-1057
    if (fabs(x) >= 0.5) { x = fmod(x, 1.0); if (x < 0.0) x += 1.0; }
    if (x >= 0.5) x -= 1.0;
    return x;
}

// CPT > 0 (phase / DM / GM fits, nchan <= NT CPT; NT threads): what the four passes read of a channel -- weight,
// frequency, the three sums, template power, data power -- is fetched ONCE, into registers, with every
// load in flight together; CPT = 0: any fit, any nchan, read pass by pass.  Same sums in the same order.
// NVW > 1 (CPT = 0 only): NT / NVW real threads walk the NT / 64 waves of the NT-thread kernel in turn and form its
// block-wide sums from the same per-wave totals in the same order (vblock_sum, as taylor_solve_body): bitwise its results.
// `tid`: the real thread.  scratch: PP_BSUM_DOUBLES(NT / 64, 31) doubles of LDS; sh: one double.
template <int CPT, int NT, int NVW>
__device__ __forceinline__ void finalize_body(const FitArgs& a, const int i, const int tid, double* scratch, double* sh) {
    static_assert(NVW == 1 || CPT == 0, "channels held in registers belong to real threads");
    constexpr int RT = NT / NVW;
    const SubState& s = a.st[i];
    int flip = 0;
    const double P = a.P[i];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const double* csum = a.csum + ((size_t)s.cur * a.nsub + i) * a.nchan * a.ncs;
    const int slot = a.slot ? a.slot[i] : 0;
    const double* msum = as_global(a.msum[slot]);
    const int* fl = a.flags;
    const double phi = s.x[0], DM = s.x[1], GM = s.x[2], alpha = s.x[4];
    const double taup = s.x[3];
    const double tau = a.log10_tau ? pow(10.0, taup) : taup;
    const bool scat_on = a.scat && (tau != 0.0);
    const double nfDM = a.nu_fit[i * 3], nfGM = a.nu_fit[i * 3 + 1], nftau = a.nu_fit[i * 3 + 2];
    double noDM = a.nu_out ? a.nu_out[i * 3] : NAN, noGM = a.nu_out ? a.nu_out[i * 3 + 1] : NAN,
           notau = a.nu_out ? a.nu_out[i * 3 + 2] : NAN;
    double rw[CPT ? CPT : 1], rf[CPT ? CPT : 1], rc0[CPT ? CPT : 1], rc1[CPT ? CPT : 1], rc2[CPT ? CPT : 1],
           rS[CPT ? CPT : 1], rd[CPT ? CPT : 1];
    if constexpr (CPT > 0) {
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int n = tid + NT * q;
            const bool in = n < a.nchan;
            rw[q] = in ? wts[n] : 0.0;
            rf[q] = in ? freqs[n] : 0.0;
            rc0[q] = in ? csum[(size_t)n * 3] : 0.0; rc1[q] = in ? csum[(size_t)n * 3 + 1] : 0.0;
            rc2[q] = in ? csum[(size_t)n * 3 + 2] : 0.0;
            rS[q] = in ? msum[n] : 0.0;
            rd[q] = in ? a.sdraw[(size_t)i * a.nchan + n] : 0.0;
        }
    }
    // body(n, q): channel n = vt + NT q of (virtual) thread vt
    auto for_channels = [&](const int vt, auto&& body) {
        if constexpr (CPT > 0) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int n = vt + NT * q;
                if (n < a.nchan) body(n, q);
            }
        } else {
            for (int n = vt; n < a.nchan; n += NT) body(n, 0);
        }
    };
    auto wt_of = [&](int n, int q) { if constexpr (CPT > 0) return rw[q]; else return wts[n]; };
    auto nu_of = [&](int n, int q) { if constexpr (CPT > 0) return rf[q]; else return freqs[n]; };
    auto load_cs = [&](int n, int q, double* cs) {
        if constexpr (CPT > 0) {
            cs[0] = rc0[q]; cs[1] = rc1[q]; cs[2] = rc2[q];
            cs[3] = cs[4] = cs[5] = 0.0; cs[6] = rS[q]; cs[7] = cs[8] = 0.0;
        } else if (a.ncs == 3) {
            cs[0] = csum[(size_t)n * 3]; cs[1] = csum[(size_t)n * 3 + 1]; cs[2] = csum[(size_t)n * 3 + 2];
            cs[3] = cs[4] = cs[5] = 0.0; cs[6] = msum[n]; cs[7] = cs[8] = 0.0;
        } else {
            for (int j = 0; j < PP_NCS; ++j) cs[j] = csum[(size_t)n * PP_NCS + j];
        }
    };
#ifdef PP_SOLVE_CLOCKS
    const long long fk0 = clock64();
#endif
    // ---- pass 0: Sd, mean frequency, used channels -------------------------
    double v3[3];
    vblock_sum<3, 31, NT, NVW>(v3, scratch, flip, nullptr, tid, [&](const int vt, double (&u3)[3], double&) __attribute__((always_inline)) {
        for_channels(vt, [&](int n, int q) {
            const double w = wt_of(n, q);
            double sd;
            if constexpr (CPT > 0) sd = rd[q]; else sd = (w != 0.0) ? a.sdraw[(size_t)i * a.nchan + n] : 0.0;
            if (w != 0.0) { u3[0] += w * sd; u3[1] += nu_of(n, q); u3[2] += 1.0; }
        });
    });
    const double Sd = v3[0], nused = v3[2], fmean = v3[1] / v3[2];
#ifdef PP_SOLVE_CLOCKS
    const long long fk1 = clock64();
#endif
    int pat = 0;
    for (int j = 0; j < 5; ++j) pat = pat * 2 + (fl[j] ? 1 : 0);
    if (pat == 0x1F) pat = 0x1B;  // [1,1,1,1,1] is approximated by [1,1,0,1,1] (:893-901)
    // effective flags of the Hessian used for nu_zero
    int efl[5];
    for (int j = 0; j < 5; ++j) efl[j] = fl[j];
    if ((fl[0] && fl[1] && fl[2] && fl[3] && fl[4])) efl[2] = 0;
    const bool need_zero = isnan(noDM) || isnan(noGM) || isnan(notau);
    double nzDM = nfDM, nzGM = nfGM, nztau = nftau;
    if (need_zero) {
        // sums over channels of local terms times powers of frequency
        // v[0..]: see per-pattern use below
        double v[20];
        vblock_sum<20, 31, NT, NVW>(v, scratch, flip, nullptr, tid, [&](const int vt, double (&v)[20], double&) __attribute__((always_inline)) {
        for_channels(vt, [&](int n, int q) {
            const double w = wt_of(n, q);
            if (w == 0.0) return;
            double cs[PP_NCS];
            load_cs(n, q, cs);
            const Local L = local_terms(cs, w);
            ChanGeom cg;
            const double nu = nu_of(n, q);
            chan_geom(nu, P, nfDM, nfGM, nftau, tau, alpha, a.log10_tau, scat_on, cg);
            const double a2 = 1.0 / (nu * nu), a4 = a2 * a2, lf = scat_on ? log(nu) : 0.0;
            const double h = L.Lpp;
            const double q2p = cg.taun;                                  // q2 / lnf
            const double q12p = a.log10_tau ? PP_LN10 * cg.taun : (scat_on ? cg.taun / tau : 0.0);
            const double Htt = L.Ltt * cg.q1 * cg.q1 + L.Gt * cg.q11;   // H[3][3]
            const double Hta = L.Ltt * cg.q1 * cg.q2 + L.Gt * cg.q12;   // H[3][4]
            const double Haa = L.Ltt * cg.q2 * cg.q2 + L.Gt * cg.q22;   // H[4][4]
            const double Hta_l = L.Ltt * cg.q1 * q2p + L.Gt * q12p;     // H[3][4]/lnf
            switch (pat) {
            case 0x18:  // [1,1,0,0,0]
                v[0] += a2 * h; v[1] += h; break;
            case 0x14:  // [1,0,1,0,0]
                v[0] += a4 * h; v[1] += h; break;
            case 0x03:  // [0,0,0,1,1]
                v[0] += lf * Hta_l; v[1] += Hta_l; break;
            case 0x1A: {  // [1,1,0,1,0]
                const double H23 = L.Lpt * cg.q1;
                v[0] += H23;            // H13 = sum Lpt q1
                v[1] += Htt;            // H33
                v[2] += a2 * H23; v[3] += a2 * h; v[4] += h;
                break; }
            case 0x1C: {  // [1,1,1,0,0]
                const double pj = (a.option == 1) ? cg.p1 : cg.p2;
                v[0] += h * a4; v[1] += h;               // A, B
                v[2] += h * pj * a2; v[3] += h * pj;     // C, D
                v[4] += h * pj * a4;                      // E  (F = D)
                v[5] += h * a2;                           // G  (H = B)
                break; }
            case 0x1B: {  // [1,1,0,1,1]
                const double H23 = L.Lpt * cg.q1, H24 = L.Lpt * cg.q2;
                const double H41 = L.Lpt * q2p, H42 = L.Lpt * cg.p1 * q2p, H43 = Hta_l;
                v[0] += h; v[1] += h * cg.p1 * cg.p1; v[2] += Htt; v[3] += Haa;   // H11 H22 H33 H44
                v[4] += h * cg.p1; v[5] += H23; v[6] += H24;                         // H12 H13 H14
                v[7] += H23 * cg.p1; v[8] += H24 * cg.p1; v[9] += Hta;               // H23 H24 H34
                v[10] += a2 * h; v[11] += a2 * H23; v[12] += a2 * H24;
                v[13] += lf * H41; v[14] += lf * H42; v[15] += lf * H43;
                v[16] += H41; v[17] += H42; v[18] += H43;
                break; }
            case 0x1E: {  // [1,1,1,1,0]
                const double k1 = PP_DCONST / P, k2 = PP_DCONST * PP_DCONST / P;
                const double Hpt = L.Lpt * cg.q1;
                v[0] += Hpt; v[1] += Htt;     // H14, H44
                if (a.option == 0) {
                    const double H21 = h * k1, H23 = h * cg.p2 * k1, H24 = Hpt * k1;
                    const double H31 = h * k2, H33 = h * cg.p2 * k2, H34 = Hpt * k2;
                    v[2] += a4 * H34; v[3] += H34; v[4] += a2 * H21; v[5] += H21;
                    v[6] += a4 * H31; v[7] += H31; v[8] += a2 * H23; v[9] += H23;
                    v[10] += a4 * H33; v[11] += H33; v[12] += a2 * H24; v[13] += H24;
                } else {
                    const double H21 = h * k1, H22 = h * cg.p1 * k1, H24 = Hpt * k1;
                    const double H31 = h * k2, H32 = h * cg.p1 * k2, H34 = Hpt * k2;
                    v[2] += a2 * H24; v[3] += H24; v[4] += a4 * H31; v[5] += H31;
                    v[6] += a2 * H21; v[7] += H21; v[8] += a4 * H32; v[9] += H32;
                    v[10] += a2 * H22; v[11] += H22; v[12] += a4 * H34; v[13] += H34;
                }
                break; }
            default: break;
            }
        });
        });
#ifdef PP_SOLVE_CLOCKS
        if (tid == 0 && (i % 200) == 0) printf("fin %d: zero-loop %lld\n", i, clock64() - fk1);
#endif
        switch (pat) {
        case 0x18: nzDM = 1.0 / sqrt(v[0] / v[1]); break;
        case 0x14: nzGM = pow(v[0] / v[1], -0.25); break;
        case 0x03: nztau = exp(v[0] / v[1]); break;
        case 0x1A: {
            const double numer = v[0] * v[2] - v[1] * v[3], denom = v[0] * v[0] - v[1] * v[4];
            nzDM = 1.0 / sqrt(numer / denom);
            break; }
        case 0x1C:
            if (a.option == 0 || a.option == 1) {
                const double A = v[0], B = v[1], C = v[2], D = v[3], E = v[4], F = v[3], G = v[5], Hh = v[1];
                // coefficients of nu^6, nu^4, nu^2, 1 -> cubic in y = nu^2
                double co[4] = {A * C - E * G, E * Hh - A * D, F * G - B * C, B * D - F * Hh};
                if (tid == 0) sh[0] = pick_poly_root(co, 3, fmean, true);
                __syncthreads();
                nzDM = nzGM = sh[0];
                __syncthreads();
            }
            break;
        case 0x1B: {
            const double H11 = v[0], H22 = v[1], H33 = v[2], H44 = v[3], H12 = v[4], H13 = v[5], H14 = v[6];
            const double H23 = v[7], H34 = v[9];
            const double c1 = H34 * H34 - H33 * H44, c2 = H13 * H44 - H14 * H34, c3 = H14 * H33 - H13 * H34;
            nzDM = 1.0 / sqrt((c1 * v[10] + c2 * v[11] + c3 * v[12]) / (c1 * v[0] + c2 * v[5] + c3 * v[6]));
            const double e1 = H13 * H22 - H12 * H23, e2 = H11 * H23 - H12 * H13, e3 = H12 * H12 - H11 * H22;
            nztau = exp((e1 * v[13] + e2 * v[14] + e3 * v[15]) / (e1 * v[16] + e2 * v[17] + e3 * v[18]));
            break; }
        case 0x1E:
            if (a.option == 0 || a.option == 1) {
                const double H14 = v[0], H44 = v[1];
                const double A = v[2], aa = v[3], B = v[4], b = v[5], C = v[6], c = v[7], D = v[8], d = v[9];
                const double E = v[10], e = v[11], F = v[12], f = v[13];
                double co[6];
                int deg;
                if (a.option == 0) {
                    co[0] = A * A * B + H44 * C * D + H14 * E * F - H44 * B * E - A * C * F - H14 * A * D;
                    co[1] = -A * A * b - H44 * C * d - H14 * E * f + H44 * b * E + A * C * f + H14 * A * d;
                    co[2] = -2 * A * aa * B - H44 * c * D - H14 * e * F + H44 * B * e + (A * c + aa * C) * F + H14 * aa * D;
                    co[3] = 2 * A * aa * b + H44 * c * d + H14 * e * f - H44 * b * e - (A * c + aa * C) * f - H14 * aa * d;
                    co[4] = aa * aa * B - aa * c * F;
                    co[5] = -aa * aa * b + aa * c * f;
                    deg = 5;
                } else {
                    co[0] = A * A * B + H44 * C * D + H14 * E * F - H44 * B * E - A * C * F - H14 * A * D;
                    co[1] = -2 * A * aa * B - H44 * c * D - H14 * e * F + H44 * B * e + (A * c + aa * C) * F + H14 * aa * D;
                    co[2] = -(A * A * b - aa * aa * B) - H44 * C * d - H14 * E * f + H44 * b * E + (A * C * f - aa * c * F) + H14 * A * d;
                    co[3] = 2 * A * aa * b + H44 * c * d + H14 * e * f - H44 * b * e - (A * c + aa * C) * f - H14 * aa * d;
                    co[4] = -aa * aa * b + aa * c * f;
                    deg = 4;
                }
                if (tid == 0) sh[0] = pick_poly_root(co, deg, fmean, true);
                __syncthreads();
                nzDM = nzGM = sh[0];
                __syncthreads();
            }
            break;
        default: break;
        }
    }
    if (isnan(noDM)) noDM = nzDM;
    if (isnan(noGM)) noGM = nzGM;
    if (isnan(notau)) notau = nztau;
    if (a.is_toa) {  // pptoaslib.py:1048-1050
        if (fl[1]) noGM = noDM;
        else if (fl[2]) noDM = noGM;
    }
    const double k1 = PP_DCONST / P, k2 = PP_DCONST * PP_DCONST / P;
    const double inv2 = (nfDM == INFINITY) ? 0.0 : 1.0 / (nfDM * nfDM);
    const double inv4 = (nfGM == INFINITY) ? 0.0 : 1.0 / (nfGM * nfGM * nfGM * nfGM);
    const double phi_inf = phi + PP_DCONST * DM * (0.0 - inv2) / P + PP_DCONST * PP_DCONST * GM * (0.0 - inv4) / P;
    const double phi_out = py_wrap_half(phi_inf + k1 * DM / (noDM * noDM) + k2 * GM / (noGM * noGM * noGM * noGM));
    const double tau_out = tau * pow(notau / nftau, alpha);
    // ---- covariance with the amplitude parameters at the output references --
    // A_ij (15 upper) + Schur correction sum_n U_i U_j/(2 S_n) (15) -> 30 sums,
    // then snr^2
#ifdef PP_SOLVE_CLOCKS
    const long long fk2 = clock64();
#endif
    double m[31];
    vblock_sum<31, 31, NT, NVW>(m, scratch, flip, nullptr, tid, [&](const int vt, double (&m)[31], double&) __attribute__((always_inline)) {
    for_channels(vt, [&](int n, int q) {
        const double w = wt_of(n, q);
        if (w == 0.0) return;
        double cs[PP_NCS];
        load_cs(n, q, cs);
        ChanGeom cg;
        chan_geom(nu_of(n, q), P, noDM, noGM, notau, tau_out, alpha, a.log10_tau, scat_on, cg);
        const double A0 = cs[0], A1 = cs[1], A2 = cs[2], T1 = cs[3], T2 = cs[4], A1T = cs[5];
        const double S0 = cs[6], S1 = cs[7], S2 = cs[8];
        const double r = A0 / S0;                     // scale a_n
        // reduced local second derivatives (pptoaslib.py:694-697)
        const double Lpp = -2.0 * w * r * A2, Lpt = -2.0 * w * r * A1T;
        const double Ltt = -2.0 * w * (r * T2 - 0.5 * r * r * S2);
        const double Gt = -2.0 * w * (r * T1 - 0.5 * r * r * S1);
        const double J[5] = {1.0, cg.p1, cg.p2, cg.q1, cg.q2};
        // cross terms with a_n (pptoaslib.py:690): U = -2 (dC - a dS)
        const double up = -2.0 * w * A1, ut = -2.0 * w * (T1 - r * S1);
        const double U[5] = {up, up * cg.p1, up * cg.p2, ut * cg.q1, ut * cg.q2};
        const double cinv = 1.0 / (2.0 * w * S0);
        int c = 0;
        for (int ii = 0; ii < 5; ++ii)
            for (int jj = ii; jj < 5; ++jj) {
                double hij;
                if (jj < 3) hij = Lpp * J[ii] * J[jj];
                else if (ii < 3) hij = Lpt * J[ii] * J[jj];
                else hij = Ltt * J[ii] * J[jj] + Gt * (ii == 3 ? (jj == 3 ? cg.q11 : cg.q12) : cg.q22);
                m[c] += hij;
                m[15 + c] += U[ii] * U[jj] * cinv;
                ++c;
            }
        m[30] += w * A0 * r;      // (a_n sqrt(S_n))^2 = w A0^2/S0
    });
    });
#ifdef PP_SOLVE_CLOCKS
    const long long fk3 = clock64();
#endif
#ifdef PP_SOLVE_CLOCKS
    const long long fk4 = clock64();
#endif
    // positions of the fitted parameters, and everything that works on the fit subspace -- the inverse, the
    // per-channel scale errors, the covariance outputs -- with its dimension a compile-time number (PP_FOR_N):
    // no dynamically indexed array (those live in scratch memory), same operations in the same order
    int ix[5] = {0, 0, 0, 0, 0}, nfit = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j)
        if (fl[j]) {
            if (nfit == 0) ix[0] = j; else if (nfit == 1) ix[1] = j; else if (nfit == 2) ix[2] = j;
            else if (nfit == 3) ix[3] = j; else ix[4] = j;
            ++nfit;
        }
    // X = inv(Afit - U Cinv U^T)   (every thread computes the same small matrix)
    double full[25];
    {
        int c = 0;
#pragma unroll
        for (int ii = 0; ii < 5; ++ii)
#pragma unroll
            for (int jj = ii; jj < 5; ++jj) {
                const double vv = (fl[ii] && fl[jj]) ? (m[c] - m[15 + c]) : 0.0;
                full[ii * 5 + jj] = vv; full[jj * 5 + ii] = vv;
                ++c;
            }
    }
    // (values, not addresses: given an array the compiler turns the selection into an indexed load from a scratch copy)
    auto sel5 = [](double v0, double v1, double v2, double v3, double v4, int k) {
        return k == 0 ? v0 : k == 1 ? v1 : k == 2 ? v2 : k == 3 ? v3 : v4;
    };
#ifdef PP_SOLVE_CLOCKS
    long long fk5 = 0;
#endif
    auto on_subspace = [&](auto NC) {
        constexpr int N = decltype(NC)::value;
        double Xs[N * N];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const int k = ix[r];
            double row[5];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) row[jj] = sel5(full[jj], full[5 + jj], full[10 + jj], full[15 + jj], full[20 + jj], k);
#pragma unroll
            for (int c2 = 0; c2 < N; ++c2) Xs[r * N + c2] = sel5(row[0], row[1], row[2], row[3], row[4], ix[c2]);
        }
        const bool inv_ok = mat_inverse<N>(Xs);
#ifdef PP_SOLVE_CLOCKS
        fk5 = clock64();
#endif
        // ---- per-channel outputs -------------------------------------------------
#pragma unroll 1
        for (int kv = 0; kv < NVW; ++kv)
        for_channels(tid + RT * kv, [&](int n, int q) {
            const double w = wt_of(n, q);
            double sc = 0.0, se = 0.0, sn = 0.0;
            if (w != 0.0) {
                double cs[PP_NCS];
                load_cs(n, q, cs);
                ChanGeom cg;
                chan_geom(nu_of(n, q), P, noDM, noGM, notau, tau_out, alpha, a.log10_tau, scat_on, cg);
                const double r = cs[0] / cs[6];
                const double up = -2.0 * w * cs[1], ut = -2.0 * w * (cs[3] - r * cs[7]);
                const double U[5] = {up, up * cg.p1, up * cg.p2, ut * cg.q1, ut * cg.q2};
                double Us[N];
#pragma unroll
                for (int r1 = 0; r1 < N; ++r1) Us[r1] = sel5(U[0], U[1], U[2], U[3], U[4], ix[r1]);
                const double cinv = 1.0 / (2.0 * w * cs[6]);
                double uXu = 0.0;
#pragma unroll
                for (int r1 = 0; r1 < N; ++r1)
#pragma unroll
                    for (int c1 = 0; c1 < N; ++c1) uXu += Us[r1] * Xs[r1 * N + c1] * Us[c1];
                sc = r;
                se = sqrt(2.0 * (cinv + cinv * cinv * uXu));
                sn = r * sqrt(w * cs[6]);
            }
            if (a.o_scales) a.o_scales[(size_t)i * a.nchan + n] = sc;
            if (a.o_scale_errs) a.o_scale_errs[(size_t)i * a.nchan + n] = se;
            if (a.o_csnr) a.o_csnr[(size_t)i * a.nchan + n] = sn;
        });
        if (tid == 0) {
            double* oe = a.o_errs + (size_t)i * 5;
            double* oc = a.o_cov + (size_t)i * 25;
            for (int j = 0; j < 5; ++j) oe[j] = 0.0;
            for (int j = 0; j < 25; ++j) oc[j] = 0.0;
#pragma unroll
            for (int r = 0; r < N; ++r) {
                if (r >= nfit) break;         // (nfit = 0 lands in the N = 5 instance)
#pragma unroll
                for (int c = 0; c < N; ++c) if (c < nfit) oc[ix[r] * 5 + ix[c]] = inv_ok ? 2.0 * Xs[r * N + c] : NAN;
                oe[ix[r]] = inv_ok ? sqrt(2.0 * Xs[r * N + r]) : NAN;
            }
        }
    };
    // (the register-cached variants serve fits without scattering: at most phi, DM, GM)
    if constexpr (CPT > 0) { PP_FOR_N3(nfit, on_subspace(std::integral_constant<int, N_>{})); }
    else { PP_FOR_N(nfit, on_subspace(std::integral_constant<int, N_>{})); }
#ifdef PP_SOLVE_CLOCKS
    if (tid == 0 && (i % 200) == 0)
        printf("fin %d: pass0 %lld zero %lld covloop %lld covred %lld inv %lld chan %lld\n", i, fk1 - fk0, fk2 - fk1, fk3 - fk2,
               fk4 - fk3, fk5 - fk4, clock64() - fk5);
#endif
    if (tid == 0) {
        double* op = a.o_params + (size_t)i * 5;
        op[0] = phi_out; op[1] = DM; op[2] = GM;
        op[3] = a.log10_tau ? log10(tau_out) : tau_out;
        op[4] = alpha;
        const double* oe = a.o_errs + (size_t)i * 5;      // (written above, by this thread)
        a.o_nu[(size_t)i * 3] = noDM; a.o_nu[(size_t)i * 3 + 1] = noGM; a.o_nu[(size_t)i * 3 + 2] = notau;
        const double chi2 = Sd + s.f;
        const double dof = nused * a.nbin - (nfit + nused);
        a.o_chi2[i] = chi2;
        a.o_rchi2[i] = chi2 / dof;
        a.o_snr[i] = sqrt(m[30]);
        a.o_nfev[i] = s.nfev;
        a.o_rc[i] = s.status;
        a.o_npass[i] = s.npass;
        if (i == 0 && !a.tail_fused) a.o_npass[a.nsub] = *a.nactive;     // (subints still unfinished: read back with the outputs)
        if (a.o_rec) {
            double* rec = a.o_rec + (size_t)i * PP_RECORD_WIDTH;
            for (int j = 0; j < 5; ++j) { rec[j] = op[j]; rec[5 + j] = oe[j]; }
            rec[10] = noDM; rec[11] = noGM; rec[12] = notau;
            rec[13] = chi2; rec[14] = chi2 / dof; rec[15] = sqrt(m[30]);
            rec[16] = (double)s.nfev; rec[17] = (double)s.status;
        }
        if (a.o_f0) a.o_f0[i] = s.f0;
        if (a.o_g0) for (int j = 0; j < 5; ++j) a.o_g0[(size_t)i * 5 + j] = s.g0[j];
        if (a.o_H0) for (int j = 0; j < 25; ++j) a.o_H0[(size_t)i * 25 + j] = s.H0[j];
    }
}

template <int CPT, int NT>
__global__ __launch_bounds__(NT) void k_finalize(FitArgs a) {
    __shared__ double scratch[PP_BSUM_DOUBLES(NT / 64, 31)];
    __shared__ double sh[2];
    finalize_body<CPT, NT, 1>(a, blockIdx.x, threadIdx.x, scratch, sh);
}
// one real wave per subint walking the NT / 64 waves of k_finalize<., NT> in turn: bitwise its results
template <int NT>
__global__ __launch_bounds__(64) void k_finalize_v(FitArgs a) {
    __shared__ double scratch[PP_BSUM_DOUBLES(NT / 64, 31)];
    __shared__ double sh[2];
    finalize_body<0, NT, NT / 64>(a, blockIdx.x, threadIdx.x, scratch, sh);
}

}  // namespace pp
