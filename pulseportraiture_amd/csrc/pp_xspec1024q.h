// k_xspec for 2048-bin rows whose template keeps fewer than 512 harmonics (MODE 2,
// noise given or measured): the benchmark shape, built on the one-exchange FFT of pp_fftq.h.
//
// Same contract as k_xspec<1024, Tin, TAIL, 2> (pp_kernels.h).  What differs is the
// path of the row through the LDS, which is the unit this kernel saturates (one LDS
// per CU, eight resident rows; see pp_fftq.h):
//
//   k_xspec          stage-1 store, stage-2 load + store, stage-3 load + store,
//                    split loads (k and M - k)                    110 ds_*_b128 per row
//   this kernel      one 16 x 16 transpose (16 + 16), then the split reads only the
//                    PARTNERS: lane t ends with Z[lam + 64 kd] in register kd and works
//                    on harmonics k = lam + 64 j (j = 0..6) -- Z_k is already in its
//                    registers; Z_{M-k} = Z[(64 - lam) + 64 (15 - j)] sits in register
//                    15 - j of the lane that owns 64 - lam, so every lane publishes its
//                    registers 9..15 and reads seven values back     46 per row
//
// The lane that owns lam = 0 has no harmonic 0 to work on (F0_fact = 0): it takes
// k = 64 (j + 1) instead, from its registers 1..7, and is its own partner.  Both
// exchanges are free of bank conflicts: the transpose writes with a pitch of 17
// elements and reads contiguously; the partner map sends every aligned group of eight
// lanes to eight lanes with distinct low three bits.
#pragma once
#ifndef PP_TAIL_M_LDS
#define PP_TAIL_M_LDS 1
#endif
#ifndef PP_MAIN_M_LDS
#define PP_MAIN_M_LDS 0     // (the noise-given kernel does not spill: measured, no gain)
#endif
#include "pp_fftq.h"

// The harmonics a lane works on come in slots of 64 (k = kb + 64 j) and a channel's template keeps a PREFIX of them
// (kt_n is a multiple of 64, wave-uniform).  Round 5: the slot loop ends with the last kept slot instead of walking
// the dropped ones for their two recurrences (split twiddle W^k and phasor e^{i kappa phi}: 8 f64 instructions a
// slot) and their partner read from LDS: the example template keeps 4.5 of 7 (of 8 at 1024 bins) slots on average.
// Same sums, same bits.
#ifndef PP_SLOT_EARLY_EXIT
#define PP_SLOT_EARLY_EXIT 1
#endif
#ifndef PP_TICKET_SPREAD
#define PP_TICKET_SPREAD 0
#endif
// a wave asks for its ticket after a wave-specific number of rows, uniform over this fraction of its share of the launch
// (every wave asks, the first `tickets` to ask get one: with 3/4 and four waves per ticket they are gone after the first fifth)
#ifndef PP_TICKET_WINDOW_NUM
#define PP_TICKET_WINDOW_NUM 3u
#define PP_TICKET_WINDOW_DEN 4u
#endif
#ifndef PP_TAIL_HOOKS
#define PP_TAIL_HOOKS 1            // the transform kernels of 2048-bin rows can work off the previous batch's solve / post-fit
                                   // tickets (tail_work).  Not the 1024-bin kernel: it is compiled for three waves per SIMD
                                   // (168 registers), and a function called from it inherits that budget -- the tail's code
                                   // then spills, in EVERY carrier: configs[1] lost 6 % with it, the headline gained less
#endif
#ifndef PP_SLOT_PUBLISH_KEPT
#define PP_SLOT_PUBLISH_KEPT 0     // k_xspec_q1024: 1 = only the kept slots' partner registers are published to LDS (2.5 fewer
                                   // ds_write_b128 of 65 LDS instructions a row).  Measured neutral to -0.5 %
                                   // (profiles/r05_publish_kept_ab.txt): the scalar jump costs what the stores save
#endif

namespace pp {

// Where the next row's loads are queued (measured choices, profiles/README.md): f64 rows in two
// halves -- eight registers' worth after the stage-1 twiddles (fftq1024's WHEN = 1), the rest
// behind the partner exchange -- and with the stage twiddles re-read per row; f32 rows whole, a
// quarter into stage 1 (WHEN = 0), twiddles held.
constexpr bool Q_SPLIT_PREFETCH = true, Q_TW_RELOAD = true;
constexpr int Q_PREFETCH_F64 = 1, Q_PREFETCH_F32 = 0;
// TAIL: also measure the noise from the top quarter of the power spectrum (errs == NULL,
// get_noise_PS): harmonics 768..1023 are this lane's registers 12..15 against the
// partner's registers 3..0 (five more registers published, four more read back), the
// Nyquist harmonic is Re Z_0 - Im Z_0.
template <typename Tin, bool TAIL>
__global__ __launch_bounds__(64, 2) void k_xspec_q1024(XspecArgs a) {
    constexpr int M = 1024, T = 64, R1 = 16, PER1 = 1;
    constexpr int NSL = 7;                     // slots: 2 Kt < M  ->  k <= 448 = 64 * 7
    typedef typename RawOf<Tin>::type Raw;
    constexpr int NRED = PP_TSTRIDE + (TAIL ? 2 : 1);       // the 12 Taylor sums, S_d (and the noise tail)
    static_assert(NRED <= 16, "wave_reduce_lds takes 16 values");
    static_assert(PP_TJ == 10, "power ladder written for order 10");
    constexpr int WRED = PP_WRED_DOUBLES(NRED) / 2;   // in cplx
    constexpr int LDSN = WRED > FFTQ_LDS_ELEMS ? WRED : FFTQ_LDS_ELEMS;
    static_assert(2 * LDSN >= PP_TAIL_LDS_DOUBLES, "tail_work's layout of this kernel's LDS");
    // TAIL, f64 rows: the template values of the last NML slots live in LDS beside the image instead of in registers
    // (with them held the kernel spilled one of them + three dwords to scratch, and the scratch reload in the middle
    // of the row queues behind the prefetched half row: vector memory returns in order)
    constexpr int NML = ((PP_TAIL_M_LDS && TAIL) || (PP_MAIN_M_LDS && !TAIL)) && sizeof(Tin) == 8 ? 2 : 0;
    __shared__ cplx lds[LDSN + 64 * NML];
    cplx* const ldsm = lds + LDSN + threadIdx.x;
    int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    Raw cur[PER1][R1];
    // stage twiddles: W_1024^tid, W_64^(tid & 15)
    // f64 rows re-read them every row (three L1-resident loads, older than the prefetch)
    // instead of holding 12 registers through the whole row: with them held, three of the
    // template values spill to scratch, and a scratch reload queues BEHIND the prefetched
    // row (vector memory returns in order) -- the split then waits for the next row's HBM
    // data (15.1 -> 14.1 ms per 1024 fits)
    constexpr bool TWR = Q_TW_RELOAD && sizeof(Tin) == 8;
    cplx t1 = a.twB[2 * tid], t2 = a.twB[32 * (tid & 15)];
    // this lane's harmonics k = kb + 64 j; split twiddle W_B^kb, stepped by W_B^64
    const int lam0 = fftq_lambda(tid);
    cplx wb0 = a.twB[lam0 ? lam0 : 64];
    const cplx wbT = a.twB[64];
    // (the previous batch's solve + post-fit stage, see tail_work: this wave draws ONE ticket after tail_after rows --
    // a different count for every wave, spread over the first three quarters of its share, so that at any time a
    // few per cent of the waves are out of the transform instead of half of them for the first millisecond --
    // and whatever is left once it has run out of rows)
    int tail_after = 0x7fffffff;
    if (PP_TAIL_HOOKS && a.tail) {
        const unsigned share = (unsigned)(nrows / (long long)gridDim.x) + 1u;
        tail_after = 1 + (int)(((blockIdx.x * 2654435761u) >> 8) % (share * PP_TICKET_WINDOW_NUM / PP_TICKET_WINDOW_DEN + 1u));
#if PP_TICKET_SPREAD
        // (experiment: only as many waves ask as there are tickets -- every (grid / tickets)-th one --, so that the tickets
        // are spread over the first three quarters of the launch instead of being gone after its first fifth)
        {
            const unsigned nt = (unsigned)a.tail_nsub;
            const unsigned div = nt ? gridDim.x / nt : 1u;
            if (div > 1u && (((blockIdx.x * 2246822519u) >> 11) % div) != 0u) tail_after = 0x7fffffff;   // (hashed: blockIdx mod 8 is the XCD)
        }
#endif
    }
    RowWalk<true> rw;
    rw.start(nrows, a.mwords, a.ticket, a.ticket_base);
    long long row = rw.row;
    int n = 0, i = 0;
    if (rw.more) {
        n = __builtin_amdgcn_readfirstlane((int)(row / a.nsub));
        i = __builtin_amdgcn_readfirstlane((int)(row % a.nsub));
        const size_t rc = (size_t)sub_of(a.act, i) * a.nchan_full + (a.coff + n * a.cstep);
        stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rc * (2 * M), tid);
    }
    // this lane's template values (unhalved: the sums are halved at the end), reloaded
    // when the channel changes
    cplx mv2[NSL];
    const cplx* mheld = nullptr;
    const cplx* mrow = nullptr;    // the template row and cut of the channel in hand (channel_lookup)
    int n_held = -1, ktn = 0;
    int i_nx = i, n_nx = n;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
    if (phase == 1) {
        if (!(PP_TAIL_HOOKS && a.tail)) break;
        // (between two rows: the next row's loads are in flight, the call keeps what it must across itself)
        tail_work(a.tail, reinterpret_cast<double*>(lds), 2 * (LDSN + 64 * NML), tid, 1);
        tail_after = 0x7fffffff;
    }
    for (; rw.more && tail_after != 0; rw.advance(), row = rw.row, i = i_nx, n = n_nx, --tail_after) {
        rw.draw(a.ticket);
        rw.peek(nrows, a.ticket_base, a.mwords);
        // (everything derived from the lane number is recomputed per row: held across
        // the row it would cost the registers the prefetched row needs)
        asm volatile("" : "+v"(tid));
        const int lam = fftq_lambda(tid);
        const bool l0 = (lam == 0);
        const int kb = l0 ? 64 : lam;
        if (TWR) {
            t1 = as_global(a.twB)[2 * tid];
            t2 = as_global(a.twB)[32 * (tid & 15)];
            wb0 = as_global(a.twB)[kb];
        } else {
            asm volatile("" : "+v"(t1.x), "+v"(t1.y), "+v"(t2.x), "+v"(t2.y), "+v"(wb0.x), "+v"(wb0.y));
        }
        const int ia = sub_of(a.act, i), ne = a.coff + n * a.cstep;   // true subint, channel
        const size_t rc = (size_t)ia * a.nchan_full + ne;
        // loads whose results are needed late are issued before the prefetch (vector
        // memory returns in order)
        if (channel_lookup(a, ia, n, ne, M, n_held, mrow, ktn) && mrow != mheld) {
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                const cplx mval = mrow[kb + 64 * j - 1];   // k <= 448: inside the row
                if (j < NSL - NML) mv2[j] = mval; else ldsm[64 * (j - (NSL - NML))] = mval;
            }
            mheld = mrow;
        }
        const double phin = a.ph0[rc];
        double sd = 0.0;
        cplx v[R1];
#pragma unroll
        for (int k = 0; k < R1; ++k) v[k] = to_cplx(cur[0][k]);
        // the next row's HBM loads are queued as soon as this row's registers are dead
        // (unconditional: the last row of the run fetches itself again, see k_xspec)
        // f64 rows: in two halves -- eight registers' worth inside the first stage, the
        // rest once the second half of this row's outputs has been published: with the
        // whole next row in flight from the start, 64 + 64 row registers on top of the
        // template row and the sums do not fit the 256 of two waves per SIMD
        constexpr bool HALVES = Q_SPLIT_PREFETCH && sizeof(Tin) == 8;
        const Tin* nxrow = nullptr;
        auto load_some = [&](int k0, int k1) {
            const char* gb = reinterpret_cast<const char*>(nxrow);
            const unsigned boff = (unsigned)tid * (unsigned)sizeof(Raw);
#pragma unroll
            for (int k = 0; k < R1; ++k)
                if (k >= k0 && k < k1)
                    cur[0][k] = load_row_once<Raw>(gb + (size_t)(k * 64) * sizeof(Raw) + boff);
        };
        // (the row after this one is decided HERE, outside the lambda: a walk captured by reference is not
        // split into registers -- its flags went through scratch memory, whose loads queue behind the
        // prefetched row -- and at the top of a row everything older than this row's own data has landed,
        // the ticket of the chunk's first row included)
#ifndef PP_NEXT_IN_LAMBDA
        rw.next(i, n, i_nx, n_nx, nrows, a.nsub, a.ticket_base, a.ticket, a.mwords);
        {
            const size_t rn = rw.more_nx
                ? (size_t)sub_of(a.act, i_nx) * a.nchan_full + (a.coff + n_nx * a.cstep) : rc;
            nxrow = reinterpret_cast<const Tin*>(a.data) + rn * (2 * M);
        }
#endif
        auto prefetch = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#ifdef PP_NEXT_IN_LAMBDA
            rw.next(i, n, i_nx, n_nx, nrows, a.nsub, a.ticket_base, a.ticket, a.mwords);
            const size_t rn = rw.more_nx
                ? (size_t)sub_of(a.act, i_nx) * a.nchan_full + (a.coff + n_nx * a.cstep) : rc;
            nxrow = reinterpret_cast<const Tin*>(a.data) + rn * (2 * M);
#endif
            load_some(0, HALVES ? R1 / 2 : R1);
            __builtin_amdgcn_sched_barrier(0);
        };
        fftq1024<(sizeof(Tin) == 8 ? Q_PREFETCH_F64 : Q_PREFETCH_F32)>(v, lds, t1, t2, tid, &sd, prefetch);
        __builtin_amdgcn_sched_barrier(0);
        // ---- partners through LDS: registers 9..15 out, seven values back ----
        {
            cplx* pub = lds + tid;
            // (slot j reads the partner's register 15 - j = published piece 6 - j: only the pieces of the KEPT slots
            // are published -- one scalar jump into the run of stores)
            const int nk = PP_SLOT_PUBLISH_KEPT ? max(1, min(NSL, __builtin_amdgcn_readfirstlane(ktn) >> 6)) : NSL;
            switch (NSL - nk) {
                case 0: pub[64 * 0] = v[9];  [[fallthrough]];
                case 1: pub[64 * 1] = v[10]; [[fallthrough]];
                case 2: pub[64 * 2] = v[11]; [[fallthrough]];
                case 3: pub[64 * 3] = v[12]; [[fallthrough]];
                case 4: pub[64 * 4] = v[13]; [[fallthrough]];
                case 5: pub[64 * 5] = v[14]; [[fallthrough]];
                default: pub[64 * 6] = v[15];
            }
            static_assert(NSL == 7, "the run of stores above is written for seven slots");
            if (TAIL) {
#pragma unroll
                for (int r = 0; r < 5; ++r) pub[64 * (NSL + r)] = v[r];
            }
            lds_sync<T>();
        }
        double tail = 0.0;
        if (TAIL) {
            // |2 d_k|^2 for k = lam + 64 kd, kd = 12..15; W_B^k = W_B^kb0 W_B^(64 kd) with
            // kb0 = lam (the lane that owns lam = 0: kb = 64, one step ahead) and
            // W_B^768 = exp(-3 pi i / 4).  Partner: register 15 - kd of the partner lane
            // (own register 16 - kd for lam = 0).
            const double h = 0.70710678118654752440;
            const cplx* pt = lds + fftq_lane_of((64 - lam) & 63) + (l0 ? 64 : 0);
            // lam != 0: W^lam W^768;  lam = 0: wb0 = W^64, wanted W^768 = W^64 W^704 -> use W^768 directly
            cplx wt = l0 ? make_double2(-h, -h) : cmul(wb0, make_double2(-h, -h));
#pragma unroll
            for (int kd = 12; kd < 16; ++kd) {
                const cplx zk = v[kd];
                cplx zc = pt[64 * (NSL + 15 - kd)];
                zc.y = -zc.y;
                const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
                const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
                const cplx wo = cmul(wt, O);
                tail += cnorm(make_double2(E.x + wo.y, E.y - wo.x));
                wt = cmul(wt, wbT);
            }
            tail *= 0.25;
            if (tid == 0) { const double dM = v[0].x - v[0].y; tail += dM * dM; }
        }
        if (HALVES) {
            __builtin_amdgcn_sched_barrier(0);
            load_some(R1 / 2, R1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const cplx* pc = lds + fftq_lane_of((64 - lam) & 63);   // slot j: register 15 - j -> pc[64 (6 - j)]
        // ---- phasors: e^{2 pi i kb phi}; lane 0 (kb = 64) holds the step ----
        const cplx el = unit_phasor<true>((double)kb, phin);
        const cplx wst = make_double2(bcast_lane0(el.x), bcast_lane0(el.y));
        cplx e = el, wb = wb0;
        const int ktu = __builtin_amdgcn_readfirstlane(ktn);
        const double kap0 = PP_TWO_PI * (double)kb;
        double tm[PP_TSTRIDE];
        cplx zc_nx = pc[64 * 6];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            if (PP_SLOT_EARLY_EXIT && j > 0 && !(64 * j < ktu)) break;     // (kept slots are a prefix)
            cplx zc = zc_nx;
            if (j + 1 < NSL && (!PP_SLOT_EARLY_EXIT || 64 * (j + 1) < ktu)) zc_nx = pc[64 * (5 - j)];
            // the template cut is a multiple of 64: a slot is kept or dropped as a whole
            if (j == 0 || 64 * j < ktu) {
                const cplx zk = csel(l0, v[j + 1], v[j]);
                zc.y = -zc.y;
                const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
                const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
                const cplx wo = cmul(wb, O);
                // 2 d_k = E - i W^k O
                const cplx mj = (j < NSL - NML) ? mv2[j < NSL - NML ? j : 0] : ldsm[64 * (j - (NSL - NML))];
                const cplx x = cmulc(make_double2(E.x + wo.y, E.y - wo.x), mj);
                const cplx z = cmul(x, e);
                const double kap = j == 0 ? kap0 : kap0 + kconst<true>(PP_TWO_PI * (double)(64 * j));
                // kappa^2, ^4 .. ^10 once per harmonic; every sum is then one FMA
                const double p2 = kap * kap, p4 = p2 * p2, p6 = p4 * p2, p8 = p4 * p4, p10 = p8 * p2;
                const double ui = z.y * kap;
                const double ax = fabs(x.x) + fabs(x.y);
                if (j == 0) {
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
                } else {
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
                }
            }
            if (!PP_SLOT_EARLY_EXIT || (j + 1 < NSL && 64 * (j + 1) < ktu)) {
                wb = cmul(wb, wbT);
                e = cmul(e, wst);
            }
        }
        // ---- the 12 sums and S_d: one reduction through LDS ----
        double tr[NRED];
#pragma unroll
        for (int j = 0; j < PP_TSTRIDE; ++j) tr[j] = tm[j];
        tr[PP_TSTRIDE] = sd;
        if (TAIL) tr[NRED - 1] = tail;
        lds_sync<T>();      // (the partner reads are older than the reduction's writes)
        double tv = wave_reduce_lds(tr, tid, reinterpret_cast<double*>(lds));
        if ((tid & 3) == 0) {
            const int q = wave_reduce16_index(tid);
            if (q < PP_TSTRIDE) {
                // Re(i^q z): +Re, -Im, -Re, +Im, ...   (x 1/2: unhalved template against 2 d_k)
                tv *= 0.5;
                a.tay[tay_idx(rc, q)] = (q <= PP_TJ && ((q & 3) == 1 || (q & 3) == 2)) ? -tv : tv;
            }
        }
        if (tid == 4 * PP_TSTRIDE) a.sdraw[rc] = tv;
        if (TAIL && tid == 4 * (PP_TSTRIDE + 1)) {
            constexpr int H = M + 1, kc = (int)(0.75 * H);   // get_noise_PS: int((1 - 1/4) * len(pows))
            a.noise[rc] = sqrt(tv / (2.0 * M) / (double)(H - kc));
        }
        lds_sync<T>();
    }
    }
    // (out of rows: the tickets of the previous batch's tail that are left)
    if (PP_TAIL_HOOKS && a.tail) tail_work(a.tail, reinterpret_cast<double*>(lds), 2 * (LDSN + 64 * NML), tid, 1 << 30);
}


// --------------------------------------------------------------------------
// The same for templates that keep 512 harmonics or more (k_xspec's MODE 3; data-derived
// spline / PCA templates keep all 1024): every lane works on all 16 of its harmonics
// k = kb + 64 j, publishes all 16 registers and reads 16 partner values.  The lane that
// owns lam = 0 takes k = 64 (j + 1) from register (j + 1) & 15 -- its slot 15 is the Nyquist
// harmonic, for which the general split formula with Z_k = Z_{M-k} = Z_0 and W_B^M = -1
// gives 2 (Re Z_0 - Im Z_0) --, so one wave-uniform test keeps or drops a slot for every lane.
// The 16 template values of a lane (64 registers) cannot live beside two rows: they are
// read every row (L2 hits) between the transpose and the last stage, and the next row's
// loads are queued only after them, at the start of the split, in two halves (vector memory
// returns in order: template reads issued after the prefetch would wait for HBM).
// --------------------------------------------------------------------------
// TAIL (errs == NULL): harmonics 768..1024 are slots 12..15 (11..15 for the lane that owns
// lam = 0, whose slot 15 is the Nyquist harmonic); their split is formed whether or not the
// template keeps them.
// M = 512 (1024-bin rows, plan 8.4.2.8 of pp_fftq.h): the same with 8 harmonics per lane; serves every
// template cut (slots beyond it are skipped) -- configs[1]'s template keeps 448 of 512 harmonics.
#ifndef PP_QF512_HOLD_TEMPLATE
#define PP_QF512_HOLD_TEMPLATE 1
#endif
#ifndef PP_QF512_WPS
#define PP_QF512_WPS 3
#endif
template <int M, typename Tin, bool TAIL>
__global__ __launch_bounds__(64, (M == 1024 ? 2 : PP_QF512_WPS)) void k_xspec_qf(XspecArgs a) {
    typedef FftQ<M> Q;
    constexpr int T = 64, R1 = Q::R, PER1 = 1;
    constexpr int NSL = R1;
    constexpr int JT = (3 * NSL) / 4;        // first slot of the noise tail (k >= int(0.75 (M + 1)) = 64 JT)
    static_assert((int)(0.75 * (M + 1)) == 64 * JT, "the noise tail starts on a slot boundary");
    typedef typename RawOf<Tin>::type Raw;
    constexpr int NRED = PP_TSTRIDE + (TAIL ? 2 : 1);       // the 12 Taylor sums, S_d (and the noise tail)
    constexpr int WRED = PP_WRED_DOUBLES(NRED) / 2;   // in cplx
    constexpr int LDSN0 = Q::LDS_ELEMS > 64 * NSL ? Q::LDS_ELEMS : 64 * NSL;     // transpose image | published registers
    constexpr int LDSN = WRED > LDSN0 ? WRED : LDSN0;
    static_assert(M != 1024 || 2 * LDSN >= PP_TAIL_LDS_DOUBLES, "tail_work's layout of this kernel's LDS");
    __shared__ cplx lds[LDSN];
    int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    Raw cur[PER1][R1];
    const cplx wbT = a.twB[64];
    int tail_after = 0x7fffffff;         // (as k_xspec_q1024)
    if (PP_TAIL_HOOKS && M == 1024 && a.tail) {
        const unsigned share = (unsigned)(nrows / (long long)gridDim.x) + 1u;
        tail_after = 1 + (int)(((blockIdx.x * 2654435761u) >> 8) % (share * PP_TICKET_WINDOW_NUM / PP_TICKET_WINDOW_DEN + 1u));
    }
    RowWalk<true> rw;
    rw.start(nrows, a.mwords, a.ticket, a.ticket_base);
    long long row = rw.row;
    int n = 0, i = 0;
    if (rw.more) {
        n = __builtin_amdgcn_readfirstlane((int)(row / a.nsub));
        i = __builtin_amdgcn_readfirstlane((int)(row % a.nsub));
        const size_t rc = (size_t)sub_of(a.act, i) * a.nchan_full + (a.coff + n * a.cstep);
        stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rc * (2 * M), tid);
    }
    int i_nx = i, n_nx = n;
    const cplx* mrow = nullptr;    // the template row and cut of the channel in hand (channel_lookup)
    int n_held = -1, ktn = 0;
    // 1024-bin rows: the lane's 8 template values are HELD while the channel does not change (32 registers; the 16
    // values of a 2048-bin row's lane do not fit beside two rows and are read every row)
    constexpr bool MHOLD = PP_QF512_HOLD_TEMPLATE && M == 512;
    cplx mv[NSL];
    const cplx* mheld = nullptr;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
    if (phase == 1) {
        if (!(PP_TAIL_HOOKS && M == 1024 && a.tail)) break;
        tail_work(a.tail, reinterpret_cast<double*>(lds), 2 * LDSN, tid, 1);
        tail_after = 0x7fffffff;
    }
    for (; rw.more && tail_after != 0; rw.advance(), row = rw.row, i = i_nx, n = n_nx, --tail_after) {
        rw.draw(a.ticket);
        rw.peek(nrows, a.ticket_base, a.mwords);
        asm volatile("" : "+v"(tid));
        const int lam = Q::lambda(tid);
        const bool l0 = (lam == 0);
        const int kb = l0 ? 64 : lam;
        const cplx wb0 = as_global(a.twB)[kb];
        const int ia = sub_of(a.act, i), ne = a.coff + n * a.cstep;   // true subint, channel
        const size_t rc = (size_t)ia * a.nchan_full + ne;
        const bool looked = channel_lookup(a, ia, n, ne, M, n_held, mrow, ktn);
        if (MHOLD && looked && mrow != mheld) {
#pragma unroll
            for (int j = 0; j < NSL; ++j) mv[j] = mrow[kb + 64 * j - 1];
            mheld = mrow;
        }
        const double phin = a.ph0[rc];
        double sd = 0.0;
        cplx v[R1];
#pragma unroll
        for (int k = 0; k < R1; ++k) v[k] = to_cplx(cur[0][k]);
        // this lane's 16 template values, read between the transpose and the last stage
        auto template_loads = [&]() {
            if (MHOLD) return;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NSL; ++j) mv[j] = mrow[kb + 64 * j - 1];   // k <= 1024: inside the row
            __builtin_amdgcn_sched_barrier(0);
        };
        Q::template run<3>(v, lds, as_global(a.twB), tid, &sd, template_loads);
        __builtin_amdgcn_sched_barrier(0);
        // ---- partners through LDS: all 16 registers out, 16 values back ----
        {
            cplx* pub = lds + tid;
#pragma unroll
            for (int s = 0; s < NSL; ++s) pub[64 * s] = v[s];
            lds_sync<T>();
        }
        // ---- the next row: first half now (behind the template reads), second half after slot 7 ----
        const Tin* nxrow;
        auto load_some = [&](int k0, int k1) {
            const char* gb = reinterpret_cast<const char*>(nxrow);
            const unsigned boff = (unsigned)tid * (unsigned)sizeof(Raw);
#pragma unroll
            for (int k = 0; k < R1; ++k)
                if (k >= k0 && k < k1) cur[0][k] = load_row_once<Raw>(gb + (size_t)(k * 64) * sizeof(Raw) + boff);
        };
        {
            __builtin_amdgcn_sched_barrier(0);
            rw.next(i, n, i_nx, n_nx, nrows, a.nsub, a.ticket_base, a.ticket, a.mwords);
            const size_t rn = rw.more_nx
                ? (size_t)sub_of(a.act, i_nx) * a.nchan_full + (a.coff + n_nx * a.cstep) : rc;
            nxrow = reinterpret_cast<const Tin*>(a.data) + rn * (2 * M);
            load_some(0, R1 / 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        const cplx* pc = lds + Q::lane_of((64 - lam) & 63);   // slot j: register NSL - 1 - j of the partner
        const cplx el = unit_phasor<true>((double)kb, phin);
        const cplx wst = make_double2(bcast_lane0(el.x), bcast_lane0(el.y));
        cplx e = el, wb = wb0;
        const int ktu = __builtin_amdgcn_readfirstlane(ktn);
        const double kap0 = PP_TWO_PI * (double)kb;
        double tm[PP_TSTRIDE];
        double tail = 0.0;
        cplx zc_nx = pc[64 * (NSL - 1)];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            // (noise given: the kept slots are a prefix and nothing beyond them is needed.  The loop may only END once
            // the second half of the next row's loads has been queued, at slot NSL / 2 -- a copy of those loads at
            // every earlier exit cost the kernel its register allocation: 54 / 123 spilled VGPRs --; before that a
            // dropped slot just skips its body, its recurrences and its partner read)
            if (PP_SLOT_EARLY_EXIT && !TAIL && j > NSL / 2 && !(64 * j < ktu)) break;
            cplx zc = zc_nx;
            if (j + 1 < NSL && (!PP_SLOT_EARLY_EXIT || TAIL || 64 * (j + 1) < ktu)) zc_nx = pc[64 * (NSL - 2 - j)];
            if (j == NSL / 2) {
                __builtin_amdgcn_sched_barrier(0);
                load_some(R1 / 2, R1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the template cut is a multiple of 64: a slot is kept or dropped as a whole
            const bool keep = (j == 0 || 64 * j < ktu);
            if (keep || (TAIL && j >= JT - 1)) {
                const cplx zk = csel(l0, v[(j + 1) & (NSL - 1)], v[j]);
                zc.y = -zc.y;
                const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
                const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
                const cplx wo = cmul(wb, O);
                // 2 d_k = E - i W^k O
                const cplx dd = make_double2(E.x + wo.y, E.y - wo.x);
                if (TAIL && j >= JT - 1) {
                    const double pw = cnorm(dd);
                    tail += (j >= JT || l0) ? pw : 0.0;
                }
              if (keep) {
                const cplx x = cmulc(dd, mv[j]);
                const cplx z = cmul(x, e);
                const double kap = j == 0 ? kap0 : kap0 + kconst<true>(PP_TWO_PI * (double)(64 * j));
                const double p2 = kap * kap, p4 = p2 * p2, p6 = p4 * p2, p8 = p4 * p4, p10 = p8 * p2;
                const double ui = z.y * kap;
                const double ax = fabs(x.x) + fabs(x.y);
                if (j == 0) {
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
                } else {
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
                }
              }
            }
            if (!PP_SLOT_EARLY_EXIT || TAIL || (j + 1 < NSL && 64 * (j + 1) < ktu)) {
                wb = cmul(wb, wbT);
                e = cmul(e, wst);
            }
        }
        double tr[NRED];
#pragma unroll
        for (int j = 0; j < PP_TSTRIDE; ++j) tr[j] = tm[j];
        tr[PP_TSTRIDE] = sd;
        if (TAIL) tr[NRED - 1] = 0.25 * tail;
        lds_sync<T>();      // (the partner reads are older than the reduction's writes)
        double tv = wave_reduce_lds(tr, tid, reinterpret_cast<double*>(lds));
        if ((tid & 3) == 0) {
            const int q = wave_reduce16_index(tid);
            if (q < PP_TSTRIDE) {
                tv *= 0.5;
                a.tay[tay_idx(rc, q)] = (q <= PP_TJ && ((q & 3) == 1 || (q & 3) == 2)) ? -tv : tv;
            }
        }
        if (tid == 4 * PP_TSTRIDE) a.sdraw[rc] = tv;
        if (TAIL && tid == 4 * (PP_TSTRIDE + 1)) {
            constexpr int H = M + 1, kc = (int)(0.75 * H);   // get_noise_PS: int((1 - 1/4) * len(pows))
            a.noise[rc] = sqrt(tv / (2.0 * M) / (double)(H - kc));
        }
        lds_sync<T>();
    }
    }
    if (PP_TAIL_HOOKS && M == 1024 && a.tail) tail_work(a.tail, reinterpret_cast<double*>(lds), 2 * LDSN, tid, 1 << 30);
}

}  // namespace pp
