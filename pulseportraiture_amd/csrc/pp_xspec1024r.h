// k_xspec_qr1024: the transform of 2048-bin rows (k_xspec_q1024: one-exchange FFT, the 12
// Taylor sums of every channel, nothing stored) that ALSO leaves what the reference's own
// initial phase guess needs, so that get_TOAs' default flow reads the portraits once:
//
//   pptoas.py:421-457   rot_prof   = average(rotate_data(portx, 0, DM_guess, P, freqsx, nu_mean),
//                                            axis=0, weights=weightsx)
//                       phi_guess  = fit_phase_shift(rot_prof, model_prof, Ns=100).phase
//
// rot_prof is a sum over the channels of a subint, phi_guess is only known after it -- and
// the Taylor model wants its expansion point BEFORE the pass.  But the model holds within
// ~1e-3 rot of its centre, so any centre phi_c that close to phi_guess will do: the pilot
// seed (a pass over every 16th channel, pp_toas.hip) supplies it, this pass expands about
// (phi_c, DM_guess), and the solve then starts SciPy's walk from the reference's
// (phi_guess, DM_guess) as a displacement from that centre (k_taylor_solve, FitArgs::xstart).
//
// The rotated spectrum needs no second phasor: with phi_n = phi_c + DM p1_n (the Taylor
// phase) and phi'_n = Dconst DM (nu_n^-2 - nu_mean^-2) / P (the reference's rotation),
// phi_n - phi'_n = Delta is the same for every channel of the subint, so
//   R_k = sum_n w_n d_nk e^{i kap phi'_n} = e^{-i kap Delta} sum_n w_n d_nk e^{i kap phi_n}
// and the per-harmonic factor is applied once per subint afterwards (k_refseed_finish).
// fit_phase_shift measures its noise from the top quarter of rot_prof's power spectrum
// (that decides where SciPy's simplex stops), so besides the harmonics the template keeps
// (k <= 448: slots 0..6) the channel sum is also taken for k = 768..1024 (slots 12..15 and
// the Nyquist term): 12 complex accumulators per lane, in registers.
//
// Rows are dealt in chunks of PP_ROW_CHUNK consecutive CHANNELS of one subint (chunk c =
// channel block c / nsub of subint c % nsub: the waves in flight work on the same few channel
// blocks of different subints, so the template rows they read stay in L2); a chunk's
// accumulators are written out as one partial sum and k_refseed_finish adds the partials of
// a subint in a fixed order (deterministic).  The template row is read per row (L2), after
// the transform and before the second half of the next row's prefetch.
#pragma once
#include "pp_xspec1024q.h"

namespace pp {

struct RefSeedArgs {
    const double* w;      // [nsub][nchan_full] channel weights of the mean, or nullptr (= 1)
    cplx* part;           // [nsub][ncc][RS_NACC][64] partial channel sums
    int ncc;              // channel blocks per subint (nchan / PP_ROW_CHUNK)
};
constexpr int RS_NACC = 12;     // slots 0..6 (kept), 12..15 (noise tail), Nyquist (lane of lam = 0)
constexpr int RS_NREG = 11;     // ... of which in registers; the Nyquist term (one lane's) sits in LDS
constexpr int RS_NYQ = FFTQ_LDS_ELEMS - 1;    // ... in the one element of the transpose image no lane touches
                                              // (highest index used: 3 * 272 + 17 * 15 + 15 = 1086)

// STORE (scattering fits): no Taylor model -- the evaluation loop iterates over the stored
// cross-spectrum, so X_nk of the kept harmonics is stored instead (as k_xspec's MODE 0); the
// phase a.ph0 then holds is the rotation alone (phase guess 0: the pass needs no centre, hence
// no pilot) and the iteration starts at the reference's guess with an ordinary first evaluation.
template <typename Tin, bool STORE = false>
__global__ __launch_bounds__(64, 2) void k_xspec_qr1024(XspecArgs a, RefSeedArgs rs) {
    constexpr int M = 1024, T = 64, R1 = 16, PER1 = 1;
    constexpr int NSL = 7;
    typedef typename RawOf<Tin>::type Raw;
    constexpr int NRED = PP_TSTRIDE + 1;
    static_assert(PP_TJ == 10, "power ladder written for order 10");
    constexpr int WRED = PP_WRED_DOUBLES(NRED) / 2;
    constexpr int LDSN = WRED > FFTQ_LDS_ELEMS ? WRED : FFTQ_LDS_ELEMS;
    // f64 rows: three of the four tail accumulators live in the LDS a wave has left beside its
    // image (8 waves x 20 KB = the CU's 160 KB) -- with a whole f64 row in flight (64 registers)
    // there is no room for them in the register file
    // (f32 rows that carry tickets: two of them -- the call's bookkeeping costs the row loop the registers of one
    // accumulator, which otherwise goes to scratch memory and is read, updated and written back every row)
    constexpr int NLA = (sizeof(Tin) == 8) ? 3 : ((PP_TAIL_HOOKS && !STORE) ? 2 : 0);
    __shared__ cplx lds[LDSN + 64 * NLA];
    int tid = threadIdx.x;
    cplx* const lacc = lds + LDSN + threadIdx.x;
    // rows = (chunk, position in chunk); RowWalk deals chunks, its "subint" is the position
    const long long nrows = (long long)a.nsub * a.nchan;
    Raw cur[PER1][R1];
    // W_2048^64 = exp(-i pi / 16), the step of the split twiddle from one slot to the next: the table's own value
    // (a.twB[64], the correctly rounded cosine and sine), as scalar-register constants formed where they are used --
    // held in four vector registers through the row loop it is what the kernel lacks once it carries tickets
    // (a prefetched piece of the next row went to scratch memory instead, behind a full vmcnt(0) wait)
#define PP_WBT make_double2(kconst<true>(0x1.f6297cff75cb0p-1), kconst<true>(-0x1.8f8b83c69a60bp-3))
    RowWalk<true> rw;
    rw.start(nrows, a.mwords, a.ticket, a.ticket_base);
    // (position in the chunk and chunk are the low and high bits of the walk's row: nothing to carry)
    int ia = 0, cc = 0;               // the chunk's subint and channel block (one division per chunk)
    if (rw.more) {
        const int c0 = (int)(rw.row >> 5), r0 = (int)(rw.row & 31u);    // (r0 = 0 unless the mask removes the chunk's first rows)
        ia = c0 % a.nsub; cc = c0 / a.nsub;
        const size_t rc = (size_t)ia * a.nchan_full + (size_t)cc * PP_ROW_CHUNK + r0;
        stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rc * (2 * M), tid);
    }
    cplx acc[RS_NREG - NLA];      // (slots 0..6, tail slot 12 [.. 15 for f32 rows])
    int ia_nx = ia, cc_nx = cc;
    const int ktg = a.Kt;             // harmonics the widest template row keeps: the channel sum takes them all
    // (the previous batch's tail -- its phase guess, solve and post-fit stage, pp_tail.h: this wave draws ONE ticket
    // after tail_after rows, at the next chunk boundary -- there the channel sums in hand have just been written out
    // and the ticket may use the whole image --, a different count for every wave as in k_xspec_q1024; what is left
    // is drawn by the waves that have run out of rows)
    constexpr bool HOOK = PP_TAIL_HOOKS && !STORE;
    static_assert(!HOOK || 2 * LDSN >= PP_TAIL_LDS_DOUBLES, "tail_work's layout of this kernel's LDS");
    // (the only state the hook carries through the row loop is `phase` -- this kernel has neither a vector nor a
    // scalar register to spare: a row count per wave, as k_xspec_q1024 keeps, cost it a spilled accumulator per row.
    // The moment is read off the walk instead: chunks are dealt in order, so the chunk number IS the launch's
    // progress, and a wave asks for its ticket at the first chunk boundary past a wave-specific threshold spread
    // over the first three quarters of the chunks)
    int phase = (HOOK && a.tail) ? 0 : 2;          // 0: ticket not asked for yet, 2: asked (or nothing to carry)
    auto due = [&]() -> bool {
        const unsigned nchunks = ((unsigned)nrows + PP_ROW_CHUNK - 1u) / PP_ROW_CHUNK;
        const unsigned thr = ((blockIdx.x * 2654435761u) >> 8) % (nchunks - nchunks / 4u + 1u);
        return (rw.row >> 5) >= thr;
    };
    for (;;) {
    for (; rw.more && !(HOOK && phase == 0 && rw.fresh && due()); rw.advance(), ia = ia_nx, cc = cc_nx) {
        rw.draw(a.ticket);
        // (no look-ahead for the next chunk's word here -- scalar registers are what this kernel is
        // short of: the word is fetched when the chunk runs out, one exposed scalar load per 32 rows)
        const int r = (int)(rw.row & 31u);
        asm volatile("" : "+v"(tid));
        const int lam = fftq_lambda(tid);
        const bool l0 = (lam == 0);
        const int kb = l0 ? 64 : lam;
        const cplx t1 = as_global(a.twB)[2 * tid], t2 = as_global(a.twB)[32 * (tid & 15)];
        const cplx wb0 = as_global(a.twB)[kb];
        const int ne = cc * PP_ROW_CHUNK + r;
        const size_t rc = (size_t)ia * a.nchan_full + ne;
        // (see channel_lookup: a vector load read at the top of a row costs a memory latency -- the channel changes
        // with every row here, so with ONE template for all subints the cut comes through the scalar unit and the row
        // pointer is arithmetic; per-subint templates are looked up per row as before.  Nothing is carried over the
        // rows: this kernel has no register to spare)
        const cplx* mrow;
        int ktn;
        if (PP_STICKY_LOOKUP && !a.slot) {
            mrow = as_global(a.mft0) + (size_t)ne * M;
            ktn = a.ktab ? load_uniform(a.kt0 + ne) : a.Kt;
        } else {
            mrow = as_global(a.slot ? a.mft[a.slot[ia]] : a.mft0) + (size_t)ne * M;
            ktn = a.ktab ? as_global(a.slot ? a.ktab[a.slot[ia]] : a.kt0)[ne] : a.Kt;
        }
        const double phin = a.ph0[rc];
        const double hw = 0.5 * (rs.w ? rs.w[rc] : 1.0);     // (2 d_k below: the half goes here, exact)
        if (rw.fresh) {       // (the first row visited of this chunk)
#pragma unroll
            for (int j = 0; j < RS_NREG - NLA; ++j) acc[j] = make_double2(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < NLA; ++j) lacc[64 * j] = make_double2(0.0, 0.0);
            if (tid == 0) lds[RS_NYQ] = make_double2(0.0, 0.0);
        }
        double sd = 0.0;
        cplx v[R1];
#pragma unroll
        for (int k = 0; k < R1; ++k) v[k] = to_cplx(cur[0][k]);
        constexpr bool HALVES = sizeof(Tin) == 8;
        const Tin* nxrow = nullptr;
        auto load_some = [&](int k0, int k1) {
            const char* gb = reinterpret_cast<const char*>(nxrow);
            const unsigned boff = (unsigned)tid * (unsigned)sizeof(Raw);
#pragma unroll
            for (int k = 0; k < R1; ++k)
                if (k >= k0 && k < k1)
                    cur[0][k] = load_row_once<Raw>(gb + (size_t)(k * 64) * sizeof(Raw) + boff);
        };
        // (the row after this one is decided here, outside the lambda: see k_xspec_q1024)
        rw.next_row(nrows, a.ticket_base, a.ticket, a.mwords);
        {
            size_t rn = rc;
            ia_nx = ia; cc_nx = cc;
            if (rw.more_nx) {
                const int c_nx = (int)(rw.row_nx >> 5);
                if (c_nx != (int)(rw.row >> 5)) { ia_nx = c_nx % a.nsub; cc_nx = c_nx / a.nsub; }
                rn = (size_t)ia_nx * a.nchan_full + (size_t)(cc_nx * PP_ROW_CHUNK + (int)(rw.row_nx & 31u));
            }
            nxrow = reinterpret_cast<const Tin*>(a.data) + rn * (2 * M);
        }
        const bool last_of_chunk = !rw.more_nx || ((rw.row_nx ^ rw.row) >> 5) != 0u;
        auto prefetch = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            load_some(0, HALVES ? R1 / 2 : R1);
            __builtin_amdgcn_sched_barrier(0);
        };
        fftq1024<(sizeof(Tin) == 8 ? Q_PREFETCH_F64 : Q_PREFETCH_F32)>(v, lds, t1, t2, tid, &sd, prefetch);
        __builtin_amdgcn_sched_barrier(0);
        // this row's template values: read now (L2), behind the first half of the prefetch --
        // which has had the whole transform to arrive -- and in front of the second half
        // (round 5: queued before the last stage instead, one stage earlier: no difference, profiles/r05_qr_ab.txt)
        cplx mv2[NSL];
#pragma unroll
        for (int j = 0; j < NSL; ++j) mv2[j] = mrow[kb + 64 * j - 1];
        // ---- partners through LDS: registers 9..15 and 0..4 out ----
        {
            cplx* pub = lds + tid;
#pragma unroll
            for (int s = 0; s < NSL; ++s) pub[64 * s] = v[9 + s];
#pragma unroll
            for (int q = 0; q < 5; ++q) pub[64 * (NSL + q)] = v[q];
            lds_sync<T>();
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
            cplx zc = zc_nx;
            if (j + 1 < NSL) zc_nx = pc[64 * (5 - j)];
            if (HALVES && j == NSL - 1) {
                // the second half of the next row: queued before the last kept slot (earlier, the
                // 64 registers of a whole f64 row in flight do not fit beside the accumulators)
                __builtin_amdgcn_sched_barrier(0);
                load_some(R1 / 2, R1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (the channel sum takes every harmonic ANY template row keeps; the Taylor sums only
            // those this channel's row keeps -- both cuts are multiples of 64, wave-uniform)
            if (j == 0 || 64 * j < ktg) {
                const cplx zk = csel(l0, v[j + 1], v[j]);
                zc.y = -zc.y;
                const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
                const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
                const cplx wo = cmul(wb, O);
                const cplx dd = make_double2(E.x + wo.y, E.y - wo.x);     // 2 d_k = E - i W^k O
                // the rotated channel sum: w_n d_k e^{i kap phi_n}
                const cplx y = cmul(dd, e);
                acc[j].x = fma(hw, y.x, acc[j].x);
                acc[j].y = fma(hw, y.y, acc[j].y);
              if (STORE) {
                if (j == 0 || 64 * j < ktu) {
                    // X_k = d_k conj(m_k) (the half of 2 d_k is exact), as k_xspec's MODE 0 stores it
                    cplx x = cmulc(dd, mv2[j]);
                    x.x *= 0.5; x.y *= 0.5;
                    store_x(a, rc, kb + 64 * j, x);
                }
              } else if (j == 0 || 64 * j < ktu) {
                // the cross-spectrum's Taylor sums, exactly as k_xspec_q1024 forms them
                const cplx x = cmulc(dd, mv2[j]);
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
            wb = cmul(wb, PP_WBT);
            e = cmul(e, wst);
        }
        // ---- the noise tail of the channel sum: k = lam + 64 kd, kd = 12..15 (lam = 0: k = 64 kd,
        // partner = own register 16 - kd) and the Nyquist term.  After the loop above
        // e = e^{2 pi i (kb + 448) phi}, wb = W^(kb + 448): five more steps reach kd = 12
        // (lam = 0: kb = 64, one step ahead -- four steps).
        {
            cplx e4 = cmul(wst, wst);
            e4 = cmul(e4, e4);                       // e^{2 pi i 256 phi}
            const cplx w4 = make_double2(0.70710678118654752440, -0.70710678118654752440);   // W_2048^256
            cplx et = cmul(e, e4), wt = cmul(wb, w4);                 // + 4 steps
            const cplx et1 = cmul(et, wst), wt1 = cmul(wt, PP_WBT);      // + 5 steps
            et = csel(l0, et, et1);
            wt = csel(l0, wt, wt1);
            const cplx* pt = lds + fftq_lane_of((64 - lam) & 63) + (l0 ? 64 : 0);
#pragma unroll
            for (int kd = 12; kd < 16; ++kd) {
                const cplx zk = v[kd];
                cplx zc = pt[64 * (NSL + 15 - kd)];
                zc.y = -zc.y;
                const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
                const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
                const cplx wo = cmul(wt, O);
                const cplx y = cmul(make_double2(E.x + wo.y, E.y - wo.x), et);
                constexpr int NRT = 4 - NLA;      // tail accumulators in registers
                if (kd - 12 < NRT) {
                    acc[7 + kd - 12].x = fma(hw, y.x, acc[7 + kd - 12].x);
                    acc[7 + kd - 12].y = fma(hw, y.y, acc[7 + kd - 12].y);
                } else {
                    cplx t = lacc[64 * (kd - 12 - NRT)];
                    t.x = fma(hw, y.x, t.x);
                    t.y = fma(hw, y.y, t.y);
                    lacc[64 * (kd - 12 - NRT)] = t;
                }
                wt = cmul(wt, PP_WBT);
                et = cmul(et, wst);
            }
            if (tid == 0) {
                const double dM = 2.0 * (v[0].x - v[0].y);
                cplx t = lds[RS_NYQ];
                t.x = fma(hw * dM, et.x, t.x);
                t.y = fma(hw * dM, et.y, t.y);
                lds[RS_NYQ] = t;
            }
        }
        if constexpr (STORE) {
            sd = group_sum<64>(sd);
            if (tid == 0) a.sdraw[rc] = sd;
            lds_sync<T>();
        } else {
            // ---- the 12 sums and S_d: one reduction through LDS ----
            double tr[NRED];
    #pragma unroll
            for (int j = 0; j < PP_TSTRIDE; ++j) tr[j] = tm[j];
            tr[PP_TSTRIDE] = sd;
            lds_sync<T>();
            double tv = wave_reduce_lds(tr, tid, reinterpret_cast<double*>(lds));
            if ((tid & 3) == 0) {
                const int q = wave_reduce16_index(tid);
                if (q < PP_TSTRIDE) {
                    tv *= 0.5;
                    a.tay[tay_idx(rc, q)] = (q <= PP_TJ && ((q & 3) == 1 || (q & 3) == 2)) ? -tv : tv;
                }
            }
            if (tid == 4 * PP_TSTRIDE) a.sdraw[rc] = tv;
        }
        if (last_of_chunk) {
            // (the last row visited of this chunk:) the chunk's share of the channel sums of subint ia
            cplx* out = rs.part + (((size_t)ia * rs.ncc + cc) * RS_NACC) * 64 + tid;
#pragma unroll
            for (int j = 0; j < RS_NREG - NLA; ++j) out[64 * j] = acc[j];
#pragma unroll
            for (int j = 0; j < NLA; ++j) out[64 * (RS_NREG - NLA + j)] = lacc[64 * j];
            if (tid == 0) out[64 * RS_NREG] = lds[RS_NYQ];
        }
        lds_sync<T>();
    }
    if (!HOOK || phase != 0 || !rw.more) break;
    tail_work(a.tail, reinterpret_cast<double*>(lds), 2 * LDSN, tid, 1);
    phase = 2;
    {
        // (the row prefetched before the ticket is fetched AGAIN: 64 registers that need not live across the call --
        // kept, one piece of them was given a scratch slot for the whole kernel and every row waited for it with
        // vmcnt(0); one row more per wave and launch, 0.1 % of the traffic)
        const size_t rc = (size_t)ia * a.nchan_full + (size_t)cc * PP_ROW_CHUNK + (size_t)(rw.row & 31u);
        stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rc * (2 * M), tid);
    }
    }
    if (HOOK && a.tail) tail_work(a.tail, reinterpret_cast<double*>(lds), 2 * LDSN, tid, 1 << 30);
}

// the weights of the channel mean with the fit's channel mask folded in: the mean is taken over the
// channels in use only (the reference averages portx = port[ok_ichans], pptoas.py:384-397, 424),
// whatever weight a masked channel carries
__global__ void k_refseed_weights(const double* w, const unsigned char* mask, long long n, double* out) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = mask[j] ? (w ? w[j] : 1.0) : 0.0;
}

// per subint: Delta_i = phi_c + Dconst DM (nu_mean^-2 - nu_fit^-2) / P, the constant by which the
// Taylor phase of every channel exceeds the reference's rotation phase, and the summed weights
// (fixed-order block reduction).  x0: [nsub][5] expansion points (the pilot seed wrote the phases).
__global__ __launch_bounds__(256) void k_refseed_prep(const double* x0, const double* P, const double* nu_fit,
                                                      const double* nu_mean, const double* w, int nchan,
                                                      double* delta, double* wsum) {
    const int i = blockIdx.x, tid = threadIdx.x;
    __shared__ double scratch[4];
    double s[1] = {0.0};
    if (w) { for (int n = tid; n < nchan; n += 256) s[0] += w[(size_t)i * nchan + n]; }
    else if (tid == 0) s[0] = (double)nchan;
    block_sum<1>(s, scratch);
    if (tid == 0) {
        const double nf = nu_fit[i * 3], nm = nu_mean[i];
        delta[i] = x0[i * 5] + PP_DCONST * x0[i * 5 + 1] * (1.0 / (nm * nm) - 1.0 / (nf * nf)) / P[i];
        wsum[i] = s[0];
    }
}

// the start points of the iteration: xs[i] holds {K_i, DM, GM, tau, alpha} on entry, K_i the term
// phase_transform adds (formed on the host from the inputs alone); the phase becomes
// wrap(fit_phase_shift's phase + K_i) with NumPy's two-step wrap to [-0.5, 0.5) (pplib.py:2612-2613)
__device__ __forceinline__ void refseed_start_one(const double* out7, const int i, double* xs, double* seed_phase) {
    double ph = out7[(size_t)i * 7] + xs[(size_t)i * 5];
    if (fabs(ph) >= 0.5) { ph = fmod(ph, 1.0); if (ph != 0.0 && ph < 0.0) ph += 1.0; }
    if (ph >= 0.5) ph -= 1.0;
    xs[(size_t)i * 5] = ph;
    seed_phase[i] = ph;
}
__global__ void k_refseed_start(const double* out7, int nsub, double* xs, double* seed_phase) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsub) return;
    refseed_start_one(out7, i, xs, seed_phase);
}

// rot_prof's spectrum of every subint from the partial channel sums: fixed-order sum over the
// channel blocks, the per-harmonic factor e^{-i kap Delta_i} that turns the Taylor phase into the
// reference's rotation phase, division by the summed weights; harmonics the pass did not
// accumulate (448 < k < 768, and 0) are zero.  spec[i][0..M].
// mws (optional): the rows-in-use words of the pass (k_mask_words' wsub): a chunk without a row in use
// was never visited and left no partial.
// harmonic k of subint i's rot_prof spectrum (0 <= k <= 1024), from the chunk partials of the pass
__device__ __forceinline__ cplx refseed_spec_value(const cplx* part, const int ncc, const double delta_i, const double wsum_i,
                                                   const int nsub, const int i, const int k, const unsigned* mws) {
    constexpr int M = 1024;
    const int lam = k & 63, kd = k >> 6;
    int slot = -1;
    if (k == M) slot = 11;
    else if (lam != 0) slot = (kd <= 6) ? kd : (kd >= 12 ? 7 + kd - 12 : -1);
    else slot = (kd >= 1 && kd <= 7) ? kd - 1 : (kd >= 12 ? 7 + kd - 12 : -1);
    cplx s = make_double2(0.0, 0.0);
    if (slot >= 0) {
        const int lane = fftq_lane_of(k == M ? 0 : lam);
        for (int cc = 0; cc < ncc; ++cc) {
            if (mws && mws[(size_t)cc * nsub + i] == 0u) continue;
            const cplx v = part[(((size_t)i * ncc + cc) * RS_NACC + slot) * 64 + lane];
            s.x += v.x; s.y += v.y;
        }
        const cplx rot = unit_phasor((double)k, -delta_i);
        s = cmul(s, rot);
        const double inv = wsum_i > 0.0 ? 1.0 / wsum_i : 0.0;
        s.x *= inv; s.y *= inv;
        if (k == M) s.y = 0.0;          // (irfft keeps the real part of the Nyquist term only)
    }
    return s;
}

__global__ __launch_bounds__(256) void k_refseed_finish(const cplx* part, int ncc, const double* delta,
                                                        const double* wsum, int nsub, cplx* spec, const unsigned* mws) {
    constexpr int M = 1024;
    const int i = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k > M) return;
    spec[(size_t)i * (M + 1) + k] = refseed_spec_value(part, ncc, delta[i], wsum[i], nsub, i, k, mws);
}

#undef PP_WBT
}  // namespace pp
