// k_xspec_qs1024: the transform of 2048-bin rows for SCATTERING fits -- the cross-spectrum of
// the kept harmonics is stored (the evaluation loop iterates over it) and the nine sums of the
// FIRST evaluation (pptoaslib.py:437-523 at the initial parameters) are taken while X_nk is in
// registers, so that evaluation is not a pass over the stored cross-spectrum.  Built on the
// one-exchange FFT (pp_fftq.h); same row walk, template row in registers and partner split as
// k_xspec_q1024 (template cut 2 Kt < M: slots k = kb + 64 j, j = 0..6).
//
// Per harmonic (k_eval_scat's arithmetic): z = X e^{i kap phi_n}, b = conj(B) = D (1 + i u),
// u = kap tau_n, D = 1 / (1 + u^2):
//   A0 = Re z b, A1 = -kap Im z b, A2 = -kap^2 Re z b, T1 = -kap Im z b^2, A1T = -kap^2 Re z b^2,
//   T2 = -2 kap^2 Re z b^3, S0 = D M, S1 = -2 kap u D^2 M, S2 = 2 kap^2 D^2 (4 u^2 D - 1) M.
#pragma once
#include "pp_xspec1024q.h"

namespace pp {

template <typename Tin>
__global__ __launch_bounds__(64, 2) void k_xspec_qs1024(XspecArgs a, const double* tau0, double* csum9) {
    constexpr int M = 1024, T = 64, R1 = 16, PER1 = 1;
    constexpr int NSL = 7;
    typedef typename RawOf<Tin>::type Raw;
    constexpr int NRED = PP_NCS + 1;          // the nine sums and S_d
    constexpr int WRED = PP_WRED_DOUBLES(NRED) / 2;
    constexpr int LDSN = WRED > FFTQ_LDS_ELEMS ? WRED : FFTQ_LDS_ELEMS;
    __shared__ cplx lds[LDSN];
    int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    Raw cur[PER1][R1];
    const cplx wbT = a.twB[64];
    RowWalk<true> rw;
    rw.start(nrows, a.mwords, a.ticket, a.ticket_base);
    long long row = rw.row;
    int n = 0, i = 0;
    if (rw.more) {
        n = __builtin_amdgcn_readfirstlane((int)(row / a.nsub));
        i = __builtin_amdgcn_readfirstlane((int)(row % a.nsub));
        const size_t rc = (size_t)i * a.nchan_full + n;
        stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rc * (2 * M), tid);
    }
    // this lane's template values, reloaded when the channel changes (|m_nk|^2 is formed from
    // them with the very operation that filled the table k_eval reads: same bits)
    cplx mv2[NSL];
    const cplx* mheld = nullptr;
    const cplx* mrow = nullptr;    // the template row and cut of the channel in hand (channel_lookup)
    int n_held = -1, ktn = 0;
    int i_nx = i, n_nx = n;
    for (; rw.more; rw.advance(), row = rw.row, i = i_nx, n = n_nx) {
        rw.draw(a.ticket);
        rw.peek(nrows, a.ticket_base, a.mwords);
        asm volatile("" : "+v"(tid));
        const int lam = fftq_lambda(tid);
        const bool l0 = (lam == 0);
        const int kb = l0 ? 64 : lam;
        const cplx t1 = as_global(a.twB)[2 * tid], t2 = as_global(a.twB)[32 * (tid & 15)];
        const cplx wb0 = as_global(a.twB)[kb];
        const size_t rc = (size_t)i * a.nchan_full + n;          // (no lists, no channel subsets here)
        if (channel_lookup(a, i, n, n, M, n_held, mrow, ktn) && mrow != mheld) {
#pragma unroll
            for (int j = 0; j < NSL; ++j) mv2[j] = mrow[kb + 64 * j - 1];
            mheld = mrow;
        }
        const double phin = a.ph0[rc], taun = tau0[rc];
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
        // (the row after this one is decided outside the lambda: see k_xspec_q1024)
        rw.next(i, n, i_nx, n_nx, nrows, a.nsub, a.ticket_base, a.ticket, a.mwords);
        {
            const size_t rn = rw.more_nx ? (size_t)i_nx * a.nchan_full + n_nx : rc;
            nxrow = reinterpret_cast<const Tin*>(a.data) + rn * (2 * M);
        }
        auto prefetch = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            load_some(0, HALVES ? R1 / 2 : R1);
            __builtin_amdgcn_sched_barrier(0);
        };
        fftq1024<(sizeof(Tin) == 8 ? Q_PREFETCH_F64 : Q_PREFETCH_F32)>(v, lds, t1, t2, tid, &sd, prefetch);
        __builtin_amdgcn_sched_barrier(0);
        {
            cplx* pub = lds + tid;
#pragma unroll
            for (int s = 0; s < NSL; ++s) pub[64 * s] = v[9 + s];
            lds_sync<T>();
        }
        if (HALVES) {
            __builtin_amdgcn_sched_barrier(0);
            load_some(R1 / 2, R1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const cplx* pc = lds + fftq_lane_of((64 - lam) & 63);
        const cplx el = unit_phasor<true>((double)kb, phin);
        const cplx wst = make_double2(bcast_lane0(el.x), bcast_lane0(el.y));
        cplx e = el, wb = wb0;
        const int ktu = __builtin_amdgcn_readfirstlane(ktn);
        const double kap0 = PP_TWO_PI * (double)kb;
        double s0 = 0, s1 = 0, s2 = 0, q1 = 0, q2 = 0, a1t = 0, S0 = 0, S1 = 0, S2 = 0;
        cplx zc_nx = pc[64 * 6];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            if (PP_SLOT_EARLY_EXIT && j > 0 && !(64 * j < ktu)) break;     // (kept slots are a prefix: pp_xspec1024q.h)
            cplx zc = zc_nx;
            if (j + 1 < NSL && (!PP_SLOT_EARLY_EXIT || 64 * (j + 1) < ktu)) zc_nx = pc[64 * (5 - j)];
            if (j == 0 || 64 * j < ktu) {
                const cplx zk = csel(l0, v[j + 1], v[j]);
                zc.y = -zc.y;
                const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
                const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
                const cplx wo = cmul(wb, O);
                // X_k = d_k conj(m_k) with 2 d_k = E - i W^k O (the half is exact)
                cplx x = cmulc(make_double2(E.x + wo.y, E.y - wo.x), mv2[j]);
                x.x *= 0.5; x.y *= 0.5;
                store_x(a, rc, kb + 64 * j, x);
                const cplx z = cmul(x, e);
                const double kap = j == 0 ? kap0 : kap0 + kconst<true>(PP_TWO_PI * (double)(64 * j));
                const double u = kap * taun;
                const double D = recip_ge1(fma(u, u, 1.0));
                const cplx b = make_double2(D, u * D);
                const cplx zb = cmul(z, b);
                const double k2 = kap * kap, Mk = cnorm(mv2[j]);
                s0 += zb.x;
                s1 = fma(kap, zb.y, s1);
                s2 = fma(k2, zb.x, s2);
                S0 = fma(D, Mk, S0);
                const cplx zb2 = cmul(zb, b);
                const cplx zb3 = cmul(zb2, b);
                q1 = fma(kap, zb2.y, q1);
                a1t = fma(k2, zb2.x, a1t);
                q2 = fma(k2, zb3.x, q2);
                const double D2 = D * D;
                S1 = fma(kap * u * D2, Mk, S1);
                S2 = fma(k2 * D2 * fma(4.0 * u * u, D, -1.0), Mk, S2);
            }
            if (!PP_SLOT_EARLY_EXIT || (j + 1 < NSL && 64 * (j + 1) < ktu)) {
                wb = cmul(wb, wbT);
                e = cmul(e, wst);
            }
        }
        double tr[NRED] = {s0, -s1, -s2, -q1, -2.0 * q2, -a1t, S0, -2.0 * S1, 2.0 * S2, sd};
        lds_sync<T>();
        const double tv = wave_reduce_lds(tr, tid, reinterpret_cast<double*>(lds));
        if ((tid & 3) == 0) {
            const int q = wave_reduce16_index(tid);
            if (q < PP_NCS) csum9[rc * PP_NCS + q] = tv;
            else if (q == PP_NCS) a.sdraw[rc] = tv;
        }
        lds_sync<T>();
    }
}

}  // namespace pp
