// k_xspec for 2048-bin rows (M = 1024 packed complex points, one wave per row)
// whose template keeps fewer than M/2 harmonics -- the benchmark shape.
//
// Same contract as k_xspec (pp_kernels.h: modes 0 / 1 / 2, TAIL) with the LAST
// Stockham stage and the even/odd split done in registers: the stage-3 outputs
// are Z[t + 128 j] (t = butterfly, j = 0..7) and the split needs the pairs
// (k, M - k), i.e. butterflies (t, 128 - t).  Lane l therefore takes butterflies
// l and 128 - l (lane 0: the two self-paired ones, 0 and 64): every pair of the
// split then sits in one lane's registers, and the row needs neither the 16
// ds_write_b128 that parked the transform in LDS nor the ds_read_b128 pairs that
// fetched it back (a third of the row's LDS traffic, which the counters and the
// store-doubling experiment of profiles/README.md show to add to the kernel's time
// one for one).
//
// Lane l owns harmonics  kA(j) = l + 128 j  (Z = va[j], partner vb[7-j]) and
//                        kB(j) = (128 - l) + 128 j  (Z = vb[j], partner va[7-j]);
// lane 0:  kA(j) = 128 j (partner va[8-j]),  kB(j) = 64 + 128 j (partner vb[7-j]).
// Split twiddles W_B^k = W_B^l * W_B^(128 j): the second factor is exp(-i pi j/8), a
// compile-time constant; phasors advance by e^{2 pi i 128 phi}.
#pragma once

namespace pp {

__device__ __forceinline__ double bcast_lane0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ cplx csel(bool c, const cplx& a, const cplx& b) {
    return make_double2(c ? a.x : b.x, c ? a.y : b.y);
}

// W_B^(128 j) for B = 2048: exp(-i pi j / 8)
__device__ __forceinline__ cplx w128(int j) {
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, h = 0.70710678118654752440;
    switch (j & 7) {
        case 0: return make_double2(1.0, 0.0);
        case 1: return make_double2(c1, -s1);
        case 2: return make_double2(h, -h);
        case 3: return make_double2(s1, -c1);
        case 4: return make_double2(0.0, -1.0);
        case 5: return make_double2(-s1, -c1);
        case 6: return make_double2(-h, -h);
        default: return make_double2(-c1, -s1);
    }
}

template <typename Tin, bool TAIL, int MODE>
__global__ __launch_bounds__(64, 2) void k_xspec_p1024(XspecArgs a) {
    constexpr int M = 1024, T = 64;
    typedef FftPlan<M> P;
    static_assert(P::T == 64 && P::R1 == 16 && P::R2 == 8 && P::R3 == 8 && P::R4 == 1, "plan 16.8.8 expected");
    constexpr int R1 = P::R1, PER1 = P::PER1, PL = P::PADLOG;
    constexpr bool M2 = (MODE == 2), FUSE = (MODE != 0);
    constexpr int NS = 4;                      // slots of each kind: 2 Kt < M  ->  k <= 448
    typedef typename RawOf<Tin>::type Raw;
    constexpr int WRED = PP_WRED_DOUBLES(PP_TSTRIDE) / 2;   // in cplx
    constexpr int LDSN = (M2 && WRED > P::LDS_ELEMS) ? WRED : P::LDS_ELEMS;
    __shared__ cplx lds[LDSN];
    int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    Raw cur[PER1][R1];
    RowWalk<true> rw;
    rw.start(nrows, a.mwords, a.ticket, a.ticket_base);
    long long row = rw.row;
    int n = 0, i = 0;
    if (rw.more) {
        n = __builtin_amdgcn_readfirstlane((int)(row / a.nsub));
        i = __builtin_amdgcn_readfirstlane((int)(row % a.nsub));
        const size_t rc = (size_t)i * a.nchan + n;
        stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rc * (2 * M), tid);
    }
    // this lane's template values (halved), reloaded when the channel changes
    cplx mA[NS], mB[NS];
    const cplx* mheld = nullptr;
    const cplx* mrow = nullptr;    // the template row and cut of the channel in hand (channel_lookup)
    int n_held = -1, ktn = 0;
    int i_nx = i, n_nx = n;
    for (; rw.more; rw.advance(), row = rw.row, i = i_nx, n = n_nx) {
        rw.draw(a.ticket);
        rw.peek(nrows, a.ticket_base, a.mwords);
        if (PP_OPAQUE_ROW == 1 || (PP_OPAQUE_ROW == 2 && M2)) asm volatile("" : "+v"(tid));
        // stage twiddles are re-read every row (three L1-resident loads, issued before
        // the prefetch) instead of living in 12 registers through the harmonic phase,
        // where the 16 outputs of the last stage, the template row and the 12 sums
        // already fill the file
        RowTwiddles<M> tw;
        load_row_twiddles<M>(tw, as_global(a.twB), tid);
        const bool l0 = (tid == 0);
        const int tb = l0 ? 64 : 128 - tid;
        const size_t rc = (size_t)i * a.nchan + n;
        const bool looked = channel_lookup(a, i, n, n, M, n_held, mrow, ktn);
        // slots that hold a kept harmonic in SOME lane (uniform)
        const int nA = ktn / 128 + 1;                               // lane 0: 128 j <= ktn
        const int nB = ktn >= 64 ? (ktn - 64) / 128 + 1 : 0;        // lane 0: 64 + 128 j <= ktn
        if (looked && mrow != mheld) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int ka = tid + 128 * j, kb = tb + 128 * j;
                // (unconditional loads from clamped indices: see k_xspec)
                const cplx ma = mrow[max(ka, 1) - 1];
                const cplx mb = mrow[min(kb, M) - 1];
                const double ha = (ka >= 1 && ka <= ktn) ? 0.5 : 0.0, hb = (kb <= ktn) ? 0.5 : 0.0;
                mA[j] = make_double2(ha * ma.x, ha * ma.y);
                mB[j] = make_double2(hb * mb.x, hb * mb.y);
            }
            mheld = mrow;
        }
        // split twiddles of this lane's two butterflies (older than the prefetch below)
        // W_B^(128 - l) = W_B^128 conj(W_B^l); lane 0: W_B^64
        const cplx wA = as_global(a.twB)[tid];
        double phin = 0.0;
        if (FUSE) phin = a.ph0[rc];
        {
            cplx v[PER1][R1];
#pragma unroll
            for (int ii = 0; ii < PER1; ++ii)
#pragma unroll
                for (int k = 0; k < R1; ++k) v[ii][k] = to_cplx(cur[ii][k]);
            fft_first_stage<M, true>(lds, v, tw, tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            rw.next(i, n, i_nx, n_nx, nrows, a.nsub, a.ticket_base, a.ticket, a.mwords);
            const size_t rn = rw.more_nx ? (size_t)i_nx * a.nchan + n_nx : rc;
            stage_load_global<M, T, R1>(cur, reinterpret_cast<const Tin*>(a.data) + rn * (2 * M), tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- stage 2 through LDS, stage 3 in registers ----
        stage_lds<M, T, P::R2, P::R1, PL>(lds, tw.t2, tid);
        __builtin_amdgcn_sched_barrier(0);
        cplx va[8], vb[8];
        {
            constexpr int KSTEP = 128 + (128 >> PL);
            const cplx* pa = lds + lds_pad<PL>(tid);
            const cplx* pb = lds + lds_pad<PL>(tb);
#pragma unroll
            for (int k = 0; k < 8; ++k) { va[k] = pa[k * KSTEP]; vb[k] = pb[k * KSTEP]; }
            dft_reg<8>(va);
            dft_reg<8>(vb);
        }
        // S_d = sum_{k=1}^{M} |d_k|^2 = sum_{k=1}^{M-1} |Z_k|^2 + (Re Z_0 - Im Z_0)^2
        double sd = 0.0, tail = 0.0;
        {
            const double dM = va[0].x - va[0].y;
            sd = l0 ? dM * dM : cnorm(va[0]);
#pragma unroll
            for (int j = 1; j < 8; ++j) sd += cnorm(va[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sd += cnorm(vb[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const cplx wB = csel(l0, make_double2(0.98078528040323044913, -0.19509032201612826785),
                             cmulc(w128(1), wA));
        // 2 d_k = E - i W^k O,  E, O = Z_k +- conj Z_{M-k}
        auto dk2 = [](const cplx& zk, const cplx& zp, const cplx& w) -> cplx {
            const cplx E = make_double2(zk.x + zp.x, zk.y - zp.y);
            const cplx O = make_double2(zk.x - zp.x, zk.y + zp.y);
            const cplx wo = cmul(w, O);
            return make_double2(E.x + wo.y, E.y - wo.x);
        };
        if (TAIL) {
            // top quarter of the power spectrum, k = 768 .. 1024 (get_noise_PS):
            // slots A6, A7, B6, B7 and the Nyquist harmonic
            static_assert((int)(0.75 * (M + 1)) == 768, "tail slots written for kc = 768");
#pragma unroll
            for (int j = 6; j < 8; ++j) {
                const cplx da = dk2(va[j], csel(l0, va[8 - j], vb[7 - j]), cmul(wA, w128(j)));
                const cplx db = dk2(vb[j], csel(l0, vb[7 - j], va[7 - j]), cmul(wB, w128(j)));
                tail += 0.25 * (cnorm(da) + cnorm(db));
            }
            if (l0) { const double dM = va[0].x - va[0].y; tail += dM * dM; }
        }
        // ---- cross-spectrum of the kept harmonics ----
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        double tm[PP_TSTRIDE];
        cplx eA = make_double2(1.0, 0.0), eB = eA, e128 = eA;
        if (FUSE) {
            // e^{2 pi i k phi} for k = lane (lane 0 takes k = 64: its own k = 0 phasor
            // is 1, and every lane needs e^{2 pi i 64 phi})
            const cplx el = unit_phasor(l0 ? 64.0 : (double)tid, phin);
            const cplx e64 = make_double2(bcast_lane0(el.x), bcast_lane0(el.y));
            e128 = cmul(e64, e64);
            eA = csel(l0, make_double2(1.0, 0.0), el);
            eB = csel(l0, e64, cmulc(e128, el));
        }
        if (M2) {
            static_assert(PP_TJ == 10, "power ladder written for order 10");
#pragma unroll
            for (int j = 0; j < PP_TSTRIDE; ++j) tm[j] = 0.0;
        }
        auto taylor_sums = [&](const cplx& x, const cplx& z, double kap) {
            const double p2 = kap * kap, p4 = p2 * p2, p6 = p4 * p2, p8 = p4 * p4, p10 = p8 * p2;
            const double ui = z.y * kap;
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
            tm[11] = fma(p10 * kap, fabs(x.x) + fabs(x.y), tm[11]);
        };
        auto consume = [&](const cplx& x, const cplx& e, int k) {
            if (M2) taylor_sums(x, cmul(x, e), PP_TWO_PI * (double)k);
            else {
                if (k >= 1 && k <= ktn) store_x(a, rc, k, x);
                if (MODE == 1) {
                    const cplx z = cmul(x, e);
                    const double kk = (double)k;
                    s0 += z.x;
                    s1 = fma(kk, z.y, s1);
                    s2 = fma(kk * kk, z.x, s2);
                }
            }
        };
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (j < nA) {
                const cplx w = (j == 0) ? wA : cmul(wA, w128(j));
                const cplx x = cmulc(dk2(va[j], csel(l0, va[(8 - j) & 7], vb[7 - j]), w), mA[j]);
                consume(x, eA, tid + 128 * j);
            }
            if (j < nB) {
                const cplx w = (j == 0) ? wB : cmul(wB, w128(j));
                const cplx x = cmulc(dk2(vb[j], csel(l0, vb[7 - j], va[7 - j]), w), mB[j]);
                consume(x, eB, tb + 128 * j);
            }
            if (FUSE && j + 1 < NS) { eA = cmul(eA, e128); eB = cmul(eB, e128); }
            // keep the slots apart: interleaving them for ILP costs more registers
            // than the file has left here
            __builtin_amdgcn_sched_barrier(0);
        }
        sd = group_sum<64>(sd);
        if (TAIL) tail = group_sum<64>(tail);
        if (MODE == 1) { s0 = group_sum<64>(s0); s1 = group_sum<64>(s1); s2 = group_sum<64>(s2); }
        if (M2) {
            // (the image is free: its last reads, of stage 3, are older than these stores)
            const double tv = wave_reduce_lds(tm, tid & 63, reinterpret_cast<double*>(lds));
            if ((tid & 3) == 0) {
                const int q = wave_reduce16_index(tid);
                if (q < PP_TSTRIDE)   // Re(i^q z): +Re, -Im, -Re, +Im, ...
                    a.tay[tay_idx(rc, q)] = (q <= PP_TJ && ((q & 3) == 1 || (q & 3) == 2)) ? -tv : tv;
            }
        }
        if (tid == 0) {
            a.sdraw[rc] = sd;
            if (TAIL) a.noise[rc] = sqrt(tail / (2.0 * M) / (double)(M + 1 - 768));
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
