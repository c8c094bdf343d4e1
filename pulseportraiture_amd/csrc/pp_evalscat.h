// k_eval_scat: the chi^2 surface evaluator of a scattering fit (objective, gradient and
// Hessian terms of pptoaslib.py:525-643 from the stored cross-spectrum X_nk), one pass over
// X per evaluation.  grid = (subints in the list, channel chunks) -- the subint runs fastest,
// so the workgroups in flight together work on the SAME channels of different subints and
// share the template's |m_nk|^2 rows in their XCD's L2 --, 256 threads.
//
// Work that is per CHANNEL (the geometry: log / pow of the frequency ratio; the chain rule
// from the nine Fourier sums to the 21 per-subint sums: a division and ~100 multiplications)
// is done by one thread per channel, 256 channels at a time, before and after the part that
// is per HARMONIC -- in the first version every one of the 16 lanes that shared a channel
// repeated it, and it was 45 % of the kernel's instructions:
//   A  thread t <-> channel t of the block: phi_n, tau_n into LDS
//   B  LPC lanes per channel walk the harmonics k = l + 1, l + 1 + LPC, ... (phasor by
//      recurrence), the next two harmonics' X and |m|^2 in flight while two are worked on;
//      nine sums reduced over the LPC lanes (DPP, no LDS traffic) into LDS
//   C  thread t <-> channel t: the nine sums out to csum (the post-fit stage and the step's
//      scales read them), local terms, chain rule, block sum of the 21
#pragma once
// (included inside namespace pp, behind the definitions it uses: pp_kernels.h)

// the value of lane (l ^ O) of the same row of 16 lanes
template <int O>
__device__ __forceinline__ double lane_xor16(double v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8, "within a row of 16 lanes");
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (O == 4) {
        // (no DPP control swaps the halves of a row's halves: the swizzle unit, bit mode, xor 4)
        lo = __builtin_amdgcn_ds_swizzle(lo, 0x101F);
        hi = __builtin_amdgcn_ds_swizzle(hi, 0x101F);
    } else {
        // quad_perm [1,0,3,2] / quad_perm [2,3,0,1] / row_ror:8
        constexpr int ctrl = O == 1 ? 0xB1 : (O == 2 ? 0x4E : 0x128);
        lo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xF, 0xF, false);
    }
    return __hiloint2double(hi, lo);
}
// sum over an aligned group of W <= 16 lanes, on every lane (the butterfly of group_sum: same
// association, same bits)
template <int W>
__device__ __forceinline__ double group_sum16(double v) {
    if constexpr (W >= 16) v += lane_xor16<8>(v);
    if constexpr (W >= 8) v += lane_xor16<4>(v);
    if constexpr (W >= 4) v += lane_xor16<2>(v);
    if constexpr (W >= 2) v += lane_xor16<1>(v);
    return v;
}

template <int LPC, bool XF32>
__global__ __launch_bounds__(256, 4) void k_eval_scat(FitArgs a) {
    constexpr int G = 256 / LPC;          // channels in work at a time
    constexpr int S = 2 * LPC;            // harmonics a group takes per stage (two per lane)
    static_assert(16 % S == 0, "kt_n is a multiple of 16 at least (64, or M for nbin < 128)");
    const int jx = blockIdx.x, i = sub_of(a.act, jx), chunk = blockIdx.y;
    SubState& st = a.st[i];
    if (st.done || st.model == 1) return;    // (model == 1: this evaluation is k_scat_model's)
    __shared__ double s_phi[256], s_tau[256], s_cs[256 * PP_NCS];
    __shared__ int s_kt[256];             // harmonics to walk (0: channel of zero weight)
    __shared__ double s_cg[7 * 256];
    __shared__ double scratch[4 * PP_NACC];
    const int tid = threadIdx.x, g = tid / LPC, l = tid % LPC;
    const double phi = st.xe[0], DM = st.xe[1], GM = st.xe[2], alpha = st.xe[4];
    const double tau = a.log10_tau ? pow(10.0, st.xe[3]) : st.xe[3];
    const bool scat_on = (tau != 0.0);
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1], nutau = a.nu_fit[i * 3 + 2];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const int slot = a.slot ? a.slot[i] : 0;
    const double* msq = as_global(a.msq[slot]);
    const int* ktab = a.ktab ? as_global(a.ktab[slot]) : nullptr;
    const int trial = 1 - st.cur;
    double* csum = a.csum + ((size_t)trial * a.nsub + i) * a.nchan * PP_NCS;
    double total = 0.0;                   // thread j < 21: sum j of this chunk
    const int n0 = chunk * a.cpc, n1 = min(n0 + a.cpc, a.nchan_x);
    for (int base = n0; base < n1; base += 256) {
        const int cnt = min(256, n1 - base);
        // ---- A: geometry of channel base + tid ----
        const int nt = a.coff + (base + tid) * a.cstep;
        double wt = 0.0;
        if (tid < cnt) {
            wt = wts[nt];
            ChanGeom cg;
            chan_geom(freqs[nt], P, nuDM, nuGM, nutau, tau, alpha, a.log10_tau, scat_on, cg);
            // reference order of operations: phi + Dconst*DM*(f^-2 - nu^-2)/P + ...
            s_phi[tid] = phi + DM * cg.p1 + GM * cg.p2;
            s_tau[tid] = (cg.taun > 1e140) ? 1e140 : cg.taun;       // (u^2 stays finite; NaN stays NaN)
            // harmonics beyond the template's kept range carry |m_nk|^2 < 2^-100 of
            // the channel's power: they drop out of S_n(tau) and of C_n alike
            s_kt[tid] = (wt != 0.0) ? (ktab ? ktab[nt] : a.Kt) : 0;
            // (kept for C in LDS: over B they would cost every lane 14 registers)
            s_cg[0 * 256 + tid] = cg.p1; s_cg[1 * 256 + tid] = cg.p2;
            s_cg[2 * 256 + tid] = cg.q1; s_cg[3 * 256 + tid] = cg.q2;
            s_cg[4 * 256 + tid] = cg.q11; s_cg[5 * 256 + tid] = cg.q12; s_cg[6 * 256 + tid] = cg.q22;
        }
        __syncthreads();
        // ---- B: the nine Fourier sums of every channel ----
        for (int t = g; t < cnt; t += G) {
            const int nn = base + t, n = a.coff + nn * a.cstep;
            const double taun = s_tau[t];
            double s0 = 0, s1 = 0, s2 = 0, t1 = 0, t2 = 0, a1t = 0, S0 = 0, S1 = 0, S2 = 0;
            const int ktn = s_kt[t];
            if (ktn > 0) {
                cplx e = unit_phasor((double)(l + 1), s_phi[t]);
                // the step e^{2 pi i LPC phi_n}: the start phasor of the group's last lane
                cplx wst;
                {
                    const int src = ((tid & 63) & ~(LPC - 1)) | (LPC - 1);
                    wst = make_double2(__shfl(e.x, src, 64), __shfl(e.y, src, 64));
                }
                const size_t xoff = a.x_full ? ((size_t)jx * a.nchan + n) * a.Xs : ((size_t)jx * a.nchan_x + nn) * a.Xs;
                const cplx* xrow = a.X + xoff;
                const float2* xrow32 = reinterpret_cast<const float2*>(a.X) + xoff;
                const double* mrow = msq + (size_t)n * a.M;
                double k = (double)(l + 1);
                auto ldx = [&](int j) -> cplx {
                    if (XF32) {
                        const float2 xf = load_row_once<float2>(reinterpret_cast<const char*>(xrow32 + j));
                        return make_double2((double)xf.x, (double)xf.y);
                    }
                    return load_row_once<cplx>(reinterpret_cast<const char*>(xrow + j));
                };
                // a stage: harmonics js and js + LPC of this lane (kt_n is a multiple of S = 16: all
                // lanes of a group agree on the test, and a stage is whole or absent)
                auto load_stage = [&](int js, cplx& x0, cplx& x1, double& m0, double& m1) {
                    if (js < ktn) {
                        x0 = ldx(js); m0 = mrow[js];
                        x1 = ldx(js + LPC); m1 = mrow[js + LPC];
                    }
                };
                // (SC: tau != 0, decided once per channel instead of once per harmonic)
                auto walk = [&](auto sc) {
                    constexpr bool SC = decltype(sc)::value;
                    auto harm = [&](const cplx& x, double Mk) {
                        const cplx z = cmul(x, e);
                        const double kap = PP_TWO_PI * k, u = kap * taun;
                        // |B_nk|^2 = 1 / (1 + u^2): hardware estimate + two Newton steps, ~1 ulp (recip_ge1
                        // without its guard: tau_n is bounded in A)
                        const double q = fma(u, u, 1.0);
                        double D = __builtin_amdgcn_rcp(q);
                        D = fma(fma(-q, D, 1.0), D, D);
                        D = fma(fma(-q, D, 1.0), D, D);
                        const cplx b = make_double2(D, u * D);          // conj(B)
                        const cplx zb = cmul(z, b);
                        s0 += zb.x;
                        s1 = fma(kap, zb.y, s1);                        // A1 = -sum kap Im(zb)
                        s2 = fma(kap * kap, zb.x, s2);                  // A2 = -sum kap^2 Re(zb)
                        S0 = fma(D, Mk, S0);
                        if (SC) {
                            const cplx zb2 = cmul(zb, b);
                            const cplx zb3 = cmul(zb2, b);
                            t1 = fma(kap, zb2.y, t1);                   // T1 = -sum kap Im(z b^2)
                            a1t = fma(kap * kap, zb2.x, a1t);           // A1T = -sum kap^2 Re(z b^2)
                            t2 = fma(kap * kap, zb3.x, t2);             // T2 = -2 sum kap^2 Re(z b^3)
                            const double D2 = D * D;
                            S1 = fma(kap * u * D2, Mk, S1);             // S1 = -2 sum kap u D^2 M
                            S2 = fma(kap * kap * D2 * fma(4.0 * u * u, D, -1.0), Mk, S2);  // *2
                        }
                        e = cmul(e, wst);
                        k += (double)LPC;
                    };
                    // two register sets: one worked on while the other is in flight
                    int j = l;
                    cplx xa0, xa1, xb0, xb1;
                    double ma0, ma1, mb0, mb1;
                    load_stage(j, xa0, xa1, ma0, ma1);
                    while (true) {
                        load_stage(j + S, xb0, xb1, mb0, mb1);
                        harm(xa0, ma0);
                        harm(xa1, ma1);
                        j += S;
                        if (j >= ktn) break;
                        load_stage(j + S, xa0, xa1, ma0, ma1);
                        harm(xb0, mb0);
                        harm(xb1, mb1);
                        j += S;
                        if (j >= ktn) break;
                    }
                };
                if (scat_on) walk(std::true_type{}); else walk(std::false_type{});
            }
            double cs[PP_NCS];
            cs[0] = group_sum16<LPC>(s0);
            cs[1] = group_sum16<LPC>(s1);
            cs[2] = group_sum16<LPC>(s2);
            cs[3] = group_sum16<LPC>(t1);
            cs[4] = group_sum16<LPC>(t2);
            cs[5] = group_sum16<LPC>(a1t);
            cs[6] = group_sum16<LPC>(S0);
            cs[7] = group_sum16<LPC>(S1);
            cs[8] = group_sum16<LPC>(S2);
            // lane q keeps sum q (LPC = 8: lane 0 sum 8 as well)
            double v = cs[0];
#pragma unroll
            for (int q = 1; q < PP_NCS && q < LPC; ++q) v = (l == q) ? cs[q] : v;
            if (l < PP_NCS) s_cs[t * PP_NCS + l] = v;
            if (LPC < PP_NCS && l == 0) s_cs[t * PP_NCS + 8] = cs[8];
        }
        __syncthreads();
        // ---- C: chain rule of channel base + tid ----
        double c[PP_NACC];
#pragma unroll
        for (int q = 0; q < PP_NACC; ++q) c[q] = 0.0;
        if (tid < cnt) {
            double cs[PP_NCS];
#pragma unroll
            for (int q = 0; q < PP_NCS; ++q) cs[q] = s_cs[tid * PP_NCS + q];
            cs[1] = -cs[1]; cs[2] = -cs[2]; cs[3] = -cs[3]; cs[4] = -2.0 * cs[4]; cs[5] = -cs[5];
            cs[7] = -2.0 * cs[7]; cs[8] = 2.0 * cs[8];
            double* co = csum + (size_t)nt * PP_NCS;
#pragma unroll
            for (int q = 0; q < PP_NCS; ++q) co[q] = cs[q];
            if (wt != 0.0) {
                ChanGeom cg;
                cg.p1 = s_cg[0 * 256 + tid]; cg.p2 = s_cg[1 * 256 + tid];
                cg.q1 = s_cg[2 * 256 + tid]; cg.q2 = s_cg[3 * 256 + tid];
                cg.q11 = s_cg[4 * 256 + tid]; cg.q12 = s_cg[5 * 256 + tid]; cg.q22 = s_cg[6 * 256 + tid];
                cg.taun = cg.lnf = 0.0;          // (not used by the chain rule)
                const Local L = local_terms(cs, wt);
                accumulate_channel(L, cg, c);
            }
        }
        block_sum<PP_NACC>(c, scratch);       // (syncs before and after its use of scratch)
        double mine = 0.0;
#pragma unroll
        for (int q = 0; q < PP_NACC; ++q) mine = (tid == q) ? c[q] : mine;
        total += mine;
        __syncthreads();                      // (s_phi, s_tau, s_cs and scratch are reused)
    }
    if (tid < PP_NACC) a.partial[((size_t)i * a.nchunk + chunk) * PP_NACC + tid] = total;
}

