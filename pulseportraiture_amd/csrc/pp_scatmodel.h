// Scattering fits: the closing iterations without passes over the cross-spectrum.
//
// (included from pp_kernels.h, inside namespace pp, after step_logic)
//
// With scattering, one evaluation of the objective (pptoaslib.py:495-640) is a pass
// over X_nk.  The reference's minimiser spends most of its evaluations within a
// hair of the optimum (SciPy's trust-ncg: ~15 evaluations, the last ~8 of them
// moving phi_n by < 1e-4 rot and tau_n by < 1 %).  There the per-channel sums are
// an analytic function of the two per-channel quantities the parameters act
// through, the phase phi_n and the scattering time tau_n:
//
//   C_n(phi_n + d, tau_n + e) = Re sum_k z_k e^{i kap d} b_k / (1 - i kap e b_k)
//                             = sum_{a,c} d^a/a! e^c G[a+c][c],
//   G[q][c] = Re sum_k (i kap)^q z_k b_k^(c+1),   z_k = X_nk e^{i kap phi_n},
//   b_k = conj(B_nk) = 1/(1 - i kap tau_n),  kap = 2 pi k,
//   S_n(tau_n + e) = sum_k |m_nk|^2 Re(b_k / (1 - i kap e b_k)) = sum_c e^c Sc[c],
//   Sc[c] = sum_k |m_nk|^2 kap^c Re(i^c b_k^(c+1))
//
// (|B|^2 = Re conj(B) for B = 1/(1 + i kap tau)).  k_scat_model takes these
// coefficients to total degree PP_MP in ONE pass over X -- that pass is also an exact
// evaluation at its centre, so it takes the place of an ordinary one -- and
// k_scat_model_solve then walks the remaining iterations of the same minimiser
// (step_logic, the code the ordinary path runs) on the polynomial, each evaluation
// O(nchan).  Every evaluation is certified: the dropped terms are bounded by
//   sum_k |z_k||b_k| sum_{a+c > P} (kap|d|)^a/a! (kap|e||b_k|)^c
//     <= W_0 e^{kap_max |d|} sum_{a+c > P} (keff|d|)^a/a! rho^c,
//   W_a = sum_k |z_k||b_k| kap^a,  keff = (W_(P+1)/W_0)^(1/(P+1))  (W_a <= W_0 keff^a for
//   a <= P+1 by the log-convexity of moments),  rho = |e| max_k kap|b_k|,
// and turned into a bound on the position of the optimum (gradient error /
// curvature) that must stay below 1e-13 rot of phase (and matching bars on the other
// parameters); an evaluation that fails the certificate is simply made over X again
// (the state is that of the ordinary path at every moment).
#pragma once

#ifndef PP_MP
#define PP_MP 8                 // total degree of the per-channel model
#endif
#define PP_MNG ((PP_MP + 1) * (PP_MP + 2) / 2)       // G[q][c], c <= q <= P, at q(q+1)/2 + c
#define PP_MROW ((PP_MNG + PP_MP + 4 + 1) & ~1)       // + Sc[0..P], W_0, tau_n, W_(P+1) (even)
static_assert(PP_MROW % 2 == 0, "rows of the scattering model are moved in pairs");
static_assert(PP_MROW <= 64, "one reduce-scatter of 64 values per channel");

// sum over a + c >= m of x^a/a! rho^c  (0 <= x <= 1/2, 0 <= rho <= 1/2), from above:
// sum_c rho^c R_(m-c)(x) with R_j(x) = sum_{a >= j} x^a/a! <= e^x x^j/j!, and the
// cheap majorants e^x <= 1 + x + x^2, 1/(1 - rho) <= 1 + 2 rho on that range
__device__ inline double series_tail(int m, double x, double rho) {
    double xf[PP_MP + 2];       // x^j / j!
    xf[0] = 1.0;
    for (int j = 1; j <= m; ++j) xf[j] = xf[j - 1] * x * (1.0 / (double)j);
    double total = 0.0, rc = 1.0;
    for (int c = 0; c <= m; ++c) { total += rc * xf[m - c]; rc *= rho; }
    return fma(x, x, 1.0 + x) * fma(rc, fma(2.0, rho, 1.0), total);
}

// largest |d phi_n / d DM|, |d phi_n / d GM|, |ln(nu_n / nu_tau)| of a subint's
// channels, and the effective harmonic scale keff of the template (from |m_nk|, the
// weighting of a noise-dominated cross-spectrum, on 16 channels across the band) -- 64 lanes;
// kept in the state for the switch criterion
__device__ inline void scat_model_geometry(const FitArgs& a, int i, SubState& s) {
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1], nutau = a.nu_fit[i * 3 + 2];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    double m1 = 0.0, m2 = 0.0, m3 = 0.0;
#pragma unroll 4
    for (int n = threadIdx.x; n < a.nchan; n += 64) {
        if (wts[n] == 0.0) continue;
        double p1, p2;
        phase_geom(freqs[n], P, nuDM, nuGM, p1, p2);
        m1 = fmax(m1, fabs(p1)); m2 = fmax(m2, fabs(p2));
        m3 = fmax(m3, fabs(log(freqs[n] / nutau)));
    }
    m1 = group_max<64>(m1); m2 = group_max<64>(m2); m3 = group_max<64>(m3);
    const int slot = a.slot ? a.slot[i] : 0;
    const double* msq = as_global(a.msq[slot]);
    const int* ktv = a.ktab ? as_global(a.ktab[slot]) : nullptr;
    // 16 channels across the band at once: 4 lanes per channel, eight loads in flight per lane
    double keff = 0.0;
    {
        const int lane = threadIdx.x, cq = lane >> 2, q = lane & 3;
        const int stride = max(1, a.nchan / 16), n = cq * stride;
        double w0 = 0.0, wp = 0.0;
        if (n < a.nchan && wts[n] != 0.0) {
            const int ktn = ktv ? ktv[n] : a.Kt;
            const double* mrow = msq + (size_t)n * a.M;
#pragma unroll 8
            for (int k = q; k < ktn; k += 4) {
                const double m = sqrt(mrow[k]), kap = PP_TWO_PI * (double)(k + 1);
                double kp = kap;
#pragma unroll
                for (int qq = 0; qq < PP_MP; ++qq) kp *= kap;
                w0 += m; wp = fma(m, kp, wp);
            }
        }
        w0 = group_sum<4>(w0); wp = group_sum<4>(wp);
        if (w0 > 0.0) keff = exp(log(wp / w0) / (double)(PP_MP + 1));
        keff = group_max<64>(keff);
    }
    if (threadIdx.x == 0) { s.geo[0] = m1; s.geo[1] = m2; s.geo[2] = m3; s.geo[3] = keff; }
}

// After a proposal: will the rest of the iteration stay where a degree-PP_MP model
// about the proposed point holds?  The Newton step of the current quadratic model
// estimates how far the optimum is; that + the proposal must keep the dropped terms
// below model_tol (1e-10) of the kept ones (a prediction only: the certificate of every model
// evaluation is what guards the result).  Marks the next evaluation as the model pass.
__device__ inline void scat_model_request(const FitArgs& a, SubState& s) {
    if (s.fresh != 0) return;            // (closing evaluation pending: nothing follows it)
    const double tau = a.log10_tau ? pow(10.0, s.xe[3]) : s.xe[3];
    if (!(tau > 0.0)) return;
    int idx[5], n = 0;
    for (int j = 0; j < 5; ++j) if (a.flags[j]) idx[n++] = j;
    double gs[5], Hs[25], mg[5], pn[5];
    for (int r = 0; r < n; ++r) {
        gs[r] = s.g[idx[r]]; mg[r] = -gs[r];
        for (int c = 0; c < n; ++c) Hs[r * n + c] = s.H[idx[r] * 5 + idx[c]];
    }
    if (!chol_solve(n, Hs, mg, pn)) return;
    double pnf[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int r = 0; r < n; ++r) pnf[idx[r]] = pn[r];
    // per-channel size of a parameter displacement: phase [rot] and relative tau_n
    auto dphi_of = [&](const double* v) { return fabs(v[0]) + fabs(v[1]) * s.geo[0] + fabs(v[2]) * s.geo[1]; };
    auto rel_of = [&](const double* v) {
        return (a.log10_tau ? PP_LN10 * fabs(v[3]) : fabs(v[3]) / tau) + fabs(v[4]) * s.geo[2];
    };
    auto inside = [&](double dphi, double rel) {
        const double x = 1.25 * s.geo[3] * dphi, rho = 2.0 * rel;
        if (!(x < 0.5) || !(rho < 0.2) || !(PP_TWO_PI * (double)a.Kt * dphi < 1.0)) return false;
        return series_tail(PP_MP, x, rho) * (PP_MP + 1) <= a.model_tol;
    };
    double pr[5], D[5];
    for (int j = 0; j < 5; ++j) { pr[j] = s.xe[j] - s.x[j]; D[j] = fabs(pnf[j]) + fabs(pr[j]); }
    // (a) nothing between x, the proposal and the predicted optimum leaves the range
    if (inside(dphi_of(D), rel_of(D))) { s.model = 1; return; }
    // (b) one evaluation earlier, on a bet: the proposal is the full Newton step (the CG
    // iteration converged, no boundary), so it lands about (relative step)^2 from the
    // optimum in tau_n -- the scale of the objective's nonlinearity in ln tau is 1 -- and
    // within a tenth of the step in phase, where the objective is far closer to quadratic
    // (measured: 0.025).  Twice that must be inside the range, to
    // a looser tolerance: the certificate of every model evaluation is what guards the
    // result, and a lost bet costs the difference between the model pass and an ordinary
    // one, once (a subint gets a second chance under (a) only).
    if (!a.model_bet || s.hits_boundary || s.nmodel > 0) return;
    double dn[5];
    for (int j = 0; j < 5; ++j) dn[j] = pnf[j] - pr[j];
    const double px = dphi_of(pr), prl = rel_of(pr), dx_ = dphi_of(dn), dr_ = rel_of(dn);
    if (!(dx_ <= 0.1 * px + 1e-7) || !(dr_ <= 0.1 * prl + 1e-6)) return;
    const double ex = 2.0 * (0.1 * px + dx_), er = 2.0 * (prl * prl + dr_);
    const double x = 1.25 * s.geo[3] * ex, rho = 2.0 * er;
    if (!(x < 0.5) || !(rho < 0.2) || !(PP_TWO_PI * (double)a.Kt * ex < 1.0)) return;
    if (series_tail(PP_MP, x, rho) * (PP_MP + 1) <= 1e4 * a.model_tol) s.model = 1;
}

// --------------------------------------------------------------------------
// The model pass.  Same grid and phases as k_eval_scat ((nsub, nchunk); A: one thread per
// channel for the geometry, B: 8 lanes per channel on the harmonics), for the subints
// whose state asks for it.  Leaves the PP_MROW coefficients of every channel in a.mdl and
// the nine sums of the centre in the trial csum buffer, exactly as k_eval_scat would.
// f64 VALU bound (~165 instructions per harmonic: two power chains in b_k, 54 FMAs): with 8
// lanes per channel the per-channel part (phasor, zeroing and reducing 58 accumulators) is
// 8 % of the instructions (16 lanes, geometry repeated by every lane: 24 %).
// --------------------------------------------------------------------------
// Sum 64 per-lane values over an aligned group of 8 lanes, leaving lane l of the group with
// the totals of v[8l .. 8l+7] (in v[0..7]): 56 exchanges (DPP / swizzle, no LDS traffic)
__device__ __forceinline__ void group8_reduce_scatter64(double (&v)[64], int lane) {
#define PP_RS8_STEP(HALF, BIT)                                            \
    {                                                                    \
        const bool up = (lane & (BIT)) != 0;                             \
        _Pragma("unroll") for (int j = 0; j < (HALF); ++j) {             \
            const double keep = up ? v[(HALF) + j] : v[j];               \
            const double send = up ? v[j] : v[(HALF) + j];               \
            v[j] = keep + lane_xor16<BIT>(send);                         \
        }                                                                \
    }
    PP_RS8_STEP(32, 4)
    PP_RS8_STEP(16, 2)
    PP_RS8_STEP(8, 1)
#undef PP_RS8_STEP
}

__global__ __launch_bounds__(256, 2) void k_scat_model(FitArgs a) {
    constexpr int LPC = 8, G_ = 256 / LPC, P_ = PP_MP;
    const int jx = blockIdx.x, i = sub_of(a.act, jx), chunk = blockIdx.y;
    const SubState& st = a.st[i];
    if (st.done || st.model != 1) return;
    __shared__ double s_phi[256], s_tau[256];
    __shared__ int s_kt[256];
    const int tid = threadIdx.x, g = tid / LPC, l = tid % LPC;
    const double phi = st.xe[0], DM = st.xe[1], GM = st.xe[2], alpha = st.xe[4];
    const double tau = a.log10_tau ? pow(10.0, st.xe[3]) : st.xe[3];
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1], nutau = a.nu_fit[i * 3 + 2];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const int slot = a.slot ? a.slot[i] : 0;
    const double* msq = as_global(a.msq[slot]);
    const int* ktab = a.ktab ? as_global(a.ktab[slot]) : nullptr;
    double* mdl = a.mdl + (size_t)i * a.nchan * PP_MROW;
    double* csum = a.csum + ((size_t)(1 - st.cur) * a.nsub + i) * a.nchan * a.ncs;
    const int n0 = chunk * a.cpc, n1 = min(n0 + a.cpc, a.nchan_x);
    const int src = ((tid & 63) & ~(LPC - 1)) | (LPC - 1);
    for (int base = n0; base < n1; base += 256) {
        const int cnt = min(256, n1 - base);
        // ---- A: geometry of channel base + tid ----
        if (tid < cnt) {
            const int nt = a.coff + (base + tid) * a.cstep;
            ChanGeom cg;
            chan_geom(freqs[nt], P, nuDM, nuGM, nutau, tau, alpha, a.log10_tau, true, cg);
            s_phi[tid] = phi + DM * cg.p1 + GM * cg.p2;
            s_tau[tid] = cg.taun;
            s_kt[tid] = (wts[nt] != 0.0) ? (ktab ? ktab[nt] : a.Kt) : 0;
        }
        __syncthreads();
        // ---- B: the coefficients of every channel ----
        for (int t = g; t < cnt; t += G_) {
            const int nn = base + t, n = a.coff + nn * a.cstep;
            const double taun = s_tau[t];
            const int ktn = s_kt[t];
            double G[PP_MNG], Sc[P_ + 1], An = 0.0, Wp = 0.0;
#pragma unroll
            for (int j = 0; j < PP_MNG; ++j) G[j] = 0.0;
#pragma unroll
            for (int j = 0; j <= P_; ++j) Sc[j] = 0.0;
            if (l < ktn) {
                cplx e = unit_phasor((double)(l + 1), s_phi[t]);
                const cplx wst = make_double2(__shfl(e.x, src, 64), __shfl(e.y, src, 64));
                const cplx* xrow = a.X + ((size_t)jx * a.nchan_x + nn) * a.Xs;
                const double* mrow = msq + (size_t)n * a.M;
                double k = (double)(l + 1);
                // the next harmonic's loads are issued before this one's arithmetic
                cplx xn = load_row_once<cplx>(reinterpret_cast<const char*>(xrow + l));
                double Mn = mrow[l];
#pragma unroll 1
                for (int j = l; j < ktn; j += LPC) {
                    const cplx x = xn;
                    const double Mk = Mn;
                    if (j + LPC < ktn) {
                        xn = load_row_once<cplx>(reinterpret_cast<const char*>(xrow + j + LPC));
                        Mn = mrow[j + LPC];
                    }
                    const cplx z = cmul(x, e);
                    const double kap = PP_TWO_PI * k, u = kap * taun;
                    const double D = recip_ge1(fma(u, u, 1.0));
                    const cplx b = make_double2(D, u * D);
                    double kp[P_ + 1];
                    kp[0] = 1.0;
#pragma unroll
                    for (int q = 1; q <= P_; ++q) kp[q] = kp[q - 1] * kap;
                    cplx zb = cmul(z, b);                           // z b^(c+1)
                    cplx bp = make_double2(Mk * b.x, Mk * b.y);     // |m_nk|^2 b^(c+1)
#pragma unroll
                    for (int c = 0; c <= P_; ++c) {
                        if (c > 0) { zb = cmul(zb, b); bp = cmul(bp, b); }
                        // Re(i^q v): +Re, -Im, -Re, +Im; the sign goes on at the end
#pragma unroll
                        for (int q = c; q <= P_; ++q)
                            G[q * (q + 1) / 2 + c] = fma(kp[q], (q & 1) ? zb.y : zb.x, G[q * (q + 1) / 2 + c]);
                        Sc[c] = fma(kp[c], (c & 1) ? bp.y : bp.x, Sc[c]);
                    }
                    // |b_k| from above: a single-precision rsqrt, widened by its error
                    const float rs = __builtin_amdgcn_rsqf((float)fma(u, u, 1.0)) * 1.0001f;
                    const double ax = (fabs(x.x) + fabs(x.y)) * (double)rs;
                    An += ax;
                    Wp = fma(ax, kp[P_] * kap, Wp);
                    e = cmul(e, wst);
                    k += (double)LPC;
                }
            }
            // signs of Re(i^q v), then one reduce-scatter of the whole row over the 8 lanes
            double V[64];
#pragma unroll
            for (int q = 0; q <= P_; ++q)
#pragma unroll
                for (int c = 0; c <= q; ++c) {
                    const int j = q * (q + 1) / 2 + c;
                    V[j] = ((q & 3) == 1 || (q & 3) == 2) ? -G[j] : G[j];
                }
#pragma unroll
            for (int c = 0; c <= P_; ++c) V[PP_MNG + c] = ((c & 3) == 1 || (c & 3) == 2) ? -Sc[c] : Sc[c];
            V[PP_MNG + P_ + 1] = An;
            V[PP_MNG + P_ + 2] = 0.0;                 // (tau_n goes here)
            V[PP_MNG + P_ + 3] = Wp;
#pragma unroll
            for (int j = PP_MNG + P_ + 4; j < 64; ++j) V[j] = 0.0;
            group8_reduce_scatter64(V, l);
            constexpr int JT = PP_MNG + P_ + 2;       // tau_n's place in the row
            if (l == JT / 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) if (q == JT % 8) V[q] = taun;
            }
            // coefficient j of channel n goes to mdl[j * nchan + n] (the 32 channels a
            // workgroup has in flight are neighbours: 256 B per coefficient)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (8 * l + q < PP_MROW) mdl[(size_t)(8 * l + q) * a.nchan + n] = V[q];
            // the nine sums of the centre for the post-fit stage:
            // A0 = G00, A1 = G10, T1 = G11, A2 = G20, A1T = G21, T2 = 2 G22; S0, S1 = Sc1, S2 = 2 Sc2
            double* co = csum + (size_t)n * a.ncs;
            if (l == 0) { co[0] = V[0]; co[1] = V[1]; co[3] = V[2]; co[2] = V[3]; co[5] = V[4]; co[4] = 2.0 * V[5]; }
            static_assert(PP_MNG % 8 <= 5, "Sc[0..2] sit in one lane's octet");
            if (l == PP_MNG / 8) {
                constexpr int o = PP_MNG % 8;
                co[6] = V[o]; co[7] = V[o + 1]; co[8] = 2.0 * V[o + 2];
            }
        }
        __syncthreads();          // (s_phi, s_tau, s_kt are reused)
    }
}

// the nine sums at (phi_n + d, tau_n + e) from a channel's coefficients (row[j * rs])
__device__ __forceinline__ void scat_model_sums(const double* row, size_t rs, double d, double e, double* cs) {
    constexpr int P_ = PP_MP;
    // Horner in e over c = P .. 0 of h_c(d), h_c'(d), h_c''(d), each a series in d
    double v = 0.0, v1 = 0.0, v2 = 0.0;      // C, dC/de, (1/2) d2C/de2
    double u = 0.0, u1 = 0.0;                // dC/dd, d2C/(dd de)
    double t = 0.0;                          // d2C/dd2
#pragma unroll
    for (int c = P_; c >= 0; --c) {
        double h0 = 0.0, h1 = 0.0, h2 = 0.0;
#pragma unroll
        for (int aa = P_ - c; aa >= 0; --aa) {
            const double G = row[(size_t)((aa + c) * (aa + c + 1) / 2 + c) * rs];
            h0 = fma(h0, d * (1.0 / (double)(aa + 1)), G);
            if (aa >= 1) h1 = fma(h1, d * (1.0 / (double)aa), G);
            if (aa >= 2) h2 = fma(h2, d * (1.0 / (double)(aa - 1)), G);
        }
        v2 = fma(v2, e, v1); v1 = fma(v1, e, v); v = fma(v, e, h0);
        u1 = fma(u1, e, u); u = fma(u, e, h1);
        t = fma(t, e, h2);
    }
    cs[0] = v; cs[1] = u; cs[2] = t; cs[3] = v1; cs[4] = 2.0 * v2; cs[5] = u1;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int c = P_; c >= 0; --c) {
        s2 = fma(s2, e, s1); s1 = fma(s1, e, s0); s0 = fma(s0, e, row[(size_t)(PP_MNG + c) * rs]);
    }
    cs[6] = s0; cs[7] = s1; cs[8] = 2.0 * s2;
}

// --------------------------------------------------------------------------
// The iterations on the model: one 256-thread block per subint whose model pass
// just ran.  Every round evaluates f, g, H at the pending proposal s.xe from the
// rows (the first round at the centre itself: exact), checks the certificate, and
// lets thread 0 run the same step_logic as k_step.  All or nothing: either the whole
// remaining iteration runs on the model (then the sums of the accepted point are
// published for the post-fit stage), or -- an evaluation fails its certificate -- the
// state the kernel was entered with is restored untouched and the ordinary path makes
// this very evaluation over the cross-spectrum: a lost model pass changes nothing
// but the time.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scat_model_solve(FitArgs a) {
    const int i = sub_of(a.act, blockIdx.x), tid = threadIdx.x;
    SubState& st = a.st[i];
    if (st.done || st.model != 1) return;
    __shared__ SubState ss, ss_in;
    __shared__ double scratch[4 * (PP_NACC + 6)];
    __shared__ int flag;
    if (tid == 0) { ss = st; ss_in = st; }
    __syncthreads();
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1], nutau = a.nu_fit[i * 3 + 2];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const double* mdl = a.mdl + (size_t)i * a.nchan * PP_MROW;
    double xc[5];                                   // the centre
    for (int j = 0; j < 5; ++j) xc[j] = ss.xe[j];
    const double tau_c = a.log10_tau ? pow(10.0, xc[3]) : xc[3];
    const double kmax = PP_TWO_PI * (double)a.Kt;
    const double tol[5] = {1e-13, 1e-11, 1e-8, a.log10_tau ? 4e-11 : 1e-10 * tau_c, 1e-9};
#ifdef PP_SOLVE_TIMING
    long long tA = 0, tB = 0, tC = 0, t0 = clock64(), t1;
#endif
    for (int round = 0;; ++round) {
        double xe[5];
        // (after the iteration has ended: one more sweep at the accepted point, which
        // only publishes its sums)
        const bool publish = round > 0 && flag == 3;
        for (int j = 0; j < 5; ++j) xe[j] = publish ? ss.x[j] : ss.xe[j];
        const int outbuf = ss.cur;
        __syncthreads();
        const double tau_e = a.log10_tau ? pow(10.0, xe[3]) : xe[3];
        const double dlnt = a.log10_tau ? PP_LN10 * (xe[3] - xc[3]) : log(tau_e / tau_c);
        const double dal = xe[4] - xc[4];
        const double dph = xe[0] - xc[0], dDM = xe[1] - xc[1], dGM = xe[2] - xc[2];
        double* csum = a.csum + ((size_t)outbuf * a.nsub + i) * a.nchan * a.ncs;
        double acc[PP_NACC + 6];
#pragma unroll
        for (int j = 0; j < PP_NACC + 6; ++j) acc[j] = 0.0;
        bool inside = (tau_e > 0.0);
        for (int n = tid; n < a.nchan; n += 256) {
            const double w = wts[n];
            // coefficient j of channel n sits at mdl[j * nchan + n]: neighbouring threads
            // read neighbouring words.  All of them are requested before the first is
            // used (one memory latency per channel, not one per Horner step)
            double row[PP_MROW];
#pragma unroll
            for (int j = 0; j < PP_MROW; ++j) row[j] = mdl[(size_t)j * a.nchan + n];
            constexpr size_t rs = 1;
            ChanGeom cg;
            phase_geom(freqs[n], P, nuDM, nuGM, cg.p1, cg.p2);
            const double lnf = log(freqs[n] / nutau);
            const double taun_c = row[(size_t)(PP_MNG + PP_MP + 2) * rs];
            const double eps = taun_c * expm1(dlnt + dal * lnf);
            const double taun = taun_c + eps;
            cg.lnf = lnf; cg.taun = taun;
            if (!a.log10_tau) {
                cg.q1 = taun / tau_e; cg.q2 = lnf * taun; cg.q11 = 0.0; cg.q12 = cg.q2 / tau_e;
            } else {
                cg.q1 = PP_LN10 * taun; cg.q2 = lnf * taun; cg.q11 = PP_LN10 * cg.q1; cg.q12 = PP_LN10 * cg.q2;
            }
            cg.q22 = lnf * cg.q2;
            const double d = dph + dDM * cg.p1 + dGM * cg.p2;
            double cs[PP_NCS];
            scat_model_sums(row, rs, d, eps, cs);
            if (publish) {
#pragma unroll
                for (int j = 0; j < PP_NCS; ++j) csum[(size_t)n * a.ncs + j] = cs[j];
                continue;
            }
            if (w == 0.0) continue;
            const Local L = local_terms(cs, w);
            double c[PP_NACC];
            accumulate_channel(L, cg, c);
#pragma unroll
            for (int j = 0; j < PP_NACC; ++j) acc[j] += c[j];
            if (round > 0) {
                // certificate: dropped terms of the first derivatives
                // (single-precision hardware rsqrt / log2 / exp2, widened by their error)
                const double ut = kmax * taun_c;
                const double kb = kmax * (double)(__builtin_amdgcn_rsqf((float)fma(ut, ut, 1.0)) * 1.0001f);
                const double W0 = row[(size_t)(PP_MNG + PP_MP + 1) * rs], Wp = row[(size_t)(PP_MNG + PP_MP + 3) * rs];
                int ex = 0;
                const float mant = (float)frexp((W0 > 0.0) ? Wp / W0 : 0.0, &ex);   // (the ratio can exceed the f32 range)
                const double keff = (W0 > 0.0)
                    ? (double)(__builtin_amdgcn_exp2f((__builtin_amdgcn_logf(mant) + (float)ex) * (1.0f / (PP_MP + 1))) * 1.0001f)
                    : 0.0;
                const double y = kmax * fabs(d);
                const double x = keff * fabs(d), rho = 2.0 * fabs(eps) * kb;
                if (!(x < 0.5) || !(rho < 0.5) || !(y < 1.0)) { inside = false; continue; }
                const double e1 = W0 * series_tail(PP_MP, x, rho) * fma(y, y, 1.0 + y);   // (e^y <= 1 + y + y^2)
                const double r = fabs(cs[0] / cs[6]) + 1e-300;
                const double gp = 3.0 * w * r * e1 * kmax, gt = 3.0 * w * r * e1 * kb;
                acc[PP_NACC] += gp; acc[PP_NACC + 1] += gp * fabs(cg.p1); acc[PP_NACC + 2] += gp * fabs(cg.p2);
                acc[PP_NACC + 3] += gt * fabs(cg.q1); acc[PP_NACC + 4] += gt * fabs(cg.q2);
                // ... and of f itself (no derivative: one order more, rho without its factor 2):
                // SciPy's decisions hang on the last bits of f
                acc[PP_NACC + 5] += 2.0 * w * r * W0 * series_tail(PP_MP + 1, x, 0.5 * rho) * fma(y, y, 1.0 + y);
            }
        }
#ifdef PP_SOLVE_TIMING
        t1 = clock64(); tA += t1 - t0; t0 = t1;
#endif
        if (publish) break;
        block_sum<PP_NACC + 6>(acc, scratch);
        if (tid == 0) flag = 0;
        __syncthreads();
        if (!inside) flag = 1;               // (any thread)
        __syncthreads();
#ifdef PP_SOLVE_TIMING
        t1 = clock64(); tB += t1 - t0; t0 = t1;
#endif
        if (tid == 0) {
            double f, g[5], H[25];
            unpack_acc(acc, a.flags, f, g, H);
            bool ok = (flag == 0);
#pragma unroll
            for (int j = 0; j < 5; ++j)      // (static indices: acc stays in registers)
                if (a.flags[j] && !(acc[PP_NACC + j] <= tol[j] * fabs(H[j * 5 + j]))) ok = false;
            // f to a quarter of its last bit
            if (!(acc[PP_NACC + 5] <= 0.25 * 2.220446049250313e-16 * fabs(f))) ok = false;
            if (!ok) {
                // back to the state of entry: the ordinary path evaluates the centre over X
                // and goes on from there; one more model pass may be asked for later
                // (criterion (a) only)
                ss = ss_in;
                ss.nmodel += 1;
                // (4 / 5: k_step of THIS iteration must leave the subint alone -- its
                // evaluation has not been made, the partial sums are an older one's -- and
                // turns the mark into 0 / 3)
                ss.model = (ss.nmodel < 2) ? 4 : 5;
                flag = 2;
            } else {
#ifdef PP_STEP_TRACE
                if (i == PP_STEP_TRACE)
                    printf("mdl round %d it %2d f %.17g f_new %.17g actual %.3e pred %.3e radius %.3e  (f-bound %.2e ulp)\n", round,
                           ss.iter, ss.f, f, ss.f - f, ss.pred_red, ss.radius, acc[PP_NACC + 5] / (2.220446049250313e-16 * fabs(f)));
#endif
                ss.model = 2;
                const bool done = step_logic(a, ss, f, g, H, round == 0);   // (the model pass itself is a pass over X)
                ss.model = 1;
                if (done) { ss.done = 1; flag = 3; }
            }
        }
        __syncthreads();
#ifdef PP_SOLVE_TIMING
        t1 = clock64(); tC += t1 - t0; t0 = t1;
        if (flag >= 2 && tid == 0 && i == 0) printf("solve i=0 rounds %d: loop %lld  sum %lld  logic %lld cycles\n", round + 1, tA, tB, tC);
#endif
        if (flag == 2) break;
        // (flag == 3: the loop comes round once more to publish)
    }
    if (tid == 0) {
        if (ss.done) { ss.model = 3; atomicSub(a.nactive, 1); }
        st = ss;
    }
}
