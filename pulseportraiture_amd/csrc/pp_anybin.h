// Rows of ANY even length (numpy.fft.rfft takes every nbin: pptoaslib.py:976-979).
//
// The tuned transforms of pp_kernels.h / pp_xspec1024*.h are plans for powers of two.  Every
// other even row length B = 2 M (M <= 2048) goes through Bluestein's identity here: the complex
// DFT of the M packed pairs z_j = x_2j + i x_2j+1,
//     Z_k = w_k sum_j (z_j w_j) conj(w_{k-j}),      w_j = exp(-i pi j^2 / M),
// is a circular convolution of length L = 2^n >= 2 M - 1, taken with the power-of-two transform
// of pp_fft.h (two transforms of L points per row; the transform of the chirp is a per-length
// table built on the host in extended precision, j^2 reduced mod 2 M in integers), followed by
// the usual even/odd split d_k = E - i W_B^k O.  One workgroup per row; the harmonics are used
// straight from LDS by the same sums the tuned kernels form (S_d, the noise tail, the stored
// cross-spectrum, the first evaluation's sums or the 12 Taylor sums), so everything downstream
// of the transform -- solvers, post-fit stage, seeds -- is shared.  The template spectrum of such
// a slot is pitched to Mp = M rounded up to 64 with zeros beyond M: the kept-harmonic counts stay
// multiples of 64 as the evaluators assume, and the padding contributes nothing.
// A compatibility path: ~3 x the LDS traffic and instructions of a tuned plan per sample.
#pragma once
#include "pp_kernels.h"
#include "pp_extra.h"

namespace pp {

struct AnyArgs {
    int nbin, M, Mp;          // true row length, harmonics M = nbin / 2, pitch of the slot's spectrum rows
    const cplx* chirp;        // [M]  w_j
    const cplx* bft;          // [L]  transform of the wrapped conj chirp
    const cplx* twL;          // twiddles of the L-point transform (table of 2 L)
    const cplx* twB;          // [M + 1] W_B^k of the true row length (the split)
    int mode;                 // -1: harmonics 0..M to hout; 0..3 as k_xspec's MODE
    int tail;                 // measure the noise from the top quarter of the power spectrum
    cplx* hout;               // mode -1: [nrows][M + 1]
    const unsigned char* mask;   // [nsub][nchan_full] rows to skip, or nullptr
};

template <int L, typename Tin>
__global__ __launch_bounds__(FftPlan<L>::T) void k_any(XspecArgs a, AnyArgs g) {
    constexpr int T = FftPlan<L>::T, PL = FftPlan<L>::PADLOG, NW = T / 64;
    __shared__ cplx lds[FftPlan<L>::LDS_ELEMS];
    __shared__ cplx buf[L];
    __shared__ double red[NW * 16];
    const int tid = threadIdx.x;
    const int M = g.M, H = M + 1;
    const int kc = (int)(0.75 * H);          // get_noise_PS: int((1 - 1/4) * len(pows))
    const long long nrows = (long long)a.nsub * a.nchan;
    const double invL = 1.0 / (double)L;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = (int)(row / a.nsub), i = (int)(row % a.nsub);
        const int ia = sub_of(a.act, i), ne = a.coff + n * a.cstep;      // true subint, channel
        const size_t rc = (size_t)ia * a.nchan_full + ne;
        const size_t rx = (size_t)i * a.nchan + n;                       // compact row of X
        if (g.mask && !g.mask[rc]) continue;                             // (uniform over the workgroup)
        const Tin* x = reinterpret_cast<const Tin*>(a.data) + rc * (size_t)g.nbin;
        // ---- a_j = z_j w_j, zero-padded ----
        for (int j = tid; j < L; j += T) {
            cplx v = make_double2(0.0, 0.0);
            if (j < M) v = cmul(make_double2((double)x[2 * j], (double)x[2 * j + 1]), g.chirp[j]);
            buf[j] = v;
        }
        __syncthreads();
        fft_row<L, cplx>(lds, buf, g.twL, tid);
        __syncthreads();
        // ---- convolution with the chirp: conj(A_k Bf_k), transformed again ----
        for (int k = tid; k < L; k += T) {
            const cplx p = cmul(lds[lds_pad<PL>(k)], g.bft[k]);
            buf[k] = make_double2(p.x, -p.y);
        }
        __syncthreads();
        fft_row<L, cplx>(lds, buf, g.twL, tid);
        __syncthreads();
        // ---- Z_k = w_k conj(.) / L ----
        for (int k = tid; k < M; k += T) {
            const cplx c = lds[lds_pad<PL>(k)];
            buf[k] = cmul(make_double2(c.x * invL, -c.y * invL), g.chirp[k]);
        }
        __syncthreads();
        // harmonic k of the real transform: d_k = E - i W_B^k O; d_0, d_M from Z_0
        auto harm = [&](int k) -> cplx {
            const cplx z0 = buf[0];
            if (k == 0) return make_double2(z0.x + z0.y, 0.0);
            if (k == M) return make_double2(z0.x - z0.y, 0.0);
            const cplx zk = buf[k];
            cplx zc = buf[M - k];
            zc.y = -zc.y;
            const cplx E = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
            const cplx O = make_double2(0.5 * (zk.x - zc.x), 0.5 * (zk.y - zc.y));
            const cplx wo = cmul(g.twB[k], O);
            return make_double2(E.x + wo.y, E.y - wo.x);
        };
        if (g.mode < 0) {
            for (int k = tid; k <= M; k += T) g.hout[(size_t)row * H + k] = harm(k);
            __syncthreads();
            continue;
        }
        const cplx* mrow = as_global(a.slot ? a.mft[a.slot[ia]] : a.mft0) + (size_t)ne * g.Mp;
        const int ktp = a.ktab ? as_global(a.slot ? a.ktab[a.slot[ia]] : a.kt0)[ne] : a.Kt;     // (a multiple of 64, <= Mp)
        const int ktn = min(M, ktp);
        const double phin = (g.mode != 0) ? a.ph0[rc] : 0.0;
        double acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0;
        // acc: 0..11 Taylor sums (modes 2, 3) or 0..2 = s0, s1, s2 (mode 1); 12 = S_d; 13 = noise tail
        for (int k = 1 + tid; k <= M; k += T) {
            const cplx d = harm(k);
            const double pw = cnorm(d);
            acc[12] += pw;
            if (g.tail && k >= kc) acc[13] += pw;
            if (k > ktn) continue;
            const cplx xk = cmulc(d, mrow[k - 1]);
            if (g.mode <= 1) store_x(a, rx, k, xk);
            if (g.mode == 0) continue;
            const cplx z = cmul(xk, unit_phasor((double)k, phin));
            if (g.mode == 1) {
                const double kk = (double)k;
                acc[0] += z.x;
                acc[1] = fma(kk, z.y, acc[1]);
                acc[2] = fma(kk * kk, z.x, acc[2]);
            } else {
                static_assert(PP_TJ == 10, "power ladder written for order 10");
                const double kap = PP_TWO_PI * (double)k;
                const double p2 = kap * kap, p4 = p2 * p2, p6 = p4 * p2, p8 = p4 * p4, p10 = p8 * p2;
                const double ui = z.y * kap;
                acc[0] += z.x;
                acc[1] += ui;
                acc[2] = fma(p2, z.x, acc[2]);
                acc[3] = fma(p2, ui, acc[3]);
                acc[4] = fma(p4, z.x, acc[4]);
                acc[5] = fma(p4, ui, acc[5]);
                acc[6] = fma(p6, z.x, acc[6]);
                acc[7] = fma(p6, ui, acc[7]);
                acc[8] = fma(p8, z.x, acc[8]);
                acc[9] = fma(p8, ui, acc[9]);
                acc[10] = fma(p10, z.x, acc[10]);
                acc[11] = fma(p10 * kap, fabs(xk.x) + fabs(xk.y), acc[11]);
            }
        }
        // the stored row runs to the kept-harmonic count, which is rounded up past M: zeros there
        if (g.mode <= 1)
            for (int k = M + 1 + tid; k <= ktp; k += T) store_x(a, rx, k, make_double2(0.0, 0.0));
        // ---- workgroup totals (fixed order: deterministic) ----
#pragma unroll
        for (int q = 0; q < 14; ++q) acc[q] = group_sum<64>(acc[q]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 14; ++q) red[(tid >> 6) * 16 + q] = acc[q];
        }
        __syncthreads();
        if (tid < 14) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v += red[w * 16 + tid];
            if (tid == 12) a.sdraw[rc] = v;
            else if (tid == 13) { if (g.tail) a.noise[rc] = sqrt(v / (2.0 * M) / (double)(H - kc)); }
            else if (g.mode >= 2) {
                // Re(i^q z): +Re, -Im, -Re, +Im, ...
                if (tid < PP_TSTRIDE)
                    a.tay[tay_idx(rc, tid)] = (tid <= PP_TJ && ((tid & 3) == 1 || (tid & 3) == 2)) ? -v : v;
            } else if (g.mode == 1 && tid < 3) {
                a.csum0[rc * 3 + tid] = (tid == 0) ? v : (tid == 1 ? -PP_TWO_PI * v : -PP_TWO_PI * PP_TWO_PI * v);
            }
        }
        __syncthreads();
    }
}

// template spectrum of a slot from the harmonics k_any left in hout[nchan][M + 1]: rows pitched
// to Mp with zeros beyond M, |m|^2, its sum and maximum, the DC term
__global__ __launch_bounds__(64) void k_model_from_harm(const cplx* hout, int M, int Mp, cplx* mft, double* msq, double* msum,
                                                        double* mmax, double* mdc) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const cplx* h = hout + (size_t)n * (M + 1);
    double s = 0.0, mx = 0.0;
    for (int k = 1 + tid; k <= Mp; k += 64) {
        const cplx d = (k <= M) ? h[k] : make_double2(0.0, 0.0);
        const double p = cnorm(d);
        mft[(size_t)n * Mp + (k - 1)] = d;
        msq[(size_t)n * Mp + (k - 1)] = p;
        s += p;
        mx = fmax(mx, p);
    }
    s = group_sum<64>(s);
    mx = group_max<64>(mx);
    if (tid == 0) { msum[n] = s; mmax[n] = mx; mdc[n] = h[0].x; }
}

// Fourier rotation of rows of any even length (k_rotate for general B; rotate_data pplib.py:2338-2426,
// rotate_portrait_full pptoaslib.py:52-81): forward chirp-z transform, harmonics times
// e^{2 pi i k phi_n} (the Nyquist harmonic keeps its real part, as irfft does), the packed
// spectrum of the inverse, and the chirp-z transform once more (x = conj(DFT(conj Z)) / M).
template <int L, typename Tio>
__global__ __launch_bounds__(FftPlan<L>::T) void k_rotate_any(RotateArgs a, AnyArgs g) {
    constexpr int T = FftPlan<L>::T, PL = FftPlan<L>::PADLOG;
    __shared__ cplx lds[FftPlan<L>::LDS_ELEMS];
    __shared__ cplx buf[L];
    const int tid = threadIdx.x;
    const int M = g.M;
    const double invL = 1.0 / (double)L;
    const long long nrows = (long long)a.nsub * a.nchan;
    // one chirp-z transform of the M values in buf (zero-padded to L): DFT_M(buf)[k] -> buf[k]
    auto czt = [&]() {
        for (int j = tid; j < L; j += T) buf[j] = (j < M) ? cmul(buf[j], g.chirp[j]) : make_double2(0.0, 0.0);
        __syncthreads();
        fft_row<L, cplx>(lds, buf, g.twL, tid);
        __syncthreads();
        for (int k = tid; k < L; k += T) {
            const cplx p = cmul(lds[lds_pad<PL>(k)], g.bft[k]);
            buf[k] = make_double2(p.x, -p.y);
        }
        __syncthreads();
        fft_row<L, cplx>(lds, buf, g.twL, tid);
        __syncthreads();
        for (int k = tid; k < M; k += T) {
            const cplx c = lds[lds_pad<PL>(k)];
            buf[k] = cmul(make_double2(c.x * invL, -c.y * invL), g.chirp[k]);
        }
        __syncthreads();
    };
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int i = (int)(row / a.nchan), n = (int)(row % a.nchan);
        const double nu = a.freqs[(size_t)i * a.freqs_stride + n], P = a.P[i];
        const double a2 = 1.0 / (nu * nu);
        const double phin = a.par[i * 3] + PP_DCONST * a.par[i * 3 + 1] * (a2 - a.inv_nuDM2) / P +
                            PP_DCONST * PP_DCONST * a.par[i * 3 + 2] * (a2 * a2 - a.inv_nuGM4) / P;
        const Tio* x = reinterpret_cast<const Tio*>(a.src) + (size_t)row * g.nbin;
        for (int j = tid; j < M; j += T) buf[j] = make_double2((double)x[2 * j], (double)x[2 * j + 1]);
        __syncthreads();
        czt();
        // harmonics d_k = E - i W_B^k O of the real transform from the packed one in buf
        auto harm = [&](int k) -> cplx {
            const cplx zk = buf[k];
            cplx zc = buf[M - k];
            zc.y = -zc.y;
            const cplx E = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
            const cplx O = make_double2(0.5 * (zk.x - zc.x), 0.5 * (zk.y - zc.y));
            const cplx wo = cmul(g.twB[k], O);
            return make_double2(E.x + wo.y, E.y - wo.x);
        };
        const cplx z0 = buf[0];
        const double y0 = z0.x + z0.y;
        const double yM = (z0.x - z0.y) * unit_phasor((double)M, phin).x;
        // rotated harmonics -> packed spectrum of the inverse, conjugated; parked in the (free) image
        for (int k = tid; k < M; k += T) {
            cplx yk, ym;
            if (k == 0) { yk = make_double2(y0, 0.0); ym = make_double2(yM, 0.0); }
            else {
                yk = cmul(harm(k), unit_phasor((double)k, phin));
                ym = cmul(harm(M - k), unit_phasor((double)(M - k), phin));
            }
            ym.y = -ym.y;
            const cplx ev = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y + ym.y));
            cplx od = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y - ym.y));
            cplx w = g.twB[k];
            w.y = -w.y;
            od = cmul(od, w);
            lds[k] = make_double2(ev.x - od.y, -(ev.y + od.x));   // conj(ev + i od)
        }
        __syncthreads();
        for (int k = tid; k < M; k += T) buf[k] = lds[k];
        __syncthreads();
        czt();
        Tio* out = reinterpret_cast<Tio*>(a.dst) + (size_t)row * g.nbin;
        const double inv = 1.0 / (double)M;
        for (int j = tid; j < M; j += T) {
            const cplx r = buf[j];
            out[2 * j] = (Tio)(r.x * inv);
            out[2 * j + 1] = (Tio)(-r.y * inv);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// Round 5: the auxiliary entry points at any even row length (ppalign's accumulation, the per-channel
// chi^2 of the zap proposals, the template synthesisers' scattering filter, the synthetic generator).
// All of them work on HARMONICS: k_any (mode -1) leaves the harmonics 0..M of every row in
// hout[row][M + 1], a small kernel does the entry point's arithmetic per harmonic, and k_irfft_any
// brings rows of harmonics back to the time domain by the chirp-z route.
// --------------------------------------------------------------------------
// one chirp-z transform of the M values in buf (zero-padded to L): DFT_M(buf)[k] -> buf[k]
template <int L>
__device__ __forceinline__ void czt_any(cplx* buf, cplx* lds, const AnyArgs& g, int tid) {
    constexpr int T = FftPlan<L>::T, PL = FftPlan<L>::PADLOG;
    const int M = g.M;
    const double invL = 1.0 / (double)L;
    for (int j = tid; j < L; j += T) buf[j] = (j < M) ? cmul(buf[j], g.chirp[j]) : make_double2(0.0, 0.0);
    __syncthreads();
    fft_row<L, cplx>(lds, buf, g.twL, tid);
    __syncthreads();
    for (int k = tid; k < L; k += T) {
        const cplx p = cmul(lds[lds_pad<PL>(k)], g.bft[k]);
        buf[k] = make_double2(p.x, -p.y);
    }
    __syncthreads();
    fft_row<L, cplx>(lds, buf, g.twL, tid);
    __syncthreads();
    for (int k = tid; k < M; k += T) {
        const cplx c = lds[lds_pad<PL>(k)];
        buf[k] = cmul(make_double2(c.x * invL, -c.y * invL), g.chirp[k]);
    }
    __syncthreads();
}

// conj of the packed spectrum of the inverse real transform: slot k of the M complex values whose
// DFT (conjugated, / M) is the row; yk = harmonic k, ym = harmonic M - k (k = 0: DC and Nyquist, both real)
__device__ __forceinline__ cplx irfft_pack(cplx yk, cplx ym, cplx wk) {
    ym.y = -ym.y;
    const cplx ev = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y + ym.y));
    cplx od = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y - ym.y));
    wk.y = -wk.y;
    od = cmul(od, wk);
    return make_double2(ev.x - od.y, -(ev.y + od.x));   // conj(ev + i od)
}

// numpy.fft.irfft of rows of harmonics h[row][M + 1] (the imaginary parts of h_0 and h_M are ignored, as
// irfft ignores them): out[row][nbin]
template <int L>
__global__ __launch_bounds__(FftPlan<L>::T) void k_irfft_any(const cplx* harm, AnyArgs g, long long nrows, double* out) {
    constexpr int T = FftPlan<L>::T;
    __shared__ cplx lds[FftPlan<L>::LDS_ELEMS];
    __shared__ cplx buf[L];
    const int tid = threadIdx.x;
    const int M = g.M;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const cplx* h = harm + (size_t)row * (M + 1);
        for (int k = tid; k < M; k += T) {
            const cplx yk = (k == 0) ? make_double2(h[0].x, 0.0) : h[k];
            const cplx ym = (k == 0) ? make_double2(h[M].x, 0.0) : h[M - k];
            buf[k] = irfft_pack(yk, ym, g.twB[k]);
        }
        __syncthreads();
        czt_any<L>(buf, lds, g, tid);
        double* o = out + (size_t)row * g.nbin;
        const double inv = 1.0 / (double)M;
        for (int j = tid; j < M; j += T) {
            const cplx r = buf[j];
            o[2 * j] = r.x * inv;
            o[2 * j + 1] = -r.y * inv;
        }
        __syncthreads();
    }
}

// k_synth for general row lengths: dst = gain irfft(m_k e^{2 pi i k phi_n}) + sigma N(0, 1) from the slot's
// spectrum (rows pitched to Mp)
template <int L, typename Tout>
__global__ __launch_bounds__(FftPlan<L>::T) void k_synth_any(SynthArgs a, AnyArgs g) {
    constexpr int T = FftPlan<L>::T;
    __shared__ cplx lds[FftPlan<L>::LDS_ELEMS];
    __shared__ cplx buf[L];
    const int tid = threadIdx.x;
    const int M = g.M;
    const long long nrows = (long long)a.nsub * a.nchan;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int i = (int)(row / a.nchan), n = (int)(row % a.nchan);
        const double nu = a.freqs[n], P = a.P[i];
        const double a2 = 1.0 / (nu * nu);
        const double phin = -(a.inj[i * 3] + PP_DCONST * a.inj[i * 3 + 1] * a2 / P +
                              PP_DCONST * PP_DCONST * a.inj[i * 3 + 2] * a2 * a2 / P);
        const cplx* mrow = a.mft + (size_t)n * g.Mp;
        const double yM = cmul(mrow[M - 1], unit_phasor((double)M, phin)).x;   // Nyquist: real part
        for (int k = tid; k < M; k += T) {
            cplx yk, ym;
            if (k == 0) { yk = make_double2(a.mdc[n], 0.0); ym = make_double2(yM, 0.0); }
            else {
                yk = cmul(mrow[k - 1], unit_phasor((double)k, phin));
                ym = cmul(mrow[M - k - 1], unit_phasor((double)(M - k), phin));
            }
            buf[k] = irfft_pack(yk, ym, g.twB[k]);
        }
        __syncthreads();
        czt_any<L>(buf, lds, g, tid);
        Tout* out = reinterpret_cast<Tout*>(a.dst) + (size_t)row * g.nbin;
        const double inv = (a.gain ? a.gain[row] : 1.0) / (double)M;
        for (int j = tid; j < M; j += T) {
            const cplx r = buf[j];
            double z0, z1;
            normal_pair(a.seed, a.first_subint + i, n, j, z0, z1);
            out[2 * j] = (Tout)(r.x * inv + a.sigma * z0);
            out[2 * j + 1] = (Tout)(-r.y * inv + a.sigma * z1);
        }
        __syncthreads();
    }
}

// ppalign's accumulation (ppalign.py:199-206; k_align_accum for general row lengths): spec[n][k] (+)= sum_i w_in
// d_ink e^{2 pi i k phi_in} over the ns subints of this chunk, in index order; hout is k_any's, channel-major
// (row = n ns + i); the DC harmonic is unchanged, the Nyquist harmonic keeps its real part (irfft)
__global__ __launch_bounds__(256) void k_align_harm(const cplx* hout, AlignArgs a, int s0, int ns, int M, cplx* spec,
                                                    double* totw, int first) {
    const int n = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M) return;
    const int H = M + 1;
    cplx acc = first ? make_double2(0.0, 0.0) : spec[(size_t)n * H + k];
    double wsum = first ? 0.0 : totw[n];
    for (int ii = 0; ii < ns; ++ii) {
        const int i = s0 + ii;
        const double w = a.w[(size_t)i * a.nchan + n];
        if (w == 0.0 || w != w) continue;
        const double nu = a.freqs[(size_t)i * a.freqs_stride + n], P = a.P[i];
        const double phase = a.par[i * 3], DM = a.par[i * 3 + 1], nuref = a.par[i * 3 + 2];
        // reference order of operations (pplib.py:2419-2424)
        const double D = PP_DCONST * DM / P;
        const double iref = (nuref == INFINITY) ? 0.0 : 1.0 / (nuref * nuref);
        const double phin = (DM == 0.0) ? phase : phase + D * (1.0 / (nu * nu) - iref);
        const cplx d = hout[((size_t)n * ns + ii) * H + k];
        cplx y;
        if (k == 0) y = make_double2(d.x, 0.0);
        else {
            const cplx e = unit_phasor((double)k, phin);
            y = (k == M) ? make_double2(d.x * e.x, 0.0) : cmul(d, e);
        }
        acc.x = fma(w, y.x, acc.x);
        acc.y = fma(w, y.y, acc.y);
        wsum += w;
    }
    spec[(size_t)n * H + k] = acc;
    if (k == 0) totw[n] = wsum;
}

// per-channel reduced chi^2 (k_chan_chi2 for general row lengths) from k_any's harmonics of the data rows,
// channel-major (row = n ns + i); the slot's spectrum rows are pitched to Mp
__global__ __launch_bounds__(256) void k_chan_chi2_harm(const cplx* hout, ChanChi2Args a, int s0, int ns, int M, int Mp) {
    __shared__ double scratch[4];
    const int tid = threadIdx.x;
    const int H = M + 1;
    const long long nrows = (long long)ns * a.nchan;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int ii = (int)(row / a.nchan), n = (int)(row % a.nchan), i = s0 + ii;
        const double nu = a.freqs[(size_t)i * a.freqs_stride + n], P = a.P[i];
        const double* pr = a.par + (size_t)i * 5;
        const double nuDM = a.nuref[i * 3], nuGM = a.nuref[i * 3 + 1], nutau = a.nuref[i * 3 + 2];
        const double a2 = 1.0 / (nu * nu);
        const double iDM = (nuDM == INFINITY) ? 0.0 : 1.0 / (nuDM * nuDM);
        const double iGM = (nuGM == INFINITY) ? 0.0 : 1.0 / (nuGM * nuGM * nuGM * nuGM);
        const double phin = pr[0] + PP_DCONST * pr[1] * (a2 - iDM) / P +
                            PP_DCONST * PP_DCONST * pr[2] * (a2 * a2 - iGM) / P;
        const double taun = (pr[3] != 0.0) ? pr[3] * pow(nu / nutau, pr[4]) : 0.0;
        const int sl = a.slot ? a.slot[i] : 0;
        const cplx* mrow = as_global(a.mft[sl]) + (size_t)n * Mp;
        const double m0 = as_global(a.mdc[sl])[n];
        const double sc = a.scales[(size_t)i * a.nchan + n], sg = a.errs[(size_t)i * a.nchan + n];
        const cplx* h = hout + ((size_t)n * ns + ii) * H;
        double sum = 0.0;
        for (int k = tid; k <= M; k += 256) {
            double wgt = 2.0;
            cplx d, m;
            if (k == 0) { d = make_double2(h[0].x, 0.0); m = make_double2(m0, 0.0); wgt = 1.0; }
            else {
                d = (k == M) ? make_double2(h[M].x, 0.0) : h[k];
                d = cmul(d, unit_phasor((double)k, phin));
                m = mrow[k - 1];
                if (taun != 0.0) {
                    const double x = PP_TWO_PI * (double)k * taun, den = 1.0 / (1.0 + x * x);
                    m = cmul(m, make_double2(den, -x * den));
                }
            }
            cplx r = make_double2(d.x - sc * m.x, d.y - sc * m.y);
            if (k == M) { r.y = 0.0; wgt = 1.0; }     // irfft drops the imaginary part at Nyquist
            sum = fma(wgt, cnorm(r), sum);
        }
        sum = group_sum<64>(sum);
        if ((tid & 63) == 0) scratch[tid >> 6] = sum;
        __syncthreads();
        if (tid == 0) {
            const double tot = ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
            a.out[(size_t)i * a.nchan + n] = tot / (2.0 * M) / (sg * sg) / (double)(2 * M - 2);
        }
        __syncthreads();
    }
}

// the scattering filter of a Gaussian-component template at general row lengths: h_nk <- h_nk / (1 + 2 pi i k tau_n),
// tau_n = tau_ref (nu_n / nu_ref)^alpha; DC unchanged, Nyquist Re(d_M B_M) with d_M real (k_gauss_portrait)
__global__ __launch_bounds__(256) void k_scatter_harm(cplx* harm, const double* freqs, double nu_ref, double tau_ref,
                                                      double alpha, int M) {
    const int n = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M || k == 0) return;
    const double taun = tau_ref * pow(freqs[n] / nu_ref, alpha);
    cplx* h = harm + (size_t)n * (M + 1);
    const double xk = PP_TWO_PI * (double)k * taun, dk = 1.0 / (1.0 + xk * xk);
    if (k == M) h[M] = make_double2(h[M].x * dk, 0.0);
    else h[k] = cmul(h[k], make_double2(dk, -xk * dk));
}

// The reference's rot_prof from the harmonics k_any left in hout (general row lengths; pp_reference_phase_seed):
// spec[i][k] = sum_n w_n d_nk e^{2 pi i k phi'_n} / sum_n w_n, phi'_n the rotation of rotate_data
// (pplib.py:2338-2426).  hout rows are channel-major (row = n nsub + i).  One thread per harmonic,
// channels in order: a fixed summation order.
__global__ __launch_bounds__(256) void k_rot_mean_harm(const cplx* hout, RotMeanArgs a, int M, cplx* spec) {
    const int i = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M) return;
    const double P = a.P[i];
    cplx s = make_double2(0.0, 0.0);
    double wt = 0.0;
    for (int n = 0; n < a.nchan; ++n) {
        const double w = a.w[(size_t)i * a.nchan + n];
        if (w == 0.0) continue;
        wt += w;
        const double nu = a.freqs[(size_t)i * a.freqs_stride + n];
        const double a2 = 1.0 / (nu * nu);
        const double phin = a.par[i * 3] + PP_DCONST * a.par[i * 3 + 1] * (a2 - a.inv_nuDM2) / P +
                            PP_DCONST * PP_DCONST * a.par[i * 3 + 2] * (a2 * a2 - a.inv_nuGM4) / P;
        const cplx d = hout[((size_t)n * a.nsub + i) * (M + 1) + k];
        cplx y;
        if (k == 0) y = d;
        else {
            const cplx e = unit_phasor((double)k, phin);
            y = (k == M) ? make_double2(d.x * e.x, 0.0) : cmul(d, e);
        }
        s.x = fma(w, y.x, s.x);
        s.y = fma(w, y.y, s.y);
    }
    const double inv = (wt > 0.0) ? 1.0 / wt : 0.0;
    spec[(size_t)i * (M + 1) + k] = make_double2(s.x * inv, s.y * inv);
}

}  // namespace pp
