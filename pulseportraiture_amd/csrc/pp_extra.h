// Kernels beside the main fit: 1-D FFTFIT (fit_phase_shift, pplib.py:2054-2100)
// and the on-device synthetic portrait generator.
#pragma once
#include "pp_kernels.h"

namespace pp {

// --------------------------------------------------------------------------
// counter-based RNG: Philox4x32-10 (Salmon et al. 2011), keyed on the seed,
// counter = (bin pair, channel, subint lo, subint hi)
// --------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// two independent N(0,1) deviates for (subint, channel, bin pair)
__device__ __forceinline__ void normal_pair(uint64_t seed, int64_t sub, int chan, int pair, double& z0, double& z1) {
    uint32_t c[4] = {(uint32_t)pair, (uint32_t)chan, (uint32_t)((uint64_t)sub & 0xffffffffu),
                     (uint32_t)((uint64_t)sub >> 32)};
    philox4x32_10(c, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32));
    const uint64_t a = ((uint64_t)c[0] << 32) | c[1], b = ((uint64_t)c[2] << 32) | c[3];
    const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double s, co;
    sincospi(2.0 * u2, &s, &co);
    z0 = r * co;
    z1 = r * s;
}

struct SynthArgs {
    const cplx* mft;      // [nchan][M] harmonics 1..M of the model slot
    const double* mdc;    // [nchan] DC harmonic (real)
    void* dst;            // [nsub][nchan][B]
    const double* freqs;  // [nchan]
    const double* P;      // [nsub]
    const double* inj;    // [nsub][3] phi, DM, GM (reference frequency infinity)
    const cplx* twB;
    double sigma;
    uint64_t seed;
    int64_t first_subint;
    int nsub, nchan;
    const double* gain;   // [nsub][nchan] amplitude of the template in every channel (scintillation), or nullptr = 1
};

// dst = gain irfft(m_k e^{2 pi i k phi_n}) + sigma N(0,1); phi_n = -(phi + DM and GM
// delays): data "delayed by" the injected values (pptoaslib.py:52-81, 57-58)
template <int M, typename Tout>
__global__ __launch_bounds__(FftPlan<M>::T) void k_synth(SynthArgs a) {
    constexpr int T = FftPlan<M>::T;
    constexpr int PL = FftPlan<M>::PADLOG;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    __shared__ cplx zin[M];
    const int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int i = (int)(row / a.nchan), n = (int)(row % a.nchan);
        const double nu = a.freqs[n], P = a.P[i];
        const double a2 = 1.0 / (nu * nu);
        const double phin = -(a.inj[i * 3] + PP_DCONST * a.inj[i * 3 + 1] * a2 / P +
                              PP_DCONST * PP_DCONST * a.inj[i * 3 + 2] * a2 * a2 / P);
        const cplx* mrow = a.mft + (size_t)n * M;
        const double yM = cmul(mrow[M - 1], unit_phasor((double)M, phin)).x;   // Nyquist: real part
        for (int k = tid; k < M; k += T) {
            cplx yk, ym;
            if (k == 0) { yk = make_double2(a.mdc[n], 0.0); ym = make_double2(yM, 0.0); }
            else {
                yk = cmul(mrow[k - 1], unit_phasor((double)k, phin));
                ym = cmul(mrow[M - k - 1], unit_phasor((double)(M - k), phin));
            }
            ym.y = -ym.y;
            const cplx ev = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y + ym.y));
            cplx od = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y - ym.y));
            cplx w = a.twB[k];
            w.y = -w.y;
            od = cmul(od, w);
            // Z = ev + i od ; store conj(Z)
            zin[k] = make_double2(ev.x - od.y, -(ev.y + od.x));
        }
        __syncthreads();
        fft_row<M, cplx>(lds, zin, a.twB, tid);
        Tout* out = reinterpret_cast<Tout*>(a.dst) + (size_t)row * (2 * M);
        const double inv = (a.gain ? a.gain[row] : 1.0) / (double)M;
        for (int j = tid; j < M; j += T) {
            const cplx r = lds[lds_pad<PL>(j)];
            double z0, z1;
            normal_pair(a.seed, a.first_subint + i, n, j, z0, z1);
            const double x0 = r.x * inv + a.sigma * z0, x1 = -r.y * inv + a.sigma * z1;
            if (sizeof(Tout) == 8) reinterpret_cast<double2*>(out)[j] = make_double2(x0, x1);
            else reinterpret_cast<float2*>(out)[j] = make_float2((float)x0, (float)x1);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// Fourier rotation of portraits (rotate_data pplib.py:2338-2426,
// rotate_portrait_full pptoaslib.py:52-81): every row is transformed, its
// harmonics multiplied by e^{2 pi i k phi_n}, phi_n = phi + Dconst DM
// (nu_n^-2 - nu_DM^-2)/P + Dconst^2 GM (nu_n^-4 - nu_GM^-4)/P, and transformed
// back -- both FFTs of a row stay in LDS.  Positive values rotate to earlier
// phase.
// --------------------------------------------------------------------------
struct RotateArgs {
    const void* src;      // [nsub][nchan][B]
    void* dst;            // [nsub][nchan][B] (may alias src)
    const double* freqs; long long freqs_stride;
    const double* P;      // [nsub] (1.0 where delays are already in rotations)
    const double* par;    // [nsub][3] phi, DM, GM
    const cplx* twB;
    double inv_nuDM2, inv_nuGM4;   // 1/nu_DM^2, 1/nu_GM^4 (0 for infinite reference)
    int nsub, nchan;
};

template <int M, typename Tio>
__global__ __launch_bounds__(FftPlan<M>::T) void k_rotate(RotateArgs a) {
    constexpr int T = FftPlan<M>::T;
    constexpr int PL = FftPlan<M>::PADLOG;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    __shared__ cplx zin[M];
    const int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int i = (int)(row / a.nchan), n = (int)(row % a.nchan);
        const double nu = a.freqs[(size_t)i * a.freqs_stride + n], P = a.P[i];
        const double a2 = 1.0 / (nu * nu);
        const double phin = a.par[i * 3] + PP_DCONST * a.par[i * 3 + 1] * (a2 - a.inv_nuDM2) / P +
                            PP_DCONST * PP_DCONST * a.par[i * 3 + 2] * (a2 * a2 - a.inv_nuGM4) / P;
        fft_row<M, Tio>(lds, reinterpret_cast<const Tio*>(a.src) + (size_t)row * (2 * M), a.twB, tid);
        // rotated harmonics Y_k, k = 0..M, then the packed spectrum of the inverse
        const cplx z0 = lds[0];
        const double y0 = z0.x + z0.y;                                   // DC, unchanged
        const double yM = (z0.x - z0.y) * unit_phasor((double)M, phin).x; // Nyquist: real part kept
        for (int k = tid; k < M; k += T) {
            cplx yk, ym;
            if (k == 0) { yk = make_double2(y0, 0.0); ym = make_double2(yM, 0.0); }
            else {
                yk = cmul(rfft_harmonic<M>(lds, a.twB, k), unit_phasor((double)k, phin));
                ym = cmul(rfft_harmonic<M>(lds, a.twB, M - k), unit_phasor((double)(M - k), phin));
            }
            ym.y = -ym.y;
            const cplx ev = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y + ym.y));
            cplx od = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y - ym.y));
            cplx w = a.twB[k];
            w.y = -w.y;
            od = cmul(od, w);
            zin[k] = make_double2(ev.x - od.y, -(ev.y + od.x));   // conj(ev + i od)
        }
        __syncthreads();
        fft_row<M, cplx>(lds, zin, a.twB, tid);
        Tio* out = reinterpret_cast<Tio*>(a.dst) + (size_t)row * (2 * M);
        const double inv = 1.0 / (double)M;
        for (int j = tid; j < M; j += T) {
            const cplx r = lds[lds_pad<PL>(j)];
            const double x0 = r.x * inv, x1 = -r.y * inv;
            if (sizeof(Tio) == 8) reinterpret_cast<double2*>(out)[j] = make_double2(x0, x1);
            else reinterpret_cast<float2*>(out)[j] = make_float2((float)x0, (float)x1);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// The data side of the reference's initial guess (pptoas.py:421-423) in one read of
// the portraits:  rot_prof = np.average(rotate_data(portx, 0, DM_guess, P, freqsx,
// nu_mean), axis=0, weights=weightsx).  Rotation and the channel mean are linear, so
// the mean is taken in the Fourier domain: every workgroup transforms a run of
// channels of one subint and accumulates  w_n d_nk e^{2 pi i k phi_n}  in registers
// (k = tid, tid + T, ...), channels of zero weight are not even read; a second
// kernel adds the runs and divides by the summed weights.  The Nyquist harmonic
// keeps its real part only, as rotate_data's irfft does.
// --------------------------------------------------------------------------
struct RotMeanArgs {
    const void* src;       // [nsub][nchan][B]
    const double* freqs; long long freqs_stride;
    const double* P;       // [nsub]
    const double* par;     // [nsub][3] phi, DM, GM
    const double* w;       // [nsub][nchan]
    const cplx* twB;
    double inv_nuDM2, inv_nuGM4;
    cplx* part;            // [nsub][nrun][M + 1]
    double* wpart;         // [nsub][nrun]
    int nsub, nchan, nrun, cpr;   // channels per run
};

template <int M, typename Tio>
__global__ __launch_bounds__(FftPlan<M>::T) void k_rot_mean(RotMeanArgs a) {
    constexpr int T = FftPlan<M>::T, R1 = FftPlan<M>::R1, PER1 = FftPlan<M>::PER1;
    constexpr int KPT = M / T + 1;            // harmonics 0..M over T lanes
    typedef typename RawOf<Tio>::type Raw;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    const int tid = threadIdx.x;
    const int i = blockIdx.x / a.nrun, run = blockIdx.x % a.nrun;
    const double P = a.P[i];
    const double* wrow = a.w + (size_t)i * a.nchan;
    const Tio* base = reinterpret_cast<const Tio*>(a.src) + (size_t)i * a.nchan * (2 * M);
    cplx acc[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) acc[j] = make_double2(0.0, 0.0);
    double wsum = 0.0;
    const int n0 = run * a.cpr, n1 = min(a.nchan, n0 + a.cpr);
    // next channel of non-zero weight at or after n (n1 if none)
    auto next_good = [&](int n) { while (n < n1 && wrow[n] == 0.0) ++n; return n; };
    RowTwiddles<M> tw;
    load_row_twiddles<M>(tw, a.twB, tid);
    // W_B^tid and W_B^T: split twiddles by recurrence
    const cplx wb0 = a.twB[min(tid, M)], wbT = a.twB[min(T, M)];
    Raw cur[PER1][R1];
    int n = next_good(n0);
    if (n < n1) stage_load_global<M, T, R1>(cur, base + (size_t)n * (2 * M), tid);
    while (n < n1) {
        const double w = wrow[n];
        wsum += w;
        const double nu = a.freqs[(size_t)i * a.freqs_stride + n];
        const double a2 = 1.0 / (nu * nu);
        const double phin = a.par[i * 3] + PP_DCONST * a.par[i * 3 + 1] * (a2 - a.inv_nuDM2) / P +
                            PP_DCONST * PP_DCONST * a.par[i * 3 + 2] * (a2 * a2 - a.inv_nuGM4) / P;
        {
            cplx v[PER1][R1];
#pragma unroll
            for (int ii = 0; ii < PER1; ++ii)
#pragma unroll
                for (int k = 0; k < R1; ++k) v[ii][k] = to_cplx(cur[ii][k]);
            fft_first_stage<M>(lds, v, tw, tid);
        }
        // the first stage has consumed the row: its registers receive the next one,
        // whose loads stay in flight under the rest of this row (as in k_xspec)
        const int nn = next_good(n + 1);
        stage_load_global<M, T, R1>(cur, base + (size_t)(nn < n1 ? nn : n) * (2 * M), tid);
        fft_later_stages<M>(lds, tw, tid);
        const cplx z0 = lds[0];
        cplx e = unit_phasor((double)tid, phin);
        const cplx wT = unit_phasor((double)T, phin);
        cplx wb = wb0;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = tid + j * T;
            if (k <= M) {
                cplx y;
                if (k == 0) y = make_double2(z0.x + z0.y, 0.0);
                else if (k == M) y = make_double2((z0.x - z0.y) * e.x, 0.0);
                else y = cmul(rfft_harmonic_w<M>(lds, wb, k), e);
                acc[j].x = fma(w, y.x, acc[j].x);
                acc[j].y = fma(w, y.y, acc[j].y);
            }
            e = cmul(e, wT);
            wb = cmul(wb, wbT);
        }
        lds_sync<T>();          // the image is rewritten by the next row
        n = nn;
    }
    cplx* out = a.part + ((size_t)i * a.nrun + run) * (M + 1);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int k = tid + j * T;
        if (k <= M) out[k] = acc[j];
    }
    if (tid == 0) a.wpart[(size_t)i * a.nrun + run] = wsum;
}

// k_rot_mean for 2048-bin rows on the one-exchange FFT (pp_fftq.h): lane t ends with
// Z[lam + 64 kd] in register kd and accumulates harmonics k = lam + 64 kd (kd = 0..15;
// the lane that owns lam = 0 also takes the Nyquist harmonic).  The even/odd split needs
// Z_{M-k} = Z[(64 - lam) + 64 (15 - kd)]: every lane publishes its 16 registers and reads
// 16 values of its partner lane -- 64 LDS instructions per row where the Stockham plan
// takes 130.
template <typename Tio>
__global__ __launch_bounds__(64, 2) void k_rot_mean_q1024(RotMeanArgs a) {
    constexpr int M = 1024, T = 64, R1 = 16;
    typedef typename RawOf<Tio>::type Raw;
    __shared__ cplx lds[FFTQ_LDS_ELEMS];
    int tid = threadIdx.x;
    const int i = blockIdx.x / a.nrun, run = blockIdx.x % a.nrun;
    const double P = a.P[i];
    const double* wrow = a.w + (size_t)i * a.nchan;
    const Tio* base = reinterpret_cast<const Tio*>(a.src) + (size_t)i * a.nchan * (2 * M);
    cplx acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = make_double2(0.0, 0.0);
    double accN = 0.0, wsum = 0.0;
    const int n0 = run * a.cpr, n1 = min(a.nchan, n0 + a.cpr);
    // next channel of non-zero weight at or after n (n1 if none)
    auto next_good = [&](int n) { while (n < n1 && wrow[n] == 0.0) ++n; return n; };
    const cplx wbT = a.twB[64];
    Raw cur[1][R1];
    int n = next_good(n0);
    if (n < n1) stage_load_global<M, T, R1>(cur, base + (size_t)n * (2 * M), tid);
    while (n < n1) {
        asm volatile("" : "+v"(tid));
        const int lam = fftq_lambda(tid);
        const bool l0 = (lam == 0);
        // twiddles re-read per row (L1-resident, older than the prefetch)
        const cplx t1 = as_global(a.twB)[2 * tid], t2 = as_global(a.twB)[32 * (tid & 15)];
        const cplx wb0 = as_global(a.twB)[lam];
        const double w = wrow[n], hw = 0.5 * w;
        wsum += w;
        const double nu = a.freqs[(size_t)i * a.freqs_stride + n];
        const double a2 = 1.0 / (nu * nu);
        const double phin = a.par[i * 3] + PP_DCONST * a.par[i * 3 + 1] * (a2 - a.inv_nuDM2) / P +
                            PP_DCONST * PP_DCONST * a.par[i * 3 + 2] * (a2 * a2 - a.inv_nuGM4) / P;
        cplx v[R1];
#pragma unroll
        for (int k = 0; k < R1; ++k) v[k] = to_cplx(cur[0][k]);
        const int nn = next_good(n + 1);
        auto prefetch = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            stage_load_global<M, T, R1>(cur, base + (size_t)(nn < n1 ? nn : n) * (2 * M), tid);
            __builtin_amdgcn_sched_barrier(0);
        };
        fftq1024<1>(v, lds, t1, t2, tid, nullptr, prefetch);
        // ---- partners through LDS ----
        {
            cplx* pub = lds + tid;
#pragma unroll
            for (int s = 0; s < 16; ++s) pub[64 * s] = v[s];
            lds_sync<T>();
        }
        // slot kd reads register 15 - kd of the partner lane; the lane that owns lam = 0 is
        // its own partner and reads its register 16 - kd (kd >= 1)
        const cplx* pc = lds + fftq_lane_of((64 - lam) & 63) + (l0 ? 64 : 0);
        // e^{2 pi i lam phi}; lane 0 computes the step e^{2 pi i 64 phi} instead and starts from 1
        const cplx el = unit_phasor<true>(l0 ? 64.0 : (double)lam, phin);
        const cplx wst = make_double2(bcast_lane0(el.x), bcast_lane0(el.y));
        cplx e = l0 ? make_double2(1.0, 0.0) : el;
        cplx wb = wb0;
        const cplx z0 = v[0];      // (lane 0: Z_0)
#pragma unroll
        for (int kd = 0; kd < 16; ++kd) {
            const cplx zk = v[kd];
            cplx zc = pc[64 * (15 - kd)];
            zc.y = -zc.y;
            // 2 d_k = E - i W^k O with E, O unhalved; the half goes into the weight (exact)
            const cplx E = make_double2(zk.x + zc.x, zk.y + zc.y);
            const cplx O = make_double2(zk.x - zc.x, zk.y - zc.y);
            const cplx wo = cmul(wb, O);
            cplx y = cmul(make_double2(E.x + wo.y, E.y - wo.x), e);
            if (kd == 0) {
                // harmonic 0 of lane 0: d_0 = Re Z_0 + Im Z_0
                y.x = l0 ? 2.0 * (z0.x + z0.y) : y.x;
                y.y = l0 ? 0.0 : y.y;
            }
            acc[kd].x = fma(hw, y.x, acc[kd].x);
            acc[kd].y = fma(hw, y.y, acc[kd].y);
            wb = cmul(wb, wbT);
            e = cmul(e, wst);
        }
        // Nyquist (lane 0; e is now e^{2 pi i 1024 phi} there): real part only, as irfft keeps it
        accN = fma(w, (z0.x - z0.y) * e.x, accN);
        lds_sync<T>();          // the image is rewritten by the next row
        n = nn;
    }
    cplx* out = a.part + ((size_t)i * a.nrun + run) * (M + 1);
    const int lam = fftq_lambda(tid);
#pragma unroll
    for (int kd = 0; kd < 16; ++kd) out[lam + 64 * kd] = acc[kd];
    if (lam == 0) out[M] = make_double2(accN, 0.0);
    if (tid == 0) a.wpart[(size_t)i * a.nrun + run] = wsum;
}

// mean spectrum of every subint: sum of its runs / summed weights -> spec[i][0..M]
__global__ void k_rot_mean_finish(const cplx* part, const double* wpart, int nsub, int nrun, int M, cplx* spec) {
    const int i = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M) return;
    double wt = 0.0;
    cplx s = make_double2(0.0, 0.0);
    for (int r = 0; r < nrun; ++r) {
        wt += wpart[(size_t)i * nrun + r];
        const cplx v = part[((size_t)i * nrun + r) * (M + 1) + k];
        s.x += v.x; s.y += v.y;
    }
    const double inv = (wt > 0.0) ? 1.0 / wt : 0.0;
    spec[(size_t)i * (M + 1) + k] = make_double2(s.x * inv, s.y * inv);
}

// --------------------------------------------------------------------------
// Gaussian-component template portraits on the device:
// gen_gaussian_portrait + gaussian_profile + evolve_parameter (pplib.py:853-930,
// 770-825, 996-1046), optionally scattered in the Fourier domain
// (pplib.py:915-922, 4049-4095).  One workgroup per channel: evolve the component
// parameters to the channel frequency, sum the wrapped unit-peak Gaussians on the
// bin centres, and -- if the model carries a scattering time -- filter the row
// with 1/(1 + 2 pi i k tau_n) between a forward and an inverse FFT in LDS.
// The arithmetic follows the host construction (gmodel.py) operation by
// operation; only exp/log may differ from NumPy's by an ulp.
// --------------------------------------------------------------------------
#define PP_MAX_GAUSS 64
struct GaussArgs {
    const double* freqs;      // [nchan]
    const double* comps;      // [ngauss][6] loc, m_loc, wid, m_wid, amp, m_amp
    const cplx* twB;
    double* out;              // [nchan][B]
    double nu_ref, dc, tau_ref /* rot at nu_ref, 0 = unscattered */, alpha;
    int nchan, ngauss;
    int code_loc, code_wid, code_amp;   // 0 = power law, else linear
};

__device__ __forceinline__ double gauss_evolve(double nu, double nu_ref, double value, double evol, int code) {
    // power law in logs (pplib.py:1017-1030), else linear in frequency
    if (code == 0) return exp(add_rn(mul_rn(log(nu) - log(nu_ref), evol), log(value)));
    return add_rn(mul_rn(nu - nu_ref, evol), value);
}

// bin centre j of nbin (get_bin_centers -> numpy.linspace: j*step + start, last = stop)
__device__ __forceinline__ double gauss_bin_centre(int j, int nbin) {
    const double start = 1.0 / (double)(nbin * 2), stop = 1.0 - start;
    if (j == nbin - 1) return stop;
    const double step = (stop - start) / (double)(nbin - 1);
    return add_rn(mul_rn((double)j, step), start);
}

struct GaussComp { double mean, sigma, norm, fact, amp; int on; };

__device__ __forceinline__ double gauss_wrap(double x, double mean) {
    if (mean < 0.5) return (x > mean + 0.5) ? x - 1.0 : x;
    return (x < mean - 0.5) ? x + 1.0 : x;
}
__device__ __forceinline__ double gauss_val(double x, const GaussComp& g) {
    const double z = (gauss_wrap(x, g.mean) - g.mean) / g.sigma;
    return (fabs(z) < 20.0) ? exp(-0.5 * mul_rn(z, z)) / g.norm : 0.0;
}

// the evolved components of channel frequency nu on a row of B bins (threads tid, tid + T, ... of the workgroup)
__device__ __forceinline__ void gauss_components(const GaussArgs& a, double nu, int B, GaussComp* gc, int tid, int T) {
    for (int c = tid; c < a.ngauss; c += T) {
        const double* p = a.comps + c * 6;
        const double loc = gauss_evolve(nu, a.nu_ref, p[0], p[1], a.code_loc);
        const double wid = gauss_evolve(nu, a.nu_ref, p[2], p[3], a.code_wid);
        GaussComp g;
        g.amp = gauss_evolve(nu, a.nu_ref, p[4], p[5], a.code_amp);
        g.on = wid > 0.0;
        g.sigma = (g.on ? wid : 1.0) / (2.0 * sqrt(2.0 * log(2.0)));
        g.mean = loc - floor(loc);                 // loc % 1.0
        g.norm = g.sigma * sqrt(2.0 * PP_TWO_PI * 0.5);   // sigma sqrt(2 pi)
        // peak bin = argmax of the sampled profile (first maximum): the bin
        // centre nearest to the mean, searched among its neighbours
        const int j0 = min(B - 1, max(0, (int)floor(g.mean * (double)B)));
        int jb = -1; double vb = -1.0;
        for (int dj = -1; dj <= 1; ++dj) {
            const int j = (j0 + dj + B) % B;
            const double v = gauss_val(gauss_bin_centre(j, B), g);
            if (v > vb || (v == vb && j < jb)) { vb = v; jb = j; }
        }
        const double zpk = (gauss_wrap(gauss_bin_centre(jb, B), g.mean) - loc) / g.sigma;
        g.fact = (vb > 0.0) ? exp(-0.5 * mul_rn(zpk, zpk)) / vb : 0.0;
        gc[c] = g;
    }
}
// the value of the (unscattered) template row at bin j of B
__device__ __forceinline__ double gauss_row_value(const GaussArgs& a, const GaussComp* gc, int j, int B) {
    const double x = gauss_bin_centre(j, B);
    double sum = 0.0;
    for (int c = 0; c < a.ngauss; ++c) {
        const GaussComp g = gc[c];
        if (g.on) sum = add_rn(sum, mul_rn(g.amp, mul_rn(g.fact, gauss_val(x, g))));
    }
    return add_rn(a.dc, sum);
}

// the unscattered rows at ANY row length B (round 5: general even nbin; the scattering filter then goes through
// the harmonics -- k_any, k_scatter_harm, k_irfft_any of pp_anybin.h)
__global__ __launch_bounds__(256) void k_gauss_rows(GaussArgs a, int B) {
    __shared__ GaussComp gc[PP_MAX_GAUSS];
    const int tid = threadIdx.x;
    for (int n = blockIdx.x; n < a.nchan; n += gridDim.x) {
        gauss_components(a, a.freqs[n], B, gc, tid, 256);
        __syncthreads();
        double* out = a.out + (size_t)n * B;
        for (int j = tid; j < B; j += 256) out[j] = gauss_row_value(a, gc, j, B);
        __syncthreads();
    }
}

template <int M>
__global__ __launch_bounds__(FftPlan<M>::T) void k_gauss_portrait(GaussArgs a) {
    constexpr int T = FftPlan<M>::T, B = 2 * M;
    constexpr int PL = FftPlan<M>::PADLOG;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    __shared__ cplx zin[M];
    __shared__ GaussComp gc[PP_MAX_GAUSS];
    const int tid = threadIdx.x;
    for (int n = blockIdx.x; n < a.nchan; n += gridDim.x) {
        const double nu = a.freqs[n];
        gauss_components(a, nu, B, gc, tid, T);
        __syncthreads();
        for (int j = tid; j < B; j += T) reinterpret_cast<double*>(zin)[j] = gauss_row_value(a, gc, j, B);
        __syncthreads();
        double* out = a.out + (size_t)n * B;
        const double taun = (a.tau_ref != 0.0) ? a.tau_ref * pow(nu / a.nu_ref, a.alpha) : 0.0;
        if (taun == 0.0) {
            for (int j = tid; j < B; j += T) out[j] = reinterpret_cast<const double*>(zin)[j];
        } else {
            // irfft(rfft(row) / (1 + 2 pi i k tau_n)): same packing as k_rotate
            fft_row<M, double>(lds, reinterpret_cast<const double*>(zin), a.twB, tid);
            __syncthreads();
            const cplx z0 = lds[0];
            const double y0 = z0.x + z0.y;
            const double xM = PP_TWO_PI * (double)M * taun;
            const double yM = (z0.x - z0.y) / (1.0 + xM * xM);     // Re(d_M B_M), d_M real
            for (int k = tid; k < M; k += T) {
                cplx yk, ym;
                if (k == 0) { yk = make_double2(y0, 0.0); ym = make_double2(yM, 0.0); }
                else {
                    const double xk = PP_TWO_PI * (double)k * taun, dk = 1.0 / (1.0 + xk * xk);
                    const double xm = PP_TWO_PI * (double)(M - k) * taun, dm = 1.0 / (1.0 + xm * xm);
                    yk = cmul(rfft_harmonic<M>(lds, a.twB, k), make_double2(dk, -xk * dk));
                    ym = cmul(rfft_harmonic<M>(lds, a.twB, M - k), make_double2(dm, -xm * dm));
                }
                ym.y = -ym.y;
                const cplx ev = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y + ym.y));
                cplx od = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y - ym.y));
                cplx w = a.twB[k];
                w.y = -w.y;
                od = cmul(od, w);
                zin[k] = make_double2(ev.x - od.y, -(ev.y + od.x));
            }
            __syncthreads();
            fft_row<M, cplx>(lds, zin, a.twB, tid);
            const double inv = 1.0 / (double)M;
            for (int j = tid; j < M; j += T) {
                const cplx r = lds[lds_pad<PL>(j)];
                reinterpret_cast<double2*>(out)[j] = make_double2(r.x * inv, -r.y * inv);
            }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// Spline (PCA + B-spline) template portraits on the device:
// gen_spline_portrait (pplib.py:932-956) = mean profile + (B-spline curve of the
// PCA coordinates evaluated at the channel frequency) x eigenvectors.  The curve
// is evaluated the way scipy.interpolate.splev does (FITPACK splev / fpbspl,
// ext = 0: the boundary pieces extrapolate): interval search, the k+1 non-zero
// B-splines by the Cox-de Boor recurrence, dot product with the coefficients.
// One workgroup per channel; basis = [mean_prof; eigvec_0; ...; eigvec_{nc-1}]
// rows of nbin (already resampled to nbin on the host when the model's own
// resolution differs: resampling is linear, so it commutes with the sum).
// --------------------------------------------------------------------------
#define PP_MAX_SPLINE_DEG 5
#define PP_MAX_SPLINE_COMP 32
struct SplineArgs {
    const double* freqs;   // [nchan]
    const double* basis;   // [ncomp + 1][nbin]
    const double* t;       // [nk] knots
    const double* c;       // [ncomp][nk] coefficients (FITPACK layout: the last k+1 unused)
    double* out;           // [nchan][nbin]
    int nchan, nbin, ncomp, nk, k;
};

__device__ inline double splev_one(const double* t, int n, const double* c, int k, double x) {
    // FITPACK splev (1-based indices kept as in the source)
    const int k1 = k + 1, k2 = k1 + 1, nk1 = n - k1;
    int l = k1, l1 = l + 1;
    while (x < t[l - 1] && l1 != k2) { l1 = l; l = l - 1; }
    while (x >= t[l1 - 1] && l != nk1) { l = l1; l1 = l + 1; }
    // fpbspl
    double h[PP_MAX_SPLINE_DEG + 2], hh[PP_MAX_SPLINE_DEG + 1];
    h[0] = 1.0;
    for (int j = 1; j <= k; ++j) {
        for (int i = 0; i < j; ++i) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; ++i) {
            const int li = l + i, lj = li - j;
            if (t[li - 1] != t[lj - 1]) {
                const double f = hh[i - 1] / (t[li - 1] - t[lj - 1]);
                h[i - 1] = h[i - 1] + f * (t[li - 1] - x);
                h[i] = f * (x - t[lj - 1]);
            } else h[i] = 0.0;
        }
    }
    double sp = 0.0;
    int ll = l - k1;
    for (int j = 1; j <= k1; ++j) { ll = ll + 1; sp = sp + c[ll - 1] * h[j - 1]; }
    return sp;
}

__global__ __launch_bounds__(256) void k_spline_portrait(SplineArgs a) {
    __shared__ double proj[PP_MAX_SPLINE_COMP];
    const int n = blockIdx.x, tid = threadIdx.x;
    if (tid < a.ncomp) proj[tid] = splev_one(a.t, a.nk, a.c + (size_t)tid * a.nk, a.k, a.freqs[n]);
    __syncthreads();
    double* out = a.out + (size_t)n * a.nbin;
    for (int b = tid; b < a.nbin; b += 256) {
        // numpy.dot(proj, eigvec.T) + mean_prof: the products summed in component order
        double s = 0.0;
        for (int cidx = 0; cidx < a.ncomp; ++cidx)
            s = add_rn(s, mul_rn(proj[cidx], a.basis[(size_t)(cidx + 1) * a.nbin + b]));
        out[b] = add_rn(s, a.basis[b]);
    }
}

// --------------------------------------------------------------------------
// Instrumental response applied to a resident template in the Fourier domain
// (instrumental_response_port_FT, pptoaslib.py:145-179; get_TOAs(add_instrumental_
// response=True), pptoas.py:388-394): m_nk <- m_nk rconst_k sinc(k wid_n), with
// rconst the product of the constant responses (host, nbin/2 + 1 complex values)
// and wid_n the dispersive smearing of channel n in rotations (0 = none).  Re-forms
// |m_nk|^2, its sum and maximum per channel, and the DC term.
// --------------------------------------------------------------------------
// (Mp: pitch of the slot's spectrum rows -- M, or M rounded up to 64 with zeros beyond M for row lengths that are
// no power of two; rconst has M + 1 entries)
__global__ __launch_bounds__(256) void k_model_response(cplx* mft, double* msq, double* msum, double* mmax,
                                                        double* mdc, const cplx* rconst, const double* wid,
                                                        int nchan, int M, int Mp) {
    __shared__ double scratch[8];
    const int n = blockIdx.x, tid = threadIdx.x;
    const double wn = wid ? wid[n] : 0.0;
    double s = 0.0, mx = 0.0;
    for (int k = 1 + tid; k <= M; k += 256) {
        cplx r = rconst ? rconst[k] : make_double2(1.0, 0.0);
        if (wn != 0.0) {
            // numpy.sinc: sin(pi x) / (pi x)
            const double y = 3.141592653589793 * ((double)k * wn);
            const double sc = sin(y) / y;
            r.x *= sc; r.y *= sc;
        }
        const cplx m = cmul(mft[(size_t)n * Mp + (k - 1)], r);
        mft[(size_t)n * Mp + (k - 1)] = m;
        const double p = cnorm(m);
        msq[(size_t)n * Mp + (k - 1)] = p;
        s += p;
        mx = fmax(mx, p);
    }
    s = group_sum<64>(s);
    mx = group_max<64>(mx);
    if ((tid & 63) == 0) { scratch[2 * (tid >> 6)] = s; scratch[2 * (tid >> 6) + 1] = mx; }
    __syncthreads();
    if (tid == 0) {
        s = 0.0; mx = 0.0;
        for (int w = 0; w < 4; ++w) { s += scratch[2 * w]; mx = fmax(mx, scratch[2 * w + 1]); }
        msum[n] = s; mmax[n] = mx;
        if (rconst) mdc[n] *= rconst[0].x;
    }
}

// --------------------------------------------------------------------------
// ppalign's accumulation (ppalign.py:199-206): aligned[n] = sum_i w_in *
// rotate_data(data_in, phase_i, DM_i, P_i, freqs, nu_ref_i), totw[n] = sum_i w_in.
// Rotation is linear, so the weighted harmonics of all subints of a channel are
// summed in the packed spectrum (LDS) and transformed back ONCE per channel: one
// forward FFT per row, one inverse per channel.  One workgroup per channel;
// subints are added in index order (deterministic).
// --------------------------------------------------------------------------
struct AlignArgs {
    const void* src;      // [nsub][nchan][B]
    const double* freqs; long long freqs_stride;
    const double* P;      // [nsub]
    const double* par;    // [nsub][3] phase, DM, nu_ref
    const double* w;      // [nsub][nchan] weights (0 or NaN: row skipped)
    const cplx* twB;
    double* aligned;      // [nchan][B]
    double* totw;         // [nchan]
    int nsub, nchan;
};

template <int M, typename Tio>
__global__ __launch_bounds__(FftPlan<M>::T) void k_align_accum(AlignArgs a) {
    constexpr int T = FftPlan<M>::T;
    constexpr int PL = FftPlan<M>::PADLOG;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    __shared__ cplx zin[M];
    const int tid = threadIdx.x;
    for (int n = blockIdx.x; n < a.nchan; n += gridDim.x) {
        for (int k = tid; k < M; k += T) zin[k] = make_double2(0.0, 0.0);
        double wsum = 0.0;
        for (int i = 0; i < a.nsub; ++i) {
            const double w = a.w[(size_t)i * a.nchan + n];
            if (w == 0.0 || w != w) continue;  // (uniform over the workgroup; negative fitted
                                               // amplitudes weigh negatively, as in the reference)
            const double nu = a.freqs[(size_t)i * a.freqs_stride + n], P = a.P[i];
            const double phase = a.par[i * 3], DM = a.par[i * 3 + 1], nuref = a.par[i * 3 + 2];
            // reference order of operations (pplib.py:2419-2424)
            const double D = PP_DCONST * DM / P;
            const double iref = (nuref == INFINITY) ? 0.0 : 1.0 / (nuref * nuref);
            const double phin = (DM == 0.0) ? phase : phase + D * (1.0 / (nu * nu) - iref);
            fft_row<M, Tio>(lds, reinterpret_cast<const Tio*>(a.src) + ((size_t)i * a.nchan + n) * (2 * M), a.twB,
                            tid);
            const cplx z0 = lds[0];
            const double y0 = z0.x + z0.y;                                    // DC, unchanged
            const double yM = (z0.x - z0.y) * unit_phasor((double)M, phin).x; // Nyquist: real part kept
            for (int k = tid; k < M; k += T) {
                cplx yk, ym;
                if (k == 0) { yk = make_double2(y0, 0.0); ym = make_double2(yM, 0.0); }
                else {
                    yk = cmul(rfft_harmonic<M>(lds, a.twB, k), unit_phasor((double)k, phin));
                    ym = cmul(rfft_harmonic<M>(lds, a.twB, M - k), unit_phasor((double)(M - k), phin));
                }
                ym.y = -ym.y;
                const cplx ev = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y + ym.y));
                cplx od = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y - ym.y));
                cplx tw = a.twB[k];
                tw.y = -tw.y;
                od = cmul(od, tw);
                cplx acc = zin[k];      // this thread's own slot
                acc.x = fma(w, ev.x - od.y, acc.x);
                acc.y = fma(w, -(ev.y + od.x), acc.y);    // conj(ev + i od)
                zin[k] = acc;
            }
            wsum += w;
            __syncthreads();            // the image is overwritten by the next row
        }
        __syncthreads();
        fft_row<M, cplx>(lds, zin, a.twB, tid);
        double* out = a.aligned + (size_t)n * (2 * M);
        const double inv = 1.0 / (double)M;
        for (int j = tid; j < M; j += T) {
            const cplx r = lds[lds_pad<PL>(j)];
            reinterpret_cast<double2*>(out)[j] = make_double2(r.x * inv, -r.y * inv);
        }
        if (tid == 0) a.totw[n] = wsum;
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// Per-channel reduced chi^2 of a fitted subint in the time domain, as
// get_channels_to_zap forms it (pptoas.py:1239-1245 with show_fit :1394-1404 and
// get_red_chi2 pplib.py:727-750): the data rotated by the fitted (phi, DM, GM)
// minus scale_n x (scattered) template, summed over bins, / sigma_n^2 / (nbin-2).
// Evaluated with Parseval on the residual spectrum (DC and Nyquist at weight 1,
// the rest at 2; irfft keeps only the real part at Nyquist), all harmonics.
// --------------------------------------------------------------------------
struct ChanChi2Args {
    const void* src;      // [nsub][nchan][B]
    const cplx* const* mft; const double* const* mdc; const int* slot;   // model slots (slot may be null: 0)
    const double* freqs; long long freqs_stride;
    const double* P;      // [nsub]
    const double* par;    // [nsub][5] phi, DM, GM, tau [rot, linear], alpha -- at nu_refs
    const double* nuref;  // [nsub][3]
    const double* scales; // [nsub][nchan]
    const double* errs;   // [nsub][nchan] time-domain sigma
    const cplx* twB;
    double* out;          // [nsub][nchan]
    int nsub, nchan;
};

template <int M, typename Tio>
__global__ __launch_bounds__(FftPlan<M>::T) void k_chan_chi2(ChanChi2Args a) {
    constexpr int T = FftPlan<M>::T;
    __shared__ cplx lds[FftPlan<M>::LDS_ELEMS];
    __shared__ double scratch[(T / 64) + 1];
    const int tid = threadIdx.x;
    const long long nrows = (long long)a.nsub * a.nchan;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int i = (int)(row / a.nchan), n = (int)(row % a.nchan);
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
        const cplx* mrow = as_global(a.mft[sl]) + (size_t)n * M;
        const double m0 = as_global(a.mdc[sl])[n];
        const double sc = a.scales[(size_t)i * a.nchan + n], sg = a.errs[(size_t)i * a.nchan + n];
        fft_row<M, Tio>(lds, reinterpret_cast<const Tio*>(a.src) + (size_t)row * (2 * M), a.twB, tid);
        const cplx z0 = lds[0];
        double sum = 0.0;
        for (int k = tid; k <= M; k += T) {
            double wgt = 2.0;
            cplx d, m;
            if (k == 0) { d = make_double2(z0.x + z0.y, 0.0); m = make_double2(m0, 0.0); wgt = 1.0; }
            else {
                d = (k == M) ? make_double2(z0.x - z0.y, 0.0) : rfft_harmonic<M>(lds, a.twB, k);
                d = cmul(d, unit_phasor((double)k, phin));
                m = mrow[k - 1];
                if (taun != 0.0) {
                    // B_k = 1 / (1 + 2 pi i k tau_n)
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
            double tot = 0.0;
            for (int wv = 0; wv < T / 64; ++wv) tot += scratch[wv];
            // Parseval: sum_t r_t^2 = (1/B) sum_k wgt_k |R_k|^2
            a.out[row] = tot / (2.0 * M) / (sg * sg) / (double)(2 * M - 2);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// 1-D FFTFIT.  spec[2*i] = rfft(data_i), spec[2*i+1] = rfft(model_i), each
// M+1 complex.  One 256-thread block per pair.
// --------------------------------------------------------------------------
struct FpsArgs {
    const cplx* spec;
    const double* noise;   // [nprof] time-domain sigma, NaN/<0 = measure
    double* out7;          // [nprof][7]
    double lo, hi;
    int Ns, M, nprof;
    int finish;            // 0: Newton polish to rounding; 1: SciPy brute's own finish (Nelder-Mead simplex)
    const cplx* specm;     // nullptr: rows of `spec` alternate data_i, model_i; else spec = data rows, specm = model rows
    int mstride;           // elements between the model rows of `specm` (M + 1; 0 = one model row for all)
};

// sum_k X_k e^{2 pi i k phi} weighted by (1, k, k^2): returns Re-sum, k*Im-sum,
// k^2*Re-sum over this thread's strided harmonics
__device__ __forceinline__ void fps_sums(const cplx* X, int M, double phi, int tid, int nt, double& s0,
                                         double& s1, double& s2) {
    s0 = s1 = s2 = 0.0;
    if (tid >= M) return;
    cplx e = unit_phasor((double)(tid + 1), phi);
    const cplx w = unit_phasor((double)nt, phi);
    for (int j = tid; j < M; j += nt) {
        const cplx z = cmul(X[j], e);
        const double k = (double)(j + 1);
        s0 += z.x;
        s1 = fma(k, z.y, s1);
        s2 = fma(k * k, z.x, s2);
        e = cmul(e, w);
    }
}

// index of the smallest of the values the threads of a 256-thread block hold
// (lowest index on ties, like a sequential scan); valid in every thread
__device__ inline int block_argmin256(double v, int j, double* shv, int* shj) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oj = __shfl_xor(j, o, 64);
        if (ov < v || (ov == v && oj < j)) { v = ov; j = oj; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = v; shj[threadIdx.x >> 6] = j; }
    __syncthreads();
    v = shv[0]; j = shj[0];
    for (int w = 1; w < 4; ++w)
        if (shv[w] < v || (shv[w] == v && shj[w] < j)) { v = shv[w]; j = shj[w]; }
    __syncthreads();
    return j;
}

#define PP_FPS_LDS 2048       // harmonics of a profile pair's cross-spectrum k_fps keeps in LDS (16 B each)
// Block-wide sums of k_fps's 256 threads.  NVW = 1: 256 real threads (block_sum).  NVW = 4: ONE real wave plays the
// four waves in turn -- lane l is threads l, l + 64, l + 128, l + 192 -- and adds the four wave totals in block_sum's
// order: bitwise its results (what a transform wave runs as a ticket, pp_tail.h).  part(vt, v): thread vt's values.
template <int NV, int NVW, typename F>
__device__ __forceinline__ void fps_bsum(double (&out)[NV], double* scratch, int tid, F&& part) {
    if constexpr (NVW == 1) {
        part(tid, out);
        block_sum<NV>(out, scratch);
        __syncthreads();
    } else {
        const int lane = tid & 63;
#pragma unroll 1
        for (int vw = 0; vw < NVW; ++vw) {
            double v[NV];
            part(lane + 64 * vw, v);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] = group_sum<64>(v[q]);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < NV; ++q) scratch[vw * NV + q] = v[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double t = 0.0;
            for (int w = 0; w < NVW; ++w) t += scratch[w * NV + q];
            out[q] = t;
        }
        __syncthreads();
    }
}

// The 1-D fit of profile pair i: dspec(k) = the data profile's harmonic k, m[k] the model's, X[0 .. M) the work array
// of their cross-spectrum (LDS where it fits).  Threads: 256 (NVW = 1) or one wave standing in for them (NVW = 4).
template <int NVW, typename DF>
__device__ __forceinline__ void fps_body(const FpsArgs& a, const int i, const int tid, const int M, DF&& dspec,
                                         const cplx* m, cplx* X, double* scratch, double* shv, int* shj) {
    constexpr int RT = 256 / NVW;        // real threads
    const int H = M + 1, kc = (int)(0.75 * H);
    double v[3];   // sum |d|^2, sum |m|^2, tail of |d|^2
    fps_bsum<3, NVW>(v, scratch, tid, [&](const int vt, double (&u)[3]) __attribute__((always_inline)) {
        u[0] = u[1] = u[2] = 0.0;
        for (int k = 1 + vt; k <= M; k += 256) {
            const cplx dk = dspec(k), mk = m[k];
            X[k - 1] = cmulc(dk, mk);
            const double pd = cnorm(dk);
            u[0] += pd; u[1] += cnorm(mk);
            if (k >= kc) u[2] += pd;
        }
    });
    const double B = 2.0 * M;
    double sig = a.noise ? a.noise[i] : NAN;
    if (!(sig >= 0.0)) sig = sqrt(v[2] / B / (double)(H - kc));   // get_noise_PS
    const double err2 = sig * sig * (0.5 * B);
    const double dd = v[0] / err2, pp_ = v[1] / err2;
    // brute grid, both ends included (scipy.optimize.brute with complex(Ns))
    const int Ns = a.Ns;
    double bestv = INFINITY;
    int bestj = 0x7fffffff;
    // (grid point j = j * step + lo, the arithmetic of numpy's mgrid)
    const double h = (Ns > 1) ? (a.hi - a.lo) / (double)(Ns - 1) : 0.5;
    for (int j = tid; j < Ns; j += RT) {
        const double phi = (Ns > 1) ? add_rn(mul_rn((double)j, h), a.lo) : a.lo;
        double s0, s1, s2;
        fps_sums(X, M, phi, 0, 1, s0, s1, s2);
        const double vv = -s0 / err2;
        if (vv < bestv) { bestv = vv; bestj = j; }
    }
    int best;
    if constexpr (NVW == 1) best = min(block_argmin256(bestv, bestj, shv, shj), Ns - 1);
    else {
        // (the same total order -- value, then index -- over the one wave that holds every grid point)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bestv, o, 64);
            const int oj = __shfl_xor(bestj, o, 64);
            if (ov < bestv || (ov == bestv && oj < bestj)) { bestv = ov; bestj = oj; }
        }
        best = min(bestj, Ns - 1);
    }
    double phi = (Ns > 1) ? add_rn(mul_rn((double)best, h), a.lo) : a.lo;
    double f = 0.0, f2 = 0.0;
    // the three sums at one phase, the same in every thread
    auto sums_at = [&](double x, double (&s)[3]) __attribute__((always_inline)) {
        fps_bsum<3, NVW>(s, scratch, tid, [&](const int vt, double (&u)[3]) __attribute__((always_inline)) {
            fps_sums(X, M, x, vt, 256, u[0], u[1], u[2]);
        });
    };
    auto feval = [&](double x) -> double {
        double s[3];
        sums_at(x, s);
        return -s[0] / err2;
    };
    if (a.finish == 1) {
        // What scipy.optimize.brute does after its grid (the reference calls it with
        // the default finish = fmin, pplib.py:2085): Nelder-Mead from the best grid
        // point, xtol = ftol = 1e-4, at most 200 iterations / evaluations --
        // operation by operation for one dimension (scipy/optimize/_optimize.py
        // _minimize_neldermead: rho 1, chi 2, psi 1/2, sigma 1/2; second vertex
        // 1.05 x0, or 0.00025 when x0 = 0).  The reference's phase IS this simplex's
        // best vertex, ~1e-5 rot from the maximum of the correlation.
        double sim0 = phi, sim1 = (phi != 0.0) ? mul_rn(1.05, phi) : 0.00025;
        double f0 = feval(sim0), f1 = feval(sim1);
        int fcalls = 2, iterations = 1;
        auto order = [&]() {           // argsort of two values, stable
            if (f1 < f0) { const double t = sim0; sim0 = sim1; sim1 = t; const double u = f0; f0 = f1; f1 = u; }
        };
        order();
        while (fcalls < 200 && iterations < 200) {
            if (fabs(sim1 - sim0) <= 1e-4 && fabs(f0 - f1) <= 1e-4) break;
            const double xbar = sim0;
            const double xr = sub_rn(mul_rn(2.0, xbar), sim1);
            const double fxr = feval(xr); ++fcalls;
            bool shrink = false;
            if (fxr < f0) {
                const double xe = sub_rn(mul_rn(3.0, xbar), mul_rn(2.0, sim1));
                const double fxe = feval(xe); ++fcalls;
                if (fxe < fxr) { sim1 = xe; f1 = fxe; } else { sim1 = xr; f1 = fxr; }
            } else if (fxr < f1) {      // (f0 <= fxr: outside contraction)
                const double xc = sub_rn(mul_rn(1.5, xbar), mul_rn(0.5, sim1));
                const double fxc = feval(xc); ++fcalls;
                if (fxc <= fxr) { sim1 = xc; f1 = fxc; } else shrink = true;
            } else {                    // inside contraction
                const double xcc = add_rn(mul_rn(0.5, xbar), mul_rn(0.5, sim1));
                const double fxcc = feval(xcc); ++fcalls;
                if (fxcc < f1) { sim1 = xcc; f1 = fxcc; } else shrink = true;
            }
            if (shrink) {
                sim1 = add_rn(sim0, mul_rn(0.5, sub_rn(sim1, sim0)));
                f1 = feval(sim1); ++fcalls;
            }
            ++iterations;
            order();
        }
        phi = sim0;
    } else {
    // polish: safeguarded Newton on f(phi) = -Re sum X e / err2 inside +-1 grid step
    double lo = phi - h, hi = phi + h;
    for (int it = 0; it < 60; ++it) {
        double s[3];
        sums_at(phi, s);
        f = -s[0] / err2;
        const double f1 = PP_TWO_PI * s[1] / err2;                   // df/dphi
        f2 = PP_TWO_PI * PP_TWO_PI * s[2] / err2;                     // d2f/dphi2
        if (f1 > 0.0) hi = phi; else lo = phi;
        double nxt = (f2 > 0.0) ? phi - f1 / f2 : 0.5 * (lo + hi);
        if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
        const double step = fabs(nxt - phi);
        phi = nxt;
        if (step < 1e-15) break;
    }
    }
    // value and curvature at the final phase
    double s[3];
    sums_at(phi, s);
    f = -s[0] / err2;
    f2 = PP_TWO_PI * PP_TWO_PI * s[2] / err2;
    if (tid == 0) {
        const double scale = -f / pp_;
        double* o = a.out7 + (size_t)i * 7;
        o[0] = phi;
        o[1] = 1.0 / sqrt(scale * f2);
        o[2] = scale;
        o[3] = 1.0 / sqrt(pp_);
        o[4] = sqrt(scale * scale * pp_);
        o[5] = (dd - f * f / pp_) / (B - 2.0);
        o[6] = 0.0;
    }
}

__global__ __launch_bounds__(256) void k_fps(FpsArgs a, cplx* xwork) {
    const int i = blockIdx.x, tid = threadIdx.x, M = a.M;
    __shared__ double scratch[4 * 4];
    __shared__ double shv[4];
    __shared__ int shj[4];
    const cplx* d = a.specm ? a.spec + (size_t)i * (M + 1) : a.spec + (size_t)(2 * i) * (M + 1);
    const cplx* m = a.specm ? a.specm + (size_t)i * a.mstride : a.spec + (size_t)(2 * i + 1) * (M + 1);
    // the cross-spectrum every grid point and every simplex vertex walks: in LDS when it fits (each of the Ns grid
    // threads reads all M harmonics one after the other -- from global memory that is M dependent round trips,
    // ~0.5 ms of the kernel's 0.7 at M = 1024), else in the work buffer
    __shared__ cplx Xs[PP_FPS_LDS];
    auto dv = [&](const int k) __attribute__((always_inline)) { return d[k]; };
    // (inlined into both call sites: LDS and global addressing)
    if (M <= PP_FPS_LDS) fps_body<1>(a, i, tid, M, dv, m, Xs, scratch, shv, shj);
    else fps_body<1>(a, i, tid, M, dv, m, xwork + (size_t)i * M, scratch, shv, shj);
}

// DM axis of the coarse grid: per subint the trial with the highest correlation peak,
// refined by the parabola through it and its neighbours; xout[i] = xbase[i] with that DM
__global__ void k_seed_dm_pick(const int* act, int nact, const double* pk, int ntrial, double step,
                               const double* xbase, double* xout) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nact) return;
    const int i = sub_of(act, j);
    const double* p = pk + (size_t)i * ntrial;
    int b = 0;
    for (int t = 1; t < ntrial; ++t) if (p[t] > p[b]) b = t;
    double frac = 0.0;
    if (b > 0 && b < ntrial - 1) {
        const double den = p[b - 1] - 2.0 * p[b] + p[b + 1];
        if (den < 0.0) frac = fmin(0.5, fmax(-0.5, 0.5 * (p[b - 1] - p[b + 1]) / den));
    }
    for (int q = 0; q < 5; ++q) xout[i * 5 + q] = xbase[i * 5 + q];
    xout[i * 5 + 1] = xbase[i * 5 + 1] + ((double)(b - (ntrial - 1) / 2) + frac) * step;
}

// --------------------------------------------------------------------------
// The subints that still need work, listed in index order (one 256-thread block):
// seedq != nullptr selects those whose seed quality is below qmin, else those whose
// solver state is not done.  act[0..*count) receives the indices.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_list_active(const SubState* st, const double* seedq, double qmin, int nsub,
                                                     int* act, int* count) {
    __shared__ int offs[257];
    const int tid = threadIdx.x;
    const int per = (nsub + 255) / 256, lo = min(nsub, tid * per), hi = min(nsub, lo + per);
    auto want = [&](int i) { return seedq ? !(seedq[i] >= qmin) : (st[i].done == 0); };
    int c = 0;
    for (int i = lo; i < hi; ++i) c += want(i) ? 1 : 0;
    offs[tid + 1] = c;
    if (tid == 0) offs[0] = 0;
    __syncthreads();
    if (tid == 0) for (int t = 1; t <= 256; ++t) offs[t] += offs[t - 1];
    __syncthreads();
    int o = offs[tid];
    for (int i = lo; i < hi; ++i) if (want(i)) act[o++] = i;
    if (tid == 0) *count = offs[256];
}

// --------------------------------------------------------------------------
// Phase seed on the device (role of pptoas.py:421-457).  With the guessed
// DM/GM the per-channel cross-correlations are aligned and summed:
//   Y_k = sum_n w_n X_nk e^{2 pi i k (phi_n - phi)},   CCF(phi) = Re sum_k Y_k B_k^* e^{2 pi i k phi}
// (B_k: scattering kernel of the guessed tau at nu_fit; w_n = 1/sigma_n^2).
// k_seed_accum: grid (nchunk, nsub); thread t owns harmonics t+1, t+257, ...
// k_seed_fit:   per subint, reduce the chunks, grid + Newton polish, write the
//               phase into x0[i][0].
// --------------------------------------------------------------------------
#define PP_SEED_KPT 16   // harmonics per lane (one wave per channel): Kt <= 1024
// dm_off: trial offset added to the guessed DM (the DM axis of the coarse (phi, DM)
// grid: one accumulation + phase search per trial value, the best correlation peak wins)
__global__ __launch_bounds__(256) void k_seed_accum(FitArgs a, cplx* ypart, int Ks, const double* xbase,
                                                    double dm_off) {
    const int jx = blockIdx.x, i = sub_of(a.act, jx), chunk = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ cplx ysh[3 * 64 * PP_SEED_KPT / 4];   // three waves' partial spectra, one quarter at a time
    const double P = a.P[i];
    const double nuDM = a.nu_fit[i * 3], nuGM = a.nu_fit[i * 3 + 1];
    const double DM = xbase[i * 5 + 1] + dm_off, GM = xbase[i * 5 + 2];
    const double* freqs = a.freqs + (size_t)i * a.freqs_stride;
    const double* wts = a.wts + (size_t)i * a.nchan;
    const int* ktv = a.ktab ? a.ktab[a.slot ? a.slot[i] : 0] : nullptr;
    // lane owns harmonics lane+1 + 64 j; each wave walks its own channels with the
    // phasor advanced by e^{2 pi i 64 phi_n} (one sincos per lane per channel)
    cplx y[PP_SEED_KPT];
#pragma unroll
    for (int j = 0; j < PP_SEED_KPT; ++j) y[j] = make_double2(0.0, 0.0);
    const int n0 = chunk * a.cpc, n1 = min(n0 + a.cpc, a.nchan_x);
    for (int nn = n0 + wave; nn < n1; nn += 4) {
        const int n = a.coff + nn * a.cstep;
        const double w = wts[n];
        if (w == 0.0) continue;
        double p1, p2;
        phase_geom(freqs[n], P, nuDM, nuGM, p1, p2);
        const double phin = DM * p1 + GM * p2;
        cplx e = unit_phasor((double)(lane + 1), phin);
        const cplx wst = make_double2(__shfl(e.x, 63, 64), __shfl(e.y, 63, 64));
        const cplx* xrow = a.X + ((size_t)jx * a.nchan_x + nn) * a.Xs;
        const int ktn = min(ktv ? ktv[n] : a.Kt, Ks);
        // all of the row's loads first (independent, 1 KB per wave-instruction)
        cplx xv[PP_SEED_KPT];
#pragma unroll
        for (int j = 0; j < PP_SEED_KPT; ++j) {
            const int k = lane + 1 + 64 * j;
            xv[j] = (k <= ktn) ? xrow[k - 1] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < PP_SEED_KPT; ++j) {
            if (64 * j < ktn) {
                const cplx z = cmul(xv[j], e);
                y[j].x = fma(w, z.x, y[j].x);
                y[j].y = fma(w, z.y, y[j].y);
                e = cmul(e, wst);
            }
        }
    }
    // sum the four waves' spectra (a quarter of the harmonics per round) and store
    cplx* yo = ypart + ((size_t)jx * a.nchunk + chunk) * Ks;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < PP_SEED_KPT / 4; ++j)
                ysh[((wave - 1) * (PP_SEED_KPT / 4) + j) * 64 + lane] = y[q * (PP_SEED_KPT / 4) + j];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < PP_SEED_KPT / 4; ++j) {
                const int jj = q * (PP_SEED_KPT / 4) + j;
                cplx s = y[jj];
                for (int ww = 0; ww < 3; ++ww) {
                    const cplx v = ysh[(ww * (PP_SEED_KPT / 4) + j) * 64 + lane];
                    s.x += v.x; s.y += v.y;
                }
                const int k = lane + 1 + 64 * jj;
                if (k <= Ks) yo[k - 1] = s;
            }
        }
    }
}

// seedq (optional): significance of the grid's maximum, (max - mean over the grid) /
// (noise rms of the correlation, from the weights and the template power) -- the
// matched-filter S/N of the channels used, which tells a seed formed from a subset
// of the channels apart from a noise peak
// pkout (optional, DM trials): only the height of the correlation peak is recorded,
// pkout[i * ntrial + trial]; nothing is written to x0 (k_seed_dm_pick then chooses the
// DM and a last accumulation + fit at it writes the seed).  xbase / dm_off as in
// k_seed_accum.
__global__ __launch_bounds__(256) void k_seed_fit(FitArgs a, const cplx* ypart, cplx* ywork, double* x0, int Ns,
                                                  int Ks, double* seedq, const double* xbase, double dm_off,
                                                  double* pkout, int trial, int ntrial) {
    const int jx = blockIdx.x, i = sub_of(a.act, jx), tid = threadIdx.x, K = Ks;
    __shared__ double scratch[4 * 4];
    __shared__ double shv[4];
    __shared__ int shj[4];
    // (the correlation's spectrum in LDS when it fits, as in k_fps: every grid thread walks all of it)
    __shared__ cplx Ys[PP_FPS_LDS];
    cplx* Y = (K <= PP_FPS_LDS) ? Ys : ywork + (size_t)jx * K;
    // scattering kernel of the guessed tau at the fit reference frequency
    const double taup = xbase[i * 5 + 3];
    const double tau = a.scat ? (a.log10_tau ? pow(10.0, taup) : taup) : 0.0;
    for (int k = tid + 1; k <= K; k += 256) {
        cplx s = make_double2(0.0, 0.0);
        for (int c = 0; c < a.nchunk; ++c) {
            const cplx v = ypart[((size_t)jx * a.nchunk + c) * K + k - 1];
            s.x += v.x; s.y += v.y;
        }
        if (tau != 0.0) {   // times conj(B_k) = (1 + i u)/(1 + u^2)
            const double u = PP_TWO_PI * k * tau, D = 1.0 / fma(u, u, 1.0);
            s = cmul(s, make_double2(D, u * D));
        }
        Y[k - 1] = s;
    }
    __syncthreads();
    Ns = max(Ns, 2);
    double bestv = INFINITY;
    int bestj = 0x7fffffff;
    double gs[1] = {0.0};                // sum of the CCF over this thread's points
    for (int j = tid; j < Ns; j += 256) {
        const double phi = -0.5 + (double)j / (double)(Ns - 1);
        double s0, s1, s2;
        fps_sums(Y, K, phi, 0, 1, s0, s1, s2);
        if (-s0 < bestv) { bestv = -s0; bestj = j; }
        gs[0] += s0;
    }
    const int best = min(block_argmin256(bestv, bestj, shv, shj), Ns - 1);
    // polish: safeguarded Newton on f(phi) = -CCF inside +-1 grid step of the best point
    const double h = 1.0 / (double)(Ns - 1);
    double phi = -0.5 + (double)best / (double)(Ns - 1), lo = phi - h, hi = phi + h;
    for (int it = 0; it < 60; ++it) {
        double s[3];
        fps_sums(Y, K, phi, tid, 256, s[0], s[1], s[2]);
        block_sum<3>(s, scratch);
        __syncthreads();
        const double f1 = s[1], f2 = s[2];     // signs of df/dphi, d2f/dphi2 of f = -sum
        if (f1 > 0.0) hi = phi; else lo = phi;
        double nxt = (f2 > 0.0) ? phi - f1 / (PP_TWO_PI * f2) : 0.5 * (lo + hi);
        if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
        const double step = fabs(nxt - phi);
        phi = nxt;
        if (step < 1e-13) break;
    }
    // height of the correlation at the polished maximum (the grid alone samples a
    // narrow peak too coarsely to compare DM trials or to quote an S/N)
    double pk = 0.0;
    if (pkout || seedq) {
        double s[3];
        fps_sums(Y, K, phi, tid, 256, s[0], s[1], s[2]);
        block_sum<3>(s, scratch);
        __syncthreads();
        pk = s[0];
    }
    if (pkout) {
        if (tid == 0) pkout[(size_t)i * ntrial + trial] = pk;
        return;
    }
    if (seedq) {
        // noise of the correlation: Var = sum_n w_n^2 sigma_Fn^2 sum_k |m_nk|^2 = sum_n w_n S_n
        // over the channels that went into Y (S_n over all harmonics: a slight
        // overestimate, on the safe side)
        const double* wts = a.wts + (size_t)i * a.nchan;
        const double* msum = as_global(a.msum[a.slot ? a.slot[i] : 0]);
        double nz = 0.0;
        for (int nn = tid; nn < a.nchan_x; nn += 256) {
            const int n = a.coff + nn * a.cstep;
            const double w = wts[n];
            if (w != 0.0) nz += w * msum[n];
        }
        double g2[2] = {gs[0], nz};
        block_sum<2>(g2, scratch);
        __syncthreads();
        const double mean = g2[0] / (double)Ns;
        if (tid == 0) seedq[i] = (g2[1] > 0.0) ? (pk - mean) / sqrt(g2[1]) : 0.0;
    }
    if (tid == 0) {
        // wrap to [-0.5, 0.5) like phase_transform(..., mod=True)
        if (fabs(phi) >= 0.5) { phi = fmod(phi, 1.0); if (phi < 0.0) phi += 1.0; }
        if (phi >= 0.5) phi -= 1.0;
        x0[i * 5] = phi;
        x0[i * 5 + 1] = xbase[i * 5 + 1] + dm_off;
    }
}

}  // namespace pp
