// Batched real-to-complex FFT along the phase axis, one row per workgroup,
// f64 arithmetic, Stockham autosort through LDS with register radix-4/8/16
// butterflies.  A real row of B samples is transformed as a complex FFT of
// M = B/2 points followed by the even/odd split.
//
// Work layout (T threads per row, T = 64 for M <= 1024 so a row is one
// wavefront): stage s reads element t + k*(M/R) (lane-contiguous, 16 B per
// lane: coalesced from HBM in the first stage, conflict-free from LDS later),
// does M/R radix-R butterflies in registers, and writes element
// (t%S) + S*R*(t/S) + S*j in place.  The LDS image is padded by one element
// every R1 (first radix) so the stride-R1 writes of the first stage spread over
// the 32 write banks.
#pragma once
#include "pp_common.h"

namespace pp {

__device__ __forceinline__ void dft2(cplx& a, cplx& b) {
    cplx t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// forward 4-point DFT, natural order in/out
__device__ __forceinline__ void dft4(cplx& a0, cplx& a1, cplx& a2, cplx& a3) {
    cplx t0 = cadd(a0, a2), t1 = csub(a0, a2);
    cplx t2 = cadd(a1, a3), t3 = cmul_mi(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

template <int R>
__device__ __forceinline__ void dft_reg(cplx (&v)[R]);

template <>
__device__ __forceinline__ void dft_reg<2>(cplx (&v)[2]) { dft2(v[0], v[1]); }

template <>
__device__ __forceinline__ void dft_reg<4>(cplx (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }

template <>
__device__ __forceinline__ void dft_reg<8>(cplx (&v)[8]) {
    const double h = 0.70710678118654752440;
    // even / odd 4-point transforms
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    // odd outputs times W8^k
    cplx o0 = v[1];
    cplx o1 = make_double2(h * (v[3].x + v[3].y), h * (v[3].y - v[3].x));   // * (1-i)/sqrt2
    cplx o2 = cmul_mi(v[5]);
    cplx o3 = make_double2(h * (v[7].y - v[7].x), -h * (v[7].x + v[7].y));  // * (-1-i)/sqrt2
    cplx e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

template <>
__device__ __forceinline__ void dft_reg<16>(cplx (&v)[16]) {
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178;
    const double h = 0.70710678118654752440;
    // F_q[j2] = DFT4 over m of v[4m + q]   (stored back in v[4*j2 + q])
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4(v[q], v[4 + q], v[8 + q], v[12 + q]);
    // twiddle v[4*j2 + q] *= W16^(q*j2)
    const cplx w1 = make_double2(c1, -s1), w2 = make_double2(h, -h), w3 = make_double2(s1, -c1);
    const cplx w6 = make_double2(-h, -h), w9 = make_double2(-c1, s1);
    v[5] = cmul(v[5], w1);   // j2=1,q=1
    v[6] = cmul(v[6], w2);   // j2=1,q=2
    v[7] = cmul(v[7], w3);   // j2=1,q=3
    v[9] = cmul(v[9], w2);   // j2=2,q=1
    v[10] = cmul_mi(v[10]);  // j2=2,q=2 : W16^4 = -i
    v[11] = cmul(v[11], w6); // j2=2,q=3
    v[13] = cmul(v[13], w3); // j2=3,q=1
    v[14] = cmul(v[14], w6); // j2=3,q=2
    v[15] = cmul(v[15], w9); // j2=3,q=3
    // b[j2 + 4*j1] = DFT4 over q of v[4*j2 + q]
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) dft4(v[4 * j2], v[4 * j2 + 1], v[4 * j2 + 2], v[4 * j2 + 3]);
    // now v[4*j2 + j1] holds b[j2 + 4*j1]: transpose the 4x4 to natural order
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            cplx t = v[4 * a + b];
            v[4 * a + b] = v[4 * b + a];
            v[4 * b + a] = t;
        }
}

template <int PADLOG>
__device__ __forceinline__ int lds_pad(int i) { return i + (i >> PADLOG); }

__device__ __forceinline__ cplx load_pair(const double* row, int idx) {
    return *reinterpret_cast<const double2*>(row + 2 * idx);
}
// first-stage input already complex (e.g. staged in LDS by the generator)
__device__ __forceinline__ cplx load_pair(const cplx* row, int idx) { return row[idx]; }
__device__ __forceinline__ cplx load_pair(const float* row, int idx) {
    float2 v = *reinterpret_cast<const float2*>(row + 2 * idx);
    return make_double2((double)v.x, (double)v.y);
}

// One Stockham stage.  S = product of the radices already applied.
// twB[k] = exp(-2 pi i k / B), k = 0..M  (W_M^j = twB[2j]).
template <int M, int T, int R, int S, int PADLOG, bool FIRST, typename Tin>
__device__ __forceinline__ void fft_stage(cplx* lds, const Tin* __restrict__ grow,
                                          const cplx* __restrict__ twB, int tid) {
    constexpr int NBF = M / R;
    constexpr int PER = (NBF + T - 1) / T;
    constexpr bool LAST = (S * R == M);
    cplx v[PER][R];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = tid + T * i;
        if (NBF % T == 0 || t < NBF) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int idx = t + k * NBF;
                if (FIRST) v[i][k] = load_pair(grow, idx);
                else v[i][k] = lds[lds_pad<PADLOG>(idx)];
            }
        }
    }
    if (!FIRST) __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = tid + T * i;
        if (NBF % T == 0 || t < NBF) {
            dft_reg<R>(v[i]);
            const int q = t & (S - 1);
            const int ob = q + S * R * (t / S);
            if (!LAST) {
                // w^j, w = W_M^(t - q), by a product tree (depth <= 4)
                const cplx w1 = twB[2 * (t - q)];
                cplx w[R];
                w[1] = w1;
#pragma unroll
                for (int j = 2; j < R; ++j) w[j] = cmul(w[j >> 1], w[j - (j >> 1)]);
#pragma unroll
                for (int j = 1; j < R; ++j) v[i][j] = cmul(v[i][j], w[j]);
            }
#pragma unroll
            for (int j = 0; j < R; ++j) lds[lds_pad<PADLOG>(ob + S * j)] = v[i][j];
        }
    }
    __syncthreads();
}

// Complex FFT of the M = B/2 packed pairs of one real row; result Z[0..M-1] is
// left in `lds` (padded with lds_pad<FftPlan<M>::PADLOG>).
template <int M>
struct FftPlan;
#define PP_PLAN(M_, T_, PL_) \
    template <> struct FftPlan<M_> { static constexpr int T = T_; static constexpr int PADLOG = PL_; \
        static constexpr int LDS_ELEMS = M_ + (M_ >> PL_) + 1; }
PP_PLAN(16, 64, 2);
PP_PLAN(32, 64, 3);
PP_PLAN(64, 64, 3);
PP_PLAN(128, 64, 3);
PP_PLAN(256, 64, 4);
PP_PLAN(512, 64, 3);
PP_PLAN(1024, 64, 4);
PP_PLAN(2048, 128, 4);
PP_PLAN(4096, 256, 4);
#undef PP_PLAN

template <int M, typename Tin>
__device__ __forceinline__ void fft_row(cplx* lds, const Tin* __restrict__ grow,
                                        const cplx* __restrict__ twB, int tid) {
    constexpr int T = FftPlan<M>::T;
    constexpr int P = FftPlan<M>::PADLOG;
    if constexpr (M == 16) {
        fft_stage<M, T, 4, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 4, 4, P, false>(lds, grow, twB, tid);
    } else if constexpr (M == 32) {
        fft_stage<M, T, 8, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 4, 8, P, false>(lds, grow, twB, tid);
    } else if constexpr (M == 64) {
        fft_stage<M, T, 8, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 8, 8, P, false>(lds, grow, twB, tid);
    } else if constexpr (M == 128) {
        fft_stage<M, T, 8, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 4, 8, P, false>(lds, grow, twB, tid);
        fft_stage<M, T, 4, 32, P, false>(lds, grow, twB, tid);
    } else if constexpr (M == 256) {
        fft_stage<M, T, 16, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 16, 16, P, false>(lds, grow, twB, tid);
    } else if constexpr (M == 512) {
        fft_stage<M, T, 8, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 8, 8, P, false>(lds, grow, twB, tid);
        fft_stage<M, T, 8, 64, P, false>(lds, grow, twB, tid);
    } else if constexpr (M == 1024) {
        fft_stage<M, T, 16, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 8, 16, P, false>(lds, grow, twB, tid);
        fft_stage<M, T, 8, 128, P, false>(lds, grow, twB, tid);
    } else if constexpr (M == 2048) {
        fft_stage<M, T, 16, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 16, 16, P, false>(lds, grow, twB, tid);
        fft_stage<M, T, 8, 256, P, false>(lds, grow, twB, tid);
    } else {
        static_assert(M == 4096, "unsupported FFT size");
        fft_stage<M, T, 16, 1, P, true>(lds, grow, twB, tid);
        fft_stage<M, T, 16, 16, P, false>(lds, grow, twB, tid);
        fft_stage<M, T, 16, 256, P, false>(lds, grow, twB, tid);
    }
}

// harmonic k (1..M) of the real transform from the packed complex transform
// left in LDS by fft_row:  d_k = E - i W_B^k O,  E,O = (Z_k +- conj Z_{M-k})/2
template <int M>
__device__ __forceinline__ cplx rfft_harmonic(const cplx* lds, const cplx* __restrict__ twB, int k) {
    constexpr int P = FftPlan<M>::PADLOG;
    const cplx zk = lds[lds_pad<P>(k & (M - 1))];
    cplx zc = lds[lds_pad<P>((M - k) & (M - 1))];
    zc.y = -zc.y;
    const cplx E = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
    const cplx O = make_double2(0.5 * (zk.x - zc.x), 0.5 * (zk.y - zc.y));
    const cplx w = twB[k];
    const cplx wo = cmul(w, O);
    // E - i*wo
    return make_double2(E.x + wo.y, E.y - wo.x);
}

}  // namespace pp
