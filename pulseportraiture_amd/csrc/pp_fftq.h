// 1024-point complex FFT of one row in ONE wave with ONE exchange through LDS.
//
// k_xspec's 16.8.8 Stockham plan moves the row through LDS three times (and a fourth
// time for the even/odd split).  The LDS is one unit per CU shared by its eight
// resident rows, and the counters put it at ~94 % busy (SQ_ACTIVE_INST_LDS: 11.7 % of
// every wave's lifetime x 8 waves): the kernel runs at the rate of its LDS
// instructions, not of its VALU instructions or its HBM bytes.  This plan is 16.4.16
// with the first exchange done in registers:
//
//   n = l + 64 r  (lane l, register r)          k = ka + 16 kc + 64 kd
//   stage 1   DFT16 over r -> ka in registers;  twiddle W_1024^(l ka)
//   swap      lane bits 5,4 <-> register bits 3,2  (v_permlane32_swap / v_permlane16_swap,
//             gfx950: one VALU instruction per pair of dwords, no LDS)
//             registers now hold (l5 l4 | a1 a0), lanes (a3 a2 | l3..l0)
//   stage 2   DFT4 over (l5 l4) -> kc;  twiddle W_64^((l & 15) kc)
//   exchange  16 x 16 transpose inside every row of 16 lanes, through LDS:
//             registers <- (l3..l0), lanes <- (a3 a2 | kc1 kc0 a1 a0)
//   stage 3   DFT16 over (l3..l0) -> kd in registers, no twiddle
//
// Lane t ends with Z[lam(t) + 64 kd] in register kd, where
//   lam(t) = 4 (t >> 4) + (t & 3) + 16 ((t >> 2) & 3)
// (the low six bits of k, bit pairs (5,4) and (3,2) of the lane number exchanged).
#pragma once
#include "pp_fft.h"

namespace pp {

// lanes 32..63 of a <-> lanes 0..31 of b
__device__ __forceinline__ void lane_swap32(double& a, double& b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
// lanes 16..31 / 48..63 of a <-> lanes 0..15 / 32..47 of b
__device__ __forceinline__ void lane_swap16(double& a, double& b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}

// low six bits of the harmonic numbers lane t ends with
__device__ __forceinline__ int fftq_lambda(int t) { return 4 * (t >> 4) + (t & 3) + 16 * ((t >> 2) & 3); }
// the lane that ends with lambda
__device__ __forceinline__ int fftq_lane_of(int lam) { return 16 * ((lam >> 2) & 3) + (lam & 3) + 4 * ((lam >> 4) & 3); }

constexpr int FFTQ_LDS_ELEMS = 4 * 272;      // four rows of 16 lanes x (16 x 17) elements

// t1 = W_1024^tid, t2 = W_64^(tid & 15).  mid() is called once, after the 16 inputs are
// dead: WHEN = 0 a quarter into stage 1, 1 after the stage-1 twiddles (their powers no
// longer live), 2 after the lane swaps, 3 after the transpose (before the last stage).  power (optional) += this lane's share of
// sum_{k=1}^{M-1} |Z_k|^2 + (Re Z_0 - Im Z_0)^2.
template <int WHEN = 0, typename Mid>
__device__ __forceinline__ void fftq1024(cplx (&v)[16], cplx* lds, const cplx t1, const cplx t2, int tid,
                                         double* power, Mid mid) {
    // ---- stage 1 ----
    dft16_first(v);
    if (WHEN == 0) mid();
    dft16_second(v);
    {
        // v[j] *= t1^j, every power formed once (product tree, as stage_finish<TREE>)
        cplx wq[16];
        wq[1] = t1;
#pragma unroll
        for (int j = 2; j < 16; ++j) {
            if (j % 2 == 0) {
                const cplx h = wq[j / 2];
                wq[j] = make_double2(fma(h.x, h.x, -h.y * h.y), 2.0 * h.x * h.y);
            } else wq[j] = cmul(wq[j - 1], wq[1]);
        }
#pragma unroll
        for (int j = 1; j < 16; ++j) v[j] = cmul(v[j], wq[j]);
    }
    if (WHEN == 1) mid();
    // ---- lane bits 5,4 <-> register bits 3,2 ----
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lane_swap32(v[j].x, v[j + 8].x);
        lane_swap32(v[j].y, v[j + 8].y);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j & 4) continue;
        lane_swap16(v[j].x, v[j + 4].x);
        lane_swap16(v[j].y, v[j + 4].y);
    }
    if (WHEN == 2) mid();
    // ---- stage 2: DFT4 over register bits 3,2; twiddle t2^kc ----
    dft16_first(v);
    {
        const cplx w2 = make_double2(fma(t2.x, t2.x, -t2.y * t2.y), 2.0 * t2.x * t2.y);
        const cplx w3 = cmul(w2, t2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v[4 + c] = cmul(v[4 + c], t2);
            v[8 + c] = cmul(v[8 + c], w2);
            v[12 + c] = cmul(v[12 + c], w3);
        }
    }
    // ---- 16 x 16 transpose inside every row of 16 lanes ----
    {
        cplx* wbase = lds + (tid >> 4) * 272 + 17 * (tid & 15);
#pragma unroll
        for (int j = 0; j < 16; ++j) wbase[j] = v[j];
        lds_sync<64>();
        const cplx* rbase = lds + (tid >> 4) * 272 + (tid & 15);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = rbase[17 * j];
        lds_sync<64>();
    }
    if (WHEN == 3) mid();
    // ---- stage 3 ----
    dft_reg<16>(v);
    if (power) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j == 0 && tid == 0) { const double dM = v[0].x - v[0].y; acc += dM * dM; }
            else acc += cnorm(v[j]);
        }
        *power += acc;
    }
}

// --------------------------------------------------------------------------
// 512-point complex FFT (1024-bin rows) of one row in ONE wave with ONE exchange through LDS:
// plan 8.4.2.8, eight values per lane.
//
//   n = l + 64 r  (lane l, register r = 0..7)      k = ka + 8 kc + 32 kh + 64 km
//   stage 1   DFT8 over r -> ka in registers;  twiddle W_512^(l ka)
//   swap      lane bits 5,4 <-> register bits 2,1 (v_permlane32_swap / v_permlane16_swap):
//             registers now hold (l5 l4 | a0), lanes (a2 a1 | l3..l0)
//   stage 2   DFT4 over (l5 l4) -> kc;  twiddle W_64^((l & 15) kc)
//   stage 3a  radix 2 over lane bit 3 (DPP row_ror:8, no LDS) -> kh;  twiddle W_16^((l & 7) kh)
//   exchange  8 x 8 transpose inside every group of 8 lanes, through LDS:
//             registers <- (l2 l1 l0), lanes <- (a2 a1 | kh | kc1 kc0 a0)
//   stage 3b  DFT8 over (l2 l1 l0) -> km in registers, no twiddle
//
// Lane t ends with Z[lam(t) + 64 km] in register km, where
//   lam(t) = 4 t5 + 2 t4 + 32 t3 + 16 t2 + 8 t1 + t0
// (the Stockham plan 8.8.8 of k_xspec<512> moves the row through LDS three times and once
// more for the split: 72 LDS instructions per row, the CU's LDS unit 78 % busy).
// --------------------------------------------------------------------------
__device__ __forceinline__ int fftq512_lambda(int t) {
    return 4 * ((t >> 5) & 1) + 2 * ((t >> 4) & 1) + 32 * ((t >> 3) & 1) + 16 * ((t >> 2) & 1) + 8 * ((t >> 1) & 1) + (t & 1);
}
__device__ __forceinline__ int fftq512_lane_of(int lam) {
    return 32 * ((lam >> 2) & 1) + 16 * ((lam >> 1) & 1) + 8 * ((lam >> 5) & 1) + 4 * ((lam >> 4) & 1) + 2 * ((lam >> 3) & 1) + (lam & 1);
}
constexpr int FFTQ512_LDS_ELEMS = 8 * 72;      // eight groups of 8 lanes x (8 x 9) elements

// the value of lane (l ^ 8) of the same row of 16 lanes (DPP row_ror:8)
__device__ __forceinline__ double lane_xor8(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// t1 = W_512^tid, t2 = W_64^(tid & 15), t3 = W_16^(tid & 7).  mid() is called once: WHEN = 0 after
// stage 1's butterflies (the 8 inputs are dead), 3 after the transpose (before the last stage).
// power (optional) += this lane's share of sum_{k=1}^{M-1} |Z_k|^2 + (Re Z_0 - Im Z_0)^2.
template <int WHEN = 0, typename Mid>
__device__ __forceinline__ void fftq512(cplx (&v)[8], cplx* lds, const cplx t1, const cplx t2, const cplx t3, int tid,
                                        double* power, Mid mid) {
    // ---- stage 1 ----
    dft_reg<8>(v);
    if (WHEN == 0) mid();
    {
        // v[j] *= t1^j, every power formed once
        cplx wq[8];
        wq[1] = t1;
#pragma unroll
        for (int j = 2; j < 8; ++j) {
            if (j % 2 == 0) {
                const cplx h = wq[j / 2];
                wq[j] = make_double2(fma(h.x, h.x, -h.y * h.y), 2.0 * h.x * h.y);
            } else wq[j] = cmul(wq[j - 1], wq[1]);
        }
#pragma unroll
        for (int j = 1; j < 8; ++j) v[j] = cmul(v[j], wq[j]);
    }
    // ---- lane bits 5,4 <-> register bits 2,1 ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lane_swap32(v[j].x, v[j + 4].x);
        lane_swap32(v[j].y, v[j + 4].y);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j & 2) continue;
        lane_swap16(v[j].x, v[j + 2].x);
        lane_swap16(v[j].y, v[j + 2].y);
    }
    // ---- stage 2: DFT4 over register bits 2,1; twiddle t2^kc ----
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    {
        const cplx w2 = make_double2(fma(t2.x, t2.x, -t2.y * t2.y), 2.0 * t2.x * t2.y);
        const cplx w3 = cmul(w2, t2);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            v[2 + c] = cmul(v[2 + c], t2);
            v[4 + c] = cmul(v[4 + c], w2);
            v[6 + c] = cmul(v[6 + c], w3);
        }
    }
    // ---- stage 3a: radix 2 over lane bit 3 (partner = lane ^ 8); twiddle t3 on the upper lanes ----
    {
        const bool up = (tid & 8) != 0;
        const double sg = up ? -1.0 : 1.0;
        const cplx w = up ? t3 : make_double2(1.0, 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cplx p = make_double2(lane_xor8(v[j].x), lane_xor8(v[j].y));
            const cplx d = make_double2(fma(sg, v[j].x, p.x), fma(sg, v[j].y, p.y));   // own + partner | partner - own
            v[j] = cmul(d, w);
        }
    }
    // ---- 8 x 8 transpose inside every group of 8 lanes ----
    {
        cplx* wbase = lds + (tid >> 3) * 72 + 9 * (tid & 7);
#pragma unroll
        for (int j = 0; j < 8; ++j) wbase[j] = v[j];
        lds_sync<64>();
        const cplx* rbase = lds + (tid >> 3) * 72 + (tid & 7);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rbase[9 * j];
        lds_sync<64>();
    }
    if (WHEN == 3) mid();
    // ---- stage 3b ----
    dft_reg<8>(v);
    if (power) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j == 0 && tid == 0) { const double dM = v[0].x - v[0].y; acc += dM * dM; }
            else acc += cnorm(v[j]);
        }
        *power += acc;
    }
}

// the plan of a row length behind one interface (k_xspec_qf)
template <int M> struct FftQ;
template <> struct FftQ<1024> {
    static constexpr int R = 16, LDS_ELEMS = FFTQ_LDS_ELEMS;
    static __device__ __forceinline__ int lambda(int t) { return fftq_lambda(t); }
    static __device__ __forceinline__ int lane_of(int lam) { return fftq_lane_of(lam); }
    template <int WHEN, typename Mid>
    static __device__ __forceinline__ void run(cplx (&v)[16], cplx* lds, const cplx* twB, int tid, double* power, Mid mid) {
        const cplx t1 = twB[2 * tid], t2 = twB[32 * (tid & 15)];
        fftq1024<WHEN>(v, lds, t1, t2, tid, power, mid);
    }
};
template <> struct FftQ<512> {
    static constexpr int R = 8, LDS_ELEMS = FFTQ512_LDS_ELEMS;
    static __device__ __forceinline__ int lambda(int t) { return fftq512_lambda(t); }
    static __device__ __forceinline__ int lane_of(int lam) { return fftq512_lane_of(lam); }
    template <int WHEN, typename Mid>
    static __device__ __forceinline__ void run(cplx (&v)[8], cplx* lds, const cplx* twB, int tid, double* power, Mid mid) {
        // (twB = W_1024^j: W_512^tid, W_64^(tid & 15), W_16^(tid & 7))
        const cplx t1 = twB[2 * tid], t2 = twB[16 * (tid & 15)], t3 = twB[64 * (tid & 7)];
        fftq512<WHEN>(v, lds, t1, t2, t3, tid, power, mid);
    }
};

}  // namespace pp
