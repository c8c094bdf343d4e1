// Shared device helpers for the wideband-TOA kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pp {

// reference pplib.py:48,51: "traditional" dispersion constant, bit-identical
// to Python's 0.000241**-1 (= 4149.377593360996)
#define PP_DCONST 4149.377593360996
#define PP_TWO_PI 6.283185307179586476925286766559
#define PP_LN10 2.302585092994045684017991454684

typedef double2 cplx;

#ifndef PP_FAST_SINCOS
#define PP_FAST_SINCOS 1
#endif

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) {
    return make_double2(fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ cplx cmul_mi(cplx a) { return make_double2(a.y, -a.x); }
__device__ __forceinline__ double cnorm(cplx a) { return fma(a.x, a.x, a.y * a.y); }

// Pointers fetched from a pointer table in memory are generic ("flat") to the
// compiler; flat loads count on both vmcnt and lgkmcnt and force a full drain of
// the vector-memory queue before use.  Everything here lives in global memory.
template <typename T>
__device__ __forceinline__ const T* as_global(const T* p) {
    typedef const T __attribute__((address_space(1)))* gp_t;
    return (const T*)(gp_t)(uintptr_t)p;
}

// A value every lane wants from the same address, fetched by the SCALAR unit (constant address space: s_load through
// the scalar cache) -- it does not queue in the CU's vector-memory path behind the rows in flight.  Only for data no
// kernel writes while this one runs (the scalar cache is not coherent within a launch).
template <typename T>
__device__ __forceinline__ T load_uniform(const T* p) {
    typedef const T __attribute__((address_space(4)))* cp_t;
    return *(cp_t)(uintptr_t)p;
}

// One rounding per operation, as NumPy does it: hipcc contracts a * b + c into an FMA
// even through __dmul_rn / __dadd_rn, which moves a Nelder-Mead vertex by an ulp -- and
// turns the grid point 18 * (1/36) - 0.5 = 0 into -2.8e-17, sending SciPy's simplex down
// its "x0 != 0" branch.  The empty asm pins the intermediate result.
__device__ __forceinline__ double rn_pin(double r) { asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ double mul_rn(double a, double b) { return rn_pin(a * b); }
__device__ __forceinline__ double add_rn(double a, double b) { return rn_pin(a + b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return rn_pin(a - b); }

// 1/x for x >= 1 (no zeros or denormals to honour; huge and infinite x give 0): the
// hardware estimate refined by two Newton steps, ~1 ulp, instead of the dozen
// instructions of the IEEE division sequence
__device__ __forceinline__ double recip_ge1(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return (x < 1e300) ? y : 0.0;      // (the refinement of 1/inf would be inf * 0)
}

// sin and cos of 2 pi t for |t| <= 1/2 (a little beyond is fine): quarter-turn
// reduction, then Taylor polynomials on |y| <= 1/8 (truncation < 1e-19; a few
// rounding errors of 2^-53).  A third of the instructions of the library
// sincospi, whose own range reduction is redundant after ours.
// A constant held in a scalar register pair: the compiler otherwise builds every
// 64-bit literal in VGPRs (two v_mov_b32 per use, on the f64-bound VALU); the
// scalar moves issue beside the vector instructions and a VOP3 takes one SGPR operand.
template <bool SC>
__device__ __forceinline__ double kconst(double v) {
    if (SC) asm volatile("" : "+s"(v));
    return v;
}
// SC: polynomial coefficients from scalar registers (k_xspec: -30 VALU per row)
template <bool SC = false>
__device__ __forceinline__ void sincos_turns(double t, double* sn, double* cs) {
    const double q = rint(4.0 * t);
    const double y = fma(-0.25, q, t);          // exact
    const double u = y * y;
    double ps = kconst<SC>(0x1.aaec32af93359p-4);
    ps = fma(ps, u, kconst<SC>(-0x1.6fadb9f155744p-1));
    ps = fma(ps, u, kconst<SC>(0x1.e8f434d018d63p+1));
    ps = fma(ps, u, kconst<SC>(-0x1.e3074fde8871fp+3));
    ps = fma(ps, u, kconst<SC>(0x1.50783487ee782p+5));
    ps = fma(ps, u, kconst<SC>(-0x1.32d2cce62bd86p+6));
    ps = fma(ps, u, kconst<SC>(0x1.466bc6775aae2p+6));
    ps = fma(ps, u, kconst<SC>(-0x1.4abbce625be53p+5));
    ps = fma(ps, u, kconst<SC>(0x1.921fb54442d18p+2));
    const double s0 = ps * y;
    double pc = kconst<SC>(0x1.20c62c2f2d7f5p-2);
    pc = fma(pc, u, kconst<SC>(-0x1.b6e24f44b128fp+0));
    pc = fma(pc, u, kconst<SC>(0x1.f9d38a3763cc3p+2));
    pc = fma(pc, u, kconst<SC>(-0x1.a6d1f2a204a8cp+4));
    pc = fma(pc, u, kconst<SC>(0x1.e1f506891babbp+5));
    pc = fma(pc, u, kconst<SC>(-0x1.55d3c7e3cbffap+6));
    pc = fma(pc, u, kconst<SC>(0x1.03c1f081b5ac4p+6));
    pc = fma(pc, u, kconst<SC>(-0x1.3bd3cc9be45dep+4));
    const double c0 = fma(pc, u, 1.0);
    const int qi = (int)q;
    const bool odd = qi & 1;
    const double so = odd ? c0 : s0, co = odd ? s0 : c0;
    *sn = (qi & 2) ? -so : so;
    *cs = ((qi + 1) & 2) ? -co : co;
}

// exp(2 pi i k phi) with the product reduced modulo 1 before the sincos:
// k*phi is formed exactly (fma residual), so large non-dedispersed phase
// shifts do not lose the fraction.
template <bool SC = false>
__device__ __forceinline__ cplx unit_phasor(double k, double phi) {
    double pfrac = phi - rint(phi);          // exact
    double prod = k * pfrac;
    double err = fma(k, pfrac, -prod);       // exact residual of the product
    double r = (prod - rint(prod)) + err;
    double s, c;
#if PP_FAST_SINCOS
    sincos_turns<SC>(r, &s, &c);
#else
    sincospi(2.0 * r, &s, &c);
#endif
    return make_double2(c, s);
}

// sum over the lanes of an aligned group of W lanes (W = 2^n <= 64)
template <int W>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int W>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum 16 per-lane values over the 64 lanes of a wave with 17 exchanges instead of
// 96: each xor step halves the values a lane carries (it keeps the half selected
// by that lane bit and adds the partner's copy of it).  On return lane l holds
// the wave total of s[wave_reduce16_index(l)].
__device__ __forceinline__ int wave_reduce16_index(int lane) { return (lane >> 2) & 15; }
template <int NS>   // NS <= 16 values given, the rest are zero
__device__ __forceinline__ double wave_reduce16(const double (&s)[NS], int lane) {
    double t8[8], t4[4], t2[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double hi = (j + 8 < NS) ? s[j + 8] : 0.0;
        const double keep = b5 ? hi : s[j], send = b5 ? s[j] : hi;
        t8[j] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double keep = b4 ? t8[j + 4] : t8[j], send = b4 ? t8[j] : t8[j + 4];
        t4[j] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double keep = b3 ? t4[j + 2] : t4[j], send = b3 ? t4[j] : t4[j + 2];
        t2[j] = keep + __shfl_xor(send, 8, 64);
    }
    const double keep = b2 ? t2[1] : t2[0], send = b2 ? t2[0] : t2[1];
    double v = keep + __shfl_xor(send, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;   // index 8*b5 + 4*b4 + 2*b3 + b2 = (lane >> 2) & 15
}

// The same reduction through LDS: every lane stores its NS values, lane 4q + p
// (q < NS, p < 4) adds up value q of lanes 16p .. 16p+15, and two quad exchanges
// finish.  ~3 NS + 22 instructions instead of ~230; `buf` holds NS * 68 doubles
// (rows padded so that the strided reads spread over the banks).  On return lane l
// holds the wave total of s[(l >> 2)] for l < 4 NS (same indexing as above).
template <int NS>
__device__ __forceinline__ double wave_reduce_lds(const double (&s)[NS], int lane, double* buf) {
    static_assert(NS <= 16, "one value per lane quad");
    const int wpos = (lane >> 4) * 17 + (lane & 15);
#pragma unroll
    for (int q = 0; q < NS; ++q) buf[q * 68 + wpos] = s[q];
    const int q = min(lane >> 2, NS - 1), part = lane & 3;
    const double* src = buf + q * 68 + part * 17;
    double v = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) v += src[t];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}
#define PP_WRED_DOUBLES(NS) ((NS) * 68)

// block-wide sum of NV values held by every thread; result valid in all
// threads.  scratch must hold (blockDim.x/64)*NV doubles.
template <int NV>
__device__ inline void block_sum(double (&v)[NV], double* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = group_sum<64>(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wid * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += scratch[w * NV + i];
        v[i] = s;
    }
}

// The same block-wide sums -- the same additions in the same order, so bitwise the same totals --
// with the transposing wave reduction (17 exchanges per 16 values instead of 96) and ONE barrier:
// `scratch` holds two buffers of (blockDim.x/64) * (NV + 1) doubles used alternately (`flip`, which
// the caller keeps between calls, starts at 0): a wave can be at most one call ahead of the slowest
// one, so the buffer it writes is never one still being read.  mx (optional): a value reduced by max.
template <int NV, int C0>
__device__ __forceinline__ void wave_totals_to(const double (&v)[NV], int lane, double* dst) {
    constexpr int n = (NV - C0) < 16 ? (NV - C0) : 16;
    double part[n];
#pragma unroll
    for (int j = 0; j < n; ++j) part[j] = v[C0 + j];
    const double tot = wave_reduce16<n>(part, lane);
    if ((lane & 3) == 0 && (lane >> 2) < n) dst[C0 + (lane >> 2)] = tot;
    if constexpr (C0 + 16 < NV) wave_totals_to<NV, C0 + 16>(v, lane, dst);
}
#define PP_BSUM_DOUBLES(NW, NVMAX) (2 * (NW) * ((NVMAX) + 1))
// NVMAX: the largest NV among the kernel's calls on this scratch (the two buffers are NW * (NVMAX + 1) doubles apart
// whatever the call's own NV: buffers of different sizes would overlap the one a slow wave still reads)
template <int NV, int NVMAX>
__device__ __forceinline__ void block_sum_t(double (&v)[NV], double* scratch, int& flip, double* mx = nullptr) {
    static_assert(NV <= NVMAX, "scratch sized for NVMAX values");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double* buf = scratch + flip * nw * (NVMAX + 1);
    flip ^= 1;
    wave_totals_to<NV, 0>(v, lane, buf + wid * (NV + 1));
    if (mx) {
        const double m = group_max<64>(*mx);
        if (lane == 0) buf[wid * (NV + 1) + NV] = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += buf[w * (NV + 1) + i];
        v[i] = s;
    }
    if (mx) {
        double m = buf[NV];
        for (int w = 1; w < nw; ++w) m = fmax(m, buf[w * (NV + 1) + NV]);
        *mx = m;
    }
}

// ---- per-subint solver state (device resident) ---------------------------
struct SubState {
    double x[5];       // accepted parameters (at the fit reference frequencies)
    double xe[5];      // parameters the next evaluation is made at
    double f;          // objective at x
    double g[5];
    double H[25];
    double f0, g0[5], H0[25];   // objective at init_params (parity hooks)
    double radius;
    double pred_red;   // predicted reduction of the pending proposal
    int hits_boundary;
    int iter, nfev, status, done, cur;  // cur: csum buffer of the accepted point; nfev: objective evaluations as
                                        // SciPy's nfev counts them (the reference's nfeval, pptoaslib.py:1017)
    int npass;         // ... of which passes over the portraits / the stored cross-spectrum
    double xl[5], fl, gl[5], Hl[25];   // the point evaluated last and its f, g, H (SciPy's cache of one point)
    int fresh;         // the next evaluation is 1: an initial one, 2: the closing one (no ratio test)
    int recentred;     // one-pass flow: 1 = the Taylor model was taken again about the first solve's answer
    int nmodel;        // model passes of this subint whose iteration left the model's range
    int model;         // scattering model of the closing iterations: 0 not yet, 1 the next evaluation
                       // is the model pass, 3 not (again) for this subint, 4 / 5 a model pass was just
                       // abandoned (k_step turns them into 0 / 3)
    double xprev[5];   // the accepted point before the last accepted step (convergence-rate estimate)
    double geo[4];     // max |d phi_n/d DM|, |d phi_n/d GM|, |ln(nu_n/nu_tau)| over the channels; template keff
};

// number of per-subint accumulators of one evaluation: f, g[5], H upper[15]
#define PP_NACC 21
// channels per workgroup of the kernels that sum over channels (evaluators, seed accumulation, moments):
// a property of the band, never of the batch (see fit_chunk's `chunking`)
#ifndef PP_CHUNK_CHANNELS
#define PP_CHUNK_CHANNELS 256
#endif
// raw per-channel sums kept for the post-fit stage
#define PP_NCS 9   // A0 A1 A2 T1 T2 A1T S0 S1 S2
// order of the per-channel Taylor model of C_n(phi_n) about the initial point:
// A_0 .. A_PP_TJ (derivatives) + a rigorous remainder coefficient
#define PP_TJ 10
#define PP_TSTRIDE (PP_TJ + 2)
// The Taylor rows in HBM.  Row-major (default): 12 doubles per row, 96 B apart -- the transform writes a row as one
// 96-byte piece.  PP_TAY_BLOCKED = 1: blocks of 64 rows with the coefficients in pairs, [row / 64][pair 0..5][row % 64][2],
// so that the solve, where lane l works on row r0 + l, reads 1 KB contiguous per load instruction and its evaluation at
// the expansion point only two of the six pairs.  Measured (profiles/README.md, round 4): the solve at 4096 channels is
// bound by the bytes it re-reads, not by how it touches the lines (0.44 against 0.45-0.46 ms), while the transform pays
// 6 more VALU instructions per row for the addresses and writes six pieces instead of one: +1 % of 14.4 ms.  Kept as a
// build option; the buffer holds a multiple of 64 rows either way.
#ifndef PP_TAY_BLOCKED
#define PP_TAY_BLOCKED 0
#endif
#if PP_TAY_BLOCKED
#define PP_TAY_PAIR_STRIDE 128     // doubles between the coefficient pairs of a row
__device__ __forceinline__ size_t tay_idx(size_t row, int q) {
    return (row >> 6) * (size_t)(64 * PP_TSTRIDE) + (size_t)(q >> 1) * 128 + (row & 63) * 2 + (q & 1);
}
#else
#define PP_TAY_PAIR_STRIDE 2
__device__ __forceinline__ size_t tay_idx(size_t row, int q) { return row * PP_TSTRIDE + (size_t)q; }
#endif

}  // namespace pp
