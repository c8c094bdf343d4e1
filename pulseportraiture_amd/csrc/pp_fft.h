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

// The radix-16 butterfly in two halves.  After the first (the four column DFT4s) none
// of the 16 inputs is live any more: a caller that streams rows can queue the next
// row's loads into those registers there, a quarter of the way into the stage,
// instead of after its twiddles and stores (k_xspec: the wave then has a row of HBM
// loads in flight for all but ~60 instructions of every row).
__device__ __forceinline__ void dft16_first(cplx (&v)[16]) {
    // F_q[j2] = DFT4 over m of v[4m + q]   (stored back in v[4*j2 + q])
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4(v[q], v[4 + q], v[8 + q], v[12 + q]);
}
__device__ __forceinline__ void dft16_second(cplx (&v)[16]) {
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178;
    const double h = 0.70710678118654752440;
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
template <>
__device__ __forceinline__ void dft_reg<16>(cplx (&v)[16]) {
    dft16_first(v);
    dft16_second(v);
}

// "nothing to do in the middle of the stage"
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <typename F> struct IsNoMid { static constexpr bool value = false; };
template <> struct IsNoMid<NoMid> { static constexpr bool value = true; };

template <int PADLOG>
__device__ __forceinline__ int lds_pad(int i) { return i + (i >> PADLOG); }

// Make the workgroup's LDS traffic visible to all its lanes WITHOUT draining
// the vector-memory counter: __syncthreads() lowers to
// `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`, which would stall on the
// prefetched HBM loads of the next row.  A one-wave workgroup (T == 64) needs
// no barrier at all: its LDS operations retire in order.
#ifndef PP_WAVE_SYNC_WAITS
#define PP_WAVE_SYNC_WAITS 0    // 1: a one-wave workgroup still drains lgkmcnt at every sync
#endif
template <int T>
__device__ __forceinline__ void lds_sync() {
    if (T == 64) {
        // one wave: the LDS unit executes its accesses in program order, so a read
        // that follows a write in the instruction stream sees it; only the compiler
        // has to be kept from reordering.  (Waiting here for lgkmcnt(0) would stop
        // the wave until EVERY outstanding read has returned, where the compiler's
        // own counted waits let the first butterfly start as its operands arrive.)
        if (PP_WAVE_SYNC_WAITS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("" ::: "memory");
    } else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- one Stockham stage, split into its load and its finish ----------------
// S = product of the radices already applied.
// twB[k] = exp(-2 pi i k / B), k = 0..M  (W_M^j = twB[2j]).
template <int M, int T, int R>
struct StageGeom {
    static constexpr int NBF = M / R;                 // butterflies in the stage
    static constexpr int PER = (NBF + T - 1) / T;     // per thread
};

// raw first-stage operands straight from HBM (kept in the input precision so a
// prefetched row costs half the registers for f32 portraits)
template <typename Tin> struct RawOf;
template <> struct RawOf<double> { typedef double2 type; };
template <> struct RawOf<float> { typedef float2 type; };
template <> struct RawOf<cplx> { typedef double2 type; };

__device__ __forceinline__ double2 load_raw(const double* row, int idx) {
    return *reinterpret_cast<const double2*>(row + 2 * idx);
}
__device__ __forceinline__ float2 load_raw(const float* row, int idx) {
    return *reinterpret_cast<const float2*>(row + 2 * idx);
}
__device__ __forceinline__ double2 load_raw(const cplx* row, int idx) { return row[idx]; }
__device__ __forceinline__ cplx to_cplx(double2 v) { return v; }
__device__ __forceinline__ cplx to_cplx(float2 v) { return make_double2((double)v.x, (double)v.y); }

// A portrait row is read exactly once per pass: its loads are marked non-temporal so that
// the stream does not displace the template rows, twiddles and per-channel tables the
// same CU keeps re-reading (k_xspec_q1024: 14.3-14.6 -> 14.0-14.1 ms per 1024 fits).
#ifndef PP_NT_ROW_LOADS
#define PP_NT_ROW_LOADS 1
#endif
template <typename Raw>
__device__ __forceinline__ Raw load_row_once(const char* pa) {
#if PP_NT_ROW_LOADS
    // (the builtin takes native vectors)
    typedef double nvd2 __attribute__((ext_vector_type(2)));
    typedef float nvf2 __attribute__((ext_vector_type(2)));
    Raw r;
    if constexpr (sizeof(Raw) == 16) {
        const nvd2 t = __builtin_nontemporal_load(reinterpret_cast<const nvd2*>(pa));
        r.x = t.x; r.y = t.y;
    } else {
        const nvf2 t = __builtin_nontemporal_load(reinterpret_cast<const nvf2*>(pa));
        r.x = t.x; r.y = t.y;
    }
    return r;
#else
    return *reinterpret_cast<const Raw*>(pa);
#endif
}

template <int M, int T, int R, typename Tin, typename Raw, int PER>
__device__ __forceinline__ void stage_load_global(Raw (&v)[PER][R], const Tin* __restrict__ grow, int tid) {
    constexpr int NBF = StageGeom<M, T, R>::NBF;
    static_assert(PER == StageGeom<M, T, R>::PER, "register tile does not match the stage");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = tid + T * i;
        if (NBF % T == 0 || t < NBF) {
            // (uniform row base + constant) + 32-bit unsigned lane offset: selects
            // the SGPR-base addressing mode, no 64-bit vector address per load
            const char* gb = reinterpret_cast<const char*>(grow);
            const unsigned boff = (unsigned)t * (unsigned)sizeof(Raw);
#pragma unroll
            for (int k = 0; k < R; ++k) v[i][k] = load_row_once<Raw>(gb + (size_t)(k * NBF) * sizeof(Raw) + boff);
        }
    }
}

// LDS addresses are written as (one per-thread base) + (compile-time offset) so
// they fold into the ds_read/ds_write immediate field: the per-element padded
// indices are affine in the element number because every stride is a multiple
// of the padding period 2^PADLOG.
template <int M, int T, int R, int PADLOG, int PER>
__device__ __forceinline__ void stage_load_lds(cplx (&v)[PER][R], const cplx* lds, int tid) {
    constexpr int NBF = StageGeom<M, T, R>::NBF;
    static_assert(PER == StageGeom<M, T, R>::PER, "register tile does not match the stage");
    static_assert(NBF % (1 << PADLOG) == 0 || NBF < (1 << PADLOG), "stride must keep the padding affine");
    constexpr int KSTEP = (NBF % (1 << PADLOG) == 0) ? NBF + (NBF >> PADLOG) : 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = tid + T * i;
        if (NBF % T == 0 || t < NBF) {
            const cplx* base = lds + lds_pad<PADLOG>(t);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (KSTEP) v[i][k] = base[k * KSTEP];
                else v[i][k] = lds[lds_pad<PADLOG>(t + k * NBF)];
            }
        }
    }
}

// butterflies + inter-stage twiddles + in-place write + LDS sync.
// tw[i] = W_M^(t - t%S) of this thread's i-th butterfly (unused in the last
// stage): loop-invariant per thread, so callers hoist it out of their row loop.
// mid(): called once, as soon as the stage's input registers are dead (after the first
// half of a single radix-16 butterfly, else after the last butterfly).
template <int M, int T, int R, int S, int PADLOG, int PER, typename TW, bool TREE = false, typename Mid = NoMid>
__device__ __forceinline__ void stage_finish(cplx (&v)[PER][R], cplx* lds, const TW& tw, int tid,
                                             double* power = nullptr, Mid mid = Mid()) {
    constexpr int NBF = StageGeom<M, T, R>::NBF;
    static_assert(PER == StageGeom<M, T, R>::PER, "register tile does not match the stage");
    constexpr bool LAST = (S * R == M);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = tid + T * i;
        if (NBF % T == 0 || t < NBF) {
            if constexpr (R == 16 && PER == 1 && !IsNoMid<Mid>::value) {
                dft16_first(v[i]);
                mid();
                dft16_second(v[i]);
            } else {
                dft_reg<R>(v[i]);
                // (only where every lane runs the butterfly: mid() is wave-uniform work)
                if constexpr (!IsNoMid<Mid>::value && NBF % T == 0) { if (i == PER - 1) mid(); }
            }
            const int q = t & (S - 1);
            const int ob = q + S * R * (t / S);
            if constexpr (!LAST) {
#ifndef PP_TWIDDLE_TREE
#define PP_TWIDDLE_TREE 1
#endif
                if (PP_TWIDDLE_TREE && TREE) {
                    // v[j] *= w^j with every power formed once, w^j = (w^(j/2))^2 or
                    // w^(j-1) w: R - 2 products for the powers + R - 1 for the elements
                    // (29 for R = 16 where the bitwise scheme below takes 35), at most
                    // log2(R) + 1 roundings per power; a few powers live at a time
                    cplx wq[R];
                    wq[1] = tw[i];
#pragma unroll
                    for (int j = 2; j < R; ++j) {
                        if (j % 2 == 0) {
                            const cplx h = wq[j / 2];
                            wq[j] = make_double2(fma(h.x, h.x, -h.y * h.y), 2.0 * h.x * h.y);
                        } else wq[j] = cmul(wq[j - 1], wq[1]);
                    }
#pragma unroll
                    for (int j = 1; j < R; ++j) v[i][j] = cmul(v[i][j], wq[j]);
                } else {
                // v[j] *= w^j as the product of w^(2^b) over the set bits of j:
                // one twiddle power live at a time (register pressure), at most
                // log2(R) roundings per element
                cplx wp = tw[i];
#pragma unroll
                for (int b = 1; b < R; b <<= 1) {
#pragma unroll
                    for (int j = 1; j < R; ++j)
                        if (j & b) v[i][j] = cmul(v[i][j], wp);
                    if (2 * b < R) wp = cmul(wp, wp);
                }
                }
            }
            if (LAST && power) {
                // sum_{k=1}^{M-1} |Z_k|^2 + (Re Z_0 - Im Z_0)^2 = sum_{k=1}^{M} |d_k|^2
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (j == 0 && ob == 0) { const double dM = v[i][0].x - v[i][0].y; acc += dM * dM; }
                    else acc += cnorm(v[i][j]);
                }
                *power += acc;
            }
            // pad(ob + S*j) = pad(ob) + j*(S + S/2^PADLOG) for S a multiple of the
            // padding period; for the first stage (S = 1, R = 2^PADLOG) it is
            // pad(ob) + j
            cplx* wbase = lds + lds_pad<PADLOG>(ob);
            constexpr int JSTEP = (S % (1 << PADLOG) == 0) ? S + (S >> PADLOG) : ((S == 1 && R == (1 << PADLOG)) ? 1 : 0);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (JSTEP) wbase[j * JSTEP] = v[i][j];
                else lds[lds_pad<PADLOG>(ob + S * j)] = v[i][j];
            }
        }
    }
    if constexpr (!IsNoMid<Mid>::value && NBF % T != 0) mid();
    lds_sync<T>();
}

// twiddles of one non-last stage for this thread
template <int M, int T, int R, int S, int PER>
__device__ __forceinline__ void stage_twiddles(cplx (&tw)[PER], const cplx* __restrict__ twB, int tid) {
    constexpr int NBF = StageGeom<M, T, R>::NBF;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = tid + T * i;
        tw[i] = (NBF % T == 0 || t < NBF) ? twB[2 * (t - (t & (S - 1)))] : make_double2(1.0, 0.0);
    }
}

// a later stage: LDS -> registers -> LDS
template <int M, int T, int R, int S, int PADLOG, typename TW>
__device__ __forceinline__ void stage_lds(cplx* lds, const TW& tw, int tid, double* power = nullptr) {
    cplx v[StageGeom<M, T, R>::PER][R];
    stage_load_lds<M, T, R, PADLOG>(v, lds, tid);
    lds_sync<T>();     // every read of this stage before any in-place write
    stage_finish<M, T, R, S, PADLOG>(v, lds, tw, tid, power);
}

// Plans: T threads per row and up to four radices R1*R2*R3*R4 = M (1 = unused).
// Radices are <= 8 and T = M/8 for M >= 512, so every thread does one radix-8
// butterfly per stage and never holds more than 8 complex values (plus 8 more
// of the prefetched next row): registers stay low enough for several waves per
// SIMD.  The LDS image is padded by one element every R1.
template <int M>
struct FftPlan;
#define PP_PLAN(M_, T_, R1_, R2_, R3_, R4_, PL_)                                                 \
    template <> struct FftPlan<M_> {                                                             \
        static constexpr int T = T_, R1 = R1_, R2 = R2_, R3 = R3_, R4 = R4_, PADLOG = PL_;       \
        static constexpr int LDS_ELEMS = M_ + (M_ >> PL_) + 1;                                   \
        static constexpr int PER1 = StageGeom<M_, T_, R1_>::PER;                                 \
        static constexpr int PER2 = StageGeom<M_, T_, R2_>::PER;                                 \
        static constexpr int PER3 = StageGeom<M_, T_, (R3_ > 1 ? R3_ : 2)>::PER;                 \
        static_assert(R1_ * R2_ * R3_ * R4_ == M_, "radices must multiply to M");                \
    }
PP_PLAN(16, 64, 4, 4, 1, 1, 2);
PP_PLAN(32, 64, 8, 4, 1, 1, 3);
PP_PLAN(64, 64, 8, 8, 1, 1, 3);
PP_PLAN(128, 64, 8, 4, 4, 1, 3);
PP_PLAN(256, 64, 8, 8, 4, 1, 3);
PP_PLAN(512, 64, 8, 8, 8, 1, 3);
#ifndef PP_PLAN1024
#define PP_PLAN1024 64, 16, 8, 8, 1, 4
#endif
#define PP_PLAN_X(M_, ...) PP_PLAN(M_, __VA_ARGS__)
PP_PLAN_X(1024, PP_PLAN1024);
PP_PLAN(2048, 256, 8, 8, 8, 4, 3);
PP_PLAN(4096, 512, 8, 8, 8, 8, 3);
#undef PP_PLAN

// per-thread, row-independent twiddles of the non-last stages
template <int M>
struct RowTwiddles {
    cplx t1[FftPlan<M>::PER1];
    cplx t2[FftPlan<M>::PER2];
    cplx t3[FftPlan<M>::PER3];
};

template <int M>
__device__ __forceinline__ void load_row_twiddles(RowTwiddles<M>& tw, const cplx* __restrict__ twB, int tid) {
    typedef FftPlan<M> P;
    stage_twiddles<M, P::T, P::R1, 1>(tw.t1, twB, tid);
    if constexpr (P::R3 > 1) stage_twiddles<M, P::T, P::R2, P::R1>(tw.t2, twB, tid);
    if constexpr (P::R4 > 1) stage_twiddles<M, P::T, P::R3, P::R1 * P::R2>(tw.t3, twB, tid);
}

// make the compiler forget what it knows about the twiddle registers
template <int M>
__device__ __forceinline__ void opaque_twiddles(RowTwiddles<M>& tw) {
    typedef FftPlan<M> P;
#pragma unroll
    for (int j = 0; j < P::PER1; ++j) asm volatile("" : "+v"(tw.t1[j].x), "+v"(tw.t1[j].y));
    if constexpr (P::R3 > 1) {
#pragma unroll
        for (int j = 0; j < P::PER2; ++j) asm volatile("" : "+v"(tw.t2[j].x), "+v"(tw.t2[j].y));
    }
    if constexpr (P::R4 > 1) {
#pragma unroll
        for (int j = 0; j < P::PER3; ++j) asm volatile("" : "+v"(tw.t3[j].x), "+v"(tw.t3[j].y));
    }
}

// first stage from registers (TREE: twiddle powers by the product tree -- fewer
// multiplications, a few more registers; for callers with registers to spare there)
template <int M, bool TREE = false, int PER1_, int R1_, typename Mid = NoMid>
__device__ __forceinline__ void fft_first_stage(cplx* lds, cplx (&v)[PER1_][R1_], const RowTwiddles<M>& tw,
                                                int tid, Mid mid = Mid()) {
    typedef FftPlan<M> P;
    static_assert(PER1_ == P::PER1 && R1_ == P::R1, "first-stage tile mismatch");
    stage_finish<M, P::T, P::R1, 1, P::PADLOG, PER1_, decltype(tw.t1), TREE, Mid>(v, lds, tw.t1, tid, nullptr, mid);
}

// the remaining stages, LDS to LDS
// `power` (optional) receives this thread's share of sum_{k=1}^{M} |d_k|^2, taken
// from the last stage's registers.
template <int M>
__device__ __forceinline__ void fft_later_stages(cplx* lds, const RowTwiddles<M>& tw, int tid,
                                                 double* power = nullptr) {
    typedef FftPlan<M> P;
    stage_lds<M, P::T, P::R2, P::R1, P::PADLOG>(lds, tw.t2, tid, power);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P::R3 > 1) stage_lds<M, P::T, P::R3, P::R1 * P::R2, P::PADLOG>(lds, tw.t3, tid, power);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P::R4 > 1) stage_lds<M, P::T, P::R4, P::R1 * P::R2 * P::R3, P::PADLOG>(lds, tw.t3, tid, power);
}

// Complex FFT of the M = B/2 packed pairs of one real row; result Z[0..M-1] is
// left in `lds` (padded with lds_pad<FftPlan<M>::PADLOG>).
template <int M, typename Tin>
__device__ __forceinline__ void fft_row(cplx* lds, const Tin* __restrict__ grow,
                                        const cplx* __restrict__ twB, int tid) {
    constexpr int T = FftPlan<M>::T, R1 = FftPlan<M>::R1, PER1 = FftPlan<M>::PER1;
    typename RawOf<Tin>::type raw[PER1][R1];
    stage_load_global<M, T, R1>(raw, grow, tid);
    RowTwiddles<M> tw;
    load_row_twiddles<M>(tw, twB, tid);
    cplx v[PER1][R1];
#pragma unroll
    for (int i = 0; i < PER1; ++i)
#pragma unroll
        for (int k = 0; k < R1; ++k) v[i][k] = to_cplx(raw[i][k]);
    fft_first_stage<M>(lds, v, tw, tid);
    fft_later_stages<M>(lds, tw, tid);
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

// same with the twiddle W_B^k supplied by the caller
template <int M>
__device__ __forceinline__ cplx rfft_harmonic_w(const cplx* lds, cplx w, int k) {
    constexpr int P = FftPlan<M>::PADLOG;
    const cplx zk = lds[lds_pad<P>(k & (M - 1))];
    cplx zc = lds[lds_pad<P>((M - k) & (M - 1))];
    zc.y = -zc.y;
    const cplx E = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
    const cplx O = make_double2(0.5 * (zk.x - zc.x), 0.5 * (zk.y - zc.y));
    const cplx wo = cmul(w, O);
    return make_double2(E.x + wo.y, E.y - wo.x);
}

}  // namespace pp
