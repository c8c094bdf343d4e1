// Standalone check of pp_fftq.h (the one-exchange 1024-point FFT) against a direct DFT.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pulseportraiture_amd/csrc -o variants/fftq_test tools/dev_fftq_test.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <complex>
#include "pp_fftq.h"
using namespace pp;

__global__ __launch_bounds__(64) void k_test(const cplx* in, cplx* out, const cplx* twB, double* pw, int nrows) {
    __shared__ cplx lds[FFTQ_LDS_ELEMS];
    const int tid = threadIdx.x;
    const cplx t1 = twB[2 * tid], t2 = twB[32 * (tid & 15)];
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        cplx v[16];
        for (int r = 0; r < 16; ++r) v[r] = in[(size_t)row * 1024 + tid + 64 * r];
        double p = 0.0;
        fftq1024(v, lds, t1, t2, tid, &p, NoMid());
        const int lam = fftq_lambda(tid);
        for (int kd = 0; kd < 16; ++kd) out[(size_t)row * 1024 + lam + 64 * kd] = v[kd];
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (tid == 0) pw[row] = p;
        if (fftq_lane_of(lam) != tid) out[0] = make_double2(1e300, 1e300);
    }
}

int main() {
    const int N = 1024, B = 2048, nrows = 8;
    std::vector<std::complex<double>> x(nrows * N), ref(nrows * N), got(nrows * N), tw(N + 1);
    srand(1);
    for (auto& z : x) z = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    for (int k = 0; k <= N; ++k) tw[k] = std::polar(1.0, -2.0 * M_PI * k / B);
    for (int r = 0; r < nrows; ++r)
        for (int k = 0; k < N; ++k) {
            std::complex<long double> s = 0;
            for (int n = 0; n < N; ++n) {
                const long double a = -2.0L * M_PIl * (long double)((long long)n * k % N) / N;
                s += std::complex<long double>(x[r * N + n]) * std::complex<long double>(cosl(a), sinl(a));
            }
            ref[r * N + k] = std::complex<double>((double)s.real(), (double)s.imag());
        }
    cplx *din, *dout, *dtw; double* dpw;
    hipMalloc(&din, sizeof(cplx) * nrows * N); hipMalloc(&dout, sizeof(cplx) * nrows * N);
    hipMalloc(&dtw, sizeof(cplx) * (N + 1)); hipMalloc(&dpw, 8 * nrows);
    hipMemcpy(din, x.data(), sizeof(cplx) * nrows * N, hipMemcpyHostToDevice);
    hipMemcpy(dtw, tw.data(), sizeof(cplx) * (N + 1), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(4), dim3(64), 0, 0, din, dout, dtw, dpw, nrows);
    hipMemcpy(got.data(), dout, sizeof(cplx) * nrows * N, hipMemcpyDeviceToHost);
    std::vector<double> pw(nrows);
    hipMemcpy(pw.data(), dpw, 8 * nrows, hipMemcpyDeviceToHost);
    double emax = 0, nmax = 0, perr = 0;
    for (int r = 0; r < nrows; ++r) {
        double p = 0;
        for (int k = 0; k < N; ++k) {
            emax = std::max(emax, std::abs(got[r * N + k] - ref[r * N + k]));
            nmax = std::max(nmax, std::abs(ref[r * N + k]));
            if (k) p += std::norm(ref[r * N + k]);
            else { const double d = ref[r * N].real() - ref[r * N].imag(); p += d * d; }
        }
        perr = std::max(perr, std::fabs(p - pw[r]) / p);
    }
    printf("fftq1024: max |err| %.3e (max |Z| %.3e), rel %.3e; power rel err %.3e  %s\n", emax, nmax, emax / nmax, perr,
           (emax / nmax < 1e-14 && perr < 1e-13) ? "OK" : "FAIL");
    return !(emax / nmax < 1e-14 && perr < 1e-13);
}
