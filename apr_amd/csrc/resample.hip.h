// resample.hip.h -- integer-factor polyphase FIR upsampler for s16 PCM (16 kHz sources -> the 48 kHz
// the scoring path expects).  In the reference this step is ffmpeg's `-ar 48000` inside the decode
// commands (src/ffmpeg/ffmpeg.go:76,113-114; ai/inspect.py:99-100); libswresample cannot be matched
// bit for bit, so the filter here is this library's own (apr_resample_taps) and the step lies outside
// the parity claim.  Arithmetic is f64, one rounded multiply and one rounded add per tap in k order
// (no fma contraction), round-half-even, saturate: the oracle's numpy loop gives the same bits.
//
//   y[up m + p] = sum_{k < tpp} taps[p][k] * x[m + tpp/2 - k]        (x outside [0, n_in) = 0)
//
// HBM-bound (2 B in, 2 up B out per input sample); taps sit in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apr {

constexpr int kResampleMaxTaps = 1024;   // up * taps_per_phase doubles in LDS (8 KiB)

// x points at global sample x_base; the caller guarantees that every sample index in
// [max(0, first_m + tpp/2 - tpp + 1), min(n_in, last_m + tpp/2 + 1)) is readable at x[i - x_base].
__global__ __launch_bounds__(256) void upsample_fir_kernel(const int16_t *__restrict__ x, long long x_base, long long n_in, int up,
                                                           int tpp, const double *__restrict__ taps, int16_t *__restrict__ y,
                                                           long long n_first, long long n_count)
{
    __shared__ double t[kResampleMaxTaps];
    for (int i = threadIdx.x; i < up * tpp; i += blockDim.x) t[i] = taps[i];
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_count; j += stride) {
        const long long n = n_first + j;
        const long long m = n / up;
        const int p = (int)(n - m * up);
        const double *tp = t + p * tpp;
        const long long i0 = m + tpp / 2;
        double acc = 0.0;
        for (int k = 0; k < tpp; k++) {
            const long long i = i0 - k;
            const double xv = (i >= 0 && i < n_in) ? (double)x[i - x_base] : 0.0;
            acc = __dadd_rn(acc, __dmul_rn(tp[k], xv));
        }
        double r = rint(acc);
        r = fmin(fmax(r, -32768.0), 32767.0);
        y[j] = (int16_t)r;
    }
}

}  // namespace apr
