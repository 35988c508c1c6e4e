// report.hip.h -- one hit per class per minute over the labels of a sweep ("1 clap or 999 claps within 1 minute is
// logged as one hit", docs/usage/review.md:48-53; the reference's code does not implement it).  The step after the
// path: a segmented OR-reduce over the per-window (label, match) rows while they are still in HBM (the gathered rows
// of a multi-GPU sweep land there), so only [minutes][classes] bytes travel to the host.
// HBM-bound: 5 bytes in per window, n_classes bytes out per bucket.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apr {

// One wave per bucket of `wpb` windows: every lane ORs the classes of its matched windows into a 64-bit mask
// (n_classes <= 64 = APR_MAX_CLASSES), the wave ORs the masks, lane c writes hits[bucket][c].
__global__ __launch_bounds__(256) void bucket_hits_kernel(const int32_t *__restrict__ label, const uint8_t *__restrict__ match,
                                                          unsigned long long n_windows, int n_classes, unsigned long long wpb,
                                                          unsigned long long n_buckets, uint8_t *__restrict__ hits)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long n_waves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    for (unsigned long long b = wave; b < n_buckets; b += n_waves) {   // wave-uniform
        const unsigned long long k0 = b * wpb, k1 = k0 + wpb < n_windows ? k0 + wpb : n_windows;
        unsigned long long mask = 0;
        for (unsigned long long k = k0 + lane; k < k1; k += 64) {
            const int l = label[k];
            if (match[k] && l >= 0 && l < n_classes) mask |= 1ull << l;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mask |= __shfl_xor(mask, o);
        if (lane < n_classes) hits[b * (unsigned long long)n_classes + lane] = (uint8_t)((mask >> lane) & 1ull);
    }
}

// apr_mfma_probe: v_mfma_f32_32x32x2_f32 back to back from registers, eight independent accumulators a wave (data that is
// neither zero nor constant across lanes: the clock a chip holds depends on it).
__global__ __launch_bounds__(256) void mfma_probe_kernel(float *sink, int iters)
{
    typedef float pf16 __attribute__((ext_vector_type(16)));
    const int lane = threadIdx.x & 63;
    float a = 0.001f * (float)(lane + 1), b = 1.0f - 0.002f * (float)lane;
    pf16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a = -a;                                                // keeps the sums bounded
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) s += acc[i][j];
    if (s == 12345.678f) sink[0] = s;                          // never true: keeps the loop alive
}

}  // namespace apr
