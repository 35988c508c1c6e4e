// What does a run of MFMAs into ONE accumulator cost against the same MFMAs rotated over three?  (wino4_fused_out issues four in a row into
// acc[g % 3]; stamps put its first-served wave at 75 cycles an MFMA against the 64 of the pipe.)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/mfma_dep tools/probe/mfma_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int PAT>
__global__ __launch_bounds__(512) void probe(float *sink, unsigned long long *cyc, int iters)
{
    extern __shared__ float big[];   // large dynamic LDS: one workgroup per CU
    const int lane = threadIdx.x & 63;
    float a = 0.001f * (float)(lane + 1), b = 1.0f - 0.002f * (float)lane;
    f16v acc[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 6; g++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int k = PAT == 0 ? (4 * g + q) % 3 : PAT == 1 ? g % 3 : (q & 1 ? (g + 1) % 3 : g % 3);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        a = -a;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    if (s == 12345.678f || big[threadIdx.x] == 12345.f) sink[0] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int PAT>
static void run(const char *name, int waves, float *sink, unsigned long long *cyc)
{
    const int iters = 2000;
    hipFuncSetAttribute((const void *)probe<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(64 * waves), 120000, 0, sink, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(64 * waves), 120000, 0, sink, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double mf = 24.0 * iters;
    printf("%-28s waves/wg %d: %.3f ms, %.1f TFLOP/s; s_memtime ticks per MFMA of wave 0: %.2f, wave %d: %.2f\n", name, waves, ms,
           256.0 * waves * mf * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12, (double)h[0] / mf, waves - 1, (double)h[waves - 1] / mf);
}

int main()
{
    float *sink;
    unsigned long long *cyc;
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, 64);
    for (int waves : {4, 8}) {
        run<0>("rotated over three", waves, sink, cyc);
        run<1>("four in a row", waves, sink, cyc);
        run<2>("alternating two", waves, sink, cyc);
    }
    return 0;
}
