// Which SIMD does wave w of a workgroup land on?  (s_getreg HW_ID: bits [5:4] = simd_id, [11:8] = cu_id on gfx9)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out)
{
    extern __shared__ float big[];   // large dynamic LDS: one workgroup per CU, as the fused kernels
    if ((threadIdx.x & 63) == 0) {
        unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID, 32 bits
        out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
    }
    if (big[threadIdx.x] == 12345.f) out[0] = 0;
}
int main()
{
    unsigned *d, h[16 * 8];
    hipMalloc(&d, sizeof h);
    hipMemset(d, 0, sizeof h);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
    for (int waves : {8, 4}) {
        hipLaunchKernelGGL(probe, dim3(8), dim3(64 * waves), 120000, 0, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int b = 0; b < 8; b++) {
            printf("waves/wg %d block %d:", waves, b);
            for (int w = 0; w < waves; w++) printf(" w%d->simd%u(cu%u,wave_slot%u)", w, (h[b * 16 + w] >> 4) & 3, (h[b * 16 + w] >> 8) & 15, h[b * 16 + w] & 15);
            printf("\n");
        }
    }
    return 0;
}
