// conv12_bench.cpp -- conv12_fused_kernel against conv1_kernel + conv3x3_igemm_f32<128,64,2,0>: bit equality and timing
// (developer tool).  hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/conv12_bench.cpp -o tools/conv12_bench
//   tools/conv12_bench [batch] [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../apr_amd/csrc/conv12_fused.hip.h"

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 5;
    const int H1 = 128, W1 = 188, Ho = 64, Wo = 94;
    const size_t n_spec = (size_t)B * H1 * W1, n_l1 = n_spec * 32, n_out = (size_t)B * Ho * Wo * 64;
    std::vector<float> h_spec(n_spec), h_w1(9 * 32), h_b1(32), h_w2(9 * 32 * 64), h_b2(64);
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto &v : h_spec) v = rnd();
    for (auto &v : h_w1) v = (rnd() - 0.5f) * 0.8f;
    for (auto &v : h_b1) v = (rnd() - 0.5f) * 0.2f;
    for (auto &v : h_w2) v = (rnd() - 0.5f) * 0.15f;
    for (auto &v : h_b2) v = (rnd() - 0.5f) * 0.2f;
    float *d_spec, *d_w1, *d_b1, *d_w2, *d_b2, *d_l1, *d_ref, *d_out;
    CK(hipMalloc(&d_spec, n_spec * 4)); CK(hipMalloc(&d_w1, h_w1.size() * 4)); CK(hipMalloc(&d_b1, 128)); CK(hipMalloc(&d_w2, h_w2.size() * 4));
    CK(hipMalloc(&d_b2, 256)); CK(hipMalloc(&d_l1, n_l1 * 4)); CK(hipMalloc(&d_ref, n_out * 4)); CK(hipMalloc(&d_out, n_out * 4));
    CK(hipMemcpy(d_spec, h_spec.data(), n_spec * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w1, h_w1.data(), h_w1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b1, h_b1.data(), 128, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w2, h_w2.data(), h_w2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b2, h_b2.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xFF, n_out * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(apr::conv12_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, apr::C12_LDS_BYTES));
    const long long M1 = (long long)B * H1 * W1, M2 = (long long)B * Ho * Wo;
    auto two = [&]() {
        const unsigned xchunks = (W1 + 3) / 4, n_rows = (unsigned)(M1 / W1);
        hipLaunchKernelGGL(apr::conv1_kernel, dim3((unsigned)std::min<long long>(((long long)n_rows * xchunks + 31) / 32, 256 * 32)), dim3(256), 0, 0, d_spec, d_w1,
                           d_b1, d_l1, H1, W1, n_rows, xchunks);
        hipLaunchKernelGGL((apr::conv3x3_igemm_f32<128, 64, 2, 0>), dim3((unsigned)((M2 + 127) / 128)), dim3(256), 0, 0, d_l1, d_w2, d_b2, d_ref, H1, W1, 32, Ho,
                           Wo, 64, (int)M2, 1);
    };
    const int tiles_x = (Wo + apr::C12_TW - 1) / apr::C12_TW, tiles_y = (Ho + apr::C12_TH - 1) / apr::C12_TH;
    const unsigned grid_cap = getenv("C12_GRID") ? (unsigned)atoi(getenv("C12_GRID")) : 512u;
    auto one = [&]() {
        const unsigned n_tiles = (unsigned)(B * tiles_x * tiles_y);
        hipLaunchKernelGGL(apr::conv12_fused_kernel, dim3(std::min(n_tiles, grid_cap)), dim3(256), apr::C12_LDS_BYTES, 0, d_spec, d_w1, d_b1, d_w2, d_b2,
                           d_out, H1, W1, Ho, Wo, tiles_x, tiles_y, n_tiles);
    };
    two(); one();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<float> a(n_out), r(n_out);
    CK(hipMemcpy(a.data(), d_out, n_out * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), d_ref, n_out * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double mx = 0;
    for (size_t i = 0; i < n_out; i++) {
        if (memcmp(&a[i], &r[i], 4)) { if (bad < 5) printf("  diff at %zu: %g vs %g\n", i, a[i], r[i]); bad++; }
        double d = fabs((double)a[i] - r[i]); if (!(d <= 1e30)) d = 1e30; if (d > mx) mx = d;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms2 = 0, ms1 = 0;
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; i++) two(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; i++) one(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
    const double fl = 2.0 * (double)M2 * 64 * 288;
    printf("B=%d: two kernels %.3f ms, fused %.3f ms (%.1f TFLOP/s of layer-2 MFMA work); %zu of %zu outputs differ in bits, max|diff| %.3e\n", B, ms2 / reps,
           ms1 / reps, fl / (ms1 / reps) / 1e9, bad, n_out, mx);
#ifdef C12_STAMPS
    {
        static unsigned long long h[1024][8];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(apr::c12_stamps), sizeof h));
        const unsigned nb = std::min(std::min((unsigned)(B * tiles_x * tiles_y), grid_cap), 1024u);
        double ph[6] = {0, 0, 0, 0, 0, 0};
        for (unsigned b = 0; b < nb; b++) for (int i = 0; i < 6; i++) ph[i] += (double)h[b][i];
        const double tiles_per_block = (double)(B * tiles_x * tiles_y) / nb;
        printf("cycles per tile (wave 0): sp->LDS + decode %.0f, barrier A %.0f, phase 1 %.0f, barrier B %.0f, phase 2 %.0f, epilogue %.0f\n", ph[0] / nb / tiles_per_block,
               ph[1] / nb / tiles_per_block, ph[2] / nb / tiles_per_block, ph[3] / nb / tiles_per_block, ph[4] / nb / tiles_per_block, ph[5] / nb / tiles_per_block);
    }
#endif
    return bad ? 2 : 0;
}
