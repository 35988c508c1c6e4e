// wino_fused_bench.cpp -- stand-alone check + timing of wino_fused_f2 against the direct igemm kernel
// (developer tool).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/wino_fused_bench.cpp -o tools/wino_fused_bench
//   tools/wino_fused_bench H W Cin Cout batch [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../apr_amd/csrc/wino_fused8.hip.h"

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 1; } } while (0)

__global__ void maxdiff_kernel(const float *a, const float *b, size_t n, unsigned *out_bits, float *out_maxabs)
{
    float m = 0.f, s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float df = fabsf(a[i] - b[i]);
        if (!(df <= 1e30f)) df = 1e30f;   // NaN (an output never written) must not hide
        m = fmaxf(m, df);
        s = fmaxf(s, fabsf(b[i]));
    }
    atomicMax(out_bits, __float_as_uint(m));
    atomicMax(reinterpret_cast<unsigned *>(out_maxabs), __float_as_uint(s));
}

int main(int argc, char **argv)
{
    if (argc < 6) { printf("usage: H W Cin Cout batch [reps]\n"); return 1; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), B = atoi(argv[5]);
    const int reps = argc > 6 ? atoi(argv[6]) : 10;
    if (Cin % 16 || Cout % 128) { printf("need Cin %% 16 == 0, Cout %% 128 == 0\n"); return 1; }
    const long long M = (long long)B * H * W;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = (long long)B * th * tw;
    const size_t n_in = (size_t)B * H * W * Cin, n_out = (size_t)M * Cout, n_w = (size_t)9 * Cin * Cout;
    std::vector<float> h_in(n_in), h_g((size_t)Cout * Cin * 9), h_w(n_w), h_b(Cout);
    unsigned s = 12345;
    for (auto &v : h_in) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.0f - 0.3f; }
    for (auto &v : h_g) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.05f; }
    for (auto &v : h_b) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    // direct layout [tap][cin][cout]
    for (int o = 0; o < Cout; o++)
        for (int c = 0; c < Cin; c++)
            for (int t = 0; t < 9; t++) h_w[((size_t)t * Cin + c) * Cout + o] = h_g[((size_t)o * Cin + c) * 9 + t];
    std::vector<float> h_u((size_t)16 * Cin * Cout);
    apr::wf_pack_u(h_g.data(), Cin, Cout, h_u.data());

    float *d_in, *d_w, *d_u, *d_b, *d_out, *d_ref, *d_stat;
    CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_u, h_u.size() * 4)); CK(hipMalloc(&d_b, Cout * 4));
    CK(hipMalloc(&d_out, n_out * 4)); CK(hipMalloc(&d_ref, n_out * 4)); CK(hipMalloc(&d_stat, 8));
    CK(hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_u, h_u.data(), h_u.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xFF, n_out * 4));
    CK(hipMemset(d_stat, 0, 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(apr::wino_fused_f2), hipFuncAttributeMaxDynamicSharedMemorySize, apr::WF_LDS_BYTES));

    // reference: the direct implicit-GEMM kernel
    {
        const int tiles_n = Cout / 128;
        const long long tiles_m = (M + 127) / 128;
        hipLaunchKernelGGL((apr::conv3x3_igemm_f32<128, 128, 1>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, 0, d_in, d_w, d_b, d_ref, H, W, Cin, H, W, Cout, (int)M, tiles_n);
    }
    const int tiles_n = Cout / apr::WF_BN;
    const long long tiles_m = (T + apr::WF_TT - 1) / apr::WF_TT;
    const bool k8 = getenv("WF8") != nullptr;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(apr::wino_fused8_f2), hipFuncAttributeMaxDynamicSharedMemorySize, apr::WF8_LDS_BYTES));
    int n_cus = 256;
    (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0);
    auto launch = [&]() {
        if (k8)
            hipLaunchKernelGGL(apr::wino_fused8_f2, dim3(apr::wf8_grid((int)(tiles_m * tiles_n), n_cus)), dim3(512), apr::WF8_LDS_BYTES, 0, d_in, d_u, d_b, d_out, H, W, Cin, Cout, th, tw, (int)T, tiles_n, apr::wf_magic(tiles_n, th, tw, tiles_m, getenv("WF_GN") ? atoi(getenv("WF_GN")) : 4), (int)(tiles_m * tiles_n));
        else
        hipLaunchKernelGGL(apr::wino_fused_f2, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), apr::WF_LDS_BYTES, 0, d_in, d_u, d_b, d_out, H, W, Cin, Cout, th, tw, (int)T, tiles_n, apr::wf_magic(tiles_n, th, tw, tiles_m));
    };
    launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(maxdiff_kernel, dim3(2048), dim3(256), 0, 0, d_out, d_ref, n_out, reinterpret_cast<unsigned *>(d_stat), d_stat + 1);
    float stat[2];
    CK(hipMemcpy(stat, d_stat, 8, hipMemcpyDeviceToHost));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = 2.0 * (double)M * Cout * 9.0 * Cin;
    const double flx = 2.0 * 16.0 * (double)T * Cin * Cout;
    printf("%dx%d Cin=%d Cout=%d B=%d: %.3f ms  effective %.1f TFLOP/s  executed %.1f TFLOP/s  blocks=%lld  max|diff|=%.3e (max|ref|=%.3e)\n", H, W, Cin,
           Cout, B, ms, fl / ms / 1e9, flx / ms / 1e9, tiles_m * tiles_n, stat[0], stat[1]);
#ifdef APR_WF_STAMPS
    {
        // per-block phase times (s_memtime ticks at 100 MHz on this chip? printed raw) and the gap between
        // consecutive blocks on the same CU (HW_ID: cu_id, sh_id, se_id; xcc via XCC_ID not read -> group by bid&7 too)
        static unsigned long long h[4096][8];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(apr::wf_stamps), sizeof h));
        const int nb = (int)std::min<long long>(4096, tiles_m * tiles_n);
        double ph[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < nb; b++) for (int i = 0; i < 5; i++) ph[i] += (double)(h[b][i + 1] - h[b][i]);
        printf("stamps (avg ticks over %d blocks): setup %.0f  first-loads+transform+barrier %.0f  K loop %.0f  epilogue issue %.0f  store drain %.0f\n", nb,
               ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb, ph[4] / nb);
        // gap: for every CU slot (xcd = b&7, hw_id), sort by start and take next.start - prev.end
        std::vector<std::pair<unsigned long long, int>> order;
        double gap = 0; int ng = 0;
        for (int b = 0; b < nb; b++) {
            unsigned long long best = ~0ull; int bi = -1;
            for (int c = 0; c < nb; c++)
                if (c != b && (c & 7) == (b & 7) && h[c][6] == h[b][6] && h[c][0] >= h[b][5] && h[c][0] < best) { best = h[c][0]; bi = c; }
            if (bi >= 0) { gap += (double)(best - h[b][5]); ng++; }
        }
        printf("gap between a block's end and the next block's first instruction on the same CU: %.0f ticks (%d pairs)\n", ng ? gap / ng : 0.0, ng);
        printf("block 0: %llu %llu %llu %llu %llu %llu hwid %llx\n", h[0][0], h[0][1] - h[0][0], h[0][2] - h[0][0], h[0][3] - h[0][0], h[0][4] - h[0][0], h[0][5] - h[0][0], h[0][6]);
    }
#endif
#ifdef W8_USTAMPS
    {
        unsigned long long h[8][10];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(apr::w8_ustamps), sizeof h));
        const double units = (double)(reps + 1) * ((tiles_m * tiles_n + apr::wf8_grid((int)(tiles_m * tiles_n), n_cus) - 1) / apr::wf8_grid((int)(tiles_m * tiles_n), n_cus));
        printf("cycles per unit of workgroup 0 (about %.0f units): next set-up+loads | epi 1 | barrier | epi 2+stores | barrier | prologue | barrier | K loop\n", units);
        for (int w = 0; w < 8; w++) {
            printf("  wave %d:", w);
            printf(" (set-up %5.0f)", (double)h[w][8] / units);
            for (int i = 0; i < 8; i++) printf(" %7.0f", (double)h[w][i] / units);
            printf("\n");
        }
        unsigned long long k[8][8];
        CK(hipMemcpyFromSymbol(k, HIP_SYMBOL(apr::w8_ksteps), sizeof k));
        printf("cycles per K step: first two | the loop's two (sum over its passes) | last four\n");
        for (int w = 0; w < 8; w++) {
            printf("  wave %d:", w);
            for (int i = 0; i < 8; i++) printf(" %7.0f", (double)k[w][i] / units);
            printf("\n");
        }
    }
#endif
#ifdef W8_TIMES
    {
        static unsigned long long h[1024][2];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(apr::w8_times), sizeof h));
        const int g = (int)apr::wf8_grid((int)(tiles_m * tiles_n), n_cus);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < g; b++) { t0 = std::min(t0, h[b][0]); t1 = std::max(t1, h[b][1]); }
        printf("last launch: %d workgroups, %.1f us from the first start to the last end (100 MHz ticks)\n", g, (double)(t1 - t0) / 100.0);
        for (int x = 0; x < 8; x++) {
            double s0 = 1e30, s1 = 0, e0 = 1e30, e1 = 0;
            for (int b = x; b < g; b += 8) {
                s0 = std::min(s0, (double)(h[b][0] - t0)); s1 = std::max(s1, (double)(h[b][0] - t0));
                e0 = std::min(e0, (double)(h[b][1] - t0)); e1 = std::max(e1, (double)(h[b][1] - t0));
            }
            printf("  XCD %d: starts %.1f .. %.1f us, ends %.1f .. %.1f us\n", x, s0 / 100, s1 / 100, e0 / 100, e1 / 100);
        }
    }
#endif
    if (argc > 7) return 0;   // timing-only ablation builds
    return std::isfinite(stat[0]) && stat[0] <= 1e-4f * fmaxf(stat[1], 1.0f) ? 0 : 2;
}
