// igemm_bench.cpp -- stand-alone timing of one conv3x3_igemm_f32 layer shape (developer tool).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/igemm_bench.cpp -o tools/igemm_bench
//   tools/igemm_bench H W Cin Cout stride batch [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../apr_amd/csrc/conv.hip.h"

#ifndef BM_
#define BM_ 128
#endif
#ifndef BN_
#define BN_ 128
#endif

#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); return 1; } } while (0)

template <int S>
void launch(const float *in, const float *w, const float *b, float *out, int H, int W, int Cin, int Ho, int Wo, int Cout, int M, const float *z)
{
    const int tiles_n = Cout / BN_;
    const int tiles_m = (M + BM_ - 1) / BM_;
    hipLaunchKernelGGL((apr::conv3x3_igemm_f32<BM_, BN_, S>), dim3(tiles_m * tiles_n), dim3(256), 0, 0, in, w, b, out, H, W, Cin, Ho, Wo, Cout, M, tiles_n);
}

int main(int argc, char **argv)
{
    if (argc < 7) { printf("usage: H W Cin Cout stride batch [reps]\n"); return 1; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), S = atoi(argv[5]), B = atoi(argv[6]);
    const int reps = argc > 7 ? atoi(argv[7]) : 10;
    const int Ho = (H + 2 - 3) / S + 1, Wo = (W + 2 - 3) / S + 1;
    const long long M = (long long)B * Ho * Wo;
    const size_t n_in = (size_t)B * H * W * Cin, n_out = (size_t)M * Cout, n_w = (size_t)9 * Cin * Cout;
    std::vector<float> h_in(n_in), h_w(n_w), h_b(Cout, 0.1f);
    unsigned s = 12345;
    for (auto &v : h_in) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.0f - 0.5f; }
    for (auto &v : h_w) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.05f; }
    float *d_in, *d_w, *d_b, *d_out, *d_z;
    CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_b, Cout * 4)); CK(hipMalloc(&d_out, n_out * 4)); CK(hipMalloc(&d_z, 256));
    CK(hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_z, 0, 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) { if (S == 1) launch<1>(d_in, d_w, d_b, d_out, H, W, Cin, Ho, Wo, Cout, (int)M, d_z); else launch<2>(d_in, d_w, d_b, d_out, H, W, Cin, Ho, Wo, Cout, (int)M, d_z); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) { if (S == 1) launch<1>(d_in, d_w, d_b, d_out, H, W, Cin, Ho, Wo, Cout, (int)M, d_z); else launch<2>(d_in, d_w, d_b, d_out, H, W, Cin, Ho, Wo, Cout, (int)M, d_z); }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = 2.0 * (double)M * Cout * 9.0 * Cin;
    // checksum so variants can be compared for (in)equality
    std::vector<float> h_out(std::min<size_t>(n_out, 1 << 20));
    CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : h_out) cs += v;
    printf("%dx%d Cin=%d Cout=%d s=%d B=%d tile=%dx%d: %.3f ms  %.1f TFLOP/s  blocks=%lld  checksum=%.6e\n", H, W, Cin, Cout, S, B, BM_, BN_, ms, fl / ms / 1e9,
           ((M + BM_ - 1) / BM_) * (long long)(Cout / BN_), cs);
    return 0;
}
