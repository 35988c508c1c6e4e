// mel_mfma_bench.cpp -- the log-mel projection as a banded MFMA contraction, timed beside the sparse (CSR) form the front end uses
// (developer tool; VERDICT r2 item 6: north_star names an MFMA log-mel, DESIGN.md only priced it).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mel_mfma_bench.cpp -o tools/mel_mfma_bench && tools/mel_mfma_bench [frames] [reps]
//
// Stand-alone: power spectra P[frames][1025] f32 in HBM (synthetic) -> mel[frames][128], go-flavour lens (src/model/dsp.go:31-80: 130 HTK
// points, FFT-bin-quantised triangles, 1 892 non-zero weights).  Three kernels over the same tiles of 32 frames staged in LDS:
//   load   : stage the tile and write zeros (what the HBM reads cost alone)
//   csr    : the product's form -- a wave per frame, two bands per lane, k-ascending fmaf over each band's bins
//   mfma   : the 128 bands as four blocks of 32; block b covers the bins [lo_b, hi_b) any of its bands touches (64 / 112 / 264 / 632 bins: the
//            bands widen with frequency), cut into chunks of 8 bins = 4 v_mfma_f32_32x32x2_f32 each; the ~134 chunks of a 32-frame tile are
//            dealt to the four waves in equal runs (a wave's run may cross a block boundary: it then adds its accumulator into the
//            tile's [4][32][32] sum in LDS and starts the next block), the dense weights of a chunk are one 16-B load per lane from an
//            L2-resident table (137 KB: more than the registers or the LDS beside the tile hold), prefetched one chunk ahead.
// The result of both compute forms is checked against a host double-precision reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NB = 1025, NM = 128, TF = 32;      // bins, bands, frames per tile
constexpr int ROW = 1036;                        // LDS row of a frame: 1025 bins + zeros up to a multiple of 4, stride 1036 % 32 = 12: b128 reads of 8 rows apart spread
constexpr int MAXCH = 192;                       // chunks of 8 bins over the four band blocks

struct Lens {
    int n_chunks;
    const int *ch_block, *ch_bin;                // chunk c: its band block, its first bin (a multiple of 4)
    const f32x4 *wch;                            // [n_chunks][64 lanes]: lane (li, half), element j = W[band 32 b + li][bin + 4 half + j]
    const int *start, *len, *off;                // CSR
    const float *w;
};

__device__ __forceinline__ void stage_tile(const float *P, long long f0, long long n_frames, float *s)
{
    for (int i = threadIdx.x; i < TF * (ROW / 4); i += 256) {
        const int r = i / (ROW / 4), c4 = (i % (ROW / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (f0 + r < n_frames) {
            const float *src = P + (f0 + r) * NB;
#pragma unroll
            for (int q = 0; q < 4; q++) if (c4 + q < NB) v[q] = src[c4 + q];
        }
        *reinterpret_cast<f32x4 *>(s + r * ROW + c4) = v;
    }
}

template <int MODE>   // 0 load only, 1 CSR, 2 MFMA
__global__ __launch_bounds__(256) void mel_kernel(const float *__restrict__ P, long long n_frames, Lens L, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float s[];   // [32][ROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_tiles = (n_frames + TF - 1) / TF;
    float *red = s + TF * ROW;                                   // [4 blocks][32 frames][32 bands]: the tile's sums (MFMA form)
    const int per = (L.n_chunks + 3) / 4, c0 = wave * per, c1 = min(c0 + per, L.n_chunks);
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long f0 = t * TF;
        __syncthreads();
        stage_tile(P, f0, n_frames, s);
        __syncthreads();
        if (MODE == 0) {
            for (int i = threadIdx.x; i < TF * NM; i += 256)
                if (f0 + i / NM < n_frames) out[(f0 + i / NM) * NM + i % NM] = s[(i / NM) * ROW + (i % NM)] * 0.0f;
        } else if (MODE == 1) {
            const int mA = lane, mB = NM - 1 - lane;                 // short band + long band per lane, like the product
            for (int r = wave; r < TF; r += 4) {
                const float *p = s + r * ROW;
                float a = 0.f, b = 0.f;
                for (int k = 0; k < L.len[mA]; k++) a = fmaf(L.w[L.off[mA] + k], p[L.start[mA] + k], a);
                for (int k = 0; k < L.len[mB]; k++) b = fmaf(L.w[L.off[mB] + k], p[L.start[mB] + k], b);
                if (f0 + r < n_frames) { out[(f0 + r) * NM + mA] = a; out[(f0 + r) * NM + mB] = b; }
            }
        } else {
            for (int i = threadIdx.x; i < 4 * 32 * 32; i += 256) red[i] = 0.0f;
            __syncthreads();
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const float *arow = s + (lane & 31) * ROW + 4 * (lane >> 5);
            int blk = c0 < c1 ? L.ch_block[c0] : 0;
            f32x4 wv = c0 < c1 ? L.wch[(size_t)c0 * 64 + lane] : f32x4{0, 0, 0, 0};
            auto flush = [&](int b) {      // acc[e]: frame (e & 3) + 8 (e >> 2) + 4 (lane >> 5), band lane & 31
#pragma unroll
                for (int e = 0; e < 16; e++) atomicAdd(&red[(b * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)], acc[e]);
            };
            for (int c = c0; c < c1; c++) {
                const int b = __builtin_amdgcn_readfirstlane(L.ch_block[c]);
                if (b != blk) {
                    flush(blk);
                    blk = b;
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
                }
                const f32x4 w = wv;
                if (c + 1 < c1) wv = L.wch[(size_t)(c + 1) * 64 + lane];
                const f32x4 av = *reinterpret_cast<const f32x4 *>(arow + __builtin_amdgcn_readfirstlane(L.ch_bin[c]));
#pragma unroll
                for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], w[j], acc, 0, 0, 0);
            }
            if (c0 < c1) flush(blk);
            __syncthreads();
            for (int i = threadIdx.x; i < TF * NM; i += 256) {
                const int f = i / NM, m = i % NM;
                if (f0 + f < n_frames) out[(f0 + f) * NM + m] = red[((m >> 5) * 32 + f) * 32 + (m & 31)];
            }
        }
    }
}

int main(int argc, char **argv)
{
    const long long n_frames = argc > 1 ? atoll(argv[1]) : 4096LL * 184;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    // ---- the go-flavour lens (dsp.go:31-80) ----
    const double max_mel = 2595.0 * log10(1.0 + 24000.0 / 700.0);
    int pts[NM + 2];
    for (int i = 0; i < NM + 2; i++) {
        const double mel = max_mel * i / (NM + 1), hz = 700.0 * (pow(10.0, mel / 2595.0) - 1.0);
        pts[i] = (int)floor((2048 + 1) * hz / 48000.0);
    }
    std::vector<float> dense((size_t)NM * NB, 0.f);
    std::vector<int> start(NM), len(NM), off(NM);
    std::vector<float> w;
    for (int m = 0; m < NM; m++) {
        const int s0 = pts[m], c = pts[m + 1], e = pts[m + 2];
        for (int k = s0; k < c; k++) dense[(size_t)m * NB + k] = (float)((double)(k - s0) / (double)(c - s0));
        for (int k = c; k < e; k++) dense[(size_t)m * NB + k] = (float)((double)(e - k) / (double)(e - c));
        start[m] = s0; len[m] = e - s0; off[m] = (int)w.size();
        for (int k = s0; k < e; k++) w.push_back(dense[(size_t)m * NB + k]);
    }
    size_t nnz = 0;
    for (float v : dense) nnz += v != 0.f;
    Lens L;
    std::vector<int> ch_block, ch_bin;
    std::vector<float> wch;
    int kbs[4], ktot = 0;
    for (int b = 0; b < 4; b++) {
        int lo = NB, hi = 0;
        for (int m = 32 * b; m < 32 * b + 32; m++) { lo = std::min(lo, start[m]); hi = std::max(hi, start[m] + len[m]); }
        lo &= ~3;
        const int kb = ((hi - lo) + 7) & ~7;
        kbs[b] = kb; ktot += kb;
        for (int k = 0; k < kb; k += 8) {
            ch_block.push_back(b); ch_bin.push_back(lo + k);
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 4; j++) {
                    const int bin = lo + k + 4 * (lane >> 5) + j;
                    wch.push_back(bin < NB ? dense[(size_t)(32 * b + (lane & 31)) * NB + bin] : 0.f);
                }
        }
    }
    L.n_chunks = (int)ch_block.size();
    if (L.n_chunks > MAXCH) { printf("too many chunks\n"); return 1; }
    printf("lens: %zu non-zero weights of %d (%.1f %%); band blocks cover %d + %d + %d + %d = %d bins = %d chunks (%.1f %% dense inside the blocks)\n", nnz, NM * NB,
           100.0 * nnz / (NM * NB), kbs[0], kbs[1], kbs[2], kbs[3], ktot, L.n_chunks, 100.0 * nnz / (32.0 * ktot));
    // ---- data ----
    std::vector<float> hP((size_t)n_frames * NB);
    unsigned sd = 1234567u;
    for (auto &v : hP) { sd = sd * 1664525u + 1013904223u; v = (float)(sd >> 8) / 16777216.0f; }
    float *dP, *dO, *dwd, *dw;
    int *dstart, *dlen, *doff, *dcb, *dcn;
    CK(hipMalloc(&dP, hP.size() * 4)); CK(hipMalloc(&dO, (size_t)n_frames * NM * 4));
    CK(hipMalloc(&dwd, wch.size() * 4)); CK(hipMalloc(&dw, w.size() * 4));
    CK(hipMalloc(&dcb, ch_block.size() * 4)); CK(hipMalloc(&dcn, ch_bin.size() * 4));
    CK(hipMemcpy(dcb, ch_block.data(), ch_block.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcn, ch_bin.data(), ch_bin.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dstart, NM * 4)); CK(hipMalloc(&dlen, NM * 4)); CK(hipMalloc(&doff, NM * 4));
    CK(hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwd, wch.data(), wch.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dstart, start.data(), NM * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlen, len.data(), NM * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(doff, off.data(), NM * 4, hipMemcpyHostToDevice));
    L.wch = reinterpret_cast<const f32x4 *>(dwd); L.ch_block = dcb; L.ch_bin = dcn; L.start = dstart; L.len = dlen; L.off = doff; L.w = dw;
    const size_t lds = (size_t)TF * ROW * 4 + 4 * 32 * 32 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(mel_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long n_tiles = (n_frames + TF - 1) / TF;
    const unsigned grid = (unsigned)std::min<long long>(n_tiles, 256);     // one 146-KiB workgroup per CU, persistent
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[3] = {"load only", "csr fmaf", "banded mfma"};
    double ms[3];
    std::vector<float> got((size_t)std::min<long long>(n_frames, 64) * NM);
    for (int mode = 0; mode < 3; mode++) {
        for (int r = 0; r < reps + 1; r++) {
            if (r == 1) CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(mel_kernel<0>, dim3(grid), dim3(256), lds, 0, dP, n_frames, L, dO);
            if (mode == 1) hipLaunchKernelGGL(mel_kernel<1>, dim3(grid), dim3(256), lds, 0, dP, n_frames, L, dO);
            if (mode == 2) hipLaunchKernelGGL(mel_kernel<2>, dim3(grid), dim3(256), lds, 0, dP, n_frames, L, dO);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        ms[mode] = t / reps;
        if (mode) {
            CK(hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (size_t f = 0; f < got.size() / NM; f++)
                for (int m = 0; m < NM; m++) {
                    double ref = 0;
                    for (int k = start[m]; k < start[m] + len[m]; k++) ref += (double)dense[(size_t)m * NB + k] * hP[f * NB + k];
                    worst = std::max(worst, fabs(ref - got[f * NM + m]) / std::max(1.0, fabs(ref)));
                }
            printf("%-12s %8.3f ms per %lld frames (%.2f us per 1024 windows of 184), max rel err vs f64 %.2e\n", names[mode], ms[mode], n_frames, 1e3 * ms[mode] * (1024.0 * 184) / n_frames, worst);
        } else {
            printf("%-12s %8.3f ms per %lld frames: %.2f TB/s of P\n", names[mode], ms[mode], n_frames, (double)n_frames * NB * 4 / (ms[mode] * 1e-3) / 1e12);
        }
    }
    printf("compute beyond the load: csr %.3f ms, mfma %.3f ms\n", ms[1] - ms[0], ms[2] - ms[0]);
    return 0;
}
