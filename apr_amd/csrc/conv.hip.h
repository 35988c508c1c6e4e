// conv.hip.h -- K4/K5: the NoiseDetector CNN (ai/model.py:55-117) in fp32 on CDNA4.
//
//   conv1_kernel            Cin = 1 first layer: direct 3x3 on the VALU, NHWC output.
//   conv3x3_igemm_f32       layers 2..12: implicit GEMM on v_mfma_f32_32x32x2_f32
//                           (exact f32 fma chain), BatchNorm folded into weights/bias,
//                           bias + ReLU fused into the epilogue.
//   head_kernel             global max-pool + Linear + f64 softmax + the daemon's
//                           label/threshold rule (src/model/model.go:211-228,
//                           src/daemon/daemon.go:189-207).
//
// Activations are NHWC ([window][h][w][c]) so that the im2col K axis (tap, cin) is
// contiguous in memory; weights are stored [tap][cin][cout].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// conv1: spec [B][H][W] (C=1) -> out [B][H][W][COUT], 3x3 pad 1 stride 1, + bias, ReLU.
// 8 lanes per pixel, 4 channels per lane: each pixel's 32 channels are one 128-B store.
// w1t is [9][32] (tap-major), BN already folded.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1_kernel(const float *__restrict__ spec, const float *__restrict__ w1t,
                                                    const float *__restrict__ bias, float *__restrict__ out,
                                                    int H, int W, long long n_pixels)
{
    const int cq = threadIdx.x & 7;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; t++) w[t] = *reinterpret_cast<const float4 *>(w1t + t * 32 + cq * 4);
    const float4 b = *reinterpret_cast<const float4 *>(bias + cq * 4);
    const long long stride = (long long)gridDim.x * 32;
    for (long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); p < n_pixels; p += stride) {
        const int x = (int)(p % W);
        const long long t = p / W;
        const int y = (int)(t % H);
        const float *base = spec + (t - y) * W;   // start of this window's plane
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int iy = y + kh - 1;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int ix = x + kw - 1;
                const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? base[(long long)iy * W + ix] : 0.0f;
                const float4 ww = w[kh * 3 + kw];
                acc.x = fmaf(ww.x, v, acc.x); acc.y = fmaf(ww.y, v, acc.y);
                acc.z = fmaf(ww.z, v, acc.z); acc.w = fmaf(ww.w, v, acc.w);
            }
        }
        acc.x = fmaxf(acc.x + b.x, 0.f); acc.y = fmaxf(acc.y + b.y, 0.f);
        acc.z = fmaxf(acc.z + b.z, 0.f); acc.w = fmaxf(acc.w + b.w, 0.f);
        *reinterpret_cast<float4 *>(out + p * 32 + cq * 4) = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM 3x3 convolution, pad 1, stride STRIDE, NHWC, fp32 MFMA.
//   M = B*Ho*Wo output pixels, N = Cout, K = 9*Cin (k = tap*Cin + cin).
//   Block tile BM x BN, K step 32; 4 waves as 2(M) x 2(N); each wave owns
//   (BM/2)x(BN/2) = TM x TN tiles of 32x32 (16 accumulator VGPRs each).
//   LDS: A tile transposed [k][m] with row stride BM+1 (conflict-free for both the
//   4x ds_write_b32 staging and the one-dword-per-lane MFMA operand reads), B tile
//   [k][n]; double buffered, one barrier per K step; next tile's global loads are
//   issued before the MFMAs of the current one.
//   Requires Cin % 32 == 0 and Cout % BN == 0 (checked on the host).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int STRIDE>
__global__ __launch_bounds__(256, 2) void conv3x3_igemm_f32(const float *__restrict__ in,
                                                            const float *__restrict__ wt,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ out, int H, int W, int Cin,
                                                            int Ho, int Wo, int Cout, int M, int tiles_n,
                                                            const float *__restrict__ zeros /* >= 16 B of 0.0f */)
{
    constexpr int BK = 32;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AROWS = BM / 32;
    constexpr int BCH = BN / 4;
    constexpr int BROWS = 256 / BCH;
    constexpr int BPASS = BK / BROWS;
    constexpr int LDA = BM + 1;
    static_assert(TM >= 1 && TN >= 1 && BPASS >= 1, "tile too small");

    __shared__ float As[2][BK * LDA];
    __shared__ float Bs[2][BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, half = lane >> 5;

    // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs; give each XCD a
    // contiguous run of tiles so neighbours (same A rows / same weights) share an L2.
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bn = lid % tiles_n;
    const int bm = lid / tiles_n;

    // ---- A (im2col) load assignment: thread -> 16-B chunk ac of rows ar0 + 32 i ----
    const int ac = tid & 7, ar0 = tid >> 3;
    long long rowbase[AROWS];
    unsigned rowmask[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; i++) {
        const int p = bm * BM + ar0 + 32 * i;   // M < 2^31 (host-checked)
        rowmask[i] = 0;
        rowbase[i] = 0;
        if (p < M) {
            const int ow = p % Wo;
            const int t = p / Wo;
            const int oh = t % Ho;
            const long long b = t / Ho;
            const int ih0 = oh * STRIDE - 1, iw0 = ow * STRIDE - 1;
            rowbase[i] = ((b * H + ih0) * W + iw0) * (long long)Cin + ac * 4;
            unsigned mk = 0;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int ih = ih0 + tap / 3, iw = iw0 + tap % 3;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) mk |= 1u << tap;
            }
            rowmask[i] = mk;
        }
    }
    // ---- B (weights) load assignment ----
    const int bch = tid % BCH, brow0 = tid / BCH;
    const float *wcol = wt + (size_t)bn * BN + bch * 4;

    const int cchunks = Cin >> 5;
    const int KT = 9 * cchunks;

    // Loads are unconditional: out-of-image taps (zero padding) and rows past M read from a
    // small zero buffer instead, so the loaded registers go to LDS untouched.  (A branch around
    // a load makes hipcc wait for every load separately; a select on the loaded value lets it
    // hoist that select, and the vmcnt wait with it, right behind the load.)
    f32x4 ra[AROWS], rb[BPASS];
#define APR_LOAD_TILES(KT_)                                                                              \
    {                                                                                                    \
        const int tap_ = (KT_) / cchunks;                                                                \
        const int c0_ = ((KT_) - tap_ * cchunks) << 5;                                                   \
        const long long tapoff_ = ((long long)(tap_ / 3) * W + (tap_ % 3)) * Cin + c0_;                  \
        _Pragma("unroll") for (int i = 0; i < AROWS; i++)                                                \
        {                                                                                                \
            const bool ok_ = (rowmask[i] >> tap_) & 1u;                                                  \
            const float *src_ = ok_ ? in + (rowbase[i] + tapoff_) : zeros;                               \
            ra[i] = *reinterpret_cast<const f32x4 *>(src_);                                              \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < BPASS; i++)                                                \
            rb[i] = *reinterpret_cast<const f32x4 *>(wcol + (size_t)((KT_) * BK + brow0 + BROWS * i) * Cout); \
    }
#define APR_STORE_TILES(BUF_)                                                                            \
    {                                                                                                    \
        float *a_ = As[BUF_];                                                                            \
        _Pragma("unroll") for (int i = 0; i < AROWS; i++)                                                \
        {                                                                                                \
            const int rr_ = ar0 + 32 * i;                                                                \
            a_[(ac * 4 + 0) * LDA + rr_] = ra[i].x;                                                      \
            a_[(ac * 4 + 1) * LDA + rr_] = ra[i].y;                                                      \
            a_[(ac * 4 + 2) * LDA + rr_] = ra[i].z;                                                      \
            a_[(ac * 4 + 3) * LDA + rr_] = ra[i].w;                                                      \
        }                                                                                                \
        float *b_ = Bs[BUF_];                                                                            \
        _Pragma("unroll") for (int i = 0; i < BPASS; i++)                                                \
            *reinterpret_cast<f32x4 *>(b_ + (brow0 + BROWS * i) * BN + bch * 4) = rb[i];                \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    APR_LOAD_TILES(0)
    APR_STORE_TILES(0)
    __syncthreads();

    // Main loop.  Everything that is not an MFMA is placed in the shadow of the MFMAs: operand
    // fragments are read one k-pair ahead (two register sets), the next tile's global loads are
    // issued after the first k-pair and staged to the other LDS buffer late in the step (that
    // buffer was released by the barrier that ended the previous step).  hipcc would otherwise
    // sink every ds_read next to its MFMA (lgkmcnt(0) before each group), so LDS reads and
    // MFMAs are fenced per k-pair; only VALU/SALU may still move (mask 0x6).
#define APR_KSTEP(HAS_NEXT)                                                                              \
    {                                                                                                    \
        const float *a = As[cur] + wm * WM + li;                                                         \
        const float *b = Bs[cur] + wn * WN + li;                                                         \
        float af[2][TM], bf[2][TN];                                                                      \
        _Pragma("unroll") for (int i = 0; i < TM; i++) af[0][i] = a[half * LDA + i * 32];                \
        _Pragma("unroll") for (int j = 0; j < TN; j++) bf[0][j] = b[half * BN + j * 32];                 \
        _Pragma("unroll") for (int kk = 0; kk < BK / 2; kk++)                                            \
        {                                                                                                \
            if (kk + 1 < BK / 2) {                                                                       \
                const int k = 2 * (kk + 1) + half;                                                       \
                _Pragma("unroll") for (int i = 0; i < TM; i++) af[(kk + 1) & 1][i] = a[k * LDA + i * 32]; \
                _Pragma("unroll") for (int j = 0; j < TN; j++) bf[(kk + 1) & 1][j] = b[k * BN + j * 32]; \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0x6);                                                         \
            if (HAS_NEXT && kk == 1) APR_LOAD_TILES(kt + 1)                                              \
            if (HAS_NEXT && kk == BK / 2 - 4) APR_STORE_TILES(cur ^ 1)                                   \
            _Pragma("unroll") for (int i = 0; i < TM; i++)                                               \
                _Pragma("unroll") for (int j = 0; j < TN; j++)                                           \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0x6);                                                         \
        }                                                                                                \
        __syncthreads();                                                                                 \
        cur ^= 1;                                                                                        \
    }

    int cur = 0;
    int kt = 0;
    for (; kt < KT - 1; kt++) APR_KSTEP(true)
    APR_KSTEP(false)
#undef APR_KSTEP

    // ---- epilogue: bias + ReLU, NHWC store.  D[row][col]: col = lane&31,
    //      row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = bn * BN + wn * WN + j * 32 + li;
        const float bv = bias[n];
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int prow = bm * BM + wm * WM + i * 32 + 4 * half;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int p = prow + (e & 3) + 8 * (e >> 2);
                if (p < M) out[(long long)p * Cout + n] = fmaxf(acc[i][j][e] + bv, 0.0f);
            }
        }
    }
}

#undef APR_LOAD_TILES
#undef APR_STORE_TILES

// ---------------------------------------------------------------------------------------------
// head: act [B][P][F] (NHWC, P = 2*3 pixels, F = 1024) -> max over P -> logits = Wfc feat + b
//       -> f64 softmax -> label / match.  One workgroup per window.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ act, int P, int F,
                                                   const float *__restrict__ wfc /*[C][F]*/,
                                                   const float *__restrict__ bfc, int C, double trust,
                                                   int empty_class, float *__restrict__ logits /*[B][C] or null*/,
                                                   double *__restrict__ probs /*[B][C] or null*/,
                                                   int32_t *__restrict__ label, uint8_t *__restrict__ match)
{
    __shared__ float s_feat[1024];
    __shared__ float s_logit[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long win = blockIdx.x;
    const float *a = act + win * (long long)P * F;
    for (int c = tid; c < F; c += 256) {
        float m = a[c];
        for (int p = 1; p < P; p++) m = fmaxf(m, a[(long long)p * F + c]);
        s_feat[c] = m;
    }
    __syncthreads();
    // one wave per class (round-robin): 64-lane strided dot + shuffle reduce
    for (int o = wave; o < C; o += 4) {
        float s = 0.0f;
        for (int c = lane; c < F; c += 64) s = fmaf(wfc[(long long)o * F + c], s_feat[c], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) s_logit[o] = s + bfc[o];
    }
    __syncthreads();
    if (tid == 0) {
        // softmax, model.go:232-254 (f64, max-subtracted)
        double mx = -1.7976931348623157e308;
        for (int o = 0; o < C; o++) { const double v = (double)s_logit[o]; if (v > mx) mx = v; }
        double ex[64];
        double sum = 0.0;
        for (int o = 0; o < C; o++) { ex[o] = exp((double)s_logit[o] - mx); sum += ex[o]; }
        int best = -1;
        double bc = 0.0;
        for (int o = 0; o < C; o++) {
            const double pr = ex[o] / sum;
            if (probs) probs[win * C + o] = pr;
            if (pr > bc) { bc = pr; best = o; }   // daemon.go:193-198 (strict >, from ("", 0.0))
        }
        if (label) label[win] = best;
        if (match) match[win] = (best >= 0 && best != empty_class && bc > trust) ? 1 : 0;   // daemon.go:203
    }
    if (logits && tid < C) logits[win * C + tid] = s_logit[tid];
}

}  // namespace apr
