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

// developer knobs for tools/igemm_bench.cpp (timing-only ablations; both must be 1 in the product)
#ifndef APR_EXP_STAGE
#define APR_EXP_STAGE 1
#endif
#ifndef APR_EXP_LOAD
#define APR_EXP_LOAD 1
#endif
#ifndef APR_EXP_STORE
#define APR_EXP_STORE 1
#endif
#ifndef APR_EXP_ST0
#define APR_EXP_ST0 3
#endif
#ifndef APR_EXP_BDMA
#define APR_EXP_BDMA 1
#endif
#ifndef APR_EXP_BK
#define APR_EXP_BK 16
#endif
#ifndef APR_EXP_SYNC
#define APR_EXP_SYNC 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// conv1: spec [B][H][W] (C=1) -> out [B][H][W][COUT], 3x3 pad 1 stride 1, + bias, ReLU.
// 8 lanes per pixel, 4 channels per lane: each pixel's 32 channels are one 128-B store.  One
// work item = 4 consecutive pixels of a row (18 loads and one 32-bit division per 4 pixels; the
// kernel is bound by its 3 MB/window of stores).  w1t is [9][32] (tap-major), BN already folded.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1_kernel(const float *__restrict__ spec, const float *__restrict__ w1t,
                                                    const float *__restrict__ bias, float *__restrict__ out,
                                                    int H, int W, unsigned n_rows, unsigned xchunks)
{
    const int cq = threadIdx.x & 7;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; t++) w[t] = *reinterpret_cast<const float4 *>(w1t + t * 32 + cq * 4);
    const float4 b = *reinterpret_cast<const float4 *>(bias + cq * 4);
    const unsigned n_items = n_rows * xchunks;
    for (unsigned i = blockIdx.x * 32 + (threadIdx.x >> 3); i < n_items; i += gridDim.x * 32) {
        const unsigned r = i / xchunks;            // row = window * H + y
        const int x0 = (int)(i - r * xchunks) * 4;
        const int y = (int)(r % (unsigned)H);
        const float *row = spec + (size_t)r * W;
        float v[3][6];
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int iy = y + kh - 1;
            const bool rok = iy >= 0 && iy < H;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int ix = x0 + j - 1;
                v[kh][j] = (rok && ix >= 0 && ix < W) ? row[(kh - 1) * W + ix] : 0.0f;
            }
        }
#pragma unroll
        for (int px = 0; px < 4; px++) {
            if (x0 + px >= W) break;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const float u = v[kh][px + kw];
                    const float4 ww = w[kh * 3 + kw];
                    acc.x = fmaf(ww.x, u, acc.x); acc.y = fmaf(ww.y, u, acc.y);
                    acc.z = fmaf(ww.z, u, acc.z); acc.w = fmaf(ww.w, u, acc.w);
                }
            acc.x = fmaxf(acc.x + b.x, 0.f); acc.y = fmaxf(acc.y + b.y, 0.f);
            acc.z = fmaxf(acc.z + b.z, 0.f); acc.w = fmaxf(acc.w + b.w, 0.f);
            __builtin_nontemporal_store(f32x4{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<f32x4 *>(out + ((size_t)r * W + x0 + px) * 32 + cq * 4));
        }
    }
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA: 64 lanes x 16 B from a buffer (descriptor rs, per-lane byte offset voff, scalar byte
// offset soff) to 1 KiB of LDS starting at byte address lds_base (wave-uniform).  Inline asm on
// purpose: hipcc makes every later ds_read wait vmcnt(0) behind a builtin LDS-DMA (it cannot
// tell the DMA's target from the buffer being read), which would serialise the whole K step.
// The caller waits (s_waitcnt vmcnt(0)) before the barrier that publishes the tile.  hipcc's
// own counted vmcnt waits do not see these loads and therefore only ever wait for MORE.
__device__ __forceinline__ void lds_dma_b128(i32x4 rs, unsigned voff, int soff, unsigned lds_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(lds_base), "s"(soff)
                 : "memory");
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
//   Requires Cin % 16 == 0 and Cout % BN == 0 (checked on the host).
//   MODE 1 runs the same pipeline as a batch of plain GEMMs (the Winograd domain):
//   blockIdx.y = transform position pos, out[pos][M][Cout] = in[pos][M][Cin] x wt[pos][Cin][Cout],
//   one "tap", no padding, raw stores (bias/ReLU belong to the output transform).
//   MODE 2 is the convolution split along K for grids that are too small to fill the chip (a
//   single clip: layer 12 has 8 tiles of 576 K steps): blockIdx.y = split s takes K steps
//   [KT s / S, KT (s+1) / S) and stores its raw partial sums to out[s][M][Cout];
//   splitk_reduce_kernel adds them in split order (deterministic) with bias + ReLU.
//   MODE 3 is the stride-2 convolution in its polyphase Winograd F(2,2) form (wino_s2.hip.h): 25 products
//   [T tiles x Cin] x [Cin x Cout], one per position p = (pr, pc), whose A operand is the input transform
//   B^T d B of the tile's 5x5 patch, formed in this kernel's loader: at most two patch rows times two patch
//   columns, signs +-1 (uniform per K step), out-of-image pixels read as 0.  The output transform only ever
//   needs m0 + m3, m1 and m2 + m4 per axis, so the positions {0,3} and {2,4} of an axis are concatenated along
//   K: blockIdx.y selects one of 3 x 3 sums q = (qi, qj), whose K loop walks its 4, 2 or 1 positions (heaviest
//   sums first) -- 9 result tensors instead of 25.  H, W = the input image, Ho, Wo = the tile grid (th, tw),
//   M = T.  wt = the 16 distinct weight matrices U[4 min(pr,3) + min(pc,3)][Cin][Cout]; raw stores to
//   out[q][T][Cout] (wino_s2_output finishes A^T m A, + bias, ReLU).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int STRIDE, int MODE = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_igemm_f32(const float *__restrict__ in,
                                                            const float *__restrict__ wt,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ out, int H, int W, int Cin,
                                                            int Ho, int Wo, int Cout, int M, int tiles_n)
{
    constexpr int BK = APR_EXP_BK;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ACH = BK / 4;          // 16-B chunks per A row (k direction)
    constexpr int AR = 256 / ACH;        // A rows covered per pass of the 256 threads
    constexpr int AROWS = BM / AR;
    constexpr int BCH = BN / 4;
    constexpr int BROWS = 256 / BCH;
    constexpr int BPASS = BK / BROWS;
    constexpr int LDA = BM + 1;
    constexpr int NK = BK / 2;           // k-pairs (one v_mfma_f32_32x32x2 each) per K step
    constexpr int ST0 = APR_EXP_ST0;     // first k-pair that stages a piece of the next tile
    static_assert(TM >= 1 && TN >= 1 && BPASS >= 1, "tile too small");
    static_assert(ST0 + AROWS <= NK && ST0 + BPASS <= NK, "staging does not fit in the K step");

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

    // ---- addressing: buffer loads (wave-uniform descriptor + scalar offset per K step + a
    //      per-lane 32-bit offset fixed for the whole kernel), so a K step costs ~3 VALU per
    //      load instead of 64-bit pointer arithmetic.  Zero padding and rows past M use an
    //      out-of-range offset: the buffer unit returns 0 for those lanes.
    constexpr unsigned OOB = 0x80000000u;            // >= num_records of the A descriptor
    constexpr int TAPS = (MODE == 1 || MODE == 3) ? 1 : 9;
    const int p0 = bm * BM;                          // first output pixel (MODE 3: tile) of this block (< M)
    const long long b0 = MODE != 1 ? (p0 / Wo) / Ho : 0;   // its window
    // MODE 3: this block's sum q = (qi, qj): axis index 0 = positions {0, 3}, 1 = {1}, 2 = {2, 4}
    int s3qi = 1, s3qj = 1, s3sh = 0, s3np = 1;
    int st_sp = 0, st_c0 = 0;   // MODE 3: position (within the block's sum) and first channel of the tile being staged
    int st_r0 = 0, st_r1 = -1, st_cs0 = 0, st_cs1 = -1, st_tap = 0;   // ... and that position's terms and weight matrix
    if (MODE == 3) {
        const int q = (int)((0x471356280ull >> (4 * blockIdx.y)) & 15);   // blockIdx.y -> q: 0,8,2,6 (K = 4 Cin), 5,3,1,7 (2 Cin), 4 (Cin)
        s3qi = q / 3;
        s3qj = q - 3 * s3qi;
        s3sh = s3qj == 1 ? 0 : 1;   // two column positions: position index = 2 ri + ci
        s3np = (s3qi == 1 ? 1 : 2) << s3sh;
        out += (size_t)q * M * Cout;
    }
    if (MODE == 1) {                                 // batched GEMM: this block's problem
        in += (size_t)blockIdx.y * M * Cin;
        wt += (size_t)blockIdx.y * Cin * Cout;
        out += ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * M * Cout;   // gridDim.z > 1: this K split's copy of M[16][T][Cout]
    }
    if (MODE == 2) out += (size_t)blockIdx.y * M * Cout;   // this split's partial sums
    // descriptor base = one row and one column before window b0's image, so every valid
    // tap has a non-negative offset (the base itself may lie before `in`; never dereferenced);
    // MODE 1: the block's first row
    const float *abase = MODE != 1 ? in + ((b0 * H - 1) * W - 1) * (long long)Cin : in + (long long)p0 * Cin;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)abase, 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(wt + (size_t)bn * BN), 0, 0x7FFFFFFF, 0x00020000);

    const int ac = tid % ACH, ar0 = tid / ACH;       // A: 16-B chunk ac of rows ar0 + AR i
    unsigned rowoff[AROWS], rowmask[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; i++) {
        const int p = p0 + ar0 + AR * i;             // M < 2^31 (host-checked)
        rowmask[i] = 0;
        rowoff[i] = OOB;
        if (MODE == 1) {
            if (p < M) { rowoff[i] = 4u * (unsigned)((ar0 + AR * i) * Cin + ac * 4); rowmask[i] = 1u; }
        } else if (MODE == 3) {
            if (p < M) {   // tile p = (window, ty, tx): patch rows 4 ty - 1 .. 4 ty + 3, columns 4 tx - 1 .. 4 tx + 3
                const int tx = p % Wo;
                const int t = p / Wo;
                const int ty = t % Ho;
                const int db = t / Ho - (int)b0;
                rowoff[i] = 4u * (unsigned)(((db * H + 4 * ty) * W + 4 * tx) * Cin + ac * 4);   // from abase (one row, one column early)
                unsigned rok = 0, cok = 0;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int ih = 4 * ty - 1 + j, iw = 4 * tx - 1 + j;
                    if (ih >= 0 && ih < H) rok |= 1u << j;
                    if (iw >= 0 && iw < W) cok |= 1u << j;
                }
                rowmask[i] = rok | (cok << 8);
            }
        } else if (p < M) {
            const int ow = p % Wo;
            const int t = p / Wo;
            const int oh = t % Ho;
            const int db = t / Ho - (int)b0;         // windows past b0 (small)
            const int ih0 = oh * STRIDE - 1, iw0 = ow * STRIDE - 1;
            // element offset from abase: ((db*H + ih0 + 1)*W + iw0 + 1)*Cin + 4 ac   (< 2^28)
            rowoff[i] = 4u * (unsigned)(((db * H + ih0 + 1) * W + iw0 + 1) * Cin + ac * 4);
            unsigned mk = 0;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int ih = ih0 + tap / 3, iw = iw0 + tap % 3;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) mk |= 1u << tap;
            }
            rowmask[i] = mk;
        }
    }
    const int bch = tid % BCH, brow0 = tid / BCH;    // B: 16-B chunk bch of k-rows brow0 + BROWS i
    unsigned boff[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; i++) boff[i] = 4u * (unsigned)((brow0 + BROWS * i) * Cout + bch * 4);
    // the same weight descriptor as plain SGPR words for the inline-asm LDS-DMA, and the LDS byte
    // address of this wave's first k-rows of Bs[0] (low 32 bits of the flat LDS address)
    const unsigned long long wb = (unsigned long long)(wt + (size_t)bn * BN);
    i32x4 rsBv;
    rsBv.x = __builtin_amdgcn_readfirstlane((int)(unsigned)wb);
    rsBv.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(wb >> 32) & 0xFFFF);
    rsBv.z = 0x7FFFFFFF;
    rsBv.w = 0x00020000;
    const unsigned bs_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(&Bs[0][0] + wave * (64 / BCH) * BN));

    const int cchunks = Cin / BK;
    const int KT = (MODE == 3 ? s3np : TAPS) * cchunks;
    // K steps of this block: all of them, or this split's share (gridDim.y <= KT, host-checked)
    // (MODE 1: the split is blockIdx.z, one-clip engines only; wino_output_f2 adds the copies in split order)
    const int ks = MODE == 2 ? (int)blockIdx.y : MODE == 1 ? (int)blockIdx.z : 0;
    const int kn = MODE == 2 ? (int)gridDim.y : MODE == 1 ? (int)gridDim.z : 1;
    const int kt0 = (int)(((long long)KT * ks) / kn);
    const int kt1 = (int)(((long long)KT * (ks + 1)) / kn);

    f32x4 ra[AROWS], rb[BPASS];
    f32x4 ra3[3];   // MODE 3: the other three terms of ONE A row (the rows of a thread are loaded and stored one after the other,
                    // so that a single set of temporaries keeps the kernel at 4 workgroups per CU)
    // K order for Cin >= 128: (tap row dy, channel chunk, tap column dx).  The three dx taps of one 64-B
    // channel chunk touch the same input lines shifted by a pixel; run back to back they hit in cache
    // instead of coming round again Cin/16 K steps later (measured HBM fetch of the stride-2 layers 4..8:
    // -17 %).  Narrow layers (Cin = 32: one pixel = one 128-B line = two chunks) keep tap-major order,
    // which reads both halves of a line in consecutive K steps (the other order fetched 30 % more).
    const bool dx_inner = MODE != 1 && MODE != 3 && cchunks >= 8;
    // MODE 3: K step -> position (pr, pc) of the block's sum and channel chunk; the position's terms: row transform
    // B^T = [d0 - d2, d2, d4 - d2, d1, d3] = first patch row (always) and second patch row (-1: none; it enters with a
    // minus sign); the same for columns.  tap_ = the position's weight matrix.
#define APR_S3_DECODE(KT_)                                                                               \
        const int c0_ = st_c0, s3r0 = st_r0, s3r1 = st_r1, s3c0 = st_cs0, s3c1 = st_cs1, tap_ = st_tap;  \
        (void)s3r0; (void)s3r1; (void)s3c0; (void)s3c1; (void)tap_;
    // (the staged tile's position changes every Cin / BK steps: decoded there, not in every step)
#define APR_S3_POSITION()                                                                                \
    {                                                                                                    \
        const int ri_ = st_sp >> s3sh, ci_ = st_sp & s3sh;                                               \
        const int pr_ = s3qi == 1 ? 1 : s3qi == 0 ? (ri_ ? 3 : 0) : (ri_ ? 4 : 2);                       \
        const int pc_ = s3qj == 1 ? 1 : s3qj == 0 ? (ci_ ? 3 : 0) : (ci_ ? 4 : 2);                       \
        st_r0 = pr_ == 0 ? 0 : pr_ == 1 ? 2 : pr_ == 2 ? 4 : pr_ == 3 ? 1 : 3;                           \
        st_r1 = (pr_ == 0 || pr_ == 2) ? 2 : -1;                                                         \
        st_cs0 = pc_ == 0 ? 0 : pc_ == 1 ? 2 : pc_ == 2 ? 4 : pc_ == 3 ? 1 : 3;                          \
        st_cs1 = (pc_ == 0 || pc_ == 2) ? 2 : -1;                                                        \
        st_tap = 4 * (pr_ < 3 ? pr_ : 3) + (pc_ < 3 ? pc_ : 3);                                          \
    }
#define APR_KT_SPLIT(KT_)                                                                                \
        int tap_, c0_;                                                                                   \
        if (MODE == 3) {                                                                                 \
            c0_ = st_c0;                                                                                 \
            tap_ = st_tap;                                                                               \
        } else if (dx_inner) {                                                                                  \
            const int dy_ = (KT_) / (3 * cchunks);                                                       \
            const int rem_ = (KT_) - dy_ * 3 * cchunks;                                                  \
            const int cc_ = rem_ / 3;                                                                    \
            tap_ = 3 * dy_ + (rem_ - 3 * cc_);                                                           \
            c0_ = cc_ * BK;                                                                              \
        } else {                                                                                         \
            tap_ = MODE != 1 ? (KT_) / cchunks : 0;                                                      \
            c0_ = ((KT_) - tap_ * cchunks) * BK;                                                         \
        }
#define APR_LOAD_TILES(KT_)                                                                              \
    {                                                                                                    \
        APR_KT_SPLIT(KT_)                                                                                \
        const int soffA_ = MODE != 1 ? 4 * (((tap_ / 3) * W + (tap_ % 3)) * Cin + c0_) : 4 * c0_;        \
        const int soffB_ = 4 * (tap_ * Cin + c0_) * Cout;                                                \
        if (MODE == 3) {   /* A rows: APR_LOAD_A3, one row at a time */                                  \
        } else {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < AROWS; i++)                                                \
        {                                                                                                \
            const unsigned vo_ = ((rowmask[i] >> tap_) & 1u) ? rowoff[i] : OOB;                          \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo_, soffA_, 0)); \
        }                                                                                                \
        }                                                                                                \
        if (!APR_EXP_BDMA) {                                                                             \
            _Pragma("unroll") for (int i = 0; i < BPASS; i++)                                            \
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, boff[i], soffB_, 0)); \
        }                                                                                                \
    }
    // MODE 3: the terms of A row I_.  Term (r, c) = patch pixel (s3r, s3c); out-of-image pixels read as 0 (out-of-range offset);
    // the position's absent terms are skipped by block-uniform branches: 49 loads per 25 positions, not 100.
#define APR_LOAD_A3(KT_, I_)                                                                             \
    {                                                                                                    \
        APR_S3_DECODE(KT_)                                                                               \
        const unsigned rk_ = rowmask[I_] & 0xFFu, ck_ = rowmask[I_] >> 8;                                \
        const bool r0_ = (rk_ >> s3r0) & 1u, r1_ = s3r1 >= 0 && ((rk_ >> (s3r1 & 7)) & 1u);              \
        const bool c0k_ = (ck_ >> s3c0) & 1u, c1k_ = s3c1 >= 0 && ((ck_ >> (s3c1 & 7)) & 1u);            \
        const int o00_ = 4 * ((s3r0 * W + s3c0) * Cin + c0_), o01_ = 4 * ((s3r0 * W + (s3c1 & 7)) * Cin + c0_); \
        const int o10_ = 4 * (((s3r1 & 7) * W + s3c0) * Cin + c0_), o11_ = 4 * (((s3r1 & 7) * W + (s3c1 & 7)) * Cin + c0_); \
        ra[I_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, r0_ && c0k_ ? rowoff[I_] : OOB, o00_, 0)); \
        if (s3c1 >= 0) ra3[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, r0_ && c1k_ ? rowoff[I_] : OOB, o01_, 0)); \
        if (s3r1 >= 0) ra3[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, r1_ && c0k_ ? rowoff[I_] : OOB, o10_, 0)); \
        if (s3r1 >= 0 && s3c1 >= 0) ra3[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, r1_ && c1k_ ? rowoff[I_] : OOB, o11_, 0)); \
    }
    // B tile straight into LDS (LDS-DMA, no VGPRs, no ds_write): one wave-instruction writes
    // 1 KiB = 64 lanes x 16 B contiguously, which is exactly (64/BCH) whole k-rows of Bs.
#define APR_DMA_B(KT_, BUF_)                                                                             \
    {                                                                                                    \
        APR_KT_SPLIT(KT_)                                                                                \
        const int soffB_ = 4 * (tap_ * Cin + c0_) * Cout;                                                \
        const unsigned lds_ = bs_lds + 4u * (unsigned)((BUF_) * BK * BN);                                \
        _Pragma("unroll") for (int i = 0; i < BPASS; i++)                                                \
            lds_dma_b128(rsBv, boff[i], soffB_, lds_ + 4u * (unsigned)(BROWS * i * BN));                 \
    }
#define APR_STORE_A(BUF_, I_)                                                                            \
    {                                                                                                    \
        float *a_ = As[BUF_] + ar0 + AR * (I_);                                                          \
        a_[(ac * 4 + 0) * LDA] = ra[I_].x;                                                               \
        a_[(ac * 4 + 1) * LDA] = ra[I_].y;                                                               \
        a_[(ac * 4 + 2) * LDA] = ra[I_].z;                                                               \
        a_[(ac * 4 + 3) * LDA] = ra[I_].w;                                                               \
    }
    // MODE 3: (d[r0][c0] - d[r0][c1]) - (d[r1][c0] - d[r1][c1]) of row I_ of K step KT_, absent terms left out
#define APR_COMBINE_A3(KT_, I_)                                                                          \
    {                                                                                                    \
        APR_S3_DECODE(KT_)                                                                               \
        (void)c0_;                                                                                       \
        if (s3c1 >= 0) ra[I_] -= ra3[0];                                                                 \
        if (s3r1 >= 0) {                                                                                 \
            f32x4 t_ = ra3[1];                                                                           \
            if (s3c1 >= 0) t_ -= ra3[2];                                                                 \
            ra[I_] -= t_;                                                                                \
        }                                                                                                \
    }
#define APR_STORE_B(BUF_, I_) *reinterpret_cast<f32x4 *>(Bs[BUF_] + (brow0 + BROWS * (I_)) * BN + bch * 4) = rb[I_];

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    if (MODE == 3) APR_S3_POSITION()
    APR_LOAD_TILES(kt0)
    if (APR_EXP_BDMA) APR_DMA_B(kt0, 0)
    if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < AROWS; i++) { APR_LOAD_A3(kt0, i) APR_COMBINE_A3(kt0, i) APR_STORE_A(0, i) }
    } else {
#pragma unroll
        for (int i = 0; i < AROWS; i++) APR_STORE_A(0, i)
    }
    if (!APR_EXP_BDMA) {
#pragma unroll
        for (int i = 0; i < BPASS; i++) APR_STORE_B(0, i)
    }
    if (APR_EXP_BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#define APR_S3_ADVANCE()                                                                                 \
    if (MODE == 3) {                                                                                     \
        st_c0 += BK;                                                                                     \
        if (st_c0 >= Cin) { st_c0 = 0; st_sp++; APR_S3_POSITION() }                                      \
    }
    APR_S3_ADVANCE()

    // Main loop.  Everything that is not an MFMA is placed in the shadow of the MFMAs: operand
    // fragments are read one k-pair ahead (two register sets); the next tile's buffer loads are
    // issued after the first k-pair and written to the other LDS buffer a piece per k-pair from
    // ST0 on (that buffer was released by the barrier that ended the previous step).  hipcc
    // would otherwise sink every ds_read next to its MFMA (lgkmcnt(0) before each group), so
    // memory instructions and MFMAs are fenced per k-pair; only VALU/SALU may move (mask 0x6).
#define APR_KSTEP(HAS_NEXT)                                                                              \
    {                                                                                                    \
        const float *a = As[cur] + wm * WM + li;                                                         \
        const float *b = Bs[cur] + wn * WN + li;                                                         \
        float af[2][TM], bf[2][TN];                                                                      \
        _Pragma("unroll") for (int i = 0; i < TM; i++) af[0][i] = a[half * LDA + i * 32];                \
        _Pragma("unroll") for (int j = 0; j < TN; j++) bf[0][j] = b[half * BN + j * 32];                 \
        _Pragma("unroll") for (int kk = 0; kk < NK; kk++)                                                \
        {                                                                                                \
            if (kk + 1 < NK) {                                                                           \
                const int k = 2 * (kk + 1) + half;                                                       \
                _Pragma("unroll") for (int i = 0; i < TM; i++) af[(kk + 1) & 1][i] = a[k * LDA + i * 32]; \
                _Pragma("unroll") for (int j = 0; j < TN; j++) bf[(kk + 1) & 1][j] = b[k * BN + j * 32]; \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0x6);                                                         \
            if (APR_EXP_STAGE && APR_EXP_LOAD && HAS_NEXT && kk == 1) APR_LOAD_TILES(kt + 1)             \
            if (APR_EXP_STAGE && APR_EXP_BDMA && HAS_NEXT && kk == 2) APR_DMA_B(kt + 1, cur ^ 1)         \
            if (MODE != 3 && APR_EXP_STAGE && APR_EXP_STORE && HAS_NEXT && kk >= ST0 && kk < ST0 + AROWS) APR_STORE_A(cur ^ 1, kk - ST0) \
            if (MODE == 3 && HAS_NEXT) {   /* row 0: loaded after k-pair 0, stored at 3; row 1: loaded at 3, stored at 6 */ \
                constexpr int R1 = AROWS > 1 ? 1 : 0;                                                    \
                if (kk == 0) APR_LOAD_A3(kt + 1, 0)                                                      \
                if (kk == 3) { APR_COMBINE_A3(kt + 1, 0) APR_STORE_A(cur ^ 1, 0) if (AROWS > 1) APR_LOAD_A3(kt + 1, R1) } \
                if (AROWS > 1 && kk == 6) { APR_COMBINE_A3(kt + 1, R1) APR_STORE_A(cur ^ 1, R1) }        \
            }                                                                                            \
            if (APR_EXP_STAGE && APR_EXP_STORE && !APR_EXP_BDMA && HAS_NEXT && kk >= ST0 && kk < ST0 + BPASS) APR_STORE_B(cur ^ 1, kk - ST0) \
            if (APR_EXP_STAGE && !APR_EXP_STORE && HAS_NEXT && kk == ST0) {                              \
                _Pragma("unroll") for (int i = 0; i < AROWS; i++) asm volatile("" ::"v"(ra[i]));          \
                if (!APR_EXP_BDMA) { _Pragma("unroll") for (int i = 0; i < BPASS; i++) asm volatile("" ::"v"(rb[i])); } \
            }                                                                                            \
            _Pragma("unroll") for (int i = 0; i < TM; i++)                                               \
                _Pragma("unroll") for (int j = 0; j < TN; j++)                                           \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0x6);                                                         \
        }                                                                                                \
        if (APR_EXP_BDMA && HAS_NEXT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
        if (APR_EXP_SYNC) __syncthreads();                                                               \
        cur ^= 1;                                                                                        \
        APR_S3_ADVANCE()                                                                                 \
    }

    int cur = 0;
    int kt = kt0;
    for (; kt < kt1 - 1; kt++) APR_KSTEP(true)
    APR_KSTEP(false)
#undef APR_KSTEP
#undef APR_S3_ADVANCE
#undef APR_S3_POSITION

    // ---- epilogue: bias + ReLU, NHWC store.  D[row][col]: col = lane&31,
    //      row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    const bool full = p0 + BM <= M;      // block-uniform: no per-row bounds test for whole tiles
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = bn * BN + wn * WN + j * 32 + li;
        const float bv = MODE == 0 ? bias[n] : 0.0f;   // (MODE 1, 2, 3: raw sums)
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int prow = p0 + wm * WM + i * 32 + 4 * half;
            float *o = out + (long long)prow * Cout + n;
            if (full) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const float v = acc[i][j][e] + bv;
                    __builtin_nontemporal_store(MODE == 0 ? fmaxf(v, 0.0f) : v, o + (long long)((e & 3) + 8 * (e >> 2)) * Cout);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int p = prow + (e & 3) + 8 * (e >> 2);
                    const float v = acc[i][j][e] + bv;
                    if (p < M) out[(long long)p * Cout + n] = MODE == 0 ? fmaxf(v, 0.0f) : v;
                }
            }
        }
    }
}

#undef APR_LOAD_TILES
#undef APR_LOAD_A3
#undef APR_COMBINE_A3
#undef APR_S3_DECODE
#undef APR_KT_SPLIT
#undef APR_STORE_A
#undef APR_STORE_B
#undef APR_DMA_B

// ---------------------------------------------------------------------------------------------
// split-K tail: out[p][n] = relu(bias[n] + part[0][p][n] + part[1][p][n] + ...), splits added in
// index order so the result does not depend on scheduling.  n4 = M * Cout / 4 float4 elements.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                            float *__restrict__ out, long long n4, int cout4, int nsplit)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 a = *reinterpret_cast<const f32x4 *>(bias + 4 * (i % cout4));
        // eight partials in flight at a time (a one-clip grid is a few workgroups: a chain of dependent loads was 64 L2
        // latencies long); the additions stay in split order
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int s = s0 + u < nsplit ? s0 + u : nsplit - 1;
                v[u] = *reinterpret_cast<const f32x4 *>(part + 4 * (s * n4 + i));
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (s0 + u < nsplit) a += v[u];
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(out + 4 * i) = __builtin_elementwise_max(a, zero);
    }
}

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
#pragma unroll 8   // loads in flight together: one clip is one workgroup, and a chain of dependent loads is all it waits for
        for (int p = 1; p < P; p++) m = fmaxf(m, a[(long long)p * F + c]);
        s_feat[c] = m;
    }
    __syncthreads();
    // one wave per class (round-robin): 64-lane strided dot + shuffle reduce
    for (int o = wave; o < C; o += 4) {
        float s = 0.0f;
#pragma unroll 16   // same fmaf chain, the 16 weight loads issued together
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
