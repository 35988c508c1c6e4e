// conv12_fused.hip.h -- layers 1 and 2 of the NoiseDetector CNN (ai/model.py:57-64) in one kernel.
//
// Layer 1 (1 -> 32 channels, stride 1) turns a 96 KB spectrogram into 3.08 MB of activations that
// layer 2 (32 -> 64, stride 2) reads back nine taps over: as two kernels that is 3.2 GB written and
// 6.5 GB fetched per 1024 windows, the only HBM-heavy stretch of the network.  Here a workgroup owns
// a 4 x 32 tile of layer-2 outputs: it computes the 9 x 65 x 32 layer-1 patch under that tile from the
// spectrogram into LDS (a small MFMA GEMM with the same fma chain as conv1_kernel: bit-identical) and
// then runs layer 2's implicit GEMM (M = 128 pixels, N = 64, K = 9 x 32) with the MFMA A operand read
// straight out of that patch -- tap (dy, dx) is just an address offset -- and the B operand (74 KB of
// weights) held in registers.  No A tile is ever staged, nothing is double-buffered, and the K loop has no barrier.
// Workgroups are persistent (two per CU, each walks tiles bid, bid + grid, ...): all 144 B fragments of a
// wave (9 taps x 16 k-pairs) stay in registers for the kernel's life, and the next tile's spectrogram
// patch is fetched into registers while the current tile's MFMAs run, so per tile only the layer-1
// phase and two barriers are not MFMA time, and no wait ever stands behind a tile's output stores.
// K order (tap, channel) and the MFMA operand order are those of conv3x3_igemm_f32<128, 64, 2, 0>, so
// the outputs are bit-identical to the two-kernel path.
//
// LDS: patch [32 ch][585] (channel stride odd: the two k-halves of a wave hit odd/even banks; lanes
// read pixel stride 2 -> 32 distinct even banks) + spectrogram patch [11][67] = 77.8 KB, 2 workgroups/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv.hip.h"

namespace apr {

constexpr int C12_TW = 32, C12_TH = 4;                    // layer-2 output tile
constexpr int C12_PW = 2 * C12_TW + 1, C12_PH = 2 * C12_TH + 1;   // layer-1 patch 65 x 9
constexpr int C12_PS = C12_PW * C12_PH;                   // 585 (odd)
constexpr int C12_SW = C12_PW + 2, C12_SH = C12_PH + 2;   // spectrogram patch 67 x 11
constexpr int C12_SN = C12_SW * C12_SH;                   // 737
constexpr int C12_LDS_BYTES = (32 * C12_PS + C12_SN) * 4;

#ifdef C12_STAMPS   // developer build (tools/conv12_bench.cpp): cycles per phase, summed over a block's tiles (wave 0 lane 0)
__device__ unsigned long long c12_stamps[1024][8];
#define C12_STAMP(I_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[I_] += now_ - last_; last_ = now_; }
#else
#define C12_STAMP(I_)
#endif

__global__ __launch_bounds__(256, 2) void conv12_fused_kernel(const float *__restrict__ spec, const float *__restrict__ w1t,
                                                              const float *__restrict__ b1, const float *__restrict__ w2,
                                                              const float *__restrict__ b2, float *__restrict__ out, int H1, int W1,
                                                              int Ho, int Wo, int tiles_x, int tiles_y, unsigned n_tiles)
{
    extern __shared__ float c12_lds[];
    float *pt = c12_lds;                    // [32][C12_PS]
    float *sp = c12_lds + 32 * C12_PS;      // [C12_SH][C12_SW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // spectrogram patch of a tile -> three registers per lane: rows 2 oy0 - 2 .. + 10, cols 2 ox0 - 2 .. + 66, zero outside
    float spr[3];
#define C12_LOAD_SP(TILE)                                                                              \
    {                                                                                                  \
        const unsigned tx_ = (TILE) % (unsigned)tiles_x, t1_ = (TILE) / (unsigned)tiles_x;             \
        const unsigned ty_ = t1_ % (unsigned)tiles_y;                                                  \
        const float *img_ = spec + (size_t)(t1_ / (unsigned)tiles_y) * H1 * W1;                        \
        _Pragma("unroll") for (int j = 0; j < 3; j++)                                                  \
        {                                                                                              \
            const int i = tid + 256 * j;                                                               \
            const int r = i / C12_SW, c = i - r * C12_SW;                                              \
            const int ys = 2 * (int)ty_ * C12_TH - 2 + r, xs = 2 * (int)tx_ * C12_TW - 2 + c;          \
            spr[j] = (i < C12_SN && ys >= 0 && ys < H1 && xs >= 0 && xs < W1) ? img_[(size_t)ys * W1 + xs] : 0.0f; \
        }                                                                                              \
    }

    // layer-1 operands of this lane: B = w1t[t][li] for tap t = 2 kp + half, the bias for the tenth "tap"
    float bw[5];
    int aoff[5];
#pragma unroll
    for (int kp = 0; kp < 5; kp++) {
        const int t = 2 * kp + half;
        bw[kp] = t < 9 ? w1t[t * 32 + li] : b1[li];
        aoff[kp] = t < 9 ? (t / 3) * C12_SW + (t % 3) : 0;
    }
    // layer-2 B fragments: all 9 x 16 k-pairs of this wave's 32 output channels stay in registers for the kernel's
    // life (144 VGPRs of the 256 a wave has at two waves per SIMD), so the tile loop issues no weight loads and no
    // vmcnt wait ever stands behind the previous tile's stores.
    float bq[9][16];
    {
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)w2, 0, 9 * 32 * 64 * 4, 0x00020000);
        const unsigned bvo = 4u * (unsigned)(half * 64 + 32 * wn + li);
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
            for (int kk = 0; kk < 16; kk++)
                bq[tap][kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, bvo + 512u * (kk % 8), (tap * 32 + 16 * (kk / 8)) * 256, 0));
    }
    const int n = 32 * wn + li;
    const float bv = b2[n];
    // wave rows: 64 wm + 32 i + li  ->  tile row 2 wm + i, tile column li
    const float *a0p = pt + (2 * (2 * wm + 0)) * C12_PW + 2 * li + half * C12_PS;
    const float *a1p = pt + (2 * (2 * wm + 1)) * C12_PW + 2 * li + half * C12_PS;

#ifdef C12_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    unsigned tile = blockIdx.x;
    if (tile < n_tiles) C12_LOAD_SP(tile)
    for (; tile < n_tiles; tile += gridDim.x) {
        const int tx = (int)(tile % (unsigned)tiles_x);
        const unsigned t1 = tile / (unsigned)tiles_x;
        const int ty = (int)(t1 % (unsigned)tiles_y);
        const size_t b = t1 / (unsigned)tiles_y;
        const int oy0 = ty * C12_TH, ox0 = tx * C12_TW;

#pragma unroll
        for (int j = 0; j < 3; j++)
            if (tid + 256 * j < C12_SN) sp[tid + 256 * j] = spr[j];
        C12_STAMP(0)
        __syncthreads();   // sp published; every wave has finished reading the previous tile's patch

        C12_STAMP(1)
        // ---- phase 1: layer-1 patch on the matrix cores: [585 pixels] x [K = 9 taps + a bias tap] x [32 channels] in
        //      32-pixel row blocks (19 blocks dealt over the 4 waves), five v_mfma_f32_32x32x2 each.  Lane li supplies
        //      pixel 32 mb + li (A: its spectrogram neighbourhood) and channel li (B).  The MFMA's fma chain in k order
        //      is conv1_kernel's: nine fmaf from 0, then fma(1, bias, acc) = acc + bias rounded once; ReLU on the VALU.
        //      A pixel outside the layer-1 image gets an all-zero A row and comes out as exactly 0: layer 2's padding ----
        constexpr int kFullBlocks = C12_PS / 32;                      // 18 whole row blocks, then one of 9 rows
#pragma unroll
        for (int g = 0; g < 2; g++) {                                  // trips j = 0,1,2 then 3,4: their MFMA chains interleave
            constexpr int JN = 3;
            f32x16 acc[JN];
            float av[JN][5];
            bool valid[JN];
#pragma unroll
            for (int jj = 0; jj < JN; jj++) {
                const int j = 3 * g + jj;
                if (j >= 5) continue;
                const int mb = wave + 4 * j;                           // wave-uniform; mb = 19 (wave 3, last trip) is past the patch
                const int p = mb * 32 + li;
                const int pc = min(p, C12_PS - 1);
                const int r = pc / C12_PW, c = pc - r * C12_PW;
                const int y1 = 2 * oy0 - 1 + r, x1 = 2 * ox0 - 1 + c;
                valid[jj] = p < C12_PS && y1 >= 0 && y1 < H1 && x1 >= 0 && x1 < W1;
                const float *sb = sp + r * C12_SW + c;
#pragma unroll
                for (int kp = 0; kp < 5; kp++) av[jj][kp] = sb[aoff[kp]];
                if (half) av[jj][4] = 1.0f;
#pragma unroll
                for (int e = 0; e < 16; e++) acc[jj][e] = 0.0f;
            }
#pragma unroll
            for (int kp = 0; kp < 5; kp++)
#pragma unroll
                for (int jj = 0; jj < JN; jj++)
                    if (3 * g + jj < 5) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(valid[jj] ? av[jj][kp] : 0.0f, bw[kp], acc[jj], 0, 0, 0);
#pragma unroll
            for (int jj = 0; jj < JN; jj++) {
                const int j = 3 * g + jj;
                if (j >= 5) continue;
                const int mb = wave + 4 * j;
                float *po = pt + li * C12_PS + mb * 32 + 4 * half;
                if (4 * j + 3 < kFullBlocks || mb < kFullBlocks) {    // whole block (always, except on the last trip): plain stores
#pragma unroll
                    for (int e = 0; e < 16; e++) po[(e & 3) + 8 * (e >> 2)] = fmaxf(acc[jj][e], 0.0f);
                } else if (mb == kFullBlocks) {                        // block 18: rows 576..584 only
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        if (mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * half < C12_PS) po[(e & 3) + 8 * (e >> 2)] = fmaxf(acc[jj][e], 0.0f);
                }
            }
        }
        C12_STAMP(2)
        __syncthreads();   // patch published; sp is free again
        C12_STAMP(3)

        if (tile + gridDim.x < n_tiles) C12_LOAD_SP(tile + gridDim.x)   // in flight during the MFMAs below

        // ---- phase 2: layer 2 as an implicit GEMM out of the patch.  A fragments are read three k-pairs ahead (ring of
        //      four register pairs); each k-pair is fenced so that hipcc keeps that order and its lgkmcnt waits stay counted ----
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; e++) { acc0[e] = 0.0f; acc1[e] = 0.0f; }
        float af[4][2];
#define C12_RD(Q)                                                                                  \
    {                                                                                              \
        const int t_ = (Q) / 16, k_ = (Q) % 16;                                                    \
        const int o_ = (t_ / 3) * C12_PW + (t_ % 3) + 2 * k_ * C12_PS;                             \
        af[(Q) & 3][0] = a0p[o_];                                                                  \
        af[(Q) & 3][1] = a1p[o_];                                                                  \
    }
        C12_RD(0) C12_RD(1) C12_RD(2)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                const int q = tap * 16 + kk;
                if (q + 3 < 144) C12_RD(q + 3)
                __builtin_amdgcn_sched_barrier(0x6);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 3][0], bq[tap][kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 3][1], bq[tap][kk], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0x6);
            }
        }
#undef C12_RD
        C12_STAMP(4)

        // ---- epilogue: bias + ReLU, NHWC streaming stores.  D row = (e&3) + 8 (e>>2) + 4 half = tile column ----
#pragma unroll
        for (int i = 0; i < 2; i++) {
            // one buffer descriptor per output row: columns past Wo (the last tile of a row) and rows past Ho fall
            // outside num_records and the hardware drops those lanes' stores -- no branch, no select
            const int oy = oy0 + 2 * wm + i;
            float *orow = out + ((b * Ho + oy) * (size_t)Wo) * 64;
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void *)orow, 0, oy < Ho ? Wo * 256 : 0, 0x00020000);
            const unsigned vo = 4u * (unsigned)((ox0 + 4 * half) * 64 + n);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float v = fmaxf((i == 0 ? acc0[e] : acc1[e]) + bv, 0.0f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, vo + 256u * (unsigned)((e & 3) + 8 * (e >> 2)), 0, 2 /* nt */);
            }
        }
        C12_STAMP(5)
    }
#ifdef C12_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 1024) for (int i = 0; i < 8; i++) c12_stamps[blockIdx.x][i] = st_[i];
#endif
#undef C12_LOAD_SP
}

}  // namespace apr
