// wino4_fused.hip.h -- Winograd F(4x4, 3x3), stride 1, in TWO kernels: neither M nor the output transform touch HBM.
//
//   wino_input_f4p   d (NHWC) -> V = B^T d B, written in the consumer's LDS image order (below)
//   wino4_fused_out  the 36 position GEMMs M[pos] = V[pos] x U[pos], the output transform A^T M A, + bias, ReLU, NHWC store
//
// Why two kernels and not one (wino_fused8_f2's form): a 6x6 patch IS its V (36 values either way), so a kernel that loads V
// moves the same bytes into the CU as one that loads patches, and the K loop then holds no VALU work at all -- fp32 VALU
// and fp32 MFMA do not overlap on this chip, and F(4x4) has 4x the transform arithmetic per MFMA of F(2x2).
// Why not 36 positions side by side: 36 x (64 tiles x 64 channels) accumulators are 2.25x the register file.  The output
// transform is linear in the positions, so the unit is computed in SIX PASSES, one row i of the 6x6 positions each:
// 6 position GEMMs (24 streams of 32x32 blocks over 8 waves: wave = block x 3 positions, 48 accumulator registers; waves w and w + 4
// share a block and a SIMD), then the two waves of a block swap half of the pass's M through LDS (each keeps 16 tiles of its three
// positions and gets the same 16 tiles of the other three: 24 registers a lane each way, lane-linear 16-B accesses, one round), and
// the transform runs in the MFMA layout (lane = one channel, eight tiles): R[b] = sum_j A^T[b][j] M[i][j] and Y[a][b] += A^T[a][i] R[b]
// accumulate the unit's 16 outputs per (tile, channel) in registers (128 a lane).  After pass 5: ReLU (Y starts from the bias) and
// 4-byte stores, 32 consecutive channels (128 B) of two pixels an instruction.
//
// Work unit = 64 tiles (1024 output pixels) x 64 channels, one persistent workgroup of 8 waves per CU (wf8_grid, wf_unit order).
// K step = 16 input channels = one slab each of V and U: [sub 2][j 6][row 64][8 floats] = 24 KiB, host- / producer-packed in
// exactly that order with wino_fused_f2's half-swizzle, so a step's operands are two contiguous 24-KiB LDS-DMA copies
// (every wave moves 3 KiB of each) and every ds_read_b128 feeds four MFMAs conflict-free.  Double buffered, the DMA of step
// s+2 (which may belong to the next pass or the next unit) issued at the top of step s into the slot step s-1 read: three slots,
// one barrier a step.  The slot a pass's last step has read is that pass's exchange buffer [wave 8][6][lane 64] x 16 B.
// LDS: 3 slots x 48 KiB + two tile tables 1 KiB = 145 KiB.
#pragma once
#include <hip/hip_runtime.h>

#include "wino4.hip.h"
#include "wino_fused8.hip.h"

namespace apr {

#ifndef W4_ABL
#define W4_ABL 0   // developer timing ablations (wrong results): 1 no slab DMA after a unit's first, 2 no exchange through X, 4 no output stores, 8 no transform arithmetic
#endif

#ifndef W4_UNROLL
#define W4_UNROLL 1   // the six passes as straight-line code: the column of A^T is then a set of constants (a third of the Y update is a multiplication by 0)
#endif

#ifdef W4_STAMPS   // developer build (tools/w4_stamps.py): s_memtime stamps of workgroup 0, per wave: cycles in the MFMA part of the K steps [0], in their
                   // closing wait + barrier [1], in the pass ends: up to the first barrier [6], exchange up to the closing barrier [7], transform [2]; between two units' K loops (stores, set-up) [3]; steps [4], passes [5]
__device__ unsigned long long w4_stamps[8][8];
#define W4_ST(I_) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); st4[I_] += t_ - stt; stt = t_; }
#else
#define W4_ST(I_)
#endif

constexpr int W4_TT = 64;                        // tiles per unit
constexpr int W4_BN = 64;                        // output channels per unit
constexpr int W4_BK = 16;                        // input channels per K step
constexpr int W4_SLAB = 2 * 6 * 64 * 8;          // floats of one operand slab (one pass row, one K step)
constexpr int W4_SLOT = 2 * W4_SLAB;             // a K step's V slab + U slab = 48 KiB = the exchange buffer [8 waves][6][64 lanes] x 16 B
constexpr int W4_LDS_BYTES = 3 * W4_SLOT * 4 + 2 * 64 * 8;
static_assert(W4_LDS_BYTES <= 160 * 1024, "wino4_fused_out: LDS over the CU's 160 KiB");

// element (row r of 64, channel c of the K step's 16) of position (i, j) of K step kc of M / N tile b, in floats
__host__ __device__ __forceinline__ size_t w4_slab_index(int b, int i, int kc, int KC, int j, int r, int c)
{
    const int sub = c >> 3, cl = c & 7;
    return ((((size_t)b * 6 + i) * KC + kc) * 2 + sub) * (6 * 512) + (size_t)j * 512 + r * 8 + ((((cl >> 2) ^ ((r >> 4) & 1)) * 4) + (cl & 3));
}

// Host: g [Cout][Cin][3][3] (OIHW, BN folded) -> U = G g G^T (f64, rounded once) in slab order
inline void w4_pack_u(const float *w, int cin, int cout, float *up)
{
    static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int KC = cin / W4_BK;
    for (int o = 0; o < cout; o++)
        for (int c = 0; c < cin; c++) {
            const float *g = &w[((size_t)o * cin + c) * 9];
            double t[6][3];
            for (int i = 0; i < 6; i++)
                for (int k = 0; k < 3; k++) t[i][k] = G[i][0] * g[k] + G[i][1] * g[3 + k] + G[i][2] * g[6 + k];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++)
                    up[w4_slab_index(o / W4_BN, i, c / W4_BK, KC, j, o % W4_BN, c % W4_BK)] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}

// in [B][H][W][C] -> V in slab order; tile t = (b, ty, tx) covers input rows 4ty-1 .. 4ty+4 (pad 1).
// One workgroup = one unit's 64 tiles x one K step's 16 channels = the 36 [sub][row][8] pieces of one (unit, kc): thread = (row, channel
// quad), so four lanes read 64 contiguous bytes of a pixel and a wave's store instruction writes two 512-B runs (rows 16 w .. 16 w + 15 of
// sub 0 and of sub 1).  Workgroups of one unit (kc = 0 .. KC-1) follow each other inside an XCD's contiguous run, so the 128-B lines
// of a pixel that the KC workgroups share are fetched into one L2.
// Rows of the last unit past T are never written: their products land in output rows that are never stored.
template <bool NT>   // NT: streaming stores (two instantiations: under a run-time branch hipcc merges the two stores and drops the hint)
__global__ __launch_bounds__(256) void wino_input_f4p(const float *__restrict__ in, float *__restrict__ Vp, int H, int W, int C, int th, int tw,
                                                      long long T, int n_blocks)
{
    const int KC = C / W4_BK;
    // XCD-aware order: workgroup blockIdx.x runs on XCD blockIdx.x % 8; give every XCD a contiguous run of logical blocks
    const int q8 = n_blocks >> 3, r8 = n_blocks & 7, xcd = blockIdx.x & 7;
    const int lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int bm = lb / KC, kc = lb - bm * KC;
    const int row = threadIdx.x >> 2, quad = threadIdx.x & 3;
    const long long t = (long long)bm * W4_TT + row;
    if (t >= T) return;
    const int c = kc * W4_BK + quad * 4;
    const int tx = (int)(t % tw);
    const long long r = t / tw;
    const int ty = (int)(r % th);
    const long long b = r / th;
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const float *base = in + (b * H * (long long)W) * C + c;
    f32x4 tmp[6][6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int x = x0 + j;
        f32x4 d[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int y = y0 + i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < H && x >= 0 && x < W) v = ld4(base + ((long long)y * W + x) * C);
            d[i] = v;
        }
        wino4_bt(d[0], d[1], d[2], d[3], d[4], d[5], tmp[0][j], tmp[1][j], tmp[2][j], tmp[3][j], tmp[4][j], tmp[5][j]);
    }
    float *o0 = Vp + w4_slab_index(bm, 0, kc, KC, 0, row, quad * 4);
    const size_t pass_stride = (size_t)KC * W4_SLAB;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        f32x4 v[6];
        wino4_bt(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            // a V far larger than the Infinity Cache is read once, by another kernel, long after: streaming stores keep it out of the caches the
            // patches' halo reads live in (layer 3 at 4096 windows: 4.31 -> 3.94 ms; a V of 0.6 GB is better left cached: 0.14 vs 0.16 ms)
            if (NT) __builtin_nontemporal_store(v[j], reinterpret_cast<f32x4 *>(o0 + i * pass_stride + j * 512));
            else st4(o0 + i * pass_stride + j * 512, v[j]);
        }
    }
}

struct W4Args {
    const float *vp, *up, *bias;
    float *out;
    int H, W, Cin, Cout, th, tw, T, tiles_n;
    WfMagic mg;
    int n_units;
};

__device__ __forceinline__ f32x2 w4_fma2(float c, f32x2 a, f32x2 b) { return __builtin_elementwise_fma(f32x2{c, c}, a, b); }

__global__ __launch_bounds__(512) void wino4_fused_out(const W4Args a)
{
    extern __shared__ __attribute__((aligned(16))) float w4_smem[];      // [3 slots][V slab, U slab]; a consumed slot doubles as the exchange buffer
    unsigned *tbl = reinterpret_cast<unsigned *>(w4_smem + 3 * W4_SLOT);  // [2][64][2]: byte offset of the tile's pixel (0,0), mask of its pixels that exist

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wave & 3, jh = wave >> 2;       // the wave's 32 x 32 block and its three positions of a pass (waves w, w + 4 share a SIMD and a block)
    const int wm = blk >> 1, wn = blk & 1;
    const int li = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, Cout = a.Cout, th = a.th, tw = a.tw, T = a.T;
    const int KC = a.Cin / W4_BK, S = 6 * KC;       // K steps of a pass, steps of a unit (a multiple of 3: every unit starts in slot 0)

    // persistent workgroups, XCD-aware (wino_fused8_f2): the units cut into 8 contiguous runs, one per XCD
    const int per = gridDim.x >> 3, xcd = blockIdx.x & 7;
    const int uq = a.n_units >> 3, ur = a.n_units & 7;
    const int x_start = xcd < ur ? xcd * (uq + 1) : ur * (uq + 1) + (xcd - ur) * uq;
    const int x_count = uq + (xcd < ur ? 1 : 0);
    int lidl = blockIdx.x >> 3;
    if (lidl >= x_count) return;

    constexpr unsigned OOB = 0x80000000u;
    i32x4 rsV, rsU, rsVn, rsUn;
    rsV.z = rsU.z = rsVn.z = rsUn.z = 0x7FFFFFFF;
    rsV.w = rsU.w = rsVn.w = rsUn.w = 0x00020000;
#define W4_DESC(RV_, RU_, BM_, BN_)                                                                      \
    {                                                                                                    \
        const unsigned long long v_ = (unsigned long long)(a.vp + (size_t)(BM_) * S * W4_SLAB);          \
        const unsigned long long u_ = (unsigned long long)(a.up + (size_t)(BN_) * S * W4_SLAB);          \
        RV_.x = __builtin_amdgcn_readfirstlane((int)(unsigned)v_);                                       \
        RV_.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(v_ >> 32) & 0xFFFF);                      \
        RU_.x = __builtin_amdgcn_readfirstlane((int)(unsigned)u_);                                       \
        RU_.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(u_ >> 32) & 0xFFFF);                      \
    }
    // a step's DMA: this wave's three 1-KiB pieces of the V slab and of the U slab, into slot SLOT_
    const unsigned lds0 = (unsigned)(unsigned long long)w4_smem;
    const unsigned dvoff = (unsigned)lane * 16u;
#define W4_DMA(RV_, RU_, STEP_, SLOT_)                                                                   \
    {                                                                                                    \
        _Pragma("unroll") for (int p_ = 0; p_ < 3; p_++)                                                 \
        {                                                                                                \
            const int pc_ = (3 * wave + p_) * 1024;                                                      \
            lds_dma_b128(RV_, dvoff, __builtin_amdgcn_readfirstlane((STEP_) * (W4_SLAB * 4) + pc_),      \
                         __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((SLOT_) * W4_SLOT * 4 + pc_))); \
            lds_dma_b128(RU_, dvoff, __builtin_amdgcn_readfirstlane((STEP_) * (W4_SLAB * 4) + pc_),      \
                         __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((SLOT_) * W4_SLOT + W4_SLAB) * 4 + pc_))); \
        }                                                                                                \
    }

    // one 1-KiB piece: P_ = 0..2 of the V slab, 3..5 of the U slab
#define W4_DMA1(RV_, RU_, STEP_, SLOT_, P_)                                                              \
    {                                                                                                    \
        const int pc_ = (3 * wave + ((P_) % 3)) * 1024;                                                  \
        if ((P_) < 3)                                                                                    \
            lds_dma_b128(RV_, dvoff, __builtin_amdgcn_readfirstlane((STEP_) * (W4_SLAB * 4) + pc_),      \
                         __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((SLOT_) * W4_SLOT * 4 + pc_))); \
        else                                                                                             \
            lds_dma_b128(RU_, dvoff, __builtin_amdgcn_readfirstlane((STEP_) * (W4_SLAB * 4) + pc_),      \
                         __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((SLOT_) * W4_SLOT + W4_SLAB) * 4 + pc_))); \
    }
    // floats into a slot: this wave's first position, its lane's row of V (the jh = 1 waves take rows 16..31 of their block first: the pass ends
    // below) and of U
    const int lr = li ^ (jh << 4);
    const int rd_a = (3 * jh) * 512 + (wm * 32 + lr) * 8 + (half ^ ((lr >> 4) & 1)) * 4;
    const int rd_b = W4_SLAB + (3 * jh) * 512 + (wn * 32 + li) * 8 + (half ^ ((li >> 4) & 1)) * 4;

    int bm, bn;
    wf_unit(x_start + lidl, a.mg, bm, bn);
    W4_DESC(rsV, rsU, bm, bn)
    W4_DMA(rsV, rsU, 0, 0)
    W4_DMA(rsV, rsU, 1, 1)
    int tb = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef W4_STAMPS
    unsigned st4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned stt = (unsigned)__builtin_amdgcn_s_memtime();
#endif

    for (;;) {
        const int nl = lidl + per;
        const bool more = nl < x_count;                 // workgroup-uniform
        int bm_n, bn_n;
        wf_unit(x_start + (more ? nl : lidl), a.mg, bm_n, bn_n);      // the last unit of a workgroup prefetches its own first steps again: staging nobody reads
        W4_DESC(rsVn, rsUn, bm_n, bn_n)
        const int g0 = bm * W4_TT;
        const int b0 = wf_div(g0, a.mg.tpw);            // the unit's first window: output offsets are relative to it
        if (tid < 64) {
            unsigned off = OOB, mask = 0;
            const int g = g0 + tid;
            if (g < T) {
                const WfUnitBase ub = wf_unit_base(g0, a.mg, th, tw);
                int bb, ty, tx;
                wf_tile(ub, tid, a.mg, th, tw, bb, ty, tx);
                off = 4u * (unsigned)((((bb - b0) * H + 4 * ty) * W + 4 * tx) * Cout);
#pragma unroll
                for (int p = 0; p < 16; p++)
                    if (4 * ty + (p >> 2) < H && 4 * tx + (p & 3) < W) mask |= 1u << p;
            }
            tbl[tb * 128 + 2 * tid] = off;
            tbl[tb * 128 + 2 * tid + 1] = mask;
        }
        const int n = bn * W4_BN + wn * 32 + li;        // the lane's output channel: the transform works in the MFMA layout
        const float bv = a.bias[n];
        f32x2 y[16][4];                                 // the unit's 16 outputs of the lane's 8 tiles, starting from the bias
#pragma unroll
        for (int p = 0; p < 16; p++)
#pragma unroll
            for (int e = 0; e < 4; e++) y[p][e] = f32x2{bv, bv};
        __syncthreads();                                // step 0 of this unit is in slot 0 (every wave waited for its own pieces)

        W4_ST(3)
        int s = 0, slot = 0;
#if W4_UNROLL
#pragma unroll
#endif
        for (int i = 0; i < 6; i++) {
            f32x16 acc[3];
            for (int kc = 0; kc < KC; kc++, s++) {
                // the slabs of the step after next, now, into the slot the last step read (the steps past this unit's end are steps 0 and 1 of
                // the workgroup's next unit)
                // one 1-KiB piece ahead of every group of four MFMAs (issued together at the top of the step they cost 14 % of it)
                const int slot2 = slot == 0 ? 2 : slot - 1;
                i32x4 rsVs = rsV, rsUs = rsU;
                int step2 = s + 2;
                if (step2 >= S) { rsVs.x = rsVn.x; rsVs.y = rsVn.y; rsUs.x = rsUn.x; rsUs.y = rsUn.y; step2 -= S; }
                if (kc == 0) {
#pragma unroll
                    for (int jj = 0; jj < 3; jj++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[jj][e] = 0.0f;
                }
                const float *va = w4_smem + slot * W4_SLOT + rd_a;
                const float *ub = w4_smem + slot * W4_SLOT + rd_b;
                f32x4 af[2], bf[2];
                af[0] = *reinterpret_cast<const f32x4 *>(va);
                bf[0] = *reinterpret_cast<const f32x4 *>(ub);
#pragma unroll
                for (int g = 0; g < 6; g++) {           // g = 3 sub + jj
                    if (!(W4_ABL & 1)) W4_DMA1(rsVs, rsUs, step2, slot2, g)
                    if (g < 5) {
                        const int o = ((g + 1) / 3) * (6 * 512) + ((g + 1) % 3) * 512;
                        af[(g + 1) & 1] = *reinterpret_cast<const f32x4 *>(va + o);
                        bf[(g + 1) & 1] = *reinterpret_cast<const f32x4 *>(ub + o);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        acc[g % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][q], bf[g & 1][q], acc[g % 3], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                W4_ST(0)
                if (kc == KC - 1) {
                    // ---- the pass's M between the two waves of a block: wave (blk, jh) holds positions 3 jh .. 3 jh + 2 of all 32 tiles of its block, the
                    //      transform wants all six positions of a tile in one lane.  Each wave keeps accumulator registers 0..7 (16 tiles: the A rows of
                    //      the jh = 1 waves are taken with bit 4 flipped, so that these are tiles 0..15 for jh = 0 and 16..31 for jh = 1) and hands registers
                    //      8..15 to its partner through X = the slot this step read, [wave][6][lane] x 16 B: lane-linear both ways, no bank conflicts ----
                    float *X = w4_smem + slot * W4_SLOT;
                    int lx = lane;
                    asm volatile("" : "+v"(lx));            // the lane's exchange addresses from scratch: kept across the K loops they are spilled
                    float *xw = X + (wave * 6 * 64 + lx) * 4;
                    const float *xr = X + ((wave ^ 4) * 6 * 64 + lx) * 4;
                    // column i of A^T as scalars (rows 0 and 5 of M reach one output row each: the multiplications by 0 are kept, the
                    // code is one straight line for every pass).  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
                    const float c0 = i < 5 ? 1.0f : 0.0f;
                    const float c1 = (i == 1) ? 1.0f : (i == 2) ? -1.0f : (i == 3) ? 2.0f : (i == 4) ? -2.0f : 0.0f;
                    const float c2 = (i == 1 || i == 2) ? 1.0f : (i == 3 || i == 4) ? 4.0f : 0.0f;
                    const float c3 = (i == 1) ? 1.0f : (i == 2) ? -1.0f : (i == 3) ? 8.0f : (i == 4) ? -8.0f : (i == 5) ? 1.0f : 0.0f;
                    f32x4 rv[3][2];
                    if (!(W4_ABL & 2)) {
                        __syncthreads();                    // every wave is done reading the slot
                        W4_ST(6)
#pragma unroll
                        for (int jj = 0; jj < 3; jj++)
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const f32x4 v_ = {acc[jj][8 + 4 * h], acc[jj][9 + 4 * h], acc[jj][10 + 4 * h], acc[jj][11 + 4 * h]};
                                *reinterpret_cast<f32x4 *>(xw + (2 * jj + h) * 256) = v_;
                            }
                        __syncthreads();
                    }
#pragma unroll
                    for (int jj = 0; jj < 3; jj++)
#pragma unroll
                        for (int h = 0; h < 2; h++) rv[jj][h] = *reinterpret_cast<const f32x4 *>(xr + (2 * jj + h) * 256);
                    // the pass's closing barrier, here: X is read (the next step's DMA may overwrite it) and the next step's slabs have landed; the
                    // transform below runs into the next step's MFMAs without another meeting of the waves
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    __syncthreads();
                    W4_ST(7)
                    // R[b] = sum_j A^T[b][j] M[i][j] (the two kinds of wave differ in which half of the row is their own: a branch around this piece only --
                    // around the Y update as well it costs hipcc 259 spilled registers), then Y[a][b] += A^T[a][i] R[b], two tiles at a time
#define W4_ROW(LO_, HI_)                                                                                 \
    _Pragma("unroll") for (int k2 = 2 * h2; k2 < 2 * h2 + 2; k2++)                                       \
    {                                                                                                    \
        const f32x2 m0 = LO_(0, k2), m1 = LO_(1, k2), m2 = LO_(2, k2), m3 = HI_(0, k2), m4 = HI_(1, k2), m5 = HI_(2, k2); \
        const f32x2 sp = m1 + m2, dp = m1 - m2, p0 = m0 + sp, sq = m3 + m4, dq = m3 - m4;                \
        R[k2][0] = p0 + sq;                                                                              \
        R[k2][1] = w4_fma2(2.0f, dq, dp);                                                                \
        R[k2][2] = w4_fma2(4.0f, sq, sp);                                                                \
        R[k2][3] = w4_fma2(8.0f, dq, dp) + m5;                                                           \
    }
#define W4_OWN(JJ_, K2_) (f32x2{acc[JJ_][2 * (K2_)], acc[JJ_][2 * (K2_) + 1]})
#define W4_GOT(JJ_, K2_) (f32x2{rv[JJ_][(K2_) >> 1][2 * ((K2_) & 1)], rv[JJ_][(K2_) >> 1][2 * ((K2_) & 1) + 1]})
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                    f32x2 R[4][4];
                    if (jh == 0) { W4_ROW(W4_OWN, W4_GOT) }
                    else { W4_ROW(W4_GOT, W4_OWN) }
#pragma unroll
                    for (int k2 = 2 * h2; k2 < 2 * h2 + 2; k2++) {
                        if (W4_ABL & 8) { y[0][k2] += R[k2][0] + R[k2][1] + R[k2][2] + R[k2][3]; continue; }
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            y[b][k2] = w4_fma2(c0, R[k2][b], y[b][k2]);
                            y[4 + b][k2] = w4_fma2(c1, R[k2][b], y[4 + b][k2]);
                            y[8 + b][k2] = w4_fma2(c2, R[k2][b], y[8 + b][k2]);
                            y[12 + b][k2] = w4_fma2(c3, R[k2][b], y[12 + b][k2]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    }
#undef W4_ROW
#undef W4_OWN
#undef W4_GOT
                }
#ifdef W4_STAMPS
                if (kc == KC - 1) { W4_ST(2) st4[5]++; }
                st4[4]++;
#endif
                if (kc != KC - 1) {                      // (a pass's last step has met above)
                    // the next step's slabs have landed (this step's six may be in flight); step 1 of a unit was waited for before the last unit's stores
                    if (s != 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    __syncthreads();
                }
                W4_ST(1)
                slot = slot == 2 ? 0 : slot + 1;
            }
        }
        // ---- ReLU, store: a lane holds one channel of eight tiles, a store instruction writes 32 consecutive channels (128 B) of two pixels.  The pixel's
        //      place in the tile is a scalar offset; a pixel outside the map turns the lane's offset out of range ----
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (long long)b0 * H * W * Cout), 0, 0x40000000, 0x00020000);
        const unsigned nb4 = 4u * (unsigned)n;
        // every slab asked for so far (steps 0 and 1 of the next unit) has landed before the first store is issued: loads and stores do not retire in
        // one order, and a count that has to cover both would make the next unit's first steps wait for the stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int tile = wm * 32 + (((k >> 2) * 8 + 4 * half + (k & 3)) ^ (jh << 4));
            const u32x2 om = *reinterpret_cast<const u32x2 *>(&tbl[tb * 128 + 2 * tile]);
            const unsigned voff = om.x + nb4;           // an absent tile's offset (OOB) stays out of range
            const int inv = (int)~om.y;
            if (__builtin_amdgcn_ballot_w64(inv != 0) == 0) {          // both tiles of the instruction whole (the common case): no per-pixel work
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const int so = 4 * ((p >> 2) * W + (p & 3)) * Cout;
                    if (!(W4_ABL & 4) || y[p][k >> 1][k & 1] == 12345.678f)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, wf_relu(y[p][k >> 1][k & 1])), rsO, voff, so, APR_WF_STORE_AUX);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const unsigned o = voff | (unsigned)__builtin_amdgcn_sbfe(inv, p, 1);     // bit p of the mask clear: all ones
                    const int so = 4 * ((p >> 2) * W + (p & 3)) * Cout;
                    if (!(W4_ABL & 4) || y[p][k >> 1][k & 1] == 12345.678f)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, wf_relu(y[p][k >> 1][k & 1])), rsO, o, so, APR_WF_STORE_AUX);
                }
            }
        }
#ifdef W4_STAMPS
#endif
        if (!more) break;
        bm = bm_n; bn = bn_n; tb ^= 1; lidl = nl;
        rsV.x = rsVn.x; rsV.y = rsVn.y; rsU.x = rsUn.x; rsU.y = rsUn.y;
    }
#ifdef W4_STAMPS
    W4_ST(3)
    if (blockIdx.x == 0 && lane == 0)
        for (int q = 0; q < 8; q++) w4_stamps[wave][q] += st4[q];
#endif
#undef W4_DESC
#undef W4_DMA
#undef W4_DMA1
}

}  // namespace apr
