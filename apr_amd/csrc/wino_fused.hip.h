// wino_fused.hip.h -- Winograd F(2x2, 3x3) convolution in ONE kernel (stride 1, pad 1, NHWC, fp32 MFMA).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        (Lavin & Gray)
//
// wino.hip.h + the batched GEMM keep V = B^T d B and M = U .* V in HBM (16 floats per 4 inputs /
// outputs): at 64..128 channels that round trip costs more than the 2.25x fewer multiplies save.
// Here neither leaves the CU:
//   * input transform in registers between the global load and the LDS store of the A operand;
//   * 16 GEMMs (one per transform position) accumulate side by side in one wave:
//     16 positions x (32 tiles x 32 channels) = 16 MFMA tiles = 256 accumulator registers, so
//     after the K loop every lane holds all 16 positions of its (tile, channel) pairs and the
//     output transform (+ bias, ReLU) is lane-local arithmetic before the NHWC store.
// Block = 4 waves as 2 (tiles) x 2 (channels) = 64 tiles (256 output pixels) x 64 channels, one
// block per CU (512 registers per lane, 128.5 KiB of LDS).  K step = 8 input channels:
//   Vs[pos][tile][8]  (32 KiB)  written by the transform, 8 B per lane per position
//   Us[pos][cout][8]  (32 KiB)  copied global -> LDS by LDS-DMA; the host stores U in exactly this
//                                image order (wf_pack_u), so a K step's slice is one contiguous 32 KiB
// double buffered, one barrier per K step.  One ds_read_b128 per operand feeds four
// v_mfma_f32_32x32x2_f32 (MFMA j multiplies channels j and 4+j of the chunk).
#pragma once
#include <hip/hip_runtime.h>

#include "conv.hip.h"

namespace apr {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef APR_WF_PK
#define APR_WF_PK 1
#endif
// f32x2 add / subtract as ONE packed instruction (left to itself hipcc splits most of them into two v_add_f32)
__device__ __forceinline__ f32x2 wf_add2(f32x2 a, f32x2 b)
{
#if APR_WF_PK
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a + b;
#endif
}
__device__ __forceinline__ f32x2 wf_sub2(f32x2 a, f32x2 b)
{
#if APR_WF_PK
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - b;
#endif
}

// -a - b
__device__ __forceinline__ f32x2 wf_nadd2(f32x2 a, f32x2 b)
{
#if APR_WF_PK
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,1] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return -a - b;
#endif
}
// max(x, 0) as one instruction (fmaxf on a value out of inline asm costs a canonicalising v_max first); the sums
// that reach it are never NaN-signalling, and v_max_f32 returns the other operand for a NaN like fmaxf
__device__ __forceinline__ float wf_relu(float x)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef APR_WF_STORE_AUX
#define APR_WF_STORE_AUX 0   // 2 = nt (streaming) measured neutral here
#endif
__device__ __forceinline__ void wf_store1(__amdgpu_buffer_rsrc_t rs, unsigned voff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, 0, APR_WF_STORE_AUX);
}

constexpr int WF_TT = 64;                       // tiles per block
constexpr int WF_BN = 64;                       // output channels per block
constexpr int WF_BK = 8;                        // input channels per K step
constexpr int WF_POS = 16 * 64 * WF_BK;         // floats of one operand image (all 16 positions)
constexpr int WF_LDS_BYTES = 2 * 2 * WF_POS * 4 + WF_TT * 8;

// developer knobs (timing-only ablations in tools/wino_fused_bench.cpp; all 1 in the product)
#ifndef APR_WF_EXP_DMA
#define APR_WF_EXP_DMA 1
#endif
#ifndef APR_WF_EXP_LOADV
#define APR_WF_EXP_LOADV 1
#endif
#ifndef APR_WF_EXP_STOREV
#define APR_WF_EXP_STOREV 1
#endif
#ifndef APR_WF_CLUMP
#define APR_WF_CLUMP 1
#endif

#ifdef APR_WF_STAMPS   // developer build (tools/wino_fused_bench.cpp): s_memtime stamps of the first blocks
__device__ unsigned long long wf_stamps[4096][8];
#define WF_STAMP(I_) if (blockIdx.x < 4096 && threadIdx.x == 0) wf_stamps[blockIdx.x][I_] = __builtin_amdgcn_s_memtime();
#else
#define WF_STAMP(I_)
#endif

// Division by the launch's small constants without the ~30-instruction expansion per quotient:
// q = (n * ceil(2^40 / d)) >> 40, exact while q < 2^24 and n < 2^31 (host: wf_magic).
struct WfMagic {
    unsigned long long tiles_n, tpw, tw, th;
    unsigned long long upg_m, gn_m;   // the persistent kernels' unit order (wf_unit): units per N group, N tiles per group
    int upg, gn;
    unsigned tw16, th16;              // ceil(2^16 / tw), ceil(2^16 / th): divisions of numbers below 2^16 / d (wf_tile)
};
__host__ __device__ __forceinline__ int wf_div(int n, unsigned long long m) { return (int)(((unsigned long long)(unsigned)n * m) >> 40); }
// Unit order of the persistent kernels: N tiles in groups of gn; within a group the units of one M tile follow each other, so
// the workgroups of an XCD, which walk consecutive units, hold per / gn M tiles x gn N tiles at a time -- gn trades re-reads of the
// input (once per group) against re-reads of U (once per per / gn concurrent units of an N tile).  gn = tiles_n: unit = bm * tiles_n + bn.
__host__ __device__ __forceinline__ void wf_unit(int lid, const WfMagic &mg, int &bm, int &bn)
{
    const int grp = wf_div(lid, mg.upg_m);
    const int r = lid - grp * mg.upg;
    bm = wf_div(r, mg.gn_m);
    bn = grp * mg.gn + (r - bm * mg.gn);
}
// Tile g = (bb * th + ty) * tw + tx of a work unit whose first tile is g0: the unit's first tile is split once (wave-uniform,
// 40-bit magic numbers), every lane then adds its small tile index t and carries with two 24-bit multiplies
// (exact while (tw + 64) * tw and (th + 64) * th stay below 2^16: tiles per unit <= 64, th, tw <= WF_MAX_TDIM; host-checked).
constexpr int WF_MAX_TDIM = 200;
struct WfUnitBase { int tx0, ty0, bb0; };
__host__ __device__ __forceinline__ WfUnitBase wf_unit_base(int g0, const WfMagic &mg, int th, int tw)
{
    const int rr0 = wf_div(g0, mg.tw), bb0 = wf_div(rr0, mg.th);
    return WfUnitBase{g0 - rr0 * tw, rr0 - bb0 * th, bb0};
}
__host__ __device__ __forceinline__ void wf_tile(const WfUnitBase &ub, int t, const WfMagic &mg, int th, int tw, int &bb, int &ty, int &tx)
{
    const unsigned x = (unsigned)(ub.tx0 + t) & 0xFFFFu;
    const unsigned dq = (x * (mg.tw16 & 0x1FFFFu)) >> 16;
    tx = (int)(x - dq * ((unsigned)tw & 0xFFFFu));
    const unsigned y = ((unsigned)ub.ty0 + dq) & 0xFFFFu;
    const unsigned db = (y * (mg.th16 & 0x1FFFFu)) >> 16;
    ty = (int)(y - db * ((unsigned)th & 0xFFFFu));
    bb = ub.bb0 + (int)db;
}
// tiles_m has no default on purpose: a unit order built for another M extent sends units to N tiles that do not exist
inline WfMagic wf_magic(int tiles_n, int th, int tw, long long tiles_m, int gn = 0)
{
    auto mk = [](long long d) { return (unsigned long long)(((1ull << 40) + (unsigned long long)d - 1) / (unsigned long long)d); };
    if (gn < 1 || gn > tiles_n) gn = tiles_n;
    while (tiles_n % gn) gn--;           // a divisor of tiles_n
    const long long upg = tiles_m * gn;
    return WfMagic{mk(tiles_n), mk((long long)th * tw), mk(tw), mk(th), mk(upg), mk(gn), (int)upg, gn, (65536u + (unsigned)tw - 1) / (unsigned)tw, (65536u + (unsigned)th - 1) / (unsigned)th};
}

// in  [B][H][W][Cin] ; up = packed U (wf_pack_u) ; out [B][H][W][Cout] = relu(conv + bias)
// T = B * th * tw tiles, th = ceil(H/2), tw = ceil(W/2).  Requires Cin % 16 == 0, Cout % 64 == 0.
__global__ __launch_bounds__(256) void wino_fused_f2(const float *__restrict__ in, const float *__restrict__ up,
                                                     const float *__restrict__ bias, float *__restrict__ out, int H,
                                                     int W, int Cin, int Cout, int th, int tw, int T, int tiles_n, WfMagic mg)
{
    extern __shared__ __attribute__((aligned(16))) float wf_smem[];
    float *Vs = wf_smem;                        // [2][WF_POS]
    float *Us = wf_smem + 2 * WF_POS;           // [2][WF_POS]
    int *tbl = reinterpret_cast<int *>(wf_smem + 4 * WF_POS);   // [64][2]: output pixel of (b, 2ty, 2tx) or -1; flags

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, half = lane >> 5;
    WF_STAMP(0)
#ifdef APR_WF_STAMPS
    if (blockIdx.x < 4096 && threadIdx.x == 0) wf_stamps[blockIdx.x][6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
#endif

    // XCD-aware block order (see conv3x3_igemm_f32)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bm = wf_div(lid, mg.tiles_n);
    const int bn = lid - bm * tiles_n;

    const int g0 = bm * WF_TT;                   // first tile of the block (< T)
    const int b0 = wf_div(g0, mg.tpw);           // its window

    // ---- input patch addressing: buffer loads, zero padding through out-of-range offsets ----
    constexpr unsigned OOB = 0x80000000u;
    const float *abase = in + (((long long)b0 * H - 1) * W - 1) * (long long)Cin;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)abase, 0, 0x40000000, 0x00020000);
    const int t_ld = tid >> 2, cp = tid & 3;     // this thread transforms tile t_ld, channels 2cp, 2cp+1 of the chunk
    unsigned voff = OOB, vmask = 0;
    {
        int tpx = -1, tfl = 0;                   // the tile's entry of the epilogue's table, from the same divisions
        const int g = g0 + t_ld;
        if (g < T) {
            const int rr = wf_div(g, mg.tw);
            const int tx = g - rr * tw;
            const int bb = wf_div(rr, mg.th);
            const int ty = rr - bb * th;
            const int db = bb - b0;
            const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
            voff = 4u * (unsigned)(((db * H + y0 + 1) * W + x0 + 1) * Cin + 2 * cp);
            tpx = (db * H + 2 * ty) * W + 2 * tx;   // relative to the block's first window
            tfl = (2 * ty + 1 < H ? 1 : 0) | (2 * tx + 1 < W ? 2 : 0);
            // patch rows 2ty-1 .. 2ty+2 and columns 2tx-1 .. 2tx+2 inside the image (rows/columns 1 always are)
            const unsigned rowok = (ty > 0 ? 1u : 0u) | 2u | (2 * ty + 1 < H ? 4u : 0u) | (2 * ty + 2 < H ? 8u : 0u);
            const unsigned colok = (tx > 0 ? 1u : 0u) | 2u | (2 * tx + 1 < W ? 4u : 0u) | (2 * tx + 2 < W ? 8u : 0u);
#pragma unroll
            for (int i = 0; i < 4; i++) vmask |= ((rowok >> i) & 1u) ? colok << (4 * i) : 0u;
        }
        if (cp == 0) {
            tbl[2 * t_ld] = tpx;
            tbl[2 * t_ld + 1] = tfl;
        }
    }
    // ---- U: descriptor words for the LDS-DMA; a K step's image is contiguous (32 KiB) ----
    const int KC = Cin / WF_BK;
    const unsigned long long ub = (unsigned long long)(up + (size_t)bn * KC * WF_POS);
    i32x4 rsU;
    rsU.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ub);
    rsU.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(ub >> 32) & 0xFFFF);
    rsU.z = 0x7FFFFFFF;
    rsU.w = 0x00020000;
    const unsigned us_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(Us + wave * (WF_POS / 4)));
    const unsigned uvoff = (unsigned)(wave * (WF_POS / 4) * 4 + lane * 16);

    // Two register sets for the raw 4x4 patches: while set DC (chunk kc+1, loaded during the previous
    // step) is transformed and stored, set DN receives chunk kc+2 -- a full K step of load latency.
    f32x2 dA[16], dB[16];
    // per-lane byte offsets of the 16 patch pixels, fixed for the whole kernel (padding pixels stay out of
    // range); a K step then only moves the scalar offset, and a load costs no VALU at all
    unsigned vo[16];
#pragma unroll
    for (int ij = 0; ij < 16; ij++)
        vo[ij] = ((vmask >> ij) & 1u) ? voff + 4u * (unsigned)(((ij >> 2) * W + (ij & 3)) * Cin) : OOB;
#define WF_LOAD_V1(D_, KC_, IJ_) D_[IJ_] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, APR_WF_EXP_LOADV == 2 ? (vo[IJ_] | OOB) : vo[IJ_], 4 * (KC_) * WF_BK, 0));
#define WF_DMA_U1(KC_, BUF_, I_) lds_dma_b128(rsU, uvoff, (KC_) * (WF_POS * 4) + (I_) * 1024, us_lds + (unsigned)((BUF_) * WF_POS * 4 + (I_) * 1024));
#define WF_DMA_U(KC_, BUF_)                                                                              \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; i++) WF_DMA_U1(KC_, BUF_, i)                            \
    }
    // rows: d <- B^T d, in place, column J_ of the patch in two halves (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1])
#define WF_ROWS_HALF(D_, J_, H_)                                                                         \
    {                                                                                                    \
        if ((H_) == 0) {                                                                                 \
            D_[J_] = wf_sub2(D_[J_], D_[8 + (J_)]);                                                      \
            D_[12 + (J_)] = wf_sub2(D_[4 + (J_)], D_[12 + (J_)]);                                        \
        } else {                                                                                         \
            const f32x2 r1_ = wf_add2(D_[4 + (J_)], D_[8 + (J_)]), r2_ = wf_sub2(D_[8 + (J_)], D_[4 + (J_)]); \
            D_[4 + (J_)] = r1_;                                                                          \
            D_[8 + (J_)] = r2_;                                                                          \
        }                                                                                                \
    }
    // column transform of row I_, position 4 I_ + Q_, and its LDS store
#define WF_STORE_POS(D_, BUF_, I_, Q_)                                                                   \
    {                                                                                                    \
        const f32x2 v_ = APR_WF_EXP_STOREV == 3 ? D_[4 * (I_) + (Q_)]                                    \
                         : (Q_) == 0 ? wf_sub2(D_[4 * (I_)], D_[4 * (I_) + 2])                           \
                         : (Q_) == 1 ? wf_add2(D_[4 * (I_) + 1], D_[4 * (I_) + 2])                       \
                         : (Q_) == 2 ? wf_sub2(D_[4 * (I_) + 2], D_[4 * (I_) + 1])                       \
                                     : wf_sub2(D_[4 * (I_) + 1], D_[4 * (I_) + 3]);                      \
        if (APR_WF_EXP_STOREV == 2) asm volatile("" ::"v"(v_));                                          \
        else *reinterpret_cast<f32x2 *>(Vs + (BUF_) * WF_POS + (4 * (I_) + (Q_)) * (64 * WF_BK) + st_off) = v_; \
    }
    // LDS row = 8 channels of one tile (or cout) = two 16-B halves; rows 16..31 of every 32 keep their
    // halves swapped, which makes the four 16-lane groups of ds_read_b128 conflict-free (wf_pack_u
    // stores U the same way).
    const int st_off = t_ld * WF_BK + (((cp >> 1) ^ ((t_ld >> 4) & 1)) * 4) + (cp & 1) * 2;
    const int rd_sw = (half ^ ((li >> 4) & 1)) * 4;
    const int rd_v = (wm * 32 + li) * WF_BK + rd_sw;
    const int rd_u = (wn * 32 + li) * WF_BK + rd_sw;

    f32x16 acc[16];

    f32x4 af[2][2], bf[2][2];
    // one operand (Q_ = 0: A of position 2G, 1: B of 2G, 2: A of 2G+1, 3: B of 2G+1) from LDS buffer BUF_ into set SET_
#define WF_READ_OP(SET_, BUF_, G_, Q_)                                                                   \
    {                                                                                                    \
        if (((Q_) & 1) == 0)                                                                             \
            af[SET_][(Q_) >> 1] = *reinterpret_cast<const f32x4 *>(Vs + (BUF_) * WF_POS + rd_v + (2 * (G_) + ((Q_) >> 1)) * (64 * WF_BK)); \
        else                                                                                             \
            bf[SET_][(Q_) >> 1] = *reinterpret_cast<const f32x4 *>(Us + (BUF_) * WF_POS + rd_u + (2 * (G_) + ((Q_) >> 1)) * (64 * WF_BK)); \
    }

    WF_STAMP(1)
    WF_DMA_U(0, 0)
#pragma unroll
    for (int ij = 0; ij < 16; ij++) WF_LOAD_V1(dB, 0, ij)
#pragma unroll
    for (int ij = 0; ij < 16; ij++) WF_LOAD_V1(dA, (KC > 1 ? 1 : 0), ij)
    // clearing the 256 accumulators is 1k cycles of VALU: here, under the latency of the first loads
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 16; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[p][e] = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; j++) { WF_ROWS_HALF(dB, j, 0) WF_ROWS_HALF(dB, j, 1) }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) WF_STORE_POS(dB, 0, i, qq)
    if (!(APR_WF_EXP_DMA && APR_WF_EXP_LOADV && APR_WF_EXP_STOREV == 1)) {   // ablation builds: real data in both buffers
        WF_DMA_U(0, 1)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) WF_STORE_POS(dB, 1, i, qq)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the DMA and V(0) have landed; the 16 loads of V(1) may still fly
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < 4; qq++) WF_READ_OP(0, 0, 0, qq)

    // K step kc = 64 slots of one MFMA each (8 groups g of two positions x 4 k-pairs); every slot is
    // closed by a full scheduling barrier, so the non-MFMA work below stays where it is written, a
    // slot's worth at a time, in the shadow of the MFMAs (hipcc left alone clumps it between them):
    //   all g     slots 0-3  operand reads of group g+1 (g = 7: of the NEXT step, after the barrier)
    //   g = 0     row transform of DC (chunk kc+1, loaded during the previous step), 2 VALU per slot
    //   g = 1     LDS-DMA of U(kc+1), one 1-KiB piece per slot -- after the last use of hipcc-counted
    //             loads, so no counted vmcnt wait ever covers a DMA that has just been issued
    //   g = 2     buffer loads of chunk kc+2 into DN, two per slot
    //   g = 3-6   column transform + LDS store of V(kc+1): row g-3, one position per odd slot
    //   g = 7     slot 0: wait for the DMA (the 16 younger V loads stay in flight), barrier
    // Chunk indices past the end are clamped: the last steps stage data nobody reads, which keeps the
    // loop free of special cases (one accumulator register assignment from loop to epilogue).
#define WF_KSTEP(DC, DN)                                                                                 \
    {                                                                                                    \
        const int kn1 = kc + 1 < KC ? kc + 1 : KC - 1;                                                   \
        const int kn2 = kc + 2 < KC ? kc + 2 : KC - 1;                                                   \
        _Pragma("unroll") for (int g = 0; g < 8; g++)                                                    \
        {                                                                                                \
            _Pragma("unroll") for (int i = 0; i < 8; i++)                                                \
            {                                                                                            \
                if (g == 7 && i == 0) {                                                                  \
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                    \
                    __syncthreads();                                                                     \
                }                                                                                        \
                if (i < 4) {                                                                             \
                    if (g < 7) WF_READ_OP((g + 1) & 1, cur, g + 1, i)                                    \
                    else WF_READ_OP(0, cur ^ 1, 0, i)                                                    \
                }                                                                                        \
                if (!APR_WF_CLUMP && APR_WF_EXP_STOREV && APR_WF_EXP_STOREV != 3 && g == 0) WF_ROWS_HALF(DC, i >> 1, i & 1) \
                if (APR_WF_CLUMP && APR_WF_EXP_STOREV && APR_WF_EXP_STOREV != 3 && g == 0 && i == 0) {   \
                    _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) { WF_ROWS_HALF(DC, j_, 0) WF_ROWS_HALF(DC, j_, 1) } \
                }                                                                                        \
                if (APR_WF_EXP_DMA && g == 0 && i >= 1) WF_DMA_U1(kn1, cur ^ 1, i - 1)                   \
                if (APR_WF_EXP_DMA && g == 0 && i == 7) WF_DMA_U1(kn1, cur ^ 1, 7)                       \
                if (APR_WF_EXP_LOADV && (g == 1 || g == 2)) WF_LOAD_V1(DN, kn2, 8 * (g - 1) + i)         \
                if (!APR_WF_CLUMP && APR_WF_EXP_STOREV && g >= 3 && g < 7 && (i & 1)) WF_STORE_POS(DC, cur ^ 1, g - 3, i >> 1) \
                if (APR_WF_CLUMP && APR_WF_EXP_STOREV && g >= 3 && g < 7 && i == 1) {                    \
                    _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) WF_STORE_POS(DC, cur ^ 1, g - 3, q_) \
                }                                                                                        \
                acc[2 * g + (i & 1)] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i & 1][i >> 1], bf[g & 1][i & 1][i >> 1], acc[2 * g + (i & 1)], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                       \
            }                                                                                            \
        }                                                                                                \
        cur ^= 1;                                                                                        \
    }

    WF_STAMP(2)
    int cur = 0;
    for (int kc = 0; kc < KC; kc += 2) {   // KC is even (Cin % 16 == 0, host-checked)
        WF_KSTEP(dA, dB)
        kc++;
        WF_KSTEP(dB, dA)
        kc--;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail loads / DMA land before the block ends
    WF_STAMP(3)
#undef WF_KSTEP
#undef WF_READ_OP
#undef WF_LOAD_V1
#undef WF_DMA_U
#undef WF_DMA_U1
#undef WF_ROWS_HALF
#undef WF_STORE_POS

    // ---- epilogue: Y = A^T m A per (tile, channel), + bias, ReLU, NHWC store.
    //      acc[pos][e] is D[tile row][channel col]: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5);
    //      A^T = [1 1 1 0; 0 1 -1 -1] ----
    const int n = bn * WF_BN + wn * 32 + li;
    const float bv = bias[n];
    // output descriptor based at the block's first window: offsets stay 32-bit
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (long long)b0 * H * W * Cout), 0, 0x40000000, 0x00020000);
    const unsigned nb4 = 4u * (unsigned)n;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int tl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const int px = tbl[2 * tl], fl = tbl[2 * tl + 1];
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t0[j] = acc[j][e] + acc[4 + j][e] + acc[8 + j][e];
            t1[j] = acc[4 + j][e] - acc[8 + j][e] - acc[12 + j][e];
        }
        const float y00 = fmaxf(t0[0] + t0[1] + t0[2] + bv, 0.0f);
        const float y01 = fmaxf(t0[1] - t0[2] - t0[3] + bv, 0.0f);
        const float y10 = fmaxf(t1[0] + t1[1] + t1[2] + bv, 0.0f);
        const float y11 = fmaxf(t1[1] - t1[2] - t1[3] + bv, 0.0f);
        // branch-free: pixels that do not exist get an out-of-range offset and the buffer unit drops them
        const unsigned o0 = 4u * (unsigned)(px * Cout) + nb4;
        wf_store1(rsO, px >= 0 ? o0 : OOB, y00);
        wf_store1(rsO, px >= 0 && (fl & 2) ? o0 + 4u * (unsigned)Cout : OOB, y01);
        wf_store1(rsO, px >= 0 && (fl & 1) ? o0 + 4u * (unsigned)(W * Cout) : OOB, y10);
        wf_store1(rsO, px >= 0 && (fl & 3) == 3 ? o0 + 4u * (unsigned)((W + 1) * Cout) : OOB, y11);
    }
    WF_STAMP(4)
#ifdef APR_WF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    WF_STAMP(5)
}

// Host: g [Cout][Cin][3][3] (OIHW, BN folded) -> packed U for wino_fused_f2:
//   up[((bn * KC + kc) * 16 + pos) * 64 * 8 + nl * 8 + c] = (G g G^T)[pos] of (cout = 64 bn + nl, cin = 8 kc + c),
//   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], evaluated in f64 and rounded once.
inline void wf_pack_u(const float *g, int Cin, int Cout, float *up)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int KC = Cin / WF_BK;
    for (int o = 0; o < Cout; o++)
        for (int c = 0; c < Cin; c++) {
            const float *w = g + ((size_t)o * Cin + c) * 9;
            double t[4][3];
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 3; k++) t[i][k] = G[i][0] * w[k] + G[i][1] * w[3 + k] + G[i][2] * w[6 + k];
            const int nl = o % WF_BN, cl = c % WF_BK;
            const size_t base = ((size_t)(o / WF_BN) * KC + c / WF_BK) * WF_POS + (size_t)nl * WF_BK + (((cl >> 2) ^ ((nl >> 4) & 1)) * 4) + (cl & 3);
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                    up[base + (size_t)(4 * i + j) * (64 * WF_BK)] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}

}  // namespace apr
