// wino_fused8.hip.h -- wino_fused_f2 with TWO waves per SIMD.
//
// Same work unit (64 tiles x 64 channels, 16 transform positions, K step = 8 input channels, same
// LDS images) as wino_fused.hip.h, but 8 waves per workgroup: the two waves of a pair own the same
// 32 tiles x 32 channels and split the 16 positions 8 / 8 -- 128 accumulator registers each, so two
// waves fit on a SIMD (256 registers per lane each) and one wave's LDS / barrier / memory waits are
// filled by the other's MFMAs.  The staging work alternates between the halves: in every K step
// the waves of one parity issue the patch loads of chunk k+2 while the waves of the other parity
// transform and store chunk k+1 and DMA its U image (one register set per wave, no wave ever has
// loads and a DMA in flight together, so every vmcnt wait is exact).
// Workgroups are persistent (one per CU) and the staging pipeline runs on across the unit boundary: the
// last two K steps of a unit, which have no chunk of their own left to fetch, load, transform, store
// and DMA chunks 0 and 1 of the workgroup's next unit (Cin >= 48: KC even and >= 6), so between two K
// loops only the epilogue and two barriers are left (-DW8_USTAMPS: 9.4 k -> 4.5 k cycles a unit).
// Epilogue: the output transform is linear in the positions, so each wave reduces its 8 positions
// to partial 2x2 outputs, the pair swaps halves through LDS (the idle Vs buffers) and each wave
// finishes and stores two of the four pixels; the exchange runs through Vs[1] and Us[1], the buffers
// the last K step read (Vs[0] / Us[0] hold the next unit's chunk 0 by then).
#pragma once
#include <hip/hip_runtime.h>

#include "wino_fused.hip.h"

namespace apr {

#ifndef W8_ABL
#define W8_ABL 0   // developer timing ablations (tools/wino_fused_bench.cpp): 1 no patch loads, 2 no transform/stores, 4 no U DMA, 8 no barrier
#endif

#ifdef W8_USTAMPS  // developer build (tools/wino_fused_bench.cpp): s_memtime per wave of workgroup 0, summed over its units: [0] next unit's set-up +
                   // first loads, [1] first half of the epilogue, [2] its barrier, [3] second half + stores issued, [4] end-of-unit barrier,
                   // [5] prologue transform + stores, [6] prologue barrier, [7] K loop
__device__ unsigned long long w8_ustamps[8][10];   // [8] the set-up part of [0]
#define W8_US(I_) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); us8[I_] += t_ - ust; ust = t_; }
#define W8_UK(I_) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); uk8[I_] += t_ - ukt; ukt = t_; }
__device__ unsigned long long w8_ksteps[8][8];     // cycles per K step: the two first, the loop's two (summed), the last four
#else
#define W8_US(I_)
#define W8_UK(I_)
#endif
#ifdef W8_TIMES    // developer build (tools/wino_fused_bench.cpp): s_memrealtime (100 MHz, one clock for the chip) at the first and last instruction of every workgroup
__device__ unsigned long long w8_times[1024][2];
#endif
#ifdef W8_STAMPS   // developer build (tools/ws_stamps.py): s_memtime stamps of workgroup 0, per wave: cycles waited at the barrier of
                   // its load steps [0] / store steps [1], K loop total [6], steps [7]
__device__ unsigned long long w8_stamps[8][8];
#endif

constexpr int WF8_LDS_BYTES = 2 * 2 * WF_POS * 4 + 2 * WF_TT * 16;   // Vs[2], Us[2], two tile tables

// grid of the persistent kernel: one workgroup per CU, a multiple of 8 (one share per XCD), no more than the units need
inline unsigned wf8_grid(int n_units, int n_cus)
{
    const int cap = n_cus >= 8 ? (n_cus / 8) * 8 : 8;
    return (unsigned)(n_units >= cap ? cap : ((n_units + 7) / 8) * 8);
}

__global__ __launch_bounds__(512) void wino_fused8_f2(const float *__restrict__ in, const float *__restrict__ up,
                                                      const float *__restrict__ bias, float *__restrict__ out, int H,
                                                      int W, int Cin, int Cout, int th, int tw, int T, int tiles_n, WfMagic mg,
                                                      int n_units)
{
    extern __shared__ __attribute__((aligned(16))) float wf_smem[];
    float *Vs = wf_smem;                        // [2][WF_POS]; after the K loop: the pairs' exchange buffers
    float *Us = wf_smem + 2 * WF_POS;           // [2][WF_POS]
    unsigned *tbl = reinterpret_cast<unsigned *>(wf_smem + 4 * WF_POS);   // [2][64][4]: byte offsets of a tile's four output pixels (channel 0), out of range if absent (one table per unit in flight)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
#ifdef W8_TIMES
    if (tid == 0) w8_times[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif
    const int phs = __builtin_amdgcn_readfirstlane(wave & 1);    // ph as a scalar: the epilogue branches on it instead of selecting per lane
    // Waves w and w + 4 share a SIMD, and the SIMD serves the older one (w < 4) first: its staging instructions run beside
    // the other wave's MFMAs, the other wave's own are exposed (wino_s2_fused.hip.h, tools/ws_stamps.py).  The U DMA is
    // therefore the work of waves 0-3 alone, a quarter of the image each in every step.
    // Stamps of the K step (-DW8_STAMPS, tools/ws_stamps.py): 4.8 k cycles against 4.1 k of MFMA issue.  The SIMDs of the
    // step's store role are busy throughout: 48 MFMAs up to the barrier + 64 v_pk_add_f32 + 32 ds_write_b64 (their
    // VGPR transfer) + the DMA issue = 3.65 k cycles against 2.0-2.3 k on the SIMDs of the load role, which then wait at the
    // barrier: fp32 VALU and LDS stores take MFMA issue time one for one.  Measured and dropped: opposite roles on the two
    // waves of a SIMD (the transform work spread over all four SIMDs): the load-role wave takes the matrix pipe first, the
    // store-role wave runs after it, 1 % slower; with s_setprio 2 on the store-role wave: equal to this form.
    const int dma_wave = __builtin_amdgcn_readfirstlane(wave < 4 ? 1 : 0);
    const int ph = wave & 1;                    // which 8 of the 16 positions; also the staging parity
    const int sub = wave >> 1;                  // the pair: 2 (tiles) x 2 (channels)
    const int subs = __builtin_amdgcn_readfirstlane(wave >> 1);
    const int wm = sub >> 1, wn = sub & 1;
    const int li = lane & 31, half = lane >> 5;
    const int sid = sub * 64 + lane;            // 0..255 among the four waves of one parity
    const int t_ld = sid >> 2, cp = sid & 3;    // every parity stages whole chunks: 256 threads = 64 tiles x 4 channel pairs
    WF_STAMP(0)

    // Persistent workgroups, XCD-aware: the units are cut into 8 contiguous runs, one per XCD (neighbouring
    // units share input rows or weights and so an L2); the gridDim.x / 8 workgroups of an XCD walk its run.
    const int per = gridDim.x >> 3, xcd = blockIdx.x & 7;       // gridDim.x is a multiple of 8 (host)
    const int uq = n_units >> 3, ur = n_units & 7;
    const int x_start = xcd < ur ? xcd * (uq + 1) : ur * (uq + 1) + (xcd - ur) * uq;
    const int x_count = uq + (xcd < ur ? 1 : 0);
    int lidl = blockIdx.x >> 3;
    if (lidl >= x_count) return;                // whole workgroup: nothing to do

    constexpr unsigned OOB = 0x80000000u;
    // unit LID -> its N tile BN_, first window B0_, this lane's patch offset VOFF_ / validity mask VMASK_, and the
    // tile table TB_ (zero padding = out-of-range buffer offsets)
#define W8_SETUP(LID_, TB_, BN_, B0_, VOFF_, VMASK_)                                                     \
    {                                                                                                    \
        /* the lane's tile and channel pair, from scratch: kept in registers across the K loop they are spilled */ \
        unsigned ln_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));               \
        asm volatile("" : "+v"(ln_));                                                                    \
        const int sid_ = subs * 64 + (int)ln_, t_ld_ = sid_ >> 2, cp_ = sid_ & 3;                        \
        int bm_;                                                                                         \
        wf_unit((LID_), mg, bm_, BN_);                                                                   \
        const int g0_ = bm_ * WF_TT;                                                                     \
        B0_ = wf_div(g0_, mg.tpw);                                                                       \
        VOFF_ = OOB; VMASK_ = 0;                                                                         \
        u32x4 to_ = {OOB, OOB, OOB, OOB};                                                                \
        const int g_ = g0_ + t_ld_;                                                                       \
        const WfUnitBase ub_ = wf_unit_base(g0_, mg, th, tw);   /* wave-uniform */                       \
        if (g_ < T) {                                                                                    \
            int bb_, ty_, tx_;                                                                           \
            wf_tile(ub_, t_ld_, mg, th, tw, bb_, ty_, tx_);                                               \
            const int db_ = bb_ - B0_;                                                                   \
            /* the tile's first output pixel, which is also pixel (1, 1) of its patch; every factor is below 2^24 */ \
            const unsigned pix_ = ((unsigned)(db_ * H + 2 * ty_) & 0xFFFFFFu) * ((unsigned)W & 0xFFFFFFu) + 2u * (unsigned)tx_; \
            VOFF_ = 4u * ((pix_ & 0xFFFFFFu) * ((unsigned)Cin & 0xFFFFFFu) + 2u * (unsigned)cp_);         \
            const unsigned o00_ = 4u * ((pix_ & 0xFFFFFFu) * ((unsigned)Cout & 0xFFFFFFu));              \
            const bool r1_ = 2 * ty_ + 1 < H, c1_ = 2 * tx_ + 1 < W;                                     \
            to_.x = o00_;                                                                                \
            to_.y = c1_ ? o00_ + 4u * (unsigned)Cout : OOB;                                              \
            to_.z = r1_ ? o00_ + 4u * (unsigned)(W * Cout) : OOB;                                        \
            to_.w = r1_ && c1_ ? o00_ + 4u * (unsigned)((W + 1) * Cout) : OOB;                           \
            const unsigned rowok_ = (ty_ > 0 ? 1u : 0u) | 2u | (2 * ty_ + 1 < H ? 4u : 0u) | (2 * ty_ + 2 < H ? 8u : 0u); \
            const unsigned colok_ = (tx_ > 0 ? 1u : 0u) | 2u | (2 * tx_ + 1 < W ? 4u : 0u) | (2 * tx_ + 2 < W ? 8u : 0u); \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) VMASK_ |= ((rowok_ >> i_) & 1u) ? colok_ << (4 * i_) : 0u; \
        }                                                                                                \
        if (cp_ == 0 && phs == 0) *reinterpret_cast<u32x4 *>(tbl + (TB_) * 256 + 4 * t_ld_) = to_;          \
    }
    // descriptors of a unit: its window's image (one row and one column early, so every valid tap has a
    // non-negative offset) and its N tile's U image
#define W8_DESCRIPTORS(RSU_, BN_, B0_)                                                                   \
    {                                                                                                    \
        rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (((long long)(B0_) * H - 1) * W - 1) * (long long)Cin), 0, 0x40000000, 0x00020000); \
        const unsigned long long ub_ = (unsigned long long)(up + (size_t)(BN_) * KC * WF_POS);           \
        RSU_.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ub_);                                     \
        RSU_.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(ub_ >> 32) & 0xFFFF);                    \
    }
    const int KC = Cin / WF_BK;
    __amdgpu_buffer_rsrc_t rsA;
    i32x4 rsU, rsU_n;                           // this unit's U image; the next unit's (set up inside the K loop)
    rsU.z = rsU_n.z = 0x7FFFFFFF;
    rsU.w = rsU_n.w = 0x00020000;
    const unsigned us_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(Us + (wave & 3) * (WF_POS / 4)));
    const unsigned uvoff = (unsigned)((wave & 3) * (WF_POS / 4) * 4 + lane * 16);

    f32x2 d[16];
    unsigned vo[16];
#define W8_OFFSETS(VOFF_, VMASK_)                                                                        \
    _Pragma("unroll") for (int ij_ = 0; ij_ < 16; ij_++)                                                 \
        vo[ij_] = (((VMASK_) >> ij_) & 1u) ? (VOFF_) + 4u * (unsigned)(((ij_ >> 2) * W + (ij_ & 3)) * Cin) : OOB;
    // first staging of a unit: waves 0-3 DMA U chunk 0 into Us[0], every parity loads its first V chunk (0 / 1)
#define W8_FIRST_LOADS()                                                                                 \
    {                                                                                                    \
        if (dma_wave) {                                                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) W8_DMA_U1(rsU, 0, 0, i_)                    \
        }                                                                                                \
        _Pragma("unroll") for (int ij_ = 0; ij_ < 16; ij_++) W8_LOAD_V1(phs, ij_)  /* KC >= 2 */          \
    }
#define W8_LOAD_V1(KC_, IJ_) d[IJ_] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, vo[IJ_], 4 * (KC_) * WF_BK, 0));
#define W8_DMA_U1(RSU_, KC_, BUF_, I_) lds_dma_b128(RSU_, uvoff, __builtin_amdgcn_readfirstlane((KC_) * (WF_POS * 4) + (I_) * 1024), __builtin_amdgcn_readfirstlane(us_lds + (unsigned)((BUF_) * WF_POS * 4 + (I_) * 1024)));
#define W8_ROWS()                                                                                        \
    {                                                                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++)                                                 \
        {                                                                                                \
            const f32x2 r0_ = wf_sub2(d[j_], d[8 + j_]), r3_ = wf_sub2(d[4 + j_], d[12 + j_]);           \
            const f32x2 r1_ = wf_add2(d[4 + j_], d[8 + j_]), r2_ = wf_sub2(d[8 + j_], d[4 + j_]);        \
            d[j_] = r0_; d[4 + j_] = r1_; d[8 + j_] = r2_; d[12 + j_] = r3_;                             \
        }                                                                                                \
    }
#define W8_STORE_POS(BUF_, I_, Q_)                                                                       \
    {                                                                                                    \
        const f32x2 v_ = (Q_) == 0   ? wf_sub2(d[4 * (I_)], d[4 * (I_) + 2])                             \
                         : (Q_) == 1 ? wf_add2(d[4 * (I_) + 1], d[4 * (I_) + 2])                         \
                         : (Q_) == 2 ? wf_sub2(d[4 * (I_) + 2], d[4 * (I_) + 1])                         \
                                     : wf_sub2(d[4 * (I_) + 1], d[4 * (I_) + 3]);                        \
        *reinterpret_cast<f32x2 *>(Vs + (BUF_) * WF_POS + (4 * (I_) + (Q_)) * (64 * WF_BK) + st_off) = v_; \
    }
    const int st_off = t_ld * WF_BK + (((cp >> 1) ^ ((t_ld >> 4) & 1)) * 4) + (cp & 1) * 2;
    const int rd_sw = (half ^ ((li >> 4) & 1)) * 4;
    const int rd_v = (ph * 8) * (64 * WF_BK) + (wm * 32 + li) * WF_BK + rd_sw;   // this wave's first position
    // one base register for all operand addresses of U: hidden from the optimiser, which otherwise keeps one register per
    // address (the constants past 64 KiB do not fit the instruction's offset field once folded with the image's own offset)
    unsigned rd_ub = (unsigned)((2 * WF_POS + (ph * 8) * (64 * WF_BK) + (wn * 32 + li) * WF_BK + rd_sw) * 4);
    asm volatile("" : "+v"(rd_ub));

    f32x16 acc[8];
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 af[2][2], bf[2][2];
#define W8_READ_OP(SET_, BUF_, G_, Q_)                                                                   \
    {                                                                                                    \
        if (((Q_) & 1) == 0)                                                                             \
            af[SET_][(Q_) >> 1] = *reinterpret_cast<const f32x4 *>(Vs + (BUF_) * WF_POS + rd_v + (2 * (G_) + ((Q_) >> 1)) * (64 * WF_BK)); \
        else                                                                                             \
            bf[SET_][(Q_) >> 1] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(wf_smem) + rd_ub + ((BUF_) * WF_POS + (2 * (G_) + ((Q_) >> 1)) * (64 * WF_BK)) * 4); \
    }

    // K step: 32 slots of one MFMA (4 groups g of two positions x 4 k-pairs), each closed by a full
    // scheduling barrier.  Slots 0-3 of a group read the next group's operands (g = 3: after the
    // barrier, the next step's first group).
    //   ROLE 0 (load):   chunk kc+2 -> d, one buffer load per slot in groups 0-1; a DMA wave: a piece of its quarter
    //                    of U(kc+1) ahead of the load of every even slot; before the barrier vmcnt(2): the two loads
    //                    behind the last piece may still be in flight
    //   ROLE 1 (store):  slot 0 row transform of d (chunk kc+1, loaded in the previous step); groups 0-1 of a DMA
    //                    wave: its quarter of U(kc+1), one piece per slot from slot 1; groups 1-2 the column
    //                    transform + LDS store, one row (four positions) per fourth slot; before the barrier vmcnt(0)
    // Chunk indices past the end are clamped (staging nobody reads).
#ifdef W8_STAMPS
#define W8_BAR_STAMPED(ROLE)                                                                             \
    {                                                                                                    \
        const unsigned t0_ = (unsigned)__builtin_amdgcn_s_memtime();                                     \
        __syncthreads();                                                                                 \
        st8[ROLE] += (unsigned)__builtin_amdgcn_s_memtime() - t0_;                                       \
    }
    unsigned st8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned ts0 = 0, ts1 = 0, ts2 = 0;
#define W8_LOADWAIT_STAMPED(ROLE, SL)                                                                    \
    if ((SL) == 0) ts0 = (unsigned)__builtin_amdgcn_s_memtime();                                         \
    if ((SL) == 1) ts1 = (unsigned)__builtin_amdgcn_s_memtime();                                         \
    if ((SL) == 10) ts2 = (unsigned)__builtin_amdgcn_s_memtime();                                        \
    if ((SL) == 23) {                                                                                    \
        const unsigned ts3 = (unsigned)__builtin_amdgcn_s_memtime();                                     \
        if (ROLE == 1) { st8[2] += ts1 - ts0; st8[3] += ts2 - ts1; st8[4] += ts3 - ts2; }                \
        else st8[5] += ts3 - ts0;                                                                        \
    }
#else
#define W8_LOADWAIT_STAMPED(ROLE, SL)
#define W8_BAR_STAMPED(ROLE) if (!(W8_ABL & 8)) __syncthreads();
#endif
#define W8_KSTEP(ROLE, FIRST, OFFS, SID)                                                                          \
    {                                                                                                    \
        /* chunks past the end are the next unit's chunks 0 and 1 (the last unit of a workgroup: its own again, staging nobody reads) */ \
        const int kn1 = kc + 1 < KC ? kc + 1 : 0;                                                        \
        const int kn2 = kc + 2 < KC ? kc + 2 : kc + 2 - KC;                                              \
        i32x4 rsUs = rsU;                                                                                \
        if (kc + 1 >= KC) { rsUs.x = rsU_n.x; rsUs.y = rsU_n.y; }                                        \
        _Pragma("unroll") for (int g = 0; g < 4; g++)                                                    \
        {                                                                                                \
            _Pragma("unroll") for (int i = 0; i < 8; i++)                                                \
            {                                                                                            \
                const int sl = 8 * g + i;                                                                \
                if (g == 3 && i == 0) {                                                                  \
                    if (dma_wave && ROLE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
                    if (dma_wave && ROLE == 0 && !(W8_ABL & 5)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); \
                    W8_BAR_STAMPED(ROLE)                                                                 \
                }                                                                                        \
                if (i < 4) {                                                                             \
                    if (g < 3) W8_READ_OP((g + 1) & 1, cur, g + 1, i)                                    \
                    else W8_READ_OP(0, cur ^ 1, 0, i)                                                    \
                }                                                                                        \
                if (!(W8_ABL & 4) && dma_wave && ROLE == 0 && sl < 16 && (sl & 1) == 0) W8_DMA_U1(rsUs, kn1, cur ^ 1, sl >> 1) \
                if (!(W8_ABL & 1) && ROLE == 0 && sl < 16) W8_LOAD_V1(kn2, sl)                           \
                W8_LOADWAIT_STAMPED(ROLE, sl)                                                            \
                if (!(W8_ABL & 2) && ROLE == 1 && sl == 0) W8_ROWS()                                     \
                if (!(W8_ABL & 4) && dma_wave && ROLE == 1 && sl >= 1 && sl <= 8) W8_DMA_U1(rsUs, kn1, cur ^ 1, sl - 1) \
                if (!(W8_ABL & 2) && ROLE == 1 && sl >= 10 && sl <= 22 && ((sl - 10) & 3) == 0) {        \
                    _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) W8_STORE_POS(cur ^ 1, (sl - 10) >> 2, q_) \
                }                                                                                        \
                if (ROLE == 1 && (OFFS) && sl == 25) {                                                   \
                    /* this wave's next load step takes the next unit's chunk 0 / 1: its offsets and descriptors, now */ \
                    W8_DESCRIPTORS(rsU_n, bn_n, b0_n)                                                    \
                    W8_OFFSETS(voff_n, vmask_n)                                                          \
                }                                                                                        \
                acc[2 * g + (i & 1)] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i & 1][i >> 1], bf[g & 1][i & 1][i >> 1], (FIRST) && i < 2 ? zero16 : acc[2 * g + (i & 1)], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                       \
            }                                                                                            \
        }                                                                                                \
        cur ^= 1;                                                                                        \
        W8_UK(SID)                                                                                       \
    }


#ifdef W8_USTAMPS
    unsigned us8[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned ust;
    unsigned uk8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned ukt;
#endif
    int bn, b0, tb = 0;
    int bn_n = 0, b0_n = 0;
    {
        unsigned voff, vmask;
        W8_SETUP(x_start + lidl, 0, bn, b0, voff, vmask)
        W8_DESCRIPTORS(rsU, bn, b0)
        W8_OFFSETS(voff, vmask)
    }
    // ---- the workgroup's first unit: first loads, parity 0 stages chunk 0.  Every later unit is staged by the K loop of
    //      the unit before it: the loads, transform, stores and U DMA of its last two steps, which have no chunk of their
    //      own left to fetch, take the next unit's chunks 0 and 1, so the staging pipeline never drains ----
    W8_FIRST_LOADS()
    if (ph == 0) {
        W8_ROWS()
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) W8_STORE_POS(0, i, qq)
    }
    if (dma_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the U DMA has landed
    for (;;) {
        WF_STAMP(1)
        // this unit's bias, now: a load issued in the epilogue would be younger than the next unit's patch loads and
        // its wait would cover them
        const int n = bn * WF_BN + wn * 32 + li;
        const float bv = bias[n];
        __builtin_amdgcn_sched_barrier(0);
#ifdef W8_USTAMPS
        ust = (unsigned)__builtin_amdgcn_s_memtime();
#endif
        const int nl = lidl + per;
        const bool more = nl < x_count;             // workgroup-uniform
        __syncthreads();                            // chunk 0 is in Vs[0] / Us[0]; every wave is done with the last unit's exchange and tile table
        W8_US(6)
        // the next unit's tile arithmetic, here, where the accumulators are not live yet (its tile table replaces the last
        // unit's); the K loop expands it to offsets when this unit's last chunk has been loaded.  The last unit of a
        // workgroup prepares itself again: loads of valid memory into staging that nobody reads
        unsigned voff_n, vmask_n;
        W8_SETUP(x_start + (more ? nl : lidl), tb ^ 1, bn_n, b0_n, voff_n, vmask_n)
        W8_US(5)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) W8_READ_OP(0, 0, 0, qq)

        WF_STAMP(2)
        int cur = 0;
#ifdef W8_STAMPS
        const unsigned tk0 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#ifdef W8_USTAMPS
        ukt = (unsigned)__builtin_amdgcn_s_memtime();
#endif
        // the first step's first MFMA of every accumulator takes C = 0: no clearing pass (128 v_mov a wave)
        // KC is even and >= 6.  The step that ends a wave's loads of this unit (KC - 3 for parity 0, KC - 2 for parity 1, its
        // store role) turns the next unit's tile arithmetic into the wave's offsets: the last four steps are straight-line
        // code of their own (a branch around that piece inside the loop costs hipcc 700 spilled registers)
        if (phs == 0) {
            int kc = 0;
            W8_KSTEP(0, true, 0, 0)
            kc = 1;
            W8_KSTEP(1, false, 0, 1)
            for (kc = 2; kc < KC - 4; kc += 2) {
                W8_KSTEP(0, false, 0, 2)
                kc++;
                W8_KSTEP(1, false, 0, 3)
                kc--;
            }
            W8_KSTEP(0, false, 0, 4)
            kc++;
            W8_KSTEP(1, false, 1, 5)
            kc++;
            W8_KSTEP(0, false, 0, 6)
            kc++;
            W8_KSTEP(1, false, 0, 7)
        } else {
            int kc = 0;
            W8_KSTEP(1, true, 0, 0)
            kc = 1;
            W8_KSTEP(0, false, 0, 1)
            for (kc = 2; kc < KC - 4; kc += 2) {
                W8_KSTEP(1, false, 0, 2)
                kc++;
                W8_KSTEP(0, false, 0, 3)
                kc--;
            }
            W8_KSTEP(1, false, 0, 4)
            kc++;
            W8_KSTEP(0, false, 0, 5)
            kc++;
            W8_KSTEP(1, false, 1, 6)
            kc++;
            W8_KSTEP(0, false, 0, 7)
        }
#ifdef W8_STAMPS
        st8[6] += (unsigned)__builtin_amdgcn_s_memtime() - tk0;
        st8[7] += KC;
#endif
        W8_US(7)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x2 bv2 = {bv, bv};
        asm volatile("" : "+v"(bv2));   // hipcc's own wait for the bias load goes here, where nothing is in flight
        WF_STAMP(3)

        // (the next unit's chunk 0 is in Vs[0] / Us[0] by now, its chunk 1 in parity 1's registers)
        WF_STAMP(4)
        W8_US(0)

        // ---- epilogue.  acc[q][e]: position ph*8 + q = row 2 ph + q/4, column q%4 of the 4x4 m; e -> tile row
        //      (e&3) + 8*(e>>2) + 4*(lane>>5), lane&31 -> channel.  A^T = [1 1 1 0; 0 1 -1 -1].
        //      partial outputs of this wave's two m rows; the pair swaps halves through LDS:
        //      parity 0 finishes pixels (0,0),(0,1), parity 1 finishes (1,0),(1,1) ----
        //      The exchange runs through the buffers the last K step read, Vs[1] (pairs 0, 1) and Us[1] (pairs 2, 3).
        float *xw = wf_smem + (2 * (sub >> 1) + 1) * WF_POS + (((sub & 1) * 2 + ph) * 16) * 128 + lane * 2;          // what this wave hands over   [2 ep + column][lane][2]
        const float *xr = wf_smem + (2 * (sub >> 1) + 1) * WF_POS + (((sub & 1) * 2 + (ph ^ 1)) * 16) * 128 + lane * 2;   // what it receives
        // two tile rows (e = 2 ep, 2 ep + 1: neighbouring accumulator registers) per packed instruction
        f32x2 keep[8][2];
        // PH_ is a literal: two straight-line copies, chosen once by the (scalar) parity
#define W8_EPI_FIRST(PH_)                                                                                \
        _Pragma("unroll") for (int ep = 0; ep < 8; ep++)                                                 \
        {                                                                                                \
            f32x2 t0[4], t1[4];                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; j++)                                                \
            {                                                                                            \
                const f32x2 a = {acc[j][2 * ep], acc[j][2 * ep + 1]};                                    \
                const f32x2 b = {acc[4 + j][2 * ep], acc[4 + j][2 * ep + 1]};                            \
                if ((PH_) == 0) { t0[j] = wf_add2(a, b); t1[j] = b; }                                    \
                else { t0[j] = a; t1[j] = wf_nadd2(a, b); }                                              \
            }                                                                                            \
            const f32x2 y00 = wf_add2(wf_add2(t0[0], t0[1]), t0[2]), y01 = wf_sub2(wf_sub2(t0[1], t0[2]), t0[3]); \
            const f32x2 y10 = wf_add2(wf_add2(t1[0], t1[1]), t1[2]), y11 = wf_sub2(wf_sub2(t1[1], t1[2]), t1[3]); \
            keep[ep][0] = (PH_) == 0 ? y00 : y10;                                                        \
            keep[ep][1] = (PH_) == 0 ? y01 : y11;                                                        \
            *reinterpret_cast<f32x2 *>(xw + (2 * ep) * 128) = (PH_) == 0 ? y10 : y00;                    \
            *reinterpret_cast<f32x2 *>(xw + (2 * ep + 1) * 128) = (PH_) == 0 ? y11 : y01;                \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }
        if (phs == 0) { W8_EPI_FIRST(0) } else { W8_EPI_FIRST(1) }
#undef W8_EPI_FIRST
        W8_US(1)
        __syncthreads();
        W8_US(2)
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (long long)b0 * H * W * Cout), 0, 0x40000000, 0x00020000);
        const unsigned nb4 = 4u * (unsigned)n;
        // parity 0 finishes the tile's pixels (0,0), (0,1), parity 1 (1,0), (1,1): its half of the tile's table row
        const unsigned *tp = tbl + tb * 256 + (wm * 32 + 4 * half) * 4 + 2 * ph;
#pragma unroll
        for (int ep = 0; ep < 8; ep++) {
            const f32x2 g0 = *reinterpret_cast<const f32x2 *>(xr + (2 * ep) * 128);
            const f32x2 g1 = *reinterpret_cast<const f32x2 *>(xr + (2 * ep + 1) * 128);
            // (rows 0,1) + (rows 2,3), then the bias: the order of wino_fused_f2
            const f32x2 ya = wf_add2(wf_add2(keep[ep][0], g0), bv2);
            const f32x2 yb = wf_add2(wf_add2(keep[ep][1], g1), bv2);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int e = 2 * ep + h;
                const u32x2 o = *reinterpret_cast<const u32x2 *>(tp + ((e & 3) + 8 * (e >> 2)) * 4);   // absent pixels stay out of range
                wf_store1(rsO, o.x + nb4, wf_relu(h ? ya.y : ya.x));
                wf_store1(rsO, o.y + nb4, wf_relu(h ? yb.y : yb.x));
            }
        }
        WF_STAMP(5)
        W8_US(3)
        if (!more) break;
        bn = bn_n; b0 = b0_n; tb ^= 1; lidl = nl;
        rsU.x = rsU_n.x; rsU.y = rsU_n.y;
    }
#ifdef W8_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; i++) w8_stamps[wave][i] += st8[i];
#endif
#ifdef W8_TIMES
    if (tid == 0) w8_times[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef W8_USTAMPS
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 10; i++) w8_ustamps[wave][i] += us8[i];
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; i++) w8_ksteps[wave][i] += uk8[i];
#endif
#undef W8_KSTEP
#undef W8_BAR_STAMPED
#undef W8_LOADWAIT_STAMPED
#undef W8_READ_OP
#undef W8_LOAD_V1
#undef W8_DMA_U1
#undef W8_ROWS
#undef W8_STORE_POS
#undef W8_SETUP
#undef W8_DESCRIPTORS
#undef W8_OFFSETS
#undef W8_FIRST_LOADS
}

}  // namespace apr
