// wino_s2_fused.hip.h -- the stride-2 3x3 convolution (pad 1) in polyphase Winograd F(2,2) form, in ONE kernel.
//
// Same algebra as wino_s2.hip.h (25 products per 2x2 outputs, accumulated into the 3 x 3 sums S[qi][qj] the output
// transform needs), same machinery as wino_fused8.hip.h (input transform in registers between the global load and the
// LDS store of the A operand, U copied global -> LDS by LDS-DMA in the kernel's own image order, fp32 MFMA, output
// transform + bias + ReLU in the epilogue, persistent workgroups) -- so neither V nor the nine S tensors ever reach HBM
// (conv3x3_igemm_f32<..., 3> + wino_s2_output write and re-read S[9][T][Cout]).
//
// Work unit = 32 tiles (2x2 outputs each) x 64 channels, one workgroup of 8 waves per CU.  Wave = (channel half wn, q):
// the four waves q of a 32 x 32 sub-block split the nine sums -- and with them the 25 positions -- 6 / 6 / 6 / 7:
//     q0: S00 (4 positions) + S01 (2)      q1: S02 (4) + S12 (2)      q2: S20 (4) + S10 (2)      q3: S22 (4) + S21 (2) + S11 (1)
// i.e. 32 or 48 accumulator registers a wave -- except that S11 of channel half 1 goes to the q0 wave of that half (position
// sets, ws_tab), so that the SIMDs issue 52 / 48 / 48 / 52 MFMAs a step instead of 48 / 48 / 48 / 56.  Waves w and w + 4 share a
// SIMD (tools/probe/hwid.hip) and a q.  They do NOT run in step: the SIMD serves the older wave (w, channel half 0) first, so
// that wave reaches the step's barrier ~2 k cycles before its partner, and its staging instructions run beside the partner's
// MFMAs while the partner's own are exposed (s_memtime stamps, -DWS_STAMPS + tools/ws_stamps.py; timing ablations: staging
// by the waves of half 0 alone costs 3 % of the step, by those of half 1 alone 6.5 %).  Hence the whole U DMA is the work of
// the half-0 waves.  (Measured and dropped: q = (w + wn) & 3, which spreads the two 7-position waves and the two waves of a
// staging group over different SIMDs, 2-4 % slower; S11 shared between the four waves, one k-pair each, so that every wave
// issues 25 MFMAs a step, 3 % slower; U straight from global memory into the B operand registers, no LDS image: 1.5 % slower
// -- the bytes through the vector memory path cost the time, not the LDS reads; three staging groups q0-q2 with the
// 7-position waves exempt: 1-2 % slower; the half-0 waves staging whole chunks with 16-B loads: 2-4 % slower.)
// K step = 8 input channels, double-buffered LDS images
//     Vs[25 positions][32 tiles][8]  (25 KiB)      Us[16 weight matrices][64 channels][8]  (32 KiB; positions 3 and 4 of an axis share one)
// one barrier per K step.  The staging of a chunk is the work of ONE wave pair (q = chunk mod 4; 128 lanes = 32 tiles x 4
// channel pairs): it loads the chunk's 5 x 5 patches THREE steps ahead (two whole K steps of latency: first touches come
// from HBM), transforms them two steps ahead and stores them one step ahead; its half-0 wave copies U in the two other steps:
//     step k:  group (k+3)%4 loads chunk k+3 | group (k+2)%4 transforms chunk k+2 |
//              group (k+1)%4 stores chunk k+1 and DMAs 12 KiB of U(k+1) | group k%4 DMAs 20 KiB of U(k+1)
// (a DMA wave waits vmcnt(0) before the step's barrier: nothing else of its own is in flight then)
// The K loop is unrolled by four, so a wave's role in every step is a compile-time constant of its code path (one path
// per q).  Workgroups are persistent (one per CU, the units cut into 8 XCD-contiguous runs) and the staging pipeline runs on
// across the unit boundary: the last four K steps of a unit (ws_kstep<..., TAIL>) fetch chunks 0, 1, 2 and U(0) of the
// workgroup's next unit, each group turning its patch offsets to that unit in its idle step of the four.  Epilogue: the sums that another wave needs (S01, S12, S10, S21, S11) are parked in the second staging
// buffers, one barrier, then wave q = 2a + b forms output pixel (a, b) of each tile,
//     Y[a][b] = (S[a][b] + S[a+1][b]) + (S[a][b+1] + S[a+1][b+1]) + bias, ReLU, NHWC store through the tile table.
#pragma once
#include <hip/hip_runtime.h>

#include "wino_fused8.hip.h"

namespace apr {

constexpr int WS_TT = 32;                         // tiles per unit
constexpr int WS_BN = 64;                         // output channels per unit
constexpr int WS_BK = 8;                          // input channels per K step
constexpr int WS_VIMG = 25 * WS_TT * WS_BK;       // floats of one V image
constexpr int WS_UIMG = 16 * WS_BN * WS_BK;       // floats of one U image (the layout of wf_pack_u: same DMA, same reads)
constexpr int WS_LDS_BYTES = (2 * WS_VIMG + 2 * WS_UIMG) * 4 + 2 * WS_TT * 16;
static_assert(WS_UIMG == WF_POS, "U images share wf_pack_u's layout");
static_assert(2 * 4 * 16 * 64 * 4 <= WS_UIMG * 4 && 2 * 16 * 64 * 4 <= WS_VIMG * 4, "the epilogue's exchange fits the second staging buffers");

#ifndef WS_DMA0
#define WS_DMA0 20   // KiB of a U image copied by the wave whose step has no other staging work (the rest: the wave of the store step; in the transform step hipcc's counted waits for the patch loads would cover the DMA: 16 KiB there cost 3 %)
#endif
#ifndef WS_ABL
#define WS_ABL 0   // developer timing ablations (wrong results): 1 no wait for the U DMA, 2 no U DMA, 4 no patch loads, 8 no transform / stores
#endif

// positions of wave q: n-th entry = (patch-transform row pi, column pj, accumulator slot)
// Position sets: 0..3 = the sums of q = 0..3 as listed above; S11 of channel half 1 is accumulated by the q = 0 wave of that half
// instead of its q = 3 wave (set 4 = set 0 + S11, set 5 = set 3 without S11), so that the four SIMDs issue 52 / 48 / 48 / 52 MFMAs
// a step instead of 48 / 48 / 48 / 56.
__host__ __device__ constexpr int ws_np(int ps) { return ps == 3 || ps == 4 ? 7 : 6; }
__host__ __device__ constexpr int ws_q(int ps) { return ps < 4 ? ps : ps == 4 ? 0 : 3; }
__host__ __device__ constexpr int ws_tab(int ps, int n, int f)
{
    constexpr int T[6][7][3] = {{{0, 0, 0}, {0, 3, 0}, {3, 0, 0}, {3, 3, 0}, {0, 1, 1}, {3, 1, 1}, {0, 0, 0}},
                                {{0, 2, 0}, {0, 4, 0}, {3, 2, 0}, {3, 4, 0}, {1, 2, 1}, {1, 4, 1}, {0, 0, 0}},
                                {{2, 0, 0}, {2, 3, 0}, {4, 0, 0}, {4, 3, 0}, {1, 0, 1}, {1, 3, 1}, {0, 0, 0}},
                                {{2, 2, 0}, {2, 4, 0}, {4, 2, 0}, {4, 4, 0}, {2, 1, 1}, {4, 1, 1}, {1, 1, 2}},
                                {{0, 0, 0}, {0, 3, 0}, {3, 0, 0}, {3, 3, 0}, {0, 1, 1}, {3, 1, 1}, {1, 1, 2}},
                                {{2, 2, 0}, {2, 4, 0}, {4, 2, 0}, {4, 4, 0}, {2, 1, 1}, {4, 1, 1}, {0, 0, 0}}};
    return T[ps][n][f];
}
__host__ __device__ constexpr int ws_umap(int p) { return p < 3 ? p : 3; }            // U = [w0, w0 + w2, w2, w1, w1]

#ifdef WS_STAMPS   // developer build (tools/ws_stamps.py): s_memtime stamps of workgroup 0, per wave: cycles waited at the barrier
                   // of the steps of each phase [0..3], epilogue [4], top of a unit [5], K loop total [6], steps [7]
__device__ unsigned long long ws_stamps[8][8];
#define WS_ST_ARG , unsigned (&st)[8]
#define WS_ST_PASS , st
#else
#define WS_ST_ARG
#define WS_ST_PASS
#endif

struct WsArgs {
    const float *in, *up, *bias;
    float *out;
    int H, W, Cin, Ho, Wo, Cout, th, tw, T, tiles_n;
    WfMagic mg;
};

struct WsCtx {   // per-wave constants of the whole launch
    float *Vs, *Us;
    unsigned *tbl;                               // [2][32][4]
    unsigned dma_v, us_lds;                      // U DMA: this lane's 16 B of a 1-KiB row; LDS address of Us[0]
    int st_off, rd_v, rd_u, KC, lane, wn, t_ld, cp;
};

struct WsUnit {   // what changes from unit to unit
    __amdgpu_buffer_rsrc_t rsA;
    i32x4 rsU;
    int bn, b0, g0;                              // N tile, first window, first tile
};

// unit lid -> its N tile, first tile and window, descriptors (wave-uniform)
__device__ __forceinline__ void ws_setup_unit(const WsArgs &a, const WsCtx &c, int lid, WsUnit &u)
{
    int bm;
    wf_unit(lid, a.mg, bm, u.bn);
    u.g0 = bm * WS_TT;                           // first tile of the unit (< T)
    u.b0 = wf_div(u.g0, a.mg.tpw);               // its window
    u.rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (((long long)u.b0 * a.H - 1) * a.W - 1) * (long long)a.Cin), 0, 0x40000000, 0x00020000);
    const unsigned long long ub = (unsigned long long)(a.up + (size_t)u.bn * c.KC * WS_UIMG);
    u.rsU.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ub);
    u.rsU.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(ub >> 32) & 0xFFFF);
    u.rsU.z = 0x7FFFFFFF;
    u.rsU.w = 0x00020000;
}

// this lane's 25 patch offsets in unit u (padding = out-of-range offsets), and (write_tbl) the tile table: byte offsets of
// every tile's four output pixels, out of range if absent
__device__ __forceinline__ void ws_setup_lanes(const WsArgs &a, const WsCtx &c, const WsUnit &u, int tb, bool write_tbl, unsigned (&vo)[25])
{
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff = OOB, vmask = 0;
    u32x4 to = {OOB, OOB, OOB, OOB};
    const int g = u.g0 + c.t_ld;
    const WfUnitBase ub = wf_unit_base(u.g0, a.mg, a.th, a.tw);   // wave-uniform
    if (g < a.T) {
        int bb, ty, tx;
        wf_tile(ub, c.t_ld, a.mg, a.th, a.tw, bb, ty, tx);
        const int db = bb - u.b0;
        // patch = input rows 4ty-1 .. 4ty+3, columns 4tx-1 .. 4tx+3; the descriptor starts one row and one column early
        voff = 4u * (unsigned)(((db * a.H + 4 * ty) * a.W + 4 * tx) * a.Cin + 2 * c.cp);
        unsigned rowok = 0, colok = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            rowok |= (4 * ty - 1 + i >= 0 && 4 * ty - 1 + i < a.H) ? 1u << i : 0u;
            colok |= (4 * tx - 1 + i >= 0 && 4 * tx - 1 + i < a.W) ? 1u << i : 0u;
        }
#pragma unroll
        for (int i = 0; i < 5; i++) vmask |= ((rowok >> i) & 1u) ? colok << (5 * i) : 0u;
        const unsigned o00 = 4u * (unsigned)(((db * a.Ho + 2 * ty) * a.Wo + 2 * tx) * a.Cout);
        const bool r1 = 2 * ty + 1 < a.Ho, c1 = 2 * tx + 1 < a.Wo;
        to.x = o00;
        to.y = c1 ? o00 + 4u * (unsigned)a.Cout : OOB;
        to.z = r1 ? o00 + 4u * (unsigned)(a.Wo * a.Cout) : OOB;
        to.w = r1 && c1 ? o00 + 4u * (unsigned)((a.Wo + 1) * a.Cout) : OOB;
    }
    if (write_tbl && c.cp == 0) *reinterpret_cast<u32x4 *>(c.tbl + tb * (WS_TT * 4) + 4 * c.t_ld) = to;
#pragma unroll
    for (int ij = 0; ij < 25; ij++)
        vo[ij] = ((vmask >> ij) & 1u) ? voff + 4u * (unsigned)(((ij / 5) * a.W + (ij % 5)) * a.Cin) : OOB;
}

// rows: d <- B^T d per patch column, B^T d = [d0 - d2, d2, d4 - d2, d1, d3]
__device__ __forceinline__ void ws_rows(f32x2 (&d)[25])
{
#pragma unroll
    for (int jj = 0; jj < 5; jj++) {
        const f32x2 t0 = wf_sub2(d[jj], d[10 + jj]), t2 = wf_sub2(d[20 + jj], d[10 + jj]);
        const f32x2 t1 = d[10 + jj], t3 = d[5 + jj], t4 = d[15 + jj];
        d[jj] = t0; d[5 + jj] = t1; d[10 + jj] = t2; d[15 + jj] = t3; d[20 + jj] = t4;
    }
}
// columns: d <- d B per transformed row, in place: row i becomes [r0 - r2, r1, r2, r3, r4 - r2] (positions 0, 3, 1, 4, 2)
__device__ __forceinline__ void ws_cols(f32x2 (&d)[25])
{
#pragma unroll
    for (int i = 0; i < 5; i++) {
        d[5 * i] = wf_sub2(d[5 * i], d[5 * i + 2]);
        d[5 * i + 4] = wf_sub2(d[5 * i + 4], d[5 * i + 2]);
    }
}
// the five LDS stores of transformed row i into V image `buf` (after ws_rows and ws_cols)
__device__ __forceinline__ void ws_store_row(const WsCtx &c, const f32x2 (&d)[25], int buf, int i)
{
    float *vs = c.Vs + buf * WS_VIMG + (5 * i) * (WS_TT * WS_BK) + c.st_off;
    *reinterpret_cast<f32x2 *>(vs + 0 * (WS_TT * WS_BK)) = d[5 * i];
    *reinterpret_cast<f32x2 *>(vs + 1 * (WS_TT * WS_BK)) = d[5 * i + 2];
    *reinterpret_cast<f32x2 *>(vs + 2 * (WS_TT * WS_BK)) = d[5 * i + 4];
    *reinterpret_cast<f32x2 *>(vs + 3 * (WS_TT * WS_BK)) = d[5 * i + 1];
    *reinterpret_cast<f32x2 *>(vs + 4 * (WS_TT * WS_BK)) = d[5 * i + 3];
}

// KiB i (0..31) of U image kc into Us[buf]
__device__ __forceinline__ void ws_dma(const WsCtx &c, const WsUnit &u, int kc, int buf, int i)
{
    lds_dma_b128(u.rsU, c.dma_v, __builtin_amdgcn_readfirstlane(kc * (WS_UIMG * 4) + i * 1024),
                 __builtin_amdgcn_readfirstlane(c.us_lds + (unsigned)(buf * WS_UIMG * 4 + i * 1024)));
}

// first staging of a unit: groups 0, 1, 2 load chunks 0, 1, 2 (transformed in the prologue, in step 0, in step 1), groups 2 and
// 3 DMA U(0) into Us[0] (group 2 first, so that its wait for the DMA does not depend on the loads behind it)
template <int Q>
__device__ __forceinline__ void ws_first_loads(const WsCtx &c, const WsUnit &u, const unsigned (&vo)[25], f32x2 (&d)[25])
{
    if (Q >= 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) ws_dma(c, u, 0, 0, (Q == 2 ? 0 : 16) + 8 * c.wn + i);
    }
    if (Q <= 2) {
        const int kq = Q < c.KC ? Q : c.KC - 1;
#pragma unroll
        for (int ij = 0; ij < 25; ij++) d[ij] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(u.rsA, vo[ij], 4 * kq * WS_BK, 0));
    }
}

// K step kc (phase PH = kc mod 4) of a wave of group Q: one MFMA per slot, each closed by a full scheduling barrier so that
// the staging work stays where it is written, a slot's worth at a time, in the shadow of the MFMAs.
// TAIL: one of the unit's last four steps (kc = KC - 4 + PH).  Their staging has no chunk of this unit left to fetch beyond
// KC - 1: it takes chunks 0, 1, 2 and U(0) of the workgroup's next unit un, so the pipeline runs on across the unit boundary;
// a group turns to the next unit in its idle step of the four (ROLE 0, after the barrier: its registers d are free then).
template <int PS, int WN, int PH, bool TAIL>
__device__ __forceinline__ void ws_kstep(const WsArgs &a, const WsCtx &c, const WsUnit &u, const WsUnit &un, int tb, int kc, int &cur, f32x16 (&acc)[3],
                                         f32x4 (&af)[2], f32x4 (&bf)[2], f32x2 (&d)[25], unsigned (&vo)[25] WS_ST_ARG)
{
    constexpr int Q = ws_q(PS);
    constexpr int NP = ws_np(PS);
    constexpr int ROLE = (Q - PH + 4) & 3;   // 3: load chunk kc+3, 2: transform chunk kc+2, 1: store chunk kc+1 (+ U DMA), 0: U DMA only
    constexpr int SET0 = (PH * NP) & 1;      // operand register set of this step's first position
    constexpr bool NEXT_V = TAIL && PH >= 1, NEXT_U = TAIL && PH == 3;
    const int kn1 = NEXT_U ? 0 : kc + 1;
    const int kn3 = NEXT_V ? PH - 1 : kc + 3;
    const WsUnit &uv = NEXT_V ? un : u;
    const WsUnit &uu = NEXT_U ? un : u;
#pragma unroll
    for (int n = 0; n < NP; n++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int sl = 4 * n + j;
            const int set = (SET0 + n) & 1;
            if (n == NP - 1 && j == 0) {
                if (!(WS_ABL & 1) && WN == 0 && (ROLE == 0 || ROLE == 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the U DMA has landed
#ifdef WS_STAMPS
                const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
                __syncthreads();
                st[PH] += (unsigned)__builtin_amdgcn_s_memtime() - t0;
#else
                __syncthreads();
#endif
            }
            // operands of the next position (of the next step's first one, out of the other buffer, after the barrier)
            if (j < 2) {
                const int nn = n + 1 < NP ? n + 1 : 0;
                const int buf = n + 1 < NP ? cur : cur ^ 1;
                const int p = 5 * ws_tab(PS, nn, 0) + ws_tab(PS, nn, 1);
                const int um = 4 * ws_umap(ws_tab(PS, nn, 0)) + ws_umap(ws_tab(PS, nn, 1));
                if (j == 0) af[set ^ 1] = *reinterpret_cast<const f32x4 *>(c.Vs + buf * WS_VIMG + p * (WS_TT * WS_BK) + c.rd_v);
                else bf[set ^ 1] = *reinterpret_cast<const f32x4 *>(c.Us + buf * WS_UIMG + um * (WS_BN * WS_BK) + c.rd_u);
            }
            if (ROLE == 3 && !(WS_ABL & 4)) {   // patch loads of chunk kc+3: one per slot (24 slots at least), the 25th with the 24th
                if (sl < 24) d[sl] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(uv.rsA, vo[sl], 4 * kn3 * WS_BK, 0));
                if (sl == 23) d[24] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(uv.rsA, vo[24], 4 * kn3 * WS_BK, 0));
            }
            // B^T d B of chunk kc+2 in the step before its store step (its loads went out a step ago)
            if (ROLE == 2 && !(WS_ABL & 8)) {
                if (sl == 14) ws_rows(d);
                if (sl == 17) ws_cols(d);
            }
            if (ROLE == 1 && !(WS_ABL & 8) && sl >= 1 && sl <= 13 && (sl - 1) % 3 == 0) ws_store_row(c, d, cur ^ 1, (sl - 1) / 3);
            // U(kc+1), 1 KiB a slot, by the waves of channel half 0 only (of the two waves of a SIMD the one with the lower id is
            // served first, tools/ws_stamps.py: its staging work runs beside the other wave's MFMAs, the other wave's does not)
            if (!(WS_ABL & 2) && WN == 0 && ((ROLE == 0 && sl < WS_DMA0) || (ROLE == 1 && sl < 32 - WS_DMA0)))
                ws_dma(c, uu, kn1, cur ^ 1, ROLE == 0 ? sl : WS_DMA0 + sl);
            if (TAIL && ROLE == 0 && sl == 4 * (NP - 1) + 1) ws_setup_lanes(a, c, un, tb ^ 1, Q == 0, vo);
            acc[ws_tab(PS, n, 2)] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][j], bf[set][j], acc[ws_tab(PS, n, 2)], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    cur ^= 1;
}

template <int PS, int WN>
__device__ __forceinline__ void ws_run(const WsArgs &a, const WsCtx &c, int x_start, int x_count, int lidl, int per)
{
    constexpr int Q = ws_q(PS);
    const int lane = c.lane, li = lane & 31, half = lane >> 5, wn = WN;
    f32x2 d[25];
    unsigned vo[25];
    f32x16 acc[3];
    f32x4 af[2], bf[2];
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    WsUnit u, un;
    int tb = 0;
#ifdef WS_STAMPS
    unsigned st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // ---- the workgroup's first unit: first loads, group 0 stages chunk 0, groups 2 and 3 see their DMA of U(0) land (group
    //      2: and its chunk 2).  Every later unit is staged by the last four K steps of the unit before it ----
    ws_setup_unit(a, c, x_start + lidl, u);
    ws_setup_lanes(a, c, u, 0, Q == 0, vo);
    ws_first_loads<Q>(c, u, vo, d);
    if (Q == 0 || Q == 1) {   // chunks 0 and 1 have no DMA step before their store step
        ws_rows(d);
        ws_cols(d);
    }
    if (Q == 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) ws_store_row(c, d, 0, i);
    }
    if (Q == 2 || Q == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (;;) {
        // this unit's bias, now: a load issued in the epilogue would be younger than the next unit's patch loads and its wait
        // would cover them
        const int n = u.bn * WS_BN + wn * 32 + li;
        const float bv = a.bias[n];
        __builtin_amdgcn_sched_barrier(0);
#ifdef WS_STAMPS
        const unsigned tu0 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int s = 0; s < 3; s++) acc[s] = zero16;
        // the next unit (the last unit of a workgroup: itself again, loads of valid memory into staging that nobody reads)
        const int nl = lidl + per;
        const bool more = nl < x_count;              // workgroup-uniform
        ws_setup_unit(a, c, x_start + (more ? nl : lidl), un);
        __syncthreads();   // chunk 0 is in Vs[0] / Us[0]; every wave is done with the last unit's exchange and tile table
        {
            constexpr int p = 5 * ws_tab(PS, 0, 0) + ws_tab(PS, 0, 1);
            constexpr int um = 4 * ws_umap(ws_tab(PS, 0, 0)) + ws_umap(ws_tab(PS, 0, 1));
            af[0] = *reinterpret_cast<const f32x4 *>(c.Vs + p * (WS_TT * WS_BK) + c.rd_v);
            bf[0] = *reinterpret_cast<const f32x4 *>(c.Us + um * (WS_BN * WS_BK) + c.rd_u);
        }
        int cur = 0;
#ifdef WS_STAMPS
        const unsigned tk0 = (unsigned)__builtin_amdgcn_s_memtime();
        st[5] += tk0 - tu0;                          // top of the unit: clear, next unit's scalars, barrier, first operands
#endif
        int kc = 0;
        for (; kc < c.KC - 4; kc += 4) {   // KC % 4 == 0 (Cin % 32 == 0, host-checked)
            ws_kstep<PS, WN, 0, false>(a, c, u, un, tb, kc, cur, acc, af, bf, d, vo WS_ST_PASS);
            ws_kstep<PS, WN, 1, false>(a, c, u, un, tb, kc + 1, cur, acc, af, bf, d, vo WS_ST_PASS);
            ws_kstep<PS, WN, 2, false>(a, c, u, un, tb, kc + 2, cur, acc, af, bf, d, vo WS_ST_PASS);
            ws_kstep<PS, WN, 3, false>(a, c, u, un, tb, kc + 3, cur, acc, af, bf, d, vo WS_ST_PASS);
        }
        ws_kstep<PS, WN, 0, true>(a, c, u, un, tb, kc, cur, acc, af, bf, d, vo WS_ST_PASS);
        ws_kstep<PS, WN, 1, true>(a, c, u, un, tb, kc + 1, cur, acc, af, bf, d, vo WS_ST_PASS);
        ws_kstep<PS, WN, 2, true>(a, c, u, un, tb, kc + 2, cur, acc, af, bf, d, vo WS_ST_PASS);
        ws_kstep<PS, WN, 3, true>(a, c, u, un, tb, kc + 3, cur, acc, af, bf, d, vo WS_ST_PASS);
#ifdef WS_STAMPS
        const unsigned tk1 = (unsigned)__builtin_amdgcn_s_memtime();
        st[6] += tk1 - tk0;
        st[7] += c.KC;
#endif
        f32x4 bv4 = {bv, bv, bv, bv};
        asm volatile("" : "+v"(bv4));   // hipcc's own wait for the bias load
        const int b0 = u.b0;
        // (the next unit's chunk 0 is in Vs[0] / Us[0] by now, its chunks 1 and 2 in the registers of groups 1 and 2)

        // ---- epilogue.  After the K loop's last barrier nobody reads the second staging buffers any more: the sums another
        //      wave needs are parked there -- slot 1 of every wave (S01, S12, S10, S21) in Us[1] as [wn][q][e / 4][lane] float4,
        //      S11 in Vs[1] as [wn][e / 4][lane] -- conflict-free 16-B accesses; one barrier; wave q = 2a + b then finishes output
        //      pixel (a, b) of every tile of its sub-block.  acc[s][e]: lane & 31 -> channel, e -> tile (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ----
        f32x4 *exs = reinterpret_cast<f32x4 *>(c.Us + WS_UIMG);
        f32x4 *exc = reinterpret_cast<f32x4 *>(c.Vs + WS_VIMG);
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) {
            exs[((wn * 4 + Q) * 4 + e4) * 64 + lane] = f32x4{acc[1][4 * e4], acc[1][4 * e4 + 1], acc[1][4 * e4 + 2], acc[1][4 * e4 + 3]};
            if (PS == 3 || PS == 4) exc[(wn * 4 + e4) * 64 + lane] = f32x4{acc[2][4 * e4], acc[2][4 * e4 + 1], acc[2][4 * e4 + 2], acc[2][4 * e4 + 3]};
        }
        __syncthreads();
        constexpr int IMP = Q == 0 ? 2 : Q == 1 ? 0 : Q == 2 ? 3 : 1;   // whose slot 1 this wave needs: S10, S01, S21, S12
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (long long)b0 * a.Ho * a.Wo * a.Cout), 0, 0x40000000, 0x00020000);
        const unsigned nb4 = 4u * (unsigned)n;
        const unsigned *tp = c.tbl + tb * (WS_TT * 4);
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) {
            const f32x4 s11 = exc[(wn * 4 + e4) * 64 + lane], imp = exs[((wn * 4 + IMP) * 4 + e4) * 64 + lane];
            const f32x4 own0 = {acc[0][4 * e4], acc[0][4 * e4 + 1], acc[0][4 * e4 + 2], acc[0][4 * e4 + 3]};
            const f32x4 own1 = {acc[1][4 * e4], acc[1][4 * e4 + 1], acc[1][4 * e4 + 2], acc[1][4 * e4 + 3]};
            // (S[a][b] + S[a+1][b]) + (S[a][b+1] + S[a+1][b+1]) + bias
            const f32x4 y = Q == 0   ? (own0 + imp) + (own1 + s11) + bv4      // (S00 + S10) + (S01 + S11)
                            : Q == 1 ? (imp + s11) + (own0 + own1) + bv4      // (S01 + S11) + (S02 + S12)
                            : Q == 2 ? (own1 + own0) + (s11 + imp) + bv4      // (S10 + S20) + (S11 + S21)
                                     : (s11 + own1) + (imp + own0) + bv4;     // (S11 + S21) + (S12 + S22)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int tl = r + 8 * e4 + 4 * half;                 // MFMA row e = 4 e4 + r
                const unsigned o = tp[tl * 4 + Q];                     // absent pixels hold an out-of-range offset: the store is dropped
                wf_store1(rsO, o + nb4, wf_relu(y[r]));
            }
        }
#ifdef WS_STAMPS
        st[4] += (unsigned)__builtin_amdgcn_s_memtime() - tk1;   // the epilogue, stores issued
#endif
        if (!more) break;
        // no barrier here: the exchange lives in the second staging buffers, which the next unit first writes in its step 0,
        // after the barrier at its top
        u = un;
        tb ^= 1;
        lidl = nl;
    }
#ifdef WS_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; i++) ws_stamps[wn * 4 + Q][i] += st[i];
#endif
}

// in [B][H][W][Cin]; up = packed U (ws_pack_u); out [B][Ho][Wo][Cout] = relu(conv_stride2 + bias).
// T = B * th * tw tiles, th = ceil(Ho/2), tw = ceil(Wo/2).  Requires Cin % 32 == 0, Cout % 64 == 0.
// n_units = ceil(T / 32) * Cout / 64; grid = wf8_grid(n_units, CUs) (persistent, a multiple of 8).
__global__ __launch_bounds__(512) void wino_s2_fused(WsArgs a, int n_units)
{
    extern __shared__ __attribute__((aligned(16))) float wf_smem[];
    WsCtx c;
    c.Vs = wf_smem;                              // [2][WS_VIMG]
    c.Us = wf_smem + 2 * WS_VIMG;                // [2][WS_UIMG]
    c.tbl = reinterpret_cast<unsigned *>(wf_smem + 2 * WS_VIMG + 2 * WS_UIMG);   // [2][32][4]: byte offsets of a tile's four output pixels (channel 0)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = __builtin_amdgcn_readfirstlane(wave >> 2);    // the wave's 32 of the unit's 64 channels
    const int q = __builtin_amdgcn_readfirstlane(wave & 3);      // its sums / staging group; waves w and w + 4 share a SIMD and a q
    const int li = lane & 31, half = lane >> 5;
    const int sid = wn * 64 + lane;              // 0..127 within the staging group
    c.t_ld = sid >> 2;                           // the group stages whole chunks: 128 lanes = 32 tiles x 4 channel pairs
    c.cp = sid & 3;
    c.KC = a.Cin / WS_BK;
    c.lane = lane;
    c.wn = wn;
    c.dma_v = (unsigned)(lane * 16);
    c.us_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)c.Us);
    // LDS row = 8 channels of one tile (or output channel) = two 16-B halves; rows 16..31 of every 32 keep their halves
    // swapped (wf_pack_u stores U the same way): the four 16-lane groups of a ds_read_b128 are conflict-free
    c.st_off = c.t_ld * WS_BK + (((c.cp >> 1) ^ ((c.t_ld >> 4) & 1)) * 4) + (c.cp & 1) * 2;
    const int rd_sw = (half ^ ((li >> 4) & 1)) * 4;
    c.rd_v = li * WS_BK + rd_sw;
    c.rd_u = (wn * 32 + li) * WS_BK + rd_sw;

    // Persistent workgroups, XCD-aware: the units are cut into 8 contiguous runs, one per XCD (neighbouring units share
    // input rows or weights and so an L2); the gridDim.x / 8 workgroups of an XCD walk its run.
    const int per = gridDim.x >> 3, xcd = blockIdx.x & 7;       // gridDim.x is a multiple of 8 (host)
    const int uq = n_units >> 3, ur = n_units & 7;
    const int x_start = xcd < ur ? xcd * (uq + 1) : ur * (uq + 1) + (xcd - ur) * uq;
    const int x_count = uq + (xcd < ur ? 1 : 0);
    const int lidl = blockIdx.x >> 3;
    if (lidl >= x_count) return;                // whole workgroup: nothing to do

    if (wn == 0) {
        if (q == 0) ws_run<0, 0>(a, c, x_start, x_count, lidl, per);
        else if (q == 1) ws_run<1, 0>(a, c, x_start, x_count, lidl, per);
        else if (q == 2) ws_run<2, 0>(a, c, x_start, x_count, lidl, per);
        else ws_run<3, 0>(a, c, x_start, x_count, lidl, per);
    } else {
        if (q == 0) ws_run<4, 1>(a, c, x_start, x_count, lidl, per);
        else if (q == 1) ws_run<1, 1>(a, c, x_start, x_count, lidl, per);
        else if (q == 2) ws_run<2, 1>(a, c, x_start, x_count, lidl, per);
        else ws_run<5, 1>(a, c, x_start, x_count, lidl, per);
    }
}

// Host: g [Cout][Cin][3][3] (OIHW, BN folded) -> packed U for wino_s2_fused, in wf_pack_u's image order:
//   up[((bn * KC + kc) * 16 + 4 ur + uc) * 64 * 8 + nl * 8 + c'] = (G g G^T)[ur][uc] of (cout = 64 bn + nl, cin = 8 kc + c),
//   G = [1 0 0; 1 0 1; 0 0 1; 0 1 0] (wino_s2.hip.h), evaluated in f64 and rounded once.
inline void ws_pack_u(const float *g, int Cin, int Cout, float *up)
{
    static const double G[4][3] = {{1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {0, 1, 0}};
    const int KC = Cin / WS_BK;
    for (int o = 0; o < Cout; o++)
        for (int c = 0; c < Cin; c++) {
            const float *w = g + ((size_t)o * Cin + c) * 9;
            double t[4][3];
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 3; k++) t[i][k] = G[i][0] * w[k] + G[i][1] * w[3 + k] + G[i][2] * w[6 + k];
            const int nl = o % WS_BN, cl = c % WS_BK;
            const size_t base = ((size_t)(o / WS_BN) * KC + c / WS_BK) * WS_UIMG + (size_t)nl * WS_BK + (((cl >> 2) ^ ((nl >> 4) & 1)) * 4) + (cl & 3);
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                    up[base + (size_t)(4 * i + j) * (WS_BN * WS_BK)] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}

}  // namespace apr
