// frontend.hip.h -- K1..K3: s16le PCM -> framed Hann STFT -> power -> mel -> dB/normalise.
//
// Replaces model.Prepare's DSP (src/model/model.go:98-171, src/model/dsp.go:83-155)
// and, with the librosa tables, ai.model.audio_to_spectrogram (ai/model.py:184-204).
//
// Kernel 1 (stft_mel_kernel): one 64-lane wave per frame.  The 2048-point real FFT
// is a 1024-point complex Stockham FFT (radix 16,16,4) with 16 points per lane in
// registers; passes exchange data through a wave-private, bank-padded LDS buffer.
// The packed-real post-process, |X|^2 and the sparse (CSR) mel projection run in
// the same wave, so the 1025x184 power matrix never leaves the CU.  Output is the
// mel power, frame-major [window][frame][128] (512-B coalesced rows).
// Kernel 2 (db_norm_kernel): one workgroup per window: max-reduce (dB reference),
// 10*log10(v/max), clip [-80,0], (x+80)/80, transposed through LDS to the
// reference's row-major [128][188] with zero columns >= n_frames.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apr {

constexpr int kNfft = 2048;
constexpr int kHop = 512;
constexpr int kNmels = 128;
constexpr int kSpecFrames = 188;
constexpr int kSampleBytes = 192000;
#ifndef APR_FE_FPB
#define APR_FE_FPB 8
#endif
constexpr int kFramesPerBlock = APR_FE_FPB;    // frames per workgroup = its waves (one frame per wave): the tables a workgroup stages in LDS
                                      // serve eight frames; 61 KB of LDS (62 464 B, asserted below) = two workgroups, 16 waves per CU
constexpr int kMaxNnz = 2304;         // CSR capacity (go lens: 1892, librosa: ~2.1k)
constexpr int kMaxNnzPad = 2688;      // the same with every band widened to whole quads of bins starting at a multiple of 4

struct FrontendTables {
    const float *window;     // [2048] analysis window
    const float2 *tw;        // [2048] W_2048^i = (cos(2*pi*i/2048), -sin(2*pi*i/2048))
    const float2 *tw23;      // [256 + 768] the twiddles of passes 2 and 3 in the order the lanes read them:
                             // [16 t + k] = W_256^(k t) (k, t < 16), then [256 (t - 1) + j] = W_1024^(j t) (t = 1..3, j < 256)
    const int *mel_start;    // [128] first bin of each band
    const int *mel_len;      // [128] bins in each band
    const int *mel_off;      // [128] offset of each band in mel_w
    const float *mel_w;      // [nnz] band weights, bin-ascending per band
    const int *mel_start4;   // [128] first bin of the band's first quad (mel_start rounded down to a multiple of 4)
    const int *mel_len4;     // [128] quads of bins the band touches
    const int *mel_off4;     // [128] offset of the band's padded weights in mel_wp (a multiple of 4)
    const float *mel_wp;     // [nnz_pad] weights of whole quads: +0 outside the band
    int nnz, nnz_pad;
    int n_frames;            // 184 (go) or 188 (librosa)
    int frame_origin;        // first frame starts at this sample: 0 (go) or -1024 (librosa centre pad)
    int librosa;             // dB stage flavour
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i : (x + iy) * (-i) = y - ix
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// forward DFT4 in place: (a0,a1,a2,a3) -> (X0,X1,X2,X3)
__device__ __forceinline__ void dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    float2 s02 = cadd(a0, a2), d02 = csub(a0, a2);
    float2 s13 = cadd(a1, a3), d13 = mul_mi(csub(a1, a3));
    a0 = cadd(s02, s13);
    a2 = csub(s02, s13);
    a1 = cadd(d02, d13);
    a3 = csub(d02, d13);
}

// forward DFT16 in place, natural order in, natural order out (4x4 Cooley-Tukey)
__device__ __forceinline__ void dft16(float2 (&v)[16])
{
    // W_16^n = exp(-2*pi*i*n/16)
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
    const float2 w1 = make_float2(c1, -s1), w2 = make_float2(r2, -r2), w3 = make_float2(s1, -c1);
    const float2 w6 = make_float2(-r2, -r2), w9 = make_float2(-c1, s1);
    // step 1: for each n2, DFT4 over n1 of v[4*n1 + n2]  -> y[n2][k1] stored at v[4*k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // step 2: twiddle y[n2][k1] *= W_16^(n2*k1)
    v[4 * 1 + 1] = cmul(v[4 * 1 + 1], w1);
    v[4 * 1 + 2] = cmul(v[4 * 1 + 2], w2);
    v[4 * 1 + 3] = cmul(v[4 * 1 + 3], w3);
    v[4 * 2 + 1] = cmul(v[4 * 2 + 1], w2);
    v[4 * 2 + 2] = mul_mi(v[4 * 2 + 2]);           // W_16^4 = -i
    v[4 * 2 + 3] = cmul(v[4 * 2 + 3], w6);
    v[4 * 3 + 1] = cmul(v[4 * 3 + 1], w3);
    v[4 * 3 + 2] = cmul(v[4 * 3 + 2], w6);
    v[4 * 3 + 3] = cmul(v[4 * 3 + 3], w9);
    // step 3: for each k1, DFT4 over n2 of v[4*k1 + n2] -> X[k1 + 4*k2] left at v[4*k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// after dft16(), X[k1 + 4*k2] sits in v[4*k1 + k2]
__device__ __forceinline__ constexpr int dft16_slot(int k) { return 4 * (k & 3) + (k >> 2); }

__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }   // LDS bank padding

// wave-level LDS ordering: LDS ops of one wave execute in issue order, so only the
// compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr int kLdsPlane = 1104;   // padded real plane of one wave (pidx(1024) = 1088 fits; the power spectrum needs 1032)

// Exchange of one pass through the wave's single real plane: every lane writes its 16 values at WI(t), the wave reads
// them back at RI(t) -- first the real parts, then the imaginary parts (half the LDS of two planes: four workgroups
// per CU instead of three).  v[SLOT(t)] is written, v[t] is read.
#define APR_FE_EXCHANGE(WI, SLOT, RI)                                              \
    {                                                                              \
        float xr_[16];                                                             \
        _Pragma("unroll") for (int t = 0; t < 16; t++) pl[WI] = v[SLOT].x;         \
        wave_lds_sync();                                                           \
        _Pragma("unroll") for (int t = 0; t < 16; t++) xr_[t] = pl[RI];            \
        wave_lds_sync();                                                           \
        _Pragma("unroll") for (int t = 0; t < 16; t++) pl[WI] = v[SLOT].y;         \
        wave_lds_sync();                                                           \
        _Pragma("unroll") for (int t = 0; t < 16; t++) v[t] = make_float2(xr_[t], pl[RI]); \
        wave_lds_sync();                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
    }

__global__ __launch_bounds__(64 * kFramesPerBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void stft_mel_kernel(
    const uint8_t *__restrict__ pcm, size_t total_bytes, size_t stride_bytes, int n_windows, FrontendTables T,
    float *__restrict__ melT /*[n_windows][188][128]*/)
{
    constexpr int NT = 64 * kFramesPerBlock;
    __shared__ __attribute__((aligned(16))) float s_pl[kFramesPerBlock][kLdsPlane];
    __shared__ __attribute__((aligned(16))) float s_w[kMaxNnzPad];   // band weights, every band padded to whole aligned quads of bins
    __shared__ float2 s_tw[1024];                                    // W_2048^i, i < 1024: the post-process reads consecutive entries
    __shared__ float2 s_tw23[1024];                                  // passes 2 and 3, lane-major (FrontendTables::tw23): no bank conflicts
    // the four arrays above: 8 x 1104 x 4 + 2688 x 4 + 8192 + 8192 = 62 464 B.  Static LDS is capped at 64 KB per workgroup, and
    // the launch shape (two workgroups = 16 waves per CU) needs twice that inside the CU's 160 KB: raising kMaxNnzPad,
    // kLdsPlane or APR_FE_FPB past either bound must fail here, not drop occupancy silently.
    constexpr size_t kFeLdsBytes = sizeof(float) * kFramesPerBlock * kLdsPlane + sizeof(float) * kMaxNnzPad + 2 * sizeof(float2) * 1024;
    static_assert(kFeLdsBytes <= 65536, "stft_mel_kernel: static LDS over the 64 KB workgroup limit");
    static_assert(2 * kFeLdsBytes <= 160 * 1024, "stft_mel_kernel: two workgroups no longer fit one CU's 160 KB of LDS");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < T.nnz_pad / 4; i += NT) reinterpret_cast<float4 *>(s_w)[i] = reinterpret_cast<const float4 *>(T.mel_wp)[i];
#pragma unroll
    for (int i = 0; i < 1024 / NT; i++) { s_tw[tid + NT * i] = T.tw[tid + NT * i]; s_tw23[tid + NT * i] = T.tw23[tid + NT * i]; }

    constexpr int fpb = kFramesPerBlock;
    const int groups = (T.n_frames + fpb - 1) / fpb;
    const int win = blockIdx.x / groups, fg = blockIdx.x % groups;
    float *pl = s_pl[wave];

    const size_t win_off = (size_t)win * stride_bytes;
    long avail = 0;   // valid bytes of this window (model.go:88-96 pad/truncate)
    if (win < n_windows && win_off < total_bytes) {
        size_t a = total_bytes - win_off;
        avail = (long)(a > (size_t)kSampleBytes ? (size_t)kSampleBytes : a);
    }
    const uint8_t *wp = pcm + win_off;
    const bool aligned4 = (((uintptr_t)wp) & 3) == 0;

    // band pair owned by this lane (short band + long band, to balance the sparse dot): first quad of bins, quads, offset
    const int mA = lane, mB = kNmels - 1 - lane;
    const int qsA = T.mel_start4[mA] >> 2, qnA = T.mel_len4[mA], qoA = T.mel_off4[mA] >> 2;
    const int qsB = T.mel_start4[mB] >> 2, qnB = T.mel_len4[mB], qoB = T.mel_off4[mB] >> 2;
    int maxq = qnA > qnB ? qnA : qnB;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(maxq, o); maxq = t > maxq ? t : maxq; }
    __syncthreads();
    if (win >= n_windows) return;

    {
        const int frame = fg * fpb + wave;
        if (frame >= T.n_frames) return;   // wave-uniform; no workgroup barrier follows
        const long s_begin = (long)T.frame_origin + (long)frame * kHop;   // first sample of the frame

        // ---- load + window: lane holds z[n] = x[2n] + i x[2n+1] for n = lane + 64 m ----
        float2 v[16];
        const bool fast = aligned4 && s_begin >= 0 && (s_begin + kNfft) * 2 <= avail;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int n = lane + 64 * m;
            const float2 w = *reinterpret_cast<const float2 *>(T.window + 2 * n);
            float x0, x1;
            if (fast) {
                const uint32_t u = *reinterpret_cast<const uint32_t *>(wp + (s_begin + 2 * n) * 2);
                x0 = (float)(int16_t)(u & 0xFFFFu);
                x1 = (float)(int16_t)(u >> 16);
            } else {
                const long b = (s_begin + 2 * n) * 2;   // byte offset of x[2n]'s low byte
                int q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) q[j] = (b + j >= 0 && b + j < avail) ? (int)wp[b + j] : 0;
                x0 = (float)(int16_t)(uint16_t)(q[0] | (q[1] << 8));
                x1 = (float)(int16_t)(uint16_t)(q[2] | (q[3] << 8));
            }
            // normalizeAudio: float32(v) / 32768 (exact), then * window
            v[m] = make_float2(x0 * (1.0f / 32768.0f) * w.x, x1 * (1.0f / 32768.0f) * w.y);
        }

        // ---- pass 1: radix 16, Ns = 1 (no twiddles); out[16 j + t] at pidx(16 lane + t); pass 2 reads x[j + 64 t] ----
        dft16(v);
        const int rbase = lane + (lane >> 4);
        APR_FE_EXCHANGE(17 * lane + t, dft16_slot(t), rbase + 68 * t)

        // ---- pass 2: radix 16, Ns = 16; x[j + 64 t] * W_256^(k t), out[(j/16)*256 + k + 16 t]; pass 3 reads x[j' + 256 t] ----
        {
            const int k = lane & 15;
#pragma unroll
            for (int t = 1; t < 16; t++) v[t] = cmul(v[t], s_tw23[16 * t + k]);
            dft16(v);
            const int wbase = (lane >> 4) * 272 + k;
            APR_FE_EXCHANGE(wbase + 17 * t, dft16_slot(t), rbase + 68 * t)
        }

        // ---- pass 3: radix 4, Ns = 256; j' = lane + 64 q: x[j' + 256 t] * W_1024^(j' t); Z[k] for k = lane + 64 m ends in v[m] ----
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int jp = lane + 64 * q;
            v[q + 4] = cmul(v[q + 4], s_tw23[256 + jp]);
            v[q + 8] = cmul(v[q + 8], s_tw23[512 + jp]);
            v[q + 12] = cmul(v[q + 12], s_tw23[768 + jp]);
            dft4(v[q], v[q + 4], v[q + 8], v[q + 12]);
        }

        // ---- packed-real post-process: X[k] = (Z[k] + conj Z[N-k])/2 - i/2 W_2048^k (Z[k] - conj Z[N-k]).
        //      Z[N-k] sits in lane (64 - lane) & 63, register 15 - m (lane 0: its own register (16 - m) & 15): a lane
        //      permute, no trip through the plane ----
        float pw[16];
        float p_nyq = 0.0f;
        const int lsrc = (64 - lane) & 63;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int k = lane + 64 * m;
            float zx = __shfl(v[15 - m].x, lsrc), zy = __shfl(v[15 - m].y, lsrc);
            if (lane == 0) { zx = v[(16 - m) & 15].x; zy = v[(16 - m) & 15].y; }
            const float2 zr = make_float2(zx, -zy);   // conj Z[N-k]
            const float2 e = cadd(v[m], zr), o = csub(v[m], zr);
            const float2 wo = mul_mi(cmul(s_tw[k], o));
            const float xr = 0.5f * (e.x + wo.x), xi = 0.5f * (e.y + wo.y);
            pw[m] = xr * xr + xi * xi;
            if (m == 0) { const float d = v[0].x - v[0].y; p_nyq = d * d; }   // only lane 0's is bin 1024
            if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four bins at a time: keeps the permutes and twiddles of all 16 from being live together
        }
        // power spectrum P[0..1024] into the plane (unpadded); the quads of the last bands may reach a little further: zeros
#pragma unroll
        for (int m = 0; m < 16; m++) pl[lane + 64 * m] = pw[m];
        if (lane == 0) pl[1024] = p_nyq;
        if (lane >= 1 && lane < 8) pl[1024 + lane] = 0.0f;
        wave_lds_sync();

        // ---- sparse mel projection (dsp.go:83-111): k-ascending sequential sum per band.  Four bins per LDS read on
        //      both operands; the padding weights are +0, and fma(+0, p, acc) = acc exactly, so the sum is the
        //      unpadded one bit for bit ----
        float accA = 0.0f, accB = 0.0f;
        const float4 *P4 = reinterpret_cast<const float4 *>(pl);
        const float4 *W4 = reinterpret_cast<const float4 *>(s_w);
#ifdef APR_FE_ABL_NOMEL   // developer timing ablation (wrong results): what the mel projection costs inside the kernel
        if (pl[lane] == 12345.678f)
#endif
        for (int i = 0; i < maxq; i++) {
            if (i < qnA) {
                const float4 p = P4[qsA + i], w = W4[qoA + i];
                accA = fmaf(w.x, p.x, accA); accA = fmaf(w.y, p.y, accA); accA = fmaf(w.z, p.z, accA); accA = fmaf(w.w, p.w, accA);
            }
            if (i < qnB) {
                const float4 p = P4[qsB + i], w = W4[qoB + i];
                accB = fmaf(w.x, p.x, accB); accB = fmaf(w.y, p.y, accB); accB = fmaf(w.z, p.z, accB); accB = fmaf(w.w, p.w, accB);
            }
        }
        float *dst = melT + ((size_t)win * kSpecFrames + frame) * kNmels;
        dst[mA] = accA;
        dst[mB] = accB;
        wave_lds_sync();
    }
}
#undef APR_FE_EXCHANGE

// ---------------------------------------------------------------------------------------------
// Kernel 2: dB + normalise + transpose.  powerToDb / fixedNormalize (dsp.go:114-155) and the pack
// loop of model.go:157-171 (go flavour, evaluated in f64 like the reference), or
// librosa.power_to_db(ref=np.max) + clip + scale (ai/model.py:197-201, f32 like numpy).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void db_norm_kernel(const float *__restrict__ melT, int n_windows, int n_frames,
                                                      int librosa, float *__restrict__ out /*[n][128][188]*/)
{
    __shared__ float s_red[4];
    __shared__ float s_tile[kNmels][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int win = blockIdx.x;
    if (win >= n_windows) return;
    const float *src = melT + (size_t)win * kSpecFrames * kNmels;
    float *dst = out + (size_t)win * kNmels * kSpecFrames;

    // global max over [n_frames][128]
    float mx = 0.0f;
    const int n4 = n_frames * kNmels / 4;
#pragma unroll 8   // a single clip is one workgroup per column chunk: without the unroll, 23 dependent round trips
    for (int i = tid; i < n4; i += 256) {
        const float4 q = reinterpret_cast<const float4 *>(src)[i];
        mx = fmaxf(mx, fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));

    const double maxd = fmax((double)mx, 1e-10);                  // dsp.go:138
    const float amin = 1e-10f;
    const float ref_db = 10.0f * log10f(fmaxf(mx, amin));         // librosa: 10*log10(max(amin, ref))
    // librosa top_db: log_spec.max() - 80 with log_spec.max() = 10*log10(max(amin,mx)) - ref_db = 0
    // gridDim.y == 1: this block walks all six 32-frame column chunks; gridDim.y == 6 (single clips): one each
    const int t_step = 32 * (int)gridDim.y;
    for (int t0 = 32 * (int)blockIdx.y; t0 < kSpecFrames; t0 += t_step) {
        __syncthreads();
        // load 32 frames x 128 bands, convert, stage transposed
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + 256 * i;          // 0..4095
            const int f = e >> 7, m = e & 127;
            const int frame = t0 + f;
            float r = 0.0f;
            if (frame < n_frames) {
                const float v = src[(size_t)frame * kNmels + m];
                if (!librosa) {
                    double vd = (double)v;
                    if (vd < 1e-10) vd = 1e-10;
                    double db = 10.0 * log10(vd / maxd);
                    if (db < -80.0) db = -80.0;
                    if (db > 0.0) db = 0.0;
                    r = (float)((db + 80.0) / 80.0);
                } else {
                    float db = 10.0f * log10f(fmaxf(v, amin)) - ref_db;
                    db = fmaxf(db, 0.0f - 80.0f);
                    db = fminf(fmaxf(db, -80.0f), 0.0f);
                    r = (db + 80.0f) / 80.0f;
                }
            }
            s_tile[m][f] = r;
        }
        __syncthreads();
        // write [128 rows][32 cols] as float4: 8 lanes per row
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (tid >> 3) + 32 * i, c4 = (tid & 7) * 4;
            const int col = t0 + c4;
            if (col < kSpecFrames) {   // 188 % 4 == 0: whole float4 in range
                float4 q = make_float4(s_tile[row][c4], s_tile[row][c4 + 1], s_tile[row][c4 + 2], s_tile[row][c4 + 3]);
                *reinterpret_cast<float4 *>(dst + (size_t)row * kSpecFrames + col) = q;
            }
        }
    }
}

}  // namespace apr
