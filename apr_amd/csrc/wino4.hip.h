// wino4.hip.h -- Winograd F(4x4, 3x3) transforms around the batched GEMM mode of conv3x3_igemm_f32
// (conv.hip.h, MODE 1 with 36 positions), for the stride-1 layers of the NoiseDetector CNN
// (ai/model.py:67-96, executed by src/model/model.go:188-209).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   with the interpolation points 0, +-1, +-2, inf (Lavin & Gray):
//   36 multiplies per 4x4 outputs instead of 144 (F(2x2): 64): 1.78x fewer MFMA FLOPs than F(2x2,3x3).
//   fp32 error, measured before this was built (tools/wino_f4_check.py, the reference's own network in
//   torch fp32 with layers 5/7/9/11 in this form): logits within 3e-6 of the reference's on the golden
//   clips against 1.2e-6 for the direct form and 2.4e-6 for F(2x2); the parity tolerance is 2e-4.
//
//   wino_input_f4   d (NHWC activations)  -> V[36][T][Cin]      T = B * ceil(Ho/4) * ceil(Wo/4)
//   conv3x3_igemm_f32<..., MODE 1>        -> M[36][T][Cout] = V[pos] x U[pos]
//   wino_output_f4  M -> out NHWC, + bias, ReLU
// U[36][Cin][Cout] = G g G^T is computed on the host in f64 and rounded once (wino4_pack_u).
#pragma once
#include <hip/hip_runtime.h>

#include "wino.hip.h"

namespace apr {

// G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
inline void wino4_pack_u(const float *w /*[cout][cin][3][3]*/, int cin, int cout, float *u /*[36][cin][cout]*/)
{
    static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int o = 0; o < cout; o++)
        for (int c = 0; c < cin; c++) {
            const float *g = &w[((size_t)o * cin + c) * 9];
            double t[6][3];
            for (int i = 0; i < 6; i++)
                for (int k = 0; k < 3; k++) t[i][k] = G[i][0] * g[k] + G[i][1] * g[3 + k] + G[i][2] * g[6 + k];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) {
                    const double v = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                    u[((size_t)(6 * i + j) * cin + c) * cout + o] = (float)v;
                }
        }
}

// one axis of B^T d:  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void wino4_bt(const f32x4 d0, const f32x4 d1, const f32x4 d2, const f32x4 d3, const f32x4 d4, const f32x4 d5,
                                         f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3, f32x4 &r4, f32x4 &r5)
{
    const f32x4 a = d4 - 4.0f * d2, b = d3 - 4.0f * d1;
    const f32x4 c = d4 - d2, e = 2.0f * (d3 - d1);
    r0 = 4.0f * d0 - 5.0f * d2 + d4;
    r1 = a + b;
    r2 = a - b;
    r3 = c + e;
    r4 = c - e;
    r5 = 4.0f * d1 - 5.0f * d3 + d5;
}

// in [B][H][W][C] -> V[36][T][C]; tile t = (b, ty, tx) covers input rows 4ty-1 .. 4ty+4 (pad 1).  One thread = one tile x 4 channels.
__global__ __launch_bounds__(256) void wino_input_f4(const float *__restrict__ in, float *__restrict__ V, int H, int W, int C, int th, int tw,
                                                     long long T)
{
    const int c4n = C >> 2;
    const long long total = T * c4n;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int c = (int)(g % c4n) * 4;
        const long long t = g / c4n;
        const int tx = (int)(t % tw);
        const long long r = t / tw;
        const int ty = (int)(r % th);
        const long long b = r / th;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        const float *base = in + (b * H * (long long)W) * C + c;
        // columns first: tmp[.][j] = B^T d[.][j]
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
        // rows: v[i][.] = tmp[i][.] B
#pragma unroll
        for (int i = 0; i < 6; i++) {
            f32x4 v[6];
            wino4_bt(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], v[0], v[1], v[2], v[3], v[4], v[5]);
            float *o = V + ((long long)(6 * i) * T + t) * C + c;
#pragma unroll
            for (int j = 0; j < 6; j++) st4(o + (long long)j * T * C, v[j]);
        }
    }
}

// one axis of A^T m:  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void wino4_at(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4, const f32x4 m5,
                                         f32x4 &y0, f32x4 &y1, f32x4 &y2, f32x4 &y3)
{
    const f32x4 sp = m1 + m2, dp = m1 - m2, sq = m3 + m4, dq = m3 - m4;
    y0 = m0 + sp + sq;
    y1 = dp + 2.0f * dq;
    y2 = sp + 4.0f * sq;
    y3 = dp + 8.0f * dq + m5;
}

// M[36][T][C] -> out [B][Ho][Wo][C] with bias + ReLU
__global__ __launch_bounds__(256) void wino_output_f4(const float *__restrict__ Mx, const float *__restrict__ bias, float *__restrict__ out, int Ho,
                                                      int Wo, int C, int th, int tw, long long T)
{
    const int c4n = C >> 2;
    const long long total = T * c4n;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int c = (int)(g % c4n) * 4;
        const long long t = g / c4n;
        const int tx = (int)(t % tw);
        const long long r = t / tw;
        const int ty = (int)(r % th);
        const long long b = r / th;
        // columns first: tmp[a][j] = sum_i A^T[a][i] m[i][j]
        f32x4 tmp[4][6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            f32x4 m[6];
#pragma unroll
            for (int i = 0; i < 6; i++) m[i] = ld4(Mx + ((long long)(6 * i + j) * T + t) * C + c);
            wino4_at(m[0], m[1], m[2], m[3], m[4], m[5], tmp[0][j], tmp[1][j], tmp[2][j], tmp[3][j]);
        }
        const f32x4 bv = ld4(bias + c);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int y = 4 * ty + a;
            if (y >= Ho) continue;
            f32x4 yv[4];
            wino4_at(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], yv[0], yv[1], yv[2], yv[3]);
            float *o = out + ((b * Ho + y) * (long long)Wo + 4 * tx) * C + c;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (4 * tx + q < Wo) st4(o + (long long)q * C, __builtin_elementwise_max(yv[q] + bv, zero));
        }
    }
}

}  // namespace apr
