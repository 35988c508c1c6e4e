// wino_s2.hip.h -- the stride-2 3x3 convolution (pad 1) in polyphase Winograd F(2,2) form.
//
//   out[y][x] = sum_{dy,dx} w[dy][dx] in[2y + dy - 1][2x + dx - 1]
//
// Per axis, outputs (2t, 2t+1) read input samples j = 0..4 at 4t - 1 + j.  The odd-phase samples (j = 0, 2, 4)
// meet the taps (w0, w2) as a 2-tap filter -> F(2,2): 3 products for 2 outputs; the even-phase samples
// (j = 1, 3) meet w1 alone: 2 products.  Five "positions" per axis:
//     V = B^T d :  [d0 - d2, d2, d4 - d2, d1, d3]       (0 / +-1: exact in fp32)
//     U = G w   :  [w0, w0 + w2, w2, w1, w1]            (4 distinct)
//     y = A^T m :  y0 = m0 + m1 + m3,   y1 = m1 + m2 + m4
// In two dimensions 25 products per 2x2 outputs instead of 36 (1.44x fewer MFMAs), with 16 distinct weight
// matrices.  tools/wino_s2_check.py: fp32 error 4e-7..6e-7 against 3e-7..4e-7 of the direct form.
// The output transform only needs s0 = m0 + m3, s1 = m1, s2 = m2 + m4 per axis (y0 = s0 + s1, y1 = s1 + s2), and a
// sum of products over positions is one product over the concatenated K: the GEMM kernel accumulates the 25
// products into 3 x 3 sums S[qi][qj] (K = 4, 2 or 1 times Cin), so only 9 tensors go through HBM.
//   conv3x3_igemm_f32<BM, BN, 1, 3>  the 9 sums [T x k Cin] x [k Cin x Cout], the input transform formed in its A loader
//   wino_s2_output                   S[9][T][Cout] -> out NHWC: Y[a][b] = sum of the 2 x 2 block of S at (a, b), + bias, ReLU
// U is computed on the host in f64 and rounded once (wino_s2_pack_u).
#pragma once
#include <hip/hip_runtime.h>

#include "wino.hip.h"

namespace apr {

// S [9][T][C] -> out [B][Ho][Wo][C]; tile t = (b, ty, tx) holds outputs (2ty..2ty+1, 2tx..2tx+1).
// One thread = one tile x 4 channels.
__global__ __launch_bounds__(256) void wino_s2_output(const float *__restrict__ Mx, const float *__restrict__ bias,
                                                      float *__restrict__ out, int Ho, int Wo, int C, int th, int tw,
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
        f32x4 m[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) m[i][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Mx + ((long long)(3 * i + j) * T + t) * C + c));   // read once
        // rows: tmp[a] = s[a] + s[a + 1]
        f32x4 tmp[2][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            tmp[0][j] = m[0][j] + m[1][j];
            tmp[1][j] = m[1][j] + m[2][j];
        }
        const f32x4 bv = ld4(bias + c);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int y = 2 * ty + i;
            if (y >= Ho) continue;
            f32x4 y0 = tmp[i][0] + tmp[i][1] + bv;
            f32x4 y1 = tmp[i][1] + tmp[i][2] + bv;
            y0 = __builtin_elementwise_max(y0, zero);
            y1 = __builtin_elementwise_max(y1, zero);
            float *o = out + ((b * Ho + y) * (long long)Wo + 2 * tx) * C + c;
            st4(o, y0);
            if (2 * tx + 1 < Wo) st4(o + C, y1);
        }
    }
}

// Host: g [Cout][Cin][3][3] (OIHW, BN folded) -> U[4 ur + uc][Cin][Cout] = (G g G^T)[ur][uc],
// G = [1 0 0; 1 0 1; 0 0 1; 0 1 0] (positions 3 and 4 of an axis share row 3), in f64, rounded once.
inline void wino_s2_pack_u(const float *g, int Cin, int Cout, float *u)
{
    static const double G[4][3] = {{1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {0, 1, 0}};
    for (int o = 0; o < Cout; o++)
        for (int c = 0; c < Cin; c++) {
            const float *w = g + ((size_t)o * Cin + c) * 9;
            double t[4][3];
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 3; k++) t[i][k] = G[i][0] * w[k] + G[i][1] * w[3 + k] + G[i][2] * w[6 + k];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                    u[((size_t)(4 * i + j) * Cin + c) * Cout + o] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}

}  // namespace apr
