// wino.hip.h -- Winograd F(2x2, 3x3) transforms around the batched GEMM mode of
// conv3x3_igemm_f32 (conv.hip.h), for the stride-1 layers of the NoiseDetector CNN.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        (Lavin & Gray), 16 multiplies per 2x2 outputs
//   instead of 36: 2.25x fewer MFMA FLOPs than the direct convolution.  In fp32 its rounding
//   error is within ~2x of the direct form's (measured: 5e-7 vs 3e-7 of the output scale at
//   K = 9*512), so it stays inside the fp32 tolerance the parity tests state.
//
//   wino_input_f2   d (NHWC activations)  -> V[16][T][Cin]      T = B * ceil(Ho/2) * ceil(Wo/2)
//   conv3x3_igemm_f32<..., MODE 1>        -> M[16][T][Cout] = V[pos] x U[pos]
//   wino_output_f2  M -> out NHWC, + bias, ReLU
// U[16][Cin][Cout] = G g G^T is computed on the host in f64 (engine.hip).
// Both transform kernels are elementwise over channels (16 B per lane, coalesced) and HBM-bound.
#pragma once
#include <hip/hip_runtime.h>

#include "conv.hip.h"

namespace apr {

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// in [B][H][W][C] -> V[16][T][C]; tile t = (b, ty, tx) covers input rows 2ty-1 .. 2ty+2 (pad 1).
// One thread = one tile x 4 channels.  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].
__global__ __launch_bounds__(256) void wino_input_f2(const float *__restrict__ in, float *__restrict__ V, int H, int W,
                                                     int C, int th, int tw, long long T)
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
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        const float *base = in + (b * H * (long long)W) * C + c;
        f32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = y0 + i;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (y >= 0 && y < H && x >= 0 && x < W) v = ld4(base + ((long long)y * W + x) * C);
                d[i][j] = v;
            }
        }
        // rows: tmp = B^T d
        f32x4 tmp[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            tmp[0][j] = d[0][j] - d[2][j];
            tmp[1][j] = d[1][j] + d[2][j];
            tmp[2][j] = d[2][j] - d[1][j];
            tmp[3][j] = d[1][j] - d[3][j];
        }
        // columns: v = tmp B
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const f32x4 v0 = tmp[i][0] - tmp[i][2];
            const f32x4 v1 = tmp[i][1] + tmp[i][2];
            const f32x4 v2 = tmp[i][2] - tmp[i][1];
            const f32x4 v3 = tmp[i][1] - tmp[i][3];
            float *o = V + ((long long)(4 * i) * T + t) * C + c;
            st4(o, v0);
            st4(o + T * C, v1);
            st4(o + 2 * T * C, v2);
            st4(o + 3 * T * C, v3);
        }
    }
}

// M[16][T][C] -> out [B][Ho][Wo][C] with bias + ReLU.  A^T = [1 1 1 0; 0 1 -1 -1].
__global__ __launch_bounds__(256) void wino_output_f2(const float *__restrict__ Mx, const float *__restrict__ bias,
                                                      float *__restrict__ out, int Ho, int Wo, int C, int th, int tw,
                                                      long long T, int nsplit)
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
        f32x4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) m[i][j] = ld4(Mx + ((long long)(4 * i + j) * T + t) * C + c);
        // one-clip engines split the GEMMs along K: copies s >= 1 of M, added in split order
        //  (two copies' loads in flight at a time: the chain of dependent round trips is what this kernel costs there)
        for (int s = 1; s < nsplit; s += 2) {
            const bool two = s + 1 < nsplit;
            const float *Ma = Mx + (long long)s * 16 * T * C, *Mb = Mx + (long long)(two ? s + 1 : s) * 16 * T * C;
            f32x4 a[4][4], b[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    a[i][j] = ld4(Ma + ((long long)(4 * i + j) * T + t) * C + c);
                    b[i][j] = ld4(Mb + ((long long)(4 * i + j) * T + t) * C + c);
                }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    m[i][j] += a[i][j];
                    if (two) m[i][j] += b[i][j];
                }
        }
        // rows: tmp = A^T m  (2 x 4)
        f32x4 tmp[2][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            tmp[0][j] = m[0][j] + m[1][j] + m[2][j];
            tmp[1][j] = m[1][j] - m[2][j] - m[3][j];
        }
        const f32x4 bv = ld4(bias + c);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int y = 2 * ty + i;
            if (y >= Ho) continue;
            f32x4 y0 = tmp[i][0] + tmp[i][1] + tmp[i][2] + bv;
            f32x4 y1 = tmp[i][1] - tmp[i][2] - tmp[i][3] + bv;
            y0 = __builtin_elementwise_max(y0, zero);
            y1 = __builtin_elementwise_max(y1, zero);
            float *o = out + ((b * Ho + y) * (long long)Wo + 2 * tx) * C + c;
            st4(o, y0);
            if (2 * tx + 1 < Wo) st4(o + C, y1);
        }
    }
}

}  // namespace apr
