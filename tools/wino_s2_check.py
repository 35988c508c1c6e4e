#!/usr/bin/env python3
"""Numerical check (numpy, f32 vs f64) of the polyphase Winograd F(2,2) form of the stride-2 3x3 convolution
(pad 1) that apr_amd/csrc/wino_s2.hip.h implements (layers 6, 8, 10) -- developer tool, not product code.

    out[y][x] = sum_{dy,dx} w[dy][dx] in[2y + dy - 1][2x + dx - 1]

Per axis, outputs (2t, 2t+1) read input samples j = 0..4 at 4t - 1 + j.  The odd samples (j = 0, 2, 4) meet the
taps (w0, w2) as a 2-tap FIR -> F(2,2): 3 products; the even samples (j = 1, 3) meet w1: 2 products.
    V = B^T d :  [d0 - d2, d2, d4 - d2, d1, d3]          (0 / +-1 only)
    U = G w   :  [w0, w0 + w2, w2, w1, w1]               (rows 3 and 4 equal: 4 distinct)
    y = A^T m :  y0 = m0 + m1 + m3,  y1 = m1 + m2 + m4
2-D: 25 products per 2x2 outputs instead of 36; 16 distinct weight matrices.
"""
import numpy as np

BT = np.array([[1, 0, -1, 0, 0], [0, 0, 1, 0, 0], [0, 0, -1, 0, 1], [0, 1, 0, 0, 0], [0, 0, 0, 1, 0]], np.float64)
G = np.array([[1, 0, 0], [1, 0, 1], [0, 0, 1], [0, 1, 0], [0, 1, 0]], np.float64)
AT = np.array([[1, 1, 0, 1, 0], [0, 1, 1, 0, 1]], np.float64)


def direct(x, w, dt):
    H, W, C = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    xp = np.zeros((H + 2, W + 2, C), dt); xp[1:-1, 1:-1] = x
    out = np.zeros((Ho, Wo, w.shape[0]), dt)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("yxc,oc->yxo", xp[dy:dy + 2 * Ho:2, dx:dx + 2 * Wo:2].astype(dt), w[:, :, dy, dx].astype(dt))
    return out


def wino(x, w, dt):
    H, W, C = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    th, tw = (Ho + 1) // 2, (Wo + 1) // 2
    xp = np.zeros((4 * th + 4, 4 * tw + 4, C), dt); xp[1:H + 1, 1:W + 1] = x      # xp[r + 1] = in[r]
    U = np.einsum("pi,ocij,qj->pqoc", G, w.astype(np.float64), G).astype(dt)        # host, f64, rounded once
    out = np.zeros((2 * th, 2 * tw, w.shape[0]), dt)
    bt, at = BT.astype(dt), AT.astype(dt)
    for ty in range(th):
        for tx in range(tw):
            d = xp[4 * ty:4 * ty + 5, 4 * tx:4 * tx + 5].astype(dt)                 # rows 4ty-1 .. 4ty+3
            V = np.einsum("pi,ijc,qj->pqc", bt, d, bt)
            M = np.einsum("pqc,pqoc->pqo", V, U)
            out[2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ap,pqo,bq->abo", at, M, at)
    return out[:Ho, :Wo]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for (H, W, Cin, Cout) in ((8, 12, 32, 16), (7, 9, 16, 8), (16, 24, 64, 8), (5, 6, 8, 8)):
        x = np.maximum(rng.standard_normal((H, W, Cin)), 0).astype(np.float32)       # post-ReLU activations
        w = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
        ref = direct(x, w, np.float64)
        e_dir = np.abs(direct(x, w, np.float32) - ref).max()
        e_win = np.abs(wino(x, w, np.float32) - ref).max()
        assert np.abs(wino(x, w, np.float64) - ref).max() < 1e-12
        print(f"{H}x{W} {Cin}->{Cout}: max|ref| {np.abs(ref).max():.3f}  f32 direct err {e_dir:.2e}  f32 polyphase-Winograd err {e_win:.2e}")
