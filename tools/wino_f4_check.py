"""F(4x4,3x3) Winograd on the stride-1 layers, measured before anything is built for it.

CPU, torch fp32: the NoiseDetector forward with chosen layers computed as F(m x m, 3x3) Winograd in fp32 (input and output
transforms in fp32, U = G g G^T formed in f64 and rounded once, as the engine does), everything else as an fp32 direct
convolution.  Reports (a) logits against tests/golden/cnn_logits_he_seed0.npy (from the reference's own class) and
(b) label / match mismatches against the direct-convolution forward over bench.py's windows (first 599 windows of recording 0
+ the 59 windows of the config-1 clip), with the near-ties listed.

    python tools/wino_f4_check.py [--windows 599] [--layers 5,7,9,11] [--m 4] [--points std|half]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from apr_amd import model, weights as W   # noqa: E402
import oracle                              # noqa: E402  (checker: spectrograms of the windows)

LOGIT_TOL, NEAR_TIE, TRUST = 2e-4, 4e-4, 0.5


def winograd_matrices(m, points):
    """Toom-Cook matrices AT [m, a], G [a, 3], BT [a, a] for F(m, 3), a = m + 2, in f64 (exact rationals as floats)."""
    from fractions import Fraction as Fr
    a = m + 2
    pts = [Fr(p) for p in points]          # a - 1 finite points, the last one is infinity
    assert len(pts) == a - 1
    # polynomial interpolation form (Lavin & Gray / wincnn): AT[i, j] = p_j^i, G rows = [1, p, p^2] / prod(p_j - p_k), BT from the Lagrange basis
    def poly_mul(p, q):
        r = [Fr(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q):
                r[i + j] += x * y
        return r
    AT = [[pts[j] ** i for j in range(a - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    G = []
    for j in range(a - 1):
        f = Fr(1)
        for k in range(a - 1):
            if k != j:
                f *= (pts[j] - pts[k])
        G.append([Fr(1) / f, pts[j] / f, pts[j] ** 2 / f])
    G.append([Fr(0), Fr(0), Fr(1)])
    BT = []
    for j in range(a - 1):
        p = [Fr(1)]
        for k in range(a - 1):
            if k != j:
                p = poly_mul(p, [-pts[k], Fr(1)])
        BT.append(p + [Fr(0)] * (a - len(p)))
    full = [Fr(1)]
    for k in range(a - 1):
        full = poly_mul(full, [-pts[k], Fr(1)])
    BT.append(full)
    to = lambda M: np.array([[float(x) for x in r] for r in M], np.float64)
    return to(AT), to(G), to(BT)


def wino_conv(x, w, b, m, mats):
    """x [B,Cin,H,W] f32, w [Cout,Cin,3,3] f32 (BN folded), stride 1 pad 1 -> [B,Cout,H,W] f32, all arithmetic fp32."""
    AT, G, BT = mats
    a = m + 2
    B_, Cin, H, Wd = x.shape
    th, tw = -(-H // m), -(-Wd // m)
    xp = F.pad(x, (1, tw * m + 1 - Wd, 1, th * m + 1 - H))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                      # [B,Cin,th,tw,a,a]
    BTt = torch.from_numpy(BT.astype(np.float32))
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", BTt, tiles, BTt)       # B^T d B
    U = torch.from_numpy(np.einsum("ij,ocjk,lk->ocil", G, w.double().numpy(), G).astype(np.float32))   # f64, rounded once
    Mm = torch.einsum("bcyxil,ocil->boyxil", V, U)
    ATt = torch.from_numpy(AT.astype(np.float32))
    Y = torch.einsum("ij,boyxjk,lk->boyxil", ATt, Mm, ATt)          # [B,Cout,th,tw,m,m]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, -1, th * m, tw * m)[:, :, :H, :Wd]
    return Y + b[None, :, None, None]


def forward(specs, convs, fc_w, fc_b, wino_layers, m, mats, chunk=8):
    out = []
    for i0 in range(0, specs.shape[0], chunk):
        x = torch.from_numpy(specs[i0:i0 + chunk]).unsqueeze(1)
        for l, (w, b, s) in enumerate(convs):
            if (l + 1) in wino_layers and s == 1:
                x = wino_conv(x, w, b, m, mats)
            else:
                x = F.conv2d(x, w, b, stride=s, padding=1)
            x = torch.relu(x)
        out.append((torch.amax(x, dim=(2, 3)) @ fc_w.T + fc_b).numpy())
    return np.concatenate(out)


def folded(flat, C):
    sd = W.to_state_dict(flat, C)
    convs = []
    for l in range(12):
        ci, bi = 3 * l, 3 * l + 1
        w = sd[f"features.{ci}.weight"].astype(np.float64); b = sd[f"features.{ci}.bias"].astype(np.float64)
        g = sd[f"features.{bi}.weight"].astype(np.float64); be = sd[f"features.{bi}.bias"].astype(np.float64)
        mu = sd[f"features.{bi}.running_mean"].astype(np.float64); va = sd[f"features.{bi}.running_var"].astype(np.float64)
        sc = g / np.sqrt(va + 1e-5)                                   # f64 fold, rounded once: csrc/model_parse.cpp
        convs.append((torch.from_numpy((w * sc[:, None, None, None]).astype(np.float32)), torch.from_numpy(((b - mu) * sc + be).astype(np.float32)), W.LAYER_STRIDE[l]))
    return convs, torch.from_numpy(sd["classifier.3.weight"].copy()), torch.from_numpy(sd["classifier.3.bias"].copy())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=599)
    ap.add_argument("--layers", default="5,7,9,11")
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--points", default="std")
    args = ap.parse_args()
    torch.set_num_threads(8)
    layers = {int(v) for v in args.layers.split(",") if v}
    pts = {(4, "std"): [0, 1, -1, 2, -2], (4, "half"): [0, 1, -1, "1/2", "-1/2"], (4, "mix"): [0, 1, -1, "1/2", -2], (2, "std"): [0, 1, -1]}[(args.m, args.points)]
    mats = winograd_matrices(args.m, pts)
    C = 4
    flat = W.random_init(C, 0, "he")
    golden = os.path.join(REPO, "tests", "golden")
    specs = np.load(os.path.join(golden, "cnn_input_specs.npy"))
    want = np.load(os.path.join(golden, "cnn_logits_he_seed0.npy"))
    convs, fw, fb = folded(flat, C)
    with torch.no_grad():
        d = forward(specs, convs, fw, fb, set(), args.m, mats)
        g = forward(specs, convs, fw, fb, layers, args.m, mats)
    print(f"golden clips: direct fp32 vs reference logits max abs {np.abs(d - want).max():.3e}; F({args.m},3)[{args.points}] on layers {sorted(layers)} {np.abs(g - want).max():.3e} (tolerance {LOGIT_TOL})", flush=True)
    if args.windows <= 0:
        return
    # bench.py's windows: recording 0 (first N windows) with the centred classifier, then the config-1 clip
    e_cal = oracle.prepare_go_batch(model.synth_pcm(0, 0, 65), 64)
    with torch.no_grad():
        cal = forward(e_cal, convs, fw, fb, set(), args.m, mats)
    flat = W.center_logits(flat, C, cal.mean(axis=0))
    convs, fw, fb = folded(flat, C)
    for name, pcm, n in (("recording 0", model.synth_pcm(0, 0, args.windows + 1), args.windows), ("config-1 clip", model.synth_pcm(1, 0, 60), 59)):
        sp = oracle.prepare_go_batch(pcm, n)
        with torch.no_grad():
            ld = forward(sp, convs, fw, fb, set(), args.m, mats)
            lw = forward(sp, convs, fw, fb, layers, args.m, mats)
        pd_, pw = torch.softmax(torch.from_numpy(ld).double(), 1).numpy(), torch.softmax(torch.from_numpy(lw).double(), 1).numpy()
        labd, labw = ld.argmax(1), lw.argmax(1)
        md = (labd != 2) & (pd_.max(1) > TRUST); mw = (labw != 2) & (pw.max(1) > TRUST)
        srt = np.sort(ld, axis=1); decided = (srt[:, -1] - srt[:, -2]) > NEAR_TIE
        near_trust = np.abs(pd_.max(1) - TRUST) < 1e-3
        err = np.abs(lw - ld)
        print(f"{name}: {n} windows; logit err max {err.max():.3e} mean {err.mean():.3e}; over {LOGIT_TOL}: {(err.max(1) > LOGIT_TOL).sum()} windows; "
              f"label mismatches {int((labd != labw).sum())} (on decided windows {int((labd != labw)[decided].sum())}); match mismatches {int((md != mw).sum())} "
              f"(away from ties/threshold {int((md != mw)[decided & ~near_trust].sum())}); near-ties {int((~decided).sum())}; logit scale {np.abs(ld).max():.2f}", flush=True)


if __name__ == "__main__":
    main()
