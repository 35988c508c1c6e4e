"""CPU oracle for apr_resample_taps / apr_upsample (TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else).

The reference has no code for this step: both of its decode commands ask ffmpeg for `-ar 48000 -ac 1`
(src/ffmpeg/ffmpeg.go:76,113-114; ai/inspect.py:99-100) and libswresample does the work.  There is
therefore no golden vector: PARITY UNPINNED by construction (SURVEY.md section 8(d) "outside parity").
What this file pins is the library against its own published definition (include/apr.h).
"""
import math

import numpy as np


def bessel_i0(x: float) -> float:
    h = 0.5 * x
    term, total = 1.0, 1.0
    for k in range(1, 500):
        f = h / k
        term = term * f * f
        total = total + term
        if term < 1e-21 * total:
            break
    return total


def resample_taps(up: int = 3, tpp: int = 32, cutoff: float = 0.97, beta: float = 9.0) -> np.ndarray:
    """h[j] = g(j - c), c = up * tpp / 2, g(t) = cutoff * sinc(cutoff * t / up) * kaiser_beta(t / c); phase p = h[up k + p] / sum."""
    c = up * tpp // 2
    i0b = bessel_i0(beta)
    taps = np.zeros((up, tpp))
    for p in range(up):
        for k in range(tpp):
            t = float(up * k + p - c)
            a = cutoff * t / up
            sinc = 1.0 if t == 0.0 else math.sin(math.pi * a) / (math.pi * a)
            u = t / c
            w = bessel_i0(beta * math.sqrt(1.0 - u * u)) / i0b
            taps[p, k] = cutoff * sinc * w
        taps[p] = taps[p] / _seq_sum(taps[p])
    return taps


def _seq_sum(v) -> float:
    s = 0.0
    for x in v:
        s = s + float(x)
    return s


def upsample(x, up: int, taps: np.ndarray) -> np.ndarray:
    """y[up m + p] = rint(sum_k taps[p][k] * x[m + tpp/2 - k]) saturated to int16; products and sums rounded one by one in k order."""
    x = np.asarray(x, np.int16).astype(np.float64)
    n, tpp = x.size, taps.shape[1]
    pad = tpp
    xp = np.concatenate([np.zeros(pad), x, np.zeros(pad)])
    y = np.empty(n * up, np.int16)
    m = np.arange(n)
    for p in range(up):
        acc = np.zeros(n)
        for k in range(tpp):
            acc = acc + taps[p, k] * xp[m + pad + tpp // 2 - k]
        y[p::up] = np.clip(np.rint(acc), -32768, 32767).astype(np.int16)
    return y
