"""CPU: the oracle against the golden vectors (from the reference's own NoiseDetector) and the
hand-derivable known answers of SURVEY 8(c).  No GPU, no product code under test here."""
import os

import numpy as np
import pytest

import oracle
from apr_amd import weights as W
from conftest import GOLDEN


def test_KA6_normalize_audio_matches_reference_fixture():
    # fixture produced by the reference's ai.model.normalize_audio (ai/model.py:163-168)
    pcm = np.array([-32768, -1, 0, 1, 32767, 12345], dtype="<i2").tobytes()
    want = np.load(os.path.join(GOLDEN, "normalize_audio.npy"))
    got = oracle.normalize_audio(pcm)
    assert np.array_equal(got, want)
    assert got[0] == -1.0 and got[2] == 0.0 and got[4] == np.float32(32767 / 32768)


def test_KA1_silence():
    out, warn = oracle.prepare_go(np.zeros(192000, np.uint8))
    assert warn == 0
    assert np.all(out[:, :184] == 1.0) and np.all(out[:, 184:] == 0.0)


def test_KA2_short_equals_zero_padded_long_equals_truncated(clips):
    pcm = clips["test_combo"]
    a, wa = oracle.prepare_go(pcm[:77777])
    b, wb = oracle.prepare_go(np.concatenate([pcm[:77777], np.zeros(192000 - 77777, np.uint8)]))
    assert wa == oracle.WARN_UNDERFLOW and wb == 0 and np.array_equal(a, b)
    c, wc = oracle.prepare_go(np.concatenate([pcm, pcm]))
    assert wc == oracle.WARN_OVERFLOW and np.array_equal(c, oracle.prepare_go(pcm)[0])


def test_KA5_mel_lens_integers():
    w, bounds, pts = oracle.mel_lens()
    assert pts[:20].tolist() == [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17, 19, 20]
    assert pts[-5:].tolist() == [914, 940, 967, 995, 1024]
    assert int((w != 0).sum()) == 1892
    assert int((bounds[:, 1] - bounds[:, 0]).max()) == 57
    assert int(((w != 0).sum(axis=1) == 0).sum()) == 0
    assert np.all(w[:, 1024] == 0)                 # bin 1024 never weighted (loops are `< end`)
    assert w.max() == 1.0                          # unnormalised peak 1.0
    # falling edge includes the centre bin with weight 1, rising edge starts at 0 (dsp.go:70-78)
    m = 50
    assert w[m, pts[m + 1]] == 1.0 and w[m, pts[m]] == 0.0


def test_KA3_full_scale_dc():
    # DC -> Hann-windowed energy in bins 0..1 only; bands covering bins 0/1 carry everything
    pcm = np.full(96000, 32767, dtype="<i2").view(np.uint8)
    out, _, power, mel = oracle.prepare_go(pcm, stages=True)
    col = power[:, 0]
    # the symmetric (L-1) Hann of go-dsp is not exactly periodic: bin 2 leaks at ~ -76 dB
    assert col[0] > 0 and col[2] / col[0] < 1e-7
    hann = oracle.hann()[0]
    x = np.float64(np.float32(32767) / np.float32(32768))
    assert np.isclose(col[0], (x * hann.sum()) ** 2, rtol=1e-12)
    assert np.isclose(col[1] / col[0], 0.25, rtol=2e-3)       # Hann leakage: X[1] ~ -X[0]/2
    assert out[:, :184].max() == 1.0
    # band 0 = bin 0 (weight 1) -> 0 dB -> 1.0; band 1 = bin 1 -> 10*log10(0.2504) = -6.01 dB
    assert out[0, 0] == 1.0 and abs(out[1, 0] - (80 + 10 * np.log10(col[1] / col[0])) / 80) < 1e-6
    assert np.all(out[3:, :184] == 0.0)                        # everything else is below the -80 dB floor


def test_KA4_bin_centred_sine():
    k = 200                                                    # f = 48000*k/2048 = 4687.5 Hz
    n = np.arange(96000)
    s = np.round(16000 * np.sin(2 * np.pi * k * n / 2048)).astype("<i2")
    out, _, power, mel = oracle.prepare_go(s.view(np.uint8), stages=True)
    col = power[:, 10]
    assert int(np.argmax(col)) == k
    assert np.isclose(col[k - 1] / col[k], 0.25, rtol=2e-3) and np.isclose(col[k + 1] / col[k], 0.25, rtol=2e-3)
    assert col[k + 3] / col[k] < 1e-6
    _, bounds, _ = oracle.mel_lens()
    covering = [m for m in range(128) if bounds[m, 0] <= k < bounds[m, 1]]
    assert int(np.argmax(mel[:, 10])) in covering


def _mel_lens_from_dsp_go():
    """The go mel lens a second time, in numpy, straight from src/model/dsp.go:31-80 and 158-165 -- written
    without looking at oracle/apr_oracle.c, so a misreading there would show as a difference."""
    nbins = 2048 // 2 + 1
    hz_to_mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)                 # dsp.go:158-160
    mel_to_hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)               # dsp.go:163-165
    lo, hi = hz_to_mel(0.0), hz_to_mel(24000.0)
    step = (hi - lo) / (128 + 1)
    pts_hz = np.array([mel_to_hz(lo + i * step) for i in range(130)])
    bins = np.floor((2048 + 1) * pts_hz / 48000.0).astype(int)              # dsp.go:48
    w = np.zeros((128, nbins))
    for i in range(128):
        start, center, end = bins[i], bins[i + 1], bins[i + 2]
        for f in range(start, center):                                       # rising edge: starts at weight 0
            if f >= nbins:
                break
            w[i, f] = (f - start) / (center - start)
        for f in range(center, end):                                         # falling edge: centre bin has weight 1
            if f >= nbins:
                break
            w[i, f] = (end - f) / (end - center)
    return w, bins


def test_mel_lens_equals_an_independent_numpy_restatement_of_dsp_go():
    w_np, bins_np = _mel_lens_from_dsp_go()
    w, bounds, pts = oracle.mel_lens()
    assert np.array_equal(pts, bins_np)
    assert np.array_equal(bounds[:, 0], bins_np[:128]) and np.array_equal(bounds[:, 1], bins_np[2:])
    assert np.array_equal(w, w_np)                                           # f64 quotients of small integers: bit-equal
    assert int((w_np != 0).sum()) == 1892


def test_window_and_frame_grids_against_scipy():
    # librosa flavour: librosa.stft's default window is scipy.signal.get_window('hann', n_fft, fftbins=True) (periodic);
    # go flavour: go-dsp window.Hann(L) is the symmetric form (divisor L-1) = get_window(..., fftbins=False)
    from scipy.signal import get_window
    sym, per = oracle.hann()
    np.testing.assert_allclose(per, get_window("hann", 2048, fftbins=True), rtol=0, atol=1e-15)
    np.testing.assert_allclose(sym, get_window("hann", 2048, fftbins=False), rtol=0, atol=1e-15)
    assert sym[0] == 0.0 and abs(sym[-1]) < 1e-15 and per[0] == 0.0 and per[-1] > 0.0
    # frame grids: go = 184 uncentred frames (model.go:106), librosa = 1 + 96000 // 512 = 188 frames centred on 512 t
    assert (96000 - 2048) // 512 + 1 == 184 and 1 + 96000 // 512 == 188
    # centred, zero-padded framing (librosa >= 0.10 pad_mode="constant"): an impulse at sample 512 t sits at the centre of
    # frame t, where the periodic Hann is exactly 1 -> a flat spectrum of power 1 in that frame and in no frame further than
    # 2 hops away
    x = np.zeros(96000, np.float32)
    x[512 * 50] = 1.0
    ref = np.pad(x.astype(np.float64), 1024)
    fr = np.stack([ref[512 * t:512 * t + 2048] * per for t in range(188)])
    P = np.abs(np.fft.rfft(fr, axis=1)) ** 2                                # [188][1025]
    assert np.allclose(P[50], 1.0) and P[47].max() == 0.0 and P[53].max() == 0.0
    M = oracle.librosa_mel_basis().astype(np.float64) @ P.T                 # [128][188]
    db = 10 * np.log10(np.maximum(M, 1e-10)) - 10 * np.log10(max(M.max(), 1e-10))
    want = ((np.clip(np.maximum(db, db.max() - 80), -80, 0) + 80) / 80).astype(np.float32)
    got = oracle.prepare_librosa(x)
    assert got.shape == (128, 188) and np.abs(got - want).max() < 2e-6


def test_numpy_cross_check_of_the_go_front_end(clips, ramp_pcm):
    # independent formulation (numpy FFT, dense matmul with the numpy lens above); not a pin of the reference itself
    w, _ = _mel_lens_from_dsp_go()
    hann = 0.5 * (1 - np.cos(2 * np.pi * np.arange(2048) / 2047))
    for pcm in (clips["test_bigdog"], clips["test_empty"], ramp_pcm):
        x = (pcm.view("<i2").astype(np.float32) / np.float32(32768)).astype(np.float64)
        fr = np.stack([x[512 * i:512 * i + 2048] * hann for i in range(184)])
        P = (np.abs(np.fft.fft(fr, axis=1))[:, :1025] ** 2).T
        M = w @ P
        db = 10 * np.log10(np.maximum(M, 1e-10) / max(1e-10, M.max()))
        ref = ((np.clip(db, -80, 0) + 80) / 80).astype(np.float32)
        out, _ = oracle.prepare_go(pcm)
        assert np.abs(out[:, :184] - ref).max() < 1e-6 and np.all(out[:, 184:] == 0)


def test_librosa_flavour_restatement_properties(clips):
    # parity UNPINNED (librosa absent): structural checks only
    basis = oracle.librosa_mel_basis()
    assert basis.shape == (128, 1025) and basis.dtype == np.float32
    assert np.all(basis >= 0) and np.all((basis != 0).sum(axis=1) > 0)
    _, pts_go = oracle.mel_lens()[1], None
    out = oracle.prepare_librosa(oracle.normalize_audio(clips["test_smalldog"]))
    assert out.shape == (128, 188) and out.min() >= 0.0 and out.max() == 1.0
    assert np.all(oracle.prepare_librosa(np.zeros(96000, np.float32)) == 1.0)
    # the two reference flavours really are different functions
    go, _ = oracle.prepare_go(clips["test_smalldog"])
    assert np.abs(go[:, 2:182] - out[:, 4:184]).max() > 1e-3


def test_librosa_flavour_against_an_independent_implementation(clips):
    # librosa itself is absent, so this flavour stays "parity unpinned" -- but `transformers.audio_utils` (installed here, written to reproduce
    # librosa) is a third party's restatement of the same functions: the mel basis with the reference's arguments (ai/model.py:28-34: htk=True,
    # the default Slaney area norm, 0 .. sr/2) and the whole of audio_to_spectrogram (ai/model.py:184-204: centred zero-padded STFT with the
    # periodic Hann window, power, mel, power_to_db(ref=max, top_db 80), clip, (x + 80) / 80).
    import sys
    stub = sys.modules.get("librosa")       # another test's stand-in for the absent package (no __spec__) would trip transformers' availability probe
    if stub is not None and getattr(stub, "__spec__", None) is None:
        del sys.modules["librosa"]
    try:
        A = pytest.importorskip("transformers.audio_utils")
    finally:
        if stub is not None:
            sys.modules.setdefault("librosa", stub)
    mel = A.mel_filter_bank(1025, 128, 0.0, 24000.0, 48000, norm="slaney", mel_scale="htk")          # [1025][128]
    basis = oracle.librosa_mel_basis()
    assert np.abs(mel.T - basis).max() <= 1e-8                                                         # f32 rounding of the oracle's table
    # the Slaney SCALE (librosa's default without htk=True) is another matrix altogether: the argument matters
    assert np.abs(A.mel_filter_bank(1025, 128, 0.0, 24000.0, 48000, norm="slaney", mel_scale="slaney").T - basis).max() > 1e-2
    win = A.window_function(2048, "hann", periodic=True)
    rng = np.random.default_rng(3)
    t = np.arange(96000) / 48000.0
    xs = [oracle.normalize_audio(clips[k]) for k in ("test_smalldog", "test_combo")]
    xs.append((0.1 * rng.standard_normal(96000) + 0.3 * np.sin(2 * np.pi * 440 * t) * np.exp(-3 * t)).astype(np.float32))
    for x in xs:
        S = A.spectrogram(x.astype(np.float64), win, 2048, 512, power=2.0, center=True, pad_mode="constant",
                          mel_filters=mel.astype(np.float64), mel_floor=1e-10, dtype=np.float64)
        db = A.power_to_db(S, reference=float(S.max()), min_value=1e-10, db_range=80.0)
        want = (np.clip(db, -80.0, 0.0) + 80.0) / 80.0
        got = oracle.prepare_librosa(x)
        assert got.shape == want.shape == (128, 188)
        assert np.abs(got - want).max() <= 2e-6


@pytest.mark.parametrize("init,seed,tol", [("he", 0, 2e-4), ("torch", 1, 1e-6)])
def test_cnn_matches_reference_noisedetector_golden(init, seed, tol):
    specs = np.load(os.path.join(GOLDEN, "cnn_input_specs.npy"))
    want = np.load(os.path.join(GOLDEN, f"cnn_logits_{init}_seed{seed}.npy"))
    flat = W.random_init(4, seed, init)
    got = oracle.cnn_forward(flat, 4, specs)
    assert np.abs(got - want).max() <= tol
    pooled = np.load(os.path.join(GOLDEN, f"cnn_pooled_{init}_seed{seed}.npy"))
    _, feat = oracle.cnn_forward_feat(flat, 4, specs[2])
    assert np.abs(feat - pooled[2]).max() <= tol * max(1.0, float(np.abs(pooled[2]).max()))


def test_KA7_softmax_and_decision():
    p = oracle.softmax(np.array([1.5, 1.5, 1.5, 1.5], np.float32))
    assert np.allclose(p, 0.25) and abs(p.sum() - 1) < 1e-15
    p = oracle.softmax(np.array([1000.0, 0.0, -1000.0], np.float32))
    assert p[0] == 1.0 and p[2] == 0.0
    # daemon.go:203: match iff best != "empty" and conf > trust (strict)
    assert oracle.decide([0.2, 0.5, 0.3], 2, 0.5) == (1, 0.5, False)
    assert oracle.decide([0.2, 0.6, 0.2], 2, 0.5) == (1, 0.6, True)
    assert oracle.decide([0.2, 0.1, 0.7], 2, 0.5) == (2, 0.7, False)
    assert oracle.decide([0.0, 0.0, 0.0], 2, 0.5)[0] == -1


def test_score_whole_path_consistency(he_weights, clips):
    pcm = np.concatenate([clips["test_bigdog"], clips["test_combo"][:96000]])   # 3 s -> 2 windows
    r = oracle.score(he_weights, 4, pcm, 2, trust=0.5, empty_class=2, want_spec=True)
    assert np.array_equal(r["spec"][0], oracle.prepare_go(clips["test_bigdog"])[0])
    assert np.allclose(r["probs"].sum(axis=1), 1.0)
    assert np.array_equal(r["label"], r["logits"].argmax(axis=1))
