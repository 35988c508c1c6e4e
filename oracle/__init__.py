"""ctypes loader for the CPU oracle (oracle/apr_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from apr_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libapr_oracle.so")

NMELS, SPEC_FRAMES, NBINS, GO_FRAMES = 128, 188, 1025, 184
SAMPLE_SIZE = 192000
WARN_UNDERFLOW, WARN_OVERFLOW = 1, 2


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "apr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, f32p, f64p, i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int32))
        L.orc_normalize_audio.argtypes = [u8p, C.c_size_t, f32p]
        L.orc_mel_lens.argtypes = [f64p, i32p, i32p]
        L.orc_librosa_mel_basis.argtypes = [f32p]
        L.orc_hann.argtypes = [f64p, f64p]
        L.orc_prepare_go_stages.argtypes = [u8p, C.c_size_t, f64p, f64p, f32p]
        L.orc_prepare_go_stages.restype = C.c_int
        L.orc_prepare_go_batch.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
        L.orc_prepare_librosa.argtypes = [f32p, C.c_size_t, f32p]
        L.orc_prepare_librosa.restype = C.c_int
        L.orc_prepare_librosa_batch.argtypes = [u8p, C.c_size_t, C.c_size_t, f32p]
        L.orc_prepare_librosa_batch.restype = C.c_int
        L.orc_blob_floats.argtypes = [C.c_int]
        L.orc_blob_floats.restype = C.c_size_t
        L.orc_cnn_forward.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p]
        L.orc_cnn_forward.restype = C.c_int
        L.orc_cnn_forward_feat.argtypes = [f32p, C.c_int, f32p, f32p, f32p]
        L.orc_cnn_forward_feat.restype = C.c_int
        L.orc_softmax.argtypes = [f32p, C.c_int, f64p]
        L.orc_decide.argtypes = [f64p, C.c_int, C.c_int, C.c_double, i32p, f64p]
        L.orc_decide.restype = C.c_int
        L.orc_score.argtypes = [f32p, C.c_int, u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_int,
                                f32p, f32p, f64p, i32p, u8p]
        L.orc_score.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u8(buf) -> np.ndarray:
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def normalize_audio(pcm) -> np.ndarray:
    b = _u8(pcm)
    out = np.empty(b.size // 2, np.float32)
    lib().orc_normalize_audio(_p(b, C.c_uint8), b.size, _p(out, C.c_float))
    return out


def mel_lens():
    w = np.empty((NMELS, NBINS), np.float64)
    bounds = np.empty((NMELS, 2), np.int32)
    pts = np.empty(NMELS + 2, np.int32)
    lib().orc_mel_lens(_p(w, C.c_double), _p(bounds, C.c_int32), _p(pts, C.c_int32))
    return w, bounds, pts


def librosa_mel_basis() -> np.ndarray:
    w = np.empty((NMELS, NBINS), np.float32)
    lib().orc_librosa_mel_basis(_p(w, C.c_float))
    return w


def hann():
    s = np.empty(2048, np.float64)
    p = np.empty(2048, np.float64)
    lib().orc_hann(_p(s, C.c_double), _p(p, C.c_double))
    return s, p


def prepare_go(pcm, stages: bool = False):
    """src/model/model.go:86 Prepare -> ([128,188] f32, warn flags[, power, mel])."""
    b = _u8(pcm)
    out = np.empty((NMELS, SPEC_FRAMES), np.float32)
    if stages:
        power = np.empty((NBINS, GO_FRAMES), np.float64)
        mel = np.empty((NMELS, GO_FRAMES), np.float64)
        w = lib().orc_prepare_go_stages(_p(b, C.c_uint8), b.size, _p(power, C.c_double), _p(mel, C.c_double),
                                        _p(out, C.c_float))
        return out, w, power, mel
    w = lib().orc_prepare_go_stages(_p(b, C.c_uint8), b.size, None, None, _p(out, C.c_float))
    return out, w


def prepare_go_batch(pcm, n_windows: int, stride: int = 96000) -> np.ndarray:
    b = _u8(pcm)
    out = np.empty((n_windows, NMELS, SPEC_FRAMES), np.float32)
    lib().orc_prepare_go_batch(_p(b, C.c_uint8), b.size, n_windows, stride, _p(out, C.c_float))
    return out


def prepare_librosa(audio_f32) -> np.ndarray:
    a = np.ascontiguousarray(audio_f32, np.float32)
    nf = 1 + a.size // 512
    out = np.empty((NMELS, nf), np.float32)
    lib().orc_prepare_librosa(_p(a, C.c_float), a.size, _p(out, C.c_float))
    return out


def prepare_librosa_batch(pcm, n_clips: int, clip_bytes: int = SAMPLE_SIZE) -> np.ndarray:
    b = _u8(pcm)
    nf = 1 + (clip_bytes // 2) // 512
    out = np.empty((n_clips, NMELS, nf), np.float32)
    lib().orc_prepare_librosa_batch(_p(b, C.c_uint8), n_clips, clip_bytes, _p(out, C.c_float))
    return out


def cnn_forward(flat_weights, n_classes: int, spec) -> np.ndarray:
    w = np.ascontiguousarray(flat_weights, np.float32)
    assert w.size == lib().orc_blob_floats(n_classes)
    s = np.ascontiguousarray(spec, np.float32).reshape(-1, NMELS, SPEC_FRAMES)
    out = np.empty((s.shape[0], n_classes), np.float32)
    rc = lib().orc_cnn_forward(_p(w, C.c_float), n_classes, _p(s, C.c_float), s.shape[0], _p(out, C.c_float))
    if rc:
        raise MemoryError("orc_cnn_forward")
    return out


def cnn_forward_feat(flat_weights, n_classes: int, spec):
    w = np.ascontiguousarray(flat_weights, np.float32)
    s = np.ascontiguousarray(spec, np.float32).reshape(NMELS, SPEC_FRAMES)
    logits = np.empty(n_classes, np.float32)
    feat = np.empty(1024, np.float32)
    rc = lib().orc_cnn_forward_feat(_p(w, C.c_float), n_classes, _p(s, C.c_float), _p(logits, C.c_float),
                                    _p(feat, C.c_float))
    if rc:
        raise MemoryError("orc_cnn_forward_feat")
    return logits, feat


def softmax(logits) -> np.ndarray:
    l = np.ascontiguousarray(logits, np.float32).reshape(-1)
    out = np.empty(l.size, np.float64)
    lib().orc_softmax(_p(l, C.c_float), l.size, _p(out, C.c_double))
    return out


def decide(probs, empty_class: int, trust: float):
    p = np.ascontiguousarray(probs, np.float64).reshape(-1)
    lab = C.c_int32(0)
    conf = C.c_double(0)
    m = lib().orc_decide(_p(p, C.c_double), p.size, empty_class, trust, C.byref(lab), C.byref(conf))
    return lab.value, conf.value, bool(m)


def score(flat_weights, n_classes: int, pcm, n_windows: int, stride: int = 96000, trust: float = 0.5,
          empty_class: int = -1, want_spec: bool = False):
    """Whole path (A1..A9).  Returns dict(probs f64 [n,C], label i32 [n], match u8 [n], logits[, spec])."""
    w = np.ascontiguousarray(flat_weights, np.float32)
    b = _u8(pcm)
    probs = np.empty((n_windows, n_classes), np.float64)
    logits = np.empty((n_windows, n_classes), np.float32)
    label = np.empty(n_windows, np.int32)
    match = np.empty(n_windows, np.uint8)
    spec = np.empty((n_windows, NMELS, SPEC_FRAMES), np.float32) if want_spec else None
    rc = lib().orc_score(_p(w, C.c_float), n_classes, _p(b, C.c_uint8), b.size, n_windows, stride, trust,
                         empty_class, _p(spec, C.c_float) if want_spec else None, _p(logits, C.c_float),
                         _p(probs, C.c_double), _p(label, C.c_int32), _p(match, C.c_uint8))
    if rc:
        raise MemoryError("orc_score")
    out = dict(probs=probs, label=label, match=match, logits=logits)
    if want_spec:
        out["spec"] = spec
    return out


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(n)
