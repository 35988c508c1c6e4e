"""Host-side mirror of the reference's Go package ``dtrack/model`` over the C ABI.

Same names, argument meaning and error behaviour as ``src/model/model.go``:
constants (model.go:23-40), ``OnnxModel`` (model.go:43-46), ``Train`` (model.go:49-51),
``Load`` (model.go:54-83), ``Prepare`` (model.go:86-184), ``Infer`` (model.go:188-229),
plus the batch entry points the sweep uses (``Engine.score`` ...).  The Go toolchain is
absent in the build image, so this Python module is what the parity tests drive; the
cgo shim a maintainer would add is shown in INTEGRATION.md and go/model/apr_cgo.go.

All arithmetic runs in libapr_amd.so on the GPU.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import sys
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _capi
from ._capi import AprError, FLAVOUR_GO, FLAVOUR_LIBROSA, WARN_OVERFLOW, WARN_UNDERFLOW

# ---- src/ffmpeg/ffmpeg.go:20-21 ----
SampleRate = 48000
BytesPerSecond = 96000
# ---- src/model/model.go:23-40 ----
SegmentSize = 2
SampleSize = BytesPerSecond * SegmentSize
Nmels = 128
Nfft = 2048
HopLength = 512
SpectrogramFrames = 188
Int16Max = 32768.0
MaxFreq = float(SampleRate) / 2
MinFreq = 0.0


class Die(SystemExit):
    """log.Die (src/log/log.go:42-46): '[ERROR]\\t msg' on stderr, exit status 1."""

    def __init__(self, msg: str):
        sys.stderr.write(f"[ERROR]\t {msg}\n")
        super().__init__(1)


def _warn(msg: str) -> None:  # log.Warn (src/log/log.go:36-39)
    sys.stderr.write(f"[WARN]\t {msg}\n")


@dataclass
class OnnxModel:
    """model.go:43-46"""
    RawBytes: bytes = b""
    Labels: List[str] = field(default_factory=list)


def Train() -> None:  # model.go:49-51
    print("[INFO]\t Model Training is handled by python -m ai.train")


def Load(model_path: str) -> OnnxModel:
    """model.go:54-83: read <name>.onnx and <name>.labels (JSON string array); Die on failure."""
    try:
        with open(model_path, "rb") as fh:
            raw = fh.read()
    except OSError as err:
        raise Die(f"could not read ONNX file: {err}")
    json_path = os.path.splitext(model_path)[0] + ".labels"
    try:
        with open(json_path, "rb") as fh:
            labels_bytes = fh.read()
    except OSError as err:
        raise Die(f"could not read Labels file: {err}")
    try:
        labels = json.loads(labels_bytes)
        if not isinstance(labels, list) or not all(isinstance(s, str) for s in labels):
            raise ValueError("not a JSON array of strings")
    except ValueError as err:
        raise Die(f"could not parse Labels JSON: {err}")
    return OnnxModel(RawBytes=raw, Labels=labels)


# ------------------------------------------------------------------------------------------
# Engine: one GPU stream + (optionally) one loaded model
# ------------------------------------------------------------------------------------------
def _addr(a) -> int:
    """address of a numpy array (host) or a torch tensor (host or device)"""
    if a is None:
        return 0
    if isinstance(a, int):          # a raw device address (Engine.device_alloc)
        return a
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return a.data_ptr()
    raise TypeError(type(a))


def _as_u8(buf) -> np.ndarray:
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(buf, dtype=np.uint8)


class Engine:
    """apr_engine handle (include/apr.h).  ``model_bytes=None`` gives a front-end-only engine."""

    def __init__(self, model_bytes: Optional[bytes] = None, device: int = 0, max_batch: int = 0):
        self._lib = _capi.lib()
        h = C.c_void_p()
        mb = bytes(model_bytes) if model_bytes else None
        errbuf = C.create_string_buffer(512)
        rc = self._lib.apr_engine_create_ex(mb, len(mb) if mb else 0, device, max_batch, C.byref(h), errbuf, len(errbuf))
        if rc != 0:
            raise AprError(rc, errbuf.value.decode(errors="replace"))
        self._h = h
        self.device = device
        self.num_classes = self._lib.apr_engine_num_classes(h)
        self._fin = weakref.finalize(self, self._lib.apr_engine_destroy, h)

    def close(self) -> None:
        self._fin()

    def _chk(self, rc: int) -> int:
        if rc < 0:
            raise AprError(rc, (self._lib.apr_last_error(self._h) or b"").decode())
        return rc

    # ---- host-pointer entry points ----
    def prepare(self, pcm, flavour: int = FLAVOUR_GO):
        b = _as_u8(pcm)
        out = np.empty((Nmels, SpectrogramFrames), np.float32)
        warn = self._chk(self._lib.apr_prepare(self._h, flavour, b.ctypes.data, b.size, out.ctypes.data))
        return out, warn

    def prepare_batch(self, pcm, n_windows: int, stride: int = BytesPerSecond, flavour: int = FLAVOUR_GO):
        b = _as_u8(pcm)
        out = np.empty((n_windows, Nmels, SpectrogramFrames), np.float32)
        self._chk(self._lib.apr_prepare_batch(self._h, flavour, b.ctypes.data, b.size, n_windows, stride, out.ctypes.data))
        return out

    def infer_logits(self, spec) -> np.ndarray:
        s = np.ascontiguousarray(spec, np.float32).reshape(-1, Nmels, SpectrogramFrames)
        out = np.empty((s.shape[0], self.num_classes), np.float32)
        self._chk(self._lib.apr_infer(self._h, s.ctypes.data, s.shape[0], out.ctypes.data))
        return out

    def classify(self, spec, trust: float = 0.5, empty_class: int = -1):
        """Infer + decision on prepared spectrograms -> dict(probs, label, match, logits)."""
        s = np.ascontiguousarray(spec, np.float32).reshape(-1, Nmels, SpectrogramFrames)
        n = s.shape[0]
        probs = np.empty((n, self.num_classes), np.float64)
        logits = np.empty((n, self.num_classes), np.float32)
        label = np.empty(n, np.int32)
        match = np.empty(n, np.uint8)
        self._chk(self._lib.apr_classify(self._h, s.ctypes.data, n, trust, empty_class, probs.ctypes.data,
                                         label.ctypes.data, match.ctypes.data, logits.ctypes.data))
        return dict(probs=probs, label=label, match=match, logits=logits)

    def classify_device(self, d_spec, batch: int, trust: float, empty_class: int, d_probs=None, d_label=None,
                        d_match=None, d_logits=None) -> None:
        self._chk(self._lib.apr_classify_device(self._h, _addr(d_spec), batch, trust, empty_class, _addr(d_probs),
                                                _addr(d_label), _addr(d_match), _addr(d_logits)))

    def score(self, pcm, n_windows: Optional[int] = None, stride: int = BytesPerSecond, trust: float = 0.5,
              empty_class: int = -1):
        """Fused sweep over one PCM buffer -> dict(probs [n,C] f64, label [n] i32, match [n] u8)."""
        b = _as_u8(pcm)
        if n_windows is None:
            n_windows = windows_in(b.size, stride)
        probs = np.empty((n_windows, self.num_classes), np.float64)
        label = np.empty(n_windows, np.int32)
        match = np.empty(n_windows, np.uint8)
        self._chk(self._lib.apr_score(self._h, b.ctypes.data, b.size, n_windows, stride, trust, empty_class,
                                      probs.ctypes.data, label.ctypes.data, match.ctypes.data))
        return dict(probs=probs, label=label, match=match)

    def stream_buffer(self, n_streams: int) -> np.ndarray:
        """apr_stream_buffer: the engine's pinned staging for one tick as a [n_streams, 96000] uint8 array."""
        p = self._lib.apr_stream_buffer(self._h, n_streams)
        if not p:
            raise AprError(_capi.ERR_INVALID, (self._lib.apr_last_error(self._h) or b"").decode())
        raw = (C.c_uint8 * (n_streams * BytesPerSecond)).from_address(p)
        return np.frombuffer(raw, np.uint8).reshape(n_streams, BytesPerSecond)

    def stream_reset(self) -> None:
        self._chk(self._lib.apr_stream_reset(self._h))

    def stream_tick(self, segments: np.ndarray, n_streams: int, trust: float = 0.5, empty_class: int = -1):
        """apr_stream_tick: None while priming, else dict(probs, label, match) for the n_streams windows."""
        seg = _as_u8(segments)
        if seg.size != n_streams * BytesPerSecond:
            raise ValueError("segments must hold n_streams x 96000 bytes")
        probs = np.empty((n_streams, self.num_classes), np.float64)
        label = np.empty(n_streams, np.int32)
        match = np.empty(n_streams, np.uint8)
        rc = self._chk(self._lib.apr_stream_tick(self._h, seg.ctypes.data, n_streams, trust, empty_class, probs.ctypes.data,
                                                 label.ctypes.data, match.ctypes.data))
        return None if rc == 1 else dict(probs=probs, label=label, match=match)

    def sweep_files(self, files, stride: int = BytesPerSecond, trust: float = 0.5, empty_class: int = -1):
        """apr_sweep_files over a list of host PCM buffers -> dict(probs, label, match, first_window [n_files+1])."""
        bufs = [_as_u8(f) for f in files]
        n = len(bufs)
        ptrs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(n, 1))(*[b.size for b in bufs])
        first = (C.c_size_t * (n + 1))()
        total = sum(windows_in(b.size, stride) for b in bufs)
        probs = np.empty((total, self.num_classes), np.float64)
        label = np.empty(total, np.int32)
        match = np.empty(total, np.uint8)
        self._chk(self._lib.apr_sweep_files(self._h, n, ptrs, lens, stride, trust, empty_class, probs.ctypes.data,
                                            label.ctypes.data, match.ctypes.data, first))
        return dict(probs=probs, label=label, match=match, first_window=np.asarray(list(first), np.int64))

    @staticmethod
    def sweep_files_multi(engines, files, stride: int = BytesPerSecond, trust: float = 0.5, empty_class: int = -1):
        """apr_sweep_files_multi: one engine per GPU of the node (the device list), recordings sharded by file
        (file f -> engines[f % n]), rows written in file order -> dict as sweep_files."""
        if not engines:
            raise ValueError("sweep_files_multi needs at least one engine")
        lib = engines[0]._lib
        bufs = [_as_u8(f) for f in files]
        n = len(bufs)
        hs = (C.c_void_p * len(engines))(*[e._h.value for e in engines])
        ptrs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(n, 1))(*[b.size for b in bufs])
        first = (C.c_size_t * (n + 1))()
        total = sum(windows_in(b.size, stride) for b in bufs)
        nc = engines[0].num_classes
        probs = np.empty((total, nc), np.float64)
        label = np.empty(total, np.int32)
        match = np.empty(total, np.uint8)
        rc = lib.apr_sweep_files_multi(hs, len(engines), n, ptrs, lens, stride, trust, empty_class, probs.ctypes.data,
                                       label.ctypes.data, match.ctypes.data, first)
        if rc < 0:
            raise AprError(rc, (lib.apr_last_error(engines[0]._h) or b"").decode())
        return dict(probs=probs, label=label, match=match, first_window=np.asarray(list(first), np.int64))

    def upsample(self, pcm, up: int = 3, taps=None) -> np.ndarray:
        """apr_upsample: s16 PCM at rate r (host array) -> s16 PCM at rate ``up`` x r, on the GPU.  16 kHz sources
        become the 48 kHz this path scores; the reference delegates this to ffmpeg (src/ffmpeg/ffmpeg.go:76)."""
        from .sweep import resample_taps
        x = np.ascontiguousarray(pcm, np.int16).reshape(-1)
        t = np.ascontiguousarray(resample_taps(up) if taps is None else taps, np.float64)
        if t.ndim != 2 or t.shape[0] != up:
            raise ValueError("taps must be [up, taps_per_phase]")
        out = np.empty(x.size * up, np.int16)
        self._chk(self._lib.apr_upsample(self._h, x.ctypes.data, x.size, up, t.ctypes.data, t.shape[1], out.ctypes.data))
        return out

    def upsample_device(self, d_in, n_in: int, d_out, up: int = 3, taps=None) -> None:
        from .sweep import resample_taps
        t = np.ascontiguousarray(resample_taps(up) if taps is None else taps, np.float64)
        self._chk(self._lib.apr_upsample_device(self._h, _addr(d_in), n_in, up, t.ctypes.data, t.shape[1], _addr(d_out)))

    # ---- device-resident entry points (torch tensors on this engine's GPU) ----
    def score_device(self, d_pcm, total_bytes: int, n_windows: int, stride: int, trust: float, empty_class: int,
                     d_probs=None, d_label=None, d_match=None, d_logits=None) -> None:
        self._chk(self._lib.apr_score_device(self._h, _addr(d_pcm), total_bytes, n_windows, stride, trust, empty_class,
                                             _addr(d_probs), _addr(d_label), _addr(d_match), _addr(d_logits)))

    def prepare_batch_device(self, d_pcm, total_bytes: int, n_windows: int, stride: int, d_out,
                             flavour: int = FLAVOUR_GO) -> None:
        self._chk(self._lib.apr_prepare_batch_device(self._h, flavour, _addr(d_pcm), total_bytes, n_windows, stride,
                                                     _addr(d_out)))

    def prepare_batch_device_static(self, d_pcm, total_bytes: int, n_windows: int, stride: int, d_out,
                                    flavour: int = FLAVOUR_GO) -> None:
        """apr_prepare_batch_device_static: the allocation-free form (<= 64 windows; scratch is a static device array)."""
        self._chk(self._lib.apr_prepare_batch_device_static(self._h, flavour, _addr(d_pcm), total_bytes, n_windows, stride,
                                                            _addr(d_out)))

    def bucket_hits_device(self, d_label, d_match, n_windows: int, n_classes: int, d_hits, windows_per_bucket: int = 60) -> None:
        """apr_bucket_hits_device: one hit per class per minute from device-resident (label, match) rows."""
        self._chk(self._lib.apr_bucket_hits_device(self._h, _addr(d_label), _addr(d_match), n_windows, n_classes,
                                                   windows_per_bucket, _addr(d_hits)))

    # ---- device memory for hosts without an allocator of their own (what the Go shim uses; torch tensors work as well) ----
    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self._lib.apr_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value or 0

    def device_free(self, d_ptr: int) -> None:
        self._chk(self._lib.apr_device_free(self._h, d_ptr))

    def device_upload(self, d_dst: int, host) -> None:
        b = _as_u8(host)
        self._chk(self._lib.apr_device_upload(self._h, d_dst, b.ctypes.data, b.size))

    def device_download(self, d_src: int, out: np.ndarray) -> np.ndarray:
        self._chk(self._lib.apr_device_download(self._h, out.ctypes.data, d_src, out.nbytes))
        return out

    @staticmethod
    def gather_rows_device(engines, n_rows, n_classes: int, d_probs, d_label, d_match, d_probs_out, d_label_out, d_match_out,
                           root: int = 0) -> None:
        """apr_gather_rows_device: the final gather of a one-process multi-GPU sweep over RCCL (device pointers: ints or torch
        tensors; a tensor that is not wanted is None in the per-engine list and in the output)."""
        if not engines:
            raise ValueError("gather_rows_device needs at least one engine")
        n = len(engines)
        lib = engines[0]._lib
        hs = (C.c_void_p * n)(*[e._h.value for e in engines])
        rows = (C.c_size_t * n)(*[int(r) for r in n_rows])

        def ptrs(lst):
            if lst is None:
                return None
            return (C.c_void_p * n)(*[_addr(t) for t in lst])
        rc = lib.apr_gather_rows_device(hs, n, root, rows, n_classes, ptrs(d_probs), ptrs(d_label), ptrs(d_match),
                                        _addr(d_probs_out), _addr(d_label_out), _addr(d_match_out))
        if rc < 0:
            # argument errors are kept per thread (apr_last_error(NULL)), later ones on the root engine
            msg = (lib.apr_last_error(engines[root]._h) if 0 <= root < n else b"") or lib.apr_last_error(None) or b""
            raise AprError(rc, msg.decode())

    def infer_device(self, d_spec, batch: int, d_logits) -> None:
        self._chk(self._lib.apr_infer_device(self._h, _addr(d_spec), batch, _addr(d_logits)))

    def sync(self) -> None:
        self._chk(self._lib.apr_sync(self._h))

    # ---- profiling ----
    def mfma_probe(self) -> float:
        """fp32 matrix TFLOP/s this GPU delivers right now (apr_mfma_probe; a diagnostic)"""
        v = C.c_double(0.0)
        self._chk(self._lib.apr_mfma_probe(self._h, C.byref(v)))
        return float(v.value)

    def profile_enable(self, on: bool = True) -> None:
        self._chk(self._lib.apr_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self) -> None:
        self._chk(self._lib.apr_profile_reset(self._h))

    def profile_read(self):
        slots = (_capi.ProfileSlot * _capi.PROF_SLOTS)()
        n = C.c_int(0)
        self._chk(self._lib.apr_profile_read(self._h, slots, C.byref(n)))
        return [dict(name=s.name.decode(), layer=s.layer, ms=s.ms, launches=s.launches, flops=s.flops, bytes=s.bytes)
                for s in slots[: n.value]]


def windows_in(n_bytes: int, stride: int = BytesPerSecond) -> int:
    """2-s windows with 1-s hop in a recording: ai/inspect.py:108-109 ``range(0, len-bps, bps)``
    (= T-1 for T whole seconds; daemon.go:115-125 delays by one segment the same way)."""
    if n_bytes <= stride:
        return 0
    return len(range(0, n_bytes - stride, stride))


# ------------------------------------------------------------------------------------------
# The reference's package-level functions
# ------------------------------------------------------------------------------------------
_frontend: Dict[int, Engine] = {}
_engines: Dict[tuple, Engine] = {}
_hash_memo: Dict[int, tuple] = {}
_default_device = 0


def set_device(device: int) -> None:
    global _default_device
    _default_device = device


def _frontend_engine() -> Engine:
    e = _frontend.get(_default_device)
    if e is None:
        e = _frontend[_default_device] = Engine(None, _default_device, 64)
    return e


def Prepare(pcmData: bytes) -> np.ndarray:
    """model.go:86-184.  Returns the float32 tensor of shape [1,1,128,188]; warns like the
    reference on short / long input.  (The reference's error return is always nil.)"""
    out, warn = _frontend_engine().prepare(pcmData, FLAVOUR_GO)
    if warn & WARN_UNDERFLOW:
        _warn("Audio Underflow; Segment was not large enough!")
    if warn & WARN_OVERFLOW:
        _warn("Audio Overflow; Segment was too large!")
    return out.reshape(1, 1, Nmels, SpectrogramFrames)


def _model_key(raw: bytes, max_batch: int, device: int) -> tuple:
    # by CONTENT, not by address: a model re-loaded after retraining has the same architecture and length and may
    # land at the address the old bytes had
    import hashlib
    return (hashlib.sha256(raw).digest(), len(raw), max_batch, device)


def engine_for(inferModel: OnnxModel, max_batch: int = 0) -> Engine:
    """Device-resident engine cached by the content of RawBytes (sha256), max_batch and device: removes the
    per-call backend build + UnmarshalBinary of model.go:190-196 without changing the API.  The hash of the
    94 MB is computed once per bytes object: the memo holds the object itself, so its address cannot be reused
    while the entry exists (Unload drops it)."""
    raw = inferModel.RawBytes
    memo = _hash_memo.get(id(raw))
    if memo is None or memo[0] is not raw:
        _hash_memo[id(raw)] = memo = (raw, _model_key(raw, 0, 0)[0])
    key = (memo[1], len(raw), max_batch, _default_device)
    e = _engines.get(key)
    if e is None:
        try:
            e = Engine(raw, _default_device, max_batch)
        except AprError as err:
            raise Die(f"could not unmarshal ONNX model: {err}")
        _engines[key] = e
    return e


def Unload(inferModel: Optional[OnnxModel] = None) -> int:
    """Destroy the cached engines of one model (or of every model, and the front-end engines, when None): GPU memory
    goes back at once.  No counterpart in the reference, whose Infer keeps nothing between calls.  Returns the count."""
    n = 0
    if inferModel is None:
        for e in list(_engines.values()) + list(_frontend.values()):
            e.close()
            n += 1
        _engines.clear()
        _frontend.clear()
        _hash_memo.clear()
        return n
    digest = _model_key(inferModel.RawBytes, 0, 0)[0]
    for key in [k for k in _engines if k[0] == digest]:
        _engines.pop(key).close()
        n += 1
    _hash_memo.pop(id(inferModel.RawBytes), None)
    return n


def softmax(logits) -> List[float]:
    """model.go:232-254, float64."""
    mx = -sys.float_info.max
    for v in logits:
        if v > mx:
            mx = v
    exps = [math.exp(v - mx) for v in logits]
    s = 0.0
    for v in exps:
        s += v
    return [v / s for v in exps]


def Infer(inferModel: OnnxModel, preparedAudio: np.ndarray) -> Dict[str, float]:
    """model.go:188-229: logits on the GPU, f64 softmax and label map on the host."""
    e = engine_for(inferModel, 1)     # one clip per call, like the Go shim's Infer engine (split-K layers, graph replay)
    try:
        logits = e.infer_logits(np.asarray(preparedAudio, np.float32).reshape(1, Nmels, SpectrogramFrames))[0]
    except AprError as err:
        raise Die(f"Inference failed: {err}")
    probs = softmax([float(v) for v in logits])           # model.go:212-218
    results: Dict[str, float] = {}
    for i, label in enumerate(inferModel.Labels):          # model.go:221-226
        if i < len(probs):
            results[label] = probs[i]
    return results


def decide(predictions: Dict[str, float], trust: float):
    """src/daemon/daemon.go:189-207 -> (bestClass, bestConf, match)."""
    best, conf = "", 0.0
    for label, score in predictions.items():
        if score > conf:
            conf, best = score, label
    return best, conf, (best != "empty" and conf > trust)


def audio_to_spectrogram(pcm: bytes) -> np.ndarray:
    """ai/model.py:184-204 on normalize_audio(pcm) for one .dat clip -> [1,128,188] f32."""
    out, _ = _frontend_engine().prepare(pcm, FLAVOUR_LIBROSA)
    return out.reshape(1, Nmels, SpectrogramFrames)


def synth_pcm(base_seed: int, file_index: int, n_seconds: int, second_offset: int = 0) -> np.ndarray:
    """Deterministic synthetic 48 kHz s16 recording (apr_synth_pcm, host side)."""
    out = np.empty(n_seconds * SampleRate, np.int16)
    rc = _capi.lib().apr_synth_pcm(base_seed, file_index, second_offset, n_seconds, out.ctypes.data)
    if rc:
        raise AprError(rc, "apr_synth_pcm")
    return out


def sweep_plan(file_bytes, n_devices: int, stride: int = BytesPerSecond):
    """apr_sweep_plan (host only): (file_device [n], first_window [n+1], device_windows [n_devices])."""
    n = len(file_bytes)
    lens = (C.c_size_t * max(n, 1))(*[int(b) for b in file_bytes])
    dev = np.zeros(max(n, 1), np.int32)
    first = (C.c_size_t * (n + 1))()
    per = (C.c_size_t * max(n_devices, 1))()
    rc = _capi.lib().apr_sweep_plan(n, lens, stride, n_devices, dev.ctypes.data, first, per)
    if rc:
        raise AprError(rc, "apr_sweep_plan: bad argument")
    return dev[:n], np.asarray(list(first), np.int64), np.asarray(list(per), np.int64)


def model_info(model_bytes: bytes):
    """apr_model_info: (n_classes, n_conv_layers) or AprError with the parser's message.  No GPU."""
    L = _capi.lib()
    nc, nl = C.c_int(0), C.c_int(0)
    rc = L.apr_model_info(bytes(model_bytes), len(model_bytes), C.byref(nc), C.byref(nl))
    if rc:
        raise AprError(rc, (L.apr_last_error(None) or b"").decode())
    return nc.value, nl.value
