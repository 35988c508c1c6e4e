"""ctypes binding of include/apr.h (libapr_amd.so).  Fails loudly if the library is missing:
there is NO CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APR_AMD_LIB") or os.path.join(_HERE, "libapr_amd.so")   # APR_AMD_LIB: developer builds

APR_OK = 0
ERR_INVALID, ERR_DEVICE, ERR_MODEL, ERR_NOMEM, ERR_NOMODEL = -1, -2, -3, -4, -5
FLAVOUR_GO, FLAVOUR_LIBROSA = 0, 1
WARN_UNDERFLOW, WARN_OVERFLOW = 1, 2
PROF_SLOTS = 48
STATIC_MAX_WINDOWS = 64

# every symbol include/apr.h declares (tests check the library exports all of them)
SYMBOLS = (
    "apr_device_count", "apr_version", "apr_engine_create", "apr_engine_destroy", "apr_engine_num_classes",
    "apr_last_error", "apr_prepare", "apr_prepare_batch", "apr_infer", "apr_score", "apr_score_device",
    "apr_prepare_batch_device", "apr_infer_device", "apr_sync", "apr_engine_stream", "apr_profile_enable", "apr_mfma_probe",
    "apr_profile_reset", "apr_profile_read", "apr_synth_pcm", "apr_model_info", "apr_classify", "apr_classify_device",
    "apr_windows_in", "apr_sweep_files", "apr_wav_data_offset", "apr_bucket_hits",
    "apr_resample_taps", "apr_upsample", "apr_upsample_device", "apr_stream_tick", "apr_stream_buffer", "apr_stream_reset",
    "apr_prepare_batch_device_static", "apr_engine_create_ex", "apr_sweep_plan", "apr_sweep_files_multi",
    "apr_bucket_hits_device", "apr_device_alloc", "apr_device_free", "apr_device_upload", "apr_device_download",
    "apr_gather_rows_device",
)


class ProfileSlot(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("layer", C.c_int32), ("reserved", C.c_int32), ("ms", C.c_double),
                ("launches", C.c_uint64), ("flops", C.c_double), ("bytes", C.c_double)]


class AprError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"apr error {code}: {msg}")
        self.code = code


_lib = None


def hip_runtime_mapped():
    """Paths of the libamdhip64 images mapped into this process (diagnostic: which HIP runtime serves libapr_amd.so)."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m apr_amd.build` (hipcc, gfx950). "
            "apr_amd has no CPU fallback.")
    # PyTorch ships a HIP runtime of its own, and the runtime that is mapped first serves the whole process: with
    # libapr_amd.so (linked against /opt/rocm) mapped first, a later `import torch` finds no GPU.  Callers that hand
    # torch device tensors to the device-pointer entry points need one shared runtime, so map torch's first when
    # torch is installed.  Nothing of torch is used here, and a torch that is absent OR broken (its import can raise
    # OSError / RuntimeError, not only ImportError) must not make the library unloadable.  APR_PRELOAD_TORCH=0 skips
    # the preload (host-only tools: apr_model_info, apr_sweep_plan ...).  Version coupling: see INTEGRATION.md.
    import sys
    if "torch" not in sys.modules and os.environ.get("APR_PRELOAD_TORCH", "1") != "0":
        try:
            import torch  # noqa: F401
        except Exception:   # noqa: BLE001
            pass
    L = C.CDLL(LIB_PATH)
    vp, u8p, f32p, f64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # raw addresses (host or device)
    L.apr_device_count.restype = C.c_int
    L.apr_version.restype = C.c_char_p
    L.apr_engine_create.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.apr_engine_create.restype = C.c_int
    L.apr_model_info.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.apr_model_info.restype = C.c_int
    L.apr_engine_destroy.argtypes = [vp]
    L.apr_engine_destroy.restype = None
    L.apr_engine_num_classes.argtypes = [vp]
    L.apr_engine_num_classes.restype = C.c_int
    L.apr_last_error.argtypes = [vp]
    L.apr_last_error.restype = C.c_char_p
    L.apr_prepare.argtypes = [vp, C.c_int, u8p, C.c_size_t, f32p]
    L.apr_prepare.restype = C.c_int
    L.apr_prepare_batch.argtypes = [vp, C.c_int, u8p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
    L.apr_prepare_batch.restype = C.c_int
    L.apr_infer.argtypes = [vp, f32p, C.c_int, f32p]
    L.apr_infer.restype = C.c_int
    L.apr_score.argtypes = [vp, u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_int, f64p, vp, vp]
    L.apr_score.restype = C.c_int
    L.apr_score_device.argtypes = [vp, u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_int, f64p, vp, vp, f32p]
    L.apr_score_device.restype = C.c_int
    L.apr_prepare_batch_device.argtypes = [vp, C.c_int, u8p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
    L.apr_prepare_batch_device.restype = C.c_int
    L.apr_prepare_batch_device_static.argtypes = [vp, C.c_int, u8p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
    L.apr_prepare_batch_device_static.restype = C.c_int
    L.apr_infer_device.argtypes = [vp, f32p, C.c_int, f32p]
    L.apr_infer_device.restype = C.c_int
    L.apr_classify.argtypes = [vp, f32p, C.c_int, C.c_double, C.c_int, f64p, vp, vp, f32p]
    L.apr_classify.restype = C.c_int
    L.apr_classify_device.argtypes = [vp, f32p, C.c_int, C.c_double, C.c_int, f64p, vp, vp, f32p]
    L.apr_classify_device.restype = C.c_int
    L.apr_windows_in.argtypes = [C.c_size_t, C.c_size_t]
    L.apr_windows_in.restype = C.c_size_t
    L.apr_sweep_files.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_double, C.c_int,
                                  f64p, vp, vp, C.POINTER(C.c_size_t)]
    L.apr_sweep_files.restype = C.c_int
    L.apr_engine_create_ex.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.apr_engine_create_ex.restype = C.c_int
    L.apr_sweep_plan.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.c_size_t, C.c_int, vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.apr_sweep_plan.restype = C.c_int
    L.apr_sweep_files_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                        C.c_double, C.c_int, f64p, vp, vp, C.POINTER(C.c_size_t)]
    L.apr_sweep_files_multi.restype = C.c_int
    L.apr_wav_data_offset.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
    L.apr_wav_data_offset.restype = C.c_size_t
    L.apr_bucket_hits.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_size_t, vp]
    L.apr_bucket_hits.restype = C.c_int
    L.apr_bucket_hits_device.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_size_t, vp]
    L.apr_bucket_hits_device.restype = C.c_int
    L.apr_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(C.c_void_p)]
    L.apr_device_alloc.restype = C.c_int
    L.apr_device_free.argtypes = [vp, vp]
    L.apr_device_free.restype = C.c_int
    L.apr_device_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.apr_device_upload.restype = C.c_int
    L.apr_device_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.apr_device_download.restype = C.c_int
    L.apr_gather_rows_device.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), vp, vp, vp]
    L.apr_gather_rows_device.restype = C.c_int
    L.apr_resample_taps.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, vp]
    L.apr_resample_taps.restype = C.c_int
    L.apr_upsample.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_int, vp]
    L.apr_upsample.restype = C.c_int
    L.apr_upsample_device.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_int, vp]
    L.apr_upsample_device.restype = C.c_int
    L.apr_stream_tick.argtypes = [vp, u8p, C.c_int, C.c_double, C.c_int, f64p, vp, vp]
    L.apr_stream_tick.restype = C.c_int
    L.apr_stream_buffer.argtypes = [vp, C.c_int]
    L.apr_stream_buffer.restype = C.c_void_p
    L.apr_stream_reset.argtypes = [vp]
    L.apr_stream_reset.restype = C.c_int
    L.apr_sync.argtypes = [vp]
    L.apr_sync.restype = C.c_int
    L.apr_engine_stream.argtypes = [vp]
    L.apr_engine_stream.restype = C.c_void_p
    L.apr_mfma_probe.argtypes = [vp, C.POINTER(C.c_double)]
    L.apr_mfma_probe.restype = C.c_int
    L.apr_profile_enable.argtypes = [vp, C.c_int]
    L.apr_profile_enable.restype = C.c_int
    L.apr_profile_reset.argtypes = [vp]
    L.apr_profile_reset.restype = C.c_int
    L.apr_profile_read.argtypes = [vp, C.POINTER(ProfileSlot), C.POINTER(C.c_int)]
    L.apr_profile_read.restype = C.c_int
    L.apr_synth_pcm.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
    L.apr_synth_pcm.restype = C.c_int
    _lib = L
    return L
