"""Developer tool: per-wave s_memtime stamps of wino_s2_fused (a -DWS_STAMPS side build, tools/build_variant.sh) or of
wino_fused8_f2 (-DW8_STAMPS; WS_PHASES=2: load steps / store steps):
    APR_AMD_LIB=tools/tmp_libs/libapr_NAME.so [WS_PHASES=2] python tools/ws_stamps.py
One 2048-window pass; the counters of workgroup 0 are summed over the layers that run the stamped kernel."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, '.')
from apr_amd import model, weights as W, _capi
C = 4
NPH = int(os.environ.get("WS_PHASES", "4"))
N = 2048
eng = model.Engine(W.pack_blob(W.random_init(C, 0, "he"), C), 0, N)
pcm = model.synth_pcm(0, 0, N + 1).view(np.uint8)
lib = _capi.lib()
lib.apr_debug_ws_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
eng.score(pcm, N)
buf = (ctypes.c_ulonglong * 64)()
assert lib.apr_debug_ws_stamps(buf, 1) == 0
eng.score(pcm, N)
assert lib.apr_debug_ws_stamps(buf, 0) == 0
a = np.array(buf[:], dtype=np.float64).reshape(8, 8)
print("wave(wn*4+q): barrier wait cycles per step of phase (approx.: every phase taken as an equal share of the steps) | K loop cycles per step | steps")
for w in range(8):
    u = max(a[w, 7], 1)
    print(w, " ".join(f"{NPH * x / u:7.0f}" for x in a[w, :NPH]), "|", f"{a[w, 6] / u:7.0f}", "|", int(a[w, 7]))
if os.environ.get("WS_KC"):   # one stamped layer (e.g. APR_WINO_S2_FUSED_MIN_PX=1000: layer 4 alone, WS_KC=16): cycles per unit
    kc = int(os.environ["WS_KC"])
    print("per unit: epilogue | top of unit | K loop")
    for w in range(8):
        units = max(a[w, 7] / kc, 1)
        print(w, f"{a[w, 4] / units:7.0f} {a[w, 5] / units:7.0f} {a[w, 6] / units:7.0f}")
