"""Developer tool: s_memtime stamps of wino4_fused_out (a -DW4_STAMPS side build, tools/build_variant.sh w4st -DW4_STAMPS), one layer at a time:
    APR_DEV_KNOBS=1 APR_AMD_LIB=tools/tmp_libs/libapr_w4st.so python tools/w4_stamps.py
One 4096-window pass per layer (the other stride-1 layers run their older forms meanwhile); counters of workgroup 0, per wave, in cycles of the
100-MHz-independent shader clock counter (s_memtime)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, '.')
C, N = 4, 4096
for cin in (64, 128, 256, 512):
    os.environ["APR_WINO_F4F_MIN_CIN"] = os.environ["APR_WINO_F4F_MAX_CIN"] = str(cin)
    os.environ["APR_WINO_F4_MIN_CIN"] = "100000"
    from apr_amd import model, weights as W, _capi
    eng = model.Engine(W.pack_blob(W.random_init(C, 0, "he"), C), 0, N)
    pcm = model.synth_pcm(0, 0, N + 1).view(np.uint8)
    lib = _capi.lib()
    lib.apr_debug_w4_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    eng.score(pcm, N)
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.apr_debug_w4_stamps(buf, 1) == 0
    eng.score(pcm, N)
    assert lib.apr_debug_w4_stamps(buf, 0) == 0
    eng.close()
    a = np.array(buf[:], dtype=np.float64).reshape(8, 8)
    print(f"Cin = {cin} (layers with that input width; Cin = 512: layers 9 and 11 summed): per wave -- cycles per K step: MFMA part | closing wait + barrier || per pass: up to the first barrier | exchange, to the closing barrier | transform || per unit: between K loops | steps, passes, units")
    for w in range(8):
        st, ps, un = max(a[w, 4], 1), max(a[w, 5], 1), max(a[w, 5], 1) / 6
        print(f"  wave {w}: {a[w, 0] / st:7.0f} | {a[w, 1] / st:6.0f} || {a[w, 6] / ps:7.0f} | {a[w, 7] / ps:7.0f} | {a[w, 2] / ps:7.0f} || {a[w, 3] / un:7.0f} | {int(st)}, {int(ps)}, {int(un)}")
