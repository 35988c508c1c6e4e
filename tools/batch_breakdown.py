"""Per-kernel times of one pass of N windows on an engine of max_batch N (developer tool):
    python tools/batch_breakdown.py 256"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from apr_amd import model, weights as W
C = 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = model.Engine(W.pack_blob(W.random_init(C, 0, "he"), C), 0, N)
pcm = model.synth_pcm(0, 0, N + 1).view(np.uint8)
for _ in range(3): eng.score(pcm, N)
eng.profile_reset(); eng.profile_enable(True)
R = 10
for _ in range(R): eng.score(pcm, N)
prof = eng.profile_read(); eng.profile_enable(False)
tot = 0
for s in prof:
    if s["launches"]:
        ms = s["ms"] / s["launches"]
        print(f'{s["layer"]:3d} {s["name"]:40s} {1e3 * ms:9.1f} us  {s["flops"] / s["launches"] / ms / 1e9 if ms else 0:7.1f} TFLOP/s')
        tot += ms
print("sum of kernels ms", tot)
lat = []
for _ in range(20):
    t1 = time.perf_counter(); eng.score(pcm, N); lat.append(1e3 * (time.perf_counter() - t1))
print("p50 ms per pass, host pointers (profiling off)", np.median(lat))
