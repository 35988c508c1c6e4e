import sys, time, numpy as np
sys.path.insert(0, '.')
from apr_amd import model, weights as W
C = 4
eng = model.Engine(W.pack_blob(W.random_init(C, 0, "he"), C), 0, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
pcm = model.synth_pcm(0, 0, 2).view(np.uint8)
for _ in range(5): eng.score(pcm, 1)
eng.profile_reset(); eng.profile_enable(True)
N = 20
t0 = time.perf_counter()
for _ in range(N): eng.score(pcm, 1)
dt = (time.perf_counter() - t0) / N
prof = eng.profile_read(); eng.profile_enable(False)
print("wall per clip ms (profiling on)", dt * 1e3)
tot = 0
for s in prof:
    if s["launches"]:
        print(f'{s["layer"]:3d} {s["name"]:40s} {1e3 * s["ms"] / s["launches"]:8.1f} us')
        tot += s["ms"] / s["launches"]
print("sum of kernels ms", tot)
lat = []
for _ in range(30):
    t1 = time.perf_counter(); eng.score(pcm, 1); lat.append(1e3 * (time.perf_counter() - t1))
print("p50 ms (profiling off)", np.median(lat))
