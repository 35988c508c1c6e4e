#!/usr/bin/env python3
"""Side measurements for BASELINE configs 4 and 5 on one GPU (bench.py stays the headline).

  config 4  live mode: 256 concurrent streams, one 1-s segment each per tick, window = previous
            segment || new segment (src/daemon/daemon.go:104-125); p50/p99 from segment arrival
            (host memory) to labels (host memory), and the sustained windows/s.
  host sweep  PCIe-inclusive rate of apr_sweep_files: recordings in pageable host memory -> pinned
            double-buffered staging -> GPU -> labels in host memory (never the headline number).
  config 5  training-set preprocessing: 10 000 tagged .dat clips (192 000 B each) -> librosa-flavour
            log-mel [10000,1,128,188] (ai/train.py:36-52), clips already in HBM.
Prints one JSON line per config.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--clips", type=int, default=10000)
    ap.add_argument("--sweep-files", type=int, default=30)
    ap.add_argument("--sweep-seconds", type=int, default=600)
    args = ap.parse_args()
    import torch
    from apr_amd import model, sweep, weights as W
    from apr_amd._capi import FLAVOUR_LIBROSA
    dev = torch.device("cuda", 0)

    # ---------------- config 4 ----------------
    C = 4
    eng = model.Engine(W.pack_blob(W.random_init(C, 0, "he"), C), 0, args.streams)
    secs = 8
    src = np.stack([model.synth_pcm(4, s, secs).view(np.uint8).reshape(secs, 96000) for s in range(args.streams)])
    sc = sweep.StreamScorer(eng, args.streams, trust=0.5, empty_class=2)
    sc.tick(np.ascontiguousarray(src[:, 0]))
    lat, lat0 = [], []
    for t in range(args.ticks):
        seg = np.ascontiguousarray(src[:, 1 + t % (secs - 1)])       # the tick's segments, pageable host memory
        t0 = time.perf_counter()
        sc.tick(seg)
        lat.append(1e3 * (time.perf_counter() - t0))
    for t in range(args.ticks):                                       # the reader deposits into the engine's pinned staging
        sc.buffer[:] = src[:, 1 + (t + args.ticks) % (secs - 1)]
        t0 = time.perf_counter()
        sc.tick()
        lat0.append(1e3 * (time.perf_counter() - t0))
    lat, lat0 = np.array(lat[3:]), np.array(lat0[3:])
    print(json.dumps({"config": 4, "streams": args.streams, "ticks": int(lat.size),
                      "latency_ms_p50": float(np.percentile(lat, 50)), "latency_ms_p99": float(np.percentile(lat, 99)),
                      "latency_ms_max": float(lat.max()), "real_time_budget_ms": 1000.0,
                      "sustained_windows_per_s": float(args.streams / (np.mean(lat) * 1e-3)),
                      "deposited_latency_ms_p50": float(np.percentile(lat0, 50)), "deposited_latency_ms_p99": float(np.percentile(lat0, 99)),
                      "note": "apr_stream_tick: the previous second of every stream stays in HBM; per tick memcpy of 256 x 96000 B into "
                              "pinned staging (skipped in the deposited_* figures: the reader wrote there), H2D, fused score, "
                              "D2H of probs/labels/matches"}),
          flush=True)
    eng.close()

    # ---------------- host-pointer sweep (PCIe inclusive) ----------------
    eng = model.Engine(W.pack_blob(W.random_init(C, 0, "he"), C), 0, 2048)
    files = [model.synth_pcm(0, f, args.sweep_seconds) for f in range(args.sweep_files)]
    eng.sweep_files(files[:2], trust=0.5, empty_class=2)           # warm-up: allocates the staging buffers
    t0 = time.perf_counter()
    res = eng.sweep_files(files, trust=0.5, empty_class=2)
    dt = time.perf_counter() - t0
    nwin = int(res["label"].size)
    print(json.dumps({"config": "host sweep", "files": args.sweep_files, "seconds_per_file": args.sweep_seconds, "windows": nwin,
                      "s": dt, "audio_hours_per_s": nwin / 3600.0 / dt, "pcm_GBps": sum(f.nbytes for f in files) / dt / 1e9,
                      "matches": int(res["match"].sum()),
                      "note": "apr_sweep_files: pageable host PCM -> pinned staging -> H2D on a copy stream overlapped with compute"}),
          flush=True)
    eng.close()

    # ---------------- config 5 ----------------
    fe = model.Engine(None, 0, 1024)
    uniq = 500
    clips = np.concatenate([model.synth_pcm(5, i, 2) for i in range(uniq)]).view(np.uint8)
    reps = (args.clips + uniq - 1) // uniq
    d_pcm = torch.from_numpy(np.tile(clips, reps)[: args.clips * 192000]).to(dev)
    d_out = torch.empty((args.clips, 128, 188), dtype=torch.float32, device=dev)
    fe.prepare_batch_device(d_pcm, d_pcm.numel(), args.clips, 192000, d_out, FLAVOUR_LIBROSA)
    fe.sync()
    t0 = time.perf_counter()
    n_rep = 5
    for _ in range(n_rep):
        fe.prepare_batch_device(d_pcm, d_pcm.numel(), args.clips, 192000, d_out, FLAVOUR_LIBROSA)
    fe.sync()
    dt = (time.perf_counter() - t0) / n_rep
    in_b, out_b = args.clips * 192000, args.clips * 128 * 188 * 4
    print(json.dumps({"config": 5, "clips": args.clips, "ms": 1e3 * dt, "clips_per_s": args.clips / dt,
                      "algorithmic_GBps": (in_b + out_b) / dt / 1e9, "flavour": "librosa",
                      "checksum": float(d_out[:uniq].double().sum().item())}), flush=True)


if __name__ == "__main__":
    main()
