#!/usr/bin/env python3
"""bench.py -- audio-hours scored per second on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (PCM -> STFT -> log-mel -> CNN -> softmax -> label) over
one rank's batch of synthetic recordings, PCM already resident in HBM.  Workload per GPU
(weak scaling): BASELINE config 2, 10 h of audio = 60 recordings x 600 s (the reference's
default record_duration, src/state/config.go:83), 48 kHz mono s16le holding 16 kHz-band
content; fp32 end to end; random-init NoiseDetector(4).  Recordings shard by
file; with N > 1 every step ends with the RCCL all_gather of the per-window score rows.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_F32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (vector), packed FMA
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E
LABELS = ["big_dog", "combination", "empty", "small_dog"]   # src/model/test_model.labels:1


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--files", type=int, default=60, help="recordings per GPU")
    ap.add_argument("--seconds", type=int, default=600, help="seconds per recording")
    ap.add_argument("--max-batch", type=int, default=2048, help="windows per GPU pass")
    ap.add_argument("--cpu-windows", type=int, default=599, help="windows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed sweep: no classifier centring, latency probe or CPU baseline "
                         "(use under rocprofv3 so that every launch it sees has the timed region's shapes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from apr_amd import model, sweep, weights as W

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    C = len(LABELS)
    empty = LABELS.index("empty")
    trust = 0.5   # inspect_trust default, src/state/config.go:86
    flat = W.random_init(C, 0, "he")

    # ---- synthetic recordings for this rank (files rank, rank+world, ... of n_files*world) ----
    t0 = time.time()
    my_files = sweep.shard_files(args.files * world, world, rank)
    pcm_host = np.empty(len(my_files) * args.seconds * 48000, np.int16)
    for i, f in enumerate(my_files):
        pcm_host[i * args.seconds * 48000:(i + 1) * args.seconds * 48000] = model.synth_pcm(0, f, args.seconds)
    plan = sweep.plan_stream([args.seconds * 96000] * len(my_files), my_files)
    d_pcm = torch.from_numpy(pcm_host.view(np.uint8)).to(dev)
    t_gen = time.time() - t0

    # centre the random-init classifier on this data so that labels vary between windows
    if not args.no_extras:
        eng0 = model.Engine(W.pack_blob(flat, C), local_rank, 64)
        # (recording 0 on every rank, so that all ranks score with identical weights)
        cal = eng0.infer_logits(eng0.prepare_batch(model.synth_pcm(0, 0, 65), 64))
        eng0.close()
        flat = W.center_logits(flat, C, cal.mean(axis=0))
    eng = model.Engine(W.pack_blob(flat, C), local_rank, args.max_batch)

    n = plan.n_windows
    d_probs = torch.empty((n, C), dtype=torch.float64, device=dev)
    d_label = torch.empty(n, dtype=torch.int32, device=dev)
    d_match = torch.empty(n, dtype=torch.uint8, device=dev)
    counts = [plan.n_windows] * world   # same shape on every rank (weak scaling)

    def step():
        eng.score_device(d_pcm, plan.total_bytes, n, 96000, trust, empty, d_probs, d_label, d_match)
        if world > 1:
            eng.sync()
            return sweep.gather_scores(d_probs, d_label, counts, dist)
        return None

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # RCCL builds its rings on the first collective: keep that out of the timed region even with --warmup 0
        d_probs.zero_()
        d_label.zero_()
        sweep.gather_scores(d_probs, d_label, counts, dist)
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_profile:
        eng.profile_reset()
        eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = None
    if not args.no_profile:
        prof = eng.profile_read()
        eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    valid_windows = plan.n_valid * world          # 1 window scored = 1 s of audio
    audio_hours_per_s = valid_windows * args.steps / 3600.0 / dt

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0

    out = {
        "metric": "audio-hours scored/sec per node", "value": audio_hours_per_s, "unit": "audio-hours/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: {args.files} recordings x {args.seconds} s per GPU "
                               f"({args.files * args.seconds / 3600:.1f} h), 48 kHz s16le mono, 2-s windows / 1-s hop",
                   "windows_per_gpu": plan.n_valid, "classes": C, "max_batch": args.max_batch,
                   "weights": "random-init NoiseDetector(4), He-uniform, classifier bias centred",
                   "sharding": "by file, round-robin" + ("; all_gather of score rows each step" if world > 1 else "")},
        "windows_per_s": valid_windows * args.steps / dt,
        "synth_seconds": t_gen,
    }

    # ---- per-kernel roofline from the HIP events recorded in the timed region ----
    if prof:
        fam = {}
        for s in prof:
            a = fam.setdefault(s["name"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            for k in ("ms", "launches", "flops", "bytes"):
                a[k] += s[k]
        tot_ms = sum(a["ms"] for a in fam.values())
        # dominant MFMA kernel = the igemm instantiation or the fused Winograd kernel with the most GPU time
        dom_name, dom = max(((k, v) for k, v in fam.items() if "igemm" in k or "wino_fused" in k), key=lambda kv: kv[1]["ms"])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py; FETCH_SIZE and
        # WRITE_SIZE cannot be collected inside this process).  Valid for launches of the profiled size only.
        traffic = None
        try:
            with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            if args.max_batch == pmc["windows_per_launch"]:
                traffic = pmc["kernels"][dom_name]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
        out["roofline"] = {"kernel": dom_name, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                           "traffic_unit": "HBM bytes per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE)",
                           "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
                           "avg_launch_ms": dom["ms"] / max(dom["launches"], 1), "launches": dom["launches"],
                           "share_of_gpu_time": dom["ms"] / tot_ms}
        def stage(k, v):
            tf = (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else None
            gb = (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None
            # the roof each stage is priced against: matrix cores for the convolutions, fp32 VALU issue
            # for the STFT/mel kernel (its HBM fraction is reported beside it), HBM for the elementwise kernels
            if "igemm" in k or "wino_fused" in k or "conv12" in k:
                bound, frac = "mfma", tf / PEAK_F32_MFMA_TFLOPS if tf else None
            elif "stft" in k:
                bound, frac = "valu", tf / PEAK_F32_VALU_TFLOPS if tf else None
            else:
                bound, frac = "hbm", gb / PEAK_HBM_GBPS if gb else None
            return {"name": k, "ms_per_step": v["ms"] / args.steps, "launches": v["launches"], "tflops": tf, "alg_GBps": gb,
                    "bound": bound, "frac": frac, "hbm_frac": gb / PEAK_HBM_GBPS if gb else None,
                    "share_of_gpu_time": v["ms"] / tot_ms}
        out["kernels"] = [stage(k, v) for k, v in fam.items()]
        out["layers"] = [{"layer": s["layer"], "kernel": s["name"], "ms_per_step": s["ms"] / args.steps,
                          "tflops": (s["flops"] / (s["ms"] * 1e-3) / 1e12) if s["ms"] > 0 else None}
                         for s in prof if s["layer"] > 0]
        cnn_flops = sum(s["flops"] for s in prof if s["layer"] > 0)
        cnn_ms = sum(s["ms"] for s in prof if s["layer"] > 0)
        out["cnn_tflops"] = cnn_flops / (cnn_ms * 1e-3) / 1e12 if cnn_ms else None      # FLOPs actually executed
        # direct-convolution-equivalent rate (5 171 281 920 FLOP per window); exceeds the
        # executed rate because the Winograd layers do 2.25x fewer multiplies for the same result
        out["cnn_effective_tflops"] = (5171281920.0 * plan.n_windows * args.steps) / (cnn_ms * 1e-3) / 1e12 if cnn_ms else None

    # ---- per-clip latency: one 2-s window through the host-pointer entry point ----
    if not args.no_extras:
        # (a one-clip engine, max_batch 1, as the daemon's one-Infer-per-segment pattern would create it)
        one = pcm_host[:96000].view(np.uint8)
        eng1 = model.Engine(W.pack_blob(flat, C), local_rank, 1)
        lat = []
        for _ in range(40):
            t1 = time.perf_counter()
            eng1.score(one, 1, trust=trust, empty_class=empty)
            lat.append(1e3 * (time.perf_counter() - t1))
        out["p50_clip_latency_ms"] = float(np.median(lat[5:]))
        out["p99_clip_latency_ms"] = float(np.percentile(lat[5:], 99))
        # the same clip through the reference API's two calls: Prepare (front-end handle), then Infer (model handle)
        fe1 = model.Engine(None, local_rank, 64)
        lat = []
        for _ in range(40):
            t1 = time.perf_counter()
            spec, _w = fe1.prepare(one)
            eng1.infer_logits(spec[None])
            lat.append(1e3 * (time.perf_counter() - t1))
        fe1.close()
        eng1.close()
        out["p50_prepare_plus_infer_ms"] = float(np.median(lat[5:]))

    # ---- CPU baseline: the oracle (port of the reference CPU path) on a bounded sample ----
    if args.cpu_windows > 0 and not args.no_extras and world == 1:   # rank 0 at N = 1 only
        import oracle   # test infrastructure; used here only as the timed CPU stand-in / checker
        nw = min(args.cpu_windows, args.seconds - 1)
        sample = pcm_host[: (nw + 1) * 48000]
        t1 = time.perf_counter()
        ref = oracle.score(flat, C, sample, nw, trust=trust, empty_class=empty)
        cdt = time.perf_counter() - t1
        got_label = d_label[:nw].cpu().numpy()
        got_probs = d_probs[:nw].cpu().numpy()
        srt = np.sort(ref["logits"], axis=1)
        decided = (srt[:, -1] - srt[:, -2]) > 4e-4
        out["cpu_baseline"] = {"value": nw / 3600.0 / cdt, "unit": "audio-hours/s", "cores": oracle.num_threads(),
                               "kind": "port",
                               "sample": f"first {nw} windows ({nw + 1} s) of recording 0, engine reused "
                                         f"(the reference re-parses the model every Infer call)",
                               "windows_per_s": nw / cdt}
        out["parity_sample"] = {"windows": int(nw), "labels_equal_on_decided": bool(np.array_equal(got_label[decided], ref["label"][decided])),
                                "near_ties_excluded": int((~decided).sum()),
                                "max_abs_prob_err": float(np.abs(got_probs - ref["probs"]).max()),
                                "label_histogram": np.bincount(ref["label"], minlength=C).tolist()}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
