#!/usr/bin/env python3
"""bench.py -- audio-hours scored per second on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (PCM -> STFT -> log-mel -> CNN -> softmax -> label) over
one rank's shard of synthetic recordings, PCM already resident in HBM.  Recordings are 48 kHz
mono s16le holding 16 kHz-band content, 600 s each (the reference's default
record_duration, src/state/config.go:83); fp32 end to end; random-init NoiseDetector(4).

  --config 2 (default)  BASELINE config 2, 10 h = 60 recordings PER GPU; with N > 1 this is the
                        weak-scaling mode ("scaling": "weak").
  --config 3            BASELINE config 3, --hours H (default 1000) = 6 H recordings in ALL,
                        sharded by file, round-robin, over the N ranks ("scaling": "strong");
                        each rank synthesises its shard in chunks straight into HBM.

  --config 5            BASELINE config 5, training-set preprocessing (ai/train.py:36-52): --clips (default 10 000)
                        tagged 192 000-byte clips -> [n,1,128,188] f32 log-mel in the librosa flavour, clips sharded
                        by index over the N ranks (contiguous blocks, 1 250 each at 8), every rank's features stay
                        on its GPU (--gather collects them on rank 0: 962 MB); the line reports clips/s.

N > 1: one process per GPU.  Started by torchrun, the ranks come from the environment; started
as plain `python bench.py --gpus N`, this process only spawns N fresh rank processes (it never
touches a GPU itself) and relays rank 0's line.  Every step ends with the one exchange of the
sweep, the gather of the per-window score rows on rank 0 (RCCL).  The world size RCCL reports
must equal --gpus, else the run fails.  Rank 0 prints ONE JSON line.

--dry-run: the launch / shard plan / gather logic on CPU (gloo, a stand-in scorer, no GPU, no
measurement: "value" is null) -- what tests/test_dist.py drives.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_F32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (vector), packed FMA
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E
LABELS = ["big_dog", "combination", "empty", "small_dog"]   # src/model/test_model.labels:1
TRUST = 0.5                    # inspect_trust default, src/state/config.go:86
NEAR_TIE = 4e-4                # top-2 logit margin below which fp32 and f64 may legitimately disagree (2 x the logit tolerance)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5))
    ap.add_argument("--clips", type=int, default=10000, help="config 5: clips in the whole job")
    ap.add_argument("--gather", action="store_true", help="config 5: gather the feature rows on rank 0 inside every step")
    ap.add_argument("--files", type=int, default=60, help="config 2: recordings per GPU")
    ap.add_argument("--hours", type=float, default=1000.0, help="config 3: hours of audio in the whole job")
    ap.add_argument("--seconds", type=int, default=600, help="seconds per recording")
    ap.add_argument("--max-batch", type=int, default=0,
                    help="windows per GPU pass (the engine holds ~11 MB of device memory per window); 0 = the largest of 16384 / 8192 / 4096 "
                         "that fits beside the PCM in 80 %% of the free HBM: 180 GB of the 288 at 16384, worth 1.9 %% over 4096 and 0.8 %% over 8192")
    ap.add_argument("--cpu-windows", type=int, default=599, help="windows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed sweep: no classifier centring, latency probe, config 1/4/5 legs or CPU baseline "
                         "(use under rocprofv3 so that every launch it sees has the timed region's shapes)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of launch/plan/gather (gloo, stand-in scorer)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# N > 1 from the plain command: spawn the ranks (this process never initialises a GPU)
# --------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args) -> int:
    import tempfile
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile() as line0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=line0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies would leave the others waiting in the rendezvous or a collective: stop them (exact PIDs)
        rcs = [None] * args.gpus
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            if any(rc not in (None, 0) for rc in rcs):
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        p.terminate()
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        try:
                            rcs[r] = p.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            rcs[r] = p.wait()
                break
            time.sleep(0.05)
        line0.seek(0)
        sys.stdout.write(line0.read().decode())
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {bad}\n")
        return 1
    return 0


# --------------------------------------------------------------------------------------------------
# stand-in scorer of the dry run: a deterministic function of (file, second) so that rank 0 can
# recompute every row it should have received
# --------------------------------------------------------------------------------------------------
def fake_rows(file_ids: np.ndarray, second: np.ndarray, C: int):
    x = (file_ids.astype(np.uint64) * np.uint64(1000003) + second.astype(np.uint64)) * np.uint64(0x9E3779B97F4A7C15)
    cols = [((x >> np.uint64(8 * c + 8)) & np.uint64(0xFF)).astype(np.float64) + 1.0 for c in range(C)]
    p = np.stack(cols, axis=1)
    p /= p.sum(axis=1, keepdims=True)
    label = p.argmax(axis=1).astype(np.int32)
    match = ((label != LABELS.index("empty")) & (p.max(axis=1) > TRUST)).astype(np.uint8)
    return p, label, match


def main() -> int:
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}\n")
        return 2

    import torch
    import torch.distributed as dist
    from apr_amd import sweep

    C = len(LABELS)
    empty = LABELS.index("empty")
    dry = args.dry_run

    if dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            sys.stderr.write(f"bench.py: rank {rank} has no GPU (only {torch.cuda.device_count()} visible, --gpus {args.gpus})\n")
            return 2
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:      # the world RCCL actually built
            sys.stderr.write(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}\n")
            return 2

    if args.config == 5:
        return run_config5(args, rank, local_rank, world, dev, dry, torch, dist)

    # ---- the job's recordings and this rank's shard (by file, round-robin: SURVEY 8(e)) ----
    n_files_total = args.files * world if args.config == 2 else max(world, int(round(args.hours * 3600 / args.seconds)))
    sp = sweep.plan_shards(n_files_total, args.seconds, world, rank)
    plan, my_files, counts = sp.plan, sp.files, sp.counts
    n = plan.n_windows
    file_bytes = args.seconds * 96000

    if dry:
        probs_np, label_np, match_np = fake_rows(plan.file_ids, plan.second, C)
        d_probs, d_label, d_match = torch.from_numpy(probs_np), torch.from_numpy(label_np), torch.from_numpy(match_np)
        eng = None
        t_gen = 0.0
        flat = None
        pcm_head = None
        args.max_batch = args.max_batch or 4096
    else:
        from apr_amd import model, weights as W
        from concurrent.futures import ThreadPoolExecutor
        flat = W.random_init(C, 0, "he")
        # synthesised in chunks of a few recordings straight into HBM: a shard of config 3 (43 GB on 8 GPUs, 173 GB on 2)
        # never exists in host memory.  apr_synth_pcm releases the GIL: 8 generator threads.
        t0 = time.time()
        d_pcm = torch.empty(len(my_files) * file_bytes + 64, dtype=torch.uint8, device=dev)
        chunk = 8
        pcm_head = None
        with ThreadPoolExecutor(chunk) as pool:
            for i0 in range(0, len(my_files), chunk):
                part = list(pool.map(lambda f: model.synth_pcm(0, f, args.seconds), my_files[i0:i0 + chunk]))
                host = np.concatenate(part).view(np.uint8)
                if pcm_head is None:
                    pcm_head = part[0]                      # this rank's first recording, kept for the CPU legs
                d_pcm[i0 * file_bytes: i0 * file_bytes + host.size].copy_(torch.from_numpy(host))
        torch.cuda.synchronize()
        t_gen = time.time() - t0

        # centre the random-init classifier on this data so that labels vary between windows
        if not args.no_extras:
            eng0 = model.Engine(W.pack_blob(flat, C), local_rank, 64)
            # (recording 0 on every rank, so that all ranks score with identical weights)
            cal = eng0.infer_logits(eng0.prepare_batch(model.synth_pcm(0, 0, 65), 64))
            eng0.close()
            flat = W.center_logits(flat, C, cal.mean(axis=0))
        # pass size: a launch costs ~0.25 ms beyond its windows (the XCDs do not end together), so fewer, larger passes pay -- HBM is there for it
        blob = W.pack_blob(flat, C)
        if args.max_batch > 0:
            eng = model.Engine(blob, local_rank, args.max_batch)
        else:
            free_b, _total_b = torch.cuda.mem_get_info(dev)
            eng = None
            for mb in (16384, 8192, 4096):
                if mb > 4096 and (mb * 11.5e6 + 8e9 > 0.8 * free_b or mb // 2 >= n):
                    continue
                try:
                    eng = model.Engine(blob, local_rank, mb)
                    args.max_batch = mb
                    break
                except Exception as ex:      # the estimate was wrong for this box: the next size down
                    print(f"bench: engine of {mb} windows not created ({ex}); trying a smaller pass", file=sys.stderr)
            if eng is None:
                raise RuntimeError("bench: no engine could be created")
        d_probs = torch.empty((n, C), dtype=torch.float64, device=dev)
        d_label = torch.empty(n, dtype=torch.int32, device=dev)
        d_match = torch.empty(n, dtype=torch.uint8, device=dev)

    gathered = [None]

    def step():
        if eng is not None:
            eng.score_device(d_pcm, plan.total_bytes, n, 96000, TRUST, empty, d_probs, d_label, d_match)
        if world > 1:
            if eng is not None:
                eng.sync()      # the engine launches on its own stream; the collective runs on torch's
            gathered[0] = sweep.gather_scores(d_probs, d_label, d_match, counts, dist)

    def fence():
        if eng is not None:
            eng.sync()
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if eng is not None:
            torch.cuda.synchronize()

    if world > 1 and not dry:
        # RCCL builds its rings on the first collective: keep that out of the timed region even with --warmup 0
        d_probs.zero_(); d_label.zero_(); d_match.zero_()
        sweep.gather_scores(d_probs, d_label, d_match, counts, dist)
    for _ in range(args.warmup):
        step()
    fence()
    if eng is not None and not args.no_profile:
        eng.profile_reset()
        eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = None
    if eng is not None and not args.no_profile:
        prof = eng.profile_read()
        eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    valid_windows = sp.valid_total                 # 1 window scored = 1 s of audio
    audio_hours_per_s = valid_windows * args.steps / 3600.0 / dt

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0

    if args.config == 2:
        workload = (f"BASELINE config 2: {args.files} recordings x {args.seconds} s per GPU ({args.files * args.seconds / 3600:.1f} h"
                    f"{' per GPU, weak scaling' if world > 1 else ''}), 48 kHz s16le mono, 2-s windows / 1-s hop")
    else:
        workload = (f"BASELINE config 3: {n_files_total} recordings x {args.seconds} s = {n_files_total * args.seconds / 3600:.1f} h in all, "
                    f"sharded by file over {world} GPU(s), 48 kHz s16le mono, 2-s windows / 1-s hop")
    out = {
        "metric": "audio-hours scored/sec per node", "value": None if dry else audio_hours_per_s, "unit": "audio-hours/s",
        "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak" if args.config == 2 else "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config, "recordings_total": n_files_total,
                   "windows_total": valid_windows, "windows_rank0": plan.n_valid, "classes": C, "max_batch": args.max_batch,
                   "weights": "random-init NoiseDetector(4), He-uniform, classifier bias centred",
                   "sharding": "by file, round-robin" + ("; gather of score rows (probs f64, label, match) on rank 0 each step" if world > 1 else "")},
        "windows_per_s": valid_windows * args.steps / dt,
        "synth_seconds": t_gen,
    }
    if dry:
        out["dry_run"] = True

    # ---- what rank 0 holds after the gather: every rank's rows, in that rank's stream order ----
    if world > 1:
        pl, ll, ml = gathered[0]
        ok = all(pl[r].shape[0] == counts[r] for r in range(world))
        if dry:   # the stand-in scorer is recomputable: every row must be the row of its (file, second)
            for r in range(world):
                rp = sweep.plan_shards(n_files_total, args.seconds, world, r).plan
                p, l, m = fake_rows(rp.file_ids, rp.second, C)
                ok = ok and np.array_equal(pl[r].numpy(), p) and np.array_equal(ll[r].numpy(), l) and np.array_equal(ml[r].numpy(), m)
        else:     # rank 0's own rows came back unchanged; every valid row of every rank is a probability row
            ok = ok and bool(torch.equal(pl[0], d_probs)) and bool(torch.equal(ll[0], d_label)) and bool(torch.equal(ml[0], d_match))
            for r in range(world):
                va = torch.from_numpy(sweep.plan_shards(n_files_total, args.seconds, world, r).plan.valid).to(pl[r].device)
                s = pl[r][va].sum(dim=1)
                ok = ok and bool(((s - 1.0).abs() < 1e-9).all()) and bool((ll[r][va] >= 0).all()) and bool((ll[r][va] < C).all())
        out["gather"] = {"rows": [int(c) for c in counts], "row_bytes": sweep.row_bytes(C), "bytes_per_step": int(sum(counts)) * sweep.row_bytes(C),
                         "verified": bool(ok)}
        if not ok:
            print(json.dumps(out), flush=True)
            sys.stderr.write("bench.py: gathered rows failed verification\n")
            dist.destroy_process_group()
            return 3

    if dry:
        print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    from apr_amd import model, weights as W

    # ---- per-kernel roofline from the HIP events recorded in the timed region ----
    if prof:
        fam = {}
        for s in prof:
            a = fam.setdefault(s["name"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            for k in ("ms", "launches", "flops", "bytes"):
                a[k] += s[k]
        tot_ms = sum(a["ms"] for a in fam.values())
        # dominant MFMA kernel = the igemm instantiation or the fused Winograd kernel with the most GPU time
        dom_name, dom = max(((k, v) for k, v in fam.items() if is_mfma_kernel(k)), key=lambda kv: kv[1]["ms"])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py; FETCH_SIZE and
        # WRITE_SIZE cannot be collected inside this process).  Valid for launches of the profiled size only.
        traffic, traffic_src = None, None
        import glob
        for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                with open(path) as fh:
                    pmc = json.load(fh)
                if args.max_batch == pmc["windows_per_launch"]:
                    traffic = pmc["kernels"][dom_name]["hbm_bytes_per_launch"]
                    traffic_src = os.path.relpath(path, REPO)
                    break
            except (OSError, KeyError, ValueError):
                continue
        out["roofline"] = {"kernel": dom_name, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                           "traffic_unit": "HBM bytes per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
                           # `traffic` is a full pass of max_batch windows; the timed launches include each step's shorter last pass
                           "traffic_windows_per_launch": args.max_batch,
                           "algorithmic_bytes_per_full_launch": dom["bytes"] / max(dom["launches"], 1) * args.max_batch
                           / (n / float(-(-n // args.max_batch))),
                           "avg_launch_ms": dom["ms"] / max(dom["launches"], 1), "launches": dom["launches"],
                           "share_of_gpu_time": dom["ms"] / tot_ms}
        # the fp32 matrix rate the box delivers right after the timed steps (MFMAs from registers, no memory): a throttled box
        # lowers this and roofline.achieved alike, a slower build lowers only the latter
        probe = eng.mfma_probe()
        out["mfma_probe"] = {"tflops": probe, "frac_of_peak": probe / PEAK_F32_MFMA_TFLOPS,
                             "dominant_kernel_vs_probe": ach / probe if probe else None}

        def stage(k, v):
            tf = (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else None
            gb = (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None
            # the roof each stage is priced against: matrix cores for the convolutions, fp32 VALU issue
            # for the STFT/mel kernel (its HBM fraction is reported beside it), HBM for the elementwise kernels
            if is_mfma_kernel(k):
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
        # executed rate because the Winograd layers do fewer multiplies for the same result
        out["cnn_effective_tflops"] = (5171281920.0 * plan.n_windows * args.steps) / (cnn_ms * 1e-3) / 1e12 if cnn_ms else None

    extras = not args.no_extras and world == 1
    blob = W.pack_blob(flat, C)

    # ---- per-clip latency: one 2-s window through the host-pointer entry point ----
    if not args.no_extras:
        # (a one-clip engine, max_batch 1, as the daemon's one-Infer-per-segment pattern would create it)
        one = pcm_head[:96000].view(np.uint8)
        eng1 = model.Engine(blob, local_rank, 1)
        lat = []
        for _ in range(40):
            t1 = time.perf_counter()
            eng1.score(one, 1, trust=TRUST, empty_class=empty)
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

    # ---- CPU baseline: the oracle (port of the reference CPU path) on a bounded sample; parity of the timed sweep's
    #      own outputs against it, near-ties included ----
    if args.cpu_windows > 0 and extras:
        import oracle   # test infrastructure; used here only as the timed CPU stand-in / checker
        nw = min(args.cpu_windows, args.seconds - 1)
        sample = pcm_head[: (nw + 1) * 48000]
        t1 = time.perf_counter()
        ref = oracle.score(flat, C, sample, nw, trust=TRUST, empty_class=empty)
        cdt = time.perf_counter() - t1
        cores = oracle.num_threads()
        n1 = min(16, nw)                       # the same path on ONE host core, a shorter sample
        oracle.set_num_threads(1)
        t1 = time.perf_counter()
        oracle.score(flat, C, sample[: (n1 + 1) * 48000], n1, trust=TRUST, empty_class=empty)
        cdt1 = time.perf_counter() - t1
        oracle.set_num_threads(cores)
        out["cpu_baseline"] = {"value": nw / 3600.0 / cdt, "unit": "audio-hours/s", "cores": cores, "kind": "port",
                               "sample": f"first {nw} windows ({nw + 1} s) of recording 0, engine reused "
                                         f"(the reference re-parses the model every Infer call)",
                               "windows_per_s": nw / cdt,
                               "one_core": {"value": n1 / 3600.0 / cdt1, "unit": "audio-hours/s", "cores": 1, "windows_per_s": n1 / cdt1,
                                            "sample": f"first {n1} windows of recording 0"}}
        out["parity_sample"] = parity_report(ref, d_probs[:nw].cpu().numpy(), d_label[:nw].cpu().numpy(), d_match[:nw].cpu().numpy(), C)

        # ---- BASELINE config 1: one 60-s clip (59 windows) through the CPU path, the GPU path beside it ----
        clip = model.synth_pcm(1, 0, 60)
        t1 = time.perf_counter()
        r1 = oracle.score(flat, C, clip, 59, trust=TRUST, empty_class=empty)
        c1 = time.perf_counter() - t1
        t1 = time.perf_counter()
        g1 = eng.score(clip, 59, trust=TRUST, empty_class=empty)
        c1g = time.perf_counter() - t1
        out["config1"] = {"workload": "BASELINE config 1: one synthetic 60-s recording, 59 windows, CPU path (oracle port)",
                          "cpu_seconds": c1, "cpu_cores": cores, "cpu_audio_hours_per_s": 59 / 3600.0 / c1,
                          "gpu_seconds_host_pointers": c1g,
                          "matches": int(r1["match"].sum()), "label_histogram": np.bincount(r1["label"], minlength=C).tolist(),
                          "parity": parity_report(r1, g1["probs"], g1["label"], g1["match"], C)}

    # ---- the other BASELINE configs on this GPU, driver-timed (details: tools/bench_configs.py) ----
    if extras:
        out["pcie_inclusive"] = leg_host_sweep(model, eng, pcm_head, args, empty)
    eng.close()
    if extras:
        del d_pcm
        out["config4"] = leg_live_streams(model, blob, local_rank, empty)
        out["config5"] = leg_training_features(model, torch, dev, local_rank)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def is_mfma_kernel(name: str) -> bool:
    """kernels whose time is matrix-core time (the Winograd transform-only kernels are HBM-bound)"""
    return "igemm" in name or "conv12" in name or ("wino" in name and "output" not in name and "input" not in name)


def parity_report(ref, probs, label, match, C):
    """GPU rows against the oracle's for the same windows.  Mismatches are counted on ALL windows; the flags say whether
    every mismatch sits on a near-tie (top-2 logit margin <= NEAR_TIE, or confidence within 1e-3 of the trust threshold)."""
    srt = np.sort(ref["logits"], axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > NEAR_TIE
    near_trust = np.abs(ref["probs"].max(axis=1) - TRUST) < 1e-3
    lab_bad = label != ref["label"]
    mat_bad = match != ref["match"]
    return {"windows": int(label.size),
            "label_mismatches_all_windows": int(lab_bad.sum()), "match_mismatches_all_windows": int(mat_bad.sum()),
            "labels_equal_on_decided": bool(not lab_bad[decided].any()),
            "matches_equal_on_decided": bool(not mat_bad[decided & ~near_trust].any()),
            "near_ties": int((~decided).sum()), "near_trust_threshold": int(near_trust.sum()),
            "max_abs_prob_err": float(np.abs(probs - ref["probs"]).max()),
            "label_histogram": np.bincount(ref["label"], minlength=C).tolist()}


def leg_host_sweep(model, eng, pcm_head, args, empty):
    """PCIe-inclusive rate (never `value`): apr_sweep_files over recordings in pageable host memory."""
    # 30 recordings = 9 passes of 2048 windows: the first pass's staging copy has nothing to hide behind (12 recordings = 4 passes
    # read 8.9 audio-hours/s for that reason alone); a directory sweep is far longer still
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as pool:
        files = [pcm_head] + list(pool.map(lambda f: model.synth_pcm(0, 100 + f, args.seconds), range(29)))
    eng.sweep_files(files[:2], trust=TRUST, empty_class=empty)           # allocates the staging buffers
    t1 = time.perf_counter()
    res = eng.sweep_files(files, trust=TRUST, empty_class=empty)
    dt = time.perf_counter() - t1
    nwin = int(res["label"].size)
    return {"entry": "apr_sweep_files", "files": len(files), "windows": nwin, "seconds": dt, "audio_hours_per_s": nwin / 3600.0 / dt,
            "pcm_GBps": sum(f.nbytes for f in files) / dt / 1e9,
            "note": "pageable host PCM -> pinned staging -> H2D on a copy stream under the previous pass -> labels in host memory"}


def leg_live_streams(model, blob, device, empty, streams=256, ticks=24):
    """BASELINE config 4: 256 live streams, one 1-s segment each per tick (src/daemon/daemon.go:104-125)."""
    from apr_amd import sweep
    eng = model.Engine(blob, device, streams)
    secs = 6
    src = np.stack([model.synth_pcm(4, s, secs).view(np.uint8).reshape(secs, 96000) for s in range(streams)])
    sc = sweep.StreamScorer(eng, streams, trust=TRUST, empty_class=empty)
    sc.tick(np.ascontiguousarray(src[:, 0]))
    lat = []
    for t in range(ticks):
        seg = np.ascontiguousarray(src[:, 1 + t % (secs - 1)])       # the tick's segments, pageable host memory
        t1 = time.perf_counter()
        sc.tick(seg)
        lat.append(1e3 * (time.perf_counter() - t1))
    eng.close()
    lat = np.array(lat[3:])
    return {"workload": f"BASELINE config 4: {streams} live streams, one 1-s segment each per tick, window = previous || new",
            "ticks": int(lat.size), "latency_ms_p50": float(np.percentile(lat, 50)), "latency_ms_p99": float(np.percentile(lat, 99)),
            "latency_ms_max": float(lat.max()), "real_time_budget_ms": 1000.0,
            "sustained_windows_per_s": float(streams / (np.mean(lat) * 1e-3)),
            "note": "apr_stream_tick, segment arrival in host memory to labels in host memory"}


CLIP_BYTES = 192000            # a tagged .dat clip: two 1-s segments (src/review/action.go:212-215)
CLIP_UNIQUE = 500              # distinct synthetic clips; clip i of the data set is synthetic clip i % 500
CLIP_ALG_BYTES = CLIP_BYTES + 128 * 188 * 4     # algorithmic bytes per clip: PCM in + [128,188] f32 out


def config5_clips(model, torch, dev, lo, hi):
    """Clips [lo, hi) of the synthetic data set, packed at stride 192000, resident in HBM."""
    host = np.concatenate([model.synth_pcm(5, i, 2) for i in range(CLIP_UNIQUE)]).view(np.uint8).reshape(CLIP_UNIQUE, CLIP_BYTES)
    idx = np.arange(lo, hi) % CLIP_UNIQUE
    return torch.from_numpy(np.ascontiguousarray(host[idx]).reshape(-1)).to(dev)


def leg_training_features(model, torch, dev, device, clips=10000):
    """BASELINE config 5: 10 000 tagged .dat clips -> librosa-flavour log-mel (ai/train.py:36-52), clips resident in HBM."""
    from apr_amd._capi import FLAVOUR_LIBROSA
    fe = model.Engine(None, device, 1024)
    d_pcm = config5_clips(model, torch, dev, 0, clips)
    d_out = torch.empty((clips, 128, 188), dtype=torch.float32, device=dev)
    fe.prepare_batch_device(d_pcm, d_pcm.numel(), clips, CLIP_BYTES, d_out, FLAVOUR_LIBROSA)
    fe.sync()
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        fe.prepare_batch_device(d_pcm, d_pcm.numel(), clips, CLIP_BYTES, d_out, FLAVOUR_LIBROSA)
    fe.sync()
    dt = (time.perf_counter() - t1) / reps
    fe.close()
    return {"workload": f"BASELINE config 5: {clips} x 192000-byte clips -> [n,1,128,188] f32, librosa flavour, one GPU",
            "ms": 1e3 * dt, "clips_per_s": clips / dt, "algorithmic_GBps": clips * CLIP_ALG_BYTES / dt / 1e9}


def fake_features(lo, hi):
    """dry run: a recomputable stand-in for the feature rows of clips [lo, hi) (small rows: the plan and the gather are under test)"""
    i = np.arange(lo, hi, dtype=np.float32)[:, None, None]
    return (i + np.arange(2, dtype=np.float32)[None, :, None] * 0.25 + np.arange(3, dtype=np.float32)[None, None, :] * 0.5)


def run_config5(args, rank, local_rank, world, dev, dry, torch, dist) -> int:
    """BASELINE config 5 over the ranks: one step = every rank's clips through the librosa-flavour front end."""
    from apr_amd import sweep
    cp = sweep.plan_clips(args.clips, world, rank)
    n = cp.n_local
    fe = None
    if dry:
        d_out = torch.from_numpy(fake_features(cp.lo, cp.hi))
    else:
        from apr_amd import model
        from apr_amd._capi import FLAVOUR_LIBROSA
        fe = model.Engine(None, local_rank, 1024)
        d_pcm = config5_clips(model, torch, dev, cp.lo, cp.hi)
        d_out = torch.empty((n, 128, 188), dtype=torch.float32, device=dev)
    gathered = [None]

    def step():
        if fe is not None:
            fe.prepare_batch_device(d_pcm, d_pcm.numel(), n, CLIP_BYTES, d_out, FLAVOUR_LIBROSA)
        if world > 1 and args.gather:
            if fe is not None:
                fe.sync()
            gathered[0] = sweep.gather_features(d_out, cp.counts, dist)

    def fence():
        if fe is not None:
            fe.sync()
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if fe is not None:
            torch.cuda.synchronize()

    if world > 1 and not dry:
        dist.all_reduce(torch.zeros(1, device=dev))          # RCCL builds its rings on the first collective
    for _ in range(args.warmup):
        step()
    fence()
    if fe is not None and not args.no_profile:
        fe.profile_reset()
        fe.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = None
    if fe is not None and not args.no_profile:
        prof = fe.profile_read()
        fe.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0

    clips_per_s = args.clips * args.steps / dt
    out = {"metric": "training clips preprocessed/sec per node (BASELINE config 5)", "value": None if dry else clips_per_s, "unit": "clips/s",
           "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE config 5: {args.clips} x {CLIP_BYTES}-byte tagged clips -> [n,1,128,188] f32 log-mel, librosa flavour "
                                  f"(ai/train.py:36-52), sharded by clip index over {world} GPU(s)",
                      "baseline_config": 5, "clips_total": args.clips, "clips_per_rank": [int(c) for c in cp.counts],
                      "sharding": "by clip index, contiguous blocks; features stay on the GPU that made them"
                                  + ("; gathered on rank 0 each step" if args.gather and world > 1 else "")},
           "audio_hours_per_s": None if dry else clips_per_s * 2.0 / 3600.0}
    if dry:
        out["dry_run"] = True
    ok = True
    if world > 1 and args.gather:
        g = gathered[0]
        ok = g is not None and g.shape[0] == args.clips
        if dry:
            ok = ok and np.array_equal(g.numpy(), fake_features(0, args.clips))
        else:
            ok = ok and bool(torch.equal(g[: cp.counts[0]], d_out)) and bool(((g >= 0) & (g <= 1)).all())
        out["gather"] = {"rows": [int(c) for c in cp.counts], "bytes_per_step": int(args.clips) * int(np.prod(d_out.shape[1:])) * 4, "verified": bool(ok)}
    if prof:
        fam = {}
        for s in prof:
            a = fam.setdefault(s["name"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            for k in ("ms", "launches", "flops", "bytes"):
                a[k] += s[k]
        tot_ms = sum(a["ms"] for a in fam.values())
        dom_name, dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
        gbps = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        # SURVEY 8(d) prices the front end against HBM; DESIGN.md section 4 shows the kernel is VALU/LDS-bound (its HBM floor is far below)
        out["roofline"] = {"kernel": dom_name, "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                           "traffic": None, "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
                           "avg_launch_ms": dom["ms"] / max(dom["launches"], 1), "launches": dom["launches"], "share_of_gpu_time": dom["ms"] / tot_ms,
                           "valu_tflops": dom["flops"] / (dom["ms"] * 1e-3) / 1e12, "valu_frac": dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / PEAK_F32_VALU_TFLOPS}
        out["kernels"] = [{"name": k, "ms_per_step": v["ms"] / args.steps, "launches": v["launches"],
                           "alg_GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else None,
                           "hbm_frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS if v["ms"] > 0 else None,
                           "share_of_gpu_time": v["ms"] / tot_ms} for k, v in fam.items()]
        out["whole_path_alg_GBps"] = args.clips * args.steps * CLIP_ALG_BYTES / dt / 1e9
    if not dry and world == 1 and args.cpu_windows > 0 and not args.no_extras:
        import oracle   # test infrastructure; here only the timed CPU stand-in / checker
        nc = min(256, n)
        host = d_pcm[: nc * CLIP_BYTES].cpu().numpy()
        t1 = time.perf_counter()
        ref = oracle.prepare_librosa_batch(host, nc)
        cdt = time.perf_counter() - t1
        got = d_out[:nc].cpu().numpy()
        out["cpu_baseline"] = {"value": nc / cdt, "unit": "clips/s", "cores": oracle.num_threads(), "kind": "port",
                               "sample": f"first {nc} clips through the oracle's librosa-flavour restatement (librosa itself is absent: parity unpinned)"}
        out["parity_sample"] = {"clips": nc, "max_abs_err": float(np.abs(got - ref).max()), "tolerance": 5e-5}
    if fe is not None:
        fe.close()
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())
