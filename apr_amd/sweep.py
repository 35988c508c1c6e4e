"""Sweep driver: recordings -> windows -> batched GPU scoring -> labels / timestamps.

Mirrors the only batch sweep the reference has, ``ai/inspect.py:73-112`` (window k covers
seconds [k, k+2); matches are printed as ``Matched {model}/{label} in {file} @{k+1} sec``),
with the Go daemon's decision rule (``src/daemon/daemon.go:189-207``) available as ``match``.

Multi-GPU: recordings shard by FILE (round-robin by index) so the 1-s window overlap never
crosses a shard; the only exchange is the final gather of per-window score rows
(``gather_scores``), one process per GPU over torch.distributed (RCCL on GPUs, gloo on CPU
in the tests).  Scoring itself is injected (``scorer``) so the sharding/gather logic can be
exercised without a GPU; the product scorer is ``Engine.score``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

BYTES_PER_SECOND = 96000
SAMPLE_SIZE = 192000


def windows_in(n_bytes: int, stride: int = BYTES_PER_SECOND) -> int:
    """ai/inspect.py:108-109: ``range(0, len(raw)-bps, bps)``."""
    return 0 if n_bytes <= stride else len(range(0, n_bytes - stride, stride))


def shard_files(n_files: int, world_size: int, rank: int) -> List[int]:
    """Round-robin by file index."""
    return list(range(rank, n_files, world_size))


@dataclass
class StreamPlan:
    """Several recordings laid end to end in one buffer and scored as ONE stream of windows.
    Windows that straddle two recordings are computed and then discarded (``valid`` is False):
    that keeps every GPU pass at full batch size for < 0.2 % wasted work on 10-minute files."""
    file_ids: np.ndarray      # [n_stream_windows] file index of the window's first second
    second: np.ndarray        # [n_stream_windows] 0-based second within that file
    valid: np.ndarray         # [n_stream_windows] bool
    total_bytes: int
    n_windows: int            # stream windows (valid + straddling)

    @property
    def n_valid(self) -> int:
        return int(self.valid.sum())


def plan_stream(file_bytes: Sequence[int], file_ids: Optional[Sequence[int]] = None) -> StreamPlan:
    """Plan for whole-second recordings (every length a multiple of 96000 bytes)."""
    file_ids = list(range(len(file_bytes))) if file_ids is None else list(file_ids)
    fid, sec, valid = [], [], []
    for f, nb in zip(file_ids, file_bytes):
        if nb % BYTES_PER_SECOND:
            raise ValueError("plan_stream needs whole-second recordings; score ragged files one by one")
        t = nb // BYTES_PER_SECOND
        fid += [f] * t
        sec += list(range(t))
        valid += [True] * (t - 1) + [False] if t else []
    total = int(sum(file_bytes))
    n = windows_in(total)
    return StreamPlan(np.asarray(fid[:n], np.int64), np.asarray(sec[:n], np.int64), np.asarray(valid[:n], bool), total, n)


def format_matches(model_name: str, labels: Sequence[str], file_names: Dict[int, str], plan: StreamPlan,
                   label: np.ndarray, match: Optional[np.ndarray] = None) -> List[str]:
    """Lines as ai/inspect.py:77-84 prints them (1-based window index; 'empty' skipped).
    With ``match`` given, only windows passing the daemon's trust rule are listed."""
    out = []
    for k in np.nonzero(plan.valid)[0]:
        lab = int(label[k])
        if lab < 0 or labels[lab] == "empty":
            continue
        if match is not None and not match[k]:
            continue
        out.append(f"Matched {model_name}/{labels[lab]} in {file_names[int(plan.file_ids[k])]} @{int(plan.second[k]) + 1} sec")
    return out


def daemon_match_line(name: str, best_class: str, conf: float) -> str:
    """src/daemon/daemon.go:204"""
    return "[INFO]\t SCANNER %s: MATCH found! Class: %s (Conf: %.4f)" % (name, best_class, conf)


# ----------------------------------------------------------------------------------------------
# multi-process gather (one process per GPU)
# ----------------------------------------------------------------------------------------------
def row_bytes(n_classes: int) -> int:
    """One gathered score row: C float64 probabilities, the int32 label, the u8 match flag."""
    return 8 * n_classes + 5


def pack_rows(probs, label, match, cap: Optional[int] = None):
    """[n, 8C+5] uint8 rows (zero-padded to ``cap`` rows): the exact bytes of probs / label / match."""
    import torch
    n, C = probs.shape
    cap = n if cap is None else cap
    rows = torch.zeros((cap, row_bytes(C)), dtype=torch.uint8, device=probs.device)
    rows[:n, : 8 * C] = probs.contiguous().view(torch.uint8).reshape(n, 8 * C)
    rows[:n, 8 * C: 8 * C + 4] = label.contiguous().view(torch.uint8).reshape(n, 4)
    rows[:n, 8 * C + 4] = match
    return rows


def unpack_rows(rows, n_classes: int):
    import torch
    C = n_classes
    n = rows.shape[0]
    probs = rows[:, : 8 * C].contiguous().view(torch.float64).reshape(n, C)
    label = rows[:, 8 * C: 8 * C + 4].contiguous().view(torch.int32).reshape(n)
    return probs, label, rows[:, 8 * C + 4].contiguous()


def gather_scores(probs, label, match, counts: Sequence[int], dist=None, dst: int = 0):
    """Gather the per-window score rows of every rank on ``dst``: the one exchange of the sweep.

    probs [n_local, C] float64, label [n_local] int32 and match [n_local] uint8 are torch tensors
    on the rank's device; ``counts[r]`` is rank r's window count (known to every rank from the
    deterministic file partition, so no sizes are exchanged).  Rows travel as packed bytes
    (``row_bytes(C)`` each), padded to max(counts): one equal-size gather to the root -- over xGMI
    every sender uses its own direct link to ``dst``.  Returns three lists (one entry per rank) on
    dst, (None, None, None) elsewhere.
    """
    import torch
    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    if world == 1:
        return [probs], [label], [match]
    if len(counts) != world or probs.shape[0] != counts[rank]:
        raise ValueError("gather_scores: counts must list every rank's rows")
    C = probs.shape[1]
    cap = int(max(counts))
    packed = pack_rows(probs, label, match, cap)
    bufs = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, bufs, dst=dst)
    if rank != dst:
        return None, None, None
    out = [unpack_rows(bufs[r][: counts[r]], C) for r in range(world)]
    return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]


# ----------------------------------------------------------------------------------------------
# BASELINE config 3: a directory of equal-length recordings swept by `world` ranks
# (the loop of ai/inspect.py:20-70 over a directory, sharded by file: SURVEY 8(d)/(e))
# ----------------------------------------------------------------------------------------------
@dataclass
class ShardPlan:
    files: List[int]            # this rank's recordings (global file indices), in scoring order
    plan: StreamPlan            # its recordings laid end to end
    counts: List[int]           # stream windows of every rank (what gather_scores needs)
    valid_total: int            # valid windows of the whole job


def plan_shards(n_files: int, seconds: int, world: int, rank: int) -> ShardPlan:
    counts, valid_total, mine = [], 0, None
    for r in range(world):
        files = shard_files(n_files, world, r)
        p = plan_stream([seconds * BYTES_PER_SECOND] * len(files), files)
        counts.append(p.n_windows)
        valid_total += p.n_valid
        if r == rank:
            mine = (files, p)
    return ShardPlan(mine[0], mine[1], counts, valid_total)


# ----------------------------------------------------------------------------------------------
# BASELINE config 5: the training set's tagged clips preprocessed by `world` ranks
# (ai/train.py:36-52 extracts one clip's features per __getitem__; SURVEY 8(e): "Cfg 5: by clip index")
# ----------------------------------------------------------------------------------------------
@dataclass
class ClipPlan:
    lo: int                     # this rank's clips are [lo, hi) of the data set's index order
    hi: int
    counts: List[int]           # clips of every rank

    @property
    def n_local(self) -> int:
        return self.hi - self.lo


def plan_clips(n_clips: int, world: int, rank: int) -> ClipPlan:
    """Contiguous blocks by clip index: rank r holds rows [r*n//world, (r+1)*n//world) of the [n,1,128,188] feature
    tensor, so the shards concatenated in rank order ARE the data set's order (10 000 clips over 8 ranks: 1 250 each).
    Clips are independent and of one size (192 000 bytes, src/review/action.go:212-215): no exchange is needed."""
    if n_clips < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError("plan_clips: bad arguments")
    edges = [r * n_clips // world for r in range(world + 1)]
    return ClipPlan(edges[rank], edges[rank + 1], [edges[r + 1] - edges[r] for r in range(world)])


def gather_features(feat, counts: Sequence[int], dist=None, dst: int = 0):
    """Optional: the per-rank feature rows [n_local,128,188] f32 gathered on ``dst`` (padded to max(counts); 962 MB for
    config 5, which is why bench.py leaves it off by default: a trainer reads each shard on the GPU that made it).
    Returns the [sum(counts),128,188] tensor on dst, None elsewhere."""
    import torch
    if dist is None:
        import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return feat
    if len(counts) != world or feat.shape[0] != counts[rank]:
        raise ValueError("gather_features: counts must list every rank's clips")
    cap = int(max(counts))
    pad = feat if feat.shape[0] == cap else torch.cat([feat, feat.new_zeros((cap - feat.shape[0],) + tuple(feat.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: counts[r]] for r in range(world)])


# ----------------------------------------------------------------------------------------------
# tagged clips on disk: tags/<tag>/<file>:<n>.dat = 2 s of raw s16le 48 kHz
# (src/review/action.go:190-224 writes them, ai/model.py:155-160 / ai/train.py:128-135 read them)
# ----------------------------------------------------------------------------------------------
def dat_name(recording: str, frame: int) -> str:
    """src/review/action.go:202-204: '<recording file name>:<first second>.dat'."""
    import os
    return f"{os.path.basename(recording)}:{frame}.dat"


def write_dat(path: str, pcm_2s) -> None:
    b = np.ascontiguousarray(pcm_2s).view(np.uint8).reshape(-1)
    if b.size != SAMPLE_SIZE:
        raise ValueError(f"a tagged clip is {SAMPLE_SIZE} bytes (two 1-s segments), got {b.size}")
    b.tofile(path)


def read_dat(path: str) -> np.ndarray:
    """Raw bytes of a tagged clip (any length: Prepare pads / truncates, model.go:87-96)."""
    return np.fromfile(path, np.uint8)


# ----------------------------------------------------------------------------------------------
# either side of the path: RIFF header of the live stream, one-minute hit buckets (host only)
# ----------------------------------------------------------------------------------------------
def wav_data_offset(buf) -> dict:
    """apr_wav_data_offset: where the PCM payload of a RIFF/WAVE stream starts (0 = raw PCM) + fmt fields.
    The reference does not strip the header of ffmpeg's `-f wav -` stream (src/ffmpeg/ffmpeg.go:111-114,
    src/daemon/daemon.go:151-154)."""
    import ctypes as C
    from . import _capi
    b = np.ascontiguousarray(np.frombuffer(buf, np.uint8) if not isinstance(buf, np.ndarray) else buf).view(np.uint8).reshape(-1)
    sr, ch, bits = C.c_uint32(), C.c_uint16(), C.c_uint16()
    off = _capi.lib().apr_wav_data_offset(b.ctypes.data, b.size, C.byref(sr), C.byref(ch), C.byref(bits))
    return dict(offset=int(off), sample_rate=sr.value, channels=ch.value, bits_per_sample=bits.value)


def bucket_hits(label, match, n_classes: int, windows_per_bucket: int = 60) -> np.ndarray:
    """apr_bucket_hits: [n_buckets, n_classes] u8, 1 iff the class matched at least once in that minute
    (docs/usage/review.md:48-53: '1 clap or 999 claps within 1 minute is logged as one hit')."""
    from . import _capi
    lab = np.ascontiguousarray(label, np.int32)
    mat = np.ascontiguousarray(match, np.uint8)
    if lab.shape != mat.shape or lab.ndim != 1:
        raise ValueError("label and match must be 1-D and of equal length")
    nb = (lab.size + windows_per_bucket - 1) // windows_per_bucket
    hits = np.zeros((nb, n_classes), np.uint8)
    rc = _capi.lib().apr_bucket_hits(lab.ctypes.data, mat.ctypes.data, lab.size, n_classes, windows_per_bucket, hits.ctypes.data)
    if rc != 0:
        raise _capi.AprError(rc, "apr_bucket_hits: bad argument")
    return hits


def resample_taps(up: int = 3, taps_per_phase: int = 32, cutoff: float = 0.97, beta: float = 9.0) -> np.ndarray:
    """apr_resample_taps: [up, taps_per_phase] f64 polyphase low-pass (Kaiser-windowed sinc, each phase sums to 1).
    The reference leaves this step to ffmpeg's ``-ar 48000`` (src/ffmpeg/ffmpeg.go:76,113-114)."""
    from . import _capi
    taps = np.empty((up, taps_per_phase), np.float64)
    rc = _capi.lib().apr_resample_taps(up, taps_per_phase, cutoff, beta, taps.ctypes.data)
    if rc != 0:
        raise _capi.AprError(rc, "apr_resample_taps: bad argument")
    return taps


# ----------------------------------------------------------------------------------------------
# multi-model fan-out: one spectrogram, every configured model
# (src/daemon/daemon.go:93-99,132-136 and ai/inspect.py:133-159)
# ----------------------------------------------------------------------------------------------
class MultiModelSweep:
    """``models`` = {name: (Engine, labels)}; ``frontend`` = a front-end Engine.  The spectrograms
    are computed ONCE per recording and handed to every model, as the reference does."""

    def __init__(self, frontend, models: Dict[str, tuple], trust: float = 0.5):
        self.frontend = frontend
        self.models = models
        self.trust = trust

    def score_recording(self, pcm, stride: int = BYTES_PER_SECOND, flavour: int = 0) -> Dict[str, dict]:
        """flavour 0 (APR_FLAVOUR_GO): the daemon's Prepare (model.go:86); 1 (APR_FLAVOUR_LIBROSA): ai.inspect's
        audio_to_spectrogram (ai/inspect.py:125 -> ai/model.py:184)."""
        b = np.ascontiguousarray(pcm).view(np.uint8).reshape(-1)
        n = windows_in(b.size, stride)
        spec = self.frontend.prepare_batch(b, n, stride, flavour)
        out = {}
        for name, (eng, labels) in self.models.items():
            empty = labels.index("empty") if "empty" in labels else -1
            out[name] = eng.classify(spec, self.trust, empty)
        return out

    def inspect_lines(self, file_name: str, pcm) -> List[str]:
        """What ``python -m ai.inspect -i <file>`` prints (ai/inspect.py:73-84): window-major,
        model-minor, 1-based window index, 'empty' skipped, no threshold; spectrograms in the LIBROSA flavour, which
        is what ai.inspect computes (ai/inspect.py:125).  One difference is kept: a ragged last slice (recording not
        a whole number of seconds) is zero-padded to 2 s here, where ai.inspect hands librosa the short slice."""
        res = self.score_recording(pcm, flavour=1)
        n = len(next(iter(res.values()))["label"]) if res else 0
        lines = []
        for k in range(n):
            for name, (_, labels) in self.models.items():
                lab = int(res[name]["label"][k])
                if lab < 0 or labels[lab] == "empty":
                    continue
                lines.append(f"Matched {name}/{labels[lab]} in {file_name} @{k + 1} sec")
        return lines

    def daemon_lines(self, pcm) -> List[str]:
        """The [INFO] lines the monitor daemon would log for this stream (daemon.go:203-204)."""
        res = self.score_recording(pcm)
        lines = []
        n = len(next(iter(res.values()))["label"]) if res else 0
        for k in range(n):
            for name, (_, labels) in self.models.items():
                r = res[name]
                if r["match"][k]:
                    lab = int(r["label"][k])
                    lines.append(daemon_match_line(name, labels[lab], float(r["probs"][k, lab])))
        return lines


# ----------------------------------------------------------------------------------------------
# live mode (BASELINE config 4): many concurrent streams, one 1-s segment each per tick
# (src/daemon/daemon.go:104-125: window = previous segment || new segment)
# ----------------------------------------------------------------------------------------------
class StreamScorer:
    """Live mode (daemon.go:104-125) for n_streams streams over apr_stream_tick: the engine keeps every stream's
    previous 1-s segment on the GPU; ``tick(segments)`` uploads the new ones and scores the n_streams windows
    [previous || new] in one pass.  The first tick only primes the buffers (daemon.go:115-119) and returns None.
    ``buffer`` is the engine's pinned staging as a [n_streams, 96000] uint8 array: a reader that deposits the
    segments there and calls ``tick()`` without an argument skips the host copy."""

    def __init__(self, engine, n_streams: int, trust: float = 0.5, empty_class: int = -1):
        self.engine = engine
        self.n = n_streams
        self.trust = trust
        self.empty = empty_class
        self.buffer = engine.stream_buffer(n_streams)
        engine.stream_reset()

    def tick(self, segments=None) -> Optional[dict]:
        if segments is None:
            seg = self.buffer
        else:
            seg = np.ascontiguousarray(segments).view(np.uint8).reshape(self.n, BYTES_PER_SECOND)
        return self.engine.stream_tick(seg, self.n, self.trust, self.empty)
