"""GPU: the BASELINE configurations at their stated sizes (config 4: 256 live streams; config 5: 10 000 clips), the
device-list sweep behind the C ABI, and the host-side caches -- size-independent properties plus oracle samples.
Tolerances as in test_gpu_parity.py."""
import time

import numpy as np
import pytest

import oracle
from apr_amd import model, sweep, weights as W
from apr_amd._capi import AprError, FLAVOUR_LIBROSA
from conftest import LABELS

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4
EMPTY = LABELS.index("empty")


def _decision_properties(r, trust, empty):
    probs, label, match = r["probs"], r["label"], r["match"]
    n = len(label)
    assert np.isfinite(probs).all() and probs.min() >= 0.0 and np.abs(probs.sum(axis=1) - 1.0).max() < 1e-12
    assert np.array_equal(label, probs.argmax(axis=1))                                   # daemon.go:193-198
    conf = probs[np.arange(n), label]
    assert np.array_equal(match.astype(bool), (label != empty) & (conf > trust))         # daemon.go:203


def test_sweep_files_multi_device_list(he_weights):
    # SURVEY 8(b) "apr_sweep_* taking a list of (buffer,len) per file and a device list".  One GPU on this box: the
    # list [0] must reproduce apr_sweep_files bit for bit, and two engines on device 0 exercise the per-device threads,
    # the round-robin shard and the file-ordered rows.
    blob = W.pack_blob(he_weights, 4)
    a, b = model.Engine(blob, 0, 16), model.Engine(blob, 0, 24)
    secs = [5.0, 1.0, 0.5, 3.3, 40.0, 0.0, 2.0, 2.01, 7.0]
    files = [model.synth_pcm(3, i, int(np.ceil(s)) if s else 1)[: int(s * 48000)] for i, s in enumerate(secs)]
    want = a.sweep_files(files, trust=0.4, empty_class=EMPTY)
    for engines in ([a], [a, b], [b, a]):
        got = model.Engine.sweep_files_multi(engines, files, trust=0.4, empty_class=EMPTY)
        for k in ("probs", "label", "match", "first_window"):
            assert np.array_equal(got[k], want[k]), (len(engines), k)
    dev, first, per = model.sweep_plan([f.nbytes for f in files], 2)
    assert np.array_equal(first, want["first_window"]) and per.sum() == want["label"].size
    assert model.Engine.sweep_files_multi([a, b], [], trust=0.4)["label"].shape == (0,)
    with pytest.raises(AprError) as ei:
        model.Engine.sweep_files_multi([a, a], files)
    assert "listed twice" in str(ei.value)
    c = model.Engine(W.pack_blob(W.random_init(3, 5, "he"), 3), 0, 8)
    with pytest.raises(AprError) as ei:
        model.Engine.sweep_files_multi([a, c], files)
    assert "different class counts" in str(ei.value)
    fe = model.Engine(None, 0, 8)
    with pytest.raises(AprError) as ei:
        model.Engine.sweep_files_multi([a, fe], files)
    assert ei.value.code == -5
    # the stride may change between calls on one engine (the staging is re-sized): packed 2-s clips, then the 1-s hop again
    clips = [np.concatenate([files[4][: 96000 * k], files[4][:96000]]) for k in (3, 2)]
    s2 = a.sweep_files(clips, stride=192000, trust=0.4, empty_class=EMPTY)
    one = a.score(clips[0], stride=192000, trust=0.4, empty_class=EMPTY)
    assert np.array_equal(s2["probs"][: one["label"].size], one["probs"])
    again = a.sweep_files(files, trust=0.4, empty_class=EMPTY)
    assert np.array_equal(again["probs"], want["probs"])
    for e in (a, b, c, fe):
        e.close()


def test_inspect_lines_use_the_librosa_front_end(he_weights):
    # ai/inspect.py:125 computes its spectrograms with ai.model.audio_to_spectrogram (librosa flavour), not with the Go Prepare
    fe = model.Engine(None, 0, 16)
    eng = model.Engine(W.pack_blob(he_weights, 4), 0, 16)
    mm = sweep.MultiModelSweep(fe, {"dogs": (eng, LABELS)})
    pcm = model.synth_pcm(9, 2, 8)
    n = model.windows_in(pcm.nbytes)
    spec = oracle.prepare_librosa_batch(np.concatenate([pcm[k * 48000:(k + 2) * 48000] for k in range(n)]), n)
    logits = oracle.cnn_forward(he_weights, 4, spec)
    # compared on the DECIDED windows (top-2 margin above twice the logit tolerance): a near-tie excuses its own window only,
    # never the whole comparison
    want, decided = {}, []
    for k in range(n):
        srt = np.sort(logits[k])
        if (srt[-1] - srt[-2]) > 2 * LOGIT_TOL:
            decided.append(k + 1)
        lab = LABELS[int(np.argmax(logits[k]))]
        want[k + 1] = f"Matched dogs/{lab} in rec.mkv @{k + 1} sec" if lab != "empty" else None
    got = mm.inspect_lines("rec.mkv", pcm)
    assert len(decided) >= n - 1
    got_by_sec = {int(ln.rsplit("@", 1)[1].split()[0]): ln for ln in got}
    assert len(got_by_sec) == len(got) and got == [got_by_sec[k] for k in sorted(got_by_sec)]     # one line per window at most, window order
    for k in decided:
        assert got_by_sec.get(k) == want[k], k
    res = mm.score_recording(pcm, flavour=FLAVOUR_LIBROSA)["dogs"]
    assert np.abs(res["logits"] - logits).max() <= LOGIT_TOL
    go = mm.score_recording(pcm)["dogs"]
    assert not np.array_equal(go["probs"], res["probs"])                  # the two front ends really differ
    fe.close(); eng.close()


def test_config4_256_streams_against_oracle(he_weights):
    # BASELINE config 4 at its stated size: 256 concurrent streams, one 1-s segment each per tick, window = previous || new
    # (src/daemon/daemon.go:104-125).  Properties on every row, oracle on a 24-window sample, the tick inside its 1-s budget.
    n_streams, secs, trust = 256, 6, 0.5
    e0 = model.Engine(W.pack_blob(he_weights, 4), 0, 64)
    flat = W.center_logits(he_weights, 4, e0.infer_logits(e0.prepare_batch(model.synth_pcm(0, 0, 65), 64)).mean(axis=0))
    e0.close()
    eng = model.Engine(W.pack_blob(flat, 4), 0, n_streams)
    src = np.stack([model.synth_pcm(4, s, secs).view(np.uint8).reshape(secs, 96000) for s in range(n_streams)])
    sc = sweep.StreamScorer(eng, n_streams, trust=trust, empty_class=EMPTY)
    assert sc.tick(np.ascontiguousarray(src[:, 0])) is None
    ticks, lat = [], []
    for t in range(1, secs):
        seg = np.ascontiguousarray(src[:, t])
        t0 = time.perf_counter()
        r = sc.tick(seg)
        lat.append(time.perf_counter() - t0)
        _decision_properties(r, trust, EMPTY)
        ticks.append(r)
    assert max(lat[1:]) < 1.0, lat                                       # real time: a tick per second (first timed tick warms up)
    assert len(np.unique(np.concatenate([r["label"] for r in ticks]))) == 4
    # tick path == batch path, bit for bit, on a few whole streams
    for s in (0, 100, 255):
        ref = eng.score(src[s].reshape(-1), trust=trust, empty_class=EMPTY)
        for t in range(secs - 1):
            assert np.array_equal(ticks[t]["probs"][s], ref["probs"][t]) and ticks[t]["match"][s] == ref["match"][t]
    # oracle sample, spread over streams and ticks; mismatches are counted on ALL sampled windows and must sit on near-ties
    rng = np.random.default_rng(4)
    bad = 0
    for s, t in zip(rng.integers(0, n_streams, 24), rng.integers(1, secs, 24)):
        win = np.concatenate([src[s, t - 1], src[s, t]])
        ref = oracle.score(flat, 4, win, 1, trust=trust, empty_class=EMPTY)
        got = ticks[t - 1]
        assert np.abs(ref["probs"][0] - got["probs"][s]).max() <= 1e-4
        srt = np.sort(ref["logits"][0])
        if ref["label"][0] != got["label"][s] or ref["match"][0] != got["match"][s]:
            bad += 1
            assert srt[-1] - srt[-2] <= 2 * LOGIT_TOL or abs(ref["probs"][0].max() - trust) < 1e-3
    assert bad <= 1
    eng.close()


def test_config5_10000_clips_librosa_features():
    # BASELINE config 5 at its stated size: 10 000 tagged .dat clips (192 000 B each) -> [10000,128,188] f32 with the librosa
    # front end (ai/train.py:36-52), clips resident in HBM.  PARITY UNPINNED against librosa itself (absent here).
    import torch
    dev = torch.device("cuda", 0)
    n, uniq = 10000, 400
    host = np.concatenate([model.synth_pcm(5, i, 2) for i in range(uniq)]).view(np.uint8)
    packed = np.tile(host, n // uniq)
    assert packed.size == n * 192000
    d_pcm = torch.from_numpy(packed).to(dev)
    d_out = torch.empty((n, 128, 188), dtype=torch.float32, device=dev)
    fe = model.Engine(None, 0, 1024)                                   # 10 passes, the last one ragged (10000 = 9 x 1024 + 784)
    fe.prepare_batch_device(d_pcm, d_pcm.numel(), n, 192000, d_out, FLAVOUR_LIBROSA)
    fe.sync()
    assert bool(torch.isfinite(d_out).all()) and float(d_out.min()) >= 0.0 and float(d_out.max()) <= 1.0
    assert bool((d_out.reshape(n, -1).max(dim=1).values == 1.0).all())     # the dB reference is each clip's own maximum
    rep = d_out.reshape(n // uniq, uniq, 128, 188)
    assert bool((rep == rep[0:1]).all())                                     # position in the batch / pass is bit-invisible
    ks = (np.arange(16) * 631) % n                                           # spread over the passes, 16 different clips
    got = d_out[torch.from_numpy(ks).to(dev)].cpu().numpy()
    want = oracle.prepare_librosa_batch(np.concatenate([packed[k * 192000:(k + 1) * 192000] for k in ks]), len(ks))
    assert np.abs(got - want).max() <= 5e-5
    assert len({got[i].tobytes() for i in range(len(ks))}) == 16            # distinct clips give distinct features
    fe.close()


def test_infer_engine_cache_is_keyed_by_content(he_weights, clips):
    # a model re-loaded after retraining (same length, possibly the same address) must not hit the old engine
    model.Unload()
    a = W.pack_blob(he_weights, 4)
    w2 = he_weights.copy()
    w2[-4:] += np.float32(3.0) * np.arange(4, dtype=np.float32)           # another classifier bias, same length
    b = W.pack_blob(w2, 4)
    assert len(a) == len(b)
    ma, ma2, mb = model.OnnxModel(a, LABELS), model.OnnxModel(bytes(bytearray(a)), LABELS), model.OnnxModel(b, LABELS)
    spec = model.Prepare(clips["test_bigdog"].tobytes())
    ra, ra2, rb = model.Infer(ma, spec), model.Infer(ma2, spec), model.Infer(mb, spec)
    assert ra == ra2 and ra != rb
    assert model.engine_for(ma, 1) is model.engine_for(ma2, 1)             # same content, other object: one engine
    assert model.engine_for(ma, 1) is not model.engine_for(mb, 1)
    assert model.engine_for(ma, 1) is not model.engine_for(ma, 8)          # max_batch is part of the key
    lb = oracle.softmax(oracle.cnn_forward(w2, 4, spec.reshape(1, 128, 188))[0])
    assert abs(rb[LABELS[3]] - lb[3]) <= 1e-4
    assert model.Unload(mb) == 1 and model.Unload(ma) == 2 and model.Unload() >= 1


def test_bucket_hits_on_the_gpu_equals_the_host_form(he_weights):
    # one hit per class per minute (docs/usage/review.md:48-53) over rows that are still in HBM
    import torch
    dev = torch.device("cuda", 0)
    fe = model.Engine(None, 0, 8)
    rng = np.random.default_rng(8)
    for n, C, wpb in ((0, 4, 60), (1, 4, 60), (59, 4, 60), (60, 4, 60), (61, 4, 60), (35999, 4, 60), (100000, 64, 7), (5000, 1, 600)):
        label = rng.integers(-1, C + 1, n).astype(np.int32)           # -1 (no label) and C (out of range) must be ignored
        match = (rng.random(n) < 0.05).astype(np.uint8)
        want = sweep.bucket_hits(np.where(label < C, label, -1), match, C, wpb)
        nb = (n + wpb - 1) // wpb
        d_hits = torch.full((max(nb, 1), C), 7, dtype=torch.uint8, device=dev)
        fe.bucket_hits_device(torch.from_numpy(label).to(dev), torch.from_numpy(match).to(dev), n, C, d_hits, wpb)
        fe.sync()
        assert np.array_equal(d_hits.cpu().numpy()[:nb], want), (n, C, wpb)
    with pytest.raises(AprError):
        fe.bucket_hits_device(d_hits, d_hits, 10, 65, d_hits)
    # on the output of a real sweep, without the rows leaving the GPU
    eng = model.Engine(W.pack_blob(he_weights, 4), 0, 64)
    pcm = model.synth_pcm(2, 1, 150)
    n = model.windows_in(pcm.nbytes)
    d_pcm = torch.from_numpy(pcm.view(np.uint8).copy()).to(dev)
    d_label = torch.empty(n, dtype=torch.int32, device=dev)
    d_match = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.score_device(d_pcm, d_pcm.numel(), n, 96000, 0.26, -1, None, d_label, d_match)   # no "empty" class: every window matches some class
    d_hits = torch.empty(((n + 59) // 60, 4), dtype=torch.uint8, device=dev)
    eng.bucket_hits_device(d_label, d_match, n, 4, d_hits)
    eng.sync()
    assert np.array_equal(d_hits.cpu().numpy(), sweep.bucket_hits(d_label.cpu().numpy(), d_match.cpu().numpy(), 4))
    assert d_match.sum().item() > 0
    fe.close(); eng.close()


def test_device_resident_sweep_gather_and_hits(he_weights):
    # The device-resident end of a one-process multi-GPU sweep, as a Go host would drive it through the C ABI alone: PCM uploaded into
    # memory from apr_device_alloc, apr_score_device, apr_gather_rows_device (RCCL communicator over the device list; this box has one GPU,
    # so the root's own rows are what moves), apr_bucket_hits_device on the gathered rows, and only the hits downloaded.
    eng = model.Engine(W.pack_blob(he_weights, 4), 0, 64)
    pcm = model.synth_pcm(11, 3, 150).view(np.uint8)
    n = model.windows_in(pcm.size)
    want = eng.score(pcm, n, trust=0.3, empty_class=EMPTY)
    d_pcm = eng.device_alloc(pcm.size)
    d_probs, d_label, d_match = eng.device_alloc(n * 4 * 8), eng.device_alloc(n * 4), eng.device_alloc(n)
    g_probs, g_label, g_match = eng.device_alloc(n * 4 * 8), eng.device_alloc(n * 4), eng.device_alloc(n)
    nb = (n + 59) // 60
    d_hits = eng.device_alloc(nb * 4)
    assert d_pcm and d_probs and g_match and d_hits
    eng.device_upload(d_pcm, pcm)
    eng.score_device(d_pcm, pcm.size, n, 96000, 0.3, EMPTY, d_probs, d_label, d_match)
    model.Engine.gather_rows_device([eng], [n], 4, [d_probs], [d_label], [d_match], g_probs, g_label, g_match, root=0)
    probs = eng.device_download(g_probs, np.empty((n, 4), np.float64))
    label = eng.device_download(g_label, np.empty(n, np.int32))
    match = eng.device_download(g_match, np.empty(n, np.uint8))
    assert np.array_equal(probs, want["probs"]) and np.array_equal(label, want["label"]) and np.array_equal(match, want["match"])
    eng.bucket_hits_device(g_label, g_match, n, 4, d_hits)
    hits = eng.device_download(d_hits, np.empty((nb, 4), np.uint8))
    assert np.array_equal(hits, sweep.bucket_hits(want["label"], want["match"], 4)) and hits.any()
    # only labels wanted: the other tensors stay NULL; a second call reuses the communicator
    model.Engine.gather_rows_device([eng], [n], 4, None, [d_label], None, None, g_label, None)
    # an output without its inputs, and a root that does not exist, are errors
    with pytest.raises(AprError):
        model.Engine.gather_rows_device([eng], [n], 4, None, [d_label], None, g_probs, g_label, None)
    with pytest.raises(AprError):
        model.Engine.gather_rows_device([eng], [n], 4, None, [d_label], None, None, g_label, None, root=1)
    e2 = model.Engine(None, 0, 8)
    with pytest.raises(AprError) as ei:           # two engines on one device: one rank per GPU
        model.Engine.gather_rows_device([eng, e2], [n, 0], 4, None, [d_label, None], None, None, g_label, None)
    assert "one device" in str(ei.value)
    e2.close()
    for p in (d_pcm, d_probs, d_label, d_match, g_probs, g_label, g_match, d_hits):
        eng.device_free(p)
    eng.close()
