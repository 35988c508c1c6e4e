"""Parity soak (GPU box): the shipped engine against the CPU oracle on MANY windows -- more than bench.py's 599 + 59 sample.

    python tools/parity_soak.py [--recordings 10] [--seconds 600] [--max-batch 4096] > profiles/r03_parity_soak.json

For every recording (synthetic, bench.py's generator, seeds apart from bench.py's): spectrogram max-abs error (go flavour), logit and
probability max-abs error, label / match mismatches on ALL windows and how many of them sit on near-ties (top-2 logit margin <= 4e-4) or
within 1e-3 of the trust threshold.  The oracle is test infrastructure: it is the checker here, nothing more.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import oracle                                   # noqa: E402
from apr_amd import model, weights as W         # noqa: E402

NEAR_TIE, TRUST, C, EMPTY = 4e-4, 0.5, 4, 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--recordings", type=int, default=10)
    ap.add_argument("--seconds", type=int, default=600)
    ap.add_argument("--max-batch", type=int, default=4096)
    ap.add_argument("--clips", type=int, default=2000, help="librosa-flavour clips (BASELINE config 5's path) checked against the oracle's restatement")
    a = ap.parse_args()
    flat = W.random_init(C, 0, "he")
    e0 = model.Engine(W.pack_blob(flat, C), 0, 64)
    flat = W.center_logits(flat, C, e0.infer_logits(e0.prepare_batch(model.synth_pcm(0, 0, 65), 64)).mean(axis=0))   # labels vary, as in bench.py
    e0.close()
    eng = model.Engine(W.pack_blob(flat, C), 0, a.max_batch)
    fe = model.Engine(None, 0, 1024)
    tot = dict(windows=0, label_mismatches=0, match_mismatches=0, label_mismatches_on_decided=0, match_mismatches_away_from_ties=0, near_ties=0, near_trust=0)
    worst = dict(spec=0.0, logit=0.0, prob=0.0)
    hist = np.zeros(C, np.int64)
    per = []
    t0 = time.time()
    for r in range(a.recordings):
        pcm = model.synth_pcm(7, 1000 + r, a.seconds)
        n = model.windows_in(pcm.nbytes)
        # the recordings laid end to end would be scored in passes of max_batch; one recording at a time here keeps the oracle's memory small
        pcm_big = np.concatenate([pcm] * 1) if n else pcm
        got = eng.score(pcm_big, n, trust=TRUST, empty_class=EMPTY)
        want = oracle.score(flat, C, pcm, n, trust=TRUST, empty_class=EMPTY, want_spec=True)
        spec = fe.prepare_batch(pcm, n)
        logits = eng.classify(spec, TRUST, EMPTY)["logits"]
        srt = np.sort(want["logits"], axis=1)
        decided = (srt[:, -1] - srt[:, -2]) > NEAR_TIE
        near_trust = np.abs(want["probs"].max(axis=1) - TRUST) < 1e-3
        lab_bad, mat_bad = got["label"] != want["label"], got["match"] != want["match"]
        d = dict(recording=r, windows=int(n), spec_max_abs_err=float(np.abs(spec - want["spec"]).max()), logit_max_abs_err=float(np.abs(logits - want["logits"]).max()),
                 prob_max_abs_err=float(np.abs(got["probs"] - want["probs"]).max()), label_mismatches=int(lab_bad.sum()), match_mismatches=int(mat_bad.sum()),
                 near_ties=int((~decided).sum()))
        per.append(d)
        tot["windows"] += int(n); tot["label_mismatches"] += int(lab_bad.sum()); tot["match_mismatches"] += int(mat_bad.sum())
        tot["label_mismatches_on_decided"] += int(lab_bad[decided].sum()); tot["match_mismatches_away_from_ties"] += int(mat_bad[decided & ~near_trust].sum())
        tot["near_ties"] += int((~decided).sum()); tot["near_trust"] += int(near_trust.sum())
        worst["spec"] = max(worst["spec"], d["spec_max_abs_err"]); worst["logit"] = max(worst["logit"], d["logit_max_abs_err"]); worst["prob"] = max(worst["prob"], d["prob_max_abs_err"])
        hist += np.bincount(want["label"], minlength=C)
        sys.stderr.write(f"recording {r}: {d}\n")
    # ---- BASELINE config 5's path: tagged 2-s clips -> librosa-flavour log-mel (parity UNPINNED against librosa itself, which is absent) ----
    from apr_amd._capi import FLAVOUR_LIBROSA
    lib_worst, lib_n = 0.0, 0
    for c0 in range(0, a.clips, 250):
        nc = min(250, a.clips - c0)
        clips = np.concatenate([model.synth_pcm(9, 5000 + c0 + i, 2) for i in range(nc)]).view(np.uint8)
        got = fe.prepare_batch(clips, nc, 192000, FLAVOUR_LIBROSA)
        want = oracle.prepare_librosa_batch(clips, nc)
        lib_worst = max(lib_worst, float(np.abs(got - want).max()))
        lib_n += nc
        sys.stderr.write(f"librosa flavour: {lib_n} clips, worst {lib_worst:.3e}\n")
    librosa = dict(clips=lib_n, max_abs_err=lib_worst, tolerance=5e-5, note="against the oracle's restatement of librosa >= 0.10 defaults; librosa itself is absent: parity unpinned")
    out = dict(what="GPU engine (shipped forms, max_batch %d) vs CPU oracle, every window of %d synthetic recordings of %d s" % (a.max_batch, a.recordings, a.seconds),
               tolerances=dict(spectrogram=5e-5, logits=2e-4, probabilities=1e-4, near_tie_margin=NEAR_TIE), totals=tot, worst=worst,
               label_histogram=hist.tolist(), librosa_flavour=librosa, seconds=time.time() - t0, per_recording=per)
    print(json.dumps(out))
    ok = tot["label_mismatches_on_decided"] == 0 and tot["match_mismatches_away_from_ties"] == 0 and worst["spec"] <= 5e-5 and worst["logit"] <= 2e-4 and worst["prob"] <= 1e-4 and lib_worst <= 5e-5
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
