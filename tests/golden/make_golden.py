"""Generate the golden CNN vectors from the REFERENCE's own NoiseDetector class.

Run in the build container only (needs /root/reference); the outputs are small
.npy files committed next to this script.  Nothing here travels as code: the
fixtures are inputs (by seed / by the reference's own test clips) and expected
outputs.

    python tests/golden/make_golden.py

How the reference is imported: ``ai/model.py:10`` does ``import librosa`` and
calls ``librosa.filters.mel`` once at import (ai/model.py:28-34).  librosa is
not installed here and may not be fetched, so an in-memory placeholder module is
registered for that one import.  NOTHING of librosa's arithmetic is used or
emulated by it (the placeholder returns zeros); only the pure-torch
``NoiseDetector`` class (ai/model.py:40-117) is exercised.  The librosa-flavour
front end therefore stays "parity unpinned".
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from apr_amd import weights as W  # noqa: E402
import oracle  # noqa: E402


def import_reference_model():
    stub = types.ModuleType("librosa")
    stub.filters = types.SimpleNamespace(mel=lambda **kw: np.zeros((kw["n_mels"], kw["n_fft"] // 2 + 1), np.float32))
    sys.modules["librosa"] = stub
    sys.path.insert(0, REF)
    import ai.model as ref_model
    return ref_model


def main():
    ref = import_reference_model()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    n_classes = 4  # src/model/test_model.labels

    # ---- inputs: the four reference test clips through the oracle go front end,
    #      plus two synthetic spectrograms (uniform noise, and a smooth ramp).
    clips = ["test_empty", "test_smalldog", "test_bigdog", "test_combo"]
    specs = []
    for c in clips:
        src = os.path.join(REF, "src/model", c + ".dat")
        pcm = np.fromfile(src, np.uint8)
        dst = os.path.join(HERE, c + ".dat")
        if not os.path.exists(dst):
            pcm.tofile(dst)  # data fixture held by the reference's own tests (model_test.go:54-71)
        specs.append(oracle.prepare_go(pcm)[0])
    specs.append(W.uniform01(1234, 128 * 188).reshape(128, 188))
    r, c = np.meshgrid(np.arange(128), np.arange(188), indexing="ij")
    specs.append(((np.sin(r / 9.0) * np.cos(c / 17.0) + 1.0) / 2.0).astype(np.float32))
    specs = np.stack(specs).astype(np.float32)
    np.save(os.path.join(HERE, "cnn_input_specs.npy"), specs)

    for init, seed in (("he", 0), ("torch", 1)):
        flat = W.random_init(n_classes, seed, init)
        net = ref.NoiseDetector(num_classes=n_classes)
        sd = {k: torch.from_numpy(v.copy()) for k, v in W.to_state_dict(flat, n_classes).items()}
        missing = net.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys
        assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
        net.eval()
        with torch.no_grad():
            x = torch.from_numpy(specs).unsqueeze(1)
            feat = net.features(x)
            pooled = torch.amax(feat, dim=(2, 3))
            logits = net(x)
        np.save(os.path.join(HERE, f"cnn_logits_{init}_seed{seed}.npy"), logits.numpy())
        np.save(os.path.join(HERE, f"cnn_pooled_{init}_seed{seed}.npy"), pooled.numpy())
        print(init, seed, "logits\n", logits.numpy())

    # A1 known answers from the reference's normalize_audio (ai/model.py:163-168)
    pcm = np.array([-32768, -1, 0, 1, 32767, 12345], dtype="<i2").tobytes()
    np.save(os.path.join(HERE, "normalize_audio.npy"), ref.normalize_audio(pcm))
    print("normalize", ref.normalize_audio(pcm))


if __name__ == "__main__":
    main()
