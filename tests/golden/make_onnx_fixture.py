"""Export the REFERENCE's own NoiseDetector with the REFERENCE's own export call and commit what the file
looks like -- not the file.

Run in the build container only (needs /root/reference):

    python tests/golden/make_onnx_fixture.py [--keep /tmp/real.onnx]

What it does: imports ``ai.model`` exactly as make_golden.py does (a placeholder module satisfies the one
``import librosa``; none of librosa's arithmetic is involved), loads the deterministic he-seed-0 weights
(apr_amd/weights.py) into ``NoiseDetector(4)``, and calls ``torch.onnx.export`` with the arguments of
``ai/model.py:149-152`` (input_names ['input'], output_names ['output'], opset_version 20, dynamo=False).

The absent ``onnx`` package: the legacy exporter's C++ serialiser needs nothing from it.  After the bytes
exist, one Python hook, ``onnx_proto_utils._add_onnxscript_fn``, imports ``onnx`` to splice in onnxscript
function definitions; for a model that uses none (this one) it returns its input unchanged.  That hook is the
only user of the package on this path, so it is replaced by the identity here.

Output: ``tests/golden/onnx_real_export.json`` (a few KB of DATA): producer / ir / opset fields, the node list
with op types, names, inputs, outputs and attributes, initialiser names / dims / dtype / storage field /
byte count / sha256, graph inputs and outputs, serialisation order, the sha256 of the file with the payloads
cut out, and how this repo's writer compares with the real payloads on this host.  The 94 MB file itself is
not committed and never travels.
"""
import hashlib
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from apr_amd import weights as W  # noqa: E402


def import_reference_model():
    stub = types.ModuleType("librosa")
    stub.filters = types.SimpleNamespace(mel=lambda **kw: np.zeros((kw["n_mels"], kw["n_fft"] // 2 + 1), np.float32))
    sys.modules["librosa"] = stub
    sys.path.insert(0, REF)
    import ai.model as ref_model
    return ref_model


def export_reference(n_classes=4, seed=0, init="he", path=None):
    """The real exporter's bytes for the reference's model with this repo's deterministic weights."""
    ref = import_reference_model()
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    flat = W.random_init(n_classes, seed, init)
    net = ref.NoiseDetector(num_classes=n_classes)
    sd = {k: torch.from_numpy(v.copy()) for k, v in W.to_state_dict(flat, n_classes).items()}
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    net.eval()                                                   # ai/model.py:139 (load() puts the model in eval mode)
    tmp = path or os.path.join(tempfile.mkdtemp(prefix="apr_onnx_"), "real.onnx")
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(net, torch.randn(1, 1, 128, 188), tmp,                 # ai/model.py:149-152
                          input_names=["input"], output_names=["output"], opset_version=20,
                          dynamo=False, verbose=False)
    data = open(tmp, "rb").read()
    if path is None:
        os.remove(tmp)
    return data, flat


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    data, flat = export_reference(path=keep)
    st = W.onnx_structure(data)
    st["how"] = {"script": "tests/golden/make_onnx_fixture.py", "torch": torch.__version__,
                 "export_call": "ai/model.py:149-152 (opset_version=20, dynamo=False, input 1x1x128x188)",
                 "weights": "apr_amd.weights.random_init(4, 0, 'he')", "file_sha256": hashlib.sha256(data).hexdigest()}
    # first floats of every payload: lets a CPU test anywhere check the fold arithmetic on real numbers
    real = W.onnx_initializers(data)
    mine = W.onnx_initializers(W.write_onnx(flat, 4))
    cmp = {}
    for name, a in real.items():
        b = mine[name]
        denom = np.maximum(np.abs(a), 1e-30)
        cmp[name] = {"head": [float(x) for x in a.reshape(-1)[:4]], "bit_mismatches": int((a != b).sum()),
                     "max_abs_diff": float(np.abs(a - b).max()),
                     "max_rel_diff_of_l2": float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(a.astype(np.float64)))}
        del denom
    st["writer_vs_real_on_this_host"] = cmp
    # what the product's reader says about the real bytes
    from apr_amd import model
    st["apr_model_info"] = list(model.model_info(data))
    out = os.path.join(HERE, "onnx_real_export.json")
    with open(out, "w") as f:
        json.dump(st, f, indent=1, sort_keys=True)
        f.write("\n")
    print(out, os.path.getsize(out), "bytes; file", len(data), "bytes; apr_model_info", st["apr_model_info"])
    print("payloads identical to the writer's:", sum(v["bit_mismatches"] == 0 for v in cmp.values()), "of", len(cmp),
          "; worst max_abs_diff", max(v["max_abs_diff"] for v in cmp.values()))


if __name__ == "__main__":
    main()
