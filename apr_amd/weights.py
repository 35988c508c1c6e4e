"""Weight containers for the NoiseDetector CNN (reference: ai/model.py:40-117).

Three on-disk forms are handled here on the host side:

* the flat ``APRW`` blob (this repo's own trivial format, documented below),
* a minimal ONNX file writer (used by tests and tools to emit the graph shape
  ``ai/model.py:143-152`` exports; the *reader* is native C++ in
  ``apr_amd/csrc/onnx_reader.cpp``),
* a deterministic random-init generator (own counter-based PRNG, so the 94 MB
  of weights never have to be committed or shipped).

APRW blob layout (little endian)::

    bytes 0..3   magic  b"APRW"
    u32          version (1)
    u32          n_classes
    u32          reserved (0)
    f32[...]     state_dict order, without num_batches_tracked:
                 for each conv layer l = 0..11:
                     conv.weight [Cout][Cin][3][3], conv.bias [Cout],
                     bn.weight [Cout], bn.bias [Cout],
                     bn.running_mean [Cout], bn.running_var [Cout]
                 classifier.3.weight [C][1024], classifier.3.bias [C]
"""
from __future__ import annotations

import struct

import numpy as np

# ai/model.py:55-103
LAYER_CIN = (1, 32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 1024)
LAYER_COUT = (32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 1024, 1024)
LAYER_STRIDE = (1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2)
FEATURES = 1024
BN_EPS = 1e-5
MAGIC = b"APRW"

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """Counter-based splitmix64: element i is mix(seed + (offset+i+1)*GOLDEN)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float32 uniforms in [0, 1) with 24 random bits (exact in f32)."""
    return ((splitmix64(seed, n, offset) >> np.uint64(40)).astype(np.float32)
            * np.float32(1.0 / (1 << 24)))


def blob_floats(n_classes: int) -> int:
    n = 0
    for cin, cout in zip(LAYER_CIN, LAYER_COUT):
        n += cout * cin * 9 + 5 * cout
    return n + n_classes * FEATURES + n_classes


def state_layout(n_classes: int):
    """[(torch state_dict key, shape)] in blob order (ai/model.py:55-110)."""
    out = []
    for l, (cin, cout) in enumerate(zip(LAYER_CIN, LAYER_COUT)):
        ci, bi = 3 * l, 3 * l + 1
        out.append((f"features.{ci}.weight", (cout, cin, 3, 3)))
        out.append((f"features.{ci}.bias", (cout,)))
        out.append((f"features.{bi}.weight", (cout,)))
        out.append((f"features.{bi}.bias", (cout,)))
        out.append((f"features.{bi}.running_mean", (cout,)))
        out.append((f"features.{bi}.running_var", (cout,)))
    out.append(("classifier.3.weight", (n_classes, FEATURES)))
    out.append(("classifier.3.bias", (n_classes,)))
    return out


def random_init(n_classes: int, seed: int = 0, init: str = "he") -> np.ndarray:
    """Deterministic random-init weights as a flat float32 array (blob payload).

    ``init="torch"`` uses torch's default bounds (kaiming_uniform(a=sqrt(5)) =
    U(+-1/sqrt(fan_in)) for weights and biases; BN gamma=1, beta=0, mean=0, var=1;
    ai/model.py relies on those defaults) drawn from this module's own PRNG.
    ``init="he"`` keeps activations O(1) through all 12 layers
    (U(+-sqrt(6/fan_in))) and randomises the BatchNorm statistics so the BN
    folding path is exercised and labels vary from window to window.
    """
    parts = []
    off = 0
    for key, shape in state_layout(n_classes):
        n = int(np.prod(shape))
        u = uniform01(seed, n, off)
        off += n
        if key.endswith("running_var"):
            v = (0.5 + u) if init == "he" else np.ones(n, np.float32)
        elif key.endswith("running_mean"):
            v = (0.2 * (u - 0.5)) if init == "he" else np.zeros(n, np.float32)
        elif ".weight" in key and len(shape) == 1:      # bn gamma
            v = (0.75 + 0.5 * u) if init == "he" else np.ones(n, np.float32)
        elif ".bias" in key and key.startswith("features") and int(key.split(".")[1]) % 3 == 1:  # bn beta
            v = (0.2 * (u - 0.5)) if init == "he" else np.zeros(n, np.float32)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else None
            if fan_in is None:  # conv / fc bias: fan_in of its weight
                fan_in = prev_fan_in
            else:
                prev_fan_in = fan_in
            if init == "he" and len(shape) > 1:
                bound = np.sqrt(6.0 / fan_in)
            else:
                bound = 1.0 / np.sqrt(fan_in)
            v = (2.0 * u - 1.0) * np.float32(bound)
        parts.append(v.astype(np.float32))
    flat = np.concatenate(parts)
    assert flat.size == blob_floats(n_classes)
    return flat


def center_logits(flat: np.ndarray, n_classes: int, mean_logits) -> np.ndarray:
    """Return a copy with ``mean_logits`` subtracted from the classifier bias (so that a
    random-init net's argmax varies from window to window instead of sitting on one class)."""
    out = np.array(flat, dtype=np.float32, copy=True)
    out[-n_classes:] -= np.asarray(mean_logits, np.float32)
    return out


def pack_blob(flat: np.ndarray, n_classes: int) -> bytes:
    flat = np.ascontiguousarray(flat, dtype="<f4")
    if flat.size != blob_floats(n_classes):
        raise ValueError("weight count does not match n_classes")
    return MAGIC + struct.pack("<III", 1, n_classes, 0) + flat.tobytes()


def unpack_blob(blob: bytes):
    if blob[:4] != MAGIC:
        raise ValueError("not an APRW blob")
    ver, n_classes, _ = struct.unpack("<III", blob[4:16])
    if ver != 1:
        raise ValueError("unsupported APRW version")
    flat = np.frombuffer(blob, dtype="<f4", offset=16)
    if flat.size != blob_floats(n_classes):
        raise ValueError("truncated APRW blob")
    return flat, n_classes


def to_state_dict(flat: np.ndarray, n_classes: int):
    """flat -> {key: ndarray} (feed to torch ``load_state_dict`` after torch.from_numpy)."""
    sd = {}
    off = 0
    for key, shape in state_layout(n_classes):
        n = int(np.prod(shape))
        sd[key] = np.asarray(flat[off:off + n], dtype=np.float32).reshape(shape)
        off += n
    return sd


def from_state_dict(sd, n_classes: int) -> np.ndarray:
    parts = []
    for key, shape in state_layout(n_classes):
        a = np.asarray(sd[key], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {a.shape} != {shape}")
        parts.append(a.reshape(-1))
    return np.concatenate(parts)


# --------------------------------------------------------------------------
# Minimal ONNX (protobuf) writer: emits the graph ai/model.py:143-152 exports
# (opset 20, input "input" [1,1,128,188], output "output" [1,C]); either with
# BatchNorm folded into Conv (what the legacy exporter does in eval mode) or
# with explicit BatchNormalization nodes.
# --------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    out = bytearray()
    n &= 0xFFFFFFFFFFFFFFFF
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wt: int) -> bytes:
    return _varint((field << 3) | wt)


def _ld(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def _vi(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(v)


def _s(field: int, s: str) -> bytes:
    return _ld(field, s.encode())


def _tensor(name: str, arr: np.ndarray, raw: bool = True) -> bytes:
    arr = np.ascontiguousarray(arr, dtype="<f4")
    t = b"".join(_vi(1, d) for d in arr.shape) + _vi(2, 1)  # dims, data_type FLOAT
    if raw:
        t += _s(8, name) + _ld(9, arr.tobytes())            # name, raw_data
    else:
        t += _ld(4, arr.tobytes()) + _s(8, name)            # packed float_data
    return t


def _attr_ints(name: str, vals) -> bytes:
    a = _s(1, name) + b"".join(_vi(8, v) for v in vals) + _vi(20, 7)  # type INTS
    return a


def _attr_int(name: str, v: int) -> bytes:
    return _s(1, name) + _vi(3, v) + _vi(20, 2)  # type INT


def _attr_float(name: str, v: float) -> bytes:
    return _s(1, name) + _key(2, 5) + struct.pack("<f", v) + _vi(20, 1)  # type FLOAT


def _node(op: str, inputs, outputs, attrs=(), name="") -> bytes:
    n = b"".join(_s(1, i) for i in inputs) + b"".join(_s(2, o) for o in outputs)
    if name:
        n += _s(3, name)
    n += _s(4, op)
    n += b"".join(_ld(5, a) for a in attrs)
    return n


def _value_info(name: str, shape) -> bytes:
    dims = b"".join(_ld(1, _vi(1, d)) for d in shape)
    tensor_type = _vi(1, 1) + _ld(2, dims)
    return _s(1, name) + _ld(2, _ld(1, tensor_type))


def write_onnx(flat: np.ndarray, n_classes: int, fold_bn: bool = True, raw: bool = True) -> bytes:
    """Serialise the NoiseDetector graph to ONNX bytes (see module docstring)."""
    sd = to_state_dict(flat, n_classes)
    nodes, inits = [], []
    cur = "input"
    for l in range(12):
        ci, bi = 3 * l, 3 * l + 1
        w = sd[f"features.{ci}.weight"].astype(np.float64)
        b = sd[f"features.{ci}.bias"].astype(np.float64)
        g = sd[f"features.{bi}.weight"].astype(np.float64)
        be = sd[f"features.{bi}.bias"].astype(np.float64)
        mu = sd[f"features.{bi}.running_mean"].astype(np.float64)
        va = sd[f"features.{bi}.running_var"].astype(np.float64)
        s = LAYER_STRIDE[l]
        conv_attrs = [_attr_ints("dilations", [1, 1]), _attr_int("group", 1),
                      _attr_ints("kernel_shape", [3, 3]), _attr_ints("pads", [1, 1, 1, 1]),
                      _attr_ints("strides", [s, s])]
        if fold_bn:
            scale = g / np.sqrt(va + BN_EPS)
            wf = (w * scale[:, None, None, None]).astype(np.float32)
            bf = ((b - mu) * scale + be).astype(np.float32)
            inits += [_tensor(f"conv{l}.w", wf, raw), _tensor(f"conv{l}.b", bf, raw)]
            nodes.append(_node("Conv", [cur, f"conv{l}.w", f"conv{l}.b"], [f"c{l}"], conv_attrs, f"Conv_{l}"))
            cur = f"c{l}"
        else:
            inits += [_tensor(f"conv{l}.w", w, raw), _tensor(f"conv{l}.b", b, raw),
                      _tensor(f"bn{l}.g", g, raw), _tensor(f"bn{l}.b", be, raw),
                      _tensor(f"bn{l}.m", mu, raw), _tensor(f"bn{l}.v", va, raw)]
            nodes.append(_node("Conv", [cur, f"conv{l}.w", f"conv{l}.b"], [f"c{l}"], conv_attrs, f"Conv_{l}"))
            nodes.append(_node("BatchNormalization",
                               [f"c{l}", f"bn{l}.g", f"bn{l}.b", f"bn{l}.m", f"bn{l}.v"], [f"n{l}"],
                               [_attr_float("epsilon", BN_EPS), _attr_float("momentum", 0.9)], f"BN_{l}"))
            cur = f"n{l}"
        nodes.append(_node("Relu", [cur], [f"r{l}"], (), f"Relu_{l}"))
        cur = f"r{l}"
    nodes.append(_node("GlobalMaxPool", [cur], ["pool"], (), "Pool"))
    nodes.append(_node("Flatten", ["pool"], ["flat"], [_attr_int("axis", 1)], "Flatten"))
    inits += [_tensor("fc.w", sd["classifier.3.weight"], raw), _tensor("fc.b", sd["classifier.3.bias"], raw)]
    nodes.append(_node("Gemm", ["flat", "fc.w", "fc.b"], ["output"],
                       [_attr_float("alpha", 1.0), _attr_float("beta", 1.0), _attr_int("transB", 1)], "Gemm"))
    graph = b"".join(_ld(1, n) for n in nodes) + _s(2, "main_graph")
    graph += b"".join(_ld(5, t) for t in inits)
    graph += _ld(11, _value_info("input", [1, 1, 128, 188]))
    graph += _ld(12, _value_info("output", [1, n_classes]))
    model = _vi(1, 8) + _s(2, "pytorch") + _s(3, "2.10.0") + _ld(7, graph)
    model += _ld(8, _s(1, "") + _vi(2, 20))  # opset_import {domain "", version 20}
    return model
