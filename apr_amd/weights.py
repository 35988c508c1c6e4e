"""Weight containers for the NoiseDetector CNN (reference: ai/model.py:40-117).

Three on-disk forms are handled here on the host side:

* the flat ``APRW`` blob (this repo's own trivial format, documented below),
* a minimal ONNX file writer (used by tests and tools to emit the graph shape
  ``ai/model.py:143-152`` exports; the *reader* is native C++ in
  ``apr_amd/csrc/model_parse.cpp``),
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


def fold_bn_exporter(w, b, g, be, mu, va):
    """Conv+BatchNorm fold with the arithmetic of the exporter ai/model.py:149-152 calls (the legacy
    TorchScript exporter's eval-mode peephole): everything in float32, ``scale = g / sqrt(var + eps)``,
    ``w' = w * scale``, ``b' = (b - mean) * scale + beta``.  Found by bit comparison with a real export
    (tests/golden/make_onnx_fixture.py): biases and weights agree bit for bit wherever the exporter host's
    vectorised ``sqrt`` is correctly rounded (it is off by one ulp on a few channels of the wide layers)."""
    f32 = np.float32
    w, b, g, be, mu, va = (np.asarray(a, f32) for a in (w, b, g, be, mu, va))
    scale = g / np.sqrt(va + f32(BN_EPS))
    return (w * scale[:, None, None, None]).astype(f32), ((b - mu) * scale + be).astype(f32)


def write_onnx(flat: np.ndarray, n_classes: int, fold_bn: bool = True, raw: bool = True, style: str = "torch") -> bytes:
    """Serialise the NoiseDetector graph to ONNX bytes (see module docstring).

    ``style="torch"`` (default) reproduces, field for field, the file the reference's own export call
    writes (ai/model.py:149-152 with torch 2.10; structure pinned by tests/golden/onnx_real_export.json):
    ir_version 9, node / value / initialiser names of the TorchScript exporter (``onnx::Conv_127`` ...,
    ``/features/features.0/Conv`` ...), BatchNorm folded in float32, the pool written as
    ``MaxPool(kernel_shape [2,3], pads 0, strides [2,3])``, classifier initialisers first, all ``raw_data``.
    ``fold_bn`` and ``raw`` are ignored in this style (the exporter always folds and always writes raw).
    ``style="plain"`` is the repo's earlier own form (short names, ``GlobalMaxPool``, optional explicit
    BatchNormalization nodes, optional packed ``float_data``): kept because the reader accepts those too.
    """
    if style == "plain":
        return _write_onnx_plain(flat, n_classes, fold_bn, raw)
    if style != "torch":
        raise ValueError("style must be 'torch' or 'plain'")
    sd = to_state_dict(flat, n_classes)
    nodes, conv_inits = [], []
    cur = "input"
    h, w_ = 128, 188
    for l in range(12):
        ci, bi = 3 * l, 3 * l + 1
        wf, bf = fold_bn_exporter(sd[f"features.{ci}.weight"], sd[f"features.{ci}.bias"], sd[f"features.{bi}.weight"],
                                  sd[f"features.{bi}.bias"], sd[f"features.{bi}.running_mean"], sd[f"features.{bi}.running_var"])
        s = LAYER_STRIDE[l]
        h, w_ = (h + 2 - 3) // s + 1, (w_ + 2 - 3) // s + 1
        wn, bn = f"onnx::Conv_{127 + 3 * l}", f"onnx::Conv_{128 + 3 * l}"
        conv_inits += [_tensor(wn, wf), _tensor(bn, bf)]
        conv_attrs = [_attr_ints("dilations", [1, 1]), _attr_int("group", 1), _attr_ints("kernel_shape", [3, 3]),
                      _attr_ints("pads", [1, 1, 1, 1]), _attr_ints("strides", [s, s])]
        co = f"/features/features.{ci}/Conv_output_0"
        nodes.append(_node("Conv", [cur, wn, bn], [co], conv_attrs, f"/features/features.{ci}/Conv"))
        cur = f"/features/features.{ci + 2}/Relu_output_0"
        nodes.append(_node("Relu", [co], [cur], (), f"/features/features.{ci + 2}/Relu"))
    # AdaptiveMaxPool2d(1) on a map of known size (2 x 3): MaxPool over the whole map
    po = "/classifier/classifier.0/MaxPool_output_0"
    nodes.append(_node("MaxPool", [cur], [po], [_attr_ints("kernel_shape", [h, w_]), _attr_ints("pads", [0, 0, 0, 0]),
                                                _attr_ints("strides", [h, w_])], "/classifier/classifier.0/MaxPool"))
    fo = "/classifier/classifier.1/Flatten_output_0"
    nodes.append(_node("Flatten", [po], [fo], [_attr_int("axis", 1)], "/classifier/classifier.1/Flatten"))
    nodes.append(_node("Gemm", [fo, "classifier.3.weight", "classifier.3.bias"], ["output"],
                       [_attr_float("alpha", 1.0), _attr_float("beta", 1.0), _attr_int("transB", 1)], "/classifier/classifier.3/Gemm"))
    inits = [_tensor("classifier.3.weight", sd["classifier.3.weight"]), _tensor("classifier.3.bias", sd["classifier.3.bias"])] + conv_inits
    graph = b"".join(_ld(1, n) for n in nodes) + _s(2, "main_graph")
    graph += b"".join(_ld(5, t) for t in inits)
    graph += _ld(11, _value_info("input", [1, 1, 128, 188]))
    graph += _ld(12, _value_info("output", [1, n_classes]))
    return _vi(1, 9) + _s(2, "pytorch") + _s(3, "2.10.0") + _ld(7, graph) + _ld(8, _vi(2, 20))


def _write_onnx_plain(flat: np.ndarray, n_classes: int, fold_bn: bool = True, raw: bool = True) -> bytes:
    """The repo's own earlier form of the graph (not what the exporter writes; see ``write_onnx``)."""
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


# --------------------------------------------------------------------------
# ONNX structure dump: everything about a file except the weight payloads, as
# plain JSON-able data.  tests/golden/make_onnx_fixture.py applies it to a real
# export of the reference's model; the CPU tests apply it to write_onnx()'s
# output and compare.  Generic protobuf walk + the ONNX field numbers
# (onnx.proto: ModelProto, GraphProto, NodeProto, AttributeProto, TensorProto,
# ValueInfoProto).
# --------------------------------------------------------------------------
def _rd_varint(b, i):
    v = 0
    sh = 0
    while True:
        x = b[i]
        i += 1
        v |= (x & 0x7F) << sh
        sh += 7
        if not x & 0x80:
            return v, i


def _fields(b):
    """[(field, wire type, value)]: varints as int, length-delimited as memoryview, fixed as bytes."""
    b = memoryview(b)
    i, out = 0, []
    while i < len(b):
        k, i = _rd_varint(b, i)
        f, wt = k >> 3, k & 7
        if wt == 0:
            v, i = _rd_varint(b, i)
        elif wt == 2:
            n, i = _rd_varint(b, i)
            if i + n > len(b):
                raise ValueError("truncated protobuf")
            v = b[i:i + n]
            i += n
        elif wt == 5:
            v = bytes(b[i:i + 4])
            i += 4
        elif wt == 1:
            v = bytes(b[i:i + 8])
            i += 8
        else:
            raise ValueError(f"wire type {wt}")
        out.append((f, wt, v))
    return out


_ATTR_TYPES = {1: "FLOAT", 2: "INT", 3: "STRING", 4: "TENSOR", 6: "FLOATS", 7: "INTS", 8: "STRINGS"}


def _attr_struct(b):
    name, typ, ints, floats, val = "", 0, [], [], None
    order = []
    for f, wt, v in _fields(b):
        order.append(f)
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = v
        elif f == 4:
            val = bytes(v).decode()
        elif f == 7 and wt == 5:
            floats.append(struct.unpack("<f", v)[0])
        elif f == 8 and wt == 0:
            ints.append(v)
        elif f == 8 and wt == 2:
            j, mv = 0, v
            while j < len(mv):
                x, j = _rd_varint(mv, j)
                ints.append(x)
        elif f == 20:
            typ = v
    if typ == 7:
        val = ints
    elif typ == 6:
        val = floats
    return {"name": name, "type": _ATTR_TYPES.get(typ, typ), "value": val, "field_order": order}


def _value_info_struct(b):
    name, elem, dims = "", None, []
    for f, wt, v in _fields(b):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            elem = v3
                        elif f3 == 2:
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:
                                    d = None
                                    for f5, wt5, v5 in _fields(v4):
                                        d = v5 if wt5 == 0 else bytes(v5).decode()
                                    dims.append(d)
    return {"name": name, "elem_type": elem, "shape": dims}


def onnx_structure(data: bytes) -> dict:
    """Everything in an ONNX file but the weight values: model fields, node list (op, name, inputs, outputs,
    attributes with types and values, in file order), initialisers (name, dims, data type, which storage field,
    byte count, sha256 of the payload), graph inputs / outputs, the order of the fields as serialised, and
    ``skeleton_sha256`` = sha256 of the file with every initialiser payload cut out (its length prefix kept):
    two files with equal skeletons differ in weight bytes only."""
    import hashlib
    data = bytes(data)
    out = {"file_bytes": len(data), "model_field_order": [], "opset_import": [], "graph": None}
    skel = hashlib.sha256()
    mv = memoryview(data)
    holes = []                                    # (start, end) of initialiser payloads, absolute offsets

    # a walk that keeps absolute offsets, for the three levels that lead to a payload
    def walk(b, base):
        i, res = 0, []
        while i < len(b):
            k, i = _rd_varint(b, i)
            f, wt = k >> 3, k & 7
            if wt == 0:
                v, i = _rd_varint(b, i)
                res.append((f, wt, v, None))
            elif wt == 2:
                n, i = _rd_varint(b, i)
                if i + n > len(b):
                    raise ValueError("truncated protobuf")
                res.append((f, wt, b[i:i + n], base + i))
                i += n
            elif wt == 5:
                res.append((f, wt, bytes(b[i:i + 4]), None))
                i += 4
            elif wt == 1:
                res.append((f, wt, bytes(b[i:i + 8]), None))
                i += 8
            else:
                raise ValueError(f"wire type {wt}")
        return res

    for f, wt, v, off in walk(mv, 0):
        out["model_field_order"].append(f)
        if f == 1:
            out["ir_version"] = v
        elif f == 2:
            out["producer_name"] = bytes(v).decode()
        elif f == 3:
            out["producer_version"] = bytes(v).decode()
        elif f == 4:
            out["domain"] = bytes(v).decode()
        elif f == 5:
            out["model_version"] = v
        elif f == 6:
            out["doc_string"] = bytes(v).decode()
        elif f == 8:
            o = {"domain": None, "version": None}
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    o["domain"] = bytes(v2).decode()
                elif f2 == 2:
                    o["version"] = v2
            out["opset_import"].append(o)
        elif f == 7:
            g = {"name": None, "field_order": [], "nodes": [], "initializers": [], "inputs": [], "outputs": [], "value_info": []}
            for gf, gwt, gv, goff in walk(v, off):
                if not g["field_order"] or g["field_order"][-1][0] != gf:
                    g["field_order"].append([gf, 1])
                else:
                    g["field_order"][-1][1] += 1
                if gf == 1:
                    n = {"op": None, "name": "", "domain": "", "inputs": [], "outputs": [], "attrs": [], "field_order": []}
                    for nf, _, nv in _fields(gv):
                        n["field_order"].append(nf)
                        if nf == 1:
                            n["inputs"].append(bytes(nv).decode())
                        elif nf == 2:
                            n["outputs"].append(bytes(nv).decode())
                        elif nf == 3:
                            n["name"] = bytes(nv).decode()
                        elif nf == 4:
                            n["op"] = bytes(nv).decode()
                        elif nf == 5:
                            n["attrs"].append(_attr_struct(nv))
                        elif nf == 7:
                            n["domain"] = bytes(nv).decode()
                    g["nodes"].append(n)
                elif gf == 2:
                    g["name"] = bytes(gv).decode()
                elif gf == 5:
                    t = {"name": "", "dims": [], "data_type": None, "storage": None, "nbytes": 0, "sha256": None, "field_order": []}
                    for tf, twt, tv, toff in walk(gv, goff):
                        t["field_order"].append(tf)
                        if tf == 1 and twt == 0:
                            t["dims"].append(tv)
                        elif tf == 2:
                            t["data_type"] = tv
                        elif tf == 8:
                            t["name"] = bytes(tv).decode()
                        elif tf in (4, 9) and twt == 2:
                            t["storage"] = "raw_data" if tf == 9 else "float_data"
                            t["nbytes"] = len(tv)
                            t["sha256"] = hashlib.sha256(tv).hexdigest()
                            holes.append((toff, toff + len(tv)))
                        elif tf == 4 and twt == 5:
                            t["storage"] = "float_data (unpacked)"
                    t["field_order"] = sorted(set(t["field_order"]), key=t["field_order"].index)
                    g["initializers"].append(t)
                elif gf == 11:
                    g["inputs"].append(_value_info_struct(gv))
                elif gf == 12:
                    g["outputs"].append(_value_info_struct(gv))
                elif gf == 13:
                    g["value_info"].append(_value_info_struct(gv))
            out["graph"] = g
    pos = 0
    for a, b in sorted(holes):
        skel.update(mv[pos:a])
        pos = b
    skel.update(mv[pos:])
    out["skeleton_sha256"] = skel.hexdigest()
    return out


def onnx_initializers(data: bytes) -> dict:
    """{name: float32 ndarray} of the FLOAT initialisers stored as raw_data or packed float_data."""
    res = {}
    for f, _, v in _fields(data):
        if f != 7:
            continue
        for gf, _, gv in _fields(v):
            if gf != 5:
                continue
            dims, name, arr, dt = [], "", None, None
            for tf, twt, tv in _fields(gv):
                if tf == 1 and twt == 0:
                    dims.append(tv)
                elif tf == 2:
                    dt = tv
                elif tf == 8:
                    name = bytes(tv).decode()
                elif tf in (4, 9) and twt == 2:
                    arr = np.frombuffer(tv, "<f4")
            if dt == 1 and arr is not None:
                res[name] = arr.reshape(dims)
    return res
