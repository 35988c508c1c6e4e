// model_parse.cpp -- minimal ONNX protobuf reader + APRW blob reader.
//
// Replaces onnx-go's `UnmarshalBinary` for the one graph family the reference exports
// (src/model/model.go:190-196 unmarshals it on EVERY Infer call; here it runs once per
// engine).  Accepted graph: [Conv(3x3, pads 1, stride 1|2) -> optional BatchNormalization
// -> Relu] x L, the global max pool, Flatten, Gemm -- i.e. ai/model.py:55-110 in eval mode,
// with BatchNorm either folded by the exporter or left as explicit nodes.
//
// The global pool: ai/model.py:106 is AdaptiveMaxPool2d(1).  The exporter the reference
// calls (ai/model.py:149-152: legacy TorchScript exporter, static 1x1x128x188 input) writes
// it as MaxPool(kernel_shape = the last conv's whole output map = [2,3], strides [2,3],
// pads 0) -- see tests/golden/onnx_real_export.json, the structure of a real export.
// GlobalMaxPool is what the same exporter writes when the map size is unknown; both are
// accepted, and a MaxPool that is anything but the whole map is rejected with the reason.
// The node list must be one chain: every node consumes the previous node's output.
#include "model_parse.h"

#include <cmath>
#include <cstring>
#include <exception>
#include <map>

namespace apr {
namespace {

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    Reader(const uint8_t *b, size_t n) : p(b), end(b + n) {}
    bool done() const { return p >= end || !ok; }
    uint64_t varint()
    {
        uint64_t v = 0;
        int shift = 0;
        while (p < end && shift < 64) {
            uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
        }
        ok = false;
        return 0;
    }
    // reads a key; returns field number, sets wire type
    int key(int &wt)
    {
        uint64_t k = varint();
        wt = (int)(k & 7);
        return (int)(k >> 3);
    }
    Reader sub()
    {
        uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return Reader(p, 0); }
        Reader r(p, (size_t)n);
        p += n;
        return r;
    }
    void skip(int wt)
    {
        switch (wt) {
        case 0: varint(); break;
        case 1: if (end - p < 8) ok = false; else p += 8; break;
        case 2: sub(); break;
        case 5: if (end - p < 4) ok = false; else p += 4; break;
        default: ok = false;
        }
    }
    std::string str()
    {
        Reader r = sub();
        return std::string((const char *)r.p, (size_t)(r.end - r.p));
    }
};

struct Tensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
    int dtype = 0;
};

struct Node {
    std::string op;
    std::vector<std::string> in, out;
    std::map<std::string, std::vector<int64_t>> ints;
    std::map<std::string, int64_t> i;
    std::map<std::string, float> f;
    std::map<std::string, std::string> s;
};

// the engine's fixed input map (src/model/model.go:33,36: Nmels x SpectrogramFrames)
constexpr int kInH = 128, kInW = 188;

bool parse_tensor(Reader r, std::string &name, Tensor &t)
{
    const uint8_t *raw = nullptr;
    size_t raw_n = 0;
    while (!r.done()) {
        int wt, f = r.key(wt);
        if (f == 1 && wt == 0) t.dims.push_back((int64_t)r.varint());
        else if (f == 1 && wt == 2) { Reader s = r.sub(); while (!s.done()) t.dims.push_back((int64_t)s.varint()); }
        else if (f == 2 && wt == 0) t.dtype = (int)r.varint();
        else if (f == 4 && wt == 2) { Reader s = r.sub(); size_t n = (size_t)(s.end - s.p) / 4; size_t o = t.data.size(); t.data.resize(o + n); memcpy(t.data.data() + o, s.p, n * 4); }
        else if (f == 4 && wt == 5) { float v; if (r.end - r.p < 4) return false; memcpy(&v, r.p, 4); r.p += 4; t.data.push_back(v); }
        else if (f == 8 && wt == 2) name = r.str();
        else if (f == 9 && wt == 2) { Reader s = r.sub(); raw = s.p; raw_n = (size_t)(s.end - s.p); }
        else r.skip(wt);
    }
    if (!r.ok) return false;
    if (raw) { t.data.resize(raw_n / 4); memcpy(t.data.data(), raw, (raw_n / 4) * 4); }
    return true;
}

bool parse_attr(Reader r, Node &n)
{
    std::string name;
    std::vector<int64_t> ints;
    bool has_i = false, has_f = false, has_s = false;
    std::string sv;
    int64_t iv = 0;
    float fv = 0;
    while (!r.done()) {
        int wt, f = r.key(wt);
        if (f == 1 && wt == 2) name = r.str();
        else if (f == 2 && wt == 5) { if (r.end - r.p < 4) return false; memcpy(&fv, r.p, 4); r.p += 4; has_f = true; }
        else if (f == 3 && wt == 0) { iv = (int64_t)r.varint(); has_i = true; }
        else if (f == 4 && wt == 2) { sv = r.str(); has_s = true; }
        else if (f == 8 && wt == 0) ints.push_back((int64_t)r.varint());
        else if (f == 8 && wt == 2) { Reader s = r.sub(); while (!s.done()) ints.push_back((int64_t)s.varint()); }
        else r.skip(wt);
    }
    if (!r.ok) return false;
    if (!ints.empty()) n.ints[name] = ints;
    if (has_i) n.i[name] = iv;
    if (has_f) n.f[name] = fv;
    if (has_s) n.s[name] = sv;
    return true;
}

bool parse_node(Reader r, Node &n)
{
    while (!r.done()) {
        int wt, f = r.key(wt);
        if (f == 1 && wt == 2) n.in.push_back(r.str());
        else if (f == 2 && wt == 2) n.out.push_back(r.str());
        else if (f == 4 && wt == 2) n.op = r.str();
        else if (f == 5 && wt == 2) { if (!parse_attr(r.sub(), n)) return false; }
        else r.skip(wt);
    }
    return r.ok;
}

// ValueInfoProto{1 name, 2 TypeProto{1 Tensor{2 TensorShapeProto{1 Dimension{1 dim_value | 2 dim_param}}}}}; a symbolic dim reads as -1
bool parse_value_info(Reader r, std::string &name, std::vector<int64_t> &dims)
{
    while (!r.done()) {
        int wt, f = r.key(wt);
        if (f == 1 && wt == 2) name = r.str();
        else if (f == 2 && wt == 2) {
            Reader ty = r.sub();
            while (!ty.done()) {
                int wt2, f2 = ty.key(wt2);
                if (f2 == 1 && wt2 == 2) {
                    Reader tt = ty.sub();
                    while (!tt.done()) {
                        int wt3, f3 = tt.key(wt3);
                        if (f3 == 2 && wt3 == 2) {
                            Reader sh = tt.sub();
                            while (!sh.done()) {
                                int wt4, f4 = sh.key(wt4);
                                if (f4 == 1 && wt4 == 2) {
                                    Reader d = sh.sub();
                                    int64_t v = -1;
                                    while (!d.done()) { int wt5, f5 = d.key(wt5); if (f5 == 1 && wt5 == 0) v = (int64_t)d.varint(); else d.skip(wt5); }
                                    if (!d.ok) return false;
                                    dims.push_back(v);
                                } else sh.skip(wt4);
                            }
                            if (!sh.ok) return false;
                        } else tt.skip(wt3);
                    }
                    if (!tt.ok) return false;
                } else ty.skip(wt2);
            }
            if (!ty.ok) return false;
        } else r.skip(wt);
    }
    return r.ok;
}

bool all_equal(const std::vector<int64_t> &v, int64_t x)
{
    for (auto e : v) if (e != x) return false;
    return true;
}

bool parse_onnx(const uint8_t *bytes, size_t nbytes, ParsedModel &out, std::string &err)
{
    Reader m(bytes, nbytes);
    const uint8_t *gp = nullptr;
    size_t gn = 0;
    while (!m.done()) {
        int wt, f = m.key(wt);
        if (f == 7 && wt == 2) { Reader g = m.sub(); gp = g.p; gn = (size_t)(g.end - g.p); }
        else m.skip(wt);
    }
    if (!m.ok || !gp) { err = "not an ONNX ModelProto (no graph)"; return false; }
    std::vector<Node> nodes;
    std::map<std::string, Tensor> inits;
    std::map<std::string, std::vector<int64_t>> graph_inputs;
    Reader g(gp, gn);
    while (!g.done()) {
        int wt, f = g.key(wt);
        if (f == 1 && wt == 2) { Node n; if (!parse_node(g.sub(), n)) { err = "bad NodeProto"; return false; } nodes.push_back(std::move(n)); }
        else if (f == 5 && wt == 2) {
            std::string name; Tensor t;
            if (!parse_tensor(g.sub(), name, t)) { err = "bad TensorProto"; return false; }
            if (t.dtype == 1) inits[name] = std::move(t);
        } else if (f == 11 && wt == 2) {
            std::string name; std::vector<int64_t> dims;
            if (!parse_value_info(g.sub(), name, dims)) { err = "bad ValueInfoProto"; return false; }
            graph_inputs[name] = std::move(dims);
        } else g.skip(wt);
    }
    if (!g.ok) { err = "truncated ONNX graph"; return false; }

    auto get = [&](const std::string &name, size_t want, const Tensor *&t) -> bool {
        auto it = inits.find(name);
        if (it == inits.end()) { err = "missing float initializer '" + name + "'"; return false; }
        if (want && it->second.data.size() != want) { err = "initializer '" + name + "' has wrong size"; return false; }
        t = &it->second;
        return true;
    };

    size_t i = 0;
    bool seen_pool = false, seen_gemm = false;
    int map_h = kInH, map_w = kInW;      // output map of the conv chain so far, at the engine's input size
    std::string cur;                     // name of the value the chain has produced so far ("" before the first node)
    // one chain: node i reads what node i-1 wrote (a graph with a branch or a skipped value would otherwise be
    // accepted and scored as a different function)
    auto follows = [&](const Node &n) -> bool {
        if (n.in.empty() || n.out.empty() || n.out[0].empty()) { err = "node '" + n.op + "' has no input or output"; return false; }
        if (!cur.empty() && n.in[0] != cur) { err = "graph is not one chain: '" + n.op + "' reads '" + n.in[0] + "', the previous node wrote '" + cur + "'"; return false; }
        return true;
    };
    while (i < nodes.size()) {
        const Node &n = nodes[i];
        if (seen_gemm && n.op != "Constant") { err = "node '" + n.op + "' after the classifier Gemm"; return false; }
        if (n.op == "Conv") {
            if (seen_pool) { err = "Conv after the global max pool"; return false; }
            if (n.in.size() < 2) { err = "Conv without weights"; return false; }
            if (!follows(n)) return false;
            if (cur.empty()) {
                // the data input: when the graph declares its shape it must be the engine's [N,1,128,188] (ai/model.py:150)
                auto gi = graph_inputs.find(n.in[0]);
                if (gi != graph_inputs.end() && !gi->second.empty()) {
                    const auto &d = gi->second;
                    if (d.size() != 4 || (d[1] >= 0 && d[1] != 1) || (d[2] >= 0 && d[2] != kInH) || (d[3] >= 0 && d[3] != kInW)) {
                        err = "model input '" + n.in[0] + "' is not [N,1," + std::to_string(kInH) + "," + std::to_string(kInW) + "]: the engine scores 128x188 spectrograms only";
                        return false;
                    }
                }
            }
            const Tensor *w = nullptr;
            if (!get(n.in[1], 0, w)) return false;
            if (w->dims.size() != 4 || w->dims[2] != 3 || w->dims[3] != 3) { err = "Conv kernel is not 3x3"; return false; }
            ConvSpec c;
            // widths are bounded before the int casts: the reference's widest layer is 1024 (ai/model.py:98-103)
            if (w->dims[0] < 1 || w->dims[0] > kMaxConvWidth || w->dims[1] < 1 || w->dims[1] > kMaxConvWidth) { err = "Conv width out of range (1.." + std::to_string(kMaxConvWidth) + ")"; return false; }
            c.cout = (int)w->dims[0];
            c.cin = (int)w->dims[1];
            if (w->data.size() != (size_t)c.cout * c.cin * 9) { err = "Conv weight size mismatch"; return false; }
            auto ks = n.ints.find("kernel_shape");
            if (ks != n.ints.end() && (ks->second.size() != 2 || ks->second[0] != 3 || ks->second[1] != 3)) { err = "Conv kernel_shape is not [3,3]"; return false; }
            auto ap = n.s.find("auto_pad");
            if (ap != n.s.end() && ap->second != "NOTSET") { err = "Conv auto_pad '" + ap->second + "' unsupported"; return false; }
            auto st = n.ints.find("strides");
            c.stride = (st != n.ints.end() && !st->second.empty()) ? (int)st->second[0] : 1;
            if (st != n.ints.end() && st->second.size() == 2 && st->second[0] != st->second[1]) { err = "anisotropic stride"; return false; }
            auto pd = n.ints.find("pads");
            if (pd == n.ints.end() || pd->second.size() != 4 || pd->second[0] != 1 || pd->second[1] != 1 || pd->second[2] != 1 || pd->second[3] != 1) { err = "Conv pads must be [1,1,1,1]"; return false; }
            auto gr = n.i.find("group");
            if (gr != n.i.end() && gr->second != 1) { err = "grouped Conv unsupported"; return false; }
            auto dl = n.ints.find("dilations");
            if (dl != n.ints.end()) for (auto d : dl->second) if (d != 1) { err = "dilated Conv unsupported"; return false; }
            c.w = w->data;
            c.b.assign(c.cout, 0.0f);
            if (n.in.size() >= 3 && !n.in[2].empty()) {
                const Tensor *b = nullptr;
                if (!get(n.in[2], (size_t)c.cout, b)) return false;
                c.b = b->data;
            }
            cur = n.out[0];
            i++;
            if (i < nodes.size() && nodes[i].op == "BatchNormalization") {
                const Node &bn = nodes[i];
                if (bn.in.size() < 5) { err = "BatchNormalization needs 5 inputs"; return false; }
                if (!follows(bn)) return false;
                cur = bn.out[0];
                const Tensor *g2, *be, *mu, *va;
                if (!get(bn.in[1], c.cout, g2) || !get(bn.in[2], c.cout, be) || !get(bn.in[3], c.cout, mu) || !get(bn.in[4], c.cout, va)) return false;
                auto ep = bn.f.find("epsilon");
                double eps = ep != bn.f.end() ? (double)ep->second : 1e-5;
                for (int o = 0; o < c.cout; o++) {
                    double scale = (double)g2->data[o] / std::sqrt((double)va->data[o] + eps);
                    for (int k = 0; k < c.cin * 9; k++) c.w[(size_t)o * c.cin * 9 + k] = (float)((double)c.w[(size_t)o * c.cin * 9 + k] * scale);
                    c.b[o] = (float)(((double)c.b[o] - (double)mu->data[o]) * scale + (double)be->data[o]);
                }
                i++;
            }
            if (i >= nodes.size() || nodes[i].op != "Relu") { err = "Conv block without Relu"; return false; }
            if (!follows(nodes[i])) return false;
            cur = nodes[i].out[0];
            i++;
            if (c.stride == 1 || c.stride == 2) { map_h = (map_h + 2 - 3) / c.stride + 1; map_w = (map_w + 2 - 3) / c.stride + 1; }
            out.convs.push_back(std::move(c));
        } else if (n.op == "GlobalMaxPool" || n.op == "MaxPool") {
            if (seen_pool) { err = "second pooling node"; return false; }
            if (out.convs.empty()) { err = "pooling before the first Conv"; return false; }
            if (!follows(n)) return false;
            if (n.op == "MaxPool") {
                // only the whole-map pool is the reference's AdaptiveMaxPool2d(1) (ai/model.py:106); anything else is another network
                const std::string want = "[" + std::to_string(map_h) + "," + std::to_string(map_w) + "]";
                auto ks = n.ints.find("kernel_shape");
                if (ks == n.ints.end() || ks->second.size() != 2 || ks->second[0] != map_h || ks->second[1] != map_w) {
                    std::string got = "none";
                    if (ks != n.ints.end()) { got = "["; for (size_t q = 0; q < ks->second.size(); q++) got += (q ? "," : "") + std::to_string(ks->second[q]); got += "]"; }
                    err = "MaxPool kernel_shape " + got + " is not the last conv's whole " + want + " output map (only the global pool of AdaptiveMaxPool2d(1) is supported)";
                    return false;
                }
                auto pd = n.ints.find("pads");
                if (pd != n.ints.end() && !all_equal(pd->second, 0)) { err = "MaxPool with padding is not a global pool"; return false; }
                auto dl = n.ints.find("dilations");
                if (dl != n.ints.end() && !all_equal(dl->second, 1)) { err = "dilated MaxPool unsupported"; return false; }
                auto st = n.ints.find("strides");
                if (st != n.ints.end()) for (auto v : st->second) if (v < 1) { err = "MaxPool stride < 1"; return false; }
                auto ap = n.s.find("auto_pad");
                if (ap != n.s.end() && ap->second != "NOTSET" && ap->second != "VALID") { err = "MaxPool auto_pad '" + ap->second + "' unsupported"; return false; }
                auto so = n.i.find("storage_order");
                if (so != n.i.end() && so->second != 0) { err = "MaxPool storage_order unsupported"; return false; }
                // ceil_mode cannot add an output when the kernel covers the whole unpadded map: ignored
                if (n.out.size() > 1 && !n.out[1].empty()) { err = "MaxPool Indices output unsupported"; return false; }
            }
            seen_pool = true;
            cur = n.out[0];
            i++;
        }
        else if (n.op == "Constant") { i++; }     // a side input (Reshape's shape): not part of the chain
        else if (n.op == "Flatten" || n.op == "Reshape" || n.op == "Dropout" || n.op == "Identity" || n.op == "Squeeze") {
            if (!follows(n)) return false;
            if (!seen_pool) { err = "'" + n.op + "' before the global max pool"; return false; }
            cur = n.out[0];
            i++;
        }
        else if (n.op == "Gemm") {
            if (!seen_pool) { err = "Gemm before the global max pool"; return false; }
            if (n.in.size() < 2) { err = "Gemm without weights"; return false; }
            if (!follows(n)) return false;
            auto ta = n.i.find("transA");
            if (ta != n.i.end() && ta->second != 0) { err = "Gemm transA unsupported"; return false; }
            const Tensor *w = nullptr;
            if (!get(n.in[1], 0, w)) return false;
            if (w->dims.size() != 2) { err = "Gemm weight is not 2-D"; return false; }
            auto tb = n.i.find("transB");
            bool transB = tb != n.i.end() && tb->second != 0;
            // dims are checked as int64 (a huge product must not reach resize()) and against the payload
            const int64_t cls64 = transB ? w->dims[0] : w->dims[1], feat64 = transB ? w->dims[1] : w->dims[0];
            if (cls64 < 1 || cls64 > kMaxClasses || feat64 < 1 || feat64 > kMaxConvWidth) { err = "Gemm shape out of range (classes 1.." + std::to_string(kMaxClasses) + ", features 1.." + std::to_string(kMaxConvWidth) + ")"; return false; }
            int d0 = (int)w->dims[0], d1 = (int)w->dims[1];
            if (w->data.size() != (size_t)d0 * (size_t)d1) { err = "Gemm weight size does not match its dims"; return false; }
            out.n_classes = transB ? d0 : d1;
            out.features = transB ? d1 : d0;
            out.fc_w.resize((size_t)d0 * d1);
            if (transB) out.fc_w = w->data;
            else for (int r = 0; r < d0; r++) for (int c = 0; c < d1; c++) out.fc_w[(size_t)c * d0 + r] = w->data[(size_t)r * d1 + c];
            float alpha = 1.0f, beta = 1.0f;
            if (n.f.count("alpha")) alpha = n.f.at("alpha");
            if (n.f.count("beta")) beta = n.f.at("beta");
            for (auto &v : out.fc_w) v *= alpha;
            out.fc_b.assign(out.n_classes, 0.0f);
            if (n.in.size() >= 3 && !n.in[2].empty()) {
                const Tensor *b = nullptr;
                if (!get(n.in[2], (size_t)out.n_classes, b)) return false;
                for (int o = 0; o < out.n_classes; o++) out.fc_b[o] = beta * b->data[o];
            }
            seen_gemm = true;
            cur = n.out[0];
            i++;
        } else { err = "unsupported ONNX op '" + n.op + "'"; return false; }
    }
    if (out.convs.empty() || !seen_pool || !seen_gemm) { err = "graph is not Conv/Relu chain + global max pool + Gemm"; return false; }
    return true;
}

bool parse_aprw(const uint8_t *bytes, size_t n, ParsedModel &out, std::string &err)
{
    static const int CIN[12] = {1, 32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 1024};
    static const int COUT[12] = {32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 1024, 1024};
    static const int STRIDE[12] = {1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2};
    if (n < 16) { err = "APRW blob truncated"; return false; }
    uint32_t ver, ncls;
    memcpy(&ver, bytes + 4, 4);
    memcpy(&ncls, bytes + 8, 4);
    if (ver != 1) { err = "unsupported APRW version"; return false; }
    size_t want = 0;
    for (int l = 0; l < 12; l++) want += (size_t)COUT[l] * CIN[l] * 9 + 5 * (size_t)COUT[l];
    want += (size_t)ncls * 1024 + ncls;
    if (ncls < 1 || ncls > (uint32_t)kMaxClasses || n != 16 + want * 4) { err = "APRW blob size does not match its class count"; return false; }
    std::vector<float> f(want);
    memcpy(f.data(), bytes + 16, want * 4);
    const float *p = f.data();
    for (int l = 0; l < 12; l++) {
        ConvSpec c;
        c.cin = CIN[l]; c.cout = COUT[l]; c.stride = STRIDE[l];
        size_t wn = (size_t)c.cout * c.cin * 9;
        const float *w = p; p += wn;
        const float *b = p; p += c.cout;
        const float *g = p; p += c.cout;
        const float *be = p; p += c.cout;
        const float *mu = p; p += c.cout;
        const float *va = p; p += c.cout;
        c.w.resize(wn);
        c.b.resize(c.cout);
        for (int o = 0; o < c.cout; o++) {
            double scale = (double)g[o] / std::sqrt((double)va[o] + 1e-5);
            for (int k = 0; k < c.cin * 9; k++) c.w[(size_t)o * c.cin * 9 + k] = (float)((double)w[(size_t)o * c.cin * 9 + k] * scale);
            c.b[o] = (float)(((double)b[o] - (double)mu[o]) * scale + (double)be[o]);
        }
        out.convs.push_back(std::move(c));
    }
    out.n_classes = (int)ncls;
    out.features = 1024;
    out.fc_w.assign(p, p + (size_t)ncls * 1024); p += (size_t)ncls * 1024;
    out.fc_b.assign(p, p + ncls);
    return true;
}

}  // namespace

bool parse_model(const void *bytes, size_t n, ParsedModel &out, std::string &err)
{
    const uint8_t *b = (const uint8_t *)bytes;
    out = ParsedModel();
    bool ok;
    try {   // no exception may cross the C ABI (include/apr.h): an allocation failure inside the parser is a bad model
        if (n >= 4 && memcmp(b, "APRW", 4) == 0) ok = parse_aprw(b, n, out, err);
        else ok = parse_onnx(b, n, out, err);
    } catch (const std::exception &ex) {
        err = std::string("parser ran out of memory or failed: ") + ex.what();
        return false;
    }
    if (!ok) return false;
    // chain consistency: channel counts must follow on, first layer is single channel
    int c = 1;
    for (size_t l = 0; l < out.convs.size(); l++) {
        const ConvSpec &cv = out.convs[l];
        if (cv.cin != c) { err = "conv layer " + std::to_string(l) + " input channels do not follow on"; return false; }
        if (cv.cin < 1 || cv.cout < 1) { err = "conv layer " + std::to_string(l) + " has no channels"; return false; }
        if (cv.stride != 1 && cv.stride != 2) { err = "conv stride must be 1 or 2"; return false; }
        c = cv.cout;
    }
    if (out.features != c) { err = "classifier input width does not match last conv"; return false; }
    return true;
}

}  // namespace apr
