// engine.hip -- C ABI (include/apr.h) of the MI355X-native DTrack detection engine.
//
// Host side of the hot path: table construction (Hann windows, twiddles, mel lenses of
// both reference flavours), weight ingest and re-layout, buffer management, kernel
// launches on one HIP stream per engine, HIP-event profiling.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded with dlopen on first use (apr_gather_rows_device)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/apr.h"
#include "conv.hip.h"
#include "frontend.hip.h"
#include "model_parse.h"
#include "wino.hip.h"
#include "wino4.hip.h"
#include "wino4_fused.hip.h"
#include "resample.hip.h"
#include "report.hip.h"
#include "conv12_fused.hip.h"
#include "wino_fused.hip.h"
#include "wino_fused8.hip.h"
#include "wino_s2.hip.h"
#include "wino_s2_fused.hip.h"

using namespace apr;

// ---- apr_prepare_batch_device_static: the front end's one scratch buffer (mel power, frame-major
// [window][188][128] f32) as a file-scope device array, so that the call allocates nothing.  Every GPU the library
// runs on holds its own copy (5.9 MiB) until the process ends.  In host code `scratch` is only a stand-in for the
// symbol: the device address is asked for once per device (hipGetSymbolAddress, at engine creation, with that
// device current) and kept in g_scratch_ptr; kernels and host calls only ever see that pointer.
static __device__ __align__(256) float scratch[(size_t)APR_STATIC_MAX_WINDOWS * kSpecFrames * kNmels];

namespace {

constexpr int kMaxDevices = 64;
float *g_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_scratch_mu;

// with `device` current; returns nullptr (and leaves the slot empty) on failure
float *resolve_scratch(int device, std::string &err)
{
    if (device < 0 || device >= kMaxDevices) { err = "device index beyond the static-scratch table"; return nullptr; }
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        float *ptr = nullptr;
        const hipError_t r = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (r != hipSuccess || !ptr) { err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(r); return nullptr; }
        g_scratch_ptr[device] = ptr;
    }
    return g_scratch_ptr[device];
}

thread_local std::string g_create_err;

struct Layer {
    int cin, cout, stride, hin, win, hout, wout;
    float *d_w = nullptr;   // [9][cin][cout]
    float *d_b = nullptr;   // [cout]
    int bn_tile = 0, bm_tile = 128;
    // Winograd F(2x2,3x3) path (stride-1 layers with enough channels to amortise the transforms)
    int wino = 0;           // 0 = direct implicit GEMM, 2 = F(2x2,3x3) in three kernels, 3 = in one (wino_fused*_f2),
                            // 4 = stride 2 in polyphase Winograd F(2,2) form (wino_s2.hip.h: 25 GEMMs + output transform),
                            // 5 = the same in one kernel (wino_s2_fused.hip.h),
                            // 6 = F(4x4,3x3) in three kernels (wino4.hip.h: 36 positions),
                            // 7 = F(4x4,3x3) in two: packed input transform + GEMMs with the output transform (wino4_fused.hip.h)
    int gn = 0;             // wino == 3 (8 waves), 5: N tiles per group in the unit order (wf_unit); 0 = all
    int fused_waves = 8;    // wino == 3: wino_fused8_f2 (8 waves a block) or wino_fused_f2 (4)
    float *d_u = nullptr;   // [16][cin][cout] = G g G^T (wino == 2), or wf_pack_u's image (wino == 3, fused kernel)
    int th = 0, tw = 0;     // 2x2-output tiles per window
    int wbm_tile = 128;
    int ksplit = 1;         // > 1: direct kernel split along K (latency engines), partial sums in d_part
    int wsplit = 1;         // > 1: the Winograd-domain GEMMs split along K (latency engines), copies of M in d_M
    int slot = -1, slot_in = -1, slot_out = -1;   // profile slots of its kernels (slot_out: also the split-K reduce)
};

struct ProfPair {
    hipEvent_t a, b;
    int slot;
};

}  // namespace

struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    size_t k0 = 0, k1 = 0, k2 = 0;   // the call shape the graph was captured for
    double kd = 0.0;
    int ki = 0;
};
constexpr int kGraphWindows = 4;                                        // largest call that is replayed
constexpr size_t kGraphIn = (size_t)kGraphWindows * APR_SAMPLE_SIZE;    // PCM span or spectrograms of such a call
constexpr size_t kGraphOut = (size_t)kGraphWindows * APR_SPEC_ELEMS * sizeof(float);

struct apr_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int n_classes = 0;
    int max_batch = 1024;
    bool fuse12 = false;   // layers 1 + 2 run as conv12_fused_kernel (conv1's output never goes to HBM)
    int n_cus = 256;       // compute units of the device (persistent-kernel grids)
    int16_t *d_rs_in = nullptr, *d_rs_out = nullptr;   // apr_upsample staging (host-pointer form)
    double *d_rs_taps = nullptr;
    size_t rs_cap = 0, rs_out_cap = 0;
    int features = 0, pool_pixels = 0;

    // front end
    std::vector<void *> allocs;   // everything hipMalloc'd (freed in destroy)
    FrontendTables tab[2];
    float *d_melT = nullptr;      // [max_batch][188][128]
    float *d_spec = nullptr;      // [max_batch][128][188]

    // CNN
    std::vector<Layer> layers;
    float *d_act[2] = {nullptr, nullptr};
    float *d_wfc = nullptr, *d_bfc = nullptr;
    float *d_V = nullptr, *d_M = nullptr;   // Winograd-domain activations [16][T][Cin] / [16][T][Cout]
    float *d_part = nullptr;                // split-K partial sums [ksplit][M][Cout]
    float *d_probe = nullptr;               // apr_mfma_probe's sink

    // staging for the host-pointer entry points
    uint8_t *d_pcm = nullptr;
    size_t pcm_cap = 0;
    double *d_probs = nullptr;
    int32_t *d_label = nullptr;
    uint8_t *d_match = nullptr;
    float *d_logits = nullptr;

    // apr_sweep_files: two pinned host buffers + two device buffers, filled on a copy stream while the
    // previous pass computes
    hipStream_t copy_stream = nullptr;
    uint8_t *h_pin[2] = {nullptr, nullptr};
    uint8_t *d_pin[2] = {nullptr, nullptr};
    double *h_probs[2] = {nullptr, nullptr};
    int32_t *h_label[2] = {nullptr, nullptr};
    uint8_t *h_match[2] = {nullptr, nullptr};
    double *d_probs2[2] = {nullptr, nullptr};
    int32_t *d_label2[2] = {nullptr, nullptr};
    uint8_t *d_match2[2] = {nullptr, nullptr};
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    size_t pin_cap = 0, pin_want = 0;   // bytes of a complete staging set / bytes the (possibly partial) set was sized for

    // Calls of at most 4 windows through the host-pointer entry points (apr_score, apr_infer, apr_prepare[_batch]: the
    // daemon's one-clip-per-call pattern) replay a captured hipGraph -- H2D from pinned staging, every kernel, D2H into
    // pinned staging -- instead of ~30 launches a call; re-captured when the call's shape or decision rule changes.
    bool graph_on = false;
    GraphSlot g_score, g_infer, g_prep[2];
    uint8_t *h_g_in = nullptr, *h_g_out = nullptr;   // kGraphIn / kGraphOut bytes, pinned

    // apr_stream_tick: the previous 1-s segment of every live stream stays on the GPU
    uint8_t *d_ring = nullptr;      // [ring_cap][APR_SAMPLE_SIZE]: previous || new
    uint8_t *h_ring = nullptr;      // [ring_cap][APR_BYTES_PER_SECOND], pinned: the new segments of a tick
    int ring_cap = 0, ring_n = 0;   // streams allocated / streams primed (0 = the next tick only primes)

    // profiling
    bool prof_on = false;
    apr_profile_slot slots[APR_PROF_SLOTS];
    int n_slots = 0;
    int head_slot = -1;
    std::vector<ProfPair> pending;
    std::vector<hipEvent_t> free_events;
};

namespace {

#define HIPCHK(e, call)                                                                         \
    do {                                                                                        \
        hipError_t _r = (call);                                                                 \
        if (_r != hipSuccess) {                                                                 \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_r);                       \
            return APR_ERR_DEVICE;                                                              \
        }                                                                                       \
    } while (0)

template <typename T>
int dev_alloc(apr_engine *e, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t r = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (r != hipSuccess) {
        e->err = std::string("hipMalloc: ") + hipGetErrorString(r);
        return APR_ERR_NOMEM;
    }
    e->allocs.push_back(q);
    *p = (T *)q;
    return APR_OK;
}

// release one dev_alloc'd buffer before destroy (buffers that are re-sized between calls)
template <typename T>
void dev_free(apr_engine *e, T **p)
{
    if (!*p) return;
    auto it = std::find(e->allocs.begin(), e->allocs.end(), (void *)*p);
    if (it != e->allocs.end()) e->allocs.erase(it);
    (void)hipFree(*p);
    *p = nullptr;
}

template <typename T>
int dev_upload(apr_engine *e, T **p, const std::vector<T> &h)
{
    int rc = dev_alloc(e, p, h.size());
    if (rc) return rc;
    HIPCHK(e, hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return APR_OK;
}

// ---- mel lenses ------------------------------------------------------------------------------
double hz_to_mel(double hz) { return 2595.0 * std::log10(1.0 + hz / 700.0); }        // dsp.go:158-160
double mel_to_hz(double mel) { return 700.0 * (std::pow(10.0, mel / 2595.0) - 1.0); } // dsp.go:163-165

struct Csr {
    std::vector<int> start, len, off;
    std::vector<float> w;
};

// src/model/dsp.go:31-80: FFT-bin-quantised, unnormalised HTK triangles.
Csr mel_lens_go()
{
    const int nbins = kNfft / 2 + 1;
    double pts[kNmels + 2];
    const double mn = hz_to_mel(0.0), mx = hz_to_mel(APR_SAMPLE_RATE / 2.0);
    const double step = (mx - mn) / (double)(kNmels + 1);
    int bins[kNmels + 2];
    for (int i = 0; i < kNmels + 2; i++) {
        pts[i] = mel_to_hz(mn + (double)i * step);
        bins[i] = (int)std::floor((kNfft + 1) * pts[i] / (double)APR_SAMPLE_RATE);
    }
    Csr c;
    for (int m = 0; m < kNmels; m++) {
        const int s = bins[m], ce = bins[m + 1], en = std::min(bins[m + 2], nbins);   // dsp.go:93-97 clamps end
        c.start.push_back(s);
        c.len.push_back(std::max(en - s, 0));
        c.off.push_back((int)c.w.size());
        for (int f = s; f < en; f++) {
            double wv = 0.0;
            if (f < ce) wv = (double)(f - s) / (double)(ce - s);
            else wv = (double)(bins[m + 2] - f) / (double)(bins[m + 2] - ce);
            c.w.push_back((float)wv);
        }
    }
    return c;
}

// ai/model.py:28-34: librosa.filters.mel(htk=True, norm='slaney', dtype=float32).
Csr mel_lens_librosa()
{
    const int nbins = kNfft / 2 + 1;
    std::vector<double> ff(nbins), mf(kNmels + 2);
    for (int k = 0; k < nbins; k++) ff[k] = (double)k * (APR_SAMPLE_RATE / 2.0) / (double)(nbins - 1);
    const double mn = hz_to_mel(0.0), mx = hz_to_mel(APR_SAMPLE_RATE / 2.0);
    for (int i = 0; i < kNmels + 2; i++) mf[i] = mel_to_hz(mn + (mx - mn) * (double)i / (double)(kNmels + 1));
    Csr c;
    for (int m = 0; m < kNmels; m++) {
        const double fd0 = mf[m + 1] - mf[m], fd1 = mf[m + 2] - mf[m + 1], enorm = 2.0 / (mf[m + 2] - mf[m]);
        int first = -1, last = -1;
        std::vector<float> row(nbins);
        for (int k = 0; k < nbins; k++) {
            const double lo = -(mf[m] - ff[k]) / fd0, up = (mf[m + 2] - ff[k]) / fd1;
            double wv = std::max(0.0, std::min(lo, up));
            row[k] = (float)((double)(float)wv * enorm);
            if (row[k] != 0.0f) { if (first < 0) first = k; last = k; }
        }
        if (first < 0) { first = 0; last = -1; }
        c.start.push_back(first);
        c.len.push_back(last - first + 1);
        c.off.push_back((int)c.w.size());
        for (int k = first; k <= last; k++) c.w.push_back(row[k]);
    }
    return c;
}

int build_frontend(apr_engine *e)
{
    std::vector<float2> tw(kNfft);
    for (int i = 0; i < kNfft; i++) {
        const double a = 2.0 * M_PI * (double)i / (double)kNfft;
        tw[i] = make_float2((float)std::cos(a), (float)(-std::sin(a)));
    }
    // the same values in the order the lanes of passes 2 and 3 read them (frontend.hip.h, FrontendTables::tw23)
    std::vector<float2> tw23(1024);
    for (int t = 0; t < 16; t++)
        for (int k = 0; k < 16; k++) tw23[16 * t + k] = tw[8 * k * t];
    for (int t = 1; t <= 3; t++)
        for (int j = 0; j < 256; j++) tw23[256 * t + j] = tw[2 * j * t];
    float2 *d_tw = nullptr, *d_tw23 = nullptr;
    int rc = dev_upload(e, &d_tw, tw);
    if (rc) return rc;
    if ((rc = dev_upload(e, &d_tw23, tw23))) return rc;
    for (int fl = 0; fl < 2; fl++) {
        std::vector<float> win(kNfft);
        for (int n = 0; n < kNfft; n++) {
            // go-dsp window.Hann (symmetric, model.go:123) vs librosa's periodic hann
            const double den = fl == APR_FLAVOUR_GO ? (double)(kNfft - 1) : (double)kNfft;
            win[n] = (float)(0.5 * (1.0 - std::cos(2.0 * M_PI * (double)n / den)));
        }
        Csr c = fl == APR_FLAVOUR_GO ? mel_lens_go() : mel_lens_librosa();
        if ((int)c.w.size() > kMaxNnz) { e->err = "mel lens exceeds CSR capacity"; return APR_ERR_INVALID; }
        // the same lens with every band widened to whole quads of bins that start at a multiple of 4 (+0 weights outside
        // the band): the kernel reads four bins per LDS access on both operands
        std::vector<int> s4(kNmels), l4(kNmels), o4(kNmels);
        std::vector<float> wp;
        for (int m = 0; m < kNmels; m++) {
            const int st = c.start[m] & ~3, en = c.start[m] + c.len[m];
            s4[m] = st;
            l4[m] = c.len[m] > 0 ? (en - st + 3) / 4 : 0;
            o4[m] = (int)wp.size();
            for (int k = st; k < st + 4 * l4[m]; k++) wp.push_back(k >= c.start[m] && k < en ? c.w[(size_t)c.off[m] + (k - c.start[m])] : 0.0f);
        }
        if ((int)wp.size() > kMaxNnzPad) { e->err = "padded mel lens exceeds its LDS capacity"; return APR_ERR_INVALID; }
        float *d_win = nullptr, *d_w = nullptr, *d_wp = nullptr;
        int *d_s = nullptr, *d_l = nullptr, *d_o = nullptr, *d_s4 = nullptr, *d_l4 = nullptr, *d_o4 = nullptr;
        if ((rc = dev_upload(e, &d_win, win)) || (rc = dev_upload(e, &d_w, c.w)) || (rc = dev_upload(e, &d_s, c.start)) ||
            (rc = dev_upload(e, &d_l, c.len)) || (rc = dev_upload(e, &d_o, c.off)) || (rc = dev_upload(e, &d_wp, wp)) ||
            (rc = dev_upload(e, &d_s4, s4)) || (rc = dev_upload(e, &d_l4, l4)) || (rc = dev_upload(e, &d_o4, o4)))
            return rc;
        FrontendTables &T = e->tab[fl];
        T.window = d_win; T.tw = d_tw; T.tw23 = d_tw23; T.mel_start = d_s; T.mel_len = d_l; T.mel_off = d_o; T.mel_w = d_w;
        T.mel_start4 = d_s4; T.mel_len4 = d_l4; T.mel_off4 = d_o4; T.mel_wp = d_wp;
        T.nnz = (int)c.w.size();
        T.nnz_pad = (int)wp.size();
        T.n_frames = fl == APR_FLAVOUR_GO ? (APR_SAMPLE_SIZE / 2 - kNfft) / kHop + 1 : 1 + (APR_SAMPLE_SIZE / 2) / kHop;
        T.frame_origin = fl == APR_FLAVOUR_GO ? 0 : -(kNfft / 2);
        T.librosa = fl;
    }
    if ((rc = dev_alloc(e, &e->d_melT, (size_t)e->max_batch * kSpecFrames * kNmels))) return rc;
    if ((rc = dev_alloc(e, &e->d_spec, (size_t)e->max_batch * APR_SPEC_ELEMS))) return rc;
    return APR_OK;
}

void add_slot(apr_engine *e, const char *name, int layer)
{
    apr_profile_slot &s = e->slots[e->n_slots++];
    memset(&s, 0, sizeof s);
    snprintf(s.name, sizeof s.name, "%s", name);
    s.layer = layer;
}

// Every kernel-selection switch below is a DEVELOPER knob (A/B timing, the tests that compare kernel forms).  A release
// process ignores them all unless APR_DEV_KNOBS=1 is set beside them, so that a stray APR_WINOGRAD=0 inherited by a
// deployment daemon cannot silently cost a quarter of the throughput.
const char *dev_getenv(const char *name)
{
    const char *on = getenv("APR_DEV_KNOBS");
    if (!on || on[0] != '1') return nullptr;
    return getenv(name);
}

int build_cnn(apr_engine *e, const ParsedModel &pm)
{
    if (pm.n_classes < 1 || pm.n_classes > APR_MAX_CLASSES) { e->err = "class count out of range"; return APR_ERR_MODEL; }
    if (3 * (int)pm.convs.size() + 3 > APR_PROF_SLOTS) { e->err = "too many conv layers"; return APR_ERR_MODEL; }
    if (pm.features > 1024 || pm.features % 4) { e->err = "unsupported classifier width"; return APR_ERR_MODEL; }
    int h = kNmels, w = kSpecFrames, rc;
    size_t act_max = 0, v_max = 0, m_max = 0, part_max = 0, v_pad = 0;
    const char *senv = dev_getenv("APR_SPLITK");
    const bool splitk_on = !(senv && senv[0] == '0');
    // APR_WINOGRAD=0 disables the Winograd path; APR_WINO_MIN_CIN moves its channel threshold
    const char *wenv = dev_getenv("APR_WINOGRAD");
    const bool wino_on = !(wenv && wenv[0] == '0');
    const char *cenv = dev_getenv("APR_WINO_MIN_CIN");
    const int wino_min_cin = cenv ? atoi(cenv) : 128;
    // the one-kernel form (wino_fused.hip.h) wins while K = Cin is short; APR_WINO_FUSED=0 turns it
    // off, APR_WINO_FUSED_MAX_CIN moves its upper channel bound
    const char *fenv = dev_getenv("APR_WINO_FUSED");
    const bool fused_on = wino_on && !(fenv && fenv[0] == '0');
    const char *fcenv = dev_getenv("APR_WINO_FUSED_MAX_CIN");
    // Cin = 512 (layers 9, 11): the one-kernel form wins once a pass holds enough tiles to fill the chip (measured at
    // 2048 windows: 58.0 vs 67.6 ms and 29.0 vs 31.9 ms a step); small engines keep transform kernels + batched GEMM there
    // one-clip engines (max_batch <= 4): 24 or 16 work units of layers 5 and 7 leave the chip idle while each walks 16 or 32 K
    // steps (39.8 and 71.5 us); transform kernels + 16 batched GEMMs spread the same work over 100+ workgroups (25.6 and 31.8 us)
    // Stride-2 layers in polyphase Winograd F(2,2) form (wino_s2.hip.h): 25/36 of the products, against a round trip of
    // M[25][T][Cout] through HBM -- pays where Cin is large and a pass is big enough to fill the 25 GEMM grids.
    // APR_WINO_S2=0 turns it off; APR_WINO_S2_MIN_CIN / _MIN_PX / _MIN_BATCH move its bounds (developer knobs, tests).
    const char *s2env = dev_getenv("APR_WINO_S2");
    const bool wino_s2_on = wino_on && !(s2env && s2env[0] == '0');
    // Passes of >= 1024 windows: layers 6, 8 and 10 (Cin >= 256, >= 24 pixels a window; measured 105.6 vs 116.4, 95.4 vs 115.5
    // and 24.1 vs 29.4 ms a step; layer 4 loses 10 ms, layer 12 pads 2x3 pixels to 2x4).  Passes of 128..1023 windows (live
    // mode): layers 6 to 12 -- there the direct kernel's grids are short (layer 12 at 256 windows: 192 workgroups of 576 K steps).
    const bool s2_big = e->max_batch >= 1024;
    const int wino_s2_min_cin = dev_getenv("APR_WINO_S2_MIN_CIN") ? atoi(dev_getenv("APR_WINO_S2_MIN_CIN")) : 256;
    const int wino_s2_min_px = dev_getenv("APR_WINO_S2_MIN_PX") ? atoi(dev_getenv("APR_WINO_S2_MIN_PX")) : (s2_big ? 24 : 1);
    const int wino_s2_min_batch = dev_getenv("APR_WINO_S2_MIN_BATCH") ? atoi(dev_getenv("APR_WINO_S2_MIN_BATCH")) : 128;
    // F(4x4,3x3) in three kernels (wino4.hip.h), measured per 4096-window pass against the fused F(2x2) kernel (round 3,
    // tools/batch_breakdown.py): layer 7 10.1 vs 12.8 ms, layer 9 4.4 vs 6.2, layer 11 2.8 vs 3.1 -- adopted from Cin = 256 on;
    // layer 5 14.2 vs 13.5 and layer 3 22.8 vs 15.2 (the V / M round trips through HBM outweigh the 1.78x fewer multiplies): not.
    // Passes of 256 windows (live mode): layers 7 / 9 / 11 = 0.70 / 0.32 / 0.22 ms against 0.88 / 0.53 / 0.26 (6.44 vs 6.85 ms a pass).
    // One-clip engines keep their split-K forms.
    const int f4_default_min_cin = e->max_batch >= 128 ? 256 : 1 << 30;
    // F(4x4,3x3) in two kernels (wino4_fused.hip.h: packed input transform, then the 36 GEMMs with the output transform in
    // their kernel), measured per 4096-window pass, input transform included: layer 3 14.6 ms (fused F(2x2) 15.2), layer 5 10.3 (13.5),
    // layer 7 8.4 (three-kernel F(4x4) 10.1), layer 9 4.0 (4.4), layer 11 2.5 (2.8): every stride-1 layer from 3 on in engines of
    // >= 1024 windows.  Passes of 256 windows: layers 3 / 5 / 7 = 0.93 / 0.71 / 0.61 ms against 0.99 / 0.93 / 0.70, but layers 9 / 11
    // (24 / 8 M tiles: fewer units than CUs) 0.36 / 0.32 against 0.32 / 0.22 in three kernels: up to Cin = 256 there.
    const int f4f_default_min_cin = e->max_batch >= 128 ? 64 : 1 << 30;
    const int f4f_default_max_cin = e->max_batch >= 1024 ? 512 : 256;
    const int f4_min_batch = dev_getenv("APR_WINO_F4_MIN_BATCH") ? atoi(dev_getenv("APR_WINO_F4_MIN_BATCH")) : 1;
    const int fused_max_cin = fcenv ? atoi(fcenv) : (e->max_batch >= 1024 ? 512 : e->max_batch <= 4 ? 64 : 256);
    for (size_t l = 0; l < pm.convs.size(); l++) {
        const ConvSpec &cv = pm.convs[l];
        Layer L;
        L.cin = cv.cin; L.cout = cv.cout; L.stride = cv.stride; L.hin = h; L.win = w;
        L.hout = (h + 2 - 3) / cv.stride + 1;
        L.wout = (w + 2 - 3) / cv.stride + 1;
        if (l == 0) {
            if (cv.cin != 1 || cv.cout != 32 || cv.stride != 1) { e->err = "first conv must be 1->32, stride 1"; return APR_ERR_MODEL; }
        } else {
            if (cv.cin % 32 || cv.cout % 64) { e->err = "conv widths must be multiples of 32 (in) / 64 (out)"; return APR_ERR_MODEL; }
            L.bn_tile = cv.cout % 128 == 0 ? 128 : 64;
            // short grids (late layers: few pixels per window) use 64-row tiles: twice the blocks, at
            // three blocks per CU, fill the 256 CUs where 128-row tiles leave a partial round.
            // Decided once from max_batch so that a layer always runs the same instantiation.
            const long long m_full = (long long)e->max_batch * L.hout * L.wout;
            if (L.bn_tile == 128 && ((m_full + 127) / 128) * (cv.cout / 128) < 1024) L.bm_tile = 64;
            // Latency engines (max_batch <= 4, the daemon's one-clip-per-call pattern): the late layers have
            // a handful of tiles with hundreds of K steps each; split K over enough workgroups to fill the
            // chip.  Only there, so that every throughput engine keeps one summation order whatever its
            // max_batch (chunking stays bit-invisible); APR_SPLITK=0 turns it off.
            if (splitk_on && e->max_batch <= 4) {
                const long long blocks = ((m_full + L.bm_tile - 1) / L.bm_tile) * (cv.cout / L.bn_tile);
                const int kt = 9 * cv.cin / 16;
                int sp = (int)std::min<long long>(512 / std::max<long long>(blocks, 1), kt / 8);
                if (sp >= 2) {
                    L.ksplit = std::min(sp, 64);
                    part_max = std::max(part_max, (size_t)L.ksplit * (size_t)m_full * cv.cout);
                }
            }
        }
        // re-layout [cout][cin][3][3] -> [tap][cin][cout]
        std::vector<float> wt((size_t)9 * cv.cin * cv.cout);
        for (int o = 0; o < cv.cout; o++)
            for (int c = 0; c < cv.cin; c++)
                for (int t = 0; t < 9; t++) wt[((size_t)t * cv.cin + c) * cv.cout + o] = cv.w[((size_t)o * cv.cin + c) * 9 + t];
        if ((rc = dev_upload(e, &L.d_w, wt)) || (rc = dev_upload(e, &L.d_b, cv.b))) return rc;
        // (the fused kernels' tile arithmetic, wf_tile, is exact up to WF_MAX_TDIM tiles a side)
        const bool tdim_ok = (L.hout + 1) / 2 <= WF_MAX_TDIM && (L.wout + 1) / 2 <= WF_MAX_TDIM;
        // F(4x4,3x3), three kernels (wino4.hip.h): APR_WINO_F4_MIN_CIN .. APR_WINO_F4_MAX_CIN select the stride-1 layers (developer knobs)
        const int f4_min_cin = dev_getenv("APR_WINO_F4_MIN_CIN") ? atoi(dev_getenv("APR_WINO_F4_MIN_CIN")) : f4_default_min_cin;
        const int f4_max_cin = dev_getenv("APR_WINO_F4_MAX_CIN") ? atoi(dev_getenv("APR_WINO_F4_MAX_CIN")) : 4096;
        // the two-kernel form (wino4_fused.hip.h): APR_WINO_F4F_MIN_CIN .. APR_WINO_F4F_MAX_CIN
        const int f4f_min_cin = dev_getenv("APR_WINO_F4F_MIN_CIN") ? atoi(dev_getenv("APR_WINO_F4F_MIN_CIN")) : f4f_default_min_cin;
        const int f4f_max_cin = dev_getenv("APR_WINO_F4F_MAX_CIN") ? atoi(dev_getenv("APR_WINO_F4F_MAX_CIN")) : f4f_default_max_cin;
        if (wino_on && l > 0 && cv.stride == 1 && cv.cin >= f4f_min_cin && cv.cin <= f4f_max_cin && cv.cin % W4_BK == 0 && cv.cout % W4_BN == 0 &&
            (L.hout + 3) / 4 <= WF_MAX_TDIM && (L.wout + 3) / 4 <= WF_MAX_TDIM && e->max_batch >= f4_min_batch) {
            std::vector<float> u((size_t)36 * cv.cin * cv.cout);
            w4_pack_u(cv.w.data(), cv.cin, cv.cout, u.data());
            if ((rc = dev_upload(e, &L.d_u, u))) return rc;
            L.wino = 7;
            L.gn = dev_getenv("APR_WINO_GN") ? atoi(dev_getenv("APR_WINO_GN")) : 4;
            L.ksplit = 1;
            L.th = (L.hout + 3) / 4;
            L.tw = (L.wout + 3) / 4;
            v_max = std::max(v_max, (size_t)36 * L.th * L.tw * cv.cin);
            v_pad = std::max(v_pad, (size_t)36 * W4_TT * cv.cin);      // the last unit's rows past T
        } else if (wino_on && l > 0 && cv.stride == 1 && cv.cin >= f4_min_cin && cv.cin <= f4_max_cin && cv.cin % 16 == 0 && cv.cout % 128 == 0 &&
            e->max_batch >= f4_min_batch) {
            std::vector<float> u((size_t)36 * cv.cin * cv.cout);
            wino4_pack_u(cv.w.data(), cv.cin, cv.cout, u.data());
            if ((rc = dev_upload(e, &L.d_u, u))) return rc;
            L.wino = 6;
            L.ksplit = 1;
            L.th = (L.hout + 3) / 4;
            L.tw = (L.wout + 3) / 4;
            const long long t_full = (long long)e->max_batch * L.th * L.tw;
            L.wbm_tile = ((t_full + 127) / 128) * (cv.cout / 128) * 36 < 1024 ? 64 : 128;
            v_max = std::max(v_max, (size_t)36 * L.th * L.tw * cv.cin);
            m_max = std::max(m_max, (size_t)36 * L.th * L.tw * cv.cout);
        } else if (fused_on && tdim_ok && l > 0 && cv.stride == 1 && cv.cin % 16 == 0 && cv.cin >= 64 && cv.cin <= fused_max_cin && cv.cout % WF_BN == 0) {
            std::vector<float> u((size_t)16 * cv.cin * cv.cout);
            wf_pack_u(cv.w.data(), cv.cin, cv.cout, u.data());
            if ((rc = dev_upload(e, &L.d_u, u))) return rc;
            const char *wv = dev_getenv("APR_WINO_FUSED_WAVES");   // developer knob: 4 = the four-wave kernel
            L.wino = 3;
            // unit order (wf_unit): N tiles in groups of four.  HBM fetch of a 2048-window pass, all N tiles of an M tile side by
            // side -> groups of four: layer 7 7.9 -> 4.9 GB, layer 9 3.7 -> 2.4, layer 11 3.3 -> 1.2 (groups of two: the same;
            // of one: 7.3 / 3.6 / 1.8 and twice the input on the layers of few N tiles); the times do not move (tools/gn_sweep.sh)
            L.gn = dev_getenv("APR_WINO_GN") ? atoi(dev_getenv("APR_WINO_GN")) : 4;
            L.fused_waves = wv && atoi(wv) == 4 ? 4 : 8;
            L.ksplit = 1;
            L.th = (L.hout + 1) / 2;
            L.tw = (L.wout + 1) / 2;
        } else if (wino_on && l > 0 && cv.stride == 1 && cv.cin >= wino_min_cin && cv.cout % 128 == 0) {
            // U[pos = 4 i + j][cin][cout] = (G g G^T)[i][j], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], in f64
            static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            std::vector<float> u((size_t)16 * cv.cin * cv.cout);
            for (int o = 0; o < cv.cout; o++)
                for (int c = 0; c < cv.cin; c++) {
                    const float *g = &cv.w[((size_t)o * cv.cin + c) * 9];
                    double t[4][3];
                    for (int i = 0; i < 4; i++)
                        for (int k = 0; k < 3; k++) t[i][k] = G[i][0] * g[k] + G[i][1] * g[3 + k] + G[i][2] * g[6 + k];
                    for (int i = 0; i < 4; i++)
                        for (int j = 0; j < 4; j++) {
                            const double v = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                            u[((size_t)(4 * i + j) * cv.cin + c) * cv.cout + o] = (float)v;
                        }
                }
            if ((rc = dev_upload(e, &L.d_u, u))) return rc;
            L.wino = 2;
            L.ksplit = 1;
            L.th = (L.hout + 1) / 2;
            L.tw = (L.wout + 1) / 2;
            const long long t_full = (long long)e->max_batch * L.th * L.tw;
            L.wbm_tile = ((t_full + 127) / 128) * (cv.cout / 128) * 16 < 1024 ? 64 : 128;
            // one-clip engines: a few dozen workgroups with Cin / 16 K steps each; split K (>= 4 steps a split) towards 512
            if (splitk_on && e->max_batch <= 4) {
                const long long blocks = ((t_full + L.wbm_tile - 1) / L.wbm_tile) * (cv.cout / 128) * 16;
                const char *wsenv = dev_getenv("APR_WSPLIT_MAX");   // developer knob
                const long long wsmax = wsenv ? atoi(wsenv) : 4;
                L.wsplit = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(512 / std::max<long long>(blocks, 1), cv.cin / 16 / 4), wsmax));
            }
            v_max = std::max(v_max, (size_t)16 * L.th * L.tw * cv.cin);
            m_max = std::max(m_max, (size_t)L.wsplit * 16 * L.th * L.tw * cv.cout);
        }
        // the one-kernel form (wino_s2_fused.hip.h): every stride-2 layer from Cin = 128 on in engines of >= 128 windows (a step of
        // 2048-window passes: layers 4 / 6 / 8 / 10 / 12 = 103.8 / 95.9 / 91.9 / 23.0 / 29.1 ms against 117.2 direct, 104.8 / 94.9 / 24.1 as
        // nine-sum GEMMs + output transform, 30.6 direct; 256 live streams: 8.8 against 9.0 ms a tick);
        // APR_WINO_S2_FUSED=0 turns it off, APR_WINO_S2_FUSED_MIN_CIN / _MIN_PX / _MIN_BATCH move its bounds (developer knobs, tests)
        {
            const char *sfenv = dev_getenv("APR_WINO_S2_FUSED");
            const bool s2f_on = wino_s2_on && !(sfenv && sfenv[0] == '0');
            const int s2f_min_cin = dev_getenv("APR_WINO_S2_FUSED_MIN_CIN") ? atoi(dev_getenv("APR_WINO_S2_FUSED_MIN_CIN")) : 128;
            const int s2f_min_px = dev_getenv("APR_WINO_S2_FUSED_MIN_PX") ? atoi(dev_getenv("APR_WINO_S2_FUSED_MIN_PX")) : 1;
            const int s2f_min_batch = dev_getenv("APR_WINO_S2_FUSED_MIN_BATCH") ? atoi(dev_getenv("APR_WINO_S2_FUSED_MIN_BATCH")) : 128;
            if (!L.wino && s2f_on && tdim_ok && l > 0 && cv.stride == 2 && cv.cin >= s2f_min_cin && cv.cin % 32 == 0 && cv.cout % WS_BN == 0 &&
                L.hout * L.wout >= s2f_min_px && e->max_batch >= s2f_min_batch) {
                std::vector<float> u((size_t)16 * cv.cin * cv.cout);
                ws_pack_u(cv.w.data(), cv.cin, cv.cout, u.data());
                if ((rc = dev_upload(e, &L.d_u, u))) return rc;
                L.wino = 5;
                L.gn = dev_getenv("APR_WINO_S2_GN") ? atoi(dev_getenv("APR_WINO_S2_GN")) : 4;   // layer 12: 4.5 -> 1.9 GB, layer 10: 2.0 -> 1.6
                L.ksplit = 1;
                L.th = (L.hout + 1) / 2;
                L.tw = (L.wout + 1) / 2;
            }
        }
        if (!L.wino && wino_s2_on && l > 0 && cv.stride == 2 && cv.cin >= wino_s2_min_cin && cv.cin % 16 == 0 && cv.cout % 128 == 0 &&
            L.hout * L.wout >= wino_s2_min_px && e->max_batch >= wino_s2_min_batch) {
            std::vector<float> u((size_t)16 * cv.cin * cv.cout);
            wino_s2_pack_u(cv.w.data(), cv.cin, cv.cout, u.data());
            if ((rc = dev_upload(e, &L.d_u, u))) return rc;
            L.wino = 4;
            L.ksplit = 1;
            L.th = (L.hout + 1) / 2;
            L.tw = (L.wout + 1) / 2;
            const long long t_full = (long long)e->max_batch * L.th * L.tw;
            L.wbm_tile = ((t_full + 127) / 128) * (cv.cout / 128) * 9 < 1024 ? 64 : 128;
            m_max = std::max(m_max, (size_t)9 * L.th * L.tw * cv.cout);
        }
        act_max = std::max(act_max, (size_t)L.cout * L.hout * L.wout);
        h = L.hout; w = L.wout;
        e->layers.push_back(L);
    }
    // layers 1 + 2 in one kernel when they are the reference's (1 -> 32 stride 1, 32 -> 64 stride 2); APR_FUSE12=0: two kernels
    // (latency engines keep the two kernels: 48 tiles of one clip leave most CUs idle and measured 20 us slower)
    if (e->layers.size() >= 2 && e->max_batch > 4 && !(dev_getenv("APR_FUSE12") && atoi(dev_getenv("APR_FUSE12")) == 0)) {
        const Layer &a = e->layers[0];
        Layer &b = e->layers[1];
        if (a.cin == 1 && a.cout == 32 && a.stride == 1 && b.cin == 32 && b.cout == 64 && b.stride == 2 && !b.wino) {
            e->fuse12 = true;
            b.ksplit = 1;
        }
    }
    e->n_classes = pm.n_classes;
    e->features = pm.features;
    e->pool_pixels = h * w;
    if ((rc = dev_upload(e, &e->d_wfc, pm.fc_w)) || (rc = dev_upload(e, &e->d_bfc, pm.fc_b))) return rc;
    for (int i = 0; i < 2; i++)
        if ((rc = dev_alloc(e, &e->d_act[i], act_max * (size_t)e->max_batch))) return rc;
    if (part_max && (rc = dev_alloc(e, &e->d_part, part_max))) return rc;
    if (v_max && (rc = dev_alloc(e, &e->d_V, v_max * (size_t)e->max_batch + v_pad))) return rc;
    if (m_max && (rc = dev_alloc(e, &e->d_M, m_max * (size_t)e->max_batch))) return rc;
    return APR_OK;
}

// ---- profiling --------------------------------------------------------------------------------
hipEvent_t get_event(apr_engine *e)
{
    if (!e->free_events.empty()) { hipEvent_t ev = e->free_events.back(); e->free_events.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

struct ProfScope {
    apr_engine *e;
    hipEvent_t a = nullptr, b = nullptr;
    int slot;
    ProfScope(apr_engine *e_, int slot_, double flops, double bytes) : e(e_), slot(slot_)
    {
        if (!e->prof_on || slot < 0) return;
        a = get_event(e); b = get_event(e);
        if (!a || !b) { a = b = nullptr; return; }
        (void)hipEventRecord(a, e->stream);
        e->slots[slot].flops += flops;
        e->slots[slot].bytes += bytes;
        e->slots[slot].launches += 1;
    }
    ~ProfScope()
    {
        if (!a) return;
        (void)hipEventRecord(b, e->stream);
        e->pending.push_back({a, b, slot});
    }
};

int prof_flush(apr_engine *e)
{
    if (e->pending.empty()) return APR_OK;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (auto &p : e->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) e->slots[p.slot].ms += (double)ms;
        e->free_events.push_back(p.a);
        e->free_events.push_back(p.b);
    }
    e->pending.clear();
    return APR_OK;
}

// ---- launches ---------------------------------------------------------------------------------
// melT: the mel-power scratch of this launch pair (the engine's own d_melT, or the static array of the allocation-free entry)
int run_frontend(apr_engine *e, int flavour, const uint8_t *d_pcm, size_t total_bytes, int n_windows, size_t stride,
                 float *d_out, float *melT = nullptr, bool timed = true)
{
    if (!melT) melT = e->d_melT;
    const FrontendTables &T = e->tab[flavour];
    // a handful of windows (live clips): one frame per wave and, in db_norm, one block per 32-frame column chunk instead
    // of one per window, so the latency is a fraction; the arithmetic per element is the same code
    const bool few = n_windows <= 8;
    const int groups = (T.n_frames + kFramesPerBlock - 1) / kFramesPerBlock;
    {
        ProfScope ps(e, timed ? 0 : -1, (double)n_windows * T.n_frames * 5.0 * kNfft * 11.0 + 2.0 * T.nnz * T.n_frames * n_windows,
                     (double)n_windows * (std::min<size_t>(stride, APR_SAMPLE_SIZE) + (double)T.n_frames * kNmels * 4));
        hipLaunchKernelGGL(stft_mel_kernel, dim3(n_windows * groups), dim3(64 * kFramesPerBlock), 0, e->stream, d_pcm, total_bytes, stride, n_windows, T, melT);
    }
    {
        ProfScope ps(e, timed ? 1 : -1, 0.0, (double)n_windows * ((double)T.n_frames * kNmels * 4 + APR_SPEC_ELEMS * 4));
        hipLaunchKernelGGL(db_norm_kernel, dim3(n_windows, few ? 6 : 1), dim3(256), 0, e->stream, melT, n_windows, T.n_frames,
                           T.librosa, d_out);
    }
    HIPCHK(e, hipGetLastError());
    return APR_OK;
}

template <int BM, int BN, int S>
void launch_igemm(apr_engine *e, const Layer &L, const float *in, float *out, long long M)
{
    const int tiles_n = L.cout / BN;
    const long long tiles_m = (M + BM - 1) / BM;   // M < 2^31: max_batch is capped at create
    hipLaunchKernelGGL((conv3x3_igemm_f32<BM, BN, S>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, e->stream, in,
                       L.d_w, L.d_b, out, L.hin, L.win, L.cin, L.hout, L.wout, L.cout, (int)M, tiles_n);
}

template <int BM, int BN, int S>
void launch_igemm_splitk(apr_engine *e, const Layer &L, const float *in, long long M)
{
    const int tiles_n = L.cout / BN;
    const long long tiles_m = (M + BM - 1) / BM;
    hipLaunchKernelGGL((conv3x3_igemm_f32<BM, BN, S, 2>), dim3((unsigned)(tiles_m * tiles_n), (unsigned)L.ksplit), dim3(256), 0, e->stream, in,
                       L.d_w, L.d_b, e->d_part, L.hin, L.win, L.cin, L.hout, L.wout, L.cout, (int)M, tiles_n);
}

template <int BM>
void launch_wino_gemm(apr_engine *e, const Layer &L, long long T)
{
    const int tiles_n = L.cout / 128;
    const long long tiles_m = (T + BM - 1) / BM;
    hipLaunchKernelGGL((conv3x3_igemm_f32<BM, 128, 1, 1>), dim3((unsigned)(tiles_m * tiles_n), 16, (unsigned)L.wsplit), dim3(256), 0, e->stream, e->d_V,
                       L.d_u, L.d_b, e->d_M, 1, 1, L.cin, 1, 1, L.cout, (int)T, tiles_n);
}

int run_cnn(apr_engine *e, const float *d_spec, int batch, double trust, int empty_class, float *d_logits, double *d_probs,
            int32_t *d_label, uint8_t *d_match)
{
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    const float *in = d_spec;
    for (size_t l = 0; l < e->layers.size(); l++) {
        const Layer &L = e->layers[l];
        float *out = e->d_act[l & 1];
        const long long M = (long long)batch * L.hout * L.wout;
        const double flops = 2.0 * (double)M * L.cout * 9.0 * L.cin;
        const double bytes = 4.0 * ((double)batch * L.hin * L.win * L.cin + (double)M * L.cout + 9.0 * L.cin * L.cout);
        if (e->fuse12 && l == 0) continue;   // computed inside layer 2's kernel
        if (e->fuse12 && l == 1) {
            const int tiles_x = (L.wout + C12_TW - 1) / C12_TW, tiles_y = (L.hout + C12_TH - 1) / C12_TH;
            const Layer &L0 = e->layers[0];
            ProfScope ps(e, L.slot, flops + 2.0 * 9.0 * 32.0 * (double)batch * L.hin * L.win,
                         4.0 * ((double)batch * L.hin * L.win + (double)M * L.cout + 9.0 * L.cin * L.cout));
            const unsigned n_tiles = (unsigned)batch * (unsigned)(tiles_x * tiles_y);   // <= 16384 * 48
            hipLaunchKernelGGL(conv12_fused_kernel, dim3(std::min(n_tiles, 2u * (unsigned)e->n_cus)), dim3(256), C12_LDS_BYTES, e->stream, d_spec,
                               L0.d_w, L0.d_b, L.d_w, L.d_b, out, L.hin, L.win, L.hout, L.wout, tiles_x, tiles_y, n_tiles);
            in = out;
            continue;
        }
        if (L.wino == 3) {
            const long long T = (long long)batch * L.th * L.tw;   // < 2^31: max_batch is capped at create
            const int tiles_n = L.cout / WF_BN;
            const long long tiles_m = (T + WF_TT - 1) / WF_TT;
            ProfScope ps(e, L.slot, 2.0 * 16.0 * (double)T * L.cin * L.cout, bytes + 4.0 * 7.0 * L.cin * L.cout);
            if (L.fused_waves == 4)
                hipLaunchKernelGGL(wino_fused_f2, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), WF_LDS_BYTES, e->stream, in, L.d_u, L.d_b,
                                   out, L.hin, L.win, L.cin, L.cout, L.th, L.tw, (int)T, tiles_n, wf_magic(tiles_n, L.th, L.tw, tiles_m));
            else
                hipLaunchKernelGGL(wino_fused8_f2, dim3(wf8_grid((int)(tiles_m * tiles_n), e->n_cus)), dim3(512), WF8_LDS_BYTES, e->stream, in, L.d_u,
                                   L.d_b, out, L.hin, L.win, L.cin, L.cout, L.th, L.tw, (int)T, tiles_n, wf_magic(tiles_n, L.th, L.tw, tiles_m, L.gn), (int)(tiles_m * tiles_n));
            in = out;
            continue;
        }
        if (L.wino == 5) {
            const long long T = (long long)batch * L.th * L.tw;   // < 2^31: max_batch is capped at create
            const int tiles_n = L.cout / WS_BN;
            const long long tiles_m = (T + WS_TT - 1) / WS_TT;
            ProfScope ps(e, L.slot, 2.0 * 25.0 * (double)T * L.cin * L.cout, bytes + 4.0 * 7.0 * L.cin * L.cout);
            WsArgs wa;
            wa.in = in; wa.up = L.d_u; wa.bias = L.d_b; wa.out = out;
            wa.H = L.hin; wa.W = L.win; wa.Cin = L.cin; wa.Ho = L.hout; wa.Wo = L.wout; wa.Cout = L.cout;
            wa.th = L.th; wa.tw = L.tw; wa.T = (int)T; wa.tiles_n = tiles_n;
            wa.mg = wf_magic(tiles_n, L.th, L.tw, tiles_m, L.gn);
            const int n_units = (int)(tiles_m * tiles_n);
            hipLaunchKernelGGL(wino_s2_fused, dim3(wf8_grid(n_units, e->n_cus)), dim3(512), WS_LDS_BYTES, e->stream, wa, n_units);
            in = out;
            continue;
        }
        if (L.wino == 4) {
            const long long T = (long long)batch * L.th * L.tw;
            const long long nm = T * (L.cout / 4);
            {
                ProfScope ps(e, L.slot, 2.0 * 25.0 * (double)T * L.cin * L.cout,
                             4.0 * ((double)batch * L.hin * L.win * L.cin + 9.0 * (double)T * L.cout + 16.0 * L.cin * L.cout));
                const int tiles_n = L.cout / 128;
                if (L.wbm_tile == 64)
                    hipLaunchKernelGGL((conv3x3_igemm_f32<64, 128, 1, 3>), dim3((unsigned)(((T + 63) / 64) * tiles_n), 9), dim3(256), 0, e->stream, in,
                                       L.d_u, L.d_b, e->d_M, L.hin, L.win, L.cin, L.th, L.tw, L.cout, (int)T, tiles_n);
                else
                    hipLaunchKernelGGL((conv3x3_igemm_f32<128, 128, 1, 3>), dim3((unsigned)(((T + 127) / 128) * tiles_n), 9), dim3(256), 0, e->stream, in,
                                       L.d_u, L.d_b, e->d_M, L.hin, L.win, L.cin, L.th, L.tw, L.cout, (int)T, tiles_n);
            }
            {
                ProfScope ps(e, L.slot_out, 0.0, 4.0 * (9.0 * T * L.cout + (double)M * L.cout));
                hipLaunchKernelGGL(wino_s2_output, dim3((unsigned)std::min<long long>((nm + 255) / 256, 65536)), dim3(256), 0, e->stream,
                                   e->d_M, L.d_b, out, L.hout, L.wout, L.cout, L.th, L.tw, T);
            }
            in = out;
            continue;
        }
        if (L.wino == 7) {
            const long long T = (long long)batch * L.th * L.tw;
            {
                const long long n_blocks = ((T + W4_TT - 1) / W4_TT) * (L.cin / W4_BK);      // < 2^31: max_batch is capped at create
                ProfScope ps(e, L.slot_in, 0.0, 4.0 * ((double)batch * L.hin * L.win * L.cin + 36.0 * T * L.cin));
                if (36.0 * 4.0 * (double)T * L.cin > 2.0e9)
                    hipLaunchKernelGGL(wino_input_f4p<true>, dim3((unsigned)n_blocks), dim3(256), 0, e->stream, in, e->d_V, L.hin, L.win, L.cin, L.th, L.tw, T, (int)n_blocks);
                else
                    hipLaunchKernelGGL(wino_input_f4p<false>, dim3((unsigned)n_blocks), dim3(256), 0, e->stream, in, e->d_V, L.hin, L.win, L.cin, L.th, L.tw, T, (int)n_blocks);
            }
            {
                const int tiles_n = L.cout / W4_BN;
                const long long tiles_m = (T + W4_TT - 1) / W4_TT;
                ProfScope ps(e, L.slot, 2.0 * 36.0 * (double)T * L.cin * L.cout, 4.0 * (36.0 * T * L.cin + (double)M * L.cout + 36.0 * L.cin * L.cout));
                W4Args wa;
                wa.vp = e->d_V; wa.up = L.d_u; wa.bias = L.d_b; wa.out = out;
                wa.H = L.hout; wa.W = L.wout; wa.Cin = L.cin; wa.Cout = L.cout; wa.th = L.th; wa.tw = L.tw; wa.T = (int)T; wa.tiles_n = tiles_n;
                wa.mg = wf_magic(tiles_n, L.th, L.tw, tiles_m, L.gn);
                wa.n_units = (int)(tiles_m * tiles_n);
                hipLaunchKernelGGL(wino4_fused_out, dim3(wf8_grid(wa.n_units, e->n_cus)), dim3(512), W4_LDS_BYTES, e->stream, wa);
            }
            in = out;
            continue;
        }
        if (L.wino == 6) {
            const long long T = (long long)batch * L.th * L.tw;
            const long long nv = T * (L.cin / 4), nm = T * (L.cout / 4);
            {
                ProfScope ps(e, L.slot_in, 0.0, 4.0 * ((double)batch * L.hin * L.win * L.cin + 36.0 * T * L.cin));
                hipLaunchKernelGGL(wino_input_f4, dim3((unsigned)std::min<long long>((nv + 255) / 256, 65536)), dim3(256), 0, e->stream, in,
                                   e->d_V, L.hin, L.win, L.cin, L.th, L.tw, T);
            }
            {
                ProfScope ps(e, L.slot, 2.0 * 36.0 * (double)T * L.cin * L.cout, 4.0 * 36.0 * ((double)T * (L.cin + L.cout) + (double)L.cin * L.cout));
                const int tiles_n = L.cout / 128;
                if (L.wbm_tile == 64)
                    hipLaunchKernelGGL((conv3x3_igemm_f32<64, 128, 1, 1>), dim3((unsigned)(((T + 63) / 64) * tiles_n), 36, 1), dim3(256), 0, e->stream, e->d_V,
                                       L.d_u, L.d_b, e->d_M, 1, 1, L.cin, 1, 1, L.cout, (int)T, tiles_n);
                else
                    hipLaunchKernelGGL((conv3x3_igemm_f32<128, 128, 1, 1>), dim3((unsigned)(((T + 127) / 128) * tiles_n), 36, 1), dim3(256), 0, e->stream, e->d_V,
                                       L.d_u, L.d_b, e->d_M, 1, 1, L.cin, 1, 1, L.cout, (int)T, tiles_n);
            }
            {
                ProfScope ps(e, L.slot_out, 0.0, 4.0 * (36.0 * T * L.cout + (double)M * L.cout));
                hipLaunchKernelGGL(wino_output_f4, dim3((unsigned)std::min<long long>((nm + 255) / 256, 65536)), dim3(256), 0, e->stream,
                                   e->d_M, L.d_b, out, L.hout, L.wout, L.cout, L.th, L.tw, T);
            }
            in = out;
            continue;
        }
        if (L.wino) {
            const long long T = (long long)batch * L.th * L.tw;
            const long long nv = T * (L.cin / 4), nm = T * (L.cout / 4);
            {
                ProfScope ps(e, L.slot_in, 0.0, 4.0 * ((double)batch * L.hin * L.win * L.cin + 16.0 * T * L.cin));
                hipLaunchKernelGGL(wino_input_f2, dim3((unsigned)std::min<long long>((nv + 255) / 256, 65536)), dim3(256), 0, e->stream, in,
                                   e->d_V, L.hin, L.win, L.cin, L.th, L.tw, T);
            }
            {
                ProfScope ps(e, L.slot, 2.0 * 16.0 * (double)T * L.cin * L.cout, 4.0 * 16.0 * ((double)T * (L.cin + L.cout) + (double)L.cin * L.cout));
                if (L.wbm_tile == 64) launch_wino_gemm<64>(e, L, T);
                else launch_wino_gemm<128>(e, L, T);
            }
            {
                ProfScope ps(e, L.slot_out, 0.0, 4.0 * (16.0 * T * L.cout + (double)M * L.cout));
                hipLaunchKernelGGL(wino_output_f2, dim3((unsigned)std::min<long long>((nm + 255) / 256, 65536)), dim3(256), 0, e->stream,
                                   e->d_M, L.d_b, out, L.hout, L.wout, L.cout, L.th, L.tw, T, L.wsplit);
            }
            in = out;
            continue;
        }
        if (L.ksplit > 1) {
            {
                ProfScope ps(e, L.slot, flops, bytes + 4.0 * L.ksplit * (double)M * L.cout);
                if (L.bn_tile == 128) {
                    if (L.bm_tile == 64) { if (L.stride == 1) launch_igemm_splitk<64, 128, 1>(e, L, in, M); else launch_igemm_splitk<64, 128, 2>(e, L, in, M); }
                    else { if (L.stride == 1) launch_igemm_splitk<128, 128, 1>(e, L, in, M); else launch_igemm_splitk<128, 128, 2>(e, L, in, M); }
                } else {
                    if (L.stride == 1) launch_igemm_splitk<128, 64, 1>(e, L, in, M); else launch_igemm_splitk<128, 64, 2>(e, L, in, M);
                }
            }
            {
                const long long n4 = M * L.cout / 4;
                ProfScope ps(e, L.slot_out, 0.0, 4.0 * (L.ksplit + 1.0) * (double)M * L.cout);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 4096)), dim3(256), 0, e->stream, e->d_part,
                                   L.d_b, out, n4, L.cout / 4, L.ksplit);
            }
            in = out;
            continue;
        }
        ProfScope ps(e, L.slot, flops, bytes);
        if (l == 0) {
            const unsigned xchunks = (unsigned)(L.win + 3) / 4, n_rows = (unsigned)(M / L.win);   // M = windows * H * W < 2^31 (max_batch cap)
            const long long blocks = std::min<long long>(((long long)n_rows * xchunks + 31) / 32, 256 * 32);
            hipLaunchKernelGGL(conv1_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, in, L.d_w, L.d_b, out, L.hin,
                               L.win, n_rows, xchunks);
        } else if (L.bn_tile == 128) {
            if (L.bm_tile == 64) {
                if (L.stride == 1) launch_igemm<64, 128, 1>(e, L, in, out, M);
                else launch_igemm<64, 128, 2>(e, L, in, out, M);
            } else {
                if (L.stride == 1) launch_igemm<128, 128, 1>(e, L, in, out, M);
                else launch_igemm<128, 128, 2>(e, L, in, out, M);
            }
        } else {
            if (L.stride == 1) launch_igemm<128, 64, 1>(e, L, in, out, M);
            else launch_igemm<128, 64, 2>(e, L, in, out, M);
        }
        in = out;
    }
    {
        ProfScope ps(e, e->head_slot, 2.0 * batch * e->features * e->n_classes,
                     4.0 * batch * e->pool_pixels * e->features);
        hipLaunchKernelGGL(head_kernel, dim3(batch), dim3(256), 0, e->stream, in, e->pool_pixels, e->features, e->d_wfc,
                           e->d_bfc, e->n_classes, trust, empty_class, d_logits, d_probs, d_label, d_match);
    }
    HIPCHK(e, hipGetLastError());
    return APR_OK;
}

int ensure_staging(apr_engine *e, size_t pcm_bytes)
{
    int rc;
    if (pcm_bytes > e->pcm_cap) {
        size_t cap = std::max(pcm_bytes, (size_t)e->max_batch * APR_BYTES_PER_SECOND + APR_SAMPLE_SIZE);
        if ((rc = dev_alloc(e, &e->d_pcm, cap + 16))) return rc;   // old buffer stays in allocs until destroy
        e->pcm_cap = cap;
    }
    if (!e->d_probs) {
        const size_t c = std::max(e->n_classes, 1);
        if ((rc = dev_alloc(e, &e->d_probs, (size_t)e->max_batch * c)) || (rc = dev_alloc(e, &e->d_logits, (size_t)e->max_batch * c)) ||
            (rc = dev_alloc(e, &e->d_label, (size_t)e->max_batch)) || (rc = dev_alloc(e, &e->d_match, (size_t)e->max_batch)))
            return rc;
    }
    return APR_OK;
}

// bytes of `pcm` that windows [w0, w0+nw) touch, relative to window w0's start
size_t span_bytes(size_t total_bytes, size_t w0, size_t nw, size_t stride)
{
    const size_t off = w0 * stride;
    if (off >= total_bytes || nw == 0) return 0;
    const size_t want = (nw - 1) * stride + APR_SAMPLE_SIZE;
    return std::min(want, total_bytes - off);
}

constexpr int kGraphUnavailable = -1000;   // internal: never returned through the C ABI

// Replay (capturing first if the key changed) the stream work that record() issues.  record() may only enqueue work on
// e->stream (no allocation, no synchronisation) and returns false on any error.  kGraphUnavailable = could not capture:
// the caller falls back to launching kernel by kernel.
template <class F>
int graph_replay(apr_engine *e, GraphSlot &s, size_t k0, size_t k1, size_t k2, double kd, int ki, F record)
{
    if (!s.exec || s.k0 != k0 || s.k1 != k1 || s.k2 != k2 || s.kd != kd || s.ki != ki) {
        if (s.exec) { (void)hipGraphExecDestroy(s.exec); s.exec = nullptr; }
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return kGraphUnavailable; }
        const bool ok = record();
        hipGraph_t g = nullptr;
        const bool ended = hipStreamEndCapture(e->stream, &g) == hipSuccess && g;
        if (ok && ended && hipGraphInstantiate(&s.exec, g, nullptr, nullptr, 0) != hipSuccess) s.exec = nullptr;
        if (g) (void)hipGraphDestroy(g);
        if (!ok || !ended || !s.exec) { s.exec = nullptr; (void)hipGetLastError(); return kGraphUnavailable; }
        s.k0 = k0; s.k1 = k1; s.k2 = k2; s.kd = kd; s.ki = ki;
    }
    HIPCHK(e, hipGraphLaunch(s.exec, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return APR_OK;
}

// pinned staging of the replayed calls (fixed size: the graphs hold its addresses)
int graph_staging(apr_engine *e)
{
    if (e->h_g_in && e->h_g_out) return APR_OK;
    if ((!e->h_g_in && hipHostMalloc((void **)&e->h_g_in, kGraphIn, hipHostMallocDefault) != hipSuccess) ||
        (!e->h_g_out && hipHostMalloc((void **)&e->h_g_out, kGraphOut, hipHostMallocDefault) != hipSuccess)) {
        (void)hipGetLastError();
        return kGraphUnavailable;
    }
    return APR_OK;
}

// may this call be replayed?  (profiling needs its events between the kernels)
bool graph_ok(const apr_engine *e, size_t n, size_t in_bytes)
{
    return e->graph_on && !e->prof_on && n >= 1 && n <= (size_t)std::min(e->max_batch, kGraphWindows) && in_bytes <= kGraphIn;
}

// apr_score as one graph launch
int score_graph(apr_engine *e, const uint8_t *pcm, size_t span, size_t nw, size_t stride, double trust, int empty_class,
                double *probs, int32_t *label, uint8_t *match)
{
    const int C = e->n_classes;
    int rc;
    if ((rc = ensure_staging(e, span)) || (rc = graph_staging(e))) return rc;
    double *hp = reinterpret_cast<double *>(e->h_g_out);                       // [nw][C] <= 4 * 64 doubles
    int32_t *hl = reinterpret_cast<int32_t *>(e->h_g_out + 4096);
    uint8_t *hm = e->h_g_out + 4096 + 64;
    if (span) memcpy(e->h_g_in, pcm, span);
    rc = graph_replay(e, e->g_score, nw, span, stride, trust, empty_class, [&]() {
        bool ok = true;
        if (span) ok = hipMemcpyAsync(e->d_pcm, e->h_g_in, span, hipMemcpyHostToDevice, e->stream) == hipSuccess;
        ok = ok && run_frontend(e, APR_FLAVOUR_GO, e->d_pcm, span, (int)nw, stride, e->d_spec) == APR_OK;
        ok = ok && run_cnn(e, e->d_spec, (int)nw, trust, empty_class, nullptr, e->d_probs, e->d_label, e->d_match) == APR_OK;
        ok = ok && hipMemcpyAsync(hp, e->d_probs, nw * C * sizeof(double), hipMemcpyDeviceToHost, e->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(hl, e->d_label, nw * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream) == hipSuccess;
        return ok && hipMemcpyAsync(hm, e->d_match, nw, hipMemcpyDeviceToHost, e->stream) == hipSuccess;
    });
    if (rc) return rc;
    if (probs) memcpy(probs, hp, nw * C * sizeof(double));
    if (label) memcpy(label, hl, nw * sizeof(int32_t));
    if (match) memcpy(match, hm, nw);
    return APR_OK;
}

// apr_infer as one graph launch
int infer_graph(apr_engine *e, const float *spec, int nb, float *out_logits)
{
    int rc;
    if ((rc = ensure_staging(e, 0)) || (rc = graph_staging(e))) return rc;
    const size_t in_bytes = (size_t)nb * APR_SPEC_ELEMS * sizeof(float), out_bytes = (size_t)nb * e->n_classes * sizeof(float);
    memcpy(e->h_g_in, spec, in_bytes);
    rc = graph_replay(e, e->g_infer, (size_t)nb, 0, 0, 0.0, 0, [&]() {
        bool ok = hipMemcpyAsync(e->d_spec, e->h_g_in, in_bytes, hipMemcpyHostToDevice, e->stream) == hipSuccess;
        ok = ok && run_cnn(e, e->d_spec, nb, 0.0, -1, e->d_logits, nullptr, nullptr, nullptr) == APR_OK;
        return ok && hipMemcpyAsync(e->h_g_out, e->d_logits, out_bytes, hipMemcpyDeviceToHost, e->stream) == hipSuccess;
    });
    if (rc) return rc;
    memcpy(out_logits, e->h_g_out, out_bytes);
    return APR_OK;
}

// apr_prepare_batch as one graph launch
int prepare_graph(apr_engine *e, int flavour, const uint8_t *pcm, size_t span, size_t nw, size_t stride, float *out)
{
    int rc;
    if ((rc = ensure_staging(e, span)) || (rc = graph_staging(e))) return rc;
    const size_t out_bytes = nw * APR_SPEC_ELEMS * sizeof(float);
    if (span) memcpy(e->h_g_in, pcm, span);
    rc = graph_replay(e, e->g_prep[flavour], nw, span, stride, 0.0, 0, [&]() {
        bool ok = true;
        if (span) ok = hipMemcpyAsync(e->d_pcm, e->h_g_in, span, hipMemcpyHostToDevice, e->stream) == hipSuccess;
        ok = ok && run_frontend(e, flavour, e->d_pcm, span, (int)nw, stride, e->d_spec) == APR_OK;
        return ok && hipMemcpyAsync(e->h_g_out, e->d_spec, out_bytes, hipMemcpyDeviceToHost, e->stream) == hipSuccess;
    });
    if (rc) return rc;
    memcpy(out, e->h_g_out, out_bytes);
    return APR_OK;
}

int check_engine(apr_engine *e)
{
    if (!e) return APR_ERR_INVALID;
    if (hipSetDevice(e->device) != hipSuccess) { e->err = "hipSetDevice failed"; return APR_ERR_DEVICE; }
    return APR_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int apr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *apr_version(void) { return "apr_amd 0.1 (gfx950)"; }

int apr_model_info(const void *model_bytes, size_t n, int *n_classes, int *n_conv_layers)
{
    ParsedModel pm;
    std::string perr;
    if (!model_bytes || !n) { g_create_err = "model: empty"; return APR_ERR_MODEL; }
    if (!parse_model(model_bytes, n, pm, perr)) { g_create_err = "model: " + perr; return APR_ERR_MODEL; }
    if (n_classes) *n_classes = pm.n_classes;
    if (n_conv_layers) *n_conv_layers = (int)pm.convs.size();
    return APR_OK;
}

static int engine_create(const void *model_bytes, size_t n, int device_id, int max_batch, apr_engine **out);

int apr_engine_create(const void *model_bytes, size_t n, int device_id, int max_batch, apr_engine **out)
{
    try {
        return engine_create(model_bytes, n, device_id, max_batch, out);
    } catch (const std::exception &ex) {   // host allocation failure in a std container: no exception crosses the C ABI
        g_create_err = std::string("out of host memory: ") + ex.what();
        if (out) *out = nullptr;
        return APR_ERR_NOMEM;
    }
}

int apr_engine_create_ex(const void *model_bytes, size_t n, int device_id, int max_batch, apr_engine **out, char *errbuf, size_t errbuf_len)
{
    const int rc = apr_engine_create(model_bytes, n, device_id, max_batch, out);
    if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", rc ? g_create_err.c_str() : "");
    return rc;
}

static int engine_create(const void *model_bytes, size_t n, int device_id, int max_batch, apr_engine **out)
{
    if (!out) return APR_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_err = "no HIP device available"; return APR_ERR_DEVICE; }
    if (device_id < 0 || device_id >= ndev) { g_create_err = "device_id out of range"; return APR_ERR_INVALID; }
    ParsedModel pm;
    const bool has_model = model_bytes && n;
    if (has_model) {
        std::string perr;
        if (!parse_model(model_bytes, n, pm, perr)) { g_create_err = "model: " + perr; return APR_ERR_MODEL; }
    }
    apr_engine *e = new (std::nothrow) apr_engine();
    if (!e) { g_create_err = "out of host memory"; return APR_ERR_NOMEM; }
    // released on every way out but success, a std::bad_alloc from the table / weight containers below included
    struct Guard { apr_engine *e; ~Guard() { if (e) apr_engine_destroy(e); } } guard{e};
    e->device = device_id;
    e->max_batch = max_batch > 0 ? std::min(max_batch, 16384) : 1024;   // keeps every layer's pixel count < 2^31
    add_slot(e, "stft_mel_kernel", -1);
    add_slot(e, "db_norm_kernel", -1);
    int rc = APR_OK;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        e->err = "could not create a stream on the device";
        rc = APR_ERR_DEVICE;
    }
    if (!rc && (hipFuncSetAttribute(reinterpret_cast<const void *>(wino_fused_f2), hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(wino_fused8_f2), hipFuncAttributeMaxDynamicSharedMemorySize, WF8_LDS_BYTES) != hipSuccess)) {
        e->err = "could not reserve LDS for wino_fused_f2";
        rc = APR_ERR_DEVICE;
    }
    if (!rc && (hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || e->n_cus < 1)) e->n_cus = 256;
    if (!rc && !resolve_scratch(device_id, e->err)) rc = APR_ERR_DEVICE;   // this device's address of the static scratch array
    if (!rc && hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_fused_out), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES) != hipSuccess) {
        e->err = "could not reserve LDS for wino4_fused_out";
        rc = APR_ERR_DEVICE;
    }
    if (!rc && hipFuncSetAttribute(reinterpret_cast<const void *>(wino_s2_fused), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) != hipSuccess) {
        e->err = "could not reserve LDS for wino_s2_fused";
        rc = APR_ERR_DEVICE;
    }
    if (!rc && hipFuncSetAttribute(reinterpret_cast<const void *>(conv12_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C12_LDS_BYTES) != hipSuccess) {
        e->err = "could not reserve LDS for conv12_fused_kernel";
        rc = APR_ERR_DEVICE;
    }
    {
        const char *genv = dev_getenv("APR_GRAPH");   // APR_GRAPH=0: small calls launch kernel by kernel too
        e->graph_on = !(genv && genv[0] == '0');
    }
    if (!rc) rc = build_frontend(e);
    if (!rc && has_model) rc = build_cnn(e, pm);
    if (!rc && has_model) {
        for (size_t l = 0; l < e->layers.size(); l++) {
            Layer &L = e->layers[l];
            char nm[48];
            if (e->fuse12 && l == 0) continue;   // no kernel of its own
            if (e->fuse12 && l == 1) snprintf(nm, sizeof nm, "conv12_fused_kernel");
            else if (L.wino == 3) snprintf(nm, sizeof nm, L.fused_waves == 4 ? "wino_fused_f2" : "wino_fused8_f2");
            else if (L.wino == 5) snprintf(nm, sizeof nm, "wino_s2_fused");
            else if (L.wino == 4) snprintf(nm, sizeof nm, "conv3x3_igemm_f32<%d, 128, 1, 3>", L.wbm_tile);
            else if (L.wino == 7) {
                L.slot_in = e->n_slots;
                add_slot(e, "wino_input_f4p", (int)l + 1);
                snprintf(nm, sizeof nm, "wino4_fused_out");
            } else if (L.wino == 6) {
                L.slot_in = e->n_slots;
                add_slot(e, "wino_input_f4", (int)l + 1);
                snprintf(nm, sizeof nm, "conv3x3_igemm_f32<%d, 128, 1, 1> x36", L.wbm_tile);
            } else if (L.wino) {
                L.slot_in = e->n_slots;
                add_slot(e, "wino_input_f2", (int)l + 1);
                snprintf(nm, sizeof nm, "conv3x3_igemm_f32<%d, 128, 1, 1>", L.wbm_tile);
            } else if (l == 0) snprintf(nm, sizeof nm, "conv1_kernel");
            else snprintf(nm, sizeof nm, "conv3x3_igemm_f32<%d, %d, %d, %d>", L.bm_tile, L.bn_tile, L.stride, L.ksplit > 1 ? 2 : 0);
            L.slot = e->n_slots;
            add_slot(e, nm, (int)l + 1);
            if (L.wino == 6) {
                L.slot_out = e->n_slots;
                add_slot(e, "wino_output_f4", (int)l + 1);
            } else if (L.wino == 2) {
                L.slot_out = e->n_slots;
                add_slot(e, "wino_output_f2", (int)l + 1);
            } else if (L.wino == 4) {
                L.slot_out = e->n_slots;
                add_slot(e, "wino_s2_output", (int)l + 1);
            } else if (L.ksplit > 1) {
                L.slot_out = e->n_slots;
                add_slot(e, "splitk_reduce_kernel", (int)l + 1);
            }
        }
        e->head_slot = e->n_slots;
        add_slot(e, "head_kernel", -1);
    }
    if (rc) {
        g_create_err = e->err;
        return rc;
    }
    guard.e = nullptr;
    *out = e;
    return APR_OK;
}

void apr_engine_destroy(apr_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto ev : e->free_events) (void)hipEventDestroy(ev);
    for (void *p : e->allocs) (void)hipFree(p);
    for (int b = 0; b < 2; b++) {
        if (e->h_pin[b]) (void)hipHostFree(e->h_pin[b]);
        if (e->h_probs[b]) (void)hipHostFree(e->h_probs[b]);
        if (e->h_label[b]) (void)hipHostFree(e->h_label[b]);
        if (e->h_match[b]) (void)hipHostFree(e->h_match[b]);
        if (e->ev_h2d[b]) (void)hipEventDestroy(e->ev_h2d[b]);
        if (e->ev_done[b]) (void)hipEventDestroy(e->ev_done[b]);
    }
    for (GraphSlot *g : {&e->g_score, &e->g_infer, &e->g_prep[0], &e->g_prep[1]})
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (e->h_ring) (void)hipHostFree(e->h_ring);
    if (e->h_g_in) (void)hipHostFree(e->h_g_in);
    if (e->h_g_out) (void)hipHostFree(e->h_g_out);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int apr_engine_num_classes(const apr_engine *e) { return e ? e->n_classes : 0; }

const char *apr_last_error(const apr_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

void *apr_engine_stream(apr_engine *e) { return e ? (void *)e->stream : nullptr; }

int apr_sync(apr_engine *e)
{
    int rc = check_engine(e);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return APR_OK;
}

int apr_prepare_batch_device(apr_engine *e, int flavour, const uint8_t *d_pcm, size_t total_bytes, size_t n_windows,
                             size_t stride_bytes, float *d_out)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if ((flavour != APR_FLAVOUR_GO && flavour != APR_FLAVOUR_LIBROSA) || (n_windows && (!d_pcm || !d_out)) || (n_windows > 1 && !stride_bytes)) {
        e->err = "apr_prepare_batch_device: bad argument";
        return APR_ERR_INVALID;
    }
    for (size_t w0 = 0; w0 < n_windows; w0 += (size_t)e->max_batch) {
        const size_t nw = std::min<size_t>(e->max_batch, n_windows - w0);
        const size_t off = w0 * stride_bytes;
        const size_t rem = off < total_bytes ? total_bytes - off : 0;
        if ((rc = run_frontend(e, flavour, d_pcm + std::min(off, total_bytes), rem, (int)nw, stride_bytes, d_out + w0 * APR_SPEC_ELEMS))) return rc;
    }
    return APR_OK;
}

int apr_prepare_batch_device_static(apr_engine *e, int flavour, const uint8_t *d_pcm, size_t total_bytes, size_t n_windows,
                                    size_t stride_bytes, float *d_out)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if ((flavour != APR_FLAVOUR_GO && flavour != APR_FLAVOUR_LIBROSA) || (n_windows && (!d_pcm || !d_out)) || (n_windows > 1 && !stride_bytes)) {
        e->err = "apr_prepare_batch_device_static: bad argument";
        return APR_ERR_INVALID;
    }
    if (n_windows > (size_t)APR_STATIC_MAX_WINDOWS) {
        e->err = "apr_prepare_batch_device_static: n_windows exceeds APR_STATIC_MAX_WINDOWS (" + std::to_string(APR_STATIC_MAX_WINDOWS) +
                 "); split the call or use apr_prepare_batch_device";
        return APR_ERR_INVALID;
    }
    if (!n_windows) return APR_OK;
    float *ptr = e->device < kMaxDevices ? g_scratch_ptr[e->device] : nullptr;   // resolved at apr_engine_create, per device
    if (!ptr) { e->err = "apr_prepare_batch_device_static: static scratch was not resolved on this device"; return APR_ERR_DEVICE; }
    // stft_mel_kernel writes every mel-power element db_norm_kernel reads (all n_frames x 128 of each of the n_windows windows),
    // so the array's previous contents never show: nothing to clear.  No event timing here (events would be allocated).
    return run_frontend(e, flavour, d_pcm, total_bytes, (int)n_windows, stride_bytes, d_out, ptr, false);
}

int apr_prepare_batch(apr_engine *e, int flavour, const uint8_t *pcm, size_t total_bytes, size_t n_windows, size_t stride_bytes,
                      float *out)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if ((flavour != APR_FLAVOUR_GO && flavour != APR_FLAVOUR_LIBROSA) || (n_windows && !out) || (total_bytes && !pcm) || (n_windows > 1 && !stride_bytes)) {
        e->err = "apr_prepare_batch: bad argument";
        return APR_ERR_INVALID;
    }
    int warn = 0;
    if (graph_ok(e, n_windows, span_bytes(total_bytes, 0, n_windows, stride_bytes))) {
        const size_t span = span_bytes(total_bytes, 0, n_windows, stride_bytes);
        rc = prepare_graph(e, flavour, pcm, span, n_windows, stride_bytes, out);
        if (rc == APR_OK) return span < (n_windows - 1) * stride_bytes + APR_SAMPLE_SIZE ? APR_WARN_UNDERFLOW : 0;
        if (rc != kGraphUnavailable) return rc;
        e->graph_on = false;
    }
    for (size_t w0 = 0; w0 < n_windows; w0 += (size_t)e->max_batch) {
        const size_t nw = std::min<size_t>(e->max_batch, n_windows - w0);
        const size_t span = span_bytes(total_bytes, w0, nw, stride_bytes);
        if (span < (nw - 1) * stride_bytes + APR_SAMPLE_SIZE) warn |= APR_WARN_UNDERFLOW;
        if ((rc = ensure_staging(e, span))) return rc;
        if (span) HIPCHK(e, hipMemcpyAsync(e->d_pcm, pcm + w0 * stride_bytes, span, hipMemcpyHostToDevice, e->stream));
        if ((rc = run_frontend(e, flavour, e->d_pcm, span, (int)nw, stride_bytes, e->d_spec))) return rc;
        HIPCHK(e, hipMemcpyAsync(out + w0 * APR_SPEC_ELEMS, e->d_spec, nw * APR_SPEC_ELEMS * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
    }
    return warn;
}

int apr_prepare(apr_engine *e, int flavour, const uint8_t *pcm, size_t nbytes, float *out)
{
    if (!e) return APR_ERR_INVALID;
    int warn = 0;
    if (nbytes > APR_SAMPLE_SIZE) { warn |= APR_WARN_OVERFLOW; nbytes = APR_SAMPLE_SIZE; }   // model.go:93-96
    int rc = apr_prepare_batch(e, flavour, pcm, nbytes, 1, APR_SAMPLE_SIZE, out);
    return rc < 0 ? rc : (rc | warn);
}

int apr_infer_device(apr_engine *e, const float *d_spec, int batch, float *d_logits)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (batch < 0 || (batch && (!d_spec || !d_logits))) { e->err = "apr_infer_device: bad argument"; return APR_ERR_INVALID; }
    for (int b0 = 0; b0 < batch; b0 += e->max_batch) {
        const int nb = std::min(e->max_batch, batch - b0);
        if ((rc = run_cnn(e, d_spec + (size_t)b0 * APR_SPEC_ELEMS, nb, 0.0, -1, d_logits + (size_t)b0 * e->n_classes, nullptr, nullptr, nullptr))) return rc;
    }
    return APR_OK;
}

int apr_infer(apr_engine *e, const float *spec, int batch, float *out_logits)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    if (batch < 0 || (batch && (!spec || !out_logits))) { e->err = "apr_infer: bad argument"; return APR_ERR_INVALID; }
    if (graph_ok(e, (size_t)batch, (size_t)batch * APR_SPEC_ELEMS * sizeof(float))) {
        rc = infer_graph(e, spec, batch, out_logits);
        if (rc != kGraphUnavailable) return rc;
        e->graph_on = false;
    }
    if ((rc = ensure_staging(e, 0))) return rc;
    for (int b0 = 0; b0 < batch; b0 += e->max_batch) {
        const int nb = std::min(e->max_batch, batch - b0);
        HIPCHK(e, hipMemcpyAsync(e->d_spec, spec + (size_t)b0 * APR_SPEC_ELEMS, (size_t)nb * APR_SPEC_ELEMS * sizeof(float), hipMemcpyHostToDevice, e->stream));
        if ((rc = run_cnn(e, e->d_spec, nb, 0.0, -1, e->d_logits, nullptr, nullptr, nullptr))) return rc;
        HIPCHK(e, hipMemcpyAsync(out_logits + (size_t)b0 * e->n_classes, e->d_logits, (size_t)nb * e->n_classes * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
    }
    return APR_OK;
}

int apr_classify_device(apr_engine *e, const float *d_spec, int batch, double trust, int empty_class, double *d_probs,
                        int32_t *d_label, uint8_t *d_match, float *d_logits)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    if (batch < 0 || (batch && !d_spec)) { e->err = "apr_classify_device: bad argument"; return APR_ERR_INVALID; }
    const int C = e->n_classes;
    for (int b0 = 0; b0 < batch; b0 += e->max_batch) {
        const int nb = std::min(e->max_batch, batch - b0);
        if ((rc = run_cnn(e, d_spec + (size_t)b0 * APR_SPEC_ELEMS, nb, trust, empty_class, d_logits ? d_logits + (size_t)b0 * C : nullptr,
                          d_probs ? d_probs + (size_t)b0 * C : nullptr, d_label ? d_label + b0 : nullptr, d_match ? d_match + b0 : nullptr)))
            return rc;
    }
    return APR_OK;
}

int apr_classify(apr_engine *e, const float *spec, int batch, double trust, int empty_class, double *probs, int32_t *label,
                 uint8_t *match, float *logits)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    if (batch < 0 || (batch && !spec)) { e->err = "apr_classify: bad argument"; return APR_ERR_INVALID; }
    if ((rc = ensure_staging(e, 0))) return rc;
    const int C = e->n_classes;
    for (int b0 = 0; b0 < batch; b0 += e->max_batch) {
        const int nb = std::min(e->max_batch, batch - b0);
        HIPCHK(e, hipMemcpyAsync(e->d_spec, spec + (size_t)b0 * APR_SPEC_ELEMS, (size_t)nb * APR_SPEC_ELEMS * sizeof(float), hipMemcpyHostToDevice, e->stream));
        if ((rc = run_cnn(e, e->d_spec, nb, trust, empty_class, e->d_logits, e->d_probs, e->d_label, e->d_match))) return rc;
        if (probs) HIPCHK(e, hipMemcpyAsync(probs + (size_t)b0 * C, e->d_probs, (size_t)nb * C * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        if (label) HIPCHK(e, hipMemcpyAsync(label + b0, e->d_label, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        if (match) HIPCHK(e, hipMemcpyAsync(match + b0, e->d_match, (size_t)nb, hipMemcpyDeviceToHost, e->stream));
        if (logits) HIPCHK(e, hipMemcpyAsync(logits + (size_t)b0 * C, e->d_logits, (size_t)nb * C * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
    }
    return APR_OK;
}

int apr_score_device(apr_engine *e, const uint8_t *d_pcm, size_t total_bytes, size_t n_windows, size_t stride_bytes, double trust,
                     int empty_class, double *d_probs, int32_t *d_label, uint8_t *d_match, float *d_logits)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    if ((n_windows && !d_pcm) || (n_windows > 1 && !stride_bytes)) { e->err = "apr_score_device: bad argument"; return APR_ERR_INVALID; }
    const int C = e->n_classes;
    for (size_t w0 = 0; w0 < n_windows; w0 += (size_t)e->max_batch) {
        const size_t nw = std::min<size_t>(e->max_batch, n_windows - w0);
        const size_t off = w0 * stride_bytes;
        const size_t rem = off < total_bytes ? total_bytes - off : 0;
        if ((rc = run_frontend(e, APR_FLAVOUR_GO, d_pcm + std::min(off, total_bytes), rem, (int)nw, stride_bytes, e->d_spec))) return rc;
        if ((rc = run_cnn(e, e->d_spec, (int)nw, trust, empty_class, d_logits ? d_logits + w0 * C : nullptr,
                          d_probs ? d_probs + w0 * C : nullptr, d_label ? d_label + w0 : nullptr, d_match ? d_match + w0 : nullptr)))
            return rc;
    }
    return APR_OK;
}

int apr_score(apr_engine *e, const uint8_t *pcm, size_t total_bytes, size_t n_windows, size_t stride_bytes, double trust,
              int empty_class, double *probs, int32_t *label, uint8_t *match)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    if ((total_bytes && !pcm) || (n_windows > 1 && !stride_bytes)) { e->err = "apr_score: bad argument"; return APR_ERR_INVALID; }
    const int C = e->n_classes;
    if (graph_ok(e, n_windows, span_bytes(total_bytes, 0, n_windows, stride_bytes))) {
        rc = score_graph(e, pcm, span_bytes(total_bytes, 0, n_windows, stride_bytes), n_windows, stride_bytes, trust, empty_class, probs, label, match);
        if (rc != kGraphUnavailable) return rc;   // capture not available: this engine launches kernel by kernel from now on
        e->graph_on = false;
    }
    for (size_t w0 = 0; w0 < n_windows; w0 += (size_t)e->max_batch) {
        const size_t nw = std::min<size_t>(e->max_batch, n_windows - w0);
        const size_t span = span_bytes(total_bytes, w0, nw, stride_bytes);
        if ((rc = ensure_staging(e, span))) return rc;
        if (span) HIPCHK(e, hipMemcpyAsync(e->d_pcm, pcm + w0 * stride_bytes, span, hipMemcpyHostToDevice, e->stream));
        if ((rc = run_frontend(e, APR_FLAVOUR_GO, e->d_pcm, span, (int)nw, stride_bytes, e->d_spec))) return rc;
        if ((rc = run_cnn(e, e->d_spec, (int)nw, trust, empty_class, nullptr, e->d_probs, e->d_label, e->d_match))) return rc;
        if (probs) HIPCHK(e, hipMemcpyAsync(probs + w0 * C, e->d_probs, nw * C * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        if (label) HIPCHK(e, hipMemcpyAsync(label + w0, e->d_label, nw * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        if (match) HIPCHK(e, hipMemcpyAsync(match + w0, e->d_match, nw, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
    }
    return APR_OK;
}


// Host copies into pinned staging, cut into pieces and spread over up to four threads (the caller's included).
struct CopyPiece { uint8_t *dst; const uint8_t *src; size_t n; };   // src == nullptr: zero fill
static void add_pieces(std::vector<CopyPiece> &pieces, uint8_t *dst, const uint8_t *src, size_t n)
{
    const size_t kPiece = (size_t)4 << 20;
    for (size_t o = 0; o < n; o += kPiece) pieces.push_back({dst + o, src ? src + o : nullptr, std::min(kPiece, n - o)});
}
static void run_pieces(const std::vector<CopyPiece> &pieces)
{
    auto one = [&pieces](size_t i) {
        if (pieces[i].src) memcpy(pieces[i].dst, pieces[i].src, pieces[i].n);
        else memset(pieces[i].dst, 0, pieces[i].n);
    };
    const size_t n_thr = std::min<size_t>(4, pieces.size());
    std::vector<std::thread> helpers;
    try {
        for (size_t t = 1; t < n_thr; t++)
            helpers.emplace_back([&one, &pieces, t, n_thr] { for (size_t i = t; i < pieces.size(); i += n_thr) one(i); });
    } catch (const std::exception &) {}   // no more threads to be had: the caller's thread copies what is left
    const size_t started = helpers.size() + 1;
    for (size_t i = 0; i < pieces.size(); i++)
        if (i % n_thr == 0 || i % n_thr >= started) one(i);
    for (std::thread &t : helpers) t.join();
}

// ---- live mode: one new second per stream and tick (src/daemon/daemon.go:104-125) ----
static int ring_reserve(apr_engine *e, int n_streams)
{
    if (n_streams <= e->ring_cap) return APR_OK;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->h_ring) { (void)hipHostFree(e->h_ring); e->h_ring = nullptr; }
    e->ring_cap = 0;
    e->ring_n = 0;
    int rc;
    if ((rc = dev_alloc(e, &e->d_ring, (size_t)n_streams * APR_SAMPLE_SIZE + 16))) return rc;   // an older ring stays in allocs until destroy
    if (hipHostMalloc((void **)&e->h_ring, (size_t)n_streams * APR_BYTES_PER_SECOND, hipHostMallocDefault) != hipSuccess) {
        e->h_ring = nullptr;
        e->err = "apr_stream_tick: could not allocate pinned staging";
        return APR_ERR_NOMEM;
    }
    e->ring_cap = n_streams;
    return APR_OK;
}

uint8_t *apr_stream_buffer(apr_engine *e, int n_streams)
{
    if (check_engine(e)) return nullptr;
    if (n_streams < 1 || n_streams > e->max_batch) { e->err = "apr_stream_buffer: n_streams must be 1 .. max_batch"; return nullptr; }
    if (ring_reserve(e, n_streams)) return nullptr;
    return e->h_ring;
}

int apr_stream_reset(apr_engine *e)
{
    int rc = check_engine(e);
    if (rc) return rc;
    e->ring_n = 0;
    return APR_OK;
}

int apr_stream_tick(apr_engine *e, const uint8_t *segments, int n_streams, double trust, int empty_class, double *probs,
                    int32_t *label, uint8_t *match)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    if (!segments || n_streams < 1 || n_streams > e->max_batch) { e->err = "apr_stream_tick: bad argument (1 <= n_streams <= max_batch)"; return APR_ERR_INVALID; }
    const bool own = e->h_ring && segments == e->h_ring && n_streams <= e->ring_cap;   // deposited in apr_stream_buffer's memory
    if (!own && (rc = ring_reserve(e, n_streams))) return rc;
    if ((rc = ensure_staging(e, 0))) return rc;
    const size_t seg = APR_BYTES_PER_SECOND, win = APR_SAMPLE_SIZE;
    const bool primed = e->ring_n == n_streams;
    if (primed)   // previous <- new of the last tick (the halves of a row do not overlap)
        HIPCHK(e, hipMemcpy2DAsync(e->d_ring, win, e->d_ring + seg, win, seg, (size_t)n_streams, hipMemcpyDeviceToDevice, e->stream));
    if (!own) {   // 24.6 MB at 256 streams: 1.3 ms of the tick on one thread
        try {
            std::vector<CopyPiece> pieces;
            add_pieces(pieces, e->h_ring, segments, (size_t)n_streams * seg);
            run_pieces(pieces);
        } catch (const std::exception &) {
            memcpy(e->h_ring, segments, (size_t)n_streams * seg);
        }
    }
    HIPCHK(e, hipMemcpy2DAsync(e->d_ring + seg, win, e->h_ring, seg, seg, (size_t)n_streams, hipMemcpyHostToDevice, e->stream));
    if (!primed) {   // the first segment of a stream has no predecessor: nothing to score yet (daemon.go:115-119)
        HIPCHK(e, hipStreamSynchronize(e->stream));
        e->ring_n = n_streams;
        return APR_STREAM_PRIMED;
    }
    const int C = e->n_classes;
    if ((rc = run_frontend(e, APR_FLAVOUR_GO, e->d_ring, (size_t)n_streams * win, n_streams, win, e->d_spec))) return rc;
    if ((rc = run_cnn(e, e->d_spec, n_streams, trust, empty_class, nullptr, e->d_probs, e->d_label, e->d_match))) return rc;
    if (probs) HIPCHK(e, hipMemcpyAsync(probs, e->d_probs, (size_t)n_streams * C * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (label) HIPCHK(e, hipMemcpyAsync(label, e->d_label, (size_t)n_streams * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (match) HIPCHK(e, hipMemcpyAsync(match, e->d_match, (size_t)n_streams, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return APR_OK;
}

// ---- integer-factor upsampler (resample.hip.h) ----
static int upsample_check(apr_engine *e, size_t n_in, int up, const double *taps, int tpp, const void *in, const void *out)
{
    if (up < 1 || tpp < 2 || tpp % 2 || up * tpp > apr::kResampleMaxTaps || !taps || (n_in && (!in || !out)) || n_in > ((size_t)1 << 40)) {
        e->err = "apr_upsample: bad argument (need up >= 1, even taps_per_phase, up * taps_per_phase <= 1024)";
        return APR_ERR_INVALID;
    }
    return APR_OK;
}

static int upsample_taps_to_device(apr_engine *e, const double *taps, int n)
{
    int rc;
    if (!e->d_rs_taps && (rc = dev_alloc(e, &e->d_rs_taps, (size_t)apr::kResampleMaxTaps))) return rc;
    HIPCHK(e, hipMemcpyAsync(e->d_rs_taps, taps, (size_t)n * sizeof(double), hipMemcpyHostToDevice, e->stream));
    return APR_OK;
}

int apr_upsample_device(apr_engine *e, const int16_t *d_in, size_t n_in, int up, const double *taps, int taps_per_phase, int16_t *d_out)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if ((rc = upsample_check(e, n_in, up, taps, taps_per_phase, d_in, d_out))) return rc;
    if (!n_in) return APR_OK;
    if ((rc = upsample_taps_to_device(e, taps, up * taps_per_phase))) return rc;
    const long long n_out = (long long)n_in * up;
    const unsigned blocks = (unsigned)std::min<long long>((n_out + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(apr::upsample_fir_kernel, dim3(blocks), dim3(256), 0, e->stream, d_in, 0LL, (long long)n_in, up, taps_per_phase,
                       e->d_rs_taps, d_out, 0LL, n_out);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));   // `taps` is the caller's host memory: the copy must be done on return
    return APR_OK;
}

int apr_upsample(apr_engine *e, const int16_t *in, size_t n_in, int up, const double *taps, int taps_per_phase, int16_t *out)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if ((rc = upsample_check(e, n_in, up, taps, taps_per_phase, in, out))) return rc;
    if (!n_in) return APR_OK;
    if ((rc = upsample_taps_to_device(e, taps, up * taps_per_phase))) return rc;
    const size_t chunk = (size_t)1 << 22, halo = (size_t)taps_per_phase;   // input samples per pass
    if (!e->d_rs_in || e->rs_cap < chunk + 2 * halo || e->rs_out_cap < chunk * (size_t)up) {
        if ((rc = dev_alloc(e, &e->d_rs_in, chunk + 2 * halo)) || (rc = dev_alloc(e, &e->d_rs_out, chunk * (size_t)up))) return rc;
        e->rs_cap = chunk + 2 * halo;
        e->rs_out_cap = chunk * (size_t)up;
    }
    for (size_t m0 = 0; m0 < n_in; m0 += chunk) {
        const size_t m1 = std::min(n_in, m0 + chunk);
        const size_t lo = m0 > halo ? m0 - halo : 0, hi = std::min(n_in, m1 + halo);
        HIPCHK(e, hipMemcpyAsync(e->d_rs_in, in + lo, (hi - lo) * sizeof(int16_t), hipMemcpyHostToDevice, e->stream));
        const long long n_count = (long long)(m1 - m0) * up;
        const unsigned blocks = (unsigned)std::min<long long>((n_count + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(apr::upsample_fir_kernel, dim3(blocks), dim3(256), 0, e->stream, e->d_rs_in, (long long)lo, (long long)n_in, up,
                           taps_per_phase, e->d_rs_taps, e->d_rs_out, (long long)m0 * up, n_count);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipMemcpyAsync(out + m0 * (size_t)up, e->d_rs_out, (size_t)n_count * sizeof(int16_t), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
    }
    return APR_OK;
}

int apr_bucket_hits_device(apr_engine *e, const int32_t *d_label, const uint8_t *d_match, size_t n_windows, int n_classes,
                           size_t windows_per_bucket, uint8_t *d_hits)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (n_classes < 1 || n_classes > APR_MAX_CLASSES || !windows_per_bucket || (n_windows && (!d_label || !d_match || !d_hits))) {
        e->err = "apr_bucket_hits_device: bad argument (1 <= n_classes <= 64, windows_per_bucket > 0)";
        return APR_ERR_INVALID;
    }
    if (!n_windows) return APR_OK;
    const unsigned long long nb = (n_windows + windows_per_bucket - 1) / windows_per_bucket;
    const unsigned blocks = (unsigned)std::min<unsigned long long>((nb + 3) / 4, 256 * 8);   // one wave per bucket, grid-stride beyond that
    hipLaunchKernelGGL(apr::bucket_hits_kernel, dim3(blocks), dim3(256), 0, e->stream, d_label, d_match, (unsigned long long)n_windows, n_classes,
                       (unsigned long long)windows_per_bucket, nb, d_hits);
    HIPCHK(e, hipGetLastError());
    return APR_OK;
}

int apr_device_alloc(apr_engine *e, size_t nbytes, void **d_out)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (!d_out) { e->err = "apr_device_alloc: null output pointer"; return APR_ERR_INVALID; }
    *d_out = nullptr;
    if (!nbytes) return APR_OK;
    if (hipMalloc(d_out, nbytes) != hipSuccess) { (void)hipGetLastError(); *d_out = nullptr; e->err = "apr_device_alloc: out of device memory"; return APR_ERR_NOMEM; }
    return APR_OK;
}

int apr_device_free(apr_engine *e, void *d_ptr)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (d_ptr) HIPCHK(e, hipFree(d_ptr));
    return APR_OK;
}

int apr_device_upload(apr_engine *e, void *d_dst, const void *h_src, size_t nbytes)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (nbytes && (!d_dst || !h_src)) { e->err = "apr_device_upload: null pointer"; return APR_ERR_INVALID; }
    if (!nbytes) return APR_OK;
    HIPCHK(e, hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return APR_OK;
}

int apr_device_download(apr_engine *e, void *h_dst, const void *d_src, size_t nbytes)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (nbytes && (!h_dst || !d_src)) { e->err = "apr_device_download: null pointer"; return APR_ERR_INVALID; }
    if (!nbytes) return APR_OK;
    HIPCHK(e, hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return APR_OK;
}

// ---- RCCL, loaded on first use: the library has no link-time dependency on it (a host that never gathers between GPUs, or a box without
//      librccl, loads libapr_amd.so all the same), and a process that already holds an RCCL (PyTorch's) gets that one by its soname ----
extern "C++" {
namespace {
struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;
bool g_rccl_tried = false;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_rccl_comms;   // device list -> its communicators; kept until the process ends

const RcclApi &rccl_api()   // g_rccl_mu held
{
    if (g_rccl_tried) return g_rccl;
    g_rccl_tried = true;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { g_rccl.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return g_rccl; }
    auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p && g_rccl.why.empty()) g_rccl.why = std::string("librccl has no symbol ") + n; return p; };
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.ok = g_rccl.why.empty();
    return g_rccl;
}
}  // namespace
}  // extern "C++"

int apr_gather_rows_device(apr_engine *const *engines, int n_engines, int root, const size_t *n_rows, int n_classes,
                           const double *const *d_probs, const int32_t *const *d_label, const uint8_t *const *d_match,
                           double *d_probs_out, int32_t *d_label_out, uint8_t *d_match_out)
{
    if (!engines || n_engines < 1 || n_engines > 64 || root < 0 || root >= n_engines || !n_rows || n_classes < 1 || n_classes > APR_MAX_CLASSES) {
        g_create_err = "apr_gather_rows_device: bad argument (1..64 engines, 0 <= root < n_engines, 1 <= n_classes <= 64)";
        return APR_ERR_INVALID;
    }
    std::vector<int> devs(n_engines);
    for (int i = 0; i < n_engines; i++) {
        if (int rc = check_engine(engines[i])) { g_create_err = "apr_gather_rows_device: null engine"; return rc; }
        devs[i] = engines[i]->device;
        for (int j = 0; j < i; j++)
            if (devs[j] == devs[i]) { engines[0]->err = g_create_err = "apr_gather_rows_device: two engines on one device (one rank per GPU)"; return APR_ERR_INVALID; }
    }
    apr_engine *er = engines[root];
    // a tensor is gathered iff its output is given; then every engine with rows must give its part
    struct Part { const void *const *src; void *dst; size_t elt; ncclDataType_t ty; };
    const Part parts[3] = {{reinterpret_cast<const void *const *>(d_probs), d_probs_out, sizeof(double) * (size_t)n_classes, ncclUint8},
                           {reinterpret_cast<const void *const *>(d_label), d_label_out, sizeof(int32_t), ncclUint8},
                           {reinterpret_cast<const void *const *>(d_match), d_match_out, sizeof(uint8_t), ncclUint8}};
    for (const Part &p : parts) {
        if (!p.dst) continue;
        for (int i = 0; i < n_engines; i++)
            if (n_rows[i] && (!p.src || !p.src[i])) { er->err = "apr_gather_rows_device: an output is given but an engine's rows are missing"; return APR_ERR_INVALID; }
    }
    std::vector<ncclComm_t> comms;
    const RcclApi *api = nullptr;
    {   // also with one engine: the communicator is what a one-GPU box can still exercise
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        api = &rccl_api();
        if (!api->ok) { er->err = "apr_gather_rows_device: " + api->why; return APR_ERR_DEVICE; }
        auto it = g_rccl_comms.find(devs);
        if (it == g_rccl_comms.end()) {
            std::vector<ncclComm_t> c(n_engines);
            const ncclResult_t r = api->CommInitAll(c.data(), n_engines, devs.data());
            if (r != ncclSuccess) { er->err = std::string("ncclCommInitAll: ") + api->GetErrorString(r); return APR_ERR_DEVICE; }
            it = g_rccl_comms.emplace(devs, std::move(c)).first;
        }
        comms = it->second;
    }
    // rows computed on each engine's stream are complete before anything is sent
    for (int i = 0; i < n_engines; i++) {
        HIPCHK(er, hipSetDevice(devs[i]));
        HIPCHK(er, hipStreamSynchronize(engines[i]->stream));
    }
    int rc = APR_OK;
    for (const Part &p : parts) {
        if (!p.dst) continue;
        ncclResult_t r = api->GroupStart();
        size_t row0 = 0;
        for (int i = 0; i < n_engines && r == ncclSuccess; i++) {
            const size_t bytes = n_rows[i] * p.elt;
            char *dst = static_cast<char *>(p.dst) + row0 * p.elt;
            row0 += n_rows[i];
            if (!bytes) continue;
            if (i == root) {
                if (hipSetDevice(devs[root]) != hipSuccess || hipMemcpyAsync(dst, p.src[i], bytes, hipMemcpyDeviceToDevice, er->stream) != hipSuccess) {
                    er->err = "apr_gather_rows_device: copy of the root's own rows failed";
                    rc = APR_ERR_DEVICE;
                }
                continue;
            }
            // rows travel as bytes: their exact bit patterns, whatever the element type
            if (hipSetDevice(devs[i]) != hipSuccess) { rc = APR_ERR_DEVICE; break; }
            r = api->Send(p.src[i], bytes, p.ty, root, comms[i], engines[i]->stream);
            if (r != ncclSuccess) break;
            if (hipSetDevice(devs[root]) != hipSuccess) { rc = APR_ERR_DEVICE; break; }
            r = api->Recv(dst, bytes, p.ty, i, comms[root], er->stream);
        }
        const ncclResult_t r2 = api->GroupEnd();     // always closed, also after an error inside the group
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { er->err = std::string("apr_gather_rows_device: RCCL: ") + api->GetErrorString(r); rc = APR_ERR_DEVICE; }
        if (rc) break;
    }
    for (int i = 0; i < n_engines; i++) {
        if (hipSetDevice(devs[i]) != hipSuccess || hipStreamSynchronize(engines[i]->stream) != hipSuccess) {
            if (!rc) { er->err = "apr_gather_rows_device: stream synchronisation failed"; rc = APR_ERR_DEVICE; }
        }
    }
    (void)hipSetDevice(devs[root]);
    return rc;
}

size_t apr_windows_in(size_t nbytes, size_t stride_bytes)
{
    // ai/inspect.py:108-109: range(0, len(raw) - bytes_per_second, bytes_per_second)
    if (!stride_bytes || nbytes <= stride_bytes) return 0;
    return (nbytes - stride_bytes + stride_bytes - 1) / stride_bytes;
}

// Staging of the file sweeps: two pinned host buffers + two device buffers + their events, sized for one stride.  Every
// step is idempotent (null checks), so a call that failed half-way is resumed by the next one instead of leaking; a
// call with another stride releases the old set first.
static void sweep_release(apr_engine *e)
{
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    for (int b = 0; b < 2; b++) {
        if (e->h_pin[b]) { (void)hipHostFree(e->h_pin[b]); e->h_pin[b] = nullptr; }
        if (e->h_probs[b]) { (void)hipHostFree(e->h_probs[b]); e->h_probs[b] = nullptr; }
        if (e->h_label[b]) { (void)hipHostFree(e->h_label[b]); e->h_label[b] = nullptr; }
        if (e->h_match[b]) { (void)hipHostFree(e->h_match[b]); e->h_match[b] = nullptr; }
        dev_free(e, &e->d_pin[b]);
        dev_free(e, &e->d_probs2[b]);
        dev_free(e, &e->d_label2[b]);
        dev_free(e, &e->d_match2[b]);
    }
    e->pin_cap = e->pin_want = 0;
}

static int sweep_reserve(apr_engine *e, size_t cap_bytes, size_t cap_w)
{
    if (e->pin_cap == cap_bytes) return APR_OK;
    if (e->pin_want && e->pin_want != cap_bytes) sweep_release(e);   // another stride: other buffer sizes
    e->pin_want = cap_bytes;
    const int C = e->n_classes;
    int rc;
    if (!e->copy_stream) HIPCHK(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {
        if (!e->h_pin[b]) HIPCHK(e, hipHostMalloc((void **)&e->h_pin[b], cap_bytes, hipHostMallocDefault));
        if (!e->h_probs[b]) HIPCHK(e, hipHostMalloc((void **)&e->h_probs[b], cap_w * C * sizeof(double), hipHostMallocDefault));
        if (!e->h_label[b]) HIPCHK(e, hipHostMalloc((void **)&e->h_label[b], cap_w * sizeof(int32_t), hipHostMallocDefault));
        if (!e->h_match[b]) HIPCHK(e, hipHostMalloc((void **)&e->h_match[b], cap_w, hipHostMallocDefault));
        if ((!e->d_pin[b] && (rc = dev_alloc(e, &e->d_pin[b], cap_bytes + 16))) || (!e->d_probs2[b] && (rc = dev_alloc(e, &e->d_probs2[b], cap_w * C))) ||
            (!e->d_label2[b] && (rc = dev_alloc(e, &e->d_label2[b], cap_w))) || (!e->d_match2[b] && (rc = dev_alloc(e, &e->d_match2[b], cap_w))))
            return rc;
        if (!e->ev_h2d[b]) HIPCHK(e, hipEventCreateWithFlags(&e->ev_h2d[b], hipEventDisableTiming));
        if (!e->ev_done[b]) HIPCHK(e, hipEventCreateWithFlags(&e->ev_done[b], hipEventDisableTiming));
    }
    e->pin_cap = cap_bytes;                     // only a complete set counts
    return APR_OK;
}

// The sweep over the files idx[0..n_idx) of the caller's list (idx == nullptr: files 0..n_files) on one engine; the rows
// of file f go to row_base[f] + k of probs / label / match.  apr_sweep_files passes every file; apr_sweep_files_multi
// gives each engine the files of its device.
static int sweep_files_on(apr_engine *e, const int *idx, int n_idx, const uint8_t *const *pcm, const size_t *nbytes, size_t stride_bytes,
                          double trust, int empty_class, double *probs, int32_t *label, uint8_t *match, const size_t *row_base)
{
    int rc;
    const int C = e->n_classes;
    const size_t span_w = (APR_SAMPLE_SIZE + stride_bytes - 1) / stride_bytes;   // strides one window spans (2 at the 1-s hop)
    const size_t cap_w = (size_t)e->max_batch;                                    // stream windows per pass
    if (span_w > cap_w) { e->err = "apr_sweep_files: stride too small for max_batch"; return APR_ERR_INVALID; }
    const size_t cap_bytes = (cap_w + span_w) * stride_bytes;
    if ((rc = sweep_reserve(e, cap_bytes, cap_w))) return rc;

    // segments: [w0, w0 + nw) of one file, laid into a pass at a stride-aligned base; the span_w - 1
    // stream windows after a segment straddle into the next one and are dropped
    // The first passes are short (512 windows, then twice as many each time up to max_batch): the staging copy and H2D of the
    // first pass have nothing to hide behind, every later one runs under the pass before it (staging 2 k windows takes about
    // as long as computing 500).  Pass sizes are bit-invisible (GPU tests).
    struct Seg { int file; size_t w0, nw, base_w, row; };
    std::vector<std::vector<Seg>> passes(1);
    auto cap_of = [&](size_t p) { return std::min(cap_w, std::max(span_w, p < 8 ? (size_t)512 << p : cap_w)); };
    size_t used = 0;
    for (int i = 0; i < n_idx; i++) {
        const int f = idx ? idx[i] : i;
        if (nbytes[f] && !pcm[f]) { e->err = "apr_sweep_files: NULL recording with a non-zero length"; return APR_ERR_INVALID; }
        size_t W = apr_windows_in(nbytes[f], stride_bytes), w0 = 0, row = row_base[f];
        while (W) {
            const size_t cap_p = cap_of(passes.size() - 1);
            const size_t room = cap_p > used ? cap_p - used : 0;
            if (room < 1) { passes.emplace_back(); used = 0; continue; }
            const size_t nw = std::min(W, room);
            passes.back().push_back({f, w0, nw, used, row});
            used += nw + span_w - 1;       // next base; may pass the cap: those windows are not computed
            if (used >= cap_p) { passes.emplace_back(); used = 0; }
            w0 += nw; W -= nw; row += nw;
        }
    }
    if (passes.back().empty()) passes.pop_back();

    auto drain = [&](int b, const std::vector<Seg> &ps) -> int {
        HIPCHK(e, hipEventSynchronize(e->ev_done[b]));
        for (const Seg &sg : ps) {
            if (probs) memcpy(probs + sg.row * C, e->h_probs[b] + sg.base_w * C, sg.nw * C * sizeof(double));
            if (label) memcpy(label + sg.row, e->h_label[b] + sg.base_w, sg.nw * sizeof(int32_t));
            if (match) memcpy(match + sg.row, e->h_match[b] + sg.base_w, sg.nw);
        }
        return APR_OK;
    };
    for (size_t p = 0; p < passes.size(); p++) {
        const int b = (int)(p & 1);
        if (p >= 2 && (rc = drain(b, passes[p - 2]))) return rc;       // buffer b is free again
        const std::vector<Seg> &ps = passes[p];
        const Seg &last = ps.back();
        const size_t n_stream = last.base_w + last.nw;                  // stream windows computed in this pass
        const size_t bytes = (n_stream - 1) * stride_bytes + APR_SAMPLE_SIZE;
        // pageable -> pinned staging on a few host threads (one thread copies 8-10 GB/s: 20-25 ms for a pass of 2048 windows,
        // which the first pass of a sweep cannot hide behind anything)
        std::vector<CopyPiece> pieces;
        size_t cur = 0;
        for (const Seg &sg : ps) {
            const size_t base = sg.base_w * stride_bytes;
            if (base > cur) add_pieces(pieces, e->h_pin[b] + cur, nullptr, base - cur);
            const size_t src0 = sg.w0 * stride_bytes;
            const size_t want = (sg.nw - 1) * stride_bytes + APR_SAMPLE_SIZE;
            const size_t have = std::min(want, nbytes[sg.file] - src0);
            add_pieces(pieces, e->h_pin[b] + base, pcm[sg.file] + src0, have);
            cur = base + have;
        }
        if (bytes > cur) add_pieces(pieces, e->h_pin[b] + cur, nullptr, bytes - cur);
        run_pieces(pieces);
        HIPCHK(e, hipMemcpyAsync(e->d_pin[b], e->h_pin[b], bytes, hipMemcpyHostToDevice, e->copy_stream));
        HIPCHK(e, hipEventRecord(e->ev_h2d[b], e->copy_stream));
        HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_h2d[b], 0));
        if ((rc = run_frontend(e, APR_FLAVOUR_GO, e->d_pin[b], bytes, (int)n_stream, stride_bytes, e->d_spec))) return rc;
        if ((rc = run_cnn(e, e->d_spec, (int)n_stream, trust, empty_class, nullptr, e->d_probs2[b], e->d_label2[b], e->d_match2[b]))) return rc;
        HIPCHK(e, hipMemcpyAsync(e->h_probs[b], e->d_probs2[b], n_stream * C * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(e->h_label[b], e->d_label2[b], n_stream * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(e->h_match[b], e->d_match2[b], n_stream, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipEventRecord(e->ev_done[b], e->stream));
    }
    for (size_t p = passes.size() >= 2 ? passes.size() - 2 : 0; p < passes.size(); p++)
        if ((rc = drain((int)(p & 1), passes[p]))) return rc;
    return APR_OK;
}

int apr_sweep_plan(int n_files, const size_t *nbytes, size_t stride_bytes, int n_devices, int32_t *file_device, size_t *first_window,
                   size_t *device_windows)
{
    if (n_files < 0 || (n_files && !nbytes) || !stride_bytes || n_devices < 1) return APR_ERR_INVALID;
    if (device_windows) for (int d = 0; d < n_devices; d++) device_windows[d] = 0;
    size_t row = 0;
    for (int f = 0; f < n_files; f++) {
        const int d = f % n_devices;               // by file, round-robin by index: a window never crosses a shard
        const size_t w = apr_windows_in(nbytes[f], stride_bytes);
        if (file_device) file_device[f] = d;
        if (first_window) first_window[f] = row;
        if (device_windows) device_windows[d] += w;
        row += w;
    }
    if (first_window) first_window[n_files] = row;
    return APR_OK;
}

int apr_sweep_files(apr_engine *e, int n_files, const uint8_t *const *pcm, const size_t *nbytes, size_t stride_bytes, double trust,
                    int empty_class, double *probs, int32_t *label, uint8_t *match, size_t *first_window)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (e->layers.empty()) { e->err = "engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
    if (n_files < 0 || (n_files && (!pcm || !nbytes)) || !stride_bytes || stride_bytes % 2) { e->err = "apr_sweep_files: bad argument"; return APR_ERR_INVALID; }
    try {
        std::vector<size_t> first((size_t)n_files + 1);
        if ((rc = apr_sweep_plan(n_files, nbytes, stride_bytes, 1, nullptr, first.data(), nullptr))) return rc;
        if (first_window) memcpy(first_window, first.data(), first.size() * sizeof(size_t));
        return sweep_files_on(e, nullptr, n_files, pcm, nbytes, stride_bytes, trust, empty_class, probs, label, match, first.data());
    } catch (const std::exception &ex) {
        e->err = std::string("apr_sweep_files: out of host memory: ") + ex.what();
        return APR_ERR_NOMEM;
    }
}

int apr_sweep_files_multi(apr_engine *const *engines, int n_engines, int n_files, const uint8_t *const *pcm, const size_t *nbytes,
                          size_t stride_bytes, double trust, int empty_class, double *probs, int32_t *label, uint8_t *match,
                          size_t *first_window)
{
    if (!engines || n_engines < 1 || !engines[0]) return APR_ERR_INVALID;
    apr_engine *e0 = engines[0];
    for (int i = 0; i < n_engines; i++) {
        if (!engines[i]) { e0->err = "apr_sweep_files_multi: NULL engine in the list"; return APR_ERR_INVALID; }
        if (engines[i]->layers.empty()) { e0->err = "apr_sweep_files_multi: engine has no model (front-end only)"; return APR_ERR_NOMODEL; }
        if (engines[i]->n_classes != e0->n_classes) { e0->err = "apr_sweep_files_multi: the engines hold models with different class counts"; return APR_ERR_INVALID; }
        for (int j = 0; j < i; j++)
            if (engines[j] == engines[i]) { e0->err = "apr_sweep_files_multi: the same engine is listed twice (a handle serves one caller at a time)"; return APR_ERR_INVALID; }
    }
    if (n_files < 0 || (n_files && (!pcm || !nbytes)) || !stride_bytes || stride_bytes % 2) { e0->err = "apr_sweep_files_multi: bad argument"; return APR_ERR_INVALID; }
    try {
        std::vector<size_t> first((size_t)n_files + 1);
        std::vector<int32_t> dev((size_t)std::max(n_files, 1));
        int rc = apr_sweep_plan(n_files, nbytes, stride_bytes, n_engines, dev.data(), first.data(), nullptr);
        if (rc) return rc;
        if (first_window) memcpy(first_window, first.data(), first.size() * sizeof(size_t));
        std::vector<std::vector<int>> mine((size_t)n_engines);
        for (int f = 0; f < n_files; f++) mine[(size_t)dev[f]].push_back(f);
        // one host thread per engine: each binds its engine's GPU (check_engine), stages its own files through its own
        // pinned buffers and writes its rows straight into the caller's arrays (disjoint rows: no gather step on the host)
        std::vector<int> rcs((size_t)n_engines, APR_OK);
        auto work = [&](int i) {
            apr_engine *e = engines[i];
            try {
                int r = check_engine(e);
                if (!r) r = sweep_files_on(e, mine[i].data(), (int)mine[i].size(), pcm, nbytes, stride_bytes, trust, empty_class, probs, label, match, first.data());
                rcs[i] = r;
            } catch (const std::exception &ex) {
                e->err = std::string("apr_sweep_files_multi: out of host memory: ") + ex.what();
                rcs[i] = APR_ERR_NOMEM;
            }
        };
        std::vector<std::thread> threads;
        threads.reserve((size_t)n_engines);
        for (int i = 1; i < n_engines; i++) {
            try {
                threads.emplace_back(work, i);
            } catch (const std::exception &) {
                work(i);                           // no thread to be had: this engine's share runs here, one after the other
            }
        }
        work(0);
        for (auto &t : threads) t.join();
        for (int i = 0; i < n_engines; i++)
            if (rcs[i]) {
                if (i) e0->err = "engine " + std::to_string(i) + " (device " + std::to_string(engines[i]->device) + "): " + engines[i]->err;
                return rcs[i];
            }
        return APR_OK;
    } catch (const std::exception &ex) {
        e0->err = std::string("apr_sweep_files_multi: ") + ex.what();
        return APR_ERR_NOMEM;
    }
}

int apr_profile_enable(apr_engine *e, int on)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (!on) rc = prof_flush(e);
    e->prof_on = on != 0;
    return rc;
}

int apr_mfma_probe(apr_engine *e, double *tflops)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (!tflops) { e->err = "apr_mfma_probe: bad argument"; return APR_ERR_INVALID; }
    if (!e->d_probe && (rc = dev_alloc(e, &e->d_probe, (size_t)64))) return rc;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        if (a) (void)hipEventDestroy(a);
        e->err = "apr_mfma_probe: no events";
        return APR_ERR_DEVICE;
    }
    const int iters = 4096, blocks = e->n_cus * 2;            // 8 waves per CU: two per SIMD
    double best = 0.0;
    for (int r = 0; r < 3; r++) {                              // the first round also warms the clock up
        (void)hipEventRecord(a, e->stream);
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, e->stream, e->d_probe, iters);
        (void)hipEventRecord(b, e->stream);
        float ms = 0.0f;
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess || ms <= 0.0f) {
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
            e->err = "apr_mfma_probe: launch failed";
            return APR_ERR_DEVICE;
        }
        const double flop = (double)blocks * 4.0 * iters * 8.0 * 4096.0;   // waves x iterations x 8 MFMAs x 2*32*32*2
        best = std::max(best, flop / (ms * 1e-3) / 1e12);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *tflops = best;
    return APR_OK;
}

int apr_profile_reset(apr_engine *e)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if ((rc = prof_flush(e))) return rc;
    for (int i = 0; i < e->n_slots; i++) { e->slots[i].ms = 0; e->slots[i].launches = 0; e->slots[i].flops = 0; e->slots[i].bytes = 0; }
    return APR_OK;
}

int apr_profile_read(apr_engine *e, apr_profile_slot *slots, int *n_slots)
{
    int rc = check_engine(e);
    if (rc) return rc;
    if (!slots || !n_slots) return APR_ERR_INVALID;
    if ((rc = prof_flush(e))) return rc;
    memcpy(slots, e->slots, sizeof(apr_profile_slot) * e->n_slots);
    *n_slots = e->n_slots;
    return APR_OK;
}

}  // extern "C"

#ifdef W4_STAMPS   // developer build (tools/w4_stamps.py)
extern "C" int apr_debug_w4_stamps(unsigned long long *dst, int reset)
{
    static const unsigned long long zero[64] = {};
    if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(apr::w4_stamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(apr::w4_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

#if defined(WS_STAMPS) || defined(W8_STAMPS)   // developer builds (tools/ws_stamps.py): 64 counters of workgroup 0 of the stamped kernel
extern "C" int apr_debug_ws_stamps(unsigned long long *dst, int reset)
{
    static const unsigned long long zero[64] = {};
#ifdef WS_STAMPS
    if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(apr::ws_stamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(apr::ws_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
#else
    if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(apr::w8_stamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(apr::w8_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
#endif
}
#endif
