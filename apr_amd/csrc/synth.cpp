// synth.cpp -- deterministic synthetic recordings (host side; SURVEY 8(d) recipe).
//
// The reference decodes every source with `ffmpeg -ar 48000 -ac 1 -f s16le`
// (src/ffmpeg/ffmpeg.go:71-81, ai/inspect.py:96-104), so the hot path only ever sees
// 48 kHz mono s16le.  There is no ffmpeg here; a "16 kHz mono" source is modelled as a
// 16 kHz signal (content < 8 kHz) linearly interpolated x3 to 48 kHz:
//   * noise at -30 dBFS (sum of four uniforms, counter-based splitmix64 per sample),
//   * per second 0..3 Hann-shaped tone/chirp bursts, 200 Hz..7.5 kHz, -20..-3 dBFS,
//     50..800 ms (a burst may run on into the next second),
//   * with probability 0.1 per second an exact-zero stretch (exercises the 1e-10 floor).
// Every second of every file can be generated independently of the others.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../include/apr.h"

namespace {

inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t rnd(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return mix64(seed + 0x9E3779B97F4A7C15ull * (idx + 1) + 0xD1B54A32D192ED03ull * (stream + 1));
}
inline double u01(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

struct Burst {
    double t0, dur, f0, f1, amp;   // seconds (absolute), Hz, linear amplitude
};

void bursts_of_second(uint64_t fseed, uint64_t sec, std::vector<Burst> &out)
{
    const uint64_t r0 = rnd(fseed, 1, sec);
    const int nb = (int)(r0 & 3);   // 0..3
    for (int b = 0; b < nb; b++) {
        Burst q;
        q.t0 = (double)sec + u01(rnd(fseed, 2, sec * 4 + b));
        q.dur = 0.05 + 0.75 * u01(rnd(fseed, 3, sec * 4 + b));
        q.f0 = 200.0 + 7300.0 * u01(rnd(fseed, 4, sec * 4 + b));
        const bool chirp = (rnd(fseed, 5, sec * 4 + b) & 1) != 0;
        q.f1 = chirp ? 200.0 + 7300.0 * u01(rnd(fseed, 6, sec * 4 + b)) : q.f0;
        const double db = -20.0 + 17.0 * u01(rnd(fseed, 7, sec * 4 + b));
        q.amp = std::pow(10.0, db / 20.0);
        out.push_back(q);
    }
}

// value of the 16 kHz signal at absolute 16 kHz sample index i (normalised to [-1, 1])
inline double sig16(uint64_t fseed, uint64_t i, const std::vector<Burst> &bursts)
{
    const uint64_t r = rnd(fseed, 0, i);
    const double s4 = (double)(r & 0xFFFF) + (double)((r >> 16) & 0xFFFF) + (double)((r >> 32) & 0xFFFF) + (double)(r >> 48);
    const double g = (s4 / 65536.0 - 2.0) * 1.7320508075688772;   // ~N(0,1)
    double v = 0.031622776601683794 * g;                           // -30 dBFS rms
    const double t = (double)i / 16000.0;
    for (const Burst &b : bursts) {
        const double x = t - b.t0;
        if (x < 0.0 || x >= b.dur) continue;
        const double env = 0.5 * (1.0 - std::cos(2.0 * M_PI * x / b.dur));
        const double ph = 2.0 * M_PI * (b.f0 * x + 0.5 * (b.f1 - b.f0) / b.dur * x * x);
        v += b.amp * env * std::sin(ph);
    }
    return v;
}

void gen_second(uint64_t fseed, uint64_t sec, int16_t *out)
{
    std::vector<Burst> bursts;
    if (sec > 0) bursts_of_second(fseed, sec - 1, bursts);
    bursts_of_second(fseed, sec, bursts);
    std::vector<double> x(16001);
    for (int i = 0; i <= 16000; i++) x[i] = sig16(fseed, sec * 16000ull + (uint64_t)i, bursts);
    // exact-zero stretch
    int z0 = -1, z1 = -1;
    const uint64_t rz = rnd(fseed, 8, sec);
    if ((rz % 10) == 0) {
        z0 = (int)(48000.0 * 0.8 * u01(rnd(fseed, 9, sec)));
        z1 = z0 + 2400 + (int)(7200.0 * u01(rnd(fseed, 10, sec)));
        if (z1 > 48000) z1 = 48000;
    }
    for (int n = 0; n < 48000; n++) {
        const int i = n / 3, ph = n - 3 * i;
        double v = x[i] + (x[i + 1] - x[i]) * ((double)ph / 3.0);
        if (n >= z0 && n < z1) v = 0.0;
        double s = std::nearbyint(v * 32768.0);
        s = std::min(32767.0, std::max(-32768.0, s));
        out[n] = (int16_t)s;
    }
}

}  // namespace

extern "C" int apr_synth_pcm(uint64_t base_seed, uint64_t file_index, uint64_t second_offset, size_t n_seconds,
                             int16_t *out_48k)
{
    if (!out_48k && n_seconds) return APR_ERR_INVALID;
    const uint64_t fseed = mix64(base_seed ^ mix64(file_index + 0x632BE59BD9B4E019ull));
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    nt = std::min<unsigned>(nt, 16u);
    nt = (unsigned)std::min<size_t>(nt, n_seconds ? n_seconds : 1);
    auto work = [&](unsigned t) {
        for (size_t s = t; s < n_seconds; s += nt) gen_second(fseed, second_offset + s, out_48k + s * 48000);
    };
    if (nt <= 1) { work(0); return APR_OK; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &t : th) t.join();
    return APR_OK;
}
