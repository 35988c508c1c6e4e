// ingest.cpp -- host-side helpers either side of the scoring path (no GPU, plain C ABI):
//   apr_wav_data_offset   where the PCM payload of a RIFF/WAVE stream starts.  The reference pipes
//                         `ffmpeg ... -f wav -` into the segment reader without stripping the header
//                         (src/ffmpeg/ffmpeg.go:111-114, src/daemon/daemon.go:151-154), so the first
//                         window of a live stream begins with header bytes read as samples; a caller that
//                         wants clean audio skips this many bytes first.
//   apr_resample_taps     the polyphase low-pass used by apr_upsample (Kaiser-windowed sinc).
//   apr_bucket_hits       the report rule "1 clap or 999 claps within 1 minute is logged as one hit"
//                         (docs/usage/review.md:48-53; not implemented in the reference's code).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/apr.h"

namespace {
uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// modified Bessel function I0 by its power series: sum_k ((x/2)^k / k!)^2
double bessel_i0(double x)
{
    const double h = 0.5 * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 500; k++) {
        const double f = h / (double)k;
        term = term * f * f;
        sum = sum + term;
        if (term < 1e-21 * sum) break;
    }
    return sum;
}
}  // namespace

extern "C" {

size_t apr_wav_data_offset(const uint8_t *buf, size_t n, uint32_t *sample_rate, uint16_t *channels, uint16_t *bits_per_sample)
{
    if (sample_rate) *sample_rate = 0;
    if (channels) *channels = 0;
    if (bits_per_sample) *bits_per_sample = 0;
    if (!buf || n < 12 || memcmp(buf, "RIFF", 4) != 0 || memcmp(buf + 8, "WAVE", 4) != 0) return 0;   // raw PCM: nothing to skip
    size_t pos = 12;
    while (pos + 8 <= n) {
        const uint32_t sz = rd32(buf + pos + 4);
        if (memcmp(buf + pos, "fmt ", 4) == 0 && pos + 8 + 16 <= n) {
            if (channels) *channels = rd16(buf + pos + 10);
            if (sample_rate) *sample_rate = rd32(buf + pos + 12);
            if (bits_per_sample) *bits_per_sample = rd16(buf + pos + 22);
        } else if (memcmp(buf + pos, "data", 4) == 0) {
            return pos + 8;   // ffmpeg's streamed WAV carries size 0xFFFFFFFF here: the payload runs to the end
        }
        const size_t adv = 8 + (size_t)sz + (sz & 1u);   // chunks are word-aligned
        if (adv > n - pos) break;
        pos += adv;
    }
    return 0;   // no data chunk in what was given
}

int apr_resample_taps(int up, int taps_per_phase, double cutoff, double beta, double *taps /*[up][taps_per_phase]*/)
{
    if (up < 1 || taps_per_phase < 2 || taps_per_phase % 2 || up * taps_per_phase > 1024 || !(cutoff > 0.0 && cutoff <= 1.0) || !(beta >= 0.0) || !taps)
        return APR_ERR_INVALID;
    // prototype h[j] = g(j - c), c = up * tpp / 2, g(t) = cutoff * sinc(cutoff * t / up) * kaiser_beta(t / c);
    // phase p takes h[up k + p]; each phase is scaled to sum 1 so that DC passes at gain 1 in every phase
    const int L = up * taps_per_phase, c = L / 2;
    const double pi = 3.14159265358979323846, i0b = bessel_i0(beta);
    for (int p = 0; p < up; p++) {
        double sum = 0.0;
        for (int k = 0; k < taps_per_phase; k++) {
            const double t = (double)(up * k + p - c);
            const double a = cutoff * t / (double)up;
            const double sinc = t == 0.0 ? 1.0 : sin(pi * a) / (pi * a);
            const double u = t / (double)c;
            const double w = bessel_i0(beta * sqrt(1.0 - u * u)) / i0b;
            const double v = cutoff * sinc * w;
            taps[p * taps_per_phase + k] = v;
            sum = sum + v;
        }
        for (int k = 0; k < taps_per_phase; k++) taps[p * taps_per_phase + k] = taps[p * taps_per_phase + k] / sum;
    }
    return APR_OK;
}

int apr_bucket_hits(const int32_t *label, const uint8_t *match, size_t n_windows, int n_classes, size_t windows_per_bucket,
                    uint8_t *hits /*[ceil(n/wpb)][n_classes]*/)
{
    if (n_classes < 1 || !windows_per_bucket || (n_windows && (!label || !match || !hits))) return APR_ERR_INVALID;
    const size_t nb = (n_windows + windows_per_bucket - 1) / windows_per_bucket;
    memset(hits, 0, nb * (size_t)n_classes);
    for (size_t k = 0; k < n_windows; k++)
        if (match[k] && label[k] >= 0 && label[k] < n_classes) hits[(k / windows_per_bucket) * (size_t)n_classes + (size_t)label[k]] = 1;
    return APR_OK;
}

}  // extern "C"
