/*
 * apr_oracle.c -- CPU restatement of the DTrack detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (apr_amd/, the C-ABI
 * library) may import, link or call this file.  It is used by tests/ as the
 * checker, by __graft_entry__.smoke() as the checker, and by bench.py's
 * cpu_baseline leg as the timed CPU stand-in for the reference.
 *
 * Parity status:
 *   - CNN (orc_cnn_forward): PINNED against the reference's own
 *     ai.model.NoiseDetector (imported in the build container, see
 *     tests/golden/make_golden.py) through tests/golden/cnn_logits_*.npy.
 *   - go-flavour DSP (orc_prepare_go): PARITY UNPINNED.  The reference Go code
 *     cannot be built here (no Go toolchain; go-dsp is not vendored) and the
 *     reference's tests pin shapes only (src/model/model_test.go:26-51).  It is
 *     a line-by-line restatement of src/model/model.go:86-184,257-265 and
 *     src/model/dsp.go:31-165 checked against hand-derived known answers.
 *     Third-party arithmetic restated from its published definition:
 *     github.com/mjibson/go-dsp @ 11479a337f12 (src/go.mod:7):
 *       window.Hann(L)[n] = 0.5*(1-cos(2*pi*n/(L-1)))      (symmetric)
 *       fft.FFT(x)[k]     = sum_n x[n]*exp(-2*pi*i*k*n/N)  (unnormalised)
 *   - librosa-flavour DSP (orc_prepare_librosa): PARITY UNPINNED.  librosa is
 *     absent and unpinned in ai/requirements.txt; restated from
 *     ai/model.py:28-34,184-204 plus librosa>=0.10 documented defaults
 *     (center=True, pad_mode="constant", periodic hann, Slaney-normalised
 *     HTK mel filters, power_to_db(ref=max, amin=1e-10, top_db=80)).
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: src/ffmpeg/ffmpeg.go:20-21, src/model/model.go:23-40 ---- */
#define SAMPLE_RATE 48000
#define BYTES_PER_SECOND 96000
#define SEGMENT_SIZE 2
#define SAMPLE_SIZE (BYTES_PER_SECOND * SEGMENT_SIZE) /* 192000 bytes */
#define NMELS 128
#define NFFT 2048
#define HOP 512
#define SPEC_FRAMES 188
#define NBINS (NFFT / 2 + 1) /* 1025 */
#define INT16_MAX_F 32768.0f
#define MAX_FREQ ((double)SAMPLE_RATE / 2.0)
#define MIN_FREQ 0.0
#define MEL_SCALAR 2595.0 /* dsp.go:13 */
#define MEL_BREAK 700.0   /* dsp.go:14 */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define ORC_WARN_UNDERFLOW 1
#define ORC_WARN_OVERFLOW 2
#define ORC_WARN_FRAMECLAMP 4

/* ------------------------------------------------------------------------ */
/* A1  normalizeAudio  (src/model/model.go:257-265)                          */
/* ------------------------------------------------------------------------ */
void orc_normalize_audio(const uint8_t *pcm, size_t nbytes, float *out)
{
    size_t n = nbytes / 2;
    for (size_t i = 0; i < n; i++) {
        int16_t v = (int16_t)((uint16_t)pcm[2 * i] | ((uint16_t)pcm[2 * i + 1] << 8));
        out[i] = (float)v / INT16_MAX_F;
    }
}

/* ------------------------------------------------------------------------ */
/* go-dsp fft.FFT restated: unnormalised forward DFT, radix-2, complex128.   */
/* call site src/model/model.go:142                                          */
/* ------------------------------------------------------------------------ */
static void fft_c128(double *re, double *im, int n, const double *twr, const double *twi)
{
    /* bit reversal */
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < half; k++) {
                double wr = twr[k * step], wi = twi[k * step];
                double ur = re[i + k], ui = im[i + k];
                double xr = re[i + k + half], xi = im[i + k + half];
                double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
                re[i + k] = ur + vr; im[i + k] = ui + vi;
                re[i + k + half] = ur - vr; im[i + k + half] = ui - vi;
            }
        }
    }
}

static double g_twr[NFFT / 2], g_twi[NFFT / 2];
static double g_hann_sym[NFFT];  /* go-dsp window.Hann(2048), model.go:123 */
static double g_hann_per[NFFT];  /* scipy/librosa get_window('hann', fftbins=True) */
static double g_mel_w[NMELS][NBINS]; /* dsp.go cachedMelWeights */
static int g_mel_start[NMELS], g_mel_end[NMELS], g_bin_points[NMELS + 2];
static float g_lib_mel[NMELS][NBINS]; /* ai/model.py MEL_BASIS */
static int g_init_done = 0;

/* dsp.go:158-165 */
static double hz_to_mel(double hz) { return MEL_SCALAR * log10(1.0 + hz / MEL_BREAK); }
static double mel_to_hz(double mel) { return MEL_BREAK * (pow(10.0, mel / MEL_SCALAR) - 1.0); }

/* ------------------------------------------------------------------------ */
/* A3  init() mel "lens"  (src/model/dsp.go:31-80)                            */
/* ------------------------------------------------------------------------ */
static void mel_lens_go(void)
{
    double min_mel = hz_to_mel(MIN_FREQ), max_mel = hz_to_mel(MAX_FREQ);
    double mel_points[NMELS + 2];
    double step = (max_mel - min_mel) / (double)(NMELS + 1);       /* dsp.go:39 */
    for (int i = 0; i < NMELS + 2; i++) mel_points[i] = mel_to_hz(min_mel + (double)i * step);
    for (int i = 0; i < NMELS + 2; i++)                            /* dsp.go:48 */
        g_bin_points[i] = (int)floor((NFFT + 1) * mel_points[i] / (double)SAMPLE_RATE);
    memset(g_mel_w, 0, sizeof g_mel_w);
    for (int i = 0; i < NMELS; i++) {
        int start = g_bin_points[i], center = g_bin_points[i + 1], end = g_bin_points[i + 2];
        g_mel_start[i] = start; g_mel_end[i] = end;                /* dsp.go:64-67 */
        for (int f = start; f < center; f++) {                     /* dsp.go:70-73 */
            if (f >= NBINS) break;
            g_mel_w[i][f] = (double)(f - start) / (double)(center - start);
        }
        for (int f = center; f < end; f++) {                       /* dsp.go:75-78 */
            if (f >= NBINS) break;
            g_mel_w[i][f] = (double)(end - f) / (double)(end - center);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* A11 MEL_BASIS = librosa.filters.mel(htk=True, sr, n_fft, n_mels, fmin=0,   */
/*     fmax=sr/2)  (ai/model.py:28-34); librosa defaults: norm='slaney',      */
/*     dtype=float32.                                                         */
/* ------------------------------------------------------------------------ */
static void mel_basis_librosa(void)
{
    double fftfreqs[NBINS], mel_f[NMELS + 2];
    for (int k = 0; k < NBINS; k++) fftfreqs[k] = (double)k * ((double)SAMPLE_RATE / 2.0) / (double)(NBINS - 1);
    double min_mel = hz_to_mel(0.0), max_mel = hz_to_mel((double)SAMPLE_RATE / 2.0);
    for (int i = 0; i < NMELS + 2; i++) {
        /* np.linspace(min_mel, max_mel, n_mels + 2) */
        double m = min_mel + (max_mel - min_mel) * (double)i / (double)(NMELS + 1);
        mel_f[i] = mel_to_hz(m);
    }
    for (int i = 0; i < NMELS; i++) {
        double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < NBINS; k++) {
            double lower = -(mel_f[i] - fftfreqs[k]) / fd0;
            double upper = (mel_f[i + 2] - fftfreqs[k]) / fd1;
            double w = lower < upper ? lower : upper;
            if (w < 0.0) w = 0.0;
            float wf = (float)w;                 /* stored into a float32 array */
            g_lib_mel[i][k] = (float)((double)wf * enorm); /* weights *= enorm (in place, f32) */
        }
    }
}

static void orc_init(void)
{
    if (g_init_done) return;
#pragma omp critical(orc_init_lock)
    {
        if (!g_init_done) {
            for (int k = 0; k < NFFT / 2; k++) {
                g_twr[k] = cos(-2.0 * M_PI * (double)k / (double)NFFT);
                g_twi[k] = sin(-2.0 * M_PI * (double)k / (double)NFFT);
            }
            for (int n = 0; n < NFFT; n++) {
                g_hann_sym[n] = 0.5 * (1.0 - cos(2.0 * M_PI * (double)n / (double)(NFFT - 1)));
                g_hann_per[n] = 0.5 * (1.0 - cos(2.0 * M_PI * (double)n / (double)NFFT));
            }
            mel_lens_go();
            mel_basis_librosa();
            g_init_done = 1;
        }
    }
}

/* Export the go-flavour lens: weights [128*1025] f64, bounds [128*2], bin_points [130]. */
void orc_mel_lens(double *weights, int *bounds, int *bin_points)
{
    orc_init();
    if (weights) memcpy(weights, g_mel_w, sizeof g_mel_w);
    if (bounds) for (int i = 0; i < NMELS; i++) { bounds[2 * i] = g_mel_start[i]; bounds[2 * i + 1] = g_mel_end[i]; }
    if (bin_points) memcpy(bin_points, g_bin_points, sizeof g_bin_points);
}

void orc_librosa_mel_basis(float *weights)
{
    orc_init();
    memcpy(weights, g_lib_mel, sizeof g_lib_mel);
}

void orc_hann(double *sym, double *per)
{
    orc_init();
    if (sym) memcpy(sym, g_hann_sym, sizeof g_hann_sym);
    if (per) memcpy(per, g_hann_per, sizeof g_hann_per);
}

/* ------------------------------------------------------------------------ */
/* A1', A2, A4, A5  Prepare  (src/model/model.go:86-184)                      */
/*   power_out: optional [1025][184] f64, mel_out: optional [128][184] f64.   */
/*   out: [128][188] f32.  Returns warn flags.                                */
/* ------------------------------------------------------------------------ */
int orc_prepare_go_stages(const uint8_t *pcm_in, size_t nbytes, double *power_out, double *mel_out, float *out)
{
    orc_init();
    int warn = 0;
    uint8_t *pcm = (uint8_t *)calloc(SAMPLE_SIZE, 1);
    if (nbytes < SAMPLE_SIZE) { warn |= ORC_WARN_UNDERFLOW; memcpy(pcm, pcm_in, nbytes); }   /* model.go:88-92 */
    else { if (nbytes > SAMPLE_SIZE) warn |= ORC_WARN_OVERFLOW; memcpy(pcm, pcm_in, SAMPLE_SIZE); } /* model.go:93-96 */

    const int nsamp = SAMPLE_SIZE / 2;
    float *af = (float *)malloc(sizeof(float) * nsamp);
    double *ad = (double *)malloc(sizeof(double) * nsamp);
    orc_normalize_audio(pcm, SAMPLE_SIZE, af);                     /* model.go:99 */
    for (int i = 0; i < nsamp; i++) ad[i] = (double)af[i];         /* model.go:100-103 */

    int nframes = (nsamp - NFFT) / HOP + 1;                        /* model.go:106 -> 184 */
    if (nframes > SPEC_FRAMES) { warn |= ORC_WARN_FRAMECLAMP; nframes = SPEC_FRAMES; }

    double *power = (double *)malloc(sizeof(double) * NBINS * nframes); /* [bin][frame] model.go:118-121 */
    double re[NFFT], im[NFFT];
    for (int i = 0; i < nframes; i++) {                            /* model.go:125-150 */
        int start = i * HOP, end = start + NFFT;
        if (end > nsamp) end = nsamp;
        int seg = end - start;
        for (int j = 0; j < NFFT; j++) { re[j] = (j < seg) ? ad[start + j] * g_hann_sym[j] : 0.0; im[j] = 0.0; }
        fft_c128(re, im, NFFT, g_twr, g_twi);
        for (int j = 0; j < NBINS; j++) power[(size_t)j * nframes + i] = re[j] * re[j] + im[j] * im[j];
    }

    /* applyMelFilterbank  dsp.go:83-111 */
    double *mel = (double *)malloc(sizeof(double) * NMELS * nframes);
    for (int m = 0; m < NMELS; m++) {
        int start = g_mel_start[m], end = g_mel_end[m];
        if (end > NBINS) end = NBINS;
        for (int t = 0; t < nframes; t++) {
            double sum = 0.0;
            for (int k = start; k < end; k++) sum += g_mel_w[m][k] * power[(size_t)k * nframes + t];
            mel[(size_t)m * nframes + t] = sum;
        }
    }
    /* powerToDb  dsp.go:132-155 */
    double max_val = 1e-10;
    for (int i = 0; i < NMELS * nframes; i++) if (mel[i] > max_val) max_val = mel[i];
    /* fixedNormalize dsp.go:114-129 + pack model.go:157-171 */
    for (int r = 0; r < NMELS; r++) {
        for (int c = 0; c < SPEC_FRAMES; c++) {
            if (c < nframes) {
                double v = mel[(size_t)r * nframes + c];
                if (v < 1e-10) v = 1e-10;
                double db = 10.0 * log10(v / max_val);
                if (db < -80.0) db = -80.0;
                if (db > 0.0) db = 0.0;
                out[r * SPEC_FRAMES + c] = (float)((db + 80.0) / 80.0);
            } else {
                out[r * SPEC_FRAMES + c] = 0.0f;
            }
        }
    }
    if (power_out) memcpy(power_out, power, sizeof(double) * NBINS * nframes);
    if (mel_out) memcpy(mel_out, mel, sizeof(double) * NMELS * nframes);
    free(mel); free(power); free(ad); free(af); free(pcm);
    return warn;
}

int orc_prepare_go(const uint8_t *pcm, size_t nbytes, float *out)
{
    return orc_prepare_go_stages(pcm, nbytes, NULL, NULL, out);
}

/* Batch helper: windows k = bytes [k*stride, k*stride + 192000) clipped to total. */
void orc_prepare_go_batch(const uint8_t *pcm, size_t total_bytes, size_t n_windows, size_t stride, float *out)
{
    orc_init();
#pragma omp parallel for schedule(dynamic)
    for (long k = 0; k < (long)n_windows; k++) {
        size_t off = (size_t)k * stride;
        size_t avail = off < total_bytes ? total_bytes - off : 0;
        if (avail > SAMPLE_SIZE) avail = SAMPLE_SIZE;
        orc_prepare_go(pcm + off, avail, out + (size_t)k * NMELS * SPEC_FRAMES);
    }
}

/* ------------------------------------------------------------------------ */
/* A11 audio_to_spectrogram  (ai/model.py:184-204), librosa>=0.10 semantics.  */
/*   audio: n float32 samples (normalize_audio output, ai/model.py:163-168).  */
/*   out: [128][n_frames] f32 with n_frames = 1 + n/512 (188 for n = 96000);  */
/*   returns n_frames.  out must hold 128*(1+n/512) floats.                   */
/* ------------------------------------------------------------------------ */
int orc_prepare_librosa(const float *audio, size_t n, float *out)
{
    orc_init();
    int nframes = 1 + (int)(n / HOP);
    float *pw = (float *)malloc(sizeof(float) * (size_t)NBINS * nframes); /* |stft|^2 as float32 [bin][frame] */
    double re[NFFT], im[NFFT];
    for (int t = 0; t < nframes; t++) {
        /* center=True: y padded by n_fft/2 zeros each side (pad_mode="constant") */
        long start = (long)t * HOP - NFFT / 2;
        for (int j = 0; j < NFFT; j++) {
            long s = start + j;
            double x = (s >= 0 && s < (long)n) ? (double)audio[s] : 0.0;
            re[j] = x * g_hann_per[j]; /* float64 window * float32 frame -> float64 */
            im[j] = 0.0;
        }
        fft_c128(re, im, NFFT, g_twr, g_twi);
        for (int k = 0; k < NBINS; k++) {
            float fr = (float)re[k], fi = (float)im[k];   /* stored as complex64 */
            float mag = hypotf(fr, fi);                   /* np.abs(complex64) */
            pw[(size_t)k * nframes + t] = mag * mag;      /* ** 2 */
        }
    }
    /* s_mel = np.dot(MEL_BASIS, stft)  (float32) */
    float *mel = (float *)malloc(sizeof(float) * (size_t)NMELS * nframes);
    float mx = 0.0f;
    for (int m = 0; m < NMELS; m++)
        for (int t = 0; t < nframes; t++) {
            float s = 0.0f;
            for (int k = 0; k < NBINS; k++) {
                float w = g_lib_mel[m][k];
                if (w != 0.0f) s += w * pw[(size_t)k * nframes + t];
            }
            mel[(size_t)m * nframes + t] = s;
            if (s > mx) mx = s;
        }
    /* power_to_db(ref=np.max): 10*log10(max(amin,S)) - 10*log10(max(amin,ref)); top_db=80 */
    const float amin = 1e-10f;
    float ref_db = 10.0f * log10f(mx > amin ? mx : amin);
    float top = -INFINITY;
    for (int i = 0; i < NMELS * nframes; i++) {
        float v = mel[i] > amin ? mel[i] : amin;
        mel[i] = 10.0f * log10f(v) - ref_db;
        if (mel[i] > top) top = mel[i];
    }
    for (int i = 0; i < NMELS * nframes; i++) {
        float v = mel[i];
        if (v < top - 80.0f) v = top - 80.0f;
        /* np.clip(s_db, -80, 0); (s_db + 80) / 80   ai/model.py:200-201 */
        if (v < -80.0f) v = -80.0f;
        if (v > 0.0f) v = 0.0f;
        out[i] = (v + 80.0f) / 80.0f;
    }
    free(mel); free(pw);
    return nframes;
}

/* Batch helper for config 5: n_clips raw s16le clips of clip_bytes each (the .dat
 * format, src/review/action.go:212-215) -> [n_clips][128][188] f32.  Follows
 * ai/train.py:41-50: open_audio_file -> normalize_audio -> audio_to_spectrogram. */
int orc_prepare_librosa_batch(const uint8_t *pcm, size_t n_clips, size_t clip_bytes, float *out)
{
    orc_init();
    size_t ns = clip_bytes / 2;
    int nframes = 1 + (int)(ns / HOP);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)n_clips; c++) {
        float *a = (float *)malloc(sizeof(float) * ns);
        orc_normalize_audio(pcm + (size_t)c * clip_bytes, clip_bytes, a);
        orc_prepare_librosa(a, ns, out + (size_t)c * NMELS * nframes);
        free(a);
    }
    return nframes;
}

/* ------------------------------------------------------------------------ */
/* A6  NoiseDetector forward, eval mode  (ai/model.py:55-117)                 */
/*   Weight blob (floats), state_dict order without num_batches_tracked:      */
/*   per conv layer l=0..11: W[Cout][Cin][3][3], b[Cout], bn.weight[Cout],    */
/*   bn.bias[Cout], bn.running_mean[Cout], bn.running_var[Cout];              */
/*   then fc.weight[C][1024], fc.bias[C].                                     */
/* ------------------------------------------------------------------------ */
static const int L_CIN[12] = {1, 32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 1024};
static const int L_COUT[12] = {32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 1024, 1024};
static const int L_STRIDE[12] = {1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2};
#define BN_EPS 1e-5f

size_t orc_blob_floats(int n_classes)
{
    size_t n = 0;
    for (int l = 0; l < 12; l++) n += (size_t)L_COUT[l] * L_CIN[l] * 9 + 5 * (size_t)L_COUT[l];
    n += (size_t)n_classes * 1024 + n_classes;
    return n;
}

/* C[M][P] = W[M][K] * col[K][P]; f32 accumulate, k ascending. */
static void gemm_rows(const float *W, const float *col, float *C, int M, int K, int P, int par)
{
#pragma omp parallel for schedule(static) if (par)
    for (int i0 = 0; i0 < M; i0 += 4) {
        int mr = M - i0 < 4 ? M - i0 : 4;
        for (int j0 = 0; j0 < P; j0 += 32) {
            int nr = P - j0 < 32 ? P - j0 : 32;
            float acc[4][32];
            memset(acc, 0, sizeof acc);
            if (mr == 4 && nr == 32) {
                for (int k = 0; k < K; k++) {
                    const float *c = col + (size_t)k * P + j0;
                    float w0 = W[(size_t)(i0 + 0) * K + k], w1 = W[(size_t)(i0 + 1) * K + k];
                    float w2 = W[(size_t)(i0 + 2) * K + k], w3 = W[(size_t)(i0 + 3) * K + k];
#pragma GCC unroll 32
                    for (int j = 0; j < 32; j++) {
                        float x = c[j];
                        acc[0][j] += w0 * x; acc[1][j] += w1 * x; acc[2][j] += w2 * x; acc[3][j] += w3 * x;
                    }
                }
            } else {
                for (int k = 0; k < K; k++) {
                    const float *c = col + (size_t)k * P + j0;
                    for (int r = 0; r < mr; r++) {
                        float w = W[(size_t)(i0 + r) * K + k];
                        for (int j = 0; j < nr; j++) acc[r][j] += w * c[j];
                    }
                }
            }
            for (int r = 0; r < mr; r++)
                for (int j = 0; j < nr; j++) C[(size_t)(i0 + r) * P + j0 + j] = acc[r][j];
        }
    }
}

/* one window: x [Cin][H][W] -> y [Cout][Ho][Wo] ; conv3x3 pad 1 + BN(eval) + ReLU */
static void conv_bn_relu(const float *x, int cin, int h, int w, int stride, const float *Wt, const float *bias,
                         const float *g, const float *beta, const float *mean, const float *var, int cout,
                         float *y, float *col, int par)
{
    int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1, P = ho * wo, K = cin * 9;
    /* im2col: col[(c*9 + kh*3 + kw)][oh*wo + ow] -- matches W[Cout][Cin][3][3] flattening */
#pragma omp parallel for schedule(static) if (par)
    for (int ck = 0; ck < K; ck++) {
        int c = ck / 9, kh = (ck % 9) / 3, kw = ck % 3;
        float *dst = col + (size_t)ck * P;
        for (int oh = 0; oh < ho; oh++) {
            int ih = oh * stride + kh - 1;
            for (int ow = 0; ow < wo; ow++) {
                int iw = ow * stride + kw - 1;
                dst[oh * wo + ow] = (ih >= 0 && ih < h && iw >= 0 && iw < w) ? x[((size_t)c * h + ih) * w + iw] : 0.0f;
            }
        }
    }
    gemm_rows(Wt, col, y, cout, K, P, par);
    for (int o = 0; o < cout; o++) {
        float invstd = 1.0f / sqrtf(var[o] + BN_EPS);
        float *row = y + (size_t)o * P;
        for (int p = 0; p < P; p++) {
            float v = row[p] + bias[o];
            v = (v - mean[o]) * invstd * g[o] + beta[o];
            row[p] = v > 0.0f ? v : 0.0f;
        }
    }
}

static void cnn_one(const float *blob, int n_classes, const float *spec, float *logits, float *bufA, float *bufB,
                    float *col, int par, float *feat_out)
{
    const float *p = blob;
    int h = NMELS, w = SPEC_FRAMES;
    const float *x = spec;
    float *y = bufA;
    for (int l = 0; l < 12; l++) {
        int cin = L_CIN[l], cout = L_COUT[l], s = L_STRIDE[l];
        const float *Wt = p; p += (size_t)cout * cin * 9;
        const float *b = p; p += cout;
        const float *g = p; p += cout;
        const float *be = p; p += cout;
        const float *mu = p; p += cout;
        const float *va = p; p += cout;
        conv_bn_relu(x, cin, h, w, s, Wt, b, g, be, mu, va, cout, y, col, par);
        h = (h + 2 - 3) / s + 1; w = (w + 2 - 3) / s + 1;
        x = y; y = (y == bufA) ? bufB : bufA;
    }
    /* AdaptiveMaxPool2d(1) + Flatten (+ Dropout = identity in eval) + Linear  ai/model.py:105-110 */
    float feat[1024];
    int P = h * w;
    for (int c = 0; c < 1024; c++) {
        float m = x[(size_t)c * P];
        for (int q = 1; q < P; q++) if (x[(size_t)c * P + q] > m) m = x[(size_t)c * P + q];
        feat[c] = m;
    }
    if (feat_out) memcpy(feat_out, feat, sizeof feat);
    const float *fw = p; p += (size_t)n_classes * 1024;
    const float *fb = p;
    for (int o = 0; o < n_classes; o++) {
        float s = 0.0f;
        for (int c = 0; c < 1024; c++) s += fw[(size_t)o * 1024 + c] * feat[c];
        logits[o] = s + fb[o];
    }
}

#define ACT_MAX ((size_t)128 * 64 * 94) /* largest activation: 128x64x94 == 32x128x188 floats */
#define COL_MAX ((size_t)576 * 64 * 94) /* largest im2col: layer 3, K=576, P=6016 */

int orc_cnn_forward(const float *blob, int n_classes, const float *spec, int batch, float *logits)
{
    if (batch <= 0) return 0;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    if (batch == 1 || nthreads == 1) {
        float *a = (float *)malloc(sizeof(float) * ACT_MAX), *b = (float *)malloc(sizeof(float) * ACT_MAX);
        float *col = (float *)malloc(sizeof(float) * COL_MAX);
        if (!a || !b || !col) return -1;
        for (int i = 0; i < batch; i++)
            cnn_one(blob, n_classes, spec + (size_t)i * NMELS * SPEC_FRAMES, logits + (size_t)i * n_classes, a, b, col, 1, NULL);
        free(a); free(b); free(col);
        return 0;
    }
    int fail = 0;
#pragma omp parallel
    {
        float *a = (float *)malloc(sizeof(float) * ACT_MAX), *b = (float *)malloc(sizeof(float) * ACT_MAX);
        float *col = (float *)malloc(sizeof(float) * COL_MAX);
        if (!a || !b || !col) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic)
            for (int i = 0; i < batch; i++)
                cnn_one(blob, n_classes, spec + (size_t)i * NMELS * SPEC_FRAMES, logits + (size_t)i * n_classes, a, b, col, 0, NULL);
        }
        free(a); free(b); free(col);
    }
    return fail ? -1 : 0;
}

/* single window, also returns the 1024 pooled features (debug aid for GPU parity) */
int orc_cnn_forward_feat(const float *blob, int n_classes, const float *spec, float *logits, float *feat)
{
    float *a = (float *)malloc(sizeof(float) * ACT_MAX), *b = (float *)malloc(sizeof(float) * ACT_MAX);
    float *col = (float *)malloc(sizeof(float) * COL_MAX);
    if (!a || !b || !col) return -1;
    cnn_one(blob, n_classes, spec, logits, a, b, col, 1, feat);
    free(a); free(b); free(col);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* A7  softmax  (src/model/model.go:232-254) : f64, max-subtracted            */
/* ------------------------------------------------------------------------ */
void orc_softmax(const float *logits_f32, int c, double *probs)
{
    double mx = -1.7976931348623157e308; /* -math.MaxFloat64 */
    for (int i = 0; i < c; i++) { double v = (double)logits_f32[i]; if (v > mx) mx = v; } /* model.go:213-216 */
    double sum = 0.0;
    for (int i = 0; i < c; i++) { probs[i] = exp((double)logits_f32[i] - mx); sum += probs[i]; }
    for (int i = 0; i < c; i++) probs[i] /= sum;
}

/* ------------------------------------------------------------------------ */
/* A8  decision (src/daemon/daemon.go:189-207).  The Go loop ranges over a    */
/* map (order undefined) with strict '>' from ("",0.0); ties are therefore    */
/* undefined in the reference -- here the lowest class index wins.            */
/* label = -1 means bestClass == "" (no prob > 0).  Returns match (0/1).      */
/* ------------------------------------------------------------------------ */
int orc_decide(const double *probs, int c, int empty_class, double trust, int *label, double *conf)
{
    int best = -1; double bc = 0.0;
    for (int i = 0; i < c; i++) if (probs[i] > bc) { bc = probs[i]; best = i; }
    *label = best; *conf = bc;
    return (best >= 0 && best != empty_class && bc > trust) ? 1 : 0;
}

/* ------------------------------------------------------------------------ */
/* Whole path over n windows of one contiguous PCM buffer (A9 windowing:      */
/* window k = bytes [k*stride, k*stride+192000), daemon.go:104-125,           */
/* ai/inspect.py:106-109).  Outputs: spec (optional) [n][128][188], logits    */
/* (optional) [n][C], probs [n][C] f64, label[n], match[n].                   */
/* ------------------------------------------------------------------------ */
int orc_score(const float *blob, int n_classes, const uint8_t *pcm, size_t total_bytes, size_t n_windows,
              size_t stride, double trust, int empty_class, float *spec_out, float *logits_out, double *probs,
              int32_t *label, uint8_t *match)
{
    float *spec = spec_out ? spec_out : (float *)malloc(sizeof(float) * n_windows * NMELS * SPEC_FRAMES);
    float *logits = logits_out ? logits_out : (float *)malloc(sizeof(float) * n_windows * n_classes);
    if (!spec || !logits) return -1;
    orc_prepare_go_batch(pcm, total_bytes, n_windows, stride, spec);
    int rc = orc_cnn_forward(blob, n_classes, spec, (int)n_windows, logits);
    if (rc == 0) {
        for (size_t k = 0; k < n_windows; k++) {
            double conf; int lab;
            orc_softmax(logits + k * n_classes, n_classes, probs + k * n_classes);
            match[k] = (uint8_t)orc_decide(probs + k * n_classes, n_classes, empty_class, trust, &lab, &conf);
            label[k] = lab;
        }
    }
    if (!spec_out) free(spec);
    if (!logits_out) free(logits);
    return rc;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
