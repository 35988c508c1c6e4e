/*
 * apr.h -- C ABI of the MI355X-native DTrack detection engine (libapr_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of audio-pattern-ranger/apr:
 *   PCM (48 kHz mono s16le, 2-s windows, 1-s hop) -> STFT -> log-mel -> NoiseDetector
 *   CNN -> softmax -> threshold/label.
 * Every entry point cites the reference interface it replaces (paths relative to
 * the reference repository).  Conventions: extern "C", plain pointers and sizes,
 * return 0 (APR_OK) or a negative error code (the prepare calls return a
 * non-negative warn-flag mask), no exceptions cross this boundary, every output
 * buffer is caller-allocated, nothing passed in is retained after the call
 * returns.  One engine handle owns one GPU stream and is safe for ONE caller at
 * a time (the daemon pattern, src/daemon/daemon.go:93-99, is one handle per
 * model goroutine plus one front-end-only handle for Prepare).
 */
#ifndef APR_H
#define APR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants: src/ffmpeg/ffmpeg.go:20-21, src/model/model.go:23-40 ---- */
#define APR_SAMPLE_RATE        48000
#define APR_BYTES_PER_SECOND   96000
#define APR_SEGMENT_SIZE       2
#define APR_SAMPLE_SIZE        192000   /* bytes in one 2-s check window        */
#define APR_NMELS              128
#define APR_NFFT               2048
#define APR_HOP_LENGTH         512
#define APR_SPECTROGRAM_FRAMES 188
#define APR_SPEC_ELEMS         (APR_NMELS * APR_SPECTROGRAM_FRAMES)
#define APR_MAX_CLASSES        64

/* error codes */
#define APR_OK            0
#define APR_ERR_INVALID  -1   /* bad argument                                   */
#define APR_ERR_DEVICE   -2   /* HIP runtime error / no such device             */
#define APR_ERR_MODEL    -3   /* model bytes are not a NoiseDetector graph      */
#define APR_ERR_NOMEM    -4
#define APR_ERR_NOMODEL  -5   /* CNN entry point called on a front-end-only engine */

/* front-end flavour: the reference has two, numerically different */
#define APR_FLAVOUR_GO      0 /* src/model/model.go:86-184 + src/model/dsp.go   */
#define APR_FLAVOUR_LIBROSA 1 /* ai/model.py:184-204 (training / ai.inspect)    */

/* warn flags returned by the prepare calls (log.Warn sites model.go:89,94)     */
#define APR_WARN_UNDERFLOW 1  /* "Audio Underflow; Segment was not large enough!" */
#define APR_WARN_OVERFLOW  2  /* "Audio Overflow; Segment was too large!"        */

typedef struct apr_engine apr_engine;

/* Number of HIP devices visible (0 if none / runtime unusable). */
int apr_device_count(void);

/* Library version string. */
const char *apr_version(void);

/*
 * Create an engine on GPU `device_id`.
 *   model_bytes/n : the bytes model.Load keeps in OnnxModel.RawBytes
 *                   (src/model/model.go:54-83): an ONNX file as exported by
 *                   ai/model.py:143-152 -- [Conv(+folded BatchNorm) -> Relu] x12,
 *                   MaxPool(kernel_shape [2,3] = the last conv's whole output map,
 *                   pads 0), Flatten, Gemm(transB 1); structure of a real export:
 *                   tests/golden/onnx_real_export.json -- or this repo's flat "APRW"
 *                   blob (apr_amd/weights.py).  Also accepted: explicit
 *                   BatchNormalization nodes and GlobalMaxPool as the pool.  A
 *                   MaxPool that is not the whole map, a graph input that is not
 *                   [N,1,128,188], or a node list that is not one chain is
 *                   APR_ERR_MODEL with the reason.  NULL/0 creates a
 *                   front-end-only engine (enough for Prepare).
 *   max_batch     : windows processed per GPU pass (activation buffers are
 *                   sized for it: ~11 MB of device memory per window: 45 GB
 *                   at 4096, 180 GB at the cap of 16384, which is worth 1.9 %
 *                   of throughput over 4096); <=0 picks the default (1024).  Engines of >= 128 windows run the stride-1
 *                   layers as Winograd F(4x4,3x3).  max_batch <= 4
 *                   makes a latency engine for the daemon's one-clip-per-Infer
 *                   pattern (src/daemon/daemon.go:132-136): the layers whose grid
 *                   is a handful of tiles run split along K.  Its results agree
 *                   with a throughput engine's to fp32 rounding (another summation
 *                   order), not bit for bit; all throughput engines agree bit for
 *                   bit with each other whatever their max_batch.
 * Replaces the per-call `gorgonnx.NewGraph()` + `UnmarshalBinary` of
 * src/model/model.go:190-196: parsing and upload happen once, here.
 */
int apr_engine_create(const void *model_bytes, size_t n, int device_id, int max_batch, apr_engine **out);
/* Same, and on failure the reason is also copied into errbuf (NUL-terminated, truncated to errbuf_len): the text the
 * Go shim hands to log.Die("could not unmarshal ONNX model: %s") (src/model/model.go:194-196).  apr_last_error(NULL)
 * keeps the reason per OS thread, which a goroutine may have left by its next cgo call; this form does not depend on it. */
int apr_engine_create_ex(const void *model_bytes, size_t n, int device_id, int max_batch, apr_engine **out, char *errbuf,
                         size_t errbuf_len);
void apr_engine_destroy(apr_engine *e);

/*
 * Host-only validation of model bytes (no GPU touched): parses exactly as
 * apr_engine_create would and reports the class count and conv-layer count.
 * On failure returns APR_ERR_MODEL and apr_last_error(NULL) says why -- the
 * text the Go shim passes to log.Die("could not unmarshal ONNX model: %s")
 * (src/model/model.go:194-196).
 */
int apr_model_info(const void *model_bytes, size_t n, int *n_classes, int *n_conv_layers);

/* Number of classes C of the loaded model (0 for a front-end-only engine). */
int apr_engine_num_classes(const apr_engine *e);

/* Last error text of this engine (or of the last failed create when e == NULL). */
const char *apr_last_error(const apr_engine *e);

/*
 * model.Prepare (src/model/model.go:86): s16le bytes -> [128][188] float32
 * spectrogram (row-major, the backing array of the reference's [1,1,128,188]
 * tensor).  Pad/truncate semantics of model.go:87-96.  Host pointers.
 * Returns a warn-flag mask >= 0, or a negative error.
 * flavour APR_FLAVOUR_LIBROSA replaces ai.model.audio_to_spectrogram
 * (ai/model.py:184) on normalize_audio(pcm) for a 192000-byte clip.
 */
int apr_prepare(apr_engine *e, int flavour, const uint8_t *pcm_s16le, size_t nbytes, float *out_128x188);

/*
 * Batched Prepare: window k = bytes [k*stride, k*stride + 192000) of `pcm`
 * clipped to total_bytes (short tail windows are zero-padded like model.go:88-92).
 * stride 96000 reproduces the daemon's / ai.inspect's 2-s window with 1-s hop
 * (src/daemon/daemon.go:104-125, ai/inspect.py:106-109); stride 192000 walks a
 * packed array of .dat clips (ai/train.py:36-52, BASELINE config 5).
 * out: [n_windows][128][188] float32, host.
 */
int apr_prepare_batch(apr_engine *e, int flavour, const uint8_t *pcm, size_t total_bytes, size_t n_windows,
                      size_t stride_bytes, float *out);

/*
 * The CNN half of model.Infer (src/model/model.go:188-209): `batch` prepared
 * spectrograms [batch][128][188] -> raw logits [batch][C] float32 (what
 * outputTensors[0] holds).  Host pointers.  The f64 softmax + label map of
 * model.go:211-228 stays on the caller's side (Go shim / apr_amd.model.Infer),
 * exactly as in the reference.
 */
int apr_infer(apr_engine *e, const float *spec, int batch, float *out_logits);

/*
 * Fused batch path: Prepare (go flavour) + Infer + softmax + the daemon's
 * decision rule (src/daemon/daemon.go:189-207) for n_windows windows of one PCM
 * buffer (window k as in apr_prepare_batch).
 *   trust       : inspect_trust (src/state/config.go:86), match needs conf > trust
 *   empty_class : index of the "empty" label, or -1 if the model has none
 *   probs       : [n_windows][C] float64  (the values of Infer's map)
 *   label       : [n_windows] argmax class index (-1 if no prob > 0)
 *   match       : [n_windows] 1 iff label != empty_class && conf > trust
 * Any of probs/label/match may be NULL.  Host pointers.
 */
int apr_score(apr_engine *e, const uint8_t *pcm, size_t total_bytes, size_t n_windows, size_t stride_bytes,
              double trust, int empty_class, double *probs, int32_t *label, uint8_t *match);

/*
 * Live mode, the tick of src/daemon/daemon.go:104-125 for n_streams concurrent streams: every stream delivers one
 * 1-s segment (APR_BYTES_PER_SECOND bytes) per tick and is scored on [its previous segment || the new one].  The
 * engine keeps the previous segments on the GPU, so a tick uploads only the new second of every stream.
 *   segments : [n_streams][APR_BYTES_PER_SECOND] bytes, host.  If it is the pointer apr_stream_buffer returned, the
 *              segments are taken from there without a copy (the reader of the streams deposits them in that pinned
 *              memory); any other host pointer is copied into it first.
 * The first tick (after create, after apr_stream_reset, or when n_streams changes) has no previous segment: it only
 * primes the buffers, writes no output and returns APR_STREAM_PRIMED, as the daemon drops a stream's first segment
 * (daemon.go:115-119).  Later ticks return 0 and write probs/label/match as apr_score does ([n_streams] rows; any may be
 * NULL).  1 <= n_streams <= max_batch.  Synchronous.
 */
#define APR_STREAM_PRIMED 1
int apr_stream_tick(apr_engine *e, const uint8_t *segments, int n_streams, double trust, int empty_class, double *probs,
                    int32_t *label, uint8_t *match);
/* Pinned host memory for the segments of a tick ([n_streams][APR_BYTES_PER_SECOND]); owned by the engine, valid until a
 * call with a larger n_streams or apr_engine_destroy.  NULL on error. */
uint8_t *apr_stream_buffer(apr_engine *e, int n_streams);
/* Forget the previous segments: the next tick primes again (the streams restarted; no counterpart in the reference, whose buffer lives in start_scanners, daemon.go:104-125). */
int apr_stream_reset(apr_engine *e);

/*
 * Same as apr_score with every buffer already resident in this engine's GPU
 * memory (hipMalloc'd / torch device tensors).  Asynchronous on the engine's
 * stream; call apr_sync before reading the outputs.  d_logits (optional)
 * receives the raw f32 logits.  This is what bench.py times.
 */
int apr_score_device(apr_engine *e, const uint8_t *d_pcm, size_t total_bytes, size_t n_windows,
                     size_t stride_bytes, double trust, int empty_class, double *d_probs, int32_t *d_label,
                     uint8_t *d_match, float *d_logits);

/* Device-resident batched Prepare (BASELINE config 5).  Asynchronous. */
int apr_prepare_batch_device(apr_engine *e, int flavour, const uint8_t *d_pcm, size_t total_bytes,
                             size_t n_windows, size_t stride_bytes, float *d_out);

/*
 * apr_prepare_batch_device for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  Same arguments, same kernels and bit-identical
 * output; the front end's one scratch buffer (mel power, [n_windows][188][128] f32) is a file-scope device array of
 * the library instead of engine memory, and the call itself performs no hipMalloc / hipFree / event creation (the
 * engine was created earlier, outside such a region).  Limit: n_windows <= APR_STATIC_MAX_WINDOWS, larger calls
 * are rejected with APR_ERR_INVALID (no chunking: split the call).  Every GPU the library runs on holds one copy of
 * the array (5.9 MiB) until the process ends, and EVERY engine on a GPU shares it: two calls that overlap on one GPU
 * (two engines, two streams) race.  Asynchronous on the engine's stream.
 */
#define APR_STATIC_MAX_WINDOWS 64
int apr_prepare_batch_device_static(apr_engine *e, int flavour, const uint8_t *d_pcm, size_t total_bytes,
                                    size_t n_windows, size_t stride_bytes, float *d_out);

/* Device-resident CNN: d_spec [batch][128][188] -> d_logits [batch][C].  Asynchronous. */
int apr_infer_device(apr_engine *e, const float *d_spec, int batch, float *d_logits);

/*
 * Infer + decision for spectrograms that are already prepared: the CNN, the f64 softmax
 * (src/model/model.go:211-228) and the daemon's rule (src/daemon/daemon.go:189-207) for `batch`
 * spectrograms [batch][128][188].  This is the per-model half of the daemon's fan-out
 * (src/daemon/daemon.go:132-136: ONE prepared tensor is handed to every configured model;
 * ai/inspect.py:133-159 loops the models over one spectrogram the same way): prepare once with
 * apr_prepare_batch[_device], then call this on each model's engine.
 * apr_classify takes host pointers and is synchronous; apr_classify_device takes GPU pointers and
 * is asynchronous on the engine's stream.  Any output may be NULL.
 */
int apr_classify(apr_engine *e, const float *spec, int batch, double trust, int empty_class, double *probs,
                 int32_t *label, uint8_t *match, float *logits);
int apr_classify_device(apr_engine *e, const float *d_spec, int batch, double trust, int empty_class,
                        double *d_probs, int32_t *d_label, uint8_t *d_match, float *d_logits);

/*
 * Number of 2-s windows a recording of `nbytes` yields at hop `stride_bytes`:
 * len(range(0, nbytes - stride, stride)) of ai/inspect.py:108-109 (T whole seconds -> T-1 windows,
 * the daemon's count too: src/daemon/daemon.go:115-119).
 */
size_t apr_windows_in(size_t nbytes, size_t stride_bytes);

/*
 * Sweep over a list of recordings held in host memory: the loop of ai/inspect.py:73-112 (and of the
 * daemon, src/daemon/daemon.go:104-136) for n_files buffers at once.  File f contributes
 * apr_windows_in(nbytes[f], stride_bytes) windows, none of which crosses into another file; rows are
 * written file after file, first_window[f] (n_files + 1 entries, optional) giving each file's first
 * row.  Files are packed back to back into full GPU passes (a long file spans several passes, short
 * files share one), through two pinned staging buffers: the host->device copy of pass p+1 runs on a
 * copy stream while pass p computes.  probs/label/match as in apr_score; host pointers; synchronous.
 * A call that fails while allocating its staging is resumed by the next call (nothing leaks); the stride may change
 * between calls (the staging is re-sized).
 */
int apr_sweep_files(apr_engine *e, int n_files, const uint8_t *const *pcm, const size_t *nbytes, size_t stride_bytes,
                    double trust, int empty_class, double *probs, int32_t *label, uint8_t *match, size_t *first_window);

/*
 * The same sweep over the GPUs of one node from ONE process: the directory loop of ai/inspect.py:20-70,73-112 /
 * the scanner fan-out of src/daemon/daemon.go:87-141 sharded BY FILE.  `engines` lists one engine per device to use
 * (created by the caller with apr_engine_create on the device ids it wants -- the device list of the sweep -- all from
 * the same model bytes; distinct handles); file f goes to engines[f % n_engines] (round-robin by index, so no window
 * crosses a shard).  One host thread per engine stages that engine's files through its own pinned buffers; every row
 * lands directly in the caller's host arrays at first_window[f] + k, i.e. in FILE order whatever device scored it.
 * Because the destination is host memory, each GPU's rows cross its own PCIe link once and there is no device-side
 * gather (that is for device-resident results of one process per GPU: apr_amd/sweep.py gather_scores over RCCL).
 * n_engines == 1 is bit-identical to apr_sweep_files.  On failure the reason is in apr_last_error(engines[0]).
 * apr_sweep_plan is the host-only plan (no GPU): file -> index into the device list, first row of every file
 * (n_files + 1 entries), windows per device (n_devices entries); any output may be NULL.
 */
int apr_sweep_plan(int n_files, const size_t *nbytes, size_t stride_bytes, int n_devices, int32_t *file_device,
                   size_t *first_window, size_t *device_windows);
int apr_sweep_files_multi(apr_engine *const *engines, int n_engines, int n_files, const uint8_t *const *pcm,
                          const size_t *nbytes, size_t stride_bytes, double trust, int empty_class, double *probs,
                          int32_t *label, uint8_t *match, size_t *first_window);

/*
 * Offset of the PCM payload in a RIFF/WAVE byte stream (0 if `buf` is not RIFF or holds no data
 * chunk yet); fills the fmt fields when non-NULL.  The reference feeds `ffmpeg -f wav -` straight
 * into its 96000-byte segment reader without skipping the header (src/ffmpeg/ffmpeg.go:111-114,
 * src/daemon/daemon.go:151-154); a caller that wants the first window clean skips this many bytes.
 * Host only, no GPU.
 */
size_t apr_wav_data_offset(const uint8_t *buf, size_t n, uint32_t *sample_rate, uint16_t *channels, uint16_t *bits_per_sample);

/*
 * Integer-factor upsampling of s16 PCM on the GPU (16 kHz sources -> the 48 kHz this path scores,
 * up = 3).  In the reference this is ffmpeg's `-ar 48000 -ac 1` inside both decode commands
 * (src/ffmpeg/ffmpeg.go:76,113-114; ai/inspect.py:99-100); libswresample cannot be matched bit for
 * bit, so this step is outside the parity claim and uses the library's own filter:
 *   out[up m + p] = round_half_even_sat16( sum_{k < tpp} taps[p][k] * in[m + tpp/2 - k] ),  in[] = 0 outside
 * in f64 with one rounded multiply and add per tap (oracle/resample.py reproduces the bits).
 * apr_resample_taps fills taps[up][taps_per_phase] (host memory) with a Kaiser(beta)-windowed sinc whose
 * cutoff is `cutoff` x the input Nyquist, each phase scaled to sum 1; taps_per_phase even,
 * up * taps_per_phase <= 1024.  Suggested: up 3, taps_per_phase 32, cutoff 0.97, beta 9.0.
 * apr_upsample: host pointers, out holds n_in * up samples; apr_upsample_device: device pointers
 * (taps stay in host memory).  Both return after the work is done.
 */
int apr_resample_taps(int up, int taps_per_phase, double cutoff, double beta, double *taps);
int apr_upsample(apr_engine *e, const int16_t *in, size_t n_in, int up, const double *taps, int taps_per_phase, int16_t *out);
int apr_upsample_device(apr_engine *e, const int16_t *d_in, size_t n_in, int up, const double *taps, int taps_per_phase, int16_t *d_out);

/*
 * Report bucketing: hits[b][c] = 1 iff some window k in [b*wpb, (b+1)*wpb) has match[k] and
 * label[k] == c -- "1 clap or 999 claps within 1 minute is logged as one hit"
 * (docs/usage/review.md:48-53; the reference's code does not implement it).  wpb = 60 at the 1-s hop.
 * hits: [ceil(n_windows / wpb)][n_classes] bytes, caller-allocated.  Host only, no GPU.
 */
int apr_bucket_hits(const int32_t *label, const uint8_t *match, size_t n_windows, int n_classes, size_t windows_per_bucket,
                    uint8_t *hits);
/* The same on the GPU, for (label, match) rows that are still in HBM -- the outputs of apr_score_device / apr_classify_device
 * or the gathered rows of a multi-GPU sweep: only [buckets][n_classes] bytes then travel to the host.  Device pointers;
 * every byte of d_hits is written; n_classes <= APR_MAX_CLASSES.  Asynchronous on the engine's stream. */
int apr_bucket_hits_device(apr_engine *e, const int32_t *d_label, const uint8_t *d_match, size_t n_windows, int n_classes,
                           size_t windows_per_bucket, uint8_t *d_hits);

/* Block until all work queued on this engine's stream has finished. */
int apr_sync(apr_engine *e);

/* The hipStream_t this engine launches on (as void*), for event timing by the caller. */
void *apr_engine_stream(apr_engine *e);

/*
 * Per-kernel timing with HIP events on the engine's stream (no reference
 * counterpart; measurement hook for bench.py).  While enabled, every kernel
 * launch is bracketed by events; apr_profile_read syncs and accumulates.
 */
#define APR_PROF_SLOTS 48
typedef struct apr_profile_slot {
    char     name[48];     /* kernel (family) name as rocprofv3 reports it     */
    int32_t  layer;        /* 1..12 for conv layers, -1 otherwise               */
    int32_t  reserved;
    double   ms;           /* summed device time                                */
    uint64_t launches;
    double   flops;        /* algorithmic FLOPs issued by those launches        */
    double   bytes;        /* algorithmic HBM bytes of those launches           */
} apr_profile_slot;
int apr_profile_enable(apr_engine *e, int on);
/* Diagnostic with no counterpart in the reference: the fp32 matrix rate this GPU delivers right now.  Every SIMD issues
 * v_mfma_f32_32x32x2_f32 back to back from registers (no memory traffic) for a few milliseconds on the engine's stream;
 * *tflops = the rate by HIP events.  bench.py prints it beside the roofline so that a throttled box (one of round 2's ran
 * every kernel 15 % slower) can be told from a slower build. */
int apr_mfma_probe(apr_engine *e, double *tflops);
int apr_profile_reset(apr_engine *e);
int apr_profile_read(apr_engine *e, apr_profile_slot *slots /*[APR_PROF_SLOTS]*/, int *n_slots);

/*
 * Device memory for hosts that have no allocator of their own (the Go shim; a Python host uses torch tensors): plain
 * hipMalloc / hipFree / hipMemcpy on the engine's GPU, so that a cgo caller can keep PCM and score rows device-resident between
 * apr_score_device, apr_gather_rows_device and apr_bucket_hits_device.  No counterpart in the reference (its path is CPU-only).
 * apr_device_upload / _download are synchronous on the engine's stream (they return when the copy is done).
 */
int apr_device_alloc(apr_engine *e, size_t nbytes, void **d_out);
int apr_device_free(apr_engine *e, void *d_ptr);
int apr_device_upload(apr_engine *e, void *d_dst, const void *h_src, size_t nbytes);
int apr_device_download(apr_engine *e, void *h_dst, const void *d_src, size_t nbytes);

/*
 * The final gather of a multi-GPU sweep for a ONE-process host (north_star: recordings shard by file over the GPUs of a node,
 * "RCCL over xGMI used only for the final gather of per-clip scores").  engines[i] is the engine of GPU i (all
 * distinct devices), n_rows[i] the windows it scored, d_probs[i] / d_label[i] / d_match[i] its device-resident rows as
 * apr_score_device wrote them ([n_rows[i]][n_classes] f64, i32, u8; a tensor that is not wanted is NULL for every engine and in
 * the outputs).  The rows of engine i land at row offset n_rows[0] + .. + n_rows[i-1] of the *_out buffers on the GPU of
 * engines[root]: engine order = the shard order of apr_sweep_plan.  Every sender uses its own point-to-point link to the root
 * (ncclSend / ncclRecv in one group -- a flat gather, not a ring: SURVEY 5), the root's own rows are a device-to-device copy.
 * The call returns when every stream involved has finished.  librccl is loaded on first use (dlopen "librccl.so.1"), the
 * communicator of a device list is created once per process (ncclCommInitAll) and kept.  The reference has no counterpart
 * (single process, CPU); the one-process-per-GPU form of the same gather is apr_amd/sweep.py: gather_scores (torch.distributed).
 * Measured on hardware with one GPU only (tests/test_gpu_configs.py); N > 1 is by construction.
 */
int apr_gather_rows_device(apr_engine *const *engines, int n_engines, int root, const size_t *n_rows, int n_classes,
                           const double *const *d_probs, const int32_t *const *d_label, const uint8_t *const *d_match,
                           double *d_probs_out, int32_t *d_label_out, uint8_t *d_match_out);

/*
 * Deterministic synthetic PCM (host side, no GPU): SURVEY 8(d) recipe.  A 16 kHz
 * signal (noise at -30 dBFS, 0-3 Hann-shaped tone/chirp bursts per second,
 * exact-zero stretches) interpolated x3 to 48 kHz, s16 saturated.  file seed =
 * mix(base_seed, file_index); any sample range of a file can be generated
 * independently (sample_offset is in 48 kHz samples, multiple of 48000).
 */
int apr_synth_pcm(uint64_t base_seed, uint64_t file_index, uint64_t second_offset, size_t n_seconds,
                  int16_t *out_48k);

#ifdef __cplusplus
}
#endif
#endif /* APR_H */
