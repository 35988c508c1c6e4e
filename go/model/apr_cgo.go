// apr_cgo.go -- the cgo binding that moves dtrack/model's DSP and CNN onto an MI355X through
// libapr_amd.so (include/apr.h).
//
// How it is applied to the reference tree (see INTEGRATION.md):
//   * src/model/dsp.go is deleted (the mel lens lives on the GPU),
//   * in src/model/model.go the bodies of Prepare (model.go:86-184) and Infer
//     (model.go:188-229) and the go-dsp / onnx-go imports are removed; the constants,
//     OnnxModel, Train, Load, softmax and normalizeAudio stay exactly as they are,
//   * this file is added next to it and supplies Prepare, Infer and the new ScoreBatch,
//     SweepFiles, SweepFilesMulti, SweepFilesGatherHits, WavDataOffset and BucketHits (go 1.24 in src/go.mod:3 has runtime.Pinner).
// The exported surface of the package is therefore unchanged; every number is computed
// behind the C ABI except the f64 softmax + label map, which remain the reference's own.
//
// NOTE: the build image has no Go toolchain, so this file has not been compiled here.
package model

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../apr_amd -lapr_amd
#include <stdlib.h>
#include "apr.h"
// byte offset into device memory (a device address is not a Go pointer: the arithmetic stays in C)
static inline void *apr_at(void *p, size_t off) { return (char *)p + off; }
*/
import "C"

import (
	"dtrack/ffmpeg"
	"dtrack/log"

	"crypto/sha256"
	"runtime"
	"sort"
	"sync"
	"unsafe"

	"gorgonia.org/tensor"
)

// ---- engine cache -------------------------------------------------------------------------
// One front-end engine for Prepare (single caller goroutine, daemon.go:123) and one engine per
// (model content, max_batch, device), each guarded by its own mutex (a handle is safe for one
// caller at a time; the daemon runs one scanner goroutine per model).  The key is the sha256 of
// RawBytes, not its address: a model re-loaded after retraining has the same length and may be
// handed the old address by the allocator.  The digest of a slice is memoised by (address, len)
// together with the slice itself, so that address cannot be reused while the memo entry lives.
// The memo is BOUNDED (maxDigests entries, oldest dropped first): a daemon that re-loads its models after
// every retraining without calling Unload would otherwise keep 94 MB of every file it ever loaded alive.
// A dropped entry only costs one more sha256 of that model on its next call.
type engine struct {
	mu sync.Mutex
	h  *C.apr_engine
	nc int
}

type engineKey struct {
	digest   [32]byte
	maxBatch int
	device   int
}

type digestMemo struct {
	raw    []byte // keeps the backing array alive
	digest [32]byte
}

var (
	cacheMu   sync.Mutex
	frontEnds = map[int]*engine{}       // Prepare, Upsample3: one per device
	engines   = map[engineKey]*engine{} // Infer (max_batch 1), ScoreBatch / SweepFiles (batchWindows), StreamScorer (n streams)
	digests   = map[uintptr]digestMemo{}
	digestAge []uintptr // insertion order of digests' keys, oldest first
	device    = 0       // guarded by cacheMu: SetDevice / currentDevice
	// windows per GPU pass of ScoreBatch / SweepFiles / SweepFilesMulti (guarded by cacheMu: SetBatchWindows / currentBatchWindows).
	// The engine holds ~11 MB of device memory per window: 45 GB at 4096; a host that owns the whole GPU can ask for the cap of
	// 16384 (180 GB), which bench.py measures 1.9 % faster (fewer, larger launches).
	batchWindows = 4096
)

const maxDigests = 8 // models whose bytes the memo may keep alive (inspect_models is a handful)

// SetBatchWindows sets the pass size of the engines created from now on by ScoreBatch / SweepFiles[Multi] (128 .. 16384).
func SetBatchWindows(n int) {
	if n < 128 {
		n = 128
	}
	if n > 16384 {
		n = 16384
	}
	cacheMu.Lock()
	batchWindows = n
	cacheMu.Unlock()
}

func currentBatchWindows() int {
	cacheMu.Lock()
	defer cacheMu.Unlock()
	return batchWindows
}

// SetDevice selects the GPU the package-level calls use (Prepare, Infer, ScoreBatch, SweepFiles, StreamScorer); call it
// before the first of them, e.g. from a config key.  SweepFilesMulti takes its own device list.
func SetDevice(d int) {
	cacheMu.Lock()
	device = d
	cacheMu.Unlock()
}

func currentDevice() int {
	cacheMu.Lock()
	defer cacheMu.Unlock()
	return device
}

func lastErr(h *C.apr_engine) string { return C.GoString(C.apr_last_error(h)) }

// createEngine returns the handle or the library's reason.  apr_engine_create_ex copies the reason into a caller
// buffer, so no second cgo call (possibly on another OS thread) is needed to fetch it.
func createEngine(raw []byte, device, maxBatch int) (*C.apr_engine, string) {
	var h *C.apr_engine
	var errbuf [512]C.char
	var p unsafe.Pointer
	if len(raw) > 0 {
		p = unsafe.Pointer(&raw[0])
	}
	rc := C.apr_engine_create_ex(p, C.size_t(len(raw)), C.int(device), C.int(maxBatch), &h, &errbuf[0], C.size_t(len(errbuf)))
	if rc != 0 {
		return nil, C.GoString(&errbuf[0])
	}
	return h, ""
}

func frontend() *engine {
	cacheMu.Lock()
	defer cacheMu.Unlock()
	fe := frontEnds[device]
	if fe == nil {
		h, why := createEngine(nil, device, 64)
		if h == nil {
			log.Die("GPU front end unavailable: %s", why)
		}
		fe = &engine{h: h}
		frontEnds[device] = fe
	}
	return fe
}

func engineFor(m OnnxModel) *engine { return cachedEngine(m, 1, currentDevice()) }

// batchEngineFor: the engine of ScoreBatch / SweepFiles (batchWindows windows per GPU pass).
func batchEngineFor(m OnnxModel) *engine { return cachedEngine(m, currentBatchWindows(), currentDevice()) }

func digestOf(raw []byte) [32]byte { // cacheMu held
	addr := uintptr(unsafe.Pointer(&raw[0]))
	if d, ok := digests[addr]; ok && len(d.raw) == len(raw) {
		return d.digest
	}
	d := digestMemo{raw: raw, digest: sha256.Sum256(raw)}
	if _, had := digests[addr]; !had {
		digestAge = append(digestAge, addr)
	}
	digests[addr] = d
	for len(digestAge) > maxDigests { // bounded: the oldest entry lets go of its model's bytes
		delete(digests, digestAge[0])
		digestAge = digestAge[1:]
	}
	return d.digest
}

// Infer scores one clip per call (daemon.go:132-136), so its engine is created with max_batch 1:
// a few MB of activations instead of 8 GB, and the library runs the late, tiny-grid layers split
// along K and replays the call as a captured hipGraph (p50 0.30 ms per clip instead of 1.7 ms).
func cachedEngine(m OnnxModel, maxBatch, device int) *engine {
	if len(m.RawBytes) == 0 {
		log.Die("could not unmarshal ONNX model: empty")
	}
	cacheMu.Lock()
	defer cacheMu.Unlock()
	key := engineKey{digestOf(m.RawBytes), maxBatch, device}
	if e, ok := engines[key]; ok {
		return e
	}
	// replaces gorgonnx.NewGraph() + UnmarshalBinary on every call (model.go:190-196)
	h, why := createEngine(m.RawBytes, device, maxBatch)
	if h == nil {
		log.Die("could not unmarshal ONNX model: %s", why)
	}
	e := &engine{h: h, nc: int(C.apr_engine_num_classes(h))}
	engines[key] = e
	return e
}

// Unload destroys every cached engine of this model (all batch sizes, all devices) and forgets its digest: the GPU
// memory goes back at once.  No counterpart in the reference, whose Infer keeps nothing between calls.  Callers must
// not be inside Infer / ScoreBatch / Tick on that model.
func Unload(m OnnxModel) int {
	if len(m.RawBytes) == 0 {
		return 0
	}
	cacheMu.Lock()
	defer cacheMu.Unlock()
	d := digestOf(m.RawBytes)
	n := 0
	for k, e := range engines {
		if k.digest == d {
			e.mu.Lock()
			C.apr_engine_destroy(e.h)
			e.h = nil
			e.mu.Unlock()
			delete(engines, k)
			n++
		}
	}
	addr := uintptr(unsafe.Pointer(&m.RawBytes[0]))
	delete(digests, addr)
	for i, a := range digestAge {
		if a == addr {
			digestAge = append(digestAge[:i], digestAge[i+1:]...)
			break
		}
	}
	return n
}

// Prepare raw audio bytes and convert them to a ready-to-infer tensor (model.go:86-184).
func Prepare(pcmData []byte) (*tensor.Dense, error) {
	e := frontend()
	flatData := make([]float32, Nmels*SpectrogramFrames)
	var p *C.uint8_t
	if len(pcmData) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&pcmData[0]))
	}
	e.mu.Lock()
	rc := C.apr_prepare(e.h, C.APR_FLAVOUR_GO, p, C.size_t(len(pcmData)), (*C.float)(unsafe.Pointer(&flatData[0])))
	e.mu.Unlock()
	if rc < 0 {
		log.Die("ML Prepare failed: %s", lastErr(e.h))
	}
	if rc&C.APR_WARN_UNDERFLOW != 0 {
		log.Warn("Audio Underflow; Segment was not large enough!")
	}
	if rc&C.APR_WARN_OVERFLOW != 0 {
		log.Warn("Audio Overflow; Segment was too large!")
	}
	shape := []int{1, 1, Nmels, SpectrogramFrames}
	inputTensor := tensor.New(tensor.Of(tensor.Float32), tensor.WithShape(shape...), tensor.WithBacking(flatData))
	return inputTensor, nil
}

// Infer runs the model and returns a MAP of probabilities (model.go:188-229).
func Infer(inferModel OnnxModel, preparedAudio *tensor.Dense) map[string]float64 {
	e := engineFor(inferModel)
	spec := preparedAudio.Data().([]float32)
	floatSlice := make([]float32, e.nc)
	e.mu.Lock()
	rc := C.apr_infer(e.h, (*C.float)(unsafe.Pointer(&spec[0])), 1, (*C.float)(unsafe.Pointer(&floatSlice[0])))
	e.mu.Unlock()
	if rc != 0 {
		log.Die("Inference failed: %v", lastErr(e.h))
	}
	logits := make([]float64, len(floatSlice))
	for i, v := range floatSlice {
		logits[i] = float64(v)
	}
	probs := softmax(logits)
	results := make(map[string]float64)
	for i, label := range inferModel.Labels {
		if i < len(probs) {
			results[label] = probs[i]
		}
	}
	return results
}

// ScoreBatch scores every 2-s window (1-s hop) of one decoded recording in one GPU pass:
// probs[k][c], label[k] (class index, -1 = none) and match[k] per daemon.go:203.
func ScoreBatch(inferModel OnnxModel, pcm []byte, trust float64) (probs [][]float64, label []int32, match []bool) {
	e := batchEngineFor(inferModel)
	n := 0
	if len(pcm) > ffmpeg.BytesPerSecond {
		n = (len(pcm) - ffmpeg.BytesPerSecond + ffmpeg.BytesPerSecond - 1) / ffmpeg.BytesPerSecond
	}
	if n == 0 {
		return
	}
	empty := -1
	for i, l := range inferModel.Labels {
		if l == "empty" {
			empty = i
		}
	}
	flat := make([]float64, n*e.nc)
	label = make([]int32, n)
	m := make([]uint8, n)
	e.mu.Lock()
	rc := C.apr_score(e.h, (*C.uint8_t)(unsafe.Pointer(&pcm[0])), C.size_t(len(pcm)), C.size_t(n),
		C.size_t(ffmpeg.BytesPerSecond), C.double(trust), C.int(empty),
		(*C.double)(unsafe.Pointer(&flat[0])), (*C.int32_t)(unsafe.Pointer(&label[0])), (*C.uint8_t)(unsafe.Pointer(&m[0])))
	e.mu.Unlock()
	if rc != 0 {
		log.Die("Inference failed: %v", lastErr(e.h))
	}
	probs = make([][]float64, n)
	match = make([]bool, n)
	for k := 0; k < n; k++ {
		probs[k] = flat[k*e.nc : (k+1)*e.nc]
		match[k] = m[k] != 0
	}
	return
}

// StreamScorer is the live tick of daemon.go:104-125 for many streams at once (apr_stream_tick): the previous
// 1-s segment of every stream stays on the GPU, a tick uploads the new ones and scores [previous || new] of all
// streams in one pass.  Buffer() is the engine's pinned staging ([nStreams*BytesPerSecond]byte): a reader that
// deposits the segments there and calls Tick(nil, ...) skips the host copy.
type StreamScorer struct {
	e *engine
	n int
	empty int
	buf []byte
}

func NewStreamScorer(inferModel OnnxModel, nStreams int) *StreamScorer {
	e := cachedEngine(inferModel, nStreams, currentDevice())
	empty := -1
	for i, l := range inferModel.Labels {
		if l == "empty" {
			empty = i
		}
	}
	e.mu.Lock()
	p := C.apr_stream_buffer(e.h, C.int(nStreams))
	if p != nil {
		C.apr_stream_reset(e.h)
	}
	e.mu.Unlock()
	if p == nil {
		log.Die("Inference failed: %v", lastErr(e.h))
	}
	return &StreamScorer{e: e, n: nStreams, empty: empty, buf: unsafe.Slice((*byte)(unsafe.Pointer(p)), nStreams*ffmpeg.BytesPerSecond)}
}

func (s *StreamScorer) Buffer() []byte { return s.buf }

// Tick returns ok == false while the first segments only prime the buffers (daemon.go:115-119).
func (s *StreamScorer) Tick(segments []byte, trust float64) (probs [][]float64, label []int32, match []bool, ok bool) {
	if segments == nil {
		segments = s.buf
	}
	if len(segments) != s.n*ffmpeg.BytesPerSecond {
		log.Die("Inference failed: StreamScorer.Tick needs %d bytes", s.n*ffmpeg.BytesPerSecond)
	}
	flat := make([]float64, s.n*s.e.nc)
	label = make([]int32, s.n)
	m := make([]uint8, s.n)
	s.e.mu.Lock()
	rc := C.apr_stream_tick(s.e.h, (*C.uint8_t)(unsafe.Pointer(&segments[0])), C.int(s.n), C.double(trust), C.int(s.empty),
		(*C.double)(unsafe.Pointer(&flat[0])), (*C.int32_t)(unsafe.Pointer(&label[0])), (*C.uint8_t)(unsafe.Pointer(&m[0])))
	s.e.mu.Unlock()
	if rc < 0 {
		log.Die("Inference failed: %v", lastErr(s.e.h))
	}
	if rc == C.APR_STREAM_PRIMED {
		return nil, nil, nil, false
	}
	probs = make([][]float64, s.n)
	match = make([]bool, s.n)
	for k := 0; k < s.n; k++ {
		probs[k] = flat[k*s.e.nc : (k+1)*s.e.nc]
		match[k] = m[k] != 0
	}
	return probs, label, match, true
}

// SweepFiles scores a list of decoded recordings in full GPU passes (short files share a pass, long
// files span several; no window crosses a file): first[f] is the first row of file f in
// label/match/probs, first[len(files)] the total.  The host->device copies run on a copy stream
// while the previous pass computes (apr_sweep_files).
func SweepFiles(inferModel OnnxModel, files [][]byte, trust float64) (probs [][]float64, label []int32, match []bool, first []int) {
	e := batchEngineFor(inferModel)
	nf := len(files)
	first = make([]int, nf+1)
	if nf == 0 {
		return
	}
	empty := -1
	for i, l := range inferModel.Labels {
		if l == "empty" {
			empty = i
		}
	}
	// the pointer table lives in C memory; the recordings themselves are pinned for the call
	// (cgo forbids passing Go memory that holds Go pointers)
	ptrs := (*[1 << 28]*C.uint8_t)(C.malloc(C.size_t(nf) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:nf:nf]
	lens := (*[1 << 28]C.size_t)(C.malloc(C.size_t(nf) * C.size_t(unsafe.Sizeof(C.size_t(0)))))[:nf:nf]
	cfirst := (*[1 << 28]C.size_t)(C.malloc(C.size_t(nf+1) * C.size_t(unsafe.Sizeof(C.size_t(0)))))[: nf+1 : nf+1]
	defer C.free(unsafe.Pointer(&ptrs[0]))
	defer C.free(unsafe.Pointer(&lens[0]))
	defer C.free(unsafe.Pointer(&cfirst[0]))
	var pin runtime.Pinner
	defer pin.Unpin()
	total := 0
	for f, b := range files {
		lens[f] = C.size_t(len(b))
		ptrs[f] = nil
		if len(b) > 0 {
			pin.Pin(&b[0])
			ptrs[f] = (*C.uint8_t)(unsafe.Pointer(&b[0]))
		}
		total += int(C.apr_windows_in(C.size_t(len(b)), C.size_t(ffmpeg.BytesPerSecond)))
	}
	flat := make([]float64, total*e.nc+1)
	label = make([]int32, total+1)
	m := make([]uint8, total+1)
	e.mu.Lock()
	rc := C.apr_sweep_files(e.h, C.int(nf), &ptrs[0], &lens[0], C.size_t(ffmpeg.BytesPerSecond), C.double(trust), C.int(empty),
		(*C.double)(unsafe.Pointer(&flat[0])), (*C.int32_t)(unsafe.Pointer(&label[0])), (*C.uint8_t)(unsafe.Pointer(&m[0])), &cfirst[0])
	e.mu.Unlock()
	if rc != 0 {
		log.Die("Inference failed: %v", lastErr(e.h))
	}
	label = label[:total]
	probs = make([][]float64, total)
	match = make([]bool, total)
	for k := 0; k < total; k++ {
		probs[k] = flat[k*e.nc : (k+1)*e.nc]
		match[k] = m[k] != 0
	}
	for f := range first {
		first[f] = int(cfirst[f])
	}
	return
}

// SweepFilesMulti is SweepFiles over a device list from ONE process (apr_sweep_files_multi): recordings are sharded
// by file, file f -> devices[f % len(devices)] (SURVEY 8(e); the loop being sharded is ai/inspect.py:20-70,73-112 /
// daemon.go:87-141), one host thread per device inside the library, rows returned in FILE order.  The results are
// host memory, so every GPU's rows cross its own PCIe link once; no device-side gather is involved.
func SweepFilesMulti(inferModel OnnxModel, devices []int, files [][]byte, trust float64) (probs [][]float64, label []int32, match []bool, first []int) {
	nf := len(files)
	first = make([]int, nf+1)
	if nf == 0 || len(devices) == 0 {
		return
	}
	// a device listed twice would list one handle twice: refused here, before any lock is taken
	seen := map[int]bool{}
	for _, d := range devices {
		if seen[d] {
			log.Die("Inference failed: SweepFilesMulti: device %d listed twice", d)
		}
		seen[d] = true
	}
	bw := currentBatchWindows()
	es := make([]*engine, len(devices))
	hs := (*[1 << 16]*C.apr_engine)(C.malloc(C.size_t(len(devices)) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:len(devices):len(devices)]
	defer C.free(unsafe.Pointer(&hs[0]))
	for i, d := range devices {
		es[i] = cachedEngine(inferModel, bw, d)
		hs[i] = es[i].h
	}
	// ONE global lock order (ascending device id), whatever order the caller listed the devices in: two concurrent calls
	// with [0,1] and [1,0], or a call overlapping a ScoreBatch / StreamScorer on one of these engines, cannot deadlock
	order := make([]int, len(devices))
	for i := range order {
		order[i] = i
	}
	sort.Slice(order, func(a, b int) bool { return devices[order[a]] < devices[order[b]] })
	empty := -1
	for i, l := range inferModel.Labels {
		if l == "empty" {
			empty = i
		}
	}
	ptrs := (*[1 << 28]*C.uint8_t)(C.malloc(C.size_t(nf) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:nf:nf]
	lens := (*[1 << 28]C.size_t)(C.malloc(C.size_t(nf) * C.size_t(unsafe.Sizeof(C.size_t(0)))))[:nf:nf]
	cfirst := (*[1 << 28]C.size_t)(C.malloc(C.size_t(nf+1) * C.size_t(unsafe.Sizeof(C.size_t(0)))))[: nf+1 : nf+1]
	defer C.free(unsafe.Pointer(&ptrs[0]))
	defer C.free(unsafe.Pointer(&lens[0]))
	defer C.free(unsafe.Pointer(&cfirst[0]))
	var pin runtime.Pinner
	defer pin.Unpin()
	total := 0
	for f, b := range files {
		lens[f] = C.size_t(len(b))
		ptrs[f] = nil
		if len(b) > 0 {
			pin.Pin(&b[0])
			ptrs[f] = (*C.uint8_t)(unsafe.Pointer(&b[0]))
		}
		total += int(C.apr_windows_in(C.size_t(len(b)), C.size_t(ffmpeg.BytesPerSecond)))
	}
	nc := es[0].nc
	flat := make([]float64, total*nc+1)
	label = make([]int32, total+1)
	m := make([]uint8, total+1)
	for _, i := range order {
		es[i].mu.Lock()
	}
	rc := C.apr_sweep_files_multi(&hs[0], C.int(len(devices)), C.int(nf), &ptrs[0], &lens[0], C.size_t(ffmpeg.BytesPerSecond), C.double(trust),
		C.int(empty), (*C.double)(unsafe.Pointer(&flat[0])), (*C.int32_t)(unsafe.Pointer(&label[0])), (*C.uint8_t)(unsafe.Pointer(&m[0])), &cfirst[0])
	for _, i := range order {
		es[i].mu.Unlock()
	}
	if rc != 0 {
		log.Die("Inference failed: %v", lastErr(es[0].h))
	}
	label = label[:total]
	probs = make([][]float64, total)
	match = make([]bool, total)
	for k := 0; k < total; k++ {
		probs[k] = flat[k*nc : (k+1)*nc]
		match[k] = m[k] != 0
	}
	for f := range first {
		first[f] = int(cfirst[f])
	}
	return
}

// SweepFilesGatherHits is the device-resident form of SweepFilesMulti, the shape north_star describes: recordings shard by file over
// the listed GPUs (file f -> devices[f % len(devices)]), every GPU scores its shard with the rows left in its HBM
// (apr_score_device), the rows are gathered on the first listed GPU over RCCL / xGMI (apr_gather_rows_device), the one-hit-per-
// minute report (docs/usage/review.md:48-53) is formed there (apr_bucket_hits_device), and only that report crosses PCIe:
// hits[f][b][c] is true iff class c matched at least once in minute b of recording f.  One process; one engine per device.
func SweepFilesGatherHits(inferModel OnnxModel, devices []int, files [][]byte, trust float64) (hits [][][]bool) {
	nf, nd := len(files), len(devices)
	hits = make([][][]bool, nf)
	if nf == 0 || nd == 0 {
		return
	}
	seen := map[int]bool{}
	for _, d := range devices {
		if seen[d] {
			log.Die("Inference failed: SweepFilesGatherHits: device %d listed twice", d)
		}
		seen[d] = true
	}
	empty := -1
	for i, l := range inferModel.Labels {
		if l == "empty" {
			empty = i
		}
	}
	bw := currentBatchWindows()
	es := make([]*engine, nd)
	for i, d := range devices {
		es[i] = cachedEngine(inferModel, bw, d)
	}
	nc := es[0].nc
	order := make([]int, nd)
	for i := range order {
		order[i] = i
	}
	sort.Slice(order, func(a, b int) bool { return devices[order[a]] < devices[order[b]] })
	for _, i := range order {
		es[i].mu.Lock()
	}
	defer func() {
		for _, i := range order {
			es[i].mu.Unlock()
		}
	}()
	// the shard plan: windows of every file, rows of every device, each file's first row in gather (= device, then file) order
	wins := make([]int, nf)
	rows := make([]int, nd)
	for f, b := range files {
		wins[f] = int(C.apr_windows_in(C.size_t(len(b)), C.size_t(ffmpeg.BytesPerSecond)))
		rows[f%nd] += wins[f]
	}
	base := make([]int, nd+1)
	for d := 0; d < nd; d++ {
		base[d+1] = base[d] + rows[d]
	}
	total := base[nd]
	if total == 0 {
		return
	}
	chk := func(e *engine, rc C.int) {
		if rc != 0 {
			log.Die("Inference failed: %v", lastErr(e.h))
		}
	}
	alloc := func(e *engine, n int) unsafe.Pointer {
		var p unsafe.Pointer
		chk(e, C.apr_device_alloc(e.h, C.size_t(n), &p))
		return p
	}
	cEng := (*[1 << 16]*C.apr_engine)(C.malloc(C.size_t(nd) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:nd:nd]
	cRows := (*[1 << 16]C.size_t)(C.malloc(C.size_t(nd) * C.size_t(unsafe.Sizeof(C.size_t(0)))))[:nd:nd]
	cLab := (*[1 << 16]*C.int32_t)(C.malloc(C.size_t(nd) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:nd:nd]
	cMat := (*[1 << 16]*C.uint8_t)(C.malloc(C.size_t(nd) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:nd:nd]
	defer C.free(unsafe.Pointer(&cEng[0]))
	defer C.free(unsafe.Pointer(&cRows[0]))
	defer C.free(unsafe.Pointer(&cLab[0]))
	defer C.free(unsafe.Pointer(&cMat[0]))
	firstRow := make([]int, nf) // row of file f's first window in gather order
	var owned []struct {
		e *engine
		p unsafe.Pointer
	}
	defer func() {
		for _, o := range owned {
			C.apr_device_free(o.e.h, o.p)
		}
	}()
	keep := func(e *engine, p unsafe.Pointer) unsafe.Pointer {
		owned = append(owned, struct {
			e *engine
			p unsafe.Pointer
		}{e, p})
		return p
	}
	for d := 0; d < nd; d++ {
		e := es[d]
		cEng[d], cRows[d] = e.h, C.size_t(rows[d])
		dLab := keep(e, alloc(e, 4*rows[d]+4))
		dMat := keep(e, alloc(e, rows[d]+4))
		cLab[d], cMat[d] = (*C.int32_t)(dLab), (*C.uint8_t)(dMat)
		done := 0
		for f := d; f < nf; f += nd { // this device's recordings, in file order: no window crosses a recording
			firstRow[f] = base[d] + done
			if wins[f] == 0 {
				continue
			}
			dPcm := alloc(e, len(files[f]))
			chk(e, C.apr_device_upload(e.h, dPcm, unsafe.Pointer(&files[f][0]), C.size_t(len(files[f]))))
			rc := C.apr_score_device(e.h, (*C.uint8_t)(dPcm), C.size_t(len(files[f])), C.size_t(wins[f]), C.size_t(ffmpeg.BytesPerSecond),
				C.double(trust), C.int(empty), nil, (*C.int32_t)(C.apr_at(dLab, C.size_t(4*done))), (*C.uint8_t)(C.apr_at(dMat, C.size_t(done))), nil)
			if rc == 0 {
				rc = C.apr_sync(e.h) // the PCM buffer is freed next
			}
			C.apr_device_free(e.h, dPcm)
			chk(e, rc)
			done += wins[f]
		}
	}
	root := es[0]
	gLab := keep(root, alloc(root, 4*total))
	gMat := keep(root, alloc(root, total))
	if rc := C.apr_gather_rows_device(&cEng[0], C.int(nd), 0, &cRows[0], C.int(nc), nil, &cLab[0], &cMat[0], nil, (*C.int32_t)(gLab), (*C.uint8_t)(gMat)); rc != 0 {
		log.Die("Inference failed: %v", lastErr(root.h))
	}
	// one report per recording, formed on the root GPU from the gathered rows
	const wpb = 60
	nbTotal := 0
	nbOf := make([]int, nf)
	for f := range files {
		nbOf[f] = (wins[f] + wpb - 1) / wpb
		nbTotal += nbOf[f]
	}
	dHits := keep(root, alloc(root, nbTotal*nc+4))
	off := 0
	for f := range files {
		if wins[f] > 0 {
			chk(root, C.apr_bucket_hits_device(root.h, (*C.int32_t)(C.apr_at(gLab, C.size_t(4*firstRow[f]))), (*C.uint8_t)(C.apr_at(gMat, C.size_t(firstRow[f]))),
				C.size_t(wins[f]), C.int(nc), wpb, (*C.uint8_t)(C.apr_at(dHits, C.size_t(off*nc)))))
		}
		off += nbOf[f]
	}
	flat := make([]uint8, nbTotal*nc+1)
	chk(root, C.apr_sync(root.h))
	chk(root, C.apr_device_download(root.h, unsafe.Pointer(&flat[0]), dHits, C.size_t(nbTotal*nc)))
	off = 0
	for f := range files {
		hits[f] = make([][]bool, nbOf[f])
		for b := range hits[f] {
			hits[f][b] = make([]bool, nc)
			for c := 0; c < nc; c++ {
				hits[f][b][c] = flat[(off+b)*nc+c] != 0
			}
		}
		off += nbOf[f]
	}
	return
}

// WavDataOffset reports where the PCM payload of ffmpeg's `-f wav -` stream starts (0 for raw PCM).
// The reference reads that stream from byte 0 (ffmpeg.go:111-114, daemon.go:151-154); skipping this
// many bytes first keeps the RIFF header out of the first window.
func WavDataOffset(head []byte) int {
	if len(head) == 0 {
		return 0
	}
	return int(C.apr_wav_data_offset((*C.uint8_t)(unsafe.Pointer(&head[0])), C.size_t(len(head)), nil, nil, nil))
}

// Upsample3 turns 16 kHz s16 PCM into the 48 kHz the scoring path expects, on the GPU (apr_upsample with the
// library's default 3 x 32-tap Kaiser low-pass).  The reference asks ffmpeg for `-ar 48000` instead
// (src/ffmpeg/ffmpeg.go:76,113-114); this is for sources that reach the daemon without ffmpeg.
func Upsample3(pcm16k []int16) []int16 {
	if len(pcm16k) == 0 {
		return nil
	}
	var taps [3 * 32]C.double
	if rc := C.apr_resample_taps(3, 32, 0.97, 9.0, &taps[0]); rc != 0 {
		log.Die("apr_resample_taps failed: %d", int(rc))
	}
	out := make([]int16, 3*len(pcm16k))
	e := frontend()
	e.mu.Lock()
	rc := C.apr_upsample(e.h, (*C.int16_t)(unsafe.Pointer(&pcm16k[0])), C.size_t(len(pcm16k)), 3, &taps[0], 32,
		(*C.int16_t)(unsafe.Pointer(&out[0])))
	e.mu.Unlock()
	if rc < 0 {
		log.Die("Upsample failed: %s", lastErr(e.h))
	}
	return out
}

// BucketHits folds per-window matches into one hit per class per minute (docs/usage/review.md:48-53).
func BucketHits(label []int32, match []bool, nClasses int) [][]bool {
	n := len(label)
	nb := (n + 59) / 60
	out := make([][]bool, nb)
	for b := range out {
		out[b] = make([]bool, nClasses)
	}
	if n == 0 {
		return out
	}
	m := make([]uint8, n)
	for k, v := range match {
		if v {
			m[k] = 1
		}
	}
	hits := make([]uint8, nb*nClasses)
	C.apr_bucket_hits((*C.int32_t)(unsafe.Pointer(&label[0])), (*C.uint8_t)(unsafe.Pointer(&m[0])), C.size_t(n), C.int(nClasses), 60,
		(*C.uint8_t)(unsafe.Pointer(&hits[0])))
	for b := 0; b < nb; b++ {
		for c := 0; c < nClasses; c++ {
			out[b][c] = hits[b*nClasses+c] != 0
		}
	}
	return out
}
