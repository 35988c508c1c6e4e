// apr_cgo.go -- the cgo binding that moves dtrack/model's DSP and CNN onto an MI355X through
// libapr_amd.so (include/apr.h).
//
// How it is applied to the reference tree (see INTEGRATION.md):
//   * src/model/dsp.go is deleted (the mel lens lives on the GPU),
//   * in src/model/model.go the bodies of Prepare (model.go:86-184) and Infer
//     (model.go:188-229) and the go-dsp / onnx-go imports are removed; the constants,
//     OnnxModel, Train, Load, softmax and normalizeAudio stay exactly as they are,
//   * this file is added next to it and supplies Prepare, Infer and the new ScoreBatch.
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
*/
import "C"

import (
	"dtrack/ffmpeg"
	"dtrack/log"

	"sync"
	"unsafe"

	"gorgonia.org/tensor"
)

// ---- engine cache -------------------------------------------------------------------------
// One front-end engine for Prepare (single caller goroutine, daemon.go:123) and one engine
// per model, keyed by the identity of RawBytes, each guarded by its own mutex (a handle is
// safe for one caller at a time; the daemon runs one scanner goroutine per model).
type engine struct {
	mu sync.Mutex
	h  *C.apr_engine
	nc int
}

var (
	cacheMu  sync.Mutex
	frontEnd *engine
	engines  = map[uintptr]*engine{}
)

func lastErr(h *C.apr_engine) string { return C.GoString(C.apr_last_error(h)) }

func frontend() *engine {
	cacheMu.Lock()
	defer cacheMu.Unlock()
	if frontEnd == nil {
		var h *C.apr_engine
		if rc := C.apr_engine_create(nil, 0, 0, 64, &h); rc != 0 {
			log.Die("GPU front end unavailable: %s", lastErr(nil))
		}
		frontEnd = &engine{h: h}
	}
	return frontEnd
}

func engineFor(m OnnxModel) *engine {
	if len(m.RawBytes) == 0 {
		log.Die("could not unmarshal ONNX model: empty")
	}
	key := uintptr(unsafe.Pointer(&m.RawBytes[0]))
	cacheMu.Lock()
	defer cacheMu.Unlock()
	if e, ok := engines[key]; ok {
		return e
	}
	var h *C.apr_engine
	// replaces gorgonnx.NewGraph() + UnmarshalBinary on every call (model.go:190-196)
	rc := C.apr_engine_create(unsafe.Pointer(&m.RawBytes[0]), C.size_t(len(m.RawBytes)), 0, 0, &h)
	if rc != 0 {
		log.Die("could not unmarshal ONNX model: %s", lastErr(nil))
	}
	e := &engine{h: h, nc: int(C.apr_engine_num_classes(h))}
	engines[key] = e
	return e
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
	e := engineFor(inferModel)
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
