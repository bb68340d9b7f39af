/*
Package gpubinpacking puts libcasim (hand-written HIP kernels for MI355X / gfx950) behind the Cluster Autoscaler's
estimator.Estimator interface (cluster-autoscaler/estimator/estimator.go:53-56).

NOT COMPILED IN THE REPOSITORY THAT CARRIES IT: that tree has no Go toolchain.  Every C function used here is declared in
include/casim.h and exported by libcasim.so (tests/test_abi.py); the call sequence of estimator.go / prefetch.go is
replayed call for call in plain C++ by tools/casim_native --shim (tests/test_native_harness.py) and mirrored in Python
(kubernetes_autoscaler_amd/estimator.py: PrefetchShared, PrefetchNodeGroupListProcessor, BinpackingNodeEstimator), both of
which run against the oracle on the MI355X.

Files: engine.go (context, errors), encode.go (pods / templates -> casim_enc_* calls), estimator.go (Estimate: prefetch
lookup, per-call path, fallback), prefetch.go (NodeGroupListProcessor wrapper, the shared cache), builder.go (the
EstimatorBuilder and the limiter wrapper), autoscaler_go.patch (core/autoscaler.go picks the builder by --estimator).
*/
package gpubinpacking

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../kubernetes_autoscaler_amd -lcasim
#include <stdlib.h>
#include "casim.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"sync"
	"unsafe"
)

// Engine owns one casim_ctx (one MI355X, one HIP stream; libcasim cuts batches of simulations into sub-batches on
// internal streams itself, casim_options.n_streams).  A nil *Engine means "no device": callers keep the Go estimator.
type Engine struct {
	mu  sync.Mutex // calls on one casim_ctx are ordered (INTEGRATION.md section 4)
	ctx *C.casim_ctx
}

// NewEngine returns an error when the machine has no gfx950 device or libcasim's ABI is not the one this file was
// written against; the caller then builds the reference estimator (fail closed, never a CPU path inside libcasim).
func NewEngine(device int) (*Engine, error) {
	if v := int(C.casim_abi_version()); v != C.CASIM_ABI_VERSION {
		return nil, fmt.Errorf("libcasim ABI %d, shim built for %d", v, int(C.CASIM_ABI_VERSION))
	}
	ctx := C.casim_ctx_create(C.int32_t(device), nil)
	if ctx == nil {
		return nil, errors.New(C.GoString(C.casim_last_error()))
	}
	return &Engine{ctx: ctx}, nil
}

// Close releases the context (streams, memory pools).
func (e *Engine) Close() {
	if e != nil && e.ctx != nil {
		C.casim_ctx_destroy(e.ctx)
		e.ctx = nil
	}
}

// PackBuild reports which build of the register packer libcasim's self-check left standing (casim_pack_build_info).
func (e *Engine) PackBuild(device int) (plain bool, compared, differing int) {
	var out [4]C.int32_t
	C.casim_pack_build_info(C.int32_t(device), &out[0])
	return out[0] == C.CASIM_PACK_BUILD_PLAIN, int(out[1]), int(out[2])
}

// ---- small cgo helpers ------------------------------------------------------------------------------------------

// cstrings keeps the C copies of the strings of one encoder session and frees them together.
type cstrings struct{ p []unsafe.Pointer }

func (c *cstrings) s(v string) *C.char {
	p := C.CString(v)
	c.p = append(c.p, unsafe.Pointer(p))
	return p
}

// arr builds a NULL-terminated-free C array of C strings (length passed separately, as casim.h wants it).
func (c *cstrings) arr(vs []string) **C.char {
	if len(vs) == 0 {
		return nil
	}
	a := (**C.char)(C.malloc(C.size_t(len(vs)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	c.p = append(c.p, unsafe.Pointer(a))
	s := unsafe.Slice(a, len(vs))
	for i, v := range vs {
		s[i] = c.s(v)
	}
	return a
}

func (c *cstrings) free() {
	for _, p := range c.p {
		C.free(p)
	}
	c.p = nil
}

func rcErr(rc C.int32_t, what string) error {
	if rc >= 0 {
		return nil
	}
	return fmt.Errorf("%s: libcasim error %d: %s", what, int(rc), C.GoString(C.casim_last_error()))
}
