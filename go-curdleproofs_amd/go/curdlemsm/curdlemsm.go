// Package curdlemsm binds libcurdlemsm.so (include/curdle_msm.h) for the Go
// reference: it is what `common.MultiExp` forwards to once the reference's 39
// `X.MultiExp(points, scalars, common.MultiExpConf)` call sites are routed
// through one function (INTEGRATION.md).
//
// UNVERIFIED: there is no Go toolchain in the build image, so this file has
// never been compiled (needs Go >= 1.21 for runtime.Pinner).  It is deliberately small: one cgo call per entry point,
// zero-copy (gnark's fp.Element / fr.Element / G1Affine / G1Jac are
// pointer-free [k]uint64 arrays, so &s[0] can be handed to C directly and the
// cgo pointer rules hold: the library keeps no pointer after returning).
package curdlemsm

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../.. -lcurdlemsm -Wl,-rpath,${SRCDIR}/../..
#include <stdlib.h>
#include <stdint.h>
#include "curdle_msm.h"

// Device addresses travel from Go as integers (uintptr_t), never as unsafe.Pointer: they are not Go
// pointers, and `unsafe.Pointer(uintptr)` is what `go vet` (unsafeptr) rejects.  The arrays themselves are
// Go slices of integers -- pointer-free memory, which cgo may pass for the duration of a call.
static int curdle_go_replicated(const uintptr_t* d_points, const uintptr_t* d_scalars, size_t n, int split, uint64_t* out) {
	return curdle_msm_g1_replicated((const void* const*)d_points, (const void* const*)d_scalars, n, split, out);
}
static int curdle_go_device_ex(uintptr_t d_points, uintptr_t d_scalars, size_t n, unsigned flags, uint64_t* out) {
	return curdle_msm_g1_device_ex((const void*)d_points, (const void*)d_scalars, n, flags, out, NULL);
}
static int curdle_go_forget_bases(uintptr_t d_points) { return curdle_msm_forget_bases((const void*)d_points); }
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc"
	bls12381 "github.com/consensys/gnark-crypto/ecc/bls12-381"
	"github.com/consensys/gnark-crypto/ecc/bls12-381/fr"
)

// The library keeps the text of the last error per OS THREAD; a goroutine can migrate between
// two cgo calls, so every entry point below runs `call` and the error fetch under
// runtime.LockOSThread (ADVICE r1).
func locked(call func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := call()
	if rc == 0 {
		return nil
	}
	var buf [256]C.char
	C.curdle_last_error(&buf[0], C.size_t(len(buf)))
	return fmt.Errorf("curdlemsm: rc=%d: %s", int(rc), C.GoString(&buf[0]))
}

// Init selects the HIP device of this process (one process per GPU).
func Init(device int) error {
	return locked(func() C.int { return C.curdle_init(C.int(device)) })
}

// InitDevices configures one context per entry of devices (HIP device ids): ONE Go process
// driving several GPUs, which is how the reference's single-process callers reach BASELINE
// configs 4 and 5 (curdle_init_devices in curdle_msm.h).  After it MultiExp splits large
// inputs by point ranges over all devices (each GPU copies and runs its own share; one host
// thread per device inside the library; the partial sums are added on the host), and
// VerifyBatch-style entry points shard their proofs over the devices.  A goroutine that
// wants a particular device for its own calls uses OnDevice.
func InitDevices(devices []int) error {
	if len(devices) == 0 {
		return errors.New("curdlemsm: InitDevices: no devices")
	}
	ids := make([]C.int, len(devices))
	for i, d := range devices {
		ids[i] = C.int(d)
	}
	return locked(func() C.int { return C.curdle_init_devices(&ids[0], C.int(len(ids))) })
}

// DeviceCount is the number of configured contexts (1 unless InitDevices said more).
func DeviceCount() int { return int(C.curdle_device_count()) }

// OnDevice runs f with the calling OS thread's current context set to `ordinal` (the
// library's selection is per OS thread, like hipSetDevice, so the goroutine is locked to its
// thread for the duration and the previous selection is restored -- "no selection" (-1) included: Go reuses OS
// threads, and a thread left selected on device 0 would keep every later MultiExp that lands on it from
// spreading over the devices; curdle_get_device() cannot tell "selected 0" from "selected nothing").
func OnDevice(ordinal int, f func() error) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	prev := C.curdle_get_device_selection()
	if rc := C.curdle_set_device(C.int(ordinal)); rc != 0 {
		return fmt.Errorf("curdlemsm: no device %d (rc=%d)", ordinal, int(rc))
	}
	defer C.curdle_set_device(prev)
	return f()
}

// MinGPUPairs is the size below which MultiExp stays on gnark's CPU path.  `bench.py --sweep`
// (profiles/r03_sweep.json) puts one synchronous GPU call from host slices at 0.23-0.30 ms for
// 8..512 pairs -- a chain of dependent point additions, whatever the size -- against 0.35 ms
// (8 pairs) .. 1.2 ms (512) for the C port of the bucket method on one core; gnark's own
// MultiExp is tighter than that port for a handful of points (a 255-bit scalar multiplication is
// ~70 us on a core), so the verifier's ten size-m calls per Verify (m = 6..9,
// innerproductargument.go:238-280, samemultiscalarargument.go:196-227) are kept on the CPU and
// everything from 32 pairs on goes to the GPU.  A variable, so a maintainer with a Go box can
// set it from go/bench/multiexp_bench_test.go's numbers.
var MinGPUPairs = 32

// MultiExp is the drop-in for (*bls12381.G1Jac).MultiExp(points, scalars, cfg):
// dst = sum_i scalars[i] * points[i].  Same contract as gnark: the receiver is
// overwritten, a length mismatch is an error, an empty input gives infinity.
//
// PRECONDITION (differs from gnark): every point must be in the prime-order subgroup G1 or
// be the point at infinity -- the kernels split every scalar with the curve endomorphism,
// which is multiplication by lambda only there (curdle_msm.h).  Every point the reference
// passes is: CRS points are multiples of the generator, proof and tracker points come out of
// gnark's Decoder / SetBytes, which check the subgroup.
//
// Inputs below MinGPUPairs stay on gnark's CPU path (cfg: the reference's common.MultiExpConf).
//
// If the GPU call fails (no device, out of memory, a HIP error) and FallbackToCPU is set, the
// result comes from gnark's MultiExp instead and the GPU's error is handed to OnFallback:
// SURVEY.md section 5's failure row wants a verifier's accept bits to be independent of the
// GPU's health.  Off by default: a deployment that bought a GPU wants to hear that it is not used.
func MultiExp(dst *bls12381.G1Jac, points []bls12381.G1Affine, scalars []fr.Element, cfg ecc.MultiExpConfig) (*bls12381.G1Jac, error) {
	return MultiExpFlags(dst, points, scalars, cfg, DefaultFlags)
}

// Flags of MultiExpFlags (CURDLE_MSM_* in curdle_msm.h).
const (
	// AnyCurvePoint: no endomorphism -- gnark's own contract: the result is k*P for every point of the
	// curve, in the prime-order subgroup or not (twice the windows; about 1.3x the time of a large MSM).
	AnyCurvePoint = 1
	// BasesUnchanged (device inputs only): the library keeps its converted copy of the base array.
	BasesUnchanged = 2
)

// DefaultFlags is what MultiExp passes.  A deployment whose callers hand curdleproof.Verify bases that
// did NOT come out of gnark's subgroup-checking decoders sets it to AnyCurvePoint and gets gnark's
// result for every input gnark accepts.
var DefaultFlags uint = 0

// MultiExpFlags is MultiExp with the options of curdle_msm_g1_ex.
func MultiExpFlags(dst *bls12381.G1Jac, points []bls12381.G1Affine, scalars []fr.Element, cfg ecc.MultiExpConfig, flags uint) (*bls12381.G1Jac, error) {
	if len(points) != len(scalars) {
		return nil, errors.New("len(points) != len(scalars)")
	}
	if len(points) < MinGPUPairs {
		return dst.MultiExp(points, scalars, cfg)
	}
	var pp, sp unsafe.Pointer
	if len(points) > 0 {
		pp = unsafe.Pointer(&points[0])
		sp = unsafe.Pointer(&scalars[0])
	}
	err := locked(func() C.int {
		return C.curdle_msm_g1_ex((*C.uint64_t)(pp), (*C.uint64_t)(sp), C.size_t(len(points)), C.uint(flags),
			(*C.uint64_t)(unsafe.Pointer(dst)))
	})
	if err != nil {
		if FallbackToCPU {
			if OnFallback != nil {
				OnFallback(err)
			}
			return dst.MultiExp(points, scalars, cfg)
		}
		return nil, err
	}
	return dst, nil
}

// FallbackToCPU: a failed GPU call is answered by gnark's CPU MultiExp (same result: both compute
// the group element) instead of an error.  OnFallback, if set, hears the GPU's error each time
// (log it, count it, page someone).
var (
	FallbackToCPU = false
	OnFallback    func(gpuErr error)
)

// Split of MultiExpReplicated (CURDLE_SPLIT_* in curdle_msm.h).
const (
	SplitAuto    = 0 // the library's rule: windows up to 2^21 pairs, point ranges beyond
	SplitWindows = 1 // north_star's partition: device d runs Pippenger windows [w_d, w_d+1) over all pairs
	SplitPoints  = 2 // device d runs all windows over pairs [d n/D, (d+1) n/D)
)

// MultiExpReplicated is INTEGRATION.md section 3.1's call: ONE MSM over inputs the caller keeps
// RESIDENT on every configured device (InitDevices), split by Pippenger windows or by point
// ranges, the D partial sums added on the host (curdle_msm_g1_replicated; BASELINE config 4 from
// one Go process).  dPoints[d] / dScalars[d] are DEVICE pointers on device d (n x 96 B gnark
// G1Affine, n x 32 B fr.Element), e.g. from hipMalloc through another binding; they are passed
// through as integers, never dereferenced by Go.
func MultiExpReplicated(dst *bls12381.G1Jac, dPoints, dScalars []uintptr, n int, split int) error {
	d := DeviceCount()
	if len(dPoints) != d || len(dScalars) != d {
		return fmt.Errorf("curdlemsm: MultiExpReplicated: %d devices configured, %d / %d pointers given", d, len(dPoints), len(dScalars))
	}
	if d == 0 {
		return errors.New("curdlemsm: MultiExpReplicated: no device configured")
	}
	// []uintptr is pointer-free Go memory: cgo may pass it for the duration of the call, whatever d is
	return locked(func() C.int {
		return C.curdle_go_replicated((*C.uintptr_t)(unsafe.Pointer(&dPoints[0])), (*C.uintptr_t)(unsafe.Pointer(&dScalars[0])),
			C.size_t(n), C.int(split), (*C.uint64_t)(unsafe.Pointer(dst)))
	})
}

// MultiExpDevice is one MSM over inputs resident on the calling thread's device (device addresses as
// integers), with the flags above: BasesUnchanged makes the library keep its converted copy of dPoints
// (ForgetBases drops it before the memory is freed or rewritten).
func MultiExpDevice(dst *bls12381.G1Jac, dPoints, dScalars uintptr, n int, flags uint) error {
	return locked(func() C.int {
		return C.curdle_go_device_ex(C.uintptr_t(dPoints), C.uintptr_t(dScalars), C.size_t(n), C.uint(flags),
			(*C.uint64_t)(unsafe.Pointer(dst)))
	})
}

// NumWindows is the number of Pippenger windows of the plan a call with these (n, windowBits, flags) runs:
// ceil(127 / c), or ceil(255 / c) with AnyCurvePoint (the whole scalar is recoded) -- the range a caller of the
// window-range entry points (curdle_msm_g1_device_windows_ex, one rank per GPU) partitions.  windowBits = 0 is the
// library's choice for n.
func NumWindows(n int, windowBits int, flags uint) (int, error) {
	w := int(C.curdle_msm_num_windows_ex(C.size_t(n), C.int(windowBits), C.uint(flags)))
	if w < 0 {
		return 0, fmt.Errorf("curdlemsm: window_bits %d outside [4, 16]", windowBits)
	}
	return w, nil
}

// ForgetBases: see MultiExpDevice.
func ForgetBases(dPoints uintptr) error {
	return locked(func() C.int { return C.curdle_go_forget_bases(C.uintptr_t(dPoints)) })
}

// ResidentBases is a base set converted once and kept on the GPU (curdle_dbases): the CRS of a
// verifier (crs.go:10-18) -- msmaccumulator.Verify's bases are mostly those.  MultiExp over it
// uploads 32 bytes per pair (the scalars) instead of 128 and converts nothing.
type ResidentBases struct{ h *C.curdle_dbases }

// NewResidentBases copies and converts the points now (on the calling thread's device; every other
// configured device makes its copy the first time it is used there).
func NewResidentBases(points []bls12381.G1Affine) (*ResidentBases, error) {
	r := &ResidentBases{}
	var pp unsafe.Pointer
	if len(points) > 0 {
		pp = unsafe.Pointer(&points[0])
	}
	if err := locked(func() C.int { return C.curdle_dbases_create((*C.uint64_t)(pp), C.size_t(len(points)), &r.h) }); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(r, func(r *ResidentBases) { r.Free() })
	return r, nil
}

// Free releases the device copies (deferred by the library while an MSM still reads them).
func (r *ResidentBases) Free() {
	if r.h != nil {
		C.curdle_dbases_free(r.h)
		r.h = nil
	}
}

// MultiExp: dst = sum_i scalars[i] * bases[i] over the first len(scalars) bases of the set.  Same
// fallback rule as MultiExp (the caller passes the bases' host copy for that case, or nil).
func (r *ResidentBases) MultiExp(dst *bls12381.G1Jac, scalars []fr.Element, hostBases []bls12381.G1Affine, cfg ecc.MultiExpConfig) (*bls12381.G1Jac, error) {
	if r.h == nil {
		return nil, errors.New("curdlemsm: resident bases were freed")
	}
	var sp unsafe.Pointer
	if len(scalars) > 0 {
		sp = unsafe.Pointer(&scalars[0])
	}
	err := locked(func() C.int {
		return C.curdle_msm_g1_dbases_host(r.h, (*C.uint64_t)(sp), C.size_t(len(scalars)), (*C.uint64_t)(unsafe.Pointer(dst)))
	})
	if err != nil {
		if FallbackToCPU && len(hostBases) >= len(scalars) {
			if OnFallback != nil {
				OnFallback(err)
			}
			return dst.MultiExp(hostBases[:len(scalars)], scalars, cfg)
		}
		return nil, err
	}
	return dst, nil
}

// MultiExpShared computes len(sets) MSMs that share one scalar vector
// (samemultiscalarargument.go:64-70, :206/:218/:231; curdleproof.go:110,:114).
func MultiExpShared(dst []bls12381.G1Jac, sets [][]bls12381.G1Affine, scalars []fr.Element) error {
	if len(dst) != len(sets) {
		return errors.New("len(dst) != len(sets)")
	}
	if len(sets) == 0 {
		return nil
	}
	for _, s := range sets {
		if len(s) != len(scalars) {
			return errors.New("len(points) != len(scalars)")
		}
	}
	if len(scalars) == 0 { // k empty MSMs: infinity each, nothing to hand to C
		for i := range dst {
			dst[i] = bls12381.G1Jac{}
			dst[i].X.SetOne()
			dst[i].Y.SetOne()
		}
		return nil
	}
	// The C side takes an array of base-set pointers.  It lives in C memory, zeroed (calloc), and
	// the Go slices it points to are pinned for the duration of the call (runtime.Pinner,
	// Go >= 1.21): storing unpinned Go pointers in C memory is what the cgo rules forbid.
	mem := C.calloc(C.size_t(len(sets)), C.size_t(unsafe.Sizeof(uintptr(0))))
	if mem == nil {
		return errors.New("curdlemsm: out of memory")
	}
	defer C.free(mem)
	arr := unsafe.Slice((**C.uint64_t)(mem), len(sets)) // no fixed-size array type: any number of sets
	var pin runtime.Pinner
	defer pin.Unpin()
	for i, s := range sets {
		pin.Pin(&s[0])
		arr[i] = (*C.uint64_t)(unsafe.Pointer(&s[0]))
	}
	return locked(func() C.int {
		return C.curdle_msm_g1_multi((**C.uint64_t)(mem), C.size_t(len(sets)),
			(*C.uint64_t)(unsafe.Pointer(&scalars[0])), C.size_t(len(scalars)), (*C.uint64_t)(unsafe.Pointer(&dst[0])))
	})
}

// MultiExpBatch runs independent MSMs in one call (many concurrent
// msmaccumulator.Verify calls, BASELINE config 5).  offsets has len(dst)+1 entries.
func MultiExpBatch(dst []bls12381.G1Jac, points []bls12381.G1Affine, scalars []fr.Element, offsets []uint64) error {
	if len(offsets) != len(dst)+1 || len(points) != len(scalars) {
		return errors.New("bad batch shape")
	}
	if len(dst) == 0 {
		return nil
	}
	var pp, sp unsafe.Pointer
	if len(points) > 0 {
		pp = unsafe.Pointer(&points[0])
		sp = unsafe.Pointer(&scalars[0])
	}
	return locked(func() C.int {
		return C.curdle_msm_g1_batch((*C.uint64_t)(pp), (*C.uint64_t)(sp),
			(*C.size_t)(unsafe.Pointer(&offsets[0])), C.size_t(len(dst)), (*C.uint64_t)(unsafe.Pointer(&dst[0])))
	})
}

// ScalarMulBatch computes out[i] = addends[i] + scalars[i]*points[i] on the GPU
// (curdle_g1_scalar_mul_batch): the prover's fold steps
// (innerproductargument.go:155-166, samemultiscalarargument.go:129-135) with one
// shared scalar (len(scalars) == 1) and its loops of plain scalar multiplications
// (grandproductargument.go:94-103, common/util.go:55-63) with addends == nil.
func ScalarMulBatch(out, points []bls12381.G1Affine, scalars []fr.Element, addends []bls12381.G1Affine) error {
	n := len(points)
	if len(out) != n || (len(scalars) != n && len(scalars) != 1) || (addends != nil && len(addends) != n) {
		return errors.New("curdlemsm: ScalarMulBatch: mismatched lengths")
	}
	if n == 0 {
		return nil
	}
	var ap unsafe.Pointer
	if addends != nil {
		ap = unsafe.Pointer(&addends[0])
	}
	return locked(func() C.int {
		return C.curdle_g1_scalar_mul_batch((*C.uint64_t)(unsafe.Pointer(&points[0])),
			(*C.uint64_t)(unsafe.Pointer(&scalars[0])), C.size_t(len(scalars)), (*C.uint64_t)(ap), C.size_t(n),
			(*C.uint64_t)(unsafe.Pointer(&out[0])))
	})
}

// Status of one record decoded by DecompressBatch (CURDLE_DECODE_* in curdle_msm.h).
const (
	DecodeOK            = 0
	DecodeInfinity      = 1
	DecodeBadEncoding   = 2
	DecodeNotOnCurve    = 3
	DecodeNotInSubgroup = 4
)

// DecompressBatch decodes len(in)/48 compressed G1 points on the GPU
// (curdle_g1_decompress_batch) -- what the per-point G1Affine.SetBytes loops of
// whisk/types.go:85-95 and the Decoder calls of the FromReader methods do one at a
// time, subgroup check included.  status[i] != DecodeOK / DecodeInfinity marks a
// record SetBytes would have rejected.
func DecompressBatch(out []bls12381.G1Affine, status []byte, in []byte, subgroupCheck bool) error {
	n := len(in) / 48
	if len(in)%48 != 0 || len(out) != n || len(status) != n {
		return errors.New("curdlemsm: DecompressBatch: mismatched lengths")
	}
	if n == 0 {
		return nil
	}
	sc := C.int(0)
	if subgroupCheck {
		sc = 1
	}
	return locked(func() C.int {
		return C.curdle_g1_decompress_batch((*C.uint8_t)(unsafe.Pointer(&in[0])), C.size_t(n), sc,
			(*C.uint64_t)(unsafe.Pointer(&out[0])), (*C.uint8_t)(unsafe.Pointer(&status[0])))
	})
}
