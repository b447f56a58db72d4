//go:build cgo && evg_mi355x

// gpu_batcher.go -- the reference's per-distro call shape served by batches (include/evg_sched.h, ABI 3.2: evg_batcher_*).
//
// The scheduler plans ONE distro per job, from concurrent amboy jobs (units/crons.go:303-332 enqueues them;
// units/scheduler.go:48-49 -> scheduler.PlanDistro -> runTunablePlanner, scheduler/scheduler.go:28-52), and allocates hosts the
// same way (units/host_allocator.go:183-188). A GPU call per distro keeps one of the chip's 256 CUs busy and the device's
// command path saturates at ~25 ms for 512 of them. So the one-distro calls of planBatch (GPUTaskPlanner, runGPUPlanner) and
// allocateBatch (GPUHostAllocator) go through ONE process-wide evg_batcher: the library collects the requests that arrive
// together (at most gpuBatchWindowUS apart, at most gpuBatchMax of them), plans them with one launch sequence and hands every
// goroutine its own slice of the result -- bit for bit what evg_plan_distros / evg_allocate_hosts return for the request alone,
// each request with its own time.Now(). Nothing changes for the caller: same function values, same errors, one request's
// contract violation is that request's error only. The batched cron (PlanAllDistros) already IS a batch and calls
// evg_plan_distros directly.
//
// NEVER COMPILED HERE (no Go toolchain in the build image); tests/cpp/test_shim_twin.cpp makes the same calls from
// concurrent std::threads, tests/test_batcher.py from 64 Python threads.
package scheduler

/*
#include <stdint.h>
#include "evg_sched.h"
*/
import "C"

import (
	"sync"
	"unsafe"

	"github.com/pkg/errors"
)

var (
	gpuBatcherMu     sync.Mutex
	gpuBatcher       *C.evg_batcher
	gpuBatchWindowUS = 200 // a request waits at most this long for company
	gpuBatchMax      = 64  // requests per batch
	gpuBatchOff      bool  // SetGPUBatching(false): every call goes straight to evg_plan_distros / evg_allocate_hosts
	// a request above this many tasks is a batch of its own: the batcher would pass it straight through anyway
	gpuBatchMaxTasks = 1 << 16
)

// SetGPUBatching configures the micro-batching front before the first planner call: on/off, the window in microseconds, the
// largest batch. Called at start-up, like SetGPUDevices.
func SetGPUBatching(on bool, windowUS, maxRequests int) {
	gpuBatcherMu.Lock()
	defer gpuBatcherMu.Unlock()
	gpuBatchOff = !on
	if windowUS >= 0 {
		gpuBatchWindowUS = windowUS
	}
	if maxRequests > 0 {
		gpuBatchMax = maxRequests
	}
}

// batcherFor returns the process-wide batcher for a request of n tasks over D distros, or nil when the request should be
// planned as its own batch (batching off, a multi-distro tick, a very large queue).
func batcherFor(n, D int) (*C.evg_batcher, error) {
	gpuBatcherMu.Lock()
	defer gpuBatcherMu.Unlock()
	if gpuBatchOff || D != 1 || n > gpuBatchMaxTasks {
		return nil, nil
	}
	if gpuBatcher == nil {
		b := C.evg_batcher_create(gpuPool.dev, C.int32_t(gpuBatchWindowUS), C.int32_t(gpuBatchMax))
		if b == nil { // no gfx950 device: there is no CPU fallback inside the library
			return nil, errors.Errorf("evg_batcher_create: %s", C.GoString(C.evg_last_error(nil)))
		}
		gpuBatcher = b
	}
	return gpuBatcher, nil
}

// batchedPlan / batchedAllocate: the two blocking calls, with the request's own error text.
func batchedPlan(b *C.evg_batcher, in *C.evg_plan_input, out *C.evg_plan_output) error {
	var msg [256]C.char // an array, not a slice: its address is always valid
	if rc := C.evg_batcher_plan(b, in, out, (*C.char)(unsafe.Pointer(&msg)), C.int32_t(len(msg))); rc != C.EVG_OK {
		return errors.Errorf("evg_batcher_plan: %s (%d)", C.GoString((*C.char)(unsafe.Pointer(&msg))), int(rc))
	}
	return nil
}

func batchedAllocate(b *C.evg_batcher, in *C.evg_alloc_input, out *C.evg_alloc_output) error {
	var msg [256]C.char
	if rc := C.evg_batcher_allocate(b, in, out, (*C.char)(unsafe.Pointer(&msg)), C.int32_t(len(msg))); rc != C.EVG_OK {
		return errors.Errorf("evg_batcher_allocate: %s (%d)", C.GoString((*C.char)(unsafe.Pointer(&msg))), int(rc))
	}
	return nil
}
