//go:build cgo && evg_mi355x

// gpu_batcher.go -- the reference's per-distro call shape served by batches (include/evg_sched.h, ABI 3.2: evg_batcher_*; 3.3: resident
// queues -- evg_batcher_plan_queue -- and bounded waits).
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
	gpuBatcherTimeouts int // slots of the current batcher that a deadline retired
	gpuDeadlineMS      = int64(30000) // SetGPUDeadline: every device wait of the shim's contexts / batcher / multi gives up after this
)

// SetGPUDeadline bounds every device wait behind the shim (evg_set_deadline_ms and its siblings; 0 = no limit). The reference bounds
// its jobs the same way: units/scheduler.go:18 (5 min), units/host_allocator.go:32 (10 min).
func SetGPUDeadline(ms int64) {
	gpuBatcherMu.Lock()
	defer gpuBatcherMu.Unlock()
	gpuDeadlineMS = ms
	if gpuBatcher != nil {
		C.evg_batcher_set_deadline_ms(gpuBatcher, C.int64_t(ms))
	}
}

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
		C.evg_batcher_set_deadline_ms(b, C.int64_t(gpuDeadlineMS))
		gpuBatcher, gpuBatcherTimeouts = b, 0
	}
	return gpuBatcher, nil
}

// fnv64 / hashWords: the two words that name a RESIDENT QUEUE (ABI 3.3). A distro's queue 15 s later
// (units/crons_remote_fifteen_second.go:21) is mostly the queue it was: the batcher keeps the packed columns of (queueID, generation)
// on the device, and a call with the same pair uploads its clock reading only. queueID = FNV-1a of the distro id; generation = FNV-1a
// over every packed column (planBatch has just written them: one pass over bytes that are in cache) -- any change of any field of any
// task of the queue changes it, now_ns does not.
const fnvOffset, fnvPrime = 14695981039346656037, 1099511628211

func fnv64(s string) uint64 {
	h := uint64(fnvOffset)
	for i := 0; i < len(s); i++ {
		h = (h ^ uint64(s[i])) * fnvPrime
	}
	return h
}

// hashWords folds `bytes` bytes at p into h: eight at a time (the arena's columns are 8-byte aligned), the tail byte by byte (what lies
// behind a column in the arena is left over from other batches and must not count).
func hashWords(h uint64, p unsafe.Pointer, bytes int) uint64 {
	for _, x := range unsafe.Slice((*uint64)(p), bytes/8) {
		h = (h ^ x) * fnvPrime
	}
	for _, x := range unsafe.Slice((*byte)(unsafe.Add(p, bytes&^7)), bytes&7) {
		h = (h ^ uint64(x)) * fnvPrime
	}
	return h
}

// batchedPlan / batchedAllocate: the two blocking calls, with the request's own error text. queueID 0: not resident.
// An EVG_E_TIMEOUT (a batch outlived evg_batcher_set_deadline_ms: the job's thread comes back, the scheduler job fails and runs again on
// the next tick -- units/scheduler.go:18 bounds the job itself at 5 min) retires one of the batcher's four slots; when the library
// reports that none is left the process-wide batcher is replaced.
func batchedPlan(b *C.evg_batcher, queueID, generation uint64, in *C.evg_plan_input, out *C.evg_plan_output) error {
	var msg [256]C.char // an array, not a slice: its address is always valid
	rc := C.evg_batcher_plan_queue(b, C.uint64_t(queueID), C.uint64_t(generation), in, out, (*C.char)(unsafe.Pointer(&msg)), C.int32_t(len(msg)))
	if rc != C.EVG_OK {
		if rc == C.EVG_E_TIMEOUT {
			retireBatcher(b)
		}
		return errors.Errorf("evg_batcher_plan_queue: %s (%d)", C.GoString((*C.char)(unsafe.Pointer(&msg))), int(rc))
	}
	return nil
}

// retireBatcher: after a timeout the next request gets a fresh batcher once the old one has no slot left; the old object is closed
// (evg_batcher_close: callers inside it leave with their errors) and destroyed off the caller's goroutine.
func retireBatcher(b *C.evg_batcher) {
	var st C.evg_batcher_stats
	gpuBatcherMu.Lock()
	defer gpuBatcherMu.Unlock()
	if gpuBatcher != b || C.evg_batcher_get_stats(b, &st) != C.EVG_OK {
		return
	}
	gpuBatcherTimeouts++
	if gpuBatcherTimeouts >= 4 { // every slot gone: the library refuses from here on
		gpuBatcher, gpuBatcherTimeouts = nil, 0
		go func() { C.evg_batcher_close(b); C.evg_batcher_destroy(b) }()
	}
}

func batchedAllocate(b *C.evg_batcher, in *C.evg_alloc_input, out *C.evg_alloc_output) error {
	var msg [256]C.char
	if rc := C.evg_batcher_allocate(b, in, out, (*C.char)(unsafe.Pointer(&msg)), C.int32_t(len(msg))); rc != C.EVG_OK {
		return errors.Errorf("evg_batcher_allocate: %s (%d)", C.GoString((*C.char)(unsafe.Pointer(&msg))), int(rc))
	}
	return nil
}
