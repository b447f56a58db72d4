//go:build cgo && evg_mi355x

// gpu_multi.go -- several MI355X behind the BATCHED planner of gpu_planner.go (include/evg_sched.h, ABI 3.1: evg_multi_*; 3.2: evg_multi_selftest).
//
// north_star / SURVEY.md 8e: "Distros shard naturally across the 8 GPUs of one node with a single RCCL broadcast of the shared
// runnable-task pool over xGMI and a gather of the per-distro TaskQueue back to rank 0." The scheduler is ONE Go process that
// enqueues a job per distro (units/crons.go:303-332), so the sharding lives behind the C ABI: the library owns one context and
// one RCCL communicator per device, cuts the contiguous distro ranges (evg_balanced_ranges) and runs the tick; this file only
// decides WHEN a batch is worth spreading and hands the library the same two structs planBatch fills for one device.
//
// NEVER COMPILED HERE (no Go toolchain in the build image), like the rest of shim/. The library side is exercised by
// tests/test_gpu_multi_abi.py: a world of one through RCCL, 3 / 4 / 5 / 8 emulated ranks on one device, a failure injected at
// every (rank, phase) of a tick, evg_multi_abort, and the start-up self-check SetGPUDevices runs.
package scheduler

/*
#include <stdint.h>
#include "evg_sched.h"
*/
import "C"

import (
	"context"
	"sync"
	"sync/atomic"

	"github.com/evergreen-ci/evergreen/model"
	"github.com/evergreen-ci/evergreen/model/distro"
	"github.com/evergreen-ci/evergreen/model/task"
	"github.com/pkg/errors"
)

// gpuDevices is what SetGPUDevices last stored: the HIP ordinals the batched planner may use. One device (the default) keeps
// every call on gpuPool's per-goroutine contexts.
var (
	gpuDevices   = []int{0}
	gpuDevicesMu sync.Mutex
	gpuShard     *gpuMulti
)

// minTasksPerDevice: below this a batch stays on one device -- the broadcast of the pool costs more than the kernels it
// spreads (LAB_NOTES.md, rounds 1-4, section 4: 84.6 MB over xGMI against ~60 us of kernels for 1 M tasks on ONE MI355X; BASELINE config 5's
// 10 M tasks are where eight devices pay).
const minTasksPerDevice = 1 << 20

type gpuMulti struct {
	mu     sync.Mutex // one tick at a time: the evg_multi owns one pool
	m      *C.evg_multi
	n      int
	closed bool // SetGPUDevices replaced this object: a planBatch that still holds it falls back to one device
	dead   atomic.Bool // a tick outlived its deadline (EVG_E_TIMEOUT): the library refuses the object; shardFor skips it
}

// SetGPUDevices is the EXPLICIT opt-in to multi-device planning, called once at start-up (one device is the default and needs no
// call). With more than one device it builds the evg_multi -- contexts, streams, RCCL communicators -- right here and runs the
// library's self-check (evg_multi_selftest: a generated pool of mixed shape planned on the first device alone and over all of them
// must give identical outputs) BEFORE any tick depends on it: the N > 1 RCCL path had never met hardware when this file was
// written, and a hang or a mismatch there would stall the scheduler's only planning path. On any failure the error is returned and
// planning stays on one device.
func SetGPUDevices(devs []int) error {
	gpuDevicesMu.Lock()
	defer gpuDevicesMu.Unlock()
	if old := gpuShard; old != nil {
		old.mu.Lock() // waits for a tick in flight; a planBatch that took the pointer earlier sees `closed` under the same lock
		old.closed = true
		C.evg_multi_destroy(old.m)
		old.m = nil
		old.mu.Unlock()
		gpuShard = nil
	}
	gpuDevices = []int{0}
	if len(devs) > 0 {
		gpuDevices = []int{devs[0]}
	}
	gpuPool.dev = C.int(gpuDevices[0])
	if len(devs) < 2 {
		return nil
	}
	cdevs := make([]C.int32_t, len(devs))
	for i, d := range devs {
		cdevs[i] = C.int32_t(d)
	}
	// unit rows: planBatch stamps SortingValueBreakdown from unit_of_task + unit_breakdown (planner.go:475)
	m := C.evg_multi_create(ptr(cdevs), C.int32_t(len(devs)), C.EVG_MULTI_UNIT_ROWS)
	if m == nil {
		return errors.Errorf("evg_multi_create: %s; planning stays on device %d", C.GoString(C.evg_multi_last_error(nil)), gpuDevices[0])
	}
	C.evg_multi_set_deadline_ms(m, C.int64_t(gpuDeadlineMS)) // ABI 3.3: a tick that does not come back aborts the communicators by itself
	if rc := C.evg_multi_selftest(m); rc != C.EVG_OK {
		err := errors.Errorf("evg_multi_selftest over %d devices: %s (%d); planning stays on device %d", len(devs),
			C.GoString(C.evg_multi_last_error(m)), int(rc), gpuDevices[0])
		C.evg_multi_destroy(m)
		return err
	}
	gpuDevices = append([]int(nil), devs...)
	gpuShard = &gpuMulti{m: m, n: len(devs)}
	return nil
}

// shardFor returns the multi-device context when a batch of n tasks over D distros should be spread, nil otherwise.
func shardFor(n, D int) (*gpuMulti, error) {
	gpuDevicesMu.Lock()
	defer gpuDevicesMu.Unlock()
	k := len(gpuDevices)
	if gpuShard == nil || gpuShard.dead.Load() || k < 2 || D < k || n < k*minTasksPerDevice {
		return nil, nil
	}
	return gpuShard, nil
}

// plan is evg_plan_distros spread over the devices: the same input and output structs (host memory of the caller's arena),
// the same results bit for bit. The pool is packed once, broadcast, every device plans its distro range, the slices come back.
func (s *gpuMulti) plan(in *C.evg_plan_input, out *C.evg_plan_output) error {
	s.mu.Lock()
	defer s.mu.Unlock()
	if s.closed { // SetGPUDevices ran between shardFor and here: the handle is gone, never touch it
		return errGPUShardClosed
	}
	if rc := C.evg_multi_load(s.m, in, nil); rc != C.EVG_OK {
		return errors.Errorf("evg_multi_load: %s (%d)", C.GoString(C.evg_multi_last_error(s.m)), int(rc))
	}
	if rc := C.evg_multi_tick(s.m, in.now_ns); rc != C.EVG_OK {
		err := errors.Errorf("evg_multi_tick: %s (%d)", C.GoString(C.evg_multi_last_error(s.m)), int(rc))
		if rc == C.EVG_E_TIMEOUT { // a device wait outlived the deadline: the object refuses further ticks -- planning goes on on one device
			s.closed = true // (planBatch falls back to evg_plan_distros on errGPUShardClosed; SetGPUDevices may build a new one)
			go func(m *C.evg_multi) { C.evg_multi_destroy(m) }(s.m)
			s.m = nil           // (SetGPUDevices destroys old.m under this lock: evg_multi_destroy(NULL) is a no-op)
			s.dead.Store(true)  // shardFor skips it from now on (no gpuDevicesMu here: SetGPUDevices takes the two locks in the other order)
		}
		return err
	}
	if rc := C.evg_multi_results(s.m, out, nil); rc != C.EVG_OK {
		return errors.Errorf("evg_multi_results: %s (%d)", C.GoString(C.evg_multi_last_error(s.m)), int(rc))
	}
	return nil
}

var errGPUShardClosed = errors.New("the multi-device context was replaced by SetGPUDevices")

// PlanAllDistros is the batched cron's body: every (distro, queue) pair of a tick in ONE call -- on one device, or sharded by
// distro over the devices of SetGPUDevices when the tick is large enough. Per distro it returns what runTunablePlanner would
// have produced (scheduler/scheduler.go:35-52): the re-ordered, stamped tasks and the DistroQueueInfo.
func PlanAllDistros(ctx context.Context, ds []*distro.Distro, queues [][]task.Task) ([][]task.Task, []model.DistroQueueInfo, error) {
	if len(ds) != len(queues) {
		return nil, nil, errors.New("PlanAllDistros: one queue per distro")
	}
	return planBatch(ctx, ds, queues)
}
