//go:build cgo && evg_mi355x

// gpu_allocator.go -- the cgo binding of include/evg_sched.h for the HOST ALLOCATOR half of the hot path: a value of
// scheduler.HostAllocator (scheduler/host_allocator.go:15) that replaces UtilizationBasedHostAllocator
// (scheduler/utilization_based_host_allocator.go:26-129). Same package, same build tag and the same context pool as
// gpu_planner.go. NEVER COMPILED HERE (no Go toolchain in the build image); twinned call for call by
// tests/cpp/test_shim_twin.cpp, which runs with the HIP backend under `pytest -m gpu`.
//
// Registering it is one case in GetHostAllocator (scheduler/host_allocator.go:23-30):
//
//	case evergreen.HostAllocatorUtilizationMI355X:   // a new constant next to HostAllocatorUtilization (globals.go:314)
//	        return GPUHostAllocator
package scheduler

/*
#include <stdint.h>
#include "evg_sched.h"
*/
import "C"

import (
	"context"
	"time"
	"unsafe"

	"github.com/evergreen-ci/evergreen"
	"github.com/evergreen-ci/evergreen/model"
	"github.com/evergreen-ci/evergreen/model/host"
	"github.com/evergreen-ci/evergreen/model/task"
	"github.com/evergreen-ci/utility"
	"github.com/pkg/errors"
)

// providerClass: 0 not ephemeral, 1 ephemeral (ec2-fleet, mock), 2 docker -- ephemeral, and exempt from the max-hosts
// early-out (utilization_based_host_allocator.go:39,142; Distro.IsEphemeral, model/distro/distro.go:513-515;
// evergreen.ProviderSpawnable, globals.go:767-774).
func providerClass(provider string) C.int32_t {
	if provider == evergreen.ProviderNameDocker {
		return 2
	}
	if utility.StringSliceContains(evergreen.ProviderSpawnable, provider) {
		return 1
	}
	return 0
}

type allocResult struct {
	newHosts, freeHosts int
	err                 error
}

// allocateBatch runs the allocator for D HostAllocatorData in ONE library call. It writes CountFree / CountRequired into
// datas[i].DistroQueueInfo.TaskGroupInfos in place (utilization_based_host_allocator.go:106-109: units/host_allocator.go:268-277
// reads them afterwards) and maps the library's per-distro status onto the reference's error values.
func allocateBatch(ctx context.Context, datas []*HostAllocatorData) ([]allocResult, error) {
	now := time.Now() // time.Since(t.StartTime) (:345)
	g, err := gpuPool.get()
	if err != nil {
		return nil, err
	}
	defer gpuPool.put(g)

	D := len(datas)
	nHosts, nGroups := 0, 0
	for _, d := range datas {
		nHosts += len(d.ExistingHosts)
		nGroups += len(d.DistroQueueInfo.TaskGroupInfos)
	}
	// the running tasks of every host, fetched in ONE query -- exactly what getSoonToBeFreeHosts fetches per task-group bucket
	// (:322: task.Find(task.ByIds(runningTaskIds)))
	var runningIDs []string
	for _, d := range datas {
		for i := range d.ExistingHosts {
			if d.ExistingHosts[i].RunningTask != "" {
				runningIDs = append(runningIDs, d.ExistingHosts[i].RunningTask)
			}
		}
	}
	running := map[string]*task.Task{}
	if len(runningIDs) > 0 {
		found, err := task.Find(ctx, task.ByIds(runningIDs))
		if err != nil {
			return nil, errors.Wrap(err, "finding the tasks the hosts are running")
		}
		for i := range found {
			running[found[i].Id] = &found[i]
		}
	}

	bytes := uintptr(D)*(unsafe.Sizeof(C.evg_alloc_params{})+unsafe.Sizeof(C.evg_distro_info{})+3*4) + 2*uintptr(D+1)*4 +
		uintptr(nHosts)*(1+4+3*8) + uintptr(D+nGroups)*unsafe.Sizeof(C.evg_group_info{}) + 64*20
	if err := g.reserve(bytes); err != nil {
		return nil, err
	}
	params := carveSlice[C.evg_alloc_params](g, D)
	hostOff, tgOff := carveSlice[C.int32_t](g, D+1), carveSlice[C.int32_t](g, D+1)
	hFlags, hKey := carveSlice[C.uint8_t](g, nHosts), carveSlice[C.int32_t](g, nHosts)
	hStart, hExp, hDev := carveSlice[C.int64_t](g, nHosts), carveSlice[C.int64_t](g, nHosts), carveSlice[C.int64_t](g, nHosts)
	distroInfo := carveSlice[C.evg_distro_info](g, D)
	groupInfo := carveSlice[C.evg_group_info](g, D+nGroups) // rows [0, D): the standalone buckets; D + key: the named groups
	for i := range groupInfo {
		groupInfo[i] = C.evg_group_info{}
	}
	newHosts, freeHosts, status := carveSlice[C.int32_t](g, D), carveSlice[C.int32_t](g, D), carveSlice[C.int32_t](g, D)

	groupRow := make([][]int, D) // TaskGroupInfos index -> group_info row, per distro
	h, key := 0, 0
	for di, data := range datas {
		d := &data.Distro
		s := d.HostAllocatorSettings
		params[di] = C.evg_alloc_params{
			future_host_fraction: C.double(s.FutureHostFraction), minimum_hosts: C.int32_t(s.MinimumHosts), maximum_hosts: C.int32_t(s.MaximumHosts),
			provider: providerClass(d.Provider), disabled: boolToC(d.Disabled),
			round_up:                   boolToC(s.RoundingRule == evergreen.HostAllocatorRoundUp),
			feedback_waits_over_thresh: boolToC(s.FeedbackRule == evergreen.HostAllocatorWaitsOverThreshFeedback),
		}
		hostOff[di], tgOff[di] = C.int32_t(h), C.int32_t(key)
		q := &data.DistroQueueInfo
		distroInfo[di] = C.evg_distro_info{
			expected_duration_ns: C.int64_t(q.ExpectedDuration), max_duration_threshold_ns: C.int64_t(q.MaxDurationThreshold),
			duration_over_threshold_ns: C.int64_t(q.DurationOverThreshold), length: C.int32_t(q.Length),
			length_with_dependencies_met:          C.int32_t(q.LengthWithDependenciesMet),
			count_dep_filled_merge_queue_tasks:    C.int32_t(q.CountDepFilledMergeQueueTasks),
			count_duration_over_threshold:         C.int32_t(q.CountDurationOverThreshold),
			count_wait_over_threshold:             C.int32_t(q.CountWaitOverThreshold),
			num_queued_large_parser_project_tasks: C.int32_t(q.NumQueuedLargeParserProjectTasks),
			secondary_queue:                       boolToC(q.SecondaryQueue), n_task_group_infos: C.int32_t(len(q.TaskGroupInfos)),
		}
		// the named groups of this distro's queue get keys in TaskGroupInfos order; "" is row di
		keyOf := make(map[string]int32, len(q.TaskGroupInfos))
		groupRow[di] = make([]int, len(q.TaskGroupInfos))
		for gi := range q.TaskGroupInfos {
			info := &q.TaskGroupInfos[gi]
			row := di
			if info.Name != "" {
				keyOf[info.Name] = int32(key)
				row = D + key
				key++
			}
			groupRow[di][gi] = row
			groupInfo[row] = C.evg_group_info{
				expected_duration_ns: C.int64_t(info.ExpectedDuration), duration_over_threshold_ns: C.int64_t(info.DurationOverThreshold),
				count: C.int32_t(info.Count), max_hosts: C.int32_t(info.MaxHosts),
				count_duration_over_threshold:      C.int32_t(info.CountDurationOverThreshold),
				count_wait_over_threshold:          C.int32_t(info.CountWaitOverThreshold),
				count_dep_filled_merge_queue_tasks: C.int32_t(info.CountDepFilledMergeQueueTasks), present: 1,
			}
		}
		for i := range data.ExistingHosts {
			eh := &data.ExistingHosts[i]
			var f uint8
			if eh.IsFree() { // RunningTask == "" && !IsTearingDown(), model/host/host.go:215-222
				f |= C.EVG_HF_FREE
			}
			hKey[h] = -1 // groupByTaskGroup (:208-224): the "" bucket unless the host runs a task of a task group
			hStart[h], hExp[h], hDev[h] = 0, 0, 0
			if eh.RunningTask != "" {
				f |= C.EVG_HF_RUNNING
				if eh.RunningTaskGroup != "" {
					if k, ok := keyOf[eh.GetTaskGroupString()]; ok { // model/host/host.go:668-670
						hKey[h] = C.int32_t(k)
					} else {
						hKey[h] = -2 // a group that has no tasks in the queue: its own bucket, never evaluated (:84-86)
					}
				}
				if t := running[eh.RunningTask]; t != nil {
					f |= C.EVG_HF_RUNNING_FOUND
					st := t.FetchExpectedDuration(ctx) // :342-344
					// unixNS: a Go-zero StartTime (dispatched, not started; a field lost in decoding) is EVG_TIME_GO_ZERO, so that the
					// library's saturating time_sub gives what time.Since(t.StartTime) gives (:345); UnixNano() of it is undefined
					hStart[h], hExp[h], hDev[h] = unixNS(t.StartTime), C.int64_t(st.Average), C.int64_t(st.StdDev)
				}
			}
			hFlags[h] = C.uint8_t(f)
			h++
		}
	}
	hostOff[D], tgOff[D] = C.int32_t(h), C.int32_t(key)

	// ptr() (gpu_planner.go), not &col[0]: the host columns are EMPTY for a distro without hosts -- the case in which the
	// allocator has to spawn some (NoExistingHosts, utilization_based_host_allocator_test.go:226-250).
	// max_concurrent_large_parser_project_tasks stays 0 (no limit): units/host_allocator.go:150 has already run
	// adjustForLargeParserProjectLimit on data.DistroQueueInfo before it calls the HostAllocator.
	in := C.evg_alloc_input{n_distros: C.int32_t(D), n_task_groups: C.int32_t(key), params: ptr(params), host_off: ptr(hostOff), tg_off: ptr(tgOff),
		distro_info: ptr(distroInfo), group_info: ptr(groupInfo), now_ns: C.int64_t(now.UnixNano())}
	in.hosts = C.evg_host_soa{n_hosts: C.int32_t(nHosts), flags: ptr(hFlags), tg_key: ptr(hKey), start_ts_ns: ptr(hStart),
		expected_duration_ns: ptr(hExp), duration_stddev_ns: ptr(hDev)}
	out := C.evg_alloc_output{new_hosts: ptr(newHosts), free_hosts: ptr(freeHosts), status: ptr(status)}
	batcher, err := batcherFor(nHosts, D) // one HostAllocator call = one distro (units/host_allocator.go:183-188): batched with its peers
	if err != nil {
		return nil, err
	}
	if batcher != nil {
		if err := batchedAllocate(batcher, &in, &out); err != nil {
			return nil, err
		}
	} else if rc := C.evg_allocate_hosts(g.c, &in, &out); rc != C.EVG_OK {
		g.dead = rc == C.EVG_E_TIMEOUT // (units/host_allocator.go:32 bounds the job at 10 min; the library's default deadline is 30 s)
		return nil, errors.Errorf("evg_allocate_hosts: %s (%d)", C.GoString(C.evg_last_error(g.c)), int(rc))
	}

	res := make([]allocResult, D)
	for di, data := range datas {
		q := &data.DistroQueueInfo
		for gi := range q.TaskGroupInfos { // in place, like :106-109 (the standalone row stays untouched there too)
			if q.TaskGroupInfos[gi].Name != "" {
				row := groupRow[di][gi]
				q.TaskGroupInfos[gi].CountFree = int(groupInfo[row].count_free)
				q.TaskGroupInfos[gi].CountRequired = int(groupInfo[row].count_required)
			}
		}
		res[di] = allocResult{newHosts: int(newHosts[di]), freeHosts: int(freeHosts[di])}
		switch status[di] { // the reference's own errors (:99-101,185-187,287-289), wrapped the way :100 wraps them
		case C.EVG_ALLOC_E_FUTURE_FRACTION:
			res[di].err = errors.Wrapf(errors.New("future host factor cannot be greater than 1"), "calculating hosts for distro '%s'", data.Distro.Id)
		case C.EVG_ALLOC_E_POOL_SIZE:
			res[di].err = errors.Wrapf(errors.Errorf("unable to plan hosts for distro %s due to pool size of %d", data.Distro.Id,
				data.Distro.HostAllocatorSettings.MaximumHosts), "calculating hosts for distro '%s'", data.Distro.Id) // :186 prints the DISTRO's maximum
		}
	}
	return res, nil
}

// GPUHostAllocator is a value of the reference's HostAllocator type (scheduler/host_allocator.go:15).
var GPUHostAllocator HostAllocator = func(ctx context.Context, data *HostAllocatorData) (int, int, error) {
	if ctx.Err() != nil { // utilization_based_host_allocator.go:146-148
		return 0, 0, errors.Wrap(ctx.Err(), "context canceled, not evaluating host utilization")
	}
	res, err := allocateBatch(ctx, []*HostAllocatorData{data})
	if err != nil {
		return 0, 0, err
	}
	return res[0].newHosts, res[0].freeHosts, res[0].err
}

var _ = host.Host{} // the ExistingHosts element type
var _ = model.DistroQueueInfo{}
