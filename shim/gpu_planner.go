//go:build cgo && evg_mi355x

// gpu_planner.go -- the cgo binding of include/evg_sched.h for the PLANNER half of Evergreen's per-distro scheduling hot
// path. Drop it into package scheduler of evergreen-ci/evergreen (next to scheduler/scheduler.go) together with
// gpu_allocator.go, and build with `-tags evg_mi355x`.
//
// NEVER COMPILED HERE: the build image has no Go toolchain, no module cache and no network. The file is complete (every
// helper it names is in it) and is twinned by tests/cpp/test_shim_twin.cpp, a C++ driver that performs the identical call
// sequence -- evg_create, evg_check_abi, the evg_host_alloc arena, first-appearance interning, evg_plan_distros, the stamp
// loop, evg_allocate_hosts with the in-place CountFree / CountRequired write-back -- on the reference's known-answer cases
// with the HIP backend (`pytest -m gpu`). Field names are the reference's at the snapshot under /root/reference.
//
// What it replaces (file:line in the reference):
//
//	scheduler/scheduler.go:43  plan := PrepareTasksForPlanning(ctx, d, tasks).Export(ctx)
//	scheduler/scheduler.go:44  info := GetDistroQueueInfo(ctx, d, plan, opts)
//
// and what stays in Go: PopulateCaches (setup_funcs.go:18-67), info.SecondaryQueue / PlanCreatedAt (scheduler.go:45-46),
// PersistTaskQueue (scheduler.go:47), every DB access.
//
// cgo rules this file keeps: C never retains Go memory (every entry point is synchronous and copies in / out); no Go
// pointer to a Go pointer crosses (the structs hold pointers into ONE C arena, evg_host_alloc memory, or into Go slices
// pinned with runtime.Pinner for the call); a cgo call pins its OS thread, and the calls are bounded (copies + kernels).
package scheduler

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/evg_sched/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/evg_sched/lib -levg_sched
#include <stdint.h>
#include <stdlib.h>
#include "evg_sched.h"
*/
import "C"

import (
	"context"
	"runtime"
	"sync"
	"time"
	"unsafe"

	"github.com/evergreen-ci/evergreen"
	"github.com/evergreen-ci/evergreen/model"
	"github.com/evergreen-ci/evergreen/model/distro"
	"github.com/evergreen-ci/evergreen/model/task"
	"github.com/evergreen-ci/utility"
	"github.com/pkg/errors"
)

// ---- contexts -------------------------------------------------------------------------------------------------------

// gpuCtx is one evg_ctx plus the page-locked arena its batches are packed into. include/evg_sched.h: one context per
// goroutine / OS thread, one per batch in flight; contexts are independent, so amboy's concurrent distro jobs
// (units/scheduler.go:48-49) each take their own from the pool.
type gpuCtx struct {
	c     *C.evg_ctx
	arena unsafe.Pointer // evg_host_alloc memory, grown when a batch needs more, re-used tick after tick
	size  uintptr
	used  uintptr
	dead  bool // a call on it came back EVG_E_TIMEOUT (ABI 3.3): the library refuses it from now on; put() destroys it
}

type gpuCtxPool struct {
	mu   sync.Mutex
	free []*gpuCtx
	dev  C.int
	once sync.Once
	err  error
}

var gpuPool = &gpuCtxPool{dev: 0}

func (p *gpuCtxPool) get() (*gpuCtx, error) {
	p.once.Do(func() { // refuse a library this file was not written against (struct sizes included)
		if rc := C.evg_check_abi(C.int32_t(C.EVG_ABI_MAJOR), C.int32_t(C.EVG_ABI_MINOR), C.size_t(unsafe.Sizeof(C.evg_plan_input{})),
			C.size_t(unsafe.Sizeof(C.evg_plan_output{})), C.size_t(unsafe.Sizeof(C.evg_alloc_input{})), C.size_t(unsafe.Sizeof(C.evg_group_info{}))); rc != C.EVG_OK {
			p.err = errors.Errorf("libevg_sched ABI %#x does not match this binding (%d.%d)", int(C.evg_abi_version()), int(C.EVG_ABI_MAJOR), int(C.EVG_ABI_MINOR))
		}
	})
	if p.err != nil {
		return nil, p.err
	}
	p.mu.Lock()
	if n := len(p.free); n > 0 {
		g := p.free[n-1]
		p.free = p.free[:n-1]
		p.mu.Unlock()
		return g, nil
	}
	p.mu.Unlock()
	c := C.evg_create(p.dev)
	if c == nil { // no gfx950 device: there is no CPU fallback inside the library
		return nil, errors.Errorf("evg_create: %s", C.GoString(C.evg_last_error(nil)))
	}
	C.evg_set_deadline_ms(c, C.int64_t(gpuDeadlineMS)) // SetGPUDeadline (gpu_batcher.go); the library's default is 30 s
	return &gpuCtx{c: c}, nil
}

func (p *gpuCtxPool) put(g *gpuCtx) {
	if g.dead { // poisoned by an expired deadline: evg_destroy waits once more and then leaks rather than blocks -- off this goroutine
		go C.evg_destroy(g.c)
		return
	}
	g.used = 0
	p.mu.Lock()
	p.free = append(p.free, g)
	p.mu.Unlock()
}

// reserve makes the arena at least `bytes` large (contents are not preserved: a batch is packed from scratch).
func (g *gpuCtx) reserve(bytes uintptr) error {
	g.used = 0
	if bytes <= g.size {
		return nil
	}
	if g.arena != nil {
		C.evg_host_free(g.c, g.arena)
		g.arena, g.size = nil, 0
	}
	want := bytes + bytes/4 + 4096
	p := C.evg_host_alloc(g.c, C.size_t(want))
	if p == nil {
		return errors.Errorf("evg_host_alloc(%d): %s", want, C.GoString(C.evg_last_error(g.c)))
	}
	g.arena, g.size = p, want
	return nil
}

// carve returns `count` elements of `elem` bytes from the arena, 64-byte aligned. reserve() sized the arena for the batch.
func (g *gpuCtx) carve(count int, elem uintptr) unsafe.Pointer {
	g.used = (g.used + 63) &^ 63
	p := unsafe.Add(g.arena, g.used)
	g.used += uintptr(count)*elem + elem // one spare element: a zero-length column still has a valid address
	return p
}

func carveSlice[T any](g *gpuCtx, n int) []T {
	var z T
	return unsafe.Slice((*T)(g.carve(n, unsafe.Sizeof(z))), n)
}

// ptr is the address a C struct field gets for a carved column: the slice's data pointer, valid (and inside the arena) even
// when the slice is EMPTY -- a queue without dependencies, a distro without hosts, an empty queue. Never &s[0]: Go
// bounds-checks the index under the & and panics on a zero-length slice (the spare element carve() adds lies past len).
func ptr[T any](s []T) *T { return unsafe.SliceData(s) }

// ---- small conversions ----------------------------------------------------------------------------------------------

// unixNS maps a time.Time onto the ABI's clock: Go's zero Time (Time.IsZero) is EVG_TIME_GO_ZERO, everything else Unix
// nanoseconds (0 == the Unix epoch == utility.ZeroTime).
func unixNS(t time.Time) C.int64_t {
	if t.IsZero() {
		return C.INT64_MIN
	}
	return C.int64_t(t.UnixNano())
}

func boolToC(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

// intern returns the dense key of s in m, numbering keys in order of first appearance from *next on.
func intern(m map[string]int32, s string, next *int) int32 {
	if k, ok := m[s]; ok {
		return k
	}
	k := int32(*next)
	m[s] = k
	*next++
	return k
}

// statusClass: what a dependent reads of a dependency's Task.Status (task.go:546-561).
func statusClass(status string) uint32 {
	switch status {
	case evergreen.TaskSucceeded:
		return 1
	case evergreen.TaskFailed:
		return 2
	}
	return 0
}

// taskFlags packs the EVG_TF_* bits of one task of distro d (SURVEY 8b' column map).
func taskFlags(t *task.Task, d *distro.Distro) C.uint16_t {
	var f uint32
	switch { // planner.go:308-312, globals.go:798-804,1224-1242
	case t.Requester == evergreen.GithubMergeRequester:
		f = C.EVG_TF_REQ_MERGE
	case evergreen.IsPatchRequester(t.Requester):
		f = C.EVG_TF_REQ_PATCH
	}
	if t.GenerateTask {
		f |= C.EVG_TF_GENERATE
	}
	if t.ActivatedBy == evergreen.StepbackTaskActivator {
		f |= C.EVG_TF_STEPBACK
	}
	if t.OverrideDependencies {
		f |= C.EVG_TF_OVERRIDE_DEPS
	}
	if t.DistroId != d.Id {
		f |= C.EVG_TF_OTHER_DISTRO
	}
	if t.CachedProjectStorageMethod == evergreen.ProjectStorageMethodS3 {
		f |= C.EVG_TF_S3_STORAGE
	}
	if t.Blocked() {
		f |= C.EVG_TF_BLOCKED
	}
	f |= statusClass(t.Status) << C.EVG_TF_STATUS_SHIFT
	return C.uint16_t(f)
}

// depRequired: SatisfiesDependency (task.go:546-561) decides at the FIRST DependsOn entry for that id whose Status it
// recognises: 0 "" / success, 1 failed, 2 "*", 3 none recognised (never satisfied).
func depRequired(t *task.Task, id string) uint8 {
	for _, d := range t.DependsOn {
		if d.TaskId != id {
			continue
		}
		switch d.Status {
		case evergreen.TaskSucceeded, "":
			return 0
		case evergreen.TaskFailed:
			return 1
		case task.AllStatuses:
			return 2
		}
	}
	return 3
}

// fetchedDepStates resolves the dependencies that are NOT in any of the queues being planned: one task.Find for all of
// them (what Task.DependenciesMet -> populateDependencyTaskCache reads one task at a time, task.go:649-688,703-740).
// A task the DB does not hold is MISSING: DependenciesMet errors and checkDependenciesMet reports unmet (scheduler.go:180-186).
func fetchedDepStates(ctx context.Context, ids []string) (map[string]uint8, error) {
	states := make(map[string]uint8, len(ids))
	if len(ids) == 0 {
		return states, nil
	}
	found, err := task.FindWithFields(ctx, task.ByIds(ids), task.StatusKey, task.DependsOnKey, task.OverrideDependenciesKey)
	if err != nil {
		return nil, errors.Wrap(err, "fetching dependencies that are not in the queue")
	}
	for i := range found {
		s := uint8(statusClass(found[i].Status) << C.EVG_DEP_STATE_SHIFT)
		if found[i].Blocked() {
			s |= C.EVG_DEP_BLOCKED
		}
		states[found[i].Id] = s
	}
	for _, id := range ids {
		if _, ok := states[id]; !ok {
			states[id] = C.EVG_DEP_MISSING
		}
	}
	return states, nil
}

// breakdownOfUnit reads unit slot u of the field-major table (field f at [f*nSlots+u], enum evg_breakdown_field).
func breakdownOfUnit(ub []C.int64_t, u, nSlots int) task.SortingValueBreakdown {
	f := func(k int) int64 { return int64(ub[k*nSlots+u]) }
	return task.SortingValueBreakdown{
		TaskGroupLength: f(C.EVG_BD_TASK_GROUP_LENGTH),
		TotalValue:      f(C.EVG_BD_TOTAL_VALUE),
		PriorityBreakdown: task.PriorityBreakdown{
			InitialPriorityImpact: f(C.EVG_BD_PRI_INITIAL), TaskGroupImpact: f(C.EVG_BD_PRI_TASK_GROUP),
			GeneratorTaskImpact: f(C.EVG_BD_PRI_GENERATOR), CommitQueueImpact: f(C.EVG_BD_PRI_COMMIT_QUEUE),
		},
		RankValueBreakdown: task.RankValueBreakdown{
			CommitQueueImpact: f(C.EVG_BD_RANK_COMMIT_QUEUE), NumDependentsImpact: f(C.EVG_BD_RANK_NUM_DEPENDENTS),
			EstimatedRuntimeImpact: f(C.EVG_BD_RANK_EST_RUNTIME), MainlineWaitTimeImpact: f(C.EVG_BD_RANK_MAINLINE_WAIT),
			StepbackImpact: f(C.EVG_BD_RANK_STEPBACK), PatchImpact: f(C.EVG_BD_RANK_PATCH), PatchWaitTimeImpact: f(C.EVG_BD_RANK_PATCH_WAIT),
		},
	}
}

// depsMetTime is Task.setDependenciesMetTime (task.go:690-701) with the batch's clock.
func depsMetTime(t *task.Task, now time.Time) time.Time {
	met := utility.ZeroTime
	for _, dep := range t.DependsOn {
		if !utility.IsZeroTime(dep.FinishedAt) && dep.FinishedAt.After(met) {
			met = dep.FinishedAt
		}
	}
	if utility.IsZeroTime(met) {
		met = now
	}
	return met
}

// queueInfoFromRows builds model.DistroQueueInfo of distro d from the info rows: row d of group_info is the standalone
// bucket (Name ""), row D + k task-group key k; only rows with present == 1 exist in the reference's TaskGroupInfos
// (scheduler.go:98-112,161-175).
func queueInfoFromRows(di []C.evg_distro_info, gi []C.evg_group_info, d, D int, tgOff []C.int32_t, tgNames []string) model.DistroQueueInfo {
	i := di[d]
	info := model.DistroQueueInfo{
		Length:                           int(i.length),
		LengthWithDependenciesMet:        int(i.length_with_dependencies_met),
		CountDepFilledMergeQueueTasks:    int(i.count_dep_filled_merge_queue_tasks),
		ExpectedDuration:                 time.Duration(i.expected_duration_ns),
		MaxDurationThreshold:             time.Duration(i.max_duration_threshold_ns),
		CountDurationOverThreshold:       int(i.count_duration_over_threshold),
		DurationOverThreshold:            time.Duration(i.duration_over_threshold_ns),
		CountWaitOverThreshold:           int(i.count_wait_over_threshold),
		NumQueuedLargeParserProjectTasks: int(i.num_queued_large_parser_project_tasks),
		SecondaryQueue:                   i.secondary_queue != 0, // runTunablePlanner overwrites it (scheduler.go:45)
		TaskGroupInfos:                   make([]model.TaskGroupInfo, 0, int(i.n_task_group_infos)),
	}
	add := func(g C.evg_group_info, name string) {
		if g.present == 0 {
			return
		}
		info.TaskGroupInfos = append(info.TaskGroupInfos, model.TaskGroupInfo{
			Name: name, Count: int(g.count), CountFree: int(g.count_free), CountRequired: int(g.count_required), MaxHosts: int(g.max_hosts),
			ExpectedDuration: time.Duration(g.expected_duration_ns), CountDurationOverThreshold: int(g.count_duration_over_threshold),
			CountWaitOverThreshold: int(g.count_wait_over_threshold), CountDepFilledMergeQueueTasks: int(g.count_dep_filled_merge_queue_tasks),
			DurationOverThreshold: time.Duration(g.duration_over_threshold_ns),
		})
	}
	add(gi[d], "")
	for k := int(tgOff[d]); k < int(tgOff[d+1]); k++ {
		add(gi[D+k], tgNames[k])
	}
	return info
}

// ---- the batch ----------------------------------------------------------------------------------------------------

// planBatch plans D (distro, queue) pairs in ONE library call: the batched cron's shape (units/crons.go:303-332 with one
// job for all distros) and, with D == 1, the body of the TaskPlanner value below. Returns, per distro, the SAME task
// values re-ordered and stamped (planner_test.go:493,507,525) and the DistroQueueInfo of GetDistroQueueInfo.
func planBatch(ctx context.Context, ds []*distro.Distro, queues [][]task.Task) ([][]task.Task, []model.DistroQueueInfo, error) {
	now := time.Now() // replaces every time.Since()/time.Now() on the path (planner.go:319-321, scheduler.go:141)
	g, err := gpuPool.get()
	if err != nil {
		return nil, nil, err
	}
	defer gpuPool.put(g)

	D := len(ds)
	n, e := 0, 0
	for _, q := range queues {
		n += len(q)
		for i := range q {
			e += len(q[i].DependsOn)
		}
	}
	// Dependencies that are not in the SAME distro's queue: fetched once, before packing. "In the queue" is per distro (the
	// planner's cache and GetDistroQueueInfo's depCache hold one distro's tasks, planner.go:453, scheduler.go:62-65): a
	// dependency that sits only in ANOTHER distro's queue of this batch is an out-of-queue dependency here and needs its
	// fetched state like any other.
	var outside []string
	{
		seen := map[string]struct{}{}
		for _, q := range queues {
			inQueue := make(map[string]struct{}, len(q))
			for i := range q {
				inQueue[q[i].Id] = struct{}{}
			}
			for i := range q {
				for _, dep := range q[i].DependsOn {
					if _, ok := inQueue[dep.TaskId]; ok {
						continue
					}
					if _, ok := seen[dep.TaskId]; !ok {
						seen[dep.TaskId] = struct{}{}
						outside = append(outside, dep.TaskId)
					}
				}
			}
		}
	}
	depState, err := fetchedDepStates(ctx, outside)
	if err != nil {
		return nil, nil, err
	}

	// Upper bound of the unit slots (one per task, task group and version): sizes the arena before interning.
	maxSlots := 3*n + 1
	bytes := uintptr(n)*(5*8+5*4+2) + uintptr(n+1)*4 + uintptr(e)*(4+1+8) + // task columns, CSR
		uintptr(D)*unsafe.Sizeof(C.evg_distro_params{}) + 3*uintptr(D+1)*4 + // per-distro tables
		uintptr(n)*(4+1+8+4) + uintptr(maxSlots)*C.EVG_BREAKDOWN_FIELDS*8 + // order, deps_met, wait_ns, unit_of_task, unit rows
		uintptr(D)*unsafe.Sizeof(C.evg_distro_info{}) + uintptr(D+n)*unsafe.Sizeof(C.evg_group_info{}) + 64*40
	if err := g.reserve(bytes); err != nil {
		return nil, nil, err
	}
	priority, expDur, queueTS := carveSlice[C.int64_t](g, n), carveSlice[C.int64_t](g, n), carveSlice[C.int64_t](g, n)
	schedTS, metTS := carveSlice[C.int64_t](g, n), carveSlice[C.int64_t](g, n)
	numDep, tgOrder, tgMaxHosts := carveSlice[C.int32_t](g, n), carveSlice[C.int32_t](g, n), carveSlice[C.int32_t](g, n)
	tgKey, verKey := carveSlice[C.int32_t](g, n), carveSlice[C.int32_t](g, n)
	flags := carveSlice[C.uint16_t](g, n)
	depOff := carveSlice[C.int32_t](g, n+1)
	depIdx, depInfo, depFin := carveSlice[C.int32_t](g, e), carveSlice[C.uint8_t](g, e), carveSlice[C.int64_t](g, e)
	params := carveSlice[C.evg_distro_params](g, D)
	taskOff, tgOff, verOff := carveSlice[C.int32_t](g, D+1), carveSlice[C.int32_t](g, D+1), carveSlice[C.int32_t](g, D+1)

	var tgNames []string // task-group key -> GetTaskGroupString()
	nTG, nVer, row, edge := 0, 0, 0, 0
	depOff[0] = 0
	for di, d := range ds {
		taskOff[di], tgOff[di], verOff[di] = C.int32_t(row), C.int32_t(nTG), C.int32_t(nVer)
		ps := d.PlannerSettings // RAW values: the library applies the <=0 -> 1 getters and the target-time defaults
		params[di] = C.evg_distro_params{
			patch_factor: C.int64_t(ps.PatchFactor), patch_time_in_queue_factor: C.int64_t(ps.PatchTimeInQueueFactor),
			commit_queue_factor: C.int64_t(ps.CommitQueueFactor), mainline_time_in_queue_factor: C.int64_t(ps.MainlineTimeInQueueFactor),
			expected_runtime_factor: C.int64_t(ps.ExpectedRuntimeFactor), generate_task_factor: C.int64_t(ps.GenerateTaskFactor),
			stepback_task_factor: C.int64_t(ps.StepbackTaskFactor), num_dependents_factor: C.double(ps.NumDependentsFactor),
			target_time_ns: C.int64_t(ps.TargetTime), merge_queue_target_time_ns: C.int64_t(ps.MergeQueueTargetTime),
			group_versions:        boolToC(ps.ShouldGroupVersions()),
			includes_dependencies: boolToC(d.DispatcherSettings.Version == evergreen.DispatcherVersionRevisedWithDependencies), // scheduler.go:29
		}
		q := queues[di]
		rowOf := make(map[string]int32, len(q)) // the planner's cache.Exists(dep.TaskId), planner.go:453; a later duplicate id wins, like the Go map
		for i := range q {
			rowOf[q[i].Id] = int32(row + i)
		}
		tgKeys, verKeys := map[string]int32{}, map[string]int32{} // first-appearance interning, per distro
		for i := range q {
			t := &q[i]
			r := row + i
			priority[r] = C.int64_t(t.Priority)
			expDur[r] = C.int64_t(t.FetchExpectedDuration(ctx).Average) // cached by PopulateCaches (setup_funcs.go:18-67)
			qt := t.ActivatedTime // planner.go:318-322
			if qt.IsZero() {
				qt = t.IngestTime
			}
			queueTS[r], schedTS[r], metTS[r] = unixNS(qt), unixNS(t.ScheduledTime), unixNS(t.DependenciesMetTime)
			numDep[r], tgOrder[r], tgMaxHosts[r] = C.int32_t(t.NumDependents), C.int32_t(t.TaskGroupOrder), C.int32_t(t.TaskGroupMaxHosts)
			tgKey[r] = -1
			if t.TaskGroup != "" {
				before := nTG
				tgKey[r] = C.int32_t(intern(tgKeys, t.GetTaskGroupString(), &nTG)) // task.go:436-438
				if nTG != before {
					tgNames = append(tgNames, t.GetTaskGroupString())
				}
			}
			verKey[r] = C.int32_t(intern(verKeys, t.Version, &nVer))
			flags[r] = taskFlags(t, d)
			for _, dep := range t.DependsOn {
				info := depRequired(t, dep.TaskId)
				idx := int32(-1)
				if j, ok := rowOf[dep.TaskId]; ok {
					idx = j
				} else {
					info |= depState[dep.TaskId]
				}
				depIdx[edge], depInfo[edge] = C.int32_t(idx), C.uint8_t(info)
				depFin[edge] = 0
				if !dep.FinishedAt.IsZero() {
					depFin[edge] = C.int64_t(dep.FinishedAt.UnixNano())
				}
				edge++
			}
			depOff[r+1] = C.int32_t(edge)
		}
		row += len(q)
	}
	taskOff[D], tgOff[D], verOff[D] = C.int32_t(row), C.int32_t(nTG), C.int32_t(nVer)

	nSlots := n + nTG + nVer // unit slots of the batch (evg_plan_output.unit_breakdown)
	order, unitOf := carveSlice[C.int32_t](g, n), carveSlice[C.int32_t](g, n)
	met, wait := carveSlice[C.uint8_t](g, n), carveSlice[C.int64_t](g, n)
	unitRows := carveSlice[C.int64_t](g, nSlots*C.EVG_BREAKDOWN_FIELDS)
	distroInfo, groupInfo := carveSlice[C.evg_distro_info](g, D), carveSlice[C.evg_group_info](g, D+nTG)

	in := C.evg_plan_input{n_distros: C.int32_t(D), n_task_groups: C.int32_t(nTG), n_versions: C.int32_t(nVer),
		distros: ptr(params), task_off: ptr(taskOff), tg_off: ptr(tgOff), ver_off: ptr(verOff), now_ns: C.int64_t(now.UnixNano())}
	// ptr(), not &col[0]: any of these columns can be empty (e == 0 for a queue without DependsOn -- most of planner_test.go --
	// and n == 0 for an empty queue, which scheduler/wrapper.go:107 still hands to PrioritizeTasks)
	in.tasks = C.evg_task_soa{n_tasks: C.int32_t(n), n_edges: C.int32_t(e),
		priority: ptr(priority), expected_duration_ns: ptr(expDur), queue_ts_ns: ptr(queueTS), scheduled_ts_ns: ptr(schedTS),
		deps_met_ts_ns: ptr(metTS), num_dependents: ptr(numDep), task_group_order: ptr(tgOrder), task_group_max_hosts: ptr(tgMaxHosts),
		tg_key: ptr(tgKey), version_key: ptr(verKey), flags: ptr(flags), dep_off: ptr(depOff), dep_idx: ptr(depIdx), dep_info: ptr(depInfo),
		dep_finished_ts_ns: ptr(depFin)}
	out := C.evg_plan_output{order: ptr(order), deps_met: ptr(met), wait_ns: ptr(wait), distro_info: ptr(distroInfo), group_info: ptr(groupInfo),
		unit_of_task: ptr(unitOf), unit_breakdown: ptr(unitRows)} // breakdown (rows by task) and n_units stay NULL
	// Everything the structs point at is C memory (the arena): nothing to pin. (With Go slices instead:
	// var pin runtime.Pinner; pin.Pin(&col[0]) for every column; defer pin.Unpin().)
	var pin runtime.Pinner
	defer pin.Unpin()

	// One device, or -- a tick large enough to pay for the broadcast, gpu_multi.go -- the same two structs spread over the
	// devices of SetGPUDevices by the library (evg_multi_*): identical results either way.
	shard, err := shardFor(n, D)
	if err != nil {
		return nil, nil, err
	}
	// A batch of ONE distro -- the reference's own call shape (scheduler/scheduler.go:28-52) -- joins whatever other goroutines
	// are planning at this moment (gpu_batcher.go): one launch sequence for all of them, the same result for each.
	batcher, err := batcherFor(n, D)
	if err != nil {
		return nil, nil, err
	}
	if shard != nil {
		err := shard.plan(&in, &out)
		if err == errGPUShardClosed { // replaced while this batch was being packed: one device plans it, same result
			err = nil
			if rc := C.evg_plan_distros(g.c, &in, &out); rc != C.EVG_OK {
				g.dead = rc == C.EVG_E_TIMEOUT
				return nil, nil, errors.Errorf("evg_plan_distros: %s (%d)", C.GoString(C.evg_last_error(g.c)), int(rc))
			}
		}
		if err != nil {
			return nil, nil, err
		}
	} else if batcher != nil {
		// the queue by name and content (gpu_batcher.go): the same queue as 15 s ago travels as a clock reading
		gen := uint64(fnvOffset)
		for _, c := range []struct {
			p unsafe.Pointer
			b int
		}{{unsafe.Pointer(ptr(priority)), 8 * n}, {unsafe.Pointer(ptr(expDur)), 8 * n}, {unsafe.Pointer(ptr(queueTS)), 8 * n}, {unsafe.Pointer(ptr(schedTS)), 8 * n},
			{unsafe.Pointer(ptr(metTS)), 8 * n}, {unsafe.Pointer(ptr(numDep)), 4 * n}, {unsafe.Pointer(ptr(tgOrder)), 4 * n}, {unsafe.Pointer(ptr(tgMaxHosts)), 4 * n},
			{unsafe.Pointer(ptr(tgKey)), 4 * n}, {unsafe.Pointer(ptr(verKey)), 4 * n}, {unsafe.Pointer(ptr(flags)), 2 * n}, {unsafe.Pointer(ptr(depOff)), 4 * (n + 1)},
			{unsafe.Pointer(ptr(depIdx)), 4 * e}, {unsafe.Pointer(ptr(depInfo)), e}, {unsafe.Pointer(ptr(depFin)), 8 * e},
			{unsafe.Pointer(ptr(params)), int(unsafe.Sizeof(params[0])) * D}} {
			if c.b > 0 {
				gen = hashWords(gen, c.p, c.b)
			}
		}
		gen = (gen ^ (uint64(nTG)<<32 | uint64(nVer))) * fnvPrime // the sizes of the two key ranges
		if err := batchedPlan(batcher, fnv64(ds[0].Id)|1, gen, &in, &out); err != nil {
			return nil, nil, err
		}
	} else if rc := C.evg_plan_distros(g.c, &in, &out); rc != C.EVG_OK {
		g.dead = rc == C.EVG_E_TIMEOUT // the job fails and runs again on the next tick, on a fresh context (units/scheduler.go:18 bounds the job too)
		return nil, nil, errors.Errorf("evg_plan_distros: %s (%d)", C.GoString(C.evg_last_error(g.c)), int(rc))
	}

	plans := make([][]task.Task, D)
	infos := make([]model.DistroQueueInfo, D)
	for di := range ds {
		lo, hi := int(taskOff[di]), int(taskOff[di+1])
		plan := make([]task.Task, 0, hi-lo)
		for p := lo; p < hi; p++ {
			r := int(order[p])
			t := queues[di][r-lo] // the same task value, re-ordered
			// the unit the task was emitted from; its row is what planner.go:475 stamps on the task
			t.SetSortingValueBreakdownAttributes(ctx, breakdownOfUnit(unitRows, int(unitOf[r]), nSlots))
			t.ExpectedDuration = time.Duration(expDur[r])         // scheduler.go:125
			t.WaitSinceDependenciesMet = time.Duration(wait[r])   // scheduler.go:141
			if met[r] != 0 && utility.IsZeroTime(t.DependenciesMetTime) && len(t.DependsOn) > 0 && !t.OverrideDependencies {
				t.DependenciesMetTime = depsMetTime(&t, now) // setDependenciesMetTime, task.go:690-701 (the DB write of :675 stays with the caller)
			}
			plan = append(plan, t)
		}
		plans[di] = plan
		infos[di] = queueInfoFromRows(distroInfo, groupInfo, di, D, tgOff, tgNames)
	}
	return plans, infos, nil
}

// GPUTaskPlanner is a value of the reference's TaskPlanner type (scheduler/scheduler.go:26): a batch of one.
var GPUTaskPlanner TaskPlanner = func(d *distro.Distro, tasks []task.Task, opts TaskPlannerOptions) ([]task.Task, error) {
	plans, _, err := planBatch(context.Background(), []*distro.Distro{d}, [][]task.Task{tasks})
	if err != nil {
		return nil, err
	}
	return plans[0], nil
}

// runGPUPlanner is runTunablePlanner (scheduler/scheduler.go:35-52) with lines 43-44 replaced; PrioritizeTasks consults it
// when the distro's PlannerSettings.Version asks for the MI355X planner.
func runGPUPlanner(ctx context.Context, d *distro.Distro, tasks []task.Task, opts TaskPlannerOptions) ([]task.Task, error) {
	tasks, err := PopulateCaches(ctx, opts.ID, tasks)
	if err != nil {
		return nil, errors.WithStack(err)
	}
	plans, infos, err := planBatch(ctx, []*distro.Distro{d}, [][]task.Task{tasks})
	if err != nil {
		return nil, errors.WithStack(err)
	}
	info := infos[0]
	info.SecondaryQueue = opts.IsSecondaryQueue // scheduler.go:45
	info.PlanCreatedAt = opts.StartedAt         // scheduler.go:46
	if err = PersistTaskQueue(ctx, d.Id, plans[0], info, opts.MaxScheduledTasksPerDistro); err != nil {
		return nil, errors.WithStack(err)
	}
	return plans[0], nil
}
