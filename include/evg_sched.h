/*
 * evg_sched.h -- C ABI of the MI355X-native Evergreen per-distro scheduling hot path.
 *
 * This is the drop-in boundary: a cgo-callable, plain-C surface (pointers
 * and sizes only, no callbacks, no retained caller memory) that replaces, for MANY distros in one
 * batched call, the bodies behind the reference's two Go function-value types
 *
 *     type TaskPlanner   func(*distro.Distro, []task.Task, TaskPlannerOptions) ([]task.Task, error)
 *                                                      /root/reference/scheduler/scheduler.go:26
 *     type HostAllocator func(context.Context, *HostAllocatorData) (int, int, error)
 *                                                      /root/reference/scheduler/host_allocator.go:15
 *
 * Every entry point below cites the reference code it replaces. The Go-side binding a maintainer
 * would add (cgo stub) is in INTEGRATION.md.
 *
 * Layout contract (checked by evg_validate_plan_input; the host shim guarantees it):
 *   - Tasks are struct-of-arrays rows grouped by distro: rows [task_off[d], task_off[d+1]) belong to
 *     distro d. "Input index" of a task == its row number; the canonical tie-break uses it.
 *   - Strings never cross the ABI. Task ids, task-group strings (Task.GetTaskGroupString(),
 *     model/task/task.go:436-438) and version ids are interned by the caller into dense int32 keys:
 *       tg_key      : -1 when Task.TaskGroup == "", else in [tg_off[d], tg_off[d+1])
 *       version_key : in [ver_off[d], ver_off[d+1])
 *     One string maps to one key per distro, one key to one string; which key, and whether every key of the
 *     range has a task, is the caller's business (ABI 3.1; until 3.0 keys had to be dense, in order of first
 *     appearance -- what a shim that interns while it packs produces anyway). A key without a task costs one
 *     idle unit slot and a group_info row with present == 0: what a resident pool keeps when
 *     evg_pool_apply_delta removes the last task of a group.
 *   - A dependency edge stores the ROW of the dependency when that task is in the SAME distro's
 *     segment (the planner's cache.Exists(dep.TaskId), scheduler/planner.go:453, and
 *     GetDistroQueueInfo's depCache, scheduler/scheduler.go:62-65), else -1 plus the resolved state
 *     of the out-of-queue task (what Task.DependenciesMet fetches from the DB, task.go:649-688).
 *   - All times are int64 Unix nanoseconds (0 is the Unix epoch == utility.ZeroTime). Go's zero
 *     time.Time (year 1, Time.IsZero()) is not representable in Unix ns and is encoded as
 *     EVG_TIME_GO_ZERO (INT64_MIN). Durations are computed like Go's Time.Sub: saturating at
 *     +/- (2^63-1), so time.Since(<Go zero>) == MaxInt64 exactly as in the reference. `now_ns`
 *     replaces every time.Since()/time.Now() on the path.
 */
#ifndef EVG_SCHED_H
#define EVG_SCHED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Go's zero time.Time (Time.IsZero()); sorts before every real time, like in Go. */
#define EVG_TIME_GO_ZERO INT64_MIN

/* ---- return codes of the entry points ------------------------------------------------------ */
#define EVG_OK 0
#define EVG_E_INVALID (-1)  /* NULL / negative size / inconsistent offsets                        */
#define EVG_E_HIP (-2)      /* HIP runtime or launch failure (message via evg_last_error)         */
#define EVG_E_NOMEM (-3)    /* device or host allocation failed                                  */
#define EVG_E_CONTRACT (-4) /* input violates the layout contract above                          */
#define EVG_E_NODEVICE (-5) /* no gfx950 device / HIP runtime unavailable: there is NO CPU fallback */
#define EVG_E_TIMEOUT (-6)  /* a device wait outlived the object's deadline (ABI 3.3): the object is poisoned and refuses
                               further work -- destroy it and create another; see evg_set_deadline_ms            */

/* ---- per-distro allocator status (out_status[d]); mirrors the reference's error returns ----- */
#define EVG_ALLOC_OK 0
/* "future host factor cannot be greater than 1"  utilization_based_host_allocator.go:287-289 */
#define EVG_ALLOC_E_FUTURE_FRACTION 1
/* "unable to plan hosts for distro %s due to pool size of %d"  ...allocator.go:185-187 */
#define EVG_ALLOC_E_POOL_SIZE 2

/* ---- task flag bits (evg_task_soa.flags) --------------------------------------------------- */
#define EVG_TF_REQ_MASK 0x0003u      /* requester class: 0 other (mainline/trigger/ad hoc),
                                        1 patch  (patch_request | github_pull_request),
                                        2 merge queue (github_merge_request)
                                        globals.go:798-804,1224-1242; planner.go:308-312        */
#define EVG_TF_REQ_PATCH 1u
#define EVG_TF_REQ_MERGE 2u
#define EVG_TF_GENERATE 0x0004u      /* Task.GenerateTask                    planner.go:315      */
#define EVG_TF_STEPBACK 0x0008u      /* Task.ActivatedBy == "stepback"       planner.go:316      */
#define EVG_TF_OVERRIDE_DEPS 0x0010u /* Task.OverrideDependencies            task.go:3406-3408   */
#define EVG_TF_OTHER_DISTRO 0x0020u  /* Task.DistroId != d.Id                scheduler.go:89     */
#define EVG_TF_S3_STORAGE 0x0040u    /* CachedProjectStorageMethod == "s3"   scheduler.go:120    */
#define EVG_TF_BLOCKED 0x0080u       /* Task.Blocked(): an Unattainable dep and !Override
                                        task.go:3688-3699 (read when this row is someone's dep) */
#define EVG_TF_STATUS_SHIFT 8        /* 2 bits, this row's Task.Status as seen by a dependent:
                                        0 anything else (undispatched, started, ...),
                                        1 "success", 2 "failed"              task.go:546-561     */
#define EVG_TF_STATUS_MASK 0x0300u

/* ---- dependency edge byte (evg_task_soa.dep_info) ------------------------------------------ */
#define EVG_DEP_REQ_MASK 0x03u     /* required status of the edge (task.go:550-557):
                                      0 "" or "success", 1 "failed", 2 "*" (AllStatuses),
                                      3 unrecognised string => never satisfied                  */
#define EVG_DEP_STATE_SHIFT 2      /* for dep_idx == -1 only: status class of the fetched task,  */
#define EVG_DEP_STATE_MASK 0x0Cu   /*   0 other, 1 "success", 2 "failed"                         */
#define EVG_DEP_BLOCKED 0x10u      /* for dep_idx == -1 only: fetched task .Blocked()            */
#define EVG_DEP_MISSING 0x20u      /* for dep_idx == -1 only: not in DB => DependenciesMet errors
                                      => unmet (scheduler.go:180-186)                           */

/* Number of int64 fields in a task.SortingValueBreakdown row (model/task/task.go:4060-4108). */
#define EVG_BREAKDOWN_FIELDS 13
enum evg_breakdown_field {
  EVG_BD_TASK_GROUP_LENGTH = 0,
  EVG_BD_TOTAL_VALUE = 1,
  EVG_BD_PRI_INITIAL = 2,        /* PriorityBreakdown.InitialPriorityImpact */
  EVG_BD_PRI_TASK_GROUP = 3,     /* PriorityBreakdown.TaskGroupImpact       */
  EVG_BD_PRI_GENERATOR = 4,      /* PriorityBreakdown.GeneratorTaskImpact   */
  EVG_BD_PRI_COMMIT_QUEUE = 5,   /* PriorityBreakdown.CommitQueueImpact     */
  EVG_BD_RANK_COMMIT_QUEUE = 6,  /* RankValueBreakdown.CommitQueueImpact    */
  EVG_BD_RANK_NUM_DEPENDENTS = 7,
  EVG_BD_RANK_EST_RUNTIME = 8,
  EVG_BD_RANK_MAINLINE_WAIT = 9,
  EVG_BD_RANK_STEPBACK = 10,
  EVG_BD_RANK_PATCH = 11,
  EVG_BD_RANK_PATCH_WAIT = 12
};

/* ---- inputs --------------------------------------------------------------------------------- */

/* The runnable-task pool, struct-of-arrays (SURVEY.md 8b' column map). All arrays have n_tasks
 * entries unless noted. For the *_device entry points every pointer is a device pointer. */
typedef struct evg_task_soa {
  int32_t n_tasks;  /* N */
  int32_t n_edges;  /* E = dep_off[N] */
  const int64_t* priority;              /* Task.Priority                planner.go:324-327,400 */
  const int64_t* expected_duration_ns;  /* resolved FetchExpectedDuration().Average
                                           planner.go:328,404; scheduler.go:87                */
  const int64_t* queue_ts_ns;           /* ActivatedTime if !IsZero() else IngestTime if
                                           !IsZero() else EVG_TIME_GO_ZERO (contributes 0)
                                                                         planner.go:318-322     */
  const int64_t* scheduled_ts_ns;       /* Task.ScheduledTime           scheduler.go:137       */
  const int64_t* deps_met_ts_ns;        /* Task.DependenciesMetTime; utility.IsZeroTime() <=>
                                           value is 0 or EVG_TIME_GO_ZERO
                                           task.go:3406-3408; scheduler.go:138-140             */
  const int32_t* num_dependents;        /* Task.NumDependents           planner.go:329-332,396 */
  const int32_t* task_group_order;      /* Task.TaskGroupOrder          planner.go:392         */
  const int32_t* task_group_max_hosts;  /* Task.TaskGroupMaxHosts       scheduler.go:105       */
  const int32_t* tg_key;                /* interned GetTaskGroupString(), -1 = no task group   */
  const int32_t* version_key;           /* interned Task.Version        planner.go:439,441     */
  const uint16_t* flags;                /* EVG_TF_*                                            */
  const int32_t* dep_off;               /* CSR, N+1 entries: edges of row i are
                                           [dep_off[i], dep_off[i+1])   Task.DependsOn          */
  const int32_t* dep_idx;               /* E entries: row of the dependency, or -1             */
  const uint8_t* dep_info;              /* E entries: EVG_DEP_*                                */
  const int64_t* dep_finished_ts_ns;    /* E entries or NULL (= all zero): Dependency.FinishedAt,
                                           read by setDependenciesMetTime  task.go:690-701     */
} evg_task_soa;

/* Resolved planner settings of one distro. RAW values: the library applies the reference's getters
 * (<= 0 -> 1, model/distro/distro.go:379-434) and target-time defaults (0 -> 30 min,
 * distro.go:448-475) itself. */
typedef struct evg_distro_params {
  int64_t patch_factor;
  int64_t patch_time_in_queue_factor;
  int64_t commit_queue_factor;
  int64_t mainline_time_in_queue_factor;
  int64_t expected_runtime_factor;
  int64_t generate_task_factor;
  int64_t stepback_task_factor;
  double num_dependents_factor;
  int64_t target_time_ns;             /* PlannerSettings.TargetTime; 0 => MaxDurationPerHost() */
  int64_t merge_queue_target_time_ns; /* PlannerSettings.MergeQueueTargetTime; <= 0 => unused  */
  int32_t group_versions;             /* PlannerSettings.ShouldGroupVersions()                 */
  int32_t includes_dependencies;      /* DispatcherSettings.Version ==
                                         "revised-with-dependencies"   scheduler.go:29          */
} evg_distro_params;

/* The batch: D distros over one task pool. */
typedef struct evg_plan_input {
  int32_t n_distros;                  /* D */
  int32_t n_task_groups;              /* tg_off[D]  */
  int32_t n_versions;                 /* ver_off[D] */
  int32_t max_distro_tasks;           /* max over d of task_off[d+1]-task_off[d], or 0 = unknown (a hint: it only
                                         selects how the generic path is launched, never what is computed)     */
  evg_task_soa tasks;
  const evg_distro_params* distros;   /* D rows */
  const int32_t* task_off;            /* D+1 */
  const int32_t* tg_off;              /* D+1 */
  const int32_t* ver_off;             /* D+1 */
  int64_t now_ns;
  int32_t promises;                   /* EVG_PROMISE_* (ABI 1.2), 0 = none. Unlike the hint above a promise is part of the
                                         contract: a false one leaves the plan uncomputed (and evg_take_device_status
                                         reports it). The host-pointer entry points work them out themselves and ignore
                                         this field; a *_device caller takes them from evg_plan_launch_hints on its host
                                         copy of the batch                                                             */
  int32_t n_big_tier_distros;         /* how many distros the 4096-task tier of the one-workgroup kernel takes (2049..4096
                                         tasks, or fewer with more unit slots / edges than the 2048-task tier holds), or
                                         0 = unknown / none. A hint like max_distro_tasks (ABI 3.0; from
                                         evg_plan_launch_hints): it sizes that tier's launch -- every one of its workgroups
                                         needs a whole CU -- and without it those distros take the large-distro pipeline,
                                         with the same result                                                           */
} evg_plan_input;

/* Every distro of the batch can be planned by the one-workgroup kernel: at most 2048 tasks, unit slots + dependency
 * edges inside its LDS budget, at most 1023 task groups, every |priority| below 2^31. The library then does not enqueue the
 * kernels that pick up what that kernel leaves (one empty launch, ~4.5 us of a ~70 us tick on BASELINE config 3). A false
 * promise is detected on the device and reported by evg_take_device_status: the plan of such a batch must not be used. */
#define EVG_PROMISE_ALL_ON_LDS_PATH 0x1
/* Every distro can be planned by one of the two tiers of the one-workgroup kernel (the above, or at most 4096 tasks with unit
 * slots + edges inside the whole CU's LDS; every |priority| below 2^31) and n_big_tier_distros counts the second kind (ABI 3.0).
 * Nothing is enqueued behind the two tiers then. A false promise is reported the same way. */
#define EVG_PROMISE_ALL_ON_LDS_TIERS 0x2
/* A HINT that travels in the same word (ABI 3.1; it can never change a plan): both the one-workgroup tiers and the large-distro
 * pipeline have substantial work in this batch (each at least an eighth of the tasks). The library then runs the pipeline's
 * launches BESIDE the tiers' on a second stream instead of behind them: 0.333 -> 0.300 ms per plan on the skewed variant of config
 * 3 (89 large distros, 423 small). Without such a mix the two event hand-overs only cost (+8 % on config 5's share, where the
 * tiers have nothing to do, and on a pool of small distros with one large one, where the chip is full). Set by
 * evg_plan_launch_hints. */
#define EVG_HINT_MIXED_POOL 0x100
/* Another hint in the same word (ABI 3.1): NO distro of the batch fits a one-workgroup tier (BASELINE config 5: every distro has
 * 19.5 k tasks). The tiers' launches -- workgroups that would all exit at once -- are then not enqueued and the large-distro
 * pipeline starts at once; a distro that does fit after all is planned by the generic kernel (slowly, correctly). */
#define EVG_HINT_NO_TIER_DISTROS 0x200

/* ---- outputs -------------------------------------------------------------------------------- */

/* model.TaskGroupInfo (model/task_queue.go:22-45) without the name: the row index is the name.
 * Row d (d < D) is distro d's standalone bucket (Name == ""); row D + k is task-group key k. */
typedef struct evg_group_info {
  int64_t expected_duration_ns;
  int64_t duration_over_threshold_ns;
  int32_t count;
  int32_t max_hosts;                /* first-seen-in-QUEUE-order task's TaskGroupMaxHosts
                                       scheduler.go:103-106                                     */
  int32_t count_duration_over_threshold;
  int32_t count_wait_over_threshold;
  int32_t count_dep_filled_merge_queue_tasks;
  int32_t present;                  /* 1 iff the reference's TaskGroupInfos would hold this row */
  int32_t count_free;               /* written by evg_allocate_hosts  ...allocator.go:106-109   */
  int32_t count_required;           /* written by evg_allocate_hosts                            */
} evg_group_info;

/* model.DistroQueueInfo (model/task_queue.go:47-78). secondary_queue is the COMPUTED value
 * (scheduler.go:89-91); runTunablePlanner then overwrites it with opts.IsSecondaryQueue and sets
 * PlanCreatedAt (scheduler.go:45-46) -- that stays in the Go shim. */
typedef struct evg_distro_info {
  int64_t expected_duration_ns;
  int64_t max_duration_threshold_ns;
  int64_t duration_over_threshold_ns;
  int32_t length;
  int32_t length_with_dependencies_met;
  int32_t count_dep_filled_merge_queue_tasks;
  int32_t count_duration_over_threshold;
  int32_t count_wait_over_threshold;
  int32_t num_queued_large_parser_project_tasks;
  int32_t secondary_queue;
  int32_t n_task_group_infos;       /* len(TaskGroupInfos) */
} evg_distro_info;

typedef struct evg_plan_output {
  int32_t* order;          /* N: order[task_off[d] + p] = row of the task at queue position p of
                              distro d  (TaskPlan.Export, planner.go:462-481)                    */
  int64_t* breakdown;      /* N x 13 by ROW (task.SortingValueBreakdown stamped at planner.go:475),
                              or NULL to skip                                                    */
  uint8_t* deps_met;       /* N by row: checkDependenciesMet result (scheduler.go:70-76); this is
                              also Task.HasDependenciesMet() after the call (persister :47)      */
  int64_t* wait_ns;        /* N by row: Task.WaitSinceDependenciesMet (scheduler.go:141), 0 when
                              the reference leaves it untouched. evg_pool_plan / evg_pool_tick take
                              NULL here (not downloaded: it is 8 of a resident tick's 14.7 bytes per
                              task over the link, and nothing in the reference reads the field
                              back -- the queue-info rows carry what it decides)                  */
  evg_distro_info* distro_info; /* D */
  evg_group_info* group_info;   /* D + n_task_groups */
  int32_t* n_units;        /* D: TaskPlan.Len() after UnitCache.Export dedup (planner.go:73-89),
                              or NULL to skip                                                    */
  /* The breakdown per UNIT (ABI 1.2). TaskPlan.Export stamps the emitting unit's value on every task of the unit
   * (planner.go:475; task.go:4118-4153), so N x 13 rows by task repeat each unit's row once per member; these two
   * outputs hold every distinct row once, and the planner kernels never write rows by task (when `breakdown` is
   * requested the library expands these, on the device). Both or neither; NULL to skip.
   * Unit slots: distro d owns slots [U(d), U(d+1)), U(d) = task_off[d] + tg_off[d] + ver_off[d] -- at most one slot per
   * task, task group and version of the distro, N + n_task_groups + n_versions in all. Rows of slots that emit no
   * task are unspecified. */
  int32_t* unit_of_task;   /* N by ROW: the slot of the unit the task is emitted from              */
  int64_t* unit_breakdown; /* 13 x n_slots, FIELD-MAJOR: field f (enum evg_breakdown_field) of slot u is
                              unit_breakdown[f * n_slots + u], n_slots = N + n_task_groups + n_versions (the kernels'
                              stores of one field are consecutive words; row-major 104-byte records cost 64 cache-line
                              requests per store instruction)                                       */
} evg_plan_output;

/* ---- allocator inputs ----------------------------------------------------------------------- */

/* distro.HostAllocatorSettings + the few Distro fields the allocator reads
 * (utilization_based_host_allocator.go:29,39,51,95,142,162,167). */
typedef struct evg_alloc_params {
  double future_host_fraction;
  int32_t minimum_hosts;
  int32_t maximum_hosts;
  int32_t provider;   /* 0 not ephemeral (static, ...), 1 ephemeral (ec2-fleet, mock),
                         2 docker (ephemeral, exempt from the max-hosts early-out :39)
                         globals.go:767-774 */
  int32_t disabled;   /* Distro.Disabled                                                :51 */
  int32_t round_up;   /* RoundingRule == "round-up"                                     :162 */
  int32_t feedback_waits_over_thresh; /* FeedbackRule == "waits-over-thresh-feedback"   :167 */
} evg_alloc_params;

#define EVG_HF_FREE 0x01u         /* Host.IsFree(): RunningTask == "" && !IsTearingDown()
                                     model/host/host.go:215-222                                 */
#define EVG_HF_RUNNING 0x02u      /* RunningTask != ""  ...allocator.go:313                     */
#define EVG_HF_RUNNING_FOUND 0x04u/* the running task was found by task.Find(ByIds) (:322); only
                                     then do start/exp/stddev contribute a fraction             */

typedef struct evg_host_soa {
  int32_t n_hosts;
  int32_t reserved;
  const uint8_t* flags;            /* EVG_HF_* */
  const int32_t* tg_key;           /* bucket of groupByTaskGroup (:208-224): -1 = "" ; >= 0 = the
                                      interned host.GetTaskGroupString() when RunningTask != "" &&
                                      RunningTaskGroup != "" and that string is a task-group key of
                                      this distro's queue; -2 = a group string not in the queue   */
  const int64_t* start_ts_ns;      /* running task StartTime                                :345 */
  const int64_t* expected_duration_ns; /* running task FetchExpectedDuration().Average      :343 */
  const int64_t* duration_stddev_ns;   /* running task FetchExpectedDuration().StdDev       :344 */
} evg_host_soa;

typedef struct evg_alloc_input {
  int32_t n_distros;
  int32_t n_task_groups;
  const evg_alloc_params* params;       /* D */
  const int32_t* host_off;              /* D+1 */
  const int32_t* tg_off;                /* D+1 (same as the plan's) */
  evg_host_soa hosts;
  const evg_distro_info* distro_info;   /* D, from evg_plan_distros (what the allocator job reads
                                           back from Mongo, units/host_allocator.go:144)         */
  evg_group_info* group_info;           /* D + n_task_groups, in/out: count_free/count_required  */
  int64_t now_ns;
  /* adjustForLargeParserProjectLimit (units/host_allocator.go:150,479-520; ABI 3.0): the allocator JOB lowers
   * DistroQueueInfo.LengthWithDependenciesMet by the queued large-parser-project (S3 storage) tasks the global limit blocks,
   * between reading the planner's queue info and calling the HostAllocator -- and the allocator clamps on exactly that field
   * (utilization_based_host_allocator.go:113-115). A batched tick hands the planner's device-resident info rows straight to
   * the allocator, so the adjustment is applied here, per distro, to the value the clamp uses (distro_info is not modified):
   *     limit <= 0 or num_queued_large_parser_project_tasks == 0      -> no change            (:481-488)
   *     blocked = num_queued - max(0, limit - running);  blocked > 0  -> length_with_dependencies_met - blocked   (:500-507)
   * A caller whose DistroQueueInfo was already adjusted by the Go job (the per-distro shim) passes limit 0. */
  int32_t max_concurrent_large_parser_project_tasks; /* model.GetMaxConcurrentLargeParserProjTasks(config); <= 0 = no limit */
  int32_t running_large_parser_project_tasks;        /* task.CountLargeParserProjectTasks(ctx)                              */
} evg_alloc_input;

typedef struct evg_alloc_output {
  int32_t* new_hosts;   /* D: newHostsNeeded      */
  int32_t* free_hosts;  /* D: estimatedFreeHosts  */
  int32_t* status;      /* D: EVG_ALLOC_*         */
} evg_alloc_output;

/* ---- entry points --------------------------------------------------------------------------- */

typedef struct evg_ctx evg_ctx;

/* Creates a planner context bound to HIP device `device_ordinal`. Returns NULL (and sets a message
 * retrievable with evg_last_error(NULL)) when no gfx950 device is usable: there is deliberately no
 * CPU fallback. One ctx per goroutine/OS thread, or serialise calls externally; ctxs are independent. One ctx per batch in
 * flight: see the stream-ordering rule at the *_device prototypes. */
evg_ctx* evg_create(int device_ordinal);
void evg_destroy(evg_ctx* ctx);

/* Page-locked host memory for the buffers of the host-pointer entry points (ABI 1.2). Those entry points accept any host
 * memory; from pageable memory every column is bounced through the driver's staging buffers (about 24 GB/s on an MI355X
 * box), from memory allocated here the copies are plain DMA at the link's rate (about 55 GB/s), and the 1M-task tick goes
 * from 4.1 ms to under 2 ms. A shim allocates its column and output buffers here once and re-uses them every tick (cgo:
 * unsafe.Slice over the returned pointer). NULL on failure (message via evg_last_error).
 * One block for a tick's small arrays (late round 6): the calls that pack their inputs into ONE staging block -- evg_pool_tick,
 * evg_pool_apply_delta, evg_pool_update, and the host-pointer calls on small batches -- do not re-pack an array they find inside
 * a block of >= 1 MiB from here at an offset that is a multiple of 256: the block is mirrored on the device and the stretch of
 * it the call names goes up in one copy per flush. A shim that builds a tick's delta and updates in such a block (sub-allocating
 * its ~30 arrays at 256-byte offsets) saves the packing: 0.13 ms of a 5 % tick of a million tasks. */
void* evg_host_alloc(evg_ctx* ctx, size_t bytes);
void evg_host_free(evg_ctx* ctx, void* p);

/* Measurement hook (ABI 1.2): when enabled, every plan call on `ctx` attaches a start and a stop HIP event to the dispatch of the
 * planner kernel (k_plan_distros; hipExtLaunchKernelGGL on the stream it is launched on -- the kernel's own begin and end
 * timestamps, what rocprofv3's kernel trace reports; until round 5 the events were recorded on the stream before and after
 * the launch and read ~3 us more than the kernel); evg_last_plan_kernel_ms waits for the last call's stop event and returns
 * the interval -- that kernel alone, without the large-distro kernels enqueued behind it. bench.py's roofline block is
 * measured with it. */
int evg_profile_plan_kernel(evg_ctx* ctx, int enable);
int evg_last_plan_kernel_ms(evg_ctx* ctx, float* ms);

/* Last error message of `ctx` (or of the failed evg_create when ctx == NULL). */
const char* evg_last_error(const evg_ctx* ctx);

/* Library/ABI version: (major << 16) | minor. MAJOR changes whenever a struct of this header changes size or layout, or an
 * entry point changes or goes (2.0: evg_plan_input / evg_plan_output as they have been since 1.2; 3.0: evg_alloc_input grew the
 * two large-parser-project fields, evg_plan_input.reserved0 became n_big_tier_distros, evg_plan_launch_hints returns that count,
 * and the one-launch evg_plan_allocate[_range]_device entry points are gone -- measured no faster than the two calls for three
 * rounds); MINOR adds entry points only. */
#define EVG_ABI_MAJOR 3
#define EVG_ABI_MINOR 3 /* 3.1: the evg_multi_* entry points (several devices from one process) and evg_balanced_ranges;
                           3.2: the evg_batcher_* entry points (micro-batching front for per-distro callers);
                           3.3: bounded device waits (EVG_E_TIMEOUT, evg_set_deadline_ms, evg_multi_set_deadline_ms,
                                evg_batcher_set_deadline_ms), evg_batcher_schedule / _plan_queue / _close (pair requests, resident
                                queues), evg_pool_tick (the fused resident tick) */
int32_t evg_abi_version(void);
/* What a binding calls once at start-up with ITS compile-time view of the header: EVG_OK iff the library's major equals
 * `major`, its minor is at least `minor`, and the four struct sizes are the library's. A binding must refuse the library
 * otherwise (the library reads every field of the structs it is handed). */
int evg_check_abi(int32_t major, int32_t minor, size_t sizeof_plan_input, size_t sizeof_plan_output, size_t sizeof_alloc_input,
                  size_t sizeof_group_info);

/* Sticky device-side status of `ctx` (ABI 2.0). The *_device entry points only enqueue, so a contract violation that only
 * the kernels can see -- today: a batch passed with EVG_PROMISE_ALL_ON_LDS_PATH that holds a distro the one-workgroup kernel
 * cannot plan (its plan was NOT computed) -- is recorded by the kernel in a host-visible word. This call returns
 * EVG_E_CONTRACT (message via evg_last_error) if a batch enqueued on `ctx` before the caller's last stream synchronisation
 * recorded one, and clears it; EVG_OK otherwise. Every later entry point on the context fails with EVG_E_CONTRACT too until
 * the status is taken. A caller that passes promises calls this after it synchronised the stream, before it uses the plan. */
int evg_take_device_status(evg_ctx* ctx);

/* Bounded calls (ABI 3.3). A cgo call pins its OS thread until it returns, and the reference bounds its own jobs (the distro
 * scheduler job: 5 min, units/scheduler.go:18; the host allocator job: 10 min, units/host_allocator.go:32). Every wait of the
 * library on the device -- each synchronous entry point ends in one -- polls the stream against a monotonic clock: when the
 * device has not finished after `ms` milliseconds the call returns EVG_E_TIMEOUT, the context is POISONED (its buffers may still
 * be in use by whatever hangs) and every later entry point on it fails with EVG_E_TIMEOUT at once; evg_destroy then waits once
 * more and, if the device still has not come back, leaks the context's device memory instead of blocking. A caller's job fails
 * and runs again on the next 15 s tick on a fresh context instead of holding a thread for ever.
 * Default: 30,000 ms (EVG_DEADLINE_MS in the environment overrides it at evg_create); 0 = wait without a limit. The *_device
 * entry points only enqueue: their caller owns the wait. */
int evg_set_deadline_ms(evg_ctx* ctx, int64_t ms);
int64_t evg_get_deadline_ms(const evg_ctx* ctx);
/* Test hook for the deadline: enqueues, on the context's own stream, a kernel that spins for `ms` milliseconds of device wall
 * clock (at most 20,000) -- the next synchronous call on the context finds the device busy for that long. */
int evg_debug_stall(evg_ctx* ctx, int32_t ms);

/* No exception leaves the library through this boundary (late round 6): a C++ exception unwinding into cgo or ctypes ends the process,
 * where the reference's jobs fail and are retried (units/scheduler.go:18, units/host_allocator.go:32). Every int-returning entry point
 * of the single-context and multi-device families catches what its body throws: host memory that ran out (std::bad_alloc,
 * std::length_error) is EVG_E_NOMEM, anything else EVG_E_HIP, the text in evg_last_error / evg_multi_last_error; evg_create and
 * evg_multi_create return NULL. Locks and stream guards have unwound by then: the context takes the next call. Threads the library
 * starts for host-side checks run code that cannot throw, and a thread that cannot be had costs parallelism, not the call.
 * The evg_batcher_* request entry points: nothing throws between a request's join and its batch's results (a slot's member list is
 * reserved up front; an exception inside the leader's run of the batch becomes the batch's code, every member gets it, the slot goes
 * on), and what is in front of the join is caught at the entry point (the code + text in `err`).
 * evg_debug_throw is the test hook: throws inside such an entry point while holding the context's mutex -- kind 0 std::bad_alloc,
 * 1 std::runtime_error, 2 a non-std object, 3 a vector grown past max_size; `ctx` may be NULL (no GPU is touched). */
int evg_debug_throw(evg_ctx* ctx, int32_t kind);

/* Host-side check of the layout contract; no GPU work. */
int evg_validate_plan_input(const evg_plan_input* in, char* msg, int32_t msg_len);

/* Host side, no GPU work: the launch hints and the promises that hold for a batch whose pointers are HOST pointers -- what a
 * *_device caller copies into the evg_plan_input it passes with device pointers (max_distro_tasks, promises,
 * n_big_tier_distros). ABI 3.0. */
int evg_plan_launch_hints(const evg_plan_input* in, int32_t* max_distro_tasks, int32_t* promises, int32_t* n_big_tier_distros);

/* Device self-test of the planner's scoring arithmetic: runs unitInfo.value() (planner.go:209-300) over `n_cases`
 * generated inputs twice -- the kernels' fast exact form and a statement that follows the Go code step by step (IEEE fp64
 * divisions, int64 divisions) -- and counts the cases where any of the 13 SortingValueBreakdown fields differs. Inputs
 * sit on and around the quotient boundaries the fast form's exactness argument depends on. Synchronous. ABI 1.2. */
int evg_selftest_unit_value(evg_ctx* ctx, uint64_t seed, uint64_t n_cases, uint64_t* mismatches, uint64_t* first_bad_case);

/* Plans all D distros: replaces, per distro,
 *   PrepareTasksForPlanning(ctx, d, tasks).Export(ctx)   scheduler/scheduler.go:43
 *   GetDistroQueueInfo(ctx, d, plan, opts)               scheduler/scheduler.go:44,57-178
 * Host pointers in, host pointers out; synchronous; nothing is retained. */
int evg_plan_distros(evg_ctx* ctx, const evg_plan_input* in, const evg_plan_output* out);

/* Same, but every pointer INSIDE in/out (not the structs themselves) is a device pointer on ctx's
 * device and the work is enqueued on `hip_stream` (a hipStream_t; NULL = default stream) without a
 * host sync. The small per-distro tables (distros, task_off, tg_off, ver_off) are device-resident
 * too. This is what the Go shim uses when the pool is kept resident between 15 s ticks. */
int evg_plan_distros_device(evg_ctx* ctx, const evg_plan_input* in, const evg_plan_output* out,
                            void* hip_stream);
/* Rule for EVERY *_device entry point: a context owns one set of device scratch (generic-path flags, sort keys, tile
 * lists, allocator bucket words), so the calls made on one context must be stream-ordered with respect to each other --
 * one stream per context, or events between streams. Batches that are in flight concurrently need one context each
 * (contexts are cheap: scratch is allocated on first use). `_device` callers must run evg_validate_plan_input on their
 * host copy of the batch: the device paths cannot validate. */

/* Multi-GPU form (one amboy job per distro, units/crons.go:303-332 => shard by distro): plans only distros
 * [d_begin, d_end) of the batch that `in` describes. The WHOLE batch is resident on this device (north_star: one RCCL
 * broadcast of the shared pool); every output keeps the full batch's numbering -- order[task_off[d] + p], info row d,
 * group row D + key -- so a rank's results are contiguous slices of the full-size output arrays and the gather to rank 0
 * is a set of slice copies. Rows outside the range are not touched. ABI 1.1. */
int evg_plan_distro_range_device(evg_ctx* ctx, const evg_plan_input* in, const evg_plan_output* out, int32_t d_begin,
                                 int32_t d_end, void* hip_stream);

/* Replaces UtilizationBasedHostAllocator (scheduler/utilization_based_host_allocator.go:26-129)
 * for all D distros. Per-distro failures the reference reports as `error` come back in
 * out->status[d] with new_hosts[d] == 0 and free_hosts[d] == #free hosts (:99-101). */
int evg_allocate_hosts(evg_ctx* ctx, const evg_alloc_input* in, const evg_alloc_output* out);
int evg_allocate_hosts_device(evg_ctx* ctx, const evg_alloc_input* in, const evg_alloc_output* out,
                              void* hip_stream);
/* The allocator for distros [d_begin, d_end) of the batch only (see evg_plan_distro_range_device). ABI 1.1. */
int evg_allocate_host_range_device(evg_ctx* ctx, const evg_alloc_input* in, const evg_alloc_output* out,
                                   int32_t d_begin, int32_t d_end, void* hip_stream);

/* capTaskQueueLength (scheduler/task_queue_persister.go:66-83) for all D distros: cut[d] = number of
 * leading queue positions of distro d to persist for limit max_scheduled (<= 0 disables). The
 * straddling test uses Task.TaskGroup (NOT the 4-part group string): tg_name_key is an interning of
 * the bare TaskGroup name, -1 for "". Device pointers; enqueued on hip_stream. */
int evg_cap_queue_device(evg_ctx* ctx, int32_t n_distros, const int32_t* task_off,
                         const int32_t* order, const int32_t* tg_name_key, int32_t max_scheduled,
                         int32_t* cut, void* hip_stream);

/* ---- PersistTaskQueue's queue materialisation -------------------------------------------------
 * scheduler/task_queue_persister.go:17-62 + model/task_queue.go:181-205,269-275: cap the plan with capTaskQueueLength
 * (task-group-aware cut), truncate to the 10,000-item limit of TaskQueue.Save, and build one model.TaskQueueItem per
 * persisted queue position. Strings never cross the ABI, so an item carries the ROW of its task: the shim copies the
 * pass-through fields -- Id / DisplayName / BuildVariant / RevisionOrderNumber / Revision / Project / Requester / Version /
 * ActivatedBy / Group and the Dependencies id list (task_queue_persister.go:30-45) -- from its own task slice by that row
 * (pinned by TestDBTaskQueuePersister, task_queue_persister_test.go:20-213, tests/golden_runner.py:check_persister);
 * the arrays below are the numeric fields the path computes or re-orders. Struct-of-arrays output,
 * items of distro d at [item_off[d], item_off[d+1]) in queue order; each array must hold n_tasks entries. */
#define EVG_TASK_QUEUE_SAVE_LIMIT 10000 /* model/task_queue.go:270-272 */
typedef struct evg_queue_items {
  int32_t* cut;                 /* D:   capTaskQueueLength result (tasks marked scheduled, persister :57)   */
  int32_t* item_off;            /* D+1: persisted items per distro, prefix sums: min(cut, 10000)            */
  int32_t* row;                 /* task row of the item (-> Id and the other strings)                       */
  int64_t* expected_duration_ns;/* TaskQueueItem.ExpectedDuration (Task.ExpectedDuration set at scheduler.go:125) */
  int64_t* priority;            /* TaskQueueItem.Priority                                                   */
  int32_t* group_max_hosts;     /* TaskQueueItem.GroupMaxHosts                                              */
  int32_t* group_index;         /* TaskQueueItem.GroupIndex  (Task.TaskGroupOrder)                          */
  int32_t* n_dependencies;      /* len(TaskQueueItem.Dependencies) == len(Task.DependsOn)                   */
  uint8_t* dependencies_met;    /* TaskQueueItem.DependenciesMet == Task.HasDependenciesMet() after planning */
  int64_t* breakdown;           /* items x 13: TaskQueueItem.SortingValueBreakdown, or NULL to skip
                                   (requires the plan's breakdown output)                                   */
} evg_queue_items;

/* `in` / `plan` are the inputs and outputs of a finished evg_plan_distros_device call on the same stream; tg_name_key
 * is the interning of the bare Task.TaskGroup name (-1 for ""), as for evg_cap_queue_device. max_scheduled <= 0
 * disables the cap (the 10,000 limit still applies). Device pointers; enqueued on hip_stream. */
int evg_materialize_queue_device(evg_ctx* ctx, const evg_plan_input* in, const evg_plan_output* plan,
                                 const int32_t* tg_name_key, int32_t max_scheduled, const evg_queue_items* items,
                                 void* hip_stream);

/* ---- the task finder's dependency filter -------------------------------------------------------
 * scheduler/task_finder.go:40-116 (LegacyFindRunnableTasks): of the undispatched tasks of each distro keep, in input
 * order, those whose project may dispatch them (dispatchable[row], decided on the host: ProjectCanDispatchTask and
 * the distro's ValidProjects, :59-84) and -- unless the distro's dispatcher is "revised-with-dependencies"
 * (evg_distro_params.includes_dependencies) -- whose dependencies are met (Task.DependenciesMet, task.go:649-688, with
 * getDependencyTaskCache's statuses in flags / dep_info; a dependency missing from the DB is an error => skipped, :86-99).
 * The Mongo queries (task.FindHostSchedulable, the project-ref cache) stay on the host. `in` is the candidate pool in
 * the planner's own layout. Outputs: deps_met[N] (what DependenciesMet returned; 1 when the check is not performed),
 * keep[N], runnable_count[D] and runnable_row[N] = the kept rows of distro d, in input order, at
 * [task_off[d], task_off[d] + runnable_count[d]). Device pointers; enqueued on hip_stream. */
int evg_filter_runnable_device(evg_ctx* ctx, const evg_plan_input* in, const uint8_t* dispatchable, uint8_t* deps_met,
                               uint8_t* keep, int32_t* runnable_row, int32_t* runnable_count, void* hip_stream);

/* ---- the DAG dispatcher's rebuild --------------------------------------------------------------
 * model/task_queue_service_dependency.go:153-250 (basicCachedDAGDispatcherImpl.rebuild): one node per persisted
 * TaskQueueItem (queueIndex = position in the distro's queue, :161-164), one line dependency -> dependent for every
 * entry of item.Dependencies that is itself in the queue (addEdge :119-150; others are skipped), then
 * topo.SortStabilized(graph, order by queueIndex) (:205-217, gonum.org/v1/gonum v0.17.0 graph/topo: Tarjan's SCC
 * search over the nodes in DESCENDING queueIndex, successors in descending queueIndex, the components in reverse
 * order of completion; a component of more than one node -- a dependency cycle -- leaves ONE nil entry), and the
 * schedulableUnit of every task group: its items in queue order, stable-sorted by GroupIndex (:166-195).
 *
 * Inputs: `in` = the planner's batch (dependency CSR, tg_key = the composite group id, task_group_order = GroupIndex)
 * and the persisted queues evg_materialize_queue_device produced (item_off[D+1], item_row[]). Outputs, all indexed
 * like the items (distro d at item_off[d]):
 *   sorted[item_off[d] + k], k < n_sorted[d] : d.sorted -- the queue index of the k-th node, -1 for a nil entry
 *   n_cycles[d]                              : len(topo.Unorderable)
 *   group_items[group_start[g] + k], k < group_count[g] : d.taskGroups[g].tasks as queue indexes, g = tg_key
 * Device pointers; enqueued on hip_stream. */
typedef struct evg_dispatch_order {
  int32_t* sorted;      /* n_tasks */
  int32_t* n_sorted;    /* D */
  int32_t* n_cycles;    /* D */
  int32_t* group_items; /* n_tasks */
  int32_t* group_start; /* n_task_groups: absolute index into group_items */
  int32_t* group_count; /* n_task_groups */
} evg_dispatch_order;

int evg_dispatch_order_device(evg_ctx* ctx, const evg_plan_input* in, const int32_t* item_off, const int32_t* item_row,
                              const evg_dispatch_order* out, void* hip_stream);

/* Host-pointer form for basicCachedDAGDispatcherImpl.rebuild(items) itself -- the dispatcher rebuilds from the persisted
 * TaskQueue document (Refresh, task_queue_service_dependency.go:74-93), not inside the scheduler job, so it takes the
 * items alone: items of distro d at [item_off[d], item_off[d+1]) in queue order; item i's Dependencies as the absolute
 * item indexes dep_idx[dep_off[i] .. dep_off[i+1]) (-1 or an item of another distro = not in this queue: no edge);
 * group_key[i] = -1 for Group == "" else a dense interning of compositeGroupID(Group, BuildVariant, Project, Version)
 * (:695-697) in [tg_off[d], tg_off[d+1]); group_index[i] = GroupIndex. Outputs as evg_dispatch_order_device. Synchronous. */
int evg_rebuild_dispatchers(evg_ctx* ctx, int32_t n_distros, const int32_t* item_off, const int32_t* dep_off, const int32_t* dep_idx,
                       const int32_t* group_key, const int32_t* tg_off, const int32_t* group_index, const evg_dispatch_order* out);

/* ---- the scheduler job end to end, host pointers ---------------------------------------------------------------
 * What a cgo shim calls once per tick for all D distros: evg_plan_distros, then -- from the plan that is still on the
 * device, so nothing is uploaded twice -- PersistTaskQueue's item lists (items != NULL; evg_materialize_queue_device)
 * and the DAG dispatcher's order for those queues (dispatch != NULL, requires items; evg_dispatch_order_device).
 * Every pointer is host memory; synchronous. Results are identical to the separate calls. */
int evg_schedule_distros(evg_ctx* ctx, const evg_plan_input* in, const evg_plan_output* out, const int32_t* tg_name_key,
                         int32_t max_scheduled, const evg_queue_items* items, const evg_dispatch_order* dispatch);

/* ---- a pool that stays on the device between ticks (ABI 2.0) -----------------------------------------------------------
 * The reference re-plans every distro every 15 s (units/crons_remote_fifteen_second.go:21) over a queue of which a few per
 * cent changed. evg_pool_load validates a batch, uploads it ONCE and keeps it on the device (owned by `ctx`: one pool per
 * context); evg_pool_update overwrites the listed rows' per-task VALUE columns and the listed dependency edges' state, so a
 * tick that changes 5 % of the pool moves 5 % of the bytes; evg_pool_plan plans the resident pool for a new `now_ns` and
 * downloads the outputs (exactly evg_plan_distros's, bit for bit, on the updated batch). The pool's STRUCTURE -- the row set,
 * the distro / key / CSR layout -- changes with evg_pool_apply_delta (below). Host pointers; synchronous; nothing of the caller's
 * is retained. */
typedef struct evg_row_update {
  int32_t n_rows;
  int32_t reserved;
  const int32_t* rows;                  /* n_rows row numbers, distinct */
  /* new values, n_rows entries each, in the order of `rows`; NULL = the column is unchanged */
  const int64_t* priority;
  const int64_t* expected_duration_ns;
  const int64_t* queue_ts_ns;
  const int64_t* scheduled_ts_ns;
  const int64_t* deps_met_ts_ns;
  const int32_t* num_dependents;
  const uint16_t* flags;                /* EVG_TF_*: status, blocked, override ... (the task-group / version keys are structure) */
} evg_row_update;
typedef struct evg_edge_update {
  int32_t n_edges;
  int32_t reserved;
  const int32_t* edges;                 /* n_edges edge numbers (positions in dep_idx), distinct */
  const uint8_t* dep_info;              /* EVG_DEP_* of the edge (a dependency outside the queue finished, became blocked ...), or NULL */
  const int64_t* dep_finished_ts_ns;    /* Dependency.FinishedAt, or NULL */
} evg_edge_update;
/* A tick's STRUCTURAL change (ABI 3.1). Every real 15 s tick removes the tasks that were dispatched, finished or deactivated and
 * adds the newly activated ones (units/crons_remote_fifteen_second.go:21,58-60); with evg_pool_update alone such a tick had to load
 * the whole pool again. evg_pool_apply_delta re-packs the resident pool ON THE DEVICE from the delta alone (kernels in
 * csrc/evg_pool_delta.hip.h; a few per cent of the pool's bytes cross the link): removed rows go, the kept rows of a distro keep their
 * relative order, the added rows of a distro follow them in the order given. The result is bit for bit the batch a caller would
 * have uploaded for the same rows in that order with the same keys -- evg_pool_plan plans it like any other.
 *   removed_rows            CURRENT row numbers, distinct
 *   removed_dep_state       per removed row: what the task now looks like to a dependent -- EVG_DEP_STATE_* | EVG_DEP_BLOCKED |
 *                           EVG_DEP_MISSING, what fetchedDepStates (shim/gpu_planner.go) reports for a dependency that is not in
 *                           the queue. Every edge that pointed at the row becomes an out-of-queue edge (dep_idx -1) with these
 *                           bits next to ITS OWN required-status bits, and removed_finished_ts_ns (or 0) as Dependency.FinishedAt
 *   added_distro, added     n_added rows: the distro of each (non-decreasing) and its columns; added.dep_off is the CSR over the
 *                           added rows; added.dep_idx: -1 = not in the queue (state in dep_info), j >= 0 = CURRENT row j of the
 *                           same distro (if j is being removed in this delta the edge takes j's removed state), -(k + 2) = added
 *                           row k of the same distro
 *   tg_off, ver_off         the NEW key ranges (D + 1 each), or NULL = unchanged: a distro's range may only GROW, at its end
 *                           (new keys for new groups / versions; an existing key k of distro d becomes k + tg_off[d] - old
 *                           tg_off[d]); the added rows' keys are in the new numbering. Keys whose last task left stay, unused
 *                           (the note on keys at the top: present == 0 rows); a full evg_pool_load compacts them away
 *   relinked_edges,         a KEPT row's edge (CURRENT edge number, distinct) whose dependency was not in the queue and now enters it:
 *   relinked_to             the edge points at added row relinked_to[i] (same distro) from now on, an in-queue edge with its own
 *                           required-status bits and nothing else (planner.go:451-455: the dependent joins that task's unit)
 * The caller keeps its own id -> row map current the same way: row numbers after the call = kept rows in order, then added rows,
 * distro by distro. Host pointers; synchronous; nothing of the caller's is retained. */
typedef struct evg_pool_delta {
  int32_t n_removed;
  int32_t n_added;
  const int32_t* removed_rows;
  const uint8_t* removed_dep_state;
  const int64_t* removed_finished_ts_ns; /* or NULL: all zero */
  const int32_t* added_distro;
  evg_task_soa added;                    /* n_tasks == n_added */
  const int32_t* tg_off;                 /* D + 1 or NULL */
  const int32_t* ver_off;                /* D + 1 or NULL */
  int32_t n_relinked;
  int32_t reserved;
  const int32_t* relinked_edges;
  const int32_t* relinked_to;
} evg_pool_delta;
int evg_pool_load(evg_ctx* ctx, const evg_plan_input* in);
int evg_pool_apply_delta(evg_ctx* ctx, const evg_pool_delta* delta);
int evg_pool_update(evg_ctx* ctx, const evg_row_update* rows, const evg_edge_update* edges); /* either may be NULL */
int evg_pool_plan(evg_ctx* ctx, int64_t now_ns, const evg_plan_output* out);
/* The fused resident tick (ABI 3.3): evg_pool_apply_delta(delta) + evg_pool_update(rows, edges) + evg_pool_plan(now_ns, out) as ONE
 * call behind ONE synchronisation -- the tick of the reference's 15 s cadence (units/crons_remote_fifteen_second.go:21,58-60) on a
 * resident pool. `delta`, `rows`, `edges` may each be NULL. The updates name rows / edges of the pool AFTER the delta. The delta and
 * the updates must fit one staging block (8 MB: a 5 % tick of a million tasks is 4.7 MB; EVG_E_INVALID otherwise -- use the three
 * calls). out->breakdown (rows by task) is not produced here: ask for unit_of_task + unit_breakdown. A delta or an update the
 * contract refuses leaves the pool as it was before the call and the outputs undefined. */
int evg_pool_tick(evg_ctx* ctx, const evg_pool_delta* delta, const evg_row_update* rows, const evg_edge_update* edges, int64_t now_ns,
                  const evg_plan_output* out);

/* Host-pointer forms of evg_filter_runnable_device and evg_allocator_report_device (stage in, run, stage out). */
int evg_filter_runnable(evg_ctx* ctx, const evg_plan_input* in, const uint8_t* dispatchable, uint8_t* deps_met, uint8_t* keep,
                        int32_t* runnable_row, int32_t* runnable_count);

/* ---- the host-allocator job's report math ------------------------------------------------------
 * units/host_allocator.go:250-334 (time-to-empty of the standalone queue on the hosts expected to be available,
 * with and without the hosts just spawned; its ratio to MaxDurationThreshold) and :393-424 (setTargetAndTerminate:
 * how many up hosts can be drawn down and the new capacity target). Closed forms per distro over the
 * DistroQueueInfo / TaskGroupInfo rows the planner and the allocator produced. float32 steps are IEEE single like Go's. */
typedef struct evg_report_params {
  int32_t n_up_hosts;        /* len(upHosts)                                              host_allocator.go:161 */
  int32_t minimum_hosts;     /* HostAllocatorSettings.MinimumHosts                                         :403 */
  int32_t drawdown_allowed;  /* HostsOverallocatedRule == terminate && provider spawnable && !hourly billing
                                                                                                   :327-333     */
  int32_t reserved;
} evg_report_params;

typedef struct evg_alloc_report {
  int64_t time_to_empty_ns;            /* timeToEmpty            :294-316 */
  int64_t time_to_empty_no_spawns_ns;  /* timeToEmptyNoSpawns             */
  float host_queue_ratio;              /* hostQueueRatio         :319     */
  float no_spawns_ratio;               /* noSpawnsRatio          :321     */
  int32_t hosts_avail;                 /* hostsAvail             :291     */
  int32_t drawdown;                    /* 1 iff setTargetAndTerminate enqueues a drawdown job (:327-333, :407) */
  int32_t new_cap_target;              /* DrawdownInfo.NewCapTarget (:399-404); 0 when drawdown == 0 */
  int32_t killable_hosts;              /* :396-398; 0 when the ratio test of :327 fails */
} evg_alloc_report;

/* hosts_spawned[d] = len(hostsSpawned) (normally new_hosts[d] of evg_allocate_hosts); free_hosts[d] = nHostsFree;
 * group_info carries count_free / count_required from the allocator. Device pointers; enqueued on hip_stream. */
int evg_allocator_report_device(evg_ctx* ctx, int32_t n_distros, const int32_t* tg_off, const evg_distro_info* distro_info,
                                const evg_group_info* group_info, const int32_t* hosts_spawned, const int32_t* free_hosts,
                                const evg_report_params* params, evg_alloc_report* report, void* hip_stream);
int evg_allocator_report(evg_ctx* ctx, int32_t n_distros, const int32_t* tg_off, const evg_distro_info* distro_info,
                         const evg_group_info* group_info, const int32_t* hosts_spawned, const int32_t* free_hosts,
                         const evg_report_params* params, evg_alloc_report* report);

/* ---- several MI355X from ONE process (ABI 3.1; SURVEY.md 8e, BASELINE configs 4 and 5) ----------------------------------------
 * north_star: "Distros shard naturally across the 8 GPUs of one node with a single RCCL broadcast of the shared runnable-task pool
 * over xGMI and a gather of the per-distro TaskQueue back to rank 0". The reference plans every distro from ONE scheduler process
 * (units/crons.go:303-332: one job per distro, all enqueued by the same process), so this is the shape a Go caller needs: one
 * evg_multi owns one context + stream per device and one RCCL communicator per device (ncclCommInitAll; RCCL is loaded on first
 * use, a single-GPU caller never touches it). Rank k = devices[k]; rank 0 holds the tick's pool and receives the results.
 *
 *   evg_multi_load     host pointers; validates the batch, packs it ONCE (page-locked) into one buffer -- a 256-byte header, then
 *                      every column at the next multiple of 256 bytes, the layout of evergreen_amd/multi.py -- uploads it to rank
 *                      0's device and cuts the contiguous distro ranges (evg_balanced_ranges). `alloc` may be NULL (plan only);
 *                      its distro_info / group_info pointers are ignored: each rank's allocator reads the rows its planner left
 *                      on the device
 *   evg_multi_tick     move-in: ONE ncclBroadcast of the packed pool from rank 0 -- or, created with EVG_MULTI_SCATTER, one group
 *                      of ncclSend / ncclRecv that hands every rank only the slices its range reads (SURVEY 8e's cheaper form);
 *                      every rank: evg_plan_distro_range_device + evg_allocate_host_range_device over its range (the reference's
 *                      two jobs, scheduler/wrapper.go:107 and units/host_allocator.go:183-188), outputs in the FULL batch's
 *                      numbering; gather: one group of ncclSend / ncclRecv, every result slice straight to its final place in
 *                      rank 0's arrays. Returns when every device has finished; a false promise on any rank is EVG_E_CONTRACT
 *   evg_multi_results  downloads rank 0's arrays into host buffers (NULL pointers are skipped). Rows by task (`breakdown`) are not
 *                      gathered: create with EVG_MULTI_UNIT_ROWS and ask for unit_of_task + unit_breakdown
 *
 * EVG_MULTI_LOOPBACK is the test transport of a one-GPU box: device copies instead of RCCL, and `devices` may repeat an ordinal
 * (RCCL refuses that), so the ranks of an N-GPU world run one after the other on one GPU. Never selected implicitly. */
typedef struct evg_multi evg_multi;
#define EVG_MULTI_SCATTER 0x1
#define EVG_MULTI_UNIT_ROWS 0x2
/* Every rank keeps ITS distro range resident (ABI 3.2): evg_multi_load cuts the ranges and loads each one as the resident pool of its
 * rank's context (evg_pool_load of the range, re-based) -- nothing is broadcast at tick time; a tick's structural change goes through
 * evg_multi_apply_delta, which hands every rank its part of the delta; evg_multi_tick plans + allocates every rank's pool and gathers the
 * result slices into rank 0's full-size arrays as before. The shape the 15 s cadence of the reference calls for
 * (units/crons_remote_fifteen_second.go:21,58-60): per tick a few per cent of a shard cross the link instead of the whole pool. */
#define EVG_MULTI_RESIDENT_SHARDS 0x4
#define EVG_MULTI_LOOPBACK 0x100
evg_multi* evg_multi_create(const int32_t* devices, int32_t n_devices, int32_t flags);
void evg_multi_destroy(evg_multi* m);
const char* evg_multi_last_error(const evg_multi* m); /* m == NULL: of the failed evg_multi_create */
int evg_multi_load(evg_multi* m, const evg_plan_input* in, const evg_alloc_input* alloc);
int evg_multi_tick(evg_multi* m, int64_t now_ns);
int evg_multi_results(evg_multi* m, const evg_plan_output* out, const evg_alloc_output* aout);
int evg_multi_ranges(const evg_multi* m, int32_t* d_begin, int32_t* d_end); /* n_devices entries each: rank k plans [d_begin[k], d_end[k]) */
/* Measurement: HIP events on every rank's stream around the four phases of a tick; ms4 = move-in | plan | allocate | gather, each
 * the maximum over the ranks. */
int evg_multi_profile(evg_multi* m, int enable);
int evg_multi_last_tick_ms(evg_multi* m, float* ms4);
/* Test hook: fills every rank's output block with `byte` (a slice that never arrived shows in the gathered result). */
int evg_multi_poison_outputs(evg_multi* m, int32_t byte);
/* Error behaviour of evg_multi_tick (ABI 3.2). Whatever fails -- a HIP or RCCL call, a rank's planner or allocator, a false promise
 * seen on a device -- the tick closes an RCCL group it had open, waits for every rank's stream and takes every rank's device status
 * before it returns the FIRST error: nothing of the tick is in flight after the return and the next tick starts clean. One case
 * cannot be waited for: an RCCL call that failed with its group half-issued (a send without its receive, a broadcast without all its
 * ranks). The tick then aborts the communicators and every later tick is refused: destroy the evg_multi and create a new one.
 *   evg_multi_inject_failure  test hook: the NEXT tick fails on `rank` in `phase` (0 move-in, 1 plan, 2 allocate, 3 gather) after that
 *                             rank's share of the phase was enqueued (inside the open RCCL group for 0 and 3). One shot; rank < 0 clears
 *   evg_multi_abort           for a tick that does not come back (called from ANOTHER thread) or communicators the caller no longer
 *                             trusts: ncclCommAbort on every rank; the object only accepts evg_multi_destroy afterwards
 *   evg_multi_selftest        start-up check before a scheduler routes its planning through several devices: a generated pool of mixed
 *                             shape planned + allocated on rank 0's device alone and over all the ranks must give identical outputs.
 *                             EVG_OK, EVG_E_CONTRACT (first difference in the message) or the failing call's code. Replaces the loaded pool */
/* EVG_MULTI_RESIDENT_SHARDS only: a tick's structural change, written against the WHOLE batch exactly like evg_pool_apply_delta's (current
 * global row and edge numbers, global added_distro, the new global key tables); every rank applies the part that concerns its range. `alloc`
 * (or NULL) is the tick's allocator input for the whole batch -- required when the delta grows key ranges (the resident hosts' tg_key is in the
 * distro's current key numbering). All or nothing since ABI 3.3: the ranks re-pack side by side, and a delta that any rank refuses
 * leaves EVERY rank's pool as it was (until then a refusal on rank k left ranks 0..k-1 changed and the pool had to be loaded again). */
int evg_multi_apply_delta(evg_multi* m, const evg_pool_delta* delta, const evg_alloc_input* alloc);
int evg_multi_inject_failure(evg_multi* m, int32_t rank, int32_t phase);
int evg_multi_abort(evg_multi* m);
int evg_multi_selftest(evg_multi* m);
/* ABI 3.3: the deadline of every device wait of the object and of its ranks' contexts (default 30,000 ms / EVG_DEADLINE_MS; 0 = no
 * limit; see evg_set_deadline_ms). A wait that outlives it -- a hung collective, a lost peer -- aborts the communicators (what
 * evg_multi_abort does from another thread), returns EVG_E_TIMEOUT, and the object refuses further work: destroy it and create
 * another (shim/gpu_multi.go then goes on with one device). evg_multi_debug_stall is the test hook (evg_debug_stall on one rank). */
int evg_multi_set_deadline_ms(evg_multi* m, int64_t ms);
int evg_multi_debug_stall(evg_multi* m, int32_t rank, int32_t ms);

/* Host only: the contiguous distro ranges `world` ranks plan, minimising the largest rank COST (a distro is never split: a rank's
 * results must be contiguous slices of the full-size outputs). Cost of a distro = tasks on the two-per-CU tier of the one-workgroup
 * kernel, x2 on its one-per-CU tier (2049..4096 tasks), x4 on the large-distro pipeline (measured, LAB_NOTES.md, rounds 1-4, section 4); the
 * same integers as evergreen_amd/multi.py:balanced_ranges, so every driver cuts the same ranges. Ranks past the last range get
 * empty ranges. */
int evg_balanced_ranges(const int32_t* task_off, int32_t n_distros, int32_t world, int32_t* d_begin, int32_t* d_end);

/* ---- a micro-batching front for PER-DISTRO callers (ABI 3.2) ---------------------------------------------------------------------
 * The reference calls the planner once per distro from concurrent jobs (units/crons.go:303-332 enqueues one distro-scheduler job
 * per distro; units/scheduler.go:48-49 -> scheduler.PlanDistro -> runTunablePlanner, scheduler/scheduler.go:28-52) and the host
 * allocator likewise (units/host_allocator.go:183-188). Through that call shape the host-pointer entry points above serve ONE distro
 * per call and the device's command path saturates at ~25 ms for 512 of them, whatever the number of threads. A batcher keeps the
 * call shape and issues batches: concurrent evg_batcher_plan (evg_batcher_allocate) calls are collected -- until `max_requests`
 * joined, every caller the batcher currently expects has joined (the recent peak of threads inside it, less those blocked in other
 * batches: a lone caller's batch leaves at once), or `max_wait_us` passed since the first -- and planned (allocated) by
 * ONE launch sequence over all of them; every caller packs its own columns and cuts out its own results on its own thread. Each
 * request keeps its own now_ns (and the allocator's large-parser-project figures), so its results are bit for bit those of
 * evg_plan_distros / evg_allocate_hosts on the request alone; inputs, outputs and return codes are theirs too. Errors stay per
 * request: one that fails the layout contract is refused before it joins a batch (EVG_E_CONTRACT, message in `err`); a failure of a
 * batch's own device work is returned to every caller of that batch. Thread-safe; the calls block. Up to four batches can be in flight.
 * evg_batcher_create: device ordinal; max_wait_us < 0 / max_requests <= 0 = the defaults (200 us, 64). NULL + evg_last_error(NULL)
 * on failure. evg_batcher_destroy waits for the batches in flight; calls that arrive after it began are refused. */
typedef struct evg_batcher evg_batcher;
typedef struct evg_batcher_stats {
  uint64_t batches;          /* launch sequences issued                                  */
  uint64_t requests;         /* requests they carried                                    */
  uint64_t direct_requests;  /* requests too large for a batch: passed straight through  */
  uint64_t largest_batch;    /* most requests in one batch                               */
} evg_batcher_stats;
evg_batcher* evg_batcher_create(int device_ordinal, int32_t max_wait_us, int32_t max_requests);
void evg_batcher_destroy(evg_batcher* b);
int evg_batcher_plan(evg_batcher* b, const evg_plan_input* in, const evg_plan_output* out, char* err, int32_t err_len);
int evg_batcher_allocate(evg_batcher* b, const evg_alloc_input* in, const evg_alloc_output* out, char* err, int32_t err_len);
int evg_batcher_get_stats(evg_batcher* b, evg_batcher_stats* stats);

/* ---- ABI 3.3 ----
 * evg_batcher_schedule: a distro's plan AND its host allocation as ONE request (scheduler.PlanDistro's planning phase,
 * scheduler/scheduler.go:28-52, and the allocator job's call for the same distro, units/host_allocator.go:183-188): the allocator
 * reads the plan's DistroQueueInfo rows where the planner left them on the device -- one round trip where evg_batcher_plan +
 * evg_batcher_allocate make two. `alloc_in` covers the same distros and task-group keys as `in` (same tg_off); its distro_info /
 * group_info are ignored (may be NULL). Results are those of the two calls in sequence; out->group_info comes back as
 * evg_batcher_allocate would leave the caller's copy (CountFree / CountRequired filled in).
 *
 * Resident queues: `queue_id` != 0 names the caller's queue (any stable 64-bit name of the distro) and `generation` its content
 * (anything that changes whenever ANY field of `in` other than now_ns changes: a counter the caller bumps on every change of the
 * queue, or a hash). The first request of a (queue_id, generation) travels whole and leaves its packed columns in a device-side
 * cache (EVG_BATCHER_CACHE_BYTES, default 1 GiB, least-recently-used queues evicted); a later request with the same pair --
 * the same queue 15 s later, a new clock (units/crons_remote_fifteen_second.go:21) -- skips the host-side contract check and
 * uploads its clock reading only. A request whose sizes differ from the resident generation's is refused (EVG_E_CONTRACT); a
 * caller that changes the queue without changing the generation gets the plan of the resident content. queue_id 0 = no cache
 * (evg_batcher_plan). evg_batcher_plan_queue is evg_batcher_plan with the two words.
 *
 * evg_batcher_set_deadline_ms: the deadline of every batch's device wait (default 30,000 ms / EVG_DEADLINE_MS; see
 * evg_set_deadline_ms). A batch that outlives it fails its members with EVG_E_TIMEOUT and RETIRES its slot (of four); with no slot
 * left every request fails with EVG_E_TIMEOUT: destroy the batcher and create another. Only between batches (EVG_E_INVALID
 * otherwise).
 * evg_batcher_close: refuses new requests, lets the batches in flight finish and returns when the last caller has left; the object
 * stays valid (and refusing) until evg_batcher_destroy, which a caller that cannot rule out concurrent callers calls after it has
 * joined them. (evg_batcher_destroy closes first; no call may START once it has returned.) */
int evg_batcher_schedule(evg_batcher* b, uint64_t queue_id, uint64_t generation, const evg_plan_input* in, const evg_plan_output* out,
                         const evg_alloc_input* alloc_in, const evg_alloc_output* alloc_out, char* err, int32_t err_len);
int evg_batcher_plan_queue(evg_batcher* b, uint64_t queue_id, uint64_t generation, const evg_plan_input* in, const evg_plan_output* out,
                           char* err, int32_t err_len);
int evg_batcher_set_deadline_ms(evg_batcher* b, int64_t ms);
void evg_batcher_close(evg_batcher* b);
int evg_batcher_debug_stall(evg_batcher* b, int32_t slot, int32_t ms); /* test hook: evg_debug_stall on batch slot 0..3's context */
int evg_batcher_get_cache_stats(evg_batcher* b, uint64_t* hits, uint64_t* fills, uint64_t* resident_queues, uint64_t* resident_bytes);

#ifdef __cplusplus
}
#endif
#endif /* EVG_SCHED_H */
