// Driver of evergreen::ResidentPlanner (include/evg_host.hpp) over worlds written by tests/test_resident_planner.py (one file = the task
// lists of several consecutive ticks in the reference's own data model).
//   record <world> <out>   no device: what the planner hands to evg_pool_load / evg_pool_tick, tick by tick, as text -- the Python test
//                          compares it, array for array, with what scheduler.ResidentPlanner hands over for the same lists (which it
//                          has held to the checker's re-pack and to PlanDistros)
//   hip <lib> <world>      on the device: every tick's plans == PlanDistros (evg_plan_distros) on the same lists in the pool's row order
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "evg_host.hpp"

using namespace evergreen;

struct Tick {
  Time now = 0;
  std::vector<Distro> distros;
  std::vector<std::vector<Task>> tasks;
  std::unordered_map<std::string, std::pair<std::string, bool>> done;
};

static std::string und(const std::string& s) { return s == "-" ? "" : s; }
static Time tm(const std::string& s) { return s == "Z" ? kGoZeroTime : (Time)std::stoll(s); }

static std::vector<Tick> read_world(const char* path) {
  std::ifstream f(path);
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  std::vector<Tick> ticks;
  std::string line, w;
  while (std::getline(f, line)) {
    std::istringstream is(line);
    is >> w;
    if (w == "TICK") {
      Tick t;
      size_t D;
      is >> t.now >> D;
      ticks.push_back(std::move(t));
    } else if (w == "DISTRO") {
      Tick& t = ticks.back();
      Distro d;
      std::string ver;
      int gv;
      auto& ps = d.PlannerSettings;
      is >> d.Id >> ps.TargetTime >> ps.MergeQueueTargetTime >> gv >> ps.PatchFactor >> ps.PatchTimeInQueueFactor >> ps.CommitQueueFactor >>
          ps.MainlineTimeInQueueFactor >> ps.ExpectedRuntimeFactor >> ps.GenerateTaskFactor >> ps.NumDependentsFactor >> ps.StepbackTaskFactor >> ver;
      ps.GroupVersions = gv != 0;
      d.DispatcherSettings.Version = und(ver);
      t.distros.push_back(d);
      t.tasks.emplace_back();
    } else if (w == "TASK") {
      Task x;
      std::string grp, act, at, it, st, dm, storage;
      int gen, ovr;
      is >> x.Id >> x.DistroId >> x.Version >> grp >> x.BuildVariant >> x.Project >> x.TaskGroupOrder >> x.TaskGroupMaxHosts >> x.Requester >> x.Priority >>
          x.NumDependents >> gen >> act >> at >> it >> st >> dm >> ovr >> x.ExpectedDuration >> x.Status >> storage;
      x.TaskGroup = und(grp); x.ActivatedBy = und(act); x.GenerateTask = gen != 0; x.OverrideDependencies = ovr != 0;
      x.ActivatedTime = tm(at); x.IngestTime = tm(it); x.ScheduledTime = tm(st); x.DependenciesMetTime = tm(dm);
      x.CachedProjectStorageMethod = und(storage);
      ticks.back().tasks.back().push_back(std::move(x));
    } else if (w == "DEP") {
      Dependency d;
      std::string status, fin;
      int un;
      is >> d.TaskId >> status >> un >> fin;
      d.Status = und(status); d.Unattainable = un != 0; d.FinishedAt = tm(fin);
      ticks.back().tasks.back().back().DependsOn.push_back(d);
    } else if (w == "D") {
      std::string id, status;
      int blocked;
      is >> id >> status >> blocked;
      ticks.back().done[id] = {status, blocked != 0};
    }
  }
  return ticks;
}

template <class T>
static void dump(FILE* o, const char* name, const T* v, size_t n) {
  fprintf(o, "%s %zu", name, v ? n : (size_t)0);
  for (size_t i = 0; v && i < n; i++) fprintf(o, " %lld", (long long)v[i]);
  fputc('\n', o);
}

static ResidentPlanner::Queues queues_of(const Tick& t) {
  ResidentPlanner::Queues q;
  for (size_t d = 0; d < t.distros.size(); d++) q.push_back({&t.distros[d], &t.tasks[d]});
  return q;
}

// owners <steps> <out>: ShardedResidentPlanner::Assign over a sequence of (distro id, task count) tables, for every rank of the world --
// the Python test compares the ownership tables, the deal counts and every rank's share with scheduler.ShardedResidentPlanner's.
static int owners_mode(const char* steps, const char* out_path) {
  std::ifstream f(steps);
  if (!f) { fprintf(stderr, "cannot open %s\n", steps); return 2; }
  FILE* o = fopen(out_path, "w");
  int world = 0;
  f >> world;
  std::vector<ShardedResidentPlanner> ranks;
  for (int r = 0; r < world; r++) ranks.emplace_back(ResidentBackend{}, r, world);
  size_t n;
  while (f >> n) {
    std::vector<std::string> ids(n);
    std::vector<int64_t> counts(n);
    for (size_t i = 0; i < n; i++) f >> ids[i] >> counts[i];
    for (int r = 0; r < world; r++) {
      const auto& mine = ranks[(size_t)r].Assign(ids, counts);
      std::map<std::string, int> sorted(ranks[(size_t)r].owner().begin(), ranks[(size_t)r].owner().end());
      fprintf(o, "rank %d deals %d owner", r, ranks[(size_t)r].deals);
      for (const auto& kv : sorted) fprintf(o, " %s=%d", kv.first.c_str(), kv.second);
      fprintf(o, " mine");
      for (size_t i : mine) fprintf(o, " %zu", i);
      fprintf(o, "\n");
    }
  }
  fclose(o);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "owners")) return owners_mode(argv[2], argv[3]);
  if (argc < 4) { fprintf(stderr, "usage: %s record <world> <out> | hip <lib> <world> | owners <steps> <out>\n", argv[0]); return 2; }
  const bool record = !strcmp(argv[1], "record");
  const std::vector<Tick> ticks = read_world(record ? argv[2] : argv[3]);
  FILE* o = record ? fopen(argv[3], "w") : nullptr;
  ResidentBackend rb;
  Backend fresh;
  size_t cur_D = 0;
  std::function<size_t()> pool_D = [&] { return cur_D; };  // the distros of the pool the planner keeps (a rank's share under EVG_TEST_WORLD)
  long long cur_N = 0;  // (record: the rows of the pool, for an identity `order` -- the plans themselves are not looked at)
  int delta_ticks = 0, refused = 0, loads_after_refusal = 0;
  const int refuse_at = getenv("EVG_TEST_REFUSE_TICK") ? atoi(getenv("EVG_TEST_REFUSE_TICK")) : 0;
  if (record) {
    rb.pool_load = [&](const evg_plan_input* in) {
      fprintf(o, "LOAD %d %d %d\n", in->n_distros, in->tasks.n_tasks, in->tasks.n_edges);
      dump(o, "task_off", in->task_off, (size_t)in->n_distros + 1);
      dump(o, "tg_key", in->tasks.tg_key, (size_t)in->tasks.n_tasks);
      dump(o, "dep_idx", in->tasks.dep_idx, (size_t)in->tasks.n_edges);
      cur_N = in->tasks.n_tasks;
      return EVG_OK;
    };
    rb.pool_tick = [&](const evg_pool_delta* dl, const evg_row_update* ru, const evg_edge_update* eu, int64_t now, const evg_plan_output* out) {
      // EVG_TEST_REFUSE_TICK=k: the k-th tick that carries a delta is refused the way the device refuses one (the pool stays as it was)
      if (dl && ++delta_ticks == refuse_at) { fprintf(o, "REFUSED\n"); return EVG_E_CONTRACT; }
      if (dl) cur_N += dl->n_added - dl->n_removed;
      for (long long i = 0; i < cur_N; i++) out->order[i] = (int32_t)i;
      fprintf(o, "TICKCALL %lld %d %d %d\n", (long long)now, dl ? 1 : 0, ru ? ru->n_rows : 0, eu ? eu->n_edges : 0);
      if (dl) {
        const size_t nr = (size_t)dl->n_removed, na = (size_t)dl->n_added, ea = (size_t)dl->added.n_edges, nl = (size_t)dl->n_relinked;
        dump(o, "removed_rows", dl->removed_rows, nr); dump(o, "removed_dep_state", dl->removed_dep_state, nr);
        dump(o, "removed_finished_ts_ns", dl->removed_finished_ts_ns, nr); dump(o, "added_distro", dl->added_distro, na);
        const evg_task_soa& t = dl->added;
        dump(o, "priority", t.priority, na); dump(o, "expected_duration_ns", t.expected_duration_ns, na); dump(o, "queue_ts_ns", t.queue_ts_ns, na);
        dump(o, "scheduled_ts_ns", t.scheduled_ts_ns, na); dump(o, "deps_met_ts_ns", t.deps_met_ts_ns, na); dump(o, "num_dependents", t.num_dependents, na);
        dump(o, "task_group_order", t.task_group_order, na); dump(o, "task_group_max_hosts", t.task_group_max_hosts, na); dump(o, "tg_key", t.tg_key, na);
        dump(o, "version_key", t.version_key, na); dump(o, "flags", t.flags, na); dump(o, "added_dep_off", t.dep_off, na + 1);
        dump(o, "a_dep_idx", t.dep_idx, ea); dump(o, "a_dep_info", t.dep_info, ea); dump(o, "a_dep_finished_ts_ns", t.dep_finished_ts_ns, ea);
        dump(o, "tg_off", dl->tg_off, pool_D() + 1); dump(o, "ver_off", dl->ver_off, pool_D() + 1);
        dump(o, "relinked_edges", dl->relinked_edges, nl); dump(o, "relinked_to", dl->relinked_to, nl);
      }
      if (ru) {
        const size_t n = (size_t)ru->n_rows;
        dump(o, "u_rows", ru->rows, n); dump(o, "u_priority", ru->priority, n); dump(o, "u_expected_duration_ns", ru->expected_duration_ns, n);
        dump(o, "u_queue_ts_ns", ru->queue_ts_ns, n); dump(o, "u_scheduled_ts_ns", ru->scheduled_ts_ns, n); dump(o, "u_deps_met_ts_ns", ru->deps_met_ts_ns, n);
        dump(o, "u_num_dependents", ru->num_dependents, n); dump(o, "u_flags", ru->flags, n);
      }
      if (eu) {
        const size_t n = (size_t)eu->n_edges;
        dump(o, "e_edges", eu->edges, n); dump(o, "e_dep_info", eu->dep_info, n); dump(o, "e_dep_finished_ts_ns", eu->dep_finished_ts_ns, n);
      }
      return EVG_OK;
    };
  } else {
    rb = HipResidentBackend(argv[2]);
    fresh = HipBackend(argv[2]);
    if (refuse_at > 0) {  // the k-th tick that carries a delta goes to the device spoiled (a relinked edge far outside the pool): the library refuses it
      auto real = rb.pool_tick;
      rb.pool_tick = [&, real](const evg_pool_delta* dl, const evg_row_update* ru, const evg_edge_update* eu, int64_t now, const evg_plan_output* out) {
        if (!dl || ++delta_ticks != refuse_at) return real(dl, ru, eu, now, out);
        static const int32_t bad_edge = 1 << 30, bad_to = 0;
        evg_pool_delta spoiled = *dl;
        spoiled.n_relinked = 1; spoiled.relinked_edges = &bad_edge; spoiled.relinked_to = &bad_to;
        const int rc = real(&spoiled, ru, eu, now, out);
        printf("spoiled tick: rc %d (%s)\n", rc, rb.last_error().c_str());
        refused += rc == EVG_E_CONTRACT || rc == EVG_E_INVALID;
        return rc;
      };
    }
  }
  // EVG_TEST_WORLD / EVG_TEST_RANK: the same ticks through ShardedResidentPlanner -- this rank's share of the distros only
  const int sh_world = getenv("EVG_TEST_WORLD") ? atoi(getenv("EVG_TEST_WORLD")) : 0, sh_rank = getenv("EVG_TEST_RANK") ? atoi(getenv("EVG_TEST_RANK")) : 0;
  ShardedResidentPlanner sharded(rb, sh_world ? sh_rank : 0, sh_world ? sh_world : 1);
  ResidentPlanner plain(rb);
  ResidentPlanner& planner = sh_world ? sharded.planner : plain;
  if (sh_world) pool_D = [&] { return sharded.mine().size(); };
  int by_delta = 0, fails = 0;
  for (size_t k = 0; k < ticks.size(); k++) {
    const Tick& t = ticks[k];
    cur_D = t.distros.size();
    const DepLookup lookup = [&](const std::string& id) -> std::optional<std::pair<std::string, bool>> {
      auto it = t.done.find(id);
      if (it == t.done.end()) return std::nullopt;
      return it->second;
    };
    const auto q = queues_of(t);
    if (sh_world) {
      std::vector<PlannedQueue> part = sharded.Plan(q, t.now, nullptr, lookup);
      if (!record) { fprintf(stderr, "EVG_TEST_WORLD is for record mode\n"); return 2; }
      fprintf(o, "SHARE deals %d mine", sharded.deals);
      for (size_t i : sharded.mine()) fprintf(o, " %zu", i);
      fprintf(o, " plans %zu\n", part.size());
      if (!sharded.mine().empty()) { fprintf(o, "MODE %s\n", planner.last.mode.c_str()); by_delta += planner.last.mode == "tick"; }
      continue;
    }
    std::vector<PlannedQueue> got = planner.Plan(q, t.now, nullptr, lookup);
    by_delta += planner.last.mode == "tick";
    loads_after_refusal += planner.last.mode == "load" && planner.last.why.rfind("the device refused", 0) == 0;
    if (record) { fprintf(o, "MODE %s%s\n", planner.last.mode.c_str(), planner.last.why.rfind("the device refused", 0) == 0 ? " refused" : ""); continue; }
    // the same lists in the pool's row order (ties between equal keys fall to the lower row)
    std::vector<std::vector<Task>> res(t.distros.size());
    for (size_t d = 0; d < t.distros.size(); d++) {
      std::unordered_map<std::string, const Task*> by_id;
      for (const Task& x : t.tasks[d]) by_id[x.Id] = &x;
      for (const std::string& id : planner.ids()[d]) res[d].push_back(*by_id.at(id));
    }
    ResidentPlanner::Queues rq;
    for (size_t d = 0; d < t.distros.size(); d++) rq.push_back({&t.distros[d], &res[d]});
    const std::vector<PlannedQueue> want = PlanDistros(fresh, rq, t.now, nullptr, lookup);
    for (size_t d = 0; d < want.size(); d++) {
      bool same = got[d].plan.size() == want[d].plan.size();
      for (size_t p = 0; same && p < want[d].plan.size(); p++) {
        const Task &a = got[d].plan[p], &b = want[d].plan[p];
        same = a.Id == b.Id && a.SortingValueBreakdown.TotalValue == b.SortingValueBreakdown.TotalValue &&
               a.SortingValueBreakdown.RankValueBreakdown.PatchWaitTimeImpact == b.SortingValueBreakdown.RankValueBreakdown.PatchWaitTimeImpact &&
               a.WaitSinceDependenciesMet == b.WaitSinceDependenciesMet && a.ExpectedDuration == b.ExpectedDuration && a.DependenciesMetTime == b.DependenciesMetTime;
      }
      const DistroQueueInfo &x = got[d].info, &y = want[d].info;
      same = same && x.Length == y.Length && x.LengthWithDependenciesMet == y.LengthWithDependenciesMet && x.ExpectedDuration == y.ExpectedDuration &&
             x.CountDurationOverThreshold == y.CountDurationOverThreshold && x.CountWaitOverThreshold == y.CountWaitOverThreshold &&
             x.DurationOverThreshold == y.DurationOverThreshold && x.TaskGroupInfos.size() == y.TaskGroupInfos.size() && got[d].n_units == want[d].n_units;
      if (same) {
        std::map<std::string, const TaskGroupInfo*> by_name;
        for (const auto& g : y.TaskGroupInfos) by_name[g.Name] = &g;
        for (const auto& g : x.TaskGroupInfos) {
          auto it = by_name.find(g.Name);
          same = same && it != by_name.end() && it->second->Count == g.Count && it->second->ExpectedDuration == g.ExpectedDuration &&
                 it->second->MaxHosts == g.MaxHosts && it->second->CountWaitOverThreshold == g.CountWaitOverThreshold;
        }
      }
      if (!same) { fails++; printf("FAIL tick %zu (%s) distro %zu\n", k, planner.last.mode.c_str(), d); }
    }
  }
  if (o) fclose(o);
  printf("resident planner: %zu ticks (%d by delta), %d mismatches\n", ticks.size(), by_delta, fails);
  if (refuse_at > 0 && !record) {
    printf("refused by the device: %d, answered by a load: %d\n", refused, loads_after_refusal);
    if (refused != 1 || loads_after_refusal != 1) return 1;
  }
  return fails ? 1 : 0;
}
