// evg_validate.hpp -- the host-side check of the layout contract (include/evg_sched.h: evg_validate_plan_input). Plain C++, no HIP:
// included by evg_sched.hip and by the CPU build of the micro-batching front's state machine (tests/cpp/test_batcher_tsan.cpp).
#pragma once

#include <algorithm>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "../../include/evg_sched.h"

// One distro's share of the layout contract; 0 or EVG_E_CONTRACT with the message in `err`.
static int validate_distro(const evg_plan_input* in, int d, char* err, size_t err_len) {
  const evg_task_soa& t = in->tasks;
  auto fail = [&](const char* fmt, long a, long b) {
    snprintf(err, err_len, fmt, a, b);
    return EVG_E_CONTRACT;
  };
  const int lo = in->task_off[d], hi = in->task_off[d + 1];
  if (hi < lo) return fail("task_off not monotone at distro %ld (%ld)", d, hi);
  if (hi - lo >= (1 << 24)) return fail("distro %ld has %ld tasks; the limit is 2^24-1", d, hi - lo);
  const int tg_lo = in->tg_off[d], tg_hi = in->tg_off[d + 1], ver_lo = in->ver_off[d], ver_hi = in->ver_off[d + 1];
  if (tg_hi < tg_lo || ver_hi < ver_lo) return fail("key offsets not monotone at distro %ld (%ld)", d, tg_hi);
  for (int r = lo; r < hi; r++) {
    // keys: any one-to-one interning of the strings into the distro's range (ABI 3.1: neither first-appearance order nor
    // density is required -- a resident pool that lost the last task of a group keeps the key, with no row behind it)
    const int g = t.tg_key[r], v = t.version_key[r];
    if (g != -1 && (g < tg_lo || g >= tg_hi)) return fail("row %ld: tg_key %ld is neither -1 nor in the distro's key range", r, g);
    if (v < ver_lo || v >= ver_hi) return fail("row %ld: version_key %ld is outside the distro's key range", r, v);
    const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
    if (e1 < e0) return fail("dep_off not monotone at row %ld (%ld)", r, e1);
    if (e0 < 0 || e1 > t.n_edges) return fail("row %ld: dep_off %ld outside [0, n_edges]", r, e1);
    // a dependency is a row of the SAME distro's queue or -1 (not in this queue: its state rides in dep_info); a row of
    // another distro would be read as "not in the queue" with status bits nobody filled
    for (int e = e0; e < e1; e++) {
      const int j = t.dep_idx[e];
      if (j != -1 && (j < lo || j >= hi)) return fail("edge %ld: dep_idx %ld is neither -1 nor a row of the same distro", e, j);
    }
  }
  return EVG_OK;
}

extern "C" int evg_validate_plan_input(const evg_plan_input* in, char* msg, int32_t msg_len) try {
  auto fail = [&](const char* fmt, long a, long b) {
    if (msg && msg_len > 0) snprintf(msg, msg_len, fmt, a, b);
    return EVG_E_CONTRACT;
  };
  if (!in) return EVG_E_INVALID;
  const int D = in->n_distros;
  const evg_task_soa& t = in->tasks;
  if (D < 0 || t.n_tasks < 0 || t.n_edges < 0) return fail("negative size (%ld, %ld)", D, t.n_tasks);
  if (D == 0) return EVG_OK;
  if (!in->task_off || !in->tg_off || !in->ver_off || !in->distros) return EVG_E_INVALID;
  if (t.n_tasks > 0 && (!t.tg_key || !t.version_key || !t.dep_off || (t.n_edges > 0 && !t.dep_idx))) return EVG_E_INVALID;
  if (in->task_off[0] != 0 || in->task_off[D] != t.n_tasks) return fail("task_off must span [0, n_tasks] (%ld..%ld)", in->task_off[0], in->task_off[D]);
  if (in->tg_off[0] != 0 || in->tg_off[D] != in->n_task_groups) return fail("tg_off must span [0, n_task_groups] (%ld..%ld)", in->tg_off[0], in->tg_off[D]);
  if (in->ver_off[0] != 0 || in->ver_off[D] != in->n_versions) return fail("ver_off must span [0, n_versions] (%ld..%ld)", in->ver_off[0], in->ver_off[D]);
  if (t.n_tasks == 0 && t.n_edges != 0) return fail("n_edges=%ld without tasks (n_tasks=%ld)", t.n_edges, t.n_tasks);
  if (t.n_tasks && t.dep_off[0] != 0) return fail("dep_off[0]=%ld must be 0 (n_edges=%ld)", t.dep_off[0], t.n_edges);
  if (t.n_tasks && t.dep_off[t.n_tasks] != t.n_edges) return fail("dep_off[N]=%ld != n_edges=%ld", t.dep_off[t.n_tasks], t.n_edges);
  if (in->max_distro_tasks < 0) return fail("max_distro_tasks %ld is negative (0 = unknown) (%ld)", in->max_distro_tasks, 0);
  // The per-row checks are independent per distro: a large batch is checked by a few threads (this runs inside every
  // host-pointer call; one thread needs ~1.5 ms for 1M rows + 1.3M edges). The FIRST failing distro's message is reported.
  const int nt = t.n_tasks + t.n_edges < (1 << 18) ? 1 : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  struct Text { char s[256]; };
  std::vector<int> first_bad(nt, D);
  std::vector<Text> errs(nt);
  auto work = [&](int w) noexcept {  // (nothing in here allocates: an exception on a worker thread would end the process)
    for (int d = (int)((long long)D * w / nt), d1 = (int)((long long)D * (w + 1) / nt); d < d1; d++)
      if (validate_distro(in, d, errs[w].s, sizeof errs[w].s) != EVG_OK) { first_bad[w] = d; return; }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    th.reserve(nt);
    int started = 1;
    for (; started < nt; started++) {
      try { th.emplace_back(work, started); } catch (...) { break; }  // no more threads to be had: the other ranges are checked here
    }
    work(0);
    for (int w = started; w < nt; w++) work(w);
    for (auto& x : th) x.join();
  }
  for (int w = 0; w < nt; w++)
    if (first_bad[w] < D) {
      if (msg && msg_len > 0) snprintf(msg, msg_len, "%s", errs[w].s);
      return EVG_E_CONTRACT;
    }
  return EVG_OK;
} catch (...) {  // the checker threads or their tables could not be had: nothing leaves through the C boundary
  if (msg && msg_len > 0) snprintf(msg, msg_len, "out of host resources while checking the batch");
  return EVG_E_NOMEM;
}

