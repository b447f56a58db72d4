// evg_generic_flat.hip.h -- the generic path for LARGE distros as flat kernels (gfx950).
//
// A distro that the LDS path cannot take and that is large (more than 1024 tasks; config 5's 19.5 k-task distros, the
// head of a Zipf pool) has far more work than one workgroup should own: with one workgroup per distro 64 such distros
// keep 64 of 256 CUs busy. Here every phase of such distros is a kernel over ALL their rows / unit slots / key
// positions at once -- one thread per element, the distro found by binary search in the offset tables -- so the whole
// chip works on them whatever their number and size, and the two packed-key sorts are the tile kernels of
// evg_plan_lds.hip.h. The arithmetic is the generic path's (evg_kernels.hip.h), element by element; intermediates live
// in the same global scratch arrays. Distros whose value ranges do not pack into the 128-bit sort keys, and flagged
// distros that are small, are left to the one-workgroup-per-distro kernel that runs afterwards.
#pragma once

#include "evg_kernels.hip.h"

namespace evg {

constexpr int kFlatBlock = 256;

// largest d in [0, D) with off[d] <= x (x < off[D]); empty distros (off[d] == off[d+1]) are skipped over.
__device__ __forceinline__ int distro_of(const int32_t* __restrict__ off, int D, int x) {
  int lo = 0, hi = D;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}
// the same over the unit-slot index space, whose per-distro base is task_off + tg_off + ver_off
__device__ __forceinline__ int distro_of_slot(const PlanArgs& a, int D, long long z) {
  int lo = 0, hi = D;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((long long)a.in.task_off[mid] + a.in.tg_off[mid] + a.in.ver_off[mid] <= z) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ bool flat_distro(const PlanArgs& a, int d) {
  return d >= a.d0 && d < a.d1 && a.w_generic[d] != 0 && a.in.task_off[d + 1] - a.in.task_off[d] > 1024;
}

__device__ __forceinline__ DC flat_context(const PlanArgs& a, int d) {  // distro_context without the edge-range loads
  DC c;
  c.d = d; c.D = a.in.n_distros;
  c.lo = a.in.task_off[d]; c.n = a.in.task_off[d + 1] - c.lo;
  c.tg_lo = a.in.tg_off[d]; c.ntg = a.in.tg_off[d + 1] - c.tg_lo;
  c.ver_lo = a.in.ver_off[d]; c.nver = a.in.ver_off[d + 1] - c.ver_lo;
  c.gv = a.in.distros[d].group_versions != 0;
  c.now = a.in.now_ns;
  if (c.gv) { c.tg_base = 0; c.ver_base = c.ntg; c.S = c.ntg + c.nver; }
  else { c.tg_base = c.n; c.ver_base = c.n + c.ntg; c.S = c.n + c.ntg; }
  int P = 1;
  while (P < c.n) P <<= 1;
  c.P = P; c.eb = 0; c.ne = 0; c.eL = false;
  return c;
}

__device__ __forceinline__ Mem make_mem(const PlanArgs& a, const DC& c) {
  const size_t sb = (size_t)c.lo + c.tg_lo + c.ver_lo;  // disjoint slot range of this distro
  Mem m;
  m.tiq = a.w_tiq + sb; m.dur = a.w_dur + sb; m.maxpri = a.w_maxpri + sb; m.val = a.w_val + sb;
  m.cnt = a.w_cnt + sb; m.maxnd = a.w_maxnd + sb; m.minrow = a.w_minrow + sb; m.hash = a.w_hash + sb;
  m.pslot = a.w_pslot + c.lo;
  m.k0 = a.w_k0 + c.lo; m.k1 = a.w_k1 + c.lo; m.idx = a.w_idx + 2 * (size_t)c.lo; m.pos = a.w_pos + c.lo;
  const evg_task_soa& t = a.in.tasks;
  m.c_pri = t.priority + c.lo; m.c_dur = t.expected_duration_ns + c.lo;
  m.c_tgo = t.task_group_order + c.lo; m.c_nd = t.num_dependents + c.lo;
  m.g_cnt = a.g_cnt; m.g_cover = a.g_cover; m.g_wait = a.g_wait; m.g_mq = a.g_mq; m.g_first = a.g_first;
  m.g_dur = a.g_dur; m.g_dover = a.g_dover;
  m.g0 = c.d; m.gk = c.D + c.tg_lo;
  return m;
}

// F1: per-distro state, primary unit slot per row, zeroed unit accumulators and group rows.
__global__ void __launch_bounds__(kFlatBlock) k_flat_init(const PlanArgs a) {
  const int D = a.in.n_distros, N = a.in.tasks.n_tasks;
  const long long gsz = (long long)gridDim.x * kFlatBlock, g0 = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  for (long long d = g0; d < D; d += gsz) {
    GState g{};
    g.vmin = g.dmin = g.pmin = ~0ull; g.tmin = g.nmin = ~0u;
    a.w_gstate[d] = g;
  }
  const evg_task_soa& t = a.in.tasks;
  for (long long r = g0; r < N; r += gsz) {
    const int d = distro_of(a.in.task_off, D, (int)r);
    if (!flat_distro(a, d)) continue;
    const DC c = flat_context(a, d);
    const int i = (int)r - c.lo;
    a.w_pslot[r] = (uint32_t)pslot_of(i, t.tg_key[r], t.version_key[r], c);
  }
  const long long Stot = (long long)N + a.in.n_task_groups + a.in.n_versions;
  for (long long z = g0; z < Stot; z += gsz) {
    const int d = distro_of_slot(a, D, z);
    if (!flat_distro(a, d)) continue;
    // Without grouped versions slot u < n is the unit keyed by row u's own id: only row u is ever its PRIMARY member, so
    // it is initialised here with that row's own contribution (plain stores) and k_flat_reduce only adds dependents.
    const int lo = a.in.task_off[d], n = a.in.task_off[d + 1] - lo;
    const long long u = z - ((long long)lo + a.in.tg_off[d] + a.in.ver_off[d]);
    int64_t tq = 0, du = 0, mp = 0;
    uint32_t cw = 0, mr = 0xFFFFFFFFu;
    int32_t mn = 0;
    if (u < n && a.in.distros[d].group_versions == 0) {
      const long long r = lo + u;
      if (t.tg_key[r] < 0) {
        const uint32_t f = t.flags[r];
        const int64_t qts = t.queue_ts_ns[r], pri = t.priority[r];
        const int32_t nd = t.num_dependents[r];
        const uint32_t rc = f & EVG_TF_REQ_MASK;
        tq = qts == EVG_TIME_GO_ZERO ? 0 : time_sub(a.in.now_ns, qts);
        du = t.expected_duration_ns[r];
        mp = pri > 0 ? pri : 0;
        mn = nd > 0 ? nd : 0;
        mr = (uint32_t)u;
        cw = 1u | UF_DISTRO | UF_NONGROUP | (rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u) |
             ((f & EVG_TF_GENERATE) ? UF_GENERATE : 0u) | ((f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u);
      }
    }
    a.w_tiq[z] = tq; a.w_dur[z] = du; a.w_maxpri[z] = mp; a.w_cnt[z] = cw; a.w_maxnd[z] = mn; a.w_minrow[z] = mr;
  }
  const long long G = (long long)D + a.in.n_task_groups;
  for (long long g = g0; g < G; g += gsz) {
    const int d = g < D ? (int)g : distro_of(a.in.tg_off, D, (int)(g - D));
    if (!flat_distro(a, d)) continue;
    a.g_cnt[g] = 0; a.g_cover[g] = 0; a.g_wait[g] = 0; a.g_mq[g] = 0; a.g_first[g] = 0xFFFFFFFFu; a.g_dur[g] = 0; a.g_dover[g] = 0;
  }
}

// F2: Unit.info (planner.go:302-337): every row adds itself to each unit it is a member of (global atomics).
__global__ void __launch_bounds__(kFlatBlock) k_flat_reduce(const PlanArgs a) {
  const int D = a.in.n_distros, N = a.in.tasks.n_tasks;
  const long long r = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  if (r >= N) return;
  const int d = distro_of(a.in.task_off, D, (int)r);
  if (!flat_distro(a, d)) return;
  const DC c = flat_context(a, d);
  const Mem m = make_mem(a, c);
  const evg_task_soa& t = a.in.tasks;
  const int i = (int)r - c.lo;
  const int tgk = t.tg_key[r], verk = t.version_key[r];
  const uint32_t f = t.flags[r];
  const int64_t pri = t.priority[r], dur = t.expected_duration_ns[r], qts = t.queue_ts_ns[r];
  const int32_t nd = t.num_dependents[r];
  const int64_t tiq = qts == EVG_TIME_GO_ZERO ? 0 : time_sub(c.now, qts);
  const uint32_t rc = f & EVG_TF_REQ_MASK;
  uint32_t uf = rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u;
  uf |= tgk < 0 ? UF_NONGROUP : 0u;
  uf |= (f & EVG_TF_GENERATE) ? UF_GENERATE : 0u;
  uf |= (f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u;
  const bool own_primary = !c.gv && tgk < 0;  // already in its own unit (k_flat_init)
  for_each_unit<true>(m, c, i, tgk, verk, t.dep_off[r], t.dep_off[r + 1], t.dep_idx, [&](int u, bool primary) {
    if (primary && own_primary) return;
    if (tiq != 0) atomicAdd((unsigned long long*)&m.tiq[u], (unsigned long long)tiq);
    atomicAdd((unsigned long long*)&m.dur[u], (unsigned long long)dur);
    atomicAdd(&m.cnt[u], 1u);
    // The monotone fields (max, min, or) are read first: a stale value can only make the atomic happen needlessly, never
    // skip one that matters, and after the first few members of a unit most rows have nothing to add.
    if (pri > 0 && __hip_atomic_load(&m.maxpri[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pri) atomicMax((long long*)&m.maxpri[u], (long long)pri);
    if (nd > 0 && __hip_atomic_load(&m.maxnd[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nd) atomicMax(&m.maxnd[u], nd);
    const uint32_t bits = uf | (primary ? UF_DISTRO : 0u);  // SetDistro only via the primary key (:447)
    if ((__hip_atomic_load(&m.cnt[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(&m.cnt[u], bits);
    if (__hip_atomic_load(&m.minrow[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint32_t)i) atomicMin(&m.minrow[u], (uint32_t)i);
  });
}

// F3: unitInfo.value() (planner.go:209-300) per unit slot; units whose distro is nil are dropped (:81).
__global__ void __launch_bounds__(kFlatBlock) k_flat_score(const PlanArgs a) {
  const int D = a.in.n_distros;
  const long long Stot = (long long)a.in.tasks.n_tasks + a.in.n_task_groups + a.in.n_versions;
  const long long z = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  if (z >= Stot) return;
  const int d = distro_of_slot(a, D, z);
  if (!flat_distro(a, d)) return;
  const evg_distro_params p = a.in.distros[d];
  const uint32_t cw = a.w_cnt[z];
  const int64_t nu = cw & UF_COUNT_MASK;
  int64_t v = INT64_MIN;
  if (nu > 0 && (cw & UF_DISTRO)) v = unit_value(p, nu, a.w_tiq[z], a.w_dur[z], a.w_maxpri[z], a.w_maxnd[z], cw, nullptr);
  a.w_val[z] = v;
}

// F4: each row's emitting unit (TaskPlan.Export's first-occurrence dedup) + the ranges the key packing needs.
__global__ void __launch_bounds__(kFlatBlock) k_flat_elect(const PlanArgs a) {
  const int D = a.in.n_distros, N = a.in.tasks.n_tasks;
  const long long r = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  const int d = r < N ? distro_of(a.in.task_off, D, (int)r) : -1;
  const bool on = d >= 0 && flat_distro(a, d);
  uint64_t uv = 0, ud = 0, up = 0;
  uint32_t ut = 0, un = 0;
  if (on) {
    const DC c = flat_context(a, d);
    const Mem m = make_mem(a, c);
    const evg_task_soa& t = a.in.tasks;
    const int i = (int)r - c.lo;
    int best = -1;
    int64_t bv = INT64_MIN;
    uint32_t bm = 0;
    for_each_unit<false>(m, c, i, t.tg_key[r], t.version_key[r], t.dep_off[r], t.dep_off[r + 1], t.dep_idx, [&](int u, bool) {
      const int64_t v = m.val[u];
      if (v == INT64_MIN) return;
      const uint32_t mr = m.minrow[u];
      if (best < 0 || v > bv || (v == bv && (mr < bm || (mr == bm && u < best)))) { best = u; bv = v; bm = mr; }
    });
    m.k0[i] = bv;
    m.k1[i] = ((Mem::k1_t)bm << Mem::kShift) | (Mem::k1_t)best;
    if (a.out.breakdown) {
      const evg_distro_params p = a.in.distros[d];
      const uint32_t cw = m.cnt[best];
      unit_value(p, cw & UF_COUNT_MASK, m.tiq[best], m.dur[best], m.maxpri[best], m.maxnd[best], cw,
                 a.out.breakdown + (size_t)r * EVG_BREAKDOWN_FIELDS);
    }
    uv = ub(bv); ud = ub(t.expected_duration_ns[r]); up = ub(t.priority[r]);
    ut = ub(t.task_group_order[r]); un = ub(t.num_dependents[r]);
  }
  // ranges: one set of atomics per BLOCK when the whole block is in one flat distro (a 19.5 k-row distro is ~76 blocks:
  // per-wave or per-lane atomics on its ten range words serialise), per lane at distro boundaries
  __shared__ unsigned long long s64[6];
  __shared__ uint32_t s32[4];
  __shared__ int s_d;
  if (threadIdx.x == 0) s_d = d;
  if (threadIdx.x < 6) s64[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull;
  if (threadIdx.x < 4) s32[threadIdx.x] = (threadIdx.x & 1) ? 0u : ~0u;
  __syncthreads();
  const bool uniform = __syncthreads_and(on && d == s_d) != 0;
  if (uniform) {
    const uint64_t a0 = wave_min(uv), a1 = wave_max(uv), a2 = wave_min(ud), a3 = wave_max(ud), a4 = wave_min(up), a5 = wave_max(up);
    const uint32_t b0 = wave_min(ut), b1 = wave_max(ut), b2 = wave_min(un), b3 = wave_max(un);
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&s64[0], (unsigned long long)a0); atomicMax(&s64[1], (unsigned long long)a1);
      atomicMin(&s64[2], (unsigned long long)a2); atomicMax(&s64[3], (unsigned long long)a3);
      atomicMin(&s64[4], (unsigned long long)a4); atomicMax(&s64[5], (unsigned long long)a5);
      atomicMin(&s32[0], b0); atomicMax(&s32[1], b1); atomicMin(&s32[2], b2); atomicMax(&s32[3], b3);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      GState* g = &a.w_gstate[d];
      atomicMin(&g->vmin, s64[0]); atomicMax(&g->vmax, s64[1]); atomicMin(&g->dmin, s64[2]); atomicMax(&g->dmax, s64[3]);
      atomicMin(&g->pmin, s64[4]); atomicMax(&g->pmax, s64[5]);
      atomicMin(&g->tmin, s32[0]); atomicMax(&g->tmax, s32[1]); atomicMin(&g->nmin, s32[2]); atomicMax(&g->nmax, s32[3]);
    }
  } else if (on) {
    GState* g = &a.w_gstate[d];
    atomicMin(&g->vmin, (unsigned long long)uv); atomicMax(&g->vmax, (unsigned long long)uv);
    atomicMin(&g->dmin, (unsigned long long)ud); atomicMax(&g->dmax, (unsigned long long)ud);
    atomicMin(&g->pmin, (unsigned long long)up); atomicMax(&g->pmax, (unsigned long long)up);
    atomicMin(&g->tmin, ut); atomicMax(&g->tmax, ut); atomicMin(&g->nmin, un); atomicMax(&g->nmax, un);
  }
}

// F5: the packed keys of the first sort (see plan_distro's P5); the thread of key 0 hands the distro's tiles to the
// sort kernels. A distro whose ranges do not pack keeps fast == 0 and is finished by the one-workgroup kernel later.
__global__ void __launch_bounds__(kFlatBlock) k_flat_keys(const PlanArgs a) {
  const int D = a.in.n_distros, N = a.in.tasks.n_tasks;
  const long long z = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  if (z >= 2 * (long long)N) return;
  const int d = distro_of(a.in.task_off, D, (int)(z >> 1));
  if (!flat_distro(a, d)) return;
  const int lo = a.in.task_off[d], n = a.in.task_off[d + 1] - lo;
  int P = 1;
  while (P < n) P <<= 1;
  const long long i = z - 2 * (long long)lo;
  if (i >= P) return;
  GState* g = &a.w_gstate[d];
  const int vb = bits_of(g->vmax - g->vmin);
  const int bt = bits_of((uint64_t)(g->tmax - g->tmin)), bn = bits_of((uint64_t)(g->nmax - g->nmin)), bp = bits_of(g->pmax - g->pmin),
            bd = bits_of(g->dmax - g->dmin);
  if (!(vb <= 55 && bt + bn + bp + bd <= 64)) return;
  K128 k{~0ull, ~0ull};
  if (i < n) {
    const uint64_t vc = g->vmax - ub(a.w_k0[lo + i]), k1 = a.w_k1[lo + i];
    const uint64_t mr = k1 >> 32, sl = k1 & 0xFFFFFFFFu;
    k.hi = (vc << 9) | (mr >> 15);
    k.lo = ((mr & 0x7FFFu) << 49) | (sl << 24) | (uint64_t)i;
  }
  ((K128*)a.w_key)[2 * (size_t)lo + i] = k;
  if (i == 0) {
    g->bn = bn; g->bp = bp; g->bd = bd; g->fast = 1;
    const int nt = P >> 11;
    const int base = atomicAdd(a.w_ntiles, nt);
    for (int q = 0; q < nt; q++) { a.w_tiles[2 * (base + q)] = d; a.w_tiles[2 * (base + q) + 1] = q; }
  }
}

// F6: queue order out + inverse permutation, from the keys of the second sort.
__global__ void __launch_bounds__(kFlatBlock) k_flat_order(const PlanArgs a) {
  const int D = a.in.n_distros, N = a.in.tasks.n_tasks;
  const long long r = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  if (r >= N) return;
  const int d = distro_of(a.in.task_off, D, (int)r);
  if (!flat_distro(a, d) || !a.w_gstate[d].fast) return;
  const int lo = a.in.task_off[d];
  const int q = (int)r - lo;
  const uint32_t i = (uint32_t)(((const K128*)a.w_key)[2 * (size_t)lo + q].lo & 0xFFFFFFu);
  a.out.order[r] = lo + (int)i;
  a.w_idx[2 * (size_t)lo + q] = i;
  a.w_pos[lo + i] = (uint32_t)q;
}

// F7: checkDependenciesMet per row (scheduler.go:70-76,180-187); the effective DependenciesMetTime is parked in wait_ns.
__global__ void __launch_bounds__(kFlatBlock) k_flat_deps_met(const PlanArgs a) {
  const int D = a.in.n_distros, N = a.in.tasks.n_tasks;
  const long long r = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  if (r >= N) return;
  const int d = distro_of(a.in.task_off, D, (int)r);
  if (!flat_distro(a, d) || !a.w_gstate[d].fast) return;
  const evg_task_soa& t = a.in.tasks;
  const int lo = a.in.task_off[d], n = a.in.task_off[d + 1] - lo;
  const int64_t now = a.in.now_ns;
  const uint32_t f = t.flags[r];
  const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
  const int64_t dmt = t.deps_met_ts_ns[r];
  bool met = (e1 == e0) || (f & EVG_TF_OVERRIDE_DEPS) || !is_zero_time(dmt);  // HasDependenciesMet task.go:3406
  int64_t mettime = dmt;
  if (!met) {
    bool all = true;
    for (int e = e0; e < e1; e++) {
      const int j = t.dep_idx[e] - lo;
      const uint32_t info = t.dep_info[e];
      uint32_t st;
      bool blk;
      if ((unsigned)j < (unsigned)n) {
        const uint32_t fj = (uint32_t)t.flags[lo + j];
        st = (fj & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT;
        blk = fj & EVG_TF_BLOCKED;
      } else {
        if (info & EVG_DEP_MISSING) { all = false; break; }
        st = (info & EVG_DEP_STATE_MASK) >> EVG_DEP_STATE_SHIFT;
        blk = info & EVG_DEP_BLOCKED;
      }
      const uint32_t req = info & EVG_DEP_REQ_MASK;  // SatisfiesDependency task.go:546-561
      const bool sat = req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk) : false;
      if (!sat) { all = false; break; }
    }
    if (all) {
      met = true;  // setDependenciesMetTime task.go:690-701
      int64_t mt = 0;
      if (t.dep_finished_ts_ns)
        for (int e = e0; e < e1; e++) {
          const int64_t fa = t.dep_finished_ts_ns[e];
          if (!is_zero_time(fa) && fa > mt) mt = fa;
        }
      mettime = is_zero_time(mt) ? now : mt;
    }
  }
  a.out.deps_met[r] = met ? 1 : 0;
  a.out.wait_ns[r] = mettime;
  if (met && (f & EVG_TF_REQ_MASK) == EVG_TF_REQ_MERGE) atomicOr(&a.w_gstate[d].any_mq, 1u);
}

// F8: GetDistroQueueInfo's per-task-group sums (scheduler.go:78-160). Wave-reduced when the whole wave is in one distro
// (the standalone row takes ~90% of the rows and would otherwise be one contended word).
__global__ void __launch_bounds__(kFlatBlock) k_flat_sums(const PlanArgs a) {
  const int D = a.in.n_distros, N = a.in.tasks.n_tasks;
  const long long r = (long long)blockIdx.x * kFlatBlock + threadIdx.x;
  const int d = r < N ? distro_of(a.in.task_off, D, (int)r) : -1;
  const bool on = d >= 0 && flat_distro(a, d) && a.w_gstate[d].fast;
  uint32_t s_cnt = 0, s_cover = 0, s_wait = 0, s_mq = 0, s_first = 0xFFFFFFFFu, n_met = 0, n_mq = 0, n_s3 = 0, sec = 0;
  uint64_t s_dur = 0, s_dover = 0;
  if (on) {
    const evg_task_soa& t = a.in.tasks;
    const evg_distro_params p = a.in.distros[d];
    const int tg_lo = a.in.tg_off[d];
    const int64_t T = target_time_for_queue(p, a.w_gstate[d].any_mq != 0);
    const bool incl = p.includes_dependencies != 0;
    const uint32_t f = t.flags[r];
    const int tgk = t.tg_key[r];
    const bool met = a.out.deps_met[r] != 0;
    const int64_t mettime = a.out.wait_ns[r];
    const int64_t dur = t.expected_duration_ns[r];
    const bool merge = (f & EVG_TF_REQ_MASK) == EVG_TF_REQ_MERGE;
    const bool count = !incl || met;
    const bool over = count && dur > T;
    int64_t wait = 0;
    bool wait_over = false;
    if (count && met) {
      int64_t start = t.scheduled_ts_ns[r];
      if (mettime > start) start = mettime;  // DependenciesMetTime.After(startTime)
      wait = time_sub(a.in.now_ns, start);
      wait_over = wait > T;
    }
    a.out.wait_ns[r] = wait;
    if (f & EVG_TF_OTHER_DISTRO) sec = 1;
    if (met) { n_met = 1; n_mq = merge; n_s3 = (f & EVG_TF_S3_STORAGE) ? 1 : 0; }
    const uint32_t posq = a.w_pos[r];
    if (tgk < 0) {
      s_first = posq; s_cnt = count; s_dur = count ? (uint64_t)dur : 0; s_cover = over; s_dover = over ? (uint64_t)dur : 0;
      s_wait = wait_over; s_mq = met && merge;
    } else {
      const int g = a.in.n_distros + tgk;  // row of task group key tgk
      (void)tg_lo;
      atomicMin(&a.g_first[g], posq);
      if (count) { atomicAdd(&a.g_cnt[g], 1u); atomicAdd((unsigned long long*)&a.g_dur[g], (unsigned long long)dur); }
      if (over) { atomicAdd(&a.g_cover[g], 1u); atomicAdd((unsigned long long*)&a.g_dover[g], (unsigned long long)dur); }
      if (wait_over) atomicAdd(&a.g_wait[g], 1u);
      if (met && merge) atomicAdd(&a.g_mq[g], 1u);
    }
  }
  auto flush = [&](int dd, uint32_t c0, uint64_t c1, uint32_t c2, uint64_t c3, uint32_t c4, uint32_t c5, uint32_t c6, uint32_t c7,
                   uint32_t c8, uint32_t c9, uint32_t c10) {
    GState* g = &a.w_gstate[dd];
    if (c0) atomicAdd(&a.g_cnt[dd], c0);
    if (c1) atomicAdd((unsigned long long*)&a.g_dur[dd], (unsigned long long)c1);
    if (c2) atomicAdd(&a.g_cover[dd], c2);
    if (c3) atomicAdd((unsigned long long*)&a.g_dover[dd], (unsigned long long)c3);
    if (c4) atomicAdd(&a.g_wait[dd], c4);
    if (c5) atomicAdd(&a.g_mq[dd], c5);
    if (c6 != 0xFFFFFFFFu) atomicMin(&a.g_first[dd], c6);
    if (c7) atomicAdd(&g->n_met, c7);
    if (c8) atomicAdd(&g->n_mq, c8);
    if (c9) atomicAdd(&g->n_s3, c9);
    if (c10) atomicOr(&g->sec, 1u);
  };
  // the standalone row and the distro counters: one flush per block when the whole block is in one flat distro
  __shared__ unsigned long long b64[2];
  __shared__ uint32_t b32[9];
  __shared__ int s_d;
  if (threadIdx.x == 0) s_d = d;
  if (threadIdx.x < 2) b64[threadIdx.x] = 0;
  if (threadIdx.x < 9) b32[threadIdx.x] = threadIdx.x == 4 ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  const bool uniform = __syncthreads_and(on && d == s_d) != 0;
  if (uniform) {
    const uint32_t c0 = wave_sum(s_cnt), c2 = wave_sum(s_cover), c4 = wave_sum(s_wait), c5 = wave_sum(s_mq), c6 = wave_min(s_first),
                   c7 = wave_sum(n_met), c8 = wave_sum(n_mq), c9 = wave_sum(n_s3), c10 = wave_max(sec);
    const uint64_t c1 = wave_sum(s_dur), c3 = wave_sum(s_dover);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&b32[0], c0); atomicAdd(&b32[1], c2); atomicAdd(&b32[2], c4); atomicAdd(&b32[3], c5); atomicMin(&b32[4], c6);
      atomicAdd(&b32[5], c7); atomicAdd(&b32[6], c8); atomicAdd(&b32[7], c9); atomicOr(&b32[8], c10);
      atomicAdd(&b64[0], (unsigned long long)c1); atomicAdd(&b64[1], (unsigned long long)c3);
    }
    __syncthreads();
    if (threadIdx.x == 0) flush(d, b32[0], b64[0], b32[1], b64[1], b32[2], b32[3], b32[4], b32[5], b32[6], b32[7], b32[8]);
  } else if (on) {
    flush(d, s_cnt, s_dur, s_cover, s_dover, s_wait, s_mq, s_first, n_met, n_mq, n_s3, sec);
  }
}

// F9: rows out (model.TaskGroupInfo, model.DistroQueueInfo): one workgroup per distro.
__global__ void __launch_bounds__(kFlatBlock) k_flat_rows(const PlanArgs a) {
  __shared__ unsigned s_red[16];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int d = a.d0 + blockIdx.x; d < a.d1; d += gridDim.x) {
    if (!flat_distro(a, d) || !a.w_gstate[d].fast) continue;
    const DC c = flat_context(a, d);
    const evg_task_soa& t = a.in.tasks;
    __syncthreads();
    if (tid < 16) s_red[tid] = 0;
    __syncthreads();
    uint64_t t_dur = 0, t_dover = 0;
    uint32_t t_cover = 0, t_wait = 0, t_rows = 0;
    for (int k = tid; k < c.ntg + 1; k += kFlatBlock) {
      const int g = k == 0 ? d : c.D + c.tg_lo + (k - 1);
      const uint32_t first = a.g_first[g];
      const bool present = first != 0xFFFFFFFFu;
      evg_group_info gi;
      gi.expected_duration_ns = (int64_t)a.g_dur[g];
      gi.duration_over_threshold_ns = (int64_t)a.g_dover[g];
      gi.count = (int32_t)a.g_cnt[g];
      gi.max_hosts = present ? t.task_group_max_hosts[c.lo + (int)a.w_idx[2 * (size_t)c.lo + first]] : 0;
      gi.count_duration_over_threshold = (int32_t)a.g_cover[g];
      gi.count_wait_over_threshold = (int32_t)a.g_wait[g];
      gi.count_dep_filled_merge_queue_tasks = (int32_t)a.g_mq[g];
      gi.present = present ? 1 : 0;
      gi.count_free = 0;
      gi.count_required = 0;
      a.out.group_info[g] = gi;
      t_dur += a.g_dur[g]; t_dover += a.g_dover[g]; t_cover += a.g_cover[g]; t_wait += a.g_wait[g];
      t_rows += present ? 1u : 0u;
    }
    t_dur = wave_sum(t_dur); t_dover = wave_sum(t_dover);
    t_cover = wave_sum(t_cover); t_wait = wave_sum(t_wait); t_rows = wave_sum(t_rows);
    if (lane == 0) {
      if (t_cover) atomicAdd(&s_red[5], t_cover);
      if (t_wait) atomicAdd(&s_red[6], t_wait);
      if (t_rows) atomicAdd(&s_red[8], t_rows);
      atomicAdd((unsigned long long*)&s_red[10], (unsigned long long)t_dur);
      atomicAdd((unsigned long long*)&s_red[12], (unsigned long long)t_dover);
    }
    __syncthreads();
    if (tid == 0) {
      const GState g = a.w_gstate[d];
      evg_distro_info di;
      di.expected_duration_ns = (int64_t)(*(unsigned long long*)&s_red[10]);
      di.max_duration_threshold_ns = target_time_for_queue(a.in.distros[d], g.any_mq != 0);
      di.duration_over_threshold_ns = (int64_t)(*(unsigned long long*)&s_red[12]);
      di.length = c.n;
      di.length_with_dependencies_met = (int32_t)g.n_met;
      di.count_dep_filled_merge_queue_tasks = (int32_t)g.n_mq;
      di.count_duration_over_threshold = (int32_t)s_red[5];
      di.count_wait_over_threshold = (int32_t)s_red[6];
      di.num_queued_large_parser_project_tasks = (int32_t)g.n_s3;
      di.secondary_queue = (int32_t)g.sec;
      di.n_task_group_infos = (int32_t)s_red[8];
      a.out.distro_info[d] = di;
    }
  }
}

}  // namespace evg
