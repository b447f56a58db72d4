// evg_dispatch.hip.h -- the DAG dispatcher's rebuild for all distros.
//
// Reference: model/task_queue_service_dependency.go:153-250 (basicCachedDAGDispatcherImpl.rebuild). The order it stores in
// d.sorted is gonum v0.17.0's topo.SortStabilized with "order by queueIndex": Tarjan's search started from the nodes in
// DESCENDING queue index, successors taken in descending queue index, components listed in reverse order of completion.
// A lexicographic depth-first search is sequential in general, but it splits exactly along the search's own roots:
//
//   top(w) = the largest queue index of a node that reaches w (w itself included).
//
// The sequential search visits w from root r iff top(w) == r (everything a larger node reaches is finished before r is
// started, and an edge into finished nodes changes nothing in Tarjan's algorithm), so the nodes with top == r form one
// independent search, and d.sorted is these blocks in ASCENDING r, each block in reverse order of completion. top() is a
// max-propagation along the edges (parallel, converges in <= longest-path rounds); every block is then searched by one
// lane with its explicit stacks carved out of the block's own slice of the scratch. Work per distro is O(n + e); the
// critical path is the largest block.
//
// One workgroup per distro. Everything lives in global scratch indexed like the items (sum of queue lengths <= n_tasks),
// so a queue of TaskQueue.Save's maximum 10,000 items needs no special case.
#pragma once

namespace evg {

constexpr int kDBlock = 256;
constexpr int kSegShort = 32;    // segments up to this length are ranked by one lane
constexpr int kSegChunk = 1024;  // keys of a long segment staged in LDS per pass

struct DispatchArgs {
  evg_plan_input in;
  const int32_t* item_off;
  const int32_t* item_row;
  evg_dispatch_order out;
  // scratch, int32: by task row
  int32_t* pos;  // queue index of the row, -1 when it is not in the persisted queue
  // by item
  int32_t *cnt, *beg, *cur;  // successors (dependents in the queue): count, list start within the distro's edge range, fill cursor
  int32_t *top, *bcnt, *bbeg, *m, *fbeg, *tmp, *own, *idx, *low, *onstk, *cstk, *sstk, *gtmp, *llist;
  // by edge
  int32_t *adj, *adj2;
  // by task group
  int32_t* gcur;
};

// Words other lanes update with atomics are read at L2.
__device__ __forceinline__ int32_t ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Exclusive prefix sums of get(0..n) -> put(q, prefix); every thread owns a contiguous chunk. Returns the total.
template <class Get, class Put>
__device__ __forceinline__ int block_scan_excl(int n, Get get, Put put, int* s_scan) {
  const int tid = threadIdx.x;
  const int per = (n + kDBlock - 1) / kDBlock;
  const int q0 = tid * per < n ? tid * per : n, q1 = q0 + per < n ? q0 + per : n;
  int sum = 0;
  for (int q = q0; q < q1; q++) sum += get(q);
  __syncthreads();
  s_scan[tid] = sum;
  __syncthreads();
  for (int o = 1; o < kDBlock; o <<= 1) {
    const int v = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  int run = tid ? s_scan[tid - 1] : 0;
  const int total = s_scan[kDBlock - 1];
  for (int q = q0; q < q1; q++) { const int v = get(q); put(q, run); run += v; }
  __syncthreads();
  return total;
}

// Sorts every segment s (src[sbeg(s) .. +slen(s)) -> dst, same places) by key(element) ascending, ties by position.
// Short segments: their owner lane ranks each element. Long ones are queued and ranked by the whole block with the keys
// staged through LDS.
template <class Beg, class Len, class Key>
__device__ __forceinline__ void segsort(int nseg, Beg sbeg, Len slen, const int32_t* src, int32_t* dst, Key key, int32_t* llist,
                                        int* s_n, uint64_t* s_keys) {
  const int tid = threadIdx.x;
  if (tid == 0) *s_n = 0;
  __syncthreads();
  for (int s = tid; s < nseg; s += kDBlock) {
    const int k = slen(s);
    if (k <= 0) continue;
    const int b = sbeg(s);
    if (k > kSegShort) { llist[atomicAdd(s_n, 1)] = s; continue; }
    for (int i = 0; i < k; i++) {
      const int32_t x = src[b + i];
      const uint64_t kx = key(x);
      int rank = 0;
      for (int j = 0; j < k; j++) {
        const uint64_t kj = key(src[b + j]);
        rank += (kj < kx || (kj == kx && j < i)) ? 1 : 0;
      }
      dst[b + rank] = x;
    }
  }
  __syncthreads();
  const int nl = *s_n;
  for (int l = 0; l < nl; l++) {
    const int s = llist[l], k = slen(s), b = sbeg(s);
    // ranks accumulate over the chunks; element i of this thread: i = tid, tid + kDBlock, ...
    for (int i0 = 0; i0 < k; i0 += kDBlock * 4) {      // four elements per thread per pass
      uint64_t kx[4];
      int32_t x[4];
      int rank[4] = {0, 0, 0, 0};
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * kDBlock + tid;
        x[u] = i < k ? src[b + i] : 0;
        kx[u] = i < k ? key(x[u]) : 0;
      }
      for (int c0 = 0; c0 < k; c0 += kSegChunk) {
        const int cn = k - c0 < kSegChunk ? k - c0 : kSegChunk;
        __syncthreads();
        for (int j = tid; j < cn; j += kDBlock) s_keys[j] = key(src[b + c0 + j]);
        __syncthreads();
        for (int j = 0; j < cn; j++) {
          const uint64_t kj = s_keys[j];
          for (int u = 0; u < 4; u++) {
            const int i = i0 + u * kDBlock + tid;
            rank[u] += (kj < kx[u] || (kj == kx[u] && c0 + j < i)) ? 1 : 0;
          }
        }
      }
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * kDBlock + tid;
        if (i < k) dst[b + rank[u]] = x[u];
      }
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kDBlock) k_dispatch_order(const DispatchArgs a) {
  __shared__ int s_scan[kDBlock];
  __shared__ uint64_t s_keys[kSegChunk];
  __shared__ int s_n, s_changed, s_ctr, s_cycles;
  const int tid = threadIdx.x;
  const evg_task_soa& t = a.in.tasks;
  for (int d = blockIdx.x; d < a.in.n_distros; d += gridDim.x) {
    const int lo = a.in.task_off[d], hi = a.in.task_off[d + 1];
    const int i0 = a.item_off[d], n = a.item_off[d + 1] - i0;
    const int g0 = a.in.tg_off[d], ng = a.in.tg_off[d + 1] - g0;
    const int eb = t.dep_off[lo];
    int32_t* cnt = a.cnt + i0; int32_t* beg = a.beg + i0; int32_t* cur = a.cur + i0; int32_t* top = a.top + i0;
    int32_t* bcnt = a.bcnt + i0; int32_t* bbeg = a.bbeg + i0; int32_t* m = a.m + i0; int32_t* fbeg = a.fbeg + i0;
    int32_t* tmp = a.tmp + i0; int32_t* own = a.own + i0; int32_t* idx = a.idx + i0; int32_t* low = a.low + i0;
    int32_t* onstk = a.onstk + i0; int32_t* llist = a.llist + i0;
    const int32_t* row = a.item_row + i0;
    int32_t* adj = a.adj + eb; int32_t* adj2 = a.adj2 + eb;
    int32_t* gcount = a.out.group_count + g0; int32_t* gstart = a.out.group_start + g0; int32_t* gcur = a.gcur + g0;

    // ---- nodes: queueIndex = position in the persisted queue (:161-164) ----
    for (int r = lo + tid; r < hi; r += kDBlock) a.pos[r] = -1;
    for (int g = tid; g < ng; g += kDBlock) { gcount[g] = 0; gcur[g] = 0; }
    if (tid == 0) { s_changed = 0; s_ctr = 0; s_cycles = 0; }
    __syncthreads();
    for (int q = tid; q < n; q += kDBlock) {
      a.pos[row[q]] = q;
      cnt[q] = 0; top[q] = q; bcnt[q] = 0; idx[q] = 0; onstk[q] = 0; m[q] = 0; own[q] = -1;
    }
    __syncthreads();
    // ---- lines dependency -> dependent for the dependencies that have a node (:197-204, addEdge :119-150); group sizes ----
    for (int q = tid; q < n; q += kDBlock) {
      const int r = row[q];
      for (int e = t.dep_off[r]; e < t.dep_off[r + 1]; e++) {
        const int j = t.dep_idx[e];
        if (j < lo || j >= hi) continue;
        const int p = a.pos[j];
        if (p >= 0) atomicAdd(&cnt[p], 1);
      }
      const int g = t.tg_key[r];
      if (g >= 0) atomicAdd(&a.out.group_count[g], 1);
    }
    __syncthreads();
    block_scan_excl(n, [&](int q) { return ld(&cnt[q]); }, [&](int q, int v) { beg[q] = v; cur[q] = v; }, s_scan);
    block_scan_excl(ng, [&](int g) { return ld(&gcount[g]); }, [&](int g, int v) { gstart[g] = i0 + v; }, s_scan);
    for (int q = tid; q < n; q += kDBlock) {
      const int r = row[q];
      for (int e = t.dep_off[r]; e < t.dep_off[r + 1]; e++) {
        const int j = t.dep_idx[e];
        if (j < lo || j >= hi) continue;
        const int p = a.pos[j];
        if (p >= 0) adj[atomicAdd(&cur[p], 1)] = q;
      }
      const int g = t.tg_key[r];
      if (g >= 0) a.gtmp[a.out.group_start[g] + atomicAdd(&a.gcur[g], 1)] = q;
    }
    __syncthreads();
    // successors in descending queue index (Reverse(order(From(id)))); a dependency listed twice is one successor twice,
    // which the search ignores the second time
    segsort(n, [&](int p) { return beg[p]; }, [&](int p) { return ld(&cnt[p]); }, adj, adj2,
            [](int32_t q) { return (uint64_t)(uint32_t)~q; }, llist, &s_n, s_keys);
    // d.taskGroups[id].tasks: queue order, then sort.SliceStable by GroupIndex (:166-195)
    segsort(ng, [&](int g) { return gstart[g]; }, [&](int g) { return ld(&gcount[g]); }, a.gtmp, a.out.group_items,
            [&](int32_t q) { return ((uint64_t)((uint32_t)t.task_group_order[row[q]] ^ 0x80000000u) << 32) | (uint32_t)q; }, llist, &s_n,
            s_keys);

    // ---- top(w): the largest queue index that reaches w ----
    for (;;) {
      for (int p = tid; p < n; p += kDBlock) {
        const int tp = ld(&top[p]);
        const int b = beg[p], e = b + ld(&cnt[p]);
        for (int x = b; x < e; x++) {
          const int w = adj2[x];
          if (ld(&top[w]) < tp) { atomicMax(&top[w], tp); s_changed = 1; }
        }
      }
      __syncthreads();
      const int ch = s_changed;
      __syncthreads();
      if (!ch) break;
      if (tid == 0) s_changed = 0;
      __syncthreads();
    }
    for (int q = tid; q < n; q += kDBlock) atomicAdd(&bcnt[ld(&top[q])], 1);
    __syncthreads();
    block_scan_excl(n, [&](int q) { return ld(&bcnt[q]); }, [&](int q, int v) { bbeg[q] = v; }, s_scan);

    // ---- one Tarjan search per root, a lane each (gonum graph/topo tarjan.strongconnect) ----
    {
      int r = -1, csp = 0, ssp = 0, k = 0, counter = 0, cyc = 0;
      int32_t *cs = nullptr, *ss = nullptr, *po = nullptr, *ow = nullptr;
      for (;;) {
        if (csp == 0) {
          if (r >= 0) { m[r] = k; r = -1; }
          const int q = atomicAdd(&s_ctr, 1);
          if (q >= n) break;
          if (ld(&top[q]) != q) continue;
          r = q;
          const int bb = bbeg[q];
          cs = a.cstk + i0 + bb; ss = a.sstk + i0 + bb; po = tmp + bb; ow = own + bb;
          k = 0; counter = 1; ssp = 0;
          idx[q] = 1; low[q] = 1; onstk[q] = 1; cur[q] = beg[q];
          ss[ssp++] = q; cs[csp++] = q;
          continue;
        }
        const int v = cs[csp - 1];
        const int c = cur[v];
        if (c < beg[v] + ld(&cnt[v])) {
          cur[v] = c + 1;
          const int w = adj2[c];
          if (ld(&top[w]) != r) continue;  // finished by the search of a larger root
          if (idx[w] == 0) {
            counter++;
            idx[w] = counter; low[w] = counter; onstk[w] = 1; cur[w] = beg[w];
            ss[ssp++] = w; cs[csp++] = w;
          } else if (onstk[w]) {
            const int iw = idx[w];
            if (iw < low[v]) low[v] = iw;
          }
        } else {
          csp--;
          const int lv = low[v];
          if (csp > 0) { const int u = cs[csp - 1]; if (lv < low[u]) low[u] = lv; }
          if (lv == idx[v]) {  // v is the root of a component: pop it
            int size = 0, w;
            do { w = ss[--ssp]; onstk[w] = 0; size++; } while (w != v);
            po[k] = size == 1 ? v : -1;  // SortStabilized: len(scc) != 1 -> a nil entry + one Unorderable
            ow[k] = r;
            k++;
            cyc += size != 1;
          }
        }
      }
      if (cyc) atomicAdd(&s_cycles, cyc);
    }
    __syncthreads();
    // ---- d.sorted: the blocks in ascending root order, each in reverse order of completion ----
    const int total = block_scan_excl(n, [&](int q) { return m[q]; }, [&](int q, int v) { fbeg[q] = v; }, s_scan);
    for (int s = tid; s < n; s += kDBlock) {
      const int r = own[s];
      if (r < 0) continue;
      const int k = s - bbeg[r];
      a.out.sorted[i0 + fbeg[r] + (m[r] - 1 - k)] = tmp[s];
    }
    if (tid == 0) { a.out.n_sorted[d] = total; a.out.n_cycles[d] = s_cycles; }
    __syncthreads();
  }
}

}  // namespace evg
