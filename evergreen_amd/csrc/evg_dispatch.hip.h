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
// One workgroup per distro. A queue whose items and dependency edges both fit the launch's LDS capacity (2048 or 4096,
// from the evg_plan_input.max_distro_tasks hint) keeps every word the loops hammer -- top, successor lists, the search's
// index / low-link / cursor words and both stacks -- in a nine-array LDS arena; anything larger (up to TaskQueue.Save's
// 10,000 items) runs the same code on a global scratch area indexed like the items.
#pragma once

namespace evg {

constexpr int kDBlock = 256;
constexpr int kSegShort = 32;    // segments up to this length are ranked by one lane
constexpr int kSegChunk = 512;   // keys of a long segment staged in LDS per pass
// LDS arena for a queue of up to cap items and 2*cap dependency edges: six int32 node arrays (top, beg, cnt, idx, low, cur)
// and three uint16 arrays of 2*cap entries (sorted successor lists, call stack, component stack)
constexpr int kArenaBytesPerItem = 6 * 4 + 3 * 2 * 2;

struct DispatchArgs {
  evg_plan_input in;
  const int32_t* item_off;
  const int32_t* item_row;
  evg_dispatch_order out;
  // scratch, int32: by task row
  int32_t* pos;  // queue index of the row, -1 when it is not in the persisted queue
  // by item
  int32_t *cnt, *beg, *cur;  // successors (dependents in the queue): count, list start within the distro's edge range, fill cursor
  int32_t *top, *bcnt, *bbeg, *m, *fbeg, *tmp, *own, *idx, *low, *cstk, *sstk, *gtmp, *llist;
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
template <class TS, class TD, class Beg, class Len, class Key>
__device__ __forceinline__ void segsort(int nseg, Beg sbeg, Len slen, const TS* src, TD* dst, Key key, int32_t* llist,
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
      const int32_t x = (int32_t)src[b + i];
      const uint64_t kx = key(x);
      int rank = 0;
      for (int j = 0; j < k; j++) {
        const uint64_t kj = key((int32_t)src[b + j]);
        rank += (kj < kx || (kj == kx && j < i)) ? 1 : 0;
      }
      dst[b + rank] = (TD)x;
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
        x[u] = i < k ? (int32_t)src[b + i] : 0;
        kx[u] = i < k ? key(x[u]) : 0;
      }
      for (int c0 = 0; c0 < k; c0 += kSegChunk) {
        const int cn = k - c0 < kSegChunk ? k - c0 : kSegChunk;
        __syncthreads();
        for (int j = tid; j < cn; j += kDBlock) s_keys[j] = key((int32_t)src[b + c0 + j]);
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
        if (i < k) dst[b + rank[u]] = (TD)x[u];
      }
    }
  }
  __syncthreads();
}

template <bool L>
__device__ __forceinline__ int32_t ldw(const int32_t* p) {
  if constexpr (L) return *p; else return ld(p);
}

// One distro. L: the hot words live in the LDS arena (`cap` words per array), else in the global scratch.
template <bool L>
__device__ __forceinline__ void dispatch_distro(const DispatchArgs& a, int d, int32_t* arena, int cap, int* s_scan, uint64_t* s_keys, int* s_n,
                                                int* s_changed, int* s_ctr, int* s_cycles) {
  const int tid = threadIdx.x;
  const evg_task_soa& t = a.in.tasks;
  const int lo = a.in.task_off[d], hi = a.in.task_off[d + 1];
  const int i0 = a.item_off[d], n = a.item_off[d + 1] - i0;
  const int g0 = a.in.tg_off[d], ng = a.in.tg_off[d + 1] - g0;
  const int eb = t.dep_off[lo];
  using ET = std::conditional_t<L, uint16_t, int32_t>;  // node ids inside the arena fit 16 bits
  int32_t *top, *beg, *cnt, *idx, *low, *cur, *gl, *bcnt, *bbl;
  ET *adj2, *cstk, *sstk, *adj;
  ET* ep = nullptr;
  int32_t* gix = nullptr;
  if constexpr (L) {
    top = arena; beg = arena + cap; cnt = arena + 2 * cap; idx = arena + 3 * cap; low = arena + 4 * cap; cur = arena + 5 * cap;
    ET* e0 = (ET*)(arena + 6 * cap);
    adj2 = e0; cstk = e0 + 2 * cap; sstk = e0 + 4 * cap;
    // before the searches start their words carry: the unsorted successor lists, each edge's dependency node (+1), the group
    // index and the unsorted group lists; the block sizes / block starts sit where a root's cursor / low link go once it is searched
    adj = cstk; ep = sstk; gix = idx; gl = low; bcnt = cur; bbl = low;
  } else {
    top = a.top + i0; beg = a.beg + i0; cnt = a.cnt + i0; adj2 = a.adj2 + eb; idx = a.idx + i0; low = a.low + i0; cur = a.cur + i0;
    cstk = a.cstk + i0; sstk = a.sstk + i0; adj = a.adj + eb; gl = a.gtmp + i0; bcnt = a.bcnt + i0; bbl = a.bbeg + i0;
  }
  int32_t* bbeg = a.bbeg + i0; int32_t* m = a.m + i0; int32_t* fbeg = a.fbeg + i0; int32_t* tmp = a.tmp + i0; int32_t* own = a.own + i0;
  int32_t* llist = a.llist + i0;
  const int32_t* row = a.item_row + i0;
  int32_t* gcount = a.out.group_count + g0; int32_t* gstart = a.out.group_start + g0; int32_t* gcur = a.gcur + g0;

  // ---- nodes: queueIndex = position in the persisted queue (:161-164) ----
  for (int r = lo + tid; r < hi; r += kDBlock) a.pos[r] = -1;
  for (int g = tid; g < ng; g += kDBlock) { gcount[g] = 0; gcur[g] = 0; }
  if (tid == 0) { *s_changed = 0; *s_ctr = 0; *s_cycles = 0; }
  __syncthreads();
  for (int q = tid; q < n; q += kDBlock) {
    a.pos[row[q]] = q;
    cnt[q] = 0; top[q] = q; m[q] = 0; own[q] = -1;
  }
  __syncthreads();
  // ---- lines dependency -> dependent for the dependencies that have a node (:197-204, addEdge :119-150); group sizes ----
  // by ROW (coalesced offsets), skipping the rows the cut left out of the queue
  for (int r = lo + tid; r < hi; r += kDBlock) {
    const int q = a.pos[r];
    if (q < 0) continue;
    for (int e = t.dep_off[r]; e < t.dep_off[r + 1]; e++) {
      const int j = t.dep_idx[e];
      const int p = j >= lo && j < hi ? a.pos[j] : -1;
      if constexpr (L) ep[e - eb] = (ET)(p + 1);
      if (p >= 0) atomicAdd(&cnt[p], 1);
    }
    const int g = t.tg_key[r];
    if (g >= 0) atomicAdd(&a.out.group_count[g], 1);
    if constexpr (L) gix[q] = t.task_group_order[r];
  }
  __syncthreads();
  block_scan_excl(n, [&](int q) { return ldw<L>(&cnt[q]); }, [&](int q, int v) { beg[q] = v; cur[q] = v; }, s_scan);
  block_scan_excl(ng, [&](int g) { return ld(&gcount[g]); }, [&](int g, int v) { gstart[g] = i0 + v; }, s_scan);
  for (int r = lo + tid; r < hi; r += kDBlock) {
    const int q = a.pos[r];
    if (q < 0) continue;
    for (int e = t.dep_off[r]; e < t.dep_off[r + 1]; e++) {
      int p;
      if constexpr (L) {
        p = (int)ep[e - eb] - 1;
      } else {
        const int j = t.dep_idx[e];
        p = j >= lo && j < hi ? a.pos[j] : -1;
      }
      if (p >= 0) adj[atomicAdd(&cur[p], 1)] = (ET)q;
    }
    const int g = t.tg_key[r];
    if (g >= 0) gl[a.out.group_start[g] - i0 + atomicAdd(&a.gcur[g], 1)] = q;
  }
  __syncthreads();
  // successors in descending queue index (Reverse(order(From(id)))); a dependency listed twice is one successor twice,
  // which the search ignores the second time
  segsort(n, [&](int p) { return beg[p]; }, [&](int p) { return ldw<L>(&cnt[p]); }, adj, adj2,
          [](int32_t q) { return (uint64_t)(uint32_t)~q; }, llist, s_n, s_keys);
  // d.taskGroups[id].tasks: queue order, then sort.SliceStable by GroupIndex (:166-195)
  segsort(ng, [&](int g) { return gstart[g] - i0; }, [&](int g) { return ld(&gcount[g]); }, gl, a.out.group_items + i0,
          [&](int32_t q) {
            int32_t gi;
            if constexpr (L) gi = gix[q]; else gi = t.task_group_order[row[q]];
            return ((uint64_t)((uint32_t)gi ^ 0x80000000u) << 32) | (uint32_t)q;
          },
          llist, s_n, s_keys);
  for (int q = tid; q < n; q += kDBlock) { idx[q] = 0; bcnt[q] = 0; }
  __syncthreads();

  // ---- top(w): the largest queue index that reaches w ----
  for (;;) {
    for (int p = tid; p < n; p += kDBlock) {
      const int tp = ldw<L>(&top[p]);
      const int b = beg[p], e = b + ldw<L>(&cnt[p]);
      for (int x = b; x < e; x++) {
        const int w = (int)adj2[x];
        if (ldw<L>(&top[w]) < tp) { atomicMax(&top[w], tp); *s_changed = 1; }
      }
    }
    __syncthreads();
    const int ch = *s_changed;
    __syncthreads();
    if (!ch) break;
    if (tid == 0) *s_changed = 0;
    __syncthreads();
  }
  for (int q = tid; q < n; q += kDBlock) atomicAdd(&bcnt[ldw<L>(&top[q])], 1);
  __syncthreads();
  block_scan_excl(n, [&](int q) { return ldw<L>(&bcnt[q]); }, [&](int q, int v) { bbeg[q] = v; if constexpr (L) bbl[q] = v; }, s_scan);

  // ---- one Tarjan search per root, a lane each (gonum graph/topo tarjan.strongconnect) ----
  // A block of one node needs no search: whatever its successors are, they belong to larger roots.
  // idx: 0 unvisited, -index while the node is on the component stack, +index once its component is popped.
  {
    int r = -1, csp = 0, ssp = 0, k = 0, counter = 0, cyc = 0;
    int v = -1, c = 0, cend = 0, lowv = 0;  // the node on top of the call stack, its cursor / end of list / low link
    ET *cs = nullptr, *ss = nullptr;
    int32_t *po = nullptr, *ow = nullptr;
    for (;;) {
      if (csp == 0) {
        if (r >= 0) { m[r] = k; r = -1; }
        const int q = atomicAdd(s_ctr, 1);
        if (q >= n) break;
        if (ldw<L>(&top[q]) != q) continue;
        const int bb = bbl[q];
        if (ldw<L>(&bcnt[q]) == 1) { tmp[bb] = q; own[bb] = q; m[q] = 1; continue; }
        r = q;
        cs = cstk + bb; ss = sstk + bb; po = tmp + bb; ow = own + bb;
        k = 0; counter = 1; ssp = 0;
        idx[q] = -1;
        ss[ssp++] = (ET)q; cs[csp++] = (ET)q;
        v = q; c = beg[q]; cend = c + ldw<L>(&cnt[q]); lowv = 1;
        continue;
      }
      if (c < cend) {
        const int w = (int)adj2[c++];
        if (ldw<L>(&top[w]) != r) continue;  // finished by the search of a larger root
        const int iw = idx[w];
        if (iw == 0) {
          cur[v] = c; low[v] = lowv;         // park the frame
          counter++;
          idx[w] = -counter;
          ss[ssp++] = (ET)w; cs[csp++] = (ET)w;
          v = w; c = beg[w]; cend = c + ldw<L>(&cnt[w]); lowv = counter;
        } else if (iw < 0) {
          if (-iw < lowv) lowv = -iw;
        }
      } else {
        csp--;
        if (lowv == -idx[v]) {  // v is the root of a component: pop it
          int size = 0, w;
          do { w = (int)ss[--ssp]; idx[w] = -idx[w]; size++; } while (w != v);
          po[k] = size == 1 ? v : -1;  // SortStabilized: len(scc) != 1 -> a nil entry + one Unorderable
          ow[k] = r;
          k++;
          cyc += size != 1;
        }
        if (csp > 0) {
          const int u = (int)cs[csp - 1];
          const int lu = low[u];
          v = u; c = cur[u]; cend = beg[u] + ldw<L>(&cnt[u]);
          lowv = lowv < lu ? lowv : lu;
        }
      }
    }
    if (cyc) atomicAdd(s_cycles, cyc);
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
  if (tid == 0) { a.out.n_sorted[d] = total; a.out.n_cycles[d] = *s_cycles; }
  __syncthreads();
}

// cap: items the LDS arena holds (with 2*cap edges); dynamic LDS = kArenaBytesPerItem * cap.
__global__ void __launch_bounds__(kDBlock) k_dispatch_order(const DispatchArgs a, int cap) {
  extern __shared__ int32_t arena[];
  __shared__ int s_scan[kDBlock];
  __shared__ uint64_t s_keys[kSegChunk];
  __shared__ int s_n, s_changed, s_ctr, s_cycles;
  for (int d = blockIdx.x; d < a.in.n_distros; d += gridDim.x) {
    const int lo = a.in.task_off[d], hi = a.in.task_off[d + 1];
    const int n = a.item_off[d + 1] - a.item_off[d];
    const int ecap = a.in.tasks.dep_off[hi] - a.in.tasks.dep_off[lo];
    if (n <= cap && ecap <= 2 * cap) dispatch_distro<true>(a, d, arena, cap, s_scan, s_keys, &s_n, &s_changed, &s_ctr, &s_cycles);
    else dispatch_distro<false>(a, d, nullptr, 0, s_scan, s_keys, &s_n, &s_changed, &s_ctr, &s_cycles);
  }
}

}  // namespace evg
