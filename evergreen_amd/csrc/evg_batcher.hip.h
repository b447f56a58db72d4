// evg_batcher.hip.h -- the HIP backend of the micro-batching front for PER-DISTRO callers (evg_batcher_*, ABI 3.2 / 3.3).
//
// The state machine -- joining, closing, packing, the segment table, fan-out, the queue cache, the retirement of slots whose device
// wait outlived the deadline -- is evg_batcher_core.hpp: plain C++ that also builds, unchanged, against a CPU backend under
// ThreadSanitizer (tests/cpp/test_batcher_tsan.cpp). This file is what a batch does on the DEVICE: one copy in, one kernel that
// moves (and re-bases) the members' column stretches to their place in the batch's columns, the ordinary planner and / or allocator
// launches over the whole batch with a clock reading per distro (PlanArgs.now_d, AllocArgs.tick_d), one copy back, one bounded wait.
#pragma once

#include "evg_batcher_core.hpp"

namespace evgb {

static_assert(sizeof(Tick) == sizeof(evg::AllocTick) && offsetof(Tick, lpp_running) == offsetof(evg::AllocTick, lpp_running), "the allocator's per-distro tick row");

// One workgroup per (segment, 32 KB chunk of it).
__global__ void __launch_bounds__(256) k_batch_segments(const Seg* segs, int n) {
  const int s = blockIdx.x;
  if (s >= n) return;
  const Seg g = segs[s];
  const uint32_t c0 = blockIdx.y * 32768u;
  if (c0 >= g.bytes) return;
  const uint32_t len = g.bytes - c0 < 32768u ? g.bytes - c0 : 32768u;
  const unsigned char* src = (const unsigned char*)g.src + c0;
  unsigned char* dst = (unsigned char*)g.dst + c0;
  if (g.mode != SEG_COPY) {  // int32 elements re-based into the batch's numbering (SEG_ADD_NONNEG: -1 and below stay)
    const bool all = g.mode == SEG_ADD;
    for (uint32_t i = threadIdx.x; i < len / 4; i += 256) {
      const int32_t v = ((const int32_t*)src)[i];
      ((int32_t*)dst)[i] = all || v >= 0 ? v + g.add : v;
    }
  }
  else if (g.esz == 16) { for (uint32_t i = threadIdx.x; i < len / 16; i += 256) ((uint4*)dst)[i] = ((const uint4*)src)[i]; }
  else if (g.esz == 8) { for (uint32_t i = threadIdx.x; i < len / 8; i += 256) ((uint64_t*)dst)[i] = ((const uint64_t*)src)[i]; }
  else if (g.esz == 4) { for (uint32_t i = threadIdx.x; i < len / 4; i += 256) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i]; }
  else if (g.esz == 2) { for (uint32_t i = threadIdx.x; i < len / 2; i += 256) ((uint16_t*)dst)[i] = ((const uint16_t*)src)[i]; }
  else { for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i]; }
}

struct HipBackend {
  struct Dev {
    evg_ctx* ctx = nullptr;
    DevBuf arena;
  };
  static Dev* dev_create(int device) {
    evg_ctx* c = evg_create(device);
    if (!c) return nullptr;
    Dev* d = new Dev();
    d->ctx = c;
    return d;
  }
  static void dev_destroy(Dev* d) {
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    bool idle = true;
    if (d->ctx->timed_out) {  // like evg_destroy: one more bounded wait, then the arena is leaked rather than the thread held
      d->ctx->timed_out = false;
      idle = wait_stream(d->ctx, d->ctx->stream, "evg_batcher_destroy") == EVG_OK;
      d->ctx->timed_out = !idle;
    }
    // (hipFree waits for the whole device: not behind a hang on another object's stream either -- evgreg; leaked otherwise)
    if (idle && d->arena.p && evgreg::quiesced_within(d->ctx->device, d->ctx->deadline_ms)) (void)hipFree(d->arena.p);
    evg_destroy(d->ctx);
    delete d;
  }
  static const char* dev_error(Dev* d) { return d->ctx->err.c_str(); }
  static void dev_set_deadline(Dev* d, int64_t ms) { if (d) (void)evg_set_deadline_ms(d->ctx, ms); }
  static int dev_debug_stall(Dev* d, int32_t ms) { return evg_debug_stall(d->ctx, ms); }
  static void* host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
  }
  // hipHostFree / hipFree wait for the whole device, without a limit: only behind evgreg's bounded look at the device's streams (the
  // batcher's own contexts have been destroyed -- or leaked -- by then); a block behind a hang is leaked with it
  static int64_t free_deadline_ms() {
    static const int64_t v = [] { const char* e = getenv("EVG_DEADLINE_MS"); const long long x = e ? atoll(e) : 30000; return (int64_t)(x >= 0 ? x : 30000); }();
    return v;
  }
  static void host_free(void* p) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || !evgreg::quiesced_within(dev, free_deadline_ms())) return;
    (void)hipHostFree(p);
  }
  static void* cache_alloc(int device, size_t bytes) {
    void* p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
  }
  static void cache_free(int device, void* p) {
    (void)hipSetDevice(device);
    if (!evgreg::quiesced_within(device, free_deadline_ms())) return;
    (void)hipFree(p);
  }
  static int launch_hints(const evg_plan_input* in, int32_t* a, int32_t* b, int32_t* c) { return evg_plan_launch_hints(in, a, b, c); }
  static int direct_plan(Dev* d, const evg_plan_input* in, const evg_plan_output* out) { return evg_plan_distros(d->ctx, in, out); }
  static int direct_alloc(Dev* d, const evg_alloc_input* in, const evg_alloc_output* out) { return evg_allocate_hosts(d->ctx, in, out); }
  static int arena(Dev* d, size_t bytes, unsigned char** A) {
    evg_ctx* c = d->ctx;
    std::lock_guard<std::mutex> ck(c->mu);
    if (c->timed_out) return refuse_timed_out(c);
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure(c, d->arena, bytes)) return rc;
    *A = (unsigned char*)d->arena.p;
    return EVG_OK;
  }
  // Whatever way the device work of a batch ends, the stream is drained (within the deadline) before the slot's blocks are handed to
  // the next batch, and a status the kernels left is taken: the context serves the next batch clean.
  struct Drain {
    evg_ctx* c;
    bool armed = true;
    ~Drain() {
      if (!armed || c->timed_out) return;
      const std::string keep = c->err;
      if (wait_stream(c, c->stream, "batch") == EVG_OK) c->err = keep;
      if (c->status_word) *(volatile uint32_t*)c->status_word = 0;
    }
  };
  static int run(Dev* d, const Launch& L) {
    using namespace evg;
    evg_ctx* c = d->ctx;
    std::lock_guard<std::mutex> ck(c->mu);
    if (c->timed_out) return refuse_timed_out(c);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    Drain drain{c};
    HIP_TRY(c, hipMemcpyAsync(L.A, L.h_in, L.up_bytes, hipMemcpyHostToDevice, st));
    if (L.zero_bytes) HIP_TRY(c, hipMemsetAsync(L.A + L.zero_off, 0, L.zero_bytes, st));
    hipLaunchKernelGGL(k_batch_segments, dim3(L.n_segs, L.seg_rows), dim3(256), 0, st, (const Seg*)(L.A + L.seg_off), (int)L.n_segs);
    HIP_TRY(c, hipGetLastError());
    if (L.kind != K_ALLOC) {
      c->now_d = L.now_d;
      const int rc = launch_plan(c, &L.plan_in, &L.plan_out, st);
      c->now_d = nullptr;
      if (rc) return rc;
    }
    if (L.kind != K_PLAN) {
      c->tick_d = L.tick_d;
      const int rc = launch_alloc(c, &L.alloc_in, &L.alloc_out, st);
      c->tick_d = nullptr;
      if (rc) return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(L.h_out, L.A + L.out_base, L.out_bytes, hipMemcpyDeviceToHost, st));
    if (int rc = wait_stream(c, st, "batch")) return rc;
    drain.armed = false;
    if (int rc2 = pending_status(c)) {  // a promise the members' own hints made cannot be false; reported all the same
      *(volatile uint32_t*)c->status_word = 0;
      return rc2;
    }
    return EVG_OK;
  }
};

}  // namespace evgb

EVGB_DEFINE_C_API(evgb::HipBackend)
