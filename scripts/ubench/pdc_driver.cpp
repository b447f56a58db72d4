// pdc_driver.cpp -- the timed loop of bench.py's `per_distro_calls` in native threads.
//
// The reference calls its TaskPlanner and its HostAllocator once per distro from concurrent amboy jobs (units/crons.go:303-332,
// units/scheduler.go:48-49): goroutines on OS threads, no interpreter lock between them. Measured from Python threads, a worker that
// returns from its C call waits for the GIL behind whichever thread is doing bookkeeping (switch interval 5 ms): with 32 threads the
// p99 of a 150 us call pair read 15.9 ms, an artefact of the harness. Here every worker is a std::thread that owns one context
// and makes the same two C-ABI calls per distro -- evg_plan_distros + evg_allocate_hosts on a batch of one -- through the function
// pointers bench.py hands over (this file links nothing: no HIP, no libevg_sched).
//   g++ -O2 -shared -fPIC -pthread pdc_driver.cpp -o libpdc.so
#include <atomic>
#include <chrono>
#include <cstddef>
#include <thread>
#include <vector>

// The workers exist before the clock starts (the reference's amboy workers are a pool, not spawned per job): they are created, wait at a
// gate, and the wall clock runs from the moment the gate opens to the last join. (Until round 6 the threads were spawned inside the timed
// region: ~20 us each -- 1.3 ms of a 64-thread run, 10 ms of a 512-thread one.)
template <class Work>
static double run_gated(int n_threads, Work work) {
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> th;
  for (int w = 0; w < n_threads; w++)
    th.emplace_back([&, w] {
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      work(w);
    });
  while (ready.load() < n_threads) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& t : th) t.join();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

typedef int (*call2_fn)(void* ctx, const void* in, const void* out);
typedef int (*call4_fn)(void* batcher, const void* in, const void* out, char* err, int err_len);  // evg_batcher_plan / evg_batcher_allocate

extern "C" int pdc_run(void* fn_plan, void* fn_alloc, void** ctxs, int n_threads, int n_distros, const char* pin, size_t pin_stride, const char* pout,
                       size_t pout_stride, const char* ain, size_t ain_stride, const char* aout, size_t aout_stride, double* lat_us, double* wall_ms) {
  const call2_fn plan = (call2_fn)fn_plan, alloc = (call2_fn)fn_alloc;
  std::vector<int> bad(n_threads, 0);
  auto work = [&](int w) {
    for (int d = w; d < n_distros; d += n_threads) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = plan(ctxs[w], pin + (size_t)d * pin_stride, pout + (size_t)d * pout_stride);
      const int rc2 = alloc(ctxs[w], ain + (size_t)d * ain_stride, aout + (size_t)d * aout_stride);
      lat_us[d] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      bad[w] += (rc != 0) + (rc2 != 0);
    }
  };
  *wall_ms = run_gated(n_threads, work);
  int errors = 0;
  for (int b : bad) errors += b;
  return errors;
}

// The same loop through the micro-batching front (ABI 3.2): every worker calls evg_batcher_plan + evg_batcher_allocate on the ONE shared
// batcher -- what shim/gpu_planner.go's runGPUPlanner / GPUHostAllocator do from the reference's concurrent per-distro jobs.
extern "C" int pdc_run_batcher(void* fn_plan, void* fn_alloc, void* batcher, int n_threads, int n_distros, const char* pin, size_t pin_stride,
                               const char* pout, size_t pout_stride, const char* ain, size_t ain_stride, const char* aout, size_t aout_stride,
                               double* lat_us, double* wall_ms) {
  const call4_fn plan = (call4_fn)fn_plan, alloc = (call4_fn)fn_alloc;
  std::vector<int> bad(n_threads, 0);
  auto work = [&](int w) {
    char err[256];
    for (int d = w; d < n_distros; d += n_threads) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = plan(batcher, pin + (size_t)d * pin_stride, pout + (size_t)d * pout_stride, err, (int)sizeof err);
      const int rc2 = alloc(batcher, ain + (size_t)d * ain_stride, aout + (size_t)d * aout_stride, err, (int)sizeof err);
      lat_us[d] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      bad[w] += (rc != 0) + (rc2 != 0);
    }
  };
  *wall_ms = run_gated(n_threads, work);
  int errors = 0;
  for (int b : bad) errors += b;
  return errors;
}

// ABI 3.3: one evg_batcher_schedule call per distro (plan + allocate as ONE request), the queue named by `queue_base + d` and the
// generation given: a second run with the same generation finds every queue resident on the device and uploads clock readings only.
// queue_base 0: no resident queues.
typedef int (*sched_fn)(void* batcher, unsigned long long queue_id, unsigned long long generation, const void* in, const void* out, const void* ain,
                        const void* aout, char* err, int err_len);
extern "C" int pdc_run_pairs(void* fn_schedule, void* batcher, int n_threads, int n_distros, unsigned long long queue_base, unsigned long long generation,
                             const char* pin, size_t pin_stride, const char* pout, size_t pout_stride, const char* ain, size_t ain_stride, const char* aout,
                             size_t aout_stride, double* lat_us, double* wall_ms) {
  const sched_fn schedule = (sched_fn)fn_schedule;
  std::vector<int> bad(n_threads, 0);
  auto work = [&](int w) {
    char err[256];
    for (int d = w; d < n_distros; d += n_threads) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = schedule(batcher, queue_base ? queue_base + (unsigned long long)d : 0ull, generation, pin + (size_t)d * pin_stride, pout + (size_t)d * pout_stride,
                              ain + (size_t)d * ain_stride, aout + (size_t)d * aout_stride, err, (int)sizeof err);
      lat_us[d] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      bad[w] += rc != 0;
    }
  };
  *wall_ms = run_gated(n_threads, work);
  int errors = 0;
  for (int b : bad) errors += b;
  return errors;
}
