// Context, memory, streams, events, profiling brackets.
#include <time.h>
#include "common.h"

static thread_local char g_err[512] = "";

int mgr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {

int mgr_version(void) { return 100; }
const char* mgr_last_error(void) { return g_err; }

int mgr_device_count(int* n) {
  MGR_REQUIRE(n, "null out pointer");
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) {
    *n = 0;
    return mgr_fail(-2, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return 0;
}

int mgr_ctx_create(int device, mgr_ctx** out) {
  MGR_REQUIRE(out, "null out pointer");
  int n = 0;
  MGR_HIP(hipGetDeviceCount(&n));
  MGR_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
  MGR_HIP(hipSetDevice(device));
  mgr_ctx* c = new mgr_ctx();
  memset(c, 0, sizeof(*c));
  c->device = device;
  hipDeviceProp_t prop;
  MGR_HIP(hipGetDeviceProperties(&prop, device));
  c->cu_count = prop.multiProcessorCount;
  c->hbm_bytes = prop.totalGlobalMem;
  snprintf(c->name, sizeof(c->name), "%s/%s", prop.name, prop.gcnArchName);
  for (int i = 0; i < MGR_NUM_STREAMS; ++i) MGR_HIP(hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking));
  for (int i = 0; i < MGR_NUM_EVENTS; ++i) MGR_HIP(hipEventCreate(&c->events[i]));
  for (int i = 0; i < 64; ++i) MGR_HIP(hipEventCreateWithFlags(&c->xev[i], hipEventDisableTiming));
  MGR_HIP(hipHostMalloc(&c->h_small_pinned, MGR_SMALL_D2H, hipHostMallocDefault));
  MGR_HIP(hipMalloc(&c->sticky_status, 256));
  MGR_HIP(hipMemset(c->sticky_status, 0, 256));
  *out = c;
  return 0;
}

int mgr_ctx_destroy(mgr_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (int i = 0; i < MGR_NUM_STREAMS; ++i) hipStreamDestroy(c->streams[i]);
  for (int i = 0; i < MGR_NUM_EVENTS; ++i) hipEventDestroy(c->events[i]);
  for (int i = 0; i < 64; ++i) hipEventDestroy(c->xev[i]);
  if (c->sticky_status) hipFree(c->sticky_status);
  if (c->h_small_pinned) hipHostFree(c->h_small_pinned);
  for (int i = 0; i < MGR_MAX_PERSIST; ++i)
    if (c->persist[i].done) hipEventDestroy(c->persist[i].done);
  for (int f = 0; f < MGR_K_COUNT; ++f) {
    for (int i = 0; i < c->prof_cap[f]; ++i) {
      hipEventDestroy(c->prof_pairs[f][i].a);
      hipEventDestroy(c->prof_pairs[f][i].b);
    }
    delete[] c->prof_pairs[f];
  }
  delete c;
  return 0;
}

int mgr_device_info(mgr_ctx* c, int* cu_count, size_t* hbm_bytes, char* name, int name_len) {
  MGR_REQUIRE(c, "null ctx");
  if (cu_count) *cu_count = c->cu_count;
  if (hbm_bytes) *hbm_bytes = c->hbm_bytes;
  if (name && name_len > 0) snprintf(name, name_len, "%s", c->name);
  return 0;
}

int mgr_alloc(mgr_ctx* c, size_t bytes, void** dptr) {
  MGR_REQUIRE(c && dptr, "null argument");
  MGR_HIP(hipSetDevice(c->device));
  if (bytes == 0) bytes = 16;
  MGR_HIP(hipMalloc(dptr, bytes));
  return 0;
}

int mgr_free(mgr_ctx* c, void* dptr) {
  MGR_REQUIRE(c, "null ctx");
  if (!dptr) return 0;
  MGR_HIP(hipSetDevice(c->device));
  if (dptr == c->status_bound) c->status_bound = nullptr;   // a freed status block is not reported into any more
  if (dptr == c->gate_flag) c->gate_flag = nullptr;
  for (int i = 0; i < MGR_MAX_FROZEN; ++i) {   // (cached weight planes: neither the weights nor the workspace outlive their buffer)
    if (c->frozen_w[i] == dptr) c->frozen_w[i] = nullptr;
    if (c->planes[i].Wp == dptr || c->planes[i].ws == dptr) c->planes[i] = mgr_ctx::PlaneEntry{nullptr, nullptr, 0, 0};
  }
  MGR_HIP(hipFree(dptr));
  return 0;
}

int mgr_memset(mgr_ctx* c, void* d, int byte, size_t n) {
  MGR_REQUIRE(c && d, "null argument");
  MGR_HIP(hipMemsetAsync(d, byte, n, mgr_stream(c)));
  return 0;
}

int mgr_h2d(mgr_ctx* c, void* d, const void* h, size_t n) {
  MGR_REQUIRE(c && d && h, "null argument");
  MGR_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, mgr_stream(c)));
  // pageable host memory: make the call synchronous w.r.t. the host buffer (caller may free it)
  MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
  return 0;
}

int mgr_host_alloc(mgr_ctx* c, size_t bytes, void** out) {
  MGR_REQUIRE(c && out && bytes > 0, "bad argument");
  MGR_HIP(hipSetDevice(c->device));
  MGR_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return 0;
}

int mgr_host_free(mgr_ctx* c, void* p) {
  MGR_REQUIRE(c, "null ctx");
  if (p) MGR_HIP(hipHostFree(p));
  return 0;
}

int mgr_h2d_async(mgr_ctx* c, void* d, const void* h_pinned, size_t n) {
  MGR_REQUIRE(c && d && h_pinned, "null argument");
  // h_pinned must come from mgr_host_alloc and stay untouched until the stream has passed this copy
  MGR_HIP(hipMemcpyAsync(d, h_pinned, n, hipMemcpyHostToDevice, mgr_stream(c)));
  return 0;
}

// host-side polling: spin for the first ~100 us (what a short kernel takes), then sleep 50 us between polls - a rank that waits
// 25 ms for its step's loss must not burn a core of a CPU-quota'd box (profiles/r03_host_stalls.txt) for it
static inline void mgr_poll_backoff(unsigned spins) {
  if (spins < 2000) {
    __builtin_ia32_pause();
  } else {
    struct timespec ts = {0, 50000};
    nanosleep(&ts, nullptr);
  }
}

int mgr_d2h(mgr_ctx* c, void* h, const void* d, size_t n) {
  MGR_REQUIRE(c && d && h, "null argument");
  if (n <= MGR_SMALL_D2H && c->h_small_pinned) {
    // Small read-backs (the loss, a status block) are what the host waits for once per step.  A copy into pageable memory
    // waits inside the runtime, whose blocked wait now and then wakes 20-45 ms late (round 3: one step in ten of a 5 ms
    // step); a copy into page-locked scratch is asynchronous and the host polls the stream itself.
    hipStream_t st = mgr_stream(c);
    MGR_HIP(hipMemcpyAsync(c->h_small_pinned, d, n, hipMemcpyDeviceToHost, st));
    for (unsigned spins = 0;; ++spins) {
      hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) MGR_HIP(q);
      mgr_poll_backoff(spins);
    }
    memcpy(h, c->h_small_pinned, n);
    return 0;
  }
  MGR_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, mgr_stream(c)));
  MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
  return 0;
}

int mgr_d2h_async(mgr_ctx* c, void* h_pinned, const void* d, size_t n) {
  MGR_REQUIRE(c && d && h_pinned, "null argument");
  // h_pinned must come from mgr_host_alloc; its contents are valid once an event recorded behind this call has completed
  MGR_HIP(hipMemcpyAsync(h_pinned, d, n, hipMemcpyDeviceToHost, mgr_stream(c)));
  return 0;
}

int mgr_event_sync(mgr_ctx* c, int ev) {
  MGR_REQUIRE(c && ev >= 0 && ev < MGR_NUM_EVENTS, "bad event index");
  // polled, not hipEventSynchronize: what the host waits for here (a loss, a decoded batch) gates the next step's enqueue, and a
  // blocked wait of the runtime wakes late now and then
  for (unsigned spins = 0;; ++spins) {
    hipError_t q = hipEventQuery(c->events[ev]);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) MGR_HIP(q);
    mgr_poll_backoff(spins);
  }
  return 0;
}

int mgr_d2d(mgr_ctx* c, void* dst, const void* src, size_t n) {
  MGR_REQUIRE(c && dst && src, "null argument");
  MGR_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, mgr_stream(c)));
  return 0;
}

int mgr_sync(mgr_ctx* c) {
  MGR_REQUIRE(c, "null ctx");
  for (int i = 0; i < MGR_NUM_STREAMS; ++i) MGR_HIP(hipStreamSynchronize(c->streams[i]));
  return 0;
}

int mgr_stream_set(mgr_ctx* c, int idx) {
  MGR_REQUIRE(c, "null ctx");
  MGR_REQUIRE(idx >= 0 && idx < MGR_NUM_STREAMS, "stream index %d out of range", idx);
  c->cur = idx;
  return 0;
}

int mgr_stream_set_priority(mgr_ctx* c, int idx, int level) {
  MGR_REQUIRE(c && idx >= 0 && idx < MGR_NUM_STREAMS && level >= -1 && level <= 1, "bad stream index / level (-1 low, 0 default, 1 high)");
  MGR_HIP(hipSetDevice(c->device));
  int least = 0, greatest = 0;   // (numerically: least >= greatest; lower numbers are higher priorities)
  MGR_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const int prio = level > 0 ? greatest : level < 0 ? least : (least + greatest) / 2;
  MGR_HIP(hipStreamSynchronize(c->streams[idx]));
  hipStream_t s;
  MGR_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
  MGR_HIP(hipStreamDestroy(c->streams[idx]));
  c->streams[idx] = s;
  return 0;
}

int mgr_stream_wait(mgr_ctx* c, int waiter, int waited) {
  MGR_REQUIRE(c, "null ctx");
  MGR_REQUIRE(waiter >= 0 && waiter < MGR_NUM_STREAMS && waited >= 0 && waited < MGR_NUM_STREAMS, "bad stream index");
  if (waiter == waited) return 0;
  hipEvent_t ev = c->xev[c->xev_next];
  c->xev_next = (c->xev_next + 1) & 63;
  MGR_HIP(hipEventRecord(ev, c->streams[waited]));
  MGR_HIP(hipStreamWaitEvent(c->streams[waiter], ev, 0));
  return 0;
}

int mgr_stream_wait_event(mgr_ctx* c, int waiter, int ev) {
  MGR_REQUIRE(c && ev >= 0 && ev < MGR_NUM_EVENTS, "bad event index");
  MGR_REQUIRE(waiter >= 0 && waiter < MGR_NUM_STREAMS, "bad stream index");
  MGR_HIP(hipStreamWaitEvent(c->streams[waiter], c->events[ev], 0));
  return 0;
}

int mgr_scan_status(mgr_ctx* c, unsigned* out) {
  MGR_REQUIRE(c && out, "null argument");
  // on the CURRENT stream: the caller decides what it is ordered after
  MGR_HIP(hipMemcpyAsync(out, mgr_status_block(c), sizeof(unsigned), hipMemcpyDeviceToHost, mgr_stream(c)));
  MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
  // MGR_SCAN_NONFINITE alone is not an error of the library: the outputs carry the NaN, like the reference's would
  MGR_REQUIRE((*out & ~(unsigned)MGR_SCAN_NONFINITE) == 0,
              "a persistent scan gave up on a bounded spin (code %u): its outputs are invalid", *out);
  return 0;
}

int mgr_scan_status_ex(mgr_ctx* c, unsigned out[4]) {
  MGR_REQUIRE(c && out, "null argument");
  MGR_HIP(hipMemcpyAsync(out, mgr_status_block(c), 4 * sizeof(unsigned), hipMemcpyDeviceToHost, mgr_stream(c)));
  MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
  out[1] = 0;   // (the resident word of the context's own block is not part of this interface)
  out[3] = 0;
  return 0;
}

int mgr_scan_status_clear(mgr_ctx* c) {
  MGR_REQUIRE(c, "null ctx");
  unsigned* blk = mgr_status_block(c);
  MGR_HIP(hipMemsetAsync(blk, 0, sizeof(unsigned), mgr_stream(c)));
  MGR_HIP(hipMemsetAsync(blk + 2, 0, sizeof(unsigned), mgr_stream(c)));
  MGR_HIP(hipMemsetAsync(blk + 8, 0, 8 * sizeof(unsigned), mgr_stream(c)));   // the per-sample non-finite bits
  return 0;
}

int mgr_scan_status_bind(mgr_ctx* c, void* block) {
  MGR_REQUIRE(c, "null ctx");
  MGR_REQUIRE((reinterpret_cast<uintptr_t>(block) & 15) == 0, "status block must be 16-byte aligned");
  c->status_bound = reinterpret_cast<unsigned*>(block);
  return 0;
}

namespace {
__global__ void k_gate_eval(const unsigned* status, unsigned mask, float* flag) { flag[0] = (status[0] & mask) ? 1.f : 0.f; }
__global__ void k_status_or(unsigned* status, unsigned bits) { atomicOr(status, bits); }
}  // namespace

int mgr_update_gate_eval(mgr_ctx* c, unsigned mask, float* flag) {
  MGR_REQUIRE(c && flag, "null argument");
  hipLaunchKernelGGL(k_gate_eval, dim3(1), dim3(1), 0, mgr_stream(c), mgr_status_block(c), mask, flag);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_update_gate_set(mgr_ctx* c, const float* flag) {
  MGR_REQUIRE(c, "null ctx");
  c->gate_flag = flag;
  return 0;
}

int mgr_scan_status_inject(mgr_ctx* c, unsigned bits) {
  MGR_REQUIRE(c, "null ctx");
  hipLaunchKernelGGL(k_status_or, dim3(1), dim3(1), 0, mgr_stream(c), mgr_status_block(c), bits);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_event_record(mgr_ctx* c, int ev) {
  MGR_REQUIRE(c && ev >= 0 && ev < MGR_NUM_EVENTS, "bad event index");
  MGR_HIP(hipEventRecord(c->events[ev], mgr_stream(c)));
  return 0;
}

int mgr_event_elapsed_ms(mgr_ctx* c, int ev0, int ev1, float* ms) {
  MGR_REQUIRE(c && ms && ev0 >= 0 && ev0 < MGR_NUM_EVENTS && ev1 >= 0 && ev1 < MGR_NUM_EVENTS, "bad argument");
  MGR_HIP(hipEventSynchronize(c->events[ev1]));
  MGR_HIP(hipEventElapsedTime(ms, c->events[ev0], c->events[ev1]));
  return 0;
}

static int prof_reserve(mgr_ctx* c, int f, int ncap) {
  if (ncap <= c->prof_cap[f]) return 0;
  mgr_ctx::ProfPair* np = new mgr_ctx::ProfPair[ncap];
  for (int i = 0; i < c->prof_cap[f]; ++i) np[i] = c->prof_pairs[f][i];
  for (int i = c->prof_cap[f]; i < ncap; ++i) {
    MGR_HIP(hipEventCreate(&np[i].a));
    MGR_HIP(hipEventCreate(&np[i].b));
  }
  delete[] c->prof_pairs[f];
  c->prof_pairs[f] = np;
  c->prof_cap[f] = ncap;
  return 0;
}

int mgr_prof_enable(mgr_ctx* c, int family_mask) {
  MGR_REQUIRE(c, "null ctx");
  // the event pairs of the enabled families are created HERE, not at the launch that first needs them: creating a few hundred
  // events costs tens of milliseconds, and it used to land inside whatever region the caller was timing (round 3: one step in ten
  // of the small configurations took 20-45 ms instead of 5)
  for (int f = 0; f < MGR_K_COUNT; ++f) {
    if (!(family_mask & (1 << f))) continue;
    int r = prof_reserve(c, f, 1024);
    if (r) return r;
  }
  c->prof_mask = family_mask;
  return 0;
}

static int prof_collect(mgr_ctx* c) {
  for (int f = 0; f < MGR_K_COUNT; ++f) {
    for (int i = 0; i < c->prof_n[f]; ++i) {
      float ms = 0;
      MGR_HIP(hipEventSynchronize(c->prof_pairs[f][i].b));
      MGR_HIP(hipEventElapsedTime(&ms, c->prof_pairs[f][i].a, c->prof_pairs[f][i].b));
      c->prof_ms[f] += ms;
      c->prof_launches[f] += 1;
    }
    c->prof_n[f] = 0;
  }
  return 0;
}

int mgr_prof_get(mgr_ctx* c, int family, int* launches, float* ms) {
  MGR_REQUIRE(c && family >= 0 && family < MGR_K_COUNT, "bad family");
  int r = mgr_sync(c);
  if (r) return r;
  r = prof_collect(c);
  if (r) return r;
  if (launches) *launches = c->prof_launches[family];
  if (ms) *ms = c->prof_ms[family];
  return 0;
}

int mgr_prof_reset(mgr_ctx* c) {
  MGR_REQUIRE(c, "null ctx");
  int r = mgr_sync(c);
  if (r) return r;
  r = prof_collect(c);
  if (r) return r;
  for (int f = 0; f < MGR_K_COUNT; ++f) {
    c->prof_ms[f] = 0;
    c->prof_launches[f] = 0;
  }
  return 0;
}

}  // extern "C"

int mgr_prof_begin(mgr_ctx* c, int f) {
  if (!(c->prof_mask & (1 << f))) return 0;
  if (c->prof_n[f] == c->prof_cap[f]) {
    if (c->prof_cap[f] >= 4096) {  // drain instead of growing without bound
      int r = mgr_sync(c);
      if (r) return r;
      r = prof_collect(c);
      if (r) return r;
    } else {
      int r = prof_reserve(c, f, c->prof_cap[f] ? c->prof_cap[f] * 2 : 64);
      if (r) return r;
    }
  }
  MGR_HIP(hipEventRecord(c->prof_pairs[f][c->prof_n[f]].a, mgr_stream(c)));
  return 0;
}

int mgr_prof_end(mgr_ctx* c, int f) {
  if (!(c->prof_mask & (1 << f))) return 0;
  MGR_HIP(hipEventRecord(c->prof_pairs[f][c->prof_n[f]].b, mgr_stream(c)));
  c->prof_n[f] += 1;
  return 0;
}
