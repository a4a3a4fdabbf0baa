// Internal helpers shared by the libmgr.so translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mgr.h"

constexpr int MGR_MAX_PERSIST = 8;
constexpr int MGR_MAX_FROZEN = 32;

constexpr size_t MGR_SMALL_D2H = 4096;

struct mgr_ctx {
  int device;
  int cu_count;
  size_t hbm_bytes;
  char name[64];
  hipStream_t streams[MGR_NUM_STREAMS];
  int cur;
  hipEvent_t events[MGR_NUM_EVENTS];
  hipEvent_t xev[64];  // round-robin events for stream_wait
  int xev_next;
  // profiling: ring of (start, stop) event pairs per family
  int prof_mask;
  struct ProfPair {
    hipEvent_t a, b;
  };
  ProfPair* prof_pairs[MGR_K_COUNT];
  int prof_n[MGR_K_COUNT];
  int prof_cap[MGR_K_COUNT];
  float prof_ms[MGR_K_COUNT];
  int prof_launches[MGR_K_COUNT];
  int tune[MGR_TUNE_COUNT];
  void* h_small_pinned;     // page-locked scratch of MGR_SMALL_D2H bytes for small blocking read-backs (mgr_d2h)
  unsigned* sticky_status;  // device words: [0] status bits of every persistent launch since the last clear, [1] resident seq,
                            // [2] optimizer updates skipped by the update gate (mgr_update_gate_*)
  unsigned* status_bound;   // mgr_scan_status_bind: the block [0] and [2] live in instead (one per engine sharing the context)
  const float* gate_flag;   // mgr_update_gate_set: device flag that turns mgr_adam_step / mgr_maxnorm_cols into no-ops when != 0
  // persistent launches that may still be running (lstm.hip: mgr_persist_admit / mgr_persist_commit)
  struct Persist {
    hipEvent_t done;
    int active, stream, wgs, waves, per_cu, fused;
    unsigned seq;
  };
  Persist persist[MGR_MAX_PERSIST];
  unsigned persist_seq;     // sequence number of the last persistent launch of this context
  int persist_serialised;   // launches that had to be ordered behind another stream's persistent launch
  unsigned attr_done;       // bit k: function attributes of kernel family k have been set on this context's device
  // split weight planes of FROZEN weights (gemm_split.hip, mgr_weight_planes_cache): weights the caller promised not to rewrite, and
  // the workspaces that hold their planes as of that promise
  struct PlaneEntry {
    const void* Wp;
    const void* ws;
    int F, H;
  };
  const void* frozen_w[MGR_MAX_FROZEN];
  PlaneEntry planes[MGR_MAX_FROZEN];
  unsigned planes_evict, frozen_evict;   // round-robin victims when a table is full
};

int mgr_fail(int code, const char* fmt, ...);

// cached split weight planes (gemm_split.hip, mgr_weight_planes_cache) live in a projection workspace: any OTHER use of that workspace
// forgets them
static inline void mgr_planes_forget_ws(mgr_ctx* c, const void* ws) {
  for (int i = 0; i < MGR_MAX_FROZEN; ++i)
    if (c->planes[i].ws == ws) c->planes[i] = mgr_ctx::PlaneEntry{nullptr, nullptr, 0, 0};
}

#define MGR_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return mgr_fail(-2, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define MGR_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) return mgr_fail(-1, __VA_ARGS__); \
  } while (0)

#define MGR_LAUNCH_CHECK() MGR_HIP(hipGetLastError())

static inline hipStream_t mgr_stream(mgr_ctx* c) { return c->streams[c->cur]; }
// the status block persistent scans report into and the update gate reads: the bound one, else the context's own
static inline unsigned* mgr_status_block(mgr_ctx* c) { return c->status_bound ? c->status_bound : c->sticky_status; }

// RAII-less profiling bracket: call mgr_prof_begin before and mgr_prof_end after the launches of a family.
int mgr_prof_begin(mgr_ctx* c, int family);
int mgr_prof_end(mgr_ctx* c, int family);

static inline size_t mgr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device-side RNG: one 64-bit mix per element index (splitmix64 finaliser); stateless ---------------
__host__ __device__ static inline uint64_t mgr_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ static inline uint32_t mgr_rand_u32(uint64_t seed, uint64_t idx) {
  return (uint32_t)(mgr_mix64(seed * 0xD1342543DE82EF95ull + idx) >> 32);
}
// dropout keep decision shared by mgr_dropout_mask and the fused dense kernels
__host__ __device__ static inline float mgr_drop_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  // uniform in [0,1): top 24 bits
  float u = (float)(mgr_rand_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
  return (u >= p) ? inv_keep : 0.0f;
}
// XT[b * xtb + f * ldt + t] = X[(b * T + t) * ldx + f], zero for T <= t < ldt_fill (gemm.hip)
int mgr_transpose_bt_strided(mgr_ctx* c, const float* X, int ldx, float* XT, int ldt, long long xtb, int ldt_fill, int B, int T, int F);
// the same into the split row format (gemm_split.hip): row (b, f) = ldt f16 hi values, then ldt f16 lo values of x 2^13
int mgr_transpose_bt_split_strided(mgr_ctx* c, const float* X, int ldx, float* XS, int ldt, long long xsb, int ldt_fill, int B, int T, int F);
// dU / db of one LSTM direction (gemm.hip); ws: mgr_lstm_param_grads_ws_bytes(B, T, F, H) bytes; dbsum: [B][4H] sums of dZ over time or null
int mgr_param_grads_du_db(mgr_ctx* c, const float* Hs, int ldh, const float* dZ, float* dUp, float* dbp, int B, int T, int F, int H, int reverse,
                          void* ws, const float* dbsum);

// zmax[b * N + col] = largest |dZ[b, t, col]| over t as float bits (gemm_split.hip)
int mgr_rowmax_bt(mgr_ctx* c, const float* dZ, int N, int T, int B, unsigned* zmax, float* zsum);   // (either output may be null)
