// Job descriptors shared by lstm.hip (planning) and lstm_cluster.hip (persistent multi-CU scan kernel).
#pragma once
#include "common.h"

constexpr int MGR_MAX_SCAN_JOBS = 8;

struct ClusterJob {
  const float* Z;
  const float* Up;
  float* Y;
  const float* R;
  float* G;
  float* Cs;
  float* xbuf;      // [nbg][2][IMG] exchange slots (B-operand image layout)
  unsigned* flags;  // [nbg][64] per-workgroup epoch
  int ldy, ldr, B, T, H, reverse;
  int ks, tpw, nw;  // k-steps (H/4), tiles per wave, active waves per workgroup
  int wg_begin;     // first blockIdx of this job
  int G_;           // workgroups per cluster (one cluster = one 16-sample batch group)
  int nbg;          // batch groups
  int pair;         // batch groups per workgroup: 1, or 2 = software-pipelined pair (cluster_run2)
  // XCD-interleaved placement: jobs with identical geometry form a CLASS whose clusters are dealt round-robin over
  // consecutive workgroup ids (w - cls_begin) % cls_nclusters, so that - under the dispatcher's observed round-robin
  // over the 8 XCDs - every member of a cluster lands on the same XCD.  Speed only; verified at run time.
  int cls_begin, cls_nclusters, cls_cluster0;
};

struct ClusterLaunch {
  int njobs;
  unsigned* xcc;     // [grid] XCC id + 1 of every workgroup, published at kernel start
  int xcd_local;     // opt-in: clusters found on one XCD exchange through its L2 (plain stores + nt loads)
  int gather_delay;  // 64-cycle sleeps between a workgroup's own publish and its first gather pass (a failed pass costs a
                     // full fabric round trip, a short wait is cheaper)
  int ksplit;        // one-tile-per-wave clusters use the K-split step (cluster_run_ks): register-direct gather
  unsigned* status;  // [0] != 0 -> a bounded spin gave up (zeroed ahead of every launch)
  unsigned* sticky;  // context-wide word, never cleared by a launch: any give-up leaves its code here (mgr_scan_status)
  ClusterJob job[MGR_MAX_SCAN_JOBS];
};

// true if (ks, tpw) has an instantiation
bool mgr_cluster_supported(int ks, int tpw);
bool mgr_cluster_pair_supported(int ks);
int mgr_cluster_launch(mgr_ctx* c, const ClusterLaunch& L, int total_wgs, bool any_exchange);

// ---- backward (lstm_cluster_bwd.hip)
struct ClusterBwdJob {
  const float* dY;
  const float* gates;
  const float* cs;
  const float* Up;
  float* dZ;
  float* xbuf;  // [nbg][2][IMG]
  int lddy, B, T, H, reverse;
  int wg_begin, G_, nbg;
  int cls_begin, cls_nclusters, cls_cluster0;
};
struct ClusterBwdLaunch {
  int njobs;
  unsigned* xcc;
  int xcd_local;
  unsigned* status;
  unsigned* sticky;
  ClusterBwdJob job[MGR_MAX_SCAN_JOBS];
};
bool mgr_cluster_bwd_supported(int H);
size_t mgr_cluster_bwd_img_floats(int H);
int mgr_cluster_bwd_launch(mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs);

#ifdef __HIPCC__
// last thing a workgroup of a cluster kernel does: copy a give-up code of this launch into the context's sticky word
__device__ __forceinline__ void mgr_cluster_exit(unsigned* status, unsigned* sticky) {
  if (threadIdx.x == 0 && sticky) {
    unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st != 0) __hip_atomic_store(sticky, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- device helpers shared by the forward and backward cluster kernels -------------------------------------------
// Publish this workgroup's XCC (XCD) id, then learn whether every member of its cluster sits on the same XCD.  Each
// wave does this for itself (no LDS), the result is wave-uniform.  If true, the cluster may exchange through its
// XCD's L2 with PLAIN stores (sc1 loads bypass L1 and are served by that L2); otherwise - any other placement - it
// uses write-through (sc1) stores.  Either way correctness never depends on placement: the predicate is computed
// from the same table by every member, and every exchanged word is validated by its epoch parity.
__device__ __forceinline__ bool mgr_cluster_same_xcd(unsigned* xcc_table, int my_wg, int cls_begin, int cls_nclusters, int cl,
                                                     int G, unsigned* status) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  const unsigned mine = (x & 0xFu) + 1u;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) __hip_atomic_store(xcc_table + my_wg, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (G <= 1) return true;
  bool same = true;
  unsigned spins = 0;
  for (;;) {
    unsigned v = mine;
    if (lane < G) v = __hip_atomic_load(xcc_table + cls_begin + lane * cls_nclusters + cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (interleaved mapping)
    if (__all(v != 0u)) {
      same = __all(v == mine);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
    if (++spins > (1u << 18)) {  // bounded: report and fall back to the always-correct path
      if (lane == 0) __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      same = false;
      break;
    }
  }
  return same;
}
#endif
