// Job descriptors shared by lstm.hip (planning, admission) and the persistent multi-CU scan kernels
// (lstm_cluster.hip forward, lstm_cluster_bwd.hip BPTT).
#pragma once
#include "common.h"

constexpr int MGR_MAX_SCAN_JOBS = 8;

// Header of every persistent launch's workspace (zeroed by a memset node ahead of the launch).
//   [0] give-up code of THIS launch (a bounded spin expired)          [1] arrival counter (workgroups that have started)
constexpr size_t kScanHdrBytes = 4096;   //   [64, 1024) XCC (XCD) id + 1 of every workgroup of the launch (XCD-local exchange)

// Status block (mgr_ctx::sticky_status, or the block bound with mgr_scan_status_bind; never cleared by a launch):
//   [0] OR of every launch's status bits since the last mgr_scan_status_clear
//   [1] (context's own block only) highest launch sequence number whose workgroups have ALL started (mgr_stream_wait_next_resident)
//   [16, 32) (context's own block only) the same per launch: word 16 + seq % 16 holds seq once launch `seq` is resident
//   [2] optimizer updates skipped by the update gate since the last clear
//   [8, 16) which samples met a non-finite hidden state: bit (sample mod 256), set with MGR_ST_NONFINITE
enum : unsigned {
  MGR_ST_GAVE_UP = MGR_SCAN_GAVE_UP,       // a bounded spin expired: a peer workgroup never showed up / never published
  MGR_ST_NONFINITE = MGR_SCAN_NONFINITE,   // a hidden state became NaN / Inf: outputs carry NaN from that step on (not a hang)
};

// What every persistent launch carries besides its jobs.
struct ClusterCommon {
  unsigned* status;   // launch header (see kScanHdrBytes)
  unsigned* sticky;   // status block the launch reports into (mgr_scan_status_bind; the context's own by default)
  unsigned* resident; // context-wide word: highest launch sequence number whose workgroups have all started
  unsigned seq;       // launch sequence number of this context (1, 2, ...)
  int total_wgs;      // grid size: the arrival that makes the counter reach it publishes `seq` as resident
};

struct ClusterJob {
  const float* Z;
  const float* Up;
  float* Y;
  const float* R;
  float* G;
  float* Cs;
  float* xbuf;      // [nbg][2][IMG] exchange slots (B-operand image layout)
  float* YT;        // optional transposed output: YT[b * ytb + unit * ldt + t] = what Y[b, t, unit] gets (K-split kernel; else null)
  long long ytb;    // ... its batch stride in floats
  int ldt;          // ... its row length (T padded; entries t in [T, ldt) are written as zero)
  int yt_split;     // ... rows in the split row format (mgr.h): ldt f16 hi values, then ldt f16 lo values of y 2^13
  int ldy, ldr, B, T, H, reverse;
  int ks, tpw, nw;  // k-steps (H/4), tiles per wave, active waves per workgroup
  int G_;           // workgroups per cluster (one cluster = one 16-sample batch group)
  int nbg;          // clusters of this job: 16-sample batch groups, or (ClusterLaunch::pair) pairs of them
  int nbg16;        // 16-sample batch groups
  // jobs with identical geometry form a CLASS that shares one contiguous workgroup range: cluster `cl` of the class
  // owns workgroups [cls_begin + cl*G_, +G_); a job's batch group bg is cluster cls_cluster0 + bg
  int cls_begin, cls_nclusters, cls_cluster0;
  int cls_rot;   // XCD-local layout: cluster c of the class sits on lane (c + cls_rot) % 8 (classes continue where the previous one stopped)
};

struct ClusterLaunch {
  ClusterCommon cm;
  int njobs;
  int ksplit;        // one-tile-per-wave clusters use the K-split step (cluster_run_ks: register-direct gather); 0 = LDS-image step
  int split16;       // K-split launches: f16 (hi, lo) operands on the f16 matrix pipe (cluster_run_k16; tune key 14 = 1: f32 MFMA step)
  int live_wgs;      // workgroups of the grid that run a cluster (the others are empty ids of the octet layout)
  int pair;          // split16 K-split launches: every workgroup runs TWO 16-sample groups (cluster_run_k16p: one workgroup per CU)
  int fused;         // split16 K-split launches: every workgroup (8 waves, a CU of its own) runs TWO unit groups of its cluster; the job
                     // table's cls_* fields then count ceil(G_ / 2) members per cluster (k_scan_cluster_k16f)
  int xcd_local;     // K-split launches: clusters are laid out on workgroup ids congruent mod 8 (one XCD under the dispatcher's
                     // round-robin); a cluster that FINDS all its members on one XCD publishes with plain stores into that L2
  ClusterJob job[MGR_MAX_SCAN_JOBS];
};

// true if (ks, tpw) has an instantiation / if ks has a K-split instantiation
bool mgr_cluster_supported(int ks, int tpw);
bool mgr_cluster_ks_supported(int ks);
// geometry of the launch that mgr_cluster_launch would issue: waves per workgroup, workgroups per CU
void mgr_cluster_geometry(const ClusterLaunch& L, bool any_exchange, int* waves, int* per_cu);
bool mgr_cluster_uses_ks(const ClusterLaunch& L, bool any_exchange);   // the launch will run the K-split kernel (which honours ClusterJob::YT)
int mgr_cluster_launch(mgr_ctx* c, const ClusterLaunch& L, int total_wgs, bool any_exchange);

// ---- backward (lstm_cluster_bwd.hip)
struct ClusterBwdJob {
  const float* dY;
  const float* gates;
  const float* cs;
  const float* Up;
  float* dZ;
  unsigned* dzmax;   // optional [B][4H]: largest |dZ| over t per (sample, gate column), float bits (mgr_scan_bwd_job)
  float* dbsum;      // optional [B][4H]: sum of dZ over t per (sample, gate column), in step order (mgr_scan_bwd_job)
  float* xbuf;  // [nbg][2][IMG]
  int lddy, B, T, H, reverse;
  int G_, nbg;
  int cls_begin, cls_nclusters, cls_cluster0, cls_rot;
};
struct ClusterBwdLaunch {
  ClusterCommon cm;
  int njobs;
  int xcd_local;   // clusters laid out on workgroup ids congruent mod 8; plain-store exchange where a cluster finds itself on one XCD
  int fused;       // narrow split-f16 layers: 8-wave workgroups that run TWO unit groups of their cluster, a CU each; the job table's
                   // cls_* fields then count ceil(G_ / 2) members per cluster (k_scan_cluster_bwd16_f / _fd)
  ClusterBwdJob job[MGR_MAX_SCAN_JOBS];
};
bool mgr_cluster_bwd_supported(int H);
bool mgr_cluster_bwd_fusable(const mgr_ctx* c, const ClusterBwdLaunch& L);   // L.xcd_local, the jobs' H / G_ filled in
size_t mgr_cluster_bwd_img_floats(int H);
void mgr_cluster_bwd_geometry(const mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs, int* waves, int* per_cu);
int mgr_cluster_bwd_launch(mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs, int form16);   // form16: 0 trimmed, 1 yielding, 2 direct (mgr.h, tune key 16)

// ---- admission of persistent launches (lstm.hip): co-residency by construction across the streams of a context
int mgr_persist_admit(mgr_ctx* c, int wgs, int waves_per_wg, int per_cu, int fused, unsigned* seq_out);
int mgr_persist_commit(mgr_ctx* c, int wgs, int waves_per_wg, int per_cu, int fused);

#ifdef __HIPCC__
// first thing a workgroup of a persistent kernel does: count itself in; the last arrival publishes the launch as resident
__device__ __forceinline__ void mgr_cluster_enter(const ClusterCommon& cm) {
  if (threadIdx.x == 0) {
    const unsigned n = __hip_atomic_fetch_add(cm.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (n == (unsigned)cm.total_wgs) {
      __hip_atomic_fetch_max(cm.resident, cm.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // per-launch residency (mgr_stream_wait_resident): a ring of 16 words behind the context-wide one, slot seq % 16
      __hip_atomic_fetch_max(cm.resident + 15 + (cm.seq & 15u), cm.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// XCD-local exchange (forward K-split and BPTT cluster kernels).  Workgroup ids are dealt round-robin over the 8 XCDs (observed,
// never relied upon), so with the octet layout the members of cluster 8o + x are the ids  cls_begin + o*8G + 8k + x  (k < G):
// congruent mod 8, i.e. ONE XCD and one L2 (x = (c + rot) % 8 for cluster c: a class starts on the lane after the previous
// class's last cluster, so that fewer than eight clusters per class still spread over all XCDs).  Every workgroup publishes the XCD it really runs on in the launch header
// (status + 64 + blockIdx); a cluster whose members all show the same id exchanges through that L2 with PLAIN stores (a
// write-through store drops the line from the L2 and every peer's load goes out to the fabric), any other placement keeps the
// write-through stores.  The decision is a function of the published table only - all members agree - and every exchanged word
// is still validated by its epoch parity: placement is speed, never correctness.  Returns the decision (wave-uniform);
// decodes (cluster, unit group) of workgroup w_ within its class.
__device__ __forceinline__ bool mgr_cluster_octet(const ClusterCommon& cm, int cls_begin, int G, int rot, int w_, int& cl, int& ug) {
  unsigned* table = cm.status + 64;
  unsigned xid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
  const unsigned mine = (xid & 0xFu) + 1u;
  if (threadIdx.x == 0) __hip_atomic_store(table + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int o = w_ / (8 * G), rem = w_ % (8 * G);
  ug = rem >> 3;
  cl = 8 * o + (((rem & 7) - rot) & 7);
  const int lane = threadIdx.x & 63;
  bool same = true;
  unsigned spins = 0;
  for (;;) {
    unsigned v = mine;
    if (lane < G) v = __hip_atomic_load(table + cls_begin + o * 8 * G + 8 * lane + (rem & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all(v != 0u)) {
      same = __all(v == mine);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
    if (++spins > (1u << 18)) {   // a member that never started: the bounded spins of the exchange will report it
      same = false;
      break;
    }
  }
  return same;
}
// which SAMPLE met a non-finite hidden state: bit (b mod 256) of words [8, 16) of the status block (mgr.h, mgr_scan_status_bind)
__device__ __forceinline__ void mgr_mark_sample(const ClusterCommon& cm, int b) {
  __hip_atomic_fetch_or(cm.sticky + 8 + ((b & 255) >> 5), 1u << (b & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// last thing: fold this launch's status bits into the context's sticky word
__device__ __forceinline__ void mgr_cluster_exit(const ClusterCommon& cm) {
  if (threadIdx.x == 0) {
    const unsigned st = __hip_atomic_load(cm.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st != 0) __hip_atomic_fetch_or(cm.sticky, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif
