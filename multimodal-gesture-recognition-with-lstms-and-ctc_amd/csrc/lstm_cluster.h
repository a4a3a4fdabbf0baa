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
};

struct ClusterLaunch {
  int njobs;
  unsigned* status;  // [0] != 0 -> a bounded spin gave up
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
};
struct ClusterBwdLaunch {
  int njobs;
  unsigned* status;
  ClusterBwdJob job[MGR_MAX_SCAN_JOBS];
};
bool mgr_cluster_bwd_supported(int H);
size_t mgr_cluster_bwd_img_floats(int H);
int mgr_cluster_bwd_launch(mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs);
