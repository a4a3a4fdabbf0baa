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
};

struct ClusterLaunch {
  int njobs;
  unsigned* status;  // [0] != 0 -> a bounded spin gave up
  ClusterJob job[MGR_MAX_SCAN_JOBS];
};

// true if (ks, tpw) has an instantiation
bool mgr_cluster_supported(int ks, int tpw);
int mgr_cluster_launch(mgr_ctx* c, const ClusterLaunch& L, int total_wgs, bool any_exchange);
