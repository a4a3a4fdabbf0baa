// C-ABI entry points of the LSTM recurrence (K3, K7-scan): pick the kernel family for each shape.
//   forward:  H/4 with an instantiation (H = 100, 128, 300, 500 and test sizes) -> lstm_cluster.hip: persistent
//             weight-stationary MFMA kernel, one CU per batch group (H <= 128) or clusters of CUs exchanging h_t
//   backward: H <= 128 -> lstm_mfma.hip (single-CU weight-stationary MFMA)
//   anything else (H <= 1024) -> lstm_simple.hip (U streamed from L2; correctness fallback)
#include <algorithm>

#include "common.h"
#include "lstm_cluster.h"

int mgr_scan_fwd_simple(mgr_ctx*, const float*, const float*, float*, int, const float*, int, float*, float*, int, int, int, int);
int mgr_scan_bwd_simple(mgr_ctx*, const float*, int, const float*, const float*, const float*, float*, int, int, int, int);
int mgr_scan_bwd_mfma(mgr_ctx*, const float*, int, const float*, const float*, const float*, float*, int, int, int, int);

namespace {

constexpr size_t kScanHdrBytes = 8192;  // [0,2048): status + stamps; [2048,8192): XCC id per workgroup of the launch

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct Cfg {
  int nw, tpw, pair;
};
// candidate (active waves, tiles per wave, batch groups per workgroup) configurations
constexpr int NCFG = 5;
const Cfg kCfgs[NCFG] = {{4, 1, 1}, {8, 1, 1}, {8, 2, 1}, {8, 4, 1}, {4, 1, 2}};

struct Plan {
  bool cluster[MGR_MAX_SCAN_JOBS];
  Cfg cfg[MGR_MAX_SCAN_JOBS];
  int G[MGR_MAX_SCAN_JOBS], nbg[MGR_MAX_SCAN_JOBS], wgs[MGR_MAX_SCAN_JOBS];
  int total;
  bool any, exchange;
};

size_t job_ws(const mgr_scan_job& j) {
  int ks = j.H / 4;
  size_t img = (size_t)((ks + 3) / 4) * 256;
  int nbg = (j.B + 15) / 16;
  return mgr_align_up((size_t)nbg * 2 * img * sizeof(float), 256) + mgr_align_up((size_t)nbg * 64 * sizeof(unsigned), 256);
}

// Choose per-job configurations: minimise the slowest job's per-step MFMA time subject to all workgroups
// being co-resident (sum <= CUs).  Jobs that cannot run on the cluster kernel are left to the other families.
void make_plan(const mgr_ctx* c, int njobs, const mgr_scan_job* jobs, Plan& P) {
  int path = c->tune[MGR_TUNE_SCAN_PATH];
  P.any = false;
  P.exchange = false;
  P.total = 0;
  int idx[MGR_MAX_SCAN_JOBS], n = 0;
  for (int i = 0; i < njobs; ++i) {
    int H = jobs[i].H, ks = H / 4;
    bool ok = (H % 4 == 0);
    if (ok) {
      ok = false;
      for (int k = 0; k < NCFG; ++k) ok = ok || mgr_cluster_supported(ks, kCfgs[k].tpw);
      ok = ok || mgr_cluster_pair_supported(ks);
    }
    if (path == 1) ok = false;
    P.cluster[i] = ok;
    if (ok) idx[n++] = i;
  }
  if (n == 0) return;
  int best[MGR_MAX_SCAN_JOBS], cur[MGR_MAX_SCAN_JOBS];
  long best_cost = -1;
  int combos = 1;
  for (int k = 0; k < n; ++k) combos *= NCFG;
  for (int code = 0; code < combos; ++code) {
    int x = code, total = 0;
    long worst = 0, sum = 0;
    bool feas = true, exch = false, anypair = false;
    for (int k = 0; k < n; ++k) {
      cur[k] = x % NCFG;
      x /= NCFG;
      const mgr_scan_job& j = jobs[idx[k]];
      Cfg f = kCfgs[cur[k]];
      int ks = j.H / 4;
      if (f.pair == 2 ? !mgr_cluster_pair_supported(ks) : !mgr_cluster_supported(ks, f.tpw)) feas = false;
      if (path == 3 && cur[k] != 0) feas = false;
      if (path == 4 && cur[k] != 1) feas = false;
      if (path == 5 && cur[k] != 4) feas = false;
      {  // experiment hooks: tune keys 4 / 5 force the configuration index (+1) of jobs with H >= 400 / H < 400
        int forced = j.H >= 400 ? c->tune[4] : c->tune[5];
        if (forced > 0 && cur[k] != forced - 1) feas = false;
      }
      if (path != 5 && f.pair == 2) feas = false;  // measured slower than one group per workgroup: opt-in only
      if (path == 2 && (f.nw * f.tpw < ks)) feas = false;  // force single-CU (no exchange)
      int tiles = f.nw * f.tpw;
      int G = (ks + tiles - 1) / tiles;
      if (G > 64) feas = false;
      if (G > 1) exch = true;
      int nbg = (j.B + 15) / 16;
      if (f.pair == 2 && (G == 1 || nbg < 2)) feas = false;  // pairing only pays when there is a hand-off to hide
      // (classes of jobs are laid out on workgroup ranges rounded up to a multiple of 8 - the XCD count - at launch; count
      // every job rounded up so that a plan accepted here always passes the launcher's co-residency check)
      total += (G * ((nbg + f.pair - 1) / f.pair) + 7) / 8 * 8;
      if (f.pair == 2) anypair = true;
      // per-step estimate in cycles: MFMA chain per SIMD (+15% issue overhead) + cell update + exchange / barrier
      int tiles_here = std::min(tiles, ks);
      int per_simd = (tiles_here + 3) / 4;
      long t = (long)per_simd * ks * 37 + 700 + (G > 1 ? 3300 : 400);
      if (f.pair == 2) t = 2 * ((long)ks * 37 + 700 + 500);  // two groups back to back, hand-off hidden
      if (total > c->cu_count && f.pair == 1) t += (long)per_simd * ks * 12;  // a second workgroup on the CU competes for the MFMA pipe part of the time
      worst = std::max(worst, t);
      sum += t;
    }
    // capacity: one 8-wave workgroup per CU, or two 4-wave workgroups (<= 80 KiB LDS each) per CU
    bool all4 = true;
    int maxks = 0;
    for (int k = 0; k < n; ++k) {
      if (kCfgs[cur[k]].nw != 4) all4 = false;
      maxks = std::max(maxks, jobs[idx[k]].H / 4);
    }
    size_t lds2 = (size_t)(anypair ? 4 : 2) * ((maxks + 3) / 4) * 1024;
    int capacity = (all4 && lds2 <= 80 * 1024) ? 2 * c->cu_count : c->cu_count;
    // the paired kernel is its own launch configuration (4 waves, one workgroup per CU): all jobs or none
    bool allpair = true;
    for (int k = 0; k < n; ++k)
      if (kCfgs[cur[k]].pair != 2) allpair = false;
    if (anypair && !allpair) feas = false;
    if (anypair) capacity = c->cu_count;
    if (!feas || (exch && total > capacity)) continue;
    long cost = worst * 1000 + sum / n;
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      for (int k = 0; k < n; ++k) best[k] = cur[k];
    }
  }
  if (best_cost < 0) {  // does not fit: leave these jobs to the fallback
    for (int k = 0; k < n; ++k) P.cluster[idx[k]] = false;
    return;
  }
  for (int k = 0; k < n; ++k) {
    int i = idx[k];
    P.cfg[i] = kCfgs[best[k]];
    int tiles = P.cfg[i].nw * P.cfg[i].tpw;
    P.G[i] = (jobs[i].H / 4 + tiles - 1) / tiles;
    P.nbg[i] = (jobs[i].B + 15) / 16;
    P.wgs[i] = P.G[i] * ((P.nbg[i] + P.cfg[i].pair - 1) / P.cfg[i].pair);
    P.total += P.wgs[i];
    if (P.G[i] > 1) P.exchange = true;
  }
  P.any = true;
}

}  // namespace

extern "C" {

int mgr_tune(mgr_ctx* c, int key, int value) {
  MGR_REQUIRE(c && key >= 0 && key < MGR_TUNE_COUNT, "bad tune key");
  c->tune[key] = value;
  return 0;
}

static size_t bwd_job_ws(const mgr_scan_bwd_job& j) {
  int nbg = (j.B + 15) / 16;
  size_t cluster = mgr_align_up((size_t)nbg * 2 * mgr_cluster_bwd_img_floats(j.H) * sizeof(float), 256);
  size_t fallback = mgr_align_up((size_t)4 * j.H * j.H * sizeof(float), 256);
  return std::max(cluster, fallback);
}

size_t mgr_lstm_scan_bwd_multi_ws_bytes(int njobs, const mgr_scan_bwd_job* jobs) {
  size_t s = kScanHdrBytes;
  for (int i = 0; i < njobs; ++i) s += bwd_job_ws(jobs[i]);
  return s;
}

size_t mgr_lstm_scan_ws_bytes(int B, int T, int H) {
  (void)T;
  mgr_scan_job j;
  memset(&j, 0, sizeof(j));
  j.B = B;
  j.H = H;
  size_t fallback = mgr_align_up((size_t)4 * H * H * sizeof(float), 256);  // U^T for the fallback backward kernel
  mgr_scan_bwd_job bj;
  memset(&bj, 0, sizeof(bj));
  bj.B = B;
  bj.H = H;
  return std::max(std::max(fallback, job_ws(j) + kScanHdrBytes), mgr_lstm_scan_bwd_multi_ws_bytes(1, &bj));
}

size_t mgr_lstm_scan_multi_ws_bytes(int njobs, const mgr_scan_job* jobs) {
  size_t s = kScanHdrBytes;  // status word + diagnostic stamps + XCC table
  for (int i = 0; i < njobs; ++i) s += job_ws(jobs[i]);
  return s;
}

int mgr_lstm_scan_fwd_multi(mgr_ctx* c, int njobs, const mgr_scan_job* jobs, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && jobs && njobs > 0 && njobs <= MGR_MAX_SCAN_JOBS, "bad job list");
  for (int i = 0; i < njobs; ++i) {
    const mgr_scan_job& j = jobs[i];
    MGR_REQUIRE(j.Z && j.Up && j.Y, "job %d: null argument", i);
    MGR_REQUIRE(j.B > 0 && j.T > 0 && j.H > 0 && j.ldy >= j.H && (!j.R || j.ldr >= j.H), "job %d: bad shape", i);
    MGR_REQUIRE(aligned16(j.Z) && aligned16(j.Up) && (!j.gates || aligned16(j.gates)), "job %d: Z/Up/gates must be 16-byte aligned", i);
  }
  int r = mgr_prof_begin(c, MGR_K_SCAN_FWD);
  if (r) return r;
  Plan P;
  make_plan(c, njobs, jobs, P);
  if (P.any && (!ws || ws_bytes < mgr_lstm_scan_multi_ws_bytes(njobs, jobs))) {
    // no workspace for the exchange: degrade to the non-cluster families
    for (int i = 0; i < njobs; ++i) P.cluster[i] = false;
    P.any = false;
  }
  unsigned* status = nullptr;
  if (P.any) {
    ClusterLaunch L;
    memset(&L, 0, sizeof(L));
    char* w = reinterpret_cast<char*>(ws);
    status = reinterpret_cast<unsigned*>(w);
    char* base = w;
    L.xcc = reinterpret_cast<unsigned*>(w + 2048);
    w += kScanHdrBytes;
    // classes: cluster jobs with identical geometry (same H and configuration, e.g. the two directions of a layer) share
    // one XCD-interleaved workgroup range
    int cls_of[MGR_MAX_SCAN_JOBS], ncls = 0, cls_first[MGR_MAX_SCAN_JOBS], cls_clusters[MGR_MAX_SCAN_JOBS];
    for (int i = 0; i < njobs; ++i) {
      if (!P.cluster[i]) continue;
      int found = -1;
      for (int k = 0; k < ncls; ++k) {
        int f = cls_first[k];
        if (jobs[f].H == jobs[i].H && P.cfg[f].nw == P.cfg[i].nw && P.cfg[f].tpw == P.cfg[i].tpw &&
            P.cfg[f].pair == P.cfg[i].pair)
          found = k;
      }
      if (found < 0) {
        found = ncls++;
        cls_first[found] = i;
        cls_clusters[found] = 0;
      }
      cls_of[i] = found;
      cls_clusters[found] += P.nbg[i];
    }
    int cls_begin[MGR_MAX_SCAN_JOBS], begin = 0, cls_next[MGR_MAX_SCAN_JOBS];
    for (int k = 0; k < ncls; ++k) {
      begin = (begin + 7) / 8 * 8;  // keep every class aligned to the 8-XCD round-robin
      cls_begin[k] = begin;
      cls_next[k] = 0;
      int f = cls_first[k];
      int per = cls_clusters[k];
      if (P.cfg[f].pair == 2) {  // pair mode maps job-major: every member job rounds its own batch groups up to pairs
        per = 0;
        for (int i = 0; i < njobs; ++i)
          if (P.cluster[i] && cls_of[i] == k) per += (P.nbg[i] + 1) / 2;
      }
      begin += P.G[f] * per;
    }
    P.total = begin;
    for (int i = 0; i < njobs; ++i) {
      if (!P.cluster[i]) continue;
      const mgr_scan_job& j = jobs[i];
      ClusterJob& cj = L.job[L.njobs++];
      int ks = j.H / 4;
      size_t img = (size_t)((ks + 3) / 4) * 256;
      cj.Z = j.Z; cj.Up = j.Up; cj.Y = j.Y; cj.R = j.R; cj.G = j.gates; cj.Cs = j.cs;
      cj.ldy = j.ldy; cj.ldr = j.ldr; cj.B = j.B; cj.T = j.T; cj.H = j.H; cj.reverse = j.reverse;
      cj.ks = ks; cj.tpw = P.cfg[i].tpw; cj.nw = P.cfg[i].nw; cj.pair = P.cfg[i].pair;
      cj.G_ = P.G[i]; cj.nbg = P.nbg[i];
      const int k = cls_of[i];
      cj.cls_begin = cls_begin[k];
      cj.cls_nclusters = cls_clusters[k];
      cj.cls_cluster0 = cls_next[k];
      cj.wg_begin = cls_begin[k];  // (pair mode maps job-major inside its class range)
      if (cj.pair == 2) {
        cj.wg_begin = cls_begin[k] + P.G[i] * cls_next[k];
        cls_next[k] += (P.nbg[i] + 1) / 2;
      } else {
        cls_next[k] += P.nbg[i];
      }
      cj.flags = reinterpret_cast<unsigned*>(w);
      w += mgr_align_up((size_t)P.nbg[i] * 64 * sizeof(unsigned), 256);
      cj.xbuf = reinterpret_cast<float*>(w);
      w += mgr_align_up((size_t)P.nbg[i] * 2 * img * sizeof(float), 256);
    }
    L.status = status;
    L.sticky = c->sticky_status;
    L.xcd_local = c->tune[3];
    L.gather_delay = c->tune[6];
    L.ksplit = c->tune[7] == 0;   // tune key 7: 1 = keep the LDS-image step for one-tile-per-wave clusters
    if (c->tune[2]) {  // tune key 2: print the plan
      for (int i = 0; i < L.njobs; ++i)
        fprintf(stderr, "[mgr scan plan] job %d: H=%d ks=%d nw=%d tpw=%d pair=%d G=%d nbg=%d wg_begin=%d\n", i, L.job[i].H,
                L.job[i].ks, L.job[i].nw, L.job[i].tpw, L.job[i].pair, L.job[i].G_, L.job[i].nbg, L.job[i].wg_begin);
      fprintf(stderr, "[mgr scan plan] total %d workgroups, exchange=%d\n", P.total, (int)P.exchange);
    }
    // flags + status must be zero at every launch (epochs count from 1 within the call)
    MGR_HIP(hipMemsetAsync(base, 0, (size_t)(w - base), mgr_stream(c)));
    r = mgr_cluster_launch(c, L, P.total, P.exchange);
    if (r) return r;
  }
  for (int i = 0; i < njobs; ++i) {
    if (P.cluster[i]) continue;
    const mgr_scan_job& j = jobs[i];
    r = mgr_scan_fwd_simple(c, j.Z, j.Up, j.Y, j.ldy, j.R, j.ldr, j.gates, j.cs, j.B, j.T, j.H, j.reverse);
    if (r < 0) return r;
  }
  r = mgr_prof_end(c, MGR_K_SCAN_FWD);
  if (r) return r;
  if (status && c->tune[1]) {  // tune key 1: synchronous status check (tests)
    unsigned st = 0;
    MGR_HIP(hipMemcpyAsync(&st, status, sizeof(st), hipMemcpyDeviceToHost, mgr_stream(c)));
    MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
    MGR_REQUIRE(st == 0, "cluster scan: a bounded spin gave up (status %u)", st);
  }
  return 0;
}

int mgr_lstm_scan_fwd(mgr_ctx* c, const float* Z, const float* Up, float* Y, int ldy, const float* R, int ldr,
                      float* gates, float* cs, int B, int T, int H, int reverse, void* ws, size_t ws_bytes) {
  mgr_scan_job j;
  j.Z = Z; j.Up = Up; j.Y = Y; j.R = R; j.gates = gates; j.cs = cs;
  j.ldy = ldy; j.ldr = ldr; j.B = B; j.T = T; j.H = H; j.reverse = reverse;
  return mgr_lstm_scan_fwd_multi(c, 1, &j, ws, ws_bytes);
}

int mgr_lstm_scan_bwd_multi(mgr_ctx* c, int njobs, const mgr_scan_bwd_job* jobs, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && jobs && njobs > 0 && njobs <= MGR_MAX_SCAN_JOBS, "bad job list");
  MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_scan_bwd_multi_ws_bytes(njobs, jobs), "workspace too small");
  for (int i = 0; i < njobs; ++i) {
    const mgr_scan_bwd_job& j = jobs[i];
    MGR_REQUIRE(j.dY && j.gates && j.cs && j.Up && j.dZ, "job %d: null argument", i);
    MGR_REQUIRE(j.B > 0 && j.T > 0 && j.H > 0 && j.lddy >= j.H, "job %d: bad shape", i);
    MGR_REQUIRE(aligned16(j.gates) && aligned16(j.Up) && aligned16(j.dZ), "job %d: gates/Up/dZ must be 16-byte aligned", i);
  }
  int r = mgr_prof_begin(c, MGR_K_SCAN_BWD);
  if (r) return r;
  const int path = c->tune[MGR_TUNE_SCAN_PATH];
  char* w = reinterpret_cast<char*>(ws);
  char* base = w;
  unsigned* status = reinterpret_cast<unsigned*>(w);
  w += kScanHdrBytes;
  ClusterBwdLaunch L;
  memset(&L, 0, sizeof(L));
  L.status = status;
  L.sticky = c->sticky_status;
  L.xcc = reinterpret_cast<unsigned*>(base + 2048);
  L.xcd_local = c->tune[3];
  int total = 0;
  bool use_cluster[MGR_MAX_SCAN_JOBS];
  // cluster kernel when instantiated and the whole launch is co-resident (two 4-wave workgroups per CU)
  for (int i = 0; i < njobs; ++i) {
    const mgr_scan_bwd_job& j = jobs[i];
    use_cluster[i] = (path == 0 || path == 3) && mgr_cluster_bwd_supported(j.H);
    if (use_cluster[i]) total += ((j.H + 15) / 16) * ((j.B + 15) / 16);
  }
  if (total + 8 * njobs > 2 * c->cu_count)
    for (int i = 0; i < njobs; ++i) use_cluster[i] = false;
  char* wj[MGR_MAX_SCAN_JOBS];
  for (int i = 0; i < njobs; ++i) {
    wj[i] = w;
    w += bwd_job_ws(jobs[i]);
  }
  // classes of identical geometry (the two directions of a layer) share one XCD-interleaved workgroup range
  int cls_of[MGR_MAX_SCAN_JOBS], ncls = 0, cls_first[MGR_MAX_SCAN_JOBS], cls_clusters[MGR_MAX_SCAN_JOBS];
  for (int i = 0; i < njobs; ++i) {
    if (!use_cluster[i]) continue;
    int found = -1;
    for (int k = 0; k < ncls; ++k)
      if (jobs[cls_first[k]].H == jobs[i].H) found = k;
    if (found < 0) {
      found = ncls++;
      cls_first[found] = i;
      cls_clusters[found] = 0;
    }
    cls_of[i] = found;
    cls_clusters[found] += (jobs[i].B + 15) / 16;
  }
  int cls_begin[MGR_MAX_SCAN_JOBS], cls_next[MGR_MAX_SCAN_JOBS], begin = 0;
  for (int k = 0; k < ncls; ++k) {
    begin = (begin + 7) / 8 * 8;
    cls_begin[k] = begin;
    cls_next[k] = 0;
    begin += ((jobs[cls_first[k]].H + 15) / 16) * cls_clusters[k];
  }
  for (int i = 0; i < njobs; ++i) {
    if (!use_cluster[i]) continue;
    const mgr_scan_bwd_job& j = jobs[i];
    ClusterBwdJob& cj = L.job[L.njobs++];
    cj.dY = j.dY; cj.gates = j.gates; cj.cs = j.cs; cj.Up = j.Up; cj.dZ = j.dZ;
    cj.lddy = j.lddy; cj.B = j.B; cj.T = j.T; cj.H = j.H; cj.reverse = j.reverse;
    cj.G_ = (j.H + 15) / 16; cj.nbg = (j.B + 15) / 16;
    const int k = cls_of[i];
    cj.cls_begin = cls_begin[k];
    cj.cls_nclusters = cls_clusters[k];
    cj.cls_cluster0 = cls_next[k];
    cj.wg_begin = cls_begin[k];
    cls_next[k] += cj.nbg;
    cj.xbuf = reinterpret_cast<float*>(wj[i]);
  }
  if (L.njobs > 0) {
    MGR_HIP(hipMemsetAsync(base, 0, (size_t)(w - base), mgr_stream(c)));
    r = mgr_cluster_bwd_launch(c, L, begin);
    if (r) return r;
  }
  for (int i = 0; i < njobs; ++i) {
    if (use_cluster[i]) continue;
    const mgr_scan_bwd_job& j = jobs[i];
    r = 0;
    if (path != 1) r = mgr_scan_bwd_mfma(c, j.dY, j.lddy, j.gates, j.cs, j.Up, j.dZ, j.B, j.T, j.H, j.reverse);
    if (r == 0) {
      float* UpT = reinterpret_cast<float*>(wj[i]);
      r = mgr_transpose(c, j.Up, UpT, j.H, 4 * j.H);
      if (r) return r;
      r = mgr_scan_bwd_simple(c, j.dY, j.lddy, j.gates, j.cs, UpT, j.dZ, j.B, j.T, j.H, j.reverse);
    }
    if (r < 0) return r;
  }
  r = mgr_prof_end(c, MGR_K_SCAN_BWD);
  if (r) return r;
  if (L.njobs > 0 && c->tune[1]) {
    unsigned st = 0;
    MGR_HIP(hipMemcpyAsync(&st, status, sizeof(st), hipMemcpyDeviceToHost, mgr_stream(c)));
    MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
    MGR_REQUIRE(st == 0, "cluster BPTT: a bounded spin gave up (status %u)", st);
  }
  return 0;
}

int mgr_lstm_scan_bwd(mgr_ctx* c, const float* dY, int lddy, const float* gates, const float* cs, const float* Up,
                      float* dZ, int B, int T, int H, int reverse, void* ws, size_t ws_bytes) {
  mgr_scan_bwd_job j;
  j.dY = dY; j.gates = gates; j.cs = cs; j.Up = Up; j.dZ = dZ;
  j.lddy = lddy; j.B = B; j.T = T; j.H = H; j.reverse = reverse;
  return mgr_lstm_scan_bwd_multi(c, 1, &j, ws, ws_bytes);
}

}  // extern "C"
