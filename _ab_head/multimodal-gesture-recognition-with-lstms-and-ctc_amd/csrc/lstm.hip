// C-ABI entry points of the LSTM recurrence (K3, K7-scan): pick the kernel family for each shape.
//   forward:  H/4 with an instantiation (H = 100, 128, 300, 500 and test sizes) -> lstm_cluster.hip: persistent
//             weight-stationary MFMA kernel, one CU per batch group (H <= 128) or clusters of CUs exchanging h_t
//   backward: H <= 128 -> lstm_mfma.hip (single-CU weight-stationary MFMA)
//   anything else (H <= 1024) -> lstm_simple.hip (U streamed from L2; correctness fallback)
#include <algorithm>
#include <cstddef>

#include "common.h"
#include "lstm_cluster.h"

int mgr_scan_fwd_simple(mgr_ctx*, const float*, const float*, float*, int, const float*, int, float*, float*, int, int, int, int);
int mgr_scan_bwd_simple(mgr_ctx*, const float*, int, const float*, const float*, const float*, float*, int, int, int, int);
int mgr_scan_bwd_mfma_multi(mgr_ctx*, int, const mgr_scan_bwd_job*);
int mgr_scan_bwd_cu16_multi(mgr_ctx*, int, const mgr_scan_bwd_job*);   // lstm_cu_bwd.hip: one CU per (direction, 16-sample group), split-f16

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct Cfg {
  int nw, tpw;
};
// candidate (active waves, tiles per wave) configurations
constexpr int NCFG = 4;
const Cfg kCfgs[NCFG] = {{4, 1}, {8, 1}, {8, 2}, {8, 4}};

struct Plan {
  bool cluster[MGR_MAX_SCAN_JOBS];
  Cfg cfg[MGR_MAX_SCAN_JOBS];
  int G[MGR_MAX_SCAN_JOBS], nbg[MGR_MAX_SCAN_JOBS], wgs[MGR_MAX_SCAN_JOBS];
  int total;
  bool any, exchange;
};

size_t job_ws(const mgr_scan_job& j) {
  int ks = j.H / 4;
  size_t img = (size_t)((ks + 7) / 8) * 512;   // 1 KiB per 16 units, in whole K-blocks of 32 (the split-f16 step's image)
  int nbg = (j.B + 15) / 16;
  return mgr_align_up((size_t)nbg * 2 * img * sizeof(float), 256);
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
    bool feas = true, exch = false;
    for (int k = 0; k < n; ++k) {
      cur[k] = x % NCFG;
      x /= NCFG;
      const mgr_scan_job& j = jobs[idx[k]];
      Cfg f = kCfgs[cur[k]];
      int ks = j.H / 4;
      if (!mgr_cluster_supported(ks, f.tpw)) feas = false;
      if (path == 3 && cur[k] != 0) feas = false;
      if (path == 4 && cur[k] != 1) feas = false;
      if (path == 2 && (f.nw * f.tpw < ks)) feas = false;  // force single-CU (no exchange)
      int tiles = f.nw * f.tpw;
      int G = (ks + tiles - 1) / tiles;
      if (G > 64) feas = false;
      if (G > 1) exch = true;
      int nbg = (j.B + 15) / 16;
      // (classes of jobs are laid out on workgroup ranges rounded up to a multiple of 8 - the XCD count - at launch; count
      // every job rounded up so that a plan accepted here always passes the launcher's co-residency check)
      total += (G * nbg + 7) / 8 * 8;
      // per-step estimate in cycles: MFMA chain per SIMD (+15% issue overhead) + cell update + exchange / barrier
      int tiles_here = std::min(tiles, ks);
      int per_simd = (tiles_here + 3) / 4;
      long t = (long)per_simd * ks * 37 + 700 + (G > 1 ? 3300 : 400);
      if (total > c->cu_count) t += (long)per_simd * ks * 12;  // a second workgroup on the CU competes for the MFMA pipe part of the time
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
    size_t lds2 = (size_t)2 * ((maxks + 3) / 4) * 1024;
    int capacity = (all4 && lds2 <= 80 * 1024) ? 2 * c->cu_count : c->cu_count;
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
    P.wgs[i] = P.G[i] * P.nbg[i];
    P.total += P.wgs[i];
    if (P.G[i] > 1) P.exchange = true;
  }
  P.any = true;
}

}  // namespace

// Lay the cluster jobs of one launch out on workgroup ranges: jobs with identical geometry (the two directions of a layer)
// form a class that shares one contiguous range, every class starts on a multiple of 8 (the XCD round-robin).
template <class SameFn, class SizeFn>
static int layout_classes(int njobs, const bool* use, SameFn same, SizeFn G_of, const int* nbg, int* cls_begin_of, int* cls_clusters_of,
                          int* cls_cluster0_of, bool octets = false, int* cls_rot_of = nullptr) {
  int cls_of[MGR_MAX_SCAN_JOBS], ncls = 0, cls_first[MGR_MAX_SCAN_JOBS], cls_clusters[MGR_MAX_SCAN_JOBS];
  for (int i = 0; i < njobs; ++i) {
    if (!use[i]) continue;
    int found = -1;
    for (int k = 0; k < ncls; ++k)
      if (same(cls_first[k], i)) found = k;
    if (found < 0) {
      found = ncls++;
      cls_first[found] = i;
      cls_clusters[found] = 0;
    }
    cls_of[i] = found;
    cls_clusters[found] += nbg[i];
  }
  int cls_begin[MGR_MAX_SCAN_JOBS], cls_next[MGR_MAX_SCAN_JOBS], cls_rot[MGR_MAX_SCAN_JOBS], begin = 0, lanes = 0;
  for (int k = 0; k < ncls; ++k) {
    begin = (begin + 7) / 8 * 8;
    cls_begin[k] = begin;
    cls_next[k] = 0;
    cls_rot[k] = lanes & 7;   // (XCD-local layout) this class's first cluster takes the lane after the previous class's last
    lanes += cls_clusters[k];
    begin += G_of(cls_first[k]) * (octets ? (cls_clusters[k] + 7) / 8 * 8 : cls_clusters[k]);
  }
  for (int i = 0; i < njobs; ++i) {
    if (!use[i]) continue;
    const int k = cls_of[i];
    cls_begin_of[i] = cls_begin[k];
    cls_clusters_of[i] = cls_clusters[k];
    cls_cluster0_of[i] = cls_next[k];
    if (cls_rot_of) cls_rot_of[i] = octets ? cls_rot[k] : 0;
    cls_next[k] += nbg[i];
  }
  return begin;   // grid size
}

static int check_launch_status(mgr_ctx* c, unsigned* status, const char* what) {
  unsigned st = 0;   // tune key 1: synchronous give-up check (tests)
  MGR_HIP(hipMemcpyAsync(&st, status, sizeof(st), hipMemcpyDeviceToHost, mgr_stream(c)));
  MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
  MGR_REQUIRE(st == 0, "%s: a bounded spin gave up (status %u)", what, st);
  return 0;
}

extern "C" {

int mgr_tune(mgr_ctx* c, int key, int value) {
  MGR_REQUIRE(c && key >= 0 && key < MGR_TUNE_COUNT, "bad tune key");
  c->tune[key] = value;
  return 0;
}

int mgr_tune_get(mgr_ctx* c, int key, int* value) {
  MGR_REQUIRE(c && value && key >= 0 && key < MGR_TUNE_COUNT, "bad tune key");
  *value = c->tune[key];
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
  size_t s = kScanHdrBytes;
  for (int i = 0; i < njobs; ++i) s += job_ws(jobs[i]);
  return s;
}

// what a mgr_scan_launch_opts says, read through its own struct_size (members beyond it are zero)
static void read_opts(const mgr_scan_launch_opts* o, int* form, unsigned** seq_out) {
  *form = 0;
  *seq_out = nullptr;
  if (!o) return;
  if (o->struct_size >= offsetof(mgr_scan_launch_opts, form) + sizeof(o->form)) *form = o->form;
  if (o->struct_size >= offsetof(mgr_scan_launch_opts, seq_out) + sizeof(o->seq_out)) *seq_out = o->seq_out;
}

int mgr_abi_struct_sizes(unsigned out[4]) {
  MGR_REQUIRE(out, "null argument");
  out[0] = (unsigned)sizeof(mgr_scan_job);
  out[1] = (unsigned)sizeof(mgr_scan_bwd_job);
  out[2] = (unsigned)sizeof(mgr_scan_launch_opts);
  out[3] = MGR_ABI_REVISION;
  return 0;
}

int mgr_lstm_scan_fwd_multi(mgr_ctx* c, int njobs, const mgr_scan_job* jobs, void* ws, size_t ws_bytes) {
  return mgr_lstm_scan_fwd_multi_ex(c, njobs, jobs, ws, ws_bytes, nullptr);
}

int mgr_lstm_scan_fwd_multi_ex(mgr_ctx* c, int njobs, const mgr_scan_job* jobs, void* ws, size_t ws_bytes,
                               const mgr_scan_launch_opts* opts) {
  MGR_REQUIRE(c && jobs && njobs > 0 && njobs <= MGR_MAX_SCAN_JOBS, "bad job list");
  int form;
  unsigned* seq_out;
  read_opts(opts, &form, &seq_out);
  MGR_REQUIRE(form >= MGR_SCAN_FORM_AUTO && form <= MGR_SCAN_FORM_FUSED_ANY, "unknown scan form %d", form);
  if (seq_out) *seq_out = MGR_SEQ_NONE;   // (until a launch of this call enters the residency ledger)
  // the form of the split-f16 K-split launches: the caller's, or tune key 4 (0 plain, 2 pair, 3 fused)
  const int key4 = form == MGR_SCAN_FORM_AUTO ? c->tune[4] : form == MGR_SCAN_FORM_PLAIN ? 0 : form == MGR_SCAN_FORM_FUSED_ANY ? 3 : form;
  const bool fused_any = form == MGR_SCAN_FORM_FUSED_ANY;
  for (int i = 0; i < njobs; ++i) {
    const mgr_scan_job& j = jobs[i];
    MGR_REQUIRE(j.Z && j.Up && j.Y, "job %d: null argument", i);
    MGR_REQUIRE(j.B > 0 && j.T > 0 && j.H > 0 && j.ldy >= j.H && (!j.R || j.ldr >= j.H), "job %d: bad shape", i);
    MGR_REQUIRE(aligned16(j.Z) && aligned16(j.Up) && (!j.gates || aligned16(j.gates)), "job %d: Z/Up/gates must be 16-byte aligned", i);
    MGR_REQUIRE(!j.YT || (aligned16(j.YT) && j.ldt % 4 == 0 && j.ldt >= (j.T + 31) / 32 * 32 && j.ytb % 4 == 0 && j.ytb >= (long long)j.H * j.ldt),
                "job %d: transposed output needs a 16-byte aligned YT, ldt %% 4 == 0, ldt >= T rounded up to 32, ytb >= H * ldt", i);
    MGR_REQUIRE(!j.YT || !j.yt_split || j.ldt % 8 == 0, "job %d: split rows need ldt %% 8 == 0", i);
  }
  int hmax = 0;
  for (int i = 0; i < njobs; ++i) hmax = jobs[i].H > hmax ? jobs[i].H : hmax;
  const int family = hmax > 128 ? MGR_K_SCAN_FWD : MGR_K_SCAN_FWD_NARROW;
  int r = mgr_prof_begin(c, family);
  if (r) return r;
  Plan P;
  make_plan(c, njobs, jobs, P);
  if (P.any && (!ws || ws_bytes < mgr_lstm_scan_multi_ws_bytes(njobs, jobs))) {
    // no workspace for the exchange: degrade to the non-cluster families
    for (int i = 0; i < njobs; ++i) P.cluster[i] = false;
    P.any = false;
  }
  unsigned* status = nullptr;
  bool yt_done[MGR_MAX_SCAN_JOBS] = {};
  if (P.any) {
    ClusterLaunch L;
    memset(&L, 0, sizeof(L));
    char* w = reinterpret_cast<char*>(ws);
    status = reinterpret_cast<unsigned*>(w);
    char* base = w;
    w += kScanHdrBytes;
    // K-split launches (every job a 4-wave, one-tile-per-wave cluster with an exchange) lay their clusters out XCD-locally
    // (tune key 3 = 1 turns that off); the table of workgroup XCD ids lives in the launch header
    // the K-split step addresses Z and the residual input with 32-bit byte offsets per lane (LDS-DMA prefetch): launches with a
    // larger tensor take the LDS-image step
    bool ks_ok = c->tune[7] == 0;
    for (int i = 0; i < njobs && ks_ok; ++i) {
      if (!P.cluster[i]) continue;
      const mgr_scan_job& j = jobs[i];
      const size_t zb = (size_t)j.B * j.T * 4 * j.H * sizeof(float), rb = j.R ? (size_t)j.B * j.T * j.ldr * sizeof(float) : 0;
      ks_ok = zb < ((size_t)1 << 32) && rb < ((size_t)1 << 32);
    }
    // Pair form of the split-f16 K-split step (lstm_cluster.hip, cluster_run_k16p): two 16-sample groups per workgroup, ONE workgroup
    // per CU (config F's encoder depths: 204 workgroups instead of 408).  Bit-identical, tested - and NOT the default: measured 3.5 us
    // per pair of steps against 2.2 for the two-workgroups-per-CU launch (one wave runs both groups' instruction streams one after the
    // other; two waves per SIMD interleave them), 30.9 against 21.5 ms per training step (profiles/r05_scan_probes.txt).
    // tune key 4: 2 = take it whenever the launch qualifies.
    int nbg16[MGR_MAX_SCAN_JOBS];
    for (int i = 0; i < njobs; ++i) nbg16[i] = P.cluster[i] ? P.nbg[i] : 0;
    bool pair = ks_ok && P.exchange && c->tune[14] == 0 && key4 == 2;
    {
      int unpaired = 0, most = 0;
      for (int i = 0; i < njobs && pair; ++i) {
        if (!P.cluster[i]) continue;
        pair = P.cfg[i].nw == 4 && P.cfg[i].tpw == 1 && P.G[i] > 1 && mgr_cluster_ks_supported(jobs[i].H / 4);
        unpaired += P.G[i] * P.nbg[i];
        most = P.nbg[i] > most ? P.nbg[i] : most;
      }
      pair = pair && most >= 2;
      (void)unpaired;
      if (pair)
        for (int i = 0; i < njobs; ++i)
          if (P.cluster[i]) P.nbg[i] = (P.nbg[i] + 1) / 2;     // clusters of the job from here on
    }
    // Fused form (lstm_cluster.hip, k_scan_cluster_k16f): 8-wave workgroups that run TWO unit groups of their cluster, one workgroup
    // per CU (config F's encoder depths: 208 workgroups on 208 CUs, 48 CUs left to the other stream).  tune key 4: 3.
    bool fused = !pair && ks_ok && P.exchange && c->tune[14] == 0 && key4 == 3 && c->tune[3] == 0;
    for (int i = 0; i < njobs && fused; ++i) {
      if (!P.cluster[i]) continue;
      fused = P.cfg[i].nw == 4 && P.cfg[i].tpw == 1 && P.G[i] > 1 && mgr_cluster_ks_supported(jobs[i].H / 4);
    }
    {   // (only launches that do not fit one workgroup per CU as they are: the fusion layer's 56 workgroups stay what they are)
      int unf = 0;
      for (int i = 0; i < njobs; ++i)
        if (P.cluster[i]) unf += P.G[i] * P.nbg[i];
      fused = fused && (fused_any || unf > c->cu_count);
    }
    auto members = [&](int i) { return fused ? (P.G[i] + 1) / 2 : P.G[i]; };   // workgroups per cluster
    bool xcd = c->tune[3] == 0 && ks_ok && P.exchange;
    {
      int tot = 0, live_x = 0;
      for (int i = 0; i < njobs && xcd; ++i) {
        if (!P.cluster[i]) continue;
        xcd = P.cfg[i].nw == 4 && P.cfg[i].tpw == 1 && P.G[i] > 1 && mgr_cluster_ks_supported(jobs[i].H / 4);
      }
      // (octets of clusters: the grid may grow; it must still fit the chip and the header's table)
      if (xcd) {
        int h[MGR_MAX_SCAN_JOBS], n = 0;
        for (int i = 0; i < njobs; ++i) {
          if (!P.cluster[i]) continue;
          bool seen = false;
          for (int k = 0; k < n; ++k) seen = seen || h[k] == jobs[i].H;
          if (seen) continue;
          h[n++] = jobs[i].H;
          int clusters = 0;
          for (int k = i; k < njobs; ++k)
            if (P.cluster[k] && jobs[k].H == jobs[i].H) clusters += P.nbg[k];
          tot += members(i) * ((clusters + 7) / 8 * 8);
          live_x += members(i) * clusters;
        }
        xcd = live_x <= ((pair || fused) ? 1 : 2) * c->cu_count && tot <= 2 * c->cu_count && (size_t)tot * sizeof(unsigned) <= kScanHdrBytes - 256;
      }
    }
    int cb[MGR_MAX_SCAN_JOBS], cn[MGR_MAX_SCAN_JOBS], c0[MGR_MAX_SCAN_JOBS], cr[MGR_MAX_SCAN_JOBS];
    // workgroups that really run (a class laid out in octets of clusters has empty ids when its cluster count is no multiple of 8:
    // those workgroups count themselves in and return): what co-residency and the admission ledger are about
    fused = fused && xcd;   // (the fused kernel understands the octet layout only)
    int live = 0;
    for (int i = 0; i < njobs; ++i)
      if (P.cluster[i]) live += members(i) * P.nbg[i];
    P.total = layout_classes(
        njobs, P.cluster,
        [&](int a, int b) { return jobs[a].H == jobs[b].H && P.cfg[a].nw == P.cfg[b].nw && P.cfg[a].tpw == P.cfg[b].tpw; },
        [&](int a) { return members(a); }, P.nbg, cb, cn, c0, xcd, cr);
    L.xcd_local = xcd;
    for (int i = 0; i < njobs; ++i) {
      if (!P.cluster[i]) continue;
      const mgr_scan_job& j = jobs[i];
      ClusterJob& cj = L.job[L.njobs++];
      int ks = j.H / 4;
      size_t img = (size_t)((ks + 7) / 8) * 512;   // (job_ws)
      cj.Z = j.Z; cj.Up = j.Up; cj.Y = j.Y; cj.R = j.R; cj.G = j.gates; cj.Cs = j.cs;
      cj.ldy = j.ldy; cj.ldr = j.ldr; cj.B = j.B; cj.T = j.T; cj.H = j.H; cj.reverse = j.reverse;
      cj.ks = ks; cj.tpw = P.cfg[i].tpw; cj.nw = P.cfg[i].nw;
      cj.G_ = P.G[i]; cj.nbg = P.nbg[i]; cj.nbg16 = nbg16[i];
      cj.cls_begin = cb[i]; cj.cls_nclusters = cn[i]; cj.cls_cluster0 = c0[i]; cj.cls_rot = cr[i];
      cj.xbuf = reinterpret_cast<float*>(w);
      w += mgr_align_up((size_t)nbg16[i] * 2 * img * sizeof(float), 256);
    }
    L.pair = pair ? 1 : 0;
    L.fused = fused ? 1 : 0;
    L.live_wgs = live;
    // tune key 7: 0 = K-split step for one-tile-per-wave clusters, 1 = LDS-image step for every cluster
    L.ksplit = ks_ok ? 1 : 0;
    L.split16 = c->tune[14] == 0;   // tune key 14: 1 = f32 MFMA in the K-split step
    // transposed outputs: the K-split kernel writes them itself; everything else gets a transpose behind the scans (below)
    if (mgr_cluster_uses_ks(L, P.exchange)) {
      int k = 0;
      for (int i = 0; i < njobs; ++i) {
        if (!P.cluster[i]) continue;
        ClusterJob& cj = L.job[k++];
        cj.YT = jobs[i].YT; cj.ytb = jobs[i].ytb; cj.ldt = jobs[i].ldt; cj.yt_split = jobs[i].yt_split;
        yt_done[i] = jobs[i].YT != nullptr;
      }
    }
    if (c->tune[2]) {  // tune key 2: print the plan
      for (int i = 0; i < L.njobs; ++i)
        fprintf(stderr, "[mgr scan plan] job %d: H=%d ks=%d nw=%d tpw=%d G=%d nbg=%d wg_begin=%d\n", i, L.job[i].H,
                L.job[i].ks, L.job[i].nw, L.job[i].tpw, L.job[i].G_, L.job[i].nbg, L.job[i].cls_begin);
      fprintf(stderr, "[mgr scan plan] total %d workgroups, exchange=%d\n", P.total, (int)P.exchange);
    }
    int waves, per_cu;
    mgr_cluster_geometry(L, P.exchange, &waves, &per_cu);
    L.cm.status = status;
    L.cm.sticky = mgr_status_block(c);
    L.cm.resident = c->sticky_status + 1;
    L.cm.total_wgs = P.total;
    // (a launch without an exchange spins on nobody: it needs no place in the ledger and is never ordered behind one)
    if (P.exchange) {
      r = mgr_persist_admit(c, live, waves, per_cu, L.fused, &L.cm.seq);
      if (r) return r;
      if (seq_out) *seq_out = L.cm.seq;
    }
    // exchange slots + status must be zero at every launch (epochs count from 1 within the call)
    MGR_HIP(hipMemsetAsync(base, 0, (size_t)(w - base), mgr_stream(c)));
    r = mgr_cluster_launch(c, L, P.total, P.exchange);
    if (r) return r;
    if (P.exchange) {
      r = mgr_persist_commit(c, live, waves, per_cu, L.fused);
      if (r) return r;
    }
  }
  for (int i = 0; i < njobs; ++i) {
    if (P.cluster[i]) continue;
    const mgr_scan_job& j = jobs[i];
    r = mgr_scan_fwd_simple(c, j.Z, j.Up, j.Y, j.ldy, j.R, j.ldr, j.gates, j.cs, j.B, j.T, j.H, j.reverse);
    if (r < 0) return r;
  }
  for (int i = 0; i < njobs; ++i) {   // transposed outputs the scan kernel did not write itself
    const mgr_scan_job& j = jobs[i];
    if (!j.YT || yt_done[i]) continue;
    r = j.yt_split ? mgr_transpose_bt_split_strided(c, j.Y, j.ldy, j.YT, j.ldt, j.ytb, j.ldt, j.B, j.T, j.H)
                   : mgr_transpose_bt_strided(c, j.Y, j.ldy, j.YT, j.ldt, j.ytb, j.ldt, j.B, j.T, j.H);
    if (r) return r;
  }
  r = mgr_prof_end(c, family);
  if (r) return r;
  if (status && c->tune[1]) return check_launch_status(c, status, "cluster scan");
  return 0;
}

int mgr_lstm_scan_fwd(mgr_ctx* c, const float* Z, const float* Up, float* Y, int ldy, const float* R, int ldr,
                      float* gates, float* cs, int B, int T, int H, int reverse, void* ws, size_t ws_bytes) {
  mgr_scan_job j;
  memset(&j, 0, sizeof(j));
  j.Z = Z; j.Up = Up; j.Y = Y; j.R = R; j.gates = gates; j.cs = cs;
  j.ldy = ldy; j.ldr = ldr; j.B = B; j.T = T; j.H = H; j.reverse = reverse;
  return mgr_lstm_scan_fwd_multi(c, 1, &j, ws, ws_bytes);
}

int mgr_lstm_scan_bwd_multi(mgr_ctx* c, int njobs, const mgr_scan_bwd_job* jobs, void* ws, size_t ws_bytes) {
  return mgr_lstm_scan_bwd_multi_ex(c, njobs, jobs, ws, ws_bytes, nullptr);
}

int mgr_lstm_scan_bwd_multi_ex(mgr_ctx* c, int njobs, const mgr_scan_bwd_job* jobs, void* ws, size_t ws_bytes,
                               const mgr_scan_launch_opts* opts) {
  MGR_REQUIRE(c && jobs && njobs > 0 && njobs <= MGR_MAX_SCAN_JOBS, "bad job list");
  int form;
  unsigned* seq_out;
  read_opts(opts, &form, &seq_out);
  MGR_REQUIRE(form >= MGR_BPTT_FORM_AUTO && form <= MGR_BPTT_FORM_SINGLE_CU, "unknown BPTT form %d", form);
  if (seq_out) *seq_out = MGR_SEQ_NONE;
  const bool want_fused = form == MGR_BPTT_FORM_FUSED || form == MGR_BPTT_FORM_FUSED_DIRECT;
  // 0 trimmed, 1 yielding, 2 direct gather (the fused forms: the trimmed step / the direct gather)
  const int key16 = form == MGR_BPTT_FORM_AUTO ? c->tune[16]
                    : (form == MGR_BPTT_FORM_FUSED || form == MGR_BPTT_FORM_SINGLE_CU) ? 0 : form == MGR_BPTT_FORM_FUSED_DIRECT ? 2 : form - 1;
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
  // The single-CU split-f16 form (lstm_cu_bwd.hip): asked for by the caller (or tune key 19 = 1), taken when the jobs are the directions
  // of ONE narrow layer (same shape, 16 < H <= 128) and the f16 matrix pipe is in use; no inter-CU exchange, no ledger entry
  if ((form == MGR_BPTT_FORM_SINGLE_CU || (form == MGR_BPTT_FORM_AUTO && c->tune[19] == 1)) && c->tune[14] == 0 && (path == 0 || path == 3)) {
    r = mgr_scan_bwd_cu16_multi(c, njobs, jobs);
    if (r < 0) return r;
    if (r == 1) {
      for (int i = 0; i < njobs; ++i) {   // (the row maxima of dZ: a reduction pass, as behind every kernel that does not keep them itself)
        const mgr_scan_bwd_job& j = jobs[i];
        if (!j.dzmax && !j.dbsum) continue;
        r = mgr_rowmax_bt(c, j.dZ, 4 * j.H, j.T, j.B, j.dzmax, j.dbsum);
        if (r) return r;
      }
      return mgr_prof_end(c, MGR_K_SCAN_BWD);
    }
  }
  char* w = reinterpret_cast<char*>(ws);
  char* base = w;
  unsigned* status = reinterpret_cast<unsigned*>(w);
  w += kScanHdrBytes;
  ClusterBwdLaunch L;
  memset(&L, 0, sizeof(L));
  int total = 0, nbg[MGR_MAX_SCAN_JOBS];
  bool use_cluster[MGR_MAX_SCAN_JOBS];
  // cluster kernel when instantiated and the whole launch is co-resident (two 4-wave workgroups per CU)
  for (int i = 0; i < njobs; ++i) {
    const mgr_scan_bwd_job& j = jobs[i];
    nbg[i] = (j.B + 15) / 16;
    // (the cluster kernel addresses the saved state with 32-bit byte offsets per lane: LDS-DMA prefetch)
    const bool small = (size_t)j.B * j.T * j.H * 16 < ((size_t)1 << 32) && (size_t)j.B * j.T * j.lddy * 4 < ((size_t)1 << 32);
    use_cluster[i] = (path == 0 || path == 3) && mgr_cluster_bwd_supported(j.H) && small;
    if (use_cluster[i]) total += ((j.H + 15) / 16) * nbg[i];
  }
  if (total + 8 * njobs > 2 * c->cu_count)
    for (int i = 0; i < njobs; ++i) use_cluster[i] = false;
  char* wj[MGR_MAX_SCAN_JOBS];
  for (int i = 0; i < njobs; ++i) {
    wj[i] = w;
    w += bwd_job_ws(jobs[i]);
  }
  int cb[MGR_MAX_SCAN_JOBS], cn[MGR_MAX_SCAN_JOBS], c0[MGR_MAX_SCAN_JOBS], cr[MGR_MAX_SCAN_JOBS];
  auto same_h = [&](int a, int b) { return jobs[a].H == jobs[b].H; };
  auto g_of = [&](int a) { return (jobs[a].H + 15) / 16; };
  // XCD-local layout (octets of clusters) where the padded grid still fits the chip and the header's table; tune key 3 = 1: off
  bool xcd = c->tune[3] == 0;
  int grid = layout_classes(njobs, use_cluster, same_h, g_of, nbg, cb, cn, c0, xcd, cr);
  if (xcd && (grid > 2 * c->cu_count || (size_t)grid * sizeof(unsigned) > kScanHdrBytes - 256)) {
    xcd = false;
    grid = layout_classes(njobs, use_cluster, same_h, g_of, nbg, cb, cn, c0, false, cr);
  }
  // Fused form (lstm_cluster_bwd.hip, k_scan_cluster_bwd16_f): asked for by the caller, taken when every job of the launch qualifies -
  // the same clusters with ceil(G / 2) eight-wave members, a CU each
  bool fused = want_fused && xcd;
  for (int i = 0; i < njobs && fused; ++i) fused = use_cluster[i] && jobs[i].H > 16 && jobs[i].H <= 128 && c->tune[14] == 0;
  if (fused) {
    auto gr_of = [&](int a) { return (g_of(a) + 1) / 2; };
    const int gridf = layout_classes(njobs, use_cluster, same_h, gr_of, nbg, cb, cn, c0, true, cr);
    if (gridf <= c->cu_count && (size_t)gridf * sizeof(unsigned) <= kScanHdrBytes - 256)
      grid = gridf;
    else {
      fused = false;
      grid = layout_classes(njobs, use_cluster, same_h, g_of, nbg, cb, cn, c0, true, cr);
    }
  }
  L.xcd_local = xcd;
  L.fused = fused ? 1 : 0;
  for (int i = 0; i < njobs; ++i) {
    if (!use_cluster[i]) continue;
    const mgr_scan_bwd_job& j = jobs[i];
    ClusterBwdJob& cj = L.job[L.njobs++];
    cj.dY = j.dY; cj.gates = j.gates; cj.cs = j.cs; cj.Up = j.Up; cj.dZ = j.dZ; cj.dzmax = j.dzmax; cj.dbsum = j.dbsum;
    cj.lddy = j.lddy; cj.B = j.B; cj.T = j.T; cj.H = j.H; cj.reverse = j.reverse;
    cj.G_ = (j.H + 15) / 16; cj.nbg = nbg[i];
    cj.cls_begin = cb[i]; cj.cls_nclusters = cn[i]; cj.cls_cluster0 = c0[i]; cj.cls_rot = cr[i];
    cj.xbuf = reinterpret_cast<float*>(wj[i]);
  }
  if (L.njobs > 0) {
    int waves, per_cu;
    mgr_cluster_bwd_geometry(c, L, grid, &waves, &per_cu);
    L.cm.status = status;
    L.cm.sticky = mgr_status_block(c);
    L.cm.resident = c->sticky_status + 1;
    L.cm.total_wgs = grid;
    r = mgr_persist_admit(c, grid, waves, per_cu, L.fused, &L.cm.seq);
    if (r) return r;
    if (seq_out) *seq_out = L.cm.seq;
    MGR_HIP(hipMemsetAsync(base, 0, (size_t)(w - base), mgr_stream(c)));
    r = mgr_cluster_bwd_launch(c, L, grid, key16);
    if (r) return r;
    r = mgr_persist_commit(c, grid, waves, per_cu, L.fused);
    if (r) return r;
  }
  // the single-CU kernel takes every job that is left in ONE launch when they share a shape (the two directions of a layer)
  bool mfma_done = false;
  {
    int rest = 0;
    bool same = true;
    mgr_scan_bwd_job left[MGR_MAX_SCAN_JOBS];
    for (int i = 0; i < njobs; ++i) {
      if (use_cluster[i]) continue;
      left[rest] = jobs[i];
      same = same && jobs[i].B == left[0].B && jobs[i].T == left[0].T && jobs[i].H == left[0].H && jobs[i].lddy == left[0].lddy;
      ++rest;
    }
    if (rest > 0 && same && path != 1) {
      r = mgr_scan_bwd_mfma_multi(c, rest, left);
      if (r < 0) return r;
      mfma_done = r == 1;
    }
  }
  for (int i = 0; i < njobs; ++i) {
    if (use_cluster[i] || mfma_done) continue;
    const mgr_scan_bwd_job& j = jobs[i];
    r = 0;
    if (path != 1) r = mgr_scan_bwd_mfma_multi(c, 1, &j);
    if (r == 0) {
      float* UpT = reinterpret_cast<float*>(wj[i]);
      r = mgr_transpose(c, j.Up, UpT, j.H, 4 * j.H);
      if (r) return r;
      r = mgr_scan_bwd_simple(c, j.dY, j.lddy, j.gates, j.cs, UpT, j.dZ, j.B, j.T, j.H, j.reverse);
    }
    if (r < 0) return r;
  }
  for (int i = 0; i < njobs; ++i) {   // the row maxima / sums of dZ where the kernel that ran did not leave them itself
    const mgr_scan_bwd_job& j = jobs[i];
    if ((!j.dzmax && !j.dbsum) || use_cluster[i]) continue;
    r = mgr_rowmax_bt(c, j.dZ, 4 * j.H, j.T, j.B, j.dzmax, j.dbsum);
    if (r) return r;
  }
  r = mgr_prof_end(c, MGR_K_SCAN_BWD);
  if (r) return r;
  if (L.njobs > 0 && c->tune[1]) return check_launch_status(c, status, "cluster BPTT");
  return 0;
}

int mgr_lstm_scan_bwd(mgr_ctx* c, const float* dY, int lddy, const float* gates, const float* cs, const float* Up,
                      float* dZ, int B, int T, int H, int reverse, void* ws, size_t ws_bytes) {
  mgr_scan_bwd_job j;
  j.dY = dY; j.gates = gates; j.cs = cs; j.Up = Up; j.dZ = dZ;
  j.lddy = lddy; j.B = B; j.T = T; j.H = H; j.reverse = reverse;
  j.dzmax = nullptr; j.dbsum = nullptr;
  return mgr_lstm_scan_bwd_multi(c, 1, &j, ws, ws_bytes);
}

}  // extern "C"

// ---- admission of persistent launches -----------------------------------------------------------------------------------
// A persistent scan's workgroups spin on their peers, so ALL of them must be resident at once.  Inside one launch the grid is
// checked against the chip (mgr_cluster_launch).  Across the streams of a context (the encoder scans of step n+1 beside the
// fusion scan / BPTT of step n, engine.py) this ledger does the same: every persistent launch records (workgroups, waves per
// workgroup, workgroups per CU) and an event behind its kernel; a new launch that would not fit beside the launches that may
// still be running is ORDERED BEHIND them (hipStreamWaitEvent) instead of being allowed to dead-lock with them.  Capacity:
// a CU holds two 4-wave or one 8-wave workgroup of these kernels (they use > 128 VGPRs); as soon as any launch needs a CU of its
// own, every workgroup in flight is counted as a whole CU (4-wave workgroups are dealt one per CU first, so each may block one).
// Kernels that are not persistent (GEMMs, ...) leave on their own and need no entry.
int mgr_persist_admit(mgr_ctx* c, int wgs, int waves_per_wg, int per_cu, int fused, unsigned* seq_out) {
  (void)waves_per_wg;
  bool ordered[MGR_MAX_PERSIST] = {};   // launches this one has been put behind
  for (;;) {
    // launches of ONE stream run one after the other: a stream can hold at most its largest launch on the chip at a time
    int per_stream[MGR_NUM_STREAMS] = {}, cus_stream[MGR_NUM_STREAMS] = {};
    int any_excl = per_cu == 1 ? 1 : 0, any_fused = fused ? 1 : 0, oldest = -1;
    for (int i = 0; i < MGR_MAX_PERSIST; ++i) {
      mgr_ctx::Persist& e = c->persist[i];
      if (!e.active || ordered[i] || e.stream == c->cur) continue;   // (same stream: ordered before this launch anyway)
      if (hipEventQuery(e.done) == hipSuccess) {
        e.active = 0;
        continue;
      }
      per_stream[e.stream] = e.wgs > per_stream[e.stream] ? e.wgs : per_stream[e.stream];
      const int cus = (e.wgs + e.per_cu - 1) / e.per_cu;
      cus_stream[e.stream] = cus > cus_stream[e.stream] ? cus : cus_stream[e.stream];
      any_excl |= e.per_cu == 1;
      any_fused |= e.fused;
      if (oldest < 0 || e.seq < c->persist[oldest].seq) oldest = i;
    }
    int shared = wgs, cus = (wgs + per_cu - 1) / per_cu;
    for (int s = 0; s < MGR_NUM_STREAMS; ++s) {
      shared += per_stream[s];
      cus += cus_stream[s];
    }
    // fused scans (8-wave workgroups that fill a CU's register file - this launch, or one in flight): beside them the 4-wave launches are
    // counted by the CUs they need two to a CU - their caller (the engine) starts them once the exclusive launch is resident, so that
    // they do land there.  (A property of the LAUNCHES in the ledger since round 6, not of a tune key that happens to be set.)
    const bool by_cus = any_excl && any_fused;
    const int capacity = any_excl ? c->cu_count : 2 * c->cu_count;
    if (oldest < 0 || (by_cus ? cus : shared) <= capacity) break;
    // does not fit beside what may still be running: run behind the oldest of them, then look again
    MGR_HIP(hipStreamWaitEvent(mgr_stream(c), c->persist[oldest].done, 0));
    ordered[oldest] = true;
    c->persist_serialised += 1;
  }
  *seq_out = ++c->persist_seq;
  return 0;
}

int mgr_persist_commit(mgr_ctx* c, int wgs, int waves_per_wg, int per_cu, int fused) {
  int slot = -1;
  for (int i = 0; i < MGR_MAX_PERSIST && slot < 0; ++i)
    if (!c->persist[i].active) slot = i;
  if (slot < 0) {   // every entry still marked active: retire those that have finished, else reuse the oldest after waiting for it
    int oldest = 0;
    for (int i = 0; i < MGR_MAX_PERSIST; ++i) {
      if (hipEventQuery(c->persist[i].done) == hipSuccess) slot = i;
      if (c->persist[i].seq < c->persist[oldest].seq) oldest = i;
    }
    if (slot < 0) {
      MGR_HIP(hipEventSynchronize(c->persist[oldest].done));
      slot = oldest;
    }
  }
  mgr_ctx::Persist& e = c->persist[slot];
  if (!e.done) MGR_HIP(hipEventCreateWithFlags(&e.done, hipEventDisableTiming));
  MGR_HIP(hipEventRecord(e.done, mgr_stream(c)));
  e.active = 1;
  e.stream = c->cur;
  e.wgs = wgs;
  e.waves = waves_per_wg;
  e.per_cu = per_cu;
  e.fused = fused ? 1 : 0;
  e.seq = c->persist_seq;
  return 0;
}

namespace {
// One lane polls the context's residency words until the launch with sequence number `seq` has all its workgroups on the chip.
// seq_word != nullptr: the number is not known yet when the wait is enqueued - it arrives in a word of page-locked host memory
// that the launch's call fills in (mgr_scan_launch_opts.seq_out); MGR_SEQ_NONE there = no such launch: nothing to wait for.
// counters[0] counts the waits that have ended, counters[1] those that ended by their timeout (mgr_resident_wait_stats).
__global__ void k_wait_resident(const unsigned* resident, const unsigned* ring, const unsigned* seq_word, unsigned seq, unsigned timeout_us,
                                unsigned* counters) {
  const unsigned long long t0 = wall_clock64();   // 100 MHz
  bool expired = false;
  for (;;) {
    if (seq_word && seq == 0u) seq = __hip_atomic_load(seq_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (seq == MGR_SEQ_NONE) break;
    if (seq != 0u) {
      const unsigned* w = ring ? ring + (seq & 15u) : resident;
      if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seq) break;
    }
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 100ull * timeout_us) {   // placement hint only: never a correctness dependency
      expired = true;
      break;
    }
  }
  __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (expired) {
    __hip_atomic_fetch_add(counters + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    counters[2] = seq;                                             // diagnostics of the LAST expired wait: the launch number it was for (0: never
    counters[3] = (unsigned)reinterpret_cast<uintptr_t>(seq_word); // learnt) and the low 32 bits of the word it polled (0: a known-number wait)
  }
}
constexpr int kWaitCounters = 32;   // words [32, 34) of the context's own status block
}  // namespace

extern "C" {

int mgr_stream_wait_next_resident(mgr_ctx* c, int timeout_us) {
  MGR_REQUIRE(c && timeout_us >= 0 && timeout_us <= 100000, "timeout_us must be in [0, 100000]");
  hipLaunchKernelGGL(k_wait_resident, dim3(1), dim3(1), 0, mgr_stream(c), c->sticky_status + 1, nullptr, nullptr, c->persist_seq + 1,
                     (unsigned)timeout_us, c->sticky_status + kWaitCounters);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_stream_wait_resident(mgr_ctx* c, unsigned seq, int timeout_us) {
  MGR_REQUIRE(c && timeout_us >= 0 && timeout_us <= 100000, "timeout_us must be in [0, 100000]");
  MGR_REQUIRE(seq != 0u, "launch numbers start at 1");
  hipLaunchKernelGGL(k_wait_resident, dim3(1), dim3(1), 0, mgr_stream(c), c->sticky_status + 1, c->sticky_status + 16, nullptr, seq,
                     (unsigned)timeout_us, c->sticky_status + kWaitCounters);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_stream_wait_resident_word(mgr_ctx* c, const unsigned* seq_word, int timeout_us) {
  MGR_REQUIRE(c && seq_word && timeout_us >= 0 && timeout_us <= 100000, "null word / timeout_us must be in [0, 100000]");
  hipLaunchKernelGGL(k_wait_resident, dim3(1), dim3(1), 0, mgr_stream(c), c->sticky_status + 1, c->sticky_status + 16, seq_word, 0u,
                     (unsigned)timeout_us, c->sticky_status + kWaitCounters);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_resident_wait_stats(mgr_ctx* c, unsigned out[4]) {
  MGR_REQUIRE(c && out, "null argument");
  MGR_HIP(hipMemcpyAsync(out, c->sticky_status + kWaitCounters, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, mgr_stream(c)));
  MGR_HIP(hipStreamSynchronize(mgr_stream(c)));
  return 0;
}

int mgr_persist_stats(mgr_ctx* c, int* launches, int* serialised) {
  MGR_REQUIRE(c, "null ctx");
  if (launches) *launches = (int)c->persist_seq;
  if (serialised) *serialised = c->persist_serialised;
  return 0;
}

}  // extern "C"
