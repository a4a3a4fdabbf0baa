// Skeletal feature extraction (reference skeletal_network/skeletal_feature_extraction.py:24-215).
// One thread per frame of the WHOLE frame table (the reference shifts "previous position / velocity" over the
// concatenated table, not per file, and zeroes only the first five rows): reads the 12 joint coordinates of frames
// i, i-1, i-2 and writes 23 features.  HBM-bound: 96 B read (+ neighbours from L2) and 184 B written per frame, fp64
// like the pandas/numpy original.  Sums and products are kept unfused (no FMA contraction) so the radicands are
// bit-identical to numpy's; the device sqrt / atan2 may differ from the host libm's in the last ulp.
#include "common.h"

namespace {

constexpr int NJ = 12;   // lhX lhY rhX rhY leX leY reX reY hipX hipY shcX shcY
constexpr int NF = 23;

__device__ __forceinline__ double dist2(double ax, double ay, double bx, double by) {
  double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by);
  return __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
}

// velocity of joint j (0..3 = lh, rh, le, re) at frame i: |p_i - p_{i-1}|, rows 0..4 forced to 0 (:98-99)
__device__ __forceinline__ double vel(const double* __restrict__ J, long long i, int j) {
  if (i < 5) return 0.0;
  const double* p = J + i * NJ + 2 * j;
  const double* q = p - NJ;
  return dist2(p[0], p[1], q[0], q[1]);
}

__global__ __launch_bounds__(256) void k_skeletal_features(const double* __restrict__ J, long long N, double* __restrict__ out) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double* p = J + i * NJ;
  double* o = out + i * NF;
  const double lhx = p[0], lhy = p[1], rhx = p[2], rhy = p[3], lex = p[4], ley = p[5], rex = p[6], rey = p[7];
  const double hpx = p[8], hpy = p[9], scx = p[10], scy = p[11];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double v = vel(J, i, j);
    o[j] = v;
    // acceleration = v_i - v_{i-1}, rows 0..4 forced to 0 (:121-126)
    o[4 + j] = (i < 5) ? 0.0 : __dsub_rn(v, vel(J, i - 1, j));
  }
  o[8] = dist2(lhx, lhy, rhx, rhy);      // hands_d
  o[9] = dist2(lhx, lhy, hpx, hpy);      // lh_hip_d
  o[10] = dist2(rhx, rhy, hpx, hpy);     // rh_hip_d
  o[11] = dist2(lex, ley, hpx, hpy);     // le_hip_d
  o[12] = dist2(rex, rey, hpx, hpy);     // re_hip_d
  o[13] = dist2(lhx, lhy, scx, scy);     // lh_shc_d
  o[14] = dist2(rhx, rhy, scx, scy);     // rh_shc_d
  o[15] = dist2(lex, ley, scx, scy);     // le_shc_d
  o[16] = dist2(rex, rey, scx, scy);     // re_shc_d
  o[17] = atan2(__dsub_rn(lhy, hpy), __dsub_rn(lhx, hpx));   // lh_hip_ang
  o[18] = atan2(__dsub_rn(rhy, hpy), __dsub_rn(rhx, hpx));   // rh_hip_ang
  o[19] = atan2(__dsub_rn(lhy, scy), __dsub_rn(lhx, scx));   // lh_shc_ang
  o[20] = atan2(__dsub_rn(rhy, scy), __dsub_rn(rhx, scx));   // rh_shc_ang
  o[21] = atan2(__dsub_rn(lhy, ley), __dsub_rn(lhx, lex));   // lh_el_ang
  o[22] = atan2(__dsub_rn(rhy, rey), __dsub_rn(rhx, rex));   // rh_el_ang
}

}  // namespace

extern "C" int mgr_skeletal_features(mgr_ctx* c, const double* joints, size_t n_frames, double* out) {
  MGR_REQUIRE(c && joints && out, "null argument");
  if (n_frames == 0) return 0;
  MGR_REQUIRE(n_frames < (1ull << 40), "too many frames");
  mgr_prof_begin(c, MGR_K_MISC);
  hipLaunchKernelGGL(k_skeletal_features, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, mgr_stream(c), joints,
                     (long long)n_frames, out);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_MISC);
  return 0;
}
