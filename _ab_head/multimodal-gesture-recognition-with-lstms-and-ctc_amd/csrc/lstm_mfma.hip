// K7-scan: BPTT of one LSTM direction, weight-stationary on the f32 matrix cores, one workgroup per 16-sample
// batch group, for layers whose recurrent matrix fits one CU's register file (H <= 128; the trainable fusion layer).
// It mirrors the forward kernel (lstm_cluster.hip) with  D[unit, sample] += U[unit, gate-col] * dz^T[gate-col, sample]:
// A operand = U rows (stationary in VGPRs), B operand = dz_t from a double-buffered LDS image
// [gate-col/16][gate][sample][unit%4], C/D = dh_rec for 4 consecutive units of the lane's sample.
#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

constexpr int NW = 8;  // waves per workgroup (2 per SIMD)

// Backward.  H units -> MTB = ceil(H/16) tiles of 16 units, one tile per wave; K = 4H packed gate columns,
// i.e. H MFMA k-steps (k-step s = unit s, kk = gate).
struct BwdMfmaJobs {
  const float* dY[MGR_MAX_SCAN_JOBS];
  const float* G[MGR_MAX_SCAN_JOBS];
  const float* Cs[MGR_MAX_SCAN_JOBS];
  const float* Up[MGR_MAX_SCAN_JOBS];
  float* dZ[MGR_MAX_SCAN_JOBS];
  int reverse[MGR_MAX_SCAN_JOBS];
};

template <int H>
__global__ __launch_bounds__(NW * 64) void k_scan_bwd_mfma(BwdMfmaJobs J, int lddy, int B, int T) {
  const float* __restrict__ dY = J.dY[blockIdx.y];
  const float* __restrict__ G = J.G[blockIdx.y];
  const float* __restrict__ Cs = J.Cs[blockIdx.y];
  const float* __restrict__ Up = J.Up[blockIdx.y];
  float* __restrict__ dZ = J.dZ[blockIdx.y];
  const int reverse = J.reverse[blockIdx.y];
  constexpr int N = 4 * H;
  constexpr int MTB = (H + 15) / 16;
  constexpr int QN = (H + 3) / 4;
  constexpr int DS = QN * 4 * 16 * 4;
  static_assert(MTB <= NW, "H too large for the single-CU backward kernel");
  __shared__ __attribute__((aligned(16))) float dzs[2 * DS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int j = lane & 15, uq = lane >> 4;
  const int b0 = blockIdx.x * 16;
  const int b = b0 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  const int tile = wave;
  const bool tvalid = tile < MTB;
  const int u0 = tile * 16 + uq * 4;  // this lane's 4 units

  // A fragment: A[i = lane&15][kk = lane>>4] = Up[unit tile*16+i][4s + kk]
  float uf[H];
  {
    int ur = tile * 16 + j;
    bool rv = tvalid && ur < H;
#pragma unroll
    for (int s = 0; s < H; ++s) uf[s] = rv ? Up[(size_t)ur * N + 4 * s + uq] : 0.f;
  }
  for (int i = tid; i < 2 * DS; i += NW * 64) dzs[i] = 0.f;
  float dcc[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dhr = {0.f, 0.f, 0.f, 0.f};
  const bool lv = tvalid && u0 < H;  // H % 4 == 0: a lane's 4 units are valid together

  // Saved forward state and dY of one time step for this lane's 4 units.  Iteration k (= T-1-n) uses ring[k % 3];
  // the loads for iteration k+2 are issued at iteration k, so c_{t-1} (= the c of iteration k+1) has already landed.
  struct Saved {
    float dy[4];
    float4 g[4];
    f32x4 c;
  };
  Saved r0, r1, r2;
  auto load = [&](Saved& sv, int k) {
    if (lv && k < T) {
      const int n = T - 1 - k;
      const int t = reverse ? T - 1 - n : n;
      const size_t row = (size_t)bc * T + t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sv.dy[e] = dY[row * lddy + u0 + e];
        sv.g[e] = *reinterpret_cast<const float4*>(G + (row * H + u0 + e) * 4);
      }
      sv.c = *reinterpret_cast<const f32x4*>(Cs + row * H + u0);
    }
  };
  load(r0, 0);
  load(r1, 1);
  __syncthreads();
  int cur = 0;

  auto do_step = [&](int k, Saved& use, Saved& prev, Saved& ld) {
    const int n = T - 1 - k;
    const int t = reverse ? T - 1 - n : n;
    const bool has_prev = n > 0;
    load(ld, k + 2);
    float* dn = dzs + (cur ^ 1) * DS;
    if (lv) {
      const size_t row = (size_t)bc * T + t;
      float4 dz[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dh = use.dy[e] + dhr[e];
        float cp = has_prev ? prev.c[e] : 0.f;
        dz[e] = mgr_cell_bwd(dh, use.g[e], use.c[e], cp, dcc[e]);
        if (bvalid) *reinterpret_cast<float4*>(dZ + row * N + (u0 + e) * 4) = dz[e];
      }
      // image [q = unit>>2][kk = gate][j][r = unit&3]; this lane's units u0..u0+3 share q = u0>>2
      const int q = u0 >> 2;
      *reinterpret_cast<f32x4*>(dn + ((q * 4 + 0) * 16 + j) * 4) = (f32x4){dz[0].x, dz[1].x, dz[2].x, dz[3].x};
      *reinterpret_cast<f32x4*>(dn + ((q * 4 + 1) * 16 + j) * 4) = (f32x4){dz[0].y, dz[1].y, dz[2].y, dz[3].y};
      *reinterpret_cast<f32x4*>(dn + ((q * 4 + 2) * 16 + j) * 4) = (f32x4){dz[0].z, dz[1].z, dz[2].z, dz[3].z};
      *reinterpret_cast<f32x4*>(dn + ((q * 4 + 3) * 16 + j) * 4) = (f32x4){dz[0].w, dz[1].w, dz[2].w, dz[3].w};
    }
    __syncthreads();
    cur ^= 1;
    if (tvalid && has_prev) {  // wave-uniform
      const float* db = dzs + cur * DS;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      constexpr int PD = 3;
      f32x4 dbuf[4];
      const float* dlane = db + (uq * 16 + j) * 4;
#pragma unroll
      for (int q = 0; q < PD && q < QN; ++q) dbuf[q] = *reinterpret_cast<const f32x4*>(dlane + q * 256);
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        if (q + PD < QN) dbuf[(q + PD) & 3] = *reinterpret_cast<const f32x4*>(dlane + (q + PD) * 256);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 dv = dbuf[q & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * q + r < H) {
            if (r & 1)
              a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * q + r], dv[r], a1, 0, 0, 0);
            else
              a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * q + r], dv[r], a0, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      dhr = a0 + a1;
    }
  };

  for (int k0 = 0; k0 < T; k0 += 3) {
    do_step(k0, r0, r1, r2);
    if (k0 + 1 < T) do_step(k0 + 1, r1, r2, r0);
    if (k0 + 2 < T) do_step(k0 + 2, r2, r0, r1);
  }
}

}  // namespace

// returns 1 if launched, 0 if the shape has no instantiation, <0 on error.  All jobs (same H, B, T, lddy) go out as ONE launch
// (blockIdx.y = job): the two directions of a layer run side by side instead of one after the other
int mgr_scan_bwd_mfma_multi(mgr_ctx* c, int njobs, const mgr_scan_bwd_job* jobs) {
  BwdMfmaJobs J;
  memset(&J, 0, sizeof(J));
  const int B = jobs[0].B, T = jobs[0].T, H = jobs[0].H, lddy = jobs[0].lddy;
  for (int i = 0; i < njobs; ++i) {
    J.dY[i] = jobs[i].dY; J.G[i] = jobs[i].gates; J.Cs[i] = jobs[i].cs; J.Up[i] = jobs[i].Up; J.dZ[i] = jobs[i].dZ;
    J.reverse[i] = jobs[i].reverse;
  }
  dim3 grid((B + 15) / 16, njobs);
  hipStream_t s = mgr_stream(c);
#define BWD_CASE(HH)                                                                                              \
  case HH:                                                                                                        \
    hipLaunchKernelGGL((k_scan_bwd_mfma<HH>), grid, dim3(NW * 64), 0, s, J, lddy, B, T); \
    break;
  switch (H) {
    BWD_CASE(4)
    BWD_CASE(8)
    BWD_CASE(16)
    BWD_CASE(32)
    BWD_CASE(64)
    BWD_CASE(100)
    BWD_CASE(128)
    default:
      return 0;
  }
#undef BWD_CASE
  MGR_LAUNCH_CHECK();
  return 1;
}
