// K3 / K7-scan: weight-stationary LSTM recurrence on the f32 matrix cores, one workgroup per
// (direction, 16-sample batch group), for layers whose recurrent matrix fits one CU's register file (H <= 128).
//
// Orientation: the MFMA computes  D[gate-col, sample] += U^T[gate-col, k] * h^T[k, sample]  with
// v_mfma_f32_16x16x4_f32, M = 16 packed gate columns = 4 units x (i,f,c,o), N = 16 samples, K = 4 per step.
//   * A operand (U^T fragment) never changes: each wave keeps its tiles' fragments in VGPRs for all T steps.
//   * B operand (h_{t-1}) is read from a double-buffered LDS image laid out [k/16][k%4][sample][(k/4)%4] so one
//     ds_read_b128 feeds four consecutive MFMA k-steps, conflict-free.
//   * C/D layout: lane (sample = lane&15, unit-in-tile = lane>>4) receives the 4 gates of ITS (unit, sample)
//     in its 4 accumulator registers, so the cell update needs no cross-lane traffic at all.
// Z[t] (gate pre-activations from the input projection) is prefetched one step ahead; h_t, the activated gates
// and c_t stream out with fire-and-forget stores; one s_barrier per time step.
//
// The backward kernel mirrors it with  D[unit, sample] += U[unit, gate-col] * dz^T[gate-col, sample].
#include "lstm_common.h"

namespace {

constexpr int NW = 8;  // waves per workgroup (2 per SIMD)

template <int KS, int TPW>
__global__ __launch_bounds__(NW * 64) void k_scan_fwd_mfma(const float* __restrict__ Z, const float* __restrict__ Up,
                                                           float* __restrict__ Y, int ldy, const float* __restrict__ R,
                                                           int ldr, float* __restrict__ G, float* __restrict__ Cs, int B,
                                                           int T, int reverse) {
  constexpr int H = 4 * KS;
  constexpr int N = 4 * H;
  constexpr int MT = KS;  // M tiles of 4 units
  constexpr int QN = (KS + 3) / 4;
  constexpr int HS = QN * 4 * 16 * 4;  // floats per h image
  __shared__ __attribute__((aligned(16))) float hs[2 * HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, uq = lane >> 4;
  const int b0 = blockIdx.x * 16;
  const int b = b0 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;

  float uf[TPW][KS];
  bool tv[TPW];
#pragma unroll
  for (int jt = 0; jt < TPW; ++jt) {
    int tile = wave + jt * NW;
    tv[jt] = tile < MT;
    int tl = tv[jt] ? tile : 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) uf[jt][s] = tv[jt] ? Up[(size_t)(4 * s + uq) * N + tl * 16 + j] : 0.f;
  }
  for (int i = tid; i < 2 * HS; i += NW * 64) hs[i] = 0.f;
  float c[TPW];
  f32x4 zc[TPW], zn[TPW];
  auto loadz = [&](f32x4 (&z)[TPW], int t) {
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) {
      int tile = wave + jt * NW;
      if (tv[jt]) z[jt] = *reinterpret_cast<const f32x4*>(Z + ((size_t)bc * T + t) * N + (tile * 4 + uq) * 4);
    }
  };
#pragma unroll
  for (int jt = 0; jt < TPW; ++jt) {
    c[jt] = 0.f;
    zc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    zn[jt] = zc[jt];
  }
  loadz(zc, reverse ? T - 1 : 0);
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    if (step + 1 < T) loadz(zn, reverse ? t - 1 : t + 1);
    f32x4 acc[TPW];
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) acc[jt] = zc[jt];
    const float* hb = hs + cur * HS;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      f32x4 hv = *reinterpret_cast<const f32x4*>(hb + ((q * 4 + uq) * 16 + j) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * q + r < KS) {
#pragma unroll
          for (int jt = 0; jt < TPW; ++jt)
            if (tv[jt]) acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[jt][4 * q + r], hv[r], acc[jt], 0, 0, 0);
        }
      }
    }
    float* hn = hs + (cur ^ 1) * HS;
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) {
      if (tv[jt]) {
        int tile = wave + jt * NW;
        int unit = tile * 4 + uq;
        float4 g4;
        float h = mgr_cell_fwd(acc[jt][0], acc[jt][1], acc[jt][2], acc[jt][3], c[jt], g4);
        // unit k = 4*tile + uq  ->  k-step s = tile, kk = uq  ->  image [q = tile>>2][kk = uq][j][r = tile&3]
        hn[(((tile >> 2) * 4 + uq) * 16 + j) * 4 + (tile & 3)] = h;
        if (bvalid) {
          size_t row = (size_t)b * T + t;
          float yo = h;
          if (R) yo += R[row * ldr + unit];
          Y[row * ldy + unit] = yo;
          if (G) *reinterpret_cast<float4*>(G + (row * H + unit) * 4) = g4;
          if (Cs) Cs[row * H + unit] = c[jt];
        }
      }
    }
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) zc[jt] = zn[jt];
    __syncthreads();
    cur ^= 1;
  }
}

// Backward.  H units -> MTB = ceil(H/16) tiles of 16 units, one tile per wave; K = 4H packed gate columns,
// i.e. H MFMA k-steps (k-step s = unit s, kk = gate).
template <int H>
__global__ __launch_bounds__(NW * 64) void k_scan_bwd_mfma(const float* __restrict__ dY, int lddy,
                                                           const float* __restrict__ G, const float* __restrict__ Cs,
                                                           const float* __restrict__ Up, float* __restrict__ dZ, int B,
                                                           int T, int reverse) {
  constexpr int N = 4 * H;
  constexpr int MTB = (H + 15) / 16;
  constexpr int QN = (H + 3) / 4;
  constexpr int DS = QN * 4 * 16 * 4;
  static_assert(MTB <= NW, "H too large for the single-CU backward kernel");
  __shared__ __attribute__((aligned(16))) float dzs[2 * DS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  __syncthreads();
  int cur = 0;
  for (int n = T - 1; n >= 0; --n) {
    const int t = reverse ? T - 1 - n : n;
    const int tp = reverse ? t + 1 : t - 1;
    const bool has_prev = n > 0;
    float* dn = dzs + (cur ^ 1) * DS;
    if (tvalid) {
      size_t row = (size_t)bc * T + t;
      float4 dz[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int u = u0 + e;
        dz[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < H) {
          float dh = dY[row * lddy + u] + dhr[e];
          float4 g4 = *reinterpret_cast<const float4*>(G + (row * H + u) * 4);
          float cc = Cs[row * H + u];
          float cp = has_prev ? Cs[((size_t)bc * T + tp) * H + u] : 0.f;
          dz[e] = mgr_cell_bwd(dh, g4, cc, cp, dcc[e]);
          if (bvalid) *reinterpret_cast<float4*>(dZ + row * N + u * 4) = dz[e];
        }
      }
      // image [q = unit>>2][kk = gate][j][r = unit&3]; this lane's units u0..u0+3 share q = u0>>2
      const int q = u0 >> 2;
      if (q < QN) {
        *reinterpret_cast<f32x4*>(dn + ((q * 4 + 0) * 16 + j) * 4) = (f32x4){dz[0].x, dz[1].x, dz[2].x, dz[3].x};
        *reinterpret_cast<f32x4*>(dn + ((q * 4 + 1) * 16 + j) * 4) = (f32x4){dz[0].y, dz[1].y, dz[2].y, dz[3].y};
        *reinterpret_cast<f32x4*>(dn + ((q * 4 + 2) * 16 + j) * 4) = (f32x4){dz[0].z, dz[1].z, dz[2].z, dz[3].z};
        *reinterpret_cast<f32x4*>(dn + ((q * 4 + 3) * 16 + j) * 4) = (f32x4){dz[0].w, dz[1].w, dz[2].w, dz[3].w};
      }
    }
    __syncthreads();
    cur ^= 1;
    if (tvalid && has_prev) {
      const float* db = dzs + cur * DS;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        f32x4 dv = *reinterpret_cast<const f32x4*>(db + ((q * 4 + uq) * 16 + j) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * q + r < H) {
            if (r & 1)
              a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * q + r], dv[r], a1, 0, 0, 0);
            else
              a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * q + r], dv[r], a0, 0, 0, 0);
          }
        }
      }
      dhr = a0 + a1;
    }
  }
}

}  // namespace

// returns 1 if launched, 0 if the shape has no instantiation, <0 on error
int mgr_scan_fwd_mfma(mgr_ctx* c, const float* Z, const float* Up, float* Y, int ldy, const float* R, int ldr, float* G,
                      float* Cs, int B, int T, int H, int reverse) {
  int grid = (B + 15) / 16;
  hipStream_t s = mgr_stream(c);
#define FWD_CASE(KS, TPW)                                                                                                  \
  case 4 * KS:                                                                                                             \
    hipLaunchKernelGGL((k_scan_fwd_mfma<KS, TPW>), dim3(grid), dim3(NW * 64), 0, s, Z, Up, Y, ldy, R, ldr, G, Cs, B, T, reverse); \
    break;
  switch (H) {
    FWD_CASE(1, 1)
    FWD_CASE(2, 1)
    FWD_CASE(4, 1)
    FWD_CASE(8, 1)
    FWD_CASE(16, 2)
    FWD_CASE(25, 4)
    FWD_CASE(32, 4)
    default:
      return 0;
  }
#undef FWD_CASE
  MGR_LAUNCH_CHECK();
  return 1;
}

int mgr_scan_bwd_mfma(mgr_ctx* c, const float* dY, int lddy, const float* G, const float* Cs, const float* Up, float* dZ, int B,
                      int T, int H, int reverse) {
  int grid = (B + 15) / 16;
  hipStream_t s = mgr_stream(c);
#define BWD_CASE(HH)                                                                                              \
  case HH:                                                                                                        \
    hipLaunchKernelGGL((k_scan_bwd_mfma<HH>), dim3(grid), dim3(NW * 64), 0, s, dY, lddy, G, Cs, Up, dZ, B, T, reverse); \
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
