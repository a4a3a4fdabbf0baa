// Generic (any H <= 1024) LSTM scan kernels: batch-split, recurrent matrix streamed from L2 every step.
// Correctness-first fallback for shapes the weight-stationary MFMA kernels (lstm_mfma.hip) do not cover.
#include "lstm_common.h"

namespace {

template <int NB, int UPT>
__global__ __launch_bounds__(256) void k_scan_fwd_simple(const float* __restrict__ Z, const float* __restrict__ Up,
                                                         float* __restrict__ Y, int ldy, const float* __restrict__ R,
                                                         int ldr, float* __restrict__ G, float* __restrict__ Cs, int B,
                                                         int T, int H, int reverse) {
  extern __shared__ __attribute__((aligned(16))) float hs[];  // [2][NB][H]
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * NB;
  const int nb = (B - b0) < NB ? (B - b0) : NB;
  const int N = 4 * H;
  float c[UPT][NB];
#pragma unroll
  for (int q = 0; q < UPT; ++q)
#pragma unroll
    for (int s = 0; s < NB; ++s) c[q][s] = 0.f;
  for (int i = tid; i < 2 * NB * H; i += 256) hs[i] = 0.f;
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    float4 acc[UPT][NB];
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      int u = tid + q * 256;
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        acc[q][s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < H && s < nb) acc[q][s] = *reinterpret_cast<const float4*>(Z + ((size_t)(b0 + s) * T + t) * N + u * 4);
      }
    }
    const float* hc = hs + cur * NB * H;
    for (int k = 0; k < H; ++k) {
      float hk[NB];
#pragma unroll
      for (int s = 0; s < NB; ++s) hk[s] = hc[s * H + k];
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        int u = tid + q * 256;
        if (u < H) {
          float4 uv = *reinterpret_cast<const float4*>(Up + (size_t)k * N + u * 4);
#pragma unroll
          for (int s = 0; s < NB; ++s) {
            acc[q][s].x += hk[s] * uv.x;
            acc[q][s].y += hk[s] * uv.y;
            acc[q][s].z += hk[s] * uv.z;
            acc[q][s].w += hk[s] * uv.w;
          }
        }
      }
    }
    float* hn = hs + (cur ^ 1) * NB * H;
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      int u = tid + q * 256;
      if (u < H) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          if (s < nb) {
            float4 g4;
            float h = mgr_cell_fwd(acc[q][s].x, acc[q][s].y, acc[q][s].z, acc[q][s].w, c[q][s], g4);
            hn[s * H + u] = h;
            size_t row = (size_t)(b0 + s) * T + t;
            float yo = h;
            if (R) yo += R[row * ldr + u];
            Y[row * ldy + u] = yo;
            if (G) *reinterpret_cast<float4*>(G + (row * H + u) * 4) = g4;
            if (Cs) Cs[row * H + u] = c[q][s];
          }
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

template <int NB, int UPT>
__global__ __launch_bounds__(256) void k_scan_bwd_simple(const float* __restrict__ dY, int lddy,
                                                         const float* __restrict__ G, const float* __restrict__ Cs,
                                                         const float* __restrict__ UpT, float* __restrict__ dZ, int B,
                                                         int T, int H, int reverse) {
  extern __shared__ __attribute__((aligned(16))) float dzs[];  // [NB][4H]
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * NB;
  const int nb = (B - b0) < NB ? (B - b0) : NB;
  const int N = 4 * H;
  float dhr[UPT][NB], dcc[UPT][NB];
#pragma unroll
  for (int q = 0; q < UPT; ++q)
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      dhr[q][s] = 0.f;
      dcc[q][s] = 0.f;
    }
  // walk the forward recursion backwards: forward order visits t = step (or T-1-step when reverse)
  for (int n = T - 1; n >= 0; --n) {
    const int t = reverse ? T - 1 - n : n;
    const int tp = reverse ? t + 1 : t - 1;  // previous step of the forward recursion
    const bool has_prev = n > 0;
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      int u = tid + q * 256;
      if (u < H) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          if (s < nb) {
            size_t row = (size_t)(b0 + s) * T + t;
            float dh = dY[row * lddy + u] + dhr[q][s];
            float4 g4 = *reinterpret_cast<const float4*>(G + (row * H + u) * 4);
            float cc = Cs[row * H + u];
            float cp = has_prev ? Cs[((size_t)(b0 + s) * T + tp) * H + u] : 0.f;
            float4 dz = mgr_cell_bwd(dh, g4, cc, cp, dcc[q][s]);
            *reinterpret_cast<float4*>(dZ + row * N + u * 4) = dz;
            *reinterpret_cast<float4*>(dzs + s * N + u * 4) = dz;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < UPT; ++q)
#pragma unroll
      for (int s = 0; s < NB; ++s) dhr[q][s] = 0.f;
    if (has_prev) {
      for (int j = 0; j < N; ++j) {
        float dv[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) dv[s] = dzs[s * N + j];
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
          int u = tid + q * 256;
          if (u < H) {
            float w = UpT[(size_t)j * H + u];
#pragma unroll
            for (int s = 0; s < NB; ++s) dhr[q][s] += dv[s] * w;
          }
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

// host-side launchers used by lstm.hip
int mgr_scan_fwd_simple(mgr_ctx* c, const float* Z, const float* Up, float* Y, int ldy, const float* R, int ldr, float* G,
                        float* Cs, int B, int T, int H, int reverse) {
  constexpr int NB = 4;
  int grid = (B + NB - 1) / NB;
  size_t lds = (size_t)2 * NB * H * sizeof(float);
  hipStream_t s = mgr_stream(c);
  if (H <= 256)
    hipLaunchKernelGGL((k_scan_fwd_simple<NB, 1>), dim3(grid), dim3(256), lds, s, Z, Up, Y, ldy, R, ldr, G, Cs, B, T, H, reverse);
  else if (H <= 512)
    hipLaunchKernelGGL((k_scan_fwd_simple<NB, 2>), dim3(grid), dim3(256), lds, s, Z, Up, Y, ldy, R, ldr, G, Cs, B, T, H, reverse);
  else if (H <= 1024)
    hipLaunchKernelGGL((k_scan_fwd_simple<NB, 4>), dim3(grid), dim3(256), lds, s, Z, Up, Y, ldy, R, ldr, G, Cs, B, T, H, reverse);
  else
    return mgr_fail(-1, "H=%d > 1024 unsupported", H);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_scan_bwd_simple(mgr_ctx* c, const float* dY, int lddy, const float* G, const float* Cs, const float* UpT, float* dZ,
                        int B, int T, int H, int reverse) {
  constexpr int NB = 4;
  int grid = (B + NB - 1) / NB;
  size_t lds = (size_t)NB * 4 * H * sizeof(float);
  hipStream_t s = mgr_stream(c);
  if (H <= 256)
    hipLaunchKernelGGL((k_scan_bwd_simple<NB, 1>), dim3(grid), dim3(256), lds, s, dY, lddy, G, Cs, UpT, dZ, B, T, H, reverse);
  else if (H <= 512)
    hipLaunchKernelGGL((k_scan_bwd_simple<NB, 2>), dim3(grid), dim3(256), lds, s, dY, lddy, G, Cs, UpT, dZ, B, T, H, reverse);
  else if (H <= 1024)
    hipLaunchKernelGGL((k_scan_bwd_simple<NB, 4>), dim3(grid), dim3(256), lds, s, dY, lddy, G, Cs, UpT, dZ, B, T, H, reverse);
  else
    return mgr_fail(-1, "H=%d > 1024 unsupported", H);
  MGR_LAUNCH_CHECK();
  return 0;
}
