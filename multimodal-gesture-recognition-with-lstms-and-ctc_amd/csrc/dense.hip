// K5: Dropout -> Dense -> softmax forward, and its backward (reference multimodal_fusion/multimodal.py:171-179).
// HBM-bound: the (B,T,D) activations are read once (coalesced, through LDS), D*C weights sit in LDS.
#include "common.h"

namespace {

constexpr int FR = 32;   // frames per tile
constexpr int DC = 128;  // feature chunk staged per pass

__device__ __forceinline__ float drop_factor(const float* dmask, float p, float inv_keep, uint64_t seed, size_t idx) {
  if (dmask) return dmask[idx];
  if (p > 0.f) return mgr_drop_scale(seed, idx, p, inv_keep);
  return 1.f;
}

// thread (c = tid % CP, fg = tid / CP); each thread accumulates FR/(256/CP) frames for one class
template <int CP>
__global__ __launch_bounds__(256) void k_dense_softmax_fwd(const float* __restrict__ A, int lda,
                                                           const float* __restrict__ dmask, float p, float inv_keep,
                                                           uint64_t seed, const float* __restrict__ Wd,
                                                           const float* __restrict__ bd, float* __restrict__ P,
                                                           size_t nframes, int D, int C) {
  __shared__ float As[FR][DC + 1];
  __shared__ float Ws[DC][CP];
  __shared__ float Ls[FR][CP + 1];
  constexpr int FG = 256 / CP;   // frame groups processed concurrently
  constexpr int FPT = FR / FG;   // frames per thread
  const int tid = threadIdx.x;
  const int c = tid % CP, fg = tid / CP;
  for (size_t f0 = (size_t)blockIdx.x * FR; f0 < nframes; f0 += (size_t)gridDim.x * FR) {
    float acc[FPT];
#pragma unroll
    for (int i = 0; i < FPT; ++i) acc[i] = (c < C) ? bd[c] : 0.f;
    for (int d0 = 0; d0 < D; d0 += DC) {
      __syncthreads();
      // stage A chunk (coalesced along d) with dropout applied
      for (int i = tid; i < FR * DC; i += 256) {
        int fr = i / DC, d = i % DC;
        size_t f = f0 + fr;
        float v = 0.f;
        if (f < nframes && d0 + d < D) {
          v = A[f * (size_t)lda + d0 + d];
          v *= drop_factor(dmask, p, inv_keep, seed, f * (size_t)D + d0 + d);
        }
        As[fr][d] = v;
      }
      for (int i = tid; i < DC * CP; i += 256) {
        int d = i / CP, cc = i % CP;
        Ws[d][cc] = (d0 + d < D && cc < C) ? Wd[(size_t)(d0 + d) * C + cc] : 0.f;
      }
      __syncthreads();
      int dn = D - d0 < DC ? D - d0 : DC;
      for (int d = 0; d < dn; ++d) {
        float w = Ws[d][c];
#pragma unroll
        for (int i = 0; i < FPT; ++i) acc[i] += As[fg * FPT + i][d] * w;
      }
    }
#pragma unroll
    for (int i = 0; i < FPT; ++i) Ls[fg * FPT + i][c] = acc[i];
    __syncthreads();
    if (tid < FR) {
      float mx = Ls[tid][0];
      for (int cc = 1; cc < C; ++cc) mx = fmaxf(mx, Ls[tid][cc]);
      float s = 0.f;
      for (int cc = 0; cc < C; ++cc) {
        float e = expf(Ls[tid][cc] - mx);
        Ls[tid][cc] = e;
        s += e;
      }
      float inv = 1.f / s;
      for (int cc = 0; cc < C; ++cc) Ls[tid][cc] *= inv;
    }
    __syncthreads();
    // the FR x C block is contiguous in P
    for (int i = tid; i < FR * C; i += 256) {
      int fr = i / C, cc = i % C;
      if (f0 + fr < nframes) P[(f0 + fr) * (size_t)C + cc] = Ls[fr][cc];
    }
  }
}

// Backward.  Thread d owns one input feature: Wd[d,:] in registers, dWd[d,:] accumulated in registers over the
// workgroup's frame range, dA written per frame.  Partial dWd/dbd slabs are reduced by k_dense_reduce.
template <int CM>
__global__ __launch_bounds__(256) void k_dense_bwd(const float* __restrict__ A, int lda, const float* __restrict__ dmask,
                                                   float p, float inv_keep, uint64_t seed,
                                                   const float* __restrict__ dL, const float* __restrict__ Wd,
                                                   float* __restrict__ slabW, float* __restrict__ slabB,
                                                   float* __restrict__ dA, int ldda, size_t nframes,
                                                   int frames_per_wg, int D, int C) {
  __shared__ float dLs[FR][CM];
  const int tid = threadIdx.x;
  size_t fbeg = (size_t)blockIdx.x * frames_per_wg;
  size_t fend = fbeg + frames_per_wg < nframes ? fbeg + frames_per_wg : nframes;
  float* mySlabW = slabW + (size_t)blockIdx.x * D * C;
  float* mySlabB = slabB + (size_t)blockIdx.x * C;
  float accb = 0.f;  // thread c < C accumulates dbd[c]
  for (int d0 = 0; d0 < D; d0 += 256) {
    int d = d0 + tid;
    bool dv = d < D;
    float w[CM], acc[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
      w[c] = (dv && c < C) ? Wd[(size_t)d * C + c] : 0.f;
      acc[c] = 0.f;
    }
    for (size_t f0 = fbeg; f0 < fend; f0 += FR) {
      __syncthreads();
      for (int i = tid; i < FR * CM; i += 256) {
        int fr = i / CM, c = i % CM;
        dLs[fr][c] = (f0 + fr < fend && c < C) ? dL[(f0 + fr) * (size_t)C + c] : 0.f;
      }
      __syncthreads();
      if (d0 == 0 && tid < C) {
        for (int fr = 0; fr < FR; ++fr) accb += dLs[fr][tid];
      }
      int fn = (int)(fend - f0 < FR ? fend - f0 : FR);
      if (dv) {
        for (int fr = 0; fr < fn; ++fr) {
          size_t f = f0 + fr;
          float dm = drop_factor(dmask, p, inv_keep, seed, f * (size_t)D + d);
          float a = A[f * (size_t)lda + d] * dm;
          float da = 0.f;
#pragma unroll
          for (int c = 0; c < CM; ++c) {
            float g = dLs[fr][c];
            acc[c] += a * g;
            da += g * w[c];
          }
          if (dA) dA[f * (size_t)ldda + d] = da * dm;
        }
      }
    }
    if (dv) {
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) mySlabW[(size_t)d * C + c] = acc[c];
    }
  }
  if (tid < C) mySlabB[tid] = accb;
}

__global__ void k_slab_reduce(const float* __restrict__ slab, float* __restrict__ out, size_t n, int nslab) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += slab[(size_t)k * n + i];
    out[i] = s;
  }
}

static int dense_bwd_wgs(size_t nframes) {
  size_t w = (nframes + 511) / 512;
  if (w > 512) w = 512;
  if (w < 1) w = 1;
  return (int)w;
}

}  // namespace

extern "C" {

int mgr_dense_softmax_fwd(mgr_ctx* c, const float* A, int lda, const float* dmask, float p, uint64_t seed,
                          const float* Wd, const float* bd, float* P, int B, int T, int D, int C) {
  MGR_REQUIRE(c && A && Wd && bd && P, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && D > 0 && C > 0 && lda >= D, "bad shape");
  MGR_REQUIRE(C <= 64, "C=%d > 64 unsupported", C);
  MGR_REQUIRE(p >= 0.f && p < 1.f, "dropout rate out of range");
  size_t nframes = (size_t)B * T;
  size_t g = (nframes + FR - 1) / FR;
  if (g > 4096) g = 4096;
  float inv_keep = 1.f / (1.f - p);
  mgr_prof_begin(c, MGR_K_DENSE_FWD);
  if (C <= 32)
    hipLaunchKernelGGL(k_dense_softmax_fwd<32>, dim3((int)g), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, Wd, bd, P, nframes, D, C);
  else
    hipLaunchKernelGGL(k_dense_softmax_fwd<64>, dim3((int)g), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, Wd, bd, P, nframes, D, C);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_DENSE_FWD);
  return 0;
}

size_t mgr_dense_bwd_ws_bytes(int B, int T, int D, int C) {
  size_t nframes = (size_t)B * T;
  int nwg = dense_bwd_wgs(nframes);
  return mgr_align_up((size_t)nwg * D * C * sizeof(float), 256) + mgr_align_up((size_t)nwg * C * sizeof(float), 256);
}

int mgr_dense_bwd(mgr_ctx* c, const float* A, int lda, const float* dmask, float p, uint64_t seed,
                  const float* dLogits, const float* Wd, float* dWd, float* dbd, float* dA, int ldda, int B, int T,
                  int D, int C, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && A && dLogits && Wd && dWd && dbd, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && D > 0 && C > 0 && lda >= D, "bad shape");
  MGR_REQUIRE(C <= 48, "C=%d > 48 unsupported", C);
  MGR_REQUIRE(ws && ws_bytes >= mgr_dense_bwd_ws_bytes(B, T, D, C), "workspace too small");
  size_t nframes = (size_t)B * T;
  int nwg = dense_bwd_wgs(nframes);
  int fpw = (int)((nframes + nwg - 1) / nwg);
  fpw = (fpw + FR - 1) / FR * FR;
  nwg = (int)((nframes + fpw - 1) / fpw);
  float* slabW = reinterpret_cast<float*>(ws);
  float* slabB = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + mgr_align_up((size_t)dense_bwd_wgs(nframes) * D * C * sizeof(float), 256));
  float inv_keep = 1.f / (1.f - p);
  mgr_prof_begin(c, MGR_K_DENSE_BWD);
  if (C <= 24)
    hipLaunchKernelGGL(k_dense_bwd<24>, dim3(nwg), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, dLogits, Wd, slabW, slabB, dA, ldda, nframes, fpw, D, C);
  else
    hipLaunchKernelGGL(k_dense_bwd<48>, dim3(nwg), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, dLogits, Wd, slabW, slabB, dA, ldda, nframes, fpw, D, C);
  MGR_LAUNCH_CHECK();
  size_t nW = (size_t)D * C;
  hipLaunchKernelGGL(k_slab_reduce, dim3((int)((nW + 255) / 256)), dim3(256), 0, mgr_stream(c), slabW, dWd, nW, nwg);
  hipLaunchKernelGGL(k_slab_reduce, dim3(1), dim3(256), 0, mgr_stream(c), slabB, dbd, (size_t)C, nwg);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_DENSE_BWD);
  return 0;
}

}  // extern "C"
