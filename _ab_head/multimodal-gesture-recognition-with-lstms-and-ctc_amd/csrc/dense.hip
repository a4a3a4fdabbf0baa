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

// ---- the same two layers on the f32 matrix cores (D <= 256, C <= 32: the fusion head D = 200 and the H = 128 heads) --------------
// Why: inside the training step these kernels run beside the encoder scans of the other stream, whose MFMA streams keep every
// SIMD busy; a vector-ALU instruction does not overlap with a SIMD's f32 MFMAs (profiles/r04_single_cu_probes.txt), so the
// ~12 vector instructions per multiply-add of the LDS-tiled forms above (1.33 / 0.88 ms in the step for 0.16 / 0.29 ms alone) are
// what made the head 4.1 ms of the step's critical chain.  v_mfma_f32_16x16x4_f32 is an exact f32 FMA chain (same arithmetic).
// K is walked in blocks of 16 with the k <-> (block q, lane group kk, register r) mapping  k = 16 q + 4 kk + r : a lane's A
// operands of four consecutive MFMAs are ONE float4 of its frame's row, loaded straight from global memory (no LDS at all).
typedef float f32x4_ __attribute__((ext_vector_type(4)));

// Register budget: <= 80 VGPRs (__launch_bounds__(256, 6)), so that a wave fits on a SIMD that already holds two 216-register
// scan waves - with the 256 registers of the first version (Wd fragments in registers) the workgroups only found room on the 104
// CUs with a single scan workgroup and the forward took 2.6 ms in the step instead of 1.3.  The Wd fragments therefore live in LDS.
template <int DB>   // DB = ceil(D / 16)
__global__ __launch_bounds__(256, 6) void k_dense_softmax_fwd_mfma(const float* __restrict__ A, int lda, const float* __restrict__ dmask,
                                                                   float p, float inv_keep, uint64_t seed,
                                                                   const float* __restrict__ Wd, const float* __restrict__ bd,
                                                                   float* __restrict__ P, size_t nframes, int D, int C) {
  // B fragments, fragment order: [q][half][lane] = { (r, nt) = (2 half, 0), (2 half, 1), (2 half + 1, 0), (2 half + 1, 1) }
  // of k-step (q, r), class tile nt: B[k = kk][n] = Wd[16 q + 4 kk + r][16 nt + n]
  __shared__ f32x4_ wfs[DB][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kk = lane >> 4;
  for (int idx = threadIdx.x; idx < DB * 2 * 64; idx += 256) {
    const int l = idx & 63, half = (idx >> 6) & 1, q = idx >> 7;
    f32x4_ w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 2 * half + (e >> 1), nt = e & 1;
      const int k = 16 * q + 4 * (l >> 4) + r, c = 16 * nt + (l & 15);
      w[e] = (k < D && c < C) ? Wd[(size_t)k * C + c] : 0.f;
    }
    wfs[q][half][l] = w;
  }
  __syncthreads();
  const float b0 = n < C ? bd[n] : 0.f, b1 = 16 + n < C ? bd[16 + n] : 0.f;
  const bool drop = dmask != nullptr || p > 0.f;
  const size_t ntile = (nframes + 15) / 16;
  for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile < ntile; tile += (size_t)gridDim.x * 4) {
    const size_t f0 = tile * 16;
    // A operand: frame m = lane & 15 (clamped: rows beyond the end are computed and dropped)
    const size_t fa = f0 + n < nframes ? f0 + n : nframes - 1;
    const float* arow = A + fa * (size_t)lda + 4 * kk;
    f32x4_ a0 = {b0, b0, b0, b0}, a1 = {b1, b1, b1, b1};
    // K loop (rolled: the kernel must stay at <= 80 registers), the activations of the next block in flight under the MFMAs
    f32x4_ nxt = (4 * kk < D) ? *reinterpret_cast<const f32x4_*>(arow) : (f32x4_){0.f, 0.f, 0.f, 0.f};   // (D % 4 == 0: whole or nothing)
#pragma nounroll
    for (int q = 0; q < DB; ++q) {
      f32x4_ av = nxt;
      if (q + 1 < DB) nxt = (16 * (q + 1) + 4 * kk < D) ? *reinterpret_cast<const f32x4_*>(arow + 16 * (q + 1)) : (f32x4_){0.f, 0.f, 0.f, 0.f};
      if (drop) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * q + 4 * kk + r;
          if (k < D) av[r] *= drop_factor(dmask, p, inv_keep, seed, fa * (size_t)D + k);
        }
      }
      const f32x4_ w01 = wfs[q][0][lane], w23 = wfs[q][1][lane];
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], w01[0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], w01[1], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], w01[2], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], w01[3], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], w23[0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], w23[1], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], w23[2], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], w23[3], a1, 0, 0, 0);
    }
    // D layout: lane holds frames 4 kk + i (i = 0..3) of class n (a0) and 16 + n (a1); softmax over the 16 lanes of a group
    const bool v0 = n < C, v1 = 16 + n < C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = v0 ? a0[i] : -__builtin_huge_valf(), x1 = v1 ? a1[i] : -__builtin_huge_valf();
      float mx = fmaxf(x0, x1);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float e0 = v0 ? expf(x0 - mx) : 0.f, e1 = v1 ? expf(x1 - mx) : 0.f;
      float sm = e0 + e1;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
      const float inv = 1.f / sm;
      const size_t f = f0 + 4 * kk + i;
      if (f < nframes) {
        if (v0) P[f * (size_t)C + n] = e0 * inv;
        if (v1) P[f * (size_t)C + 16 + n] = e1 * inv;
      }
    }
  }
}

// Backward on the matrix cores.  A workgroup of 4 waves takes a 16-feature slice of the feature range (blockIdx.y; NTH = 1 tile of
// 16: what keeps the kernel at <= 80 registers) and walks 16-frame tiles; a lane owns the elements (frame 4 kk + i, feature 16 t + n),
// i < 4, t < NTH, in BOTH products:
//   dA[f, d]  = dm * sum_c dL[f, c] Wd[d, c]      M = frames, N = features, K = classes (6 k-steps of 4; Wd^T fragments stationary)
//   dWd[d, c] = sum_f (A dm)[f, d] dL[f, c]       M = features, N = classes, K = frames with k-step i <-> frames 4 kk + i
// so A is loaded and its dropout factor evaluated ONCE per element.  Per-workgroup partial dWd / dbd slabs, reduced as before.
// KS k-steps of 4 classes in the dA product, NC class tiles of 16 in the dWd product: (6, 2) for C <= 24, (12, 3) for C <= 48 (the
// audio head, C = 44; 4 workgroups per CU instead of 6)
template <int NTH, int KS = 6, int NC = 2>
__global__ __launch_bounds__(256, KS > 6 ? 4 : 6) void k_dense_bwd_mfma(const float* __restrict__ A, int lda, const float* __restrict__ dmask, float p,
                                                           float inv_keep, uint64_t seed, const float* __restrict__ dL,
                                                           const float* __restrict__ Wd, float* __restrict__ slabW,
                                                           float* __restrict__ slabB, float* __restrict__ dA, int ldda, size_t nframes,
                                                           int D, int C) {
  __shared__ f32x4_ red[4][NTH * NC][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kk = lane >> 4;
  const int d0 = blockIdx.y * NTH * 16;     // first feature of this workgroup's slice
  // Wd^T fragments: k-step s (classes 4 s + kk), feature tile t: B[k = kk][n] = Wd[d0 + 16 t + n][4 s + kk]
  float wt[NTH][KS];
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int s6 = 0; s6 < KS; ++s6) {
      const int d = d0 + 16 * t + n, c = 4 * s6 + kk;
      wt[t][s6] = (d < D && c < C) ? Wd[(size_t)d * C + c] : 0.f;
    }
  f32x4_ accW[NTH][NC];
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) accW[t][nt] = (f32x4_){0.f, 0.f, 0.f, 0.f};
  float accb[NC];
#pragma unroll
  for (int nt = 0; nt < NC; ++nt) accb[nt] = 0.f;
  // (32-bit element offsets: the launcher sends tensors of 2^31 elements or more to the vector-ALU kernels)
  const unsigned nf = (unsigned)nframes, ntile = (nf + 15u) / 16u;
  for (unsigned tile = blockIdx.x * 4u + wave; tile < ntile; tile += gridDim.x * 4u) {
    const unsigned f0 = tile * 16u;
    // dL as A operand of the dA product: lane (m = frame n, k = class 4 s + kk)
    float gl[KS];
    {
      const unsigned f = f0 + n;
#pragma unroll
      for (int s6 = 0; s6 < KS; ++s6) {
        const int c = 4 * s6 + kk;
        gl[s6] = (f < nf && c < C) ? dL[f * (unsigned)C + c] : 0.f;
      }
    }
    // dL as B operand of the dWd product: k-step i <-> frame 4 kk + i, lane (k = kk, n = class)
    float gb[4][NC];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned f = f0 + 4 * kk + i;
#pragma unroll
      for (int nt = 0; nt < NC; ++nt) {
        gb[i][nt] = (f < nf && 16 * nt + n < C) ? dL[f * (unsigned)C + 16 * nt + n] : 0.f;
        accb[nt] += gb[i][nt];
      }
    }
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
      const int d = d0 + 16 * t + n;
      // this lane's four elements (frames 4 kk + i, feature d): activation x dropout factor, and the factor
      float am[4], dm[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned f = f0 + 4 * kk + i;
        const bool ok = f < nf && d < D;
        dm[i] = ok ? drop_factor(dmask, p, inv_keep, seed, (size_t)(f * (unsigned)D + d)) : 0.f;
        am[i] = ok ? A[f * (unsigned)lda + d] * dm[i] : 0.f;
        __builtin_amdgcn_sched_barrier(0);   // (one 64-bit hash at a time: four interleaved ones spill at 80 registers)
      }
      if (dA) {
        f32x4_ o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s6 = 0; s6 < KS; ++s6) o = __builtin_amdgcn_mfma_f32_16x16x4f32(gl[s6], wt[t][s6], o, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned f = f0 + 4 * kk + i;
          if (f < nf && d < D) dA[f * (unsigned)ldda + d] = o[i] * dm[i];
        }
      }
      // dWd: A operand lane (m = feature n of tile t, k = kk) for k-step i = am[i]
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nt = 0; nt < NC; ++nt) accW[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[i], gb[i][nt], accW[t][nt], 0, 0, 0);
    }
  }
  // the four waves' partial dWd tiles meet in LDS (fixed order), one slab per workgroup column: a lane holds features
  // 16 t + 4 kk + i of class 16 nt + n
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) red[wave][t * NC + nt][lane] = accW[t][nt];
  __syncthreads();
  float* mySlabW = slabW + (size_t)blockIdx.x * D * C;
  for (int idx = threadIdx.x; idx < NTH * NC * 64; idx += 256) {
    const int tn = idx >> 6, l = idx & 63;
    const f32x4_ sum = red[0][tn][l] + red[1][tn][l] + red[2][tn][l] + red[3][tn][l];
    const int t = tn / NC, nt = tn % NC, c = 16 * nt + (l & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = d0 + 16 * t + 4 * (l >> 4) + i;
      if (d < D && c < C) mySlabW[(size_t)d * C + c] = sum[i];
    }
  }
  if (blockIdx.y == 0) {   // dbd: sum over the four frame groups kk of a wave, then the waves
    __syncthreads();
    float* rb = reinterpret_cast<float*>(&red[0][0][0]);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
      accb[nt] += __shfl_xor(accb[nt], 16);
      accb[nt] += __shfl_xor(accb[nt], 32);
      if (kk == 0) rb[wave * (16 * NC) + 16 * nt + n] = accb[nt];
    }
    __syncthreads();
    if ((int)threadIdx.x < C)
      slabB[(size_t)blockIdx.x * C + threadIdx.x] = rb[threadIdx.x] + rb[16 * NC + threadIdx.x] + rb[2 * 16 * NC + threadIdx.x] + rb[3 * 16 * NC + threadIdx.x];
  }
}

// the matrix-core forms: D <= maxD (forward: 256, its Wd fragments live in LDS; backward: a workgroup takes a 16-feature slice,
// any width - 1024 covers the unimodal heads, D = 600 / 1000), C <= 24, rows readable as float4, 32-bit element offsets
static bool dense_mfma_ok(const mgr_ctx* c, const float* A, int lda, int ldo, int D, int C, size_t nframes, int maxD = 256, int maxC = 24) {
  const size_t ld = (size_t)(lda > ldo ? lda : ldo);
  return c->tune[13] == 0 && D <= maxD && D % 4 == 0 && C <= maxC && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
         nframes * (ld > (size_t)D ? ld : (size_t)D) < ((size_t)1 << 31);
}

// both reductions of the backward pass in one launch (each launch of the step queues behind the resident scans)
__global__ void k_slab_reduce2(const float* __restrict__ slabW, float* __restrict__ outW, size_t nW, const float* __restrict__ slabB,
                               float* __restrict__ outB, size_t nB, int nslab) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nW + nB; i += (size_t)gridDim.x * blockDim.x) {
    const bool w = i < nW;
    const float* src = w ? slabW + i : slabB + (i - nW);
    const size_t n = w ? nW : nB;
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += src[(size_t)k * n];
    if (w)
      outW[i] = s;
    else
      outB[i - nW] = s;
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
  if (dense_mfma_ok(c, A, lda, 0, D, C, nframes)) {
    size_t gm = ((nframes + 15) / 16 + 3) / 4;
    if (gm > 2048) gm = 2048;
    if (D <= 208)
      hipLaunchKernelGGL(k_dense_softmax_fwd_mfma<13>, dim3((int)gm), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, Wd, bd, P, nframes, D, C);
    else
      hipLaunchKernelGGL(k_dense_softmax_fwd_mfma<16>, dim3((int)gm), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, Wd, bd, P, nframes, D, C);
  } else if (C <= 32)
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
  if (dense_mfma_ok(c, A, lda, ldda, D, C, nframes, 1024, 48)) {
    // one slab per workgroup column; 4 waves x 16-frame tiles, two halves of the feature range side by side (blockIdx.y)
    int gx = (int)(((nframes + 15) / 16 + 3) / 4);
    const int cap = dense_bwd_wgs(nframes);      // (what the workspace was sized for)
    if (gx > cap) gx = cap;
    if (gx > 256) gx = 256;
    nwg = gx;
    if (C <= 24)
      hipLaunchKernelGGL((k_dense_bwd_mfma<1, 6, 2>), dim3(gx, (D + 15) / 16), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, dLogits, Wd, slabW, slabB, dA, ldda, nframes, D, C);
    else
      hipLaunchKernelGGL((k_dense_bwd_mfma<1, 12, 3>), dim3(gx, (D + 15) / 16), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, dLogits, Wd, slabW, slabB, dA, ldda, nframes, D, C);
  } else if (C <= 24)
    hipLaunchKernelGGL(k_dense_bwd<24>, dim3(nwg), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, dLogits, Wd, slabW, slabB, dA, ldda, nframes, fpw, D, C);
  else
    hipLaunchKernelGGL(k_dense_bwd<48>, dim3(nwg), dim3(256), 0, mgr_stream(c), A, lda, dmask, p, inv_keep, seed, dLogits, Wd, slabW, slabB, dA, ldda, nframes, fpw, D, C);
  MGR_LAUNCH_CHECK();
  size_t nW = (size_t)D * C;
  hipLaunchKernelGGL(k_slab_reduce2, dim3((int)((nW + C + 255) / 256)), dim3(256), 0, mgr_stream(c), slabW, dWd, nW, slabB, dbd, (size_t)C, nwg);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_DENSE_BWD);
  return 0;
}

size_t mgr_head_ws_bytes(int B, int T, int D, int C, int Lmax) {
  return mgr_align_up(mgr_ctc_ws_bytes(B, T, C, Lmax), 256) + mgr_dense_bwd_ws_bytes(B, T, D, C);
}

// The whole head of a training step in one call (reference multimodal_fusion/multimodal.py:171-179 + losses.py:4-15 and their
// backward pass): Dropout -> Dense -> softmax (P is written: the predict path and the tests read it), CTC loss + dLogits, Dense
// backward (dA, dWd, dbd) and, if asked for, the mean loss.  The kernels are mgr_dense_softmax_fwd / mgr_ctc_loss_grad /
// mgr_dense_bwd's own - the results are bit for bit those of the three calls.
int mgr_head_fwd_bwd(mgr_ctx* c, const float* A, int lda, const float* dmask, float p, uint64_t seed, const float* Wd, const float* bd,
                     const int32_t* labels, const int32_t* input_len, const int32_t* label_len, int B, int T, int D, int C, int Lmax,
                     int skip, int blank, float eps, float gscale, float* P, float* loss, float* loss_mean, float* dLogits, float* dWd,
                     float* dbd, float* dA, int ldda, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && dLogits && ws && ws_bytes >= mgr_head_ws_bytes(B, T, D, C, Lmax), "head: null argument or workspace too small");
  int r = mgr_dense_softmax_fwd(c, A, lda, dmask, p, seed, Wd, bd, P, B, T, D, C);
  if (r) return r;
  const size_t wc = mgr_align_up(mgr_ctc_ws_bytes(B, T, C, Lmax), 256);
  r = mgr_ctc_loss_grad(c, P, labels, input_len, label_len, B, T, C, Lmax, skip, blank, eps, gscale, loss, dLogits, ws, wc);
  if (r) return r;
  if (loss_mean) {
    r = mgr_mean(c, loss, B, loss_mean);
    if (r) return r;
  }
  return mgr_dense_bwd(c, A, lda, dmask, p, seed, dLogits, Wd, dWd, dbd, dA, ldda, B, T, D, C, reinterpret_cast<char*>(ws) + wc, ws_bytes - wc);
}

}  // extern "C"
