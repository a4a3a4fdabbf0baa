// HBM-bound elementwise / small reduction kernels: noise, dropout masks, weight packing, Adam+clip,
// max-norm, mean, frame argmax.  All are grid-stride, coalesced, one launch each.
#include "common.h"

namespace {

constexpr int kBlock = 256;
static inline int grid_for(size_t n, int per_block = kBlock) {
  size_t g = (n + per_block - 1) / per_block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void k_add_noise(const float* __restrict__ X, float* __restrict__ Y, size_t n, float stddev, uint64_t seed) {
  // Box-Muller on two 24-bit uniforms per pair of elements
  size_t npair = (n + 1) / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npair; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t r = mgr_mix64(seed * 0xD1342543DE82EF95ull + i);
    float u1 = ((float)((uint32_t)(r >> 40)) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
    float u2 = (float)((uint32_t)(r >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    float rad = sqrtf(-2.0f * logf(u1)) * stddev;
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    size_t e = 2 * i;
    Y[e] = X[e] + rad * c;
    if (e + 1 < n) Y[e + 1] = X[e + 1] + rad * s;
  }
}

__global__ void k_dropout_mask(float* __restrict__ m, size_t n, float p, float inv_keep, uint64_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m[i] = mgr_drop_scale(seed, i, p, inv_keep);
}

// [rows, 4H]: keras col g*H+u  <->  packed col u*4+g
__global__ void k_pack(const float* __restrict__ src, float* __restrict__ dst, int rows, int H, int to_keras) {
  size_t n = (size_t)rows * 4 * H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / (4 * H);
    int c = (int)(i % (4 * H));  // destination column
    int sc;
    if (to_keras) {  // dst keras col c = g*H+u  <- src packed u*4+g
      int g = c / H, u = c % H;
      sc = u * 4 + g;
    } else {  // dst packed col c = u*4+g <- src keras g*H+u
      int u = c / 4, g = c % 4;
      sc = g * H + u;
    }
    dst[i] = src[r * 4 * H + sc];
  }
}

__global__ void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];
  int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = by + j, c = bx + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = bx + j, r = by + tx;  // dst[c][r]
    if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[tx][j];
  }
}

// `gate` (may be null): device flag of the update gate (mgr_update_gate_set) - non-zero means a scan of this step reported a
// give-up / non-finite state (on any rank: the flag travels with the gradient all-reduce), the update is skipped as a whole and
// counted in the status block
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       size_t n, float lr_t, float b1, float b2, float eps, float clipvalue, float gscale,
                       const float* __restrict__ gate, unsigned* __restrict__ skipped) {
  if (gate && gate[0] != 0.f) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1u);
    return;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    if (clipvalue > 0.f) gi = fminf(fmaxf(gi, -clipvalue), clipvalue);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// one block per 32 columns; 32 row-groups (1024 threads) reduce through LDS: the matrix is small (1600 x 400 for the fusion
// layer), the kernel sits on the critical chain between the optimizer and the next step's projections and runs beside
// chip-filling GEMMs there, so what counts is the length of each thread's dependent load chain
constexpr int MN_RG = 32;
__global__ __launch_bounds__(32 * MN_RG) void k_maxnorm(float* __restrict__ W, int rows, int cols, float maxv, float eps,
                                                         const float* __restrict__ gate) {
  __shared__ float part[MN_RG][32];
  if (gate && gate[0] != 0.f) return;   // update gate closed (k_adam): the weights stay exactly as they were
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  int c = blockIdx.x * 32 + tx;
  float s = 0.f;
  if (c < cols)
    for (int r = ty; r < rows; r += MN_RG) {
      float w = W[(size_t)r * cols + c];
      s += w * w;
    }
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0) {
    float t = 0.f;
    for (int k = 0; k < MN_RG; ++k) t += part[k][tx];
    float nrm = sqrtf(t);
    part[0][tx] = fminf(fmaxf(nrm, 0.f), maxv) / (eps + nrm);
  }
  __syncthreads();
  float sc = part[0][tx];
  if (c < cols)
    for (int r = ty; r < rows; r += MN_RG) W[(size_t)r * cols + c] *= sc;
}

__global__ void k_add2d(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb, float* __restrict__ O,
                        int ldo, size_t rows, int cols) {
  size_t n = rows * (size_t)cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / cols;
    int c = (int)(i % cols);
    O[r * ldo + c] = A[r * lda + c] + Bm[r * ldb + c];
  }
}

__global__ void k_mean(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ double sh[kBlock];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sh[0] / (double)n);
}

// one thread per output frame; first index wins ties (numpy argmax)
__global__ void k_frame_argmax(const float* __restrict__ P, int B, int T, int C, int skip, int32_t* __restrict__ best,
                               float* __restrict__ prob) {
  int To = T - skip;
  size_t n = (size_t)B * To;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int b = (int)(i / To), t = (int)(i % To);
    const float* row = P + ((size_t)b * T + skip + t) * C;
    float mx = row[0];
    int am = 0;
    for (int c = 1; c < C; ++c) {
      float v = row[c];
      if (v > mx) {
        mx = v;
        am = c;
      }
    }
    best[i] = am;
    prob[i] = mx;
  }
}

__global__ void k_probe_xcc(int32_t* out) {
  extern __shared__ float dummy[];
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int32_t)(x & 0xF);
}

// Guest probe: what a collective's kernel (RCCL all-reduce: a few workgroups with tens of KiB of LDS that wait for other GPUs)
// experiences when it is launched beside resident persistent scans.  Block 0 of a 1-block launch is the MARKER (records when
// the stream reached this point); every block of the guest records when it started and when it left.  100 MHz wall clock.
__global__ void k_guest(long long* __restrict__ out, int us) {
  extern __shared__ float dummy[];
  const unsigned long long t0 = wall_clock64();
  if (us > 0) {
    const unsigned long long ticks = (unsigned long long)us * 100ull;
    for (int i = 0; i < (1 << 22); ++i) {
      if (wall_clock64() - t0 >= ticks) break;
      __builtin_amdgcn_s_sleep(16);
    }
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (long long)t0;
    out[2 * blockIdx.x + 1] = (long long)wall_clock64();
  }
}

}  // namespace

// Holds the stream for ~`us` microseconds (constant 100 MHz counter), bounded.  Used to order the PLACEMENT of two launches
// that become ready at the same moment on different streams (a persistent cluster scan must be resident before chip-filling
// GEMM waves arrive, see Engine.enqueue_train_step).
__global__ void k_delay(int us) {
  const unsigned long long t0 = wall_clock64();
  const unsigned long long ticks = (unsigned long long)us * 100ull;
  for (int i = 0; i < (1 << 22); ++i) {
    if (wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(16);
  }
}

extern "C" {

int mgr_stream_delay(mgr_ctx* c, int us) {
  MGR_REQUIRE(c && us >= 0 && us <= 100000, "bad argument");
  if (us == 0) return 0;
  hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, mgr_stream(c), us);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_probe_guest(mgr_ctx* c, int nblocks, int threads, int lds_bytes, int us, int64_t* out) {
  MGR_REQUIRE(c && out && nblocks > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && us >= 0 && us <= 100000,
              "bad argument");
  MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_guest), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(k_guest, dim3(1), dim3(64), 0, mgr_stream(c), reinterpret_cast<long long*>(out), 0);   // marker
  hipLaunchKernelGGL(k_guest, dim3(nblocks), dim3(threads), lds_bytes, mgr_stream(c), reinterpret_cast<long long*>(out) + 2, us);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_probe_xcc(mgr_ctx* c, int nblocks, int threads, int lds_bytes, int32_t* out) {
  MGR_REQUIRE(c && out && nblocks > 0 && threads > 0, "bad argument");
  MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe_xcc), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(k_probe_xcc, dim3(nblocks), dim3(threads), lds_bytes, mgr_stream(c), out);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_add_gaussian_noise(mgr_ctx* c, const float* X, float* Y, size_t n, float stddev, uint64_t seed) {
  MGR_REQUIRE(c && X && Y, "null argument");
  if (n == 0) return 0;
  mgr_prof_begin(c, MGR_K_MISC);
  hipLaunchKernelGGL(k_add_noise, dim3(grid_for((n + 1) / 2)), dim3(kBlock), 0, mgr_stream(c), X, Y, n, stddev, seed);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_MISC);
  return 0;
}

int mgr_dropout_mask(mgr_ctx* c, float* mask, size_t n, float p, uint64_t seed) {
  MGR_REQUIRE(c && mask, "null argument");
  MGR_REQUIRE(p >= 0.f && p < 1.f, "dropout rate %f out of [0,1)", p);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_dropout_mask, dim3(grid_for(n)), dim3(kBlock), 0, mgr_stream(c), mask, n, p, 1.0f / (1.0f - p), seed);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_lstm_pack(mgr_ctx* c, const float* src, float* dst, int rows, int H, int to_keras) {
  MGR_REQUIRE(c && src && dst && src != dst, "null or aliased argument");
  MGR_REQUIRE(rows > 0 && H > 0, "bad shape");
  hipLaunchKernelGGL(k_pack, dim3(grid_for((size_t)rows * 4 * H)), dim3(kBlock), 0, mgr_stream(c), src, dst, rows, H, to_keras);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_transpose(mgr_ctx* c, const float* src, float* dst, int rows, int cols) {
  MGR_REQUIRE(c && src && dst && src != dst, "null or aliased argument");
  MGR_REQUIRE(rows > 0 && cols > 0, "bad shape");
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(k_transpose, grid, dim3(kBlock), 0, mgr_stream(c), src, dst, rows, cols);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_adam_step(mgr_ctx* c, float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                  float eps, float clipvalue, float gscale) {
  MGR_REQUIRE(c && p && g && m && v, "null argument");
  if (n == 0) return 0;
  mgr_prof_begin(c, MGR_K_ADAM);
  hipLaunchKernelGGL(k_adam, dim3(grid_for(n)), dim3(kBlock), 0, mgr_stream(c), p, g, m, v, n, lr_t, b1, b2, eps, clipvalue, gscale,
                     c->gate_flag, mgr_status_block(c) + 2);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_ADAM);
  return 0;
}

int mgr_maxnorm_cols(mgr_ctx* c, float* W, int rows, int cols, float maxv, float eps) {
  MGR_REQUIRE(c && W, "null argument");
  MGR_REQUIRE(rows > 0 && cols > 0, "bad shape");
  hipLaunchKernelGGL(k_maxnorm, dim3((cols + 31) / 32), dim3(32 * MN_RG), 0, mgr_stream(c), W, rows, cols, maxv, eps, c->gate_flag);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_add2d(mgr_ctx* c, const float* A, int lda, const float* Bm, int ldb, float* Out, int ldo, size_t rows, int cols) {
  MGR_REQUIRE(c && A && Bm && Out, "null argument");
  MGR_REQUIRE(cols > 0 && lda >= cols && ldb >= cols && ldo >= cols, "bad shape");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_add2d, dim3(grid_for(rows * (size_t)cols)), dim3(kBlock), 0, mgr_stream(c), A, lda, Bm, ldb, Out, ldo, rows, cols);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_mean(mgr_ctx* c, const float* x, int n, float* out) {
  MGR_REQUIRE(c && x && out && n > 0, "bad argument");
  hipLaunchKernelGGL(k_mean, dim3(1), dim3(kBlock), 0, mgr_stream(c), x, n, out);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_frame_argmax(mgr_ctx* c, const float* P, int B, int T, int C, int skip, int32_t* best, float* prob) {
  MGR_REQUIRE(c && P && best && prob, "null argument");
  MGR_REQUIRE(B > 0 && T > skip && C > 0 && skip >= 0, "bad shape");
  hipLaunchKernelGGL(k_frame_argmax, dim3(grid_for((size_t)B * (T - skip))), dim3(kBlock), 0, mgr_stream(c), P, B, T, C, skip, best, prob);
  MGR_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
