// Does vector-ALU work overlap with back-to-back f32 MFMAs on gfx950?  One workgroup per CU.
//   mode 0: 4 waves (one per SIMD): NM MFMAs each, nothing else
//   mode 1: 4 waves: NV v_fma each, nothing else
//   mode 2: 4 waves: each MFMA followed by K independent v_fma (same wave)
//   mode 3: 8 waves (two per SIMD): waves 0-3 MFMAs only, waves 4-7 v_fma only (K per MFMA of the partner)
//   mode 4: like 3 with bf16 MFMAs (v_mfma_f32_16x16x16_bf16 ... 16x16x32) for comparison
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/probes/mfma_valu_coissue.hip ; run: /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NM = 4096;

template <int MODE, int K>
__global__ __launch_bounds__(512) void k(float* out, float a, float b) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
  const bool do_m = MODE == 0 || MODE == 2 || ((MODE == 3 || MODE == 4) && wave < 4);
  const bool do_v = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 4) && wave >= 4);
  if (MODE == 2) {
    for (int i = 0; i < NM; i += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
    }
  } else if (do_m) {
    for (int i = 0; i < NM; i += 2) {
      if (MODE == 4) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
      }
    }
  } else if (do_v) {
    for (int i = 0; i < NM; i += 2) {
#pragma unroll
      for (int q = 0; q < 2 * K; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
    }
  }
  float s = acc0[0] + acc1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int K>
float run(int threads, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(threads), 0, 0, d, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(threads), 0, 0, d, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3;   // us
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  printf("f32 MFMAs only (4 waves, %d each):          %8.1f us\n", NM, run<0, 0>(256, d));
  printf("v_fma only, 4 per MFMA slot (4 waves):        %8.1f us\n", run<1, 4>(256, d));
  printf("v_fma only, 8 per MFMA slot (4 waves):        %8.1f us\n", run<1, 8>(256, d));
  printf("same wave: MFMA + 2 v_fma:                    %8.1f us\n", run<2, 2>(256, d));
  printf("same wave: MFMA + 4 v_fma:                    %8.1f us\n", run<2, 4>(256, d));
  printf("same wave: MFMA + 6 v_fma:                    %8.1f us\n", run<2, 6>(256, d));
  printf("same wave: MFMA + 8 v_fma:                    %8.1f us\n", run<2, 8>(256, d));
  printf("two waves per SIMD: MFMA wave + 2 v_fma wave: %8.1f us\n", run<3, 2>(512, d));
  printf("two waves per SIMD: MFMA wave + 4 v_fma wave: %8.1f us\n", run<3, 4>(512, d));
  printf("two waves per SIMD: MFMA wave + 8 v_fma wave: %8.1f us\n", run<3, 8>(512, d));
  printf("bf16 16x16x32 x2 per slot + 4 v_fma wave:     %8.1f us\n", run<4, 4>(512, d));
  printf("bf16 16x16x32 x2 per slot + 8 v_fma wave:     %8.1f us\n", run<4, 8>(512, d));
  return 0;
}
