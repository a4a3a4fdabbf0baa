// Experiment (never part of the product): one-way latency of a 4-byte hand-off between two workgroups, as a function of
// the cache-control bits on the store and the polling load, for a same-XCD pair and a cross-XCD pair.
//   hipcc --offload-arch=gfx950 -O3 -o pingpong tools/probes/pingpong.hip && ./pingpong
// Block b runs on XCD (b + const) % 8 (tools/probe_xcc.py), so blocks (0, 8) share an XCD and (0, 1) do not; the kernel
// re-checks with s_getreg XCC_ID and reports it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int AUX>
__device__ __forceinline__ void st(int* p, int v) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 4, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(v, r, 0, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ int ld(int* p) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 4, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, AUX);
}

// flags[0]: written by A, polled by B; flags[64]: written by B, polled by A (separate 256-byte lines)
template <int SAUX, int LAUX>
__global__ void k_pingpong(int* flags, int iters, int partner, long long* out, int* xcc) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (b == 0) xcc[0] = id & 0xf;
  if (b == partner) xcc[1] = id & 0xf;
  if (b != 0 && b != partner) return;
  int* mine = flags + (b == 0 ? 0 : 64);
  int* theirs = flags + (b == 0 ? 64 : 0);
  long long t0 = 0;
  int bad = 0;
  for (int i = 1; i <= iters; ++i) {
    if (i == 9) t0 = wall_clock64();
    if (b == 0) {
      st<SAUX>(mine, i);
      int spins = 0;
      while (ld<LAUX>(theirs) != i) { asm volatile("" ::: "memory"); if (++spins > (1 << 22)) { bad = 1; break; } }
    } else {
      int spins = 0;
      while (ld<LAUX>(theirs) != i) { asm volatile("" ::: "memory"); if (++spins > (1 << 22)) { bad = 1; break; } }
      st<SAUX>(mine, i);
    }
    if (bad) break;
  }
  long long t1 = wall_clock64();
  if (b == 0) { out[0] = bad ? -1 : (t1 - t0); }
}

template <int SAUX, int LAUX>
void run(const char* name, int* flags, long long* out, int* xcc, int partner) {
  const int iters = 2008;
  CHECK(hipMemset(flags, 0, 1024));
  hipLaunchKernelGGL((k_pingpong<SAUX, LAUX>), dim3(16), dim3(64), 0, 0, flags, iters, partner, out, xcc);
  CHECK(hipDeviceSynchronize());
  long long t; int x[2];
  CHECK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
  if (t < 0) printf("  %-34s partner=%2d xcc=(%d,%d): TIMEOUT (not visible)\n", name, partner, x[0], x[1]);
  else printf("  %-34s partner=%2d xcc=(%d,%d): one-way %7.1f ns\n", name, partner, x[0], x[1], t * 10.0 / (2.0 * (iters - 8)));
}

int main(int argc, char** argv) {
  int* flags; long long* out; int* xcc;
  const int kind = argc > 1 ? atoi(argv[1]) : 0;   // 0 hipMalloc, 1 uncached device memory, 2 fine-grained device memory
  if (kind == 1) { CHECK(hipExtMallocWithFlags((void**)&flags, 1024, hipDeviceMallocUncached)); printf("exchange words in UNCACHED device memory\n"); }
  else if (kind == 2) { CHECK(hipExtMallocWithFlags((void**)&flags, 1024, hipDeviceMallocFinegrained)); printf("exchange words in FINE-GRAINED device memory\n"); }
  else { CHECK(hipMalloc(&flags, 1024)); printf("exchange words in hipMalloc memory\n"); }
  CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&xcc, 64));
  // aux bits: 1 = sc0, 2 = nt, 16 = sc1
  for (int partner : {8, 1}) {
    printf("%s pair:\n", partner == 8 ? "same-XCD (expected)" : "cross-XCD (expected)");
    run<16, 16>("store sc1      / load sc1", flags, out, xcc, partner);
    run<17, 17>("store sc0 sc1  / load sc0 sc1", flags, out, xcc, partner);
    run<0, 16>("store plain    / load sc1", flags, out, xcc, partner);
    run<0, 1>("store plain    / load sc0", flags, out, xcc, partner);
    run<1, 1>("store sc0      / load sc0", flags, out, xcc, partner);
    run<0, 2>("store plain    / load nt", flags, out, xcc, partner);
    run<0, 3>("store plain    / load sc0 nt", flags, out, xcc, partner);
    run<2, 3>("store nt       / load sc0 nt", flags, out, xcc, partner);
    run<16, 1>("store sc1      / load sc0", flags, out, xcc, partner);
  }
  return 0;
}
