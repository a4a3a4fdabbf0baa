// K7-scan, multi-CU: BPTT of LSTM directions spread over clusters of CUs (the backward twin of lstm_cluster.hip).
//
// dh_rec_{t-1}[unit, sample] = sum over the 4H packed gate columns c of U[unit, c] * dz_t[sample, c].
// A cluster = the G = ceil(H/16) workgroups serving one (direction, 16-sample batch group).  Workgroup g OWNS the 16
// units S_g = [16g, 16g+16): their carried dc, their saved gates, and therefore their 64 gate columns of dz_t.
// The contraction is split over K (REDUCE-SCATTER), not over the output: with its own dz slice as the B operand
// (a 4 KiB LDS image) and the matching 64 columns of U stationary in VGPRs, workgroup g computes its partial sum for
// ALL units - G tiles of 16 units x 16 samples, v_mfma_f32_16x16x4_f32, 16 k-steps each - and sends tile m to workgroup
// m as ONE contiguous 1 KiB block (a 16-byte write-through store per lane: eight whole 128-byte lines).  Workgroup g
// then only has to fetch the G-1 tiles addressed to it (G-1 KiB) instead of the whole dz_t (4H x 16 floats, 4x more):
// dh is 4x smaller than dz, so reducing partial dh beats all-gathering dz.  The sums are formed in a fixed order
// (wave w adds sources g' = w, w+4, ... ascending, then waves 0..3 through LDS) so results are run-to-run identical.
// Hand-off: the data is the flag - the mantissa LSB of every exchanged word carries the epoch parity (a 1-ulp
// perturbation of a partial sum); two slots per (destination, source) pair suffice (see lstm_cluster.hip).
// After the reduction every thread runs the cell backward for ONE (unit, sample): 256 threads = 16 units x 16 samples;
// saved forward state and dY are prefetched two steps ahead by LDS-DMA through 3-deep per-wave LDS rings.
// Bounded spins, status word, one launch for all concurrently scanned directions (co-residency by construction).
//
// Round 3 (what the forward K-split step taught, lstm_cluster.hip): (i) memory operations complete in issue order, so the
// prefetch of the saved state (three HBM misses) must be issued BEHIND the gather of the step, not in front of it; (ii) any load
// hipcc can still see pending at the head of the time loop makes it put an s_waitcnt vmcnt(0) there and in front of the
// gathered data, which waits for those misses AND for the acknowledgement of the wave's own stores - so the saved state comes
// by LDS-DMA (no register destination), the weight loads are retired by a visible wait before the loop, and the status word is
// read through an opaque asm.  H = 100 (config F's fusion layer): 3.2 -> see profiles/r03_*; H = 300 / 500 (split roles) likewise.
#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;
constexpr unsigned POLL_LIMIT = 1u << 20;
constexpr int BW_WAVES = 4;
constexpr int BW_RING_FLOATS = 3 * (256 + 64 + 64);   // per compute wave: 3 slots x { gates [64] float4 | dy [64] | c [64] }
constexpr int BW16_RING_FLOATS = 4 * (256 + 64 + 64);  // cluster_bwd_run16: 4 slots, fetched three steps ahead
constexpr int BW16_IMG = 4 * 2 * 64 * 2;               // cluster_bwd_run16: floats of one B-operand image: [source wave][hi | lo][lane] 8 bytes
constexpr int BW_LDS_FLOATS_A = 4 * 256 + 5 * 256 + BW_WAVES * BW_RING_FLOATS + 16;   // (+16: the four dz factors of the split-f16 form)
constexpr int BW_LDS_FLOATS_B = 2 * BW16_IMG + 16 + BW_WAVES * BW16_RING_FLOATS + 1024;
constexpr int BW_LDS_FLOATS = BW_LDS_FLOATS_A > BW_LDS_FLOATS_B ? BW_LDS_FLOATS_A : BW_LDS_FLOATS_B;   // (what a caller may assume at most)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// LDS-DMA: one wave-instruction copies 64 x 16 B (64 x 4 B) from global memory [gbase + voff] (gbase wave-uniform, voff per
// lane) to LDS [lds_addr + 16 (4) * lane]; M0 carries the LDS address and is restored (hipcc does not know it was touched)
__device__ __forceinline__ void mgr_dma_b128(const void* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void mgr_dma_b32(const void* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");
}

// SPLIT = false: 4 waves, each gathers its share of the incoming partial tiles AND computes / stores its outgoing ones.
// SPLIT = true : 8 waves, one workgroup per CU.  Waves 0-3 only compute and store, waves 4-7 only gather: a gather wave
//   never stores, so the s_waitcnt in front of its gathered data no longer covers write-through stores (on gfx9 stores and
//   loads share vmcnt; the acknowledgement of a wave's 8 x 16-byte stores per step was 2.7 of 7.8 us at H = 500), and the
//   compute waves never wait on vmcnt for the exchange at all.  Plain compiler-managed loads - no register polling.
// F16 (round 4): the partial products on the f16 matrix pipe with split-f16 operands (lstm_cluster.hip, cluster_run_k16).  The B
// operand is the workgroup's own dz slice, a gate GRADIENT without an a-priori bound: every wave scales what IT contributes by the
// power of two that puts its own largest |dz| of the step in [2^14, 2^15) and leaves the factor beside the image; K is ordered so
// that a wave's 16 gate columns are ONE K-block of v_mfma_f32_16x16x16_f16 (k = 16 wave + 4 uq + gate: the four gate gradients of a
// thread's cell are exactly its lane's operand of that block - it writes its own (hi, lo) pair, nobody gathers), the K loop keeps
// one accumulator per source wave and the four partial sums meet as f32, each divided by its source's factor: exact scaling, no
// maximum to agree on, no extra barrier.  12 MFMAs of 16 cycles per tile instead of 16 of 35.
#ifdef MGR_STAMP
__device__ unsigned long long g_bstamps[64];
#define BSTAMP(i, dep) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); \
    st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
extern "C" int mgr_debug_bstamps(unsigned long long* out) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamps), sizeof(g_bstamps));
  unsigned long long z[64] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), z, sizeof(z));
  return 0;
}
#else
#define BSTAMP(i, dep) do { } while (0)
#endif
// DPP with the lane's own value where a row has no source (ctc.hip, dpp_f32)
template <int CTRL>
__device__ __forceinline__ float bw_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// mgr_cell_bwd with tanh(c) given: both forms of the step below go through THIS function with an opaque tc, so that hipcc contracts the
// cell arithmetic the same way in both (their results are compared bit for bit)
__device__ __forceinline__ float4 mgr_cell_bwd_tc(float dh, float4 g4, float tc, float c_prev, float& dc_carry) {
#pragma clang fp contract(off)   // (every fused operation below is written out: no context-dependent contraction)
  const float i = g4.x, f = g4.y, g = g4.z, o = g4.w;
  const float dO = dh * tc;
  const float dc = __builtin_fmaf(dh * o, __builtin_fmaf(-tc, tc, 1.f), dc_carry);
  const float di = dc * g, df = dc * c_prev, dg = dc * i;
  dc_carry = dc * f;
  return make_float4(di * mgr_hsig_grad(i), df * mgr_hsig_grad(f), dg * __builtin_fmaf(-g, g, 1.f), dO * mgr_hsig_grad(o));
}
template <int H, bool SPLIT, bool F16>
__device__ __forceinline__ void cluster_bwd_run(const ClusterBwdJob& jb, int bg, int ug, float* smem, unsigned* status, bool fast) {
  constexpr int N = 4 * H;
  constexpr int GT = (H + 15) / 16;          // tiles of 16 units = workgroups per cluster
  constexpr int TPW = (GT + BW_WAVES - 1) / BW_WAVES;  // tiles per wave (tile m = wave + 4*i)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool gatherer = SPLIT ? wave_id >= BW_WAVES : true;    // runs step 1 (gather)
  const bool computer = SPLIT ? wave_id < BW_WAVES : true;     // runs steps 2 and 3 (cell backward, MFMAs, stores)
  const int wave = SPLIT ? (wave_id & (BW_WAVES - 1)) : wave_id;   // role-local wave index 0..3
  const int j = lane & 15, uq = lane >> 4;
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  float* dzi = smem;                 // [4 blocks][4 gates][16 samples][4] own dz image: k-step s = own unit s, kk = gate
  float* red = smem + 4 * 256;       // [4 waves (+1: own tile, SPLIT)][64 lanes][4] partial sums of the tiles addressed to this workgroup
  // saved-state rings of the compute wave `wave`: [3 slots] x { gates [64] float4 | dy [64] | c [64] }
  float* ring = smem + 4 * 256 + 5 * 256 + wave * BW_RING_FLOATS;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)ring);

  // A fragments: tile m (units 16m..16m+15) x this workgroup's 64 gate columns: k-step s (own unit s), kk = gate
  //   A[i = lane&15][kk = lane>>4] = Up[unit 16m+i][4*(16*ug + s) + kk]
  float uf[F16 ? 1 : TPW][16];
  f16x4 ah[F16 ? TPW : 1][4], al[F16 ? TPW : 1][4];   // F16: tile i, K-block w' (source wave): U[16 m + j][64 ug + 16 uq + 4 w' + e], e < 4
  float* scl = smem + 4 * 256 + 5 * 256 + BW_WAVES * BW_RING_FLOATS;   // [4] 1 / (factor of wave w's dz), [8..11] prologue scratch
  float sUinv = 1.f;
  if constexpr (F16) {
    auto uval = [&](int i, int wsrc, int e) -> float {
      const int m = wave + BW_WAVES * i, ur = m * 16 + j, su = ug * 16 + 4 * uq + wsrc;
      return (computer && m < GT && ur < H && su < H) ? jb.Up[(size_t)ur * N + 4 * su + e] : 0.f;
    };
    float umax = 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int wsrc = 0; wsrc < 4; ++wsrc)
#pragma unroll
        for (int e = 0; e < 4; ++e) umax = fmaxf(umax, fabsf(uval(i, wsrc, e)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
    if (lane == 0 && computer) scl[8 + wave] = umax;
    if (tid < 4) scl[tid] = 0.f;
    __syncthreads();
    umax = fmaxf(fmaxf(scl[8], scl[9]), fmaxf(scl[10], scl[11]));
    int ex = 0;
    if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &ex);
    ex = ex < -60 ? -60 : ex;
    const float sU = ldexpf(1.f, 15 - ex);   // largest |U| sU in [2^14, 2^15)
    sUinv = ldexpf(1.f, ex - 15);
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int wsrc = 0; wsrc < 4; ++wsrc)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = uval(i, wsrc, e) * sU;
          asm volatile("" : "+v"(x));   // (hi and the residual from ONE f32 value: gemm.hip, mgr_split_f16)
          const _Float16 hi = (_Float16)x;
          ah[i][wsrc][e] = hi;
          al[i][wsrc][e] = (_Float16)(x - (float)hi);
        }
  } else {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int m = wave + BW_WAVES * i;
      const int ur = m * 16 + j;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int su = ug * 16 + s;
        uf[i][s] = (computer && m < GT && ur < H && su < H) ? jb.Up[(size_t)ur * N + 4 * su + uq] : 0.f;
      }
    }
  }
  for (int i = tid; i < 4 * 256; i += (int)blockDim.x) dzi[i] = 0.f;

  // cell backward ownership: unit = 16*ug + 4*uq + wave (row 4*(lane>>4)+reg of the reduced tile, reg = wave), sample j
  const int unit = ug * 16 + uq * 4 + wave;
  const bool uvalid = unit < H;
  // exchange slots: [slot][dest GT][src GT][256 floats]
  constexpr int SLOT = GT * GT * 256;
  float* xb = jb.xbuf + (size_t)bg * 2 * SLOT;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * SLOT * 4, 0x00020000);

  // byte offsets of this lane's (sample, unit) within dY / gates / cs at t = 0 (lanes without a unit read unit 0: a valid address);
  // the launcher admits only tensors below 4 GiB to this kernel
  const int ul = uvalid ? unit : 0;
  const unsigned dyoff = (unsigned)(((size_t)bc * T * jb.lddy + ul) * sizeof(float));
  const unsigned goff = (unsigned)(((size_t)bc * T * H + ul) * 4 * sizeof(float));
  const unsigned coff = (unsigned)(((size_t)bc * T * H + ul) * sizeof(float));
  auto prefetch = [&](int k) {       // saved state of iteration k -> ring slot k % 3 (three DMAs; everything wave-uniform but the offsets)
    if (computer && k < T) {
      const int n = T - 1 - k;
      const int t = reverse ? T - 1 - n : n;
      const unsigned base = ring_lds + (unsigned)(k % 3) * (BW_RING_FLOATS / 3 * 4);
      mgr_dma_b128(jb.gates + (size_t)t * H * 4, goff, base);
      mgr_dma_b32(jb.dY + (size_t)t * jb.lddy, dyoff, base + 1024);
      mgr_dma_b32(jb.cs + (size_t)t * H, coff, base + 1280);
    }
  };
  prefetch(0);
  prefetch(1);
  // (a wait hipcc can see: with the weight loads retired here its scoreboard enters the time loop empty - lstm_cluster.hip)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  float dcc = 0.f;
  float4 zmx = make_float4(0.f, 0.f, 0.f, 0.f);   // largest |dz| of this thread's (sample, unit) per gate over all steps (jb.dzmax)
  float4 zsm = make_float4(0.f, 0.f, 0.f, 0.f);   // and the sum of its dz per gate, in step order (jb.dbsum)
  f32x4 own_tile = {0.f, 0.f, 0.f, 0.f};  // the partial tile this workgroup computed for itself (held by wave ug % 4)
  bool failed = false;
  __syncthreads();

#ifdef MGR_STAMP
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev));
#endif
  for (int k = 0; k < T; ++k) {
    const int n = T - 1 - k;
    const int t = reverse ? T - 1 - n : n;
    const bool has_prev = n > 0;
    // ---- 1. reduce the partial tiles addressed to this workgroup (published at iteration k-1)
    float dhr = 0.f;
    if (k > 0) {
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      if (GT > 1 && gatherer) {
        const int slot = (k - 1) & 1;
        const unsigned par = (((unsigned)(k - 1) >> 1) & 1u) ^ 1u;
        u32x4 v[TPW];
        unsigned pend = 0;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int src = wave + BW_WAVES * i;
          if (src < GT && src != ug) pend |= 1u << i;
        }
        unsigned spins = 0;
        while (pend && !failed) {
#pragma unroll
          for (int i = 0; i < TPW; ++i)
            if (pend & (1u << i))
              v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * SLOT + (ug * GT + wave + BW_WAVES * i) * 256 + lane * 4) * 4, 0, 16);  // sc1
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            if (pend & (1u << i)) {
              const unsigned a = par ? (v[i].x & v[i].y & v[i].z & v[i].w) : (v[i].x | v[i].y | v[i].z | v[i].w);
              if (__all((a & 1u) == par)) pend &= ~(1u << i);
            }
          }
          if (pend) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 255u) == 0) {   // (opaque to hipcc, waited for on the spot: no load of its own may stay pending)
              unsigned st;
              asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st) : "v"(status) : "memory");
              if (__builtin_amdgcn_readfirstlane(st) != 0u) failed = true;
            }
            if (spins > POLL_LIMIT) {
              failed = true;
              if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        // fixed summation order: sources wave, wave+4, ... ascending (own tile in its place)
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int src = wave + BW_WAVES * i;
          if (src < GT) {
            if (src == ug) {
              if (!SPLIT) sum += own_tile;   // (SPLIT: the own tile lives in a compute wave and goes through red[4])
            } else {
              sum[0] += __uint_as_float(v[i].x);
              sum[1] += __uint_as_float(v[i].y);
              sum[2] += __uint_as_float(v[i].z);
              sum[3] += __uint_as_float(v[i].w);
            }
          }
        }
      } else if (GT <= 1) {
        if (wave == 0 && computer) sum = own_tile;
      }
      BSTAMP(0, sum[0]);
      prefetch(k + 2);   // behind the gather of this step (memory operations complete in issue order)
      if (SPLIT && GT > 1) {
        if (gatherer) *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = sum;
        if (computer && wave == (ug & (BW_WAVES - 1))) *reinterpret_cast<f32x4*>(red + (4 * 64 + lane) * 4) = own_tile;
      } else if (computer) {
        *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = sum;
      }
      __syncthreads();
      dhr = red[(0 * 64 + lane) * 4 + wave] + red[(1 * 64 + lane) * 4 + wave] + red[(2 * 64 + lane) * 4 + wave] +
            red[(3 * 64 + lane) * 4 + wave];
      if (SPLIT && GT > 1) dhr += red[(4 * 64 + lane) * 4 + wave];
    } else {
      prefetch(k + 2);
    }
    BSTAMP(1, dhr);
    // ---- 2. cell backward for (unit, sample); own dz slice -> global dZ and the LDS B-operand image
    float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (computer) {
      // the saved state of iterations k and k + 1 was fetched one and two steps ago; the only DMAs of this wave that may still be
      // in flight are the three of iteration k + 2: a counted wait makes the landing explicit (in practice it never waits)
      // (the last two iterations issue no prefetch: the three newest operations are then older DMAs / stores, not those of
      // iteration k + 2 - wait for everything there; two steps of 1900)
      if (k + 2 < T)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (computer && uvalid) {
      const float* ru = ring + (k % 3) * (BW_RING_FLOATS / 3);
      const float* rp = ring + ((k + 1) % 3) * (BW_RING_FLOATS / 3);
      const float4 ug4 = *reinterpret_cast<const float4*>(ru + lane * 4);
      const float dh = ru[256 + lane] + dhr;
      const float cp = has_prev ? rp[320 + lane] : 0.f;
      float tc = mgr_tanh(ru[320 + lane]);
      asm volatile("" : "+v"(tc));
      dz = mgr_cell_bwd_tc(dh, ug4, tc, cp, dcc);
      if (bvalid) *reinterpret_cast<float4*>(jb.dZ + ((size_t)b * T + t) * N + unit * 4) = dz;
      zmx = make_float4(fmaxf(zmx.x, fabsf(dz.x)), fmaxf(zmx.y, fabsf(dz.y)), fmaxf(zmx.z, fabsf(dz.z)), fmaxf(zmx.w, fabsf(dz.w)));
      zsm = make_float4(zsm.x + dz.x, zsm.y + dz.y, zsm.z + dz.z, zsm.w + dz.w);
    }
    BSTAMP(2, dz.x);
    if (!has_prev) break;   // the first forward step has no predecessor: nothing to send (workgroup-uniform; it is the last iteration)
    if (computer) {
      if constexpr (F16) {
        // this wave's factor, then the thread's own lane operand of K-block `wave`: image [wave][hi | lo][lane] 8 bytes
        float m = fmaxf(fmaxf(fabsf(dz.x), fabsf(dz.y)), fmaxf(fabsf(dz.z), fabsf(dz.w)));
        float sz;
        if constexpr (SPLIT) {
          // (round 5, the wide layers' split-role kernel: DPP row maxima + v_readlane + scalar exponent arithmetic instead of six
          //  ds_bpermute round trips and frexpf / ldexpf - the same factor bit for bit, cluster_bwd_run16 below; the 4-wave form keeps
          //  the old sequence: it is the one that runs beside the encoder scans of config F, where every change of its timing is a
          //  change of the whole step, profiles/r05_bptt_probes.txt)
          m = fmaxf(m, bw_dpp<0x111>(m));   // row_shr:1, 2, 4, 8: lane 15 of a row holds the row's maximum
          m = fmaxf(m, bw_dpp<0x112>(m));
          m = fmaxf(m, bw_dpp<0x114>(m));
          m = fmaxf(m, bw_dpp<0x118>(m));
          const int mi = __float_as_int(m);
          m = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 15)), __int_as_float(__builtin_amdgcn_readlane(mi, 31))),
                    fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 47)), __int_as_float(__builtin_amdgcn_readlane(mi, 63))));
          const int mb = __builtin_amdgcn_readfirstlane(__float_as_int(m));
          int e2 = ((mb >> 23) & 0xff) - 126;
          e2 = e2 < -100 ? -100 : e2;
          if (!(mb > 0 && mb < 0x7f61b1e6)) e2 = 0;   // (0x7f61b1e6 = 3.0e38f; zero, Inf, NaN)
          sz = __int_as_float((127 + 15 - e2) << 23);
          if (lane == 0) scl[wave] = __int_as_float((127 + e2 - 15) << 23) * sUinv;
        } else {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
          int e2 = 0;
          if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e2);
          e2 = e2 < -100 ? -100 : e2;
          sz = ldexpf(1.f, 15 - e2);   // (an Inf / NaN gradient: NaN products - it stays visible in dZ)
          if (lane == 0) scl[wave] = ldexpf(1.f, e2 - 15) * sUinv;
        }
        float vs[4] = {dz.x * sz, dz.y * sz, dz.z * sz, dz.w * sz};
        f16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          asm volatile("" : "+v"(vs[e]));
          const _Float16 h = (_Float16)vs[e];
          hi[e] = h;
          lo[e] = (_Float16)(vs[e] - (float)h);
        }
        *reinterpret_cast<f16x4*>(dzi + ((wave * 2) * 64 + lane) * 2) = hi;
        *reinterpret_cast<f16x4*>(dzi + ((wave * 2 + 1) * 64 + lane) * 2) = lo;
      } else {
        // own unit index s = 4*uq + wave -> image [q = s>>2 = uq][kk = gate][j][r = s&3 = wave]
        float* p = dzi + ((uq * 4) * 16 + j) * 4 + wave;
        p[0 * 64] = dz.x;
        p[1 * 64] = dz.y;
        p[2 * 64] = dz.z;
        p[3 * 64] = dz.w;
      }
    }
    BSTAMP(3, dz.y);
    __syncthreads();
    BSTAMP(4, dz.y);
    // ---- 3. partial sums for every tile of 16 units from this workgroup's 64 gate columns; send tile m to workgroup m
    if (computer) {
      const int slot = k & 1;
      const unsigned par = (((unsigned)k >> 1) & 1u) ^ 1u;
      f32x4 dv[4];
      f16x4 bh[4], bl[4];
      float fs[4];
      if constexpr (F16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bh[q] = *reinterpret_cast<const f16x4*>(dzi + ((q * 2) * 64 + lane) * 2);
          bl[q] = *reinterpret_cast<const f16x4*>(dzi + ((q * 2 + 1) * 64 + lane) * 2);
          fs[q] = scl[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[q] = *reinterpret_cast<const f32x4*>(dzi + ((q * 4 + uq) * 16 + j) * 4);
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int m = wave + BW_WAVES * i;
        if (m < GT) {  // wave-uniform
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (F16) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 as[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) as[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i][q], bh[q], zero, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) as[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[i][q], bh[q], as[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) as[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i][q], bl[q], as[q], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) a0[r] = fmaf(as[3][r], fs[3], fmaf(as[2][r], fs[2], fmaf(as[1][r], fs[1], as[0][r] * fs[0])));
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (r & 1)
                  a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i][4 * q + r], dv[q][r], a1, 0, 0, 0);
                else
                  a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i][4 * q + r], dv[q][r], a0, 0, 0, 0);
              }
            }
            a0 += a1;
          }
          if (m == ug) {
            // (kept in registers - with the parity bit a published tile would carry, so that every form of the step, the direct
            //  gather of cluster_bwd_run16 included, sums the same words)
#pragma unroll
            for (int r = 0; r < 4; ++r) own_tile[r] = __uint_as_float((__float_as_uint(a0[r]) & ~1u) | par);
          } else {
            u32x4 w;
            w.x = (__float_as_uint(a0[0]) & ~1u) | par;
            w.y = (__float_as_uint(a0[1]) & ~1u) | par;
            w.z = (__float_as_uint(a0[2]) & ~1u) | par;
            w.w = (__float_as_uint(a0[3]) & ~1u) | par;
            if (fast)   // whole cluster on one XCD (verified at start): plain store into the L2 the peers' sc1 loads are served from
              __builtin_amdgcn_raw_buffer_store_b128(w, rs, (slot * SLOT + (m * GT + ug) * 256 + lane * 4) * 4, 0, 0);
            else
              __builtin_amdgcn_raw_buffer_store_b128(w, rs, (slot * SLOT + (m * GT + ug) * 256 + lane * 4) * 4, 0, 16);  // sc1
          }
        }
      }
    }
    BSTAMP(5, dz.z);
  }
#ifdef MGR_STAMP
  if (lane == 0 && computer) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_bstamps[i], st_acc[i]);
    atomicAdd(&g_bstamps[8], (unsigned long long)T);
  }
#endif
  if (jb.dzmax && computer && uvalid && bvalid)   // (fmaxf drops a NaN: a NaN gradient shows in dZ itself, not here)
    *reinterpret_cast<float4*>(jb.dzmax + (size_t)b * N + unit * 4) = zmx;
  if (jb.dbsum && computer && uvalid && bvalid) *reinterpret_cast<float4*>(jb.dbsum + (size_t)b * N + unit * 4) = zsm;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: the split-f16 step of the NARROW layers (4 waves, 2 <= G <= 8) trimmed along its dependent chain, for launches that have
// the chip to themselves.  Cycle stamps of cluster_bwd_run<100, false, true> per step, alone: gather 1544 | partial sums through LDS
// + barrier 1199 | cell 759 | factor + image 793 | barrier 94 | MFMA + publish 1002 = 5.4 k cycles for 24 MFMAs.  Here:
//   * the saved state of the step is read from a FOUR-slot ring (fetched three steps ahead: what a step reads at its top was
//     retired by the previous step's counted wait) and tanh(c) is computed while the gather is in flight;
//   * the wave's dz factor comes from DPP row maxima + v_readlane and scalar exponent arithmetic instead of six ds_bpermute round
//     trips and frexpf / ldexpf;
//   * the B-operand image and the factors are double-buffered on the step parity; the retry loop of the gather is scalar.
// Same arithmetic, same summation order, same exchange layout: bit-identical to cluster_bwd_run<H, false, true>.  H = 100: 2.29 ->
// 1.77 us per step alone, H = 128: 2.12 -> 1.76; configuration S 4.62 -> 4.15 ms per step (profiles/r05_bptt_probes.txt).
// NOT used beside other persistent launches (the fusion layer of config F under the encoder scans): there the step is paced by
// contention, not by this chain, and the shorter chain takes issue slots from the encoder scans (18.2 -> 18.4 ms per step).
// Measured on the way and not kept (same notes): every wave gathering the words of its own cells from all G sources - no partial
// sums through LDS, ONE barrier per step (1.58 us alone at H = 100 with 16-byte elements, 1.76 with component-major tiles; 20.2 /
// 19.1 ms per step in config F: 28 KiB instead of 6 per workgroup and step through the texture path the encoder scans saturate).
// DIRECT: every wave gathers the words of ITS OWN cells from all G sources (one dword per source and lane: component `wave` of the
// 16-byte element a source lane published; a workgroup's own tile travels through the exchange like the others) and sums them in the
// same order - no partial sums through LDS, ONE barrier per step.  1.58 us per step alone at H = 100 (1.77 without), and 28 KiB per
// workgroup and step through the texture path instead of 6: it lost 1 - 2 ms per step while encoder-scan workgroups shared its CUs,
// and is the form the engine asks for (tune key 16 = 2) once the fused encoder scans leave the fusion layer CUs of its own.
// FUSED (round 6, k_scan_cluster_bwd16_f): the workgroup has 512 threads and runs TWO unit groups of one cluster - threads 0..255 the
// member 2 j, threads 256..511 the member 2 j + 1 - each through this function with its own half of the LDS; they share the CU and
// the barriers (the same count in both halves: two in the prologue, per step one in front of the reduction unless DIRECT, one behind the
// image), nothing else: the tiles the two halves owe each other travel through the exchange like every other tile.  H = 100: 4
// workgroups per cluster, 32 per launch, a CU each - instead of 56 four-wave workgroups on the 48 CUs the fused encoder scans leave.
template <int H, bool DIRECT = false, bool FUSED = false>
__device__ __forceinline__ void cluster_bwd_run16(const ClusterBwdJob& jb, int bg, int ug, float* smem, unsigned* status, bool fast) {
  constexpr int N = 4 * H;
  constexpr int GT = (H + 15) / 16;
  constexpr int TPW = (GT + BW_WAVES - 1) / BW_WAVES;
  static_assert(GT >= 2 && GT <= 8, "narrow layers with an exchange");
  const int tid = FUSED ? (int)(threadIdx.x & 255u) : (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, uq = lane >> 4;
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  float* dzi = smem;                          // [2 step parities] BW16_IMG
  float* scl = smem + 2 * BW16_IMG;           // [2][4] 1 / (factor of wave w's dz); [8..11] prologue scratch
  // saved-state ring of this wave: [4 slots] x { gates [64] float4 | dy [64] | c [64] }, fetched THREE steps ahead: what step k reads
  // at its top (slots k and k + 1) was retired by the counted wait of step k - 1
  float* ring = smem + 2 * BW16_IMG + 16 + wave * BW16_RING_FLOATS;
  float* red = smem + 2 * BW16_IMG + 16 + BW_WAVES * BW16_RING_FLOATS;   // [4 waves][64 lanes][4]
  f32x4 own_tile = {0.f, 0.f, 0.f, 0.f};
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)ring);

  // A fragments (as cluster_bwd_run, F16): tile i = units 16 (wave + 4 i) .., K-block w' = the 16 gate columns of source wave w'
  f16x4 ah[TPW][4], al[TPW][4];
  float sUinv;
  {
    auto uval = [&](int i, int wsrc, int e) -> float {
      const int m = wave + BW_WAVES * i, ur = m * 16 + j, su = ug * 16 + 4 * uq + wsrc;
      return (m < GT && ur < H && su < H) ? jb.Up[(size_t)ur * N + 4 * su + e] : 0.f;
    };
    float umax = 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int wsrc = 0; wsrc < 4; ++wsrc)
#pragma unroll
        for (int e = 0; e < 4; ++e) umax = fmaxf(umax, fabsf(uval(i, wsrc, e)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
    if (lane == 0) scl[8 + wave] = umax;
    __syncthreads();
    umax = fmaxf(fmaxf(scl[8], scl[9]), fmaxf(scl[10], scl[11]));
    int ex = 0;
    if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &ex);
    ex = ex < -60 ? -60 : ex;
    const float sU = ldexpf(1.f, 15 - ex);   // largest |U| sU in [2^14, 2^15)
    sUinv = ldexpf(1.f, ex - 15);
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int wsrc = 0; wsrc < 4; ++wsrc)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = uval(i, wsrc, e) * sU;
          asm volatile("" : "+v"(x));   // (hi and the residual from ONE f32 value: gemm.hip, mgr_split_f16)
          const _Float16 hi = (_Float16)x;
          ah[i][wsrc][e] = hi;
          al[i][wsrc][e] = (_Float16)(x - (float)hi);
        }
  }
  const unsigned sUinv_bits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sUinv));

  // cell backward ownership: unit = 16 ug + 4 uq + wave (component `wave` of the 16-byte element lane (j, uq) holds of a tile), sample j
  const int unit = ug * 16 + uq * 4 + wave;
  const bool uvalid = unit < H;
  constexpr int SLOT = GT * GT * 256;   // exchange slots: [slot][dest GT][src GT][256 floats]
  float* xb = jb.xbuf + (size_t)bg * 2 * SLOT;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * SLOT * 4, 0x00020000);

  const int ul = uvalid ? unit : 0;
  const unsigned dyoff = (unsigned)(((size_t)bc * T * jb.lddy + ul) * sizeof(float));
  const unsigned goff = (unsigned)(((size_t)bc * T * H + ul) * 4 * sizeof(float));
  const unsigned coff = (unsigned)(((size_t)bc * T * H + ul) * sizeof(float));
  auto prefetch = [&](int k) {       // saved state of iteration k -> ring slot k & 3 (three DMAs; everything wave-uniform but the offsets)
    if (k < T) {
      const int n = T - 1 - k;
      const int t = reverse ? T - 1 - n : n;
      const unsigned base = ring_lds + (unsigned)(k & 3) * (BW16_RING_FLOATS / 4 * 4);
      mgr_dma_b128(jb.gates + (size_t)t * H * 4, goff, base);
      mgr_dma_b32(jb.dY + (size_t)t * jb.lddy, dyoff, base + 1024);
      mgr_dma_b32(jb.cs + (size_t)t * H, coff, base + 1280);
    }
  };
  prefetch(0);
  prefetch(1);
  prefetch(2);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the weight loads and the first ring slots (a wait hipcc can see)
  float dcc = 0.f;
  float4 zmx = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 zsm = make_float4(0.f, 0.f, 0.f, 0.f);
  bool failed = false;
  unsigned spins = 0;
  __syncthreads();
#ifdef MGR_STAMP
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev));
#endif

  for (int k = 0; k < T; ++k) {
    const int n = T - 1 - k;
    const int t = reverse ? T - 1 - n : n;
    const bool has_prev = n > 0;
    const int p = k & 1;
    // ---- 1. the saved state of this step (its DMAs were retired a step ago) and what does not depend on dh, under the gather
    const float* ru = ring + (k & 3) * (BW16_RING_FLOATS / 4);
    const float* rp = ring + ((k + 1) & 3) * (BW16_RING_FLOATS / 4);
    const unsigned sbase = (unsigned)__builtin_amdgcn_readfirstlane(((k - 1) & 1) * SLOT * 4);   // (an SGPR operand: no waterfall loop)
    const bool gather = k > 0 && !failed;
    u32x4 v4[TPW];
    unsigned vd[GT];
    const unsigned g4base = (unsigned)((ug * GT * 256 + lane * 4) * 4);
    const unsigned gdbase = g4base + 4u * (unsigned)wave;   // DIRECT: component `wave` of the lane's element
    if (gather) {
      if constexpr (DIRECT) {
#pragma unroll
        for (int sx = 0; sx < GT; ++sx) vd[sx] = __builtin_amdgcn_raw_buffer_load_b32(rs, gdbase + sx * 1024u, sbase, 16);   // sc1
      } else {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int src = wave + BW_WAVES * i;
          if (src < GT && src != ug) v4[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, g4base + src * 1024u, sbase, 16);
        }
      }
    }
    const float4 ug4 = *reinterpret_cast<const float4*>(ru + lane * 4);
    const float dy = ru[256 + lane];
    float tc = mgr_tanh(ru[320 + lane]);
    const float cp = has_prev ? rp[320 + lane] : 0.f;
    asm volatile("" : "+v"(tc));            // (computed HERE, under the gather - not sunk to its use behind the verification)
    __builtin_amdgcn_sched_barrier(0);
    float dhr = 0.f;
    if constexpr (DIRECT) {
      if (gather) {
        const unsigned par = (((unsigned)(k - 1) >> 1) & 1u) ^ 1u;
        for (bool again = false;; again = true) {   // (every exit lies behind a verification: lstm_cluster.hip, cluster_run_k16)
          if (again) {
#pragma unroll
            for (int sx = 0; sx < GT; ++sx) vd[sx] = __builtin_amdgcn_raw_buffer_load_b32(rs, gdbase + sx * 1024u, sbase, 16);
          }
          unsigned a_and = 0xFFFFFFFFu, a_or = 0u;
#pragma unroll
          for (int sx = 0; sx < GT; ++sx) {
            a_and &= vd[sx];
            a_or |= vd[sx];
          }
          const bool fresh = par ? (a_and & 1u) != 0u : (a_or & 1u) == 0u;
          if (__builtin_amdgcn_readfirstlane((int)(__all(fresh) || failed))) break;
          spins = (unsigned)__builtin_amdgcn_readfirstlane((int)(spins + 1u));
          if ((spins & 255u) == 0) {
            unsigned st;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st) : "v"(status) : "memory");
            if (__builtin_amdgcn_readfirstlane(st) != 0u) failed = true;
          }
          if (spins > POLL_LIMIT) {
            failed = true;
            if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          failed = __builtin_amdgcn_readfirstlane((int)failed) != 0;
          if (failed) break;
        }
        // the order of the other forms: wave w' summed its sources w', w' + 4 (ascending, from zero), then waves 0..3 in turn
        float pw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          pw[w] = 0.f;
#pragma unroll
          for (int i = 0; i < TPW; ++i)
            if (w + BW_WAVES * i < GT) pw[w] += __uint_as_float(vd[w + BW_WAVES * i]);
        }
        dhr = pw[0] + pw[1] + pw[2] + pw[3];
      }
    } else
    {
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      if (gather) {
        const unsigned par = (((unsigned)(k - 1) >> 1) & 1u) ^ 1u;
        for (bool again = false;; again = true) {
          if (again) {
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
              const int src = wave + BW_WAVES * i;
              if (src < GT && src != ug) v4[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, g4base + src * 1024u, sbase, 16);
            }
          }
          unsigned a_and = 0xFFFFFFFFu, a_or = 0u;
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            const int src = wave + BW_WAVES * i;
            if (src < GT && src != ug) {
              a_and &= v4[i].x & v4[i].y & v4[i].z & v4[i].w;
              a_or |= v4[i].x | v4[i].y | v4[i].z | v4[i].w;
            }
          }
          const bool fresh = par ? (a_and & 1u) != 0u : (a_or & 1u) == 0u;
          if (__builtin_amdgcn_readfirstlane((int)(__all(fresh) || failed))) break;
          spins = (unsigned)__builtin_amdgcn_readfirstlane((int)(spins + 1u));
          if ((spins & 255u) == 0) {
            unsigned st;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st) : "v"(status) : "memory");
            if (__builtin_amdgcn_readfirstlane(st) != 0u) failed = true;
          }
          if (spins > POLL_LIMIT) {
            failed = true;
            if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          failed = __builtin_amdgcn_readfirstlane((int)failed) != 0;
          if (failed) break;
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int src = wave + BW_WAVES * i;
          if (src < GT) {
            if (src == ug) {
              sum += own_tile;
            } else {
              sum[0] += __uint_as_float(v4[i].x);
              sum[1] += __uint_as_float(v4[i].y);
              sum[2] += __uint_as_float(v4[i].z);
              sum[3] += __uint_as_float(v4[i].w);
            }
          }
        }
      }
      if (k > 0) {
        *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = sum;
        __syncthreads();
        dhr = red[(0 * 64 + lane) * 4 + wave] + red[(1 * 64 + lane) * 4 + wave] + red[(2 * 64 + lane) * 4 + wave] + red[(3 * 64 + lane) * 4 + wave];
      }
    }
    BSTAMP(0, dhr);
    prefetch(k + 3);   // behind the gather of this step (memory operations complete in issue order)
    // the saved state of iterations k + 1 and k + 2 must have landed before the next step reads it: everything but the three newest
    // operations (the DMAs of iteration k + 3; the last three iterations issue none: wait for everything there)
    if (k + 3 < T)
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BSTAMP(1, dhr);
    // ---- 2. cell backward for (unit, sample); own dz slice -> global dZ and the LDS B-operand image
    float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (uvalid) {
      dz = mgr_cell_bwd_tc(dy + dhr, ug4, tc, cp, dcc);
      if (bvalid) *reinterpret_cast<float4*>(jb.dZ + ((size_t)b * T + t) * N + unit * 4) = dz;
      zmx = make_float4(fmaxf(zmx.x, fabsf(dz.x)), fmaxf(zmx.y, fabsf(dz.y)), fmaxf(zmx.z, fabsf(dz.z)), fmaxf(zmx.w, fabsf(dz.w)));
      zsm = make_float4(zsm.x + dz.x, zsm.y + dz.y, zsm.z + dz.z, zsm.w + dz.w);
    }
    BSTAMP(2, dz.x);
    if (!has_prev) break;   // the first forward step has no predecessor: nothing to send (workgroup-uniform; it is the last iteration)
    {
      // this wave's factor: the power of two that puts its largest |dz| of the step in [2^14, 2^15) (wave-uniform: scalar arithmetic
      // on the exponent field; zero, Inf / NaN -> 2^15 as before - a NaN gradient stays visible in dZ)
      float m = fmaxf(fmaxf(fabsf(dz.x), fabsf(dz.y)), fmaxf(fabsf(dz.z), fabsf(dz.w)));
      m = fmaxf(m, bw_dpp<0x111>(m));   // row_shr:1, 2, 4, 8: lane 15 of a row holds the row's maximum
      m = fmaxf(m, bw_dpp<0x112>(m));
      m = fmaxf(m, bw_dpp<0x114>(m));
      m = fmaxf(m, bw_dpp<0x118>(m));
      const int mi = __float_as_int(m);
      m = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 15)), __int_as_float(__builtin_amdgcn_readlane(mi, 31))),
                fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 47)), __int_as_float(__builtin_amdgcn_readlane(mi, 63))));   // (fmaxf drops a NaN as before)
      const int mb = __builtin_amdgcn_readfirstlane(__float_as_int(m));
      int e2 = ((mb >> 23) & 0xff) - 126;                       // m = f 2^e2, f in [0.5, 1) (denormals: below the clamp anyway)
      e2 = e2 < -100 ? -100 : e2;
      if (!(mb > 0 && mb < 0x7f61b1e6)) e2 = 0;                 // (0x7f61b1e6 = 3.0e38f; zero, Inf, NaN)
      const float sz = __int_as_float((127 + 15 - e2) << 23);
      if (lane == 0) scl[p * 4 + wave] = __int_as_float((127 + e2 - 15) << 23) * __uint_as_float(sUinv_bits);
      float vs[4] = {dz.x * sz, dz.y * sz, dz.z * sz, dz.w * sz};
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        asm volatile("" : "+v"(vs[e]));
        const _Float16 h = (_Float16)vs[e];
        hi[e] = h;
        lo[e] = (_Float16)(vs[e] - (float)h);
      }
      float* img = dzi + p * BW16_IMG;
      *reinterpret_cast<f16x4*>(img + ((wave * 2) * 64 + lane) * 2) = hi;
      *reinterpret_cast<f16x4*>(img + ((wave * 2 + 1) * 64 + lane) * 2) = lo;
    }
    BSTAMP(3, dz.y);
    __syncthreads();   // the only barrier of the step: image and factors of parity p complete; those of parity p ^ 1 are free again
    BSTAMP(4, dz.y);
    // ---- 3. partial sums for every tile of 16 units from this workgroup's 64 gate columns; send tile m to workgroup m
    {
      const int slot = k & 1;
      const unsigned par = (((unsigned)k >> 1) & 1u) ^ 1u;
      const float* img = dzi + p * BW16_IMG;
      f16x4 bh[4], bl[4];
      float fs[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bh[q] = *reinterpret_cast<const f16x4*>(img + ((q * 2) * 64 + lane) * 2);
        bl[q] = *reinterpret_cast<const f16x4*>(img + ((q * 2 + 1) * 64 + lane) * 2);
        fs[q] = scl[p * 4 + q];
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int m = wave + BW_WAVES * i;
        if (m < GT) {  // wave-uniform
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          f32x4 as[4], a0;
#pragma unroll
          for (int q = 0; q < 4; ++q) as[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i][q], bh[q], zero, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) as[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[i][q], bh[q], as[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) as[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[i][q], bl[q], as[q], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) a0[r] = fmaf(as[3][r], fs[3], fmaf(as[2][r], fs[2], fmaf(as[1][r], fs[1], as[0][r] * fs[0])));
          if (m == ug && !DIRECT) {
            // (kept in registers - with the parity bit a published tile would carry, so that every form of the step, the direct
            //  gather included, sums the same words)
#pragma unroll
            for (int r = 0; r < 4; ++r) own_tile[r] = __uint_as_float((__float_as_uint(a0[r]) & ~1u) | par);
          } else {
            u32x4 w;
            w.x = (__float_as_uint(a0[0]) & ~1u) | par;
            w.y = (__float_as_uint(a0[1]) & ~1u) | par;
            w.z = (__float_as_uint(a0[2]) & ~1u) | par;
            w.w = (__float_as_uint(a0[3]) & ~1u) | par;
            if (fast)
              __builtin_amdgcn_raw_buffer_store_b128(w, rs, (slot * SLOT + (m * GT + ug) * 256 + lane * 4) * 4, 0, 0);
            else
              __builtin_amdgcn_raw_buffer_store_b128(w, rs, (slot * SLOT + (m * GT + ug) * 256 + lane * 4) * 4, 0, 16);  // sc1
          }
        }
      }
    }
    BSTAMP(5, dz.z);
  }
#ifdef MGR_STAMP
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_bstamps[i], st_acc[i]);
    atomicAdd(&g_bstamps[8], (unsigned long long)T);
    atomicAdd(&g_bstamps[9], (unsigned long long)spins);
  }
#endif
  if (jb.dzmax && uvalid && bvalid)   // (fmaxf drops a NaN: a NaN gradient shows in dZ itself, not here)
    *reinterpret_cast<float4*>(jb.dzmax + (size_t)b * N + unit * 4) = zmx;
  if (jb.dbsum && uvalid && bvalid) *reinterpret_cast<float4*>(jb.dbsum + (size_t)b * N + unit * 4) = zsm;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends
}

#define BW_SMALL(X) X(8) X(16) X(32) X(64) X(100) X(128)
#define BW_LARGE(X) X(300) X(500)
#define BW_FOREACH(X) BW_SMALL(X) BW_LARGE(X)

// SMALL: every job of the launch is narrow (H <= 128) - its own kernel, so that the fusion layer's BPTT (56 workgroups beside
// the projection GEMMs of the other stream) is allocated ~100 VGPRs instead of the 256 the H = 500 instantiation needs
template <bool SPLIT, bool SMALL = false, bool F16 = false, int LEAN = 0>
__device__ __forceinline__ void scan_cluster_bwd_body(const ClusterBwdLaunch& L, float* smem) {
  mgr_cluster_enter(L.cm);
  const int bid = blockIdx.x;
  for (int k = 0; k < L.njobs; ++k) {
    const ClusterBwdJob& jb = L.job[k];
    const int w = bid - jb.cls_begin;
    int ug, cl;
    bool fast = false;
    if (L.xcd_local) {   // octet layout + same-XCD verification: lstm_cluster.h, mgr_cluster_octet
      if (w < 0 || w >= (jb.cls_nclusters + 7) / 8 * 8 * jb.G_) continue;
      fast = mgr_cluster_octet(L.cm, jb.cls_begin, jb.G_, jb.cls_rot, w, cl, ug);
      if (cl >= jb.cls_nclusters) continue;
    } else {
      // members of a cluster are CONTIGUOUS workgroup ids (the round-robin dispatcher then spreads them over all XCDs)
      if (w < 0 || w >= jb.cls_nclusters * jb.G_) continue;
      ug = w % jb.G_;
      cl = w / jb.G_;
    }
    const int bg = cl - jb.cls_cluster0;
    if (bg < 0 || bg >= jb.nbg) continue;
#define BW_CASE(HH)                                                                                                      \
  if (jb.H == HH) {                                                                                                      \
    if constexpr (!SPLIT && SMALL && F16 && (LEAN != 0) && (HH > 16))                                                            \
      cluster_bwd_run16<HH, LEAN == 2>(jb, bg, ug, smem, L.cm.status, fast);                                             \
    else                                                                                                                 \
      cluster_bwd_run<HH, SPLIT, F16>(jb, bg, ug, smem, L.cm.status, fast);                                              \
    return mgr_cluster_exit(L.cm);                                                                                       \
  }
    if constexpr (SMALL) {
      BW_SMALL(BW_CASE)
    } else {
      BW_FOREACH(BW_CASE)
    }
#undef BW_CASE
    return;
  }
}

__global__ __launch_bounds__(BW_WAVES * 64) void k_scan_cluster_bwd_s(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<false, true>(L, smem);
}

__global__ __launch_bounds__(BW_WAVES * 64) void k_scan_cluster_bwd(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<false>(L, smem);
}

// split roles: 4 compute + 4 gather waves, one workgroup per CU
__global__ __launch_bounds__(2 * BW_WAVES * 64) void k_scan_cluster_bwd_split(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<true>(L, smem);
}

// the same three with split-f16 operands (cluster_bwd_run<.., true>; tune key 14 = 1 keeps the f32 MFMA kernels above)
__global__ __launch_bounds__(BW_WAVES * 64) void k_scan_cluster_bwd16_s(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<false, true, true>(L, smem);
}
// narrow layers with the chip to themselves: cluster_bwd_run16
__global__ __launch_bounds__(BW_WAVES * 64) void k_scan_cluster_bwd16_sl(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<false, true, true, 1>(L, smem);
}
// ... with CUs of their own (beside fused encoder scans): the direct gather, one barrier per step
__global__ __launch_bounds__(BW_WAVES * 64) void k_scan_cluster_bwd16_sd(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<false, true, true, 2>(L, smem);
}
// the fused form of the narrow layers (cluster_bwd_run16<H, DIRECT, true>): one 8-wave workgroup per CU = two unit groups (2 j, 2 j + 1) of
// one cluster; the launch lays out ceil(G / 2) members per cluster in XCD-local octets; a unit group beyond G (odd G) only keeps the
// barrier count of the step
template <bool DIRECT>
__device__ __forceinline__ void scan_cluster_bwd_fused_body(const ClusterBwdLaunch& L, float* smem) {
  mgr_cluster_enter(L.cm);
  const int tg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  for (int k_ = 0; k_ < L.njobs; ++k_) {
    const ClusterBwdJob& jb = L.job[k_];
    const int G = jb.G_, Gr = (G + 1) / 2;
    const int w = (int)blockIdx.x - jb.cls_begin;
    if (w < 0 || w >= (jb.cls_nclusters + 7) / 8 * 8 * Gr) continue;
    int cl, ugr;
    const bool fast = mgr_cluster_octet(L.cm, jb.cls_begin, Gr, jb.cls_rot, w, cl, ugr);
    if (cl >= jb.cls_nclusters) continue;
    const int bg = cl - jb.cls_cluster0;
    if (bg < 0 || bg >= jb.nbg) continue;
    const int ug = 2 * ugr + tg;
    float* sm = smem + tg * BW_LDS_FLOATS_B;
    if (ug >= G) {   // (odd G: the last workgroup's second half keeps the barrier count of cluster_bwd_run16)
      __syncthreads();
      __syncthreads();
      for (int k = 0; k < jb.T; ++k) {
        if (!DIRECT && k > 0) __syncthreads();
        if (k < jb.T - 1) __syncthreads();
      }
      return mgr_cluster_exit(L.cm);
    }
#define BWF_CASE(HH)                                                              \
  if (jb.H == HH) {                                                               \
    if constexpr ((HH) > 16) cluster_bwd_run16<HH, DIRECT, true>(jb, bg, ug, sm, L.cm.status, fast); \
    return mgr_cluster_exit(L.cm);                                                \
  }
    BW_SMALL(BWF_CASE)
#undef BWF_CASE
    return;
  }
}
__global__ __launch_bounds__(2 * BW_WAVES * 64, 1) void k_scan_cluster_bwd16_f(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_fused_body<false>(L, smem);
}
__global__ __launch_bounds__(2 * BW_WAVES * 64, 1) void k_scan_cluster_bwd16_fd(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_fused_body<true>(L, smem);
}
__global__ __launch_bounds__(BW_WAVES * 64, 2) void k_scan_cluster_bwd16(ClusterBwdLaunch L) {   // (two workgroups per CU: <= 256 VGPRs)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<false, false, true>(L, smem);
}
__global__ __launch_bounds__(2 * BW_WAVES * 64) void k_scan_cluster_bwd16_split(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_bwd_body<true, false, true>(L, smem);
}

}  // namespace

// floats of exchange memory per batch group and slot: [dest G][src G][256]
size_t mgr_cluster_bwd_img_floats(int H) {
  size_t g = (size_t)(H + 15) / 16;
  return g * g * 256;
}

bool mgr_cluster_bwd_supported(int H) {
#define BW_CASE(HH) \
  if (H == HH) return true;
  BW_FOREACH(BW_CASE)
#undef BW_CASE
  return false;
}

// split roles (8 waves, > 80 KiB of LDS requested so that exactly one workgroup sits on a CU) whenever the launch fits one
// workgroup per CU, has an exchange at all and the layers are wide: at H = 100 (7 workgroups per cluster, two 16-byte stores
// per lane and step) the store acknowledgement is 0.35 of 2.2 us and the 8-wave workgroups cost config F 0.7 % end to end,
// at H = 300 / 500 they save a third of the step (E: 60 -> 48 ms/step, S_ref 24 -> 21).  tune key 8: 1 = never, 2 = always.
static bool bwd_split(const mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs) {
  int maxH = 0;
  bool exchange = false;
  for (int i = 0; i < L.njobs; ++i) {
    maxH = L.job[i].H > maxH ? L.job[i].H : maxH;
    exchange = exchange || L.job[i].G_ > 1;
  }
  return exchange && (maxH >= 200 || c->tune[8] == 2) && total_wgs <= c->cu_count && c->tune[8] != 1;
}

// the fused form runs narrow layers (16 < H <= 128) on the split-f16 path, laid out in XCD-local octets
bool mgr_cluster_bwd_fusable(const mgr_ctx* c, const ClusterBwdLaunch& L) {
  bool ok = c->tune[14] == 0 && L.xcd_local && L.njobs > 0;
  for (int i = 0; i < L.njobs; ++i) ok = ok && L.job[i].H > 16 && L.job[i].H <= 128 && L.job[i].G_ >= 2;
  return ok;
}

void mgr_cluster_bwd_geometry(const mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs, int* waves, int* per_cu) {
  if (L.fused) {
    *waves = 2 * BW_WAVES;
    *per_cu = 1;
    return;
  }
  const bool split = bwd_split(c, L, total_wgs);
  *waves = split ? 2 * BW_WAVES : BW_WAVES;
  *per_cu = split ? 1 : 2;
}

int mgr_cluster_bwd_launch(mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs, int form16) {
  const bool alone = form16 == 0;
  MGR_REQUIRE(total_wgs <= 2 * c->cu_count, "cluster BPTT needs %d co-resident workgroups", total_wgs);
  if (!(c->attr_done & 2u)) {
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd_split), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd_s), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd16_split), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd16_s), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd16_sl), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd16_sd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd16_f), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd16_fd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->attr_done |= 2u;
  }
  const bool f16 = c->tune[14] == 0;   // split-f16 operands (tune key 14 = 1: f32 MFMA)
  if (L.fused) {
    MGR_REQUIRE(mgr_cluster_bwd_fusable(c, L) && total_wgs <= c->cu_count, "the fused BPTT form needs narrow split-f16 layers in octets, one workgroup per CU");
    // (>= 84 KiB requested: the workgroup sits alone on its CU whatever its registers would allow)
    size_t lds = 2 * (size_t)BW_LDS_FLOATS_B * sizeof(float);
    lds = lds < 84 * 1024 ? 84 * 1024 : lds;
    if (form16 == 2)
      hipLaunchKernelGGL(k_scan_cluster_bwd16_fd, dim3(total_wgs), dim3(2 * BW_WAVES * 64), lds, mgr_stream(c), L);
    else
      hipLaunchKernelGGL(k_scan_cluster_bwd16_f, dim3(total_wgs), dim3(2 * BW_WAVES * 64), lds, mgr_stream(c), L);
  } else if (bwd_split(c, L, total_wgs)) {
    size_t lds = 84 * 1024;
    if (f16)
      hipLaunchKernelGGL(k_scan_cluster_bwd16_split, dim3(total_wgs), dim3(2 * BW_WAVES * 64), lds, mgr_stream(c), L);
    else
      hipLaunchKernelGGL(k_scan_cluster_bwd_split, dim3(total_wgs), dim3(2 * BW_WAVES * 64), lds, mgr_stream(c), L);
  } else {
    size_t lds = (size_t)BW_LDS_FLOATS_A * sizeof(float);
    bool small = true;
    for (int i = 0; i < L.njobs; ++i) small = small && L.job[i].H <= 128;
    if (small && f16 && form16 == 2)
      hipLaunchKernelGGL(k_scan_cluster_bwd16_sd, dim3(total_wgs), dim3(BW_WAVES * 64), (size_t)BW_LDS_FLOATS_B * sizeof(float), mgr_stream(c), L);
    else if (small && f16 && alone)
      hipLaunchKernelGGL(k_scan_cluster_bwd16_sl, dim3(total_wgs), dim3(BW_WAVES * 64), (size_t)BW_LDS_FLOATS_B * sizeof(float), mgr_stream(c), L);
    else if (small && f16)
      hipLaunchKernelGGL(k_scan_cluster_bwd16_s, dim3(total_wgs), dim3(BW_WAVES * 64), lds, mgr_stream(c), L);
    else if (small)
      hipLaunchKernelGGL(k_scan_cluster_bwd_s, dim3(total_wgs), dim3(BW_WAVES * 64), lds, mgr_stream(c), L);
    else if (f16)
      hipLaunchKernelGGL(k_scan_cluster_bwd16, dim3(total_wgs), dim3(BW_WAVES * 64), lds, mgr_stream(c), L);
    else
      hipLaunchKernelGGL(k_scan_cluster_bwd, dim3(total_wgs), dim3(BW_WAVES * 64), lds, mgr_stream(c), L);
  }
  MGR_LAUNCH_CHECK();
  return 0;
}
