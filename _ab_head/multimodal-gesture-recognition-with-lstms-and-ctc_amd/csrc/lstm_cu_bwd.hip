// K7-scan on ONE CU per (direction, 16-sample group), split-f16 (round 6): BPTT of narrow LSTM layers (16 < H <= 128; the trainable
// fusion layer of multimodal_fusion/multimodal.py:159-168) WITHOUT an inter-CU exchange.
//
// Why (profiles/r06_schedule_probes.txt): in the pipelined training step the multi-CU BPTT of the fusion layer (7 workgroups per cluster,
// lstm_cluster_bwd.hip) takes 5.5 ms for 4.2 alone - and just as long in a form that gives every workgroup a CU of its own: what it pays
// is the latency of its exchange through an L2 / fabric the other stream's scans and GEMMs keep busy.  Round 4 dropped the single-CU
// kernel (lstm_mfma.hip) because v_mfma_f32_16x16x4_f32 made it matrix-bound (100 MFMAs of 35 cycles per wave and step: 4.9 us per
// step).  With the split-f16 product of the scans (three v_mfma_f32_16x16x32_f16 per f32 product, 16 cycles each, K = 32 per
// instruction) the same contraction is 42 MFMAs per wave and step: 0.3 us of matrix pipe, no hand-off, no dependence on what the
// rest of the chip does.
//
//   dh_rec[unit, sample] = sum over the 4H packed gate columns c of U[unit, c] dz_t[sample, c]
// Workgroup = 8 waves on one CU; wave w (w < MT = ceil(H / 16)) OWNS units 16 w .. 16 w + 15: lane (sample j = lane & 15, uq = lane >> 4)
// runs the cell backward of the four units 16 w + 4 e + uq (e = 0..3) for sample j - STRIDED, so that instruction e of the wave reads /
// writes, for every sample, the 64 contiguous bytes of units 4 e .. 4 e + 3 from its four lanes uq = 0..3 (with units 4 uq + e per lane
// every 16-byte piece of a store sat in a 64-byte line of its own: 1.5 of 4.8 us per step) - and the same wave computes dh_rec for
// exactly those cells: the rows of its A tile are permuted (row 4 uq + r = unit 4 r + uq), so the MFMA D layout (rows 4 uq + 0..3 of
// column j in lane (j, uq)) delivers them in place: dh never leaves its registers.  What the waves share is dz_t, the B operand - through
// LDS only: K is ordered so that a lane's 16 gate gradients ARE two B fragments of its own lane slot (K-block 2 w + blk holds its
// elements e = 2 blk, 2 blk + 1: k = 8 uq + 4 (e & 1) + gate): it writes its own (hi, lo) pairs, nobody gathers.
// A operand: U rows of the wave's 16 units against that K order, as f16 (hi, lo) fragments, stationary in 8 MT VGPR quads.
// Scaling as in lstm_cluster_bwd.hip (F16): U by the power of two that puts the workgroup's largest |U| in [2^14, 2^15); dz per source
// wave and step by the power of two that puts ITS largest |dz| in [2^14, 2^15), the factor left beside the image; one accumulator
// per source wave, the partial sums meet as f32, each multiplied by its source's inverse factor (exact scaling) in ascending wave order.
// Saved forward state comes by LDS-DMA: gates (16 B per cell) and dY rows into a two-slot ring per wave, fetched two steps ahead; the cell
// states into a four-slot ring fetched THREE steps ahead (a step needs c of its own time step and of the next iteration's).  One barrier
// per step (the dz image is double-buffered on the step parity).
#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) float lds_float;
constexpr int CB_WAVES = 8;

__device__ __forceinline__ void cb_dma_b128(const void* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");
}
template <int CTRL>
__device__ __forceinline__ float cb_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// (the arithmetic of lstm_cluster_bwd.hip's mgr_cell_bwd_tc: every fused operation written out)
__device__ __forceinline__ float4 cb_cell_bwd(float dh, float4 g4, float tc, float c_prev, float& dc_carry) {
#pragma clang fp contract(off)
  const float i = g4.x, f = g4.y, g = g4.z, o = g4.w;
  const float dO = dh * tc;
  const float dc = __builtin_fmaf(dh * o, __builtin_fmaf(-tc, tc, 1.f), dc_carry);
  const float di = dc * g, df = dc * c_prev, dg = dc * i;
  dc_carry = dc * f;
  return make_float4(di * mgr_hsig_grad(i), df * mgr_hsig_grad(f), dg * __builtin_fmaf(-g, g, 1.f), dO * mgr_hsig_grad(o));
}

template <int H>
struct CbCfg {
  static constexpr int MT = (H + 15) / 16;             // tiles of 16 units = active waves
  static constexpr int NKB = 2 * MT;                   // K-blocks of 32 gate columns (two per source wave)
  static constexpr int IMG = NKB * 2 * 64 * 4;         // floats of one dz image: [K-block][hi | lo][64 lanes] 16 bytes
  static constexpr int SLOT = (4 + 1) * 64 * 4;        // floats of one ring slot of a wave: gates [4 e][64 lanes] 16 B | dY [64 lanes] 16 B
  static constexpr int CSLOT = 64 * 4;                 // ... of one slot of its cell-state ring: c [64 lanes] 16 B
  static constexpr int LDS_FLOATS = 2 * IMG + 32 + MT * (2 * SLOT + 4 * CSLOT);
  static_assert(MT >= 2 && MT < CB_WAVES && (size_t)LDS_FLOATS * 4 <= 160 * 1024, "16 < H <= 112");
};

struct CuBwdJobs {
  const float* dY[MGR_MAX_SCAN_JOBS];
  const float* G[MGR_MAX_SCAN_JOBS];
  const float* Cs[MGR_MAX_SCAN_JOBS];
  const float* Up[MGR_MAX_SCAN_JOBS];
  float* dZ[MGR_MAX_SCAN_JOBS];
  int reverse[MGR_MAX_SCAN_JOBS];
};

template <int H>
__global__ __launch_bounds__(CB_WAVES * 64, 1) void k_scan_bwd_cu16(CuBwdJobs J, int lddy, int B, int T) {
  typedef CbCfg<H> C;
  constexpr int N = 4 * H, MT = C::MT, NKB = C::NKB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const float* __restrict__ dY = J.dY[blockIdx.y];
  const float* __restrict__ G = J.G[blockIdx.y];
  const float* __restrict__ Cs = J.Cs[blockIdx.y];
  const float* __restrict__ Up = J.Up[blockIdx.y];
  float* __restrict__ dZ = J.dZ[blockIdx.y];
  const int reverse = J.reverse[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, uq = lane >> 4;
  const int b = (int)blockIdx.x * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  const bool wact = wave < MT;                 // wave-uniform
  // this lane's units: 16 wave + 4 e + uq, e = 0..3; NE of them exist (H % 4 == 0: the same for every lane of the wave - no divergence)
  const int NE = __builtin_amdgcn_readfirstlane(!wact ? 0 : (H - 16 * wave >= 16 ? 4 : (H - 16 * wave) / 4));
  float* img = smem;                           // [2 parities] IMG
  float* scl = smem + 2 * C::IMG;              // [2][8] inverse factors of the source waves' dz; [16..23] prologue scratch
  float* ring = smem + 2 * C::IMG + 32 + (wact ? wave : 0) * (2 * C::SLOT + 4 * C::CSLOT);
  float* cring = ring + 2 * C::SLOT;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)ring);
  const unsigned cring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)cring);

  // ---- A fragments: tile = this wave's 16 units, row i = lane & 15 <-> unit 16 w + 4 (i & 3) + (i >> 2); K-block kb = 2 w' + blk,
  // element e8 of lane (i, kg = lane >> 4): gate column 4 su + (e8 & 3) with su = 16 w' + 4 (2 blk + (e8 >> 2)) + kg
  f16x8 ah[NKB], al[NKB];
  float sUinv;
  {
    auto uval = [&](int kb, int e) -> float {
      const int ur = 16 * wave + 4 * (j & 3) + (j >> 2), su = 16 * (kb >> 1) + 4 * (2 * (kb & 1) + (e >> 2)) + uq;
      return (wact && ur < H && su < H) ? Up[(size_t)ur * N + 4 * su + (e & 3)] : 0.f;
    };
    float umax = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int e = 0; e < 8; ++e) umax = fmaxf(umax, fabsf(uval(kb, e)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
    if (lane == 0) scl[16 + wave] = umax;
    __syncthreads();
    umax = 0.f;
#pragma unroll
    for (int w = 0; w < CB_WAVES; ++w) umax = fmaxf(umax, scl[16 + w]);
    int ex = 0;
    if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &ex);
    ex = ex < -60 ? -60 : ex;
    const float sU = ldexpf(1.f, 15 - ex);   // largest |U| sU in [2^14, 2^15)
    sUinv = ldexpf(1.f, ex - 15);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = uval(kb, e) * sU;
        asm volatile("" : "+v"(x));   // (hi and the residual from ONE f32 value)
        const _Float16 hi = (_Float16)x;
        ah[kb][e] = hi;
        al[kb][e] = (_Float16)(x - (float)hi);
      }
  }
  const unsigned sUinv_bits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sUinv));
  // both images start as zeros: the fragments of lanes without valid units are never written
  for (int i = tid; i < 2 * C::IMG; i += CB_WAVES * 64) img[i] = 0.f;

  // ---- saved state.  LDS-DMA: lane L's 16 bytes land at slot + 16 L.  Gates: instruction e fetches unit 16 w + 4 e + uq of sample j (four
  // lanes = 64 contiguous bytes); dY and c rows: lane (j, q) fetches the 16 bytes of units 16 w + 4 q .. + 3 of sample j, and the cell
  // of element e reads component uq of lane (j, e)'s piece.  Rows beyond H of the last tile re-read valid addresses and are ignored.
  const int ug = (16 * wave + uq < H && wact) ? 16 * wave + uq : 0;            // gates: + 4 e
  const int ur4 = (16 * wave + 4 * uq < H && wact) ? 16 * wave + 4 * uq : 0;   // dY / c pieces
  const unsigned goff = (unsigned)(((size_t)bc * T * H + ug) * 4 * sizeof(float));
  const unsigned coff = (unsigned)(((size_t)bc * T * H + ur4) * sizeof(float));
  const unsigned dyoff = (unsigned)(((size_t)bc * T * lddy + ur4) * sizeof(float));
  auto row_of = [&](int k) {
    const int n = T - 1 - k;
    return reverse ? T - 1 - n : n;
  };
  auto prefetch = [&](int k) {       // gates + dY of iteration k -> ring slot k & 1 (five instructions)
    if (wact && k < T) {
      const int t = row_of(k);
      const unsigned base = ring_lds + (unsigned)(k & 1) * (C::SLOT * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ec = 16 * wave + 4 * e < H ? e : 0;   // (wave-uniform)
        cb_dma_b128(G + ((size_t)t * H + 4 * ec) * 4, goff, base + e * 1024);
      }
      cb_dma_b128(dY + (size_t)t * lddy, dyoff, base + 4096);
    }
  };
  auto prefetch_c = [&](int k) {     // cell states of iteration k -> c ring slot k & 3 (one instruction)
    if (wact && k < T) cb_dma_b128(Cs + (size_t)row_of(k) * H, coff, cring_lds + (unsigned)(k & 3) * 1024);
  };
  prefetch(0);
  prefetch(1);
  prefetch_c(0);
  prefetch_c(1);
  prefetch_c(2);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the weight loads and the first ring slots (a wait hipcc can see)
  float dcc[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dhr = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int k = 0; k < T; ++k) {
    const int n = T - 1 - k;
    const int t = reverse ? T - 1 - n : n;
    const bool has_prev = n > 0;
    const int p = k & 1;
    // the saved state of this step and the cell state of the next iteration (c_{t-1}): both landed (the counted wait at the end of the
    // previous step covers everything that was issued two steps ago)
    const float* ru = ring + p * C::SLOT;
    const float* cu = cring + (k & 3) * C::CSLOT;
    const float* cn = cring + ((k + 1) & 3) * C::CSLOT;
    float4 dz[4];
    float dy[4], cc[4], cp[4];
    float4 g4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g4[e] = *reinterpret_cast<const float4*>(ru + e * 256 + lane * 4);
      dy[e] = ru[1024 + (16 * e + j) * 4 + uq];
      cc[e] = cu[(16 * e + j) * 4 + uq];
      cp[e] = has_prev ? cn[(16 * e + j) * 4 + uq] : 0.f;
    }
    if (wact) {   // (wave-uniform: ONE set of DMA instructions per wave and step - the counted wait below relies on it)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of these slots are done before the DMAs may overwrite them
      prefetch(k + 2);
      prefetch_c(k + 3);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dz[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NE) {   // (wave-uniform)
        dz[e] = cb_cell_bwd(dy[e] + dhr[e], g4[e], mgr_tanh(cc[e]), cp[e], dcc[e]);
        if (bvalid) *reinterpret_cast<float4*>(dZ + ((size_t)b * T + t) * N + (16 * wave + 4 * e + uq) * 4) = dz[e];
      }
    }
    if (!has_prev) break;   // the first forward step has no predecessor: nothing to hand on (workgroup-uniform; the last iteration)
    if (wact) {
      // this wave's factor: the power of two that puts its largest |dz| of the step in [2^14, 2^15) (scalar exponent arithmetic; zero,
      // Inf / NaN -> 2^15: a NaN gradient stays visible in dZ)
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fmaxf(fabsf(dz[e].x), fabsf(dz[e].y)), fmaxf(fabsf(dz[e].z), fabsf(dz[e].w))));
      m = fmaxf(m, cb_dpp<0x111>(m));   // row_shr:1, 2, 4, 8: lane 15 of a row holds the row's maximum
      m = fmaxf(m, cb_dpp<0x112>(m));
      m = fmaxf(m, cb_dpp<0x114>(m));
      m = fmaxf(m, cb_dpp<0x118>(m));
      const int mi = __float_as_int(m);
      m = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 15)), __int_as_float(__builtin_amdgcn_readlane(mi, 31))),
                fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 47)), __int_as_float(__builtin_amdgcn_readlane(mi, 63))));
      const int mb = __builtin_amdgcn_readfirstlane(__float_as_int(m));
      int e2 = ((mb >> 23) & 0xff) - 126;                       // m = f 2^e2, f in [0.5, 1)
      e2 = e2 < -100 ? -100 : e2;
      if (!(mb > 0 && mb < 0x7f61b1e6)) e2 = 0;                 // (zero, Inf, NaN)
      const float sz = __int_as_float((127 + 15 - e2) << 23);
      if (lane == 0) scl[p * 8 + wave] = __int_as_float((127 + e2 - 15) << 23) * __uint_as_float(sUinv_bits);
      {   // (every lane: elements beyond NE are zeros)
        float* im = img + p * C::IMG;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const float4 d0 = dz[2 * blk], d1 = dz[2 * blk + 1];
          float vs[8] = {d0.x * sz, d0.y * sz, d0.z * sz, d0.w * sz, d1.x * sz, d1.y * sz, d1.z * sz, d1.w * sz};
          f16x8 hi, lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            asm volatile("" : "+v"(vs[e]));
            const _Float16 h = (_Float16)vs[e];
            hi[e] = h;
            lo[e] = (_Float16)(vs[e] - (float)h);
          }
          *reinterpret_cast<f16x8*>(im + (((2 * wave + blk) * 2 + 0) * 64 + lane) * 4) = hi;
          *reinterpret_cast<f16x8*>(im + (((2 * wave + blk) * 2 + 1) * 64 + lane) * 4) = lo;
        }
      }
    }
    // what the next step reads (gates / dY of iteration k + 1, c of k + 1 and k + 2: issued a step ago or earlier) must have landed:
    // everything but the newest operations of this wave - this step's six DMAs and its NE dZ stores (the count is exact: a smaller one
    // would only wait longer, a larger one would let the previous step's DMAs slip); memory operations complete in issue order
    if (k + 3 >= T || NE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (NE == 4) {
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if (NE == 3) {
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    } else if (NE == 2) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    __syncthreads();   // the only barrier of the step: image and factors of parity p complete; those of parity p ^ 1 are free again
    if (wact) {
      // B fragments are read PD K-blocks ahead of the MFMAs that use them (the compiler, left alone, reads one K-block, waits, multiplies:
      // 14 LDS round trips in a row per step); sched_barriers keep the order: read kb + PD | three MFMAs of kb
      const float* im = img + p * C::IMG + lane * 4;
      constexpr int PD = 4;
      f16x8 bh[PD], bl[PD];
#pragma unroll
      for (int q = 0; q < PD && q < NKB; ++q) {
        bh[q] = *reinterpret_cast<const f16x8*>(im + (q * 2 + 0) * 256);
        bl[q] = *reinterpret_cast<const f16x8*>(im + (q * 2 + 1) * 256);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      // three INDEPENDENT accumulator chains per source wave (hi hi | lo hi | hi lo): a single chain of 42 dependent MFMAs paid the
      // matrix pipe's latency 42 times per step
      f32x4 t0 = zero, t1 = zero, t2 = zero;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const f16x8 ch = bh[kb % PD], cl = bl[kb % PD];
        __builtin_amdgcn_sched_barrier(0);
        if (kb + PD < NKB) {
          bh[kb % PD] = *reinterpret_cast<const f16x8*>(im + ((kb + PD) * 2 + 0) * 256);
          bl[kb % PD] = *reinterpret_cast<const f16x8*>(im + ((kb + PD) * 2 + 1) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb], ch, (kb & 1) ? t0 : zero, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kb], ch, (kb & 1) ? t1 : zero, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb], cl, (kb & 1) ? t2 : zero, 0, 0, 0);
        if (kb & 1) {   // the source wave kb >> 1 is complete: its partial sum times its inverse factor
          const float fs = scl[p * 8 + (kb >> 1)];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = fmaf((t0[r] + t1[r]) + t2[r], fs, acc[r]);
        }
      }
      dhr = acc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends
}

#define CB_FOREACH(X) X(32) X(64) X(100)   // (H = 128: eight tiles - image + rings would need 160.1 KiB of LDS; it keeps the multi-CU forms)

}  // namespace

bool mgr_scan_bwd_cu16_supported(int H) {
#define CB_CASE(HH) \
  if (H == HH) return true;
  CB_FOREACH(CB_CASE)
#undef CB_CASE
  return false;
}

// All jobs (same H, B, T, lddy: the two directions of a layer) as ONE launch, blockIdx.y = job: (B + 15) / 16 x njobs workgroups, a CU
// each.  No exchange between workgroups: the launch needs no entry in the residency ledger.  Returns 1 if launched, 0 if the shape has
// no instantiation or an operand is not laid out for the 16-byte LDS-DMA rows, < 0 on error.
int mgr_scan_bwd_cu16_multi(mgr_ctx* c, int njobs, const mgr_scan_bwd_job* jobs) {
  const int B = jobs[0].B, T = jobs[0].T, H = jobs[0].H, lddy = jobs[0].lddy;
  if (!mgr_scan_bwd_cu16_supported(H) || lddy % 4 != 0) return 0;
  if ((size_t)B * T * H * 16 >= ((size_t)1 << 32) || (size_t)B * T * lddy * 4 >= ((size_t)1 << 32)) return 0;   // (32-bit lane offsets)
  CuBwdJobs J;
  memset(&J, 0, sizeof(J));
  for (int i = 0; i < njobs; ++i) {
    if (jobs[i].B != B || jobs[i].T != T || jobs[i].H != H || jobs[i].lddy != lddy) return 0;
    if ((reinterpret_cast<uintptr_t>(jobs[i].dY) & 15) || (reinterpret_cast<uintptr_t>(jobs[i].cs) & 15)) return 0;
    J.dY[i] = jobs[i].dY; J.G[i] = jobs[i].gates; J.Cs[i] = jobs[i].cs; J.Up[i] = jobs[i].Up; J.dZ[i] = jobs[i].dZ;
    J.reverse[i] = jobs[i].reverse;
  }
  dim3 grid((B + 15) / 16, njobs);
  hipStream_t s = mgr_stream(c);
#define CB_CASE(HH)                                                                                                              \
  if (H == HH) {                                                                                                                 \
    const size_t lds = (size_t)CbCfg<HH>::LDS_FLOATS * sizeof(float);                                                            \
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_bwd_cu16<HH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    hipLaunchKernelGGL((k_scan_bwd_cu16<HH>), grid, dim3(CB_WAVES * 64), lds, s, J, lddy, B, T);                                  \
  }
  CB_FOREACH(CB_CASE)
#undef CB_CASE
  MGR_LAUNCH_CHECK();
  return 1;
}
