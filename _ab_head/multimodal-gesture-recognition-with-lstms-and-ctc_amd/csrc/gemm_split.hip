// K2 / K7 on PRE-SPLIT operands (round 5): the wide dropout-aware projection and weight-gradient products as loader + matrix
// pipelines - the operands arrive in HBM already as f16 (hi, lo) pairs of their scaled values, travel to LDS by LDS-DMA
// (global_load_lds_dwordx4: no register, no conversion, no ds_write) through a three-stage ring behind counted vmcnt waits, and the
// waves of a workgroup do nothing but fragment reads (ds_read_b64_tr_b16 where the K index is the slow one) and
// v_mfma_f32_32x32x16_f16.  gemm.hip's k_gemm_nn_sparse16 / k_gemm_tn_sparse16 did the f32 -> (hi, lo) conversion of every element
// while staging (~70 vector instructions per stage beside 6 MFMAs: the kernels were bound by their staging, profiles/r04_scan_probes.txt).
//
// SPLIT ROW FORMAT ("XS"): a transposed activation copy XT[b][f][ldt] of f32 (mgr.h, mgr_scan_job.YT) keeps its shape and strides;
// the 4 ldt bytes of row (b, f) hold ldt f16 values hi(t) followed by ldt f16 values lo(t) with  x 2^13 = hi + lo,  hi = rn_f16(x 2^13),
// lo = rn_f16(x 2^13 - hi)  (|x| < 7.99; exact scaling, 22+ significant bits - gemm.hip, mgr_split_f16).  The scans write it
// themselves (lstm_cluster.hip, mgr_scan_job.yt_split), mgr_transpose_bt_split makes it from a row-major tensor.
// Arithmetic: as everywhere on the split path, A B ~ (Ahi Bhi + Alo Bhi + Ahi Blo) / (sA sB) in ONE f32 accumulator.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short tr8 __attribute__((__vector_size__(8 * sizeof(short))));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) tr4 lds_tr4;

constexpr float XS_SCALE = 8192.f;       // 2^13: the scale of a split row
constexpr int PS_TM = 128, PS_SK = 32;   // rows (time steps) of a tile, kept features per stage
constexpr int PS_HP = 128;               // the weight planes are padded to whole unit tiles of the widest kernel

// LDS-DMA: 64 lanes x 16 B from global [gbase + voff] (voff per lane) to LDS [lds_addr + 16 lane]; M0 carries the LDS address
__device__ __forceinline__ void ps_dma_b128(const void* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");
}
// chunk swizzle of a [rows][256 B] image (16 chunks of 16 B per row): conflict-free for ds_read_b64_tr_b16 blocks of 4 rows x 16
// columns taken pairwise by a 32-lane half (the CDNA guide's image (b), T10)
__device__ __forceinline__ int ps_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// ---- lists of a call: per (gate, sample) the kept features, padded to whole stages of 32.  Entry = a | w << 16: a = feature whose
// activation row the position reads, w = row of the weight planes it meets - the feature itself, or (padding) a = the first kept
// feature, w = F, the planes' all-zero row.  cword: the ONE mask factor of the call as float bits (0: nothing kept anywhere);
// two different non-zero factors turn it into a NaN pattern - the products cannot carry a factor per (gate, sample, feature), and
// an input the kernel was not written for shows as NaN in Z, never as a plausible number.
__global__ __launch_bounds__(64) void k_lists32(const float* __restrict__ mask4, int F, int Fp32, int* __restrict__ lists, int* __restrict__ kcnt,
                                                int* __restrict__ kpos /* [4B][F] list position of a kept feature, -1 if dropped; may be null */,
                                                unsigned* __restrict__ cword) {
  const int gb = blockIdx.x, lane = threadIdx.x;
  int* out = lists + (size_t)gb * Fp32;
  int n = 0;
  unsigned seen = 0u;
  bool mixed = false;
  if (mask4) {
    const float* m = mask4 + (size_t)gb * F;
    for (int f0 = 0; f0 < F; f0 += 64) {
      const int f = f0 + lane;
      const float v = f < F ? m[f] : 0.f;
      const bool take = f < F && v != 0.f;
      const unsigned long long bal = __ballot(take);
      if (take) {
        out[n + __popcll(bal & ((1ull << lane) - 1ull))] = f | (f << 16);
        const unsigned vb = __float_as_uint(v);
        mixed = mixed || (seen != 0u && seen != vb);
        seen = vb;
      }
      if (kpos && f < F) kpos[(size_t)gb * F + f] = take ? n + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
      n += __popcll(bal);
    }
  } else {   // no mask: every feature kept, factor 1
    for (int f = lane; f < F; f += 64) {
      out[f] = f | (f << 16);
      if (kpos) kpos[(size_t)gb * F + f] = f;
    }
    n = F;
    seen = __float_as_uint(1.f);
  }
  // one factor per wave: lanes that kept something must agree
  const unsigned long long have = __ballot(seen != 0u);
  if (have) {
    const unsigned first = __shfl(seen, __ffsll((long long)have) - 1);
    mixed = __any(mixed || (seen != 0u && seen != first));
    if (lane == 0) {
      const unsigned old = atomicCAS(cword, 0u, first);
      if (mixed || (old != 0u && old != first)) atomicExch(cword, 0x7FC00000u);
    }
  }
  if (lane == 0) kcnt[gb] = n;
  __syncthreads();   // (one wave: orders the list writes above with the read of out[0] below)
  const int first_kept = n > 0 ? (out[0] & 0xFFFF) : 0;
  for (int i = n + lane; i < (n + 31) / 32 * 32; i += 64) out[i] = first_kept | (F << 16);
}

// largest |W| of the packed kernel, as float bits by atomic max (the word is zeroed by the caller)
__global__ __launch_bounds__(256) void k_wmax(const float* __restrict__ Wp, size_t n4, unsigned* __restrict__ wmax) {
  __shared__ float part[4];
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 w = reinterpret_cast<const float4*>(Wp)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  // ONE atomic per workgroup, 256 workgroups: 4096 atomics on one word (one per wave of 1024 workgroups) took 49 us of a 1.3 ms call
  if (threadIdx.x == 0) atomicMax(wmax, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
__device__ __forceinline__ float ps_wscale(const unsigned* wmax) {   // the power of two that puts the largest |W| in [2^14, 2^15)
  const float m = __uint_as_float(*wmax);
  int ex = 0;
  if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
  ex = ex < -60 ? -60 : ex;
  return m < 3.0e38f ? ldexpf(1.f, 15 - ex) : 0.f;    // (an Inf weight: scale 0, the output is NaN - visible)
}
__device__ __forceinline__ void ps_split(float x, _Float16& hi, _Float16& lo) {   // (gemm.hip, mgr_split_f16)
  asm volatile("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
// weight planes WS[g][f][hi Hp | lo Hp] f16 of W sw (gate-major, Hp = H rounded up to 64, zero beyond H, row F all zero):
// Wp[f][4u + g] is the packed kernel
__global__ __launch_bounds__(256) void k_wplanes(const float* __restrict__ Wp, _Float16* __restrict__ WS, int F, int H, int Hp,
                                                 const unsigned* __restrict__ wmax) {
  const float sw = ps_wscale(wmax);
  const size_t n = (size_t)(F + 1) * Hp;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int u = (int)(i % Hp), f = (int)(i / Hp);
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < F && u < H) w = *reinterpret_cast<const float4*>(Wp + ((size_t)f * H + u) * 4);
    const float g4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      _Float16 hi, lo;
      ps_split(g4[g] * sw, hi, lo);
      _Float16* row = WS + ((size_t)g * (F + 1) + f) * 2 * Hp;
      row[u] = hi;
      row[Hp + u] = lo;
    }
  }
}

// XS[b][f] = split row of X[b][0..T)[f] (zero for t >= T): the generic producer of the split row format.  tshift: entry t of a row is
// X[b][t + tshift][f], zero where that step does not exist (tshift = -1 / +1: the rows h_{t-1} / h_{t+1} the recurrent weight gradient of a
// forward / reverse direction multiplies dz_t with)
__global__ __launch_bounds__(256) void k_transpose_split(const float* __restrict__ X, int ldx, float* __restrict__ XS, int ldt, int T, int F,
                                                         long long xsb /* batch stride of XS in floats; 0: F * ldt */, int fill, int tshift) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* Xb = X + (size_t)b * T * ldx;
  float* XSb = XS + (size_t)b * (xsb ? (size_t)xsb : (size_t)F * ldt);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = t0 + ty + 4 * i, f = f0 + tx, ts = t + tshift;
    tile[ty + 4 * i][tx] = (t < T && ts >= 0 && ts < T && f < F) ? Xb[(size_t)ts * ldx + f] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = f0 + ty + 4 * i, t = t0 + tx;
    if (f < F && t < fill) {
      _Float16 hi, lo;
      ps_split(tile[tx][ty + 4 * i] * XS_SCALE, hi, lo);
      _Float16* row = reinterpret_cast<_Float16*>(XSb + (size_t)f * ldt);
      row[t] = hi;
      row[ldt + t] = lo;
    }
  }
}

// ------------------------------------------------------------------------------------------------ nn on pre-split operands
// Z[b, t, 4u + g] = bias + c sum_{f kept by (g, b)} X[b, t, f] W[f, 4u + g]      (c = the mask factor of the call)
// Tile: 128 time steps x TU = 32 WC units x 4 gates per workgroup of NW = 2 WC waves, wave (wr, wc) = 64 steps x 32 units; ONE continuous
// sequence of stages of 32 kept features through the four gates (the accumulator set changes, the pipeline does not drain).
//   WC = 2: 128 x 64, 4 waves, 3-stage ring of 24 KiB, two workgroups per CU
//   WC = 4: 128 x 128, 8 waves, 4-stage ring of 32 KiB, one workgroup per CU: half the A traffic per FLOP and three stages in flight
//           (the first version of this kernel - WC = 2 only - moved 11.8 GB from L2 / MALL into LDS per audio depth-2 call and was
//           bound by exactly that: profiles/r05_gemm_split_probes.txt)
// Per stage every wave issues its share of the LDS-DMA instructions (1 KiB each: two feature rows of the A image [k][hi 128 t | lo
// 128 t]; two or four feature rows of the B image [k][hi TU u | lo TU u]) NBUF - 1 stages ahead, waits for its OWN DMAs of the
// current stage with a counted vmcnt, meets the others at ONE barrier, and runs 24 transposed fragment reads + 12 MFMAs.  The kept-feature
// indices of a stage are wave-uniform: scalar loads, issued one iteration before the DMAs that use them and unpacked where they are used.
template <int WC>
struct PsCfg {
  static constexpr int NW = 2 * WC, TU = 32 * WC, NBUF = WC == 2 ? 3 : 4;
  static constexpr int B_ROW = 4 * TU;                     // bytes of one B image row: hi | lo
  static constexpr int A_BYTES = 2 * PS_SK * 256, B_BYTES = PS_SK * B_ROW, STAGE = A_BYTES + B_BYTES, LDS = NBUF * STAGE;
  static constexpr int NI = PS_SK / NW;                    // list entries per wave and stage (its A features = its B rows)
  static constexpr int AI = 16 / NW;                       // A instructions per wave and stage
  static constexpr int BI = B_BYTES / 1024 / NW;           // B instructions per wave and stage
  static constexpr int RB = 1024 / B_ROW;                  // B rows per instruction
};

template <int WC>
__global__ __launch_bounds__(128 * WC, WC == 2 ? 2 : 1) void k_proj_split(const char* __restrict__ XS, int ldt, const int* __restrict__ lists,
                                                                          const int* __restrict__ kcnt, const unsigned* __restrict__ cword,
                                                                          const char* __restrict__ WS, int Hp, const unsigned* __restrict__ wmax,
                                                                          const float* __restrict__ bp, float* __restrict__ Z, int B, int T,
                                                                          int Fp32, int F, int H) {
  typedef PsCfg<WC> C;
  extern __shared__ __attribute__((aligned(16))) char ps_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave / WC, wc = wave % WC;
  const int N = 4 * H;
  const int ncol = (H + C::TU - 1) / C::TU, nrow = (T + PS_TM - 1) / PS_TM;
  const int x = blockIdx.x & 7, jj = blockIdx.x >> 3;   // XCD-aware tile order (gemm.hip, k_gemm_nn_sparse): the column tiles of one
  const int rt = (jj / ncol) * 8 + x;                   // (sample, row tile) meet in one L2
  if (rt >= nrow * B) return;
  const int u0 = (jj % ncol) * C::TU, r0 = (rt % nrow) * PS_TM, b = rt / nrow;
  const char* XSb = XS + (size_t)b * F * ldt * 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)ps_smem;

  // ---- stage sequence: gate g has its own number of stages (scalars, not an array: a private array indexed by a run-time gate
  // would live in scratch memory)
  const int c1 = (__builtin_amdgcn_readfirstlane(kcnt[0 * B + b]) + PS_SK - 1) / PS_SK;
  const int c2 = c1 + (__builtin_amdgcn_readfirstlane(kcnt[1 * B + b]) + PS_SK - 1) / PS_SK;
  const int c3 = c2 + (__builtin_amdgcn_readfirstlane(kcnt[2 * B + b]) + PS_SK - 1) / PS_SK;
  const int NT = c3 + (__builtin_amdgcn_readfirstlane(kcnt[3 * B + b]) + PS_SK - 1) / PS_SK;

  // ---- loader role of this lane.  One DMA instruction moves 1 KiB of an image; the feature a lane reads is picked from wave-uniform
  // list entries (scalar registers) by masks, never by a branch.
  //   A: instruction id = AI wave + i carries features k = 2 id + (lane >> 5), part (lane >> 4) & 1, chunk position lane & 15
  //   B: instruction id = BI wave + j carries rows k = RB id + lane / (64 / RB), chunk position lane % (64 / RB) of [hi | lo]
  const int dch = lane & 15;
  const int hsel = -(lane >> 5);                         // all ones in the upper half of the wave
  const int drow = lane >> 4;
  const int m1 = -(int)(drow == 1), m2 = -(int)(drow == 2), m3 = -(int)(drow == 3);
  unsigned a_src[C::AI], b_src[C::BI];
#pragma unroll
  for (int i = 0; i < C::AI; ++i) {
    const int k = 2 * (C::AI * wave + i) + (lane >> 5);
    const int ch = dch ^ ps_swz(k);                       // the logical chunk that lands at this lane's position
    a_src[i] = (unsigned)((drow & 1) * 2 * ldt + (r0 + 8 * ch) * 2);
  }
#pragma unroll
  for (int j = 0; j < C::BI; ++j) {
    if constexpr (WC == 2) {     // 256-byte rows [hi 64 u | lo 64 u]: the swizzle runs over the 16 chunks of the row
      const int k = 4 * (C::BI * wave + j) + drow;
      const int ch = dch ^ ps_swz(k);
      b_src[j] = (unsigned)((ch >> 3) * 2 * Hp + (u0 + 8 * (ch & 7)) * 2);
    } else {                     // 512-byte rows [hi 128 u | lo 128 u]: over the 16 chunks of each half
      const int k = 2 * (C::BI * wave + j) + (lane >> 5);
      const int ch = dch ^ ps_swz(k);
      b_src[j] = (unsigned)((drow & 1) * 2 * Hp + (u0 + 8 * ch) * 2);
    }
  }
  struct Idx {
    int e[C::NI];
  };
  auto stage_gate = [&](int n) { return (n >= c1) + (n >= c2) + (n >= c3); };
  auto load_idx = [&](Idx& I, int n) {     // (n clamped: a stage beyond the last re-issues the last one into a free buffer)
    n = n < NT ? n : NT - 1;
    const int g = stage_gate(n);
    const int first = n >= c3 ? c3 : n >= c2 ? c2 : n >= c1 ? c1 : 0;
    const int* lp = lists + ((size_t)g * B + b) * Fp32 + (n - first) * PS_SK + C::NI * wave;
#pragma unroll
    for (int r = 0; r < C::NI; ++r) I.e[r] = lp[r];   // raw: unpacked in issue(), an iteration later (no wait for the scalar load here)
  };
  auto issue = [&](const Idx& I, int n) {
    const unsigned buf = lds0 + (unsigned)(n % C::NBUF) * C::STAGE;
    const int wbase = stage_gate(n < NT ? n : NT - 1) * (F + 1);
#pragma unroll
    for (int i = 0; i < C::AI; ++i) {
      const int f0 = I.e[2 * i] & 0xFFFF, f1 = I.e[2 * i + 1] & 0xFFFF;
      const int f = f0 + ((f1 - f0) & hsel);
      ps_dma_b128(XSb, (unsigned)f * (unsigned)(4 * ldt) + a_src[i], buf + (unsigned)((C::AI * wave + i) * 1024));
    }
#pragma unroll
    for (int j = 0; j < C::BI; ++j) {
      int wrow;
      if constexpr (WC == 2) {
        const int w0 = I.e[4 * j] >> 16;
        wrow = w0 + (((I.e[4 * j + 1] >> 16) - w0) & m1) + (((I.e[4 * j + 2] >> 16) - w0) & m2) + (((I.e[4 * j + 3] >> 16) - w0) & m3);
      } else {
        const int w0 = I.e[2 * j] >> 16;
        wrow = w0 + (((I.e[2 * j + 1] >> 16) - w0) & hsel);
      }
      ps_dma_b128(WS, (unsigned)(wbase + wrow) * (unsigned)(4 * Hp) + b_src[j], buf + (unsigned)(C::A_BYTES + (C::BI * wave + j) * 1024));
    }
  };

  // ---- matrix role: operand lane (column = lane & 31 of its block, k half = lane >> 5); a transposed read serves 16 lanes: lane
  // 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3 of a 4 x 16 block and receives its column
  const int kh = lane >> 5, mhalf = (lane >> 4) & 1, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
  unsigned offA[2][2], offB[2][2];   // [read j][row block / plane]: byte offsets inside one k-step (16 rows) of the image
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * kh + 4 * j + q;                    // (+ 16 per k-step: the swizzle does not see it)
    const int sw = ps_swz(row);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) offA[j][mb] = (unsigned)(512 * row + 16 * ((wr * 8 + mb * 4 + 2 * mhalf + (p >> 1)) ^ sw) + 8 * (p & 1));
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      offB[j][pl] = WC == 2 ? (unsigned)(256 * row + 16 * ((pl * 8 + wc * 4 + 2 * mhalf + (p >> 1)) ^ sw) + 8 * (p & 1))
                            : (unsigned)(512 * row + 256 * pl + 16 * ((wc * 4 + 2 * mhalf + (p >> 1)) ^ sw) + 8 * (p & 1));
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][mb][e] = 0.f;

  if (NT > 0) {
    constexpr int PD = C::NBUF - 1;   // stages in flight
    Idx I0;
#pragma unroll
    for (int s0 = 0; s0 < PD; ++s0) {
      load_idx(I0, s0);
      issue(I0, s0);
    }
    load_idx(I0, PD);
    int n = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nst = g == 0 ? c1 : g == 1 ? c2 - c1 : g == 2 ? c3 - c2 : NT - c3;
      for (int ls = 0; ls < nst; ++ls, ++n) {
        // this wave's DMAs of stage n have landed (stages n .. n + PD - 1 are in flight); the barrier makes that true of every wave's,
        // and says everybody is done reading stage n - 1, whose buffer the DMAs of stage n + PD overwrite
        if constexpr (WC == 2)
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // 6 per stage x (PD - 1)
        else
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // 4 per stage x (PD - 1)
        __syncthreads();
        issue(I0, n + PD);
        load_idx(I0, n + PD + 1);
        // Fragment reads as inline asm: hipcc does not fold the constant part of an LDS address into the instruction's offset field here
        // (it kept 24 address registers and re-added the ring position to each: ~120 vector instructions per stage beside 12 MFMAs).
        // The ring position is added ONCE per base register (8 adds); k-step and plane are immediates.  The waits are explicit and
        // carry the fragments as operands, so that no MFMA can be scheduled in front of the wait that covers its operands.
        const unsigned sb = lds0 + (unsigned)(n % C::NBUF) * C::STAGE;
        unsigned cA[2][2], cB[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x2 = 0; x2 < 2; ++x2) {
            cA[j][x2] = sb + offA[j][x2];
            cB[j][x2] = sb + C::A_BYTES + offB[j][x2];
          }
        // ONE asm statement per k-step: its 12 reads AND the wait for them.  The destination of an LDS read is written when the data
        // arrive, not when the instruction issues: a destination that the compiler can see before the wait may be COPIED by it while
        // the read is in flight (round 5 found exactly that in k_dw_split: a v_mov_b64 of a fragment in front of the s_waitcnt, 3 NaN
        // rows in one run of ten when two processes shared the GPU).  Inside one statement nothing can come between; the next k-step's
        // reads still fly under this k-step's MFMAs, which are queued in the matrix pipe by then.
#define PS_KSTEP(FA, FB, AOFF0, AOFF1, BOFF)                                                                                          \
  asm volatile("ds_read_b64_tr_b16 %0, %12 offset:" #AOFF0 "\n\tds_read_b64_tr_b16 %1, %13 offset:" #AOFF0                             \
               "\n\tds_read_b64_tr_b16 %8, %16 offset:" #BOFF "\n\tds_read_b64_tr_b16 %9, %17 offset:" #BOFF                            \
               "\n\tds_read_b64_tr_b16 %4, %14 offset:" #AOFF0 "\n\tds_read_b64_tr_b16 %5, %15 offset:" #AOFF0                          \
               "\n\tds_read_b64_tr_b16 %2, %12 offset:" #AOFF1 "\n\tds_read_b64_tr_b16 %3, %13 offset:" #AOFF1                          \
               "\n\tds_read_b64_tr_b16 %6, %14 offset:" #AOFF1 "\n\tds_read_b64_tr_b16 %7, %15 offset:" #AOFF1                          \
               "\n\tds_read_b64_tr_b16 %10, %18 offset:" #BOFF "\n\tds_read_b64_tr_b16 %11, %19 offset:" #BOFF                          \
               "\n\ts_waitcnt lgkmcnt(0)"                                                                                              \
               : "=&v"(FA[0][0][0]), "=&v"(FA[0][0][1]), "=&v"(FA[0][1][0]), "=&v"(FA[0][1][1]), "=&v"(FA[1][0][0]), "=&v"(FA[1][0][1]), \
                 "=&v"(FA[1][1][0]), "=&v"(FA[1][1][1]), "=&v"(FB[0][0]), "=&v"(FB[0][1]), "=&v"(FB[1][0]), "=&v"(FB[1][1])               \
               : "v"(cA[0][0]), "v"(cA[1][0]), "v"(cA[0][1]), "v"(cA[1][1]), "v"(cB[0][0]), "v"(cB[1][0]), "v"(cB[0][1]), "v"(cB[1][1])   \
               : "memory")
        auto frag = [](tr4 x0, tr4 x1) {
          const tr8 v = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
          return __builtin_bit_cast(f16x8, v);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          tr4 fa[2][2][2], fb[2][2];   // A: [row block][plane][read], B: [plane][read]
          if (ks == 0) {
            PS_KSTEP(fa, fb, 0, 256, 0);
          } else if constexpr (WC == 2) {
            PS_KSTEP(fa, fb, 8192, 8448, 4096);
          } else {
            PS_KSTEP(fa, fb, 8192, 8448, 8192);
          }
          const f16x8 bh = frag(fb[0][0], fb[0][1]), bl = frag(fb[1][0], fb[1][1]);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const f16x8 ah = frag(fa[mb][0][0], fa[mb][0][1]), al = frag(fa[mb][1][0], fa[mb][1][1]);
            acc[g][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[g][mb], 0, 0, 0);
            acc[g][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[g][mb], 0, 0, 0);
            acc[g][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[g][mb], 0, 0, 0);
          }
        }
#undef PS_KSTEP
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the re-issued last stages: nothing of this workgroup's LDS is in flight at exit)
  }
  const float sw = ps_wscale(wmax);
  const float cf = __uint_as_float(*cword);
  const float inv = sw > 0.f ? cf / (sw * XS_SCALE) : __uint_as_float(0x7FC00000u);
  const int l31 = lane & 31, lh = lane >> 5;
  const int unit = u0 + wc * 32 + l31;
  if (unit < H) {
    const float4 bias = *reinterpret_cast<const float4*>(bp + unit * 4);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * 64 + mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (row < T)
          *reinterpret_cast<float4*>(Z + ((size_t)b * T + row) * N + unit * 4) =
              make_float4(fmaf(acc[0][mb][reg], inv, bias.x), fmaf(acc[1][mb][reg], inv, bias.y), fmaf(acc[2][mb][reg], inv, bias.z),
                          fmaf(acc[3][mb][reg], inv, bias.w));
      }
  }
}

// ------------------------------------------------------------------------------------------------ tn on pre-split operands
// Dropout-aware dW: P[(g, b)][q][u] = c sum_t X[b, t, f_q] dZ[b, t, 4u + g] for the kept features f_q of (gate, sample) - K is TIME, so
// both operands are read along their rows: A = rows of the split activation copy XS (gathered by the kept list: fixed for the whole K
// loop, the per-lane source offsets are computed once), B = rows of dZS, the split transposed gate gradients.  dZS[b][4u + g] is a split
// row of dZ sz(row) with sz the power of two that puts the ROW's largest |dZ| in [2^14, 2^15) (k_rowmax_bt finds it, k_transpose_split
// applies it): a row's scale leaves the sum over time exactly and is divided out of its output column, so the gradient's dynamic range
// across units, samples and gates costs nothing (gemm.hip, k_gemm_tn_sparse16).
// Tile 128 kept features x 128 units, stages of 32 time steps: per operand and part a [128 rows][64 B] image, 16-byte chunk c of row r at
// position c ^ ((r >> 2) & 3) (ds_read_b128 of 32 consecutive rows at one chunk: conflict-free); 32 KiB per stage.
constexpr int DW_BM = 128, DW_BN = 128, DW_TK = 32, DW_STAGE = 4 * 128 * 64;

// zmax[b][col] = largest |dZ[b, t, col]| over t, as float bits; zsum[b][col] = sum of dZ[b, t, col] over t (either may be null)
__global__ __launch_bounds__(256) void k_rowmax_bt(const float* __restrict__ dZ, int N, int T, unsigned* __restrict__ zmax, float* __restrict__ zsum) {
  __shared__ float part[2][4][64];
  const int b = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float m = 0.f, sm = 0.f;
  if (col < N) {
    const float* p = dZ + (size_t)b * T * N + col;
    for (int t = w; t < T; t += 4) {
      const float v = p[(size_t)t * N];
      m = fmaxf(m, fabsf(v));
      sm += v;
    }
  }
  part[0][w][threadIdx.x & 63] = m;
  part[1][w][threadIdx.x & 63] = sm;
  __syncthreads();
  if (w == 0 && col < N) {
    if (zmax) zmax[(size_t)b * N + col] = __float_as_uint(fmaxf(fmaxf(part[0][0][threadIdx.x], part[0][1][threadIdx.x]), fmaxf(part[0][2][threadIdx.x], part[0][3][threadIdx.x])));
    if (zsum) zsum[(size_t)b * N + col] = (part[1][0][threadIdx.x] + part[1][1][threadIdx.x]) + (part[1][2][threadIdx.x] + part[1][3][threadIdx.x]);
  }
}
__device__ __forceinline__ float dw_zscale(unsigned zm_bits) {   // the power of two that puts the row's largest |dZ| in [2^14, 2^15)
  const float zm = __uint_as_float(zm_bits);
  int ex = 0;
  if (zm > 0.f && zm < 3.0e38f) (void)frexpf(zm, &ex);
  ex = ex < -100 ? -100 : ex;
  return zm < 3.0e38f ? ldexpf(1.f, 15 - ex) : __uint_as_float(0x7FC00000u);   // (an Inf / NaN gradient stays visible)
}
// dZS[b][col] = split row of dZ[b][0..T)[col] * sz(b, col), zero for t >= T
__global__ __launch_bounds__(256) void k_transpose_split_scaled(const float* __restrict__ dZ, int N, float* __restrict__ dZS, int ldt, int T,
                                                                const unsigned* __restrict__ zmax) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* Zb = dZ + (size_t)b * T * N;
  float* Sb = dZS + (size_t)b * N * ldt;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = t0 + ty + 4 * i, cc = c0 + tx;
    tile[ty + 4 * i][tx] = (t < T && cc < N) ? Zb[(size_t)t * N + cc] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int cc = c0 + ty + 4 * i, t = t0 + tx;
    if (cc < N && t < ldt) {
      const float sz = dw_zscale(zmax[(size_t)b * N + cc]);
      _Float16 hi, lo;
      ps_split(tile[tx][ty + 4 * i] * sz, hi, lo);
      _Float16* row = reinterpret_cast<_Float16*>(Sb + (size_t)cc * ldt);
      row[t] = hi;
      row[ldt + t] = lo;
    }
  }
}

typedef float dw_f4 __attribute__((ext_vector_type(4)));

// NW = 8 waves (wave = 32 features x 64 units), four-stage ring (128 KiB: a CU of its own), or NW = 4 waves (wave = 64 x 64), three-stage
// ring (96 KiB, < 160 registers: fits on a CU beside ONE workgroup of a persistent scan - in the training step this kernel runs while
// the next batch's encoder scans hold the chip, where the 8-wave form cannot be placed at all until they end)
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_dw_split(const char* __restrict__ XS, int ldt, const int* __restrict__ lists,
                                                         const int* __restrict__ kcnt, const unsigned* __restrict__ cword,
                                                         const char* __restrict__ dZS, const unsigned* __restrict__ zmax, float* __restrict__ P,
                                                         int B, int T, int Fp32, int F, int H) {
  constexpr int MB = 8 / NW;                  // 32-row blocks of A per wave
  constexpr int NBUF = NW == 8 ? 4 : 3;
  constexpr int DPW = 16 / NW;                // DMA instructions per wave, stage and operand
  extern __shared__ __attribute__((aligned(16))) char dw_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
  const int N = 4 * H;
  // XCD-aware decode (gemm.hip, k_gemm_tn_sparse): all workgroups of a sample - which share its X rows and dZ rows - in one L2
  const int nft = (Fp32 + DW_BM - 1) / DW_BM, nut = (H + DW_BN - 1) / DW_BN, wps = 4 * nft * nut;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int b = (jj / wps) * 8 + xcd;
  if (b >= B) return;
  const int w_ = jj % wps, g = w_ / (nft * nut), gb = g * B + b;
  const int q0 = ((w_ / nut) % nft) * DW_BM, u0 = (w_ % nut) * DW_BN;
  const int cnt = kcnt[gb];
  if (q0 >= cnt) return;   // (uniform) no kept feature in this row tile
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)dw_smem;
  const char* XSb = XS + (size_t)b * F * ldt * 4;
  const char* ZSb = dZS + (size_t)b * N * ldt * 4;

  // ---- loader role: instruction id = DPW wave + i of each operand: part = id >> 3, rows 16 (id & 7) + (lane >> 2), chunk position
  // lane & 3.  The rows of a tile do not change over the K loop: the per-lane source offsets are computed once.
  unsigned a_off[DPW], b_off[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int id = DPW * wave + i, part = id >> 3, r = 16 * (id & 7) + (lane >> 2);
    const int ch = (lane & 3) ^ ((r >> 2) & 3);            // the logical chunk that lands at this lane's position
    int q = q0 + r;
    q = q < cnt ? q : cnt - 1;                             // (rows beyond the list are computed from a valid row and never stored)
    const int f = lists[(size_t)gb * Fp32 + q] & 0xFFFF;
    a_off[i] = (unsigned)f * (unsigned)(4 * ldt) + (unsigned)(part * 2 * ldt + ch * 16);
    int u = u0 + r;
    u = u < H ? u : H - 1;
    b_off[i] = (unsigned)(4 * u + g) * (unsigned)(4 * ldt) + (unsigned)(part * 2 * ldt + ch * 16);
  }
  const int nst = (T + DW_TK - 1) / DW_TK;     // (the padded rows are zero behind T; ldt >= nst * 32)
  auto issue = [&](int n) {                    // (a stage beyond the last re-reads the last one into a free buffer)
    const unsigned buf = lds0 + (unsigned)(n % NBUF) * DW_STAGE;
    const int nn = (n < nst ? n : nst - 1) * DW_TK * 2;   // byte offset of the stage's first time step in a part
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      ps_dma_b128(XSb + nn, a_off[i], buf + (unsigned)((DPW * wave + i) * 1024));
      ps_dma_b128(ZSb + nn, b_off[i], buf + (unsigned)(16384 + (DPW * wave + i) * 1024));
    }
  };
  // ---- matrix role
  const int l31 = lane & 31, kh = lane >> 5;
  unsigned oA[MB][2], oB[2][2];   // [row block][k-step] / [column block][k-step]: byte offsets of this lane's 16-byte operand in the hi image
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = 32 * MB * wr + 32 * mb + l31;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) oA[mb][ks] = (unsigned)(m * 64 + 16 * ((2 * ks + kh) ^ ((m >> 2) & 3)));
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int nn = 64 * wc + 32 * nb + l31;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) oB[nb][ks] = (unsigned)(16384 + nn * 64 + 16 * ((2 * ks + kh) ^ ((nn >> 2) & 3)));
  }
  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;
  constexpr int PD = NBUF - 1;
#pragma unroll
  for (int s0 = 0; s0 < PD; ++s0) issue(s0);
  for (int n = 0; n < nst; ++n) {
    // 2 DPW DMAs per stage and wave, PD - 1 stages may stay in flight: this wave's DMAs of stage n have landed
    if constexpr (NW == 8)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // (4 waves: 8 per stage x 1)
    __syncthreads();
    issue(n + PD);
    const unsigned sb = lds0 + (unsigned)(n % NBUF) * DW_STAGE;
    // ONE asm statement per k-step: its reads AND the wait for them (see k_proj_split: a fragment the compiler can see before the
    // wait may be copied while the read is in flight)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      dw_f4 fa[MB][2], fb[2][2];   // [block][part]
      const unsigned ca0 = sb + oA[0][ks], ca1 = sb + oA[MB - 1][ks], cb0 = sb + oB[0][ks], cb1 = sb + oB[1][ks];
      if constexpr (MB == 1)
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %2, %7\n\tds_read_b128 %4, %8\n\tds_read_b128 %1, %6 offset:8192"
                     "\n\tds_read_b128 %3, %7 offset:8192\n\tds_read_b128 %5, %8 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(fa[0][0]), "=&v"(fa[0][1]), "=&v"(fb[0][0]), "=&v"(fb[0][1]), "=&v"(fb[1][0]), "=&v"(fb[1][1])
                     : "v"(ca0), "v"(cb0), "v"(cb1)
                     : "memory");
      else
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %4, %10\n\tds_read_b128 %6, %11\n\tds_read_b128 %2, %9"
                     "\n\tds_read_b128 %1, %8 offset:8192\n\tds_read_b128 %5, %10 offset:8192\n\tds_read_b128 %7, %11 offset:8192"
                     "\n\tds_read_b128 %3, %9 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(fa[0][0]), "=&v"(fa[0][1]), "=&v"(fa[MB - 1][0]), "=&v"(fa[MB - 1][1]), "=&v"(fb[0][0]), "=&v"(fb[0][1]),
                       "=&v"(fb[1][0]), "=&v"(fb[1][1])
                     : "v"(ca0), "v"(ca1), "v"(cb0), "v"(cb1)
                     : "memory");
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const f16x8 ah = __builtin_bit_cast(f16x8, fa[mb][0]), al = __builtin_bit_cast(f16x8, fa[mb][1]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const f16x8 bh = __builtin_bit_cast(f16x8, fb[nb][0]), bl = __builtin_bit_cast(f16x8, fb[nb][1]);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mb][nb], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float cf = __uint_as_float(*cword) * (1.f / XS_SCALE);
  float* out = P + (size_t)gb * Fp32 * H;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int u = u0 + 64 * wc + 32 * nb + l31;
    // (the reciprocals apart from each other: sz reaches 2^115 for a row of tiny gradients; all powers of two, the products are exact)
    const float cz = u < H ? 1.f / dw_zscale(zmax[(size_t)b * N + 4 * u + g]) : 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = 32 * MB * wr + 32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        if (q0 + r < cnt && u < H) out[(size_t)(q0 + r) * H + u] = acc[mb][nb][reg] * cf * cz;
      }
  }
}

// dWp[f][4u+g] = sum over the samples that kept feature f for gate g, in sample order
__global__ __launch_bounds__(256) void k_dw_gather32(const float* __restrict__ P, const int* __restrict__ kpos, float* __restrict__ dWp, int B, int F,
                                                     int Fp32, int H) {
  const size_t n = (size_t)4 * F * H;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int u = (int)(i % H);
    const int f = (int)((i / H) % F);
    const int g = (int)(i / ((size_t)H * F));
    float sacc = 0.f;
    for (int b = 0; b < B; ++b) {
      const int pos = kpos[((size_t)g * B + b) * F + f];
      if (pos >= 0) sacc += P[(((size_t)g * B + b) * Fp32 + pos) * H + u];
    }
    dWp[(size_t)f * 4 * H + 4 * u + g] = sacc;
  }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static int hp_of(int H) { return (H + PS_HP - 1) / PS_HP * PS_HP; }
static int fp32_of(int F) { return (F + PS_SK - 1) / PS_SK * PS_SK; }

}  // namespace

extern "C" {

// workspace of the projection: kept lists [4B][Fp32] | counts [4B] | words | weight planes | list positions [4B][F] (the last for
// mgr_lstm_param_grads_dropout_ts, which may take its lists from the projection of the same mask instead of building them again)
static size_t proj_ts_planes_bytes(int F, int H) { return mgr_align_up((size_t)4 * (F + 1) * 2 * hp_of(H) * sizeof(_Float16), 256); }
size_t mgr_lstm_input_proj_dropout_ts_ws_bytes(int B, int F, int H) {
  return mgr_align_up((size_t)4 * B * fp32_of(F) * sizeof(int), 256) + mgr_align_up((size_t)4 * B * sizeof(int), 256) + 256 +
         proj_ts_planes_bytes(F, H) + mgr_align_up((size_t)4 * B * F * sizeof(int), 256);
}

int mgr_lstm_input_proj_dropout_ts(mgr_ctx* c, const float* XS, int ldt, const float* mask4, float drop_rate, const float* Wp,
                                   const float* bp, float* Z, int B, int T, int F, int H, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && XS && Wp && bp && Z, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && H > 0 && F >= 16 && F <= 2048, "bad shape (16 <= F <= 2048)");
  MGR_REQUIRE(ldt % PS_TM == 0 && ldt >= T, "the split copy must be padded to whole row tiles of %d (ldt %d, T %d)", PS_TM, ldt, T);
  MGR_REQUIRE(aligned16(XS) && aligned16(bp) && aligned16(Z) && aligned16(Wp), "XS / bp / Z / Wp must be 16-byte aligned");
  MGR_REQUIRE((size_t)F * ldt * 4 < (1ull << 32), "sample block too large");
  MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_input_proj_dropout_ts_ws_bytes(B, F, H), "workspace too small");
  (void)drop_rate;
  const int Fp32 = fp32_of(F), Hp = hp_of(H);
  char* w = reinterpret_cast<char*>(ws);
  int* lists = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * Fp32 * sizeof(int), 256);
  int* kcnt = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * sizeof(int), 256);
  unsigned* words = reinterpret_cast<unsigned*>(w);   // [0] largest |W|, [1] the mask factor
  w += 256;
  _Float16* WSp = reinterpret_cast<_Float16*>(w);
  w += proj_ts_planes_bytes(F, H);
  int* kpos = reinterpret_cast<int*>(w);
  hipStream_t s = mgr_stream(c);
  if (!(c->attr_done & 16u)) {
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_split<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PsCfg<2>::LDS));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_split<4>), hipFuncAttributeMaxDynamicSharedMemorySize, PsCfg<4>::LDS));
    c->attr_done |= 16u;
  }
  mgr_prof_begin(c, MGR_K_GEMM_NN);
  // Frozen weights (mgr_weight_planes_cache): the planes this workspace holds from an earlier call are still those of Wp - the largest
  // |W| (words[0]) and the (hi, lo) planes are not rebuilt, only the mask factor word is reset (4 of the 6 conversions of a config-F step)
  bool frozen = false, cached = false;
  int free_slot = -1;
  for (int i = 0; i < MGR_MAX_FROZEN; ++i) frozen = frozen || (c->frozen_w[i] == Wp);
  if (frozen) {
    for (int i = 0; i < MGR_MAX_FROZEN; ++i) {
      const mgr_ctx::PlaneEntry& e = c->planes[i];
      if (e.Wp == Wp && e.ws == ws && e.F == F && e.H == H) cached = true;
      if (!e.Wp && free_slot < 0) free_slot = i;
    }
  }
  MGR_HIP(hipMemsetAsync(words + (cached ? 1 : 0), 0, (cached ? 1 : 2) * sizeof(unsigned), s));
  hipLaunchKernelGGL(k_lists32, dim3(4 * B), dim3(64), 0, s, mask4, F, Fp32, lists, kcnt, kpos, words + 1);
  if (!cached) {
    const size_t n4 = (size_t)F * H;
    hipLaunchKernelGGL(k_wmax, dim3((int)((n4 + 255) / 256 < 256 ? (n4 + 255) / 256 : 256)), dim3(256), 0, s, Wp, n4, words);
    const size_t n = (size_t)(F + 1) * Hp;
    hipLaunchKernelGGL(k_wplanes, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, Wp, WSp, F, H, Hp, words);
    if (frozen) {   // (a full table forgets its oldest entry: forgetting is always safe)
      if (free_slot < 0) free_slot = (int)(c->planes_evict++ % MGR_MAX_FROZEN);
      c->planes[free_slot] = mgr_ctx::PlaneEntry{Wp, ws, F, H};
    }
  }
  // 128-unit tiles (8 waves, one workgroup per CU) where they waste little of their width; tune key 12: 1 = always 64, 2 = always 128
  const int waste128 = (H + 127) / 128 * 128 - H, waste64 = (H + 63) / 64 * 64 - H;
  const bool wide = c->tune[12] == 2 || (c->tune[12] == 0 && waste128 - waste64 <= H / 8);
  const int tu = wide ? 128 : 64;
  const int ntiles = ((H + tu - 1) / tu) * ((((T + PS_TM - 1) / PS_TM) * B + 7) / 8) * 8;
  if (wide)
    hipLaunchKernelGGL(k_proj_split<4>, dim3(ntiles), dim3(512), PsCfg<4>::LDS, s, reinterpret_cast<const char*>(XS), ldt, lists, kcnt, words + 1,
                       reinterpret_cast<const char*>(WSp), Hp, words, bp, Z, B, T, Fp32, F, H);
  else
    hipLaunchKernelGGL(k_proj_split<2>, dim3(ntiles), dim3(256), PsCfg<2>::LDS, s, reinterpret_cast<const char*>(XS), ldt, lists, kcnt, words + 1,
                       reinterpret_cast<const char*>(WSp), Hp, words, bp, Z, B, T, Fp32, F, H);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_NN);
  return 0;
}

// lists | counts | words | list positions | partial tiles of a dW-shaped product with F input rows
static size_t dw_ts_lists_and_tiles(int B, int F, int H) {
  const size_t Fp32 = (size_t)fp32_of(F);
  return mgr_align_up((size_t)4 * B * Fp32 * sizeof(int), 256) + mgr_align_up((size_t)4 * B * sizeof(int), 256) + 256 +
         mgr_align_up((size_t)4 * B * F * sizeof(int), 256) + mgr_align_up((size_t)4 * B * Fp32 * H * sizeof(float), 256);
}
static size_t dw_ts_extra(int B, int F, int H, int ldt) {
  return dw_ts_lists_and_tiles(B, F, H) + mgr_align_up((size_t)B * 4 * H * ldt * sizeof(float), 256) +
         mgr_align_up((size_t)B * 4 * H * sizeof(unsigned), 256) + dw_ts_lists_and_tiles(B, H, H);   // (the last: dU from HsT, F = H rows)
}

size_t mgr_lstm_param_grads_dropout_ts_ws_bytes(int B, int T, int F, int H, int ldt) {
  return mgr_lstm_param_grads_ws_bytes(B, T, F, H) + dw_ts_extra(B, F, H, ldt);
}

int mgr_lstm_param_grads_dropout_ts(mgr_ctx* c, const float* XS, int ldt, const float* mask4, float drop_rate, const float* Hs, int ldh,
                                    const float* dZ, float* dWp, float* dUp, float* dbp, int B, int T, int F, int H, int reverse,
                                    void* ws, size_t ws_bytes, const unsigned* dzmax, const float* dbsum, const void* proj_ws,
                                    const float* HsT) {
  MGR_REQUIRE(c && XS && mask4 && Hs && dZ && dWp && dUp && dbp, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && H > 0 && F >= 16 && F <= 2048 && ldh >= H, "bad shape (16 <= F <= 2048)");
  MGR_REQUIRE(ldt % 32 == 0 && ldt >= (T + DW_TK - 1) / DW_TK * DW_TK, "the split copy must be padded to whole stages of %d time steps (ldt %d, T %d)", DW_TK, ldt, T);
  MGR_REQUIRE(aligned16(dZ) && aligned16(XS), "dZ / XS must be 16-byte aligned");
  MGR_REQUIRE((size_t)F * ldt * 4 < (1ull << 32) && (size_t)4 * H * ldt * 4 < (1ull << 32), "sample block too large");
  MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_param_grads_dropout_ts_ws_bytes(B, T, F, H, ldt), "workspace too small");
  (void)drop_rate;
  mgr_prof_begin(c, MGR_K_GEMM_TN);
  // dU / db first: they are short, and in the training step the long dW kernel then ends this direction's work (gemm.hip)
  MGR_REQUIRE(!HsT || (H >= 16 && aligned16(HsT) && (size_t)H * ldt * 4 < (1ull << 32)), "HsT: 16 <= H, 16-byte aligned");
  int r = mgr_param_grads_du_db(c, Hs, ldh, dZ, HsT ? nullptr : dUp, dbp, B, T, F, H, reverse, ws, dbsum);
  if (r) return r;
  const int Fp32 = fp32_of(F), N = 4 * H;
  char* w = reinterpret_cast<char*>(ws) + mgr_lstm_param_grads_ws_bytes(B, T, F, H);
  int* lists = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * Fp32 * sizeof(int), 256);
  int* kcnt = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * sizeof(int), 256);
  unsigned* words = reinterpret_cast<unsigned*>(w);   // [1] the mask factor
  w += 256;
  int* kpos = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * F * sizeof(int), 256);
  float* P = reinterpret_cast<float*>(w);
  w += mgr_align_up((size_t)4 * B * Fp32 * H * sizeof(float), 256);
  float* dZS = reinterpret_cast<float*>(w);
  w += mgr_align_up((size_t)B * N * ldt * sizeof(float), 256);
  const unsigned* zmax = dzmax ? dzmax : reinterpret_cast<unsigned*>(w);   // (the BPTT's own row maxima, or found here)
  w += mgr_align_up((size_t)B * N * sizeof(unsigned), 256);
  // dU from HsT: the same product with the H rows h_prev in place of the kept input features (every row kept, factor 1)
  const int Hp32 = fp32_of(H);
  int* lists2 = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * Hp32 * sizeof(int), 256);
  int* kcnt2 = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * sizeof(int), 256);
  unsigned* words2 = reinterpret_cast<unsigned*>(w);
  w += 256;
  int* kpos2 = reinterpret_cast<int*>(w);
  w += mgr_align_up((size_t)4 * B * H * sizeof(int), 256);
  float* P2 = reinterpret_cast<float*>(w);
  hipStream_t s = mgr_stream(c);
  if (!(c->attr_done & 32u)) {
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dw_split<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DW_STAGE));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dw_split<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * DW_STAGE));
    c->attr_done |= 32u;
  }
  if (proj_ws) {
    // the lists, counts, list positions and the mask factor the projection of the SAME mask left in its workspace (the layout of
    // mgr_lstm_input_proj_dropout_ts above): nothing to build
    char* pw = reinterpret_cast<char*>(const_cast<void*>(proj_ws));
    lists = reinterpret_cast<int*>(pw);
    pw += mgr_align_up((size_t)4 * B * Fp32 * sizeof(int), 256);
    kcnt = reinterpret_cast<int*>(pw);
    pw += mgr_align_up((size_t)4 * B * sizeof(int), 256);
    words = reinterpret_cast<unsigned*>(pw);
    pw += 256 + proj_ts_planes_bytes(F, H);
    kpos = reinterpret_cast<int*>(pw);
  } else {
    MGR_HIP(hipMemsetAsync(words, 0, 2 * sizeof(unsigned), s));
    hipLaunchKernelGGL(k_lists32, dim3(4 * B), dim3(64), 0, s, mask4, F, Fp32, lists, kcnt, kpos, words + 1);
  }
  if (!dzmax) hipLaunchKernelGGL(k_rowmax_bt, dim3((N + 63) / 64, B), dim3(256), 0, s, dZ, N, T, const_cast<unsigned*>(zmax), (float*)nullptr);
  hipLaunchKernelGGL(k_transpose_split_scaled, dim3((ldt + 63) / 64, (N + 63) / 64, B), dim3(256), 0, s, dZ, N, dZS, ldt, T, zmax);
  if (HsT) {
    const int grid2 = 8 * ((B + 7) / 8) * 4 * ((Hp32 + DW_BM - 1) / DW_BM) * ((H + DW_BN - 1) / DW_BN);
    MGR_HIP(hipMemsetAsync(words2, 0, 2 * sizeof(unsigned), s));
    hipLaunchKernelGGL(k_lists32, dim3(4 * B), dim3(64), 0, s, (const float*)nullptr, H, Hp32, lists2, kcnt2, kpos2, words2 + 1);
    if (c->tune[12] == 1)
      hipLaunchKernelGGL(k_dw_split<4>, dim3(grid2), dim3(256), 3 * DW_STAGE, s, reinterpret_cast<const char*>(HsT), ldt, lists2, kcnt2, words2 + 1,
                         reinterpret_cast<const char*>(dZS), zmax, P2, B, T, Hp32, H, H);
    else
      hipLaunchKernelGGL(k_dw_split<8>, dim3(grid2), dim3(512), 4 * DW_STAGE, s, reinterpret_cast<const char*>(HsT), ldt, lists2, kcnt2, words2 + 1,
                         reinterpret_cast<const char*>(dZS), zmax, P2, B, T, Hp32, H, H);
    const size_t n2 = (size_t)4 * H * H;
    hipLaunchKernelGGL(k_dw_gather32, dim3((int)((n2 + 255) / 256 < 4096 ? (n2 + 255) / 256 : 4096)), dim3(256), 0, s, P2, kpos2, dUp, B, H, Hp32, H);
  }
  const int grid = 8 * ((B + 7) / 8) * 4 * ((Fp32 + DW_BM - 1) / DW_BM) * ((H + DW_BN - 1) / DW_BN);
  // tune key 12 (the tile switch of the projection): 1 = the 4-wave form, which fits on a CU beside a workgroup of a persistent scan
  if (c->tune[12] == 1)
    hipLaunchKernelGGL(k_dw_split<4>, dim3(grid), dim3(256), 3 * DW_STAGE, s, reinterpret_cast<const char*>(XS), ldt, lists, kcnt, words + 1,
                       reinterpret_cast<const char*>(dZS), zmax, P, B, T, Fp32, F, H);
  else
    hipLaunchKernelGGL(k_dw_split<8>, dim3(grid), dim3(512), 4 * DW_STAGE, s, reinterpret_cast<const char*>(XS), ldt, lists, kcnt, words + 1,
                       reinterpret_cast<const char*>(dZS), zmax, P, B, T, Fp32, F, H);
  const size_t n = (size_t)4 * F * H;
  hipLaunchKernelGGL(k_dw_gather32, dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, P, kpos, dWp, B, F, Fp32, H);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_TN);
  return 0;
}

int mgr_weight_planes_cache(mgr_ctx* c, const float* Wp, int frozen) {
  MGR_REQUIRE(c && Wp, "null argument");
  // whatever planes were kept for Wp are dropped: the call marks a point where the weights may have been rewritten
  for (int i = 0; i < MGR_MAX_FROZEN; ++i) {
    if (c->planes[i].Wp == Wp) c->planes[i] = mgr_ctx::PlaneEntry{nullptr, nullptr, 0, 0};
    if (c->frozen_w[i] == Wp) c->frozen_w[i] = nullptr;
  }
  if (!frozen) return 0;
  for (int i = 0; i < MGR_MAX_FROZEN; ++i)
    if (!c->frozen_w[i]) {
      c->frozen_w[i] = Wp;
      return 0;
    }
  // table full (engines that were never closed): the oldest promise is forgotten - its weights are simply rebuilt per call again
  const int victim = (int)(c->frozen_evict++ % MGR_MAX_FROZEN);
  for (int i = 0; i < MGR_MAX_FROZEN; ++i)
    if (c->planes[i].Wp == c->frozen_w[victim]) c->planes[i] = mgr_ctx::PlaneEntry{nullptr, nullptr, 0, 0};
  c->frozen_w[victim] = Wp;
  return 0;
}

int mgr_transpose_bt_split(mgr_ctx* c, const float* X, int ldx, float* XS, int ldt, int B, int T, int F) {
  MGR_REQUIRE(c && X && XS, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && ldx >= F && ldt >= T && ldt % 8 == 0, "bad shape");
  mgr_prof_begin(c, MGR_K_MISC);
  hipLaunchKernelGGL(k_transpose_split, dim3((ldt + 63) / 64, (F + 63) / 64, B), dim3(256), 0, mgr_stream(c), X, ldx, XS, ldt, T, F, 0LL, ldt, 0);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_MISC);
  return 0;
}

int mgr_transpose_bt_split_shift(mgr_ctx* c, const float* X, int ldx, float* XS, int ldt, int B, int T, int F, int tshift) {
  MGR_REQUIRE(c && X && XS, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && ldx >= F && ldt >= T && ldt % 8 == 0 && tshift >= -1 && tshift <= 1, "bad shape");
  mgr_prof_begin(c, MGR_K_MISC);
  hipLaunchKernelGGL(k_transpose_split, dim3((ldt + 63) / 64, (F + 63) / 64, B), dim3(256), 0, mgr_stream(c), X, ldx, XS, ldt, T, F, 0LL, ldt, tshift);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_MISC);
  return 0;
}

}  // extern "C"

int mgr_rowmax_bt(mgr_ctx* c, const float* dZ, int N, int T, int B, unsigned* zmax, float* zsum) {
  hipLaunchKernelGGL(k_rowmax_bt, dim3((N + 63) / 64, B), dim3(256), 0, mgr_stream(c), dZ, N, T, zmax, zsum);
  MGR_LAUNCH_CHECK();
  return 0;
}

int mgr_transpose_bt_split_strided(mgr_ctx* c, const float* X, int ldx, float* XS, int ldt, long long xsb, int ldt_fill, int B, int T, int F) {
  hipLaunchKernelGGL(k_transpose_split, dim3((ldt_fill + 63) / 64, (F + 63) / 64, B), dim3(256), 0, mgr_stream(c), X, ldx, XS, ldt, T, F, xsb, ldt_fill, 0);
  MGR_LAUNCH_CHECK();
  return 0;
}
