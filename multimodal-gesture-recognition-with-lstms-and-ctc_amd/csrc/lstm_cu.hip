// K3 / K7-scan for NARROW layers (H <= 128: the trainable fusion layer, H = 100): ONE workgroup = ONE WHOLE CU per
// (direction, 16-sample batch group), no inter-CU exchange at all.
//
// Why a third scan family.  The clusters of lstm_cluster.hip spread a narrow layer over 7 CUs per batch group and pay one
// cross-CU hand-off per time step; alone that is 1.5 (forward) / 2.2 us (BPTT) per step, but inside the pipelined training step
// the hand-off rides on memory queues that the encoder scans of the other stream keep busy: 3.2 / 4.3 us per step, 14 of the
// 29 ms of the step's critical chain (profiles/r04_single_cu_probes.txt).  The older single-CU kernels (lstm_cluster.hip with
// G = 1, lstm_mfma.hip) have no hand-off but run at 5-6 us per step: their MFMA waves also issue every vector-memory instruction
// (60-130 cycles of issue each on a SIMD with one wave) and wait for LDS round trips between short MFMA groups.
//
// This form: 8 waves, two per SIMD.
//   * waves 0-3 (one per SIMD) are MATRIX waves: U^T stationary in VGPRs, h_{t-1} as B operand from a double-buffered LDS image
//     read ONCE per step into registers, 6 whole tiles per wave in pairs of two interleaved accumulator chains plus a quarter of
//     K of the 25th tile; the cell update of a pair runs in the shadow of the next pair's MFMAs.  They touch LDS only.
//   * waves 4-7 are HELPER waves and own every vector-memory instruction: Z_t arrives by LDS-DMA two steps ahead (3-deep ring),
//     h / gates / c leave from an LDS staging area as the pairs complete (LDS flags, no barrier); helper 0 also finishes the
//     K-split 25th tile.  ONE s_barrier per time step.
//   * matrix-pipe floor: 625 MFMAs per step = 157 per SIMD x 32 cycles = 5.0 k cycles = 2.2 us; the workgroup asks for the whole
//     CU (141 KiB of LDS, 2 x 256 VGPRs per SIMD), so nothing of another stream shares its matrix pipe - the launch must find
//     free CUs: the engine enqueues it BEFORE the encoder scans of the next step become resident (mgr_stream_wait_last_resident).
#include "lstm_cluster.h"
#include <type_traits>

#include "lstm_common.h"

#define CU_STAMP 1
namespace {
#ifdef CU_STAMP
// diagnostic build: cycle stamps of one time step, per role, into LDS rows [32][64] behind the flags; dumped to hdr + 64 at the end
#define STAMP(role, idx)                                                                                     \
  if (step == 1000 && blockIdx.x == 0) {                                                                     \
    const unsigned long long tm_ = __builtin_amdgcn_s_memtime();                                             \
    *(volatile lds_u32_*)(reinterpret_cast<unsigned*>(smem + G::LDS_FLOATS) + ((role)*16 + (idx))) = (unsigned)tm_; \
  }
typedef __attribute__((address_space(3))) unsigned lds_u32_;
#else
#define STAMP(role, idx)
#endif

typedef __attribute__((address_space(3))) float lds_float;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned CU_SPIN_LIMIT = 1u << 22;

// LDS-DMA: one wave-instruction copies 64 x 16 B from global memory [gbase + voff] (gbase wave-uniform, voff per lane) to LDS
// [lds_addr + 16 * lane]; M0 carries the LDS address and is restored (hipcc does not know it was touched)
__device__ __forceinline__ void cu_dma_b128(const void* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void cu_dma_b32(const void* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");
}

// flag words in LDS (one writer, one reader): the address space must stay visible, or hipcc emits flat_ accesses (+ a vmcnt wait)
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ unsigned lds_flag_read(const unsigned* p) { return *(const volatile lds_u32*)p; }
__device__ __forceinline__ void lds_flag_write(unsigned* p, unsigned v) { *(volatile lds_u32*)p = v; }

struct CuFwdJob {
  const float* Z;
  const float* Up;
  float* Y;
  float* G;    // gates (B, T, H, 4) or null
  float* Cs;   // c (B, T, H) or null
  int ldy, B, T, reverse, nbg, wg_begin;
};
struct CuFwdLaunch {
  ClusterCommon cm;
  int njobs;
  CuFwdJob job[MGR_MAX_SCAN_JOBS];
};

// ---- geometry shared by the roles ---------------------------------------------------------------------------------------
template <int KS>
struct CuGeo {
  static constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256;
  static constexpr int FT = KS / 4;                 // whole tiles per matrix wave
  static constexpr int E = KS % 4;                  // tiles left over: K-split over the four matrix waves, finished by helper 0
  static_assert(E <= 1 && FT >= 1, "KS % 4 must be 0 or 1");
  // A matrix wave works through its tiles in GROUPS: pairs of tiles (two interleaved accumulator chains), the last two tiles as
  // singles.  The gate pre-activations of a group go to LDS and the partner helper wave runs the cell update (the vector
  // instructions of six cell updates per step do not fit between a wave's own MFMAs: ~110 each against ~4 issue slots per MFMA);
  // only the LAST tile's cell, which nothing could cover anyway, is the matrix wave's own.
  static constexpr int NG = FT == 1 ? 1 : (FT % 2 == 0 ? FT / 2 + 1 : (FT + 1) / 2);
  __host__ __device__ static constexpr int g_first(int g) { return (FT % 2 == 0 && FT >= 2 && g == NG - 1) ? FT - 1 : 2 * g; }
  __host__ __device__ static constexpr int g_size(int g) {
    return FT == 1 ? 1 : (FT % 2 == 0 ? (g >= NG - 2 ? 1 : 2) : (g == NG - 1 ? 1 : 2));
  }
  static constexpr int XB = KS / 4, XR = KS % 4;    // K split of the left-over tile: wave w takes XB + (w < XR) k-steps
  static constexpr int XS = XB + (XR ? 1 : 0);
  // LDS map (floats)
  static constexpr int OFF_IMG = 0;                              // [2][IMG]        h image, B-operand layout [q][kk][j][r]
  static constexpr int OFF_Z = OFF_IMG + 2 * IMG;                // [3][KS][64] x4  Z ring, tile-major, lane-linear
  static constexpr int OFF_PRE = OFF_Z + 3 * KS * 256;           // [4 waves][FT][64] x4  gate pre-activations of the tiles the helpers finish
  static constexpr int OFF_STG = OFF_PRE + 4 * FT * 256;         // [4 waves][2][384]: outputs of a wave's last tile { gates x4 | h | c }, by step parity
  static constexpr int OFF_PART = OFF_STG + 4 * 2 * 384;         // [4 waves][64] x4 partial sums of the left-over tile
  static constexpr int OFF_FLAG = OFF_PART + 4 * 256;            // [8][64]: groups done per matrix wave (0-3), partial written (4-7);
  static constexpr int LDS_FLOATS = OFF_FLAG + 8 * 64;           //          every lane writes its own word (no exec masking), word 0 is read
  __host__ __device__ static constexpr int xs0(int w) { return w * XB + (w < XR ? w : XR); }
  __host__ __device__ static constexpr int xsn(int w) { return XB + (w < XR ? 1 : 0); }
  __device__ static int img_idx(int tile, int uq, int j) { return (((tile >> 2) * 4 + uq) * 16 + j) * 4 + (tile & 3); }
};

// ---- matrix waves ------------------------------------------------------------------------------------------------------------
template <int KS, int MW, bool SAVE>
__device__ __forceinline__ void cu_fwd_matrix(const CuFwdJob& jb, float* smem, bool& bad) {
  using G = CuGeo<KS>;
  constexpr int N = G::N, QN = G::QN, IMG = G::IMG, FT = G::FT, NG = G::NG;
  const int lane = threadIdx.x & 63, j = lane & 15, uq = lane >> 4;
  const int T = jb.T;
  const float* __restrict__ Up = jb.Up;
  float* img = smem + G::OFF_IMG;
  const float* zring = smem + G::OFF_Z;
  float* pre = smem + G::OFF_PRE + MW * FT * 256;
  float* stg = smem + G::OFF_STG + MW * 2 * 384;
  float* part = smem + G::OFF_PART + MW * 256;
  unsigned* flag_g = reinterpret_cast<unsigned*>(smem + G::OFF_FLAG) + MW * 64 + lane;
  unsigned* flag_x = reinterpret_cast<unsigned*>(smem + G::OFF_FLAG) + (4 + MW) * 64 + lane;

  // A fragments (stationary): tile tau = MW * FT + i, k-step s: A[m = j][k = uq] = U[unit 4s + uq][column tau * 16 + j]
  float uf[FT][KS];
#pragma unroll
  for (int i = 0; i < FT; ++i)
#pragma unroll
    for (int s = 0; s < KS; ++s) uf[i][s] = Up[(size_t)(4 * s + uq) * N + (MW * FT + i) * 16 + j];
  float ux[G::XS > 0 ? G::XS : 1];
  if constexpr (G::E) {
#pragma unroll
    for (int sl = 0; sl < G::XS; ++sl) ux[sl] = sl < G::xsn(MW) ? Up[(size_t)(4 * (G::xs0(MW) + sl) + uq) * N + (4 * FT) * 16 + j] : 0.f;
  }
  float c_last = 0.f;                   // cell state of the wave's own (last) tile
  const int last_tile = MW * FT + FT - 1;
  const int last_idx = G::img_idx(last_tile, uq, j);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the weight loads are retired before the time loop
  __builtin_amdgcn_s_barrier();         // images zeroed, Z_0 / Z_1 landed (helpers)

  int zslot = 0;
  for (int step = 0; step < T; ++step) {
    const float* hb = img + (step & 1) * IMG;
    float* hn = img + ((step & 1) ^ 1) * IMG;
    const float* zs = zring + zslot * (KS * 256);
    zslot = zslot == 2 ? 0 : zslot + 1;
    STAMP(MW, 0)
    // h_{t-1}: the whole B operand of the step, once
    f32x4 hf[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) hf[q] = *reinterpret_cast<const f32x4*>(hb + ((q * 4 + uq) * 16 + j) * 4);
    if constexpr (G::E) {
      // this wave's quarter of K of the left-over tile first: helper 0 finishes that tile while the whole tiles run
      f32x4 ax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < G::XS; ++sl) {
        constexpr int s0 = G::xs0(MW);
        const int s = s0 + sl < KS ? s0 + sl : KS - 1;   // (k-steps beyond the wave's range carry zero weights)
        ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ux[sl], hf[s >> 2][s & 3], ax, 0, 0, 0);
      }
      *reinterpret_cast<f32x4*>(part + lane * 4) = ax;
      lds_flag_write(flag_x, (unsigned)step + 1u);      // (behind the data: the LDS operations of a wave run in order)
      STAMP(MW, 1)
    }
    // Z of every tile of the wave at the top of the step (the reads fly under the first MFMAs)
    f32x4 zt[FT];
#pragma unroll
    for (int i = 0; i < FT; ++i) zt[i] = *reinterpret_cast<const f32x4*>(zs + ((MW * FT + i) * 64 + lane) * 4);
    auto group = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int i0 = G::g_first(g);
      f32x4 a0 = zt[i0];
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (G::g_size(g) == 2) {
        a1 = zt[i0 + 1];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i0][s], hf[s >> 2][s & 3], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i0 + 1][s], hf[s >> 2][s & 3], a1, 0, 0, 0);
        }
      } else {   // a single tile: two accumulators hide the 40-cycle dependent-MFMA latency
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (s & 1)
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i0][s], hf[s >> 2][s & 3], a1, 0, 0, 0);
          else
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i0][s], hf[s >> 2][s & 3], a0, 0, 0, 0);
        }
        a0 += a1;
      }
      if constexpr (g + 1 < NG) {
        // gate pre-activations to the partner helper wave
        *reinterpret_cast<f32x4*>(pre + (i0 * 64 + lane) * 4) = a0;
        if constexpr (G::g_size(g) == 2) *reinterpret_cast<f32x4*>(pre + ((i0 + 1) * 64 + lane) * 4) = a1;
        lds_flag_write(flag_g, (unsigned)(step * NG + g + 1));
      } else {
        // the wave's last tile: its own cell update (nothing left to overlap it with), h into the next image, outputs staged
        float4 g4;
        const float h = mgr_cell_fwd(a0[0], a0[1], a0[2], a0[3], c_last, g4);
        bad = bad || !(fabsf(h) <= 1.f);
        hn[last_idx] = h;
        float* sp = stg + (step & 1) * 384;
        if constexpr (SAVE) {
          *reinterpret_cast<f32x4*>(sp + lane * 4) = (f32x4){g4.x, g4.y, g4.z, g4.w};
          sp[320 + lane] = c_last;
        }
        sp[256 + lane] = h;
      }
      STAMP(MW, 2 + g)
    };
    group(std::integral_constant<int, 0>{});
    if constexpr (NG > 1) group(std::integral_constant<int, 1>{});
    if constexpr (NG > 2) group(std::integral_constant<int, 2>{});
    if constexpr (NG > 3) group(std::integral_constant<int, 3>{});
    if constexpr (NG > 4) group(std::integral_constant<int, 4>{});
    static_assert(NG <= 5, "more groups than unrolled");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(MW, 8)
    __builtin_amdgcn_s_barrier();
    STAMP(MW, 9)
  }
}

// ---- helper waves ------------------------------------------------------------------------------------------------------------
template <int KS, int HW, bool SAVE>
__device__ __forceinline__ void cu_fwd_helper(const CuFwdJob& jb, const ClusterCommon& cm, int bg, float* smem, bool& bad) {
  using G = CuGeo<KS>;
  constexpr int H = G::H, N = G::N, IMG = G::IMG, FT = G::FT, NG = G::NG;
  const int lane = threadIdx.x & 63, j = lane & 15, uq = lane >> 4;
  const int T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < jb.B;
  const int bc = bvalid ? b : jb.B - 1;
  constexpr bool save = SAVE;
  float* img = smem + G::OFF_IMG;
  const float* zring = smem + G::OFF_Z;
  const float* pre = smem + G::OFF_PRE + HW * FT * 256;
  const float* stg = smem + G::OFF_STG + HW * 2 * 384;
  const float* part = smem + G::OFF_PART;
  const unsigned* flags = reinterpret_cast<const unsigned*>(smem + G::OFF_FLAG);   // [8][64], word 0 of each row is read
  const unsigned zring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)(smem + G::OFF_Z));
  const float* Zp = jb.Z;

  // per-lane byte offsets of (sample, unit uq of tile 0) at t = 0; invalid samples store out of range (dropped by the buffer)
  const unsigned zvoff = (unsigned)(((size_t)bc * T * N + (size_t)uq * 4) * sizeof(float));
  const unsigned oob = 0x80000000u;   // (tensors of this kernel stay below 2 GiB: offset + tile / gate bytes never wraps, always out of range)
  const unsigned goff = bvalid ? (unsigned)(((size_t)b * T * H + uq) * 16) : oob;
  const unsigned coff = bvalid ? (unsigned)(((size_t)b * T * H + uq) * 4) : oob;
  const unsigned yoff = bvalid ? (unsigned)(((size_t)b * T * jb.ldy + uq) * 4) : oob;
  const size_t gbytes = (size_t)jb.B * T * H * 16, cbytes = (size_t)jb.B * T * H * 4, ybytes = ((size_t)jb.B * T - 1) * jb.ldy * 4 + (size_t)H * 4;
  // (descriptor inputs made provably wave-uniform: otherwise every buffer store is wrapped in a waterfall loop)
  auto uni = [](const void* p) {
    const uintptr_t v = (uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((uintptr_t)hi << 32) | lo);
  };
  __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(uni(jb.G), 0, __builtin_amdgcn_readfirstlane(save ? (unsigned)gbytes : 0u), 0x00020000);
  __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(uni(jb.Cs), 0, __builtin_amdgcn_readfirstlane(save ? (unsigned)cbytes : 0u), 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(uni(jb.Y), 0, __builtin_amdgcn_readfirstlane((unsigned)ybytes), 0x00020000);
  const int ldy = __builtin_amdgcn_readfirstlane(jb.ldy);

  auto t_of = [&](int step) { return reverse ? T - 1 - step : step; };
  auto dma_z = [&](int step) {    // Z of `step` -> ring slot step % 3: this helper's partner's tiles (+ the left-over tile: helper 0)
    if (step < T) {
      const float* base = Zp + (size_t)t_of(step) * N;
      const unsigned slot = zring_lds + (unsigned)(step % 3) * (KS * 1024);
#pragma unroll
      for (int i = 0; i < FT; ++i) cu_dma_b128(base, zvoff + (HW * FT + i) * 64, slot + (HW * FT + i) * 1024);
      if (G::E && HW == 0) cu_dma_b128(base, zvoff + (4 * FT) * 64, slot + (4 * FT) * 1024);
    }
  };
  auto store_out = [&](int tile, int t, float h, const float4& g4, float cc) {   // outputs of one (unit, sample) cell, from registers
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), ry, yoff + tile * 16, t * ldy * 4, 0);
    if (save) {
      const u32x4 g = {__float_as_uint(g4.x), __float_as_uint(g4.y), __float_as_uint(g4.z), __float_as_uint(g4.w)};
      __builtin_amdgcn_raw_buffer_store_b128(g, rg, goff + tile * 64, t * H * 16, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cc), rc, coff + tile * 16, t * H * 4, 0);
    }
  };
  auto drain_last = [&](int step) {   // outputs of the partner's own (last) tile of `step`: staging -> global
    const float* sp = stg + (step & 1) * 384;
    const float h = sp[256 + lane];
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float cc = 0.f;
    if (save) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(sp + lane * 4);
      g4 = make_float4(g[0], g[1], g[2], g[3]);
      cc = sp[320 + lane];
    }
    store_out(HW * FT + FT - 1, t_of(step), h, g4, cc);
  };
  auto wait_flag = [&](const unsigned* f, unsigned target) {
    unsigned spins = 0;
    while (lds_flag_read(f) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > CU_SPIN_LIMIT) {   // (cannot happen: the matrix waves wait for nobody; never hang the GPU on a bug)
        if (lane == 0) __hip_atomic_store(cm.status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  };

  // prologue: zero both h images (h_{-1} = 0; the k-steps of the last block that do not exist stay zero), first two Z
  for (int i = (int)threadIdx.x - 256; i < 2 * IMG; i += 256) img[i] = 0.f;
  if (HW == 0) {
    unsigned* fw = reinterpret_cast<unsigned*>(smem + G::OFF_FLAG);
    for (int i = lane; i < 8 * 64; i += 64) fw[i] = 0u;
  }
  dma_z(0);
  dma_z(1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  float c[FT > 1 ? FT - 1 : 1];   // cell states of the partner's tiles 0 .. FT - 2
#pragma unroll
  for (int i = 0; i + 1 < FT; ++i) c[i] = 0.f;
  float cx = 0.f;                 // ... and of the left-over tile (helper 0)
  for (int step = 0; step < T; ++step) {
    const int t = t_of(step);
    float* hn = img + ((step & 1) ^ 1) * IMG;
    STAMP(4 + HW, 0)
    dma_z(step + 2);
    STAMP(4 + HW, 1)
    if (step > 0) drain_last(step - 1);
    STAMP(4 + HW, 2)
    if (G::E && HW == 0) {
      // the left-over tile: sum of the four K-quarters + Z, cell, h into the next image, outputs straight to global memory
      constexpr int tile = 4 * FT;
#pragma unroll
      for (int w = 0; w < 4; ++w) wait_flag(flags + (4 + w) * 64, (unsigned)step + 1u);
      const float* zs = zring + (step % 3) * (KS * 256);
      f32x4 tot = *reinterpret_cast<const f32x4*>(zs + (tile * 64 + lane) * 4);
#pragma unroll
      for (int w = 0; w < 4; ++w) tot += *reinterpret_cast<const f32x4*>(part + w * 256 + lane * 4);
      float4 g4;
      const float h = mgr_cell_fwd(tot[0], tot[1], tot[2], tot[3], cx, g4);
      bad = bad || !(fabsf(h) <= 1.f);
      hn[G::img_idx(tile, uq, j)] = h;
      store_out(tile, t, h, g4, cx);
    }
    STAMP(4 + HW, 3)
#pragma unroll
    for (int g = 0; g + 1 < NG; ++g) {   // the partner's groups as they complete: cell update, h into the next image, outputs
      wait_flag(flags + HW * 64, (unsigned)(step * NG + g + 1));
#pragma unroll
      for (int e = 0; e < G::g_size(g); ++e) {
        const int i = G::g_first(g) + e, tile = HW * FT + i;
        const f32x4 a = *reinterpret_cast<const f32x4*>(pre + (i * 64 + lane) * 4);
        float4 g4;
        const float h = mgr_cell_fwd(a[0], a[1], a[2], a[3], c[i], g4);
        bad = bad || !(fabsf(h) <= 1.f);
        hn[G::img_idx(tile, uq, j)] = h;
        store_out(tile, t, h, g4, c[i]);
      }
      STAMP(4 + HW, 4 + g)
    }
    // Z of step + 1 (issued a step ago) must have landed before the barrier lets the matrix waves read it.  Vector-memory
    // operations complete in issue order, so it is enough that all but the operations issued in THIS step are done - counted,
    // because a vmcnt(0) would also wait for the acknowledgement of the stores just issued (1-2 us: measured 4.2 instead of
    // 2.6 us per step).  Every store / DMA below is issued unconditionally (invalid samples store out of range), so the count
    // of a step is a function of (step == 0, step + 2 < T) only.
    {
      constexpr int SPT = SAVE ? 3 : 1;                                   // stores per tile
      constexpr int D = FT + ((G::E && HW == 0) ? 1 : 0);                 // DMAs of dma_z
      constexpr int S0 = SPT * (FT - 1) + ((G::E && HW == 0) ? SPT : 0);  // groups + left-over tile
      const bool dma = step + 2 < T, first = step == 0;
      if (dma && !first)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + S0 + SPT) : "memory");
      else if (dma)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + S0) : "memory");
      else if (!first)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S0 + SPT) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S0) : "memory");
    }
    STAMP(4 + HW, 8)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    STAMP(4 + HW, 9)
  }
  if (T > 0) drain_last(T - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int KS, bool SAVE>
__global__ __launch_bounds__(512) void k_scan_cu_fwd(CuFwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  for (int k = 0; k < L.njobs; ++k) {
    const CuFwdJob& jb = L.job[k];
    const int bg = (int)blockIdx.x - jb.wg_begin;
    if (bg < 0 || bg >= jb.nbg) continue;
    bool bad = false;
    switch (wv) {
      case 0: cu_fwd_matrix<KS, 0, SAVE>(jb, smem, bad); break;
      case 1: cu_fwd_matrix<KS, 1, SAVE>(jb, smem, bad); break;
      case 2: cu_fwd_matrix<KS, 2, SAVE>(jb, smem, bad); break;
      case 3: cu_fwd_matrix<KS, 3, SAVE>(jb, smem, bad); break;
      case 4: cu_fwd_helper<KS, 0, SAVE>(jb, L.cm, bg, smem, bad); break;
      case 5: cu_fwd_helper<KS, 1, SAVE>(jb, L.cm, bg, smem, bad); break;
      case 6: cu_fwd_helper<KS, 2, SAVE>(jb, L.cm, bg, smem, bad); break;
      default: cu_fwd_helper<KS, 3, SAVE>(jb, L.cm, bg, smem, bad); break;
    }
    // a NaN / Inf hidden state travels through the recurrence as it does in the reference; the launch still raises
    // MGR_SCAN_NONFINITE so that the update gate keeps the step away from the weights (mgr.h)
    if (__any(bad) && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(L.cm.sticky, MGR_ST_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CU_STAMP
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 128)
      L.cm.status[64 + threadIdx.x] = reinterpret_cast<unsigned*>(smem + CuGeo<KS>::LDS_FLOATS)[threadIdx.x];
#endif
    break;
  }
  mgr_cluster_exit(L.cm);
}

#define CU_FOREACH(X) X(25) X(16) X(9) X(8) X(4)

}  // namespace

bool mgr_scan_cu_supported(int H) {
#define CU_CASE(KS) \
  if (H == 4 * KS) return true;
  CU_FOREACH(CU_CASE)
#undef CU_CASE
  return false;
}

// All jobs share H; every job gets ceil(B / 16) workgroups of 512 threads that each want a whole CU.  hdr: the zeroed launch
// header (arrival counter), seq: the launch's sequence number for mgr_stream_wait_last_resident.
int mgr_scan_cu_fwd_launch(mgr_ctx* c, int njobs, const mgr_scan_job* jobs, unsigned* hdr, unsigned seq) {
  CuFwdLaunch L;
  memset(&L, 0, sizeof(L));
  const int H = jobs[0].H;
  int grid = 0;
  for (int i = 0; i < njobs; ++i) {
    const mgr_scan_job& j = jobs[i];
    MGR_REQUIRE(j.H == H && !j.R && !j.YT && (!j.gates == !j.cs), "single-CU scan: jobs must share H, no residual / transposed output");
    MGR_REQUIRE((size_t)j.B * j.T * 4 * j.H * sizeof(float) < ((size_t)1 << 31) && (size_t)j.B * j.T * j.ldy * sizeof(float) < ((size_t)1 << 31),
                "single-CU scan: tensors must stay below 2 GiB");
    CuFwdJob& q = L.job[L.njobs++];
    q.Z = j.Z; q.Up = j.Up; q.Y = j.Y; q.G = j.gates; q.Cs = j.cs;
    q.ldy = j.ldy; q.B = j.B; q.T = j.T; q.reverse = j.reverse;
    q.nbg = (j.B + 15) / 16;
    q.wg_begin = grid;
    grid += q.nbg;
  }
  const bool save = jobs[0].gates != nullptr;
  for (int i = 0; i < njobs; ++i) MGR_REQUIRE((jobs[i].gates != nullptr) == save, "single-CU scan: all jobs save their state or none does");
  MGR_REQUIRE(grid <= c->cu_count, "single-CU scan: %d workgroups want a CU each, the device has %d", grid, c->cu_count);
  L.cm.status = hdr;
  L.cm.sticky = mgr_status_block(c);
  L.cm.resident = c->sticky_status + 1;
  L.cm.seq = seq;
  L.cm.total_wgs = grid;
#define CU_CASE(KS)                                                                                                          \
  if (H == 4 * KS) {                                                                                                         \
    const size_t lds = (size_t)CuGeo<KS>::LDS_FLOATS * sizeof(float);                                                        \
    const size_t ask = (lds < 120 * 1024 ? 120 * 1024 : lds) + 1024;   /* a CU of its own: no other workgroup with an LDS footprint fits */ \
    if (save) {                                                                                                              \
      MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cu_fwd<KS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      hipLaunchKernelGGL((k_scan_cu_fwd<KS, true>), dim3(grid), dim3(512), ask, mgr_stream(c), L);                           \
    } else {                                                                                                                 \
      MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cu_fwd<KS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      hipLaunchKernelGGL((k_scan_cu_fwd<KS, false>), dim3(grid), dim3(512), ask, mgr_stream(c), L);                          \
    }                                                                                                                        \
    MGR_LAUNCH_CHECK();                                                                                                      \
    return 0;                                                                                                                \
  }
  CU_FOREACH(CU_CASE)
#undef CU_CASE
  return mgr_fail(-1, "single-CU scan: no instantiation for H = %d", H);
}
