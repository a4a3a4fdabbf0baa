// K3, paired form of the multi-CU forward recurrence: ONE 8-wave workgroup per CU serves TWO batch groups (A, B) of the same
// layer-direction and alternates between them, so that the hand-off of one group's h_t - cell update, publish, fabric flight,
// gather - runs while the matrix cores work on the other group.
//
// Why.  In the one-group form (lstm_cluster.hip: cluster_run_ks) a time step is a dependent chain
//     gather (one store -> load flight) -> MFMAs (1.7 us at H = 500) -> reduce + cell (~0.5 us) -> publish -> ...
// of ~3.9 us, and the chip only stays busy because TWO workgroups of different clusters share a CU.  Their phases are not locked
// (audio and skeletal steps take different times), so their MFMA phases collide on the SIMDs as often as not: 5.4 us per step for
// the four encoder scans of config F.  Here the two groups that share a CU are in lock step by construction:
//     matrix waves :  | MFMA A(s) | MFMA B(s) | MFMA A(s+1) | ...
//     cell waves   :              | cell A(s) | cell B(s)   | ...          (publish at the end of each)
//     fabric       :                          |<- A's h_s in flight ->|    (gather issued from inside MFMA B(s))
// and a group's hand-off has the whole MFMA phase of the other group to complete.
//
// Roles (fixed per wave, one wave of each role per SIMD):
//   * MATRIX waves 0..3: wave w owns a QUARTER of K for all four tiles of the workgroup (U^T fragments stationary in 128 VGPRs).
//     It fetches its image blocks of the group's h_{t-1} by LDS-DMA (global_load_lds_dwordx4: no register destination, so no
//     compiler-visible value ever has a load pending - the first version of this kernel polled registers across the phases and
//     hipcc copied them while the loads were in flight), POLLS THE LANDING ZONE in LDS (the data is the flag: a word still showing
//     the preset has not landed, a landed word with the previous epoch's parity means the producer was late -> fetch again), runs
//     its 4 x 32 MFMAs on the polled copies and leaves four partial tiles in LDS.  It issues no other global memory operation in
//     the loop, so nothing it executes ever waits on vmcnt - in particular not on a write-through store acknowledgement.
//   * CELL waves 4..7: wave 4+u finishes the cells (unit-in-tile u) x (4 tiles) x (16 samples): sums the four partial tiles, adds
//     Z_t, runs the cell, publishes h_t (one coalesced 256-byte write-through store per wave, parity in the mantissa LSB) and
//     streams Y / gates / c out.  Z is prefetched two steps ahead; these waves never gather.
//   One s_barrier per (group, step) hands the partial sums over; partial sums and landing zones are double-buffered BY GROUP (the
//   other group's phase separates writer and reader generations).
// Same private unit order, hand-off protocol (two exchange slots, parity = epoch, bounded spins, status words, non-finite guard)
// as cluster_run_ks.  Config F's four encoder scans: 128 + 76 = 204 workgroups, one per CU.
#include <type_traits>

#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;
constexpr unsigned KP_ROUND_LIMIT = 1u << 16;

template <int KS>
__device__ __forceinline__ void cluster_run_pair(const ClusterJob& jb, const ClusterCommon& cm, int issue_at, int pr, int ug, float* smem) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256, NB = (QN + 3) / 4;   // image blocks per matrix wave
  static_assert(NB >= 1 && NB <= 8, "1..8 image blocks per matrix wave (H <= 512)");
  constexpr int PART = 4 * 4 * 64 * 4;   // floats of one partial-sum buffer: [tile 4][src wave 4][64 lanes] f32x4
  unsigned* status = cm.status;
  issue_at = (NB * issue_at + 4) / 8;   // eighths of the chain -> block index
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const bool matrix = wave < 4;
  const int j = lane & 15, uq = lane >> 4;
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int ngr = (2 * pr + 1 < jb.nbg) ? 2 : 1;   // groups this workgroup really serves (workgroup-uniform)
  float* zone = smem;                 // [group][IMG]  landing zones of the gathers
  float* part = smem + 2 * IMG;       // [group][PART] partial sums
  const unsigned zone_lds = (unsigned)(uintptr_t)(lds_float*)zone;   // LDS byte address (M0 of the LDS-DMA loads)

  auto unit_of = [](int tile, int u) {   // hidden unit of MFMA slot (tile, unit-in-tile) = of k-slot (s = tile, kk = u): see cluster_run_ks
    const int q = tile >> 2, nv = (KS - 4 * q) < 4 ? (KS - 4 * q) : 4;
    return 16 * q + nv * u + (tile & 3);
  };
  const char* xbase[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) xbase[g] = reinterpret_cast<const char*>(jb.xbuf + (size_t)(2 * pr + (g < ngr ? g : 0)) * 2 * IMG);

  if (matrix) {
    // =========================================================================================== matrix waves
    const int qb = wave * NB;
    int nb = QN - qb;
    nb = nb < 0 ? 0 : (nb > NB ? NB : nb);
    nb = __builtin_amdgcn_readfirstlane(nb);
    const float* __restrict__ Up = jb.Up;
    float uf[4][NB * 4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int gt = ug * 4 + tt;
#pragma unroll
      for (int sl = 0; sl < NB * 4; ++sl) {
        const int s = qb * 4 + sl;
        uf[tt][sl] = (gt < KS && s < KS) ? Up[(size_t)unit_of(s, uq) * N + unit_of(gt, j >> 2) * 4 + (j & 3)] : 0.f;
      }
    }
    bool failed = false;

    // fetch group g's h_{step-1} into its landing zone: preset the zone to a pattern no published word can have (quiet-NaN
    // exponent, wrong epoch parity), then one LDS-DMA load per block (a wave-instruction writes the block's 1 KiB contiguously)
    auto issue = [&](int g, int step) {
      const int slot = (step - 1) & 1;
      const unsigned par = ((((unsigned)(step - 1)) >> 1) & 1u) ^ 1u;
      const unsigned bad = 0x7FC00000u | (par ^ 1u);
      float* zg = zone + g * IMG;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (i < nb) *reinterpret_cast<u32x4*>(zg + (qb + i) * 256 + lane * 4) = (u32x4){bad, bad, bad, bad};   // (nb: wave-uniform)
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the presets are in LDS before any DMA write can land
      const char* p = xbase[g] + ((size_t)slot * IMG + (size_t)qb * 256 + lane * 4) * 4;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (i < nb) {
          const unsigned m0v = __builtin_amdgcn_readfirstlane(zone_lds + (unsigned)((g * IMG + (qb + i) * 256) * 4));
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(p + i * 1024), "s"(m0v)
                       : "memory");
        }
      }
    };
    // wait until every word of the wave's blocks shows the parity of step-1; returns the words (MFMA B operands) in t
    // (block slots beyond the image - nb is wave-uniform - read the wave's first block again: fresh words that meet zero weights)
    auto await = [&](int g, int step, u32x4 (&t)[NB]) {
      const unsigned par = ((((unsigned)(step - 1)) >> 1) & 1u) ^ 1u;
      const float* zg = zone + g * IMG;
      unsigned rounds = 0, spins = 0;
      for (;;) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
          t[i] = *reinterpret_cast<const volatile u32x4*>(zg + ((i < nb) ? qb + i : (nb > 0 ? qb : 0)) * 256 + lane * 4);
        unsigned a_and = t[0].x, a_or = t[0].x;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          a_and &= t[i].x & t[i].y & t[i].z & t[i].w;
          a_or |= t[i].x | t[i].y | t[i].z | t[i].w;
        }
        const bool lane_fresh = par ? (a_and & 1u) != 0u : (a_or & 1u) == 0u;
        if (__all(lane_fresh) || failed || nb == 0) break;   // every word shows this epoch (hence has landed)
        bool again = false;
        if (__all(((a_or >> 30) & 1u) == 0u)) {       // everything landed, something was still the previous epoch
          again = true;
          ++rounds;
        } else if (++spins > 4096u) {                 // a load cannot take this long (~1 ms): drain and start over
          again = true;
          rounds += 64;
        }
        if (again) {
          if ((rounds & 63u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) failed = true;
          if (rounds > KP_ROUND_LIMIT) {
            failed = true;
            if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of the stale round may land after the new presets
          issue(g, step);
          spins = 0;
        } else {
          __builtin_amdgcn_s_sleep(1);
        }
      }
    };
    auto mfma_block = [&](auto ic, const u32x4 (&t)[NB], f32x4 (&acc)[4]) {
      constexpr int i = decltype(ic)::value;
      const float hv[4] = {__uint_as_float(t[i].x), __uint_as_float(t[i].y), __uint_as_float(t[i].z), __uint_as_float(t[i].w)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[tt][i * 4 + r], hv[r], acc[tt], 0, 0, 0);
      }
    };
    // one phase: time step `step` of group g; the gather of the other group (for its step `ostep`) leaves after issue_at blocks
    auto phase = [&](int g, int step, bool do_other, int ostep, float* pbuf) {
      f32x4 acc[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (step > 0) {
        u32x4 t[NB];
        await(g, step, t);
#define KP_BLK(I)                                                        \
  if constexpr (I < NB) {                                                \
    if (do_other && issue_at == I) issue(g ^ 1, ostep);                  \
    mfma_block(std::integral_constant<int, (I < NB ? I : 0)>{}, t, acc); \
  }
        KP_BLK(0) KP_BLK(1) KP_BLK(2) KP_BLK(3) KP_BLK(4) KP_BLK(5) KP_BLK(6) KP_BLK(7)
#undef KP_BLK
        if (do_other && issue_at >= NB) issue(g ^ 1, ostep);
      } else if (do_other) {
        __builtin_amdgcn_s_sleep(16);   // (step 0 has no MFMA chain to hide behind: a short head start for the peers' publish)
        issue(g ^ 1, ostep);
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(pbuf + ((tt * 4 + wave) * 64 + lane) * 4) = acc[tt];
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): partial sums written
      __builtin_amdgcn_s_barrier();
    };
    if (ngr == 2) {
      // phases A(0) B(0) A(1) B(1) ...: inside A(s) the gather for B(s) [B's h of step s-1], inside B(s) the one for A(s+1)
      for (int step = 0; step < T; ++step) {
        phase(0, step, step > 0, step, part);
        phase(1, step, step + 1 < T, step + 1, part + PART);
      }
    } else {
      // a lone group: nothing to overlap with; its partial sums alternate between the two buffers
      for (int step = 0; step < T; ++step) {
        if (step > 0) {
          __builtin_amdgcn_s_sleep(8);
          issue(0, step);
        }
        phase(0, step, false, 0, part + (step & 1) * PART);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may still be in flight when the wave ends
  } else {
    // =========================================================================================== cell waves
    const int u = wave & 3;
    const float* __restrict__ Z = jb.Z;
    const int ftile = ug * 4 + uq;                 // lane (r = uq, sample j) of cell wave u owns slot (tile 4*ug + r, unit-in-tile u)
    const bool cvalid = ftile < KS;
    const int unit = cvalid ? unit_of(ftile, u) : 0;
    const int red_off = ((uq * 4) * 64 + u * 16 + j) * 4;   // [tile = uq][src 0..3][slot lane = u*16 + j]
    bool nonfinite = false;
    struct Group {
      int b, bc;
      bool bvalid;
      float c;
      f32x4 z0, z1, z2;
    };
    Group gr[2];
    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int bg = 2 * pr + (g < ngr ? g : 0);
      gr[g].b = bg * 16 + j;
      gr[g].bvalid = g < ngr && gr[g].b < B;
      gr[g].bc = gr[g].b < B ? gr[g].b : B - 1;
      rs[g] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase[g]), 0, 2 * IMG * 4, 0x00020000);
      gr[g].c = 0.f;
      gr[g].z0 = gr[g].z1 = gr[g].z2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto loadz = [&](f32x4& z, int g, int step) {
      if (step < T && cvalid && g < ngr) {
        const int t = reverse ? T - 1 - step : step;
        z = *reinterpret_cast<const f32x4*>(Z + ((size_t)gr[g].bc * T + t) * N + unit * 4);
      }
    };
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      loadz(gr[g].z0, g, 0);
      loadz(gr[g].z1, g, 1);
    }
    auto finish = [&](int g, int step, const float* pbuf, const f32x4& zuse, f32x4& zload) {
      loadz(zload, g, step + 2);
      __builtin_amdgcn_s_barrier();   // the matrix waves have left this phase's partial sums in pbuf
      const int t = reverse ? T - 1 - step : step;
      const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
      unsigned hbits = par;   // cells of a padding tile: value 0 with the current parity (consumers test whole blocks)
      float h = 0.f, yv = 0.f;
      float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cvalid) {
        f32x4 tot = zuse;
        const float* mine = pbuf + red_off;
#pragma unroll
        for (int src = 0; src < 4; ++src) tot += *reinterpret_cast<const f32x4*>(mine + src * 64 * 4);
        h = mgr_cell_fwd(tot[0], tot[1], tot[2], tot[3], gr[g].c, g4);
        yv = h;
        if (!(fabsf(h) < 2.f)) {   // NaN / Inf: Y keeps it, the published / recurrent value stays finite (cluster_run_ks)
          h = 0.f;
          gr[g].c = 0.f;
          if (!nonfinite) __hip_atomic_fetch_or(cm.sticky, MGR_ST_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          nonfinite = true;
        }
        hbits = (__float_as_uint(h) & ~1u) | par;  // epoch parity rides in the mantissa LSB
        h = __uint_as_float(hbits);
        if (!nonfinite) yv = h;
      }
      if (step + 1 < T) {
        // lane (r, j) holds image word j*4 + r of the wave's 64-word segment: bring word l to lane l, one coalesced store
        const unsigned w = __builtin_amdgcn_ds_bpermute((((lane & 3) << 4) | (lane >> 2)) << 2, hbits);
        __builtin_amdgcn_raw_buffer_store_b32(w, rs[g], ((step & 1) * IMG + (ug * 4 + u) * 64 + lane) * 4, 0, 16);  // sc1
      }
      if (cvalid && gr[g].bvalid) {
        size_t row = (size_t)gr[g].b * T + t;
        float yo = yv;
        if (jb.R) yo += jb.R[row * jb.ldr + unit];
        jb.Y[row * jb.ldy + unit] = yo;
        if (jb.G) *reinterpret_cast<float4*>(jb.G + (row * H + unit) * 4) = g4;
        if (jb.Cs) jb.Cs[row * H + unit] = gr[g].c;
      }
    };
    if (ngr == 2) {
      auto both = [&](int step, f32x4& zau, f32x4& zal, f32x4& zbu, f32x4& zbl) {
        finish(0, step, part, zau, zal);
        finish(1, step, part + PART, zbu, zbl);
      };
      for (int s0 = 0; s0 < T; s0 += 3) {
        both(s0, gr[0].z0, gr[0].z2, gr[1].z0, gr[1].z2);
        if (s0 + 1 < T) both(s0 + 1, gr[0].z1, gr[0].z0, gr[1].z1, gr[1].z0);
        if (s0 + 2 < T) both(s0 + 2, gr[0].z2, gr[0].z1, gr[1].z2, gr[1].z1);
      }
    } else {
      for (int s0 = 0; s0 < T; s0 += 3) {
        finish(0, s0, part + (s0 & 1) * PART, gr[0].z0, gr[0].z2);
        if (s0 + 1 < T) finish(0, s0 + 1, part + ((s0 + 1) & 1) * PART, gr[0].z1, gr[0].z0);
        if (s0 + 2 < T) finish(0, s0 + 2, part + ((s0 + 2) & 1) * PART, gr[0].z2, gr[0].z1);
      }
    }
  }
}

#define CLP_FOREACH(X) X(125) X(75) X(32) X(25)

__global__ __launch_bounds__(512, 1) void k_scan_cluster_pair(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
  for (int k = 0; k < L.njobs; ++k) {
    const ClusterJob& jb = L.job[k];
    const int w = (int)blockIdx.x - jb.cls_begin;
    if (w < 0 || w >= jb.cls_nclusters * jb.G_) continue;
    const int ug = w % jb.G_, pr = w / jb.G_ - jb.cls_cluster0;   // (clusters of a pair-mode class are PAIRS of batch groups)
    if (pr < 0 || pr >= (jb.nbg + 1) / 2) continue;
#define CLP_CASE(KS) \
  if (jb.ks == KS) { cluster_run_pair<KS>(jb, L.cm, L.issue_at, pr, ug, smem); return mgr_cluster_exit(L.cm); }
    CLP_FOREACH(CLP_CASE)
#undef CLP_CASE
    return;
  }
}

}  // namespace

bool mgr_cluster_pair_supported(int ks) {
#define CLP_CASE(KS) \
  if (ks == KS) return true;
  CLP_FOREACH(CLP_CASE)
#undef CLP_CASE
  return false;
}

int mgr_cluster_pair_launch(mgr_ctx* c, const ClusterLaunch& L, int total_wgs) {
  MGR_REQUIRE(total_wgs <= c->cu_count, "paired cluster scan needs %d co-resident workgroups but the device has %d CUs", total_wgs,
              c->cu_count);
  if (!(c->attr_done & 4u)) {
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_pair), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->attr_done |= 4u;
  }
  // landing zones (two images) + two partial-sum buffers; at least 84 KiB so that exactly one of these workgroups sits on a CU
  size_t lds = 0;
  for (int i = 0; i < L.njobs; ++i) {
    const size_t need = (size_t)(2 * ((L.job[i].ks + 3) / 4) * 256 + 2 * 4 * 4 * 64 * 4) * sizeof(float);
    lds = need > lds ? need : lds;
  }
  if (lds < 84 * 1024) lds = 84 * 1024;
  hipLaunchKernelGGL(k_scan_cluster_pair, dim3(total_wgs), dim3(512), lds, mgr_stream(c), L);
  MGR_LAUNCH_CHECK();
  return 0;
}
