// K3: the forward LSTM recurrence - persistent, weight-stationary, on the f32 matrix cores.
//
// Orientation: the MFMA computes  D[gate-col, sample] += U^T[gate-col, k] * h^T[k, sample]  with
// v_mfma_f32_16x16x4_f32: M = 16 packed gate columns = one TILE of 4 units x (i,f,c,o), N = 16 samples (one batch
// group), K = 4 per MFMA k-step.
//   * A operand (U^T fragment) never changes: a wave keeps the fragments of its tiles in VGPRs for all T steps.
//   * B operand (h_{t-1}) is read from a double-buffered LDS image laid out [k/16][k%4][sample][(k/4)%4] floats, so one
//     ds_read_b128 feeds four consecutive k-steps conflict-free; reads run 3 blocks ahead of their MFMAs.
//   * C/D layout: lane (sample = lane&15, unit-in-tile = lane>>4) receives the 4 gates of ITS (unit, sample) in its
//     4 accumulator registers, so the cell update needs no cross-lane traffic.
//   * Z[t] (gate pre-activations from the input projection) is prefetched two steps ahead through a 3-deep register
//     ring; h_t, the activated gates and c_t stream out with fire-and-forget stores; ONE s_barrier per time step.
//
// A CLUSTER = the G workgroups (one per CU) that together hold one direction's recurrent matrix for one batch group;
// workgroup `ug` owns tiles [ug*tpwg, (ug+1)*tpwg), dealt round-robin to its waves.  G = 1 (H <= 128) needs no
// exchange.  For G > 1 (H = 300, 500) every step ends with an all-gather of h_t inside the cluster:
//   PUBLISH: each h value is stored the moment it is computed - a 4-byte write-through (sc1) store into the cluster's
//      exchange slot (t&1), same image layout.  THE DATA IS THE FLAG: the least-significant mantissa bit of each value
//      carries the epoch parity ((t>>1)&1)^1, which flips every time a slot word is rewritten (the local copy, Y and the
//      recurrence all use the same 1-ulp-adjusted value, so all replicas agree bit for bit).
//   GATHER: each of the 8 waves sweeps its share of the image's 1 KiB blocks with 16-byte sc1 loads, accepts a block
//      once all of its words show the expected parity, and writes it to the next LDS image.
// This is the CDNA guide's granule hand-off (Guideline 16 R2, "the data is the flag") with a 4-byte granule: every
// word is written by exactly one aligned store per epoch; a reader of epoch t can only ever see the word of epoch t-2
// (opposite parity) or t, never t+2, because a producer cannot publish epoch t+2 before every peer has published t+1,
// i.e. finished consuming t.  No fence, no flag round trip, no drain: a step costs ONE store->load flight.
// hipMalloc memory; one workgroup per CU (enforced by requesting > 80 KiB of LDS); slots are zeroed by a memset node
// ahead of every launch.  Every spin is bounded; a give-up sets status[0] and the host reports an error instead of
// hanging the GPU.
//
// Several layer-directions ("jobs": audio fwd/rev, skeletal fwd/rev) share ONE launch so that all spinning workgroups
// are co-resident by construction (grid <= number of CUs).
#include <type_traits>

#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CL_WAVES = 8;
constexpr unsigned POLL_LIMIT = 1u << 20;
constexpr unsigned KS_ROUND_LIMIT = 1u << 16;   // K-split step: ~0.1 s of re-polling a late producer, ~1 s of lost loads

template <int KS, int TPW>
__device__ __forceinline__ void cluster_run(const ClusterJob& jb, int bg, int ug, int cl, unsigned* xcc, int xcd_local,
                                            int gather_delay, float* smem, unsigned* status) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256;
  static_assert(QN <= 32, "gather sweep covers at most 32 image blocks (H <= 512)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int nwv = blockDim.x >> 6;  // waves in this workgroup: 8, or 4 when every job runs one tile per SIMD
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches
  const int j = lane & 15, uq = lane >> 4;
  const int G = jb.G_;
  const int nw = jb.nw;
  const int tpwg = nw * TPW;  // tiles per workgroup, a multiple of 4
  // XCD-local exchange (plain stores into the shared L2 + nt loads) measured SLOWER than write-through on MI355X
  // (65-71 vs 58 ms per F step: the 64 KiB slot hammers a few L2 channels), so it is opt-in (mgr_tune key 3)
  const bool fast = xcd_local && mgr_cluster_same_xcd(xcc, blockIdx.x, jb.cls_begin, jb.cls_nclusters, cl, G, status);
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  const float* __restrict__ Z = jb.Z;
  const float* __restrict__ Up = jb.Up;
  float* img = smem;  // [2][IMG]

  // this wave's tiles: ug*tpwg + jt*nw + wave, jt < nt   (nt is wave-uniform)
  int own = KS - ug * tpwg;
  own = own > tpwg ? tpwg : own;
  int nt = 0;
  if (wave < nw) {
    for (int jt = 0; jt < TPW; ++jt)
      if (jt * nw + wave < own) nt = jt + 1;
  }
  nt = __builtin_amdgcn_readfirstlane(nt);

  float uf[TPW][KS];
  int tl[TPW];
#pragma unroll
  for (int jt = 0; jt < TPW; ++jt) {
    const bool v = jt < nt;
    tl[jt] = v ? ug * tpwg + jt * nw + wave : 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) uf[jt][s] = v ? Up[(size_t)(4 * s + uq) * N + tl[jt] * 16 + j] : 0.f;
  }
  for (int i = tid; i < 2 * IMG; i += nwv * 64) img[i] = 0.f;

  const int q0 = (ug * tpwg) >> 2, nq = tpwg >> 2;  // own 1 KiB blocks of the image
  float* xb = jb.xbuf + (size_t)bg * 2 * IMG;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * IMG * 4, 0x00020000);

  float c[TPW];
#pragma unroll
  for (int jt = 0; jt < TPW; ++jt) c[jt] = 0.f;
  // Z ring: step s uses ring[s % 3]; the load for step s+2 is issued at step s
  f32x4 zr0[TPW], zr1[TPW], zr2[TPW];
  auto loadz = [&](f32x4 (&z)[TPW], int step) {
    if (step < T) {
      const int t = reverse ? T - 1 - step : step;
#pragma unroll
      for (int jt = 0; jt < TPW; ++jt)
        if (jt < nt) z[jt] = *reinterpret_cast<const f32x4*>(Z + ((size_t)bc * T + t) * N + (tl[jt] * 4 + uq) * 4);
    }
  };
#pragma unroll
  for (int jt = 0; jt < TPW; ++jt) {
    zr0[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    zr1[jt] = zr0[jt];
    zr2[jt] = zr0[jt];
  }
  loadz(zr0, 0);
  loadz(zr1, 1);
  bool failed = false;
  __syncthreads();
  int cur = 0;
#ifdef MGR_STAMP
  unsigned long long st_mfma = 0, st_cell = 0, st_gather = 0, st_bar = 0, st_passes = 0;
#define STAMP(x) unsigned long long x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0)
#else
#define STAMP(x)
#endif

  // the MFMA chain of one step, specialised on the number of tiles this wave really owns
  auto mfma_phase = [&](auto ntc, f32x4 (&acc)[TPW], const float* hb) {
    constexpr int NT = decltype(ntc)::value;
    if constexpr (NT > 0) {
      constexpr int PD = 3;
      f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
      f32x4 hbuf[4];
      const float* hlane = hb + (uq * 16 + j) * 4;
#pragma unroll
      for (int q = 0; q < PD && q < QN; ++q) hbuf[q] = *reinterpret_cast<const f32x4*>(hlane + q * 256);
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        // B-operand reads run PD blocks ahead of their MFMAs (sched_barrier pins the order; left alone, hipcc sinks
        // each ds_read next to its use and the LDS latency shows between MFMA groups)
        if (q + PD < QN) hbuf[(q + PD) & 3] = *reinterpret_cast<const f32x4*>(hlane + (q + PD) * 256);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 hv = hbuf[q & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * q + r < KS) {
            if (NT == 1 && (r & 1)) {  // one tile: two accumulators hide the 40-cycle dependent-MFMA latency
              acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[0][4 * q + r], hv[r], acc2, 0, 0, 0);
            } else {
#pragma unroll
              for (int jt = 0; jt < NT; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[jt][4 * q + r], hv[r], acc[jt], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NT == 1) acc[0] += acc2;
    }
  };

  auto do_step = [&](int step, f32x4 (&zuse)[TPW], f32x4 (&zload)[TPW]) {
    const int t = reverse ? T - 1 - step : step;
    loadz(zload, step + 2);
    f32x4 acc[TPW];
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) acc[jt] = zuse[jt];
    const float* hb = img + cur * IMG;
    float* hn = img + (cur ^ 1) * IMG;
    const int slot = step & 1;
    const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
    STAMP(s0);
    // wave-uniform dispatch on the owned tile count (scalar branches): no per-MFMA exec masking
    if (nt == 1) {
      mfma_phase(std::integral_constant<int, 1>{}, acc, hb);
    } else if (nt == 2) {
      if constexpr (TPW >= 2) mfma_phase(std::integral_constant<int, 2>{}, acc, hb);
    } else if (nt == 3) {
      if constexpr (TPW >= 3) mfma_phase(std::integral_constant<int, 3>{}, acc, hb);
    } else if (nt == 4) {
      if constexpr (TPW >= 4) mfma_phase(std::integral_constant<int, 4>{}, acc, hb);
    }
#ifdef MGR_STAMP
    asm volatile("" ::"v"(acc[0][0]));
#endif
    STAMP(s1);
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) {
      if (jt < nt) {
        const int tile = tl[jt];
        const int unit = tile * 4 + uq;
        float4 g4;
        float h = mgr_cell_fwd(acc[jt][0], acc[jt][1], acc[jt][2], acc[jt][3], c[jt], g4);
        // unit k = 4*tile + uq -> k-step s = tile, kk = uq -> image [q = tile>>2][kk = uq][j][r = tile&3]
        const int idx = (((tile >> 2) * 4 + uq) * 16 + j) * 4 + (tile & 3);
        if (G > 1) {
          const unsigned hbits = (__float_as_uint(h) & ~1u) | par;  // epoch parity rides in the mantissa LSB
          h = __uint_as_float(hbits);
          if (step + 1 < T) {
            if (fast)  // whole cluster on one XCD: a plain store lands in the L2 every peer's sc1 load is served from
              *reinterpret_cast<volatile unsigned*>(xb + slot * IMG + idx) = hbits;
            else
              __builtin_amdgcn_raw_buffer_store_b32(hbits, rs, (slot * IMG + idx) * 4, 0, 16);  // sc1 write-through
          }
        }
        hn[idx] = h;
        if (bvalid) {
          size_t row = (size_t)b * T + t;
          float yo = h;
          if (jb.R) yo += jb.R[row * jb.ldr + unit];
          jb.Y[row * jb.ldy + unit] = yo;
          if (jb.G) *reinterpret_cast<float4*>(jb.G + (row * H + unit) * 4) = g4;
          if (jb.Cs) jb.Cs[row * H + unit] = c[jt];
        }
      }
    }
    STAMP(s2);
    if (G > 1 && step + 1 < T) {
      // gather: wave w sweeps blocks w, w+nwv, w+2*nwv, ... of the exchange slot (up to 8 loads in flight per round) until
      // every word of a block shows this epoch's parity
      constexpr int NF = 8;  // loads in flight per wave and round
      for (int d = 0; d < gather_delay; ++d) __builtin_amdgcn_s_sleep(1);  // see ClusterLaunch::gather_delay
      for (int base = 0; base < QN && !failed; base += NF * nwv) {
        u32x4 v[NF];
        unsigned pend = 0;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          int q = base + wave + nwv * i;
          if (q < QN && (q < q0 || q >= q0 + nq)) pend |= 1u << i;
        }
        unsigned spins = 0;
        while (pend && !failed) {
#pragma unroll
          for (int i = 0; i < NF; ++i)
            if (pend & (1u << i))
              v[i] = fast ? __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * IMG + (base + wave + nwv * i) * 256 + lane * 4) * 4, 0, 2)    // nt: L1 bypass, served by the XCD's L2
                          : __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * IMG + (base + wave + nwv * i) * 256 + lane * 4) * 4, 0, 16);  // sc1
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            if (pend & (1u << i)) {
              const int q = base + wave + nwv * i;
              const int nvalid = KS - 4 * q;  // k-steps of this block that exist (words r >= nvalid are never written)
              unsigned a = par ? 0xFFFFFFFFu : 0u;
              if (par) {
                a &= v[i].x;
                if (nvalid > 1) a &= v[i].y;
                if (nvalid > 2) a &= v[i].z;
                if (nvalid > 3) a &= v[i].w;
              } else {
                a |= v[i].x;
                if (nvalid > 1) a |= v[i].y;
                if (nvalid > 2) a |= v[i].z;
                if (nvalid > 3) a |= v[i].w;
              }
              if (__all((a & 1u) == par)) {
                *reinterpret_cast<u32x4*>(hn + q * 256 + lane * 4) = v[i];
                pend &= ~(1u << i);
              }
            }
          }
#ifdef MGR_STAMP
          st_passes += 1;
#endif
          if (pend) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 255u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) failed = true;
            if (spins > POLL_LIMIT) {
              failed = true;
              if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
    }
    STAMP(s3);
    __syncthreads();  // next image complete (own slice + gathered peers); everyone is done reading the current one
#ifdef MGR_STAMP
    {
      STAMP(s4);
      st_mfma += s1 - s0;
      st_cell += s2 - s0;  // whole compute phase incl. MFMA
      st_gather += s3 - s2;
      st_bar += s4 - s3;
    }
#endif
    cur ^= 1;
  };

  for (int s0 = 0; s0 < T; s0 += 3) {
    do_step(s0, zr0, zr2);
    if (s0 + 1 < T) do_step(s0 + 1, zr1, zr0);
    if (s0 + 2 < T) do_step(s0 + 2, zr2, zr1);
  }
#ifdef MGR_STAMP
  if (lane == 0 && ug == 0 && bg < 2 && jb.cls_cluster0 == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(status + 16) + (bg * 8 + wave) * 8;
    dbg[0] = st_mfma; dbg[1] = st_cell; dbg[2] = st_gather; dbg[3] = st_bar; dbg[4] = st_passes;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Two batch groups per workgroup, software-pipelined, with dedicated gather waves.
// The workgroup keeps ONE copy of its U^T fragments (compute waves 0-3, one tile each) and alternates between batch
// groups A and B of the same direction.  Waves 4-7 never compute and never store to global memory: while the compute
// waves run group A's MFMA chain + cell update, the gather waves poll and fetch group B's h_{t-1} from the cluster's
// exchange slot into B's next LDS image, and vice versa; one barrier per half-step joins the two roles.  The hand-off
// flight (write-through store -> sc1 load, ~2-4k cycles under load) is thereby hidden behind the other group's MFMA
// work, and - because the gather waves issue no stores - their s_waitcnt never waits on a store acknowledgement (in
// the one-group kernel the compiler's vmcnt(0) in front of the gathered data also drains the wave's own sc1 stores).
// Cost: 2 x 16 samples per workgroup, i.e. half as many CUs per job; the MFMA pipe becomes the bound.
template <int KS>
__device__ __forceinline__ void cluster_run2(const ClusterJob& jb, int wg, float* smem, unsigned* status) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256;
  constexpr int NGW = 3;   // gather waves (5, 6, 7); wave 4 publishes
  constexpr int NF = 11;   // gather loads in flight per gather wave
  static_assert(QN <= NGW * NF, "gather covers at most 33 image blocks");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0-3 compute, 4 publish, 5-7 gather
  const bool is_compute = wave < 4;
  const int j = lane & 15, uq = lane >> 4;
  const int G = jb.G_;
  const int pr = wg / G, ug = wg % G;   // pair index, unit group
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const float* __restrict__ Z = jb.Z;
  const int tile = ug * 4 + (wave & 3);
  const bool tvalid = is_compute && tile < KS;  // wave-uniform
  const int tl = tile < KS ? tile : 0;
  float uf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) uf[s] = tvalid ? jb.Up[(size_t)(4 * s + uq) * N + tl * 16 + j] : 0.f;
  float* img = smem;  // [group][2][IMG]
  for (int i = tid; i < 4 * IMG; i += 512) img[i] = 0.f;
  const int q0 = ug;  // own image block (4 tiles = 1 block)
  const int unit = tl * 4 + uq;
  const int idx_own = (((tl >> 2) * 4 + uq) * 16 + j) * 4 + (tl & 3);

  int bgi[2], bcl[2];
  bool gvalid[2], bvalid[2];
  __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    bgi[g] = pr * 2 + g;
    gvalid[g] = bgi[g] < jb.nbg;  // workgroup-uniform
    const int b = bgi[g] * 16 + j;
    bvalid[g] = gvalid[g] && b < B;
    bcl[g] = b < B ? b : B - 1;
    rs[g] = __builtin_amdgcn_make_buffer_rsrc(jb.xbuf + (size_t)(gvalid[g] ? bgi[g] : 0) * 2 * IMG, 0, 2 * IMG * 4, 0x00020000);
  }
  float c[2] = {0.f, 0.f};
  f32x4 zr[2][3];
  auto loadz = [&](f32x4& z, int g, int step) {
    if (tvalid && gvalid[g] && step < T) {
      const int t = reverse ? T - 1 - step : step;
      z = *reinterpret_cast<const f32x4*>(Z + ((size_t)bcl[g] * T + t) * N + unit * 4);
    }
  };
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int r = 0; r < 3; ++r) zr[g][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    loadz(zr[g][0], g, 0);
    loadz(zr[g][1], g, 1);
  }
  bool failed = false;
  __syncthreads();
#ifdef MGR_STAMP
  unsigned long long st2_comp = 0, st2_fin = 0, st2_bar = 0, st2_retry = 0;
#endif

  // ---- gather role: fetch group g's h_step into its next image; returns when every block carries the epoch parity
  auto gather = [&](int g, int step) {
    if (!(G > 1 && gvalid[g] && step >= 0 && step + 1 < T)) return;  // workgroup-uniform
    const int gw = wave - 5;
    const int slot = step & 1;
    const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
    float* hn = img + (g * 2 + ((step + 1) & 1)) * IMG;
    u32x4 gv[NF];
    unsigned pend = 0;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int q = gw + NGW * i;
      if (q < QN && q != q0) pend |= 1u << i;
    }
    unsigned spins = 0;
    // the peers publish this epoch right after the barrier we just left; polling earlier than the store->load flight
    // only adds fabric traffic that delays those very stores
    __builtin_amdgcn_s_sleep(48);
    while (pend && !failed) {
#pragma unroll
      for (int i = 0; i < NF; ++i)
        if (pend & (1u << i))
          gv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[g], (slot * IMG + (gw + NGW * i) * 256 + lane * 4) * 4, 0, 16);  // sc1
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        if (pend & (1u << i)) {
          const int q = gw + NGW * i;
          const int nvalid = KS - 4 * q;
          unsigned a = par ? 0xFFFFFFFFu : 0u;
          if (par) {
            a &= gv[i].x;
            if (nvalid > 1) a &= gv[i].y;
            if (nvalid > 2) a &= gv[i].z;
            if (nvalid > 3) a &= gv[i].w;
          } else {
            a |= gv[i].x;
            if (nvalid > 1) a |= gv[i].y;
            if (nvalid > 2) a |= gv[i].z;
            if (nvalid > 3) a |= gv[i].w;
          }
          if (__all((a & 1u) == par)) {
            *reinterpret_cast<u32x4*>(hn + q * 256 + lane * 4) = gv[i];
            pend &= ~(1u << i);
          }
        }
      }
      if (pend) {
#ifdef MGR_STAMP
        st2_retry += 1;
#endif
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if ((spins & 255u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) failed = true;
        if (spins > POLL_LIMIT) {
          failed = true;
          if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  };
  // ---- publisher role (wave 4): after the barrier that completes group g's own slice in LDS, copy the workgroup's
  // 1 KiB block to the exchange slot as eight whole 128-byte lines (one 16-byte sc1 store per lane).  Full-line
  // write-through keeps the memory side free of the 256 partial-line writes per step that 4-byte stores would cost.
  auto publish = [&](int g, int step) {
    if (!(G > 1 && gvalid[g] && step >= 0 && step + 1 < T) || q0 >= QN) return;
    const float* hn = img + (g * 2 + ((step + 1) & 1)) * IMG;
    u32x4 v = *reinterpret_cast<const u32x4*>(hn + q0 * 256 + lane * 4);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs[g], ((step & 1) * IMG + q0 * 256 + lane * 4) * 4, 0, 16);  // sc1
  };
  // ---- compute role: one time step of group g
  auto compute = [&](int g, int step, f32x4& zuse, f32x4& zload) {
    if (!gvalid[g]) return;  // workgroup-uniform
    const int t = reverse ? T - 1 - step : step;
    loadz(zload, g, step + 2);
    if (!tvalid) return;  // wave-uniform
    const float* hb = img + (g * 2 + (step & 1)) * IMG;
    float* hn = img + (g * 2 + ((step + 1) & 1)) * IMG;
    f32x4 acc = zuse, acc2 = {0.f, 0.f, 0.f, 0.f};
    {
      constexpr int PD = 3;
      f32x4 hbuf[4];
      const float* hlane = hb + (uq * 16 + j) * 4;
#pragma unroll
      for (int q = 0; q < PD && q < QN; ++q) hbuf[q] = *reinterpret_cast<const f32x4*>(hlane + q * 256);
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        if (q + PD < QN) hbuf[(q + PD) & 3] = *reinterpret_cast<const f32x4*>(hlane + (q + PD) * 256);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 hv = hbuf[q & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * q + r < KS) {
            if (r & 1)
              acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * q + r], hv[r], acc2, 0, 0, 0);
            else
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * q + r], hv[r], acc, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc += acc2;
    }
    float4 g4;
    float h = mgr_cell_fwd(acc[0], acc[1], acc[2], acc[3], c[g], g4);
    if (G > 1) {
      const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
      const unsigned hbits = (__float_as_uint(h) & ~1u) | par;
      h = __uint_as_float(hbits);   // published after the barrier by the publisher wave, as whole 128-byte lines
    }
    hn[idx_own] = h;
    if (bvalid[g]) {
      const size_t row = (size_t)(bgi[g] * 16 + j) * T + t;
      float yo = h;
      if (jb.R) yo += jb.R[row * jb.ldr + unit];
      jb.Y[row * jb.ldy + unit] = yo;
      if (jb.G) *reinterpret_cast<float4*>(jb.G + (row * H + unit) * 4) = g4;
      if (jb.Cs) jb.Cs[row * H + unit] = c[g];
    }
  };
  auto do_step = [&](int step, f32x4& zua, f32x4& zla, f32x4& zub, f32x4& zlb) {
#ifdef MGR_STAMP
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    if (is_compute)
      compute(0, step, zua, zla);     // A's step ...
    else if (wave == 4)
      publish(1, step - 1);           // ... B's h_{t-1} block (completed at the last barrier) goes out ...
    else
      gather(1, step - 1);            // ... and the peers' B blocks come in
#ifdef MGR_STAMP
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef MGR_STAMP
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
    if (is_compute)
      compute(1, step, zub, zlb);     // B's step ...
    else if (wave == 4)
      publish(0, step);
    else
      gather(0, step);                // ... while A's h_t arrives
#ifdef MGR_STAMP
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef MGR_STAMP
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    st2_comp += (t1 - t0) + (t3 - t2);
    st2_bar += (t2 - t1) + (t4 - t3);
#endif
  };
  for (int s0 = 0; s0 < T; s0 += 3) {
    do_step(s0, zr[0][0], zr[0][2], zr[1][0], zr[1][2]);
    if (s0 + 1 < T) do_step(s0 + 1, zr[0][1], zr[0][0], zr[1][1], zr[1][0]);
    if (s0 + 2 < T) do_step(s0 + 2, zr[0][2], zr[0][1], zr[1][2], zr[1][1]);
  }
#ifdef MGR_STAMP
  if (lane == 0 && wg < 2) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(status + 16) + (wg * 8 + wave) * 8;
    dbg[0] = st2_comp; dbg[1] = st2_comp; dbg[2] = st2_fin; dbg[3] = st2_bar; dbg[4] = st2_retry;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// K-split variant of the one-tile-per-wave cluster step (4 waves, 4 tiles = ONE 1 KiB image block per workgroup).
// Instead of gathering the whole h_{t-1} image into LDS, joining at a barrier and then letting every wave run the full
// K loop for its own tile, wave w here owns a QUARTER OF K for ALL FOUR tiles of the workgroup:
//   * it polls only the image blocks of its K range and takes them STRAIGHT INTO REGISTERS as MFMA B operands (the
//     block layout [kk][sample][r] is exactly the B fragment of four consecutive k-steps) - no LDS image, no B-operand
//     ds_reads under the MFMAs, no barrier between gather and MFMA; blocks that have arrived are consumed at once,
//     blocks still showing the previous epoch are polled again.  (A second register set, so that the re-poll is in
//     flight under the MFMAs of the arrived blocks, does not fit: 128 weight + 64 polling registers spill at 256.)
//   * the four partial sums per tile are exchanged through 12 KiB of LDS (double-buffered on the step parity: ONE
//     barrier per step) and wave w finishes tile w: adds Z_t, runs the cell, publishes h_t (same data-is-the-flag
//     parity words as cluster_run) and streams Y / gates / c out.
// Exchange bytes per workgroup and step are unchanged (each wave fetches a distinct quarter of the image).
template <int KS>
__device__ __forceinline__ void cluster_run_ks(const ClusterJob& jb, int bg, int ug, float* smem, unsigned* status, int dbg_mode) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256, NBW = (QN + 3) / 4;
  static_assert(NBW <= 8, "at most 8 image blocks per wave (H <= 512)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3
  const int j = lane & 15, uq = lane >> 4;
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  const float* __restrict__ Z = jb.Z;
  const float* __restrict__ Up = jb.Up;

  // K range of this wave: image blocks [qb, qb + nb)
  const int qb = wave * NBW;
  int nb = QN - qb;
  nb = nb < 0 ? 0 : (nb > NBW ? NBW : nb);
  nb = __builtin_amdgcn_readfirstlane(nb);

  // U^T fragments of the workgroup's four tiles for this wave's k-steps (zero where tile or k-step does not exist)
  float uf[4][NBW * 4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int gt = ug * 4 + tt;
#pragma unroll
    for (int sl = 0; sl < NBW * 4; ++sl) {
      const int s = qb * 4 + sl;
      uf[tt][sl] = (gt < KS && s < KS) ? Up[(size_t)(4 * s + uq) * N + gt * 16 + j] : 0.f;
    }
  }
  // the tile this wave finishes
  const int tile = ug * 4 + wave;
  const bool tvalid = tile < KS;  // wave-uniform
  const int tl = tvalid ? tile : 0;
  const int unit = tl * 4 + uq;

  float* red = smem;  // [2][src wave][tile][lane] f32x4
  float* xb = jb.xbuf + (size_t)bg * 2 * IMG;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * IMG * 4, 0x00020000);

  float c = 0.f;
  f32x4 zr0 = {0.f, 0.f, 0.f, 0.f}, zr1 = zr0, zr2 = zr0;
  auto loadz = [&](f32x4& z, int step) {
    if (step < T && tvalid) {
      const int t = reverse ? T - 1 - step : step;
      z = *reinterpret_cast<const f32x4*>(Z + ((size_t)bc * T + t) * N + (tl * 4 + uq) * 4);
    }
  };
  loadz(zr0, 0);
  loadz(zr1, 1);
  bool failed = false;
#ifdef MGR_ABLATE
  const int ab = dbg_mode >= 200 ? dbg_mode - 200 : 0;   // bit flags, see tools/ablate_ks.py
#endif
#ifdef MGR_STAMP
  unsigned long long ks_wait = 0, ks_mfma = 0, ks_red = 0, ks_cell = 0, ks_rounds = 0, ks_total = 0, ks_pre = 0, ks_t0 = 0;
#define KSTAMP(x) unsigned long long x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define KSTAMP(x)
#endif

  // ---- gather: the image blocks of this wave's K range go straight into registers.
  // The loads are issued from inline asm, so hipcc does not know that the registers have loads pending and inserts no
  // s_waitcnt in front of their readers; instead the wave POLLS THE REGISTERS: they are preset to a pattern no h word can
  // have (quiet-NaN exponent, wrong epoch parity) and an empty asm with "+v" constraints makes every iteration re-read
  // them.  A word that still shows the preset has not landed; a landed word with the previous epoch's parity means the
  // producer was late and the block is fetched again.  Nothing here waits on vmcnt, so the wave's own write-through
  // stores (whose acknowledgement takes longer than a load round trip) are never waited for.
  auto hidden_load = [&](u32x4& dst, const char* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(dst) : "v"(p) : "memory");
  };
  auto touch = [&](u32x4 (&v)[NBW]) {
    if constexpr (NBW == 8)
      asm volatile("s_sleep 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
    else if constexpr (NBW == 5)
      asm volatile("s_sleep 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4])::"memory");
    else if constexpr (NBW == 2)
      asm volatile("s_sleep 1" : "+v"(v[0]), "+v"(v[1])::"memory");
    else
      static_assert(NBW == 8 || NBW == 5 || NBW == 2, "add a touch() arm for this block count");
  };
  auto mfmas = [&](const u32x4 (&v)[NBW], f32x4 (&acc)[4]) {
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const float hv[4] = {__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)   // k-steps / blocks that do not exist carry zero weights
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[tt][i * 4 + r], hv[r], acc[tt], 0, 0, 0);
      }
    }
  };

  auto do_step = [&](int step, f32x4& zuse, f32x4& zload) {
    const int t = reverse ? T - 1 - step : step;
#ifdef MGR_STAMP
    {
      KSTAMP(b0);
      if (step > 0) ks_total += b0 - ks_t0;
      ks_t0 = b0;
    }
#endif
#ifdef MGR_ABLATE
    if (!(ab & 32))
#endif
    loadz(zload, step + 2);
    f32x4 acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef MGR_ABLATE
    if (step > 0 && nb > 0 && !failed && !(ab & 4)) {
#else
    if (step > 0 && nb > 0 && !failed) {
#endif
      const int slot = (step - 1) & 1;
      const unsigned par = ((((unsigned)(step - 1)) >> 1) & 1u) ^ 1u;
      const unsigned bad = 0x7FC00000u | (par ^ 1u);   // never a real h word (|h| < 1): bit 30 set, wrong parity
      const char* gp[NBW];
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        const int q = (i < nb) ? qb + i : QN - 1;        // unused slots re-read a valid block (their weights are zero)
        gp[i] = reinterpret_cast<const char*>(xb) + ((size_t)slot * IMG + q * 256 + lane * 4) * 4;
      }
      u32x4 v[NBW];
      unsigned rounds = 0, spins = 0;
      bool issue = true;
      for (;;) {
        if (issue) {
#pragma unroll
          for (int i = 0; i < NBW; ++i) v[i] = (u32x4){bad, bad, bad, bad};
#pragma unroll
          for (int i = 0; i < NBW; ++i) hidden_load(v[i], gp[i]);
          issue = false;
          spins = 0;
        }
        touch(v);
        unsigned a_and = v[0].x, a_or = v[0].x;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          a_and &= v[i].x & v[i].y & v[i].z & v[i].w;
          a_or |= v[i].x | v[i].y | v[i].z | v[i].w;
        }
        const bool lane_fresh = par ? (a_and & 1u) != 0u : (a_or & 1u) == 0u;
#ifdef MGR_ABLATE
        if ((ab & 1) && __all(((a_or >> 30) & 1u) == 0u)) break;   // timing ablation: take whatever landed
#endif
        if (__all(lane_fresh)) break;                 // every word shows this epoch (hence has landed)
        if (__all(((a_or >> 30) & 1u) == 0u)) {       // everything landed, something was still the previous epoch
          issue = true;
          ++rounds;
        } else if (++spins > 4096u) {                 // a load cannot take this long (~1 ms): drain and start over
          __builtin_amdgcn_s_waitcnt(0x0F70);
          issue = true;
          rounds += 64;                               // (so that this path, too, gives up after about a second)
        }
        if (issue) {
          if ((rounds & 63u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) failed = true;
          if (rounds > KS_ROUND_LIMIT) {
            failed = true;
            if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (failed) break;
        }
      }
#ifdef MGR_ABLATE
      if (!(ab & 2))
#endif
      mfmas(v, acc);
      // keep the polling registers allocated until here, then make sure no re-issued load is still in flight before this
      // wave publishes (a producer may overwrite the slot only after it has seen that publish)
      touch(v);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    // the four partial sums of every tile meet in LDS (all four go through it: selecting "my own" accumulator by the
    // run-time wave id would force the accumulators into scratch memory)
    float* rbuf = red + (step & 1) * (16 * 64 * 4);
    KSTAMP(r0);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(rbuf + ((tt * 4 + wave) * 64 + lane) * 4) = acc[tt];
    __syncthreads();
    KSTAMP(r1);
    if (!tvalid) {
      // padding tile of the last workgroup: its image words are still written every step (value 0, current parity), so
      // that consumers can test whole blocks without knowing which words exist
      if (step + 1 < T) {
        const unsigned par0 = (((unsigned)step >> 1) & 1u) ^ 1u;
        const int idx0 = (((tile >> 2) * 4 + uq) * 16 + j) * 4 + (tile & 3);
        __builtin_amdgcn_raw_buffer_store_b32(par0, rs, ((step & 1) * IMG + idx0) * 4, 0, 16);
      }
    } else {
      f32x4 tot = zuse;
      const float* mine = rbuf + (wave * 4 * 64 + lane) * 4;   // [tile = wave][src][lane]
#pragma unroll
      for (int src = 0; src < 4; ++src) tot += *reinterpret_cast<const f32x4*>(mine + src * 64 * 4);
      float4 g4;
      float h = mgr_cell_fwd(tot[0], tot[1], tot[2], tot[3], c, g4);
      const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
      const unsigned hbits = (__float_as_uint(h) & ~1u) | par;  // epoch parity rides in the mantissa LSB
      h = __uint_as_float(hbits);
#ifdef MGR_ABLATE
      if (step + 1 < T && !(ab & 8)) {
#else
      if (step + 1 < T) {
#endif
        // unit k = 4*tile + uq -> image [q = tile>>2][kk = uq][j][r = tile&3]
        const int idx = (((tile >> 2) * 4 + uq) * 16 + j) * 4 + (tile & 3);
        __builtin_amdgcn_raw_buffer_store_b32(hbits, rs, ((step & 1) * IMG + idx) * 4, 0, 16);  // sc1 write-through
      }
#ifdef MGR_ABLATE
      if (bvalid && !(ab & 16)) {
#else
      if (bvalid) {
#endif
        size_t row = (size_t)b * T + t;
        float yo = h;
        if (jb.R) yo += jb.R[row * jb.ldr + unit];
        jb.Y[row * jb.ldy + unit] = yo;
        if (jb.G) *reinterpret_cast<float4*>(jb.G + (row * H + unit) * 4) = g4;
        if (jb.Cs) jb.Cs[row * H + unit] = c;
      }
    }
#ifdef MGR_STAMP
    {
      KSTAMP(r2);
      ks_red += r1 - r0;
      ks_cell += r2 - r1;
    }
#endif
  };

  for (int s0 = 0; s0 < T; s0 += 3) {
    do_step(s0, zr0, zr2);
    if (s0 + 1 < T) do_step(s0 + 1, zr1, zr0);
    if (s0 + 2 < T) do_step(s0 + 2, zr2, zr1);
  }
#ifdef MGR_STAMP
  if (lane == 0 && ug == 0 && bg < 2 && jb.cls_cluster0 == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(status + 16) + (bg * 8 + wave) * 8;
    dbg[0] = ks_mfma; dbg[1] = ks_cell; dbg[2] = ks_wait; dbg[3] = ks_red; dbg[4] = ks_rounds; dbg[5] = ks_total; dbg[6] = ks_pre;
  }
#endif
}

#define CLKS_FOREACH(X) X(125) X(75) X(32) X(25)

#define CL2_FOREACH(X) X(125) X(75) X(32) X(25) X(8)

#define CL_FOREACH(X) \
  X(125, 1) X(75, 1) X(75, 2) X(32, 1) X(32, 2) X(32, 4) X(25, 1) X(25, 2) X(25, 4) X(16, 1) X(16, 2) X(8, 1) X(8, 2) \
  X(4, 1) X(3, 1) X(2, 1) X(1, 1)

__global__ __launch_bounds__(CL_WAVES * 64) void k_scan_cluster(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  // class-interleaved mapping: (w - cls_begin) % cls_nclusters = cluster within the class, / = unit group
  for (int k = 0; k < L.njobs; ++k) {
    const ClusterJob& jb = L.job[k];
    const int w = bid - jb.cls_begin;
    if (w < 0 || w >= jb.cls_nclusters * jb.G_) continue;
    // members of a cluster are CONTIGUOUS workgroup ids by default (the round-robin dispatcher then spreads them over
    // all XCDs, which measured best for the write-through exchange); the XCD-local experiment interleaves them instead
    const int cl = L.xcd_local ? w % jb.cls_nclusters : w / jb.G_;
    const int ug = L.xcd_local ? w / jb.cls_nclusters : w % jb.G_;
    const int bg = cl - jb.cls_cluster0;
    if (bg < 0 || bg >= jb.nbg) continue;
#define CL_CASE(KS, TPW) \
  if (jb.ks == KS && jb.tpw == TPW) { cluster_run<KS, TPW>(jb, bg, ug, cl, L.xcc, L.xcd_local, L.gather_delay, smem, L.status); return mgr_cluster_exit(L.status, L.sticky); }
    CL_FOREACH(CL_CASE)
#undef CL_CASE
    return;
  }
}

// K-split step: every job of the launch is a one-tile-per-wave, 4-wave cluster with an exchange (two workgroups per CU)
__global__ __launch_bounds__(256, 2) void k_scan_cluster_ks(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  for (int k = 0; k < L.njobs; ++k) {
    const ClusterJob& jb = L.job[k];
    const int w = bid - jb.cls_begin;
    if (w < 0 || w >= jb.cls_nclusters * jb.G_) continue;
    const int cl = w / jb.G_, ug = w % jb.G_;   // members of a cluster are contiguous workgroup ids
    const int bg = cl - jb.cls_cluster0;
    if (bg < 0 || bg >= jb.nbg) continue;
#define CLKS_CASE(KS) \
  if (jb.ks == KS) { cluster_run_ks<KS>(jb, bg, ug, smem, L.status, L.gather_delay); return mgr_cluster_exit(L.status, L.sticky); }
    CLKS_FOREACH(CLKS_CASE)
#undef CLKS_CASE
    return;
  }
}

// pair mode: 4 compute + 4 gather waves, one workgroup per CU
__global__ __launch_bounds__(512) void k_scan_cluster2(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  int ji = 0;
  for (int k = 1; k < L.njobs; ++k)
    if (bid >= L.job[k].wg_begin) ji = k;
  const ClusterJob& jb = L.job[ji];
  const int wg = bid - jb.wg_begin;
  if (wg >= jb.G_ * ((jb.nbg + 1) / 2)) return;
#define CL2_CASE(KS) \
  if (jb.ks == KS) { cluster_run2<KS>(jb, wg, smem, L.status); return mgr_cluster_exit(L.status, L.sticky); }
  CL2_FOREACH(CL2_CASE)
#undef CL2_CASE
}

}  // namespace

bool mgr_cluster_pair_supported(int ks) {
#define CL2_CASE(KS) \
  if (ks == KS) return true;
  CL2_FOREACH(CL2_CASE)
#undef CL2_CASE
  return false;
}

bool mgr_cluster_supported(int ks, int tpw) {
#define CL_CASE(KS, TPW) \
  if (ks == KS && tpw == TPW) return true;
  CL_FOREACH(CL_CASE)
#undef CL_CASE
  return false;
}

int mgr_cluster_launch(mgr_ctx* c, const ClusterLaunch& L, int total_wgs, bool any_exchange) {
  int maxks = 0, maxnw = 0;
  for (int i = 0; i < L.njobs; ++i) {
    maxks = L.job[i].ks > maxks ? L.job[i].ks : maxks;
    maxnw = L.job[i].nw > maxnw ? L.job[i].nw : maxnw;
  }
  const int waves = maxnw <= 4 ? 4 : CL_WAVES;
  size_t lds = 0;
  for (int i = 0; i < L.njobs; ++i) {
    size_t img = (size_t)((L.job[i].ks + 3) / 4) * 256 * sizeof(float);
    size_t need = (L.job[i].pair == 2 ? 4 : 2) * img;
    lds = need > lds ? need : lds;
  }

  int per_cu = 1;
  if (any_exchange) {
    // co-residency of every spinning workgroup is what makes the in-launch hand-off deadlock-free.  4-wave workgroups
    // with <= 80 KiB of LDS fit two per CU (8 waves, <= 256 VGPRs each); otherwise force one per CU through the LDS size.
    if (waves == 4 && lds <= 80 * 1024 && L.job[0].pair != 2) {
      per_cu = 2;
    } else if (lds < 84 * 1024) {
      lds = 84 * 1024;
    }
    MGR_REQUIRE(total_wgs <= per_cu * c->cu_count, "cluster scan needs %d co-resident workgroups but the device holds %d",
                total_wgs, per_cu * c->cu_count);
  }
  if (!(c->attr_done & 1u)) {   // (function attributes are per device, hence per context)
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_ks), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->attr_done |= 1u;
  }
  bool ks_all = L.ksplit && any_exchange && !L.xcd_local && waves == 4;
  for (int i = 0; i < L.njobs && ks_all; ++i) {
    const ClusterJob& j = L.job[i];
    bool inst = false;
#define CLKS_CASE(KS) \
  if (j.ks == KS) inst = true;
    CLKS_FOREACH(CLKS_CASE)
#undef CLKS_CASE
    ks_all = inst && j.G_ > 1 && j.tpw == 1 && j.nw == 4 && j.pair != 2;
  }
  bool pair = L.njobs > 0 && L.job[0].pair == 2;
  for (int i = 0; i < L.njobs; ++i) MGR_REQUIRE((L.job[i].pair == 2) == pair, "paired and unpaired jobs cannot share a launch");
  if (ks_all) {
    // partial-sum exchange only (no h image); the size still keeps at most two of these workgroups on a CU
    size_t lds_ks = 2 * 16 * 64 * 4 * sizeof(float);
    hipLaunchKernelGGL(k_scan_cluster_ks, dim3(total_wgs), dim3(256), lds_ks, mgr_stream(c), L);
  } else if (pair)
    hipLaunchKernelGGL(k_scan_cluster2, dim3(total_wgs), dim3(512), lds, mgr_stream(c), L);
  else
    hipLaunchKernelGGL(k_scan_cluster, dim3(total_wgs), dim3(waves * 64), lds, mgr_stream(c), L);
  MGR_LAUNCH_CHECK();
  return 0;
}
