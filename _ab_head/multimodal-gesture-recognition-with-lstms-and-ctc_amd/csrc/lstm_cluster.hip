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
// hipMalloc memory; slots are zeroed by a memset node ahead of every launch.  Every spin is bounded; a give-up sets
// status[0] and the host reports an error instead of hanging the GPU.
//
// Several layer-directions ("jobs": audio fwd/rev, skeletal fwd/rev) share ONE launch so that all spinning workgroups
// are co-resident by construction (grid <= workgroup slots of the chip); launches on DIFFERENT streams are admitted by
// lstm.hip::mgr_persist_admit, which serialises a launch that would not fit beside the persistent launches in flight.
// Every workgroup counts itself in at start (mgr_cluster_enter); the last arrival publishes the launch as resident, which is
// what mgr_stream_wait_next_resident lets another stream wait for before it sends chip-filling GEMMs.
#include <algorithm>
#include <type_traits>

#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CL_WAVES = 8;
constexpr unsigned POLL_LIMIT = 1u << 20;

template <int KS, int TPW>
__device__ __forceinline__ void cluster_run(const ClusterJob& jb, int bg, int ug, float* smem, unsigned* status) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256;
  static_assert(QN <= 32, "gather sweep covers at most 32 image blocks (H <= 512)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int nwv = blockDim.x >> 6;  // waves in this workgroup: 8, or 4 when every job runs one tile per SIMD
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches
  const int j = lane & 15, uq = lane >> 4;
  const int G = jb.G_;
  const int nw = jb.nw;
  const int tpwg = nw * TPW;  // tiles per workgroup, a multiple of 4
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
          if (step + 1 < T) __builtin_amdgcn_raw_buffer_store_b32(hbits, rs, (slot * IMG + idx) * 4, 0, 16);  // sc1 write-through
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
    if (G > 1 && step + 1 < T) {
      // gather: wave w sweeps blocks w, w+nwv, w+2*nwv, ... of the exchange slot (up to 8 loads in flight per round) until
      // every word of a block shows this epoch's parity
      constexpr int NF = 8;  // loads in flight per wave and round
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
              v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * IMG + (base + wave + nwv * i) * 256 + lane * 4) * 4, 0, 16);  // sc1
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
          if (pend) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 255u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) failed = true;
            if (spins > POLL_LIMIT) {
              failed = true;
              if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
    }
    __syncthreads();  // next image complete (own slice + gathered peers); everyone is done reading the current one
    cur ^= 1;
  };

  for (int s0 = 0; s0 < T; s0 += 3) {
    do_step(s0, zr0, zr2);
    if (s0 + 1 < T) do_step(s0 + 1, zr1, zr0);
    if (s0 + 2 < T) do_step(s0 + 2, zr2, zr1);
  }
}
// ---------------------------------------------------------------------------------------------------------------
// K-split variant of the one-tile-per-wave cluster step (4 waves, 4 tiles = ONE 1 KiB image block per workgroup; the default
// for every cluster with an exchange that the planner gives one tile per wave).
// Instead of gathering the whole h_{t-1} image into LDS, joining at a barrier and then letting every wave run the full
// K loop for its own tile, wave w here owns a QUARTER OF K for ALL FOUR tiles of the workgroup:
//   * it fetches only the image blocks of its K range and takes them STRAIGHT INTO REGISTERS as MFMA B operands (the
//     block layout [kk][sample][r] is exactly the B fragment of four consecutive k-steps) - no LDS image, no B-operand
//     ds_reads under the MFMAs, no barrier between gather and MFMA.  The loads are ordinary sc1 buffer loads that hipcc
//     sees and waits for; every word is validated by its epoch parity (the data is the flag), and a wave that finds a
//     word of the previous epoch fetches its blocks again.
//   * the four partial sums per tile are exchanged through 16 KiB of LDS (double-buffered on the step parity: ONE
//     barrier per step), then every wave finishes a quarter of the workgroup's 16 x 16 (unit, sample) cells: adds Z_t,
//     runs the cell, publishes h_t (same parity words as cluster_run) and streams Y / gates / c out.
//
// Which hidden unit sits in which MFMA slot is this kernel's private choice (U rows / columns, Z, Y, gates and c are
// addressed through it; nothing outside sees it).  Block q of the image holds the nv = min(4, KS - 4q) tiles 4q .. 4q+nv-1;
//   slot (tile 4q + r, unit-in-tile u)  <->  hidden unit 16q + nv*u + r
// and the finishing lane (r = lane>>4, sample j = lane&15) of wave u owns exactly that slot.  With this order
//   * a wave's 64 h words of one step are the 256 CONTIGUOUS bytes [q][kk = u][j][r] of the image: after one ds_bpermute the
//     wave publishes them as ONE coalesced store instruction = two whole 128-byte lines (the identity order makes every wave
//     write one dword of every 16-byte chunk of the block: 32 quarter-filled line writes per workgroup and step, and a
//     reader that sees a line between two of them fetches again);
//   * the four lanes r = 0..3 of a sample hold four CONSECUTIVE units: Y / gate / c stores stay coalesced.
//
// What keeps the dependent chain of a step short (round 3; the register-polling form it replaces - gather loads hidden from
// hipcc in inline asm, destination registers polled - was 3.5 / 5.1 us per step at H = 500 alone / with the skeletal clusters
// beside it, this form 2.9 / 4.6, and it needs no check of the generated assembly):
//   * NO load in the time loop that hipcc can see EXCEPT the gather itself.  hipcc merges its wait-count scoreboard
//     conservatively across the time loop: any load that may still be pending at the loop head (a Z prefetch, a status poll,
//     the weight loads of the prologue) turns into an s_waitcnt vmcnt(0) in front of the MFMA chain and at the top of every
//     step, and vmcnt(0) also waits for the acknowledgement of the wave's own stores of the step before and for the prefetch
//     from HBM.  So: the weight loads are retired by a wait hipcc can see before the loop; the status word is read through an
//     opaque asm (waited for on the spot, rare path); and Z_t / R_t arrive by LDS-DMA (global_load_lds: no register
//     destination, nothing for the compiler to track) through 2-deep per-wave LDS rings, one step ahead, with an explicit
//     counted wait where they are read.
//   * memory operations complete in issue order: a prefetch from HBM issued in FRONT of the gather holds the gathered blocks
//     (L2 hits) back for the length of its miss.  The prefetch of step t+1 is issued from inside the MFMA chain of step t (the
//     matrix pipe is busy anyway), behind the gather.
//   * XCD-local clusters publish with plain stores (lstm_cluster.h), whose acknowledgement comes from the local L2: the
//     vmcnt(0) of the next gather, which covers them, costs next to nothing.
// Measured and not kept (profiles/r03_scan_*): hint flags (every publishing wave also stores its epoch; a consumer polls the 32
// flags of its producers and fetches the payload once) - the extra round trip costs more than the re-fetches it saves (3.6 / 5.0);
// LDS-DMA landing zones for the payload, polled with ds_reads - presetting the zones, issuing eight DMA instructions (~150
// cycles each) and the LDS traffic put 1.4k cycles in front of every gather (4.5 / 5.6); re-fetching only the stale blocks
// (4.1 / 5.3); a raised wave priority for the cell phase, or for the wider layer's waves (no change).
typedef __attribute__((address_space(3))) float lds_float;
constexpr int KS_STG = 8;                      // steps per staged chunk of the transposed output
typedef _Float16 yt_f16x8 __attribute__((ext_vector_type(8)));
// A finished chunk of 8 time steps of one (sample, unit) row of the transposed output leaves the staging tile: as 8 floats, or - split
// row format, mgr.h - as 8 f16 hi values into the row's first half and 8 f16 lo values into its second half (16 + 16 bytes either way).
// row: the row's first float; t8: first time step of the chunk; ldt: row length in floats.
__device__ __forceinline__ void ks_flush_chunk(const float* stg, int lane, float* row, int t8, int ldt, bool split) {
  float v[KS_STG];
#pragma unroll
  for (int i = 0; i < KS_STG; ++i) v[i] = stg[i * 64 + lane];
  if (!split) {
    *reinterpret_cast<f32x4*>(row + t8) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(row + t8 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  } else {
    yt_f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < KS_STG; ++i) {
      float xs = v[i] * 8192.f;
      asm volatile("" : "+v"(xs));   // (hi and the residual from ONE f32 value: gemm.hip, mgr_split_f16)
      hi[i] = (_Float16)xs;
      lo[i] = (_Float16)(xs - (float)hi[i]);
    }
    _Float16* r16 = reinterpret_cast<_Float16*>(row);
    *reinterpret_cast<yt_f16x8*>(r16 + t8) = hi;
    *reinterpret_cast<yt_f16x8*>(r16 + ldt + t8) = lo;
  }
}
// zeros behind T up to the row length (both halves of a split row: 2 ldt f16 = ldt floats of zero bits)
__device__ __forceinline__ void ks_zero_tail(float* row, int T, int ldt, bool split) {
  const int t0 = (T + KS_STG - 1) / KS_STG * KS_STG;
  if (!split) {
    for (int t = t0; t + 4 <= ldt; t += 4) *reinterpret_cast<f32x4*>(row + t) = (f32x4){0.f, 0.f, 0.f, 0.f};
  } else {
    _Float16* r16 = reinterpret_cast<_Float16*>(row);
    for (int t = t0; t + 8 <= ldt; t += 8) {
      *reinterpret_cast<f32x4*>(r16 + t) = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(r16 + ldt + t) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
}
constexpr unsigned KS_ROUND_LIMIT = 1u << 20;  // re-fetch rounds of one wave before it gives up (~1 s)

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

// LDS of one workgroup (floats): partial sums [2][16][64] f32x4 | 4 staging tiles [8][64] of the transposed output | 4 Z rings
// [2][64] f32x4 | 4 residual rings [2][64]
constexpr int KS_LDS_FLOATS = 2 * 16 * 64 * 4 + 4 * KS_STG * 64 + 4 * 2 * 256 + 4 * 2 * 64;

template <int KS>
__device__ __forceinline__ void cluster_run_ks(const ClusterJob& jb, const ClusterCommon& cm, int bg, int ug, float* smem, bool fast) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256, NBW = (QN + 3) / 4;
  static_assert(NBW >= 1 && NBW <= 8, "1..8 image blocks per wave (H <= 512)");
  unsigned* status = cm.status;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3
  const int j = lane & 15, uq = lane >> 4;
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  const float* __restrict__ Z = jb.Z;
  const float* __restrict__ Up = jb.Up;
  // Job fields the time loop uses, as values hipcc cannot re-derive from the kernel argument: left to itself it re-loads them from
  // the argument segment inside the loop (s_load + s_waitcnt lgkmcnt(0), one of them right behind the barrier, on the critical path)
  const float* Rp = jb.R;
  float *Yp = jb.Y, *Gp = jb.G, *Csp = jb.Cs;
  int ldr = jb.ldr, ldy = jb.ldy;
  asm volatile("" : "+s"(Rp), "+s"(Yp), "+s"(Gp), "+s"(Csp), "+s"(ldr), "+s"(ldy));

  auto unit_of = [](int tile, int u) {   // hidden unit of MFMA slot (tile, unit-in-tile): see cluster_run_ks
    const int q = tile >> 2, nv = (KS - 4 * q) < 4 ? (KS - 4 * q) : 4;
    return 16 * q + nv * u + (tile & 3);
  };
  const int qb = wave * NBW;     // K range of this wave: image blocks [qb, qb + nb) = what unit groups qb .. qb + nb - 1 publish
  int nb = QN - qb;
  nb = nb < 0 ? 0 : (nb > NBW ? NBW : nb);
  nb = __builtin_amdgcn_readfirstlane(nb);

  float uf[4][NBW * 4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int gt = ug * 4 + tt;
#pragma unroll
    for (int sl = 0; sl < NBW * 4; ++sl) {
      const int s = qb * 4 + sl;
      uf[tt][sl] = (gt < KS && s < KS) ? Up[(size_t)unit_of(s, uq) * N + unit_of(gt, j >> 2) * 4 + (j & 3)] : 0.f;
    }
  }
  const int ftile = ug * 4 + uq;     // the cell this lane finishes: slot (tile 4*ug + uq, unit-in-tile wave)
  const bool cvalid = ftile < KS;
  const int unit = cvalid ? unit_of(ftile, wave) : 0;
  const int red_off = ((uq * 4) * 64 + wave * 16 + j) * 4;

  float* red = smem;                                               // [2][tile][src wave][lane] f32x4
  float* stg = smem + 2 * 16 * 64 * 4 + wave * (KS_STG * 64);      // [KS_STG][64 lanes]
  float* zring = smem + 2 * 16 * 64 * 4 + 4 * KS_STG * 64 + wave * (2 * 256);                  // [2][64] f32x4
  float* rring = smem + 2 * 16 * 64 * 4 + 4 * KS_STG * 64 + 4 * 2 * 256 + wave * (2 * 64);    // [2][64]
  const unsigned zring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)zring);
  const unsigned rring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)rring);
  float* ytrow = nullptr;
  const bool yt_split = jb.yt_split != 0;
  if (jb.YT && cvalid && bvalid) {
    ytrow = jb.YT + (size_t)b * jb.ytb + (size_t)unit * jb.ldt;
#pragma unroll
    for (int i = 0; i < KS_STG; ++i) stg[i * 64 + lane] = 0.f;
  }
  float* xb = jb.xbuf + (size_t)bg * 2 * IMG;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * IMG * 4, 0x00020000);
  // Z_t (and the residual input R_t) of this lane's cell come through 2-deep per-wave LDS rings filled by LDS-DMA one step
  // ahead.  Why not a plain load: hipcc puts an s_waitcnt vmcnt(0) for ANY pending load it can see in front of the MFMA chain
  // and at the loop head (it merges its scoreboard conservatively across the time loop), so a visible prefetch from HBM is
  // waited for in full, on the critical path, right after it is issued.  A DMA has no register destination - nothing for the
  // compiler to wait for - and its landing is covered by the wait of the NEXT step's gather (memory operations complete in
  // issue order).  Byte offsets of the lane within Z / R: the launcher admits only tensors below 4 GiB.
  const unsigned zvoff = (unsigned)(((size_t)bc * T * N + (size_t)unit * 4) * sizeof(float));
  const unsigned rvoff = Rp ? (unsigned)(((size_t)bc * T * ldr + unit) * sizeof(float)) : 0u;
  auto prefetch = [&](int step) {   // (everything wave-uniform except the lane offsets)
    if (step < T) {
      const int t = reverse ? T - 1 - step : step;
      mgr_dma_b128(Z + (size_t)t * N, zvoff, zring_lds + (step & 1) * 1024);
      if (Rp) mgr_dma_b32(Rp + (size_t)t * ldr, rvoff, rring_lds + (step & 1) * 256);
    }
  };
  prefetch(0);
  // (a wait hipcc can see: with the weight loads retired before the time loop its scoreboard enters the loop empty; otherwise the
  // loop-head merge keeps them "maybe pending" and every MFMA chain gets an s_waitcnt vmcnt(0) in front - behind the Z prefetch)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

  float c = 0.f;
  bool nonfinite = false;
  bool failed = false;
  unsigned rounds = 0;        // re-fetches / re-polls of the whole launch: the bound of every spin below
  auto tick = [&]() {         // a wasted round: look at the launch's give-up word now and then, give up after ~1 s of them
    ++rounds;
    if ((rounds & 255u) == 0) {
      // (through an opaque asm, waited for on the spot: a load hipcc can see inside the polling loops leaves a "maybe pending"
      // register in its scoreboard, and it then puts an s_waitcnt vmcnt(0) in front of the MFMA chain - behind the Z prefetch)
      unsigned st;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st) : "v"(status) : "memory");
      if (__builtin_amdgcn_readfirstlane(st) != 0u) failed = true;
    }
    if (rounds > KS_ROUND_LIMIT) {
      failed = true;
      if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    // (no per-step zeroing of the 32 gather registers and the 16 accumulators: vector instructions do not overlap with this
    // SIMD's MFMAs - profiles/r04_single_cu_probes.txt - so every v_mov of a step is step time; the first MFMA of each
    // accumulator takes a literal zero instead)
    f32x4 acc[4];
    u32x4 v[NBW];
    const bool gather = step > 0 && nb > 0 && !failed;
    if (gather) {
      const int slot = (step - 1) & 1;
      const unsigned par = ((((unsigned)(step - 1)) >> 1) & 1u) ^ 1u;
      // nb blocks of 1 KiB straight into registers (the block layout [kk][sample][r] IS the B fragment of four k-steps); every
      // word is validated by its epoch parity - the data is the flag - and what still shows the previous epoch is fetched again
      for (;;) {
#pragma unroll
        for (int i = 0; i < NBW; ++i)   // (blocks beyond nb - wave-uniform - re-read a valid block: they meet zero weights)
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * IMG + (qb + (i < nb ? i : 0)) * 256 + lane * 4) * 4, 0, 16);   // sc1
        unsigned a_and = 0xFFFFFFFFu, a_or = 0u;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          a_and &= v[i].x & v[i].y & v[i].z & v[i].w;
          a_or |= v[i].x | v[i].y | v[i].z | v[i].w;
        }
        const bool lane_fresh = par ? (a_and & 1u) != 0u : (a_or & 1u) == 0u;
        if (__all(lane_fresh) || failed) break;
        tick();
        if (failed) break;
      }
    }
    // Z / R of the NEXT step leave BEHIND the gather (an HBM miss in front of it would hold the gathered blocks - L2 hits - back
    // for the length of the miss: memory operations complete in issue order), from inside the MFMA chain: the matrix pipe is
    // busy anyway, the eight scalar / vector-memory instructions of the two DMAs cost nothing there
    if (gather && !failed) {
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        if (i == (NBW > 1 ? 1 : 0)) prefetch(step + 1);
        const float hv[4] = {__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {   // k-steps / blocks that do not exist carry zero weights
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[tt][i * 4 + r], hv[r], (i == 0 && r == 0) ? zero : acc[tt], 0, 0, 0);
          }
        }
      }
    } else {   // the first step (h_{-1} = 0), a wave without K range, a launch that gave up
      prefetch(step + 1);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the four partial sums of every tile meet in LDS (double-buffered on the step parity: one barrier per step)
    float* rbuf = red + (step & 1) * (16 * 64 * 4);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(rbuf + ((tt * 4 + wave) * 64 + lane) * 4) = acc[tt];
    // Z_t / R_t were fetched one step ago; the only vector-memory operations this wave has issued since that may still be in
    // flight are the one or two DMAs of step t + 1: a counted wait makes their landing explicit (in practice it never waits)
    if (Rp)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    const f32x4 zt = *reinterpret_cast<const f32x4*>(zring + (step & 1) * 256 + lane * 4);
    const float rt = Rp ? rring[(step & 1) * 64 + lane] : 0.f;
    __syncthreads();
    const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
    unsigned hbits = par;   // cells of a padding tile: value 0 with the current parity, so that consumers can test whole blocks
    float h = 0.f, yv = 0.f;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cvalid) {
      f32x4 tot = zt;   // (same summation order as cluster_run_ks: the K-split steps are bit-identical)
      const float* mine = rbuf + red_off;
#pragma unroll
      for (int src = 0; src < 4; ++src) tot += *reinterpret_cast<const f32x4*>(mine + src * 64 * 4);
      h = mgr_cell_fwd(tot[0], tot[1], tot[2], tot[3], c, g4);
      if (!(fabsf(h) < 2.f) && !nonfinite) {
        // NaN / Inf (diverged weights, bad checkpoint): what is published - and fed back - stays finite (0), Y of this (sample,
        // unit) is NaN from here on (latched) and the launch raises MGR_SCAN_NONFINITE (mgr.h)
        __hip_atomic_fetch_or(cm.sticky, MGR_ST_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nonfinite = true;
      }
      if (nonfinite) {
        h = 0.f;
        c = 0.f;
      }
      hbits = (__float_as_uint(h) & ~1u) | par;  // epoch parity rides in the mantissa LSB
      h = __uint_as_float(hbits);
      yv = nonfinite ? __uint_as_float(0x7FC00000u) : h;
    }
    if (step + 1 < T) {
      // lane (r, j) holds image word j*4 + r of the wave's 64-word segment: bring word l to lane l, one coalesced store (storing
      // the words from the lanes that hold them - the same 256 bytes in permuted lane order - is slower: 3.01 against 2.93 us per
      // step at H = 500).  XCD-local clusters: a plain store into the L2 every peer's sc1 load is served from; else write-through
      const unsigned w = __builtin_amdgcn_ds_bpermute((((lane & 3) << 4) | (lane >> 2)) << 2, hbits);
      if (fast)
        __builtin_amdgcn_raw_buffer_store_b32(w, rs, ((step & 1) * IMG + (ug * 4 + wave) * 64 + lane) * 4, 0, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b32(w, rs, ((step & 1) * IMG + (ug * 4 + wave) * 64 + lane) * 4, 0, 16);  // sc1
    }
    if (cvalid && bvalid) {
      size_t row = (size_t)b * T + t;
      const float yo = yv + rt;
      // (the pinned pointers lost their address space: name it, or hipcc emits flat_store - counted on lgkmcnt as well)
      typedef __attribute__((address_space(1))) float gfloat;
      typedef __attribute__((address_space(1))) f32x4 gf32x4;
      ((gfloat*)Yp)[row * ldy + unit] = yo;
      if (Gp) *(gf32x4*)(Gp + (row * H + unit) * 4) = (f32x4){g4.x, g4.y, g4.z, g4.w};
      if (Csp) ((gfloat*)Csp)[row * H + unit] = c;
      if (ytrow) {
        stg[(t & (KS_STG - 1)) * 64 + lane] = yo;
        // the chunk [t & ~7, +8) is complete when the walk leaves it (all lanes of the launch agree on t)
        if (reverse ? (t & (KS_STG - 1)) == 0 : ((t & (KS_STG - 1)) == KS_STG - 1 || t == T - 1)) {
          ks_flush_chunk(stg, lane, ytrow, t & ~(KS_STG - 1), jb.ldt, yt_split);
#pragma unroll
          for (int i = 0; i < KS_STG; ++i) stg[i * 64 + lane] = 0.f;   // (a partial last chunk pads with zeros)
        }
      }
    }
  }
  if (nonfinite) mgr_mark_sample(cm, b);   // (latched: marked once, behind the time loop - inside it the call cost the step 0.1 us)
  if (ytrow) ks_zero_tail(ytrow, T, jb.ldt, yt_split);   // (mgr.h: the transposed copy is zero in [T, ldt))
}

// ---------------------------------------------------------------------------------------------------------------
// Split-f16 variant of the K-split step (round 4; the default - tune key 14 = 1 keeps the f32 MFMA step above).
//
// The recurrence h_{t-1} U is an f32 product; v_mfma_f32_16x16x4_f32 delivers it at 64 FLOP per cycle and SIMD, the f16 form
// v_mfma_f32_16x16x32_f16 at 1024.  Here every f32 operand x travels as TWO f16 values, x*s = hi + lo (s a power of two, hi =
// rn_f16(x*s), lo = rn_f16(x*s - hi): |x*s - hi - lo| <= 2^-23 |x*s|, 22+ significant bits), and the product is taken as
//     U h  ~  (Uhi hhi + Ulo hhi + Uhi hlo) / (sU sh)                      [three f16 MFMAs, ONE f32 accumulator]
// The dropped Ulo hlo term is 2^-22 of the product; f16 x f16 products are exact in the f32 accumulator.  The representation
// error (~2^-21.5 per product) is BELOW what an f32 dot product of this length loses to rounding in its accumulation
// (measured, K = 500: 2.0e-7 absolute against 1.5e-6 for an f32 sgemm, both against f64; tests/test_gpu_kernels.py holds the same
// bounds against the f64 oracle for both steps).  Scales: h in [-1, 1] -> sh = 2^15 (hi <= 32768 < 65504; a lo below the f16
// normal range is an absolute error <= 2^-14 / 2^15 = 2^-29); U -> sU = the power of two that puts the workgroup's largest
// |U| in [2^14, 2^15) (computed in the prologue from the slice the workgroup holds).  12 MFMAs of 16 cycles replace 128 of 32 per
// wave and step at H = 500: 0.3 instead of 1.8 us of matrix pipe per step.
//
// Same cluster geometry as cluster_run_ks (a workgroup = 16 hidden units = 4 tiles, wave w = a quarter of K for all four tiles,
// partial sums meet in LDS, ONE barrier per step, parity words as the flag), with these differences:
//   * K is walked in blocks of 32 units.  The exchange image of a slot is [K-block][hi | lo][1 KiB]; the 1 KiB of a part holds, for
//     B-operand lane (kg, n), the 16 bytes = 8 f16 = units 32 kb + 8 kg + 0..7 of sample n.  A block is produced by two workgroups
//     (16 units each, half `kg >> 1`), inside a half the 16-byte chunks are ordered [n >> 2][n & 3][kg & 1]:
//   * the FINISHING lanes are dealt so that wave u owns samples 4u .. 4u+3 and lane l the unit (l & 15) of the workgroup: the two
//     lanes of a unit pair swap their packed (hi, lo) word by DPP, the even lane keeps (hi_even, hi_odd), the odd lane (lo_even,
//     lo_odd), and ONE store instruction of the wave writes two whole 128-byte lines: the hi line and the lo line of its four samples.
//     No ds_bpermute.  (Z / Y / gate / c rows of a sample are read and written as 16 consecutive units.)
//   * every published word carries the epoch parity in bit 0: the last mantissa bit of the even unit's hi (moved to the nearest f16 with
//     that bit BEFORE lo is taken, so lo absorbs it) resp. of the even unit's lo (moved likewise): an even unit keeps
//     |h 2^15 - hi - lo| <= 2^-20 |h 2^15|, an odd unit 2^-22.
//   * Y, the saved gates and c are the f32 values; the recurrence sees h rounded to 22+ bits (as every f32 consumer of Y would
//     see it rounded to 24).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// the f16 nearest to x among those whose last mantissa bit is `lsb`, given bits = rn_f16(x): bits itself, or its neighbour on the
// side x lies on (the f16 bit patterns of one sign are ordered like their magnitudes, exponent boundaries included; x is finite
// and far below the f16 maximum here)
__device__ __forceinline__ unsigned k16_with_lsb(unsigned bits, unsigned lsb, float x, unsigned on) {   // on = 0: bits as they are
  // branch-free (the step is one dependent chain; a divergent branch costs it more than these six instructions): bits + 1 is the next
  // magnitude of the same sign, bits - 1 the previous one (never below zero: |x| >= 0 picks + 1 there; far below the f16 maximum)
  const unsigned need = (bits ^ lsb) & on;
  const float v = (float)__builtin_bit_cast(_Float16, (unsigned short)(bits & 0x7FFFu));
  return bits + (fabsf(x) >= v ? need : 0u - need);
}
// Partial sums in LDS (round 6: laid out by the bank rules of MI355X_MICROARCH.md, LDS - round 4's layout assumed 16 contiguous lanes
// and 64 banks for the stores as well, and SQ_LDS_BANK_CONFLICT counted ~280 extra cycles per CU and step):
//   * a tile's partial sums of one source wave are 64 cells of 16 bytes: cell (unit-in-tile uq, sample n) at slot 4 n + ((uq + (n >> 1)) & 3);
//   * ds_write_b128 is served in groups of 8 consecutive lanes over 32 banks (128 bytes): the writer lanes of a group hold uq fixed and
//     n = 8 g .. 8 g + 7, their slots mod 8 are 4 (n & 1) + ((uq + (n >> 1)) & 3): all eight different;
//   * ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27}, ... over 64 banks (256 bytes = 16 slots): a finishing group reads,
//     for sample n0, the units of tiles 0 and 3 and, for sample n0 + 1, those of tiles 1 and 2 - with tiles 264 slots apart (8 mod 16)
//     the four 4-slot blocks are (n0, n0 + 2, n0 + 3, n0 + 1) mod 4: all sixteen slots different.
constexpr int K16_TILE_SLOTS = 264;
constexpr int K16_TILE = K16_TILE_SLOTS * 4;   // floats of one tile's partial sums: 4 source waves x 64 cells x f32x4, + 8 cells of padding
__device__ __forceinline__ int k16_wslot(int om, int okg) { return om * 4 + ((okg + (om >> 1)) & 3); }
__device__ __forceinline__ int k16_rslot(int fu, int fn) { return (fu >> 2) * K16_TILE_SLOTS + fn * 4 + (((fu & 3) + (fn >> 1)) & 3); }
constexpr int K16_LDS_FLOATS = 2 * 4 * K16_TILE + 4 * KS_STG * 64 + 4 * 2 * 256 + 4 * 2 * 64 + 16;

#ifdef MGR_STAMP
__device__ unsigned long long g_stamps[64];
#define KSTAMP(i, dep) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); \
    st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
extern "C" int mgr_debug_stamps(unsigned long long* out) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
  unsigned long long z[64] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
  return 0;
}
#else
#define KSTAMP(i, dep) do { } while (0)
#endif
// FUSED (round 5, k_scan_cluster_k16f): the workgroup has 512 threads and runs TWO unit groups of one cluster - threads 0..255 the
// member 2 j, threads 256..511 the member 2 j + 1 - each through this function with its own half of the LDS; they share the CU and
// the barriers (the same count in both: one in the prologue, one per step), nothing else.
// SHARE_TG >= 0 (round 6, k_scan_cluster_k16fs; FUSED only): the two halves of the workgroup belong to the SAME cluster, so wave w of
// half 0 and wave w of half 1 need the same K-blocks of the same h image - and used to fetch and verify them twice, 64 KiB per CU and
// step through the L2.  Now half 0 fetches and verifies the first ceil(NBW / 2) K-blocks of the wave's range, half 1 the rest; both
// leave what they verified in a shared LDS image (xs: [wave][K-block][hi | lo][64 lanes] 16 bytes), ONE more workgroup barrier, and
// each reads the other's blocks from there: half the L2 gather traffic and half the verification chain per wave.  Same operands, same
// MFMA order: bit-identical.  The image is single-buffered: a wave writes step s + 1's blocks behind the step-s barrier, which its
// partner reaches only after its MFMAs consumed step s's.  A unit group beyond G (odd G) runs as a member WITHOUT valid cells - it
// still owes its partner half of the image.
template <int NBW, bool FUSED = false, int SHARE_TG = -1>   // K-blocks (of 32 units) per wave: H <= 128 * NBW
__device__ __forceinline__ void cluster_run_k16(const ClusterJob& jb, const ClusterCommon& cm, int bg, int ug, float* smem, bool fast,
                                                float* xs = nullptr) {
  static_assert(NBW >= 1 && NBW <= 4, "1..4 K-blocks per wave (H <= 512)");
  static_assert(SHARE_TG < 0 || FUSED, "the shared gather is a property of the fused form");
  constexpr bool SHARE = SHARE_TG >= 0;
  constexpr int NA = (NBW + 1) / 2;                       // K-blocks half 0 fetches; half 1: the other NBW - NA
  constexpr int MLO = !SHARE ? 0 : (SHARE_TG == 0 ? 0 : NA), MHI = !SHARE ? NBW : (SHARE_TG == 0 ? NA : NBW);   // this wave fetches [MLO, MHI)
  unsigned* status = cm.status;
  const int tid = FUSED ? (int)(threadIdx.x & 255u) : (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3
  const int H = jb.H, N = 4 * H, G = jb.G_;
  const int NKB = (H + 31) >> 5;          // K-blocks of the layer
  const int IMGB = NKB * 2048;            // bytes of one exchange slot
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  // operand roles (MFMA A / B / C lanes)
  const int om = lane & 15, okg = lane >> 4;
  // finishing role: cell (unit 16 ug + (lane & 15), sample 4 wave + (lane >> 4))
  const int fn = 4 * wave + (lane >> 4), fu = lane & 15;
  const int b = bg * 16 + fn;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  const int unit = 16 * ug + fu;
  const bool cvalid = unit < H;
  const float* __restrict__ Z = jb.Z;
  const float* __restrict__ Up = jb.Up;
  const float* Rp = jb.R;
  float *Yp = jb.Y, *Gp = jb.G, *Csp = jb.Cs;
  int ldr = jb.ldr, ldy = jb.ldy;
  asm volatile("" : "+s"(Rp), "+s"(Yp), "+s"(Gp), "+s"(Csp), "+s"(ldr), "+s"(ldy));   // (see cluster_run_ks)

  const int qb = wave * NBW;     // K range of this wave: K-blocks [qb, qb + nb)
  int nb = NKB - qb;
  nb = nb < 0 ? 0 : (nb > NBW ? NBW : nb);
  nb = __builtin_amdgcn_readfirstlane(nb);

  float* red = smem;                                                          // [2][tile] K16_TILE
  float* stg = smem + 2 * 4 * K16_TILE + wave * (KS_STG * 64);                // [KS_STG][64 lanes]
  float* zring = smem + 2 * 4 * K16_TILE + 4 * KS_STG * 64 + wave * (2 * 256);                 // [2][64] f32x4
  float* rring = smem + 2 * 4 * K16_TILE + 4 * KS_STG * 64 + 4 * 2 * 256 + wave * (2 * 64);   // [2][64]
  float* wmax = smem + 2 * 4 * K16_TILE + 4 * KS_STG * 64 + 4 * 2 * 256 + 4 * 2 * 64;         // [4]
  const unsigned zring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)zring);
  const unsigned rring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)rring);

  // ---- weights: the workgroup's slice of U as f16 (hi, lo) A fragments.  Lane (m = lane & 15, kg = lane >> 4) of tile tt, K-block kb
  // holds U[32 kb + 8 kg + e][gate column m of tile tt], e < 8.  Two passes over the slice: its largest magnitude, then the split.
  auto uval = [&](int tt, int i, int e) -> float {
    const int k = 32 * (qb + i) + 8 * okg + e, uu = 16 * ug + 4 * tt + (om >> 2);
    return (k < H && uu < H) ? Up[(size_t)k * N + uu * 4 + (om & 3)] : 0.f;
  };
  float umax = 0.f;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) umax = fmaxf(umax, fabsf(uval(tt, i, e)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
  if (lane == 0) wmax[wave] = umax;
  __syncthreads();
  umax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  int ex = 0;
  if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &ex);   // umax = m 2^ex, m in [0.5, 1)   (Inf / NaN weights: the products are
  ex = ex < -60 ? -60 : ex;                                    //  NaN, the non-finite guard below reports the launch)
  const float sU = ldexpf(1.f, 15 - ex);        // largest |U| sU in [2^14, 2^15)
  const float inv = ldexpf(1.f, ex - 30);       // 1 / (sU 2^15)
  f16x8 ah[4][NBW], al[4][NBW];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = uval(tt, i, e) * sU;
        asm volatile("" : "+v"(x));   // (hi and the residual from ONE f32 value: gemm.hip, mgr_split_f16)
        const _Float16 hi = (_Float16)x;
        ah[tt][i][e] = hi;
        al[tt][i][e] = (_Float16)(x - (float)hi);
      }

  float* ytrow = nullptr;
  const bool yt_split = jb.yt_split != 0;
  if (jb.YT && cvalid && bvalid) {
    ytrow = jb.YT + (size_t)b * jb.ytb + (size_t)unit * jb.ldt;
#pragma unroll
    for (int i = 0; i < KS_STG; ++i) stg[i * 64 + lane] = 0.f;
  }
  char* xb = reinterpret_cast<char*>(jb.xbuf) + (size_t)bg * 2 * IMGB;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * IMGB, 0x00020000);
  // gather: byte offset of this lane's 16-byte chunk in the hi part of its K-blocks (the lo part is 1 KiB further).  A half whose
  // workgroup does not exist (the last K-block of an odd G) or a K-block beyond nb re-reads a valid chunk: it meets zero weights.
  unsigned goff[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int kb = qb + (i < nb ? i : 0);
    const int half = (2 * kb + (okg >> 1) < G) ? (okg >> 1) : 0;
    goff[i] = (unsigned)(kb * 2048 + half * 512 + (om >> 2) * 128 + (om & 3) * 32 + (okg & 1) * 16);
  }
  // publish: the even lane of a unit pair stores the hi word, the odd lane the lo word
  const unsigned poff = (unsigned)((ug >> 1) * 2048 + (lane & 1) * 1024 + (ug & 1) * 512 + wave * 128 + (lane >> 4) * 32 + (fu >> 1) * 4);

  const int zunit = cvalid ? unit : 0;   // (padding cells fetch a valid address and ignore it)
  const unsigned zvoff = (unsigned)(((size_t)bc * T * N + (size_t)zunit * 4) * sizeof(float));
  const unsigned rvoff = Rp ? (unsigned)(((size_t)bc * T * ldr + zunit) * sizeof(float)) : 0u;
  auto prefetch = [&](int step) {   // (everything wave-uniform except the lane offsets; no branch on the last step: it re-fetches
    const int sc = step < T ? step : T - 1;   //  its own row into the ring slot nobody reads any more, retired behind the loop)
    const int t = reverse ? T - 1 - sc : sc;
    mgr_dma_b128(Z + (size_t)t * N, zvoff, zring_lds + (step & 1) * 1024);
    if (Rp) mgr_dma_b32(Rp + (size_t)t * ldr, rvoff, rring_lds + (step & 1) * 256);
  };
  prefetch(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the weight loads are retired before the time loop (see cluster_run_ks)

  // partial sums in LDS: cell (unit-in-tile uq, sample n) of a tile sits at slot n*4 + ((uq + (n >> 2)) & 3) of its source wave's 64,
  // (layout: K16_TILE above - the writes and the finishing reads are conflict-free)
  const int wslot = k16_wslot(om, okg);
  const int rslot = k16_rslot(fu, fn);

  float c = 0.f;
  bool nonfinite = false;
  bool failed = false;
  unsigned rounds = 0;
  // (rounds / failed are wave-uniform and hipcc must SEE that - readfirstlane: taken as divergent, the retry loop of the gather is an
  //  exec-masked region of a dozen s_cbranch_execz)
  auto tick = [&]() {
    rounds = (unsigned)__builtin_amdgcn_readfirstlane((int)(rounds + 1u));
    if ((rounds & 255u) == 0) {
      unsigned st;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st) : "v"(status) : "memory");
      if (__builtin_amdgcn_readfirstlane(st) != 0u) failed = true;
    }
    if (rounds > KS_ROUND_LIMIT) {
      failed = true;
      if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    failed = __builtin_amdgcn_readfirstlane((int)failed) != 0;
  };

#ifdef MGR_STAMP
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev));
#endif
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    f32x4 acc[4];
    u32x4 v[2 * NBW];
    const bool gather = step > 0 && nb > 0 && !failed;
    if (gather) {
      const unsigned sbase = (unsigned)(((step - 1) & 1) * IMGB);
      const unsigned par = ((((unsigned)(step - 1)) >> 1) & 1u) ^ 1u;
      for (;;) {
#pragma unroll
        for (int i = MLO; i < MHI; ++i) {
          v[2 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], sbase, 16);              // sc1
          v[2 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] + 1024u, sbase, 16);   // sc1
        }
        // (both chains, selected afterwards: a branch on the wave-uniform parity would save 16 of these 32 three-input operations, but
        //  hipcc then merges its wait-count scoreboard over a path that runs neither chain and waits for the gathered blocks - and with
        //  them for the Z prefetch issued in between - inside the MFMA chain: +240 cycles per step, measured)
        unsigned a_and = 0xFFFFFFFFu, a_or = 0u;
#pragma unroll
        for (int i = 2 * MLO; i < 2 * MHI; ++i) {
          a_and &= v[i].x & v[i].y & v[i].z & v[i].w;
          a_or |= v[i].x | v[i].y | v[i].z | v[i].w;
        }
        const bool lane_fresh = par ? (a_and & 1u) != 0u : (a_or & 1u) == 0u;
        if (__builtin_amdgcn_readfirstlane((int)(__all(lane_fresh) || failed))) break;
        tick();
        if (failed) break;
      }
    }
    if constexpr (SHARE) {
      if (step > 0) {   // (workgroup-uniform: every wave of both halves takes the barrier, whatever its own gather did)
        u32x4* ximg = reinterpret_cast<u32x4*>(xs) + wave * (NBW * 2 * 64) + lane;
#pragma unroll
        for (int i = 2 * MLO; i < 2 * MHI; ++i) ximg[i * 64] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * NBW; ++i)
          if (i < 2 * MLO || i >= 2 * MHI) v[i] = ximg[i * 64];
      }
    }
    KSTAMP(0, v[0].x);
    if (gather && !failed) {
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        if (i == (NBW > 1 ? 1 : 0)) prefetch(step + 1);
        const f16x8 bh = __builtin_bit_cast(f16x8, v[2 * i]), bl = __builtin_bit_cast(f16x8, v[2 * i + 1]);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tt][i], bh, i == 0 ? zero : acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tt][i], bh, acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tt][i], bl, acc[tt], 0, 0, 0);
      }
    } else {   // the first step (h_{-1} = 0), a wave without K range, a launch that gave up
      prefetch(step + 1);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float* rbuf = red + (step & 1) * (4 * K16_TILE);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(rbuf + tt * K16_TILE + (wave * 64 + wslot) * 4) = acc[tt];
    KSTAMP(1, acc[0][0]);
    if (Rp)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    const f32x4 zt = *reinterpret_cast<const f32x4*>(zring + (step & 1) * 256 + lane * 4);
    const float rt = Rp ? rring[(step & 1) * 64 + lane] : 0.f;
    __syncthreads();
    KSTAMP(2, zt[0]);
    const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
    unsigned packed = par | (par << 16);   // a padding cell: (hi, lo) = (0, 0) with the current parity
    float h = 0.f, yv = 0.f;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cvalid) {
      const float* mine = rbuf + rslot * 4;
      f32x4 sum = *reinterpret_cast<const f32x4*>(mine);
#pragma unroll
      for (int src = 1; src < 4; ++src) sum += *reinterpret_cast<const f32x4*>(mine + src * 64 * 4);
      f32x4 tot;
#pragma unroll
      for (int g = 0; g < 4; ++g) tot[g] = fmaf(sum[g], inv, zt[g]);
      h = mgr_cell_fwd(tot[0], tot[1], tot[2], tot[3], c, g4);
      nonfinite |= !(fabsf(h) < 2.f);   // (latched; reported once, behind the time loop)
      h = nonfinite ? 0.f : h;
      c = nonfinite ? 0.f : c;
      yv = nonfinite ? __uint_as_float(0x7FC00000u) : h;
      // h 2^15 = hi + lo.  The epoch parity rides in bit 0 of each published word: the last mantissa bit of the EVEN unit's hi (word
      // (hi_even, hi_odd)) and of the even unit's lo (word (lo_even, lo_odd)).  The bit is not forced: the value moves to the NEAREST
      // f16 whose last bit is the parity (k16_with_lsb) - the hi BEFORE lo is taken, so that lo absorbs the move.  Cost (round 5, found
      // by tests/test_gpu_split_adversarial.py: round 4 forced both bits, |h 2^15 - hi - lo| up to 2^-18.8 |h| on even units): even
      // unit |residual| <= 1 ulp(hi), lo within 1 ulp(lo) of it: <= 2^-20 |h|, odd unit untouched (2^-22).  (Measured and not kept: the
      // lo word's flag on the ODD unit's lo, bit 16 - every unit <= 2^-21 - but the two-mask check it needs in the gather cost the
      // audio step 0.3 us of 1.6: profiles/r05_scan_probes.txt.)
      float hs = h * 32768.f;
      asm volatile("" : "+v"(hs));   // (as above)
      unsigned hib = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)hs);
      hib = k16_with_lsb(hib, par, hs, ~lane & 1u);
      const float hif = (float)__builtin_bit_cast(_Float16, (unsigned short)hib);
      float ls = hs - hif;
      asm volatile("" : "+v"(ls));
      unsigned lob = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)ls);
      lob = k16_with_lsb(lob, par, ls, ~lane & 1u);
      packed = hib | (lob << 16);
    }
    KSTAMP(3, packed);
    {   // (the last step publishes too: nobody reads it, and the step has one branch less)
      const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)packed, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: the pair's other lane
      const unsigned w = (lane & 1) ? ((other >> 16) | (packed & 0xFFFF0000u))      // (lo_even, lo_odd)
                                    : ((packed & 0xFFFFu) | (other << 16));         // (hi_even, hi_odd)
      if (fast)
        __builtin_amdgcn_raw_buffer_store_b32(w, rs, poff, (step & 1) * IMGB, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b32(w, rs, poff, (step & 1) * IMGB, 16);  // sc1
    }
    KSTAMP(4, packed);
    if (cvalid && bvalid) {
      size_t row = (size_t)b * T + t;
      const float yo = yv + rt;
      typedef __attribute__((address_space(1))) float gfloat;
      typedef __attribute__((address_space(1))) f32x4 gf32x4;
      ((gfloat*)Yp)[row * ldy + unit] = yo;
      if (Gp) *(gf32x4*)(Gp + (row * H + unit) * 4) = (f32x4){g4.x, g4.y, g4.z, g4.w};
      if (Csp) ((gfloat*)Csp)[row * H + unit] = c;
      if (ytrow) {
        stg[(t & (KS_STG - 1)) * 64 + lane] = yo;
        if (reverse ? (t & (KS_STG - 1)) == 0 : ((t & (KS_STG - 1)) == KS_STG - 1 || t == T - 1)) {
          ks_flush_chunk(stg, lane, ytrow, t & ~(KS_STG - 1), jb.ldt, yt_split);
#pragma unroll
          for (int i = 0; i < KS_STG; ++i) stg[i * 64 + lane] = 0.f;
        }
      }
    }
    KSTAMP(5, yv);
  }
#ifdef MGR_STAMP
  if (lane == 0) {
    const int cls = (H > 400 ? 0 : 1) * 16;
    for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[cls + i], st_acc[i]);
    atomicAdd(&g_stamps[cls + 8], (unsigned long long)T);
    atomicAdd(&g_stamps[cls + 9], (unsigned long long)rounds);
    if (wave == 0 && ug == 0 && bg == 0 && !reverse) for (int i = 0; i < 6; ++i) g_stamps[32 + cls / 2 + i] = st_acc[i];
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last step's ring re-fetch: no LDS-DMA may outlive the workgroup)
  if (nonfinite) {   // (latched: reported and marked once, behind the time loop)
    __hip_atomic_fetch_or(cm.sticky, MGR_ST_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mgr_mark_sample(cm, b);
  }
  if (ytrow) ks_zero_tail(ytrow, T, jb.ldt, yt_split);
}

// ---------------------------------------------------------------------------------------------------------------
// PAIR form of the split-f16 K-split step (round 5): ONE workgroup per CU.  A workgroup holds its 16 hidden units' slice of U once and
// runs TWO 16-sample batch groups through it (cluster `bg` of a paired job owns the groups 2 bg and 2 bg + 1; an odd group count
// leaves the last cluster with one).  Same exchange images (one per 16-sample group), same layouts, same arithmetic and summation order
// as cluster_run_k16 - bit-identical results - but per step ONE round of gathers for both groups (one L2 round trip), one barrier,
// and half the workgroups: the encoder launch of config F is 204 workgroups instead of 408, no CU holds two of them (the pace of a
// cluster was set by its members that shared their CU with a workgroup of another cluster - uneven by construction), 52 CUs stay free
// and every scan CU keeps one wave slot per SIMD and ~50 KiB of LDS for the other stream's kernels.
constexpr int K16P_PER_S = 4 * K16_TILE + 4 * KS_STG * 64 + 4 * 2 * 256 + 4 * 2 * 64;
constexpr int K16P_LDS_FLOATS = 2 * K16P_PER_S + 16;

template <int NBW>
__device__ __forceinline__ void cluster_run_k16p(const ClusterJob& jb, const ClusterCommon& cm, int bg, int ug, float* smem, bool fast) {
  static_assert(NBW >= 1 && NBW <= 4, "1..4 K-blocks per wave (H <= 512)");
  unsigned* status = cm.status;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3
  const int H = jb.H, N = 4 * H, G = jb.G_;
  const int NKB = (H + 31) >> 5;
  const int IMGB = NKB * 2048;            // bytes of one exchange slot of one 16-sample group
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int om = lane & 15, okg = lane >> 4;
  const int fn = 4 * wave + (lane >> 4), fu = lane & 15;
  const int ns = (2 * bg + 1 < jb.nbg16) ? 2 : 1;     // 16-sample groups of this cluster (the same on every member)
  int b[2];
  bool bvalid[2];
  int bc[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    b[s] = (2 * bg + s) * 16 + fn;
    bvalid[s] = s < ns && b[s] < B;
    bc[s] = bvalid[s] ? b[s] : B - 1;
  }
  const int unit = 16 * ug + fu;
  const bool cvalid = unit < H;
  const float* __restrict__ Z = jb.Z;
  const float* __restrict__ Up = jb.Up;
  const float* Rp = jb.R;
  float *Yp = jb.Y, *Gp = jb.G, *Csp = jb.Cs;
  int ldr = jb.ldr, ldy = jb.ldy;
  asm volatile("" : "+s"(Rp), "+s"(Yp), "+s"(Gp), "+s"(Csp), "+s"(ldr), "+s"(ldy));   // (see cluster_run_ks)

  const int qb = wave * NBW;
  int nb = NKB - qb;
  nb = nb < 0 ? 0 : (nb > NBW ? NBW : nb);
  nb = __builtin_amdgcn_readfirstlane(nb);

  // LDS of sample group s at smem + s K16P_PER_S: partial sums [tile] K16_TILE | staging [4 waves][KS_STG][64] | Z rings | R rings
  float *red[2], *stg[2], *zring[2], *rring[2];
  unsigned zring_lds[2], rring_lds[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float* base = smem + s * K16P_PER_S;
    red[s] = base;
    stg[s] = base + 4 * K16_TILE + wave * (KS_STG * 64);
    zring[s] = base + 4 * K16_TILE + 4 * KS_STG * 64 + wave * (2 * 256);
    rring[s] = base + 4 * K16_TILE + 4 * KS_STG * 64 + 4 * 2 * 256 + wave * (2 * 64);
    zring_lds[s] = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)zring[s]);
    rring_lds[s] = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)rring[s]);
  }
  float* wmax = smem + 2 * K16P_PER_S;

  // ---- weights (as cluster_run_k16)
  auto uval = [&](int tt, int i, int e) -> float {
    const int k = 32 * (qb + i) + 8 * okg + e, uu = 16 * ug + 4 * tt + (om >> 2);
    return (k < H && uu < H) ? Up[(size_t)k * N + uu * 4 + (om & 3)] : 0.f;
  };
  float umax = 0.f;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) umax = fmaxf(umax, fabsf(uval(tt, i, e)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
  if (lane == 0) wmax[wave] = umax;
  __syncthreads();
  umax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  int ex = 0;
  if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &ex);
  ex = ex < -60 ? -60 : ex;
  const float sU = ldexpf(1.f, 15 - ex);
  const float inv = ldexpf(1.f, ex - 30);
  f16x8 ah[4][NBW], al[4][NBW];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = uval(tt, i, e) * sU;
        asm volatile("" : "+v"(x));
        const _Float16 hi = (_Float16)x;
        ah[tt][i][e] = hi;
        al[tt][i][e] = (_Float16)(x - (float)hi);
      }

  float* ytrow[2] = {nullptr, nullptr};
  const bool yt_split = jb.yt_split != 0;
#pragma unroll
  for (int s = 0; s < 2; ++s)
    if (jb.YT && cvalid && bvalid[s]) {
      ytrow[s] = jb.YT + (size_t)b[s] * jb.ytb + (size_t)unit * jb.ldt;
#pragma unroll
      for (int i = 0; i < KS_STG; ++i) stg[s][i * 64 + lane] = 0.f;
    }
  // the exchange slots of the two groups are neighbours in the job's buffer: [group][slot][IMGB]
  char* xb = reinterpret_cast<char*>(jb.xbuf) + (size_t)(2 * bg) * 2 * IMGB;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, ns * 2 * IMGB, 0x00020000);
  unsigned goff[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int kb = qb + (i < nb ? i : 0);
    const int half = (2 * kb + (okg >> 1) < G) ? (okg >> 1) : 0;
    goff[i] = (unsigned)(kb * 2048 + half * 512 + (om >> 2) * 128 + (om & 3) * 32 + (okg & 1) * 16);
  }
  const unsigned poff = (unsigned)((ug >> 1) * 2048 + (lane & 1) * 1024 + (ug & 1) * 512 + wave * 128 + (lane >> 4) * 32 + (fu >> 1) * 4);

  const int zunit = cvalid ? unit : 0;
  unsigned zvoff[2], rvoff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    zvoff[s] = (unsigned)(((size_t)bc[s] * T * N + (size_t)zunit * 4) * sizeof(float));
    rvoff[s] = Rp ? (unsigned)(((size_t)bc[s] * T * ldr + zunit) * sizeof(float)) : 0u;
  }
  {   // Z / R of step 0 of both groups
    const int t0 = reverse ? T - 1 : 0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < ns) {
        mgr_dma_b128(Z + (size_t)t0 * N, zvoff[s], zring_lds[s]);
        if (Rp) mgr_dma_b32(Rp + (size_t)t0 * ldr, rvoff[s], rring_lds[s]);
      }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the weight loads are retired before the time loop (see cluster_run_ks)

  const int wslot = k16_wslot(om, okg);
  const int rslot = k16_rslot(fu, fn);

  float c[2] = {0.f, 0.f};
  bool nonfinite[2] = {false, false};
  bool failed = false;
  unsigned rounds = 0;
  auto tick = [&]() {
    ++rounds;
    if ((rounds & 255u) == 0) {
      unsigned st;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st) : "v"(status) : "memory");
      if (__builtin_amdgcn_readfirstlane(st) != 0u) failed = true;
    }
    if (rounds > KS_ROUND_LIMIT) {
      failed = true;
      if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // One time step of ONE group: gather its h_{t-1} blocks, the matrix products, partial sums through LDS, barrier, cell, publish.
  // The two groups alternate: while the peers' publishes of group s travel (publish -> L2 -> visible: the hand-off latency that a
  // single chain waits for), this workgroup works on the other group - two barriers per pair of steps, and the partial-sum buffer of
  // a group needs no second copy (a wave re-writes it only behind the OTHER group's barrier, which every wave reaches after it has
  // read this one).
  auto half_step = [&](auto sc, int step) {
    constexpr int s = decltype(sc)::value;
    const int t = reverse ? T - 1 - step : step;
    f32x4 acc[4];
    u32x4 v[2 * NBW];
    const bool gather = step > 0 && nb > 0 && !failed;
    if (gather) {
      const unsigned sbase = (unsigned)(s * 2 * IMGB + ((step - 1) & 1) * IMGB);
      const unsigned par = ((((unsigned)(step - 1)) >> 1) & 1u) ^ 1u;
      for (;;) {
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          v[2 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], sbase, 16);              // sc1
          v[2 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] + 1024u, sbase, 16);   // sc1
        }
        unsigned a_and = 0xFFFFFFFFu, a_or = 0u;
#pragma unroll
        for (int i = 0; i < 2 * NBW; ++i) {
          a_and &= v[i].x & v[i].y & v[i].z & v[i].w;
          a_or |= v[i].x | v[i].y | v[i].z | v[i].w;
        }
        const bool lane_fresh = par ? (a_and & 1u) != 0u : (a_or & 1u) == 0u;
        if (__all(lane_fresh) || failed) break;
        tick();
        if (failed) break;
      }
    }
    auto prefetch_s = [&](int st) {
      if (st < T) {
        const int tt2 = reverse ? T - 1 - st : st;
        mgr_dma_b128(Z + (size_t)tt2 * N, zvoff[s], zring_lds[s] + (st & 1) * 1024);
        if (Rp) mgr_dma_b32(Rp + (size_t)tt2 * ldr, rvoff[s], rring_lds[s] + (st & 1) * 256);
      }
    };
    if (gather && !failed) {
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        if (i == (NBW > 1 ? 1 : 0)) prefetch_s(step + 1);
        const f16x8 bh = __builtin_bit_cast(f16x8, v[2 * i]), bl = __builtin_bit_cast(f16x8, v[2 * i + 1]);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tt][i], bh, i == 0 ? zero : acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tt][i], bh, acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tt][i], bl, acc[tt], 0, 0, 0);
      }
    } else {
      prefetch_s(step + 1);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float* rbuf = red[s];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(rbuf + tt * K16_TILE + (wave * 64 + wslot) * 4) = acc[tt];
    // (Z_t / R_t of this group were fetched a pair of steps ago; what this wave may still have in flight are the DMAs just issued)
    if (Rp)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    const f32x4 zt = *reinterpret_cast<const f32x4*>(zring[s] + (step & 1) * 256 + lane * 4);
    const float rt = Rp ? rring[s][(step & 1) * 64 + lane] : 0.f;
    __syncthreads();
    const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
    unsigned packed = par | (par << 16);
    float h = 0.f, yv = 0.f;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cvalid) {
      const float* mine = rbuf + rslot * 4;
      f32x4 sum = *reinterpret_cast<const f32x4*>(mine);
#pragma unroll
      for (int src = 1; src < 4; ++src) sum += *reinterpret_cast<const f32x4*>(mine + src * 64 * 4);
      f32x4 tot;
#pragma unroll
      for (int g = 0; g < 4; ++g) tot[g] = fmaf(sum[g], inv, zt[g]);
      h = mgr_cell_fwd(tot[0], tot[1], tot[2], tot[3], c[s], g4);
      if (!(fabsf(h) < 2.f) && !nonfinite[s]) {
        __hip_atomic_fetch_or(cm.sticky, MGR_ST_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nonfinite[s] = true;
      }
      if (nonfinite[s]) {
        h = 0.f;
        c[s] = 0.f;
      }
      yv = nonfinite[s] ? __uint_as_float(0x7FC00000u) : h;
      float hs = h * 32768.f;   // (the flag bits: cluster_run_k16)
      asm volatile("" : "+v"(hs));
      unsigned hib = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)hs);
      hib = k16_with_lsb(hib, par, hs, ~lane & 1u);
      const float hif = (float)__builtin_bit_cast(_Float16, (unsigned short)hib);
      float ls = hs - hif;
      asm volatile("" : "+v"(ls));
      unsigned lob = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)ls);
      lob = k16_with_lsb(lob, par, ls, ~lane & 1u);
      packed = hib | (lob << 16);
    }
    if (step + 1 < T) {
      const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)packed, 0xB1, 0xF, 0xF, true);
      const unsigned w = (lane & 1) ? ((other >> 16) | (packed & 0xFFFF0000u)) : ((packed & 0xFFFFu) | (other << 16));
      const unsigned so = (unsigned)(s * 2 * IMGB + (step & 1) * IMGB);
      if (fast)
        __builtin_amdgcn_raw_buffer_store_b32(w, rs, poff, so, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b32(w, rs, poff, so, 16);  // sc1
    }
    if (cvalid && bvalid[s]) {
      size_t row = (size_t)b[s] * T + t;
      const float yo = yv + rt;
      typedef __attribute__((address_space(1))) float gfloat;
      typedef __attribute__((address_space(1))) f32x4 gf32x4;
      ((gfloat*)Yp)[row * ldy + unit] = yo;
      if (Gp) *(gf32x4*)(Gp + (row * H + unit) * 4) = (f32x4){g4.x, g4.y, g4.z, g4.w};
      if (Csp) ((gfloat*)Csp)[row * H + unit] = c[s];
      if (ytrow[s]) {
        stg[s][(t & (KS_STG - 1)) * 64 + lane] = yo;
        if (reverse ? (t & (KS_STG - 1)) == 0 : ((t & (KS_STG - 1)) == KS_STG - 1 || t == T - 1)) {
          ks_flush_chunk(stg[s], lane, ytrow[s], t & ~(KS_STG - 1), jb.ldt, yt_split);
#pragma unroll
          for (int i = 0; i < KS_STG; ++i) stg[s][i * 64 + lane] = 0.f;
        }
      }
    }
  };
  for (int step = 0; step < T; ++step) {
    half_step(std::integral_constant<int, 0>{}, step);
    if (ns == 2) half_step(std::integral_constant<int, 1>{}, step);   // (uniform over the cluster)
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (nonfinite[s]) mgr_mark_sample(cm, b[s]);
    if (ytrow[s]) ks_zero_tail(ytrow[s], T, jb.ldt, yt_split);
  }
}

#define CLKS_FOREACH(X) X(125) X(75) X(32) X(25)

#define CL_FOREACH(X) \
  X(125, 1) X(75, 1) X(75, 2) X(32, 1) X(32, 2) X(32, 4) X(25, 1) X(25, 2) X(25, 4) X(16, 1) X(16, 2) X(8, 1) X(8, 2) \
  X(4, 1) X(3, 1) X(2, 1) X(1, 1)

// A workgroup locates its (job, batch group, unit group) by walking the launch's job table IN the kernel body (taking the
// address of the kernel argument in a helper would make hipcc copy the whole struct to scratch memory).  Members of a cluster
// are CONTIGUOUS workgroup ids: the round-robin dispatcher then spreads them over all XCDs, which measured best for the
// write-through exchange.
#define MGR_FOR_MY_JOB(L, jb, bg, ug)                                  \
  for (int k_ = 0; k_ < (L).njobs; ++k_)                               \
    if (const ClusterJob& jb = (L).job[k_]; true)                      \
      if (const int w_ = (int)blockIdx.x - jb.cls_begin; w_ >= 0 && w_ < jb.cls_nclusters * jb.G_) \
        if (const int ug = w_ % jb.G_, bg = w_ / jb.G_ - jb.cls_cluster0; bg >= 0 && bg < jb.nbg)

__global__ __launch_bounds__(CL_WAVES * 64) void k_scan_cluster(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
  MGR_FOR_MY_JOB(L, jb, bg, ug) {
#define CL_CASE(KS, TPW) \
  if (jb.ks == KS && jb.tpw == TPW) { cluster_run<KS, TPW>(jb, bg, ug, smem, L.cm.status); return mgr_cluster_exit(L.cm); }
    CL_FOREACH(CL_CASE)
#undef CL_CASE
    return;
  }
}

// K-split step: every job of the launch is a one-tile-per-wave, 4-wave cluster with an exchange (two workgroups per CU); XCD-local
// layout where the launcher chose it (lstm_cluster.h, mgr_cluster_octet), contiguous workgroup ids otherwise.
// Two kernels, by layer width: a kernel's register allocation is that of its largest instantiation (209 VGPRs for H = 500), and
// a narrow layer's scan (the fusion layer, H = 100: 56 workgroups that run BESIDE the 408 of the encoder scans) should not
// ask a CU for registers it never touches - with ~100 it fits on any CU that has a wave slot left.
#define CLKS_LARGE(X) X(125) X(75)
#define CLKS_SMALL(X) X(32) X(25)
template <bool SMALL>
__device__ __forceinline__ void scan_cluster_ks_body(const ClusterLaunch& L, float* smem) {
  mgr_cluster_enter(L.cm);
#define CLKS_CASE(KS) \
  if (jb.ks == KS) { cluster_run_ks<KS>(jb, L.cm, bg, ug, smem, same); return mgr_cluster_exit(L.cm); }
  if (L.xcd_local) {
    for (int k_ = 0; k_ < L.njobs; ++k_) {
      const ClusterJob& jb = L.job[k_];
      const int w_ = (int)blockIdx.x - jb.cls_begin, G = jb.G_;
      if (w_ < 0 || w_ >= (jb.cls_nclusters + 7) / 8 * 8 * G) continue;
      int cl, ug;
      const bool same = mgr_cluster_octet(L.cm, jb.cls_begin, G, jb.cls_rot, w_, cl, ug);
      const int bg = cl - jb.cls_cluster0;
      if (cl >= jb.cls_nclusters || bg < 0 || bg >= jb.nbg) continue;
      if constexpr (SMALL) {
        CLKS_SMALL(CLKS_CASE)
      } else {
        CLKS_LARGE(CLKS_CASE)
        CLKS_SMALL(CLKS_CASE)
      }
      return;
    }
    return;
  }
  MGR_FOR_MY_JOB(L, jb, bg, ug) {
    const bool same = false;
    if constexpr (SMALL) {
      CLKS_SMALL(CLKS_CASE)
    } else {
      CLKS_LARGE(CLKS_CASE)
      CLKS_SMALL(CLKS_CASE)
    }
    return;
  }
#undef CLKS_CASE
}

__global__ __launch_bounds__(256, 2) void k_scan_cluster_ks(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_ks_body<false>(L, smem);
}

// every job of the launch is narrow (H <= 128)
__global__ __launch_bounds__(256, 2) void k_scan_cluster_ks_s(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_ks_body<true>(L, smem);
}

// the split-f16 step (cluster_run_k16): same launch geometry and layouts as the two kernels above
template <bool SMALL>
__device__ __forceinline__ void scan_cluster_k16_body(const ClusterLaunch& L, float* smem) {
  mgr_cluster_enter(L.cm);
#define K16_RUN(NBW) \
  if (nbw == NBW) { cluster_run_k16<NBW>(jb, L.cm, bg, ug, smem, same); return mgr_cluster_exit(L.cm); }
#define K16_DISPATCH                                     \
  {                                                      \
    const int nbw = (((jb.H + 31) >> 5) + 3) >> 2;       \
    K16_RUN(1)                                           \
    if constexpr (!SMALL) { K16_RUN(2) K16_RUN(3) K16_RUN(4) } \
    return;                                              \
  }
  if (L.xcd_local) {
    for (int k_ = 0; k_ < L.njobs; ++k_) {
      const ClusterJob& jb = L.job[k_];
      const int w_ = (int)blockIdx.x - jb.cls_begin, G = jb.G_;
      if (w_ < 0 || w_ >= (jb.cls_nclusters + 7) / 8 * 8 * G) continue;
      int cl, ug;
      const bool same = mgr_cluster_octet(L.cm, jb.cls_begin, G, jb.cls_rot, w_, cl, ug);
      const int bg = cl - jb.cls_cluster0;
      if (cl >= jb.cls_nclusters || bg < 0 || bg >= jb.nbg) continue;
      K16_DISPATCH
    }
    return;
  }
  MGR_FOR_MY_JOB(L, jb, bg, ug) {
    const bool same = false;
    K16_DISPATCH
  }
#undef K16_DISPATCH
#undef K16_RUN
}

__global__ __launch_bounds__(256, 2) void k_scan_cluster_k16(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_k16_body<false>(L, smem);
}

__global__ __launch_bounds__(256, 2) void k_scan_cluster_k16_s(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  scan_cluster_k16_body<true>(L, smem);
}

// the pair form (cluster_run_k16p): one workgroup per CU, two 16-sample groups per workgroup; same launch layout, `bg` counts pairs
__global__ __launch_bounds__(256, 1) void k_scan_cluster_k16p(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
#define K16P_RUN(NBW) \
  if (nbw == NBW) { cluster_run_k16p<NBW>(jb, L.cm, bg, ug, smem, same); return mgr_cluster_exit(L.cm); }
#define K16P_DISPATCH                                    \
  {                                                      \
    const int nbw = (((jb.H + 31) >> 5) + 3) >> 2;       \
    K16P_RUN(1) K16P_RUN(2) K16P_RUN(3) K16P_RUN(4)      \
    return;                                              \
  }
  if (L.xcd_local) {
    for (int k_ = 0; k_ < L.njobs; ++k_) {
      const ClusterJob& jb = L.job[k_];
      const int w_ = (int)blockIdx.x - jb.cls_begin, G = jb.G_;
      if (w_ < 0 || w_ >= (jb.cls_nclusters + 7) / 8 * 8 * G) continue;
      int cl, ug;
      const bool same = mgr_cluster_octet(L.cm, jb.cls_begin, G, jb.cls_rot, w_, cl, ug);
      const int bg = cl - jb.cls_cluster0;
      if (cl >= jb.cls_nclusters || bg < 0 || bg >= jb.nbg) continue;
      K16P_DISPATCH
    }
    return;
  }
  MGR_FOR_MY_JOB(L, jb, bg, ug) {
    const bool same = false;
    K16P_DISPATCH
  }
#undef K16P_DISPATCH
#undef K16P_RUN
}

// the fused form: one 8-wave workgroup per CU = two unit groups (2 j, 2 j + 1) of one cluster; the launch lays out ceil(G / 2) members
// per cluster (XCD-local octets as above); a unit group beyond G (odd G) only keeps the barrier count
__global__ __launch_bounds__(512, 1) void k_scan_cluster_k16f(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
  const int tg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  for (int k_ = 0; k_ < L.njobs; ++k_) {
    const ClusterJob& jb = L.job[k_];
    const int G = jb.G_, Gr = (G + 1) / 2;
    const int w_ = (int)blockIdx.x - jb.cls_begin;
    if (w_ < 0 || w_ >= (jb.cls_nclusters + 7) / 8 * 8 * Gr) continue;
    int cl, ugr;
    const bool same = mgr_cluster_octet(L.cm, jb.cls_begin, Gr, jb.cls_rot, w_, cl, ugr);
    const int bg = cl - jb.cls_cluster0;
    if (cl >= jb.cls_nclusters || bg < 0 || bg >= jb.nbg) continue;
    const int ug = 2 * ugr + tg;
    float* sm = smem + tg * K16_LDS_FLOATS;
    if (ug >= G) {   // (odd G: the last workgroup's second half)
      __syncthreads();
      for (int step = 0; step < jb.T; ++step) __syncthreads();
      return mgr_cluster_exit(L.cm);
    }
    const int nbw = (((jb.H + 31) >> 5) + 3) >> 2;
    if (nbw == 1) cluster_run_k16<1, true>(jb, L.cm, bg, ug, sm, same);
    else if (nbw == 2) cluster_run_k16<2, true>(jb, L.cm, bg, ug, sm, same);
    else if (nbw == 3) cluster_run_k16<3, true>(jb, L.cm, bg, ug, sm, same);
    else cluster_run_k16<4, true>(jb, L.cm, bg, ug, sm, same);
    return mgr_cluster_exit(L.cm);
  }
}

// the fused form with the SHARED gather (cluster_run_k16<.., true, half>): as k_scan_cluster_k16f, plus 2 KiB of LDS per wave and K-block
// for the image the two halves exchange; a unit group beyond G (odd G) runs as a member without valid cells
__global__ __launch_bounds__(512, 1) void k_scan_cluster_k16fs(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
  const int tg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  for (int k_ = 0; k_ < L.njobs; ++k_) {
    const ClusterJob& jb = L.job[k_];
    const int G = jb.G_, Gr = (G + 1) / 2;
    const int w_ = (int)blockIdx.x - jb.cls_begin;
    if (w_ < 0 || w_ >= (jb.cls_nclusters + 7) / 8 * 8 * Gr) continue;
    int cl, ugr;
    const bool same = mgr_cluster_octet(L.cm, jb.cls_begin, Gr, jb.cls_rot, w_, cl, ugr);
    const int bg = cl - jb.cls_cluster0;
    if (cl >= jb.cls_nclusters || bg < 0 || bg >= jb.nbg) continue;
    const int ug = 2 * ugr + tg;
    float* sm = smem + tg * K16_LDS_FLOATS;
    float* xs = smem + 2 * K16_LDS_FLOATS;
    const int nbw = (((jb.H + 31) >> 5) + 3) >> 2;
#define K16FS_RUN(NBW)                                                        \
  if (nbw == NBW) {                                                           \
    if (tg == 0) cluster_run_k16<NBW, true, 0>(jb, L.cm, bg, ug, sm, same, xs); \
    else cluster_run_k16<NBW, true, 1>(jb, L.cm, bg, ug, sm, same, xs);       \
    return mgr_cluster_exit(L.cm);                                            \
  }
    K16FS_RUN(1) K16FS_RUN(2) K16FS_RUN(3) K16FS_RUN(4)
#undef K16FS_RUN
    return;
  }
}
constexpr size_t K16FS_LDS_BYTES = (2 * (size_t)K16_LDS_FLOATS + 4 * 4 * 2 * 256) * sizeof(float);

}  // namespace

bool mgr_cluster_supported(int ks, int tpw) {
#define CL_CASE(KS, TPW) \
  if (ks == KS && tpw == TPW) return true;
  CL_FOREACH(CL_CASE)
#undef CL_CASE
  return false;
}

bool mgr_cluster_ks_supported(int ks) {
#define CLKS_CASE(KS) \
  if (ks == KS) return true;
  CLKS_FOREACH(CLKS_CASE)
#undef CLKS_CASE
  return false;
}

static bool ks_eligible(const ClusterLaunch& L, bool any_exchange, int waves) {
  bool ks_all = L.ksplit && any_exchange && waves == 4;
  for (int i = 0; i < L.njobs && ks_all; ++i) {
    const ClusterJob& j = L.job[i];
    bool inst = false;
#define CLKS_CASE(KS) \
  if (j.ks == KS) inst = true;
    CLKS_FOREACH(CLKS_CASE)
#undef CLKS_CASE
    ks_all = inst && j.G_ > 1 && j.tpw == 1 && j.nw == 4;
  }
  return ks_all;
}

static size_t image_lds(const ClusterLaunch& L) {
  size_t lds = 0;
  for (int i = 0; i < L.njobs; ++i) {
    size_t need = 2 * (size_t)((L.job[i].ks + 3) / 4) * 256 * sizeof(float);
    lds = need > lds ? need : lds;
  }
  return lds;
}

void mgr_cluster_geometry(const ClusterLaunch& L, bool any_exchange, int* waves, int* per_cu) {
  int maxnw = 0;
  for (int i = 0; i < L.njobs; ++i) maxnw = L.job[i].nw > maxnw ? L.job[i].nw : maxnw;
  *waves = maxnw <= 4 ? 4 : CL_WAVES;
  // 4-wave workgroups with <= 80 KiB of LDS fit two per CU (8 waves, <= 256 VGPRs each); anything else sits alone on its CU
  *per_cu = (*waves == 4 && (ks_eligible(L, any_exchange, *waves) || image_lds(L) <= 80 * 1024)) ? 2 : 1;
  if (L.pair && L.split16 && ks_eligible(L, any_exchange, *waves)) *per_cu = 1;   // (the pair form's 101 KiB of LDS: alone among scans on its CU)
  if (L.fused && L.split16 && ks_eligible(L, any_exchange, *waves)) {              // (the fused form: 8 waves, 104 KiB, a CU of its own)
    *waves = 8;
    *per_cu = 1;
  }
}

bool mgr_cluster_uses_ks(const ClusterLaunch& L, bool any_exchange) {
  if (L.fused && L.split16 && ks_eligible(L, any_exchange, 4)) return true;
  int waves, per_cu;
  mgr_cluster_geometry(L, any_exchange, &waves, &per_cu);
  return ks_eligible(L, any_exchange, waves);
}

int mgr_cluster_launch(mgr_ctx* c, const ClusterLaunch& L, int total_wgs, bool any_exchange) {
  int waves, per_cu;
  mgr_cluster_geometry(L, any_exchange, &waves, &per_cu);
  size_t lds = image_lds(L);
  if (any_exchange) {
    // co-residency of every spinning workgroup is what makes the in-launch hand-off deadlock-free; a workgroup that must sit
    // alone on its CU says so through its LDS request
    if (per_cu == 1 && lds < 84 * 1024) lds = 84 * 1024;
    const int live = L.live_wgs > 0 ? L.live_wgs : total_wgs;
    MGR_REQUIRE(live <= per_cu * c->cu_count, "cluster scan needs %d co-resident workgroups but the device holds %d", live,
                per_cu * c->cu_count);
  }
  if (!(c->attr_done & 1u)) {   // (function attributes are per device, hence per context)
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_ks), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_ks_s), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_k16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_k16_s), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_k16p), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->attr_done |= 1u;
  }
  const bool fused = L.fused && L.split16 && ks_eligible(L, any_exchange, 4);
  MGR_REQUIRE(!L.xcd_local || fused || ks_eligible(L, any_exchange, waves), "XCD-local layout is only understood by the K-split kernel");
  if (fused) {
    MGR_REQUIRE(L.xcd_local, "the fused form is laid out in octets");
    if (!(c->attr_done & 4u)) {   // (per device, hence per context - like the block above)
      MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_k16f), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      c->attr_done |= 4u;
    }
    if (!(c->attr_done & 64u)) {
      MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_k16fs), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      c->attr_done |= 64u;
    }
    // tune key 17: 1 = the two halves of a fused workgroup each fetch the whole h image themselves (round 5's form)
    if (c->tune[17] == 0)
      hipLaunchKernelGGL(k_scan_cluster_k16fs, dim3(total_wgs), dim3(512), K16FS_LDS_BYTES, mgr_stream(c), L);
    else
      hipLaunchKernelGGL(k_scan_cluster_k16f, dim3(total_wgs), dim3(512), 2 * K16_LDS_FLOATS * sizeof(float), mgr_stream(c), L);
  } else if (ks_eligible(L, any_exchange, waves)) {
    // partial-sum exchange, staging tiles of the transposed output, Z / R rings (no h image): 50 KiB, two workgroups per CU
    bool small = true;
    for (int i = 0; i < L.njobs; ++i) small = small && L.job[i].ks <= 32;
    if (L.split16 && L.pair) {
      hipLaunchKernelGGL(k_scan_cluster_k16p, dim3(total_wgs), dim3(256), K16P_LDS_FLOATS * sizeof(float), mgr_stream(c), L);
    } else if (L.split16) {
      if (small)
        hipLaunchKernelGGL(k_scan_cluster_k16_s, dim3(total_wgs), dim3(256), K16_LDS_FLOATS * sizeof(float), mgr_stream(c), L);
      else
        hipLaunchKernelGGL(k_scan_cluster_k16, dim3(total_wgs), dim3(256), K16_LDS_FLOATS * sizeof(float), mgr_stream(c), L);
    } else if (small)
      hipLaunchKernelGGL(k_scan_cluster_ks_s, dim3(total_wgs), dim3(256), KS_LDS_FLOATS * sizeof(float), mgr_stream(c), L);
    else
      hipLaunchKernelGGL(k_scan_cluster_ks, dim3(total_wgs), dim3(256), KS_LDS_FLOATS * sizeof(float), mgr_stream(c), L);
  } else {
    hipLaunchKernelGGL(k_scan_cluster, dim3(total_wgs), dim3(waves * 64), lds, mgr_stream(c), L);
  }
  MGR_LAUNCH_CHECK();
  return 0;
}
