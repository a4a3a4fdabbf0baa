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
#include <type_traits>

#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CL_WAVES = 8;
constexpr unsigned POLL_LIMIT = 1u << 20;
constexpr int KS_STG_ROW = 36;   // floats per lane of the transposed-output staging rows (cluster_run_ks)
constexpr unsigned KS_ROUND_LIMIT = 1u << 16;   // K-split step: ~0.1 s of re-polling a late producer, ~1 s of lost loads

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
// K-split variant of the one-tile-per-wave cluster step (4 waves, 4 tiles = ONE 1 KiB image block per workgroup).
// Instead of gathering the whole h_{t-1} image into LDS, joining at a barrier and then letting every wave run the full
// K loop for its own tile, wave w here owns a QUARTER OF K for ALL FOUR tiles of the workgroup:
//   * it polls only the image blocks of its K range and takes them STRAIGHT INTO REGISTERS as MFMA B operands (the
//     block layout [kk][sample][r] is exactly the B fragment of four consecutive k-steps) - no LDS image, no B-operand
//     ds_reads under the MFMAs, no barrier between gather and MFMA; blocks still showing the previous epoch are polled again.
//   * the four partial sums per tile are exchanged through 16 KiB of LDS (double-buffered on the step parity: ONE
//     barrier per step), then every wave finishes a quarter of the workgroup's 16 x 16 (unit, sample) cells: adds Z_t,
//     runs the cell, publishes h_t (same data-is-the-flag parity words as cluster_run) and streams Y / gates / c out.
//
// Which hidden unit sits in which MFMA slot is this kernel's private choice (U rows / columns, Z, Y, gates and c are
// addressed through it; nothing outside sees it).  Block q of the image holds the nv = min(4, KS - 4q) tiles 4q .. 4q+nv-1;
//   slot (tile 4q + r, unit-in-tile u)  <->  hidden unit 16q + nv*u + r        (PERM; identity order 4*(4q+r) + u otherwise)
// and the finishing lane (r = lane>>4, sample j = lane&15) of wave u owns exactly that slot.  With this order
//   * a wave's 64 h words of one step are the 256 CONTIGUOUS bytes [q][kk = u][j][r] of the image: after one ds_bpermute the
//     wave publishes them as ONE coalesced store instruction = two whole 128-byte lines (the identity order makes every wave
//     write one dword of every 16-byte chunk of the block: 32 quarter-filled line writes per workgroup and step, and a
//     reader that sees a line between two of them polls again);
//   * the four lanes r = 0..3 of a sample hold four CONSECUTIVE units: Z loads, Y / gate / c stores stay as coalesced as
//     in the identity order.
template <int KS, bool PERM>
__device__ __forceinline__ void cluster_run_ks(const ClusterJob& jb, const ClusterCommon& cm, int bg, int ug, float* smem, bool fast = false) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256, NBW = (QN + 3) / 4;
  static_assert(NBW <= 8, "at most 8 image blocks per wave (H <= 512)");
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

  // hidden unit of MFMA slot (tile, unit-in-tile) = of k-slot (s = tile, kk = unit-in-tile)
  auto unit_of = [](int tile, int u) {
    if (!PERM) return tile * 4 + u;
    const int q = tile >> 2, nv = (KS - 4 * q) < 4 ? (KS - 4 * q) : 4;
    return 16 * q + nv * u + (tile & 3);
  };

  // K range of this wave: image blocks [qb, qb + nb)
  const int qb = wave * NBW;
  int nb = QN - qb;
  nb = nb < 0 ? 0 : (nb > NBW ? NBW : nb);
  nb = __builtin_amdgcn_readfirstlane(nb);

  // U^T fragments of the workgroup's four tiles for this wave's k-steps (zero where tile or k-step does not exist):
  // A[m = lane&15 = 4*(unit-in-tile) + gate][k = lane>>4] = U[unit of k-slot (s, uq)][packed column of slot (tile, j>>2), gate j&3]
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
  // the cell this lane finishes: PERM: slot (tile 4*ug + uq, unit-in-tile wave); identity: slot (tile 4*ug + wave, unit-in-tile uq)
  const int ftile = ug * 4 + (PERM ? uq : wave);
  const bool cvalid = ftile < KS;   // (identity order: wave-uniform)
  const int unit = cvalid ? unit_of(ftile, PERM ? wave : uq) : 0;
  // where its partial sums lie in the reduction buffer [tile][src wave][slot lane = u*16 + j], and its word of the image
  const int red_off = PERM ? ((uq * 4) * 64 + wave * 16 + j) * 4 : ((wave * 4) * 64 + lane) * 4;
  const int idx = PERM ? ((ug * 4 + wave) * 16 + j) * 4 + uq : ((ug * 4 + uq) * 16 + j) * 4 + wave;   // [q][kk][j][r]

  float* red = smem;  // [2][tile][src wave][lane] f32x4
  // transposed output (jb.YT): this lane's last <= 32 outputs wait in LDS (row of 36 floats: 16-byte aligned, 8 banks apart)
  // and leave as one 128-byte row segment YT[b][unit][32-step chunk] - the transposed copy the next layer's dropout-aware
  // projection and dW read (gemm.hip) costs no kernel of its own and no second pass over Y
  float* stg = smem + 2 * 16 * 64 * 4 + (wave * 64 + lane) * KS_STG_ROW;
  float* ytrow = nullptr;
  if (jb.YT && cvalid && bvalid) {
    ytrow = jb.YT + (size_t)b * jb.ytb + (size_t)unit * jb.ldt;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(stg + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float* xb = jb.xbuf + (size_t)bg * 2 * IMG;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * IMG * 4, 0x00020000);

  float c = 0.f;
  bool nonfinite = false;
  f32x4 zr0 = {0.f, 0.f, 0.f, 0.f}, zr1 = zr0, zr2 = zr0;
  auto loadz = [&](f32x4& z, int step) {
    if (step < T && cvalid) {
      const int t = reverse ? T - 1 - step : step;
      z = *reinterpret_cast<const f32x4*>(Z + ((size_t)bc * T + t) * N + unit * 4);
    }
  };
  loadz(zr0, 0);
  loadz(zr1, 1);
  bool failed = false;

  // ---- gather: the image blocks of this wave's K range go straight into registers.
  // The loads are issued from inline asm, so hipcc does not know that the registers have loads pending and inserts no
  // s_waitcnt in front of their readers; instead the wave POLLS THE REGISTERS: they are preset to a pattern no h word can
  // have (quiet-NaN exponent, wrong epoch parity) and an empty asm with "+v" constraints makes every iteration re-read
  // them.  A word that still shows the preset has not landed; a landed word with the previous epoch's parity means the
  // producer was late and the block is fetched again.  Nothing here waits on vmcnt, so the wave's own write-through
  // stores (whose acknowledgement takes longer than a load round trip) are never waited for.  That no compiler-inserted
  // copy or spill touches these registers while a load may be in flight is verified on the device assembly of every build
  // (_build.check_hidden_loads).  A non-finite h can never be mistaken for the preset: the cell replaces it before
  // publishing (below).
  auto hidden_load = [&](u32x4& dst, const char* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(dst) : "v"(p) : "memory");
  };
  // both statements only tell the compiler "these registers may have changed" (they are written by loads it cannot see);
  // touch() sleeps a few cycles between polls, poll_end() drains the loads and marks the end of a polling window for
  // _build.check_hidden_loads
#define MGR_POLLED_ASM(TEXT)                                                                                                  \
  if constexpr (NBW == 8)                                                                                                       \
    asm volatile(TEXT : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory"); \
  else if constexpr (NBW == 5)                                                                                                  \
    asm volatile(TEXT : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4])::"memory");                                 \
  else if constexpr (NBW == 2)                                                                                                  \
    asm volatile(TEXT : "+v"(v[0]), "+v"(v[1])::"memory");                                                                     \
  else                                                                                                                          \
    static_assert(NBW == 8 || NBW == 5 || NBW == 2, "add an arm for this block count");
  auto touch = [&](u32x4 (&v)[NBW]) { MGR_POLLED_ASM("s_sleep 1") };
  auto poll_end = [&](u32x4 (&v)[NBW]) { MGR_POLLED_ASM("s_waitcnt vmcnt(0) ; MGR_POLL_END") };
#undef MGR_POLLED_ASM
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
    loadz(zload, step + 2);
    f32x4 acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (step > 0 && nb > 0 && !failed) {
      const int slot = (step - 1) & 1;
      const unsigned par = ((((unsigned)(step - 1)) >> 1) & 1u) ^ 1u;
      const unsigned bad = 0x7FC00000u | (par ^ 1u);   // never a published word (|h| < 2): bit 30 set, wrong parity
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
            if (lane == 0) __hip_atomic_store(status, MGR_ST_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (failed) break;
        }
      }
      mfmas(v, acc);
      // keep the polling registers allocated until here, then make sure no re-issued load is still in flight before this
      // wave publishes (a producer may overwrite the slot only after it has seen that publish)
      poll_end(v);
    }
    // the four partial sums of every tile meet in LDS (all four go through it: selecting "my own" accumulator by the
    // run-time wave id would force the accumulators into scratch memory)
    float* rbuf = red + (step & 1) * (16 * 64 * 4);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(rbuf + ((tt * 4 + wave) * 64 + lane) * 4) = acc[tt];
    __syncthreads();
    const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
    unsigned hbits = par;   // cells of a padding tile: value 0 with the current parity, so that consumers can test whole blocks
    float h = 0.f, yv = 0.f;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cvalid) {
      f32x4 tot = zuse;
      const float* mine = rbuf + red_off;
#pragma unroll
      for (int src = 0; src < 4; ++src) tot += *reinterpret_cast<const f32x4*>(mine + src * 64 * 4);
      h = mgr_cell_fwd(tot[0], tot[1], tot[2], tot[3], c, g4);
      yv = h;
      if (!(fabsf(h) < 2.f)) {
        // NaN / Inf (diverged weights, bad checkpoint): Y keeps the NaN so that the loss turns NaN like the reference's, but
        // what is published - and fed back - is finite: a NaN word would look like a load that has not landed (bit 30)
        h = 0.f;
        c = 0.f;
        if (!nonfinite) __hip_atomic_fetch_or(cm.sticky, MGR_ST_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nonfinite = true;
      }
      hbits = (__float_as_uint(h) & ~1u) | par;  // epoch parity rides in the mantissa LSB
      h = __uint_as_float(hbits);
      if (!nonfinite) yv = h;
    }
    if (step + 1 < T) {
      if (PERM) {
        // lane (r, j) holds image word j*4 + r of the wave's 64-word segment: bring word l to lane l, one coalesced store
        const unsigned w = __builtin_amdgcn_ds_bpermute((((lane & 3) << 4) | (lane >> 2)) << 2, hbits);
        if (fast)   // whole cluster on one XCD (verified at start): the line stays in the L2 every peer's sc1 load is served from
          __builtin_amdgcn_raw_buffer_store_b32(w, rs, ((step & 1) * IMG + (ug * 4 + wave) * 64 + lane) * 4, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(w, rs, ((step & 1) * IMG + (ug * 4 + wave) * 64 + lane) * 4, 0, 16);  // sc1
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(hbits, rs, ((step & 1) * IMG + idx) * 4, 0, 16);  // sc1 write-through
      }
    }
    if (cvalid && bvalid) {
      size_t row = (size_t)b * T + t;
      float yo = yv;
      if (jb.R) yo += jb.R[row * jb.ldr + unit];
      jb.Y[row * jb.ldy + unit] = yo;
      if (jb.G) *reinterpret_cast<float4*>(jb.G + (row * H + unit) * 4) = g4;
      if (jb.Cs) jb.Cs[row * H + unit] = c;
      if (ytrow) {
        stg[t & 31] = yo;
        // the chunk [t & ~31, +32) is complete when the walk leaves it (all lanes of the launch agree on t)
        if (reverse ? (t & 31) == 0 : ((t & 31) == 31 || t == T - 1)) {
          float* dst = ytrow + (t & ~31);
#pragma unroll 1
          for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<f32x4*>(dst + 4 * i) = *reinterpret_cast<const f32x4*>(stg + 4 * i);
            *reinterpret_cast<f32x4*>(stg + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};   // (a partial last chunk pads with zeros)
          }
        }
      }
    }
  };

  for (int s0 = 0; s0 < T; s0 += 3) {
    do_step(s0, zr0, zr2);
    if (s0 + 1 < T) do_step(s0 + 1, zr1, zr0);
    if (s0 + 2 < T) do_step(s0 + 2, zr2, zr1);
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

// K-split step: every job of the launch is a one-tile-per-wave, 4-wave cluster with an exchange (two workgroups per CU)
__global__ __launch_bounds__(256, 2) void k_scan_cluster_ks(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
  if (L.xcd_local) {   // XCD-local exchange: lstm_cluster.h, mgr_cluster_octet
    for (int k_ = 0; k_ < L.njobs; ++k_) {
      const ClusterJob& jb = L.job[k_];
      const int w_ = (int)blockIdx.x - jb.cls_begin, G = jb.G_;
      if (w_ < 0 || w_ >= (jb.cls_nclusters + 7) / 8 * 8 * G) continue;
      int cl, ug;
      const bool same = mgr_cluster_octet(L.cm, jb.cls_begin, G, jb.cls_rot, w_, cl, ug);
      const int bg = cl - jb.cls_cluster0;
      if (cl >= jb.cls_nclusters || bg < 0 || bg >= jb.nbg) continue;
#define CLKS_CASE(KS) \
  if (jb.ks == KS) { cluster_run_ks<KS, true>(jb, L.cm, bg, ug, smem, same); return mgr_cluster_exit(L.cm); }
      CLKS_FOREACH(CLKS_CASE)
#undef CLKS_CASE
      return;
    }
    return;
  }
  MGR_FOR_MY_JOB(L, jb, bg, ug) {
#define CLKS_CASE(KS) \
  if (jb.ks == KS) { cluster_run_ks<KS, true>(jb, L.cm, bg, ug, smem); return mgr_cluster_exit(L.cm); }
    CLKS_FOREACH(CLKS_CASE)
#undef CLKS_CASE
    return;
  }
}

// the same step with hidden units in identity order (mgr_tune key 7 = 2): kept as the cross-check of the unit permutation
__global__ __launch_bounds__(256, 2) void k_scan_cluster_ks_id(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mgr_cluster_enter(L.cm);
  MGR_FOR_MY_JOB(L, jb, bg, ug) {
#define CLKS_CASE(KS) \
  if (jb.ks == KS) { cluster_run_ks<KS, false>(jb, L.cm, bg, ug, smem); return mgr_cluster_exit(L.cm); }
    CLKS_FOREACH(CLKS_CASE)
#undef CLKS_CASE
    return;
  }
}

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
}

bool mgr_cluster_uses_ks(const ClusterLaunch& L, bool any_exchange) {
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
    MGR_REQUIRE(total_wgs <= per_cu * c->cu_count, "cluster scan needs %d co-resident workgroups but the device holds %d",
                total_wgs, per_cu * c->cu_count);
  }
  if (!(c->attr_done & 1u)) {   // (function attributes are per device, hence per context)
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_ks), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_ks_id), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->attr_done |= 1u;
  }
  MGR_REQUIRE(!L.xcd_local || (ks_eligible(L, any_exchange, waves) && L.ksplit == 1), "XCD-local layout is only understood by the K-split kernel");
  if (ks_eligible(L, any_exchange, waves)) {
    // partial-sum exchange only (no h image)
    size_t lds_ks = 2 * 16 * 64 * 4 * sizeof(float);
    for (int i = 0; i < L.njobs; ++i)
      if (L.job[i].YT) lds_ks = (2 * 16 * 64 * 4 + 256 * KS_STG_ROW) * sizeof(float);   // + staging rows of the transposed output
    if (L.ksplit == 2)
      hipLaunchKernelGGL(k_scan_cluster_ks_id, dim3(total_wgs), dim3(256), lds_ks, mgr_stream(c), L);
    else
      hipLaunchKernelGGL(k_scan_cluster_ks, dim3(total_wgs), dim3(256), lds_ks, mgr_stream(c), L);
  } else {
    hipLaunchKernelGGL(k_scan_cluster, dim3(total_wgs), dim3(waves * 64), lds, mgr_stream(c), L);
  }
  MGR_LAUNCH_CHECK();
  return 0;
}
