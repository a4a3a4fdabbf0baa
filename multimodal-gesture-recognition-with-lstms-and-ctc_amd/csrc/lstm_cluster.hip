// K3 for large H (300, 500): persistent, weight-stationary LSTM scan spread over CLUSTERS of CUs.
//
// A cluster = the G workgroups (one per CU) that together hold one direction's recurrent matrix for one
// 16-sample batch group: workgroup `ug` keeps the U^T fragments of its nw*TPW M-tiles (4 units x 4 gates each)
// in VGPRs for all T steps.  Per time step every workgroup
//   1. runs its MFMA chain  D[gate-col, sample] += U^T . h_{t-1}^T  (v_mfma_f32_16x16x4_f32, B operand from an LDS
//      image of h_{t-1} laid out [k/16][k%4][sample][(k/4)%4], same as lstm_mfma.hip),
//   2. applies the cell update in registers, writes its h_t slice into the next LDS image,
//   3. PUBLISHES every h value the moment it is computed: a 4-byte write-through (sc1) store into the cluster's
//      exchange slot (t&1), in the same image layout.  THE DATA IS THE FLAG: the least-significant mantissa bit of
//      each value carries the epoch parity ((t>>1)&1)^1, which flips every time a slot word is rewritten (the
//      local copy, Y and the recurrence all use the same 1-ulp-adjusted value, so all replicas agree bit for bit),
//   4. GATHERS the peers' slices: each of the 8 waves sweeps its share of the image's 1 KiB blocks with 16-byte
//      sc1 loads, accepts a block once all of its words show the expected parity, and writes it to the next LDS
//      image; one workgroup barrier per time step.
// This is the guide's granule hand-off (Guideline 16 R2, "the data is the flag") with a 4-byte granule: every word
// is written by exactly one aligned store per epoch; a reader of epoch t can only ever see the word of epoch t-2
// (opposite parity) or t, never t+2, because a producer cannot publish epoch t+2 before every peer has published
// t+1, i.e. finished consuming t.  It needs no fence, no flag round trip and no drain, so a step costs ONE
// store->load flight instead of three.  hipMalloc memory; one workgroup per CU (enforced by requesting > 80 KiB
// of LDS); slots are zeroed by a memset node ahead of every launch.
// Every spin is bounded; a give-up sets status[0] and the host reports an error instead of hanging the GPU.
//
// Several layer-directions ("jobs": audio fwd/rev, skeletal fwd/rev) share ONE launch so that all spinning
// workgroups are co-resident by construction (grid <= number of CUs).
#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CL_WAVES = 8;
constexpr unsigned POLL_LIMIT = 1u << 20;

template <int KS, int TPW>
__device__ __forceinline__ void cluster_run(const ClusterJob& jb, int wg, float* smem, unsigned* status) {
  constexpr int H = 4 * KS, N = 4 * H, QN = (KS + 3) / 4, IMG = QN * 256;
  static_assert(QN <= 4 * CL_WAVES, "gather sweep covers at most 32 image blocks (H <= 512)");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, uq = lane >> 4;
  const int G = jb.G_;
  const int bg = wg / G, ug = wg % G;
  const int nw = jb.nw;
  const int tiles_per_wg = nw * TPW;  // multiple of 4
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  const float* __restrict__ Z = jb.Z;
  const float* __restrict__ Up = jb.Up;
  float* img = smem;  // [2][IMG]

  float uf[TPW][KS];
  bool tv[TPW];
  int tl[TPW];
#pragma unroll
  for (int jt = 0; jt < TPW; ++jt) {
    int tile = ug * tiles_per_wg + wave * TPW + jt;
    tv[jt] = wave < nw && tile < KS;
    tl[jt] = tv[jt] ? tile : 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) uf[jt][s] = tv[jt] ? Up[(size_t)(4 * s + uq) * N + tl[jt] * 16 + j] : 0.f;
  }
  for (int i = tid; i < 2 * IMG; i += CL_WAVES * 64) img[i] = 0.f;

  const int q0 = (ug * tiles_per_wg) >> 2, nq = tiles_per_wg >> 2;  // own 1 KiB blocks of the image
  float* xb = jb.xbuf + (size_t)bg * 2 * IMG;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * IMG * 4, 0x00020000);

  float c[TPW];
  f32x4 zc[TPW], zn[TPW];
  auto loadz = [&](f32x4 (&z)[TPW], int t) {
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt)
      if (tv[jt]) z[jt] = *reinterpret_cast<const f32x4*>(Z + ((size_t)bc * T + t) * N + (tl[jt] * 4 + uq) * 4);
  };
#pragma unroll
  for (int jt = 0; jt < TPW; ++jt) {
    c[jt] = 0.f;
    zc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    zn[jt] = zc[jt];
  }
  loadz(zc, reverse ? T - 1 : 0);
  bool failed = false;
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    if (step + 1 < T) loadz(zn, reverse ? t - 1 : t + 1);
    f32x4 acc[TPW];
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) acc[jt] = zc[jt];
    const float* hb = img + cur * IMG;
    float* hn = img + (cur ^ 1) * IMG;
    const int slot = step & 1;
    const unsigned par = (((unsigned)step >> 1) & 1u) ^ 1u;
    if (wave < nw) {
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        f32x4 hv = *reinterpret_cast<const f32x4*>(hb + ((q * 4 + uq) * 16 + j) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * q + r < KS) {
#pragma unroll
            for (int jt = 0; jt < TPW; ++jt)
              if (tv[jt]) acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[jt][4 * q + r], hv[r], acc[jt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int jt = 0; jt < TPW; ++jt) {
        if (tv[jt]) {
          const int tile = tl[jt];
          const int unit = tile * 4 + uq;
          float4 g4;
          float h = mgr_cell_fwd(acc[jt][0], acc[jt][1], acc[jt][2], acc[jt][3], c[jt], g4);
          const int idx = (((tile >> 2) * 4 + uq) * 16 + j) * 4 + (tile & 3);
          if (G > 1) {
            const unsigned hbits = (__float_as_uint(h) & ~1u) | par;  // epoch parity rides in the mantissa LSB
            h = __uint_as_float(hbits);
            if (step + 1 < T) __builtin_amdgcn_raw_buffer_store_b32(hbits, rs, (slot * IMG + idx) * 4, 0, 16);  // sc1
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
    }
#pragma unroll
    for (int jt = 0; jt < TPW; ++jt) zc[jt] = zn[jt];
    if (G > 1 && step + 1 < T) {
      // gather: wave w sweeps blocks w, w+8, w+16, w+24 of the exchange slot until every word has this epoch's parity
      u32x4 v[4];
      unsigned pend = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int q = wave + CL_WAVES * i;
        if (q < QN && (q < q0 || q >= q0 + nq)) pend |= 1u << i;
      }
      unsigned spins = 0;
      while (pend && !failed) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (pend & (1u << i))
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * IMG + (wave + CL_WAVES * i) * 256 + lane * 4) * 4, 0, 16);  // sc1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (pend & (1u << i)) {
            const int q = wave + CL_WAVES * i;
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
            if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    __syncthreads();  // next image complete (own slice + gathered peers); everyone is done reading the current one
    cur ^= 1;
  }
}

__global__ __launch_bounds__(CL_WAVES * 64) void k_scan_cluster(ClusterLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  int ji = 0;
  for (int k = 1; k < L.njobs; ++k)
    if (bid >= L.job[k].wg_begin) ji = k;
  const ClusterJob& jb = L.job[ji];
  const int wg = bid - jb.wg_begin;
  if (wg >= jb.G_ * jb.nbg) return;
#define CL_CASE(KS, TPW) \
  if (jb.ks == KS && jb.tpw == TPW) return cluster_run<KS, TPW>(jb, wg, smem, L.status);
  CL_CASE(125, 1)
  CL_CASE(75, 1)
  CL_CASE(75, 2)
  CL_CASE(32, 1)
  CL_CASE(32, 2)
  CL_CASE(25, 1)
  CL_CASE(25, 2)
  CL_CASE(8, 1)
  CL_CASE(8, 2)
  CL_CASE(3, 1)
#undef CL_CASE
}

}  // namespace

bool mgr_cluster_supported(int ks, int tpw) {
  switch (ks) {
    case 125: return tpw == 1;
    case 3: return tpw == 1;
    case 75: case 32: case 25: case 8: return tpw == 1 || tpw == 2;
    default: return false;
  }
}

int mgr_cluster_launch(mgr_ctx* c, const ClusterLaunch& L, int total_wgs) {
  int maxks = 0;
  for (int i = 0; i < L.njobs; ++i) maxks = L.job[i].ks > maxks ? L.job[i].ks : maxks;
  size_t img = (size_t)((maxks + 3) / 4) * 256 * sizeof(float);
  size_t lds = 2 * img;
  if (lds < 84 * 1024) lds = 84 * 1024;  // > half of the 160 KiB LDS: at most one workgroup per CU
  MGR_REQUIRE(total_wgs <= c->cu_count, "cluster scan needs %d co-resident workgroups but the device has %d CUs", total_wgs, c->cu_count);
  static bool attr_set = false;
  if (!attr_set) {
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_scan_cluster, dim3(total_wgs), dim3(CL_WAVES * 64), lds, mgr_stream(c), L);
  MGR_LAUNCH_CHECK();
  return 0;
}
