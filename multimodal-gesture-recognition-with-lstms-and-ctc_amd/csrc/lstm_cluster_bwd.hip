// K7-scan, multi-CU: BPTT of LSTM directions spread over clusters of CUs (the backward twin of lstm_cluster.hip).
//
// dh_rec_{t-1}[unit, sample] = sum over the 4H packed gate columns of U[unit, col] * dz_t[sample, col].
// A cluster = the G = ceil(H/16) workgroups serving one (direction, 16-sample batch group); workgroup `ug` owns the
// 16 output units [16*ug, 16*ug+16) = ONE MFMA M-tile (v_mfma_f32_16x16x4_f32, M = units, N = samples, K = 4 gate
// columns = one unit's i,f,c,o).  Its 4 waves split the K loop (H k-steps) four ways with the U fragments stationary
// in VGPRs, reduce the four partial tiles through LDS, and then every thread runs the cell backward for ONE
// (unit, sample): 256 threads = 16 units x 16 samples.  dz_t is published to the cluster with the same
// data-is-the-flag write-through hand-off as the forward kernel (epoch parity in the mantissa LSB of every dz word)
// and gathered into the next LDS image [unit/4][gate][sample][unit%4].  Saved forward state (gates, c) and dY are
// prefetched two steps ahead through a 3-deep register ring.  Bounded spins, status word, one launch for all
// concurrently scanned directions (co-residency by construction).
#include "lstm_cluster.h"
#include "lstm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned POLL_LIMIT = 1u << 20;
constexpr int BW_WAVES = 4;

template <int H>
__device__ __forceinline__ void cluster_bwd_run(const ClusterBwdJob& jb, int bg, int ug, int cl, unsigned* xcc, int xcd_local,
                                                float* smem, unsigned* status) {
  constexpr int N = 4 * H;
  constexpr int QN = (H + 3) / 4;      // image blocks (1 KiB each): 4 units x 4 gates x 16 samples
  constexpr int BQ = (QN + 3) / 4;     // image blocks per wave: the K loop (H k-steps = QN blocks of 4) is split over 4 waves
  constexpr int KQ = 4 * BQ;           // k-steps per wave
  constexpr int IMG = 4 * BQ * 256;    // image padded to whole per-wave ranges (padding stays zero)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, uq = lane >> 4;
  const int G = jb.G_;
  const bool fast = xcd_local && mgr_cluster_same_xcd(xcc, blockIdx.x, jb.cls_begin, jb.cls_nclusters, cl, G, status);  // opt-in, see lstm_cluster.hip
  const int B = jb.B, T = jb.T, reverse = jb.reverse;
  const int b = bg * 16 + j;
  const bool bvalid = b < B;
  const int bc = bvalid ? b : B - 1;
  float* img = smem;                       // [2][IMG] dz images
  float* red = smem + 2 * IMG;             // [4 waves][4 regs][64 lanes] partial tiles

  // A fragment of k-step s (unit s): A[i = lane&15][kk = lane>>4] = Up[unit 16*ug+i][4s + kk]; this wave owns
  // k-steps s = wave*KQ + k
  float uf[KQ];
  {
    const int ur = ug * 16 + j;
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      const int s = wave * KQ + k;
      uf[k] = (ur < H && s < H) ? jb.Up[(size_t)ur * N + 4 * s + uq] : 0.f;
    }
  }
  for (int i = tid; i < 2 * IMG; i += BW_WAVES * 64) img[i] = 0.f;

  // this thread's (unit, sample) for the cell backward: unit = 16*ug + 4*uq + wave  (D row = 4*(lane>>4) + reg, reg = wave)
  const int unit = ug * 16 + uq * 4 + wave;
  const bool uvalid = unit < H;
  const int q0 = ug * 4;  // own image blocks [q0, q0+4)
  float* xb = jb.xbuf + (size_t)bg * 2 * IMG;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 2 * IMG * 4, 0x00020000);

  struct Saved {
    float dy, c;
    float4 g;
  };
  Saved r0, r1, r2;
  r0.dy = r0.c = r1.dy = r1.c = r2.dy = r2.c = 0.f;
  r0.g = r1.g = r2.g = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load = [&](Saved& sv, int k) {
    if (uvalid && k < T) {
      const int n = T - 1 - k;
      const int t = reverse ? T - 1 - n : n;
      const size_t row = (size_t)bc * T + t;
      sv.dy = jb.dY[row * jb.lddy + unit];
      sv.g = *reinterpret_cast<const float4*>(jb.gates + (row * H + unit) * 4);
      sv.c = jb.cs[row * H + unit];
    }
  };
  load(r0, 0);
  load(r1, 1);
  float dcc = 0.f;
  bool failed = false;
  __syncthreads();
  int cur = 0;

  auto do_step = [&](int k, Saved& use, Saved& prev, Saved& ld) {
    const int n = T - 1 - k;
    const int t = reverse ? T - 1 - n : n;
    const bool has_prev = n > 0;
    load(ld, k + 2);
    const float* db = img + cur * IMG;
    float* dn = img + (cur ^ 1) * IMG;
    const int slot = k & 1;
    const unsigned par = (((unsigned)k >> 1) & 1u) ^ 1u;
    // ---- 1. dh_rec from the previous step's dz (zero at the first iteration: image is zero-initialised)
    float dhr = 0.f;
    if (k > 0) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      // this wave's k-steps are the blocks q = wave*BQ + bi (4 k-steps each): static fragment indices
      constexpr int PD = 3;
      f32x4 dbuf[4];
      const float* dlane = db + (uq * 16 + j) * 4 + (size_t)wave * BQ * 256;
#pragma unroll
      for (int bi = 0; bi < PD && bi < BQ; ++bi) dbuf[bi] = *reinterpret_cast<const f32x4*>(dlane + bi * 256);
#pragma unroll
      for (int bi = 0; bi < BQ; ++bi) {
        if (bi + PD < BQ) dbuf[(bi + PD) & 3] = *reinterpret_cast<const f32x4*>(dlane + (bi + PD) * 256);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 dv = dbuf[bi & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r & 1)
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * bi + r], dv[r], a1, 0, 0, 0);
          else
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[4 * bi + r], dv[r], a0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      a0 += a1;
      // ---- 2. reduce the four K-slices through LDS
      *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = a0;
      __syncthreads();
      dhr = red[(0 * 64 + lane) * 4 + wave] + red[(1 * 64 + lane) * 4 + wave] + red[(2 * 64 + lane) * 4 + wave] +
            red[(3 * 64 + lane) * 4 + wave];
    }
    // ---- 3. cell backward for (unit, sample)
    float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (uvalid) {
      const float dh = use.dy + dhr;
      const float cp = has_prev ? prev.c : 0.f;
      dz = mgr_cell_bwd(dh, use.g, use.c, cp, dcc);
      if (bvalid) *reinterpret_cast<float4*>(jb.dZ + ((size_t)b * T + t) * N + unit * 4) = dz;
    }
    {
      // image [q = unit>>2][gate][j][r = unit&3]
      const int base = ((unit >> 2) * 4 * 16 + j) * 4 + (unit & 3);
      float v[4] = {dz.x, dz.y, dz.z, dz.w};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float val = v[g];
        const int idx = base + g * 64;
        if (G > 1) {
          const unsigned bits = (__float_as_uint(val) & ~1u) | par;
          val = __uint_as_float(bits);
          if (has_prev && uvalid) {
            if (fast)  // cluster on one XCD: plain store into the shared L2
              *reinterpret_cast<volatile unsigned*>(xb + slot * IMG + idx) = bits;
            else
              __builtin_amdgcn_raw_buffer_store_b32(bits, rs, (slot * IMG + idx) * 4, 0, 16);  // sc1 write-through
          }
        }
        if (uvalid) dn[idx] = val;
      }
    }
    // ---- 4. gather the peers' dz blocks
    if (G > 1 && has_prev) {
      constexpr int NF = 8;
      for (int base = 0; base < QN && !failed; base += NF * BW_WAVES) {
        u32x4 v[NF];
        unsigned pend = 0;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          int q = base + wave + BW_WAVES * i;
          if (q < QN && (q < q0 || q >= q0 + 4)) pend |= 1u << i;
        }
        unsigned spins = 0;
        while (pend && !failed) {
#pragma unroll
          for (int i = 0; i < NF; ++i)
            if (pend & (1u << i))
              v[i] = fast ? __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * IMG + (base + wave + BW_WAVES * i) * 256 + lane * 4) * 4, 0, 2)
                          : __builtin_amdgcn_raw_buffer_load_b128(rs, (slot * IMG + (base + wave + BW_WAVES * i) * 256 + lane * 4) * 4, 0, 16);
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            if (pend & (1u << i)) {
              const int q = base + wave + BW_WAVES * i;
              const int nvalid = H - 4 * q;  // units of this block that exist
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
                *reinterpret_cast<u32x4*>(dn + q * 256 + lane * 4) = v[i];
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
    }
    __syncthreads();
    cur ^= 1;
  };

  for (int k0 = 0; k0 < T; k0 += 3) {
    do_step(k0, r0, r1, r2);
    if (k0 + 1 < T) do_step(k0 + 1, r1, r2, r0);
    if (k0 + 2 < T) do_step(k0 + 2, r2, r0, r1);
  }
}

#define BW_FOREACH(X) X(8) X(16) X(32) X(64) X(100) X(128)

__global__ __launch_bounds__(BW_WAVES * 64) void k_scan_cluster_bwd(ClusterBwdLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  for (int k = 0; k < L.njobs; ++k) {
    const ClusterBwdJob& jb = L.job[k];
    const int w = bid - jb.cls_begin;
    if (w < 0 || w >= jb.cls_nclusters * jb.G_) continue;
    // members of a cluster are CONTIGUOUS workgroup ids by default (the round-robin dispatcher then spreads them over
    // all XCDs, which measured best for the write-through exchange); the XCD-local experiment interleaves them instead
    const int cl = L.xcd_local ? w % jb.cls_nclusters : w / jb.G_;
    const int ug = L.xcd_local ? w / jb.cls_nclusters : w % jb.G_;
    const int bg = cl - jb.cls_cluster0;
    if (bg < 0 || bg >= jb.nbg) continue;
#define BW_CASE(HH) \
  if (jb.H == HH) return cluster_bwd_run<HH>(jb, bg, ug, cl, L.xcc, L.xcd_local, smem, L.status);
    BW_FOREACH(BW_CASE)
#undef BW_CASE
    return;
  }
}

}  // namespace

size_t mgr_cluster_bwd_img_floats(int H) {
  int qn = (H + 3) / 4, bq = (qn + 3) / 4;
  return (size_t)4 * bq * 256;
}

bool mgr_cluster_bwd_supported(int H) {
#define BW_CASE(HH) \
  if (H == HH) return true;
  BW_FOREACH(BW_CASE)
#undef BW_CASE
  return false;
}

int mgr_cluster_bwd_launch(mgr_ctx* c, const ClusterBwdLaunch& L, int total_wgs) {
  int maxH = 0;
  for (int i = 0; i < L.njobs; ++i) maxH = L.job[i].H > maxH ? L.job[i].H : maxH;
  size_t lds = ((size_t)2 * mgr_cluster_bwd_img_floats(maxH) + 4 * 64 * 4) * sizeof(float);
  MGR_REQUIRE(total_wgs <= 2 * c->cu_count, "cluster BPTT needs %d co-resident workgroups", total_wgs);
  static bool attr_set = false;
  if (!attr_set) {
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_cluster_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_scan_cluster_bwd, dim3(total_wgs), dim3(BW_WAVES * 64), lds, mgr_stream(c), L);
  MGR_LAUNCH_CHECK();
  return 0;
}
