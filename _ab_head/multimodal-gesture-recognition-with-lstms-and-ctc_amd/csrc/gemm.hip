// K2 / K7: the dense contractions of the LSTM layers on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   nn : Z[b,t,:]   = (X[b,t,:] (.) mask4[g,b,:]) . Wp + bp            (input projection, all T at once)
//   tn : dWp / dUp  = sum_{b,t} A[b,t,:]^T dZ[b,t,:]                   (parameter gradients, split over samples)
//   nt : dX[b,t,:]  = sum_g mask4[g,b,:] (.) (dZ_g[b,t,:] . Wp_g^T)    (gradient to the layer below)
//
// One 256-thread workgroup (4 waves, 2x2) computes a 128x128 output tile, each wave a 64x64 sub-tile as 2x2
// MFMA 32x32 tiles (64 accumulator VGPRs); K advances 16 per LDS stage (8 MFMA k-steps).  Both operands are
// staged k-major in LDS (As[k][m], Bs[k][n]) so an MFMA fragment read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32).  Global->register prefetch of the next stage overlaps the MFMAs of the current.
// The Keras per-gate input-dropout masks are folded into the B-operand staging (nn, nt: one sample per row
// tile) or the epilogue (tn), so no masked copy of X is ever materialised.  The gate of a packed column is col&3.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef MGR_GEMM_BK
#define MGR_GEMM_BK 16
#endif
#ifndef MGR_GEMM_NBUF
#define MGR_GEMM_NBUF 2
#endif
constexpr int BM = 128, BN = 128, BK = MGR_GEMM_BK, NBUF = MGR_GEMM_NBUF, LDS_LD = BM + 4;
constexpr int LPT = BK / 8;  // float4 loads per thread per operand per stage (256 threads)

// element (m,k) at base[m*ld + k]  ("k-contiguous"); thread handles 2 float4 along k
struct RegTile {
  float v[4 * LPT];
};

__device__ __forceinline__ void load_kc(RegTile& r, const float* __restrict__ base, size_t ld, int m0, int k0, int Mlim,
                                        int Klim, bool vec, int tid) {
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    int idx4 = tid + i * 256;
    int m = idx4 / (BK / 4), k4 = (idx4 % (BK / 4)) * 4;
    int gm = m0 + m, gk = k0 + k4;
    const float* p = base + (size_t)gm * ld + gk;
    if (vec && gm < Mlim && gk + 3 < Klim) {
      float4 t = *reinterpret_cast<const float4*>(p);
      r.v[i * 4 + 0] = t.x;
      r.v[i * 4 + 1] = t.y;
      r.v[i * 4 + 2] = t.z;
      r.v[i * 4 + 3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) r.v[i * 4 + e] = (gm < Mlim && gk + e < Klim) ? p[e] : 0.f;
    }
  }
}
__device__ __forceinline__ void store_kc(float (*S)[LDS_LD], const RegTile& r, int tid) {
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    int idx4 = tid + i * 256;
    int m = idx4 / (BK / 4), k4 = (idx4 % (BK / 4)) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) S[k4 + e][m] = r.v[i * 4 + e];
  }
}
// element (m,k) at base[k*ld + m]  ("m-contiguous"); thread handles 2 float4 along m.
// kshift/Klo: row index = k0+k+kshift must lie in [Klo,Klim) else zero (time-shifted h_prev view)
__device__ __forceinline__ void load_mc(RegTile& r, const float* __restrict__ base, size_t ld, int m0, int k0, int Mlim,
                                        int Klim, bool vec, int tid, int kshift = 0) {
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    int idx4 = tid + i * 256;
    int k = idx4 >> 5, m4 = (idx4 & 31) * 4;
    int gm = m0 + m4, gk = k0 + k + kshift;
    bool kv = (k0 + k) < Klim && gk >= 0 && gk < Klim;
    const float* p = base + (ptrdiff_t)gk * (ptrdiff_t)ld + gm;
    if (vec && kv && gm + 3 < Mlim) {
      float4 t = *reinterpret_cast<const float4*>(p);
      r.v[i * 4 + 0] = t.x;
      r.v[i * 4 + 1] = t.y;
      r.v[i * 4 + 2] = t.z;
      r.v[i * 4 + 3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) r.v[i * 4 + e] = (kv && gm + e < Mlim) ? p[e] : 0.f;
    }
  }
}
__device__ __forceinline__ void store_mc(float (*S)[LDS_LD], const RegTile& r, int tid) {
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    int idx4 = tid + i * 256;
    int k = idx4 >> 5, m4 = (idx4 & 31) * 4;
    *reinterpret_cast<float4*>(&S[k][m4]) = make_float4(r.v[i * 4 + 0], r.v[i * 4 + 1], r.v[i * 4 + 2], r.v[i * 4 + 3]);
  }
}

// Branch-free interior loaders: out-of-range M indices are CLAMPED (they read valid memory; the rows / columns they
// feed are never stored), so the main loop carries no exec-mask games and its loads stay in flight under the MFMAs.
__device__ __forceinline__ void load_kc_fast(RegTile& r, const float* __restrict__ base, size_t ld, int m0, int k0, int Mlim,
                                             int tid) {
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    int idx4 = tid + i * 256;
    int m = idx4 / (BK / 4), k4 = (idx4 % (BK / 4)) * 4;
    int gm = m0 + m;
    gm = gm < Mlim ? gm : Mlim - 1;
    float4 t = *reinterpret_cast<const float4*>(base + (size_t)gm * ld + k0 + k4);
    r.v[i * 4 + 0] = t.x;
    r.v[i * 4 + 1] = t.y;
    r.v[i * 4 + 2] = t.z;
    r.v[i * 4 + 3] = t.w;
  }
}
__device__ __forceinline__ void load_mc_fast(RegTile& r, const float* __restrict__ base, size_t ld, int m0, int k0, int Mlim,
                                             int tid, int kshift = 0) {
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    int idx4 = tid + i * 256;
    int k = idx4 >> 5, m4 = (idx4 & 31) * 4;
    int gm = m0 + m4;
    gm = gm + 3 < Mlim ? gm : Mlim - 4;
    float4 t = *reinterpret_cast<const float4*>(base + (ptrdiff_t)(k0 + k + kshift) * (ptrdiff_t)ld + gm);
    r.v[i * 4 + 0] = t.x;
    r.v[i * 4 + 1] = t.y;
    r.v[i * 4 + 2] = t.z;
    r.v[i * 4 + 3] = t.w;
  }
}

// Two-matrix view of the m-contiguous operand: columns [0, nd) come from base, columns [nd, 2 nd) from base2 (both with row
// stride nd).  Used to run the two directions of a Bidirectional layer as ONE GEMM over N = 8H columns: for 4H = 400 that
// is 7 column tiles instead of 2 x 4 (the fourth tile of a 400-column matrix is 12.5 % used).  nd is a multiple of 4.
__device__ __forceinline__ void load_mc_fast2(RegTile& r, const float* __restrict__ base, const float* __restrict__ base2, int nd,
                                              int m0, int k0, int tid, int kshift = 0) {
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    int idx4 = tid + i * 256;
    int k = idx4 >> 5, m4 = (idx4 & 31) * 4;
    int gm = m0 + m4;
    gm = gm + 3 < 2 * nd ? gm : 2 * nd - 4;
    const float* src = gm >= nd ? base2 + (gm - nd) : base + gm;
    float4 t = *reinterpret_cast<const float4*>(src + (ptrdiff_t)(k0 + k + kshift) * (ptrdiff_t)nd);
    r.v[i * 4 + 0] = t.x;
    r.v[i * 4 + 1] = t.y;
    r.v[i * 4 + 2] = t.z;
    r.v[i * 4 + 3] = t.w;
  }
}

__device__ __forceinline__ void mma_stage(const float (*As)[LDS_LD], const float (*Bs)[LDS_LD], f32x16 (&acc)[2][2], int wr,
                                          int wc, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < BK / 2; ++ks) {
    float a0 = As[ks * 2 + lh][wr * 64 + l31];
    float a1 = As[ks * 2 + lh][wr * 64 + 32 + l31];
    float b0 = Bs[ks * 2 + lh][wc * 64 + l31];
    float b1 = Bs[ks * 2 + lh][wc * 64 + 32 + l31];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// accumulator element -> (row, col) inside the 128x128 tile
#define ACC_ROW(wr, mt, reg, lane) ((wr) * 64 + (mt) * 32 + ((reg) & 3) + 8 * ((reg) >> 2) + 4 * ((lane) >> 5))
#define ACC_COL(wc, nt, lane) ((wc) * 64 + (nt) * 32 + ((lane) & 31))

// ------------------------------------------------------------------------------------------------ nn
// grid: (ceil(N/128), ceil(T/128), B)
__global__ __launch_bounds__(256, 2) void k_gemm_nn(const float* __restrict__ X, int ldx, const float* __restrict__ mask4,
                                                 const float* __restrict__ Wp, const float* __restrict__ bp,
                                                 float* __restrict__ Z, int B, int T, int F, int N, int vecA) {
  __shared__ __attribute__((aligned(16))) float As2[NBUF][BK][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs2[NBUF][BK][LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * BN, r0 = blockIdx.y * BM, b = blockIdx.z;
  const float* Xb = X + (size_t)b * T * ldx;
  f32x16 acc[2][2];
  zero_acc(acc);
  RegTile ra, rb, rm;
  // The mask factors are only LOADED at fetch time and multiplied in at the LDS-store phase, so that the prefetch of
  // the next stage stays in flight under the current stage's MFMAs (a multiply at fetch would force vmcnt(0) at once).
  auto fetch_mask = [&](int k0) {
    if (mask4) {
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        int idx4 = tid + i * 256;
        int k = k0 + (idx4 >> 5);
        k = k < F ? k : F - 1;
        // n is a multiple of 4: the 4 lanes of the float4 are gates 0..3 of one unit
#pragma unroll
        for (int g = 0; g < 4; ++g) rm.v[i * 4 + g] = mask4[((size_t)g * B + b) * F + k];
      }
    }
  };
  auto stash = [&](int buf) {
    if (mask4) {
#pragma unroll
      for (int e = 0; e < 4 * LPT; ++e) rb.v[e] *= rm.v[e];
    }
    store_kc(As2[buf], ra, tid);
    store_mc(Bs2[buf], rb, tid);
  };
  // interior K stages: software-pipelined, branch-free loads (separate code from the guarded tail so that no
  // register-merging moves - and hence no early vmcnt waits - appear between the loads and the MFMAs)
  const int nfast = vecA ? F / BK : 0;
  if (nfast > 0) {
    load_kc_fast(ra, Xb, (size_t)ldx, r0, 0, T, tid);
    load_mc_fast(rb, Wp, (size_t)N, n0, 0, N, tid);
    fetch_mask(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt < nfast; ++kt) {
      const bool more = kt + 1 < nfast;
      if (more) {
        load_kc_fast(ra, Xb, (size_t)ldx, r0, (kt + 1) * BK, T, tid);
        load_mc_fast(rb, Wp, (size_t)N, n0, (kt + 1) * BK, N, tid);
        fetch_mask((kt + 1) * BK);
      }
      mma_stage(As2[buf], Bs2[buf], acc, wr, wc, lane);
      if (NBUF == 1) __syncthreads();
      if (more) stash(buf ^ (NBUF - 1));  // NBUF == 2: the other buffer was last read before the previous barrier
      __syncthreads();
      buf ^= (NBUF - 1);
    }
  }
  // guarded stages (K tail, or everything when the operands are not 16-byte aligned), not pipelined
  for (int k0 = nfast * BK; k0 < F; k0 += BK) {
    load_kc(ra, Xb, (size_t)ldx, r0, k0, T, F, vecA != 0, tid);
    load_mc(rb, Wp, (size_t)N, n0, k0, N, F, true, tid);
    fetch_mask(k0);
    stash(0);
    __syncthreads();
    mma_stage(As2[0], Bs2[0], acc, wr, wc, lane);
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      int col = n0 + ACC_COL(wc, nt, lane);
      float bias = (col < N) ? bp[col] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int r = r0 + ACC_ROW(wr, mt, reg, lane);
        if (r < T && col < N) Z[((size_t)b * T + r) * N + col] = acc[mt][nt][reg] + bias;
      }
    }
}


// nn over both directions of a Bidirectional layer: columns [0,Nd) -> (Wp, bp, mask4, Z), [Nd, 2Nd) -> (Wp2, bp2, mask4b, Z2).
// Requires F % BK == 0 and 16-byte aligned rows (the host falls back to two k_gemm_nn launches otherwise).
// grid: (ceil(2Nd/128), ceil(T/128), B)
__global__ __launch_bounds__(256, 4) void k_gemm_nn2(const float* __restrict__ X, int ldx, const float* __restrict__ mask4,
                                                  const float* __restrict__ mask4b, const float* __restrict__ Wp,
                                                  const float* __restrict__ Wp2, const float* __restrict__ bp,
                                                  const float* __restrict__ bp2, float* __restrict__ Z, float* __restrict__ Z2,
                                                  int B, int T, int F, int Nd) {
  __shared__ __attribute__((aligned(16))) float As2[NBUF][BK][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs2[NBUF][BK][LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * BN, r0 = blockIdx.y * BM, b = blockIdx.z;
  const float* Xb = X + (size_t)b * T * ldx;
  f32x16 acc[2][2];
  zero_acc(acc);
  RegTile ra, rb, rm;
  const bool masked = mask4 != nullptr;
  // this thread's B columns (4 gates of one unit) all belong to one direction
  const float* mk = (n0 + (tid & 31) * 4 >= Nd) ? mask4b : mask4;
  auto fetch_mask = [&](int k0) {
    if (masked) {
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        int k = k0 + ((tid + i * 256) >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) rm.v[i * 4 + g] = mk[((size_t)g * B + b) * F + k];
      }
    }
  };
  auto stash = [&](int buf) {
    if (masked) {
#pragma unroll
      for (int e = 0; e < 4 * LPT; ++e) rb.v[e] *= rm.v[e];
    }
    store_kc(As2[buf], ra, tid);
    store_mc(Bs2[buf], rb, tid);
  };
  const int nst = F / BK;
  load_kc_fast(ra, Xb, (size_t)ldx, r0, 0, T, tid);
  load_mc_fast2(rb, Wp, Wp2, Nd, n0, 0, tid);
  fetch_mask(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nst; ++kt) {
    const bool more = kt + 1 < nst;
    if (more) {
      load_kc_fast(ra, Xb, (size_t)ldx, r0, (kt + 1) * BK, T, tid);
      load_mc_fast2(rb, Wp, Wp2, Nd, n0, (kt + 1) * BK, tid);
      fetch_mask((kt + 1) * BK);
    }
    mma_stage(As2[buf], Bs2[buf], acc, wr, wc, lane);
    if (NBUF == 1) __syncthreads();
    if (more) stash(buf ^ (NBUF - 1));
    __syncthreads();
    buf ^= (NBUF - 1);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      int col = n0 + ACC_COL(wc, nt, lane);
      const bool second = col >= Nd;
      const int cd = second ? col - Nd : col;
      float* Zd = second ? Z2 : Z;
      float bias = (col < 2 * Nd) ? (second ? bp2 : bp)[cd] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int r = r0 + ACC_ROW(wr, mt, reg, lane);
        if (r < T && col < 2 * Nd) Zd[((size_t)b * T + r) * Nd + cd] = acc[mt][nt][reg] + bias;
      }
    }
}


// A stated bound on |X| that the data violate (mgr.h, x_absmax): k_absmax_gate looks at the operand BEFORE the split-f16 kernel that
// trusts the bound, and raises a device word when a scaled value would leave the f16 range (the split would hold Inf).  Both kernels
// of the call are enqueued: the split-f16 one returns at once when the word is raised, the f32 MFMA one when it is not - the
// decision never travels to the host.  (NaN / Inf inputs are not violations: both kernels carry them into the output.)
#define MGR_GATED(gate, run_if_raised) \
  if ((gate) && ((*(gate) != 0u) != (run_if_raised))) return
__global__ __launch_bounds__(256) void k_absmax_gate(const float* __restrict__ X, size_t n4, float limit, unsigned* __restrict__ gate) {
  bool over = false;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(X)[i];
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    over = over || (m > limit && m < __uint_as_float(0x7F800000u));
  }
  if (__any(over) && (threadIdx.x & 63) == 0) atomicOr(gate, 1u);
}

// ------------------------------------------------------------------------------------------------ nn, dropout-aware
// Keras input dropout zeroes a fraction p of the input features per (gate, sample) (speech_lstm_ctc_words.py:61,73:
// p = 0.4 / 0.5; skeletal_lstm_ctc.py:313,327: 0.6; multimodal.py:159-168: 0.5): (x (.) m_g) . W_g only needs the kept
// features.  k_mask_compact lists, per (gate, sample), the kept feature indices (ascending) followed by the dropped ones;
// k_gemm_nn_sparse then runs ONE K LOOP PER GATE over the kept indices only (rounded up to a stage of 16 with dropped
// ones, whose mask factor is 0): X columns and W rows are gathered by index while staging into LDS, each gate pass fills
// its own accumulators for the same (row, unit) positions, and the epilogue stores the four gates of a unit as one float4
// in the packed order the scans read.  At p = 0.5 that is half the MFMA work of the dense kernel for the same result.
__global__ __launch_bounds__(64) void k_mask_compact(const float* __restrict__ mask4, int F, int Fp, int* __restrict__ kidx,
                                                     float* __restrict__ kval, int* __restrict__ kcnt,
                                                     int* __restrict__ kpos /* [4B][F] list position of a kept feature, -1 if dropped; may be null */,
                                                     unsigned* __restrict__ zero_word /* set to 0 (the max |W| word k_gate_major fills next); may be null */) {
  const int gb = blockIdx.x, lane = threadIdx.x;
  if (zero_word && gb == 0 && lane == 0) *zero_word = 0u;
  const float* m = mask4 + (size_t)gb * F;
  int* out = kidx + (size_t)gb * Fp;
  float* val = kval + (size_t)gb * Fp;
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) {   // kept features first, then the dropped ones (factor 0: stage padding)
    for (int f0 = 0; f0 < F; f0 += 64) {
      const int f = f0 + lane;
      const float v = f < F ? m[f] : 0.f;
      const bool take = f < F && ((v != 0.f) == (pass == 0));
      const unsigned long long bal = __ballot(take);
      if (take) {
        const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
        out[pos] = f;
        val[pos] = v;
        if (kpos) kpos[(size_t)gb * F + f] = pass == 0 ? pos : -1;
      }
      n += __popcll(bal);
    }
    if (pass == 0 && lane == 0) kcnt[gb] = n;
  }
  for (int i = F + lane; i < Fp; i += 64) {   // F not a multiple of the stage depth: the last stage is filled up with zero terms
    out[i] = 0;
    val[i] = 0.f;
  }
}

// the lists of a projection WITHOUT a mask: every feature kept with factor 1 (the f32 kernel as the plain dense projection)
__global__ __launch_bounds__(64) void k_mask_all(int F, int Fp, int* __restrict__ kidx, float* __restrict__ kval, int* __restrict__ kcnt,
                                                 unsigned* __restrict__ zero_word) {
  const int gb = blockIdx.x, lane = threadIdx.x;
  if (zero_word && gb == 0 && lane == 0) *zero_word = 0u;
  for (int i = lane; i < Fp; i += 64) {
    kidx[(size_t)gb * Fp + i] = i < F ? i : 0;
    kval[(size_t)gb * Fp + i] = i < F ? 1.f : 0.f;
  }
  if (lane == 0) kcnt[gb] = F;
}

// Wg[g][f][u] = Wp[f][4u + g]: gate-major copy of the packed kernel, so that the row gather of one gate pass reads
// contiguous units instead of every fourth float (a quarter of the L2 traffic of the B operand); F x 4H floats per call.
// wmax (may be null): the largest |W| of the call as float bits, by atomic max (zeroed by k_mask_compact in front) - the scale of
// the split-f16 projection.
__global__ __launch_bounds__(256) void k_gate_major(const float* __restrict__ Wp, float* __restrict__ Wg, int F, int H, unsigned* __restrict__ wmax) {
  const size_t n = (size_t)F * H;
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 w = *reinterpret_cast<const float4*>(Wp + i * 4);   // (f, u): gates 0..3
    Wg[i] = w.x;
    Wg[n + i] = w.y;
    Wg[2 * n + i] = w.z;
    Wg[3 * n + i] = w.w;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
  }
  if (wmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // (a NaN weight does not survive fmaxf; Inf does: the projection then scales by 0 and its output is NaN - visible)
    if ((threadIdx.x & 63) == 0) atomicMax(wmax, __float_as_uint(m));
  }
}

constexpr int SP_MAXF = 2048;   // feature-count limit of the sparse kernel (index + factor lists of one gate pass live in LDS)
constexpr int SP_TM = 128, SP_SK = 16;   // tile: 128 rows x (32 WC) units x 4 gates, 16 k per stage

// One workgroup per tile, units fastest (workgroups that hold a CU for the whole kernel were measured: no faster alone -
// the float4 stores of a tile are 0.1 of 3.3 ms - and they starve the small kernels of the other stream).  Global loads
// run two stages ahead of the MFMAs (two register sets): a gathered element costs an LDS index read plus a scattered
// 4-byte global load.
// WC = waves along the units: 2 (256 threads, 64 units, two workgroups per CU) or 4 (512 threads, 128 units, one per CU:
// the X tile is shared by twice the units, i.e. half the A-operand traffic per FLOP).
// TR: X is the TRANSPOSED activation copy XT[b][f][t] (row stride ldx = padded T, mgr_transpose_bt): a kept feature is then a
// contiguous ROW of 128 time steps and the A stage is two coalesced float4 loads per thread that go to LDS as they are (the LDS
// image is k-major already), instead of eight scattered 4-byte loads that fetch a 128-byte line for 64 useful bytes.
// Measured and not kept (round 3, profiles/r03_gemm_sparse_probes.txt): TWO workgroups per tile with two gates each (64 instead of
// 128 accumulator VGPRs, four workgroups per CU): 2.74 ms against 2.47 ms at F = 1000, H = 500 - occupancy is not what this kernel lacks.
template <int WC, bool TR>
__global__ __launch_bounds__(128 * WC, WC == 2 ? 2 : 1) void k_gemm_nn_sparse(const float* __restrict__ X, int ldx, const int* __restrict__ kidx,
                                                        const float* __restrict__ kval, const int* __restrict__ kcnt,
                                                        const float* __restrict__ Wp, const float* __restrict__ bp,
                                                        float* __restrict__ Z, int B, int T, int Fp, int F, int H,
                                                        const unsigned* __restrict__ gate /* may be null: MGR_GATED */) {
  MGR_GATED(gate, true);
  constexpr int TM = SP_TM, TU = 32 * WC, SK = SP_SK, NT = 128 * WC;
  constexpr int QT = NT / 4;          // threads per k-quad of a stage: thread (quad q = tid / QT, r = tid % QT)
  constexpr int RPT = TM / QT;        // A rows per thread (r, r + QT, ...): 2 (WC = 2) or 1 (WC = 4)
  static_assert(TU == QT && SK == 16, "one B unit per thread and quad; 16 list positions per stage = 4 quads x 4 slots");
  // LDS images, K-QUAD-MAJOR (round 3): list position p of a stage lives at [quad p % 4][row / unit][slot p / 4].  The MFMA
  // k-steps are dealt so that a lane's operands of all eight steps are two whole quads: lane half lh takes position
  // (2 lh + ks / 4) + 4 (ks % 4) in step ks, i.e. quads 2 lh and 2 lh + 1 in slot order.  Fragment reads per stage and lane:
  // 6 ds_read_b128 instead of 24 ds_read_b32 (the k-major image needed one read per operand and step; which k meets which
  // step is free as long as A and B agree).  Staging: B, and A from the transposed copy: a thread holds the four positions of ONE
  // quad for its unit / its rows - four coalesced 4-byte loads, ONE ds_write_b128; A from the row-major input: a thread holds
  // ONE position for eight rows (16 neighbouring lanes read the 16 gathered features of one row) and writes eight dwords.
  __shared__ __attribute__((aligned(16))) float As[2][4][TM + 4][4];   // (+4 rows: the quads of a non-transposed stage land 16 banks apart)
  __shared__ __attribute__((aligned(16))) float Bs[2][4][TU][4];
  __shared__ unsigned short Ls[SP_MAXF];
  __shared__ float Vs[SP_MAXF];
  static_assert(SP_MAXF <= 65536, "feature indices are kept as 16-bit values in LDS");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave / WC, wc = wave % WC;
  const int N = 4 * H;
  const int q = tid / QT, r = tid % QT;      // B staging, and A staging from the transposed copy: quad, unit / row
  const int ak = tid & 15, ar = tid >> 4;    // A staging from the row-major input: list position, first row (then + NT / 16, ...)
  constexpr int ARS = NT / 16, APT = TM * SK / NT;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ncol = (H + TU - 1) / TU, nrow = (T + TM - 1) / TM;
  struct Regs {
    float a[APT], w[4], v[4];
  };
  static_assert(APT == RPT * 4, "both A staging forms hold TM * SK / NT elements per thread");
  // Workgroup -> tile, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the
  // ncol workgroups that share one (sample, row tile) of X - and read it four times each, once per gate pass - are the ids
  // x, x + 8, x + 16, ...: they meet in ONE L2 instead of pulling the same rows into all eight.
  {
    const int x = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int rt = (jj / ncol) * 8 + x;   // linear (sample, row tile)
    if (rt >= nrow * B) return;
    const int u0 = (jj % ncol) * TU, r0 = (rt % nrow) * TM, b = rt / nrow;
    // TR: X is the transposed copy XT[b][f][t] (row stride ldx = padded T): a kept feature is a contiguous row of time steps, the
    // 64 threads of a quad read 256 contiguous bytes of it; else X[b][t][f]: a kept feature is a column (scattered 4-byte loads)
    const float* Xb = TR ? X + (size_t)b * F * ldx + r0 + r : X + (size_t)b * T * ldx;
    const int ucl = (u0 + r < H) ? u0 + r : H - 1;
    f32x16 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][mt][e] = 0.f;
    int arow[APT];   // row-major input: element offsets of this thread's rows (clamped: rows >= T are computed but never stored)
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int row = r0 + ar + ARS * i;
      arow[i] = (row < T ? row : T - 1) * ldx;   // (one sample's [T, ldx] block stays below 2^31 elements)
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nst = (kcnt[g * B + b] + SK - 1) / SK;   // (<= Fp / 16)
      {
        const int* list = kidx + ((size_t)g * B + b) * Fp;
        const float* lval = kval + ((size_t)g * B + b) * Fp;
        for (int i = tid; i < nst * SK; i += NT) {
          Ls[i] = (unsigned short)list[i];
          Vs[i] = lval[i];
        }
      }
      __syncthreads();
      const float* Wg = Wp + (size_t)g * F * H + ucl;   // (Wp: the gate-major copy [4][F][H])
      auto fetch = [&](Regs& R, int st) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int p = st * SK + q + 4 * c;
          const int f = Ls[p];
          R.v[c] = Vs[p];
          R.w[c] = Wg[(size_t)f * H];
          if constexpr (TR) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) R.a[i * 4 + c] = Xb[(size_t)f * ldx + QT * i];
          }
        }
        if constexpr (!TR) {
          const float* xp = Xb + Ls[st * SK + ak];
#pragma unroll
          for (int i = 0; i < APT; ++i) R.a[i] = xp[arow[i]];
        }
      };
      auto stash = [&](const Regs& R, int buf) {
        if constexpr (TR) {
#pragma unroll
          for (int i = 0; i < RPT; ++i)
            *reinterpret_cast<float4*>(&As[buf][q][r + QT * i][0]) = make_float4(R.a[i * 4], R.a[i * 4 + 1], R.a[i * 4 + 2], R.a[i * 4 + 3]);
        } else {
#pragma unroll
          for (int i = 0; i < APT; ++i) As[buf][ak & 3][ar + ARS * i][ak >> 2] = R.a[i];
        }
        *reinterpret_cast<float4*>(&Bs[buf][q][r][0]) = make_float4(R.w[0] * R.v[0], R.w[1] * R.v[1], R.w[2] * R.v[2], R.w[3] * R.v[3]);
      };
      auto mma = [&](int buf) {
        const int ra = wr * 64 + l31, ub = wc * 32 + l31;
        const f32x4 a00 = *reinterpret_cast<const f32x4*>(&As[buf][2 * lh][ra][0]);
        const f32x4 a01 = *reinterpret_cast<const f32x4*>(&As[buf][2 * lh + 1][ra][0]);
        const f32x4 a10 = *reinterpret_cast<const f32x4*>(&As[buf][2 * lh][ra + 32][0]);
        const f32x4 a11 = *reinterpret_cast<const f32x4*>(&As[buf][2 * lh + 1][ra + 32][0]);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bs[buf][2 * lh][ub][0]);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bs[buf][2 * lh + 1][ub][0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00[c], b0[c], acc[g][0], 0, 0, 0);
          acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10[c], b0[c], acc[g][1], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[c], b1[c], acc[g][0], 0, 0, 0);
          acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11[c], b1[c], acc[g][1], 0, 0, 0);
        }
      };
      if (nst > 0) {
        Regs R0, R1;
        fetch(R0, 0);
        if (nst > 1) fetch(R1, 1);
        stash(R0, 0);
        __syncthreads();
        // stage st computes from LDS buffer st & 1; its data were fetched two iterations ago and stashed in the previous one
        for (int st = 0; st < nst; st += 2) {
          if (st + 2 < nst) fetch(R0, st + 2);
          mma(0);
          if (st + 1 < nst) stash(R1, 1);
          __syncthreads();
          if (st + 1 < nst) {
            if (st + 3 < nst) fetch(R1, st + 3);
            mma(1);
            if (st + 2 < nst) stash(R0, 0);
            __syncthreads();
          }
        }
      }
    }
    const int unit = u0 + wc * 32 + l31;
    if (unit < H) {
      const float4 bias = *reinterpret_cast<const float4*>(bp + unit * 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = r0 + wr * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
          if (row < T) {
            typedef float nt_f4 __attribute__((ext_vector_type(4)));
            const nt_f4 v = {acc[0][mt][reg] + bias.x, acc[1][mt][reg] + bias.y, acc[2][mt][reg] + bias.z, acc[3][mt][reg] + bias.w};
            __builtin_nontemporal_store(v, reinterpret_cast<nt_f4*>(Z + ((size_t)b * T + row) * N + unit * 4));
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ nn, dropout-aware, split-f16
// The same tile walk as k_gemm_nn_sparse<2, true> (transposed activations, 128 rows x 64 units x 4 gates, one K loop per gate over the
// kept features, 16 list positions per stage) on the f16 matrix pipe (round 4): every f32 operand goes to LDS as an f16 (hi, lo)
// pair of its scaled value - x sx = hi + lo to 22+ bits - and a stage is THREE v_mfma_f32_32x32x16_f16 per 32 x 32 block
// (hi hi + lo hi + hi lo, f32 accumulation; the dropped lo lo term is 2^-22 of the product) instead of sixteen
// v_mfma_f32_32x32x2_f32: 96 instead of 1024 matrix-pipe cycles per stage and block.  See lstm_cluster.hip (cluster_run_k16) for the
// error argument; the parity tests hold both kernels to the same bounds against the f64 oracle.
// Scales (powers of two, so scaling is exact): sx from the caller's bound on |X| (activations of LSTM layers: 1, with a residual
// sum 2), sw from the largest |W| of the call (k_gate_major) times the largest mask factor 1 / (1 - p); the largest scaled magnitude
// lies in [2^14, 2^15).  An input beyond the stated bound overflows f16 and shows as Inf / NaN in Z - never silently.
// List position p = q + 4c of a stage (quad q, slot c: the staging threads' order) is MFMA k-slot (half q >> 1, element 4 (q & 1) + c).
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
// x = hi + lo, both f16: hi = rn(x), lo = rn(x - hi), x an f32 VALUE.  The empty asm pins that value: without it hipcc contracts a
// product that feeds x into the subtraction (v_fma_mix: exact product - hi') AND takes that hi' by rounding the exact product to
// f16 in one step, while the hi it stores was rounded from the f32 product - two different roundings of x now and then, i.e. a lo
// that belongs to another hi: one operand in a few hundred with 11 instead of 22 bits (seen with the mask factor 2.5; a power-of-two
// factor makes the product exact and hides it).
__device__ __forceinline__ void mgr_split_f16(float x, _Float16& hi, _Float16& lo) {
  asm volatile("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
__global__ __launch_bounds__(256, 2) void k_gemm_nn_sparse16(const float* __restrict__ X, int ldx, const int* __restrict__ kidx,
                                                             const float* __restrict__ kval, const int* __restrict__ kcnt,
                                                             const float* __restrict__ Wp, const float* __restrict__ bp,
                                                             float* __restrict__ Z, int B, int T, int Fp, int F, int H,
                                                             const unsigned* __restrict__ wmax, float vmax, float sx, const unsigned* __restrict__ gate) {
  MGR_GATED(gate, false);
  constexpr int TM = SP_TM, TU = 64, SK = SP_SK, NT = 256, QT = 64, RPT = 2;
  __shared__ __attribute__((aligned(16))) _Float16 Ah[2][2][TM][8], Al[2][2][TM][8];   // [buffer][k half][row][8 k-slots]
  __shared__ __attribute__((aligned(16))) _Float16 Bh[2][2][TU][8], Bl[2][2][TU][8];
  __shared__ unsigned short Ls[SP_MAXF];
  __shared__ float Vs[SP_MAXF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int N = 4 * H;
  const int q = tid / QT, r = tid % QT;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ncol = (H + TU - 1) / TU, nrow = (T + TM - 1) / TM;
  float sw = 1.f;
  {
    const float m = __uint_as_float(*wmax) * vmax;
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
    ex = ex < -60 ? -60 : ex;
    sw = m < 3.0e38f ? ldexpf(1.f, 15 - ex) : 0.f;
  }
  const float inv = sw > 0.f ? 1.f / (sw * sx) : __uint_as_float(0x7FC00000u);
  struct Regs {
    float a[RPT * 4], w[4], v[4];
  };
  const int x = blockIdx.x & 7, jj = blockIdx.x >> 3;   // XCD-aware tile order: see k_gemm_nn_sparse
  const int rt = (jj / ncol) * 8 + x;
  if (rt >= nrow * B) return;
  const int u0 = (jj % ncol) * TU, r0 = (rt % nrow) * TM, b = rt / nrow;
  const float* Xb = X + (size_t)b * F * ldx + r0 + r;
  const int ucl = (u0 + r < H) ? u0 + r : H - 1;
  f32x16 acc[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][mt][e] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int nst1 = (kcnt[g * B + b] + SK - 1) / SK;
    // an EVEN number of stages (a padding stage has factor 0), and no conditional load in the time loop below: hipcc merges its
    // wait-count scoreboard over the branches of a loop body, and with `if (st + 2 < nst) fetch(...)` it assumes the fetch did not
    // happen - the s_waitcnt in front of the stash then counts down to vmcnt(0) and waits for the loads issued two stages AHEAD as
    // well, i.e. every stage paid a memory latency (the f32 kernel hid that under 1024 cycles of MFMA per stage): 1.99 -> 1.63 ms
    const int nst = (nst1 + 1) & ~1;
    {
      const int* list = kidx + ((size_t)g * B + b) * Fp;
      const float* lval = kval + ((size_t)g * B + b) * Fp;
      for (int i = tid; i < nst * SK; i += NT) {
        const bool in = i < nst1 * SK;
        Ls[i] = in ? (unsigned short)list[i] : (unsigned short)0;
        Vs[i] = in ? lval[i] : 0.f;
      }
    }
    __syncthreads();
    const float* Wg = Wp + (size_t)g * F * H + ucl;
    auto fetch = [&](Regs& R, int st) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int p = st * SK + q + 4 * c;
        const int f = Ls[p];
        R.v[c] = Vs[p];
        R.w[c] = Wg[(size_t)f * H];
#pragma unroll
        for (int i = 0; i < RPT; ++i) R.a[i * 4 + c] = Xb[(size_t)f * ldx + QT * i];
      }
    };
    auto split4 = [](const float (&xs)[4], f16x4_& hi, f16x4_& lo) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        _Float16 h, l;
        mgr_split_f16(xs[c], h, l);
        hi[c] = h;
        lo[c] = l;
      }
    };
    auto stash = [&](const Regs& R, int buf) {
      f16x4_ hi, lo;
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const float xs[4] = {R.a[i * 4] * sx, R.a[i * 4 + 1] * sx, R.a[i * 4 + 2] * sx, R.a[i * 4 + 3] * sx};
        split4(xs, hi, lo);
        *reinterpret_cast<f16x4_*>(&Ah[buf][q >> 1][r + QT * i][(q & 1) * 4]) = hi;
        *reinterpret_cast<f16x4_*>(&Al[buf][q >> 1][r + QT * i][(q & 1) * 4]) = lo;
      }
      const float ws[4] = {R.w[0] * R.v[0] * sw, R.w[1] * R.v[1] * sw, R.w[2] * R.v[2] * sw, R.w[3] * R.v[3] * sw};
      split4(ws, hi, lo);
      *reinterpret_cast<f16x4_*>(&Bh[buf][q >> 1][r][(q & 1) * 4]) = hi;
      *reinterpret_cast<f16x4_*>(&Bl[buf][q >> 1][r][(q & 1) * 4]) = lo;
    };
    auto mma = [&](int buf) {
      const int ra = wr * 64 + l31, ub = wc * 32 + l31;
      const f16x8_ bh = *reinterpret_cast<const f16x8_*>(&Bh[buf][lh][ub][0]);
      const f16x8_ bl = *reinterpret_cast<const f16x8_*>(&Bl[buf][lh][ub][0]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f16x8_ ah = *reinterpret_cast<const f16x8_*>(&Ah[buf][lh][ra + 32 * mt][0]);
        const f16x8_ al = *reinterpret_cast<const f16x8_*>(&Al[buf][lh][ra + 32 * mt][0]);
        acc[g][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[g][mt], 0, 0, 0);
        acc[g][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[g][mt], 0, 0, 0);
        acc[g][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[g][mt], 0, 0, 0);
      }
    };
    if (nst > 0) {
      Regs R0, R1;
      fetch(R0, 0);
      fetch(R1, 1);
      stash(R0, 0);
      __syncthreads();
      for (int st = 0; st < nst; st += 2) {   // (fetches beyond the last stage re-read it; what they stash is never multiplied)
        fetch(R0, st + 2 < nst ? st + 2 : nst - 1);
        mma(0);
        stash(R1, 1);
        __syncthreads();
        fetch(R1, st + 3 < nst ? st + 3 : nst - 1);
        mma(1);
        stash(R0, 0);
        __syncthreads();
      }
    }
  }
  const int unit = u0 + wc * 32 + l31;
  if (unit < H) {
    const float4 bias = *reinterpret_cast<const float4*>(bp + unit * 4);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (row < T)
          *reinterpret_cast<float4*>(Z + ((size_t)b * T + row) * N + unit * 4) =
              make_float4(fmaf(acc[0][mt][reg], inv, bias.x), fmaf(acc[1][mt][reg], inv, bias.y), fmaf(acc[2][mt][reg], inv, bias.z),
                          fmaf(acc[3][mt][reg], inv, bias.w));
      }
  }
}

// ------------------------------------------------------------------------------------------------ nn, split-f16, dense K
// At f16 matrix rates the per-gate K loops of k_gemm_nn_sparse16 no longer pay: its stages are bound by staging (index reads, gathered
// loads, the f32 -> (hi, lo) conversion of the A tile - once per GATE), not by the 6 MFMAs they feed.  This kernel walks ALL features
// once: the A tile of a stage (128 rows x 16 features, from the transposed copy) is fetched, split and written to LDS ONCE for the four
// gates, the B tiles carry the dropout mask as a factor (W_g[f, u] m_g[b, f] sw: a dropped feature is a zero row of that gate's B tile;
// the factor is wave-uniform, a scalar load), and a stage is 24 MFMAs per wave (4 gates x 2 row blocks x (hi hi + lo hi + hi lo)).
// Twice the MFMA work of the dropout-aware kernel at p = 0.5, a quarter of its A staging; without a mask (inference: mask4 = NULL) it
// is the plain dense projection on the f16 pipe.  Scales and error: as k_gemm_nn_sparse16.
__global__ __launch_bounds__(256, 2) void k_gemm_nn_dense16(const float* __restrict__ X, int ldx, const float* __restrict__ mask4,
                                                            const float* __restrict__ Wp, const float* __restrict__ bp,
                                                            float* __restrict__ Z, int B, int T, int F, int H,
                                                            const unsigned* __restrict__ wmax, float vmax, float sx, const unsigned* __restrict__ gate) {
  MGR_GATED(gate, false);
  constexpr int TM = SP_TM, TU = 64, NT = 256, QT = 64, RPT = 2;
  __shared__ __attribute__((aligned(16))) _Float16 Ah[2][2][TM][8], Al[2][2][TM][8];        // [buffer][k half][row][8 k-slots]
  __shared__ __attribute__((aligned(16))) _Float16 Bh[2][4][2][TU][8], Bl[2][4][2][TU][8];  // [buffer][gate][k half][unit][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int N = 4 * H;
  const int q = __builtin_amdgcn_readfirstlane(tid / QT), r = tid % QT;   // quad of list positions (= the wave), unit / row
  const int l31 = lane & 31, lh = lane >> 5;
  const int ncol = (H + TU - 1) / TU, nrow = (T + TM - 1) / TM;
  float sw = 1.f;
  {
    const float m = __uint_as_float(*wmax) * vmax;
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
    ex = ex < -60 ? -60 : ex;
    sw = m < 3.0e38f ? ldexpf(1.f, 15 - ex) : 0.f;
  }
  const float inv = sw > 0.f ? 1.f / (sw * sx) : __uint_as_float(0x7FC00000u);
  const int x = blockIdx.x & 7, jj = blockIdx.x >> 3;   // XCD-aware tile order: see k_gemm_nn_sparse
  const int rt = (jj / ncol) * 8 + x;
  if (rt >= nrow * B) return;
  const int u0 = (jj % ncol) * TU, r0 = (rt % nrow) * TM, b = rt / nrow;
  const float* Xb = X + (size_t)b * F * ldx + r0 + r;
  const int ucl = (u0 + r < H) ? u0 + r : H - 1;
  f32x16 acc[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][mt][e] = 0.f;
  struct Regs {
    float ra[RPT * 4], rw[16], rm[16];
  };
  auto fetch = [&](Regs& R, int st) {
    float (&ra)[RPT * 4] = R.ra;
    float (&rw)[16] = R.rw;
    float (&rm)[16] = R.rm;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int f = st * 16 + q + 4 * c;   // wave-uniform
      const bool fv = f < F;
      const int fc = fv ? f : F - 1;
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const float a = Xb[(size_t)fc * ldx + QT * i];
        ra[i * 4 + c] = fv ? a : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        rw[g * 4 + c] = Wp[((size_t)g * F + fc) * H + ucl];
        const float m = mask4 ? mask4[((size_t)g * B + b) * F + fc] : 1.f;
        rm[g * 4 + c] = fv ? m * sw : 0.f;
      }
    }
  };
  auto split4 = [](const float (&xs)[4], f16x4_& hi, f16x4_& lo) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      _Float16 h, l;
      mgr_split_f16(xs[c], h, l);
      hi[c] = h;
      lo[c] = l;
    }
  };
  auto stash = [&](const Regs& R, int buf) {
    const float (&ra)[RPT * 4] = R.ra;
    const float (&rw)[16] = R.rw;
    const float (&rm)[16] = R.rm;
    f16x4_ hi, lo;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const float xs[4] = {ra[i * 4] * sx, ra[i * 4 + 1] * sx, ra[i * 4 + 2] * sx, ra[i * 4 + 3] * sx};
      split4(xs, hi, lo);
      *reinterpret_cast<f16x4_*>(&Ah[buf][q >> 1][r + QT * i][(q & 1) * 4]) = hi;
      *reinterpret_cast<f16x4_*>(&Al[buf][q >> 1][r + QT * i][(q & 1) * 4]) = lo;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float ws[4] = {rw[g * 4] * rm[g * 4], rw[g * 4 + 1] * rm[g * 4 + 1], rw[g * 4 + 2] * rm[g * 4 + 2], rw[g * 4 + 3] * rm[g * 4 + 3]};
      split4(ws, hi, lo);
      *reinterpret_cast<f16x4_*>(&Bh[buf][g][q >> 1][r][(q & 1) * 4]) = hi;
      *reinterpret_cast<f16x4_*>(&Bl[buf][g][q >> 1][r][(q & 1) * 4]) = lo;
    }
  };
  auto mma = [&](int buf) {
    const int ra_ = wr * 64 + l31, ub = wc * 32 + l31;
    f16x8_ ah[2], al[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      ah[mt] = *reinterpret_cast<const f16x8_*>(&Ah[buf][lh][ra_ + 32 * mt][0]);
      al[mt] = *reinterpret_cast<const f16x8_*>(&Al[buf][lh][ra_ + 32 * mt][0]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f16x8_ bh = *reinterpret_cast<const f16x8_*>(&Bh[buf][g][lh][ub][0]);
      const f16x8_ bl = *reinterpret_cast<const f16x8_*>(&Bl[buf][g][lh][ub][0]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        acc[g][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[g][mt], 0, 0, 0);
        acc[g][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[g][mt], 0, 0, 0);
        acc[g][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[g][mt], 0, 0, 0);
      }
    }
  };
  // (two register sets, loads two stages ahead, no conditional load in the loop: k_gemm_nn_sparse16; a stage index beyond the
  // last stage has no valid feature: its tiles are zero)
  const int nst = (F + 15) / 16;
  Regs R0, R1;
  fetch(R0, 0);
  fetch(R1, 1);
  stash(R0, 0);
  __syncthreads();
  for (int st = 0; st < nst; st += 2) {
    fetch(R0, st + 2);
    mma(0);
    stash(R1, 1);
    __syncthreads();
    fetch(R1, st + 3);
    if (st + 1 < nst) mma(1);
    stash(R0, 0);
    __syncthreads();
  }
  const int unit = u0 + wc * 32 + l31;
  if (unit < H) {
    const float4 bias = *reinterpret_cast<const float4*>(bp + unit * 4);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (row < T)
          *reinterpret_cast<float4*>(Z + ((size_t)b * T + row) * N + unit * 4) =
              make_float4(fmaf(acc[0][mt][reg], inv, bias.x), fmaf(acc[1][mt][reg], inv, bias.y), fmaf(acc[2][mt][reg], inv, bias.z),
                          fmaf(acc[3][mt][reg], inv, bias.w));
      }
  }
}

// ------------------------------------------------------------------------------------------------ tn
// slab[z][f][n] = sum over samples b = z, z+SG, ...  of  mask(b,f,n) * sum_t A[b,t+shift,f] * dZ[b,t,n]
// grid: (ceil(N/128), ceil(F/128), SG)
__global__ __launch_bounds__(256, 2) void k_gemm_tn(const float* __restrict__ A, int lda, int shift,
                                                 const float* __restrict__ mask4, const float* __restrict__ dZ,
                                                 float* __restrict__ slab, int B, int T, int F, int N, int SG, int vecA) {
  __shared__ __attribute__((aligned(16))) float As2[NBUF][BK][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs2[NBUF][BK][LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * BN, f0 = blockIdx.y * BM, z = blockIdx.z;
  f32x16 tot[2][2];
  zero_acc(tot);
  RegTile ra, rb;
  for (int b = z; b < B; b += SG) {
    const float* Ab = A + (size_t)b * T * lda;
    const float* Zb = dZ + (size_t)b * T * N;
    f32x16 acc[2][2];
    zero_acc(acc);
    auto slow_stage = [&](int k0) {
      load_mc(ra, Ab, (size_t)lda, f0, k0, F, T, vecA != 0, tid, shift);
      load_mc(rb, Zb, (size_t)N, n0, k0, N, T, true, tid);
      __syncthreads();
      store_mc(As2[0], ra, tid);
      store_mc(Bs2[0], rb, tid);
      __syncthreads();
      mma_stage(As2[0], Bs2[0], acc, wr, wc, lane);
      __syncthreads();
    };
    // interior stages [kbeg, kend): all rows (and their time-shifted partners) in range -> branch-free pipelined loads
    int kbeg = 0, kend = vecA ? T / BK : 0;
    if (shift < 0 && kend > 0) kbeg = 1;                       // stage 0 touches row -1
    if (shift > 0 && kend > 0 && kend * BK - 1 + shift >= T) kend -= 1;  // last full stage touches row T
    if (kend <= kbeg) kbeg = kend = 0;
    for (int kt = 0; kt < kbeg; ++kt) slow_stage(kt * BK);
    if (kend > kbeg) {
      load_mc_fast(ra, Ab, (size_t)lda, f0, kbeg * BK, F, tid, shift);
      load_mc_fast(rb, Zb, (size_t)N, n0, kbeg * BK, N, tid);
      __syncthreads();
      store_mc(As2[0], ra, tid);
      store_mc(Bs2[0], rb, tid);
      __syncthreads();
      int buf = 0;
      for (int kt = kbeg; kt < kend; ++kt) {
        const bool more = kt + 1 < kend;
        if (more) {
          load_mc_fast(ra, Ab, (size_t)lda, f0, (kt + 1) * BK, F, tid, shift);
          load_mc_fast(rb, Zb, (size_t)N, n0, (kt + 1) * BK, N, tid);
        }
        mma_stage(As2[buf], Bs2[buf], acc, wr, wc, lane);
        if (NBUF == 1) __syncthreads();
        if (more) {
          store_mc(As2[buf ^ (NBUF - 1)], ra, tid);
          store_mc(Bs2[buf ^ (NBUF - 1)], rb, tid);
        }
        __syncthreads();
        buf ^= (NBUF - 1);
      }
    }
    for (int k0 = kend * BK; k0 < T; k0 += BK) slow_stage(k0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        int col = n0 + ACC_COL(wc, nt, lane);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          int f = f0 + ACC_ROW(wr, mt, reg, lane);
          float m = 1.f;
          if (mask4 && f < F && col < N) m = mask4[((size_t)(col & 3) * B + b) * F + f];
          tot[mt][nt][reg] += acc[mt][nt][reg] * m;
        }
      }
  }
  float* out = slab + (size_t)z * F * N;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      int col = n0 + ACC_COL(wc, nt, lane);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int f = f0 + ACC_ROW(wr, mt, reg, lane);
        if (f < F && col < N) out[(size_t)f * N + col] = tot[mt][nt][reg];
      }
    }
}

// ------------------------------------------------------------------------------------------------ tn, dropout-aware
// dW_g[f, u] = sum_b m_g[b, f] * sum_t X[b, t, f] dZ_g[b, t, u]: for a (gate, sample) only the KEPT features have rows.
// One workgroup takes 128 kept features of one (gate, sample) x 128 units of that gate and runs the dense K loop over
// t (X columns gathered by index, dZ columns of the gate taken with stride 4); the partial tile goes, scaled by the
// dropout factor, to P[(gate, sample)][list position][unit].  k_dw_gather then sums, per (feature, unit, gate), the samples
// that kept the feature, in sample order (deterministic).  Half the MFMA work of the dense kernel at p = 0.5.
// grid: 8 * ceil(B/8) * 4 * ceil(Fp/128) * ceil(H/128) workgroups, decoded below
// TR: both operands come from TRANSPOSED copies - X is XT[b][f][t] (row stride ldx = padded T, the copy the forward projection
// made) and dZ is dZT[b][4u+g][t] (row stride ldz, zero for t >= T like XT): the K dimension of this product is TIME, so a
// thread's 8 values per stage are two float4 of one row instead of 8 scattered 4-byte loads with strides of a whole frame
// (X: 64 useful bytes per 128-byte line; dZ: every fourth float).  Same stage layout in LDS, same MFMA order: bit-identical.
template <bool TR>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_sparse(const float* __restrict__ X, int ldx, const int* __restrict__ kidx,
                                                        const float* __restrict__ kval, const int* __restrict__ kcnt,
                                                        const float* __restrict__ dZ, int ldz, float* __restrict__ P, int B, int T,
                                                        int Fp, int F, int H, const unsigned* __restrict__ gate /* may be null: MGR_GATED */) {
  MGR_GATED(gate, true);
  __shared__ __attribute__((aligned(16))) float As2[NBUF][BK][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs2[NBUF][BK][LDS_LD];
  __shared__ float rowf[BM];
  static_assert(NBUF == 2 && BK == 16, "staging below assumes two LDS buffers of 16 k");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  // XCD-aware decode (consecutive workgroup ids go round-robin over the 8 XCDs): all workgroups of a sample - which share
  // its X rows and dZ columns - get ids congruent mod 8, i.e. one L2
  const int nft = (Fp + BM - 1) / BM, nut = (H + BN - 1) / BN, wps = 4 * nft * nut;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int b = (jj / wps) * 8 + xcd;
  if (b >= B) return;
  const int w = jj % wps, g = w / (nft * nut), gb = g * B + b;
  const int q0 = ((w / nut) % nft) * BM, u0 = (w % nut) * BN;
  const int cnt = kcnt[gb];
  if (q0 >= cnt) return;   // (uniform) no kept feature in this row tile
  const int N = 4 * H;
  const int m = tid & 127, kb = tid >> 7;   // staging: this thread's row (feature) / column (unit) and first k (then +2, ...; TR: 8 kb ..)
  const int q = q0 + m < Fp ? q0 + m : Fp - 1;
  const int un = u0 + m < H ? u0 + m : H - 1;
  const float* xcol = TR ? X + ((size_t)b * F + kidx[(size_t)gb * Fp + q]) * ldx + 8 * kb : X + (size_t)b * T * ldx + kidx[(size_t)gb * Fp + q];
  const float* zcol = TR ? dZ + ((size_t)b * N + 4 * un + g) * ldz + 8 * kb : dZ + (size_t)b * T * N + 4 * un + g;
  if (tid < BM) rowf[tid] = (q0 + tid < cnt) ? kval[(size_t)gb * Fp + q0 + tid] : 0.f;
  f32x16 acc[2][2];
  zero_acc(acc);
  float ra[8], rb[8];
  auto fetch = [&](int t0) {
    if constexpr (TR) {   // (t0 + 16 <= the padded row length; the pad is zero)
      const float4 a0 = *reinterpret_cast<const float4*>(xcol + t0), a1 = *reinterpret_cast<const float4*>(xcol + t0 + 4);
      const float4 z0 = *reinterpret_cast<const float4*>(zcol + t0), z1 = *reinterpret_cast<const float4*>(zcol + t0 + 4);
      ra[0] = a0.x; ra[1] = a0.y; ra[2] = a0.z; ra[3] = a0.w; ra[4] = a1.x; ra[5] = a1.y; ra[6] = a1.z; ra[7] = a1.w;
      rb[0] = z0.x; rb[1] = z0.y; rb[2] = z0.z; rb[3] = z0.w; rb[4] = z1.x; rb[5] = z1.y; rb[6] = z1.z; rb[7] = z1.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int t = t0 + kb + 2 * i;
        const bool ok = t < T;
        t = ok ? t : T - 1;
        const float a = xcol[(size_t)t * ldx], z = zcol[(size_t)t * N];
        ra[i] = ok ? a : 0.f;
        rb[i] = ok ? z : 0.f;
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = TR ? 8 * kb + i : kb + 2 * i;
      As2[buf][k][m] = ra[i];
      Bs2[buf][k][m] = rb[i];
    }
  };
  const int nst = (T + BK - 1) / BK;
  fetch(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int st = 0; st < nst; ++st) {
    const bool more = st + 1 < nst;
    if (more) fetch((st + 1) * BK);
    mma_stage(As2[buf], Bs2[buf], acc, wr, wc, lane);
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  float* out = P + (size_t)gb * Fp * H;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int u = u0 + ACC_COL(wc, nt, lane);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = ACC_ROW(wr, mt, reg, lane);
        if (q0 + r < cnt && u < H) out[(size_t)(q0 + r) * H + u] = acc[mt][nt][reg] * rowf[r];
      }
    }
}

// The transposed-operand form of k_gemm_tn_sparse on the f16 matrix pipe (round 4): both operands as split-f16 (hi, lo) pairs of
// their scaled values, three v_mfma_f32_32x32x16_f16 per 32 x 32 block and 16 time steps (hi hi + lo hi + hi lo, f32 accumulation)
// instead of eight v_mfma_f32_32x32x2_f32 - see k_gemm_nn_sparse16 / lstm_cluster.hip for the error argument.  The K dimension is
// time: a thread's 8 consecutive time steps of its row ARE one lane's operand of one MFMA, so a stage of 32 steps goes to LDS as two
// 16-byte writes per operand part.  Scales (powers of two): X by the caller's bound on |X|; dZT PER ROW (sample, gate column) by
// the row's largest magnitude (k_transpose_bt): a row's scale factors out of the sum over time exactly and is divided out of its
// output column, so the gradient's dynamic range across units, samples and gates costs nothing.
__global__ __launch_bounds__(256, 2) void k_gemm_tn_sparse16(const float* __restrict__ X, int ldx, const int* __restrict__ kidx,
                                                             const float* __restrict__ kval, const int* __restrict__ kcnt,
                                                             const float* __restrict__ dZ, int ldz, const unsigned* __restrict__ zmax,
                                                             float* __restrict__ P, int B, int T, int Fp, int F, int H, float sx, const unsigned* __restrict__ gate) {
  MGR_GATED(gate, false);
  constexpr int TK = 32;   // time steps per stage
  __shared__ __attribute__((aligned(16))) _Float16 Ah[2][2][2][BM][8], Al[2][2][2][BM][8];   // [buffer][k-step][k half][row][8]
  __shared__ __attribute__((aligned(16))) _Float16 Bh[2][2][2][BN][8], Bl[2][2][2][BN][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nft = (Fp + BM - 1) / BM, nut = (H + BN - 1) / BN, wps = 4 * nft * nut;   // (decode: see k_gemm_tn_sparse)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int b = (jj / wps) * 8 + xcd;
  if (b >= B) return;
  const int w = jj % wps, g = w / (nft * nut), gb = g * B + b;
  const int q0 = ((w / nut) % nft) * BM, u0 = (w % nut) * BN;
  const int cnt = kcnt[gb];
  if (q0 >= cnt) return;
  const int N = 4 * H;
  const int m = tid & 127, kb = tid >> 7;   // staging: this thread's row (feature / unit) and k-step of the stage (16 time steps)
  const int q = q0 + m < Fp ? q0 + m : Fp - 1;
  const int un = u0 + m < H ? u0 + m : H - 1;
  const float* xrow = X + ((size_t)b * F + kidx[(size_t)gb * Fp + q]) * ldx + 16 * kb;
  const float* zrow = dZ + ((size_t)b * N + 4 * un + g) * ldz + 16 * kb;
  auto zscale = [&](int unit) -> float {   // the power of two that puts the row's largest |dZ| in [2^14, 2^15)
    const float zm = __uint_as_float(zmax[(size_t)b * N + 4 * unit + g]);
    int ex = 0;
    if (zm > 0.f && zm < 3.0e38f) (void)frexpf(zm, &ex);
    ex = ex < -100 ? -100 : ex;
    return zm < 3.0e38f ? ldexpf(1.f, 15 - ex) : __uint_as_float(0x7FC00000u);   // (an Inf / NaN gradient stays visible)
  };
  const float sz = zscale(un);
  f32x16 acc[2][2];
  zero_acc(acc);
  struct Regs {
    float4 a[4], b[4];
  };
  auto fetch = [&](Regs& R, int t0) {   // (t0 + 32 <= the padded row length; the pad is zero)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      R.a[i] = *reinterpret_cast<const float4*>(xrow + t0 + 4 * i);
      R.b[i] = *reinterpret_cast<const float4*>(zrow + t0 + 4 * i);
    }
  };
  auto split8 = [](const float4& v0, const float4& v1, float s, f16x8_& hi, f16x8_& lo) {
    const float xs[8] = {v0.x * s, v0.y * s, v0.z * s, v0.w * s, v1.x * s, v1.y * s, v1.z * s, v1.w * s};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 h, l;
      mgr_split_f16(xs[e], h, l);
      hi[e] = h;
      lo[e] = l;
    }
  };
  auto stash = [&](const Regs& R, int buf) {
    f16x8_ hi, lo;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      split8(R.a[2 * hf], R.a[2 * hf + 1], sx, hi, lo);
      *reinterpret_cast<f16x8_*>(&Ah[buf][kb][hf][m][0]) = hi;
      *reinterpret_cast<f16x8_*>(&Al[buf][kb][hf][m][0]) = lo;
      split8(R.b[2 * hf], R.b[2 * hf + 1], sz, hi, lo);
      *reinterpret_cast<f16x8_*>(&Bh[buf][kb][hf][m][0]) = hi;
      *reinterpret_cast<f16x8_*>(&Bl[buf][kb][hf][m][0]) = lo;
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8_ ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f16x8_*>(&Ah[buf][ks][lh][wr * 64 + i * 32 + l31][0]);
        al[i] = *reinterpret_cast<const f16x8_*>(&Al[buf][ks][lh][wr * 64 + i * 32 + l31][0]);
        bh[i] = *reinterpret_cast<const f16x8_*>(&Bh[buf][ks][lh][wc * 64 + i * 32 + l31][0]);
        bl[i] = *reinterpret_cast<const f16x8_*>(&Bl[buf][ks][lh][wc * 64 + i * 32 + l31][0]);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
    }
  };
  // global loads run two stages ahead in two register sets, and NO load of the loop is conditional (a conditional fetch makes
  // hipcc count its s_waitcnt down to vmcnt(0): k_gemm_nn_sparse16); fetches beyond the last stage re-read it, what they stash is
  // never multiplied
  const int nst = (T + TK - 1) / TK;
  Regs R0, R1;
  fetch(R0, 0);
  fetch(R1, (nst > 1 ? 1 : 0) * TK);
  stash(R0, 0);
  __syncthreads();
  for (int st = 0; st < nst; st += 2) {
    fetch(R0, (st + 2 < nst ? st + 2 : nst - 1) * TK);
    mma(0);
    stash(R1, 1);
    __syncthreads();
    fetch(R1, (st + 3 < nst ? st + 3 : nst - 1) * TK);
    if (st + 1 < nst) mma(1);
    stash(R0, 0);
    __syncthreads();
  }
  float* out = P + (size_t)gb * Fp * H;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int u = u0 + ACC_COL(wc, nt, lane);
      // (the two reciprocals apart: zscale reaches 2^115 for a row of tiny gradients, and zscale * sx would overflow to Inf there -
      // cf = 0 flushed such a row's dW contribution to zero instead of rescaling it; both are powers of two, the products are exact)
      const float cz = 1.f / zscale(u < H ? u : H - 1), cx = 1.f / sx;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = ACC_ROW(wr, mt, reg, lane);
        if (q0 + r < cnt && u < H) out[(size_t)(q0 + r) * H + u] = acc[mt][nt][reg] * cx * cz * kval[(size_t)gb * Fp + q0 + r];
      }
    }
}

// dWp[f][4u+g] = sum over the samples that kept feature f for gate g, in sample order
__global__ __launch_bounds__(256) void k_dw_gather(const float* __restrict__ P, const int* __restrict__ kpos, float* __restrict__ dWp,
                                                   int B, int F, int Fp, int H) {
  const size_t n = (size_t)4 * F * H;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int u = (int)(i % H);
    const int f = (int)((i / H) % F);
    const int g = (int)(i / ((size_t)H * F));
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const int pos = kpos[((size_t)g * B + b) * F + f];
      if (pos >= 0) s += P[(((size_t)g * B + b) * Fp + pos) * H + u];
    }
    dWp[(size_t)f * 4 * H + 4 * u + g] = s;
  }
}

// ------------------------------------------------------------------------------------------------ nt
// dX[b,r,f] (+)= sum_j dZ[b,r,j] * Wp[f,j] * mask4[j&3,b,f];  grid: (ceil(F/128), ceil(T/128), B)
__global__ __launch_bounds__(256, 2) void k_gemm_nt(const float* __restrict__ dZ, const float* __restrict__ Wp,
                                                 const float* __restrict__ mask4, float* __restrict__ dX, int lddx,
                                                 int accumulate, int B, int T, int F, int N) {
  __shared__ __attribute__((aligned(16))) float As2[NBUF][BK][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs2[NBUF][BK][LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int f0 = blockIdx.x * BN, r0 = blockIdx.y * BM, b = blockIdx.z;
  const float* Zb = dZ + (size_t)b * T * N;
  f32x16 acc[2][2];
  zero_acc(acc);
  RegTile ra, rb, rm;
  // the mask factor of a B element depends only on (gate, f): constant over the K loop -> load once
  if (mask4) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      int idx4 = tid + i * 256;
      int f = f0 + idx4 / (BK / 4);
      f = f < F ? f : F - 1;
      // k0 and k4 are multiples of 4: the float4 covers gates 0..3 of one unit
#pragma unroll
      for (int g = 0; g < 4; ++g) rm.v[i * 4 + g] = mask4[((size_t)g * B + b) * F + f];
    }
  }
  auto stash = [&](int buf) {
    if (mask4) {
#pragma unroll
      for (int e = 0; e < 4 * LPT; ++e) rb.v[e] *= rm.v[e];
    }
    store_kc(As2[buf], ra, tid);
    store_kc(Bs2[buf], rb, tid);
  };
  const int nfast = N / BK;
  if (nfast > 0) {
    load_kc_fast(ra, Zb, (size_t)N, r0, 0, T, tid);
    load_kc_fast(rb, Wp, (size_t)N, f0, 0, F, tid);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt < nfast; ++kt) {
      const bool more = kt + 1 < nfast;
      if (more) {
        load_kc_fast(ra, Zb, (size_t)N, r0, (kt + 1) * BK, T, tid);
        load_kc_fast(rb, Wp, (size_t)N, f0, (kt + 1) * BK, F, tid);
      }
      mma_stage(As2[buf], Bs2[buf], acc, wr, wc, lane);
      if (NBUF == 1) __syncthreads();
      if (more) stash(buf ^ (NBUF - 1));
      __syncthreads();
      buf ^= (NBUF - 1);
    }
  }
  for (int k0 = nfast * BK; k0 < N; k0 += BK) {
    load_kc(ra, Zb, (size_t)N, r0, k0, T, N, true, tid);
    load_kc(rb, Wp, (size_t)N, f0, k0, F, N, true, tid);
    stash(0);
    __syncthreads();
    mma_stage(As2[0], Bs2[0], acc, wr, wc, lane);
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      int f = f0 + ACC_COL(wc, nt, lane);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int r = r0 + ACC_ROW(wr, mt, reg, lane);
        if (r < T && f < F) {
          float* o = dX + ((size_t)b * T + r) * lddx + f;
          *o = accumulate ? (*o + acc[mt][nt][reg]) : acc[mt][nt][reg];
        }
      }
    }
}

// slab reduce: out[i] = sum_k slab[k][i]
__global__ void k_reduce(const float* __restrict__ slab, float* __restrict__ out, size_t n, int nslab) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += slab[(size_t)k * n + i];
    out[i] = s;
  }
}

// column sums of dZ [rows, N] -> slab[wg][N]; each WG takes a contiguous row range; thread = (four columns, row lane):
// float4 loads, two row lanes summed through LDS (N = 4H is a multiple of 4)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dZ, float* __restrict__ slab, size_t rows, int N,
                                                int rows_per_wg) {
  __shared__ float4 part[128];
  const size_t rbeg = (size_t)blockIdx.x * rows_per_wg;
  const size_t rend = rbeg + rows_per_wg < rows ? rbeg + rows_per_wg : rows;
  const int cl = threadIdx.x & 127, ry = threadIdx.x >> 7, N4 = N / 4;
  for (int c0 = 0; c0 < N4; c0 += 128) {
    const int c4 = c0 + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < N4)
      for (size_t r = rbeg + ry; r < rend; r += 2) {
        const float4 v = *reinterpret_cast<const float4*>(dZ + r * N + 4 * c4);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
      }
    if (ry == 1) part[cl] = s;
    __syncthreads();
    if (ry == 0 && c4 < N4) {
      const float4 o = part[cl];
      *reinterpret_cast<float4*>(slab + (size_t)blockIdx.x * N + 4 * c4) = make_float4(s.x + o.x, s.y + o.y, s.z + o.z, s.w + o.w);
    }
    __syncthreads();
  }
}

// out[i] = sum_k slab[k][i] for MANY slabs of a SHORT vector (the bias gradient: hundreds of row-block partial sums of 4H
// numbers): 8 slab lanes per element, summed through LDS in a fixed order
__global__ __launch_bounds__(256) void k_reduce_tall(const float* __restrict__ slab, float* __restrict__ out, int n, int nslab) {
  __shared__ float part[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), kl = threadIdx.x >> 5;
  float s = 0.f;
  if (e < n)
    for (int k = kl; k < nslab; k += 8) s += slab[(size_t)k * n + e];
  part[kl][threadIdx.x & 31] = s;
  __syncthreads();
  if (kl == 0 && e < n) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x & 31];
    out[e] = t;
  }
}

static int tn_groups(int B, int F, int N) {
  int tiles = ((F + BM - 1) / BM) * ((N + BN - 1) / BN);
  int sg = (1024 + tiles - 1) / tiles;
  if (sg > B) sg = B;
  if (sg < 1) sg = 1;
  return sg;
}
static int colsum_wgs(size_t rows) {
  // (32 rows per workgroup, up to 512 workgroups: with 256 rows each a short batch - the audio configuration, 1600 rows - was summed
  //  by 7 workgroups of serial loads, 0.14 ms per call and 12 % of its step)
  size_t w = (rows + 31) / 32;
  if (w > 512) w = 512;
  if (w < 1) w = 1;
  return (int)w;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int mgr_lstm_input_proj(mgr_ctx* c, const float* X, int ldx, const float* mask4, const float* Wp, const float* bp,
                        float* Z, int B, int T, int F, int H) {
  MGR_REQUIRE(c && X && Wp && bp && Z, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0 && ldx >= F, "bad shape");
  MGR_REQUIRE(aligned16(Wp) && aligned16(Z), "Wp/Z must be 16-byte aligned");
  int N = 4 * H;
  int vecA = (ldx % 4 == 0) && (F % 4 == 0) && aligned16(X);
  dim3 grid((N + BN - 1) / BN, (T + BM - 1) / BM, B);
  mgr_prof_begin(c, MGR_K_GEMM_NN);
  hipLaunchKernelGGL(k_gemm_nn, grid, dim3(256), 0, mgr_stream(c), X, ldx, mask4, Wp, bp, Z, B, T, F, N, vecA);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_NN);
  return 0;
}

int mgr_lstm_input_proj_pair(mgr_ctx* c, const float* X, int ldx, const float* mask4_fwd, const float* Wp_fwd,
                             const float* bp_fwd, float* Z_fwd, const float* mask4_rev, const float* Wp_rev,
                             const float* bp_rev, float* Z_rev, int B, int T, int F, int H) {
  MGR_REQUIRE(c && X && Wp_fwd && bp_fwd && Z_fwd && Wp_rev && bp_rev && Z_rev, "null argument");
  MGR_REQUIRE((mask4_fwd == nullptr) == (mask4_rev == nullptr), "both directions masked or neither");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0 && ldx >= F, "bad shape");
  const int Nd = 4 * H;
  const bool fast = (ldx % 4 == 0) && (F % BK == 0) && aligned16(X) && aligned16(Wp_fwd) && aligned16(Wp_rev);
  // one launch pays when it saves column tiles; otherwise (or for shapes the fused kernel does not take) two plain ones
  if (!fast || (2 * Nd + BN - 1) / BN >= 2 * ((Nd + BN - 1) / BN)) {
    int r = mgr_lstm_input_proj(c, X, ldx, mask4_fwd, Wp_fwd, bp_fwd, Z_fwd, B, T, F, H);
    if (r) return r;
    return mgr_lstm_input_proj(c, X, ldx, mask4_rev, Wp_rev, bp_rev, Z_rev, B, T, F, H);
  }
  dim3 grid((2 * Nd + BN - 1) / BN, (T + BM - 1) / BM, B);
  mgr_prof_begin(c, MGR_K_GEMM_NN);
  hipLaunchKernelGGL(k_gemm_nn2, grid, dim3(256), 0, mgr_stream(c), X, ldx, mask4_fwd, mask4_rev, Wp_fwd, Wp_rev, bp_fwd, bp_rev,
                     Z_fwd, Z_rev, B, T, F, Nd);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_NN);
  return 0;
}

size_t mgr_lstm_input_proj_dropout_ws_bytes(int B, int F, int H) {
  const size_t Fp = (size_t)(F + SP_SK - 1) / SP_SK * SP_SK;
  return 2 * mgr_align_up((size_t)4 * B * Fp * sizeof(int), 256) + mgr_align_up((size_t)4 * B * sizeof(int), 256) + 256 +
         mgr_align_up((size_t)F * 4 * H * sizeof(float), 256);   // (+256: the max |W| word of the split-f16 kernel)
}

static bool sparse_proj_shape(const mgr_ctx* c, float drop_rate, int F) {
  // the per-gate K loops pay when enough features are dropped; at small F (depth-1 layers, F = 39 / 20) the GEMM is bound
  // by the Z stores and the float4 epilogue of this kernel is what helps (0.39 / 0.18 ms against 0.47 / 0.24)
  return drop_rate >= 0.3f && F >= 16 && F <= SP_MAXF && c->tune[9] == 0;
}

int mgr_lstm_input_proj_dropout_wants_transposed(mgr_ctx* c, float drop_rate, int F) {
  // the transposed copy pays where the A operand dominates the staging traffic: wide inputs (depth-2 / fusion layers)
  return (c && sparse_proj_shape(c, drop_rate, F) && F >= 128) ? 1 : 0;
}

static int input_proj_dropout_impl(mgr_ctx* c, const float* X, int ldx, bool transposed, const float* mask4, float drop_rate,
                                   const float* Wp, const float* bp, float* Z, int B, int T, int F, int H, void* ws, size_t ws_bytes,
                                   float x_absmax = 0.f) {
  const int Fp = (F + SP_SK - 1) / SP_SK * SP_SK;
  const size_t lbytes = mgr_align_up((size_t)4 * B * Fp * sizeof(int), 256);
  char* w = reinterpret_cast<char*>(ws);
  int* kidx = reinterpret_cast<int*>(w);
  float* kval = reinterpret_cast<float*>(w + lbytes);
  int* kcnt = reinterpret_cast<int*>(w + 2 * lbytes);
  unsigned* wmax = reinterpret_cast<unsigned*>(w + 2 * lbytes + mgr_align_up((size_t)4 * B * sizeof(int), 256));
  float* Wg = reinterpret_cast<float*>(w + 2 * lbytes + mgr_align_up((size_t)4 * B * sizeof(int), 256) + 256);
  hipStream_t s = mgr_stream(c);
  // x_absmax > 0: a bound the CALLER states - checked on the device (k_absmax_gate), f32 kernel if violated; < 0: |x_absmax| is a bound
  // the PRODUCER of XT guarantees (mgr.h): no check
  const bool trusted = x_absmax < 0.f;
  const float xb = fabsf(x_absmax);
  // split-f16 kernel (tune key 15 = 1: never): transposed input with a bound on |X|, a drop rate that bounds the mask factor
  const bool f16 = transposed && xb > 0.f && xb < 1.0e30f && drop_rate < 0.99f && c->tune[15] == 0;
  // dense K loop with the mask as a factor (k_gemm_nn_dense16): where there is no mask (inference); tune key 10 = 2: always.  With a
  // mask the per-gate K loops over the kept features are faster (audio depth 2: 1.99 against 2.28 ms)
  const bool dense = f16 && (!mask4 || c->tune[10] == 2);
  MGR_REQUIRE(mask4 || transposed, "a projection without a dropout mask is only handled from the transposed copy");
  unsigned* gate = (f16 && !trusted) ? wmax + 1 : nullptr;
  const bool lists = !dense || gate;   // the kept-feature lists: what every kernel but the dense one walks (no mask: all features)
  mgr_prof_begin(c, MGR_K_GEMM_NN);
  if (lists && mask4)
    hipLaunchKernelGGL(k_mask_compact, dim3(4 * B), dim3(64), 0, s, mask4, F, Fp, kidx, kval, kcnt, (int*)nullptr, wmax);
  else if (lists)
    hipLaunchKernelGGL(k_mask_all, dim3(4 * B), dim3(64), 0, s, F, Fp, kidx, kval, kcnt, wmax);
  else
    MGR_HIP(hipMemsetAsync(wmax, 0, sizeof(unsigned), s));
  {
    const size_t n = (size_t)F * H;
    const int wgs = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_gate_major, dim3(wgs), dim3(256), 0, s, Wp, Wg, F, H, wmax);
  }
  // 128-unit tiles (tune key 11 = 2) are faster alone (audio L2 2.77 against 3.03 ms) but slower in the training step
  // (39.7 against 38.5 ms/step): a 512-thread workgroup needs two free wave slots on all four SIMDs of a CU at once and
  // gets in the way of the BPTT scan and the small kernels of the other stream
  const bool wide = c->tune[11] == 2 && !transposed;
  const int tu = wide ? 128 : 64;
  const int ntiles = ((H + tu - 1) / tu) * ((((T + SP_TM - 1) / SP_TM) * B + 7) / 8) * 8;   // (row tiles padded to the 8 XCDs)
  if (f16) {
    int ex;
    (void)frexpf(xb, &ex);                       // xb = m 2^ex, m in [0.5, 1): |X| sx < 2^15
    const float sx = ldexpf(1.f, 15 - ex);
    if (gate) {   // what |X| the f16 range holds at this scale: beyond it the split would carry Inf (ldx is the padded row length)
      MGR_HIP(hipMemsetAsync(gate, 0, sizeof(unsigned), s));
      const size_t n4 = (size_t)B * F * ldx / 4;
      hipLaunchKernelGGL(k_absmax_gate, dim3((int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096)), dim3(256), 0, s, X, n4, 65000.f / sx, gate);
    }
    if (dense)
      hipLaunchKernelGGL(k_gemm_nn_dense16, dim3(ntiles), dim3(256), 0, s, X, ldx, mask4, Wg, bp, Z, B, T, F, H, wmax,
                         mask4 ? 1.f / (1.f - drop_rate) : 1.f, sx, gate);
    else
      hipLaunchKernelGGL(k_gemm_nn_sparse16, dim3(ntiles), dim3(256), 0, s, X, ldx, kidx, kval, kcnt, Wg, bp, Z, B, T, Fp, F, H, wmax,
                         1.f / (1.f - drop_rate), sx, gate);
    if (gate)   // the f32 MFMA kernel over the same lists: runs only if the gate was raised
      hipLaunchKernelGGL((k_gemm_nn_sparse<2, true>), dim3(ntiles), dim3(256), 0, s, X, ldx, kidx, kval, kcnt, Wg, bp, Z, B, T, Fp, F, H, gate);
  } else if (transposed)
    hipLaunchKernelGGL((k_gemm_nn_sparse<2, true>), dim3(ntiles), dim3(256), 0, s, X, ldx, kidx, kval, kcnt, Wg, bp, Z, B, T, Fp, F, H, (const unsigned*)nullptr);
  else if (wide)
    hipLaunchKernelGGL((k_gemm_nn_sparse<4, false>), dim3(ntiles), dim3(512), 0, s, X, ldx, kidx, kval, kcnt, Wg, bp, Z, B, T, Fp, F, H, (const unsigned*)nullptr);
  else
    hipLaunchKernelGGL((k_gemm_nn_sparse<2, false>), dim3(ntiles), dim3(256), 0, s, X, ldx, kidx, kval, kcnt, Wg, bp, Z, B, T, Fp, F, H, (const unsigned*)nullptr);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_NN);
  return 0;
}

int mgr_lstm_input_proj_dropout(mgr_ctx* c, const float* X, int ldx, const float* mask4, float drop_rate, const float* Wp,
                                const float* bp, float* Z, int B, int T, int F, int H, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && X && Wp && bp && Z, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0 && ldx >= F, "bad shape");
  const bool sparse = mask4 && sparse_proj_shape(c, drop_rate, F) && (size_t)T * ldx < (1u << 31) && aligned16(bp) && aligned16(Z) &&
                      aligned16(Wp);
  if (!sparse) return mgr_lstm_input_proj(c, X, ldx, mask4, Wp, bp, Z, B, T, F, H);
  MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_input_proj_dropout_ws_bytes(B, F, H), "workspace too small");
  mgr_planes_forget_ws(c, ws);   // (this call writes its own lists / weight copies into the workspace: cached split planes in it are gone)
  return input_proj_dropout_impl(c, X, ldx, false, mask4, drop_rate, Wp, bp, Z, B, T, F, H, ws, ws_bytes);
}

int mgr_lstm_input_proj_dropout_t(mgr_ctx* c, const float* XT, int ldt, const float* mask4, float drop_rate, const float* Wp,
                                  const float* bp, float* Z, int B, int T, int F, int H, void* ws, size_t ws_bytes, float x_absmax) {
  MGR_REQUIRE(c && XT && Wp && bp && Z, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0, "bad shape");
  MGR_REQUIRE(ldt % 4 == 0 && ldt >= (T + SP_TM - 1) / SP_TM * SP_TM, "the transposed copy must be padded to whole row tiles of %d (ldt %d, T %d)", SP_TM, ldt, T);
  MGR_REQUIRE(mask4 ? sparse_proj_shape(c, drop_rate, F) : (F >= 16 && F <= SP_MAXF),
              "shape / drop rate not handled by the dropout-aware kernel (ask mgr_lstm_input_proj_dropout_wants_transposed)");
  if (!mask4) drop_rate = 0.f;
  MGR_REQUIRE(aligned16(XT) && aligned16(bp) && aligned16(Z) && aligned16(Wp), "XT / bp / Z / Wp must be 16-byte aligned");
  MGR_REQUIRE((size_t)F * ldt < (1u << 31), "sample block too large");
  MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_input_proj_dropout_ws_bytes(B, F, H), "workspace too small");
  mgr_planes_forget_ws(c, ws);   // (this call writes its own lists / weight copies into the workspace: cached split planes in it are gone)
  return input_proj_dropout_impl(c, XT, ldt, true, mask4, drop_rate, Wp, bp, Z, B, T, F, H, ws, ws_bytes, x_absmax);
}

// XT[b][f][0..ldt) = X[b][0..T)[f], zero for t >= T (ldt: T padded to whole row tiles of the dropout-aware projection)
namespace {
__global__ __launch_bounds__(256) void k_transpose_bt(const float* __restrict__ X, int ldx, float* __restrict__ XT, int ldt, int T, int F,
                                                      long long xtb /* batch stride of XT; 0: F * ldt */, int fill /* columns written: ldt or less */,
                                                      unsigned* __restrict__ rowmax /* [B][F] largest |x| of a row of XT as float bits, by atomic max (zeroed by the caller); may be null */) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  const float* Xb = X + (size_t)b * T * ldx;
  float* XTb = XT + (size_t)b * (xtb ? (size_t)xtb : (size_t)F * ldt);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = t0 + ty + 4 * i, f = f0 + tx;
    tile[ty + 4 * i][tx] = (t < T && f < F) ? Xb[(size_t)t * ldx + f] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = f0 + ty + 4 * i, t = t0 + tx;
    const float v = tile[tx][ty + 4 * i];
    if (f < F && t < fill) XTb[(size_t)f * ldt + t] = v;
    if (rowmax) {   // (a wave holds 64 time steps of ONE row)
      float m = fabsf(v);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if (tx == 0 && f < F) atomicMax(rowmax + (size_t)b * F + f, __float_as_uint(m));
    }
  }
}
}  // namespace

int mgr_transpose_bt(mgr_ctx* c, const float* X, int ldx, float* XT, int ldt, int B, int T, int F) {
  MGR_REQUIRE(c && X && XT, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && ldx >= F && ldt >= T, "bad shape");
  mgr_prof_begin(c, MGR_K_MISC);
  hipLaunchKernelGGL(k_transpose_bt, dim3((ldt + 63) / 64, (F + 63) / 64, B), dim3(256), 0, mgr_stream(c), X, ldx, XT, ldt, T, F, 0LL, ldt, (unsigned*)nullptr);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_MISC);
  return 0;
}

size_t mgr_lstm_param_grads_ws_bytes(int B, int T, int F, int H) {
  int N = 4 * H;
  size_t a = mgr_align_up((size_t)tn_groups(B, F, N) * F * N * sizeof(float), 256);
  size_t b = mgr_align_up((size_t)tn_groups(B, H, N) * H * N * sizeof(float), 256);
  size_t d = mgr_align_up((size_t)colsum_wgs((size_t)B * T) * N * sizeof(float), 256);
  return a + b + d;
}

static int param_grads_impl(mgr_ctx* c, const float* X, int ldx, const float* mask4, const float* Hs, int ldh, const float* dZ,
                            float* dWp, float* dUp, float* dbp, int B, int T, int F, int H, int reverse, void* ws, bool with_dW,
                            const float* dbsum = nullptr) {
  int N = 4 * H;
  int sgW = tn_groups(B, F, N), sgU = tn_groups(B, H, N);
  char* w = reinterpret_cast<char*>(ws);
  float* slabW = reinterpret_cast<float*>(w);
  w += mgr_align_up((size_t)sgW * F * N * sizeof(float), 256);
  float* slabU = reinterpret_cast<float*>(w);
  w += mgr_align_up((size_t)sgU * H * N * sizeof(float), 256);
  float* slabB = reinterpret_cast<float*>(w);
  hipStream_t s = mgr_stream(c);
  if (with_dW) {
    int vecA = (ldx % 4 == 0) && (F % 4 == 0) && aligned16(X);
    dim3 grid((N + BN - 1) / BN, (F + BM - 1) / BM, sgW);
    hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, s, X, ldx, 0, mask4, dZ, slabW, B, T, F, N, sgW, vecA);
    size_t n = (size_t)F * N;
    hipLaunchKernelGGL(k_reduce, dim3((int)((n + 255) / 256)), dim3(256), 0, s, slabW, dWp, n, sgW);
  }
  if (dUp) {   // (null: the caller forms dU itself - gemm_split.hip, from the split transposed copy of h_prev)
    // h_prev: forward direction uses h[t-1], reverse direction uses h[t+1]
    int vecA = (ldh % 4 == 0) && (H % 4 == 0) && aligned16(Hs);
    dim3 grid((N + BN - 1) / BN, (H + BM - 1) / BM, sgU);
    hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, s, Hs, ldh, reverse ? 1 : -1, (const float*)nullptr, dZ, slabU, B, T, H, N, sgU, vecA);
    size_t n = (size_t)H * N;
    hipLaunchKernelGGL(k_reduce, dim3((int)((n + 255) / 256)), dim3(256), 0, s, slabU, dUp, n, sgU);
  }
  if (dbsum) {   // the BPTT's per-sample sums over time: db = their sum over the samples, in sample order
    hipLaunchKernelGGL(k_reduce_tall, dim3((N + 31) / 32), dim3(256), 0, s, dbsum, dbp, N, B);
  } else {
    size_t rows = (size_t)B * T;
    int nwg = colsum_wgs(rows);
    int rpw = (int)((rows + nwg - 1) / nwg);
    nwg = (int)((rows + rpw - 1) / rpw);
    hipLaunchKernelGGL(k_colsum, dim3(nwg), dim3(256), 0, s, dZ, slabB, rows, N, rpw);
    hipLaunchKernelGGL(k_reduce_tall, dim3((N + 31) / 32), dim3(256), 0, s, slabB, dbp, N, nwg);
  }
  return 0;
}

int mgr_lstm_param_grads(mgr_ctx* c, const float* X, int ldx, const float* mask4, const float* Hs, int ldh,
                         const float* dZ, float* dWp, float* dUp, float* dbp, int B, int T, int F, int H, int reverse,
                         void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && X && Hs && dZ && dWp && dUp && dbp, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0 && ldx >= F && ldh >= H, "bad shape");
  MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_param_grads_ws_bytes(B, T, F, H), "workspace too small");
  MGR_REQUIRE(aligned16(dZ), "dZ must be 16-byte aligned");
  mgr_prof_begin(c, MGR_K_GEMM_TN);
  param_grads_impl(c, X, ldx, mask4, Hs, ldh, dZ, dWp, dUp, dbp, B, T, F, H, reverse, ws, true);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_TN);
  return 0;
}

static size_t pg_dropout_extra(int B, int F, int H) {
  const size_t Fp = (size_t)(F + SP_SK - 1) / SP_SK * SP_SK;
  return 2 * mgr_align_up((size_t)4 * B * Fp * sizeof(int), 256) + mgr_align_up((size_t)4 * B * sizeof(int), 256) +
         mgr_align_up((size_t)4 * B * F * sizeof(int), 256) + mgr_align_up((size_t)4 * B * Fp * H * sizeof(float), 256);
}

size_t mgr_lstm_param_grads_dropout_ws_bytes(int B, int T, int F, int H) {
  return mgr_lstm_param_grads_ws_bytes(B, T, F, H) + pg_dropout_extra(B, F, H);
}

// dU / db (and the dense dW when the shape is not sparse) + the dropout-aware dW; XT / ldt != 0: operands of the dW product
// from transposed copies (XT given by the caller, dZT made here)
static int param_grads_dropout_impl(mgr_ctx* c, const float* X, int ldx, const float* XT, int ldt, const float* mask4, float drop_rate,
                                    const float* Hs, int ldh, const float* dZ, float* dWp, float* dUp, float* dbp, int B, int T, int F,
                                    int H, int reverse, void* ws, bool sparse, float x_absmax = 0.f) {
  mgr_prof_begin(c, MGR_K_GEMM_TN);
  // dU / db first: they are short, and in the training step the long dW kernel then ends this direction's work (the step
  // runs these under an encoder scan; what is left over after the scan is exposed)
  param_grads_impl(c, X, ldx, mask4, Hs, ldh, dZ, dWp, dUp, dbp, B, T, F, H, reverse, ws, !sparse);
  if (sparse) {
    const int Fp = (F + SP_SK - 1) / SP_SK * SP_SK;
    const size_t lbytes = mgr_align_up((size_t)4 * B * Fp * sizeof(int), 256);
    char* w = reinterpret_cast<char*>(ws) + mgr_lstm_param_grads_ws_bytes(B, T, F, H);
    int* kidx = reinterpret_cast<int*>(w);
    float* kval = reinterpret_cast<float*>(w + lbytes);
    int* kcnt = reinterpret_cast<int*>(w + 2 * lbytes);
    w += 2 * lbytes + mgr_align_up((size_t)4 * B * sizeof(int), 256);
    int* kpos = reinterpret_cast<int*>(w);
    w += mgr_align_up((size_t)4 * B * F * sizeof(int), 256);
    float* P = reinterpret_cast<float*>(w);
    w += mgr_align_up((size_t)4 * B * Fp * H * sizeof(float), 256);
    hipStream_t s = mgr_stream(c);
    hipLaunchKernelGGL(k_mask_compact, dim3(4 * B), dim3(64), 0, s, mask4, F, Fp, kidx, kval, kcnt, kpos, (unsigned*)nullptr);
    const int grid = 8 * ((B + 7) / 8) * 4 * ((Fp + BM - 1) / BM) * ((H + BN - 1) / BN);
    if (XT) {
      float* dZT = reinterpret_cast<float*>(w);   // [B][4H][ldt]
      w += mgr_align_up((size_t)B * 4 * H * ldt * sizeof(float), 256);
      // split-f16 kernel (tune key 15 = 1: never): a bound on |X| (stated: checked on the device, f32 kernel if violated; negative:
      // guaranteed by the producer of XT), whole stages of 32 time steps in the padded rows
      const bool trusted = x_absmax < 0.f;
      const float xb = fabsf(x_absmax);
      const bool f16 = xb > 0.f && xb < 1.0e30f && c->tune[15] == 0 && ldt >= (T + 31) / 32 * 32;
      unsigned* zmax = reinterpret_cast<unsigned*>(w);   // [B][4H] largest |dZ| of a (sample, gate column), + the gate word
      unsigned* gate = (f16 && !trusted) ? zmax + (size_t)B * 4 * H : nullptr;
      if (f16) MGR_HIP(hipMemsetAsync(zmax, 0, ((size_t)B * 4 * H + 1) * sizeof(unsigned), s));
      hipLaunchKernelGGL(k_transpose_bt, dim3((ldt + 63) / 64, (4 * H + 63) / 64, B), dim3(256), 0, s, dZ, 4 * H, dZT, ldt, T, 4 * H, 0LL, ldt,
                         f16 ? zmax : (unsigned*)nullptr);
      if (f16) {
        int ex;
        (void)frexpf(xb, &ex);
        const float sx = ldexpf(1.f, 15 - ex);
        if (gate) {
          const size_t n4 = (size_t)B * F * ldt / 4;
          hipLaunchKernelGGL(k_absmax_gate, dim3((int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096)), dim3(256), 0, s, XT, n4, 65000.f / sx, gate);
        }
        hipLaunchKernelGGL(k_gemm_tn_sparse16, dim3(grid), dim3(256), 0, s, XT, ldt, kidx, kval, kcnt, dZT, ldt, zmax, P, B, T, Fp, F, H, sx, gate);
        if (gate)
          hipLaunchKernelGGL((k_gemm_tn_sparse<true>), dim3(grid), dim3(256), 0, s, XT, ldt, kidx, kval, kcnt, dZT, ldt, P, B, T, Fp, F, H, gate);
      } else {
        hipLaunchKernelGGL((k_gemm_tn_sparse<true>), dim3(grid), dim3(256), 0, s, XT, ldt, kidx, kval, kcnt, dZT, ldt, P, B, T, Fp, F, H, (const unsigned*)nullptr);
      }
    } else {
      hipLaunchKernelGGL((k_gemm_tn_sparse<false>), dim3(grid), dim3(256), 0, s, X, ldx, kidx, kval, kcnt, dZ, 0, P, B, T, Fp, F, H, (const unsigned*)nullptr);
    }
    const size_t n = (size_t)4 * F * H;
    hipLaunchKernelGGL(k_dw_gather, dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, P, kpos, dWp, B, F, Fp, H);
  }
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_TN);
  return 0;
}

static bool sparse_dw_shape(mgr_ctx* c, const float* mask4, float drop_rate, int F) {
  return mask4 && drop_rate >= 0.3f && F >= 128 && c->tune[9] == 0;
}

int mgr_lstm_param_grads_dropout(mgr_ctx* c, const float* X, int ldx, const float* mask4, float drop_rate, const float* Hs,
                                 int ldh, const float* dZ, float* dWp, float* dUp, float* dbp, int B, int T, int F, int H,
                                 int reverse, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && X && Hs && dZ && dWp && dUp && dbp, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0 && ldx >= F && ldh >= H, "bad shape");
  MGR_REQUIRE(aligned16(dZ), "dZ must be 16-byte aligned");
  const bool sparse = sparse_dw_shape(c, mask4, drop_rate, F);
  MGR_REQUIRE(ws && ws_bytes >= (sparse ? mgr_lstm_param_grads_dropout_ws_bytes(B, T, F, H) : mgr_lstm_param_grads_ws_bytes(B, T, F, H)),
              "workspace too small");
  return param_grads_dropout_impl(c, X, ldx, nullptr, 0, mask4, drop_rate, Hs, ldh, dZ, dWp, dUp, dbp, B, T, F, H, reverse, ws, sparse);
}

int mgr_lstm_param_grads_dropout_wants_transposed(mgr_ctx* c, float drop_rate, int F) {
  return (c && drop_rate >= 0.3f && F >= 128 && c->tune[9] == 0) ? 1 : 0;
}

size_t mgr_lstm_param_grads_dropout_t_ws_bytes(int B, int T, int F, int H, int ldt) {
  return mgr_lstm_param_grads_dropout_ws_bytes(B, T, F, H) + mgr_align_up((size_t)B * 4 * H * ldt * sizeof(float), 256) +
         mgr_align_up(((size_t)B * 4 * H + 1) * sizeof(unsigned), 256);   // (dZT, the row maxima of dZT + the bound-violation word)
}

int mgr_lstm_param_grads_dropout_t(mgr_ctx* c, const float* XT, int ldt, const float* mask4, float drop_rate, const float* Hs, int ldh,
                                   const float* dZ, float* dWp, float* dUp, float* dbp, int B, int T, int F, int H, int reverse,
                                   void* ws, size_t ws_bytes, float x_absmax) {
  MGR_REQUIRE(c && XT && mask4 && Hs && dZ && dWp && dUp && dbp, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0 && ldh >= H, "bad shape");
  MGR_REQUIRE(ldt % 4 == 0 && ldt >= (T + BK - 1) / BK * BK, "the transposed copy must be padded to whole stages of %d time steps (ldt %d, T %d)", BK, ldt, T);
  MGR_REQUIRE(aligned16(dZ) && aligned16(XT), "dZ / XT must be 16-byte aligned");
  MGR_REQUIRE(sparse_dw_shape(c, mask4, drop_rate, F), "shape / drop rate not handled by the dropout-aware kernel (ask mgr_lstm_param_grads_dropout_wants_transposed)");
  MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_param_grads_dropout_t_ws_bytes(B, T, F, H, ldt), "workspace too small");
  return param_grads_dropout_impl(c, nullptr, 0, XT, ldt, mask4, drop_rate, Hs, ldh, dZ, dWp, dUp, dbp, B, T, F, H, reverse, ws, true, x_absmax);
}

int mgr_lstm_input_grad(mgr_ctx* c, const float* dZ, const float* Wp, const float* mask4, float* dX, int lddx,
                        int accumulate, int B, int T, int F, int H) {
  MGR_REQUIRE(c && dZ && Wp && dX, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && F > 0 && H > 0 && lddx >= F, "bad shape");
  MGR_REQUIRE(aligned16(dZ) && aligned16(Wp), "dZ/Wp must be 16-byte aligned");
  int N = 4 * H;
  dim3 grid((F + BN - 1) / BN, (T + BM - 1) / BM, B);
  mgr_prof_begin(c, MGR_K_GEMM_NT);
  hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, mgr_stream(c), dZ, Wp, mask4, dX, lddx, accumulate, B, T, F, N);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_GEMM_NT);
  return 0;
}

}  // extern "C"

// dU / db of one direction (the part of the parameter gradients that does not read the layer input): for gemm_split.hip.  ws: the
// first mgr_lstm_param_grads_ws_bytes(B, T, F, H) bytes of the caller's workspace
int mgr_param_grads_du_db(mgr_ctx* c, const float* Hs, int ldh, const float* dZ, float* dUp, float* dbp, int B, int T, int F, int H, int reverse,
                          void* ws, const float* dbsum) {
  return param_grads_impl(c, nullptr, 0, nullptr, Hs, ldh, dZ, nullptr, dUp, dbp, B, T, F, H, reverse, ws, false, dbsum);
}

int mgr_transpose_bt_strided(mgr_ctx* c, const float* X, int ldx, float* XT, int ldt, long long xtb, int ldt_fill, int B, int T, int F) {
  hipLaunchKernelGGL(k_transpose_bt, dim3((ldt_fill + 63) / 64, (F + 63) / 64, B), dim3(256), 0, mgr_stream(c), X, ldx, XT, ldt, T, F, xtb, ldt_fill, (unsigned*)nullptr);
  MGR_LAUNCH_CHECK();
  return 0;
}

