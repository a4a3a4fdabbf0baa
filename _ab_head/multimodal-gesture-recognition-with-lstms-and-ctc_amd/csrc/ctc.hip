// K6: CTC loss + gradient w.r.t. the Dense logits: three kernels - emissions (a thread per frame), the alpha / beta recursions (two
// waves per sample), gradient (a thread per frame).
//
// Restates K.ctc_batch_cost -> tf.nn.ctc_loss as called by ctc_lambda_func
// (reference multimodal_fusion/losses.py:4-15): y = softmax(log(P[:,skip:]+eps)), blank = C-1,
// log-space alpha/beta DP over l' = [blank,l1,blank,...,lL,blank].
//
// Recursions: one wave runs alpha forward in time, one runs beta backward in time, concurrently.  The
// extended label sequence lives across the lanes of the wave as (blank,label) PAIRS: lane*PPL+j holds
// states 2p (blank) and 2p+1 (label p), so the only cross-lane traffic per time step is one
// shift-by-one of the neighbouring pair's label state; the three-way log-sum-exp is max-shifted fp32.
// Emissions log2 y(t,.) are precomputed (k_ctc_emissions) and software-prefetched a chunk of time
// steps ahead of the recursion, so the serial chain per step is shift -> lse -> add.
// Numerics: alpha and beta are RENORMALISED every 16 time steps (the wave-wide max is subtracted and summed
// into an fp64 scalar), so the stored log-values stay O(10) instead of O(-5000) at T=1900, where an fp32 ulp is
// 5e-4 and would put percent-level noise on the gradient.  The loss adds the fp64 offset back; the gradient
// needs no offsets at all because sum_u alpha(t,u)beta(t,u) = p(l|x) at every t, so each frame's occupancies
// are normalised by their own sum.
// The gradient (k_ctc_grad) combines alpha+beta into per-class occupancies through an LDS row per thread and
// chains through softmax(log(P+eps)) and the network's own softmax to dLogits.
// Vector-memory instructions of the serial chain (round 4; a wave that is alone with its chain pays 60-130 cycles of issue for
// each): the emissions are stored CLASS-MAJOR, forward in time for alpha and reversed for beta, so that a lane fetches eight
// steps of its label's (and the blank's) emissions with two 16-byte loads instead of 16 four-byte ones, and alpha / beta rows
// are stored two time steps at a time (one 16-byte store per lane and pair of steps): ~1 instead of ~3 per step.
#include "common.h"

namespace {

constexpr float kNegInf = -__builtin_huge_valf();

// The recursions run in BASE-2 log units (round 6): log2 y emissions, log2 alpha / beta, so that a log-sum-exp is v_exp_f32 / v_log_f32
// on their own - no log2(e) / ln 2 multiplications, and none of logf's denormal-range and last-ulp fix-ups either: the argument of the
// logarithm lies in [1, 3] (the largest term contributes exactly 1), where the raw instruction's 1 ulp is 1e-7 absolute on values that
// are added to numbers of size O(10).  That was 30 of the 95 instructions of a step of the serial chain (12 for each logf).
// All-(-inf) inputs (log 0): the maximum is floored at a huge finite negative number, x - floor stays -inf, 2^-inf = 0, log2 0 = -inf.
constexpr float kLseFloor = -3.0e38f;
constexpr double kLn2 = 0.693147180559945309417232121458;
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_raw(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(fmaxf(a, b), kLseFloor);
  return m + log2_raw(exp2_raw(a - m) + exp2_raw(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(fmaxf(a, b), c), kLseFloor);
  return m + log2_raw(exp2_raw(a - m) + exp2_raw(b - m) + exp2_raw(c - m));
}

// Cross-lane traffic of the recursions through DPP (one v_mov_b32_dpp, a few cycles) instead of __shfl_* (a ds_bpermute round trip
// through the LDS crossbar, ~100 cycles, on the serial chain of every time step): wave_shr:1 / wave_shl:1 move a value to the
// next / previous lane of the whole wave; lanes without a source receive `fill`.
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// wave-wide maximum: DPP within the rows of 16 lanes, then the four row results through v_readlane (uniform)
__device__ __forceinline__ float wave_max_f32(float m) {
  m = fmaxf(m, dpp_f32<0x111>(m, m));   // row_shr:1 (lanes without a source keep their own value)
  m = fmaxf(m, dpp_f32<0x112>(m, m));
  m = fmaxf(m, dpp_f32<0x114>(m, m));
  m = fmaxf(m, dpp_f32<0x118>(m, m));   // lane 15 of a row holds the row's maximum
  const int i = __float_as_int(m);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 15)), __int_as_float(__builtin_amdgcn_readlane(i, 31))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 47)), __int_as_float(__builtin_amdgcn_readlane(i, 63))));
}

template <int PPL>
struct Chunk {
  static constexpr int CH = PPL <= 2 ? 8 : 4;   // time steps per prefetched chunk (a multiple of 4: float4 loads along time)
  float eb[CH];
  float el[CH][PPL];
};
// row length of the class-major emission copies: T' + the over-read of two chunks + the 3-float offset that puts t = 1 on a
// 16-byte boundary (alpha's chunks start at t = 1, 9, 17, ...)
__host__ __device__ inline size_t ctc_ts(int To) { return ((size_t)To + 24 + 3) / 4 * 4; }
// alpha / beta rows, two time steps per block: element (t, state 2p + e) at  (t >> 1) * 2 S2 + 4 p + 2 (t & 1) + e
__device__ __forceinline__ size_t ab_off(int t, int S2) { return (size_t)(t >> 1) * (2 * S2) + 2 * (t & 1); }

// One sample's three phases as functions (the recurrence kernel runs ONE or TWO samples per workgroup).
// Per-sample state: which sample, its clipped lengths, its slices of the workspace, its labels in the workgroup's LDS.
struct CtcSample {
  int b, Tp, L;
  float *LYTb, *LYRb, *ALb, *BEb;
  int* s_lab;
};

// the sample's labels into the workgroup's LDS (clipped into the class range)
__device__ __forceinline__ void ctc_labels(const CtcSample& cs_, int tid, int nthreads, const int32_t* __restrict__ labels, int C, int Lmax) {
  for (int i = tid; i < Lmax; i += nthreads) {
    int v = (i < cs_.L) ? labels[(size_t)cs_.b * Lmax + i] : -1;
    v = v < 0 ? 0 : (v >= C ? C - 1 : v);
    cs_.s_lab[i] = v;
  }
}
// ---- phase 0: emissions log2 y(t,c) = log2(P+eps) - log2(sum_c (P+eps)), class-major, forward and reversed in time.  One frame per
// thread: workgroup `blk` of the sample's ceil(To / 256) (workgroup 0 also writes the rows' zero tails)
__device__ __forceinline__ void ctc_phase0(const CtcSample& cs_, int tid, int blk, const float* __restrict__ P, int T, int C, int skip, float eps,
                                           size_t TS) {
  const int b = cs_.b, Tp = cs_.Tp;
  float *LYTb = cs_.LYTb, *LYRb = cs_.LYRb;
  const int t = blk * 256 + tid;
  if (t < Tp) {
    const float* row = P + ((size_t)b * T + skip + t) * C;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += row[c] + eps;
    float ls = log2f(s);
    for (int c = 0; c < C; ++c) {
      const float v = log2f(row[c] + eps) - ls;   // (base-2 units, as everything the recursions touch)
      LYTb[(size_t)c * TS + t + 3] = v;
      LYRb[(size_t)c * TS + (Tp - 1 - t)] = v;
    }
  }
  if (blk != 0) return;
  // the two-chunks-ahead prefetch of the recursions reads up to 2 CH + 2 = 18 floats behind the last emission of a row (values it
  // never uses): they are zeros, not whatever the workspace held (forward rows: [Tp + 3, Tp + 24), reversed rows: [Tp, Tp + 21);
  // TS >= To + 24 holds both)
  for (int i = tid; i < C * 21; i += 256) {
    const int c = i / 21, k = i % 21;
    LYTb[(size_t)c * TS + Tp + 3 + k] = 0.f;
    LYRb[(size_t)c * TS + Tp + k] = 0.f;
  }
}

// ---- phase 1 (one wave per role): role 0 runs alpha forward in time, role 1 beta backward
template <int PPL>
__device__ __forceinline__ void ctc_phase1(const CtcSample& cs_, int role, int lane, int blank, int Lmax, int S2, size_t TS,
                                           float* __restrict__ loss) {
  constexpr int CH = Chunk<PPL>::CH;
  const int b = cs_.b, Tp = cs_.Tp, L = cs_.L;
  float *LYTb = cs_.LYTb, *LYRb = cs_.LYRb, *ALb = cs_.ALb, *BEb = cs_.BEb;
  int* s_lab = cs_.s_lab;
  if (role < 2 && Tp > 0) {
    int lab[PPL];
    bool vl[PPL], vb[PPL], cs[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      int p = lane * PPL + j;
      vl[j] = p < L;
      vb[j] = p <= L;
      lab[j] = vl[j] ? s_lab[p] : blank;
    }
    {
      int prev_last = __shfl_up(lab[PPL - 1], 1);
      bool prev_vl = __shfl_up((int)vl[PPL - 1], 1) != 0;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        int p = lane * PPL + j;
        int pl = (j > 0) ? lab[j - 1] : prev_last;
        bool pv = (j > 0) ? vl[j - 1] : (lane > 0 && prev_vl);
        cs[j] = vl[j] && p >= 1 && pv && lab[j] != blank && lab[j] != pl;
      }
    }
    if (role == 0) {
      float ab[PPL], al[PPL];
      float hb[PPL], hl[PPL];   // the even step of the pair in flight (rows are stored two steps at a time)
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        int p = lane * PPL + j;
        ab[j] = (p == 0) ? LYTb[(size_t)blank * TS + 3] : kNegInf;
        al[j] = (p == 0 && vl[j]) ? LYTb[(size_t)lab[j] * TS + 3] : kNegInf;
        hb[j] = ab[j];
        hl[j] = al[j];
        if (Tp == 1 && p <= Lmax) *reinterpret_cast<float2*>(ALb + 4 * p) = make_float2(ab[j], al[j]);
      }
      Chunk<PPL> cur, nxt;
      auto load = [&](Chunk<PPL>& ch, int t0) {   // steps t0 .. t0 + CH - 1 (t0 = 1 mod CH: 16-byte aligned; over-read stays in the row)
        const float* rb = LYTb + (size_t)blank * TS + t0 + 3;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(rb + 4 * q);
          ch.eb[4 * q] = v.x; ch.eb[4 * q + 1] = v.y; ch.eb[4 * q + 2] = v.z; ch.eb[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          const float* rl = LYTb + (size_t)lab[j] * TS + t0 + 3;
#pragma unroll
          for (int q = 0; q < CH / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(rl + 4 * q);
            ch.el[4 * q][j] = v.x; ch.el[4 * q + 1][j] = v.y; ch.el[4 * q + 2][j] = v.z; ch.el[4 * q + 3][j] = v.w;
          }
        }
      };
      load(cur, 1);
      double coff = 0.0;
      int since = 0;
      for (int t0 = 1; t0 < Tp; t0 += CH) {
        load(nxt, t0 + CH);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          int t = t0 + k;
          if (t < Tp) {
            const float carry = dpp_f32<DPP_WAVE_SHR1>(al[PPL - 1], kNegInf);   // (lane 0: log 0)
            float nb[PPL], nl[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
              float up = (j > 0) ? al[j - 1] : carry;
              nb[j] = vb[j] ? cur.eb[k] + lse2(ab[j], up) : kNegInf;
              nl[j] = vl[j] ? cur.el[k][j] + lse3(al[j], ab[j], cs[j] ? up : kNegInf) : kNegInf;
            }
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
              ab[j] = nb[j];
              al[j] = nl[j];
              int p = lane * PPL + j;
              if (t & 1) {          // the pair (t - 1, t) is complete: one 16-byte store
                if (p <= Lmax) *reinterpret_cast<float4*>(ALb + ab_off(t - 1, S2) + 4 * p) = make_float4(hb[j], hl[j], ab[j], al[j]);
              } else {
                hb[j] = ab[j];
                hl[j] = al[j];
                if (t == Tp - 1 && p <= Lmax) *reinterpret_cast<float2*>(ALb + ab_off(t, S2) + 4 * p) = make_float2(ab[j], al[j]);
              }
            }
          }
        }
        cur = nxt;
        since += CH;
        if (since >= 16) {  // renormalise: keep the running log-values O(10)
          since = 0;
          float m = kNegInf;
#pragma unroll
          for (int j = 0; j < PPL; ++j) m = fmaxf(m, fmaxf(ab[j], al[j]));
          m = wave_max_f32(m);
          if (m != kNegInf) {
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
              ab[j] -= m;
              al[j] -= m;
            }
            coff += (double)m;
          }
        }
      }
      // log p(l|x) = lse(alpha(2L, Tp-1), alpha(2L-1, Tp-1)) + accumulated offset
      float fb = kNegInf, fl = kNegInf;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        int p = lane * PPL + j;
        if (p == L) fb = ab[j];
        if (p == L - 1) fl = al[j];
      }
      // reduce across lanes (exactly one lane holds each)
      for (int o = 32; o > 0; o >>= 1) {
        fb = fmaxf(fb, __shfl_xor(fb, o));
        fl = fmaxf(fl, __shfl_xor(fl, o));
      }
      float lfin = lse2(fb, fl);
      if (lane == 0) {
        // (+inf: no alignment fits - what k_ctc_grad reads as "no gradient")
        loss[b] = (lfin == kNegInf) ? __builtin_huge_valf() : (float)(-((double)lfin + coff) * kLn2);
      }
    } else {
      float bb[PPL], bl[PPL];
      bool csn[PPL];  // can_skip of pair p+1
      {
        int nfirst = __shfl_down((int)cs[0], 1);
#pragma unroll
        for (int j = 0; j < PPL; ++j) csn[j] = (j < PPL - 1) ? cs[j + 1] : (lane < 63 && nfirst != 0);
      }
      float hb[PPL], hl[PPL];   // the odd step of the pair in flight (beta walks down: t + 1 comes before t)
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        int p = lane * PPL + j;
        bb[j] = (p == L) ? 0.f : kNegInf;
        bl[j] = (p == L - 1) ? 0.f : kNegInf;
        hb[j] = bb[j];
        hl[j] = bl[j];
        // t = Tp - 1: an even t is a pair's first half and nothing pairs with it from above - stored on its own
        if (((Tp - 1) & 1) == 0 && p <= Lmax) *reinterpret_cast<float2*>(BEb + ab_off(Tp - 1, S2) + 4 * p) = make_float2(bb[j], bl[j]);
      }
      Chunk<PPL> cur, nxt;
      // step index n = 0.. walks t = Tp-2-n; uses emissions at t+1 = Tp-1-n = reversed index n
      auto load = [&](Chunk<PPL>& ch, int n0) {   // reversed steps n0 .. n0 + CH - 1 (n0 = 0 mod CH: aligned)
        const float* rb = LYRb + (size_t)blank * TS + n0;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(rb + 4 * q);
          ch.eb[4 * q] = v.x; ch.eb[4 * q + 1] = v.y; ch.eb[4 * q + 2] = v.z; ch.eb[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          const float* rl = LYRb + (size_t)lab[j] * TS + n0;
#pragma unroll
          for (int q = 0; q < CH / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(rl + 4 * q);
            ch.el[4 * q][j] = v.x; ch.el[4 * q + 1][j] = v.y; ch.el[4 * q + 2][j] = v.z; ch.el[4 * q + 3][j] = v.w;
          }
        }
      };
      load(cur, 0);
      const int nsteps = Tp - 1;
      int since = 0;
      for (int n0 = 0; n0 < nsteps; n0 += CH) {
        load(nxt, n0 + CH);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          int n = n0 + k;
          if (n < nsteps) {
            int t = Tp - 2 - n;
            float xb[PPL], xl[PPL];  // beta(.,t+1) + emission(.,t+1)
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
              xb[j] = vb[j] ? bb[j] + cur.eb[k] : kNegInf;
              xl[j] = vl[j] ? bl[j] + cur.el[k][j] : kNegInf;
            }
            const float nxb = dpp_f32<DPP_WAVE_SHL1>(xb[0], kNegInf);   // (lane 63: log 0)
            const float nxl = dpp_f32<DPP_WAVE_SHL1>(xl[0], kNegInf);
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
              float b1 = (j < PPL - 1) ? xb[j + 1] : nxb;  // blank of pair p+1 (state u+1 for the label state)
              float l1 = (j < PPL - 1) ? xl[j + 1] : nxl;  // label of pair p+1 (state u+2)
              bb[j] = vb[j] ? lse2(xb[j], xl[j]) : kNegInf;
              bl[j] = vl[j] ? lse3(xl[j], b1, csn[j] ? l1 : kNegInf) : kNegInf;
              int p = lane * PPL + j;
              if (t & 1) {          // first half of the pair (t - 1, t) to arrive: keep it (or store it alone at t = ... never: t >= 0 even ends)
                hb[j] = bb[j];
                hl[j] = bl[j];
              } else {              // the pair (t, t + 1) is complete (t + 1 was held, or is the initial row when Tp - 1 is odd)
                if (p <= Lmax) *reinterpret_cast<float4*>(BEb + ab_off(t, S2) + 4 * p) = make_float4(bb[j], bl[j], hb[j], hl[j]);
              }
            }
          }
        }
        cur = nxt;
        since += CH;
        if (since >= 16) {
          since = 0;
          float m = kNegInf;
#pragma unroll
          for (int j = 0; j < PPL; ++j) m = fmaxf(m, fmaxf(bb[j], bl[j]));
          m = wave_max_f32(m);
          if (m != kNegInf) {
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
              bb[j] -= m;
              bl[j] -= m;
            }
          }
        }
      }
    }
  }
}

// ---- phase 2 (one frame per thread): gradient w.r.t. the Dense logits
__device__ __forceinline__ void ctc_phase2(const CtcSample& cs_, int tid, int blk, float* smem, const float* __restrict__ P, int T, int C, int skip,
                                           int blank, float eps, float gscale, int S2, bool dead, float* __restrict__ dLogits) {
  const int b = cs_.b, Tp = cs_.Tp, L = cs_.L;
  float *ALb = cs_.ALb, *BEb = cs_.BEb;
  int* s_lab = cs_.s_lab;
  float* occ = smem + (size_t)tid * (C + 1);
  const int f = blk * 256 + tid;
  if (f >= T) return;
  float* out = dLogits + ((size_t)b * T + f) * C;
  const int tt = f - skip;
  if (tt < 0 || tt >= Tp || dead) {   // (dead: p(l|x) = 0 - no alignment fits - or no frames at all: the loss is +inf, the gradient zero)
    for (int c = 0; c < C; ++c) out[c] = 0.f;
    return;
  }
  for (int c = 0; c < C; ++c) occ[c] = 0.f;
  const float* ar = ALb + ab_off(tt, S2);
  const float* br = BEb + ab_off(tt, S2);
  float vmax = kNegInf;
  for (int p = 0; p <= L; ++p) {
    float2 a = *reinterpret_cast<const float2*>(ar + 4 * p);
    float2 be = *reinterpret_cast<const float2*>(br + 4 * p);
    vmax = fmaxf(vmax, a.x + be.x);
    if (p < L) vmax = fmaxf(vmax, a.y + be.y);
  }
  float den = 0.f;
  for (int p = 0; p <= L; ++p) {
    float2 a = *reinterpret_cast<const float2*>(ar + 4 * p);
    float2 be = *reinterpret_cast<const float2*>(br + 4 * p);
    float wb = exp2_raw(a.x + be.x - vmax);
    occ[blank] += wb;
    den += wb;
    if (p < L) {
      float wl = exp2_raw(a.y + be.y - vmax);
      occ[s_lab[p]] += wl;
      den += wl;
    }
  }
  const float iden = 1.f / den;  // sum_u alpha*beta = p(l|x) for every t: normalise by the frame's own sum
  const float* row = P + ((size_t)b * T + f) * C;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += row[c] + eps;
  float inv = 1.f / s;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) {
    float u = row[c] + eps;
    float gp = (u * inv - occ[c] * iden) / u;
    occ[c] = gp;
    dot += row[c] * gp;
  }
  for (int c = 0; c < C; ++c) out[c] = row[c] * (occ[c] - dot) * gscale;
}

// One sample's slices of the workspace and its clipped lengths.  s_lab: the sample's label row in the workgroup's LDS (or null).
__device__ __forceinline__ CtcSample ctc_sample(int b, int B, const int32_t* __restrict__ input_len, const int32_t* __restrict__ label_len, int To,
                                                int C, int Lmax, float* LY, float* AL, float* BE, int* s_lab) {
  CtcSample s_;
  s_.b = b < B ? b : -1;
  const int bc = b < B ? b : B - 1;
  const int S2 = 2 * (Lmax + 1);
  const size_t TS = ctc_ts(To);
  int Tp = input_len[bc];
  s_.Tp = Tp < 0 ? 0 : (Tp > To ? To : Tp);
  int L = label_len[bc];
  s_.L = L < 0 ? 0 : (L > Lmax ? Lmax : L);
  s_.LYTb = LY + (size_t)bc * 2 * C * TS;           // [C][TS]: y(t, c) at c * TS + t + 3
  s_.LYRb = s_.LYTb + (size_t)C * TS;               // [C][TS]: y(Tp - 1 - r, c) at c * TS + r
  s_.ALb = AL + (size_t)bc * (To + 1) * S2;
  s_.BEb = BE + (size_t)bc * (To + 1) * S2;
  s_.s_lab = s_lab;
  return s_;
}

// Three kernels (round 6; one kernel with three phases before): the emissions and the gradient are one independent piece of work per
// FRAME - 121,600 of them at config F's shape - and ran on the 256 threads of the workgroup that owns the sample's two chains, one sample
// after the other: 0.09 + 0.20 ms of the kernel's 0.56, for microseconds of work once every frame has a thread of its own.
//
// k_ctc_emissions: grid (ceil(To / 256), B) - phase 0.
__global__ __launch_bounds__(256) void k_ctc_emissions(const float* __restrict__ P, const int32_t* __restrict__ input_len,
                                                       const int32_t* __restrict__ label_len, int B, int T, int C, int Lmax, int skip, float eps,
                                                       float* __restrict__ LY, float* __restrict__ AL, float* __restrict__ BE) {
  const CtcSample s_ = ctc_sample((int)blockIdx.y, B, input_len, label_len, T - skip, C, Lmax, LY, AL, BE, nullptr);
  ctc_phase0(s_, threadIdx.x, (int)blockIdx.x, P, T, C, skip, eps, ctc_ts(T - skip));
}

// k_ctc_chains - phase 1.  SPW = samples per workgroup.  1: wave 0 = alpha, wave 1 = beta of the workgroup's sample (128 threads).  2: waves
// 0, 1 run the chains of sample 2 j, waves 2, 3 those of sample 2 j + 1 - one chain per SIMD.  In the training step the kernel runs on
// the 48 CUs the fused encoder scans leave: 64 one-sample workgroups there put two on 16 CUs, whose alpha (beta) waves then SHARE a SIMD,
// and every chain of the launch waits for those (0.97 ms in the step for 0.55 alone); 32 two-sample workgroups have a CU each.
template <int PPL, int SPW>
__global__ __launch_bounds__(128 * SPW) void k_ctc_chains(const int32_t* __restrict__ labels, const int32_t* __restrict__ input_len,
                                                          const int32_t* __restrict__ label_len, int B, int T, int C, int Lmax, int skip, int blank,
                                                          float* __restrict__ loss, float* __restrict__ LY, float* __restrict__ AL,
                                                          float* __restrict__ BE) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // SPW label rows
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int To = T - skip;
  const int S2 = 2 * (Lmax + 1);
  const size_t TS = ctc_ts(To);
  auto sample = [&](int si) {
    return ctc_sample((int)blockIdx.x * SPW + si, B, input_len, label_len, To, C, Lmax, LY, AL, BE,
                      reinterpret_cast<int*>(smem) + si * (Lmax + 1));
  };
  {
    const CtcSample s_ = sample(wave >> 1);
    if (s_.b >= 0) ctc_labels(s_, tid & 127, 128, labels, C, Lmax);
  }
  __syncthreads();
  {
    const CtcSample s_ = sample(wave >> 1);
    if (s_.b >= 0) ctc_phase1<PPL>(s_, wave & 1, lane, blank, Lmax, S2, TS, loss);
    if (s_.b >= 0 && s_.Tp == 0 && (tid & 127) == 0) loss[s_.b] = __builtin_huge_valf();
  }
}

// k_ctc_grad: grid (ceil(T / 256), B) - phase 2.
__global__ __launch_bounds__(256) void k_ctc_grad(const float* __restrict__ P, const int32_t* __restrict__ labels,
                                                  const int32_t* __restrict__ input_len, const int32_t* __restrict__ label_len, int B, int T, int C,
                                                  int Lmax, int skip, int blank, float eps, float gscale, const float* __restrict__ loss,
                                                  float* __restrict__ dLogits, float* __restrict__ LY, float* __restrict__ AL,
                                                  float* __restrict__ BE) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [256][C+1] occupancy rows, then the label row
  const CtcSample s_ = ctc_sample((int)blockIdx.y, B, input_len, label_len, T - skip, C, Lmax, LY, AL, BE,
                                  reinterpret_cast<int*>(smem + 256 * (C + 1)));
  ctc_labels(s_, threadIdx.x, 256, labels, C, Lmax);
  __syncthreads();
  const bool dead = loss[s_.b] == __builtin_huge_valf();
  ctc_phase2(s_, threadIdx.x, (int)blockIdx.x, smem, P, T, C, skip, blank, eps, gscale, 2 * (Lmax + 1), dead, dLogits);
}

}  // namespace

extern "C" {

size_t mgr_ctc_ws_bytes(int B, int T, int C, int Lmax) {
  size_t To = (size_t)(T > 0 ? T : 1);
  size_t ly = mgr_align_up((size_t)B * 2 * C * ctc_ts((int)To) * sizeof(float), 256);   // class-major emissions, forward + reversed
  size_t ab = mgr_align_up((size_t)B * (To + 1) * 2 * (Lmax + 1) * sizeof(float), 256);
  return ly + 2 * ab;
}

int mgr_ctc_loss_grad(mgr_ctx* c, const float* P, const int32_t* labels, const int32_t* input_len,
                      const int32_t* label_len, int B, int T, int C, int Lmax, int skip, int blank, float eps,
                      float gscale, float* loss, float* dLogits, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && P && labels && input_len && label_len && loss, "null argument");
  MGR_REQUIRE(B > 0 && T > skip && skip >= 0 && C > 1 && Lmax > 0, "bad shape B=%d T=%d C=%d Lmax=%d skip=%d", B, T, C, Lmax, skip);
  MGR_REQUIRE(blank >= 0 && blank < C, "blank %d out of range", blank);
  MGR_REQUIRE(Lmax + 1 <= 256, "Lmax %d too large (max 255)", Lmax);
  MGR_REQUIRE(ws && ws_bytes >= mgr_ctc_ws_bytes(B, T, C, Lmax), "workspace too small");
  size_t To = (size_t)T;  // sized with T (>= T-skip) to keep the query simple
  size_t ly = mgr_align_up((size_t)B * 2 * C * ctc_ts((int)To) * sizeof(float), 256);
  size_t ab = mgr_align_up((size_t)B * (To + 1) * 2 * (Lmax + 1) * sizeof(float), 256);
  float* LY = reinterpret_cast<float*>(ws);
  float* AL = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ly);
  float* BE = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ly + ab);
  // two samples per workgroup (one chain per SIMD) from B = 2 on; tune key 18 = 1: one sample per workgroup (rounds 1 - 5)
  const int spw = (B >= 2 && c->tune[18] == 0) ? 2 : 1;
  size_t lds_grad = (size_t)256 * (C + 1) * sizeof(float) + (size_t)(Lmax + 1) * sizeof(int);
  size_t lds_chains = (size_t)spw * (Lmax + 1) * sizeof(int) + 16, lds_emis = 0;
  MGR_REQUIRE(lds_grad <= 160 * 1024, "C=%d too large for the LDS occupancy tile", C);
  // tune keys 20 / 21 = KiB of LDS the recurrence / the per-frame kernels ask for at least.  Placement: a workgroup that asks for more
  // than a persistent scan workgroup leaves on its CU can only land on a CU without one (the engine sets them for the steps of its
  // fused schedule, where this call runs beside 208 whole-CU scan workgroups: 96 KiB = the 32 recurrence workgroups on a CU each)
  if (c->tune[20] > 0 && lds_chains < (size_t)c->tune[20] * 1024) lds_chains = (size_t)c->tune[20] * 1024;
  if (c->tune[21] > 0) {
    if (lds_grad < (size_t)c->tune[21] * 1024) lds_grad = (size_t)c->tune[21] * 1024;
    lds_emis = (size_t)c->tune[21] * 1024;
  }
  if (!(c->attr_done & 128u)) {
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_emissions), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_grad), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_chains<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->attr_done |= 128u;
  }
  int npairs = Lmax + 1;
  int ppl = (npairs + 63) / 64;
  hipStream_t s = mgr_stream(c);
  mgr_prof_begin(c, MGR_K_CTC);
  hipLaunchKernelGGL(k_ctc_emissions, dim3((unsigned)((To + 255) / 256), B), dim3(256), lds_emis, s, P, input_len, label_len, B, T, C, Lmax, skip, eps, LY, AL, BE);
#define MGR_CTC_LAUNCH(N)                                                                                                              \
  do {                                                                                                                                 \
    if (spw == 2)                                                                                                                      \
      hipLaunchKernelGGL((k_ctc_chains<N, 2>), dim3((B + 1) / 2), dim3(256), lds_chains, s, labels, input_len, label_len, B, T, C, Lmax, \
                         skip, blank, loss, LY, AL, BE);                                                                               \
    else                                                                                                                               \
      hipLaunchKernelGGL((k_ctc_chains<N, 1>), dim3(B), dim3(128), lds_chains, s, labels, input_len, label_len, B, T, C, Lmax, skip,   \
                         blank, loss, LY, AL, BE);                                                                                     \
  } while (0)
  switch (ppl) {
    case 1: MGR_CTC_LAUNCH(1); break;
    case 2: MGR_CTC_LAUNCH(2); break;
    case 3: MGR_CTC_LAUNCH(3); break;
    default: MGR_CTC_LAUNCH(4); break;
  }
#undef MGR_CTC_LAUNCH
  if (dLogits)
    hipLaunchKernelGGL(k_ctc_grad, dim3((unsigned)((T + 255) / 256), B), dim3(256), lds_grad, s, P, labels, input_len, label_len, B, T, C, Lmax, skip,
                       blank, eps, gscale, loss, dLogits, LY, AL, BE);
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_CTC);
  return 0;
}

}  // extern "C"
