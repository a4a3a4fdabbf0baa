// Device helpers shared by the LSTM scan kernels (Keras LSTMCell semantics; reference call sites
// multimodal_fusion/multimodal.py:159-168: activation='tanh', recurrent_activation='hard_sigmoid').
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mgr_hsig(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }
// tanh via one v_exp_f32 + one v_rcp_f32 (1 ulp; __frcp_rn would be the ten-instruction IEEE division sequence, and vector
// instructions are step time in the scans: they do not overlap with the SIMD's MFMAs): |abs err| ~1e-7, saturates cleanly to +-1
__device__ __forceinline__ float mgr_tanh(float x) {
  float e = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ float mgr_hsig_grad(float a) { return (a > 0.f && a < 1.f) ? 0.2f : 0.f; }

// One LSTM cell step on activated pre-activations z (i,f,c,o); returns h, updates c, emits activated gates.
__device__ __forceinline__ float mgr_cell_fwd(const float zi, const float zf, const float zc, const float zo, float& c,
                                              float4& gates) {
  float i = mgr_hsig(zi), f = mgr_hsig(zf), g = mgr_tanh(zc), o = mgr_hsig(zo);
  c = f * c + i * g;
  gates = make_float4(i, f, g, o);
  return o * mgr_tanh(c);
}

// One BPTT cell step. dh: total dL/dh_t; dc_carry: dL/dc_t arriving from step t+1 (in/out).
__device__ __forceinline__ float4 mgr_cell_bwd(float dh, float4 g4, float c, float c_prev, float& dc_carry) {
  float i = g4.x, f = g4.y, g = g4.z, o = g4.w;
  float tc = mgr_tanh(c);
  float dO = dh * tc;
  float dc = dh * o * (1.f - tc * tc) + dc_carry;
  float di = dc * g, df = dc * c_prev, dg = dc * i;
  dc_carry = dc * f;
  return make_float4(di * mgr_hsig_grad(i), df * mgr_hsig_grad(f), dg * (1.f - g * g), dO * mgr_hsig_grad(o));
}
