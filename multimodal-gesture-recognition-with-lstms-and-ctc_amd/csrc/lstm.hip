// C-ABI entry points of the LSTM recurrence (K3, K7-scan): pick the kernel family for the shape.
//   H in the weight-stationary set  -> lstm_mfma.hip   (one CU per 16-sample group, U in VGPRs)
//   anything else (H <= 1024)       -> lstm_simple.hip (U streamed from L2; correctness fallback)
#include "common.h"

int mgr_scan_fwd_simple(mgr_ctx*, const float*, const float*, float*, int, const float*, int, float*, float*, int, int, int, int);
int mgr_scan_bwd_simple(mgr_ctx*, const float*, int, const float*, const float*, const float*, float*, int, int, int, int);
int mgr_scan_fwd_mfma(mgr_ctx*, const float*, const float*, float*, int, const float*, int, float*, float*, int, int, int, int);
int mgr_scan_bwd_mfma(mgr_ctx*, const float*, int, const float*, const float*, const float*, float*, int, int, int, int);

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

size_t mgr_lstm_scan_ws_bytes(int B, int T, int H) {
  (void)B;
  (void)T;
  return mgr_align_up((size_t)4 * H * H * sizeof(float), 256);  // U^T for the fallback backward kernel
}

int mgr_lstm_scan_fwd(mgr_ctx* c, const float* Z, const float* Up, float* Y, int ldy, const float* R, int ldr,
                      float* gates, float* cs, int B, int T, int H, int reverse, void* ws, size_t ws_bytes) {
  (void)ws;
  (void)ws_bytes;
  MGR_REQUIRE(c && Z && Up && Y, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && H > 0 && ldy >= H && (!R || ldr >= H), "bad shape");
  MGR_REQUIRE(aligned16(Z) && aligned16(Up) && (!gates || aligned16(gates)), "Z/Up/gates must be 16-byte aligned");
  int r = mgr_prof_begin(c, MGR_K_SCAN_FWD);
  if (r) return r;
  r = mgr_scan_fwd_mfma(c, Z, Up, Y, ldy, R, ldr, gates, cs, B, T, H, reverse);
  if (r == 0) r = mgr_scan_fwd_simple(c, Z, Up, Y, ldy, R, ldr, gates, cs, B, T, H, reverse);
  if (r < 0) return r;
  return mgr_prof_end(c, MGR_K_SCAN_FWD);
}

int mgr_lstm_scan_bwd(mgr_ctx* c, const float* dY, int lddy, const float* gates, const float* cs, const float* Up,
                      float* dZ, int B, int T, int H, int reverse, void* ws, size_t ws_bytes) {
  MGR_REQUIRE(c && dY && gates && cs && Up && dZ, "null argument");
  MGR_REQUIRE(B > 0 && T > 0 && H > 0 && lddy >= H, "bad shape");
  MGR_REQUIRE(aligned16(gates) && aligned16(Up) && aligned16(dZ), "gates/Up/dZ must be 16-byte aligned");
  int r = mgr_prof_begin(c, MGR_K_SCAN_BWD);
  if (r) return r;
  r = mgr_scan_bwd_mfma(c, dY, lddy, gates, cs, Up, dZ, B, T, H, reverse);
  if (r == 0) {
    MGR_REQUIRE(ws && ws_bytes >= mgr_lstm_scan_ws_bytes(B, T, H), "workspace too small");
    float* UpT = reinterpret_cast<float*>(ws);
    r = mgr_transpose(c, Up, UpT, H, 4 * H);
    if (r) return r;
    r = mgr_scan_bwd_simple(c, dY, lddy, gates, cs, UpT, dZ, B, T, H, reverse);
  }
  if (r < 0) return r;
  return mgr_prof_end(c, MGR_K_SCAN_BWD);
}

}  // extern "C"
