// K9: CTC prefix beam search (placeholder translation unit; the kernel lands in a later commit of this round).
#include "common.h"

extern "C" {

size_t mgr_ctc_beam_ws_bytes(int B, int T, int C, int beam) {
  (void)B; (void)T; (void)C; (void)beam;
  return 256;
}

int mgr_ctc_beam_search(mgr_ctx* c, const float* P, const int32_t* input_len, int B, int T, int C, int skip, int blank,
                        int beam, float eps, int merge_repeated, int32_t* out, int32_t* out_len, double* logp, void* ws,
                        size_t ws_bytes) {
  (void)c; (void)P; (void)input_len; (void)B; (void)T; (void)C; (void)skip; (void)blank; (void)beam; (void)eps;
  (void)merge_repeated; (void)out; (void)out_len; (void)logp; (void)ws; (void)ws_bytes;
  return mgr_fail(-4, "mgr_ctc_beam_search: not implemented yet");
}

}  // extern "C"
