/*
 * mgr.h - C ABI of libmgr.so: the MI355X (gfx950) BiLSTM + CTC training / decode hot path.
 *
 * The reference (AlexGidiotis/Multimodal-Gesture-Recognition-with-LSTMs-and-CTC) has NO native or FFI
 * interface: its hot path is whatever Keras 2.1.4 / TensorFlow 1.12.1 execute for the Python call sites
 * cited on each entry point below (paths relative to the reference root).  This header is therefore the
 * boundary the build defines: plain pointers and sizes, no framework types.  The Python host
 * (mgr_amd/_capi.py) binds it with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; mgr_last_error() (thread-local) explains.
 *   - all tensors are row-major fp32 unless stated; integer inputs are int32.
 *   - device pointers come from mgr_alloc(); the caller owns every buffer; the library keeps no caller
 *     pointer past the call.  Scratch is passed explicitly (ws, ws_bytes) - see the *_ws_bytes queries.
 *   - one mgr_ctx per device; calls on a ctx are serialised by the caller.  Kernels are enqueued on the
 *     ctx's CURRENT stream (mgr_stream_set, 8 streams); mgr_sync() waits for all of them.
 *   - LSTM weights use the "packed" gate-interleaved layout: column u*4+g of a packed matrix is column
 *     g*H+u of the Keras matrix (g in i,f,c,o).  mgr_lstm_pack converts both ways.  Z / dZ (gate
 *     pre-activations and their gradients) use the same interleaved column order.
 */
#ifndef MGR_H_
#define MGR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgr_ctx mgr_ctx;
typedef struct mgr_comm mgr_comm;

#define MGR_NUM_STREAMS 8
#define MGR_NUM_EVENTS 64
#define MGR_UNIQUE_ID_BYTES 128

/* ---- library / context ------------------------------------------------------------------------------ */
int mgr_version(void);
const char* mgr_last_error(void);
int mgr_device_count(int* n);
int mgr_ctx_create(int device, mgr_ctx** out);
int mgr_ctx_destroy(mgr_ctx* ctx);
/* name may be NULL */
int mgr_device_info(mgr_ctx* ctx, int* cu_count, size_t* hbm_bytes, char* name, int name_len);

/* ---- memory / streams / events ---------------------------------------------------------------------- */
int mgr_alloc(mgr_ctx* ctx, size_t bytes, void** dptr);
int mgr_free(mgr_ctx* ctx, void* dptr);
int mgr_memset(mgr_ctx* ctx, void* d, int byte, size_t n);
/* Keras feeds numpy arrays through feed_dict (multimodal_fusion/multimodal.py:264); these are that copy. */
int mgr_h2d(mgr_ctx* ctx, void* d, const void* h, size_t n);
/* Pinned host memory + a copy that does NOT synchronise: the host returns at once and the copy is ordered on the current
 * stream.  (mgr_h2d on pageable memory blocks the host, and on this runtime for as long as OTHER streams have work queued.) */
int mgr_host_alloc(mgr_ctx* ctx, size_t bytes, void** out);
int mgr_host_free(mgr_ctx* ctx, void* p);
int mgr_h2d_async(mgr_ctx* ctx, void* d, const void* h_pinned, size_t n);
int mgr_d2h(mgr_ctx* ctx, void* h, const void* d, size_t n);
/* The way back without a host stall: device -> pinned host memory (mgr_host_alloc), ordered on the current stream; the bytes are
 * valid once an event recorded behind the call has completed (mgr_event_record + mgr_event_sync).  What predict_generator's
 * pipeline downloads posteriors / decoded paths with while the next batch is computed (sequence_decoding.py:118-127). */
int mgr_d2h_async(mgr_ctx* ctx, void* h_pinned, const void* d, size_t n);
int mgr_event_sync(mgr_ctx* ctx, int ev);                 /* the host waits for event ev as last recorded */
int mgr_d2d(mgr_ctx* ctx, void* dst, const void* src, size_t n);
int mgr_sync(mgr_ctx* ctx);
int mgr_stream_set(mgr_ctx* ctx, int idx);
int mgr_stream_wait(mgr_ctx* ctx, int waiter, int waited); /* waiter waits for everything queued on waited */
/* Dispatch priority of stream idx: level 1 high, 0 default, -1 low (hipStreamCreateWithPriority).  The stream is recreated: it must be
 * idle (the call waits for it) and is best set before first use.  What it buys: when two streams have chip-filling GEMMs ready at the same
 * time, the workgroups of the higher-priority stream are placed first - the engine gives the stream that carries a training step's
 * dependent chain (fusion projections -> scan -> CTC -> BPTT -> dW -> Adam) the chip when the encoder stream has slack
 * (Schedule.chain_stream_priority). */
int mgr_stream_set_priority(mgr_ctx* ctx, int idx, int level);
int mgr_event_record(mgr_ctx* ctx, int ev);               /* on the current stream */
int mgr_stream_wait_event(mgr_ctx* ctx, int waiter, int ev); /* stream `waiter` waits for event ev as last recorded */
int mgr_event_elapsed_ms(mgr_ctx* ctx, int ev0, int ev1, float* ms);
/* Per-kernel-family device timing (HIP events around each launch, on the launch stream).
 * family ids: MGR_K_*.  mgr_prof_get syncs the device and returns accumulated launches / milliseconds. */
enum {
  MGR_K_GEMM_NN = 0, MGR_K_GEMM_TN = 1, MGR_K_GEMM_NT = 2, MGR_K_SCAN_FWD = 3, MGR_K_SCAN_BWD = 4,
  MGR_K_DENSE_FWD = 5, MGR_K_DENSE_BWD = 6, MGR_K_CTC = 7, MGR_K_ADAM = 8, MGR_K_MISC = 9, MGR_K_ALLREDUCE = 10,
  /* (round 6) multi-scan calls whose widest layer has H <= 128 - the fusion layer's own scan - are counted apart from the encoder
   * depths (MGR_K_SCAN_FWD): a roofline figure of the dominant kernel must not average it with a launch of 3 % of its FLOP */
  MGR_K_SCAN_FWD_NARROW = 11, MGR_K_COUNT = 12
};
int mgr_prof_enable(mgr_ctx* ctx, int family_mask);
int mgr_prof_get(mgr_ctx* ctx, int family, int* launches, float* ms);
int mgr_prof_reset(mgr_ctx* ctx);

/* ---- K1: GaussianNoise (multimodal_fusion/multimodal.py:103-106) ------------------------------------ */
/* Y = X + stddev * N(0,1), counter-based RNG keyed by (seed, element index).  X may equal Y. */
int mgr_add_gaussian_noise(mgr_ctx* ctx, const float* X, float* Y, size_t n, float stddev, uint64_t seed);
/* mask[i] = (uniform(seed,i) >= p) ? 1/(1-p) : 0  - Keras dropout mask (inverted dropout) */
int mgr_dropout_mask(mgr_ctx* ctx, float* mask, size_t n, float p, uint64_t seed);

/* ---- K2/K3/K7: Bidirectional(LSTM) (multimodal_fusion/multimodal.py:109-118,159-168;
 *      audio_network/speech_lstm_ctc_words.py:56-77; skeletal_network/skeletal_lstm_ctc.py:309-335) ---- */
/* Keras layout <-> packed layout for a [rows, 4H] matrix (W: rows=F, U: rows=H, b: rows=1). */
int mgr_lstm_pack(mgr_ctx* ctx, const float* src, float* dst, int rows, int H, int to_keras);
/* dst[c][r] = src[r][c] */
int mgr_transpose(mgr_ctx* ctx, const float* src, float* dst, int rows, int cols);
/* Gate pre-activations for all T:  Z[b,t,:] = (X[b,t,:F] (.) mask4[g,b,:]) . Wp + bp   (f32 MFMA GEMM).
 * X has row stride ldx floats (>= F).  mask4 [4,B,F] may be NULL (no input dropout).  Z is [B,T,4H] packed order. */
int mgr_lstm_input_proj(mgr_ctx* ctx, const float* X, int ldx, const float* mask4, const float* Wp,
                        const float* bp, float* Z, int B, int T, int F, int H);
/* The same for BOTH directions of a Bidirectional layer (multimodal.py:159-168: two LSTMs over the same input, each with
 * its own kernel, bias and dropout masks) as one GEMM over 8H columns when that saves column tiles (4H = 400: 7 instead
 * of 2 x 4); otherwise it is two mgr_lstm_input_proj calls.  Results are identical to the two calls. */
int mgr_lstm_input_proj_pair(mgr_ctx* ctx, const float* X, int ldx, const float* mask4_fwd, const float* Wp_fwd,
                             const float* bp_fwd, float* Z_fwd, const float* mask4_rev, const float* Wp_rev,
                             const float* bp_rev, float* Z_rev, int B, int T, int F, int H);
/* The same projection for a layer with Keras input dropout at rate drop_rate (the rate only selects the kernel; mask4
 * holds the actual factors, 0 or 1/(1-p)): from drop_rate >= 0.3 on (and 16 <= F <= 2048) the K loop runs per gate
 * over the kept features only (gemm.hip, k_gemm_nn_sparse) - the same sums with the zero terms left out, i.e. equal to
 * mgr_lstm_input_proj up to fp32 summation order.  ws from mgr_lstm_input_proj_dropout_ws_bytes (index lists, rebuilt by
 * every call).  tune key 9 = 1 keeps the dense kernel. */
size_t mgr_lstm_input_proj_dropout_ws_bytes(int B, int F, int H);
int mgr_lstm_input_proj_dropout(mgr_ctx* ctx, const float* X, int ldx, const float* mask4, float drop_rate,
                                const float* Wp, const float* bp, float* Z, int B, int T, int F, int H, void* ws,
                                size_t ws_bytes);
/* The same projection from a TRANSPOSED copy of the layer input, XT[b][f][t] with row stride ldt (T padded to a multiple of
 * 128, zeros behind T; mgr_transpose_bt writes it): a kept feature is then a contiguous row, and the kernel's A operand is
 * staged with coalesced 16-byte loads instead of one scattered 4-byte load per element.  Only for shapes the dropout-aware
 * kernel handles - mgr_lstm_input_proj_dropout_wants_transposed says whether a copy is worth making for (drop_rate, F).
 * x_absmax: a bound on |XT|, or 0 if there is none.  With a bound the products run on the f16 matrix pipe with every f32 operand
 * split into an f16 (hi, lo) pair of its scaled value and f32 accumulation (22+ significant bits per operand: gemm.hip,
 * k_gemm_nn_sparse16; the representation error is below the rounding an f32 accumulation of the same length commits).
 *   x_absmax > 0: a bound the CALLER states.  It is checked on the device in front of the product (k_absmax_gate: one pass over
 *     XT); data that would leave the f16 range at the scale the bound implies make the call fall back - on the device, without a
 *     host round trip - to the f32 MFMA kernel, so a wrong bound costs time, never Inf / NaN in Z.
 *   x_absmax < 0: |x_absmax| is a bound the PRODUCER of XT guarantees by construction - the transposed copies the scans of this
 *     library write (mgr_scan_job.YT) hold h = o * tanh(c), |h| <= 1, or the sum of two of them (residual input), <= 2 - and no
 *     check is made.  This is what the engine passes for the buffers its own scans fill.
 *   0, or tune key 15 = 1: v_mfma_f32_32x32x2_f32.
 * mask4 may be NULL (no dropout: inference): with a bound the plain dense projection on the f16 pipe (k_gemm_nn_dense16: one
 * K loop over all features, the A tile staged once for the four gates; tune key 10 = 2 takes it with a mask as well, the mask
 * factors folded into the weight tiles), without one the f32 kernel over all features. */
int mgr_lstm_input_proj_dropout_wants_transposed(mgr_ctx* ctx, float drop_rate, int F);
int mgr_lstm_input_proj_dropout_t(mgr_ctx* ctx, const float* XT, int ldt, const float* mask4, float drop_rate,
                                  const float* Wp, const float* bp, float* Z, int B, int T, int F, int H, void* ws,
                                  size_t ws_bytes, float x_absmax);
int mgr_transpose_bt(mgr_ctx* ctx, const float* X, int ldx, float* XT, int ldt, int B, int T, int F);
/* The same projection from a PRE-SPLIT transposed copy (round 5; gemm_split.hip).  XS has the shape and strides of XT - [B][F][ldt]
 * floats' worth of bytes - but the 4 ldt bytes of row (b, f) hold ldt f16 values hi(t) followed by ldt f16 values lo(t) with
 * x 2^13 = hi + lo, hi = rn_f16(x 2^13), lo = rn_f16(x 2^13 - hi): the split-f16 operand pair of the f16 matrix pipe, made ONCE by
 * whoever produces the activations instead of by every product that reads them.  Producers: the scans (mgr_scan_job.yt_split - their
 * outputs are bounded by construction: |h| <= 1, with a residual sum <= 2) and mgr_transpose_bt_split (any row-major tensor with
 * |x| < 7.99; a larger value overflows the f16 range and shows as Inf / NaN).  ldt % 128 == 0, zero behind T.
 * The kernel is a loader + matrix pipeline: both operands travel HBM -> LDS by LDS-DMA through a three-stage ring (no register
 * staging, no conversion), fragments come out of LDS with transposed reads, the waves issue nothing but those and MFMAs.
 * mask4: entries 0 or ONE common factor c (what mgr_dropout_mask and Keras' Dropout produce: c = 1 / (1 - rate)); checked on the
 * device - two different factors in one call make Z NaN (the factor is applied once, in the epilogue).  NULL: no dropout.
 * drop_rate is informative only.  16 <= F <= 2048. */
size_t mgr_lstm_input_proj_dropout_ts_ws_bytes(int B, int F, int H);
int mgr_lstm_input_proj_dropout_ts(mgr_ctx* ctx, const float* XS, int ldt, const float* mask4, float drop_rate,
                                   const float* Wp, const float* bp, float* Z, int B, int T, int F, int H, void* ws,
                                   size_t ws_bytes);
/* XS[b][f] = the split row (above) of X[b][0..T)[f], zero for t in [T, ldt); ldt % 8 == 0. */
int mgr_transpose_bt_split(mgr_ctx* ctx, const float* X, int ldx, float* XS, int ldt, int B, int T, int F);
/* the same with entry t of a row = X[b, t + tshift, f] (tshift in {-1, 0, 1}; zero where that step does not exist) */
int mgr_transpose_bt_split_shift(mgr_ctx* ctx, const float* X, int ldx, float* XS, int ldt, int B, int T, int F, int tshift);
/* Frozen weights: frozen != 0 promises that the contents of Wp (a packed input-weight matrix passed to mgr_lstm_input_proj_dropout_ts) do
 * not change until the next call of this function for the same pointer; the (hi, lo) weight planes and the largest |W| that
 * mgr_lstm_input_proj_dropout_ts leaves in its workspace are then reused by later calls with the same (Wp, workspace, F, H) instead of
 * being rebuilt (the frozen encoders of the fusion network, multimodal_fusion/multimodal.py:118-130: 4 of the 6 conversions of a step).
 * Every call (either value) drops what was kept for Wp - call it again after rewriting the weights.  Host-side state only. */
int mgr_weight_planes_cache(mgr_ctx* ctx, const float* Wp, int frozen);
/* Recurrence. reverse=1 walks t = T-1..0 and writes outputs at their original t (Bidirectional backward
 * sub-layer).  Y[b,t,0:H] with row stride ldy gets h_t (+ R[b,t,0:H] with stride ldr when R != NULL: the
 * residual add / concat fusion of multimodal.py:111,117,155).  gates [B,T,H,4] (i,f,g,o after activation) and
 * cs [B,T,H] are saved for BPTT when non-NULL.  ws from mgr_lstm_scan_ws_bytes (may be 0/NULL). */
size_t mgr_lstm_scan_ws_bytes(int B, int T, int H);
int mgr_lstm_scan_fwd(mgr_ctx* ctx, const float* Z, const float* Up, float* Y, int ldy, const float* R,
                      int ldr, float* gates, float* cs, int B, int T, int H, int reverse, void* ws,
                      size_t ws_bytes);
/* Several independent recurrences (e.g. audio fwd/rev + skeletal fwd/rev of one encoder depth) in ONE call.
 * Layers whose recurrent matrix does not fit one CU (H = 300, 500) run as a single persistent multi-CU launch
 * (clusters of CUs exchanging h_t each step, lstm_cluster.hip); sharing the launch keeps every spinning
 * workgroup co-resident.  Fields as in mgr_lstm_scan_fwd.  ws from mgr_lstm_scan_multi_ws_bytes. */
typedef struct mgr_scan_job {
  const float* Z;
  const float* Up;
  float* Y;
  const float* R;
  float* gates;
  float* cs;
  int ldy, ldr, B, T, H, reverse;
  /* optional transposed copy of the output, written by the scan itself (the K-split multi-CU kernel stages 32 steps per lane
   * in LDS and stores 128-byte row segments; any other kernel family is followed by a transpose inside the call):
   * YT[b * ytb + u * ldt + t] = what Y[b, t, u] gets, u < H; t in [T, ldt) is written as zero (the buffer may arrive dirty).
   * ldt % 4 == 0, ldt >= T rounded up to 32.  NULL: none.  This is the layout mgr_lstm_input_proj_dropout_t / mgr_lstm_param_grads_dropout_t
   * read; with ytb > H * ldt several jobs fill column ranges of one wider [B][F][ldt] copy. */
  float* YT;
  long long ytb;
  int ldt;
  /* yt_split != 0: the rows of YT are written in the SPLIT ROW FORMAT of mgr_lstm_input_proj_dropout_ts - row (b, u) holds ldt f16
   * values hi(t) followed by ldt f16 values lo(t) of y 2^13 - instead of ldt floats: what the pre-split products read without
   * converting anything (the scan's outputs are bounded by construction: |h| <= 1, with a residual input <= 2).  ldt % 8 == 0. */
  int yt_split;
} mgr_scan_job;
size_t mgr_lstm_scan_multi_ws_bytes(int njobs, const mgr_scan_job* jobs);
int mgr_lstm_scan_fwd_multi(mgr_ctx* ctx, int njobs, const mgr_scan_job* jobs, void* ws, size_t ws_bytes);
/* Explicit launch options of the two multi-scan calls (round 6; the same calls, with the choices a scheduler makes PER CALL passed
 * as arguments instead of through context-wide tune keys, and with the launch number handed back).
 *   struct_size  sizeof(mgr_scan_launch_opts) of the CALLER's header: members beyond it are taken as zero, so the struct can grow.
 *   form         forward call: MGR_SCAN_FORM_*; backward call: MGR_BPTT_FORM_*.  AUTO (0) = what tune keys 4 / 16 say.
 *   seq_out      host word (ordinary or page-locked memory; NULL: not wanted) that receives, BEFORE the call returns, the launch
 *                number of the persistent multi-CU launch this call enqueued - what mgr_stream_wait_resident takes - or MGR_SEQ_NONE
 *                when the call enqueued no launch that enters the residency ledger (single-CU kernels, fallbacks).  With a word from
 *                mgr_host_alloc this is the hand-over for a wait that was enqueued BEFORE the launch it waits for
 *                (mgr_stream_wait_resident_word). */
enum { MGR_SCAN_FORM_AUTO = 0, MGR_SCAN_FORM_PLAIN = 1, MGR_SCAN_FORM_PAIR = 2, MGR_SCAN_FORM_FUSED = 3,
       MGR_SCAN_FORM_FUSED_ANY = 4 };   /* FUSED: launches that do not fit one workgroup per CU as they are; FUSED_ANY: every launch the
                                         * fused kernel can run (a narrow layer then holds ceil(G / 2) whole CUs per cluster) */
enum { MGR_BPTT_FORM_AUTO = 0, MGR_BPTT_FORM_TRIMMED = 1, MGR_BPTT_FORM_YIELDING = 2, MGR_BPTT_FORM_DIRECT = 3,   /* tune key 16 = 0 / 1 / 2 */
       /* round 6, narrow layers (16 < H <= 128) with an exchange: 8-wave workgroups that run TWO unit groups of their cluster, a CU each
        * (H = 100: 32 workgroups instead of 56), with the trimmed step / the direct gather inside; the same bits as every other form.  A
        * launch that does not qualify takes the trimmed / direct form. */
       MGR_BPTT_FORM_FUSED = 4, MGR_BPTT_FORM_FUSED_DIRECT = 5,
       /* round 6, the directions of ONE narrow layer (same shape, H in {32, 64, 100}): one 8-wave workgroup per (direction, 16-sample
        * group) that holds the whole recurrent matrix as f16 (hi, lo) fragments and keeps dh in registers - NO inter-CU exchange, so its
        * step does not depend on what the rest of the chip does to the L2 / fabric (lstm_cu_bwd.hip).  Same arithmetic as the multi-CU
        * forms with another summation order: equal to them to rounding, not bit for bit.  A call that does not qualify takes the
        * trimmed multi-CU form.  mgr_scan_bwd_job.dzmax is filled by a reduction pass behind the kernel. */
       MGR_BPTT_FORM_SINGLE_CU = 6 };
#define MGR_SEQ_NONE 0xFFFFFFFFu
typedef struct mgr_scan_launch_opts {
  unsigned struct_size;
  int form;
  unsigned* seq_out;
} mgr_scan_launch_opts;
int mgr_lstm_scan_fwd_multi_ex(mgr_ctx* ctx, int njobs, const mgr_scan_job* jobs, void* ws, size_t ws_bytes,
                               const mgr_scan_launch_opts* opts);
/* ABI guard for bindings that fill mgr_scan_job / mgr_scan_bwd_job / mgr_scan_launch_opts field by field: the sizes the LIBRARY was
 * built with (round 5 appended mgr_scan_bwd_job.dzmax and turned mgr_scan_job's reserved word into yt_split; revision 7 appended
 * mgr_scan_bwd_job.dbsum and the dbsum / proj_ws / HsT arguments of mgr_lstm_param_grads_dropout_ts - a caller built against
 * an older header must not pass its structs to this library).  out[0..2] = sizeof of the three structs, out[3] =
 * MGR_ABI_REVISION. */
#define MGR_ABI_REVISION 7
int mgr_abi_struct_sizes(unsigned out[4]);
/* Tuning / test hooks.  key 0 (MGR_TUNE_SCAN_PATH): 0 auto, 1 force the L2-streaming fallback kernels,
 * 2 force one workgroup per batch group (no inter-CU exchange) where it fits, 3 force clusters with 4 tiles per
 * workgroup, 4 same with 8 tiles per workgroup.  key 1: !=0 makes scan_fwd check the give-up word synchronously.
 * key 2: print the scan plan.  key 3: 1 = K-split scan launches keep contiguous cluster ids and write-through publishes (no
 *        XCD-local exchange).
 * key 4: 2 = split-f16 K-split scan launches take the PAIR form (two 16-sample groups per workgroup, one workgroup per CU) whenever
 *        they qualify; bit-identical to the default (two workgroups per CU) and slower: kept as a measured alternative.
 *        3 = launches that do not fit ONE workgroup per CU as they are (config F's encoder depths: 408 workgroups) take the FUSED form:
 *        8-wave workgroups that run two unit groups of their cluster, a CU each (208), bit-identical; the CUs they leave free are what
 *        4-wave persistent launches of other streams get - the admission ledger then counts those by the CUs they need two to a CU,
 *        and their caller starts them once the fused launch is resident (mgr_stream_wait_resident) so that they do land there.
 * key 7: one-tile-per-wave clusters: 0 = K-split step (register-direct gather), 1 = LDS-image step.
 * key 8: 1 = the multi-CU BPTT keeps its 4-wave kernel instead of the split-role (4 compute + 4 gather waves) one, 2 = always split.
 * key 9: 1 = mgr_lstm_input_proj_dropout always takes the dense kernel.
 * key 10: 2 = mgr_lstm_input_proj_dropout_t with a bound on |XT| takes the dense-K split-f16 kernel with a mask as well.
 * key 12: mgr_lstm_input_proj_dropout_ts tile: 0 = the library's choice, 1 = 128 x 64 (4 waves, two workgroups per CU), 2 = 128 x 128 (8 waves).
 * key 13: 1 = mgr_dense_softmax_fwd / mgr_dense_bwd keep their LDS-tiled vector-ALU kernels where the matrix-core forms would run.
 * key 14: K-split scan step: 0 = recurrent product on the f16 matrix pipe with every f32 operand split into an f16 (hi, lo) pair and
 *        f32 accumulation (22+ significant bits per operand; lstm_cluster.hip cluster_run_k16), 1 = v_mfma_f32_16x16x4_f32.
 * key 15: 1 = the transposed-input projection / parameter-gradient GEMMs keep their f32 MFMA kernels whatever bound the caller states.
 * key 20 / 21: KiB of LDS the CTC recurrence kernel / the CTC per-frame kernels ask for at least (0: what they use).  A placement hint
 *         for callers that run mgr_ctc_loss_grad / mgr_head_fwd_bwd beside persistent scan launches of another stream: a workgroup that
 *         asks for more LDS than a scan workgroup leaves on its CU lands on a CU without one (engine.py sets 96 / 64 for such steps).
 * key 19: 1 = mgr_lstm_scan_bwd_multi[_ex] with form AUTO takes MGR_BPTT_FORM_SINGLE_CU where it qualifies (A/B of whole runs).
 * key 18: 1 = mgr_ctc_loss_grad runs one sample per workgroup (rounds 1 - 5); 0 = two (from B = 2 on: the alpha / beta chains of a
 *         workgroup's two samples on its four SIMDs - 32 workgroups for config F's 64 samples, which the 48 CUs beside fused encoder scans
 *         hold one per CU); the same bits.
 * key 17: 1 = the two halves of a FUSED scan workgroup each fetch and verify the whole h image of the step themselves (round 5's
 *         k_scan_cluster_k16f); 0 = they share ONE gather through LDS (k_scan_cluster_k16fs: half the L2 traffic); the same bits.
 * key 16: 1 = the BPTT of narrow layers (H <= 128) launched next runs BESIDE persistent scans of another stream: it takes the form that
 *         yields to them (two barriers, partial sums through LDS) instead of the one trimmed along its dependent chain, which is faster
 *         alone (H = 100: 1.77 against 2.29 us per step) and costs the step beside them; same results bit for bit.  The engine sets it
 *         from its schedule (a deterministic choice: it never depends on what happens to be running).
 *         2 = the direct gather: every wave fetches the words of its own cells from all sources, no partial sums through LDS, one barrier
 *         per step - the fastest form alone (1.58 us per step), 4.7 x the texture-path traffic: for launches that have CUs of their own
 *         (beside fused encoder scans).  All three forms give the same bits. */
enum { MGR_TUNE_SCAN_PATH = 0, MGR_TUNE_COUNT = 24 };
int mgr_tune(mgr_ctx* ctx, int key, int value);
int mgr_tune_get(mgr_ctx* ctx, int key, int* value);   /* what a key is set to (a host of the library that lays out buffers by it) */
/* Health of the persistent multi-CU scans launched on this context since the last mgr_scan_status_clear: *out receives the OR
 * of their status bits.  MGR_SCAN_GAVE_UP: a bounded spin expired (a dead-locked or lost peer) - the launch returned promptly
 * but its outputs are garbage; the call FAILS (mgr_last_error).  MGR_SCAN_NONFINITE: a hidden state became NaN / Inf (diverged
 * weights, bad checkpoint): the output Y of that (sample, unit) is NaN from that time step on (the multi-CU exchange feeds 0 back
 * in its place - a NaN word cannot travel through the hand-off - so the OTHER units of the sample stay finite where the
 * reference's would turn NaN one step later; whoever consumes the outputs must treat the whole pass as NaN, as Engine.read_loss
 * does); the call succeeds.  Ordered on the current stream; cheap (one 4-byte read back) - call it where results are consumed,
 * e.g. with the loss. */
enum { MGR_SCAN_GAVE_UP = 1, MGR_SCAN_NONFINITE = 8 };
int mgr_scan_status(mgr_ctx* ctx, unsigned* out);
/* The same read without the failure: out[0] = status bits, out[2] = optimizer updates skipped by the update gate (below) since
 * the last clear, out[1] = out[3] = 0. */
int mgr_scan_status_ex(mgr_ctx* ctx, unsigned out[4]);
/* Forget the recorded status bits and the skipped-update count (enqueued on the current stream), e.g. after restoring a good
 * checkpoint. */
int mgr_scan_status_clear(mgr_ctx* ctx);
/* Several engines may share one context: each binds its OWN status block (>= 64 zeroed bytes from mgr_alloc, 16-byte aligned)
 * before it enqueues work; scans launched while a block is bound report into it and mgr_scan_status* / the update gate read
 * it.  NULL binds the context's own block again.  Host-side state only (no stream order).
 * Words of a block: [0] status bits, [2] skipped updates, [8, 16) WHICH samples met a non-finite hidden state - bit (b mod 256) of
 * the 256-bit field is set together with MGR_SCAN_NONFINITE by the scan that saw sample b of its job go NaN / Inf, so that a
 * consumer can hand out NaN for exactly those samples (Engine.predict: a block of its own per inference pass). */
int mgr_scan_status_bind(mgr_ctx* ctx, void* block);
/* Test hook: OR `bits` into the bound status block on the current stream, as a scan that gave up would. */
int mgr_scan_status_inject(mgr_ctx* ctx, unsigned bits);
/* Update gate: keeps a bad step away from the weights WITHOUT a host round trip (the host reads the status with the loss, by
 * which time the optimizer kernels of the same step are already queued).  mgr_update_gate_eval writes flag[0] = 1.0f if
 * (status bits & mask) else 0.0f on the current stream - put flag behind the gradient buffer and it travels with the gradient
 * all-reduce, so that every replica takes the same decision (sum > 0 on all ranks if any rank raised it).  While a flag is set
 * with mgr_update_gate_set (host-side state; NULL = gate open), mgr_adam_step and mgr_maxnorm_cols read it on the device and
 * leave parameters and moments untouched when it is non-zero; each skipped mgr_adam_step is counted (mgr_scan_status_ex). */
int mgr_update_gate_eval(mgr_ctx* ctx, unsigned mask, float* flag);
int mgr_update_gate_set(mgr_ctx* ctx, const float* flag);
/* Placement aid (never a correctness dependency): the current stream waits, on the device, until every workgroup of the NEXT
 * persistent scan launched on this context (on any stream) has started, or timeout_us (<= 100000) has passed.  Chip-filling
 * GEMMs enqueued behind it therefore arrive when the scan is resident instead of racing its workgroups for the CUs. */
int mgr_stream_wait_next_resident(mgr_ctx* ctx, int timeout_us);
/* The same for ONE given launch: `seq` is the context's count of persistent launches (mgr_persist_stats: `launches`) read right
 * before that launch was enqueued, plus one.  Lets a stream wait for a launch that is already enqueued on another stream. */
int mgr_stream_wait_resident(mgr_ctx* ctx, unsigned seq, int timeout_us);
/* The same when the wait has to be enqueued BEFORE the launch it is for (the launch sits later in the host's order, on another
 * stream): `seq_word` is a word of page-locked host memory (mgr_host_alloc) that the caller sets to 0 before this call and that the
 * launch's call fills in (mgr_scan_launch_opts.seq_out = seq_word).  The wait kernel polls the word until it is non-zero, then waits
 * for that launch's residency as mgr_stream_wait_resident does; MGR_SEQ_NONE releases it at once.  Bounded like the others. */
int mgr_stream_wait_resident_word(mgr_ctx* ctx, const unsigned* seq_word, int timeout_us);
/* Counters of the residency waits of this context since it was created (read on the current stream, a 16-byte read-back):
 * out[0] = waits enqueued that have finished, out[1] = those that ran into their timeout (a wait whose launch never came, came too
 * late, or could not become resident while the wait held its stream: each costs its full bound - silently, which is why this exists),
 * out[2], out[3] = diagnostics of the LAST wait that expired: the launch number it was for (0: its word was never filled) and the low
 * 32 bits of the address of the word it polled (0: a wait for a known number). */
int mgr_resident_wait_stats(mgr_ctx* ctx, unsigned out[4]);
/* Persistent launches on different streams are admitted against the chip's workgroup slots; one that would not fit beside the
 * launches still in flight is ordered behind them (co-residency by construction).  Counters: launches so far, and how many of
 * them had to be serialised that way. */
int mgr_persist_stats(mgr_ctx* ctx, int* launches, int* serialised);
/* Diagnostic: out[b] = XCC (XCD) id the workgroup b of a (nblocks, threads, lds_bytes) launch ran on. */
int mgr_probe_xcc(mgr_ctx* ctx, int nblocks, int threads, int lds_bytes, int32_t* out);
/* Diagnostic: what a guest kernel of the shape of a collective (RCCL all-reduce: a few workgroups, tens of KiB of LDS, ~100 us)
 * experiences on the current stream, e.g. beside resident persistent scans: a 1-block marker launch followed by the guest
 * (nblocks x threads, lds_bytes of dynamic LDS, every block busy for ~us microseconds).  out[0], out[1] = the marker's start /
 * end, out[2 + 2b], out[3 + 2b] = start / end of guest block b, all in ticks of the 100 MHz device wall clock.  out: device,
 * 2 + 2 * nblocks int64. */
int mgr_probe_guest(mgr_ctx* ctx, int nblocks, int threads, int lds_bytes, int us, int64_t* out);
/* Diagnostic (tools/overlap_probe.py): hold the current stream for ~us microseconds on the device (bounded; 0 <= us <= 100000). */
int mgr_stream_delay(mgr_ctx* ctx, int us);
/* BPTT: dY[b,t,0:H] (row stride lddy) is dLoss/dh_t from above; Y (stride ldy) is the layer's own output as
 * written by scan_fwd WITHOUT residual (needed only through gates/cs here).  Produces dZ [B,T,4H] packed. */
int mgr_lstm_scan_bwd(mgr_ctx* ctx, const float* dY, int lddy, const float* gates, const float* cs,
                      const float* Up, float* dZ, int B, int T, int H, int reverse, void* ws, size_t ws_bytes);
/* Several BPTT recurrences (both directions of a Bidirectional layer) in ONE call; layers with a multi-CU
 * instantiation run as a single persistent launch of CU clusters exchanging dz_t each step (lstm_cluster_bwd.hip). */
typedef struct mgr_scan_bwd_job {
  const float* dY;
  const float* gates;
  const float* cs;
  const float* Up;
  float* dZ;
  int lddy, B, T, H, reverse;
  /* optional: dzmax[b * 4H + col] = the largest |dZ[b, t, col]| over t as float bits - what mgr_lstm_param_grads_dropout_ts scales
   * the rows of dZ^T by.  The multi-CU kernel keeps it in four registers of the thread that owns a (sample, unit) for all T steps
   * (free); any other kernel family is followed by a reduction pass inside the call.  NULL: not wanted. */
  unsigned* dzmax;
  /* optional: dbsum[b * 4H + col] = sum over t of dZ[b, t, col], added in step order - the bias gradient's per-sample partial sums
   * (mgr_lstm_param_grads_dropout_ts takes them in place of its own pass over dZ).  Kept by the multi-CU kernels like dzmax, in four
   * more registers; any other kernel family: the same reduction pass.  NULL: not wanted. */
  float* dbsum;
} mgr_scan_bwd_job;
size_t mgr_lstm_scan_bwd_multi_ws_bytes(int njobs, const mgr_scan_bwd_job* jobs);
int mgr_lstm_scan_bwd_multi(mgr_ctx* ctx, int njobs, const mgr_scan_bwd_job* jobs, void* ws, size_t ws_bytes);
int mgr_lstm_scan_bwd_multi_ex(mgr_ctx* ctx, int njobs, const mgr_scan_bwd_job* jobs, void* ws, size_t ws_bytes,
                               const mgr_scan_launch_opts* opts);   /* opts->form: MGR_BPTT_FORM_* */
/* Parameter gradients from dZ (all packed layouts, f32 MFMA split-K GEMMs, deterministic slab reduce):
 *   dWp[F,4H] = sum_rows (X (.) mask4)^T dZ ;  dUp[H,4H] = sum_rows hprev^T dZ ;  dbp[4H] = sum_rows dZ
 * Hs is the layer's own un-residualed output h (row stride ldh); hprev is its time-shifted view. */
size_t mgr_lstm_param_grads_ws_bytes(int B, int T, int F, int H);
int mgr_lstm_param_grads(mgr_ctx* ctx, const float* X, int ldx, const float* mask4, const float* Hs, int ldh,
                         const float* dZ, float* dWp, float* dUp, float* dbp, int B, int T, int F, int H,
                         int reverse, void* ws, size_t ws_bytes);
/* The same with Keras input dropout at rate drop_rate on X (mask4 holds the factors): from drop_rate >= 0.3 and F >= 128 on,
 * dW is computed per (gate, sample) over the rows of the kept features only and gathered in sample order (gemm.hip,
 * k_gemm_tn_sparse / k_dw_gather); dU and db as in mgr_lstm_param_grads.  Equal to it up to fp32 summation order. */
size_t mgr_lstm_param_grads_dropout_ws_bytes(int B, int T, int F, int H);
int mgr_lstm_param_grads_dropout(mgr_ctx* ctx, const float* X, int ldx, const float* mask4, float drop_rate,
                                 const float* Hs, int ldh, const float* dZ, float* dWp, float* dUp, float* dbp, int B,
                                 int T, int F, int H, int reverse, void* ws, size_t ws_bytes);
/* The dropout-aware dW from the TRANSPOSED activation copy XT[b][f][0..ldt) (mgr_transpose_bt - the copy the forward
 * projection mgr_lstm_input_proj_dropout_t was fed): the K dimension of dW is time, so both operands (XT rows of the kept
 * features; dZ transposed into the workspace by this call) are read as contiguous float4 along t.  Results bit-identical
 * to mgr_lstm_param_grads_dropout.  Only for shapes where ..._wants_transposed() says 1; ldt % 4 == 0, ldt >= T rounded
 * up to 16, the pad zero.
 * x_absmax: as for mgr_lstm_input_proj_dropout_t - with a bound on |XT| (and ldt >= T rounded up to 32) the dW product runs on the
 * f16 matrix pipe with split-f16 (hi, lo) operands and f32 accumulation (k_gemm_tn_sparse16; dZ is scaled per (sample, gate
 * column) by its own largest magnitude, so its dynamic range costs nothing); then equal to mgr_lstm_param_grads_dropout to the
 * f32 tolerance instead of bit for bit.  > 0: checked on the device, f32 MFMA kernel if violated; < 0: guaranteed by the producer of
 * XT, unchecked; 0, or tune key 15 = 1: the f32 MFMA kernel. */
int mgr_lstm_param_grads_dropout_wants_transposed(mgr_ctx* ctx, float drop_rate, int F);
size_t mgr_lstm_param_grads_dropout_t_ws_bytes(int B, int T, int F, int H, int ldt);
int mgr_lstm_param_grads_dropout_t(mgr_ctx* ctx, const float* XT, int ldt, const float* mask4, float drop_rate,
                                   const float* Hs, int ldh, const float* dZ, float* dWp, float* dUp, float* dbp, int B,
                                   int T, int F, int H, int reverse, void* ws, size_t ws_bytes, float x_absmax);
/* The same from a PRE-SPLIT transposed copy XS (the format of mgr_lstm_input_proj_dropout_ts, ldt % 32 == 0): dW on the f16 matrix pipe
 * as a loader + matrix pipeline (gemm_split.hip, k_dw_split) - dZ is transposed into the workspace as split rows scaled per (sample,
 * gate column) by that row's own largest magnitude.  mask4: entries 0 or ONE common factor (checked on the device; else dW is NaN).
 * 16 <= F <= 2048.  dU / db as in mgr_lstm_param_grads. */
size_t mgr_lstm_param_grads_dropout_ts_ws_bytes(int B, int T, int F, int H, int ldt);
int mgr_lstm_param_grads_dropout_ts(mgr_ctx* ctx, const float* XS, int ldt, const float* mask4, float drop_rate,
                                    const float* Hs, int ldh, const float* dZ, float* dWp, float* dUp, float* dbp, int B,
                                    int T, int F, int H, int reverse, void* ws, size_t ws_bytes, const unsigned* dzmax,
                                    const float* dbsum, const void* proj_ws, const float* HsT);
/* (dzmax: the row maxima of dZ if the BPTT left them - mgr_scan_bwd_job.dzmax - or NULL: this call finds them with one more pass.
 *  dbsum: the per-sample sums of dZ over time if the BPTT left them - mgr_scan_bwd_job.dbsum - then db = their sum over the samples in
 *  sample order; or NULL: db from a pass over dZ, as in mgr_lstm_param_grads.
 *  proj_ws: the workspace of the mgr_lstm_input_proj_dropout_ts call that projected with the SAME mask4 (same B, F, H), untouched since -
 *  its kept lists are used instead of being built again; or NULL.
 *  HsT: the rows h_prev in the split row format - mgr_transpose_bt_split_shift of this direction's outputs Hs with tshift = -1 (forward) /
 *  +1 (reverse), [B][H][ldt] - then dU is formed like dW, on the f16 matrix pipe against the same dZ^T rows (16 <= H); or NULL: the f32
 *  product of mgr_lstm_param_grads.) */
/* dX[b,t,0:F] (stride lddx) (+)= sum_g mask4[g] (.) (dZ_g . W_g^T); accumulate=1 adds into dX. */
int mgr_lstm_input_grad(mgr_ctx* ctx, const float* dZ, const float* Wp, const float* mask4, float* dX,
                        int lddx, int accumulate, int B, int T, int F, int H);

/* ---- K5: Dropout -> Dense -> softmax (multimodal_fusion/multimodal.py:171-179) ---------------------- */
/* P[B,T,C] = softmax((A (.) dm) . Wd + bd).  A has row stride lda.  The dropout mask dm is either the
 * explicit array dmask[B,T,D] (parity runs) or, when dmask==NULL and p>0, generated in-kernel from
 * (seed, element index) exactly as mgr_dropout_mask would; p==0 & dmask==NULL means no dropout. */
int mgr_dense_softmax_fwd(mgr_ctx* ctx, const float* A, int lda, const float* dmask, float p, uint64_t seed,
                          const float* Wd, const float* bd, float* P, int B, int T, int D, int C);
size_t mgr_dense_bwd_ws_bytes(int B, int T, int D, int C);
/* dWd[D,C], dbd[C], dA[B,T,D] (stride ldda) from dLogits[B,T,C]. */
int mgr_dense_bwd(mgr_ctx* ctx, const float* A, int lda, const float* dmask, float p, uint64_t seed,
                  const float* dLogits, const float* Wd, float* dWd, float* dbd, float* dA, int ldda, int B,
                  int T, int D, int C, void* ws, size_t ws_bytes);
/* The whole head of a training step behind ONE entry point - a host-side sequence of four launches on the context's stream
 * (k_dense_softmax_fwd*, k_ctc, k_mean, k_dense_bwd*), NOT a fused kernel: Dropout -> Dense -> softmax (P written), CTC loss +
 * gradient, Dense backward, and the mean loss if loss_mean != NULL (written between the CTC kernel and the Dense backward, so a
 * read-back of it does not wait for the backward).  Reference: multimodal_fusion/multimodal.py:171-179 (Dropout / Dense / Activation('softmax')), losses.py:4-15
 * (ctc_lambda_func) and Keras' backward pass through them.  Bit for bit the results of mgr_dense_softmax_fwd + mgr_ctc_loss_grad
 * (+ mgr_mean) + mgr_dense_bwd with the same arguments; dLogits [B,T,C] is a required scratch / output; ws >= mgr_head_ws_bytes. */
size_t mgr_head_ws_bytes(int B, int T, int D, int C, int Lmax);
int mgr_head_fwd_bwd(mgr_ctx* ctx, const float* A, int lda, const float* dmask, float p, uint64_t seed, const float* Wd, const float* bd,
                     const int32_t* labels, const int32_t* input_len, const int32_t* label_len, int B, int T, int D, int C, int Lmax,
                     int skip, int blank, float eps, float gscale, float* P, float* loss, float* loss_mean, float* dLogits, float* dWd,
                     float* dbd, float* dA, int ldda, void* ws, size_t ws_bytes);

/* ---- K6: ctc_lambda_func (multimodal_fusion/losses.py:4-15 -> K.ctc_batch_cost -> tf.nn.ctc_loss) ---- */
/* CTC on P[:, skip:, :] with y = softmax(log(P+eps)); labels int32 [B,Lmax] padded -1; blank = C-1 in the
 * reference.  loss[B] = -log p(l|x).  dLogits[B,T,C] (may be NULL) = gscale * dloss_b/d(Dense logits),
 * zero on dropped / out-of-length frames.  Edge cases: label_len 0 is accepted (the single state is the blank, as in
 * tf.nn.ctc_loss); a label sequence that does not fit its input length (tf.nn.ctc_loss raises "Not enough time for target
 * transition sequence") yields loss = +inf and a ZERO gradient for that sample, the other samples of the batch are unaffected. */
size_t mgr_ctc_ws_bytes(int B, int T, int C, int Lmax);
int mgr_ctc_loss_grad(mgr_ctx* ctx, const float* P, const int32_t* labels, const int32_t* input_len,
                      const int32_t* label_len, int B, int T, int C, int Lmax, int skip, int blank, float eps,
                      float gscale, float* loss, float* dLogits, void* ws, size_t ws_bytes);

/* ---- K7: Adam(clipvalue) + maxnorm (multimodal_fusion/multimodal.py:159-168,206-213) ---------------- */
/* g' = clip(g*gscale, +-clipvalue) (clipvalue<=0: no clip); m,v,p Keras-Adam update with step size lr_t. */
int mgr_adam_step(mgr_ctx* ctx, float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1,
                  float b2, float eps, float clipvalue, float gscale);
/* per column j of W[rows,cols]: W[:,j] *= clip(n_j,0,maxv)/(eps+n_j), n_j = ||W[:,j]||_2 */
int mgr_maxnorm_cols(mgr_ctx* ctx, float* W, int rows, int cols, float maxv, float eps);
/* Out[r, 0:cols] = A[r, 0:cols] + Bm[r, 0:cols] with independent row strides (layers.add, multimodal.py:111) */
int mgr_add2d(mgr_ctx* ctx, const float* A, int lda, const float* Bm, int ldb, float* Out, int ldo, size_t rows,
              int cols);
/* out[0] = mean(x[0:n]) */
int mgr_mean(mgr_ctx* ctx, const float* x, int n, float* out);

/* ---- K8: data parallel gradient all-reduce (not in the reference; BASELINE.json config 4) ----------- */
int mgr_comm_unique_id(uint8_t id[MGR_UNIQUE_ID_BYTES]);
int mgr_comm_init_rank(mgr_ctx* ctx, int nranks, int rank, const uint8_t id[MGR_UNIQUE_ID_BYTES], mgr_comm** out);
int mgr_allreduce_sum(mgr_comm* comm, float* dbuf, size_t n); /* in place, on the ctx's current stream */
int mgr_allreduce_max(mgr_comm* comm, float* dbuf, size_t n);
int mgr_comm_destroy(mgr_comm* comm);
/* What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank), not what the caller passed to
 * mgr_comm_init_rank: bench.py prints it, so that a multi-GPU line proves from the library's side that N ranks met.
 * The gradient all-reduce's device time is profiling family MGR_K_ALLREDUCE (events on the stream it is enqueued on). */
int mgr_comm_count(mgr_comm* comm, int* nranks_seen, int* rank_seen);

/* ---- K9: decode (multimodal_fusion/sequence_decoding.py:38-53; audio_network/sequence_decoding.py:38-53) */
/* best[b,t-skip] = argmax_c P[b,t,c] (first index on ties, numpy semantics), prob = that max. */
int mgr_frame_argmax(mgr_ctx* ctx, const float* P, int B, int T, int C, int skip, int32_t* best, float* prob);
/* CTC prefix beam search (K.ctc_decode(greedy=False, beam_width) semantics; BASELINE.json config 5).
 * out[B,T-skip] padded -1 holds the best path (after merge_repeated collapse when merge_repeated!=0). */
size_t mgr_ctc_beam_ws_bytes(int B, int T, int C, int beam);
int mgr_ctc_beam_search(mgr_ctx* ctx, const float* P, const int32_t* input_len, int B, int T, int C, int skip,
                        int blank, int beam, float eps, int merge_repeated, int32_t* out, int32_t* out_len,
                        double* logp, void* ws, size_t ws_bytes);

/* ---- skeletal feature extraction (skeletal_network/skeletal_feature_extraction.py:24-215), fp64 like the original.
 * joints[n_frames][12] = lhX lhY rhX rhY leX leY reX reY hipX hipY shcX shcY of the WHOLE frame table in file order;
 * out[n_frames][23] = lh_v rh_v le_v re_v | lh_a rh_a le_a re_a | hands_d | lh,rh,le,re _hip_d | lh,rh,le,re _shc_d |
 * lh_hip_ang rh_hip_ang lh_shc_ang rh_shc_ang lh_el_ang rh_el_ang.  Velocities / accelerations of rows 0..4 are 0. */
#define MGR_SKELETAL_JOINT_COLS 12
#define MGR_SKELETAL_FEATURE_COLS 23
int mgr_skeletal_features(mgr_ctx* ctx, const double* joints, size_t n_frames, double* out);

#ifdef __cplusplus
}
#endif
#endif /* MGR_H_ */
