/* star_nugget.h -- C ABI of libstar_nugget.so: STAR's nugget-compression hot path on MI355X (gfx950).
 *
 * The reference (steventan0110/STAR) has no FFI / plugin interface: its boundary is a set of Python
 * method signatures.  Each entry point below names the reference lines it replaces; the Python
 * mirror of those signatures lives in star_amd/nugget.py and binds these symbols through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory
 *     (workspaces, streams and events included): the library never allocates or frees device memory;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no host
 *     synchronisation and no readback, and is re-entrant and thread-safe: the only process-wide state is
 *     a set of idempotent "kernel attribute set on device n" bits (atomics) and the test-only switches of
 *     star_test_override(); the one documented exception is star_scorer_fwd_static (its scratch array);
 *   - return value: 0 on success, a negative STAR_ERR_* otherwise; star_last_error() then returns a
 *     thread-local message.  Nothing throws across the ABI;
 *   - tensors are dense row-major unless strides are given (strides are in ELEMENTS);
 *   - bf16 is the storage type of activations (uint16 bit pattern of the upper half of an fp32),
 *     fp32 the type of scores, biases, log-sum-exps and parameter gradients.
 */
#ifndef STAR_NUGGET_H
#define STAR_NUGGET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAR_OK 0
#define STAR_ERR_ARG (-1)      /* bad shape / null pointer / unsupported size */
#define STAR_ERR_LAUNCH (-2)   /* HIP reported a launch error */
#define STAR_ERR_WORKSPACE (-3) /* workspace too small */

#define STAR_DT_BF16 0
#define STAR_DT_F32 1

int star_version(void);
const char* star_last_error(void);

/* TEST-ONLY: force a kernel variant that the library would not pick on its own for the given shape, so that tests
 * and A/B timing runs reach every kernel.  Process-wide atomics, 0 = the library's own choice (the default); not
 * meant for product callers.  Returns the previous value (>= 0), or STAR_ERR_ARG for an unknown switch. */
#define STAR_OVR_XA_BWD 0  /* attention backward: 1 = split dQ + dK/dV kernels, 2 = the fused kernel whenever legal,
                            * 3 = (head dim 128) the experimental 4-wave form of the fused kernel, else as 2 */
#define STAR_OVR_XA_DQ 1   /* 1 = the general (register-staged) dQ kernel */
#define STAR_OVR_XA_DKV 2  /* 1 = the general dK/dV kernel */
#define STAR_OVR_XA_NG 3   /* key groups of the register-staged attention forward / general dQ: 1 or 2 */
#define STAR_OVR_XA_FWD 4  /* attention forward: 1 = the register-staged kernel instead of the LDS-DMA ring */
#define STAR_OVR_SCORER_NT 5 /* scorer backward dW1 GEMM: 1 = NT on transposed copies (0 / 2: TN on the row-major
                              operands wherever a workgroup's row ids fit its LDS) */
#define STAR_OVR_SCORER_FWD 6 /* scorer forward GEMM: 1 = the experimental 4-wave register-staged kernel (d >= 256) */
#define STAR_OVR_COUNT 7
int star_test_override(int which, int value);

/* ---------------------------------------------------------------- a2 + a3: top-k anchor select
 * Replaces W2VASRModel.apply_nugget_compression (src/models/w2v_asr.py:208-244; copy at
 * w2v_simul.py:222-260) fused with add_nugget_type_embedding (w2v_asr.py:316-323).
 *   scores      [B,T] fp32
 *   attn_mask   [B,T] uint8 (bool) or NULL; pads are mask==0.  When NULL, lengths[b] gives the
 *               number of valid (leading) frames.
 *   lengths     [B] int32 or NULL (exactly one of attn_mask / lengths must be given)
 *   n_nugget    [B] int32 explicit nugget counts, or NULL to use trunc(n_tok*ratio + 0.99) clamped
 *               to n_tok with fp32 arithmetic (ratio = (float)(1/compress_rate))
 *   K           width of the outputs = max_b k_b (computed by the caller from the lengths)
 * Outputs (any may be NULL except idx_out):
 *   idx_out [B,K] int64 ascending kept indices then ascending pad-slot indices,
 *   mask_out [B,K] uint8 (j < k_b), scores_gather [B,K] fp32 = scores[b, idx_out],
 *   type_table [B,T] int64 0/1, positions [B,K] int64 (= idx_out with pad slots set to -1),
 *   flat_rows [B*K] int32 = b*T + idx_out[b,j]: the frame ids of the selected slots in the flattened [B*T]
 *   frame axis, i.e. the `rows` list star_scorer_bwd takes (needs B*T < 2^31).
 * Tie rule: equal keys rank by lower frame index first. */
int star_topk_select(const float* scores, const uint8_t* attn_mask, const int32_t* lengths,
                     const int32_t* n_nugget, float ratio, int B, int T, int K,
                     int64_t* idx_out, uint8_t* mask_out, float* scores_gather,
                     int64_t* type_table, int64_t* positions, int32_t* flat_rows, void* stream);

/* Stand-alone add_nugget_type_embedding (w2v_asr.py:316-323; w2v_simul.py:273-280) for callers that already
 * hold indices and mask: type_table [B,T] int64 = 1 at kept frames, positions [B,K] int64 = idx with masked
 * slots set to -1 (positions may be NULL). */
int star_nugget_type_table(const int64_t* idx, const uint8_t* nugget_mask, int B, int T, int K,
                           int64_t* type_table, int64_t* positions, void* stream);

/* ---------------------------------------------------------------- a4: row gather / scatter-add
 * Replaces semantic_repr[arange(B)[:,None], nugget_indices] (w2v_asr.py:566-567,
 * w2v_simul.py:446-447) and, with row_bytes == 4, the score gather of w2v_simul.py:427.
 * idx [B,K] int64; an index < 0 addresses frame T + idx (so -1 is the last frame), as in Python.  An index
 * outside [-T, T) (an IndexError in torch) is CLAMPED to the nearest frame, identically in forward and backward.
 * row_bytes must be a multiple of 2.  src [B,T,row], dst [B,K,row]. */
int star_gather_rows_fwd(const void* src, const int64_t* idx, void* dst, int B, int T, int K,
                         int64_t row_bytes, void* stream);
/* Backward of the gather: dsrc[b,t,:] = sum_{j: idx[b,j]==t} ddst[b,j,:]  (index_put accumulate),
 * written for EVERY (b,t) (zero where no j matches), summed in ascending j with fp32 accumulation.
 * dtype is STAR_DT_BF16 or STAR_DT_F32; d = elements per row. */
int star_gather_rows_bwd(const void* ddst, const int64_t* idx, void* dsrc, int B, int T, int K,
                         int d, int dtype, void* stream);

/* ---------------------------------------------------------------- f-3 (second half): gather at the output of the
 * semantic encoder.  The reference normalises all T frames with the encoder's final LayerNorm (ln_f,
 * my_transformer.py:945) and then keeps the nugget rows (w2v_asr.py:566-567, w2v_simul.py:446-447); LayerNorm is
 * row-wise, so the rows are gathered first and only they are normalised:
 *   out[b,j,:] = LayerNorm(src[b, idx[b,j], :]) * gamma + beta        (idx as in star_gather_rows_fwd: -1 = last frame)
 *   src [B,T,d] and out [B,K,d] in dtype (STAR_DT_BF16 / STAR_DT_F32), gamma / beta [d] fp32, statistics in fp32
 *   mean, rstd [B,K] fp32: outputs of the forward, inputs of the backward.  d % 8 == 0, d <= 4096 (backward: 1536).
 * Backward: drows [B,K,d] (dtype) = LayerNorm backward of every gathered row; pass it to star_gather_rows_bwd to get
 * d src (duplicates summed, every row written once).  dgamma, dbeta [d] fp32 (fixed summation order). */
int star_gather_layernorm_fwd(const void* src, int dtype, const int64_t* idx, const float* gamma,
                              const float* beta, float eps, int B, int T, int K, int d, void* out, float* mean,
                              float* rstd, void* stream);
size_t star_gather_layernorm_bwd_workspace(int B, int K, int d);
int star_gather_layernorm_bwd(const void* src, int dtype, const int64_t* idx, const float* gamma,
                              const float* mean, const float* rstd, const void* dout, int B, int T, int K,
                              int d, void* drows, float* dgamma, float* dbeta, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- a5: CIF quantity loss
 * Replaces w2v_simul.py:419-424.  alphas = sigmoid(scores)*mask; diff_b = sum_t alphas/thr - L_b;
 * loss = mean_b diff_b^2.  attn_mask/lengths as in star_topk_select.  target_lengths [B] int32.
 * Outputs: alphas [B,T] fp32, diff [B] fp32, loss [1] fp32. */
int star_cif_quantity_fwd(const float* scores, const uint8_t* attn_mask, const int32_t* lengths,
                          const int32_t* target_lengths, float threshold, int B, int T,
                          float* alphas, float* diff, float* loss, void* stream);
/* dscores[b,t] (+)= gloss * 2*diff_b/(B*thr) * a*(1-a) with a = alphas[b,t]  (0 at pads).
 * accumulate != 0 adds into dscores instead of overwriting. gloss [1] fp32 on device. */
int star_cif_quantity_bwd(const float* alphas, const float* diff, const float* gloss,
                          float threshold, int B, int T, int accumulate, float* dscores, void* stream);

/* ---------------------------------------------------------------- a6: CIF resize
 * Replaces W2VSimulModel.resize (w2v_simul.py:504-535, noise = 0): rescale each row to sum L_b,
 * then up to max_rounds (reference: 10) halve-and-add-mean fix-up rounds.  Row sums accumulate in
 * fp64 and are rounded once to fp32.  alphas [B,T] fp32 -> out [B,T] fp32 (may alias).  T <= 16384.
 * fixups_out [B] int32 or NULL: the number of fix-ups applied to each row (what the backward needs). */
int star_cif_resize(const float* alphas, const int32_t* target_lengths, float threshold,
                    int max_rounds, int B, int T, float* out, int32_t* fixups_out, void* stream);
/* Backward of resize: the gradient that lm_cif_training sends through it into the scorer (w2v_simul.py:558,
 * w2v_asr.py:679; the nugget path detaches its output, w2v_simul.py:426).  alphas = the forward's INPUT,
 * fixups = the forward's fixups_out (NULL = no fix-ups), dout [B,T] fp32 -> dalphas [B,T] fp32.  The mask
 * (row != 0) and its count are constants, as in the reference's autograd graph (w2v_simul.py:529-532). */
int star_cif_resize_bwd(const float* alphas, const int32_t* target_lengths, const int32_t* fixups,
                        const float* dout, int B, int T, float* dalphas, void* stream);

/* ---------------------------------------------------------------- a7: cumulative fire scan
 * Replaces apply_threshold_yield_nugget + get_valid_mask_indices (w2v_simul.py:376-403,196-220).
 * Exact serial fp32 recurrence per utterance over all T frames:
 *   cur = fl(cur + a_t*mask_t); fire = cur >= 1 || t == n_b-1; cur = fl(cur-1) if fire.
 *   alphas [B,T] fp32; attn_mask [B,T] uint8 or NULL (then lengths [B] int32 defines a prefix mask; a
 *          length > T makes every frame valid and switches the forced last-frame fire off for that row)
 *   cap    width of the index outputs (>= max fire count; fires beyond cap are dropped and
 *          counted in counts, so the caller can detect overflow)
 *   scores optional [B,T] fp32: scores_gather[b,j] = scores[b, idx] (idx -1 -> frame T-1)
 * Outputs: idx_out [B,cap] int64 ascending fire positions, -1 padded; mask_out [B,cap] uint8;
 *   counts [B] int32; max_count [1] int32 (max over b); scores_gather [B,cap] fp32 or NULL. */
int star_cif_fire_scan(const float* alphas, const uint8_t* attn_mask, const int32_t* lengths,
                       const float* scores, int B, int T, int cap, int64_t* idx_out,
                       uint8_t* mask_out, int32_t* counts, int32_t* max_count,
                       float* scores_gather, void* stream);
/* ---------------------------------------------------------------- a5 + a6 + a7 + a3 in one launch
 * The segmenter of the simultaneous path as the reference chains it (w2v_simul.py:419-431): alphas = sigmoid(scores)
 * * mask and the quantity loss, resize(alphas, target_lengths), apply_threshold_yield_nugget on the resized weights,
 * nugget_scores = scores[b, idx] and add_nugget_type_embedding's table -- one workgroup per utterance keeps the row
 * in LDS through all stages.  Outputs are bit-identical to star_cif_quantity_fwd -> star_cif_resize ->
 * star_cif_fire_scan (with scores) -> star_nugget_type_table.  Prefix masks only (lengths [B] int32, unclamped as in
 * star_cif_fire_scan).  type_table [B,T] int64 and positions [B,cap] int64 may be NULL; max_count may be NULL.
 * Gradients: star_cif_quantity_bwd for the loss, star_gather_rows_bwd (d = 1) for scores_gather. */
int star_cif_segment(const float* scores, const int32_t* lengths, const int32_t* target_lengths,
                     float threshold, int max_rounds, int B, int T, int cap, float* alphas, float* diff,
                     float* loss, float* resized, int32_t* fixups_out, int64_t* idx_out, uint8_t* mask_out,
                     int32_t* counts, int32_t* max_count, float* scores_gather, int64_t* type_table,
                     int64_t* positions, void* stream);

/* ---------------------------------------------------------------- f4: streaming fire decision
 * The same recurrence fed one chunk of frames at a time (online fire decision; the
 * reference runs apply_threshold_yield_nugget, w2v_simul.py:376-403, on whole utterances only, so this entry point
 * has no call site there: its contract is "any chunking of a row gives the fires of star_cif_fire_scan on the whole
 * row, bit for bit").  The fp32 integrator and the number of frames consumed are carried per utterance:
 *   alpha_chunk [B,Tc] fp32    the next Tc frames of every utterance (frames at or beyond the utterance's length
 *                              count as masked: weight 0)
 *   lengths     [B] int32      utterance lengths in ABSOLUTE frames; the frame lengths[b]-1 always fires.  While an
 *                              utterance's end is not known yet pass INT32_MAX.  The forced fire is only emitted when
 *                              the length is known IN the call whose chunk holds frame lengths[b]-1: a length that
 *                              arrives later is not acted upon here (star_cif_stream_push has the late flush)
 *   score_chunk optional [B,Tc] fp32 scorer logits of the chunk: scores_gather[b,j] = score at the j-th fire
 *   state_cur   [B] fp32  in/out  integrator carry (0 before the first chunk)
 *   state_pos   [B] int32 in/out  frames consumed so far (0 before the first chunk; += Tc per call)
 * Outputs: idx_out [B,cap] int64 ABSOLUTE positions of this chunk's fires, ascending, -1 padded; counts [B] int32
 * fires in this chunk (fires beyond cap are dropped but counted); scores_gather [B,cap] fp32 or NULL. */
int star_cif_fire_step(const float* alpha_chunk, const int32_t* lengths, const float* score_chunk, int B,
                       int Tc, float* state_cur, int32_t* state_pos, int cap, int64_t* idx_out,
                       int32_t* counts, float* scores_gather, void* stream);
/* The streaming segmenter's whole per-chunk tail in ONE launch (no call site in the
 * reference, which segments whole utterances, w2v_simul.py:419-447): alphas = sigmoid(score) * mask (:420-421, the
 * bits of star_cif_quantity_fwd), the fire recurrence of star_cif_fire_step, and the append of every fired frame's
 * semantic row (:446-447), scorer logit (:427, the attention bias) and absolute position to the utterance's nugget
 * memory at counts[b].  Any chunking gives the memory of the offline kernels bit for bit.
 *   score_chunk [B,Tc] fp32   scorer logits of the chunk      s_chunk [B,Tc,d] (dtype) semantic states of the chunk
 *   lengths     [B] int32     utterance lengths in absolute frames, INT32_MAX while unknown.  The end of an utterance
 *                             must become known no later than the call AFTER the one whose chunk ends with its last
 *                             frame (lengths[b] == frames consumed): the forced fire of that frame is then emitted
 *                             from the saved row (a "late flush").  Frames pushed as valid beyond an end that is
 *                             announced later cannot be taken back.  Tc == 0: the flush alone (end of a stream).
 *   state_cur [B] fp32, state_pos [B] int32, state_last [B] int32 (position of the newest fire, start at -1): in/out
 *   mem [B,kmax,d] (dtype), mem_scores [B,kmax] fp32, mem_pos [B,kmax] int64, counts [B] int32 (in/out, <= kmax),
 *   dropped [B] int32 (in/out): fires that found no slot because the memory was full -- poll it, a non-zero entry
 *   means the memory of that utterance is truncated;  last_row [B,d] (dtype), last_score [B] fp32: scratch state.
 *   idx_out [B,cap] int64 / cnt_out [B] int32 (both optional): this chunk's fires as star_cif_fire_step reports them
 *   (the late-flush fire is not listed).  d * sizeof(dtype) must be a multiple of 16, buffers 16-byte aligned,
 *   Tc <= 8192. */
int star_cif_stream_push(const float* score_chunk, const void* s_chunk, int dtype, const int32_t* lengths,
                         int B, int Tc, int d, float* state_cur, int32_t* state_pos, int32_t* state_last,
                         int kmax, void* mem, float* mem_scores, int64_t* mem_pos, int32_t* counts,
                         int32_t* dropped, void* last_row, float* last_score, int cap, int64_t* idx_out,
                         int32_t* cnt_out, void* stream);
/* block_attn_mask[b,i,j] = 1.0f iff prev_fire(i) < j <= i and frame i valid (w2v_simul.py:382-402),
 * built on demand from the fire positions.  idx [B,cap] int64 (-1 padded), counts [B] int32,
 * out [B,T,T] fp32. */
int star_cif_block_mask(const int64_t* idx, const int32_t* counts, const uint8_t* attn_mask,
                        const int32_t* lengths, int B, int T, int cap, float* out, void* stream);

/* ---------------------------------------------------------------- 8f-2: CIF weighted aggregation
 * Replaces the frame loop + per-utterance index_select of lm_cif_training (w2v_simul.py:558-606; same loop at
 * w2v_asr.py:681-722):  dc = 1 - I; I += a_t; fire = I >= threshold; I -= 1 where fired; cur = fire ? dc : a_t;
 * rem = a_t - cur; frame += cur * s_t; the frame is emitted at a fire and restarts as rem * s_t.  No forced fire
 * at the last valid frame; the partial segment after the last fire is dropped (as in the reference).
 *
 * star_cif_weights: the scalar recurrence.  alphas [B,T] fp32 (already masked / resized).
 *   Outputs: weights [B,T,4] fp32, 16-byte aligned, one record per frame {cur, rem, bits of the int32
 *   (number of fires before t) << 1 | fired(t), 0}; idx_out [B,cap] int64 ascending fire positions, -1 padded;
 *   mask_out [B,cap] uint8 or NULL; counts [B] int32 (fires beyond cap are dropped but counted);
 *   max_count [1] int32 or NULL.  T <= 16384. */
int star_cif_weights(const float* alphas, float threshold, int B, int T, int cap, float* weights,
                     int64_t* idx_out, uint8_t* mask_out, int32_t* counts, int32_t* max_count,
                     void* stream);
/* out[b,j,:] = rem[f_{j-1}] s[f_{j-1}] + sum_{t=f_{j-1}+1..f_j} cur[t] s[t] for j < min(counts[b], cap), else 0;
 * fp32 accumulation in frame order, product and sum rounded separately.  s [B,T,d], out [B,cap,d], dtype
 * STAR_DT_BF16 or STAR_DT_F32 (both tensors); weights / idx / counts as written by star_cif_weights. */
int star_cif_aggregate_fwd(const void* s, int dtype, const float* weights, const int64_t* idx,
                           const int32_t* counts, int B, int T, int cap, int d, void* out, void* stream);
/* Backward: ds [B,T,d] (every frame written), dalphas [B,T] fp32 (fire decisions are constants; the gradient of
 * a firing frame's weight reaches every earlier alpha).  workspace: 2*B*T floats. */
int star_cif_aggregate_bwd(const void* dout, const void* s, int dtype, const float* weights,
                           const int32_t* counts, int B, int T, int cap, int d, void* ds,
                           float* dalphas, float* workspace, void* stream);

/* ---------------------------------------------------------------- 8f-3: nugget type embedding
 * Replaces  inputs_embed = encoder_out.last_hidden_state + semantic_encoder.type_embed(type_ind_table)
 * (w2v_asr.py:383,400,456,560; w2v_simul.py:432; type_embed = nn.Embedding(2, d), my_transformer.py:761).
 * x, out [N,d] (N = B*T rows), table [N] int64 (0 / non-zero = plain frame / nugget), emb [2,d]; x, emb, out share
 * dtype (STAR_DT_BF16 / STAR_DT_F32); the sum is formed in fp32 and rounded once (what torch's add does). */
int star_type_embed_add(const void* x, const int64_t* table, const void* emb, int dtype, int64_t N, int d,
                        void* out, void* stream);
/* Its backward for the embedding table: demb[c][:] = sum of dout rows whose table entry is c, fp32 [2,d], in a fixed
 * summation order (d x = dout needs no kernel).  workspace: star_type_embed_grad_workspace(N, d, dtype) bytes. */
size_t star_type_embed_grad_workspace(int64_t N, int d, int dtype);
int star_type_embed_grad(const void* dout, const int64_t* table, int dtype, int64_t N, int d, float* demb,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- a1: scorer MLP
 * Replaces nugget_model = Sequential(Linear(d,d), ReLU, Linear(d,1)) + squeeze(-1)
 * (w2v_asr.py:113-117,553; w2v_simul.py:144-148,410-411).
 *   x   [N,d] bf16 frames (N = B*T)       w1 [d,d] bf16 in nn.Linear layout [out,in]
 *   b1  [d] fp32   w2 [d] fp32   b2 [1] fp32 (device)      scores [N] fp32
 * d must be a multiple of 128.  The hidden activation H is never written to memory; what the backward needs of it,
 * the ReLU sign bits (1 bit per element: 6 MB at N = 48,000, d = 1024), goes to relu_bits when that is non-NULL:
 *   relu_bits  star_scorer_relu_bits_bytes(N,d) bytes, 16-byte aligned, opaque to the caller (the sign bits in the
 *              forward GEMM's accumulator order: one 8-KiB blob per 256 x 256 output tile); only star_scorer_bwd
 *              reads it.  NULL (inference) skips it.
 * workspace: star_scorer_fwd_workspace(N,d) bytes (per-column-tile partial scores). */
size_t star_scorer_fwd_workspace(int64_t N, int d);
size_t star_scorer_relu_bits_bytes(int64_t N, int d);
int star_scorer_fwd(const void* x, const void* w1, const float* b1, const float* w2,
                    const float* b2, int64_t N, int d, float* scores, void* relu_bits, void* workspace,
                    size_t workspace_bytes, void* stream);
/* Ragged form of star_scorer_fwd for a padded batch x [B*T, d] whose utterance b has lengths[b] valid frames (a
 * prefix of its T rows): the reference scores every padded frame too (w2v_asr.py:553, w2v_simul.py:411) and masks them
 * afterwards (alphas * mask, top-k on masked scores), so a pad frame's score is never observable; here it is 0 and its
 * row tile is not computed when the whole 256-row tile is padding.
 *   lengths    [B] int32 (device)   valid frames per utterance
 *   live_tiles [n_live] int32 (device), ascending: the 256-row tiles (tile i = rows 256 i ..) that hold at least one
 *              valid frame -- host code derives it from the lengths (star_amd.ops.ValidFrames); a tile that is not
 *              listed gets scores 0 and unspecified relu_bits, so listing too few tiles loses frames without an error
 * Everything else as star_scorer_fwd (relu_bits stays indexed by frame, so star_scorer_bwd takes it unchanged; pass
 * it only row ids of valid frames). */
int star_scorer_fwd_ragged(const void* x, const void* w1, const float* b1, const float* w2, const float* b2,
                           const int32_t* lengths, int B, int T, int d, const int32_t* live_tiles, int n_live,
                           float* scores, void* relu_bits, void* workspace, size_t workspace_bytes,
                           void* stream);
/* Compact ragged form: the GEMM runs on the valid frames only (gathered through `rows`), so no row of a computed tile
 * is padding: at MuST-C-shaped lengths (median 300 of T = 1500 frames) the 256-row live tiles of the form above are
 * 60 % full.  Same observable results (pad frames score 0; valid frames as star_scorer_fwd up to the summation order
 * inside the MFMA tiles, which does not depend on the row).  No counterpart in the reference (it scores every padded
 * frame, w2v_asr.py:553, w2v_simul.py:411).
 *   lengths [B] int32, offsets [B] int32 = exclusive prefix sum of lengths, rows [n_valid] int32 = the frame ids
 *   b*T + t of the valid frames in ascending order (so frame (b,t) is row offsets[b] + t) -- all on the device;
 *   star_amd.ops.ValidFrames builds the three from the lengths.
 *   relu_bits: as star_scorer_fwd, but indexed by ROW of the compacted list: hand star_scorer_bwd_mapped the row of
 *   every listed frame (bit_rows).  workspace: star_scorer_fwd_workspace(n_valid, d) bytes. */
int star_scorer_fwd_compact(const void* x, const void* w1, const float* b1, const float* w2, const float* b2,
                            const int32_t* lengths, const int32_t* offsets, const int32_t* rows, int64_t n_valid,
                            int B, int T, int d, float* scores, void* relu_bits, void* workspace,
                            size_t workspace_bytes, void* stream);
/* star_scorer_fwd without a workspace argument and without any allocation.  NO counterpart in the reference: it is
 * for callers that cannot allocate device memory around the call (code under HIP-graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  The partial scores go to a file-scope __device__ array of
 * STAR_SCORER_STATIC_SCRATCH_BYTES inside the library (one copy per GPU, looked up once per device with
 * hipGetSymbolAddress at the first call on that device -- make that first call outside a stream capture).
 * Limit: star_scorer_fwd_workspace(N, d) <= STAR_SCORER_STATIC_SCRATCH_BYTES, i.e. N <= 262,144 frames at d = 1024;
 * larger inputs return STAR_ERR_WORKSPACE.  Every call on a GPU shares that GPU's array: calls that overlap in time
 * on one GPU (two streams, two threads) race -- serialise them or use star_scorer_fwd.  Same kernels and the same
 * bits as star_scorer_fwd.  Device rule: the scratch pointer is the one of the CURRENT device, and `stream` (when
 * non-NULL) must belong to that device -- a stream of another device is refused with STAR_ERR_ARG (each GPU holds its
 * own copy of the array at its own address). */
#define STAR_SCORER_STATIC_SCRATCH_BYTES (4u << 20)
int star_scorer_fwd_static(const void* x, const void* w1, const float* b1, const float* w2, const float* b2,
                           int64_t N, int d, float* scores, void* relu_bits, void* stream);
/* Backward for the R rows listed in rows[] (frame ids into x; the frames whose dscore is non-zero; no duplicates).
 * There is no recompute of H: with G' = ds * relu'(H) (from relu_bits) and U = G'^T X_sel,
 *   dW1 = diag(w2) U,  db1 = w2 * colsum(G'),  dw2[n] = <W1[n], U[n]> + b1[n] colsum(G')[n],  dX = G' (diag(w2) W1).
 *   dscores [N] fp32 (full), rows [R] int32, relu_bits as written by star_scorer_fwd for the same x / w1 / b1,
 *   w1 [d,d] bf16 ([out,in], as in the forward), w1ts = bf16 of (diag(w2) W1)^T ([in,out]; star_cast_transpose_bf16
 *   with t_scale = w2 writes it), needed only for dx.
 * Outputs: dw1 [d,d] fp32, db1 [d] fp32, dw2 [d] fp32, db2 [1] fp32 (all overwritten),
 *   dx [N,d] bf16 or NULL: rows listed in rows[] are written; all other rows are zero-filled by the call
 *   unless dx_zeroed != 0 says the caller has done so already.
 * dW1 GEMM: a TN kernel reads G' and the listed rows of x in place; beyond 114688 rows (d <= 1024: 16 K splits) an NT
 *   kernel runs on transposed copies (G'^T from the G' pass, X_sel^T from a gather-transpose pass) -- same results up
 *   to fp32 rounding.
 * aux_stream (may be NULL): a second stream of the same device.  The TN form does not use it (everything runs on
 *   `stream`: the zero-fill of dx rides in the G' launch, the reductions in the dX GEMM's).  NT form, when given: X_sel^T,
 *   the zero-fill of dx and the dX GEMM run on it beside the G' pass, the dW1 GEMM and the reductions; the call forks
 *   from and joins back into `stream`, so the caller still orders everything on `stream` alone (HIP-graph capture of
 *   `stream` captures both branches).  dx must be 16-byte aligned.
 * aux_events_host: HOST array of four hipEvent_t created by the caller (hipEventDisableTiming suffices), required
 *   when aux_stream is given: the fork / join events of this call.  One set per (stream, aux_stream) pair: calls that
 *   may overlap in time (other threads, other stream pairs) bring their own sets, so the library holds no event.
 * workspace: star_scorer_bwd_workspace(R,d) bytes. */
size_t star_scorer_bwd_workspace(int64_t R, int d);
int star_scorer_bwd(const void* x, const void* w1, const void* w1ts, const float* b1,
                    const float* w2, const void* relu_bits, const float* dscores, const int32_t* rows, int64_t R,
                    int64_t N, int d, float* dw1, float* db1, float* dw2, float* db2, void* dx,
                    int dx_zeroed, void* workspace, size_t workspace_bytes, void* stream,
                    void* aux_stream, void* const* aux_events_host);
/* star_scorer_bwd for relu_bits written by star_scorer_fwd_compact: bit_rows [R] int32 (device) = the forward's row
 * (offsets[b] + t) of every frame listed in rows[]; NULL = the frame ids themselves (star_scorer_bwd). */
int star_scorer_bwd_mapped(const void* x, const void* w1, const void* w1ts, const float* b1, const float* w2,
                           const void* relu_bits, const float* dscores, const int32_t* rows, const int32_t* bit_rows,
                           int64_t R, int64_t N, int d, float* dw1, float* db1, float* dw2, float* db2, void* dx,
                           int dx_zeroed, void* workspace, size_t workspace_bytes, void* stream, void* aux_stream,
                           void* const* aux_events_host);
/* fp32 [rows,cols] -> bf16 copy (dst, may be NULL) and bf16 transpose (dst_t, may be NULL): the per-step weight cast
 * of W1.  t_scale [rows] fp32 or NULL: dst_t[c][r] = bf16(src[r][c] * t_scale[r]) (dst is never scaled). */
int star_cast_transpose_bf16(const float* src, int rows, int cols, void* dst, void* dst_t, const float* t_scale,
                             void* stream);

/* ---------------------------------------------------------------- a9 / 8f-1: the cross-attention projections
 * HF Conv1D projections around the core (c_attn / q_attn / c_proj, my_transformer.py:500-501,526: y = x W + b with W
 * [in, out]).  star_gemm_nt_bf16: out[M,N] = a[M,K] . b[N,K]^T + bias[N] -- bf16 operands (b is the weight in
 * [out, in] layout: star_cast_transpose_bf16 writes it from the Conv1D weight), fp32 accumulate, bias fp32 or NULL,
 * one rounding to bf16.  N and K multiples of 64; leading dimensions in elements, multiples of 8; 16-byte aligned.
 * Meant for wide projections (the c_attn of ALL decoder layers on the shared encoder memory: N = nl * 2d); the
 * 256 x 256 tile leaves the chip part-filled below ~256 output tiles. */
int star_gemm_nt_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const float* bias, int64_t M, int N,
                      int K, void* out, int64_t ldo, void* stream);
/* dst_t[(i * cols + c)][r] = bf16(src_i[r][c]), i < n <= 16: the Conv1D weights of n layers (each [rows = in,
 * cols = out] contiguous, fp32 or bf16 by `dtype`), transposed and laid side by side in one launch -- the [n * out, in]
 * B operand of star_gemm_nt_bf16 for the all-layer c_attn (my_transformer.py:500-501 on the memory shared by every
 * block, :917-928), rebuilt once per optimizer step.  src_list: HOST array of device pointers; rows, cols multiples
 * of 64. */
int star_transpose_list_bf16(const void* const* src_list, int n, int dtype, int rows, int cols, void* dst_t,
                             void* stream);
/* Weight gradients of nbatch projections of one shape in one launch (the dW = x^T g of HF Conv1D's backward, which
 * autograd forms per layer for my_transformer.py:500-501,526): out[i * ksplit + s][m][n] = sum over the s-th of ksplit
 * equal ranges of r of a_i[r][m] * b_i[r][n], fp32, i < nbatch.  a_i [R, M] (leading dimension lda) = the layer's
 * input activations, b_i [R, N] (ldb) = its output gradient, both bf16 and row-major by r as they lie -- no transposed
 * copies.  a_list / b_list: HOST arrays of nbatch device pointers.  nbatch * ksplit <= 16; R a multiple of 64 with at
 * most 7168 rows per range; M, N >= 128, multiples of 8; 16-byte aligned operands.  The caller adds the ksplit
 * partial products of each i.  (One such product at M = N = 1024 has 16 output tiles of 256 x 256: batching the
 * layers is what fills the chip.) */
int star_gemm_tn_batched_bf16(const void* const* a_list, const void* const* b_list, int nbatch, int ksplit, int64_t R,
                              int M, int N, int64_t lda, int64_t ldb, float* out, void* stream);
/* Bias gradient of a projection: out[n] = sum_m x[m][n], x [M,N] (dtype, leading dimension ld), out [N] fp32, in a
 * fixed summation order (bitwise reproducible).  N % 8 == 0.  workspace: star_colsum_workspace(M, N) bytes. */
size_t star_colsum_workspace(int64_t M, int N);
int star_colsum(const void* x, int dtype, int64_t M, int N, int64_t ld, float* out, void* workspace,
                size_t workspace_bytes, void* stream);
/* The same for nbatch <= 16 matrices of one shape in one pair of launches (the bias gradients that go with
 * star_gemm_tn_batched_bf16): out [nbatch][N], x_list a HOST array of device pointers, workspace nbatch *
 * star_colsum_workspace(M, N) bytes; every out[i] is bit-identical to star_colsum of x_list[i]. */
int star_colsum_batched(const void* const* x_list, int nbatch, int dtype, int64_t M, int N, int64_t ld, float* out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- a8 (+a11): score-biased attention core
 * Replaces MyGPT2Attention._attn for cross-attention (src/models/my_transformer.py:535-603):
 *   P = softmax(Q K^T * scale + ILK + bias[b,j]),  O = P V
 *   q [B,h,L,hd], k/v [B,h,Kn,hd], o like q: bf16 with element strides (sb, sh, sr), hd contiguous
 *   bias [B,bias_len] fp32 or NULL: scorer logits (+ the additive encoder pad mask, if the caller
 *        folded it in); keys j >= bias_len get bias 0 (prefix-bias variant, my_transformer.py:114-125)
 *   key_len [B] int32 or NULL: keys j >= key_len[b] are masked (P = 0 exactly)
 *   ilk_diag < 0: no infinite-lookback mask; otherwise key j is visible to query i iff j <= i + ilk_diag
 *        (the reference uses 2, my_transformer.py:566)
 *   lse [B,h,L] fp32: log-sum-exp of the masked, biased logits (saved for backward)
 *   p_out optional [B,h,L,Kn] fp32 attention weights (output_attentions), NULL to skip
 * hd in {64, 96, 128}.
 *
 * Attention-probability dropout (self.attn_dropout, my_transformer.py:595; GPT2Config.attn_pdrop, 0.1 in the
 * reference's default decoder config, w2v_asr.py:134-145):  dropout_p in [0, 1).  With dropout_p > 0
 *   O = (P o M) V / (1 - p_eff),  M[b,h,i,j] = keep(b,h,i,j) in {0,1},
 * LSE is that of the undropped softmax, p_out (if requested) holds the dropped, rescaled weights (what the
 * reference returns as attn_weights), and star_xattn_bwd regenerates the same mask from the same (seed, offset).
 * The mask is a pure function of (seed, offset, b, h, i, j) -- a counter-based generator, specified here so that
 * a caller (or a test) can reproduce it:
 *   mix(x)   = x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16      (uint32)
 *   k1       = mix(lo32(seed) ^ mix(hi32(seed) ^ mix(lo32(offset) ^ mix(hi32(offset) + 0x9e3779b9 * (b*h_count + h + 1)))))
 *   k2       = mix(k1 ^ 0x85ebca6b)
 *   x        = k1 + i * Kn + j;  x ^= x >> 16; x *= 0x7feb352d; x ^= k2; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16
 *   keep     = x >= thr,  thr = min(floor(dropout_p * 2^32), 2^32 - 1),  p_eff = thr / 2^32
 * rng_state (optional DEVICE pointer to two uint64 {seed, offset}): when non-NULL the kernels read seed / offset
 * from it at run time instead of the two fields, so a step captured in a HIP graph can draw a fresh mask at every
 * replay (the caller advances the offset with its own stream-ordered write between forward calls).
 * dropout_p == 0 runs exactly the kernels (and produces exactly the bits) of a build without dropout. */
typedef struct {
    int B, h, L, Kn, hd;
    int64_t q_sb, q_sh, q_sr;
    int64_t k_sb, k_sh, k_sr;
    int64_t v_sb, v_sh, v_sr;
    int64_t o_sb, o_sh, o_sr;
    int bias_len;
    int ilk_diag;
    float scale;
    float dropout_p;
    uint64_t seed;
    uint64_t offset;
    const uint64_t* rng_state;
} star_xattn_shape;

int star_xattn_fwd(const void* q, const void* k, const void* v, const float* bias,
                   const int32_t* key_len, const star_xattn_shape* s, void* o, float* lse,
                   float* p_out, void* stream);
/* Backward.  do_/o share o's strides; dq uses q's strides, dk/dv use k's/v's strides.
 *   delta [B,h,L] fp32 scratch (row-sum of dO*O, written by the call)
 *   dbias_part [B,h,Kn] fp32 scratch; dbias [B,bias_len] fp32 = sum over heads and queries of dS,
 *   overwritten when accumulate == 0, added to otherwise (the reference feeds the same
 *   scorer_logits to every decoder layer, my_transformer.py:917-928). dbias may be NULL. */
int star_xattn_bwd(const void* q, const void* k, const void* v, const void* o, const void* do_,
                   const float* lse, const float* bias, const int32_t* key_len,
                   const star_xattn_shape* s, void* dq, void* dk, void* dv, float* delta,
                   float* dbias_part, float* dbias, int accumulate, void* stream);
/* Bias gradient of several decoder layers at once: dbias[b,j] = sum_l sum_h parts[l][b,h,j] in the fixed order
 * (layer, head).  parts_host: HOST array of nl (<= 32) device pointers to the dbias_part [B,h,Kn] buffers the
 * per-layer star_xattn_bwd calls filled (called with dbias == NULL).  Replaces the per-layer head reduction plus
 * autograd's nl-1 additions of the reference's shared scorer_logits gradient (my_transformer.py:917-928, 582-589). */
int star_xattn_dbias_sum(const float* const* parts_host, int nl, int B, int h, int Kn, int bias_len,
                         float* dbias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STAR_NUGGET_H */
