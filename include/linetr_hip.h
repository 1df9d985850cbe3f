/*
 * linetr_hip.h -- C ABI of the MI355X-native (gfx950) Line-Transformer descriptor + matcher.
 *
 * This is the drop-in boundary for the hot path of yosungho/LineTR: every entry point names the
 * reference interface it replaces (paths relative to the reference checkout).  Plain pointers and
 * sizes only -- no torch/ATen types.  All `d_*` pointers are DEVICE pointers (HIP), `h_*` are HOST
 * pointers.  `stream` is a hipStream_t passed as void* (NULL = default stream).  Every function
 * returns 0 on success or a negative LINETR_E_* code; linetr_last_error() returns a message for the
 * calling thread.  Nothing here ever falls back to a CPU implementation: if no HIP device is
 * usable, linetr_create() fails.
 *
 * Call sequence for one batch of B images (one image = B=1):
 *     linetr_prefilter / linetr_pack_lines   (host, O(K) per image)  -> line records
 *     linetr_tokenize                         (device)                -> the tensors of `preprocess`
 *     linetr_forward                          (device)                -> line_desc
 *     linetr_match                            (device)                -> Dk, match indices
 * or, for throughput:  linetr_prefilter_batch -> linetr_describe (tokenise + forward fused, var-len batch) -> linetr_match;
 * streams of batches:  linetr_describe_submit / linetr_describe_join (the same call as a software pipeline over consecutive batches).
 * ABI version 5 (r06): + linetr_describe_submit, linetr_describe_join, linetr_pipeline_max_slots.
 */
#ifndef LINETR_HIP_H
#define LINETR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LINETR_ABI_VERSION 5

enum {
  LINETR_OK = 0,
  LINETR_E_ARG = -1,       /* bad argument / unsupported configuration            */
  LINETR_E_HIP = -2,       /* a HIP runtime call failed                            */
  LINETR_E_WEIGHTS = -3,   /* state_dict tensor missing / wrong size               */
  LINETR_E_ASSERT = -4,    /* reference AssertionError: token beyond the line end  */
  LINETR_E_WORKSPACE = -5, /* workspace too small                                  */
  LINETR_E_CAPACITY = -6   /* caller-provided output capacity too small            */
};

typedef struct LinetrHandle LinetrHandle;

/* Model hyper-parameters: LineTransformer.default_config, models/line_transformer.py:187-201. */
typedef struct {
  int32_t d_model;          /* descriptor_dim, must be 256                                  */
  int32_t n_heads;          /* must be 4                                                    */
  int32_t d_inner;          /* FFN width, multiple of 128 (default 1024)                    */
  int32_t n_sig_layers;     /* line-signature layers (7)                                    */
  int32_t n_desc_layers;    /* n_line_descriptive_layers; only the LAST one reaches the     */
                            /* output (models/line_transformer.py:123-125)                  */
  int32_t enc_channels[4];  /* keyline_encoder = {32,64,128,256}                            */
  int32_t norm_height;      /* constructor-time image_shape used by normalize_keylines      */
  int32_t norm_width;       /* (models/line_transformer.py:206,:238)                        */
  int32_t bn_batch_stats;   /* 0: BatchNorm(eval) folded into the convolutions (inference).  1: a TRAINING-mode handle      */
                            /* (train.py:127 -> model.train()): the convolutions stay unfolded and only                      */
                            /* linetr_forward_train may run on it                                                            */
} LinetrModelConfig;

/* One key-line after pre-filtering (float64 geometry exactly as the reference keeps it in NumPy). */
typedef struct {
  double sp[2];        /* start point (x,y), post remove_borders clip                           */
  double ep[2];        /* end point                                                            */
  double length;       /* lineLength * 2^octave (models/line_process.py:220) -- NOT geometric   */
  double angle[2];     /* (cos 2theta, sin 2theta), models/line_process.py:28-41                */
  int32_t first_sub;   /* index of this line's first sub-line inside the whole batch           */
  int32_t n_tok;       /* ceil(length / token_distance), models/line_process.py:109            */
  int32_t n_sub;       /* ceil(n_tok / max_tokens), models/line_process.py:121                 */
  int32_t image;       /* image index inside the batch                                         */
  int32_t line_local;  /* index of this key-line inside its image                              */
  int32_t first_tok;   /* index of this line's first REAL token in the batch-wide compact list  */
} LinetrLineRec;       /* 80 bytes */

/* Device outputs of the tokeniser == the tensor entries LineTransformer.preprocess returns
 * (models/line_process.py:182-193), leading batch-1 axis dropped, images concatenated.
 * K = total key-lines, N = total sub-lines, T = max_tokens, S = T+1.  All float32. */
typedef struct {
  float* klines;      /* [K,2,2]                                                    */
  float* length;      /* [K]                                                        */
  float* angles;      /* [K,2]                                                      */
  float* sublines;    /* [N,2,2]                                                    */
  float* pnt;         /* [N,T,2]   pnt_sublines                                     */
  float* mask;        /* [N,S]     mask_sublines (trailing 1-axis dropped)          */
  float* resp;        /* [N]       resp_sublines                                    */
  float* angle_sub;   /* [N,2]     angle_sublines                                   */
  float* desc;        /* [N,T,256] desc_sublines                                    */
  float* score;       /* [N,T]     score_sublines                                   */
  float* mat;         /* mat_klines2sublines (models/line_process.py:160-165): per image a [K_i,N_i] block, 1/num_sublines  */
                      /*           over a key-line's own sub-lines and 0 elsewhere, the blocks back to back (image i at float     */
                      /*           offset sum_{j<i} K_j N_j).  Optional (NULL = not written); calls of up to 8 images -- the rows   */
                      /*           are written by extra blocks of the tokeniser's own launch                                  */
  const int32_t* h_cu_klines; /* HOST prefix sums [n_images+1] of key-lines per image; needed with `mat` when n_images > 1     */
} LinetrTokens;

/* ---- lifetime ------------------------------------------------------------------------------ */

int linetr_abi_version(void);
const char* linetr_last_error(void);

/* Replaces LineTransformer.__init__ + load_state_dict (models/line_transformer.py:203-223).
 * `names[i]` are state_dict keys, `h_data[i]` host float32 arrays of
 * `numel[i]` elements; integer buffers (num_batches_tracked) may be omitted or passed as NULL.
 * BatchNorm folding, head permutation and the CLS-query constants are derived here in float64
 * and uploaded once (weights stay resident in HBM/Infinity Cache). */
int linetr_create(const LinetrModelConfig* cfg, int32_t n_tensors, const char* const* names,
                  const float* const* h_data, const int64_t* numel, int32_t device,
                  LinetrHandle** out);
void linetr_destroy(LinetrHandle* h);

/* ---- host pre-filter ------------------------------------------------------------------------ */

/* change_cv2_T_np + remove_borders + filter_by_length (models/line_process.py:203-231, :59-84,
 * :6-21) on one image.  h_lines6 = [K,6] rows (startX,startY,endX,endY,lineLength,octave).
 * h_valid_mask: NULL (== the reference's non-ndarray mask, i.e. ignored) or [height,width] float64.
 * max_keylines follows the reference's slice semantics ([:max_keylines], so -1 drops the shortest).
 * Ties in length are ordered by descending original index (== a stable ascending argsort, reversed); linetr_prefilter_tied_images
 * reports them, for hosts that want NumPy's order instead.
 * Writes up to `capacity` records (first_sub/n_tok/n_sub/image filled as by linetr_pack_lines;
 * `sub_base` / `tok_base` = number of sub-lines / real tokens of the images that precede this one in the batch) and
 * returns K' in *k_out, the number of sub-lines of this image in *n_out.  LINETR_E_ASSERT if a token distance
 * exceeds the geometric line length (the reference's AssertionError, line_process.py:44-45).
 * Survivors are written straight into h_recs: on ANY error return the contents of h_recs are unspecified (the same holds
 * for linetr_prefilter_batch). */
int linetr_prefilter(const double* h_lines6, int32_t K, int32_t height, int32_t width, int32_t border,
                     double min_length, int32_t max_keylines, const double* h_valid_mask,
                     double token_distance, int32_t max_tokens, int32_t image_index, int32_t sub_base,
                     int32_t tok_base, LinetrLineRec* h_recs, int32_t capacity, int32_t* k_out, int32_t* n_out);

/* linetr_prefilter for a whole batch in one call (images processed by up to `n_threads` host threads,
 * 0 = library default).  h_lines6 holds the [K_i,6] blocks of all images back to back, h_line_off [B+1]
 * their row offsets; h_valid_masks is NULL or an array of B (nullable) [height,width] float64 pointers.
 * Writes the records image-major into h_recs (may be pinned memory) and the prefix sums of surviving
 * key-lines / sub-lines into h_cu_k / h_cu_n [B+1]. */
int linetr_prefilter_batch(const double* h_lines6, const int32_t* h_line_off, int32_t n_images, int32_t height,
                           int32_t width, int32_t border, double min_length, int32_t max_keylines,
                           const double* const* h_valid_masks, double token_distance, int32_t max_tokens,
                           int32_t n_threads, LinetrLineRec* h_recs, int32_t capacity, int32_t* h_cu_k,
                           int32_t* h_cu_n);

/* Which images of the LAST linetr_prefilter / linetr_prefilter_batch call on the calling thread had two candidates of EQUAL
 * length in front of the length sort (models/line_process.py:15-16): there the reference's order -- np.argsort, an unstable,
 * CPU-dispatched sort -- is whatever NumPy does on the host, and a tie across the [:max_keylines] cut even changes the set.
 * Writes up to `capacity` image indices (ascending; 0 for linetr_prefilter) and returns their number.  A host that has NumPy
 * re-orders exactly those images with it and re-packs them through linetr_pack_lines (linetr_amd/engine.py prefilter,
 * tie_order="numpy"), which makes the batched path identical by index to the reference on that machine; every other image
 * has a unique order. */
int32_t linetr_prefilter_tied_images(int32_t* h_images, int32_t capacity);

/* Same record packing for lines that were already filtered/sorted by the caller (the Python shim
 * keeps NumPy's own argsort so that tie order is the reference's on the same machine).
 * h_klines [K,2,2], h_length [K], h_angles [K,2] float64. */
int linetr_pack_lines(const double* h_klines, const double* h_length, const double* h_angles, int32_t K,
                      double token_distance, int32_t max_tokens, int32_t image_index, int32_t sub_base,
                      int32_t tok_base, LinetrLineRec* h_recs, int32_t* n_out);

/* ---- device: tokenise ----------------------------------------------------------------------- */

/* Bytes of scratch linetr_tokenize needs (NHWC copy of the dense descriptor maps + index maps). */
int64_t linetr_tokenize_workspace_bytes(int32_t n_images, int32_t height, int32_t width, int32_t N);

/* line_tokenizer + sample_descriptors + score gather (models/line_process.py:100-196, :86-98).
 * d_recs: [K] records of all images (image-major, per-image order preserved); h_recs the same
 * array on the host (used only to size launches).  d_dense_desc [B,256,height/8,width/8] NCHW
 * (dense_is_nhwc = 0, the reference's 'dense_descriptor') or [B,height/8,width/8,256] (dense_is_nhwc != 0, what
 * linetr_superpoint_heads emits: no layout pass), d_dense_score [B,height,width].  The end-point clip of line_process.py:114-116 is applied, and
 * the clipped end points are what `out.klines` holds (reference quirk: it mutates through a view).
 * d_sub2line [N] int32 (key-line index of every sub-line INSIDE ITS IMAGE, non-decreasing per image)
 * is written for linetr_match.  `out.desc` may be NULL to skip descriptor sampling.  `h` may be NULL (no weights are
 * involved; the current HIP device is used).  clip_height / clip_width: the `image_shape` ARGUMENT of line_tokenizer
 * (models/line_process.py:101), which only sets the end-point clip (:115-116) and need not be the maps' shape -- the dataset
 * builder passes (640, 480) for 480 x 640 images (dataloaders/utils/util_lines.py:682,703); <= 0 = height / width. */
int linetr_tokenize(LinetrHandle* h, const LinetrLineRec* d_recs, int32_t K, int32_t N,
                    double token_distance, int32_t max_tokens, const float* d_dense_desc,
                    const float* d_dense_score, int32_t n_images, int32_t height, int32_t width,
                    int32_t clip_height, int32_t clip_width, int32_t align_corners, int32_t dense_is_nhwc,
                    LinetrTokens out, int32_t* d_sub2line, void* d_workspace, int64_t workspace_bytes, void* stream);

/* sample_descriptors (models/line_process.py:86-98) on its own, for the module-level function of the shim: n points
 * d_points [n,2] (x,y in pixels) of ONE image sampled bilinearly (zero padding) from its dense descriptor map
 * ([256,Hc,Wc], or [Hc,Wc,256] with dense_is_nhwc) and L2-normalised; d_out [n,256] row-major.  `h` may be NULL. */
int64_t linetr_sample_descriptors_workspace_bytes(int32_t Hc, int32_t Wc, int32_t dense_is_nhwc);
int linetr_sample_descriptors(LinetrHandle* h, const float* d_points, int64_t n, const float* d_dense_desc, int32_t Hc,
                              int32_t Wc, int32_t align_corners, int32_t dense_is_nhwc, float* d_out, void* d_workspace,
                              int64_t workspace_bytes, void* stream);

/* ---- device: descriptor network ------------------------------------------------------------- */

int64_t linetr_forward_workspace_bytes(const LinetrHandle* h, int32_t N, int32_t max_tokens);

/* LineTransformer.forward (models/line_transformer.py:225-249) on a var-len batch.
 * h_cu_sub [B+1]: host prefix sums of sub-lines per image (signature attention is per image).
 * Inputs are the tokeniser tensors; mask is accepted for interface fidelity and ignored, because it
 * masks query rows only and the CLS row is never masked (models/line_attention.py:16).
 * d_cu_sub: the same array already on the device, or NULL (the library then copies h_cu_sub itself).
 * d_line_desc [N,256] row-major (the shim exposes the reference's [1,256,N] as a transposed view). */
int linetr_forward(LinetrHandle* h, const LinetrTokens* tok, const int32_t* h_cu_sub, const int32_t* d_cu_sub,
                   int32_t n_images, int32_t max_tokens, float* d_line_desc, void* d_workspace,
                   int64_t workspace_bytes, void* stream);

/* ---- device: fused tokenise + describe (batched fast path) ---------------------------------------- */

/* Training-time forward (train.py:127,163-164: model.train(); pred = model(data) on a batch of B fixed-size samples): the same
 * network as linetr_forward with every BatchNorm1d of the three MLP stacks (models/line_transformer.py:9-20: 4 in
 * WordPositionalEncoder, 4 in LinePositionalEncoder, 1 per AttentionalPropagation) in TRAINING mode -- normalised with the mean and
 * biased variance of this batch (over all B*N*T token positions / B*N sub-lines), running statistics moved by `momentum` towards
 * the batch mean / UNBIASED variance.  Dropout (models/line_attention.py:11,39,84) is taken at probability 0: the caller has to
 * make sure of that (the Python surface refuses otherwise); no gradients are produced.
 * `h` must have been created with bn_batch_stats = 1.  The images of the batch are the n_images entries of h_cu_sub, as in
 * linetr_forward.  d_bn_running (in/out) and d_bn_batch (out, may be NULL) are packed per BatchNorm layer in THIS order --
 * klenc.word_position_enc.encoder.{1,4,7,10} | klenc.line_position_enc.encoder.{1,4,7,10} | selfattn.layers.l.mlp.1 (l = 0 ..) --
 * (note: the state_dict lists the line encoder first), each layer as mean[C] | var[C]; d_bn_batch
 * receives the batch mean | biased variance.  linetr_bn_stats_floats() = the length of both arrays. */
int64_t linetr_bn_stats_floats(const LinetrHandle* h);
int64_t linetr_forward_train_workspace_bytes(const LinetrHandle* h, int32_t N, int32_t max_tokens);
int linetr_forward_train(LinetrHandle* h, const LinetrTokens* tok, const int32_t* h_cu_sub, const int32_t* d_cu_sub,
                         int32_t n_images, int32_t max_tokens, float momentum, float* d_bn_running, float* d_bn_batch,
                         float* d_line_desc, void* d_workspace, int64_t workspace_bytes, void* stream);

int64_t linetr_describe_workspace_bytes(const LinetrHandle* h, int32_t n_images, int32_t height, int32_t width,
                                        int32_t N, int64_t n_real_tokens);

/* preprocess + forward of LineTransformer (models/line_transformer.py:251-275 + :225-249) for a batch in one
 * call, without materialising the [N,T,256] token descriptors: the word-position MLP runs on the REAL tokens
 * only (n_real_tokens = sum of n_tok over d_recs, plus one shared zero-padding token per image -- every padded
 * slot of an image holds the same coordinate (0,0), hence the same descriptor, score and key/value), and the
 * CLS-row attention pooling samples the NHWC descriptor map on the fly, counting the padding token with its
 * multiplicity.  Results equal linetr_tokenize + linetr_forward up to fp32 rounding.
 * dense_is_nhwc != 0: d_dense_desc is already [B, H/8, W/8, 256] (a producer that emits channel-last, e.g. a fused
 * SuperPoint descriptor head) and the NCHW->NHWC copy is skipped.
 * Small tokeniser outputs (klines, length, angles, sublines, resp, angle_sub) are written when their pointers in
 * `out` are non-NULL; pnt / mask / score / desc are written only if non-NULL (dense [N,T,...] layout). */
int linetr_describe(LinetrHandle* h, const LinetrLineRec* d_recs, int32_t K, int32_t N, int64_t n_real_tokens,
                    const int32_t* h_cu_sub, const int32_t* d_cu_sub, int32_t n_images, double token_distance,
                    int32_t max_tokens, const float* d_dense_desc, const float* d_dense_score, int32_t height,
                    int32_t width, int32_t align_corners, int32_t dense_is_nhwc, LinetrTokens out,
                    int32_t* d_sub2line, float* d_line_desc, void* d_workspace, int64_t workspace_bytes,
                    void* stream);

/* linetr_describe as a software pipeline over CONSECUTIVE batches ("batch/varlen plumbing + stream
 * pipelining"; the reference is serial per pair, models/matching.py:34-60).  A batch is cut into stages at fixed points of the network
 * -- e.g. front = layout pass, tokeniser, positional-encoder MLPs, CLS pooling, descriptive-layer tail (HBM-bound for half of its
 * time); back = the line-signature network (MFMA-bound, launches whose tile rounds leave CUs empty) -- and stage k of every batch is
 * queued on the k-th of a set of library-owned streams, so stage k of batch i + 1 runs under stage k + 1 of batch i.  Every GEMM still
 * sees the FULL batch (nothing is split), and the results are bit-identical to linetr_describe's: the same kernels on the same data.
 *   linetr_describe_submit  same arguments as linetr_describe + slot and n_slots: the caller keeps n_slots (2 .. linetr_pipeline_max_slots())
 *                           batches in flight and gives batch i the slot i mod n_slots.  The work starts behind everything already
 *                           queued on `stream` (the upload of d_recs, the producer of the dense maps) and is NOT joined back: `stream`
 *                           does not wait for it.  Every slot needs its OWN d_workspace and output buffers, alive until the slot is
 *                           joined.  Host run-ahead is bounded: a submit first waits (on the host) for the batch previously submitted
 *                           to the same slot.
 *   linetr_describe_join    `stream` waits for the batch last submitted to `slot`; its outputs may be read on `stream` afterwards.
 * A caller pipelines by  submit(i, i mod n);  join(i - n + 1, (i - n + 1) mod n);  -- n - 1 batches of latency for the overlap.
 * Like every entry point that takes a handle, the pair is not re-entrant: one submitting thread per handle.  A failed submit leaves
 * the library's streams idle and the slot free (nothing half-queued survives it). */
int linetr_describe_submit(LinetrHandle* h, const LinetrLineRec* d_recs, int32_t K, int32_t N, int64_t n_real_tokens,
                           const int32_t* h_cu_sub, const int32_t* d_cu_sub, int32_t n_images, double token_distance,
                           int32_t max_tokens, const float* d_dense_desc, const float* d_dense_score, int32_t height,
                           int32_t width, int32_t align_corners, int32_t dense_is_nhwc, LinetrTokens out,
                           int32_t* d_sub2line, float* d_line_desc, void* d_workspace, int64_t workspace_bytes,
                           int32_t slot, int32_t n_slots, void* stream);
int32_t linetr_pipeline_max_slots(void);
int linetr_describe_join(LinetrHandle* h, int32_t slot, void* stream);

/* ---- device: matcher ------------------------------------------------------------------------ */

int64_t linetr_match_workspace_bytes(int32_t n_pairs, int64_t sum_n0n1, int64_t sum_k0k1, int64_t sum_k);

/* get_dist_matrix + subline2keyline + nn_matcher_distmat (models/line_process.py:198-201,
 * models/line_transformer.py:277-282, models/nn_matcher.py:3-31) for P pairs in one launch set.
 * Pair p uses descriptors d_desc0[h_off_n0[p] .. +n0) (rows of 256) and key-line maps
 * d_sub2line0 (LOCAL key-line index per sub-line, non-decreasing), same for side 1.
 * h_dims [P,4] = (n0,k0,n1,k1); offsets are host int64 arrays [P].
 * Outputs: d_dk at h_off_dk[p] holds Dk [k0,k1] float32; d_match01 at h_off_k0[p] holds, per
 * key-line of image 0, the matched key-line of image 1 or -1 (the non-zero of the reference's 0/1
 * matrix; first-index argmin, strict `<`, optional mutual check). */
int linetr_match(LinetrHandle* h, int32_t n_pairs, const int32_t* h_dims, const float* d_desc0,
                 const int64_t* h_off_n0, const int32_t* d_sub2line0, const float* d_desc1,
                 const int64_t* h_off_n1, const int32_t* d_sub2line1, float nn_thresh, int32_t mutual,
                 float* d_dk, const int64_t* h_off_dk, int32_t* d_match01, const int64_t* h_off_k0,
                 void* d_workspace, int64_t workspace_bytes, void* stream);

/* linetr_match with separate offsets for the key-line maps: pair p reads its descriptors at row h_off_n0[p] of d_desc0
 * and its sub-line -> key-line map at element h_off_s0[p] of d_sub2line0 (same for side 1; NULL offsets = h_off_n*).
 * This is the form global matching uses after the multi-GPU all-gather, where descriptors and maps of all ranks sit
 * at different places of ONE gathered buffer (linetr_amd/parallel.py; no reference counterpart, BASELINE.json cfg4). */
int linetr_match_gathered(LinetrHandle* h, int32_t n_pairs, const int32_t* h_dims, const float* d_desc0,
                          const int64_t* h_off_n0, const int32_t* d_sub2line0, const int64_t* h_off_s0,
                          const float* d_desc1, const int64_t* h_off_n1, const int32_t* d_sub2line1,
                          const int64_t* h_off_s1, float nn_thresh, int32_t mutual, float* d_dk,
                          const int64_t* h_off_dk, int32_t* d_match01, const int64_t* h_off_k0, void* d_workspace,
                          int64_t workspace_bytes, void* stream);

/* nn_matcher_distmat (models/nn_matcher.py:3-31) on a distance matrix that already lives on the device:
 * d_dist [n0,n1] float32 -> d_match01 [n0] (index into side 1 or -1).  `h` may be NULL for the three
 * matcher entry points (they need no weights); the current HIP device is used then.
 * Contract: distances are float32 and NaN-free (the reference's cosine distances are clipped to [0,4]); a NaN entry
 * is skipped by the argmin here, whereas np.argmin would return it.  All three matcher entry points are fully
 * asynchronous on `stream` (host tables are staged in a library-owned pinned ring). */
int64_t linetr_match_distmat_workspace_bytes(int32_t n0, int32_t n1);
int linetr_match_distmat(LinetrHandle* h, const float* d_dist, int32_t n0, int32_t n1, float nn_thresh,
                         int32_t mutual, int32_t* d_match01, void* d_workspace, int64_t workspace_bytes,
                         void* stream);

/* nn_matcher_distmat (models/nn_matcher.py:3-31) on a float64 device matrix d_dist [n0,n1], compared in float64 the way NumPy
 * compares a float64 matrix (the reference's own caller passes float32: linetr_match_distmat).  nn_thresh is a double.  `h` may
 * be NULL; asynchronous on `stream`. */
int64_t linetr_match_distmat_f64_workspace_bytes(int32_t n0, int32_t n1);
int linetr_match_distmat_f64(LinetrHandle* h, const double* d_dist, int32_t n0, int32_t n1, double nn_thresh, int32_t mutual,
                             int32_t* d_match01, void* d_workspace, int64_t workspace_bytes, void* stream);

/* LineTransformer.subline2keyline (models/line_transformer.py:277-282) alone: Dk [k0,k1] = A0 D A1^T for a sub-line distance
 * matrix d_dist [n0,n1] that already lives on the device, the two mat_klines2sublines given as the sub-line -> key-line maps
 * linetr_tokenize writes (non-decreasing; rows of A are 1/num_sublines).  `h` may be NULL.  Asynchronous on `stream`. */
int64_t linetr_pool_distmat_workspace_bytes(int32_t k0, int32_t k1);
int linetr_pool_distmat(LinetrHandle* h, const float* d_dist, int32_t n0, int32_t n1, const int32_t* d_sub2line0, int32_t k0,
                        const int32_t* d_sub2line1, int32_t k1, float* d_dk, void* d_workspace, int64_t workspace_bytes,
                        void* stream);

/* LineTransformer.subline2keyline (models/line_transformer.py:277-282) on the two mat_klines2sublines MATRICES, the way the
 * reference's call sites pass them (models/matching.py:80: data['mat_klines2sublines0'][0], [K,N] float32 on the device): a matrix
 * of the form line_tokenizer writes (models/line_process.py:163-167: one non-zero per column, key-lines in order, rows of
 * float32(1 / num_sublines)) is reduced to its sub-line -> key-line map on the device and pooled like linetr_pool_distmat;
 * any other matrix is multiplied out as given, Dk = (A0 D) A1^T in fp32.  The decision is made on the device: asynchronous on
 * `stream`, no host synchronisation.  `h` may be NULL. */
int64_t linetr_pool_distmat_dense_workspace_bytes(int32_t k0, int32_t n0, int32_t k1, int32_t n1);
int linetr_pool_distmat_dense(LinetrHandle* h, const float* d_dist, int32_t n0, int32_t n1, const float* d_A0, int32_t k0,
                              const float* d_A1, int32_t k1, float* d_dk, void* d_workspace, int64_t workspace_bytes,
                              void* stream);

/* nn_matcher (models/nn_matcher.py:33-42): point-descriptor variant, desc given [256,n] column-major
 * like SuperPoint's `descriptors` -- section 8(f) "next" row, same kernels. */
int linetr_match_points(LinetrHandle* h, const float* d_desc0_cn, int32_t n0, const float* d_desc1_cn,
                        int32_t n1, float nn_thresh, int32_t mutual, float* d_dist, int32_t* d_match01,
                        void* d_workspace, int64_t workspace_bytes, void* stream);

/* The matching tail of Matching.forward (models/matching.py:67-84) in ONE call: the point matcher on the two [256,n] SuperPoint
 * descriptor sets (nn_matcher, models/nn_matcher.py:33-42), the line matcher on the two images' line descriptors (get_dist_matrix +
 * subline2keyline + nn_matcher_distmat) and the device -> host copies of the four results into ONE caller-provided PINNED host block,
 * everything asynchronous on `stream` (the caller waits for the stream / an event once).  Either branch is skipped when its sizes are
 * 0.  linetr_pair_tail_output_bytes returns the size of the block and the byte offsets of its four segments:
 *   h_offsets[0] point distances [np0][np1] float32, [1] point match01 [np0] int32, [2] Dk [k0][k1] float32, [3] line match01 [k0] int32. */
int64_t linetr_pair_tail_workspace_bytes(int32_t np0, int32_t np1, int32_t n0, int32_t k0, int32_t n1, int32_t k1);
int64_t linetr_pair_tail_output_bytes(int32_t np0, int32_t np1, int32_t k0, int32_t k1, int64_t* h_offsets);
int linetr_pair_tail(LinetrHandle* h, const float* d_pdesc0_cn, int32_t np0, const float* d_pdesc1_cn, int32_t np1, float nn_thresh_points,
                     const float* d_ldesc0, int32_t n0, const int32_t* d_sub2line0, int32_t k0, const float* d_ldesc1, int32_t n1,
                     const int32_t* d_sub2line1, int32_t k1, float nn_thresh_lines, int32_t mutual, void* h_pinned_out,
                     int64_t pinned_bytes, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- dense-map producer (section 8(f) "next" row 2) ------------------------------------------------- */

/* Post-processing of SuperPoint's two heads, fused with the layout change the tokeniser needs; replaces
 *   models/superpoint.py:161-167  scores = softmax(convPb(.), 1)[:, :-1]; permute/reshape to [B, 8Hc, 8Wc]
 *   models/superpoint.py:190-193  dense_descriptor = F.normalize(convDb(.), p=2, dim=1)
 * d_score_logits [B,65,Hc,Wc] and d_desc_raw [B,256,Hc,Wc] are the raw convPb / convDb outputs (float32, NCHW,
 * contiguous).  Outputs (any may be NULL): d_dense_score [B,8Hc,8Wc]; d_dense_desc_nhwc [B,Hc,Wc,256] -- pass it
 * to linetr_describe with dense_is_nhwc = 1 and the NCHW->NHWC pass disappears; d_dense_desc_nchw [B,256,Hc,Wc],
 * the reference's 'dense_descriptor' layout.  `h` may be NULL (no weights involved; current HIP device). */
int linetr_superpoint_heads(LinetrHandle* h, const float* d_score_logits, const float* d_desc_raw, int32_t B,
                            int32_t Hc, int32_t Wc, float* d_dense_score, float* d_dense_desc_nhwc,
                            float* d_dense_desc_nchw, void* stream);

/* ---- arithmetic mode of the dense contractions ----------------------------------------------------- */

/* All Linear/Conv1d(k=1) contractions run on one of three MFMA paths (fp32 in, fp32 accumulate, fp32 out):
 *   LINETR_PREC_F32     v_mfma_f32_32x32x2_f32, exact fp32 products                (157 TF ceiling)
 *   LINETR_PREC_BF16X6  operands split into 3 bf16 planes, 6 cross products: fp32-faithful (~2^-23 per
 *                       product, same class as fp32 summation-order noise)          (417 TF-equivalent)
 *   LINETR_PREC_BF16X3  2 planes, 3 cross products: ~1e-5 relative per product      (833 TF-equivalent)
 *   LINETR_PREC_F16X3   2 fp16 planes (22 significand bits), 3 cross products: ~2^-22 per product, i.e. fp32-class
 *                       (measured 1.1e-6 on the descriptors vs 4.4e-7 for BF16X6), but every GEMM operand must
 *                       stay below 65504 in magnitude (fp16 range)                  (833 TF-equivalent)
 * Default: LINETR_PREC_BF16X6, overridable with the environment variable LINETR_PRECISION=f32|bf16x6|bf16x3|f16x3
 * at linetr_create time.  The signature attention's two contractions (S^T = K Q^T and O^T = V^T P^T) follow the mode as well: exact
 * fp32 MFMA in LINETR_PREC_F32, the six-product bf16 split (fp32-faithful) in the three split modes; its softmax, the token sampler,
 * the pooling, LayerNorm / L2 normalisation and every other non-GEMM stage are fp32 in all modes.  The matcher's distance products
 * are exact fp32 MFMA in all modes (an argmin must not inherit a split's error). */
enum { LINETR_PREC_F32 = 0, LINETR_PREC_BF16X3 = 1, LINETR_PREC_BF16X6 = 2, LINETR_PREC_F16X3 = 3 };
int linetr_set_precision(LinetrHandle* h, int32_t mode);
int linetr_get_precision(const LinetrHandle* h);

/* ---- diagnostics ---------------------------------------------------------------------------- */

/* Runs layers 1-3 of a positional encoder (MLP of models/line_transformer.py:9-20, BN folded, ReLU) alone, for the
 * unit tests: which = 0 WordPositionalEncoder (line_transformer.py:52-73; d_in0 = token points [rows,2] in pixels,
 * d_in1 = token scores [rows], d_in2 unused), which = 1 LinePositionalEncoder (:40-50; d_in0 = sub-lines [rows,2,2] in
 * pixels, d_in1 = resp [rows], d_in2 = angles [rows,2]).  d_out [rows,128] = activations after the third ReLU.
 * Coordinates are normalised with the handle's image_shape as in normalize_keylines (:22-38). */
int linetr_debug_posenc(LinetrHandle* h, int32_t which, const float* d_in0, const float* d_in1, const float* d_in2,
                           int64_t rows, float* d_out, void* stream);

/* Runs the library's fp32-MFMA GEMM  Y[M,N] = act(A[M,K] W[N,K]^T + bias) (+ R)  on device buffers; used
 * by the unit tests (vs a plain PyTorch fp32 reference) and by the kernel micro-benchmarks.  act: 0 none,
 * 1 ReLU, 2 erf-GELU, 3 max(2-2x,0).  d_bias / d_residual may be NULL.  N % 64 == 0, K % 32 == 0.
 * lda / ldy are the row strides of A and Y (and of the residual) in floats, multiples of 4.
 * cache_weights != 0 keeps the split-bf16 copy of d_W (keyed by pointer) for later calls, so repeated
 * calls time the GEMM kernel alone; the caller then promises not to change the contents of d_W. */
int linetr_debug_gemm(LinetrHandle* h, const float* d_A, int32_t lda, const float* d_W, const float* d_bias,
                      const float* d_residual, float* d_Y, int32_t ldy, int32_t M, int32_t N, int32_t K,
                      int32_t act, int32_t cache_weights, void* stream);

#ifdef LINETR_EXPERIMENTS
/* ---- split-tile ("ST") operands (csrc/lt_st_image.h; the GEMM on them: experiments/csrc/lt_gemm_st.h): experiments build only --------------------
 * (liblinetr_hip_experiments.so, `python -m linetr_amd.build --experiments`; measured and not shipped)
 * The signature network keeps its activations in HBM pre-split into three bf16 planes, in 512-byte chunks that are the
 * LDS image of a 16-row x 16-column block (K-step-major), so that a GEMM's K steps travel by LDS-DMA.  These three entry points expose
 * the format and the kernel to the unit tests and micro-benchmarks (no reference counterpart: the reference's
 * nn.Conv1d(k=1) calls, models/line_transformer.py:157-183, are what the GEMM replaces).
 * linetr_st_bytes: size of the ST image of a [rows][K] matrix (K % 16 == 0; rows are padded to a multiple of 128),
 * -1 on bad arguments.
 * linetr_debug_to_st / from_st: fp32 rows (row stride ld floats, multiple of 4) <-> ST image; exact both ways.
 * linetr_debug_gemm_st: Y = act([A1 | A2] W^T + bias) (+ R), all operands ST images (d_A2 / d_bias / d_R may be NULL),
 * result as an ST image (d_Yst) or, when d_Yst is NULL, as fp32 rows d_Y (row stride ldy).  N % 256 == 0,
 * (K1 + K2) % 32 == 0. */
int64_t linetr_st_bytes(int64_t rows, int32_t K);
int linetr_debug_to_st(LinetrHandle* h, const float* d_X, int32_t ld, int32_t rows, int32_t K, void* d_st, void* stream);
int linetr_debug_from_st(LinetrHandle* h, const void* d_st, int32_t rows, int32_t K, float* d_X, int32_t ld, void* stream);
int linetr_debug_gemm_st(LinetrHandle* h, const void* d_A1, int32_t K1, const void* d_A2, int32_t K2, const void* d_W,
                         const float* d_bias, const void* d_R, void* d_Yst, float* d_Y, int32_t ldy, int32_t M, int32_t N,
                         int32_t act, void* stream);
/* Diagnostics of the single-pair persistent signature network (csrc/lt_pairnet.h): the next launches write wall-clock stamps
 * (100 MHz ticks) per block and stage into d_buf [blocks][4 n_sig_layers + 1][8] = {first unit picked up, its producers seen,
 * last body done, last unit published, four probes inside the unit}; the caller zeroes the buffer.  NULL switches the stamps off.  tools/pairnet_timeline.py */
int linetr_debug_pairnet_stamps(LinetrHandle* h, unsigned long long* d_buf);
#endif  /* LINETR_EXPERIMENTS */

/* ---- multi-GPU: the descriptor all-gather as a C entry point --------------------------------------
 * ONE all-gather of the ranks' packed slabs (header | sub-line -> key-line map | line_desc rows; layout in
 * linetr_amd/parallel.py, which the Python surface uses through torch.distributed) over an RCCL communicator the CALLER
 * owns: d_out [world][slab_bytes] receives every rank's d_slab.  `nccl_comm` is an ncclComm_t.  The library does not link
 * against RCCL: ncclAllGather is resolved at the first call from the RCCL already loaded in the process (PyTorch's or
 * /opt/rocm's librccl.so); LINETR_E_HIP if none is loaded.  No reference counterpart (BASELINE.json cfg4); asynchronous
 * on `stream`.  linetr_match_gathered then matches straight out of d_out. */
int linetr_allgather_desc(void* nccl_comm, const void* d_slab, void* d_out, int64_t slab_bytes, void* stream);

/* The ncclAllGather to call.  A process can hold two RCCL copies (PyTorch's bundled one and /opt/rocm's); the function must
 * come from the copy that created `nccl_comm`, so a caller that knows which one that is hands its ncclAllGather in here
 * (dlsym on its own library handle).  Without this call linetr_allgather_desc takes the first ncclAllGather the process-wide
 * symbol resolution finds -- correct when only one RCCL is loaded.  NULL resets to that default. */
int linetr_set_allgather_fn(void* nccl_allgather_fn);

/* One rank's slab for that all-gather in ONE launch (the layout linetr_amd/parallel.py documents and GatheredSet /
 * linetr_match_gathered read): float32 [hr + mr + rows_cap][256] = int32 header {n_images, sub-lines per image[cap],
 * key-lines per image[cap]} | int32 sub2line[rows_cap] | line_desc rows, with hr = ceil((1 + 2 cap) / 256) and
 * mr = ceil(rows_cap / 256).  The counts are taken from the DEVICE prefix sums d_cu_n / d_cu_k [n_images + 1] (what
 * linetr_describe leaves there), so nothing is copied from the host; d_cu_k / d_sub2line may be NULL.  zero_tail != 0 also
 * clears the descriptor rows N .. rows_cap.  LINETR_E_CAPACITY if the batch does not fit.  No reference counterpart. */
int linetr_pack_slab(const float* d_line_desc, int32_t N, const int32_t* d_cu_n, const int32_t* d_cu_k, int32_t n_images,
                     const int32_t* d_sub2line, int32_t n_images_cap, int32_t rows_cap, int32_t zero_tail, float* d_slab,
                     void* stream);

/* ---- instrumentation ------------------------------------------------------------------------ */

/* Per-kernel-class HIP-event timing.  linetr_set_profiling(h,1) clears the accumulators and makes
 * every subsequent kernel launch of this handle be bracketed by hipEventRecord on its stream;
 * linetr_get_profile synchronises the recorded events and returns, per kernel class, the number of
 * launches, the summed duration and the ALGORITHMIC flops / bytes (SURVEY.md section 8d accounting)
 * those launches performed.  bench.py derives its roofline object from this. */
typedef struct {
  const char* name;   /* static string, e.g. "gemm_f32_128x128" */
  int32_t calls;
  float ms;           /* summed over calls */
  double flops;       /* summed algorithmic flops */
  double bytes;       /* summed compulsory HBM bytes */
} LinetrProfileEntry;
int linetr_set_profiling(LinetrHandle* h, int32_t on);
int linetr_get_profile(LinetrHandle* h, LinetrProfileEntry* out, int32_t max_entries, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* LINETR_HIP_H */
