/*
 * sehip.h -- C ABI of libsehip.so: the MI355X (gfx950) implementation of the cosine-embedding
 * training + retrieval hot path of cvjena/semantic-embeddings.
 *
 * The reference is pure Python (Keras/TF + NumPy) and has no FFI; every entry point below names
 * the reference Python interface it replaces (paths relative to the reference checkout) so a
 * maintainer can bind it with ctypes -- INTEGRATION.md shows the stubs.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless marked "host";
 *   - row-major, innermost dimension contiguous, explicit leading dimensions in ELEMENTS;
 *   - the caller owns every buffer; scratch memory is sized by the *_workspace_bytes() queries
 *     and passed in -- the library never allocates device memory;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it.  Exceptions, each marked at its
 *     declaration: se_rank_rows_init and se_rank_rows_check synchronise the stream (they hand a verdict to the host), and
 *     se_rank_rows synchronises ONCE per process and device when se_rank_rows_init was not called first (see there);
 *   - return value: SE_OK (0) or a negative SE_ERR_* code; se_last_error() (thread-local text)
 *     explains the last failure; no exceptions; re-entrant, safe from any host thread.  Process-wide state is limited to
 *     read-mostly caches filled at first use: device properties (CU count, kernel occupancies), the plan table of
 *     se_retrieve_topk, and -- the only one that influences which kernel runs -- the per-device verdict of the ranking's
 *     capability probe / self-test (atomics; see se_rank_rows_init).  Three environment variables are read by the product build,
 *     all of them by the ranking only: SE_RANK_SAFE=1 (never use the hardware-ordered kernels), SE_RANK_CHECK=1 (audit every row
 *     of every se_rank_rows call: synchronises every call), SE_RANK_VERBOSE=1 (probe / self-test verdicts on stderr).  Results
 *     never depend on them; no other switch exists in libsehip.so (libsehip_tuning.so is the build with tuning switches);
 *   - float32 arithmetic on the bit-exact paths follows the "canonical arithmetic" of
 *     DESIGN.md section 3 (sequential fp32 FMA chain over k, NumPy pairwise row sums,
 *     ascending (distance, index) order, NaN last, -0 == +0).
 */
#ifndef SEHIP_H
#define SEHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE_OK 0
#define SE_ERR_INVALID (-1)     /* bad argument (shape, null pointer, unsupported k ...) */
#define SE_ERR_HIP (-2)         /* a HIP runtime call / kernel launch failed              */
#define SE_ERR_UNSUPPORTED (-3) /* valid request that this build does not implement        */
#define SE_ERR_WORKSPACE (-4)   /* workspace too small                                     */

/* element type of the feature tensor handed to the loss kernels */
#define SE_DTYPE_F32 0
#define SE_DTYPE_BF16 1

/* distance / similarity flavours of the all-pairs kernel */
#define SE_METRIC_COSINE 0 /* out = -(a . b)              evaluate_retrieval.py:59  */
#define SE_METRIC_EUCLID 1 /* out = (|a|^2 + |b|^2) - 2ab evaluate_retrieval.py:61-62 */
#define SE_METRIC_DOT 2    /* out = a . b                 utils.py:90 (K.dot(y_pred, centroids)) */

typedef void *se_stream_t; /* hipStream_t */

int se_version(void);
const char *se_last_error(void);
/* host: name of the GPU architecture the embedded code objects were built for ("gfx950") */
const char *se_build_arch(void);

/*
 * Phase timing -- a measuring aid (bench.py's per-leg rooflines), not part of any result.  se_phase_timing(1): from now on the
 * multi-kernel entry points (se_retrieve_topk) record a HIP event on their stream behind each of their phases ("convert", "sample",
 * "threshold", "filter", "refine", "fallback") and count, per query, the candidates of the filter pass and the entries the
 * refinement recomputed exactly; se_phase_timing(0) (the default state) switches it off.  One process-wide switch; calls made while
 * it is on must not run concurrently.
 * se_phase_timing_read: waits for the last recorded event and returns the number of phases written to names_host / ms_host (HOST
 * arrays of `cap` entries, either may be NULL; a phase that ran several times appears several times, in order); counters_host (HOST,
 * 5 int64 or NULL) receives the statistics words of the last se_retrieve_topk call -- [1] queries redone exactly, [2] entries
 * recomputed with the exact chain, [3] candidates listed, [4] queries -- or -1 each; the workspace of that call must still be alive.
 * The recording restarts empty afterwards.  Negative return value: an SE_ERR_* code.
 */
int se_phase_timing(int on);
int se_phase_timing_read(const char **names_host, float *ms_host, int cap, int64_t *counters_host);

/* ------------------------------------------------------------------------------------------
 * Training side
 * ------------------------------------------------------------------------------------------ */

/*
 * Fused  l2norm  +  embedding gather  +  cosine loss  (forward).
 * Replaces: utils.l2norm (utils.py:125-127) wrapped as the model's last layer
 *           (learn_image_embeddings.py:127-128), transform_inputs' host gather embedding[y]
 *           (learn_image_embeddings.py:48-50), utils.inv_correlation (utils.py:44-46) and the
 *           Keras batch mean.
 *   x        [B, D] features (f32 or bf16, ldx elements between rows)
 *   labels   [B] int64 class indices in [0, C)
 *   emb      [C, D] f32 class embeddings (lde)
 *   xhat     [B, D] f32 out: x * rsqrt(max(sum x^2, 1e-12))        (may be NULL)
 *   inv_norm [B] f32 out: rsqrt(max(sum x^2, 1e-12))               (may be NULL)
 *   loss_i   [B] f32 out: 1 - sum_d emb[labels[i], d] * xhat[i, d]
 *   loss_mean[1] f32 out: mean_i loss_i, fixed summation order     (may be NULL)
 */
int se_cosine_loss_fwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels,
                       const float *emb, int64_t lde, int64_t B, int64_t D, int64_t C,
                       float *xhat, int64_t ldxhat, float *inv_norm, float *loss_i,
                       float *loss_mean, se_stream_t stream);

/*
 * Backward of the above (what TF autodiff derives from utils.py:44-46,125-127):
 *   g_i = -grad_loss_i[i] * emb[labels[i]]
 *   dx_i = (g_i - xhat_i (xhat_i . g_i)) * inv_norm_i      where sum x^2 >= 1e-12
 *   dx_i = g_i * inv_norm_i                                 where the max() clamp is active
 *   x [B,D] is re-read (f32/bf16); dx is written in dx_dtype (f32/bf16).
 *   grad_loss_i [B] f32 (upstream gradient per sample; 1/B for the plain batch mean);
 *   if grad_loss_i is NULL, grad_scale is used for every row.
 */
int se_cosine_loss_bwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels,
                       const float *emb, int64_t lde, const float *grad_loss_i, float grad_scale,
                       int64_t B, int64_t D, int64_t C, void *dx, int dx_dtype, int64_t lddx,
                       se_stream_t stream);

/*
 * Squared-distance loss of `--loss mse` against the gathered class embedding, and its backward.
 * Replaces: utils.squared_distance (utils.py:34-36) as the training loss (learn_image_embeddings.py:160-163) on
 *           y_true = embedding[y] (transform_inputs, learn_image_embeddings.py:48-50), and the utils.mean_distance metric
 *           (utils.py:39-41) of the same compile() call.
 *   loss_i [B] f32 = sum_d (x_d - emb[y]_d)^2;  dist_i [B] f32 = sqrt(loss_i) (may be NULL);  loss_mean [1] (may be NULL).
 *   bwd: dx [B, D] (f32 / bf16) = 2 w (x - emb[y]), w = grad_loss_i[row] (NULL: the scalar grad_scale, e.g. 1 / B).
 */
int se_sqdist_loss_fwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels, const float *emb,
                       int64_t lde, int64_t B, int64_t D, int64_t C, float *loss_i, float *dist_i,
                       float *loss_mean, se_stream_t stream);
int se_sqdist_loss_bwd(const void *x, int x_dtype, int64_t ldx, const int64_t *labels, const float *emb,
                       int64_t lde, const float *grad_loss_i, float grad_scale, int64_t B, int64_t D,
                       int64_t C, void *dx, int dx_dtype, int64_t lddx, se_stream_t stream);

/*
 * Stand-alone L2-normalisation head and its backward.
 * Replaces: utils.l2norm (utils.py:125-127) used as `Lambda(utils.l2norm, name='l2norm')`
 *           (learn_image_embeddings.py:127-128) when the normalised embedding itself is wanted
 *           (feature dumps, learn_image_embeddings.py:270-275; inference).
 *   xhat [B, D] f32 = x * rsqrt(max(sum x^2, 1e-12)); inv_norm [B] f32 (may be NULL in fwd).
 *   bwd: dx = (grad - xhat (xhat . grad)) * inv_norm  (rows on the epsilon clamp: grad * inv_norm)
 */
int se_l2norm_fwd(const void *x, int x_dtype, int64_t ldx, int64_t B, int64_t D, float *xhat,
                  int64_t ldxhat, float *inv_norm, se_stream_t stream);
int se_l2norm_bwd(const float *grad, int64_t ldg, const float *xhat, int64_t ldxhat,
                  const float *inv_norm, int64_t B, int64_t D, float *dx, int64_t lddx,
                  se_stream_t stream);

/*
 * Nearest-class-embedding accuracy metric.
 * Replaces: utils.nn_accuracy(embedding, dot_prod_sim, k) (utils.py:57-100): the dense
 *           contraction y_pred @ embedding.T (utils.py:90 / :78) on MFMA plus the
 *           |best - true| < 1e-6 test (top-k: any of the k best within 1e-6).
 *   y_pred  [B, D] f32 (already normalised when dot_prod_sim, as in the reference)
 *   labels  [B] int64; y_true of the reference is emb[labels]
 *   dot_prod_sim != 0: similarity = y_pred . emb^T, larger is better   (utils.py:87-95)
 *   dot_prod_sim == 0: squared Euclidean distance, smaller is better    (utils.py:73-85)
 *   acc     [B] f32 out (0/1)
 *   scores  [B, C] f32 out, the similarity / distance matrix           (may be NULL)
 *   best    [B] int32 out, argmax / argmin class (lowest index on ties) (may be NULL)
 *   workspace: se_nn_accuracy_workspace_bytes(B, C) bytes, 8-byte aligned (0 for small class sets: one workgroup then walks
 *           all class tiles of its 32 samples; large sets -- C = 1000 -- are cut into class slices whose partial counts meet there)
 */
int64_t se_nn_accuracy_workspace_bytes(int64_t B, int64_t C);
int se_nn_accuracy(const float *y_pred, int64_t ldp, const int64_t *labels, const float *emb,
                   int64_t lde, int64_t B, int64_t D, int64_t C, int dot_prod_sim, int k,
                   float *acc, float *scores, int64_t lds, int32_t *best, void *workspace,
                   int64_t workspace_bytes, se_stream_t stream);

/*
 * Label-embedding baseline loss (Sun et al.), forward and backward.
 * Replaces: labelembed_loss(out1, out2, tar, targets, tau, alpha, beta) and cross_entropy
 *           (learn_labelembedding.py:17-37); the backward is what TF autodiff derives from it
 *           (softmax(out2 / tau), softmax(tar) inside L_o1_emb and the arg-max mask are stop_gradient).
 *   out1, out2, tar [B, C] f32 logits (ld* elements between rows); targets [B] int64
 *   loss_i [B] f32 out (the reference returns it as [B, 1], learn_labelembedding.py:54)
 *   aux    se_labelembed_aux_floats(B) floats, caller-owned: per-sample log-sum-exps, the mask and the
 *          batch scale B / (sum mask + 1e-8); written by fwd, read by bwd
 *   bwd: grad_loss_i [B] f32 upstream gradient (NULL: grad_scale for every sample);
 *        d_out1, d_out2, d_tar [B, C] f32 out, any of them may be NULL
 */
int64_t se_labelembed_aux_floats(int64_t B);
int se_labelembed_loss_fwd(const float *out1, int64_t ld1, const float *out2, int64_t ld2,
                           const float *tar, int64_t ldt, const int64_t *targets, int64_t B, int64_t C,
                           float tau, float alpha, float beta, float *loss_i, float *aux,
                           se_stream_t stream);
int se_labelembed_loss_bwd(const float *out1, int64_t ld1, const float *out2, int64_t ld2,
                           const float *tar, int64_t ldt, const int64_t *targets,
                           const float *grad_loss_i, float grad_scale, int64_t B, int64_t C, float tau,
                           float alpha, float beta, const float *aux, float *d_out1, int64_t ldd1,
                           float *d_out2, int64_t ldd2, float *d_tar, int64_t lddt, se_stream_t stream);

/*
 * DeViSE ranking loss on the class-embedding contraction, forward + backward.
 * Replaces: utils.devise_ranking_loss(embedding, margin)(y_true, y_pred)  (utils.py:103-122)
 *           loss_i = sum_c relu(margin - <y_true_i, y_pred_i> + (y_pred . E^T)[i, c]) - margin
 *           and what TF autodiff derives from it w.r.t. y_pred.
 *   y_pred [B, D] f32; the target rows are E[labels[i]] (labels != NULL, y_true == NULL: the gather of
 *   learn_image_embeddings.py:48-50 done on the device) or the explicit matrix y_true [B, D] (the reference's convention);
 *   emb [C, D] f32; loss_i [B] out.
 *   aux: se_devise_aux_floats(B, C) floats, caller-owned: true_sim [B], active-hinge count [B] and the 0 / 1 hinge mask [B, C]
 *        written by the forward pass and consumed by the backward pass (d y_pred = g_i (mask . E - count_i y_true_i)), followed by
 *        scratch for the forward pass's per-class-slice partial sums when C is large (always ask se_devise_aux_floats for the size).
 *   grad_loss_i [B] or NULL (then every sample uses grad_scale).
 */
int64_t se_devise_aux_floats(int64_t B, int64_t C);
int se_devise_loss_fwd(const float *y_pred, int64_t ldp, const int64_t *labels, const float *y_true, int64_t ldt,
                       const float *emb, int64_t lde, int64_t B, int64_t D, int64_t C, float margin, float *loss_i,
                       float *aux, se_stream_t stream);
int se_devise_loss_bwd(const int64_t *labels, const float *y_true, int64_t ldt, const float *emb, int64_t lde,
                       const float *grad_loss_i, float grad_scale, int64_t B, int64_t D, int64_t C, const float *aux,
                       float *d_pred, int64_t lddp, se_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Retrieval side  (evaluate_retrieval.pairwise_retrieval, evaluate_retrieval.py:22-73)
 * ------------------------------------------------------------------------------------------ */

/* sq[i] = float32 np.sum(x[i]**2)  with NumPy's pairwise summation order, bit-exact.
 * Replaces: `sqnorm = np.sum(features ** 2, axis = -1)`          evaluate_retrieval.py:61 */
int se_row_sqnorm(const float *x, int64_t ldx, int64_t n, int64_t d, float *sq, se_stream_t stream);

/* x[i] /= sqrt(np.sum(x[i]*x[i]))  in place, bit-exact w.r.t. float32 NumPy.
 * Replaces: `features /= np.linalg.norm(features, axis = -1, keepdims = True)`
 *                                                                 evaluate_retrieval.py:58 */
int se_normalize_rows(float *x, int64_t ldx, int64_t n, int64_t d, se_stream_t stream);

/*
 * All-pairs distance matrix  out[i, j] = dist(a[i], b[j]),  i < q, j < n.
 * Replaces: `pdist = -np.dot(features, features.T)`               evaluate_retrieval.py:59
 *           `pdist = ne.evaluate('A + B - 2 * C', ...)`           evaluate_retrieval.py:61-62
 * The dot product is a sequential fp32 FMA chain over k = 0..d-1 (v_mfma_f32_32x32x2_f32),
 * bit-identical to the OpenBLAS sgemm/ssyrk result the reference obtains for d <= 448.
 * kblocks (HOST pointer, may be NULL): K-block lengths summing to d; the chain restarts from 0
 * at each block and block results are added in order (OpenBLAS behaviour for large d).
 *   sqa [q], sqb [n]: row square norms, required for SE_METRIC_EUCLID only.
 */
int se_pairwise_dist(const float *a, int64_t lda, const float *b, int64_t ldb, const float *sqa,
                     const float *sqb, int64_t q, int64_t n, int64_t d, int metric,
                     const int32_t *kblocks, int nkb, float *out, int64_t ldo,
                     se_stream_t stream);

/*
 * Full ranking of every row:  rank[i, r] = column index of the r-th smallest pdist[i, :]
 * under the canonical order (distance ascending, index ascending; NaN last; -0 == +0).
 * Replaces: `ranking = np.argsort(pdist, axis = -1)`              evaluate_retrieval.py:67
 * (the reference's sort is unstable: ties are returned in canonical order here).
 *   rank: int32 [q, n] when idx64 == 0, int64 [q, n] (NumPy's dtype) when idx64 == 1, uint16 [q, n] when idx64 == 2 (rows of at
 *   most 53,248 columns only -- SE_ERR_UNSUPPORTED otherwise: the width the register-resident kernel holds its ranks in; half the
 *   bytes for se_hierarchical_precision_r16 to read).  ldr in elements of that type.
 *   workspace: se_rank_rows_workspace_bytes(q, n) bytes of device memory, 16-byte aligned.  n <= 53,248: one workgroup sorts a row in registers
 *   (4.4 KB of workspace: probe / guard words); 53,248 < n <= 425,984: 2, 4 or 8 segments of a row are sorted the same way into
 *   runs and merged pairwise (merge tree; up to 3 GB of run planes / level buffers for a chunk of rows at a time); longer rows:
 *   LDS-tiled radix sort through 16 bytes per key of scratch per resident workgroup.
 */
int64_t se_rank_rows_workspace_bytes(int64_t q, int64_t n);
int se_rank_rows(const float *pdist, int64_t ldp, int64_t q, int64_t n, void *rank, int idx64,
                 int64_t ldr, void *workspace, int64_t workspace_bytes, se_stream_t stream);

/*
 * One-time set-up of the ranking on the CURRENT device -- the only entry point of the ranking that synchronises by design.
 * The fastest kernels of se_rank_rows take their stable order from the lane order in which the LDS serves same-address
 * returning adds of one wave instruction: gfx950 serves them in ascending lane order, the ISA does not promise it.  This call
 * (1) probes the property (64 workgroups, four conflict patterns, the production counter format) and (2) ranks crafted
 * tie-heavy rows through EVERY hardware-ordered kernel variant (short / long instantiation x plain / group-peeling / two-pass,
 * segment runs + merge) and audits every row of every result (se_rank_rows_check's kernel).  The verdict is cached per device:
 * afterwards se_rank_rows is purely asynchronous -- no probe, no guard, no host round trip; it can be captured into a HIP graph --
 * and uses the hardware-ordered kernels iff both steps passed (else the ballot / tiled kernels, whose order holds by construction).
 * WITHOUT this call the first se_rank_rows of a process on a device does the probe itself and audits 512 sampled rows of its own
 * result (hipStreamSynchronize + one 4-byte copy, ~0.5 ms) before it returns; a call that cannot (no workspace for the guard, or
 * a stream under capture) uses the ballot kernel.
 *   workspace: se_rank_rows_init_workspace_bytes() bytes (~5 MB), 256-byte aligned; may be freed afterwards.  ~3 ms.
 */
int64_t se_rank_rows_init_workspace_bytes(void);
int se_rank_rows_init(void *workspace, int64_t workspace_bytes, se_stream_t stream);

/*
 * Order guard of se_rank_rows: counts the rows of a finished ranking that violate the canonical order -- along every row
 * (pdist[rank[r]], rank[r]) must precede (pdist[rank[r + 1]], rank[r + 1]), every index lies in [0, n).  One gather of the
 * distances through the ranks (random 4-byte reads: ~1 us per row of 50k columns).  se_rank_rows runs the same check by
 * itself -- on 512 evenly spaced rows behind the first ranking a process does with its fastest kernel (whose stable order rests
 * on a hardware property the library can probe but the ISA does not promise), on every row of every call when SE_RANK_CHECK=1
 * is set -- and re-ranks with the guaranteed-order kernel when it finds a violation; this entry point lets a caller audit any
 * ranking in full (np.argsort(pdist, axis=-1, kind='stable') passes it).
 *   workspace: se_rank_rows_check_workspace_bytes() bytes; *bad_rows_host (HOST pointer) receives the count; synchronises.
 */
int64_t se_rank_rows_check_workspace_bytes(void);
int se_rank_rows_check(const float *pdist, int64_t ldp, int64_t q, int64_t n, const void *rank, int idx64,
                       int64_t ldr, void *workspace, int64_t workspace_bytes, int64_t *bad_rows_host,
                       se_stream_t stream);

/*
 * The k nearest columns of every row of a distance matrix, canonical order, with a global
 * column offset (sharded galleries: shard r passes col_offset = first gallery row of the shard).
 *   out_d [q, k] f32, out_i [q, k] int32 (global indices).  k <= n, k <= SE_TOPK_MAX.
 */
#define SE_TOPK_MAX 2048
int se_topk_rows(const float *pdist, int64_t ldp, int64_t q, int64_t n, int64_t col_offset, int k,
                 float *out_d, int32_t *out_i, se_stream_t stream);

/*
 * Merge per-shard top-k lists (e.g. the RCCL all-gather of every rank's se_topk_rows output)
 * into the global top-k; the result is independent of how the gallery was sharded.
 *   d, idx: [parts, q, k];  out_d, out_i: [q, k].  parts * k <= SE_TOPK_MAX * 4.
 *   Lists as se_retrieve_topk / se_topk_rows write them (ascending under the canonical (distance, index) order) merge fastest
 *   (k <= 1024: one wave per query, the running best list in registers); a list that is not ascending is sorted first -- same result.
 */
int se_topk_merge(const float *d, const int32_t *idx, int parts, int64_t q, int k, float *out_d,
                  int32_t *out_i, se_stream_t stream);

/*
 * The same merge on PACKED lists: part p is one contiguous block of 2 q k 32-bit words -- its [q, k] distances (f32) followed by
 * its [q, k] indices (i32).  That is the receive buffer of ONE all-gather in which every rank contributes its (dist | idx) block
 * (sharded_retrieval.all_gather_lists): half the collectives of gathering distances and indices separately.
 */
int se_topk_merge_packed(const void *packed, int parts, int64_t q, int k, float *out_d, int32_t *out_i,
                         se_stream_t stream);

/*
 * Fused distance + top-k: the k nearest gallery rows of every query WITHOUT the [q, n] distance matrix
 * (SURVEY.md section 8d "fused top-k": bytes = 4 (q + n) d + 8 q k).
 * Replaces: the head of evaluate_retrieval.py:57-67 (normalise / distances / np.argsort) for consumers that read only the
 *           first k entries of every ranking (P@k and clipped AHP without AP, class_hierarchy.py:300-309), and it is the
 *           per-shard step of the sharded-gallery split (gallery shard r passes col_offset = its first global row).
 * Same arithmetic as se_pairwise_dist (sequential fp32 FMA chain, optional K-block list `kblocks` -- HOST pointer, may be
 * NULL -- for d > 448, evaluate_retrieval.py:59 on OpenBLAS) and the same canonical order as se_rank_rows: out_i[i, :] ==
 * the first k entries of se_rank_rows(se_pairwise_dist(queries, gallery))[i], out_d the distances, bit for bit.
 * Galleries of >= 16384 rows (k <= 512): the Q x N distances are first BOUNDED, not computed -- a pass on the fp16 matrix cores
 * (v_mfma_f32_32x32x16_f16) over half-precision IMAGES of the operands (each matrix scaled by one power of two so that its largest
 * entry sits just below 2^14, entries below fp16's normal range flushed to zero, columns padded to a multiple of 128) gives d~
 * with |d~ - d| <= eps(query) (a rigorous bound from the operands' actual rounding residuals).  A sample of <= 4096 gallery rows gives every query a threshold, the full pass appends the few
 * items with d~ <= threshold to per-query candidate lists, and a per-query kernel recomputes, with the exact fp32 chain, the items
 * within 2 eps of the list's k-th smallest d~, sorts them and accepts the first k once it has proved that nothing outside that set
 * can precede them; queries it cannot prove (short / overflowing lists, NaN rows, tie groups of thousands) are redone exactly over the
 * whole gallery.  The output does not depend on what the half-precision pass computed.  Smaller galleries go through a [rows, n] distance slab
 * in the workspace; k > 512 through the fp32 form of the same passes.
 *   metric: SE_METRIC_COSINE or SE_METRIC_EUCLID (then sqq [q], sqg [n] = se_row_sqnorm of the operands).
 *   workspace: se_retrieve_topk_workspace_bytes(q, n, d, ldg, k) bytes, 256-byte aligned -- the ONLY supported way to size it (it
 *   holds the candidate sub-lists, the fp16 images at 2 bytes x (n + q) x (d rounded up to a multiple of 128), per-row norms and
 *   bounds, and the scratch rows of the exact fallback; the split between them follows the pass geometry chosen for (q, n, d)).
 */
int64_t se_retrieve_topk_workspace_bytes(int64_t q, int64_t n, int64_t d, int64_t ldg, int k);
int se_retrieve_topk(const float *queries, int64_t ldq, const float *gallery, int64_t ldg,
                     const float *sqq, const float *sqg, int64_t q, int64_t n, int64_t d,
                     int metric, const int32_t *kblocks, int nkb, int64_t col_offset, int k,
                     float *out_d, int32_t *out_i, void *workspace, int64_t workspace_bytes,
                     se_stream_t stream);

/*
 * Hierarchical retrieval metrics of every query from its ranking (the consumer of se_rank_rows / se_retrieve_topk).
 * Replaces: the per-query loop of ClassHierarchy.hierarchical_precision (class_hierarchy.py:211-316):
 *           P@k (WUP / LCS_HEIGHT), AHP or AHP@K (np.trapz of cum / best, :303-309), AP (:310-314), with the
 *           reference's handling of the query inside its own ranking (:280-290).
 *   rank      [q, list_len] int32 gallery indices, best first (ldr elements between rows)
 *   cls       [gallery] int32 class index of every gallery item (the kernel keeps a byte / 16-bit copy in LDS when
 *             num_classes and gallery allow);  qcls [q] class index of every query
 *   qidx      [q] int32 gallery index of the query itself (dropped from its ranking), NULL = keep everything
 *   wup, lcs  [C, C] f64 class similarity tables (Wu-Palmer, 1 - LCS height / max height)
 *   rcp       the best-possible cumulative similarity per query class (class_hierarchy.py:266,275), pre-divided
 *             for the kernel by se_hprec_reciprocal_curves (once per gallery); rcp_len = the list_len it was built
 *             for (>= this call's list_len)
 *   ks        [nk] int32 cut-offs; ahp_len: -1 no AHP, 0 whole list, K > 0 clipped AHP@K; want_ap: 0 / 1
 *   out       [q, 2 nk + 3] f64: P@k WUP x nk, P@k LCS x nk, AHP WUP, AHP LCS, AP (ldo elements between rows)
 *   order_ws  NULL, or se_hprec_order_workspace_bytes(q) bytes of 16-byte aligned device scratch: the queries are then visited in
 *             class order (one contiguous part of it per XCD), which keeps the best curve being streamed in L2.
 *             The results do not depend on it.
 */
int64_t se_hprec_order_workspace_bytes(int64_t q);
int se_hierarchical_precision(const int32_t *rank, int64_t ldr, int64_t q, int64_t list_len,
                              const int32_t *cls, int64_t gallery, const int32_t *qcls, const int32_t *qidx,
                              const double *wup, const double *lcs, int num_classes,
                              const double *rcp, int64_t rcp_len,
                              const int32_t *ks, int nk, int64_t ahp_len, int want_ap, double *out,
                              int64_t ldo, void *order_ws, se_stream_t stream);
/* The same for rankings of uint16 gallery indices (se_rank_rows with idx64 == 2; gallery <= 65,536): identical results, half the
 * ranking bytes to read.  Replaces the same lines of class_hierarchy.py:211-316. */
int se_hierarchical_precision_r16(const uint16_t *rank, int64_t ldr, int64_t q, int64_t list_len,
                                  const int32_t *cls, int64_t gallery, const int32_t *qcls, const int32_t *qidx,
                                  const double *wup, const double *lcs, int num_classes,
                                  const double *rcp, int64_t rcp_len,
                                  const int32_t *ks, int nk, int64_t ahp_len, int want_ap, double *out,
                                  int64_t ldo, void *order_ws, se_stream_t stream);

/*
 * The best-possible curves of se_hierarchical_precision, pre-divided and laid out for its loads.
 *   best_*    [num_classes, ldb] f64: for query class c, the cumulative sum of the descending-sorted similarities of
 *             the WHOLE gallery to c (class_hierarchy.py:266,275) -- host-side, once per gallery; list_len positions used
 *   rcp       [num_classes, 2, se_hprec_curve_len(list_len), 2] f64 out, per class: the divisors of the ranks BEHIND the
 *             query, (1 / (best_wup[i] - 1), 1 / (best_lcs[i] - 1)) -- dropping the query from its ranking shifts the
 *             curve and subtracts its self-similarity (class_hierarchy.py:280-290) -- then those of the ranks AHEAD of
 *             it, (1 / best_wup[i], 1 / best_lcs[i]).  Both chunk-transposed (4096-position chunks, position 16 t + e of
 *             a chunk at slot 256 e + t) so that the 256 threads of the metric kernel, each owning 16 consecutive ranks,
 *             read contiguous 16-byte pairs.
 */
int64_t se_hprec_curve_len(int64_t list_len);
int se_hprec_reciprocal_curves(const double *best_wup, const double *best_lcs, int64_t ldb,
                               int num_classes, int64_t list_len, double *rcp, se_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEHIP_H */
