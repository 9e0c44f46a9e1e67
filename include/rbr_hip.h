/*
 * rbr_hip.h -- C ABI of librbr_hip.so, the MI355X (gfx950) review-encoder hot path.
 *
 * Drop-in boundary.  The reference (H263/review-based-recommender) is pure Python on
 * PyTorch and has no FFI of its own: the boundary it exposes is the nn.Module layer
 * (SURVEY.md §8b).  Each entry point below replaces the ATen call sequence of one
 * reference layer; the reference file:line it replaces is cited per function.  The
 * Python modules in review-based-recommender_amd/ (same class names, ctor/forward
 * signatures and state_dict keys as the reference) are the only callers.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host";
 *   - tensors are dense row-major fp32 unless stated; ids are int64, masks are uint8
 *     (torch.bool storage), argmax indices are int32;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); launches are
 *     asynchronous, nothing is allocated, freed or synchronised inside a call
 *     (graph-capture safe);
 *   - return value: 0 on success, otherwise a hipError_t / negative rbr error code;
 *     rbr_last_error() returns a thread-local description.
 */
#ifndef RBR_HIP_H
#define RBR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBR_MAX_WIDTHS 8

#define RBR_PAD_SAME 0  /* odd kz, zero pad (kz-1)/2 each side: MyConv1d (deepconn/layers.py:41-44) */
#define RBR_PAD_VALID 1 /* no padding, pool over L-kz+1:  GlobalAttention convs (dual_att/layers.py:68-79) */
#define RBR_ACT_RELU 0
#define RBR_ACT_TANH 1

#define RBR_ERR_BAD_ARG (-1)
#define RBR_ERR_UNSUPPORTED (-2)

/* Shape of one TextCNN call: n_docs documents of L tokens, a [V, D] word table and
 * n_widths conv banks (width-major output channel order, as torch.cat in
 * deepconn/layers.py:58). */
typedef struct rbr_textcnn_desc {
    int32_t n_docs;
    int32_t L;
    int32_t D;
    int32_t V;
    int32_t n_widths;
    int32_t kz[RBR_MAX_WIDTHS]; /* kernel width of bank w */
    int32_t ch[RBR_MAX_WIDTHS]; /* output channels of bank w */
    int32_t pad_mode;           /* RBR_PAD_* */
    int32_t act;                /* RBR_ACT_* */
    int32_t padding_idx;        /* table row that never receives gradient (nn.Embedding padding_idx); -1 = none */
    int32_t flags;              /* RBR_CONV_* bits; 0 = none */
} rbr_textcnn_desc;
/* RBR_CONV_PAD_RUNS (un-masked convs: D-ATT, whose documents are right-padded with padding_idx and carry no masks,
 * dual_att/layers.py:43-53,81-89): the caller guarantees that a position whose tokens are all padding_idx within 8 positions
 * either side sees the same gate value as every other such position of its document (true for the local gate -- a 5-wide
 * window of the same tokens -- and for the global gate, one scalar per document).  Every such position then has the SAME conv
 * output, so a 32-token slab that is all padding (halo included) and FOLLOWS another such slab cannot change max / first argmax
 * and is not computed.  Exact; valid-padded or width-1 banks only (the flag is ignored otherwise). */
#define RBR_CONV_PAD_RUNS 1
/* RBR_CONV_GATE_SPLIT(k), 1 <= k < n_widths (token-product formulation only): the banks [0, k) and [k, n_widths) have a gate of
 * their own -- `gate` (and `dgate`) are then [2][n_docs][L], plane 0 for the first group.  D-ATT's two gated convs of a tower
 * over the same tokens (dual_att/layers.py:43-53 and 81-89: the 1-wide local conv with the per-token gate, the 2/3/4-wide
 * global convs with the per-document gate; a 1-wide 'same' conv IS a valid conv) become ONE conv call: one token list, one
 * product-table GEMM, one gather launch, one G and one sparse product in the backward.  Entry points of the dense formulation
 * return RBR_ERR_UNSUPPORTED for a split gate. */
#define RBR_CONV_GATE_SPLIT(k) (((k) & 0xf) << 8)
#define RBR_CONV_GATE_SPLIT_OF(flags) (((flags) >> 8) & 0xf)
/* Arithmetic class STAMPED into a descriptor (rbr_textcnn_desc_stamp): bits 16-17 the RBR_PROD_* precision, bit 18 bf16 storage of
 * the product table, bit 19 "stamped".  A stamped descriptor carries the class it was planned with from the forward to the
 * backward: the workspace layout (weight image, bf16 row copy, element type of T) depends on it, and reading the process-wide
 * settings again in every stage let a change between a forward and its backward -- or between a graph capture and a later
 * eager call on the same workspace -- silently reinterpret T.  An unstamped descriptor (flags bit 19 clear) reads the
 * process-wide settings at every call, as before.  Reference counterpart: none (the reference has one arithmetic: fp32). */
#define RBR_CONV_CLASS_STAMPED (1 << 19)
#define RBR_CONV_CLASS_PRECISION_OF(flags) (((flags) >> 16) & 0x3)
#define RBR_CONV_CLASS_T_BF16 (1 << 18)
void rbr_textcnn_desc_stamp(rbr_textcnn_desc* d);

int rbr_version(void);
const char* rbr_last_error(void);

/* ---- Pair regions: two independent problems of EQUAL SHAPES through shared launches.
 * D-ATT's two towers (reference models/dual_att/dual_att.py:45-57: separate parameters and documents, the same op sequence)
 * are ~40 mostly short kernels each.  Between rbr_pair_begin() and rbr_pair_end() the entry points of this library that launch
 * through rbr::launch (csrc/rbr_launch.h: every kernel of the D-ATT step -- gates, the token-product conv chain and its backward
 * -- and the zero fills) do not launch: they RECORD their launches, for problem 0 until rbr_pair_next(), then for problem 1.
 * rbr_pair_end() zips the two lists: launches i of both problems that are the same kernel over the same grid leave as ONE
 * launch with gridDim.z = 2 (a workgroup takes its argument set by blockIdx.z); anything that does not line up is launched
 * one by one in recorded order, so the results never depend on whether pairs formed -- only the launch count and the time do.
 * Contract: inside a region (a) the two problems must not depend on each other, (b) nothing else may be enqueued on the
 * streams involved (the recorded launches run at rbr_pair_end, not where they were called), (c) entry points that launch
 * directly refuse with RBR_ERR_UNSUPPORTED.  The region is per host thread.  rbr_pair_abort() drops an open region (error
 * paths).  n_paired / n_single (may be NULL): launches that left as pairs / singly.
 * Reference counterpart: none (the reference runs the towers one after the other, dual_att.py:47-57). */
int rbr_pair_begin(void);
int rbr_pair_next(void);
int rbr_pair_end(int32_t* n_paired, int32_t* n_single);
void rbr_pair_abort(void);

/* ---- TextCNN encoder: WordEmbedding + masked_tensor + MyConv1d + ReLU/Tanh + MaxPool1d(seq_len)
 *      replaces deepconn/layers.py:22-24 (embedding), deepconn/utils.py:49-61 (masked_fill),
 *      layers.py:46-60 (multi-width conv1d + cat), layers.py:107-109 (ReLU, MaxPool1d) and their
 *      narre/layers.py:119-153,365-401 twins; with RBR_PAD_VALID/RBR_ACT_TANH also the gated
 *      convs of dual_att/layers.py:37-40,68-79.                                            ---- */

/* number of floats of the packed-weight buffer / number of elements of EACH of the two partial-max
 * workspaces (pval: float, pidx: int32) for `d`.  The count includes a scheduling region at the end of pidx
 * (flags | work list | counter) that rbr_textcnn_conv_fwd fills and rbr_textcnn_pool_finalize reads: wave-tiles
 * whose tokens are all masked are not computed, their pooled value is exactly 0 */
size_t rbr_textcnn_packed_floats(const rbr_textcnn_desc* d);
size_t rbr_textcnn_partial_elems(const rbr_textcnn_desc* d);

/* Stage 1: repack the per-width Conv1d weights W[w] ([ch[w], D, kz[w]], torch layout) into the
 * MFMA tile-major image the conv kernel streams.  `W` is a HOST array of device pointers. */
int rbr_textcnn_pack(const rbr_textcnn_desc* d, const float* const* W, float* packed, void* stream);

/* Stage 2 (the dominant stage): (max, first-argmax) of every conv channel over each 32-token slab:
 * pval/pidx[partial_elems].  Rows are table[ids], zero where mask == 0 or out of range, scaled by gate[doc,l]
 * when gate != NULL.  Two exact formulations, chosen per call:
 *   dense  : gather the rows and run every conv width on the f32 MFMA pipe (what the reference computes);
 *   product: when `ws` (rbr_textcnn_fwd_ws_bytes(d) bytes, > 0 only when worthwhile) and `W` (HOST array of the
 *            torch-layout conv weights) are given: T[token][tap, channel] = <table[token], W[channel, :, tap]> for
 *            the DISTINCT tokens of the batch on the MFMA pipe, then a gather-add of kz rows of T per position.
 * rbr_set_conv_mode: 0 auto (default; env RBR_CONV_MODE=dense|product overrides), 1 dense, 2 product. */
size_t rbr_textcnn_fwd_ws_bytes(const rbr_textcnn_desc* d);
void rbr_set_conv_mode(int32_t mode);
/* Arithmetic of the product formulation's GEMM (the contraction of MyConv1d.forward, deepconn/layers.py:46-60, as
 * called from narre.py:175-176), all with f32 accumulation and an f32 product table:
 *   RBR_PROD_F32    v_mfma_f32_32x32x2_f32 on f32 operands (an f32 fma chain, bit for bit)
 *   RBR_PROD_BF16X3 default: each f32 operand split exactly into three bf16 planes, 6 plane products per f32
 *                   product on v_mfma_f32_32x32x16_bf16 (neglected terms < 2^-25 |a b|: f32-class accuracy)
 *   RBR_PROD_BF16X2 two planes, 3 plane products (~2^-17 relative per product)
 *   RBR_PROD_BF16   operands rounded to bf16, one MFMA per 16 products: the reduced-precision row of
 *                   BASELINE configs 3 and 5 (tolerance class of its own, see tests/test_precision_gpu.py)
 * Operand range of RBR_PROD_BF16X3: the split is exact (and the class f32-accurate) while the LOW plane, 2^-16 of the
 * operand, is a normal number: |x| >= ~2^-110.  Smaller operands lose their low plane (accuracy slides towards BF16X2, measured
 * 1e-3 relative at 2^-118); bf16 has f32's exponent range, so large operands overflow exactly where f32 products would.
 * A row holding inf / NaN is outside the supported domain in every mode (the split's x - hi = inf - inf poisons the lower
 * planes, and the max-pool's comparisons drop NaN where torch's max_pool1d propagates it); documents that do not read such a
 * row are untouched (tests/test_precision_gpu.py).
 * The word table and the conv weights stay f32 in memory in every mode.  Set once, before the first forward of a step
 * (the workspace layout depends on it); -1 restores the default (env RBR_PROD_PRECISION=f32|bf16x3|bf16x2|bf16).
 * D % 4 != 0 always takes the f32 kernel. */
#define RBR_PROD_F32 0
#define RBR_PROD_BF16X3 1
#define RBR_PROD_BF16X2 2
#define RBR_PROD_BF16 3
void rbr_set_prod_precision(int32_t mode);
int32_t rbr_get_prod_precision(void);
/* RBR_PROD_BF16 with bf16 STORAGE (default on; 0 = f32 streams, -1 = default / env RBR_B16_STORAGE): the byte streams of the
 * conv stage are bf16 as well -- the distinct tokens' rows are rounded once into a compact bf16 copy the GEMM reads (600-byte
 * rows by plain index, no conversion in the loop), and the product table T is stored in bf16 (the gather-add reads half the
 * bytes; its sums stay f32).  Same tolerance class as RBR_PROD_BF16 with f32 streams (tests/test_precision_gpu.py); parameters,
 * Adam state and every gradient stay f32.  Like the precision itself: set between steps only. */
void rbr_set_b16_storage(int32_t on);
int rbr_textcnn_conv_fwd(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                         const float* table, const float* const* W, const float* packed, float* pval, int32_t* pidx,
                         void* ws, void* stream);

/* The three stages rbr_textcnn_conv_fwd runs for the product formulation, callable separately (same `ws`):
 *   prepare: work list of the documents (tail of pidx), distinct unmasked tokens of ids -> list / inverse map,
 *            product weight images (forward tile image and row-major Wprod^T for the backward);
 *   table  : ONE kernel, T = table[distinct tokens] @ Wprod on the f32 MFMA pipe;
 *   pool   : ONE kernel, per 32-token slab add the kz rows of T per position (x gate), max / first argmax ->
 *            pval / pidx (same pidx as prepare).
 * They fail with RBR_ERR_UNSUPPORTED when rbr_textcnn_fwd_ws_bytes(d) == 0. */
int rbr_textcnn_prod_prepare(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* const* W,
                             int32_t* pidx, void* ws, void* stream);
int rbr_textcnn_prod_table(const rbr_textcnn_desc* d, const float* table, void* ws, void* stream);
int rbr_textcnn_prod_pool(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                          float* pval, int32_t* pidx, void* ws, void* stream);
/* Step-level fusion of the prepare stage (round 3; removes a launch from a training step):
 *   prod_prepare_ids: prepare with the model's id range check (rbr_sanitize_ids below, nn.Embedding's IndexError,
 *            deepconn/layers.py:23) in its first launch.  `sets` (HOST array) are the forward's raw id tensors and their clean
 *            copies; `ids` = sets[0].out must hold the conv's d->n_docs * d->L token ids (one set, or two adjacent = both towers);
 */
struct rbr_id_set;
int rbr_textcnn_prod_prepare_ids(const rbr_textcnn_desc* d, int32_t n_sets, const struct rbr_id_set* sets, int64_t* err,
                                 const int64_t* ids, const uint8_t* mask, const float* const* W, int32_t* pidx, void* ws,
                                 void* stream);

/* Stage 3: reduce the slabs of each document, add the conv bias, apply the activation.
 * feat[n_docs, C] (C = sum ch[w]); argmax[n_docs, C] = first position attaining the max.
 * `bias` is a HOST array of device pointers (one [ch[w]] vector per width). */
int rbr_textcnn_pool_finalize(const rbr_textcnn_desc* d, const float* pval, const int32_t* pidx,
                              const float* const* bias, float* feat, int32_t* argmax, void* stream);

/* Backward of the whole encoder from d_feat[n_docs, C], using the max-pool sparsity (the gradient
 * of out[doc,c] reaches exactly one conv window).  Replaces convolution_backward, max_pool
 * backward, masked_fill backward and embedding_dense_backward of loss.backward()
 * (trainer/train_deepconn_pp.py:165).
 *   dW[w]    [ch[w], D, kz[w]]  overwritten          (host arrays of device pointers)
 *   dbias[w] [ch[w]]            overwritten
 *   dtable   [V, D]             ACCUMULATED (caller zeroes); NULL = frozen embeddings
 *   dgate    [n_docs, L]        ACCUMULATED (caller zeroes); NULL when gate == NULL
 *   ws       workspace of rbr_textcnn_bwd_ws_floats(d) floats                                  */
size_t rbr_textcnn_bwd_ws_floats(const rbr_textcnn_desc* d);
/* the two halves of rbr_textcnn_bwd, callable separately (conv weight / bias gradients; table / gate gradients) */
int rbr_textcnn_bwd_dw(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                       const float* table, const float* feat, const int32_t* argmax, const float* d_feat,
                       float* const* dW, float* const* dbias, float* ws, void* stream);
int rbr_textcnn_bwd_dtable(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                           const float* table, const float* packed, const float* feat, const int32_t* argmax,
                           const float* d_feat, float* dtable, float* dgate, void* stream);
/* Table (and gate) gradient through the token-product formulation (same maths as rbr_textcnn_bwd_dtable; with a gate,
 * dgate[doc,p] is OVERWRITTEN -- zeroed by the call, then summed from the forward's product table T, which must still be
 * intact in `fwd_ws`):
 * G[token][tap, channel] = sum of g over the argmax windows touching that token, for the DISTINCT tokens the
 * forward listed in `fwd_ws` (the workspace rbr_textcnn_conv_fwd was given for the SAME ids/mask, still intact),
 * then dtable[token, :] = G[token, :] @ Wprod^T as a sparse row product (G is ~2 % dense); each listed row of
 * `dtable` is written once with plain stores and every other row is set to zero: the whole [V, D] gradient is
 * OVERWRITTEN (no pre-fill needed).  `bwd_ws`: rbr_textcnn_bwd_prod_ws_bytes(d)
 * bytes (0 = formulation not applicable: use rbr_textcnn_bwd_dtable; env RBR_DTABLE_MODE=scatter forces that). */
size_t rbr_textcnn_bwd_prod_ws_bytes(const rbr_textcnn_desc* d);
/* The same table gradient for a forward that ran in the DENSE formulation (no workspace of its own): builds the
 * distinct-token list of ids/mask and Wprod^T from the conv weights W (HOST array) in `ws`, then G and the sparse product as
 * above; the whole dtable [V, D] is OVERWRITTEN.  Un-gated convs only.  ws: rbr_textcnn_bwd_dtable_list_ws_bytes(d) bytes
 * (0: D % 4 != 0, the list's worst-case G exceeds 256 MB, or RBR_DTABLE_MODE=scatter -> use rbr_textcnn_bwd_dtable, whose
 * window scatter issues one row of f32 atomics per (document, channel, tap)). */
size_t rbr_textcnn_bwd_dtable_list_ws_bytes(const rbr_textcnn_desc* d);
int rbr_textcnn_bwd_dtable_list(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* const* W,
                                const float* feat, const int32_t* argmax, const float* d_feat, void* ws, float* dtable,
                                void* stream);
int rbr_textcnn_bwd_dtable_prod(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                                float* dtable, float* dgate, void* stream);
/* rbr_textcnn_bwd_dtable_prod with the rows added to dtable instead of the whole [V,D] gradient being overwritten (see
 * `accumulate` of the D-ATT gate backwards). */
int rbr_textcnn_bwd_dtable_prod_acc(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                    const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                                    float* dtable, float* dgate, void* stream);
/* The two halves of rbr_textcnn_bwd_dtable_prod as separate calls (G is always built; dgate as above): for a caller that
 * runs other consumers of G -- rbr_textcnn_bwd_dw_from_g -- beside the product, on another stream. */
int rbr_textcnn_bwd_g_build(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                            const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                            float* dgate, void* stream);
int rbr_textcnn_bwd_g_product(const rbr_textcnn_desc* d, void* fwd_ws, void* bwd_ws, float* dtable, void* stream);
/* Every form of the token-product table gradient behind one entry (the four calls above are fixed-flag forms of it):
 *   RBR_G_BUILD | RBR_G_PRODUCT  the phases to run (G from the argmax windows; dtable = G @ Wprod^T);
 *   RBR_G_ACCUMULATE             the batch's rows are ADDED to the dense dtable [V, D];
 *   RBR_G_ROWS                   `dtable` is the gradient in COMPACT form [list rows, D]: row r belongs to token tok_of_row[r]
 *                                of the forward's list (rbr_textcnn_token_list), tokens the batch does not hold have no row
 *                                (their gradient is zero and is never written: at cfg2 that is 57 % of embedding_dense_backward's
 *                                [V, D] output, deepconn/layers.py:22-24), and sq_part[rbr_textcnn_row_grad_partials(d)] receives
 *                                per-workgroup sums of squares of the rows in a fixed order (the table's share of
 *                                clip_grad_norm_'s norm).  Consumers: rbr_clip_adam_step_rows, rbr_row_grad_to_dense;
 *   RBR_G_ZEROED                 G's rows are already zero (a caller that cleared `bwd_ws` itself): no zero launch here.
 * sq_part is only read with RBR_G_ROWS; dgate as for rbr_textcnn_bwd_dtable_prod. */
#define RBR_G_BUILD 1
#define RBR_G_PRODUCT 2
#define RBR_G_ACCUMULATE 4
#define RBR_G_ROWS 8
#define RBR_G_ZEROED 16
int rbr_textcnn_bwd_dtable_prod_ex(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                                   const float* feat, const int32_t* argmax, const float* d_feat, void* fwd_ws, void* bwd_ws,
                                   float* dtable, float* dgate, float* sq_part, int32_t flags, void* stream);
size_t rbr_textcnn_row_grad_partials(const rbr_textcnn_desc* d);

/* Device addresses of the forward's token list inside `fwd_ws` (its layout is private): row_of_token [V] (list row or -1),
 * n_rows [1] (rows in the list), tok_of_row [cap]; *cap = rows a compact gradient must have room for.  Any out-pointer may be NULL. */
int rbr_textcnn_token_list(const rbr_textcnn_desc* d, void* fwd_ws, const int32_t** row_of_token, const int32_t** n_rows,
                           const int64_t** tok_of_row, int32_t* cap);
/* Conv weight / bias gradients from the G the call above left in `bwd_ws` (dW = G^T @ table[distinct tokens] on the f32
 * MFMA pipe, split over token ranges, fixed-order reduce).  For many short documents (NARRE's reviews) this replaces
 * rbr_textcnn_bwd_dw; rbr_textcnn_bwd_dw_from_g_ws_floats(d) == 0 means "use rbr_textcnn_bwd_dw".  Needs the SAME fwd_ws /
 * bwd_ws as the rbr_textcnn_bwd_dtable_prod call that ran just before it (with dtable != NULL). */
size_t rbr_textcnn_bwd_dw_from_g_ws_floats(const rbr_textcnn_desc* d);
int rbr_textcnn_bwd_dw_from_g(const rbr_textcnn_desc* d, const float* table, const float* feat, const float* d_feat,
                              void* fwd_ws, void* bwd_ws, float* const* dW, float* const* dbias, float* ws, void* stream);
/* Data-parallel exchange of the table gradient in "tap" form (replaces the dense all-reduce of the [V, D] gradient that
 * nn.DataParallel / a DDP bucket would do for the embedding, train_deepconn_pp.py:129-131): with identical conv weights on
 * every rank, dtable = ((1/N) sum_r G_r) @ Wprod^T, and G_r has one non-zero per (document, channel, tap).
 *   rbr_textcnn_taps_count(d)      : n_docs * C * KF entries (KF = widest kernel); entry e = ((doc * C + c) * KF + j)
 *   rbr_textcnn_bwd_taps           : tok[e] = token the tap lands on or -1 (tap beyond the kernel, g == 0, masked,
 *                                    out of the document, padding_idx), val[e] = d_feat * act'(feat) or 0.  gate == NULL only.
 *   rbr_textcnn_dtable_from_taps   : `tok` / `val` hold n_sets concatenated tap sets of the SAME descriptor (the all-gathered
 *                                    ranks); OVERWRITES the whole dtable [V, D] with the MEAN over the sets.  The taps are
 *                                    sorted by token, each token's row is accumulated in LDS in 64-bit fixed point (2^-40
 *                                    units, integer atomics: independent of the order, so replicas stay bit-identical) and
 *                                    multiplied by Wprod^T.  W: HOST array of the conv weights;
 *                                    ws: rbr_textcnn_dtable_from_taps_ws_bytes(d, n_sets) bytes (0: unsupported, D % 4 != 0). */
size_t rbr_textcnn_taps_count(const rbr_textcnn_desc* d);
int rbr_textcnn_bwd_taps(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* feat,
                         const int32_t* argmax, const float* d_feat, int32_t* tok, float* val, void* stream);
size_t rbr_textcnn_dtable_from_taps_ws_bytes(const rbr_textcnn_desc* d, int32_t n_sets);
int rbr_textcnn_dtable_from_taps(const rbr_textcnn_desc* d, int32_t n_sets, const int32_t* tok, const float* val,
                                 const float* const* W, void* ws, float* dtable, void* stream);
/* Owner partition of the same rebuild (rank r of n_sets builds only the rows of the tokens t with t % n_sets == r; the ranks
 * exchange the slabs afterwards -- distributed.OwnerExchange -- so sort and row build shrink to 1 / n_sets per rank):
 *   rbr_textcnn_taps_owner_rows(d, n_sets)  : rows of a slab = ceil(V / n_sets); row i of rank r's slab is token i * n_sets + r
 *                                             (rows past the vocabulary, and tokens without taps, are zero rows);
 *   rbr_textcnn_dtable_from_taps_owner      : OVERWRITES slab [rows, D] with the MEAN over the sets; same `tok` / `val` / W / ws
 *                                             (rbr_textcnn_dtable_from_taps_ws_bytes) as above.  The rank's taps are compacted
 *                                             (stably) before the sort, which is sized for twice the even share of the taps:
 *                                             *overflow (device, caller-zeroed, sticky) is set to 1 when the rank owns more --
 *                                             its slab is then incomplete: the caller sends the flag along with the slab and,
 *                                             when ANY rank raised it, every rank rebuilds that step with
 *                                             rbr_textcnn_dtable_from_taps instead (distributed.TapExchange._finish_owner). */
int32_t rbr_textcnn_taps_owner_rows(const rbr_textcnn_desc* d, int32_t n_sets);
int rbr_textcnn_dtable_from_taps_owner(const rbr_textcnn_desc* d, int32_t n_sets, int32_t rank, const int32_t* tok,
                                       const float* val, const float* const* W, void* ws, float* slab, int32_t* overflow,
                                       void* stream);
int rbr_textcnn_bwd(const rbr_textcnn_desc* d, const int64_t* ids, const uint8_t* mask, const float* gate,
                    const float* table, const float* packed, const float* feat, const int32_t* argmax,
                    const float* d_feat, float* const* dW, float* const* dbias, float* dtable, float* dgate,
                    float* ws, void* stream);

/* ---- Rating head: LastFeat x2 + FM  (deepconn/layers.py:156-165,189-209; narre.py:84-93,118-137)
 *   ul = u_feat @ Wu + bu + Eu[u_id];  il = i_feat @ Wi + bi + Ei[i_id]
 *   pred = (relu(ul*il) * drop) @ h + ub[u_id] + ib[i_id] + g
 * drop [B,K] is the dropout multiplier (0 or 1/(1-p)) drawn by the caller; NULL = no dropout.
 * ul, il [B,K] are saved for the backward.                                                  ---- */
typedef struct rbr_head_params {
    const float* Wu; const float* bu; const float* Eu;   /* [H,K] [K] [U,K] */
    const float* Wi; const float* bi; const float* Ei;   /* [H,K] [K] [I,K] */
    const float* h;  const float* g;                     /* [K]   [1] */
    const float* ub; const float* ib;                    /* [U]   [I] */
} rbr_head_params;

typedef struct rbr_head_grads {
    float* dWu; float* dbu; float* dEu;                  /* dEu/dEi/dub/dib ACCUMULATED (caller zeroes) */
    float* dWi; float* dbi; float* dEi;
    float* dh;  float* dg;
    float* dub; float* dib;
} rbr_head_grads;

int rbr_pair_head_fwd(int32_t B, int32_t H, int32_t K, const float* u_feat, const float* i_feat,
                      const int64_t* u_id, const int64_t* i_id, const rbr_head_params* p, const float* drop,
                      float* ul, float* il, float* pred, void* stream);

/* Training forward in one launch: the same computation with the dropout multiplier drawn inside the kernel -- element
 * (b, k) gets exactly the value rbr_dropout_multiplier(B*K, p_drop, seed, rng_state, ...) would write at b*K + k for the
 * same call number, and the call number advances once -- and written to drop_out [B,K] for the backward (p_drop == 0: no
 * dropout, rng_state / drop_out may be NULL).  zero_buf / zero_n (optional): a float buffer cleared by spare workgroups of
 * the same launch; the module passes the embedding-style gradients rbr_pair_head_bwd accumulates into. */
int rbr_pair_head_fwd_train(int32_t B, int32_t H, int32_t K, const float* u_feat, const float* i_feat,
                            const int64_t* u_id, const int64_t* i_id, const rbr_head_params* p, float p_drop,
                            uint64_t seed, uint64_t* rng_state, float* drop_out, float* zero_buf, int64_t zero_n,
                            float* ul, float* il, float* pred, void* stream);

/* The head forward of a two-tower model whose encoder ran rbr_textcnn_prod_pool / _conv_fwd on the STACKED batch (B user
 * documents, then B item documents; d->n_docs = 2B): rbr_textcnn_pool_finalize (deepconn/layers.py:107-109: bias, ReLU, the
 * max over all slabs), the head above (drop: a given multiplier, or p_drop > 0: drawn in-kernel as rbr_pair_head_fwd_train
 * does) and -- when target != NULL -- the trainers' nn.MSELoss(mean) (train_deepconn_pp.py:137,164: loss[0], d_pred_unit as
 * rbr_mse_loss_fwd) in ONE launch instead of three.  feat / argmax [2B, C] are written for the backward.
 * first [2B] or NULL: document r takes its pool partials from document first[r] (rbr_dedup_rows: the repeated documents of the
 * batch were blanked and not encoded); feat / argmax rows of r are then copies of first[r]'s.  ticket: one int32
 * in device memory, zero before the first call (the launch re-arms it); calls that may be in flight at the same time (different
 * streams) need a ticket each -- the binding keeps one per (device, stream).  RBR_ERR_UNSUPPORTED when the conv has more than
 * 256 channel slots. */
int rbr_pair_head_fwd_pool(const rbr_textcnn_desc* d, const float* pval, const int32_t* pidx, const float* const* bias,
                           float* feat, int32_t* argmax, const int64_t* first, int32_t K, const int64_t* u_id, const int64_t* i_id,
                           const rbr_head_params* p, const float* drop, float p_drop, uint64_t seed, uint64_t* rng_state,
                           float* drop_out, float* zero_buf, int64_t zero_n, float* ul, float* il, float* pred,
                           const float* target, float* loss, float* d_pred_unit, int32_t* ticket, void* stream);

/* d_ufeat/d_ifeat [B,H] overwritten; dense grads overwritten; embedding grads accumulated
 * (rows u_id==pad_u / i_id==pad_i get none: nn.Embedding padding_idx).
 * ws: rbr_pair_head_bwd_ws_floats(B, K) floats (may be 0 / NULL: the current kernel needs no scratch). */
size_t rbr_pair_head_bwd_ws_floats(int32_t B, int32_t K);
int rbr_pair_head_bwd(int32_t B, int32_t H, int32_t K, const float* u_feat, const float* i_feat,
                      const int64_t* u_id, const int64_t* i_id, const rbr_head_params* p, const float* drop,
                      const float* ul, const float* il, const float* d_pred, int32_t pad_u, int32_t pad_i,
                      const rbr_head_grads* g, float* d_ufeat, float* d_ifeat, float* ws, void* stream);
/* ---- nn.Dropout multiplier (deepconn/layers.py:202, narre.py:73, dual_att/dual_att.py:33, simple_siamese/layers.py:7-68):
 *   out[i] = 0 with probability p, else 1/(1-p), i < n.  Philox4x32-10 keyed by `seed`, counter (i/4, call number).
 *   state: 2 x uint64 in device memory, zero-initialised by the caller once; state[0] is the call number, advanced by
 *   the kernel itself (graph-replay safe: every replay draws a new mask), state[1] a workgroup ticket.      ---- */
int rbr_dropout_multiplier(int64_t n, float p, uint64_t seed, uint64_t* state, float* out, void* stream);

/* ---- nn.MSELoss(reduction="mean") of the trainers (trainer/train_deepconn_pp.py:137,164):
 *   loss[0] = sum_i (pred[i]-target[i])^2 / n   (one workgroup, fixed summation order)
 *   d_pred[i] = 2 (pred[i]-target[i]) / n * d_loss[0]       (d_loss is a device scalar)
 *   d_pred_unit (optional, [n]): d loss / d pred for an upstream gradient of exactly 1, written by the forward launch, so
 *   that a backward called with that gradient (loss.backward()) needs no launch of its own.
 */
int rbr_mse_loss_fwd(int64_t n, const float* pred, const float* target, float* loss, float* d_pred_unit, void* stream);
/* D-ATT's rating (dual_att.py:58): out[b] = <x[b,:], x[B+b,:]> over the stacked [2B, K] output of the shared fc (user rows
 * first); backward: d_x[b,:] = d_out[b] * x[B+b,:], d_x[B+b,:] = d_out[b] * x[b,:] (d_x OVERWRITTEN). */
int rbr_pair_dot_fwd(int32_t B, int32_t K, const float* x, float* out, void* stream);
int rbr_pair_dot_bwd(int32_t B, int32_t K, const float* x, const float* d_out, float* d_x, void* stream);
/* out [2B, C1+C2] = [[a, b], [c, d]] (a, c [B,C1]; b, d [B,C2]) and its inverse: D-ATT's cat(local, global) per tower stacked
 * user-over-item for the shared fc (dual_att.py:50-57), as one launch each way. */
int rbr_block_cat(int32_t B, int32_t C1, int32_t C2, const float* a, const float* b, const float* c, const float* d, float* out,
                  void* stream);
int rbr_block_split(int32_t B, int32_t C1, int32_t C2, const float* g, float* a, float* b, float* c, float* d, void* stream);
int rbr_mse_loss_bwd(int64_t n, const float* pred, const float* target, const float* d_loss, float* d_pred,
                     void* stream);

/* ---- NARRE review-level attention pool (narre.py:40-64)
 *   e = ebd[other_id];  logit = relu(feat@W_rv + e@W_id + b1) @ h + b2
 *   att = exp(logit) / (sum_R exp(logit) + 1e-8)   (unmasked, no max subtraction)
 *   out = sum_R att * feat
 * feat [B,R,H], other_id [B,R]; out [B,H], att [B,R]; hid [B,R,A] saved for backward.       ---- */
typedef struct rbr_attn_params {
    const float* W_rv; const float* W_id; const float* h; const float* b1; const float* b2; const float* ebd;
} rbr_attn_params;                                       /* [H,A] [A,A] [A] [A] [1] [N,A] */

typedef struct rbr_attn_grads {
    float* dW_rv; float* dW_id; float* dh; float* db1; float* db2; float* debd; /* debd ACCUMULATED */
} rbr_attn_grads;

/* `drop` [B,H] or NULL: the multiplier of the nn.Dropout that follows the pooled feature (narre.py:62), applied to `out` in
 * the forward and to d_out in the backward (pass the same tensor to both). */
int rbr_review_attn_fwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                        const rbr_attn_params* p, const float* drop, float* out, float* att, float* hid, void* stream);
size_t rbr_review_attn_bwd_ws_floats(int32_t B, int32_t R, int32_t H, int32_t A);
int rbr_review_attn_bwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                        const rbr_attn_params* p, const float* drop, const float* att, const float* hid, const float* d_out,
                        const float* d_att, int32_t pad_idx, const rbr_attn_grads* g, float* d_feat, float* ws,
                        void* stream);
/* Both attention pools of a two-tower model (NARRE: user_att keyed by item ids, item_att by user ids, narre.py:177-178) in one
 * launch per stage.  Every per-side tensor is a stacked block, side 0 first: feat [2,B,R,H], other_id [2,B,R], drop [2,B,H] or
 * NULL, out [2,B,H], att [2,B,R], hid [2,B,R,A], d_out [2,B,H], d_att [2,B,R] or NULL, d_feat [2,B,R,H]; p0/p1, g0/g1 the sides'
 * parameters and gradients.  The backward ZEROES g0->debd [n_ebd0, A] and g1->debd [n_ebd1, A] itself before accumulating.
 * ws: 2 * rbr_review_attn_bwd_ws_floats(B, R, H, A) floats. */
int rbr_review_attn2_fwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                         const rbr_attn_params* p0, const rbr_attn_params* p1, const float* drop, float* out, float* att,
                         float* hid, void* stream);
int rbr_review_attn2_bwd(int32_t B, int32_t R, int32_t H, int32_t A, const float* feat, const int64_t* other_id,
                         const rbr_attn_params* p0, const rbr_attn_params* p1, const float* drop, const float* att,
                         const float* hid, const float* d_out, const float* d_att, int32_t pad_idx0, int32_t pad_idx1,
                         const rbr_attn_grads* g0, const rbr_attn_grads* g1, int64_t n_ebd0, int64_t n_ebd1, float* d_feat,
                         float* ws, void* stream);

/* ---- D-ATT gates (dual_att/layers.py:34-36,50 and 65-67,84)
 * local : gate[b,l] = sigmoid(b0 + sum_{j<win} sum_e w[e,j] * x[b, l+j-(win-1)/2, e])   (zero padded)
 * global: gate[b,l] = sigmoid(b0 + sum_l sum_e w[e,l] * x[b,l,e])  (one scalar per doc, broadcast over l)
 * x = table[ids];  w is the Conv1d weight [1,E,win] / [1,E,L].                               ---- */
int rbr_datt_local_gate_fwd(int32_t B, int32_t L, int32_t E, int32_t win, const int64_t* ids, const float* table,
                            const float* w, const float* b0, float* gate, void* stream);
int rbr_datt_global_gate_fwd(int32_t B, int32_t L, int32_t E, const int64_t* ids, const float* table,
                             const float* w, const float* b0, float* gate, void* stream);
/* Backward of a gate from dgate[b,l] (for the global gate: summed over l inside).
 * dw, db0 overwritten; dtable accumulated (row pad_idx excluded). ws: B*(E*win) / B*... floats, see .hip */
size_t rbr_datt_gate_bwd_ws_floats(int32_t B, int32_t L, int32_t E, int32_t win, int32_t is_global);
/* Token-product form of the local gate (same results): S[token][j] = <table[token], w[:,j]> for the DISTINCT tokens of
 * ids (table of V rows), the gate is a gather of `win` scalars per position; the backward folds dpre into win tap sums
 * per token and OVERWRITES the whole dtable [V,E] (absent tokens and row pad_idx: 0), dw [E*win] and db0.  `ws`:
 * rbr_datt_local_gate_prod_ws_bytes bytes (0: not worthwhile / unsupported -> use the functions above), the SAME buffer,
 * untouched, for the forward and its backward.  win odd, <= 8.  `rows`: the tower's shared distinct-token maps
 * (rbr_datt_token_rows, the same pointer for the forward and the backward) or NULL (the call builds private ones). */
size_t rbr_datt_local_gate_prod_ws_bytes(int32_t B, int32_t L, int32_t E, int32_t win, int32_t V);
int rbr_datt_local_gate_fwd_prod(int32_t B, int32_t L, int32_t E, int32_t win, int32_t V, const int64_t* ids, const float* table,
                                 const float* w, const float* b0, float* gate, void* ws, const void* rows, void* stream);
int rbr_datt_local_gate_bwd_prod(int32_t B, int32_t L, int32_t E, int32_t win, int32_t V, const int64_t* ids, const float* table,
                                 const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw, float* db0,
                                 float* dtable, void* ws, const void* rows, int32_t accumulate, void* stream);
int rbr_datt_local_gate_bwd(int32_t B, int32_t L, int32_t E, int32_t win, const int64_t* ids, const float* table,
                            const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw,
                            float* db0, float* dtable, float* ws, void* stream);
int rbr_datt_global_gate_bwd(int32_t B, int32_t L, int32_t E, const int64_t* ids, const float* table,
                             const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw,
                             float* db0, float* dtable, float* ws, void* stream);
/* Distinct-token rows of one tower's documents (ids [B,L] over a table of V rows): row_of_token / tok_of_row maps in `rows`
 * (rbr_datt_token_rows_ws_bytes bytes), built once per tower and step and handed to the gate backwards below.
 * rbr_datt_global_gate_bwd_rows == rbr_datt_global_gate_bwd, except that the table gradient goes through the occurrence
 * matrix A[row, p] = sum of dpre over the documents that carry the row's token at position p (one scalar atomic per position
 * instead of a row of E per distinct token and window), every table row is then written once from its non-zeros, and
 * dtable [V,E] is OVERWRITTEN (absent tokens and pad_idx: 0).
 * ws: rbr_datt_global_gate_bwd_rows_ws_floats floats (0: E > 256 or L % 4 != 0 -> use rbr_datt_global_gate_bwd).
 * `accumulate` != 0 (here and in rbr_datt_local_gate_bwd_prod): the rows of the batch's tokens are ADDED to dtable -- a gradient
 * buffer shared by the producers of one step, which run one after the other on one stream -- and nothing else is written.
 * rbr_datt_global_gate_bwd_rows in two calls (the phases on different streams): first with dtable == NULL (dw, db0; dpre stays
 * in `ws`), then with accumulate | 2 and dtable (the table rows alone, from the dpre in `ws`; table / gate / dgate / dw / db0
 * are not read and may be NULL). */
size_t rbr_datt_token_rows_ws_bytes(int32_t B, int32_t L, int32_t V);
int rbr_datt_token_rows(int32_t B, int32_t L, int32_t V, const int64_t* ids, void* rows, void* stream);
size_t rbr_datt_global_gate_bwd_rows_ws_floats(int32_t B, int32_t L, int32_t E, int32_t V);
int rbr_datt_global_gate_bwd_rows(int32_t B, int32_t L, int32_t E, int32_t V, const int64_t* ids, const float* table,
                                  const float* w, const float* gate, const float* dgate, int32_t pad_idx, float* dw,
                                  float* db0, float* dtable, float* ws, const void* rows, int32_t accumulate, void* stream);

/* ---- nn.Linear (+ReLU / Tanh, + dropout multiplier) on the f32 MFMA pipe: y = act(x @ W^T + b) * drop
 *      (`relu`: 0 none, 1 ReLU, 2 Tanh -- SimpleSiamese's latent_transform_layer, simple_siamese.py:24-26)
 *      replaces D-ATT's shared fc (dual_att/dual_att.py:31-35,51,57) and HierPooling's projection
 *      (deepconn/layers.py:76-79,96).  x [N,IN], W [OUT,IN] (torch layout), b [OUT] or NULL,
 *      drop [N,OUT] multiplier or NULL, y [N,OUT].
 *      Backward: d_x [N,IN] (NULL to skip), dW [OUT,IN], db [OUT] (NULL to skip) overwritten;
 *      ws: rbr_linear_bwd_ws_floats(N, OUT) floats.                                           ---- */
int rbr_linear_fwd(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* b, int32_t relu,
                   const float* drop, float* y, void* stream);
size_t rbr_linear_bwd_ws_floats(int32_t N, int32_t OUT);
int rbr_linear_bwd(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* y, const float* d_y,
                   int32_t relu, const float* drop, float* d_x, float* dW, float* db, float* ws, void* stream);
/* The same layer with its products SPLIT ALONG K over several workgroups per output tile (D-ATT's shared fc, dual_att.py:31-35:
 * 1024 x 500 x 500 is 128 output tiles on 256 CUs, its weight gradients 64 and 8): every slice writes a partial tile to `ws`, a
 * second launch adds the partial tiles in slice order and applies bias / activation / dropout.  Same results as the calls above
 * up to the summation order over K; run to run the same bits.
 *   ws: rbr_linear_fwd_ws_floats(N, IN, OUT) floats (0 = the shape is not split: ws may be NULL) /
 *       rbr_linear_bwd_ex_ws_floats(N, IN, OUT) floats (covers rbr_linear_bwd_ws_floats). */
size_t rbr_linear_fwd_ws_floats(int32_t N, int32_t IN, int32_t OUT);
size_t rbr_linear_bwd_ex_ws_floats(int32_t N, int32_t IN, int32_t OUT);
int rbr_linear_fwd_ex(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* b, int32_t relu,
                      const float* drop, float* y, float* ws, void* stream);
int rbr_linear_bwd_ex(int32_t N, int32_t IN, int32_t OUT, const float* x, const float* W, const float* y, const float* d_y,
                      int32_t relu, const float* drop, float* d_x, float* dW, float* db, float* ws, void* stream);

/* ---- MyConv1d.forward (deepconn/layers.py:46-60; dup narre/layers.py:139-153) on materialised inputs: the contraction is an
 *      rbr_linear_* product T [bz * L, sum kz*ch] = x @ Wprod^T (column (w, j, c) = poff[w] + j * ch[w] + c); these two calls are its
 *      epilogue -- 'same' zero padding, the kz shifted adds, the bias and the channel concatenation, written in the reference's
 *      N x C x L layout -- and the epilogue's backward (dT overwritten, dbias[w] [ch[w]] overwritten).  kz odd.              ---- */
int rbr_conv_shift_add_fwd(int32_t bz, int32_t L, int32_t n_widths, const int32_t* kz, const int32_t* ch, const float* T,
                           const float* const* bias, float* out, void* stream);
int rbr_conv_shift_add_bwd(int32_t bz, int32_t L, int32_t n_widths, const int32_t* kz, const int32_t* ch, const float* d_out,
                           float* dT, float* const* dbias, void* stream);

/* ---- standalone word-embedding row gather / scatter-add (WordEmbedding.forward, deepconn/layers.py:22-24).
 *      The models never call these (the gather is fused into the conv kernel); they serve callers that
 *      want the materialised rows.  out [n_tok, D];  dtable ACCUMULATED, row pad_idx excluded.   ---- */
int rbr_embedding_fwd(int64_t n_tok, int32_t D, const int64_t* ids, const float* table, float* out, void* stream);
int rbr_embedding_bwd(int64_t n_tok, int32_t D, const int64_t* ids, const float* d_out, int32_t pad_idx, float* dtable,
                      void* stream);

/* ---- id range check in front of every embedding-style lookup (nn.Embedding raises IndexError, deepconn/layers.py:23;
 *      a kernel cannot, and an unchecked id would read or -- in the backward -- write outside its table).
 *      ONE launch for up to RBR_MAX_ID_SETS tensors: out[k] = in[k] when 0 <= in[k] < limit, else `replace` (a valid
 *      row, 0 <= replace < limit) and the device record err (int64[4], zero while clean: err[0] = bad ids so far,
 *      err[1] = one offending value, err[2] = its set) is updated; the host reads `err` at its next synchronisation
 *      point and raises.  `sets` is a HOST array; the outputs may be adjacent slices of one buffer (stacks the towers). ---- */
#define RBR_MAX_ID_SETS 8
typedef struct rbr_id_set {
    const int64_t* in;
    int64_t* out;
    int64_t n, limit, replace;
} rbr_id_set;
int rbr_sanitize_ids(int32_t n_sets, const rbr_id_set* sets, int64_t* err, void* stream);

/* ---- in-batch document dedup (the reference re-encodes a document for every pair it appears in,
 *      deepconn/deepconn.py:46-47).  Rows [0,B) = user side (ids u_ids, table of U ids), rows [B,2B) = item side.
 *      first[r] = first row of the same side with the same id (item rows offset by B); mask_out[r,:] = mask_in[r,:]
 *      (all ones when mask_in is NULL) for rows that are their own first occurrence, 0 otherwise -- the encoder then skips
 *      the repeated documents and the caller gathers their features from row first[r].  No sync, static shapes.
 *      ws: rbr_dedup_ws_bytes(U, I) bytes.  Ids outside their table count as unique.                           ---- */
size_t rbr_dedup_ws_bytes(int32_t U, int32_t I);
int rbr_dedup_rows(int32_t B, int32_t L, const int64_t* u_ids, const int64_t* i_ids, int32_t U, int32_t I,
                   const uint8_t* mask_in, void* ws, int64_t* first, uint8_t* mask_out, void* stream);
/* Backward counterpart: d_rows [n_rows, H] (the gradient of the per-document features) -- every repeated row r (first[r] != r) is
 * ADDED onto row first[r] and cleared, so the encoder's backward, which skips the blanked documents, sees the whole gradient. */
int rbr_dedup_fold_rows(int32_t n_rows, int32_t H, const int64_t* first, float* d_rows, void* stream);

/* ---- NgramFeat arch="HierPooling" (deepconn/layers.py:62-98,110-114): pooled[doc,d] =
 *      max_l mean_{j<k} x[doc,l+j,d] over l in [0, L-k], x = mask * table[ids]; relu != 0 applies the
 *      trailing ReLU when there is no projection layer.  argmax[doc,d] = first maximising window start.
 *      Backward: dtable ACCUMULATED (each of the k rows of the winning window gets g/k).          ---- */
int rbr_hier_pool_fwd(int32_t n_docs, int32_t L, int32_t D, int32_t k, const int64_t* ids, const uint8_t* mask,
                      const float* table, int32_t relu, float* pooled, int32_t* argmax, void* stream);
int rbr_hier_pool_bwd(int32_t n_docs, int32_t L, int32_t D, int32_t k, const int64_t* ids, const uint8_t* mask,
                      const int32_t* argmax, const float* pooled, const float* d_pooled, int32_t relu, int32_t pad_idx,
                      float* dtable, void* stream);

/* ---- SimpleSiamese encoder (models/simple_siamese/simple_siamese.py:57-74).
 *      review_bag: WordEmbedding -> VariationalDropout -> MaskedAvgPooling1d (layers.py:24-50,53-68,90-110) in one pass:
 *        out[r, :] = drop[r, :] * sum_l mask[r,l] * table[ids[r,l], :] / (sum_l mask[r,l] + 1e-8)
 *        ids [n_rev, T] int64, mask [n_rev, T] or NULL, drop [n_rev, D] multiplier or NULL, out [n_rev, D],
 *        inv_len [n_rev] (kept for the backward).  Backward: dtable [V, D] ACCUMULATED, row padding_idx excluded;
 *        the occurrences are sorted by token first (ws: rbr_review_bag_bwd_ws_bytes(n_rev, T) bytes).
 *      additive_attn: NodeDropout + AddictiveAttention (layers.py:7-22,171-197) for B users / items with R reviews each:
 *        x = node_drop[b,r] * rev[b,r,:];  t = tanh(x Wp^T + bp);  s = softmax_r(masked_fill(<t, wi>, ~mask, -1e8));
 *        out[b,:] = sum_r s[r] x[r,:].   rev [B,R,H], mask [B,R] or NULL, node_drop [B,R] or NULL, Wp [K,H], bp [K],
 *        wi [K]; scores [B,R] and t_out [B,R,K] are kept for the backward, which overwrites d_rev, d_Wp, d_bp, d_wi.
 *        R <= 64; ws: rbr_additive_attn_bwd_ws_floats(B,R,H,K) floats.                                     ---- */
int rbr_review_bag_fwd(int32_t n_rev, int32_t T, int32_t D, const int64_t* ids, const uint8_t* mask, const float* table,
                       const float* drop, float* out, float* inv_len, void* stream);
size_t rbr_review_bag_bwd_ws_bytes(int32_t n_rev, int32_t T);
int rbr_review_bag_bwd(int32_t n_rev, int32_t T, int32_t D, int32_t V, const int64_t* ids, const uint8_t* mask, const float* drop,
                       const float* inv_len, const float* d_out, int32_t padding_idx, float* dtable, void* ws, void* stream);
int rbr_additive_attn_fwd(int32_t B, int32_t R, int32_t H, int32_t K, const float* rev, const uint8_t* mask,
                          const float* node_drop, const float* Wp, const float* bp, const float* wi, float* out, float* scores,
                          float* t_out, void* stream);
size_t rbr_additive_attn_bwd_ws_floats(int32_t B, int32_t R, int32_t H, int32_t K);
int rbr_additive_attn_bwd(int32_t B, int32_t R, int32_t H, int32_t K, const float* rev, const uint8_t* mask,
                          const float* node_drop, const float* Wp, const float* wi, const float* scores, const float* t_in,
                          const float* d_out, float* d_rev, float* d_Wp, float* d_bp, float* d_wi, float* ws, void* stream);

/* ---- clip_grad_norm_(params, max_norm) followed by torch.optim.Adam.step()  (trainer/train_deepconn_pp.py:166-167;
 *      optimizer of :135: lr only, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad), two launches over all
 *      tensors.  params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors device pointers (fp32, contiguous,
 *      numel[k] elements each).  `step`: device float holding the number of steps taken so far; it is advanced by one
 *      and the bias corrections use the new value.  max_norm <= 0 skips the clipping.  The gradients are left clipped
 *      (as clip_grad_norm_ leaves them), *gnorm_out (device, may be NULL) receives the norm before clipping.
 *      ws: rbr_clip_adam_ws_floats() floats.                                                             ---- */
#define RBR_OPT_MAX_TENSORS 64
size_t rbr_clip_adam_ws_floats(void);
int rbr_clip_adam_step(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr, float beta1, float beta2,
                       float eps, float* step, float* gnorm_out, float* ws, void* stream);
/* The same step when ONE of the tensors is an embedding table [V, D] whose gradient arrives in compact row form (the
 * token-product conv's backward with RBR_G_ROWS): g[v, :] = rows[row_of_token[v], :] for the tokens of the batch, exactly 0 for
 * every other row -- the update formula is unchanged (so parameters and Adam state come out as with the dense gradient
 * nn.Embedding's backward builds, deepconn/layers.py:22-24 + trainer/train_deepconn_pp.py:166-167, bit for bit), but the
 * zero rows are neither read for the norm, nor read for the update, nor written back when the clip scales the gradient.
 * grads[rg->tensor] is ignored (may be NULL); `rows` is left clipped.  D % 4 == 0, V * D < 2^32, 16-byte aligned pointers. */
typedef struct rbr_row_grad {
    int32_t tensor;                 /* index of the table among the n_tensors */
    int32_t V, D;
    const int32_t* row_of_token;    /* [V] list row or -1 */
    float* rows;                    /* [list rows, D] */
    const float* sq_part;           /* [n_sq] partial sums of squares of `rows` */
    int32_t n_sq;
} rbr_row_grad;
int rbr_clip_adam_step_rows(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr, float beta1, float beta2,
                            float eps, float* step, float* gnorm_out, float* ws, const rbr_row_grad* rg, void* stream);
/* dense[v, :] = rows[row_of_token[v], :] for listed tokens, 0 elsewhere: the [V, D] gradient for consumers outside the fused step */
int rbr_row_grad_to_dense(int32_t V, int32_t D, const int32_t* row_of_token, const float* rows, float* dense, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RBR_HIP_H */
