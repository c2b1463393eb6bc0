/*
 * magic_hip.h -- C ABI of libmagic_hip.so, the MI355X (gfx950) implementation of the
 * MAGIC masked-GAT auto-encoder hot path and KNN outlier scorer.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless a comment says "host";
 *   - every function returns 0 (MAGIC_OK) or a negative magic_status; none throws,
 *     exits or synchronises the device; work is enqueued on `stream`
 *     (a hipStream_t passed as void*, NULL = the default stream);
 *   - the library allocates nothing: scratch comes in through `workspace`, whose size the
 *     matching *_workspace_bytes() function reports;
 *   - "CSC order" = edges sorted by destination node, ties by ascending original edge id;
 *     "CSR order" = sorted by source node, same tie rule;
 *   - row strides (`ld_*`) are in elements, not bytes.
 *
 * The reference has no FFI: its seam is the Python API of model/gat.py,
 * model/autoencoder.py and model/eval.py plus the DGL / scikit-learn calls those files
 * make.  Each entry point names the reference lines (relative to the reference tree) whose
 * third-party call it replaces.
 */
#ifndef MAGIC_HIP_H
#define MAGIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum magic_status {
    MAGIC_OK = 0,
    MAGIC_ERR_INVALID_ARG = -1,   /* null pointer, negative size, inconsistent sizes */
    MAGIC_ERR_UNSUPPORTED = -2,   /* head count / head width without a compiled kernel */
    MAGIC_ERR_WORKSPACE = -3,     /* workspace missing or too small */
    MAGIC_ERR_LAUNCH = -4,        /* hipGetLastError() reported a launch failure */
    MAGIC_ERR_RUNTIME = -5,       /* another HIP runtime call failed */
    MAGIC_ERR_STATIC_LIMIT = -6   /* input larger than the static scratch of a no-allocation entry point */
} magic_status;

/* Library ABI version (bumped on any signature change). */
int magic_abi_version(void);
/* Static, human readable text for a status code. */
const char *magic_status_string(int status);

/* ------------------------------------------------------------------------------------
 * Per-kernel timing (HIP events on the launch stream).  While enabled, every kernel the
 * library launches is bracketed by two events; magic_prof_read() synchronises those
 * events and returns, per kernel name, the launch count and the summed duration.
 * Has no counterpart in the reference (which has no timers: model/eval.py:7 imports
 * `time` and never uses it).
 * ---------------------------------------------------------------------------------- */
int magic_prof_enable(int on);
int magic_prof_reset(void);
/* host buffers; returns number of distinct kernels written (<= max_entries) or <0 */
int magic_prof_read(int max_entries, char *names /* max_entries x 64 bytes */,
                    int64_t *launches, double *total_ms);
/* the names of the bracketed launches in launch order, oldest first (host buffer); returns the number of
 * launches recorded since the last reset (which may exceed max_entries: only the first max_entries are written) */
int magic_prof_sequence(int max_entries, char *names /* max_entries x 64 bytes */);

/* ------------------------------------------------------------------------------------
 * CSC + CSR construction: stable counting sort of the COO edge list by destination and
 * by source.  Replaces DGL's lazy COO->CSR/CSC conversion behind every message-passing
 * call of the reference (model/gat.py:302-303 apply_edges, :308 edge_softmax,
 * :311-312 update_all) and, through offset concatenation on the host side, dgl.batch
 * (model/train.py:39).  Integer work: results are bit-exact.
 * ---------------------------------------------------------------------------------- */
int magic_csr_build_workspace_bytes(int64_t n_nodes, int64_t n_edges, size_t *bytes /* host */);
int magic_csr_build(int64_t n_nodes, int64_t n_edges,
                    const int64_t *src, const int64_t *dst,            /* [E] COO, DGL's int64 ids */
                    int32_t *in_ptr, int32_t *in_src, int32_t *in_eid,    /* [N+1],[E],[E] CSC */
                    int32_t *out_ptr, int32_t *out_dst, int32_t *out_eid, /* [N+1],[E],[E] CSR */
                    void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Segment work plan shared by the GAT kernels.  Segments (in-edges of one destination in
 * CSC order, out-edges of one source in CSR order) of at most `light_max` edges are
 * reduced by one lane group inside the fused kernels; longer ones ("heavy") are cut into
 * chunks that separate waves reduce, combined in chunk order.
 * ---------------------------------------------------------------------------------- */
typedef struct magic_plan {
    const int32_t *ptr;        /* [N+1] segment offsets */
    int32_t light_max;         /* segments longer than this are heavy */
    int32_t n_heavy;           /* number of heavy nodes */
    const int32_t *heavy_node; /* [n_heavy] node id */
    const int32_t *heavy_cptr; /* [n_heavy+1] first chunk of each heavy node */
    int32_t n_chunks;
    const int32_t *chunk_node; /* [n_chunks] heavy slot (index into heavy_node) */
    const int32_t *chunk_beg;  /* [n_chunks] first edge of the chunk */
    const int32_t *chunk_end;  /* [n_chunks] one past the last edge */
    const int32_t *node_order; /* [N] walk order of the node-driven kernels, or NULL for 0..N-1: node ids
                                  sorted by (XCD range, segment-length class), stable, so that the lane
                                  groups of one wavefront hold segments of equal length (results do not
                                  depend on it) */
    const int32_t *xcd_start;  /* [9] or NULL: positions that cut node_order (and, being monotone in the
                                  node id, the id range 0..N-1 as well) into 8 contiguous ranges of equal
                                  work (edges + 2 per node).  Workgroups b, b+8, b+16, ... share one XCD and
                                  its L2; they walk range b % 8, so an XCD's gathers stay inside the rows
                                  its own node range touches (whole component graphs of a batch) */
} magic_plan;

/* ------------------------------------------------------------------------------------
 * Dense node projection with fused attention scores.
 *   Y[n, :] = X[n, :] . W^T                                   X [N,F], W [M,F] row-major
 * Columns of Y are split into up to three outputs so that the gathered operand keeps
 * 128-byte aligned rows:  cols [0,C) -> hs [N,C];  cols [C, C+n_score) -> score [N,n_score]
 * (eh then et, model/gat.py:291-292, pre-folded into W by the caller);  cols
 * [C+n_score, M) -> res [N, M-C-n_score] (residual projection, model/gat.py:323).
 * Replaces torch.nn.Linear at model/gat.py:284 and :323.  fp32 in, fp32 accumulate, fp32-grade
 * products: the main kernel splits each operand into three bf16 pieces and keeps the six
 * leading cross products (v_mfma_f32_32x32x16_bf16; what it drops is below one fp32 rounding
 * of the product), the fallback kernels use v_mfma_f32_32x32x2_f32.  Fixed summation order:
 * results are reproducible run to run.
 * ---------------------------------------------------------------------------------- */
int magic_proj_fwd(int64_t n_rows, int32_t in_dim, int32_t out_c, int32_t n_score, int32_t res_c,
                   const float *x, int64_t ld_x,
                   const uint8_t *x_mask /* [N,in_dim] dropout keep mask applied to x, or NULL */, float x_scale,
                   const float *w /* [C+n_score+res_c, in_dim] */,
                   float *hs, int64_t ld_hs, float *score, float *res, int64_t ld_res,
                   void *workspace /* may be NULL: the wide-layer kernel keeps the split weight there */,
                   size_t workspace_bytes, void *stream);
size_t magic_proj_fwd_workspace_bytes(int32_t in_dim, int32_t out_total /* out_c + n_score + res_c */);

/* Backward of magic_proj_fwd (the adjoints torch autograd forms for nn.Linear at
 * model/gat.py:284,:323 and model/autoencoder.py:237, and for nn.Dropout at model/gat.py:283
 * and nn.PReLU at :335 when they are fused in):
 *   magic_proj_bwd_input : dX[N,out_dim] = epilogue( dY[N,k_dim] . W[k_dim,out_dim] )
 *   magic_proj_bwd_weight: dW[M,F]       = dY[N,M]^T . (X[N,F] (.) x_mask * x_scale)
 *                          (fixed-order reduction over N)
 * The epilogue (all parts optional, applied in this order to every output element):
 *   v += add[row,col];  v = out_mask[row,col] ? v*out_scale : 0;  v += add2[row,col];
 *   PReLU adjoint of the layer that produced the projected input: with pre the
 *   pre-activation (recovered from act_out when *prelu_w > 0, read from act_pre otherwise),
 *   pre <= 0 ?  (prelu_grad += v*pre, v *= *prelu_w) : nothing;
 *   col_sum[col] += v  (bias gradient of that layer).
 */
typedef struct magic_proj_epilogue {
    const float *add;        int64_t ld_add;
    const uint8_t *out_mask; float out_scale;      /* [N,out_dim] contiguous */
    const float *add2;       int64_t ld_add2;
    const float *act_out;    int64_t ld_act;       /* forward output of the producing layer */
    const float *act_pre;                          /* [N,out_dim] its pre-activation */
    const float *prelu_w;                          /* device scalar */
    float *col_sum;                                /* [out_dim] or NULL */
    float *prelu_grad;                             /* [1] or NULL */
} magic_proj_epilogue;

/* magic_proj_fwd with the layer's attention vectors given as well (attn_src = attn_h, attn_dst = attn_t of
 * model/gat.py:291-292, flattened to [out_c], 16-byte aligned): for wide layers with 64-wide heads and no
 * residual projection the score columns are then formed as the reference forms them, (feat * attn).sum(-1)
 * over the head's outputs, in the epilogue of the block-tiled kernel -- its 2H extra output columns would cost
 * a third, nearly idle column block.  w is still the full augmented weight (its folded score rows are used
 * whenever this shortcut does not apply); same results to fp32 rounding. */
int magic_proj_fwd_attn(int64_t n_rows, int32_t in_dim, int32_t out_c, int32_t n_score, int32_t res_c,
                        const float *x, int64_t ld_x, const uint8_t *x_mask, float x_scale, const float *w,
                        float *hs, int64_t ld_hs, float *score, float *res, int64_t ld_res,
                        const float *attn_src, const float *attn_dst,
                        void *workspace, size_t workspace_bytes, void *stream);
size_t magic_proj_bwd_input_workspace_bytes(int64_t n_rows, int32_t k_dim, int32_t out_dim);
int magic_proj_bwd_input(int64_t n_rows, int32_t k_dim, int32_t out_dim,
                         const float *dy, int64_t ld_dy, const float *w /* [k_dim,out_dim] */,
                         const magic_proj_epilogue *epilogue /* host struct, may be NULL */,
                         float *dx, int64_t ld_dx,
                         void *workspace, size_t workspace_bytes, void *stream);
size_t magic_proj_bwd_weight_workspace_bytes(int64_t n_rows, int32_t m, int32_t f);
int magic_proj_bwd_weight(int64_t n_rows, int32_t m, int32_t f,
                          const float *dy, int64_t ld_dy, const float *x, int64_t ld_x,
                          const uint8_t *x_mask /* [N,f] or NULL */, float x_scale,
                          float *dw /* [m,f] */, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * One edge-feature-aware GAT layer, edge phase (model/gat.py:302-335):
 *   logit[e,h] = leaky_relu(eh[src e,h] + ee[e,h] + et[dst e,h])           (:302-307)
 *   a[e,h]     = softmax over the in-edges of dst e                        (:308)
 *   agg[v,h,:] = sum_{e: dst e = v} a[e,h] * hs[src e,h,:]                 (:311-312)
 *   pre[v]     = agg[v] + bias + res[v]                                    (:316-324)
 *   out[v]     = pre >= 0 ? pre : prelu_w * pre      (when prelu_w != NULL) (:334-335)
 * ee comes either from a per-edge-type table (`ee_table[e_dim,H]` + `etype_csc[E]`; edge
 * attributes are one-hot, utils/loaddata.py:259) or from a dense `ee_csc[E,H]` array.
 * H*D must be one of the compiled shapes (D in {4,8,16,32,64}, H*D/4 a power of two
 * <= 64); callers zero-pad narrower heads.
 * Outputs: out [N,H*D]; pre [N,H*D] (optional, needed by backward when prelu_w is given);
 * attn [E,H] (optional; the second result of GATConv.forward(get_attention=True),
 * model/gat.py:337-338) in CSC order with the SIGN BIT set where the pre-activation logit was
 * negative (|attn| is the attention weight);
 * stat [N,H,2] (optional, what magic_gat_bwd needs): per destination and head the pair
 * (et[v,h], max + log sum exp of the leaky-ReLU logits of v's in-edges), from which backward
 * recomputes a[e,h] = exp(leaky_relu(logit) - stat[v,h,1]).  Nodes without in-edges get (et, 0).
 * The logits are formed inside the sweeps (staged in LDS per destination); only the edges of
 * segments longer than the plan's light_max (at most 64) pass through the workspace.  in_dst is
 * not read (the destination of a CSC segment is its node) and may be NULL.
 * ---------------------------------------------------------------------------------- */
size_t magic_gat_fwd_workspace_bytes(int32_t n_edges, int32_t n_chunks, int32_t n_heads, int32_t head_dim);
int magic_gat_fwd(int32_t n_nodes, int32_t n_edges, int32_t n_heads, int32_t head_dim,
                  const magic_plan *in_plan /* host struct of device pointers */,
                  const int32_t *in_src, const int32_t *in_dst /* [E] CSC order */,
                  const float *hs, int64_t ld_hs,
                  const float *eh, const float *et, int64_t ld_score /* row stride of eh and et */,
                  const float *ee_table, const int32_t *etype_csc, const float *ee_csc,
                  float negative_slope, const float *bias, const float *res, int64_t ld_res,
                  const uint8_t *res_mask /* [N,H*D] dropout keep mask of res, or NULL */, float res_scale,
                  const float *prelu_w,
                  float *out, int64_t ld_out, float *pre, float *attn, float *stat,
                  void *workspace, size_t workspace_bytes, void *stream);
/* magic_gat_fwd for a layer of which only some output rows are read -- the decoder: the loss takes
 * recon[mask_nodes] alone (model/autoencoder.py:240-245).  dst_active [N] uint8 marks those rows
 * (NULL = all: magic_gat_fwd).  An unmarked destination is treated as a node without in-edges: none
 * of its edges is visited, out = bias (+ res), stat = (.., +inf), so that every attention weight
 * magic_gat_bwd[_active] recomputes from it is an exact zero.  The marked rows are bit for bit those
 * of magic_gat_fwd.  attn must be NULL. */
int magic_gat_fwd_active(int32_t n_nodes, int32_t n_edges, int32_t n_heads, int32_t head_dim,
                         const magic_plan *in_plan, const int32_t *in_src, const int32_t *in_dst,
                         const float *hs, int64_t ld_hs, const float *eh, const float *et, int64_t ld_score,
                         const float *ee_table, const int32_t *etype_csc, const float *ee_csc,
                         float negative_slope, const float *bias, const float *res, int64_t ld_res,
                         const uint8_t *res_mask, float res_scale, const float *prelu_w,
                         float *out, int64_t ld_out, float *pre, float *attn, float *stat,
                         const uint8_t *dst_active, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Coded input of the first encoder layer.  The reference feeds that layer one_hot(node type)
 * (utils/loaddata.py:258) with the masked rows overwritten by the mask token
 * (model/autoencoder.py:201-210), through feature dropout (model/gat.py:283):
 *   x[u, j] = keep_j(u) * scale * (masked(u) ? token[j] : [j == type(u)]).
 * Such a row takes one of n_dim + 1 + 2^n_dim values, so everything the layer derives linearly from
 * it (model/gat.py:284 fc, :291-292 the head scores, :323 res_fc) is one row of a small table:
 * row r < n_dim = a kept one of type r, row n_dim = the dropped one (zeros), row n_dim + 1 + b = the
 * token under keep pattern b (bit j = column j kept); columns [0,C) hs | [C,C+H) eh | [C+H,C+2H) et |
 * [C+2H,2C+2H) residual, i.e. x(r) . W^T for the augmented weight W [M, n_dim].  The *_coded entry
 * points fetch those rows by a 4-byte code word per node (the row index) from the caches instead of
 * gathering rows of materialised [N,C] arrays from HBM: no projection launch, no [N, n_dim] input
 * matrix.  Results equal the dense path up to fp32 rounding.  n_dim <= 13.
 *   magic_coded_table        the table [magic_coded_table_rows(n_dim), ld_table] from W, the token and
 *                            scale = 1 / keep probability (1 without dropout);
 *   magic_node_code          code words from node types [N] int64, the uint8 keep mask [N, n_dim]
 *                            (NULL: keep all) and the masked row ids [n_mask] (distinct);
 *   magic_gat_fwd_coded      magic_gat_fwd for such a layer (4 heads, typed edges, linear residual
 *                            taken from the table, no attention output);
 *   magic_gat_bwd_coded      magic_gat_bwd for it (d_y [N, >= 2C+2H] receives d_hs | d_eh | d_et as
 *                            usual; the caller's d_pre columns are the residual's gradient);
 *   magic_coded_weight_grad  dW [M, ld_dw] (columns [0, n_dim)) of W [M, ld_w] from d_y [N, M], and
 *                            the mask token's gradient d_token [n_dim] (or NULL): sums of d_y rows
 *                            grouped by code word, fixed summation order (token, n_dim and scale of
 *                            `input` are read here; the sweeps read code and table only).
 * ---------------------------------------------------------------------------------- */
#define MAGIC_CODED_MAX_DIM 13
typedef struct magic_coded_input {
    const uint32_t *code; /* [N] table row of every node */
    const float *table;   /* [magic_coded_table_rows(n_dim), ld_table], 16-byte aligned rows */
    int64_t ld_table;
    const float *token;   /* [n_dim] */
    int32_t n_dim;
    float scale;          /* 1 / keep probability (1 without dropout) */
} magic_coded_input;
int64_t magic_coded_table_rows(int32_t n_dim);
int magic_coded_table(int32_t n_dim, int32_t m, const float *w, int64_t ld_w, const float *token, float scale,
                      float *table, int64_t ld_table, void *stream);
int magic_node_code(int64_t n_nodes, int32_t n_dim, const int64_t *node_type, const uint8_t *keep_mask,
                    int64_t n_mask, const int64_t *mask_rows, uint32_t *code, void *stream);
int magic_gat_fwd_coded(int32_t n_nodes, int32_t n_edges, int32_t n_heads, int32_t head_dim,
                        const magic_plan *in_plan, const int32_t *in_src, const magic_coded_input *input,
                        const float *ee_table, const int32_t *etype_csc, float negative_slope,
                        const float *bias, const float *prelu_w, float *out, int64_t ld_out, float *pre,
                        float *stat, void *workspace, size_t workspace_bytes, void *stream);
int magic_gat_bwd_coded(int32_t n_nodes, int32_t n_edges, int32_t n_heads, int32_t head_dim,
                        const magic_plan *in_plan, const int32_t *in_pos, const magic_plan *out_plan,
                        const int32_t *out_dst, const int32_t *out_pos, const magic_coded_input *input,
                        const float *stat, const float *ee_table, const int32_t *etype_csr, int32_t e_dim,
                        float negative_slope, const float *pre, const float *act_out, int64_t ld_act,
                        const float *prelu_w, const float *d_out, int64_t ld_dout, float *d_pre, int64_t ld_dpre,
                        float *d_y, int64_t ld_dy, float *d_ee_table, float *d_bias, float *d_prelu_w,
                        void *workspace, size_t workspace_bytes, void *stream);
size_t magic_coded_weight_grad_workspace_bytes(int32_t m, int32_t n_dim);
int magic_coded_weight_grad(int64_t n_nodes, int32_t m, const magic_coded_input *input, const float *dy,
                            int64_t ld_dy, const float *w, int64_t ld_w, float *dw, int64_t ld_dw,
                            float *d_token, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Adjoint of a row gather y = x[idx] with repeated indices (model/autoencoder.py:257-258:
 * enc_rep[sample_src], enc_rep[sample_dst]):  dst[idx[s], :] += src[s, :], idx in [0, n_rows).
 * (row, position) pairs pass through a stable radix sort over the bits a row index needs; the
 * first position of every run adds the run in position order: no floating-point atomics,
 * bitwise reproducible.  idx must not contain out-of-range values (not checked on the device).
 * ---------------------------------------------------------------------------------- */
size_t magic_rows_scatter_add_workspace_bytes(int64_t n_src, int64_t n_rows);
int magic_rows_scatter_add(int64_t n_src, int64_t n_rows, int32_t width, const int64_t *idx,
                           const float *src, int64_t ld_src, float *dst, int64_t ld_dst,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Keep mask of the feature dropout (model/gat.py:283, nn.Dropout(feat_drop) on the layer
 * input): mask[i] = 1 with probability keep_prob, else 0, from Philox4x32-10 keyed by
 * (seed, i / 16).  The mask is a function of (seed, i) only.  mask must be 16-byte aligned.
 * The consumers (magic_proj_fwd x_mask, magic_gat_fwd res_mask, magic_proj_bwd_* masks)
 * apply it together with the 1/keep_prob scale where the operand is loaded, so the dropped
 * tensor itself is never written.
 * ---------------------------------------------------------------------------------- */
int magic_dropout_mask(int64_t n_elems, float keep_prob, uint64_t seed, uint8_t *mask, void *stream);
/* Same mask, the seed read from device memory when the kernel runs (seed_dev[0]): a launch recorded
 * in a HIP graph draws a new mask at every replay after the caller rewrote that word. */
int magic_dropout_mask_dev(int64_t n_elems, float keep_prob, const uint64_t *seed_dev, uint8_t *mask, void *stream);

/* ------------------------------------------------------------------------------------
 * Operands of magic_proj_fwd / magic_gat_fwd built from the layer's parameters in one launch
 * (and the adjoint in one launch), replacing the element-wise chain torch would run for
 * model/gat.py:284-299 and :323:
 *   w_aug  [H*Dp + 2H (+ H*Dp), F] = [W_fc ; fold(W_fc, attn_h) ; fold(W_fc, attn_t) ; W_res]
 *          with every head padded from D to Dp rows (zeros),
 *          fold(W, a)[h, k] = sum_d a[h, d] W[h*D + d, k]
 *   table  [e_dim, H], table[t, h] = sum_d attn_e[h, d] W_efc[h*D + d, t]
 *   bias_p [H*Dp] head-padded bias (when bias != NULL)
 * magic_fold_bwd maps (d_w_aug, d_table, d_bias_p; any may be NULL = zero) back to the
 * parameter gradients; d_w_res / d_bias may be NULL when the layer has none.
 * ---------------------------------------------------------------------------------- */
int magic_fold_fwd(int32_t n_heads, int32_t head_dim, int32_t head_dim_padded, int32_t in_dim,
                   int32_t e_dim, const float *w_fc, const float *attn_h, const float *attn_t,
                   const float *w_res, const float *w_efc, const float *attn_e, const float *bias,
                   float *w_aug, float *table, float *bias_p, void *stream);
int magic_fold_bwd(int32_t n_heads, int32_t head_dim, int32_t head_dim_padded, int32_t in_dim,
                   int32_t e_dim, const float *w_fc, const float *attn_h, const float *attn_t,
                   const float *w_efc, const float *attn_e, const float *d_w_aug,
                   const float *d_table, const float *d_bias_p, float *d_w_fc, float *d_attn_h,
                   float *d_attn_t, float *d_w_res, float *d_w_efc, float *d_attn_e, float *d_bias,
                   void *stream);

/* The same two maps for every layer of a model in one launch each (a HOST array of n_layers
 * entries; the pointers travel by value in the kernel arguments, 8 layers per launch).  _fwd
 * reads the inputs and writes w_aug / table / bias_p; _bwd reads the inputs and d_w_aug /
 * d_table / d_bias_p (NULL = zero) and writes the d_* outputs (d_w_res / d_bias NULL for a layer
 * without residual projection / bias).  Element for element what the per-layer calls produce. */
typedef struct magic_fold_layer {
    int32_t H, D, Dp, F, e_dim;     /* heads, head width, padded head width, input width, edge types */
    int32_t reserved;
    const float *w_fc, *attn_h, *attn_t, *w_res, *w_efc, *attn_e, *bias;
    float *w_aug, *table, *bias_p;
    const float *d_w_aug, *d_table, *d_bias_p;
    float *d_w_fc, *d_attn_h, *d_attn_t, *d_w_res, *d_w_efc, *d_attn_e, *d_bias;
} magic_fold_layer;
int magic_fold_fwd_multi(int32_t n_layers, const magic_fold_layer *layers, void *stream);
int magic_fold_bwd_multi(int32_t n_layers, const magic_fold_layer *layers, void *stream);

/* ------------------------------------------------------------------------------------
 * Backward of magic_gat_fwd (DGL's EdgeSoftmax / GSpMM / GSDDMM backward + the PReLU,
 * bias and residual adjoints that torch autograd runs for model/gat.py:302-335).
 * Inputs: the forward's hs, eh (the node's score as a source, row stride ld_score), stat, pre and
 * its edge term (ee_table + etype_csr[E] = edge types in CSR order, or the dense ee_csc[E,H] in
 * CSC order); d_out [N,H*D].  in_pos[E] = CSR position of every CSC edge, out_pos[E] = CSC
 * position of every CSR edge (inverse permutations of each other).
 * Outputs:
 *   d_pre  [N,ld_dpre] gradient w.r.t. the pre-activation (= gradient of the residual input)
 *   d_y    [N,ld_dy] cols [0,C) d_hs, [C,C+H) d_eh, [C+H,C+2H) d_et  (C = H*D)
 *   d_ee_table [e_dim,H] (with ee_table)   -- fixed-order reduction, no atomics
 *   d_logit [E,H]   optional, CSC order: gradient w.r.t. the attention logit of every edge and
 *                    head (= gradient w.r.t. ee[e,:] on the dense-ee path)
 *   d_bias [H*D] (optional), d_prelu_w [1] (optional)
 * The source (CSR) sweep gathers one node row per edge, d_pre[dst], and uses it for
 * d_hs[src] += a d_pre[dst] and for t = <hs[src], d_pre[dst]>; a is recomputed from eh[src],
 * the edge term and stat[dst].  The (a, t) pairs cross to destination order once (through
 * in_pos) for delta[v] = sum a t and d_et; d_logit = a (t - delta[dst]) leaky' and d_eh are
 * formed in source order again.  d_pre may alias d_out when prelu_w is NULL (no copy is made).
 * ---------------------------------------------------------------------------------- */
size_t magic_gat_bwd_workspace_bytes(int32_t n_nodes, int32_t n_edges, int32_t n_chunks_in,
                                     int32_t n_chunks_out, int32_t n_heads, int32_t head_dim,
                                     int32_t e_dim);
int magic_gat_bwd(int32_t n_nodes, int32_t n_edges, int32_t n_heads, int32_t head_dim,
                  const magic_plan *in_plan, const int32_t *in_pos,
                  const magic_plan *out_plan, const int32_t *out_dst, const int32_t *out_pos,
                  const float *hs, int64_t ld_hs, const float *eh, int64_t ld_score, const float *stat,
                  const float *ee_table, const int32_t *etype_csr, const float *ee_csc, int32_t e_dim,
                  float negative_slope, const float *pre,
                  const float *act_out, int64_t ld_act /* forward output, used when *prelu_w > 0 */,
                  const float *prelu_w,
                  const float *d_out, int64_t ld_dout,
                  float *d_pre, int64_t ld_dpre, float *d_y, int64_t ld_dy, float *d_ee_table,
                  float *d_logit,
                  float *d_bias, float *d_prelu_w,
                  void *workspace, size_t workspace_bytes, void *stream);
/* magic_gat_bwd when gradient arrives at the marked destinations only (dst_active [N] uint8; the rows
 * of d_out at unmarked destinations must be zero, as the loss gradient of model/autoencoder.py:242-245
 * is outside mask_nodes).  The edges into unmarked destinations contribute exact zeros; the sweeps
 * gather nothing for them (their destination ids are rewritten in the workspace to one unmarked node,
 * whose rows stay in cache) and the destination pass skips unmarked nodes.  Same results as
 * magic_gat_bwd on such a d_out (up to the sign of zeros).  No d_logit output. */
int magic_gat_bwd_active(int32_t n_nodes, int32_t n_edges, int32_t n_heads, int32_t head_dim,
                         const magic_plan *in_plan, const int32_t *in_pos,
                         const magic_plan *out_plan, const int32_t *out_dst, const int32_t *out_pos,
                         const float *hs, int64_t ld_hs, const float *eh, int64_t ld_score, const float *stat,
                         const float *ee_table, const int32_t *etype_csr, const float *ee_csc, int32_t e_dim,
                         float negative_slope, const float *pre, const float *act_out, int64_t ld_act,
                         const float *prelu_w, const float *d_out, int64_t ld_dout,
                         float *d_pre, int64_t ld_dpre, float *d_y, int64_t ld_dy, float *d_ee_table,
                         float *d_bias, float *d_prelu_w, const uint8_t *dst_active,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Masked-node feature reconstruction loss, forward and backward in one pass
 * (model/autoencoder.py:242-245 + model/loss_func.py:28-31):
 *   loss = mean_i (1 - cos(x_rec[mask i], target_i))^alpha,  F.normalize eps = 1e-12.
 * target_i is row mask[i] of `target` (ld_t > 0) or the single row `target` (ld_t == 0:
 * the mask token, which is what DGL's shared-storage clone leaves in g.ndata['attr']).
 * Writes loss[1], d_x [N, ld_x] (zero outside the masked rows, scaled by `grad_scale`)
 * and, when d_target_row != NULL, the summed gradient of the single target row.
 * ---------------------------------------------------------------------------------- */
size_t magic_sce_workspace_bytes(int64_t n_mask, int32_t n_dim);
int magic_sce_fwd_bwd(int64_t n_rows, int32_t n_dim, int64_t n_mask, const int64_t *mask_idx,
                      const float *x, int64_t ld_x, const float *target, int64_t ld_t,
                      float alpha, float grad_scale,
                      float *loss, float *d_x, float *d_target_row,
                      void *workspace, size_t workspace_bytes, void *stream);

/* magic_sce_fwd_bwd without a workspace argument and without any allocation.  Has no
 * counterpart in the reference; it is for callers that cannot allocate device memory or hand
 * over a workspace: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  The per-block partial sums live in a static device array of the library
 * (1024 x (1 + 16) floats = 69,632 bytes per GPU, held until the process ends), so n_dim is
 * limited to MAGIC_SCE_STATIC_MAX_DIM; a wider row returns MAGIC_ERR_STATIC_LIMIT (use
 * magic_sce_fwd_bwd).  Same arguments, results bit-identical to magic_sce_fwd_bwd.  The call
 * runs no hipMalloc / hipFree / event creation; the first call on a GPU resolves that GPU's
 * copy of the array (hipGetSymbolAddress) with the caller's device current.  All calls on one
 * GPU share the array: calls that overlap on one GPU (two streams, two threads) race. */
#define MAGIC_SCE_STATIC_MAX_DIM 16
int magic_sce_fwd_bwd_static(int64_t n_rows, int32_t n_dim, int64_t n_mask, const int64_t *mask_idx,
                             const float *x, int64_t ld_x, const float *target, int64_t ld_t,
                             float alpha, float grad_scale,
                             float *loss, float *d_x, float *d_target_row, void *stream);

/* ------------------------------------------------------------------------------------
 * Graph read-out (utils/poolers.py:49-73, called from model/eval.py:59-68): mean (mean != 0) or
 * sum of the node embeddings feat [N,width].  n_types == 0: over all nodes, out [width]
 * (:50-53).  n_types > 0: per node type t in [0, n_types) (node_type [N] int64, the reference's
 * graph.ndata['type']; ids outside that range belong to no type), the per-type rows
 * concatenated, out [n_types * width], zeros for a type without nodes (:59-72; Unicorn-Wget:
 * 8 types x 256).  The reference's 'max' read-out returns a (values, indices) tuple and is not
 * provided.  One pass over feat, fixed summation order, no host synchronisation (32 types per pair of launches; an empty
 * node set gives NaN for the mean over all nodes, as torch does, and zeros otherwise).
 * ---------------------------------------------------------------------------------- */
size_t magic_readout_workspace_bytes(int64_t n_rows, int32_t width, int32_t n_types);
int magic_readout(int64_t n_rows, int32_t width, const float *feat, int64_t ld_feat,
                  const int64_t *node_type, int32_t n_types, int32_t mean, float *out /* [max(n_types,1)*width] */,
                  void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Union of device-resident component graphs: dgl.batch at model/train.py:39 (node ids and
 * edge positions of component i shift by the node / edge counts of the components before
 * it).  Integer work: results are bit-exact.
 *
 * magic_concat_offset runs a table of jobs {src, dst, len, add}: dst[i] = src[i] + add for
 * i < len (src == NULL: dst[i] = add), all of one element width (4 or 8 bytes); the table
 * (`n_jobs` x magic_concat_job, 32 bytes each) lives in DEVICE memory, `max_len` = the longest
 * job.  One launch concatenates every index array of every component.
 *
 * magic_walk_order: for up to 4 segment arrays ptr[s] ([n_nodes[s] + 1], device) the walk
 * order of the node-driven sweeps -- node ids cut into 8 contiguous ranges of equal work (per
 * node: its edges when the segment has at most `light_max` of them, plus 2), inside a range by
 * ascending segment-length class (0: empty, 1: 1-4 edges, then ceil(log2(length))), ties by
 * node id -- into order[s] [n_nodes[s]] and the 9 range starts into xcd_start[s] (what
 * magic_plan.node_order / xcd_start hold).  ptr / n_nodes / order / xcd_start are HOST arrays
 * of n_arrays entries; every n_nodes[s] > 0.
 * ---------------------------------------------------------------------------------- */
typedef struct magic_concat_job {
    const void *src;
    void *dst;
    int64_t len;
    int64_t add;
} magic_concat_job;
int magic_concat_offset(int32_t elem_bytes, int32_t n_jobs, const void *jobs_device, int64_t max_len, void *stream);
size_t magic_walk_order_workspace_bytes(int64_t total_nodes);
int magic_walk_order(int32_t n_arrays, const int32_t *const *ptr, const int32_t *n_nodes, int32_t light_max,
                     int32_t *const *order, int32_t *const *xcd_start, void *workspace, size_t workspace_bytes,
                     void *stream);

/* ------------------------------------------------------------------------------------
 * Acceptance test of uniformly drawn negative pairs: the rejection step of
 * dgl.sampling.global_uniform_negative_sampling(g, k) as called at model/autoencoder.py:251
 * (pairs that are neither edges of g nor self-loops, sampled without replacement).  Of the
 * n_candidates pairs (cand_src[i], cand_dst[i]) (int64 node ids, device), taken in order,
 * accept[i] = 1 for the first `num_samples` that are no edge, no self-loop and no repeat of an
 * earlier candidate, 0 for all others (also for ids outside [0, n_nodes)).  The graph comes as
 * both segment arrays (out_ptr/out_dst by source, in_ptr/in_src by destination, as
 * magic_csr_build writes them); the shorter segment of a pair is scanned.  Integer work, two
 * launches, nothing read back.  n_candidates <= MAGIC_NEGATIVE_FILTER_MAX.
 * ---------------------------------------------------------------------------------- */
#define MAGIC_NEGATIVE_FILTER_MAX 16384
int magic_negative_filter(int64_t n_nodes, const int32_t *out_ptr, const int32_t *out_dst,
                          const int32_t *in_ptr, const int32_t *in_src, int32_t n_candidates,
                          const int64_t *cand_src, const int64_t *cand_dst, int32_t num_samples,
                          uint8_t *accept /* [n_candidates] */, void *stream);

/* ------------------------------------------------------------------------------------
 * Optimizer step: torch.optim.Adam as the reference builds it (utils/utils.py:24-31,
 * lr / weight_decay from utils/config.py:20-28; weight decay is the L2 term added to the
 * gradient, no amsgrad) over all parameter tensors at once:
 *   t = *step_device + 1;  g = grad + weight_decay * param;
 *   exp_avg += (1 - beta1) (g - exp_avg);  exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g^2;
 *   param -= lr / (1 - beta1^t) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps)
 * and *step_device = t afterwards (int64 in device memory: a step recorded into a HIP graph
 * counts on its own).  The hyper-parameters are doubles and the bias corrections 1 - beta^t and the step size
 * lr / (1 - beta1^t) are formed in double before they are rounded to float (torch forms them from Python
 * floats), the element-wise updates run in float.  `slots` is a HOST array of n_tensors entries (contiguous float32 device
 * tensors); the pointers are passed to the kernels by value, 64 tensors per launch.
 * ---------------------------------------------------------------------------------- */
typedef struct magic_adam_slot {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t n;
} magic_adam_slot;
int magic_adam_step(int32_t n_tensors, const magic_adam_slot *slots, int64_t *step_device, double lr, double beta1,
                    double beta2, double eps, double weight_decay, void *stream);

/* ------------------------------------------------------------------------------------
 * Tail of the sampled edge-reconstruction loss (model/autoencoder.py:257-266 through the
 * head of :114-119): for the first-layer outputs h_pre [n_rows, width] of the pairs
 * (n_pos positive pairs first, then the negative ones)
 *   z = <LeakyReLU(h_pre[r]), w2> + b2;  p = sigmoid(z);
 *   loss = sum_r weight_r * BCE(p_r, r < n_pos) / sum_r weight_r
 * with torch's BCELoss conventions (logarithms clamped at -100); weight_r = 1 for positive
 * pairs and neg_valid[r - n_pos] != 0 for negative ones (neg_valid == NULL: all 1, the plain
 * mean of :262).  _fwd also leaves d loss / d z times the weight sum in d_logit [n_rows] and
 * the weight sum in weight_sum [1]; _bwd turns them, with the incoming gradient grad_out [1]
 * (device), into d_h_pre [n_rows, width], d_w2 [width] and d_b2 [1].  b1 [width] (or NULL) is
 * the first layer's bias when h_pre comes without it: it is added on the fly, and d_b1 [width]
 * (the column sums of d_h_pre) comes out of _bwd.  Fixed summation
 * order.  width % 4 == 0, width <= 1024, 16-byte aligned rows.  Both take the workspace of
 * magic_edge_head_workspace_bytes.
 * ---------------------------------------------------------------------------------- */
size_t magic_edge_head_workspace_bytes(int64_t n_rows, int32_t width);
int magic_edge_head_fwd(int64_t n_rows, int64_t n_pos, int32_t width, const float *h_pre, int64_t ld_h,
                        const float *b1, const float *w2, const float *b2, const uint8_t *neg_valid, float slope,
                        float *loss, float *d_logit, float *weight_sum, void *workspace, size_t workspace_bytes,
                        void *stream);
int magic_edge_head_bwd(int64_t n_rows, int32_t width, const float *h_pre, int64_t ld_h, const float *b1,
                        const float *w2, const float *d_logit, const float *weight_sum, const float *grad_out,
                        float slope, float *d_h_pre, int64_t ld_dh, float *d_b1, float *d_w2, float *d_b2,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Brute-force k nearest neighbours (Euclidean), replacing
 * sklearn.neighbors.NearestNeighbors(...).kneighbors at model/eval.py:154,158 and
 * :306,311.  The result is the exact float64 one -- squared distances |q|^2 - 2 q.t + |t|^2
 * of the float32 inputs evaluated in float64 (as scikit-learn's float32 path does), the k
 * smallest per query ordered by (distance, train index), then sqrt -- found in two stages:
 * a filter sweeps every (query, train row) pair on the bf16 matrix cores with float32
 * operands split into three bf16 pieces (six products per float32-grade product) and keeps a
 * short candidate list per query; the float64 keys of the candidates are then recomputed and
 * an error bound proves, per query, that no dropped row can belong to the k nearest.  Queries
 * whose proof fails, searches with lists too long for the filter's LDS and train sets under
 * 4,096 rows take a full sweep on the f64 matrix cores instead (k = 200 of the cadets
 * evaluation, model/eval.py:289: every query then owns an unsorted candidate buffer of at
 * least 2k + 64 entries in the workspace, which its wavefront sorts whenever it fills; the
 * workspace grows by 12 bytes x buffer size per query and train split).  Either way the reported
 * key of a (query, row) pair comes from one scalar float64 expression, so shard results merge
 * into the single-search result bit for bit.
 * `train_index_offset` is added to every returned index (row offset of this shard of the
 * train set).  magic_knn_merge combines `n_parts` such results (e.g. one per train
 * shard / per GPU) into the global k best, same order rule.
 * ---------------------------------------------------------------------------------- */
size_t magic_knn_workspace_bytes(int64_t n_train, int64_t n_query, int32_t dim, int32_t k);
int magic_knn_topk(int64_t n_train, int64_t n_query, int32_t dim, int32_t k,
                   const float *train, int64_t ld_train, const float *query, int64_t ld_query,
                   int64_t train_index_offset,
                   double *dist /* [Q,k] */, int64_t *idx /* [Q,k] */,
                   void *workspace, size_t workspace_bytes, void *stream);
int magic_knn_merge(int64_t n_query, int32_t k, int32_t n_parts,
                    const double *part_dist /* [n_parts,Q,k] */, const int64_t *part_idx,
                    double *dist, int64_t *idx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGIC_HIP_H */
