/* regtgcn.h -- C ABI of libregtgcn_hip.so, the MI355X (gfx950) implementation of the RegT-GCN
 * forward/backward hot path.
 *
 * The reference (raynbowy23/RegT-GCN) is pure Python: its hot path has no FFI of its own, it
 * reaches the device through torch / torch_geometric operator calls.  The entry points below are
 * therefore the op sites of that path, one C function per site, so that a reference maintainer can
 * bind them with ctypes (INTEGRATION.md shows the stub).  Citations are relative to the reference
 * root.  All pointers are DEVICE pointers unless the name ends in `_host`; all matrices are
 * row-major fp32; indices are int64 on input (the reference's edge_index dtype) and int32 inside
 * prepared operators.  Every function enqueues on `stream` (a hipStream_t passed as void*), never
 * synchronises, allocates nothing, and returns 0 on success or a non-zero code with a message
 * available from regt_last_error().  Buffers are owned by the caller for the duration of the
 * enqueued work.
 */
#ifndef REGTGCN_H
#define REGTGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* regt_stream_t;

#define REGT_ABI_VERSION 7

int32_t regt_abi_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char* regt_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Graph preparation (once per static graph).
 *
 * Replaces the normalisation PyG recomputes inside every conv call because the reference builds
 * its layers with cached=False (models/RegionalTemporalGCN.py:54,73):
 *   GCNConv  gcn_norm          -- call sites models/utils.py:169,175,181
 *   ChebConv __norm__/get_laplacian -- call sites models/RegionalTemporalGCN.py:136-140,
 *                                      models/TemporalGCN.py:88
 * edge_index is the reference's (2,E) int64 tensor (row 0 = source, row 1 = target), edge_weight
 * its (E,) fp32 edge_attr or NULL for unit weights.
 * flags_dev[0] receives bit0 = an index was out of [0,N), bit1 = a negative weight was seen.
 * ---------------------------------------------------------------------------------------------- */
size_t regt_graph_workspace_bytes(int64_t num_edges, int32_t num_nodes);

/* A_hat = D^-1/2 (A + I) D^-1/2 as a destination-sorted CSR; rowptr (N+1), col/val (E + N). */
int32_t regt_gcn_csr(const int64_t* edge_index, const float* edge_weight, int64_t num_edges, int32_t num_nodes,
                     int32_t* rowptr, int32_t* col, float* val, int32_t* flags_dev,
                     void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* D^-1/2 of that normalisation alone: dis_out (N) = (in-degree sum in edge order + self loop)^-1/2, 0 where the degree is 0 --
 * the value regt_gcn_csr multiplies into every entry.  A region shard computes it for its own nodes and publishes it (one
 * all-reduce at graph preparation) so that the ranks that read those nodes as halo sources can finish their rows of A_hat
 * without normalising the global graph. */
int32_t regt_gcn_dis(const int64_t* edge_index, const float* edge_weight, int64_t num_edges, int32_t num_nodes, float* dis_out,
                     int32_t* flags_dev, void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* Per-edge scaled-Laplacian weights of ChebConv(K=2, 'sym', lambda_max=None): out_weight (E). */
int32_t regt_cheb_edge_weights(const int64_t* edge_index, const float* edge_weight, int64_t num_edges,
                               int32_t num_nodes, float* out_weight, int32_t* flags_dev,
                               void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* Destination-sorted CSR of given per-edge values (self loops dropped); rowptr (N+1), col/val (E). */
int32_t regt_raw_csr(const int64_t* edge_index, const float* edge_value, int64_t num_edges, int32_t num_nodes,
                     int32_t* rowptr, int32_t* col, float* val, int32_t* flags_dev,
                     void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* Mean-aggregation operator of SAGEConv(aggr='mean') (base block 'graphsage' of the TGCN cell, models/utils.py:99-100): row i
 * holds every listed in-edge j -> i (self loops and duplicate edges as listed) with weight 1 / in-degree(i); rowptr (N+1),
 * col/val (E).  Nodes without in-edges get an empty row (their mean is 0, as in PyG). */
int32_t regt_mean_csr(const int64_t* edge_index, int64_t num_edges, int32_t num_nodes, int32_t* rowptr, int32_t* col,
                      float* val, int32_t* flags_dev, void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* Order-independent 64-bit fingerprint of (edge_index, edge_weight) -- cache key for prepared graphs. */
int32_t regt_graph_fingerprint(const int64_t* edge_index, const float* edge_weight, int64_t num_edges,
                               uint64_t* out_dev, regt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse aggregation -- MessagePassing.propagate (gather x[row], scale, scatter-add at col) of both
 * conv types, as a pull over the prepared CSR:  Y[r,:] = sum_e val[e] * X[col[e],:].
 * X has nrows_x rows, Y nrows rows, both `width` fp32 wide (multiple of 4), 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
int32_t regt_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const float* X, float* Y,
                      int32_t nrows, int32_t nrows_x, int32_t width, regt_stream_t stream);

/* Both operators in one pass over a merged CSR (two weights per entry): YA = A x, YL = L x; width % 4 == 0 (widths that are no
 * multiple of 32 floats -- e.g. the reference's T * F = 48 -- take a whole-row kernel and must not exceed 2048). */
int32_t regt_spmm_dual(const int32_t* rowptr, const int32_t* col, const float* val_a, const float* val_l, const float* X,
                       float* YA, float* YL, int32_t num_nodes, int32_t width, regt_stream_t stream);

/* The same with bf16 rows in and out (REGT_GEMM_MODE=bf16: x, A_hat x and L~ x only ever feed bf16 matrix-core operands
 * there): X is (x_rows >= N) x width bf16 -- own rows first, then the halo rows of a region shard --, YA / YL are N x width
 * bf16; fp32 accumulation in CSR order, one rounding (to nearest even) per output element; width % 64 == 0. */
int32_t regt_spmm_dual_bf16(const int32_t* rowptr, const int32_t* col, const float* val_a, const float* val_l, const void* X,
                            void* YA, void* YL, int32_t num_nodes, int32_t x_rows, int32_t width, regt_stream_t stream);

/* Snapshot layout change (N,F,T) time-innermost (load_dataset.py:451-457) -> (N,T,F) rows. */
int32_t regt_pack_x(const float* x, float* x_packed, int32_t num_nodes, int32_t num_features, int32_t periods,
                    regt_stream_t stream);

/* ... rounding the snapshot to bf16 (nearest even) on the way: x_packed is (N, T, F) bf16, F % 8 == 0. */
int32_t regt_pack_x_bf16(const float* x, void* x_packed, int32_t num_nodes, int32_t num_features, int32_t periods,
                         regt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction -- torch.nn.Linear / PyG Linear call sites:
 *   out[M,N] = act(A[M,K] W[N,K]^T + bias)     act: 0 none, 1 leaky_relu(slope), 2 relu
 *                                              (3 sigmoid, 4 tanh: gate epilogues of the zero-hidden cell)
 * and the matching weight gradient  dW[N,K] = dOut[M,N]^T A[M,K]  (+ optional dbias = column sums).
 * regt_wgrad needs `slab` of regt_wgrad_slab_floats(...) floats.
 * ---------------------------------------------------------------------------------------------- */
int32_t regt_linear(const float* A, int64_t lda, int64_t M, int32_t K, const float* W, int64_t ldw, int32_t N,
                    const float* bias, int32_t act, float slope, float* out, int64_t ldo, regt_stream_t stream);
size_t regt_wgrad_slab_floats(int64_t M, int32_t N, int32_t K, int32_t with_bias);
int32_t regt_wgrad(const float* dOut, int64_t ldd, const float* A, int64_t lda, int64_t M, int32_t N, int32_t K,
                   float* dW, int64_t ldw, float* dbias, float* slab, regt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-model forward / backward.
 *
 * regt_dims:  N nodes, T periods (<= 255), F node features, C hidden (256 in the reference), R regions,
 *             O output_dim, H1 head hidden (128);  regional = 1 for RegionalTemporalGCN
 *             (models/RegionalTemporalGCN.py:9-149), 0 for TemporalGCN / A3TGCN (models/TemporalGCN.py:7-91).
 * regt_graph: stacked CSR with 2N rows -- rows [0,N) = A_hat of the full graph, rows [N,2N) = the
 *             (merged, node-disjoint) regional scaled Laplacians; node_region (N) = region whose
 *             Laplacian owns the node's row; chunk_tab (n_chunks,2) = [row_begin,row_end) ranges of
 *             (node*T + t) rows that lie inside one region, chunk_region (n_chunks) their region.
 * regt_params: the reference's state_dict tensors (names in comments), read-only.
 * regt_grads:  same tensors, written (not accumulated) by regt_backward; NULL entries are skipped.
 *
 * Reproducibility: for T <= 64 (the reference uses 6 and 12) two calls on the same inputs give bit-identical outputs and
 * gradients -- every sum has a fixed order, and an element of the hidden state receives at most two atomically added
 * partial sums (two addends commute).  For 64 < T <= 255 a node spans three or more 64-row blocks of the candidate
 * kernel, its hidden state is the sum of three or more float atomics in arrival order, and results can differ in the last
 * bits from run to run (parity within 1e-5 still holds; tests/test_gpu_model.py runs T = 150).
 * ---------------------------------------------------------------------------------------------- */
typedef struct regt_dims {
    int32_t N, T, F, C, R, O, H1;
    int32_t regional;
    float lrelu_slope;      /* 0.01 (F.leaky_relu default, RegionalTemporalGCN.py:143) */
    /* ABI v6 -- per-call configuration (no process state involved; zero-initialised = the process defaults):
     * arith: GEMM arithmetic of THIS call: 0 = the process default (regt_set_gemm_mode / REGT_GEMM_MODE), 1 = fp32 MFMA,
     *        2 = exact bf16x3 split, 3 = bf16 operands.  regt_backward must be given the value its forward ran with
     *        (checked against what the forward stored in the workspace).
     * flags: REGT_DIMS_* bits below. */
    int32_t arith;
    uint32_t flags;
} regt_dims;
#define REGT_ARITH_DEFAULT 0
#define REGT_ARITH_FP32 1
#define REGT_ARITH_BF16X3 2
#define REGT_ARITH_BF16 3
#define REGT_DIMS_NO_BF16_ROWS 1u   /* bf16 arithmetic without the bf16-row layout / fused forward (= regt_set_option("xbf", 0)) */
#define REGT_DIMS_NO_FUSED_BWD 2u   /* bf16 arithmetic with the three data-gradient launches (= regt_set_option("fused_bwd", 0)) */
#define REGT_DIMS_NO_SIDE_STREAM 4u /* every kernel of this call on `stream` itself (no library side stream) */

typedef struct regt_graph {
    const int32_t* rowptr;        /* (2N+1) */
    const int32_t* col;
    const float* val;
    const int32_t* node_region;   /* (N) */
    const int32_t* chunk_tab;     /* (n_chunks, 2) */
    const int32_t* chunk_region;  /* (n_chunks) */
    int32_t n_chunks;
    /* optional merged operator (N rows): one entry per distinct (row, col) of the two halves above with the
     * A_hat weight and the L~ weight side by side; when present (and T*F % 32 == 0 or T*F <= 2048) both aggregations are
     * produced by ONE gather pass.  All four NULL = not provided. */
    const int32_t* m_rowptr;      /* (N+1) */
    const int32_t* m_col;
    const float* m_val_a;
    const float* m_val_l;
    /* overlap = 1: the regional graphs are NOT node-disjoint (the reference's "random" decomposition,
     * load_dataset.py:324-329).  rowptr/col/val then hold (1+R)*N rows: A_hat, then one scaled Laplacian per
     * region; node_region / chunk tables / merged operator are ignored. */
    int32_t overlap;
    /* Region ids [region_lo, region_hi) that own rows of THIS graph (a region shard of a multi-GPU run owns a block of
     * the global regions; tgnn.linear stays replicated at its global size).  Per-region composed weights are only
     * built for these, and the gradient blocks of the other regions receive just the term every region shares.
     * 0, 0 = all of [0, R). */
    int32_t region_lo, region_hi;
    /* 1 = node_region is non-decreasing (the nodes of a region are contiguous: what every shipped decomposition produces).  The bf16
     * arithmetic then runs the row-owning fused forward kernel (csrc/fused_rows.hip); 0 = unknown / not sorted: the 64-row fused
     * kernel, which handles any order.  (ABI v7) */
    int32_t region_sorted;
} regt_graph;

typedef struct regt_params {
    const float* attention;      /* tgnn._attention (T) */
    const float* conv_lin_w[3];  /* tgnn._base_tgcn.conv_{z,r,h}.lin.weight (C,F) */
    const float* conv_bias[3];   /* tgnn._base_tgcn.conv_{z,r,h}.bias (C) */
    const float* gate_w[3];      /* tgnn._base_tgcn.linear_{z,r,h}.weight (C,2C) */
    const float* gate_b[3];      /* tgnn._base_tgcn.linear_{z,r,h}.bias (C) */
    const float* cheb_w0;        /* tgnn.conv.lins.0.weight (C,F) */
    const float* cheb_w1;        /* tgnn.conv.lins.1.weight (C,F) */
    const float* cheb_bias;      /* tgnn.conv.bias (C) */
    const float* region_w;       /* tgnn.linear.weight (C,R*C)   (regional only) */
    const float* region_b;       /* tgnn.linear.bias (C)         (regional only) */
    const float* head1_w;        /* linear1.weight (H1,C) */
    const float* head1_b;        /* linear1.bias (H1) */
    const float* head2_w;        /* linear2.weight (O,H1) */
    const float* head2_b;        /* linear2.bias (O) */
} regt_params;

typedef struct regt_grads {
    float* attention;
    float* conv_lin_w[3];
    float* conv_bias[3];
    float* gate_w[3];
    float* gate_b[3];
    float* cheb_w0;
    float* cheb_w1;
    float* cheb_bias;
    float* region_w;
    float* region_b;
    float* head1_w;
    float* head1_b;
    float* head2_w;
    float* head2_b;
} regt_grads;

/* Bytes of device workspace regt_forward / regt_backward need (same buffer for both: the forward
 * leaves the activations the backward reads). */
size_t regt_workspace_bytes(const regt_dims* dims, int32_t n_chunks, int32_t overlap);

/* x (N,F,T) -> pred (N,O), hidden (N,C).  RegionalTemporalGCN.forward / TemporalGCN.forward. */
int32_t regt_forward(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const float* x,
                     float* pred, float* hidden, void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* Region-sharded variant (one process per GPU): the caller has already packed its own N rows with
 * regt_pack_x into the first N rows of x_packed (x_rows >= N rows of T*F floats) and filled rows
 * [N, x_rows) with the halo rows received from the other ranks; graph->col of the A_hat half may
 * point at any of the x_rows rows.  Everything else is as regt_forward. */
int32_t regt_forward_packed(const regt_dims* dims, const regt_graph* graph, const regt_params* params,
                            const float* x_packed, int32_t x_rows, float* pred, float* hidden,
                            void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* The same for a caller that packs (regt_pack_x_bf16) and exchanges its rows as bf16 -- half the bytes on the xGMI links and
 * no conversion pass; needs REGT_GEMM_MODE=bf16 and a shape the fused forward covers (C = 256, F = 64, node-disjoint
 * regions, merged operator), otherwise an error is returned.  regt_backward must then get the same buffer as x_packed. */
int32_t regt_forward_packed_bf16(const regt_dims* dims, const regt_graph* graph, const regt_params* params,
                                 const void* x_packed_bf16, int32_t x_rows, float* pred, float* hidden,
                                 void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* Gradients of all parameters given dL/dpred (N,O) and optionally dL/dhidden (N,C) (may be NULL).
 * Must follow regt_forward on the same workspace; `hidden` is that forward's hidden output;
 * x_packed is NULL after regt_forward, or the buffer given to regt_forward_packed. */
int32_t regt_backward(const regt_dims* dims, const regt_graph* graph, const regt_params* params,
                      const regt_grads* grads, const float* dpred, const float* dhidden, const float* hidden,
                      const float* x_packed, void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* The part of the path every A3TGCN-style model of the reference shares -- TGCN cell (models/utils.py:163-203) over all
 * periods, attention-weighted sum, relu/linear head -- on a hidden input the CALLER computed: h_in is (N*T, C), row
 * node*T + t.  Replaces `self._base_tgcn(X[:, :, period], edge_index, edge_weight, h)` + head for models whose
 * embedding stage is not the regional one, e.g. models/ConvStackedTemporalGCN.py:117-126 (five stacked GCNConv).
 * dims.regional must be 0, dims.R is ignored (pass 1); graph needs rowptr/col/val of A_hat only (N rows, regt_gcn_csr);
 * params/grads: cheb_* and region_* are not used.  Workspace: regt_workspace_bytes(dims, 0, 0).
 * regt_cell_backward additionally returns dL/dh_in in dh_in (N*T, C). */
int32_t regt_cell_forward(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const float* x,
                          const float* h_in, float* pred, float* hidden, void* workspace, size_t workspace_bytes,
                          regt_stream_t stream);
int32_t regt_cell_backward(const regt_dims* dims, const regt_graph* graph, const regt_params* params,
                           const regt_grads* grads, const float* dpred, const float* dhidden, const float* hidden,
                           const float* h_in, float* dh_in, void* workspace, size_t workspace_bytes, regt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Zero-hidden TGCN cell + attention over periods + head, on gate inputs the CALLER aggregated.
 *
 * The reference's GraphSAGETemporalGCN and GATTemporal call `self._base_tgcn(X[:, :, period], edge_index, H)`
 * (models/GraphSAGETemporalGCN.py:93-95, models/GATTemporal.py:78-80): the third positional parameter of TGCN.forward is
 * edge_weight, so H stays None and the cell starts from zeros (models/utils.py:163-166).  With H = 0 the GRU reduces to
 *     Z = sigmoid(a_z gz^T + cz),   H~ = tanh(a_h gh^T + ch),   H' = (1 - Z) * H~,   hidden = sum_t softmax(att)_t H'_t
 * (the reset gate only ever multiplies H = 0).  a_z (M, kz) / a_h (M, kh), M = N*T rows ordered node*T + t, are the
 * aggregated inputs of the two live gates -- [mean-neighbour x | x] for SAGEConv, the attention-weighted neighbour sum of
 * GATConv (regt_gat_forward) -- and gz (C, kz) / gh (C, kh), cz / ch (C) the composed weights U_k[:, :C] W_k and biases
 * U_k[:, :C] b_k + u_k.  dims: N, T, C, O, H1 are used.  regt_cell0_backward writes the gradients of args' tensors; the
 * optional a_z / a_h entries (M, kz) / (M, kh) receive dL/d(input) (needed when the inputs depend on parameters: GAT).
 * ---------------------------------------------------------------------------------------------- */
typedef struct regt_cell0_args {
    const float* a_z; const float* a_h;
    int32_t kz, kh;
    const float* gz; const float* gh; const float* cz; const float* ch;
    const float* attention;                                          /* (T) */
    const float* head1_w; const float* head1_b; const float* head2_w; const float* head2_b;
} regt_cell0_args;
typedef struct regt_cell0_grads {
    float* a_z; float* a_h;                                          /* optional (NULL = not needed) */
    float* gz; float* gh; float* cz; float* ch;
    float* attention;                                                /* optional */
    float* head1_w; float* head1_b; float* head2_w; float* head2_b;
} regt_cell0_grads;
size_t regt_cell0_workspace_bytes(const regt_dims* dims, int32_t kz, int32_t kh);
int32_t regt_cell0_forward(const regt_dims* dims, const regt_cell0_args* args, float* pred, float* hidden, void* workspace,
                           size_t workspace_bytes, regt_stream_t stream);
int32_t regt_cell0_backward(const regt_dims* dims, const regt_cell0_args* args, const regt_cell0_grads* grads, const float* dpred,
                            const float* dhidden, const float* hidden, void* workspace, size_t workspace_bytes,
                            regt_stream_t stream);

/* GATConv (heads = 1, add_self_loops, negative_slope `slope`) attention aggregation on the INPUT rows, all T periods at once:
 *   e_ij = leaky_relu(<x_j, u_src> + <x_i, u_dst>),  alpha_ij = softmax over the in-edges of i,  out_i = sum_j alpha_ij x_j
 * with u_src = W^T att_src, u_dst = W^T att_dst (F) -- the conv's output is then out W^T + bias (models/utils.py:97-98 call
 * sites).  rowptr/col: the pattern of regt_gcn_csr (in-edges without self loops + one self loop per node).  x, out:
 * (N, T, F) packed rows; stats (N*T, 4) floats are kept for the backward.  regt_gat_backward turns dL/dout (N, T, F) into
 * the score gradients dsd (N*T, 2) = (dL/ds_j, dL/dd_i) per row; du_src = dsd[:, 0]^T x and du_dst = dsd[:, 1]^T x are then
 * ordinary regt_wgrad contractions.  t_rowptr/t_col: the transposed pattern (out-edges of every node). */
int32_t regt_gat_forward(const int32_t* rowptr, const int32_t* col, const float* x, const float* u_src, const float* u_dst,
                         float slope, int32_t num_nodes, int32_t periods, int32_t num_features, float* out, float* stats,
                         regt_stream_t stream);
int32_t regt_gat_backward(const int32_t* rowptr, const int32_t* col, const int32_t* t_rowptr, const int32_t* t_col,
                          const float* x, const float* u_src, float slope, int32_t num_nodes, int32_t periods,
                          int32_t num_features, const float* dout, float* stats, float* dsd, regt_stream_t stream);

/* Arithmetic of the dense contractions.  0 (default): fp32 MFMA (v_mfma_f32_32x32x2_f32).  1: every fp32 operand is
 * split exactly into three bf16 pieces and the six leading partial products run on the bf16 MFMA with fp32
 * accumulation -- fp32-level rounding error (dropped terms <= 3 * 2^-24 of a product), ~2x the matrix-pipe rate.
 * Also selectable with REGT_GEMM_MODE=bf16x3 before the first call.  Returns the previous mode. */
int32_t regt_set_gemm_mode(int32_t mode);

/* Developer switches (A/B timing and the bit-for-bit comparisons of tests/test_gpu_fused.py); returns the previous value, -1 for
 * an unknown name.  "xbf" (default 1): under REGT_GEMM_MODE=bf16, bf16 rows of x / A_hat x / L~ x and the fused forward kernel
 * where the shape allows; 0 = the three-launch forward on fp32 rows.  "fused_bwd" (default 1): the three data-gradient launches of that
 * arithmetic as one kernel.  "spmm_rows" (default 0 -- opt-in, measured slower): the row-block aggregation kernel (CSR entries of a
 * workgroup's rows held in LDS) instead of the column-panel kernels.  "dgrad1_gen" (default 1): fp32 arithmetic, the candidate data gradient
 * forms its left operand dhp from Z, H~, dOH while staging it and its epilogue writes dzp and the attention dots (no separate
 * cell-backward pass); 0 = the two launches.  "tgcn_collapse" (default 1): regional = 0 (TemporalGCN), fp32 / bf16x3: the gates' linear use of
 * the activation-free hidden input folded into x and L~ x (gate GEMM at K = 3F, no K = 2C gate data gradient); 0 = uncollapsed.
 * Weight gradients of the bf16-row layout (both operands stored as bf16): "wgrad_ring" (default 6): 16-row half slabs requested ahead
 * through a register ring (4 | 6 | 8); 0 = the one-ahead kernel, same partial sums bit for bit.  "wgrad_tile" (default 256): output
 * rows per tile (128 | 256); "wgrad_ring256" (default 2): ring depth of the 256-row tile (2 | 4).  "wgrad_pairs" (default 2 = on with the ring kernel; 0 | 1): the two gradients of each left operand as
 * one launch.  "wgrad_wave" (default 1): row chunks of those launches sized so that all their workgroups are resident at once
 * (another summation order over chunk boundaries than 0, the fixed ~128 chunks).  "wgrad_bnw64" (default 1): fp32 rows, a 33..64-wide
 * right-hand side ([x | L~ x] at F = 32) as one 64-column tile instead of two of 32 (bit-identical).  "xbf" / "fused_bwd" also exist per call: regt_dims.flags. */
int32_t regt_set_option(const char* name, int32_t value);

/* Developer hook (REGT_FUSED_TRACE=1, tools/fused_trace.py): shader-clock stamps of the last fused forward launch, 8 per 64-row
 * tile (start, tables, h, R / q, Z_0, candidate_0, Z_1, candidate_1), copied to out_host; synchronises the device.  Returns
 * the number of values written, 0 when tracing is off. */
int64_t regt_debug_trace(int64_t* out_host, int64_t capacity);

/* Per-stage timing with HIP events recorded on the launch stream (used by bench.py for the
 * roofline figures).  collect() waits for the recorded events and writes "name count total_ms"
 * lines into buf. */
int32_t regt_profile_enable(int32_t on);
/* Optional (environment REGT_HIPGRAPH=1; =2 for every size): small problems (N*T <= 32768 rows) replay a captured
 * hipGraph from the second call on identical buffers.  Off by default -- not faster at TPIMS size.  out[0..5] = forward {eager, captured, replayed},
 * backward {eager, captured, replayed} call counts. */
int32_t regt_graph_stats(int64_t* out);
int32_t regt_profile_collect(char* buf, size_t buf_bytes);

/* loss = mean((pred - y)^2) over `count` entries with mean taken over `global_count`
 * (run.py:180); writes dpred = dloss/dpred and the scalar loss. */
int32_t regt_mse_loss_grad(const float* pred, const float* y, float* dpred, float* loss_out, int64_t count,
                           int64_t global_count, regt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REGTGCN_H */
