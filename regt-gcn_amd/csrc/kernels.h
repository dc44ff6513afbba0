// Internal C++ launcher interface of the regtgcn HIP library (not exported; see include/regtgcn.h
// for the C ABI).  Every launcher enqueues on the given stream and returns REGT_OK / error code.
#pragma once
#include "gemm_core.h"

namespace regt {
// developer build (-DREGT_WG_TRACE, tools/wg_trace.py): per-workgroup wall-clock marks inside the GEMM cores
#ifdef REGT_WG_TRACE
#ifndef REGT_WG_TRACE_N
#define REGT_WG_TRACE_N 512
#endif
constexpr int WG_TRACE_MAX = 40000;
extern __device__ long g_wg_marks[8 * WG_TRACE_MAX];
// id of the tile at hand: wherever the macros are used, rm / n0 / N name the tile
#define WG_TILE_ID ((int)(rm.base / 128) * ((N + 127) / 128) + n0 / 128)
#define WG_MARK(i) do { if (threadIdx.x == 0 && WG_TILE_ID < WG_TRACE_MAX && N == REGT_WG_TRACE_N) g_wg_marks[8L * WG_TILE_ID + (i)] = wall_clock64(); } while (0)
#else
#define WG_MARK(i) do { } while (0)
#endif

// ---- buffer-descriptor addressing for the straight-line epilogues (gemm.hip functors) ----------------------------------
// One descriptor per array and tile (its base is the tile's origin, wave-uniform), a per-thread byte offset and a
// wave-uniform scalar offset per row slot: no vector instruction is spent on addresses inside the row loops.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_srd(const void* p) {      // p must be wave-uniform
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, int voff, int soff, float4 v) {
    const u32x4_t w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, soff, 0);
}
// four fp32 -> four bf16 (round to nearest even), one 8-byte store
__device__ __forceinline__ void buf_st4_bf16(__amdgpu_buffer_rsrc_t r, int voff, int soff, float4 v) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = {v.x, v.y}, hi = {v.z, v.w};
    const u32x2_t w = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_t)),
                       __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_t))};
    __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, soff, 0);
}
// GRU blend H' = Z h + (1 - Z) H~ (models/utils.py:186-188) with ONE fixed instruction sequence -- fma(Z, h, fl((1 - Z) H~)) --
// wherever it is computed (candidate epilogues in gemm.hip, fused forward in fused.hip): left to -ffp-contract the two
// kernels could round differently, and tests/test_gpu_fused.py compares them bit for bit.
__device__ __forceinline__ float gru_blend(float Z, float h, float ht) { return fmaf(Z, h, __fmul_rn(1.0f - Z, ht)); }
// The element-wise steps of the cell's backward pass, likewise with one fixed instruction sequence each (cell_bwd8_kernel and the
// data-gradient epilogues of the three-launch path, the fused backward kernel in fused.hip):  with g = p_t dOH[node]
//   dhp = g (1 - Z) (1 - H~^2)        dzp = g (h - H~) Z (1 - Z)        drp = dq h R (1 - R)        dh = dq R + g Z
//   ds = (dh + dzp Uz + drp Ur) act'(h)
__device__ __forceinline__ float cb_dhp(float g, float z, float ht) { return __fmul_rn(__fmul_rn(g, 1.0f - z), fmaf(-ht, ht, 1.0f)); }
__device__ __forceinline__ float cb_dzp(float g, float h, float ht, float z) { return __fmul_rn(__fmul_rn(g, h - ht), __fmul_rn(z, 1.0f - z)); }
__device__ __forceinline__ float cb_drp(float dq, float h, float r) { return __fmul_rn(__fmul_rn(dq, h), __fmul_rn(r, 1.0f - r)); }
__device__ __forceinline__ float cb_dh(float dq, float r, float p, float d, float z) { return fmaf(dq, r, __fmul_rn(__fmul_rn(p, d), z)); }
__device__ __forceinline__ float cb_ds(float dh, float acc, float factor) { return __fmul_rn(dh + acc, factor); }

// per tile row: byte offset of the row's node in an (N, C) fp32 array and its period's attention probability (EpiDgrad1F)
struct EpiRowEnt { int off; float p; };
// where a thread sits in the epilogue of a full tile: rows rr + step * i (i = row slot), columns c .. of the tile at (m0, n0)
struct EpiGeom { long m0; int n0, rr, c, step; const EpiRowEnt* rowtab; };
constexpr int EPI_ROWTAB_BYTES = 128 * (int)sizeof(EpiRowEnt);


// ---- epilogue descriptors for the flat segmented GEMM ------------------------------------------
enum : int { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };

// `*_bf16` flags (REGT_GEMM_MODE=bf16 with bf16 storage of the M x C activations): the array holds bf16 elements (strides
// in elements); the bf16-operand core then runs its 8-column-per-thread epilogue so that every access stays 16 bytes wide.
struct EpiBiasAct {     // out[m, c] = act(v + bias[c])
    float* out; long ldo; const float* bias; int act; float slope;
    int out_bf16 = 0;
};
struct EpiGates {       // N = 2C: c <  C: Z = sigmoid(v + b) -> ZR[m, c]
    float* ZR;          //         c >= C: R = sigmoid(v + b) -> ZR[m, c], q[m, c-C] = h[m, c-C] * R
    const float* h; float* q; const float* bias; int C;
    int q_bf16 = 0;     // q is stored as bf16 (row stride C elements): it only ever feeds matrix-core operands
    int h_bf16 = 0, zr_bf16 = 0;
};
struct EpiDgrad1 {      // v = dq[m, c]:  dzr[m, C+c] = v*h*R*(1-R);  dh[m, c] = v*R + p[t]*dOH[node, c]*Z
    const float* h; const float* ZR; const float* dOH; const float* probs;
    float* dzr; float* dh; int C; int T;
    int dzr_bf16 = 0;   // dzr is stored as bf16 (row stride 2C elements)
    int h_bf16 = 0, zr_bf16 = 0, dh_bf16 = 0;
    // launch_gemm_dgrad1_gen (fp32): the left operand dhp = g (1 - Z)(1 - H~^2) is GENERATED from Z, H~, dOH while it is staged (and
    // stored to dhp for the weight gradients); the epilogue also writes dzp = g (h - H~) Z (1 - Z) -> dzr[m, c] and the row's
    // partial attention dot <dOH, Z h + (1 - Z) H~> over the tile's 128 columns -> rowdot[m * (C / 128) + c / 128]
    const float* Ht = nullptr; float* dhp = nullptr; float* rowdot = nullptr; int num_nodes = 0;
};
struct EpiDgrad2 {      // ds[m, c] = (dh[m, c] + v) * act'(h[m, c])   (in place on dh)
    float* dh; const float* h; int C; int act; float slope;
    int h_bf16 = 0, dh_bf16 = 0;
};
struct EpiMaskAdd {     // out[m, c] = v * (mask[m, c] > 0) + (add ? add[m, c] : 0)
    float* out; long ldo; const float* mask; long ldm; const float* add; long ldadd;
};

int launch_gemm_bias_act(const GemmSegs& S, long M, int N, const EpiBiasAct& e, hipStream_t st);
int launch_gemm_gates(const GemmSegs& S, long M, int N, const EpiGates& e, hipStream_t st);
int launch_gemm_dgrad1(const GemmSegs& S, long M, int N, const EpiDgrad1& e, hipStream_t st);
// cell_bwd + dgrad_candidate in one launch (fp32 arithmetic, C % 128 == 0, big-tile regime): see EpiDgrad1's last fields
bool gemm_dgrad1_gen_ok(long M, int C, int num_nodes);
int wgrad_bnw64_option(int value);   // fp32 rows: one 64-column tile for a 33..64-wide right-hand side (regt_set_option "wgrad_bnw64")
int wgrad_ring256_option(int value);  // ring depth of the 256-row tile (2 | 4)
int wgrad_tile_option(int value);    // 128 | 256 output rows per tile of the ring kernel (regt_set_option "wgrad_tile")
int wgrad_wave_option(int value);    // one-wave row chunking of ring-kernel launches (regt_set_option "wgrad_wave"; -1 = query)
bool wgrad_ring_chunking(int Nout, int Nin, long M, int* kchunk, int* nchunks);
bool wgrad_skinny_chunking(int Nout, long M, int* kchunk, int* nchunks);            // skinny (Nin <= 32) fp32-MFMA kernel
bool wgrad_wide_chunking(int Nout, int Nin, long M, int* kchunk, int* nchunks);   // wide fp32 / bf16x3 kernels (experiment)
long wgrad_chunk_bound(int Nout, int Nin, long M);   // upper bound of what the three can return (slab sizing, api.hip make_layout)
bool wgrad_ring_active();             // the ring kernel takes the bf16-stored weight gradients (pairs pay off with it)
int wgrad_ring_option(int value);    // runtime A/B switch (regt_set_option "wgrad_ring"): ring depth of the bf16 weight gradient
int dgrad1_gen_option(int value);    // runtime A/B switch (regt_set_option "dgrad1_gen")
int launch_gemm_dgrad1_gen(const GemmSegs& S, long M, int N, const EpiDgrad1& e, hipStream_t st);
int launch_gemm_dgrad2(const GemmSegs& S, long M, int N, const EpiDgrad2& e, hipStream_t st);
int launch_gemm_mask_add(const GemmSegs& S, long M, int N, const EpiMaskAdd& e, hipStream_t st);

// Candidate-state GEMM with the loop over the T periods inside the workgroup:
//   for t: Ht = tanh(q_t Uh2^T + ax_t Gh^T + ch);  Hn = Z*h + (1-Z)*Ht;  OH += p[t]*Hn
struct CandArgs {
    GemmSegs S;             // segments with A rows addressed as (node*T + t)
    int num_nodes, T, C;
    const float* bias;      // ch (C)
    const float* ZR;        // (M, 2C)
    const float* h;         // (M, C)
    const float* probs;     // (T)
    float* Ht;              // (M, C) out
    float* OH;              // (num_nodes, C) out
    int act_bf16 = 0;       // ZR, h and Ht hold bf16 elements (all three together)
    int node_sum_rows = 64; // bf16 storage: the rows of a node are summed in row order inside blocks of 64 (the staged half) or 16
                            //   rows (what a wave of fused_rows.hip owns: the two paths then agree bit for bit); T <= 16 for the latter
};
int launch_gemm_candidate(const CandArgs& a, hipStream_t st);

// ---- weight-gradient GEMM: out[Nout x Nin] = P^T Q, split over row chunks ----------------------
struct WgradArgs {
    const float* P; long ldp; int Nout;
    const float* Q; long ldq; int Nin;
    int q_relu;               // apply relu to Q while staging
    long M;                   // rows
    int kchunk;               // rows per chunk (uniform) when chunk_tab == nullptr
    const int* chunk_tab;     // optional (nchunks, 2) [row_start, row_end)
    int nchunks;
    float* slab;              // (nchunks, Nout*Nin + (colsum ? Nout : 0))
    int colsum;               // also produce column sums of P (bias gradient)
    // optional second right-hand operand (skinny kernel only): output columns >= nin_split come from Q2 (column j -
    // nin_split).  Two gradients that share P -- dA0 = ds^T x and dA_r = ds^T (L~ x) -- then read P from HBM once.
    const float* Q2 = nullptr; long ldq2 = 0; int nin_split = 0;
    int p_bf16 = 0, q_bf16 = 0;   // P / Q hold bf16 elements (ldp / ldq in elements); vector kernels only
    int all_csum = 0;             // ring kernel: every column tile forms the column sums (only tile 0 stores them)
};
int launch_wgrad(const WgradArgs& a, hipStream_t st);
long wgrad_slab_stride(const WgradArgs& a);

struct WgradReduceArgs {
    const float* slab; int nchunks; long slab_stride;
    long elem_offset;         // first slab element of the (Nout x Nin) block to reduce
    long slab_ld;             // row stride of that block inside a slab (0: dense, = Nin)
    int Nout, Nin;
    const int* chunk_group;   // optional (nchunks): output group of each chunk
    int ngroups;              // groups group_base .. group_base + ngroups - 1 are reduced (into out blocks 0 .. ngroups - 1)
    int group_base;
    float* out; long ldo; long group_stride;   // out[g*group_stride + i*ldo + j]
    float* colsum_out;        // optional (ncolsum), only ngroups == 1
    long colsum_offset;       // slab element of the first column sum
    int ncolsum;
    int accumulate;           // add into out instead of overwrite
};
int launch_wgrad_reduce(const WgradReduceArgs& a, hipStream_t st);
// several independent reductions in one launch (the backward pass defers all of its slab reductions to one place)
constexpr int WR_MAX_TASKS = 12;
struct WgradReduceBatch {
    int n;
    int block_start[WR_MAX_TASKS + 1];
    WgradReduceArgs t[WR_MAX_TASKS];
};
int launch_wgrad_reduce_multi(WgradReduceBatch& b, hipStream_t st);

// ---- tiny strided batched GEMM (weight composition and its backward) ---------------------------
//   C[b][i, j] (+)= sum_over_batch? sum_k A[b][i, k] * B[b][k, j]      arbitrary strides
struct SmallGemm {
    const float* A; long sai, sak, sab;
    const float* B; long sbk, sbj, sbb;
    float* C; long sci, scj, scb;
    int m, n, k, batch;
    int sum_batch;     // 1: reduce over the batch index into one C
    int accumulate;    // 1: C += result
};
int launch_small_gemm(const SmallGemm& g, hipStream_t st);

// Several such products in ONE launch (the per-step weight compositions are ~27 tiny dependent-free GEMMs; one
// launch each would dominate the step at TPIMS size).  out[b][i,j] = init[i] + sum_terms sum_k A[..]*B[..].
// (compact: 32-bit strides, 16-bit counts -- a whole batch is passed BY VALUE in the kernel-argument block, 4 KB at most, and the
// composition backward is one launch of 19 tasks since round 4)
struct SgTerm {                  // 48 bytes
    const float* A; const float* B;
    int sai, sak, sab, sbk, sbj, sbb;
    int k;                       // sum length (< 0: a stride did not fit 32 bits -- add_task refuses the task)
    short batch, sum_batch;      // sum_batch: reduce over `batch` operand pairs; else use the output's batch index
};
struct SgTask {                  // 200 bytes
    float* C; const float* init; // optional init[i*init_si + j*init_sj] added to the sum (nullptr: 0; init_sj = 0: the same value
    int sci, scj, scb, init_si, init_sj;   // for every column)
    int m, n;                    // output (nbatch, m, n)
    short nbatch, nterm;
    short split, pad_;           // lanes per output element (set by launch_small_gemm_multi: 1 for short sums, 8 for long ones, 0 tiled)
    SgTerm term[3];
};
constexpr int SG_MAX_TASKS = 19;
struct SgBatch {
    int ntask;
    int overflow;                // set by add_task when a task did not fit (launch_small_gemm_multi then fails)
    int block_start[SG_MAX_TASKS + 1];
    SgTask task[SG_MAX_TASKS];
};
static_assert(sizeof(SgBatch) <= 4000, "SgBatch travels in the kernel-argument block");
int launch_small_gemm_multi(SgBatch& b, hipStream_t st);

// ---- graph preparation (graph.hip) ---------------------------------------------------------------
size_t graph_workspace_bytes(long E, int N);
int graph_gcn_csr(const int64_t* ei, const float* w, long E, int N, int* rowptr, int* col, float* val,
                  int* flags_out_dev, void* ws, size_t ws_bytes, hipStream_t st);
int graph_gcn_dis(const int64_t* ei, const float* w, long E, int N, float* dis_out, int* flags_out_dev, void* ws, size_t ws_bytes,
                  hipStream_t st);
int graph_cheb_edge_weights(const int64_t* ei, const float* w, long E, int N, float* out_w, int* flags_out_dev,
                            void* ws, size_t ws_bytes, hipStream_t st);
int graph_raw_csr(const int64_t* ei, const float* w, long E, int N, int* rowptr, int* col, float* val,
                  int* flags_out_dev, void* ws, size_t ws_bytes, hipStream_t st);
int graph_fingerprint(const int64_t* ei, const float* w, long E, unsigned long long* out_dev, hipStream_t st);
int graph_mean_csr(const int64_t* ei, long E, int N, int* rowptr, int* col, float* val, int* flags_out_dev, void* ws,
                   size_t ws_bytes, hipStream_t st);

// ---- sparse aggregation ------------------------------------------------------------------------
int launch_pack_x(const float* x, float* xp, int N, int F, int T, hipStream_t st);            // (N,F,T) -> (N,T,F)
int launch_spmm_csr(const int* rowptr, const int* col, const float* val, const float* X, float* Y,
                    int nrows, int nrows_x, int W, int nstack, hipStream_t st);                  // Y[r] = sum val*X[col]; rows = nstack x nodes

int launch_spmm_dual(const int* rowptr, const int* col, const float* val_a, const float* val_l, const float* X, float* YA,
                     float* YL, int nnodes, int W, hipStream_t st);   // YA = A x, YL = L x from one merged CSR (two weights/entry)
// the same with X holding x_rows >= nnodes rows (region shard: own rows, then halo rows)
int launch_spmm_dual_x(const int* rowptr, const int* col, const float* val_a, const float* val_l, const float* X, float* YA,
                       float* YL, int nnodes, int x_rows, int W, hipStream_t st);
// bf16 rows in (X: x_rows x W bf16) and out (YA, YL: nnodes x W bf16), fp32 accumulation, one rounding at the end; W % 64 == 0
int launch_spmm_dual_bf16(const int* rowptr, const int* col, const float* val_a, const float* val_l, const void* X, void* YA, void* YL,
                          int nnodes, int x_rows, int W, hipStream_t st);

int spmm_rows_option(int value);   // runtime A/B switch of the row-block aggregation kernel (regt_set_option "spmm_rows")
int launch_pack_x_bf16(const float* x, void* xp, int N, int F, int T, hipStream_t st);        // (N,F,T) fp32 -> (N,T,F) bf16, F % 8 == 0
int launch_cvt_rows_bf16(const float* src, void* dst, long n, hipStream_t st);                // n % 8 == 0 elements fp32 -> bf16

// ---- fused forward of the cell for the bf16 arithmetic (fused.hip) ----------------------------------------------------------
struct FusedFwdArgs {
    const void *X, *LX, *AX;                  // bf16 rows (M x F): packed input, L~ x, A_hat x (rows node * T + t)
    const void *A0f, *Aallf; long ar_stride;  // composed (C x F) weights in MFMA fragment order (launch_cvt_bf16_frag), one block per region
    const void *Uzf, *Urf, *Uhf;              // the h-halves of linear_z / _r / _h (C x C), fragment order
    const void *Gzrf, *Ghf;                   // composed [Gz; Gr] (2C x F) and Gh (C x F), fragment order
    const float *bprime, *czr, *ch, *probs;   // composed biases (C, 2C, C), softmax(attention) (T)
    const int* node_region;                   // (nodes) or nullptr (one region)
    void *h, *ZR, *q, *Ht;                    // bf16 outputs: (M x C), (M x 2C) = [Z | R], (M x C), (M x C)
    float* OH;                                // (nodes x C) fp32, zero-initialised: the attention-weighted hidden state
    long M; int T; float slope; int act_lrelu;
    long nodes;                               // M / T                                            } filled in by launch_fused_forward
    unsigned long long pmask;                 // bit k T for every k T < 64 (rows that start a node) }
    unsigned* tile_ctr;                       // workspace word (zeroed by the launcher): the persistent workgroups draw their tiles from it
    const char* wbase;                        // fused_rows.hip: the weight blocks as 32-bit offsets from one base (filled in by its launcher)
    unsigned o_uz, o_ur, o_uh, o_gzr, o_gh, o_a0, o_aall;
    int dbg;                                  // timing-only switches (REGT_FUSED_DBG, fused.hip); 0 in normal operation
    long* trace;                              // developer trace buffer (REGT_FUSED_TRACE) or nullptr
};
long fused_trace_fetch(long* out, long capacity);
int launch_fused_forward(const FusedFwdArgs& a, int C, int F, hipStream_t st);
bool fused_forward_ok(int C, int F);
// the row-owning form (fused_rows.hip): a wave owns 16 whole rows, weights stream through LDS; needs region ids sorted by node.
// waves = 8: one workgroup of eight waves per CU (the product form); 4: two workgroups of four with half the ring each (slower --
// twice the weight traffic -- kept because its short ring is the harder test of the hand-counted waits)
int launch_fused_forward_rows(const FusedFwdArgs& a, int C, int F, int waves, hipStream_t st);
bool fused_forward_rows_ok(int C, int F, int T);

// ---- fused data gradients of the cell for the bf16 arithmetic (fused.hip): cell_bwd + dgrad_candidate + dgrad_gates in one kernel
struct FusedBwdArgs {
    const void *ZR, *h, *Ht;                  // bf16, stored by the forward: (M x 2C) = [Z | R], (M x C), (M x C)
    const float *dOH, *probs;                 // (nodes x C) fp32 gradient of the attention-weighted hidden state, softmax(attention) (T)
    const void *UhTf, *UzTf, *UrTf;           // TRANSPOSED h-halves of linear_h / _z / _r (C x C) in MFMA fragment order
    void *dhp, *dzr, *dh;                     // bf16 outputs: (M x C), (M x 2C) = [dzp | drp], (M x C) = ds
    float* rowdot;                            // (M) fp32: <dOH[node], H'[m]> per row (attention-probability gradient, summed later)
    unsigned* tile_ctr;                       // workspace word (zeroed by the launcher): the persistent workgroups draw their tiles from it
    long M; int T; float slope; int act_lrelu;
    long* trace;                              // developer trace buffer (REGT_FUSED_TRACE=2) or nullptr
};
int launch_fused_backward(const FusedBwdArgs& a, int C, hipStream_t st);
bool fused_backward_ok(int C);
// dp_partial[b][t] = sum over the nodes of block b (nodes_per_block consecutive nodes, ascending) of rowdot[node * T + t]
// (`parts` > 1: a row holds `parts` consecutive partial dots, added in ascending order)
int launch_rowdot_reduce(const float* rowdot, float* dp_partial, int num_nodes, int T, int nodes_per_block, hipStream_t st, int parts = 1);

// ---- cell backward head / small element-wise kernels -------------------------------------------
// GEMM arithmetic: 0 = fp32 MFMA (default), 1 = exact 3-way bf16 split on the bf16 MFMA (gemm_split.h)
int gemm_mode();
void set_gemm_mode(int mode);
int gemm_mode_override(int mode);   // thread-local override for one call (-1 = none); returns the previous override
bool fp32_core_wide();   // REGT_FP32_CORE=wide (A/B timing of the two fp32 GEMM cores)

// fp32 -> bf16 (round to nearest even) copies of up to 8 weight blocks in MFMA fragment order (SEG_B_FRAG), one launch:
// block t = rows x cols (cols % 16 == 0) at src with leading dimension ld; dst holds ceil(rows / 128) * 128 rows (zero padded)
struct CvtTask { const float* src; long ld; int rows, cols; void* dst; };
struct CvtBatch { int n; CvtTask t[8]; int block_start[9]; };
int launch_cvt_bf16_frag(CvtBatch& b, hipStream_t st);
inline long frag_bytes(long rows, long cols) { return ((rows + 127) / 128) * 128 * cols * 2; }
bool gemm_desc_table_forced();     // REGT_GEMM_DESC=table
int launch_transpose3(const float* s0, const float* s1, const float* s2, int count, float* dst, int rows, int cols, long ld,
                      hipStream_t st);   // dst[b] = src[b]^T, b < count <= 3
// last head layer for output_dim <= 4 as row-wise fp32 kernels (cell.hip)
bool head2_skinny_ok(int H1, int O, const void* y1, const void* W2);
int launch_head2_fwd(const float* y1, const float* W2, const float* b2, float* pred, int N, int H1, int O, hipStream_t st);
int launch_head2_bwd(const float* dpred, const float* W2, const float* y1, float* d1, int N, int H1, int O, hipStream_t st);
int launch_head2_wgrad(const float* dpred, const float* y1, float* slab, int N, int H1, int O, int kchunk, int nchunks, int colsum,
                       hipStream_t st);
// S[i, j] = sum_r W[i, r*C + j]: the sum of the R (C x C) blocks of tgnn.linear.weight (cell.hip)
int launch_sum_region_blocks(const float* W, float* S, int C, int R, hipStream_t st);
int launch_softmax_small(const float* att, float* probs, int T, hipStream_t st);
struct CellBwdArgs {
    const float* dOH; const float* probs; const float* ZR; const float* h; const float* Ht;
    float* dhp; float* dzr; float* dp_partial; int num_nodes, T, C; int nodes_per_block;
    // zero-hidden cell (GraphSAGE / GAT models): h == nullptr means H = 0; Z and dzp then are plain (M, C) arrays
    int ldz = 0, lddz = 0;     // row strides of ZR and dzr in floats; 0 = 2C ([Z|R] and [dzp|drp] layouts of the GRU cell)
    int out_bf16 = 0;   // dhp / dzr are stored as bf16 (same element strides)
    int in_bf16 = 0;    // ZR, h and Ht hold bf16 elements (all three together; needs out_bf16 and C % 8 == 0)
};
int launch_cell_bwd(const CellBwdArgs& a, hipStream_t st);
int cell_bwd_blocks(int num_nodes, int nodes_per_block);
int launch_att_bwd(const float* dp_partial, int nblocks, const float* probs, float* datt, int T, hipStream_t st);
// zero-hidden cell: hidden[n, :] = sum_t probs[t] * (1 - Z[n*T+t, :]) * Ht[n*T+t, :]
int launch_blend0_fwd(const float* Z, const float* Ht, const float* probs, float* hidden, int num_nodes, int T, int C, hipStream_t st);
// GATConv attention aggregation on input rows (gat.hip)
int launch_gat_forward(const int* rowptr, const int* col, const float* x, const float* us, const float* ud, float slope, int N,
                       int T, int F, float* out, float* stats, hipStream_t st);
int launch_gat_backward(const int* rowptr, const int* col, const int* t_rowptr, const int* t_col, const float* x, const float* us,
                        float slope, int N, int T, int F, const float* dout, float* stats, float* dsd, hipStream_t st);
int launch_copy_f32(float* dst, const float* src, long n, hipStream_t st);
int launch_zero_f32(float* dst, long n, hipStream_t st);
int launch_mse_grad(const float* pred, const float* y, float* dpred, float* loss_out, long n, float scale, hipStream_t st);

// the regional embedding of the fp32 path at C = 256, F = 32 as a kernel of its own (embed.hip); region ids must ascend with the node number
bool embed_fp32_ok(long M, int C, int F, int T);
int launch_embed_fp32(const float* X, const float* LX, const float* A0, const float* Aall, const int* node_region, const float* bias,
                      float* out, long M, int T, int act, float slope, hipStream_t st);

// hipFuncSetAttribute is a (slow, host-synchronous) driver call: do it once per kernel, not per launch.
template <class K>
static int set_lds_once(K kernel, int bytes, bool* done) {
    if (*done) return REGT_OK;
    REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    *done = true;
    return REGT_OK;
}

}  // namespace regt
