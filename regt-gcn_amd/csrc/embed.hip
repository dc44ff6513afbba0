// Regional embedding of the fp32 path at the cfg-3 shape: h = act(x A0^T + (L~ x) A_region^T + b')   (M x 256, K = 2 F = 64)
//
// Op site: models/RegionalTemporalGCN.py:136-148 in the composed-weight form of DESIGN.md section 3 (the ChebConv + GCNConv +
// linear of the regional stage collapse into two (C x F) matrices per region).  Through the general GEMM core (gemm_split.h) this
// product takes 0.52 ms at cfg-3: a K = 64 problem has two 32-k slabs per 128 x 128 tile, so every tile pays the core's prologue,
// exposes the HBM latency of both slabs and runs an epilogue of the size of its K loop (profiles/r04_gemm_regional_wg_trace.txt: 11 us
// of K loop for 3 us of matrix work).  Its floors are 0.29 ms of fp32 MFMA (39 GFLOP at 157 TFLOP/s -- the 16x16x4 instruction
// has the rate of the 32x32x2 one) and 0.26 ms of HBM (1.54 GB, four fifths of it the store of h).  This kernel is written for
// exactly that shape:
//   * persistent, one workgroup of 8 waves per CU, 128-row tiles; wave w owns rows 64 (w / 4) .. + 63 and columns 64 (w % 4) .. + 63;
//   * the weights never move: [A0 | A_region] of the workgroup's current region are staged once into LDS (k-major, padded) and from
//     there into 64 registers per lane -- the MFMA "A" operand of all tiles of that region (a tile that starts in another region
//     restages: with node-sorted regions that is once per region and workgroup);
//   * roles swapped: weights are the A operand (16 output columns x 4 k), the rows of x / L~ x the B operand (4 k x 16 rows), so an
//     accumulator lane holds FOUR CONSECUTIVE OUTPUT COLUMNS of one row: the epilogue is bias (the accumulators start as the bias) +
//     activation + one 16-byte store per block, 16 rows x 64 contiguous bytes per instruction -- no transposition through LDS;
//   * the next tile's rows are requested (4 x 16 bytes per thread) before the current tile's MFMAs and written into the other LDS
//     buffer behind them; a row block's four stores go out while the next row block's 64 MFMAs run.
// A tile whose rows belong to two regions (region ids ascend with the node number) is computed once per region and each pass
// stores only its own rows.  Everything else (F != 32, C != 256, overlapping regions, bf16 / split arithmetic) keeps the general core.
#include "kernels.h"

namespace regt {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int EM_ROWS = 128, EM_C = 256, EM_F = 32, EM_K = 2 * EM_F;
constexpr int EM_WS = EM_C + 16;         // floats per k row of the weight image: 4 lane groups (k, k + 1, ...) land on 4 x 16 distinct banks
constexpr int EM_XS = EM_ROWS + 18;      // ... of a row-tile image: 4 * XS = 8 (mod 64) keeps the transposing writes conflict-free too
constexpr int EM_W_BYTES = EM_K * EM_WS * 4, EM_X_BYTES = EM_K * EM_XS * 4;
constexpr int EM_LDS = EM_W_BYTES + 2 * EM_X_BYTES;

struct EmbedArgs {
    const float* X; const float* LX;     // (M x F) rows
    const float* A0; const float* Aall;  // (C x F), (R x C x F)
    const int* node_region;              // per node (NULL: one region)
    const float* bias;                   // (C)
    float* out;                          // (M x C)
    long M; int T; float ns;             // ns: what a negative value is multiplied by (1 none, slope leaky relu, 0 relu)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t em_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes > 0x7ffffff0L ? 0x7ffffff0 : (bytes < 0 ? 0 : (int)bytes), 0x00020000);
}

__global__ __launch_bounds__(512, 2) void embed_fp32_kernel(EmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char elds[];
    float* W = reinterpret_cast<float*>(elds);                                    // [k][EM_WS]
    float* XB = reinterpret_cast<float*>(elds + EM_W_BYTES);                      // two buffers [k][EM_XS]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), rh = w >> 2, cq = w & 3;
    const long tiles = (a.M + EM_ROWS - 1) / EM_ROWS;
    const unsigned uT = (unsigned)a.T;

    // the thread's part of a row tile: float4 number f = tid + 512 i (i = 0, 1) of x and of L~ x -- row f / 8, k = 4 (f % 8) ..
    f32x4 px[2], pl[2];
    auto request_rows = [&](long tile) {
        const long m0 = tile * EM_ROWS;
        const long left = a.M - m0;
        const long nv = left < 0 ? 0 : (left < EM_ROWS ? left : EM_ROWS);
        const __amdgpu_buffer_rsrc_t sx = em_rsrc(a.X + m0 * EM_F, nv * EM_F * 4), sl = em_rsrc(a.LX + m0 * EM_F, nv * EM_F * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = (tid + 512 * i) * 16;
            px[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sx, off, 0, 0));
            pl[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sl, off, 0, 0));
        }
    };
    auto park_rows = [&](int buf) {                              // registers -> LDS image [k][row] (x: k 0..31, L~ x: k 32..63)
        float* xb = XB + buf * (EM_K * EM_XS);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 512 * i, row = f >> 3, k0 = 4 * (f & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xb[(k0 + j) * EM_XS + row] = px[i][j];
                xb[(EM_F + k0 + j) * EM_XS + row] = pl[i][j];
            }
        }
    };
    // weights of region rg into the LDS image W[k][column] (k 0..31: A0, 32..63: A_rg), then into the lane's registers:
    // afr[s][cb] = W[4 s + g][64 cq + 16 cb + c], the A operand of k step s and column block cb
    float afr[16][4];
    auto stage_weights = [&](int rg) {
        const float* Ar = a.Aall + (long)rg * EM_C * EM_F;
        for (int f = tid; f < EM_C * EM_F / 4; f += 512) {       // float4 f: column f / 8, k = 4 (f % 8) ..
            const int col = f >> 3, k0 = 4 * (f & 7);
            const float4 v0 = *reinterpret_cast<const float4*>(a.A0 + (long)col * EM_F + k0);
            const float4 v1 = *reinterpret_cast<const float4*>(Ar + (long)col * EM_F + k0);
            W[(k0 + 0) * EM_WS + col] = v0.x; W[(k0 + 1) * EM_WS + col] = v0.y; W[(k0 + 2) * EM_WS + col] = v0.z; W[(k0 + 3) * EM_WS + col] = v0.w;
            W[(EM_F + k0 + 0) * EM_WS + col] = v1.x; W[(EM_F + k0 + 1) * EM_WS + col] = v1.y; W[(EM_F + k0 + 2) * EM_WS + col] = v1.z; W[(EM_F + k0 + 3) * EM_WS + col] = v1.w;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) afr[s][cb] = W[(4 * s + g) * EM_WS + 64 * cq + 16 * cb + c];
        __syncthreads();                                         // (the image may be restaged by the next region)
    };
    auto region_of_row = [&](long m) { return a.node_region ? a.node_region[(unsigned)m / uT] : 0; };

    // the lane's bias: columns 64 cq + 16 cb + 4 g .. + 3
    f32x4 bias4[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bias4[cb] = *reinterpret_cast<const f32x4*>(a.bias + 64 * cq + 16 * cb + 4 * g);

    long tile = blockIdx.x;
    if (tile >= tiles) return;
    int cur_rg = -1, buf = 0;
    // first / last region of a tile (looked up one tile ahead, like its rows: two dependent global loads at the top of a tile were
    // 1-2 us of every 12)
    auto tile_regions = [&](long t, int* first, int* last) {
        *first = *last = 0;
        if (t < tiles) {
            const long m0 = t * EM_ROWS, m1 = (m0 + EM_ROWS < a.M ? m0 + EM_ROWS : a.M) - 1;
            *first = region_of_row(m0);
            *last = region_of_row(m1);
        }
    };
    int rg_first, rg_last;
    tile_regions(tile, &rg_first, &rg_last);
    request_rows(tile);
    park_rows(0);
#pragma unroll 1
    for (; tile < tiles; tile += gridDim.x) {
        const long m0 = tile * EM_ROWS;
        const long left = a.M - m0;
        const int nv = (int)(left < EM_ROWS ? left : EM_ROWS);
        __syncthreads();                                         // this tile's image is written; the other buffer is free again
        const long tnext = tile + gridDim.x;
        request_rows(tnext);                                     // (past the end: empty descriptors, zeros)
        int nrg_first, nrg_last;
        tile_regions(tnext, &nrg_first, &nrg_last);
        const float* xb = XB + buf * (EM_K * EM_XS) + 64 * rh + c;
#pragma unroll 1
        for (int rg = rg_first; rg <= rg_last; ++rg) {
            if (rg != cur_rg) { stage_weights(rg); cur_rg = rg; }
            // rows of this region inside the tile: [lo, hi) (regions ascend with the node number); the stores of other rows are dropped
            int lo = 0, hi = nv;
            if (rg_first != rg_last) {
                // (rare: a linear walk over the tile's node boundaries would do; a binary search over rows is as short)
                int l2 = 0, h2 = nv;                             // first row whose region is >= rg
                while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (region_of_row(m0 + mid) < rg) l2 = mid + 1; else h2 = mid; }
                lo = l2;
                l2 = lo; h2 = nv;                                // first row whose region is > rg
                while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (region_of_row(m0 + mid) <= rg) l2 = mid + 1; else h2 = mid; }
                hi = l2;
            }
            const __amdgpu_buffer_rsrc_t so = em_rsrc(a.out + (m0 + lo) * EM_C, (long)(hi - lo) * EM_C * 4);
#pragma unroll 1
            for (int rb = 0; rb < 4; ++rb) {                     // 16-row blocks of the wave's 64 rows
                f32x4 acc[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = bias4[cb];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float b = xb[(4 * s + g) * EM_XS + 16 * rb];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[s][cb], b, acc[cb], 0, 0, 0);
                }
                const int row = 64 * rh + 16 * rb + c - lo;      // relative to the pass's first row: negative = before it (dropped: huge offset)
                const int voff = row < 0 ? 0x7ffffff0 : row * (EM_C * 4) + (64 * cq + 4 * g) * 4;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    f32x4 v = acc[cb];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * a.ns;
                    // (the column block's offset goes into the vector offset -- an immediate for the instruction -- not into soffset: a
                    // 16-byte buffer store with an SGPR soffset whose data registers the next instruction overwrites picks up the NEW
                    // value now and then on gfx950, and hipcc pads that hazard only when soffset is an immediate (gemm.hip, top).  With
                    // the row-block loop unrolled the next MFMA's destination WAS the store's data: wrong rows in half of the runs)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), so, voff + 64 * cb, 0, 0);
                }
            }
        }
        park_rows(buf ^ 1);                                      // the next tile's rows (requested above) into the other buffer
        buf ^= 1;
        rg_first = nrg_first; rg_last = nrg_last;
    }
}

}  // namespace

int fused_cus();

// (small problems keep the general core and its 64 x 64 tiles: a persistent kernel of 128-row tiles needs a few tiles per CU)
bool embed_fp32_ok(long M, int C, int F, int T) { return C == EM_C && F == EM_F && M >= 512L * EM_ROWS && T > 0; }

int launch_embed_fp32(const float* X, const float* LX, const float* A0, const float* Aall, const int* node_region, const float* bias,
                      float* out, long M, int T, int act, float slope, hipStream_t st) {
    REGT_CHECK_ARG(act == ACT_NONE || act == ACT_LRELU || act == ACT_RELU, "embedding kernel: activation %d not covered", act);
    EmbedArgs a{X, LX, A0, Aall, node_region, bias, out, M, T, act == ACT_NONE ? 1.0f : (act == ACT_LRELU ? slope : 0.0f)};
    static bool attr_done = false;
    if (const int rc = set_lds_once(&embed_fp32_kernel, EM_LDS, &attr_done)) return rc;
    const long tiles = (M + EM_ROWS - 1) / EM_ROWS;
    const long cus = fused_cus();
    hipLaunchKernelGGL(embed_fp32_kernel, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(512), EM_LDS, st, a);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
