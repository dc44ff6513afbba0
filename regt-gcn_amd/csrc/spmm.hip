// Neighbour aggregation: Y[r, :] = sum_e val[e] * X[col[e], :] over a destination-sorted CSR.
//
// This is the gather / scale / scatter-add of PyG's MessagePassing.propagate (the reference's
// GCNConv / ChebConv call sites) restated as a pull: every output row is owned by one lane group,
// so there are no float atomics and the sum runs in a fixed (edge) order.
//
// Mapping (CDNA4, wave = 64): a group of G lanes owns one row; each lane holds CH float4 column
// chunks, so a neighbour row is fetched with CH coalesced 16-B loads per lane (G*16 contiguous bytes
// per instruction).  The group first pulls up to G (col, val) pairs of its CSR segment with one
// coalesced load and then broadcasts them lane-to-lane (__shfl within the group) -- the segmented
// reduction never touches LDS or atomics.  Two edges are processed per step so that 2*CH gathers
// are in flight per lane before the first FMA.
//
// HBM-bound: algorithmic bytes = read X once + (col,val) + write Y.
#include <stdlib.h>

#include "kernels.h"

namespace regt {

// (N, F, T) time-innermost (the reference's snapshot layout, load_dataset.py:451-457) -> (N, T, F) rows.
__global__ void pack_x_kernel(const float* __restrict__ x, float* __restrict__ xp, long N, int F, int T) {
    const long total = N * F * T;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        long n = o / ((long)F * T);
        int rem = (int)(o - n * F * T);
        int t = rem / F, f = rem - t * F;
        xp[o] = x[n * F * T + (long)f * T + t];
    }
}

int launch_pack_x(const float* x, float* xp, int N, int F, int T, hipStream_t st) {
    long total = (long)N * F * T;
    int blocks = cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_x_kernel, dim3(blocks), dim3(256), 0, st, x, xp, (long)N, F, T);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

template <int G, int CH>
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ X,
                                                       float* __restrict__ Y, int nrows, int nrows_x, int W4, int ld4) {
    constexpr int GROUPS = 256 / G;
    const int gl = threadIdx.x % G;                // lane inside the group
    const int gid = threadIdx.x / G;
    const long W = (long)ld4 * 4;                  // row stride of X and Y; W4 = float4 columns handled by this launch
    for (long row = (long)blockIdx.x * GROUPS + gid; row < nrows; row += (long)gridDim.x * GROUPS) {
        float4 acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int beg = rowptr[row], end = rowptr[row + 1];
        for (int base = beg; base < end; base += G) {
            const int n = end - base < G ? end - base : G;
            int myc = 0;
            float myv = 0.f;
            if (gl < n) { myc = col[base + gl]; myv = val[base + gl]; }
            int e = 0;
            for (; e + 1 < n; e += 2) {
                const int c0 = __shfl(myc, e, G), c1 = __shfl(myc, e + 1, G);
                const float v0 = __shfl(myv, e, G), v1 = __shfl(myv, e + 1, G);
                const float4* x0 = reinterpret_cast<const float4*>(X + (long)c0 * W);
                const float4* x1 = reinterpret_cast<const float4*>(X + (long)c1 * W);
                float4 a[CH], b[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int ch = gl + c * G;
                    a[c] = ch < W4 ? x0[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                    b[c] = ch < W4 ? x1[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    acc[c].x = fmaf(v0, a[c].x, acc[c].x); acc[c].y = fmaf(v0, a[c].y, acc[c].y);
                    acc[c].z = fmaf(v0, a[c].z, acc[c].z); acc[c].w = fmaf(v0, a[c].w, acc[c].w);
                    acc[c].x = fmaf(v1, b[c].x, acc[c].x); acc[c].y = fmaf(v1, b[c].y, acc[c].y);
                    acc[c].z = fmaf(v1, b[c].z, acc[c].z); acc[c].w = fmaf(v1, b[c].w, acc[c].w);
                }
            }
            if (e < n) {
                const int c0 = __shfl(myc, e, G);
                const float v0 = __shfl(myv, e, G);
                const float4* x0 = reinterpret_cast<const float4*>(X + (long)c0 * W);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int ch = gl + c * G;
                    float4 a = ch < W4 ? x0[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                    acc[c].x = fmaf(v0, a.x, acc[c].x); acc[c].y = fmaf(v0, a.y, acc[c].y);
                    acc[c].z = fmaf(v0, a.z, acc[c].z); acc[c].w = fmaf(v0, a.w, acc[c].w);
                }
            }
        }
        float4* y = reinterpret_cast<float4*>(Y + row * W);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = gl + c * G;
            if (ch < W4) y[ch] = acc[c];
        }
    }
}

// ---- XCD-aware column-panel variant (large graphs) --------------------------------------------------
// The plain kernel above re-reads every neighbour row from the Infinity Cache: X (N*W*4 B, 154 MB at
// cfg-3) does not fit the 4 MiB L2 of an XCD, so the gather traffic is nnz*W*4 B (3.1 GB) and the
// kernel runs at the cache-fabric rate, not at the HBM rate.  Here the row is cut into 128-byte column
// panels (one cache line per neighbour per panel) and the work is laid out so that ALL workgroups
// resident on one XCD work on the same (node chunk, panel) at the same time: workgroup b lands on XCD
// b % 8 (round-robin dispatch -- used for speed only), XCD x owns the contiguous node chunk x (with
// region-contiguous node ids that is ~one region, whose neighbours are mostly inside the chunk) and
// walks panel 0, 1, ... over that chunk.  The live working set per XCD is chunk_nodes * 128 B
// (1.6 MB at cfg-3) and stays in L2, so each X line is fetched from HBM / MALL about once.
// Eight lanes own one (row, panel): 8 x 16 B = one 128-B line per neighbour; the 8 lanes pull 8
// (col, val) pairs with one coalesced load and broadcast them by shuffle, four gathers in flight.
template <int PL>     // lanes per row = panel width in float4: 8 = one 128-B line per neighbour, 16 = two (half the CSR re-reads)
__global__ __launch_bounds__(256) void spmm_panel_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const float* __restrict__ val, const float* __restrict__ X,
                                                         float* __restrict__ Y, int nnodes, int nstack, int W4,
                                                         int npanels, int nrb) {
    constexpr int ROWS = 256 / PL;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int per_panel = nstack * nrb;
    const int panel = li / per_panel;
    if (panel >= npanels) return;
    const int rem = li - panel * per_panel;
    const int s = rem / nrb, rb = rem - s * nrb;
    const int q = nnodes / 8, r8 = nnodes % 8;
    const int c0 = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int csz = q + (xcd < r8 ? 1 : 0);
    const int g = threadIdx.x / PL, gl = threadIdx.x % PL;
    if (rb * ROWS + g >= csz) return;                     // whole lane group leaves together
    const long row = (long)s * nnodes + c0 + rb * ROWS + g;
    const float4* X4 = reinterpret_cast<const float4*>(X) + panel * PL + gl;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rowptr[row], end = rowptr[row + 1];
    for (int base = beg; base < end; base += PL) {
        const int n = end - base < PL ? end - base : PL;
        int myc = 0;
        float myv = 0.f;
        if (gl < n) { myc = col[base + gl]; myv = val[base + gl]; }
#pragma unroll
        for (int h = 0; h < PL / 4; ++h) {
            if (h * 4 < n) {
                float4 x[4];
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {               // lanes >= n carry (col 0, val 0): harmless
                    const int c = __shfl(myc, h * 4 + e, PL);
                    v[e] = __shfl(myv, h * 4 + e, PL);
                    x[e] = X4[(long)c * W4];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc.x = fmaf(v[e], x[e].x, acc.x); acc.y = fmaf(v[e], x[e].y, acc.y);
                    acc.z = fmaf(v[e], x[e].z, acc.z); acc.w = fmaf(v[e], x[e].w, acc.w);
                }
            }
        }
    }
    // streaming stores: the output is not read again here and must not evict the XCD's slice of X from its L2
    float4* py = reinterpret_cast<float4*>(Y) + row * W4 + panel * PL + gl;
    __builtin_nontemporal_store(acc.x, &py->x); __builtin_nontemporal_store(acc.y, &py->y);
    __builtin_nontemporal_store(acc.z, &py->z); __builtin_nontemporal_store(acc.w, &py->w);
}

// ---- dual-operator panel variant ----------------------------------------------------------------------
// The regional Laplacian rows are (almost) a subset of the full-graph rows: the same neighbour row
// x[col] feeds both A_hat x and L~ x.  With a merged CSR that carries two weights per entry, one gather
// serves both outputs -- half the gather volume of the stacked operator.  Same XCD/panel schedule as above.
template <int PL, int IDX, bool NTS>   // PL lanes per row = panel width in float4 (8: one 128-B line per neighbour, 16: two);
                                        // IDX (<= PL) CSR entries fetched per index load; NTS: streaming output stores
__global__ __launch_bounds__(256) void spmm_dual_panel_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                              const float* __restrict__ val_a, const float* __restrict__ val_l,
                                                              const float* __restrict__ X, float* __restrict__ YA,
                                                              float* __restrict__ YL, int nnodes, int W4, int npanels, int nrb) {
    static_assert(IDX % 8 == 0 && IDX <= PL, "index chunk is a multiple of the 8-gather round and fits the lane group");
    constexpr int ROWS = 256 / PL;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int panel = li / nrb;
    if (panel >= npanels) return;
    const int rb = li - panel * nrb;
    const int q = nnodes / 8, r8 = nnodes % 8;
    const int c0 = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int csz = q + (xcd < r8 ? 1 : 0);
    const int g = threadIdx.x / PL, gl = threadIdx.x % PL;
    if (rb * ROWS + g >= csz) return;
    const long row = c0 + rb * ROWS + g;
    const float4* X4 = reinterpret_cast<const float4*>(X) + panel * PL + gl;
    float4 aa = make_float4(0.f, 0.f, 0.f, 0.f), al = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rowptr[row], end = rowptr[row + 1];
    for (int base = beg; base < end; base += IDX) {
        const int n = end - base < IDX ? end - base : IDX;
        int myc = 0;
        float mya = 0.f, myl = 0.f;
        if (gl < n) { myc = col[base + gl]; mya = val_a[base + gl]; myl = val_l[base + gl]; }
#pragma unroll
        for (int r = 0; r < IDX / 8; ++r) {
            if (r * 8 < n) {
                // all 8 gathers of the round in flight before the first FMA; entries >= n carry (col 0, weights 0)
                float4 x[8];
                float va[8], vl[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = __shfl(myc, r * 8 + e, PL);
                    va[e] = __shfl(mya, r * 8 + e, PL);
                    vl[e] = __shfl(myl, r * 8 + e, PL);
                    x[e] = r * 8 + e < n ? X4[(long)c * W4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    aa.x = fmaf(va[e], x[e].x, aa.x); aa.y = fmaf(va[e], x[e].y, aa.y);
                    aa.z = fmaf(va[e], x[e].z, aa.z); aa.w = fmaf(va[e], x[e].w, aa.w);
                    al.x = fmaf(vl[e], x[e].x, al.x); al.y = fmaf(vl[e], x[e].y, al.y);
                    al.z = fmaf(vl[e], x[e].z, al.z); al.w = fmaf(vl[e], x[e].w, al.w);
                }
            }
        }
    }
    float4* pa = reinterpret_cast<float4*>(YA) + row * W4 + panel * PL + gl;
    float4* pl = reinterpret_cast<float4*>(YL) + row * W4 + panel * PL + gl;
    if (NTS) {
        // the outputs are not read again by this kernel: streaming stores keep them from evicting the XCD's slice of X
        // from its L2 (cfg-3, X resident in the Infinity Cache: 186 -> 149 us, L2 hits 64 -> 70 %; inside a training
        // step, where X comes from HBM, 174 -> 167 us)
        __builtin_nontemporal_store(aa.x, &pa->x); __builtin_nontemporal_store(aa.y, &pa->y);
        __builtin_nontemporal_store(aa.z, &pa->z); __builtin_nontemporal_store(aa.w, &pa->w);
        __builtin_nontemporal_store(al.x, &pl->x); __builtin_nontemporal_store(al.y, &pl->y);
        __builtin_nontemporal_store(al.z, &pl->z); __builtin_nontemporal_store(al.w, &pl->w);
    } else {
        *pa = aa;
        *pl = al;
    }
}

int launch_spmm_dual(const int* rowptr, const int* col, const float* val_a, const float* val_l, const float* X, float* YA,
                     float* YL, int nnodes, int W, hipStream_t st) {
    REGT_CHECK_ARG(nnodes > 0 && W > 0 && W % 32 == 0, "spmm_dual: width %d must be a multiple of 32 floats", W);
    const int W4 = W / 4;
    static int pl_env = -1;
    if (pl_env < 0) { const char* e = getenv("REGT_SPMM_PL"); pl_env = e ? atoi(e) : 0; }
    // two cache lines per neighbour (halves the CSR re-reads) while the XCD's slice of X (chunk x 256 B) stays
    // around the 4 MiB L2: measured 182 vs 197 us at cfg-3 (3.2 MB slice)
    const bool wide = pl_env ? pl_env == 16 : (W4 % 16 == 0 && (long)cdiv(nnodes, 8) * 256 <= (7L << 19));
    const int PL = wide ? 16 : 8;
    const int npanels = W4 / PL;
    const int nrb = cdiv(cdiv(nnodes, 8), 256 / PL);
    const long grid = 8L * npanels * nrb;
    REGT_CHECK_ARG(grid < (1L << 31), "spmm_dual: grid too large");
    static int nt_env = -1, idx_env = -1;
    if (nt_env < 0) { const char* e = getenv("REGT_SPMM_NT"); nt_env = e ? atoi(e) : 1; }
    if (idx_env < 0) { const char* e = getenv("REGT_SPMM_IDX"); idx_env = e ? atoi(e) : 16; }
#define REGT_DUAL(PLL, IDXX, NTT)                                                                                              \
    hipLaunchKernelGGL((spmm_dual_panel_kernel<PLL, IDXX, NTT>), dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val_a, val_l, \
                       X, YA, YL, nnodes, W4, npanels, nrb)
    if (wide) {
        if (nt_env == 0) REGT_DUAL(16, 8, false);
        else if (idx_env == 8) REGT_DUAL(16, 8, true);
        else REGT_DUAL(16, 16, true);
    } else {
        if (nt_env == 0) REGT_DUAL(8, 8, false);
        else REGT_DUAL(8, 8, true);
    }
#undef REGT_DUAL
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

template <int G, int CH>
static int launch_spmm_t(const int* rowptr, const int* col, const float* val, const float* X, float* Y, int nrows,
                         int nrows_x, int W4, int ld4, hipStream_t st) {
    constexpr int GROUPS = 256 / G;
    long blocks = ((long)nrows + GROUPS - 1) / GROUPS;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipLaunchKernelGGL((spmm_csr_kernel<G, CH>), dim3((unsigned)blocks), dim3(256), 0, st, rowptr, col, val, X, Y, nrows,
                       nrows_x, W4, ld4);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_spmm_csr(const int* rowptr, const int* col, const float* val, const float* X, float* Y, int nrows,
                    int nrows_x, int W, int nstack, hipStream_t st) {
    REGT_CHECK_ARG(nrows > 0 && W > 0, "spmm: empty problem");
    REGT_CHECK_ARG(W % 4 == 0, "spmm: row width %d must be a multiple of 4 floats", W);
    REGT_CHECK_ARG(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0, "spmm: X/Y must be 16-B aligned");
    REGT_CHECK_ARG(nstack >= 1 && nrows % nstack == 0, "spmm: nrows=%d not a multiple of nstack=%d", nrows, nstack);
    const int W4 = W / 4;
    // Panel variant when the X matrix cannot live in the XCDs' L2s anyway and rows are whole cache lines.
    if (W4 % 8 == 0 && (long)nrows_x * W * 4 > (24L << 20) && nrows / nstack >= 4096) {
        const int nnodes = nrows / nstack;
        // 256-byte panels (half the passes over the CSR) while an XCD's slice of X (chunk x 256 B) stays around its 4 MiB L2
        static int pl_env = -1;
        if (pl_env < 0) { const char* e = getenv("REGT_SPMM_PL"); pl_env = e ? atoi(e) : 0; }
        const bool wide = pl_env ? pl_env == 16 : (W4 % 16 == 0 && (long)cdiv(nnodes, 8) * 256 <= (7L << 19));
        const int PL = wide ? 16 : 8;
        const int npanels = W4 / PL;
        const int nrb = cdiv(cdiv(nnodes, 8), 256 / PL);
        const long grid = 8L * npanels * nstack * nrb;
        REGT_CHECK_ARG(grid < (1L << 31), "spmm: grid too large");
        if (wide)
            hipLaunchKernelGGL(spmm_panel_kernel<16>, dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val, X, Y, nnodes, nstack, W4,
                               npanels, nrb);
        else
            hipLaunchKernelGGL(spmm_panel_kernel<8>, dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val, X, Y, nnodes, nstack, W4,
                               npanels, nrb);
        REGT_CHECK_LAUNCH();
        return REGT_OK;
    }
    // rows wider than 2048 floats (hidden-state aggregation of the stacked-conv baseline: T * 512): column passes
    if (W4 > 512) {
        for (int c4 = 0; c4 < W4; c4 += 512) {
            const int w4 = W4 - c4 < 512 ? W4 - c4 : 512;
            if (int rc = launch_spmm_t<64, 8>(rowptr, col, val, X + 4L * c4, Y + 4L * c4, nrows, nrows_x, w4, W4, st)) return rc;
        }
        return REGT_OK;
    }
#define REGT_SPMM(G, CH) return launch_spmm_t<G, CH>(rowptr, col, val, X, Y, nrows, nrows_x, W4, W4, st)
    if (W4 <= 8) REGT_SPMM(8, 1);
    if (W4 <= 16) REGT_SPMM(16, 1);
    if (W4 <= 32) REGT_SPMM(32, 1);
    if (W4 <= 64) REGT_SPMM(64, 1);
    if (W4 <= 96) REGT_SPMM(32, 3);
    if (W4 <= 128) REGT_SPMM(64, 2);
    if (W4 <= 192) REGT_SPMM(64, 3);
    if (W4 <= 256) REGT_SPMM(64, 4);
    if (W4 <= 512) REGT_SPMM(64, 8);
#undef REGT_SPMM
    set_error("spmm: row width %d floats not supported", W);
    return REGT_ERR_ARG;
}

}  // namespace regt
