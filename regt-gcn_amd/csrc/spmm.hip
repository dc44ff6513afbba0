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

// The same transposition staged through LDS: a workgroup takes NB whole nodes (NB * F * T floats, contiguous on both sides), reads
// them with coalesced 16-byte loads, and writes 16-byte pieces of (node, period) rows.  The snapshot is read exactly once ->
// non-temporal loads, so that the lines of x do not displace the packed rows (which the aggregation reads next) from the
// Infinity Cache.  F % 4 == 0, NB * F * T <= PACK_LDS_FLOATS.
constexpr int PACK_LDS_FLOATS = 4096;
__global__ __launch_bounds__(256) void pack_x_lds_kernel(const float* __restrict__ x, float* __restrict__ xp, int N, int F, int T, int NB) {
    __shared__ float4 tile4[PACK_LDS_FLOATS / 4];
    float* tile = reinterpret_cast<float*>(tile4);
    const int FT = F * T, F4 = F / 4;
    for (long n0 = (long)blockIdx.x * NB; n0 < N; n0 += (long)gridDim.x * NB) {
        const int nb = N - n0 < NB ? (int)(N - n0) : NB;
        const int tot4 = nb * FT / 4;                    // FT % 4 == 0 since F % 4 == 0
        const float4* src = reinterpret_cast<const float4*>(x + n0 * FT);
        for (int i = threadIdx.x; i < tot4; i += 256) {
            float4 v;
            const float* s = reinterpret_cast<const float*>(src + i);
            v.x = __builtin_nontemporal_load(s); v.y = __builtin_nontemporal_load(s + 1);
            v.z = __builtin_nontemporal_load(s + 2); v.w = __builtin_nontemporal_load(s + 3);
            tile4[i] = v;
        }
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(xp + n0 * FT);
        for (int o = threadIdx.x; o < tot4; o += 256) {  // o = (node, t, f4)
            const int n = o / (T * F4), rem = o - n * T * F4;
            const int t = rem / F4, f = (rem - t * F4) * 4;
            const float* b = tile + n * FT + f * T + t;
            dst[o] = make_float4(b[0], b[T], b[2 * T], b[3 * T]);      // (ordinary stores: the packed rows are read next)
        }
        __syncthreads();
    }
}

// LDS-staged with non-temporal loads of the snapshot (round 4: profiles/r04_pack_ab.txt -- the other variants measured there are gone);
// rows wider than the LDS tile or unaligned pointers take the element-wise kernel
int launch_pack_x(const float* x, float* xp, int N, int F, int T, hipStream_t st) {
    long total = (long)N * F * T;
    const int FT = F * T;
    if (F % 4 == 0 && FT <= PACK_LDS_FLOATS && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xp)) & 15) == 0) {
        const int NB = PACK_LDS_FLOATS / FT;
        long blocks = cdiv((long)N, NB);
        if (blocks > 256L * 16) blocks = 256L * 16;
        hipLaunchKernelGGL(pack_x_lds_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, xp, N, F, T, NB);
        REGT_CHECK_LAUNCH();
        return REGT_OK;
    }
    int blocks = cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_x_kernel, dim3(blocks), dim3(256), 0, st, x, xp, (long)N, F, T);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// bf16 rows for the bf16 arithmetic (REGT_GEMM_MODE=bf16): x, A_hat x and L~ x only ever feed matrix-core operands there, so
// the snapshot is rounded once while it is packed (round to nearest even) and the aggregation reads and writes bf16 rows.
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk2(float a, float b) {
    const pk_f32x2 p = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(p, pk_bf16x2));
}
// (N, F, T) fp32 -> (N, T, F) bf16; F % 8 == 0: a thread writes the 16 bytes of 8 consecutive features of one (node, period)
__global__ void pack_x_bf16_kernel(const float* __restrict__ x, uint4* __restrict__ xp, long N, int F, int T) {
    const int f8 = F / 8;
    const long total = N * T * f8;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long nt = o / f8;
        const int f0 = (int)(o - nt * f8) * 8;
        const long n = nt / T;
        const int t = (int)(nt - n * T);
        const float* s = x + n * F * T + (long)f0 * T + t;
        uint4 v;
#define REGT_NTL(k) __builtin_nontemporal_load(s + (k) * (long)T)      /* the snapshot is read once: keep it out of the caches' way */
        v.x = pk2(REGT_NTL(0), REGT_NTL(1)); v.y = pk2(REGT_NTL(2), REGT_NTL(3)); v.z = pk2(REGT_NTL(4), REGT_NTL(5)); v.w = pk2(REGT_NTL(6), REGT_NTL(7));
#undef REGT_NTL
        xp[o] = v;
    }
}
// LDS-staged form of the same (see pack_x_lds_kernel): coalesced non-temporal 16-byte reads of whole nodes, 16-byte bf16 pieces out
__global__ __launch_bounds__(256) void pack_x_bf16_lds_kernel(const float* __restrict__ x, uint4* __restrict__ xp, int N, int F, int T, int NB) {
    __shared__ float4 tile4[PACK_LDS_FLOATS / 4];
    const float* tile = reinterpret_cast<const float*>(tile4);
    const int FT = F * T, F8 = F / 8;
    for (long n0 = (long)blockIdx.x * NB; n0 < N; n0 += (long)gridDim.x * NB) {
        const int nb = N - n0 < NB ? (int)(N - n0) : NB;
        const int tot4 = nb * FT / 4, tot8 = nb * FT / 8;
        const float4* src = reinterpret_cast<const float4*>(x + n0 * FT);
        for (int i = threadIdx.x; i < tot4; i += 256) {
            const float* s = reinterpret_cast<const float*>(src + i);
            tile4[i] = make_float4(__builtin_nontemporal_load(s), __builtin_nontemporal_load(s + 1), __builtin_nontemporal_load(s + 2),
                                   __builtin_nontemporal_load(s + 3));
        }
        __syncthreads();
        uint4* dst = xp + n0 * (FT / 8);
        for (int o = threadIdx.x; o < tot8; o += 256) {  // o = (node, t, f8)
            const int n = o / (T * F8), rem = o - n * T * F8;
            const int t = rem / F8, f = (rem - t * F8) * 8;
            const float* b = tile + n * FT + f * T + t;
            dst[o] = make_uint4(pk2(b[0], b[T]), pk2(b[2 * T], b[3 * T]), pk2(b[4 * T], b[5 * T]), pk2(b[6 * T], b[7 * T]));
        }
        __syncthreads();
    }
}
int launch_pack_x_bf16(const float* x, void* xp, int N, int F, int T, hipStream_t st) {
    REGT_CHECK_ARG(F % 8 == 0, "pack_x (bf16 rows): F = %d must be a multiple of 8", F);
    const int FT = F * T;
    if (FT <= PACK_LDS_FLOATS && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xp)) & 15) == 0) {
        const int NB = PACK_LDS_FLOATS / FT;
        long blocks = cdiv((long)N, NB);
        if (blocks > 256L * 16) blocks = 256L * 16;
        hipLaunchKernelGGL(pack_x_bf16_lds_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, reinterpret_cast<uint4*>(xp), N, F, T, NB);
        REGT_CHECK_LAUNCH();
        return REGT_OK;
    }
    const long total = (long)N * T * (F / 8);
    int blocks = cdiv(total, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pack_x_bf16_kernel, dim3(blocks), dim3(256), 0, st, x, reinterpret_cast<uint4*>(xp), (long)N, F, T);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
// n8 groups of 8 fp32 -> 8 bf16 (packed rows handed over as fp32 by a caller of regt_forward_packed).  Read once: non-temporal loads, so
// that the fp32 rows do not displace the bf16 rows (read next by the aggregation) from the Infinity Cache.
__global__ void cvt_rows_bf16_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, long n8) {
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n8; o += (long)gridDim.x * blockDim.x) {
        const float* s = reinterpret_cast<const float*>(src + 2 * o);
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_nontemporal_load(s + i);
        dst[o] = make_uint4(pk2(a[0], a[1]), pk2(a[2], a[3]), pk2(a[4], a[5]), pk2(a[6], a[7]));
    }
}
int launch_cvt_rows_bf16(const float* src, void* dst, long n, hipStream_t st) {
    REGT_CHECK_ARG(n % 8 == 0, "cvt_rows_bf16: element count must be a multiple of 8");
    int blocks = cdiv(n / 8, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(cvt_rows_bf16_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(src), reinterpret_cast<uint4*>(dst), n / 8);
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

// The same row-per-lane-group pull with the MERGED operator (two weights per entry, both outputs from one gather): rows of any
// width that is a multiple of 4 floats -- the case the panel kernels (whole 128-byte panels, W % 32 == 0) do not cover, e.g. the
// reference's own TPIMS widths T * F = 48 and 96.  Small graphs only (X lives in the L2s), so no panel schedule is needed.
template <int G, int CH>
__global__ __launch_bounds__(256) void spmm_dual_csr_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                            const float* __restrict__ val_a, const float* __restrict__ val_l,
                                                            const float* __restrict__ X, float* __restrict__ YA, float* __restrict__ YL,
                                                            int nrows, int W4) {
    constexpr int GROUPS = 256 / G;
    const int gl = threadIdx.x % G, gid = threadIdx.x / G;
    const long W = (long)W4 * 4;
    for (long row = (long)blockIdx.x * GROUPS + gid; row < nrows; row += (long)gridDim.x * GROUPS) {
        float4 aa[CH], al[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) aa[c] = al[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int beg = rowptr[row], end = rowptr[row + 1];
        for (int base = beg; base < end; base += G) {
            const int n = end - base < G ? end - base : G;
            int myc = 0;
            float mya = 0.f, myl = 0.f;
            if (gl < n) { myc = col[base + gl]; mya = val_a[base + gl]; myl = val_l[base + gl]; }
            for (int e = 0; e < n; ++e) {
                const int c0 = __shfl(myc, e, G);
                const float va = __shfl(mya, e, G), vl = __shfl(myl, e, G);
                const float4* x0 = reinterpret_cast<const float4*>(X + (long)c0 * W);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int ch = gl + c * G;
                    const float4 a = ch < W4 ? x0[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
                    aa[c].x = fmaf(va, a.x, aa[c].x); aa[c].y = fmaf(va, a.y, aa[c].y); aa[c].z = fmaf(va, a.z, aa[c].z); aa[c].w = fmaf(va, a.w, aa[c].w);
                    al[c].x = fmaf(vl, a.x, al[c].x); al[c].y = fmaf(vl, a.y, al[c].y); al[c].z = fmaf(vl, a.z, al[c].z); al[c].w = fmaf(vl, a.w, al[c].w);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = gl + c * G;
            if (ch < W4) {
                reinterpret_cast<float4*>(YA + row * W)[ch] = aa[c];
                reinterpret_cast<float4*>(YL + row * W)[ch] = al[c];
            }
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
// 256-byte panels (half the passes over the CSR, 16 lanes per row) while an XCD's slice of X (chunk x 256 B) stays NEAR its 4 MiB L2:
// measured faster up to a 4.0 MB slice (cfg-5 shard, 15 625 nodes per XCD: 0.27 vs 0.38 ms with 128-byte panels, round 3; the
// round-1 limit of 3.7 MB was a guess that left this case on 128-byte panels)
constexpr long SP_WIDE_SLICE_MAX = 9L << 19;     // 4.7 MB
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
template <int PL, int IDX>              // PL lanes per row = panel width in float4 (8: one 128-B line per neighbour, 16: two);
                                        // IDX (<= PL) CSR entries fetched per index load
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
    {
        // the outputs are not read again by this kernel: streaming stores keep them from evicting the XCD's slice of X
        // from its L2 (cfg-3, X resident in the Infinity Cache: 186 -> 149 us, L2 hits 64 -> 70 %; inside a training
        // step, where X comes from HBM, 174 -> 167 us)
        __builtin_nontemporal_store(aa.x, &pa->x); __builtin_nontemporal_store(aa.y, &pa->y);
        __builtin_nontemporal_store(aa.z, &pa->z); __builtin_nontemporal_store(aa.w, &pa->w);
        __builtin_nontemporal_store(al.x, &pl->x); __builtin_nontemporal_store(al.y, &pl->y);
        __builtin_nontemporal_store(al.z, &pl->z); __builtin_nontemporal_store(al.w, &pl->w);
    }
}

// ---- 16 bytes per lane of either row format: 4 fp32 or 8 bf16 elements; fp32 accumulation, one rounding at the end ----------
struct SpEnt { int col; float wa, wl; int pad; };
template <bool BF> struct SpAcc { float v[BF ? 8 : 4]; };
template <bool BF>
__device__ __forceinline__ void sp_fma(SpAcc<BF>& a, float w, const u32x4_t& x) {
    if constexpr (BF) {
        a.v[0] = fmaf(w, __uint_as_float(x.x << 16), a.v[0]); a.v[1] = fmaf(w, __uint_as_float(x.x & 0xffff0000u), a.v[1]);
        a.v[2] = fmaf(w, __uint_as_float(x.y << 16), a.v[2]); a.v[3] = fmaf(w, __uint_as_float(x.y & 0xffff0000u), a.v[3]);
        a.v[4] = fmaf(w, __uint_as_float(x.z << 16), a.v[4]); a.v[5] = fmaf(w, __uint_as_float(x.z & 0xffff0000u), a.v[5]);
        a.v[6] = fmaf(w, __uint_as_float(x.w << 16), a.v[6]); a.v[7] = fmaf(w, __uint_as_float(x.w & 0xffff0000u), a.v[7]);
    } else {
        a.v[0] = fmaf(w, __uint_as_float(x.x), a.v[0]); a.v[1] = fmaf(w, __uint_as_float(x.y), a.v[1]);
        a.v[2] = fmaf(w, __uint_as_float(x.z), a.v[2]); a.v[3] = fmaf(w, __uint_as_float(x.w), a.v[3]);
    }
}
template <bool BF>
__device__ __forceinline__ u32x4_t sp_pack(const SpAcc<BF>& a) {
    if constexpr (BF) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        u32x4_t r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2_t p = {a.v[2 * i], a.v[2 * i + 1]};
            r[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2_t));
        }
        return r;
    } else {
        return u32x4_t{__float_as_uint(a.v[0]), __float_as_uint(a.v[1]), __float_as_uint(a.v[2]), __float_as_uint(a.v[3])};
    }
}

// The dual-operator panel kernel on bf16 rows (REGT_GEMM_MODE=bf16: x, A_hat x and L~ x only ever feed bf16 matrix-core operands):
// same schedule, a lane holds 8 elements of its 16 bytes, sums in fp32 in CSR order, outputs rounded once (nearest even).
template <int PL, int IDX>
__global__ __launch_bounds__(256, 5) void spmm_dual_panel_bf16_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                   const float* __restrict__ val_a, const float* __restrict__ val_l,
                                                                   const uint4* __restrict__ X, uint4* __restrict__ YA,
                                                                   uint4* __restrict__ YL, int nnodes, int W16, int npanels, int nrb,
                                                                   unsigned x_bytes) {
    static_assert(IDX % 8 == 0 && IDX <= PL, "index chunk is a multiple of the 8-gather round and fits the lane group");
    constexpr int ROWS = 256 / PL;
    // rows of X through a buffer descriptor: 32-bit byte offsets (host-checked), no 64-bit address arithmetic per gather
    const __amdgpu_buffer_rsrc_t sx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(X), 0, x_bytes, 0x00020000);
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
    const unsigned lane_off = (unsigned)(panel * PL + gl) * 16u, rowbytes = (unsigned)W16 * 16u;
    SpAcc<true> aa, al;
#pragma unroll
    for (int i = 0; i < 8; ++i) { aa.v[i] = 0.f; al.v[i] = 0.f; }
    const int beg = rowptr[row], end = rowptr[row + 1];
    for (int base = beg; base < end; base += IDX) {
        const int n = end - base < IDX ? end - base : IDX;
        int myc = 0;
        float mya = 0.f, myl = 0.f;
        if (gl < n) { myc = col[base + gl]; mya = val_a[base + gl]; myl = val_l[base + gl]; }
#pragma unroll
        for (int r = 0; r < IDX / 8; ++r) {
            if (r * 8 < n) {
                u32x4_t x[8];
                float va[8], vl[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = __shfl(myc, r * 8 + e, PL);
                    va[e] = __shfl(mya, r * 8 + e, PL);
                    vl[e] = __shfl(myl, r * 8 + e, PL);
                    x[e] = r * 8 + e < n ? __builtin_amdgcn_raw_buffer_load_b128(sx, (unsigned)c * rowbytes + lane_off, 0, 0) : u32x4_t{0, 0, 0, 0};
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sp_fma<true>(aa, va[e], x[e]);
                    sp_fma<true>(al, vl[e], x[e]);
                }
            }
        }
    }
    const u32x4_t pa = sp_pack<true>(aa), pl = sp_pack<true>(al);
    unsigned* oa = reinterpret_cast<unsigned*>(YA + row * W16 + panel * PL + gl);
    unsigned* ol = reinterpret_cast<unsigned*>(YL + row * W16 + panel * PL + gl);
    __builtin_nontemporal_store(pa.x, oa); __builtin_nontemporal_store(pa.y, oa + 1);
    __builtin_nontemporal_store(pa.z, oa + 2); __builtin_nontemporal_store(pa.w, oa + 3);
    __builtin_nontemporal_store(pl.x, ol); __builtin_nontemporal_store(pl.y, ol + 1);
    __builtin_nontemporal_store(pl.z, ol + 2); __builtin_nontemporal_store(pl.w, ol + 3);
}

// ---- row-block variant: the CSR entries of a workgroup's rows live in LDS, the workgroup walks ALL column panels ------------
// The panel kernels above read a row's (col, weight, weight) entries again for every panel: 6 x 13 MB at cfg-3, and those
// streams compete with the X slice for the XCD's 4 MiB L2.  Here a workgroup owns GROUPS x rpg consecutive-ish rows of its
// XCD's node chunk for the whole launch: it copies their entries to LDS once (one coalesced pass over the CSR: 13 MB in total)
// and then walks panel 0, 1, ... over them.  rpg is chosen on the host so that ALL workgroups of an XCD are resident at
// once (8 per CU): they start together and advance through the panels at about the same pace, so the live slice of X per
// XCD is still ~one panel wide (speed only -- nothing depends on the pacing).  A lane group of PL lanes owns one row at a time;
// it reads entries back with broadcast ds_read_b128 (no shuffles) and keeps 8 gathers in flight.  Rows of X / Y are addressed
// through buffer descriptors: 32-bit byte offsets, the panel offset is the instruction's scalar offset.
// BF: rows of X and of both outputs hold bf16 (16 bytes = 8 elements per lane), accumulation in fp32, one rounding at the end.
// LDS entries per workgroup (16 B each, behind the 260-int row pointer slice): what 8 (fp32 rows, 64 VGPRs) or 5 (bf16 rows, 84 VGPRs)
// resident workgroups per CU leave each of them
static int rows_cap(bool bf) { return (163840 / (bf ? 5 : 8) - 260 * 4 - 64) / 16; }

template <int PL, bool DUAL, bool BF>
__global__ __launch_bounds__(256, BF ? 5 : 8) void spmm_rows_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                        const float* __restrict__ val_a, const float* __restrict__ val_l,
                                                        const void* __restrict__ X, void* __restrict__ YA, void* __restrict__ YL,
                                                        int nnodes, unsigned x_bytes, unsigned y_bytes, int rowbytes, int npanels, int rpg, int cap) {
    constexpr int GROUPS = 256 / PL;
    extern __shared__ __attribute__((aligned(16))) int sp_lds[];
    int* rp = sp_lds;
    SpEnt* ents = reinterpret_cast<SpEnt*>(sp_lds + 260);
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q = nnodes / 8, r8 = nnodes % 8;
    const int c0 = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int csz = q + (xcd < r8 ? 1 : 0);
    const int rows_wg = GROUPS * rpg;
    const int rb = li * rows_wg;                       // first row of the workgroup inside the chunk
    if (rb >= csz) return;
    const int nrows = csz - rb < rows_wg ? csz - rb : rows_wg;
    const int row0 = c0 + rb;
    const int tid = threadIdx.x, g = tid / PL, gl = tid % PL;
    const int first = rowptr[row0], last = rowptr[row0 + nrows];
    const int ntile = last - first;
    const bool in_lds = ntile <= cap;          // (workgroup-uniform) else: entries are read from global memory again per panel
    if (tid <= nrows) rp[tid] = rowptr[row0 + tid] - first;
    if (in_lds) {
        for (int i = tid; i < ntile; i += 256) {
            SpEnt e;
            e.col = col[first + i]; e.wa = val_a[first + i]; e.wl = DUAL ? val_l[first + i] : 0.f; e.pad = 0;
            ents[i] = e;
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t sx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(YA, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sl = __builtin_amdgcn_make_buffer_rsrc(DUAL ? YL : YA, 0, y_bytes, 0x00020000);
    for (int p = 0; p < npanels; ++p) {
        const int soff = p * PL * 16;
        for (int k = 0; k < rpg; ++k) {
            const int rl = g + GROUPS * k;
            if (rl >= nrows) break;                     // whole lane group leaves together
            const int beg = rp[rl], end = rp[rl + 1];
            SpAcc<BF> aa, al;
#pragma unroll
            for (int i = 0; i < (BF ? 8 : 4); ++i) { aa.v[i] = 0.f; al.v[i] = 0.f; }
            for (int e = beg; e < end; e += 8) {
                u32x4_t x[8];
                float wa[8], wl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = e + j < end;
                    SpEnt en{0, 0.f, 0.f, 0};
                    if (ok) {
                        if (in_lds) en = ents[e + j];
                        else { en.col = col[first + e + j]; en.wa = val_a[first + e + j]; en.wl = DUAL ? val_l[first + e + j] : 0.f; }
                    }
                    wa[j] = en.wa; wl[j] = en.wl;
                    x[j] = ok ? __builtin_amdgcn_raw_buffer_load_b128(sx, (unsigned)en.col * (unsigned)rowbytes + gl * 16, soff, 0) : u32x4_t{0, 0, 0, 0};
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sp_fma<BF>(aa, wa[j], x[j]);
                    if (DUAL) sp_fma<BF>(al, wl[j], x[j]);
                }
            }
            // streaming stores: the outputs must not evict the XCD's slice of X from its L2 (aux bit 1 = nt)
            const unsigned yo = (unsigned)(row0 + rl) * (unsigned)rowbytes + gl * 16 + soff;
            __builtin_amdgcn_raw_buffer_store_b128(sp_pack<BF>(aa), sa, yo, 0, 2);
            if (DUAL) __builtin_amdgcn_raw_buffer_store_b128(sp_pack<BF>(al), sl, yo, 0, 2);
        }
    }
}

// host side of the row-block kernel: eligibility and the rows-per-group choice
static int rows_rpg(int nnodes, int PL, bool bf) {
    const int chunk = cdiv(nnodes, 8), groups = 256 / PL;
    int rpg = cdiv(chunk, groups * (bf ? 150 : 240));     // all workgroups of an XCD resident at once: 32 CUs x 8 (5) workgroups, some slack
    if (rpg < 1) rpg = 1;
    return rpg;
}
template <bool DUAL, bool BF>
static int launch_spmm_rows(const int* rowptr, const int* col, const float* val_a, const float* val_l, const void* X, void* YA, void* YL,
                            int nnodes, long x_rows, int rowbytes, int PL, hipStream_t st) {
    const int npanels = rowbytes / (PL * 16);
    const int rpg = rows_rpg(nnodes, PL, BF);
    const int cap = rows_cap(BF);
    const int nrb = cdiv(cdiv(nnodes, 8), (256 / PL) * rpg);
    const long grid = 8L * nrb;
    const unsigned xb = (unsigned)(x_rows * rowbytes), yb = (unsigned)((long)nnodes * rowbytes);
    const int lds = cap * 16 + 260 * 4;
    if (PL == 16)
        hipLaunchKernelGGL((spmm_rows_kernel<16, DUAL, BF>), dim3((unsigned)grid), dim3(256), lds, st, rowptr, col, val_a, val_l, X, YA, YL,
                           nnodes, xb, yb, rowbytes, npanels, rpg, cap);
    else
        hipLaunchKernelGGL((spmm_rows_kernel<8, DUAL, BF>), dim3((unsigned)grid), dim3(256), lds, st, rowptr, col, val_a, val_l, X, YA, YL,
                           nnodes, xb, yb, rowbytes, npanels, rpg, cap);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
// 256-byte column panels (two cache lines per neighbour: half the passes over the CSR) while an XCD's slice of X (node chunk x 256 B)
// stays around its 4 MiB L2, else 128-byte ones; REGT_SPMM_PL=8|16 forces one (tools/spmm_pmc.sh)
static bool panel_wide(int nnodes) {
    static int pl_env = -1;
    if (pl_env < 0) { const char* e = getenv("REGT_SPMM_PL"); pl_env = e ? atoi(e) : 0; }
    return pl_env ? pl_env == 16 : (long)cdiv(nnodes, 8) * 256 <= SP_WIDE_SLICE_MAX;
}
// REGT_SPMM_ROWS / regt_set_option("spmm_rows"): 1 = row-block kernel where eligible, 0 (default) = panel kernels
static int g_rows_opt = -1;
static bool rows_wanted() {
    if (g_rows_opt < 0) { const char* e = getenv("REGT_SPMM_ROWS"); g_rows_opt = e ? atoi(e) : 0; }   // opt-in: measured slower
    return g_rows_opt != 0;
}
int spmm_rows_option(int value) {       // regt_set_option("spmm_rows", v): returns the previous setting
    const int prev = rows_wanted() ? 1 : 0;
    g_rows_opt = value ? 1 : 0;
    return prev;
}
// eligible: rows are whole panels, every byte offset fits 32 bits, at most 256 rows per workgroup
static bool rows_ok(int nnodes, long x_rows, long rowbytes, int PL) {
    return rows_wanted() && rowbytes % (PL * 16) == 0 && x_rows * rowbytes < (1L << 32) - 4096 && (long)nnodes * rowbytes < (1L << 32) - 4096 &&
           (256 / PL) * rows_rpg(nnodes, PL, false) <= 256;      // the row pointer slice in LDS holds 256 rows
}

int launch_spmm_dual_bf16(const int* rowptr, const int* col, const float* val_a, const float* val_l, const void* X, void* YA, void* YL,
                          int nnodes, int x_rows, int W, hipStream_t st) {
    REGT_CHECK_ARG(nnodes > 0 && W > 0 && W % 64 == 0, "spmm_dual (bf16 rows): width %d must be a multiple of 64 elements", W);
    const long rowbytes = 2L * W;
    const bool wide = panel_wide(nnodes) && rowbytes % 256 == 0;
    const int PL = wide ? 16 : 8;
    REGT_CHECK_ARG(x_rows * rowbytes < (1L << 32) - 4096, "spmm_dual (bf16 rows): X larger than 4 GB");
    if (rows_wanted() && (256 / PL) * rows_rpg(nnodes, PL, true) <= 256)
        return launch_spmm_rows<true, true>(rowptr, col, val_a, val_l, X, YA, YL, nnodes, x_rows, (int)rowbytes, PL, st);
    const unsigned xb = (unsigned)(x_rows * rowbytes);
    const int W16 = (int)(rowbytes / 16), npanels = W16 / PL, nrb = cdiv(cdiv(nnodes, 8), 256 / PL);
    const long grid = 8L * npanels * nrb;
    REGT_CHECK_ARG(grid < (1L << 31), "spmm_dual (bf16 rows): grid too large");
    const uint4* X16 = reinterpret_cast<const uint4*>(X);
    uint4 *A16 = reinterpret_cast<uint4*>(YA), *L16 = reinterpret_cast<uint4*>(YL);
    if (wide) hipLaunchKernelGGL((spmm_dual_panel_bf16_kernel<16, 16>), dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val_a, val_l, X16, A16, L16, nnodes, W16, npanels, nrb, xb);
    else hipLaunchKernelGGL((spmm_dual_panel_bf16_kernel<8, 8>), dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val_a, val_l, X16, A16, L16, nnodes, W16, npanels, nrb, xb);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_spmm_dual(const int* rowptr, const int* col, const float* val_a, const float* val_l, const float* X, float* YA,
                     float* YL, int nnodes, int W, hipStream_t st) {
    return launch_spmm_dual_x(rowptr, col, val_a, val_l, X, YA, YL, nnodes, nnodes, W, st);
}

int launch_spmm_dual_x(const int* rowptr, const int* col, const float* val_a, const float* val_l, const float* X, float* YA,
                       float* YL, int nnodes, int x_rows, int W, hipStream_t st) {
    REGT_CHECK_ARG(nnodes > 0 && W > 0 && W % 4 == 0, "spmm_dual: width %d must be a multiple of 4 floats", W);
    const int W4 = W / 4;
    if (W % 32 != 0 || ((long)x_rows * W * 4 <= (24L << 20) && W4 <= 512)) {
        // rows that are not whole 128-byte panels, or an X that lives in the L2s anyway: one lane group per row, whole rows
        REGT_CHECK_ARG(W4 <= 512, "spmm_dual: width %d is neither a multiple of 32 floats nor at most 2048", W);
        long blocks;
#define REGT_DUALCSR(G, CH)                                                                                                     \
    do {                                                                                                                       \
        blocks = ((long)nnodes + 256 / G - 1) / (256 / G);                                                                     \
        if (blocks > 256L * 64) blocks = 256L * 64;                                                                            \
        hipLaunchKernelGGL((spmm_dual_csr_kernel<G, CH>), dim3((unsigned)blocks), dim3(256), 0, st, rowptr, col, val_a, val_l, X, YA, YL, \
                           nnodes, W4);                                                                                        \
    } while (0)
        if (W4 <= 8) REGT_DUALCSR(8, 1);
        else if (W4 <= 16) REGT_DUALCSR(16, 1);
        else if (W4 <= 32) REGT_DUALCSR(32, 1);
        else if (W4 <= 64) REGT_DUALCSR(64, 1);
        else if (W4 <= 128) REGT_DUALCSR(64, 2);
        else if (W4 <= 256) REGT_DUALCSR(64, 4);
        else REGT_DUALCSR(64, 8);
#undef REGT_DUALCSR
        REGT_CHECK_LAUNCH();
        return REGT_OK;
    }
    {
        const bool wide0 = panel_wide(nnodes) && W4 % 16 == 0;
        const int PL0 = wide0 ? 16 : 8;
        if (rows_ok(nnodes, x_rows, 4L * W, PL0))
            return launch_spmm_rows<true, false>(rowptr, col, val_a, val_l, X, YA, YL, nnodes, x_rows, 4 * W, PL0, st);
    }
    const bool wide = panel_wide(nnodes) && W4 % 16 == 0;     // (a forced 256-byte panel still needs whole panels)
    const int PL = wide ? 16 : 8;
    const int npanels = W4 / PL;
    const int nrb = cdiv(cdiv(nnodes, 8), 256 / PL);
    const long grid = 8L * npanels * nrb;
    REGT_CHECK_ARG(grid < (1L << 31), "spmm_dual: grid too large");
    // (streaming output stores, 16-entry index chunks: the variants measured against them -- ordinary stores, 8-entry chunks,
    // non-temporal CSR loads, a padded X stride: profiles/r03_spmm_pmc_variants.txt, r04_spmm_ab.txt -- are gone)
    if (wide) hipLaunchKernelGGL((spmm_dual_panel_kernel<16, 16>), dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val_a, val_l, X, YA, YL, nnodes, W4, npanels, nrb);
    else hipLaunchKernelGGL((spmm_dual_panel_kernel<8, 8>), dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val_a, val_l, X, YA, YL, nnodes, W4, npanels, nrb);
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
        if (nstack == 1) {      // one operator: the row-block kernel (CSR entries in LDS, the workgroup walks all panels)
            const bool wide1 = panel_wide(nnodes) && W4 % 16 == 0;
            if (rows_ok(nnodes, nrows_x, 4L * W, wide1 ? 16 : 8))
                return launch_spmm_rows<false, false>(rowptr, col, val, nullptr, X, Y, nullptr, nnodes, nrows_x, 4 * W, wide1 ? 16 : 8, st);
        }
        // 256-byte panels (half the passes over the CSR) while an XCD's slice of X (chunk x 256 B) stays around its 4 MiB L2
        const bool wide = panel_wide(nnodes) && W4 % 16 == 0;   // a forced 256-byte panel still needs whole panels
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
