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
                                                       float* __restrict__ Y, int nrows, int nrows_x, int W4) {
    constexpr int GROUPS = 256 / G;
    const int gl = threadIdx.x % G;                // lane inside the group
    const int gid = threadIdx.x / G;
    const long W = (long)W4 * 4;
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

template <int G, int CH>
static int launch_spmm_t(const int* rowptr, const int* col, const float* val, const float* X, float* Y, int nrows,
                         int nrows_x, int W4, hipStream_t st) {
    constexpr int GROUPS = 256 / G;
    long blocks = ((long)nrows + GROUPS - 1) / GROUPS;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipLaunchKernelGGL((spmm_csr_kernel<G, CH>), dim3((unsigned)blocks), dim3(256), 0, st, rowptr, col, val, X, Y, nrows,
                       nrows_x, W4);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_spmm_csr(const int* rowptr, const int* col, const float* val, const float* X, float* Y, int nrows,
                    int nrows_x, int W, hipStream_t st) {
    REGT_CHECK_ARG(nrows > 0 && W > 0, "spmm: empty problem");
    REGT_CHECK_ARG(W % 4 == 0, "spmm: row width %d must be a multiple of 4 floats", W);
    REGT_CHECK_ARG(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0, "spmm: X/Y must be 16-B aligned");
    const int W4 = W / 4;
#define REGT_SPMM(G, CH) return launch_spmm_t<G, CH>(rowptr, col, val, X, Y, nrows, nrows_x, W4, st)
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
    set_error("spmm: row width %d floats not supported (max 2048)", W);
    return REGT_ERR_ARG;
}

}  // namespace regt
