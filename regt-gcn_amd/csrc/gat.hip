// Attention aggregation of torch_geometric's GATConv (heads = 1, concat, add_self_loops, no edge features) -- the base block
// `baseblock='gat'` of the reference's TGCN cell (models/utils.py:97-98, used by models/GATTemporal.py:57-61).
//
//   x'_j = W x_j,   e_ij = leaky_relu(<a_src, x'_j> + <a_dst, x'_i>, 0.2),   alpha_ij = softmax over the in-edges j -> i,
//   out_i = sum_j alpha_ij x'_j + bias
//
// Aggregate-first, like the rest of the pipeline: <a_src, W x_j> = <W^T a_src, x_j> = <u_src, x_j> is a width-F dot product and
// sum_j alpha_ij (W x_j) = W (sum_j alpha_ij x_j), so the sparse work runs on the INPUT rows at width F (all T periods of a
// snapshot in one launch) and the dense C x F contraction is left to the MFMA GEMM of the cell.  What is new compared with
// GCN / Cheb: the edge weights depend on the parameters (through u_src, u_dst), so there IS a sparse backward -- the
// gradient of the scores -- done here as two pull passes (destination-major, then source-major over the transposed
// pattern) without float atomics: sums run in CSR order, bit-reproducible.
//
// Mapping: a group of G = F/4 lanes (padded to a power of two) owns one (node, period) row; every lane holds one float4 of
// the row, dot products are reduced with xor-shuffles inside the group.  Softmax is the one-pass (running max) form.
#include "kernels.h"

namespace regt {

namespace {

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// out[i,t,:] = sum_j alpha_ij x[j,t,:];  stats[(i,t)] = (running max m, sum l, d_i, 0)
template <int G>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const float* __restrict__ x, const float* __restrict__ us,
                                                      const float* __restrict__ ud, float slope, int N, int T, int F4,
                                                      float* __restrict__ out, float4* __restrict__ stats) {
    constexpr int GROUPS = 256 / G;
    const int gl = threadIdx.x % G, gid = threadIdx.x / G;
    const bool live = gl < F4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 u_s = live ? reinterpret_cast<const float4*>(us)[gl] : zero;
    const float4 u_d = live ? reinterpret_cast<const float4*>(ud)[gl] : zero;
    const long rows = (long)N * T;
    const long rs4 = (long)T * F4;                      // float4 stride between nodes
    for (long m = (long)blockIdx.x * GROUPS + gid; m < rows; m += (long)gridDim.x * GROUPS) {
        const int i = (int)(m / T), t = (int)(m - (long)i * T);
        const float4* xt = reinterpret_cast<const float4*>(x) + (long)t * F4 + gl;
        const float4 xi = live ? xt[(long)i * rs4] : zero;
        const float di = group_sum<G>(dot4(xi, u_d));
        float mx = -__builtin_inff(), l = 0.f;
        float4 acc = zero;
        const int beg = rowptr[i], end = rowptr[i + 1];
        for (int e = beg; e < end; ++e) {
            const int j = col[e];
            const float4 xj = live ? xt[(long)j * rs4] : zero;
            const float ev = lrelu(group_sum<G>(dot4(xj, u_s)) + di, slope);
            const float mn = fmaxf(mx, ev);
            const float sc = __expf(mx - mn), w = __expf(ev - mn);      // first edge: mx = -inf -> sc = 0
            l = l * sc + w;
            acc.x = fmaf(w, xj.x, acc.x * sc); acc.y = fmaf(w, xj.y, acc.y * sc);
            acc.z = fmaf(w, xj.z, acc.z * sc); acc.w = fmaf(w, xj.w, acc.w * sc);
            mx = mn;
        }
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (live) {
            float4* o = reinterpret_cast<float4*>(out) + m * F4 + gl;
            *o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        }
        if (gl == 0) stats[m] = make_float4(mx, l, di, 0.f);
    }
}

// destination-major backward: with g_ij = <dOut_i, x_j> and alpha_ij from the saved (m, l):
//   D_i = sum_j alpha_ij g_ij;   de_ij = alpha_ij (g_ij - D_i) * lrelu'(s_j + d_i);   dd_i = sum_j de_ij
// writes D_i into stats.w and dd into dsd[m][1]
template <int G>
__global__ __launch_bounds__(256) void gat_bwd_dst_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const float* __restrict__ x, const float* __restrict__ us,
                                                          float slope, int N, int T, int F4, const float* __restrict__ dout,
                                                          float4* __restrict__ stats, float* __restrict__ dsd) {
    constexpr int GROUPS = 256 / G;
    const int gl = threadIdx.x % G, gid = threadIdx.x / G;
    const bool live = gl < F4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 u_s = live ? reinterpret_cast<const float4*>(us)[gl] : zero;
    const long rows = (long)N * T;
    const long rs4 = (long)T * F4;
    for (long m = (long)blockIdx.x * GROUPS + gid; m < rows; m += (long)gridDim.x * GROUPS) {
        const int i = (int)(m / T), t = (int)(m - (long)i * T);
        const float4* xt = reinterpret_cast<const float4*>(x) + (long)t * F4 + gl;
        const float4 go = live ? reinterpret_cast<const float4*>(dout)[m * F4 + gl] : zero;
        const float4 st = stats[m];
        const float inv = st.y > 0.f ? 1.0f / st.y : 0.f;
        const int beg = rowptr[i], end = rowptr[i + 1];
        float D = 0.f;
        for (int e = beg; e < end; ++e) {
            const float4 xj = live ? xt[(long)col[e] * rs4] : zero;
            const float raw = group_sum<G>(dot4(xj, u_s)) + st.z;
            const float a = __expf(lrelu(raw, slope) - st.x) * inv;
            D = fmaf(a, group_sum<G>(dot4(go, xj)), D);
        }
        float dd = 0.f;
        for (int e = beg; e < end; ++e) {
            const float4 xj = live ? xt[(long)col[e] * rs4] : zero;
            const float raw = group_sum<G>(dot4(xj, u_s)) + st.z;
            const float a = __expf(lrelu(raw, slope) - st.x) * inv;
            dd += a * (group_sum<G>(dot4(go, xj)) - D) * (raw > 0.f ? 1.0f : slope);
        }
        if (gl == 0) {
            stats[m].w = D;
            dsd[2 * m + 1] = dd;
        }
    }
}

// source-major backward over the transposed pattern: ds_j = sum over out-edges j -> i of de_ij (recomputed from the saved
// per-destination statistics); writes dsd[m][0]
template <int G>
__global__ __launch_bounds__(256) void gat_bwd_src_kernel(const int* __restrict__ t_rowptr, const int* __restrict__ t_col,
                                                          const float* __restrict__ x, const float* __restrict__ us,
                                                          float slope, int N, int T, int F4, const float* __restrict__ dout,
                                                          const float4* __restrict__ stats, float* __restrict__ dsd) {
    constexpr int GROUPS = 256 / G;
    const int gl = threadIdx.x % G, gid = threadIdx.x / G;
    const bool live = gl < F4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 u_s = live ? reinterpret_cast<const float4*>(us)[gl] : zero;
    const long rows = (long)N * T;
    for (long m = (long)blockIdx.x * GROUPS + gid; m < rows; m += (long)gridDim.x * GROUPS) {
        const int j = (int)(m / T), t = (int)(m - (long)j * T);
        const float4 xj = live ? reinterpret_cast<const float4*>(x)[m * F4 + gl] : zero;
        const float sj = group_sum<G>(dot4(xj, u_s));
        const int beg = t_rowptr[j], end = t_rowptr[j + 1];
        float ds = 0.f;
        for (int e = beg; e < end; ++e) {
            const long mi = (long)t_col[e] * T + t;
            const float4 st = stats[mi];
            const float4 go = live ? reinterpret_cast<const float4*>(dout)[mi * F4 + gl] : zero;
            const float raw = sj + st.z;
            const float a = __expf(lrelu(raw, slope) - st.x) * (st.y > 0.f ? 1.0f / st.y : 0.f);
            ds += a * (group_sum<G>(dot4(go, xj)) - st.w) * (raw > 0.f ? 1.0f : slope);
        }
        if (gl == 0) dsd[2 * m] = ds;
    }
}

int pick_group(int F4) { return F4 <= 2 ? 2 : (F4 <= 4 ? 4 : (F4 <= 8 ? 8 : (F4 <= 16 ? 16 : (F4 <= 32 ? 32 : 64)))); }

}  // namespace

#define REGT_GAT_DISPATCH(KERNEL, ...)                                                                                  \
    switch (pick_group(F4)) {                                                                                           \
        case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(blocks(2)), dim3(256), 0, st, __VA_ARGS__); break;                  \
        case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(blocks(4)), dim3(256), 0, st, __VA_ARGS__); break;                  \
        case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(blocks(8)), dim3(256), 0, st, __VA_ARGS__); break;                  \
        case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(blocks(16)), dim3(256), 0, st, __VA_ARGS__); break;               \
        case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(blocks(32)), dim3(256), 0, st, __VA_ARGS__); break;               \
        default: hipLaunchKernelGGL((KERNEL<64>), dim3(blocks(64)), dim3(256), 0, st, __VA_ARGS__); break;               \
    }

int launch_gat_forward(const int* rowptr, const int* col, const float* x, const float* us, const float* ud, float slope, int N,
                       int T, int F, float* out, float* stats, hipStream_t st) {
    REGT_CHECK_ARG(N > 0 && T > 0 && F > 0 && F % 4 == 0 && F <= 256, "gat: F=%d must be a multiple of 4, at most 256", F);
    const int F4 = F / 4;
    const long rows = (long)N * T;
    auto blocks = [&](int g) { long b = (rows + 256 / g - 1) / (256 / g); return (unsigned)(b > 65536 ? 65536 : b); };
    REGT_GAT_DISPATCH(gat_fwd_kernel, rowptr, col, x, us, ud, slope, N, T, F4, out, reinterpret_cast<float4*>(stats));
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_gat_backward(const int* rowptr, const int* col, const int* t_rowptr, const int* t_col, const float* x, const float* us,
                        float slope, int N, int T, int F, const float* dout, float* stats, float* dsd, hipStream_t st) {
    REGT_CHECK_ARG(N > 0 && T > 0 && F > 0 && F % 4 == 0 && F <= 256, "gat: F=%d must be a multiple of 4, at most 256", F);
    const int F4 = F / 4;
    const long rows = (long)N * T;
    auto blocks = [&](int g) { long b = (rows + 256 / g - 1) / (256 / g); return (unsigned)(b > 65536 ? 65536 : b); };
    REGT_GAT_DISPATCH(gat_bwd_dst_kernel, rowptr, col, x, us, slope, N, T, F4, dout, reinterpret_cast<float4*>(stats), dsd);
    REGT_CHECK_LAUNCH();
    REGT_GAT_DISPATCH(gat_bwd_src_kernel, t_rowptr, t_col, x, us, slope, N, T, F4, dout, reinterpret_cast<const float4*>(stats), dsd);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
