// Element-wise / reduction kernels around the GRU cell (models/utils.py:163-203) and the attention
// over periods (models/RegionalTemporalGCN.py:134,146).  The gate non-linearities of the *forward*
// live in the GEMM epilogues (gemm.hip); this file holds the head of the backward pass -- one fused
// kernel that turns dL/dH_accum into the three gate pre-activation gradients -- plus the softmax of
// the T attention logits, its backward, and the MSE loss gradient of run.py:180.
#include "kernels.h"

namespace regt {

__global__ void softmax_small_kernel(const float* att, float* probs, int T) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float mx = att[0];
        for (int t = 1; t < T; ++t) mx = fmaxf(mx, att[t]);
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += expf(att[t] - mx);
        for (int t = 0; t < T; ++t) probs[t] = expf(att[t] - mx) / s;
    }
}

int launch_softmax_small(const float* att, float* probs, int T, hipStream_t st) {
    hipLaunchKernelGGL(softmax_small_kernel, dim3(1), dim3(64), 0, st, att, probs, T);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// One wave per (node, t) row of width C.  With g = p_t * dOH[node]:
//   dhp = g (1-Z) (1-Ht^2)            (candidate pre-activation gradient)
//   dzp = g (h - Ht) Z (1-Z)          (update-gate pre-activation gradient) -> dzr[:, 0:C]
//   dp_t += <dOH[node], Z h + (1-Z) Ht>   (attention-probability gradient; fixed-order partial sums)
constexpr int CB_MAXT = 64;
__global__ __launch_bounds__(256) void cell_bwd_kernel(CellBwdArgs a) {
    __shared__ float dp[4][CB_MAXT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int t = lane; t < a.T; t += 64) dp[wid][t] = 0.f;
    const long C = a.C;
    const int C4 = a.C / 4;
    const int n0 = blockIdx.x * a.nodes_per_block;
    const int n1 = n0 + a.nodes_per_block < a.num_nodes ? n0 + a.nodes_per_block : a.num_nodes;
    for (int node = n0 + wid; node < n1; node += 4) {
        for (int t = 0; t < a.T; ++t) {
            const long m = (long)node * a.T + t;
            const float pt = a.probs[t];
            float dot = 0.f;
            for (int c4 = lane; c4 < C4; c4 += 64) {
                const float4 d = reinterpret_cast<const float4*>(a.dOH + node * C)[c4];
                const float4 z = reinterpret_cast<const float4*>(a.ZR + m * 2 * C)[c4];
                const float4 h = reinterpret_cast<const float4*>(a.h + m * C)[c4];
                const float4 ht = reinterpret_cast<const float4*>(a.Ht + m * C)[c4];
                float4 dhp, dzp;
#define REGT_CB(x)                                                   \
    {                                                                \
        float g = pt * d.x;                                          \
        dhp.x = g * (1.0f - z.x) * (1.0f - ht.x * ht.x);             \
        dzp.x = g * (h.x - ht.x) * (z.x * (1.0f - z.x));             \
        dot += d.x * (z.x * h.x + (1.0f - z.x) * ht.x);              \
    }
                REGT_CB(x) REGT_CB(y) REGT_CB(z) REGT_CB(w)
#undef REGT_CB
                reinterpret_cast<float4*>(a.dhp + m * C)[c4] = dhp;
                reinterpret_cast<float4*>(a.dzr + m * 2 * C)[c4] = dzp;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
            if (lane == 0) dp[wid][t] += dot;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < a.T; t += 256)
        a.dp_partial[(long)blockIdx.x * a.T + t] = (dp[0][t] + dp[1][t]) + (dp[2][t] + dp[3][t]);
}

int cell_bwd_blocks(int num_nodes, int nodes_per_block) { return cdiv(num_nodes, nodes_per_block); }

int launch_cell_bwd(const CellBwdArgs& a, hipStream_t st) {
    REGT_CHECK_ARG(a.T <= CB_MAXT, "cell_bwd: T=%d exceeds %d", a.T, CB_MAXT);
    REGT_CHECK_ARG(a.C % 4 == 0, "cell_bwd: C must be a multiple of 4");
    hipLaunchKernelGGL(cell_bwd_kernel, dim3(cell_bwd_blocks(a.num_nodes, a.nodes_per_block)), dim3(256), 0, st, a);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// datt_t = p_t (dp_t - sum_s p_s dp_s), dp_t = fixed-order sum of the per-block partials.
__global__ __launch_bounds__(256) void att_bwd_kernel(const float* dp_partial, int nblocks, const float* probs, float* datt, int T) {
    __shared__ float red[256];
    __shared__ float dp[CB_MAXT];
    for (int t = 0; t < T; ++t) {
        float s = 0.f;
        for (int b = threadIdx.x; b < nblocks; b += 256) s += dp_partial[(long)b * T + t];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) dp[t] = red[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float mean = 0.f;
        for (int t = 0; t < T; ++t) mean += probs[t] * dp[t];
        for (int t = 0; t < T; ++t) datt[t] = probs[t] * (dp[t] - mean);
    }
}

int launch_att_bwd(const float* dp_partial, int nblocks, const float* probs, float* datt, int T, hipStream_t st) {
    REGT_CHECK_ARG(T <= CB_MAXT, "att_bwd: T=%d exceeds %d", T, CB_MAXT);
    hipLaunchKernelGGL(att_bwd_kernel, dim3(1), dim3(256), 0, st, dp_partial, nblocks, probs, datt, T);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// Tiny device-to-device copies / clears as kernels: inside a captured hipGraph a memcpy / memset node costs far
// more than a kernel node, and these sit on the launch-bound path of small graphs.
__global__ void copy_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void zero_f32_kernel(float4* __restrict__ dst, long n4, float* __restrict__ tail, int ntail) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}

int launch_copy_f32(float* dst, const float* src, long n, hipStream_t st) {
    int blocks = cdiv(n, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_f32_kernel, dim3(blocks), dim3(256), 0, st, dst, src, n);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_zero_f32(float* dst, long n, hipStream_t st) {
    REGT_CHECK_ARG((reinterpret_cast<uintptr_t>(dst) & 15) == 0, "zero_f32: destination must be 16-byte aligned");
    const long n4 = n / 4;
    int blocks = cdiv(n4 > 0 ? n4 : 1, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_f32_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float4*>(dst), n4, dst + 4 * n4, (int)(n - 4 * n4));
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// dst[b] (cols x rows, ld = rows) = src[b]^T for up to three row-major (rows x cols, ld = lds) matrices: the
// [K][N] -> [N][K] copies of the gate weights that the bf16x3 split GEMM core wants for the data gradients.
__global__ __launch_bounds__(256) void transpose3_kernel(const float* s0, const float* s1, const float* s2, float* dst, int rows,
                                                         int cols, long lds_) {
    __shared__ float tile[32][33];
    const float* src = blockIdx.z == 0 ? s0 : (blockIdx.z == 1 ? s1 : s2);
    float* out = dst + (long)blockIdx.z * rows * cols;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(long)(r0 + i) * lds_ + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

int launch_transpose3(const float* s0, const float* s1, const float* s2, int count, float* dst, int rows, int cols, long ld,
                      hipStream_t st) {
    REGT_CHECK_ARG(count >= 1 && count <= 3 && rows > 0 && cols > 0, "transpose3: bad argument");
    hipLaunchKernelGGL(transpose3_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), count), dim3(256), 0, st, s0, s1, s2, dst, rows, cols, ld);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// loss = scale * sum (pred - y)^2 ;  dpred = 2 * scale * (pred - y)     (scale = 1 / (N_global * O))
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* pred, const float* y, float* dpred, float* loss_out,
                                                       long n, float scale) {
    __shared__ float red[256];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        float d = pred[i] - y[i];
        if (dpred) dpred[i] = 2.0f * scale * d;
        s += d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss_out) *loss_out = red[0] * scale;
}

int launch_mse_grad(const float* pred, const float* y, float* dpred, float* loss_out, long n, float scale, hipStream_t st) {
    hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(256), 0, st, pred, y, dpred, loss_out, n, scale);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
