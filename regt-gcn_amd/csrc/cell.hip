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
constexpr int CB_MAXT = 255;     // periods per snapshot (the row tables of the candidate kernels keep the period in 8 bits)
// four fp32 -> four bf16 (round to nearest even), one 8-byte store at bf16 element index `elem`
__device__ __forceinline__ void store_bf16x4(void* base, long elem, float4 v) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = {v.x, v.y}, hi = {v.z, v.w};
    const bf16x2_t bl = __builtin_convertvector(lo, bf16x2_t), bh = __builtin_convertvector(hi, bf16x2_t);
    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base) + 2 * elem) =
        make_uint2(__builtin_bit_cast(unsigned, bl), __builtin_bit_cast(unsigned, bh));
}
__global__ __launch_bounds__(256) void cell_bwd_kernel(CellBwdArgs a) {
    __shared__ float dp[4][CB_MAXT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int t = lane; t < a.T; t += 64) dp[wid][t] = 0.f;
    const long C = a.C;
    const int C4 = a.C / 4;
    const long ldz = a.ldz ? a.ldz : 2 * C, lddz = a.lddz ? a.lddz : 2 * C;
    const int n0 = blockIdx.x * a.nodes_per_block;
    const int n1 = n0 + a.nodes_per_block < a.num_nodes ? n0 + a.nodes_per_block : a.num_nodes;
    for (int node = n0 + wid; node < n1; node += 4) {
        for (int t = 0; t < a.T; ++t) {
            const long m = (long)node * a.T + t;
            const float pt = a.probs[t];
            float dot = 0.f;
            for (int c4 = lane; c4 < C4; c4 += 64) {
                const float4 d = reinterpret_cast<const float4*>(a.dOH + node * C)[c4];
                const float4 z = reinterpret_cast<const float4*>(a.ZR + m * ldz)[c4];
                const float4 h = a.h ? reinterpret_cast<const float4*>(a.h + m * C)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
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
                if (a.out_bf16) {      // GEMM-only intermediates, rounded once here (REGT_GEMM_MODE=bf16)
                    store_bf16x4(a.dhp, m * C + 4 * c4, dhp);
                    store_bf16x4(a.dzr, m * lddz + 4 * c4, dzp);
                } else {
                    reinterpret_cast<float4*>(a.dhp + m * C)[c4] = dhp;
                    reinterpret_cast<float4*>(a.dzr + m * lddz)[c4] = dzp;
                }
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

// The same with Z, h, H~ stored as bf16 and dhp / dzp written as bf16 (REGT_GEMM_MODE=bf16): a lane owns 8 columns (16 bytes
// of every array), a half wave one (node, t) row, so a wave handles two periods per pass.
__device__ __forceinline__ void widen8(const uint4 raw, float (&v)[8]) {
    v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
    v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
    v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 narrow8(const float (&v)[8]) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    uint4 raw;
    const f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]}, c = {v[4], v[5]}, d = {v[6], v[7]};
    raw.x = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t));
    raw.y = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_t));
    raw.z = __builtin_bit_cast(unsigned, __builtin_convertvector(c, bf16x2_t));
    raw.w = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2_t));
    return raw;
}
__global__ __launch_bounds__(256) void cell_bwd8_kernel(CellBwdArgs a) {
    __shared__ float dp[4][CB_MAXT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int hl = lane & 31, half = lane >> 5;
    for (int t = lane; t < a.T; t += 64) dp[wid][t] = 0.f;
    const long C = a.C;
    const int C8 = a.C / 8;
    const long ldz = a.ldz ? a.ldz : 2 * C, lddz = a.lddz ? a.lddz : 2 * C;
    const char* Zb = reinterpret_cast<const char*>(a.ZR);
    const char* Hb = reinterpret_cast<const char*>(a.h);
    const char* Tb = reinterpret_cast<const char*>(a.Ht);
    char* Pb = reinterpret_cast<char*>(a.dhp);
    char* Db = reinterpret_cast<char*>(a.dzr);
    const int n0 = blockIdx.x * a.nodes_per_block;
    const int n1 = n0 + a.nodes_per_block < a.num_nodes ? n0 + a.nodes_per_block : a.num_nodes;
    for (int node = n0 + wid; node < n1; node += 4) {
        for (int t0 = 0; t0 < a.T; t0 += 2) {
            const int t = t0 + half;
            const bool live = t < a.T;
            const long m = (long)node * a.T + (live ? t : 0);
            const float pt = live ? a.probs[t] : 0.f;
            float dot = 0.f;
            if (live) {
                for (int c8 = hl; c8 < C8; c8 += 32) {
                    const float4 d0 = reinterpret_cast<const float4*>(a.dOH + node * C)[2 * c8];
                    const float4 d1 = reinterpret_cast<const float4*>(a.dOH + node * C)[2 * c8 + 1];
                    const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                    float z[8], h[8], ht[8], dhp[8], dzp[8];
                    widen8(*reinterpret_cast<const uint4*>(Zb + 2 * (m * ldz + 8 * c8)), z);
                    widen8(*reinterpret_cast<const uint4*>(Tb + 2 * (m * C + 8 * c8)), ht);
                    if (a.h) widen8(*reinterpret_cast<const uint4*>(Hb + 2 * (m * C + 8 * c8)), h);
                    else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) h[k] = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float g = __fmul_rn(pt, dd[k]);
                        dhp[k] = cb_dhp(g, z[k], ht[k]);
                        dzp[k] = cb_dzp(g, h[k], ht[k], z[k]);
                        dot += dd[k] * (z[k] * h[k] + (1.0f - z[k]) * ht[k]);
                    }
                    *reinterpret_cast<uint4*>(Pb + 2 * (m * C + 8 * c8)) = narrow8(dhp);
                    *reinterpret_cast<uint4*>(Db + 2 * (m * lddz + 8 * c8)) = narrow8(dzp);
                }
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);      // inside each half wave
            if (hl == 0 && live) dp[wid][t] += dot;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < a.T; t += 256)
        a.dp_partial[(long)blockIdx.x * a.T + t] = (dp[0][t] + dp[1][t]) + (dp[2][t] + dp[3][t]);
}

int cell_bwd_blocks(int num_nodes, int nodes_per_block) { return cdiv(num_nodes, nodes_per_block); }

// Attention-probability partial sums from the per-row dots the fused backward kernel leaves behind: block b owns
// nodes_per_block consecutive nodes; thread (g, t) adds rowdot[node * T + t] for the nodes n0 + g, n0 + g + G, ... (ascending),
// thread t then adds the G group sums in order -- a fixed summation order, like cell_bwd_kernel's.
__global__ __launch_bounds__(256) void rowdot_reduce_kernel(const float* __restrict__ rowdot, float* __restrict__ dp_partial,
                                                            int num_nodes, int T, int nodes_per_block, int parts) {
    __shared__ float part[256];
    const int G = 256 / T, g = threadIdx.x / T, t = threadIdx.x - g * T;
    const int n0 = blockIdx.x * nodes_per_block;
    const int n1 = n0 + nodes_per_block < num_nodes ? n0 + nodes_per_block : num_nodes;
    float s = 0.f;
    if (g < G)
        for (int node = n0 + g; node < n1; node += G) {
            const float* q = rowdot + ((long)node * T + t) * parts;
            float r = q[0];
            for (int k = 1; k < parts; ++k) r += q[k];
            s += r;
        }
    part[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < T) {
        float acc = 0.f;
        for (int k = 0; k < G; ++k) acc += part[k * T + threadIdx.x];
        dp_partial[(long)blockIdx.x * T + threadIdx.x] = acc;
    }
}
int launch_rowdot_reduce(const float* rowdot, float* dp_partial, int num_nodes, int T, int nodes_per_block, hipStream_t st, int parts) {
    REGT_CHECK_ARG(T >= 1 && T <= CB_MAXT && parts >= 1, "rowdot_reduce: T=%d outside 1..%d", T, CB_MAXT);
    hipLaunchKernelGGL(rowdot_reduce_kernel, dim3(cell_bwd_blocks(num_nodes, nodes_per_block)), dim3(256), 0, st, rowdot, dp_partial,
                       num_nodes, T, nodes_per_block, parts);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// Zero-hidden cell (the reference's GraphSAGE / GAT models call the cell with H = None -> zeros, models/utils.py:163-166):
// H' = Z * 0 + (1 - Z) * H~, summed over the T periods with the attention probabilities.  One wave per node, lanes over C.
__global__ __launch_bounds__(256) void blend0_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ Ht,
                                                         const float* __restrict__ probs, float* __restrict__ hidden,
                                                         int num_nodes, int T, int C4) {
    const int lane = threadIdx.x & 63;
    const long waves = (long)gridDim.x * 4;
    for (long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6); node < num_nodes; node += waves) {
        for (int c4 = lane; c4 < C4; c4 += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < T; ++t) {
                const long m = node * T + t;
                const float p = probs[t];
                const float4 z = reinterpret_cast<const float4*>(Z)[m * C4 + c4];
                const float4 h = reinterpret_cast<const float4*>(Ht)[m * C4 + c4];
                acc.x += p * ((1.0f - z.x) * h.x); acc.y += p * ((1.0f - z.y) * h.y);
                acc.z += p * ((1.0f - z.z) * h.z); acc.w += p * ((1.0f - z.w) * h.w);
            }
            reinterpret_cast<float4*>(hidden)[node * C4 + c4] = acc;
        }
    }
}

int launch_blend0_fwd(const float* Z, const float* Ht, const float* probs, float* hidden, int num_nodes, int T, int C, hipStream_t st) {
    REGT_CHECK_ARG(C % 4 == 0 && num_nodes > 0 && T > 0, "blend0: bad shape");
    int blocks = cdiv(num_nodes, 4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(blend0_fwd_kernel, dim3(blocks), dim3(256), 0, st, Z, Ht, probs, hidden, num_nodes, T, C / 4);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_cell_bwd(const CellBwdArgs& a, hipStream_t st) {
    REGT_CHECK_ARG(a.T <= CB_MAXT, "cell_bwd: T=%d exceeds %d", a.T, CB_MAXT);
    REGT_CHECK_ARG(a.C % 4 == 0, "cell_bwd: C must be a multiple of 4");
    if (a.in_bf16) {
        REGT_CHECK_ARG(a.out_bf16 && a.C % 8 == 0, "cell_bwd: bf16-stored activations need bf16 outputs and C %% 8 == 0");
        hipLaunchKernelGGL(cell_bwd8_kernel, dim3(cell_bwd_blocks(a.num_nodes, a.nodes_per_block)), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(cell_bwd_kernel, dim3(cell_bwd_blocks(a.num_nodes, a.nodes_per_block)), dim3(256), 0, st, a);
    }
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

__global__ __launch_bounds__(256) void cvt_bf16_frag_kernel(CvtBatch b) {
    int t = 0;
    while (t + 1 < b.n && (int)blockIdx.x >= b.block_start[t + 1]) ++t;
    const CvtTask k = b.t[t];
    const int kg_per_row = k.cols / 8, rows_pad = (k.rows + 127) / 128 * 128;
    const long idx = (long)(blockIdx.x - b.block_start[t]) * 256 + threadIdx.x;      // one 16-byte chunk: (row n, k-group kg)
    if (idx >= (long)rows_pad * kg_per_row) return;
    // chunk order of the destination: block (n / 32, kg / 2), inside it lane 32 (kg % 2) + n % 32
    const long blk = idx / 64;
    const int lane = (int)(idx % 64);
    const int nb = (int)(blk / (k.cols / 16)), ks = (int)(blk % (k.cols / 16));
    const int n = 32 * nb + (lane & 31), c = 16 * ks + 8 * (lane >> 5);
    uint4 out = make_uint4(0, 0, 0, 0);
    if (n < k.rows) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const float4 lo = *reinterpret_cast<const float4*>(k.src + (long)n * k.ld + c), hi = *reinterpret_cast<const float4*>(k.src + (long)n * k.ld + c + 4);
        const f32x2_t p0 = {lo.x, lo.y}, p1 = {lo.z, lo.w}, p2 = {hi.x, hi.y}, p3 = {hi.z, hi.w};
        out = make_uint4(__builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2_t)),
                         __builtin_bit_cast(unsigned, __builtin_convertvector(p2, bf16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(p3, bf16x2_t)));
    }
    reinterpret_cast<uint4*>(k.dst)[idx] = out;
}
int launch_cvt_bf16_frag(CvtBatch& b, hipStream_t st) {
    REGT_CHECK_ARG(b.n >= 1 && b.n <= 8, "cvt_bf16_frag: %d tasks", b.n);
    int blocks = 0;
    for (int t = 0; t < b.n; ++t) {
        REGT_CHECK_ARG(b.t[t].cols % 16 == 0 && b.t[t].ld % 4 == 0 && b.t[t].rows > 0, "cvt_bf16_frag: block %d: cols %% 16", t);
        b.block_start[t] = blocks;
        blocks += (int)(((long)((b.t[t].rows + 127) / 128 * 128) * (b.t[t].cols / 8) + 255) / 256);
    }
    b.block_start[b.n] = blocks;
    hipLaunchKernelGGL(cvt_bf16_frag_kernel, dim3(blocks), dim3(256), 0, st, b);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_transpose3(const float* s0, const float* s1, const float* s2, int count, float* dst, int rows, int cols, long ld,
                      hipStream_t st) {
    REGT_CHECK_ARG(count >= 1 && count <= 3 && rows > 0 && cols > 0, "transpose3: bad argument");
    hipLaunchKernelGGL(transpose3_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), count), dim3(256), 0, st, s0, s1, s2, dst, rows, cols, ld);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// ---- the last layer of the head for a short horizon (output_dim <= 4) -------------------------------------------------
// linear2 is (H1 -> O) with O = 1..4 in every configuration of the reference (run.py --num_timesteps_out): as a tiled
// MFMA GEMM it is a 128-wide tile with one live column (and its gradients a K = O contraction).  These three kernels
// do the same arithmetic as plain row-wise fp32 work at the HBM rate: one pass over y1 (N x H1) each.
constexpr int SK_MAXO = 4;

// pred[n, o] = sum_k y1[n, k] W2[o, k] + b2[o]; 32 lanes per row (float4 each), two rows per wave
__global__ __launch_bounds__(256) void head2_fwd_kernel(const float* __restrict__ y1, const float* __restrict__ W2,
                                                        const float* __restrict__ b2, float* __restrict__ pred, int N, int H1, int O) {
    const int sub = threadIdx.x & 31;
    const long groups = (long)gridDim.x * 8;                      // eight 32-lane groups per workgroup, one row each
    for (long n = (long)blockIdx.x * 8 + (threadIdx.x >> 5); n < N; n += groups) {
        float s[SK_MAXO] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 4 * sub; k < H1; k += 128) {
            const float4 v = *reinterpret_cast<const float4*>(y1 + n * H1 + k);
#pragma unroll
            for (int o = 0; o < SK_MAXO; ++o)
                if (o < O) {
                    const float4 w = *reinterpret_cast<const float4*>(W2 + (long)o * H1 + k);
                    s[o] = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, fmaf(v.w, w.w, s[o]))));
                }
        }
#pragma unroll
        for (int o = 0; o < SK_MAXO; ++o)
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) s[o] += __shfl_xor(s[o], off, 32);     // inside the row's 32 lanes
        if (sub < O) pred[n * O + sub] = (sub == 0 ? s[0] : sub == 1 ? s[1] : sub == 2 ? s[2] : s[3]) + b2[sub];
    }
}

// d1[n, k] = (y1[n, k] > 0) * sum_o dpred[n, o] W2[o, k]
__global__ __launch_bounds__(256) void head2_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ W2,
                                                        const float* __restrict__ y1, float* __restrict__ d1, long N, int H1, int O) {
    const int h4 = H1 / 4;
    const long total = N * h4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / h4;
        const int k = 4 * (int)(i - n * h4);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < SK_MAXO; ++o)
            if (o < O) {
                const float g = dpred[n * O + o];
                const float4 w = *reinterpret_cast<const float4*>(W2 + (long)o * H1 + k);
                a.x = fmaf(g, w.x, a.x); a.y = fmaf(g, w.y, a.y); a.z = fmaf(g, w.z, a.z); a.w = fmaf(g, w.w, a.w);
            }
        const float4 m = *reinterpret_cast<const float4*>(y1 + n * H1 + k);
        a.x = m.x > 0.f ? a.x : 0.f; a.y = m.y > 0.f ? a.y : 0.f; a.z = m.z > 0.f ? a.z : 0.f; a.w = m.w > 0.f ? a.w : 0.f;
        *reinterpret_cast<float4*>(d1 + n * H1 + k) = a;
    }
}

// slab[chunk] = (dW2 partial (O x H1), db2 partial (O)) over the chunk's rows: dW2[o, k] = sum_n dpred[n, o] y1[n, k].
// Thread = (float4 of k, one of 8 row subsets); the subsets are combined in fixed order through LDS, the chunks by the
// ordinary wgrad_reduce_kernel -- same slab layout as wgrad_kernel, deterministic.
__global__ __launch_bounds__(256) void head2_wgrad_kernel(const float* __restrict__ dpred, const float* __restrict__ y1,
                                                          float* __restrict__ slab, long N, int H1, int O, int kchunk, int colsum) {
    __shared__ float4 red[8][32][SK_MAXO];
    __shared__ float redb[8][32][SK_MAXO];
    const int c4 = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const long r0 = (long)blockIdx.x * kchunk, r1 = r0 + kchunk < N ? r0 + kchunk : N;
    const long stride = (long)O * H1 + (colsum ? O : 0);
    float* out = slab + (long)blockIdx.x * stride;
    for (int kb = 0; kb < H1; kb += 128) {
        const int k = kb + 4 * c4;
        float4 acc[SK_MAXO];
        float accb[SK_MAXO];
#pragma unroll
        for (int o = 0; o < SK_MAXO; ++o) { acc[o] = make_float4(0.f, 0.f, 0.f, 0.f); accb[o] = 0.f; }
        if (k < H1)
            for (long n = r0 + rs; n < r1; n += 8) {
                const float4 v = *reinterpret_cast<const float4*>(y1 + n * H1 + k);
#pragma unroll
                for (int o = 0; o < SK_MAXO; ++o)
                    if (o < O) {
                        const float g = dpred[n * O + o];
                        acc[o].x = fmaf(g, v.x, acc[o].x); acc[o].y = fmaf(g, v.y, acc[o].y);
                        acc[o].z = fmaf(g, v.z, acc[o].z); acc[o].w = fmaf(g, v.w, acc[o].w);
                        accb[o] += g;
                    }
            }
#pragma unroll
        for (int o = 0; o < SK_MAXO; ++o) { red[rs][c4][o] = acc[o]; redb[rs][c4][o] = accb[o]; }
        __syncthreads();
        if (rs == 0 && k < H1) {
#pragma unroll
            for (int o = 0; o < SK_MAXO; ++o)
                if (o < O) {
                    float4 s = red[0][c4][o];
                    for (int g = 1; g < 8; ++g) { const float4 t = red[g][c4][o]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
                    *reinterpret_cast<float4*>(out + (long)o * H1 + k) = s;
                }
        }
        if (colsum && kb == 0 && threadIdx.x < O) {
            float s = 0.f;
            for (int g = 0; g < 8; ++g) s += redb[g][0][threadIdx.x];
            out[(long)O * H1 + threadIdx.x] = s;
        }
        __syncthreads();
    }
}

bool head2_skinny_ok(int H1, int O, const void* y1, const void* W2) {
    return O >= 1 && O <= SK_MAXO && H1 % 4 == 0 && ((reinterpret_cast<uintptr_t>(y1) | reinterpret_cast<uintptr_t>(W2)) & 15) == 0;
}
int launch_head2_fwd(const float* y1, const float* W2, const float* b2, float* pred, int N, int H1, int O, hipStream_t st) {
    int blocks = cdiv(N, 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(head2_fwd_kernel, dim3(blocks), dim3(256), 0, st, y1, W2, b2, pred, N, H1, O);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
int launch_head2_bwd(const float* dpred, const float* W2, const float* y1, float* d1, int N, int H1, int O, hipStream_t st) {
    int blocks = cdiv((long)N * (H1 / 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(head2_bwd_kernel, dim3(blocks), dim3(256), 0, st, dpred, W2, y1, d1, (long)N, H1, O);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
int launch_head2_wgrad(const float* dpred, const float* y1, float* slab, int N, int H1, int O, int kchunk, int nchunks, int colsum,
                       hipStream_t st) {
    hipLaunchKernelGGL(head2_wgrad_kernel, dim3(nchunks), dim3(256), 0, st, dpred, y1, slab, (long)N, H1, O, kchunk, colsum);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// S = sum over the R column blocks of the region linear layer (C x R*C): every composition that touches "all regions"
// (A0, b', dW0, db_c) goes through S, so their cost does not grow with the region count of a multi-GPU graph.
__global__ __launch_bounds__(256) void sum_region_blocks_kernel(const float* __restrict__ W, float* __restrict__ S, int C, int R) {
    const long total = (long)C * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e / C, j = e - i * C;
        const float* w = W + i * (long)R * C + j;
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += w[(long)r * C];
        S[e] = s;
    }
}
int launch_sum_region_blocks(const float* W, float* S, int C, int R, hipStream_t st) {
    int blocks = cdiv((long)C * C, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sum_region_blocks_kernel, dim3(blocks), dim3(256), 0, st, W, S, C, R);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// loss = scale * sum (pred - y)^2 ;  dpred = 2 * scale * (pred - y)     (scale = 1 / (N_global * O))
// Up to 256 workgroups, each a contiguous chunk; the last one to finish (ticket counter) adds the per-workgroup sums in index
// order -- the same result whatever the execution order.  Scratch: one of 64 static slots per call (calls in flight on different
// streams do not share one).
constexpr int MSE_MAX_BLOCKS = 256, MSE_SLOTS = 64;
__device__ float g_mse_part[MSE_SLOTS][MSE_MAX_BLOCKS];
__device__ unsigned g_mse_ticket[MSE_SLOTS];
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* pred, const float* y, float* dpred, float* loss_out,
                                                       long n, float scale, int slot) {
    __shared__ float red[256];
    __shared__ bool last;
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long i0 = (long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    float s = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
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
    if (!loss_out) return;
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) *loss_out = red[0] * scale;
        return;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&g_mse_part[slot][blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(&g_mse_ticket[slot], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        float tot = 0.f;
        for (unsigned b = 0; b < gridDim.x; ++b) tot += __hip_atomic_load(&g_mse_part[slot][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *loss_out = tot * scale;
        g_mse_ticket[slot] = 0;
    }
}

int launch_mse_grad(const float* pred, const float* y, float* dpred, float* loss_out, long n, float scale, hipStream_t st) {
    static unsigned next_slot = 0;
    long blocks = (n + 2047) / 2048;             // >= 8 elements per thread before a second workgroup pays
    if (blocks > MSE_MAX_BLOCKS) blocks = MSE_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const int slot = (int)(next_slot++ % MSE_SLOTS);
    hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pred, y, dpred, loss_out, n, scale, slot);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
