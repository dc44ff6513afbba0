// Does a warp-specialised GEMM loop (4 MFMA waves, one per SIMD, + 4 loader waves that stream HBM -> LDS) keep the
// fp32 matrix pipe nearer its peak than two MFMA-issuing waves per SIMD?  Synthetic: no useful result is produced.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROW = 36, TILE = 128 * ROW, STAGE = 2 * TILE;

template <int MODE>   // 0: 8 waves, 4 consumers + 4 producers ; 1: 4 waves, everybody loads and computes (as gemm_fast.h)
__global__ __launch_bounds__(MODE == 0 ? 512 : 256) void k(const float* A, float* out, int iters, long rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const long base = ((long)blockIdx.x * 257) % (rows - 256);
    if (MODE == 0) {
        const bool producer = wid >= 4;
        const int ptid = tid - 256;
        float4 r[8];
        if (producer) for (int i = 0; i < 8; ++i) { int s = ptid + 256 * i; r[i] = *(const float4*)(A + (base + (s >> 3)) * 256 + 4 * (s & 7)); }
        for (int it = 0; it < iters; ++it) {
            float* st = lds + (it & 1) * STAGE;
            if (producer) {
                float* nx = lds + ((it + 1) & 1) * STAGE;
                for (int i = 0; i < 8; ++i) { int s = ptid + 256 * i; *(float4*)(nx + (s >> 3) * ROW + 4 * (s & 7)) = r[i]; }
                const int k0 = ((it + 2) * 32) & 255;
                for (int i = 0; i < 8; ++i) { int s = ptid + 256 * i; r[i] = *(const float4*)(A + (base + (s >> 3) + it) * 256 + k0 + 4 * (s & 7) % 32); }
            } else {
                const int wr = wid >> 1, wc = wid & 1, lr = lane & 31, lh = lane >> 5;
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    float4 a[2], b[2];
                    for (int mi = 0; mi < 2; ++mi) a[mi] = *(const float4*)(st + (wr * 64 + mi * 32 + lr) * ROW + kg * 8 + lh * 4);
                    for (int ni = 0; ni < 2; ++ni) b[ni] = *(const float4*)(st + TILE + (wc * 64 + ni * 32 + lr) * ROW + kg * 8 + lh * 4);
                    const float* a0 = (const float*)&a[0]; const float* a1 = (const float*)&a[1];
                    const float* b0 = (const float*)&b[0]; const float* b1 = (const float*)&b[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    } else {
        const int wr = wid >> 1, wc = wid & 1, lr = lane & 31, lh = lane >> 5;
        float4 r[8];
        for (int i = 0; i < 8; ++i) { int s = tid + 256 * i; r[i] = *(const float4*)(A + (base + (s >> 3)) * 256 + 4 * (s & 7)); }
        for (int it = 0; it < iters; ++it) {
            float* st = lds + (it & 1) * STAGE;
            float* nx = lds + ((it + 1) & 1) * STAGE;
            for (int i = 0; i < 8; ++i) { int s = tid + 256 * i; *(float4*)(nx + (s >> 3) * ROW + 4 * (s & 7)) = r[i]; }
            const int k0 = ((it + 2) * 32) & 255;
            for (int i = 0; i < 8; ++i) { int s = tid + 256 * i; r[i] = *(const float4*)(A + (base + (s >> 3) + it) * 256 + k0 + 4 * (s & 7) % 32); }
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                float4 a[2], b[2];
                for (int mi = 0; mi < 2; ++mi) a[mi] = *(const float4*)(st + (wr * 64 + mi * 32 + lr) * ROW + kg * 8 + lh * 4);
                for (int ni = 0; ni < 2; ++ni) b[ni] = *(const float4*)(st + TILE + (wc * 64 + ni * 32 + lr) * ROW + kg * 8 + lh * 4);
                const float* a0 = (const float*)&a[0]; const float* a1 = (const float*)&a[1];
                const float* b0 = (const float*)&b[0]; const float* b1 = (const float*)&b[1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + tid] = s;
}

int main() {
    const long rows = 1 << 20;
    float *A, *out;
    hipMalloc(&A, rows * 256 * 4); hipMalloc(&out, 1 << 22);
    std::vector<float> h(1 << 22);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2 - 1;
    for (long o = 0; o < rows * 256; o += h.size()) hipMemcpy(A + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, lds = 2 * STAGE * 4;
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int mode : {0, 1, 0, 1}) {
        const int blocks = mode == 0 ? 256 : 512;     // mode 0: one 8-wave workgroup per CU; mode 1: two 4-wave workgroups per CU
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), lds, 0, A, out, iters, rows);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), lds, 0, A, out, iters, rows);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * 4 * iters * 64.0 * 4096;      // 4 MFMA waves per workgroup, 64 MFMAs per iteration
            if (rep) printf("mode %d (%s): %.3f ms  %.1f TFLOP/s\n", mode, mode == 0 ? "4 MFMA + 4 loader waves, 1 WG/CU" : "4 waves do both, 2 WG/CU", ms, fl / ms / 1e9);
        }
    }
    return 0;
}
