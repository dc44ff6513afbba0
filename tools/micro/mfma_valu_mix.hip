// fp32 MFMA rate (v_mfma_f32_32x32x2_f32) of waves that interleave K independent VALU instructions per MFMA, at 1-3 waves
// per SIMD.  Question behind it: how many vector instructions (address arithmetic, v_readfirstlane, conversions) can a
// GEMM K loop afford per MFMA before the matrix pipe starves, when several workgroups share a CU?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: K x v_fma_f32 per MFMA; 1: K x ds_read_b32; 2: K x ds_read_b128; 3: K x buffer/global_load_dword (L2-resident)
template <int K, int KIND = 0>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    extern __shared__ float lds[];
    const int tid = blockIdx.x * 256 + threadIdx.x;
    lds[threadIdx.x] = in[tid & 0xFFFFF];
    float a = in[(tid * 3) & 0xFFFFF], b = in[(tid * 5 + 1) & 0xFFFFF];
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[(tid + i * 977) & 0xFFFFF];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < K; ++v) {
                float& t = x[(j * K + v) & 7];
                if (KIND == 0) t = __builtin_fmaf(t, 1.0001f, 0.5f);
                else if (KIND == 1) t += lds[(threadIdx.x + 64 * (j * K + v)) & 4095];
                else if (KIND == 2) { const float4 q = *reinterpret_cast<const float4*>(lds + ((4 * threadIdx.x + 256 * (j * K + v)) & 4095)); t += q.x; }
                else t += in[(tid + 4096 * (j * K + v + it)) & 0xFFFFF];
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(KIND == 0 ? 0x002 : (KIND == 3 ? 0x020 : 0x100), K, 0);
        }
    }
    float s = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[tid] = s;
}

template <int K, int KIND = 0>
static void run(int per_cu, const float* in, float* out) {
    const int blocks = 256 * per_cu, iters = 2000;
    const size_t ldsb = (size_t)(128 / per_cu) * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<K, KIND>), dim3(blocks), dim3(256), ldsb, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    static const char* names[] = {"v_fma_f32", "ds_read_b32", "ds_read_b128", "global_load_dword"};
    printf("%d waves/SIMD, %d x %s (+ its add) per MFMA: %.1f TFLOP/s\n", per_cu, K, names[KIND], (double)blocks * 4 * iters * 8.0 * 4096 / ms / 1e9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, (1 << 20) * 4); hipMalloc(&out, 256 * 4 * 256 * 4);
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (float)rand() / (float)RAND_MAX;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
        run<0>(per_cu, in, out); run<1>(per_cu, in, out); run<2>(per_cu, in, out); run<4>(per_cu, in, out); run<8>(per_cu, in, out);
        run<1, 1>(per_cu, in, out); run<2, 1>(per_cu, in, out); run<1, 2>(per_cu, in, out); run<1, 3>(per_cu, in, out);
    }
    return 0;
}
