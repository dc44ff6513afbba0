// Sustained fp32 MFMA rate on random data: 32x32x2 vs 16x16x4 (which shape lets the chip hold a higher clock?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFFF]; b[i] = in[(tid * 8 + i + 77) & 0xFFFFF]; }
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 1) & 7], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 1) & 7], b[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 3) & 7], b[(j + 2) & 7], acc[3], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[tid] = s;
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + t) & 7], b[(j + 2 * t + 1) & 7], acc[t], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[tid] = s;
    }
}

int main(int argc, char** argv) {
    const int blocks = 256 * (argc > 1 ? atoi(argv[1]) : 2), iters = 4000;
    float *in, *out;
    hipMalloc(&in, (1 << 20) * 4); hipMalloc(&out, blocks * 256 * 4);
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2 - 1;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape : {32, 16, 32, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // flops: 32x32x2: 4096 per MFMA, 32 MFMA/iter/wave ; 16x16x4: 2048 per MFMA, 128 MFMA/iter/wave
            double fl = (double)blocks * 4 * iters * (shape == 32 ? 32.0 * 4096 : 128.0 * 2048);
            if (rep) printf("shape %dx%d: %.3f ms  %.1f TFLOP/s\n", shape, shape, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
