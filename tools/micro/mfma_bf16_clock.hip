// Sustained rate of v_mfma_f32_32x32x16_bf16 on random data (what would a 3-way bf16 split of the fp32 GEMMs run at?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(const uint4* in, float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[6], b[6];
    for (int i = 0; i < 6; ++i) {
        uint4 u = in[(tid * 6 + i) & 0xFFFF], v = in[(tid * 6 + i + 77) & 0xFFFF];
        a[i] = *reinterpret_cast<bf16x8*>(&u);
        b[i] = *reinterpret_cast<bf16x8*>(&v);
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + 1) % 6], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + 1) % 6], b[j], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + 3) % 6], b[(j + 2) % 6], acc[3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int blocks = 256 * (argc > 1 ? atoi(argv[1]) : 2), iters = 8000;
    uint4* in; float* out;
    hipMalloc(&in, (1 << 16) * 16); hipMalloc(&out, blocks * 256 * 4);
    std::vector<unsigned short> h((1 << 16) * 8);
    for (auto& v : h) { float f = (float)rand() / RAND_MAX * 2 - 1; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 24.0 * (2.0 * 32 * 32 * 16);
        printf("bf16 32x32x16, %d blocks: %.3f ms  %.1f TFLOP/s bf16  = %.1f TFLOP/s fp32-equivalent at 6 products\n", blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 6);
    }
    return 0;
}
