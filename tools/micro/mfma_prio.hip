// Two waves per SIMD issuing v_mfma_f32_32x32x2_f32 reach only ~2/3 of the one-wave rate (tools/micro/mfma_clock.hip).
// Question: is that the alternation of the two waves' MFMAs on the shared matrix pipe, and does a static priority for ONE
// of the two co-resident workgroups (chosen by its LDS allocation base, HW_REG_LDS_ALLOC) restore the one-wave rate?
//   variant 0: plain;  1: s_setprio 1 for the workgroup whose LDS base is 0;  2: s_setprio 3 for it;  3: prio by wave-slot parity
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters, int* bases) {
    extern __shared__ float lds[];
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const unsigned alloc = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 6);      // HW_REG_LDS_ALLOC[15:0]
    const unsigned base = alloc & 0x1ff;
    if (threadIdx.x == 0) bases[blockIdx.x] = (int)alloc;
    if (VAR == 1) { if (base == 0) __builtin_amdgcn_s_setprio(1); }
    if (VAR == 2) { if (base == 0) __builtin_amdgcn_s_setprio(3); }
    if (VAR == 3) {
        const unsigned hwid = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID: wave_id [3:0]
        if (hwid & 1) __builtin_amdgcn_s_setprio(2);
    }
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFFF]; b[i] = in[(tid * 8 + i + 77) & 0xFFFFF]; }
    lds[threadIdx.x] = a[0];
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
    float s = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int blocks = 256 * per_cu, iters = 4000;
    float *in, *out; int* bases;
    hipMalloc(&in, (1 << 20) * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&bases, blocks * 4);
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2 - 1;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t ldsb = (size_t)(128 / per_cu) * 1024;     // exactly per_cu workgroups per CU by LDS
    for (int round = 0; round < 3; ++round)                  // three rounds: the first launches run on a cold (low) clock
    for (int var = 0; var < 4; ++var) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (var) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), ldsb, 0, in, out, iters, bases); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), ldsb, 0, in, out, iters, bases); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), ldsb, 0, in, out, iters, bases); break;
                default: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), ldsb, 0, in, out, iters, bases); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * 4 * iters * 32.0 * 4096;
            if (rep) printf("round %d blocks/CU %d variant %d: %.3f ms  %.1f TFLOP/s\n", round, per_cu, var, ms, fl / ms / 1e9);
        }
    }
    std::vector<int> hb(blocks);
    hipMemcpy(hb.data(), bases, blocks * 4, hipMemcpyDeviceToHost);
    printf("LDS_ALLOC of blocks 0..7: ");
    for (int i = 0; i < 8 && i < blocks; ++i) printf("0x%x ", hb[i]);
    printf(" ... block 256: 0x%x\n", blocks > 256 ? hb[256] : 0);
    return 0;
}
