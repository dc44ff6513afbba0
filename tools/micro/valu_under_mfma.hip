// How fast do VALU instructions of one wave issue while OTHER waves of the same SIMD keep the matrix pipe busy with
// v_mfma_f32_32x32x2_f32?  (Epilogue / prologue waves of a GEMM share their SIMD with the K loops of other workgroups.)
// Workgroups alternate roles by blockIdx parity: even = MFMA loop (or idle in the control run), odd = a VALU loop of
// independent FMAs (or of transcendental ops), at default or at raised wave priority (s_setprio 3); LDS sizing puts `per_cu` workgroups on every CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const float* in, float* out, long* ticks, int mfma_iters, int valu_iters, int mode, int nmfma, int prio) {
    extern __shared__ float lds[];
    const int tid = blockIdx.x * 256 + threadIdx.x;
    lds[threadIdx.x] = in[tid & 0xFFFFF];
    const bool valu_role = (blockIdx.x % (nmfma + 1)) == nmfma;
    float s = 0.f;
    if (!valu_role) {
        float a = in[(tid * 3) & 0xFFFFF], b = in[(tid * 5 + 1) & 0xFFFFF];
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = in[(tid + i * 977) & 0xFFFFF];
        if (prio) __builtin_amdgcn_s_setprio(3);
        const long t0 = wall_clock64();
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (mode == 0) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
                    else x[i] = __builtin_amdgcn_rcpf(x[i] + 1.5f);
                }
            }
        }
        const long t1 = wall_clock64();
        for (int i = 0; i < 8; ++i) s += x[i];
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    }
    out[tid] = s + lds[(threadIdx.x + 1) & 255];
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2;        // workgroups per CU; one of them has the VALU role
    const int blocks = 256 * per_cu;
    float *in, *out; long* ticks;
    hipMalloc(&in, (1 << 20) * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ticks, blocks * 8);
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (float)rand() / (float)RAND_MAX;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const size_t ldsb = (size_t)(128 / per_cu) * 1024;
    const int valu_iters = 2000;                              // 2000 x 64 VALU instructions per wave
    for (int prio = 0; prio < 2; ++prio)
    for (int mode = 0; mode < 2; ++mode)
        for (int busy = 0; busy < 2; ++busy) {
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(ticks, 0, blocks * 8);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), ldsb, 0, in, out, ticks, busy ? 6000 : 0, valu_iters, mode, per_cu - 1, prio);
                hipDeviceSynchronize();
            }
            std::vector<long> t(blocks);
            hipMemcpy(t.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int i = 0; i < blocks; ++i) if (t[i] > 0) { sum += t[i]; ++n; }
            const double us = sum / n / 100.0, instr = (double)valu_iters * 64;
            printf("%d WG/CU, VALU wave priority %d, %s, MFMA neighbours %s: %.1f us for %.0f VALU instructions per wave = %.1f ns (%.1f cycles @2.4 GHz) each\n", per_cu, prio ? 3 : 0,
                   mode ? "v_rcp_f32+v_add" : "v_fma_f32", busy ? "busy" : "idle", us, instr, 1e3 * us / instr, 2.4e3 * us / instr);
        }
    return 0;
}
