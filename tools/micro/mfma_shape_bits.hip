// Do the two bf16 MFMA shapes round alike?  (gfx950)
// The three-launch bf16 path and the 64-row fused kernels accumulate with v_mfma_f32_32x32x16_bf16 over ascending 16-k blocks; a
// kernel whose waves own 16 rows would use v_mfma_f32_16x16x32_bf16 (32 k per instruction).  Bit-for-bit tests between the two need
//     mfma_16x16x32(A[:, 0:32], B[0:32, :], C)  ==  mfma_32x32x16(A[:, 16:32], B[16:32, :], mfma_32x32x16(A[:, 0:16], B[0:16, :], C))
// element for element.  This probe compares them on random bf16 data (several scales, long accumulation chains) and also
// prints how both relate to an fp32 fma chain in ascending k and to the exactly rounded sum (fp64).
// Build + run: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_bits mfma_shape_bits.hip && ./mfma_shape_bits
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: 32 x K row-major bf16 (as ushort), B: K x 32 (stored [n][k], i.e. B^T row-major), K multiple of 32; one wave.
__global__ void probe(const unsigned short* A, const unsigned short* Bt, int K, float* out32, float* out16) {
    const int lane = threadIdx.x;
    // 32x32x16: lane (r = l & 31, h = l >> 5): A[r][16 kb + 8 h + j], B[16 kb + 8 h + j][col r]
    {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int r = lane & 31, h = lane >> 5;
        for (int kb = 0; kb < K / 16; ++kb) {
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) {
                a[j] = __builtin_bit_cast(__bf16, A[r * K + 16 * kb + 8 * h + j]);
                b[j] = __builtin_bit_cast(__bf16, Bt[r * K + 16 * kb + 8 * h + j]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h, col = r;
            out32[row * 32 + col] = acc[i];
        }
    }
    // 16x16x32 on the four 16 x 16 quadrants: lane (r = l & 15, g = l >> 4): A[r][32 kb + 8 g + j]
    for (int qi = 0; qi < 2; ++qi)
        for (int qj = 0; qj < 2; ++qj) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int r = lane & 15, g = lane >> 4;
            for (int kb = 0; kb < K / 32; ++kb) {
                bf16x8 a, b;
                for (int j = 0; j < 8; ++j) {
                    a[j] = __builtin_bit_cast(__bf16, A[(16 * qi + r) * K + 32 * kb + 8 * g + j]);
                    b[j] = __builtin_bit_cast(__bf16, Bt[(16 * qj + r) * K + 32 * kb + 8 * g + j]);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) out16[(16 * qi + 4 * g + i) * 32 + 16 * qj + r] = acc[i];
        }
}

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
    return (unsigned short)u;
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    int total_diff = 0;
    for (int trial = 0; trial < 12; ++trial) {
        const int K = trial < 6 ? 320 : 2048;
        const float scale = trial % 3 == 0 ? 1.f : (trial % 3 == 1 ? 37.f : 1e-3f);
        std::vector<unsigned short> A(32 * K), B(32 * K);
        srand(100 + trial);
        for (auto& v : A) v = f2bf(scale * ((rand() / (float)RAND_MAX) * 2 - 1) * (trial % 2 ? 1.f : expf(4.f * (rand() / (float)RAND_MAX))));
        for (auto& v : B) v = f2bf((rand() / (float)RAND_MAX) * 2 - 1);
        unsigned short *dA, *dB; float *d32, *d16;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&d32, 4096); hipMalloc(&d16, 4096);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, K, d32, d16);
        float o32[1024], o16[1024];
        hipMemcpy(o32, d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(o16, d16, 4096, hipMemcpyDeviceToHost);
        int diff = 0, d_fma32 = 0, d_fma16 = 0, d_exact32 = 0, d_exact16 = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float chain = 0.f; double ex = 0.0;
                for (int k = 0; k < K; ++k) {
                    const float a = bf2f(A[i * K + k]), b = bf2f(B[j * K + k]);
                    chain = fmaf(a, b, chain);
                    ex += (double)a * b;
                }
                const float v32 = o32[i * 32 + j], v16 = o16[i * 32 + j];
                diff += memcmp(&v32, &v16, 4) != 0;
                d_fma32 += memcmp(&v32, &chain, 4) != 0; d_fma16 += memcmp(&v16, &chain, 4) != 0;
                const float exf = (float)ex;
                d_exact32 += memcmp(&v32, &exf, 4) != 0; d_exact16 += memcmp(&v16, &exf, 4) != 0;
            }
        printf("trial %2d K %4d scale %-6g: 32x32x16 vs 16x16x32 differ in %4d / 1024 elements | vs fma chain: %4d / %4d | vs rounded exact sum: %4d / %4d\n", trial, K,
               scale, diff, d_fma32, d_fma16, d_exact32, d_exact16);
        total_diff += diff;
        hipFree(dA); hipFree(dB); hipFree(d32); hipFree(d16);
    }
    printf(total_diff ? "the two shapes do NOT round alike\n" : "the two shapes agree bit for bit on all trials\n");
    return 0;
}
