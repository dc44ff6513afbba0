// Probe of ds_read_b64_tr_b16 (gfx950): which (row, column) of a row-major 16-bit LDS tile lands in which lane/element?
// tile[k][i] = 256*k + i (16 rows x 128 columns); lane 4q+p of a 16-lane group supplies the address of block row q,
// columns 4p..4p+3 (cdna_hip_programming.md T10).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[16 * 128];
    for (int i = threadIdx.x; i < 16 * 128; i += 64) tile[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const unsigned short* a = tile + (8 * (g >> 1) + q) * 128 + 16 * (g & 1) + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short h[16 * 128], o[256], *din, *dout;
    for (int k = 0; k < 16; ++k) for (int i = 0; i < 128; ++i) h[k * 128 + i] = (unsigned short)(256 * k + i);
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (k=%2d,i=%3d)", o[l * 4 + e] / 256, o[l * 4 + e] % 256);
        printf("\n");
    }
    return 0;
}
