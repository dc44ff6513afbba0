#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
// buffer_load_dwordx4 ... lds with a range-checked descriptor: do out-of-range lanes write ZEROS into LDS (or leave it alone)?
__global__ void k(const uint4* x, uint4* out, int valid_bytes) {
    __shared__ __attribute__((aligned(16))) uint4 img[64];
    img[threadIdx.x] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(x), 0, valid_bytes, 0x00020000);
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)img);
    const unsigned off = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" :: "s"(m0v), "v"(off), "s"(r) : "memory");
    __syncthreads();
    out[threadIdx.x] = img[threadIdx.x];
}
int main() {
    uint4 h[64], o[64], *d, *dout;
    for (int i = 0; i < 64; ++i) h[i] = make_uint4(i + 1, i + 1, i + 1, i + 1);
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout, 40 * 16);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("lane 0: %u (expect 1), lane 39: %u (expect 40), lane 40 (first out of range): 0x%x, lane 63: 0x%x  -> %s\n", o[0].x, o[39].x, o[40].x, o[63].x,
           o[40].x == 0 ? "out-of-range lanes write ZEROS to LDS" : (o[40].x == 0xAAAAAAAAu ? "out-of-range lanes leave LDS untouched" : "unexpected"));
    return 0;
}
