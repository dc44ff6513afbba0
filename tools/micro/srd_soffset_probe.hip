// Is the scalar offset of a raw buffer load part of the range check on gfx950?  (LLVM documents soffset as "excluded from
// bounds checking"; SplitCore::run_u relies on it: num_records = bytes of the valid rows, k offset in soffset.)
// num_records = 64 bytes; lane 0 loads one dword at voffset 32, soffset 48: data[20] if soffset is excluded (32 < 64),
// 0 if the check is voffset >= num_records - soffset (32 >= 16).  Second probe: num_records = 0, soffset 64 -> must be 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* data, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(data), 0, 64, 0x00020000);
    __amdgpu_buffer_rsrc_t z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(data), 0, 0, 0x00020000);
    const int soff = (int)data[1024];      // 48, opaque to the compiler
    out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 32, soff, 0));
    out[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 64, soff - 48, 0));   // voffset == num_records: 0
    out[2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(z, 0, soff + 16, 0));     // num_records 0, soffset 64: 0
    out[3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 60, soff + 16, 0));    // last dword + soffset 64: data[31]
}
int main() {
    float h[1025], *d, *o, r[4];
    for (int i = 0; i < 1024; ++i) h[i] = 100.f + i;
    h[1024] = 48.f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("voffset 32 + soffset 48, num_records 64: %.0f (120 = soffset excluded from the range check, 0 = included)\n", r[0]);
    printf("voffset 64 (== num_records): %.0f (expect 0)\nnum_records 0, soffset 64: %.0f (expect 0)\nvoffset 60 + soffset 64: %.0f (131 = excluded)\n", r[1], r[2], r[3]);
    return 0;
}
