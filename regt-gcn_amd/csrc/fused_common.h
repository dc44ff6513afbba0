// Device helpers shared by the fused kernels of the bf16 arithmetic (fused.hip: 64-row tiles, activations as A operands in LDS
// planes, weights straight from L2; fused_rows.hip: 128-row tiles, a wave owns 16 whole rows, weights streamed through LDS).
#pragma once
#include <stdlib.h>

#include "kernels.h"
#include "gemm_split.h"

namespace regt {
namespace {

constexpr int FT_TRACE_SLOTS = 32;       // shader-clock stamps per tile of the developer trace (REGT_FUSED_TRACE, tools/fused_trace.py)

__device__ __forceinline__ float f_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float f_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float4 f_widen4(unsigned lo, unsigned hi) {
    return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u));
}
struct V8 { float v[8]; };
// Packed (two-lane) fp32 arithmetic for the epilogues: a VALU instruction of this kernel costs matrix-pipe time of its SIMD partner
// (DESIGN 5c.1), and v_pk_add / v_pk_mul / v_pk_fma do two elements per instruction with the same IEEE results as the scalar forms.
// The operation SEQUENCES are those of fast_sigmoid / fast_tanh (gemm.hip): mul by -log2(e), exp2, add 1, rcp -- bit for bit.
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ V8 f_sigmoid8(const V8& v, const V8& b) {
    V8 o;
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 2)      // timing-only developer build: no gate arithmetic
    for (int i = 0; i < 8; ++i) o.v[i] = v.v[i] + b.v[i];
    return o;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f2_t s = {v.v[2 * i], v.v[2 * i + 1]};
        const f2_t bb = {b.v[2 * i], b.v[2 * i + 1]};
        s = s + bb;
        const f2_t t = s * -1.44269502162933349609375f;            // (0xbfb8aa3b: the constant __expf(-x) multiplies by)
        f2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        e = e + 1.0f;
        o.v[2 * i] = __builtin_amdgcn_rcpf(e.x);
        o.v[2 * i + 1] = __builtin_amdgcn_rcpf(e.y);
    }
    return o;
}
__device__ __forceinline__ V8 f_tanh8(const V8& v, const V8& b) {
    V8 o;
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 2)
    for (int i = 0; i < 8; ++i) o.v[i] = v.v[i] + b.v[i];
    return o;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f2_t s = {v.v[2 * i], v.v[2 * i + 1]};
        const f2_t bb = {b.v[2 * i], b.v[2 * i + 1]};
        s = s + bb;
        s = s + s;                                                  // 2 x (exact), as fast_tanh's __expf(2.0f * x)
        const f2_t t = s * 1.44269502162933349609375f;
        f2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        e = e + 1.0f;
        const f2_t r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
        const f2_t one = {1.0f, 1.0f}, m2 = {-2.0f, -2.0f};
        const f2_t h = __builtin_elementwise_fma(r, m2, one);       // 1 - 2 r as ONE fma (what hipcc contracts fast_tanh's last step into)
        o.v[2 * i] = h.x;
        o.v[2 * i + 1] = h.y;
    }
    return o;
}
__device__ __forceinline__ V8 f_widen8(u32x4_t r) {
    V8 o;
    o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
    o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    o.v[4] = __uint_as_float(r.z << 16); o.v[5] = __uint_as_float(r.z & 0xffff0000u);
    o.v[6] = __uint_as_float(r.w << 16); o.v[7] = __uint_as_float(r.w & 0xffff0000u);
    return o;
}
__device__ __forceinline__ u32x4_t f_pack8(const V8& a) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    u32x4_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t p = {a.v[2 * i], a.v[2 * i + 1]};
        r[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2_t));
    }
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes > 0x7ffffff0L ? 0x7ffffff0 : (int)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 f_ldfrag(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 16)     // timing-only developer build: weight fragments are not loaded at all
__device__ __forceinline__ bf16x8 f_ldw(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("" : "+v"(z) : "v"(voff), "s"(soff));
    return z;
}
#else
#define f_ldw f_ldfrag
#endif

}  // namespace
}  // namespace regt
