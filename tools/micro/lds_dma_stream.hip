// LDS-DMA streaming probe (gfx950): can a workgroup keep S stages of 16-byte `global_load_lds` requests in flight without a
// register ring, with the XOR-swizzled row image the weight-gradient kernels read, and at what rate does it stream?
//   * every workgroup walks a row chunk of a bf16 matrix (rows x 128 columns = 256 B per row) in 16-row stages (4 KB): one
//     __builtin_amdgcn_global_load_lds(…, 16, …) per thread and stage; lane l of a wave instruction lands at LDS base + 16 l, so
//     the swizzle is applied on the SOURCE address: LDS slot (row, pos) is filled from global chunk pos ^ swz(row);
//   * S stages are requested ahead; a stage is consumed (ds_read_b128 of the thread's own slot + every other thread's, summed)
//     after `s_waitcnt vmcnt(S - 1)` + barrier -- the waits are written by hand (the compiler is asked for nothing: the loads are
//     behind inline-asm-free builtins, so its own vmcnt(0) in front of LDS reads would show in the ISA and in the rate);
//   * checks: the per-workgroup sums against a host sum (bit-exact integer sums), and that the swizzled image holds what
//     ws_off() expects; prints GB/s for S = 1, 2, 4, 8.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o lds_dma_stream lds_dma_stream.hip && ./lds_dma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int swz(int m) { return ((m & 3) << 2) | ((m >> 2) & 3); }      // ws_off's chunk permutation (gemm.hip)

template <int S>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ x, unsigned long long* __restrict__ sums, long rows, int chunk_rows,
                                                     int* __restrict__ image_ok) {
    extern __shared__ __attribute__((aligned(16))) char lds[];                  // S stages x 4 KB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long r0 = (long)blockIdx.x * chunk_rows;
    const int nst = chunk_rows / 16;
    // lane l of wave w fills LDS slot (row = 4 w + l / 16, pos = l % 16) of a stage: source chunk pos ^ swz(row)
    const int row = 4 * wave + (lane >> 4), pos = lane & 15;
    const uint4* src = x + (r0 + row) * 16 + (pos ^ swz(row));
    auto issue = [&](int st) {                                                  // stage st of the chunk -> ring slot st % S
        char* dst = lds + (st % S) * 4096 + wave * 1024;                        // wave-uniform LDS base (M0); lane offset is implicit
#ifdef USE_BUILTIN
        // the compiler knows this one writes LDS behind vmcnt -- and drains it (vmcnt(0)) in front of EVERY LDS read: no pipelining
        __builtin_amdgcn_global_load_lds((glb_void*)(src + (long)st * 256), (lds_void*)dst, 16, 0, 0);
#else
        // invisible to the compiler's wait-count insertion: the waits below are the only ones (no other vector-memory load may be
        // outstanding in this loop -- the compiler's own vmcnt arithmetic would be off by the requests it does not know about)
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)dst);
        const uint4* g = src + (long)st * 256;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(g) : "memory");
#endif
    };
    for (int st = 0; st < S && st < nst; ++st) issue(st);
    unsigned long long acc = 0;
    int ok = 1;
    for (int st = 0; st < nst; ++st) {
        // the oldest of the (up to) S outstanding requests of this wave has landed; all waves' -> barrier
        // s_waitcnt immediate on gfx9: vmcnt = bits 3:0 and 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8 (15 = no wait)
        constexpr int VM = S - 1, IMM = 0x0F70 | (VM & 15) | ((VM >> 4) << 14);
        if (S == 1) __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        else if (st + S <= nst) __builtin_amdgcn_s_waitcnt(IMM);                // vmcnt(S - 1): the oldest request has landed
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        const uint4* stg = reinterpret_cast<const uint4*>(lds + (st % S) * 4096);
        // every thread reads the 16 slots of "its" row transposed back (slot pos' holds source chunk pos' ^ swz(row)) and sums
        const int rr = tid >> 4, cc = tid & 15;
        const uint4 v = stg[rr * 16 + (cc ^ swz(rr))];                          // = source chunk cc of row rr
        acc += (unsigned long long)v.x + v.y + v.z + v.w;
        if (st == 0 && blockIdx.x == 0) {                                       // the image is what ws_off() addresses
            const uint4 w = x[(r0 + rr) * 16 + cc];
            if (w.x != v.x || w.y != v.y || w.z != v.z || w.w != v.w) ok = 0;
        }
        __syncthreads();                                                        // everybody is done with the slot before it is refilled
        if (st + S < nst) issue(st + S);
    }
    if (blockIdx.x == 0 && !ok) atomicExch(image_ok, 0);
    // workgroup sum (integer: order-free)
    __shared__ unsigned long long red[256];
    red[tid] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    if (tid == 0) sums[blockIdx.x] = red[0];
}

template <int S>
static void run(const uint4* d, unsigned long long* dsums, int* dok, long rows, int chunk_rows, const std::vector<unsigned long long>& want,
                int lds_bytes = 0) {                     // lds_bytes > S * 4096: pad the allocation so that fewer workgroups share a CU
    const int blocks = (int)(rows / chunk_rows);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int one = 1;
    hipMemcpy(dok, &one, 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) {
        if (it == 1) hipEventRecord(e0);
        const int bytes = lds_bytes > S * 4096 ? lds_bytes : S * 4096;
        if (bytes > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL(stream_kernel<S>, dim3(blocks), dim3(256), bytes, 0, d, dsums, rows, chunk_rows, dok);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> got(blocks);
    hipMemcpy(got.data(), dsums, blocks * 8, hipMemcpyDeviceToHost);
    int ok = 0, bad = 0;
    hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b) bad += got[b] != want[b];
    printf("S = %d stages (%d KB per workgroup, %s): %.1f us per pass, %.0f GB/s, %d of %d workgroup sums wrong, swizzled image %s\n", S, 4 * S,
           lds_bytes > 80 * 1024 ? "1 workgroup per CU" : lds_bytes > 40 * 1024 ? "2 workgroups per CU" : "8 per CU", 500.0 * ms,
           rows * 256.0 / (0.5e-3 * ms) / 1e9, bad, blocks, ok ? "ok" : "WRONG");
}

int main() {
    const long rows = 1L << 22;              // 4 M rows x 256 B = 1 GiB
    const int chunk_rows = 4096;             // 1024 workgroups
    std::vector<u32> h(rows * 64);
    u32 s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
    std::vector<unsigned long long> want(rows / chunk_rows, 0);
    for (long r = 0; r < rows; ++r) {
        unsigned long long a = 0;
        for (int c = 0; c < 64; ++c) a += h[r * 64 + c];
        want[r / chunk_rows] += a;
    }
    uint4* d; unsigned long long* dsums; int* dok;
    if (hipMalloc(&d, rows * 256) != hipSuccess || hipMalloc(&dsums, 8 * (rows / chunk_rows)) != hipSuccess || hipMalloc(&dok, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemcpy(d, h.data(), rows * 256, hipMemcpyHostToDevice);
    run<1>(d, dsums, dok, rows, chunk_rows, want);
    run<2>(d, dsums, dok, rows, chunk_rows, want);
    run<4>(d, dsums, dok, rows, chunk_rows, want);
    run<8>(d, dsums, dok, rows, chunk_rows, want);
    // what a register-starved kernel has: two (one) workgroups per CU -- the ring depth is all that hides the latency
    run<1>(d, dsums, dok, rows, chunk_rows, want, 72 * 1024);
    run<2>(d, dsums, dok, rows, chunk_rows, want, 72 * 1024);
    run<4>(d, dsums, dok, rows, chunk_rows, want, 72 * 1024);
    run<8>(d, dsums, dok, rows, chunk_rows, want, 72 * 1024);
    run<16>(d, dsums, dok, rows, chunk_rows, want, 72 * 1024);
    run<4>(d, dsums, dok, rows, chunk_rows, want, 150 * 1024);
    run<8>(d, dsums, dok, rows, chunk_rows, want, 150 * 1024);
    run<16>(d, dsums, dok, rows, chunk_rows, want, 150 * 1024);
    run<32>(d, dsums, dok, rows, chunk_rows, want, 150 * 1024);
    return 0;
}
