// Vector-memory path probe (gfx950): what does ONE 16-byte-per-lane wave instruction (1 KB) cost a CU's texture path, by shape?
// The fused forward kernel (csrc/fused.hip) moves 8.5 KB of weight fragments (L2 hits, straight to registers) and 2.5 KB of
// activation stores per row through that path; its counters say TA_BUSY 0.74 / TD_BUSY 0.88 at 18 B/clk/CU of loads.  This
// probe measures the rate of each access shape alone, everything L2-resident (no HBM in the way), 2 workgroups of 4 waves per CU:
//   load  contiguous : lane l reads 16 B at base + 16 l of a 544 KB table (the weight-fragment loads)
//   load  rows64     : 16 rows x 64 B (4 lanes per row, 512 B row stride) -- an activation tile read
//   store contiguous / rows64 / rows32 / rows128 : the same shapes as stores (rows32: 32 rows x 32 B; rows128: 8 rows x 128 B)
//   mixed            : 5 contiguous loads per rows64 store (the kernel's ratio)
//   LDS-DMA          : the contiguous load as `global_load_lds_dwordx4` into an LDS ring (what fused_rows.hip streams its weights with)
// Prints bytes per clock and CU and the cycles one wave instruction occupies the CU's path (at the measured shader clock).
// Build + run: hipcc --offload-arch=gfx950 -O3 -o ta_path ta_path.hip && ./ta_path
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// shape: 0 contiguous, 1 rows64, 2 rows32, 3 rows128
__device__ __forceinline__ int shape_off(int shape, int lane) {
    switch (shape) {
        case 0: return lane * 16;
        case 1: return (lane >> 2) * 512 + (lane & 3) * 16;
        case 2: return (lane >> 1) * 512 + (lane & 1) * 16;
        default: return (lane >> 3) * 512 + (lane & 7) * 16;
    }
}

template <int MODE, int SHAPE, int U>      // MODE 0 loads (U in flight per wave), 1 stores, 2 mixed (5 contiguous loads + 1 store of SHAPE)
__global__ __launch_bounds__(256, 2) void probe(const char* table, int table_bytes, char* out, int iters, long long* clocks, u32x4* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t st = rsrc(table, table_bytes);
    // every workgroup owns a 64 KB slice of `out` (rewritten over and over: stays in L2); a wave's instruction covers at most 16 KB of it
    char* mine = out + ((long)blockIdx.x * 4 + wave) * 16384;
    const __amdgpu_buffer_rsrc_t so = rsrc(mine, 16384);
    const int voff = shape_off(SHAPE, lane);
    u32x4 acc = {0, 0, 0, 0}, val = {(unsigned)lane, 1u, 2u, 3u};
    const long long t0 = __builtin_amdgcn_s_memtime();
    int toff = (blockIdx.x * 4 + wave) * 8192 % table_bytes;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = SHAPE == 0 ? (toff + u * 1024) % table_bytes : (toff + u * 8192) % (table_bytes - 8192);
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(st, voff, o, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
            toff = (toff + U * 1024) % table_bytes;
        } else if (MODE == 3) {
            // LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction, lane-linear destination): U requests in flight per wave,
            // into a per-wave ring of U slots of this workgroup's LDS; the oldest is waited for before its slot is reused
            extern __shared__ char ring[];
            const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ring + (wave * U) * 1024);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((toff + u * 1024) % table_bytes));
                const char* src = table + so;
                const unsigned m0v = base + u * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(lane * 16), "s"(src) : "memory");
            }
            toff = (toff + U * 1024) % table_bytes;
            if (U >= 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // (rows shapes: 8 instructions tile the wave's 16 KB -- rows64: 8 column groups of 64 B; others likewise by offset)
                const int o = SHAPE == 0 ? u * 1024 : (SHAPE == 1 ? u * 64 : (SHAPE == 2 ? u * 32 : (u & 3) * 128 + (u >> 2) * 4096));
                __builtin_amdgcn_raw_buffer_store_b128(val, so, voff, o, 0);
            }
            val.y += 1;
        } else {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(st, lane * 16, (toff + u * 1024) % table_bytes, 0);
                acc ^= v;
            }
            toff = (toff + 5120) % table_bytes;
            const int o = (it & 7) * 64;
            __builtin_amdgcn_raw_buffer_store_b128(val, so, voff, o, 0);
            val.y += 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u) sink[0] = acc;
}

template <int MODE, int SHAPE, int U = 8>
static void run(const char* name, const char* table, int table_bytes, char* out, long long* clocks, u32x4* sink, int cus) {
    const int grid = 2 * cus, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, SHAPE, U><<<grid, 256, MODE == 3 ? 4 * U * 1024 : 0>>>(table, table_bytes, out, 200, clocks, sink);
    hipEventRecord(e0);
    probe<MODE, SHAPE, U><<<grid, 256, MODE == 3 ? 4 * U * 1024 : 0>>>(table, table_bytes, out, iters, clocks, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long* h = (long long*)malloc(grid * sizeof(long long));
    hipMemcpy(h, clocks, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < grid; ++i) cyc += (double)h[i];
    cyc /= grid;                                                   // shader cycles of the timed loop (mean over workgroups)
    const double per_iter = MODE == 2 ? 6.0 : (MODE == 0 || MODE == 3 ? (double)U : 8.0);                 // wave instructions per iteration
    const double insts_cu = 8.0 * iters * per_iter;                // 8 waves per CU
    const double ghz = cyc / (ms * 1e6);
    printf("%-28s %8.3f ms  clock %.2f GHz  %6.1f B/clk/CU  %6.1f cycles per wave instruction (CU path)  chip %.2f TB/s\n", name, ms, ghz,
           insts_cu * 1024.0 / cyc, cyc / insts_cu, insts_cu * 1024.0 * cus / (ms * 1e-3) / 1e12);
    free(h);
}

int main() {
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int table_bytes = 544 * 1024;
    char *table, *out; long long* clocks; u32x4* sink;
    hipMalloc(&table, table_bytes); hipMemset(table, 1, table_bytes);
    hipMalloc(&out, (size_t)2 * cus * 65536); hipMemset(out, 0, (size_t)2 * cus * 65536);
    hipMalloc(&clocks, 2 * cus * sizeof(long long)); hipMalloc(&sink, 64);
    printf("%d CUs, 2 workgroups x 4 waves per CU, 16 B per lane\n", cus);
    run<0, 0, 4>("load  contiguous, 4 in flight", table, table_bytes, out, clocks, sink, cus);
    run<0, 0, 8>("load  contiguous, 8 in flight", table, table_bytes, out, clocks, sink, cus);
    run<0, 0, 16>("load  contiguous, 16 in flight", table, table_bytes, out, clocks, sink, cus);
    run<0, 0, 32>("load  contiguous, 32 in flight", table, table_bytes, out, clocks, sink, cus);
    run<3, 0, 4>("LDS-DMA contiguous, 4 in flight", table, table_bytes, out, clocks, sink, cus);
    run<3, 0, 8>("LDS-DMA contiguous, 8 in flight", table, table_bytes, out, clocks, sink, cus);
    run<3, 0, 16>("LDS-DMA contiguous, 16 in flight", table, table_bytes, out, clocks, sink, cus);
    run<0, 1>("load  16 rows x 64 B", table, table_bytes, out, clocks, sink, cus);
    run<1, 0>("store contiguous (1 KB)", table, table_bytes, out, clocks, sink, cus);
    run<1, 3>("store 8 rows x 128 B", table, table_bytes, out, clocks, sink, cus);
    run<1, 1>("store 16 rows x 64 B", table, table_bytes, out, clocks, sink, cus);
    run<1, 2>("store 32 rows x 32 B", table, table_bytes, out, clocks, sink, cus);
    run<2, 1>("5 loads + 1 store (rows64)", table, table_bytes, out, clocks, sink, cus);
    return 0;
}
