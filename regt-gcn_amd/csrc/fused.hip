// Fused forward of the RegT-GCN cell for the bf16 arithmetic (REGT_GEMM_MODE=bf16, BASELINE configs[4]): regional embedding ->
// update / reset gates -> candidate state -> GRU blend -> attention-weighted sum over the periods, ONE kernel per 64-row tile.
//
// Replaces, per period and node (rows m = node * T + t), the op sites
//     h  = leaky_relu(linear(concat_r ChebConv_r(x)))                 models/RegionalTemporalGCN.py:136-143
//     Z  = sigmoid(linear_z([conv_z(x) | h])),  R = sigmoid(linear_r([conv_r(x) | h]))      models/utils.py:168-178
//     H~ = tanh(linear_h([conv_h(x) | h * R])),  H' = Z h + (1 - Z) H~                        models/utils.py:180-188
//     H_accum += softmax(attention)[t] H'                                                    models/RegionalTemporalGCN.py:146
// in the composed-weight form of DESIGN.md section 3.  The three-launch version (gemm_regional, gemm_gates, gemm_candidate)
// writes h, [Z|R], q = h R and H~ for the backward pass AND reads h (twice), Z and q back: 8.5 GB per step at the cfg-5
// shard.  Here a workgroup keeps its 64 rows of h and q in LDS as matrix-core A operands (bf16, the layout of SplitCore's
// stage planes) and Z in registers, so every activation is written once and nothing is read back: 4.4 GB.
//
// Shape of the work: 256 threads = 4 waves; the tile's outputs are produced in 128-column tiles, wave w owns columns
// 32 w .. 32 w + 31 of the tile and all 64 rows (two 32 x 32 accumulators).  The B operands are the per-step bf16 copies of
// the weights in MFMA fragment order (launch_cvt_bf16_frag): a wave loads its fragments of a whole K loop up front, straight
// into registers.  The K = F operands (x, L~ x, A_hat x: bf16 rows written by the aggregation kernel) are loaded as A
// fragments directly from global memory (A_hat x once per tile, kept in registers for all six K loops that use it).
// Epilogues: every wave transposes its OWN 64 x 32 accumulator strip through a wave-private 16-row fp32 image in LDS (4 rounds
// per 64 x 128 tile, no workgroup barrier): a lane then owns (row, 8 consecutive columns) = 16 bytes of every bf16 array -- the
// same lane owns the same (row, columns) in the Z and in the candidate epilogue, which is what lets Z stay in registers.
// LDS: 32 KB h planes + 32 KB q planes + 4 x 2.3 KB images = 74,752 B, two workgroups per CU: one's epilogue VALU and stores
// overlap the other's matrix work.  DESIGN.md section 5d has the measurements behind each of these choices.
//
// Arithmetic, rounding points and summation orders are exactly those of the three-launch path (bf16 MFMA operands, fp32
// accumulate over k in ascending 16-k blocks, fp32 gate math, one rounding per stored element, per-node partial sums over a
// 64-row block in row order): with the same bf16 inputs both paths give bit-identical results (tests/test_gpu_fused.py).
#include "fused_common.h"

namespace regt {

namespace {

constexpr int FT_ROWS = 64;              // rows of a tile
constexpr int FT_IMG_LD = 32;            // floats per row of a wave's image: its 32 columns, no padding -- the 16-byte chunk c of row r
constexpr int FT_IMG_ROWS = 16;          //   lies at chunk c ^ ft_par(r) (below): every access pattern of the epilogues is conflict-free
// Swizzle of a wave's epilogue image.  Three access patterns meet in it: the accumulator rows (ds_write_b32, 32 lanes = one row),
// the (row, 8 columns) reads / writes of the element-wise work (ds_read_b128 / ds_write_b128: lane = (row l >> 2, chunks 2 (l & 3),
// 2 (l & 3) + 1)) and the column reads of the per-node sums (ds_read_b32, 32 lanes = one row).  b128 reads are served in the lane
// groups {0-3, 12-15, 20-27}, ... over 64 banks, i.e. rows {0, 3, 5, 6}, {1, 2, 4, 7} (+8): two rows of equal parity in a group
// must use different chunk parities; b128 writes in groups of 8 lanes = rows {2g, 2g + 1} over 32 banks: the two rows must differ
// too.  par(r) = (r ^ (r >> 2)) & 1 satisfies both (0 1 0 1 1 0 1 0 for rows 0..7).
__device__ __forceinline__ constexpr int ft_par(int r) { return (r ^ (r >> 2)) & 1; }

}  // namespace

template <int C, int F>
struct FusedFwdLds {
    static constexpr int PLANE_B = FT_ROWS * 32;                 // one 16-k block of 64 rows
    static constexpr int OPER_B = (C / 16) * PLANE_B;            // h (or q) of the tile as an A operand
    static constexpr int IMG_OFF = 2 * OPER_B;                   // four wave-private 16-row fp32 images
    static constexpr int BIAS_OFF = IMG_OFF + 4 * FT_IMG_ROWS * FT_IMG_LD * 4;   // b' (C), [cz | cr] (2C), ch (C): fp32, copied once per workgroup
    static constexpr int NEXT_OFF = BIAS_OFF + 4 * C * 4;        // the workgroup's next tile (drawn from the tile counter by thread 0)
    static constexpr int BYTES = NEXT_OFF + 16;
};

// Persistent: a workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the grid is two workgroups per CU) and requests
// the x / L~ x rows, the region ids and the first weight fragments of its NEXT tile before the last epilogue of the current one:
// a freshly dispatched workgroup waited ~9.5 k cycles (of a tile's 70 k) for those first loads -- HBM latency under load, which
// nothing else of that workgroup could hide.  The biases live in LDS (4 KB, copied once): as global loads issued at the end of a
// K loop they sat in front of every epilogue's first round (one L2 round trip each, eight per tile).
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 32)     // timing-only developer build: no activation stores
#define FT_STORE16(v, ...) asm volatile("" :: "v"(v))
#else
#define FT_STORE16(...) __builtin_amdgcn_raw_buffer_store_b128(__VA_ARGS__)
#endif
template <int C, int F>
__global__ __launch_bounds__(256, 2) void fused_fwd_kernel(FusedFwdArgs a) {
    static_assert(C % 128 == 0 && F % 16 == 0, "tile shapes");
    using L = FusedFwdLds<C, F>;
    constexpr int KBC = C / 16, KBF = F / 16, NT = C / 128;
    extern __shared__ __attribute__((aligned(16))) char flds[];
    char* Hp = flds;
    char* Qp = flds + L::OPER_B;
    float* biasl = reinterpret_cast<float*>(flds + L::BIAS_OFF);
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: fragment addresses go into scalar offsets
    const unsigned uT = (unsigned)a.T;                           // N * T < 2^31 (host-checked): 32-bit row arithmetic
    const long tiles = (a.M + FT_ROWS - 1) / FT_ROWS;
    // developer trace (REGT_FUSED_TRACE=1, tools/fused_trace.py): shader-clock stamps of thread 0 at the phase boundaries
    // (-DREGT_FUSED_FINE, a developer build: FT_FINE stamps inside the phases as well -- slots 8 .. 31 of the tile's record)
#define FT_MARK(i) do { if (a.trace && tid == 0) a.trace[(long)FT_TRACE_SLOTS * tile + (i)] = (long)__builtin_amdgcn_s_memtime(); } while (0)
#ifdef REGT_FUSED_FINE
#define FT_FINE(i) FT_MARK(i)
#else
#define FT_FINE(i) do {} while (0)
#endif
    // (timing-only switches, REGT_FUSED_DBG: bit 0 = zero-record store descriptors: every activation store is dropped by the range
    // check; bit 1 = zero-record weight descriptors: fragment loads return 0 without traffic -- cdna_hip_programming.md section 7)
    const int st_on = (a.dbg & 1) ? 0 : 1, w_on = (a.dbg & 2) ? 0 : 1;
    // A-fragment offsets of the K = F operands: lane (lr, lh) holds k = 8 lh .. 8 lh + 7 of row 32 mi + lr of a 16-k block
    const int afo = lr * F * 2 + lh * 16;

    // ---- what a tile needs first, requested one tile ahead: x and L~ x rows as A fragments (L~ x unmasked: rows of another region
    //      than the tile's first are zeroed in registers) and the region of each lane's two fragment rows -- the HBM part; the
    //      weight fragments of phase 0 (L2) are requested at the top of the tile.  Vector-memory operations retire in issue order (stores included -- gfx9 has one counter), so a load
    //      issued BEHIND an epilogue's stores only returns once those stores are acknowledged; hence the rule of this kernel: what
    //      the NEXT K loop (or tile) needs is requested before the current epilogue stores.
    bf16x8 xf[2][KBF], lf[2][KBF];
    int rg_a, rg_b, rg0;                                         // region of rows lr / 32 + lr (-1: past the end) / of the tile's first row
    auto request_tile = [&](long tile, int rg_first) {
        const long m0 = tile * FT_ROWS;
        const int nvalid = (int)(a.M - m0 < FT_ROWS ? a.M - m0 : FT_ROWS);
        const unsigned mrow0 = (unsigned)m0;
        const __amdgpu_buffer_rsrc_t sX = f_rsrc(reinterpret_cast<const char*>(a.X) + m0 * F * 2, (long)nvalid * F * 2);
        const __amdgpu_buffer_rsrc_t sLX = f_rsrc(reinterpret_cast<const char*>(a.LX) + m0 * F * 2, (long)nvalid * F * 2);
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) {
            xf[0][kb] = f_ldfrag(sX, afo, kb * 32);
            xf[1][kb] = f_ldfrag(sX, afo + 32 * F * 2, kb * 32);
        }
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) {
            lf[0][kb] = f_ldfrag(sLX, afo, kb * 32);
            lf[1][kb] = f_ldfrag(sLX, afo + 32 * F * 2, kb * 32);
        }
        // every lane looks up the region of ITS two fragment rows (32 mi + lr); the waves vote on "more than one region in this
        // tile" (rare: a tile that straddles a region boundary) where the rows of other regions are masked
        auto region_of = [&](int r) { return r < nvalid ? (a.node_region ? a.node_region[(mrow0 + (unsigned)r) / uT] : 0) : -1; };
        rg_a = region_of(lr);
        rg_b = region_of(32 + lr);
        rg0 = rg_first;
    };
    // (the region of a tile's first row through the scalar cache, one tile ahead of the fragment requests that depend on it)
    auto first_region = [&](long tile) { return a.node_region ? a.node_region[(unsigned)(tile * FT_ROWS) / uT] : 0; };
    long tile = blockIdx.x;
    request_tile(tile, first_region(tile));
    for (int i = tid; i < 4 * C; i += 256) biasl[i] = i < C ? a.bprime[i] : (i < 3 * C ? a.czr[i - C] : a.ch[i - 3 * C]);

    // Epilogue geometry.  Every wave transposes its OWN 64 x 32 accumulator strip through a wave-private 16-row image, so an
    // epilogue needs no workgroup barrier at all (LDS operations of one wave execute in order): round rnd = rows 16 rnd ..
    // 16 rnd + 15 of the tile; lane = (row lane >> 2, columns 8 (lane & 3) .. + 7 of the wave's 32) = 16 bytes of a bf16 array.
    // The same lane owns the same (row, columns) in the Z and in the candidate epilogue, which is what lets Z stay in registers.
    float* imgw = reinterpret_cast<float*>(flds + L::IMG_OFF) + w * (FT_IMG_ROWS * FT_IMG_LD);
    const int er0 = lane >> 2, ec0 = 32 * w + 8 * (lane & 3);
    int er = er0, ec = ec0;
    // image addresses (floats; swizzle: ft_par above): an accumulator register of lane (lr, lh) is row (q & 3) + 8 (q >> 2) + 4 lh of
    // the round, whose parity is (q & 1) ^ lh; the lane's 8 epilogue columns are chunks 2 (lane & 3), 2 (lane & 3) + 1 of row er
    const int st_even = 4 * lh * FT_IMG_LD + (lr ^ (4 * lh)), st_odd = 4 * lh * FT_IMG_LD + (lr ^ (4 * (lh ^ 1)));
    const int e_lo = er * FT_IMG_LD + 4 * ((2 * (lane & 3)) ^ ft_par(er)), e_hi = er * FT_IMG_LD + 4 * ((2 * (lane & 3) + 1) ^ ft_par(er));
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 4)      // timing-only developer build: no transposition through LDS
    V8 abl_v;
    auto stage = [&](const f32x16 (&acc)[2], int rnd) {
        const int mi = rnd >> 1, rd = rnd & 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) abl_v.v[q] = mi ? acc[1][8 * rd + q] : acc[0][8 * rd + q];
    };
    auto img8 = [&]() { return abl_v; };
#else
    auto stage = [&](const f32x16 (&acc)[2], int rnd) {          // accumulators of round rnd -> the wave's image
        const int mi = rnd >> 1, rd = rnd & 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int reg = 4 * (2 * rd + (q >> 2)) + (q & 3);
            imgw[((q & 3) + 8 * (q >> 2)) * FT_IMG_LD + ((q & 1) ? st_odd : st_even)] = mi ? acc[1][reg] : acc[0][reg];
        }
    };
    auto img8 = [&]() {
        const float4 lo = *reinterpret_cast<const float4*>(imgw + e_lo), hi = *reinterpret_cast<const float4*>(imgw + e_hi);
        return V8{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
#endif
    auto bias8 = [&](int i) {                                    // (i: offset into [b' | cz | cr | ch])
        const float4 lo = *reinterpret_cast<const float4*>(biasl + i), hi = *reinterpret_cast<const float4*>(biasl + i + 4);
        return V8{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
    // the 16 bytes of (row, columns c .. c + 7) inside an operand's planes
    auto plane_off = [&](int row, int c) { return (c >> 4) * L::PLANE_B + sp_off(row, (c >> 3) & 1); };
    __syncthreads();                                            // the biases are in LDS

#pragma unroll 1
    while (tile < tiles) {
    // (the next tile is drawn from a counter in the workspace: see fused_bwd_kernel)
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(flds + L::NEXT_OFF) = a.tile_ctr ? atomicAdd(a.tile_ctr, 1u) + gridDim.x : (unsigned)(tile + gridDim.x);
    // (opaque per tile: hoisted out of the tile loop, the ~30 global / plane offsets derived from them stay live through every K loop)
    asm volatile("" : "+v"(er), "+v"(ec));
    const long m0 = tile * FT_ROWS;
    const int nvalid = (int)(a.M - m0 < FT_ROWS ? a.M - m0 : FT_ROWS);
    const unsigned mrow0 = (unsigned)m0;
    const unsigned node0 = mrow0 / uT;
    const int t0 = (int)(mrow0 - node0 * uT);                    // period of the tile's first row (wave-uniform)
    // Node boundaries of the tile as row masks (bit r = row r; all scalar): a.pmask has the bits k T < 64 (the host's division), so
    // the rows that START a node are pmask shifted to the first such row; a row ENDS a node (inside this tile) when the next one
    // starts one or it is the tile's last row; the ends whose sum is only a part of the node's go to memory as atomic adds: the
    // first one when the tile starts inside a node, the last one when the tile ends inside one.
    const unsigned long long vmask = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
    const int s0 = t0 == 0 ? 0 : a.T - t0;
    const unsigned long long smask = s0 < 64 ? (a.pmask << s0) & vmask : 0ull;
    const unsigned long long emask = ((smask >> 1) | (1ull << (nvalid - 1))) & vmask;
    const unsigned long long amask = (t0 != 0 ? emask & (0ull - emask) : 0ull) | (((unsigned)(t0 + nvalid) % uT) != 0 ? 1ull << (nvalid - 1) : 0ull);
    FT_MARK(0);

    // ---- descriptors: the tile's rows of every activation array (rows past the end are out of range: loads return 0, stores are dropped)
    const __amdgpu_buffer_rsrc_t sLX = f_rsrc(reinterpret_cast<const char*>(a.LX) + m0 * F * 2, (long)nvalid * F * 2);
    const __amdgpu_buffer_rsrc_t sAX = f_rsrc(reinterpret_cast<const char*>(a.AX) + m0 * F * 2, (long)nvalid * F * 2);
    const __amdgpu_buffer_rsrc_t sh = f_rsrc(reinterpret_cast<char*>(a.h) + m0 * C * 2, (long)nvalid * C * 2 * st_on);
    const __amdgpu_buffer_rsrc_t sq = f_rsrc(reinterpret_cast<char*>(a.q) + m0 * C * 2, (long)nvalid * C * 2 * st_on);
    const __amdgpu_buffer_rsrc_t sHt = f_rsrc(reinterpret_cast<char*>(a.Ht) + m0 * C * 2, (long)nvalid * C * 2 * st_on);
    const __amdgpu_buffer_rsrc_t sZR = f_rsrc(reinterpret_cast<char*>(a.ZR) + m0 * C * 4, (long)nvalid * C * 4 * st_on);
    bf16x8 axf[2][KBF], b0[2][KBF], b1[2][KBF];                  // A_hat x fragments; A0 ([0]) / A_region ([1]) fragments of column tile 0 / 1
    const int rgt = rg0, rga = rg_a, rgb = rg_b;                 // (this tile's; the loop-carried set is refilled further down)
    {
        const __amdgpu_buffer_rsrc_t sA0 = f_rsrc(a.A0f, (long)C * F * 2);
        const __amdgpu_buffer_rsrc_t sAr = f_rsrc(reinterpret_cast<const char*>(a.Aallf) + (long)rgt * a.ar_stride, (long)C * F * 2);
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) b0[0][kb] = f_ldw(sA0, lane * 16, (w * KBF + kb) * 1024);
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) b0[1][kb] = f_ldw(sAr, lane * 16, (w * KBF + kb) * 1024);
    }
    float ptr_[4];                                               // attention probability of the lane's epilogue row of each round
#pragma unroll
    for (int rnd = 0; rnd < 4; ++rnd) {
        const unsigned m = mrow0 + 16 * rnd + (lane >> 2);
        ptr_[rnd] = a.probs[m - (m / uT) * uT];
    }
    const bool multi = __ballot((rga >= 0 && rga != rgt) || (rgb >= 0 && rgb != rgt)) != 0;
    if (multi) {                                                 // rare: rows of the tile's other regions contribute to THEIR region's pass
        const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) {
            lf[0][kb] = rga == rgt ? lf[0][kb] : zero;
            lf[1][kb] = rgb == rgt ? lf[1][kb] : zero;
        }
    }
    FT_MARK(1);

    // B fragments of one K = C + F loop: W (C x C, column block nb) and G (column block nbg of a (rows x F) composed weight)
    bf16x8 bw[KBC], bg[KBF];
    auto issue_b = [&](const void* Wf, int nb, const void* Gf, int nbg) {
        const __amdgpu_buffer_rsrc_t sW = f_rsrc(Wf, (long)C * C * 2 * w_on), sG = f_rsrc(Gf, (long)0x7ffffff0 * w_on);
#pragma unroll
        for (int kb = 0; kb < KBC; ++kb) bw[kb] = f_ldw(sW, lane * 16, (nb * KBC + kb) * 1024);
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) bg[kb] = f_ldw(sG, lane * 16, (nbg * KBF + kb) * 1024);
    };
    // acc = P (planes, K = C) x W^T + (A_hat x) x G^T with the fragments requested by the last issue_b
    auto kloop = [&](f32x16 (&acc)[2], const char* P) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // A fragments from the planes, three 16-k blocks ahead of the MFMAs that consume them (an LDS round trip is ~2-3 MFMA
        // pairs long and this wave has one partner on its SIMD); the scheduling barriers keep the reads where they are written
        // (left alone, the scheduler sinks every read to its use)
        const char* pa0 = P + sp_off(lr, lh);
        const char* pa1 = P + sp_off(32 + lr, lh);
        constexpr int AHEAD = 3;
        bf16x8 fa[KBC + AHEAD][2];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < AHEAD; ++kb) {
            fa[kb][0] = *reinterpret_cast<const bf16x8*>(pa0 + kb * L::PLANE_B);
            fa[kb][1] = *reinterpret_cast<const bf16x8*>(pa1 + kb * L::PLANE_B);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < KBC; ++kb) {
            if (kb + AHEAD < KBC) {
                fa[kb + AHEAD][0] = *reinterpret_cast<const bf16x8*>(pa0 + (kb + AHEAD) * L::PLANE_B);
                fa[kb + AHEAD][1] = *reinterpret_cast<const bf16x8*>(pa1 + (kb + AHEAD) * L::PLANE_B);
            }
            __builtin_amdgcn_sched_barrier(0);
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 1)      // timing-only developer build: no matrix instructions (operands kept alive)
            asm volatile("" :: "v"(fa[kb][0]), "v"(fa[kb][1]), "v"(bw[kb]));
#else
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][0], bw[kb], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][1], bw[kb], acc[1], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) {
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 1)
            asm volatile("" :: "v"(axf[0][kb]), "v"(axf[1][kb]), "v"(bg[kb]));
#else
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(axf[0][kb], bg[kb], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(axf[1][kb], bg[kb], acc[1], 0, 0, 0);
#endif
        }
    };

    // ---- phase 0: regional embedding h = act(x A0^T + (L~ x) A_region^T + b') -> global + h planes --------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        static_assert(NT == 2, "phase 0 double-buffers the fragments of two column tiles");
        if (j == 0) {           // the fragments of column tile 1 are requested before tile 0's epilogue stores
            const __amdgpu_buffer_rsrc_t sA0 = f_rsrc(a.A0f, (long)C * F * 2);
            const __amdgpu_buffer_rsrc_t sAr = f_rsrc(reinterpret_cast<const char*>(a.Aallf) + (long)rgt * a.ar_stride, (long)C * F * 2);
#pragma unroll
            for (int kb = 0; kb < KBF; ++kb) {
                b1[0][kb] = f_ldw(sA0, lane * 16, ((4 + w) * KBF + kb) * 1024);
                b1[1][kb] = f_ldw(sAr, lane * 16, ((4 + w) * KBF + kb) * 1024);
            }
        }
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[0][kb], j ? b1[0][kb] : b0[0][kb], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[1][kb], j ? b1[0][kb] : b0[0][kb], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int kb = 0; kb < KBF; ++kb) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf[0][kb], j ? b1[1][kb] : b0[1][kb], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf[1][kb], j ? b1[1][kb] : b0[1][kb], acc[1], 0, 0, 0);
        }
        // further regions of the tile (rare), in order of first appearance by row: rows of other regions contribute zeros
        bool done_a = rga < 0 || rga == rgt, done_b = rgb < 0 || rgb == rgt;
#pragma unroll 1
        while (multi) {
            const unsigned long long ma = __ballot(!done_a), mb = __ballot(!done_b);
            if (!(ma | mb)) break;
            const int rg = ma ? __builtin_amdgcn_readlane(rga, __builtin_ctzll(ma)) : __builtin_amdgcn_readlane(rgb, __builtin_ctzll(mb));
            done_a = done_a || rga == rg;
            done_b = done_b || rgb == rg;
            const __amdgpu_buffer_rsrc_t sAr = f_rsrc(reinterpret_cast<const char*>(a.Aallf) + (long)rg * a.ar_stride, (long)C * F * 2);
            const int o0 = rga == rg ? afo : 0x7ffffff0, o1 = rgb == rg ? afo + 32 * F * 2 : 0x7ffffff0;
            bf16x8 b[KBF], x[2][KBF];
#pragma unroll
            for (int kb = 0; kb < KBF; ++kb) {
                b[kb] = f_ldfrag(sAr, lane * 16, ((4 * j + w) * KBF + kb) * 1024);
                x[0][kb] = f_ldfrag(sLX, o0, kb * 32);
                x[1][kb] = f_ldfrag(sLX, o1, kb * 32);
            }
#pragma unroll
            for (int kb = 0; kb < KBF; ++kb) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0][kb], b[kb], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1][kb], b[kb], acc[1], 0, 0, 0);
            }
        }
        if (j == NT - 1) {
            issue_b(a.Urf, w, a.Gzrf, C / 32 + w);                          // reset gate, column tile 0
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)                                  // A_hat x: A fragments of all six K loops to come
#pragma unroll
                for (int kb = 0; kb < KBF; ++kb) axf[mi][kb] = f_ldfrag(sAX, afo + mi * 32 * F * 2, kb * 32);
        }
        FT_FINE(8 + 2 * j);
        // (every wave is done with the planes of the workgroup's previous tile: its Z / candidate K loops read them)
        if (j == 0) __syncthreads();
        const V8 b = bias8(128 * j + ec);
        const float ns = a.act_lrelu ? a.slope : 1.0f;
        // (tried in round 4 and measured no faster: staging and reading back round rnd + 1 before round rnd is consumed -- the LDS
        // round trip under the arithmetic of the previous round -- 1.75 vs 1.70-1.72 ms; packed two-lane arithmetic in the gate
        // epilogues, f_sigmoid8 / f_tanh8: 9 % fewer vector instructions, same time)
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            stage(acc, rnd);
            const V8 v = img8();
            V8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float s = v.v[i] + b.v[i]; o.v[i] = s > 0.f ? s : s * ns; }
            const u32x4_t pk = f_pack8(o);
            const int row = 16 * rnd + er, c = 128 * j + ec;
            FT_STORE16(pk, sh, (row * C + c) * 2, 0, 0);
            *reinterpret_cast<u32x4_t*>(Hp + plane_off(row, c)) = pk;
        }
        FT_FINE(9 + 2 * j);
    }
    FT_MARK(2);
    __syncthreads();                                            // h planes complete
    // the tile after this one (clamped: the last round requests the last tile once more instead of branching around the loads)
    const long tdrawn = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(flds + L::NEXT_OFF));
    const long tnext = tdrawn < tiles ? tdrawn : tiles - 1;
    const int rg_next = first_region(tnext);
    FT_FINE(12);

    // ---- phase 1: reset gate R = sigmoid(h Ur^T + (A_hat x) Gr^T + cr), q = h R -> global + q planes ---------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x16 acc[2];
        kloop(acc, Hp);
        if (j + 1 < NT) issue_b(a.Urf, 4 * (j + 1) + w, a.Gzrf, C / 32 + 4 * (j + 1) + w);
        else issue_b(a.Uzf, w, a.Gzrf, w);                                  // update gate, column tile 0
        const V8 b = bias8(2 * C + 128 * j + ec);
        FT_FINE(13 + 5 * j);
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            if (rnd) FT_FINE(13 + 5 * j + rnd);
            stage(acc, rnd);
            const V8 v = img8();
            const int row = 16 * rnd + er, c = 128 * j + ec;
            const V8 hv = f_widen8(*reinterpret_cast<const u32x4_t*>(Hp + plane_off(row, c)));
            const V8 g = f_sigmoid8(v, b);
            V8 qv;
#pragma unroll
            for (int i = 0; i < 8; ++i) qv.v[i] = hv.v[i] * g.v[i];
            FT_STORE16(f_pack8(g), sZR, (row * 2 * C + C + c) * 2, 0, 0);
            const u32x4_t pq = f_pack8(qv);
            FT_STORE16(pq, sq, (row * C + c) * 2, 0, 0);
            *reinterpret_cast<u32x4_t*>(Qp + plane_off(row, c)) = pq;
        }
        FT_FINE(17 + 5 * j);
    }
    FT_MARK(3);
    __syncthreads();                                            // q planes complete
    FT_FINE(23);

    // ---- phases 2 + 3 per column tile: update gate Z (kept in registers), candidate H~, blend, sum over the node's periods ---
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        u32x4_t zk[4];
        {
            f32x16 acc[2];
            kloop(acc, Hp);
            if (j == 0) FT_FINE(24);
            issue_b(a.Uhf, 4 * j + w, a.Ghf, 4 * j + w);                      // candidate, same column tile
            const V8 b = bias8(C + 128 * j + ec);
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) {
                stage(acc, rnd);
                const V8 v = img8();
                const V8 g = f_sigmoid8(v, b);
                zk[rnd] = f_pack8(g);
                FT_STORE16(zk[rnd], sZR, ((16 * rnd + er) * 2 * C + 128 * j + ec) * 2, 0, 0);
            }
        }
        FT_MARK(4 + 2 * j);
        f32x16 acc[2];
        kloop(acc, Qp);
        if (j + 1 < NT) issue_b(a.Uzf, 4 * (j + 1) + w, a.Gzrf, 4 * (j + 1) + w);
        else request_tile(tnext, rg_next);                                  // the next tile's first operands
        const V8 b = bias8(3 * C + 128 * j + ec);
        if (j == 0) FT_FINE(25);
        // Per-node sums of the blended rows (the attention-weighted sum over the periods).  Every lane keeps ONE running column sum
        // (column 32 w + lr of the tile; lanes 32..63 duplicate lanes 0..31 and store nothing) over the tile's rows in row order;
        // node boundaries are wave-uniform bit masks of the tile's rows (smask / emask / amask above): a row that starts a node
        // multiplies the carried sum by 0 (fma(c, 0, v) = v, fma(c, 1, v) = c + v: the sums are exactly those of a plain `c += v`
        // chain per node), a row that ends one hands the sum over.
        float csum = 0.f;
        int ohs = 0;                     // byte offset of the current node's row of OH behind the tile's first node (scalar)
        const long ohcol = (long)node0 * C + 128 * j + 32 * w;
        const __amdgpu_buffer_rsrc_t sOH = f_rsrc(a.OH + ohcol, (a.nodes * C - ohcol) * 4);
        const int ohv = lane < 32 ? lr * 4 : 0x7ffffff0;       // (out of range: dropped)
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            if (j == 0 && rnd) FT_FINE(25 + rnd);
            stage(acc, rnd);
            const V8 v = img8();
            const int row = 16 * rnd + er, c = 128 * j + ec;
            const V8 hv = f_widen8(*reinterpret_cast<const u32x4_t*>(Hp + plane_off(row, c)));
            const V8 Zv = f_widen8(zk[rnd]);
            const float pt = ptr_[rnd];
            const V8 ht = f_tanh8(v, b);
            V8 bl;
#pragma unroll
            for (int i = 0; i < 8; ++i) bl.v[i] = __fmul_rn(pt, gru_blend(Zv.v[i], hv.v[i], ht.v[i]));
            FT_STORE16(f_pack8(ht), sHt, (row * C + c) * 2, 0, 0);
            *reinterpret_cast<float4*>(imgw + e_lo) = make_float4(bl.v[0], bl.v[1], bl.v[2], bl.v[3]);
            *reinterpret_cast<float4*>(imgw + e_hi) = make_float4(bl.v[4], bl.v[5], bl.v[6], bl.v[7]);
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 8)      // timing-only developer build: no per-node sums
            continue;
#endif
            float cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) cv[r] = imgw[r * FT_IMG_LD + (ft_par(r) ? lr ^ 4 : lr)];
            // (the masks are made opaque per round: left alone, hipcc shares the 128 bit tests of the two column tiles through
            // spilled scalar registers; the 0 / 1 factor is selected on the scalar unit: as a float select it becomes a v_cndmask)
            unsigned sm = (unsigned)(smask >> (16 * rnd)), em = (unsigned)(emask >> (16 * rnd)), am = (unsigned)(amask >> (16 * rnd));
            asm volatile("" : "+s"(sm), "+s"(em), "+s"(am));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float keep;
                asm volatile("s_bitcmp1_b32 %1, %2\n\ts_cselect_b32 %0, 0, 1.0" : "=s"(keep) : "s"(sm), "n"(r) : "scc");
                csum = fmaf(csum, keep, cv[r]);
                if ((em >> r) & 1u) {                                   // (wave-uniform, one row in T)
                    // a node whose rows all lie in this tile has no rows elsewhere: a plain store into the zeroed array; only the
                    // tile's first / last node can continue in a neighbouring tile (two addends at most: still bit-reproducible).
                    // Both instructions are issued, the one that does not apply with an out-of-range offset (no branch).
                    const bool part = (am >> r) & 1u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(csum), sOH, ohv, part ? 0x7ffffff0 : ohs, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(csum, sOH, ohv, part ? ohs : 0x7ffffff0, 0);
                    ohs += C * 4;
                }
            }
        }
        FT_MARK(5 + 2 * j);
    }
    tile = tdrawn;
    }
#undef FT_MARK
#undef FT_FINE
}

int fused_cus() {
    static int cus = 0;
    if (cus <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}

static long* g_fused_trace = nullptr;
static long g_fused_trace_n = 0;
// REGT_FUSED_TRACE = 1: stamps of the forward kernel, 2: of the backward kernel (8 per tile); nullptr when tracing is off
long* fused_trace_buffer(int which, long tiles) {
    static int tr = -1;
    if (tr < 0) { const char* e = getenv("REGT_FUSED_TRACE"); tr = e ? atoi(e) : 0; }
    if (tr != which) return nullptr;
    if (!g_fused_trace || g_fused_trace_n < FT_TRACE_SLOTS * tiles) {
        if (g_fused_trace) (void)hipFree(g_fused_trace);
        g_fused_trace = nullptr;
        if (hipMalloc(&g_fused_trace, FT_TRACE_SLOTS * tiles * sizeof(long)) != hipSuccess) return nullptr;
        g_fused_trace_n = FT_TRACE_SLOTS * tiles;
    }
    return g_fused_trace;
}

// ---- fused data gradients of the cell ---------------------------------------------------------------------------------------------
// cell_bwd + dgrad_candidate + dgrad_gates of the three-launch backward in one kernel per 64-row tile.  With g = p_t dOH[node]:
//     dhp = g (1 - Z) (1 - H~^2),   dzp = g (h - H~) Z (1 - Z)                       (gate pre-activation gradients: cell.hip)
//     dq  = dhp Uh2,   drp = dq h R (1 - R),   dh = dq R + g Z                        (back through linear_h's h-half and q = h R)
//     ds  = (dh + dzp Uz2 + drp Ur2) act'(h)                                          (back through linear_z / _r, into the embedding)
// i.e. the transposes of models/utils.py:168-188 in the composed form of DESIGN.md section 3.  The three launches read Z (twice),
// R, h (three times), H~, dhp, [dzp|drp] and dh back from HBM: 11 activation reads + 5 writes of M x C bf16 per step; here Z, R, h,
// H~ are read and dhp, dzp, drp, ds written once (Z and dOH are read a second time in the dq epilogue, dzp comes back once; h waits
// in LDS).
// Same tile shape, fragment loads, plane layout and wave-private epilogue images as the forward kernel above: dhp is the A operand of
// the dq product (planes P), drp fills planes R from the dq epilogues, dzp is fetched back (L2) while drp Ur2 runs and takes dhp's
// place in P; ds accumulates drp Ur2 + dzp Uz2 in one K = 2C loop (the order of the three-launch kernel), dh makes the same round
// trip through its bf16 array as in the three-launch path (same lane writes and reads it: L2).  Bit-identical to the three launches except for the summation order of
// the attention-probability gradient (per-row dots -> rowdot, summed in a fixed order by rowdot_reduce_kernel).
template <int C>
struct FusedBwdLds {
    static constexpr int PLANE_B = FT_ROWS * 32;
    static constexpr int OPER_B = (C / 16) * PLANE_B;
    static constexpr int IMG_OFF = 2 * OPER_B;
    static constexpr int DOT_OFF = IMG_OFF + 4 * FT_IMG_ROWS * FT_IMG_LD * 4;
    static constexpr int NEXT_OFF = DOT_OFF + 4 * FT_ROWS * 4;   // the workgroup's next tile (drawn from the tile counter by thread 0)
    static constexpr int BYTES = NEXT_OFF + 16;
};

template <int C>
__global__ __launch_bounds__(256, 2) void fused_bwd_kernel(FusedBwdArgs a) {
    static_assert(C % 128 == 0, "tile shapes");
    using L = FusedBwdLds<C>;
    constexpr int KBC = C / 16, NT = C / 128;
    extern __shared__ __attribute__((aligned(16))) char flds[];
    char* Pp = flds;                         // dhp, later dzp
    char* Rp = flds + L::OPER_B;             // drp
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned uT = (unsigned)a.T;
    const long nodes = a.M / a.T;
    const long tiles = (a.M + FT_ROWS - 1) / FT_ROWS;
#define FB_MARK(i) do { if (a.trace && tid == 0) a.trace[(long)FT_TRACE_SLOTS * tile + (i)] = (long)__builtin_amdgcn_s_memtime(); } while (0)
    float* imgw = reinterpret_cast<float*>(flds + L::IMG_OFF) + w * (FT_IMG_ROWS * FT_IMG_LD);
    float* dotw = reinterpret_cast<float*>(flds + L::DOT_OFF);
    const int er00 = lane >> 2, ec00 = 32 * w + 8 * (lane & 3); // the lane's epilogue row of a round / first of its 8 columns of a tile
    int er = er00, ec = ec00;
    // image addresses (floats; swizzle: ft_par, fused_fwd_kernel)
    const int st_even = 4 * lh * FT_IMG_LD + (lr ^ (4 * lh)), st_odd = 4 * lh * FT_IMG_LD + (lr ^ (4 * (lh ^ 1)));
    const int e_lo = er00 * FT_IMG_LD + 4 * ((2 * (lane & 3)) ^ ft_par(er00)), e_hi = er00 * FT_IMG_LD + 4 * ((2 * (lane & 3) + 1) ^ ft_par(er00));
    auto stage = [&](const f32x16 (&acc)[2], int rnd) {
        const int mi = rnd >> 1, rd = rnd & 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int reg = 4 * (2 * rd + (q >> 2)) + (q & 3);
            imgw[((q & 3) + 8 * (q >> 2)) * FT_IMG_LD + ((q & 1) ? st_odd : st_even)] = mi ? acc[1][reg] : acc[0][reg];
        }
    };
    auto img8 = [&]() {
        const float4 lo = *reinterpret_cast<const float4*>(imgw + e_lo), hi = *reinterpret_cast<const float4*>(imgw + e_hi);
        return V8{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
    // Persistent (two workgroups per CU draw tiles from a counter): Z, h, H~ of the NEXT tile's first 128 columns are requested before
    // the last epilogue of the current one.  A freshly dispatched workgroup spent a third of a tile
    // (22 k of 68 k cycles) waiting for those first loads -- HBM latency that nothing else of the workgroup could hide.
    u32x4_t zr0[4], hr0[4], tr0[4];
    auto request_tile0 = [&](long tile) {
        const long m0 = tile * FT_ROWS;
        const int nvalid = (int)(a.M - m0 < FT_ROWS ? a.M - m0 : FT_ROWS);
        const __amdgpu_buffer_rsrc_t sZR = f_rsrc(reinterpret_cast<const char*>(a.ZR) + m0 * C * 4, (long)nvalid * C * 4);
        const __amdgpu_buffer_rsrc_t sH = f_rsrc(reinterpret_cast<const char*>(a.h) + m0 * C * 2, (long)nvalid * C * 2);
        const __amdgpu_buffer_rsrc_t sHt = f_rsrc(reinterpret_cast<const char*>(a.Ht) + m0 * C * 2, (long)nvalid * C * 2);
        // (offsets recomputed from an opaque copy of the lane's row / column: hoisted out of the tile loop they are spilled, and every
        // reload waits for vmcnt(0) -- in the middle of this very burst of requests)
        int er0 = er00, ec0 = ec00;
        asm volatile("" : "+v"(er0), "+v"(ec0));
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            const int row = 16 * rnd + er0, c = ec0;
            zr0[rnd] = __builtin_amdgcn_raw_buffer_load_b128(sZR, (row * 2 * C + c) * 2, 0, 0);
            hr0[rnd] = __builtin_amdgcn_raw_buffer_load_b128(sH, (row * C + c) * 2, 0, 0);
            tr0[rnd] = __builtin_amdgcn_raw_buffer_load_b128(sHt, (row * C + c) * 2, 0, 0);
        }
    };
    long tile = blockIdx.x;
    request_tile0(tile);
#pragma unroll 1
    while (tile < tiles) {
    // (opaque per tile: hoisted out of the tile loop, the global / plane offsets derived from them stay live through every K loop)
    asm volatile("" : "+v"(er), "+v"(ec));
    const long m0 = tile * FT_ROWS;
    const int nvalid = (int)(a.M - m0 < FT_ROWS ? a.M - m0 : FT_ROWS);
    const unsigned mrow0 = (unsigned)m0;
    const unsigned node0 = mrow0 / uT;
    // The next tile comes from a counter in the workspace (zeroed by the launcher), not from blockIdx + k gridDim: when a kernel of
    // the side stream holds some CUs the grid is not fully resident, and workgroups that start late would find a whole stride of
    // tiles waiting for them (measured: 1.65 -> 2.4 ms with the head's weight gradients running beside this kernel).
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(flds + L::NEXT_OFF) = a.tile_ctr ? atomicAdd(a.tile_ctr, 1u) + gridDim.x : (unsigned)(tile + gridDim.x);
    FB_MARK(0);
    // the tile's rows of every activation array; dOH from the tile's first node on (rows past the end read zeros, stores are dropped)
    const __amdgpu_buffer_rsrc_t sZR = f_rsrc(reinterpret_cast<const char*>(a.ZR) + m0 * C * 4, (long)nvalid * C * 4);
    const __amdgpu_buffer_rsrc_t sH = f_rsrc(reinterpret_cast<const char*>(a.h) + m0 * C * 2, (long)nvalid * C * 2);
    const __amdgpu_buffer_rsrc_t sHt = f_rsrc(reinterpret_cast<const char*>(a.Ht) + m0 * C * 2, (long)nvalid * C * 2);
    const __amdgpu_buffer_rsrc_t sD = f_rsrc(a.dOH + (long)node0 * C, (nodes - (long)node0) * C * 4);
    const __amdgpu_buffer_rsrc_t sdhp = f_rsrc(reinterpret_cast<char*>(a.dhp) + m0 * C * 2, (long)nvalid * C * 2);
    const __amdgpu_buffer_rsrc_t sdzr = f_rsrc(reinterpret_cast<char*>(a.dzr) + m0 * C * 4, (long)nvalid * C * 4);
    const __amdgpu_buffer_rsrc_t sdh = f_rsrc(reinterpret_cast<char*>(a.dh) + m0 * C * 2, (long)nvalid * C * 2);
    float pt_[4];                                                // per round: attention probability of the lane's row,
    int dof_[4];                                                 //   byte offset of its node's dOH row behind node0's
#pragma unroll
    for (int rnd = 0; rnd < 4; ++rnd) {
        const unsigned m = mrow0 + 16 * rnd + er, nd = m / uT;
        pt_[rnd] = a.probs[m - nd * uT];
        dof_[rnd] = (int)(nd - node0) * C * 4;
    }
    __syncthreads();                                            // every wave is done with the planes and row dots of the previous tile
    auto plane_off = [&](int row, int c) { return (c >> 4) * L::PLANE_B + sp_off(row, (c >> 3) & 1); };
    auto ld16 = [&](__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    auto ldd8 = [&](int off) {
        const u32x4_t lo = __builtin_amdgcn_raw_buffer_load_b128(sD, off, 0, 0), hi = __builtin_amdgcn_raw_buffer_load_b128(sD, off + 16, 0, 0);
        return V8{{__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w),
                   __uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w)}};
    };
    // B fragments: column block nb of a transposed C x C weight block in fragment order, HALF a K loop (8 of 16 blocks) in registers at
    // a time -- a slot is refilled with the block eight further on (of this product, then of the next one) as soon as it has been
    // multiplied.  (All 16 in registers cost 32 registers more: with the next tile's operands in flight the kernel spilled, and a
    // spill reload waits for vmcnt(0), i.e. for the prefetch.)
    bf16x8 bw[KBC / 2];
    auto issue_b = [&](const void* Wf, int nb, int from, int to) {
        const __amdgpu_buffer_rsrc_t sW = f_rsrc(Wf, (long)C * C * 2);
#pragma unroll
        for (int kb = 0; kb < KBC; ++kb)
            if (kb >= from && kb < to) bw[kb % (KBC / 2)] = f_ldfrag(sW, lane * 16, (nb * KBC + kb) * 1024);
    };
    // acc (+)= P (planes, K = C) x Wcur^T; the first half of Wcur's fragments is in bw (requested by the previous K loop or by the
    // caller), the second half follows the MFMAs of the first, then the first half of (Wnext, nbnext) follows the second (nullptr: none)
    auto kloop = [&](f32x16 (&acc)[2], const char* P, bool zero, const void* Wcur, int nbcur, const void* Wnext, int nbnext) {
        if (zero) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        const char* pa0 = P + sp_off(lr, lh);
        const char* pa1 = P + sp_off(32 + lr, lh);
        constexpr int AHEAD = 2;
        bf16x8 fa[KBC + AHEAD][2];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < AHEAD; ++kb) {
            fa[kb][0] = *reinterpret_cast<const bf16x8*>(pa0 + kb * L::PLANE_B);
            fa[kb][1] = *reinterpret_cast<const bf16x8*>(pa1 + kb * L::PLANE_B);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < KBC; ++kb) {
            if (kb + AHEAD < KBC) {
                fa[kb + AHEAD][0] = *reinterpret_cast<const bf16x8*>(pa0 + (kb + AHEAD) * L::PLANE_B);
                fa[kb + AHEAD][1] = *reinterpret_cast<const bf16x8*>(pa1 + (kb + AHEAD) * L::PLANE_B);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][0], bw[kb % (KBC / 2)], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][1], bw[kb % (KBC / 2)], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kb < KBC / 2) issue_b(Wcur, nbcur, kb + KBC / 2, kb + KBC / 2 + 1);
            else if (Wnext) issue_b(Wnext, nbnext, kb - KBC / 2, kb - KBC / 2 + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- phase A: gate pre-activation gradients (element-wise): dhp -> global + planes P, dzp -> global ----------------------------
    float rdot[4] = {0.f, 0.f, 0.f, 0.f};
    u32x4_t zkeep[NT][4];                    // Z as read in phase A: the dq epilogues need it again (dh = dq R + g Z) -- 32 registers
                                             //   instead of a second read of the array (0.77 GB per step at the cfg-5 shard)
    {
        // every operand is requested before anything is stored (a load issued behind a store waits for that store): Z, h, H~ of both
        // column tiles up front, dOH of the second tile once the first tile's values are in registers, the stores after that
        u32x4_t zr_[NT][4], hr_[NT][4], tr_[NT][4];
        V8 dd_[NT][4];
        auto ld_zht = [&](int j) {
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) {
                const int row = 16 * rnd + er, c = 128 * j + ec;
                zr_[j][rnd] = ld16(sZR, (row * 2 * C + c) * 2);
                hr_[j][rnd] = ld16(sH, (row * C + c) * 2);
                tr_[j][rnd] = ld16(sHt, (row * C + c) * 2);
            }
        };
        auto ld_d = [&](int j) {
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) dd_[j][rnd] = ldd8(dof_[rnd] + (128 * j + ec) * 4);
        };
        // (the first column tile's Z, h, H~ came with the previous tile; everything else this phase reads is requested now, before its
        // first store -- a load issued behind a store waits for that store; the results are then stored round by round)
        ld_d(0);
        ld_zht(1);
        ld_d(1);
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) { zr_[0][rnd] = zr0[rnd]; hr_[0][rnd] = hr0[rnd]; tr_[0][rnd] = tr0[rnd]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) {
                const V8 z = f_widen8(zr_[j][rnd]), h = f_widen8(hr_[j][rnd]), ht = f_widen8(tr_[j][rnd]);
                zkeep[j][rnd] = zr_[j][rnd];
                V8 dhp, dzp;
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float g = __fmul_rn(pt_[rnd], dd_[j][rnd].v[i]);
                    dhp.v[i] = cb_dhp(g, z.v[i], ht.v[i]);
                    dzp.v[i] = cb_dzp(g, h.v[i], ht.v[i], z.v[i]);
                    dot += dd_[j][rnd].v[i] * (z.v[i] * h.v[i] + (1.0f - z.v[i]) * ht.v[i]);
                }
                rdot[rnd] += dot;
                asm volatile("" : "+v"(rdot[rnd]));              // the sum is formed here, not at its first use (keeps 1 - Z, h, H~ short-lived)
                const u32x4_t o_p = f_pack8(dhp), o_z = f_pack8(dzp);
                const int row = 16 * rnd + er, c = 128 * j + ec;
                *reinterpret_cast<u32x4_t*>(Pp + plane_off(row, c)) = o_p;
                // h waits for the dq epilogue in the (still empty) planes R, at the very place the same lane will put drp
                *reinterpret_cast<u32x4_t*>(Rp + plane_off(row, c)) = hr_[j][rnd];
                __builtin_amdgcn_raw_buffer_store_b128(o_p, sdhp, (row * C + c) * 2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o_z, sdzr, (row * 2 * C + c) * 2, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    FB_MARK(1);
    issue_b(a.UhTf, w, 0, KBC / 2);                              // dq, column tile 0: the first half of its fragments
#pragma unroll
    for (int rnd = 0; rnd < 4; ++rnd) {                          // the wave's 32 columns x 2 tiles of row 16 rnd + er
        float s = rdot[rnd];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if ((lane & 3) == 0) dotw[w * FT_ROWS + 16 * rnd + er] = s;
    }
    __syncthreads();                                            // planes P (dhp) and the four waves' row dots complete
    const long tdrawn = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(flds + L::NEXT_OFF));
    const long tnext = tdrawn < tiles ? tdrawn : tiles - 1;     // (clamped for the prefetch: no branch around its loads)
    if (tid < FT_ROWS) {                                        // (through a descriptor that ends with the tile's rows: no 64-bit address per lane)
        const __amdgpu_buffer_rsrc_t sdot = f_rsrc(a.rowdot + m0, (long)nvalid * 4);
        const float dsum = (dotw[tid] + dotw[FT_ROWS + tid]) + (dotw[2 * FT_ROWS + tid] + dotw[3 * FT_ROWS + tid]);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dsum), sdot, tid * 4, 0, 0);
    }

    // ---- phase B: dq = dhp Uh2 per column tile; drp -> global + planes R, dh -> global (bf16: read back by the same lane in phase D,
    //      which is exactly the rounding point of the three-launch path) ----------------------------------------------------------------
    static_assert(NT == 2, "h-sign masks of two column tiles");
    unsigned hpos0 = 0, hpos1 = 0;                               // bit 8 rnd + i: h > 0 (the leaky-relu derivative of phase D)
    u32x4_t dhk[NT][4] = {};                                     // dh, rounded to bf16 where the three-launch path stores it
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {                               // (rolled: unrolled, the scheduler's hoisting costs ~1 KB of spills)
        unsigned hp = 0;
        u32x4_t xr[2];                                           // operands of the epilogue rounds, requested one round ahead
        V8 xd[2];
        auto aux = [&](int rnd) {
            const int row = 16 * rnd + er, c = 128 * j + ec;
            xr[rnd & 1] = ld16(sZR, (row * 2 * C + C + c) * 2);  // R: first touch (HBM); Z waits in registers, h in planes R, dOH was read in phase A
            xd[rnd & 1] = ldd8(dof_[rnd] + c * 4);
        };
        aux(0);
        f32x16 acc[2];
        // (the next product's first fragments: dq of the next column tile, then ds of column tile 0)
        kloop(acc, Pp, true, a.UhTf, 4 * j + w, j + 1 < NT ? a.UhTf : a.UrTf, j + 1 < NT ? 4 * (j + 1) + w : w);
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            if (rnd + 1 < 4) aux(rnd + 1);
            stage(acc, rnd);
            const V8 v = img8();
            const int row = 16 * rnd + er, c = 128 * j + ec;
            const V8 h = f_widen8(*reinterpret_cast<const u32x4_t*>(Rp + plane_off(row, c)));
            // (j is a run-time value in this rolled loop: a select, not a dynamic register index)
            u32x4_t zsel;
#pragma unroll
            for (int i = 0; i < 4; ++i) zsel[i] = j == 0 ? zkeep[0][rnd][i] : zkeep[1][rnd][i];
            const V8 Z = f_widen8(zsel), R = f_widen8(xr[rnd & 1]);
            V8 drp, dh;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                drp.v[i] = cb_drp(v.v[i], h.v[i], R.v[i]);
                dh.v[i] = cb_dh(v.v[i], R.v[i], pt_[rnd], xd[rnd & 1].v[i], Z.v[i]);
                hp |= (h.v[i] > 0.f ? 1u : 0u) << (8 * rnd + i);
            }
            const u32x4_t pr = f_pack8(drp);
            *reinterpret_cast<u32x4_t*>(Rp + plane_off(row, c)) = pr;
            __builtin_amdgcn_raw_buffer_store_b128(pr, sdzr, (row * 2 * C + C + c) * 2, 0, 0);
            {
                u32x4_t pk = f_pack8(dh);
                asm volatile("" : "+v"(pk));                     // (pinned: hipcc otherwise sinks this arithmetic to its use in phase D)
                if (j == 0) dhk[0][rnd] = pk;
                else dhk[1][rnd] = pk;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 0) hpos0 = hp;
        else hpos1 = hp;
        FB_MARK(2 + j);
    }
    __syncthreads();                                            // planes R (drp) complete, every wave is done with dhp in planes P

    // ---- phase D: ds = (dh + drp Ur2 + dzp Uz2) act'(h) per column tile -> global.  dzp comes back from global memory (the lane's own
    //      stores of phase A) while the first product runs, and takes dhp's place in planes P --------------------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        u32x4_t dzk[NT][4];
        if (j == 0) {
#pragma unroll
            for (int jj = 0; jj < NT; ++jj)
#pragma unroll
                for (int rnd = 0; rnd < 4; ++rnd) dzk[jj][rnd] = ld16(sdzr, ((16 * rnd + er) * 2 * C + 128 * jj + ec) * 2);
        }
        f32x16 acc[2];
        kloop(acc, Rp, true, a.UrTf, 4 * j + w, a.UzTf, 4 * j + w);
        if (j == 0) FB_MARK(4);
        if (j == 0) {
#pragma unroll
            for (int jj = 0; jj < NT; ++jj)
#pragma unroll
                for (int rnd = 0; rnd < 4; ++rnd) *reinterpret_cast<u32x4_t*>(Pp + plane_off(16 * rnd + er, 128 * jj + ec)) = dzk[jj][rnd];
            __syncthreads();                                    // planes P (dzp) complete
        }
        kloop(acc, Pp, false, a.UzTf, 4 * j + w, j + 1 < NT ? a.UrTf : nullptr, 4 * (j + 1) + w);
        if (j + 1 == NT) request_tile0(tnext);                   // the next tile's first operands, before this tile's last stores
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            stage(acc, rnd);
            const V8 v = img8();
            const V8 d = f_widen8(dhk[j][rnd]);
            V8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool pos = ((j ? hpos1 : hpos0) >> (8 * rnd + i)) & 1u;
                o.v[i] = cb_ds(d.v[i], v.v[i], (a.act_lrelu && !pos) ? a.slope : 1.0f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(f_pack8(o), sdh, ((16 * rnd + er) * C + 128 * j + ec) * 2, 0, 0);
        }
        FB_MARK(5 + j);
    }
    tile = tdrawn;
    }
#undef FB_MARK
}

int launch_fused_backward(const FusedBwdArgs& a_, int C, hipStream_t st) {
    FusedBwdArgs a = a_;
    a.trace = fused_trace_buffer(2, (a_.M + FT_ROWS - 1) / FT_ROWS);
    REGT_CHECK_ARG(a.M > 0 && a.T > 0 && a.M % a.T == 0, "fused backward: empty problem");
    REGT_CHECK_ARG(C == 256, "fused backward: built for C = 256 (got C = %d)", C);
    const long tiles = (a.M + FT_ROWS - 1) / FT_ROWS;
    REGT_CHECK_ARG(tiles < (1L << 31) && a.M < (1L << 31), "fused backward: too many rows");
    using L = FusedBwdLds<256>;
    static bool attr_done = false;
    if (!attr_done) {
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_bwd_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES));
        attr_done = true;
    }
    const long slots = 2L * fused_cus();                        // persistent: two workgroups per CU
    if (a.tile_ctr) REGT_CHECK_HIP(hipMemsetAsync(a.tile_ctr, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL((fused_bwd_kernel<256>), dim3((unsigned)(tiles < slots ? tiles : slots)), dim3(256), L::BYTES, st, a);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
bool fused_backward_ok(int C) { return C == 256; }

// copies the stamps of the last traced launch to the host (synchronises); returns the number of values
long fused_trace_fetch(long* out, long capacity) {
    if (!g_fused_trace || !out) return 0;
    const long n = capacity < g_fused_trace_n ? capacity : g_fused_trace_n;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(out, g_fused_trace, n * sizeof(long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
int launch_fused_forward(const FusedFwdArgs& a_, int C, int F, hipStream_t st) {
    REGT_CHECK_ARG(a_.M > 0 && a_.T > 0, "fused forward: empty problem");
    FusedFwdArgs a = a_;
    a.dbg = 0;                  // (timing-only descriptor switches of the kernel: developer builds set them, see tools/fused_ablation.sh)
    REGT_CHECK_ARG(a.M % a.T == 0, "fused forward: M = %ld rows are no whole number of T = %d periods", a.M, a.T);
    a.nodes = a.M / a.T;
    a.pmask = 0;
    for (int r = 0; r < 64; r += a.T) a.pmask |= 1ull << r;
    a.trace = fused_trace_buffer(1, (a_.M + FT_ROWS - 1) / FT_ROWS);
    REGT_CHECK_ARG(fused_forward_ok(C, F), "fused forward: built for C = 256, F = 32 or 64 (got C = %d, F = %d)", C, F);
    const long tiles = (a.M + FT_ROWS - 1) / FT_ROWS;
    REGT_CHECK_ARG(tiles < (1L << 31), "fused forward: too many tiles");
    using L = FusedFwdLds<256, 64>;              // (the LDS footprint does not depend on F)
    static bool attr_done = false;
    if (!attr_done) {
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_kernel<256, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES));
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_kernel<256, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES));
        attr_done = true;
    }
    // persistent: two workgroups per CU (what LDS and registers admit), each walks its tiles with a stride of the grid
    const long slots = 2L * fused_cus();
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    if (a.tile_ctr) REGT_CHECK_HIP(hipMemsetAsync(a.tile_ctr, 0, sizeof(unsigned), st));
    if (F == 64) hipLaunchKernelGGL((fused_fwd_kernel<256, 64>), dim3(grid), dim3(256), L::BYTES, st, a);
    else hipLaunchKernelGGL((fused_fwd_kernel<256, 32>), dim3(grid), dim3(256), L::BYTES, st, a);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
bool fused_forward_ok(int C, int F) { return C == 256 && (F == 64 || F == 32); }

}  // namespace regt
