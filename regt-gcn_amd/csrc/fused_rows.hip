// Fused forward of the RegT-GCN cell for the bf16 arithmetic, second form: a wave owns 16 WHOLE rows, the weights stream through LDS.
//
// Same op sites as fused.hip (regional embedding -> gates -> candidate -> GRU blend -> attention-weighted sum over the periods:
// models/RegionalTemporalGCN.py:136-148, models/utils.py:168-188, composed-weight form of DESIGN.md section 3), same arithmetic to
// the bit.  What changes is who owns what.  fused_fwd_kernel gives a wave a 32-column strip of a 64-row tile: the activations h / q
// cross LDS as A operands (two plane sets, two workgroup barriers) and every wave pulls ITS weight fragments straight from L2 --
// 544 KB per 64 rows through the CU's vector-memory path, which is what bounds that kernel (profiles/r05_fused_unit_pmc.txt: TA busy
// 0.74; tools/micro/ta_path.hip: a CU takes 1 KB of 16-byte loads per ~25 cycles, of stores per ~30).  Here:
//   * a workgroup is 8 waves x 16 rows = 128 rows; a wave computes ALL 256 columns of its rows, 128 at a time (8 accumulators of
//     v_mfma_f32_16x16x32_bf16 -- it rounds exactly like two chained 32x32x16, tools/micro/mfma_shape_bits.hip);
//   * the weights are the B operands of every wave alike: they are copied ONCE per workgroup into a ring of LDS slices (32 k x 128
//     columns = 8 KB, each wave copies 1 KB of it with one LDS-DMA instruction, `global_load_lds_dwordx4`, seven slices ahead) and
//     read from there by all eight waves -- half the L2 -> CU weight traffic per row of the 64-row kernel;
//   * the A operands never leave the wave: the epilogue's (row, 8 consecutive columns) layout IS the A-fragment layout of the
//     16x16x32 instruction, so the packed bf16 output of one stage (h, q = h R) is the next stage's A operand as it stands, in
//     registers.  No planes, no barriers between the stages; the only barriers are the ring's (one per slice).
// The epilogue transposes 16 x 32 accumulator pieces through a wave-private 2 KB image as in fused.hip.  Per-node sums: a wave sums
// the rows of a node inside ITS 16-row block in row order and hands partial sums over with atomic adds -- two addends at most per
// element while T <= 16 (a node then meets at most two blocks); the three-launch path sums the same blocks (`node_sum_rows` of
// CandArgs), so outputs stay bit-identical to it (tests/test_gpu_fused.py).  Longer periods, C != 256 or region ids that are not
// sorted by node keep fused_fwd_kernel.
#include "fused_common.h"

namespace regt {

namespace {

constexpr int FR_ROWS = 128;             // rows of a tile: 8 waves x 16
constexpr int FR_SLOTS = 8;              // ring slots
constexpr int FR_AHEAD = 7;              // slices requested ahead of the one being consumed
constexpr int FR_SLICE_B = 8192;         // 32 k x 128 columns of bf16 in MFMA fragment order (eight 1 KB blocks of launch_cvt_bf16_frag)
constexpr int FR_IMG_B = 2048;           // a wave's epilogue image: 16 rows x 32 columns fp32
constexpr int FR_C = 256;

// (biases and images first: their addresses then fit the 16-bit offset field of the LDS instructions -- behind the 64 KB ring every
// one of the 32 bias reads of a tile needed an address register of its own, which hipcc spilled)
struct FusedRowsLds {
    static constexpr int BIAS_OFF = 0;
    static constexpr int IMG_OFF = BIAS_OFF + 4 * FR_C * 4;
    static constexpr int RING_OFF = IMG_OFF + 8 * FR_IMG_B;
    static constexpr int BYTES = RING_OFF + FR_SLOTS * FR_SLICE_B;
};

typedef __attribute__((address_space(3))) void fr_lds_void;

// swizzle of the image: the 16-byte chunk c of row r lies at chunk c ^ fr_par(r).  ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) over 64 banks: with lane = (row l & 15, chunks 2 (l >> 4), 2 (l >> 4) + 1) a group
// holds rows {0-3, 12-15} of one chunk pair and rows {4-11} of the next -- par(r) = ((r >> 1) & 3) | (r & 4) spreads the eight rows
// of equal parity over the eight chunks of their half of the bank row; the accumulator rows 4 g + i and 4 (g + 1) + i that one
// ds_write_b32 group holds differ in bit 2 of the chunk.
__device__ __forceinline__ constexpr int fr_par(int r) { return ((r >> 1) & 3) | (r & 4); }

}  // namespace

// what the ring's producer and the tile loop need to know about a tile: its first region and the number of regions it touches
// (region ids sorted by node: a tile's regions are a range)
struct FrTileInfo { int rg_first, nreg; };

template <int F>
__global__ __launch_bounds__(512, 2) void fused_fwd_rows_kernel(FusedFwdArgs a) {
    constexpr int C = FR_C, KF = F / 32;                         // k blocks (32 k) of a K = F operand
    static_assert(F == 32 || F == 64, "row widths");
    using L = FusedRowsLds;
    extern __shared__ __attribute__((aligned(16))) char flds[];
    float* biasl = reinterpret_cast<float*>(flds + L::BIAS_OFF);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4, lr = lane & 31;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);       // the wave: rows 16 v .. 16 v + 15 of the tile
    const unsigned uT = (unsigned)a.T;
    const long tiles = (a.M + FR_ROWS - 1) / FR_ROWS;
    const unsigned ring_lds = (unsigned)(size_t)(fr_lds_void*)(flds + L::RING_OFF);      // LDS byte address of the ring
    float* imgw = reinterpret_cast<float*>(flds + L::IMG_OFF + v * FR_IMG_B);
// developer trace (REGT_FUSED_TRACE=1, tools/fused_trace.py rows): stamps of thread 0 -- 0 tile start; per unit u (embedding j=0,1; R j=0,1;
    // Z0, cand0, Z1, cand1): 1 + 2 u after its K loop, 2 + 2 u after its epilogue
#define FT_MARK(i) do { if (a.trace && tid == 0) a.trace[(long)FT_TRACE_SLOTS * tile + (i)] = (long)__builtin_amdgcn_s_memtime(); } while (0)

    auto tile_info = [&](long tile) {
        FrTileInfo t{0, 1};
        if (a.node_region && tile < tiles) {
            const long m0 = tile * FR_ROWS, m1 = (m0 + FR_ROWS < a.M ? m0 + FR_ROWS : a.M) - 1;
            t.rg_first = a.node_region[(unsigned)m0 / uT];
            t.nreg = a.node_region[(unsigned)m1 / uT] - t.rg_first + 1;
        }
        return t;
    };

    // ---- the ring: one LDS-DMA per wave and slice, FR_AHEAD slices ahead of the consumer ----------------------------------------------
    // A tile's slices in the order they are consumed (one slice = 32 k x the 128 output columns of N-half j); the sequence is the
    // same for every tile, so the request that accompanies slice s is known where the code is written (s is a constant at every
    // call site once the loops are unrolled):
    //   segments  0..3: A0 j=0, A_region j=0, A0 j=1, A_region j=1  (KF slices each; the region is the tile's FIRST one -- further
    //                   regions of a tile, rare, are read straight from L2, see the embedding below)
    //             4..7: Ur j=0, Gr j=0, Ur j=1, Gr j=1               (8 / KF slices)
    //            8..15: Uz, Gz, Uh, Gh for j=0, then for j=1
    // Wave v copies block (row block nb + v / 2, 16-k block 2 kb + v % 2) of the matrix: 1 KB, contiguous, lane-linear.
    constexpr int S_TILE = 4 * KF + 6 * (8 + KF);
    const int dma_voff = lane * 16;
    unsigned voff_c = (unsigned)(((v >> 1) * (C / 16) + (v & 1)) * 1024), voff_f = (unsigned)(((v >> 1) * (F / 16) + (v & 1)) * 1024);
    unsigned p_slot = 0;                                         // ring slot of the next slice to be requested
    // request slice s (0 <= s < S_TILE) of a tile whose first region is rg
    auto request = [&](int s, int rg) {
        int seg, kb;
        if (s < 4 * KF) { seg = s / KF; kb = s % KF; }
        else {
            const int u = s - 4 * KF, pr = u / (8 + KF), w = u % (8 + KF);     // pair (U, G) number pr = 0..5, position inside it
            seg = 4 + 2 * pr + (w >= 8 ? 1 : 0);
            kb = w >= 8 ? w - 8 : w;
        }
        // (one base pointer + 32-bit offsets: with a 64-bit pointer per matrix the 75 request sites keep ~40 scalar registers busy)
        unsigned mo; int nb, k16;
        if (seg < 4) { mo = (seg & 1) ? a.o_aall + (unsigned)rg * (unsigned)a.ar_stride : a.o_a0; nb = 4 * (seg >> 1); k16 = F / 16; }
        else if (seg < 8) { const int j = (seg - 4) >> 1; if (seg & 1) { mo = a.o_gzr; nb = C / 32 + 4 * j; k16 = F / 16; } else { mo = a.o_ur; nb = 4 * j; k16 = C / 16; } }
        else {
            const int t = seg - 8, j = t >> 2, w = t & 3;                       // Uz Gz Uh Gh
            mo = w == 0 ? a.o_uz : (w == 1 ? a.o_gzr : (w == 2 ? a.o_uh : a.o_gh));
            nb = 4 * j; k16 = (w & 1) ? F / 16 : C / 16;
        }
        const char* src = a.wbase + (mo + (k16 == C / 16 ? voff_c : voff_f) + (unsigned)((nb * k16 + 2 * kb) * 1024));
        const unsigned m0v = ring_lds + p_slot * FR_SLICE_B + v * 1024;
        // (inline asm: hipcc drains the builtin form -- vmcnt(0) in front of every LDS read; the waits for these requests are
        // written by hand in `consume`.  M0 = LDS base of the wave's 1 KB, lane l lands at + 16 l)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(dma_voff), "s"(src) : "memory");
        p_slot = (p_slot + 1) & (FR_SLOTS - 1);
    };
    // ---- the consumer side of the ring ------------------------------------------------------------------------------------------------
    unsigned c_slot = 0;
    // B fragment of column block cb (16 columns) of the slice: lane (c = l & 15, g): k = 8 g .. 8 g + 7 of column c.  In the slice's
    // blocks (32 columns x 16 k each, lane' = 32 (k-group % 2) + column % 32): block 2 (cb / 2) + g / 2, lane' 32 (g % 2) + 16 (cb % 2) + c
    const int bf_lane = 1024 * (g >> 1) + 512 * (g & 1) + 16 * r;
    int rg_cur = 0, rg_nxt = 0;                                  // first region of the tile at hand / of the workgroup's next tile
    bool has_next = false;
    // acc[cb] += A (16 rows x 32 k, this wave's) x slice[cb]^T for the 8 column blocks of slice s of the tile.  The B fragments are
    // read one HALF slice ahead of the MFMAs that use them (bA: column blocks 0..3 of the slice at hand, read during the previous
    // step; bB: blocks 4..7, read at the start of this one), so an LDS round trip always has four MFMAs of this wave in front of it:
    //     read bB(s) | MFMA bA(s) | slice s + 1 landed: wait + barrier | request slice s + 7 | read bA(s + 1) | MFMA bB(s)
    // The barrier in the middle says (a) every wave's piece of slice s + 1 has landed and (b) every wave has finished step s - 1,
    // i.e. all reads of slice s - 1 -- whose slot the new request overwrites.
    // vector-memory operations (other than ring requests) issued right after slice t of a tile: the 16-byte stores of the epilogue
    // that follows it (the per-node sums' stores / atomics come on top: not counted, the wait is then a little stricter than needed),
    // after the last slice also the next tile's row loads
    auto vm_epi = [&](int t) {
        const int u0 = 2 * KF, g0 = 4 * KF, gl = 8 + KF;
        if (t == u0 - 1 || t == 2 * u0 - 1) return 4;                                         // embedding: h
        if (t == g0 + gl - 1 || t == g0 + 2 * gl - 1) return 8;                               // R: R and q
        if (t == g0 + 3 * gl - 1 || t == g0 + 5 * gl - 1) return 4;                           // Z
        if (t == g0 + 4 * gl - 1) return 4;                                                   // candidate: H~
        if (t == g0 + 6 * gl - 1) return 4 + 3 * KF + 1;                                      // ... and the next tile's rows
        return 0;
    };
    // ... issued between the request for slice s + 1 (in the middle of step s - 6) and the middle of step s
    auto vm_after = [&](int s) {
        int n = 0;
#pragma unroll
        for (int t = s - 6; t < s; ++t) n += vm_epi(t < 0 ? t + S_TILE : t);
        return n;
    };
    auto wait_landed = [&](int n) {                              // (n is a constant at every call site: the switch folds)
        switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")\n\ts_barrier" ::: "memory"); break;
            W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16) W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28)
#undef W_
            default: asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); break;
        }
    };
    bf16x8 bA[4];
    auto read_half = [&](bf16x8 (&b)[4], unsigned slot, int hf) {
        const char* sl = flds + L::RING_OFF + slot * FR_SLICE_B + bf_lane + 4096 * hf;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) b[cb] = *reinterpret_cast<const bf16x8*>(sl + 2048 * (cb >> 1) + 256 * (cb & 1));
    };
    auto consume = [&](int s, f32x4 (&acc)[8], const bf16x8& af) {
        bf16x8 bB[4];
        read_half(bB, c_slot, 1);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bA[cb], acc[cb], 0, 0, 0);
        // This wave's own request for slice s + 1 is done when all of its vector-memory operations are, except the ones issued after
        // it (they retire in issue order): the requests s + 2 .. s + 6 and whatever the epilogues in between stored -- vm_after(s),
        // counted from the static schedule.  Counting too few is safe (a stricter wait); counting exactly matters: waiting for an
        // epilogue's last stores (or the next tile's row loads) to be acknowledged cost every K loop its first ~2 k cycles.
        if (s + FR_AHEAD < S_TILE) {
            wait_landed(5 + vm_after(s));
            request(s + FR_AHEAD, rg_cur);
        } else if (has_next) {
            wait_landed(5 + vm_after(s));
            request(s + FR_AHEAD - S_TILE, rg_nxt);
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // (the workgroup's last tile: fewer requests are outstanding)
        }
        c_slot = (c_slot + 1) & (FR_SLOTS - 1);
        read_half(bA, c_slot, 0);                                // (past the workgroup's last slice: stale bytes, never used)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[4 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bB[cb], acc[4 + cb], 0, 0, 0);
    };
    static_assert(FR_AHEAD == 7 && FR_SLOTS == 8, "the hand-written vmcnt counts and the slot arithmetic assume seven requests ahead in eight slots");

    // ---- epilogue geometry: accumulator pieces of 16 rows x 32 columns through the wave's image ------------------------------------
    // accumulator lane (c = l & 15, g): rows 4 g + i (i = register), column c of its 16-column block; epilogue lane (r = l & 15, g):
    // row r, columns 8 g .. 8 g + 7 of the 32-column piece = 16 bytes of a bf16 array = the A fragment of that 32-k block
    // (the offsets are recomputed from the lane number at the start of every epilogue: kept in registers across the K loops they
    // are what hipcc spills, and a scratch reload waits for vmcnt(0) -- which drains the ring)
    struct EpiGeo { int st_row, st_col[2][2], e_lo, e_hi, ro, rzo, lr, lane; };
    auto epi_geo = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int rr_ = l & 15, gg = l >> 4;
        EpiGeo e;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) e.st_col[b][ih] = 4 * ((4 * b + (rr_ >> 2)) ^ (6 * (gg & 1) + ih)) + (rr_ & 3);   // [16-column block of the piece][i >> 1]
        e.st_row = 4 * gg * 32;
        e.e_lo = rr_ * 32 + 4 * ((2 * gg) ^ fr_par(rr_));
        e.e_hi = rr_ * 32 + 4 * ((2 * gg + 1) ^ fr_par(rr_));
        e.ro = rr_ * C * 2 + gg * 16;                            // the lane's 16 bytes in a row of an (M x C) bf16 array
        e.rzo = rr_ * C * 4 + gg * 16;                           //   ... of the (M x 2C) array [Z | R]
        e.lr = l & 31; e.lane = l;
        return e;
    };
    auto stage = [&](const EpiGeo& e, const f32x4& a0, const f32x4& a1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            imgw[e.st_row + i * 32 + e.st_col[0][i >> 1]] = a0[i];
            imgw[e.st_row + i * 32 + e.st_col[1][i >> 1]] = a1[i];
        }
    };
    auto img8 = [&](const EpiGeo& e) {
        const float4 lo = *reinterpret_cast<const float4*>(imgw + e.e_lo), hi = *reinterpret_cast<const float4*>(imgw + e.e_hi);
        return V8{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
    auto bias8 = [&](int i) {                                    // (i: offset into [b' | cz | cr | ch])
        const float4 lo = *reinterpret_cast<const float4*>(biasl + i), hi = *reinterpret_cast<const float4*>(biasl + i + 4);
        return V8{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
    auto zero8 = [&](f32x4 (&acc)[8]) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    for (int i = tid; i < 4 * C; i += 512) biasl[i] = i < C ? a.bprime[i] : (i < 3 * C ? a.czr[i - C] : a.ch[i - 3 * C]);
    __syncthreads();                                            // the biases are in LDS

    // ---- what a tile needs from global memory, requested one tile ahead: the wave's rows of x, L~ x, A_hat x as A fragments (lane
    //      (r, g): k = 32 kb + 8 g .. + 7 of row r), the region and the attention probability of the lane's row
    bf16x8 xf[KF], lf[KF], axf[KF];
    int rg_row; float pt_row;
    auto request_rows = [&](long tile) {
        const long m0 = tile * FR_ROWS + 16 * v;                 // (past the last tile / row: zero records, loads return 0)
        const long left = a.M - m0;
        const int nv = (int)(left < 0 ? 0 : (left < 16 ? left : 16));
        const __amdgpu_buffer_rsrc_t sX = f_rsrc(reinterpret_cast<const char*>(a.X) + m0 * F * 2, (long)nv * F * 2);
        const __amdgpu_buffer_rsrc_t sLX = f_rsrc(reinterpret_cast<const char*>(a.LX) + m0 * F * 2, (long)nv * F * 2);
        const __amdgpu_buffer_rsrc_t sAX = f_rsrc(reinterpret_cast<const char*>(a.AX) + m0 * F * 2, (long)nv * F * 2);
        const int afo = r * F * 2 + g * 16;
#pragma unroll
        for (int kb = 0; kb < KF; ++kb) {
            xf[kb] = f_ldfrag(sX, afo, kb * 64);
            lf[kb] = f_ldfrag(sLX, afo, kb * 64);
            axf[kb] = f_ldfrag(sAX, afo, kb * 64);
        }
        const unsigned m = (unsigned)m0 + (unsigned)r, nd = m / uT;
        const bool ok = r < nv;
        rg_row = ok ? (a.node_region ? a.node_region[nd] : 0) : -1;
        pt_row = a.probs[ok ? m - nd * uT : 0];
    };
    long tile = blockIdx.x;
    FrTileInfo info = tile_info(tile);
    request_rows(tile);
#pragma unroll
    for (int s = 0; s < FR_AHEAD; ++s) request(s, info.rg_first);
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");              // slice 0 has landed (everything older than the 6 youngest requests is done)
    read_half(bA, 0, 0);

#pragma unroll 1
    for (; tile < tiles; tile += gridDim.x) {
        const long m0 = tile * FR_ROWS + 16 * v;                 // the wave's first row
        const long left = a.M - m0;
        const int nv = (int)(left < 0 ? 0 : (left < 16 ? left : 16));      // its valid rows
        const unsigned mrow0 = (unsigned)m0;
        const unsigned node0 = mrow0 / uT;
        const int t0 = (int)(mrow0 - node0 * uT);
        const long tnext = tile + gridDim.x;
        const FrTileInfo info_next = tile_info(tnext);
        rg_cur = info.rg_first; rg_nxt = info_next.rg_first; has_next = tnext < tiles;
        // (opaque per tile: left alone, hipcc hoists the source addresses of all 75 request sites out of the tile loop -- 150 scalar
        // registers, spilled to vector lanes)
        asm volatile("" : "+s"(voff_c), "+s"(voff_f));
        const int nreg = info.nreg;
        // node boundaries of the wave's 16 rows as bit masks (all scalar; see fused.hip): starts, ends, ends that are partial sums
        const unsigned vmask = nv >= 16 ? 0xffffu : ((1u << nv) - 1u);
        const int s0 = t0 == 0 ? 0 : a.T - t0;
        const unsigned smask = s0 < 16 ? (unsigned)(a.pmask << s0) & vmask : 0u;
        const unsigned emask = nv > 0 ? ((smask >> 1) | (1u << (nv - 1))) & vmask : 0u;
        const unsigned amask = (t0 != 0 ? emask & (0u - emask) : 0u) | ((nv > 0 && ((unsigned)(t0 + nv) % uT) != 0) ? 1u << (nv - 1) : 0u);
        FT_MARK(0);
        const __amdgpu_buffer_rsrc_t sh = f_rsrc(reinterpret_cast<char*>(a.h) + m0 * C * 2, (long)nv * C * 2);
        const __amdgpu_buffer_rsrc_t sq = f_rsrc(reinterpret_cast<char*>(a.q) + m0 * C * 2, (long)nv * C * 2);
        const __amdgpu_buffer_rsrc_t sHt = f_rsrc(reinterpret_cast<char*>(a.Ht) + m0 * C * 2, (long)nv * C * 2);
        const __amdgpu_buffer_rsrc_t sZR = f_rsrc(reinterpret_cast<char*>(a.ZR) + m0 * C * 4, (long)nv * C * 4);
        const int rgl = rg_row; const float pt = pt_row;
        bf16x8 hA[8], qA[8], axA[KF];
#pragma unroll
        for (int kb = 0; kb < KF; ++kb) axA[kb] = axf[kb];

        // ---- regional embedding h = act(x A0^T + (L~ x) A_region^T + b'): the output IS the A operand of both gates ------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 acc[8];
            zero8(acc);
            const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < KF; ++kb) consume(2 * KF * j + kb, acc, xf[kb]);
            {                                                    // the tile's first region: rows of other regions contribute zeros
                const bool mine = rgl == rg_cur;
#pragma unroll
                for (int kb = 0; kb < KF; ++kb) consume(2 * KF * j + KF + kb, acc, mine ? lf[kb] : zero);
            }
#pragma unroll 1
            for (int p = 1; p < nreg; ++p) {                     // further regions of the tile (rare): their weights straight from L2
                const bool mine = rgl == rg_cur + p;
                const __amdgpu_buffer_rsrc_t sAr = f_rsrc(reinterpret_cast<const char*>(a.Aallf) + (long)(rg_cur + p) * a.ar_stride, (long)C * F * 2);
#pragma unroll
                for (int kb = 0; kb < KF; ++kb) {
                    const bf16x8 af = mine ? lf[kb] : zero;
                    bf16x8 bfr[8];
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb)     // block (4 j + cb / 2, 2 kb + g / 2), lane' 32 (g % 2) + 16 (cb % 2) + c: as in the slices
                        bfr[cb] = f_ldfrag(sAr, bf_lane + 256 * (cb & 1), ((4 * j + (cb >> 1)) * (F / 16) + 2 * kb) * 1024);
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[cb], acc[cb], 0, 0, 0);
                }
            }
            FT_MARK(1 + 2 * j);
            const EpiGeo eg = epi_geo();
            const float ns = a.act_lrelu ? a.slope : 1.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);      // (one round at a time: interleaved rounds cost registers, then spills whose reloads drain vmcnt)
                stage(eg, acc[2 * q], acc[2 * q + 1]);
                const V8 vv = img8(eg);
                const V8 b = bias8(128 * j + 32 * q + 8 * g);
                V8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float s = vv.v[i] + b.v[i]; o.v[i] = s > 0.f ? s : s * ns; }
                const u32x4_t pk = f_pack8(o);
                __builtin_amdgcn_raw_buffer_store_b128(pk, sh, eg.ro + (128 * j + 32 * q) * 2, 0, 0);
                hA[4 * j + q] = __builtin_bit_cast(bf16x8, pk);
            }
            FT_MARK(2 + 2 * j);
        }
        // ---- reset gate R = sigmoid(h Ur^T + (A_hat x) Gr^T + cr), q = h R: the candidate's A operand --------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 acc[8];
            zero8(acc);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) consume(4 * KF + (8 + KF) * j + kb, acc, hA[kb]);
#pragma unroll
            for (int kb = 0; kb < KF; ++kb) consume(4 * KF + (8 + KF) * j + 8 + kb, acc, axA[kb]);
            FT_MARK(5 + 2 * j);
            const EpiGeo eg = epi_geo();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                stage(eg, acc[2 * q], acc[2 * q + 1]);
                const V8 vv = img8(eg);
                const V8 b = bias8(2 * C + 128 * j + 32 * q + 8 * g);
                const V8 hv = f_widen8(__builtin_bit_cast(u32x4_t, hA[4 * j + q]));
                const V8 gt = f_sigmoid8(vv, b);
                V8 qv;
#pragma unroll
                for (int i = 0; i < 8; ++i) qv.v[i] = hv.v[i] * gt.v[i];
                __builtin_amdgcn_raw_buffer_store_b128(f_pack8(gt), sZR, eg.rzo + (C + 128 * j + 32 * q) * 2, 0, 0);
                const u32x4_t pq = f_pack8(qv);
                __builtin_amdgcn_raw_buffer_store_b128(pq, sq, eg.ro + (128 * j + 32 * q) * 2, 0, 0);
                qA[4 * j + q] = __builtin_bit_cast(bf16x8, pq);
            }
            FT_MARK(6 + 2 * j);
        }
        // ---- per 128 columns: update gate Z (kept packed), candidate H~, blend, per-node sums ----------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4_t zk[4];
            {
                f32x4 acc[8];
                zero8(acc);
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) consume(4 * KF + (8 + KF) * (2 + 2 * j) + kb, acc, hA[kb]);
#pragma unroll
                for (int kb = 0; kb < KF; ++kb) consume(4 * KF + (8 + KF) * (2 + 2 * j) + 8 + kb, acc, axA[kb]);
                FT_MARK(9 + 4 * j);
                const EpiGeo eg = epi_geo();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_barrier(0);
                    stage(eg, acc[2 * q], acc[2 * q + 1]);
                    const V8 vv = img8(eg);
                    const V8 b = bias8(C + 128 * j + 32 * q + 8 * g);
                    zk[q] = f_pack8(f_sigmoid8(vv, b));
                    __builtin_amdgcn_raw_buffer_store_b128(zk[q], sZR, eg.rzo + (128 * j + 32 * q) * 2, 0, 0);
                }
            }
            FT_MARK(10 + 4 * j);
            f32x4 acc[8];
            zero8(acc);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) consume(4 * KF + (8 + KF) * (3 + 2 * j) + kb, acc, qA[kb]);
#pragma unroll
            for (int kb = 0; kb < KF; ++kb) consume(4 * KF + (8 + KF) * (3 + 2 * j) + 8 + kb, acc, axA[kb]);
            FT_MARK(11 + 4 * j);
            const EpiGeo eg = epi_geo();
            if (j == 1) { info = info_next; request_rows(tnext); }   // the next tile's rows, before this tile's last stores
            const long ohcol = (long)node0 * C + 128 * j;
            const __amdgpu_buffer_rsrc_t sOH = f_rsrc(a.OH + ohcol, (a.nodes * C - ohcol) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                stage(eg, acc[2 * q], acc[2 * q + 1]);
                const V8 vv = img8(eg);
                const V8 b = bias8(3 * C + 128 * j + 32 * q + 8 * g);
                const V8 hv = f_widen8(__builtin_bit_cast(u32x4_t, hA[4 * j + q]));
                const V8 Zv = f_widen8(zk[q]);
                const V8 ht = f_tanh8(vv, b);
                V8 bl;
#pragma unroll
                for (int i = 0; i < 8; ++i) bl.v[i] = __fmul_rn(pt, gru_blend(Zv.v[i], hv.v[i], ht.v[i]));
                __builtin_amdgcn_raw_buffer_store_b128(f_pack8(ht), sHt, eg.ro + (128 * j + 32 * q) * 2, 0, 0);
                *reinterpret_cast<float4*>(imgw + eg.e_lo) = make_float4(bl.v[0], bl.v[1], bl.v[2], bl.v[3]);
                *reinterpret_cast<float4*>(imgw + eg.e_hi) = make_float4(bl.v[4], bl.v[5], bl.v[6], bl.v[7]);
                // Per-node sums over the wave's 16 rows, in row order (see fused.hip: one running sum per lane = column lr of the
                // piece; lanes 32..63 duplicate and store nothing; a start row multiplies the carried sum by 0, an end row hands it
                // over -- a plain store when all of the node's rows lie in this block, else an atomic add).
                float cv[16];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) cv[rr] = imgw[rr * 32 + (eg.lr ^ (4 * fr_par(rr)))];
                unsigned sm = smask, em = emask, am = amask;
                asm volatile("" : "+s"(sm), "+s"(em), "+s"(am));
                const int ohv = eg.lane < 32 ? (32 * q + eg.lr) * 4 : 0x7ffffff0;
                float csum = 0.f;
                int ohs = 0;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    float keep;
                    asm volatile("s_bitcmp1_b32 %1, %2\n\ts_cselect_b32 %0, 0, 1.0" : "=s"(keep) : "s"(sm), "n"(rr) : "scc");
                    csum = fmaf(csum, keep, cv[rr]);
                    if ((em >> rr) & 1u) {
                        const bool part = (am >> rr) & 1u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(csum), sOH, ohv, part ? 0x7ffffff0 : ohs, 0);
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(csum, sOH, ohv, part ? ohs : 0x7ffffff0, 0);
                        ohs += C * 4;
                    }
                }
            }
            FT_MARK(12 + 4 * j);
        }
    }
    // (the ring holds no request any more: the producer stopped with the last tile's last slice, which has been consumed)
#undef FT_MARK
}

long* fused_trace_buffer(int which, long tiles);
int fused_cus();

bool fused_forward_rows_ok(int C, int F, int T) { return C == FR_C && (F == 64 || F == 32) && T <= 16; }

int launch_fused_forward_rows(const FusedFwdArgs& a_, int C, int F, hipStream_t st) {
    REGT_CHECK_ARG(a_.M > 0 && a_.T > 0, "fused forward: empty problem");
    REGT_CHECK_ARG(fused_forward_rows_ok(C, F, a_.T), "fused forward (row form): built for C = 256, F = 32 or 64, T <= 16 (got C = %d, F = %d, T = %d)", C, F, a_.T);
    FusedFwdArgs a = a_;
    REGT_CHECK_ARG(a.M % a.T == 0, "fused forward: M = %ld rows are no whole number of T = %d periods", a.M, a.T);
    a.nodes = a.M / a.T;
    a.pmask = 0;
    for (int rr = 0; rr < 64; rr += a.T) a.pmask |= 1ull << rr;
    const long tiles = (a.M + FR_ROWS - 1) / FR_ROWS;
    REGT_CHECK_ARG(a.M < (1L << 31), "fused forward: too many rows");
    a.trace = fused_trace_buffer(1, tiles);
    {   // the weight blocks live in one workspace buffer (api.hip wb_ptrs): one base + 32-bit offsets
        const char* ptrs[7] = {(const char*)a.Uzf, (const char*)a.Urf, (const char*)a.Uhf, (const char*)a.Gzrf, (const char*)a.Ghf, (const char*)a.A0f, (const char*)a.Aallf};
        const char* base = ptrs[0];
        for (int i = 1; i < 7; ++i) base = ptrs[i] < base ? ptrs[i] : base;
        unsigned* offs[7] = {&a.o_uz, &a.o_ur, &a.o_uh, &a.o_gzr, &a.o_gh, &a.o_a0, &a.o_aall};
        for (int i = 0; i < 7; ++i) {
            const long o = ptrs[i] - base;
            REGT_CHECK_ARG(o >= 0 && o < (1L << 30) && a.ar_stride >= 0 && a.ar_stride < (1L << 24), "fused forward (row form): weight blocks too far apart");
            *offs[i] = (unsigned)o;
        }
        a.wbase = base;
    }
    using L = FusedRowsLds;
    static bool attr_done = false;
    if (!attr_done) {
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_rows_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES));
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_rows_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES));
        attr_done = true;
    }
    // persistent: one workgroup of eight waves per CU
    const long slots = fused_cus();
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    if (F == 64) hipLaunchKernelGGL((fused_fwd_rows_kernel<64>), dim3(grid), dim3(512), L::BYTES, st, a);
    else hipLaunchKernelGGL((fused_fwd_rows_kernel<32>), dim3(grid), dim3(512), L::BYTES, st, a);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
