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
// Tried on this schedule and dropped, all bit-identical (profiles/r05_fused_rows_offset_form.txt): waves 4-7 running four
// barrier-to-barrier steps behind waves 0-3, so that every SIMD has one wave in a K loop and one in an epilogue (1.91 vs 1.69 ms: a
// step costs ~1000 cycles either way -- barrier + LDS-DMA issue, not MFMA -- and the candidate rounds of the two groups then run one
// after the other); the whole accumulator block transposed through the image before the epilogue rounds (1.72 vs 1.63); the next
// tile's rows and the region ids loaded in inline asm with hand-placed waits, so that hipcc's own vmcnt(k) in front of their first
// use does not wait for the ring's requests (1.667 vs 1.630: what those waits drain has landed by then anyway); per-node sums by a
// DPP scan in the epilogue layout or with one store + one atomic per round (1.85 / 1.68-1.87 vs 1.65: the epilogues are bound by
// instruction issue).
#include <type_traits>

#include "fused_common.h"

namespace regt {

namespace {

// NW waves x 16 rows make a tile.  NW = 8: one workgroup per CU, 16 ring slots (128 KB: a request takes ~3 k cycles to land under load,
// and what is in flight sets the rate -- with 8 slots the K loops waited for the ring, matrix work or not).  NW = 4: two workgroups
// per CU with a ring of 8 slots each -- twice the L2 -> LDS weight traffic per row, one workgroup's epilogues (stores, gate math)
// under the other's K loops: 1.67 vs 1.61 ms, kept as a test form (regt_set_option("fused_rows", 2)).
constexpr int FR_SLICE_B = 8192;         // 32 k x 128 columns of bf16 in MFMA fragment order (eight 1 KB blocks of launch_cvt_bf16_frag)
constexpr int FR_IMG_B = 2048;           // a wave's epilogue image: 16 rows x 32 columns fp32
constexpr int FR_C = 256;

// (biases and images first: their addresses then fit the 16-bit offset field of the LDS instructions -- behind the 64 KB ring every
// one of the 32 bias reads of a tile needed an address register of its own, which hipcc spilled)
template <int NW>
struct FusedRowsLds {
    static constexpr int SLOTS = NW == 8 ? 16 : 8;                // ring slots; SLOTS - 1 slices are requested ahead of the one being consumed
    static constexpr int BIAS_OFF = 0;
    static constexpr int IMG_OFF = BIAS_OFF + 4 * FR_C * 4;
    static constexpr int RING_OFF = IMG_OFF + NW * FR_IMG_B;
    static constexpr int NEXT_OFF = RING_OFF + SLOTS * FR_SLICE_B;   // the workgroup's next tile (drawn from the tile counter)
    static constexpr int BYTES = NEXT_OFF + 16;
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

template <int F, int NW>
__global__ __launch_bounds__(64 * NW, 2) void fused_fwd_rows_kernel(FusedFwdArgs a) {
    constexpr int C = FR_C, KF = F / 32;                         // k blocks (32 k) of a K = F operand
    static_assert(F == 32 || F == 64, "row widths");
    static_assert(NW == 8 || NW == 4, "waves of a workgroup");
    using L = FusedRowsLds<NW>;
    constexpr int FR_ROWS = 16 * NW, FR_SLOTS = L::SLOTS, FR_AHEAD = FR_SLOTS - 1;
    constexpr int RPW = 8 / NW;                                  // LDS-DMA requests of a wave per slice (1 KB each)
    extern __shared__ __attribute__((aligned(16))) char flds[];
    float* biasl = reinterpret_cast<float*>(flds + L::BIAS_OFF);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
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
    // (NW = 4: the wave also copies block v + 4 -- row block + 2 of the same 16-k block)
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
        // written by hand in `consume`.  M0 = LDS base of the wave's 1 KB, lane l lands at + 16 l.  s_nop 3: the source pointer may
        // have been restored from a VGPR lane (v_readlane) by the instruction before the asm statement, and a vector-memory
        // instruction needs five wait states behind a VALU write of an SGPR it reads -- hipcc pads its own instructions, not the
        // inside of an asm statement; an experimental build with asm row loads faulted at address 0 on exactly that.  Free in an A/B)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(dma_voff), "s"(src) : "memory");
        if (RPW == 2) {
            const char* src2 = src + 2 * k16 * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v + 4096u), "v"(dma_voff), "s"(src2) : "memory");
        }
        p_slot = (p_slot + 1) & (FR_SLOTS - 1);
    };
    // ---- the consumer side of the ring ------------------------------------------------------------------------------------------------
    unsigned c_slot = 0;
    // B fragment of column block cb (16 columns) of the slice: lane (c = l & 15, g): k = 8 g .. 8 g + 7 of column c.  In the slice's
    // blocks (32 columns x 16 k each, lane' = 32 (k-group % 2) + column % 32): block 2 (cb / 2) + g / 2, lane' 32 (g % 2) + 16 (cb % 2) + c
    const int bf_lane = 1024 * (g >> 1) + 512 * (g & 1) + 16 * r;
    int rg_cur = 0, rg_nxt = 0;                                  // first region of the tile at hand / of the workgroup's next tile
    bool has_next = false, first_tile = true;
    // acc[cb] += A (16 rows x 32 k, this wave's) x slice[cb]^T for the 8 column blocks of slice s of the tile.  The B fragments are
    // read TWO half slices (eight MFMAs of this wave) ahead of the MFMAs that use them, through three half-slice buffers: half h of a
    // tile (h = 2 s + 0 / 1: column blocks 0..3 / 4..7 of slice s) lives in buffer h % 3 -- static at every call site --
    //     even s: slices s + 1, s + 2 landed: wait + barrier | two requests | read half 2 s + 2 | MFMA half 2 s | read half 2 s + 3 | MFMA half 2 s + 1
    // (all eight waves read at once after a barrier: with only four MFMAs in front of a read its latency was in the open -- 500
    // cycles per slice for 256 of matrix work).  The barrier says (a) every wave's piece of slice s + 1 has landed and (b) every wave
    // has issued the MFMAs of step s - 1, i.e. has read slice s - 1 -- whose slot the new request overwrites.  A tile's first step
    // reads its own two halves as well (the mapping restarts with every tile).
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
    // ... issued between the request for slice s + 2 (early in step s + 3 - FR_AHEAD, together with the one for s + 3) and the barrier
    // of the even step s: the epilogues behind the slices s + 3 - FR_AHEAD .. s - 1.  (An epilogue behind an earlier slice is OLDER
    // than that request: counting it would let the request itself be one of the operations the wait leaves outstanding.)
    auto vm_after = [&](int s) {
        int n = 0;
#pragma unroll
        for (int t = s - (FR_AHEAD - 3); t < s; ++t) n += vm_epi(t < 0 ? t + S_TILE : t);
        return n;
    };
    auto wait_landed = [&](int n) {                              // (n is a constant at every call site: the switch folds)
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 256)    // timing-only developer builds: no wait and no barrier / (128) barrier only
        return;
#elif defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 128)
        asm volatile("s_barrier" ::: "memory");
        return;
#endif
        switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")\n\ts_barrier" ::: "memory"); break;
            W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16) W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31) W_(32) W_(33) W_(34) W_(35) W_(36) W_(37) W_(38) W_(39) W_(40) W_(41) W_(42) W_(43) W_(44) W_(45) W_(46) W_(47) W_(48) W_(49) W_(50) W_(51) W_(52) W_(53) W_(54) W_(55) W_(56) W_(57) W_(58) W_(59) W_(60) W_(61) W_(62) W_(63)
#undef W_
            default: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
        }
    };
    bf16x8 bq0[4], bq1[4], bq2[4];                               // (three named arrays, chosen by if-chains that fold at every call site: a
                                                                 // [3][4] array indexed with h % 3 ends up in scratch memory)
    auto read_half = [&](int k, unsigned slot, int hf) {
        const char* sl = flds + L::RING_OFF + slot * FR_SLICE_B + bf_lane + 4096 * hf;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 64)     // timing-only developer build: no B fragment reads
            bf16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
            asm volatile("" : "+v"(x) : "v"(sl));
#else
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(sl + 2048 * (cb >> 1) + 256 * (cb & 1));
#endif
            if (k == 0) bq0[cb] = x; else if (k == 1) bq1[cb] = x; else bq2[cb] = x;
        }
    };
    auto mfma_half = [&](int k, f32x4 (&acc)[8], int hf, const bf16x8& af) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const bf16x8 x = k == 0 ? bq0[cb] : (k == 1 ? bq1[cb] : bq2[cb]);
#if defined(REGT_FUSED_ABL) && (REGT_FUSED_ABL & 1)      // timing-only developer build: no matrix instructions (operands kept alive)
            asm volatile("" :: "v"(x), "v"(af));
#else
            if (hf == 0) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, x, acc[cb], 0, 0, 0);
            else acc[4 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, x, acc[4 + cb], 0, 0, 0);
#endif
        }
    };
    auto consume = [&](int s, f32x4 (&acc)[8], const bf16x8& af) {
        // One barrier per TWO slices (even s; 16 MFMAs per wave in between -- with one per slice the K loops ran at ~500 cycles per
        // slice for 256 of matrix work): slices s + 1 and s + 2 have landed -- this wave's own requests for them are done when all of
        // its vector-memory operations are, except the ones issued after them (they retire in issue order): the requests s + 3 ..
        // s + FR_AHEAD - 2 and whatever the epilogues in between stored (vm_after(s), counted from the static schedule; counting too few is
        // safe -- a stricter wait -- but waiting for an epilogue's last stores to be acknowledged costs the K loop its first steps).
        // The barrier also says that every wave has issued the MFMAs of steps s - 2 and s - 1, i.e. has read slices s - 2 and
        // s - 1: their slots take the two new requests.
        if ((s & 1) == 0) {
            if (s + FR_AHEAD < S_TILE) {
                // (the workgroup's first tile has no epilogue behind it: counting one would make the wait too weak)
                if (first_tile && vm_after(s) > 0 && s < FR_AHEAD) wait_landed(RPW * (FR_AHEAD - 4));
                else wait_landed(RPW * (FR_AHEAD - 4) + vm_after(s));
            } else if (has_next) {
                wait_landed(RPW * (FR_AHEAD - 4) + vm_after(s));
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // (the workgroup's last tile: fewer requests are outstanding)
            }
        }
        // The two requests that the barrier of the even step makes room for are issued at two points of that step (waves 0-3 before
        // its first four MFMAs, waves 4-7 before the second four; always inside the even step: the vm_after() counts assume that the
        // requests precede the epilogue that may follow it).  All eight waves asking right behind the barrier queued up at the
        // texture unit -- the last wave started its MFMAs ~300 cycles late, and the next barrier waits for the last wave.
        auto issue_pair = [&](int se) {                          // se: the even step
            if (se + FR_AHEAD < S_TILE) { request(se + FR_AHEAD - 1, rg_cur); request(se + FR_AHEAD, rg_cur); }
            else if (has_next) { request(se + FR_AHEAD - 1 - S_TILE, rg_nxt); request(se + FR_AHEAD - S_TILE, rg_nxt); }
        };
        if ((s & 1) == 0 && v < NW / 2) issue_pair(s);
        const unsigned nslot = (c_slot + 1) & (FR_SLOTS - 1);
        if (s == 0) { read_half(0, c_slot, 0); read_half(1, c_slot, 1); }
        if (s + 1 < S_TILE) read_half((2 * s + 2) % 3, nslot, 0);
        mfma_half((2 * s) % 3, acc, 0, af);
        if ((s & 1) == 0 && v >= NW / 2) issue_pair(s);
        if (s + 1 < S_TILE) read_half((2 * s + 3) % 3, nslot, 1);
        mfma_half((2 * s + 1) % 3, acc, 1, af);
        c_slot = nslot;
    };
    static_assert(FR_AHEAD == FR_SLOTS - 1 && (FR_SLOTS & (FR_SLOTS - 1)) == 0 && FR_AHEAD >= 7 && S_TILE % 2 == 0 && S_TILE > FR_AHEAD, "the waits and the slot arithmetic assume SLOTS - 1 requests ahead in a power-of-two ring, slices in pairs");

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

    for (int i = tid; i < 4 * C; i += 64 * NW) biasl[i] = i < C ? a.bprime[i] : (i < 3 * C ? a.czr[i - C] : a.ch[i - 3 * C]);
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
    for (int s = 0; s < FR_AHEAD - 1; ++s) request(s, info.rg_first);
    // (the first tile's first wait counts operations that a steady-state tile has behind its requests -- here they are in front:
    // make sure slices 0, 1 and 2 have landed before the loop)
    wait_landed(RPW * (FR_AHEAD - 4));

#pragma unroll 1
    while (tile < tiles) {
        // (the next tile is drawn from a counter in the workspace, not blockIdx + k gridDim: workgroups that start late -- a kernel of
        // the side stream holding their CU -- must not find a whole stride of tiles waiting for them; read behind the ring's barriers)
        if (tid == 0) *reinterpret_cast<volatile unsigned*>(flds + L::NEXT_OFF) = a.tile_ctr ? atomicAdd(a.tile_ctr, 1u) + gridDim.x : (unsigned)(tile + gridDim.x);
        const long m0 = tile * FR_ROWS + 16 * v;                 // the wave's first row
        const long left = a.M - m0;
        const int nv = (int)(left < 0 ? 0 : (left < 16 ? left : 16));      // its valid rows
        const unsigned mrow0 = (unsigned)m0;
        const unsigned node0 = mrow0 / uT;
        const int t0 = (int)(mrow0 - node0 * uT);
        rg_cur = info.rg_first;
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
        // (a generic lambda over a constant j, not a loop: the bodies are too long for hipcc to unroll, and every index must be static)
        auto unit_embed = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
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
        };
        unit_embed(std::integral_constant<int, 0>{});
        unit_embed(std::integral_constant<int, 1>{});
        const long tnext = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(flds + L::NEXT_OFF));
        const FrTileInfo info_next = tile_info(tnext);           // (first used by the requests that reach into the next tile: the last FR_AHEAD slices)
        rg_nxt = info_next.rg_first; has_next = tnext < tiles;
        // ---- reset gate R = sigmoid(h Ur^T + (A_hat x) Gr^T + cr), q = h R: the candidate's A operand --------------------------------
        // (a generic lambda over a constant j, not a loop: the bodies are too long for hipcc to unroll, and every index must be static)
        auto unit_r = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
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
        };
        unit_r(std::integral_constant<int, 0>{});
        unit_r(std::integral_constant<int, 1>{});
        // ---- per 128 columns: update gate Z (kept packed), candidate H~, blend, per-node sums ----------------------------------------
        // (a generic lambda over a constant j, not a loop: the bodies are too long for hipcc to unroll, and every index must be static)
        auto unit_zc = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
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
                // (opaque: left alone, hipcc keeps the WIDENED h of the reset-gate epilogue alive until here -- 8 registers per round, spilled)
                u32x4_t hraw = __builtin_bit_cast(u32x4_t, hA[4 * j + q]);
                asm volatile("" : "+v"(hraw));
                const V8 hv = f_widen8(hraw);
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
                unsigned sm = smask, em = emask, am = amask;
                asm volatile("" : "+s"(sm), "+s"(em), "+s"(am));
                const int ohv = eg.lane < 32 ? (32 * q + eg.lr) * 4 : 0x7ffffff0;
                float csum = 0.f;
                int ohs = 0;
#pragma unroll
                for (int r8 = 0; r8 < 16; r8 += 8) {             // (eight rows at a time: the LDS reads first, then the serial chain)
                    float cv[8];
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) cv[rr] = imgw[(r8 + rr) * 32 + (eg.lr ^ (4 * fr_par(r8 + rr)))];
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        float keep;
                        asm volatile("s_bitcmp1_b32 %1, %2\n\ts_cselect_b32 %0, 0, 1.0" : "=s"(keep) : "s"(sm), "n"(r8 + rr) : "scc");
                        csum = fmaf(csum, keep, cv[rr]);
                        if ((em >> (r8 + rr)) & 1u) {
                            const bool part = (am >> (r8 + rr)) & 1u;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(csum), sOH, ohv, part ? 0x7ffffff0 : ohs, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(csum, sOH, ohv, part ? ohs : 0x7ffffff0, 0);
                            ohs += C * 4;
                        }
                    }
                }
            }
            FT_MARK(12 + 4 * j);
        };
        unit_zc(std::integral_constant<int, 0>{});
        unit_zc(std::integral_constant<int, 1>{});
        first_tile = false;
        tile = tnext;
    }
    // (the ring holds no request any more: the producer stopped with the last tile's last slice, which has been consumed)
#undef FT_MARK
}

long* fused_trace_buffer(int which, long tiles);
int fused_cus();

bool fused_forward_rows_ok(int C, int F, int T) { return C == FR_C && (F == 64 || F == 32) && T <= 16; }

int launch_fused_forward_rows(const FusedFwdArgs& a_, int C, int F, int waves, hipStream_t st) {
    REGT_CHECK_ARG(a_.M > 0 && a_.T > 0, "fused forward: empty problem");
    REGT_CHECK_ARG(fused_forward_rows_ok(C, F, a_.T), "fused forward (row form): built for C = 256, F = 32 or 64, T <= 16 (got C = %d, F = %d, T = %d)", C, F, a_.T);
    FusedFwdArgs a = a_;
    REGT_CHECK_ARG(a.M % a.T == 0, "fused forward: M = %ld rows are no whole number of T = %d periods", a.M, a.T);
    a.nodes = a.M / a.T;
    a.pmask = 0;
    for (int rr = 0; rr < 64; rr += a.T) a.pmask |= 1ull << rr;
    const int nw = waves == 4 ? 4 : 8;
    const long tiles = (a.M + 16 * nw - 1) / (16 * nw);
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
    static bool attr_done = false;
    if (!attr_done) {
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_rows_kernel<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedRowsLds<8>::BYTES));
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_rows_kernel<32, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedRowsLds<8>::BYTES));
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_rows_kernel<64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedRowsLds<4>::BYTES));
        REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_fwd_rows_kernel<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedRowsLds<4>::BYTES));
        attr_done = true;
    }
    // persistent: 8 waves per CU (one workgroup of eight or two of four)
    const long slots = (long)fused_cus() * (8 / nw);
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    if (a.tile_ctr) REGT_CHECK_HIP(hipMemsetAsync(a.tile_ctr, 0, sizeof(unsigned), st));
    if (nw == 8) {
        if (F == 64) hipLaunchKernelGGL((fused_fwd_rows_kernel<64, 8>), dim3(grid), dim3(512), FusedRowsLds<8>::BYTES, st, a);
        else hipLaunchKernelGGL((fused_fwd_rows_kernel<32, 8>), dim3(grid), dim3(512), FusedRowsLds<8>::BYTES, st, a);
    } else {
        if (F == 64) hipLaunchKernelGGL((fused_fwd_rows_kernel<64, 4>), dim3(grid), dim3(256), FusedRowsLds<4>::BYTES, st, a);
        else hipLaunchKernelGGL((fused_fwd_rows_kernel<32, 4>), dim3(grid), dim3(256), FusedRowsLds<4>::BYTES, st, a);
    }
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
