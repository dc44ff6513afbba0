// Fused data gradients of the RegT-GCN cell for the bf16 arithmetic, second form: a wave owns 16 WHOLE rows, the weights stream
// through LDS -- the backward counterpart of fused_rows.hip (same ring, same A-operands-in-registers scheme).
//
// Same op sites and arithmetic as fused_bwd_kernel (fused.hip; the transposes of models/utils.py:168-188), to the bit except the
// attention gradient's row dots (another fixed summation order over a row's 256 columns; tests allow 1e-4 of its scale, as for
// fused_bwd_kernel).  With g = p_t dOH[node]:
//   phase A (element-wise, the lane's row and 8 consecutive columns of each 32-column piece):
//           dhp = g (1 - Z)(1 - H~^2) -> global + A operand of phase B;  dzp = g (h - H~) Z (1 - Z) -> global;  <dOH, H'> per row
//   phase B per 128 columns:  dq = dhp Uh2;  drp = dq h R (1 - R) -> global + A operand of phase D;  dh = dq R + g Z (bf16, registers)
//   phase D per 128 columns:  ds = (dh + drp Ur2 + dzp Uz2) act'(h) -> global   (dzp read back from the lane's own stores of phase A)
// A workgroup is 8 waves x 16 rows = 128 rows; the three transposed C x C weight blocks (384 KB in MFMA fragment order) are copied once
// per workgroup and tile into a 16-slot ring of 8 KB slices by LDS-DMA (one instruction per wave and slice, fifteen slices ahead) and
// read by all eight waves; fused_bwd_kernel pulls them per 64-row workgroup straight into registers.  Z and h are read a second time
// in the epilogues of phase B (kept in registers they cost 64 and the kernel spills); the h > 0 bits for phase D take two registers.
#include <type_traits>

#include "fused_common.h"

namespace regt {

namespace {

constexpr int FB_SLICE_B = 8192;         // 32 k x 128 columns of bf16 in fragment order
constexpr int FB_IMG_B = 2048;           // a wave's epilogue image: 16 rows x 32 columns fp32
constexpr int FB_C = 256, FB_ROWS = 128, FB_SLOTS = 16, FB_AHEAD = 15;
constexpr int FB_S_TILE = 48;            // slices of a tile: Uh j=0, Uh j=1, (Ur, Uz) j=0, (Ur, Uz) j=1 -- 8 each

struct FusedRowsBwdLds {
    static constexpr int IMG_OFF = 0;
    static constexpr int RING_OFF = IMG_OFF + 8 * FB_IMG_B;
    static constexpr int NEXT_OFF = RING_OFF + FB_SLOTS * FB_SLICE_B;
    static constexpr int BYTES = NEXT_OFF + 16;
};

typedef __attribute__((address_space(3))) void fb_lds_void;

__device__ __forceinline__ constexpr int fb_par(int r) { return ((r >> 1) & 3) | (r & 4); }   // image swizzle, as fused_rows.hip

struct FusedRowsBwdArgs {
    FusedBwdArgs b;
    const char* wbase;                   // the three weight blocks as 32-bit offsets from one base
    unsigned o_uh, o_uz, o_ur;
    long nodes;
};

}  // namespace

__global__ __launch_bounds__(512, 2) void fused_bwd_rows_kernel(FusedRowsBwdArgs aa) {
    const FusedBwdArgs& a = aa.b;
    constexpr int C = FB_C, S_TILE = FB_S_TILE, RPW = 1;
    using L = FusedRowsBwdLds;
    extern __shared__ __attribute__((aligned(16))) char flds[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);       // the wave: rows 16 v .. 16 v + 15 of the tile
    const unsigned uT = (unsigned)a.T;
    const long tiles = (a.M + FB_ROWS - 1) / FB_ROWS;
    const unsigned ring_lds = (unsigned)(size_t)(fb_lds_void*)(flds + L::RING_OFF);
    float* imgw = reinterpret_cast<float*>(flds + L::IMG_OFF + v * FB_IMG_B);
    unsigned* nextw = reinterpret_cast<unsigned*>(flds + L::NEXT_OFF);
#define FB_MARK(i) do { if (a.trace && tid == 0) a.trace[(long)FT_TRACE_SLOTS * tile + (i)] = (long)__builtin_amdgcn_s_memtime(); } while (0)

    // ---- the ring (see fused_rows.hip): slice s of a tile = k block s % 8 (32 k) of segment s / 8 ------------------------------------
    const int dma_voff = lane * 16;
    unsigned voff_c = (unsigned)(((v >> 1) * (C / 16) + (v & 1)) * 1024);
    unsigned p_slot = 0;
    auto request = [&](int s) {
        const int seg = s >> 3, kb = s & 7;
        const unsigned mo = seg < 2 ? aa.o_uh : ((seg & 1) ? aa.o_uz : aa.o_ur);
        const int j = seg < 2 ? seg : (seg - 2) >> 1;
        const char* src = aa.wbase + (mo + voff_c + (unsigned)((4 * j * (C / 16) + 2 * kb) * 1024));
        const unsigned m0v = ring_lds + p_slot * FB_SLICE_B + v * 1024;
        // (s_nop 3: see fused_rows.hip -- an SGPR operand restored by v_readlane right in front of the statement)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(dma_voff), "s"(src) : "memory");
        p_slot = (p_slot + 1) & (FB_SLOTS - 1);
    };
    unsigned c_slot = 0;
    const int bf_lane = 1024 * (g >> 1) + 512 * (g & 1) + 16 * r;
    bool has_next = false, first_tile = true;
    // counted vector-memory operations behind slice t (fewer than issued is safe: a stricter wait): the epilogue of the unit that
    // ends with it -- B: R, Z, h, dOH loads and the drp store of four rounds, behind B1 and D0 also the eight dzp loads of the next unit; D: four
    // stores; behind the tile's last slice also the next tile's prefetch and its phase A (capped: vmcnt is a 6-bit count)
    auto vm_epi = [&](int t) {
        if (t == 7) return 24;
        if (t == 15) return 32;
        if (t == 31) return 12;
        if (t == S_TILE - 1) return 40;
        return 0;
    };
    auto vm_after = [&](int s) {
        int n = 0;
#pragma unroll
        for (int t = s - (FB_AHEAD - 3); t < s; ++t) n += vm_epi(t < 0 ? t + S_TILE : t);
        return n;
    };
    auto wait_landed = [&](int n) {
        switch (n > 63 ? 63 : n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")\n\ts_barrier" ::: "memory"); break;
            W_(11) W_(12) W_(13) W_(14) W_(15) W_(16) W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31) W_(32) W_(33) W_(34) W_(35) W_(36) W_(37) W_(38) W_(39) W_(40) W_(41) W_(42) W_(43) W_(44) W_(45) W_(46) W_(47) W_(48) W_(49) W_(50) W_(51) W_(52) W_(53) W_(54) W_(55) W_(56) W_(57) W_(58) W_(59) W_(60) W_(61) W_(62) W_(63)
#undef W_
            default: asm volatile("s_waitcnt vmcnt(11)\n\ts_barrier" ::: "memory"); break;
        }
    };
    bf16x8 bq0[4], bq1[4], bq2[4];
    auto read_half = [&](int k, unsigned slot, int hf) {
        const char* sl = flds + L::RING_OFF + slot * FB_SLICE_B + bf_lane + 4096 * hf;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(sl + 2048 * (cb >> 1) + 256 * (cb & 1));
            if (k == 0) bq0[cb] = x; else if (k == 1) bq1[cb] = x; else bq2[cb] = x;
        }
    };
    auto mfma_half = [&](int k, f32x4 (&acc)[8], int hf, const bf16x8& af) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const bf16x8 x = k == 0 ? bq0[cb] : (k == 1 ? bq1[cb] : bq2[cb]);
            if (hf == 0) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, x, acc[cb], 0, 0, 0);
            else acc[4 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, x, acc[4 + cb], 0, 0, 0);
        }
    };
    // one barrier per two slices; requests at two points of the even step (fused_rows.hip: `consume`)
    auto consume = [&](int s, f32x4 (&acc)[8], const bf16x8& af) {
        if ((s & 1) == 0) {
            if (s + FB_AHEAD < S_TILE) {
                if (first_tile && vm_after(s) > 0 && s < FB_AHEAD) wait_landed(RPW * (FB_AHEAD - 4));
                else wait_landed(RPW * (FB_AHEAD - 4) + vm_after(s));
            } else if (has_next) {
                wait_landed(RPW * (FB_AHEAD - 4) + vm_after(s));
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        auto issue_pair = [&](int se) {
            if (se + FB_AHEAD < S_TILE) { request(se + FB_AHEAD - 1); request(se + FB_AHEAD); }
            else if (has_next) { request(se + FB_AHEAD - 1 - S_TILE); request(se + FB_AHEAD - S_TILE); }
        };
        if ((s & 1) == 0 && v < 4) issue_pair(s);
        const unsigned nslot = (c_slot + 1) & (FB_SLOTS - 1);
        // (no reads across a unit's end: the 32 registers of the next slice's fragments are what the epilogues lack -- with them the
        // kernel spills ~200 dwords; a unit's first step reads its own two halves, ~200 cycles in the open, four times per tile)
        const bool ufirst = s == 0 || s == 8 || s == 16 || s == 32, ulast = s == 7 || s == 15 || s == 31 || s == S_TILE - 1;
        if (ufirst) { read_half((2 * s) % 3, c_slot, 0); read_half((2 * s + 1) % 3, c_slot, 1); }
        if (!ulast) read_half((2 * s + 2) % 3, nslot, 0);
        mfma_half((2 * s) % 3, acc, 0, af);
        if ((s & 1) == 0 && v >= 4) issue_pair(s);
        if (!ulast) read_half((2 * s + 3) % 3, nslot, 1);
        mfma_half((2 * s + 1) % 3, acc, 1, af);
        c_slot = nslot;
    };

    // ---- epilogue geometry (fused_rows.hip) ------------------------------------------------------------------------------------------
    struct EpiGeo { int st_row, st_col[2][2], e_lo, e_hi, ro, rzo, lane; };
    auto epi_geo = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int rr_ = l & 15, gg = l >> 4;
        EpiGeo e;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) e.st_col[b][ih] = 4 * ((4 * b + (rr_ >> 2)) ^ (6 * (gg & 1) + ih)) + (rr_ & 3);
        e.st_row = 4 * gg * 32;
        e.e_lo = rr_ * 32 + 4 * ((2 * gg) ^ fb_par(rr_));
        e.e_hi = rr_ * 32 + 4 * ((2 * gg + 1) ^ fb_par(rr_));
        e.ro = rr_ * C * 2 + gg * 16;                            // the lane's 16 bytes in a row of an (M x C) bf16 array
        e.rzo = rr_ * C * 4 + gg * 16;                           //   ... of an (M x 2C) array
        e.lane = l;
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
    auto zero8 = [&](f32x4 (&acc)[8]) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- the next tile's Z, h, H~ of the first 128 columns (pieces 0..3), requested before the current tile's last stores --------------
    u32x4_t zp[4], hp_[4], tp[4];
    auto request_tile0 = [&](long tile) {
        const long m0 = tile * FB_ROWS + 16 * v;
        const long left = a.M - m0;
        const int nv = (int)(left < 0 ? 0 : (left < 16 ? left : 16));
        const __amdgpu_buffer_rsrc_t sZR = f_rsrc(reinterpret_cast<const char*>(a.ZR) + m0 * C * 4, (long)nv * C * 4);
        const __amdgpu_buffer_rsrc_t sH = f_rsrc(reinterpret_cast<const char*>(a.h) + m0 * C * 2, (long)nv * C * 2);
        const __amdgpu_buffer_rsrc_t sHt = f_rsrc(reinterpret_cast<const char*>(a.Ht) + m0 * C * 2, (long)nv * C * 2);
        int l = lane;
        asm volatile("" : "+v"(l));
        const int ro = (l & 15) * C * 2 + (l >> 4) * 16, rzo = (l & 15) * C * 4 + (l >> 4) * 16;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            zp[pc] = __builtin_amdgcn_raw_buffer_load_b128(sZR, rzo + 64 * pc, 0, 0);
            hp_[pc] = __builtin_amdgcn_raw_buffer_load_b128(sH, ro + 64 * pc, 0, 0);
            tp[pc] = __builtin_amdgcn_raw_buffer_load_b128(sHt, ro + 64 * pc, 0, 0);
        }
    };

    long tile = blockIdx.x;
    request_tile0(tile);
#pragma unroll
    for (int s = 0; s < FB_AHEAD - 1; ++s) request(s);
    wait_landed(RPW * (FB_AHEAD - 4));
    int tpar = 0;

#pragma unroll 1
    while (tile < tiles) {
        if (tid == 0) nextw[tpar] = a.tile_ctr ? atomicAdd(a.tile_ctr, 1u) + gridDim.x : (unsigned)(tile + gridDim.x);
        const long m0 = tile * FB_ROWS + 16 * v;                 // the wave's first row
        const long left = a.M - m0;
        const int nv = (int)(left < 0 ? 0 : (left < 16 ? left : 16));
        asm volatile("" : "+s"(voff_c));
        FB_MARK(0);
        const __amdgpu_buffer_rsrc_t sZR = f_rsrc(reinterpret_cast<const char*>(a.ZR) + m0 * C * 4, (long)nv * C * 4);
        const __amdgpu_buffer_rsrc_t sH = f_rsrc(reinterpret_cast<const char*>(a.h) + m0 * C * 2, (long)nv * C * 2);
        const __amdgpu_buffer_rsrc_t sHt = f_rsrc(reinterpret_cast<const char*>(a.Ht) + m0 * C * 2, (long)nv * C * 2);
        const __amdgpu_buffer_rsrc_t sdhp = f_rsrc(reinterpret_cast<char*>(a.dhp) + m0 * C * 2, (long)nv * C * 2);
        const __amdgpu_buffer_rsrc_t sdzr = f_rsrc(reinterpret_cast<char*>(a.dzr) + m0 * C * 4, (long)nv * C * 4);
        const __amdgpu_buffer_rsrc_t sdh = f_rsrc(reinterpret_cast<char*>(a.dh) + m0 * C * 2, (long)nv * C * 2);
        // the lane's row: its period's attention probability and its node's row of dOH
        const unsigned mrow = (unsigned)m0 + (unsigned)r, nd = mrow / uT;
        const bool rok = r < nv;
        const float pt = a.probs[rok ? mrow - nd * uT : 0];
        const __amdgpu_buffer_rsrc_t sD = f_rsrc(a.dOH, aa.nodes * C * 4);
        const int dof = rok ? (int)(nd * (unsigned)(C * 4)) + g * 32 : 0x7ffffff0;      // (bytes; rows past the end read zeros)
        auto ldd8 = [&](int pc) {
            const u32x4_t lo = __builtin_amdgcn_raw_buffer_load_b128(sD, dof, 128 * pc, 0), hi = __builtin_amdgcn_raw_buffer_load_b128(sD, dof, 128 * pc + 16, 0);
            return V8{{__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w),
                       __uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w)}};
        };

        // ---- phase A ------------------------------------------------------------------------------------------------------------------
        // (piece by piece, the loads four pieces ahead: with all of a tile's Z, h, H~ and dOH in registers at once -- and h kept for
        // phase B -- the kernel needed ~60 registers more than it has; Z, h are read a second time in phase B's epilogues instead)
        bf16x8 dhpA[8];                                          // A operands of phase B
        u32x4_t hk[8];                                           // h as read: the epilogues of phase B need it again
        float rdot = 0.f;
        {
            const EpiGeo eg = epi_geo();
            u32x4_t zk[8], tk[8];
            V8 dd[8];
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) { zk[pc] = zp[pc]; hk[pc] = hp_[pc]; tk[pc] = tp[pc]; dd[pc] = ldd8(pc); }
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                if (pc + 4 < 8) {
                    zk[pc + 4] = __builtin_amdgcn_raw_buffer_load_b128(sZR, eg.rzo + 64 * (pc + 4), 0, 0);
                    hk[pc + 4] = __builtin_amdgcn_raw_buffer_load_b128(sH, eg.ro + 64 * (pc + 4), 0, 0);
                    tk[pc + 4] = __builtin_amdgcn_raw_buffer_load_b128(sHt, eg.ro + 64 * (pc + 4), 0, 0);
                    dd[pc + 4] = ldd8(pc + 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                const V8 z = f_widen8(zk[pc]), h = f_widen8(hk[pc]), ht = f_widen8(tk[pc]);
                V8 dhp, dzp;
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float gg = __fmul_rn(pt, dd[pc].v[i]);
                    dhp.v[i] = cb_dhp(gg, z.v[i], ht.v[i]);
                    dzp.v[i] = cb_dzp(gg, h.v[i], ht.v[i], z.v[i]);
                    dot += dd[pc].v[i] * (z.v[i] * h.v[i] + (1.0f - z.v[i]) * ht.v[i]);
                }
                rdot += dot;
                asm volatile("" : "+v"(rdot));
                const u32x4_t o_p = f_pack8(dhp), o_z = f_pack8(dzp);
                dhpA[pc] = __builtin_bit_cast(bf16x8, o_p);
                __builtin_amdgcn_raw_buffer_store_b128(o_p, sdhp, eg.ro + 64 * pc, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o_z, sdzr, eg.rzo + 64 * pc, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the row's dot: the four lanes of a row (one per 8-column group) hold 64 columns each
            float s = rdot;
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const __amdgpu_buffer_rsrc_t sdot = f_rsrc(a.rowdot + m0, (long)nv * 4);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), sdot, eg.lane < 16 ? eg.lane * 4 : 0x7ffffff0, 0, 0);
        }
        FB_MARK(1);

        // ---- phase B: dq = dhp Uh2 per 128 columns; drp -> global + A operand of phase D, dh -> registers (bf16) ------------------------
        bf16x8 drpA[8];
        u32x4_t dhk[8];
        unsigned hpos0 = 0, hpos1 = 0;                           // bit 8 q + i of piece 4 j + q: h > 0
        long tnext = 0;
        auto unit_b = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            f32x4 acc[8];
            zero8(acc);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) consume(8 * j + kb, acc, dhpA[kb]);
            FB_MARK(2 + 2 * j);
            const EpiGeo eg = epi_geo();
            unsigned hbits = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                const int pc = 4 * j + q;
                const u32x4_t rraw = __builtin_amdgcn_raw_buffer_load_b128(sZR, eg.rzo + C * 2 + 64 * pc, 0, 0);    // R: first touch
                const u32x4_t zraw = __builtin_amdgcn_raw_buffer_load_b128(sZR, eg.rzo + 64 * pc, 0, 0);            // Z, h: second read
                const u32x4_t hraw = hk[pc];
                const V8 d = ldd8(pc);
                stage(eg, acc[2 * q], acc[2 * q + 1]);
                const V8 vv = img8(eg);
                const V8 h = f_widen8(hraw), Z = f_widen8(zraw), R = f_widen8(rraw);
                V8 drp, dh;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    drp.v[i] = cb_drp(vv.v[i], h.v[i], R.v[i]);
                    dh.v[i] = cb_dh(vv.v[i], R.v[i], pt, d.v[i], Z.v[i]);
                    hbits |= (h.v[i] > 0.f ? 1u : 0u) << (8 * q + i);
                }
                const u32x4_t pr = f_pack8(drp);
                drpA[pc] = __builtin_bit_cast(bf16x8, pr);
                __builtin_amdgcn_raw_buffer_store_b128(pr, sdzr, eg.rzo + C * 2 + 64 * pc, 0, 0);
                // (opaque: left alone, hipcc sinks this arithmetic into phase D's epilogue, where dh is used -- and keeps dq, R, Z and
                // dOH of every round alive until then: 700 bytes of spills)
                u32x4_t pk = f_pack8(dh);
                asm volatile("" : "+v"(pk));
                dhk[pc] = pk;
            }
            if (j == 0) hpos0 = hbits; else hpos1 = hbits;
            FB_MARK(3 + 2 * j);
        };
        unit_b(std::integral_constant<int, 0>{});
        tnext = __builtin_amdgcn_readfirstlane(nextw[tpar]);
        has_next = tnext < tiles;
        unit_b(std::integral_constant<int, 1>{});

        // ---- phase D: ds = (dh + drp Ur2 + dzp Uz2) act'(h) per 128 columns -------------------------------------------------------------
        bf16x8 dzpA[8];
        {
            const EpiGeo eg = epi_geo();
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) dzpA[pc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(sdzr, eg.rzo + 64 * pc, 0, 0));
        }
        auto unit_d = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            f32x4 acc[8];
            zero8(acc);
            // dzp back from the lane's own stores of phase A, once per unit under the drp product (kept across both units it is 32
            // registers in the epilogue of the first -- the kernel then spills)
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) consume(16 + 16 * j + kb, acc, drpA[kb]);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) consume(16 + 16 * j + 8 + kb, acc, dzpA[kb]);
            FB_MARK(6 + 2 * j);
            const EpiGeo eg = epi_geo();
            if (j == 1) request_tile0(tnext < tiles ? tnext : tiles - 1);      // the next tile's first operands, before this tile's last stores
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                const int pc = 4 * j + q;
                stage(eg, acc[2 * q], acc[2 * q + 1]);
                const V8 vv = img8(eg);
                const V8 d = f_widen8(dhk[pc]);
                V8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool pos = ((j ? hpos1 : hpos0) >> (8 * q + i)) & 1u;
                    o.v[i] = cb_ds(d.v[i], vv.v[i], (a.act_lrelu && !pos) ? a.slope : 1.0f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(f_pack8(o), sdh, eg.ro + 64 * pc, 0, 0);
            }
            FB_MARK(7 + 2 * j);
        };
        unit_d(std::integral_constant<int, 0>{});
        unit_d(std::integral_constant<int, 1>{});
        first_tile = false;
        tile = tnext;
        tpar ^= 1;
    }
#undef FB_MARK
}

long* fused_trace_buffer(int which, long tiles);
int fused_cus();

bool fused_backward_rows_ok(int C, int T) { return C == FB_C && T > 0; }

int launch_fused_backward_rows(const FusedBwdArgs& a_, int C, hipStream_t st) {
    REGT_CHECK_ARG(a_.M > 0 && a_.T > 0 && a_.M % a_.T == 0, "fused backward (row form): empty problem");
    REGT_CHECK_ARG(fused_backward_rows_ok(C, a_.T), "fused backward (row form): built for C = 256 (got C = %d)", C);
    FusedRowsBwdArgs aa{};
    aa.b = a_;
    const long tiles = (a_.M + FB_ROWS - 1) / FB_ROWS;
    REGT_CHECK_ARG(a_.M < (1L << 31) / 4, "fused backward (row form): too many rows");
    aa.nodes = a_.M / a_.T;
    REGT_CHECK_ARG(aa.nodes * C * 4 < 0x7ffffff0L, "fused backward (row form): dOH larger than a buffer descriptor reaches");
    aa.b.trace = fused_trace_buffer(2, tiles);
    {
        const char* ptrs[3] = {(const char*)a_.UhTf, (const char*)a_.UzTf, (const char*)a_.UrTf};
        const char* base = ptrs[0];
        for (int i = 1; i < 3; ++i) base = ptrs[i] < base ? ptrs[i] : base;
        unsigned* offs[3] = {&aa.o_uh, &aa.o_uz, &aa.o_ur};
        for (int i = 0; i < 3; ++i) {
            const long o = ptrs[i] - base;
            REGT_CHECK_ARG(o >= 0 && o < (1L << 30), "fused backward (row form): weight blocks too far apart");
            *offs[i] = (unsigned)o;
        }
        aa.wbase = base;
    }
    using L = FusedRowsBwdLds;
    static bool attr_done = false;
    if (const int rc = set_lds_once(&fused_bwd_rows_kernel, L::BYTES, &attr_done)) return rc;
    const long slots = fused_cus();
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    if (aa.b.tile_ctr) REGT_CHECK_HIP(hipMemsetAsync(aa.b.tile_ctr, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(fused_bwd_rows_kernel, dim3(grid), dim3(512), L::BYTES, st, aa);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
