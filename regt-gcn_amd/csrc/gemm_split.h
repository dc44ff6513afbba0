// fp32 GEMM on the bf16 matrix pipe: every fp32 operand element is split exactly into three bf16 pieces
// (x = x1 + x2 + x3, 8 significant bits each -- bf16 has the fp32 exponent range, so the split never
// under/overflows where fp32 does not) and the product is expanded into the six partial products whose
// weight is >= 2^-16 of the leading one:
//     a*b  ~=  a3 b1 + a2 b2 + a1 b3 + a2 b1 + a1 b2 + a1 b1           (dropped: a2 b3 + a3 b2 + a3 b3 <= 3 * 2^-24 |a b|)
// Each bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the result carries the same
// order of rounding error as an fp32 FMA chain (the dropped terms are one fp32 ulp of the product), while
// v_mfma_f32_32x32x16_bf16 retires 16x the k-extent of v_mfma_f32_32x32x2_f32 in half the cycles: 6 bf16 MFMAs
// replace 8 fp32 MFMAs per 16 k and take 192 instead of 512 matrix-pipe cycles.
//
// The split is opt-in (regt_set_gemm_mode / REGT_GEMM_MODE=bf16x3).  The core itself is the default GEMM core of every
// arithmetic since round 2: NP = 0 keeps the operands fp32 and multiplies on the fp32 MFMA (same results as gemm_fast.h's
// two-workgroup core), NP = 1 rounds them to bf16 (REGT_GEMM_MODE=bf16).  Round-2 additions, further down: the compact LDS
// layout with a half-tile epilogue (three workgroups per CU), slab descriptors in scalar registers (run_u / run_u1),
// branch-free epilogue bodies for full tiles (vec_body_halves / vec8_body_halves).
//
// Same 128x128 tile, 2x2 waves, row map, iteration table, buffer-descriptor loads and LDS-staged epilogue as
// FastCore<true, REGION> (B given as [N][K], k contiguous).  What changes is the K loop:
//   * LDS holds bf16 planes: stage h (h = 0, 1) = the h-th 16-k half of the current 32-k slab, per operand three
//     planes of 128 rows x 32 B, k-group bit swizzled by row bit 3 (sp_off: conflict-free ds_read_b128 AND
//     ds_write_b64 without padding).  2 stages x 24,576 B (NP = 3), three workgroups per CU with the compact layout;
//   * a thread owns 2 float4 of A and 2 of B per half (row = tid/4 (+64), k-quad = tid%4 (+4 for the second half)),
//     splits them while storing (v_cvt_pk_bf16_f32 + packed fp32 subtract: 9 VALU per element pair);
//   * the two halves double-buffer each other:  compute(h0) | barrier | store next h0 | compute(h1) | barrier |
//     store next h1, so the conversion VALU work of one half overlaps the MFMAs of the other;
//   * a half's registers are refilled with the same half of the slab after next as soon as they have been stored,
//     i.e. every global load has two compute blocks of distance.
#pragma once
#include <type_traits>
#include "gemm_fast.h"

namespace regt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SP_ROW_B = 32;                   // bytes of one (row, plane) of a 16-k half slab: two 16-B k-groups, no pad
constexpr int SP_PLANE_B = 128 * SP_ROW_B;     // 4096
// NP = planes per operand: 3 = exact 3-way split (fp32-level accuracy, six partial products), 1 = plain bf16 operands
// (round-to-nearest-even at staging, ONE product: the "bf16 GEMM inputs, fp32 accumulate" arithmetic of BASELINE configs[4]).
// NP = 0: no split at all -- the operands stay fp32 (rows of 16 k = 64 B) and the products run on v_mfma_f32_32x32x2_f32:
// the fp32 arithmetic of FastCore in this core's compact LDS layout and half-slab schedule, i.e. THREE workgroups per CU
// instead of two (a third workgroup's K loop fills the matrix pipe while another one is in its epilogue).
constexpr int CP_ROW_B = 64;                   // bytes of one fp32 row of a 16-k half slab: four 16-B k-quads, no pad
// k-quad g of row r: the quad index is XORed with row bits 2-3, so that the 16 rows a ds_read_b128 / ds_write_b128 service
// group touches (4 consecutive rows x 4 values of bits 2-3) fall on 16 distinct 16-B bank quads
__device__ __forceinline__ constexpr int cp_off(int r, int g) { return r * CP_ROW_B + ((g ^ ((r >> 2) & 3)) << 4); }
template <int NP>
struct SplitGeom {
    static constexpr int OPER_B = NP == 0 ? 128 * CP_ROW_B : NP * SP_PLANE_B;       // 8192 / 12288 / 4096: the planes of one operand
    static constexpr int STAGE_B = 2 * OPER_B;           // A planes, then B planes
    // LDS of a workgroup that stages its epilogue in two 64-row halves (for_each_vec_halves below): the two bf16 stages (or
    // the half-tile epilogue image, whichever is larger), then the iteration table -- 54,528 B (NP = 3: THREE workgroups fit
    // a CU's 163,840 B, which the 166-VGPR kernels also allow) / 39,168 B (NP = 1).
    static constexpr int EPI_HALF_B = 64 * G_LDS_KROW * 4;
    static constexpr int TABLE_OFF_B = 2 * STAGE_B > EPI_HALF_B ? 2 * STAGE_B : EPI_HALF_B;
    static constexpr int LDS_BYTES = TABLE_OFF_B + G_TABLE_BYTES;
    static_assert(2 * STAGE_B <= 2 * G_STAGE * 4, "split stages fit the fp32 core's LDS footprint (table offset, epilogue image)");
};
constexpr int SP_LDS_BYTES = SplitGeom<3>::LDS_BYTES;
// Byte offset of k-group g (8 k = 16 B) of row r inside a plane.  Rows are 32 B apart without padding; the k-group bit
// is flipped for rows with bit 3 set, which makes both access patterns bank-conflict free (MI355X_MICROARCH.md, LDS):
// ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) -- their 16 rows then fall
// on 16 distinct 16-B bank quads; ds_write_b64 in 16 consecutive lanes = 4 rows x 4 k-quads = 32 distinct banks.
__device__ __forceinline__ constexpr int sp_off(int r, int g) { return r * SP_ROW_B + ((g ^ ((r >> 3) & 1)) << 4); }

// F::EPI_AUX_BYTES if the functor declares it, else 96 (the budget of the 168-VGPR, three-workgroup kernels)
template <class F> constexpr auto epi_aux_budget(int) -> decltype(F::EPI_AUX_BYTES) { return F::EPI_AUX_BYTES; }
template <class F> constexpr int epi_aux_budget(long) { return 96; }

template <bool REGION, int NP = 3>
struct SplitCore : FastCore<true, REGION> {
    using Base = FastCore<true, REGION>;
    using Geom = SplitGeom<NP>;
    static constexpr int OPER_B = Geom::OPER_B, STAGE_B = Geom::STAGE_B;
    using Srds = typename Base::Srds;
    using Base::S;
    using Base::rm;
    using Base::n0;
    using Base::N;
    using Base::lds;
    using Base::table;
    using Base::nit;
    using Base::tid;
    using Base::lane;
    using Base::wr;
    using Base::wc;
    int sreg[2];   // region of the thread's two staging rows (REGION only)

    __device__ __forceinline__ SplitCore(const GemmSegs& s, RowMap r, int n0_, int N_, float* lds_, bool compact = false)
        : Base(s, r, n0_, N_, lds_) {
        // compact: the kernel allocated SP_LDS_BYTES only (table right behind the two bf16 stages, half-tile epilogue)
        if (compact) table = reinterpret_cast<ItDesc*>(reinterpret_cast<char*>(lds) + Geom::TABLE_OFF_B);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = (tid >> 2) + 64 * j;
            sreg[j] = 0;
            if (REGION && rl < rm.nvalid) sreg[j] = S.node_region[rm.grow(rl) / S.row_div];
        }
    }

    // slot i of a thread: staging row (tid/4) + 64*(i/2), k-quad (tid%4) + 4*(i%2) of the 32-k slab -> half i%2
    __device__ __forceinline__ float4 sload_a(const Srds& d, int i) const {
        const int rl = (tid >> 2) + 64 * (i >> 1);
        const int k = d.k0 + 4 * ((tid & 3) + 4 * (i & 1));
        bool ok = rl < rm.nvalid && k < d.K;
        if (REGION) ok = ok && (d.region < 0 || sreg[i >> 1] == d.region);
        return Base::srd_load(d.a, ok ? 4u * (unsigned)(rl * (int)rm.mul * d.lda + k) : Base::SRD_OOB);
    }
    __device__ __forceinline__ float4 sload_b(const Srds& d, int i) const {
        const int nl = (tid >> 2) + 64 * (i >> 1);
        const int k = d.k0 + 4 * ((tid & 3) + 4 * (i & 1));
        const bool ok = n0 + nl < N && k < d.K;
        return Base::srd_load(d.b, ok ? 4u * (unsigned)(nl * d.ldb + k) : Base::SRD_OOB);
    }

    // exact NP-way bf16 split of four consecutive-k values (NP = 1: plain round-to-nearest-even), written to the NP
    // planes (8 B each)
    __device__ __forceinline__ static void split_store(char* q, float4 v, int plane_stride = SP_PLANE_B) {
        f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bf16x2 bl = __builtin_convertvector(lo, bf16x2), bh = __builtin_convertvector(hi, bf16x2);
            const unsigned ul = __builtin_bit_cast(unsigned, bl), uh = __builtin_bit_cast(unsigned, bh);
            *reinterpret_cast<uint2*>(q + p * plane_stride) = make_uint2(ul, uh);
            if (p < NP - 1) {
                const f32x2 fl = {__uint_as_float(ul << 16), __uint_as_float(ul & 0xffff0000u)};
                const f32x2 fh = {__uint_as_float(uh << 16), __uint_as_float(uh & 0xffff0000u)};
                lo -= fl;
                hi -= fh;
            }
        }
    }
    // store the thread's slots of half h (slots h and h+2) of one register set
    // `abf` (wave-uniform): ra[h] holds 16 raw bytes = the 8 bf16 of k-group (tid & 1) of row tid >> 1, loaded from an
    // operand its producer already rounded to bf16 (SEG_A_BF16): they go to LDS as they are, one ds_write_b128, no VALU
    template <bool RELU>
    __device__ __forceinline__ void store_half(int h, const float4 (&ra)[4], const float4 (&rb)[4], int abf = 0) const {
        char* st = reinterpret_cast<char*>(lds) + h * STAGE_B;
        if constexpr (NP == 0) {       // fp32 rows as they are: one ds_write_b128 per slot
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = cp_off((tid >> 2) + 64 * j, tid & 3);
                float4 a = ra[h + 2 * j];
                if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                *reinterpret_cast<float4*>(st + off) = a;
                *reinterpret_cast<float4*>(st + OPER_B + off) = rb[h + 2 * j];
            }
            return;
        }
        if (NP == 1 && abf) *reinterpret_cast<float4*>(st + sp_off(tid >> 1, tid & 1)) = ra[h];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = sp_off((tid >> 2) + 64 * j, (tid >> 1) & 1) + (tid & 1) * 8;
            if (!(NP == 1 && abf)) {
                float4 a = ra[h + 2 * j];
                if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                split_store(st + off, a);
            }
            split_store(st + OPER_B + off, rb[h + 2 * j]);
        }
    }
    // (re)load the thread's slots of half h (slots h and h+2) from slab `t`; `live` = false requests nothing (offsets
    // beyond the descriptor range return 0 without touching memory) so that the K loop needs no branch
    __device__ __forceinline__ Srds slab_srds(const ItDesc& t, bool live) const {
        Srds d = Base::make_srds(t);
        if (!live) d.K = 0;
        return d;
    }
    __device__ __forceinline__ void load_half(int h, const Srds& d, float4 (&ra)[4], float4 (&rb)[4]) const {
        if (NP == 1 && d.abf) {      // bf16 rows: ONE 16-byte load per thread = 8 k of row tid >> 1 (128 rows x 2 k-groups)
            const int rl = tid >> 1, k = d.k0 + 16 * h + 8 * (tid & 1);
            const bool ok = rl < rm.nvalid && k < d.K;
            ra[h] = Base::srd_load(d.a, ok ? 2u * (unsigned)(rl * (int)rm.mul * d.lda + k) : Base::SRD_OOB);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) ra[h + 2 * j] = sload_a(d, h + 2 * j);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) rb[h + 2 * j] = sload_b(d, h + 2 * j);
    }
    // ---- scalar slab descriptors ("uniform" path) ---------------------------------------------------------------------
    // VALU instructions are not free next to MFMAs: on this hardware they share the SIMD's vector issue, every vector
    // instruction of any wave of the SIMD takes ~7 cycles away from the matrix pipe (tools/micro/mfma_valu_mix.hip: one
    // VALU per fp32 MFMA costs 10 % of the MFMA rate).  The table path above spends ~28 of them per 16-k half slab (15
    // v_readfirstlane to get a descriptor out of LDS, 64-bit address arithmetic, per-load guards and selects).  Here the
    // walk over (segment, repeat, k0) and the descriptors stay in scalar registers, computed by the scalar unit from
    // the kernel arguments: the base pointer of the tile's rows, num_records = the bytes of the VALID rows (rows past the
    // end return 0 through the range check: no per-load guard), the slab's k offset as the instruction's scalar offset;
    // what is left per slab are four v_mad for the per-thread row offsets.  Requirements (host: uniform_ok in gemm.hip):
    // no region-masked segment, every K a multiple of 32, rows of a tile consecutive (rm.mul == 1).
    // Region-masked segments (REGION kernels): the segment repeats once per DISTINCT region among the tile's rows (sorted
    // list from tile_regions, copied to the unused iteration-table area by uniform_regions()); a row takes part in the
    // repeat of its own region only -- its vector offset is out of range otherwise (one compare + select per row slot
    // and slab).
    struct SegCursor { int s, ri, k0; };
    int nreg_u;                // number of distinct regions in the tile (uniform path of REGION kernels)
    __device__ __forceinline__ const GemmSeg& cseg(int sidx) const { return S.seg[sidx]; }
    __device__ __forceinline__ int seg_count(const GemmSeg& g) const {
        if (REGION && (g.flags & SEG_REGION)) return nreg_u;
        return (g.flags & SEG_REPEAT) ? g.nrep : 1;
    }
    __device__ __forceinline__ void uniform_regions() {
        nreg_u = 0;
        if (REGION) {
            int* red = reinterpret_cast<int*>(lds);
            const int* list = tile_regions<GBM>(S, rm, red, tid);
            __syncthreads();
            nreg_u = __builtin_amdgcn_readfirstlane(red[3]);
            int* keep = reinterpret_cast<int*>(table);
            if (tid < nreg_u) keep[tid] = list[tid];
            __syncthreads();
        }
    }
    __device__ __forceinline__ int total_slabs() const {
        int n = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < S.nseg) n += seg_count(S.seg[q]) * (S.seg[q].K / GBK);
        return n;
    }
    __device__ __forceinline__ void cursor_next(SegCursor& c) const {
        const GemmSeg& g = cseg(c.s);
        c.k0 += GBK;
        if (c.k0 >= g.K) {
            c.k0 = 0;
            if (++c.ri >= seg_count(g)) { c.ri = 0; ++c.s; }
        }
    }
    struct SrdsU { __amdgpu_buffer_rsrc_t a, b; int abf, va[2], vb[2], sa, sb, bks; };   // bks: bytes between 32-column blocks (SEG_B_FRAG)
    __device__ __forceinline__ SrdsU make_u(const SegCursor& c, bool live) const {
        const GemmSeg& g = cseg(live ? c.s : 0);
        const bool rep = (g.flags & SEG_REPEAT) != 0;
        const bool reg = REGION && (g.flags & SEG_REGION) != 0;
        const int region = reg && live ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(table)[c.ri]) : -1;
        SrdsU d;
        d.abf = (g.flags & SEG_A_BF16) ? 1 : 0;
        const int lda_b = (int)g.lda * (d.abf ? 2 : 4), ldb_b = (int)g.ldb * 4;
        const char* ap = reinterpret_cast<const char*>(g.A) + (rep ? (long)c.ri * g.a_rep_stride * 4 : 0) + rm.base * lda_b;
        const long boff = reg ? (long)region * g.b_region_stride : (rep ? (long)c.ri * g.b_region_stride : 0);
        const bool lowb = n0 < g.nsplit;
        d.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ap), 0, live ? rm.nvalid * lda_b : 0, 0x00020000);
        d.bks = 0;
        if (NP == 1 && (g.flags & SEG_B_FRAG)) {
            // weights in fragment order: 1 KB blocks (n / 32, k / 16); the tile's four 32-column blocks start at block row nl / 32
            d.bks = (g.K / 16) * 1024;
            const int nl = lowb ? n0 : n0 - g.nsplit;
            const char* bp = reinterpret_cast<const char*>(lowb ? g.B0 : g.B1) + boff + (long)(nl / 32) * d.bks;
            d.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bp), 0, live ? 4 * d.bks : 0, 0x00020000);
            d.vb[0] = 2 * wc * d.bks + lane * 16;
            d.vb[1] = 0;
            d.sb = (c.k0 / 16) * 1024;
        } else {
        const float* bp = lowb ? g.B0 + boff + (long)n0 * g.ldb : g.B1 + boff + (long)(n0 - g.nsplit) * g.ldb;
        const int blim = (lowb && g.nsplit < N ? g.nsplit : N) - n0, brows = blim < GBN ? blim : GBN;
        d.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bp), 0, live ? brows * ldb_b : 0, 0x00020000);
        d.vb[0] = (tid >> 2) * ldb_b + (tid & 3) * 16;
        d.vb[1] = d.vb[0] + 64 * ldb_b;
        d.sb = c.k0 * 4;
        }
        if (NP == 1 && d.abf) {
            d.va[0] = (tid >> 1) * lda_b + (tid & 1) * 16;
            d.va[1] = 0;
        } else {
            d.va[0] = (tid >> 2) * lda_b + (tid & 3) * 16;
            d.va[1] = d.va[0] + 64 * lda_b;
            if (REGION && reg) {
                d.va[0] = sreg[0] == region ? d.va[0] : (int)Base::SRD_OOB;
                d.va[1] = sreg[1] == region ? d.va[1] : (int)Base::SRD_OOB;
            }
        }
        d.sa = c.k0 * (d.abf ? 2 : 4);
        return d;
    }
    __device__ __forceinline__ void load_half(int h, const SrdsU& d, float4 (&ra)[4], float4 (&rb)[4]) const {
        if (NP == 1 && d.abf) {
            ra[h] = buf_ld4(d.a, d.va[0], d.sa + 32 * h);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) ra[h + 2 * j] = buf_ld4(d.a, d.va[j], d.sa + 64 * h);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) rb[h + 2 * j] = buf_ld4(d.b, d.vb[j], d.sb + 64 * h);
    }
    // the schedule of run_t with scalar descriptors
    template <bool RELU>
    __device__ __forceinline__ void run_u(f32x16 (&acc)[2][2]) const {
        const int nslab = total_slabs();
        if (nslab == 0) return;
        float4 ra[4], rb[4];
        int held;
        SegCursor c{0, 0, 0};
        {
            const SrdsU d = make_u(c, true);
            load_half(0, d, ra, rb);
            load_half(1, d, ra, rb);
            held = d.abf;
        }
        store_half<RELU>(0, ra, rb, held);
        store_half<RELU>(1, ra, rb, held);
        {
            const bool two = nslab > 1;
            if (two) cursor_next(c);
            const SrdsU d = make_u(c, two);
            load_half(0, d, ra, rb);
            load_half(1, d, ra, rb);
            held = d.abf;
        }
        __syncthreads();
        compute(0, acc);
        for (int it = 0; it + 1 < nslab; ++it) {
            const bool live = it + 2 < nslab;
            if (live) cursor_next(c);
            const SrdsU nx = make_u(c, live);
            __syncthreads();
            fused<RELU>(0, 1, nx, ra, rb, acc, held);
            __syncthreads();
            fused<RELU>(1, 0, nx, ra, rb, acc, held);
            held = nx.abf;
        }
        __syncthreads();
        compute(1, acc);
        __syncthreads();
    }

    // ---- GENERATED A operand (fp32 core, NP = 0): the candidate data gradient without a stored left operand ------------------
    // dq = dhp Uh2 with dhp[m, k] = g (1 - Z)(1 - H~^2), g = p_t dOH[node]  (the transposes of models/utils.py:181-188).  Round 1-3:
    // cell_bwd_kernel read Z, h, H~ and wrote dhp, dzp (5 C floats per row) only to hand dhp to this GEMM, which read it back.
    // Here the A slots of a thread are FORMED from Z, H~ and dOH while they are staged -- same rows, same k-quads, same LDS image
    // as run_u -- and the column-tile-0 workgroup of a row tile stores them to dhp on the way (the weight gradients dUh / dGh
    // still need it); the epilogue (EpiDgrad1GenF, gemm.hip) adds dzp and the per-row attention dot, so cell_bwd disappears:
    // 8 C floats per row instead of 11 cross HBM.  Three loads per slot instead of one: 48 instead of 16 operand VGPRs -- this
    // kernel runs two workgroups per CU.
    struct AGen {
        const float* ZR; const float* Ht; const float* dOH; float* dhp;
        int C; unsigned doh_bytes;                 // dOH is (num_nodes, C) fp32: doh_bytes = num_nodes * C * 4 (host-checked < 2^31)
    };
    struct GenRegs { float4 z[4], t[4], d[4]; };   // slot h + 2 j: Z, H~, dOH quads of row (tid >> 2) + 64 j, half h
    struct GenRows { int vz[2], vh[2], vd[2]; float p[2]; };
    struct GenSrd { __amdgpu_buffer_rsrc_t z, t, d, o; };
    __device__ __forceinline__ GenRows gen_rows(const AGen& g) const {       // needs the row table (fill_rowtab + barrier)
        GenRows r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = (tid >> 2) + 64 * j;
            const EpiRowEnt re = Base::rowtab()[rl];
            r.p[j] = re.p;
            r.vd[j] = re.off + (tid & 3) * 16;
            r.vz[j] = rl * g.C * 8 + (tid & 3) * 16;
            r.vh[j] = rl * g.C * 4 + (tid & 3) * 16;
        }
        return r;
    }
    // descriptors of the tile's rows of Z (inside [Z|R], row stride 2 C), H~, dhp and of all of dOH; dead: zero records --
    // a load returns 0 without touching memory, a store is dropped (`writer`: only column tile 0 stores dhp)
    __device__ __forceinline__ GenSrd gen_srd(const AGen& g, bool live, bool writer) const {
        GenSrd s;
        const int C = g.C;
        s.z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.ZR + rm.base * 2L * C), 0, live ? rm.nvalid * C * 8 : 0, 0x00020000);
        s.t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Ht + rm.base * (long)C), 0, live ? rm.nvalid * C * 4 : 0, 0x00020000);
        s.d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dOH), 0, live ? (int)g.doh_bytes : 0, 0x00020000);
        s.o = __builtin_amdgcn_make_buffer_rsrc(g.dhp + rm.base * (long)C, 0, writer ? rm.nvalid * C * 4 : 0, 0x00020000);
        return s;
    }
    __device__ __forceinline__ void gen_load_half(int h, const GenSrd& s, const GenRows& r, int k0, GenRegs& q) const {
        const int so = (k0 + 16 * h) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            q.z[h + 2 * j] = buf_ld4(s.z, r.vz[j], so);
            q.t[h + 2 * j] = buf_ld4(s.t, r.vh[j], so);
            q.d[h + 2 * j] = buf_ld4(s.d, r.vd[j], so);
        }
    }
    __device__ __forceinline__ void gen_store_half(int h, const GenSrd& s, const GenRows& r, int k0, const GenRegs& q, const float4 (&rb)[4]) const {
        char* st = reinterpret_cast<char*>(lds) + h * STAGE_B;
        const int ko = (k0 + 16 * h) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = cp_off((tid >> 2) + 64 * j, tid & 3);
            const float4 z = q.z[h + 2 * j], t = q.t[h + 2 * j], d = q.d[h + 2 * j];
            const float p = r.p[j];
            float4 a;
            a.x = cb_dhp(__fmul_rn(p, d.x), z.x, t.x); a.y = cb_dhp(__fmul_rn(p, d.y), z.y, t.y);
            a.z = cb_dhp(__fmul_rn(p, d.z), z.z, t.z); a.w = cb_dhp(__fmul_rn(p, d.w), z.w, t.w);
            if constexpr (NP == 0) {
                *reinterpret_cast<float4*>(st + off) = a;
                *reinterpret_cast<float4*>(st + OPER_B + off) = rb[h + 2 * j];
            } else {                                   // bf16x3: the generated fp32 values are split exactly like loaded ones (store_half)
                const int offs = sp_off((tid >> 2) + 64 * j, (tid >> 1) & 1) + (tid & 1) * 8;
                // (opaque to the optimiser: left alone, hipcc contracts cb_dhp's last multiply with the split's first subtraction
                // -- fma(t1, t2, -piece1) -- so planes 2 and 3 held the residual of the UNROUNDED product while plane 1 and the
                // stored dhp hold the rounded one: 1e-7 away from the two-launch path, found by tests/test_gpu_ops.py)
                asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
                split_store(st + offs, a);
                split_store(st + OPER_B + offs, rb[h + 2 * j]);
            }
            buf_st4(s.o, r.vh[j] + ko, 0, a);          // (row + k offset in the VECTOR offset: see the store hazard note in DESIGN 5c.3)
        }
    }
    __device__ __forceinline__ void load_b_half(int h, const SrdsU& d, float4 (&rb)[4]) const {
#pragma unroll
        for (int j = 0; j < 2; ++j) rb[h + 2 * j] = buf_ld4(d.b, d.vb[j], d.sb + 64 * h);
    }
    // run_u's schedule with generated A slots; ONE segment (its B = the weights, [N][K]; its A pointer is not read), K % 32 == 0.
    // (walking the k slabs rotated so that the tile's own 128 columns come last was measured WORSE: profiles/r04_gen_pmc.txt)
    __device__ __forceinline__ void run_u_gen(f32x16 (&acc)[2][2], const AGen& g) {
        static_assert(NP == 0 || NP == 3, "generated A operand: fp32 storage (fp32 MFMA or the exact bf16x3 split)");
        nreg_u = 0;
        const int nslab = S.seg[0].K / GBK;
        if (nslab == 0) return;
        const bool writer = n0 == 0;
        auto kof = [&](int i) { return i * GBK; };      // k0 of the i-th slab walked
        __syncthreads();                            // the row table (fill_rowtab) is complete
        const GenRows rows = gen_rows(g);
        GenRegs q;
        float4 rb[4];
        int k_held = kof(0);
        {
            const SrdsU d = make_u(SegCursor{0, 0, k_held}, true);
            const GenSrd s = gen_srd(g, true, writer);
            gen_load_half(0, s, rows, k_held, q);
            gen_load_half(1, s, rows, k_held, q);
            load_b_half(0, d, rb);
            load_b_half(1, d, rb);
            gen_store_half(0, s, rows, k_held, q, rb);
            gen_store_half(1, s, rows, k_held, q, rb);
        }
        {
            const bool two = nslab > 1;
            k_held = kof(two ? 1 : 0);
            const SrdsU d = make_u(SegCursor{0, 0, k_held}, two);
            const GenSrd s = gen_srd(g, two, writer);
            gen_load_half(0, s, rows, k_held, q);
            gen_load_half(1, s, rows, k_held, q);
            load_b_half(0, d, rb);
            load_b_half(1, d, rb);
        }
        __syncthreads();
        compute(0, acc);
        const GenSrd sw = gen_srd(g, true, writer);      // for the stores of the held slab (always a live one)
        for (int it = 0; it + 1 < nslab; ++it) {
            const bool live = it + 2 < nslab;
            const int k_next = kof(live ? it + 2 : 0);
            const SrdsU nx = make_u(SegCursor{0, 0, k_next}, live);
            const GenSrd sn = gen_srd(g, live, writer);
            __syncthreads();
            fused_gen(0, 1, nx, sn, sw, rows, k_held, k_next, q, rb, acc);
            __syncthreads();
            fused_gen(1, 0, nx, sn, sw, rows, k_held, k_next, q, rb, acc);
            k_held = k_next;
        }
        __syncthreads();
        compute(1, acc);
        __syncthreads();
    }
    __device__ __forceinline__ void fused_gen(int hs, int hc, const SrdsU& nx, const GenSrd& sn, const GenSrd& sw, const GenRows& rows,
                                              int k_held, int k_next, GenRegs& q, float4 (&rb)[4], f32x16 (&acc)[2][2]) const {
        __builtin_amdgcn_sched_barrier(0);
        const Frags f = read_frags(hc);
        gen_store_half(hs, sw, rows, k_held, q, rb);
        mfmas(f, acc);
        gen_load_half(hs, sn, rows, k_next, q);
        load_b_half(hs, nx, rb);
        if constexpr (NP == 0) {
            // 32 MFMAs of 64 pipe cycles: the 8 fragment reads up front, then per MFMA gap a share of the generation arithmetic
            // (~12 VALU per slot), one LDS write or dhp store, later two of the eight global loads
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);   // VALU | SALU
                if (r < 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);        // DS write
                else __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);               // VMEM read
                if (r >= 2 && r < 4) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // VMEM write (dhp)
            }
        } else {
            // 24 bf16 MFMAs of 32 cycles: all 12 fragment reads, then per MFMA a share of the generation + split arithmetic
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int r = 0; r < 24; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // VALU
                if (r & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // DS write
                if (r >= 16) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                if (r >= 6 && r < 8) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // VMEM write (dhp)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // fragments of half h: NP planes of two 32-row blocks per operand (4 NP x ds_read_b128)
    struct FragsB { bf16x8 a[2][NP ? NP : 1], b[2][NP ? NP : 1]; };
    // NP = 0: [32-row block][k-quad pair kk]: lane half lh holds k = 4 (2 kk + lh) .. + 3 of its row, element j feeds the
    // MFMA of k-step 4 kk + j (the k order inside a half slab is free as long as A and B agree)
    struct FragsF { float4 a[2][2], b[2][2]; };
    using Frags = std::conditional_t<NP == 0, FragsF, FragsB>;
    __device__ __forceinline__ Frags read_frags(int h) const {
        const char* st = reinterpret_cast<const char*>(lds) + h * STAGE_B;
        const int lr = lane & 31, lh = lane >> 5;
        Frags f;
        if constexpr (NP == 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f.a[t][kk] = *reinterpret_cast<const float4*>(st + cp_off(wr * 64 + t * 32 + lr, 2 * kk + lh));
                    f.b[t][kk] = *reinterpret_cast<const float4*>(st + OPER_B + cp_off(wc * 64 + t * 32 + lr, 2 * kk + lh));
                }
            return f;
        } else {
#pragma unroll
        for (int p = NP - 1; p >= 0; --p)        // the last plane of A and plane 0 of B feed the first MFMAs
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f.a[t][p] = *reinterpret_cast<const bf16x8*>(st + p * SP_PLANE_B + sp_off(wr * 64 + t * 32 + lr, lh));
                f.b[t][NP - 1 - p] = *reinterpret_cast<const bf16x8*>(st + OPER_B + (NP - 1 - p) * SP_PLANE_B + sp_off(wc * 64 + t * 32 + lr, lh));
            }
        return f;
        }
    }
    // acc += A_h x B_h^T over 16 k: NP = 3: 24 MFMAs (6 partial products x 4 tiles, the four accumulators round-robin);
    // NP = 1: 4 MFMAs
    static constexpr int NPROD = NP == 3 ? 6 : 1;
    __device__ __forceinline__ static void mfmas(const Frags& f, f32x16 (&acc)[2][2]) {
        if constexpr (NP == 0) {     // 8 k-steps x 4 tiles of v_mfma_f32_32x32x2_f32
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(reinterpret_cast<const float*>(&f.a[mt][kk])[j],
                                                                              reinterpret_cast<const float*>(&f.b[nt][kk])[j], acc[mt][nt], 0, 0, 0);
        } else {
        constexpr int PA[6] = {NP == 3 ? 2 : 0, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest partial products first
#pragma unroll
        for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mt][PA[q]], f.b[nt][PB[q]], acc[mt][nt], 0, 0, 0);
        }
    }
    __device__ __forceinline__ void compute(int h, f32x16 (&acc)[2][2]) const { mfmas(read_frags(h), acc); }

    // ---- bf16-operand core (NP = 1): whole 32-k slabs per barrier ------------------------------------------------------------
    // A bf16 MFMA half step is 4 x 32 = 128 matrix-pipe cycles: with the half-step schedule a tile of K = 288 pays 18 barriers
    // and 18 LDS round trips for 2.3 k cycles of MFMA work (tools/wg_trace.py 2: K loop 15 us per tile).  One plane per
    // operand leaves room for TWO full-slab buffers (4 half stages, 32 KB, inside the half-tile epilogue image): slab it is
    // multiplied (8 MFMAs, 8 fragment reads) while slab it + 1 is stored into the other buffer and slab it + 2 requested --
    // one barrier per 32 k.
    __device__ __forceinline__ Frags read_frags_at(int stage) const {
        const char* st = reinterpret_cast<const char*>(lds) + stage * STAGE_B;
        const int lr = lane & 31, lh = lane >> 5;
        Frags f;
        if constexpr (NP == 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f.a[t][0] = *reinterpret_cast<const bf16x8*>(st + sp_off(wr * 64 + t * 32 + lr, lh));
                f.b[t][0] = *reinterpret_cast<const bf16x8*>(st + OPER_B + sp_off(wc * 64 + t * 32 + lr, lh));
            }
        }
        return f;
    }
    template <bool RELU>
    __device__ __forceinline__ void store_half_at(int stage, int h, const float4 (&ra)[4], const float4 (&rb)[4], int abf) const {
        char* st = reinterpret_cast<char*>(lds) + stage * STAGE_B;
        if (abf) *reinterpret_cast<float4*>(st + sp_off(tid >> 1, tid & 1)) = ra[h];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = sp_off((tid >> 2) + 64 * j, (tid >> 1) & 1) + (tid & 1) * 8;
            if (!abf) {
                float4 a = ra[h + 2 * j];
                if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                split_store(st + off, a);
            }
            split_store(st + OPER_B + off, rb[h + 2 * j]);
        }
    }
    template <bool RELU>
    __device__ __forceinline__ void run_u1(f32x16 (&acc)[2][2], int nslab) const {
        if (nslab == 0) return;
        float4 ra[4], rb[4];
        int held;
        {
            const SrdsU d = fetch_u(0, true);
            load_half(0, d, ra, rb);
            load_half(1, d, ra, rb);
            held = d.abf;
        }
        store_half_at<RELU>(0, 0, ra, rb, held);
        store_half_at<RELU>(1, 1, ra, rb, held);
        {
            const bool two = nslab > 1;
            const SrdsU d = fetch_u(1, two);
            load_half(0, d, ra, rb);
            load_half(1, d, ra, rb);
            held = d.abf;
        }
        __syncthreads();
        for (int it = 0; it < nslab; ++it) {
            const int cur = 2 * (it & 1), nxt = 2 - cur;
            const bool more = it + 1 < nslab, live = it + 2 < nslab;
            const SrdsU nx = fetch_u(it + 2, live);
            __builtin_amdgcn_sched_barrier(0);
            const Frags f0 = read_frags_at(cur), f1 = read_frags_at(cur + 1);
            if (more) {
                store_half_at<RELU>(nxt, 0, ra, rb, held);
                store_half_at<RELU>(nxt + 1, 1, ra, rb, held);
            }
            mfmas(f0, acc);
            mfmas(f1, acc);
            if (more) {
                load_half(0, nx, ra, rb);
                load_half(1, nx, ra, rb);
            }
            __builtin_amdgcn_sched_barrier(0);
            held = nx.abf;
            __syncthreads();
        }
    }

    // ---- tile-final slab descriptors in LDS (bf16-operand core) ------------------------------------------------------------------
    // make_u() reads the segment's fields from the kernel arguments with dynamically indexed scalar loads: ~10 DEPENDENT s_load +
    // s_waitcnt pairs per slab, ~2500 cycles.  Behind 2048-cycle fp32 half steps that is free; behind the 256 MFMA cycles of a
    // bf16 slab it WAS the K loop (13.8 of the gate GEMM's 19 us per tile).  For NP = 1 the descriptors of all slabs of the tile
    // are therefore computed once, one thread per slab, into the (otherwise unused) iteration-table area; the loop fetches
    // slab it + 2's twelve dwords with three broadcast ds_read_b128 and v_readfirstlane.
    struct UDesc { unsigned a_lo, a_hi, b_lo, b_hi; int a_rec, b_rec, lda_b, ldb_b, sa, sb, bks, fmt; };   // fmt: bit 0 abf, bit 1 frag
    static_assert(sizeof(UDesc) == 48 && G_MAX_ITERS * sizeof(ItDesc) >= 64 * sizeof(UDesc), "descriptor table fits the iteration table");
    __device__ __forceinline__ UDesc* utab() const { return reinterpret_cast<UDesc*>(reinterpret_cast<char*>(table) + 1024); }   // after the region list
    __device__ __forceinline__ int plan_u() {
        uniform_regions();
        int n = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < S.nseg) {
                const GemmSeg& g = S.seg[q];
                const bool rep = (g.flags & SEG_REPEAT) != 0, reg = REGION && (g.flags & SEG_REGION) != 0;
                const int cnt = seg_count(g), nk = g.K / GBK, local = tid - n;
                if (local >= 0 && local < cnt * nk && tid < 64) {
                    const int ri = local / nk, k0 = (local - ri * nk) * GBK;
                    const int region = reg ? reinterpret_cast<const int*>(table)[ri] : -1;
                    UDesc u;
                    const int abf = (g.flags & SEG_A_BF16) ? 1 : 0, frag = (g.flags & SEG_B_FRAG) ? 1 : 0;
                    u.lda_b = (int)g.lda * (abf ? 2 : 4);
                    u.ldb_b = (int)g.ldb * 4;
                    const unsigned long long ap = reinterpret_cast<unsigned long long>(g.A) + (rep ? (long)ri * g.a_rep_stride * 4 : 0) + rm.base * u.lda_b;
                    const long boff = reg ? (long)region * g.b_region_stride : (rep ? (long)ri * g.b_region_stride : 0);
                    const bool lowb = n0 < g.nsplit;
                    unsigned long long bp;
                    if (frag) {
                        u.bks = (g.K / 16) * 1024;
                        const int nl = lowb ? n0 : n0 - g.nsplit;
                        bp = reinterpret_cast<unsigned long long>(lowb ? g.B0 : g.B1) + boff + (long)(nl / 32) * u.bks;
                        u.b_rec = 4 * u.bks;
                        u.sb = (k0 / 16) * 1024;
                    } else {
                        u.bks = 0;
                        bp = reinterpret_cast<unsigned long long>(lowb ? g.B0 + boff + (long)n0 * g.ldb : g.B1 + boff + (long)(n0 - g.nsplit) * g.ldb);
                        const int blim = (lowb && g.nsplit < N ? g.nsplit : N) - n0;
                        u.b_rec = (blim < GBN ? blim : GBN) * u.ldb_b;
                        u.sb = k0 * 4;
                    }
                    u.a_lo = (unsigned)ap; u.a_hi = (unsigned)(ap >> 32); u.b_lo = (unsigned)bp; u.b_hi = (unsigned)(bp >> 32);
                    u.a_rec = rm.nvalid * u.lda_b;
                    u.sa = k0 * (abf ? 2 : 4);
                    u.fmt = abf | (frag << 1) | ((region + 1) << 8);
                    utab()[tid] = u;
                }
                n += cnt * nk;
            }
        }
        __syncthreads();
        return n;
    }
    // descriptor of slab `it` from the table (dead: num_records 0); region-masked rows get an out-of-range offset
    __device__ __forceinline__ SrdsU fetch_u(int it, bool live) const {
        const int* w = reinterpret_cast<const int*>(utab() + (live ? it : 0));
        const int4 w0 = *reinterpret_cast<const int4*>(w), w1 = *reinterpret_cast<const int4*>(w + 4), w2 = *reinterpret_cast<const int4*>(w + 8);
#define RFL_(x) __builtin_amdgcn_readfirstlane(x)
        const unsigned long long ap = ((unsigned long long)(unsigned)RFL_(w0.y) << 32) | (unsigned)RFL_(w0.x);
        const unsigned long long bp = ((unsigned long long)(unsigned)RFL_(w0.w) << 32) | (unsigned)RFL_(w0.z);
        const int a_rec = RFL_(w1.x), b_rec = RFL_(w1.y), lda_b = RFL_(w1.z), ldb_b = RFL_(w1.w);
        const int fmt = RFL_(w2.w);
        SrdsU d;
        d.sa = RFL_(w2.x); d.sb = RFL_(w2.y); d.bks = RFL_(w2.z);
#undef RFL_
        d.abf = fmt & 1;
        const int region = (fmt >> 8) - 1;
        d.a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ap), 0, live ? a_rec : 0, 0x00020000);
        d.b = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bp), 0, live ? b_rec : 0, 0x00020000);
        if (d.abf) {
            d.va[0] = (tid >> 1) * lda_b + (tid & 1) * 16;
            d.va[1] = 0;
        } else {
            d.va[0] = (tid >> 2) * lda_b + (tid & 3) * 16;
            d.va[1] = d.va[0] + 64 * lda_b;
            if (REGION && region >= 0) {
                d.va[0] = sreg[0] == region ? d.va[0] : (int)Base::SRD_OOB;
                d.va[1] = sreg[1] == region ? d.va[1] : (int)Base::SRD_OOB;
            }
        }
        if (fmt & 2) {
            d.vb[0] = 2 * wc * d.bks + lane * 16;
            d.vb[1] = 0;
        } else {
            d.vb[0] = (tid >> 2) * ldb_b + (tid & 3) * 16;
            d.vb[1] = d.vb[0] + 64 * ldb_b;
        }
        return d;
    }

    // ---- bf16-operand core, weights in fragment order (SEG_B_FRAG): the B fragments never touch LDS ----------------------------
    // Timing experiments on the bf16 gate GEMM (tools/wg_trace.py 2): without ANY global load in the K loop it still took 12 of
    // its 16 us per tile, without the MFMAs 14.5 -- the loop is bound by its LDS round trips and the conversion of the weight
    // tile (per 32-k slab and wave: 4 + 2 ds_write, 8 ds_read_b128, 8 v_cvt_pk behind a barrier, for 8 MFMAs of 32 cycles).
    // The weights are the same for every tile: they are copied once per step to bf16 in the order the MFMA wants them
    // (launch_cvt_bf16_frag), and every wave loads its own four 1 KB fragments per slab with four coalesced 16-byte loads,
    // one slab ahead, straight into the registers the MFMAs read.  LDS carries the A operand only.
    struct BFrags { bf16x8 v[2][2]; };          // [16-k half][32-column block of the wave]
    __device__ __forceinline__ void load_bfrags(const SrdsU& d, BFrags& b) const {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                b.v[h][ni] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(d.b, d.vb[0], d.sb + h * 1024 + ni * d.bks, 0));
    }
    __device__ __forceinline__ void load_a_half(int h, const SrdsU& d, float4 (&ra)[4]) const {
        if (d.abf) {
            ra[h] = buf_ld4(d.a, d.va[0], d.sa + 32 * h);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) ra[h + 2 * j] = buf_ld4(d.a, d.va[j], d.sa + 64 * h);
        }
    }
    template <bool RELU>
    __device__ __forceinline__ void store_a_at(int stage, int h, const float4 (&ra)[4], int abf) const {
        char* st = reinterpret_cast<char*>(lds) + stage * STAGE_B;
        if (abf) *reinterpret_cast<float4*>(st + sp_off(tid >> 1, tid & 1)) = ra[h];
        else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float4 a = ra[h + 2 * j];
                if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                split_store(st + sp_off((tid >> 2) + 64 * j, (tid >> 1) & 1) + (tid & 1) * 8, a);
            }
        }
    }
    template <bool RELU>
    __device__ __forceinline__ void run_u1f(f32x16 (&acc)[2][2], int nslab) const {
        static_assert(NP == 1, "fragment-order weights: bf16-operand core only");
        if (nslab == 0) return;
        float4 ra[4];
        BFrags b0, b1;
        int held;
        {
            const SrdsU d = fetch_u(0, true);
            load_a_half(0, d, ra);
            load_a_half(1, d, ra);
            load_bfrags(d, b0);
            held = d.abf;
        }
        store_a_at<RELU>(0, 0, ra, held);
        store_a_at<RELU>(1, 1, ra, held);
        {
            const bool two = nslab > 1;
            const SrdsU d = fetch_u(1, two);
            load_a_half(0, d, ra);
            load_a_half(1, d, ra);
            load_bfrags(d, b1);
            held = d.abf;
        }
        __syncthreads();
        // slab it: A in LDS buffer it & 1, its B fragments in bc; registers ra / bn hold slab it + 1; slab it + 2 is requested
        auto step = [&](int it, const BFrags& bc, BFrags& bn_out) {
            const int cur = 2 * (it & 1), nxt = 2 - cur;
            const bool more = it + 1 < nslab, live = it + 2 < nslab;
            const SrdsU nx = fetch_u(it + 2, live);
            __builtin_amdgcn_sched_barrier(0);
            const char* st = reinterpret_cast<const char*>(lds) + cur * STAGE_B;
            const int lr = lane & 31, lh = lane >> 5;
            bf16x8 fa[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t) fa[h][t] = *reinterpret_cast<const bf16x8*>(st + h * STAGE_B + sp_off(wr * 64 + t * 32 + lr, lh));
            if (more) {
                store_a_at<RELU>(nxt, 0, ra, held);
                store_a_at<RELU>(nxt + 1, 1, ra, held);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][mt], bc.v[h][nt], acc[mt][nt], 0, 0, 0);
            if (more) {
                load_a_half(0, nx, ra);
                load_a_half(1, nx, ra);
            }
            load_bfrags(nx, bn_out);        // the fragments of slab it + 2 replace those of slab it (dead descriptor: zeros)
            __builtin_amdgcn_sched_barrier(0);
            held = nx.abf;
            __syncthreads();
        };
        for (int it = 0; it < nslab; it += 2) {
            step(it, b0, b0);
            if (it + 1 < nslab) step(it + 1, b1, b1);
        }
    }

    // store half hs of the next slab while the MFMAs of half hc of the current slab run: one MFMA, then a few of the
    // conversion VALU ops and now and then a plane write, so that a single wave keeps its SIMD's matrix pipe busy (a
    // bf16 32x32x16 MFMA occupies the pipe for 32 cycles = 8 issue slots).  The registers just stored are refilled
    // with the same half of the slab after next.
    template <bool RELU, class D>
    __device__ __forceinline__ void fused(int hs, int hc, const D& next, float4 (&ra)[4], float4 (&rb)[4], f32x16 (&acc)[2][2],
                                          int held_abf) const {
        __builtin_amdgcn_sched_barrier(0);       // the interleaving pattern below applies to this block only
        // fragment reads first in program order: the LDS writes below cannot be proven disjoint from them and would
        // otherwise pin the reads (and with them every MFMA) behind the whole conversion
        const Frags f = read_frags(hc);
        store_half<RELU>(hs, ra, rb, held_abf);
        mfmas(f, acc);
        load_half(hs, next, ra, rb);
        if (NP == 0) {
            // 32 MFMAs of 64 pipe cycles each: the 8 fragment reads up front (the first MFMAs need kk = 0 only, but a read
            // issued late would wait behind the stores), then one LDS write / one global load per MFMA gap
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);   // VALU | SALU (addresses, relu)
                if (r < 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // DS write
                else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // VMEM read
            }
        } else if (NP == 3) {
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);      // all fragment reads
#pragma unroll
            for (int r = 0; r < 24; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU (split arithmetic)
                if (r & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // DS write
                if (r >= 18 && r < 22) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);       // the four fragment reads
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU (conversion)
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // Schedule (slab it in LDS, slab it+1 in registers, the halves of slab it+2 requested as their registers free up):
    //   C0(0) | { bar ; S0(it+1)+C1(it) ; bar ; S1(it+1)+C0(it+1) } ... | bar ; C1(last) ; bar
    // stage 0 is rewritten only after the barrier that follows every wave's C0, stage 1 after the one that follows C1.
    template <bool RELU>
    __device__ __forceinline__ void run_t(f32x16 (&acc)[2][2]) const {
        float4 ra[4], rb[4];
        int held;                  // format of the A rows currently in the registers (bf16-stored segment or fp32)
        {
            const Srds d = slab_srds(table[0], true);
            load_half(0, d, ra, rb);
            load_half(1, d, ra, rb);
            held = d.abf;
        }
        store_half<RELU>(0, ra, rb, held);
        store_half<RELU>(1, ra, rb, held);
        {
            const bool two = nit > 1;
            const Srds d = slab_srds(table[two ? 1 : 0], two);
            load_half(0, d, ra, rb);
            load_half(1, d, ra, rb);
            held = d.abf;
        }
        __syncthreads();
        compute(0, acc);
        for (int it = 0; it + 1 < nit; ++it) {
            const bool live = it + 2 < nit;
            const Srds nx = slab_srds(table[live ? it + 2 : it + 1], live);      // once per slab, shared by both halves
            __syncthreads();
            fused<RELU>(0, 1, nx, ra, rb, acc, held);
            __syncthreads();
            fused<RELU>(1, 0, nx, ra, rb, acc, held);
            held = nx.abf;
        }
        __syncthreads();
        compute(1, acc);
        __syncthreads();
    }
    // Epilogue through LDS in two 64-row halves (the waves with wr == half own those rows): half the image of
    // FastCore::for_each_vec, which is what lets a third workgroup onto the CU.
    // stage the accumulators of the waves that own 64-row half `half` (between two barriers)
    __device__ __forceinline__ void stage_half(int half, f32x16 (&acc)[2][2]) const {
        const int lr = lane & 31, lh = lane >> 5;
        __syncthreads();
        if (wr == half) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        lds[(mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * G_LDS_KROW + wc * 64 + ni * 32 + lr] = acc[mi][ni][reg];
        }
        __syncthreads();
    }
    // Straight-line body for a full tile (FastCore::vec_body's counterpart): rounds k = 0 .. NR-1 of RR rows walk half 0,
    // then half 1; round k + 1 is requested before round k is applied when two rounds of operands fit the registers.
    template <class F, int V>
    __device__ __forceinline__ void vec_body_halves(f32x16 (&acc)[2][2], const F& f) const {
        // (a thread owns 8 rows of each half; two rounds in flight within ~48 registers: 8, 4, 2 or 1 rows per round)
        constexpr int AB = (int)sizeof(typename F::VAux);
        constexpr int BUDGET = epi_aux_budget<F>(0);      // bytes of auxiliary operands per round (functors of 2-workgroup kernels ask for more)
        constexpr int RR = AB * 8 <= BUDGET ? 8 : (AB * 4 <= BUDGET ? 4 : (AB * 2 <= BUDGET ? 2 : 1)), RPH = 8 / RR, NR = 2 * RPH;
        const EpiGeom geo{rm.base, n0, tid >> 5, 4 * (tid & 31), 8, Base::rowtab()};
        const typename F::Tile tl = f.template vtile<V>(geo);
        const typename F::Col col = f.template vcol<V>(Base::ecol());
        typename F::VAux aux[2][RR];
#pragma unroll
        for (int j = 0; j < RR; ++j) aux[0][j] = f.template vload<V>(tl, j);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (k % RPH == 0) {
                WG_MARK(3 * (k / RPH));
                stage_half(k / RPH, acc);
                WG_MARK(3 * (k / RPH) + 1);
            }
            if (k + 1 < NR) {
#pragma unroll
                for (int j = 0; j < RR; ++j) aux[(k + 1) & 1][j] = f.template vload<V>(tl, RR * (k + 1) + j);
            }
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                const int i = RR * k + j;         // row slot: row (tid >> 5) + 8 i; inside its half: slot i & 7
                f.template vapply<V>(tl, i, *reinterpret_cast<const float4*>(lds + ((tid >> 5) + 8 * (i & 7)) * G_LDS_KROW + 4 * (tid & 31)),
                                     col, aux[k & 1][j]);
            }
            if (k % RPH == RPH - 1) WG_MARK(3 * (k / RPH) + 2);
        }
        __syncthreads();
    }
    template <class F>
    __device__ __forceinline__ void for_each_vec_halves(f32x16 (&acc)[2][2], const F& f) const {
        constexpr int RR = F::ROUND_ROWS / 2;      // a thread owns 8 rows of each half; half the round of the 256-VGPR cores
        static_assert(RR == 4 || RR == 8, "rows per epilogue round");   // (these kernels are capped at 168 VGPRs: three workgroups per CU)
        const int v = Base::tile_variant(f);
        if (v >= 0) {
            Base::template dispatch_variant<F, 0>(v, [&](auto tag) { vec_body_halves<F, decltype(tag)::value>(acc, f); });
            return;
        }
        const int lr = lane & 31, lh = lane >> 5;
        const int c = Base::ecol();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // first round of auxiliary loads before the half is staged (see FastCore::for_each_vec)
            typename F::Aux aux[RR];
            if (c < N) {
#pragma unroll
                for (int j = 0; j < RR; ++j) {
                    const int r = half * 64 + (tid >> 5) + 8 * j;
                    if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
                }
            }
            WG_MARK(3 * half);
            __syncthreads();
            if (wr == half) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            lds[(mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * G_LDS_KROW + wc * 64 + ni * 32 + lr] = acc[mi][ni][reg];
            }
            __syncthreads();
            WG_MARK(3 * half + 1);
            if (c < N) {
#pragma unroll
                for (int g = 0; g < 8 / RR; ++g) {
                    if (g > 0) {
#pragma unroll
                        for (int j = 0; j < RR; ++j) {
                            const int r = half * 64 + (tid >> 5) + 8 * (RR * g + j);
                            if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RR; ++j) {
                        const int rl = (tid >> 5) + 8 * (RR * g + j), r = half * 64 + rl;
                        if (r < rm.nvalid)
                            f.apply(rm.grow(r), c, *reinterpret_cast<const float4*>(lds + rl * G_LDS_KROW + 4 * (tid & 31)), aux[j]);
                    }
                }
            }
            WG_MARK(3 * half + 2);
        }
        __syncthreads();
    }

    // 8-column-per-thread form of the half-tile epilogue (functors of gemm.hip's "8F" family): thread = (row tid >> 4 (+16 i),
    // columns 8 (tid & 15) .. +7) -- 16 bytes per access of a bf16 array, two float4 of an fp32 one.
    // straight-line body of the 8-column form (full tile + functor variant): row slots i = 0 .. 7, row (tid >> 4) + 16 i
    template <class F, int V>
    __device__ __forceinline__ void vec8_body_halves(f32x16 (&acc)[2][2], const F& f) const {
        constexpr int AB = (int)sizeof(typename F::VAux);
        constexpr int RR = AB * 4 <= 96 ? 4 : (AB * 2 <= 96 ? 2 : 1), RPH = 4 / RR, NR = 2 * RPH;
        const EpiGeom geo{rm.base, n0, tid >> 4, 8 * (tid & 15), 16, Base::rowtab()};
        const typename F::Tile tl = f.template vtile<V>(geo);
        const typename F::Col col = f.template vcol<V>(n0 + 8 * (tid & 15));
        typename F::VAux aux[2][RR];
#pragma unroll
        for (int j = 0; j < RR; ++j) aux[0][j] = f.template vload<V>(tl, j);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (k % RPH == 0) stage_half(k / RPH, acc);
            if (k + 1 < NR) {
#pragma unroll
                for (int j = 0; j < RR; ++j) aux[(k + 1) & 1][j] = f.template vload<V>(tl, RR * (k + 1) + j);
            }
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                const int i = RR * k + j;
                const float4* img = reinterpret_cast<const float4*>(lds + ((tid >> 4) + 16 * (i & 3)) * G_LDS_KROW + 8 * (tid & 15));
                f.template vapply<V>(tl, i, typename F::Vec{img[0], img[1]}, col, aux[k & 1][j]);
            }
        }
        __syncthreads();
    }
    template <class F>
    __device__ __forceinline__ void for_each_vec8_halves(f32x16 (&acc)[2][2], const F& f) const {
        constexpr int RR = F::ROUND_ROWS;          // of the thread's 4 rows per half
        static_assert(RR == 1 || RR == 2 || RR == 4, "rows per epilogue round");
        const int v = Base::tile_variant(f);
        if (v >= 0) {
            Base::template dispatch_variant<F, 0>(v, [&](auto tag) { vec8_body_halves<F, decltype(tag)::value>(acc, f); });
            return;
        }
        const int lr = lane & 31, lh = lane >> 5;
        const int c = n0 + 8 * (tid & 15);
        const bool live = c < N;
        typename F::ColAux ca;
        if (live) ca = f.col(c);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            typename F::Aux aux[RR];
            if (live) {
#pragma unroll
                for (int j = 0; j < RR; ++j) {
                    const int r = half * 64 + (tid >> 4) + 16 * j;
                    if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
                }
            }
            __syncthreads();
            if (wr == half) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            lds[(mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * G_LDS_KROW + wc * 64 + ni * 32 + lr] = acc[mi][ni][reg];
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int g = 0; g < 4 / RR; ++g) {
                    if (g > 0) {
#pragma unroll
                        for (int j = 0; j < RR; ++j) {
                            const int r = half * 64 + (tid >> 4) + 16 * (RR * g + j);
                            if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RR; ++j) {
                        const int rl = (tid >> 4) + 16 * (RR * g + j), r = half * 64 + rl;
                        if (r < rm.nvalid) {
                            const float4* img = reinterpret_cast<const float4*>(lds + rl * G_LDS_KROW + 8 * (tid & 15));
                            f.apply(rm.grow(r), c, typename F::Vec{img[0], img[1]}, ca, aux[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    __device__ __forceinline__ void run(f32x16 (&acc)[2][2], bool relu_a) const {
        if (nit == 0) return;
        if (relu_a) run_t<true>(acc);            // head only: relu on A while staging
        else run_t<false>(acc);
    }
    // scalar-descriptor path (no iteration table: plan() is not needed); host-checked eligibility
    // scalar descriptors + weights in fragment order (every segment SEG_B_FRAG; host-checked)
    __device__ __forceinline__ void run_uniform_frag(f32x16 (&acc)[2][2], bool relu_a) {
        if constexpr (NP == 1) {
            const int nslab = plan_u();
            if (relu_a) run_u1f<true>(acc, nslab);
            else run_u1f<false>(acc, nslab);
        }
    }
    __device__ __forceinline__ void run_uniform(f32x16 (&acc)[2][2], bool relu_a) {
        if constexpr (NP == 1) {
            const int nslab = plan_u();
            if (relu_a) run_u1<true>(acc, nslab);
            else run_u1<false>(acc, nslab);
        } else {
            uniform_regions();
            if (relu_a) run_u<true>(acc);
            else run_u<false>(acc);
        }
    }
};

}  // namespace regt
