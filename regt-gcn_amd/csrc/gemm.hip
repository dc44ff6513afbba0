// Dense contractions of the RegT-GCN pipeline on the matrix cores.
//
//   gemm_flat_split_kernel<Epi, REGION, NP>  C = sum_seg A_seg B_seg^T with a fused epilogue (forward + data gradients) on the
//                                      three-workgroup core SplitCore (gemm_split.h): NP = 0 fp32 planes on
//                                      v_mfma_f32_32x32x2_f32 (the default), 3 = exact 3-way bf16 split, 1 = bf16 operands
//                                      (v_mfma_f32_32x32x16_bf16); gemm_flat_split8_kernel: NP = 1 with bf16-stored activations
//   gemm_flat_fast_kernel<Epi, Core>   the two-workgroup fp32 core FastCore (gemm_fast.h): weights stored [K][N],
//                                      REGT_FP32_CORE=wide
//   gemm_flat_small_kernel<Epi, ..>    the same with 64 x 64 tiles for problems of fewer than 128 big tiles (gemm_small.h)
//   gemm_flat_kernel<Epi>              generic fallback (operands that are not 16-byte tileable)
//   gemm_cand_split_kernel<NP> / gemm_cand_split8_kernel   candidate state + GRU blend + attention-weighted sum over the T
//                                      periods (gemm_cand_flat_kernel<Core>: small tiles / two-workgroup core)
//   (weight gradients and the reduction of their slabs: wgrad.hip)
//   small_gemm_multi_kernel            strided batched C = A B for the (C x F)-sized weight compositions
#include "kernels.h"
#include "gemm_fast.h"
#include "gemm_split.h"
#include "gemm_small.h"

namespace regt {

// Developer build (REGT_HIPCC_FLAGS=-DREGT_WG_TRACE, tools/wg_trace.py): the flat GEMM kernels with N == REGT_WG_TRACE_N record,
// per workgroup, the 100 MHz wall clock at the start of the K loop, at its end and after the epilogue, plus HW_ID / XCC_ID
// (which CU it ran on) -- the data behind DESIGN.md's "who overlaps with whom on a CU" analysis.
#ifdef REGT_WG_TRACE
__device__ long g_wg_trace[4 * WG_TRACE_MAX];
__device__ long g_wg_marks[8 * WG_TRACE_MAX];
#define WG_TRACE_T(name) const long name = wall_clock64()
#define WG_TRACE_END(N, ta, tb)                                                                                              \
    if (threadIdx.x == 0 && WG_TILE_ID < WG_TRACE_MAX && (N) == REGT_WG_TRACE_N) {                                            \
        long* q_ = g_wg_trace + 4L * WG_TILE_ID;                                                                              \
        q_[0] = ta; q_[1] = tb; q_[2] = wall_clock64();                                                                       \
        q_[3] = ((long)__builtin_amdgcn_s_getreg((31 << 11) | 4)) | ((long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
    }
#else
#define WG_TRACE_T(name)
#define WG_TRACE_END(N, ta, tb)
#endif

// ---- epilogue functors ---------------------------------------------------------------------------
// Each functor has a scalar form (m, c, v) used when the output is not 16-byte tileable (e.g. the
// (N, O) head output) and a vector form: load() fetches the auxiliary operands of one float4,
// apply() finishes and stores it (driver: GemmCore::for_each_vec).
//
// Straight-line variants (vcol<V> / vtile<V> / vload<V> / vapply<V>, drivers: FastCore::for_each_vec, SplitCore::for_each_vec_halves):
// every decision of the functor that is uniform over a tile (which activation, bf16 or fp32 store, does this column tile
// hold the r gate, ...) is folded into the compile-time variant V = variant(n0) in [0, NVAR) -- or -1: no specialisation,
// use load/apply.  For a full tile the driver then runs a body without a single branch.  That is what keeps hipcc's
// s_waitcnt exact: with per-row `if`s every row of the epilogue became its own basic block and waited vmcnt(0) -- for
// its own loads AND for the stores of the row before it (stores count on vmcnt on gfx9): a chain of one memory round trip
// per row, 11 us (two-workgroup core) to 21 us (three-workgroup core) per 128 x 128 tile (tools/wg_trace.py).
// Column constants (bias) are loaded once per thread (Col), not once per row.
// Addresses: per array one buffer descriptor for the tile's origin (SGPRs), one per-thread byte offset (vtile) and a
// wave-uniform row step: loads take it as the instruction's scalar offset (no vector instruction per row goes into their
// addressing), STORES add it to the vector offset and keep soffset = 0.  A 16-byte buffer store with an SGPR soffset
// whose data registers the very next VALU instruction overwrites picked up the NEW value now and then on gfx950 (the
// first dword of a row came out as 1 + e^-x instead of the sigmoid: tests/test_gpu_ops.py::test_linear_sigmoid_rows); the
// compiler only pads that hazard with a wait state when soffset is an immediate (GCNHazardRecognizer: "this hazard only
// exists if the instruction is not using a register in the soffset field").  That matters more than it looks: a wave that shares its SIMD with waves streaming MFMAs gets a VALU issue
// slot only every ~64-200 cycles (tools/micro/valu_under_mfma.hip: 10x slower next to two MFMA-bound waves, whatever its
// s_setprio), so the epilogue's duration is its VALU instruction count times that, not its memory traffic.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// four fp32 -> four bf16 (round to nearest even, v_cvt_pk_bf16_f32), one 8-byte store; `p` addresses bf16 elements
__device__ __forceinline__ void st4_bf16(void* base, long elem, float4 v) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = {v.x, v.y}, hi = {v.z, v.w};
    const bf16x2_t bl = __builtin_convertvector(lo, bf16x2_t), bh = __builtin_convertvector(hi, bf16x2_t);
    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base) + 2 * elem) =
        make_uint2(__builtin_bit_cast(unsigned, bl), __builtin_bit_cast(unsigned, bh));
}
#define REGT_V4(expr_x) make_float4(expr_x(x), expr_x(y), expr_x(z), expr_x(w))

struct EpiBiasActF {
    EpiBiasAct e;
    // none / leaky_relu / relu are all "v > 0 ? v : v * ns" with ns = 1 / slope / 0; sigmoid / tanh for the gate GEMMs of the
    // zero-hidden cell (regt_cell0_forward)
    __device__ __forceinline__ float act(float v) const {
        if (e.act == ACT_SIGMOID) return fast_sigmoid(v);
        if (e.act == ACT_TANH) return fast_tanh(v);
        const float ns = e.act == ACT_NONE ? 1.0f : (e.act == ACT_LRELU ? e.slope : 0.0f);
        return v > 0.f ? v : v * ns;
    }
    __device__ __forceinline__ void operator()(long m, int c, float v) const {
        e.out[m * e.ldo + c] = act(v + (e.bias ? e.bias[c] : 0.f));
    }
    static constexpr int ROUND_ROWS = 16;
    struct Aux { float4 b; };
    __device__ __forceinline__ Aux load(long, int c) const { return Aux{e.bias ? ld4(e.bias + c) : make_float4(0, 0, 0, 0)}; }
    __device__ __forceinline__ void apply(long m, int c, float4 v, const Aux& a) const {
#define F_(k) act(v.k + a.b.k)
        st4(e.out + m * e.ldo + c, REGT_V4(F_));
#undef F_
    }
    // variants: 0 = none / leaky relu / relu (one select), 1 = sigmoid, 2 = tanh
    static constexpr int NVAR = 3;
    static constexpr bool HAS_ROWTAB = false;
    struct Col { float4 b; };
    struct VAux {};
    struct Tile { __amdgpu_buffer_rsrc_t out; int v, s; };
    // (-1 = the guarded path: the straight-line body addresses a tile with 32-bit byte offsets, 120 rows x ldo x 4 B must fit)
    __device__ __forceinline__ int variant(int) const { return e.ldo >= (1L << 22) ? -1 : (e.act == ACT_SIGMOID ? 1 : (e.act == ACT_TANH ? 2 : 0)); }
    template <int V> __device__ __forceinline__ Col vcol(int c) const { return Col{e.bias ? ld4(e.bias + c) : make_float4(0, 0, 0, 0)}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        return Tile{buf_srd(e.out + g.m0 * e.ldo + g.n0), (g.rr * (int)e.ldo + g.c) * 4, g.step * (int)e.ldo * 4};
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile&, int) const { return VAux{}; }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, float4 v, const Col& col, const VAux&) const {
        const float ns = e.act == ACT_NONE ? 1.0f : (e.act == ACT_LRELU ? e.slope : 0.0f);
#define F_(k) (V == 1 ? fast_sigmoid(v.k + col.b.k) : V == 2 ? fast_tanh(v.k + col.b.k) : ((v.k + col.b.k) > 0.f ? (v.k + col.b.k) : (v.k + col.b.k) * ns))
        buf_st4(t.out, t.v + i * t.s, 0, REGT_V4(F_));
#undef F_
    }
};
struct EpiGatesF {
    EpiGates e;
    __device__ __forceinline__ void operator()(long m, int c, float v) const {
        float g = fast_sigmoid(v + e.bias[c]);
        e.ZR[m * (2L * e.C) + c] = g;
        if (c >= e.C) e.q[m * e.C + c - e.C] = e.h[m * e.C + c - e.C] * g;      // fp32 storage only (vector path handles bf16)
    }
    static constexpr int ROUND_ROWS = 16;
    struct Aux { float4 b, h; };
    __device__ __forceinline__ Aux load(long m, int c) const {
        Aux a;
        a.b = ld4(e.bias + c);
        a.h = c >= e.C ? ld4(e.h + m * e.C + c - e.C) : make_float4(0, 0, 0, 0);
        return a;
    }
    __device__ __forceinline__ void apply(long m, int c, float4 v, const Aux& a) const {
#define F_(k) fast_sigmoid(v.k + a.b.k)
        const float4 g = REGT_V4(F_);
#undef F_
        st4(e.ZR + m * (2L * e.C) + c, g);
        if (c >= e.C) {
            const float4 qv = make_float4(a.h.x * g.x, a.h.y * g.y, a.h.z * g.z, a.h.w * g.w);
            if (e.q_bf16) st4_bf16(e.q, m * e.C + c - e.C, qv);
            else st4(e.q + m * e.C + c - e.C, qv);
        }
    }
    // variants: bit 0 = the tile holds r columns (reads h, writes q = r * h), bit 1 = q stored as bf16; tiles are pure z
    // or pure r when C is a multiple of the tile width.  The arithmetic is load()/apply()'s to the bit: the backward pass
    // forms R (1 - R) from the stored gate, and for a saturated gate one ulp of R is a percent of 1 - R -- a tile must not
    // round differently from its partial neighbour (folding the bias into the exponent's fma saved a VALU per element
    // and moved the r-gate gradients by 1 %).
    static constexpr int NVAR = 4;
    static constexpr bool HAS_ROWTAB = false;
    struct Col { float4 b; };
    struct VAux { float4 h; };
    struct Tile { __amdgpu_buffer_rsrc_t zr, h, q; int vzr, vh, vq, szr, sh, sq; };
    __device__ __forceinline__ int variant(int n0) const { return e.C % GBN ? -1 : (n0 >= e.C ? 1 : 0) + (e.q_bf16 ? 2 : 0); }
    template <int V> __device__ __forceinline__ Col vcol(int c) const { return Col{ld4(e.bias + c)}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.zr = buf_srd(e.ZR + g.m0 * (2L * e.C) + g.n0);
        t.vzr = (g.rr * 2 * e.C + g.c) * 4;
        t.szr = g.step * 2 * e.C * 4;
        if (V & 1) {
            const long o = g.m0 * e.C + g.n0 - e.C;
            t.h = buf_srd(e.h + o);
            t.vh = (g.rr * e.C + g.c) * 4;
            t.sh = g.step * e.C * 4;
            t.q = buf_srd((V & 2) ? reinterpret_cast<const char*>(e.q) + 2 * o : reinterpret_cast<const char*>(e.q) + 4 * o);
            t.vq = (V & 2) ? t.vh >> 1 : t.vh;
            t.sq = (V & 2) ? t.sh >> 1 : t.sh;
        }
        return t;
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        VAux a;
        if (V & 1) a.h = buf_ld4(t.h, t.vh, i * t.sh);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, float4 v, const Col& col, const VAux& a) const {
#define F_(k) fast_sigmoid(v.k + col.b.k)
        const float4 g = REGT_V4(F_);
#undef F_
        buf_st4(t.zr, t.vzr + i * t.szr, 0, g);
        if (V & 1) {
            const float4 qv = make_float4(a.h.x * g.x, a.h.y * g.y, a.h.z * g.z, a.h.w * g.w);
            if (V & 2) buf_st4_bf16(t.q, t.vq + i * t.sq, 0, qv);
            else buf_st4(t.q, t.vq + i * t.sq, 0, qv);
        }
    }
};
// (drp / dh through the fixed instruction sequences cb_drp / cb_dh of kernels.h since round 4: left to -ffp-contract, the two-launch
// kernel and the generated-operand kernel of the bf16x3 arithmetic contracted `v R + p d Z` differently -- last-bit differences between
// two paths that tests/test_gpu_ops.py compares bit for bit)
struct EpiDgrad1F {
    EpiDgrad1 e;
    __device__ __forceinline__ void operator()(long m, int c, float v) const {
        long node = m / e.T;
        int t = (int)(m - node * e.T);
        float hv = e.h[m * e.C + c];
        float Z = e.ZR[m * (2L * e.C) + c];
        float R = e.ZR[m * (2L * e.C) + e.C + c];
        e.dzr[m * (2L * e.C) + e.C + c] = cb_drp(v, hv, R);
        e.dh[m * e.C + c] = cb_dh(v, R, e.probs[t], e.dOH[node * e.C + c], Z);
    }
    static constexpr int ROUND_ROWS = 8;
    struct Aux { float4 h, Z, R, d; float p; };
    __device__ __forceinline__ Aux load(long m, int c) const {
        const long node = m / e.T;
        Aux a;
        a.p = e.probs[(int)(m - node * e.T)];
        a.h = ld4(e.h + m * e.C + c);
        a.Z = ld4(e.ZR + m * (2L * e.C) + c);
        a.R = ld4(e.ZR + m * (2L * e.C) + e.C + c);
        a.d = ld4(e.dOH + node * e.C + c);
        return a;
    }
    __device__ __forceinline__ void apply(long m, int c, float4 v, const Aux& a) const {
#define F_(k) cb_drp(v.k, a.h.k, a.R.k)
        if (e.dzr_bf16) st4_bf16(e.dzr, m * (2L * e.C) + e.C + c, REGT_V4(F_));
        else st4(e.dzr + m * (2L * e.C) + e.C + c, REGT_V4(F_));
#undef F_
#define F_(k) cb_dh(v.k, a.R.k, a.p, a.d.k, a.Z.k)
        st4(e.dh + m * e.C + c, REGT_V4(F_));
#undef F_
    }
    // variants: bit 0 = dzr stored as bf16.  The row -> (node, period) map needs an integer division: it is done once
    // per tile row (vrow, 128 threads, kept in LDS) instead of once per row slot of every thread.
    static constexpr int NVAR = 2;
    static constexpr bool HAS_ROWTAB = true;
    struct Col {};
    struct Tile { __amdgpu_buffer_rsrc_t h, zr, d, dzr, dh; int vc, vzr, sc, szr, vd, vdzr, sdzr, rstep; const EpiRowEnt* rt; };
    __device__ __forceinline__ int variant(int) const { return e.dzr_bf16 ? 1 : 0; }
    __device__ __forceinline__ EpiRowEnt vrow(long m) const {
        const long node = m / e.T;
        return EpiRowEnt{(int)(node * e.C * 4), e.probs[(int)(m - node * e.T)]};
    }
    template <int V> __device__ __forceinline__ Col vcol(int) const { return Col{}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.h = buf_srd(e.h + g.m0 * e.C + g.n0);
        t.dh = buf_srd(e.dh + g.m0 * e.C + g.n0);
        t.zr = buf_srd(e.ZR + g.m0 * (2L * e.C) + g.n0);
        t.d = buf_srd(e.dOH + g.n0);
        const long o = g.m0 * (2L * e.C) + e.C + g.n0;
        t.dzr = buf_srd((V & 1) ? reinterpret_cast<const char*>(e.dzr) + 2 * o : reinterpret_cast<const char*>(e.dzr) + 4 * o);
        t.vc = (g.rr * e.C + g.c) * 4;
        t.sc = g.step * e.C * 4;
        t.vzr = 2 * t.vc - g.c * 4;
        t.szr = 2 * t.sc;
        t.vd = g.c * 4;
        t.vdzr = (V & 1) ? t.vzr >> 1 : t.vzr;
        t.sdzr = (V & 1) ? t.szr >> 1 : t.szr;
        t.rt = g.rowtab + g.rr;
        t.rstep = g.step;
        return t;
    }
    struct VAux { float4 h, Z, R, d; float p; };
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        const EpiRowEnt re = t.rt[i * t.rstep];
        VAux a;
        a.p = re.p;
        a.h = buf_ld4(t.h, t.vc, i * t.sc);
        a.Z = buf_ld4(t.zr, t.vzr, i * t.szr);
        a.R = buf_ld4(t.zr, t.vzr, i * t.szr + e.C * 4);
        a.d = buf_ld4(t.d, t.vd + re.off, 0);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, float4 v, const Col&, const VAux& a) const {
#define F_(k) cb_drp(v.k, a.h.k, a.R.k)
        if (V & 1) buf_st4_bf16(t.dzr, t.vdzr + i * t.sdzr, 0, REGT_V4(F_));
        else buf_st4(t.dzr, t.vdzr + i * t.sdzr, 0, REGT_V4(F_));
#undef F_
#define F_(k) cb_dh(v.k, a.R.k, a.p, a.d.k, a.Z.k)
        buf_st4(t.dh, t.vc + i * t.sc, 0, REGT_V4(F_));
#undef F_
    }
};
// EpiDgrad1F behind a GENERATED left operand (SplitCore::run_u_gen): with H~ at hand as well, the epilogue also forms
// dzp = g (h - H~) Z (1 - Z) -> dzr[m, c] and the row's partial attention dot <dOH[node], Z h + (1 - Z) H~> over the tile's
// 128 columns (summed over a half wave: the 32 lanes that share a row) -> rowdot[m * parts + column tile]; cell_bwd_kernel's
// three outputs without its pass over Z, h, H~.  fp32 arrays only.
struct EpiDgrad1GenF {
    EpiDgrad1 e;
    int parts;              // C / 128 column tiles
    static constexpr int ROUND_ROWS = 8;
    // sum over the 32 lanes of a half wave (the lanes that share a tile row), in a fixed order, on the VALU's DPP path -- no LDS
    // round trips: quads, 8-lane halves, 16-lane rows, then row 0's total into row 1 (row 2's into row 3).  The total ends up in
    // lanes 16-31 / 48-63; lane 31 / 63 stores it.
    __device__ __forceinline__ static float half_wave_sum(float s) {
#define REGT_DPP_ADD(ctrl, rmask) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), ctrl, rmask, 0xF, true))
        REGT_DPP_ADD(0xB1, 0xF);     // quad_perm [1, 0, 3, 2]
        REGT_DPP_ADD(0x4E, 0xF);     // quad_perm [2, 3, 0, 1]
        REGT_DPP_ADD(0x141, 0xF);    // row_half_mirror
        REGT_DPP_ADD(0x140, 0xF);    // row_mirror
        REGT_DPP_ADD(0x142, 0xA);    // row_bcast:15 into rows 1 and 3
#undef REGT_DPP_ADD
        return s;
    }
    static constexpr int SUM_LANE = 31;
    static constexpr int EPI_AUX_BYTES = 192;     // two rounds of two rows in flight (this kernel runs two workgroups per CU: 256 VGPRs)
    struct Aux { float4 h, Z, R, d, t; float p; };
    __device__ __forceinline__ Aux load(long m, int c) const {
        const long node = m / e.T;
        Aux a;
        a.p = e.probs[(int)(m - node * e.T)];
        a.h = ld4(e.h + m * e.C + c);
        a.Z = ld4(e.ZR + m * (2L * e.C) + c);
        a.R = ld4(e.ZR + m * (2L * e.C) + e.C + c);
        a.d = ld4(e.dOH + node * e.C + c);
        a.t = ld4(e.Ht + m * e.C + c);
        return a;
    }
#define REGT_GEN_DOT(a) ((a.d.x * (a.Z.x * a.h.x + (1.0f - a.Z.x) * a.t.x) + a.d.y * (a.Z.y * a.h.y + (1.0f - a.Z.y) * a.t.y)) + \
                         (a.d.z * (a.Z.z * a.h.z + (1.0f - a.Z.z) * a.t.z) + a.d.w * (a.Z.w * a.h.w + (1.0f - a.Z.w) * a.t.w)))
    __device__ __forceinline__ void apply(long m, int c, float4 v, const Aux& a) const {
#define F_(k) cb_drp(v.k, a.h.k, a.R.k)
        st4(e.dzr + m * (2L * e.C) + e.C + c, REGT_V4(F_));
#undef F_
#define F_(k) cb_dh(v.k, a.R.k, a.p, a.d.k, a.Z.k)
        st4(e.dh + m * e.C + c, REGT_V4(F_));
#undef F_
#define F_(k) cb_dzp(__fmul_rn(a.p, a.d.k), a.h.k, a.t.k, a.Z.k)
        st4(e.dzr + m * (2L * e.C) + c, REGT_V4(F_));
#undef F_
        // (guarded path of a partial tile: the 32 lanes of a row take this branch together -- rows are uniform per half wave)
        const float s = half_wave_sum(REGT_GEN_DOT(a));
        if ((threadIdx.x & 31) == SUM_LANE) e.rowdot[m * parts + c / GBN] = s;
    }
    static constexpr int NVAR = 1;
    static constexpr bool HAS_ROWTAB = true;
    struct Col {};
    struct Tile { __amdgpu_buffer_rsrc_t h, zr, d, dzr, dh, t, rd; int vc, vzr, sc, szr, vd, vrd, srd, rstep; const EpiRowEnt* rt; };
    __device__ __forceinline__ int variant(int) const { return 0; }
    __device__ __forceinline__ EpiRowEnt vrow(long m) const {
        const long node = m / e.T;
        return EpiRowEnt{(int)(node * e.C * 4), e.probs[(int)(m - node * e.T)]};
    }
    template <int V> __device__ __forceinline__ Col vcol(int) const { return Col{}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.h = buf_srd(e.h + g.m0 * e.C + g.n0);
        t.t = buf_srd(e.Ht + g.m0 * e.C + g.n0);
        t.dh = buf_srd(e.dh + g.m0 * e.C + g.n0);
        t.zr = buf_srd(e.ZR + g.m0 * (2L * e.C) + g.n0);
        t.d = buf_srd(e.dOH + g.n0);
        t.dzr = buf_srd(e.dzr + g.m0 * (2L * e.C) + g.n0);
        t.rd = buf_srd(e.rowdot + g.m0 * parts + g.n0 / GBN);
        t.vc = (g.rr * e.C + g.c) * 4;
        t.sc = g.step * e.C * 4;
        t.vzr = 2 * t.vc - g.c * 4;
        t.szr = 2 * t.sc;
        t.vd = g.c * 4;
        t.vrd = g.rr * parts * 4;
        t.srd = g.step * parts * 4;
        t.rt = g.rowtab + g.rr;
        t.rstep = g.step;
        return t;
    }
    struct VAux { float4 h, Z, R, d, t; float p; };
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        const EpiRowEnt re = t.rt[i * t.rstep];
        VAux a;
        a.p = re.p;
        a.h = buf_ld4(t.h, t.vc, i * t.sc);
        a.Z = buf_ld4(t.zr, t.vzr, i * t.szr);
        a.R = buf_ld4(t.zr, t.vzr, i * t.szr + e.C * 4);
        a.d = buf_ld4(t.d, t.vd + re.off, 0);
        a.t = buf_ld4(t.t, t.vc, i * t.sc);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, float4 v, const Col&, const VAux& a) const {
#define F_(k) cb_drp(v.k, a.h.k, a.R.k)
        buf_st4(t.dzr, t.vzr + i * t.szr + e.C * 4, 0, REGT_V4(F_));
#undef F_
#define F_(k) cb_dh(v.k, a.R.k, a.p, a.d.k, a.Z.k)
        buf_st4(t.dh, t.vc + i * t.sc, 0, REGT_V4(F_));
#undef F_
#define F_(k) cb_dzp(__fmul_rn(a.p, a.d.k), a.h.k, a.t.k, a.Z.k)
        buf_st4(t.dzr, t.vzr + i * t.szr, 0, REGT_V4(F_));
#undef F_
        const float s = half_wave_sum(REGT_GEN_DOT(a));
        if ((threadIdx.x & 31) == SUM_LANE) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), t.rd, t.vrd + i * t.srd, 0, 0);
    }
#undef REGT_GEN_DOT
};
struct EpiDgrad2F {
    EpiDgrad2 e;
    __device__ __forceinline__ void operator()(long m, int c, float v) const {
        long i = m * e.C + c;
        float d = e.dh[i] + v;
        if (e.act == ACT_LRELU) d = e.h[i] > 0.f ? d : d * e.slope;
        e.dh[i] = d;
    }
    static constexpr int ROUND_ROWS = 16;
    struct Aux { float4 d, h; };
    __device__ __forceinline__ Aux load(long m, int c) const {
        Aux a;
        a.d = ld4(e.dh + m * e.C + c);
        a.h = e.act == ACT_LRELU ? ld4(e.h + m * e.C + c) : make_float4(1, 1, 1, 1);
        return a;
    }
    __device__ __forceinline__ void apply(long m, int c, float4 v, const Aux& a) const {
#define F_(k) ((a.d.k + v.k) * (a.h.k > 0.f ? 1.0f : e.slope))
        st4(e.dh + m * e.C + c, REGT_V4(F_));
#undef F_
    }
    // variants: bit 0 = leaky-relu derivative (reads h)
    static constexpr int NVAR = 2;
    static constexpr bool HAS_ROWTAB = false;
    struct Col {};
    struct VAux { float4 d, h; };
    struct Tile { __amdgpu_buffer_rsrc_t dh, h; int v, s; };
    __device__ __forceinline__ int variant(int) const { return e.act == ACT_LRELU ? 1 : 0; }
    template <int V> __device__ __forceinline__ Col vcol(int) const { return Col{}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.dh = buf_srd(e.dh + g.m0 * e.C + g.n0);
        if (V & 1) t.h = buf_srd(e.h + g.m0 * e.C + g.n0);
        t.v = (g.rr * e.C + g.c) * 4;
        t.s = g.step * e.C * 4;
        return t;
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        VAux a;
        a.d = buf_ld4(t.dh, t.v, i * t.s);
        if (V & 1) a.h = buf_ld4(t.h, t.v, i * t.s);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, float4 v, const Col&, const VAux& a) const {
#define F_(k) ((a.d.k + v.k) * ((V & 1) ? (a.h.k > 0.f ? 1.0f : e.slope) : 1.0f))
        buf_st4(t.dh, t.v + i * t.s, 0, REGT_V4(F_));
#undef F_
    }
};
struct EpiMaskAddF {
    EpiMaskAdd e;
    __device__ __forceinline__ void operator()(long m, int c, float v) const {
        float o = e.mask[m * e.ldm + c] > 0.f ? v : 0.f;
        if (e.add) o += e.add[m * e.ldadd + c];
        e.out[m * e.ldo + c] = o;
    }
    static constexpr int ROUND_ROWS = 16;
    struct Aux { float4 mk, ad; };
    __device__ __forceinline__ Aux load(long m, int c) const {
        Aux a;
        a.mk = ld4(e.mask + m * e.ldm + c);
        a.ad = e.add ? ld4(e.add + m * e.ldadd + c) : make_float4(0, 0, 0, 0);
        return a;
    }
    __device__ __forceinline__ void apply(long m, int c, float4 v, const Aux& a) const {
#define F_(k) ((a.mk.k > 0.f ? v.k : 0.f) + a.ad.k)
        st4(e.out + m * e.ldo + c, REGT_V4(F_));
#undef F_
    }
    // variants: bit 0 = an addend is given
    static constexpr int NVAR = 2;
    static constexpr bool HAS_ROWTAB = false;
    struct Col {};
    struct VAux { float4 mk, ad; };
    struct Tile { __amdgpu_buffer_rsrc_t out, mask, add; int vo, vm, va, so, sm, sa; };
    __device__ __forceinline__ int variant(int) const {
        return (e.ldo >= (1L << 22) || e.ldm >= (1L << 22) || (e.add && e.ldadd >= (1L << 22))) ? -1 : (e.add ? 1 : 0);    // 32-bit tile offsets
    }
    template <int V> __device__ __forceinline__ Col vcol(int) const { return Col{}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.out = buf_srd(e.out + g.m0 * e.ldo + g.n0);
        t.mask = buf_srd(e.mask + g.m0 * e.ldm + g.n0);
        if (V & 1) t.add = buf_srd(e.add + g.m0 * e.ldadd + g.n0);
        t.vo = (g.rr * (int)e.ldo + g.c) * 4; t.so = g.step * (int)e.ldo * 4;
        t.vm = (g.rr * (int)e.ldm + g.c) * 4; t.sm = g.step * (int)e.ldm * 4;
        t.va = (g.rr * (int)e.ldadd + g.c) * 4; t.sa = g.step * (int)e.ldadd * 4;
        return t;
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        VAux a;
        a.mk = buf_ld4(t.mask, t.vm, i * t.sm);
        a.ad = (V & 1) ? buf_ld4(t.add, t.va, i * t.sa) : make_float4(0, 0, 0, 0);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, float4 v, const Col&, const VAux& a) const {
#define F_(k) ((a.mk.k > 0.f ? v.k : 0.f) + a.ad.k)
        buf_st4(t.out, t.vo + i * t.so, 0, REGT_V4(F_));
#undef F_
    }
};

// ---- 8-column-per-thread epilogues (bf16-operand core with bf16 STORAGE of the M x C activations) ------------------------
// A thread owns 8 consecutive columns of a row: 16 bytes of a bf16 array, two float4 of an fp32 one -- every access of the
// epilogue stays 16 bytes wide whichever format an array has (8-byte accesses run at about half the rate).
// Functor interface: ColAux col(c) once per thread (bias), Aux load(m, c) per row (the first round is requested before the
// accumulators are staged), apply(m, c, v, col, aux).  `bf` flags are wave-uniform.
struct F8 { float4 lo, hi; };
__device__ __forceinline__ F8 ld8(const void* base, long elem, int bf) {
    F8 r;
    if (bf) {
        const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + 2 * elem);
        r.lo = widen_bf16x4(raw.x, raw.y);
        r.hi = widen_bf16x4(raw.z, raw.w);
    } else {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + 4 * elem);
        r.lo = p[0];
        r.hi = p[1];
    }
    return r;
}
__device__ __forceinline__ void st8(void* base, long elem, const F8& v, int bf) {
    if (bf) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t a = {v.lo.x, v.lo.y}, b = {v.lo.z, v.lo.w}, c = {v.hi.x, v.hi.y}, d = {v.hi.z, v.hi.w};
        uint4 raw;
        raw.x = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t));
        raw.y = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_t));
        raw.z = __builtin_bit_cast(unsigned, __builtin_convertvector(c, bf16x2_t));
        raw.w = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2_t));
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + 2 * elem) = raw;
    } else {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + 4 * elem);
        p[0] = v.lo;
        p[1] = v.hi;
    }
}
#define REGT_F8(expr) F8{make_float4(expr(lo.x), expr(lo.y), expr(lo.z), expr(lo.w)), make_float4(expr(hi.x), expr(hi.y), expr(hi.z), expr(hi.w))}
// 16 bytes = 8 bf16 through a buffer descriptor (straight-line variants; stores keep soffset = 0, see the note on top)
__device__ __forceinline__ u32x4_t buf_ld16(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }
__device__ __forceinline__ F8 widen8(u32x4_t raw) { return F8{widen_bf16x4(raw.x, raw.y), widen_bf16x4(raw.z, raw.w)}; }
__device__ __forceinline__ void buf_st8_bf16(__amdgpu_buffer_rsrc_t r, int voff, const F8& v) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t a = {v.lo.x, v.lo.y}, b = {v.lo.z, v.lo.w}, c = {v.hi.x, v.hi.y}, d = {v.hi.z, v.hi.w};
    const u32x4_t raw = {__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_t)),
                         __builtin_bit_cast(unsigned, __builtin_convertvector(c, bf16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2_t))};
    __builtin_amdgcn_raw_buffer_store_b128(raw, r, voff, 0, 0);
}

struct EpiBiasAct8F {
    typedef F8 Vec;
    EpiBiasAct e;
    static constexpr int ROUND_ROWS = 4;
    struct ColAux { F8 b; };
    struct Aux {};
    __device__ __forceinline__ float act(float v) const {
        if (e.act == ACT_SIGMOID) return fast_sigmoid(v);
        if (e.act == ACT_TANH) return fast_tanh(v);
        const float ns = e.act == ACT_NONE ? 1.0f : (e.act == ACT_LRELU ? e.slope : 0.0f);
        return v > 0.f ? v : v * ns;
    }
    __device__ __forceinline__ ColAux col(int c) const {
        return ColAux{e.bias ? ld8(e.bias, c, 0) : F8{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}};
    }
    __device__ __forceinline__ Aux load(long, int) const { return Aux{}; }
    __device__ __forceinline__ void apply(long m, int c, const F8& v, const ColAux& ca, const Aux&) const {
#define F_(k) act(v.k + ca.b.k)
        st8(e.out, m * e.ldo + c, REGT_F8(F_), e.out_bf16);
#undef F_
    }
    // straight-line variants (bf16 output): 0 = none / leaky relu / relu, 1 = sigmoid, 2 = tanh
    static constexpr int NVAR = 3;
    static constexpr bool HAS_ROWTAB = false;
    typedef ColAux Col;
    struct VAux {};
    struct Tile { __amdgpu_buffer_rsrc_t out; int v, s; };
    __device__ __forceinline__ int variant(int) const { return (!e.out_bf16 || e.ldo >= (1L << 22)) ? -1 : (e.act == ACT_SIGMOID ? 1 : (e.act == ACT_TANH ? 2 : 0)); }
    template <int V> __device__ __forceinline__ Col vcol(int c) const { return col(c); }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        return Tile{buf_srd(reinterpret_cast<const char*>(e.out) + 2 * (g.m0 * e.ldo + g.n0)), (g.rr * (int)e.ldo + g.c) * 2, g.step * (int)e.ldo * 2};
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile&, int) const { return VAux{}; }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, const F8& v, const Col& ca, const VAux&) const {
        const float ns = e.act == ACT_NONE ? 1.0f : (e.act == ACT_LRELU ? e.slope : 0.0f);
#define F_(k) (V == 1 ? fast_sigmoid(v.k + ca.b.k) : V == 2 ? fast_tanh(v.k + ca.b.k) : ((v.k + ca.b.k) > 0.f ? (v.k + ca.b.k) : (v.k + ca.b.k) * ns))
        buf_st8_bf16(t.out, t.v + i * t.s, REGT_F8(F_));
#undef F_
    }
};
struct EpiGates8F {
    typedef F8 Vec;
    EpiGates e;
    static constexpr int ROUND_ROWS = 4;
    struct ColAux { F8 b; };
    struct Aux { F8 h; };
    __device__ __forceinline__ ColAux col(int c) const { return ColAux{ld8(e.bias, c, 0)}; }
    __device__ __forceinline__ Aux load(long m, int c) const {
        Aux a;
        if (c >= e.C) a.h = ld8(e.h, m * e.C + c - e.C, e.h_bf16);
        return a;
    }
    __device__ __forceinline__ void apply(long m, int c, const F8& v, const ColAux& ca, const Aux& a) const {
#define F_(k) fast_sigmoid(v.k + ca.b.k)
        const F8 g = REGT_F8(F_);
#undef F_
        st8(e.ZR, m * (2L * e.C) + c, g, e.zr_bf16);
        if (c >= e.C) {
#define F_(k) (a.h.k * g.k)
            st8(e.q, m * e.C + c - e.C, REGT_F8(F_), e.q_bf16);
#undef F_
        }
    }
    // straight-line variants (h, [Z|R] and q all stored as bf16): bit 0 = the tile holds r columns
    static constexpr int NVAR = 2;
    static constexpr bool HAS_ROWTAB = false;
    typedef ColAux Col;
    struct VAux { u32x4_t h; };
    struct Tile { __amdgpu_buffer_rsrc_t zr, h, q; int vzr, vh, szr, sh; };
    __device__ __forceinline__ int variant(int n0) const { return (e.C % GBN || !e.h_bf16 || !e.zr_bf16 || !e.q_bf16) ? -1 : (n0 >= e.C ? 1 : 0); }
    template <int V> __device__ __forceinline__ Col vcol(int c) const { return col(c); }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.zr = buf_srd(reinterpret_cast<const char*>(e.ZR) + 2 * (g.m0 * (2L * e.C) + g.n0));
        t.vzr = (g.rr * 2 * e.C + g.c) * 2;
        t.szr = g.step * 2 * e.C * 2;
        if (V & 1) {
            const long o = g.m0 * e.C + g.n0 - e.C;
            t.h = buf_srd(reinterpret_cast<const char*>(e.h) + 2 * o);
            t.q = buf_srd(reinterpret_cast<const char*>(e.q) + 2 * o);
            t.vh = (g.rr * e.C + g.c) * 2;
            t.sh = g.step * e.C * 2;
        }
        return t;
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        VAux a;
        if (V & 1) a.h = buf_ld16(t.h, t.vh, i * t.sh);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, const F8& v, const Col& ca, const VAux& a) const {
#define F_(k) fast_sigmoid(v.k + ca.b.k)
        const F8 g = REGT_F8(F_);
#undef F_
        buf_st8_bf16(t.zr, t.vzr + i * t.szr, g);
        if (V & 1) {
            const F8 h = widen8(a.h);
#define F_(k) (h.k * g.k)
            buf_st8_bf16(t.q, t.vh + i * t.sh, REGT_F8(F_));
#undef F_
        }
    }
};
struct EpiDgrad18F {
    typedef F8 Vec;
    EpiDgrad1 e;
    static constexpr int ROUND_ROWS = 1;      // 33 registers of auxiliary operands per row under the 168-VGPR cap
    struct ColAux {};
    struct Aux { F8 h, Z, R, d; float p; };
    __device__ __forceinline__ ColAux col(int) const { return ColAux{}; }
    __device__ __forceinline__ Aux load(long m, int c) const {
        const long node = m / e.T;
        Aux a;
        a.p = e.probs[(int)(m - node * e.T)];
        a.h = ld8(e.h, m * e.C + c, e.h_bf16);
        a.Z = ld8(e.ZR, m * (2L * e.C) + c, e.zr_bf16);
        a.R = ld8(e.ZR, m * (2L * e.C) + e.C + c, e.zr_bf16);
        a.d = ld8(e.dOH, node * e.C + c, 0);
        return a;
    }
    __device__ __forceinline__ void apply(long m, int c, const F8& v, const ColAux&, const Aux& a) const {
#define F_(k) cb_drp(v.k, a.h.k, a.R.k)
        st8(e.dzr, m * (2L * e.C) + e.C + c, REGT_F8(F_), e.dzr_bf16);
#undef F_
#define F_(k) cb_dh(v.k, a.R.k, a.p, a.d.k, a.Z.k)
        st8(e.dh, m * e.C + c, REGT_F8(F_), e.dh_bf16);
#undef F_
    }
    // straight-line variant (h, [Z|R], dzr, dh all stored as bf16); row -> (node, period) through the LDS row table
    static constexpr int NVAR = 1;
    static constexpr bool HAS_ROWTAB = true;
    typedef ColAux Col;
    struct VAux { u32x4_t h, Z, R; float4 d0, d1; float p; };
    struct Tile { __amdgpu_buffer_rsrc_t h, zr, d, dzr, dh; int vc, vzr, sc, szr, vd, rstep; const EpiRowEnt* rt; };
    __device__ __forceinline__ int variant(int) const { return (e.h_bf16 && e.zr_bf16 && e.dzr_bf16 && e.dh_bf16) ? 0 : -1; }
    __device__ __forceinline__ EpiRowEnt vrow(long m) const {
        const long node = m / e.T;
        return EpiRowEnt{(int)(node * e.C * 4), e.probs[(int)(m - node * e.T)]};
    }
    template <int V> __device__ __forceinline__ Col vcol(int) const { return Col{}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.h = buf_srd(reinterpret_cast<const char*>(e.h) + 2 * (g.m0 * e.C + g.n0));
        t.dh = buf_srd(reinterpret_cast<const char*>(e.dh) + 2 * (g.m0 * e.C + g.n0));
        t.zr = buf_srd(reinterpret_cast<const char*>(e.ZR) + 2 * (g.m0 * (2L * e.C) + g.n0));
        t.dzr = buf_srd(reinterpret_cast<const char*>(e.dzr) + 2 * (g.m0 * (2L * e.C) + e.C + g.n0));
        t.d = buf_srd(e.dOH + g.n0);
        t.vc = (g.rr * e.C + g.c) * 2;
        t.sc = g.step * e.C * 2;
        t.vzr = (g.rr * 2 * e.C + g.c) * 2;
        t.szr = 2 * t.sc;
        t.vd = g.c * 4;
        t.rstep = g.step;
        t.rt = g.rowtab + g.rr;
        return t;
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        const EpiRowEnt re = t.rt[i * t.rstep];
        VAux a;
        a.p = re.p;
        a.h = buf_ld16(t.h, t.vc, i * t.sc);
        a.Z = buf_ld16(t.zr, t.vzr, i * t.szr);
        a.R = buf_ld16(t.zr, t.vzr, i * t.szr + e.C * 2);
        a.d0 = buf_ld4(t.d, t.vd + re.off, 0);
        a.d1 = buf_ld4(t.d, t.vd + re.off, 16);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, const F8& v, const Col&, const VAux& a) const {
        const F8 h = widen8(a.h), Z = widen8(a.Z), R = widen8(a.R), d = F8{a.d0, a.d1};
#define F_(k) cb_drp(v.k, h.k, R.k)
        buf_st8_bf16(t.dzr, t.vzr + i * t.szr, REGT_F8(F_));
#undef F_
#define F_(k) cb_dh(v.k, R.k, a.p, d.k, Z.k)
        buf_st8_bf16(t.dh, t.vc + i * t.sc, REGT_F8(F_));
#undef F_
    }
};
struct EpiDgrad28F {
    typedef F8 Vec;
    EpiDgrad2 e;
    static constexpr int ROUND_ROWS = 4;
    struct ColAux {};
    struct Aux { F8 d, h; };
    __device__ __forceinline__ ColAux col(int) const { return ColAux{}; }
    __device__ __forceinline__ Aux load(long m, int c) const {
        Aux a;
        a.d = ld8(e.dh, m * e.C + c, e.dh_bf16);
        if (e.act == ACT_LRELU) a.h = ld8(e.h, m * e.C + c, e.h_bf16);
        else a.h = F8{make_float4(1, 1, 1, 1), make_float4(1, 1, 1, 1)};
        return a;
    }
    __device__ __forceinline__ void apply(long m, int c, const F8& v, const ColAux&, const Aux& a) const {
#define F_(k) cb_ds(a.d.k, v.k, a.h.k > 0.f ? 1.0f : e.slope)
        st8(e.dh, m * e.C + c, REGT_F8(F_), e.dh_bf16);
#undef F_
    }
    // straight-line variants (dh and h stored as bf16): bit 0 = leaky-relu derivative (reads h)
    static constexpr int NVAR = 2;
    static constexpr bool HAS_ROWTAB = false;
    typedef ColAux Col;
    struct VAux { u32x4_t d, h; };
    struct Tile { __amdgpu_buffer_rsrc_t dh, h; int v, s; };
    __device__ __forceinline__ int variant(int) const { return (e.dh_bf16 && e.h_bf16) ? (e.act == ACT_LRELU ? 1 : 0) : -1; }
    template <int V> __device__ __forceinline__ Col vcol(int) const { return Col{}; }
    template <int V> __device__ __forceinline__ Tile vtile(const EpiGeom& g) const {
        Tile t;
        t.dh = buf_srd(reinterpret_cast<const char*>(e.dh) + 2 * (g.m0 * e.C + g.n0));
        if (V & 1) t.h = buf_srd(reinterpret_cast<const char*>(e.h) + 2 * (g.m0 * e.C + g.n0));
        t.v = (g.rr * e.C + g.c) * 2;
        t.s = g.step * e.C * 2;
        return t;
    }
    template <int V> __device__ __forceinline__ VAux vload(const Tile& t, int i) const {
        VAux a;
        a.d = buf_ld16(t.dh, t.v, i * t.s);
        if (V & 1) a.h = buf_ld16(t.h, t.v, i * t.s);
        return a;
    }
    template <int V> __device__ __forceinline__ void vapply(const Tile& t, int i, const F8& v, const Col&, const VAux& a) const {
        const F8 d = widen8(a.d);
        F8 h = d;
        if (V & 1) h = widen8(a.h);
#define F_(k) cb_ds(d.k, v.k, (V & 1) ? (h.k > 0.f ? 1.0f : e.slope) : 1.0f)
        buf_st8_bf16(t.dh, t.v + i * t.s, REGT_F8(F_));
#undef F_
    }
};

template <class EpiF>
__global__ __launch_bounds__(256, 2) void gemm_flat_kernel(GemmSegs S, long M, int N, EpiF epi, int vec) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles_n = (N + GBN - 1) / GBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    RowMap rm{m0, 1, (int)((M - m0) < GBM ? (M - m0) : GBM)};
    GemmCore core(S, rm, n0, N, lds);
    core.find_regions();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    core.run(acc);
    if (vec) core.for_each_vec(acc, epi);
    else core.for_each(acc, [&](int r, int c, float v) { epi(m0 + r, c, v); });
}

// Fast variant: straight-line K loop with interleaved loads (gemm_fast.h).  Requires vector-aligned
// operands, one B layout for all segments and N % 4 == 0; everything else takes the generic kernel.
template <class EpiF, class Core>
__global__ __launch_bounds__(256, 2) void gemm_flat_fast_kernel(GemmSegs S, long M, int N, EpiF epi, int relu_a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles_n = (N + GBN - 1) / GBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    RowMap rm{m0, 1, (int)((M - m0) < GBM ? (M - m0) : GBM)};
    Core core(S, rm, n0, N, lds);
    core.fill_rowtab(epi);
    core.plan();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    WG_TRACE_T(t_a);
    core.run(acc, relu_a != 0);
    WG_TRACE_T(t_b);
    core.for_each_vec(acc, epi);
    WG_TRACE_END(N, t_a, t_b);
}

// Split core with the compact LDS layout (two stages + table, epilogue in two 64-row halves): three workgroups per CU.
// NP = 0: fp32 planes on the fp32 MFMA; NP = 3: exact 3-way bf16 split (fp32-level accuracy); NP = 1: plain bf16 operands,
// fp32 accumulate (REGT_GEMM_MODE=bf16).
// (Persistent workgroups walking the tiles were tried and measured slower: the turn-around between two tiles of a slot
// goes from 7 us to 0.6 us, but the three workgroups of a CU then run in step -- all in their K loops, then all in their
// epilogues -- and the tile loop costs registers; gates GEMM 3.40 ms against 3.19 ms.  DESIGN.md section 6.)
template <class EpiF, bool REGION, int NP>
__global__ __launch_bounds__(256, 3) void gemm_flat_split_kernel(GemmSegs S, long M, int N, EpiF epi, int relu_a, int uniform) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles_n = (N + GBN - 1) / GBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    RowMap rm{m0, 1, (int)((M - m0) < GBM ? (M - m0) : GBM)};
    WG_MARK(6);
    SplitCore<REGION, NP> core(S, rm, n0, N, lds, true);
    WG_MARK(7);
    core.fill_rowtab(epi);
    const bool uni = uniform != 0;                 // scalar slab descriptors (host: uniform_ok): no iteration table
    if (!uni) core.plan();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    WG_TRACE_T(t_a);
    if (uni) core.run_uniform(acc, relu_a != 0);
    else core.run(acc, relu_a != 0);
    WG_TRACE_T(t_b);
    core.for_each_vec_halves(acc, epi);
    WG_TRACE_END(N, t_a, t_b);
}

// dgrad_candidate with its left operand generated on the way (SplitCore::run_u_gen + EpiDgrad1GenF): cell_bwd_kernel and the candidate
// data gradient in one launch.  fp32 arithmetic; two workgroups per CU (three operand arrays in flight per A slot).
template <int NP>     // 0: fp32 MFMA; 3: exact bf16x3 split (REGT_GEMM_MODE=bf16x3) -- fp32 storage either way
__global__ __launch_bounds__(256, 2) void gemm_dgrad1_gen_kernel(GemmSegs S, long M, int N, EpiDgrad1GenF epi) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles_n = (N + GBN - 1) / GBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    RowMap rm{m0, 1, (int)((M - m0) < GBM ? (M - m0) : GBM)};
    SplitCore<false, NP> core(S, rm, n0, N, lds, true);
    core.fill_rowtab(epi);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const typename SplitCore<false, NP>::AGen g{epi.e.ZR, epi.e.Ht, epi.e.dOH, epi.e.dhp, epi.e.C, (unsigned)epi.e.num_nodes * (unsigned)epi.e.C * 4u};
    core.run_u_gen(acc, g);
    core.for_each_vec_halves(acc, epi);
}

// bf16-operand core + 8-column epilogue (bf16 storage of the activations): same K loop as gemm_flat_split_kernel<.., 1>
template <class EpiF8, bool REGION>
__global__ __launch_bounds__(256, 3) void gemm_flat_split8_kernel(GemmSegs S, long M, int N, EpiF8 epi, int uniform) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles_n = (N + GBN - 1) / GBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    RowMap rm{m0, 1, (int)((M - m0) < GBM ? (M - m0) : GBM)};
    SplitCore<REGION, 1> core(S, rm, n0, N, lds, true);
    core.fill_rowtab(epi);
    const bool uni = uniform != 0;
    if (!uni) core.plan();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    WG_TRACE_T(t_a);
    if (uniform == 2) core.run_uniform_frag(acc, false);
    else if (uni) core.run_uniform(acc, false);
    else core.run(acc, false);
    WG_TRACE_T(t_b);
    core.for_each_vec8_halves(acc, epi);
    WG_TRACE_END(N, t_a, t_b);
}

// Small problems: 64 x 64 tiles (gemm_small.h) -- same operands, segments and epilogues, a quarter of the work per tile.
template <class EpiF, bool BT, bool REGION>
__global__ __launch_bounds__(256, 4) void gemm_flat_small_kernel(GemmSegs S, long M, int N, EpiF epi, int relu_a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles_n = (N + SM_B - 1) / SM_B;
    const long m0 = (long)(blockIdx.x / tiles_n) * SM_B;
    const int n0 = (blockIdx.x % tiles_n) * SM_B;
    RowMap rm{m0, 1, (int)((M - m0) < SM_B ? (M - m0) : SM_B)};
    SmallCore<BT, REGION> core(S, rm, n0, N, lds);
    core.plan();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    core.run(acc, relu_a != 0);
    core.for_each_vec(acc, epi);
}

// 0: fp32 MFMA (default).  1: exact 3-way bf16 split of both operands, six partial products on the bf16 matrix pipe
// (gemm_split.h) for the GEMMs whose B operand is stored [N][K].  2: plain bf16 operands (one product), fp32 accumulate --
// reduced precision, the arithmetic BASELINE configs[4] names.
static int g_gemm_mode = -1;
// per-call override (regt_dims.arith, set for the duration of one entry point on the calling thread by CallScope): two models of
// one process can run different arithmetics without touching the process default
static thread_local int t_gemm_mode = -1;
int gemm_mode_override(int mode) { const int prev = t_gemm_mode; t_gemm_mode = mode; return prev; }
int gemm_mode() {
    if (t_gemm_mode >= 0) return t_gemm_mode;
    if (g_gemm_mode < 0) {
        const char* e = getenv("REGT_GEMM_MODE");
        g_gemm_mode = 0;
        if (e && (!strcmp(e, "bf16x3") || !strcmp(e, "1"))) g_gemm_mode = 1;
        if (e && (!strcmp(e, "bf16") || !strcmp(e, "2"))) g_gemm_mode = 2;
    }
    return g_gemm_mode;
}
void set_gemm_mode(int m) { g_gemm_mode = (m == 1 || m == 2) ? m : 0; }
// REGT_FP32_CORE=wide: fp32 GEMMs on the 2-workgroup-per-CU core (gemm_fast.h) instead of the 3-workgroup one, for A/B timing
bool fp32_core_wide() {
    static int wide = -1;
    if (wide < 0) { const char* e = getenv("REGT_FP32_CORE"); wide = e && !strcmp(e, "wide") ? 1 : 0; }
    return wide == 1;
}

// the K loop may keep its slab descriptors in scalar registers (SplitCore::run_u): every K a multiple of the 32-k slab, byte offsets of a tile's rows within 31 bits.  REGT_GEMM_DESC=table forces the LDS table.
bool gemm_desc_table_forced() {
    static int force_table = -1;
    if (force_table < 0) { const char* e = getenv("REGT_GEMM_DESC"); force_table = e && !strcmp(e, "table") ? 1 : 0; }
    return force_table == 1;
}
// 0: LDS table; 1: scalar descriptors; 2: scalar descriptors and every segment's weights in fragment order (SEG_B_FRAG)
static int uniform_ok(const GemmSegs& S, long M) {
    if (gemm_desc_table_forced()) return 0;
    int frag = 0;
    long slabs = 0;
    for (int s = 0; s < S.nseg; ++s) {
        const GemmSeg& g = S.seg[s];
        if (g.K % GBK != 0 || g.K <= 0) return 0;
        if ((g.flags & SEG_REGION) && (g.flags & SEG_A_BF16)) return 0;       // bf16 rows are never region-masked
        if (g.lda * 4 * (GBM + 1) >= (1L << 31) || g.ldb * 4 * (GBN + 1) >= (1L << 31)) return 0;
        if (g.flags & SEG_B_FRAG) ++frag;
        // the bf16-operand core keeps the tile's slab descriptors in LDS, 64 at most (SplitCore::plan_u)
        const long reps = (g.flags & SEG_REGION) ? std::min<long>(S.num_regions > 0 ? S.num_regions : 1, GBM / (S.row_div > 0 ? S.row_div : 1) + 2)
                                                 : ((g.flags & SEG_REPEAT) ? g.nrep : 1);
        slabs += reps * (g.K / GBK);
    }
    if (gemm_mode() == 2 && slabs > 64) return 0;
    (void)M;
    if (frag == 0) return 1;
    return frag == S.nseg && gemm_mode() == 2 ? 2 : 0;
}

// 0: not eligible, else bit0 = BT, bit1 = has a region-masked segment, bit2 = relu on A
static int fast_class(const GemmSegs& S, int N, bool vec) {
    if (!vec || N % 4 != 0 || S.nseg < 1) return -1;
    int bt = -1, region = 0, relu = 0;
    long iters = 0;
    for (int s = 0; s < S.nseg; ++s) {
        const GemmSeg& g = S.seg[s];
        if (!(g.flags & SEG_VEC_A) || !(g.flags & SEG_VEC_B) || g.K % 4 != 0) return -1;
        // only the bf16-operand core reads bf16 rows: 16-byte loads of 8 k, never region-masked
        if ((g.flags & SEG_A_BF16) && (gemm_mode() != 2 || !(g.flags & SEG_BT) || g.K % 8 != 0 || g.lda % 8 != 0 ||
                                       (g.flags & (SEG_REGION | SEG_REPEAT | SEG_RELU_A)))) return -1;
        const int b = (g.flags & SEG_BT) ? 1 : 0;
        if (bt >= 0 && bt != b) return -1;
        bt = b;
        if (g.flags & SEG_REGION) region = 1;
        if (g.flags & SEG_RELU_A) relu = 1;
        if (b && g.nsplit < N && g.nsplit % GBN != 0) return -1;      // a column tile must not straddle B0 | B1
        if (g.lda >= (1L << 22) || g.ldb >= (1L << 22)) return -1;    // 32-bit byte offsets inside a tile
        if ((g.flags & SEG_REPEAT) && g.a_rep_stride % 4 != 0) return -1;
        // worst case of a region-masked segment: every node of a row tile lies in another region (node ids not sorted by
        // region) -- a 128-row tile of node-major rows meets at most 128 / row_div + 2 nodes; never more than all regions
        long reps = 1;
        if (g.flags & SEG_REGION) {
            const long by_rows = GBM / (S.row_div > 0 ? S.row_div : 1) + 2;
            reps = S.num_regions > 0 ? S.num_regions : 1;
            if (by_rows < reps) reps = by_rows;
        } else if (g.flags & SEG_REPEAT) {
            reps = g.nrep;
        }
        iters += (long)cdiv(g.K, GBK) * reps;
    }
    if (relu && S.nseg != 1) return -1;
    if (iters > G_MAX_ITERS) return -1;
    return bt | (region << 1) | (relu << 2);
}

template <class EpiF, class Core>
static int launch_fast_core(const GemmSegs& S, long M, int N, EpiF f, int relu, hipStream_t st) {
    long tiles = (long)cdiv(M, GBM) * cdiv(N, GBN);
    REGT_CHECK_ARG(tiles < (1L << 31), "gemm: too many tiles");
    static bool attr_done = false;
    if (int rc = set_lds_once(&gemm_flat_fast_kernel<EpiF, Core>, G_FAST_LDS_BYTES, &attr_done)) return rc;
    hipLaunchKernelGGL((gemm_flat_fast_kernel<EpiF, Core>), dim3((unsigned)tiles), dim3(256), G_FAST_LDS_BYTES, st, S, M, N, f,
                       relu);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
// fewer 128 x 128 tiles than this: the chip is mostly idle and one tile's latency is the kernel's duration
constexpr long SMALL_TILE_LIMIT = 128;

template <class EpiF, bool BT, bool REGION>
static int launch_fast(const GemmSegs& S, long M, int N, EpiF f, int relu, hipStream_t st) {
    if (gemm_mode() == 0 && (long)cdiv(M, GBM) * cdiv(N, GBN) < SMALL_TILE_LIMIT) {
        const long tiles = (long)cdiv(M, SM_B) * cdiv(N, SM_B);
        hipLaunchKernelGGL((gemm_flat_small_kernel<EpiF, BT, REGION>), dim3((unsigned)tiles), dim3(256), SM_LDS_BYTES, st, S, M, N, f,
                           relu);
        REGT_CHECK_LAUNCH();
        return REGT_OK;
    }
    if constexpr (BT) {
        if (gemm_mode() != 0 || !fp32_core_wide()) {
            const long tiles = (long)cdiv(M, GBM) * cdiv(N, GBN);
            REGT_CHECK_ARG(tiles < (1L << 31), "gemm: too many tiles");
            if (gemm_mode() == 0)
                hipLaunchKernelGGL((gemm_flat_split_kernel<EpiF, REGION, 0>), dim3((unsigned)tiles), dim3(256), SplitGeom<0>::LDS_BYTES, st, S, M, N, f, relu, uniform_ok(S, M));
            else if (gemm_mode() == 1)
                hipLaunchKernelGGL((gemm_flat_split_kernel<EpiF, REGION, 3>), dim3((unsigned)tiles), dim3(256), SplitGeom<3>::LDS_BYTES, st, S, M, N, f, relu, uniform_ok(S, M));
            else
                hipLaunchKernelGGL((gemm_flat_split_kernel<EpiF, REGION, 1>), dim3((unsigned)tiles), dim3(256), SplitGeom<1>::LDS_BYTES, st, S, M, N, f, relu, uniform_ok(S, M));
            REGT_CHECK_LAUNCH();
            return REGT_OK;
        }
    }
    return launch_fast_core<EpiF, FastCore<BT, REGION>>(S, M, N, f, relu, st);
}

// bf16-operand core with the 8-column epilogue: arrays of the epilogue may be stored as bf16
template <class EpiF8, bool REGION>
static int launch_split8(const GemmSegs& S, long M, int N, EpiF8 f, hipStream_t st) {
    REGT_CHECK_ARG(gemm_mode() == 2 && N % 8 == 0, "gemm: bf16-stored activations need REGT_GEMM_MODE=bf16 and N %% 8 == 0");
    for (int q = 0; q < S.nseg; ++q)
        REGT_CHECK_ARG(!(S.seg[q].flags & SEG_B_FRAG) || uniform_ok(S, M) == 2, "gemm: fragment-order weights need every segment in that order and K %% 32 == 0");
    const long tiles = (long)cdiv(M, GBM) * cdiv(N, GBN);
    REGT_CHECK_ARG(tiles < (1L << 31), "gemm: too many tiles");
    hipLaunchKernelGGL((gemm_flat_split8_kernel<EpiF8, REGION>), dim3((unsigned)tiles), dim3(256), SplitGeom<1>::LDS_BYTES, st, S, M, N, f, uniform_ok(S, M));
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

template <class EpiF>
static int launch_flat(const GemmSegs& S, long M, int N, EpiF f, bool vec, hipStream_t st) {
    REGT_CHECK_ARG(M > 0 && N > 0, "gemm: empty problem M=%ld N=%d", M, N);
    for (int q = 0; q < S.nseg; ++q)
        REGT_CHECK_ARG(!(S.seg[q].flags & SEG_A_BF16), "gemm: a bf16-stored operand needs the bf16-operand vector path");
    long tiles = (long)cdiv(M, GBM) * cdiv(N, GBN);
    REGT_CHECK_ARG(tiles < (1L << 31), "gemm: too many tiles");
    static bool attr_done = false;
    if (int rc = set_lds_once(&gemm_flat_kernel<EpiF>, G_LDS_BYTES, &attr_done)) return rc;
    hipLaunchKernelGGL(gemm_flat_kernel<EpiF>, dim3((unsigned)tiles), dim3(256), G_LDS_BYTES, st, S, M, N, f, vec ? 1 : 0);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}


int launch_gemm_bias_act(const GemmSegs& S, long M, int N, const EpiBiasAct& e, hipStream_t st) {
    const bool vec = N % 4 == 0 && e.ldo % 4 == 0 && a16(e.out) && a16(e.bias);
    const int fc = fast_class(S, N, vec);
    if (e.out_bf16) {
        REGT_CHECK_ARG(fc >= 0 && (fc & 1) && !(fc & 4) && e.ldo % 8 == 0, "bias/act gemm: bf16 output needs the bf16-operand vector path");
        if (fc & 2) return launch_split8<EpiBiasAct8F, true>(S, M, N, EpiBiasAct8F{e}, st);
        return launch_split8<EpiBiasAct8F, false>(S, M, N, EpiBiasAct8F{e}, st);
    }
    if (fc >= 0 && (fc & 1)) {
        if (fc & 2) return launch_fast<EpiBiasActF, true, true>(S, M, N, EpiBiasActF{e}, (fc >> 2) & 1, st);
        return launch_fast<EpiBiasActF, true, false>(S, M, N, EpiBiasActF{e}, (fc >> 2) & 1, st);
    }
    return launch_flat(S, M, N, EpiBiasActF{e}, vec, st);
}
int launch_gemm_gates(const GemmSegs& S, long M, int N, const EpiGates& e, hipStream_t st) {
    REGT_CHECK_ARG(N == 2 * e.C, "gates gemm expects N == 2C");
    const bool vec = e.C % 4 == 0 && a16(e.ZR) && a16(e.h) && a16(e.q) && a16(e.bias);
    if (e.h_bf16 || e.zr_bf16) {
        REGT_CHECK_ARG(fast_class(S, N, vec) == 1 && e.C % 8 == 0, "gates gemm: bf16-stored activations need the bf16-operand vector path");
        return launch_split8<EpiGates8F, false>(S, M, N, EpiGates8F{e}, st);
    }
    if (fast_class(S, N, vec) == 1) return launch_fast<EpiGatesF, true, false>(S, M, N, EpiGatesF{e}, 0, st);
    REGT_CHECK_ARG(!e.q_bf16, "gates gemm: bf16 storage of q needs the vector path");
    return launch_flat(S, M, N, EpiGatesF{e}, vec, st);
}
int launch_gemm_dgrad1(const GemmSegs& S, long M, int N, const EpiDgrad1& e, hipStream_t st) {
    REGT_CHECK_ARG(N == e.C, "dgrad1 gemm expects N == C");
    const bool vec = e.C % 4 == 0 && a16(e.h) && a16(e.ZR) && a16(e.dOH) && a16(e.dzr) && a16(e.dh);
    if (e.h_bf16 || e.zr_bf16 || e.dh_bf16) {
        REGT_CHECK_ARG(fast_class(S, N, vec) == 1 && e.C % 8 == 0, "dgrad1 gemm: bf16-stored activations need the bf16-operand vector path");
        return launch_split8<EpiDgrad18F, false>(S, M, N, EpiDgrad18F{e}, st);
    }
    if (fast_class(S, N, vec) == 1) return launch_fast<EpiDgrad1F, true, false>(S, M, N, EpiDgrad1F{e}, 0, st);
    REGT_CHECK_ARG(!e.dzr_bf16, "dgrad1 gemm: bf16 storage of dzr needs the bf16-operand vector path");
    if (fast_class(S, N, vec) == 0) return launch_fast<EpiDgrad1F, false, false>(S, M, N, EpiDgrad1F{e}, 0, st);
    return launch_flat(S, M, N, EpiDgrad1F{e}, vec, st);
}
// REGT_DGRAD1_GEN=0: keep cell_bwd + dgrad_candidate as two launches (A/B timing; same dhp / dzp / drp / dh to the bit, the
// attention gradient in another fixed summation order)
static int g_dgrad1_gen = -1;
static bool dgrad1_gen_wanted() {
    if (g_dgrad1_gen < 0) { const char* e = getenv("REGT_DGRAD1_GEN"); g_dgrad1_gen = e ? atoi(e) : 1; }
    return g_dgrad1_gen != 0;
}
int dgrad1_gen_option(int value) {       // regt_set_option("dgrad1_gen", v): returns the previous setting
    const int prev = dgrad1_gen_wanted() ? 1 : 0;
    g_dgrad1_gen = value ? 1 : 0;
    return prev;
}
bool gemm_dgrad1_gen_ok(long M, int C, int num_nodes) {
    return dgrad1_gen_wanted() && (gemm_mode() == 0 || gemm_mode() == 1) && !fp32_core_wide() && !gemm_desc_table_forced() && C % GBN == 0 && C % GBK == 0 &&
           (long)cdiv(M, GBM) * (C / GBN) >= SMALL_TILE_LIMIT && M < (1L << 31) && (long)num_nodes * C * 4 < (1L << 31) &&
           (long)C * 8 * (GBM + 1) < (1L << 31);
}
int launch_gemm_dgrad1_gen(const GemmSegs& S, long M, int N, const EpiDgrad1& e, hipStream_t st) {
    REGT_CHECK_ARG(N == e.C && S.nseg == 1 && S.seg[0].K == e.C && (S.seg[0].flags & SEG_BT) && (S.seg[0].flags & SEG_VEC_B) &&
                   !(S.seg[0].flags & (SEG_A_BF16 | SEG_B_FRAG | SEG_REGION | SEG_REPEAT)), "dgrad1 (generated operand): one [N][K] weight segment with K = C");
    REGT_CHECK_ARG(gemm_dgrad1_gen_ok(M, e.C, e.num_nodes), "dgrad1 (generated operand): shape / arithmetic not covered");
    REGT_CHECK_ARG(e.Ht && e.dhp && e.rowdot && !e.dzr_bf16 && !e.h_bf16 && !e.zr_bf16 && !e.dh_bf16, "dgrad1 (generated operand): fp32 arrays, all outputs given");
    REGT_CHECK_ARG(a16(e.h) && a16(e.ZR) && a16(e.dOH) && a16(e.dzr) && a16(e.dh) && a16(e.Ht) && a16(e.dhp), "dgrad1 (generated operand): 16-byte aligned arrays");
    const long tiles = (long)cdiv(M, GBM) * (N / GBN);
    REGT_CHECK_ARG(tiles < (1L << 31), "gemm: too many tiles");
    if (gemm_mode() == 1)
        hipLaunchKernelGGL(gemm_dgrad1_gen_kernel<3>, dim3((unsigned)tiles), dim3(256), SplitGeom<3>::LDS_BYTES, st, S, M, N, EpiDgrad1GenF{e, e.C / GBN});
    else
        hipLaunchKernelGGL(gemm_dgrad1_gen_kernel<0>, dim3((unsigned)tiles), dim3(256), SplitGeom<0>::LDS_BYTES, st, S, M, N, EpiDgrad1GenF{e, e.C / GBN});
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}
int launch_gemm_dgrad2(const GemmSegs& S, long M, int N, const EpiDgrad2& e, hipStream_t st) {
    REGT_CHECK_ARG(N == e.C, "dgrad2 gemm expects N == C");
    const bool vec = e.C % 4 == 0 && a16(e.dh) && a16(e.h);
    if (e.h_bf16 || e.dh_bf16) {
        REGT_CHECK_ARG(fast_class(S, N, vec) == 1 && e.C % 8 == 0, "dgrad2 gemm: bf16-stored activations need the bf16-operand vector path");
        return launch_split8<EpiDgrad28F, false>(S, M, N, EpiDgrad28F{e}, st);
    }
    if (fast_class(S, N, vec) == 1) return launch_fast<EpiDgrad2F, true, false>(S, M, N, EpiDgrad2F{e}, 0, st);
    if (fast_class(S, N, vec) == 0) return launch_fast<EpiDgrad2F, false, false>(S, M, N, EpiDgrad2F{e}, 0, st);
    return launch_flat(S, M, N, EpiDgrad2F{e}, vec, st);
}
int launch_gemm_mask_add(const GemmSegs& S, long M, int N, const EpiMaskAdd& e, hipStream_t st) {
    const bool vec = N % 4 == 0 && e.ldo % 4 == 0 && e.ldm % 4 == 0 && (!e.add || e.ldadd % 4 == 0) && a16(e.out) &&
                     a16(e.mask) && a16(e.add);
    if (fast_class(S, N, vec) == 0) return launch_fast<EpiMaskAddF, false, false>(S, M, N, EpiMaskAddF{e}, 0, st);
    return launch_flat(S, M, N, EpiMaskAddF{e}, vec, st);
}

// ---- candidate state, T loop inside the workgroup ------------------------------------------------
// Per period t: Ht = tanh(q_t Uh2^T + (A_hat x)_t Gh^T + ch) is stored for the backward pass, blended
// with the gate (Z*h + (1-Z)*Ht) and accumulated with the attention probability p_t in registers;
// the hidden state (N, C) is written once after the last period.
// Generic (scalar-epilogue) fallback of the candidate stage: one workgroup walks the T periods of a node tile.
__global__ __launch_bounds__(256, 1) void gemm_cand_kernel(CandArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles_n = (a.C + GBN - 1) / GBN;
    const long C = a.C;
    f32x16 acc[2][2];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int i0 = (bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    const int nvalid = (a.num_nodes - i0) < GBM ? (a.num_nodes - i0) : GBM;
    f32x16 oh[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) oh[i][j][r] = 0.f;
    for (int t = 0; t < a.T; ++t) {
        RowMap rm{(long)i0 * a.T + t, a.T, nvalid};
        GemmCore core(a.S, rm, n0, a.C, lds);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        core.run(acc);
        const float pt = a.probs[t];
        core.for_each2(acc, oh, [&](int r, int c, float v, float o) {
            long m = rm.grow(r);
            float ht = fast_tanh(v + a.bias[c]);
            a.Ht[m * C + c] = ht;
            float Z = a.ZR[m * 2 * C + c];
            float hv = a.h[m * C + c];
            return o + pt * (Z * hv + (1.0f - Z) * ht);
        });
    }
    RowMap rm{(long)i0, 1, nvalid};
    GemmCore core(a.S, rm, n0, a.C, lds);
    core.for_each(oh, [&](int r, int c, float v) { a.OH[(long)(i0 + r) * C + c] = v; });
}

// Vector path of the candidate stage: a FLAT GEMM over the (node*T + t) rows, same geometry and
// efficiency as the gate GEMM.  Rows of one node are adjacent, so the attention-weighted sum over the
// T periods is a segmented reduction inside the 128-row tile, done in LDS after the blend; a node
// that straddles two tiles (T <= 64 < 128, so never more than two) gets one partial sum from each,
// added atomically into the zero-initialised hidden state -- two addends commute, so the result is
// bit-reproducible.
template <class Core>
__global__ __launch_bounds__(256, 2) void gemm_cand_flat_kernel(CandArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = Core::BM, BN = Core::BN, KROW = Core::EKROW, TPR = Core::ETPR;
    const long C = a.C, M = (long)a.num_nodes * a.T;
    const int tiles_n = (a.C + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * BM;
    const int n0 = (bid % tiles_n) * BN;
    const RowMap rm{m0, 1, (int)((M - m0) < BM ? (M - m0) : BM)};
    Core core(a.S, rm, n0, a.C, lds);
    core.plan();
    typename Core::Acc acc;
    Core::zero(acc);
    core.run(acc, false);
    const int c = core.ecol();
    const int node0 = (int)(m0 / a.T);
    // epilogue rows in rounds of RR; the first round's Z / h rows are requested before the accumulators are staged through
    // LDS (their HBM latency overlaps the staging), see FastCore::for_each_vec
    constexpr int RR = Core::EROWS >= 8 ? 8 : Core::EROWS;
    float4 Z[RR], hv[RR];
    if (c < a.C) {
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            const int r = core.erow(j);
            if (r < rm.nvalid) {
                Z[j] = ld4(a.ZR + (m0 + r) * 2 * C + c);
                hv[j] = ld4(a.h + (m0 + r) * C + c);
            }
        }
    }
    core.stage(acc);
    if (c < a.C) {
        const float4 b = ld4(a.bias + c);
#pragma unroll
        for (int g = 0; g < Core::EROWS / RR; ++g) {
            if (g > 0) {
#pragma unroll
                for (int j = 0; j < RR; ++j) {
                    const int r = core.erow(RR * g + j);
                    if (r < rm.nvalid) {
                        Z[j] = ld4(a.ZR + (m0 + r) * 2 * C + c);
                        hv[j] = ld4(a.h + (m0 + r) * C + c);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                const int r = core.erow(RR * g + j);
                if (r < rm.nvalid) {
                    const long m = m0 + r;
                    const float pt = a.probs[(int)(m % a.T)];
                    const float4 v = core.eread(RR * g + j);
#define F_(k) fast_tanh(v.k + b.k)
                    const float4 ht = REGT_V4(F_);
#undef F_
                    st4(a.Ht + m * C + c, ht);
                    float4 o;
                    o.x = pt * (Z[j].x * hv[j].x + (1.0f - Z[j].x) * ht.x);
                    o.y = pt * (Z[j].y * hv[j].y + (1.0f - Z[j].y) * ht.y);
                    o.z = pt * (Z[j].z * hv[j].z + (1.0f - Z[j].z) * ht.z);
                    o.w = pt * (Z[j].w * hv[j].w + (1.0f - Z[j].w) * ht.w);
                    *reinterpret_cast<float4*>(lds + r * KROW + 4 * (threadIdx.x & (TPR - 1))) = o;   // own element
                }
            }
        }
    }
    __syncthreads();
    // segmented sum over each node's rows inside the tile
    const int node1 = (int)((m0 + rm.nvalid - 1) / a.T);
    const int items = (node1 - node0 + 1) * TPR;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int node = node0 + it / TPR, c4 = it % TPR;
        const int cc = n0 + 4 * c4;
        if (cc >= a.C) continue;
        long lo = (long)node * a.T - m0, hi = lo + a.T - 1;
        // all T rows of the node inside this tile: nobody else adds to its row of the zero-initialised hidden state -- one
        // 16-byte store instead of four atomics (which remain for the tile's first / last, partial nodes)
        const bool whole = lo >= 0 && hi <= rm.nvalid - 1;
        if (lo < 0) lo = 0;
        if (hi > rm.nvalid - 1) hi = rm.nvalid - 1;
        float4 s4 = make_float4(0, 0, 0, 0);
        for (long r = lo; r <= hi; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(lds + r * KROW + 4 * c4);
            s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        }
        float* o = a.OH + (long)node * C + cc;
        if (whole) *reinterpret_cast<float4*>(o) = s4;
        else { atomicAdd(o + 0, s4.x); atomicAdd(o + 1, s4.y); atomicAdd(o + 2, s4.z); atomicAdd(o + 3, s4.w); }
    }
}

// The same with ZR, h and Ht stored as bf16 (bf16-operand core, REGT_GEMM_MODE=bf16): 8 columns per thread, so that the
// epilogue's reads of Z and h and its store of H~ are 16-byte accesses.
__global__ __launch_bounds__(256, 2) void gemm_cand_flat8_kernel(CandArgs a) {
    using Core = SplitCore<false, 1>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = Core::BM, BN = Core::BN, KROW = Core::EKROW, TPR = Core::ETPR;
    const long C = a.C, M = (long)a.num_nodes * a.T;
    const int tiles_n = (a.C + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * BM;
    const int n0 = (bid % tiles_n) * BN;
    const RowMap rm{m0, 1, (int)((M - m0) < BM ? (M - m0) : BM)};
    Core core(a.S, rm, n0, a.C, lds);
    core.plan();
    typename Core::Acc acc;
    Core::zero(acc);
    core.run(acc, false);
    const int tid = threadIdx.x;
    const int c = n0 + 8 * (tid & 15);
    const int node0 = (int)(m0 / a.T);
    constexpr int RR = 4;                          // rows per round of the thread's 8 rows (tid >> 4) + 16 i
    F8 Z[RR], hv[RR];
    if (c < a.C) {
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            const int r = (tid >> 4) + 16 * j;
            if (r < rm.nvalid) {
                Z[j] = ld8(a.ZR, (m0 + r) * 2 * C + c, 1);
                hv[j] = ld8(a.h, (m0 + r) * C + c, 1);
            }
        }
    }
    core.stage(acc);
    if (c < a.C) {
        const F8 b = ld8(a.bias, c, 0);
#pragma unroll
        for (int g = 0; g < 8 / RR; ++g) {
            if (g > 0) {
#pragma unroll
                for (int j = 0; j < RR; ++j) {
                    const int r = (tid >> 4) + 16 * (RR * g + j);
                    if (r < rm.nvalid) {
                        Z[j] = ld8(a.ZR, (m0 + r) * 2 * C + c, 1);
                        hv[j] = ld8(a.h, (m0 + r) * C + c, 1);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                const int r = (tid >> 4) + 16 * (RR * g + j);
                if (r < rm.nvalid) {
                    const long m = m0 + r;
                    const float pt = a.probs[(int)(m % a.T)];
                    float4* img = reinterpret_cast<float4*>(lds + r * KROW + 8 * (tid & 15));
                    const F8 v{img[0], img[1]};
#define F_(k) fast_tanh(v.k + b.k)
                    const F8 ht = REGT_F8(F_);
#undef F_
                    st8(a.Ht, m * C + c, ht, 1);
#define F_(k) (pt * (Z[j].k * hv[j].k + (1.0f - Z[j].k) * ht.k))
                    const F8 o = REGT_F8(F_);
#undef F_
                    img[0] = o.lo;                 // own elements of the staged tile
                    img[1] = o.hi;
                }
            }
        }
    }
    __syncthreads();
    // segmented sum over each node's rows inside the tile (as gemm_cand_flat_kernel)
    const int node1 = (int)((m0 + rm.nvalid - 1) / a.T);
    const int items = (node1 - node0 + 1) * TPR;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int node = node0 + it / TPR, c4 = it % TPR;
        const int cc = n0 + 4 * c4;
        if (cc >= a.C) continue;
        long lo = (long)node * a.T - m0, hi = lo + a.T - 1;
        // all T rows of the node inside this tile: nobody else adds to its row of the zero-initialised hidden state -- one
        // 16-byte store instead of four atomics (which remain for the tile's first / last, partial nodes)
        const bool whole = lo >= 0 && hi <= rm.nvalid - 1;
        if (lo < 0) lo = 0;
        if (hi > rm.nvalid - 1) hi = rm.nvalid - 1;
        float4 s4 = make_float4(0, 0, 0, 0);
        for (long r = lo; r <= hi; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(lds + r * KROW + 4 * c4);
            s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        }
        float* o = a.OH + (long)node * C + cc;
        if (whole) *reinterpret_cast<float4*>(o) = s4;
        else { atomicAdd(o + 0, s4.x); atomicAdd(o + 1, s4.y); atomicAdd(o + 2, s4.z); atomicAdd(o + 3, s4.w); }
    }
}

// Candidate stage on the three-workgroup core (fp32 planes NP = 0, bf16x3 NP = 3), fp32 storage.  Per 64-row half of the
// tile: H~ = tanh(acc + ch) is stored, blended (p_t (Z h + (1 - Z) H~)) back into the half's LDS image, and the rows of
// every node that intersects the half are summed and added to OH with one atomic per element -- a node of T <= 64 rows
// meets at most two halves (tile boundaries are half boundaries), so every OH element is 0 + a + b: order-independent.
// The row -> (node, period) map costs one integer division per tile ROW (row table in LDS, written before the K loop),
// not one 64-bit modulo per row slot of every thread; arrays are addressed through buffer descriptors (one per array
// and tile); a full tile runs without a single branch (see the functor notes at the top of this file).
struct CandRowEnt { int nt; float p; };        // nt = (node - first node of the tile) << 8 | period
template <int NP, bool FULL>
__device__ __forceinline__ void cand_epilogue(const CandArgs& a, const SplitCore<false, NP>& core, f32x16 (&acc)[2][2], float* lds,
                                              const CandRowEnt* rowtab, long m0, int n0, int nvalid, int node0, int t0) {
    const int tid = threadIdx.x, rr = tid >> 5, c4 = 4 * (tid & 31);
    const int C = a.C;
    const bool col_ok = FULL || n0 + c4 < C;
    const __amdgpu_buffer_rsrc_t szr = buf_srd(a.ZR + m0 * (2L * C) + n0), sh = buf_srd(a.h + m0 * C + n0), sht = buf_srd(a.Ht + m0 * C + n0);
    const int vc = (rr * C + c4) * 4, vzr = (rr * 2 * C + c4) * 4, sc = 8 * C * 4, szs = 2 * sc;
    const float4 b = col_ok ? ld4(a.bias + n0 + c4) : make_float4(0, 0, 0, 0);
    constexpr int RR = 4, NR = 16 / RR;
    float4 Z[2][RR], hv[2][RR];
    auto request = [&](int k, float4 (&Zd)[RR], float4 (&hd)[RR]) {
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            const int i = RR * k + j;
            if (FULL || (rr + 8 * i < nvalid && col_ok)) {
                Zd[j] = buf_ld4(szr, vzr, i * szs);
                hd[j] = buf_ld4(sh, vc, i * sc);
            }
        }
    };
    request(0, Z[0], hv[0]);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int half = k / (NR / 2);
        if (k % (NR / 2) == 0) core.stage_half(half, acc);
        if (k + 1 < NR) request(k + 1, Z[(k + 1) & 1], hv[(k + 1) & 1]);
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            const int i = RR * k + j, rl = rr + 8 * (i & 7);          // row rr + 8 i of the tile = row rl of its half
            if (FULL || (rr + 8 * i < nvalid && col_ok)) {
                float* img = lds + rl * G_LDS_KROW + c4;
                const float4 v = *reinterpret_cast<const float4*>(img);
                const float pt = rowtab[rr + 8 * i].p;
#define F_(q) fast_tanh(v.q + b.q)
                const float4 ht = REGT_V4(F_);
#undef F_
                buf_st4(sht, vc + i * sc, 0, ht);
                const float4 Zv = Z[k & 1][j], hh = hv[k & 1][j];
#define F_(q) __fmul_rn(pt, gru_blend(Zv.q, hh.q, ht.q))
                *reinterpret_cast<float4*>(img) = REGT_V4(F_);
#undef F_
            }
        }
        if (k % (NR / 2) == NR / 2 - 1) {
            // the half is blended: segmented sums over the nodes that intersect it
            __syncthreads();
            const int r_lo = 64 * half, r_hi = (FULL ? 64 * half + 63 : (nvalid - 1 < 64 * half + 63 ? nvalid - 1 : 64 * half + 63));
            if (FULL || r_hi >= r_lo) {
                const int n_first = rowtab[r_lo].nt >> 8, n_last = rowtab[r_hi].nt >> 8;
                for (int nd = n_first + rr; nd <= n_last; nd += 8) {
                    int lo = nd * a.T - t0, hi = lo + a.T - 1;               // the node's rows in tile coordinates
                    const bool whole = lo >= r_lo && hi <= r_hi;             // all of the node's rows in this half: plain store
                    lo = lo < r_lo ? r_lo : lo;
                    hi = hi > r_hi ? r_hi : hi;
                    float4 s4 = make_float4(0, 0, 0, 0);
                    for (int r = lo; r <= hi; ++r) {
                        const float4 v = *reinterpret_cast<const float4*>(lds + (r - r_lo) * G_LDS_KROW + c4);
                        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
                    }
                    if (col_ok) {
                        float* o = a.OH + (long)(node0 + nd) * C + n0 + c4;
                        if (whole) *reinterpret_cast<float4*>(o) = s4;
                        else { atomicAdd(o + 0, s4.x); atomicAdd(o + 1, s4.y); atomicAdd(o + 2, s4.z); atomicAdd(o + 3, s4.w); }
                    }
                }
            }
        }
    }
}
template <int NP>
__global__ __launch_bounds__(256, 3) void gemm_cand_split_kernel(CandArgs a, int uniform) {
    using Core = SplitCore<false, NP>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const long M = (long)a.num_nodes * a.T;
    const int tiles_n = (a.C + GBN - 1) / GBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    const RowMap rm{m0, 1, (int)((M - m0) < GBM ? (M - m0) : GBM)};
    Core core(a.S, rm, n0, a.C, lds, true);
    CandRowEnt* rowtab = reinterpret_cast<CandRowEnt*>(core.rowtab());
    // first node / period of the tile: ONE 32-bit division per thread (M < 2^31, host-checked); row r then needs a small
    // quotient only ((t0 + r) / T by float reciprocal, exact below 2^16: the +0.5 keeps it off the integer boundaries)
    const int node0 = (int)((unsigned)m0 / (unsigned)a.T), t0 = (int)m0 - node0 * a.T;
    if (threadIdx.x < GBM) {
        const int x = t0 + (threadIdx.x < rm.nvalid ? threadIdx.x : 0), q = (int)(((float)x + 0.5f) * (1.0f / (float)a.T));
        rowtab[threadIdx.x] = CandRowEnt{(q << 8) | (x - q * a.T), a.probs[x - q * a.T]};
    }
    if (!uniform) core.plan();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (uniform) core.run_uniform(acc, false);
    else core.run(acc, false);
    if (rm.nvalid == GBM && n0 + GBN <= a.C) cand_epilogue<NP, true>(a, core, acc, lds, rowtab, m0, n0, GBM, node0, t0);
    else cand_epilogue<NP, false>(a, core, acc, lds, rowtab, m0, n0, rm.nvalid, node0, t0);
}

// The same with Z, h and H~ stored as bf16 (REGT_GEMM_MODE=bf16, bf16-operand core NP = 1): 8 columns per thread (16-byte
// accesses of the bf16 arrays), row slots i = 0 .. 7: row (tid >> 4) + 16 i.
template <bool FULL>
__device__ __forceinline__ void cand8_epilogue(const CandArgs& a, const SplitCore<false, 1>& core, f32x16 (&acc)[2][2], float* lds,
                                               const CandRowEnt* rowtab, long m0, int n0, int nvalid, int node0, int t0) {
    const int tid = threadIdx.x, rr = tid >> 4, c8 = 8 * (tid & 15);
    const int C = a.C;
    const bool col_ok = FULL || n0 + c8 < C;
    const __amdgpu_buffer_rsrc_t szr = buf_srd(reinterpret_cast<const char*>(a.ZR) + 2 * (m0 * (2L * C) + n0)),
                                 sh = buf_srd(reinterpret_cast<const char*>(a.h) + 2 * (m0 * C + n0)),
                                 sht = buf_srd(reinterpret_cast<const char*>(a.Ht) + 2 * (m0 * C + n0));
    const int vc = (rr * C + c8) * 2, vzr = (rr * 2 * C + c8) * 2, sc = 16 * C * 2, szs = 2 * sc;
    F8 b{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (col_ok) b = ld8(a.bias, n0 + c8, 0);
    constexpr int RR = 2, NR = 8 / RR;
    u32x4_t Z[2][RR], hv[2][RR];
    auto request = [&](int k, u32x4_t (&Zd)[RR], u32x4_t (&hd)[RR]) {
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            const int i = RR * k + j;
            if (FULL || (rr + 16 * i < nvalid && col_ok)) {
                Zd[j] = buf_ld16(szr, vzr, i * szs);
                hd[j] = buf_ld16(sh, vc, i * sc);
            }
        }
    };
    request(0, Z[0], hv[0]);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int half = k / (NR / 2);
        if (k % (NR / 2) == 0) core.stage_half(half, acc);
        if (k + 1 < NR) request(k + 1, Z[(k + 1) & 1], hv[(k + 1) & 1]);
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            const int i = RR * k + j, rl = rr + 16 * (i & 3);
            if (FULL || (rr + 16 * i < nvalid && col_ok)) {
                float4* img = reinterpret_cast<float4*>(lds + rl * G_LDS_KROW + c8);
                const F8 v{img[0], img[1]};
                const float pt = rowtab[rr + 16 * i].p;
#define F_(q) fast_tanh(v.q + b.q)
                const F8 ht = REGT_F8(F_);
#undef F_
                buf_st8_bf16(sht, vc + i * sc, ht);
                const F8 Zv = widen8(Z[k & 1][j]), hh = widen8(hv[k & 1][j]);
#define F_(q) __fmul_rn(pt, gru_blend(Zv.q, hh.q, ht.q))
                const F8 o = REGT_F8(F_);
#undef F_
                img[0] = o.lo;
                img[1] = o.hi;
            }
        }
        if (k % (NR / 2) == NR / 2 - 1) {
            __syncthreads();
            // the half is blended: per block of `node_sum_rows` rows (the whole half, or the 16 rows a wave of the row-owning fused
            // kernel holds) the rows of every node that meets the block are summed in row order
            const int brows = a.node_sum_rows == 16 ? 16 : 64;
            for (int r_lo = 64 * half; r_lo < 64 * half + 64; r_lo += brows) {
            const int r_end = r_lo + brows - 1, r_hi = FULL ? r_end : (nvalid - 1 < r_end ? nvalid - 1 : r_end);
            if (FULL || r_hi >= r_lo) {
                const int n_first = rowtab[r_lo].nt >> 8, n_last = rowtab[r_hi].nt >> 8;
                for (int nd = n_first + rr; nd <= n_last; nd += 16) {
                    int lo = nd * a.T - t0, hi = lo + a.T - 1;
                    const bool whole = lo >= r_lo && hi <= r_hi;
                    lo = lo < r_lo ? r_lo : lo;
                    hi = hi > r_hi ? r_hi : hi;
                    float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
                    for (int r = lo; r <= hi; ++r) {
                        const float4* p = reinterpret_cast<const float4*>(lds + (r - 64 * half) * G_LDS_KROW + c8);
                        const float4 u = p[0], w = p[1];
                        s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
                        s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
                    }
                    if (col_ok) {
                        float* o = a.OH + (long)(node0 + nd) * C + n0 + c8;
                        if (whole) {
                            reinterpret_cast<float4*>(o)[0] = s0;
                            reinterpret_cast<float4*>(o)[1] = s1;
                        } else {
                            atomicAdd(o + 0, s0.x); atomicAdd(o + 1, s0.y); atomicAdd(o + 2, s0.z); atomicAdd(o + 3, s0.w);
                            atomicAdd(o + 4, s1.x); atomicAdd(o + 5, s1.y); atomicAdd(o + 6, s1.z); atomicAdd(o + 7, s1.w);
                        }
                    }
                }
            }
            }
        }
    }
}
__global__ __launch_bounds__(256, 3) void gemm_cand_split8_kernel(CandArgs a, int uniform) {
    using Core = SplitCore<false, 1>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const long M = (long)a.num_nodes * a.T;
    const int tiles_n = (a.C + GBN - 1) / GBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(bid / tiles_n) * GBM;
    const int n0 = (bid % tiles_n) * GBN;
    const RowMap rm{m0, 1, (int)((M - m0) < GBM ? (M - m0) : GBM)};
    Core core(a.S, rm, n0, a.C, lds, true);
    CandRowEnt* rowtab = reinterpret_cast<CandRowEnt*>(core.rowtab());
    const int node0 = (int)((unsigned)m0 / (unsigned)a.T), t0 = (int)m0 - node0 * a.T;
    if (threadIdx.x < GBM) {
        const int x = t0 + (threadIdx.x < rm.nvalid ? threadIdx.x : 0), q = (int)(((float)x + 0.5f) * (1.0f / (float)a.T));
        rowtab[threadIdx.x] = CandRowEnt{(q << 8) | (x - q * a.T), a.probs[x - q * a.T]};
    }
    if (!uniform) core.plan();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (uniform == 2) core.run_uniform_frag(acc, false);
    else if (uniform) core.run_uniform(acc, false);
    else core.run(acc, false);
    if (rm.nvalid == GBM && n0 + GBN <= a.C) cand8_epilogue<true>(a, core, acc, lds, rowtab, m0, n0, GBM, node0, t0);
    else cand8_epilogue<false>(a, core, acc, lds, rowtab, m0, n0, rm.nvalid, node0, t0);
}

int launch_gemm_candidate(const CandArgs& a, hipStream_t st) {
    REGT_CHECK_ARG(a.num_nodes > 0 && a.T > 0 && a.C > 0, "candidate gemm: empty problem");
    long tiles = (long)cdiv(a.num_nodes, GBM) * cdiv(a.C, GBN);
    const bool vec = a.C % 4 == 0 && a16(a.ZR) && a16(a.h) && a16(a.Ht) && a16(a.OH) && a16(a.bias) &&
                     fast_class(a.S, a.C, true) == 1;
    REGT_CHECK_ARG(!a.act_bf16 || (vec && gemm_mode() == 2), "candidate gemm: bf16-stored activations need the bf16-operand vector path");
    if (vec) {
        static bool attr_done = false, attr_done_split = false, attr_done_bf16 = false;
        if (int rc = set_lds_once(&gemm_cand_flat_kernel<FastCore<true, false>>, G_FAST_LDS_BYTES, &attr_done)) return rc;
        if (int rc = set_lds_once(&gemm_cand_flat_kernel<SplitCore<false, 3>>, G_FAST_LDS_BYTES, &attr_done_split)) return rc;
        if (int rc = set_lds_once(&gemm_cand_flat_kernel<SplitCore<false, 1>>, G_FAST_LDS_BYTES, &attr_done_bf16)) return rc;
        const long M = (long)a.num_nodes * a.T;
        const long ftiles = (long)cdiv(M, GBM) * cdiv(a.C, GBN);
        REGT_CHECK_ARG(ftiles < (1L << 31) && a.T <= 255, "candidate gemm: too many tiles / T > 255");
        if (int rc = launch_zero_f32(a.OH, (long)a.num_nodes * a.C, st)) return rc;
        // three-workgroup kernels (fp32 storage): 64-row halves need T <= 64 (a node meets at most two halves)
        const bool three = !a.act_bf16 && !fp32_core_wide() && gemm_mode() != 2 && (gemm_mode() == 1 || ftiles >= SMALL_TILE_LIMIT) && M < (1L << 31);
        if (three && gemm_mode() == 0)
            hipLaunchKernelGGL((gemm_cand_split_kernel<0>), dim3((unsigned)ftiles), dim3(256), SplitGeom<0>::LDS_BYTES, st, a, uniform_ok(a.S, M));
        else if (three)
            hipLaunchKernelGGL((gemm_cand_split_kernel<3>), dim3((unsigned)ftiles), dim3(256), SplitGeom<3>::LDS_BYTES, st, a, uniform_ok(a.S, M));
        else if (gemm_mode() == 1)
            hipLaunchKernelGGL((gemm_cand_flat_kernel<SplitCore<false, 3>>), dim3((unsigned)ftiles), dim3(256), G_FAST_LDS_BYTES, st, a);
        else if (gemm_mode() == 2 && a.act_bf16 && a.C % 8 == 0 && M < (1L << 31) && !fp32_core_wide()) {
            for (int q = 0; q < a.S.nseg; ++q)
                REGT_CHECK_ARG(!(a.S.seg[q].flags & SEG_B_FRAG) || uniform_ok(a.S, M) == 2, "candidate gemm: fragment-order weights need K %% 32 == 0");
            hipLaunchKernelGGL(gemm_cand_split8_kernel, dim3((unsigned)ftiles), dim3(256), SplitGeom<1>::LDS_BYTES, st, a, uniform_ok(a.S, M));
        } else if (gemm_mode() == 2 && a.act_bf16) {
            for (int q = 0; q < a.S.nseg; ++q)
                REGT_CHECK_ARG(!(a.S.seg[q].flags & SEG_B_FRAG), "candidate gemm: fragment-order weights need the three-workgroup kernel");
            static bool attr_done8 = false;
            REGT_CHECK_ARG(a.C % 8 == 0, "candidate gemm: bf16 storage needs C %% 8 == 0");
            if (int rc = set_lds_once(&gemm_cand_flat8_kernel, G_FAST_LDS_BYTES, &attr_done8)) return rc;
            hipLaunchKernelGGL(gemm_cand_flat8_kernel, dim3((unsigned)ftiles), dim3(256), G_FAST_LDS_BYTES, st, a);
        } else if (gemm_mode() == 2)
            hipLaunchKernelGGL((gemm_cand_flat_kernel<SplitCore<false, 1>>), dim3((unsigned)ftiles), dim3(256), G_FAST_LDS_BYTES, st, a);
        else if (ftiles < SMALL_TILE_LIMIT)      // small graph: 64 x 64 tiles (a node's T <= 64 rows still span at most two)
            hipLaunchKernelGGL((gemm_cand_flat_kernel<SmallCore<true, false>>), dim3((unsigned)(cdiv(M, SM_B) * cdiv(a.C, SM_B))),
                               dim3(256), SM_LDS_BYTES, st, a);
        else
            hipLaunchKernelGGL((gemm_cand_flat_kernel<FastCore<true, false>>), dim3((unsigned)ftiles), dim3(256), G_FAST_LDS_BYTES, st, a);
    } else {
        for (int q = 0; q < a.S.nseg; ++q)
            REGT_CHECK_ARG(!(a.S.seg[q].flags & SEG_A_BF16), "candidate gemm: a bf16-stored operand needs the bf16-operand vector path");
        static bool attr_done2 = false;
        if (int rc = set_lds_once(&gemm_cand_kernel, G_LDS_BYTES, &attr_done2)) return rc;
        hipLaunchKernelGGL(gemm_cand_kernel, dim3((unsigned)tiles), dim3(256), G_LDS_BYTES, st, a);
    }
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// One output element per group of 8 adjacent lanes: the K (and summed-batch) range is strided over the
// group and combined with a fixed xor-shuffle tree (deterministic).  Sizes here are <= 64 x 256 x 256
// outputs with K <= R*C, so the point is latency (enough waves), not FLOP/s.
template <int SG_SPLIT>   // 8: long K (weight compositions over K = C or R*C); 1: short K, many outputs
__global__ __launch_bounds__(256) void small_gemm_kernel(SmallGemm g) {
    const long per = (long)g.m * g.n;
    const int nb = g.sum_batch ? 1 : g.batch;
    const long total = per * nb;
    const int sub = threadIdx.x % SG_SPLIT;
    const long stride = (long)gridDim.x * blockDim.x / SG_SPLIT;
    long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) / SG_SPLIT;
    long wfirst = idx - (threadIdx.x % 64) / SG_SPLIT;           // wave-uniform loop bound (shuffles inside)
    for (; wfirst < total; wfirst += stride, idx += stride) {
        const bool valid = idx < total;
        float s = 0.f;
        int b = 0, i = 0, j = 0;
        if (valid) {
            b = (int)(idx / per);
            const long e = idx - (long)b * per;
            i = (int)(e / g.n);
            j = (int)(e % g.n);                                   // j fastest: coalesced when scj == 1 / sbj == 1
            const int b0 = g.sum_batch ? 0 : b, b1 = g.sum_batch ? g.batch : b + 1;
            for (int bb = b0; bb < b1; ++bb) {
                const float* A = g.A + (long)bb * g.sab + (long)i * g.sai;
                const float* B = g.B + (long)bb * g.sbb + (long)j * g.sbj;
                for (int k = sub; k < g.k; k += SG_SPLIT) s = fmaf(A[(long)k * g.sak], B[(long)k * g.sbk], s);
            }
        }
        if (SG_SPLIT == 8) {
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
        }
        if (valid && sub == 0) {
            float* c = g.C + (long)b * g.scb + (long)i * g.sci + (long)j * g.scj;
            *c = g.accumulate ? *c + s : s;
        }
    }
}

int launch_small_gemm(const SmallGemm& g, hipStream_t st) {
    REGT_CHECK_ARG(g.m > 0 && g.n > 0 && g.batch > 0, "small_gemm: empty problem");
    const long outputs = (long)g.m * g.n * (g.sum_batch ? 1 : g.batch);
    const long klen = (long)g.k * (g.sum_batch ? g.batch : 1);
    const bool split = klen >= 64;
    long total = outputs * (split ? 8 : 1);
    int blocks = cdiv(total, 256);
    if (blocks > 32768) blocks = 32768;
    if (split) hipLaunchKernelGGL(small_gemm_kernel<8>, dim3(blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(small_gemm_kernel<1>, dim3(blocks), dim3(256), 0, st, g);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

__global__ __launch_bounds__(256) void small_gemm_multi_kernel(SgBatch B) {
    int ti = 0;
    while (ti + 1 < B.ntask && (int)blockIdx.x >= B.block_start[ti + 1]) ++ti;
    // field-wise copy of the selected task (scalar selects; no dynamically indexed kernarg struct in scratch)
    const SgTask& T = B.task[ti];
    if (T.split == 0) {
        // Tiled form for the matrix-sized tasks (C x C, C x F outputs with K = F .. R C): one 32 x 32 output tile per workgroup,
        // 32-k chunks of both operands staged through LDS (each element is read from memory once per tile instead of once per
        // output element), a thread owns 2 x 2 outputs; sums run over (term, batch, k) in ascending order: deterministic.
        // Whichever of a tile's two indices is contiguous in memory is the one consecutive lanes walk.
        __shared__ float As[32][34], Bs[32][34];
        const int tiles_m = (T.m + 31) / 32, tiles_n = (T.n + 31) / 32;
        int w = blockIdx.x - B.block_start[ti];
        const int b = w / (tiles_m * tiles_n);
        w -= b * tiles_m * tiles_n;
        const int i0 = (w / tiles_n) * 32, j0 = (w % tiles_n) * 32;
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
        float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
        for (int t = 0; t < T.nterm; ++t) {
            const SgTerm& q = T.term[t];
            const int b0 = q.sum_batch ? 0 : b, b1 = q.sum_batch ? q.batch : b + 1;
            const bool a_kfast = q.sak == 1, b_kfast = q.sbk == 1;
            for (int bb = b0; bb < b1; ++bb) {
                const float* A = q.A + (long)bb * q.sab;
                const float* Bp = q.B + (long)bb * q.sbb;
                for (int k0 = 0; k0 < q.k; k0 += 32) {
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = threadIdx.x + 256 * e4;
                        const int ia = a_kfast ? e >> 5 : e & 31, ka = a_kfast ? e & 31 : e >> 5;
                        As[ka][ia] = (i0 + ia < T.m && k0 + ka < q.k) ? A[(long)(i0 + ia) * q.sai + (long)(k0 + ka) * q.sak] : 0.f;
                        const int jb = b_kfast ? e >> 5 : e & 31, kb = b_kfast ? e & 31 : e >> 5;
                        Bs[kb][jb] = (j0 + jb < T.n && k0 + kb < q.k) ? Bp[(long)(k0 + kb) * q.sbk + (long)(j0 + jb) * q.sbj] : 0.f;
                    }
                    __syncthreads();
#pragma unroll 8
                    for (int k = 0; k < 32; ++k) {
                        const float a0 = As[k][2 * ty], a1 = As[k][2 * ty + 1], v0 = Bs[k][2 * tx], v1 = Bs[k][2 * tx + 1];
                        c00 = fmaf(a0, v0, c00); c01 = fmaf(a0, v1, c01);
                        c10 = fmaf(a1, v0, c10); c11 = fmaf(a1, v1, c11);
                    }
                    __syncthreads();
                }
            }
        }
        const float cc[2][2] = {{c00, c01}, {c10, c11}};
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int i = i0 + 2 * ty + di, j = j0 + 2 * tx + dj;
                if (i < T.m && j < T.n) {
                    float v = cc[di][dj];
                    if (T.init) v += T.init[(long)i * T.init_si + (long)j * T.init_sj];
                    T.C[(long)b * T.scb + (long)i * T.sci + (long)j * T.scj] = v;
                }
            }
        return;
    }
    const long per = (long)T.m * T.n, total = per * T.nbatch;
    const int sp = T.split;                    // uniform per workgroup: a task starts on a workgroup boundary
    const int sub = sp == 8 ? (threadIdx.x & 7) : 0;
    const long idx = ((long)(blockIdx.x - B.block_start[ti]) * 256 + threadIdx.x) / sp;
    const bool valid = idx < total;
    float s = 0.f;
    int b = 0, i = 0, j = 0;
    if (valid) {
        b = (int)(idx / per);
        const long e = idx - (long)b * per;
        i = (int)(e / T.n);
        j = (int)(e % T.n);
        for (int t = 0; t < T.nterm; ++t) {
            const SgTerm& q = T.term[t];
            const int b0 = q.sum_batch ? 0 : b, b1 = q.sum_batch ? q.batch : b + 1;
            for (int bb = b0; bb < b1; ++bb) {
                const float* A = q.A + (long)bb * q.sab + (long)i * q.sai;
                const float* Bp = q.B + (long)bb * q.sbb + (long)j * q.sbj;
                for (int k = sub; k < q.k; k += sp) s = fmaf(A[(long)k * q.sak], Bp[(long)k * q.sbk], s);
            }
        }
    }
    if (sp == 8) {
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
    }
    if (valid && sub == 0) {
        if (T.init) s += T.init[(long)i * T.init_si + (long)j * T.init_sj];
        T.C[(long)b * T.scb + (long)i * T.sci + (long)j * T.scj] = s;
    }
}

int launch_small_gemm_multi(SgBatch& b, hipStream_t st) {
    REGT_CHECK_ARG(b.ntask > 0 && b.ntask <= SG_MAX_TASKS && !b.overflow, "small_gemm_multi: %d tasks%s", b.ntask,
                   b.overflow ? " (more than SG_MAX_TASKS were added)" : "");
    int blocks = 0;
    for (int t = 0; t < b.ntask; ++t) {
        b.block_start[t] = blocks;
        SgTask& task = b.task[t];
        const long outputs = (long)task.m * task.n * task.nbatch;
        REGT_CHECK_ARG(outputs > 0, "small_gemm_multi: empty task %d", t);
        long ksum = 0;                           // multiply-adds per output element
        for (int q = 0; q < task.nterm; ++q) ksum += (long)task.term[q].k * (task.term[q].sum_batch ? task.term[q].batch : 1);
        // eight lanes per output (strided k + xor tree) only pay off for long sums; a K = F product is one lane's work;
        // matrix-sized outputs take the tiled form (split = 0)
        // (sums longer than 256 over few tiles -- d cheb_w1 = sum over the owned regions, K = R C -- stay on the 8-lane form: 16
        // workgroups walking 64 chunks each were slower, 0.25 vs 0.15 ms for the launch; REGT_SG_TILED_MAXK: developer switch)
        constexpr long tiled_maxk = 256;
        if (task.m >= 16 && task.n >= 16 && ksum >= 16 && ksum <= tiled_maxk) {
            task.split = 0;
            blocks += (int)((long)cdiv(task.m, 32) * cdiv(task.n, 32) * task.nbatch);
        } else {
            task.split = ksum > 32 ? 8 : 1;
            blocks += cdiv(outputs * task.split, 256);
        }
    }
    b.block_start[b.ntask] = blocks;
    hipLaunchKernelGGL(small_gemm_multi_kernel, dim3(blocks), dim3(256), 0, st, b);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt

#ifdef REGT_WG_TRACE
// developer build only (not part of include/regtgcn.h): copy the workgroup trace to the host
extern "C" int regt_wg_trace_read(long* host, long nblocks) {
    if (nblocks > regt::WG_TRACE_MAX) nblocks = regt::WG_TRACE_MAX;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(regt::g_wg_trace), sizeof(long) * 4 * nblocks);
}
extern "C" int regt_wg_marks_read(long* host, long nblocks) {
    if (nblocks > regt::WG_TRACE_MAX) nblocks = regt::WG_TRACE_MAX;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(regt::g_wg_marks), sizeof(long) * 8 * nblocks);
}
#endif
