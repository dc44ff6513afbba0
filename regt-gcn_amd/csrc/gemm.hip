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
//   wgrad3_kernel / wgrad_kernel<BNW> / wgrad_split_kernel   out = P^T Q over row chunks (weight gradients), partial slabs
//   wgrad_reduce(_multi)_kernel        deterministic reduction of the slabs (all of a backward pass in one launch)
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

// hipFuncSetAttribute is a (slow, host-synchronous) driver call: do it once per kernel, not per launch.
template <class K>
static int set_lds_once(K kernel, int bytes, bool* done) {
    if (*done) return REGT_OK;
    REGT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    *done = true;
    return REGT_OK;
}

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
// 4 bf16 (8 bytes) widened to fp32 bit patterns
__device__ __forceinline__ float4 widen_bf16x4(unsigned lo, unsigned hi) {
    return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                       __uint_as_float(hi & 0xffff0000u));
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

static inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
// Depth of the register ring of wgrad_bf16_ring_kernel: REGT_WGRAD_RING / regt_set_option("wgrad_ring", d) = 0 (the one-half-slab-
// ahead kernel wgrad_split_kernel<1, true, true>: same slabs bit for bit) | 4 | 6 | 8 half slabs of lead.
static int g_wgrad_ring = -1;
static int wgrad_ring_depth() {
    if (g_wgrad_ring < 0) g_wgrad_ring = 6;
    return g_wgrad_ring;
}
static int g_wgrad_ring256 = -1;
int wgrad_ring256_option(int value) {      // ring depth of the 256-row tile variant: 2 (default) | 4; -1 = query
    if (g_wgrad_ring256 < 0) g_wgrad_ring256 = 2;
    const int prev = g_wgrad_ring256;
    if (value >= 0) g_wgrad_ring256 = value == 4 ? 4 : 2;
    return prev;
}
static int g_wgrad_bnw64 = -1;
int wgrad_bnw64_option(int value) {      // regt_set_option("wgrad_bnw64", 0 | 1); -1 = query
    if (g_wgrad_bnw64 < 0) g_wgrad_bnw64 = 1;
    const int prev = g_wgrad_bnw64;
    if (value >= 0) g_wgrad_bnw64 = value ? 1 : 0;
    return prev;
}
static int g_wgrad_tile = -1;
static int wgrad_tile_rows() {
    if (g_wgrad_tile < 0) g_wgrad_tile = 256;
    return g_wgrad_tile;
}
int wgrad_tile_option(int value) {
    const int prev = wgrad_tile_rows();
    g_wgrad_tile = value == 256 ? 256 : 128;
    return prev;
}
bool wgrad_ring_active() { return wgrad_ring_depth() > 0; }

// Row chunking for a ring-kernel launch whose workgroups are ALL resident at once and fill every slot: chunks x tiles = CUs x
// workgroups per CU.  The tiles of a chunk share their operands through L2 only while they walk the chunk in step; started
// together they do, started as slots free up (1.5 waves of workgroups at 128 chunks x 6 tiles) they do not, and the half-filled
// last wave costs as much as a full one.  Fewer, longer chunks also mean fewer slabs to write and reduce.
// REGT_WGRAD_WAVE=0 / regt_set_option("wgrad_wave", 0): the layout's ~128 chunks.  false: not applicable, keep the caller's chunking.
static int g_wgrad_wave = -1;
int wgrad_wave_option(int value) {
    if (g_wgrad_wave < 0) g_wgrad_wave = 1;
    const int prev = g_wgrad_wave;
    if (value >= 0) g_wgrad_wave = value ? 1 : 0;
    return prev;
}
// The same for the wide fp32 / bf16x3 kernels (wgrad3_kernel: three workgroups per CU; wgrad_split_kernel<3>: two): chunks x
// (128 x 128 tiles) = one full wave of workgroups instead of ~128 chunks (768 instead of 1024 / 512 workgroups for dUzr / dUh at
// C = 256).  These kernels are MFMA-bound, so it buys little: -0.05 ms of 4.0 at cfg-3, -0.01 ms at the W = 8 shard shape
// (profiles/r04_wgrad_wave32_ab.txt).  REGT_WGRAD_WAVE32=0: the layout's chunks; =2: two waves (more slabs to reduce: slower).
static int g_wgrad_wave32 = -1;
bool wgrad_wide_chunking(int Nout, int Nin, long M, int* kchunk, int* nchunks) {
    if (g_wgrad_wave32 < 0) g_wgrad_wave32 = 1;
    if (!g_wgrad_wave32 || gemm_mode() == 2 || fp32_core_wide() || Nin <= 32) return false;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const int per_cu = gemm_mode() == 1 ? 2 : 3;
    const long tpc = (long)cdiv(Nout, 128) * cdiv(Nin, 128);
    long nch = (long)cus * per_cu * g_wgrad_wave32 / tpc;       // g_wgrad_wave32 waves of workgroups
    if (nch < 1) return false;
    long kc = ((M + nch - 1) / nch + 31) / 32 * 32;
    if (kc < 512) return false;
    if (kc > 32768) {
        const long waves = (kc + 32767) / 32768;
        kc = ((M + nch * waves - 1) / (nch * waves) + 31) / 32 * 32;
    }
    *kchunk = (int)kc;
    *nchunks = (int)((M + kc - 1) / kc);
    return true;
}
// Skinny gradients (Nin <= 32: wgrad_kernel<32>, HBM-bound on their left operand): chunks x row tiles = REGT_WGRAD_SKINNY (default 2)
// workgroups per CU, all resident at once -- at cfg-3 the layout's 507 chunks are 1.3 (dGh) / 2.6 (dGzr) waves of workgroups.
bool wgrad_skinny_chunking(int Nout, long M, int* kchunk, int* nchunks) {
    constexpr int per_cu = 2;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const long nch = (long)cus * per_cu / cdiv(Nout, 128);
    if (nch < 1) return false;
    long kc = ((M + nch - 1) / nch + 31) / 32 * 32;
    if (kc < 512) return false;
    if (kc > 32768) {
        const long waves = (kc + 32767) / 32768;
        kc = ((M + nch * waves - 1) / (nch * waves) + 31) / 32 * 32;
    }
    *kchunk = (int)kc;
    *nchunks = (int)((M + kc - 1) / kc);
    return true;
}
bool wgrad_ring_chunking(int Nout, int Nin, long M, int* kchunk, int* nchunks) {
    if (!wgrad_ring_active() || !wgrad_wave_option(-1)) return false;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const bool wide = wgrad_tile_rows() == 256 && Nout % 256 == 0;
    const int per_cu = wide || wgrad_ring_depth() >= 8 ? 2 : 3;           // register-limited workgroups per CU of the variant launched
    const long tpc = (long)cdiv(Nout, wide ? 256 : 128) * cdiv(Nin, 128);
    const long nch = (long)cus * per_cu / tpc;
    if (nch < 1) return false;
    long kc = ((M + nch - 1) / nch + 31) / 32 * 32;
    if (kc < 512) return false;                                            // small problems: many short chunks (latency-bound regime)
    if (kc > 32768) {                                                      // the kernels' 32-bit row offsets: whole waves of shorter chunks
        const long waves = (kc + 32767) / 32768;
        kc = ((M + nch * waves - 1) / (nch * waves) + 31) / 32 * 32;
    }
    *kchunk = (int)kc;
    *nchunks = (int)((M + kc - 1) / kc);
    return true;
}
// Upper bound of the row chunks ANY of the three per-launch chunkers above can return for a (Nout x Nin) gradient over M rows, whatever
// the arithmetic / switches at launch time: make_layout sizes the slab regions with it (the layout's own ~128 / ~512 chunks were too few
// once the launches started to pick their counts: C = 128 in fp32 asks for 768 chunks of dUh).
long wgrad_chunk_bound(int Nout, int Nin, long M) {
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int skinny = 2, wave32 = 1;
    const long cand[3] = {(long)cus * 3 * wave32 / ((long)cdiv(Nout, 128) * cdiv(Nin, 128)),      // wide fp32 / bf16x3, ring (128-row tiles)
                          (long)cus * skinny / cdiv(Nout, 128),                                   // skinny
                          (long)cus * 3 / ((long)cdiv(Nout, 256) > 0 ? (long)cdiv(Nout, 256) * cdiv(Nin, 128) : 1)};
    long best = 0;
    for (long nch : cand) {
        if (nch < 1) continue;
        long kc = ((M + nch - 1) / nch + 31) / 32 * 32;
        if (kc > 32768) {
            const long waves = (kc + 32767) / 32768;
            kc = ((M + nch * waves - 1) / (nch * waves) + 31) / 32 * 32;
        }
        if (kc < 32) kc = 32;
        const long n = (M + kc - 1) / kc;
        best = n > best ? n : best;
    }
    return best;
}
int wgrad_ring_option(int value) {
    const int prev = wgrad_ring_depth();
    g_wgrad_ring = value < 0 ? 0 : value;
    return prev;
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

// ---- weight gradients: out[Nout x Nin] = P^T Q ----------------------------------------------------
// Tile 128 (Nout) x BNW (Nin), K = rows of P/Q.  Both operands are staged k-major ([k][i]) exactly
// as they lie in HBM (row m contiguous along i), read back with conflict-free ds_read_b32.
constexpr int W_BK = 32;
constexpr int W_LDP = 128 + 4;

// 64 (round 4): waves 4x1, each 1 x 2 MFMA tiles -- a 64-wide right-hand side ([x | L~ x] at F = 32) as ONE column tile: P crosses
// HBM / L2 once instead of once per 32 columns and a fragment of P feeds two MFMAs; the two-part right-hand side may split INSIDE it
template <int BNW, bool PBF = false>   // 128: waves 2x2, each 2x2 MFMA tiles;  32: waves 4x1, each one MFMA tile
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
    static_assert(BNW == 128 || BNW == 64 || BNW == 32, "column tile");
    constexpr int WM = BNW == 128 ? 2 : 1, WN = BNW == 32 ? 1 : 2;
    constexpr int LDQ = BNW + 4;
    constexpr int P_TILE = W_BK * W_LDP, Q_TILE = W_BK * LDQ;
    constexpr int QSLOTS = (W_BK * BNW / 4) / 256;          // float4 slots per thread for Q (4 or 1)
    using Core = FastCore<true, false>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = BNW == 128 ? (wid >> 1) : wid, wc = BNW == 128 ? (wid & 1) : 0;
    const int tiles_i = (a.Nout + 127) / 128, tiles_j = (a.Nin + BNW - 1) / BNW;
    // XCD-aware mapping: all tiles of one row chunk run on the same XCD (workgroup b lands on XCD b % 8), back to
    // back, so the chunk's P and Q rows are fetched from HBM once and served to the other tiles from that XCD's L2.
    const int tpc = tiles_i * tiles_j;
    int tile, chunk;
    {
        const int nfull = (a.nchunks / 8) * 8;                 // chunks that can be dealt 8 at a time
        const int b = blockIdx.x;
        if (b < nfull * tpc) {
            const int xcd = b & 7, li = b >> 3;
            chunk = (li / tpc) * 8 + xcd;
            tile = li % tpc;
        } else {                                               // remainder chunks: plain order
            const int r = b - nfull * tpc;
            chunk = nfull + r / tpc;
            tile = r % tpc;
        }
    }
    const int i0 = (tile / tiles_j) * 128, j0 = (tile % tiles_j) * BNW;
    long r0, r1;
    if (a.chunk_tab) { r0 = a.chunk_tab[2 * chunk]; r1 = a.chunk_tab[2 * chunk + 1]; }
    else { r0 = (long)chunk * a.kchunk; r1 = r0 + a.kchunk < a.M ? r0 + a.kchunk : a.M; }
    const int nrows = (int)(r1 - r0);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x2 csum2 = {0.f, 0.f};

    // Both operands stream through wave-uniform buffer descriptors based at the chunk's first row
    // (vector path: ldp/ldq multiples of 4 and 16-B aligned bases, checked on the host).
    // PBF: P holds bf16 elements (REGT_GEMM_MODE=bf16: dhp, dzp|drp are rounded once by their producer): a 32 x 128 slab is
    // 8 KB = two 16-byte loads per thread (8 columns each), widened to fp32 on the way into LDS -- this kernel's arithmetic
    // stays the fp32 MFMA
    const bool second = a.Q2 != nullptr && j0 >= a.nin_split;          // this column tile reads the second operand
    // BNW = 64: the split may run through the tile -- columns past it come from Q2 through a second descriptor (both requested by
    // every lane with complementary out-of-range offsets, OR-ed: an out-of-range lane returns 0 without touching memory)
    const bool straddle = BNW == 64 && a.Q2 != nullptr && !second && j0 + BNW > a.nin_split;
    const int ldp = (int)a.ldp, ldq = second ? (int)a.ldq2 : (int)a.ldq;
    // Descriptors based at the chunk's first row with num_records = the chunk's bytes: a row past the chunk's end is out of
    // range (returns 0) without a per-load guard.  Per thread the offsets inside a 32-row slab are constants (a column past
    // Nout / Nin gets an out-of-range constant), the slab's first row goes into the instruction's scalar offset: no vector
    // instruction per load in the K loop (VALU work shares the SIMD's issue with the MFMAs, gemm_split.h).
    const char* pbase = reinterpret_cast<const char*>(a.P) + (PBF ? 2 : 4) * (r0 * a.ldp + i0);
    const float* qbase = second ? a.Q2 + r0 * a.ldq2 + (j0 - a.nin_split) : a.Q + r0 * a.ldq + j0;
    const long pbytes = (long)nrows * ldp * (PBF ? 2 : 4), qbytes = (long)nrows * ldq * 4;
    const __amdgpu_buffer_rsrc_t sp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pbase), 0, (int)(pbytes < 0x7FFFFFF0L ? pbytes : 0x7FFFFFF0L), 0x00020000);
    const __amdgpu_buffer_rsrc_t sq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qbase), 0, (int)(qbytes < 0x7FFFFFF0L ? qbytes : 0x7FFFFFF0L), 0x00020000);
    const long q2bytes = straddle ? (long)nrows * a.ldq2 * 4 : 0;
    const __amdgpu_buffer_rsrc_t sq2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(straddle ? a.Q2 + r0 * a.ldq2 : qbase), 0,
                                                                         (int)(q2bytes < 0x7FFFFFF0L ? q2bytes : 0x7FFFFFF0L), 0x00020000);
    int vp[4], vq[QSLOTS], vq2[QSLOTS];
    if (PBF) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int idx = tid + 256 * s, i = 8 * (idx & 15);
            vp[s] = i0 + i < a.Nout ? 2 * ((idx >> 4) * ldp + i) : (int)Core::SRD_OOB;
        }
        vp[2] = vp[3] = 0;
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int slot = tid + 256 * s, i = 4 * (slot & 31);
            vp[s] = i0 + i < a.Nout ? 4 * ((slot >> 5) * ldp + i) : (int)Core::SRD_OOB;
        }
    }
#pragma unroll
    for (int s = 0; s < QSLOTS; ++s) {
        const int slot = tid + 256 * s, j = 4 * (slot % (BNW / 4));
        const bool in2 = straddle && j0 + j >= a.nin_split;
        vq[s] = j0 + j < a.Nin && !in2 ? 4 * ((slot / (BNW / 4)) * ldq + j) : (int)Core::SRD_OOB;
        vq2[s] = in2 && j0 + j < a.Nin ? 4 * ((slot / (BNW / 4)) * (int)a.ldq2 + (j0 + j - a.nin_split)) : (int)Core::SRD_OOB;
    }
    const int sp_step = ldp * (PBF ? 2 : 4), sq_step = ldq * 4, sq2_step = (int)a.ldq2 * 4;      // bytes per row

    auto load = [&](int k0, float4 (&rp)[4], float4 (&rq)[QSLOTS]) {
        const int sop = k0 * sp_step, soq = k0 * sq_step;
        if (PBF) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float4 raw = buf_ld4(sp, vp[s], sop);
                rp[2 * s] = widen_bf16x4(__float_as_uint(raw.x), __float_as_uint(raw.y));
                rp[2 * s + 1] = widen_bf16x4(__float_as_uint(raw.z), __float_as_uint(raw.w));
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) rp[s] = buf_ld4(sp, vp[s], sop);
        }
#pragma unroll
        for (int s = 0; s < QSLOTS; ++s) rq[s] = buf_ld4(sq, vq[s], soq);
        if (BNW == 64) {        // unconditional (a tile that does not straddle the split has out-of-range offsets here): no branch around loads
#pragma unroll
            for (int s = 0; s < QSLOTS; ++s) {
                const float4 v = buf_ld4(sq2, vq2[s], k0 * sq2_step);
                rq[s].x = __uint_as_float(__float_as_uint(rq[s].x) | __float_as_uint(v.x));
                rq[s].y = __uint_as_float(__float_as_uint(rq[s].y) | __float_as_uint(v.y));
                rq[s].z = __uint_as_float(__float_as_uint(rq[s].z) | __float_as_uint(v.z));
                rq[s].w = __uint_as_float(__float_as_uint(rq[s].w) | __float_as_uint(v.w));
            }
        }
    };
    auto store = [&](int stage, const float4 (&rp)[4], float4 (&rq)[QSLOTS]) {
        float* lp = lds + stage * (P_TILE + Q_TILE);
        float* lq = lp + P_TILE;
        if (PBF) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int idx = tid + 256 * s;
                float* d = lp + (idx >> 4) * W_LDP + 8 * (idx & 15);
                *reinterpret_cast<float4*>(d) = rp[2 * s];
                *reinterpret_cast<float4*>(d + 4) = rp[2 * s + 1];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int slot = tid + 256 * s;
                *reinterpret_cast<float4*>(lp + (slot >> 5) * W_LDP + 4 * (slot & 31)) = rp[s];
            }
        }
#pragma unroll
        for (int s = 0; s < QSLOTS; ++s) {
            int slot = tid + 256 * s;
            float4 v = rq[s];
            if (a.q_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(lq + (slot / (BNW / 4)) * LDQ + 4 * (slot % (BNW / 4))) = v;
        }
    };
    struct Frag { float a[WM][4], b[WN][4]; };
    auto read_frag = [&](int stage, int kg) {
        const float* lp = lds + stage * (P_TILE + Q_TILE);
        const float* lq = lp + P_TILE;
        Frag f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kg * 8 + lh * 4 + j;
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) f.a[mi][j] = lp[k * W_LDP + wr * (32 * WM) + mi * 32 + lr];
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) f.b[ni][j] = lq[k * LDQ + wc * (32 * WN) + ni * 32 + lr];
        }
        return f;
    };
    auto mfma = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mi][j], f.b[ni][j], acc[mi][ni], 0, 0, 0);
    };
    auto colsum = [&](int stage) {
        if (a.colsum && (j0 == 0 || a.all_csum) && tid < 128) {      // (all_csum: same work in every column tile, launch_wgrad)
            const float* lp = lds + stage * (P_TILE + Q_TILE);
#pragma unroll
            for (int k = 0; k < W_BK; k += 2) {      // two partial sums (even / odd rows): one packed add per two rows
                const f32x2 v = {lp[k * W_LDP + tid], lp[(k + 1) * W_LDP + tid]};
                csum2 += v;
            }
        }
    };

    const int nit = (nrows + W_BK - 1) / W_BK;
    if (nit > 0) {
        float4 rp[4], rq[QSLOTS];
        load(0, rp, rq);
        store(0, rp, rq);
        __syncthreads();
        Frag cur = read_frag(0, 0);
        for (int it = 0; it + 1 < nit; ++it) {
            const int stage = it & 1;
            load((it + 1) * W_BK, rp, rq);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                Frag nxt;
                if (kg < 3) nxt = read_frag(stage, kg + 1);
#pragma unroll
                for (int r = 0; r < 4 * WM * WN; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // VALU | SALU
                }
                mfma(cur);
                if (kg < 3) cur = nxt;
            }
            colsum(stage);
            store(stage ^ 1, rp, rq);
            __syncthreads();
            cur = read_frag(stage ^ 1, 0);
        }
        {
            const int stage = (nit - 1) & 1;
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                Frag nxt;
                if (kg < 3) nxt = read_frag(stage, kg + 1);
                mfma(cur);
                if (kg < 3) cur = nxt;
            }
            colsum(stage);
        }
    }
    const long stride = (long)a.Nout * a.Nin + (a.colsum ? a.Nout : 0);
    float* out = a.slab + (long)chunk * stride;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int i = i0 + wr * (32 * WM) + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (i < a.Nout) {
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    int j = j0 + wc * (32 * WN) + ni * 32 + lr;
                    if (j < a.Nin) out[(long)i * a.Nin + j] = acc[mi][ni][reg];
                }
            }
        }
    if (a.colsum && j0 == 0 && tid < 128 && i0 + tid < a.Nout) out[(long)a.Nout * a.Nin + i0 + tid] = csum2[0] + csum2[1];
}

// fp32 weight gradients on THREE workgroups per CU: 16-row half slabs (two LDS stages of 16 x 132 floats per operand,
// 33,792 B), the half-step schedule of SplitCore::run_t (stage 0 / 1 = the halves of the current 32-row slab, the next slab
// in registers, its halves stored while the other half is multiplied), <= 168 VGPRs.  The two-workgroup kernel above
// reaches ~0.73 of the fp32 MFMA peak: one wave per SIMD and workgroup, every barrier and LDS round trip of a workgroup
// idles its share of the matrix pipe unless another workgroup fills in.  Same tiling, chunking, arithmetic order inside a
// chunk (k ascending, the same pairing of k to MFMA lanes) and output as wgrad_kernel<128, false>.
__global__ __launch_bounds__(256, 3) void wgrad3_kernel(WgradArgs a) {
    constexpr int HK = 16, LDT = 128 + 4, OP_T = HK * LDT, STAGE = 2 * OP_T;      // floats
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_i = (a.Nout + 127) / 128, tiles_j = (a.Nin + 127) / 128;
    const int tpc = tiles_i * tiles_j;
    int tile, chunk;
    {
        const int nfull = (a.nchunks / 8) * 8;
        const int b = blockIdx.x;
        if (b < nfull * tpc) {
            const int xcd = b & 7, li = b >> 3;
            chunk = (li / tpc) * 8 + xcd;
            tile = li % tpc;
        } else {
            const int r = b - nfull * tpc;
            chunk = nfull + r / tpc;
            tile = r % tpc;
        }
    }
    const int i0 = (tile / tiles_j) * 128, j0 = (tile % tiles_j) * 128;
    long r0, r1;
    if (a.chunk_tab) { r0 = a.chunk_tab[2 * chunk]; r1 = a.chunk_tab[2 * chunk + 1]; }
    else { r0 = (long)chunk * a.kchunk; r1 = r0 + a.kchunk < a.M ? r0 + a.kchunk : a.M; }
    const int nrows = (int)(r1 - r0);
    const int ldp = (int)a.ldp, ldq = (int)a.ldq;
    const long pbytes = (long)nrows * ldp * 4, qbytes = (long)nrows * ldq * 4;
    const __amdgpu_buffer_rsrc_t sp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P + r0 * a.ldp + i0), 0,
                                                                       (int)(pbytes < 0x7FFFFFF0L ? pbytes : 0x7FFFFFF0L), 0x00020000);
    const __amdgpu_buffer_rsrc_t sq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q + r0 * a.ldq + j0), 0,
                                                                       (int)(qbytes < 0x7FFFFFF0L ? qbytes : 0x7FFFFFF0L), 0x00020000);
    // a half slab = 16 rows x 128 columns per operand = 512 float4: slots tid, tid + 256 -> row slot >> 5, column 4 (slot & 31)
    int vp[2], vq[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int slot = tid + 256 * s, c4 = 4 * (slot & 31);
        vp[s] = i0 + c4 < a.Nout ? 4 * ((slot >> 5) * ldp + c4) : (int)0x7FFFFFF8;
        vq[s] = j0 + c4 < a.Nin ? 4 * ((slot >> 5) * ldq + c4) : (int)0x7FFFFFF8;
    }
    const int sp_step = HK * ldp * 4, sq_step = HK * ldq * 4;        // bytes per half slab

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x2 csum2 = {0.f, 0.f};
    const bool do_colsum = a.colsum && (j0 == 0 || a.all_csum) && tid < 128;      // stored by column tile 0 only
    const bool store_colsum = a.colsum && j0 == 0 && tid < 128;

    // registers of half h of a slab: rp[2 h + s], rq[2 h + s]
    auto load_half = [&](int g, int h, float4 (&rp)[4], float4 (&rq)[4]) {       // g = half-slab index (rows 16 g ..); past the end: zeros
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            rp[2 * h + s] = buf_ld4(sp, vp[s], g * sp_step);
            rq[2 * h + s] = buf_ld4(sq, vq[s], g * sq_step);
        }
    };
    auto store_half = [&](int h, const float4 (&rp)[4], const float4 (&rq)[4]) {
        float* lp = lds + h * STAGE;
        float* lq = lp + OP_T;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int slot = tid + 256 * s;
            float4 v = rq[2 * h + s];
            if (a.q_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(lp + (slot >> 5) * LDT + 4 * (slot & 31)) = rp[2 * h + s];
            *reinterpret_cast<float4*>(lq + (slot >> 5) * LDT + 4 * (slot & 31)) = v;
        }
    };
    struct Frag { float a[2][8], b[2][8]; };       // [32-row block][k slot]: lane half lh holds k = 8 kk + 4 lh + j at slot 4 kk + j
    auto read_frag = [&](int h) {
        const float* lp = lds + h * STAGE;
        const float* lq = lp + OP_T;
        Frag f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kk * 8 + lh * 4 + j;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f.a[t][4 * kk + j] = lp[k * LDT + wr * 64 + t * 32 + lr];
                    f.b[t][4 * kk + j] = lq[k * LDT + wc * 64 + t * 32 + lr];
                }
            }
        return f;
    };
    auto mfma = [&](const Frag& f) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mi][q], f.b[ni][q], acc[mi][ni], 0, 0, 0);
    };
    auto colsum = [&](int h) {
        if (do_colsum) {
            const float* lp = lds + h * STAGE;
#pragma unroll
            for (int k = 0; k < HK; k += 2) {
                const f32x2 v = {lp[k * LDT + tid], lp[(k + 1) * LDT + tid]};
                csum2 += v;
            }
        }
    };
    // multiply half hc (in LDS) while half hs of the next slab is stored and the same half of the slab after next requested
    auto fused = [&](int hs, int hc, int gnext, float4 (&rp)[4], float4 (&rq)[4]) {
        __builtin_amdgcn_sched_barrier(0);
        const Frag f = read_frag(hc);
        store_half(hs, rp, rq);
        mfma(f);
        load_half(gnext, hs, rp, rq);
        __builtin_amdgcn_sched_group_barrier(0x100, 32, 0);          // the fragment reads (ds_read_b32 / ds_read2)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
            __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);       // VALU | SALU
            if (r < 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nslab = (nrows + 2 * HK - 1) / (2 * HK);
    if (nslab > 0) {
        float4 rp[4], rq[4];
        load_half(0, 0, rp, rq);
        load_half(1, 1, rp, rq);
        store_half(0, rp, rq);
        store_half(1, rp, rq);
        load_half(2, 0, rp, rq);          // slab 1 (zeros past the chunk's end)
        load_half(3, 1, rp, rq);
        __syncthreads();
        mfma(read_frag(0));
        colsum(0);
        for (int t = 0; t + 1 < nslab; ++t) {
            __syncthreads();
            fused(0, 1, 2 * t + 4, rp, rq);       // store half 2t+2 -> stage 0, multiply half 2t+1, request half 2t+4
            colsum(1);
            __syncthreads();
            fused(1, 0, 2 * t + 5, rp, rq);
            colsum(0);
        }
        __syncthreads();
        mfma(read_frag(1));
        colsum(1);
    }
    const long stride = (long)a.Nout * a.Nin + (a.colsum ? a.Nout : 0);
    float* out = a.slab + (long)chunk * stride;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = i0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (i < a.Nout) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int j = j0 + wc * 64 + ni * 32 + lr;
                    if (j < a.Nin) out[(long)i * a.Nin + j] = acc[mi][ni][reg];
                }
            }
        }
    if (store_colsum && i0 + tid < a.Nout) out[(long)a.Nout * a.Nin + i0 + tid] = csum2[0] + csum2[1];
}


// ---- weight gradients on the bf16 matrix pipe (opt-in bf16x3 split, gemm_split.h) ---------------------------
// Same tile (128 x 128), chunking, slab output and XCD mapping as wgrad_kernel<128>; the K loop follows SplitCore:
// P and Q rows are split exactly into three bf16 planes while they are staged, two 16-row half slabs double-buffer
// each other, six partial products per tile pair.  The operands lie k-major in HBM (row m contiguous along i) and are
// staged exactly so: a [16 m][128 i] bf16 image per plane with 256-byte rows whose 16-byte chunks are XOR-swizzled
// (image (b) of cdna_hip_programming.md T10) -- conflict-free for the ds_write_b64 of the staging pass and for
// ds_read_b64_tr_b16, the transposing LDS read that hands every lane 4 consecutive k of one column: two of them per
// plane make the 8-k operand of v_mfma_f32_32x32x16_bf16 without any shuffle.
constexpr int WS_PLANE_B = 16 * 256;             // one plane of one operand of one half slab
__device__ __forceinline__ int ws_off(int m, int ch) { return 256 * m + 16 * (ch ^ (((m & 3) << 2) | ((m >> 2) & 3))); }

// PBF / QBF (NP = 1 only): the operand is STORED as bf16 (dhp, dzp|drp, q: rounded once by their producers).  A half slab
// of such an operand is 16 rows x 128 columns x 2 B = 4 KB = one 16-byte load per thread (row tid >> 4, columns 8 (tid & 15)
// ..+7) that goes to LDS as it is: the chunk layout of ws_off already is 8 bf16 per 16 bytes.
template <int NP, bool PBF = false, bool QBF = false>   // NP 3: exact 3-way split; 1: plain bf16 operands (REGT_GEMM_MODE=bf16)
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WgradArgs a) {
    static_assert(NP == 1 || (!PBF && !QBF), "bf16-stored operands only with the plain bf16 arithmetic");
    constexpr int WS_OPER_B = NP * WS_PLANE_B;       // 12288 / 4096
    constexpr int WS_STAGE_B = 2 * WS_OPER_B;        // P planes, then Q planes
    constexpr int WS_RED_B = 256 * 16;               // column-sum reduction image
    static_assert(2 * WS_STAGE_B >= WS_RED_B, "column-sum image fits the stages");
    using Core = FastCore<true, false>;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_i = (a.Nout + 127) / 128, tiles_j = (a.Nin + 127) / 128;
    const int tpc = tiles_i * tiles_j;
    int tile, chunk;
    {   // all tiles of one row chunk on the same XCD (see wgrad_kernel)
        const int nfull = (a.nchunks / 8) * 8;
        const int b = blockIdx.x;
        if (b < nfull * tpc) {
            const int xcd = b & 7, li = b >> 3;
            chunk = (li / tpc) * 8 + xcd;
            tile = li % tpc;
        } else {
            const int r = b - nfull * tpc;
            chunk = nfull + r / tpc;
            tile = r % tpc;
        }
    }
    const int i0 = (tile / tiles_j) * 128, j0 = (tile % tiles_j) * 128;
    long r0, r1;
    if (a.chunk_tab) { r0 = a.chunk_tab[2 * chunk]; r1 = a.chunk_tab[2 * chunk + 1]; }
    else { r0 = (long)chunk * a.kchunk; r1 = r0 + a.kchunk < a.M ? r0 + a.kchunk : a.M; }
    const int nrows = (int)(r1 - r0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);      // column sums of the thread's 4 columns of P over its rows
    float4 csum2 = make_float4(0.f, 0.f, 0.f, 0.f);     // PBF: the thread owns 8 columns (csum: 0-3, csum2: 4-7)
    const bool do_csum = a.colsum && j0 == 0;

    const __amdgpu_buffer_rsrc_t sp = Core::make_srd(reinterpret_cast<const float*>(
        reinterpret_cast<const char*>(a.P) + (PBF ? 2 : 4) * (r0 * a.ldp + i0)));
    const __amdgpu_buffer_rsrc_t sq = Core::make_srd(reinterpret_cast<const float*>(
        reinterpret_cast<const char*>(a.Q) + (QBF ? 2 : 4) * (r0 * a.ldq + j0)));
    const int ldp = (int)a.ldp, ldq = (int)a.ldq;
    const int c4 = tid & 31;                             // fp32 operand: the thread's float4 column (4 i's)
    const int c8 = tid & 15;                             // bf16 operand: the thread's 16-byte chunk (8 i's)
    const bool okp = PBF ? i0 + 8 * c8 < a.Nout : i0 + 4 * c4 < a.Nout;
    // two-part right-hand side [Q | Q2] (fp32, columns >= nin_split come from Q2; the host admits it for Nin <= 128): a wave's
    // lanes straddle the split, so every slot is requested from both descriptors with complementary out-of-range masks
    // (an out-of-range lane returns 0 without touching memory) and the two results are OR-ed
    // (bf16-stored Q: the same with the thread's 8-column chunk; nin_split % 8 == 0, host-checked)
    const bool has_q2 = a.Q2 != nullptr;
    // which operand(s) this column tile reads: 1 = Q only, 2 = Q2 only, 3 = both (the split runs through the tile)
    const int q_tile = !has_q2 || j0 + 128 <= a.nin_split ? 1 : (j0 >= a.nin_split ? 2 : 3);
    const int qcol = QBF ? j0 + 8 * c8 : j0 + 4 * c4;        // first of the thread's columns of [Q | Q2]
    const bool in_q2 = has_q2 && qcol >= a.nin_split;
    const bool okq = qcol < a.Nin && !in_q2;
    const bool okq2 = in_q2 && qcol < a.Nin;
    const int ldq2 = (int)a.ldq2, cq2 = qcol - a.nin_split;
    const __amdgpu_buffer_rsrc_t sq2 = Core::make_srd(has_q2 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.Q2) + (QBF ? 2 : 4) * (r0 * a.ldq2))
                                                             : a.Q);
    const float qfloor = a.q_relu ? 0.f : -__builtin_inff();

    // fp32 operand: slot s of a thread: half h = s & 1, row 16 h + (tid >> 5) + 8 (s >> 1) of the 32-row slab (registers
    // h and h + 2); bf16 operand: register h holds the 16 raw bytes of row 16 h + (tid >> 4)
    auto load_half = [&](int h, int k0, bool live, float4 (&rp)[4], float4 (&rq)[4]) {
        if (PBF) {
            const int m = k0 + 16 * h + (tid >> 4);
            rp[h] = Core::srd_load(sp, live && m < nrows && okp ? 2u * (unsigned)(m * ldp + 8 * c8) : Core::SRD_OOB);
        }
        if (QBF) {
            const int m = k0 + 16 * h + (tid >> 4);
            float4 q;
            if (q_tile == 2) {               // the whole column tile lies in the second operand: ONE load (workgroup-uniform)
                q = Core::srd_load(sq2, live && m < nrows && okq2 ? 2u * (unsigned)(m * ldq2 + cq2) : Core::SRD_OOB);
            } else {
                q = Core::srd_load(sq, live && m < nrows && okq ? 2u * (unsigned)(m * ldq + 8 * c8) : Core::SRD_OOB);
                if (q_tile == 3) {           // the tile straddles the split: both descriptors, complementary masks
                    const float4 q2 = Core::srd_load(sq2, live && m < nrows && okq2 ? 2u * (unsigned)(m * ldq2 + cq2) : Core::SRD_OOB);
                    q.x = __uint_as_float(__float_as_uint(q.x) | __float_as_uint(q2.x));
                    q.y = __uint_as_float(__float_as_uint(q.y) | __float_as_uint(q2.y));
                    q.z = __uint_as_float(__float_as_uint(q.z) | __float_as_uint(q2.z));
                    q.w = __uint_as_float(__float_as_uint(q.w) | __float_as_uint(q2.w));
                }
            }
            rq[h] = q;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = k0 + 16 * h + (tid >> 5) + 8 * j;
            const bool ok = live && m < nrows;
            if (!PBF) rp[h + 2 * j] = Core::srd_load(sp, ok && okp ? 4u * (unsigned)(m * ldp + 4 * c4) : Core::SRD_OOB);
            if (!QBF) {
                float4 q = Core::srd_load(sq, ok && okq ? 4u * (unsigned)(m * ldq + 4 * c4) : Core::SRD_OOB);
                if (has_q2 && !QBF) {
                    const float4 q2 = Core::srd_load(sq2, ok && okq2 ? 4u * (unsigned)(m * ldq2 + cq2) : Core::SRD_OOB);
                    q.x = __uint_as_float(__float_as_uint(q.x) | __float_as_uint(q2.x));
                    q.y = __uint_as_float(__float_as_uint(q.y) | __float_as_uint(q2.y));
                    q.z = __uint_as_float(__float_as_uint(q.z) | __float_as_uint(q2.z));
                    q.w = __uint_as_float(__float_as_uint(q.w) | __float_as_uint(q2.w));
                }
                rq[h + 2 * j] = q;
            }
        }
    };
    auto store_half = [&](int h, const float4 (&rp)[4], const float4 (&rq)[4]) {
        char* st = ldsb + h * WS_STAGE_B;
        if (PBF) {
            const float4 raw = rp[h];
            *reinterpret_cast<float4*>(st + ws_off(tid >> 4, c8)) = raw;
            const float4 lo = widen_bf16x4(__float_as_uint(raw.x), __float_as_uint(raw.y));
            const float4 hi = widen_bf16x4(__float_as_uint(raw.z), __float_as_uint(raw.w));
            csum.x += lo.x; csum.y += lo.y; csum.z += lo.z; csum.w += lo.w;
            csum2.x += hi.x; csum2.y += hi.y; csum2.z += hi.z; csum2.w += hi.w;
        }
        if (QBF) *reinterpret_cast<float4*>(st + WS_OPER_B + ws_off(tid >> 4, c8)) = rq[h];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = ws_off((tid >> 5) + 8 * j, c4 >> 1) + 8 * (c4 & 1);
            if (!PBF) {
                const float4 p = rp[h + 2 * j];
                csum.x += p.x; csum.y += p.y; csum.z += p.z; csum.w += p.w;
                SplitCore<false, NP>::split_store(st + off, p, WS_PLANE_B);
            }
            if (!QBF) {
                float4 q = rq[h + 2 * j];
                q.x = fmaxf(q.x, qfloor); q.y = fmaxf(q.y, qfloor); q.z = fmaxf(q.z, qfloor); q.w = fmaxf(q.w, qfloor);
                SplitCore<false, NP>::split_store(st + WS_OPER_B + off, q, WS_PLANE_B);
            }
        }
    };
    struct Frags { bf16x8 a[2][NP], b[2][NP]; };
    // lane 4q+p of a 16-lane group addresses block row q, columns 4p..4p+3; the group receives 4 k x 16 columns
    // transposed.  Groups 0,1 take columns 0-15 / 16-31 of the 32-column tile at k = 0..3, groups 2,3 the same
    // columns at k = 8..11; a second read 4 rows further down completes the 8-k operand.
    const int gq = (lane >> 2) & 3, gp = lane & 3, gg = lane >> 4;
    auto read_frags = [&](int h) {
        const char* st = ldsb + h * WS_STAGE_B;
        Frags f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ca = (wr * 64 + t * 32 + 16 * (gg & 1) + 4 * gp) >> 3;      // 16-byte chunk of the lane's 4 columns
            const int cb = (wc * 64 + t * 32 + 16 * (gg & 1) + 4 * gp) >> 3;
            const int sub = 8 * (gp & 1);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                s16x4 lo, hi;
                const int m0 = 8 * (gg >> 1) + gq;
                lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + p * WS_PLANE_B + ws_off(m0, ca) + sub));
                hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + p * WS_PLANE_B + ws_off(m0 + 4, ca) + sub));
                f.a[t][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + WS_OPER_B + p * WS_PLANE_B + ws_off(m0, cb) + sub));
                hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + WS_OPER_B + p * WS_PLANE_B + ws_off(m0 + 4, cb) + sub));
                f.b[t][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
        return f;
    };
    auto mfmas = [&](const Frags& f) {
        constexpr int PA[6] = {NP == 3 ? 2 : 0, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < (NP == 3 ? 6 : 1); ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mi][PA[q]], f.b[ni][PB[q]], acc[mi][ni], 0, 0, 0);
    };
    auto fused = [&](int hs, int hc, int k_next, bool live, float4 (&rp)[4], float4 (&rq)[4]) {
        __builtin_amdgcn_sched_barrier(0);
        const Frags f = read_frags(hc);
        store_half(hs, rp, rq);
        mfmas(f);
        load_half(hs, k_next, live, rp, rq);
        if (NP == 3) {
            __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
#pragma unroll
            for (int r = 0; r < 24; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                if (r & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (r >= 18 && r < 22) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nit = (nrows + 31) / 32;
    if (nit > 0) {
        float4 rp[4], rq[4];
        load_half(0, 0, true, rp, rq);
        load_half(1, 0, true, rp, rq);
        store_half(0, rp, rq);
        store_half(1, rp, rq);
        load_half(0, 32, nit > 1, rp, rq);
        load_half(1, 32, nit > 1, rp, rq);
        __syncthreads();
        mfmas(read_frags(0));
        for (int it = 0; it + 1 < nit; ++it) {
            const bool live = it + 2 < nit;
            __syncthreads();
            fused(0, 1, (it + 2) * 32, live, rp, rq);
            __syncthreads();
            fused(1, 0, (it + 2) * 32, live, rp, rq);
        }
        __syncthreads();
        mfmas(read_frags(1));
    }
    const long stride = (long)a.Nout * a.Nin + (a.colsum ? a.Nout : 0);
    float* out = a.slab + (long)chunk * stride;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int i = i0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (i < a.Nout) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    int j = j0 + wc * 64 + ni * 32 + lr;
                    if (j < a.Nin) out[(long)i * a.Nin + j] = acc[mi][ni][reg];
                }
            }
        }
    if (do_csum && !PBF) {      // 8 threads (tid >> 5) hold partial sums of the same 4 columns: fixed-order reduction through LDS
        __syncthreads();
        reinterpret_cast<float4*>(lds)[tid] = csum;
        __syncthreads();
        if (tid < 128 && i0 + tid < a.Nout) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += lds[(g * 32 + (tid >> 2)) * 4 + (tid & 3)];
            out[(long)a.Nout * a.Nin + i0 + tid] = s;
        }
    }
    if (do_csum && PBF) {       // 16 threads (tid >> 4) hold partial sums of the same 8 columns
        __syncthreads();
        reinterpret_cast<float4*>(lds)[2 * tid] = csum;
        reinterpret_cast<float4*>(lds)[2 * tid + 1] = csum2;
        __syncthreads();
        if (tid < 128 && i0 + tid < a.Nout) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) s += lds[(g * 16 + (tid >> 3)) * 8 + (tid & 7)];
            out[(long)a.Nout * a.Nin + i0 + tid] = s;
        }
    }
}

// ---- both operands STORED as bf16 (the bf16-row layout of the cfg-5 path): deep register ring --------------------------------
// Same tile, chunking, LDS image, MFMA and summation order as wgrad_split_kernel<1, true, true> (bit-identical slabs), but the rows
// are requested D half slabs ahead instead of one: a half slab of both operands is 8 KB per workgroup and the K loop consumes one per
// ~0.1 us, so with one half slab of lead every step waited out a full memory latency (wait_any 0.64, 4.1 TB/s).  D half slabs of
// lead keep D x 8 KB per workgroup in flight (Little: 8 TB/s x ~2 us / 256 CUs = 64 KB per CU).  The ring lives in registers
// (two float4 per slot), statically indexed: the K loop is unrolled D times.  A two-part right-hand side [Q | Q2] is taken when the
// split falls on a column-tile boundary (the tile reads one of the two).
// MI = 32-row MFMA tiles of a wave along the output rows: 2 = the 128 x 128 tile of wgrad_split_kernel; 4 = a 256 x 128 tile (the
// left operand's half slab is two 128-column images, wave row wr reads image wr): a column tile's rows of P cross L2 -> LDS once per
// 256 instead of once per 128 output rows (the paired gradients dhp^T [q | A_hat x], dzr^T [h | A_hat x]: 1088 / 2176 instead of
// 1536 / 3072 operand elements per row and chunk), twice the MFMA work per barrier.  Every output element still sums the same
// products in the same order: the slabs do not depend on MI.
template <int D, int MI>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_ring_kernel(WgradArgs a) {
    static_assert(D >= 2 && D % 2 == 0, "ring depth: even (the LDS stage of a slot is static)");
    static_assert(MI == 2 || MI == 4, "128- or 256-row tile");
    constexpr int NPL = MI / 2;                          // 128-column images of P per half slab
    constexpr int TI = 64 * MI;                          // output rows of a tile
    constexpr int WS_Q_B = NPL * WS_PLANE_B, WS_STAGE_B = (NPL + 1) * WS_PLANE_B;
    using Core = FastCore<true, false>;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_i = (a.Nout + TI - 1) / TI, tiles_j = (a.Nin + 127) / 128;
    const int tpc = tiles_i * tiles_j;
    int tile, chunk;
    {   // all tiles of one row chunk on the same XCD (see wgrad_kernel)
        const int nfull = (a.nchunks / 8) * 8;
        const int b = blockIdx.x;
        if (b < nfull * tpc) {
            const int xcd = b & 7, li = b >> 3;
            chunk = (li / tpc) * 8 + xcd;
            tile = li % tpc;
        } else {
            const int r = b - nfull * tpc;
            chunk = nfull + r / tpc;
            tile = r % tpc;
        }
    }
    const int i0 = (tile / tiles_j) * TI, j0 = (tile % tiles_j) * 128;
    long r0, r1;
    if (a.chunk_tab) { r0 = a.chunk_tab[2 * chunk]; r1 = a.chunk_tab[2 * chunk + 1]; }
    else { r0 = (long)chunk * a.kchunk; r1 = r0 + a.kchunk < a.M ? r0 + a.kchunk : a.M; }
    const int nrows = (int)(r1 - r0);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 csum[NPL][2];                                 // column sums of the thread's 8 columns of each image over its rows
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) csum[pl][0] = csum[pl][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_csum = a.colsum && j0 == 0;

    const bool second = a.Q2 != nullptr && j0 >= a.nin_split;        // workgroup-uniform: this column tile lies in Q2
    const int ldp = (int)a.ldp, ldq = second ? (int)a.ldq2 : (int)a.ldq;
    // Descriptors that END with the chunk's last row: a row past the chunk is out of range by itself (the hardware returns zeros
    // without touching memory), so the K loop carries no row test -- one running byte offset per operand.  A thread whose columns
    // lie outside the matrix starts at 2^31: beyond every range the host admits (chunk rows x row bytes < 2^31), and the walk
    // ((rows + 32 + 16 D) x row bytes) cannot wrap it back into range.
    auto chunk_srd = [](const char* base, unsigned bytes) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t sp = chunk_srd(reinterpret_cast<const char*>(a.P) + 2 * (r0 * a.ldp + i0), 2u * (unsigned)nrows * (unsigned)ldp);
    const __amdgpu_buffer_rsrc_t sq = chunk_srd(second ? reinterpret_cast<const char*>(a.Q2) + 2 * (r0 * a.ldq2 + (j0 - a.nin_split))
                                                       : reinterpret_cast<const char*>(a.Q) + 2 * (r0 * a.ldq + j0), 2u * (unsigned)nrows * (unsigned)ldq);
    const int c8 = tid & 15, mrow = tid >> 4;            // the thread's 16-byte chunk (8 columns) and row of a half slab
    constexpr unsigned MASKED = 0x80000000u;
    // (MI = 4 is launched for Nout % 256 == 0 only: both images of P lie inside the matrix)
    unsigned vp = i0 + 8 * c8 < a.Nout ? 2u * (unsigned)(mrow * ldp + 8 * c8) : MASKED;      // running byte offsets: half slab g
    unsigned vq = j0 + 8 * c8 < a.Nin ? 2u * (unsigned)(mrow * ldq + 8 * c8) : MASKED;
    const unsigned stepp = 32u * (unsigned)ldp, stepq = 32u * (unsigned)ldq;                   // bytes per half slab (16 rows)

    float4 rp[D][NPL], rq[D];
    auto load = [&](int slot) {                          // the NEXT half slab of the chunk (requests are issued in row order)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) rp[slot][pl] = Core::srd_load(sp, vp + 256u * pl);
        rq[slot] = Core::srd_load(sq, vq);
        vp += stepp;
        vq += stepq;
        asm volatile("" : "+v"(vp), "+v"(vq));           // ONE running offset per operand (not one per unrolled slot)
    };
    const int lds_w = ws_off(mrow, c8);
    struct Frags { bf16x8 a[MI], b[2]; };
    const int gq = (lane >> 2) & 3, gp = lane & 3, gg = lane >> 4;
    auto read_frags = [&](int stage) {
        const char* st = ldsb + stage * WS_STAGE_B;
        Frags f;
        const int sub = 8 * (gp & 1);
        const int m0 = 8 * (gg >> 1) + gq;
#pragma unroll
        for (int t = 0; t < MI; ++t) {
            const int col = wr * (32 * MI) + t * 32;         // first column of the wave's tile t in the 64 MI-column left operand
            const int ca = ((col & 127) + 16 * (gg & 1) + 4 * gp) >> 3;
            const char* pl = st + (col >> 7) * WS_PLANE_B;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pl + ws_off(m0, ca) + sub));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pl + ws_off(m0 + 4, ca) + sub));
            f.a[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int cb = (wc * 64 + t * 32 + 16 * (gg & 1) + 4 * gp) >> 3;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + WS_Q_B + ws_off(m0, cb) + sub));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + WS_Q_B + ws_off(m0 + 4, cb) + sub));
            f.b[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        return f;
    };
    auto mfmas = [&](const Frags& f, int half) {             // half 0 / 1: the first / last MI / 2 row tiles
#pragma unroll
        for (int m = 0; m < MI / 2; ++m)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int mi = half * (MI / 2) + m;
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mi], f.b[ni], acc[mi][ni], 0, 0, 0);
            }
    };

    const int G = 2 * ((nrows + 31) / 32);               // half slabs, the last one possibly all zero (as wgrad_split_kernel walks them)
    // The K loop exists twice: with the column sums of P (the workgroups of column tile 0 when a bias gradient is asked for) and
    // without -- 12 of a step's ~30 vector instructions, and the loop is bound by instruction issue once the ring hides the latency.
    auto k_loop = [&](auto cs_tag) {
        constexpr bool CS = decltype(cs_tag)::value;
        auto store = [&](int slot, int stage) {
            char* st = ldsb + stage * WS_STAGE_B;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const float4 raw = rp[slot][pl];
                *reinterpret_cast<float4*>(st + pl * WS_PLANE_B + lds_w) = raw;
                if (CS) {
                    const float4 lo = widen_bf16x4(__float_as_uint(raw.x), __float_as_uint(raw.y));
                    const float4 hi = widen_bf16x4(__float_as_uint(raw.z), __float_as_uint(raw.w));
                    float4& c0 = csum[pl][0];
                    float4& c1 = csum[pl][1];
                    c0.x += lo.x; c0.y += lo.y; c0.z += lo.z; c0.w += lo.w;
                    c1.x += hi.x; c1.y += hi.y; c1.z += hi.z; c1.w += hi.w;
                    // pins the sums to this place (instruction selection otherwise sinks the whole chain to the end of the unrolled
                    // turn, keeping every slot's old rows alive past its reload)
                    asm volatile("" : "+v"(c0.x), "+v"(c0.y), "+v"(c0.z), "+v"(c0.w), "+v"(c1.x), "+v"(c1.y), "+v"(c1.z), "+v"(c1.w));
                }
            }
            *reinterpret_cast<float4*>(st + WS_Q_B + lds_w) = rq[slot];
        };
#pragma unroll
        for (int u = 0; u < D; ++u) load(u);
        store(0, 0);
        load(0);
        // step g: multiply half slab g (stage g & 1) while half slab g + 1 goes to the other stage and g + 1 + D is requested
        auto step = [&](int u) {
            const int sn = (u + 1) % D;
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            const Frags f = read_frags(u & 1);
            store(sn, (u + 1) & 1);
            mfmas(f, 0);
            // the slot's old contents are consumed (LDS write, column sums) before it is requested again: if the scheduler lets the two
            // live ranges overlap, the new rows land in other registers and are COPIED into the slot at the loop's back edge -- behind a
            // wait for the whole ring
            __builtin_amdgcn_sched_barrier(0);
            load(sn);
            mfmas(f, 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        // whole turns of the ring as ONE basic block (a test per step gives every step a second predecessor, and the wait-count
        // insertion then assumes the slot's loads are the youngest: vmcnt(0) before every store); the last G % D steps test
        int gb = 0;
        for (; gb + D <= G; gb += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) step(u);
        }
#pragma unroll
        for (int u = 0; u < D - 1; ++u)
            if (gb + u < G) step(u);
    };
    if (G > 0) {
        if (do_csum || a.all_csum) k_loop(std::true_type{});
        else k_loop(std::false_type{});
    }
    const long stride = (long)a.Nout * a.Nin + (a.colsum ? a.Nout : 0);
    float* out = a.slab + (long)chunk * stride;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = i0 + wr * (32 * MI) + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (i < a.Nout) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int j = j0 + wc * 64 + ni * 32 + lr;
                    if (j < a.Nin) out[(long)i * a.Nin + j] = acc[mi][ni][reg];
                }
            }
        }
    if (do_csum) {       // 16 threads (tid >> 4) hold partial sums of the same 8 columns of an image: fixed-order sum through LDS
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            reinterpret_cast<float4*>(lds)[512 * pl + 2 * tid] = csum[pl][0];
            reinterpret_cast<float4*>(lds)[512 * pl + 2 * tid + 1] = csum[pl][1];
        }
        __syncthreads();
        if (tid < 128 * NPL && i0 + tid < a.Nout) {
            const int pl = tid >> 7, c = tid & 127;
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) s += lds[2048 * pl + (g * 16 + (c >> 3)) * 8 + (c & 7)];
            out[(long)a.Nout * a.Nin + i0 + tid] = s;
        }
    }
}

// Generic fallback (scalar-guarded loads) for operands that are not 16-byte tileable, e.g. the (N, O) head gradient.
template <int BNW>   // 128: waves 2x2, each 2x2 MFMA tiles;  32: waves 4x1, each one MFMA tile
__global__ __launch_bounds__(256, 2) void wgrad_kernel_generic(WgradArgs a) {
    constexpr int WM = BNW == 128 ? 2 : 1, WN = BNW == 128 ? 2 : 1;
    constexpr int LDQ = BNW + 4;
    constexpr int P_TILE = W_BK * W_LDP, Q_TILE = W_BK * LDQ;
    constexpr int QSLOTS = (W_BK * BNW / 4) / 256;          // float4 slots per thread for Q (4 or 1)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = BNW == 128 ? (wid >> 1) : wid, wc = BNW == 128 ? (wid & 1) : 0;
    const int tiles_i = (a.Nout + 127) / 128, tiles_j = (a.Nin + BNW - 1) / BNW;
    const int tile = blockIdx.x % (tiles_i * tiles_j), chunk = blockIdx.x / (tiles_i * tiles_j);
    const int i0 = (tile / tiles_j) * 128, j0 = (tile % tiles_j) * BNW;
    long r0, r1;
    if (a.chunk_tab) { r0 = a.chunk_tab[2 * chunk]; r1 = a.chunk_tab[2 * chunk + 1]; }
    else { r0 = (long)chunk * a.kchunk; r1 = r0 + a.kchunk < a.M ? r0 + a.kchunk : a.M; }
    const bool vecP = (a.ldp % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.P) & 15) == 0);
    const bool vecQ = (a.ldq % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.Q) & 15) == 0);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum = 0.f;

    auto load = [&](long k0, float4 (&rp)[4], float4 (&rq)[QSLOTS]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            int slot = tid + 256 * s;
            long m = k0 + (slot >> 5);
            int i = i0 + 4 * (slot & 31);
            rp[s] = (m < r1 && i < a.Nout) ? ld4_guard(a.P + m * a.ldp + i, a.Nout - i, vecP) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < QSLOTS; ++s) {
            int slot = tid + 256 * s;
            long m = k0 + slot / (BNW / 4);
            int j = j0 + 4 * (slot % (BNW / 4));
            float4 v = (m < r1 && j < a.Nin) ? ld4_guard(a.Q + m * a.ldq + j, a.Nin - j, vecQ) : make_float4(0, 0, 0, 0);
            if (a.q_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            rq[s] = v;
        }
    };
    auto store = [&](int stage, const float4 (&rp)[4], const float4 (&rq)[QSLOTS]) {
        float* lp = lds + stage * (P_TILE + Q_TILE);
        float* lq = lp + P_TILE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            int slot = tid + 256 * s;
            *reinterpret_cast<float4*>(lp + (slot >> 5) * W_LDP + 4 * (slot & 31)) = rp[s];
        }
#pragma unroll
        for (int s = 0; s < QSLOTS; ++s) {
            int slot = tid + 256 * s;
            *reinterpret_cast<float4*>(lq + (slot / (BNW / 4)) * LDQ + 4 * (slot % (BNW / 4))) = rq[s];
        }
    };
    auto compute = [&](int stage) {
        const float* lp = lds + stage * (P_TILE + Q_TILE);
        const float* lq = lp + P_TILE;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kg * 8 + lh * 4 + j;
                float av[WM], bv[WN];
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) av[mi] = lp[k * W_LDP + wr * (32 * WM) + mi * 32 + lr];
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) bv[ni] = lq[k * LDQ + wc * (32 * WN) + ni * 32 + lr];
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
            }
        }
        if (a.colsum && j0 == 0 && tid < 128) {
#pragma unroll 8
            for (int k = 0; k < W_BK; ++k) csum += lp[k * W_LDP + tid];
        }
    };

    const long nit = (r1 - r0 + W_BK - 1) / W_BK;
    if (nit > 0) {
        float4 rp[4], rq[QSLOTS];
        load(r0, rp, rq);
        store(0, rp, rq);
        __syncthreads();
        for (long it = 0; it < nit; ++it) {
            const bool more = it + 1 < nit;
            if (more) load(r0 + (it + 1) * W_BK, rp, rq);
            compute((int)(it & 1));
            if (more) store((int)((it + 1) & 1), rp, rq);
            __syncthreads();
        }
    }
    const long stride = (long)a.Nout * a.Nin + (a.colsum ? a.Nout : 0);
    float* out = a.slab + (long)chunk * stride;
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int i = i0 + wr * (32 * WM) + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (i < a.Nout) {
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    int j = j0 + wc * (32 * WN) + ni * 32 + lr;
                    if (j < a.Nin) out[(long)i * a.Nin + j] = acc[mi][ni][reg];
                }
            }
        }
    if (a.colsum && j0 == 0 && tid < 128 && i0 + tid < a.Nout) out[(long)a.Nout * a.Nin + i0 + tid] = csum;
}

long wgrad_slab_stride(const WgradArgs& a) { return (long)a.Nout * a.Nin + (a.colsum ? a.Nout : 0); }

static int launch_wgrad_impl(const WgradArgs& a, hipStream_t st);
// Every column tile of a row chunk forms the column sums of P although only tile 0 stores them:
// the tiles of a chunk share P (and Q between row tiles) through their XCD's L2 and only find each other's lines there while they walk
// the chunk in step -- with less work the other tiles run ahead and every tile reads its operands from HBM (measured on the bf16
// ring kernel: 0.90 / 0.51 ms against 0.71 / 0.45 ms for the two paired gradients of the cfg-5 shard).
int launch_wgrad(const WgradArgs& a, hipStream_t st) {
    WgradArgs am = a;
    am.all_csum = a.colsum && a.p_bf16 && a.q_bf16;     // (fp32 kernels: measured no gain, +0.02 ms on the MFMA-bound wgrad3_kernel)
    return launch_wgrad_impl(am, st);
}
static int launch_wgrad_impl(const WgradArgs& a, hipStream_t st) {
    REGT_CHECK_ARG(a.Nout > 0 && a.Nin > 0 && a.nchunks > 0, "wgrad: empty problem");
    // a two-part right-hand side runs on the skinny fp32 kernel, except under the bf16 arithmetic with Nin <= 128 (one
    // column tile of the bf16-pipe kernel: at F = 64 the fused dA0 / dA_r gradient is fp32-MFMA-bound on the skinny kernel)
    // (any width when both parts are stored as bf16: the fused [q | A_hat x] / [h | A_hat x] gradients of the bf16-row layout)
    const bool q2_split = a.Q2 && gemm_mode() == 2 && a.Nin > 32 && (a.Nin <= 128 || a.q_bf16) && !a.q_relu && a.nin_split % (a.q_bf16 ? 8 : 4) == 0 &&
                          !fp32_core_wide() && (!a.q_bf16 || a.ldq2 % 8 == 0);
    // (a bf16-stored right-hand side of width <= 32 -- A_hat x rows at F = 32 -- also takes the bf16-pipe kernel: the skinny one
    // stages fp32 rows only; the stage is HBM-bound on its left operand either way)
    const bool wide = (a.Nin > 32 || (a.q_bf16 && gemm_mode() == 2 && !a.Q2)) && (!a.Q2 || q2_split);
    const bool fast = a.ldp % 4 == 0 && a.ldq % 4 == 0 && a.Nout % 4 == 0 && a.Nin % 4 == 0 && a16(a.P) && a16(a.Q) &&
                      a.ldp < (1L << 20) && a.ldq < (1L << 20) && (a.chunk_tab || a.kchunk <= 65536) &&
                      (!a.Q2 || (a.ldq2 % 4 == 0 && a.ldq2 < (1L << 20) && a16(a.Q2) && a.nin_split % 32 == 0));
    // (fp32 rows, 32 < Nin <= 64 -- the fused [x | L~ x] right-hand side at F = 32: one 64-column tile; REGT_WGRAD_BNW64=0: two of 32)
    const int bnw64 = wgrad_bnw64_option(-1);
    const bool mid = !wide && fast && bnw64 && !a.p_bf16 && !a.q_bf16 && a.Nin > 32 && a.Nin <= 64 && (!a.Q2 || a.nin_split < 64);
    const int bnw = wide ? 128 : (mid ? 64 : 32);
    long blocks = (long)cdiv(a.Nout, 128) * cdiv(a.Nin, bnw) * a.nchunks;
    REGT_CHECK_ARG(blocks < (1L << 31), "wgrad: too many blocks");
    size_t lds = 2 * (size_t)(W_BK * W_LDP + W_BK * (bnw + 4)) * 4;
    REGT_CHECK_ARG(!a.Q2 || fast, "wgrad: a second right-hand operand needs 16-byte tileable operands and nin_split %% 32 == 0");
    REGT_CHECK_ARG(!(a.p_bf16 || a.q_bf16) || fast, "wgrad: bf16 operands need the vector kernels");
    if (wide) {
        static bool attr_done = false, attr_done_g = false, attr_done_s = false;
        if (fast && gemm_mode() == 1) {
            if (int rc = set_lds_once(&wgrad_split_kernel<3>, 4 * 3 * WS_PLANE_B, &attr_done_s)) return rc;
            hipLaunchKernelGGL(wgrad_split_kernel<3>, dim3((unsigned)blocks), dim3(256), 4 * 3 * WS_PLANE_B, st, a);
        } else if (fast && gemm_mode() == 2) {
            REGT_CHECK_ARG(!(a.p_bf16 || a.q_bf16) || (a.Nout % 8 == 0 && a.Nin % 8 == 0 && a.ldp % 8 == 0 && a.ldq % 8 == 0 && !a.q_relu),
                           "wgrad: bf16-stored operands need 8-element aligned rows");
            const size_t lb = 4 * 1 * WS_PLANE_B;
            const int ring = wgrad_ring_depth();
            // (the ring kernel's descriptors end with the chunk: chunk rows x row bytes must stay below 2^31)
            const long ld_max = std::max(a.ldp, std::max(a.ldq, a.Q2 ? a.ldq2 : 0L));
            const long rows_max = a.chunk_tab ? a.M : (long)a.kchunk + 32;
            const bool ring_ok = a.p_bf16 && a.q_bf16 && ring > 0 && (!a.Q2 || a.nin_split % 128 == 0) && 2 * rows_max * ld_max < (1L << 31);
            auto launch_ring = [&](auto kernel, long nblocks, size_t need) -> int {
                static bool attr_done_r = false;
                const size_t bytes = need;
                if (bytes > 48 * 1024) { if (int rc = set_lds_once(kernel, (int)bytes, &attr_done_r)) return rc; }
                hipLaunchKernelGGL(kernel, dim3((unsigned)nblocks), dim3(256), bytes, st, a);
                return REGT_OK;
            };
            // 256-row tiles where the output has them (regt_set_option("wgrad_tile", 128 | 256) / REGT_WGRAD_TILE); ring of 2 there
            // (REGT_WGRAD_RING256 / "wgrad_ring256" = 2 | 4; 6 half slabs of three 16-byte loads spill).  Two beats four, 0.585 + 0.350
            // against 0.63 + 0.383 ms at the cfg-5 shard: what is in flight (workgroups x slots x 12 KB per XCD) competes with the lines
            // the chunk's other tiles are about to ask for in the 4 MiB L2, and the tile that comes second finds its rows there anyway
            if (ring_ok && wgrad_tile_rows() == 256 && a.Nout % 256 == 0) {
                const long blocks4 = (long)(a.Nout / 256) * cdiv(a.Nin, 128) * a.nchunks;
                if (wgrad_ring256_option(-1) == 2) { if (int rc = launch_ring(&wgrad_bf16_ring_kernel<2, 4>, blocks4, 2 * 3 * WS_PLANE_B)) return rc; }
                else if (int rc = launch_ring(&wgrad_bf16_ring_kernel<4, 4>, blocks4, 2 * 3 * WS_PLANE_B)) return rc;
            }
            else if (ring_ok && ring >= 8) { if (int rc = launch_ring(&wgrad_bf16_ring_kernel<8, 2>, blocks, lb)) return rc; }
            else if (ring_ok && ring >= 6) { if (int rc = launch_ring(&wgrad_bf16_ring_kernel<6, 2>, blocks, lb)) return rc; }
            else if (ring_ok) { if (int rc = launch_ring(&wgrad_bf16_ring_kernel<4, 2>, blocks, lb)) return rc; }
            else if (a.p_bf16 && a.q_bf16) hipLaunchKernelGGL((wgrad_split_kernel<1, true, true>), dim3((unsigned)blocks), dim3(256), lb, st, a);
            else if (a.p_bf16) hipLaunchKernelGGL((wgrad_split_kernel<1, true, false>), dim3((unsigned)blocks), dim3(256), lb, st, a);
            else if (a.q_bf16) hipLaunchKernelGGL((wgrad_split_kernel<1, false, true>), dim3((unsigned)blocks), dim3(256), lb, st, a);
            else hipLaunchKernelGGL((wgrad_split_kernel<1, false, false>), dim3((unsigned)blocks), dim3(256), lb, st, a);
        } else if (fast && !fp32_core_wide()) {
            REGT_CHECK_ARG(!(a.p_bf16 || a.q_bf16), "wgrad: bf16-stored operands with the fp32 kernel");
            hipLaunchKernelGGL(wgrad3_kernel, dim3((unsigned)blocks), dim3(256), 4 * 16 * 132 * 4, st, a);
        } else if (fast) {
            REGT_CHECK_ARG(!(a.p_bf16 || a.q_bf16), "wgrad: bf16-stored operands with the fp32 wide kernel");
            if (int rc = set_lds_once(&wgrad_kernel<128>, (int)lds, &attr_done)) return rc;
            hipLaunchKernelGGL(wgrad_kernel<128>, dim3((unsigned)blocks), dim3(256), lds, st, a);
        } else {
            if (int rc = set_lds_once(&wgrad_kernel_generic<128>, (int)lds, &attr_done_g)) return rc;
            hipLaunchKernelGGL(wgrad_kernel_generic<128>, dim3((unsigned)blocks), dim3(256), lds, st, a);
        }
    } else if (fast) {
        REGT_CHECK_ARG(!a.q_bf16 && (!a.p_bf16 || (a.Nout % 8 == 0 && a.ldp % 8 == 0)), "wgrad: skinny kernel takes a bf16-stored P only");
        if (a.p_bf16) hipLaunchKernelGGL((wgrad_kernel<32, true>), dim3((unsigned)blocks), dim3(256), lds, st, a);
        else if (mid) {
            static bool attr_done_m = false;
            if (int rc = set_lds_once(&wgrad_kernel<64>, (int)lds, &attr_done_m)) return rc;      // 51 200 B of dynamic LDS
            hipLaunchKernelGGL(wgrad_kernel<64>, dim3((unsigned)blocks), dim3(256), lds, st, a);
        }
        else hipLaunchKernelGGL(wgrad_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    } else {
        hipLaunchKernelGGL(wgrad_kernel_generic<32>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    }
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

// Eight adjacent lanes share one output element: the chunk range is strided over them and combined with
// a fixed xor-shuffle tree, so the order of the additions is fixed (deterministic) and small outputs
// (the C x F gradients) still fill the chip.
// 16-byte form of the main part (round 4): a lane owns FOUR consecutive output elements (one float4 per chunk: a quarter of the load
// instructions, 512 contiguous bytes per 32-lane group and chunk), a workgroup 128.  Every element is summed over the same chunks in
// the same order and the eight partial sums meet in the same tree as in the scalar form below: bit-identical results.  Needs
// Nin % 4 == 0 and 16-byte aligned slab rows / output rows (wgrad_reduce_vec_ok).
__device__ __forceinline__ bool wgrad_reduce_vec_ok(const WgradReduceArgs& a) {
    return a.Nin % 4 == 0 && a.slab_stride % 4 == 0 && a.elem_offset % 4 == 0 && (a.slab_ld == 0 || a.slab_ld % 4 == 0) && a.ldo % 4 == 0 &&
           a.group_stride % 4 == 0 && ((reinterpret_cast<unsigned long long>(a.slab) | reinterpret_cast<unsigned long long>(a.out)) & 15) == 0;
}
__device__ __forceinline__ void wgrad_reduce_main_v4(const WgradReduceArgs& a, long block, long nblocks, float4 (*part)[33]) {
    const long per = (long)a.Nout * a.Nin;
    const long total = per * a.ngroups;
    const int sub = threadIdx.x >> 5, el = threadIdx.x & 31;
    for (long base = block * 128; base < total; base += nblocks * 128) {
        const long idx = base + 4 * el;
        const bool valid = idx < total;                 // total % 4 == 0: a float4 never straddles the end (or a row: Nin % 4 == 0)
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int g = 0;
        long e = 0;
        if (valid) {
            g = (int)(idx / per);
            e = idx - (long)g * per;
            const long se = a.slab_ld ? (e / a.Nin) * a.slab_ld + e % a.Nin : e;
            const float* p = a.slab + a.elem_offset + se;
            int c = sub;
#define REGT_ADD4(v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            if (!a.chunk_group) {
                for (; c + 24 < a.nchunks; c += 32) {
                    const float4 v0 = *reinterpret_cast<const float4*>(p + (long)c * a.slab_stride);
                    const float4 v1 = *reinterpret_cast<const float4*>(p + (long)(c + 8) * a.slab_stride);
                    const float4 v2 = *reinterpret_cast<const float4*>(p + (long)(c + 16) * a.slab_stride);
                    const float4 v3 = *reinterpret_cast<const float4*>(p + (long)(c + 24) * a.slab_stride);
                    REGT_ADD4(v0) REGT_ADD4(v1) REGT_ADD4(v2) REGT_ADD4(v3)
                }
            }
            for (; c < a.nchunks; c += 8)
                if (!a.chunk_group || a.chunk_group[c] == g + a.group_base) {
                    const float4 v = *reinterpret_cast<const float4*>(p + (long)c * a.slab_stride);
                    REGT_ADD4(v)
                }
#undef REGT_ADD4
        }
        part[sub][el] = s;
        __syncthreads();
        if (valid && sub == 0) {
#define REGT_TREE(k) (((part[0][el].k + part[1][el].k) + (part[2][el].k + part[3][el].k)) + ((part[4][el].k + part[5][el].k) + (part[6][el].k + part[7][el].k)))
            s = make_float4(REGT_TREE(x), REGT_TREE(y), REGT_TREE(z), REGT_TREE(w));
#undef REGT_TREE
            const int i = (int)(e / a.Nin), j = (int)(e % a.Nin);
            float4* o = reinterpret_cast<float4*>(a.out + (long)g * a.group_stride + (long)i * a.ldo + j);
            if (a.accumulate) { const float4 t = *o; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
            *o = s;
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void wgrad_reduce_body(const WgradReduceArgs& a, long block, long nblocks) {
    // A workgroup owns 32 consecutive output elements; its eight 32-lane groups each sum every eighth chunk of them (a wave
    // reads two chunks x 128 contiguous bytes per step -- with the eight partial sums of an element in ADJACENT lanes a wave
    // touched eight chunks x 32 bytes), and the eight partial sums meet in LDS in the association of the former xor-shuffle
    // tree: ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)).  Results are bit-identical to the shuffle version.
    __shared__ float4 part4[8][33];
    float (*part)[33] = reinterpret_cast<float (*)[33]>(&part4[0][0]);
    const long per = (long)a.Nout * a.Nin;
    const long total = per * a.ngroups;
    const long ncs = a.colsum_out ? a.ncolsum : 0;
    const int sub = threadIdx.x >> 5, el = threadIdx.x & 31;
    const bool vec = wgrad_reduce_vec_ok(a);            // (uniform: kernel arguments only)
    if (vec) wgrad_reduce_main_v4(a, block, nblocks, part4);
    // scalar form: everything when the block is not vectorisable, else only the column sums behind the main part
    for (long base = (vec ? total : 0) + block * 32; base < total + ncs; base += nblocks * 32) {
        const long idx = base + el;
        const bool valid = idx < total + ncs;
        float s = 0.f;
        int g = 0;
        long e = 0;
        if (valid) {
            if (idx < total) {
                g = (int)(idx / per);
                e = idx - (long)g * per;
                const long se = a.slab_ld ? (e / a.Nin) * a.slab_ld + e % a.Nin : e;
                const float* p = a.slab + a.elem_offset + se;
                int c = sub;
                if (!a.chunk_group) {
                    // four loads in flight, added in chunk order (the association of the plain loop)
                    for (; c + 24 < a.nchunks; c += 32) {
                        const float v0 = p[(long)c * a.slab_stride], v1 = p[(long)(c + 8) * a.slab_stride];
                        const float v2 = p[(long)(c + 16) * a.slab_stride], v3 = p[(long)(c + 24) * a.slab_stride];
                        s += v0; s += v1; s += v2; s += v3;
                    }
                }
                for (; c < a.nchunks; c += 8)
                    if (!a.chunk_group || a.chunk_group[c] == g + a.group_base) s += p[(long)c * a.slab_stride];
            } else {
                for (int c = sub; c < a.nchunks; c += 8) s += a.slab[(long)c * a.slab_stride + a.colsum_offset + (idx - total)];
            }
        }
        part[sub][el] = s;
        __syncthreads();
        if (valid && sub == 0) {
            s = ((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) + ((part[4][el] + part[5][el]) + (part[6][el] + part[7][el]));
            if (idx < total) {
                const int i = (int)(e / a.Nin), j = (int)(e % a.Nin);
                float* o = a.out + (long)g * a.group_stride + (long)i * a.ldo + j;
                *o = a.accumulate ? *o + s : s;
            } else {
                const int i = (int)(idx - total);
                a.colsum_out[i] = a.accumulate ? a.colsum_out[i] + s : s;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceArgs a) { wgrad_reduce_body(a, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WgradReduceBatch B) {
    int ti = 0;
    while (ti + 1 < B.n && (int)blockIdx.x >= B.block_start[ti + 1]) ++ti;
    wgrad_reduce_body(B.t[ti], (long)blockIdx.x - B.block_start[ti], (long)B.block_start[ti + 1] - B.block_start[ti]);
}

static int wgrad_reduce_blocks(const WgradReduceArgs& a) {
    long total = ((long)a.Nout * a.Nin * a.ngroups + (a.colsum_out ? a.ncolsum : 0)) * 8;
    int blocks = cdiv(total, 256);
    return blocks > 16384 ? 16384 : blocks;
}

int launch_wgrad_reduce_multi(WgradReduceBatch& b, hipStream_t st) {
    REGT_CHECK_ARG(b.n > 0 && b.n <= WR_MAX_TASKS, "wgrad_reduce_multi: %d tasks", b.n);
    int blocks = 0;
    for (int t = 0; t < b.n; ++t) {
        REGT_CHECK_ARG(!(b.t[t].colsum_out && b.t[t].ngroups != 1), "wgrad_reduce: colsum only with one group");
        b.block_start[t] = blocks;
        blocks += wgrad_reduce_blocks(b.t[t]);
    }
    b.block_start[b.n] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, st, b);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

int launch_wgrad_reduce(const WgradReduceArgs& a, hipStream_t st) {
    REGT_CHECK_ARG(!(a.colsum_out && a.ngroups != 1), "wgrad_reduce: colsum only with one group");
    long total = ((long)a.Nout * a.Nin * a.ngroups + (a.colsum_out ? a.ncolsum : 0)) * 8;
    int blocks = cdiv(total, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
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
