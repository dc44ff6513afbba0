// fp32 MFMA GEMM core for gfx950 (CDNA4): 128x128 output tile per 256-thread workgroup,
// 4 waves in a 2x2 arrangement, each wave 2x2 tiles of v_mfma_f32_32x32x2_f32.
//
// The K loop runs over up to three *segments*; a segment is one (A, B) operand pair that
// contributes A_seg[M x K_seg] * B_seg^T to the same accumulator.  That is how the RegT-GCN
// pipeline expresses  h*U2^T + (A_hat X)*G^T  or  x*A0^T + (L~ x)*A_region^T  as ONE launch
// without concatenating operands in HBM.  A segment may be region-masked: it is then repeated
// for every region id present in the row tile, with A rows of other regions zeroed and the B
// pointer advanced by the region stride (per-region composed weights).
//
// Staging: global -> registers (prefetch of tile it+1 issued before the MFMAs of tile it) ->
// LDS (two stages, one barrier per K tile).  LDS rows are padded to 36 floats so that the
// ds_read_b128 operand reads are bank-conflict free (16-B slot index 9*lane mod 16 is a
// permutation inside every 16-lane service group).
#pragma once
#include "common.h"

namespace regt {

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int G_LDS_ROW = GBK + 4;          // 36 floats
constexpr int G_LDS_KROW = GBN + 4;         // 132 floats (B stored k-major)
constexpr int G_A_TILE = GBM * G_LDS_ROW;   // 4608 floats
constexpr int G_B_TILE = GBN * G_LDS_ROW;   // 4608 floats (>= 32*132)
constexpr int G_STAGE = G_A_TILE + G_B_TILE;
constexpr int G_LDS_BYTES = 2 * G_STAGE * 4;  // 73728 B -> 2 workgroups per CU

enum : int {
    SEG_BT = 1,       // B is stored [N][K] (nn.Linear weight layout); else [K][N]
    SEG_RELU_A = 2,   // apply max(0, .) to A while staging
    SEG_REGION = 4,   // region-masked segment (see header comment)
    SEG_VEC_A = 8,    // A rows may be read as aligned float4
    SEG_VEC_B = 16,   // B rows may be read as aligned float4
    SEG_A_BF16 = 64,  // A holds bf16 elements (lda in elements); bf16-operand core only (REGT_GEMM_MODE=bf16): the GEMM-only
                      // intermediates q, dhp, dzp|drp are rounded once by their producer instead of by every consumer
    SEG_B_FRAG = 128, // B0 / B1 point at per-step bf16 copies of the weights in MFMA FRAGMENT order (launch_cvt_bf16_frag): the
                      // 1 KB block (n / 32, k / 16) holds, for lane = 32 ((k % 16) / 8) + n % 32, the 8 bf16 k = 8 (k / 8) .. + 7
                      // of row n; a wave loads its B fragments straight into registers (no LDS); rows padded to 128 with
                      // zeros; b_region_stride in bytes.  bf16-operand core on the scalar-descriptor path only.
    SEG_REPEAT = 32,  // unmasked repeat: rep r = 0..nrep-1 uses A + r*a_rep_stride and B + r*b_region_stride
                      // (overlapping regional graphs: one (L~_r x) operand and one composed weight per region)
};

struct GemmSeg {
    const float* A;
    long lda;
    const float* B0;   // rows/cols n <  nsplit (BT only; nsplit >= N means "all from B0")
    const float* B1;   // rows n >= nsplit
    long ldb;
    int nsplit;
    int K;
    int flags;
    long b_region_stride;
    long a_rep_stride;
    int nrep;
};

struct GemmSegs {
    int nseg;
    GemmSeg seg[3];
    const int* node_region;  // (num nodes) or nullptr
    int row_div;             // global row -> node: grow / row_div
    int num_regions;         // upper bound of the region ids in node_region (sizing of the fast-path table)
};

// Row map: local tile row r -> global row.
struct RowMap {
    long base;
    long mul;
    int nvalid;   // rows r >= nvalid are padding
    __device__ __forceinline__ long grow(int r) const { return base + (long)r * mul; }
};

struct TileIter {
    int seg, region, k0;
};

__device__ __forceinline__ float4 ld4_guard(const float* p, int nvalid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        if (nvalid > 0) v.x = p[0];
        if (nvalid > 1) v.y = p[1];
        if (nvalid > 2) v.z = p[2];
        if (nvalid > 3) v.w = p[3];
    }
    return v;
}

struct GemmCore {
    const GemmSegs& S;
    RowMap rm;
    int n0, N;
    int rmin, rmax;   // region range of this row tile (only for SEG_REGION)
    float* lds;
    int tid, lane, wr, wc;

    __device__ __forceinline__ GemmCore(const GemmSegs& s, RowMap r, int n0_, int N_, float* lds_)
        : S(s), rm(r), n0(n0_), N(N_), rmin(0), rmax(0), lds(lds_) {
        tid = threadIdx.x;
        lane = tid & 63;
        int wid = tid >> 6;
        wr = wid >> 1;
        wc = wid & 1;
    }

    // Region range of the tile rows (wave-uniform result, via LDS).  Call once before run().
    __device__ __forceinline__ void find_regions() {
        bool any = false;
        for (int s = 0; s < S.nseg; ++s) any |= (pick(s).flags & SEG_REGION) != 0;
        if (!any) return;
        int* red = reinterpret_cast<int*>(lds);
        if (tid == 0) { red[0] = 0x7fffffff; red[1] = -1; }
        __syncthreads();
        if (tid < GBM && tid < rm.nvalid) {
            int reg = S.node_region[rm.grow(tid) / S.row_div];
            atomicMin(&red[0], reg);
            atomicMax(&red[1], reg);
        }
        __syncthreads();
        rmin = red[0];
        rmax = red[1];
        __syncthreads();
        if (rmax < rmin) { rmin = 0; rmax = -1; }
    }

    // Field-wise select instead of S.seg[s]: a dynamically indexed kernel-argument struct would be
    // copied to scratch; scalar selects keep the descriptors in SGPRs.
    __device__ __forceinline__ GemmSeg pick(int s) const {
        GemmSeg g;
#define REGT_PICK(f) g.f = s == 0 ? S.seg[0].f : (s == 1 ? S.seg[1].f : S.seg[2].f)
        REGT_PICK(A); REGT_PICK(lda); REGT_PICK(B0); REGT_PICK(B1); REGT_PICK(ldb); REGT_PICK(nsplit);
        REGT_PICK(K); REGT_PICK(flags); REGT_PICK(b_region_stride); REGT_PICK(a_rep_stride); REGT_PICK(nrep);
#undef REGT_PICK
        return g;
    }

    __device__ __forceinline__ int seg_iters(int s) const {
        const int K = s == 0 ? S.seg[0].K : (s == 1 ? S.seg[1].K : S.seg[2].K);
        const int fl = s == 0 ? S.seg[0].flags : (s == 1 ? S.seg[1].flags : S.seg[2].flags);
        const int nr = s == 0 ? S.seg[0].nrep : (s == 1 ? S.seg[1].nrep : S.seg[2].nrep);
        int nk = (K + GBK - 1) / GBK;
        int reps = (fl & SEG_REGION) ? (rmax - rmin + 1) : ((fl & SEG_REPEAT) ? nr : 1);
        return nk * reps;
    }

    __device__ __forceinline__ TileIter decode(int it) const {
        TileIter t{0, 0, 0};
        int s = 0;
        for (; s < S.nseg - 1; ++s) {
            int n = seg_iters(s);
            if (it < n) break;
            it -= n;
        }
        const int K = s == 0 ? S.seg[0].K : (s == 1 ? S.seg[1].K : S.seg[2].K);
        int nk = (K + GBK - 1) / GBK;
        const int fl = s == 0 ? S.seg[0].flags : (s == 1 ? S.seg[1].flags : S.seg[2].flags);
        t.seg = s;
        t.region = ((fl & SEG_REGION) ? rmin : 0) + it / nk;
        t.k0 = (it % nk) * GBK;
        return t;
    }

    __device__ __forceinline__ void load_regs(int it, float4 (&ra)[4], float4 (&rb)[4]) const {
        TileIter ti = decode(it);
        const GemmSeg g = pick(ti.seg);
        const bool region = (g.flags & SEG_REGION) != 0;
        const bool repeat = (g.flags & SEG_REPEAT) != 0;
        const float* Abase = g.A + (repeat ? (long)ti.region * g.a_rep_stride : 0);
        const bool vecA = (g.flags & SEG_VEC_A) != 0, vecB = (g.flags & SEG_VEC_B) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int slot = tid + 256 * i;
            int r = slot >> 3, k = ti.k0 + 4 * (slot & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rm.nvalid && k < g.K) {
                long gr = rm.grow(r);
                bool ok = true;
                if (region) ok = S.node_region[gr / S.row_div] == ti.region;
                if (ok) v = ld4_guard(Abase + gr * g.lda + k, g.K - k, vecA);
            }
            if (g.flags & SEG_RELU_A) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            ra[i] = v;
        }
        const long boff = (region || repeat) ? (long)ti.region * g.b_region_stride : 0;
        if (g.flags & SEG_BT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int slot = tid + 256 * i;
                int n = n0 + (slot >> 3), k = ti.k0 + 4 * (slot & 7);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < N && k < g.K) {
                    const float* p = (n < g.nsplit) ? g.B0 + (long)n * g.ldb : g.B1 + (long)(n - g.nsplit) * g.ldb;
                    v = ld4_guard(p + boff + k, g.K - k, vecB);
                }
                rb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int slot = tid + 256 * i;
                int k = ti.k0 + (slot >> 5), n = n0 + 4 * (slot & 31);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K && n < N) v = ld4_guard(g.B0 + boff + (long)k * g.ldb + n, N - n, vecB);
                rb[i] = v;
            }
        }
    }

    __device__ __forceinline__ void store_lds(int it, int stage, const float4 (&ra)[4], const float4 (&rb)[4]) const {
        float* la = lds + stage * G_STAGE;
        float* lb = la + G_A_TILE;
        const bool bt = (pick(decode(it).seg).flags & SEG_BT) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int slot = tid + 256 * i;
            *reinterpret_cast<float4*>(la + (slot >> 3) * G_LDS_ROW + 4 * (slot & 7)) = ra[i];
            if (bt)
                *reinterpret_cast<float4*>(lb + (slot >> 3) * G_LDS_ROW + 4 * (slot & 7)) = rb[i];
            else
                *reinterpret_cast<float4*>(lb + (slot >> 5) * G_LDS_KROW + 4 * (slot & 31)) = rb[i];
        }
    }

    __device__ __forceinline__ void compute(int it, int stage, f32x16 (&acc)[2][2]) const {
        const float* la = lds + stage * G_STAGE;
        const float* lb = la + G_A_TILE;
        const bool bt = (pick(decode(it).seg).flags & SEG_BT) != 0;
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi] = *reinterpret_cast<const float4*>(la + (wr * 64 + mi * 32 + lr) * G_LDS_ROW + kg * 8 + lh * 4);
            if (bt) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    b[ni] = *reinterpret_cast<const float4*>(lb + (wc * 64 + ni * 32 + lr) * G_LDS_ROW + kg * 8 + lh * 4);
            } else {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const float* q = lb + (kg * 8 + lh * 4) * G_LDS_KROW + wc * 64 + ni * 32 + lr;
                    b[ni] = make_float4(q[0], q[G_LDS_KROW], q[2 * G_LDS_KROW], q[3 * G_LDS_KROW]);
                }
            }
            const float* af0 = reinterpret_cast<const float*>(&a[0]);
            const float* af1 = reinterpret_cast<const float*>(&a[1]);
            const float* bf0 = reinterpret_cast<const float*>(&b[0]);
            const float* bf1 = reinterpret_cast<const float*>(&b[1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[j], bf0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[j], bf1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[j], bf0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[j], bf1[j], acc[1][1], 0, 0, 0);
            }
        }
    }

    // acc += sum over all segments.  Ends with a barrier (LDS free for reuse).
    __device__ __forceinline__ void run(f32x16 (&acc)[2][2]) const {
        int nit = 0;
        for (int s = 0; s < S.nseg; ++s) nit += seg_iters(s);
        if (nit == 0) return;
        float4 ra[4], rb[4];
        load_regs(0, ra, rb);
        store_lds(0, 0, ra, rb);
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            const bool more = it + 1 < nit;
            if (more) load_regs(it + 1, ra, rb);
            compute(it, it & 1, acc);
            if (more) store_lds(it + 1, (it + 1) & 1, ra, rb);
            __syncthreads();
        }
    }

    // Visit every valid accumulator element: f(local_row, global_col, value).
    template <class F>
    __device__ __forceinline__ void for_each(f32x16 (&acc)[2][2], F f) const {
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int r = wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (r < rm.nvalid) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        int c = n0 + wc * 64 + ni * 32 + lr;
                        if (c < N) f(r, c, acc[mi][ni][reg]);
                    }
                }
            }
    }

    // ---- LDS-staged epilogue: accumulators -> [128][132] fp32 tile in LDS -> each thread owns the float4
    // column block (tid & 31) of rows (tid >> 5) + 8*i, i = 0..15.  A wave instruction then touches two
    // whole 512-B row segments: 16-B vector, fully coalesced global accesses in the epilogue.
    __device__ __forceinline__ void stage(f32x16 (&acc)[2][2]) const {
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    lds[(wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * G_LDS_KROW + wc * 64 + ni * 32 + lr] =
                        acc[mi][ni][reg];
        __syncthreads();
    }
    __device__ __forceinline__ int erow(int i) const { return (tid >> 5) + 8 * i; }
    __device__ __forceinline__ int ecol() const { return n0 + 4 * (tid & 31); }
    __device__ __forceinline__ float4 eread(int i) const {
        return *reinterpret_cast<const float4*>(lds + erow(i) * G_LDS_KROW + 4 * (tid & 31));
    }

    // Vector epilogue driver: f.load(m, c) gathers the auxiliary operands of 4 rows first (loads in
    // flight together), then f.apply(m, c, v, aux) computes and stores.  m = global row, c = first of 4 columns.
    template <class F>
    __device__ __forceinline__ void for_each_vec(f32x16 (&acc)[2][2], const F& f) const {
        stage(acc);
        const int c = ecol();
        if (c < N) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename F::Aux aux[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = erow(4 * g + j);
                    if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = erow(4 * g + j);
                    if (r < rm.nvalid) f.apply(rm.grow(r), c, eread(4 * g + j), aux[j]);
                }
            }
        }
        __syncthreads();
    }

    // Same walk over two accumulators: acc2[..] = f(local_row, global_col, acc[..], acc2[..]).
    template <class F>
    __device__ __forceinline__ void for_each2(f32x16 (&acc)[2][2], f32x16 (&acc2)[2][2], F f) const {
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int r = wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (r < rm.nvalid) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        int c = n0 + wc * 64 + ni * 32 + lr;
                        if (c < N) acc2[mi][ni][reg] = f(r, c, acc[mi][ni][reg], acc2[mi][ni][reg]);
                    }
                }
            }
    }
};

// XCD-aware bijective remap of the linear block id: blocks b and b+8 share an XCD (and its L2),
// so give every XCD a contiguous chunk of tiles -- neighbouring tiles share A rows / B panels.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

// 4 bf16 (8 bytes) widened to fp32 bit patterns
__device__ __forceinline__ float4 widen_bf16x4(unsigned lo, unsigned hi) {
    return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                       __uint_as_float(hi & 0xffff0000u));
}
static inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace regt
