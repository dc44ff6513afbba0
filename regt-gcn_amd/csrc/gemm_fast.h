// Fast path of the segmented fp32-MFMA GEMM (same tile geometry and LDS images as gemm_core.h).
//
// Differences from the generic core, all aimed at keeping the matrix pipe of every SIMD fed:
//   * the K loop body is ONE straight-line basic block: no layout / guard / segment branches inside.
//     The (segment, region repeat, k0) walk is precomputed per workgroup into a small descriptor
//     table in LDS; operand layout (B stored [N][K] or [K][N]) and region masking are template
//     parameters; all operands are 16-byte vectorisable (checked on the host, else the generic
//     kernel runs);
//   * global loads of tile it+1 and the LDS fragment reads of k-group kg+1 are issued BETWEEN the
//     16-MFMA groups of k-group kg, so that a single wave can keep its SIMD's matrix pipe busy
//     (an fp32 32x32x2 MFMA occupies the pipe for 64 cycles: plenty of issue slots per gap);
//   * per-thread row offsets / validity are hoisted out of the loop.
#pragma once
#include <type_traits>
#include "gemm_core.h"

namespace regt {

constexpr int G_MAX_ITERS = 80;   // descriptor table capacity (e.g. K=512 in 3 segments with region repeats)

struct ItDesc {
    const float* A;
    const float* B0;
    const float* B1;
    long lda, ldb;
    int K, k0, region, nsplit;
    int abf, pad_;        // A operand stored as bf16 (SEG_A_BF16)
};
// the iteration table, then the epilogue's row table (EpiRowEnt per tile row, functors with HAS_ROWTAB)
constexpr int G_TABLE_BYTES = G_MAX_ITERS * (int)sizeof(ItDesc) + EPI_ROWTAB_BYTES;
constexpr int G_FAST_LDS_BYTES = G_LDS_BYTES + G_TABLE_BYTES;

// Sorted list of the DISTINCT regions among a tile's rows, built by thread 0 in LDS (red[8 ..): region of each row, then the
// list; red[3] = its length).  A region-masked segment is repeated once per listed region only, so the iteration count is
// bounded by the regions a tile can meet (its node count + 1 for node-major rows), not by the region count of the graph --
// 64 regions (the 8-GPU global graph on one GPU) stay on the fast path.  Ascending order = the order of the old
// [rmin, rmax] walk minus the regions that contributed only zeros, so sums are unchanged bit for bit.
template <int BMT>
__device__ __forceinline__ const int* tile_regions(const GemmSegs& S, const RowMap& rm, int* red, int tid) {
    // red[4..8): per-wave masks of the rows that START a run of equal regions; red[8 ..): region of each row, then the list.
    // Thread 0 only visits the run starts (one or two per tile when node ids are sorted by region) and inserts their
    // regions into a sorted, duplicate-free list.
    static_assert(BMT <= 128, "two waves of row flags");
    int* rowreg = red + 8;
    int* list = red + 8 + BMT;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(red + 4);
    const int nv = rm.nvalid < BMT ? rm.nvalid : BMT;
    int reg = -1;
    if (tid < nv) {
        reg = S.node_region[rm.grow(tid) / S.row_div];
        rowreg[tid] = reg;
    }
    __syncthreads();
    if (tid < 128) {
        const bool first = tid < nv && (tid == 0 || rowreg[tid - 1] != reg);
        const unsigned long long m = __ballot(first);
        if ((tid & 63) == 0) masks[tid >> 6] = m;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < 2; ++w) {
            unsigned long long m = masks[w];
            while (m) {
                const int r = 64 * w + __builtin_ctzll(m);
                m &= m - 1;
                const int rg = rowreg[r];
                int pos = 0;
                while (pos < n && list[pos] < rg) ++pos;
                if (pos < n && list[pos] == rg) continue;
                for (int j = n; j > pos; --j) list[j] = list[j - 1];
                list[pos] = rg;
                ++n;
            }
        }
        red[3] = n;
    }
    return list;        // complete after the caller's next barrier
}

// Build the iteration table of a tile, one entry per thread: a single thread walking the (segment, repeat, k0) nest took
// ~6 us per workgroup next to two MFMA-bound waves on its SIMD (tools/wg_trace.py) -- a quarter of the K loop's duration.
// Every thread walks the <= 3 segments (wave-uniform arithmetic) and the thread whose id falls into a segment's range of
// slabs writes that slab's descriptor.  Returns the table length; ends with a barrier.
template <int BMT, bool REGION>
__device__ __forceinline__ int plan_table(const GemmSegs& S, const RowMap& rm, float* lds, ItDesc* table, int tid) {
    int* red = reinterpret_cast<int*>(lds);
    const int* rlist = nullptr;
    int nreg = 0;
    if (REGION) {
        rlist = tile_regions<BMT>(S, rm, red, tid);
        __syncthreads();
        nreg = red[3];
    }
    int n = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < S.nseg) {
            const GemmSeg& g = S.seg[s];
            const bool reg = REGION && (g.flags & SEG_REGION);
            const bool rep = (g.flags & SEG_REPEAT) != 0;
            const int cnt = reg ? nreg : (rep ? g.nrep : 1);
            const int nk = (g.K + GBK - 1) / GBK;
            const int local = tid - n;
            if (local >= 0 && local < cnt * nk && tid < G_MAX_ITERS) {
                const int ri = local / nk, r = reg ? rlist[ri] : ri;
                const long off = (reg || rep) ? (long)r * g.b_region_stride : 0;
                ItDesc d;
                d.A = g.A + (rep ? (long)r * g.a_rep_stride : 0);
                d.B0 = g.B0 + off; d.B1 = g.B1 + off; d.lda = g.lda; d.ldb = g.ldb;
                d.K = g.K; d.k0 = (local - ri * nk) * GBK; d.region = reg ? r : -1; d.nsplit = g.nsplit;
                d.abf = (g.flags & SEG_A_BF16) ? 1 : 0; d.pad_ = 0;
                table[tid] = d;
            }
            n += cnt * nk;
        }
    }
    __syncthreads();
    return n < G_MAX_ITERS ? n : G_MAX_ITERS;
}

template <bool BT, bool REGION>
struct FastCore {
    // tile geometry seen by kernels with their own epilogue (candidate kernel)
    static constexpr int BM = GBM, BN = GBN, EROWS = 16, ETPR = 32, EKROW = G_LDS_KROW;
    typedef f32x16 Acc[2][2];
    __device__ __forceinline__ static void zero(Acc& acc) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const GemmSegs& S;
    RowMap rm;
    int n0, N;
    float* lds;
    ItDesc* table;
    int nit;
    int tid, lane, wr, wc;
    long arow[4];      // global row of the thread's 4 A slots (or -1)
    int areg[4];       // region of that row (REGION only)

    __device__ __forceinline__ FastCore(const GemmSegs& s, RowMap r, int n0_, int N_, float* lds_)
        : S(s), rm(r), n0(n0_), N(N_), lds(lds_) {
        tid = threadIdx.x;
        lane = tid & 63;
        const int wid = tid >> 6;
        wr = wid >> 1;
        wc = wid & 1;
        table = reinterpret_cast<ItDesc*>(lds + 2 * G_STAGE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = (tid + 256 * i) >> 3;
            arow[i] = rr < rm.nvalid ? rm.grow(rr) : -1;
            areg[i] = 0;
            if (REGION && arow[i] >= 0) areg[i] = S.node_region[arow[i] / S.row_div];
        }
    }

    // Re-target the row map (candidate kernel: same tile, next period) without re-planning.
    __device__ __forceinline__ void set_rows(RowMap r) {
        rm = r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = (tid + 256 * i) >> 3;
            arow[i] = rr < rm.nvalid ? rm.grow(rr) : -1;
        }
    }

    // Build the iteration table (plan_table above).  Ends with a barrier.
    __device__ __forceinline__ void plan() { nit = plan_table<GBM, REGION>(S, rm, lds, table, tid); }

    // Guarded loads as raw buffer loads: the per-tile base goes into a wave-uniform buffer descriptor
    // (SGPRs) and every lane supplies a 32-bit byte offset; a masked-out slot gets an offset beyond
    // num_records, for which the hardware returns 0 without touching memory.  No branch, no flat
    // address (a flat load would also count on lgkmcnt and serialise with the LDS fragment reads).
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    static constexpr unsigned SRD_RANGE = 0x7FFFFFF0u, SRD_OOB = 0x7FFFFFF8u;
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t make_srd(const float* p) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)SRD_RANGE, 0x00020000);
    }
    __device__ __forceinline__ static float4 srd_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    struct Srds { __amdgpu_buffer_rsrc_t a, b; int lda, ldb, K, k0, region, abf; };
    __device__ __forceinline__ Srds make_srds(const ItDesc& d) const {
        Srds r;
        r.abf = __builtin_amdgcn_readfirstlane(d.abf);
        r.lda = __builtin_amdgcn_readfirstlane((int)d.lda);
        r.ldb = __builtin_amdgcn_readfirstlane((int)d.ldb);
        r.K = __builtin_amdgcn_readfirstlane(d.K);
        r.k0 = __builtin_amdgcn_readfirstlane(d.k0);
        r.region = __builtin_amdgcn_readfirstlane(d.region);
        // byte address of the tile's first A row: bf16 rows are half as long
        r.a = make_srd(r.abf ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(d.A) + 2 * rm.base * d.lda)
                             : d.A + rm.base * d.lda);
        const int ns = __builtin_amdgcn_readfirstlane(d.nsplit);
        if (BT) r.b = make_srd(n0 < ns ? d.B0 + (long)n0 * d.ldb : d.B1 + (long)(n0 - ns) * d.ldb);
        else r.b = make_srd(d.B0 + n0);
        return r;
    }
    __device__ __forceinline__ float4 load_a(const Srds& d, int i) const {
        const int slot = tid + 256 * i;
        const int k = d.k0 + 4 * (slot & 7);
        bool ok = (slot >> 3) < rm.nvalid && k < d.K;
        if (REGION) ok = ok && (d.region < 0 || areg[i] == d.region);
        const unsigned off = ok ? 4u * (unsigned)((slot >> 3) * (int)rm.mul * d.lda + k) : SRD_OOB;
        return srd_load(d.a, off);
    }
    __device__ __forceinline__ float4 load_b(const Srds& d, int i) const {
        const int slot = tid + 256 * i;
        if (BT) {
            const int nl = slot >> 3, k = d.k0 + 4 * (slot & 7);
            const bool ok = n0 + nl < N && k < d.K;
            return srd_load(d.b, ok ? 4u * (unsigned)(nl * d.ldb + k) : SRD_OOB);
        } else {
            const int k = d.k0 + (slot >> 5), nl = 4 * (slot & 31);
            const bool ok = k < d.K && n0 + nl < N;
            return srd_load(d.b, ok ? 4u * (unsigned)(k * d.ldb + nl) : SRD_OOB);
        }
    }
    __device__ __forceinline__ void store_a(float* st, int i, float4 v, bool relu) const {
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const int slot = tid + 256 * i;
        *reinterpret_cast<float4*>(st + (slot >> 3) * G_LDS_ROW + 4 * (slot & 7)) = v;
    }
    __device__ __forceinline__ void store_b(float* st, int i, float4 v) const {
        const int slot = tid + 256 * i;
        float* lb = st + G_A_TILE;
        if (BT) *reinterpret_cast<float4*>(lb + (slot >> 3) * G_LDS_ROW + 4 * (slot & 7)) = v;
        else *reinterpret_cast<float4*>(lb + (slot >> 5) * G_LDS_KROW + 4 * (slot & 31)) = v;
    }
    struct Frag { float4 a[2], b[2]; };
    __device__ __forceinline__ Frag read_frag(const float* st, int kg) const {
        const int lr = lane & 31, lh = lane >> 5;
        Frag f;
        const float* lb = st + G_A_TILE;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            f.a[mi] = *reinterpret_cast<const float4*>(st + (wr * 64 + mi * 32 + lr) * G_LDS_ROW + kg * 8 + lh * 4);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if (BT) {
                f.b[ni] = *reinterpret_cast<const float4*>(lb + (wc * 64 + ni * 32 + lr) * G_LDS_ROW + kg * 8 + lh * 4);
            } else {
                const float* q = lb + (kg * 8 + lh * 4) * G_LDS_KROW + wc * 64 + ni * 32 + lr;
                f.b[ni] = make_float4(q[0], q[G_LDS_KROW], q[2 * G_LDS_KROW], q[3 * G_LDS_KROW]);
            }
        }
        return f;
    }
    __device__ __forceinline__ void mfma16(const Frag& f, f32x16 (&acc)[2][2]) const {
        const float* a0 = reinterpret_cast<const float*>(&f.a[0]);
        const float* a1 = reinterpret_cast<const float*>(&f.a[1]);
        const float* b0 = reinterpret_cast<const float*>(&f.b[0]);
        const float* b1 = reinterpret_cast<const float*>(&f.b[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }

    // acc += sum over the planned iterations.  `relu_a`: apply relu to A while staging (head).
    __device__ __forceinline__ void run(f32x16 (&acc)[2][2], bool relu_a) const {
        if (nit == 0) return;
        float4 ra[4], rb[4];
        {
            const Srds d = make_srds(table[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra[i] = load_a(d, i); rb[i] = load_b(d, i); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { store_a(lds, i, ra[i], relu_a); store_b(lds, i, rb[i]); }
        }
        __syncthreads();
        Frag cur = read_frag(lds, 0);
        for (int it = 0; it + 1 < nit; ++it) {
            const float* st = lds + (it & 1) * G_STAGE;
            float* nx = lds + ((it + 1) & 1) * G_STAGE;
            const Srds d = make_srds(table[it + 1]);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                Frag nxt;
                if (kg < 3) nxt = read_frag(st, kg + 1);
                ra[kg] = load_a(d, kg);
                rb[kg] = load_b(d, kg);
                // Keep this group's global loads / LDS reads ahead of its MFMAs (hipcc otherwise sinks the
                // loads to their use at the end of the iteration and exposes the HBM latency) and spread
                // them over the first MFMA gaps: 1 MFMA, then one memory op and a few VALU, repeated.
#pragma unroll
                for (int r = 0; r < 10; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);   // VALU | SALU
                }
                mfma16(cur, acc);
                if (kg < 3) cur = nxt;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { store_a(nx, i, ra[i], relu_a); store_b(nx, i, rb[i]); }
            __syncthreads();
            cur = read_frag(nx, 0);
        }
        {
            const float* st = lds + ((nit - 1) & 1) * G_STAGE;
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                Frag nxt;
                if (kg < 3) nxt = read_frag(st, kg + 1);
                mfma16(cur, acc);
                if (kg < 3) cur = nxt;
            }
        }
        __syncthreads();
    }

    // ---- LDS-staged vector epilogue (same as GemmCore) ----------------------------------------------
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
    // Epilogue driver.  The auxiliary operands (h, Z, R, ... rows of this tile) come from HBM: every round of loads exposes
    // one memory latency (2-4k cycles under load) to a wave that has nothing else to do, so the rows are handled in as few
    // rounds as the functor's operand count allows (F::ROUND_ROWS of the thread's 16 rows per round: 16 for one float4 per
    // row, 8 for two to four), and the first round is requested BEFORE the accumulators are staged through LDS, so that
    // its latency overlaps the 64 LDS writes and the barrier.
    // Full tile + a functor variant: the straight-line body (no row / column guards, no functor branches; see the note on
    // the functors in gemm.hip).  Rounds of RR rows; when two rounds of auxiliary operands fit the registers the
    // accumulators free up once they are staged, round g + 1 is requested BEFORE round g is applied, so that its loads
    // are ahead of round g's stores in the (in-order) vmcnt queue and never wait for a store to complete.
    __device__ __forceinline__ EpiRowEnt* rowtab() const { return reinterpret_cast<EpiRowEnt*>(table + G_MAX_ITERS); }
    // fill the row table of a functor that wants one (before a barrier that precedes the epilogue, e.g. plan()'s)
    template <class F>
    __device__ __forceinline__ void fill_rowtab(const F& f) const {
        if constexpr (F::HAS_ROWTAB) {
            if (tid < GBM) rowtab()[tid] = f.vrow(rm.base + (tid < rm.nvalid ? tid : 0));
        }
    }
    template <class F, int V>
    __device__ __forceinline__ void vec_body(f32x16 (&acc)[2][2], const F& f) const {
        // one round if the operands of the thread's 16 rows fit 128 registers, else double-buffered rounds of <= 72 each
        constexpr int AB = (int)sizeof(typename F::VAux);
        constexpr int RR = AB * 16 <= 512 ? 16 : (AB * 8 <= 288 ? 8 : (AB * 4 <= 288 ? 4 : 2)), NR = 16 / RR;
        constexpr bool DB = NR > 1;
        const EpiGeom geo{rm.base, n0, tid >> 5, 4 * (tid & 31), 8, rowtab()};
        const typename F::Tile tl = f.template vtile<V>(geo);
        const typename F::Col col = f.template vcol<V>(ecol());
        typename F::VAux aux[DB ? 2 : 1][RR];
#pragma unroll
        for (int j = 0; j < RR; ++j) aux[0][j] = f.template vload<V>(tl, j);
        stage(acc);
#pragma unroll
        for (int g = 0; g < NR; ++g) {
            if (DB && g + 1 < NR) {
#pragma unroll
                for (int j = 0; j < RR; ++j) aux[(g + 1) & 1][j] = f.template vload<V>(tl, RR * (g + 1) + j);
            }
#pragma unroll
            for (int j = 0; j < RR; ++j) f.template vapply<V>(tl, RR * g + j, eread(RR * g + j), col, aux[DB ? (g & 1) : 0][j]);
        }
        __syncthreads();
    }
    template <class F, int V, class Body>
    __device__ __forceinline__ static void dispatch_variant(int v, const Body& body) {
        if constexpr (V < F::NVAR) {
            if (v == V) body(std::integral_constant<int, V>{});
            else dispatch_variant<F, V + 1>(v, body);
        }
    }
    // true if the tile is full and the functor has a variant for it (rows m = base + r)
    template <class F>
    __device__ __forceinline__ int tile_variant(const F& f) const {
#ifdef REGT_EPI_GENERIC      // developer switch: every tile through the guarded load()/apply() path (A/B checks of the variants)
        return -1;
#endif
        const int v = __builtin_amdgcn_readfirstlane(f.variant(n0));
        return (rm.nvalid == GBM && rm.mul == 1 && n0 + GBN <= N) ? v : -1;
    }
    template <class F>
    __device__ __forceinline__ void for_each_vec(f32x16 (&acc)[2][2], const F& f) const {
        constexpr int RR = F::ROUND_ROWS;
        static_assert(RR == 4 || RR == 8 || RR == 16, "rows per epilogue round");
        const int v = tile_variant(f);
        if (v >= 0) {
            dispatch_variant<F, 0>(v, [&](auto tag) { vec_body<F, decltype(tag)::value>(acc, f); });
            return;
        }
        const int c = ecol();
        typename F::Aux aux[RR];
        if (c < N) {
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                const int r = erow(j);
                if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
            }
        }
        stage(acc);
        if (c < N) {
#pragma unroll
            for (int g = 0; g < 16 / RR; ++g) {
                if (g > 0) {
#pragma unroll
                    for (int j = 0; j < RR; ++j) {
                        const int r = erow(RR * g + j);
                        if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
                    }
                }
#pragma unroll
                for (int j = 0; j < RR; ++j) {
                    const int r = erow(RR * g + j);
                    if (r < rm.nvalid) f.apply(rm.grow(r), c, eread(RR * g + j), aux[j]);
                }
            }
        }
        __syncthreads();
    }
};

}  // namespace regt
