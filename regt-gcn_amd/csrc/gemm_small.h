// 64 x 64 tile variant of the segmented fp32-MFMA GEMM for SMALL problems (the reference's own graph: TPIMS, 104 nodes,
// M = N*T ~ 10^3 rows).  With 128 x 128 tiles such a GEMM is 10-40 workgroups on a 256-CU chip and its duration is the
// latency of ONE tile (K/32 slabs x 64 MFMAs x 64 cycles per wave: 30-50 us at K = 256-512).  Quartering the tile gives
// 4x the workgroups and a quarter of the MFMA chain per wave; nothing else changes: same segments, region masking,
// iteration table, buffer-descriptor loads (FastCore's statics), LDS row padding and epilogue functors.
//
// 256 threads = 4 waves 2 x 2, each wave one 32 x 32 tile as 2 x 2 fragments of v_mfma_f32_16x16x4_f32.  No scheduling
// tricks: operands of problems this small come from L2 and there are four co-resident workgroups per CU to cover each other.
#pragma once
#include "gemm_fast.h"

namespace regt {

constexpr int SM_B = 64;                              // tile edge
constexpr int SM_KROW = SM_B + 4;                     // 68 floats: k-major B rows / epilogue image rows
constexpr int SM_TILE = SM_B * G_LDS_ROW;             // 2304 floats (>= 32 * 68)
constexpr int SM_STAGE = 2 * SM_TILE;                 // A tile + B tile
constexpr int SM_LDS_BYTES = 2 * SM_STAGE * 4 + G_TABLE_BYTES;

template <bool BT, bool REGION>
struct SmallCore {
    using F = FastCore<BT, REGION>;                   // statics only: make_srd, srd_load, SRD_OOB
    static constexpr int BM = SM_B, BN = SM_B, EROWS = 4, ETPR = 16, EKROW = SM_KROW;
    typedef f32x16 Acc;
    __device__ __forceinline__ static void zero(Acc& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    const GemmSegs& S;
    RowMap rm;
    int n0, N;
    float* lds;
    ItDesc* table;
    int nit;
    int tid, lane, wr, wc;
    int areg[2];                                      // region of the thread's two A staging rows (REGION only)

    __device__ __forceinline__ SmallCore(const GemmSegs& s, RowMap r, int n0_, int N_, float* lds_)
        : S(s), rm(r), n0(n0_), N(N_), lds(lds_) {
        tid = threadIdx.x;
        lane = tid & 63;
        const int wid = tid >> 6;
        wr = wid >> 1;
        wc = wid & 1;
        table = reinterpret_cast<ItDesc*>(lds + 2 * SM_STAGE);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = (tid + 256 * i) >> 3;
            areg[i] = 0;
            if (REGION && rr < rm.nvalid) areg[i] = S.node_region[rm.grow(rr) / S.row_div];
        }
    }

    // iteration table: as FastCore::plan
    __device__ __forceinline__ void plan() { nit = plan_table<SM_B, REGION>(S, rm, lds, table, tid); }

    struct Srds { __amdgpu_buffer_rsrc_t a, b; int lda, ldb, K, k0, region; };
    __device__ __forceinline__ Srds make_srds(const ItDesc& d) const {
        Srds r;
        r.lda = __builtin_amdgcn_readfirstlane((int)d.lda);
        r.ldb = __builtin_amdgcn_readfirstlane((int)d.ldb);
        r.K = __builtin_amdgcn_readfirstlane(d.K);
        r.k0 = __builtin_amdgcn_readfirstlane(d.k0);
        r.region = __builtin_amdgcn_readfirstlane(d.region);
        r.a = F::make_srd(d.A + rm.base * d.lda);
        const int ns = __builtin_amdgcn_readfirstlane(d.nsplit);
        if (BT) r.b = F::make_srd(n0 < ns ? d.B0 + (long)n0 * d.ldb : d.B1 + (long)(n0 - ns) * d.ldb);
        else r.b = F::make_srd(d.B0 + n0);
        return r;
    }
    __device__ __forceinline__ float4 load_a(const Srds& d, int i) const {
        const int slot = tid + 256 * i;
        const int k = d.k0 + 4 * (slot & 7);
        bool ok = (slot >> 3) < rm.nvalid && k < d.K;
        if (REGION) ok = ok && (d.region < 0 || areg[i] == d.region);
        return F::srd_load(d.a, ok ? 4u * (unsigned)((slot >> 3) * (int)rm.mul * d.lda + k) : F::SRD_OOB);
    }
    __device__ __forceinline__ float4 load_b(const Srds& d, int i) const {
        const int slot = tid + 256 * i;
        if (BT) {
            const int nl = slot >> 3, k = d.k0 + 4 * (slot & 7);
            const bool ok = n0 + nl < N && k < d.K;
            return F::srd_load(d.b, ok ? 4u * (unsigned)(nl * d.ldb + k) : F::SRD_OOB);
        } else {
            const int k = d.k0 + (slot >> 4), nl = 4 * (slot & 15);
            const bool ok = k < d.K && n0 + nl < N;
            return F::srd_load(d.b, ok ? 4u * (unsigned)(k * d.ldb + nl) : F::SRD_OOB);
        }
    }
    __device__ __forceinline__ void store(float* st, int i, float4 a, float4 b, bool relu) const {
        if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        const int slot = tid + 256 * i;
        *reinterpret_cast<float4*>(st + (slot >> 3) * G_LDS_ROW + 4 * (slot & 7)) = a;
        float* lb = st + SM_TILE;
        if (BT) *reinterpret_cast<float4*>(lb + (slot >> 3) * G_LDS_ROW + 4 * (slot & 7)) = b;
        else *reinterpret_cast<float4*>(lb + (slot >> 4) * SM_KROW + 4 * (slot & 15)) = b;
    }
    // The wave's 32 x 32 tile as 2 x 2 fragments of v_mfma_f32_16x16x4_f32: acc[(mi * 2 + ni) * 4 + r] = C[16 mi + 4 (lane / 16) + r]
    // [16 ni + lane % 16].  This shape keeps the full fp32 matrix rate at four waves per SIMD (the occupancy of this kernel:
    // four workgroups per CU), where 32x32x2 drops to 0.64 of it (tools/micro/mfma_prio.hip).
    __device__ __forceinline__ void mfma_slab(const float* st, f32x16& acc) const {
        const int lr = lane & 15, lq = lane >> 4;
        const float* lb = st + SM_TILE;
        f32x4 c[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) c[mi][ni][r] = acc[(mi * 2 + ni) * 4 + r];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {          // 16 k per group: lane quarter lq supplies k = 16 kg + 4 lq + j to MFMA j
            float4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi] = *reinterpret_cast<const float4*>(st + (wr * 32 + mi * 16 + lr) * G_LDS_ROW + kg * 16 + lq * 4);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (BT) {
                    b[ni] = *reinterpret_cast<const float4*>(lb + (wc * 32 + ni * 16 + lr) * G_LDS_ROW + kg * 16 + lq * 4);
                } else {
                    const float* q = lb + (kg * 16 + lq * 4) * SM_KROW + wc * 32 + ni * 16 + lr;
                    b[ni] = make_float4(q[0], q[SM_KROW], q[2 * SM_KROW], q[3 * SM_KROW]);
                }
            }
            const float* af0 = reinterpret_cast<const float*>(&a[0]);
            const float* af1 = reinterpret_cast<const float*>(&a[1]);
            const float* bf0 = reinterpret_cast<const float*>(&b[0]);
            const float* bf1 = reinterpret_cast<const float*>(&b[1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0[j], bf0[j], c[0][0], 0, 0, 0);
                c[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0[j], bf1[j], c[0][1], 0, 0, 0);
                c[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af1[j], bf0[j], c[1][0], 0, 0, 0);
                c[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af1[j], bf1[j], c[1][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[(mi * 2 + ni) * 4 + r] = c[mi][ni][r];
    }

    __device__ __forceinline__ void run(f32x16& acc, bool relu_a) const {
        if (nit == 0) return;
        float4 ra[2], rb[2];
        {
            const Srds d = make_srds(table[0]);
#pragma unroll
            for (int i = 0; i < 2; ++i) { ra[i] = load_a(d, i); rb[i] = load_b(d, i); }
#pragma unroll
            for (int i = 0; i < 2; ++i) store(lds, i, ra[i], rb[i], relu_a);
        }
        __syncthreads();
        for (int it = 0; it + 1 < nit; ++it) {
            const float* st = lds + (it & 1) * SM_STAGE;
            float* nx = lds + ((it + 1) & 1) * SM_STAGE;
            const Srds d = make_srds(table[it + 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i) { ra[i] = load_a(d, i); rb[i] = load_b(d, i); }
            mfma_slab(st, acc);
#pragma unroll
            for (int i = 0; i < 2; ++i) store(nx, i, ra[i], rb[i], relu_a);
            __syncthreads();
        }
        mfma_slab(lds + ((nit - 1) & 1) * SM_STAGE, acc);
        __syncthreads();
    }

    // LDS-staged vector epilogue: 16 threads per 64-column row, 16 rows per pass, 4 passes
    __device__ __forceinline__ void stage(const f32x16& acc) const {
        const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lds[(wr * 32 + mi * 16 + 4 * lq + r) * SM_KROW + wc * 32 + ni * 16 + lr] = acc[(mi * 2 + ni) * 4 + r];
        __syncthreads();
    }
    __device__ __forceinline__ int erow(int i) const { return (tid >> 4) + 16 * i; }
    __device__ __forceinline__ int ecol() const { return n0 + 4 * (tid & 15); }
    __device__ __forceinline__ float4 eread(int i) const {
        return *reinterpret_cast<const float4*>(lds + erow(i) * SM_KROW + 4 * (tid & 15));
    }
    template <class Fn>
    __device__ __forceinline__ void for_each_vec(const f32x16& acc, const Fn& f) const {
        stage(acc);
        const int c = ecol();
        if (c < N) {
            typename Fn::Aux aux[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = erow(j);
                if (r < rm.nvalid) aux[j] = f.load(rm.grow(r), c);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = erow(j);
                if (r < rm.nvalid) f.apply(rm.grow(r), c, eread(j), aux[j]);
            }
        }
        __syncthreads();
    }
};

}  // namespace regt
