// Weight gradients out[Nout x Nin] = P^T Q of the RegT-GCN cell (the transposes of models/utils.py:168-188 and of the regional embedding,
// models/RegionalTemporalGCN.py:136-148) and the reduction of their row-chunk slabs: the fp32-MFMA kernels, the bf16-split kernels
// and the register-ring kernel for operands stored as bf16 rows, with the row chunking every launch uses.
// (Split out of gemm.hip in round 5; the forward / data-gradient GEMMs, the candidate kernels and the small GEMMs stay there.)
#include "kernels.h"
#include "gemm_fast.h"
#include "gemm_split.h"
#include "gemm_small.h"

namespace regt {

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

}  // namespace regt
