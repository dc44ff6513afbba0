// C ABI (include/regtgcn.h) and the forward / backward pipeline of the RegT-GCN hot path.
//
// Formulation (validated against the oracle in tests/test_fused_math.py):
// every sparse operator of the reference acts on the *input* x, so
//   * A_hat x and L~ x are aggregated once per snapshot at width T*F (one stacked SpMM) and are
//     constants w.r.t. the parameters -- the backward pass has no sparse op;
//   * every weight that multiplies a width-F quantity is folded into a (C,F) "composed" weight
//     (A0 = (sum_r Wl_r) W0, A_r = Wl_r W1, G_k = U_k[:, :C] V_k), so the only K=C contractions
//     left are the three hidden-state GEMMs of the GRU cell.
// Rows of all (M = N*T, .) activations are ordered node-major: m = node*T + t.
#include <limits.h>
#include <stdarg.h>

#include <functional>
#include <initializer_list>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/regtgcn.h"
#include <algorithm>
#include "kernels.h"

namespace regt {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- optional per-kernel timing with HIP events (bench.py roofline) ---------------------------------
// When enabled, every pipeline stage is bracketed by two events recorded on the launch stream.
struct ProfRec { const char* name; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;

struct ProfScope {
    hipStream_t st; bool on; ProfRec r;
    ProfScope(const char* name, hipStream_t s) : st(s), on(g_prof_on) {
        if (!on) return;
        r.name = name;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.e0, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.e1, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};
#define PROF(name, st) ProfScope _prof_scope_(name, st)

// ---- hipGraph replay for launch-bound problem sizes -------------------------------------------------------
// A forward or backward of a small graph (TPIMS: 104 nodes) is ~40-75 kernel launches of a few microseconds
// each.  Optionally (REGT_HIPGRAPH=1) the launch sequence of such sizes is captured
// into a hipGraph the SECOND time the same set of buffers is seen (pointers are the cache key: a graph is only
// ever replayed onto exactly the buffers it was captured with) and replayed from then on.
// Capture cannot run on the legacy default stream PyTorch uses, so graphs are captured and replayed on a
// library-owned stream that is ordered against the caller's stream with two events.
struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; };
struct GraphCache {
    std::unordered_map<unsigned long long, GraphEntry> map;
    std::mutex mu;
    long eager = 0, captured = 0, replayed = 0;
};
static GraphCache g_fwd_graphs, g_bwd_graphs;
static hipStream_t g_graph_stream = nullptr;
static hipEvent_t g_ev_in = nullptr, g_ev_out = nullptr;
static int g_graph_mode = -1;          // REGT_HIPGRAPH: 0 (default) off, 1 small problems only, 2 always
static const long GRAPH_MAX_ROWS = 1L << 15;   // N*T rows below which a step is launch-bound

static unsigned long long hash_bytes(const void* p, size_t n, unsigned long long h) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}

static bool graphs_wanted(long rows) {
    if (g_graph_mode < 0) {
        const char* e = getenv("REGT_HIPGRAPH");
        g_graph_mode = e ? atoi(e) : 0;     // opt-in: measured 1.05 vs 0.89 ms/step at TPIMS size (the step is bound
                                            // by the latency of many tiny dependent kernels, not by their launches)
    }
    if (g_prof_on || g_graph_mode == 0) return false;
    return g_graph_mode == 2 || rows <= GRAPH_MAX_ROWS;
}

// Runs `enqueue(stream)` either eagerly on `st` or as a cached graph replay ordered after / before `st`.
static std::mutex g_graph_stream_mu;   // g_graph_stream / g_ev_in / g_ev_out are shared by the forward and backward caches
static int run_maybe_graphed(GraphCache& cache, unsigned long long key, hipStream_t st,
                             const std::function<int(hipStream_t)>& enqueue) {
    std::lock_guard<std::mutex> lk_stream(g_graph_stream_mu);
    std::lock_guard<std::mutex> lk(cache.mu);
    const int gm = gemm_mode();            // a captured launch sequence is only valid for the arithmetic it was captured with
    key = hash_bytes(&gm, sizeof(gm), key);
    GraphEntry& e = cache.map[key];
    if (!e.exec) {
        if (e.seen++ == 0 || cache.map.size() > 256) {   // first sighting (also sets kernel attributes), or buffers
            ++cache.eager;                               // keep changing: plain launches
            return enqueue(st);
        }
        if (!g_graph_stream) {
            REGT_CHECK_HIP(hipStreamCreateWithFlags(&g_graph_stream, hipStreamNonBlocking));
            REGT_CHECK_HIP(hipEventCreateWithFlags(&g_ev_in, hipEventDisableTiming));
            REGT_CHECK_HIP(hipEventCreateWithFlags(&g_ev_out, hipEventDisableTiming));
        }
        hipGraph_t graph = nullptr;
        REGT_CHECK_HIP(hipStreamBeginCapture(g_graph_stream, hipStreamCaptureModeRelaxed));
        const int rc = enqueue(g_graph_stream);
        const hipError_t ce = hipStreamEndCapture(g_graph_stream, &graph);
        if (rc != REGT_OK || ce != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            e.seen = -(1 << 30);                         // never try again for this key
            return rc != REGT_OK ? rc : enqueue(st);
        }
        const hipError_t ie = hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) { e.exec = nullptr; e.seen = -(1 << 30); (void)hipGetLastError(); return enqueue(st); }
        ++cache.captured;
    }
    ++cache.replayed;
    REGT_CHECK_HIP(hipEventRecord(g_ev_in, st));
    REGT_CHECK_HIP(hipStreamWaitEvent(g_graph_stream, g_ev_in, 0));
    REGT_CHECK_HIP(hipGraphLaunch(e.exec, g_graph_stream));
    REGT_CHECK_HIP(hipEventRecord(g_ev_out, g_graph_stream));
    REGT_CHECK_HIP(hipStreamWaitEvent(st, g_ev_out, 0));
    return REGT_OK;
}

namespace {

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// REGT_GEMM_MODE=bf16: the intermediates that only ever feed matrix-core operands -- q = h*R (forward -> backward), dhp and
// dzp|drp (inside the backward) -- are STORED as bf16: their producer rounds them once instead of every consumer rounding
// them while staging, which is the same arithmetic at half the HBM bytes for these buffers (7.5 of the step's ~37 row-units).
// Needs the vector kernels everywhere (C a multiple of the 128-column tile, so that no GEMM falls back to the generic core).
bool bf16_intermediates(const regt_dims& d) { return gemm_mode() == 2 && d.C % GBN == 0 && d.F % 4 == 0; }
// q crosses from regt_forward to regt_backward: remember per workspace how the forward stored it, so that a mode change in
// between is caught instead of misread.
std::mutex g_qfmt_mu;
std::unordered_map<const void*, int> g_qfmt;
void note_q_format(const void* ws, int bf16) {
    std::lock_guard<std::mutex> lk(g_qfmt_mu);
    if (g_qfmt.size() > 4096) g_qfmt.clear();
    g_qfmt[ws] = bf16;
}
int q_format(const void* ws) {
    std::lock_guard<std::mutex> lk(g_qfmt_mu);
    auto it = g_qfmt.find(ws);
    return it == g_qfmt.end() ? 0 : it->second;
}
// ... and, where the fused forward kernel applies (fused.hip: C = 256, F = 64, node-disjoint regions), x, A_hat x and L~ x as
// well: the snapshot is rounded once while it is packed, the aggregation reads and writes bf16 rows (cfg-5).
// REGT_XBF=0 keeps them fp32 and the three-launch forward (A/B timing; tests/test_gpu_fused.py compares the two bit for bit).
// per-call switches (regt_dims.flags) of the entry point running on this thread; the process-wide options are the defaults
thread_local unsigned t_call_flags = 0;
int g_opt_fused_bwd = -1;
bool fused_bwd_wanted() {
    if (t_call_flags & REGT_DIMS_NO_FUSED_BWD) return false;
    if (g_opt_fused_bwd < 0) { const char* e = getenv("REGT_FUSED_BWD"); g_opt_fused_bwd = e ? atoi(e) : 1; }
    return g_opt_fused_bwd != 0;
}
int g_opt_xbf = -1;
bool xbf_wanted() {
    if (t_call_flags & REGT_DIMS_NO_BF16_ROWS) return false;
    if (g_opt_xbf < 0) { const char* e = getenv("REGT_XBF"); g_opt_xbf = e ? atoi(e) : 1; }
    return g_opt_xbf != 0;
}
// regt_dims.arith / .flags hold for the duration of one entry point on the calling thread
struct CallScope {
    int prev_mode;
    unsigned prev_flags;
    explicit CallScope(const regt_dims* d) {
        prev_flags = t_call_flags;
        t_call_flags = d ? d->flags : 0;
        const int a = d ? d->arith : 0;
        prev_mode = gemm_mode_override(a >= REGT_ARITH_FP32 && a <= REGT_ARITH_BF16 ? a - 1 : -1);
    }
    ~CallScope() { gemm_mode_override(prev_mode); t_call_flags = prev_flags; }
};
// workspace formats remembered between forward and backward (note_q_format): bit 0 = bf16 intermediates, bit 1 = bf16 rows of
// x / A_hat x / L~ x, bit 2 = the packed input was the CALLER's bf16 buffer (else the rounded copy lives in the workspace)
enum : int { FMT_QBF = 1, FMT_XBF = 2, FMT_XCALLER = 4, FMT_TCOLLAPSE = 8 };
// TemporalGCN / A3T-GCN (regional = 0): the hidden input h = x W0^T + (L~ x) W1^T + b has NO activation (models/TemporalGCN.py:88 --
// RegT-GCN applies leaky_relu, RegionalTemporalGCN.py:143), so wherever the gates use h LINEARLY it folds into the input:
//     h [Uz2; Ur2]^T = x ([Uz2; Ur2] W0)^T + (L~ x) ([Uz2; Ur2] W1)^T + ([Uz2; Ur2] b)^T
// -- the gate GEMM runs at K = 3 F instead of C + F, and in the backward pass the K = 2 C data gradient of the gates (ds, which only
// ever fed weight gradients of that linear map) and the (2C x C) weight gradient dzr^T h become (2C x F) contractions on x and L~ x
// plus tiny compositions.  h itself is still formed: the reset gate multiplies it (q = h * R) and the blend reads it.
// REGT_TGCN_COLLAPSE=0 / regt_set_option("tgcn_collapse", 0): the uncollapsed form (A/B, tests).
int g_opt_wgrad_pairs = -1;
int wgrad_pairs_setting() {
    if (g_opt_wgrad_pairs < 0) g_opt_wgrad_pairs = 2;
    return g_opt_wgrad_pairs;
}
int g_opt_tcollapse = -1;
bool tcollapse_wanted() {
    if (g_opt_tcollapse < 0) { const char* e = getenv("REGT_TGCN_COLLAPSE"); g_opt_tcollapse = e ? atoi(e) : 1; }
    return g_opt_tcollapse != 0;
}
inline const float* byte_off(const float* p, long bytes) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + bytes); }

// bf16 mode with bf16-stored activations: the GEMM weights get per-step bf16 copies in MFMA fragment order (SEG_B_FRAG: every
// wave loads its B fragments straight into registers) when every K is a multiple of the 32-k slab
bool weights_frag(const regt_dims& d) {
    // ... and every GEMM of the step fits the 64-slab descriptor table of the bf16-operand core (SplitCore::plan_u): the
    // regional embedding repeats its K = F segment once per region a 128-row tile can meet, the gate data gradient has K = 2C
    const long reg_slabs = (long)(std::min<long>(d.R, 128 / d.T + 2) + 1) * (d.F / 32);
    return bf16_intermediates(d) && d.F % 32 == 0 && d.C % 128 == 0 && !gemm_desc_table_forced() && !fp32_core_wide() &&
           reg_slabs <= 64 && (2L * d.C + d.F) / 32 <= 64;
}
bool xbf_ok(const regt_dims& d, const regt_graph& g, bool h_ext, int x_rows, bool packed_fp32) {
    // (fragment-order weights as in weights_frag(), without its bound on the regional GEMM's slab table: the fused kernel has none)
    const bool frag = bf16_intermediates(d) && d.F % 32 == 0 && d.C % 128 == 0 && !gemm_desc_table_forced() && !fp32_core_wide();
    return xbf_wanted() && frag && d.regional && d.R > 1 && !g.overlap && !h_ext && g.m_rowptr && g.m_col && g.m_val_a && g.m_val_l &&
           g.chunk_tab && g.chunk_region && g.n_chunks > 0 &&
           fused_forward_ok(d.C, d.F) && ((long)d.T * d.F) % 64 == 0 && (!packed_fp32 || x_rows <= 2 * d.N) &&
           (long)(x_rows > d.N ? x_rows : d.N) * d.T * d.F * 2 < (1L << 32) - 4096;
}
// The row-owning fused forward (fused_rows.hip) instead of the 64-row one: C = 256, F = 32 / 64, T <= 16 and region ids sorted by node.
// The three-launch path of the same arithmetic follows with its per-node sums (CandArgs::node_sum_rows), whichever forward runs:
// the forms stay bit-identical (tests/test_gpu_fused.py).  regt_set_option("fused_rows", 0): the 64-row kernel everywhere; 2: the
// row-owning kernel as two workgroups of four waves per CU (a test variant, see kernels.h).
static int g_opt_fused_rows = 1;
// regt_set_option("embed_kernel", 0): the regional embedding of the fp32 path through the general GEMM core instead of embed.hip (A/B)
static int g_opt_embed_kernel = 1;
bool fused_rows_form(const regt_dims& d, const regt_graph& g) {
    return g_opt_fused_rows && fused_forward_rows_ok(d.C, d.F, d.T) && d.regional && !g.overlap && (d.R == 1 || g.region_sorted);
}
struct WbPtrs { const float *U[3], *UT[3], *Gzr, *Gh, *A0, *Aall; long ar_stride; };
WbPtrs wb_ptrs(const float* Wb, long C, long F, long R) {
    const char* b = reinterpret_cast<const char*>(Wb);
    WbPtrs w;
    long o = 0;
    for (int k = 0; k < 3; ++k) { w.U[k] = reinterpret_cast<const float*>(b + o); o += frag_bytes(C, C); }
    for (int k = 0; k < 3; ++k) { w.UT[k] = reinterpret_cast<const float*>(b + o); o += frag_bytes(C, C); }
    w.Gzr = reinterpret_cast<const float*>(b + o); o += frag_bytes(2 * C, F);
    w.Gh = reinterpret_cast<const float*>(b + o); o += frag_bytes(C, F);
    w.A0 = reinterpret_cast<const float*>(b + o); o += frag_bytes(C, F);
    w.Aall = reinterpret_cast<const float*>(b + o);
    w.ar_stride = frag_bytes(C, F);
    (void)R;
    return w;
}

GemmSeg make_seg(const float* A, long lda, const float* B0, const float* B1, long ldb, int nsplit, int K, bool bt,
                 int extra_flags = 0, long region_stride = 0) {
    GemmSeg s{};
    s.A = A; s.lda = lda; s.B0 = B0; s.B1 = B1 ? B1 : B0; s.ldb = ldb; s.nsplit = nsplit; s.K = K;
    s.b_region_stride = region_stride;
    int f = extra_flags | (bt ? SEG_BT : 0);
    if (lda % 4 == 0 && al16(A)) f |= SEG_VEC_A;
    if (ldb % 4 == 0 && al16(B0) && al16(s.B1) && region_stride % 4 == 0) f |= SEG_VEC_B;
    s.flags = f;
    return s;
}

struct Layout {
    // saved by forward
    float *Xp, *AX, *LX, *h, *ZR, *q, *Ht, *y1, *probs;
    float *A0, *Aall, *bprime, *Gzr, *Gh, *czr, *ch;
    float *P0zr, *P1zr, *czr2;    // FMT_TCOLLAPSE: [Uz2; Ur2] W0, [Uz2; Ur2] W1 (2C x F each), czr + [Uz2; Ur2] b (2C)
    float *dP01;                  // ... and the gradient of [P0 | P1] (2C x 2F)
    float *S;    // (C, C): sum of the region blocks of tgnn.linear.weight (forward, reused by backward)
    float *G0;   // (C, C): the part of d tgnn.linear.weight every region block shares (backward)
    // backward temporaries
    float *dOH, *d1, *dhp, *dzr, *dh, *dp_partial, *rowdot, *slab;
    float *UT;   // (3, C, C): transposed H-halves of the gate weights (h, z, r) for the data-gradient GEMMs
    float *Wb;   // fragment-order bf16 copies of the GEMM weights (bf16 mode)
    float *dA0, *dAall, *dbprime, *dGzr, *dGh, *dczr, *dch;
    int kchunk, nchunks, kchunk_s, nchunks_s, kchunk_head, nchunks_head, cb_npb, cb_blocks;
    long slab_floats;
    unsigned* tile_ctr;
    size_t bytes;
};

// Row chunks of the head's two weight gradients (N rows -- nodes, not node x period rows).  Rounds 1-4 gave both max(512, N / 64)-row
// chunks: 64 workgroups for dW2 (a latency-bound serial walk: 199 us for 51 MB at cfg-3) and 128 for dW1 -- and at one region per
// GPU (12 500 rows) 25 chunks, fewer workgroups than at twice the rows.  Now: dW2 ~1000 chunks of >= 64 rows (its slabs are O x H1
// floats), dW1 ~three workgroups per CU with >= 128 rows per chunk (its slabs are H1 x C floats: more chunks = more slab traffic).
struct HeadChunks { int k1, n1, k2, n2; };
HeadChunks head_chunks(long N, int H1, int C) {
    HeadChunks h;
    long k2 = ((N + 1023) / 1024 + 7) / 8 * 8;
    if (k2 < 64) k2 = 64;
    h.k2 = (int)k2; h.n2 = (int)((N + k2 - 1) / k2);
    const long tiles = (long)((H1 + 127) / 128) * ((C + 127) / 128), want = 768 / (tiles > 0 ? tiles : 1);
    long k1 = ((N + want - 1) / (want > 0 ? want : 1) + 31) / 32 * 32;
    if (k1 < 128) k1 = 128;
    h.k1 = (int)k1; h.n1 = (int)((N + k1 - 1) / k1);
    return h;
}

Layout make_layout(const regt_dims& d, int n_chunks_tab, int overlap, char* base) {
    Layout L{};
    const long N = d.N, T = d.T, F = d.F, C = d.C, R = d.R, O = d.O, H1 = d.H1;
    const long M = N * T;
    size_t off = 0;
    auto take = [&](long nfloats) {
        size_t o = off;
        off += ((size_t)nfloats * 4 + 255) & ~size_t(255);
        return base ? reinterpret_cast<float*>(base + o) : nullptr;
    };
    L.Xp = take(M * F);
    L.AX = take((overlap ? 1 + R : 2) * M * F);      // A_hat x, then L~ x (merged) or one L~_r x per region
    L.LX = L.AX ? L.AX + M * F : nullptr;
    L.h = take(M * C);
    L.ZR = take(M * 2 * C);
    L.q = take(M * C);
    L.Ht = take(M * C);
    L.y1 = take(N * H1);
    L.probs = take(T);
    L.S = take(C * C);
    L.G0 = take(C * C);
    L.A0 = take(C * F);
    L.Aall = take(R * C * F);
    L.bprime = take(C);
    L.Gzr = take(2 * C * F);
    L.Gh = take(C * F);
    L.czr = take(2 * C);
    L.ch = take(C);
    L.P0zr = take(2 * C * F);
    L.P1zr = take(2 * C * F);
    L.czr2 = take(2 * C);
    L.dP01 = take(2 * C * 2 * F);
    L.UT = take(3 * C * C);
    // bf16 copies of the GEMM weights in MFMA fragment order (REGT_GEMM_MODE=bf16, weights_frag()): Uz, Ur, Uh, UT x 3 (C x C
    // each), Gzr (2C x F), Gh (C x F), A0 (C x F), A_r (R x (C x F)); rows padded to 128 -- sized in floats
    L.Wb = take((6 * frag_bytes(C, C) + frag_bytes(2 * C, F) + (2 + R) * frag_bytes(C, F)) / 4 + 64);
    L.dOH = take(N * C);
    L.d1 = take(N * H1);
    L.dhp = take(M * C);
    L.dzr = take(M * 2 * C);
    L.dh = take(M * C);
    long kc = ((M + 127) / 128 + 31) / 32 * 32;     // ~128 row chunks: 512-1024 wgrad workgroups, half the slab traffic of 256
    if (kc < 128) kc = 128;                          // small graphs: short K loops in many workgroups (latency-bound regime)
    L.kchunk = (int)kc;
    L.nchunks = (int)((M + kc - 1) / kc);
    long ks = ((M + 511) / 512 + 31) / 32 * 32;     // skinny (C x F) gradients: memory-bound, want >= 1024 small workgroups
                                                    // (dGh / dGzr on fp32 rows pick their own count per launch: wgrad_skinny_chunking)
    if (ks < 128) ks = 128;
    L.kchunk_s = (int)ks;
    L.nchunks_s = (int)((M + ks - 1) / ks);
    long kh = ((N + 63) / 64 + 31) / 32 * 32;
    if (kh < 512) kh = 512;
    L.kchunk_head = (int)kh;
    L.nchunks_head = (int)((N + kh - 1) / kh);
    L.cb_npb = (int)((N + 2047) / 2048);
    L.cb_npb = (L.cb_npb + 3) / 4 * 4;
    L.cb_blocks = cell_bwd_blocks((int)N, L.cb_npb);
    L.dp_partial = take((long)L.cb_blocks * T);
    // per-row <dOH, H'>: one float per row (fused backward kernel, fused.hip) or one partial dot per 128-column tile of the row
    // (fp32 candidate data gradient with a generated left operand, gemm_dgrad1_gen_kernel)
    L.rowdot = take(M * (C / 128 > 1 ? C / 128 : 1));
    L.tile_ctr = reinterpret_cast<unsigned*>(take(16));      // tile counter of the persistent fused kernels (fused.hip, fused_rows.hip)
    // one slab region per weight gradient (their reductions are deferred into one launch, ReduceQueue): the sum of
    // Uh, Uzr (wide), Gh, Gzr, A0, A_r (skinny), head1, head2 -- 64 floats of slack each for alignment
    // (chunk counts: the launches pick their own -- wgrad_wide / _skinny / _ring_chunking -- so every region is sized for the larger of
    // the layout's count and what those can return; the paired bf16 launches write (C + F)-wide slabs)
    auto nmax = [&](long layout_chunks, int nout, int nin) { const long b = wgrad_chunk_bound(nout, nin, M); return b > layout_chunks ? b : layout_chunks; };
    long slab = nmax(L.nchunks, C, C + F) * ((long)C * (C + F) + C) + nmax(L.nchunks, 2 * C, C + F) * (2L * C * (C + F) + 2 * C)
              + nmax(L.nchunks_s, C, F) * (C * F) + nmax(L.nchunks_s, 2 * C, F) * (2 * C * F) + (long)L.nchunks_s * (C * F + C)
              + (long)head_chunks(N, (int)H1, (int)C).n1 * (H1 * C + H1 + O * H1 + O) + (long)head_chunks(N, (int)H1, (int)C).n2 * (O * H1 + O) + 8 * 64
              + nmax(L.nchunks_s, 2 * C, 2 * F) * (2 * C * 2 * F + 2 * C) + 64;   // FMT_TCOLLAPSE: dzr^T [x | L~ x] (one or two launches)
    const long ar_uniform = (long)L.nchunks_s * C * F, ar_tab = (long)(n_chunks_tab > 0 ? n_chunks_tab : 1) * C * F;
    slab += ar_tab > ar_uniform ? ar_tab : ar_uniform;
    slab += (long)(n_chunks_tab > 0 ? n_chunks_tab : 1) * (C * F + C);     // fused dA0 | dA_r slabs over the region chunk table
    L.slab_floats = slab;
    L.slab = take(slab);
    L.dA0 = take(C * F);
    L.dAall = take(R * C * F);
    L.dbprime = take(C);
    L.dGzr = take(2 * C * F);
    L.dGh = take(C * F);
    L.dczr = take(2 * C);
    L.dch = take(C);
    L.bytes = off;
    return L;
}

int check_dims(const regt_dims* d) {
    REGT_CHECK_ARG(d != nullptr, "dims is NULL");
    REGT_CHECK_ARG(d->N > 0 && d->T > 0 && d->F > 0 && d->C > 0 && d->R > 0 && d->O > 0 && d->H1 > 0,
                   "dims: all of N,T,F,C,R,O,H1 must be positive (N=%d T=%d F=%d C=%d R=%d O=%d H1=%d)", d->N, d->T,
                   d->F, d->C, d->R, d->O, d->H1);
    REGT_CHECK_ARG(d->F % 4 == 0, "dims: F=%d must be a multiple of 4 (16-byte feature rows)", d->F);
    REGT_CHECK_ARG(d->C % 4 == 0, "dims: C=%d must be a multiple of 4", d->C);
    // (T <= 64: every element of the hidden state is the sum of at most two partial sums -- bit-reproducible; beyond that a node
    // spans three or more 64-row blocks and the order of the float atomics shows in the last bits)
    REGT_CHECK_ARG(d->T <= 255, "dims: T=%d exceeds 255 periods", d->T);
    REGT_CHECK_ARG((long)d->N * d->T < (1L << 31), "dims: N*T too large");
    REGT_CHECK_ARG(d->arith >= REGT_ARITH_DEFAULT && d->arith <= REGT_ARITH_BF16, "dims: arith=%d is not one of REGT_ARITH_*", d->arith);
    return REGT_OK;
}

#define TRY(x)               \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)

// C[m x n] = A[m x k] * B[k x n] helper for contiguous-ish operands (strides given explicitly).
SmallGemm sg(const float* A, long sai, long sak, long sab, const float* B, long sbk, long sbj, long sbb, float* C,
             long sci, long scj, long scb, int m, int n, int k, int batch, int sum_batch, int accumulate) {
    return SmallGemm{A, sai, sak, sab, B, sbk, sbj, sbb, C, sci, scj, scb, m, n, k, batch, sum_batch, accumulate};
}

inline bool fits32(long v) { return v >= INT_MIN && v <= INT_MAX; }
SgTerm term(const float* A, long sai, long sak, long sab, const float* B, long sbk, long sbj, long sbb, int k, int batch = 1,
            int sum_batch = 0) {
    const bool ok = fits32(sai) && fits32(sak) && fits32(sab) && fits32(sbk) && fits32(sbj) && fits32(sbb) && batch <= SHRT_MAX && k >= 0;
    return SgTerm{A, B, (int)sai, (int)sak, (int)sab, (int)sbk, (int)sbj, (int)sbb, ok ? k : -1, (short)batch, (short)sum_batch};
}
void add_task(SgBatch& b, float* C, long sci, long scj, long scb, int m, int n, int nbatch, const float* init, long init_si,
              std::initializer_list<SgTerm> terms, long init_sj = 0) {
    bool ok = b.ntask < SG_MAX_TASKS && fits32(sci) && fits32(scj) && fits32(scb) && fits32(init_si) && fits32(init_sj) && nbatch <= SHRT_MAX &&
              terms.size() <= 3;
    for (const SgTerm& q : terms) ok = ok && q.k >= 0;
    if (!ok) { b.overflow = 1; return; }
    SgTask& t = b.task[b.ntask++];
    t = SgTask{};
    t.C = C; t.sci = (int)sci; t.scj = (int)scj; t.scb = (int)scb; t.m = m; t.n = n; t.nbatch = (short)nbatch; t.init = init;
    t.init_si = (int)init_si; t.init_sj = (int)init_sj;
    for (const SgTerm& q : terms) t.term[t.nterm++] = q;
}

// All weight compositions of one step in ONE launch (SURVEY/DESIGN section 3, item 2).
// owned region block [lo, hi) of a graph (regt_graph.region_lo / region_hi; 0, 0 = all)
void region_range(const regt_dims& d, const regt_graph& g, int* lo, int* hi) {
    *lo = 0; *hi = d.R;
    if (g.region_hi > g.region_lo && g.region_lo >= 0 && g.region_hi <= d.R) { *lo = g.region_lo; *hi = g.region_hi; }
}

int compose_forward(const regt_dims& d, const regt_graph& g, const regt_params& p, const Layout& L, hipStream_t st, bool tcol = false) {
    const int C = d.C, F = d.F, R = d.R;
    const long RC = (long)R * C;
    SgBatch b{};
    if (d.regional) {
        int lo, hi;
        region_range(d, g, &lo, &hi);
        // S = sum_r Wl_r ;  A0 = S W0 ;  A_r = Wl_r W1 (owned regions only) ;  b' = S b_c + b_l
        TRY(launch_sum_region_blocks(p.region_w, L.S, C, R, st));
        add_task(b, L.A0, F, 1, 0, C, F, 1, nullptr, 0, {term(L.S, C, 1, 0, p.cheb_w0, F, 1, 0, C)});
        add_task(b, L.Aall + (long)lo * C * F, F, 1, (long)C * F, C, F, hi - lo, nullptr, 0,
                 {term(p.region_w + (long)lo * C, RC, 1, C, p.cheb_w1, F, 1, 0, C)});
        add_task(b, L.bprime, 1, 0, 0, C, 1, 1, p.region_b, 1, {term(L.S, C, 1, 0, p.cheb_bias, 1, 0, 0, C)});
    }
    for (int k = 0; k < 3; ++k) {
        float* G = k < 2 ? L.Gzr + (long)k * C * F : L.Gh;
        float* c = k < 2 ? L.czr + (long)k * C : L.ch;
        // G_k = U_k[:, :C] V_k ;  c_k = U_k[:, :C] beta_k + u_k
        add_task(b, G, F, 1, 0, C, F, 1, nullptr, 0, {term(p.gate_w[k], 2L * C, 1, 0, p.conv_lin_w[k], F, 1, 0, C)});
        add_task(b, c, 1, 0, 0, C, 1, 1, p.gate_b[k], 1, {term(p.gate_w[k], 2L * C, 1, 0, p.conv_bias[k], 1, 0, 0, C)});
    }
    if (tcol) {
        for (int k = 0; k < 2; ++k) {
            const float* U2 = p.gate_w[k] + C;          // (C x C), row stride 2C: the H half of linear_z / linear_r
            // P0_k = U_k2 W0 ; P1_k = U_k2 W1 ; c'_k = u_k + U_k1 beta_k + U_k2 b
            add_task(b, L.P0zr + (long)k * C * F, F, 1, 0, C, F, 1, nullptr, 0, {term(U2, 2L * C, 1, 0, p.cheb_w0, F, 1, 0, C)});
            add_task(b, L.P1zr + (long)k * C * F, F, 1, 0, C, F, 1, nullptr, 0, {term(U2, 2L * C, 1, 0, p.cheb_w1, F, 1, 0, C)});
            add_task(b, L.czr2 + (long)k * C, 1, 0, 0, C, 1, 1, p.gate_b[k], 1,
                     {term(p.gate_w[k], 2L * C, 1, 0, p.conv_bias[k], 1, 0, 0, C), term(U2, 2L * C, 1, 0, p.cheb_bias, 1, 0, 0, C)});
        }
    }
    return launch_small_gemm_multi(b, st);
}

// head of every model on the path: relu -> linear1 -> relu -> linear2 (models/RegionalTemporalGCN.py:35-38); y1 (N, H1) is kept
// for the backward pass
int head_forward(const regt_dims& d, const regt_params& p, const float* hidden, float* y1, float* pred, hipStream_t st) {
    const int N = d.N, C = d.C, O = d.O, H1 = d.H1;
    GemmSegs S{};
    S.nseg = 1;
    S.seg[0] = make_seg(hidden, C, p.head1_w, nullptr, C, INT_MAX, C, true, SEG_RELU_A);
    S.row_div = 1;
    EpiBiasAct e{y1, H1, p.head1_b, ACT_RELU, 0.f};
    PROF("head_fwd", st);
    TRY(launch_gemm_bias_act(S, N, H1, e, st));
    if (head2_skinny_ok(H1, O, y1, p.head2_w)) {
        TRY(launch_head2_fwd(y1, p.head2_w, p.head2_b, pred, N, H1, O, st));
    } else {
        GemmSegs S2{};
        S2.nseg = 1;
        S2.seg[0] = make_seg(y1, H1, p.head2_w, nullptr, H1, INT_MAX, H1, true);
        S2.row_div = 1;
        EpiBiasAct e2{pred, O, p.head2_b, ACT_NONE, 0.f};
        TRY(launch_gemm_bias_act(S2, N, O, e2, st));
    }
    return REGT_OK;
}

static hipStream_t side_fork(hipStream_t st);      // library side stream (defined with the backward pass below)
static int side_join(hipStream_t st);

// `h_ext` != NULL: the cell's hidden input (M x C, rows node*T + t) comes from the caller (regt_cell_forward); the
// regional / Cheb embedding stage is skipped and only A_hat x is aggregated (graph = the N rows of A_hat).
int forward_impl(const regt_dims& d, const regt_graph& g, const regt_params& p, const float* x, const float* xp_ext,
                 int x_rows, float* pred, float* hidden, const Layout& L, hipStream_t st, bool skip_pack = false,
                 const float* h_ext = nullptr, int fmt = 0) {
    const int N = d.N, T = d.T, F = d.F, C = d.C, R = d.R;
    const long M = (long)N * T;
    const float* H = h_ext ? h_ext : L.h;
    const int qbf = bf16_intermediates(d) ? 1 : 0;      // q (and dhp, dzp|drp in the backward) stored as bf16
    const int abf = qbf && !h_ext ? 1 : 0;              // ... and h, [Z|R], H~ (dh in the backward) too: everything M x C
    // The weight compositions do not depend on the snapshot: they run on the side stream next to pack_x + aggregation and are
    // joined in front of their first consumer (0.03-0.10 ms per step off the critical path at every size).
    {
        hipStream_t sc = side_fork(st);
        PROF("compose_fwd", sc);
        TRY(launch_softmax_small(p.attention, L.probs, T, sc));
        TRY(compose_forward(d, g, p, L, sc, (fmt & FMT_TCOLLAPSE) != 0));
    }
    if (fmt & FMT_XBF) {
        // bf16 rows of x, A_hat x, L~ x + the fused cell kernel (fused.hip)
        const void* Xb = (fmt & FMT_XCALLER) ? static_cast<const void*>(xp_ext) : static_cast<const void*>(L.Xp);
        if (!(fmt & FMT_XCALLER) && !skip_pack) {
            PROF("pack_x", st);
            if (xp_ext) TRY(launch_cvt_rows_bf16(xp_ext, L.Xp, (long)x_rows * T * F, st));
            else TRY(launch_pack_x_bf16(x, L.Xp, N, F, T, st));
        }
        {
            PROF("spmm", st);
            TRY(launch_spmm_dual_bf16(g.m_rowptr, g.m_col, g.m_val_a, g.m_val_l, Xb, L.AX, L.LX, N, xp_ext ? x_rows : N, T * F, st));
        }
        const WbPtrs wbf = wb_ptrs(L.Wb, C, F, R);
        TRY(side_join(st));                      // composed weights ready
        {
            CvtBatch cb{};
            cb.n = 0;
            for (int k = 0; k < 3; ++k) cb.t[cb.n++] = CvtTask{p.gate_w[k] + C, 2L * C, C, C, const_cast<float*>(wbf.U[k])};
            cb.t[cb.n++] = CvtTask{L.Gzr, F, 2 * C, F, const_cast<float*>(wbf.Gzr)};
            cb.t[cb.n++] = CvtTask{L.Gh, F, C, F, const_cast<float*>(wbf.Gh)};
            cb.t[cb.n++] = CvtTask{L.A0, F, C, F, const_cast<float*>(wbf.A0)};
            cb.t[cb.n++] = CvtTask{L.Aall, F, R * C, F, const_cast<float*>(wbf.Aall)};
            PROF("weights_bf16", st);
            TRY(launch_cvt_bf16_frag(cb, st));
        }
        {
            FusedFwdArgs a{};
            a.X = Xb; a.LX = L.LX; a.AX = L.AX;
            a.A0f = wbf.A0; a.Aallf = wbf.Aall; a.ar_stride = wbf.ar_stride;
            a.Uzf = wbf.U[0]; a.Urf = wbf.U[1]; a.Uhf = wbf.U[2]; a.Gzrf = wbf.Gzr; a.Ghf = wbf.Gh;
            a.bprime = L.bprime; a.czr = L.czr; a.ch = L.ch; a.probs = L.probs;
            a.node_region = R > 1 ? g.node_region : nullptr;
            a.h = L.h; a.ZR = L.ZR; a.q = L.q; a.Ht = L.Ht; a.OH = hidden;
            a.M = M; a.T = T; a.slope = d.lrelu_slope; a.act_lrelu = 1; a.tile_ctr = L.tile_ctr;
            PROF("fused_forward", st);
            TRY(launch_zero_f32(hidden, (long)N * C, st));
            if (fused_rows_form(d, g)) TRY(launch_fused_forward_rows(a, C, F, g_opt_fused_rows == 2 ? 4 : 8, st));
            else TRY(launch_fused_forward(a, C, F, st));
        }
        return head_forward(d, p, hidden, L.y1, pred, st);
    }
    // 1. pack the snapshot and aggregate: [A_hat; L~] x  (one stacked SpMM over 2N rows, width T*F)
    const float* Xp = xp_ext ? xp_ext : L.Xp;
    if (!xp_ext && !skip_pack) {
        PROF("pack_x", st);
        TRY(launch_pack_x(x, L.Xp, N, F, T, st));
    }
    {
        PROF("spmm", st);
        if (h_ext)
            TRY(launch_spmm_csr(g.rowptr, g.col, g.val, Xp, L.AX, N, xp_ext ? x_rows : N, T * F, 1, st));
        else if (g.overlap)
            TRY(launch_spmm_csr(g.rowptr, g.col, g.val, Xp, L.AX, (1 + R) * N, xp_ext ? x_rows : N, T * F, 1 + R, st));
        else if (g.m_rowptr && g.m_col && g.m_val_a && g.m_val_l && ((T * F) % 32 == 0 || T * F <= 2048))
            TRY(launch_spmm_dual_x(g.m_rowptr, g.m_col, g.m_val_a, g.m_val_l, Xp, L.AX, L.LX, N, xp_ext ? x_rows : N, T * F, st));
        else
            TRY(launch_spmm_csr(g.rowptr, g.col, g.val, Xp, L.AX, 2 * N, xp_ext ? x_rows : N, T * F, 2, st));
    }
    TRY(side_join(st));                          // composed weights ready
    const float* A0 = d.regional ? L.A0 : p.cheb_w0;
    const float* Aall = d.regional ? L.Aall : p.cheb_w1;
    const float* bpr = d.regional ? L.bprime : p.cheb_bias;
    const bool wfr = abf && weights_frag(d) && d.regional && !g.overlap;
    const WbPtrs wb = wb_ptrs(L.Wb, C, F, R);
    if (wfr) {
        CvtBatch cb{};
        cb.n = 0;
        for (int k = 0; k < 3; ++k) cb.t[cb.n++] = CvtTask{p.gate_w[k] + C, 2L * C, C, C, const_cast<float*>(wb.U[k])};
        cb.t[cb.n++] = CvtTask{L.Gzr, F, 2 * C, F, const_cast<float*>(wb.Gzr)};
        cb.t[cb.n++] = CvtTask{L.Gh, F, C, F, const_cast<float*>(wb.Gh)};
        cb.t[cb.n++] = CvtTask{A0, F, C, F, const_cast<float*>(wb.A0)};
        cb.t[cb.n++] = CvtTask{Aall, F, R * C, F, const_cast<float*>(wb.Aall)};      // C % 128 == 0: region r starts at block row r C / 32
        PROF("weights_bf16", st);
        TRY(launch_cvt_bf16_frag(cb, st));
    }
    // 2. regional embedding h = act(x A0^T + (L~ x) A_region^T + b')
    if (!h_ext) {
        GemmSegs S{};
        S.nseg = 2;
        if (wfr) {
            S.seg[0] = make_seg(Xp, F, wb.A0, nullptr, F, INT_MAX, F, true, SEG_B_FRAG);
            S.seg[1] = make_seg(L.LX, F, wb.Aall, nullptr, F, INT_MAX, F, true, (R > 1 ? SEG_REGION : 0) | SEG_B_FRAG, wb.ar_stride);
        } else {
            S.seg[0] = make_seg(Xp, F, A0, nullptr, F, INT_MAX, F, true);
            S.seg[1] = make_seg(L.LX, F, Aall, nullptr, F, INT_MAX, F, true,
                                g.overlap ? SEG_REPEAT : (R > 1 ? SEG_REGION : 0), (long)C * F);
        }
        S.seg[1].a_rep_stride = M * F;
        S.seg[1].nrep = R;
        S.node_region = g.node_region;
        S.row_div = T;
        int lo, hi;
        region_range(d, g, &lo, &hi);
        S.num_regions = hi - lo;                 // a row tile can only meet the regions that own rows here
        EpiBiasAct e{L.h, C, bpr, d.regional ? ACT_LRELU : ACT_NONE, d.lrelu_slope};
        e.out_bf16 = abf;
        PROF("gemm_regional", st);
        // fp32 at C = 256, F = 32 with node-sorted region ids: the kernel written for this shape (embed.hip); everything else: the general core
        if (!wfr && !abf && gemm_mode() == 0 && !fp32_core_wide() && !gemm_desc_table_forced() && g_opt_embed_kernel && d.regional && !g.overlap &&
            (R == 1 || g.region_sorted) && embed_fp32_ok(M, C, F, T)) {
            TRY(launch_embed_fp32(Xp, L.LX, A0, Aall, R > 1 ? g.node_region : nullptr, bpr, L.h, M, T, ACT_LRELU, d.lrelu_slope, st));
        } else {
            TRY(launch_gemm_bias_act(S, M, C, e, st));
        }
    }
    // 3. update + reset gates: [Z|R] = sigmoid(h [Uz2;Ur2]^T + (A_hat x) [Gz;Gr]^T + [cz;cr]),  q = h*R
    {
        GemmSegs S{};
        S.nseg = 2;
        if (wfr) {
            S.seg[0] = make_seg(H, C, wb.U[0], wb.U[1], C, C, C, true, SEG_A_BF16 | SEG_B_FRAG);
            S.seg[1] = make_seg(L.AX, F, wb.Gzr, nullptr, F, INT_MAX, F, true, SEG_B_FRAG);
        } else if (fmt & FMT_TCOLLAPSE) {      // K = 3 F: the linear hidden input folded into x and L~ x
            S.nseg = 3;
            S.seg[0] = make_seg(Xp, F, L.P0zr, nullptr, F, INT_MAX, F, true);
            S.seg[1] = make_seg(L.LX, F, L.P1zr, nullptr, F, INT_MAX, F, true);
            S.seg[2] = make_seg(L.AX, F, L.Gzr, nullptr, F, INT_MAX, F, true);
        } else {
            S.seg[0] = make_seg(H, C, p.gate_w[0] + C, p.gate_w[1] + C, 2L * C, C, C, true, abf ? SEG_A_BF16 : 0);
            S.seg[1] = make_seg(L.AX, F, L.Gzr, nullptr, F, INT_MAX, F, true);
        }
        S.row_div = T;
        EpiGates e{L.ZR, H, L.q, (fmt & FMT_TCOLLAPSE) ? L.czr2 : L.czr, C};
        e.q_bf16 = qbf; e.h_bf16 = abf; e.zr_bf16 = abf;
        PROF("gemm_gates", st);
        TRY(launch_gemm_gates(S, M, 2 * C, e, st));
    }
    // 4. candidate state, GRU blend and attention-weighted sum over periods -> hidden (N,C)
    {
        CandArgs a{};
        a.S.nseg = 2;
        if (wfr) {
            a.S.seg[0] = make_seg(L.q, C, wb.U[2], nullptr, C, INT_MAX, C, true, (qbf ? SEG_A_BF16 : 0) | SEG_B_FRAG);
            a.S.seg[1] = make_seg(L.AX, F, wb.Gh, nullptr, F, INT_MAX, F, true, SEG_B_FRAG);
        } else {
            a.S.seg[0] = make_seg(L.q, C, p.gate_w[2] + C, nullptr, 2L * C, INT_MAX, C, true, qbf ? SEG_A_BF16 : 0);
            a.S.seg[1] = make_seg(L.AX, F, L.Gh, nullptr, F, INT_MAX, F, true);
        }
        a.S.row_div = T;
        a.num_nodes = N; a.T = T; a.C = C;
        a.bias = L.ch; a.ZR = L.ZR; a.h = H; a.probs = L.probs; a.Ht = L.Ht; a.OH = hidden;
        a.act_bf16 = abf;
        a.node_sum_rows = abf && fused_rows_form(d, g) ? 16 : 64;
        PROF("gemm_candidate", st);
        TRY(launch_gemm_candidate(a, st));
    }
    // 5. head: relu -> linear1 -> relu -> linear2
    return head_forward(d, p, hidden, L.y1, pred, st);
}

// out[Nout x Nin] (+ column sums) = P^T Q over uniform chunks, reduced deterministically.
// Weight-gradient slabs and their reductions inside one backward pass: every wgrad gets its own slab region and its
// reduction is only recorded; flush() runs all recorded reductions in ONE launch.  A region request that does not fit
// flushes first and starts over at the base of the slab (stream order keeps that safe on `st`; the side stream is re-forked
// behind the flush, side_resync).
static int side_join(hipStream_t st);      // work forked to the library's side stream completes before any slab is reduced
static void side_resync(hipStream_t st);
struct ReduceQueue {
    float* base;
    long capacity, used = 0;
    hipStream_t st;
    WgradReduceBatch batch{};
    ReduceQueue(float* b, long cap, hipStream_t s) : base(b), capacity(cap), st(s) {}
    int take(long floats, float** out) {
        floats = (floats + 63) & ~63L;
        REGT_CHECK_ARG(floats <= capacity, "backward: weight-gradient slab of %ld floats exceeds the workspace region (%ld)", floats, capacity);
        if (used + floats > capacity) {
            // the reduction just enqueued on `st` still reads the slabs: work on the side stream must not start overwriting
            // the region handed out next before it has run (side_join only orders `st` behind the side stream)
            TRY(flush());
            side_resync(st);
        }
        *out = base + used;
        used += floats;
        return REGT_OK;
    }
    int push(const WgradReduceArgs& r) {
        if (batch.n == WR_MAX_TASKS) TRY(flush_keep_slab());
        batch.t[batch.n++] = r;
        return REGT_OK;
    }
    int flush_keep_slab() {
        TRY(side_join(st));
        if (batch.n) {
            PROF("wgrad_reduce", st);
            TRY(launch_wgrad_reduce_multi(batch, st));
        }
        batch.n = 0;
        return REGT_OK;
    }
    int flush() {
        TRY(flush_keep_slab());
        used = 0;
        return REGT_OK;
    }
};

// ---- side stream for work off the critical path of a backward pass -----------------------------------------------------------------
// The two weight gradients of the head (a few hundred MB of reads, grids far smaller than the chip) depend only on the head's own
// data gradient; the cell backward that follows on the launch stream does not need them.  They run on a library-owned stream,
// forked behind the kernel that produces d1 and joined in front of the slab reduction: 0.18 ms of the cfg-5 shard step (and
// of a one-region-per-GPU shard, where such fixed costs are what strong scaling loses) overlap the cell backward instead of
// preceding it.  REGT_SIDE_STREAM=0 keeps everything on the launch stream; captured launch sequences (REGT_HIPGRAPH) do too.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool forked = false;
};
// One side stream (+ its event pair and fork state) per (device, launch stream): two launch streams, two devices or two host
// threads with streams of their own never share fork / join state.  (Two threads enqueueing on the SAME launch stream at once
// are the caller's race, as for any stream.)  The table is guarded by a mutex; HIP calls on the entry run under it, too -- they
// only enqueue.
static std::mutex g_side_mu;
static std::map<std::pair<int, hipStream_t>, SideStream> g_sides;
static int g_side_enabled = -1;
static const size_t SIDE_MAX_STREAMS = 64;
static hipStream_t side_fork(hipStream_t st) {       // returns the stream to launch on (st itself when the side stream is off)
    if (t_call_flags & REGT_DIMS_NO_SIDE_STREAM) return st;
    std::lock_guard<std::mutex> lk(g_side_mu);
    if (g_side_enabled < 0) { const char* e = getenv("REGT_SIDE_STREAM"); g_side_enabled = e ? atoi(e) : 1; }
    if (!g_side_enabled || g_graph_mode > 0) return st;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return st; }
    const auto key = std::make_pair(dev, st);
    auto it = g_sides.find(key);
    if (it == g_sides.end()) {
        if (g_sides.size() >= SIDE_MAX_STREAMS) return st;        // a caller cycling through many streams: stay on the launch stream
        SideStream ns;
        if (hipStreamCreateWithFlags(&ns.s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ns.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ns.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (ns.s) (void)hipStreamDestroy(ns.s);
            if (ns.fork) (void)hipEventDestroy(ns.fork);
            return st;
        }
        it = g_sides.emplace(key, ns).first;
    }
    SideStream& ss = it->second;
    if (hipEventRecord(ss.fork, st) != hipSuccess || hipStreamWaitEvent(ss.s, ss.fork, 0) != hipSuccess) { (void)hipGetLastError(); return st; }
    ss.forked = true;
    return ss.s;
}
static int side_join(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return REGT_OK; }
    auto it = g_sides.find(std::make_pair(dev, st));
    if (it == g_sides.end() || !it->second.forked) return REGT_OK;
    SideStream& ss = it->second;
    ss.forked = false;
    REGT_CHECK_HIP(hipEventRecord(ss.join, ss.s));
    REGT_CHECK_HIP(hipStreamWaitEvent(st, ss.join, 0));
    return REGT_OK;
}
// the side stream (if this launch stream has one in use) continues only after everything enqueued on `st` so far: used when a slab
// region is handed out again after an overflow flush -- the reduction that still reads it runs on `st`
static void side_resync(hipStream_t st) {
    {
        std::lock_guard<std::mutex> lk(g_side_mu);
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
        if (g_sides.find(std::make_pair(dev, st)) == g_sides.end()) return;
    }
    (void)side_fork(st);
}

int wgrad_full(ReduceQueue& q, const char* name, const float* P, long ldp, int Nout, const float* Q, long ldq, int Nin, int q_relu,
               long M, int kchunk, int nchunks, float* out, long ldo, float* colsum, hipStream_t st, int p_bf16 = 0, int q_bf16 = 0) {
    WgradArgs a{P, ldp, Nout, Q, ldq, Nin, q_relu, M, kchunk, nullptr, nchunks, nullptr, colsum ? 1 : 0};
    a.p_bf16 = p_bf16; a.q_bf16 = q_bf16;
    TRY(q.take((long)nchunks * wgrad_slab_stride(a), &a.slab));
    {
        PROF(name, st);
        TRY(launch_wgrad(a, st));
    }
    WgradReduceArgs r{};
    r.slab = a.slab; r.nchunks = nchunks; r.slab_stride = wgrad_slab_stride(a); r.elem_offset = 0;
    r.Nout = Nout; r.Nin = Nin; r.chunk_group = nullptr; r.ngroups = 1; r.out = out; r.ldo = ldo; r.group_stride = 0;
    r.colsum_out = colsum; r.colsum_offset = (long)Nout * Nin; r.ncolsum = Nout; r.accumulate = 0;
    return q.push(r);
}

// backward of head_forward: weight / bias gradients of linear2 and linear1 (slabs queued on `rq`) and
// dOH = (d1 A1) * (hidden > 0) + dhidden, the gradient of the attention-weighted hidden state
int head_backward(const regt_dims& d, const regt_params& p, const regt_grads& gr, const float* dpred, const float* dhidden,
                  const float* hidden, const float* y1, float* d1, float* dOH, int kchunk_head_unused, int nchunks_head_unused,
                  ReduceQueue& rq, hipStream_t st) {
    const int N = d.N, C = d.C, O = d.O, H1 = d.H1;
    (void)kchunk_head_unused; (void)nchunks_head_unused;
    const HeadChunks hc = head_chunks(N, H1, C);
    const int kchunk_head = hc.k1, nchunks_head = hc.n1;
    const bool skinny = head2_skinny_ok(H1, O, y1, p.head2_w);
    if (skinny) {
        float* slab = nullptr;
        TRY(rq.take((long)hc.n2 * ((long)O * H1 + O), &slab));
        {   // d1 first: the weight gradients leave the critical path behind it (side stream)
            PROF("head_bwd", st);
            TRY(launch_head2_bwd(dpred, p.head2_w, y1, d1, N, H1, O, st));
        }
        hipStream_t ss = side_fork(st);
        {
            PROF("wgrad_head2", ss);
            TRY(launch_head2_wgrad(dpred, y1, slab, N, H1, O, hc.k2, hc.n2, 1, ss));
        }
        WgradReduceArgs r{};
        r.slab = slab; r.nchunks = hc.n2; r.slab_stride = (long)O * H1 + O; r.elem_offset = 0;
        r.Nout = O; r.Nin = H1; r.chunk_group = nullptr; r.ngroups = 1; r.out = gr.head2_w; r.ldo = H1; r.group_stride = 0;
        r.colsum_out = gr.head2_b; r.colsum_offset = (long)O * H1; r.ncolsum = O; r.accumulate = 0;
        TRY(rq.push(r));
        TRY(wgrad_full(rq, "wgrad_head1", d1, H1, H1, hidden, C, C, 1, N, kchunk_head, nchunks_head, gr.head1_w, C, gr.head1_b, ss));
    } else {
        TRY(wgrad_full(rq, "wgrad_head2", dpred, O, O, y1, H1, H1, 0, N, kchunk_head, nchunks_head, gr.head2_w, H1, gr.head2_b, st));
        // d1 = (dpred A2) * (y1 > 0)
        GemmSegs S{};
        S.nseg = 1;
        S.seg[0] = make_seg(dpred, O, p.head2_w, nullptr, H1, INT_MAX, O, false);
        S.row_div = 1;
        EpiMaskAdd e{d1, H1, y1, H1, nullptr, 0};
        PROF("head_bwd", st);
        TRY(launch_gemm_mask_add(S, N, H1, e, st));
        TRY(wgrad_full(rq, "wgrad_head1", d1, H1, H1, hidden, C, C, 1, N, kchunk_head, nchunks_head, gr.head1_w, C, gr.head1_b, st));
    }
    {   // dOH = (d1 A1) * (hidden > 0) + dhidden
        GemmSegs S{};
        S.nseg = 1;
        S.seg[0] = make_seg(d1, H1, p.head1_w, nullptr, C, INT_MAX, H1, false);
        S.row_div = 1;
        EpiMaskAdd e{dOH, C, hidden, C, dhidden, C};
        PROF("head_bwd", st);
        TRY(launch_gemm_mask_add(S, N, C, e, st));
    }
    return REGT_OK;
}

// `h_ext` / `dh_ext` != NULL (regt_cell_backward): the hidden input was supplied by the caller; its gradient is
// written to dh_ext and the embedding-stage gradients (A0 / A_r / Cheb weights) are skipped.
int backward_impl(const regt_dims& d, const regt_graph& g, const regt_params& p, const regt_grads& gr,
                  const float* dpred, const float* dhidden, const float* hidden, const float* xp_ext, const Layout& L,
                  hipStream_t st, int fmt, const float* h_ext = nullptr, float* dh_ext = nullptr) {
    const int N = d.N, T = d.T, F = d.F, C = d.C, R = d.R;
    const long M = (long)N * T;
    const int qbf = fmt & FMT_QBF, xbf = (fmt & FMT_XBF) ? 1 : 0;     // xbf: x, A_hat x, L~ x hold bf16 rows (the forward's format)
    const bool tcol = (fmt & FMT_TCOLLAPSE) != 0 && !h_ext;            // TemporalGCN: the gates' linear use of h folded into x, L~ x
    const float* Xp = (xp_ext && (!xbf || (fmt & FMT_XCALLER))) ? xp_ext : L.Xp;
    const float* H = h_ext ? h_ext : L.h;
    float* DH = dh_ext ? dh_ext : L.dh;
    const int ibf = bf16_intermediates(d) ? 1 : 0;      // dhp, dzp|drp stored as bf16 (and q, by the forward: checked by the caller)
    const int abf = ibf && !h_ext ? 1 : 0;              // h, [Z|R], H~ were stored as bf16 by the forward; dh / ds follow
    // ---- head ----------------------------------------------------------------------------------
    ReduceQueue rq(L.slab, L.slab_floats, st);
    TRY(head_backward(d, p, gr, dpred, dhidden, hidden, L.y1, L.d1, L.dOH, L.kchunk_head, L.nchunks_head, rq, st));
    // The three-workgroup cores (gemm_split.h: fp32 planes, bf16x3 split, bf16) take weights as [N][K] only: give the data
    // gradients transposed copies of the three C x C blocks.  REGT_FP32_CORE=wide keeps the fp32 path on the two-workgroup
    // core, which reads the weights as they are.
    const bool split = gemm_mode() != 0 || !fp32_core_wide();
    if (split) {
        PROF("transpose_gate_w", st);
        TRY(launch_transpose3(p.gate_w[2] + C, p.gate_w[0] + C, p.gate_w[1] + C, 3, L.UT, C, C, 2L * C, st));
    }
    const bool wfr = abf && ibf && weights_frag(d) && d.regional;
    const WbPtrs wb = wb_ptrs(L.Wb, C, F, R);
    if (wfr) {
        CvtBatch cb{};
        cb.n = 3;
        for (int k = 0; k < 3; ++k) cb.t[k] = CvtTask{L.UT + (long)k * C * C, C, C, C, const_cast<float*>(wb.UT[k])};
        PROF("weights_bf16", st);
        TRY(launch_cvt_bf16_frag(cb, st));
    }
    // bf16 arithmetic with fragment-order weights: the three data-gradient launches below as ONE kernel (fused.hip), every
    // activation read and written once.  Same results bit for bit except the summation order of the attention gradient.
    const bool fused = wfr && !h_ext && fused_backward_ok(C) && fused_bwd_wanted();
    if (fused) {
        FusedBwdArgs a{};
        a.ZR = L.ZR; a.h = H; a.Ht = L.Ht; a.dOH = L.dOH; a.probs = L.probs;
        a.UhTf = wb.UT[0]; a.UzTf = wb.UT[1]; a.UrTf = wb.UT[2];
        a.dhp = L.dhp; a.dzr = L.dzr; a.dh = DH; a.rowdot = L.rowdot;
        a.M = M; a.T = T; a.slope = d.lrelu_slope; a.act_lrelu = d.regional ? 1 : 0; a.tile_ctr = L.tile_ctr;
        {
            PROF("fused_backward", st);
            TRY(launch_fused_backward(a, C, st));
        }
        if (gr.attention) {      // (one-workgroup tail of the attention gradient: off the critical path, joined before the slab reduction)
            hipStream_t sa = side_fork(st);
            TRY(launch_rowdot_reduce(L.rowdot, L.dp_partial, N, T, L.cb_npb, sa));
            TRY(launch_att_bwd(L.dp_partial, L.cb_blocks, L.probs, gr.attention, T, sa));
        }
    } else {
    // fp32 arithmetic at sizes the 128 x 128 core covers: dhp is GENERATED inside the candidate data gradient's K loop and dzp /
    // the attention dots come out of its epilogue -- cell_bwd's pass over Z, h, H~ (5 C floats per row) does not happen
    // (gemm_split.h: run_u_gen; REGT_DGRAD1_GEN=0 restores the two launches)
    const bool gen = split && !ibf && !abf && gemm_dgrad1_gen_ok(M, C, N) && al16(L.Ht) && al16(L.dhp);
    if (gen) {
        GemmSegs S{};
        S.nseg = 1;
        S.seg[0] = make_seg(L.dhp, C, L.UT, nullptr, C, INT_MAX, C, true);      // (A is generated: the pointer is not read)
        S.row_div = T;
        EpiDgrad1 e{H, L.ZR, L.dOH, L.probs, L.dzr, DH, C, T};
        e.Ht = L.Ht; e.dhp = L.dhp; e.rowdot = L.rowdot; e.num_nodes = N;
        {
            PROF("dgrad_candidate", st);
            TRY(launch_gemm_dgrad1_gen(S, M, C, e, st));
        }
        if (gr.attention) {      // (tail of the attention gradient: off the critical path, joined before the slab reduction)
            hipStream_t sa = side_fork(st);
            TRY(launch_rowdot_reduce(L.rowdot, L.dp_partial, N, T, L.cb_npb, sa, C / 128));
            TRY(launch_att_bwd(L.dp_partial, L.cb_blocks, L.probs, gr.attention, T, sa));
        }
    } else {
    // ---- cell: gate pre-activation gradients ------------------------------------------------------
    {
        CellBwdArgs a{L.dOH, L.probs, L.ZR, H, L.Ht, L.dhp, L.dzr, L.dp_partial, N, T, C, L.cb_npb};
        a.out_bf16 = ibf; a.in_bf16 = abf;
        {
            PROF("cell_bwd", st);
            TRY(launch_cell_bwd(a, st));
        }
        if (gr.attention) TRY(launch_att_bwd(L.dp_partial, L.cb_blocks, L.probs, gr.attention, T, side_fork(st)));
    }
    {   // dq = dhp Uh2 ; drp -> dzr[:, C:], dh = dq*R + p_t dOH Z
        GemmSegs S{};
        S.nseg = 1;
        if (wfr) S.seg[0] = make_seg(L.dhp, C, wb.UT[0], nullptr, C, INT_MAX, C, true, SEG_A_BF16 | SEG_B_FRAG);
        else if (split) S.seg[0] = make_seg(L.dhp, C, L.UT, nullptr, C, INT_MAX, C, true, ibf ? SEG_A_BF16 : 0);
        else S.seg[0] = make_seg(L.dhp, C, p.gate_w[2] + C, nullptr, 2L * C, INT_MAX, C, false);
        S.row_div = T;
        EpiDgrad1 e{H, L.ZR, L.dOH, L.probs, L.dzr, DH, C, T};
        e.dzr_bf16 = ibf; e.h_bf16 = abf; e.zr_bf16 = abf; e.dh_bf16 = abf;
        PROF("dgrad_candidate", st);
        TRY(launch_gemm_dgrad1(S, M, C, e, st));
    }
    }
    if (!tcol) {   // ds = (dh + dzp Uz2 + drp Ur2) * act'(h)      (FMT_TCOLLAPSE: never formed -- dh alone feeds dW0 / dW1 / db below)
        GemmSegs S{};
        S.nseg = 2;
        if (wfr) {      // (drp first: the accumulation order of the fused kernel, which multiplies drp while dzp is still on its way)
            S.seg[0] = make_seg(byte_off(L.dzr, 2L * C), 2L * C, wb.UT[2], nullptr, C, INT_MAX, C, true, SEG_A_BF16 | SEG_B_FRAG);
            S.seg[1] = make_seg(L.dzr, 2L * C, wb.UT[1], nullptr, C, INT_MAX, C, true, SEG_A_BF16 | SEG_B_FRAG);
        } else if (split) {
            const int fl = ibf ? SEG_A_BF16 : 0;
            S.seg[0] = make_seg(byte_off(L.dzr, (ibf ? 2L : 4L) * C), 2L * C, L.UT + 2L * C * C, nullptr, C, INT_MAX, C, true, fl);
            S.seg[1] = make_seg(L.dzr, 2L * C, L.UT + (long)C * C, nullptr, C, INT_MAX, C, true, fl);
        } else {
            S.seg[0] = make_seg(L.dzr, 2L * C, p.gate_w[0] + C, nullptr, 2L * C, INT_MAX, C, false);
            S.seg[1] = make_seg(L.dzr + C, 2L * C, p.gate_w[1] + C, nullptr, 2L * C, INT_MAX, C, false);
        }
        S.row_div = T;
        EpiDgrad2 e{DH, H, C, d.regional ? ACT_LRELU : ACT_NONE, d.lrelu_slope};
        e.h_bf16 = abf; e.dh_bf16 = abf;
        PROF("dgrad_gates", st);
        TRY(launch_gemm_dgrad2(S, M, C, e, st));
    }
    }
    // The (C x F)-sized gradients (Gh, Gzr, A0 | A_r) are HBM-bound -- they stream dhp / dzp|drp / ds for a K = F..2F product --
    // while the two big ones (Uh, Uzr) sit on the matrix pipe: REGT_SIDE_WGRADS=1 issues the former on the side stream so that
    // the two kinds overlap (A/B switch; see DESIGN.md section 6 for the measurement).
    hipStream_t sw = st;        // (the skinny weight gradients on the side stream: measured noise, round 3 -- the switch is gone)
    // ---- weight gradients of the K=C contractions and of the composed (C,F) weights -----------------
    // bf16 rows (fused kernels' layout): dUh | dGh = dhp^T [q | A_hat x] and dUzr | dGzr = dzr^T [h | A_hat x] as ONE launch each -- the
    // A_hat x part is a third column tile of the same row chunk on the same XCD, so dhp / dzp|drp cross HBM once instead of twice.
    // (Round 3 measured this form slower, 1.86 vs 1.44 ms for the four: every tile issued the loads of BOTH right-hand operands.
    // Since round 4 a column tile that lies entirely in one operand issues one load, wgrad_split_kernel q_tile.)  REGT_WGRAD_PAIRS=0/1.
    // With the ring kernel (round 4) the paired form wins (0.64 + 0.43 against 0.54 + 0.35 + 0.29 + 0.18 ms at the cfg-5 shard) and is
    // the default whenever that kernel is on; REGT_WGRAD_PAIRS / regt_set_option("wgrad_pairs", 0 | 1 | 2 = follow the ring kernel).
    const int pairs_opt = wgrad_pairs_setting();
    const bool pairs = (pairs_opt == 2 ? wgrad_ring_active() : pairs_opt == 1) && ibf && qbf && abf && xbf && !h_ext && !tcol && C % 128 == 0 && F % 8 == 0 && sw == st;
    if (pairs) {
        int kc = L.kchunk, nc = L.nchunks;
        wgrad_ring_chunking(C, C + F, M, &kc, &nc);          // one wave of workgroups (ring kernel), else the layout's chunks
        WgradArgs a{L.dhp, C, C, L.q, C, C + F, 0, M, kc, nullptr, nc, nullptr, 1};
        a.p_bf16 = 1; a.q_bf16 = 1; a.Q2 = L.AX; a.ldq2 = F; a.nin_split = C;
        TRY(rq.take((long)nc * wgrad_slab_stride(a), &a.slab));
        {
            PROF("wgrad_UhGh", st);
            TRY(launch_wgrad(a, st));
        }
        WgradReduceArgs r{};
        r.slab = a.slab; r.nchunks = nc; r.slab_stride = wgrad_slab_stride(a); r.slab_ld = C + F; r.ngroups = 1;
        r.elem_offset = 0; r.Nout = C; r.Nin = C; r.out = gr.gate_w[2] + C; r.ldo = 2L * C;
        r.colsum_out = L.dch; r.colsum_offset = (long)C * (C + F); r.ncolsum = C;
        TRY(rq.push(r));
        WgradReduceArgs g{};
        g.slab = a.slab; g.nchunks = nc; g.slab_stride = wgrad_slab_stride(a); g.slab_ld = C + F; g.ngroups = 1;
        g.elem_offset = C; g.Nout = C; g.Nin = F; g.out = L.dGh; g.ldo = F;
        TRY(rq.push(g));
    } else {
    {
        int kc = L.kchunk, nc = L.nchunks;
        if (!ibf && !qbf) wgrad_wide_chunking(C, C, M, &kc, &nc);
        TRY(wgrad_full(rq, "wgrad_Uh", L.dhp, C, C, L.q, C, C, 0, M, kc, nc, gr.gate_w[2] + C, 2L * C, L.dch, st, ibf, qbf));
    }
    {
        int kc = L.kchunk_s, nc = L.nchunks_s;
        if (!ibf && !xbf && F <= 32) wgrad_skinny_chunking(C, M, &kc, &nc);
        TRY(wgrad_full(rq, "wgrad_Gh", L.dhp, C, C, L.AX, F, F, 0, M, kc, nc, L.dGh, F, nullptr, sw, ibf, xbf));
    }
    }
    if (pairs) {
        int kc = L.kchunk, nc = L.nchunks;
        wgrad_ring_chunking(2 * C, C + F, M, &kc, &nc);
        WgradArgs a{L.dzr, 2L * C, 2 * C, H, C, C + F, 0, M, kc, nullptr, nc, nullptr, 1};
        a.p_bf16 = 1; a.q_bf16 = 1; a.Q2 = L.AX; a.ldq2 = F; a.nin_split = C;
        TRY(rq.take((long)nc * wgrad_slab_stride(a), &a.slab));
        {
            PROF("wgrad_UzrGzr", st);
            TRY(launch_wgrad(a, st));
        }
        for (int k = 0; k < 2; ++k) {
            WgradReduceArgs r{};
            r.slab = a.slab; r.nchunks = nc; r.slab_stride = wgrad_slab_stride(a); r.slab_ld = C + F; r.ngroups = 1;
            r.elem_offset = (long)k * C * (C + F); r.Nout = C; r.Nin = C; r.out = gr.gate_w[k] + C; r.ldo = 2L * C;
            r.colsum_out = k == 0 ? L.dczr : nullptr; r.colsum_offset = 2L * C * (C + F); r.ncolsum = 2 * C;
            TRY(rq.push(r));
        }
        WgradReduceArgs g{};
        g.slab = a.slab; g.nchunks = nc; g.slab_stride = wgrad_slab_stride(a); g.slab_ld = C + F; g.ngroups = 1;
        g.elem_offset = C; g.Nout = 2 * C; g.Nin = F; g.out = L.dGzr; g.ldo = F;
        TRY(rq.push(g));
    } else if (tcol) {
        // [dP0 | dP1] = dzr^T [x | L~ x]  (2C x 2F), column sums -> [dcz; dcr]: what is left of dzr^T h (the composition backward
        // below turns it into dUz2 / dUr2 / dW0 / dW1 / db).  One launch with a two-part right-hand side when F is a multiple of
        // the 32-column tile, else one launch per part.
        const bool two = F % 32 == 0;
        for (int part = 0; part < (two ? 1 : 2); ++part) {
            int kcs = L.kchunk_s, ncs = L.nchunks_s;
            if (!ibf && !xbf && (two ? 2 * F : F) <= 64) wgrad_skinny_chunking(2 * C, M, &kcs, &ncs);      // one wave of workgroups
            WgradArgs a{L.dzr, 2L * C, 2 * C, part ? L.LX : Xp, F, two ? 2 * F : F, 0, M, kcs, nullptr, ncs, nullptr, part == 0 ? 1 : 0};
            if (two) { a.Q2 = L.LX; a.ldq2 = F; a.nin_split = F; }
            TRY(rq.take((long)ncs * wgrad_slab_stride(a), &a.slab));
            {
                PROF("wgrad_P01", st);
                TRY(launch_wgrad(a, st));
            }
            WgradReduceArgs r{};
            r.slab = a.slab; r.nchunks = ncs; r.slab_stride = wgrad_slab_stride(a); r.elem_offset = 0;
            r.Nout = 2 * C; r.Nin = two ? 2 * F : F; r.ngroups = 1;
            r.out = L.dP01 + (two ? 0 : part * F); r.ldo = 2L * F;
            r.colsum_out = part == 0 ? L.dczr : nullptr; r.colsum_offset = (long)2 * C * (two ? 2 * F : F); r.ncolsum = 2 * C;
            TRY(rq.push(r));
        }
    } else {   // [dUz2; dUr2] = dzr^T h, column sums -> [dcz; dcr]
        int kc = L.kchunk, nc = L.nchunks;
        if (!ibf && !abf) wgrad_wide_chunking(2 * C, C, M, &kc, &nc);
        WgradArgs a{L.dzr, 2L * C, 2 * C, H, C, C, 0, M, kc, nullptr, nc, nullptr, 1};
        a.p_bf16 = ibf; a.q_bf16 = abf;
        TRY(rq.take((long)nc * wgrad_slab_stride(a), &a.slab));
        {
            PROF("wgrad_Uzr", st);
            TRY(launch_wgrad(a, st));
        }
        for (int k = 0; k < 2; ++k) {
            WgradReduceArgs r{};
            r.slab = a.slab; r.nchunks = nc; r.slab_stride = wgrad_slab_stride(a);
            r.elem_offset = (long)k * C * C; r.Nout = C; r.Nin = C; r.ngroups = 1;
            r.out = gr.gate_w[k] + C; r.ldo = 2L * C;
            r.colsum_out = k == 0 ? L.dczr : nullptr; r.colsum_offset = 2L * C * C; r.ncolsum = 2 * C;
            TRY(rq.push(r));
        }
    }
    // (One launch per pair with a two-part right-hand side [q | A_hat x] / [h | A_hat x] -- so that dhp and dzp|drp are read
    // once -- was measured and is slower: 1.86 vs 1.44 ms for the four at the cfg-5 shard; the third, half-empty column tile and
    // the doubled load instructions of the two-descriptor staging cost more than the second pass over the left operand.)
    if (!pairs) {
        int kc = L.kchunk_s, nc = L.nchunks_s;
        if (!ibf && !xbf && F <= 32) wgrad_skinny_chunking(2 * C, M, &kc, &nc);
        TRY(wgrad_full(rq, "wgrad_Gzr", L.dzr, 2L * C, 2 * C, L.AX, F, F, 0, M, kc, nc, L.dGzr, F, nullptr, sw, ibf, xbf));
    }
    float* dA0 = d.regional ? L.dA0 : gr.cheb_w0;
    float* dAall = d.regional ? L.dAall : gr.cheb_w1;
    float* dbpr = d.regional ? L.dbprime : gr.cheb_bias;
    // node-disjoint regions (the headline case): dA0 = ds^T x and dA_r = ds^T (L~ x) share ds -- one launch over the
    // region-pure row chunks with [x | L~ x] as a two-part right-hand side, so that ds is read from HBM once
    const bool fuse_a = !h_ext && !g.overlap && R > 1 && F % 32 == 0;
    if (fuse_a) {
        REGT_CHECK_ARG(g.chunk_tab && g.chunk_region && g.n_chunks > 0, "backward: region chunk table missing");
        WgradArgs a{L.dh, C, C, Xp, F, 2 * F, 0, M, 0, g.chunk_tab, g.n_chunks, nullptr, 1};
        a.Q2 = L.LX; a.ldq2 = F; a.nin_split = F;
        a.p_bf16 = abf; a.q_bf16 = xbf;
        TRY(rq.take((long)g.n_chunks * wgrad_slab_stride(a), &a.slab));
        {
            PROF("wgrad_A0_Ar", sw);
            TRY(launch_wgrad(a, sw));
        }
        WgradReduceArgs r0{};
        r0.slab = a.slab; r0.nchunks = g.n_chunks; r0.slab_stride = wgrad_slab_stride(a); r0.elem_offset = 0; r0.slab_ld = 2 * F;
        r0.Nout = C; r0.Nin = F; r0.chunk_group = nullptr; r0.ngroups = 1; r0.out = dA0; r0.ldo = F;
        r0.colsum_out = dbpr; r0.colsum_offset = 2L * C * F; r0.ncolsum = C;
        TRY(rq.push(r0));
        WgradReduceArgs r1{};
        r1.slab = a.slab; r1.nchunks = g.n_chunks; r1.slab_stride = wgrad_slab_stride(a); r1.elem_offset = F; r1.slab_ld = 2 * F;
        int lo, hi;
        region_range(d, g, &lo, &hi);           // only the owned region blocks have rows here (and are read later)
        r1.Nout = C; r1.Nin = F; r1.chunk_group = g.chunk_region; r1.ngroups = hi - lo; r1.group_base = lo;
        r1.out = dAall + (long)lo * C * F; r1.ldo = F;
        r1.group_stride = (long)C * F;
        TRY(rq.push(r1));
    }
    if (!h_ext && !fuse_a) TRY(wgrad_full(rq, "wgrad_A0", L.dh, C, C, Xp, F, F, 0, M, L.kchunk_s, L.nchunks_s, dA0, F, dbpr, st, abf, 0));
    if (h_ext || fuse_a) {
        // no embedding stage behind a caller-supplied hidden input / already done above
    } else if (g.overlap) {   // one unmasked (C x F) gradient per region: dA_r = ds^T (L~_r x)
        for (int r = 0; r < R; ++r)
            TRY(wgrad_full(rq, "wgrad_Ar", L.dh, C, C, L.LX + (long)r * M * F, F, F, 0, M, L.kchunk_s, L.nchunks_s,
                           dAall + (long)r * C * F, F, nullptr, st, abf, 0));
    } else if (R > 1) {   // per-region dA_r = sum over the region's rows of ds^T (L~ x)
        REGT_CHECK_ARG(g.chunk_tab && g.chunk_region && g.n_chunks > 0, "backward: region chunk table missing");
        WgradArgs a{L.dh, C, C, L.LX, F, F, 0, M, 0, g.chunk_tab, g.n_chunks, nullptr, 0};
        a.p_bf16 = abf;
        TRY(rq.take((long)g.n_chunks * wgrad_slab_stride(a), &a.slab));
        {
            PROF("wgrad_Ar", st);
            TRY(launch_wgrad(a, st));
        }
        WgradReduceArgs r{};
        r.slab = a.slab; r.nchunks = g.n_chunks; r.slab_stride = wgrad_slab_stride(a); r.elem_offset = 0;
        r.Nout = C; r.Nin = F; r.chunk_group = g.chunk_region; r.ngroups = R; r.out = dAall; r.ldo = F;
        r.group_stride = (long)C * F;
        TRY(rq.push(r));
    } else {
        TRY(wgrad_full(rq, "wgrad_Ar", L.dh, C, C, L.LX, F, F, 0, M, L.kchunk_s, L.nchunks_s, dAall, F, nullptr, st, abf, 0));
    }
    TRY(rq.flush());      // every slab reduction of this backward pass, one launch
    // ---- back through the weight compositions (tiny; one launch) ------------------------------------------
    PROF("compose_bwd", st);
    {
        SgBatch b{};
        for (int k = 0; k < 3; ++k) {
            const float* dG = k < 2 ? L.dGzr + (long)k * C * F : L.dGh;
            const float* dc = k < 2 ? L.dczr + (long)k * C : L.dch;
            // dU_k[:, :C] = dG_k V_k^T + dc_k beta_k^T
            add_task(b, gr.gate_w[k], 2L * C, 1, 0, C, C, 1, nullptr, 0,
                     {term(dG, F, 1, 0, p.conv_lin_w[k], 1, F, 0, F), term(dc, 1, 0, 0, p.conv_bias[k], 0, 1, 0, 1)});
            // dV_k = U_k[:, :C]^T dG_k ; dbeta_k = U_k[:, :C]^T dc_k
            // (stated transposed -- output "rows" j, "columns" i -- so that consecutive lanes walk the CONTIGUOUS index i of the
            // left factor U_k[k, i]; as (i, j) every k-step of a lane group touched lines 2C floats apart)
            add_task(b, gr.conv_lin_w[k], 1, F, 0, F, C, 1, nullptr, 0, {term(dG, 1, F, 0, p.gate_w[k], 2L * C, 1, 0, C)});
            add_task(b, gr.conv_bias[k], 1, 0, 0, C, 1, 1, nullptr, 0, {term(p.gate_w[k], 1, 2L * C, 0, dc, 1, 0, 0, C)});
        }
        // (one launch since round 4: nothing below reads what the tasks above write -- G0 = dA0 W0^T + db' b_c^T, what EVERY
        // block of d tgnn.linear.weight receives, is summed inside each block's task instead of through a buffer)
        for (int k = 0; k < 3; ++k)      // du_k = dc_k
            add_task(b, gr.gate_b[k], 1, 0, 0, C, 1, 1, k < 2 ? L.dczr + (long)k * C : L.dch, 1, {});
        if (tcol) {
            // back through P0_k = U_k2 W0, P1_k = U_k2 W1, c'_k = c_k + U_k2 b  (k = z, r):
            //   dU_k2 = dP0_k W0^T + dP1_k W1^T + dc_k b^T ;  dW0 += sum_k U_k2^T dP0_k ;  dW1 += sum_k U_k2^T dP1_k ;  db += sum_k U_k2^T dc_k
            // (dW0 / dW1 / db already hold the direct path dh^T x / dh^T L~ x / colsum dh from the slab reduction above: added in place;
            // the last three stated transposed -- output "rows" f, "columns" c -- so that lanes walk the contiguous index of U_k2)
            const float* dP0[2] = {L.dP01, L.dP01 + (long)C * 2 * F};
            const float* dP1[2] = {L.dP01 + F, L.dP01 + (long)C * 2 * F + F};
            const float* dc[2] = {L.dczr, L.dczr + C};
            for (int k = 0; k < 2; ++k)
                add_task(b, gr.gate_w[k] + C, 2L * C, 1, 0, C, C, 1, nullptr, 0,
                         {term(dP0[k], 2L * F, 1, 0, p.cheb_w0, 1, F, 0, F), term(dP1[k], 2L * F, 1, 0, p.cheb_w1, 1, F, 0, F),
                          term(dc[k], 1, 0, 0, p.cheb_bias, 0, 1, 0, 1)});
            add_task(b, gr.cheb_w0, 1, F, 0, F, C, 1, gr.cheb_w0, 1,
                     {term(dP0[0], 1, 2L * F, 0, p.gate_w[0] + C, 2L * C, 1, 0, C), term(dP0[1], 1, 2L * F, 0, p.gate_w[1] + C, 2L * C, 1, 0, C)}, F);
            add_task(b, gr.cheb_w1, 1, F, 0, F, C, 1, gr.cheb_w1, 1,
                     {term(dP1[0], 1, 2L * F, 0, p.gate_w[0] + C, 2L * C, 1, 0, C), term(dP1[1], 1, 2L * F, 0, p.gate_w[1] + C, 2L * C, 1, 0, C)}, F);
            add_task(b, gr.cheb_bias, 1, 0, 0, C, 1, 1, gr.cheb_bias, 1,
                     {term(p.gate_w[0] + C, 1, 2L * C, 0, dc[0], 1, 0, 0, C), term(p.gate_w[1] + C, 1, 2L * C, 0, dc[1], 1, 0, 0, C)});
        }
        if (d.regional) {
            const long RC = (long)R * C;
            int lo, hi;
            region_range(d, g, &lo, &hi);
            // dWl_r = G0 + dA_r W1^T for the owned regions, G0 alone for the others (their rows live on other GPUs);
            // G0 = dA0 W0^T + db' b_c^T (A0 and b' sum over all regions)
            const SgTerm g0a = term(L.dA0, F, 1, 0, p.cheb_w0, 1, F, 0, F), g0b = term(L.dbprime, 1, 0, 0, p.cheb_bias, 0, 1, 0, 1);
            add_task(b, gr.region_w + (long)lo * C, RC, 1, C, C, C, hi - lo, nullptr, 0,
                     {g0a, g0b, term(L.dAall + (long)lo * C * F, F, 1, (long)C * F, p.cheb_w1, 1, F, 0, F)});
            if (lo > 0) add_task(b, gr.region_w, RC, 1, C, C, C, lo, nullptr, 0, {g0a, g0b});
            if (hi < R) add_task(b, gr.region_w + (long)hi * C, RC, 1, C, C, C, R - hi, nullptr, 0, {g0a, g0b});
            // dW0 = S^T dA0 ; dW1 = sum_{owned r} Wl_r^T dA_r ; db_c = S^T db' ; db_l = db'
            // (both stated transposed, see dV_k above: tgnn.linear.weight rows are R*C floats apart)
            add_task(b, gr.cheb_w0, 1, F, 0, F, C, 1, nullptr, 0, {term(L.dA0, 1, F, 0, L.S, C, 1, 0, C)});
            add_task(b, gr.cheb_w1, 1, F, 0, F, C, 1, nullptr, 0,
                     {term(L.dAall + (long)lo * C * F, 1, F, (long)C * F, p.region_w + (long)lo * C, RC, 1, C, C, hi - lo, 1)});
            add_task(b, gr.cheb_bias, 1, 0, 0, C, 1, 1, nullptr, 0, {term(L.S, 1, C, 0, L.dbprime, 1, 0, 0, C)});
            add_task(b, gr.region_b, 1, 0, 0, C, 1, 1, L.dbprime, 1, {});
        }
        TRY(launch_small_gemm_multi(b, st));
    }
    return REGT_OK;
}

int check_ptrs(const regt_params* p, const regt_dims& d, bool cell_only = false) {
    REGT_CHECK_ARG(p != nullptr, "params is NULL");
    bool ok = p->attention && p->head1_w && p->head1_b && p->head2_w && p->head2_b;
    if (!cell_only) ok = ok && p->cheb_w0 && p->cheb_w1 && p->cheb_bias;
    for (int k = 0; k < 3; ++k) ok = ok && p->conv_lin_w[k] && p->conv_bias[k] && p->gate_w[k] && p->gate_b[k];
    if (d.regional && !cell_only) ok = ok && p->region_w && p->region_b;
    REGT_CHECK_ARG(ok, "params: a required tensor pointer is NULL");
    return REGT_OK;
}

}  // namespace
}  // namespace regt

using namespace regt;

extern "C" {

int32_t regt_abi_version(void) { return REGT_ABI_VERSION; }

int32_t regt_set_gemm_mode(int32_t mode) {
    const int prev = gemm_mode();
    set_gemm_mode(mode);
    return prev;
}
const char* regt_last_error(void) { return g_err; }

int32_t regt_set_option(const char* name, int32_t value) {
    REGT_CHECK_ARG(name != nullptr, "regt_set_option: name is NULL");
    if (!strcmp(name, "xbf")) { const int prev = xbf_wanted() ? 1 : 0; g_opt_xbf = value ? 1 : 0; return prev; }
    if (!strcmp(name, "fused_rows")) { const int prev = g_opt_fused_rows; g_opt_fused_rows = value == 2 ? 2 : (value ? 1 : 0); return prev; }
    if (!strcmp(name, "embed_kernel")) { const int prev = g_opt_embed_kernel; g_opt_embed_kernel = value ? 1 : 0; return prev; }
    if (!strcmp(name, "fused_bwd")) { const int prev = fused_bwd_wanted() ? 1 : 0; g_opt_fused_bwd = value ? 1 : 0; return prev; }
    if (!strcmp(name, "spmm_rows")) return spmm_rows_option(value);
    if (!strcmp(name, "dgrad1_gen")) return dgrad1_gen_option(value);
    if (!strcmp(name, "wgrad_ring")) return wgrad_ring_option(value);
    if (!strcmp(name, "wgrad_tile")) return wgrad_tile_option(value);
    if (!strcmp(name, "wgrad_ring256")) return wgrad_ring256_option(value == 4 ? 4 : 2);
    if (!strcmp(name, "wgrad_bnw64")) return wgrad_bnw64_option(value ? 1 : 0);
    if (!strcmp(name, "wgrad_wave")) return wgrad_wave_option(value ? 1 : 0);
    if (!strcmp(name, "wgrad_pairs")) { const int prev = wgrad_pairs_setting(); g_opt_wgrad_pairs = value < 0 || value > 2 ? 2 : value; return prev; }
    if (!strcmp(name, "tgcn_collapse")) { const int prev = tcollapse_wanted() ? 1 : 0; g_opt_tcollapse = value ? 1 : 0; return prev; }
    set_error("regt_set_option: unknown option '%s'", name);
    return -1;
}

size_t regt_graph_workspace_bytes(int64_t E, int32_t N) { return graph_workspace_bytes((long)E, N); }

int32_t regt_gcn_csr(const int64_t* ei, const float* w, int64_t E, int32_t N, int32_t* rowptr, int32_t* col, float* val,
                     int32_t* flags_dev, void* ws, size_t ws_bytes, regt_stream_t st) {
    REGT_CHECK_ARG((ei || E == 0) && rowptr && col && val && flags_dev && ws, "regt_gcn_csr: NULL pointer");
    return graph_gcn_csr(ei, w, (long)E, N, rowptr, col, val, flags_dev, ws, ws_bytes, (hipStream_t)st);
}

int32_t regt_gcn_dis(const int64_t* ei, const float* w, int64_t E, int32_t N, float* dis_out, int32_t* flags_dev, void* ws, size_t ws_bytes,
                     regt_stream_t st) {
    REGT_CHECK_ARG((ei || E == 0) && dis_out && flags_dev && ws, "regt_gcn_dis: NULL pointer");
    return graph_gcn_dis(ei, w, (long)E, N, dis_out, flags_dev, ws, ws_bytes, (hipStream_t)st);
}

int32_t regt_cheb_edge_weights(const int64_t* ei, const float* w, int64_t E, int32_t N, float* out, int32_t* flags_dev,
                               void* ws, size_t ws_bytes, regt_stream_t st) {
    REGT_CHECK_ARG((ei || E == 0) && (out || E == 0) && flags_dev && ws, "regt_cheb_edge_weights: NULL pointer");
    return graph_cheb_edge_weights(ei, w, (long)E, N, out, flags_dev, ws, ws_bytes, (hipStream_t)st);
}

int32_t regt_raw_csr(const int64_t* ei, const float* v, int64_t E, int32_t N, int32_t* rowptr, int32_t* col, float* val,
                     int32_t* flags_dev, void* ws, size_t ws_bytes, regt_stream_t st) {
    REGT_CHECK_ARG((ei || E == 0) && rowptr && col && val && flags_dev && ws, "regt_raw_csr: NULL pointer");
    return graph_raw_csr(ei, v, (long)E, N, rowptr, col, val, flags_dev, ws, ws_bytes, (hipStream_t)st);
}

int32_t regt_graph_fingerprint(const int64_t* ei, const float* w, int64_t E, uint64_t* out_dev, regt_stream_t st) {
    REGT_CHECK_ARG((ei || E == 0) && out_dev, "regt_graph_fingerprint: NULL pointer");
    return graph_fingerprint(ei, w, (long)E, reinterpret_cast<unsigned long long*>(out_dev), (hipStream_t)st);
}

int32_t regt_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const float* X, float* Y, int32_t nrows,
                      int32_t nrows_x, int32_t width, regt_stream_t st) {
    REGT_CHECK_ARG(rowptr && col && val && X && Y, "regt_spmm_csr: NULL pointer");
    return launch_spmm_csr(rowptr, col, val, X, Y, nrows, nrows_x, width, 1, (hipStream_t)st);
}

int32_t regt_spmm_dual(const int32_t* rowptr, const int32_t* col, const float* val_a, const float* val_l, const float* X,
                       float* YA, float* YL, int32_t N, int32_t width, regt_stream_t st) {
    REGT_CHECK_ARG(rowptr && col && val_a && val_l && X && YA && YL, "regt_spmm_dual: NULL pointer");
    return launch_spmm_dual(rowptr, col, val_a, val_l, X, YA, YL, N, width, (hipStream_t)st);
}

int32_t regt_pack_x(const float* x, float* xp, int32_t N, int32_t F, int32_t T, regt_stream_t st) {
    REGT_CHECK_ARG(x && xp && N > 0 && F > 0 && T > 0, "regt_pack_x: bad argument");
    return launch_pack_x(x, xp, N, F, T, (hipStream_t)st);
}

int32_t regt_linear(const float* A, int64_t lda, int64_t M, int32_t K, const float* W, int64_t ldw, int32_t N,
                    const float* bias, int32_t act, float slope, float* out, int64_t ldo, regt_stream_t st) {
    REGT_CHECK_ARG(A && W && out && M > 0 && K > 0 && N > 0, "regt_linear: bad argument");
    REGT_CHECK_ARG(act >= 0 && act <= 4, "regt_linear: act must be 0 (none), 1 (leaky_relu), 2 (relu), 3 (sigmoid) or 4 (tanh)");
    GemmSegs S{};
    S.nseg = 1;
    S.seg[0] = make_seg(A, lda, W, nullptr, ldw, INT_MAX, K, true);
    S.row_div = 1;
    EpiBiasAct e{out, ldo, bias, act, slope};
    return launch_gemm_bias_act(S, M, N, e, (hipStream_t)st);
}

static void wgrad_chunks(int64_t M, int* kchunk, int* nchunks) {
    long kc = ((M + 127) / 128 + 31) / 32 * 32;
    if (kc < 512) kc = 512;
    *kchunk = (int)kc;
    *nchunks = (int)((M + kc - 1) / kc);
}

size_t regt_wgrad_slab_floats(int64_t M, int32_t N, int32_t K, int32_t with_bias) {
    int kc, nc;
    wgrad_chunks(M, &kc, &nc);
    return (size_t)nc * ((size_t)N * K + (with_bias ? N : 0));
}

int32_t regt_wgrad(const float* dOut, int64_t ldd, const float* A, int64_t lda, int64_t M, int32_t N, int32_t K, float* dW,
                   int64_t ldw, float* dbias, float* slab, regt_stream_t st) {
    REGT_CHECK_ARG(dOut && A && dW && slab && M > 0 && N > 0 && K > 0, "regt_wgrad: bad argument");
    int kc, nc;
    wgrad_chunks(M, &kc, &nc);
    // the caller's slab is sized by regt_wgrad_slab_floats; the 64-float alignment slack of the queue is not needed here
    ReduceQueue rq(slab, ((long)nc * ((long)N * K + (dbias ? N : 0)) + 63) & ~63L, (hipStream_t)st);
    TRY(wgrad_full(rq, "wgrad", dOut, ldd, N, A, lda, K, 0, M, kc, nc, dW, ldw, dbias, (hipStream_t)st));
    return rq.flush();
}

size_t regt_workspace_bytes(const regt_dims* dims, int32_t n_chunks, int32_t overlap) {
    if (check_dims(dims)) return 0;
    return make_layout(*dims, n_chunks, overlap, nullptr).bytes;
}

static int32_t forward_common(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const float* x,
                              const float* xp_ext, int32_t x_rows, float* pred, float* hidden, void* ws, size_t ws_bytes,
                              regt_stream_t st, bool xp_is_bf16 = false) {
    TRY(check_dims(dims));
    CallScope call(dims);
    REGT_CHECK_ARG(graph && graph->rowptr && graph->col && graph->val && (graph->node_region || graph->overlap), "regt_forward: graph incomplete");
    TRY(check_ptrs(params, *dims));
    REGT_CHECK_ARG((x || xp_ext) && pred && hidden && ws, "regt_forward: NULL pointer");
    REGT_CHECK_ARG(al16(x) && al16(xp_ext) && al16(hidden) && al16(ws),
                   "regt_forward: x, hidden and workspace must be 16-byte aligned");
    REGT_CHECK_ARG(!xp_ext || x_rows >= dims->N, "regt_forward_packed: x_rows=%d < N=%d", x_rows, dims->N);
    Layout L = make_layout(*dims, graph->n_chunks, graph->overlap, (char*)ws);
    REGT_CHECK_ARG(ws_bytes >= L.bytes, "regt_forward: workspace %zu < required %zu bytes", ws_bytes, L.bytes);
    hipStream_t hs = (hipStream_t)st;
    int fmt = bf16_intermediates(*dims) ? FMT_QBF : 0;
    if (fmt && xbf_ok(*dims, *graph, false, xp_ext ? x_rows : dims->N, xp_ext && !xp_is_bf16)) fmt |= FMT_XBF | (xp_is_bf16 ? FMT_XCALLER : 0);
    if (!dims->regional && !graph->overlap && !(fmt & FMT_QBF) && tcollapse_wanted()) fmt |= FMT_TCOLLAPSE;
    REGT_CHECK_ARG(!xp_is_bf16 || (fmt & FMT_XBF), "regt_forward_packed_bf16: bf16 input rows need REGT_GEMM_MODE=bf16 and a shape the fused "
                   "forward covers (C = 256, F = 64, node-disjoint regions, merged operator)");
    note_q_format(ws, fmt);
    if (!graphs_wanted((long)dims->N * dims->T))
        return forward_impl(*dims, *graph, *params, x, xp_ext, x_rows, pred, hidden, L, hs, false, nullptr, fmt);
    // the snapshot changes every step: pack it with a plain launch, replay everything behind it
    if (!xp_ext) {
        if (fmt & FMT_XBF) TRY(launch_pack_x_bf16(x, L.Xp, dims->N, dims->F, dims->T, hs));
        else TRY(launch_pack_x(x, L.Xp, dims->N, dims->F, dims->T, hs));
    } else if ((fmt & FMT_XBF) && !(fmt & FMT_XCALLER)) {
        TRY(launch_cvt_rows_bf16(xp_ext, L.Xp, (long)x_rows * dims->T * dims->F, hs));
    }
    unsigned long long key = hash_bytes(dims, sizeof(*dims), 0xcbf29ce484222325ull);
    key = hash_bytes(graph, sizeof(*graph), key);
    key = hash_bytes(params, sizeof(*params), key);
    const void* ptrs[6] = {xp_ext, pred, hidden, ws, (const void*)(long)x_rows, (const void*)(long)fmt};
    key = hash_bytes(ptrs, sizeof(ptrs), key);
    const regt_dims dd = *dims; const regt_graph gg = *graph; const regt_params pp = *params;
    return run_maybe_graphed(g_fwd_graphs, key, hs, [=](hipStream_t s) {
        return forward_impl(dd, gg, pp, x, xp_ext, x_rows, pred, hidden, L, s, /*skip_pack=*/true, nullptr, fmt);
    });
}

int32_t regt_forward(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const float* x, float* pred,
                     float* hidden, void* ws, size_t ws_bytes, regt_stream_t st) {
    REGT_CHECK_ARG(x != nullptr, "regt_forward: x is NULL");
    return forward_common(dims, graph, params, x, nullptr, 0, pred, hidden, ws, ws_bytes, st);
}

int32_t regt_forward_packed(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const float* x_packed,
                            int32_t x_rows, float* pred, float* hidden, void* ws, size_t ws_bytes, regt_stream_t st) {
    REGT_CHECK_ARG(x_packed != nullptr, "regt_forward_packed: x_packed is NULL");
    return forward_common(dims, graph, params, nullptr, x_packed, x_rows, pred, hidden, ws, ws_bytes, st);
}

int32_t regt_forward_packed_bf16(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const void* x_packed_bf16,
                                 int32_t x_rows, float* pred, float* hidden, void* ws, size_t ws_bytes, regt_stream_t st) {
    REGT_CHECK_ARG(x_packed_bf16 != nullptr, "regt_forward_packed_bf16: x_packed is NULL");
    return forward_common(dims, graph, params, nullptr, static_cast<const float*>(x_packed_bf16), x_rows, pred, hidden, ws, ws_bytes, st, true);
}

int32_t regt_pack_x_bf16(const float* x, void* xp, int32_t N, int32_t F, int32_t T, regt_stream_t st) {
    REGT_CHECK_ARG(x && xp && N > 0 && F > 0 && T > 0, "regt_pack_x_bf16: bad argument");
    return launch_pack_x_bf16(x, xp, N, F, T, (hipStream_t)st);
}

int32_t regt_spmm_dual_bf16(const int32_t* rowptr, const int32_t* col, const float* val_a, const float* val_l, const void* X,
                            void* YA, void* YL, int32_t N, int32_t x_rows, int32_t width, regt_stream_t st) {
    REGT_CHECK_ARG(rowptr && col && val_a && val_l && X && YA && YL && x_rows >= N, "regt_spmm_dual_bf16: NULL pointer / x_rows < N");
    return launch_spmm_dual_bf16(rowptr, col, val_a, val_l, X, YA, YL, N, x_rows, width, (hipStream_t)st);
}

int32_t regt_backward(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const regt_grads* grads,
                      const float* dpred, const float* dhidden, const float* hidden, const float* x_packed, void* ws,
                      size_t ws_bytes, regt_stream_t st) {
    TRY(check_dims(dims));
    CallScope call(dims);
    REGT_CHECK_ARG(graph && graph->rowptr && (graph->node_region || graph->overlap), "regt_backward: graph incomplete");
    TRY(check_ptrs(params, *dims));
    REGT_CHECK_ARG(grads && dpred && hidden && ws, "regt_backward: NULL pointer");
    {
        const regt_grads& g = *grads;
        bool ok = g.cheb_w0 && g.cheb_w1 && g.cheb_bias && g.head1_w && g.head1_b && g.head2_w && g.head2_b;
        for (int k = 0; k < 3; ++k) ok = ok && g.conv_lin_w[k] && g.conv_bias[k] && g.gate_w[k] && g.gate_b[k];
        if (dims->regional) ok = ok && g.region_w && g.region_b;
        REGT_CHECK_ARG(ok, "regt_backward: a required gradient pointer is NULL (only `attention` may be NULL)");
    }
    Layout L = make_layout(*dims, graph->n_chunks, graph->overlap, (char*)ws);
    REGT_CHECK_ARG(ws_bytes >= L.bytes, "regt_backward: workspace %zu < required %zu bytes", ws_bytes, L.bytes);
    hipStream_t hs = (hipStream_t)st;
    const int qbf = q_format(ws);
    REGT_CHECK_ARG((qbf & FMT_QBF) == (bf16_intermediates(*dims) ? 1 : 0),
                   "regt_backward: the GEMM arithmetic changed since the forward on this workspace (another regt_dims.arith, or regt_set_gemm_mode between forward and backward)");
    REGT_CHECK_ARG(!(qbf & FMT_XCALLER) || x_packed, "regt_backward: the forward ran on the caller's bf16 packed input; pass the same buffer as x_packed");
    if (!graphs_wanted((long)dims->N * dims->T))
        return backward_impl(*dims, *graph, *params, *grads, dpred, dhidden, hidden, x_packed, L, hs, qbf);
    unsigned long long key = hash_bytes(dims, sizeof(*dims), 0x84222325cbf29ce4ull);
    key = hash_bytes(graph, sizeof(*graph), key);
    key = hash_bytes(params, sizeof(*params), key);
    key = hash_bytes(grads, sizeof(*grads), key);
    const void* ptrs[6] = {dpred, dhidden, hidden, x_packed, ws, (const void*)(long)qbf};
    key = hash_bytes(ptrs, sizeof(ptrs), key);
    const regt_dims dd = *dims; const regt_graph gg = *graph; const regt_params pp = *params; const regt_grads gr = *grads;
    return run_maybe_graphed(g_bwd_graphs, key, hs, [=](hipStream_t s) {
        return backward_impl(dd, gg, pp, gr, dpred, dhidden, hidden, x_packed, L, s, qbf);
    });
}

/* out[0..5] = forward {eager, captured, replayed}, backward {eager, captured, replayed} launch-sequence counts */
int32_t regt_graph_stats(int64_t* out) {
    REGT_CHECK_ARG(out != nullptr, "regt_graph_stats: NULL pointer");
    out[0] = g_fwd_graphs.eager; out[1] = g_fwd_graphs.captured; out[2] = g_fwd_graphs.replayed;
    out[3] = g_bwd_graphs.eager; out[4] = g_bwd_graphs.captured; out[5] = g_bwd_graphs.replayed;
    return REGT_OK;
}

/* ---- TGCN cell + attention + head on a caller-supplied hidden input (regtgcn.h) --------------------------------- */
int32_t regt_cell_forward(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const float* x,
                          const float* h_in, float* pred, float* hidden, void* ws, size_t ws_bytes, regt_stream_t st) {
    TRY(check_dims(dims));
    CallScope call(dims);
    REGT_CHECK_ARG(dims->regional == 0, "regt_cell_forward: dims.regional must be 0");
    REGT_CHECK_ARG(graph && graph->rowptr && graph->col && graph->val && !graph->overlap, "regt_cell_forward: graph incomplete");
    TRY(check_ptrs(params, *dims, true));
    REGT_CHECK_ARG(x && h_in && pred && hidden && ws, "regt_cell_forward: NULL pointer");
    REGT_CHECK_ARG(al16(x) && al16(h_in) && al16(hidden) && al16(ws), "regt_cell_forward: x, h_in, hidden and workspace must be 16-byte aligned");
    Layout L = make_layout(*dims, 0, 0, (char*)ws);
    REGT_CHECK_ARG(ws_bytes >= L.bytes, "regt_cell_forward: workspace %zu < required %zu bytes", ws_bytes, L.bytes);
    note_q_format(ws, bf16_intermediates(*dims) ? 1 : 0);
    return forward_impl(*dims, *graph, *params, x, nullptr, 0, pred, hidden, L, (hipStream_t)st, false, h_in);
}

int32_t regt_cell_backward(const regt_dims* dims, const regt_graph* graph, const regt_params* params, const regt_grads* grads,
                           const float* dpred, const float* dhidden, const float* hidden, const float* h_in, float* dh_in,
                           void* ws, size_t ws_bytes, regt_stream_t st) {
    TRY(check_dims(dims));
    CallScope call(dims);
    REGT_CHECK_ARG(dims->regional == 0, "regt_cell_backward: dims.regional must be 0");
    REGT_CHECK_ARG(graph && graph->rowptr && !graph->overlap, "regt_cell_backward: graph incomplete");
    TRY(check_ptrs(params, *dims, true));
    REGT_CHECK_ARG(grads && dpred && hidden && h_in && dh_in && ws, "regt_cell_backward: NULL pointer");
    REGT_CHECK_ARG(al16(h_in) && al16(dh_in), "regt_cell_backward: h_in and dh_in must be 16-byte aligned");
    {
        const regt_grads& g = *grads;
        bool ok = g.head1_w && g.head1_b && g.head2_w && g.head2_b;
        for (int k = 0; k < 3; ++k) ok = ok && g.conv_lin_w[k] && g.conv_bias[k] && g.gate_w[k] && g.gate_b[k];
        REGT_CHECK_ARG(ok, "regt_cell_backward: a required gradient pointer is NULL (only `attention` may be NULL)");
    }
    Layout L = make_layout(*dims, 0, 0, (char*)ws);
    REGT_CHECK_ARG(ws_bytes >= L.bytes, "regt_cell_backward: workspace %zu < required %zu bytes", ws_bytes, L.bytes);
    const int qbf = q_format(ws);
    REGT_CHECK_ARG((qbf & FMT_QBF) == (bf16_intermediates(*dims) ? 1 : 0),
                   "regt_cell_backward: the GEMM arithmetic changed since the forward on this workspace");
    return backward_impl(*dims, *graph, *params, *grads, dpred, dhidden, hidden, nullptr, L, (hipStream_t)st, qbf, h_in, dh_in);
}

/* ---- zero-hidden cell (GraphSAGE / GAT models of the reference) ---------------------------------------------------------- */
namespace regt { namespace {
struct Layout0 {
    float *Z, *Ht, *y1, *probs, *dOH, *d1, *dzp, *dhp, *dp_partial, *slab;
    int kchunk, nchunks, kchunk_head, nchunks_head, cb_npb, cb_blocks;
    long slab_floats;
    size_t bytes;
};
Layout0 make_layout0(const regt_dims& d, int kz, int kh, char* base) {
    Layout0 L{};
    const long N = d.N, T = d.T, C = d.C, O = d.O, H1 = d.H1, M = N * T;
    size_t off = 0;
    auto take = [&](long nfloats) {
        size_t o = off;
        off += ((size_t)nfloats * 4 + 255) & ~size_t(255);
        return base ? reinterpret_cast<float*>(base + o) : nullptr;
    };
    L.Z = take(M * C); L.Ht = take(M * C); L.y1 = take(N * H1); L.probs = take(T);
    L.dOH = take(N * C); L.d1 = take(N * H1); L.dzp = take(M * C); L.dhp = take(M * C);
    long ks = ((M + 511) / 512 + 31) / 32 * 32;           // skinny (C x k) gradients: memory-bound, many small workgroups
    if (ks < 128) ks = 128;
    L.kchunk = (int)ks; L.nchunks = (int)((M + ks - 1) / ks);
    long kh_ = ((N + 63) / 64 + 31) / 32 * 32;
    if (kh_ < 512) kh_ = 512;
    L.kchunk_head = (int)kh_; L.nchunks_head = (int)((N + kh_ - 1) / kh_);
    L.cb_npb = (int)((N + 2047) / 2048);
    L.cb_npb = (L.cb_npb + 3) / 4 * 4;
    L.cb_blocks = cell_bwd_blocks((int)N, L.cb_npb);
    L.dp_partial = take((long)L.cb_blocks * T);
    L.slab_floats = (long)L.nchunks * (C * kz + C) + (long)L.nchunks * (C * kh + C) + (long)head_chunks(N, (int)H1, (int)C).n1 * (H1 * C + H1 + O * H1 + O) + (long)head_chunks(N, (int)H1, (int)C).n2 * (O * H1 + O) + 8 * 64;
    L.slab = take(L.slab_floats);
    L.bytes = off;
    return L;
}
int check_cell0(const regt_dims* d, const regt_cell0_args* a) {
    REGT_CHECK_ARG(d && a, "regt_cell0: NULL dims / args");
    REGT_CHECK_ARG(d->N > 0 && d->T > 0 && d->C > 0 && d->O > 0 && d->H1 > 0 && d->C % 4 == 0 && d->T <= 255, "regt_cell0: bad dims");
    REGT_CHECK_ARG((long)d->N * d->T < (1L << 31), "regt_cell0: N*T too large");
    REGT_CHECK_ARG(a->kz > 0 && a->kh > 0 && a->a_z && a->a_h && a->gz && a->gh && a->cz && a->ch && a->attention && a->head1_w &&
                   a->head1_b && a->head2_w && a->head2_b, "regt_cell0: a required pointer is NULL");
    return REGT_OK;
}
regt_params head_params(const regt_cell0_args& a) {
    regt_params p{};
    p.head1_w = a.head1_w; p.head1_b = a.head1_b; p.head2_w = a.head2_w; p.head2_b = a.head2_b;
    return p;
}
}}  // namespace regt::(anonymous)

size_t regt_cell0_workspace_bytes(const regt_dims* dims, int32_t kz, int32_t kh) {
    if (!dims || dims->N <= 0 || dims->T <= 0 || dims->C <= 0 || dims->O <= 0 || dims->H1 <= 0 || kz <= 0 || kh <= 0) return 0;
    return make_layout0(*dims, kz, kh, nullptr).bytes;
}

int32_t regt_cell0_forward(const regt_dims* dims, const regt_cell0_args* args, float* pred, float* hidden, void* ws, size_t ws_bytes,
                           regt_stream_t st_) {
    TRY(check_cell0(dims, args));
    CallScope call(dims);
    REGT_CHECK_ARG(pred && hidden && ws && al16(hidden) && al16(ws), "regt_cell0_forward: NULL / unaligned pointer");
    const regt_dims& d = *dims;
    const regt_cell0_args& a = *args;
    Layout0 L = make_layout0(d, a.kz, a.kh, (char*)ws);
    REGT_CHECK_ARG(ws_bytes >= L.bytes, "regt_cell0_forward: workspace %zu < required %zu bytes", ws_bytes, L.bytes);
    hipStream_t st = (hipStream_t)st_;
    const long M = (long)d.N * d.T;
    TRY(launch_softmax_small(a.attention, L.probs, d.T, st));
    for (int k = 0; k < 2; ++k) {      // Z = sigmoid(a_z gz^T + cz), H~ = tanh(a_h gh^T + ch)
        GemmSegs S{};
        S.nseg = 1;
        S.seg[0] = make_seg(k ? a.a_h : a.a_z, k ? a.kh : a.kz, k ? a.gh : a.gz, nullptr, k ? a.kh : a.kz, INT_MAX, k ? a.kh : a.kz, true);
        S.row_div = 1;
        EpiBiasAct e{k ? L.Ht : L.Z, d.C, k ? a.ch : a.cz, k ? ACT_TANH : ACT_SIGMOID, 0.f};
        PROF(k ? "cell0_candidate" : "cell0_gate", st);
        TRY(launch_gemm_bias_act(S, M, d.C, e, st));
    }
    {
        PROF("cell0_blend", st);
        TRY(launch_blend0_fwd(L.Z, L.Ht, L.probs, hidden, d.N, d.T, d.C, st));
    }
    return head_forward(d, head_params(a), hidden, L.y1, pred, st);
}

int32_t regt_cell0_backward(const regt_dims* dims, const regt_cell0_args* args, const regt_cell0_grads* grads, const float* dpred,
                            const float* dhidden, const float* hidden, void* ws, size_t ws_bytes, regt_stream_t st_) {
    TRY(check_cell0(dims, args));
    CallScope call(dims);
    REGT_CHECK_ARG(grads && dpred && hidden && ws, "regt_cell0_backward: NULL pointer");
    const regt_cell0_grads& g = *grads;
    REGT_CHECK_ARG(g.gz && g.gh && g.cz && g.ch && g.head1_w && g.head1_b && g.head2_w && g.head2_b,
                   "regt_cell0_backward: a required gradient pointer is NULL (only attention, a_z, a_h may be NULL)");
    const regt_dims& d = *dims;
    const regt_cell0_args& a = *args;
    Layout0 L = make_layout0(d, a.kz, a.kh, (char*)ws);
    REGT_CHECK_ARG(ws_bytes >= L.bytes, "regt_cell0_backward: workspace %zu < required %zu bytes", ws_bytes, L.bytes);
    hipStream_t st = (hipStream_t)st_;
    const int N = d.N, T = d.T, C = d.C;
    const long M = (long)N * T;
    ReduceQueue rq(L.slab, L.slab_floats, st);
    regt_grads hg{};
    hg.head1_w = g.head1_w; hg.head1_b = g.head1_b; hg.head2_w = g.head2_w; hg.head2_b = g.head2_b;
    TRY(head_backward(d, head_params(a), hg, dpred, dhidden, hidden, L.y1, L.d1, L.dOH, L.kchunk_head, L.nchunks_head, rq, st));
    {   // dhp = g (1-Z)(1-H~^2), dzp = -g H~ Z (1-Z), g = p_t dOH: the GRU backward head with h = 0
        CellBwdArgs c{L.dOH, L.probs, L.Z, nullptr, L.Ht, L.dhp, L.dzp, L.dp_partial, N, T, C, L.cb_npb};
        c.ldz = C; c.lddz = C;
        PROF("cell_bwd", st);
        TRY(launch_cell_bwd(c, st));
        if (g.attention) TRY(launch_att_bwd(L.dp_partial, L.cb_blocks, L.probs, g.attention, T, st));
    }
    TRY(wgrad_full(rq, "wgrad_gz", L.dzp, C, C, a.a_z, a.kz, a.kz, 0, M, L.kchunk, L.nchunks, g.gz, a.kz, g.cz, st));
    TRY(wgrad_full(rq, "wgrad_gh", L.dhp, C, C, a.a_h, a.kh, a.kh, 0, M, L.kchunk, L.nchunks, g.gh, a.kh, g.ch, st));
    TRY(rq.flush());
    for (int k = 0; k < 2; ++k) {      // optional input gradients: da = dpre G  (M x C) (C x k)
        float* da = k ? g.a_h : g.a_z;
        if (!da) continue;
        const int kk = k ? a.kh : a.kz;
        GemmSegs S{};
        S.nseg = 1;
        S.seg[0] = make_seg(k ? L.dhp : L.dzp, C, k ? a.gh : a.gz, nullptr, kk, INT_MAX, C, false);
        S.row_div = 1;
        EpiBiasAct e{da, kk, nullptr, ACT_NONE, 0.f};
        PROF("cell0_dinput", st);
        TRY(launch_gemm_bias_act(S, M, kk, e, st));
    }
    return REGT_OK;
}

int32_t regt_gat_forward(const int32_t* rowptr, const int32_t* col, const float* x, const float* u_src, const float* u_dst, float slope,
                         int32_t N, int32_t T, int32_t F, float* out, float* stats, regt_stream_t st) {
    REGT_CHECK_ARG(rowptr && col && x && u_src && u_dst && out && stats, "regt_gat_forward: NULL pointer");
    REGT_CHECK_ARG(al16(x) && al16(u_src) && al16(u_dst) && al16(out) && al16(stats), "regt_gat_forward: pointers must be 16-byte aligned");
    return launch_gat_forward(rowptr, col, x, u_src, u_dst, slope, N, T, F, out, stats, (hipStream_t)st);
}

int32_t regt_gat_backward(const int32_t* rowptr, const int32_t* col, const int32_t* t_rowptr, const int32_t* t_col, const float* x,
                          const float* u_src, float slope, int32_t N, int32_t T, int32_t F, const float* dout, float* stats, float* dsd,
                          regt_stream_t st) {
    REGT_CHECK_ARG(rowptr && col && t_rowptr && t_col && x && u_src && dout && stats && dsd, "regt_gat_backward: NULL pointer");
    REGT_CHECK_ARG(al16(x) && al16(u_src) && al16(dout) && al16(stats), "regt_gat_backward: pointers must be 16-byte aligned");
    return launch_gat_backward(rowptr, col, t_rowptr, t_col, x, u_src, slope, N, T, F, dout, stats, dsd, (hipStream_t)st);
}

int32_t regt_mean_csr(const int64_t* ei, int64_t E, int32_t N, int32_t* rowptr, int32_t* col, float* val, int32_t* flags_dev, void* ws,
                      size_t ws_bytes, regt_stream_t st) {
    REGT_CHECK_ARG((ei || E == 0) && rowptr && col && val && flags_dev && ws, "regt_mean_csr: NULL pointer");
    return graph_mean_csr(ei, (long)E, N, rowptr, col, val, flags_dev, ws, ws_bytes, (hipStream_t)st);
}

int64_t regt_debug_trace(int64_t* out_host, int64_t capacity) { return fused_trace_fetch(reinterpret_cast<long*>(out_host), (long)capacity); }

int32_t regt_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return REGT_OK;
}

/* Waits for all recorded events, writes one line per stage "name count total_ms\n" into buf, clears the records. */
int32_t regt_profile_collect(char* buf, size_t buf_bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, std::pair<long, double>> agg;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto& a = agg[r.name];
            a.first += 1;
            a.second += ms;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    std::string out;
    char line[160];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    REGT_CHECK_ARG(buf && buf_bytes > out.size(), "regt_profile_collect: buffer too small (%zu needed)", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return REGT_OK;
}

int32_t regt_mse_loss_grad(const float* pred, const float* y, float* dpred, float* loss_out, int64_t count,
                           int64_t global_count, regt_stream_t st) {
    REGT_CHECK_ARG(pred && y && count > 0 && global_count > 0, "regt_mse_loss_grad: bad argument");
    return launch_mse_grad(pred, y, dpred, loss_out, (long)count, 1.0f / (float)global_count, (hipStream_t)st);
}

}  // extern "C"
