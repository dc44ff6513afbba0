// Graph preparation on the GPU (once per static graph; the result is cached by the caller).
//
// Builds destination-sorted CSR operators from the reference's COO inputs
// (edge_index (2,E) int64, optional edge weights):
//   * GCN  : A_hat = D^-1/2 (A + I) D^-1/2, D = weighted in-degree incl. the self loop
//            (what GCNConv recomputes on every call, models/utils.py:169,175,181)
//   * Cheb : L~ = -D^-1/2 A D^-1/2, D = weighted out-degree, zero diagonal
//            (what ChebConv(K=2, sym, lambda_max=None) recomputes, RegionalTemporalGCN.py:136-140)
// Everything is deterministic: edges are bucketed by key with integer atomics, every bucket is then
// sorted by edge id, and all floating-point sums run in edge order (= the order of PyG's CPU
// scatter_add), so two builds of the same graph give bit-identical operators.
#include "kernels.h"

namespace regt {

namespace {

constexpr int TPB = 256;

__global__ void k_fill_int(int* p, long n, int v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// count edges per key (key_row: 0 = source, 1 = destination); self loops are skipped and, when
// loop_eid != nullptr, the largest edge id of each node's own loops is recorded ("last one wins").
__global__ void k_count(const int64_t* ei, long E, int N, int key_row, int* cnt, int* loop_eid, int* bad, int keep_loops) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        int64_t s = ei[e], d = ei[E + e];
        if (s < 0 || s >= N || d < 0 || d >= N) { atomicOr(bad, 1); continue; }
        if (s == d && !keep_loops) { if (loop_eid) atomicMax(&loop_eid[s], (int)e); continue; }
        atomicAdd(&cnt[key_row ? d : s], 1);
    }
}

// exclusive scan of (cnt[i] + extra) into ptr[0..N]; single workgroup, two passes.
__global__ __launch_bounds__(1024) void k_scan(const int* cnt, int N, int extra, int* ptr) {
    __shared__ long part[1024];
    const int tid = threadIdx.x;
    const long per = ((long)N + 1023) / 1024;
    const long b = tid * per, e = b + per < N ? b + per : N;
    long s = 0;
    for (long i = b; i < e; ++i) s += cnt[i] + extra;
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long run = tid ? part[tid - 1] : 0;
    for (long i = b; i < e; ++i) { ptr[i] = (int)run; run += cnt[i] + extra; }
    if (tid == 1023) ptr[N] = (int)part[1023];
}

__global__ void k_bucket(const int64_t* ei, long E, int N, int key_row, const int* ptr, int* cursor, int* eid, int keep_loops) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        int64_t s = ei[e], d = ei[E + e];
        if (s < 0 || s >= N || d < 0 || d >= N || (s == d && !keep_loops)) continue;
        int k = (int)(key_row ? d : s);
        eid[ptr[k] + atomicAdd(&cursor[k], 1)] = (int)e;
    }
}

// sort the first cnt[i] entries of every bucket by edge id (insertion sort for short buckets,
// heap sort otherwise); one thread per bucket.
__global__ void k_sort_buckets(const int* ptr, const int* cnt, int N, int* eid) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        int* a = eid + ptr[i];
        const int n = cnt[i];
        if (n <= 32) {
            for (int x = 1; x < n; ++x) {
                int v = a[x], y = x - 1;
                while (y >= 0 && a[y] > v) { a[y + 1] = a[y]; --y; }
                a[y + 1] = v;
            }
        } else {
            for (int start = n / 2 - 1; start >= 0; --start) {
                int root = start, v = a[root];
                for (;;) {
                    int ch = 2 * root + 1;
                    if (ch >= n) break;
                    if (ch + 1 < n && a[ch + 1] > a[ch]) ++ch;
                    if (a[ch] <= v) break;
                    a[root] = a[ch]; root = ch;
                }
                a[root] = v;
            }
            for (int end = n - 1; end > 0; --end) {
                int v = a[end]; a[end] = a[0];
                int root = 0;
                for (;;) {
                    int ch = 2 * root + 1;
                    if (ch >= end) break;
                    if (ch + 1 < end && a[ch + 1] > a[ch]) ++ch;
                    if (a[ch] <= v) break;
                    a[root] = a[ch]; root = ch;
                }
                a[root] = v;
            }
        }
    }
}

__device__ __forceinline__ float inv_sqrt_or_zero(float d) {
    float r = 1.0f / sqrtf(d);       // correctly rounded sqrt and divide; deg == 0 -> inf -> 0 (PyG masked_fill)
    return isinf(r) ? 0.f : r;
}

// dis[i] = (sum of bucket weights in edge order [+ loop weight])^-1/2
__global__ void k_degree(const int* ptr, const int* cnt, const int* eid, const float* w, const int* loop_eid,
                         int with_loop, int N, float* dis, float* loopw, int* bad) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        float s = 0.f;
        const int b = ptr[i];
        for (int p = 0; p < cnt[i]; ++p) {
            float wv = w ? w[eid[b + p]] : 1.0f;
            if (wv < 0.f) atomicOr(bad, 2);
            s += wv;
        }
        if (with_loop) {
            float lw = (w && loop_eid[i] >= 0) ? w[loop_eid[i]] : 1.0f;
            loopw[i] = lw;
            s += lw;
        }
        dis[i] = inv_sqrt_or_zero(s);
    }
}

// GCN rows: kept in-edges in edge order, then the node's self loop.
__global__ void k_emit_gcn(const int64_t* ei, long E, const float* w, const int* ptr, const int* cnt, const int* eid,
                           const float* dis, const float* loopw, int N, int* col, float* val) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int b = ptr[i], n = cnt[i];
        const float di = dis[i];
        for (int p = 0; p < n; ++p) {
            int e = eid[b + p];
            int s = (int)ei[e];
            col[b + p] = s;
            val[b + p] = (dis[s] * (w ? w[e] : 1.0f)) * di;
        }
        col[b + n] = i;
        val[b + n] = (di * loopw[i]) * di;
    }
}

// Cheb per-edge weights: w~_e = -(dis[src] * w_e) * dis[dst]; loops -> 0.  (2 w / lambda_max with
// lambda_max = 2 max(w~, 1) = 2 is exact, and the +1/-1 on the diagonal cancels.)
__global__ void k_cheb_edge(const int64_t* ei, long E, int N, const float* w, const float* dis, float* out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        int64_t s = ei[e], d = ei[E + e];
        float v = 0.f;
        if (s >= 0 && s < N && d >= 0 && d < N && s != d) v = -((dis[s] * (w ? w[e] : 1.0f)) * dis[d]);
        out[e] = v;
    }
}

__global__ void k_emit_raw(const int64_t* ei, long E, const float* w, const int* ptr, const int* cnt, const int* eid,
                           int N, int* col, float* val) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int b = ptr[i];
        for (int p = 0; p < cnt[i]; ++p) {
            int e = eid[b + p];
            col[b + p] = (int)ei[e];
            val[b + p] = w ? w[e] : 1.0f;
        }
    }
}

// mean aggregation rows (SAGEConv aggr='mean'): every listed in-edge j -> i, self loops and duplicates included as listed,
// weight 1 / (number of in-edges of i)
__global__ void k_emit_mean(const int64_t* ei, long E, const int* ptr, const int* cnt, const int* eid, int N, int* col, float* val) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int b = ptr[i], n = cnt[i];
        const float w = n > 0 ? 1.0f / (float)n : 0.f;
        for (int p = 0; p < n; ++p) {
            col[b + p] = (int)ei[eid[b + p]];
            val[b + p] = w;
        }
    }
}

__global__ void k_fingerprint(const int64_t* ei, const float* w, long E, unsigned long long* out) {
    unsigned long long h = 0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)ei[e] * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)ei[E + e] + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        if (w) x ^= (unsigned long long)__float_as_uint(w[e]) * 0x165667B19E3779F9ull;
        x ^= (unsigned long long)e * 0xD6E8FEB86659FD93ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        h += x;     // order-independent combination: integer add commutes
    }
    atomicAdd(out, h);
}

inline int nblk(long n) { long b = (n + TPB - 1) / TPB; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

struct Ws {
    int *cnt, *cursor, *loop_eid, *eid, *ptr, *flags;
    float *dis, *loopw;
};

size_t ws_layout(long E, int N, char* base, Ws* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return base ? base + o : nullptr; };
    int* cnt = (int*)take((size_t)(N + 1) * 4);
    int* cursor = (int*)take((size_t)N * 4);
    int* loop_eid = (int*)take((size_t)N * 4);
    int* eid = (int*)take((size_t)(E > 0 ? E : 1) * 4);
    int* ptr = (int*)take((size_t)(N + 1) * 4);
    int* flags = (int*)take(256);
    float* dis = (float*)take((size_t)N * 4);
    float* loopw = (float*)take((size_t)N * 4);
    if (w) *w = Ws{cnt, cursor, loop_eid, eid, ptr, flags, dis, loopw};
    return off;
}

// bucket edges by key into (ptr, eid) inside the workspace; `extra` slots are reserved at the end of each bucket.
int bucket(const int64_t* ei, long E, int N, int key_row, int extra, const Ws& w, int* ptr_out, hipStream_t st, int keep_loops = 0) {
    hipLaunchKernelGGL(k_fill_int, dim3(nblk(N + 1)), dim3(TPB), 0, st, w.cnt, (long)N + 1, 0);
    hipLaunchKernelGGL(k_fill_int, dim3(nblk(N)), dim3(TPB), 0, st, w.cursor, (long)N, 0);
    hipLaunchKernelGGL(k_fill_int, dim3(nblk(N)), dim3(TPB), 0, st, w.loop_eid, (long)N, -1);
    if (E > 0) hipLaunchKernelGGL(k_count, dim3(nblk(E)), dim3(TPB), 0, st, ei, E, N, key_row, w.cnt, w.loop_eid, w.flags, keep_loops);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, w.cnt, N, extra, ptr_out);
    if (E > 0) hipLaunchKernelGGL(k_bucket, dim3(nblk(E)), dim3(TPB), 0, st, ei, E, N, key_row, ptr_out, w.cursor, w.eid, keep_loops);
    hipLaunchKernelGGL(k_sort_buckets, dim3(nblk(N)), dim3(TPB), 0, st, ptr_out, w.cnt, N, w.eid);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace

size_t graph_workspace_bytes(long E, int N) { return ws_layout(E, N, nullptr, nullptr); }

// flags_out_dev[0]: bit0 = index out of range, bit1 = negative weight.
int graph_gcn_csr(const int64_t* ei, const float* w, long E, int N, int* rowptr, int* col, float* val,
                  int* flags_out_dev, void* ws, size_t ws_bytes, hipStream_t st) {
    REGT_CHECK_ARG(N > 0 && E >= 0, "graph: N=%d E=%ld", N, E);
    REGT_CHECK_ARG(ws_bytes >= graph_workspace_bytes(E, N), "graph: workspace too small");
    Ws W;
    ws_layout(E, N, (char*)ws, &W);
    REGT_CHECK_HIP(hipMemsetAsync(W.flags, 0, 256, st));
    int rc = bucket(ei, E, N, /*key=dst*/ 1, /*extra*/ 1, W, rowptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_degree, dim3(nblk(N)), dim3(TPB), 0, st, rowptr, W.cnt, W.eid, w, W.loop_eid, 1, N, W.dis, W.loopw, W.flags);
    hipLaunchKernelGGL(k_emit_gcn, dim3(nblk(N)), dim3(TPB), 0, st, ei, E, w, rowptr, W.cnt, W.eid, W.dis, W.loopw, N, col, val);
    REGT_CHECK_LAUNCH();
    REGT_CHECK_HIP(hipMemcpyAsync(flags_out_dev, W.flags, 4, hipMemcpyDeviceToDevice, st));
    return REGT_OK;
}

// D^-1/2 of GCNConv's normalisation (in-degree sums in edge order + the self loop, exactly as graph_gcn_csr forms them): what a
// region shard publishes for its own nodes so that the ranks that read them as halo sources can finish their A_hat rows.
int graph_gcn_dis(const int64_t* ei, const float* w, long E, int N, float* dis_out, int* flags_out_dev, void* ws, size_t ws_bytes,
                  hipStream_t st) {
    REGT_CHECK_ARG(N > 0 && E >= 0, "graph: N=%d E=%ld", N, E);
    REGT_CHECK_ARG(ws_bytes >= graph_workspace_bytes(E, N), "graph: workspace too small");
    Ws W;
    ws_layout(E, N, (char*)ws, &W);
    REGT_CHECK_HIP(hipMemsetAsync(W.flags, 0, 256, st));
    int rc = bucket(ei, E, N, /*key=dst*/ 1, /*extra*/ 1, W, W.ptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_degree, dim3(nblk(N)), dim3(TPB), 0, st, W.ptr, W.cnt, W.eid, w, W.loop_eid, 1, N, dis_out, W.loopw, W.flags);
    REGT_CHECK_LAUNCH();
    REGT_CHECK_HIP(hipMemcpyAsync(flags_out_dev, W.flags, 4, hipMemcpyDeviceToDevice, st));
    return REGT_OK;
}

int graph_cheb_edge_weights(const int64_t* ei, const float* w, long E, int N, float* out_w, int* flags_out_dev,
                            void* ws, size_t ws_bytes, hipStream_t st) {
    REGT_CHECK_ARG(N > 0 && E >= 0, "graph: N=%d E=%ld", N, E);
    REGT_CHECK_ARG(ws_bytes >= graph_workspace_bytes(E, N), "graph: workspace too small");
    Ws W;
    ws_layout(E, N, (char*)ws, &W);
    REGT_CHECK_HIP(hipMemsetAsync(W.flags, 0, 256, st));
    int rc = bucket(ei, E, N, /*key=src*/ 0, 0, W, W.ptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_degree, dim3(nblk(N)), dim3(TPB), 0, st, W.ptr, W.cnt, W.eid, w, W.loop_eid, 0, N, W.dis, W.loopw, W.flags);
    if (E > 0) hipLaunchKernelGGL(k_cheb_edge, dim3(nblk(E)), dim3(TPB), 0, st, ei, E, N, w, W.dis, out_w);
    REGT_CHECK_LAUNCH();
    REGT_CHECK_HIP(hipMemcpyAsync(flags_out_dev, W.flags, 4, hipMemcpyDeviceToDevice, st));
    return REGT_OK;
}

int graph_raw_csr(const int64_t* ei, const float* w, long E, int N, int* rowptr, int* col, float* val,
                  int* flags_out_dev, void* ws, size_t ws_bytes, hipStream_t st) {
    REGT_CHECK_ARG(N > 0 && E >= 0, "graph: N=%d E=%ld", N, E);
    REGT_CHECK_ARG(ws_bytes >= graph_workspace_bytes(E, N), "graph: workspace too small");
    Ws W;
    ws_layout(E, N, (char*)ws, &W);
    REGT_CHECK_HIP(hipMemsetAsync(W.flags, 0, 256, st));
    int rc = bucket(ei, E, N, 1, 0, W, rowptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_emit_raw, dim3(nblk(N)), dim3(TPB), 0, st, ei, E, w, rowptr, W.cnt, W.eid, N, col, val);
    REGT_CHECK_LAUNCH();
    REGT_CHECK_HIP(hipMemcpyAsync(flags_out_dev, W.flags, 4, hipMemcpyDeviceToDevice, st));
    return REGT_OK;
}

int graph_mean_csr(const int64_t* ei, long E, int N, int* rowptr, int* col, float* val, int* flags_out_dev, void* ws,
                   size_t ws_bytes, hipStream_t st) {
    REGT_CHECK_ARG(N > 0 && E >= 0, "graph: N=%d E=%ld", N, E);
    REGT_CHECK_ARG(ws_bytes >= graph_workspace_bytes(E, N), "graph: workspace too small");
    Ws W;
    ws_layout(E, N, (char*)ws, &W);
    REGT_CHECK_HIP(hipMemsetAsync(W.flags, 0, 256, st));
    int rc = bucket(ei, E, N, 1, 0, W, rowptr, st, /*keep_loops=*/1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_emit_mean, dim3(nblk(N)), dim3(TPB), 0, st, ei, E, rowptr, W.cnt, W.eid, N, col, val);
    REGT_CHECK_LAUNCH();
    REGT_CHECK_HIP(hipMemcpyAsync(flags_out_dev, W.flags, 4, hipMemcpyDeviceToDevice, st));
    return REGT_OK;
}

int graph_fingerprint(const int64_t* ei, const float* w, long E, unsigned long long* out_dev, hipStream_t st) {
    REGT_CHECK_HIP(hipMemsetAsync(out_dev, 0, 8, st));
    if (E > 0) hipLaunchKernelGGL(k_fingerprint, dim3(nblk(E) > 1024 ? 1024 : nblk(E)), dim3(TPB), 0, st, ei, w, E, out_dev);
    REGT_CHECK_LAUNCH();
    return REGT_OK;
}

}  // namespace regt
