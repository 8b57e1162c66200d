// capi.cpp -- the extern "C" surface declared in include/cge_hip.h.
#include <algorithm>
#include <cmath>
#include <future>
#include <thread>
#include <unordered_set>

#include "common.hpp"
#include "../../include/cge_hip_testing.h"

void k_gather_i32(cge_ctx *c, const i32 *arr, const i32 *idx, i64 S, i32 *out);

#define CGE_TRY(ctx) try {
#define CGE_CATCH(ctx)                                             \
    }                                                              \
    catch (const CgeError &e) {                                    \
        if (ctx) (ctx)->err = e.msg;                               \
        return e.code;                                             \
    }                                                              \
    catch (const std::bad_alloc &) {                               \
        if (ctx) (ctx)->err = "host allocation failed";            \
        return CGE_E_OOM;                                          \
    }                                                              \
    catch (const std::exception &e) {                              \
        if (ctx) (ctx)->err = e.what();                            \
        return CGE_E_ARG;                                          \
    }                                                              \
    return CGE_OK;

static void flush_timers(cge_ctx *c) {
    for (auto &kv : c->timers) {
        for (auto &pr : kv.second.pending) {
            float ms = 0.f;
            (void)hipEventSynchronize(pr.second);
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.total_ms += ms;
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

// Pageable host memory -> device through two pinned staging buffers: the host workers convert / copy chunk k into one
// buffer while chunk k-1 is on the wire from the other (a plain hipMemcpy from pageable memory is a single-threaded
// bounce copy).  fill(dst, e0, e1) writes elements [e0, e1) of the output into `dst` (e1 - e0 <= chunk) and may be
// called from several threads on disjoint sub-ranges.
template <typename T, typename F, typename A>
static void staged_upload_chunks(cge_ctx *c, T *dev, size_t total, size_t chunk, size_t dev_ring, F fill, A after) {
    for (int b = 0; b < 2; b++) c->stage[b].ensure(CGE_STAGE_BYTES);
    const int nt = std::max(1, std::min(c->n_threads, 8)); // (more fill threads than that slow the link down: profiles/r05_microbench_upload.txt)
    for (size_t off = 0, k = 0; off < total; off += chunk, k++) {
        const int b = (int)(k & 1);
        if (k >= 2) HIP_CHECK(hipEventSynchronize(c->stage_ev[b])); // the copy that last read this buffer is done
        const size_t len = std::min(chunk, total - off);
        T *dst = (T *)c->stage[b].p;
        const size_t per = (len + nt - 1) / nt;
        const std::function<void(i64)> job = [&](i64 t) {
            const size_t a = std::min(len, (size_t)t * per), e = std::min(len, a + per);
            if (e > a) fill(dst + a, off + a, off + e);
        };
        c->pool->run(nt, job);
        T *where = dev_ring ? dev + (k % dev_ring) * chunk : dev + off; // (a ring on the device: `after` consumes the chunk in stream order)
        HIP_CHECK(hipMemcpyAsync(where, dst, sizeof(T) * len, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipEventRecord(c->stage_ev[b], c->stream));
        after(where, off, len);
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
template <typename T, typename F>
static void staged_upload(cge_ctx *c, T *dev, size_t total, F fill) {
    // at least ~8 chunks, so that the fill of one overlaps the copy of the one before it (a 40 MB column of edge ids in one
    // 64 MiB chunk would be filled, then copied), of at least 2 MiB, at most a staging buffer
    const size_t cap = CGE_STAGE_BYTES / sizeof(T), lo = ((size_t)2 << 20) / sizeof(T);
    const size_t chunk = std::min(cap, std::max(lo, (total + 7) / 8));
    staged_upload_chunks<T>(c, dev, total, chunk, 0, fill, [](T *, size_t, size_t) {});
}

// host mirror of the row-major embedding: only the generic round-based rss path (ties at the maximum of z, NaNs) and the
// exact unique-row count read it, so it is fetched on first demand instead of at every upload (1 GB at the headline)
void cge_ensure_host_embedding(cge_ctx *c) {
    const size_t need = (size_t)lm_rows(c) * (size_t)c->d; // (option shard_rows: this rank's rows, local ids)
    if (c->h_Xr.size() == need) return;
    if (!c->Xr.p || need == 0) CGE_THROW(CGE_E_ARG, "embedding not resident");
    c->h_Xr.resize(need);
    HIP_CHECK(hipMemcpyAsync(c->h_Xr.data(), c->Xr.p, sizeof(double) * need, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

extern "C" {

int cge_abi_version(void) { return CGE_ABI_VERSION; }

int cge_create(cge_ctx **out, int device, void *stream) {
    if (!out) return CGE_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CGE_E_HIP; // no GPU: fail loudly, no fallback
    if (device < 0 || device >= ndev) return CGE_E_ARG;
    cge_ctx *c = new (std::nothrow) cge_ctx();
    if (!c) return CGE_E_OOM;
    try {
        HIP_CHECK(hipSetDevice(device));
        c->device = device;
        if (stream) {
            c->stream = (hipStream_t)stream;
        } else {
            HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
        HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIP_CHECK(hipEventCreateWithFlags(&c->sweep_ev[i], hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIP_CHECK(hipEventCreateWithFlags(&c->tab_ev[i], hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIP_CHECK(hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
        unsigned hc = std::thread::hardware_concurrency();
        c->n_threads = (int)std::max(1u, std::min(hc ? hc : 8u, 16u));
        c->pool = new ThreadPool(c->n_threads - 1);
        // the pinned staging buffers of the uploads: made here, once per context, not inside the first cge_set_graph (pinning
        // 128 MiB is ~10 ms of page work that has nothing to do with any graph)
        for (int b = 0; b < 2; b++) c->stage[b].ensure(CGE_STAGE_BYTES);
        { // ... and the copy path itself walked once (the first host-to-device copy of a process sets up the DMA queues)
            DevBuf<unsigned char> warm;
            warm.ensure((size_t)1 << 20);
            for (int b = 0; b < 2; b++) {
                memset(c->stage[b].p, 0, (size_t)1 << 20);
                HIP_CHECK(hipMemcpyAsync(warm.p, c->stage[b].p, (size_t)1 << 20, hipMemcpyHostToDevice, c->stream));
            }
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        if (const char *nap = getenv("CGE_FIT_TEST_DELAY")) { // stress runs of whole suites: option fit_persistent_test_delay for
            // every context.  A testing knob in a production path: clamped to 2000 naps (~6 ms, far below the 1 s hand-off
            // deadline, so it can never force the time-out path) and announced once per process.
            c->opt_fit_test_delay = std::max(0, std::min(atoi(nap), 2000));
            static std::atomic<bool> told{false};
            if (c->opt_fit_test_delay > 0 && !told.exchange(true))
                fprintf(stderr, "cge: CGE_FIT_TEST_DELAY=%d is set: every persistent fit starts its tile waves late (testing knob)\n",
                        c->opt_fit_test_delay);
        }
    } catch (const CgeError &e) {
        delete c;
        return e.code;
    }
    *out = c;
    return CGE_OK;
}

void cge_destroy(cge_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    flush_timers(c);
    (void)cge_comm_finalize(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->event_pool.clear();
    if (c->copy_ev) (void)hipEventDestroy(c->copy_ev);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->samp_ev) (void)hipEventDestroy(c->samp_ev);
    for (int i = 0; i < 2; i++) {
        if (c->sweep_ev[i]) (void)hipEventDestroy(c->sweep_ev[i]);
        if (c->tab_ev[i]) (void)hipEventDestroy(c->tab_ev[i]);
    }
    for (int i = 0; i < 2; i++)
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
    delete c->pool;
    c->pool = nullptr;
    delete c;
}

const char *cge_last_error(const cge_ctx *c) { return c ? c->err.c_str() : "null context"; }

int cge_set_host_threads(cge_ctx *c, int n) {
    if (!c || n < 1) return CGE_E_ARG;
    c->n_threads = n;
    delete c->pool;
    c->pool = new ThreadPool(n - 1);
    return CGE_OK;
}

int cge_set_collectives(cge_ctx *c, const cge_collectives *coll) {
    if (!c) return CGE_E_ARG;
    if (!coll || !coll->allreduce_f64 || coll->world <= 1) {
        c->has_coll = false;
        c->coll_ext = cge_collectives_ext{};
        return CGE_OK;
    }
    c->coll = *coll;
    c->coll_ext = cge_collectives_ext{};
    c->has_coll = true;
    return CGE_OK;
}
int cge_set_collectives_ext(cge_ctx *c, const cge_collectives_ext *ext) {
    if (!c) return CGE_E_ARG;
    c->coll_ext = ext ? *ext : cge_collectives_ext{};
    return CGE_OK;
}

int cge_exchange_buffer(cge_ctx *c, int64_t min_doubles, void **dev_ptr, int64_t *cap) {
    if (!c) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if ((size_t)min_doubles > c->xown.n || !c->xown.p) c->xown.alloc_exact((size_t)std::max<i64>(min_doubles, 1024));
    c->xptr = c->xown.p;
    c->xcap = c->xown.n;
    if (dev_ptr) *dev_ptr = c->xptr;
    if (cap) *cap = (int64_t)c->xcap;
    CGE_CATCH(c)
}

int cge_set_exchange_buffer(cge_ctx *c, void *dev_ptr, int64_t cap) {
    if (!c || !dev_ptr || cap < 1) return CGE_E_ARG;
    c->xptr = (double *)dev_ptr;
    c->xcap = (size_t)cap;
    return CGE_OK;
}

// ---- resident inputs ------------------------------------------------------------------------------
static void allreduce(cge_ctx *c, double *dev, i64 count, int op);
static double allreduce_scalar_max(cge_ctx *c, double v);
// N > 1 with option "shard_ingest": which rows of the caller's edge list / embedding this rank uploads
static bool ingest_sharded(const cge_ctx *c) { return c->opt_shard_ingest && c->has_coll && c->coll.world > 1; }

// ---- option "shard_rows": the embedding rows sharded by community (common.hpp) ---------------------------------------------
static bool rows_shard_wanted(const cge_ctx *c) { return c->opt_shard_rows && c->has_coll && c->coll.world > 1; }
static void rows_unshard(cge_ctx *c) {
    c->rows_sharded = false;
    c->n_loc = 0;
    c->h_loc2glob.clear(); c->h_glob2loc.clear(); c->comm_owner.clear(); c->h_vw_loc.clear();
    c->loc2glob.release(); c->glob2loc.release(); c->comm_loc.release(); c->vw_loc.release();
}
// local copies of the per-vertex tables the row passes read (weights, communities of this rank's rows)
static void rows_refresh_local_tables(cge_ctx *c) {
    if (!c->rows_sharded) return;
    const i64 nl = c->n_loc;
    if ((i64)c->h_vw.size() == c->n) {
        c->h_vw_loc.resize(nl);
        for (i64 i = 0; i < nl; i++) c->h_vw_loc[i] = c->h_vw[c->h_loc2glob[i]];
        c->vw_loc.alloc_exact(nl);
        HIP_CHECK(hipMemcpyAsync(c->vw_loc.p, c->h_vw_loc.data(), sizeof(double) * nl, hipMemcpyHostToDevice, c->stream));
    }
    std::vector<i32> cl(nl);
    for (i64 i = 0; i < nl; i++) cl[i] = c->h_comm[c->h_loc2glob[i]];
    c->comm_loc.alloc_exact(nl);
    HIP_CHECK(hipMemcpyAsync(c->comm_loc.p, cl.data(), sizeof(i32) * nl, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
// THE OWNERSHIP RULE (the same on every rank: it reads the replicated community vector only): communities by decreasing
// size (ties: lower id first), each to the rank with the fewest rows so far (ties: lower rank).  cge.jl_amd/dist.py
// restates it (community_owner) and tests/test_distributed_gloo.py holds the two together.
static void rows_assign_ownership(cge_ctx *c) {
    const i64 n = c->n, C = c->n_comm_max, W = c->coll.world;
    if ((i64)c->h_comm.size() != n || C <= 0)
        CGE_THROW(CGE_E_ARG, "option shard_rows: upload the communities (cge_set_vertex_data) before the embedding -- the rows are sharded by community");
    std::vector<i64> size(C, 0), ord(C), load(W, 0);
    for (i64 i = 0; i < n; i++) size[c->h_comm[i]]++;
    for (i64 q = 0; q < C; q++) ord[q] = q;
    std::stable_sort(ord.begin(), ord.end(), [&](i64 a, i64 b) { return size[a] > size[b]; });
    c->comm_owner.assign(C, 0);
    for (i64 q : ord) {
        const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        c->comm_owner[q] = r;
        load[r] += size[q];
    }
    const int me = c->coll.rank;
    c->h_glob2loc.assign(n, -1);
    c->h_loc2glob.clear();
    c->h_loc2glob.reserve(load[me]);
    for (i64 i = 0; i < n; i++)
        if (c->comm_owner[c->h_comm[i]] == me) {
            c->h_glob2loc[i] = (i32)c->h_loc2glob.size();
            c->h_loc2glob.push_back((i32)i);
        }
    c->n_loc = (i64)c->h_loc2glob.size();
    if (c->n_loc <= 0) CGE_THROW(CGE_E_ARG, "option shard_rows: fewer communities than ranks (rank %d would own no row)", me);
    c->loc2glob.alloc_exact(c->n_loc);
    c->glob2loc.alloc_exact(n);
    HIP_CHECK(hipMemcpyAsync(c->loc2glob.p, c->h_loc2glob.data(), sizeof(i32) * c->n_loc, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(c->glob2loc.p, c->h_glob2loc.data(), sizeof(i32) * n, hipMemcpyHostToDevice, c->stream));
    c->rows_sharded = true;
    rows_refresh_local_tables(c);
}

int cge_set_graph(cge_ctx *c, const int64_t *src, const int64_t *dst, const double *w, int64_t m, int64_t n) {
    if (!c || !src || !dst || m <= 0 || n <= 0 || n >= (1LL << 31)) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    // N > 1, option "shard_ingest": this rank uploads and keeps rows [e0, e1) of the list only (the edge passes are sums over
    // edges: every rank scatters what it holds and the all-reduce adds; the sampler's look-ups are exchanged, kernels_fit.hip).
    // Not for graphs small enough for the sampler to enumerate their non-edges on the host (wgcl_host.cpp).
    const bool shard = ingest_sharded(c) && (double)n * (double)(n - 1) > 33554432.0 && m >= c->coll.world;
    const i64 e0 = shard ? m * c->coll.rank / c->coll.world : 0, e1 = shard ? m * (c->coll.rank + 1) / c->coll.world : m;
    const i64 ml = e1 - e0;
    c->src.alloc_exact(ml);
    c->dst.alloc_exact(ml);
    // ids: validated and narrowed to 0-based int32 by the host workers on their way into the staging buffers
    std::atomic<i64> bad{-1};
    for (int col = 0; col < 2; col++) {
        const int64_t *h = (col ? dst : src) + e0;
        staged_upload<i32>(c, col ? c->dst.p : c->src.p, (size_t)ml, [&](i32 *o, size_t a0, size_t a1) {
            for (size_t e = a0; e < a1; e++) {
                const int64_t v = h[e];
                if (v < 1 || v > n) { i64 exp = -1; bad.compare_exchange_strong(exp, (i64)e); }
                o[e - a0] = (i32)(v - 1);
            }
        });
    }
    // (sharded: every rank must take the same exit -- the verdicts are exchanged before anybody throws)
    const bool any_bad = shard ? allreduce_scalar_max(c, bad.load() >= 0 ? 1.0 : 0.0) != 0.0 : bad.load() >= 0;
    if (any_bad) {
        c->src.release(); c->dst.release(); c->m = c->m_total = 0; // (the previous resident graph is gone: cge_hip.h says so)
        c->blocked_ready = false; c->be_nchunks = 0; c->lm_ready = false;
        if (bad.load() >= 0)
            CGE_THROW(CGE_E_ARG, "edge %lld has a vertex id outside 1..%lld", (long long)(e0 + bad.load()) + 1, (long long)n);
        CGE_THROW(CGE_E_ARG, "an edge held by another rank has a vertex id outside 1..%lld", (long long)n);
    }
    // weights: all ones (an unweighted list, src/auxilary.jl:105) => neither a device copy nor a host mirror is kept
    bool unit = true;
    if (w) {
        const int nt = std::max(1, c->n_threads);
        std::vector<char> nonunit(nt, 0);
        const i64 per = (ml + nt - 1) / nt;
        const std::function<void(i64)> job = [&](i64 t) {
            const i64 a = std::min<i64>(ml, t * per), e = std::min<i64>(ml, a + per);
            char f = 0;
            for (i64 k = a; k < e && !f; k++) f = w[e0 + k] != 1.0;
            nonunit[t] = f;
        };
        c->pool->run(nt, job);
        for (char f : nonunit) unit = unit && !f;
    }
    if (shard) unit = allreduce_scalar_max(c, unit ? 0.0 : 1.0) == 0.0;
    c->unit_weights = unit;
    c->h_w.clear();
    c->w.release();
    if (!unit) {
        c->h_w.assign(w + e0, w + e1); // mirror: weights of host-side sample draws
        c->w.alloc_exact(ml);
        staged_upload<double>(c, c->w.p, (size_t)ml, [&](double *o, size_t a0, size_t a1) { memcpy(o, w + e0 + a0, sizeof(double) * (a1 - a0)); });
    }
    c->m_total = m;
    c->e_first = e0;
    c->edges_sharded = shard;
    m = ml;
    c->m = m;
    if (c->n && c->n != n) { // another vertex set: nothing that was sized for the old one may survive (stale or short buffers)
        c->h_Xr.clear(); c->h_vw.clear(); c->h_comm.clear();
        c->Xr.release(); c->Xc.release(); c->rnorm.release(); c->vw.release(); c->comm.release(); c->comm16.release();
        c->d = 0;
        c->centred_ready = false;
        rows_unshard(c);
    }
    c->n = n;
    c->lm_ready = false;
    c->blocked_ready = false; // the blocked copy of the edge list is rebuilt by the first edge pass
    CGE_CATCH(c)
}

// what every form of embedding upload ends with: sizes, the global feature mean (the centre of the diameter kernels'
// operands).  The centred feature-major copy of the brute-force diameter kernel is built on first use (it is as large as
// the embedding and the pruned path never reads it).
static void embedding_resident(cge_ctx *c, i64 n, i64 d) {
    c->h_Xr.clear();
    c->h_Xr.shrink_to_fit();
    c->n = n;
    c->d = d;
    c->ldn = (n + 127) / 128 * 128;
    c->dpad = (d + 15) / 16 * 16;
    c->Xc.release();
    c->rnorm.release();
    c->gmean.alloc_exact((size_t)d);
    if (c->rows_sharded) { // column sums of the local rows, added over the ranks (every rank ends with the same bits), / n
        k_col_mean(c, c->Xr.p, c->n_loc, d, c->gmean.p, 1.0);
        allreduce(c, c->gmean.p, d, 0);
        k_scale_vector(c, c->gmean.p, d, 1.0 / (double)n);
    } else
        k_col_mean(c, c->Xr.p, n, d, c->gmean.p);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->centred_ready = false;
    c->lm_ready = false;
}
static void ensure_centred(cge_ctx *c) {
    if (c->centred_ready) return;
    if (c->rows_sharded) CGE_THROW(CGE_E_ARG, "the brute-force diameter over the whole embedding needs every row on one rank; the resident rows are sharded (option shard_rows)");
    if (!c->Xr.p || c->d <= 0) CGE_THROW(CGE_E_ARG, "diameter: embedding not resident");
    c->Xc.alloc_exact((size_t)c->ldn * c->dpad);
    c->rnorm.alloc_exact((size_t)c->ldn);
    k_gather_centre_fm(c, c->Xr.p, nullptr, c->gmean.p, c->Xc.p, c->rnorm.p, c->n, c->d, c->ldn, c->dpad);
    c->centred_ready = true;
}

int cge_set_embedding(cge_ctx *c, const double *X, int64_t n, int64_t d) {
    if (!c || !X || n <= 0 || d <= 0) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if (c->n && c->n != n) CGE_THROW(CGE_E_ASSERT, "No. rows in embedding and no. vertices in a graph differ.");
    DevBuf<double> col;
    if (rows_shard_wanted(c)) {
        // N > 1, option "shard_rows": this rank uploads and KEEPS the rows of its own communities only (n / world rows over
        // its own PCIe link, nothing over xGMI): the caller's column-major matrix is read as a gather of rows per column
        rows_assign_ownership(c);
        const i64 nl = c->n_loc;
        const i32 *l2g = c->h_loc2glob.data();
        col.alloc_exact((size_t)nl * d);
        staged_upload<double>(c, col.p, (size_t)nl * d, [&](double *o, size_t a0, size_t a1) {
            for (size_t e = a0; e < a1; e++) { // element e of the (nl x d, column-major) slice: column e / nl, local row e % nl
                const size_t k = e / (size_t)nl, i = e % (size_t)nl;
                o[e - a0] = X[k * (size_t)n + (size_t)l2g[i]];
            }
        });
        c->Xr.alloc_exact((size_t)nl * d);
        k_transpose_to_rowmajor(c, col.p, c->Xr.p, nl, d);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        col.release();
        embedding_resident(c, n, d);
        return CGE_OK;
    }
    rows_unshard(c);
    if (ingest_sharded(c)) {
        // N > 1, option "shard_ingest": every rank uploads n / world ROWS (a strided piece of each column of the caller's
        // column-major matrix) over its own PCIe link, transposes them into its place of Xr, and the pieces are all-gathered
        // device to device (xGMI) -- instead of world full uploads side by side.  Equal pieces of `per` rows (ncclAllGather):
        // Xr carries up to world - 1 rows of padding behind row n.
        const i64 W = c->coll.world, r = c->coll.rank, per = (n + W - 1) / W;
        const i64 r0 = std::min<i64>(n, per * r), r1 = std::min<i64>(n, r0 + per), nl = r1 - r0;
        c->Xr.alloc_exact((size_t)per * W * d);
        if (nl < per) HIP_CHECK(hipMemsetAsync(c->Xr.p + (size_t)(per * r + nl) * d, 0, sizeof(double) * (size_t)(per - nl) * d, c->stream));
        if (nl > 0) {
            col.alloc_exact((size_t)nl * d);
            staged_upload<double>(c, col.p, (size_t)nl * d, [&](double *o, size_t a0, size_t a1) {
                for (size_t e = a0; e < a1;) { // element e of the (nl x d, column-major) slice: column e / nl, row r0 + e % nl
                    const size_t k = e / (size_t)nl, i = e % (size_t)nl, run = std::min<size_t>(a1 - e, (size_t)nl - i);
                    memcpy(o + (e - a0), X + k * (size_t)n + (size_t)r0 + i, sizeof(double) * run);
                    e += run;
                }
            });
            k_transpose_to_rowmajor(c, col.p, c->Xr.p + (size_t)per * r * d, nl, d);
        }
        cge_allgather_dev(c, c->Xr.p, per * d);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        col.release();
        embedding_resident(c, n, d);
        return CGE_OK;
    }
    // The caller's column-major matrix goes up in chunks of whole columns (of row pieces of one column, when a column is
    // longer than a staging buffer); every chunk is transposed into its place of the row-major Xr right behind its copy, on the
    // stream, while the next chunk is on the wire: no n x d column-major device buffer, no separate transpose pass.
    c->Xr.alloc_exact((size_t)n * d);
    const size_t cap = CGE_STAGE_BYTES / sizeof(double);
    if ((size_t)n <= cap) {
        const size_t kc = std::min<size_t>((size_t)d, cap / (size_t)n), chunk = kc * (size_t)n; // whole columns per chunk
        col.alloc_exact(2 * chunk);
        staged_upload_chunks<double>(c, col.p, (size_t)n * d, chunk, 2,
                                     [&](double *o, size_t e0, size_t e1) { memcpy(o, X + e0, sizeof(double) * (e1 - e0)); },
                                     [&](double *piece, size_t off, size_t len) {
                                         k_transpose_piece(c, piece, c->Xr.p, n, (i64)(len / (size_t)n), 0, (i64)(off / (size_t)n), d);
                                     });
    } else { // one column in row pieces
        col.alloc_exact(2 * cap);
        for (i64 k = 0; k < d; k++)
            staged_upload_chunks<double>(c, col.p, (size_t)n, cap, 2,
                                         [&](double *o, size_t e0, size_t e1) { memcpy(o, X + (size_t)k * n + e0, sizeof(double) * (e1 - e0)); },
                                         [&](double *piece, size_t off, size_t len) { k_transpose_piece(c, piece, c->Xr.p, (i64)len, 1, (i64)off, k, d); });
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    col.release();
    embedding_resident(c, n, d);
    CGE_CATCH(c)
}

int cge_set_embedding_device(cge_ctx *c, const double *X_dev, int64_t n, int64_t d, int row_major) {
    if (!c || !X_dev || n <= 0 || d <= 0) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if (c->n && c->n != n) CGE_THROW(CGE_E_ASSERT, "No. rows in embedding and no. vertices in a graph differ.");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, X_dev) != hipSuccess || at.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        CGE_THROW(CGE_E_ARG, "set_embedding_device: the pointer is not device memory");
    }
    if (rows_shard_wanted(c)) { // option shard_rows: this rank's rows are gathered out of the caller's matrix
        rows_assign_ownership(c);
        c->Xr.alloc_exact((size_t)c->n_loc * d);
        k_gather_rows_f64(c, X_dev, n, d, row_major, c->loc2glob.p, c->n_loc, c->Xr.p);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        embedding_resident(c, n, d);
        return CGE_OK;
    }
    rows_unshard(c);
    c->Xr.alloc_exact((size_t)n * d);
    if (row_major)
        HIP_CHECK(hipMemcpyAsync(c->Xr.p, X_dev, sizeof(double) * (size_t)n * d, hipMemcpyDeviceToDevice, c->stream));
    else
        k_transpose_to_rowmajor(c, X_dev, c->Xr.p, n, d);
    HIP_CHECK(hipStreamSynchronize(c->stream)); // the caller may free or reuse its buffer on return
    embedding_resident(c, n, d);
    CGE_CATCH(c)
}

int cge_set_vertex_data(cge_ctx *c, const int64_t *comm, const double *vw, int64_t n) {
    if (!c || n <= 0) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if (c->n && c->n != n) CGE_THROW(CGE_E_ASSERT, "No. communities (%lld) differ from no. nodes (%lld)", (long long)n, (long long)c->n);
    c->n = n;
    if (comm && c->rows_sharded) {
        // the rows are sharded BY COMMUNITY: another community vector is another ownership -- the resident rows are dropped
        // (upload the embedding again after this call)
        bool same = (i64)c->h_comm.size() == n;
        for (i64 i = 0; same && i < n; i++) same = c->h_comm[i] == (i32)(comm[i] - 1);
        if (!same) {
            c->Xr.release(); c->h_Xr.clear();
            c->d = 0;
            rows_unshard(c);
        }
    }
    if (comm) {
        c->h_comm.resize(n);
        i64 cmax = 0;
        for (i64 i = 0; i < n; i++) {
            if (comm[i] < 1) CGE_THROW(CGE_E_ARG, "community ids must be 1-based");
            c->h_comm[i] = (i32)(comm[i] - 1);
            cmax = std::max<i64>(cmax, comm[i]);
        }
        c->n_comm_max = cmax;
        c->comm.alloc_exact(n);
        HIP_CHECK(hipMemcpyAsync(c->comm.p, c->h_comm.data(), sizeof(i32) * n, hipMemcpyHostToDevice, c->stream));
        c->comm16.release();
        if (cmax < 65536) {
            const i64 npad = (n + CGE_COMM16_PAD - 1) / CGE_COMM16_PAD * CGE_COMM16_PAD; // whole vertex blocks (edge pass)
            std::vector<unsigned short> c16(npad, 0);
            for (i64 i = 0; i < n; i++) c16[i] = (unsigned short)c->h_comm[i];
            c->comm16.alloc_exact(npad);
            HIP_CHECK(hipMemcpyAsync(c->comm16.p, c16.data(), sizeof(unsigned short) * npad, hipMemcpyHostToDevice, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream)); // c16 goes out of scope
        }
    }
    if (vw) {
        c->h_vw.assign(vw, vw + n);
        c->vw.alloc_exact(n);
        HIP_CHECK(hipMemcpyAsync(c->vw.p, c->h_vw.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    rows_refresh_local_tables(c);
    c->lm_ready = false;
    CGE_CATCH(c)
}

// ---- landmarks ------------------------------------------------------------------------------------
// `size(unique(embedding, dims=1), 1)` clamp (src/landmarks.jl:371-376).  Equal rows have equal
// hashes, so #distinct hashes <= #unique rows: once `land` distinct hashes are seen no clamp can
// apply; otherwise count exactly on the host mirror.
static i64 clamp_to_unique_rows(cge_ctx *c, i64 land, int *truncated) {
    const i64 n = c->n, d = c->d;
    *truncated = 0;
    if (land <= 1) return land;
    // `land` distinct hashes among a prefix of the rows already prove `land` distinct rows: hash 8*land rows first,
    // the whole matrix only when that prefix does not settle it.  The distinct hashes are counted on the device (a set
    // of atomicCAS slots): one 8-byte read-back instead of the hashes themselves and a host set.
    i64 done = 0;
    c->uniq_hash.ensure(n);
    if (c->rows_sharded) {
        // option shard_rows: every rank hashes its rows, the hashes are gathered by vertex id (8 bytes per vertex) and every
        // rank counts the distinct ones; only if that leaves the clamp open are the rows with a shared hash -- the only
        // candidates for equal rows -- gathered and compared bit for bit
        const i64 nl = c->n_loc;
        DevBuf<uint64_t> hl;
        hl.ensure(nl);
        k_row_hash(c, c->Xr.p, hl.p, nl, d);
        HIP_CHECK(hipMemsetAsync(c->uniq_hash.p, 0, sizeof(uint64_t) * n, c->stream));
        k_scatter_u64(c, hl.p, c->loc2glob.p, nl, c->uniq_hash.p);
        allreduce(c, reinterpret_cast<double *>(c->uniq_hash.p), n, 2);
        if (k_count_distinct(c, c->uniq_hash.p, n) >= land) return land;
        std::vector<uint64_t> hh(n);
        HIP_CHECK(hipMemcpyAsync(hh.data(), c->uniq_hash.p, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        std::vector<i64> ix(n);
        for (i64 i = 0; i < n; i++) ix[i] = i;
        std::sort(ix.begin(), ix.end(), [&](i64 a, i64 b) { return hh[a] < hh[b] || (hh[a] == hh[b] && a < b); });
        std::vector<i32> dup; // vertices whose hash is shared, grouped by hash
        std::vector<i64> run_off(1, 0);
        i64 uniq = 0;
        for (i64 a = 0; a < n;) {
            i64 b = a + 1;
            while (b < n && hh[ix[b]] == hh[ix[a]]) b++;
            if (b - a > 1) {
                for (i64 q = a; q < b; q++) dup.push_back((i32)ix[q]);
                run_off.push_back((i64)dup.size());
            } else
                uniq++;
            a = b;
        }
        if (!dup.empty()) {
            const i64 nd = (i64)dup.size();
            std::vector<i32> lidx(nd);
            for (i64 q = 0; q < nd; q++) lidx[q] = c->h_glob2loc[dup[q]];
            DevBuf<i32> didx;
            DevBuf<double> rows;
            didx.ensure(nd);
            rows.ensure((size_t)nd * d);
            HIP_CHECK(hipMemcpyAsync(didx.p, lidx.data(), sizeof(i32) * nd, hipMemcpyHostToDevice, c->stream));
            k_gather_rows_f64(c, c->Xr.p, nl, d, 1, didx.p, nd, rows.p);
            allreduce(c, rows.p, nd * d, 2);
            std::vector<double> hr((size_t)nd * d);
            HIP_CHECK(hipMemcpyAsync(hr.data(), rows.p, sizeof(double) * nd * d, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (size_t r = 0; r + 1 < run_off.size(); r++) {
                std::vector<i64> q(run_off[r + 1] - run_off[r]);
                for (size_t t = 0; t < q.size(); t++) q[t] = run_off[r] + (i64)t;
                std::sort(q.begin(), q.end(), [&](i64 a, i64 b) { return memcmp(&hr[a * d], &hr[b * d], sizeof(double) * d) < 0; });
                uniq++;
                for (size_t t = 1; t < q.size(); t++)
                    if (memcmp(&hr[q[t - 1] * d], &hr[q[t] * d], sizeof(double) * d) != 0) uniq++;
            }
        }
        if (land > uniq) {
            *truncated = 1;
            return uniq;
        }
        return land;
    }
    for (int pass = 0; pass < 2 && done < n; pass++) {
        const i64 upto = pass == 0 ? std::min<i64>(n, 8 * land) : n;
        k_row_hash(c, c->Xr.p + done * d, c->uniq_hash.p + done, upto - done, d);
        if (k_count_distinct(c, c->uniq_hash.p, upto) >= land) return land;
        done = upto;
    }
    // fewer distinct hashes than `land`: count bitwise-distinct rows exactly
    std::vector<i64> ix(n);
    for (i64 i = 0; i < n; i++) ix[i] = i;
    cge_ensure_host_embedding(c);
    const double *X = c->h_Xr.data();
    auto cmp = [&](i64 a, i64 b) { return memcmp(X + a * d, X + b * d, sizeof(double) * d) < 0; };
    std::sort(ix.begin(), ix.end(), cmp);
    i64 uniq = n > 0 ? 1 : 0;
    for (i64 i = 1; i < n; i++)
        if (memcmp(X + ix[i - 1] * d, X + ix[i] * d, sizeof(double) * d) != 0) uniq++;
    if (land > uniq) {
        *truncated = 1;
        return uniq;
    }
    return land;
}

static void allreduce(cge_ctx *c, double *dev, i64 count, int op) {
    if (!c->has_coll) return;
    if (c->rccl_comm) { // in-library RCCL: stream-ordered, in place, no host synchronisation
        cge_rccl_allreduce(c, dev, count, op);
        return;
    }
    // the hook works on the ctx exchange buffer (the host side wrapped that pointer once); a vector that lives elsewhere
    // and is longer than the buffer goes through it in pieces (an all-reduce is element-wise)
    if (!c->xptr || c->xcap == 0 || (dev == c->xptr && (size_t)count > c->xcap))
        CGE_THROW(CGE_E_COLLECTIVE, "exchange buffer too small: need %lld doubles, have %lld", (long long)count, (long long)c->xcap);
    for (i64 off = 0; off < count; off += (i64)c->xcap) {
        const i64 piece = std::min<i64>((i64)c->xcap, count - off);
        if (dev != c->xptr)
            HIP_CHECK(hipMemcpyAsync(c->xptr, dev + off, sizeof(double) * piece, hipMemcpyDeviceToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->coll.allreduce_f64(c->coll.user, c->xptr, piece, op) != 0) CGE_THROW(CGE_E_COLLECTIVE, "allreduce hook failed");
        c->stat_coll_calls++;
        c->stat_coll_bytes += 8 * piece;
        if (dev != c->xptr)
            HIP_CHECK(hipMemcpyAsync(dev + off, c->xptr, sizeof(double) * piece, hipMemcpyDeviceToDevice, c->stream));
    }
}

static double allreduce_scalar_max(cge_ctx *c, double v) {
    if (!c->has_coll) return v;
    if (!cge_exchange_fits(c, 1)) CGE_THROW(CGE_E_COLLECTIVE, "no exchange buffer set");
    HIP_CHECK(hipMemcpyAsync(c->xptr, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
    allreduce(c, c->xptr, 1, 1);
    HIP_CHECK(hipMemcpyAsync(&v, c->xptr, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return v;
}

static void build_landmark_index(cge_ctx *c, const std::vector<i32> &v2l0, i64 N);
// the resident inputs a landmark / score run reads: all present and all sized for the same vertex set
static void check_resident(cge_ctx *c, const char *who) {
    if (!c->Xr.p || !c->vw.p || !c->comm.p || !c->src.p)
        CGE_THROW(CGE_E_ARG, "%s: graph, embedding and vertex data must be resident (cge_set_graph / cge_set_embedding / "
                             "cge_set_vertex_data; a cge_wgcl call in exact mode replaces the resident graph)", who);
    const size_t n = (size_t)c->n;
    const size_t rows = (size_t)lm_rows(c);
    if (c->rows_sharded && (!c->vw_loc.p || !c->comm_loc.p))
        CGE_THROW(CGE_E_ARG, "%s: option shard_rows needs the vertex weights and communities resident (cge_set_vertex_data)", who);
    if (c->n <= 0 || c->d <= 0 || c->m <= 0 || c->Xr.n < rows * (size_t)c->d || c->vw.n < n || c->comm.n < n ||
        c->src.n < (size_t)c->m || c->dst.n < (size_t)c->m)
        CGE_THROW(CGE_E_ARG, "%s: resident inputs are inconsistent (n = %lld, d = %lld, m = %lld): upload them again", who,
                  (long long)c->n, (long long)c->d, (long long)c->m);
}
// per-edge scatter of the resident graph into the landmark-pair matrix (and its positive-entry count)
// the row block of the landmark-pair matrix this rank ends up with after the reduce-scatter: equal blocks of `per` rows
static inline i64 wedge_rows_per_rank(const cge_ctx *c, i64 N) { return c->has_coll ? (N + c->coll.world - 1) / c->coll.world : N; }
static void scatter_wedges(cge_ctx *c, int directed) {
    const i64 N = c->N;
    hipStream_t st = c->stream;
    const i64 per = wedge_rows_per_rank(c, N), Npad = c->has_coll ? per * c->coll.world : N; // (padding rows behind row N: zeros)
    c->wedges.ensure((size_t)Npad * N);
    c->wedges_block_only = false;
    DevBuf<i64> &cnt = c->wed_cnt;
    cnt.ensure(1);
    // this rank's share of the edges: of a replicated list its slice of the chunks; of a sharded list all that it holds
    const int rank = (c->has_coll && !c->edges_sharded) ? c->coll.rank : 0, world = (c->has_coll && !c->edges_sharded) ? c->coll.world : 1;
    // the tiled two-pass form on the blocked copy of the edge list (kernels_scatter.hip): the tiles are written whole, the
    // positive entries counted on the way (one rank) -- else the gather + atomics kernel into a zeroed matrix
    bool tiled = (c->blocked_ready || (k_blocked_edges_possible(c) && k_build_blocked_edges(c))) &&
                 k_wedge_scatter_blocked(c, c->v2l.p, N, c->be_nchunks * rank / world, c->be_nchunks * (rank + 1) / world, directed,
                                         c->wedges.p, cnt.p);
    if (!tiled) {
        HIP_CHECK(hipMemsetAsync(c->wedges.p, 0, sizeof(double) * N * N, st));
        const i64 e0 = c->m * rank / world, e1 = c->m * (rank + 1) / world;
        k_edge_scatter(c, c->src.p, c->dst.p, c->unit_weights ? nullptr : c->w.p, e0, e1, c->v2l.p, c->comm.p, N,
                       c->n_comm_max, directed, c->wedges.p, nullptr);
    }
    if (c->has_coll) {
        // every rank has summed ITS edges into a full N x N matrix; the sums over the ranks go out BY ROW BLOCK (SURVEY 8(e):
        // reduce-scatter, not all-reduce): rank r ends with rows [per r, per (r + 1)).  What reads the matrix afterwards works
        // on row blocks (the count below, the directed score's degrees); landmarks_fetch all-gathers the blocks when the host
        // asks for the edge list.  Through the hook (gloo tests) or a librccl without the symbol: an all-reduce.
        if (Npad > N) HIP_CHECK(hipMemsetAsync(c->wedges.p + (size_t)N * N, 0, sizeof(double) * (size_t)(Npad - N) * N, st));
        // Option "wedges_reduce_scatter" (default 0: the all-reduce, after which every consumer -- cge_landmarks_fetch on ONE
        // rank included -- is local).  With it the matrix goes out by row blocks; the consumers then work on blocks, and a
        // fetch of the edge list is COLLECTIVE (every rank must call it: the blocks are all-gathered).
        bool by_blocks = false;
        if (c->opt_wedges_rs) {
            if (c->rccl_comm) by_blocks = cge_rccl_reduce_scatter(c, c->wedges.p, per * N);
            else if (c->coll_ext.reduce_scatter_f64 && c->xptr && (size_t)(per * N * c->coll.world) <= c->xcap) {
                const i64 tot = per * N * c->coll.world;
                HIP_CHECK(hipMemcpyAsync(c->xptr, c->wedges.p, sizeof(double) * tot, hipMemcpyDeviceToDevice, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (c->coll_ext.reduce_scatter_f64(c->coll.user, c->xptr, per * N) != 0) CGE_THROW(CGE_E_COLLECTIVE, "reduce-scatter hook failed");
                c->stat_coll_calls++;
                c->stat_coll_bytes += 8 * tot;
                HIP_CHECK(hipMemcpyAsync(c->wedges.p, c->xptr, sizeof(double) * tot, hipMemcpyDeviceToDevice, st));
                by_blocks = true;
            }
        }
        if (by_blocks) c->wedges_block_only = true;
        else allreduce(c, c->wedges.p, N * N, 0);
        const i64 r0 = std::min<i64>(N, per * c->coll.rank), r1 = std::min<i64>(N, r0 + per);
        k_compact_count(c, c->wedges.p, N, directed, cnt.p, r0, r1);
        allreduce(c, reinterpret_cast<double *>(cnt.p), 1, 2); // (integer sum of the ranks' counts)
    } else if (!tiled)
        k_compact_count(c, c->wedges.p, N, directed, cnt.p);
    HIP_CHECK(hipMemcpyAsync(&c->n_ledges, cnt.p, sizeof(i64), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    c->wedges_ready = true;
}
// the whole matrix on this rank (landmarks_fetch): the row blocks of a reduce-scattered matrix are all-gathered in place
static void wedges_whole(cge_ctx *c) {
    if (!c->wedges_block_only) return;
    cge_allgather_dev(c, c->wedges.p, wedge_rows_per_rank(c, c->N) * c->N);
    c->wedges_block_only = false;
}

// vect_C of the resident graph (src/divergence.jl:59-63 / :337-345 on the original edges): the blocked two-pass form
// (kernels_scatter.hip) where it applies, else the gather + atomics kernel; this rank's share, then the all-reduce
static void scatter_vectC_resident(cge_ctx *c, i64 C, int directed, double *vectC) {
    const i64 vlen = directed ? C * C : packed_len(C);
    const int rank = (c->has_coll && !c->edges_sharded) ? c->coll.rank : 0, world = (c->has_coll && !c->edges_sharded) ? c->coll.world : 1;
    bool done = false;
    if (k_edge_scatter_blocked_applies(c, C) && (c->blocked_ready || k_build_blocked_edges(c))) {
        k_edge_scatter_blocked(c, c->be_nchunks * rank / world, c->be_nchunks * (rank + 1) / world, C, directed, vectC);
        done = true;
    } else if (C > 1 && C <= 16384 && (c->blocked_ready || (k_blocked_edges_possible(c) && k_build_blocked_edges(c)))) {
        // beyond the 2048 row counters of the row-bucketed form: the community pairs as a dense C x C matrix through the
        // TILED two-pass form of the landmark-pair matrix (tiles of rows in LDS, written whole), then packed
        DevBuf<i64> &cnt = c->wed_cnt;
        cnt.ensure(1);
        double *dense = vectC;
        if (!directed) { c->cc_dense.ensure((size_t)C * C); dense = c->cc_dense.p; }
        if (k_wedge_scatter_blocked(c, c->comm.p, C, c->be_nchunks * rank / world, c->be_nchunks * (rank + 1) / world, directed,
                                    dense, cnt.p, "edge_scatter")) {
            if (!directed) k_pack_upper(c, dense, C, vectC);
            done = true;
        }
    }
    if (!done) {
        HIP_CHECK(hipMemsetAsync(vectC, 0, sizeof(double) * vlen, c->stream));
        k_edge_scatter(c, c->src.p, c->dst.p, c->unit_weights ? nullptr : c->w.p, c->m * rank / world, c->m * (rank + 1) / world,
                       nullptr, c->comm.p, 1, C, directed, nullptr, vectC);
    }
    if (c->has_coll) allreduce(c, vectC, vlen, 0);
}

static void landmarks_run_impl(cge_ctx *c, const i64 *cl_flat, const i64 *cl_off, i64 ncl, i64 land, i64 forced,
                               int method, int directed, bool need_wedges) {
    check_resident(c, "landmarks");
    if (method < 0 || method > 3) CGE_THROW(CGE_E_ARG, "unknown split method %d", method);
    const i64 d = c->d;
    hipStream_t st = c->stream;
    double t0 = now_ms();
    land = clamp_to_unique_rows(c, land, &c->lm_truncated);
    c->phases.ms["lm_unique"] = now_ms() - t0;
    if (c->after_unique) { // (cge_score: the sample draws go in here -- the host now sets up runsplit for a few hundred microseconds)
        std::function<void()> f;
        f.swap(c->after_unique);
        f();
    }
    std::vector<i64> gid;
    host_runsplit(c, cl_flat, cl_off, ncl, land, forced, method, gid, true); // leaves v2l and the landmark index on the device
    HIP_CHECK(hipStreamSynchronize(st));
    c->phases.ms["landmarks"] = now_ms() - t0;
    t0 = now_ms();
    const i64 N = (i64)c->h_mem_off.size() - 1; // every group is non-empty
    c->N = N;
    c->h_v2l.clear(); // v_to_l (:379) is read back from the device by landmarks_fetch
    c->lemb.ensure((size_t)N * d);
    c->lweight.ensure(N);
    c->dii.ensure(N);
    c->lcomm.ensure(N);
    {
        ScopedKernelTimer tm(c, "landmark_aggregate");
        k_landmark_aggregate(c, c->Xr.p, lm_vw(c), lm_comm(c), c->lm_memoff.p, c->lm_mem.p, N, d, c->lemb.p, c->lweight.p, c->dii.p,
                             c->lcomm.p);
    }
    if (c->rows_sharded) {
        // option shard_rows: a landmark's members live on one rank, which has just aggregated it (the others wrote zeros for
        // it); centroids, weights, d_ii and communities of ALL landmarks on every rank by one gather (N (d + 3) words)
        DevBuf<double> &X = c->samp_xchg;
        const i64 words = N * (d + 3);
        X.ensure(words);
        k_pack_landmarks(c, c->lemb.p, c->lweight.p, c->dii.p, c->lcomm.p, N, d, X.p, 0);
        allreduce(c, X.p, words, 2);
        k_pack_landmarks(c, c->lemb.p, c->lweight.p, c->dii.p, c->lcomm.p, N, d, X.p, 1);
    }
    HIP_CHECK(hipStreamSynchronize(st));
    c->phases.ms["aggregate"] = now_ms() - t0;
    t0 = now_ms();
    // per-edge scatter.  vect_C (C x C cluster pairs, from the original edges: every landmark lies in one
    // community, so this equals the reference's sum over landmark edges, src/divergence.jl:59-63) is what the
    // score needs; the N x N landmark-pair matrix (src/landmarks.jl:433-451) only feeds landmarks_fetch.
    const i64 C = c->n_comm_max;
    const i64 vlen = directed ? C * C : packed_len(C);
    c->vectC.ensure(vlen);
    scatter_vectC_resident(c, C, directed, c->vectC.p);
    c->lm_directed = directed;
    c->wedges_ready = false;
    c->n_ledges = -1;
    if (need_wedges) scatter_wedges(c, directed);
    HIP_CHECK(hipStreamSynchronize(st));
    c->phases.ms["scatter"] = now_ms() - t0;
    c->lm_ready = true;
}

int cge_landmarks_run(cge_ctx *c, const int64_t *cl_flat, const int64_t *cl_off, int64_t ncl, int64_t land,
                      int64_t forced, int method, int directed, int64_t *N_out, int64_t *n_ledges_out, int *truncated) {
    if (!c || !cl_flat || !cl_off) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    landmarks_run_impl(c, cl_flat, cl_off, ncl, land, forced, method, directed, true);
    if (N_out) *N_out = c->N;
    if (n_ledges_out) *n_ledges_out = c->n_ledges;
    if (truncated) *truncated = c->lm_truncated;
    CGE_CATCH(c)
}

int cge_landmarks_info(cge_ctx *c, int64_t *N_out, int64_t *n_ledges_out, int *truncated) {
    if (!c) return CGE_E_ARG;
    if (!c->lm_ready) {
        c->err = "landmarks_info: run cge_landmarks_run first";
        return CGE_E_ARG;
    }
    if (!c->wedges_ready) { // the score path skips the landmark-pair matrix; build it on first demand
        try {
            HIP_CHECK(hipSetDevice(c->device));
            scatter_wedges(c, c->lm_directed);
        } catch (const CgeError &e) {
            c->err = e.msg;
            return e.code;
        }
    }
    if (N_out) *N_out = c->N;
    if (n_ledges_out) *n_ledges_out = c->n_ledges;
    if (truncated) *truncated = c->lm_truncated;
    return CGE_OK;
}

int cge_landmarks_fetch(cge_ctx *c, double *dii, double *embed, int64_t *cluster, int64_t *ledges, double *lw_e,
                        double *lweight, int64_t *v_to_l) {
    if (!c) return CGE_E_ARG;
    CGE_TRY(c)
    if (!c->lm_ready) CGE_THROW(CGE_E_ARG, "landmarks_fetch: run cge_landmarks_run first");
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->wedges_ready) scatter_wedges(c, c->lm_directed);
    if (ledges || lw_e) wedges_whole(c);
    const i64 N = c->N, d = c->d, n = c->n;
    hipStream_t st = c->stream;
    if (dii) HIP_CHECK(hipMemcpyAsync(dii, c->dii.p, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    if (lweight) HIP_CHECK(hipMemcpyAsync(lweight, c->lweight.p, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    std::vector<double> rm;
    std::vector<i32> lc;
    if (embed) {
        rm.resize((size_t)N * d);
        HIP_CHECK(hipMemcpyAsync(rm.data(), c->lemb.p, sizeof(double) * N * d, hipMemcpyDeviceToHost, st));
    }
    if (cluster) {
        lc.resize(N);
        HIP_CHECK(hipMemcpyAsync(lc.data(), c->lcomm.p, sizeof(i32) * N, hipMemcpyDeviceToHost, st));
    }
    std::vector<double> we;
    if (ledges || lw_e) {
        we.resize((size_t)N * N);
        HIP_CHECK(hipMemcpyAsync(we.data(), c->wedges.p, sizeof(double) * N * N, hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    if (embed)
        for (i64 l = 0; l < N; l++)
            for (i64 k = 0; k < d; k++) embed[l + k * N] = rm[l * d + k]; // column-major out
    if (cluster)
        for (i64 l = 0; l < N; l++) cluster[l] = lc[l] + 1;
    if (ledges || lw_e) { // rows in idx order / N*(i-1)+j order, w > 0 only (src/landmarks.jl:441-463)
        const i64 ne = c->n_ledges;
        i64 k = 0;
        for (i64 a = 0; a < N; a++)
            for (i64 b = c->lm_directed ? 0 : a; b < N; b++) {
                const double wv = we[a * N + b];
                if (wv > 0) {
                    if (k >= ne) CGE_THROW(CGE_E_ASSERT, "landmark edge count changed between run and fetch");
                    if (ledges) { ledges[k] = a + 1; ledges[k + ne] = b + 1; }
                    if (lw_e) lw_e[k] = wv;
                    k++;
                }
            }
    }
    if (v_to_l) { // 1-based landmark of every vertex (:379), from the device copy the score path works on
        std::vector<i32> v0(n);
        HIP_CHECK(hipMemcpyAsync(v0.data(), c->v2l.p, sizeof(i32) * n, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        for (i64 i = 0; i < n; i++) v_to_l[i] = (i64)v0[i] + 1;
    }
    CGE_CATCH(c)
}

int cge_runsplit(cge_ctx *c, const int64_t *cl_flat, const int64_t *cl_off, int64_t ncl, int64_t nland,
                 int64_t forced, int method, int64_t *group_ids) {
    if (!c || !group_ids) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->Xr.p || !c->vw.p) CGE_THROW(CGE_E_ARG, "runsplit: embedding and vertex weights must be resident");
    std::vector<i64> gid;
    host_runsplit(c, cl_flat, cl_off, ncl, nland, forced, method, gid);
    memcpy(group_ids, gid.data(), sizeof(i64) * c->n);
    CGE_CATCH(c)
}

// ---- samples --------------------------------------------------------------------------------------
int cge_draw_samples(cge_ctx *c, int64_t seed, int64_t stream_id, int64_t S, int directed, int64_t *pos_idx,
                     int64_t *neg_i, int64_t *neg_j) {
    if (!c || S <= 0 || !pos_idx || !neg_i || !neg_j) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->src.p) CGE_THROW(CGE_E_ARG, "draw_samples: no resident graph");
    if (c->edges_sharded) CGE_THROW(CGE_E_ARG, "draw_samples: the resident edge list is sharded over the ranks (option shard_ingest); cge_score draws on the device");
    host_draw_samples(c, seed, stream_id, S, directed, pos_idx, neg_i, neg_j);
    CGE_CATCH(c)
}

// whether make_samples can be started ahead of the rest of a score: one seeded set, drawn on the device from a local edge list
static bool samples_can_start_early(cge_ctx *c, i64 seed, bool exact_directed);
// phase 0: everything; 1: enqueue the draws and the first rejection round, no synchronisation (samples_can_start_early only);
// 2: the rest of a draw begun with phase 1
static void make_samples(cge_ctx *c, i64 seed, i64 S, int directed, bool exact_directed, SampleSet &smp, int phase = 0) {
    if (phase == 2) {
        k_draw_samples_finish(c);
        return;
    }
    // seeded: one set reused at every alpha (Random.seed! before each draw, src/divergence.jl:184,193);
    // unseeded: a fresh set per alpha, keyed by an arbitrary fixed base seed and the alpha index
    const i64 n_alpha = 40;
    smp.S = S;
    smp.n_sets = (seed != -1) ? 1 : n_alpha;
    const i64 base = (seed != -1) ? seed : 0x5eedc0de;
    if (sampler_uses_device(c)) { // large resident graph: drawn, rejected and kept on the device (the same stream of draws)
        smp.on_device = true;
        smp.d_pos.ensure(smp.n_sets * S); smp.d_ni.ensure(smp.n_sets * S); smp.d_nj.ensure(smp.n_sets * S);
        if (phase == 1) { // (one set, no second draw: samples_can_start_early)
            k_draw_samples_begin(c, base, 0, S, directed, smp.d_pos.p, smp.d_ni.p, smp.d_nj.p);
            return;
        }
        for (i64 t = 0; t < smp.n_sets; t++)
            k_draw_samples_dev(c, base, t, S, directed, smp.d_pos.p + t * S, smp.d_ni.p + t * S, smp.d_nj.p + t * S);
        if (exact_directed) { // the un-reseeded second positive draw of :510 (its non-edges are not used)
            smp.d_pos2.ensure(smp.n_sets * S);
            DevBuf<i32> di, dj;
            di.ensure(S); dj.ensure(S);
            for (i64 t = 0; t < smp.n_sets; t++)
                k_draw_samples_dev(c, base + 0x7777, 1000 + t, S, directed, smp.d_pos2.p + t * S, di.p, dj.p);
        }
        return;
    }
    smp.pos_idx.resize(smp.n_sets * S);
    smp.neg_i.resize(smp.n_sets * S);
    smp.neg_j.resize(smp.n_sets * S);
    for (i64 t = 0; t < smp.n_sets; t++)
        host_draw_samples(c, base, t, S, directed, &smp.pos_idx[t * S], &smp.neg_i[t * S], &smp.neg_j[t * S]);
    if (exact_directed) { // the un-reseeded second positive draw of :510
        smp.pos_idx2.resize(smp.n_sets * S);
        std::vector<i64> di(S), dj(S);
        for (i64 t = 0; t < smp.n_sets; t++)
            host_draw_samples(c, base + 0x7777, 1000 + t, S, directed, &smp.pos_idx2[t * S], di.data(), dj.data());
    }
}

static bool samples_can_start_early(cge_ctx *c, i64 seed, bool exact_directed) {
    return seed != -1 && !exact_directed && sampler_uses_device(c) && !c->edges_sharded;
}

// exact distance of one vertex pair with dist()'s own arithmetic (src/auxilary.jl:14-20)
static double exact_pair_distance(cge_ctx *c, i64 bi, i64 bj) {
    DevBuf<i32> &pij = c->epd_i;
    DevBuf<double> &dd = c->epd_d;
    pij.ensure(2);
    dd.ensure(1);
    if (c->rows_sharded) { // the two rows come from their owners (zero-filled gather, exact), then the same kernel on the pair
        const i64 d = c->d;
        const i32 l[2] = {c->h_glob2loc[bi], c->h_glob2loc[bj]}, two[2] = {0, bi == bj ? 0 : 1};
        DevBuf<double> &rows = c->dm_seed;
        rows.ensure(2 * d + 2);
        double hi = 0.0;
        HIP_CHECK(hipMemcpyAsync(pij.p, l, sizeof(l), hipMemcpyHostToDevice, c->stream));
        k_gather_rows_f64(c, c->Xr.p, c->n_loc, d, 1, pij.p, 2, rows.p);
        allreduce(c, rows.p, 2 * d, 2);
        HIP_CHECK(hipMemcpyAsync(pij.p, two, sizeof(two), hipMemcpyHostToDevice, c->stream));
        k_pair_dist(c, rows.p, d, pij.p, pij.p + 1, 1, 1.0, dd.p);
        HIP_CHECK(hipMemcpyAsync(&hi, dd.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return hi;
    }
    const i32 h[2] = {(i32)bi, (i32)bj};
    double hi = 0.0;
    HIP_CHECK(hipMemcpyAsync(pij.p, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    k_pair_dist(c, c->Xr.p, c->d, pij.p, pij.p + 1, 1, 1.0, dd.p);
    HIP_CHECK(hipMemcpyAsync(&hi, dd.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return hi;
}

// diameter of the resident embedding (this rank's share), brute force over all pair tiles
static double resident_diameter(cge_ctx *c, int part, int nparts, i64 *ai, i64 *aj) {
    ensure_centred(c);
    double bv;
    i64 bi, bj;
    k_max_pair(c, c->Xc.p, c->rnorm.p, c->n, c->ldn, c->dpad, part, nparts, &bv, &bi, &bj);
    c->stat_diameter_path = 1;
    const double hi = bv >= 0.0 ? exact_pair_distance(c, bi, bj) : 0.0;
    c->stat_hi_i = bi; c->stat_hi_j = bj;
    if (ai) *ai = bi + 1;
    if (aj) *aj = bj + 1;
    return hi;
}

// the same with landmark-pair pruning in front (diameter_host.cpp); `mu` = N reference points (device, row-major)
static double resident_diameter_lm(cge_ctx *c, const double *mu, const double *lw, const std::vector<i32> &lcomm, i64 C,
                                   i64 N, int part, int nparts) {
    if ((c->opt_diameter != 1 || c->rows_sharded) && (i64)c->h_mem_off.size() == N + 1) { // (sharded rows: the pruned search only)
        double d2;
        i64 bi, bj;
        if (host_diameter_pruned(c, mu, lw, lcomm, C, N, c->h_mem_off, c->h_mem, part, nparts, &d2, &bi, &bj)) {
            c->stat_diameter_path = 2;
            c->stat_hi_i = bi; c->stat_hi_j = bj;
            return exact_pair_distance(c, bi, bj);
        }
    }
    return resident_diameter(c, part, nparts, nullptr, nullptr);
}

// landmark -> members CSR (ascending vertex id) from a 0-based assignment
static void build_landmark_index(cge_ctx *c, const std::vector<i32> &v2l0, i64 N) {
    const i64 n = (i64)v2l0.size();
    if (c->rows_sharded) { // this rank's members (local row ids, ascending) + the global sizes; the index goes to the device
        const i64 nl = c->n_loc;
        c->h_gl_off.assign(N + 1, 0);
        c->h_mem_off.assign(N + 1, 0);
        c->h_mem.resize(nl);
        for (i64 i = 0; i < n; i++) c->h_gl_off[v2l0[i] + 1]++;
        for (i64 i = 0; i < nl; i++) c->h_mem_off[v2l0[c->h_loc2glob[i]] + 1]++;
        for (i64 l = 0; l < N; l++) { c->h_gl_off[l + 1] += c->h_gl_off[l]; c->h_mem_off[l + 1] += c->h_mem_off[l]; }
        std::vector<i32> cur(c->h_mem_off.begin(), c->h_mem_off.end() - 1);
        for (i64 i = 0; i < nl; i++) c->h_mem[cur[v2l0[c->h_loc2glob[i]]]++] = (i32)i;
        c->lm_memoff.ensure(N + 1);
        c->lm_mem.ensure(nl);
        HIP_CHECK(hipMemcpyAsync(c->lm_memoff.p, c->h_mem_off.data(), sizeof(i32) * (N + 1), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->lm_mem.p, c->h_mem.data(), sizeof(i32) * nl, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        c->lm_index_on_device = true;
        return;
    }
    c->lm_index_on_device = false;
    c->h_mem_off.assign(N + 1, 0);
    c->h_mem.resize(n);
    for (i64 i = 0; i < n; i++) c->h_mem_off[v2l0[i] + 1]++;
    for (i64 l = 0; l < N; l++) c->h_mem_off[l + 1] += c->h_mem_off[l];
    std::vector<i32> cur(c->h_mem_off.begin(), c->h_mem_off.end() - 1);
    for (i64 i = 0; i < n; i++) c->h_mem[cur[v2l0[i]]++] = (i32)i;
}

int cge_max_pair_dist(cge_ctx *c, int part, int nparts, double *hi, int64_t *arg_i, int64_t *arg_j) {
    if (!c || !hi || nparts < 1 || part < 0 || part >= nparts) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    *hi = resident_diameter(c, part, nparts, arg_i, arg_j);
    CGE_CATCH(c)
}

// ---- wGCL -------------------------------------------------------------------------------------------
static void upload_i64_as_i32(cge_ctx *c, const i64 *h, i64 cnt, i64 lo, i64 hi, DevBuf<i32> &out, const char *what) {
    std::vector<i32> t(cnt);
    for (i64 i = 0; i < cnt; i++) {
        if (h[i] < lo || h[i] > hi) CGE_THROW(CGE_E_ARG, "%s: id %lld outside %lld..%lld", what, (long long)h[i], (long long)lo, (long long)hi);
        t[i] = (i32)(h[i] - 1);
    }
    out.ensure(cnt);
    HIP_CHECK(hipMemcpyAsync(out.p, t.data(), sizeof(i32) * cnt, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

// star-graph guard of wGCL_directed (src/divergence.jl:321-334)
static bool is_star(const std::vector<i32> &star, i64 N) {
    bool has_nm1 = false, has_2nm1 = false;
    i64 sum = 0, cnt2 = 0;
    for (i64 i = 0; i < N; i++) {
        if (star[i] == N - 1) has_nm1 = true;
        if (star[i] == 2 * (N - 1)) has_2nm1 = true;
        sum += star[i];
        if (star[i] == 2) cnt2++;
    }
    return (has_nm1 && sum == 2 * (N - 1)) || (has_2nm1 && cnt2 == N - 1);
}

int cge_wgcl(cge_ctx *c, const cge_wgcl_args *a, double out[7], int *out_len, cge_trace *trace) {
    if (!c || !a || !out || !out_len) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int directed = a->directed;
    // the score graph of this entry point (and the init_* graph it may upload) is held whole on every rank
    struct KeepOption { int &ref; int val; ~KeepOption() { ref = val; } } keep_ingest{c->opt_shard_ingest, c->opt_shard_ingest};
    c->opt_shard_ingest = 0;
    if (!a->edges_src || !a->edges_dst || a->m <= 0) CGE_THROW(CGE_E_ARG, "wGCL: empty edge list");
    i64 N = 0;
    for (i64 e = 0; e < a->m; e++) N = std::max(N, std::max(a->edges_src[e], a->edges_dst[e])); // maximum(edges) :41
    const bool landmarks = a->n_v_to_l > 0;                                                      // :44
    if (a->n_comm != N) CGE_THROW(CGE_E_ASSERT, "AssertionError: No. communities not matching no. vertices"); // :50
    if (a->n_distances != N) CGE_THROW(CGE_E_ASSERT, "AssertionError: Distances vector length is not equal to no. vertices"); // :81
    if (a->embed_rows < N) CGE_THROW(CGE_E_ARG, "wGCL: embedding has fewer rows than vertices");
    i64 C = 0;
    for (i64 i = 0; i < N; i++) C = std::max(C, a->comm[i]);
    const i64 d = a->d;

    // score graph -> device scratch
    DevBuf<i32> g_src, g_dst;
    DevBuf<double> g_w, colbuf;
    upload_i64_as_i32(c, a->edges_src, a->m, 1, N, g_src, "edges");
    upload_i64_as_i32(c, a->edges_dst, a->m, 1, N, g_dst, "edges");
    g_w.ensure(a->m);
    HIP_CHECK(hipMemcpyAsync(g_w.p, a->eweights, sizeof(double) * a->m, hipMemcpyHostToDevice, st));
    upload_i64_as_i32(c, a->comm, N, 1, C, c->s_comm, "comm");
    colbuf.ensure((size_t)a->embed_rows * d);
    HIP_CHECK(hipMemcpyAsync(colbuf.p, a->embed, sizeof(double) * a->embed_rows * d, hipMemcpyHostToDevice, st));
    c->s_emb.ensure((size_t)a->embed_rows * d);
    k_transpose_to_rowmajor(c, colbuf.p, c->s_emb.p, a->embed_rows, d);
    c->s_dist.ensure(N);
    c->s_vw.ensure(N);
    HIP_CHECK(hipMemcpyAsync(c->s_dist.p, a->distances, sizeof(double) * N, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(c->s_vw.p, a->vweights, sizeof(double) * N, hipMemcpyHostToDevice, st));
    const i64 vlen = directed ? C * C : packed_len(C);
    c->s_vectC.ensure(vlen);
    HIP_CHECK(hipMemsetAsync(c->s_vectC.p, 0, sizeof(double) * vlen, st));
    k_edge_scatter(c, g_src.p, g_dst.p, g_w.p, 0, a->m, nullptr, c->s_comm.p, N, C, directed, nullptr, c->s_vectC.p);
    ScoreGraph G;
    G.N = N; G.d = d; G.C = C;
    G.emb = c->s_emb.p; G.dist = c->s_dist.p; G.vw = c->s_vw.p; G.comm = c->s_comm.p; G.vectC = c->s_vectC.p;
    if (directed) {
        c->s_degin.ensure(N);
        c->s_degout.ensure(N);
        DevBuf<i32> star;
        star.ensure(N);
        HIP_CHECK(hipMemsetAsync(c->s_degin.p, 0, sizeof(double) * N, st));
        HIP_CHECK(hipMemsetAsync(c->s_degout.p, 0, sizeof(double) * N, st));
        HIP_CHECK(hipMemsetAsync(star.p, 0, sizeof(i32) * N, st));
        k_edge_degrees(c, g_src.p, g_dst.p, g_w.p, a->m, c->s_degout.p, c->s_degin.p, star.p);
        std::vector<i32> hstar(N);
        HIP_CHECK(hipMemcpyAsync(hstar.data(), star.p, sizeof(i32) * N, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (is_star(hstar, N)) {
            out[0] = -1.0;
            for (int k = 1; k < 6; k++) out[k] = 0.0;
            *out_len = 6;
            return CGE_OK;
        }
        G.deg_in = c->s_degin.p;
        G.deg_out = c->s_degout.p;
    }
    HIP_CHECK(hipStreamSynchronize(st));

    // the graph the local score samples from: the original graph in landmark mode, else the score graph
    OrigView ov;
    if (landmarks) {
        const bool have_init = a->init_embed && a->init_edges_src && a->init_edges_dst && a->init_vweights;
        if (have_init) { // (re)load the original graph as the resident one
            const i64 n0 = a->n_init;
            if (a->n_v_to_l != n0) CGE_THROW(CGE_E_ARG, "wGCL: v_to_l and init_vweights differ in length");
            int rc = cge_set_graph(c, a->init_edges_src, a->init_edges_dst, a->init_eweights, a->m_init, n0);
            if (rc) throw CgeError{rc, c->err};
            rc = cge_set_embedding(c, a->init_embed, n0, d);
            if (rc) throw CgeError{rc, c->err};
            rc = cge_set_vertex_data(c, nullptr, a->init_vweights, n0);
            if (rc) throw CgeError{rc, c->err};
        } else if (!c->Xr.p || !c->src.p || !c->vw.p || c->n != a->n_v_to_l)
            CGE_THROW(CGE_E_ARG, "wGCL: landmark mode needs init_* arrays or matching resident inputs");
        upload_i64_as_i32(c, a->v_to_l, a->n_v_to_l, 1, N, c->v2l, "v_to_l");
        {
            std::vector<i32> v2l0(a->n_v_to_l);
            for (i64 i = 0; i < a->n_v_to_l; i++) v2l0[i] = (i32)(a->v_to_l[i] - 1);
            build_landmark_index(c, v2l0, N);
        }
        ov.n = c->n; ov.m = c->m; ov.Xr = c->Xr.p; ov.vw = c->vw.p; ov.v2l = c->v2l.p;
        ov.lweight = c->s_vw.p; ov.src = c->src.p; ov.dst = c->dst.p; ov.h_w = c->h_w.empty() ? nullptr : c->h_w.data();
        std::vector<i32> lcomm0(N);
        for (i64 i = 0; i < N; i++) lcomm0[i] = (i32)(a->comm[i] - 1);
        double hi = resident_diameter_lm(c, c->s_emb.p, c->s_vw.p, lcomm0, C, N, c->has_coll ? c->coll.rank : 0,
                                         c->has_coll ? c->coll.world : 1);
        hi = allreduce_scalar_max(c, hi);
        ov.hi = hi;
        c->stat_last_hi = hi;
    } else {
        // exact mode: make the score graph the resident graph so the sampler can reject its edges
        int rc = cge_set_graph(c, a->edges_src, a->edges_dst, a->eweights, a->m, N);
        if (rc) throw CgeError{rc, c->err};
    }
    SampleSet smp;
    if (a->pos_idx && a->neg_i && a->neg_j && a->n_sample_sets > 0) {
        if (c->edges_sharded) CGE_THROW(CGE_E_ARG, "wGCL: caller-drawn samples index the whole edge list, the resident one is sharded (option shard_ingest)");
        smp.S = a->auc_samples;
        smp.n_sets = a->n_sample_sets;
        const i64 tot = smp.S * smp.n_sets;
        smp.pos_idx.assign(a->pos_idx, a->pos_idx + tot);
        smp.neg_i.assign(a->neg_i, a->neg_i + tot);
        smp.neg_j.assign(a->neg_j, a->neg_j + tot);
        if (a->pos_idx2) smp.pos_idx2.assign(a->pos_idx2, a->pos_idx2 + tot);
    } else
        make_samples(c, a->seed, a->auc_samples, directed, directed && !landmarks, smp);
    host_wgcl_sweep(c, G, landmarks ? &ov : nullptr, c->src.p, c->dst.p, c->h_w.empty() ? nullptr : c->h_w.data(), c->m, directed, a->split, smp,
                    out, out_len, trace);
    flush_timers(c);
    CGE_CATCH(c)
}


int cge_score(cge_ctx *c, const cge_score_args *a, double out[7], int *out_len, cge_trace *trace) {
    if (!c || !a || !out || !out_len) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    check_resident(c, "score");
    c->phases.ms.clear();
    const int directed = a->directed;
    const i64 d = c->d;
    ScoreGraph G;
    OrigView ov;
    std::vector<i32> lcomm_host;
    const bool landmarks = a->land != -1;
    if (c->samp_pending.on) { // (a score that failed between the two halves of its draw)
        HIP_CHECK(hipStreamSynchronize(st));
        c->samp_pending.on = false;
    }
    // The local score's samples depend on the resident graph and the seed only: their draw and the first round of the rejection are
    // enqueued early -- behind the first synchronisation of the landmark phase, whose host-side set-up then leaves the device idle
    // for a few hundred microseconds; the verdict is looked at where the samples used to be drawn (a star graph's early return or
    // an error in between leaves the draw pending: see above)
    const bool samples_early = landmarks && samples_can_start_early(c, a->seed, false);
    // (the hook captures this call's arguments: whatever happens, it does not outlive the call)
    struct HookGuard {
        cge_ctx *c;
        ~HookGuard() { c->after_unique = nullptr; }
    } hook_guard{c};
    c->after_unique = nullptr;
    if (samples_early)
        c->after_unique = [c, a, directed]() {
            c->smp.reset();
            make_samples(c, a->seed, a->auc_samples, directed, false, c->smp, 1);
        };
    DevBuf<double> &zeros = c->sw_zeros;
    double t0;
    DevBuf<i32> &star = c->s_star;
    auto star_exit = [&](i64 Nv) -> bool { // star-graph guard of wGCL_directed (src/divergence.jl:321-334)
        std::vector<i32> hstar(Nv);
        HIP_CHECK(hipMemcpyAsync(hstar.data(), star.p, sizeof(i32) * Nv, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (!is_star(hstar, Nv)) return false;
        out[0] = -1.0;
        for (int k = 1; k < 6; k++) out[k] = 0.0;
        *out_len = 6;
        return true;
    };
    if (landmarks) {
        landmarks_run_impl(c, a->clusters_flat, a->clusters_off, a->n_clusters, a->land, a->forced, a->method, directed,
                           directed != 0 || c->opt_landmark_edges != 0);
        c->after_unique = nullptr;
        const i64 N = c->N, C = c->n_comm_max;
        // wGCL's own `maximum(edges)` / size asserts (src/divergence.jl:41,50): the highest-numbered
        // landmark must carry an edge -- always true when every vertex has positive weight
        G.N = N; G.d = d; G.C = C;
        G.emb = c->lemb.p; G.dist = c->dii.p; G.vw = c->lweight.p; G.comm = c->lcomm.p; G.vectC = c->vectC.p;
        if (directed) { // degrees / star counts of the landmark graph from the landmark-pair matrix
            c->s_degin.ensure(N);
            c->s_degout.ensure(N);
            star.ensure(N);
            if (c->wedges_block_only) { // a reduce-scattered matrix: this rank's row block, then the ranks add the three vectors
                const i64 per = wedge_rows_per_rank(c, N), r0 = std::min<i64>(N, per * c->coll.rank), r1 = std::min<i64>(N, r0 + per);
                DevBuf<double> &X = c->samp_xchg;
                X.ensure(3 * N);
                k_wedge_degrees_block(c, c->wedges.p, N, r0, r1, X.p);
                allreduce(c, X.p, 3 * N, 0);
                k_degrees_unpack(c, X.p, N, c->s_degout.p, c->s_degin.p, star.p);
            } else
                k_wedge_degrees(c, c->wedges.p, N, c->s_degout.p, c->s_degin.p, star.p);
            if (star_exit(N)) return CGE_OK;
            G.deg_in = c->s_degin.p;
            G.deg_out = c->s_degout.p;
        }
        t0 = now_ms();
        ov.n = c->n; ov.m = c->m; ov.Xr = c->Xr.p; ov.vw = c->vw.p; ov.v2l = c->v2l.p; ov.lweight = c->lweight.p;
        ov.src = c->src.p; ov.dst = c->dst.p; ov.h_w = c->h_w.empty() ? nullptr : c->h_w.data();
        double hi = 0.0;
        lcomm_host.resize(N); // community of a landmark = community of any member (landmarks never span two): :427
        {
            std::vector<i32> &lcomm0 = lcomm_host;
            HIP_CHECK(hipMemcpyAsync(lcomm0.data(), c->lcomm.p, sizeof(i32) * N, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            hi = resident_diameter_lm(c, c->lemb.p, c->lweight.p, lcomm0, C, N, c->has_coll ? c->coll.rank : 0,
                                      c->has_coll ? c->coll.world : 1);
            hi = allreduce_scalar_max(c, hi);
        }
        ov.h_lcomm = lcomm_host.data(); // (the sweep groups the landmarks by community: no second read-back)
        ov.hi = hi;
        c->stat_last_hi = hi;
        c->phases.ms["diameter"] = now_ms() - t0; // what the main thread still waited for
    } else {
        if (c->edges_sharded)
            CGE_THROW(CGE_E_ARG, "score: exact mode reads the whole edge list on every rank; the resident one is sharded (option shard_ingest)");
        if (c->rows_sharded)
            CGE_THROW(CGE_E_ARG, "score: exact mode reads every embedding row on every rank; the resident rows are sharded (option shard_rows)");
        const i64 N = c->n, C = c->n_comm_max;
        zeros.ensure(N);
        HIP_CHECK(hipMemsetAsync(zeros.p, 0, sizeof(double) * N, st)); // distances = zeros (CGE_CLI.jl:4)
        const i64 vlen = directed ? C * C : packed_len(C);
        c->vectC.ensure(vlen);
        scatter_vectC_resident(c, C, directed, c->vectC.p);
        G.N = N; G.d = d; G.C = C;
        G.emb = c->Xr.p; G.dist = zeros.p; G.vw = c->vw.p; G.comm = c->comm.p; G.vectC = c->vectC.p;
        if (directed) {
            c->s_degin.ensure(N);
            c->s_degout.ensure(N);
            star.ensure(N);
            HIP_CHECK(hipMemsetAsync(c->s_degin.p, 0, sizeof(double) * N, st));
            HIP_CHECK(hipMemsetAsync(c->s_degout.p, 0, sizeof(double) * N, st));
            HIP_CHECK(hipMemsetAsync(star.p, 0, sizeof(i32) * N, st));
            k_edge_degrees(c, c->src.p, c->dst.p, c->unit_weights ? nullptr : c->w.p, c->m, c->s_degout.p, c->s_degin.p,
                           star.p);
            if (star_exit(N)) return CGE_OK;
            G.deg_in = c->s_degin.p;
            G.deg_out = c->s_degout.p;
        }
    }
    t0 = now_ms();
    SampleSet &smp = c->smp;
    if (samples_early && c->samp_pending.on) make_samples(c, a->seed, a->auc_samples, directed, false, smp, 2);
    else {
        smp.reset();
        make_samples(c, a->seed, a->auc_samples, directed, directed && !landmarks, smp);
    }
    c->phases.ms["samples"] = now_ms() - t0;
    t0 = now_ms();
    host_wgcl_sweep(c, G, landmarks ? &ov : nullptr, c->src.p, c->dst.p, c->h_w.empty() ? nullptr : c->h_w.data(), c->m, directed, a->split, smp,
                    out, out_len, trace);
    HIP_CHECK(hipStreamSynchronize(st));
    c->phases.ms["sweep"] = now_ms() - t0;
    flush_timers(c);
    CGE_CATCH(c)
}

// ---- helpers ----------------------------------------------------------------------------------------
int64_t cge_idx(int64_t n, int64_t i, int64_t j) { return n * (i - 1) - (i - 1) * (i - 2) / 2 + j - i + 1; }

int cge_js(cge_ctx *c, const double *vC, const double *vB, int64_t len, const uint8_t *vI, int internal, double *out) {
    if (!c || !vC || !vB || !out || len <= 0) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    // vI (when given) selects bins; the device kernel derives the diagonal mask from the packed/square
    // layout, so here the selected bins are compacted on the host first and scored with mode 0.
    std::vector<double> p, q;
    for (i64 k = 0; k < len; k++)
        if (!vI || ((vI[k] != 0) == (internal != 0))) { p.push_back(vC[k]); q.push_back(vB[k]); }
    const i64 L = (i64)p.size();
    DevBuf<double> dp, dq, r;
    dp.ensure(L); dq.ensure(L); r.ensure(1);
    HIP_CHECK(hipMemcpyAsync(dp.p, p.data(), sizeof(double) * L, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(dq.p, q.data(), sizeof(double) * L, hipMemcpyHostToDevice, c->stream));
    k_js(c, dp.p, dq.p, L, 1, 0, 0, r.p);
    HIP_CHECK(hipMemcpyAsync(out, r.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    CGE_CATCH(c)
}

int cge_edge_scatter(cge_ctx *c, const int64_t *v_to_l, int64_t N, int64_t C, int directed, int64_t e0, int64_t e1,
                     double *wedges_out, double *vect_C_out) {
    if (!c || N <= 0 || C <= 0 || e0 < 0 || e1 < e0) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->src.p || !c->comm.p) CGE_THROW(CGE_E_ARG, "edge_scatter: graph and vertex data must be resident");
    if (c->edges_sharded) CGE_THROW(CGE_E_ARG, "edge_scatter: the resident edge list is sharded over the ranks (option shard_ingest)");
    if (e1 > c->m) CGE_THROW(CGE_E_ARG, "edge_scatter: edge range beyond m");
    hipStream_t st = c->stream;
    DevBuf<i32> dv;
    if (v_to_l) upload_i64_as_i32(c, v_to_l, c->n, 1, N, dv, "v_to_l");
    const i64 vlen = directed ? C * C : packed_len(C);
    DevBuf<double> dw, dc;
    if (wedges_out) {
        if (!v_to_l) CGE_THROW(CGE_E_ARG, "edge_scatter: wedges need v_to_l");
        dw.ensure((size_t)N * N);
        HIP_CHECK(hipMemsetAsync(dw.p, 0, sizeof(double) * N * N, st));
    }
    if (vect_C_out) {
        dc.ensure(vlen);
        HIP_CHECK(hipMemsetAsync(dc.p, 0, sizeof(double) * vlen, st));
    }
    if (!wedges_out && vect_C_out && e0 == 0 && e1 == c->m && C == c->n_comm_max && !c->has_coll)
        scatter_vectC_resident(c, C, directed, dc.p); // the score path's forms of the whole-list pass
    else
        k_edge_scatter(c, c->src.p, c->dst.p, c->unit_weights ? nullptr : c->w.p, e0, e1, v_to_l ? dv.p : nullptr,
                       c->comm.p, N, C, directed, wedges_out ? dw.p : nullptr, vect_C_out ? dc.p : nullptr);
    if (wedges_out) HIP_CHECK(hipMemcpyAsync(wedges_out, dw.p, sizeof(double) * N * N, hipMemcpyDeviceToHost, st));
    if (vect_C_out) HIP_CHECK(hipMemcpyAsync(vect_C_out, dc.p, sizeof(double) * vlen, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    flush_timers(c);
    CGE_CATCH(c)
}

// ---- louvain_clust (src/clustering.jl:14-68): level-1 communities of the resident graph -------------------------------
int cge_louvain(cge_ctx *c, int64_t *comm_out, int64_t *n_comm, double *modularity, int64_t *rounds) {
    if (!c || !comm_out) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->src.p || c->m <= 0 || c->n <= 0) CGE_THROW(CGE_E_ARG, "louvain: no resident graph (cge_set_graph)");
    if (c->edges_sharded) CGE_THROW(CGE_E_ARG, "louvain: the resident edge list is sharded over the ranks (option shard_ingest)");
    k_louvain_level1(c, comm_out, n_comm, modularity, rounds);
    CGE_CATCH(c)
}

// ---- options / statistics ---------------------------------------------------------------------------
int cge_set_option(cge_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return CGE_E_ARG;
    if (!strcmp(key, "diameter")) { // 0 auto, 1 brute force, 2 pruned only
        if (value < 0 || value > 2) return CGE_E_ARG;
        c->opt_diameter = (int)value;
        return CGE_OK;
    }
    if (!strcmp(key, "diameter_f32")) { // point-to-reference maxima of the pruned diameter: 2 (default) bf16 matrix pipe on two-term operands
        if (value < 0 || value > 2) return CGE_E_ARG; // (K <= 128, else as 1), 1 fp32-input MFMA (both: rigorous upper bounds); 0: fp64 MFMA
        c->opt_diameter_f32 = (int)value;
        return CGE_OK;
    }
    if (!strcmp(key, "fit_persistent")) { // 0 auto, 1 never, 2 whenever the score graph fits the register file
        if (value < 0 || value > 2) return CGE_E_ARG;
        c->opt_fit_persistent = (int)value;
        return CGE_OK;
    }
    if (!strcmp(key, "pow_exp2")) { // 1 (default): (1 - D)^alpha from log2(1 - D) kept per score; 0: the library pow per alpha
        c->opt_pow_exp2 = value != 0;
        return CGE_OK;
    }
    if (!strcmp(key, "shard_runsplit")) { // N > 1 only: 0 = runsplit replicated, 1 (default) = forced phase and big batches of the
        if (value < 0 || value > 2) return CGE_E_ARG; // global phase split over the ranks, 2 = every batch (tests)
        c->opt_shard_forced = (int)value;
        return CGE_OK;
    }
    if (!strcmp(key, "shard_samples")) { // N > 1: 0 = tallies replicated, 1 (default) = split from 10^5 samples on with the in-library communicator, 2 = always
        if (value < 0 || value > 2) return CGE_E_ARG;
        c->opt_shard_samples = (int)value;
        return CGE_OK;
    }
    if (!strcmp(key, "exact_relabel")) { // exact mode beyond 8192 vertices: 1 (default) = score graph relabelled by community, 0 = as given (A/B, tests)
        c->opt_exact_relabel = value != 0;
        return CGE_OK;
    }
    if (!strcmp(key, "bvec_blocks")) { // 1: sweeps from 256 vertices on relabel the score graph by community and sum vect_B by tiles
        c->opt_bvec_blocks = value != 0; // (measured slower); 0 (default): row bins + row sums + fold
        return CGE_OK;
    }
    if (!strcmp(key, "wedges_reduce_scatter")) { // N > 1: 1 = the N x N landmark-pair matrix is reduce-scattered by row blocks (a fetch of
        c->opt_wedges_rs = value != 0;            // the landmark edge list is then collective); 0 (default): all-reduced, every consumer local
        return CGE_OK;
    }
    if (!strcmp(key, "fit_fused")) { // 1 (default): in landmark mode the power matrix, vect_B's tile sums and the local score's tallies ride
        c->opt_fit_fused = value != 0; // on the launch of the undirected persistent fit; 0: separate launches (A/B, cross-check)
        return CGE_OK;
    }
    if (!strcmp(key, "shard_ingest")) { // N > 1: 1 = cge_set_graph keeps this rank's slice of the edge list only and cge_set_embedding uploads a
        // slice of rows per rank and all-gathers them over xGMI (set the collectives first); 0 (default): every rank uploads and keeps everything
        c->opt_shard_ingest = value != 0;
        return CGE_OK;
    }
    if (!strcmp(key, "shard_rows")) { // N > 1: 1 = cge_set_embedding / cge_set_embedding_device keep the rows of this rank's communities only
        // (sharded BY COMMUNITY; set the collectives and upload the communities first); 0 (default): every rank holds every row
        c->opt_shard_rows = value != 0;
        return CGE_OK;
    }
    if (!strcmp(key, "landmark_edges")) { // 1: cge_score also builds the landmark-pair matrix / edge count that landmarks() returns
        c->opt_landmark_edges = value != 0; // (src/landmarks.jl:433-463; the undirected score itself does not read it); 0 (default): on first fetch
        return CGE_OK;
    }
    if (!strcmp(key, "fit_max_iterations")) { // iterations after which a Chung-Lu fit that has not converged raises CGE_E_ASSERT (default 2 000 000)
        if (value < 1) return CGE_E_ARG;
        c->opt_fit_max_iters = value;
        return CGE_OK;
    }
    return CGE_E_ARG;
}
// the testing knobs (include/cge_hip_testing.h): not part of the boundary
int cge_set_test_option(void *ctx, const char *key, int64_t value) {
    cge_ctx *c = (cge_ctx *)ctx;
    if (!c || !key) return CGE_E_ARG;
    if (!strcmp(key, "test_bvec_plain")) { // testing: 1 = vect_B by the kernels of score graphs beyond the LDS budget
        c->opt_test_bvec_plain = value != 0;
        return CGE_OK;
    }
    if (!strcmp(key, "fit_persistent_test_delay")) { // testing: start skew of the persistent fits' tile waves, in naps of ~3 us
        if (value < 0 || value > 100000) return CGE_E_ARG;
        c->opt_fit_test_delay = (int)value;
        return CGE_OK;
    }
    if (!strcmp(key, "fit_persistent_test_timeout")) { // testing: 1 = the persistent fit abandons every launch at once
        c->opt_fit_test_timeout = value != 0;
        return CGE_OK;
    }
    return CGE_E_ARG;
}
int cge_get_stat(cge_ctx *c, const char *key, int64_t *value) {
    if (!c || !key || !value) return CGE_E_ARG;
    if (!strcmp(key, "landmarks")) *value = c->lm_ready ? c->N : 0; // no side effects (cge_landmarks_info may build the N x N matrix)
    else if (!strcmp(key, "diameter_path")) *value = c->stat_diameter_path;
    else if (!strcmp(key, "diameter_candidate_pairs")) *value = c->stat_cand_pairs;
    else if (!strcmp(key, "diameter_candidate_tiles")) *value = c->stat_cand_tiles;
    else if (!strcmp(key, "diameter_refs")) *value = c->stat_nref;
    else if (!strcmp(key, "diameter_bound_pass")) *value = c->stat_bound_pass; // of the last pruned diameter: 2 bf16-split, 1 fp32 MFMA, 0 fp64 MFMA
    else if (!strcmp(key, "diameter_arg_i")) *value = c->stat_hi_i + 1; // the arg-max pair of the last diameter (1-based vertex ids)
    else if (!strcmp(key, "diameter_arg_j")) *value = c->stat_hi_j + 1;
    else if (!strcmp(key, "fit_persistent_alphas")) *value = c->stat_fit_persistent;
    else if (!strcmp(key, "fit_iterations")) *value = c->stat_fit_iters;
    else if (!strcmp(key, "fit_fused_alphas")) *value = c->stat_fit_fused; // alphas of the last sweep whose chain rode on the fit's launch
    else if (!strcmp(key, "fit_persistent_fallbacks")) *value = c->stat_fit_fallbacks;
    else if (!strcmp(key, "landmark_batches")) *value = c->stat_lm_batches;
    else if (!strcmp(key, "landmark_batch_rows")) *value = c->stat_lm_rows;
    else if (!strcmp(key, "landmark_splits")) *value = c->stat_lm_splits;
    else if (!strcmp(key, "cut_tie_tasks")) { // (read from the device on request: a synchronising copy)
        int v = 0;
        if (c->cut_ties.p && (hipStreamSynchronize(c->stream) != hipSuccess ||
                              hipMemcpy(&v, c->cut_ties.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess))
            return CGE_E_HIP;
        *value = v;
    }
    else if (!strcmp(key, "edge_layout_build_us")) *value = c->stat_layout_build_us;
    else if (!strcmp(key, "edge_chunks")) *value = c->be_nchunks;
    else if (!strcmp(key, "edges_resident")) *value = c->m;
    else if (!strcmp(key, "edges_total")) *value = c->m_total;
    else if (!strcmp(key, "embedding_words_resident")) *value = (i64)c->Xr.n;
    else if (!strcmp(key, "rows_resident")) *value = c->Xr.p ? lm_rows(c) : 0; // embedding rows held by this rank (option shard_rows: ~ n / world)
    else if (!strcmp(key, "rows_total")) *value = c->n;
    else if (!strcmp(key, "collective_calls")) *value = c->stat_coll_calls;
    else if (!strcmp(key, "collective_bytes")) *value = c->stat_coll_bytes;
    else if (!strcmp(key, "diameter_bits")) memcpy(value, &c->stat_last_hi, sizeof(double)); // bit pattern of the last `hi`
    else return CGE_E_ARG;
    return CGE_OK;
}

// ---- profiling --------------------------------------------------------------------------------------
int cge_profile_enable(cge_ctx *c, int on) {
    if (!c) return CGE_E_ARG;
    c->profiling = on != 0;
    return CGE_OK;
}
int cge_profile_select(cge_ctx *c, const char *names) {
    if (!c) return CGE_E_ARG;
    c->profile_only.clear();
    std::string cur;
    for (const char *p = names ? names : ""; ; p++) {
        if (*p == ',' || *p == 0) {
            if (!cur.empty()) c->profile_only.push_back(cur);
            cur.clear();
            if (*p == 0) break;
        } else
            cur.push_back(*p);
    }
    return CGE_OK;
}
int cge_profile_reset(cge_ctx *c) {
    if (!c) return CGE_E_ARG;
    flush_timers(c);
    c->timers.clear();
    return CGE_OK;
}
int cge_profile_get(cge_ctx *c, const char *name, int64_t *launches, double *total_ms) {
    if (!c || !name) return CGE_E_ARG;
    flush_timers(c);
    auto it = c->timers.find(name);
    if (launches) *launches = it == c->timers.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == c->timers.end() ? 0.0 : it->second.total_ms;
    return CGE_OK;
}
int cge_profile_names(cge_ctx *c, char *buf, int64_t buf_len) {
    if (!c || !buf || buf_len <= 0) return CGE_E_ARG;
    std::string s;
    for (auto &kv : c->timers) {
        if (!s.empty()) s += ",";
        s += kv.first;
    }
    snprintf(buf, (size_t)buf_len, "%s", s.c_str());
    return CGE_OK;
}
int cge_phase_names(cge_ctx *c, char *buf, int64_t buf_len) {
    if (!c || !buf || buf_len <= 0) return CGE_E_ARG;
    std::string s;
    for (auto &kv : c->phases.ms) {
        if (!s.empty()) s += ",";
        s += kv.first;
    }
    snprintf(buf, (size_t)buf_len, "%s", s.c_str());
    return CGE_OK;
}
int cge_phase_ms(cge_ctx *c, const char *phase, double *ms) {
    if (!c || !phase || !ms) return CGE_E_ARG;
    auto it = c->phases.ms.find(phase);
    *ms = it == c->phases.ms.end() ? 0.0 : it->second;
    return CGE_OK;
}

// ---- host-only test hooks (include/cge_hip_testing.h) ------------------------------------------------
int cge_host_eig_top(const double *A, int64_t d, double *v) {
    if (!A || !v || d <= 0) return CGE_E_ARG;
    host_eig_top(A, d, v);
    return CGE_OK;
}
int cge_host_pos_draw(int64_t seed, int64_t stream_id, int64_t S, int64_t m, int64_t *pos_idx) {
    if (!pos_idx || S <= 0 || m <= 0) return CGE_E_ARG;
    host_pos_draw(seed, stream_id, S, m, pos_idx);
    return CGE_OK;
}
int cge_group_eig(void *ctx, const double *A, int64_t T, int64_t d, double *v) {
    cge_ctx *c = (cge_ctx *)ctx;
    if (!c || !A || !v || T <= 0 || d <= 0 || d > 512) return CGE_E_ARG;
    CGE_TRY(c)
    DevBuf<double> dA, dv;
    dA.ensure((size_t)T * d * d);
    dv.ensure((size_t)T * d);
    HIP_CHECK(hipMemcpyAsync(dA.p, A, sizeof(double) * T * d * d, hipMemcpyHostToDevice, c->stream));
    k_group_eig(c, dA.p, T, d, dv.p);
    HIP_CHECK(hipMemcpyAsync(v, dv.p, sizeof(double) * T * d, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    CGE_CATCH(c)
}

int cge_pow_test(void *ctx, const double *x, int64_t n, double alpha, int method, double *out) {
    cge_ctx *c = (cge_ctx *)ctx;
    if (!c || !x || !out || n <= 0) return CGE_E_ARG;
    CGE_TRY(c)
    k_pow_test(c, x, n, alpha, method, out);
    CGE_CATCH(c)
}

// testing hook (include/cge_hip_testing.h): the per-group stable sort of the projections as runsplit calls it
int cge_segment_sort_test(void *ctx, const double *z, const int32_t *task_row_off, int64_t T, double *zs_out, int32_t *perm_out) {
    cge_ctx *c = (cge_ctx *)ctx;
    if (!c || !z || !task_row_off || !zs_out || !perm_out || T <= 0) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    const i64 R = task_row_off[T];
    if (R <= 0) CGE_THROW(CGE_E_ARG, "segment_sort_test: no rows");
    std::vector<i32> rt(R), rows(R);
    i64 max_len = 0;
    for (i64 t = 0; t < T; t++) {
        if (task_row_off[t + 1] <= task_row_off[t]) CGE_THROW(CGE_E_ARG, "segment_sort_test: empty group");
        max_len = std::max<i64>(max_len, task_row_off[t + 1] - task_row_off[t]);
        for (i64 j = task_row_off[t]; j < task_row_off[t + 1]; j++) { rt[j] = (i32)t; rows[j] = (i32)j; }
    }
    DevBuf<double> dz, dzs;
    DevBuf<i32> dtro, drt, drows, dperm, dsrows, dstatus;
    dz.ensure(R); dzs.ensure(R); dtro.ensure(T + 1); drt.ensure(R); drows.ensure(R); dperm.ensure(R); dsrows.ensure(R); dstatus.ensure(T);
    HIP_CHECK(hipMemcpyAsync(dz.p, z, sizeof(double) * R, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(dtro.p, task_row_off, sizeof(i32) * (T + 1), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(drt.p, rt.data(), sizeof(i32) * R, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(drows.p, rows.data(), sizeof(i32) * R, hipMemcpyHostToDevice, c->stream));
    k_segmented_sort_z(c, dz.p, drows.p, drt.p, dtro.p, R, T, dzs.p, dperm.p, dsrows.p, dstatus.p, max_len);
    HIP_CHECK(hipMemcpyAsync(zs_out, dzs.p, sizeof(double) * R, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(perm_out, dperm.p, sizeof(i32) * R, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    CGE_CATCH(c)
}

// testing hook (include/cge_hip_testing.h): lane 0's sum of 64 values per row by the shfl_down tree and by the VALU lane
// swaps that replace it in the projection kernel -- the same bits are expected
int cge_wave_tree_test(void *ctx, const double *x, int64_t n_rows, double *out_ref, double *out_new) {
    cge_ctx *c = (cge_ctx *)ctx;
    if (!c || !x || !out_ref || !out_new || n_rows <= 0) return CGE_E_ARG;
    CGE_TRY(c)
    HIP_CHECK(hipSetDevice(c->device));
    DevBuf<double> dx, da, db;
    dx.ensure((size_t)n_rows * 64); da.ensure(n_rows); db.ensure(n_rows);
    HIP_CHECK(hipMemcpyAsync(dx.p, x, sizeof(double) * n_rows * 64, hipMemcpyHostToDevice, c->stream));
    k_wave_tree_test(c, dx.p, n_rows, da.p, db.p);
    HIP_CHECK(hipMemcpyAsync(out_ref, da.p, sizeof(double) * n_rows, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(out_new, db.p, sizeof(double) * n_rows, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    CGE_CATCH(c)
}

} // extern "C"

bool cge_exchange_fits(cge_ctx *c, size_t need) {
    if (c->xptr && need <= c->xcap) return true;
    if (!c->rccl_comm || (c->xptr && c->xptr != c->xown.p)) return false;
    c->xown.alloc_exact(std::max<size_t>(need + need / 4, 1 << 20));
    c->xptr = c->xown.p;
    c->xcap = c->xown.n;
    return true;
}

// all-gather of 8-byte words in place (the sharded ingest of the embedding): ncclAllGather on the ctx stream with the
// in-library communicator; through the hook (tests), or with a librccl that lacks the symbol, a zero-filled all-reduce of
// the words as integers -- exact on the bit patterns (a sum of doubles would turn -0.0 into +0.0)
void cge_allgather_dev(cge_ctx *c, double *buf, i64 wpr) {
    if (!c->has_coll || wpr <= 0) return;
    if (c->rccl_comm && cge_rccl_allgather(c, buf, wpr)) return;
    const i64 W = c->coll.world, r = c->coll.rank, total = wpr * W;
    if (!c->rccl_comm && c->coll_ext.allgather && c->xptr && (size_t)total <= c->xcap) { // the hook's own all-gather, through the exchange buffer
        if (buf != c->xptr)
            HIP_CHECK(hipMemcpyAsync(c->xptr + wpr * r, buf + wpr * r, sizeof(double) * (size_t)wpr, hipMemcpyDeviceToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->coll_ext.allgather(c->coll.user, c->xptr, wpr) != 0) CGE_THROW(CGE_E_COLLECTIVE, "all-gather hook failed");
        c->stat_coll_calls++;
        c->stat_coll_bytes += 8 * total;
        if (buf != c->xptr) HIP_CHECK(hipMemcpyAsync(buf, c->xptr, sizeof(double) * (size_t)total, hipMemcpyDeviceToDevice, c->stream));
        return;
    }
    if (r > 0) HIP_CHECK(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)(wpr * r), c->stream));
    if (r + 1 < W) HIP_CHECK(hipMemsetAsync(buf + wpr * (r + 1), 0, sizeof(double) * (size_t)(wpr * (W - 1 - r)), c->stream));
    const i64 piece = c->rccl_comm ? total : (i64)c->xcap;
    if (piece <= 0) CGE_THROW(CGE_E_COLLECTIVE, "all-gather: no exchange buffer set");
    for (i64 off = 0; off < total; off += piece) allreduce(c, buf + off, std::min(piece, total - off), 2);
}

// for the other translation units (diameter_host.cpp)
void cge_allreduce_dev(cge_ctx *c, double *dev, i64 count, int op) { allreduce(c, dev, count, op); }
double cge_allreduce_scalar_max(cge_ctx *c, double v) { return allreduce_scalar_max(c, v); }
