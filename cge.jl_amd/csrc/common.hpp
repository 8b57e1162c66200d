// common.hpp -- context, device buffers, error handling and launch helpers of libcge_hip.so.
// gfx950 (MI355X) only; there is no CPU fallback anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/cge_hip.h"

typedef int64_t i64;
typedef int32_t i32;

struct CgeError {
    int code;
    std::string msg;
};

#define CGE_THROW(code, ...)                                  \
    do {                                                      \
        char _b[512];                                         \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                \
        throw CgeError{(code), std::string(_b)};              \
    } while (0)

#define HIP_CHECK(expr)                                                                                 \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            CGE_THROW(_e == hipErrorOutOfMemory ? CGE_E_OOM : CGE_E_HIP, "%s failed: %s (%s:%d)", #expr, \
                      hipGetErrorString(_e), __FILE__, __LINE__);                                       \
    } while (0)

// Every kernel launch of the library is followed by hipGetLastError: a refused launch (dynamic LDS beyond what the device
// grants, a bad grid) raises CGE_E_HIP instead of leaving the output at its memset zeros.
inline void cge_launch_check(const char *what, const char *file, int line) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) CGE_THROW(CGE_E_HIP, "launch of %s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                         \
    do {                                                            \
        hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);      \
        cge_launch_check(#kernelName, __FILE__, __LINE__);          \
    } while (0)

// Dynamic LDS beyond 64 KB has to be granted per kernel AND per device (the attribute belongs to the function's code object on
// the current device); the grant is checked.
inline void cge_allow_lds(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> done;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    auto it = done.find({fn, dev});
    if (it != done.end() && it->second >= bytes) return;
    HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[{fn, dev}] = bytes;
}

// RAII device buffer
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool borrowed = false; // a view of another context's buffer (the side context of cge_score): never freed here
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        n = 0;
        borrowed = false;
    }
    void borrow(const DevBuf &o) {
        if (p == o.p && n == o.n && borrowed) return;
        release();
        p = o.p;
        n = o.n;
        borrowed = o.p != nullptr;
    }
    // grow-only allocation (contents are NOT preserved)
    void ensure(size_t count) {
        if (count <= n && p) return;
        release();
        if (count == 0) count = 1;
        HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
    }
    void alloc_exact(size_t count) {
        release();
        HIP_CHECK(hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)));
        n = count ? count : 1;
    }
};

// pinned (page-locked) host buffer, grow-only
template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    PinBuf() {}
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    void ensure(size_t count) {
        if (count <= n && p) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        if (count == 0) count = 1;
        HIP_CHECK(hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault));
        n = count;
    }
};

// Persistent worker pool: run(n, fn) executes fn(0..n-1) on the workers + the calling thread.
// Completion is counted in ITEMS, not in workers: the caller takes items itself and returns as soon as the last one is
// done, so a worker that is slow to wake up (a sleeping core: milliseconds, seen as sporadic 2-6 ms stalls of a 0.1 ms
// job) delays nobody -- it finds the job drained and goes back to sleep.
class ThreadPool {
  public:
    explicit ThreadPool(int n_workers) {
        for (int t = 0; t < n_workers; t++) workers_.emplace_back([this]() { loop(); });
    }
    ~ThreadPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    void run(i64 n, const std::function<void(i64)> &fn) {
        if (n <= 0) return;
        if (n == 1 || workers_.empty()) {
            for (i64 i = 0; i < n; i++) fn(i);
            return;
        }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->n = n;
        {
            std::lock_guard<std::mutex> lk(m_);
            cur_ = job;
            gen_++;
        }
        cv_.notify_all();
        work(*job);
        while (job->done.load(std::memory_order_acquire) < n) std::this_thread::yield(); // items other threads still hold
        // (a late worker may still look at `job` through its own reference: it finds next >= n and never touches fn)
    }

  private:
    struct Job {
        const std::function<void(i64)> *fn = nullptr;
        i64 n = 0;
        std::atomic<i64> next{0}, done{0};
    };
    static void work(Job &j) {
        for (;;) {
            const i64 i = j.next.fetch_add(1);
            if (i >= j.n) break;
            (*j.fn)(i);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = cur_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::shared_ptr<Job> cur_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    i64 launches = 0;
    double total_ms = 0.0;
};

struct Phase {
    std::map<std::string, double> ms;
};

// Device view of the graph that wGCL scores ("score graph": the landmark graph in landmark mode,
// the original graph in exact mode).  All ids 0-based on the device.
struct ScoreGraph {
    i64 N = 0, d = 0, C = 0;
    const double *emb = nullptr;   // N x d row-major
    const double *dist = nullptr;  // N   (diagonal of D: dii or zeros)
    const double *vw = nullptr;    // N   (Chung-Lu target weights; undirected)
    const i32 *comm = nullptr;     // N   0-based community
    const double *vectC = nullptr; // packed p(C) (undirected) or C*C (directed)
    const double *deg_in = nullptr, *deg_out = nullptr; // directed only
};

struct SampleSet {
    i64 S = 0, n_sets = 0;
    std::vector<i64> pos_idx, neg_i, neg_j, pos_idx2; // 1-based (caller-provided draws, small graphs)
    // library-drawn samples of a large resident graph stay on the device (0-based, n_sets * S each)
    bool on_device = false;
    DevBuf<i32> d_pos, d_ni, d_nj, d_pos2;
    void reset() { // the device buffers are grow-only scratch: they stay
        S = n_sets = 0;
        on_device = false;
        pos_idx.clear(); neg_i.clear(); neg_j.clear(); pos_idx2.clear();
    }
};
struct DevSamples { // one sample set as the AUC kernels read it (device, 0-based)
    DevBuf<i32> pi, pj, ni, nj;
    DevBuf<double> wts, dpos, dneg;
    DevBuf<i32> aidx;     // the fused fit's prepared tally operands (k_auc_prepare)
    DevBuf<double> afac;
};

struct cge_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<hipEvent_t> event_pool; // recycled kernel-timer events
    PinBuf<unsigned char> stage[2];    // pinned staging of the uploads (cge_set_graph / cge_set_embedding)
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr; // device->host result copies that overlap the kernels queued behind them
    hipEvent_t copy_ev = nullptr, copy_done = nullptr;
    std::string err;
    int n_threads = 8;
    ThreadPool *pool = nullptr; // persistent host workers (n_threads - 1 + caller)
    cge_collectives coll{};
    cge_collectives_ext coll_ext{}; // optional further ops of the hook (all-gather, reduce-scatter)
    bool has_coll = false;
    int opt_wedges_rs = 0; // 1: the N x N landmark-pair matrix goes out by row blocks (reduce-scatter); 0 (default): all-reduce
    void *rccl_comm = nullptr;           // in-library communicator (collectives.cpp); takes precedence over the hook
    i64 stat_coll_calls = 0, stat_coll_bytes = 0; // all-reduces issued since the context was created
    DevBuf<double> xown;  // library-owned exchange buffer (cge_exchange_buffer)
    double *xptr = nullptr; // exchange buffer in use (library- or caller-owned)
    size_t xcap = 0;

    // ---- resident original graph --------------------------------------------------------
    i64 n = 0, m = 0, d = 0;   // m = edges RESIDENT on this rank (all of them unless the list is sharded, below)
    // N > 1 with option "shard_ingest": a rank uploads and keeps rows [e_first, e_first + m) of the caller's edge list only
    // (m_total = the caller's count); the embedding is uploaded as a slice of rows per rank and all-gathered over xGMI
    i64 m_total = 0, e_first = 0;
    bool edges_sharded = false;
    int opt_shard_ingest = 0;
    DevBuf<double> samp_xchg;  // sampled edges of a sharded list on their way through the all-reduce
    // N > 1 with option "shard_rows": the EMBEDDING ROWS are sharded BY COMMUNITY (north star: "edge list and embedding rows
    // shard across the GPUs").  A community lives on one rank (largest first, each to the least loaded rank); rank r keeps
    // the rows of its communities only: Xr is n_loc x d, local row i = the i-th owned vertex in ascending vertex id
    // (loc2glob / glob2loc, -1 = another rank's).  Everything that walks rows (runsplit, aggregation, the bound pass and the
    // sweeps of the diameter, row hashes) runs over LOCAL ids against Xr / vw_loc / comm_loc; the small per-vertex tables
    // (vw, comm, v2l: 4-8 bytes per vertex) stay replicated, global ids.  What crosses ranks is listed in DESIGN.md section 6.
    int opt_shard_rows = 0;
    bool rows_sharded = false;
    i64 n_loc = 0;
    std::vector<i32> h_loc2glob, h_glob2loc;
    std::vector<int> comm_owner;     // owner rank of community q (0-based)
    DevBuf<i32> loc2glob, glob2loc, comm_loc;
    DevBuf<double> vw_loc;
    std::vector<double> h_vw_loc;
    std::vector<i32> h_gl_off;       // landmark -> GLOBAL member counts as a prefix (N + 1); h_mem_off holds the LOCAL ones
    std::vector<int> lm_owner;       // owner rank of every landmark of the last run
    DevBuf<i32> v2l_loc;             // landmark of every local row
    bool wedges_block_only = false;  // the N x N landmark-pair matrix was reduce-scattered: only this rank's row block is summed
    bool unit_weights = false;
    bool blocked_ready = false; // blocked copy of the edge list (edge pass) matches src/dst
    DevBuf<i32> src, dst;     // 0-based
    DevBuf<double> w;         // edge weights
    DevBuf<double> Xr;        // n x d row-major (node-major)
    DevBuf<double> Xc;        // dpad x ldn feature-major, zero padded, centred at the global mean (diameter kernel)
    DevBuf<double> rnorm;     // ldn: squared norm of the centred rows (0 in the padding)
    i64 ldn = 0, dpad = 0;
    bool centred_ready = false;
    DevBuf<i32> comm;         // n, 0-based
    DevBuf<unsigned short> comm16; // the same as uint16 when C < 65536 (2 MB at n = 10^6: L2-resident gather table)
    DevBuf<double> vw;        // n
    // blocked copy of the edge list for the cluster-pair scatter (kernels_scatter.hip)
    DevBuf<unsigned> be_edge;           // m words (u mod 32768) << 16 | (v mod 32768), grouped by (block of u, block of v)
    DevBuf<double> be_w, be_wkeys, be_diag; // weights in that order (weighted lists); pass-1 outputs
    DevBuf<i32> be_chunk;               // be_nchunks x {block of u, block of v, first edge, edges}
    i64 be_nchunks = 0;
    i64 stat_layout_build_us = 0;       // wall time of the last build of the blocked copy (one-off per resident graph)
    int be_per = 16;                    // edges per thread of the edge pass the chunks were cut for (kernels_scatter.hip)
    DevBuf<unsigned short> be_keys, be_runoff;
    DevBuf<unsigned short> v2l16;       // uint16 landmark of every vertex (padded like comm16): table of the landmark-pair passes
    DevBuf<unsigned> be_base, be_cursor;
    std::vector<double> h_Xr; // host mirror, row-major (cut rules + RSS run on the host)
    std::vector<i32> h_comm;
    std::vector<double> h_vw;
    std::vector<double> h_w;       // edge weights mirror (sample weights)
    i64 n_comm_max = 0;            // C of the original graph

    // ---- landmark state (output of cge_landmarks_run) ------------------------------------
    bool lm_ready = false;
    int lm_directed = 0;
    i64 N = 0;
    int lm_truncated = 0;
    std::vector<i64> h_v2l;   // 1-based landmark of each vertex
    std::vector<i32> h_v2l0;  // the same, 0-based (upload staging)
    DevBuf<i32> v2l;          // 0-based
    DevBuf<double> lemb;      // N x d row-major
    DevBuf<double> lweight, dii;
    DevBuf<i32> lcomm;        // 0-based
    DevBuf<double> wedges;    // N x N, [a*N + b]
    DevBuf<double> vectC;     // from the original edges
    i64 n_ledges = 0;
    bool wedges_ready = false; // the N x N landmark-pair matrix is built lazily (cge_score does not need it)

    // ---- scratch for host-array wGCL ------------------------------------------------------
    DevBuf<double> s_emb, s_dist, s_vw, s_vectC, s_degin, s_degout;
    DevBuf<i32> s_comm;
    DevBuf<double> auc_part, js_part;
    DevBuf<unsigned> js_counter; // bins_js_kernel's arrival counter (monotonic: js_launches x CGE_PARTIAL_BLOCKS arrivals so far)
    unsigned js_launches = 0;
    // alpha-sweep scratch (grow-only: no hipMalloc/hipFree inside a scoring call after the first)
    DevBuf<double> sw_D, sw_GD, sw_T1, sw_T2, sw_S1, sw_S2, sw_rowbins, sw_vectB, sw_scal, sw_lohi, sw_fitstate, sw_mm;
    DevBuf<int> sw_flags;
    DevBuf<unsigned long long> sw_fring;
    // persistent Chung-Lu fit (kernels_fitp.hip): T double buffer, partial vectors, per-workgroup maxima, barrier words
    DevBuf<double> fp_T, fp_Tsave, fp_P, fp_fpart, fp_fq, fp_Td, fp_flow;
    DevBuf<double> ls_eigscr;            // wide eigen-solver: partial vectors of its tile sweeps (kernels_lm.hip)
    bool bvec_contig = false;            // the score graph of the running sweep has contiguous communities (relabelled)
    bool bvec_blocks = false;            // ... and vect_B is summed by tiles (kernels_fit.hip: bvec_tile_kernel + bvec_bins_kernel)
    int opt_bvec_blocks = 0;             // 1: relabel the score graph of a sweep by community (from 256 vertices on) and sum vect_B by
                                         // tiles (measured slower: profiles/r04_bvec_tiles_ab.txt); 0 (default): the row-bin form
                                         // (relabelled only beyond 8192 vertices)
    DevBuf<i32> sw_bt_fc, sw_bt_ns, sw_bt_base; // per 64-vertex block: first community, communities; per tile: base of its partials
    i64 flow_armed_words = 0;            // > 0: the persistent fit's hand-off slots (that many 4-byte words) are armed by the previous alpha's last launch
    DevBuf<double> sw_bt_part;
    DevBuf<i32> sw_bt_desc;             // per community-pair bin: the positions of (up to four of) its tile partials (k_bins_prepare)
    DevBuf<double> sw_fused_pw;          // ... and the two powers per sample its prologue writes for its epilogue
    DevBuf<char> sw_fused_epi;           // the fused chain's tables, one cge_fit_fused per sample set (wgcl_host.cpp)
    int opt_fit_fused = 1;               // 1 (default): landmark-mode sweeps let the rest of an alpha's chain ride on the fit's launch
    i64 stat_fit_fused = 0;              // alphas of the last sweep that did
    DevBuf<i32> sw_rl_order, sw_rl_comm; // exact mode, N > 8192: the score graph relabelled by community (wgcl_host.cpp)
    DevBuf<double> sw_rl_emb, sw_rl_vec, sw_rl_T;
    DevBuf<unsigned> fp_sync;
    DevBuf<int> fp_flags;
    PinBuf<double> pin_scal;    // the scalars of an alpha (AUC sums, divergences, the fit's verdict), two alphas in flight
    hipEvent_t sweep_ev[2] = {nullptr, nullptr};
    int opt_pow_exp2 = 1; // (1 - D)^alpha as exp2(alpha * log2(1 - D)) with the logarithm computed once per score
    i64 pow_logs_N = 0;   // log2(1 - D) of the current sweep is in sw_Lh / sw_Ll (0: not prepared)
    bool pow_logs_upper = false;
    i64 pow_logs_blocked_N = 0; // ... or, for the fused persistent fit only, tile-blocked (kernels_fit.hip: log_matrix_tiles_kernel)
    DevBuf<double> sw_Lh;
    DevBuf<float> sw_Ll;
    bool opt_exact_relabel = true; // exact mode, N > 8192: relabel the score graph by community (wgcl_host.cpp)
    int opt_shard_samples = 1; // N > 1: 1 = local-score tallies split over the ranks from 10^5 samples on (in-library RCCL), 2 = always, 0 = never
    int opt_shard_forced = 1; // N > 1: the forced per-community phase of runsplit is split over the ranks
    int opt_test_bvec_plain = 0; // testing: vect_B without LDS staging / rows in flight (the forms of very large score graphs)
    int opt_fit_persistent = 0; // 0 auto (score graphs of >= 128 vertices that fit the register file), 1 never, 2 whenever it fits
    i64 opt_fit_max_iters = 2000000; // a Chung-Lu fit that has not met `diff <= delta` (src/divergence.jl:151,434) after this many
                                     // iterations raises CGE_E_ASSERT -- the reference's loop has no bound and would not return
    int opt_fit_test_timeout = 0; // testing: the persistent fit gives up at once, so the fallback path runs
    int opt_fit_test_delay = 0;   // testing: the tile waves of the data-as-signal fits nap this many times (~3 us each) before
                                  // their first load -- start skew, as under contention; results must not change
    i64 stat_lm_batches = 0, stat_lm_rows = 0, stat_lm_splits = 0; // last runsplit: device batches, their rows, groups split
    DevBuf<int> cut_ties;                                          // last runsplit: tasks of the cut rules with a row ON the cut (device counter)
    i64 stat_fit_persistent = 0; // alphas fitted by the persistent kernel in the last sweep
    i64 stat_fit_iters = 0;      // Chung-Lu iterations of the last sweep (all alphas)
    // A hand-off of a persistent fit timed out (e.g. another process holds CUs): the rest of THIS sweep runs one launch per
    // iteration.  Not latched for the life of the context: the next sweep tries the persistent form again, backing off
    // (1, 3, 7, ... sweeps skipped) while the time-outs repeat; stat "fit_persistent_fallbacks" counts them.
    bool fit_persistent_broken = false;
    i64 stat_fit_fallbacks = 0;
    int fit_fallback_streak = 0;
    i64 fit_skip_sweeps = 0;
    DevBuf<i32> sw_cm_off, sw_cm_mem, sw_cm_pos; // community -> members CSR of the score graph and its inverse
    DevBuf<double> sw_zeros, sw_zsum;
    // diameter scratch
    DevBuf<double> mp_recs;  // MaxRec records (3 doubles each)
    DevBuf<i64> mp_count;
    DevBuf<double> mp_rd2, mp_refmu; // reference-point distances / centroids of the pruned diameter
    DevBuf<double> mp_commax;        // per-community maxima of the bound matrix (two-level candidate selection)
    DevBuf<i32> mp_plist;            // surviving community pairs (int2 each)
    DevBuf<i32> mp_lref, mp_refoff, mp_refmem;
    DevBuf<double> gmean;    // global feature mean (the centre used by Xc)
    DevBuf<double> Xs, rns, Ms, mnorm, Pm; // landmark-sorted centred copy, centroids, P matrix
    DevBuf<double> pc_groups;              // per (16-row group, reference point) maxima of the bound pass
    DevBuf<float> Xs32, Ms32;              // fp32 copies: operands of the fp32-MFMA bound pass (upper bounds only)
    DevBuf<unsigned short> Xb16, Mb16;     // two-plane bf16 operands of the bf16-split bound pass
    DevBuf<int> dm_flag;                   // raised by the bf16 gather when a value is unfit for the split
    // exact stage of the pruned diameter: the candidate landmarks' rows gathered per round (centred, feature-major), their
    // norms, source rows and global vertex ids; the seed row of the farthest-point sweep (option shard_rows)
    DevBuf<double> xe, xe_rns, dm_seed;
    DevBuf<i32> xe_pos, xe_glob, xe_sub, dm_soffE;
    int opt_diameter_f32 = 2;              // the point-to-reference maxima: 2 = bf16 matrix pipe, operands split in two terms (K <= 128; else 1),
                                           // 1 = fp32-input MFMA, 0 = fp64 MFMA; 1 and 2 are upper bounds with a rigorous error margin
    DevBuf<i32> pos2node, sub_land, dm_soff, dm_memoff, dm_mem;
    DevBuf<double> bound_list;             // BoundRec records (2 doubles each)
    DevBuf<i32> tile_list;
    std::vector<i32> h_mem_off, h_mem;     // landmark -> members (ascending vertex id), host copy
    DevBuf<uint64_t> uniq_hash;            // row hashes of the unique-row check
    DevBuf<unsigned long long> uniq_table; // ... and the device set that counts the distinct ones
    int opt_diameter = 0;                  // 0 auto (pruned with brute-force fallback), 1 brute force, 2 pruned only
    i64 stat_cand_pairs = 0, stat_cand_tiles = 0; // last pruned run
    int stat_diameter_path = 0;            // 1 brute, 2 pruned
    double stat_last_hi = 0.0;
    i64 stat_hi_i = -1, stat_hi_j = -1; // its arg-max pair (0-based vertex ids)
    int stat_bound_pass = 0;            // bound pass of the last pruned diameter: 2 bf16-split, 1 fp32 MFMA, 0 fp64 MFMA
    i64 stat_nref = 0; // reference points of the last pruned diameter (communities or landmarks)
    // scratch of the batched split engine (landmarks_host.cpp)
    DevBuf<i32> ls_rows, ls_row_task, ls_ct, ls_cb, ls_ce, ls_tco;
    DevBuf<double> ls_part, ls_mean, ls_sw, ls_cov, ls_vec, ls_z, ls_sums;
    DevBuf<unsigned char> ls_side, ls_state;
    DevBuf<double> ls_params; // per-task round parameters of the rss rule
    PinBuf<double> pin_sums, pin_params, pin_cmeans;
    PinBuf<i32> pin_rows[2], pin_row_task[2]; // [0]: cluster upload staging, [1]: rows of a host-built (generic rss path) batch
    // sorted-prefix rss path
    DevBuf<i32> sp_srows, sp_meta, sp_rounds, sp_tro, sp_perm, sp_status, sort_idx, sort_idx2, sort_keys32, sort_k32b, sort_cnt;
    DevBuf<unsigned char> sort_keys8;
    // member lists of the landmark phase: a group is a range of this arena (vertex ids, reference order)
    DevBuf<i32> lm_arena;
    i64 lm_arena_used = 0;
    DevBuf<double> lm_means; // weighted means of groups, known from their parents' splits (d doubles each)
    i64 lm_means_used = 0;
    // covariances of the groups that have been split (d*d doubles each, about the group's own mean): a child's covariance is
    // its parent's minus its sibling's, so only the smaller child of a pair is summed over its rows (landmarks_host.cpp)
    DevBuf<i64> ls_moff;
    PinBuf<i64> pin_moff;
    // small host <-> device tables of a landmark batch travel packed: one pinned staging area, one copy, one kernel that
    // scatters (gathers) the 4-byte words to (from) their device arrays (landmarks_host.cpp: WordPacker)
    PinBuf<i32> pin_tab[2], pin_res;
    DevBuf<i32> dev_tab, dev_res;
    hipEvent_t tab_ev[2] = {nullptr, nullptr};
    int tab_slot = 0;
    bool lm_index_on_device = false; // lm_memoff / lm_mem mirror h_mem_off / h_mem (set by runsplit, cleared when the host rebuilds the index)
    DevBuf<i32> ls_toff, ls_nlow, lm_goff, lm_glen, lm_mem, lm_memoff; // task arena offsets, low-child counts, final groups, landmark index
    DevBuf<unsigned char> ls_keys;
    PinBuf<i32> pin_small;
    DevBuf<unsigned char> sort_tmp;
    DevBuf<unsigned long long> sort_k64; // sorted 4096-row pieces of the long groups on their way to the rank merge

    DevBuf<double> sp_zs, sp_ctot, sp_coff, sp_prefix, sp_vals;
    DevBuf<double> r2_F, r2_ck;
    i64 r2_rows = 0;            // rows of the batch the rss2 kernels are about to see

    int opt_landmark_edges = 0;  // 1: cge_score builds the N x N landmark-pair matrix too (what landmarks() returns)
    // grow-only scratch of per-score helpers (no hipMalloc / hipFree inside a scoring call after the first: a hipFree waits
    // for every stream of the device)
    DevBuf<i32> epd_i, s_star;
    DevBuf<double> epd_d;
    DevBuf<i64> wed_cnt;
    DevBuf<double> cc_dense; // dense C x C stage of vect_C beyond 2048 communities (tiled two-pass form)
    DevBuf<unsigned> samp_attempt;
    DevBuf<i32> samp_todo_a, samp_todo_b, samp_hit, samp_flag;
    DevBuf<unsigned long long> samp_table, samp_count;
    // a draw of the device sampler whose first round has been enqueued but not looked at yet (k_draw_samples_begin / _finish)
    struct DrawPending {
        bool on = false;
        i64 seed = 0, stream_id = 0, S = 0, cnt = 0;
        int directed = 0, round = 0;
        i32 *d_pos = nullptr, *d_ni = nullptr, *d_nj = nullptr;
        const i32 *todo = nullptr;
        i32 *next = nullptr;
    } samp_pending;
    PinBuf<unsigned long long> samp_pin_cnt;
    hipEvent_t samp_ev = nullptr;
    std::function<void()> after_unique; // cge_score: work to enqueue behind the first synchronisation of the landmark phase (once)
    SampleSet smp;                 // library-drawn samples of the running score
    std::vector<std::unique_ptr<DevSamples>> dsets; // their device form (wgcl_host.cpp)

    // ---- profiling -------------------------------------------------------------------------
    bool profiling = false;
    std::vector<std::string> profile_only; // empty: every timer
    std::map<std::string, KernelTimer> timers;
    Phase phases;
};

// Bracket a launch with events when profiling is on.
struct ScopedKernelTimer {
    cge_ctx *c;
    KernelTimer *t = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedKernelTimer(cge_ctx *ctx, const char *name) : c(ctx) {
        if (!c->profiling) return;
        if (!c->profile_only.empty()) {
            bool hit = false;
            for (const std::string &n : c->profile_only) hit = hit || n == name;
            if (!hit) return;
        }
        t = &c->timers[name];
        a = take_event();
        b = take_event();
        (void)hipEventRecord(a, c->stream);
    }
    hipEvent_t take_event() { // events are recycled by flush_timers: no create/destroy in the timed region
        if (!c->event_pool.empty()) {
            hipEvent_t e = c->event_pool.back();
            c->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    ~ScopedKernelTimer() {
        if (!t) return;
        (void)hipEventRecord(b, c->stream);
        t->pending.push_back({a, b});
        t->launches++;
    }
};

#define CGE_FIT_TIMEOUT_TICKS 100000000LL // 1 s of the 100 MHz wall clock, re-armed at every Chung-Lu iteration
static inline void note_fit_fallback(cge_ctx *c) {
    if (c->opt_fit_test_timeout) return; // the testing hook abandons on purpose
    c->fit_persistent_broken = true;
    if (c->stat_fit_fallbacks++ == 0)
        fprintf(stderr, "libcge_hip: a persistent Chung-Lu fit timed out waiting for another workgroup (is the GPU shared?); "
                        "falling back to one launch per iteration for this sweep (stat fit_persistent_fallbacks)\n");
    c->fit_fallback_streak = std::min(c->fit_fallback_streak + 1, 6);
    c->fit_skip_sweeps = (1LL << (c->fit_fallback_streak - 1)) - 1;
}
// called at the start of a sweep: decide whether the persistent form is tried again
static inline void fit_sweep_begin(cge_ctx *c) {
    if (!c->fit_persistent_broken) return;
    if (c->fit_skip_sweeps > 0) { c->fit_skip_sweeps--; return; }
    c->fit_persistent_broken = false;
}

// the row space the landmark phase works in: all vertices, or this rank's rows (option shard_rows)
static inline i64 lm_rows(const cge_ctx *c) { return c->rows_sharded ? c->n_loc : c->n; }
static inline const double *lm_vw(const cge_ctx *c) { return c->rows_sharded ? c->vw_loc.p : c->vw.p; }
static inline const i32 *lm_comm(const cge_ctx *c) { return c->rows_sharded ? c->comm_loc.p : c->comm.p; }
static inline const double *lm_hvw(const cge_ctx *c) { return c->rows_sharded ? c->h_vw_loc.data() : c->h_vw.data(); }

static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static inline i64 packed_len(i64 n) { return n * (n + 1) / 2; }
// 0-based packed upper-triangular index, i <= j  (== idx(n,i+1,j+1) - 1, src/auxilary.jl:57-59)
static inline i64 pidx0(i64 n, i64 i, i64 j) { return n * i - i * (i - 1) / 2 + (j - i); }

static inline unsigned grid_for(i64 work, int block, i64 cap = 256 * 8) {
    i64 g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

#define CGE_STAGE_BYTES ((size_t)64 << 20) // one staging buffer of the uploads (profiles/r05_microbench_upload.txt: 64 MiB chunks reach the link's 54-57 GB/s)
void cge_ensure_host_embedding(cge_ctx *c); // capi.cpp: fetch the host mirror of Xr on first demand

// ---- kernels_*.hip entry points (host launchers) ---------------------------------------------
// layout
void k_transpose_to_rowmajor(cge_ctx *c, const double *Xcol, double *Xrow, i64 n, i64 d);
void k_transpose_piece(cge_ctx *c, const double *piece, double *Xrow, i64 rows, i64 cols, i64 i0, i64 k0, i64 d); // rows [i0, i0 + rows) x columns [k0, k0 + cols), column-major piece -> its place in the row-major matrix
void k_row_hash(cge_ctx *c, const double *Xrow, uint64_t *hash, i64 n, i64 d);
i64 k_count_distinct(cge_ctx *c, const uint64_t *hash, i64 n); // distinct values among the hashes (device set; synchronises)
// landmark split primitives (batched over tasks; rows = concatenated 0-based vertex ids)
void k_group_mean(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const i32 *chunk_task,
                  const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks,
                  i64 d, double *part, double *mean, double *sw);
void k_group_cov(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const i32 *chunk_task,
                 const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks,
                 i64 d, const double *mean, double *part, double *cov);
void k_group_side_sums(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const unsigned char *side,
                       const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks,
                       i64 d, double *part, double *out);
void k_side_values_means(cge_ctx *c, const double *sums, i64 n_tasks, i64 d, double *vals, double *means);
void k_sorted_prefix(cge_ctx *c, const double *Xr, const double *vw, const i32 *srows, const i32 *chunk_beg,
                     const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks, i64 d, double *ctot,
                     double *coff, double *prefix);
void k_rss2_walk(cge_ctx *c, const double *Xr, const double *vw, const i32 *srows, const i32 *task_row_off, i64 n_tasks, i64 d,
                 i32 *meta, double *vals, double *cmeans);
void k_cut_sides(cge_ctx *c, const double *z, const double *zs, const i32 *task_row_off, i64 n_tasks, int use_median,
                 unsigned char *side, i32 *nlow_out = nullptr, // nlow_out: rows of the low side per task (replaces k_side_counts)
                 int *tie_tasks = nullptr);                     // tie_tasks: counts the tasks that held a row with z == cut
void k_rss_rounds(cge_ctx *c, const double *Xr, const double *vw, const i32 *srows, const double *zs,
                  const i32 *task_row_off, const i32 *task_chunk_off, const double *prefix, const double *coff,
                  i64 n_tasks, i64 d, i32 *meta, i32 *rounds, double *vals, double *cmeans /* [task][2][d] */);
#define CGE_RR_MAXROUNDS 63
#define CGE_CHUNK_ROWS 1024 // rows per chunk of a batch (build_batch); the rounds kernel uses r >> 10
#define CGE_PARTIAL_BLOCKS 64 // block partials of the JS / AUC reductions (summed in block order)
#define CGE_PREFIX_STRIDE 8 // the sorted-order WSSE prefix is stored every 8th row of a chunk (CGE_CHUNK_ROWS % 8 == 0)
void k_gather_means(cge_ctx *c, const double *arena, const i64 *off, i64 T, i64 d, double *mean);
void k_gather_rows(cge_ctx *c, const i32 *arena, const i32 *task_off, const i32 *task_row_off, const i32 *chunk_task,
                   const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, i32 *rows, i32 *row_task);
void k_rss_child_keys(cge_ctx *c, const i32 *perm, const i32 *row_task, const i32 *task_row_off, const i32 *meta,
                      const i32 *rounds, i64 R, i64 T, unsigned char *keys, i32 *nlow);
void k_side_counts(cge_ctx *c, const unsigned char *side, const i32 *task_row_off, i64 T, i32 *nlow);
void k_sort_children(cge_ctx *c, const unsigned char *keys, const i32 *rows, const i32 *task_row_off, const i32 *chunk_beg,
                     const i32 *chunk_end, const i32 *task_chunk_off, i64 n_chunks, i64 R, i64 T, int key_bits, i32 *out);
i64 k_groups_to_index(cge_ctx *c, const i32 *arena, const i32 *goff, const i32 *glen, i64 N, i64 n, i32 *v2l, i32 *mem);
void k_segmented_sort_z(cge_ctx *c, const double *z, const i32 *rows, const i32 *row_task, const i32 *task_row_off,
                        i64 R, i64 T, double *zs, i32 *perm, i32 *srows, i32 *status, i64 max_len = 0 /* longest group, 0: unknown */);
void k_rss_side(cge_ctx *c, const double *z, const i32 *row_task, i64 n_rows, const double *params,
                unsigned char *state, unsigned char *side);
bool k_group_eig(cge_ctx *c, const double *cov, i64 n_tasks, i64 d, double *vec);
void k_group_project(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const i32 *row_task, i64 n_rows,
                     i64 d, const double *mean, const double *vec, double *z);
// landmark aggregation (src/landmarks.jl:387-430) with CSR landmark -> members (ascending vertex id)
void k_landmark_aggregate(cge_ctx *c, const double *Xr, const double *vw, const i32 *comm, const i32 *mem_off,
                          const i32 *mem, i64 N, i64 d, double *lemb, double *lweight, double *dii, i32 *lcomm);
// per-edge scatter
bool k_edge_scatter_blocked_applies(const cge_ctx *c, i64 C);
bool k_blocked_edges_possible(const cge_ctx *c);
void k_pack_upper(cge_ctx *c, const double *dense, i64 C, double *packed);
bool k_build_blocked_edges(cge_ctx *c);
void k_edge_scatter_blocked(cge_ctx *c, i64 c0, i64 c1, i64 C, int directed, double *vectC);
// the N x N landmark-pair matrix by the tiled two-pass form (kernels_scatter.hip); false: does not apply
bool k_wedge_scatter_blocked(cge_ctx *c, const i32 *v2l, i64 N, i64 c0, i64 c1, int directed, double *wedges, i64 *positive,
                             const char *timer = "edge_scatter_wedges");
#define CGE_COMM16_PAD 32768 // the uint16 community table is padded to a multiple of the edge pass's vertex block
void k_edge_scatter(cge_ctx *c, const i32 *src, const i32 *dst, const double *w, i64 e0, i64 e1, const i32 *v2l,
                    const i32 *comm, i64 N, i64 C, int directed, double *wedges, double *vectC);
void k_edge_degrees(cge_ctx *c, const i32 *src, const i32 *dst, const double *w, i64 m, double *deg_out,
                    double *deg_in, i32 *star);
void k_wedge_degrees(cge_ctx *c, const double *wedges, i64 N, double *deg_out, double *deg_in, i32 *star);
void k_compact_count(cge_ctx *c, const double *wedges, i64 N, int directed, i64 *count, i64 row0 = 0, i64 row1 = -1);
void k_wedge_degrees_block(cge_ctx *c, const double *wedges, i64 N, i64 row0, i64 row1, double *out3N);
void k_degrees_unpack(cge_ctx *c, const double *in3N, i64 N, double *deg_out, double *deg_in, i32 *star);
void k_louvain_level1(cge_ctx *c, i64 *comm_out_host, i64 *n_comm, double *quality, i64 *rounds); // kernels_louvain.hip
// distances
void k_dist_matrix(cge_ctx *c, const double *emb, const double *diag, i64 N, i64 d, double *D);
void k_minmax_upper(cge_ctx *c, const double *D, i64 N, double *lo_hi);
void k_normalise(cge_ctx *c, double *D, i64 N, const double *lo_hi);
void k_max_pair(cge_ctx *c, const double *Xc, const double *rnorm, i64 n, i64 ldn, i64 dpad, int part, int nparts,
                double *best_val, i64 *best_i, i64 *best_j);
void k_pair_dist(cge_ctx *c, const double *Xr, i64 d, const i32 *pi, const i32 *pj, i64 S, double inv_scale_den,
                 double *out);
void k_diameter_layout(cge_ctx *c, const i32 *mem_off, const i32 *mem, const i32 *soff, i64 N, i32 *pos2node, i32 *sub_land,
                       i64 n_sub);
void cge_allreduce_dev(cge_ctx *c, double *dev, i64 count, int op /*0 sum, 1 max*/); // no-op without collectives
void cge_rccl_allreduce(cge_ctx *c, void *dev, i64 count, int op); // collectives.cpp: in place, on the ctx stream
bool cge_rccl_allgather(cge_ctx *c, void *dev, i64 words_per_rank); // in place (rank r's piece at r * words_per_rank); false: no such symbol
// all-gather of 8-byte words in place: `buf` holds world pieces of `words_per_rank`, this rank's piece is filled in
void cge_allgather_dev(cge_ctx *c, double *buf, i64 words_per_rank);
bool cge_rccl_reduce_scatter(cge_ctx *c, void *dev, i64 words_per_rank); // in place: rank r's sums land at r * words_per_rank; false: no such symbol
double cge_allreduce_scalar_max(cge_ctx *c, double v); // max of one double over the ranks (synchronises); v itself without collectives
// can the exchange buffer hold `need` doubles?  With the in-library communicator a library-owned buffer grows on demand
// (contents are not preserved); a caller-provided one (cge_set_exchange_buffer, the hook path) is what it is.
bool cge_exchange_fits(cge_ctx *c, size_t need);
void k_pcent(cge_ctx *c, const double *Xs, const double *rns, i64 lds_rows, const double *Ms, const double *mnorm,
             i64 ldm, i64 n_land, i64 N, i64 dpad, const i32 *soff, double *P, int part = 0, int nparts = 1);
void k_pair_list(cge_ctx *c, const double *Xs, const double *rns, i64 lds_rows, i64 npos, i64 dpad, const void *tiles,
                 i64 ntiles, double *best_val, i64 *best_i, i64 *best_j);
i64 k_bound_select(cge_ctx *c, const double *Q, const i32 *lref, const double *mu_ref, i64 N, i64 nref, i64 d, double L,
                   void *list, i64 cap, const i32 *ref_off = nullptr, const i32 *ref_mem = nullptr, const double *Ms_fm = nullptr,
                   i64 dpad = 0, i64 ldm = 0,
                   bool rd2_ready = false);
void k_ref_dist2_fm(cge_ctx *c, const double *Ms_fm, i64 nref, i64 dpad, i64 ldm);
i64 k_argmax_mapped(cge_ctx *c, const double *v, i64 n, const i32 *map, double *val = nullptr);
void k_pair_local_idx(cge_ctx *c, const i32 *pi, const i32 *pj, i64 S, const i32 *glob2loc, i32 *idx);
void k_pair_dist_rows(cge_ctx *c, const double *B, i64 d, const i32 *pi, const i32 *pj, i64 S, double den, double *out);
void k_position_ids(cge_ctx *c, const i32 *pos2node, const i32 *loc2glob, i64 npos, i32 *ids); // map[argmax v] (synchronises the stream)
void k_ref_centroids(cge_ctx *c, const double *mu, const double *lw, const i32 *ref_off, const i32 *ref_mem, i64 nref,
                     i64 d, double *out);
void k_farthest(cge_ctx *c, const double *Xr, i64 n, i64 d, i64 src, double *best_val, i64 *best_i);
void k_farthest_enqueue(cge_ctx *c, const double *Xr, i64 n, i64 d, i64 src, const double *srow = nullptr); // launch only (c->stream) ...
void k_farthest_collect(cge_ctx *c, double *best_val, i64 *best_i);            // ... and its result (synchronises c->stream)
void k_col_mean(cge_ctx *c, const double *Xrow, i64 n, i64 d, double *mean, double sums_only = 0.0);
void k_scale_vector(cge_ctx *c, double *v, i64 n, double f);
void k_pack_landmarks(cge_ctx *c, double *lemb, double *lweight, double *dii, i32 *lcomm, i64 N, i64 d, double *X, int unpack);
void k_scatter_u64(cge_ctx *c, const uint64_t *src, const i32 *idx, i64 cnt, uint64_t *dst);
void k_scatter_i32(cge_ctx *c, const i32 *src, const i32 *idx, i64 cnt, i32 add, i32 *dst); // dst[idx[i]] = src[i] + add
void k_add_i32(cge_ctx *c, const i32 *src, i64 n, i32 add, i32 *dst);
void k_gather_rows_f64(cge_ctx *c, const double *X, i64 n, i64 d, int row_major, const i32 *idx, i64 cnt, double *out);
void k_gather_centre_fm(cge_ctx *c, const double *src_rowmajor, const i32 *idx, const double *mean, double *dst,
                        double *rnorm, i64 npos, i64 d, i64 ld, i64 dpad, float *dst32 = nullptr,
                        unsigned short *planes = nullptr, i64 KP = 0, int *flag = nullptr); // planes: two bf16 terms, row-major [2][ld][KP]
void k_pcent_f32(cge_ctx *c, const float *Xs32, const double *rns, i64 lds_rows, const float *Ms32, const double *mnorm,
                 i64 ldm, i64 n_land, i64 N, i64 dpad, const i32 *sub_land, double *P, int part = 0, int nparts = 1);
// the bound pass on the bf16 matrix pipe with two-term operands (kernels_dist.hip (2c))
bool k_pcent_bf16_applies(i64 dpad);
void k_pcent_bf16(cge_ctx *c, const unsigned short *Xb, const double *rns, i64 lds_rows, const unsigned short *Mb,
                  const double *mnorm, i64 ldm, i64 n_land, i64 N, i64 KP, const i32 *sub_land, double *P, int part = 0, int nparts = 1);
// alpha sweep
void k_wave_tree_test(cge_ctx *c, const double *x, i64 n_rows, double *out_ref, double *out_new); // testing hook
void k_copy_segments(cge_ctx *c, const i32 *src, const i64 *seg, i64 nseg, i32 *dst);
void k_permute_rows(cge_ctx *c, const double *src, const i32 *order, i64 n, i64 width, double *dst); // dst[q] = src[order[q]]
void k_permute_i32(cge_ctx *c, const i32 *src, const i32 *order, i64 n, i32 *dst);
void k_remap_i32(cge_ctx *c, i32 *idx, const i32 *map, i64 n); // idx[k] = map[idx[k]]
#define CGE_WORD_SEGS 8
void k_copy_words(cge_ctx *c, int nseg, void *const *dst, const void *const *src, const i64 *words); // 4-byte words, nseg <= 8
// Several small host arrays -> their device arrays with ONE copy: the words are packed into a pinned staging buffer
// (two of them alternate, each guarded by an event), copied to a device staging area and scattered by one kernel.
// Every copy of its own from pageable memory costs ~20 us of idle stream; a batch has seven of them.
struct WordPacker { // (landmarks_host.cpp: the tables of a batch; wgcl_host.cpp: the tables of a sweep)
    cge_ctx *c;
    std::vector<void *> dst;
    std::vector<const void *> src;
    std::vector<i64> words;
    explicit WordPacker(cge_ctx *c_) : c(c_) {}
    template <class T>
    void add(T *device, const T *host, i64 count) {
        static_assert(sizeof(T) % 4 == 0, "4-byte words");
        if (count <= 0) return;
        dst.push_back(device);
        src.push_back(host);
        words.push_back(count * (i64)(sizeof(T) / 4));
    }
    void flush() {
        if (dst.empty()) return;
        i64 tot = 0;
        for (i64 w : words) tot += w + (w & 1); // keep 8-byte items aligned
        const int slot = c->tab_slot;
        c->tab_slot ^= 1;
        HIP_CHECK(hipEventSynchronize(c->tab_ev[slot])); // the copy that last read this staging buffer is done
        c->pin_tab[slot].ensure((size_t)tot);
        c->dev_tab.ensure((size_t)tot);
        i32 *h = c->pin_tab[slot].p;
        std::vector<const void *> dsrc(dst.size());
        i64 pos = 0;
        for (size_t q = 0; q < dst.size(); q++) {
            std::memcpy(h + pos, src[q], (size_t)words[q] * 4);
            dsrc[q] = c->dev_tab.p + pos;
            pos += words[q] + (words[q] & 1);
        }
        HIP_CHECK(hipMemcpyAsync(c->dev_tab.p, h, sizeof(i32) * (size_t)tot, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipEventRecord(c->tab_ev[slot], c->stream));
        for (size_t q0 = 0; q0 < dst.size(); q0 += CGE_WORD_SEGS) {
            const int n = (int)std::min<size_t>(CGE_WORD_SEGS, dst.size() - q0);
            k_copy_words(c, n, &dst[q0], &dsrc[q0], &words[q0]);
        }
        dst.clear(); src.clear(); words.clear();
    }
};
void k_gather_means_slots(cge_ctx *c, const double *arena, const i64 *off, const i64 *slot, i64 T, i64 d, i64 stride, i64 lead,
                          double *dst);
void k_pow_prepare(cge_ctx *c, const double *D, i64 N, bool upper_only, bool blocked = false);
void k_pow_matrix(cge_ctx *c, const double *D, i64 N, double alpha, double *GD, bool upper_only = false);
void k_pow_test(cge_ctx *c, const double *x, i64 n, double alpha, int method, double *out);
// What rides on the launch of the undirected persistent fit in a landmark-mode sweep relabelled by community (round 5): the
// power matrix computed in the prologue from the stored logarithm, vect_B's tile partials and the local score's tallies in the
// epilogue (kernels_fitp.hip: fit_flow_kernel<.., true>).  partial / auc_part == nullptr in the host copy: that part is not
// wanted this alpha (the device copy, written once per sweep and sample set, always holds both).
#define CGE_FLOW_NP 32 // runs of one community inside an 8-column chunk ("pieces") per 64-vertex block at most
struct cge_fit_fused {
    const double *Lh; const float *Ll; double alpha;   // host copy only: log2(1 - D) in two parts, this alpha
    const i32 *comm, *fc, *ns, *base; double *partial; // vect_B by tiles (relabelled sweep: communities are ranges of vertices)
    // the local score's samples, prepared once per sweep and sample set (k_auc_prepare): everything that depends neither on
    // alpha nor on T -- T's indices in the sweep's numbering [4][S], the factors vw_i, lw_li, vw_j, lw_lj, ... [8][S], the
    // weight sums of the CGE_PARTIAL_BLOCKS blocks -- and per alpha the two powers [2][S], which the fit's prologue writes
    i64 S; const double *dpos, *dneg, *wts; const i32 *aidx; const double *afac, *aden; double *apw; double *auc_part;
};
bool k_fit_flow_enqueue(cge_ctx *c, const double *GD, i64 N, const double *T0, double *Tout, i64 Tld, const double *w,
                        double eps, double delta, int *dev_flags, const cge_fit_fused *ff = nullptr,
                        const cge_fit_fused *ff_dev = nullptr); // ff_dev: the device copy of *ff the epilogue reads (alpha, partial and
                                                                // auc_part are taken from *ff: they change per alpha)
bool k_fit_flow_fused_applies(cge_ctx *c, i64 N); // the geometry the fused instances exist for (one tile per wave, one quarter block per workgroup)
void k_bins_prepare(cge_ctx *c, const i32 *cm_off, i64 N, i64 C, i64 n_partials); // once per sweep, behind the tile tables
// what bins_js_kernel, the last launch of an alpha's chain, does on the way out: the alpha's scalars straight into the host's
// pinned slot (host_out; scal = their device copy, [res_js, res_js + 2 x CGE_PARTIAL_BLOCKS) = the JS partials this launch
// computes), and the sentinel fill of the next alpha's persistent fit (arm: 16-byte units)
struct cge_chain_tail {
    double *host_out; const double *scal; int res_js, res_len;
    uint4 *arm; i64 arm_n16; unsigned arm_word;
};
void k_bins_js(cge_ctx *c, const i32 *cm_off, i64 N, i64 C, const double *vC, double *vectB, int n_modes, double *fpart,
               const cge_chain_tail *tail = nullptr);
bool k_fit_flow_arm_region(cge_ctx *c, i64 N, i64 Tld, uint4 **ptr, i64 *n16, unsigned *word); // the fit's hand-off slots (for the tail above)
void k_bvec_tiles(cge_ctx *c, const double *GD, const double *Ta, const double *Tb, const i32 *cm_off, i64 N, int directed); // the tile partials only
void k_auc_prepare(cge_ctx *c, const i32 *v2l, const i32 *old2new, const double *vw_orig, const double *lweight, const i32 *pi,
                   const i32 *pj, const i32 *ni, const i32 *nj, const double *wts, i64 S, i32 *aidx, double *afac, double *aden);
void k_bvec_bins(cge_ctx *c, const i32 *cm_off, i64 N, i64 C, int directed, double *vectB); // folds the tile partials into vect_B
bool k_fit_persistent_dir(cge_ctx *c, const double *GD, i64 N, double *Tin, double *Tout, const double *deg_in,
                          const double *deg_out, double eps0, double f0, double delta, i64 *iters, int *dev_flags = nullptr,
                          bool *enqueued_only = nullptr);
void k_fit_verdict(cge_ctx *c, const int *flags, int async, double *out); // 1.0 when an enqueued fit was abandoned
// the same iteration over the upper 64 x 64 tiles only (kernels_fitp.hip): half the matrix traffic
void k_fit_sym_step(cge_ctx *c, const double *GD, const double *Tin, double *Tout, const double *w, i64 N, double eps,
                double delta, int k, unsigned long long *fring, int *done, int *iters);
void k_fit_symv_dir(cge_ctx *c, const double *GD, const double *Tin, const double *Tout, i64 N, double *Sin,
                    double *Sout, const int *done);
void k_fit_update_dir(cge_ctx *c, double *Tin, double *Tout, const double *Sin, const double *Sout,
                      const double *deg_in, const double *deg_out, i64 N, double delta, int *done, int *iters,
                      double *state /* [0]=eps,[1]=diff */);
void k_bvec(cge_ctx *c, const double *GD, const double *Ta, const double *Tb, const i32 *cm_pos, const i32 *cm_off,
            const i32 *cm_mem, i64 N, i64 C, int directed, double *rowbins, double *vectB);
void k_js(cge_ctx *c, const double *vC, const double *vB, i64 len, i64 C, int directed, int mode /*0 all,1 int,2 ext*/,
          double *out, double *partials = nullptr);
void k_auc_landmark(cge_ctx *c, const double *Ta, const double *Tb, const i32 *v2l, const double *vw_orig,
                    const double *lweight, const i32 *pi, const i32 *pj, const i32 *ni, const i32 *nj,
                    const double *dpos, const double *dneg, const double *wts, i64 S, double alpha, double *out2, double *partials = nullptr);
void k_auc_exact(cge_ctx *c, const double *GD, const double *Ta, const double *Tb, i64 N, const i32 *pi,
                 const i32 *pj, const i32 *ni, const i32 *nj, const double *wts, i64 S, double *out2, double *partials = nullptr);
void k_mark_edge_hits(cge_ctx *c, const i32 *src, const i32 *dst, i64 m, int directed, const uint64_t *table,
                      i64 table_size, i32 *hit);

// ---- host modules -----------------------------------------------------------------------------
// landmarks_host.cpp
void host_runsplit(cge_ctx *c, const i64 *cl_flat, const i64 *cl_off, i64 ncl, i64 nland, i64 forced, int method,
                   std::vector<i64> &group_ids /*0-based*/, bool want_index = false); // also fills c->v2l / lm_mem / lm_memoff (device)
void host_eig_top(const double *A, i64 d, double *v); // largest-eigenvalue eigenvector, sign: max |.| component > 0
// diameter_host.cpp
bool host_diameter_pruned(cge_ctx *c, const double *mu, const double *lw, const std::vector<i32> &lcomm, i64 C, i64 N,
                          const std::vector<i32> &mem_off, const std::vector<i32> &mem, int part, int nparts,
                          double *best_d2, i64 *bi, i64 *bj);
// wgcl_host.cpp
// ---- counter-based RNG of the sampler (splitmix64 finaliser over a 4-word counter): the same stream on host and device
__host__ __device__ inline uint64_t cge_sm64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t cge_ctr_rand(uint64_t seed, uint64_t stream, uint64_t k, uint64_t attempt, uint64_t which) {
    uint64_t h = cge_sm64(seed ^ 0x6a09e667f3bcc909ULL);
    h = cge_sm64(h ^ (stream * 0xd1342543de82ef95ULL + 1));
    h = cge_sm64(h ^ (k * 0x2545f4914f6cdd1dULL + 2));
    h = cge_sm64(h ^ (attempt * 0x9e6c63d0676a9a99ULL + 3));
    return cge_sm64(h ^ (which + 4));
}
// uniform integer in [0, range): multiply-high (bias < range / 2^64)
__host__ __device__ inline uint64_t cge_bounded(uint64_t r, uint64_t range) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(r, range);
#else
    return (uint64_t)(((__uint128_t)r * range) >> 64);
#endif
}
// positive / non-edge draws of one sample set on the device (kernels_fit.hip); 0-based i32 outputs of S entries
void k_draw_samples_dev(cge_ctx *c, i64 seed, i64 stream_id, i64 S, int directed, i32 *d_pos, i32 *d_ni, i32 *d_nj);
// the same in two halves: _begin enqueues the draws and the first round of the rejection and returns (no synchronisation);
// _finish looks at the round's verdict and runs whatever rounds are left (a local edge list only: no exchange in between)
void k_draw_samples_begin(cge_ctx *c, i64 seed, i64 stream_id, i64 S, int directed, i32 *d_pos, i32 *d_ni, i32 *d_nj);
void k_draw_samples_finish(cge_ctx *c);
void k_prep_samples(cge_ctx *c, const i32 *pos, const i32 *pos_pairs, const i32 *ni_in, const i32 *nj_in, const i32 *e_src,
                    const i32 *e_dst, const double *e_w, i64 S, int directed, i32 *pi, i32 *pj, i32 *ni, i32 *nj, double *wts);
void host_draw_samples(cge_ctx *c, i64 seed, i64 stream_id, i64 S, int directed, i64 *pos_idx, i64 *neg_i, i64 *neg_j);
void host_pos_draw(i64 seed, i64 stream_id, i64 S, i64 m, i64 *pos_idx);
bool sampler_uses_device(const cge_ctx *c); // large resident graph: the sampler runs on the device
struct OrigView { // original graph pieces needed in landmark mode (device, 0-based)
    i64 n = 0, m = 0;
    const double *Xr = nullptr;
    const double *vw = nullptr;
    const i32 *v2l = nullptr;
    const double *lweight = nullptr; // == score graph vweights
    const i32 *src = nullptr, *dst = nullptr;
    const double *h_w = nullptr; // host edge weights
    double hi = 0.0;             // diameter
    const i32 *h_lcomm = nullptr; // host copy of the landmarks' communities (the score graph's G.comm), when the caller has one
};
void host_wgcl_sweep(cge_ctx *c, const ScoreGraph &G, const OrigView *orig, const i32 *ex_src, const i32 *ex_dst,
                     const double *ex_hw, i64 ex_m, int directed, int split, const SampleSet &smp, double out[7],
                     int *out_len, cge_trace *trace);
