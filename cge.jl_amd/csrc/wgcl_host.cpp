// wgcl_host.cpp -- host side of the alpha sweep (wGCL / wGCL_directed, src/divergence.jl:27-257,
// :282-561) and the local-score sampler.  The host owns only the alpha bookkeeping (best/patience
// counters, :215-223, :242-253) and the batch-wise launch of fit iterations; every O(N^2) loop of the
// reference runs in kernels_fit.hip / kernels_dist.hip.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.hpp"

// the counter-based RNG of the sampler lives in common.hpp (the device draws the same stream)
static inline uint64_t ctr_rand(uint64_t seed, uint64_t stream, uint64_t k, uint64_t attempt, uint64_t which) {
    return cge_ctr_rand(seed, stream, k, attempt, which);
}
static inline uint64_t bounded(uint64_t r, uint64_t range) { return cge_bounded(r, range); }

void host_pos_draw(i64 seed, i64 stream_id, i64 S, i64 m, i64 *pos_idx) {
    for (i64 k = 0; k < S; k++)
        pos_idx[k] = (i64)bounded(ctr_rand((uint64_t)seed, (uint64_t)stream_id, (uint64_t)k, 0, 0), (uint64_t)m) + 1;
}

// `sample(E, S, replace=true)` and `sample(NE, S, replace=true)` (src/divergence.jl:185,194,203,210;
// directed :485,495,505,513) on the RESIDENT graph.  Uniform with replacement over the edge rows and
// over the non-edge pairs (rejection against the resident edge list, checked on the device).
void host_draw_samples(cge_ctx *c, i64 seed, i64 stream_id, i64 S, int directed, i64 *pos_idx, i64 *neg_i, i64 *neg_j) {
    const i64 n = c->n, m = c->m;
    if (m <= 0 || n < 2) CGE_THROW(CGE_E_ARG, "draw_samples: no resident graph");
    const uint64_t sd = (uint64_t)seed, st = (uint64_t)stream_id;
    host_pos_draw(seed, stream_id, S, m, pos_idx);
    // Small graphs (n(n-1) <= 2^25): enumerate NE itself (lexicographic order) and index into it --
    // exact and immune to dense graphs; larger graphs: rejection against the edge list.
    if ((double)n * (double)(n - 1) <= 33554432.0) {
        std::vector<i32> hs(m), hd(m);
        HIP_CHECK(hipMemcpyAsync(hs.data(), c->src.p, sizeof(i32) * m, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipMemcpyAsync(hd.data(), c->dst.p, sizeof(i32) * m, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        std::vector<uint8_t> adj((size_t)n * n, 0);
        for (i64 e = 0; e < m; e++) {
            i64 a = hs[e], b = hd[e];
            if (!directed && a > b) std::swap(a, b);
            adj[(size_t)a * n + b] = 1;
        }
        std::vector<uint32_t> ne;
        ne.reserve((size_t)n * (n - 1) / (directed ? 1 : 2));
        for (i64 i = 0; i < n; i++)
            for (i64 j = directed ? 0 : i + 1; j < n; j++)
                if (i != j && !adj[(size_t)i * n + j]) ne.push_back((uint32_t)(i * n + j));
        if (ne.empty()) CGE_THROW(CGE_E_ARG, "draw_samples: the graph has no non-edges");
        for (i64 k = 0; k < S; k++) {
            const uint32_t code = ne[bounded(ctr_rand(sd, st, (uint64_t)k, 0, 1), (uint64_t)ne.size())];
            neg_i[k] = (i64)(code / n) + 1;
            neg_j[k] = (i64)(code % n) + 1;
        }
        return;
    }
    // large graphs: drawn and rejected against the resident edge list on the device (kernels_fit.hip), copied back here
    DevBuf<i32> d_pos, d_ni, d_nj;
    d_pos.ensure(S); d_ni.ensure(S); d_nj.ensure(S);
    k_draw_samples_dev(c, seed, stream_id, S, directed, d_pos.p, d_ni.p, d_nj.p);
    std::vector<i32> hi_(S), hj_(S);
    HIP_CHECK(hipMemcpyAsync(hi_.data(), d_ni.p, sizeof(i32) * S, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(hj_.data(), d_nj.p, sizeof(i32) * S, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    for (i64 k = 0; k < S; k++) {
        neg_i[k] = (i64)hi_[k] + 1;
        neg_j[k] = (i64)hj_[k] + 1;
    }
}
bool sampler_uses_device(const cge_ctx *c) { return (double)c->n * (double)(c->n - 1) > 33554432.0; }

// ------------------------------------------------------------------------------------------------

// Endpoints / weights of sampled edge rows are gathered on the host from small D2H reads of the
// resident edge arrays (S entries), so no host mirror of the edge list is needed.
static void gather_rows(cge_ctx *c, const i32 *d_arr, const std::vector<i64> &rows0, std::vector<i32> &out,
                        DevBuf<i32> &d_idx, DevBuf<i32> &d_out);

// full_graph_D of sampled vertex pairs (src/divergence.jl:104-114 restricted to the draws), dist()'s own arithmetic.  Option
// shard_rows: a pair's two rows may live on two ranks -- the rows of a chunk of pairs are gathered from their owners into a
// zero-filled buffer (all-reduce of the words: exact) and every rank evaluates the chunk from the gathered rows.
static void sampled_pair_dist(cge_ctx *c, const double *Xr, i64 d, const i32 *pi, const i32 *pj, i64 S, double den, double *out) {
    if (!c->rows_sharded) {
        k_pair_dist(c, Xr, d, pi, pj, S, den, out);
        return;
    }
    const i64 K = std::max<i64>(1024, std::min<i64>(S, ((i64)64 << 20) / (2 * d * 8))); // <= 64 MB of rows per exchange
    DevBuf<double> &B = c->samp_xchg;
    DevBuf<i32> &idx = c->epd_i;
    B.ensure((size_t)2 * K * d);
    idx.ensure(2 * K);
    for (i64 k0 = 0; k0 < S; k0 += K) {
        const i64 kc = std::min(K, S - k0);
        k_pair_local_idx(c, pi + k0, pj + k0, kc, c->glob2loc.p, idx.p);
        k_gather_rows_f64(c, c->Xr.p, c->n_loc, d, 1, idx.p, 2 * kc, B.p);
        cge_allreduce_dev(c, B.p, 2 * kc * d, 2);
        k_pair_dist_rows(c, B.p, d, pi + k0, pj + k0, kc, den, out + k0);
    }
}

void host_wgcl_sweep(cge_ctx *c, const ScoreGraph &G_in, const OrigView *orig, const i32 *ex_src, const i32 *ex_dst,
                     const double *ex_hw, i64 ex_m, int directed, int split, const SampleSet &smp, double out[7],
                     int *out_len, cge_trace *trace) {
    const double delta = 0.001, AlphaMax = 10.0, AlphaStep = 0.25; // :35-37 / :288-290
    ScoreGraph G = G_in; // the per-vertex arrays may be replaced by community-sorted copies (below)
    const i64 N = G.N, C = G.C, d = G.d;
    hipStream_t st = c->stream;
    const i64 vlen = directed ? C * C : packed_len(C);
    if ((double)N * (double)N * 8.0 * 2.2 > 200e9) CGE_THROW(CGE_E_OOM, "score graph with %lld vertices does not fit", (long long)N);

    DevBuf<double> &D = c->sw_D, &GD = c->sw_GD, &T1 = c->sw_T1, &T2 = c->sw_T2, &S1 = c->sw_S1, &S2 = c->sw_S2,
                   &rowbins = c->sw_rowbins, &vectB = c->sw_vectB, &scal = c->sw_scal, &lohi = c->sw_lohi,
                   &fitstate = c->sw_fitstate;
    DevBuf<int> &flags = c->sw_flags; // [0]=done, [1]=iters
    D.ensure((size_t)N * N);
    GD.ensure((size_t)N * N);
    T1.ensure(N); T2.ensure(N); S1.ensure(N); S2.ensure(N);
    rowbins.ensure((size_t)N * C);
    vectB.ensure(vlen);
    scal.ensure(4 * CGE_PARTIAL_BLOCKS + 18); // per alpha: AUC block tallies (+ the shared verdict), JS block sums (two modes), the fit's verdict
    lohi.ensure(2);
    fitstate.ensure(4);
    flags.ensure(4);
    c->sw_fring.ensure(4);

    // community -> members CSR of the score graph
    std::vector<i32> hcomm(N);
    if (orig && orig->h_lcomm) std::memcpy(hcomm.data(), orig->h_lcomm, sizeof(i32) * N); // (the caller read them back for the diameter)
    else {
        HIP_CHECK(hipMemcpyAsync(hcomm.data(), G.comm, sizeof(i32) * N, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    std::vector<i32> cm_off(C + 1, 0), cm_mem(N);
    for (i64 i = 0; i < N; i++) {
        if (hcomm[i] < 0 || hcomm[i] >= C) CGE_THROW(CGE_E_ARG, "community id out of range");
        cm_off[hcomm[i] + 1]++;
    }
    for (i64 q = 0; q < C; q++) cm_off[q + 1] += cm_off[q];
    {
        std::vector<i32> cur(cm_off.begin(), cm_off.end() - 1);
        for (i64 i = 0; i < N; i++) cm_mem[cur[hcomm[i]]++] = (i32)i;
    }
    // Exact mode beyond the LDS-staged vect_B (N > 8192): the score graph is RELABELLED so that every community is a range
    // of consecutive vertices (members keep their ascending order, so every community sum adds in the reference's order).
    // vect_B then reads its members as contiguous runs of a row -- with the vertex ids of a real graph (no relation to
    // the communities) it was a 8-byte gather per element, 93 ms per alpha at n = 60 000 against 3 ms for the stream.
    // Only what is indexed by vertex moves: embedding rows, weights / degrees, communities, the sampled pairs.
    DevBuf<i32> d_old2new;
    // (also the landmark graph of a landmark-mode score with more than 8192 landmarks -- config 5 has 12 000; the local
    // score then reads T through the landmark ids of the original numbering, see the un-permuted copy in the sweep)
    // Option bvec_blocks = 1: an exact-mode sweep is relabelled from 256 vertices on and vect_B is summed BY TILES (kernels_fit.hip:
    // bvec_tile_kernel + bins: GD read once, no row bins) instead of row bins + row sums + fold (the same speed there,
    // profiles/r04_bvec_tiles_ab.txt); landmark-mode sweeps always do (below).
    // (beyond 8192 vertices the sweep is relabelled anyway and the staged row-bin kernel no longer fits LDS: there the tile
    // form replaces the plain gather -- config 5, N = 12 000: 1.1 ms per alpha for the row bins alone)
    const bool blocks_req = (c->opt_bvec_blocks || N > 8192) && c->opt_exact_relabel && N >= 256 && C >= 2 &&
                            !c->opt_test_bvec_plain;
    // Round 5, the default in landmark mode wherever the undirected persistent fit runs with one tile per wave: the rest of
    // an alpha's chain RIDES ON THE FIT'S LAUNCH (kernels_fitp.hip, fit_flow_kernel<.., true>: the power matrix in its
    // prologue, vect_B's tile partials and the local score's tallies in its epilogue).  It needs the relabelled sweep and the
    // tile tables below; option "fit_fused" = 0 keeps the separate launches (the cross-check of the parity tests).
    // Every undirected landmark-mode sweep of >= 256 landmarks is relabelled and sums vect_B by tiles then, whichever form of the
    // fit runs (so that all forms add in the same order and give the same bits); the fused launch itself needs the default
    // persistent form with one tile per wave.
    const bool tiles_req = orig != nullptr && !directed && c->opt_fit_fused && c->opt_exact_relabel && N >= 256 &&
                           C >= 2 && !c->opt_test_bvec_plain;
    const bool fuse_req = tiles_req && !c->fit_persistent_broken && c->opt_fit_persistent != 1 &&
                          c->opt_pow_exp2 && k_fit_flow_fused_applies(c, N);
    const bool blocks = blocks_req || tiles_req;
    bool blocks_ok = blocks, pieces_ok = true;
    // the sweep's small tables travel together: one pinned staging buffer, one copy and one scatter kernel per flush instead of a
    // copy from pageable memory (~20 us of idle stream) per table and a synchronisation wherever a table is a local
    WordPacker pk(c);
    std::vector<i32> bt_fc, bt_ns, bt_base;
    i64 bt_total = 0; // tile partials in all
    if (blocks) { // per 64-vertex block of the relabelled graph: first community and number of communities; per tile: its partials
        const i64 Nt = (N + 63) / 64;
        std::vector<i32> comm_new(N);
        for (i64 q2 = 0; q2 < C; q2++)
            for (i32 t2 = cm_off[q2]; t2 < cm_off[q2 + 1]; t2++) comm_new[t2] = (i32)q2; // position t2 of the community-sorted order
        bt_fc.resize(Nt); bt_ns.resize(Nt); bt_base.assign(Nt * Nt + 1, 0);
        for (i64 b = 0; b < Nt; b++) {
            bt_fc[b] = comm_new[64 * b];
            bt_ns[b] = comm_new[std::min<i64>(N, 64 * b + 64) - 1] - bt_fc[b] + 1;
            if (bt_ns[b] > 64) blocks_ok = false; // (empty communities in between: the row-bin form takes such a graph)
            // the fused epilogue stages one "piece" per run of a community inside an 8-column chunk (the tail beyond N is a run)
            int pieces = 0;
            for (i64 q = 0; q < 64; q++) {
                const i64 v = 64 * b + q, u = v - 1;
                const i32 cv = v < N ? comm_new[v] : -1, cu = (q > 0) ? (u < N ? comm_new[u] : -1) : -2;
                if ((q & 7) == 0 || cv != cu) pieces++;
            }
            if (pieces > CGE_FLOW_NP) pieces_ok = false;
        }
        i64 tot = 0;
        for (i64 I = 0; I < Nt && blocks_ok; I++)
            for (i64 J = 0; J < Nt; J++) {
                bt_base[I * Nt + J] = (i32)tot;
                if (directed || J >= I) tot += (i64)bt_ns[I] * bt_ns[J];
                if (tot > (i64)1 << 30) blocks_ok = false;
            }
        bt_total = tot;
        if (blocks_ok) {
            c->sw_bt_fc.ensure(Nt); c->sw_bt_ns.ensure(Nt); c->sw_bt_base.ensure(Nt * Nt + 1); c->sw_bt_part.ensure(std::max<i64>(tot, 1) + 1); // (+ the +0.0 slot of k_bins_prepare)
            pk.add(c->sw_bt_fc.p, bt_fc.data(), Nt);
            pk.add(c->sw_bt_ns.p, bt_ns.data(), Nt);
            pk.add(c->sw_bt_base.p, bt_base.data(), Nt * Nt);
        }
    }
    bool fuse = fuse_req && blocks_ok && pieces_ok;
    if (!blocks_req && !tiles_req) blocks_ok = false;
    const bool relabel = (N > 8192 && c->opt_exact_relabel) || blocks_ok;
    c->bvec_blocks = blocks_ok;
    c->bvec_contig = relabel && N >= 64 * C; // a wave per (row, community) pays off for communities of a wave's width or more
    if (relabel) {
        DevBuf<i32> &d_order = c->sw_rl_order;
        d_order.ensure(N);
        d_old2new.ensure(N);
        std::vector<i32> old2new(N);
        for (i64 q = 0; q < N; q++) old2new[cm_mem[q]] = (i32)q;
        pk.add(d_order.p, cm_mem.data(), N);
        pk.add(d_old2new.p, old2new.data(), N);
        pk.flush(); // (with the tile tables above)
        c->sw_rl_emb.ensure((size_t)N * d);
        c->sw_rl_vec.ensure((size_t)4 * N);
        c->sw_rl_comm.ensure(N);
        k_permute_rows(c, G.emb, d_order.p, N, d, c->sw_rl_emb.p);
        G.emb = c->sw_rl_emb.p;
        double *v = c->sw_rl_vec.p;
        if (G.dist) { k_permute_rows(c, G.dist, d_order.p, N, 1, v); G.dist = v; }
        if (G.vw) { k_permute_rows(c, G.vw, d_order.p, N, 1, v + N); G.vw = v + N; }
        if (G.deg_in) { k_permute_rows(c, G.deg_in, d_order.p, N, 1, v + 2 * N); G.deg_in = v + 2 * N; }
        if (G.deg_out) { k_permute_rows(c, G.deg_out, d_order.p, N, 1, v + 3 * N); G.deg_out = v + 3 * N; }
        k_permute_i32(c, G.comm, d_order.p, N, c->sw_rl_comm.p);
        G.comm = c->sw_rl_comm.p;
        for (i64 q = 0; q < N; q++) cm_mem[q] = (i32)q; // the member lists in the new numbering (the old ones sit in the staging buffer)
    }

    // D and its normalisation (:79-93 / :359-375)
    k_dist_matrix(c, G.emb, G.dist, N, d, D.p);
    k_minmax_upper(c, D.p, N, lohi.p);
    k_normalise(c, D.p, N, lohi.p);
    DevBuf<i32> &d_cm_off = c->sw_cm_off, &d_cm_mem = c->sw_cm_mem;
    d_cm_off.ensure(C + 1);
    d_cm_mem.ensure(N);
    pk.add(d_cm_off.p, cm_off.data(), C + 1);
    pk.add(d_cm_mem.p, cm_mem.data(), N);
    std::vector<i32> cm_pos(N); // position of every vertex in the community-sorted list
    for (i64 q = 0; q < N; q++) cm_pos[cm_mem[q]] = (i32)q;
    c->sw_cm_pos.ensure(N);
    pk.add(c->sw_cm_pos.p, cm_pos.data(), N);
    if (!directed) { // (the directed sweep reads the degrees back first, below)
        std::vector<double> ones(N, 1.0); // T (:118)
        pk.add(T1.p, ones.data(), N);
        pk.add(T2.p, ones.data(), N);
        pk.flush();
    } else
        pk.flush();
    if (c->bvec_blocks && !directed) k_bins_prepare(c, d_cm_off.p, N, C, bt_total); // where every bin's tile partials sit

    // T (:118) / Tin,Tout (:399-402)
    std::vector<double> hT1(N, 1.0), hT2(N, 1.0);
    if (directed) {
        std::vector<double> din(N), dout(N);
        HIP_CHECK(hipMemcpyAsync(din.data(), G.deg_in, sizeof(double) * N, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(dout.data(), G.deg_out, sizeof(double) * N, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (i64 i = 0; i < N; i++) {
            if (din[i] == 0) hT1[i] = 0.0;  // Tin
            if (dout[i] == 0) hT2[i] = 0.0; // Tout
        }
    }
    if (directed) {
        pk.add(T1.p, hT1.data(), N);
        pk.add(T2.p, hT2.data(), N);
        pk.flush();
    }
    const double *Tin = T1.p, *Tout = T2.p; // directed
    // undirected: T rotates through the three parts of TT (Tld doubles each, zero beyond N); Tcur = the current iterate.
    // Three, so that T_0 of an alpha survives the fit of the next alpha, which may be enqueued before this one is checked.
    DevBuf<double> &TT = c->fp_T;
    const i64 Tld = (N + 63) / 64 * 64;
    int tpar = 0;
    TT.ensure((size_t)3 * Tld);
    c->fp_Tsave.ensure(N);
    HIP_CHECK(hipMemsetAsync(TT.p, 0, sizeof(double) * 3 * Tld, st));
    HIP_CHECK(hipMemcpyAsync(TT.p, T1.p, sizeof(double) * N, hipMemcpyDeviceToDevice, st));
    double *Tcur = TT.p;
    // sample tallies split over the ranks: with the in-library communicator (stream-ordered, no host synchronisation in the
    // enqueued chain) from 10^5 samples on; option "shard_samples" = 2 forces it (tests, also through the hook), 0 disables
    const bool shard_samples = c->has_coll && smp.S >= c->coll.world &&
                               (c->opt_shard_samples == 2 || (c->opt_shard_samples == 1 && c->rccl_comm && smp.S >= 100000));
    fit_sweep_begin(c);
    bool use_persistent = !directed && !c->fit_persistent_broken && c->opt_fit_persistent != 1 &&
                          (c->opt_fit_persistent >= 2 || N >= 128);
    bool use_persistent_dir = directed && !c->fit_persistent_broken && c->opt_fit_persistent != 1 &&
                              (c->opt_fit_persistent >= 2 || N >= 128);
    // one launch (pair) per iteration, when the register-resident form does not apply: over the upper tiles only
    // (kernels_fitp.hip: k_fit_sym_step)
    c->stat_fit_persistent = 0;
    c->stat_fit_iters = 0;

    // ---- samples -> device ---------------------------------------------------------------------
    const bool landmarks = orig != nullptr;
    const i64 S = smp.S;
    const i32 *e_src = landmarks ? orig->src : ex_src, *e_dst = landmarks ? orig->dst : ex_dst;
    const double *e_hw = landmarks ? orig->h_w : ex_hw;
    const i64 e_m = landmarks ? orig->m : ex_m;
    std::vector<std::unique_ptr<DevSamples>> &dsets = c->dsets; // grow-only buffers kept by the context
    while ((i64)dsets.size() < smp.n_sets) dsets.emplace_back(new DevSamples());
    if (smp.on_device) { // library-drawn samples of the resident graph: everything stays on the device
        for (i64 t = 0; t < smp.n_sets; t++) {
            DevSamples &ds = *dsets[t];
            ds.pi.ensure(S); ds.pj.ensure(S); ds.ni.ensure(S); ds.nj.ensure(S); ds.wts.ensure(S);
            const i32 *pos = smp.d_pos.p + t * S;
            const i32 *pos_pairs = (directed && !landmarks && smp.d_pos2.p) ? smp.d_pos2.p + t * S : pos; // the overwriting draw (:510)
            k_prep_samples(c, pos, pos_pairs, smp.d_ni.p + t * S, smp.d_nj.p + t * S, e_src, e_dst, c->w.p, S, directed, ds.pi.p,
                           ds.pj.p, ds.ni.p, ds.nj.p, ds.wts.p);
            if (landmarks) { // full_graph_D of the sampled pairs, normalised by hi (lo == 0) :104-114
                ds.dpos.ensure(S); ds.dneg.ensure(S);
                sampled_pair_dist(c, orig->Xr, d, ds.pi.p, ds.pj.p, S, orig->hi, ds.dpos.p);
                sampled_pair_dist(c, orig->Xr, d, ds.ni.p, ds.nj.p, S, orig->hi, ds.dneg.p);
            }
        }
    } else {
        DevBuf<i32> d_idx, d_tmp;
        std::vector<i64> rows0(S);
        std::vector<i32> hs, hd, hs2, hd2;
        for (i64 t = 0; t < smp.n_sets; t++) {
            DevSamples &ds = *dsets[t];
            for (i64 k = 0; k < S; k++) {
                rows0[k] = smp.pos_idx[t * S + k] - 1;
                if (rows0[k] < 0 || rows0[k] >= e_m) CGE_THROW(CGE_E_ARG, "positive sample row out of range");
            }
            gather_rows(c, e_src, rows0, hs, d_idx, d_tmp);
            gather_rows(c, e_dst, rows0, hd, d_idx, d_tmp);
            std::vector<double> wts(S);
            for (i64 k = 0; k < S; k++) wts[k] = e_hw ? e_hw[rows0[k]] : 1.0;
            if (directed && !landmarks && !smp.pos_idx2.empty()) { // the overwriting second draw (:510)
                for (i64 k = 0; k < S; k++) rows0[k] = smp.pos_idx2[t * S + k] - 1;
                gather_rows(c, e_src, rows0, hs, d_idx, d_tmp);
                gather_rows(c, e_dst, rows0, hd, d_idx, d_tmp);
            }
            std::vector<i32> pi(S), pj(S), ni(S), nj(S);
            for (i64 k = 0; k < S; k++) {
                i32 a = hs[k], b = hd[k];
                if (!directed && a > b) std::swap(a, b); // E tuple (min,max) :133
                pi[k] = a; pj[k] = b;
                i64 u = smp.neg_i[t * S + k] - 1, v = smp.neg_j[t * S + k] - 1;
                if (!directed && u > v) std::swap(u, v);
                const i64 lim = landmarks ? orig->n : N;
                if (u < 0 || v < 0 || u >= lim || v >= lim) CGE_THROW(CGE_E_ARG, "negative sample out of range");
                ni[k] = (i32)u; nj[k] = (i32)v;
            }
            ds.pi.ensure(S); ds.pj.ensure(S); ds.ni.ensure(S); ds.nj.ensure(S); ds.wts.ensure(S);
            HIP_CHECK(hipMemcpyAsync(ds.pi.p, pi.data(), sizeof(i32) * S, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(ds.pj.p, pj.data(), sizeof(i32) * S, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(ds.ni.p, ni.data(), sizeof(i32) * S, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(ds.nj.p, nj.data(), sizeof(i32) * S, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(ds.wts.p, wts.data(), sizeof(double) * S, hipMemcpyHostToDevice, st));
            if (landmarks) { // full_graph_D of the sampled pairs, normalised by hi (lo == 0) :104-114
                ds.dpos.ensure(S); ds.dneg.ensure(S);
                sampled_pair_dist(c, orig->Xr, d, ds.pi.p, ds.pj.p, S, orig->hi, ds.dpos.p);
                sampled_pair_dist(c, orig->Xr, d, ds.ni.p, ds.nj.p, S, orig->hi, ds.dneg.p);
            }
            HIP_CHECK(hipStreamSynchronize(st)); // host vectors go out of scope
        }
    }

    if (relabel && !landmarks) // the sampled pairs index the score graph: into the new numbering (GD is a full symmetric matrix here)
        for (i64 t = 0; t < smp.n_sets; t++) {
            DevSamples &ds = *dsets[t];
            k_remap_i32(c, ds.pi.p, d_old2new.p, S);
            k_remap_i32(c, ds.pj.p, d_old2new.p, S);
            k_remap_i32(c, ds.ni.p, d_old2new.p, S);
            k_remap_i32(c, ds.nj.p, d_old2new.p, S);
        }

    // ---- the fused chain's table, one per sample set (device copies: the fit's epilogue loads them after its loop) ----------
    constexpr i64 RES_AUC_EARLY = 0; // == RES_AUC below
    const i64 fz_s0 = shard_samples ? S * c->coll.rank / c->coll.world : 0;
    const i64 fz_s1 = shard_samples ? S * (c->coll.rank + 1) / c->coll.world : S;
    const bool fuse_auc = landmarks && fz_s1 - fz_s0 < 65536 && fz_s1 > fz_s0; // (beyond: auc_landmark_kernel's wide form, as its own launch)
    std::vector<cge_fit_fused> h_epi;
    if (fuse) {
        h_epi.resize(smp.n_sets);
        for (i64 t = 0; t < smp.n_sets; t++) {
            const DevSamples &ds = *dsets[t];
            cge_fit_fused &e = h_epi[t];
            e = cge_fit_fused{};
            e.comm = G.comm; e.fc = c->sw_bt_fc.p; e.ns = c->sw_bt_ns.p; e.base = c->sw_bt_base.p; e.partial = c->sw_bt_part.p;
            e.S = fz_s1 - fz_s0;
            e.dpos = ds.dpos.p + fz_s0; e.dneg = ds.dneg.p + fz_s0; e.wts = ds.wts.p + fz_s0;
            if (fuse_auc && e.S > 0) { // everything of the tally that depends neither on alpha nor on T, once
                DevSamples &dsw = *dsets[t];
                dsw.aidx.ensure((size_t)4 * e.S); dsw.afac.ensure((size_t)8 * e.S + CGE_PARTIAL_BLOCKS);
                c->sw_fused_pw.ensure((size_t)2 * e.S);
                k_auc_prepare(c, orig->v2l, d_old2new.p, orig->vw, orig->lweight, ds.pi.p + fz_s0, ds.pj.p + fz_s0, ds.ni.p + fz_s0,
                              ds.nj.p + fz_s0, e.wts, e.S, dsw.aidx.p, dsw.afac.p, dsw.afac.p + 8 * e.S);
                e.aidx = dsw.aidx.p; e.afac = dsw.afac.p; e.aden = dsw.afac.p + 8 * e.S; e.apw = c->sw_fused_pw.p;
            }
            e.auc_part = scal.p + RES_AUC_EARLY;
        }
        c->sw_fused_epi.ensure(h_epi.size() * sizeof(cge_fit_fused));
        static_assert(sizeof(cge_fit_fused) % 8 == 0, "packed as 4-byte words");
        pk.add(reinterpret_cast<i32 *>(c->sw_fused_epi.p), reinterpret_cast<const i32 *>(h_epi.data()),
               (i64)(h_epi.size() * sizeof(cge_fit_fused) / 4));
        pk.flush();
    }
    c->stat_fit_fused = 0;

    // ---- alpha sweep ---------------------------------------------------------------------------
    int alpha_div_counter = 5, alpha_auc_counter = 5; // :38
    bool skip_div = false, skip_auc = false;
    double best_div = INFINITY, best_div_ext = INFINITY, best_div_int = INFINITY, best_auc_err = INFINITY,
           best_auc = INFINITY; // typemax(Float64)
    double best_alpha = -1.0, best_alpha_auc = -1.0;
    if (trace) trace->n_alpha = 0;
    const i64 n_alpha_total = (i64)std::floor((AlphaMax + delta) / AlphaStep + 1e-9);
    i64 prev_iters = 16;
    // An alpha is a chain on the stream: pow, the fit, AUC, vect_B, JS, scalars -> pinned host memory, an event.  With the
    // enqueue-only persistent fit (undirected) nothing in the chain needs the host, so the chain of alpha i+1 is
    // enqueued before the host waits for alpha i -- unless the sweep may end at alpha i (both patience counters at their
    // last value), so nothing is ever computed in vain.  T alternates between the two halves of TT; the scalars and the
    // fit's verdict of an alpha land in slot (alpha index mod 2).
    struct AlphaSlot {
        bool fit_async = false, did_auc = false, did_div = false;
        bool fused = false; // the rest of the chain rode on the fit's launch
        bool shared_verdict = false; // N > 1, tallies split: the verdict of the fit travelled with the all-reduced tallies
        int t0_par = 0;  // the half of TT that held T_0 of this alpha
        i64 iters = 0;
    } slots[2];
    // RES_VERD sits right behind the tallies: one all-reduce(sum) covers both (the slot after it only keeps RES_JS 16-byte aligned)
    static_assert(RES_AUC_EARLY == 0, "");
    constexpr i64 RES_AUC = 0, RES_VERD = 2 * CGE_PARTIAL_BLOCKS, RES_JS = RES_VERD + 2, RES_FIT = RES_JS + 2 * CGE_PARTIAL_BLOCKS,
                  RES_LEN = RES_FIT + 2, RES_STRIDE = RES_FIT + 16;
    c->pin_scal.ensure(2 * RES_STRIDE);
    // (Round 4 tried the next alpha's power matrix on a side stream beside this alpha's vect_B / JS / AUC: +0.65 ms, the two
    // cross-queue dependencies per alpha cost more than they hid -- profiles/r04_pow_overlap_ab.txt; removed in round 5, when
    // the power matrix moved into the fit's prologue anyway.)
    auto enqueue_alpha = [&](i64 ia, bool want_auc, bool want_div) {
        AlphaSlot &sl = slots[ia & 1];
        const int slot = (int)(ia & 1);
        const double alpha = AlphaStep * (double)ia;
        sl = AlphaSlot();
        sl.did_auc = want_auc;
        sl.did_div = want_div;
        // the undirected persistent fit and vect_B read the upper triangle only; the exact-mode AUC, the directed vect_B
        // and the launch-per-iteration fits read whole rows
        const bool gd_upper = landmarks && !directed;
        double *const GDc = GD.p; // this alpha's matrix
        const bool fused_now = fuse && use_persistent && c->pow_logs_blocked_N == N; // (a fallback in mid-sweep ends it: the matrix is needed then)
        if (fuse && !fused_now && c->pow_logs_N != N) k_pow_prepare(c, D.p, N, landmarks && !directed); // (left the fused path: the row-major logarithm)
        bool auc_done = false, bvec_partials = false, copied_out = false;
        if (!fused_now) k_pow_matrix(c, D.p, N, alpha, GDc, gd_upper);
        if (directed || !use_persistent) HIP_CHECK(hipMemsetAsync(flags.p, 0, sizeof(int) * 4, st));
        if (directed) {
            const double init[2] = {0.9, 1.0}; // epsilon, diff (:434-435)
            HIP_CHECK(hipMemcpyAsync(fitstate.p, init, sizeof(init), hipMemcpyHostToDevice, st));
        }
        i64 iters = 0;
        bool dir_async = false;
        i64 batch = std::max<i64>(4, std::min<i64>(prev_iters, 128));
        if (!directed) {
            bool fitted = false;
            sl.t0_par = tpar;
            if (use_persistent) { // the whole fit in one launch, GD's upper triangle in registers (kernels_fitp.hip)
                const int tnext = (tpar + 1) % 3;
                cge_fit_fused ff{};
                const cge_fit_fused *ffp = nullptr, *ffd = nullptr;
                if (fused_now) {
                    const i64 set = smp.n_sets == 1 ? 0 : ia - 1;
                    ff = h_epi[set];
                    ff.Lh = c->sw_Lh.p; ff.Ll = c->sw_Ll.p; ff.alpha = alpha;
                    if (!want_div) ff.partial = nullptr;
                    if (!(want_auc && fuse_auc)) ff.auc_part = nullptr;
                    ffp = &ff;
                    ffd = reinterpret_cast<const cge_fit_fused *>(c->sw_fused_epi.p) + set;
                }
                if (k_fit_flow_enqueue(c, fused_now ? nullptr : GDc, N, TT.p + (i64)tpar * Tld, TT.p + (i64)tnext * Tld,
                                                           Tld, G.vw, 0.25, delta, (int *)(scal.p + RES_FIT), ffp, ffd)) {
                    if (fused_now) {
                        auc_done = ff.auc_part != nullptr;
                        bvec_partials = ff.partial != nullptr;
                        sl.fused = true;
                    }
                    sl.fit_async = true; // the verdict is looked at when the alpha is collected
                    fitted = true;
                    tpar = tnext;
                } else { // the register-resident form does not apply to this size: one launch per iteration from here on
                    use_persistent = false;
                    if (fused_now) { // (the fused launch was to supply the matrix)
                        k_pow_prepare(c, D.p, N, landmarks && !directed);
                        k_pow_matrix(c, D.p, N, alpha, GDc, gd_upper);
                    }
                    HIP_CHECK(hipMemsetAsync(flags.p, 0, sizeof(int) * 4, st));
                }
            }
            if (!fitted) {
                // one launch per iteration (kernels_fit.hip: fit_step_kernel); T alternates between the two buffers
                HIP_CHECK(hipMemsetAsync(c->sw_fring.p, 0, sizeof(unsigned long long) * 4, st));
                double *Tb2[2] = {TT.p + (i64)tpar * Tld, TT.p + (i64)((tpar + 1) % 3) * Tld};
                i64 k = 0;
                for (;;) {
                    for (i64 b = 0; b < batch; b++, k++)
                        k_fit_sym_step(c, GDc, Tb2[k & 1], Tb2[(k + 1) & 1], G.vw, N, 0.25, delta, (int)k, c->sw_fring.p, flags.p,
                                       flags.p + 1);
                    int hf[2];
                    unsigned long long hr[3];
                    HIP_CHECK(hipMemcpyAsync(hf, flags.p, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
                    HIP_CHECK(hipMemcpyAsync(hr, c->sw_fring.p, sizeof(hr), hipMemcpyDeviceToHost, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                    iters = hf[1];
                    if (hf[0]) break;
                    double flast;
                    std::memcpy(&flast, &hr[(k - 1) % 3], sizeof(double));
                    if (!(flast > delta)) break; // the last launch of the batch was the converging iteration (iters == k)
                    if (iters > c->opt_fit_max_iters) CGE_THROW(CGE_E_ASSERT, "Chung-Lu fit did not converge at alpha=%g (%lld iterations; the reference's `while diff > delta` would not return)", alpha, (long long)iters);
                    batch = std::max<i64>(4, std::min<i64>(batch, 32));
                }
                if (iters & 1) tpar = (tpar + 1) % 3;
            }
            Tcur = TT.p + (i64)tpar * Tld;
        } else if (use_persistent_dir &&
                   k_fit_persistent_dir(c, GDc, N, T1.p, T2.p, G.deg_in, G.deg_out, 0.9, 1.0, delta, &iters, (int *)(scal.p + RES_FIT),
                                        &dir_async)) {
            // the whole directed fit in one launch (kernels_fitp.hip).  The default form is only enqueued: the rest of the
            // alpha's chain is queued behind it and its verdict arrives with the alpha's scalars (Tin / Tout are written
            // on success only, so a failed launch is redone from the same iterates with one launch pair per iteration).
            if (dir_async) sl.fit_async = true;
            else c->stat_fit_persistent++;
        } else
        for (;;) {
            if (use_persistent_dir) { // abandoned: T1 / T2 are untouched; one launch pair per iteration from here on
                use_persistent_dir = false;
                HIP_CHECK(hipMemsetAsync(flags.p, 0, sizeof(int) * 4, st));
            }
            for (i64 b = 0; b < batch; b++) {
                k_fit_symv_dir(c, GDc, T1.p, T2.p, N, S1.p, S2.p, flags.p);
                k_fit_update_dir(c, T1.p, T2.p, S1.p, S2.p, G.deg_in, G.deg_out, N, delta, flags.p, flags.p + 1,
                                 fitstate.p);
            }
            int hf[2];
            HIP_CHECK(hipMemcpyAsync(hf, flags.p, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            iters = hf[1];
            if (hf[0]) break;
            if (iters > c->opt_fit_max_iters) CGE_THROW(CGE_E_ASSERT, "Chung-Lu fit did not converge at alpha=%g (%lld iterations; the reference's `while diff > delta` would not return)", alpha, (long long)iters);
            batch = std::max<i64>(4, std::min<i64>(batch, 32));
        }
        sl.iters = iters;
        if (!sl.fit_async) prev_iters = iters;
        const double *Ta = directed ? Tout : Tcur, *Tb = directed ? Tin : Tcur;
        const double *Ta_auc = Ta, *Tb_auc = Tb;
        if (want_auc && !auc_done && relabel && landmarks) { // v_to_l holds the landmark ids of the original numbering
            c->sw_rl_T.ensure((size_t)2 * N);
            k_permute_rows(c, Ta, d_old2new.p, N, 1, c->sw_rl_T.p);
            Ta_auc = Tb_auc = c->sw_rl_T.p;
            if (directed) {
                k_permute_rows(c, Tb, d_old2new.p, N, 1, c->sw_rl_T.p + N);
                Tb_auc = c->sw_rl_T.p + N;
            }
        }
        if (want_auc && !auc_done) {
            const DevSamples &ds = *dsets[smp.n_sets == 1 ? 0 : ia - 1];
            // N > 1 with many samples: rank r tallies the samples [S r / W, S (r + 1) / W) and the block tallies
            // are summed over the ranks -- the same array on every rank afterwards, so all ranks take the same early stops
            const i64 s0 = shard_samples ? S * c->coll.rank / c->coll.world : 0;
            const i64 s1 = shard_samples ? S * (c->coll.rank + 1) / c->coll.world : S;
            if (landmarks)
                k_auc_landmark(c, Ta_auc, Tb_auc, orig->v2l, orig->vw, orig->lweight, ds.pi.p + s0, ds.pj.p + s0, ds.ni.p + s0,
                               ds.nj.p + s0, ds.dpos.p + s0, ds.dneg.p + s0, ds.wts.p + s0, s1 - s0, alpha, nullptr,
                               scal.p + RES_AUC);
            else
                k_auc_exact(c, GDc, Ta, Tb, N, ds.pi.p + s0, ds.pj.p + s0, ds.ni.p + s0, ds.nj.p + s0, ds.wts.p + s0, s1 - s0,
                            nullptr, scal.p + RES_AUC);
        }
        if (shard_samples) {
            // The verdict of an enqueued fit is rank-local (a hand-off may time out on one rank only), but a redo re-issues
            // this exchange and changes what the rank enqueues from then on: the ranks must take it together.  So the verdict
            // rides along as one more summand -- at EVERY alpha of a sweep with split tallies, with or without a local score --
            // and every rank redoes the alpha when any rank's fit was abandoned: the ranks never leave lock-step.
            k_fit_verdict(c, (const int *)(scal.p + RES_FIT), sl.fit_async ? 1 : 0, scal.p + RES_VERD);
            sl.shared_verdict = true;
            if (want_auc) cge_allreduce_dev(c, scal.p + RES_AUC, 2 * CGE_PARTIAL_BLOCKS + 1, 0);
            else cge_allreduce_dev(c, scal.p + RES_VERD, 1, 0);
        }
        if (want_div) {
            if ((bvec_partials || c->bvec_blocks) && !directed && !c->opt_test_bvec_plain) {
                // tile partials (from the fit's epilogue, or one pass over GD) -> vect_B and its divergence(s) in one launch
                if (!bvec_partials) k_bvec_tiles(c, GDc, Ta, Tb, d_cm_off.p, N, directed);
                // the last launch of the alpha: it also hands the alpha's scalars to the host's pinned slot and arms the hand-off
                // slots of the next alpha's persistent fit (instead of a copy and a fill of their own)
                cge_chain_tail tail{};
                tail.host_out = c->pin_scal.p + RES_STRIDE * slot;
                tail.scal = scal.p;
                tail.res_js = (int)RES_JS;
                tail.res_len = (int)RES_LEN;
                bool arms = false;
                if (use_persistent && ia < n_alpha_total) arms = k_fit_flow_arm_region(c, N, Tld, &tail.arm, &tail.arm_n16, &tail.arm_word);
                if (!arms) { tail.arm = nullptr; tail.arm_n16 = 0; }
                k_bins_js(c, d_cm_off.p, N, C, G.vectC, vectB.p, split ? 2 : 1, scal.p + RES_JS, &tail);
                if (arms) c->flow_armed_words = 4 * tail.arm_n16;
                copied_out = true;
            } else {
                if (bvec_partials) k_bvec_bins(c, d_cm_off.p, N, C, directed, vectB.p); // the tile partials came with the fit
                else k_bvec(c, GDc, Ta, Tb, c->sw_cm_pos.p, d_cm_off.p, d_cm_mem.p, N, C, directed, rowbins.p, vectB.p);
                if (!split)
                    k_js(c, G.vectC, vectB.p, vlen, C, directed, 0, nullptr, scal.p + RES_JS);
                else {
                    k_js(c, G.vectC, vectB.p, vlen, C, directed, 1, nullptr, scal.p + RES_JS);
                    k_js(c, G.vectC, vectB.p, vlen, C, directed, 2, nullptr, scal.p + RES_JS + CGE_PARTIAL_BLOCKS);
                }
            }
        }
        // the block partials of the alpha's reductions and (behind them) the verdict of an enqueued fit, one copy; the host
        // adds the partials in block order -- what the one-thread "final" kernels did, without their launches
        if (!copied_out)
            HIP_CHECK(hipMemcpyAsync(c->pin_scal.p + RES_STRIDE * slot, scal.p, sizeof(double) * RES_LEN, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventRecord(c->sweep_ev[slot], st));
    };

    // log2(1 - D) once for the whole sweep (the upper tiles only when every alpha reads only those); a fallback of the
    // persistent fit in mid-sweep makes k_pow_matrix use the library pow for the whole rows it then needs
    k_pow_prepare(c, D.p, N, landmarks && !directed, fuse);
    i64 next_enqueue = 1;
    for (i64 ia = 1; ia <= n_alpha_total; ia++) {
        const double alpha = AlphaStep * (double)ia;
        AlphaSlot &sl = slots[ia & 1];
        if (next_enqueue == ia) {
            enqueue_alpha(ia, !skip_auc, !skip_div);
            next_enqueue = ia + 1;
        }
        const bool may_end_here = (skip_div || alpha_div_counter == 1) && (skip_auc || alpha_auc_counter == 1);
        // (directed: Tin / Tout are updated in place, so the next alpha is not queued before this one's verdict is known)
        if (sl.fit_async && !directed && !may_end_here && ia < n_alpha_total) { // keep the device busy while the host reads alpha ia
            enqueue_alpha(ia + 1, !skip_auc, !skip_div);
            next_enqueue = ia + 2;
        }
        HIP_CHECK(hipEventSynchronize(c->sweep_ev[ia & 1]));
        const bool peer_failed = sl.shared_verdict && c->pin_scal.p[RES_STRIDE * (ia & 1) + RES_VERD] != 0.0;
        if (peer_failed && !sl.fit_async) // cannot happen while the ranks are in lock-step (they enqueue the same form of fit)
            CGE_THROW(CGE_E_COLLECTIVE, "another rank abandoned a persistent fit this rank did not enqueue: the ranks diverged");
        if (sl.fit_async) {
            const int *hf = (const int *)(c->pin_scal.p + RES_STRIDE * (ia & 1) + RES_FIT);
            if (hf[2] || !hf[0] || peer_failed) { // a wait timed out (here or on another rank): drain what was enqueued behind
                HIP_CHECK(hipStreamSynchronize(st)); // it and redo this alpha from its T_0 (still in place) with one launch per
                note_fit_fallback(c);                // iteration, as every later alpha
                if (directed) use_persistent_dir = false;
                else {
                    use_persistent = false;
                    tpar = sl.t0_par;
                }
                next_enqueue = ia;
                ia--;
                continue;
            }
            sl.iters = hf[1];
            prev_iters = sl.iters;
            c->stat_fit_persistent++;
            if (sl.fused) c->stat_fit_fused++;
        }
        const i64 iters = sl.iters;
        c->stat_fit_iters += iters;
        double auc_val = NAN, div_val = NAN, div_int = 0.0, div_ext = 0.0;
        const double *res = c->pin_scal.p + RES_STRIDE * (ia & 1);
        double hs[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        if (!skip_auc)
            for (int b = 0; b < CGE_PARTIAL_BLOCKS; b++) { hs[0] += res[RES_AUC + 2 * b]; hs[1] += res[RES_AUC + 2 * b + 1]; }
        if (!skip_div) {
            double fa = 0.0, fb = 0.0;
            for (int b = 0; b < CGE_PARTIAL_BLOCKS; b++) { fa += res[RES_JS + b]; fb += res[RES_JS + CGE_PARTIAL_BLOCKS + b]; }
            if (!split) hs[2] = fa / 2.0;
            else { hs[3] = fa / 2.0; hs[4] = fb / 2.0; }
        }
        if (!skip_auc) {
            const double auc = 1.0 - hs[0] / hs[1]; // :213
            auc_val = auc;
            if (auc < best_auc) {
                best_auc = auc;
                best_auc_err = 1.96 * std::sqrt(auc * (1.0 - auc) / (double)S); // :217
                best_alpha_auc = alpha;
                alpha_auc_counter = 5;
            } else {
                alpha_auc_counter -= 1;
                skip_auc = alpha_auc_counter == 0;
            }
        }
        if (!skip_div) {
            double f;
            if (!split)
                f = hs[2];
            else {
                div_int = hs[3];
                div_ext = hs[4];
                f = (div_int + div_ext) / 2.0;
            }
            div_val = f;
            if (f < best_div) {
                best_div = f;
                best_alpha = alpha;
                best_div_ext = !split ? 0.0 : div_ext;
                best_div_int = !split ? 0.0 : div_int;
                alpha_div_counter = 5;
            } else {
                alpha_div_counter -= 1;
                skip_div = alpha_div_counter == 0;
            }
        }
        if (trace && trace->n_alpha < 64) {
            trace->iters[trace->n_alpha] = iters;
            trace->div[trace->n_alpha] = div_val;
            trace->auc[trace->n_alpha] = auc_val;
            trace->n_alpha++;
        }
        if (skip_div && skip_auc) break; // :253
    }
    out[0] = best_alpha; out[1] = best_div; out[2] = best_div_ext; out[3] = best_div_int;
    out[4] = best_alpha_auc; out[5] = best_auc; out[6] = best_auc_err; // :256
    *out_len = 7;
    if (c->stat_fit_persistent > 0 && !c->fit_persistent_broken) c->fit_fallback_streak = 0; // a clean persistent sweep
}

// ---- small helpers ---------------------------------------------------------------------------------
void k_gather_i32(cge_ctx *c, const i32 *arr, const i32 *idx, i64 S, i32 *out); // kernels_fit.hip

static void gather_rows(cge_ctx *c, const i32 *d_arr, const std::vector<i64> &rows0, std::vector<i32> &out,
                        DevBuf<i32> &d_idx, DevBuf<i32> &d_out) {
    const i64 S = (i64)rows0.size();
    std::vector<i32> idx(S);
    for (i64 k = 0; k < S; k++) idx[k] = (i32)rows0[k];
    d_idx.ensure(S);
    d_out.ensure(S);
    out.resize(S);
    HIP_CHECK(hipMemcpyAsync(d_idx.p, idx.data(), sizeof(i32) * S, hipMemcpyHostToDevice, c->stream));
    k_gather_i32(c, d_arr, d_idx.p, S, d_out.p);
    HIP_CHECK(hipMemcpyAsync(out.data(), d_out.p, sizeof(i32) * S, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
