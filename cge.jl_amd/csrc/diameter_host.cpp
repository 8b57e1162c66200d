// diameter_host.cpp -- exact point-set diameter `hi = maximum(full_graph_D)` (src/divergence.jl:104-113)
// by branch-and-bound over landmark pairs, with the brute-force MFMA kernel as fallback.
//
// For x_i in landmark a and x_j in landmark b, with ANY reference points mu_a, mu_b:
//   ||x_i - x_j||^2 <= P_ab + P_ba - ||mu_a - mu_b||^2 + 2 sqrt(P_aa P_bb),   P_ab = max_{i in a} ||x_i - mu_b||^2
// (expand both sides; the only inequality is Cauchy-Schwarz on <x_i - mu_a, x_j - mu_b>).
// 1. P (N x N) by one fp64-MFMA pass of all n vertices against the N centroids (n*N*d FMA, ~1 % of the
//    brute-force work for N = 4000 landmarks on 10^6 vertices);
// 2. a lower bound L of the diameter^2 from three farthest-point sweeps;
// 3. every landmark pair whose bound reaches L is evaluated exactly, in decreasing bound order, L rising
//    as better pairs are found; pairs whose bound falls below L are dropped.
// The result is the exact arg-max pair whatever the data; only the amount of pruning is data dependent.
// When fewer than half of the brute-force tiles can be pruned the brute-force kernel runs instead.
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace {
struct BoundRec {
    double B;
    i32 a, b;
};
} // namespace

// returns false when the caller should fall back to brute force
// `mu` = N landmark centroids (device, row-major), `lw` their weights (device) and `lcomm` their communities
// (host, 0-based, C communities).  Reference points: the community centroids when there are at least 32
// communities (the MFMA pass is then n*C*d instead of n*N*d), else the landmark centroids themselves.
//
// THE EXACT STAGE WORKS ON GATHERED ROWS (round 4).  The bound pass reads the bf16 planes (or the f32 copy) of the
// landmark-sorted rows, so no fp64 copy of the whole embedding is written any more (it was 1 GB written and 1 GB cleared per
// score at the headline); the few landmarks that survive the bounds are gathered per round -- centred, feature-major, fp64 --
// into `xe`, and the fp64-MFMA tile kernel runs on that.
//
// OPTION shard_rows (c->rows_sharded): `mem_off` / the device index hold THIS RANK's members (local row ids, other ranks'
// landmarks are empty ranges), c->h_gl_off the global sizes.  Every rank bounds its own rows (all-reduce(max) of the bound
// matrix), the seed row of the farthest-point sweep is fetched from its owner, and the gathered rows of a round are
// completed by an all-reduce of the zero-filled gather (op 2: exact), so a candidate pair whose landmarks live on two ranks
// is evaluated like any other; the rounds are taken in lock-step (the best value so far is all-reduced after each).
bool host_diameter_pruned(cge_ctx *c, const double *mu, const double *lw, const std::vector<i32> &lcomm, i64 C, i64 N,
                          const std::vector<i32> &mem_off, const std::vector<i32> &mem, int part, int nparts,
                          double *best_d2, i64 *bi, i64 *bj) {
    const i64 n = c->n, d = c->d, dpad = c->dpad;
    const bool RS = c->rows_sharded;
    const i64 n_rows = lm_rows(c);
    const int W = RS ? c->coll.world : 1, me = RS ? c->coll.rank : 0;
    hipStream_t st = c->stream;
    c->stat_cand_pairs = c->stat_cand_tiles = 0;
    double tphase = now_ms();
    auto lap = [&](const char *name) { // host wall time per stage (the stream is not synchronised here)
        const double t = now_ms();
        c->phases.ms[name] += t - tphase;
        tphase = t;
    };
    if (RS && ((i64)c->h_gl_off.size() != N + 1 || !c->lm_index_on_device || mem_off.data() != c->h_mem_off.data()))
        CGE_THROW(CGE_E_ARG, "diameter (shard_rows): the landmark index of this context's own landmark phase is required");
    const std::vector<i32> &gl_off = RS ? c->h_gl_off : mem_off; // sizes of the WHOLE landmarks
    // ---- landmark-sorted layout: landmark a owns positions [soff[a], soff[a] + cnt16[a]) (this rank's members) ------------
    std::vector<i64> soff(N + 1, 0);
    for (i64 a = 0; a < N; a++) soff[a + 1] = soff[a] + ((mem_off[a + 1] - mem_off[a] + 15) / 16) * 16;
    const i64 npos = soff[N];
    const i64 lds_rows = (npos + 127) / 128 * 128 + 128;
    c->pos2node.ensure(std::max<i64>(npos, 1));
    c->sub_land.ensure(lds_rows / 16);
    const i32 *d_off = c->lm_memoff.p, *d_mem = c->lm_mem.p;
    {
        std::vector<i32> soff32(soff.begin(), soff.end());
        c->dm_soff.ensure(N + 1);
        WordPacker pk(c); // (small tables: one pinned staging buffer, no copy from pageable memory, no synchronisation for the local)
        pk.add(c->dm_soff.p, soff32.data(), N + 1);
        pk.flush();
        if (!(c->lm_index_on_device && mem_off.data() == c->h_mem_off.data())) {
            // index not produced by the landmark phase of this context (exact-mode callers): upload it
            if ((i64)mem.size() != n) return false; // no member lists on the host: the caller takes the brute-force path
            c->dm_memoff.ensure(N + 1);
            c->dm_mem.ensure(n);
            HIP_CHECK(hipMemcpyAsync(c->dm_memoff.p, mem_off.data(), sizeof(i32) * (N + 1), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(c->dm_mem.p, mem.data(), sizeof(i32) * n, hipMemcpyHostToDevice, st));
            d_off = c->dm_memoff.p;
            d_mem = c->dm_mem.p;
        }
        k_diameter_layout(c, d_off, d_mem, c->dm_soff.p, N, c->pos2node.p, c->sub_land.p, lds_rows / 16);
        if (d_off == c->dm_memoff.p) HIP_CHECK(hipStreamSynchronize(st)); // (the caller's index was copied from pageable memory)
    }
    lap("dm_layout");
    // ---- reference points ---------------------------------------------------------------------------------
    const bool by_comm = C >= 32 && (i64)lcomm.size() == N && lw != nullptr;
    const i64 nref = by_comm ? C : N;
    c->stat_nref = nref;
    std::vector<i32> lref(N);
    const double *mu_ref = mu;
    if (by_comm) {
        std::vector<i32> roff(C + 1, 0), rmem(N);
        for (i64 a = 0; a < N; a++) {
            if (lcomm[a] < 0 || lcomm[a] >= C) CGE_THROW(CGE_E_ARG, "diameter: landmark community out of range");
            lref[a] = lcomm[a];
            roff[lcomm[a] + 1]++;
        }
        for (i64 q = 0; q < C; q++) roff[q + 1] += roff[q];
        std::vector<i32> cur(roff.begin(), roff.end() - 1);
        for (i64 a = 0; a < N; a++) rmem[cur[lcomm[a]]++] = (i32)a;
        c->mp_refoff.ensure(C + 1);
        c->mp_refmem.ensure(N);
        c->mp_refmu.ensure((size_t)C * d);
        c->mp_lref.ensure(N);
        WordPacker pk(c);
        pk.add(c->mp_refoff.p, roff.data(), C + 1);
        pk.add(c->mp_refmem.p, rmem.data(), N);
        pk.add(c->mp_lref.p, lref.data(), N);
        pk.flush();
        k_ref_centroids(c, mu, lw, c->mp_refoff.p, c->mp_refmem.p, C, d, c->mp_refmu.p);
        mu_ref = c->mp_refmu.p;
    } else {
        for (i64 a = 0; a < N; a++) lref[a] = (i32)a;
        c->mp_lref.ensure(N);
        WordPacker pk(c);
        pk.add(c->mp_lref.p, lref.data(), N);
        pk.flush();
    }
    const i64 ldm = (nref + 127) / 128 * 128;
    c->rns.ensure(lds_rows);
    c->Ms.ensure((size_t)ldm * dpad);
    c->mnorm.ensure(ldm);
    c->Pm.ensure((size_t)N * nref);
    // the maxima as upper bounds from a low-precision matrix pass (kernels_dist.hip (2b), (2c)) or exactly in fp64
    bool b16 = c->opt_diameter_f32 >= 2 && k_pcent_bf16_applies(dpad);
    bool f32 = c->opt_diameter_f32 != 0 && !b16;
    const i64 KP = (dpad + 31) / 32 * 32;
    if (f32) {
        c->Xs32.ensure((size_t)lds_rows * dpad);
        c->Ms32.ensure((size_t)ldm * dpad);
    }
    if (b16) {
        c->Xb16.ensure((size_t)2 * lds_rows * KP);
        c->Mb16.ensure((size_t)2 * ldm * KP);
    }
    const bool lowp = b16 || f32;
    if (lowp) {
        c->dm_flag.ensure(1);
        HIP_CHECK(hipMemsetAsync(c->dm_flag.p, 0, sizeof(int), st));
    } else
        c->Xs.ensure((size_t)lds_rows * dpad);
    // the landmark-sorted rows: norms + the operands of the bound pass.  The fp64 feature-major copy is written only when the
    // bound pass itself is fp64 (the exact stage gathers what it needs, below)
    k_gather_centre_fm(c, c->Xr.p, c->pos2node.p, c->gmean.p, lowp ? nullptr : c->Xs.p, c->rns.p, npos, d, lds_rows, dpad,
                       f32 ? c->Xs32.p : nullptr, b16 ? c->Xb16.p : nullptr, KP, lowp ? c->dm_flag.p : nullptr);
    k_gather_centre_fm(c, mu_ref, nullptr, c->gmean.p, c->Ms.p, c->mnorm.p, nref, d, ldm, dpad, f32 ? c->Ms32.p : nullptr,
                       b16 ? c->Mb16.p : nullptr, KP, lowp ? c->dm_flag.p : nullptr);
    int unfit = 0;
    if (lowp) HIP_CHECK(hipMemcpyAsync(&unfit, c->dm_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    // the seed of the farthest-point sweep: the vertex farthest from the centre (largest squared norm of the centred rows, just
    // computed by the gather); its read-back is the synchronisation the flag needs anyway
    double seed_norm = -1.0;
    i64 seed_vertex = k_argmax_mapped(c, c->rns.p, npos, c->pos2node.p, &seed_norm); // (a row id of this rank)
    // The exact fp64 pass (below) instead of a low-precision one when a centred value lies beyond 2^+-100 or is not finite
    // (bit 0), or when NO centred value reaches 2^-40 (bit 1 clear): products below 2^-126 are flushed to zero in the fp32
    // accumulators, an absolute error of < 3K 2^-126 per dot product that the relative margin e only covers while the
    // distances that matter (>= L >= the largest centred norm squared >= 2^-80) dwarf it.
    if (RS && lowp) {
        // option shard_rows: the verdict is taken TOGETHER -- "some rank holds an unfit value" (max of bit 0) and "some
        // rank holds a value of normal size" (max of bit 1) -- so every rank runs the same bound pass whatever rows it happens
        // to hold (a rank without rows of its own simply agrees): the candidate set, the statistics and the timing no longer
        // depend on the sharding
        const double bad = cge_allreduce_scalar_max(c, (unfit & 1) ? 1.0 : 0.0), big = cge_allreduce_scalar_max(c, (unfit & 2) ? 1.0 : 0.0);
        unfit = (bad != 0.0 ? 1 : 0) | (big != 0.0 ? 2 : 0);
    }
    if (lowp && ((unfit & 1) || !(unfit & 2))) {
        b16 = f32 = false;
        c->Xs.ensure((size_t)lds_rows * dpad);
        k_gather_centre_fm(c, c->Xr.p, c->pos2node.p, c->gmean.p, c->Xs.p, c->rns.p, npos, d, lds_rows, dpad, nullptr, nullptr, KP, nullptr);
    }
    // Q is a maximum over vertices: with several ranks each takes its share of the vertex tiles (all of its own rows when the
    // rows are sharded) and the maxima are combined by one all-reduce(max)
    const bool shard_q = !RS && nparts > 1 && c->has_coll && (c->rccl_comm || (c->xptr && (size_t)(N * nref) <= c->xcap));
    c->stat_bound_pass = b16 ? 2 : (f32 ? 1 : 0);
    // One farthest-point sweep from the vertex farthest from the centre (a memory-bound read of Xr) on the side stream, LAUNCHED
    // AHEAD of the bound pass: its small workgroups then fill the CUs first and the sweep runs at its own speed (~0.3 ms) while
    // the bound pass's one-workgroup-per-CU tiles move in beside them -- launched behind it, the sweep was left the gaps
    // (1.0 ms, the longer of the two concurrent branches).  Its result is collected after the bound pass has been enqueued.
    i64 seed_glob = seed_vertex;
    if (RS) {
        // the seed is the farthest of ALL ranks' rows: the largest norm wins (ties: the lowest rank), its owner hands the row
        // itself (and its vertex id) to everybody; the sweep then runs over every rank's own rows on the main stream
        const double vmax = cge_allreduce_scalar_max(c, seed_norm);
        const double claim = cge_allreduce_scalar_max(c, (seed_norm == vmax && seed_vertex >= 0) ? (double)(W - me) : 0.0);
        const int winner = W - (int)claim;
        c->dm_seed.ensure(d + 1);
        HIP_CHECK(hipMemsetAsync(c->dm_seed.p, 0, sizeof(double) * (d + 1), st));
        if (winner == me) {
            HIP_CHECK(hipMemcpyAsync(c->dm_seed.p, c->Xr.p + seed_vertex * d, sizeof(double) * d, hipMemcpyDeviceToDevice, st));
            const double gid = (double)(c->h_loc2glob[seed_vertex] + 1);
            HIP_CHECK(hipMemcpyAsync(c->dm_seed.p + d, &gid, sizeof(double), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st)); // gid is a stack variable
        }
        cge_allreduce_dev(c, c->dm_seed.p, d + 1, 2);
        double gid = 0.0;
        HIP_CHECK(hipMemcpyAsync(&gid, c->dm_seed.p + d, sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        seed_glob = (i64)gid - 1;
        k_farthest_enqueue(c, c->Xr.p, n_rows, d, 0, c->dm_seed.p);
        k_ref_dist2_fm(c, c->Ms.p, nref, dpad, ldm);
    } else {
        std::swap(c->stream, c->copy_stream);
        try {
            k_farthest_enqueue(c, c->Xr.p, n, d, seed_vertex);
            // the reference points' mutual distances (candidate selection) do not depend on the bound pass either: beside it
            // (the gathers that wrote Ms are done: the seed's read-back above synchronised the main stream)
            k_ref_dist2_fm(c, c->Ms.p, nref, dpad, ldm);
        } catch (...) {
            std::swap(c->stream, c->copy_stream);
            throw;
        }
        std::swap(c->stream, c->copy_stream);
    }
    // the sweep's result (a double and an id in c->mp_recs) is read before the bound pass may reuse that scratch
    double L = 0.0;
    i64 far_i = seed_glob, far_j = seed_glob;
    if (RS) {
        double v = -1.0;
        i64 q = 0;
        k_farthest_collect(c, &v, &q);
        const double Lmax = cge_allreduce_scalar_max(c, v);
        const double claim = cge_allreduce_scalar_max(c, v == Lmax ? (double)(W - me) : 0.0);
        const double qid = cge_allreduce_scalar_max(c, (W - (int)claim) == me ? (double)(c->h_loc2glob[q] + 1) : 0.0);
        if (Lmax > L) { L = Lmax; far_j = (i64)qid - 1; }
    }
    if (npos > 0) {
        if (b16)
            k_pcent_bf16(c, c->Xb16.p, c->rns.p, lds_rows, c->Mb16.p, c->mnorm.p, ldm, N, nref, KP, c->sub_land.p, c->Pm.p,
                         shard_q ? part : 0, shard_q ? nparts : 1);
        else if (f32)
            k_pcent_f32(c, c->Xs32.p, c->rns.p, lds_rows, c->Ms32.p, c->mnorm.p, ldm, N, nref, dpad, c->dm_soff.p, c->Pm.p,
                        shard_q ? part : 0, shard_q ? nparts : 1);
        else
            k_pcent(c, c->Xs.p, c->rns.p, lds_rows, c->Ms.p, c->mnorm.p, ldm, N, nref, dpad, c->sub_land.p, c->Pm.p,
                    shard_q ? part : 0, shard_q ? nparts : 1);
    } else
        HIP_CHECK(hipMemsetAsync(c->Pm.p, 0, sizeof(double) * (size_t)(N * nref), st));
    if (shard_q || RS) cge_allreduce_dev(c, c->Pm.p, N * nref, 1);
    lap("dm_refs_pcent");
    // ---- lower bound from the farthest-point sweep ----------------------------------------------------------
    if (!RS) {
        std::swap(c->stream, c->copy_stream);
        try {
            double v;
            i64 q;
            k_farthest_collect(c, &v, &q);
            if (v > L) { L = v; far_i = seed_vertex; far_j = q; }
        } catch (...) {
            std::swap(c->stream, c->copy_stream);
            throw;
        }
        std::swap(c->stream, c->copy_stream);
    }
    lap("dm_farthest");
    // ---- candidate landmark pairs ---------------------------------------------------------------------------
    const i64 cap = std::min<i64>(N * (N + 1) / 2, (i64)4 << 20);
    c->bound_list.ensure((size_t)2 * cap);
    const i64 cnt = k_bound_select(c, c->Pm.p, c->mp_lref.p, mu_ref, N, nref, d, L * (1.0 - 1e-9), c->bound_list.p, cap,
                                   by_comm ? c->mp_refoff.p : nullptr, by_comm ? c->mp_refmem.p : nullptr, nullptr, dpad, ldm, true);
    c->stat_cand_pairs = cnt;
    if (cnt > cap) {
        if (RS) CGE_THROW(CGE_E_ARG, "diameter (shard_rows): the bounds prune too little (more than %lld candidate landmark pairs) and the brute-force "
                                     "fallback needs every row on one rank", (long long)cap);
        return false;
    }
    std::vector<BoundRec> cand(cnt);
    if (cnt > 0) {
        HIP_CHECK(hipMemcpyAsync(cand.data(), c->bound_list.p, sizeof(BoundRec) * cnt, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    lap("dm_select");
    // candidates are consumed in decreasing-bound order (ties: by landmark pair) until the bound drops below the best
    // pair found: a heap delivers exactly that order without sorting the (mostly never visited) tail
    auto later = [](const BoundRec &x, const BoundRec &y) { // x comes after y
        return x.B < y.B || (x.B == y.B && (x.a > y.a || (x.a == y.a && x.b > y.b)));
    };
    std::make_heap(cand.begin(), cand.end(), later);
    auto len16 = [&](i64 a) { return (i64)((gl_off[a + 1] - gl_off[a] + 15) / 16 * 16); }; // positions of the WHOLE landmark
    auto ntiles_of = [&](i64 a) { return (len16(a) + 127) / 128; };
    double tiles_total = 0.0;
    for (const auto &r : cand) {
        const double ta = (double)ntiles_of(r.a), tb = (double)ntiles_of(r.b);
        tiles_total += (r.a == r.b) ? ta * (ta + 1) / 2 : ta * tb;
    }
    const double nT = (double)((n + 127) / 128);
    if (!RS && c->opt_diameter != 2 && tiles_total > 0.5 * nT * (nT + 1) / 2) return false; // pruning too weak: brute force
    // ---- exact evaluation in decreasing-bound order, round by round, on the gathered rows of the round's landmarks --------
    const int xparts = RS ? W : nparts, xpart = RS ? me : part; // who evaluates which tile of a round
    double best = L; // the farthest-point pair is a valid answer so far
    double my_best = -1.0;
    i64 best_gi = -1, best_gj = -1;
    size_t left = cand.size(); // the heap is cand[0, left)
    i64 tile_cap = 4096, global_tile = 0;
    const i64 pos_cap = std::max<i64>(8192, ((i64)512 << 20) / (dpad * 8)); // <= 512 MB of gathered rows per round
    std::vector<int2> tiles;
    std::vector<BoundRec> round;
    std::vector<i32> eoff(N + 1);
    std::vector<char> in_round(N, 0);
    std::vector<i32> lms;
    while (left > 0) {
        round.clear();
        lms.clear();
        i64 E = 0, ntile = 0;
        while (left > 0 && ntile < tile_cap) {
            std::pop_heap(cand.begin(), cand.begin() + left, later);
            const BoundRec r = cand[left - 1];
            if (r.B * (1.0 + 1e-9) + 1e-9 < best) { left = 0; break; } // decreasing order: nothing further can win
            const i64 extra = (in_round[r.a] ? 0 : len16(r.a)) + ((r.b != r.a && !in_round[r.b]) ? len16(r.b) : 0);
            if (!round.empty() && E + extra > pos_cap) { // the round is full: the pair goes back, the next round takes it
                std::push_heap(cand.begin(), cand.begin() + left, later);
                break;
            }
            left--;
            round.push_back(r);
            for (i32 a : {r.a, r.b})
                if (!in_round[a]) { in_round[a] = 1; lms.push_back(a); E += len16(a); }
            const i64 ta = ntiles_of(r.a), tb = ntiles_of(r.b);
            ntile += (r.a == r.b) ? ta * (ta + 1) / 2 : ta * tb;
        }
        if (round.empty()) break;
        // positions of the round: its landmarks in ascending id, each padded to 16 (a landmark outside the round: no position)
        {
            i64 at = 0;
            for (i64 a = 0; a < N; a++) {
                eoff[a] = (i32)at;
                if (in_round[a]) at += len16(a);
            }
            eoff[N] = (i32)at;
        }
        const i64 ldE = (E + 127) / 128 * 128 + 128;
        c->dm_soffE.ensure(N + 1); c->xe_pos.ensure(E); c->xe_glob.ensure(E + 2); c->xe_sub.ensure(ldE / 16);
        c->xe.ensure((size_t)ldE * dpad); c->xe_rns.ensure(ldE);
        {
            WordPacker pk(c);
            pk.add(c->dm_soffE.p, eoff.data(), N + 1);
            pk.flush();
        }
        k_diameter_layout(c, d_off, d_mem, c->dm_soffE.p, N, c->xe_pos.p, c->xe_sub.p, ldE / 16);
        k_gather_centre_fm(c, c->Xr.p, c->xe_pos.p, c->gmean.p, c->xe.p, c->xe_rns.p, E, d, ldE, dpad);
        k_position_ids(c, c->xe_pos.p, RS ? c->loc2glob.p : nullptr, E, c->xe_glob.p);
        if (RS) { // every rank has filled the rows it owns: the ranks add the words (zeros elsewhere: exact)
            HIP_CHECK(hipMemsetAsync(c->xe_glob.p + E, 0, sizeof(i32) * 2, st));
            cge_allreduce_dev(c, c->xe.p, ldE * dpad, 2);
            cge_allreduce_dev(c, c->xe_rns.p, ldE, 2);
            cge_allreduce_dev(c, reinterpret_cast<double *>(c->xe_glob.p), (E + 1) / 2, 2);
        }
        tiles.clear();
        for (const BoundRec &r : round) {
            const i64 ta = ntiles_of(r.a), tb = ntiles_of(r.b);
            for (i64 x = 0; x < ta; x++)
                for (i64 y = (r.a == r.b ? x : 0); y < tb; y++) {
                    if ((global_tile++ % xparts) != xpart) continue;
                    tiles.push_back(make_int2((int)(eoff[r.a] + 128 * x), (int)(eoff[r.b] + 128 * y)));
                }
        }
        for (i32 a : lms) in_round[a] = 0;
        if (!tiles.empty()) {
            c->stat_cand_tiles += (i64)tiles.size();
            c->tile_list.ensure(2 * tiles.size());
            {
                WordPacker pk(c);
                pk.add(reinterpret_cast<i32 *>(c->tile_list.p), reinterpret_cast<const i32 *>(tiles.data()), (i64)(2 * tiles.size()));
                pk.flush();
            }
            double v;
            i64 pi, pj;
            k_pair_list(c, c->xe.p, c->xe_rns.p, ldE, E, dpad, c->tile_list.p, (i64)tiles.size(), &v, &pi, &pj);
            if (v > best) {
                i32 g2[2] = {0, 0};
                HIP_CHECK(hipMemcpyAsync(&g2[0], c->xe_glob.p + pi, sizeof(i32), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipMemcpyAsync(&g2[1], c->xe_glob.p + pj, sizeof(i32), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                best = my_best = v;
                best_gi = (i64)g2[0] - 1;
                best_gj = (i64)g2[1] - 1;
            }
        }
        if (RS) best = cge_allreduce_scalar_max(c, best); // the rounds are taken in lock-step with one threshold
        tile_cap = 131072;
    }
    lap("dm_exact");
    if (RS && best > L) { // somebody's tile beat the sweep: the lowest rank that holds the best value names the pair
        const double claim = cge_allreduce_scalar_max(c, my_best == best ? (double)(W - me) : 0.0);
        const bool mine = (W - (int)claim) == me;
        const double gi = cge_allreduce_scalar_max(c, mine ? (double)(best_gi + 1) : 0.0);
        const double gj = cge_allreduce_scalar_max(c, mine ? (double)(best_gj + 1) : 0.0);
        far_i = (i64)gi - 1;
        far_j = (i64)gj - 1;
    } else if (!RS && best_gi >= 0) {
        far_i = best_gi;
        far_j = best_gj;
    }
    if (far_i > far_j) std::swap(far_i, far_j);
    *best_d2 = best;
    *bi = far_i;
    *bj = far_j;
    return true;
}
