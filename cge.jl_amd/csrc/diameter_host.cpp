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
bool host_diameter_pruned(cge_ctx *c, const double *mu, const double *lw, const std::vector<i32> &lcomm, i64 C, i64 N,
                          const std::vector<i32> &mem_off, const std::vector<i32> &mem, int part, int nparts,
                          double *best_d2, i64 *bi, i64 *bj) {
    const i64 n = c->n, d = c->d, dpad = c->dpad;
    hipStream_t st = c->stream;
    c->stat_cand_pairs = c->stat_cand_tiles = 0;
    double tphase = now_ms();
    auto lap = [&](const char *name) { // host wall time per stage (the stream is not synchronised here)
        const double t = now_ms();
        c->phases.ms[name] += t - tphase;
        tphase = t;
    };
    // ---- landmark-sorted layout: landmark a owns positions [soff[a], soff[a] + cnt16[a]) ------------
    std::vector<i64> soff(N + 1, 0);
    for (i64 a = 0; a < N; a++) soff[a + 1] = soff[a] + ((mem_off[a + 1] - mem_off[a] + 15) / 16) * 16;
    const i64 npos = soff[N];
    const i64 lds_rows = (npos + 127) / 128 * 128 + 128;
    c->pos2node.ensure(npos);
    c->sub_land.ensure(lds_rows / 16);
    {
        std::vector<i32> soff32(soff.begin(), soff.end());
        c->dm_soff.ensure(N + 1);
        HIP_CHECK(hipMemcpyAsync(c->dm_soff.p, soff32.data(), sizeof(i32) * (N + 1), hipMemcpyHostToDevice, st));
        const i32 *d_off = c->lm_memoff.p, *d_mem = c->lm_mem.p;
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
        HIP_CHECK(hipStreamSynchronize(st)); // soff32 goes out of scope
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
        HIP_CHECK(hipMemcpyAsync(c->mp_refoff.p, roff.data(), sizeof(i32) * (C + 1), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(c->mp_refmem.p, rmem.data(), sizeof(i32) * N, hipMemcpyHostToDevice, st));
        k_ref_centroids(c, mu, lw, c->mp_refoff.p, c->mp_refmem.p, C, d, c->mp_refmu.p);
        HIP_CHECK(hipStreamSynchronize(st)); // roff / rmem go out of scope
        mu_ref = c->mp_refmu.p;
    } else
        for (i64 a = 0; a < N; a++) lref[a] = (i32)a;
    c->mp_lref.ensure(N);
    HIP_CHECK(hipMemcpyAsync(c->mp_lref.p, lref.data(), sizeof(i32) * N, hipMemcpyHostToDevice, st));
    const i64 ldm = (nref + 127) / 128 * 128;
    c->Xs.ensure((size_t)lds_rows * dpad);
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
    }
    k_gather_centre_fm(c, c->Xr.p, c->pos2node.p, c->gmean.p, c->Xs.p, c->rns.p, npos, d, lds_rows, dpad, f32 ? c->Xs32.p : nullptr,
                       b16 ? c->Xb16.p : nullptr, KP, lowp ? c->dm_flag.p : nullptr);
    k_gather_centre_fm(c, mu_ref, nullptr, c->gmean.p, c->Ms.p, c->mnorm.p, nref, d, ldm, dpad, f32 ? c->Ms32.p : nullptr,
                       b16 ? c->Mb16.p : nullptr, KP, lowp ? c->dm_flag.p : nullptr);
    int unfit = 0;
    if (lowp) HIP_CHECK(hipMemcpyAsync(&unfit, c->dm_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    // the seed of the farthest-point sweep: the vertex farthest from the centre (largest squared norm of the centred rows, just
    // computed by the gather); its read-back is the synchronisation the flag needs anyway
    const i64 seed_vertex = k_argmax_mapped(c, c->rns.p, npos, c->pos2node.p);
    // The exact fp64 pass (below) instead of a low-precision one when a centred value lies beyond 2^+-100 or is not finite
    // (bit 0), or when NO centred value reaches 2^-40 (bit 1 clear): products below 2^-126 are flushed to zero in the fp32
    // accumulators, an absolute error of < 3K 2^-126 per dot product that the relative margin e only covers while the
    // distances that matter (>= L >= the largest centred norm squared >= 2^-80) dwarf it.
    if ((unfit & 1) || (lowp && !(unfit & 2))) b16 = f32 = false;
    // Q is a maximum over vertices: with several ranks each takes its share of the vertex tiles and the maxima are
    // combined by one all-reduce(max) (only when the exchange buffer can hold N x nref doubles)
    const bool shard_q = nparts > 1 && c->has_coll && (c->rccl_comm || (c->xptr && (size_t)(N * nref) <= c->xcap));
    c->stat_bound_pass = b16 ? 2 : (f32 ? 1 : 0);
    // One farthest-point sweep from the vertex farthest from the centre (a memory-bound read of Xr) on the side stream, LAUNCHED
    // AHEAD of the bound pass: its small workgroups then fill the CUs first and the sweep runs at its own speed (~0.3 ms) while
    // the bound pass's one-workgroup-per-CU tiles move in beside them -- launched behind it, the sweep was left the gaps
    // (1.0 ms, the longer of the two concurrent branches).  Its result is collected after the bound pass has been enqueued.
    std::swap(c->stream, c->copy_stream);
    try {
        k_farthest_enqueue(c, c->Xr.p, n, d, seed_vertex);
    } catch (...) {
        std::swap(c->stream, c->copy_stream);
        throw;
    }
    std::swap(c->stream, c->copy_stream);
    if (b16)
        k_pcent_bf16(c, c->Xb16.p, c->rns.p, lds_rows, c->Mb16.p, c->mnorm.p, ldm, N, nref, KP, c->sub_land.p, c->Pm.p,
                     shard_q ? part : 0, shard_q ? nparts : 1);
    else if (f32)
        k_pcent_f32(c, c->Xs32.p, c->rns.p, lds_rows, c->Ms32.p, c->mnorm.p, ldm, N, nref, dpad, c->dm_soff.p, c->Pm.p,
                    shard_q ? part : 0, shard_q ? nparts : 1);
    else
        k_pcent(c, c->Xs.p, c->rns.p, lds_rows, c->Ms.p, c->mnorm.p, ldm, N, nref, dpad, c->sub_land.p, c->Pm.p,
                shard_q ? part : 0, shard_q ? nparts : 1);
    if (shard_q) cge_allreduce_dev(c, c->Pm.p, N * nref, 1);
    lap("dm_refs_pcent");
    // ---- lower bound from farthest-point sweeps ----------------------------------------------------------
    double L = 0.0;
    i64 p0 = seed_vertex, far_i = 0, far_j = 0;
    std::swap(c->stream, c->copy_stream);
    try {
        double v;
        i64 q;
        k_farthest_collect(c, &v, &q);
        if (v > L) { L = v; far_i = p0; far_j = q; }
    } catch (...) {
        std::swap(c->stream, c->copy_stream);
        throw;
    }
    std::swap(c->stream, c->copy_stream);
    lap("dm_farthest");
    // ---- candidate landmark pairs ---------------------------------------------------------------------------
    const i64 cap = std::min<i64>(N * (N + 1) / 2, (i64)4 << 20);
    c->bound_list.ensure((size_t)2 * cap);
    const i64 cnt = k_bound_select(c, c->Pm.p, c->mp_lref.p, mu_ref, N, nref, d, L * (1.0 - 1e-9), c->bound_list.p, cap,
                                   by_comm ? c->mp_refoff.p : nullptr, by_comm ? c->mp_refmem.p : nullptr, c->Ms.p, dpad, ldm);
    c->stat_cand_pairs = cnt;
    if (cnt > cap) return false;
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
    auto ntiles_of = [&](i64 a) { return (soff[a + 1] - soff[a] + 127) / 128; };
    double tiles_total = 0.0;
    for (const auto &r : cand) {
        const double ta = (double)ntiles_of(r.a), tb = (double)ntiles_of(r.b);
        tiles_total += (r.a == r.b) ? ta * (ta + 1) / 2 : ta * tb;
    }
    const double nT = (double)((n + 127) / 128);
    if (c->opt_diameter != 2 && tiles_total > 0.5 * nT * (nT + 1) / 2) return false; // pruning too weak: brute force
    // ---- exact evaluation in decreasing-bound order ----------------------------------------------------------
    double best = L; // the farthest-point pair is a valid answer so far
    i64 best_pi = -1, best_pj = -1;
    size_t left = cand.size(); // the heap is cand[0, left)
    i64 chunk_cap = 4096, global_tile = 0;
    std::vector<int2> tiles;
    while (left > 0) {
        tiles.clear();
        while (left > 0 && (i64)tiles.size() < chunk_cap) {
            std::pop_heap(cand.begin(), cand.begin() + left, later);
            const BoundRec r = cand[--left];
            if (r.B * (1.0 + 1e-9) + 1e-9 < best) { left = 0; break; } // decreasing order: nothing further can win
            const i64 ta = ntiles_of(r.a), tb = ntiles_of(r.b);
            for (i64 x = 0; x < ta; x++)
                for (i64 y = (r.a == r.b ? x : 0); y < tb; y++) {
                    if ((global_tile++ % nparts) != part) continue;
                    tiles.push_back(make_int2((int)(soff[r.a] + 128 * x), (int)(soff[r.b] + 128 * y)));
                }
        }
        if (tiles.empty()) continue;
        c->stat_cand_tiles += (i64)tiles.size();
        c->tile_list.ensure(2 * tiles.size());
        HIP_CHECK(hipMemcpyAsync(c->tile_list.p, tiles.data(), sizeof(int2) * tiles.size(), hipMemcpyHostToDevice, st));
        double v;
        i64 pi, pj;
        k_pair_list(c, c->Xs.p, c->rns.p, lds_rows, npos, dpad, c->tile_list.p, (i64)tiles.size(), &v, &pi, &pj);
        if (v > best) { best = v; best_pi = pi; best_pj = pj; }
        chunk_cap = 131072;
    }
    lap("dm_exact");
    if (best_pi >= 0) {
        i32 a = 0, b = 0;
        HIP_CHECK(hipMemcpyAsync(&a, c->pos2node.p + best_pi, sizeof(i32), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(&b, c->pos2node.p + best_pj, sizeof(i32), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        far_i = a;
        far_j = b;
    }
    if (far_i > far_j) std::swap(far_i, far_j);
    *best_d2 = best;
    *bi = far_i;
    *bj = far_j;
    return true;
}
