// kernels_louvain.hip -- level 1 of Louvain on the resident graph: the communities `louvain_clust` writes to <file>.ecg
// when the command line has no `-c` (src/clustering.jl:14-68, src/auxilary.jl:115-121).
//
// The reference shells out to the "generic Louvain" executables (convert / louvain -l -1 -q 0 / hierarchy -l 1): the
// partition after the FIRST pass of local moving, nodes visited one at a time in a rand()-shuffled order.  On the GPU the
// pass is synchronous: in every round each vertex looks at the communities of its neighbours as they stood at the start
// of the round (its adjacency segment sorted by community: a segmented radix sort, so the sums over a community are
// contiguous runs and the choice is deterministic), takes the one of largest gain  k_vc - tot_c k_v / 2m  (its own
// community first, so ties stay; the smallest id among equal gains otherwise), and all moves are applied together.  Two
// guards keep simultaneous moves from chasing each other: two singletons only merge towards the smaller id, and a vertex
// may only move in every other round (by a hash of its id).  The host keeps the best partition seen (synchronous rounds
// are not monotone in the modularity) and stops after three rounds without an improvement of more than 1e-6 (the
// reference's own threshold) or when nothing moves.  Not bit-comparable with the reference (whose visiting order is
// random): tests compare modularity and structure with the sequential restatement in the oracle and with networkx.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

__global__ void lv_count_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst, i64 m, i32 *__restrict__ cnt) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const i32 u = src[e], v = dst[e];
        if (u == v) continue; // self loops are kept aside (lv_fill_kernel)
        atomicAdd(&cnt[u], 1);
        atomicAdd(&cnt[v], 1);
    }
}
__global__ void lv_fill_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst, const double *__restrict__ w, i64 m,
                               const i32 *__restrict__ off, i32 *__restrict__ cur, i32 *__restrict__ adj,
                               double *__restrict__ aw, double *__restrict__ self) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const i32 u = src[e], v = dst[e];
        const double we = w ? w[e] : 1.0;
        if (u == v) { unsafeAtomicAdd(&self[u], we); continue; }
        const i32 pu = off[u] + atomicAdd(&cur[u], 1), pv = off[v] + atomicAdd(&cur[v], 1);
        adj[pu] = v; adj[pv] = u;
        if (aw) { aw[pu] = we; aw[pv] = we; }
    }
}
// weighted degree k_v (self loop counted once, as `convert` stores it), tot_c = k_v for the singleton start
__global__ void lv_degree_kernel(const i32 *__restrict__ off, const double *__restrict__ aw, const double *__restrict__ self,
                                 i64 n, double *__restrict__ k, double *__restrict__ tot, i32 *__restrict__ comm,
                                 i32 *__restrict__ size, double *__restrict__ m2) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    double kv = 0.0;
    if (v < n) {
        if (aw) for (i32 q = off[v]; q < off[v + 1]; q++) kv += aw[q];
        else kv = (double)(off[v + 1] - off[v]);
        kv += self[v];
        k[v] = kv;
        tot[v] = kv;
        comm[v] = (i32)v;
        size[v] = 1;
    }
    for (int o = 32; o > 0; o >>= 1) kv += __shfl_down(kv, o);
    if ((threadIdx.x & 63) == 0 && kv != 0.0) unsafeAtomicAdd(m2, kv);
}
__global__ void lv_keys_kernel(const i32 *__restrict__ adj, const i32 *__restrict__ comm, i64 len, unsigned *__restrict__ keys) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < len; q += stride) keys[q] = (unsigned)comm[adj[q]];
}
// one thread per vertex: the runs of its community-sorted adjacency segment -> the community of largest gain.
// stats[0] += moves, stats[1] += sum_v (k_{v,own} + self_v)  (the "in" part of the modularity of the CURRENT partition)
__global__ void lv_decide_kernel(const i32 *__restrict__ off, const unsigned *__restrict__ skeys, const double *__restrict__ saw,
                                 const double *__restrict__ self, const double *__restrict__ k, const double *__restrict__ tot,
                                 const i32 *__restrict__ size, const i32 *__restrict__ comm, i64 n, double m2, int round,
                                 i32 *__restrict__ newcomm, double *__restrict__ stats) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    double in_part = 0.0, moved = 0.0;
    if (v < n) {
        const i32 own = comm[v];
        const double kv = k[v];
        // pass 1: weight towards the own community
        double w_own = 0.0;
        for (i32 q = off[v]; q < off[v + 1]; q++)
            if ((i32)skeys[q] == own) w_own += saw ? saw[q] : 1.0;
        in_part = w_own + self[v];
        i32 best = own;
        double best_inc = w_own - (tot[own] - kv) * kv / m2;
        // pass 2: the other communities, ascending id (the segment is sorted), strict improvement only
        i32 q = off[v];
        const i32 qe = off[v + 1];
        while (q < qe) {
            const i32 c = (i32)skeys[q];
            double ws = 0.0;
            while (q < qe && (i32)skeys[q] == c) { ws += saw ? saw[q] : 1.0; q++; }
            if (c == own) continue;
            const double inc = ws - tot[c] * kv / m2;
            if (inc > best_inc) { best = c; best_inc = inc; }
        }
        const bool my_turn = ((((unsigned)v * 2654435761u) >> 15) & 1u) == (unsigned)(round & 1);
        if (best != own && (!my_turn || (size[own] == 1 && size[best] == 1 && best > own))) best = own;
        newcomm[v] = best;
        moved = best != own ? 1.0 : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) { in_part += __shfl_down(in_part, o); moved += __shfl_down(moved, o); }
    if ((threadIdx.x & 63) == 0) {
        if (moved != 0.0) unsafeAtomicAdd(&stats[0], moved);
        if (in_part != 0.0) unsafeAtomicAdd(&stats[1], in_part);
    }
}
__global__ void lv_apply_kernel(i32 *__restrict__ comm, const i32 *__restrict__ newcomm, const double *__restrict__ k, i64 n,
                                double *__restrict__ tot, i32 *__restrict__ size) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const i32 a = comm[v], b = newcomm[v];
    if (a == b) return;
    unsafeAtomicAdd(&tot[a], -k[v]);
    unsafeAtomicAdd(&tot[b], k[v]);
    atomicAdd(&size[a], -1);
    atomicAdd(&size[b], 1);
    comm[v] = b;
}
// stats[2] += sum_c tot_c^2 over the non-empty communities
__global__ void lv_totsq_kernel(const double *__restrict__ tot, const i32 *__restrict__ size, i64 n, double *__restrict__ stats) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    double s = (c < n && size[c] > 0) ? tot[c] * tot[c] : 0.0;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s != 0.0) unsafeAtomicAdd(&stats[2], s);
}
__global__ void lv_used_kernel(const i32 *__restrict__ comm, i64 n, i32 *__restrict__ used) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) used[comm[v]] = 1;
}
__global__ void lv_renumber_kernel(const i32 *__restrict__ comm, const i32 *__restrict__ newid, i64 n, i64 *__restrict__ out) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) out[v] = newid[comm[v]];
}

// comm_out[v] in 0 .. *n_comm-1 (ascending order of the old community ids, as the reference's renumbering), *quality =
// modularity of the returned partition, *rounds = synchronous rounds run
void k_louvain_level1(cge_ctx *c, i64 *comm_out_host, i64 *n_comm, double *quality, i64 *rounds_out) {
    const i64 n = c->n, m = c->m;
    hipStream_t st = c->stream;
    const unsigned gv = (unsigned)((n + 255) / 256);
    DevBuf<i32> off, cur, adj, comm, newcomm, size, best, used, newid;
    DevBuf<unsigned> keys, skeys;
    DevBuf<double> aw, saw, self, k, tot, stats;
    DevBuf<i64> out;
    off.ensure(n + 1); cur.ensure(n);
    HIP_CHECK(hipMemsetAsync(off.p, 0, sizeof(i32) * (n + 1), st));
    HIP_CHECK(hipMemsetAsync(cur.p, 0, sizeof(i32) * n, st));
    hipLaunchKernelGGL(lv_count_kernel, dim3(grid_for(m, 256)), dim3(256), 0, st, c->src.p, c->dst.p, m, off.p + 1);
    {
        size_t bytes = 0;
        HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, off.p, off.p, (size_t)(n + 1), rocprim::plus<i32>(), st));
        c->sort_tmp.ensure(bytes);
        HIP_CHECK(rocprim::inclusive_scan(c->sort_tmp.p, bytes, off.p, off.p, (size_t)(n + 1), rocprim::plus<i32>(), st));
    }
    i32 len32 = 0;
    HIP_CHECK(hipMemcpyAsync(&len32, off.p + n, sizeof(i32), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const i64 len = len32;
    const bool weighted = !c->unit_weights;
    adj.ensure(len); keys.ensure(len); skeys.ensure(len);
    if (weighted) { aw.ensure(len); saw.ensure(len); }
    self.ensure(n); k.ensure(n); tot.ensure(n); comm.ensure(n); newcomm.ensure(n); size.ensure(n); best.ensure(n);
    stats.ensure(4);
    HIP_CHECK(hipMemsetAsync(self.p, 0, sizeof(double) * n, st));
    HIP_CHECK(hipMemsetAsync(stats.p, 0, sizeof(double) * 4, st));
    hipLaunchKernelGGL(lv_fill_kernel, dim3(grid_for(m, 256)), dim3(256), 0, st, c->src.p, c->dst.p, weighted ? c->w.p : nullptr, m,
                       off.p, cur.p, adj.p, weighted ? aw.p : nullptr, self.p);
    hipLaunchKernelGGL(lv_degree_kernel, dim3(gv), dim3(256), 0, st, off.p, weighted ? aw.p : nullptr, self.p, n, k.p, tot.p, comm.p,
                       size.p, stats.p + 3);
    double hs[4];
    HIP_CHECK(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const double m2 = hs[3];
    int bits = 1;
    while (((i64)1 << bits) < n) bits++;
    double best_q = -1e300;
    int since_best = 0;
    i64 rounds = 0;
    if (m2 > 0.0 && len > 0) {
        for (int round = 0; round < 200; round++) {
            rounds = round + 1;
            hipLaunchKernelGGL(lv_keys_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, adj.p, comm.p, len, keys.p);
            size_t bytes = 0;
            if (weighted) {
                HIP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, bytes, keys.p, skeys.p, aw.p, saw.p, (unsigned)len, (unsigned)n,
                                                              off.p, off.p + 1, 0, bits, st));
                c->sort_tmp.ensure(bytes);
                HIP_CHECK(rocprim::segmented_radix_sort_pairs(c->sort_tmp.p, bytes, keys.p, skeys.p, aw.p, saw.p, (unsigned)len,
                                                              (unsigned)n, off.p, off.p + 1, 0, bits, st));
            } else {
                HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, bytes, keys.p, skeys.p, (unsigned)len, (unsigned)n, off.p,
                                                             off.p + 1, 0, bits, st));
                c->sort_tmp.ensure(bytes);
                HIP_CHECK(rocprim::segmented_radix_sort_keys(c->sort_tmp.p, bytes, keys.p, skeys.p, (unsigned)len, (unsigned)n, off.p,
                                                             off.p + 1, 0, bits, st));
            }
            HIP_CHECK(hipMemsetAsync(stats.p, 0, sizeof(double) * 3, st));
            hipLaunchKernelGGL(lv_decide_kernel, dim3(gv), dim3(256), 0, st, off.p, skeys.p, weighted ? saw.p : nullptr, self.p, k.p,
                               tot.p, size.p, comm.p, n, m2, round, newcomm.p, stats.p);
            hipLaunchKernelGGL(lv_totsq_kernel, dim3(gv), dim3(256), 0, st, tot.p, size.p, n, stats.p);
            HIP_CHECK(hipMemcpyAsync(hs, stats.p, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const double q = hs[1] / m2 - hs[2] / (m2 * m2); // modularity of the partition this round STARTED from
            if (q > best_q + 1e-6) {
                best_q = q;
                since_best = 0;
                HIP_CHECK(hipMemcpyAsync(best.p, comm.p, sizeof(i32) * n, hipMemcpyDeviceToDevice, st));
            } else if (++since_best >= 3)
                break;
            if (hs[0] == 0.0) break; // nothing wants to move
            hipLaunchKernelGGL(lv_apply_kernel, dim3(gv), dim3(256), 0, st, comm.p, newcomm.p, k.p, n, tot.p, size.p);
        }
    } else {
        best_q = 0.0;
        HIP_CHECK(hipMemcpyAsync(best.p, comm.p, sizeof(i32) * n, hipMemcpyDeviceToDevice, st));
    }
    // renumber 0.. in ascending order of the surviving community ids
    used.ensure(n + 1); newid.ensure(n + 1); out.ensure(n);
    HIP_CHECK(hipMemsetAsync(used.p, 0, sizeof(i32) * (n + 1), st));
    hipLaunchKernelGGL(lv_used_kernel, dim3(gv), dim3(256), 0, st, best.p, n, used.p);
    {
        size_t bytes = 0;
        HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, used.p, newid.p, (i32)0, (size_t)(n + 1), rocprim::plus<i32>(), st));
        c->sort_tmp.ensure(bytes);
        HIP_CHECK(rocprim::exclusive_scan(c->sort_tmp.p, bytes, used.p, newid.p, (i32)0, (size_t)(n + 1), rocprim::plus<i32>(), st));
    }
    hipLaunchKernelGGL(lv_renumber_kernel, dim3(gv), dim3(256), 0, st, best.p, newid.p, n, out.p);
    i32 nc = 0;
    HIP_CHECK(hipMemcpyAsync(&nc, newid.p + n, sizeof(i32), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(comm_out_host, out.p, sizeof(i64) * n, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (n_comm) *n_comm = nc;
    if (quality) *quality = best_q;
    if (rounds_out) *rounds_out = rounds;
}
