// kernels_scatter.hip -- the per-edge cluster-pair scatter-add  vect_C[bin(c_u, c_v)] += w  of wGCL / wGCL_directed
// (src/divergence.jl:55-63, :337-345) over the RESIDENT edge list, without global atomics and without per-edge gathers
// from a table in memory.
//
// Data layout (built once per resident graph, by the first edge pass after cge_set_graph):
//   * the edge list is kept a second time BLOCKED: edges grouped by (block of 131072 source vertices, block of 32768
//     target vertices), sorted by source inside a group, one 32-bit word (u mod 131072) << 15 | (v mod 32768) per edge --
//     4 bytes instead of the reference's 16 (two Int64), plus the weight (8 bytes) only for a weighted list.  A group is
//     cut into chunks of <= 16384 edges.
//   * the community table is uint16 (C <= 2048 on this path), padded to a multiple of 32768 entries.
// Pass 1 (edge_pass_kernel, one 1024-thread workgroup per chunk, two per CU): the 64 KB slice of the community table that
//   the chunk's TARGETS can touch is copied into LDS (coalesced 16-byte loads); the communities of the SOURCES come from
//   memory with ascending addresses (the chunk is sorted by source: a wave's loads fall into a few cache lines).  Intra-
//   community edges (the bulk of a graph with community structure) are counted / summed in LDS per community; the others
//   are counted per row (= smaller community), then written grouped by row (the column as uint16, staged in LDS and
//   copied out coalesced; the weight beside it when weighted) into the chunk's own piece of a key array, with the per-row
//   offsets beside it.
// Pass 2 (edge_row_reduce_kernel, grid = rows x batches of 256 chunks, one chunk per thread): adds the chunk's keys of the
//   row into the workgroup's LDS copy of the row and the workgroup adds that into vect_C with contiguous atomics.
// No per-edge atomic ever leaves the CU.  Unit weights: all sums are exact integers, results are bitwise reproducible.
// Weighted lists: additions of one bin are unordered (last-bit differences for non-dyadic weights), as with the per-edge
// atomics this replaces.
#include "common.hpp"

#define EB_MAXC 2048
#define EB_NONE 0xFFFFFFFFu
#define EB_RBATCH 256 // chunks per workgroup of the row reduction (one per thread)
#define EB_RKEYS 16   // keys a thread of the row reduction requests at once

#define EB_UBITS 17 // source block: 131072 vertices
#define EB_VBITS 15 // target block: 32768 vertices (a 64 KB uint16 slice in LDS)
#define EB_THREADS 1024

// sort key of an edge: (tile, source inside the tile); the value is the edge's row in the list
__global__ void eb_tile_keys_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst, i64 m, unsigned nbv,
                                    unsigned *__restrict__ keys, i32 *__restrict__ idx) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const unsigned u = (unsigned)src[e], v = (unsigned)dst[e];
        keys[e] = (((u >> EB_UBITS) * nbv + (v >> EB_VBITS)) << EB_UBITS) | (u & ((1u << EB_UBITS) - 1u));
        idx[e] = (i32)e;
    }
}
// blocked words (and weights) in sorted order; first position of every tile that occurs
__global__ void eb_gather_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst, const double *__restrict__ w,
                                 const unsigned *__restrict__ skeys, const i32 *__restrict__ sidx, i64 m,
                                 unsigned *__restrict__ bedge, double *__restrict__ bw, i32 *__restrict__ tile_first) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
        const i32 e = sidx[k];
        bedge[k] = (((unsigned)src[e] & ((1u << EB_UBITS) - 1u)) << EB_VBITS) | ((unsigned)dst[e] & ((1u << EB_VBITS) - 1u));
        if (bw) bw[k] = w[e];
        const unsigned t = skeys[k] >> EB_UBITS;
        if (k == 0 || t != (skeys[k - 1] >> EB_UBITS)) tile_first[t] = (i32)k;
    }
}

void k_sort_pairs_u32(cge_ctx *c, const unsigned *keys_in, unsigned *keys_out, const i32 *vals_in, i32 *vals_out, i64 n,
                      int bits); // kernels_sort.hip (rocPRIM)

// Build the blocked copy of the resident edge list.  Returns false when this path does not apply (the caller then uses
// the gather + atomics kernel): more than 32768 tiles (n > ~1.4e7) or an edge count beyond int32.
bool k_build_blocked_edges(cge_ctx *c) {
    const i64 n = c->n, m = c->m;
    const i64 nbu = (n + (1 << EB_UBITS) - 1) >> EB_UBITS, nbv = (n + (1 << EB_VBITS) - 1) >> EB_VBITS;
    if (nbu * nbv > 32768 || m <= 0 || m >= (1LL << 31)) return false; // the sort key (tile, u mod 131072) has 32 bits
    hipStream_t st = c->stream;
    HIP_CHECK(hipStreamSynchronize(st)); // so that the wall time below is the build's own (stat "edge_layout_build_us")
    const double t_build0 = now_ms();
    const i64 T = nbu * nbv;
    // 16 edges per thread: with 20 or 24 the edge pass no longer fits 64 registers (two workgroups per CU) and spills
    const int per_thread = 16;
    c->be_per = per_thread;
    const i64 EB_CHUNK = (i64)EB_THREADS * per_thread;
    DevBuf<unsigned> keys, skeys;
    DevBuf<i32> idx, sidx, tfirst;
    keys.ensure(m); skeys.ensure(m); idx.ensure(m); sidx.ensure(m); tfirst.ensure(T + 1);
    hipLaunchKernelGGL(eb_tile_keys_kernel, dim3(grid_for(m, 256)), dim3(256), 0, st, c->src.p, c->dst.p, m, (unsigned)nbv, keys.p,
                       idx.p);
    int bits = 1;
    while (((i64)1 << bits) < T) bits++;
    k_sort_pairs_u32(c, keys.p, skeys.p, idx.p, sidx.p, m, bits + EB_UBITS);
    c->be_edge.alloc_exact(m);
    if (!c->unit_weights) c->be_w.alloc_exact(m); else c->be_w.release();
    HIP_CHECK(hipMemsetAsync(tfirst.p, 0xFF, sizeof(i32) * (T + 1), st));
    hipLaunchKernelGGL(eb_gather_kernel, dim3(grid_for(m, 256)), dim3(256), 0, st, c->src.p, c->dst.p,
                       c->unit_weights ? nullptr : c->w.p, skeys.p, sidx.p, m, c->be_edge.p, c->unit_weights ? nullptr : c->be_w.p,
                       tfirst.p);
    std::vector<i32> tf(T + 1);
    HIP_CHECK(hipMemcpyAsync(tf.data(), tfirst.p, sizeof(i32) * (T + 1), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    tf[T] = (i32)m;
    for (i64 t = T - 1; t >= 0; t--)
        if (tf[t] < 0) tf[t] = tf[t + 1]; // tiles without edges
    std::vector<i32> ch;
    const i64 ideal = 0; // (a target chunk count was an A/B knob of round 2)
    for (i64 t = 0; t < T; t++) {
        const i64 len = tf[t + 1] - tf[t];
        if (len <= 0) continue;
        i64 parts = (len + EB_CHUNK - 1) / EB_CHUNK;
        if (ideal > 0) parts = std::max(parts, (len + ideal / 2) / ideal);
        const i64 per = (len + parts - 1) / parts; // equal pieces
        for (i64 s = 0; s < len; s += per) {
            ch.push_back((i32)(t / nbv)); ch.push_back((i32)(t % nbv));
            ch.push_back((i32)(tf[t] + s)); ch.push_back((i32)std::min(per, len - s));
        }
    }
    c->be_nchunks = (i64)ch.size() / 4;
    c->be_chunk.alloc_exact(ch.size());
    HIP_CHECK(hipMemcpyAsync(c->be_chunk.p, ch.data(), sizeof(i32) * ch.size(), hipMemcpyHostToDevice, st));
    c->be_keys.ensure(m + 64); // + a readable tail (the row reduction clamps empty runs to their start)
    if (!c->unit_weights) c->be_wkeys.ensure(m + 64);
    HIP_CHECK(hipStreamSynchronize(st));
    c->blocked_ready = true;
    c->stat_layout_build_us = (i64)((now_ms() - t_build0) * 1e3);
    return true;
}

// exclusive scan over the THREADS threads of a workgroup; returns the exclusive value, *total (LDS) = block sum
template <int THREADS>
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *wave_sums, unsigned *total) {
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const unsigned x = wave_sums[w];
        if (w < wave) before += x;
        all += x;
    }
    if (threadIdx.x == 0) *total = all;
    __syncthreads();
    return before + inc - v;
}

// ---- pass 1 ------------------------------------------------------------------------------------------------------------
// LDS: sv (32768 uint16: communities of the target block; reused as the key staging area once the look-ups are done) |
//      dsum[Cpad] f64 (weighted lists: intra-community sums) | cnt[2 Cpad] u32: [0, Cpad) off-diagonal edges per row (then
//      the row cursors), [Cpad, 2 Cpad) intra-community edge counts (unit weights) | wave sums, total
template <int EB_PER, bool WEIGHTED, bool DIRECTED>
__global__ __launch_bounds__(EB_THREADS, WEIGHTED ? 4 : 8) void edge_pass_kernel(const unsigned *__restrict__ bedge, const double *__restrict__ bw,
                                                               const i32 *__restrict__ chunks, int chunk0,
                                                               const unsigned short *__restrict__ comm16, int C, int Cpad,
                                                               unsigned short *__restrict__ keys, double *__restrict__ wkeys,
                                                               unsigned short *__restrict__ runoff,
                                                               double *__restrict__ diag_out, double *__restrict__ vectC,
                                                               i64 vlen, int stop /* timing diagnostics only */) {
    constexpr int VBLOCK = 1 << EB_VBITS, EPT = EB_MAXC / EB_THREADS > 0 ? EB_MAXC / EB_THREADS : 1;
    static_assert(VBLOCK >= EB_THREADS * EB_PER, "the key staging area reuses the slice");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short *sv = (unsigned short *)lds;
    unsigned short *stage = sv;
    double *dsum = (double *)(lds + 2 * VBLOCK);
    unsigned *cnt = (unsigned *)(dsum + (WEIGHTED ? Cpad : 0));
    unsigned *wave_sums = cnt + 2 * Cpad;
    unsigned *misc = wave_sums + 16; // [0] total
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    const i32 *ch = chunks + 4 * (i64)(chunk0 + wg);
    const int bu = ch[0], bv = ch[1], start = ch[2], len = ch[3];
    // All the edge words of a thread are requested up front, ahead of the table slice (a load under `if (e < len)` would
    // wait for each in turn), and the source communities right behind them.
    unsigned packed[EB_PER];
    double wv[WEIGHTED ? EB_PER : 1];
#pragma unroll
    for (int k = 0; k < EB_PER; k++) {
        const int e = tid + k * EB_THREADS;
        const int ec = e < len ? e : len - 1;
        packed[k] = bedge[start + ec];
        if (WEIGHTED) wv[k] = bw[start + ec];
    }
    {
        const uint4 *gv = (const uint4 *)(comm16 + (i64)bv * VBLOCK);
        uint4 *lv = (uint4 *)sv;
        for (int i = tid; i < VBLOCK / 8; i += EB_THREADS) lv[i] = gv[i];
    }
    for (int k = tid; k < 2 * Cpad; k += EB_THREADS) cnt[k] = 0u;
    if (WEIGHTED)
        for (int k = tid; k < Cpad; k += EB_THREADS) dsum[k] = 0.0;
    { // this workgroup's share of zeroing the output (the row reduction adds into it)
        const i64 per = (vlen + nwg - 1) / nwg, z0 = per * wg, z1 = min(vlen, z0 + per);
        for (i64 k = z0 + tid; k < z1; k += EB_THREADS) vectC[k] = 0.0;
    }
    const unsigned short *cu_tab = comm16 + ((i64)bu << EB_UBITS);
    unsigned cus[EB_PER];
#pragma unroll
    for (int k = 0; k < EB_PER; k++) cus[k] = cu_tab[packed[k] >> EB_VBITS]; // ascending addresses across a wave
    __syncthreads();
    if (stop == 1) { if (cus[0] == 0x12345u && (!WEIGHTED || wv[0] == 1.5)) keys[0] = 1; return; }
    // sweep 1: communities of every edge of the chunk; intra-community mass and per-row counts
#pragma unroll
    for (int k = 0; k < EB_PER; k++) {
        unsigned cu = cus[k], cv = sv[packed[k] & (VBLOCK - 1)];
        if (!DIRECTED && cu > cv) { const unsigned t = cu; cu = cv; cv = t; }
        packed[k] = (cu << 16) | cv;
    }
    if (stop == 2) { unsigned x = 0; for (int k = 0; k < EB_PER; k++) x ^= packed[k]; if (x == 0x12345u) keys[0] = 1; return; }
#pragma unroll
    for (int k = 0; k < EB_PER; k++) {
        const int e = tid + k * EB_THREADS;
        const unsigned cu = packed[k] >> 16, cv = packed[k] & 0xFFFFu;
        const bool intra = cu == cv;
        if (e >= len)
            packed[k] = EB_NONE;
        else if (WEIGHTED && intra) {
            unsafeAtomicAdd(&dsum[cu], wv[k]);
            packed[k] = EB_NONE;
        } else { // one 32-bit LDS add per edge: a row count, or (unit weights) an intra-community count
            atomicAdd(&cnt[cu + (intra ? Cpad : 0)], 1u);
            if (intra) packed[k] = EB_NONE;
        }
    }
    __syncthreads(); // all look-ups are done: sv becomes the staging area
    if (stop == 3) { if (cnt[tid % Cpad] == 0x12345u) keys[0] = 1; return; }
    // row offsets inside the chunk's own piece of the key array, keys[start ..) (a thread owns EPT consecutive rows)
    unsigned loc[EPT], mine = 0;
#pragma unroll
    for (int q = 0; q < EPT; q++) {
        const int r = tid * EPT + q;
        loc[q] = r < Cpad ? cnt[r] : 0u;
        mine += loc[q];
    }
    unsigned off = block_exclusive_scan<EB_THREADS>(mine, wave_sums, &misc[0]);
    unsigned short *ro = runoff + (i64)wg * (C + 1);
#pragma unroll
    for (int q = 0; q < EPT; q++) {
        const int r = tid * EPT + q;
        if (r < Cpad) cnt[r] = off;
        if (r < C) ro[r] = (unsigned short)off;
        off += loc[q];
    }
    __syncthreads();
    const unsigned total = misc[0];
    if (tid == 0) ro[C] = (unsigned short)total;
    if (stop == 4) return;
    // sweep 2: the columns of the off-diagonal edges, grouped by row, staged in LDS (weights go straight to memory)
#pragma unroll
    for (int k = 0; k < EB_PER; k++) {
        if (packed[k] != EB_NONE) {
            const unsigned pos = atomicAdd(&cnt[packed[k] >> 16], 1u);
            stage[pos] = (unsigned short)(packed[k] & 0xFFFFu);
            if (WEIGHTED) wkeys[start + pos] = wv[k];
        }
    }
    if (stop == 5) { __syncthreads(); if (stage[tid] == 0xFFFF) keys[0] = 1; return; }
    for (int k = tid; k < C; k += EB_THREADS) diag_out[(i64)wg * C + k] = WEIGHTED ? dsum[k] : (double)cnt[Cpad + k];
    __syncthreads();
    for (unsigned k = tid; k < total; k += EB_THREADS) keys[start + k] = stage[k]; // coalesced
}

// ---- pass 2 ------------------------------------------------------------------------------------------------------------
// grid (C rows, batches of EB_RBATCH chunks).  Thread t fetches where chunk batch * EB_RBATCH + t keeps its keys of the row;
// then 8 lanes share a chunk and read its (contiguous) keys as aligned 16-byte windows of 8 keys, all windows of all chunks
// in flight together.  Keys are counted (unit weights) / summed (weighted) into the workgroup's LDS copy of the row, which is
// added into vect_C with contiguous atomics (exact for unit weights).
template <bool WEIGHTED, bool DIRECTED>
__global__ __launch_bounds__(EB_RBATCH) void edge_row_reduce_kernel(const unsigned short *__restrict__ keys,
                                                                    const double *__restrict__ wkeys,
                                                                    const unsigned short *__restrict__ runoff,
                                                                    const i32 *__restrict__ chunks, int chunk0,
                                                                    const double *__restrict__ diag, int nwg, int C,
                                                                    double *__restrict__ vectC) {
    static_assert(EB_RBATCH == 256, "8 steps of 32 eight-lane groups");
    __shared__ double accd[WEIGHTED ? EB_MAXC : 1];
    __shared__ unsigned accu[WEIGHTED ? 1 : EB_MAXC];
    __shared__ double red[4];
    __shared__ unsigned saddr[EB_RBATCH], slen[EB_RBATCH];
    const int r = blockIdx.x, tid = threadIdx.x, wg = blockIdx.y * EB_RBATCH + tid;
    for (int k = tid; k < C; k += EB_RBATCH) {
        if (WEIGHTED) accd[k] = 0.0; else accu[k] = 0u;
    }
    unsigned len = 0, a = 0;
    double dg = 0.0;
    if (wg < nwg) {
        const unsigned short *ro = runoff + (i64)wg * (C + 1) + r;
        const unsigned b = ro[0], e = ro[1];
        a = (unsigned)chunks[4 * (i64)(chunk0 + wg) + 2] + b;
        len = e - b;
        dg = diag[(i64)wg * C + r];
    }
    saddr[tid] = a;
    slen[tid] = len;
    __syncthreads();
    const int g = tid >> 3, l8 = tid & 7;
    unsigned A[8], L[8], maxwin = 0; // window = 8 keys = one aligned 16-byte load
#pragma unroll
    for (int st = 0; st < 8; st++) {
        A[st] = saddr[st * 32 + g];
        L[st] = slen[st * 32 + g];
        maxwin = max(maxwin, L[st] ? ((A[st] & 7u) + L[st] + 7u) >> 3 : 0u);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxwin = max(maxwin, (unsigned)__shfl_xor((int)maxwin, off)); // wave-uniform
    for (unsigned w0 = 0; w0 < maxwin; w0 += 8) { // one trip unless a run is longer than ~60 keys
        uint4 K[8];
#pragma unroll
        for (int st = 0; st < 8; st++) {
            const unsigned nwin = L[st] ? ((A[st] & 7u) + L[st] + 7u) >> 3 : 0u, w = w0 + l8;
            K[st] = *(const uint4 *)(keys + ((A[st] & ~7u) + 8u * (w < nwin ? w : 0u))); // the array has a readable tail
        }
#pragma unroll
        for (int st = 0; st < 8; st++) {
            const unsigned p0 = (A[st] & ~7u) + 8u * (w0 + l8), lo = A[st], hi = A[st] + L[st];
            const unsigned kw[4] = {K[st].x, K[st].y, K[st].z, K[st].w};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned p = p0 + i, key = (kw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                if (p >= lo && p < hi) {
                    if (WEIGHTED) unsafeAtomicAdd(&accd[key], wkeys[p]);
                    else atomicAdd(&accu[key], 1u);
                }
            }
        }
    }
    // diagonal partials of the batch: in-wave butterfly, then the four wave sums in order (a fixed tree)
    double dsum = dg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
    if ((tid & 63) == 0) red[tid >> 6] = dsum;
    __syncthreads(); // also: every LDS add has been issued
    const double dtot = ((red[0] + red[1]) + red[2]) + red[3];
    const i64 rowbase = DIRECTED ? (i64)r * C : ((i64)C * r - (i64)r * (r - 1) / 2 - r); // + cv
    for (int cv = tid; cv < C; cv += EB_RBATCH) {
        const double v = cv == r ? dtot : (WEIGHTED ? accd[cv] : (double)accu[cv]);
        if (v != 0.0 && (DIRECTED || cv >= r)) unsafeAtomicAdd(&vectC[rowbase + cv], v);
    }
}

// vect_C of chunks [c0, c1) of the blocked resident edge list (all of them: c0 = 0, c1 = be_nchunks).
void k_edge_scatter_blocked(cge_ctx *c, i64 c0, i64 c1, i64 C, int directed, double *vectC) {
    const i64 nwg = c1 - c0;
    hipStream_t st = c->stream;
    const i64 vlen = directed ? C * C : packed_len(C);
    if (nwg <= 0) { // an empty shard still owes zeros
        HIP_CHECK(hipMemsetAsync(vectC, 0, sizeof(double) * vlen, st));
        return;
    }
    c->be_runoff.ensure((size_t)nwg * (C + 1));
    c->be_diag.ensure((size_t)nwg * C);
    ScopedKernelTimer t(c, "edge_scatter");
    const int Cpad = (int)((C + 63) / 64 * 64);
    const bool wt = !c->unit_weights;
    const size_t lds = (size_t)2 * (1 << EB_VBITS) + (wt ? (size_t)8 * Cpad : 0) + (size_t)8 * Cpad + sizeof(unsigned) * 18;
    const dim3 gridB((unsigned)C, (unsigned)((nwg + EB_RBATCH - 1) / EB_RBATCH));
    const int stop = 0; // (the kernel's timing diagnostics)
#define EB_GO(W, D)                                                                                                        \
    do {                                                                                                                   \
        EB_GO2(16, W, D);                                                                                             \
    } while (0)
#define EB_GO2(P, W, D)                                                                                                    \
    do {                                                                                                                   \
        auto kern = edge_pass_kernel<P, W, D>;                                                                              \
        cge_allow_lds((const void *)kern, 160 * 1024); \
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(EB_THREADS), lds, st, c->be_edge.p, c->be_w.p, c->be_chunk.p, (int)c0, \
                           c->comm16.p, (int)C, Cpad, c->be_keys.p, c->be_wkeys.p, c->be_runoff.p, c->be_diag.p, vectC, vlen,  \
                           stop);                                                                                          \
        hipLaunchKernelGGL((edge_row_reduce_kernel<W, D>), gridB, dim3(EB_RBATCH), 0, st, c->be_keys.p, c->be_wkeys.p,       \
                           c->be_runoff.p, c->be_chunk.p, (int)c0, c->be_diag.p, (int)nwg, (int)C, vectC);                  \
    } while (0)
    if (wt && directed) EB_GO(true, true);
    else if (wt) EB_GO(true, false);
    else if (directed) EB_GO(false, true);
    else EB_GO(false, false);
#undef EB_GO
#undef EB_GO2
}


// ======================================================================================================================
// The N x N landmark-pair matrix of landmarks() (src/landmarks.jl:433-451: wedges[min(l_u,l_v), max] += w, directed
// wedges[l_u, l_v] += w) in the same blocked two-pass form.  What changes against vect_C: a "row" of the output is N
// doubles wide (N = 4000 ... 12000), so pass 2 keeps a TILE of R consecutive rows in LDS (R * N counters, 64 KB), one
// workgroup per tile over ALL chunks -- the tile is then written out whole with plain coalesced stores: no memset of the
// N x N output, no atomic ever reaches memory, and the count of positive entries (the length of the landmark edge list,
// :454-461) falls out of the same sweep.  Pass 1 groups a chunk's keys by tile; the 16-bit key is (row inside the tile,
// column).  The landmark table is uint16 (N <= 65535), built from v2l after runsplit.
// ======================================================================================================================
__global__ void wedge_table_kernel(const i32 *__restrict__ v2l, i64 n, i64 npad, unsigned short *__restrict__ t16) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += stride) t16[i] = i < n ? (unsigned short)v2l[i] : 0;
}

// LDS: sv (32768 uint16 landmarks of the target block; reused as the key staging area) | cnt[ntpad] u32 | wave sums, total
template <int EB_PER, bool WEIGHTED, bool DIRECTED>
__global__ __launch_bounds__(EB_THREADS) void wedge_pass_kernel(const unsigned *__restrict__ bedge, const double *__restrict__ bw,
                                                                const i32 *__restrict__ chunks, int chunk0,
                                                                const unsigned short *__restrict__ tab16, int ntile, int ntpad,
                                                                int rshift, int colbits, unsigned short *__restrict__ keys,
                                                                double *__restrict__ wkeys, unsigned short *__restrict__ runoff) {
    constexpr int VBLOCK = 1 << EB_VBITS;
    static_assert(VBLOCK >= EB_THREADS * EB_PER, "the key staging area reuses the slice");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short *sv = (unsigned short *)lds;
    unsigned short *stage = sv;
    unsigned *cnt = (unsigned *)(lds + 2 * VBLOCK);
    unsigned *wave_sums = cnt + ntpad;
    unsigned *misc = wave_sums + 16;
    const int tid = threadIdx.x, wg = blockIdx.x;
    const i32 *ch = chunks + 4 * (i64)(chunk0 + wg);
    const int bu = ch[0], bv = ch[1], start = ch[2], len = ch[3];
    unsigned packed[EB_PER];
    double wv[WEIGHTED ? EB_PER : 1];
#pragma unroll
    for (int k = 0; k < EB_PER; k++) {
        const int e = tid + k * EB_THREADS;
        const int ec = e < len ? e : len - 1;
        packed[k] = bedge[start + ec];
        if (WEIGHTED) wv[k] = bw[start + ec];
    }
    {
        const uint4 *gv = (const uint4 *)(tab16 + (i64)bv * VBLOCK);
        uint4 *lv = (uint4 *)sv;
        for (int i = tid; i < VBLOCK / 8; i += EB_THREADS) lv[i] = gv[i];
    }
    for (int k = tid; k < ntpad; k += EB_THREADS) cnt[k] = 0u;
    const unsigned short *lu_tab = tab16 + ((i64)bu << EB_UBITS);
    unsigned lus[EB_PER];
#pragma unroll
    for (int k = 0; k < EB_PER; k++) lus[k] = lu_tab[packed[k] >> EB_VBITS]; // ascending addresses across a wave
    __syncthreads();
    const unsigned rmask = (1u << rshift) - 1u;
#pragma unroll
    for (int k = 0; k < EB_PER; k++) { // (tile << 16) | key, key = (row inside the tile) << colbits | column
        const int e = tid + k * EB_THREADS;
        unsigned lu = lus[k], lv = sv[packed[k] & (VBLOCK - 1)];
        if (!DIRECTED && lu > lv) { const unsigned t = lu; lu = lv; lv = t; }
        if (e < len) {
            packed[k] = ((lu >> rshift) << 16) | ((lu & rmask) << colbits) | lv;
            atomicAdd(&cnt[lu >> rshift], 1u);
        } else
            packed[k] = EB_NONE;
    }
    __syncthreads(); // all look-ups are done: sv becomes the staging area
    // tile offsets inside the chunk's own piece of the key array (a thread owns `ept` consecutive tiles)
    const int ept = ntpad / EB_THREADS;
    unsigned mine = 0;
    for (int q = 0; q < ept; q++) mine += cnt[tid * ept + q];
    unsigned off = block_exclusive_scan<EB_THREADS>(mine, wave_sums, &misc[0]);
    unsigned short *ro = runoff + (i64)wg * (ntile + 1);
    for (int q = 0; q < ept; q++) {
        const int r = tid * ept + q;
        const unsigned v = cnt[r];
        cnt[r] = off;
        if (r < ntile) ro[r] = (unsigned short)off;
        off += v;
    }
    __syncthreads();
    const unsigned total = misc[0];
    if (tid == 0) ro[ntile] = (unsigned short)total;
#pragma unroll
    for (int k = 0; k < EB_PER; k++) {
        if (packed[k] != EB_NONE) {
            const unsigned pos = atomicAdd(&cnt[packed[k] >> 16], 1u);
            stage[pos] = (unsigned short)(packed[k] & 0xFFFFu);
            if (WEIGHTED) wkeys[start + pos] = wv[k];
        }
    }
    __syncthreads();
    for (unsigned k = tid; k < total; k += EB_THREADS) keys[start + k] = stage[k]; // coalesced
}

// One workgroup per tile of R = 1 << rshift rows.  Thread t fetches where chunk (base + t) keeps its keys of the tile; 8 lanes
// then share a chunk and read its (contiguous) keys as aligned 16-byte windows, all windows of 1024 chunks in flight; the
// keys are counted (unit weights) / summed (weighted) into the LDS tile, which is finally written out whole.
#define WT_THREADS 1024
template <bool WEIGHTED>
__global__ __launch_bounds__(WT_THREADS) void wedge_tile_kernel(const unsigned short *__restrict__ keys, const double *__restrict__ wkeys,
                                                                const unsigned short *__restrict__ runoff,
                                                                const i32 *__restrict__ chunks, int chunk0, int nwg, int ntile, int N,
                                                                int rshift, int colbits, double *__restrict__ wedges,
                                                                unsigned long long *__restrict__ positive) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int R = 1 << rshift, tid = threadIdx.x, tile = blockIdx.x;
    double *accd = (double *)lds;
    unsigned *accu = (unsigned *)lds;
    unsigned *saddr = (unsigned *)(lds + (size_t)R * N * (WEIGHTED ? 8 : 4)), *slen = saddr + WT_THREADS;
    unsigned *red = slen + WT_THREADS;
    for (int k = tid; k < R * N; k += WT_THREADS) {
        if (WEIGHTED) accd[k] = 0.0; else accu[k] = 0u;
    }
    const unsigned cmask = (1u << colbits) - 1u;
    const int g = tid >> 3, l8 = tid & 7;
    for (int base = 0; base < nwg; base += WT_THREADS) {
        const int wg = base + tid;
        unsigned len = 0, a = 0;
        if (wg < nwg) {
            const unsigned short *ro = runoff + (i64)wg * (ntile + 1) + tile;
            const unsigned b = ro[0], e = ro[1];
            a = (unsigned)chunks[4 * (i64)(chunk0 + wg) + 2] + b;
            len = e - b;
        }
        __syncthreads(); // the previous batch's look-ups are done (and, first trip, the tile is zeroed)
        saddr[tid] = a;
        slen[tid] = len;
        __syncthreads();
        unsigned A[8], L[8], maxwin = 0; // window = 8 keys = one aligned 16-byte load
#pragma unroll
        for (int st = 0; st < 8; st++) {
            A[st] = saddr[st * 128 + g];
            L[st] = slen[st * 128 + g];
            maxwin = max(maxwin, L[st] ? ((A[st] & 7u) + L[st] + 7u) >> 3 : 0u);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxwin = max(maxwin, (unsigned)__shfl_xor((int)maxwin, off)); // wave-uniform
        for (unsigned w0 = 0; w0 < maxwin; w0 += 8) {
            uint4 K[8];
#pragma unroll
            for (int st = 0; st < 8; st++) {
                const unsigned nwin = L[st] ? ((A[st] & 7u) + L[st] + 7u) >> 3 : 0u, w = w0 + l8;
                K[st] = *(const uint4 *)(keys + ((A[st] & ~7u) + 8u * (w < nwin ? w : 0u))); // the array has a readable tail
            }
#pragma unroll
            for (int st = 0; st < 8; st++) {
                const unsigned p0 = (A[st] & ~7u) + 8u * (w0 + l8), lo = A[st], hi = A[st] + L[st];
                const unsigned kw[4] = {K[st].x, K[st].y, K[st].z, K[st].w};
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const unsigned p = p0 + i, key = (kw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                    if (p >= lo && p < hi) {
                        const unsigned cell = (key >> colbits) * (unsigned)N + (key & cmask);
                        if (WEIGHTED) unsafeAtomicAdd(&accd[cell], wkeys[p]);
                        else atomicAdd(&accu[cell], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    // the tile, whole: rows [tile * R, ...) of the N x N output; entries > 0 are the landmark edges (src/landmarks.jl:461)
    const i64 row0 = (i64)tile * R;
    const int rows = min(R, N - (int)row0);
    unsigned npos = 0;
    for (int k = tid; k < rows * N; k += WT_THREADS) {
        const double v = WEIGHTED ? accd[k] : (double)accu[k];
        wedges[row0 * N + k] = v;
        npos += v > 0.0 ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) npos += (unsigned)__shfl_xor((int)npos, off);
    if ((tid & 63) == 0) red[tid >> 6] = npos;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < WT_THREADS / 64; w++) t += red[w];
        if (t) atomicAdd(positive, t);
    }
}

// geometry of the tiled form for N landmarks; false: it does not apply (the caller uses the gather + atomics kernel)
static bool wedge_geometry(const cge_ctx *c, i64 N, i64 nchunks, int *rshift, int *colbits, i64 *ntile, size_t *lds2) {
    if (N < 1 || N > 65535 || nchunks <= 0) return false;
    const size_t elt = c->unit_weights ? 4 : 8, fixed = sizeof(unsigned) * (2 * WT_THREADS + 32);
    int cb = 1;
    while (((i64)1 << cb) < N) cb++;
    int rs = 0;
    while (rs + 1 + cb <= 16 && ((size_t)2 << rs) * N * elt <= (size_t)64 * 1024) rs++; // 64 KB tiles: two workgroups per CU
    if (((size_t)1 << rs) * N * elt + fixed > (size_t)150 * 1024) return false;
    const i64 nt = (N + ((i64)1 << rs) - 1) >> rs;
    if (nchunks * (nt + 1) > ((i64)1 << 27)) return false; // the per-chunk tile offsets would outgrow what pass 2 can gather
    *rshift = rs; *colbits = cb; *ntile = nt;
    *lds2 = ((size_t)1 << rs) * N * elt + fixed;
    return true;
}

// wedges (N x N, row-major [a * N + b]) and *positive (device counter) from chunks [c0, c1) of the blocked edge list;
// returns false when the tiled form does not apply.  v2l: 0-based landmark of every vertex (device).
bool k_wedge_scatter_blocked(cge_ctx *c, const i32 *v2l, i64 N, i64 c0, i64 c1, int directed, double *wedges, i64 *positive,
                             const char *timer) {
    const i64 nwg = c1 - c0;
    int rshift, colbits;
    i64 ntile;
    size_t lds2;
    if (!c->blocked_ready || !wedge_geometry(c, N, std::max<i64>(nwg, 1), &rshift, &colbits, &ntile, &lds2)) return false;
    hipStream_t st = c->stream;
    HIP_CHECK(hipMemsetAsync(positive, 0, sizeof(i64), st));
    if (nwg <= 0) { // an empty shard still owes zeros
        HIP_CHECK(hipMemsetAsync(wedges, 0, sizeof(double) * N * N, st));
        return true;
    }
    const i64 npad = (c->n + CGE_COMM16_PAD - 1) / CGE_COMM16_PAD * CGE_COMM16_PAD;
    c->v2l16.ensure(npad);
    c->be_keys.ensure(c->m + 64);
    if (!c->unit_weights) c->be_wkeys.ensure(c->m + 64);
    c->be_runoff.ensure((size_t)nwg * (ntile + 1));
    ScopedKernelTimer t(c, timer);
    hipLaunchKernelGGL(wedge_table_kernel, dim3(grid_for(npad, 256)), dim3(256), 0, st, v2l, c->n, npad, c->v2l16.p);
    const int ntpad = (int)((ntile + EB_THREADS - 1) / EB_THREADS * EB_THREADS);
    const size_t lds1 = (size_t)2 * (1 << EB_VBITS) + sizeof(unsigned) * ((size_t)ntpad + 18);
    const bool wt = !c->unit_weights;
#define WG_GO(W, D)                                                                                                        \
    do {                                                                                                                   \
        auto k1 = wedge_pass_kernel<16, W, D>;                                                                              \
        auto k2 = wedge_tile_kernel<W>;                                                                                     \
        cge_allow_lds((const void *)k1, 160 * 1024);                                                                        \
        cge_allow_lds((const void *)k2, 160 * 1024);                                                                        \
        hipLaunchKernelGGL(k1, dim3((unsigned)nwg), dim3(EB_THREADS), lds1, st, c->be_edge.p, c->be_w.p, c->be_chunk.p, (int)c0, \
                           c->v2l16.p, (int)ntile, ntpad, rshift, colbits, c->be_keys.p, c->be_wkeys.p, c->be_runoff.p);    \
        hipLaunchKernelGGL(k2, dim3((unsigned)ntile), dim3(WT_THREADS), lds2, st, c->be_keys.p, c->be_wkeys.p, c->be_runoff.p, \
                           c->be_chunk.p, (int)c0, (int)nwg, (int)ntile, (int)N, rshift, colbits, wedges,                   \
                           (unsigned long long *)positive);                                                                \
    } while (0)
    if (wt && directed) WG_GO(true, true);
    else if (wt) WG_GO(true, false);
    else if (directed) WG_GO(false, true);
    else WG_GO(false, false);
#undef WG_GO
    return true;
}

// can the blocked copy of the resident edge list exist at all (sort key of 32 bits: tile, source inside the tile)?
bool k_blocked_edges_possible(const cge_ctx *c) {
    return c->m > 0 && c->m < (1LL << 31) &&
           ((c->n + (1 << EB_UBITS) - 1) >> EB_UBITS) * ((c->n + (1 << EB_VBITS) - 1) >> EB_VBITS) <= 32768; // n <= ~1.4e7
}
// dense C x C (row-major, a <= b when undirected) -> the packed upper triangle of vect_C (src/auxilary.jl:57-59)
__global__ void pack_upper_kernel(const double *__restrict__ dense, i64 C, double *__restrict__ packed) {
    const i64 total = C * C, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 a = e / C, b = e - a * C;
        if (b >= a) packed[C * a - a * (a - 1) / 2 + (b - a)] = dense[e];
    }
}
void k_pack_upper(cge_ctx *c, const double *dense, i64 C, double *packed) {
    hipLaunchKernelGGL(pack_upper_kernel, dim3(grid_for(C * C, 256)), dim3(256), 0, c->stream, dense, C, packed);
}

bool k_edge_scatter_blocked_applies(const cge_ctx *c, i64 C) {
    return c->comm16.p && C >= 1 && C <= EB_MAXC && c->m < (1LL << 31) &&
           ((c->n + (1 << EB_UBITS) - 1) >> EB_UBITS) * ((c->n + (1 << EB_VBITS) - 1) >> EB_VBITS) <= 32768; // n <= ~1.4e7
}
