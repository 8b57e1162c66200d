// kernels_scatter.hip -- the per-edge cluster-pair scatter-add  vect_C[bin(c_u, c_v)] += w  of wGCL / wGCL_directed
// (src/divergence.jl:55-63, :337-345) over the RESIDENT edge list, without global atomics and without per-edge gathers
// from a table in memory.
//
// Data layout (built once per resident graph, by the first edge pass after cge_set_graph):
//   * the edge list is kept a second time BLOCKED: vertices in blocks of 32768, edges grouped by (block of u, block of v)
//     and stored as one 32-bit word (u mod 32768) << 16 | (v mod 32768) -- 4 bytes per edge instead of the reference's
//     16 (two Int64), plus the weight (8 bytes) only for a weighted list.  A group is cut into chunks of <= 16384 edges.
//   * the community table is uint16 (C <= 1024 on this path), padded to a multiple of 32768 entries.
// Pass 1 (edge_pass_kernel, one 1024-thread workgroup per chunk): the two 64 KB slices of the community table that the
//   chunk can touch are copied into LDS (coalesced 16-byte loads), so the 2 m community look-ups are LDS reads; intra-
//   community edges (the bulk of a graph with community structure) are summed in LDS per community; the others are
//   counted per row (= smaller community), the workgroup claims a contiguous piece of a key array with ONE atomic, and
//   writes its off-diagonal edges there grouped by row (the column as uint16 + the weight when weighted), with the
//   per-row offsets beside it.
// Pass 2 (edge_row_reduce_kernel, one workgroup per row of the C x C matrix): walks the row's piece of every chunk,
//   accumulates the row in LDS, adds the diagonal partials of all chunks in a fixed order and writes the row with plain
//   stores -- every output bin is written exactly once (no memset, no atomics to memory).
// Unit weights: all sums are exact integers, results are bitwise reproducible.  Weighted lists: the LDS additions of one
// bin are unordered (last-bit differences for non-dyadic weights), as with the atomics this replaces.
#include "common.hpp"

#define EB_BLOCK_BITS 15
#define EB_BLOCK (1 << EB_BLOCK_BITS)   // vertices per block
#define EB_THREADS 1024
#define EB_PER_THREAD 16
#define EB_CHUNK (EB_THREADS * EB_PER_THREAD) // edges per chunk at most
#define EB_MAXC 1024
#define EB_NONE 0xFFFFFFFFu

__global__ void eb_tile_keys_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst, i64 m, unsigned nb,
                                    unsigned *__restrict__ keys, i32 *__restrict__ idx) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        keys[e] = ((unsigned)src[e] >> EB_BLOCK_BITS) * nb + ((unsigned)dst[e] >> EB_BLOCK_BITS);
        idx[e] = (i32)e;
    }
}
// blocked words (and weights) in tile order; first position of every tile that occurs
__global__ void eb_gather_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst, const double *__restrict__ w,
                                 const unsigned *__restrict__ skeys, const i32 *__restrict__ sidx, i64 m,
                                 unsigned *__restrict__ bedge, double *__restrict__ bw, i32 *__restrict__ tile_first) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
        const i32 e = sidx[k];
        bedge[k] = (((unsigned)src[e] & (EB_BLOCK - 1)) << 16) | ((unsigned)dst[e] & (EB_BLOCK - 1));
        if (bw) bw[k] = w[e];
        if (k == 0 || skeys[k] != skeys[k - 1]) tile_first[skeys[k]] = (i32)k;
    }
}

void k_sort_pairs_u32(cge_ctx *c, const unsigned *keys_in, unsigned *keys_out, const i32 *vals_in, i32 *vals_out, i64 n,
                      int bits); // kernels_sort.hip (rocPRIM)

// Build the blocked copy of the resident edge list.  Returns false when this path does not apply (the caller then uses
// the gather + atomics kernel): more than 64 vertex blocks (n > 2 097 152) or no uint16 community table.
bool k_build_blocked_edges(cge_ctx *c) {
    const i64 n = c->n, m = c->m;
    const i64 nb = (n + EB_BLOCK - 1) >> EB_BLOCK_BITS;
    if (nb > 64 || m <= 0 || m >= (1LL << 31)) return false;
    hipStream_t st = c->stream;
    const i64 T = nb * nb;
    DevBuf<unsigned> keys, skeys;
    DevBuf<i32> idx, sidx, tfirst;
    keys.ensure(m); skeys.ensure(m); idx.ensure(m); sidx.ensure(m); tfirst.ensure(T + 1);
    hipLaunchKernelGGL(eb_tile_keys_kernel, dim3(grid_for(m, 256)), dim3(256), 0, st, c->src.p, c->dst.p, m, (unsigned)nb,
                       keys.p, idx.p);
    int bits = 1;
    while (((i64)1 << bits) < T) bits++;
    k_sort_pairs_u32(c, keys.p, skeys.p, idx.p, sidx.p, m, bits);
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
    for (i64 t = 0; t < T; t++) {
        const i64 len = tf[t + 1] - tf[t];
        if (len <= 0) continue;
        const i64 parts = (len + EB_CHUNK - 1) / EB_CHUNK, per = (len + parts - 1) / parts; // equal pieces
        for (i64 s = 0; s < len; s += per) {
            ch.push_back((i32)(t / nb)); ch.push_back((i32)(t % nb));
            ch.push_back((i32)(tf[t] + s)); ch.push_back((i32)std::min(per, len - s));
        }
    }
    c->be_nchunks = (i64)ch.size() / 4;
    c->be_chunk.alloc_exact(ch.size());
    HIP_CHECK(hipMemcpyAsync(c->be_chunk.p, ch.data(), sizeof(i32) * ch.size(), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    c->be_keys.ensure(m);
    if (!c->unit_weights) c->be_wkeys.ensure(m);
    c->be_cursor.ensure(1);
    c->blocked_ready = true;
    return true;
}

// exclusive prefix sum of cnt[0..C) (C <= 1024, one entry per thread) -> returned; total in *tot (LDS word)
__device__ __forceinline__ unsigned block_exclusive_scan_1024(unsigned v, unsigned *wave_sums, unsigned *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        unsigned s = lane < EB_THREADS / 64 ? wave_sums[lane] : 0u, si = s;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const unsigned t = __shfl_up(si, off);
            if (lane >= off) si += t;
        }
        if (lane < EB_THREADS / 64) wave_sums[lane] = si - s; // exclusive wave offsets
        if (lane == EB_THREADS / 64 - 1) *tot = si;
    }
    __syncthreads();
    return wave_sums[wave] + inc - v;
}

template <bool WEIGHTED, bool DIRECTED>
__global__ __launch_bounds__(EB_THREADS) void edge_pass_kernel(const unsigned *__restrict__ bedge,
                                                               const double *__restrict__ bw,
                                                               const i32 *__restrict__ chunks, int chunk0,
                                                               const unsigned short *__restrict__ comm16, int C,
                                                               unsigned *__restrict__ cursor,
                                                               unsigned short *__restrict__ keys,
                                                               double *__restrict__ wkeys,
                                                               unsigned short *__restrict__ runoff,
                                                               unsigned *__restrict__ base_out,
                                                               double *__restrict__ diag_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short *su = (unsigned short *)lds;          // communities of the block of u
    unsigned short *sv = su + EB_BLOCK;                   // ... of the block of v (aliases su on a diagonal tile)
    double *diag = (double *)(lds + 4 * EB_BLOCK);        // [EB_MAXC] intra-community sums
    unsigned *cnt = (unsigned *)(diag + EB_MAXC);         // [EB_MAXC] off-diagonal edges per row, then the row cursors
    unsigned *wave_sums = cnt + EB_MAXC;                  // [16]
    unsigned *misc = wave_sums + 16;                      // [0] total, [1] base
    const int tid = threadIdx.x, wg = blockIdx.x;
    const i32 *ch = chunks + 4 * (i64)(chunk0 + wg);
    const int bu = ch[0], bv = ch[1], start = ch[2], len = ch[3];
    {
        const uint4 *gu = (const uint4 *)(comm16 + (i64)bu * EB_BLOCK);
        uint4 *lu = (uint4 *)su;
#pragma unroll
        for (int i = 0; i < EB_BLOCK / 8 / EB_THREADS; i++) lu[tid + i * EB_THREADS] = gu[tid + i * EB_THREADS];
        if (bv != bu) {
            const uint4 *gv = (const uint4 *)(comm16 + (i64)bv * EB_BLOCK);
            uint4 *lv = (uint4 *)sv;
#pragma unroll
            for (int i = 0; i < EB_BLOCK / 8 / EB_THREADS; i++) lv[tid + i * EB_THREADS] = gv[tid + i * EB_THREADS];
        } else
            sv = su;
    }
    diag[tid] = 0.0;
    cnt[tid] = 0u;
    __syncthreads();
    // sweep 1: communities of every edge of the chunk; intra-community mass and per-row counts
    unsigned packed[EB_PER_THREAD];
    double wv[WEIGHTED ? EB_PER_THREAD : 1];
#pragma unroll
    for (int k = 0; k < EB_PER_THREAD; k++) {
        const int e = tid + k * EB_THREADS;
        packed[k] = EB_NONE;
        if (e < len) {
            const unsigned be = bedge[start + e];
            unsigned cu = su[be >> 16], cv = sv[be & 0xFFFFu];
            if (!DIRECTED && cu > cv) { const unsigned t = cu; cu = cv; cv = t; }
            const double we = WEIGHTED ? bw[start + e] : 1.0;
            if (WEIGHTED) wv[k] = we;
            if (cu == cv)
                unsafeAtomicAdd(&diag[cu], we);
            else {
                atomicAdd(&cnt[cu], 1u);
                packed[k] = (cu << 16) | cv;
            }
        }
    }
    __syncthreads();
    // row offsets inside this workgroup's piece of the key array; one atomic claims the piece
    const unsigned mine = cnt[tid];
    const unsigned off = block_exclusive_scan_1024(mine, wave_sums, &misc[0]);
    if (tid == 0) misc[1] = misc[0] ? atomicAdd(cursor, misc[0]) : 0u;
    cnt[tid] = off;
    unsigned short *ro = runoff + (i64)wg * (C + 1);
    if (tid < C) ro[tid] = (unsigned short)off;
    __syncthreads();
    if (tid == 0) {
        ro[C] = (unsigned short)misc[0];
        base_out[wg] = misc[1];
    }
    const unsigned base = misc[1];
    // sweep 2: the off-diagonal edges, grouped by row
#pragma unroll
    for (int k = 0; k < EB_PER_THREAD; k++) {
        if (packed[k] != EB_NONE) {
            const unsigned pos = base + atomicAdd(&cnt[packed[k] >> 16], 1u);
            keys[pos] = (unsigned short)(packed[k] & 0xFFFFu);
            if (WEIGHTED) wkeys[pos] = wv[k];
        }
    }
    if (tid < C) diag_out[(i64)wg * C + tid] = diag[tid];
}

template <bool WEIGHTED, bool DIRECTED>
__global__ __launch_bounds__(256) void edge_row_reduce_kernel(const unsigned short *__restrict__ keys,
                                                              const double *__restrict__ wkeys,
                                                              const unsigned short *__restrict__ runoff,
                                                              const unsigned *__restrict__ base,
                                                              const double *__restrict__ diag, int nwg, int C,
                                                              double *__restrict__ vectC) {
    __shared__ double acc[EB_MAXC];
    __shared__ double red[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < C; k += 256) acc[k] = 0.0;
    __syncthreads();
    double dsum = 0.0;
    for (int wg = tid; wg < nwg; wg += 256) { // a thread's chunks in ascending order: a fixed order of additions
        const unsigned short *ro = runoff + (i64)wg * (C + 1);
        const unsigned b = ro[r], e = ro[r + 1], bs = base[wg];
        for (unsigned k = b; k < e; k++) unsafeAtomicAdd(&acc[keys[bs + k]], WEIGHTED ? wkeys[bs + k] : 1.0);
        dsum += diag[(i64)wg * C + r];
    }
    red[tid] = dsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double dtot = red[0];
    const i64 rowbase = DIRECTED ? (i64)r * C : ((i64)C * r - (i64)r * (r - 1) / 2 - r); // + cv
    for (int cv = tid; cv < C; cv += 256)
        if (DIRECTED || cv >= r) vectC[rowbase + cv] = cv == r ? dtot : acc[cv];
}

// vect_C of chunks [c0, c1) of the blocked resident edge list (all of them: c0 = 0, c1 = be_nchunks).
void k_edge_scatter_blocked(cge_ctx *c, i64 c0, i64 c1, i64 C, int directed, double *vectC) {
    const i64 nwg = c1 - c0;
    hipStream_t st = c->stream;
    if (nwg <= 0) { // an empty shard still owes zeros
        HIP_CHECK(hipMemsetAsync(vectC, 0, sizeof(double) * (directed ? C * C : packed_len(C)), st));
        return;
    }
    k_edge_scatter_blocked_init();
    c->be_runoff.ensure((size_t)nwg * (C + 1));
    c->be_base.ensure(nwg);
    c->be_diag.ensure((size_t)nwg * C);
    ScopedKernelTimer t(c, "edge_scatter");
    HIP_CHECK(hipMemsetAsync(c->be_cursor.p, 0, sizeof(unsigned), st));
    const size_t lds = 4 * EB_BLOCK + sizeof(double) * EB_MAXC + sizeof(unsigned) * (EB_MAXC + 16 + 2);
    const bool wt = !c->unit_weights;
#define EB_LAUNCH_A(W, D)                                                                                              \
    hipLaunchKernelGGL((edge_pass_kernel<W, D>), dim3((unsigned)nwg), dim3(EB_THREADS), lds, st, c->be_edge.p, c->be_w.p,   \
                       c->be_chunk.p, (int)c0, c->comm16.p, (int)C, c->be_cursor.p, c->be_keys.p, c->be_wkeys.p,         \
                       c->be_runoff.p, c->be_base.p, c->be_diag.p)
#define EB_LAUNCH_B(W, D)                                                                                              \
    hipLaunchKernelGGL((edge_row_reduce_kernel<W, D>), dim3((unsigned)C), dim3(256), 0, st, c->be_keys.p, c->be_wkeys.p,  \
                       c->be_runoff.p, c->be_base.p, c->be_diag.p, (int)nwg, (int)C, vectC)
    if (wt && directed) { EB_LAUNCH_A(true, true); EB_LAUNCH_B(true, true); }
    else if (wt) { EB_LAUNCH_A(true, false); EB_LAUNCH_B(true, false); }
    else if (directed) { EB_LAUNCH_A(false, true); EB_LAUNCH_B(false, true); }
    else { EB_LAUNCH_A(false, false); EB_LAUNCH_B(false, false); }
#undef EB_LAUNCH_A
#undef EB_LAUNCH_B
}

// one-time attribute: the edge pass uses ~145 KB of dynamic LDS
void k_edge_scatter_blocked_init() {
    static bool done = false;
    if (done) return;
    const int lds = 4 * EB_BLOCK + (int)sizeof(double) * EB_MAXC + (int)sizeof(unsigned) * (EB_MAXC + 16 + 2);
    (void)hipFuncSetAttribute((const void *)edge_pass_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void *)edge_pass_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void *)edge_pass_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void *)edge_pass_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    done = true;
}

bool k_edge_scatter_blocked_applies(const cge_ctx *c, i64 C) {
    static const bool off = getenv("CGE_SCATTER_GATHER") != nullptr; // A/B switch: force the gather + atomics kernel
    return !off && c->comm16.p && C >= 1 && C <= EB_MAXC && c->n <= (i64)64 * EB_BLOCK && c->m < (1LL << 31);
}
