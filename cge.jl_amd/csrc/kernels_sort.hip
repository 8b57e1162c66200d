// kernels_sort.hip -- per-group ascending sort of the projections z on the device (rss rule on sorted
// order, src/landmarks.jl:155-210).  rocPRIM's segmented radix sort is stable, so equal z keep their
// original (ascending index) order -- the same order a stable sortperm gives.
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

// rocPRIM hands any sort of <= 1 Mi items to its merge sort, which compares whole keys (begin/end bit ignored) in ~log2(n/1024)
// small launches.  For the 64-bit z keys that is the faster choice (8 onesweep passes otherwise); the few-bit key sorts of
// this path (task ids, tile keys, community ids) are one or two onesweep passes instead: headline landmarks phase 17.4 ->
// 16.4 ms.
using RadixOnly = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;
static int sort_radix_mask() { return 1; } // bit 0: the few-bit sorts by onesweep, bit 1: the z sort too
template <class K, class V>
static hipError_t radix_pairs(void *tmp, size_t &bytes, const K *ki, K *ko, const V *vi, V *vo, size_t n, int b0, int b1,
                              hipStream_t st) {
    if (sort_radix_mask() & (sizeof(K) > 4 ? 2 : 1))
        return rocprim::radix_sort_pairs<RadixOnly>(tmp, bytes, ki, ko, vi, vo, n, b0, b1, st);
    return rocprim::radix_sort_pairs(tmp, bytes, ki, ko, vi, vo, n, b0, b1, st);
}
__global__ void iota_local_kernel(const i32 *__restrict__ row_task, const i32 *__restrict__ task_row_off, i64 R,
                                  i32 *__restrict__ idx) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < R) idx[j] = (i32)(j - task_row_off[row_task[j]]);
}
// srows[j] = vertex at sorted rank j of its task
__global__ void sorted_gather_kernel(const i32 *__restrict__ rows, const i32 *__restrict__ row_task,
                                     const i32 *__restrict__ task_row_off, const i32 *__restrict__ perm, i64 R,
                                     i32 *__restrict__ srows) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < R) srows[j] = rows[task_row_off[row_task[j]] + perm[j]];
}
// status: 0 = sorted-order path applies, 1 = generic path (NaN, or a tie at the maximum: the arg-max of the
// reference is then not the last rank), 2 = homogeneous (argmin == argmax, src/landmarks.jl:165-167)
__global__ void sort_status_kernel(const double *__restrict__ zs, const i32 *__restrict__ task_row_off, i64 T,
                                   i32 *__restrict__ status) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    const double a = zs[o], b = zs[o + k - 1], b2 = zs[o + k - 2];
    int st = 0;
    if (a != a || b != b) st = 1; // radix order puts NaNs at the two ends
    else if (a == b) st = 2;
    else if (b == b2) st = 1;
    status[t] = st;
}

// helpers of the two-pass form
__global__ void iota_kernel(i32 *__restrict__ x, i64 n);
__global__ void gather_task_keys_kernel(const i32 *__restrict__ pos, const i32 *__restrict__ row_task, i64 R,
                                        unsigned *__restrict__ keys) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < R) keys[j] = (unsigned)row_task[pos[j]];
}
__global__ void finish_two_pass_kernel(const i32 *__restrict__ pos, const double *__restrict__ z,
                                       const i32 *__restrict__ row_task, const i32 *__restrict__ task_row_off, i64 R,
                                       double *__restrict__ zs, i32 *__restrict__ perm) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R) return;
    const i32 p = pos[j];
    zs[j] = z[p];
    perm[j] = p - task_row_off[row_task[p]];
}
// Per-group sort of (z, local index) pairs without a device-wide sort.  A group is cut into pieces of SEGSORT_CAP rows; one
// workgroup sorts a piece in LDS by a bitonic network on the radix order of the doubles (sign-flipped bit patterns, -0 taken
// as +0, NaNs at the two ends -- the order of rocPRIM's radix sort, which the other batches use), ties by index, i.e. the
// permutation of a stable sort.  A group of one piece is finished there; the pieces of a longer group are merged by RANK: every element
// counts, by a binary search in each of the other sorted pieces of its group, how many elements precede it in the (key,
// index) order and goes straight to its final place.  For the batches of long groups this replaces two device-wide sorts of
// all rows (a 64-bit merge sort in ~20 launches, then a radix sort by task: 0.35 ms per batch) by two launches.
#define SEGSORT_CAP 4096
#define SEGSORT_MAXPIECES 8
__device__ __forceinline__ unsigned long long z_sort_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    if (b == 0x8000000000000000ULL) b = 0ULL; // -0.0 sorts as +0.0 (rocPRIM's radix sort and the oracle's `<` agree on that)
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
#define SEGSORT_T 512
__global__ __launch_bounds__(SEGSORT_T) void segment_piece_sort_kernel(const double *__restrict__ z, const i32 *__restrict__ task_row_off,
                                                                 int cap, double *__restrict__ zs, i32 *__restrict__ perm,
                                                                 unsigned long long *__restrict__ tkey, i32 *__restrict__ tidx) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skey[]; // [cap] keys, then [cap] indices
    int *sidx = reinterpret_cast<int *>(skey + cap);
    const i64 t = blockIdx.x, o = task_row_off[t];
    const int len = (int)(task_row_off[t + 1] - o), piece = blockIdx.y, p0 = piece * SEGSORT_CAP, tid = threadIdx.x;
    if (p0 >= len) return; // (uniform) the group has fewer pieces
    const int k = min(SEGSORT_CAP, len - p0);
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    for (int i = tid; i < n2; i += SEGSORT_T) {
        skey[i] = i < k ? z_sort_key(z[o + p0 + i]) : ~0ULL;
        sidx[i] = i < k ? p0 + i : 0x7fffffff;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            // two compare-exchanges per trip: their eight LDS reads are in flight together
            for (int i = tid; i < (n2 >> 1); i += 2 * SEGSORT_T) {
                const int j = i + SEGSORT_T;
                const bool two = j < (n2 >> 1);
                const int lo0 = 2 * i - (i & (stride - 1)), hi0 = lo0 + stride;
                const int lo1 = two ? 2 * j - (j & (stride - 1)) : lo0, hi1 = lo1 + stride;
                const unsigned long long ka0 = skey[lo0], kb0 = skey[hi0], ka1 = skey[lo1], kb1 = skey[hi1];
                const int ia0 = sidx[lo0], ib0 = sidx[hi0], ia1 = sidx[lo1], ib1 = sidx[hi1];
                const bool gt0 = ka0 > kb0 || (ka0 == kb0 && ia0 > ib0), gt1 = ka1 > kb1 || (ka1 == kb1 && ia1 > ib1);
                if (gt0 == ((lo0 & size) == 0)) {
                    skey[lo0] = kb0; skey[hi0] = ka0;
                    sidx[lo0] = ib0; sidx[hi0] = ia0;
                }
                if (two && gt1 == ((lo1 & size) == 0)) {
                    skey[lo1] = kb1; skey[hi1] = ka1;
                    sidx[lo1] = ib1; sidx[hi1] = ia1;
                }
            }
            __syncthreads();
        }
    if (len <= SEGSORT_CAP) { // the whole group: done
        for (int i = tid; i < k; i += SEGSORT_T) {
            zs[o + i] = z[o + sidx[i]]; // (the value itself: the key has lost the sign of a zero)
            perm[o + i] = sidx[i];
        }
    } else {
        for (int i = tid; i < k; i += SEGSORT_T) {
            tkey[o + p0 + i] = skey[i];
            tidx[o + p0 + i] = sidx[i];
        }
    }
}
__global__ __launch_bounds__(256) void segment_rank_merge_kernel(const double *__restrict__ z, const unsigned long long *__restrict__ tkey,
                                                                 const i32 *__restrict__ tidx, const i32 *__restrict__ row_task,
                                                                 const i32 *__restrict__ task_row_off, i64 R, double *__restrict__ zs,
                                                                 i32 *__restrict__ perm) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R) return;
    const i32 t = row_task[j];
    const i64 o = task_row_off[t];
    const int len = (int)(task_row_off[t + 1] - o);
    if (len <= SEGSORT_CAP) return; // finished by the piece sort
    const int q = (int)(j - o), mine = q / SEGSORT_CAP;
    const unsigned long long key = tkey[j];
    const int ix = tidx[j];
    int rank = q - mine * SEGSORT_CAP; // elements of my own piece in front of me
    const int npieces = (len + SEGSORT_CAP - 1) / SEGSORT_CAP;
    for (int p = 0; p < npieces; p++) {
        if (p == mine) continue;
        const i64 b = o + (i64)p * SEGSORT_CAP;
        int lo = 0, hi = min(SEGSORT_CAP, len - p * SEGSORT_CAP);
        while (lo < hi) { // first element of piece p that does not precede (key, ix)
            const int mid = (lo + hi) >> 1;
            const unsigned long long km = tkey[b + mid];
            if (km < key || (km == key && tidx[b + mid] < ix)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    zs[o + rank] = z[o + ix];
    perm[o + rank] = ix;
}
void k_segmented_sort_z(cge_ctx *c, const double *z, const i32 *rows, const i32 *row_task, const i32 *task_row_off,
                        i64 R, i64 T, double *zs, i32 *perm, i32 *srows, i32 *status, i64 max_len) {
    ScopedKernelTimer tm(c, "segmented_sort");
    c->sort_idx.ensure(R);
    const unsigned nb = (unsigned)((R + 255) / 256);
    // the LDS network for every batch whose longest group fits its pieces (round 5: also for batches of many short groups, where
    // rocPRIM's segmented radix sort used to run: 0.57 -> 0.41 ms per step at config 2, same order); beyond that, two device-wide
    // stable sorts (few long groups) or rocPRIM's segmented sort
    const bool long_groups = R / std::max<i64>(T, 1) >= 768;
    if (max_len > 0 && max_len <= (i64)SEGSORT_CAP * SEGSORT_MAXPIECES) {
        int cap = 64;
        while (cap < std::min<i64>(max_len, SEGSORT_CAP)) cap <<= 1;
        const unsigned npieces = (unsigned)((max_len + SEGSORT_CAP - 1) / SEGSORT_CAP);
        cge_allow_lds((const void *)segment_piece_sort_kernel, SEGSORT_CAP * 12);
        if (npieces > 1) { c->sort_k64.ensure(R); c->sort_idx2.ensure(R); }
        hipLaunchKernelGGL(segment_piece_sort_kernel, dim3((unsigned)T, npieces), dim3(SEGSORT_T), (size_t)cap * 12, c->stream, z, task_row_off, cap,
                           zs, perm, c->sort_k64.p, c->sort_idx2.p);
        if (npieces > 1)
            hipLaunchKernelGGL(segment_rank_merge_kernel, dim3(nb), dim3(256), 0, c->stream, z, c->sort_k64.p, c->sort_idx2.p, row_task,
                               task_row_off, R, zs, perm);
    } else if (long_groups) {
        // Few long segments: the segmented sort walks each of them alone through all its radix passes.  Two device-wide
        // stable sorts do the same job at full occupancy: all rows by z, then (stably) by task.
        c->sort_keys32.ensure(R);
        c->sort_idx2.ensure(R);
        c->sort_k32b.ensure(R);
        hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, c->stream, c->sort_idx.p, R);
        size_t bytes = 0;
        HIP_CHECK(radix_pairs(nullptr, bytes, z, zs, c->sort_idx.p, c->sort_idx2.p, (size_t)R, 0, 64, c->stream));
        c->sort_tmp.ensure(bytes);
        HIP_CHECK(radix_pairs(c->sort_tmp.p, bytes, z, zs, c->sort_idx.p, c->sort_idx2.p, (size_t)R, 0, 64,
                                            c->stream));
        hipLaunchKernelGGL(gather_task_keys_kernel, dim3(nb), dim3(256), 0, c->stream, c->sort_idx2.p, row_task, R,
                           (unsigned *)c->sort_keys32.p);
        int bits = 1;
        while (((i64)1 << bits) < T) bits++;
        bytes = 0;
        HIP_CHECK(radix_pairs(nullptr, bytes, (const unsigned *)c->sort_keys32.p, (unsigned *)c->sort_k32b.p,
                                            c->sort_idx2.p, c->sort_idx.p, (size_t)R, 0, bits, c->stream));
        c->sort_tmp.ensure(bytes);
        HIP_CHECK(radix_pairs(c->sort_tmp.p, bytes, (const unsigned *)c->sort_keys32.p,
                                            (unsigned *)c->sort_k32b.p, c->sort_idx2.p, c->sort_idx.p, (size_t)R, 0, bits,
                                            c->stream));
        hipLaunchKernelGGL(finish_two_pass_kernel, dim3(nb), dim3(256), 0, c->stream, c->sort_idx.p, z, row_task,
                           task_row_off, R, zs, perm);
    } else {
        hipLaunchKernelGGL(iota_local_kernel, dim3(nb), dim3(256), 0, c->stream, row_task, task_row_off, R, c->sort_idx.p);
        size_t bytes = 0;
        HIP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, bytes, z, zs, c->sort_idx.p, perm, (unsigned)R, (unsigned)T,
                                                      task_row_off, task_row_off + 1, 0, 64, c->stream));
        c->sort_tmp.ensure(bytes);
        HIP_CHECK(rocprim::segmented_radix_sort_pairs(c->sort_tmp.p, bytes, z, zs, c->sort_idx.p, perm, (unsigned)R,
                                                      (unsigned)T, task_row_off, task_row_off + 1, 0, 64, c->stream));
    }
    hipLaunchKernelGGL(sorted_gather_kernel, dim3(nb), dim3(256), 0, c->stream, rows, row_task, task_row_off, perm, R, srows);
    hipLaunchKernelGGL(sort_status_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, c->stream, zs, task_row_off, T,
                       status);
}

// device-wide radix sort of (uint32 key, int32 value) pairs on the low `bits` bits (stable)
void k_sort_pairs_u32(cge_ctx *c, const unsigned *keys_in, unsigned *keys_out, const i32 *vals_in, i32 *vals_out, i64 n,
                      int bits) {
    size_t bytes = 0;
    HIP_CHECK(radix_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, bits, c->stream));
    c->sort_tmp.ensure(bytes);
    HIP_CHECK(radix_pairs(c->sort_tmp.p, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, bits,
                                        c->stream));
}

// ------------------------------------------------------------------------------------------------
// Member lists on the device.  A group is a range of the member arena (vertex ids, 0-based, in the reference's
// order); a split writes its two children behind each other into a fresh range, low first.
//
// rows of a batch <- the arena ranges of its tasks (one workgroup per 1024-row chunk of a task)
__global__ __launch_bounds__(256) void gather_rows_kernel(const i32 *__restrict__ arena, const i32 *__restrict__ task_off,
                                                          const i32 *__restrict__ task_row_off,
                                                          const i32 *__restrict__ chunk_task,
                                                          const i32 *__restrict__ chunk_beg,
                                                          const i32 *__restrict__ chunk_end, i32 *__restrict__ rows,
                                                          i32 *__restrict__ row_task) {
    const i64 ch = blockIdx.x;
    const i32 t = chunk_task[ch], beg = chunk_beg[ch], end = chunk_end[ch];
    const i64 src = (i64)task_off[t] - task_row_off[t];
    for (i32 p = beg + (i32)threadIdx.x; p < end; p += 256) {
        rows[p] = arena[src + p];
        row_task[p] = t;
    }
}
void k_gather_rows(cge_ctx *c, const i32 *arena, const i32 *task_off, const i32 *task_row_off, const i32 *chunk_task,
                   const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, i32 *rows, i32 *row_task) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_chunks), dim3(256), 0, c->stream, arena, task_off, task_row_off,
                       chunk_task, chunk_beg, chunk_end, rows, row_task);
}

// rss rule: the position of every row in its parent's child lists, as a bucket key.  The reference's order is: the
// low seed (arg-min), then every batch the low side absorbed, in round order, each in ascending original index
// (src/landmarks.jl:163-164, :189, :204-206); then the same for the high side.  Bucket = that position's block;
// a stable sort of the rows (which are in original index order) by bucket gives exactly those two lists.
// One thread per sorted rank; rounds[t][r] = {first rank, end rank, side}, meta[t] = {rounds, rc}.
#define CK_MAXROUNDS 63
__global__ void rss_child_keys_kernel(const i32 *__restrict__ perm, const i32 *__restrict__ row_task,
                                      const i32 *__restrict__ task_row_off, const i32 *__restrict__ meta,
                                      const i32 *__restrict__ rounds, i64 R, unsigned char *__restrict__ keys) {
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= R) return;
    const i64 t = row_task[p];
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o, q = p - o;
    const int nr = meta[2 * t];
    const i32 *rl = rounds + t * 3 * CK_MAXROUNDS;
    int n_low_rounds = 0, mine = -1, before_same = 0;
    for (int r = 0; r < nr; r++) {
        const int side = rl[3 * r + 2];
        n_low_rounds += (side == 1);
        if (mine < 0 && q >= rl[3 * r] && q < rl[3 * r + 1]) {
            mine = side;
            before_same = 0;
            for (int r2 = 0; r2 < r; r2++) before_same += (rl[3 * r2 + 2] == side);
        }
    }
    int b;
    if (q == 0) b = 0;                          // low seed
    else if (q == k - 1) b = 1 + n_low_rounds;  // high seed
    else if (mine == 1) b = 1 + before_same;
    else b = 2 + n_low_rounds + before_same;    // mine == 2 (or a task that failed: overwritten by the host path)
    keys[o + perm[p]] = (unsigned char)b;
}
__global__ void rss_child_counts_kernel(const i32 *__restrict__ meta, const i32 *__restrict__ rounds, i64 T,
                                        i32 *__restrict__ nlow) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int nr = meta[2 * t];
    const i32 *rl = rounds + t * 3 * CK_MAXROUNDS;
    i32 cnt = 1;
    for (int r = 0; r < nr; r++)
        if (rl[3 * r + 2] == 1) cnt += rl[3 * r + 1] - rl[3 * r];
    nlow[t] = cnt;
}
void k_rss_child_keys(cge_ctx *c, const i32 *perm, const i32 *row_task, const i32 *task_row_off, const i32 *meta,
                      const i32 *rounds, i64 R, i64 T, unsigned char *keys, i32 *nlow) {
    hipLaunchKernelGGL(rss_child_keys_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, c->stream, perm, row_task,
                       task_row_off, meta, rounds, R, keys);
    hipLaunchKernelGGL(rss_child_counts_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, c->stream, meta, rounds, T,
                       nlow);
}
// cut rules: side flag 1 / 2 per row -> number of low rows per task (one wave per task)
__global__ __launch_bounds__(64) void side_counts_kernel(const unsigned char *__restrict__ side,
                                                         const i32 *__restrict__ task_row_off, i32 *__restrict__ nlow) {
    const i64 t = blockIdx.x;
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    i32 cnt = 0;
    for (i64 j = threadIdx.x; j < k; j += 64) cnt += (side[o + j] == 1);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (threadIdx.x == 0) nlow[t] = cnt;
}
void k_side_counts(cge_ctx *c, const unsigned char *side, const i32 *task_row_off, i64 T, i32 *nlow) {
    hipLaunchKernelGGL(side_counts_kernel, dim3((unsigned)T), dim3(64), 0, c->stream, side, task_row_off, nlow);
}
// children lists: every task's rows (vertex ids, original order) stably grouped by key, written to `out` (the fresh arena range of
// the batch; task t lands at out + task_row_off[t], low children first).  A key is a small bucket number (rss: 2 + the rounds of
// the task, at most 128; the cut rules: side 1 / 2), so this is a stable multi-way partition, not a sort: per chunk of the batch a
// histogram, per task the buckets' start positions chunk by chunk, per chunk the scatter -- a row's position is its bucket's start
// in the chunk + the rows of the same bucket before it (ranks inside a wave by ballots over the keys present, wave counts through
// LDS).  (Rounds 1-4 called rocPRIM's segmented radix sort here: 0.44 ms per headline step, 1.75 ms at config 3's 30 batches.)
#define CP_MAXB 128
__global__ __launch_bounds__(256) void child_hist_kernel(const unsigned char *__restrict__ keys, const i32 *__restrict__ chunk_beg,
                                                         const i32 *__restrict__ chunk_end, int B, i32 *__restrict__ chist) {
    __shared__ int h[CP_MAXB];
    const int tid = threadIdx.x;
    const i64 ch = blockIdx.x;
    for (int i = tid; i < B; i += 256) h[i] = 0;
    __syncthreads();
    const i32 beg = chunk_beg[ch], end = chunk_end[ch];
    for (i32 p = beg + tid; p < end; p += 256) atomicAdd(&h[keys[p] & (B - 1)], 1);
    __syncthreads();
    for (int i = tid; i < B; i += 256) chist[ch * B + i] = h[i];
}
// in: the chunks' counts per bucket; out (in place): where a chunk's rows of a bucket start in `out`
__global__ __launch_bounds__(CP_MAXB) void child_offsets_kernel(const i32 *__restrict__ task_chunk_off,
                                                                const i32 *__restrict__ task_row_off, int B, i32 *__restrict__ chist) {
    __shared__ int pre[CP_MAXB];
    const int b = threadIdx.x;
    const i64 t = blockIdx.x;
    const i32 c0 = task_chunk_off[t], c1 = task_chunk_off[t + 1];
    int tot = 0;
    if (b < B)
        for (i32 ch = c0; ch < c1; ch++) tot += chist[(i64)ch * B + b];
    pre[b] = b < B ? tot : 0;
    __syncthreads();
    if (b == 0) {
        int s2 = 0;
        for (int i = 0; i < B; i++) {
            const int v = pre[i];
            pre[i] = s2;
            s2 += v;
        }
    }
    __syncthreads();
    if (b < B) {
        int run = task_row_off[t] + pre[b];
        for (i32 ch = c0; ch < c1; ch++) {
            const int v = chist[(i64)ch * B + b];
            chist[(i64)ch * B + b] = run;
            run += v;
        }
    }
}
__global__ __launch_bounds__(256) void child_scatter_kernel(const unsigned char *__restrict__ keys, const i32 *__restrict__ rows,
                                                            const i32 *__restrict__ chunk_beg, const i32 *__restrict__ chunk_end,
                                                            int B, const i32 *__restrict__ cstart, i32 *__restrict__ out) {
    __shared__ int run[CP_MAXB];
    __shared__ int wcnt[4][CP_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 ch = blockIdx.x;
    const i32 beg = chunk_beg[ch], end = chunk_end[ch];
    for (int i = tid; i < B; i += 256) run[i] = cstart[ch * B + i];
    for (i32 tile = beg; tile < end; tile += 256) {
        for (int i = tid; i < 4 * CP_MAXB; i += 256) (&wcnt[0][0])[i] = 0;
        __syncthreads();
        const i32 p = tile + tid;
        const bool valid = p < end;
        const int key = valid ? (int)(keys[p] & (B - 1)) : -1;
        int rank = 0;
        unsigned long long rem = __ballot(valid);
        while (rem) { // one turn per key present in the wave
            const int first = __builtin_ctzll(rem);
            const int k0 = __shfl(key, first);
            const unsigned long long m = __ballot(key == k0);
            if (key == k0) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == first) wcnt[wave][k0] = __popcll(m);
            rem &= ~m;
        }
        __syncthreads();
        if (valid) {
            int base = run[key];
            for (int w2 = 0; w2 < wave; w2++) base += wcnt[w2][key];
            out[base + rank] = rows[p];
        }
        __syncthreads();
        for (int k = tid; k < B; k += 256) run[k] += (wcnt[0][k] + wcnt[1][k]) + (wcnt[2][k] + wcnt[3][k]);
        __syncthreads();
    }
}
void k_sort_children(cge_ctx *c, const unsigned char *keys, const i32 *rows, const i32 *task_row_off, const i32 *chunk_beg,
                     const i32 *chunk_end, const i32 *task_chunk_off, i64 n_chunks, i64 R, i64 T, int key_bits, i32 *out) {
    ScopedKernelTimer tm(c, "children_sort");
    if (R <= 0 || T <= 0 || n_chunks <= 0) return;
    const int B = 1 << key_bits;
    if (B > CP_MAXB) CGE_THROW(CGE_E_ASSERT, "children lists: %d buckets", B);
    c->sort_cnt.ensure((size_t)n_chunks * B);
    hipLaunchKernelGGL(child_hist_kernel, dim3((unsigned)n_chunks), dim3(256), 0, c->stream, keys, chunk_beg, chunk_end, B, c->sort_cnt.p);
    hipLaunchKernelGGL(child_offsets_kernel, dim3((unsigned)T), dim3(CP_MAXB), 0, c->stream, task_chunk_off, task_row_off, B,
                       c->sort_cnt.p);
    hipLaunchKernelGGL(child_scatter_kernel, dim3((unsigned)n_chunks), dim3(256), 0, c->stream, keys, rows, chunk_beg, chunk_end, B,
                       c->sort_cnt.p, out);
}

// The final groups (the heap array, src/landmarks.jl:337-342) -> v2l and the landmark -> members index.
__global__ __launch_bounds__(256) void assign_groups_kernel(const i32 *__restrict__ arena, const i32 *__restrict__ goff,
                                                            const i32 *__restrict__ glen, i32 *__restrict__ v2l) {
    const i64 g = blockIdx.x;
    const i64 o = goff[g], k = glen[g];
    for (i64 q = threadIdx.x; q < k; q += 256) v2l[arena[o + q]] = (i32)g;
}
__global__ void count_unassigned_kernel(const i32 *__restrict__ v2l, i64 n, i32 *__restrict__ cnt) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && v2l[i] < 0) atomicAdd(cnt, 1);
}
__global__ void iota_kernel(i32 *__restrict__ x, i64 n) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = (i32)i;
}
// v2l[i] = group of vertex i (0-based); mem = vertices grouped by landmark, ascending inside a landmark (stable sort
// of 0..n-1 by landmark); returns the number of vertices no group claimed (the reference asserts 0, :343)
i64 k_groups_to_index(cge_ctx *c, const i32 *arena, const i32 *goff, const i32 *glen, i64 N, i64 n, i32 *v2l, i32 *mem) {
    hipStream_t st = c->stream;
    HIP_CHECK(hipMemsetAsync(v2l, 0xFF, sizeof(i32) * n, st));
    hipLaunchKernelGGL(assign_groups_kernel, dim3((unsigned)N), dim3(256), 0, st, arena, goff, glen, v2l);
    c->sort_cnt.ensure(1);
    HIP_CHECK(hipMemsetAsync(c->sort_cnt.p, 0, sizeof(i32), st));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(count_unassigned_kernel, dim3(nb), dim3(256), 0, st, v2l, n, c->sort_cnt.p);
    c->sort_idx.ensure(n);
    c->sort_keys32.ensure(n);
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, st, c->sort_idx.p, n);
    int bits = 1;
    while (((i64)1 << bits) < N) bits++;
    size_t bytes = 0;
    HIP_CHECK(radix_pairs(nullptr, bytes, (const unsigned *)v2l, (unsigned *)c->sort_keys32.p, c->sort_idx.p, mem,
                                        (size_t)n, 0, bits, st));
    c->sort_tmp.ensure(bytes);
    HIP_CHECK(radix_pairs(c->sort_tmp.p, bytes, (const unsigned *)v2l, (unsigned *)c->sort_keys32.p,
                                        c->sort_idx.p, mem, (size_t)n, 0, bits, st));
    i32 bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, c->sort_cnt.p, sizeof(i32), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return bad;
}

// dst[idx[i]] = src[i] + add  /  dst[i] = src[i] + add  (option shard_rows: per-row results into per-vertex tables)
__global__ void scatter_i32_kernel(const i32 *__restrict__ src, const i32 *__restrict__ idx, i64 cnt, i32 add, i32 *__restrict__ dst) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) dst[idx[i]] = src[i] + add;
}
__global__ void add_i32_kernel(const i32 *__restrict__ src, i64 n, i32 add, i32 *__restrict__ dst) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + add;
}
void k_scatter_i32(cge_ctx *c, const i32 *src, const i32 *idx, i64 cnt, i32 add, i32 *dst) {
    if (cnt > 0) hipLaunchKernelGGL(scatter_i32_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, src, idx, cnt, add, dst);
}
void k_add_i32(cge_ctx *c, const i32 *src, i64 n, i32 add, i32 *dst) {
    if (n > 0) hipLaunchKernelGGL(add_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src, n, add, dst);
}

// mean[t][:] <- the d doubles at off[t] of the means arena
__global__ void gather_means_kernel(const double *__restrict__ arena, const i64 *__restrict__ off, i64 d,
                                    double *__restrict__ mean) {
    const i64 t = blockIdx.x;
    const double *src = arena + off[t];
    for (i64 q = threadIdx.x; q < d; q += blockDim.x) mean[t * d + q] = src[q];
}
// dst[seg[3s+1] + j] = src[seg[3s] + j], j < seg[3s+2]: one workgroup per segment (member lists moved between ranges)
__global__ void copy_segments_kernel(const i32 *__restrict__ src, const i64 *__restrict__ seg, i32 *__restrict__ dst) {
    const i64 so = seg[3 * blockIdx.x], dof = seg[3 * blockIdx.x + 1], len = seg[3 * blockIdx.x + 2];
    for (i64 j = threadIdx.x; j < len; j += blockDim.x) dst[dof + j] = src[so + j];
}
void k_copy_segments(cge_ctx *c, const i32 *src, const i64 *seg, i64 nseg, i32 *dst) {
    if (nseg > 0) hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)nseg), dim3(256), 0, c->stream, src, seg, dst);
}
// dst[slot[t] * stride + lead + q] = arena[off[t] + q], q < d (means of groups into strided records)
__global__ void gather_means_slots_kernel(const double *__restrict__ arena, const i64 *__restrict__ off,
                                          const i64 *__restrict__ slot, i64 d, i64 stride, i64 lead, double *__restrict__ dst) {
    const i64 t = blockIdx.x;
    const double *src = arena + off[t];
    double *out = dst + slot[t] * stride + lead;
    for (i64 q = threadIdx.x; q < d; q += blockDim.x) out[q] = src[q];
}
void k_gather_means_slots(cge_ctx *c, const double *arena, const i64 *off, const i64 *slot, i64 T, i64 d, i64 stride, i64 lead,
                          double *dst) {
    if (T > 0)
        hipLaunchKernelGGL(gather_means_slots_kernel, dim3((unsigned)T), dim3(128), 0, c->stream, arena, off, slot, d, stride, lead, dst);
}
void k_gather_means(cge_ctx *c, const double *arena, const i64 *off, i64 T, i64 d, double *mean) {
    hipLaunchKernelGGL(gather_means_kernel, dim3((unsigned)T), dim3(128), 0, c->stream, arena, off, d, mean);
}
