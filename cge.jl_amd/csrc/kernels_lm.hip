// kernels_lm.hip -- layout conversion, landmark split primitives, landmark aggregation and the
// per-edge scatter.  gfx950, wave = 64.
//
// Reference loops replaced here:
//   group mean / covariance / projection : matrix_w_mean, y'y, y*v   src/landmarks.jl:71-81, :97-99 (=:160-162,...)
//   landmark aggregation                : src/landmarks.jl:387-430
//   per-edge scatter                    : src/landmarks.jl:433-451, src/divergence.jl:59-63, :337-345
#include "common.hpp"
#include <type_traits>
#include "mfma_tile.hpp"

#define WAVE 64

// ---- in-wave reductions by DPP (no LDS crossbar): used by the eigen-solver and the sequential rule kernels ----------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) { // uniform result (two v_readlane_b32)
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// sum over the 64 lanes of a wave, the same bits in every lane and in every wave that sums the same values:
// butterfly inside each row of 16 (quad swaps, half mirror, mirror), then the four row totals in fixed order
__device__ __forceinline__ double wave_allsum(double v) {
    v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v); // row_half_mirror
    v += dpp_move<0x140>(v); // row_mirror
    return ((lane_value(v, 0) + lane_value(v, 16)) + lane_value(v, 32)) + lane_value(v, 48);
}

// Lane 0's result of `for (off = 32; off > 0; off >>= 1) v += __shfl_down(v, off)` -- the same pairs added in the same order,
// hence the same bits -- without the LDS crossbar (12 ds_bpermute per double and their waits): gfx950's lane swaps for the
// distances 32 and 16 (v_permlane32_swap: lanes 0..31 of the second result hold lanes 32..63; v_permlane16_swap: rows 0 / 2
// of the second result hold rows 1 / 3), DPP row shifts for 8, 4, 2, 1 inside the first row of 16.  Other lanes: garbage.
template <int CTRL>
__device__ __forceinline__ double dpp_shift(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_tree_sum_lane0(double v) {
    {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
        const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v += __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
    }
    {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
        const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v += __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
    }
    v += dpp_shift<0x108>(v); // row_shl:8
    v += dpp_shift<0x104>(v); // row_shl:4
    v += dpp_shift<0x102>(v); // row_shl:2
    v += dpp_shift<0x101>(v); // row_shl:1
    return v;
}
// testing hook: both forms on n_rows x 64 values, one wave per row
__global__ void wave_tree_test_kernel(const double *__restrict__ x, i64 n_rows, double *__restrict__ out_ref, double *__restrict__ out_new) {
    const i64 r = ((i64)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    const double v = x[r * 64 + lane];
    double s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    const double t = wave_tree_sum_lane0(v);
    if (lane == 0) { out_ref[r] = s; out_new[r] = t; }
}
void k_wave_tree_test(cge_ctx *c, const double *x, i64 n_rows, double *out_ref, double *out_new) {
    hipLaunchKernelGGL(wave_tree_test_kernel, dim3((unsigned)((n_rows * WAVE + 255) / 256)), dim3(256), 0, c->stream, x, n_rows,
                       out_ref, out_new);
}

// ------------------------------------------------------------------------------------------------
// column-major (n x d, Julia) -> row-major (node-major).  32x32 tiles through LDS.  The source may be a PIECE of the matrix --
// rows [i0, i0 + rows) of the columns [k0, k0 + cols), column-major with leading dimension `rows` -- as the chunks of the
// embedding's upload arrive (cge_set_embedding transposes every chunk behind its copy: no n x d column-major device buffer
// and no separate pass over it).
__global__ void transpose_kernel(const double *__restrict__ Xcol, double *__restrict__ Xrow, i64 rows, i64 cols, i64 i0, i64 k0, i64 d) {
    __shared__ double tile[32][33];
    i64 ib = (i64)blockIdx.x * 32, kb = (i64)blockIdx.y * 32;
    int tx = threadIdx.x, ty = threadIdx.y; // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        i64 i = ib + tx, k = kb + r;
        if (i < rows && k < cols) tile[r][tx] = Xcol[i + k * rows];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        i64 i = ib + r, k = kb + tx;
        if (i < rows && k < cols) Xrow[(i0 + i) * d + k0 + k] = tile[tx][r];
    }
}
void k_transpose_to_rowmajor(cge_ctx *c, const double *Xcol, double *Xrow, i64 n, i64 d) {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((d + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, c->stream, Xcol, Xrow, n, d, (i64)0, (i64)0, d);
}
void k_transpose_piece(cge_ctx *c, const double *piece, double *Xrow, i64 rows, i64 cols, i64 i0, i64 k0, i64 d) {
    dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, c->stream, piece, Xrow, rows, cols, i0, k0, d);
}

// ------------------------------------------------------------------------------------------------
// Global mean (any centre is valid: distances are translation invariant), then the centred
// feature-major copy Xc[k*n + i] and squared row norms for the diameter kernel.
__global__ void colsum_partial_kernel(const double *__restrict__ Xrow, i64 n, i64 d, double *__restrict__ part) {
    // block b sums rows b, b+G, ...; thread t handles columns t, t+256, ...
    for (i64 k = threadIdx.x; k < d; k += blockDim.x) {
        double s = 0.0;
        for (i64 i = blockIdx.x; i < n; i += gridDim.x) s += Xrow[i * d + k];
        part[(i64)blockIdx.x * d + k] = s;
    }
}
__global__ void colsum_final_kernel(const double *__restrict__ part, i64 nb, i64 n, i64 d, double *__restrict__ mean) {
    i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d) return;
    double s = 0.0;
    for (i64 b = 0; b < nb; b++) s += part[b * d + k];
    mean[k] = s / (double)n;
}
// dst[k*ld + p] = src[idx ? idx[p] : p][k] - mean[k]  (row-major rows -> centred feature-major, 32x32 LDS tiles)
// `planes` (optional): the same values as two row-major bf16 planes [2][ld][KP] (v = h + l + O(2^-16 v)), the operands of
// the bf16-split bound pass (kernels_dist.hip (2c)); *flag |= 1 when a value is unfit for that split.
// A workgroup takes 32 rows and walks their features in tiles of 32, so the squared norms of the centred rows
// (rnorm[p], optional) come out of the same pass (per row: the features of a tile in ascending order per thread, the
// eight threads of a row and then the tiles combined in fixed order).
__global__ void gather_centre_fm_kernel(const double *__restrict__ src, const i32 *__restrict__ idx,
                                        const double *__restrict__ mean, double *__restrict__ dst, i64 npos, i64 d,
                                        i64 ld, float *__restrict__ dst32, unsigned short *__restrict__ planes, i64 KP,
                                        int *__restrict__ flag, double *__restrict__ rnorm) {
    __shared__ double tile[32][33];
    __shared__ double nrm[8][33];
    const i64 p0 = (i64)blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
    bool bad = false, big = false;
    double sq = 0.0; // of row p0 + tx, the features this thread visits in the second phase
    for (i64 k0 = 0; k0 < d; k0 += 32) {
        for (int r = ty; r < 32; r += 8) {
            const i64 p = p0 + r, k = k0 + tx;
            tile[r][tx] = 0.0;
            // (a negative index: a position whose row lives on another rank -- option shard_rows; it stays zero here and the
            // gather over the ranks fills it)
            if (p < npos && k < d && (!idx || idx[p] >= 0)) {
                const double v = src[(idx ? (i64)idx[p] : p) * d + k] - mean[k];
                tile[r][tx] = v;
                if (flag) { // fitness for the low-precision bound passes (bit 0: unfit value; bit 1: a value of ordinary size exists)
                    const double av = fabs(v);
                    if (!(av < 1.2676506002282294e30) || (av != 0.0 && av < 7.888609052210118e-31)) bad = true; // 2^100, 2^-100
                    if (av >= 9.094947017729282e-13) big = true; // 2^-40
                }
                if (planes) {
                    unsigned u = __float_as_uint((float)v);
                    u += 0x7FFFu + ((u >> 16) & 1u); // bf16, round to nearest even
                    const unsigned short h = (unsigned short)(u >> 16);
                    unsigned w = __float_as_uint((float)(v - (double)__uint_as_float((unsigned)h << 16)));
                    w += 0x7FFFu + ((w >> 16) & 1u);
                    planes[p * KP + k] = h;
                    planes[ld * KP + p * KP + k] = (unsigned short)(w >> 16);
                }
            }
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const i64 p = p0 + tx, k = k0 + r;
            if (p < npos && k < d) {
                const double v = tile[tx][r];
                if (dst) dst[k * ld + p] = v;
                if (dst32) dst32[k * ld + p] = (float)v; // operand of the fp32-MFMA bound pass
                sq += v * v;
            }
        }
        __syncthreads();
    }
    if (flag) { // one atomic per workgroup (every thread of a fit embedding sees an ordinary-sized value: millions of
        // same-address atomics cost the launch 0.8 ms)
        const int any_bad = __syncthreads_or(bad ? 1 : 0), any_big = __syncthreads_or(big ? 1 : 0);
        const int need = (any_bad ? 1 : 0) | (any_big ? 2 : 0);
        if (tx == 0 && ty == 0 && need && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & need) != need)
            atomicOr(flag, need); // (a plain look first: same-address atomics of every workgroup serialise)
    }
    if (rnorm) {
        nrm[ty][tx] = sq;
        __syncthreads();
        if (ty == 0 && p0 + tx < npos) {
            double s = nrm[0][tx];
#pragma unroll
            for (int q = 1; q < 8; q++) s += nrm[q][tx];
            rnorm[p0 + tx] = s;
        }
    }
}
// `sums_only` != 0: the column SUMS (the sharded rows: the ranks add their sums and divide by the global n)
void k_col_mean(cge_ctx *c, const double *Xrow, i64 n, i64 d, double *mean, double sums_only) {
    const int NB = 512;
    DevBuf<double> part;
    part.ensure((size_t)NB * d);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(NB), dim3(256), 0, c->stream, Xrow, n, d, part.p);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, c->stream, part.p,
                       (i64)NB, sums_only != 0.0 ? (i64)1 : n, d, mean);
    HIP_CHECK(hipStreamSynchronize(c->stream)); // part is freed on return
}
__global__ void scale_vector_kernel(double *__restrict__ v, i64 n, double f) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= f;
}
void k_scale_vector(cge_ctx *c, double *v, i64 n, double f) {
    hipLaunchKernelGGL(scale_vector_kernel, dim3(grid_for(n, 256, 1 << 20)), dim3(256), 0, c->stream, v, n, f);
}
// out[i][k] = X[idx[i]][k] for a row-major (n x d) or column-major X: this rank's rows of a device-resident embedding
__global__ void gather_rows_f64_kernel(const double *__restrict__ X, i64 n, i64 d, int row_major, const i32 *__restrict__ idx,
                                       i64 cnt, double *__restrict__ out) {
    const i64 total = cnt * d, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 i = e / d, k = e - i * d, g = idx[i];
        out[e] = g < 0 ? 0.0 : (row_major ? X[g * d + k] : X[k * n + g]); // (a negative index: a zero row -- another rank fills it)
    }
}
void k_gather_rows_f64(cge_ctx *c, const double *X, i64 n, i64 d, int row_major, const i32 *idx, i64 cnt, double *out) {
    if (cnt <= 0) return;
    hipLaunchKernelGGL(gather_rows_f64_kernel, dim3(grid_for(cnt * d, 256, 8192)), dim3(256), 0, c->stream, X, n, d, row_major,
                       idx, cnt, out);
}
// The planes-only form (round 4): when the bound pass reads the bf16 planes (row-major) and nobody needs the feature-major
// fp64 copy, no transposition is needed at all -- a wave takes whole rows (two consecutive features per lane: one 16-byte
// load per lane, a row of d = 128 in ONE request), four rows in flight, writes the two planes with 4-byte stores and reduces
// the squared norm inside the wave.  The tile kernel above asks for every row in four 256-byte pieces between barriers and
// ran at half of what HBM gives a gather of 1 KB rows (profiles/r04_rocprofv3_summary.md).
__global__ __launch_bounds__(256) void gather_planes_kernel(const double *__restrict__ src, const i32 *__restrict__ idx,
                                                            const double *__restrict__ mean, i64 npos, i64 d, i64 ld,
                                                            unsigned short *__restrict__ planes, i64 KP, int *__restrict__ flag,
                                                            double *__restrict__ rnorm) {
    const int lane = threadIdx.x & 63;
    const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    bool bad = false, big = false;
    constexpr int R = 4;
    for (i64 p0 = wave * R; p0 < npos; p0 += nwaves * R) {
        i64 node[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const i64 p = p0 + r;
            node[r] = p < npos ? (idx ? (i64)idx[p] : p) : -1; // (negative: a row of another rank / past the end: zeros)
        }
        double sq[R];
#pragma unroll
        for (int r = 0; r < R; r++) sq[r] = 0.0;
        for (i64 k = 2 * lane; k < KP; k += 128) {
            double m0 = k < d ? mean[k] : 0.0, m1 = k + 1 < d ? mean[k + 1] : 0.0;
            double v0[R], v1[R];
#pragma unroll
            for (int r = 0; r < R; r++) { // all loads of the R rows in flight before any use
                const double *row = src + (node[r] < 0 ? 0 : node[r]) * d;
                const bool in = node[r] >= 0;
                if ((d & 1) == 0 && k + 1 < d) {
                    const double2 t = *reinterpret_cast<const double2 *>(row + k);
                    v0[r] = in ? t.x : m0;
                    v1[r] = in ? t.y : m1;
                } else {
                    v0[r] = (in && k < d) ? row[k] : m0;
                    v1[r] = (in && k + 1 < d) ? row[k + 1] : m1;
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const i64 p = p0 + r;
                if (p >= npos) continue;
                const double a = v0[r] - m0, b = v1[r] - m1;
                unsigned short h[2], l[2];
                const double vv[2] = {a, b};
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    // The split in fp32 (the fp64 conversions and subtraction of the tile kernel made the gather compute-bound):
                    // f = fl32(v), h = bf16(f), l = bf16(f - h) with f - h exact, so v = h + l + e with
                    // |e| <= |v - f| + |(f - h) - l| <= (2^-24 + 2^-9 2^-9) |v| < 2^-16 |v|: inside the margin of (2c).
                    // (A double below the fp32 range becomes 0: an absolute error far below anything the margin compares.)
                    const float f = (float)vv[q], af = fabsf(f);
                    if (flag) {
                        if (!(af < 1.2676506e30f) || (af != 0.0f && af < 7.8886091e-31f)) bad = true; // 2^100, 2^-100, not finite
                        if (af >= 9.094947e-13f) big = true; // 2^-40
                    }
                    unsigned u = __float_as_uint(f);
                    u += 0x7FFFu + ((u >> 16) & 1u); // bf16, round to nearest even
                    h[q] = (unsigned short)(u >> 16);
                    unsigned w = __float_as_uint(f - __uint_as_float((unsigned)h[q] << 16));
                    w += 0x7FFFu + ((w >> 16) & 1u);
                    l[q] = (unsigned short)(w >> 16);
                }
                // (KP is a multiple of 32 and k even: the pair is inside the row and 4-byte aligned)
                *reinterpret_cast<unsigned *>(planes + p * KP + k) = (unsigned)h[0] | ((unsigned)h[1] << 16);
                *reinterpret_cast<unsigned *>(planes + ld * KP + p * KP + k) = (unsigned)l[0] | ((unsigned)l[1] << 16);
                sq[r] += a * a + b * b;
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double s = wave_allsum(sq[r]);
            if (lane == 0 && p0 + r < npos) rnorm[p0 + r] = s;
        }
    }
    if (flag) { // one atomic per wave
        // (a plain look first: once the bits are up -- after the first few waves -- nobody touches the word again; 65 000
        // same-address atomics alone took 0.5 ms)
        const unsigned long long mb = __ballot(bad), mg = __ballot(big);
        const int need = (mb ? 1 : 0) | (mg ? 2 : 0);
        if (lane == 0 && need && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & need) != need) atomicOr(flag, need);
    }
}
// Centred, zero-padded feature-major copy (dpad x ld) of npos gathered rows + squared row norms (ld entries).
void k_gather_centre_fm(cge_ctx *c, const double *src_rowmajor, const i32 *idx, const double *mean, double *dst,
                        double *rnorm, i64 npos, i64 d, i64 ld, i64 dpad, float *dst32, unsigned short *planes, i64 KP,
                        int *flag) {
    // (`dst` = nullptr: no fp64 copy -- the low-precision bound passes read the planes / the f32 copy, and the exact stage of
    // the diameter gathers only the candidate landmarks' rows)
    const bool planes_only = planes && !dst && !dst32 && npos > 0;
    if (dst) HIP_CHECK(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)(ld * dpad), c->stream));
    if (planes_only) { // the kernel writes every row below npos whole (padding features included): only the tail rows are cleared
        for (int q = 0; q < 2; q++)
            HIP_CHECK(hipMemsetAsync(planes + (size_t)q * ld * KP + (size_t)npos * KP, 0, sizeof(unsigned short) * (size_t)((ld - npos) * KP), c->stream));
    } else if (planes)
        HIP_CHECK(hipMemsetAsync(planes, 0, sizeof(unsigned short) * (size_t)(2 * ld * KP), c->stream));
    if (dst32) HIP_CHECK(hipMemsetAsync(dst32, 0, sizeof(float) * (size_t)(ld * dpad), c->stream));
    HIP_CHECK(hipMemsetAsync(rnorm, 0, sizeof(double) * (size_t)ld, c->stream));
    if (planes_only) { // the operands of the bf16 bound pass and nothing else
        const unsigned nb = (unsigned)std::min<i64>((npos + 15) / 16, 16384); // 4 waves x 4 rows per workgroup and round
        hipLaunchKernelGGL(gather_planes_kernel, dim3(nb), dim3(256), 0, c->stream, src_rowmajor, idx, mean, npos, d, ld, planes, KP,
                           flag, rnorm);
        return;
    }
    dim3 grid((unsigned)((npos + 31) / 32));
    hipLaunchKernelGGL(gather_centre_fm_kernel, grid, dim3(32, 8), 0, c->stream, src_rowmajor, idx, mean, dst, npos, d,
                       ld, dst32, planes, KP, flag, rnorm);
}

// ------------------------------------------------------------------------------------------------
// 64-bit hash of each row's bit pattern (for the `unique(embedding, dims=1)` clamp, src/landmarks.jl:371).
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__global__ void row_hash_kernel(const double *__restrict__ Xrow, uint64_t *__restrict__ hash, i64 n, i64 d) {
    i64 row = ((i64)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    int lane = threadIdx.x & (WAVE - 1);
    if (row >= n) return;
    const uint64_t *p = reinterpret_cast<const uint64_t *>(Xrow + row * d);
    uint64_t h = 0;
    for (i64 k = lane; k < d; k += WAVE) h += mix64(p[k] ^ (0x9e3779b97f4a7c15ULL * (uint64_t)(k + 1)));
    for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
    if (lane == 0) hash[row] = mix64(h);
}
void k_row_hash(cge_ctx *c, const double *Xrow, uint64_t *hash, i64 n, i64 d) {
    i64 threads = n * WAVE;
    hipLaunchKernelGGL(row_hash_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, Xrow, hash,
                       n, d);
}
// number of DISTINCT values among hash[0 .. n): an open-addressing set in `table` (tsize slots, a power of two >= 2 n, all
// ones = empty), every first insertion counted.  (A hash equal to the empty marker is folded onto another value: the count
// is then a lower bound of the distinct rows as well, which is all the caller needs.)
__global__ void count_distinct_kernel(const uint64_t *__restrict__ hash, i64 n, unsigned long long *__restrict__ table, i64 mask,
                                      unsigned long long *__restrict__ count) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long key = hash[i];
        if (key == ~0ULL) key = 0x5bd1e995ULL;
        i64 slot = (i64)(mix64(key) & (uint64_t)mask);
        for (;;) {
            const unsigned long long old = atomicCAS(&table[slot], ~0ULL, key);
            if (old == ~0ULL) { mine++; break; }
            if (old == key) break;
            slot = (slot + 1) & mask;
        }
    }
    if (mine) atomicAdd(count, mine);
}
i64 k_count_distinct(cge_ctx *c, const uint64_t *hash, i64 n) {
    i64 tsize = 1024;
    while (tsize < 2 * n) tsize <<= 1;
    c->uniq_table.ensure((size_t)tsize + 1);
    HIP_CHECK(hipMemsetAsync(c->uniq_table.p, 0xFF, sizeof(unsigned long long) * tsize, c->stream));
    HIP_CHECK(hipMemsetAsync(c->uniq_table.p + tsize, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(count_distinct_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, c->stream, hash, n, c->uniq_table.p, tsize - 1,
                       c->uniq_table.p + tsize);
    unsigned long long cnt = 0;
    HIP_CHECK(hipMemcpyAsync(&cnt, c->uniq_table.p + tsize, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return (i64)cnt;
}

// ------------------------------------------------------------------------------------------------
// Batched group statistics.  A batch is a list of tasks (groups); `rows` holds their 0-based
// vertex ids back to back; a task is cut into chunks of rows so that big groups use many CUs;
// chunk partials are combined in chunk order (fixed order => reproducible).
#define MAXSLOT 16 // columns handled per lane: d <= 64*MAXSLOT = 1024

template <int NS> // NS = columns per lane (d <= 64*NS)
__global__ __launch_bounds__(256) void group_mean_partial_kernel(const double *__restrict__ Xr,
                                                                 const double *__restrict__ vw,
                                                                 const i32 *__restrict__ rows,
                                                                 const i32 *__restrict__ chunk_beg,
                                                                 const i32 *__restrict__ chunk_end, i64 d,
                                                                 double *__restrict__ part /* [chunk][d+1] */) {
    __shared__ double red[4][WAVE * 4 + 1]; // reused per slot group
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 ch = blockIdx.x;
    const i32 beg = chunk_beg[ch], end = chunk_end[ch];
    double acc[MAXSLOT];
#pragma unroll
    for (int s = 0; s < MAXSLOT; s++) acc[s] = 0.0;
    double wsum = 0.0;
    constexpr int RF = NS <= 4 ? 4 : 1; // rows in flight per wave (the sums keep the row order)
    i32 j = beg + wave;
    for (; j + 4 * (RF - 1) < end; j += 4 * RF) {
        double xv[RF][NS], w[RF];
#pragma unroll
        for (int q = 0; q < RF; q++) {
            const i64 v = rows[j + 4 * q];
            w[q] = vw[v];
            const double *x = Xr + v * d;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const i64 col = lane + 64 * s;
                xv[q][s] = (col < d) ? x[col] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < RF; q++) {
            wsum += w[q];
#pragma unroll
            for (int s = 0; s < NS; s++) acc[s] += xv[q][s] * w[q];
        }
    }
    for (; j < end; j += 4) {
        const i64 v = rows[j];
        const double w = vw[v];
        wsum += w;
        const double *x = Xr + v * d;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const i64 col = lane + 64 * s;
            if (col < d) acc[s] += x[col] * w;
        }
    }
    double *out = part + ch * (d + 1);
    // combine the 4 waves in wave order
    for (int s0 = 0; s0 < MAXSLOT; s0 += 4) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; s++) red[wave][lane * 4 + s] = acc[s0 + s];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                i64 col = lane + 64 * (s0 + s);
                if (col < d) out[col] = ((red[0][lane * 4 + s] + red[1][lane * 4 + s]) + red[2][lane * 4 + s]) +
                                        red[3][lane * 4 + s];
            }
        }
        if ((i64)64 * (s0 + 4) >= d) break;
    }
    __syncthreads();
    if (lane == 0) red[wave][0] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) out[d] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
}
__global__ void group_mean_final_kernel(const double *__restrict__ part, const i32 *__restrict__ task_chunk_off, i64 d,
                                        double *__restrict__ mean, double *__restrict__ sw) {
    const i64 t = blockIdx.x;
    const i32 c0 = task_chunk_off[t], c1 = task_chunk_off[t + 1];
    double wt = 0.0;
    for (i32 ch = c0; ch < c1; ch++) wt += part[(i64)ch * (d + 1) + d];
    for (i64 col = threadIdx.x; col < d; col += blockDim.x) {
        double s = 0.0;
        for (i32 ch = c0; ch < c1; ch++) s += part[(i64)ch * (d + 1) + col];
        mean[t * d + col] = s / wt;
    }
    if (threadIdx.x == 0) sw[t] = wt;
}
void k_group_mean(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const i32 *chunk_task,
                  const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks,
                  i64 d, double *part, double *mean, double *sw) {
    (void)chunk_task;
    if (d > 64 * MAXSLOT) CGE_THROW(CGE_E_ARG, "embedding dimension %lld > %d not supported", (long long)d, 64 * MAXSLOT);
    const dim3 grid((unsigned)n_chunks), block(256);
#define CGE_MEAN_LAUNCH(NS)                                                                                            \
    hipLaunchKernelGGL((group_mean_partial_kernel<NS>), grid, block, 0, c->stream, Xr, vw, rows, chunk_beg, chunk_end, d, \
                       part)
    if (d <= 64) CGE_MEAN_LAUNCH(1);
    else if (d <= 128) CGE_MEAN_LAUNCH(2);
    else if (d <= 256) CGE_MEAN_LAUNCH(4);
    else if (d <= 512) CGE_MEAN_LAUNCH(8);
    else CGE_MEAN_LAUNCH(16);
#undef CGE_MEAN_LAUNCH
    hipLaunchKernelGGL(group_mean_final_kernel, dim3((unsigned)n_tasks), dim3(256), 0, c->stream, part,
                       task_chunk_off, d, mean, sw);
}

// covariance A = sum_j w_j (x_j - mu)(x_j - mu)^T, per chunk then per task.
// Thread tiles of TS x TS accumulators over an LDS tile of centred, sqrt(w)-scaled rows.
template <int TS>
__global__ __launch_bounds__(256) void group_cov_partial_kernel(const double *__restrict__ Xr,
                                                                const double *__restrict__ vw,
                                                                const i32 *__restrict__ rows,
                                                                const i32 *__restrict__ chunk_task,
                                                                const i32 *__restrict__ chunk_beg,
                                                                const i32 *__restrict__ chunk_end, i64 d, int dp,
                                                                int RT, const double *__restrict__ mean,
                                                                double *__restrict__ part /* [chunk][d*d] */) {
    extern __shared__ __attribute__((aligned(16))) double ytile[]; // RT x (dp + 2)
    const int ld = dp + 2;
    const i64 ch = blockIdx.x;
    const i32 beg = chunk_beg[ch], end = chunk_end[ch];
    const double *mu = mean + (i64)chunk_task[ch] * d;
    const int nt = dp / TS;          // tiles per dimension
    const int ntiles = nt * nt;
    double *out = part + ch * d * d;
    for (int tbase = 0; tbase < ntiles; tbase += 256) {
        const int tile = tbase + threadIdx.x;
        const bool active = tile < ntiles;
        const int ta = active ? (tile / nt) * TS : 0, tb = active ? (tile % nt) * TS : 0;
        double acc[TS][TS];
#pragma unroll
        for (int a = 0; a < TS; a++)
#pragma unroll
            for (int b = 0; b < TS; b++) acc[a][b] = 0.0;
        for (i32 j0 = beg; j0 < end; j0 += RT) {
            const int nr = min(RT, end - j0);
            __syncthreads();
            for (int e = threadIdx.x; e < nr * dp; e += 256) {
                const int r = e / dp, col = e - r * dp;
                const i64 v = rows[j0 + r];
                double val = 0.0;
                if (col < d) val = (Xr[v * d + col] - mu[col]) * sqrt(vw[v]);
                ytile[r * ld + col] = val;
            }
            __syncthreads();
            if (active) {
                for (int r = 0; r < nr; r++) {
                    double ya[TS], yb[TS];
#pragma unroll
                    for (int a = 0; a < TS; a++) ya[a] = ytile[r * ld + ta + a];
#pragma unroll
                    for (int b = 0; b < TS; b++) yb[b] = ytile[r * ld + tb + b];
#pragma unroll
                    for (int a = 0; a < TS; a++)
#pragma unroll
                        for (int b = 0; b < TS; b++) acc[a][b] = fma(ya[a], yb[b], acc[a][b]);
                }
            }
        }
        if (active) {
#pragma unroll
            for (int a = 0; a < TS; a++)
#pragma unroll
                for (int b = 0; b < TS; b++)
                    if (ta + a < d && tb + b < d) out[(i64)(ta + a) * d + (tb + b)] = acc[a][b];
        }
    }
}
// ---- covariance on the fp64 matrix cores (d >= 96) ----------------------------------------------------
// A_chunk = Y^T Y with y_j = (x_j - mu) * sqrt(w_j) (src/landmarks.jl:71-81,:97-98) is the same 128x128 MFMA Gram
// tile as the distance kernels, with the sample index as the contraction dimension (operand "k-row" = one sample,
// 128 contiguous features).  The rows are gathered from the embedding by vertex id, centred and scaled on the
// way into LDS (no intermediate Y buffer); chunks are padded with zero rows to a multiple of 16.  A diagonal tile
// (the only one when d <= 128) stages its operand once.
template <bool SAME>
__global__ __launch_bounds__(256, 2) void group_cov_mfma_kernel(const double *__restrict__ Xr,
                                                                const double *__restrict__ vw,
                                                                const i32 *__restrict__ rows,
                                                                const i32 *__restrict__ chunk_task,
                                                                const i32 *__restrict__ chunk_beg,
                                                                const i32 *__restrict__ chunk_end, i64 d, i64 nT,
                                                                const double *__restrict__ mean,
                                                                double *__restrict__ part /* [chunk][d*d] */,
                                                                const i32 *__restrict__ task_chunk_off, double *__restrict__ cov) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ i32 s_row[CGE_CHUNK_ROWS];
    __shared__ double s_sq[CGE_CHUNK_ROWS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4, c2 = lane * 2;
    const i64 ch = blockIdx.x;
    i64 ta, tb; // tile pair: all of them, or (SAME) the diagonal ones
    if (SAME) {
        ta = tb = blockIdx.y;
    } else { // tile pairs above the diagonal only (the reduction mirrors them): blockIdx.y -> (ta < tb)
        i64 rem = blockIdx.y;
        ta = 0;
        while (rem >= nT - 1 - ta) { rem -= nT - 1 - ta; ta++; }
        tb = ta + 1 + rem;
    }
    const i64 a0 = ta * 128, b0 = tb * 128;
    const i32 beg = chunk_beg[ch];
    const i64 len = chunk_end[ch] - beg, len16 = (len + 15) / 16 * 16;
    for (i64 r = tid; r < len; r += 256) {
        const i32 v = rows[beg + r];
        s_row[r] = v;
        s_sq[r] = sqrt(vw[v]);
    }
    const double *mu = mean + (i64)chunk_task[ch] * d;
    const i64 ca = a0 + c2, cb = b0 + c2;
    const bool pair_ok = (d & 1) == 0; // 16-byte aligned pairs
    const double ma0 = ca < d ? mu[ca] : 0.0, ma1 = ca + 1 < d ? mu[ca + 1] : 0.0;
    const double mb0 = (!SAME && cb < d) ? mu[cb] : 0.0, mb1 = (!SAME && cb + 1 < d) ? mu[cb + 1] : 0.0;
    __syncthreads();
    // The loader only LOADS (from clamped, always valid addresses: rows beyond the chunk read its first row with a zero
    // scale, columns beyond d read column 0 and are masked), the centring and scaling happen when the pair is staged.
    struct RawRow { d2 x; double sq; };
    auto load = [&](i64 kc, int q, i64 col) {
        const i64 kr = kc * MP_BK + wave + 4 * q, krc = kr < len ? kr : 0;
        const double *x = Xr + (i64)s_row[krc] * d;
        const i64 e0 = col < d ? col : 0, e1 = col + 1 < d ? col + 1 : 0;
        RawRow r;
        r.sq = kr < len ? s_sq[krc] : 0.0;
        if (pair_ok) r.x = *reinterpret_cast<const d2 *>(x + e0); // d even, col even: e1 == e0 + 1 whenever col < d
        else r.x = (d2){x[e0], x[e1]};
        return r;
    };
    auto finish = [&](const RawRow &r, i64 col, double m0, double m1) {
        return (d2){col < d ? (r.x[0] - m0) * r.sq : 0.0, col + 1 < d ? (r.x[1] - m1) * r.sq : 0.0};
    };
    double *out = part + ch * d * d;
    if (SAME && nT == 1) { // a group of ONE chunk (most groups of a batch): its partial IS the covariance -- straight to `cov`, and
        const i64 t = chunk_task[ch]; // group_cov_final_kernel skips the group (no round trip of d x d doubles through `part`)
        if (task_chunk_off[t + 1] - task_chunk_off[t] == 1) out = cov + t * d * d;
    }
    if (SAME) { // upper-triangular blocks only, mirrored on the way out
        d4 acc[9];
        auto la = [&](i64 kc, int q) { return load(kc, q, ca); };
        auto fa = [&](const RawRow &r) { return finish(r, ca, ma0, ma1); };
        switch (wave) {
        case 0: syrk_tile_128_wave<0>(la, fa, len16 / MP_BK, lds, acc, wave, c2, lr, lk); break;
        case 1: syrk_tile_128_wave<1>(la, fa, len16 / MP_BK, lds, acc, wave, c2, lr, lk); break;
        case 2: syrk_tile_128_wave<2>(la, fa, len16 / MP_BK, lds, acc, wave, c2, lr, lk); break;
        default: syrk_tile_128_wave<3>(la, fa, len16 / MP_BK, lds, acc, wave, c2, lr, lk); break;
        }
#pragma unroll
        for (int q = 0; q < 9; q++) {
            int bi, bj;
            syrk_slot_block(wave, q, bi, bj);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const i64 ia = a0 + 16 * bi + lk + 4 * r, ib = a0 + 16 * bj + lr;
                if (ia < d && ib < d) {
                    out[ia * d + ib] = acc[q][r];
                    if (bi != bj) out[ib * d + ia] = acc[q][r];
                }
            }
        }
        return;
    }
    d4 acc[4][4];
    gram_tile_128_ld<false>([&](i64 kc, int q) { return load(kc, q, ca); }, [&](i64 kc, int q) { return load(kc, q, cb); },
                            [&](const RawRow &r) { return finish(r, ca, ma0, ma1); },
                            [&](const RawRow &r) { return finish(r, cb, mb0, mb1); }, len16 / MP_BK, lds, acc, wave, c2, wr,
                            wc, lr, lk);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const i64 ia = a0 + wr * 64 + a * 16 + lk + 4 * r, ib = b0 + wc * 64 + b * 16 + lr;
                if (ia < d && ib < d) out[ia * d + ib] = acc[a][b][r];
            }
}
// `tiled` (d > 128, MFMA path): the chunk partials hold the 128 x 128 tiles on and above the diagonal only; the tiles
// below it are the mirror images, written from the reduced sums.
__global__ void group_cov_final_kernel(const double *__restrict__ part, const i32 *__restrict__ task_chunk_off, i64 d,
                                       int tiled, int direct_single, double *__restrict__ cov) {
    const i64 t = blockIdx.y, dd = d * d;
    const i32 c0 = task_chunk_off[t], c1 = task_chunk_off[t + 1];
    if (direct_single && c1 - c0 == 1) return; // (written by the chunk's own kernel)
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < dd; e += (i64)gridDim.x * blockDim.x) {
        const i64 i = e / d, j = e - i * d;
        if (tiled && (i >> 7) > (j >> 7)) continue;
        double s = 0.0;
        for (i32 ch = c0; ch < c1; ch++) s += part[(i64)ch * dd + e];
        cov[t * dd + e] = s;
        if (tiled && (i >> 7) < (j >> 7)) cov[t * dd + j * d + i] = s;
    }
}
void k_group_cov(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const i32 *chunk_task,
                 const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks,
                 i64 d, const double *mean, double *part, double *cov) {
    dim3 grid((unsigned)n_chunks), block(256);
    if (d >= 48) { // fp64 MFMA SYRK (below a full tile too: a chunk is bound by its latency chain, not by the flops of the padding)
        const i64 nT = (d + 127) / 128;
        const size_t stage = (size_t)2 * MP_BK * MP_LD * sizeof(double);
        hipLaunchKernelGGL((group_cov_mfma_kernel<true>), dim3((unsigned)n_chunks, (unsigned)nT), block, stage, c->stream,
                           Xr, vw, rows, chunk_task, chunk_beg, chunk_end, d, nT, mean, part, task_chunk_off, cov);
        if (nT > 1)
            hipLaunchKernelGGL((group_cov_mfma_kernel<false>), dim3((unsigned)n_chunks, (unsigned)(nT * (nT - 1) / 2)), block,
                               2 * stage, c->stream, Xr, vw, rows, chunk_task, chunk_beg, chunk_end, d, nT, mean, part, task_chunk_off, cov);
    } else {
        const int dpv = (int)((d + 7) / 8 * 8);
        int RT = 64;
        while ((size_t)RT * (dpv + 2) * sizeof(double) > 64 * 1024 && RT > 4) RT /= 2;
        const size_t lds = (size_t)RT * (dpv + 2) * sizeof(double);
        if (dpv >= 128)
            hipLaunchKernelGGL(group_cov_partial_kernel<8>, grid, block, lds, c->stream, Xr, vw, rows, chunk_task,
                               chunk_beg, chunk_end, d, dpv, RT, mean, part);
        else if (dpv >= 64)
            hipLaunchKernelGGL(group_cov_partial_kernel<4>, grid, block, lds, c->stream, Xr, vw, rows, chunk_task,
                               chunk_beg, chunk_end, d, dpv, RT, mean, part);
        else
            hipLaunchKernelGGL(group_cov_partial_kernel<2>, grid, block, lds, c->stream, Xr, vw, rows, chunk_task,
                               chunk_beg, chunk_end, d, dpv, RT, mean, part);
    }
    const i64 dd = d * d;
    dim3 g2((unsigned)std::min<i64>((dd + 255) / 256, 64), (unsigned)n_tasks);
    hipLaunchKernelGGL(group_cov_final_kernel, g2, dim3(256), 0, c->stream, part, task_chunk_off, d, (int)(d > 128),
                       (int)(d >= 48 && d <= 128), cov);
}

// ------------------------------------------------------------------------------------------------
// Per-task, per-side WSSE column sums (src/landmarks.jl:50-67): for the rows of a task whose side[j] is
// 1 or 2, out[task][side-1] = { sum w x^2 [d], sum w x [d], sum w }.  Used by the rss rule's median-cut
// rounds (:184-185, :201-202) and for the children's total_rss (:269).  Rows with side 0 are skipped.
// Fixed order: rows of a chunk are dealt to the 4 waves round-robin, waves and chunks combined in order.
#define SS_SLOTS 8 // columns per lane: d <= 512
template <int NS> // NS = columns per lane actually used (d <= 64*NS)
__global__ __launch_bounds__(256) void group_side_sums_partial_kernel(const double *__restrict__ Xr,
                                                                      const double *__restrict__ vw,
                                                                      const i32 *__restrict__ rows,
                                                                      const unsigned char *__restrict__ side,
                                                                      const i32 *__restrict__ chunk_beg,
                                                                      const i32 *__restrict__ chunk_end, i64 d,
                                                                      double *__restrict__ part /* [chunk][2][2d+1] */) {
    __shared__ double red[4][64 + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 ch = blockIdx.x;
    const i32 beg = chunk_beg[ch], end = chunk_end[ch];
    double ss[2][SS_SLOTS], s1[2][SS_SLOTS], ws[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int t = 0; t < SS_SLOTS; t++) { ss[q][t] = 0.0; s1[q][t] = 0.0; }
    // rows in flight per wave; the additions keep the row order (wave w takes rows beg + w, + 4, + 8, ...).  Lanes 0 .. RF-1
    // fetch the ids, sides and weights of the wave's next RF rows in one request each, the ids travel to scalar registers
    // (v_readlane) and all row loads are in flight together -- what the projection kernel does
    constexpr int RF = NS <= 2 ? 16 : NS <= 4 ? 8 : 4;
    for (i32 j0 = beg + wave; j0 < end; j0 += 4 * RF) {
        const i32 jl = j0 + 4 * (lane & (RF - 1));
        const bool onl = jl < end;
        const i32 vl = rows[onl ? jl : beg];
        const int sdl = onl ? (int)side[jl] : 0;
        const double wl = vw[vl];
        double xv[RF][NS];
#pragma unroll
        for (int u = 0; u < RF; u++) {
            const i64 v = __builtin_amdgcn_readlane(vl, u);
            const double *x = Xr + v * d;
#pragma unroll
            for (int t = 0; t < NS; t++) {
                const i64 col = lane + 64 * t;
                xv[u][t] = (col < d) ? x[col] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < RF; u++) {
            const int sd = __builtin_amdgcn_readlane(sdl, u); // wave-uniform
            if (sd == 0) continue;
            const int q = sd == 1 ? 0 : 1;
            const double w = lane_value(wl, u);
            ws[q] += w;
#pragma unroll
            for (int t = 0; t < NS; t++) { ss[q][t] += w * (xv[u][t] * xv[u][t]); s1[q][t] += w * xv[u][t]; }
        }
    }
    double *out = part + ch * 2 * (2 * d + 1);
#pragma unroll
    for (int q = 0; q < 2; q++) {
#pragma unroll
        for (int t = 0; t < SS_SLOTS; t++) {
            if ((i64)64 * t >= d) break;
            const i64 col = lane + 64 * t;
            __syncthreads();
            red[wave][lane] = ss[q][t];
            __syncthreads();
            if (wave == 0 && col < d) out[q * (2 * d + 1) + col] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
            __syncthreads();
            red[wave][lane] = s1[q][t];
            __syncthreads();
            if (wave == 0 && col < d) out[q * (2 * d + 1) + d + col] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        }
        __syncthreads();
        if (lane == 0) red[wave][0] = ws[q];
        __syncthreads();
        if (threadIdx.x == 0) out[q * (2 * d + 1) + 2 * d] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
    }
}
__global__ void group_side_sums_final_kernel(const double *__restrict__ part, const i32 *__restrict__ task_chunk_off,
                                             i64 width, double *__restrict__ out) {
    const i64 t = blockIdx.x;
    const i32 c0 = task_chunk_off[t], c1 = task_chunk_off[t + 1];
    for (i64 e = threadIdx.x; e < width; e += blockDim.x) {
        double s = 0.0;
        for (i32 ch = c0; ch < c1; ch++) s += part[(i64)ch * width + e];
        out[t * width + e] = s;
    }
}
void k_group_side_sums(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const unsigned char *side,
                       const i32 *chunk_beg, const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks,
                       i64 d, double *part, double *out) {
    if (d > 64 * SS_SLOTS) CGE_THROW(CGE_E_ARG, "embedding dimension %lld > %d not supported", (long long)d, 64 * SS_SLOTS);
    ScopedKernelTimer t(c, "group_side_sums");
    const dim3 grid((unsigned)n_chunks), block(256);
#define CGE_SS_LAUNCH(NS) \
    hipLaunchKernelGGL((group_side_sums_partial_kernel<NS>), grid, block, 0, c->stream, Xr, vw, rows, side, chunk_beg, chunk_end, d, part)
    if (d <= 64) CGE_SS_LAUNCH(1);
    else if (d <= 128) CGE_SS_LAUNCH(2);
    else if (d <= 256) CGE_SS_LAUNCH(4);
    else CGE_SS_LAUNCH(8);
#undef CGE_SS_LAUNCH
    hipLaunchKernelGGL(group_side_sums_final_kernel, dim3((unsigned)n_tasks), dim3(256), 0, c->stream, part,
                       task_chunk_off, 2 * (2 * d + 1), out);
}

// The cut rules' children from the side sums, on the device (round 4): vals[2t + q] = -total_rss of child q of task t (the
// host's rss_from_sums: sum over the columns of ss - s * s / ws, sequential, unfused) and the two means s / ws -- so the host
// reads 2 doubles per task instead of 2 (2 d + 1) and the means never make the round trip through the host.
__global__ __launch_bounds__(256) void side_values_means_kernel(const double *__restrict__ sums, i64 d, double *__restrict__ vals,
                                                                double *__restrict__ means /* [task][2][d] */) {
    const i64 t = blockIdx.x, W = 2 * d + 1;
    const double *q0 = sums + t * 2 * W;
    for (i64 e = threadIdx.x; e < 2 * d; e += blockDim.x) {
        const i64 q = e / d, c2 = e - q * d;
        const double *qq = q0 + q * W;
        means[(t * 2 + q) * d + c2] = __ddiv_rn(qq[d + c2], qq[2 * d]);
    }
    // the columns' terms ss - s * s / ws in parallel (the IEEE divisions are the expensive part), then the host's sequential sum
    extern __shared__ double term[]; // [2][d]
    for (i64 e = threadIdx.x; e < 2 * d; e += blockDim.x) {
        const i64 q = e / d, c2 = e - q * d;
        const double *qq = q0 + q * W;
        term[e] = __dsub_rn(qq[c2], __ddiv_rn(__dmul_rn(qq[d + c2], qq[d + c2]), qq[2 * d]));
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const double *tq = term + (i64)threadIdx.x * d;
        double tot = 0.0;
        for (i64 c2 = 0; c2 < d; c2++) tot = __dadd_rn(tot, tq[c2]);
        vals[2 * t + threadIdx.x] = -tot;
    }
}
void k_side_values_means(cge_ctx *c, const double *sums, i64 n_tasks, i64 d, double *vals, double *means) {
    if (n_tasks > 0)
        hipLaunchKernelGGL(side_values_means_kernel, dim3((unsigned)n_tasks), dim3(256), sizeof(double) * 2 * d, c->stream, sums, d, vals, means);
}

// ------------------------------------------------------------------------------------------------
// rss rule on sorted order.  Along ascending z every "gray" set of split_cluster_rss
// (src/landmarks.jl:168-199) is a contiguous rank range, and `t1 = {z < med}` is its lower part.  With
// inclusive prefix sums P[r] = sum_{q<=r} (w x^2 [d], w x [d], w) over the rows in sorted order, the WSSE
// triple of any range is P[b-1] - P[a-1], so the whole median-cut loop runs in one small kernel.
// srows = vertex ids in (task, ascending z) order; W = 2d+1.
// One pass: the inclusive prefix INSIDE each chunk (starting from 0), kept only at the end of every block of
// CGE_PREFIX_STRIDE rows (the rounds kernel re-adds the few rows of a partial block in the same order, so every
// P[r] has the bits of the full running sum at 1/8 of the write traffic), and the chunk total; a tiny second
// kernel turns the totals of a task's chunks into exclusive offsets.  P[r] = prefix-in-chunk[r] + coff[chunk of r].
// Block slots of chunk ch start at chunk_beg[ch] / STRIDE + ch (chunks never share a slot).
__global__ void scan_write_kernel(const double *__restrict__ Xr, const double *__restrict__ vw,
                                  const i32 *__restrict__ srows, const i32 *__restrict__ chunk_beg,
                                  const i32 *__restrict__ chunk_end, i64 d, i64 W, double *__restrict__ ctot,
                                  double *__restrict__ prefix) {
    constexpr int SB = CGE_PREFIX_STRIDE;
    const i64 ch = blockIdx.x;
    const i32 beg = chunk_beg[ch], end = chunk_end[ch];
    double *slots = prefix + ((i64)(beg / SB) + ch) * W;
    // thread c < d owns columns c (w x^2) and d + c (w x): one load of x per row; thread 0 also owns column 2d (w)
    for (i64 c = threadIdx.x; c < d; c += blockDim.x) {
        double run_ss = 0.0, run_s = 0.0, run_w = 0.0;
        i32 j = beg;
        i64 bi = 0;
        constexpr int NB = 4; // blocks of SB rows requested together: a workgroup is two waves at d = 128 and there are only a
                              // few hundred chunks, so the rows in flight per thread are what fills the memory pipeline
        for (; j + NB * SB - 1 < end; j += NB * SB, bi += NB) {
            double xv[NB * SB], wv[NB * SB];
#pragma unroll
            for (int q = 0; q < NB * SB; q++) {
                const i64 v = srows[j + q];
                wv[q] = vw[v];
                xv[q] = Xr[v * d + c];
            }
#pragma unroll
            for (int b = 0; b < NB; b++) {
#pragma unroll
                for (int q = 0; q < SB; q++) {
                    run_ss += wv[b * SB + q] * (xv[b * SB + q] * xv[b * SB + q]);
                    run_s += wv[b * SB + q] * xv[b * SB + q];
                    run_w += wv[b * SB + q];
                }
                slots[(bi + b) * W + c] = run_ss;
                slots[(bi + b) * W + d + c] = run_s;
                if (c == 0) slots[(bi + b) * W + 2 * d] = run_w;
            }
        }
        for (; j + SB - 1 < end; j += SB, bi++) { // SB rows in flight
            double xv[SB], wv[SB];
#pragma unroll
            for (int q = 0; q < SB; q++) {
                const i64 v = srows[j + q];
                wv[q] = vw[v];
                xv[q] = Xr[v * d + c];
            }
#pragma unroll
            for (int q = 0; q < SB; q++) {
                run_ss += wv[q] * (xv[q] * xv[q]);
                run_s += wv[q] * xv[q];
                run_w += wv[q];
            }
            slots[bi * W + c] = run_ss;
            slots[bi * W + d + c] = run_s;
            if (c == 0) slots[bi * W + 2 * d] = run_w;
        }
        for (; j < end; j++) {
            const i64 v = srows[j];
            const double w = vw[v], xv = Xr[v * d + c];
            run_ss += w * (xv * xv);
            run_s += w * xv;
            run_w += w;
        }
        ctot[ch * W + c] = run_ss;
        ctot[ch * W + d + c] = run_s;
        if (c == 0) ctot[ch * W + 2 * d] = run_w;
    }
}
__global__ void scan_chunk_offsets_kernel(const double *__restrict__ ctot, const i32 *__restrict__ task_chunk_off, i64 W,
                                          double *__restrict__ coff) {
    const i64 t = blockIdx.x;
    const i32 c0 = task_chunk_off[t], c1 = task_chunk_off[t + 1];
    for (i64 c = threadIdx.x; c < W; c += blockDim.x) {
        double run = 0.0;
        for (i32 ch = c0; ch < c1; ch++) {
            coff[(i64)ch * W + c] = run;
            run += ctot[(i64)ch * W + c];
        }
    }
}
void k_sorted_prefix(cge_ctx *c, const double *Xr, const double *vw, const i32 *srows, const i32 *chunk_beg,
                     const i32 *chunk_end, i64 n_chunks, const i32 *task_chunk_off, i64 n_tasks, i64 d, double *ctot,
                     double *coff, double *prefix) {
    const i64 W = 2 * d + 1;
    const int bs = (int)std::min<i64>(1024, (W + 63) / 64 * 64), bs1 = (int)std::min<i64>(1024, (d + 63) / 64 * 64);
    ScopedKernelTimer t(c, "sorted_prefix");
    hipLaunchKernelGGL(scan_write_kernel, dim3((unsigned)n_chunks), dim3(bs1), 0, c->stream, Xr, vw, srows, chunk_beg,
                       chunk_end, d, W, ctot, prefix);
    hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3((unsigned)n_tasks), dim3(bs), 0, c->stream, ctot, task_chunk_off,
                       W, coff);
}

#define RR_SLOTS 8      // columns per lane: d <= 512
#define RR_MAXROUNDS 63 // rounds logged per task (log2 of the group size + a few)
__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// one wave per task.  meta[t] = {n_rounds, rc}; rounds[t][r] = {first rank, end rank, side (1 low / 2 high)};
// vals[t] = {vlow, vhigh} = -total_rss of the children (eps() for singletons is applied by the host).
template <int NS> // NS = columns per lane (d <= 64*NS): no work is issued for slots beyond the embedding width
__global__ __launch_bounds__(64) void rss_rounds_kernel(const double *__restrict__ Xr, const double *__restrict__ vw,
                                                        const i32 *__restrict__ srows, const double *__restrict__ zs,
                                                        const i32 *__restrict__ task_row_off,
                                                        const i32 *__restrict__ task_chunk_off,
                                                        const double *__restrict__ prefix,
                                                        const double *__restrict__ coff, i64 d,
                                                        i32 *__restrict__ meta, i32 *__restrict__ rounds,
                                                        double *__restrict__ vals, double *__restrict__ cmeans) {
    const i64 t = blockIdx.x, W = 2 * d + 1;
    const int lane = threadIdx.x;
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    const double *z = zs + o;
    const i64 tco = task_chunk_off[t];
    const double *Coff = coff + tco * W; // exclusive offsets of the task's chunks
    i32 *rlog = rounds + t * 3 * RR_MAXROUNDS;
    // seeds: rank 0 (arg-min) and rank k-1 (arg-max), exact terms as :169-170
    double rl_ss[NS], rl_s[NS], rh_ss[NS], rh_s[NS];
    const i64 v1 = srows[o], v2 = srows[o + k - 1];
    const double w1 = vw[v1], w2 = vw[v2];
    double rl_w = w1, rh_w = w2;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const i64 c = lane + 64 * s;
        rl_ss[s] = rl_s[s] = rh_ss[s] = rh_s[s] = 0.0;
        if (c < d) {
            const double x1 = Xr[v1 * d + c], x2 = Xr[v2 * d + c];
            rl_ss[s] = x1 * x1 * w1; rl_s[s] = x1 * w1;
            rh_ss[s] = x2 * x2 * w2; rh_s[s] = x2 * w2;
        }
    }
    // inclusive prefix P[r] for this lane's columns: the stored running sum at the end of the previous block of
    // the chunk, plus the rows of the partial block in scan order, plus the chunk offset (the same additions in the
    // same order as a full running sum)
    auto prefix_at = [&](i64 r, double (&ss)[NS], double (&s1)[NS], double &w) {
        constexpr int SB = CGE_PREFIX_STRIDE;
        const i64 chl = r / CGE_CHUNK_ROWS, cbeg = o + chl * CGE_CHUNK_ROWS, rin = r - chl * CGE_CHUNK_ROWS, bi = rin / SB;
        const double *Sp = prefix + (cbeg / SB + tco + chl + (bi > 0 ? bi - 1 : 0)) * W;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const i64 c = lane + 64 * s;
            ss[s] = (c < d && bi > 0) ? Sp[c] : 0.0;
            s1[s] = (c < d && bi > 0) ? Sp[d + c] : 0.0;
        }
        w = (bi > 0) ? Sp[2 * d] : 0.0;
        { // the <= SB-1 rows of the partial block: all loads first, then the additions in scan order
            double xr[SB][NS], wr[SB];
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const i64 j = bi * SB + u;
                const bool on = j <= rin;
                const i64 v = srows[cbeg + (on ? j : rin)];
                wr[u] = on ? vw[v] : 0.0;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const i64 c = lane + 64 * s;
                    xr[u][s] = (on && c < d) ? Xr[v * d + c] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < SB; u++) {
                if (bi * SB + u > rin) break; // uniform
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    ss[s] += wr[u] * (xr[u][s] * xr[u][s]);
                    s1[s] += wr[u] * xr[u][s];
                }
                w += wr[u];
            }
        }
        const double *Cp = Coff + chl * W;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const i64 c = lane + 64 * s;
            if (c < d) { ss[s] += Cp[c]; s1[s] += Cp[d + c]; }
        }
        w += Cp[2 * d];
    };
    // sum over the columns of wsse(base + add)
    auto fsum = [&](const double (&bss)[NS], const double (&bs1)[NS], double bw, const double (&ass)[NS],
                    const double (&as1)[NS], double aw) {
        double acc = 0.0;
        const double w = bw + aw;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const i64 c = lane + 64 * s;
            if (c < d) {
                const double ss = bss[s] + ass[s], s1 = bs1[s] + as1[s];
                acc += ss - s1 * s1 / w;
            }
        }
        return wave_allsum(acc);
    };
    auto median = [&](i64 a, i64 b) { // Statistics.median of z[a..b) (sorted)
        const i64 cnt = b - a;
        return (cnt & 1) ? z[a + cnt / 2] : z[a + cnt / 2 - 1] / 2.0 + z[a + cnt / 2] / 2.0;
    };
    double zero_ss[NS], zero_s[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) zero_ss[s] = zero_s[s] = 0.0;
    i64 ga = 1, gb = k - 1; // gray = ranks [ga, gb)
    int nr = 0, rc = 0;
    double med = median(0, k);
    double a_ss[NS], a_s[NS], b_ss[NS], b_s[NS], aw, bw;
    bool leftover = false;
    // The prefixes at the two ends of the gray range are carried from round to round (PA = P[ga-1], PB = P[gb-1]): a round
    // needs ONE new prefix, at its cut -- the same values subtracted in the same way as four prefix look-ups per round gave
    // (two of them were the same look-up, two the previous round's).  PK = P[k-1] for the children's totals.
    double PA_ss[NS], PA_s[NS], PA_w, PB_ss[NS], PB_s[NS], PB_w, PK_ss[NS], PK_s[NS], PK_w, PC_ss[NS], PC_s[NS], PC_w;
    prefix_at(ga - 1, PA_ss, PA_s, PA_w);
    prefix_at(gb - 1 > 0 ? gb - 1 : 0, PB_ss, PB_s, PB_w);
    prefix_at(k - 1, PK_ss, PK_s, PK_w);
    while (ga < gb) {
        // number of gray ranks with z < med (z ascending): a 64-way search, the lanes probing 64 places at once -- two or
        // three dependent loads instead of log2 of the range
        i64 lo = ga, hi = gb;
        while (hi - lo > 64) {
            const i64 step = (hi - lo + 63) / 64, idx = lo + (i64)lane * step;
            const bool below = idx < hi && z[idx < hi ? idx : lo] < med;
            const int cnt = __popcll(__ballot(below)); // probes 0 .. cnt-1 are below the median, probe cnt is not (or beyond hi)
            if (cnt == 0) { hi = lo; break; }
            const i64 nlo = lo + (i64)(cnt - 1) * step + 1, nhi = lo + (i64)cnt * step;
            hi = nhi < hi ? nhi : hi;
            lo = nlo;
        }
        if (hi > lo) {
            const i64 idx = lo + lane;
            const bool below = idx < hi && z[idx < hi ? idx : lo] < med;
            lo += __popcll(__ballot(below));
        }
        const i64 cut = lo; // t1 = [ga, cut), t2 = [cut, gb)
        if (cut > ga && cut < gb) prefix_at(cut - 1, PC_ss, PC_s, PC_w);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (cut <= ga) { a_ss[s] = 0.0; a_s[s] = 0.0; b_ss[s] = PB_ss[s] - PA_ss[s]; b_s[s] = PB_s[s] - PA_s[s]; }
            else if (cut >= gb) { a_ss[s] = PB_ss[s] - PA_ss[s]; a_s[s] = PB_s[s] - PA_s[s]; b_ss[s] = 0.0; b_s[s] = 0.0; }
            else { a_ss[s] = PC_ss[s] - PA_ss[s]; a_s[s] = PC_s[s] - PA_s[s]; b_ss[s] = PB_ss[s] - PC_ss[s]; b_s[s] = PB_s[s] - PC_s[s]; }
        }
        if (cut <= ga) { aw = 0.0; bw = PB_w - PA_w; }
        else if (cut >= gb) { aw = PB_w - PA_w; bw = 0.0; }
        else { aw = PC_w - PA_w; bw = PB_w - PC_w; }
        const double f1 = fsum(rl_ss, rl_s, rl_w, a_ss, a_s, aw);
        const double f2 = fsum(rh_ss, rh_s, rh_w, b_ss, b_s, bw);
        if (nr >= RR_MAXROUNDS - 1) { rc = 1; break; }
        if (f1 < f2) {
            if (cut == ga) { leftover = true; break; }
#pragma unroll
            for (int s = 0; s < NS; s++) { rl_ss[s] += a_ss[s]; rl_s[s] += a_s[s]; }
            rl_w += aw;
            if (lane == 0) { rlog[3 * nr] = (i32)ga; rlog[3 * nr + 1] = (i32)cut; rlog[3 * nr + 2] = 1; }
            nr++;
            ga = cut;
            if (cut < gb) {
#pragma unroll
                for (int s = 0; s < NS; s++) { PA_ss[s] = PC_ss[s]; PA_s[s] = PC_s[s]; }
                PA_w = PC_w;
            } else { // the whole gray range went low: ga == gb, the loop ends
#pragma unroll
                for (int s = 0; s < NS; s++) { PA_ss[s] = PB_ss[s]; PA_s[s] = PB_s[s]; }
                PA_w = PB_w;
            }
        } else {
            if (cut == gb) { leftover = true; break; }
#pragma unroll
            for (int s = 0; s < NS; s++) { rh_ss[s] += b_ss[s]; rh_s[s] += b_s[s]; }
            rh_w += bw;
            if (lane == 0) { rlog[3 * nr] = (i32)cut; rlog[3 * nr + 1] = (i32)gb; rlog[3 * nr + 2] = 2; }
            nr++;
            gb = cut;
            if (cut > ga) {
#pragma unroll
                for (int s = 0; s < NS; s++) { PB_ss[s] = PC_ss[s]; PB_s[s] = PC_s[s]; }
                PB_w = PC_w;
            } else { // the whole gray range went high: gb == ga
#pragma unroll
                for (int s = 0; s < NS; s++) { PB_ss[s] = PA_ss[s]; PB_s[s] = PA_s[s]; }
                PB_w = PA_w;
            }
        }
        if (ga < gb) med = median(ga, gb);
    }
    if (leftover && rc == 0) { // :200-208
#pragma unroll
        for (int s = 0; s < NS; s++) { a_ss[s] = PB_ss[s] - PA_ss[s]; a_s[s] = PB_s[s] - PA_s[s]; }
        aw = PB_w - PA_w;
        const double fa1 = fsum(rl_ss, rl_s, rl_w, a_ss, a_s, aw), fa2 = fsum(rh_ss, rh_s, rh_w, zero_ss, zero_s, 0.0);
        const double fb1 = fsum(rl_ss, rl_s, rl_w, zero_ss, zero_s, 0.0), fb2 = fsum(rh_ss, rh_s, rh_w, a_ss, a_s, aw);
        const int side = (fmax(fa1, fa2) < fmax(fb1, fb2)) ? 1 : 2;
        if (lane == 0) { rlog[3 * nr] = (i32)ga; rlog[3 * nr + 1] = (i32)gb; rlog[3 * nr + 2] = side; }
        nr++;
        if (side == 1) { // ga = gb
            ga = gb;
#pragma unroll
            for (int s = 0; s < NS; s++) { PA_ss[s] = PB_ss[s]; PA_s[s] = PB_s[s]; }
            PA_w = PB_w;
        } else
            gb = ga;
    }
    // children: low = ranks [0, ga), high = ranks [ga, k); total_rss from the prefix sums: P[ga-1] and P[k-1] - P[ga-1]
#pragma unroll
    for (int s = 0; s < NS; s++) { a_ss[s] = PA_ss[s]; a_s[s] = PA_s[s]; b_ss[s] = PK_ss[s] - PA_ss[s]; b_s[s] = PK_s[s] - PA_s[s]; }
    aw = PA_w;
    bw = PK_w - PA_w;
    const double vlow = -fsum(zero_ss, zero_s, 0.0, a_ss, a_s, aw), vhigh = -fsum(zero_ss, zero_s, 0.0, b_ss, b_s, bw);
    // the children's weighted means (matrix_w_mean, :71-81) are the same column sums: sum w x / sum w
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const i64 c = lane + 64 * s;
        if (c < d) {
            cmeans[(2 * t) * d + c] = a_s[s] / aw;
            cmeans[(2 * t + 1) * d + c] = b_s[s] / bw;
        }
    }
    if (lane == 0) {
        meta[2 * t] = nr;
        meta[2 * t + 1] = rc;
        vals[2 * t] = vlow;
        vals[2 * t + 1] = vhigh;
    }
}
void k_rss_rounds(cge_ctx *c, const double *Xr, const double *vw, const i32 *srows, const double *zs,
                  const i32 *task_row_off, const i32 *task_chunk_off, const double *prefix, const double *coff,
                  i64 n_tasks, i64 d, i32 *meta, i32 *rounds, double *vals, double *cmeans) {
    if (d > 64 * RR_SLOTS) CGE_THROW(CGE_E_ARG, "embedding dimension %lld > %d not supported", (long long)d, 64 * RR_SLOTS);
    ScopedKernelTimer t(c, "rss_rounds");
    const dim3 grid((unsigned)n_tasks), block(64);
#define CGE_RR_LAUNCH(NS)                                                                                              \
    hipLaunchKernelGGL((rss_rounds_kernel<NS>), grid, block, 0, c->stream, Xr, vw, srows, zs, task_row_off, task_chunk_off, \
                       prefix, coff, d, meta, rounds, vals, cmeans)
    if (d <= 64) CGE_RR_LAUNCH(1);
    else if (d <= 128) CGE_RR_LAUNCH(2);
    else if (d <= 256) CGE_RR_LAUNCH(4);
    else CGE_RR_LAUNCH(8);
#undef CGE_RR_LAUNCH
}

// ------------------------------------------------------------------------------------------------
// split_cluster_rss2 (src/landmarks.jl:92-152) on the device, one wave per task, rows in ascending-z order (srows):
// the two-pointer walk absorbs one row per step into the low or the high WSSE triple -- the reference's own
// additions in the reference's order (:105-117) -- then the boundary adjustment loops (:118-150).  Lane l holds the
// columns l + 64 s.  The next candidate row of either side is loaded ahead of the decision.
// meta[t] = {lo, hi}: low = ranks [0, lo], high = ranks [hi, k); vals / cmeans as in the rss kernel.
template <int NS> // NS = columns per lane (d <= 64*NS)
__global__ __launch_bounds__(64) void rss2_walk_kernel(const double *__restrict__ Xr, const double *__restrict__ vw,
                                                       const i32 *__restrict__ srows,
                                                       const i32 *__restrict__ task_row_off, i64 d,
                                                       i32 *__restrict__ meta, double *__restrict__ vals,
                                                       double *__restrict__ cmeans) {
    // The walk consumes WR-row windows on either side.  A window is filled with all its row loads in flight together,
    // and the RSS after absorbing each of its rows -- sum(wsse, rss_low) of :106 for the next WR low ranks, likewise on
    // the high side -- is computed right away: the WR running triples are a short dependent chain, the WR
    // divide-and-reduce evaluations behind them are independent and overlap.  The walk itself then only compares two
    // precomputed numbers per step.
    constexpr int WR = NS <= 2 ? 16 : NS <= 4 ? 8 : 4;
    __shared__ __attribute__((aligned(16))) double win[2][WR][NS][64]; // rows of the two windows
    __shared__ double wwin[2][WR], fwin[2][WR];                          // their weights, the RSS after each row
    const i64 t = blockIdx.x;
    const int lane = threadIdx.x;
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    const i32 *p = srows + o;
    double b_ss[2][NS], b_s[2][NS], b_w[2]; // WSSE triples at the start of either window
    auto wsse_part = [&](const double (&ss)[NS], const double (&s1)[NS], double w) { // this lane's share of sum(wsse, r)
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) acc += ss[s] - s1[s] * s1[s] / w; // padded columns are 0
        return acc;
    };
    // window of `side` <- ranks first, first+step, ... (step +1 low / -1 high); fwin[side][q] = RSS with slots 0..q absorbed
    auto fill = [&](const int side, i64 first, i64 step) {
        constexpr int U = NS <= 1 ? 16 : NS <= 2 ? 8 : NS <= 4 ? 4 : 2;
        const i64 rankq = first + step * lane;
        const bool okq = lane < WR && rankq >= 0 && rankq < k;
        const int vq = okq ? p[rankq] : p[0]; // slots past the end: a valid row with weight 0
        if (lane < WR) wwin[side][lane] = okq ? vw[vq] : 0.0;
        for (int q0 = 0; q0 < WR; q0 += U) {
            double x[U][NS];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const i64 v = __shfl(vq, (q0 + u) & 63);
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const i64 c = lane + 64 * s;
                    x[u][s] = (c < d) ? Xr[v * d + c] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (q0 + u < WR) {
#pragma unroll
                    for (int s = 0; s < NS; s++) win[side][q0 + u][s][lane] = x[u][s];
                }
        }
        __builtin_amdgcn_wave_barrier();
        double r_ss[NS], r_s[NS], r_w = b_w[side], part[WR];
#pragma unroll
        for (int s = 0; s < NS; s++) { r_ss[s] = b_ss[side][s]; r_s[s] = b_s[side][s]; }
#pragma unroll
        for (int q = 0; q < WR; q++) {
            const double w = wwin[side][q];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const double xv = win[side][q][s][lane];
                r_ss[s] += w * (xv * xv);
                r_s[s] += w * xv;
            }
            r_w += w;
            part[q] = wsse_part(r_ss, r_s, r_w);
        }
#pragma unroll
        for (int q = 0; q < WR; q++) {
            const double f = wave_allsum(part[q]);
            if (lane == 0) fwin[side][q] = f;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // b[side] += the first `cnt` rows of the window
    auto absorb = [&](const int side, int cnt) {
        for (int q = 0; q < cnt; q++) {
            const double w = wwin[side][q];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const double xv = win[side][q][s][lane];
                b_ss[side][s] += w * (xv * xv);
                b_s[side][s] += w * xv;
            }
            b_w[side] += w;
        }
    };
    auto term = [&](i64 rank, double (&ss)[NS], double (&s1)[NS], double &w) { // WSSE(m[p[rank], :], w[p[rank]])
        const i64 v = p[rank];
        w = vw[v];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const i64 c = lane + 64 * s;
            const double x = (c < d) ? Xr[v * d + c] : 0.0;
            ss[s] = w * (x * x);
            s1[s] = w * x;
        }
    };
    auto fsum = [&](const double (&ss)[NS], const double (&s1)[NS], double w) { return wave_allsum(wsse_part(ss, s1, w)); };
    term(0, b_ss[0], b_s[0], b_w[0]);
    term(k - 1, b_ss[1], b_s[1], b_w[1]);
    double fl = fsum(b_ss[0], b_s[0], b_w[0]), fh = fsum(b_ss[1], b_s[1], b_w[1]);
    i64 lo = 0, hi = k - 1;
    int il = 0, ih = 0; // slots of the windows already absorbed
    fill(0, lo + 1, 1);
    fill(1, hi - 1, -1);
    while (lo + 1 < hi) {
        if (fl < fh) {
            lo++;
            fl = fwin[0][il++];
            if (il == WR && lo + 1 < hi) { absorb(0, WR); il = 0; fill(0, lo + 1, 1); }
        } else {
            hi--;
            fh = fwin[1][ih++];
            if (ih == WR && lo + 1 < hi) { absorb(1, WR); ih = 0; fill(1, hi - 1, -1); }
        }
    }
    absorb(0, il);
    absorb(1, ih);
    double l_ss[NS], l_s[NS], h_ss[NS], h_s[NS], l_w = b_w[0], h_w = b_w[1];
#pragma unroll
    for (int s = 0; s < NS; s++) { l_ss[s] = b_ss[0][s]; l_s[s] = b_s[0][s]; h_ss[s] = b_ss[1][s]; h_s[s] = b_s[1][s]; }
    // boundary adjustment: move the last low row up, or the first high row down, while the larger RSS shrinks
    bool moved_low = false;
    double a_ss[NS], a_s[NS], t_ss[NS], t_s[NS], u_ss[NS], u_s[NS], aw;
    while (lo > 0) {
        term(lo, a_ss, a_s, aw);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            t_ss[s] = l_ss[s] - a_ss[s]; t_s[s] = l_s[s] - a_s[s];
            u_ss[s] = h_ss[s] + a_ss[s]; u_s[s] = h_s[s] + a_s[s];
        }
        const double tw = l_w - aw, uw = h_w + aw;
        const double ft = fsum(t_ss, t_s, tw), fu = fsum(u_ss, u_s, uw);
        if (fmax(ft, fu) < fmax(fl, fh)) {
            moved_low = true;
            lo--; hi--;
#pragma unroll
            for (int s = 0; s < NS; s++) { l_ss[s] = t_ss[s]; l_s[s] = t_s[s]; h_ss[s] = u_ss[s]; h_s[s] = u_s[s]; }
            l_w = tw; h_w = uw; fl = ft; fh = fu;
        } else
            break;
    }
    if (!moved_low)
        while (hi < k - 1) {
            term(hi, a_ss, a_s, aw);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                t_ss[s] = l_ss[s] + a_ss[s]; t_s[s] = l_s[s] + a_s[s];
                u_ss[s] = h_ss[s] - a_ss[s]; u_s[s] = h_s[s] - a_s[s];
            }
            const double tw = l_w + aw, uw = h_w - aw;
            const double ft = fsum(t_ss, t_s, tw), fu = fsum(u_ss, u_s, uw);
            if (fmax(ft, fu) < fmax(fl, fh)) {
                lo++; hi++;
#pragma unroll
                for (int s = 0; s < NS; s++) { l_ss[s] = t_ss[s]; l_s[s] = t_s[s]; h_ss[s] = u_ss[s]; h_s[s] = u_s[s]; }
                l_w = tw; h_w = uw; fl = ft; fh = fu;
            } else
                break;
        }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const i64 c = lane + 64 * s;
        if (c < d) {
            cmeans[(2 * t) * d + c] = l_s[s] / l_w;
            cmeans[(2 * t + 1) * d + c] = h_s[s] / h_w;
        }
    }
    if (lane == 0) {
        meta[2 * t] = (i32)lo;
        meta[2 * t + 1] = (i32)hi;
        vals[2 * t] = -fl;
        vals[2 * t + 1] = -fh;
    }
}
void k_rss2_walk_one_kernel(cge_ctx *c, const double *Xr, const double *vw, const i32 *srows, const i32 *task_row_off, i64 n_tasks, i64 d,
                 i32 *meta, double *vals, double *cmeans) {
    if (d > 64 * RR_SLOTS) CGE_THROW(CGE_E_ARG, "embedding dimension %lld > %d not supported", (long long)d, 64 * RR_SLOTS);
    ScopedKernelTimer t(c, "rss2_walk");
    const int ns = d <= 64 ? 1 : d <= 128 ? 2 : d <= 256 ? 4 : 8;
    const dim3 grid((unsigned)n_tasks), block(64);
#define CGE_RSS2_LAUNCH(NS) \
    hipLaunchKernelGGL((rss2_walk_kernel<NS>), grid, block, 0, c->stream, Xr, vw, srows, task_row_off, d, meta, vals, cmeans)
    if (ns == 1) CGE_RSS2_LAUNCH(1);
    else if (ns == 2) CGE_RSS2_LAUNCH(2);
    else if (ns == 4) CGE_RSS2_LAUNCH(4);
    else CGE_RSS2_LAUNCH(8);
#undef CGE_RSS2_LAUNCH
}

// ---- split_cluster_rss2 in two kernels ---------------------------------------------------------------------------------
// The walk of :105-117 compares sum(wsse, rss_low) with sum(wsse, rss_high); the first depends only on how many rows
// the low side has absorbed (its additions run in rank order whatever the high side does), the second only on the high
// side.  So  FL[i] = RSS of ranks 0..i  and  FH[j] = RSS of ranks k-1-j..k-1  can be produced for ALL i, j by two
// independent chains with the reference's own additions in the reference's order (rss2_chain_lds_kernel: one workgroup per
// task and direction; every wave runs the cheap chain of additions over all rows, but pays the divisions and the
// 64-lane reduction only for every fourth block of 16 rows), and the walk becomes a scalar merge of two arrays
// (rss2_merge_kernel).  Same bits as rss2_walk_kernel: the same additions in the same order, the same reduction tree.
#define R2_BR 16 // rows per block of the chain

// ---- the chains with the rows SHARED through LDS (round 3) ------------------------------------------------------------------
// (Rounds 2-3 let each of the four waves load every row of the group itself; removed in round 5.)  The waves load DIFFERENT
// blocks: in a round of four blocks (64 rows) wave w fetches block 4r + w into registers (two rounds ahead, two
// register sets), parks it in LDS, and after one barrier every wave runs its chain over the 64 rows from LDS -- the same
// additions in the same order, the divisions and tree sums of a block by the wave that owns it, as before.  Same bits.
// Config 2: 4.24 -> 3.91 ms per step (the loads were not the bound: a wave issues an fp64 instruction per ~5.5 ns when it is
// alone on its SIMD, and a row costs ~15 of them in the chain plus ~40 per IEEE division in its owner's quarter).
template <int NS, int NW>
__global__ __launch_bounds__(64 * NW) void rss2_chain_lds_kernel(const double *__restrict__ Xr, const double *__restrict__ vw,
                                                             const i32 *__restrict__ srows,
                                                             const i32 *__restrict__ task_row_off, i64 d, i64 R,
                                                             double *__restrict__ F /* [2][R] */,
                                                             double *__restrict__ ck /* [2][slots][2 NS 64 + 64] */, i64 slots) {
    extern __shared__ __attribute__((aligned(16))) double r2lds[];
    constexpr int XW = NS * 64; // doubles per row
    constexpr int RR = 16 * NW;           // rows per round
    constexpr int NB = (NW == 4 && NS == 1) ? 2 : 1; // LDS buffers of a round's rows (one when eight waves / 128 columns need the room)
    double(*tile)[R2_BR][65] = reinterpret_cast<double(*)[R2_BR][65]>(r2lds);          // [NW][16][65]
    double *xs = r2lds + NW * R2_BR * 65;                                               // [NB][RR][XW]
    double *wsh = xs + NB * RR * XW;                                                    // [NB][RR]
    const i64 t = blockIdx.x;
    const int dir = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    const i32 *p = srows + o;
    const i64 nblk = (k + R2_BR - 1) / R2_BR, nround = (nblk + NW - 1) / NW, slot0 = o / R2_BR + t;
    double *Fo = F + (i64)dir * R + o;
    double *cko = ck + ((i64)dir * slots + slot0) * (2 * NS * 64 + 64);
    double ss[NS], s1[NS], wacc = 0.0;
#pragma unroll
    for (int s = 0; s < NS; s++) ss[s] = s1[s] = 0.0;
    if (k <= 0) return;
    // This wave's block of round r: rows RR r + 16 wave + (0..15); lane u < 16 keeps the id / weight of row u.  The row ids
    // are requested a round BEFORE the rows themselves: asked for together, every round waited out a memory round trip
    // between the two (the ids, then the rows), which was a third of what a row cost.
    const int p0 = p[0];
    auto request_ids = [&](i64 r) -> int { // -1: a slot past the end
        const i64 q = r * RR + 16 * wave + lane;
        return (lane < R2_BR && q < k) ? p[dir ? k - 1 - q : q] : -1;
    };
    auto load_rows = [&](const int vid, double (&x)[R2_BR][NS], double &wl) {
        const int vq = vid >= 0 ? vid : p0; // slots past the end: a valid row with weight 0
        wl = vid >= 0 ? vw[vq] : 0.0;
#pragma unroll
        for (int u = 0; u < R2_BR; u++) {
            const i64 v = __builtin_amdgcn_readlane(vq, u);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const i64 c = lane + 64 * s;
                x[u][s] = (c < d) ? Xr[v * d + c] : 0.0;
            }
        }
    };
    auto park = [&](int buf, const double (&x)[R2_BR][NS], const double wl) {
        double *xb = xs + (size_t)buf * RR * XW + (size_t)(16 * wave) * XW;
#pragma unroll
        for (int u = 0; u < R2_BR; u++)
#pragma unroll
            for (int s = 0; s < NS; s++) xb[u * XW + lane + 64 * s] = x[u][s];
        if (lane < R2_BR) wsh[buf * RR + 16 * wave + lane] = wl;
    };
    // A block's 16 rows and weights come out of LDS in ONE batch, a block ahead of their use: read where they are used, every
    // row waited for its own LDS round trip (the tile stores between the rows keep the compiler from hoisting the reads), and
    // that latency -- not the additions, the divisions or the global loads -- was what a row of the longest group cost.
    auto lds_fetch = [&](const double *xb, const double *wb, double (&x)[R2_BR][NS], double &wl) {
#pragma unroll
        for (int q = 0; q < R2_BR; q++)
#pragma unroll
            for (int s = 0; s < NS; s++) x[q][s] = xb[q * XW + lane + 64 * s];
        wl = wb[lane & (R2_BR - 1)];
    };
    // Block b, its 16 rows and weights in registers.  The same operations as the walk, in an order that lets them overlap: the
    // 48 products first (independent of the chain), then the chain as bare additions, and in the owner's block the running
    // sums of all 16 rows are kept so that the 16 divisions -- independent of one another -- follow as one straight-line
    // stretch.  Written row by row behind a branch per row, every instruction waited for the one before it.
    auto process = [&](i64 b, const double (&x)[R2_BR][NS], const double wl) {
        const bool mine = (int)(b % NW) == wave;
        double p2[R2_BR][NS], p1[R2_BR][NS], wq[R2_BR];
#pragma unroll
        for (int q = 0; q < R2_BR; q++) {
            const double w = lane_value(wl, q);
            wq[q] = w;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const double xv = x[q][s];
                p2[q][s] = w * (xv * xv);
                p1[q][s] = w * xv;
            }
        }
        if (!mine) {
#pragma unroll
            for (int q = 0; q < R2_BR; q++) {
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    ss[s] += p2[q][s];
                    s1[s] += p1[q][s];
                }
                wacc += wq[q];
            }
            return;
        }
        // checkpoint: the triple before this block
#pragma unroll
        for (int s = 0; s < NS; s++) {
            cko[b * (2 * NS * 64 + 64) + (2 * s) * 64 + lane] = ss[s];
            cko[b * (2 * NS * 64 + 64) + (2 * s + 1) * 64 + lane] = s1[s];
        }
        if (lane == 0) cko[b * (2 * NS * 64 + 64) + 2 * NS * 64] = wacc;
#pragma unroll
        for (int q = 0; q < R2_BR; q++) { // the running sums after every row, in place of the products
#pragma unroll
            for (int s = 0; s < NS; s++) {
                ss[s] += p2[q][s];
                s1[s] += p1[q][s];
                p2[q][s] = ss[s];
                p1[q][s] = s1[s];
            }
            wacc += wq[q];
            wq[q] = wacc;
        }
#pragma unroll
        for (int q = 0; q < R2_BR; q++) {
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < NS; s++) acc += p2[q][s] - p1[q][s] * p1[q][s] / wq[q]; // padded columns are 0
            tile[wave][q][lane] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        { // wave_allsum's tree for 16 rows at once: lane (g, row) adds adjacent pairs inside group g of 16 lanes of `row`,
          // then ((R0 + R1) + R2) + R3 on the lanes of g = 0
            const int row = lane & (R2_BR - 1), g = lane >> 4;
            const double *u = tile[wave][row] + 16 * g;
            const double a0 = (u[0] + u[1]) + (u[2] + u[3]), a1 = (u[4] + u[5]) + (u[6] + u[7]);
            const double a2 = (u[8] + u[9]) + (u[10] + u[11]), a3 = (u[12] + u[13]) + (u[14] + u[15]);
            const double r = (a0 + a1) + (a2 + a3);
            const double r1 = __shfl(r, row + 16), r2 = __shfl(r, row + 32), r3 = __shfl(r, row + 48);
            const i64 q = b * R2_BR + row;
            if (g == 0 && q < k) Fo[q] = ((r + r1) + r2) + r3;
        }
        __builtin_amdgcn_wave_barrier();
    };
    double xa[R2_BR][NS], xbq[R2_BR][NS], wa = 0.0, wbq = 0.0;
    double y0[R2_BR][NS], y1[R2_BR][NS], u0 = 0.0, u1 = 0.0; // the blocks being added, out of LDS
    int vqp; // the ids of the round whose rows are requested next
    {
        const int v0 = request_ids(0), v1 = request_ids(1);
        vqp = request_ids(2);
        load_rows(v0, xa, wa);
        if (1 < nround) load_rows(v1, xbq, wbq);
    }
    for (i64 r = 0; r < nround; r++) {
        const int buf = (NB == 2) ? (int)(r & 1) : 0;
        if (NB == 1 && r > 0) __syncthreads(); // everybody is done with the round that is in the buffer
        int vqn; // (requested behind the park: the park waits for every load in flight, and those are a round old by then)
        if ((r & 1) == 0) {
            park(buf, xa, wa);
            vqn = request_ids(r + 3);
            if (r + 2 < nround) load_rows(vqp, xa, wa);
        } else {
            park(buf, xbq, wbq);
            vqn = request_ids(r + 3);
            if (r + 2 < nround) load_rows(vqp, xbq, wbq);
        }
        vqp = vqn;
        __syncthreads(); // the round's rows are in LDS (and everybody is done with the round that used this buffer before)
        const double *xr = xs + (size_t)buf * RR * XW;
        const double *wr = wsh + buf * RR;
        lds_fetch(xr, wr, y0, u0);
#pragma unroll 1
        for (int bb = 0; bb < NW; bb += 2) {
            const i64 b = NW * r + bb;
            if (b >= nblk) break;
            lds_fetch(xr + (size_t)(16 * (bb + 1)) * XW, wr + 16 * (bb + 1), y1, u1);
            process(b, y0, u0);
            if (b + 1 >= nblk) break;
            if (bb + 2 < NW) lds_fetch(xr + (size_t)(16 * (bb + 2)) * XW, wr + 16 * (bb + 2), y0, u0);
            process(b + 1, y1, u1);
        }
    }
}

// (Rounds 3-4 also had the chains in two steps -- the additions by one wave per (group, direction), the divisions by everybody --
// and as streams of precomputed products; both gave the same bits and were measured slower (profiles/
// r03 / r04 notes): removed in round 5.)
template <int NS>
__global__ __launch_bounds__(64) void rss2_merge_kernel(const double *__restrict__ Xr, const double *__restrict__ vw,
                                                        const i32 *__restrict__ srows,
                                                        const i32 *__restrict__ task_row_off, i64 d, i64 R,
                                                        const double *__restrict__ F, const double *__restrict__ ck,
                                                        i64 slots, i32 *__restrict__ meta, double *__restrict__ vals,
                                                        double *__restrict__ cmeans) {
    const i64 t = blockIdx.x;
    const int lane = threadIdx.x;
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    const i32 *p = srows + o;
    const double *FL = F + o, *FH = F + R + o; // FL[i]: low side holds ranks 0..i; FH[j]: high side holds ranks k-1-j..k-1
    // ---- the walk: i + j goes from 0 to k - 2, low advances when FL[i] < FH[j] ------------------------------------------
    // Both arrays non-decreasing (the usual case): the walk is the merge of two sorted lists, its end point a bisection.
    bool mono = true;
    for (i64 q = lane; q + 1 < k; q += 64) mono = mono && FL[q] <= FL[q + 1] && FH[q] <= FH[q + 1];
    mono = __all(mono);
    const i64 S = k - 2;
    i64 li; // rows the low side has taken beyond rank 0 when the walk ends
    if (mono) {
        // smallest i in [0, S] such that the walk does NOT take an (i+1)-th low step before its j-th high step, j = S - i:
        // low step i+1 (comparing FL[i]) comes before high step j (comparing FH[j-1]) iff FL[i] < FH[j-1]
        i64 lo_ = 0, hi_ = S;
        while (lo_ < hi_) {
            const i64 mid = (lo_ + hi_) >> 1, j = S - mid;
            if (j >= 1 && FL[mid] < FH[j - 1]) lo_ = mid + 1; else hi_ = mid;
        }
        li = lo_;
    } else { // the reference's loop as it stands: scalar, the two arrays in 64-entry register windows (next one in flight)
        int i = 0, j = 0, bi = 0, bj = 0;
        const int Sn = (int)S, kn = (int)k;
        double cl = FL[min(kn - 1, lane)], ch = FH[min(kn - 1, lane)];
        double nl = FL[min(kn - 1, 64 + lane)], nh = FH[min(kn - 1, 64 + lane)];
        double fl0 = lane_value(cl, 0), fh0 = lane_value(ch, 0);
        while (i + j < Sn) {
            if (fl0 < fh0) {
                i++;
                if (i - bi == 64) { cl = nl; bi += 64; nl = FL[min(kn - 1, bi + 64 + lane)]; }
                fl0 = lane_value(cl, __builtin_amdgcn_readfirstlane(i - bi));
            } else {
                j++;
                if (j - bj == 64) { ch = nh; bj += 64; nh = FH[min(kn - 1, bj + 64 + lane)]; }
                fh0 = lane_value(ch, __builtin_amdgcn_readfirstlane(j - bj));
            }
        }
        li = i;
    }
    i64 lo = li, hi = k - 1 - (S - li);
    double fl = FL[lo], fh = FH[k - 1 - hi];
    // ---- the two WSSE triples at the meeting point: checkpoint of the block + its first rows again, in order ----------------
    const i64 slot0 = o / R2_BR + t, stride = 2 * NS * 64 + 64;
    double l_ss[NS], l_s[NS], h_ss[NS], h_s[NS], l_w, h_w;
    auto restore = [&](int dir, i64 q_last, double (&ss)[NS], double (&s1)[NS], double &w) {
        const i64 b = q_last / R2_BR;
        const double *c0 = ck + ((i64)dir * slots + slot0 + b) * stride;
#pragma unroll
        for (int s = 0; s < NS; s++) { ss[s] = c0[(2 * s) * 64 + lane]; s1[s] = c0[(2 * s + 1) * 64 + lane]; }
        w = c0[2 * NS * 64];
        for (i64 q = b * R2_BR; q <= q_last; q++) {
            const i64 v = p[dir ? k - 1 - q : q];
            const double wv = vw[v];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const i64 c = lane + 64 * s;
                const double xv = (c < d) ? Xr[v * d + c] : 0.0;
                ss[s] += wv * (xv * xv);
                s1[s] += wv * xv;
            }
            w += wv;
        }
    };
    restore(0, lo, l_ss, l_s, l_w);
    restore(1, k - 1 - hi, h_ss, h_s, h_w);
    auto term = [&](i64 rank, double (&ss)[NS], double (&s1)[NS], double &w) { // WSSE(m[p[rank], :], w[p[rank]])
        const i64 v = p[rank];
        w = vw[v];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const i64 c = lane + 64 * s;
            const double x = (c < d) ? Xr[v * d + c] : 0.0;
            ss[s] = w * (x * x);
            s1[s] = w * x;
        }
    };
    auto fsum = [&](const double (&ss)[NS], const double (&s1)[NS], double w) {
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) acc += ss[s] - s1[s] * s1[s] / w;
        return wave_allsum(acc);
    };
    // boundary adjustment (:118-150): move the last low row up, or the first high row down, while the larger RSS shrinks
    bool moved_low = false;
    double a_ss[NS], a_s[NS], t_ss[NS], t_s[NS], u_ss[NS], u_s[NS], aw;
    while (lo > 0) {
        term(lo, a_ss, a_s, aw);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            t_ss[s] = l_ss[s] - a_ss[s]; t_s[s] = l_s[s] - a_s[s];
            u_ss[s] = h_ss[s] + a_ss[s]; u_s[s] = h_s[s] + a_s[s];
        }
        const double tw = l_w - aw, uw = h_w + aw;
        const double ft = fsum(t_ss, t_s, tw), fu = fsum(u_ss, u_s, uw);
        if (fmax(ft, fu) < fmax(fl, fh)) {
            moved_low = true;
            lo--; hi--;
#pragma unroll
            for (int s = 0; s < NS; s++) { l_ss[s] = t_ss[s]; l_s[s] = t_s[s]; h_ss[s] = u_ss[s]; h_s[s] = u_s[s]; }
            l_w = tw; h_w = uw; fl = ft; fh = fu;
        } else
            break;
    }
    if (!moved_low)
        while (hi < k - 1) {
            term(hi, a_ss, a_s, aw);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                t_ss[s] = l_ss[s] + a_ss[s]; t_s[s] = l_s[s] + a_s[s];
                u_ss[s] = h_ss[s] - a_ss[s]; u_s[s] = h_s[s] - a_s[s];
            }
            const double tw = l_w + aw, uw = h_w - aw;
            const double ft = fsum(t_ss, t_s, tw), fu = fsum(u_ss, u_s, uw);
            if (fmax(ft, fu) < fmax(fl, fh)) {
                lo++; hi++;
#pragma unroll
                for (int s = 0; s < NS; s++) { l_ss[s] = t_ss[s]; l_s[s] = t_s[s]; h_ss[s] = u_ss[s]; h_s[s] = u_s[s]; }
                l_w = tw; h_w = uw; fl = ft; fh = fu;
            } else
                break;
        }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const i64 c = lane + 64 * s;
        if (c < d) {
            cmeans[(2 * t) * d + c] = l_s[s] / l_w;
            cmeans[(2 * t + 1) * d + c] = h_s[s] / h_w;
        }
    }
    if (lane == 0) {
        meta[2 * t] = (i32)lo;
        meta[2 * t + 1] = (i32)hi;
        vals[2 * t] = -fl;
        vals[2 * t + 1] = -fh;
    }
}

void k_rss2_walk(cge_ctx *c, const double *Xr, const double *vw, const i32 *srows, const i32 *task_row_off, i64 n_tasks, i64 d,
                 i32 *meta, double *vals, double *cmeans) {
    if (d > 64 * RR_SLOTS) CGE_THROW(CGE_E_ARG, "embedding dimension %lld > %d not supported", (long long)d, 64 * RR_SLOTS);
    const int ns0 = d <= 64 ? 1 : d <= 128 ? 2 : d <= 256 ? 4 : 8;
    if (ns0 <= 2 && c->r2_rows > 0) { // chain + merge (the register blocks of the chain allow d <= 128)
        ScopedKernelTimer t(c, "rss2_walk");
        const i64 R = c->r2_rows, slots = R / R2_BR + n_tasks + 2, stride = 2 * ns0 * 64 + 64;
        c->r2_F.ensure((size_t)2 * R);
        c->r2_ck.ensure((size_t)2 * slots * stride);
        const dim3 gridA((unsigned)n_tasks, 2);
        // the four waves of a chain share the rows through LDS; 64 < d <= 128: one LDS buffer of rows, an extra barrier per round
        if (ns0 == 2) {
            const size_t lds = (size_t)(4 * R2_BR * 65 + 64 * 128 + 64) * sizeof(double);
            cge_allow_lds((const void *)rss2_chain_lds_kernel<2, 4>, 160 * 1024);
            hipLaunchKernelGGL((rss2_chain_lds_kernel<2, 4>), gridA, dim3(256), lds, c->stream, Xr, vw, srows, task_row_off, d, R,
                               c->r2_F.p, c->r2_ck.p, slots);
            hipLaunchKernelGGL((rss2_merge_kernel<2>), dim3((unsigned)n_tasks), dim3(64), 0, c->stream, Xr, vw, srows, task_row_off,
                               d, R, c->r2_F.p, c->r2_ck.p, slots, meta, vals, cmeans);
        } else {
            const size_t lds = (size_t)(4 * R2_BR * 65 + 2 * 64 * 64 + 2 * 64) * sizeof(double);
            cge_allow_lds((const void *)rss2_chain_lds_kernel<1, 4>, 160 * 1024);
            hipLaunchKernelGGL((rss2_chain_lds_kernel<1, 4>), gridA, dim3(256), lds, c->stream, Xr, vw, srows, task_row_off, d, R,
                               c->r2_F.p, c->r2_ck.p, slots);
            hipLaunchKernelGGL((rss2_merge_kernel<1>), dim3((unsigned)n_tasks), dim3(64), 0, c->stream, Xr, vw, srows, task_row_off,
                               d, R, c->r2_F.p, c->r2_ck.p, slots, meta, vals, cmeans);
        }
        return;
    }
    k_rss2_walk_one_kernel(c, Xr, vw, srows, task_row_off, n_tasks, d, meta, vals, cmeans);
}

// split_cluster_size / split_cluster_diameter (src/landmarks.jl:212-262): the 1-D cut of z at its median
// (`zs` = z sorted per task) or at (min + max)/2, one wave per task.  side[j] = 1 (low) / 2 (high) in the rows' own
// order.  A row with z == cut joins the side that is smaller at that moment of the reference's sequential pass
// (:229-236, :255-261); the running sizes are carried across 64-row chunks as ballot counts, the ties of a chunk are
// settled in lane order.
// Round 4: a workgroup of four waves per task, and the sequential tie rule only where a tie exists.  The cut (median of the
// sorted z, or (min + max) / 2) almost never EQUALS a projection; without such a row the sides are `z < cut` row by row and
// the size of the low side is a count -- both fully parallel (and the count replaces the side_counts launch).  A task that
// does hold a row with z == cut is redone by its first wave with the reference's sequential rule (the code of rounds 1-3).
__global__ __launch_bounds__(256) void cut_sides_kernel(const double *__restrict__ z, const double *__restrict__ zs,
                                                        const i32 *__restrict__ task_row_off, int use_median,
                                                        unsigned char *__restrict__ side, i32 *__restrict__ nlow_out,
                                                        int *__restrict__ tie_tasks) {
    __shared__ double slo[4], shi[4];
    __shared__ int scnt[4];
    const i64 t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    if (k <= 0) { if (tid == 0 && nlow_out) nlow_out[t] = 0; return; }
    double cut;
    if (use_median)
        cut = (k & 1) ? zs[o + k / 2] : zs[o + k / 2 - 1] / 2.0 + zs[o + k / 2] / 2.0;
    else {
        double lo = z[o], hi = z[o];
        i64 j = tid;
        for (; j + 768 < k; j += 1024) { // four loads in flight per thread
            const double v0 = z[o + j], v1 = z[o + j + 256], v2 = z[o + j + 512], v3 = z[o + j + 768];
            lo = fmin(fmin(lo, v0), fmin(fmin(v1, v2), v3));
            hi = fmax(fmax(hi, v0), fmax(fmax(v1, v2), v3));
        }
        for (; j < k; j += 256) {
            const double v = z[o + j];
            lo = fmin(lo, v);
            hi = fmax(hi, v);
        }
        for (int off = 32; off > 0; off >>= 1) {
            lo = fmin(lo, __shfl_xor(lo, off));
            hi = fmax(hi, __shfl_xor(hi, off));
        }
        if (lane == 0) { slo[wv] = lo; shi[wv] = hi; }
        __syncthreads();
        lo = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3])); // (min / max: any grouping gives the same value)
        hi = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
        cut = (lo + hi) / 2.0;
    }
    // the parallel pass: sides as if there were no tie (a tie provisionally high), the low side counted
    int cnt = 0, tie = 0;
    {
        i64 j = tid;
        for (; j + 768 < k; j += 1024) {
            const double v0 = z[o + j], v1 = z[o + j + 256], v2 = z[o + j + 512], v3 = z[o + j + 768];
            side[o + j] = v0 < cut ? 1 : 2; side[o + j + 256] = v1 < cut ? 1 : 2;
            side[o + j + 512] = v2 < cut ? 1 : 2; side[o + j + 768] = v3 < cut ? 1 : 2;
            cnt += (v0 < cut) + (v1 < cut) + (v2 < cut) + (v3 < cut);
            tie |= (v0 == cut) | (v1 == cut) | (v2 == cut) | (v3 == cut);
        }
        for (; j < k; j += 256) {
            const double v = z[o + j];
            side[o + j] = v < cut ? 1 : 2;
            cnt += v < cut;
            tie |= v == cut;
        }
    }
    // The provisional sides have been ACKNOWLEDGED by memory before the vote: the redo below rewrites rows that other waves
    // stored, and a barrier alone orders the instructions of the waves, not the arrival of their stores (a provisional side
    // landing behind the final one would leave `side` and the counted sizes in disagreement).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int any_tie = __syncthreads_or(tie);
    if (!any_tie) {
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane == 0) scnt[wv] = cnt;
        __syncthreads();
        if (tid == 0 && nlow_out) nlow_out[t] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
        return;
    }
    if (wv != 0) return;
    if (lane == 0 && tie_tasks) atomicAdd(tie_tasks, 1); // (statistics: how often the sequential rule is needed)
    // a row with z == cut joins the side that is smaller at that moment of the reference's sequential pass (:229-236,
    // :255-261): one wave walks the task, the running sizes carried across 64-row chunks as ballot counts, the ties of a
    // chunk settled in lane order
    i64 nlow = 0, nhigh = 0;
    double znext[4]; // the next four 64-row chunks are requested while the current ones are settled
#pragma unroll
    for (int u = 0; u < 4; u++) znext[u] = (64 * u + lane < k) ? z[o + 64 * u + lane] : 0.0;
    for (i64 base0 = 0; base0 < k; base0 += 256) {
        double zcur[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            zcur[u] = znext[u];
            const i64 jn = base0 + 256 + 64 * u + lane;
            znext[u] = jn < k ? z[o + jn] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
        const i64 base = base0 + 64 * u;
        if (base >= k) break; // uniform
        const i64 j = base + lane;
        const bool valid = j < k;
        const double zj = zcur[u];
        const bool isl = valid && zj < cut, ise = valid && zj == cut, ish = valid && !isl && !ise;
        const unsigned long long ml = __ballot(isl), mh = __ballot(ish);
        unsigned long long me = __ballot(ise), al = 0ULL; // al: ties that go low
        i64 el = 0, eh = 0;
        while (me) {
            const int i = __ffsll((long long)me) - 1;
            me &= me - 1;
            const unsigned long long below = (i == 0) ? 0ULL : (~0ULL >> (64 - i));
            const i64 cl = nlow + __popcll(ml & below) + el, ch = nhigh + __popcll(mh & below) + eh;
            if (cl < ch) { al |= 1ULL << i; el++; } else eh++;
        }
        if (valid) side[o + j] = (isl || (ise && ((al >> lane) & 1ULL))) ? 1 : 2;
        nlow += __popcll(ml) + el;
        nhigh += __popcll(mh) + eh;
        }
    }
    if (lane == 0 && nlow_out) nlow_out[t] = (i32)nlow;
}
void k_cut_sides(cge_ctx *c, const double *z, const double *zs, const i32 *task_row_off, i64 n_tasks, int use_median,
                 unsigned char *side, i32 *nlow_out, int *tie_tasks) {
    ScopedKernelTimer t(c, "cut_sides");
    hipLaunchKernelGGL(cut_sides_kernel, dim3((unsigned)n_tasks), dim3(256), 0, c->stream, z, zs, task_row_off, use_median, side, nlow_out,
                       tie_tasks);
}

// Side flags of one rss round, derived on the device (no per-round row-sized upload).  Per task t,
// params[4t..4t+3] = { prev_med, absorb (0/1/2: the side that was absorbed after the previous round),
// cur_med, mode (0 idle, 1 median round, 2 leftover round) }.  state[j]: 0 = gray, 1 = low, 2 = high.
__global__ void rss_side_kernel(const double *__restrict__ z, const i32 *__restrict__ row_task, i64 n_rows,
                                const double *__restrict__ params, unsigned char *__restrict__ state,
                                unsigned char *__restrict__ side) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rows) return;
    const double *p = params + 4 * (i64)row_task[j];
    unsigned char st = state[j], sd = 0;
    if (st == 0) {
        const int absorb = (int)p[1], mode = (int)p[3];
        const double zj = z[j];
        if (absorb != 0 && ((zj < p[0]) ? 1 : 2) == absorb) {
            st = (unsigned char)absorb;
            state[j] = st;
        }
        if (st == 0 && mode != 0) sd = (mode == 2) ? 1 : ((zj < p[2]) ? 1 : 2);
    }
    side[j] = sd;
}
void k_rss_side(cge_ctx *c, const double *z, const i32 *row_task, i64 n_rows, const double *params,
                unsigned char *state, unsigned char *side) {
    hipLaunchKernelGGL(rss_side_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, z, row_task,
                       n_rows, params, state, side);
}

// ------------------------------------------------------------------------------------------------
// Up to CGE_WORD_SEGS runs of 4-byte words copied device to device by one launch: the small tables of a landmark batch
// are staged through ONE pinned buffer (a copy per table costs ~20 us of idle stream each).
struct WordSegs {
    unsigned *dst[CGE_WORD_SEGS];
    const unsigned *src[CGE_WORD_SEGS];
    long long end[CGE_WORD_SEGS]; // running totals
};
__global__ void copy_words_kernel(WordSegs s, int n) {
    const long long total = s.end[n - 1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long beg = 0;
#pragma unroll
        for (int q = 0; q < CGE_WORD_SEGS; q++) {
            if (q < n && i >= beg && i < s.end[q]) s.dst[q][i - beg] = s.src[q][i - beg];
            if (q < n) beg = s.end[q];
        }
    }
}
void k_copy_words(cge_ctx *c, int nseg, void *const *dst, const void *const *src, const i64 *words) {
    if (nseg <= 0) return;
    if (nseg > CGE_WORD_SEGS) CGE_THROW(CGE_E_ARG, "k_copy_words: %d segments", nseg);
    WordSegs s;
    long long tot = 0;
    for (int q = 0; q < CGE_WORD_SEGS; q++) {
        s.dst[q] = q < nseg ? (unsigned *)dst[q] : nullptr;
        s.src[q] = q < nseg ? (const unsigned *)src[q] : nullptr;
        if (q < nseg) tot += words[q];
        s.end[q] = tot;
    }
    if (tot == 0) return;
    const unsigned nb = (unsigned)std::min<long long>((tot + 255) / 256, 1024);
    hipLaunchKernelGGL(copy_words_kernel, dim3(nb), dim3(256), 0, c->stream, s, nseg);
}

// ------------------------------------------------------------------------------------------------
// Batched principal eigenvector, one workgroup per d x d covariance, d <= 128.  Same algorithm as
// host_eig_top (landmarks_host.cpp): Householder tridiagonalisation, largest eigenvalue by (64-way)
// multisection on the Sturm count, inverse iteration with a pivoted tridiagonal LU, back-transformation,
// sign = largest-|component| positive.  Replaces `eigvecs(A)[:, end]` (src/landmarks.jl:99,162,225,254).
//
// The matrix lives in REGISTERS: wave w owns the columns [NC*w, NC*w+NC), lane l the rows l + 64*r (r < NR),
// so a thread holds an NR x NC block.  A Householder step then needs two workgroup barriers only: everything
// that is O(d) (the reflector, its norm, p.u, w) is recomputed by every wave from LDS with in-wave DPP
// reductions; the O(d^2) work (B u and the rank-2 update) runs on the register blocks with u, w broadcast
// from LDS.  Finished columns are never touched again (u and w are zero there), so column k keeps the
// reflector of step k below its sub-diagonal -- the back-transformation reads it from the same registers,
// wave by wave, with in-wave reductions (4 barriers in all).
//
// 1/q to ~1 ulp without the IEEE division sequence: hardware reciprocal + one Newton step.  Used only in
// chains whose results are themselves iterated (Sturm counts, inverse iteration), never in sums that
// are compared with the reference.
__device__ __forceinline__ double fast_rcp(double q) {
    double r = __builtin_amdgcn_rcp(q);
    r = fma(fma(-q, r, 1.0), r, r);
    return r;
}
// acc += (lane N of the reader's row of 16 lanes of `bc`) * a as ONE instruction: gfx950's fp64 FMA takes a DPP broadcast on its
// first factor at the plain FMA's rate (profiles/microbench_dpp_fmac.hip), so a value that is the same for all lanes needs no
// LDS broadcast read and no v_readlane.  `bc` must have been written at least two instructions earlier (dpp_fence).
template <int N>
__device__ __forceinline__ void fmac_bcast(double &acc, double bc, double a) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(a), "n"(N));
}
// acc -= (that lane's value) * a: the same FMA with the sign flipped on the second factor (a source modifier: exact)
template <int N>
__device__ __forceinline__ void fnmac_bcast(double &acc, double bc, double a) {
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(a), "n"(N));
}
template <int NG>
__device__ __forceinline__ void dpp_fence(double (&v)[NG]) { // VALU write -> DPP read of the same register: two wait states
    if constexpr (NG == 1) asm volatile("s_nop 1" : "+v"(v[0]));
    else asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[NG - 1]));
}
// the column loops of a Householder step with compile-time column numbers (the DPP lane is an immediate); R0 = the first
// row block that still has live rows (the blocks above it hold finished rows only: u = w = 0 there, nothing would change)
template <int JJ, int NR, int NC, int R0 = 0>
struct EigCols {
    static constexpr int NG = (NC + 15) / 16;
    static __device__ __forceinline__ void matvec(double (&s)[NR], const double (&a)[NR][NC], const double (&uc)[NG]) {
#pragma unroll
        for (int r = R0; r < NR; r++) fmac_bcast<JJ % 16>(s[r], uc[JJ / 16], a[r][JJ]);
        EigCols<JJ + 1, NR, NC, R0>::matvec(s, a, uc);
    }
    static __device__ __forceinline__ void rank2(double (&a)[NR][NC], const double (&uc)[NG], const double (&wc)[NG],
                                                 const double (&u)[NR], const double (&w)[NR]) {
#pragma unroll
        for (int r = R0; r < NR; r++) {
            fnmac_bcast<JJ % 16>(a[r][JJ], uc[JJ / 16], w[r]); // a - w_r u_j
            fnmac_bcast<JJ % 16>(a[r][JJ], wc[JJ / 16], u[r]); //   - u_r w_j
        }
        EigCols<JJ + 1, NR, NC, R0>::rank2(a, uc, wc, u, w);
    }
};
template <int NR, int NC, int R0>
struct EigCols<NC, NR, NC, R0> {
    static constexpr int NG = (NC + 15) / 16;
    static __device__ __forceinline__ void matvec(double (&)[NR], const double (&)[NR][NC], const double (&)[NG]) {}
    static __device__ __forceinline__ void rank2(double (&)[NR][NC], const double (&)[NG], const double (&)[NG], const double (&)[NR],
                                                 const double (&)[NR]) {}
};
// v[r] = a[r][jj] for a run-time column slot jj (uniform): a switch, so that the register block stays in registers
template <int NR, int NC>
__device__ __forceinline__ void eig_pick_column(const double (&a)[NR][NC], int jj, double (&v)[NR]) {
#define EP_CASE(J)                                                                                                          \
    case J:                                                                                                                 \
        if constexpr (J < NC) {                                                                                             \
            _Pragma("unroll") for (int r = 0; r < NR; r++) v[r] = a[r][J];                                                  \
        }                                                                                                                   \
        break;
#pragma unroll
    for (int r = 0; r < NR; r++) v[r] = 0.0;
    switch (jj) {
        EP_CASE(0) EP_CASE(1) EP_CASE(2) EP_CASE(3) EP_CASE(4) EP_CASE(5) EP_CASE(6) EP_CASE(7)
        EP_CASE(8) EP_CASE(9) EP_CASE(10) EP_CASE(11) EP_CASE(12) EP_CASE(13) EP_CASE(14) EP_CASE(15)
        EP_CASE(16) EP_CASE(17) EP_CASE(18) EP_CASE(19) EP_CASE(20) EP_CASE(21) EP_CASE(22) EP_CASE(23)
        EP_CASE(24) EP_CASE(25) EP_CASE(26) EP_CASE(27) EP_CASE(28) EP_CASE(29) EP_CASE(30) EP_CASE(31)
    default: break;
    }
#undef EP_CASE
}
// DPPF: the two O(d^2) loops of a step take u_j / w_j as DPP broadcasts from registers that hold the wave's column values
// (read from LDS once per step: NC / 16 reads instead of NC) -- and u's column values come from the published row itself, so
// the matrix-vector product no longer waits for the reflector's square root and divisions (only the column next to the
// diagonal does).  Same operations on the same values as the LDS-broadcast form: the same bits.
template <int NR, int NC, bool DPPF>
__global__ __launch_bounds__(256, 2) void group_eig_kernel(const double *__restrict__ cov, int d,
                                                           double *__restrict__ vec, int diag_stage) {
    constexpr int DP = 64 * NR; // padded dimension (rows held); 4*NC >= d columns held
    __shared__ __attribute__((aligned(16))) double X[DP], U[DP], W[DP], Pp[4][DP], V[DP];
    __shared__ __attribute__((aligned(16))) double diag[DP], off[DP], beta[DP], V0[DP], tri[4 * DP], red[32];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6); // (the wave's number as a scalar)
    const double *src = cov + (size_t)blockIdx.x * d * d;
    double *out = vec + (size_t)blockIdx.x * d;
    if (d == 1) {
        if (tid == 0) out[0] = 1.0;
        return;
    }
    double a[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int row = lane + 64 * r;
#pragma unroll
        for (int jj = 0; jj < NC; jj++) {
            const int col = NC * wv + jj;
            a[r][jj] = (row < d && col < d) ? src[(size_t)col * d + row] : 0.0; // symmetric: coalesced along rows
        }
    }
    // publish COLUMN kn of the matrix: X[row] = A[row][kn], two full-wave stores by the wave that owns the column (the register
    // is picked by a switch on the column's slot: scalar branches).  Rounds 1-4 published ROW kn instead -- a row is spread
    // over the waves' static column slots, no switch -- but a row sits in ONE lane: 16 single-lane 16-byte stores per wave,
    // 64 LDS instructions per step through the CU's one LDS pipeline, ~1000 cycles of every step (CGE_EIG_CLOCK).  The
    // matrix is symmetric up to the rounding of the two fused updates (the rounding level of the method itself), so the
    // column differs from the row in the last bits only; readers mask the part above the sub-diagonal themselves.
    auto extract = [&](int kn, auto) {
        if (wv == kn / NC) { // uniform
            double v[NR];
            eig_pick_column<NR, NC>(a, kn % NC, v);
#pragma unroll
            for (int r = 0; r < NR; r++) X[lane + 64 * r] = v[r];
        }
    };
    using RBlk0 = std::integral_constant<int, 0>;
    using RBlkL = std::integral_constant<int, NR - 1>;
    if (tid < DP) { beta[tid] = 0.0; off[tid] = 0.0; V0[tid] = 0.0; X[tid] = 0.0; }
    __syncthreads();
    extract(0, RBlk0{});
    // CGE_EIG_CLOCK (a build flag, diagnostics only): s_memtime stamps around the sections of a step, summed over the steps and
    // printed by waves 0 and 3 of the first matrix (the stamps wait for the LDS queue: shares, not absolute times)
#ifdef CGE_EIG_CLOCK
    unsigned long long ck_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = __builtin_amdgcn_s_memtime();
#define CK(j) { const unsigned long long ck_n = __builtin_amdgcn_s_memtime(); ck_acc[j] += ck_n - ck_t; ck_t = ck_n; }
#else
#define CK(j)
#endif
    // ---- tridiagonalisation ------------------------------------------------------------------------
    // One Householder step; R0 (compile-time) = the first row block with live rows: once k has passed row 63 the upper block
    // of a 128-row matrix is finished and the two O(d^2) loops skip it (exact: u = w = 0 there).
    auto hh_step = [&](const int k, auto r0c) {
        constexpr int R0 = decltype(r0c)::value;
        // The owner's store of column k has LANDED before anybody passes the barrier: spelled out, because the compiler leaves
        // the wait out of the loop's back edge (the store sits in a conditional block in front of it) and the readers then
        // see the previous column now and then -- found as run-to-run differences of the landmark pipeline.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads(); // (a) X, diag of column k are visible; U, W, Pp of the previous step are dead
        CK(0)
        const int o = k + 1;
        double x[NR], u[NR];
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int row = lane + 64 * r;
            x[r] = (row > k) ? X[row] : 0.0;
            part += (row > o) ? x[r] * x[r] : 0.0;
        }
        constexpr int NG = (NC + 15) / 16;
        double uc[NG], wc[NG];
        if (DPPF) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int col = NC * wv + 16 * g + (lane & 15);
                uc[g] = (col > k && col < DP) ? X[col < DP ? col : 0] : 0.0;
            }
        }
        const double alpha = X[o];
        if (tid == 0) diag[k] = X[k];
        const double sigma = wave_allsum(part);
        CK(1)
        // sigma == 0: no reflection at this step.  It runs through the same code with u = w = 0 (the update then
        // leaves every register bit as it is) -- a branch around the update would make the compiler keep two copies
        // of the matrix block.
        const bool refl = sigma != 0.0;
        double mu, v0r, bpr;
        if (diag_stage == 13) { mu = alpha + sigma; v0r = alpha - mu; bpr = sigma * 0.5; } // timing diagnostics (wrong results)
        else if (diag_stage == 11) {
            mu = sqrt(alpha * alpha + sigma);
            v0r = (alpha <= 0.0) ? alpha - mu : -sigma * fast_rcp(alpha + mu);
            bpr = 2.0 * fast_rcp(sigma + v0r * v0r);
        } else {
            mu = sqrt(alpha * alpha + sigma);
            v0r = (alpha <= 0.0) ? alpha - mu : -sigma / (alpha + mu);
            bpr = 2.0 / (sigma + v0r * v0r);
        }
        const double v0 = refl ? v0r : 0.0;
        const double bp = refl ? bpr : 0.0; // H = I - bp u u^T, u = (v0, x[o+1..])
        CK(2)
        if (tid == 0) { beta[k] = bp; off[k] = refl ? mu : alpha; V0[k] = v0; }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int row = lane + 64 * r;
            u[r] = (row == o) ? v0 : x[r];
            if (!DPPF) U[row] = u[r]; // every wave writes the same values
        }
        if (DPPF) {
#pragma unroll
            for (int g = 0; g < NG; g++)
                if (NC * wv + 16 * g + (lane & 15) == o) uc[g] = v0;
            dpp_fence(uc);
        }
        __builtin_amdgcn_wave_barrier();
        const bool live = NC * wv + NC > o && diag_stage != 10; // this wave still owns unfinished columns
        { // partial p = B u over the wave's columns
            double s[NR];
#pragma unroll
            for (int r = 0; r < NR; r++) s[r] = 0.0;
            if (DPPF) {
                if (live) EigCols<0, NR, NC, R0>::matvec(s, a, uc);
            } else if (live) {
#pragma unroll
                for (int jj = 0; jj < NC; jj++) {
                    if (jj % 8 == 0) asm volatile("" ::: "memory"); // at most 8 broadcast reads in flight (registers)
                    const double uj = U[NC * wv + jj];
#pragma unroll
                    for (int r = R0; r < NR; r++) s[r] = fma(a[r][jj], uj, s[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < NR; r++) Pp[wv][lane + 64 * r] = s[r];
        }
        CK(3)
        __syncthreads(); // (b)
        CK(4)
        double w[NR];
        part = 0.0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int row = lane + 64 * r;
            w[r] = bp * (((Pp[0][row] + Pp[1][row]) + Pp[2][row]) + Pp[3][row]);
            part += w[r] * u[r];
        }
        const double K = 0.5 * bp * wave_allsum(part);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int row = lane + 64 * r;
            w[r] = (row >= o) ? w[r] - K * u[r] : 0.0; // finished rows / columns stay as they are
            W[row] = w[r];
        }
        __builtin_amdgcn_wave_barrier();
        CK(5)
        if (DPPF) {
            if (diag_stage != 10) {
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int col = NC * wv + 16 * g + (lane & 15);
                    wc[g] = W[col < DP ? col : 0];
                }
                dpp_fence(wc);
                EigCols<0, NR, NC, R0>::rank2(a, uc, wc, u, w);
            }
        } else if (diag_stage != 10) { // rank-2 update; waves whose columns are all finished run it too (u = w = 0 there: nothing changes)
#pragma unroll
            for (int jj = 0; jj < NC; jj++) {
                if (jj % 8 == 0) asm volatile("" ::: "memory");
                const double uj = U[NC * wv + jj], wj = W[NC * wv + jj];
#pragma unroll
                for (int r = R0; r < NR; r++) a[r][jj] = fma(-u[r], wj, fma(-w[r], uj, a[r][jj]));
            }
        }
        CK(6)
        extract(k + 1, r0c); // row k + 1 lies in block R0 (k + 1 < 64 in the first loop below, >= 64 in the second)
        CK(7)
    };
    {
        int k = 0;
        if constexpr (NR > 1) {
            for (; k + 2 < d && k + 1 < 64; k++) hh_step(k, RBlk0{});
            for (; k + 2 < d; k++) hh_step(k, RBlkL{});
        } else {
            for (; k + 2 < d; k++) hh_step(k, RBlk0{});
        }
    }
#ifdef CGE_EIG_CLOCK
    if (blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 3))
        printf("eig clock wave %d: barrier_a %llu  reads+sigma %llu  sqrt/div %llu  matvec+Pp %llu  barrier_b %llu  w %llu  rank2 %llu  extract %llu (s_memtime ticks over %d steps)\n",
               wv, ck_acc[0], ck_acc[1], ck_acc[2], ck_acc[3], ck_acc[4], ck_acc[5], ck_acc[6], ck_acc[7], d - 2);
#endif
#undef CK
    __syncthreads();
    if (tid == 0) { diag[d - 2] = X[d - 2]; off[d - 2] = X[d - 1]; } // column d-2 was the last one published
    __syncthreads();
    if (NR > 1 && d - 1 >= 64) extract(d - 1, RBlkL{});
    else extract(d - 1, RBlk0{});
    __syncthreads();
    if (tid == 0) diag[d - 1] = X[d - 1];
    __syncthreads();
    if (diag_stage == 1) { if (tid < d) out[tid] = diag[tid]; return; } // timing diagnostic only
    // ---- Gershgorin bounds ---------------------------------------------------------------------------
    double glo = 1e300, ghi = -1e300, gn = 0.0;
    if (tid < d) {
        const double rad = (tid > 0 ? fabs(off[tid - 1]) : 0.0) + (tid + 1 < d ? fabs(off[tid]) : 0.0);
        glo = diag[tid] - rad;
        ghi = diag[tid] + rad;
        gn = fabs(diag[tid]) + rad;
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        glo = fmin(glo, __shfl_xor(glo, o2));
        ghi = fmax(ghi, __shfl_xor(ghi, o2));
        gn = fmax(gn, __shfl_xor(gn, o2));
    }
    double *gs = red + 8;
    if (lane == 0) { gs[wv * 3] = glo; gs[wv * 3 + 1] = ghi; gs[wv * 3 + 2] = gn; }
    __syncthreads();
    glo = fmin(fmin(gs[0], gs[3]), fmin(gs[6], gs[9]));
    ghi = fmax(fmax(gs[1], gs[4]), fmax(gs[7], gs[10]));
    gn = fmax(fmax(gs[2], gs[5]), fmax(gs[8], gs[11]));
    const double tiny = fmax(gn, 2.2250738585072014e-308) * 2.220446049250313e-16;
    // ---- largest eigenvalue: 64-way multisection on the Sturm count (wave 0) ------------------------------
    if (tid < 64) {
        double lo = glo, hi = ghi + tiny;
        for (int it = 0; it < 64; it++) {
            const double x = lo + (hi - lo) * ((double)(tid + 1) / 65.0);
            int cnt = 0;
            double q = diag[0] - x;
            if (q < 0) cnt++;
            for (int i = 1; i < d; i++) {
                if (q == 0.0) q = tiny;
                q = diag[i] - x - (off[i - 1] * off[i - 1]) * fast_rcp(q);
                if (q < 0) cnt++;
            }
            const unsigned long long mask = __ballot(cnt >= d);
            double nlo, nhi;
            if (mask == 0ULL) {
                nlo = __shfl(x, 63);
                nhi = hi;
            } else {
                const int f = __ffsll((long long)mask) - 1;
                nhi = __shfl(x, f);
                nlo = (f > 0) ? __shfl(x, f - 1) : lo;
            }
            if (!(nhi > nlo) || (nlo == lo && nhi == hi)) break;
            lo = fmax(lo, nlo);
            hi = fmin(hi, nhi);
        }
        if (tid == 0) red[4] = 0.5 * (lo + hi);
    }
    __syncthreads();
    if (diag_stage == 2) { if (tid < d) out[tid] = red[4]; return; } // timing diagnostic only
    // (Round 4 also had a REGISTER form of the inverse iteration -- factors and iterate in the wave's registers, v_readlane instead
    // of LDS reads by lane 0: same bits, 0.60 against 0.50 ms per launch, profiles/r04_eig_tail_ab.txt; removed in round 5.)
    // ---- inverse iteration, LDS form (wave 0: lane 0 runs the O(d) recurrences with the carried values in registers, the
    //      wave does the element-wise parts) ---------------------------------------------------------------------
    if (wv == 0) {
        const double lam = red[4];
        double *dl = tri, *dd = tri + DP, *du = tri + 2 * DP, *du2 = tri + 3 * DP;
        double *y = V;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = lane + 64 * r;
            dd[i] = (i < d) ? diag[i] - lam : 1.0;
            dl[i] = du[i] = (i + 1 < d) ? off[i] : 0.0;
            du2[i] = 0.0;
            y[i] = (i < d) ? 1.0 + 0.01 * (double)(((unsigned)i * 2654435761u) % 97u) / 97.0 : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long swp0 = 0ULL, swp1 = 0ULL; // pivot flags, d <= 128 (lane 0)
        // The three recurrences below run in lane 0 over values in LDS.  Written step by step (load, compute, store), every step
        // waited for its own LDS round trip: the stores of step i may alias the loads of step i + 1 as far as the compiler can
        // tell.  So the operands of EB steps are read in one batch, the steps run on registers with selects instead of branches
        // (the same operations on the same values: the same bits), and the results are stored behind them.
        constexpr int EB = 8;
        if (lane == 0) { // LU with partial pivoting of the shifted tridiagonal matrix
            double di = dd[0], ui = du[0];
            auto lu_step = [&](int i, double li, double dn, double un, double &o_dd, double &o_dl, double &o_du, double &o_du2,
                               bool &o_sw) {
                const bool keep = fabs(di) >= fabs(li); // no row swap
                const double den = keep ? (di == 0.0 ? tiny : di) : li, num = keep ? li : di;
                const double f = num * fast_rcp(den);
                o_dd = den;
                o_dl = f;
                o_du = keep ? ui : dn;
                const double ya = keep ? dn : ui, yb = keep ? ui : dn;
                di = ya - f * yb;
                const double fu = f * un;
                ui = keep ? un : -fu;
                o_du2 = un;
                o_sw = !keep;
                if (!keep) { if (i < 64) swp0 |= 1ULL << i; else swp1 |= 1ULL << (i - 64); }
            };
            int i = 0;
            for (; i + EB < d; i += EB) { // steps i .. i + EB - 1 (all of them have i + 1 < d)
                double li[EB], dn[EB], un[EB], odd[EB], odl[EB], odu[EB], odu2[EB];
                bool osw[EB];
#pragma unroll
                for (int q = 0; q < EB; q++) { li[q] = dl[i + q]; dn[q] = dd[i + q + 1]; un[q] = du[i + q + 1]; }
#pragma unroll
                for (int q = 0; q < EB; q++) lu_step(i + q, li[q], dn[q], un[q], odd[q], odl[q], odu[q], odu2[q], osw[q]);
#pragma unroll
                for (int q = 0; q < EB; q++) {
                    dd[i + q] = odd[q];
                    dl[i + q] = odl[q];
                    du[i + q] = odu[q];
                    if (osw[q] && i + q + 2 < d) du2[i + q] = odu2[q];
                }
            }
            for (; i + 1 < d; i++) {
                double o_dd, o_dl, o_du, o_du2;
                bool o_sw;
                lu_step(i, dl[i], dd[i + 1], du[i + 1], o_dd, o_dl, o_du, o_du2, o_sw);
                dd[i] = o_dd;
                dl[i] = o_dl;
                du[i] = o_du;
                if (o_sw && i + 2 < d) du2[i] = o_du2;
            }
            if (di == 0.0) di = tiny;
            dd[d - 1] = di;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < NR; r++) { // the solves multiply by the reciprocal pivots
            const int i = lane + 64 * r;
            if (i < d) dd[i] = fast_rcp(dd[i]);
        }
        __builtin_amdgcn_wave_barrier();
        for (int it = 0; it < 3; it++) {
            if (lane == 0) {
                double yi = y[0];
                auto fw_step = [&](int i, double yn, double li, double &o_y) { // forward: L with the recorded row swaps
                    const bool sw = (i < 64) ? ((swp0 >> i) & 1ULL) : ((swp1 >> (i - 64)) & 1ULL);
                    o_y = sw ? yn : yi;
                    const double ya = sw ? yi : yn, yb = sw ? yn : yi;
                    yi = ya - li * yb;
                };
                int i = 0;
                for (; i + EB < d; i += EB) {
                    double yn[EB], li[EB], oy[EB];
#pragma unroll
                    for (int q = 0; q < EB; q++) { yn[q] = y[i + q + 1]; li[q] = dl[i + q]; }
#pragma unroll
                    for (int q = 0; q < EB; q++) fw_step(i + q, yn[q], li[q], oy[q]);
#pragma unroll
                    for (int q = 0; q < EB; q++) y[i + q] = oy[q];
                }
                for (; i + 1 < d; i++) {
                    double oy;
                    fw_step(i, y[i + 1], dl[i], oy);
                    y[i] = oy;
                }
                double y1 = yi * dd[d - 1]; // backward: U with two super-diagonals
                y[d - 1] = y1;
                double y0 = (y[d - 2] - du[d - 2] * y1) * dd[d - 2];
                y[d - 2] = y0;
                i = d - 3;
                for (; i - (EB - 1) >= 0; i -= EB) { // steps i, i - 1, .., i - EB + 1
                    double yy[EB], uu[EB], u2[EB], pv[EB], ot[EB];
#pragma unroll
                    for (int q = 0; q < EB; q++) { yy[q] = y[i - q]; uu[q] = du[i - q]; u2[q] = du2[i - q]; pv[q] = dd[i - q]; }
#pragma unroll
                    for (int q = 0; q < EB; q++) {
                        const double t = (yy[q] - uu[q] * y0 - u2[q] * y1) * pv[q];
                        ot[q] = t;
                        y1 = y0;
                        y0 = t;
                    }
#pragma unroll
                    for (int q = 0; q < EB; q++) y[i - q] = ot[q];
                }
                for (; i >= 0; i--) {
                    const double t = (y[i] - du[i] * y0 - du2[i] * y1) * dd[i];
                    y[i] = t;
                    y1 = y0;
                    y0 = t;
                }
            }
            __builtin_amdgcn_wave_barrier();
            double yv[NR], amax = 0.0;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                yv[r] = y[lane + 64 * r];
                amax = fmax(amax, fabs(yv[r]));
            }
            for (int o2 = 32; o2 > 0; o2 >>= 1) amax = fmax(amax, __shfl_xor(amax, o2));
            if (!(amax > 0.0) || !(amax < 1e300)) { // uniform over the wave
#pragma unroll
                for (int r = 0; r < NR; r++) y[lane + 64 * r] = (lane + 64 * r == 0) ? 1.0 : 0.0;
                __builtin_amdgcn_wave_barrier();
                break;
            }
            const double ra = 1.0 / amax;
            double part = 0.0;
#pragma unroll
            for (int r = 0; r < NR; r++) { yv[r] *= ra; part += yv[r] * yv[r]; }
            const double rn = 1.0 / sqrt(wave_allsum(part));
#pragma unroll
            for (int r = 0; r < NR; r++) y[lane + 64 * r] = yv[r] * rn;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (diag_stage == 3) { if (tid < d) out[tid] = V[tid]; return; } // timing diagnostic only
    // ---- back-transformation x = H_0 H_1 ... H_{d-3} y: the wave that owns column k holds reflector k ------------
    for (int ww = 3; ww >= 0; ww--) {
        if (wv == ww && NC * ww <= d - 3) {
            double y[NR];
#pragma unroll
            for (int r = 0; r < NR; r++) y[r] = V[lane + 64 * r];
#pragma unroll
            for (int jj = NC - 1; jj >= 0; jj--) {
                int k = NC * ww + jj;
                asm volatile("" : "+v"(k)); // keeps the 2*NC*NR row masks below from being hoisted (and spilled) together
                if (k > d - 3) continue;
                const double bp = beta[k];
                if (bp == 0.0) continue; // no reflection at this step
                const double v0 = V0[k];
                double uu[NR], part = 0.0;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const int row = lane + 64 * r;
                    uu[r] = (row > k + 1) ? a[r][jj] : (row == k + 1 ? v0 : 0.0);
                    part += uu[r] * y[r];
                }
                const double s = bp * wave_allsum(part);
#pragma unroll
                for (int r = 0; r < NR; r++) y[r] -= s * uu[r];
            }
#pragma unroll
            for (int r = 0; r < NR; r++) V[lane + 64 * r] = y[r];
        }
        __syncthreads();
    }
    // normalise; sign: the component of largest magnitude (the first one on ties) is positive
    if (wv == 0) {
        double yv[NR], part = 0.0, best = -1.0;
        int bi = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            yv[r] = V[lane + 64 * r]; // rows >= d are zero
            part += yv[r] * yv[r];
            if (fabs(yv[r]) > best) { best = fabs(yv[r]); bi = lane + 64 * r; }
        }
        double nrm = sqrt(wave_allsum(part));
        const bool degenerate = !(nrm > 0.0) || !(nrm < 1e300); // zero matrix: any unit vector is an eigenvector
        if (degenerate) {
#pragma unroll
            for (int r = 0; r < NR; r++) { yv[r] = (lane + 64 * r == 0) ? 1.0 : 0.0; }
            nrm = 1.0;
            best = (lane == 0) ? 1.0 : 0.0;
            bi = lane;
        }
        double bv = best;
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
            const double ob = __shfl_xor(bv, o2);
            const int oi = __shfl_xor(bi, o2);
            if (ob > bv || (ob == bv && oi < bi)) { bv = ob; bi = oi; }
        }
        const double lead = (bi >= 64 && NR > 1) ? lane_value(yv[NR - 1], bi & 63) : lane_value(yv[0], bi & 63);
        const double sg = ((lead < 0.0) ? -1.0 : 1.0) / nrm;
#pragma unroll
        for (int r = 0; r < NR; r++)
            if (lane + 64 * r < d) out[lane + 64 * r] = yv[r] * sg;
    }
}
// (Rounds 3-4 carried two more forms of this solver -- the columns dealt cyclically to 4 or 8 waves with dead work skipped,
// and for 128 < d <= 512 an unblocked form that swept the trailing matrix three times per step: 8 d^3 bytes per matrix, 1.07 GB
// at d = 512.  Both gave the same bits and were measured slower (profiles/r03_eig_variants.txt): removed in round 5.)
// ---- 128 < d <= 512, panel form.  The matrix stays in global memory (read and written by one workgroup only) and is
// overwritten: row k keeps the reflector of step k right of the sub-diagonal.  A plain Householder step sweeps the trailing
// matrix three times (one read for B v, a read and a write for the rank-2 update): 8 d^3 bytes per matrix, 1.07 GB at d = 512, and a batch of a few
// thousand 2 MB matrices is far beyond any cache -- the solver is bound by HBM.  Here the rank-2 updates of EWP_NB
// consecutive steps are deferred (LAPACK's dlatrd idea): the panel's reflectors v_s and vectors w_s stay in LDS, column k
// and B v of the *current* matrix are formed as "stored matrix minus the panel's corrections"
//     a_k = A0[:,k] - sum_s (v_s w_s[k] + w_s v_s[k]),      B v = A0 v - sum_s (v_s (w_s.v) + w_s (v_s.v)),
// and the trailing matrix is rewritten once per panel (A0 -= V W^T + W V^T).  Both sweeps touch the 32 x 32 tiles on and
// below the diagonal only (B v reads a tile once for both products of the symmetric pair).  Traffic: half a read per step
// + half a read and write per panel = (1/2 + 1/NB)/3 of the above.  Same reflector convention (v[0] = 1, kept in row k right of the
// sub-diagonal), same Sturm / inverse iteration / back-transformation.  diag[k], beta[k], off[k] are parked in the dead part
// of column k (A[k][k], A[k+1][k], A[k+2][k]) until the tridiagonal solve collects them.
// 512 threads: thread t owns index t of every O(d) vector; for the O(d^2) sweeps the threads are re-mapped every step to
// (column, row slice) over the 64-aligned window of live columns, so that all of them keep loading as the block shrinks.
// v[0..4) combined across the 8 lanes that differ in lane bits SH, SH+1, SH+2: afterwards the lane whose bits SH+1, SH
// spell q holds the sum over the 8 lanes of v[q] (the two lanes that differ in bit SH+2 hold the same value).  Fixed tree.
template <int SH>
__device__ __forceinline__ double transpose_reduce4(const double (&v)[4], int lane) {
    const bool h1 = (lane >> (SH + 1)) & 1, h0 = (lane >> SH) & 1;
    double w2[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const double keep = h1 ? v[q + 2] : v[q], send = h1 ? v[q] : v[q + 2];
        w2[q] = keep + __shfl_xor(send, 2 << SH);
    }
    const double keep = h0 ? w2[1] : w2[0], send = h0 ? w2[0] : w2[1];
    const double w1 = keep + __shfl_xor(send, 1 << SH);
    return w1 + __shfl_xor(w1, 4 << SH);
}
#ifndef EWP_OCC
#define EWP_OCC 4
#endif
constexpr int EWP_NB = 8, EWP_T = 512;
constexpr int EWP_SCR = 16 * 16 * 32; // partial vectors of the 32 x 32 tile sweep: [row block][other block][32]
// The workgroup's barrier of the panel solver: its waves hand each other data through GLOBAL memory (the matrix, the partial
// vectors of the tile sweep), and the compiler's barrier waits for LDS traffic only -- it relies on the CU's memory pipeline
// keeping one wave's store ahead of another wave's later load.  Round 4 saw exactly that assumption fail for LDS
// (group_eig_kernel); here the store queue is drained explicitly (tests/test_device_asm.py scans for it).
__device__ __forceinline__ void panel_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ double block_sum_512(double v, double *red8) {
    v = wave_allsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red8[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red8[0] + red8[1]) + (red8[2] + red8[3])) + ((red8[4] + red8[5]) + (red8[6] + red8[7]));
}
__global__ __launch_bounds__(EWP_T, EWP_OCC) void group_eig_panel_kernel(double *__restrict__ cov, int d, double *__restrict__ vec,
                                                                   double *__restrict__ scratch /* [task][EWP_SCR] */) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *Vp = sh;                // [NB][d] reflectors of the open panel (zero above their sub-diagonal)
    double *Wp = Vp + EWP_NB * d;   // [NB][d]
    double *v = Wp + EWP_NB * d;    // [d] the current reflector; later the iterate of the inverse iteration
    double *red = v + d;            // 8
    double *wpart = red + 8;        // [8 waves][2 NB]
    double *gh = wpart + 8 * 2 * EWP_NB; // [2 NB]
    double *misc = gh + 2 * EWP_NB; // 32
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *A = cov + (size_t)blockIdx.x * d * d;
    double *out = vec + (size_t)blockIdx.x * d;
    double *Pt = scratch + (size_t)blockIdx.x * EWP_SCR;
    const int rq = lane >> 3, cq = lane & 7;
    for (int i = tid; i < 2 * EWP_NB * d; i += EWP_T) Vp[i] = 0.0;
    panel_sync();
    int q = 0;
    for (int k = 0; k + 2 < d; k++) {
        const int o = k + 1;
        // window of live columns [jbase, jbase + CW), jbase 64-aligned; NS row slices
        const int jbase = o & ~63, CW = ((d - jbase) + 63) & ~63, NS = EWP_T / CW;
        const int sl = tid / CW, j = jbase + (tid - sl * CW);
        const bool active = sl < NS && j >= o && j < d;
        // column k of the current matrix (row k of the stored one, it is symmetric)
        double ai = 0.0;
        if (tid >= k && tid < d) {
            // the stored matrix is kept current on and below the diagonal 32 x 32 tiles only: row k inside k's own tile,
            // column k (stride d) below it
            ai = ((tid >> 5) == (k >> 5)) ? A[k * d + tid] : A[tid * d + k];
#pragma unroll
            for (int s = 0; s < EWP_NB; s++)
                if (s < q) ai -= Vp[s * d + tid] * Wp[s * d + k] + Wp[s * d + tid] * Vp[s * d + k];
        }
        if (tid == k) A[k * d + k] = ai; // diag[k]
        if (tid == o) misc[0] = ai;
        const double sigma = block_sum_512((tid > o && tid < d) ? ai * ai : 0.0, red);
        const double alpha = misc[0];
        if (sigma == 0.0) { // uniform: no reflection; the panel column stays zero
            if (tid == 0) { A[o * d + k] = 0.0; A[(o + 1) * d + k] = alpha; }
        } else {
            const double mu = sqrt(alpha * alpha + sigma);
            const double v0 = (alpha <= 0.0) ? alpha - mu : -sigma / (alpha + mu);
            const double bk = 2.0 * v0 * v0 / (sigma + v0 * v0);
            const double vi = (tid == o) ? 1.0 : ((tid > o && tid < d) ? ai / v0 : 0.0);
            if (tid < d) { v[tid] = vi; Vp[q * d + tid] = vi; }
            if (tid > o && tid < d) A[k * d + tid] = vi; // the reflector stays in row k (v[0] = 1 implicit)
            if (tid == 0) { A[o * d + k] = bk; A[(o + 1) * d + k] = mu; }
#pragma unroll
            for (int s = 0; s < EWP_NB; s++) {
                if (s < q) { // g_s = w_s . v, h_s = v_s . v
                    const double g = wave_allsum(tid < d ? Wp[s * d + tid] * vi : 0.0);
                    const double h = wave_allsum(tid < d ? Vp[s * d + tid] * vi : 0.0);
                    if (lane == 0) { wpart[wv * 2 * EWP_NB + 2 * s] = g; wpart[wv * 2 * EWP_NB + 2 * s + 1] = h; }
                }
            }
            panel_sync();
            if (tid < 2 * q) {
                double t = wpart[tid];
                for (int w2 = 1; w2 < 8; w2++) t += wpart[w2 * 2 * EWP_NB + tid];
                gh[tid] = t;
            }
            // B v over the 32 x 32 tiles on and below the diagonal of the stored trailing block (32-grid from jb): every
            // tile is read ONCE and serves both products of the symmetric pair -- its row sums go to the row block, its
            // column sums to the column block (a diagonal tile is read whole and gives row sums only).  Lane 8*rq + cq
            // holds the 4 x 4 block (rows 4rq.., columns 4cq..); the four row (column) partials of a lane are combined over
            // the 8 lanes that share rq (cq) by a transposing butterfly.  v is zero left of column o, so whatever is parked
            // there (reflectors, diag / beta / off) drops out.  Partial vectors Pt[row block][other block][32] in this
            // workgroup's scratch; thread i then adds its block's Gb partials in order.
            const int jb = o & ~31, Gb = (d - jb + 31) >> 5, ntile = Gb * (Gb + 1) / 2;
            for (int t = wv; t < ntile; t += 8) {
                int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while ((I + 1) * (I + 2) / 2 <= t) I++;
                while (I * (I + 1) / 2 > t) I--;
                const int J = t - I * (I + 1) / 2;
                const int r0 = jb + 32 * I + 4 * rq, c0 = jb + 32 * J + 4 * cq;
                double g[4][4];
                if ((d & 1) == 0 && jb + 32 * I + 32 <= d) { // whole tile inside: 16-byte loads
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const d2 *rowp = reinterpret_cast<const d2 *>(A + (r0 + a) * d + c0);
                        const d2 x0 = rowp[0], x1 = rowp[1];
                        g[a][0] = x0[0]; g[a][1] = x0[1]; g[a][2] = x1[0]; g[a][3] = x1[1];
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b2 = 0; b2 < 4; b2++) g[a][b2] = (r0 + a < d && c0 + b2 < d) ? A[(r0 + a) * d + c0 + b2] : 0.0;
                }
                double tr[4], tc[4], pr[4], pc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    tr[u] = r0 + u < d ? v[r0 + u] : 0.0;
                    tc[u] = c0 + u < d ? v[c0 + u] : 0.0;
                    pr[u] = 0.0;
                    pc[u] = 0.0;
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b2 = 0; b2 < 4; b2++) {
                        pr[a] = fma(g[a][b2], tc[b2], pr[a]);
                        pc[b2] = fma(g[a][b2], tr[a], pc[b2]);
                    }
                const double rsum = transpose_reduce4<0>(pr, lane); // row 4*rq + (cq & 3) of the tile
                if (cq < 4) Pt[(I * Gb + J) * 32 + 4 * rq + cq] = rsum;
                if (I != J) {
                    const double csum = transpose_reduce4<3>(pc, lane); // column 4*cq + (rq & 3) of the tile
                    if (rq < 4) Pt[(J * Gb + I) * 32 + 4 * cq + rq] = csum;
                }
            }
            panel_sync();
            double pi = 0.0;
            if (tid >= o && tid < d) {
                const int bi = (tid - jb) >> 5, li = (tid - jb) & 31;
                const double *pp = Pt + bi * Gb * 32 + li;
                double t = pp[0];
                for (int q2 = 1; q2 < Gb; q2++) t += pp[q2 * 32];
#pragma unroll
                for (int s = 0; s < EWP_NB; s++)
                    if (s < q) t -= Vp[s * d + tid] * gh[2 * s] + Wp[s * d + tid] * gh[2 * s + 1];
                pi = bk * t;
            }
            const double K = 0.5 * bk * block_sum_512(pi * vi, red);
            if (tid < d) Wp[q * d + tid] = (tid >= o) ? pi - K * vi : 0.0;
        }
        q++;
        panel_sync();
        if (q == EWP_NB || k + 3 >= d) { // close the panel: the stored trailing block catches up
            if (active) {
                double vj[EWP_NB], wj[EWP_NB];
#pragma unroll
                for (int s = 0; s < EWP_NB; s++) { vj[s] = Vp[s * d + j]; wj[s] = Wp[s * d + j]; }
                double *col = A + j;
                int i = max(o, j & ~31) + sl; // rows of the tiles on and below the diagonal (the sweep reads nothing else)
                for (; i + 3 * NS < d; i += 4 * NS) {
                    double a[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) a[u] = col[(i + u * NS) * d];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int iu = i + u * NS;
#pragma unroll
                        for (int s = 0; s < EWP_NB; s++) a[u] -= Vp[s * d + iu] * wj[s] + Wp[s * d + iu] * vj[s];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) col[(i + u * NS) * d] = a[u];
                }
                for (; i < d; i += NS) {
                    double a = col[i * d];
#pragma unroll
                    for (int s = 0; s < EWP_NB; s++) a -= Vp[s * d + i] * wj[s] + Wp[s * d + i] * vj[s];
                    col[i * d] = a;
                }
            }
            panel_sync();
            for (int i = tid; i < 2 * EWP_NB * d; i += EWP_T) Vp[i] = 0.0;
            q = 0;
            panel_sync();
        }
    }
    panel_sync();
    // the tridiagonal matrix -> LDS (the panel storage is free now)
    double *diag = sh, *off = diag + d, *beta = off + d, *tri = beta + d; // tri: 4 d
    unsigned char *swp = reinterpret_cast<unsigned char *>(tri + 4 * d);
    for (int i = tid; i < d; i += EWP_T) {
        diag[i] = A[i * d + i];
        off[i] = (i + 2 < d) ? A[(i + 2) * d + i] : ((i == d - 2) ? A[(i + 1) * d + i] : 0.0); // the element below the diagonal
        beta[i] = (i + 2 < d) ? A[(i + 1) * d + i] : 0.0;
    }
    panel_sync();
    // ---- Gershgorin bounds ---------------------------------------------------------------------------
    double glo = 1e300, ghi = -1e300, gn = 0.0;
    for (int i = tid; i < d; i += EWP_T) {
        const double rad = (i > 0 ? fabs(off[i - 1]) : 0.0) + (i + 1 < d ? fabs(off[i]) : 0.0);
        glo = fmin(glo, diag[i] - rad);
        ghi = fmax(ghi, diag[i] + rad);
        gn = fmax(gn, fabs(diag[i]) + rad);
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        glo = fmin(glo, __shfl_xor(glo, o2));
        ghi = fmax(ghi, __shfl_xor(ghi, o2));
        gn = fmax(gn, __shfl_xor(gn, o2));
    }
    if (lane == 0) { misc[wv * 3] = glo; misc[wv * 3 + 1] = ghi; misc[wv * 3 + 2] = gn; }
    panel_sync();
    for (int w2 = 0; w2 < 8; w2++) {
        glo = fmin(glo, misc[w2 * 3]);
        ghi = fmax(ghi, misc[w2 * 3 + 1]);
        gn = fmax(gn, misc[w2 * 3 + 2]);
    }
    const double tiny = fmax(gn, 2.2250738585072014e-308) * 2.220446049250313e-16;
    // ---- largest eigenvalue: 64-way multisection on the Sturm count (wave 0) ------------------------------
    if (tid < 64) {
        double lo = glo, hi = ghi + tiny;
        for (int it = 0; it < 64; it++) {
            const double x = lo + (hi - lo) * ((double)(tid + 1) / 65.0);
            int cnt = 0;
            double qq = diag[0] - x;
            if (qq < 0) cnt++;
            for (int i = 1; i < d; i++) {
                if (qq == 0.0) qq = tiny;
                qq = diag[i] - x - (off[i - 1] * off[i - 1]) * fast_rcp(qq);
                if (qq < 0) cnt++;
            }
            const unsigned long long mask = __ballot(cnt >= d);
            double nlo, nhi;
            if (mask == 0ULL) {
                nlo = __shfl(x, 63);
                nhi = hi;
            } else {
                const int f = __ffsll((long long)mask) - 1;
                nhi = __shfl(x, f);
                nlo = (f > 0) ? __shfl(x, f - 1) : lo;
            }
            if (!(nhi > nlo) || (nlo == lo && nhi == hi)) break;
            lo = fmax(lo, nlo);
            hi = fmin(hi, nhi);
        }
        if (tid == 0) red[4] = 0.5 * (lo + hi);
    }
    panel_sync();
    // ---- inverse iteration (lane 0 of wave 0 runs the recurrences, the wave the element-wise parts) -----------------
    if (wv == 0) {
        const double lam = red[4];
        double *dl = tri, *dd = tri + d, *du = tri + 2 * d, *du2 = tri + 3 * d;
        double *y = v;
        for (int i = lane; i < d; i += 64) {
            dd[i] = diag[i] - lam;
            dl[i] = du[i] = (i + 1 < d) ? off[i] : 0.0;
            du2[i] = 0.0;
            swp[i] = 0;
            y[i] = 1.0 + 0.01 * (double)(((unsigned)i * 2654435761u) % 97u) / 97.0;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) { // LU with partial pivoting of the shifted tridiagonal matrix
            double di = dd[0], ui = du[0];
            for (int i = 0; i + 1 < d; i++) {
                const double li = dl[i], dn = dd[i + 1], un = du[i + 1];
                if (fabs(di) >= fabs(li)) {
                    if (di == 0.0) di = tiny;
                    const double f = li * fast_rcp(di);
                    dd[i] = di;
                    dl[i] = f;
                    du[i] = ui;
                    di = dn - f * ui;
                    ui = un;
                } else {
                    const double f = di * fast_rcp(li);
                    dd[i] = li;
                    dl[i] = f;
                    du[i] = dn;
                    di = ui - f * dn;
                    if (i + 2 < d) du2[i] = un;
                    ui = -f * un;
                    swp[i] = 1;
                }
            }
            if (di == 0.0) di = tiny;
            dd[d - 1] = di;
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < d; i += 64) dd[i] = fast_rcp(dd[i]);
        __builtin_amdgcn_wave_barrier();
        for (int it = 0; it < 3; it++) {
            if (lane == 0) {
                double yi = y[0];
                for (int i = 0; i + 1 < d; i++) {
                    const bool sw = swp[i] != 0;
                    const double yn = y[i + 1], li = dl[i];
                    y[i] = sw ? yn : yi;
                    yi = sw ? yi - li * yn : yn - li * yi;
                }
                double y1 = yi * dd[d - 1];
                y[d - 1] = y1;
                double y0 = (y[d - 2] - du[d - 2] * y1) * dd[d - 2];
                y[d - 2] = y0;
                for (int i = d - 3; i >= 0; i--) {
                    const double t = (y[i] - du[i] * y0 - du2[i] * y1) * dd[i];
                    y[i] = t;
                    y1 = y0;
                    y0 = t;
                }
            }
            __builtin_amdgcn_wave_barrier();
            double amax = 0.0;
            for (int i = lane; i < d; i += 64) amax = fmax(amax, fabs(y[i]));
            for (int o2 = 32; o2 > 0; o2 >>= 1) amax = fmax(amax, __shfl_xor(amax, o2));
            if (!(amax > 0.0) || !(amax < 1e300)) {
                for (int i = lane; i < d; i += 64) y[i] = (i == 0) ? 1.0 : 0.0;
                __builtin_amdgcn_wave_barrier();
                break;
            }
            const double ra = 1.0 / amax;
            double part = 0.0;
            for (int i = lane; i < d; i += 64) { const double t = y[i] * ra; part += t * t; }
            const double rn = ra / sqrt(wave_allsum(part));
            for (int i = lane; i < d; i += 64) y[i] *= rn;
            __builtin_amdgcn_wave_barrier();
        }
    }
    panel_sync();
    // ---- back-transformation x = H_0 H_1 ... H_{d-3} y (thread t owns component t) ---------------------------------
    {
        double yt = (tid < d) ? v[tid] : 0.0;
        double vnext = (d >= 3 && tid > d - 2 && tid < d) ? A[(d - 3) * d + tid] : 0.0; // reflector d-3, prefetched
        for (int k = d - 3; k >= 0; k--) {
            const int o = k + 1;
            const double vk = (tid == o) ? 1.0 : ((tid > o && tid < d) ? vnext : 0.0);
            if (k > 0) vnext = (tid > k && tid < d) ? A[(k - 1) * d + tid] : 0.0;
            const double bk = beta[k];
            if (bk == 0.0) continue; // uniform
            const double sc = bk * block_sum_512(vk * yt, red);
            yt -= sc * vk;
        }
        panel_sync();
        if (tid < d) v[tid] = yt;
        panel_sync();
    }
    // normalise; sign: the component of largest magnitude (the first one on ties) is positive
    double part = 0.0, best = -1.0;
    int bi = 0;
    for (int i = tid; i < d; i += EWP_T) {
        const double t = v[i];
        part += t * t;
        if (fabs(t) > best) { best = fabs(t); bi = i; }
    }
    double nrm = sqrt(block_sum_512(part, red));
    const bool degenerate = !(nrm > 0.0) || !(nrm < 1e300);
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        const double ob = __shfl_xor(best, o2);
        const int oi = __shfl_xor(bi, o2);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    panel_sync();
    if (lane == 0) { misc[2 * wv] = best; misc[2 * wv + 1] = (double)bi; }
    panel_sync();
    best = misc[0];
    bi = (int)misc[1];
    for (int w2 = 1; w2 < 8; w2++) {
        const double ob = misc[2 * w2];
        const int oi = (int)misc[2 * w2 + 1];
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (degenerate) {
        for (int i = tid; i < d; i += EWP_T) out[i] = (i == 0) ? 1.0 : 0.0;
        return;
    }
    const double sg = ((v[bi] < 0.0) ? -1.0 : 1.0) / nrm;
    for (int i = tid; i < d; i += EWP_T) out[i] = v[i] * sg;
}
bool k_group_eig(cge_ctx *c, const double *cov, i64 n_tasks, i64 d, double *vec) {
    if (d > 512) return false; // beyond the LDS budget of the wide solver: the caller uses the host solver
    if (d > 128) { // the matrix stays in global memory and is overwritten
        ScopedKernelTimer t(c, "group_eig");
        const size_t plds = (size_t)(2 * EWP_NB * d + d + 8 + 8 * 2 * EWP_NB + 2 * EWP_NB + 32) * sizeof(double);
        cge_allow_lds((const void *)group_eig_panel_kernel, 160 * 1024);
        c->ls_eigscr.ensure((size_t)n_tasks * EWP_SCR);
        hipLaunchKernelGGL(group_eig_panel_kernel, dim3((unsigned)n_tasks), dim3(EWP_T), plds, c->stream,
                           const_cast<double *>(cov), (int)d, vec, c->ls_eigscr.p);
        return true;
    }
    ScopedKernelTimer t(c, "group_eig");
    static const int diag_stage = getenv("CGE_EIG_DIAG") ? atoi(getenv("CGE_EIG_DIAG")) : 0; // timing diagnostics: 0 = normal
    const dim3 grid((unsigned)n_tasks), block(256);
    // blocked columns, reflectors in place, the column values of the two O(d^2) loops as DPP broadcasts of the FMAs
    if (d <= 32) hipLaunchKernelGGL((group_eig_kernel<1, 8, true>), grid, block, 0, c->stream, cov, (int)d, vec, diag_stage);
    else if (d <= 64) hipLaunchKernelGGL((group_eig_kernel<1, 16, true>), grid, block, 0, c->stream, cov, (int)d, vec, diag_stage);
    else hipLaunchKernelGGL((group_eig_kernel<2, 32, true>), grid, block, 0, c->stream, cov, (int)d, vec, diag_stage);
    return true;
}

// projection z_j = sum_c ((x_jc - mu_c) * sqrt(w_j)) * v_c
// d <= 128: one wave per PR = 16 consecutive rows of the batch.  Lanes 0..15 fetch the rows' vertex ids, tasks and weights
// (one coalesced request each instead of three per row), the ids travel to scalar registers by v_readlane, and all 32 row
// loads of the wave (two 512-byte requests per row: columns lane and lane + 64, the per-lane split the sum has always had)
// are in flight together; the task's mean and direction are fetched once per wave (rows are in task order: a second task
// inside a wave's 16 rows is the exception and reloads).  Per row the same two FMAs per lane and the same tree as ever.
__global__ __launch_bounds__(256) void group_project_kernel(const double *__restrict__ Xr, const double *__restrict__ vw,
                                                            const i32 *__restrict__ rows, const i32 *__restrict__ row_task,
                                                            i64 n_rows, i64 d, const double *__restrict__ mean,
                                                            const double *__restrict__ vec, double *__restrict__ z) {
    constexpr int PR = 16;
    const i64 j0 = (((i64)blockIdx.x * blockDim.x + threadIdx.x) / WAVE) * PR;
    const int lane = threadIdx.x & 63;
    if (j0 >= n_rows) return;
    if (d <= 128) {
        const i64 jl = (j0 + (lane & 15) < n_rows) ? j0 + (lane & 15) : n_rows - 1;
        const i32 vrow = rows[jl], trow = row_task[jl];
        const double sqv = sqrt(vw[vrow]);
        const bool hb = lane + 64 < d, ha = lane < d;
        double xa[PR], xb[PR];
#pragma unroll
        for (int u = 0; u < PR; u++) {
            const i64 v = __builtin_amdgcn_readlane(vrow, u);
            const double *x = Xr + v * d;
            xa[u] = ha ? x[lane] : 0.0;
            xb[u] = hb ? x[lane + 64] : 0.0;
        }
        int cur = __builtin_amdgcn_readlane(trow, 0);
        double ma, mb, ea, eb;
        {
            const double *mu = mean + (i64)cur * d, *ev = vec + (i64)cur * d;
            ma = ha ? mu[lane] : 0.0; ea = ha ? ev[lane] : 0.0;
            mb = hb ? mu[lane + 64] : 0.0; eb = hb ? ev[lane + 64] : 0.0;
        }
        double zs = 0.0; // lane u keeps the result of row u: one coalesced store at the end
#pragma unroll
        for (int u = 0; u < PR; u++) {
            const int t = __builtin_amdgcn_readlane(trow, u);
            if (t != cur) { // wave-uniform
                cur = t;
                const double *mu = mean + (i64)cur * d, *ev = vec + (i64)cur * d;
                ma = ha ? mu[lane] : 0.0; ea = ha ? ev[lane] : 0.0;
                mb = hb ? mu[lane + 64] : 0.0; eb = hb ? ev[lane + 64] : 0.0;
            }
            const double sq = lane_value(sqv, u);
            double s = 0.0;
            if (ha) s = fma((xa[u] - ma) * sq, ea, s);
            if (hb) s = fma((xb[u] - mb) * sq, eb, s);
            s = wave_tree_sum_lane0(s); // the pairs of the shfl_down tree, in VALU lane swaps
            const double s0 = lane_value(s, 0);
            if (lane == u) zs = s0;
        }
        if (lane < PR && j0 + lane < n_rows) z[j0 + lane] = zs;
        return;
    }
    if (d <= 512) { // the same with up to eight columns per lane (column lane + 64 t), two rows in flight at a time
        constexpr int NS = 8, RF = 2;
        const i64 jl = (j0 + (lane & 15) < n_rows) ? j0 + (lane & 15) : n_rows - 1;
        const i32 vrow = rows[jl], trow = row_task[jl];
        const double sqv = sqrt(vw[vrow]);
        int cur = __builtin_amdgcn_readlane(trow, 0);
        double mu[NS], ev[NS];
        auto load_task = [&](int t) {
            const double *m = mean + (i64)t * d, *e = vec + (i64)t * d;
#pragma unroll
            for (int q = 0; q < NS; q++) {
                const i64 col = lane + 64 * q;
                mu[q] = col < d ? m[col] : 0.0;
                ev[q] = col < d ? e[col] : 0.0;
            }
        };
        load_task(cur);
        double zs = 0.0;
#pragma unroll
        for (int u0 = 0; u0 < PR; u0 += RF) {
            double xv[RF][NS];
#pragma unroll
            for (int f = 0; f < RF; f++) {
                const i64 v = __builtin_amdgcn_readlane(vrow, u0 + f);
                const double *x = Xr + v * d;
#pragma unroll
                for (int q = 0; q < NS; q++) {
                    const i64 col = lane + 64 * q;
                    xv[f][q] = col < d ? x[col] : 0.0;
                }
            }
#pragma unroll
            for (int f = 0; f < RF; f++) {
                const int u = u0 + f;
                const int t = __builtin_amdgcn_readlane(trow, u);
                if (t != cur) { cur = t; load_task(cur); } // wave-uniform
                const double sq = lane_value(sqv, u);
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < NS; q++)
                    if (lane + 64 * q < d) s = fma((xv[f][q] - mu[q]) * sq, ev[q], s);
                s = wave_tree_sum_lane0(s);
                const double s0 = lane_value(s, 0);
                if (lane == u) zs = s0;
            }
        }
        if (lane < PR && j0 + lane < n_rows) z[j0 + lane] = zs;
        return;
    }
    for (int u = 0; u < PR && j0 + u < n_rows; u++) {
        const i64 j = j0 + u;
        const i64 v = rows[j];
        const i64 t = row_task[j];
        const double sq = sqrt(vw[v]);
        const double *x = Xr + v * d, *mu = mean + t * d, *ev = vec + t * d;
        double s = 0.0;
        for (i64 col = lane; col < d; col += WAVE) s = fma((x[col] - mu[col]) * sq, ev[col], s);
        s = wave_tree_sum_lane0(s);
        if (lane == 0) z[j] = s;
    }
}
void k_group_project(cge_ctx *c, const double *Xr, const double *vw, const i32 *rows, const i32 *row_task, i64 n_rows,
                     i64 d, const double *mean, const double *vec, double *z) {
    i64 threads = (n_rows + 15) / 16 * WAVE; // one wave per 16 rows
    hipLaunchKernelGGL(group_project_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, Xr, vw,
                       rows, row_task, n_rows, d, mean, vec, z);
}

// ------------------------------------------------------------------------------------------------
// Landmark aggregation, one workgroup per landmark, members in ascending vertex order so every
// sum runs in the reference's order (src/landmarks.jl:391-397, :408-415) with unfused mul/add.
__global__ __launch_bounds__(256) void landmark_aggregate_kernel(const double *__restrict__ Xr,
                                                                 const double *__restrict__ vw,
                                                                 const i32 *__restrict__ comm,
                                                                 const i32 *__restrict__ mem_off,
                                                                 const i32 *__restrict__ mem, i64 d,
                                                                 double *__restrict__ lemb,
                                                                 double *__restrict__ lweight,
                                                                 double *__restrict__ dii, i32 *__restrict__ lcomm) {
    extern __shared__ __attribute__((aligned(16))) double sh[]; // centroid[d] + buf[256]
    double *cen = sh, *buf = sh + d;
    const i64 l = blockIdx.x;
    const i32 b = mem_off[l], e = mem_off[l + 1];
    if (e == b) { // no member here (option shard_rows: another rank's landmark): zeros, which the gather over the ranks fills
        for (i64 col = threadIdx.x; col < d; col += blockDim.x) lemb[l * d + col] = 0.0;
        if (threadIdx.x == 0) { lweight[l] = 0.0; dii[l] = 0.0; lcomm[l] = -1; }
        return;
    }
    double lw = 0.0;
    {
        i32 t = b;
        for (; t + 7 < e; t += 8) { // eight (dependent index -> weight) loads in flight; the additions keep the member order
            double wv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) wv[u] = vw[mem[t + u]];
#pragma unroll
            for (int u = 0; u < 8; u++) lw = __dadd_rn(lw, wv[u]);
        }
        for (; t < e; t++) lw = __dadd_rn(lw, vw[mem[t]]);
    }
    for (i64 col = threadIdx.x; col < d; col += blockDim.x) {
        double acc = 0.0;
        i32 t = b;
        for (; t + 7 < e; t += 8) { // 8 member rows in flight; the additions keep the member order
            double wv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const i64 v = mem[t + u];
                wv[u] = vw[v];
                xv[u] = Xr[v * d + col];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc = __dadd_rn(acc, __dmul_rn(wv[u], xv[u]));
        }
        for (; t < e; t++) {
            const i64 v = mem[t];
            acc = __dadd_rn(acc, __dmul_rn(vw[v], Xr[v * d + col]));
        }
        const double cv = acc / lw; // :402
        cen[col] = cv;
        lemb[l * d + col] = cv;
    }
    __syncthreads();
    double tot = 0.0; // meaningful in thread 0
    for (i32 base = b; base < e; base += 256) {
        const i32 t = base + (i32)threadIdx.x;
        if (t < e) {
            const double *x = Xr + (i64)mem[t] * d;
            double dist = 0.0;
            for (i64 col = 0; col < d; col++) {
                const double df = __dsub_rn(cen[col], x[col]);
                dist = __dadd_rn(dist, __dmul_rn(df, df));
            }
            buf[threadIdx.x] = dist;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = min(256, e - base);
            for (int q = 0; q < cnt; q++) tot = __dadd_rn(tot, buf[q]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        lweight[l] = lw;
        dii[l] = lw > 0 ? sqrt(tot / lw) : tot; // :418-423
        lcomm[l] = (e > b) ? comm[mem[e - 1]] : 0; // last writer wins (:427-429)
    }
}
// The same aggregation without dependent loads on the chains (round 4).  Above, every sum walks the member list with an
// index -> weight / row look-up per step (two memory round trips per eight members, for the weight sum, again for the
// centroid).  Here the ids and weights of 256 members at a time are staged in
// LDS by one coalesced pass and the centroid columns have 16 rows in flight behind a single look-up.  The same additions in
// the same order with the same unfused arithmetic: the same bits.  d <= 1024 (four columns per thread).
#define AG_MC 256 // members staged at a time (= the workgroup)
__global__ __launch_bounds__(256) void landmark_aggregate2_kernel(const double *__restrict__ Xr, const double *__restrict__ vw,
                                                                  const i32 *__restrict__ comm,
                                                                  const i32 *__restrict__ mem_off,
                                                                  const i32 *__restrict__ mem, i64 d,
                                                                  double *__restrict__ lemb, double *__restrict__ lweight,
                                                                  double *__restrict__ dii, i32 *__restrict__ lcomm) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    // cen[d] | bufd[256] | s_w[AG_MC] | s_id[AG_MC]
    double *cen = sh, *bufd = sh + d, *s_w = bufd + 256;
    i32 *s_id = reinterpret_cast<i32 *>(s_w + AG_MC);
    __shared__ double s_lw;
    const int tid = threadIdx.x;
    const i64 l = blockIdx.x;
    const i32 b = mem_off[l], e = mem_off[l + 1];
    if (e == b) { // no member here (option shard_rows: another rank's landmark): zeros, which the gather over the ranks fills
        for (i64 col = tid; col < d; col += 256) lemb[l * d + col] = 0.0;
        if (tid == 0) { lweight[l] = 0.0; dii[l] = 0.0; lcomm[l] = -1; }
        return;
    }
    // ---- weight sum and centroid (:391-402): one pass over the members, 256 at a time -----------------------------------
    double lw = 0.0, acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (i32 base = b; base < e; base += AG_MC) {
        const int cnt = min(AG_MC, e - base);
        if (tid < cnt) {
            const i32 v = mem[base + tid];
            s_id[tid] = v;
            s_w[tid] = vw[v];
        }
        __syncthreads();
        if (tid == 0) { // (the additions keep the member order; the weights come out of LDS eight at a time)
            int t = 0;
            for (; t + 7 < cnt; t += 8) {
                double w8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w8[u] = s_w[t + u];
#pragma unroll
                for (int u = 0; u < 8; u++) lw = __dadd_rn(lw, w8[u]);
            }
            for (; t < cnt; t++) lw = __dadd_rn(lw, s_w[t]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const i64 col = tid + 256 * q;
            if (col >= d) break;
            double a = acc[q];
            int t = 0;
            for (; t + 15 < cnt; t += 16) { // 16 member rows in flight behind one look-up in LDS
                double xv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) xv[u] = Xr[(i64)s_id[t + u] * d + col];
#pragma unroll
                for (int u = 0; u < 16; u++) a = __dadd_rn(a, __dmul_rn(s_w[t + u], xv[u]));
            }
            for (; t < cnt; t++) a = __dadd_rn(a, __dmul_rn(s_w[t], Xr[(i64)s_id[t] * d + col]));
            acc[q] = a;
        }
        __syncthreads(); // the staging area is rewritten by the next chunk
    }
    if (tid == 0) s_lw = lw;
    __syncthreads();
    lw = s_lw;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const i64 col = tid + 256 * q;
        if (col >= d) break;
        const double cv = acc[q] / lw; // :402
        cen[col] = cv;
        lemb[l * d + col] = cv;
    }
    __syncthreads();
    // ---- dii (:408-423): the squared distances of the members to the centroid, summed in member order: a thread per member
    //      walks its row (the rows were read a moment ago: L2).  (A tile of rows staged coalesced in LDS with a thread per row
    //      walking it there was measured slower: 32 rows at a time against 256.)
    double tot = 0.0; // meaningful in thread 0
    for (i32 base = b; base < e; base += 256) {
        const i32 t = base + (i32)tid;
        if (t < e) {
            const double *x = Xr + (i64)mem[t] * d;
            double dist = 0.0;
            i64 col = 0;
            for (; col + 15 < d; col += 16) { // 16 values of the row in flight, then the 16 dependent steps
                double xv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) xv[u] = x[col + u];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const double df = __dsub_rn(cen[col + u], xv[u]);
                    dist = __dadd_rn(dist, __dmul_rn(df, df));
                }
            }
            for (; col < d; col++) {
                const double df = __dsub_rn(cen[col], x[col]);
                dist = __dadd_rn(dist, __dmul_rn(df, df));
            }
            bufd[tid] = dist;
        }
        __syncthreads();
        if (tid == 0) {
            const int cnt = min(256, e - base);
            for (int q = 0; q < cnt; q++) tot = __dadd_rn(tot, bufd[q]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        lweight[l] = lw;
        dii[l] = lw > 0 ? sqrt(tot / lw) : tot; // :418-423
        lcomm[l] = comm[mem[e - 1]];             // last writer wins (:427-429)
    }
}
// the landmark tables to / from one exchange vector: [lemb N x d | lweight N | dii N | lcomm + 1 as a double N]
__global__ void pack_landmarks_kernel(double *__restrict__ lemb, double *__restrict__ lweight, double *__restrict__ dii,
                                      i32 *__restrict__ lcomm, i64 N, i64 d, double *__restrict__ X, int unpack) {
    const i64 total = N * (d + 3), stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        if (e < N * d) { if (unpack) lemb[e] = X[e]; else X[e] = lemb[e]; continue; }
        const i64 q = e - N * d, l = q % N, w = q / N;
        if (w == 0) { if (unpack) lweight[l] = X[e]; else X[e] = lweight[l]; }
        else if (w == 1) { if (unpack) dii[l] = X[e]; else X[e] = dii[l]; }
        else { if (unpack) lcomm[l] = (i32)X[e] - 1; else X[e] = (double)(lcomm[l] + 1); }
    }
}
void k_pack_landmarks(cge_ctx *c, double *lemb, double *lweight, double *dii, i32 *lcomm, i64 N, i64 d, double *X, int unpack) {
    hipLaunchKernelGGL(pack_landmarks_kernel, dim3(grid_for(N * (d + 3), 256, 4096)), dim3(256), 0, c->stream, lemb, lweight, dii,
                       lcomm, N, d, X, unpack);
}
__global__ void scatter_u64_kernel(const uint64_t *__restrict__ src, const i32 *__restrict__ idx, i64 cnt, uint64_t *__restrict__ dst) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) dst[idx[i]] = src[i];
}
void k_scatter_u64(cge_ctx *c, const uint64_t *src, const i32 *idx, i64 cnt, uint64_t *dst) {
    if (cnt > 0) hipLaunchKernelGGL(scatter_u64_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, src, idx, cnt, dst);
}
void k_landmark_aggregate(cge_ctx *c, const double *Xr, const double *vw, const i32 *comm, const i32 *mem_off,
                          const i32 *mem, i64 N, i64 d, double *lemb, double *lweight, double *dii, i32 *lcomm) {
    if (d <= 1024) { // (beyond: the form of rounds 1-3)
        const size_t lds2 = (size_t)(d + 256 + AG_MC) * sizeof(double) + (size_t)AG_MC * sizeof(i32);
        hipLaunchKernelGGL(landmark_aggregate2_kernel, dim3((unsigned)N), dim3(256), lds2, c->stream, Xr, vw, comm, mem_off, mem, d,
                           lemb, lweight, dii, lcomm);
        return;
    }
    size_t lds = (size_t)(d + 256) * sizeof(double);
    hipLaunchKernelGGL(landmark_aggregate_kernel, dim3((unsigned)N), dim3(256), lds, c->stream, Xr, vw, comm, mem_off,
                       mem, d, lemb, lweight, dii, lcomm);
}

// ------------------------------------------------------------------------------------------------
// Per-edge scatter: wedges[min(lu,lv), max(lu,lv)] += w  and  vect_C[bin(c_u,c_v)] += w.
// Coalesced SoA reads of (src,dst,w); v2l/comm gathers are served by L2.  Community-pair bins on
// the diagonal (the bulk of the edges of a graph with community structure) are pre-aggregated in
// LDS; everything else goes out as no-return f64 atomics (exact for unit weights).
template <typename CT> // CT = community table element type (uint16 when C < 65536: the table stays L2-resident)
__global__ __launch_bounds__(256) void edge_scatter_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst,
                                                           const double *__restrict__ w, i64 e0, i64 e1,
                                                           const i32 *__restrict__ v2l, const CT *__restrict__ comm,
                                                           i64 N, i64 C, int directed, double *__restrict__ wedges,
                                                           double *__restrict__ vectC) {
    extern __shared__ __attribute__((aligned(16))) double cdiag[]; // C diagonal bins
    for (i64 k = threadIdx.x; k < C; k += blockDim.x) cdiag[k] = 0.0;
    __syncthreads();
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = e0 + (i64)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += stride) {
        const i32 u = src[e], v = dst[e];
        const double we = w ? w[e] : 1.0;
        if (wedges) { // landmark-pair matrix (v2l != nullptr)
            i64 a = v2l[u], b = v2l[v];
            if (!directed && a > b) { i64 t = a; a = b; b = t; }
            unsafeAtomicAdd(&wedges[a * N + b], we);
        }
        if (vectC) {
            i64 cu = comm[u], cv = comm[v];
            if (!directed && cu > cv) { i64 t = cu; cu = cv; cv = t; }
            if (cu == cv)
                unsafeAtomicAdd(&cdiag[cu], we); // LDS atomic: the bulk of a graph with community structure
            else
                unsafeAtomicAdd(&vectC[directed ? cu * C + cv : (C * cu - cu * (cu - 1) / 2 + (cv - cu))], we);
        }
    }
    __syncthreads();
    if (vectC)
        for (i64 k = threadIdx.x; k < C; k += blockDim.x) {
            const double s = cdiag[k];
            if (s != 0.0) unsafeAtomicAdd(&vectC[directed ? k * C + k : (C * k - k * (k - 1) / 2)], s);
        }
}
void k_edge_scatter(cge_ctx *c, const i32 *src, const i32 *dst, const double *w, i64 e0, i64 e1, const i32 *v2l,
                    const i32 *comm, i64 N, i64 C, int directed, double *wedges, double *vectC) {
    if (e1 <= e0) return;
    if (wedges && !v2l) CGE_THROW(CGE_E_ARG, "edge_scatter: the landmark-pair matrix needs v_to_l");
    // two timers: the C x C cluster-pair scatter (the score path) and the N x N landmark-pair scatter
    ScopedKernelTimer t(c, wedges ? (vectC ? "edge_scatter_both" : "edge_scatter_wedges") : "edge_scatter");
    const unsigned grid = grid_for(e1 - e0, 256, 512); // measured: 512 > 256, 1024, 2048
    const bool use16 = comm == c->comm.p && c->comm16.p && C < 65536;
    if (use16)
        hipLaunchKernelGGL(edge_scatter_kernel<unsigned short>, dim3(grid), dim3(256), (size_t)C * sizeof(double), c->stream,
                           src, dst, w, e0, e1, v2l, c->comm16.p, N, C, directed, wedges, vectC);
    else
        hipLaunchKernelGGL(edge_scatter_kernel<i32>, dim3(grid), dim3(256), (size_t)C * sizeof(double), c->stream, src, dst,
                           w, e0, e1, v2l, comm, N, C, directed, wedges, vectC);
}

__global__ void edge_degrees_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst,
                                    const double *__restrict__ w, i64 m, double *__restrict__ deg_out,
                                    double *__restrict__ deg_in, i32 *__restrict__ star) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const double we = w ? w[e] : 1.0;
        unsafeAtomicAdd(&deg_out[src[e]], we);
        unsafeAtomicAdd(&deg_in[dst[e]], we);
        if (star) {
            atomicAdd(&star[src[e]], 1);
            atomicAdd(&star[dst[e]], 1);
        }
    }
}
void k_edge_degrees(cge_ctx *c, const i32 *src, const i32 *dst, const double *w, i64 m, double *deg_out,
                    double *deg_in, i32 *star) {
    hipLaunchKernelGGL(edge_degrees_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, src, dst, w, m, deg_out,
                       deg_in, star);
}

// degrees of the (directed) landmark graph from the landmark-pair matrix: out-degree = row sums, in-degree =
// column sums, star[v] = number of landmark-edge rows (w > 0) in which v appears (src/divergence.jl:311-319).
__global__ __launch_bounds__(256) void wedge_degrees_kernel(const double *__restrict__ wedges, i64 N,
                                                            double *__restrict__ deg_out, double *__restrict__ deg_in,
                                                            i32 *__restrict__ star) {
    __shared__ double sh[256];
    __shared__ int shc[256];
    const i64 a = blockIdx.x;
    double ro = 0.0, ci = 0.0;
    int cnt = 0;
    for (i64 b = threadIdx.x; b < N; b += 256) {
        const double wr = wedges[a * N + b], wc = wedges[b * N + a];
        ro += wr;
        ci += wc;
        cnt += (wr > 0) + (wc > 0);
    }
    sh[threadIdx.x] = ro;
    shc[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[threadIdx.x] += sh[threadIdx.x + s]; shc[threadIdx.x] += shc[threadIdx.x + s]; }
        __syncthreads();
    }
    const double row_sum = sh[0];
    const int count = shc[0];
    __syncthreads();
    sh[threadIdx.x] = ci;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) { deg_out[a] = row_sum; deg_in[a] = sh[0]; star[a] = count; }
}
void k_wedge_degrees(cge_ctx *c, const double *wedges, i64 N, double *deg_out, double *deg_in, i32 *star) {
    hipLaunchKernelGGL(wedge_degrees_kernel, dim3((unsigned)N), dim3(256), 0, c->stream, wedges, N, deg_out, deg_in, star);
}

// reference point r = weighted mean (weights lw) of the landmark centroids listed in ref_mem[ref_off[r]..)
__global__ void ref_centroids_kernel(const double *__restrict__ mu, const double *__restrict__ lw,
                                     const i32 *__restrict__ ref_off, const i32 *__restrict__ ref_mem, i64 d,
                                     double *__restrict__ out) {
    const i64 r = blockIdx.x;
    const i32 b = ref_off[r], e = ref_off[r + 1];
    double wsum = 0.0;
    for (i32 t = b; t < e; t++) wsum += lw[ref_mem[t]];
    for (i64 k = threadIdx.x; k < d; k += blockDim.x) {
        double s = 0.0;
        for (i32 t = b; t < e; t++) s += lw[ref_mem[t]] * mu[(i64)ref_mem[t] * d + k];
        out[r * d + k] = wsum > 0 ? s / wsum : 0.0;
    }
}
void k_ref_centroids(cge_ctx *c, const double *mu, const double *lw, const i32 *ref_off, const i32 *ref_mem, i64 nref,
                     i64 d, double *out) {
    hipLaunchKernelGGL(ref_centroids_kernel, dim3((unsigned)nref), dim3(128), 0, c->stream, mu, lw, ref_off, ref_mem, d, out);
}

__global__ void compact_count_kernel(const double *__restrict__ wedges, i64 N, int directed, i64 row0, i64 row1,
                                     unsigned long long *__restrict__ count) {
    const i64 total = row1 * N, stride = (i64)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (i64 e = row0 * N + (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 a = e / N, b = e - a * N;
        if ((directed || b >= a) && wedges[e] > 0) local++;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}
// positive entries of the rows [row0, row1) (row1 < 0: all rows); upper triangle incl. the diagonal when undirected
void k_compact_count(cge_ctx *c, const double *wedges, i64 N, int directed, i64 *count, i64 row0, i64 row1) {
    if (row1 < 0) { row0 = 0; row1 = N; }
    HIP_CHECK(hipMemsetAsync(count, 0, sizeof(i64), c->stream));
    if (row1 > row0)
        hipLaunchKernelGGL(compact_count_kernel, dim3(grid_for((row1 - row0) * N, 256)), dim3(256), 0, c->stream, wedges, N, directed,
                           row0, row1, reinterpret_cast<unsigned long long *>(count));
}
// degrees / star counts of the landmark graph (wedge_degrees_kernel) from a ROW BLOCK [row0, row1) of the landmark-pair matrix
// (the matrix was reduce-scattered over the ranks): out[0..N) row sums of the block's rows (0 elsewhere), out[N..2N) this
// block's share of every column sum, out[2N..3N) its share of the star counts; the ranks add the three vectors.
__global__ __launch_bounds__(256) void wedge_degrees_block_kernel(const double *__restrict__ wedges, i64 N, i64 row0, i64 row1,
                                                                  double *__restrict__ out) {
    __shared__ double sh[256];
    __shared__ int shc[256];
    const i64 a = blockIdx.x; // a < N: row a (if in the block); a >= N: column a - N over the block's rows
    double acc = 0.0;
    int cnt = 0;
    if (a < N) {
        if (a >= row0 && a < row1)
            for (i64 b = threadIdx.x; b < N; b += 256) { const double w = wedges[a * N + b]; acc += w; cnt += w > 0; }
    } else {
        const i64 col = a - N;
        for (i64 r = row0 + threadIdx.x; r < row1; r += 256) { const double w = wedges[r * N + col]; acc += w; cnt += w > 0; }
    }
    sh[threadIdx.x] = acc;
    shc[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[threadIdx.x] += sh[threadIdx.x + s]; shc[threadIdx.x] += shc[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[a] = sh[0];
        // the star count of a vertex = positive entries of its row + of its column: both halves into the same slot
        atomicAdd(&out[2 * N + (a < N ? a : a - N)], (double)shc[0]);
    }
}
__global__ void degrees_unpack_kernel(const double *__restrict__ in, i64 N, double *__restrict__ deg_out, double *__restrict__ deg_in,
                                      i32 *__restrict__ star) {
    const i64 a = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N) return;
    deg_out[a] = in[a];
    deg_in[a] = in[N + a];
    star[a] = (i32)in[2 * N + a];
}
void k_wedge_degrees_block(cge_ctx *c, const double *wedges, i64 N, i64 row0, i64 row1, double *out3N) {
    HIP_CHECK(hipMemsetAsync(out3N, 0, sizeof(double) * 3 * N, c->stream));
    hipLaunchKernelGGL(wedge_degrees_block_kernel, dim3((unsigned)(2 * N)), dim3(256), 0, c->stream, wedges, N, row0, row1, out3N);
}
void k_degrees_unpack(cge_ctx *c, const double *in3N, i64 N, double *deg_out, double *deg_in, i32 *star) {
    hipLaunchKernelGGL(degrees_unpack_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, in3N, N, deg_out, deg_in, star);
}
