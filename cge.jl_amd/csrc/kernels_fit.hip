// kernels_fit.hip -- the alpha sweep on the device.
//   k_pow_matrix     GD = (1 - D)^alpha                               src/divergence.jl:142-148 (:426-432)
//   k_fit_symv_dir/_update_dir  directed (Tin/Tout, adaptive eps)     src/divergence.jl:434-467
//   k_bvec           vect_B = community-pair sums of P                src/divergence.jl:226-234 (:530-538)
//   k_js             JS(vect_C, vect_B[, vI])                         src/auxilary.jl:34-52
//   k_auc_*          1 - AUC tallies over sampled pairs               src/divergence.jl:178-213 (:478-517)
//   k_mark_edge_hits non-edge rejection for the sampler               src/divergence.jl:137 (NE = all pairs \ E)
// All reductions use a fixed order (no float atomics) so a run is bitwise reproducible.
#include "common.hpp"

#define WAVE 64

__device__ __forceinline__ double block_sum_256(double v, double *sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

#include "pow_parts.hpp"
__global__ void pow_matrix_kernel(const double *__restrict__ D, i64 total, double alpha, double *__restrict__ GD) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) GD[e] = pow(1.0 - D[e], alpha);
}
__global__ void log_matrix_kernel(const double *__restrict__ D, i64 total, double *__restrict__ Lh, float *__restrict__ Ll) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) log_parts(D[e], Lh[e], Ll[e]);
}
__global__ void exp2_matrix_kernel(const double *__restrict__ Lh, const float *__restrict__ Ll, i64 total, double alpha,
                                   double *__restrict__ GD) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) GD[e] = exp2_parts(alpha, Lh[e], Ll[e]);
}
// upper triangle only (64 x 64 tiles with tile-row <= tile-column): what the persistent fit and vect_B read
__global__ __launch_bounds__(256) void pow_matrix_upper_kernel(const double *__restrict__ D, i64 N, double alpha,
                                                               double *__restrict__ GD) {
    const i64 I = blockIdx.y, J = blockIdx.x;
    if (J < I) return;
    const i64 c0 = J * 64 + (threadIdx.x & 63);
    if (c0 >= N) return;
    for (i64 r = I * 64 + (threadIdx.x >> 6); r < std::min<i64>(N, I * 64 + 64); r += 4) {
        const i64 e = r * N + c0;
        GD[e] = pow(1.0 - D[e], alpha);
    }
}
__global__ __launch_bounds__(256) void log_matrix_upper_kernel(const double *__restrict__ D, i64 N, double *__restrict__ Lh,
                                                               float *__restrict__ Ll) {
    const i64 I = blockIdx.y, J = blockIdx.x;
    if (J < I) return;
    const i64 c0 = J * 64 + (threadIdx.x & 63);
    if (c0 >= N) return;
    for (i64 r = I * 64 + (threadIdx.x >> 6); r < std::min<i64>(N, I * 64 + 64); r += 4) {
        const i64 e = r * N + c0;
        log_parts(D[e], Lh[e], Ll[e]);
    }
}
__global__ __launch_bounds__(256) void exp2_matrix_upper_kernel(const double *__restrict__ Lh, const float *__restrict__ Ll,
                                                                i64 N, double alpha, double *__restrict__ GD) {
    const i64 I = blockIdx.y, J = blockIdx.x;
    if (J < I) return;
    const i64 c0 = J * 64 + (threadIdx.x & 63);
    if (c0 >= N) return;
    const i64 r0 = I * 64 + (threadIdx.x >> 6), rend = std::min<i64>(N, I * 64 + 64);
    i64 r = r0;
    for (; r + 12 < rend; r += 16) { // four entries of the column requested together (the power is ~40 dependent instructions)
        double lh[4];
        float ll[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { lh[u] = Lh[(r + 4 * u) * N + c0]; ll[u] = Ll[(r + 4 * u) * N + c0]; }
#pragma unroll
        for (int u = 0; u < 4; u++) GD[(r + 4 * u) * N + c0] = exp2_parts(alpha, lh[u], ll[u]);
    }
    for (; r < rend; r += 4) {
        const i64 e = r * N + c0;
        GD[e] = exp2_parts(alpha, Lh[e], Ll[e]);
    }
}
// the same TILE-BLOCKED for the fused persistent fit (kernels_fitp.hip): upper tile t = (I, J), I <= J, in the order the fit deals
// them to its waves; inside a tile the 8 x 8 block of lane 8 rq + cq (rows 8 rq.., columns 8 cq..) is contiguous, row by row:
// element ((t * 64 + lane) * 64 + 8 a + b).  Outside the matrix: 0 (the fit masks those elements).
__global__ __launch_bounds__(256) void log_matrix_tiles_kernel(const double *__restrict__ D, i64 N, int Nt, double *__restrict__ Lh,
                                                               float *__restrict__ Ll) {
    const i64 t = blockIdx.x;
    int I = 0;
    i64 rem = t;
    while (rem >= Nt - I) { rem -= Nt - I; I++; }
    const i64 J = I + rem;
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int lane = e >> 6, ab = e & 63, a = ab >> 3, b = ab & 7, rq = lane >> 3, cq = lane & 7;
        const i64 row = 64 * (i64)I + 8 * rq + a, col = 64 * J + 8 * cq + b;
        double lh = 0.0;
        float ll = 0.f;
        if (row < N && col < N) log_parts(D[row * N + col], lh, ll);
        Lh[t * 4096 + e] = lh;
        Ll[t * 4096 + e] = ll;
    }
}
// log2(1 - D) for the sweep that follows (whole matrix, or the upper tiles only); false = not enough room, k_pow_matrix
// then uses the library pow.  `blocked`: the tile-blocked form for the fused fit ONLY (pow_logs_blocked_N; k_pow_matrix cannot
// read it -- a sweep that leaves the fused path prepares the row-major form then)
void k_pow_prepare(cge_ctx *c, const double *D, i64 N, bool upper_only, bool blocked) {
    c->pow_logs_N = 0;
    c->pow_logs_blocked_N = 0;
    if (!c->opt_pow_exp2) return;
    if (blocked) {
        const i64 Nt = (N + 63) / 64, NT = Nt * (Nt + 1) / 2;
        c->sw_Lh.ensure((size_t)NT * 4096);
        c->sw_Ll.ensure((size_t)NT * 4096);
        ScopedKernelTimer t(c, "pow_log2");
        hipLaunchKernelGGL(log_matrix_tiles_kernel, dim3((unsigned)NT), dim3(256), 0, c->stream, D, N, (int)Nt, c->sw_Lh.p, c->sw_Ll.p);
        c->pow_logs_blocked_N = N;
        return;
    }
    { // 12 bytes per entry of D: only when they fit comfortably (exact mode at n ~ 10^5 keeps two n x n matrices already)
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)N * N * 12;
        const size_t have = c->sw_Lh.n * sizeof(double) + c->sw_Ll.n * sizeof(float);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (need > have && need - have > free_b / 2)) {
            (void)hipGetLastError();
            return;
        }
    }
    c->sw_Lh.ensure((size_t)N * N);
    c->sw_Ll.ensure((size_t)N * N);
    ScopedKernelTimer t(c, "pow_log2");
    if (upper_only) {
        const unsigned nt = (unsigned)((N + 63) / 64);
        hipLaunchKernelGGL(log_matrix_upper_kernel, dim3(nt, nt), dim3(256), 0, c->stream, D, N, c->sw_Lh.p, c->sw_Ll.p);
    } else
        hipLaunchKernelGGL(log_matrix_kernel, dim3(grid_for(N * N, 256, 256 * 16)), dim3(256), 0, c->stream, D, N * N,
                           c->sw_Lh.p, c->sw_Ll.p);
    c->pow_logs_N = N;
    c->pow_logs_upper = upper_only;
}
void k_pow_matrix(cge_ctx *c, const double *D, i64 N, double alpha, double *GD, bool upper_only) {
    ScopedKernelTimer t(c, "pow_matrix");
    const bool logs = c->pow_logs_N == N && (upper_only || !c->pow_logs_upper);
    const unsigned nt = (unsigned)((N + 63) / 64);
    if (upper_only) {
        if (logs)
            hipLaunchKernelGGL(exp2_matrix_upper_kernel, dim3(nt, nt), dim3(256), 0, c->stream, c->sw_Lh.p, c->sw_Ll.p, N, alpha, GD);
        else
            hipLaunchKernelGGL(pow_matrix_upper_kernel, dim3(nt, nt), dim3(256), 0, c->stream, D, N, alpha, GD);
        return;
    }
    if (logs)
        hipLaunchKernelGGL(exp2_matrix_kernel, dim3(grid_for(N * N, 256, 256 * 16)), dim3(256), 0, c->stream, c->sw_Lh.p,
                           c->sw_Ll.p, N * N, alpha, GD);
    else
        hipLaunchKernelGGL(pow_matrix_kernel, dim3(grid_for(N * N, 256, 256 * 16)), dim3(256), 0, c->stream, D, N * N,
                           alpha, GD);
}
// testing: element-wise (1 - x)^alpha of a host vector by either method
__global__ void pow_test_kernel(const double *__restrict__ x, i64 n, double alpha, int method, double *__restrict__ out) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (method == 0) { out[e] = pow(1.0 - x[e], alpha); return; }
    double Lh;
    float Ll;
    log_parts(x[e], Lh, Ll);
    out[e] = exp2_parts(alpha, Lh, Ll);
}
void k_pow_test(cge_ctx *c, const double *x, i64 n, double alpha, int method, double *out) {
    DevBuf<double> dx, dout;
    dx.ensure(n);
    dout.ensure(n);
    HIP_CHECK(hipMemcpyAsync(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pow_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dx.p, n, alpha, method, dout.p);
    HIP_CHECK(hipMemcpyAsync(out, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

// ------------------------------------------------------------------------------------------------
typedef double dbl2 __attribute__((ext_vector_type(2)));
// (The undirected launch-per-iteration fit lives in kernels_fitp.hip -- fit_symtile_kernel + fit_symreduce_kernel, over the
// upper tiles only; the whole-row kernel of rounds 1-3 was removed in round 5.)
// directed: Sin_i = sum_j (Tin_i*Tout_j)*g_ij + diagonal once more; Sout_i = sum_j (Tin_j*Tout_i)*g_ij + diagonal
// once more (the reference's i..N inner loop visits j == i and adds both tmp1 and tmp2, :439-449)
__global__ __launch_bounds__(256) void fit_symv_dir_kernel(const double *__restrict__ GD,
                                                           const double *__restrict__ Tin,
                                                           const double *__restrict__ Tout, i64 N,
                                                           double *__restrict__ Sin, double *__restrict__ Sout,
                                                           const int *__restrict__ done) {
    __shared__ double sh[256];
    if (*done) return;
    const i64 i = blockIdx.x;
    const double tin_i = Tin[i], tout_i = Tout[i];
    const double *row = GD + i * N;
    double a = 0.0, b = 0.0;
    for (i64 j = threadIdx.x; j < N; j += 256) {
        const double g = row[j];
        double t1 = (tin_i * Tout[j]) * g, t2 = (Tin[j] * tout_i) * g;
        if (j == i) { t1 += t1; t2 += t2; } // both tmp1 and tmp2 land in Sin[i] and in Sout[i]
        a += t1;
        b += t2;
    }
    // at j == i the reference adds tmp1 + tmp2 to Sin[i] and tmp2 + tmp1 to Sout[i]; tmp1 == tmp2 there
    a = block_sum_256(a, sh);
    b = block_sum_256(b, sh);
    if (threadIdx.x == 0) { Sin[i] = a; Sout[i] = b; }
}
void k_fit_symv_dir(cge_ctx *c, const double *GD, const double *Tin, const double *Tout, i64 N, double *Sin,
                    double *Sout, const int *done) {
    ScopedKernelTimer t(c, "fit_symv");
    hipLaunchKernelGGL(fit_symv_dir_kernel, dim3((unsigned)N), dim3(256), 0, c->stream, GD, Tin, Tout, N, Sin, Sout,
                       done);
}
__global__ __launch_bounds__(1024) void fit_update_dir_kernel(double *__restrict__ Tin, double *__restrict__ Tout,
                                                              const double *__restrict__ Sin,
                                                              const double *__restrict__ Sout,
                                                              const double *__restrict__ deg_in,
                                                              const double *__restrict__ deg_out, i64 N, double delta,
                                                              int *__restrict__ done, int *__restrict__ iters,
                                                              double *__restrict__ state) {
    __shared__ double sh[1024];
    if (*done) return;
    const double eps = state[0];
    double f = 0.0;
    for (i64 i = threadIdx.x; i < N; i += 1024) {
        const double di = deg_in[i], dout = deg_out[i];
        if (di > 0) {
            const double t = Tin[i], s = Sin[i];
            Tin[i] = t + (eps * t) * (di / s - 1.0);
            f = fmax(f, fabs(di - s));
        }
        if (dout > 0) {
            const double t = Tout[i], s = Sout[i];
            Tout[i] = t + (eps * t) * (dout / s - 1.0);
            f = fmax(f, fabs(dout - s));
        }
    }
    sh[threadIdx.x] = f;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double fv = sh[0];
        if (fv > state[1]) state[0] = eps * 0.99; // :462-464
        state[1] = fv;
        *iters += 1;
        if (!(fv > delta)) *done = 1;
    }
}
void k_fit_update_dir(cge_ctx *c, double *Tin, double *Tout, const double *Sin, const double *Sout,
                      const double *deg_in, const double *deg_out, i64 N, double delta, int *done, int *iters,
                      double *state) {
    hipLaunchKernelGGL(fit_update_dir_kernel, dim3(1), dim3(1024), 0, c->stream, Tin, Tout, Sin, Sout, deg_in,
                       deg_out, N, delta, done, iters, state);
}

// ------------------------------------------------------------------------------------------------
// relabelling of a score graph (exact mode, wgcl_host.cpp): rows / entries in a new order, index arrays through a map
__global__ void permute_rows_kernel(const double *__restrict__ src, const i32 *__restrict__ order, i64 n, i64 width,
                                    double *__restrict__ dst) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * width) return;
    const i64 q = e / width, col = e - q * width;
    dst[e] = src[(i64)order[q] * width + col];
}
__global__ void permute_i32_kernel(const i32 *__restrict__ src, const i32 *__restrict__ order, i64 n, i32 *__restrict__ dst) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = src[order[q]];
}
__global__ void remap_i32_kernel(i32 *__restrict__ idx, const i32 *__restrict__ map, i64 n) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) idx[k] = map[idx[k]];
}
void k_permute_rows(cge_ctx *c, const double *src, const i32 *order, i64 n, i64 width, double *dst) {
    hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((n * width + 255) / 256)), dim3(256), 0, c->stream, src, order, n,
                       width, dst);
}
void k_permute_i32(cge_ctx *c, const i32 *src, const i32 *order, i64 n, i32 *dst) {
    hipLaunchKernelGGL(permute_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src, order, n, dst);
}
void k_remap_i32(cge_ctx *c, i32 *idx, const i32 *map, i64 n) {
    hipLaunchKernelGGL(remap_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, idx, map, n);
}

// ------------------------------------------------------------------------------------------------
// vect_B.  Stage 1: rowbins[i][c] = sum over the members j of community c (ascending; j >= i when
// undirected) of (Ta_i*Tb_j)*GD_ij.  Stage 2: sum the rows of each community, then fold the two orientations.
// The row is streamed once, coalesced, into LDS as the products (Ta_i*Tb_j)*GD_ij (only j >= i when undirected),
// each product stored at the position of j in the community-sorted member list (cm_pos = the inverse of cm_mem):
// the sum of community c is then a contiguous LDS range in member (= ascending j) order -- the same additions in the
// same order as a direct gather from the row (the skipped j < i hold 0.0, which leaves a sum's bits alone), without
// scattered global reads or dependent index loads.  STAGED = false: rows beyond the LDS budget.
// MODE 1 (STAGED): the whole row fits the LDS budget.  MODE 0: the plain gather (rows beyond it whose communities are not
// ranges of consecutive vertices, and the testing option).  Exact mode with N > 8192 relabels the score graph so that
// every community IS a range (wgcl_host.cpp) and uses bvec_rows_contig_kernel below.
template <int MODE>
__global__ __launch_bounds__(256) void bvec_rows_kernel(const double *__restrict__ GD, const double *__restrict__ Ta,
                                                        const double *__restrict__ Tb,
                                                        const i32 *__restrict__ cm_off, const i32 *__restrict__ cm_mem,
                                                        const i32 *__restrict__ cm_pos, i64 N, i64 C, int directed,
                                                        double *__restrict__ rowbins) {
    extern __shared__ __attribute__((aligned(16))) double prod[];
    const i64 i = blockIdx.x;
    const double ti = Ta[i];
    const double *row = GD + i * N;
    const i64 j0 = directed ? 0 : i;
    if (MODE == 1) {
        i32 ob[2], oe[2]; // the member ranges of this thread's first two communities: asked for ahead of the barrier
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const i64 cc = threadIdx.x + 256 * it;
            ob[it] = cc < C ? cm_off[cc] : 0;
            oe[it] = cc < C ? cm_off[cc + 1] : 0;
        }
        // one pass, eight elements per thread in flight: the skipped j < i store the 0.0 that leaves a sum's bits alone
        const i64 jlo = j0 & ~(i64)255;
        // the skipped j < jlo hold 0.0: the whole row of products is cleared first (LDS stores only -- clearing just the
        // positions cm_pos[j] of the skipped j was a serial loop of dependent global loads, longest for the rows that
        // have the least to add)
        if (jlo > 0) {
            for (i64 j = threadIdx.x; j < N; j += 256) prod[j] = 0.0;
            __syncthreads();
        }
        for (i64 base = jlo; base < N; base += 8 * 256) {
            double r[8], t[8];
            i32 pos[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const i64 j = base + 256 * u + threadIdx.x;
                const bool in = j < N, live = in && j >= j0;
                pos[u] = in ? cm_pos[j] : 0;
                r[u] = live ? row[j] : 0.0;
                t[u] = live ? Tb[j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const i64 j = base + 256 * u + threadIdx.x;
                if (j < N) prod[pos[u]] = (j >= j0) ? (ti * t[u]) * r[u] : 0.0;
            }
        }
        __syncthreads();
        for (i64 cc = threadIdx.x, it = 0; cc < C; cc += 256, it++) {
            double s = 0.0;
            const i32 b = it < 2 ? ob[it] : cm_off[cc], e = it < 2 ? oe[it] : cm_off[cc + 1];
            i32 t = b;
            for (; t + 7 < e; t += 8) { // eight products out of LDS, then the eight additions in member order
                double p8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) p8[u] = prod[t + u];
#pragma unroll
                for (int u = 0; u < 8; u++) s += p8[u];
            }
            for (; t < e; t++) s += prod[t];
            rowbins[i * C + cc] = s;
        }
        return;
    }
    for (i64 cc = threadIdx.x; cc < C; cc += 256) {
        double s = 0.0;
        const i32 b = cm_off[cc], e = cm_off[cc + 1];
        for (i32 t = b; t < e; t++) {
            const i64 j = cm_mem[t];
            if (j >= j0) s += (ti * Tb[j]) * row[j];
        }
        rowbins[i * C + cc] = s;
    }
}
// Rows of a score graph whose communities are ranges of consecutive vertices [cm_off[c], cm_off[c+1]): rowbins[i][c] is a
// sum over a contiguous piece of row i.  One workgroup per row, the four waves take the communities in turn: lane-strided
// partial sums (ascending j per lane) and a fixed xor tree -- coalesced, no staging, no dependent index loads; a thread per
// community walking its members one by one (the forms above) is bound by the largest community (1.1 ms per row at
// n = 60 000 with communities of a few thousand members).
__global__ __launch_bounds__(256) void bvec_rows_contig_kernel(const double *__restrict__ GD, const double *__restrict__ Ta,
                                                               const double *__restrict__ Tb,
                                                               const i32 *__restrict__ cm_off, i64 N, i64 C, int directed,
                                                               double *__restrict__ rowbins) {
    const i64 i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double ti = Ta[i];
    const double *row = GD + i * N;
    const i64 j0 = directed ? 0 : i;
    for (i64 cc = wave; cc < C; cc += 4) {
        const i64 b = max((i64)cm_off[cc], j0), e = cm_off[cc + 1];
        double s0 = 0.0, s1 = 0.0;
        i64 j = b + lane;
        for (; j + 64 < e; j += 128) { // two independent loads per operand in flight
            const double g0 = row[j], g1 = row[j + 64], t0 = Tb[j], t1 = Tb[j + 64];
            s0 += (ti * t0) * g0;
            s1 += (ti * t1) * g1;
        }
        if (j < e) s0 += (ti * Tb[j]) * row[j];
        double sv = s0 + s1;
        for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off);
        if (lane == 0) rowbins[i * C + cc] = sv;
    }
}
// Stage 2a: Z[c1][c2] = sum over the members i of community c1 (ascending) of rowbins[i][c2] -- whole rows of
// rowbins, coalesced.  Stage 2b: vect_B[c1, c2] = Z[c1][c2] + Z[c2][c1] for c1 < c2 (both orientations of an
// unordered community pair), Z[c1][c1] on the diagonal; the directed vector is Z itself.
__global__ __launch_bounds__(256) void bvec_zsum_kernel(const double *__restrict__ rowbins, const i32 *__restrict__ cm_off,
                                                        const i32 *__restrict__ cm_mem, i64 C, double *__restrict__ Z,
                                                        int plain) {
    __shared__ i32 mem[256];
    const i64 c1 = blockIdx.x;
    const i32 b = cm_off[c1], e = cm_off[c1 + 1];
    if (C <= 512 && !plain) { // columns threadIdx.x and threadIdx.x + 256: the member ids once per workgroup, eight rows in flight
        double s[2] = {0.0, 0.0};
        for (i32 base = b; base < e; base += 256) {
            __syncthreads();
            if (base + (i32)threadIdx.x < e) mem[threadIdx.x] = cm_mem[base + threadIdx.x];
            __syncthreads();
            const int cnt = min(256, e - base);
            for (int t0 = 0; t0 < cnt; t0 += 8) {
                double v[2][8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const bool in = t0 + u < cnt;
                    const double *r = rowbins + (i64)mem[in ? t0 + u : 0] * C;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const i64 c2 = threadIdx.x + 256 * h;
                        v[h][u] = (in && c2 < C) ? r[c2] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (t0 + u < cnt) { // member order: the additions of the plain loop
                        s[0] += v[0][u];
                        s[1] += v[1][u];
                    }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const i64 c2 = threadIdx.x + 256 * h;
            if (c2 < C) Z[c1 * C + c2] = s[h];
        }
        return;
    }
    for (i64 c2 = threadIdx.x; c2 < C; c2 += 256) {
        double a = 0.0;
        for (i32 t = b; t < e; t++) a += rowbins[(i64)cm_mem[t] * C + c2];
        Z[c1 * C + c2] = a;
    }
}
__global__ void bvec_fold_kernel(const double *__restrict__ Z, i64 C, int directed, double *__restrict__ vectB) {
    const i64 total = C * C, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 c1 = e / C, c2 = e - c1 * C;
        if (directed) { vectB[e] = Z[e]; continue; }
        if (c2 < c1) continue;
        vectB[C * c1 - c1 * (c1 - 1) / 2 + (c2 - c1)] = (c1 == c2) ? Z[e] : Z[e] + Z[c2 * C + c1];
    }
}
// vect_B by TILES (round 4), for a score graph whose communities are ranges of consecutive vertices (the sweep relabels it,
// wgcl_host.cpp).  One workgroup per 64 x 64 tile of GD (the upper tiles when undirected): the tile is read once, coalesced, as
// the products (Ta_i Tb_j) GD_ij into LDS; inside the tile a community pair is a rectangle (row segment x column segment), so
// (a) every row is summed over each column segment (ascending j), (b) every rectangle over its rows (ascending i): one
// partial per (tile, row segment, column segment), at base[tile] + rs * ns[J] + cs.  A second small kernel adds, per bin, the
// partials of the tiles its rectangle touches (I ascending, then J): every sum has a fixed order, nothing is atomic, GD is
// read once, and the row bins (N x C doubles written and read back) are gone: 57 -> ~20 us per alpha at the headline.
// fc[b] / ns[b]: first community and number of communities (empty ones included) of the 64-vertex block b.
__global__ __launch_bounds__(256) void bvec_tile_kernel(const double *__restrict__ GD, const double *__restrict__ Ta,
                                                        const double *__restrict__ Tb, const i32 *__restrict__ cm_off,
                                                        const i32 *__restrict__ fc, const i32 *__restrict__ ns,
                                                        const i32 *__restrict__ base, i64 N, int Nt, int directed,
                                                        double *__restrict__ partial) {
    const int I = blockIdx.y, J = blockIdx.x;
    if (!directed && J < I) return;
    __shared__ double prod[64][65], rp[64][65];
    __shared__ int segI[66], segJ[66]; // the communities' boundaries inside the tile's row / column range
    const int t = threadIdx.x;
    const int nsI = ns[I], nsJ = ns[J], fcI = fc[I], fcJ = fc[J];
    if (t <= nsJ) segJ[t] = min(max(cm_off[fcJ + t], 64 * J), 64 * J + 64) - 64 * J;
    if (t >= 128 && t - 128 <= nsI) segI[t - 128] = min(max(cm_off[fcI + t - 128], 64 * I), 64 * I + 64) - 64 * I;
    { // products of the tile (zero outside the matrix and, undirected, below the diagonal): a wave takes 16 rows, one row
        // per load instruction (64 consecutive doubles: four cache lines), all 16 loads in flight
        const int lane = t & 63, w = t >> 6;
        const i64 j = (i64)64 * J + lane;
        const double tbj = j < N ? Tb[j] : 0.0;
        double g[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const i64 i = (i64)64 * I + 16 * w + u;
            const bool live = i < N && j < N && (directed || j >= i);
            g[u] = live ? GD[i * N + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const i64 i = (i64)64 * I + 16 * w + u;
            const double ti = i < N ? Ta[i] : 0.0;
            prod[16 * w + u][lane] = __dmul_rn(__dmul_rn(ti, tbj), g[u]); // (0 where the element is not live: g is 0 there)
        }
    }
    __syncthreads();
    { // (a) row r over the columns of community fcJ + s: four threads per row take every fourth segment
        // (by 8-column chunks: the columns of a chunk in ascending order, then the chunks in ascending order -- the order in
        // which the fused epilogue of the persistent fit, whose lanes hold 8-column chunks, adds the same products)
        const int r = t >> 2;
        for (int s2 = t & 3; s2 < nsJ; s2 += 4) {
            const int b0 = segJ[s2], b1 = segJ[s2 + 1];
            double acc = 0.0;
            for (int c2 = b0; c2 < b1;) {
                const int ce = min(b1, (c2 & ~7) + 8);
                double ch = prod[r][c2];
                for (int c3 = c2 + 1; c3 < ce; c3++) ch = __dadd_rn(ch, prod[r][c3]);
                acc = (c2 == b0) ? ch : __dadd_rn(acc, ch);
                c2 = ce;
            }
            rp[r][s2] = acc;
        }
    }
    __syncthreads();
    { // (b) the rectangle (fcI + rs) x (fcJ + s): a wave's lanes take neighbouring column segments
        const int s2 = t & 63;
        if (s2 < nsJ)
            for (int rs = t >> 6; rs < nsI; rs += 4) {
                const int a0 = segI[rs], a1 = segI[rs + 1];
                double acc = 0.0;
                for (int r = a0; r < a1; r++) acc = __dadd_rn(acc, rp[r][s2]);
                partial[(i64)base[I * Nt + J] + rs * nsJ + s2] = acc;
            }
    }
}
__global__ __launch_bounds__(256) void bvec_bins_kernel(const double *__restrict__ partial, const i32 *__restrict__ cm_off,
                                                        const i32 *__restrict__ fc, const i32 *__restrict__ ns,
                                                        const i32 *__restrict__ base, i64 C, int Nt, int directed,
                                                        double *__restrict__ vectB) {
    const i64 total = C * C, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 ca = e / C, cb = e - ca * C;
        if (!directed && cb < ca) continue;
        const i32 a0 = cm_off[ca], a1 = cm_off[ca + 1], b0 = cm_off[cb], b1 = cm_off[cb + 1];
        double acc = 0.0;
        if (a1 > a0 && b1 > b0)
            for (int I = a0 >> 6; I <= (a1 - 1) >> 6; I++)
                for (int J = b0 >> 6; J <= (b1 - 1) >> 6; J++) {
                    if (!directed && J < I) continue; // (the mirrored part of a diagonal bin: the reference sums j >= i only)
                    acc = __dadd_rn(acc, partial[(i64)base[I * Nt + J] + (i64)(ca - fc[I]) * ns[J] + (cb - fc[J])]);
                }
        vectB[directed ? e : C * ca - ca * (ca - 1) / 2 + (cb - ca)] = acc;
    }
}
void k_bvec_tiles(cge_ctx *c, const double *GD, const double *Ta, const double *Tb, const i32 *cm_off, i64 N, int directed) {
    ScopedKernelTimer t(c, "bvec");
    const int Nt = (int)((N + 63) / 64);
    hipLaunchKernelGGL(bvec_tile_kernel, dim3((unsigned)Nt, (unsigned)Nt), dim3(256), 0, c->stream, GD, Ta, Tb, cm_off, c->sw_bt_fc.p,
                       c->sw_bt_ns.p, c->sw_bt_base.p, N, Nt, directed, c->sw_bt_part.p);
}
void k_bvec_bins(cge_ctx *c, const i32 *cm_off, i64 N, i64 C, int directed, double *vectB) {
    ScopedKernelTimer t(c, "bvec");
    const int Nt = (int)((N + 63) / 64);
    hipLaunchKernelGGL(bvec_bins_kernel, dim3(grid_for(C * C, 256, 1024)), dim3(256), 0, c->stream, c->sw_bt_part.p, cm_off,
                       c->sw_bt_fc.p, c->sw_bt_ns.p, c->sw_bt_base.p, C, Nt, directed, vectB);
}
void k_bvec(cge_ctx *c, const double *GD, const double *Ta, const double *Tb, const i32 *cm_pos, const i32 *cm_off,
            const i32 *cm_mem, i64 N, i64 C, int directed, double *rowbins, double *vectB) {
    ScopedKernelTimer t(c, "bvec");
    const int plain = c->opt_test_bvec_plain; // testing: the forms for score graphs beyond the LDS budget / 512 communities
    if (c->bvec_blocks && !plain) {
        const int Nt = (int)((N + 63) / 64);
        hipLaunchKernelGGL(bvec_tile_kernel, dim3((unsigned)Nt, (unsigned)Nt), dim3(256), 0, c->stream, GD, Ta, Tb, cm_off, c->sw_bt_fc.p,
                           c->sw_bt_ns.p, c->sw_bt_base.p, N, Nt, directed, c->sw_bt_part.p);
        hipLaunchKernelGGL(bvec_bins_kernel, dim3(grid_for(C * C, 256, 1024)), dim3(256), 0, c->stream, c->sw_bt_part.p, cm_off,
                           c->sw_bt_fc.p, c->sw_bt_ns.p, c->sw_bt_base.p, C, Nt, directed, vectB);
        return;
    }
    if (N * sizeof(double) <= 64 * 1024 && !plain)
        hipLaunchKernelGGL((bvec_rows_kernel<1>), dim3((unsigned)N), dim3(256), N * sizeof(double), c->stream, GD, Ta,
                           Tb, cm_off, cm_mem, cm_pos, N, C, directed, rowbins);
    else if (c->bvec_contig && !plain)
        hipLaunchKernelGGL(bvec_rows_contig_kernel, dim3((unsigned)N), dim3(256), 0, c->stream, GD, Ta, Tb, cm_off, N, C,
                           directed, rowbins);
    else
        hipLaunchKernelGGL((bvec_rows_kernel<0>), dim3((unsigned)N), dim3(256), 0, c->stream, GD, Ta, Tb, cm_off,
                           cm_mem, cm_pos, N, C, directed, rowbins);
    c->sw_zsum.ensure((size_t)C * C);
    hipLaunchKernelGGL(bvec_zsum_kernel, dim3((unsigned)C), dim3(256), 0, c->stream, rowbins, cm_off, cm_mem, C, c->sw_zsum.p,
                       plain);
    hipLaunchKernelGGL(bvec_fold_kernel, dim3(grid_for(C * C, 256)), dim3(256), 0, c->stream, c->sw_zsum.p, C, directed,
                       vectB);
}

// ------------------------------------------------------------------------------------------------
// JS divergence with the +1 prior; one workgroup, fixed-order tree reductions.
// mode 0: all bins; 1: internal (diagonal) bins; 2: external bins.
__device__ __forceinline__ bool js_selected(i64 k, i64 C, int directed, int mode) {
    if (mode == 0) return true;
    bool diag;
    if (directed)
        diag = (k % (C + 1)) == 0;
    else {
        // packed row c starts at off(c) = C*c - c(c-1)/2; k is diagonal iff k == off(c) for some c
        const double b = 2.0 * (double)C + 1.0;
        i64 cc = (i64)((b - sqrt(b * b - 8.0 * (double)k)) * 0.5);
        if (cc < 0) cc = 0;
        if (cc > C - 1) cc = C - 1;
        while (cc > 0 && C * cc - cc * (cc - 1) / 2 > k) cc--;
        while (cc + 1 < C && C * (cc + 1) - (cc + 1) * cc / 2 <= k) cc++;
        diag = (C * cc - cc * (cc - 1) / 2) == k;
    }
    return mode == 1 ? diag : !diag;
}
// three small launches, every sum in a fixed order: (1) per-block partial (sum C, sum B, count),
// (2) per-block partial of the divergence terms (each block first re-adds the stage-1 partials in block
// order, so all blocks use identical normalisers), (3) the final sum.
#define JS_BLOCKS CGE_PARTIAL_BLOCKS
__global__ __launch_bounds__(256) void js_sums_kernel(const double *__restrict__ vC, const double *__restrict__ vB,
                                                      i64 len, i64 C, int directed, int mode,
                                                      double *__restrict__ part /* [JS_BLOCKS][3] */) {
    __shared__ double sh[256];
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < len; k += (i64)gridDim.x * 256)
        if (js_selected(k, C, directed, mode)) { s1 += vC[k]; s2 += vB[k]; cnt += 1.0; }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
    cnt = block_sum_256(cnt, sh);
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = s1; part[3 * blockIdx.x + 1] = s2; part[3 * blockIdx.x + 2] = cnt; }
}
__global__ __launch_bounds__(256) void js_terms_kernel(const double *__restrict__ vC, const double *__restrict__ vB,
                                                       i64 len, i64 C, int directed, int mode,
                                                       const double *__restrict__ part, double *__restrict__ fpart) {
    __shared__ double sh[256];
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int b = 0; b < JS_BLOCKS; b++) { s1 += part[3 * b]; s2 += part[3 * b + 1]; cnt += part[3 * b + 2]; }
    const double sp1 = s1 + cnt, sp2 = s2 + cnt;
    double f = 0.0;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < len; k += (i64)gridDim.x * 256)
        if (js_selected(k, C, directed, mode)) {
            const double p = (vC[k] + 1.0) / sp1, q = (vB[k] + 1.0) / sp2;
            const double m = (p + q) / 2.0;
            f += p * log(p / m) + q * log(q / m);
        }
    f = block_sum_256(f, sh);
    if (threadIdx.x == 0) fpart[blockIdx.x] = f;
}
__global__ void js_final_kernel(const double *__restrict__ fpart, double *__restrict__ out) { // one wave
    const double v = fpart[threadIdx.x]; // JS_BLOCKS == 64: one load per lane, then the sum in block order
    double f = 0.0;
#pragma unroll
    for (int b = 0; b < JS_BLOCKS; b++) f += __shfl(v, b);
    if (threadIdx.x == 0) *out = f / 2.0;
}
// vect_B from the tile partials AND its divergence from vect_C in ONE launch (round 5; undirected, behind the fused fit or
// bvec_tile_kernel): bvec_bins_kernel + js_sums_kernel + js_terms_kernel with their additions in their order, so the bits are
// those of the three launches.  JS_BLOCKS workgroups (co-resident: a quarter of the chip, and the stream's previous kernel has
// finished); block b owns the bins k = 256 b + t + 256 JS_BLOCKS i.  Phase 1: vect_B[k] = the partials of the tiles its
// rectangle touches (I ascending, then J), and the block sums of vect_C / vect_B / count per mode.  One arrival counter
// (monotonic over the launches of a context: `target` = JS_BLOCKS x launches so far) separates the phases; a block that waits
// too long gives up and publishes NaN (the sweep then fails its checks loudly instead of hanging).  Phase 2: js_terms_kernel.
// n_modes = 1: all bins; 2: internal then external bins (--split-global).
// (ca, cb), ca <= cb, of the packed index k: row ca of the packed triangle starts at off(ca) = C ca - ca (ca - 1) / 2
__device__ __forceinline__ void packed_pair(i64 k, i64 C, i64 &ca, i64 &cb) {
    const double bq = 2.0 * (double)C + 1.0;
    ca = (i64)((bq - sqrt(bq * bq - 8.0 * (double)k)) * 0.5);
    if (ca < 0) ca = 0;
    if (ca > C - 1) ca = C - 1;
    while (ca > 0 && C * ca - ca * (ca - 1) / 2 > k) ca--;
    while (ca + 1 < C && C * (ca + 1) - (ca + 1) * ca / 2 <= k) ca++;
    cb = ca + (k - (C * ca - ca * (ca - 1) / 2));
}
// Once per sweep: where the partials of bin k sit.  A community pair's rectangle touches one tile, two or four (a community of
// more than 64 landmarks: more) -- desc[k] = the positions in `partial` of up to four of them, in the order bvec_bins_kernel adds
// them (I ascending, then J), unused slots = `zero_slot` (a position that holds +0.0: adding it leaves the sum's bits alone);
// .w = -2: more than four tiles, the per-alpha kernel walks them as bvec_bins_kernel does.  The per-alpha kernel then needs two
// loads in a row per bin instead of a chain of five.
__global__ __launch_bounds__(256) void bins_desc_kernel(const i32 *__restrict__ cm_off, const i32 *__restrict__ fc,
                                                        const i32 *__restrict__ ns, const i32 *__restrict__ base, i64 C, int Nt,
                                                        i32 zero_slot, int4 *__restrict__ desc) {
    const i64 len = C * (C + 1) / 2;
    const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
    if (k >= len) return;
    i64 ca, cb;
    packed_pair(k, C, ca, cb);
    const i32 a0 = cm_off[ca], a1 = cm_off[ca + 1], b0 = cm_off[cb], b1 = cm_off[cb + 1];
    i32 slot[4] = {zero_slot, zero_slot, zero_slot, zero_slot};
    int cnt = 0;
    if (a1 > a0 && b1 > b0)
        for (int I = a0 >> 6; I <= (a1 - 1) >> 6; I++)
            for (int J = b0 >> 6; J <= (b1 - 1) >> 6; J++) {
                if (J < I) continue; // (the mirrored part of a diagonal bin: the reference sums j >= i only)
                if (cnt < 4) slot[cnt] = base[I * Nt + J] + (i32)(ca - fc[I]) * ns[J] + (i32)(cb - fc[J]);
                cnt++;
            }
    desc[k] = make_int4(slot[0], slot[1], slot[2], cnt > 4 ? -2 : slot[3]);
}
__global__ __launch_bounds__(256) void bins_js_kernel(const double *__restrict__ partial, const i32 *__restrict__ cm_off,
                                                      const i32 *__restrict__ fc, const i32 *__restrict__ ns,
                                                      const i32 *__restrict__ base, i64 C, int Nt, const double *__restrict__ vC,
                                                      double *__restrict__ vectB, int n_modes, double *__restrict__ part,
                                                      unsigned *counter, unsigned target, double *__restrict__ fpart,
                                                      const int4 *__restrict__ desc, const cge_chain_tail tail) {
    __shared__ double sh[256];
    __shared__ int ok_sh;
    const i64 len = C * (C + 1) / 2;
    const int first_mode = n_modes == 1 ? 0 : 1;
    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0}, cnt[2] = {0.0, 0.0};
    // phase 1, eight bins at a time: all descriptors first, then all partials (a bin is two loads in a row, and the eight are
    // independent), then the sums in the order of the one-bin-at-a-time loop
    for (i64 k0 = (i64)blockIdx.x * 256 + threadIdx.x; k0 < len; k0 += (i64)8 * JS_BLOCKS * 256) {
        int4 dsc[8];
        double c_k[8], pv[8][4];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i64 k = k0 + (i64)u * JS_BLOCKS * 256;
            dsc[u] = k < len ? desc[k] : make_int4(0, 0, 0, 0);
            c_k[u] = k < len ? vC[k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool far = dsc[u].w == -2;
            pv[u][0] = partial[dsc[u].x]; pv[u][1] = partial[dsc[u].y]; pv[u][2] = partial[dsc[u].z];
            pv[u][3] = far ? 0.0 : partial[dsc[u].w];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i64 k = k0 + (i64)u * JS_BLOCKS * 256;
            if (k >= len) break;
            double acc;
            i64 ca = 0, cb = 1;
            if (dsc[u].w != -2)
                acc = __dadd_rn(__dadd_rn(__dadd_rn(__dadd_rn(0.0, pv[u][0]), pv[u][1]), pv[u][2]), pv[u][3]);
            else { // a rectangle over more than four tiles (communities of more than 64 landmarks): bvec_bins_kernel's walk
                packed_pair(k, C, ca, cb);
                const i32 a0 = cm_off[ca], a1 = cm_off[ca + 1], b0 = cm_off[cb], b1 = cm_off[cb + 1];
                acc = 0.0;
                for (int I = a0 >> 6; I <= (a1 - 1) >> 6; I++)
                    for (int J = b0 >> 6; J <= (b1 - 1) >> 6; J++) {
                        if (J < I) continue;
                        acc = __dadd_rn(acc, partial[(i64)base[I * Nt + J] + (i64)(ca - fc[I]) * ns[J] + (cb - fc[J])]);
                    }
            }
            vectB[k] = acc;
            if (n_modes == 1) { s1[0] += c_k[u]; s2[0] += acc; cnt[0] += 1.0; }
            else {
                const bool diag = js_selected(k, C, 0, 1);
                const int m = diag ? 0 : 1; // modes 1 (internal) and 2 (external) partition the bins
                s1[m] += c_k[u]; s2[m] += acc; cnt[m] += 1.0;
            }
        }
    }
    for (int m = 0; m < n_modes; m++) {
        const double a = block_sum_256(s1[m], sh), b = block_sum_256(s2[m], sh), c3 = block_sum_256(cnt[m], sh);
        if (threadIdx.x == 0) {
            double *pp = part + (i64)m * 3 * JS_BLOCKS + 3 * blockIdx.x;
            __hip_atomic_store(pp, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 2, c3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- every block's sums are in memory before anybody reads them ----
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        const long long deadline = wall_clock64() + CGE_FIT_TIMEOUT_TICKS;
        for (;;) {
            // (the counter only grows; a difference that has not wrapped means "at least `target` arrivals")
            if ((int)(__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { ok = 1; break; }
            if (wall_clock64() > deadline) break;
            __builtin_amdgcn_s_sleep(2);
        }
        ok_sh = ok;
    }
    __syncthreads();
    const bool ok = ok_sh != 0;
    // the blocks' sums: fetched ONCE per workgroup (one L2-bypassing load per value; every thread fetching all of them was
    // 3 million requests to a dozen cache lines), then added by every thread in block order from LDS
    __shared__ double tot_sh[2 * 3 * JS_BLOCKS];
    for (int i = threadIdx.x; i < n_modes * 3 * JS_BLOCKS; i += 256)
        tot_sh[i] = __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int m = 0; m < n_modes; m++) {
        const int mode = first_mode + m;
        double t1 = 0.0, t2 = 0.0, tc = 0.0;
        const double *pm = tot_sh + m * 3 * JS_BLOCKS;
        for (int b = 0; b < JS_BLOCKS; b++) {
            t1 += pm[3 * b];
            t2 += pm[3 * b + 1];
            tc += pm[3 * b + 2];
        }
        const double sp1 = t1 + tc, sp2 = t2 + tc;
        double f = 0.0;
        for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < len; k += (i64)JS_BLOCKS * 256)
            if (js_selected(k, C, 0, mode)) {
                const double p = (vC[k] + 1.0) / sp1, q = (vectB[k] + 1.0) / sp2; // (vectB[k]: this thread's own store)
                const double mm = (p + q) / 2.0;
                f += p * log(p / mm) + q * log(q / mm);
            }
        f = block_sum_256(f, sh);
        if (threadIdx.x == 0) {
            const double fv = ok ? f : __builtin_nan("");
            fpart[(i64)m * JS_BLOCKS + blockIdx.x] = fv;
            if (tail.host_out) tail.host_out[tail.res_js + m * JS_BLOCKS + blockIdx.x] = fv; // straight to the host's pinned slot
        }
    }
    // ---- the tail of the alpha's chain, folded into this launch (two launches and two boundaries less per alpha) ----
    // the alpha's other scalars (tallies, the shared verdict, the fit's flags: complete before this launch began) to the host ...
    if (tail.host_out && blockIdx.x == 0)
        for (int i = threadIdx.x; i < tail.res_len; i += 256)
            if (i < tail.res_js || i >= tail.res_js + 2 * JS_BLOCKS) tail.host_out[i] = tail.scal[i];
    // ... and the hand-off slots of the NEXT alpha's persistent fit armed (what a memset did in front of every fit)
    if (tail.arm) {
        const uint4 v = make_uint4(tail.arm_word, tail.arm_word, tail.arm_word, tail.arm_word);
        for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < tail.arm_n16; i += (i64)JS_BLOCKS * 256) tail.arm[i] = v;
    }
}
// fpart: n_modes x CGE_PARTIAL_BLOCKS block sums of the divergence terms (the caller adds them in block order, then / 2)
void k_bins_js(cge_ctx *c, const i32 *cm_off, i64 N, i64 C, const double *vC, double *vectB, int n_modes, double *fpart,
               const cge_chain_tail *tail) {
    const cge_chain_tail tl = tail ? *tail : cge_chain_tail{};
    ScopedKernelTimer t(c, "bvec_js");
    const int Nt = (int)((N + 63) / 64);
    c->js_part.ensure(8 * JS_BLOCKS);
    if (!c->js_counter.p) {
        c->js_counter.ensure(32);
        HIP_CHECK(hipMemsetAsync(c->js_counter.p, 0, 32 * sizeof(unsigned), c->stream));
        c->js_launches = 0;
    }
    c->js_launches++;
    hipLaunchKernelGGL(bins_js_kernel, dim3(JS_BLOCKS), dim3(256), 0, c->stream, c->sw_bt_part.p, cm_off, c->sw_bt_fc.p,
                       c->sw_bt_ns.p, c->sw_bt_base.p, C, Nt, vC, vectB, n_modes, c->js_part.p, c->js_counter.p,
                       (unsigned)(c->js_launches * JS_BLOCKS), fpart, reinterpret_cast<const int4 *>(c->sw_bt_desc.p), tl);
}
// once per sweep, after the tile tables are up: the bins' descriptors (positions of their partials) and the +0.0 slot behind
// the partials that unused descriptor entries point to
void k_bins_prepare(cge_ctx *c, const i32 *cm_off, i64 N, i64 C, i64 n_partials) {
    const int Nt = (int)((N + 63) / 64);
    const i64 len = C * (C + 1) / 2;
    c->sw_bt_desc.ensure((size_t)4 * len);
    HIP_CHECK(hipMemsetAsync(c->sw_bt_part.p + n_partials, 0, sizeof(double), c->stream));
    hipLaunchKernelGGL(bins_desc_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->stream, cm_off, c->sw_bt_fc.p, c->sw_bt_ns.p,
                       c->sw_bt_base.p, C, Nt, (i32)n_partials, reinterpret_cast<int4 *>(c->sw_bt_desc.p));
}
// `partials` != nullptr: the CGE_PARTIAL_BLOCKS block sums of the divergence terms go there and the caller adds them (in
// block order, then / 2: what js_final_kernel does) -- the sweep does that on the host, behind the copy it makes anyway
void k_js(cge_ctx *c, const double *vC, const double *vB, i64 len, i64 C, int directed, int mode, double *out,
          double *partials) {
    c->js_part.ensure(4 * JS_BLOCKS);
    ScopedKernelTimer t(c, "js");
    hipLaunchKernelGGL(js_sums_kernel, dim3(JS_BLOCKS), dim3(256), 0, c->stream, vC, vB, len, C, directed, mode,
                       c->js_part.p);
    double *fpart = partials ? partials : c->js_part.p + 3 * JS_BLOCKS;
    hipLaunchKernelGGL(js_terms_kernel, dim3(JS_BLOCKS), dim3(256), 0, c->stream, vC, vB, len, C, directed, mode,
                       c->js_part.p, fpart);
    if (!partials) hipLaunchKernelGGL(js_final_kernel, dim3(1), dim3(64), 0, c->stream, fpart, out);
}

// ------------------------------------------------------------------------------------------------
// Local score tallies: out2 = { sum_k [pos_k > neg_k] * w_k , sum_k w_k }  (:213 / :517)
#define AUC_BLOCKS CGE_PARTIAL_BLOCKS
__global__ __launch_bounds__(256) void auc_landmark_kernel(const double *__restrict__ Ta, const double *__restrict__ Tb,
                                                           const i32 *__restrict__ v2l,
                                                           const double *__restrict__ vw_orig,
                                                           const double *__restrict__ lweight,
                                                           const i32 *__restrict__ pi, const i32 *__restrict__ pj,
                                                           const i32 *__restrict__ ni, const i32 *__restrict__ nj,
                                                           const double *__restrict__ dpos,
                                                           const double *__restrict__ dneg,
                                                           const double *__restrict__ wts, i64 S, double alpha,
                                                           double *__restrict__ part) {
    __shared__ double sh[256];
    double num = 0.0, den = 0.0;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < S; k += (i64)gridDim.x * 256) {
        const i64 i = pi[k], j = pj[k], u = ni[k], v = nj[k];
        const i64 li = v2l[i], lj = v2l[j], lu = v2l[u], lv = v2l[v];
        const double ai = (Ta[li] * vw_orig[i]) / lweight[li], aj = (Tb[lj] * vw_orig[j]) / lweight[lj];
        const double au = (Ta[lu] * vw_orig[u]) / lweight[lu], av = (Tb[lv] * vw_orig[v]) / lweight[lv];
        const double pos = (ai * aj) * pow(1.0 - dpos[k], alpha);
        const double neg = (au * av) * pow(1.0 - dneg[k], alpha);
        const double w = wts[k];
        num += (pos > neg ? 1.0 : 0.0) * w;
        den += w;
    }
    num = block_sum_256(num, sh);
    den = block_sum_256(den, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = num; part[2 * blockIdx.x + 1] = den; }
}
// The parts of auc_landmark_kernel's tally that depend neither on alpha nor on T, once per sweep and sample set, for the
// epilogue of the fused persistent fit (kernels_fitp.hip): idx[4][S] = T's indices of i, j, u, v (through old2new when the sweep
// is relabelled), fac[8][S] = vw_i, lw_li, vw_j, lw_lj, vw_u, lw_lu, vw_v, lw_lv, den[b] = block b's weight sum (the additions of
// auc_landmark_kernel's own `den`).  gridDim.x == AUC_BLOCKS.
__global__ __launch_bounds__(256) void auc_prepare_kernel(const i32 *__restrict__ v2l, const i32 *__restrict__ old2new,
                                                          const double *__restrict__ vw_orig, const double *__restrict__ lweight,
                                                          const i32 *__restrict__ pi, const i32 *__restrict__ pj,
                                                          const i32 *__restrict__ ni, const i32 *__restrict__ nj,
                                                          const double *__restrict__ wts, i64 S, i32 *__restrict__ idx,
                                                          double *__restrict__ fac, double *__restrict__ den_part) {
    __shared__ double sh[256];
    double den = 0.0;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < S; k += (i64)gridDim.x * 256) {
        const i64 v[4] = {pi[k], pj[k], ni[k], nj[k]};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const i64 l = v2l[v[q]];
            idx[q * S + k] = old2new ? old2new[l] : (i32)l;
            fac[(2 * q) * S + k] = vw_orig[v[q]];
            fac[(2 * q + 1) * S + k] = lweight[l];
        }
        den += wts[k];
    }
    den = block_sum_256(den, sh);
    if (threadIdx.x == 0) den_part[blockIdx.x] = den;
}
void k_auc_prepare(cge_ctx *c, const i32 *v2l, const i32 *old2new, const double *vw_orig, const double *lweight, const i32 *pi,
                   const i32 *pj, const i32 *ni, const i32 *nj, const double *wts, i64 S, i32 *aidx, double *afac, double *aden) {
    hipLaunchKernelGGL(auc_prepare_kernel, dim3(AUC_BLOCKS), dim3(256), 0, c->stream, v2l, old2new, vw_orig, lweight, pi, pj, ni, nj,
                       wts, S, aidx, afac, aden);
}
__global__ __launch_bounds__(256) void auc_exact_kernel(const double *__restrict__ GD, const double *__restrict__ Ta,
                                                        const double *__restrict__ Tb, i64 N,
                                                        const i32 *__restrict__ pi, const i32 *__restrict__ pj,
                                                        const i32 *__restrict__ ni, const i32 *__restrict__ nj,
                                                        const double *__restrict__ wts, i64 S,
                                                        double *__restrict__ part) {
    __shared__ double sh[256];
    double num = 0.0, den = 0.0;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < S; k += (i64)gridDim.x * 256) {
        const i64 i = pi[k], j = pj[k], u = ni[k], v = nj[k];
        const double pos = (Ta[i] * Tb[j]) * GD[i * N + j];
        const double neg = (Ta[u] * Tb[v]) * GD[u * N + v];
        const double w = wts[k];
        num += (pos > neg ? 1.0 : 0.0) * w;
        den += w;
    }
    num = block_sum_256(num, sh);
    den = block_sum_256(den, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = num; part[2 * blockIdx.x + 1] = den; }
}
__global__ void auc_final_kernel(const double *__restrict__ part, int nb, double *__restrict__ out2) { // one wave
    const int lane = threadIdx.x; // nb <= 64: one pair of loads per lane, then the sums in block order
    const double a = lane < nb ? part[2 * lane] : 0.0, b2 = lane < nb ? part[2 * lane + 1] : 0.0;
    double num = 0.0, den = 0.0;
    for (int b = 0; b < nb; b++) { num += __shfl(a, b); den += __shfl(b2, b); }
    if (lane == 0) {
        out2[0] = num;
        out2[1] = den;
    }
}
// Many samples (config 4: 10^6 per alpha, two pow() each): AUC_WIDE x as many workgroups tally, and one wave folds
// AUC_WIDE consecutive block tallies into each of the AUC_BLOCKS slots the callers expect, in a fixed order.
#define AUC_WIDE 16
#define AUC_WIDE_MIN_SAMPLES 65536
__global__ void auc_fold_kernel(const double *__restrict__ wide, double *__restrict__ part) { // one wave, AUC_BLOCKS <= 64
    const int s = threadIdx.x;
    if (s >= AUC_BLOCKS) return;
    double num = 0.0, den = 0.0;
    for (int q = 0; q < AUC_WIDE; q++) {
        num += wide[2 * (s * AUC_WIDE + q)];
        den += wide[2 * (s * AUC_WIDE + q) + 1];
    }
    part[2 * s] = num;
    part[2 * s + 1] = den;
}
static double *auc_partials(cge_ctx *c) {
    c->auc_part.ensure(2 * AUC_BLOCKS + 2 * AUC_BLOCKS * AUC_WIDE);
    return c->auc_part.p;
}
void k_auc_landmark(cge_ctx *c, const double *Ta, const double *Tb, const i32 *v2l, const double *vw_orig,
                    const double *lweight, const i32 *pi, const i32 *pj, const i32 *ni, const i32 *nj,
                    const double *dpos, const double *dneg, const double *wts, i64 S, double alpha, double *out2,
                    double *partials) {
    ScopedKernelTimer t(c, "auc_tally");
    double *own = auc_partials(c);
    double *part = partials ? partials : own; // partials: 2 * CGE_PARTIAL_BLOCKS block tallies, summed by the caller
    if (S >= AUC_WIDE_MIN_SAMPLES) {
        double *wide = own + 2 * AUC_BLOCKS;
        hipLaunchKernelGGL(auc_landmark_kernel, dim3(AUC_BLOCKS * AUC_WIDE), dim3(256), 0, c->stream, Ta, Tb, v2l, vw_orig,
                           lweight, pi, pj, ni, nj, dpos, dneg, wts, S, alpha, wide);
        hipLaunchKernelGGL(auc_fold_kernel, dim3(1), dim3(64), 0, c->stream, wide, part);
    } else {
        hipLaunchKernelGGL(auc_landmark_kernel, dim3(AUC_BLOCKS), dim3(256), 0, c->stream, Ta, Tb, v2l, vw_orig, lweight,
                           pi, pj, ni, nj, dpos, dneg, wts, S, alpha, part);
    }
    if (!partials) hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(64), 0, c->stream, part, AUC_BLOCKS, out2);
}
void k_auc_exact(cge_ctx *c, const double *GD, const double *Ta, const double *Tb, i64 N, const i32 *pi,
                 const i32 *pj, const i32 *ni, const i32 *nj, const double *wts, i64 S, double *out2, double *partials) {
    ScopedKernelTimer t(c, "auc_tally");
    double *own = auc_partials(c);
    double *part = partials ? partials : own;
    if (S >= AUC_WIDE_MIN_SAMPLES) {
        double *wide = own + 2 * AUC_BLOCKS;
        hipLaunchKernelGGL(auc_exact_kernel, dim3(AUC_BLOCKS * AUC_WIDE), dim3(256), 0, c->stream, GD, Ta, Tb, N, pi, pj, ni,
                           nj, wts, S, wide);
        hipLaunchKernelGGL(auc_fold_kernel, dim3(1), dim3(64), 0, c->stream, wide, part);
    } else {
        hipLaunchKernelGGL(auc_exact_kernel, dim3(AUC_BLOCKS), dim3(256), 0, c->stream, GD, Ta, Tb, N, pi, pj, ni, nj, wts,
                           S, part);
    }
    if (!partials) hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(64), 0, c->stream, part, AUC_BLOCKS, out2);
}

// ------------------------------------------------------------------------------------------------
// Sampler support: stream the resident edge list once and flag every candidate pair that IS an
// edge.  `table` is an open-addressing set of candidate keys ((i << 32) | j, 0-based; i < j when
// undirected), empty slots = ~0.
__device__ __forceinline__ uint64_t mixk(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__global__ void mark_edge_hits_kernel(const i32 *__restrict__ src, const i32 *__restrict__ dst, i64 m, int directed,
                                      const uint64_t *__restrict__ table, i64 mask, i32 *__restrict__ hit) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        uint64_t a = (uint64_t)(uint32_t)src[e], b = (uint64_t)(uint32_t)dst[e];
        if (!directed && a > b) { uint64_t t = a; a = b; b = t; }
        const uint64_t key = (a << 32) | b;
        i64 slot = (i64)(mixk(key) & (uint64_t)mask);
        for (;;) {
            const uint64_t tk = table[slot];
            if (tk == ~0ULL) break;
            if (tk == key) { hit[slot] = 1; break; }
            slot = (slot + 1) & mask;
        }
    }
}
void k_mark_edge_hits(cge_ctx *c, const i32 *src, const i32 *dst, i64 m, int directed, const uint64_t *table,
                      i64 table_size, i32 *hit) {
    ScopedKernelTimer t(c, "mark_edge_hits");
    hipLaunchKernelGGL(mark_edge_hits_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, src, dst, m, directed,
                       table, table_size - 1, hit);
}

// ------------------------------------------------------------------------------------------------
// The sampler on the device (`sample(E, S)` / `sample(NE, S)`, src/divergence.jl:185-210): the counter-based draws of
// common.hpp, non-edges by rejection -- the candidate pairs of a round go into an open-addressing table, one pass over
// the resident edge list marks the candidates that ARE edges (mark_edge_hits_kernel), those are drawn again with the
// next attempt number.  A sample depends only on (seed, stream, k, attempt): the result does not depend on the order in
// which threads insert or re-queue.
__global__ void draw_pos_kernel(uint64_t seed, uint64_t stream, i64 S, i64 m, i32 *__restrict__ pos) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < S) pos[k] = (i32)cge_bounded(cge_ctr_rand(seed, stream, (uint64_t)k, 0, 0), (uint64_t)m);
}
__global__ void draw_neg_kernel(uint64_t seed, uint64_t stream, i64 n, int directed, const i32 *__restrict__ todo, i64 cnt,
                                const unsigned *__restrict__ attempt, i32 *__restrict__ ni, i32 *__restrict__ nj,
                                unsigned long long *__restrict__ table, i64 mask) {
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cnt) return;
    const i64 k = todo ? todo[idx] : idx;
    const uint64_t a = attempt[k];
    uint64_t i = cge_bounded(cge_ctr_rand(seed, stream, (uint64_t)k, a, 1), (uint64_t)n);
    uint64_t j = cge_bounded(cge_ctr_rand(seed, stream, (uint64_t)k, a, 2), (uint64_t)(n - 1));
    if (j >= i) j++; // uniform over ordered pairs i != j
    if (!directed && i > j) { const uint64_t t = i; i = j; j = t; }
    ni[k] = (i32)i;
    nj[k] = (i32)j;
    const unsigned long long key = (i << 32) | j;
    i64 slot = (i64)(mixk(key) & (uint64_t)mask);
    for (;;) {
        const unsigned long long old = atomicCAS(&table[slot], ~0ULL, key);
        if (old == ~0ULL || old == key) break;
        slot = (slot + 1) & mask;
    }
}
__global__ void check_neg_kernel(const i32 *__restrict__ todo, i64 cnt, const i32 *__restrict__ ni, const i32 *__restrict__ nj,
                                 const unsigned long long *__restrict__ table, i64 mask, const i32 *__restrict__ hit,
                                 unsigned *__restrict__ attempt, i32 *__restrict__ next, unsigned long long *__restrict__ next_cnt) {
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cnt) return;
    const i64 k = todo ? todo[idx] : idx;
    const unsigned long long key = ((unsigned long long)(uint32_t)ni[k] << 32) | (unsigned long long)(uint32_t)nj[k];
    i64 slot = (i64)(mixk(key) & (uint64_t)mask);
    while (table[slot] != key) slot = (slot + 1) & mask; // present: inserted by draw_neg_kernel
    if (hit[slot]) {
        attempt[k] += 1;
        next[atomicAdd(next_cnt, 1ULL)] = (i32)k;
    }
}
// A sharded edge list: the verdicts are exchanged BY SAMPLE INDEX k, which is the same on every rank.  (A key's slot is not:
// draw_neg_kernel places keys by atomicCAS with linear probing, so two keys of one probe run can sit in swapped slots on two
// ranks, and flags added slot by slot would pin one rank's hit on another rank's key.)  flag[k] = "my edges contain the pair
// of sample k" for the samples of this round, 0 elsewhere; the ranks add the flags; check_neg_flag_kernel reads flag[k].
__global__ void collect_hit_kernel(const i32 *__restrict__ todo, i64 cnt, const i32 *__restrict__ ni, const i32 *__restrict__ nj,
                                   const unsigned long long *__restrict__ table, i64 mask, const i32 *__restrict__ hit,
                                   i32 *__restrict__ flag) {
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cnt) return;
    const i64 k = todo ? todo[idx] : idx;
    const unsigned long long key = ((unsigned long long)(uint32_t)ni[k] << 32) | (unsigned long long)(uint32_t)nj[k];
    i64 slot = (i64)(mixk(key) & (uint64_t)mask);
    while (table[slot] != key) slot = (slot + 1) & mask; // present: inserted by draw_neg_kernel
    flag[k] = hit[slot];
}
__global__ void check_neg_flag_kernel(const i32 *__restrict__ todo, i64 cnt, const i32 *__restrict__ flag,
                                      unsigned *__restrict__ attempt, i32 *__restrict__ next, unsigned long long *__restrict__ next_cnt) {
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cnt) return;
    const i64 k = todo ? todo[idx] : idx;
    if (flag[k]) {
        attempt[k] += 1;
        next[atomicAdd(next_cnt, 1ULL)] = (i32)k;
    }
}
// One round of the rejection: candidates for the samples of `todo` (round 0: all), marked against the edges, the rejected ones
// listed in `next` and counted.  The verdict (count) is read back by the caller.
static void draw_round(cge_ctx *c, cge_ctx::DrawPending &P) {
    const i64 n = c->n, m = c->m; // m: the edges resident on this rank
    hipStream_t st = c->stream;
    DevBuf<unsigned> &attempt = c->samp_attempt; // grow-only scratch of the context
    DevBuf<i32> &hit = c->samp_hit;
    DevBuf<unsigned long long> &table = c->samp_table, &count = c->samp_count;
    const i64 cnt = P.cnt, S = P.S;
    i64 tsize = 1024;
    while (tsize < 4 * cnt) tsize <<= 1;
    table.ensure(tsize);
    hit.ensure(tsize);
    HIP_CHECK(hipMemsetAsync(table.p, 0xFF, sizeof(unsigned long long) * tsize, st));
    HIP_CHECK(hipMemsetAsync(hit.p, 0, sizeof(i32) * tsize, st));
    HIP_CHECK(hipMemsetAsync(count.p, 0, sizeof(unsigned long long), st));
    const unsigned g = (unsigned)((cnt + 255) / 256);
    hipLaunchKernelGGL(draw_neg_kernel, dim3(g), dim3(256), 0, st, (uint64_t)P.seed, (uint64_t)P.stream_id, n, P.directed, P.todo, cnt,
                       attempt.p, P.d_ni, P.d_nj, table.p, tsize - 1);
    k_mark_edge_hits(c, c->src.p, c->dst.p, m, P.directed, reinterpret_cast<const uint64_t *>(table.p), tsize, hit.p);
    if (c->edges_sharded) {
        // every rank has marked the candidates that are among ITS edges; the verdicts travel by sample index (0 / 1 in 32-bit
        // words, two to an 8-byte word, at most `world` per word half: no carry) and are added over the ranks
        DevBuf<i32> &flag = c->samp_flag;
        const i64 words = (S + 1) / 2;
        flag.ensure(2 * words);
        HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(i32) * 2 * words, st));
        hipLaunchKernelGGL(collect_hit_kernel, dim3(g), dim3(256), 0, st, P.todo, cnt, P.d_ni, P.d_nj, table.p, tsize - 1, hit.p, flag.p);
        cge_allreduce_dev(c, reinterpret_cast<double *>(flag.p), words, 2);
        hipLaunchKernelGGL(check_neg_flag_kernel, dim3(g), dim3(256), 0, st, P.todo, cnt, flag.p, attempt.p, P.next, count.p);
    } else {
        hipLaunchKernelGGL(check_neg_kernel, dim3(g), dim3(256), 0, st, P.todo, cnt, P.d_ni, P.d_nj, table.p, tsize - 1, hit.p,
                           attempt.p, P.next, count.p);
    }
    c->samp_pin_cnt.ensure(1);
    HIP_CHECK(hipMemcpyAsync(c->samp_pin_cnt.p, count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (!c->samp_ev) HIP_CHECK(hipEventCreateWithFlags(&c->samp_ev, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(c->samp_ev, st));
}
void k_draw_samples_begin(cge_ctx *c, i64 seed, i64 stream_id, i64 S, int directed, i32 *d_pos, i32 *d_ni, i32 *d_nj) {
    if (c->samp_pending.on) CGE_THROW(CGE_E_ASSERT, "draw_samples: a draw is already pending");
    hipStream_t st = c->stream;
    const unsigned nb = (unsigned)((S + 255) / 256);
    // positive draws are rows of the caller's WHOLE list (a sharded list: the owner of a row answers for it, k_prep_samples)
    hipLaunchKernelGGL(draw_pos_kernel, dim3(nb), dim3(256), 0, st, (uint64_t)seed, (uint64_t)stream_id, S,
                       c->edges_sharded ? c->m_total : c->m, d_pos);
    c->samp_attempt.ensure(S); c->samp_todo_a.ensure(S); c->samp_todo_b.ensure(S); c->samp_count.ensure(1);
    HIP_CHECK(hipMemsetAsync(c->samp_attempt.p, 0, sizeof(unsigned) * S, st));
    cge_ctx::DrawPending &P = c->samp_pending;
    P = cge_ctx::DrawPending();
    P.on = true;
    P.seed = seed; P.stream_id = stream_id; P.S = S; P.directed = directed;
    P.d_pos = d_pos; P.d_ni = d_ni; P.d_nj = d_nj;
    P.cnt = S;
    P.todo = nullptr; // round 0: every sample
    P.next = c->samp_todo_a.p;
    P.round = 0;
    draw_round(c, P);
}
void k_draw_samples_finish(cge_ctx *c) {
    cge_ctx::DrawPending &P = c->samp_pending;
    if (!P.on) CGE_THROW(CGE_E_ASSERT, "draw_samples: no draw is pending");
    for (;;) {
        HIP_CHECK(hipEventSynchronize(c->samp_ev));
        P.cnt = (i64)c->samp_pin_cnt.p[0];
        P.todo = P.next;
        P.next = (P.next == c->samp_todo_a.p) ? c->samp_todo_b.p : c->samp_todo_a.p;
        P.round++;
        if (P.cnt <= 0 || P.round >= 64) break;
        draw_round(c, P);
    }
    P.on = false;
    if (P.cnt > 0) CGE_THROW(CGE_E_ARG, "draw_samples: could not find enough non-edges (graph too dense?)");
}
void k_draw_samples_dev(cge_ctx *c, i64 seed, i64 stream_id, i64 S, int directed, i32 *d_pos, i32 *d_ni, i32 *d_nj) {
    k_draw_samples_begin(c, seed, stream_id, S, directed, d_pos, d_ni, d_nj);
    k_draw_samples_finish(c);
}
// sampled pairs -> what the AUC kernels read: the edge of every positive draw (canonical order when undirected) with the
// weight of the FIRST draw (the directed exact-mode quirk of :510 overwrites the pairs, not the weights), the non-edges
__global__ void prep_samples_kernel(const i32 *__restrict__ pos, const i32 *__restrict__ pos_pairs, const i32 *__restrict__ ni_in,
                                    const i32 *__restrict__ nj_in, const i32 *__restrict__ e_src, const i32 *__restrict__ e_dst,
                                    const double *__restrict__ e_w, i64 S, int directed, i32 *__restrict__ pi,
                                    i32 *__restrict__ pj, i32 *__restrict__ ni, i32 *__restrict__ nj, double *__restrict__ wts) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S) return;
    const i32 rp = pos_pairs[k];
    i32 a = e_src[rp], b = e_dst[rp];
    if (!directed && a > b) { const i32 t = a; a = b; b = t; } // E tuple (min,max) :133
    pi[k] = a;
    pj[k] = b;
    wts[k] = e_w ? e_w[pos[k]] : 1.0;
    i32 u = ni_in[k], v = nj_in[k];
    if (!directed && u > v) { const i32 t = u; u = v; v = t; }
    ni[k] = u;
    nj[k] = v;
}
// the same over a SHARDED edge list: rows held by this rank are looked up (ids + 1, weight), the others contribute zeros;
// after the all-reduce(sum) of the 3 S doubles (exact: one non-zero term each) the second kernel orders and stores them
__global__ void prep_lookup_sharded_kernel(const i32 *__restrict__ pos, const i32 *__restrict__ pos_pairs, const i32 *__restrict__ e_src,
                                           const i32 *__restrict__ e_dst, const double *__restrict__ e_w, i64 e_first, i64 m_local, i64 S,
                                           double *__restrict__ x) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S) return;
    const i64 rp = (i64)pos_pairs[k] - e_first, rw = (i64)pos[k] - e_first;
    const bool mine = rp >= 0 && rp < m_local, minew = rw >= 0 && rw < m_local;
    x[k] = mine ? (double)(e_src[rp] + 1) : 0.0;
    x[S + k] = mine ? (double)(e_dst[rp] + 1) : 0.0;
    x[2 * S + k] = minew ? (e_w ? e_w[rw] : 1.0) : 0.0;
}
__global__ void prep_store_sharded_kernel(const double *__restrict__ x, const i32 *__restrict__ ni_in, const i32 *__restrict__ nj_in,
                                          i64 S, int directed, i32 *__restrict__ pi, i32 *__restrict__ pj, i32 *__restrict__ ni,
                                          i32 *__restrict__ nj, double *__restrict__ wts) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S) return;
    i32 a = (i32)x[k] - 1, b = (i32)x[S + k] - 1;
    if (!directed && a > b) { const i32 t = a; a = b; b = t; }
    pi[k] = a;
    pj[k] = b;
    wts[k] = x[2 * S + k];
    i32 u = ni_in[k], v = nj_in[k];
    if (!directed && u > v) { const i32 t = u; u = v; v = t; }
    ni[k] = u;
    nj[k] = v;
}
void k_prep_samples(cge_ctx *c, const i32 *pos, const i32 *pos_pairs, const i32 *ni_in, const i32 *nj_in, const i32 *e_src,
                    const i32 *e_dst, const double *e_w, i64 S, int directed, i32 *pi, i32 *pj, i32 *ni, i32 *nj, double *wts) {
    if (c->edges_sharded && e_src == c->src.p) {
        const unsigned nb = (unsigned)((S + 255) / 256);
        c->samp_xchg.ensure((size_t)3 * S);
        hipLaunchKernelGGL(prep_lookup_sharded_kernel, dim3(nb), dim3(256), 0, c->stream, pos, pos_pairs, e_src, e_dst,
                           c->unit_weights ? nullptr : e_w, c->e_first, c->m, S, c->samp_xchg.p);
        cge_allreduce_dev(c, c->samp_xchg.p, 3 * S, 0);
        hipLaunchKernelGGL(prep_store_sharded_kernel, dim3(nb), dim3(256), 0, c->stream, c->samp_xchg.p, ni_in, nj_in, S, directed, pi,
                           pj, ni, nj, wts);
        return;
    }
    hipLaunchKernelGGL(prep_samples_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, c->stream, pos, pos_pairs, ni_in, nj_in,
                       e_src, e_dst, e_w, S, directed, pi, pj, ni, nj, wts);
}

// ------------------------------------------------------------------------------------------------
__global__ void gather_i32_kernel(const i32 *__restrict__ arr, const i32 *__restrict__ idx, i64 S,
                                  i32 *__restrict__ out) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < S; k += stride) out[k] = arr[idx[k]];
}
void k_gather_i32(cge_ctx *c, const i32 *arr, const i32 *idx, i64 S, i32 *out) {
    if (S <= 0) return;
    hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for(S, 256)), dim3(256), 0, c->stream, arr, idx, S, out);
}

// ------------------------------------------------------------------------------------------------
// N > 1 with the local-score tallies split over the ranks: the verdict of an enqueued persistent fit {converged,
// iterations, failed} as ONE double next to the tallies, so that the all-reduce(sum) of the tallies carries it and every
// rank redoes an alpha when any rank's fit was abandoned (wgcl_host.cpp).  `async` = 0: this alpha's fit was waited for
// (or ran launch by launch): nothing to report.
__global__ void fit_verdict_kernel(const int *__restrict__ flags, int async, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (async && (flags[2] != 0 || flags[0] == 0)) ? 1.0 : 0.0;
}
void k_fit_verdict(cge_ctx *c, const int *flags, int async, double *out) {
    hipLaunchKernelGGL(fit_verdict_kernel, dim3(1), dim3(64), 0, c->stream, flags, async, out);
}
