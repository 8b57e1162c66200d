// kernels_dist.hip -- distance kernels.
//   k_dist_matrix : D[i,j] = dist(i,j,embed), diagonal = distances[i]      src/divergence.jl:79-91
//   k_minmax_upper / k_normalise : lo,hi = extrema(D); D = (D-lo)/(hi-lo)  src/divergence.jl:92-93
//   k_max_pair    : arg-max of all n(n-1)/2 pairwise distances (the `hi` of
//                   extrema(full_graph_D), never materialised)             src/divergence.jl:104-113
//   k_pair_dist   : distances of sampled pairs                              src/divergence.jl:189,198
#include "common.hpp"
#include <type_traits>
#include "mfma_tile.hpp"

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// Landmark-level D (N x N, full symmetric storage).  64x64 output tile per workgroup, each thread
// a 4x4 sub-tile; operands staged through LDS in k-chunks; the k-sum runs in ascending k with
// unfused sub/mul/add, i.e. the arithmetic of dist() (src/auxilary.jl:14-20) term for term.
#define DT 64
#define DK 16
__global__ __launch_bounds__(256) void dist_matrix_kernel(const double *__restrict__ emb,
                                                          const double *__restrict__ diag, i64 N, i64 d,
                                                          double *__restrict__ D) {
    __shared__ double As[DT][DK + 1], Bs[DT][DK + 1];
    const i64 bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return; // upper triangle of tiles; the mirror is written by the same block
    const i64 i0 = bi * DT, j0 = bj * DT;
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16; // thread tile rows ty*4.., cols tx*4..
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
    for (i64 k0 = 0; k0 < d; k0 += DK) {
        __syncthreads();
        for (int e = threadIdx.x; e < DT * DK; e += 256) {
            const int r = e / DK, kk = e % DK;
            const i64 k = k0 + kk;
            As[r][kk] = (i0 + r < N && k < d) ? emb[(i0 + r) * d + k] : 0.0;
            Bs[r][kk] = (j0 + r < N && k < d) ? emb[(j0 + r) * d + k] : 0.0;
        }
        __syncthreads();
        const int kmax = (int)min((i64)DK, d - k0);
        for (int kk = 0; kk < kmax; kk++) {
            double a4[4], b4[4];
#pragma unroll
            for (int a = 0; a < 4; a++) a4[a] = As[ty * 4 + a][kk];
#pragma unroll
            for (int b = 0; b < 4; b++) b4[b] = Bs[tx * 4 + b][kk];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const double df = __dsub_rn(a4[a], b4[b]);
                    acc[a][b] = __dadd_rn(acc[a][b], __dmul_rn(df, df));
                }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const i64 i = i0 + ty * 4 + a, j = j0 + tx * 4 + b;
            if (i < N && j < N) {
                const double v = (i == j) ? diag[i] : sqrt(acc[a][b]);
                D[i * N + j] = v;
                if (bi != bj) D[j * N + i] = v;
            }
        }
}
void k_dist_matrix(cge_ctx *c, const double *emb, const double *diag, i64 N, i64 d, double *D) {
    ScopedKernelTimer t(c, "dist_matrix");
    const unsigned nb = (unsigned)((N + DT - 1) / DT);
    hipLaunchKernelGGL(dist_matrix_kernel, dim3(nb, nb), dim3(256), 0, c->stream, emb, diag, N, d, D);
}

// extrema over the upper triangle (incl. diagonal); two-stage, order independent (min/max are exact)
__global__ void minmax_partial_kernel(const double *__restrict__ D, i64 N, double *__restrict__ part) {
    __shared__ double slo[256], shi[256];
    double lo = INFINITY, hi = -INFINITY;
    const i64 total = N * N, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 i = e / N, j = e - i * N;
        if (j >= i) {
            const double v = D[e];
            lo = fmin(lo, v);
            hi = fmax(hi, v);
        }
    }
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + s]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = slo[0];
        part[2 * blockIdx.x + 1] = shi[0];
    }
}
__global__ void minmax_final_kernel(const double *__restrict__ part, int nb, double *__restrict__ lo_hi) { // one block of 256 (min / max: any order)
    __shared__ double slo[256], shi[256];
    double lo = INFINITY, hi = -INFINITY;
    for (int b = threadIdx.x; b < nb; b += 256) {
        lo = fmin(lo, part[2 * b]);
        hi = fmax(hi, part[2 * b + 1]);
    }
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + s]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        lo_hi[0] = slo[0];
        lo_hi[1] = shi[0];
    }
}
void k_minmax_upper(cge_ctx *c, const double *D, i64 N, double *lo_hi) {
    const int nb = (int)grid_for(N * N, 256, 1024);
    DevBuf<double> &part = c->sw_mm;
    part.ensure((size_t)2 * nb);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, c->stream, D, N, part.p);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, c->stream, part.p, nb, lo_hi);
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
__global__ void normalise_kernel(double *__restrict__ D, i64 total, const double *__restrict__ lo_hi) {
    const double lo = lo_hi[0], den = lo_hi[1] - lo_hi[0];
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) D[e] = (D[e] - lo) / den;
}
void k_normalise(cge_ctx *c, double *D, i64 N, const double *lo_hi) {
    hipLaunchKernelGGL(normalise_kernel, dim3(grid_for(N * N, 256)), dim3(256), 0, c->stream, D, N * N, lo_hi);
}

// ------------------------------------------------------------------------------------------------
// Point-set diameter.  dist^2(i,j) = r_i + r_j - 2 <x_i, x_j> on mean-centred rows; the Gram
// tiles run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  Only tiles with J >= I are
// visited; nothing n x n is ever stored.  Each workgroup walks a strided list of 128x128 tiles in
// super-block order (32x32 tiles share two 4 MiB row panels => L2/Infinity-Cache resident) and
// keeps its running (max, i, j); one record per workgroup is written at the end.
//
// Operand layout: Xc is FEATURE-major, dpad x ldn doubles (ldn = n rounded up to 128, dpad = d
// rounded up to MP_BK, zero padded), so a k-row of a tile is 1 KiB contiguous and 16-B aligned.
//
// MFMA operand maps (cdna_hip_programming.md §3, f64 note): A lane l -> A[row l&15][k l>>4],
// B lane l -> B[k l>>4][col l&15]; C/D lane l, reg r -> row (l>>4) + 4r, col l&15.
struct MaxRec {
    double val;
    i64 i, j;
};

__device__ __forceinline__ void tile_from_linear(i64 t, i64 nS, i64 &SI, i64 &I, i64 &J) {
    // Linear order: super-blocks (SI, SJ >= SI) row-major; inside a super-block tiles row-major.
    const i64 per = (i64)MP_SB * MP_SB;
    const i64 sb = t / per, loc = t - sb * per;
    // row SI of the upper triangle: off(SI) = SI*nS - SI*(SI-1)/2 <= sb < off(SI+1)
    const double b = 2.0 * (double)nS + 1.0;
    i64 si = (i64)((b - sqrt(b * b - 8.0 * (double)sb)) * 0.5);
    if (si < 0) si = 0;
    if (si > nS - 1) si = nS - 1;
    while (si > 0 && si * nS - si * (si - 1) / 2 > sb) si--;
    while (si + 1 < nS && (si + 1) * nS - (si + 1) * si / 2 <= sb) si++;
    const i64 SJ = si + (sb - (si * nS - si * (si - 1) / 2));
    SI = si;
    I = si * MP_SB + loc / MP_SB;
    J = SJ * MP_SB + loc % MP_SB;
}

// workgroup reduction of (best, i, j) -> recs[blockIdx.x]: value max, ties -> smallest (i,j)
__device__ __forceinline__ void reduce_best(double best, i64 best_i, i64 best_j, double *lds, MaxRec *recs) {
    const int tid = threadIdx.x;
    __syncthreads();
    double *sv = lds;
    i64 *si = reinterpret_cast<i64 *>(lds + 256), *sj = reinterpret_cast<i64 *>(lds + 512);
    sv[tid] = best; si[tid] = best_i; sj[tid] = best_j;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double ov = sv[tid + s];
            const i64 oi = si[tid + s], oj = sj[tid + s];
            if (ov > sv[tid] || (ov == sv[tid] && (oi < si[tid] || (oi == si[tid] && oj < sj[tid])))) {
                sv[tid] = ov; si[tid] = oi; sj[tid] = oj;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        recs[blockIdx.x].val = sv[0];
        recs[blockIdx.x].i = si[0];
        recs[blockIdx.x].j = sj[0];
    }
}

// (1) brute force over all tile pairs J >= I of ONE operand
__global__ __launch_bounds__(256, 2) void max_pair_kernel(const double *__restrict__ Xc,
                                                          const double *__restrict__ rnorm, i64 n, i64 ldn, i64 dpad,
                                                          int part, int nparts, MaxRec *__restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4, c2 = lane * 2;
    const i64 nT = ldn / MP_BM;
    const i64 nS = (nT + MP_SB - 1) / MP_SB;
    const i64 total = nS * (nS + 1) / 2 * MP_SB * MP_SB;
    const i64 nchunk = dpad / MP_BK;
    double best = -1.0;
    i64 best_i = 0, best_j = 0;
    for (i64 t = blockIdx.x; t < total; t += gridDim.x) {
        i64 SI, I, J;
        tile_from_linear(t, nS, SI, I, J);
        if (I >= nT || J >= nT || J < I || (SI % nparts) != part) continue; // uniform per workgroup
        const i64 i0 = I * MP_BM, j0 = J * MP_BN;
        d4 acc[4][4];
        gram_tile_128(Xc + (i64)wave * ldn + i0 + c2, Xc + (i64)wave * ldn + j0 + c2, ldn, ldn, nchunk, lds, acc, wave,
                      c2, wr, wc, lr, lk);
        // acc[a][b][r]: row = i0 + wr*64 + a*16 + lk + 4r, col = j0 + wc*64 + b*16 + lr
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const i64 j = j0 + wc * 64 + b * 16 + lr;
            const double rj = rnorm[j]; // rnorm is padded to ldn
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const i64 i = i0 + wr * 64 + a * 16 + lk + 4 * r;
                    if (j < n && i < j) {
                        const double v = rnorm[i] + rj - 2.0 * acc[a][b][r];
                        if (v > best) { best = v; best_i = i; best_j = j; }
                    }
                }
        }
    }
    reduce_best(best, best_i, best_j, lds, recs);
}

// (2) point-to-centroid maxima: P[a][b] = max over the rows of landmark a of ||x - mu_b||^2.
// Rows are the landmark-sorted copy Xs (every landmark padded to a multiple of 16 rows, so a 16-row
// MFMA sub-tile belongs to one landmark: sub_land[pos/16], -1 in the tail padding).  Results are
// combined with integer atomic max on the bit pattern (values are clamped to >= 0).
__global__ __launch_bounds__(256, 2) void pcent_kernel(const double *__restrict__ Xs, const double *__restrict__ rns,
                                                       i64 lds_rows, const double *__restrict__ Ms,
                                                       const double *__restrict__ mnorm, i64 ldm, i64 N, i64 dpad,
                                                       const i32 *__restrict__ sub_land,
                                                       unsigned long long *__restrict__ P, i64 I0, i64 I1) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4, c2 = lane * 2;
    const i64 nTJ = ldm / MP_BN;
    const i64 nchunk = dpad / MP_BK;
    for (i64 t = blockIdx.x; t < (I1 - I0) * nTJ; t += gridDim.x) { // this rank's row tiles [I0, I1)
        const i64 I = I0 + t / nTJ, J = t % nTJ; // consecutive workgroups share the row tile
        const i64 i0 = I * MP_BM, j0 = J * MP_BN;
        d4 acc[4][4];
        gram_tile_128(Xs + (i64)wave * lds_rows + i0 + c2, Ms + (i64)wave * ldm + j0 + c2, lds_rows, ldm, nchunk, lds,
                      acc, wave, c2, wr, wc, lr, lk);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const i64 col = j0 + wc * 64 + b * 16 + lr;
            const double mn = mnorm[col];
            double cur = 0.0;
            int curland = -1;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const i64 r0 = i0 + wr * 64 + a * 16;
                const int land = sub_land[r0 >> 4]; // wave-uniform
                double v = -1e300;
#pragma unroll
                for (int r = 0; r < 4; r++) v = fmax(v, rns[r0 + lk + 4 * r] - 2.0 * acc[a][b][r]);
                v = fmax(v, __shfl_xor(v, 16));
                v = fmax(v, __shfl_xor(v, 32));
                if (land != curland) {
                    if (curland >= 0 && lk == 0 && col < N)
                        atomicMax(&P[(i64)curland * N + col], (unsigned long long)__double_as_longlong(fmax(cur + mn, 0.0)));
                    curland = land;
                    cur = v;
                } else
                    cur = fmax(cur, v);
            }
            if (curland >= 0 && lk == 0 && col < N)
                atomicMax(&P[(i64)curland * N + col], (unsigned long long)__double_as_longlong(fmax(cur + mn, 0.0)));
        }
    }
}

// (2b) the same maxima as UPPER BOUNDS by fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 flop/clk/SIMD, about three times the
// sustained fp64 matrix rate).  The operands are the fp32 roundings of the centred rows / reference points; squared norms
// stay fp64.  For a row x and a reference m the fp32 chain returns s with |s - <x, m>| <= (K + 2) u sum|x_k m_k|
// <= (K + 2) u (|x|^2 + |m|^2) / 2, u = 2^-24 (one rounding per operand, one per product-accumulate of the k-ordered chain),
// so  |x|^2 (1 + e) - 2 s + |m|^2 (1 + e),  e = 1.01 (K + 2) u,  is never below |x - m|^2: the pruning stays exact, it only
// keeps marginally more candidates (e ~ 8e-6 at K = 128, 3e-5 at K = 512).  The candidates are evaluated in fp64 as before.
typedef float f16v __attribute__((ext_vector_type(16)));
#define PF_BK 16
#define PF_LDS_BYTES ((size_t)2 * 2 * PF_BK * 128 * sizeof(float))
__global__ __launch_bounds__(256, 2) void pcent_f32_kernel(const float *__restrict__ Xs, const double *__restrict__ rns,
                                                           i64 lds_rows, const float *__restrict__ Ms,
                                                           const double *__restrict__ mnorm, i64 ldm, i64 N, i64 dpad,
                                                           double *__restrict__ G, i64 I0, i64 I1, double e1) {
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, l32 = lane & 31, lh = lane >> 5;
    const int krow = tid >> 4, seg = (tid & 15) * 8; // staging: 16 k-rows x 128 floats per operand and chunk
    const i64 nTJ = ldm / 128, nchunk = dpad / PF_BK;
    const size_t stage = (size_t)2 * PF_BK * 128;
    const i64 ntile = (I1 - I0) * nTJ;
    float4 ra0, ra1, rb0, rb1; // named scalars: as arrays they end up in scratch memory
    if ((i64)blockIdx.x < ntile) { // first chunk of the first tile
        const i64 I = I0 + (i64)blockIdx.x / nTJ, J = (i64)blockIdx.x % nTJ;
        const float *pa = Xs + (i64)krow * lds_rows + I * 128 + seg, *pb = Ms + (i64)krow * ldm + J * 128 + seg;
        ra0 = *reinterpret_cast<const float4 *>(pa);
        ra1 = *reinterpret_cast<const float4 *>(pa + 4);
        rb0 = *reinterpret_cast<const float4 *>(pb);
        rb1 = *reinterpret_cast<const float4 *>(pb + 4);
    }
    for (i64 t = blockIdx.x; t < ntile; t += gridDim.x) {
        const i64 I = I0 + t / nTJ, J = t % nTJ; // consecutive workgroups share the row tile
        const i64 i0 = I * 128, j0 = J * 128;
        const float *pa = Xs + (i64)krow * lds_rows + i0 + seg, *pb = Ms + (i64)krow * ldm + j0 + seg;
        f16v acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
        __syncthreads(); // the previous tile's readers are done with both stages
        {
            float *As = ldsf + krow * 128 + seg, *Bs = As + PF_BK * 128;
            *reinterpret_cast<float4 *>(As) = ra0;
            *reinterpret_cast<float4 *>(As + 4) = ra1;
            *reinterpret_cast<float4 *>(Bs) = rb0;
            *reinterpret_cast<float4 *>(Bs + 4) = rb1;
        }
        __syncthreads();
        for (i64 kc = 0; kc < nchunk; kc++) {
            const int s = (int)(kc & 1);
            const bool more = kc + 1 < nchunk;
            if (more) { // the next chunk's global loads land while the MFMAs run
                const float *qa = pa + (kc + 1) * PF_BK * lds_rows, *qb = pb + (kc + 1) * PF_BK * ldm;
                ra0 = *reinterpret_cast<const float4 *>(qa);
                ra1 = *reinterpret_cast<const float4 *>(qa + 4);
                rb0 = *reinterpret_cast<const float4 *>(qb);
                rb1 = *reinterpret_cast<const float4 *>(qb + 4);
            }
            const float *As = ldsf + (size_t)s * stage, *Bs = As + PF_BK * 128;
            float af[PF_BK / 2][2], bf[PF_BK / 2][2]; // all fragments of the chunk first: the 32 MFMAs then issue back to back
#pragma unroll
            for (int ks = 0; ks < PF_BK / 2; ks++) {
#pragma unroll
                for (int a = 0; a < 2; a++) af[ks][a] = As[(2 * ks + lh) * 128 + wr * 64 + a * 32 + l32];
#pragma unroll
                for (int b = 0; b < 2; b++) bf[ks][b] = Bs[(2 * ks + lh) * 128 + wc * 64 + b * 32 + l32];
            }
            __builtin_amdgcn_sched_barrier(0); // keep the reads ahead of the MFMA block (the scheduler sinks them otherwise)
#pragma unroll
            for (int ks = 0; ks < PF_BK / 2; ks++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks][a], bf[ks][b], acc[a][b], 0, 0, 0);
            if (more) {
                float *An = ldsf + (size_t)(s ^ 1) * stage + krow * 128 + seg, *Bn = An + PF_BK * 128;
                *reinterpret_cast<float4 *>(An) = ra0;
                *reinterpret_cast<float4 *>(An + 4) = ra1;
                *reinterpret_cast<float4 *>(Bn) = rb0;
                *reinterpret_cast<float4 *>(Bn + 4) = rb1;
            }
            __syncthreads();
        }
        if (t + gridDim.x < ntile) { // the next tile's first chunk is requested before this tile's epilogue
            const i64 tn = t + gridDim.x, In = I0 + tn / nTJ, Jn = tn % nTJ;
            const float *qa = Xs + (i64)krow * lds_rows + In * 128 + seg, *qb = Ms + (i64)krow * ldm + Jn * 128 + seg;
            ra0 = *reinterpret_cast<const float4 *>(qa);
            ra1 = *reinterpret_cast<const float4 *>(qa + 4);
            rb0 = *reinterpret_cast<const float4 *>(qb);
            rb1 = *reinterpret_cast<const float4 *>(qb + 4);
        }
        // C/D layout of 32x32x2: lane l, register 4g + j -> row 8g + 4 (l >> 5) + j, column l & 31.  A landmark owns whole
        // 16-row groups: group h of a 32-row block = registers g in {2h, 2h+1} of both lane halves.  One value per (group,
        // reference point) goes out with a plain 256-byte store per half wave; the groups of a landmark are combined by
        // pcent_groups_kernel (per-element atomic maxima on P would be ~3 x 10^7 memory-side read-modify-writes).
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const i64 col = j0 + wc * 64 + b * 32 + l32;
            const double mn = mnorm[col] * e1;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const i64 r0 = i0 + wr * 64 + a * 32 + 16 * h;
                    double v = -1e300;
#pragma unroll
                    for (int gg = 0; gg < 2; gg++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                        {
                            const float af = acc[a][b][4 * (2 * h + gg) + j]; // a non-finite sum (overflow) must not prune: it counts as +Inf
                            v = fmax(v, (af - af == 0.f) ? rns[r0 + 8 * gg + 4 * lh + j] * e1 - 2.0 * (double)af : 1e300);
                        }
                    v = fmax(v, __shfl_xor(v, 32));
                    if (lh == 0) G[(r0 >> 4) * ldm + col] = fmax(v + mn, 0.0);
                }
        }
    }
}

// The same for K = dpad <= 128 with the 128-row tile of X RESIDENT in LDS (64 KB): a workgroup takes a row tile, loads it
// once and sweeps all column tiles of the reference points against it, so X is read from memory exactly once (the form
// above re-reads a row tile once per column tile: 4x at C = 500); only the small reference operand is streamed (from L2).
__global__ __launch_bounds__(256, 2) void pcent_f32_rowres_kernel(const float *__restrict__ Xs, const double *__restrict__ rns,
                                                                  i64 lds_rows, const float *__restrict__ Ms,
                                                                  const double *__restrict__ mnorm, i64 ldm, i64 N, i64 dpad,
                                                                  double *__restrict__ G, i64 I0, i64 I1, double e1) {
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    float *Afull = ldsf;                      // [dpad][128]
    float *Bst = ldsf + (size_t)dpad * 128;   // [2][PF_BK][128]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, l32 = lane & 31, lh = lane >> 5;
    const int krow = tid >> 4, seg = (tid & 15) * 8;
    const i64 nTJ = ldm / 128, nchunk = dpad / PF_BK, nstep = nTJ * nchunk; // (column tile, k chunk) pairs of a row tile
    for (i64 I = I0 + blockIdx.x; I < I1; I += gridDim.x) {
        const i64 i0 = I * 128;
        __syncthreads(); // the previous row tile's readers are done
        for (i64 kr = krow; kr < dpad; kr += 16) { // the whole row tile, all k
            const float *pa = Xs + kr * lds_rows + i0 + seg;
            const float4 v0 = *reinterpret_cast<const float4 *>(pa), v1 = *reinterpret_cast<const float4 *>(pa + 4);
            *reinterpret_cast<float4 *>(Afull + kr * 128 + seg) = v0;
            *reinterpret_cast<float4 *>(Afull + kr * 128 + seg + 4) = v1;
        }
        float4 rb0, rb1;
        {
            const float *pb = Ms + (i64)krow * ldm + seg; // step 0: column tile 0, chunk 0
            rb0 = *reinterpret_cast<const float4 *>(pb);
            rb1 = *reinterpret_cast<const float4 *>(pb + 4);
            *reinterpret_cast<float4 *>(Bst + krow * 128 + seg) = rb0;
            *reinterpret_cast<float4 *>(Bst + krow * 128 + seg + 4) = rb1;
        }
        __syncthreads();
        f16v acc[2][2];
        for (i64 stp = 0; stp < nstep; stp++) {
            const i64 J = stp / nchunk, kc = stp - J * nchunk;
            const int s = (int)(stp & 1);
            if (kc == 0) {
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
            }
            const bool more = stp + 1 < nstep;
            if (more) { // the next (column tile, chunk) of the reference operand lands while the MFMAs run
                const i64 Jn = (stp + 1) / nchunk, kn = (stp + 1) - Jn * nchunk;
                const float *qb = Ms + (kn * PF_BK + krow) * ldm + Jn * 128 + seg;
                rb0 = *reinterpret_cast<const float4 *>(qb);
                rb1 = *reinterpret_cast<const float4 *>(qb + 4);
            }
            const float *As = Afull + (size_t)kc * PF_BK * 128, *Bs = Bst + (size_t)s * PF_BK * 128;
            float af[PF_BK / 2][2], bf[PF_BK / 2][2]; // all fragments of the chunk first: the 32 MFMAs then issue back to back
#pragma unroll
            for (int ks = 0; ks < PF_BK / 2; ks++) {
#pragma unroll
                for (int a = 0; a < 2; a++) af[ks][a] = As[(2 * ks + lh) * 128 + wr * 64 + a * 32 + l32];
#pragma unroll
                for (int b = 0; b < 2; b++) bf[ks][b] = Bs[(2 * ks + lh) * 128 + wc * 64 + b * 32 + l32];
            }
            __builtin_amdgcn_sched_barrier(0); // keep the reads ahead of the MFMA block (the scheduler sinks them otherwise)
#pragma unroll
            for (int ks = 0; ks < PF_BK / 2; ks++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks][a], bf[ks][b], acc[a][b], 0, 0, 0);
            if (more) {
                float *Bn = Bst + (size_t)(s ^ 1) * PF_BK * 128 + krow * 128 + seg;
                *reinterpret_cast<float4 *>(Bn) = rb0;
                *reinterpret_cast<float4 *>(Bn + 4) = rb1;
            }
            if (kc == nchunk - 1) { // epilogue of column tile J (see pcent_f32_kernel)
                const i64 j0 = J * 128;
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const i64 col = j0 + wc * 64 + b * 32 + l32;
                    const double mn = mnorm[col] * e1;
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const i64 r0 = i0 + wr * 64 + a * 32 + 16 * h;
                            double v = -1e300;
#pragma unroll
                            for (int gg = 0; gg < 2; gg++)
#pragma unroll
                                for (int j = 0; j < 4; j++)
                                {
                            const float af = acc[a][b][4 * (2 * h + gg) + j]; // a non-finite sum (overflow) must not prune: it counts as +Inf
                            v = fmax(v, (af - af == 0.f) ? rns[r0 + 8 * gg + 4 * lh + j] * e1 - 2.0 * (double)af : 1e300);
                        }
                            v = fmax(v, __shfl_xor(v, 32));
                            if (lh == 0) G[(r0 >> 4) * ldm + col] = fmax(v + mn, 0.0);
                        }
                }
            }
            __syncthreads();
        }
    }
}

// (2c) The same bound pass on the bf16 matrix pipe (16x the f32-input rate), with every operand split into TWO bf16 terms:
// x = xh + xl + ex with |ex| <= 2^-16 |x| (xh = bf16(x), xl = bf16(x - xh)), and x.mu taken as xh.muh + xh.mul + xl.muh --
// three MFMAs (`v_mfma_f32_32x32x16_bf16`, K = 16 each) per f32-input K = 16, i.e. 3/16 of the matrix time.  The products of
// two bf16 numbers are exact in fp32; what is dropped (xl.mul, ex.mu, x.emu) is at most 3.2 * 2^-16 |x||mu|, the fp32
// accumulation of the 3K products at most 3 (K + 2) 2^-23 |x||mu| (twice the round-to-nearest bound: nothing is assumed
// about the adder tree inside the instruction), so with
//     e = 1.05 * (3 (K + 2) 2^-23 + 3.2 * 2^-16),   |x.mu - computed| <= e/2 (||x||^2 + ||mu||^2)
// (rns + mnorm)(1 + e) - 2 computed is a rigorous upper bound of ||x - mu||^2, as in (2b); e = 9.8e-5 at K = 128, which
// moves a bound of ~2000 by 0.1.  Operands: row-major planes Xb[plane][row][KP] (KP = K rounded up to 32, zero padded).
// The 128-row tile of X is RESIDENT in LDS (both planes, K <= 128: 68 KB), the reference points stream through two staging
// buffers from L2; X is read from memory once.  The gather kernel (kernels_lm.hip: gather_centre_fm_kernel writes the two
// planes beside the fp64 copy) raises a flag when a centred value is outside
// [2^-100, 2^100] or not finite (the split would over- / underflow): the caller then takes the fp64 pass.
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef short s8v __attribute__((ext_vector_type(8)));
#define PB_APAD 8         // bf16 elements of padding per LDS row (16 bytes: the 16-byte fragment reads of 32 rows spread over the banks)
#define PB_T 512 // 8 waves: 4 (row blocks of 32) x 2 (column blocks of 64); two waves per SIMD cover each other's LDS / epilogue latencies
// Both operands of a (row tile, column tile) pair are RESIDENT in LDS with their whole K (2 planes x 128 x K bf16 each:
// 139 KB at K = 128).  The next column tile of the reference points (or the next row tile of X and its first column tile)
// is requested into registers before the MFMA phase of the current pair and stored behind it: one pair is ~100 MFMAs per
// wave (1.3 us), about the latency of the loads it hides.
template <int NKS> // k-steps of 16: KP = 16 NKS
__global__ __launch_bounds__(PB_T) void pcent_bf16_kernel(const unsigned short *__restrict__ Xb, const double *__restrict__ rns,
                                                          i64 lds_rows, const unsigned short *__restrict__ Mb,
                                                          const double *__restrict__ mnorm, i64 ldm,
                                                          const i32 *__restrict__ sub_land, i64 nref,
                                                          unsigned long long *__restrict__ P, i64 I0, i64 I1, double e1,
                                                          int diag /* timing diagnostics only: 1 no epilogue, 2 no MFMA loop, 3 neither */) {
    extern __shared__ __attribute__((aligned(16))) unsigned short ldsb[];
    constexpr i64 KP = 16 * NKS;
    constexpr int LDA = (int)KP + PB_APAD;
    unsigned short *Ah = ldsb, *Al = Ah + (size_t)128 * LDA, *Bh = Al + (size_t)128 * LDA, *Bl = Bh + (size_t)128 * LDA;
    double *gmax = reinterpret_cast<double *>(Bl + (size_t)128 * LDA); // [8 groups of 16 rows][128 columns] of the pair in hand
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, l32 = lane & 31, lh = lane >> 5;
    const i64 nTJ = ldm / 128;
    // a 128-row tile of either operand: half a wave per row, lane = 16-byte piece of the row's two planes (2 KP / 8 <= 32
    // pieces), 16 rows per sweep of the 512 threads, 8 sweeps; all loads of a thread are issued before the first LDS store
    constexpr int apieces = (int)(KP / 8);
    const int arow = tid >> 5, apc = tid & 31;
    const bool aact = apc < 2 * apieces;
    const int apl = aact && apc >= apieces ? 1 : 0, akk = 8 * (apc - apl * apieces);
    auto fetch = [&](const unsigned short *base, i64 rows_ld, i64 r0, uint4 (&v)[8]) {
#pragma unroll
        for (int it = 0; it < 8; it++)
            v[it] = aact ? *reinterpret_cast<const uint4 *>(base + ((size_t)apl * rows_ld + r0 + 16 * it + arow) * KP + akk)
                         : make_uint4(0, 0, 0, 0);
    };
    auto stash = [&](unsigned short *hi, unsigned short *lo, const uint4 (&v)[8]) {
        if (aact) {
#pragma unroll
            for (int it = 0; it < 8; it++) *reinterpret_cast<uint4 *>((apl ? lo : hi) + (size_t)(16 * it + arow) * LDA + akk) = v[it];
        }
    };
    uint4 av[8], bv[8];
    if (I0 + blockIdx.x < I1) {
        fetch(Xb, lds_rows, (I0 + blockIdx.x) * 128, av);
        fetch(Mb, ldm, 0, bv);
    }
    for (i64 I = I0 + blockIdx.x; I < I1; I += gridDim.x) {
        const i64 i0 = I * 128;
        __syncthreads(); // the previous row tile's readers are done
        stash(Ah, Al, av);
        stash(Bh, Bl, bv);
        double rn[2][8]; // the squared norms of this lane's 16 rows of the tile, inflated (all column tiles use them)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 8; q++) rn[h][q] = rns[i0 + wr * 32 + 16 * h + 8 * (q >> 2) + 4 * lh + (q & 3)] * e1;
        __syncthreads();
        for (i64 J = 0; J < nTJ; J++) {
            const bool more_cols = J + 1 < nTJ, more_rows = I + gridDim.x < I1;
            double mn[2];
#pragma unroll
            for (int b = 0; b < 2; b++) mn[b] = mnorm[J * 128 + wc * 64 + b * 32 + l32];
            if (more_cols) fetch(Mb, ldm, (J + 1) * 128, bv);
            else if (more_rows) {
                fetch(Xb, lds_rows, (I + gridDim.x) * 128, av);
                fetch(Mb, ldm, 0, bv);
            }
            f16v acc[2];
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
            if (!(diag & 2))
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                const size_t oa = (size_t)(wr * 32 + l32) * LDA + 16 * ks + 8 * lh;
                const s8v ah = *reinterpret_cast<const s8v *>(Ah + oa), al = *reinterpret_cast<const s8v *>(Al + oa);
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const size_t ob = (size_t)(wc * 64 + b * 32 + l32) * LDA + 16 * ks + 8 * lh;
                    const s8v bh = *reinterpret_cast<const s8v *>(Bh + ob), bl = *reinterpret_cast<const s8v *>(Bl + ob);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, ah), __builtin_bit_cast(bf8v, bh), acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, ah), __builtin_bit_cast(bf8v, bl), acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, al), __builtin_bit_cast(bf8v, bh), acc[b], 0, 0, 0);
                }
            }
            if (!(diag & 1)) { // epilogue of column tile J (see pcent_f32_kernel): a non-finite sum counts as +Inf
                const i64 j0 = J * 128;
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const i64 col = j0 + wc * 64 + b * 32 + l32;
                    const double mne = mn[b] * e1;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const i64 r0 = i0 + wr * 32 + 16 * h;
                        double v = -1e300;
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const float af = acc[b][8 * h + q]; // row 16 h + 8 (q >> 2) + 4 lh + (q & 3)
                            const double t = rn[h][q] - 2.0 * (double)af;
                            v = fmax(v, (af - af == 0.f) ? t : 1e300); // NaN / Inf in the accumulator: no pruning on this entry
                        }
                        v = fmax(v, __shfl_xor(v, 32));
                        if (lh == 0) gmax[(wr * 2 + h) * 128 + wc * 64 + b * 32 + l32] = fmax(v + mne, 0.0);
                        (void)r0; (void)col;
                    }
                }
            }
            __syncthreads(); // every wave is done with column tile J (and its group maxima are in LDS)
            if (!(diag & 1) && tid < 128) {
                // The rows are landmark-sorted: the 8 groups of the tile belong to one, two, rarely more landmarks.  The groups of
                // a landmark are combined here and the result joins P[landmark][column] by one atomic max per run (bit patterns of
                // non-negative doubles order like integers; the atomics of a wave go to consecutive addresses).
                const i64 col = J * 128 + tid;
                double cur = 0.0;
                int curland = -1;
#pragma unroll
                for (int gq = 0; gq < 8; gq++) {
                    const int land = sub_land[(i0 >> 4) + gq]; // uniform
                    const double v = gmax[gq * 128 + tid];
                    if (land != curland) {
                        if (curland >= 0 && col < nref) atomicMax(&P[(i64)curland * nref + col], (unsigned long long)__double_as_longlong(cur));
                        curland = land;
                        cur = v;
                    } else
                        cur = fmax(cur, v);
                }
                if (curland >= 0 && col < nref) atomicMax(&P[(i64)curland * nref + col], (unsigned long long)__double_as_longlong(cur));
            }
            if (more_cols) stash(Bh, Bl, bv);
            __syncthreads(); // gmax may be overwritten, the next column tile read
        }
    }
}
// P[a][r] = max over the 16-row groups of landmark a (groups goff[a] .. goff[a+1], from the landmark-sorted layout)
__global__ void pcent_groups_kernel(const double *__restrict__ G, const i32 *__restrict__ soff, i64 ldm, i64 N,
                                    double *__restrict__ P) {
    const i64 a = blockIdx.x;
    const i32 g0 = soff[a] >> 4, g1 = soff[a + 1] >> 4;
    for (i64 col = threadIdx.x; col < N; col += blockDim.x) {
        double v = 0.0;
        for (i32 g = g0; g < g1; g++) v = fmax(v, G[(i64)g * ldm + col]);
        P[a * N + col] = v;
    }
}

// (3) exact evaluation of a list of 128x128 tiles of the landmark-sorted copy (row offsets posA, posB)
__global__ __launch_bounds__(256, 2) void pair_list_kernel(const double *__restrict__ Xs,
                                                           const double *__restrict__ rns, i64 lds_rows, i64 npos,
                                                           i64 dpad, const int2 *__restrict__ tiles, i64 ntiles,
                                                           MaxRec *__restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4, c2 = lane * 2;
    const i64 nchunk = dpad / MP_BK;
    double best = -1.0;
    i64 best_i = 0, best_j = 0;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 i0 = tiles[t].x, j0 = tiles[t].y;
        d4 acc[4][4];
        gram_tile_128(Xs + (i64)wave * lds_rows + i0 + c2, Xs + (i64)wave * lds_rows + j0 + c2, lds_rows, lds_rows,
                      nchunk, lds, acc, wave, c2, wr, wc, lr, lk);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const i64 j = j0 + wc * 64 + b * 16 + lr;
            const double rj = rns[j];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const i64 i = i0 + wr * 64 + a * 16 + lk + 4 * r;
                    if (i < npos && j < npos) { // rows past npos are zero padding (not vertices)
                        const double v = rns[i] + rj - 2.0 * acc[a][b][r];
                        if (v > best) { best = v; best_i = i; best_j = j; }
                    }
                }
        }
    }
    reduce_best(best, best_i, best_j, lds, recs);
}

static void best_of_recs(cge_ctx *c, const MaxRec *d_recs, int nwg, double *bv, i64 *bi, i64 *bj) {
    std::vector<MaxRec> h(nwg);
    HIP_CHECK(hipMemcpyAsync(h.data(), d_recs, sizeof(MaxRec) * nwg, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    *bv = -1.0; *bi = 0; *bj = 0;
    for (int k = 0; k < nwg; k++)
        if (h[k].val > *bv || (h[k].val == *bv && (h[k].i < *bi || (h[k].i == *bi && h[k].j < *bj)))) {
            *bv = h[k].val; *bi = h[k].i; *bj = h[k].j;
        }
}
#define MP_NWG 512
#define MP_LDS_BYTES ((size_t)2 * 2 * MP_BK * MP_LD * sizeof(double))

// Shard `part` of `nparts` owns the super-block rows SI with SI % nparts == part (balanced to
// one super-row; no tile is visited twice across shards).
void k_max_pair(cge_ctx *c, const double *Xc, const double *rnorm, i64 n, i64 ldn, i64 dpad, int part, int nparts,
                double *best_val, i64 *best_i, i64 *best_j) {
    c->mp_recs.ensure(MP_NWG * 3);
    MaxRec *recs = reinterpret_cast<MaxRec *>(c->mp_recs.p);
    {
        ScopedKernelTimer t(c, "max_pair_dist");
        hipLaunchKernelGGL(max_pair_kernel, dim3(MP_NWG), dim3(256), MP_LDS_BYTES, c->stream, Xc, rnorm, n, ldn, dpad,
                           part, nparts, recs);
    }
    best_of_recs(c, recs, MP_NWG, best_val, best_i, best_j);
}

// P is (number of landmarks) x nref, row stride nref: P[a][r] = max over the rows of landmark a of ||x - ref_r||^2
void k_pcent(cge_ctx *c, const double *Xs, const double *rns, i64 lds_rows, const double *Ms, const double *mnorm,
             i64 ldm, i64 n_land, i64 N, i64 dpad, const i32 *sub_land, double *P, int part, int nparts) {
    HIP_CHECK(hipMemsetAsync(P, 0, sizeof(double) * n_land * N, c->stream));
    ScopedKernelTimer t(c, "pcent");
    const i64 nTI = lds_rows / MP_BM, I0 = nTI * part / nparts, I1 = nTI * (part + 1) / nparts;
    const i64 ntiles = (I1 - I0) * (ldm / MP_BN);
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(pcent_kernel, dim3((unsigned)std::min<i64>(ntiles, 2048)), dim3(256), MP_LDS_BYTES, c->stream, Xs,
                       rns, lds_rows, Ms, mnorm, ldm, N, dpad, sub_land, reinterpret_cast<unsigned long long *>(P), I0, I1);
}

// `soff` = device copy of the landmark offsets of the sorted layout (N_land + 1 entries, multiples of 16).  With several
// ranks every rank fills the groups of its own row tiles; the groups of the others stay 0 and the all-reduce(max) of P
// combines them.
void k_pcent_f32(cge_ctx *c, const float *Xs32, const double *rns, i64 lds_rows, const float *Ms32, const double *mnorm,
                 i64 ldm, i64 n_land, i64 N, i64 dpad, const i32 *soff, double *P, int part, int nparts) {
    const i64 ngroups = lds_rows / 16;
    c->pc_groups.ensure((size_t)ngroups * ldm);
    if (nparts > 1) HIP_CHECK(hipMemsetAsync(c->pc_groups.p, 0, sizeof(double) * (size_t)ngroups * ldm, c->stream));
    ScopedKernelTimer t(c, "pcent");
    const i64 nTI = lds_rows / 128, I0 = nTI * part / nparts, I1 = nTI * (part + 1) / nparts;
    const i64 ntiles = (I1 - I0) * (ldm / 128);
    const double e1 = 1.0 + 1.01 * (double)(dpad + 3) * 5.9604644775390625e-08; // 1 + e, e = 1.01 (K + 3) 2^-24 (covers separately rounded products too)
    if (ntiles > 0 && dpad <= 128) { // the row tile of X stays in LDS: X is read once
        cge_allow_lds((const void *)pcent_f32_rowres_kernel, 160 * 1024);
        const size_t lds = ((size_t)dpad * 128 + 2 * PF_BK * 128) * sizeof(float);
        hipLaunchKernelGGL(pcent_f32_rowres_kernel, dim3((unsigned)std::min<i64>(I1 - I0, 512)), dim3(256), lds, c->stream, Xs32, rns,
                           lds_rows, Ms32, mnorm, ldm, N, dpad, c->pc_groups.p, I0, I1, e1);
    } else if (ntiles > 0)
        hipLaunchKernelGGL(pcent_f32_kernel, dim3((unsigned)std::min<i64>(ntiles, 2048)), dim3(256), PF_LDS_BYTES, c->stream,
                           Xs32, rns, lds_rows, Ms32, mnorm, ldm, N, dpad, c->pc_groups.p, I0, I1, e1);
    hipLaunchKernelGGL(pcent_groups_kernel, dim3((unsigned)n_land), dim3(256), 0, c->stream, c->pc_groups.p, soff, ldm, N, P);
}

// the bf16-split form (2c): Xb / Mb = the two-plane row-major operands written by k_gather_centre_fm (KP = dpad rounded up to 32)
bool k_pcent_bf16_applies(i64 dpad) { return (dpad + 31) / 32 * 32 <= 128; } // a tile of either operand, both planes, whole K, fits LDS
void k_pcent_bf16(cge_ctx *c, const unsigned short *Xb, const double *rns, i64 lds_rows, const unsigned short *Mb,
                  const double *mnorm, i64 ldm, i64 n_land, i64 N, i64 KP, const i32 *sub_land, double *P, int part, int nparts) {
    HIP_CHECK(hipMemsetAsync(P, 0, sizeof(double) * n_land * N, c->stream));
    ScopedKernelTimer t(c, "pcent");
    const i64 nTI = lds_rows / 128, I0 = nTI * part / nparts, I1 = nTI * (part + 1) / nparts;
    const double e1 = 1.0 + 1.05 * (3.0 * (double)(KP + 2) * 1.1920928955078125e-07 + 3.2 * 1.52587890625e-05); // 2^-23, 2^-16
    if (I1 > I0) {
        const size_t lds = (size_t)4 * 128 * (KP + PB_APAD) * sizeof(unsigned short) + (size_t)8 * 128 * sizeof(double); // both operands, both planes, whole K + the group maxima
        const int pb_diag = 0; // (the kernel's timing diagnostics: 1 no epilogue, 2 no MFMA loop)
#define PB_GO(NKS)                                                                                                         \
    do {                                                                                                                   \
        auto kern = pcent_bf16_kernel<NKS>;                                                                                 \
        cge_allow_lds((const void *)kern, 160 * 1024); \
        hipLaunchKernelGGL(kern, dim3((unsigned)std::min<i64>(I1 - I0, 256)), dim3(PB_T), lds, c->stream, Xb, rns, lds_rows, Mb, \
                           mnorm, ldm, sub_land, N, reinterpret_cast<unsigned long long *>(P), I0, I1, e1, pb_diag);        \
    } while (0)
        if (KP == 32) PB_GO(2);
        else if (KP == 64) PB_GO(4);
        else if (KP == 96) PB_GO(6);
        else PB_GO(8);
#undef PB_GO
    }
}

void k_pair_list(cge_ctx *c, const double *Xs, const double *rns, i64 lds_rows, i64 npos, i64 dpad, const void *tiles,
                 i64 ntiles, double *best_val, i64 *best_i, i64 *best_j) {
    c->mp_recs.ensure(MP_NWG * 3);
    MaxRec *recs = reinterpret_cast<MaxRec *>(c->mp_recs.p);
    const int nwg = (int)std::max<i64>(1, std::min<i64>(ntiles, MP_NWG));
    {
        ScopedKernelTimer t(c, "pair_list");
        hipLaunchKernelGGL(pair_list_kernel, dim3(nwg), dim3(256), MP_LDS_BYTES, c->stream, Xs, rns, lds_rows, npos,
                           dpad, reinterpret_cast<const int2 *>(tiles), ntiles, recs);
    }
    best_of_recs(c, recs, nwg, best_val, best_i, best_j);
}

// ------------------------------------------------------------------------------------------------
// Landmark-pair upper bounds.  For x_i in landmark a, x_j in landmark b (any reference points mu):
//   ||x_i - x_j||^2 = ||x_i - mu_b||^2 + ||x_j - mu_a||^2 - ||mu_a - mu_b||^2 - 2 <x_i - mu_a, x_j - mu_b>
//                  <= P_ab + P_ba - D2_ab + 2 sqrt(P_aa P_bb)  =: B_ab .
// Pairs with B_ab (slightly inflated for rounding) >= L, a known lower bound of the diameter^2,
// are appended to `list` as (B, a, b); *count may exceed cap (then the caller falls back).
struct BoundRec {
    double B;
    i32 a, b;
};
// Bounds with reference points: landmark a uses reference lref[a] (its community centroid, or itself).
//   B_ab = Q[a][ref(b)] + Q[b][ref(a)] - ||ref(a) - ref(b)||^2 + 2 sqrt(Q[a][ref(a)] Q[b][ref(b)])
// rd2 = nref x nref squared distances of the reference points.
__global__ __launch_bounds__(256) void bound_select_kernel(const double *__restrict__ Q, const i32 *__restrict__ lref,
                                                           const double *__restrict__ rd2, i64 N, i64 nref, double L,
                                                           BoundRec *__restrict__ list, i64 cap,
                                                           unsigned long long *__restrict__ count) {
    const i64 total = N * N, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 a = e / N, b = e - a * N;
        if (b < a) continue;
        const i64 ra = lref[a], rb = lref[b];
        const double qaa = Q[a * nref + ra], qbb = Q[b * nref + rb];
        const double pre = Q[a * nref + rb] + Q[b * nref + ra] + 2.0 * sqrt(qaa * qbb);
        if (pre * (1.0 + 1e-9) + 1e-9 < L) continue;
        const double Bv = pre - rd2[ra * nref + rb] * (1.0 - 1e-9);
        if (Bv * (1.0 + 1e-9) + 1e-9 >= L) {
            const unsigned long long idx = atomicAdd(count, 1ULL);
            if ((i64)idx < cap) list[idx] = BoundRec{Bv, (i32)a, (i32)b};
        }
    }
}
// the same from the centred feature-major copy Ms (dpad x ldm, zero padded): workgroup a, thread b, coalesced along b
__global__ __launch_bounds__(256) void ref_dist2_fm_kernel(const double *__restrict__ Ms, i64 nref, i64 dpad, i64 ldm,
                                                           double *__restrict__ rd2) {
    const i64 a = blockIdx.x;
    for (i64 b = threadIdx.x; b < nref; b += blockDim.x) {
        double s = 0.0;
#pragma unroll 8
        for (i64 k = 0; k < dpad; k++) { // (dpad is a multiple of 16: eight pairs of loads in flight per round)
            const double df = Ms[k * ldm + a] - Ms[k * ldm + b];
            s += df * df;
        }
        rd2[a * nref + b] = s;
    }
}
// squared distances of the nref reference points (row-major nref x d)
__global__ void ref_dist2_kernel(const double *__restrict__ mu, i64 nref, i64 d, double *__restrict__ rd2) {
    const i64 total = nref * nref, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 a = e / nref, b = e - a * nref;
        const double *ma = mu + a * d, *mb = mu + b * d;
        double s = 0.0;
        for (i64 k = 0; k < d; k++) {
            const double df = ma[k] - mb[k];
            s += df * df;
        }
        rd2[e] = s;
    }
}
// ---- the same selection in two levels, when the reference points are the landmarks' communities -----------------------
// Mx[ca][r] = max over the landmarks a of community ca of Q[a][r].  Every landmark-pair bound of the community pair
// (ca, cb) is at most  Mx[ca][cb] + Mx[cb][ca] + 2 sqrt(Mx[ca][ca] Mx[cb][cb]) - rd2[ca][cb]  (the bound is monotone in its
// four Q terms), so a community pair below L is dropped whole: C (C + 1) / 2 tests instead of N (N + 1) / 2, and only the
// surviving community pairs (a fraction of a percent) are expanded to landmark pairs -- with the same arithmetic, hence the
// same (B, a, b) records as the flat kernel.
__global__ __launch_bounds__(256) void comm_max_kernel(const double *__restrict__ Q, const i32 *__restrict__ ref_off,
                                                       const i32 *__restrict__ ref_mem, i64 nref, double *__restrict__ Mx) {
    const i64 ca = blockIdx.x;
    const i32 b = ref_off[ca], e = ref_off[ca + 1];
    for (i64 r = threadIdx.x; r < nref; r += blockDim.x) {
        double v = 0.0;
        for (i32 t = b; t < e; t++) v = fmax(v, Q[(i64)ref_mem[t] * nref + r]);
        Mx[ca * nref + r] = v;
    }
}
__global__ __launch_bounds__(256) void comm_pairs_kernel(const double *__restrict__ Mx, const double *__restrict__ rd2, i64 C,
                                                         double L, int2 *__restrict__ plist, unsigned *__restrict__ pcount) {
    const i64 total = C * C, stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const i64 ca = e / C, cb = e - ca * C;
        if (cb < ca) continue;
        const double pre = Mx[ca * C + cb] + Mx[cb * C + ca] + 2.0 * sqrt(Mx[ca * C + ca] * Mx[cb * C + cb]);
        if (pre * (1.0 + 1e-9) + 1e-9 < L) continue;
        const double Bv = pre - rd2[ca * C + cb] * (1.0 - 1e-9);
        if (Bv * (1.0 + 1e-9) + 1e-9 >= L) plist[atomicAdd(pcount, 1u)] = make_int2((int)ca, (int)cb);
    }
}
// one workgroup per surviving community pair (grid-stride over the list): its landmark pairs, the flat kernel's arithmetic
__global__ __launch_bounds__(256) void bound_expand_kernel(const double *__restrict__ Q, const double *__restrict__ rd2,
                                                           const i32 *__restrict__ ref_off, const i32 *__restrict__ ref_mem,
                                                           i64 nref, double L, const int2 *__restrict__ plist,
                                                           const unsigned *__restrict__ pcount, BoundRec *__restrict__ list,
                                                           i64 cap, unsigned long long *__restrict__ count) {
    const unsigned np = *pcount;
    for (unsigned q = blockIdx.x; q < np; q += gridDim.x) {
        const i64 ca = plist[q].x, cb = plist[q].y;
        const i32 a0 = ref_off[ca], na = ref_off[ca + 1] - a0, b0 = ref_off[cb], nb = ref_off[cb + 1] - b0;
        const double r2 = rd2[ca * nref + cb] * (1.0 - 1e-9);
        for (i64 e = threadIdx.x; e < (i64)na * nb; e += blockDim.x) {
            const i64 ia = e / nb, ib = e - ia * nb;
            i64 a = ref_mem[a0 + ia], b = ref_mem[b0 + ib];
            if (ca == cb && b < a) continue; // an unordered pair once
            i64 ra = ca, rb = cb;
            if (b < a) { const i64 t = a; a = b; b = t; ra = cb; rb = ca; } // records are (a <= b), as the flat kernel writes them
            const double qaa = Q[a * nref + ra], qbb = Q[b * nref + rb];
            const double pre = Q[a * nref + rb] + Q[b * nref + ra] + 2.0 * sqrt(qaa * qbb);
            if (pre * (1.0 + 1e-9) + 1e-9 < L) continue;
            const double Bv = pre - r2;
            if (Bv * (1.0 + 1e-9) + 1e-9 >= L) {
                const unsigned long long idx = atomicAdd(count, 1ULL);
                if ((i64)idx < cap) list[idx] = BoundRec{Bv, (i32)a, (i32)b};
            }
        }
    }
}
// `ref_off` / `ref_mem` (optional, device): the landmarks grouped by reference point (lref[a] = community of a): two-level form
// the squared distances of the reference points from their centred feature-major copy, into c->mp_rd2 -- on c->stream, which the
// caller may have pointed at a side stream: they do not depend on the bound pass and run beside it (k_bound_select(..., Ms_fm =
// nullptr, rd2_ready = true) then takes them as they are)
void k_ref_dist2_fm(cge_ctx *c, const double *Ms_fm, i64 nref, i64 dpad, i64 ldm) {
    c->mp_rd2.ensure((size_t)nref * nref);
    // (differences of centred values: the centre cancels; the 1e-9 margins of the bound cover the rounding)
    hipLaunchKernelGGL(ref_dist2_fm_kernel, dim3((unsigned)nref), dim3(256), 0, c->stream, Ms_fm, nref, dpad, ldm, c->mp_rd2.p);
}
i64 k_bound_select(cge_ctx *c, const double *Q, const i32 *lref, const double *mu_ref, i64 N, i64 nref, i64 d, double L,
                   void *list, i64 cap, const i32 *ref_off, const i32 *ref_mem, const double *Ms_fm, i64 dpad, i64 ldm,
                   bool rd2_ready) {
    c->mp_count.ensure(2);
    c->mp_rd2.ensure((size_t)nref * nref);
    HIP_CHECK(hipMemsetAsync(c->mp_count.p, 0, 2 * sizeof(i64), c->stream));
    ScopedKernelTimer tm(c, "bound_select");
    if (rd2_ready) {
    } else if (Ms_fm)
        hipLaunchKernelGGL(ref_dist2_fm_kernel, dim3((unsigned)nref), dim3(256), 0, c->stream, Ms_fm, nref, dpad, ldm, c->mp_rd2.p);
    else
        hipLaunchKernelGGL(ref_dist2_kernel, dim3(grid_for(nref * nref, 256)), dim3(256), 0, c->stream, mu_ref, nref, d,
                           c->mp_rd2.p);
    if (ref_off && ref_mem) {
        c->mp_commax.ensure((size_t)nref * nref);
        c->mp_plist.ensure((size_t)nref * (nref + 1)); // int2 per community pair
        hipLaunchKernelGGL(comm_max_kernel, dim3((unsigned)nref), dim3(256), 0, c->stream, Q, ref_off, ref_mem, nref, c->mp_commax.p);
        hipLaunchKernelGGL(comm_pairs_kernel, dim3(grid_for(nref * nref, 256)), dim3(256), 0, c->stream, c->mp_commax.p, c->mp_rd2.p,
                           nref, L, reinterpret_cast<int2 *>(c->mp_plist.p), reinterpret_cast<unsigned *>(c->mp_count.p + 1));
        hipLaunchKernelGGL(bound_expand_kernel, dim3(2048), dim3(256), 0, c->stream, Q, c->mp_rd2.p, ref_off, ref_mem, nref, L,
                           reinterpret_cast<const int2 *>(c->mp_plist.p), reinterpret_cast<const unsigned *>(c->mp_count.p + 1),
                           reinterpret_cast<BoundRec *>(list), cap, reinterpret_cast<unsigned long long *>(c->mp_count.p));
    } else
        hipLaunchKernelGGL(bound_select_kernel, dim3(grid_for(N * N, 256)), dim3(256), 0, c->stream, Q, lref, c->mp_rd2.p, N,
                           nref, L, reinterpret_cast<BoundRec *>(list), cap,
                           reinterpret_cast<unsigned long long *>(c->mp_count.p));
    i64 cnt = 0;
    HIP_CHECK(hipMemcpyAsync(&cnt, c->mp_count.p, sizeof(i64), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return cnt;
}

// farthest vertex from row `src` of Xr (exact dist() arithmetic is not needed: this only seeds a lower bound).
// A quarter wave (one DPP row of 16 lanes) per vertex, two vertices per quarter in flight: eight rows per wave and
// iteration, the 16 partial sums of a row combined inside the DPP row (no LDS crossbar).
__device__ __forceinline__ double row16_sum(double v) {
    const auto mv = [](double x, auto ctrl) {
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, decltype(ctrl)::value, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), decltype(ctrl)::value, 0xf, 0xf, false);
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    v += mv(v, std::integral_constant<int, 0xB1>());  // quad_perm [1,0,3,2]
    v += mv(v, std::integral_constant<int, 0x4E>());  // quad_perm [2,3,0,1]
    v += mv(v, std::integral_constant<int, 0x141>()); // row_half_mirror
    v += mv(v, std::integral_constant<int, 0x140>()); // row_mirror
    return v;
}
__global__ __launch_bounds__(256) void farthest_kernel(const double *__restrict__ Xr, i64 n, i64 d, i64 src,
                                                       const double *__restrict__ srow, MaxRec *__restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, sub = lane & 15, quarter = lane >> 4;
    const i64 wave_global = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    const double *s = srow ? srow : Xr + src * d; // (srow: the seed row itself -- it may live on another rank, option shard_rows)
    double best = -1.0;
    i64 bi = 0;
    for (i64 base = wave_global * 8; base < n; base += nwaves * 8) {
        const i64 i0 = base + quarter, i1 = base + 4 + quarter;
        const bool in0 = i0 < n, in1 = i1 < n;
        const double *x0 = Xr + (in0 ? i0 : 0) * d, *x1 = Xr + (in1 ? i1 : 0) * d;
        double a0 = 0.0, a1 = 0.0;
        if ((d & 1) == 0) { // 16-byte loads
            const d2 *p0 = reinterpret_cast<const d2 *>(x0), *p1 = reinterpret_cast<const d2 *>(x1),
                     *s2 = reinterpret_cast<const d2 *>(s);
            for (i64 k = sub; k < (d >> 1); k += 16) {
                const d2 sv = s2[k], v0 = p0[k], v1 = p1[k];
                const double t0 = v0.x - sv.x, t1 = v0.y - sv.y, u0 = v1.x - sv.x, u1 = v1.y - sv.y;
                a0 += t0 * t0 + t1 * t1;
                a1 += u0 * u0 + u1 * u1;
            }
        } else
            for (i64 k = sub; k < d; k += 16) {
                const double t = x0[k] - s[k], u = x1[k] - s[k];
                a0 += t * t;
                a1 += u * u;
            }
        a0 = row16_sum(a0);
        a1 = row16_sum(a1);
        if (in0 && a0 > best) { best = a0; bi = i0; }
        if (in1 && a1 > best) { best = a1; bi = i1; }
    }
    reduce_best(best, bi, src, lds, recs);
}
// position of the largest of n values (ties: the smallest position), mapped through `map` (optional): the vertex farthest from
// the centre of the point set -- the seed of the farthest-point sweep
__global__ __launch_bounds__(256) void argmax_kernel(const double *__restrict__ v, i64 n, const i32 *__restrict__ map,
                                                     MaxRec *__restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double best = -1.0;
    i64 bi = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        if (v[i] > best) { best = v[i]; bi = i; }
    reduce_best(best, bi, map ? (i64)map[bi] : bi, lds, recs);
}
i64 k_argmax_mapped(cge_ctx *c, const double *v, i64 n, const i32 *map, double *val) {
    if (n <= 0) { if (val) *val = -1.0; return -1; }
    c->mp_recs.ensure(MP_NWG * 3);
    MaxRec *recs = reinterpret_cast<MaxRec *>(c->mp_recs.p);
    const int nwg = (int)std::max<i64>(1, std::min<i64>((n + 255) / 256, 256));
    hipLaunchKernelGGL(argmax_kernel, dim3(nwg), dim3(256), 768 * sizeof(double), c->stream, v, n, map, recs);
    double bv;
    i64 bi, bj;
    best_of_recs(c, recs, nwg, &bv, &bi, &bj);
    if (val) *val = bv;
    return bj;
}
// the sweep in two halves: the launch (on c->stream), and the read-back of its result (synchronises c->stream).  Nothing else
// may use c->mp_recs in between (the bound passes do not).
static const int FAR_NWG = 1024 > MP_NWG ? MP_NWG : 1024;
void k_farthest_enqueue(cge_ctx *c, const double *Xr, i64 n, i64 d, i64 src, const double *srow) {
    c->mp_recs.ensure(MP_NWG * 3);
    MaxRec *recs = reinterpret_cast<MaxRec *>(c->mp_recs.p);
    hipLaunchKernelGGL(farthest_kernel, dim3(FAR_NWG), dim3(256), 768 * sizeof(double), c->stream, Xr, n, d, src, srow, recs);
}
void k_farthest_collect(cge_ctx *c, double *best_val, i64 *best_i) {
    i64 bj;
    best_of_recs(c, reinterpret_cast<MaxRec *>(c->mp_recs.p), FAR_NWG, best_val, best_i, &bj);
}
void k_farthest(cge_ctx *c, const double *Xr, i64 n, i64 d, i64 src, double *best_val, i64 *best_i) {
    k_farthest_enqueue(c, Xr, n, d, src, nullptr);
    k_farthest_collect(c, best_val, best_i);
}

// ------------------------------------------------------------------------------------------------
// distances of S sampled pairs, exact dist() arithmetic (ascending k, unfused), times 1/den
__global__ void pair_dist_kernel(const double *__restrict__ Xr, i64 d, const i32 *__restrict__ pi,
                                 const i32 *__restrict__ pj, i64 S, double den, double *__restrict__ out) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < S; k += stride) {
        const double *a = Xr + (i64)pi[k] * d, *b = Xr + (i64)pj[k] * d;
        double s = 0.0;
        if (pi[k] != pj[k]) {
            for (i64 q = 0; q < d; q++) {
                const double df = __dsub_rn(a[q], b[q]);
                s = __dadd_rn(s, __dmul_rn(df, df));
            }
            s = sqrt(s);
        }
        out[k] = s / den;
    }
}
// option shard_rows: the rows of a chunk of sampled pairs are gathered from their owners into B (rows 2k, 2k + 1 of pair k)
// and the same arithmetic runs on the gathered rows
__global__ void pair_local_idx_kernel(const i32 *__restrict__ pi, const i32 *__restrict__ pj, i64 S, const i32 *__restrict__ glob2loc,
                                      i32 *__restrict__ idx) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S) return;
    idx[2 * k] = glob2loc[pi[k]];
    idx[2 * k + 1] = glob2loc[pj[k]];
}
__global__ void pair_dist_rows_kernel(const double *__restrict__ B, i64 d, const i32 *__restrict__ pi, const i32 *__restrict__ pj,
                                      i64 S, double den, double *__restrict__ out) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < S; k += stride) {
        const double *a = B + (2 * k) * d, *b = a + d;
        double s = 0.0;
        if (pi[k] != pj[k]) {
            for (i64 q = 0; q < d; q++) {
                const double df = __dsub_rn(a[q], b[q]);
                s = __dadd_rn(s, __dmul_rn(df, df));
            }
            s = sqrt(s);
        }
        out[k] = s / den;
    }
}
void k_pair_local_idx(cge_ctx *c, const i32 *pi, const i32 *pj, i64 S, const i32 *glob2loc, i32 *idx) {
    if (S > 0) hipLaunchKernelGGL(pair_local_idx_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, c->stream, pi, pj, S, glob2loc, idx);
}
void k_pair_dist_rows(cge_ctx *c, const double *B, i64 d, const i32 *pi, const i32 *pj, i64 S, double den, double *out) {
    if (S > 0) hipLaunchKernelGGL(pair_dist_rows_kernel, dim3(grid_for(S, 128)), dim3(128), 0, c->stream, B, d, pi, pj, S, den, out);
}
// The same for MANY pairs (round 4): a thread per pair walks two 1 KB rows eight bytes at a time -- 64 different rows per load
// instruction -- and took 2.3 - 3.5 ms per million pairs (config 4: 7 ms of a 70 ms step, found in the kernel sequence).
// Here a workgroup takes 16 pairs at a time: its four waves read the 32 rows coalesced (a wave per row, 512 bytes per load),
// write the squared differences (a_k - b_k)^2 -- the reference's own products -- to LDS, and 16 threads add them in
// ascending k, unfused, one accumulator per pair: dist()'s arithmetic (src/auxilary.jl:14-20), the same bits as the kernel above.
__global__ __launch_bounds__(256) void pair_dist_tile_kernel(const double *__restrict__ Xr, i64 d, const i32 *__restrict__ pi,
                                                             const i32 *__restrict__ pj, i64 S, double den,
                                                             double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sq[]; // [16][d + 1]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const i64 ld = d + 1;
    for (i64 p0 = (i64)blockIdx.x * 16; p0 < S; p0 += (i64)gridDim.x * 16) {
        // wave wv takes the pairs p0 + wv, p0 + wv + 4, ...: both rows of a pair in flight together, four pairs per wave
        i64 ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const i64 p = p0 + wv + 4 * u;
            ra[u] = p < S ? (i64)pi[p] : 0;
            rb[u] = p < S ? (i64)pj[p] : 0;
        }
        for (i64 k0 = 0; k0 < d; k0 += 64) {
            const i64 k = k0 + lane;
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = k < d ? Xr[ra[u] * d + k] : 0.0;
                b[u] = k < d ? Xr[rb[u] * d + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (k < d) {
                    const double df = __dsub_rn(a[u], b[u]);
                    sq[(i64)(wv + 4 * u) * ld + k] = __dmul_rn(df, df);
                }
        }
        __syncthreads();
        if (tid < 16 && p0 + tid < S) {
            const i64 p = p0 + tid;
            double acc = 0.0;
            if (pi[p] != pj[p]) {
                const double *q = sq + (i64)tid * ld;
                for (i64 k = 0; k < d; k++) acc = __dadd_rn(acc, q[k]);
                acc = sqrt(acc);
            }
            out[p] = acc / den;
        }
        __syncthreads();
    }
}
void k_pair_dist(cge_ctx *c, const double *Xr, i64 d, const i32 *pi, const i32 *pj, i64 S, double den, double *out) {
    if (S <= 0) return;
    const size_t lds = (size_t)16 * (d + 1) * sizeof(double);
    if (S >= 4096 && lds <= 64 * 1024) {
        const unsigned nb = (unsigned)std::min<i64>((S + 15) / 16, 256 * 16);
        hipLaunchKernelGGL(pair_dist_tile_kernel, dim3(nb), dim3(256), lds, c->stream, Xr, d, pi, pj, S, den, out);
        return;
    }
    hipLaunchKernelGGL(pair_dist_kernel, dim3(grid_for(S, 128)), dim3(128), 0, c->stream, Xr, d, pi, pj, S, den, out);
}

// landmark-sorted layout of the pruned diameter: landmark a owns the positions [soff[a], soff[a+1]) (its members in
// ascending order, padded to a multiple of 16 by repeating the last one); sub_land[pos/16] = a.
__global__ __launch_bounds__(256) void diameter_layout_kernel(const i32 *__restrict__ mem_off, const i32 *__restrict__ mem,
                                                              const i32 *__restrict__ soff, i32 *__restrict__ pos2node,
                                                              i32 *__restrict__ sub_land) {
    const i64 a = blockIdx.x;
    const i64 m0 = mem_off[a], cnt = mem_off[a + 1] - m0, p0 = soff[a], p1 = soff[a + 1];
    for (i64 p = p0 + threadIdx.x; p < p1; p += 256) {
        const i64 q = p - p0;
        pos2node[p] = cnt > 0 ? mem[m0 + (q < cnt ? q : cnt - 1)] : -1; // (no member here: another rank's landmark, option shard_rows)
        if ((q & 15) == 0) sub_land[p >> 4] = (i32)a;
    }
}
void k_diameter_layout(cge_ctx *c, const i32 *mem_off, const i32 *mem, const i32 *soff, i64 N, i32 *pos2node, i32 *sub_land,
                       i64 n_sub) {
    HIP_CHECK(hipMemsetAsync(sub_land, 0xFF, sizeof(i32) * n_sub, c->stream));
    hipLaunchKernelGGL(diameter_layout_kernel, dim3((unsigned)N), dim3(256), 0, c->stream, mem_off, mem, soff, pos2node,
                       sub_land);
}
// ids[p] = (global vertex id of position p) + 1, 0 for a position whose row lives on another rank (the ranks add the words)
__global__ void position_ids_kernel(const i32 *__restrict__ pos2node, const i32 *__restrict__ loc2glob, i64 npos, i32 *__restrict__ ids) {
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const i32 v = pos2node[p];
    ids[p] = v < 0 ? 0 : (loc2glob ? loc2glob[v] : v) + 1;
}
void k_position_ids(cge_ctx *c, const i32 *pos2node, const i32 *loc2glob, i64 npos, i32 *ids) {
    if (npos > 0) hipLaunchKernelGGL(position_ids_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, c->stream, pos2node, loc2glob, npos, ids);
}
