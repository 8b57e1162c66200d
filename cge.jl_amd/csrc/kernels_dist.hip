// kernels_dist.hip -- distance kernels.
//   k_dist_matrix : D[i,j] = dist(i,j,embed), diagonal = distances[i]      src/divergence.jl:79-91
//   k_minmax_upper / k_normalise : lo,hi = extrema(D); D = (D-lo)/(hi-lo)  src/divergence.jl:92-93
//   k_max_pair    : arg-max of all n(n-1)/2 pairwise distances (the `hi` of
//                   extrema(full_graph_D), never materialised)             src/divergence.jl:104-113
//   k_pair_dist   : distances of sampled pairs                              src/divergence.jl:189,198
#include "common.hpp"

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
__global__ void minmax_final_kernel(const double *__restrict__ part, int nb, double *__restrict__ lo_hi) {
    double lo = INFINITY, hi = -INFINITY;
    for (int b = 0; b < nb; b++) {
        lo = fmin(lo, part[2 * b]);
        hi = fmax(hi, part[2 * b + 1]);
    }
    lo_hi[0] = lo;
    lo_hi[1] = hi;
}
void k_minmax_upper(cge_ctx *c, const double *D, i64 N, double *lo_hi) {
    const int nb = (int)grid_for(N * N, 256, 1024);
    DevBuf<double> part;
    part.ensure((size_t)2 * nb);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, c->stream, D, N, part.p);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(1), 0, c->stream, part.p, nb, lo_hi);
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
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define MP_BM 128          // tile rows  (I side)
#define MP_BN 128          // tile cols  (J side)
#define MP_BK 16           // k-chunk
#define MP_LD (MP_BM + 16) // LDS row stride in doubles: (2*LD) % 64 == 32 => the 2 k-rows of a half-wave hit disjoint banks
#define MP_SB 32           // super-block edge in tiles

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

__global__ __launch_bounds__(256, 2) void max_pair_kernel(const double *__restrict__ Xc,
                                                          const double *__restrict__ rnorm, i64 n, i64 ldn, i64 dpad,
                                                          int part, int nparts, MaxRec *__restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) double lds[]; // 2 stages x (A + B) x MP_BK x MP_LD
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1; // wave sub-tile: rows wr*64.., cols wc*64..
    const int lr = lane & 15, lk = lane >> 4;
    const i64 nT = ldn / MP_BM;
    const i64 nS = (nT + MP_SB - 1) / MP_SB;
    const i64 total = nS * (nS + 1) / 2 * MP_SB * MP_SB;
    const i64 nchunk = dpad / MP_BK;
    double best = -1.0;
    i64 best_i = 0, best_j = 0;
    const size_t stage_doubles = (size_t)2 * MP_BK * MP_LD;
    // loader geometry: slot q of this thread is k-row (wave + 4q), doubles [2*lane, 2*lane+1]
    const int c2 = lane * 2;

    for (i64 t = blockIdx.x; t < total; t += gridDim.x) {
        i64 SI, I, J;
        tile_from_linear(t, nS, SI, I, J);
        if (I >= nT || J >= nT || J < I || (SI % nparts) != part) continue; // uniform per workgroup
        const i64 i0 = I * MP_BM, j0 = J * MP_BN;
        d4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
        d2 ra[4], rb[4];
        const double *pa = Xc + (i64)wave * ldn + i0 + c2;
        const double *pb = Xc + (i64)wave * ldn + j0 + c2;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ra[q] = *reinterpret_cast<const d2 *>(pa + (i64)(4 * q) * ldn);
            rb[q] = *reinterpret_cast<const d2 *>(pb + (i64)(4 * q) * ldn);
        }
        __syncthreads(); // the previous tile's readers are done with both stages
        {
            double *As = lds, *Bs = lds + (size_t)MP_BK * MP_LD;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                *reinterpret_cast<d2 *>(As + (wave + 4 * q) * MP_LD + c2) = ra[q];
                *reinterpret_cast<d2 *>(Bs + (wave + 4 * q) * MP_LD + c2) = rb[q];
            }
        }
        __syncthreads();
        for (i64 kc = 0; kc < nchunk; kc++) {
            const int s = (int)(kc & 1);
            const bool more = kc + 1 < nchunk;
            if (more) { // issue the next chunk's global loads; they land while the MFMAs run
                const i64 koff = (kc + 1) * MP_BK * ldn;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    ra[q] = *reinterpret_cast<const d2 *>(pa + koff + (i64)(4 * q) * ldn);
                    rb[q] = *reinterpret_cast<const d2 *>(pb + koff + (i64)(4 * q) * ldn);
                }
            }
            const double *As = lds + (size_t)s * stage_doubles;
            const double *Bs = As + (size_t)MP_BK * MP_LD;
#pragma unroll
            for (int ks = 0; ks < MP_BK / 4; ks++) {
                double af[4], bf[4];
#pragma unroll
                for (int a = 0; a < 4; a++) af[a] = As[(ks * 4 + lk) * MP_LD + wr * 64 + a * 16 + lr];
#pragma unroll
                for (int b = 0; b < 4; b++) bf[b] = Bs[(ks * 4 + lk) * MP_LD + wc * 64 + b * 16 + lr];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
            if (more) {
                double *An = lds + (size_t)(s ^ 1) * stage_doubles;
                double *Bn = An + (size_t)MP_BK * MP_LD;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    *reinterpret_cast<d2 *>(An + (wave + 4 * q) * MP_LD + c2) = ra[q];
                    *reinterpret_cast<d2 *>(Bn + (wave + 4 * q) * MP_LD + c2) = rb[q];
                }
            }
            __syncthreads();
        }
        // epilogue: dist^2 and running max.  acc[a][b][r]: row = i0 + wr*64 + a*16 + lk + 4r, col = j0 + wc*64 + b*16 + lr
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
    // workgroup reduction of (best, i, j): value max, ties -> smallest (i,j)
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

// Shard `part` of `nparts` owns the super-block rows SI with SI % nparts == part (balanced to
// one super-row; no tile is visited twice across shards).
void k_max_pair(cge_ctx *c, const double *Xc, const double *rnorm, i64 n, i64 ldn, i64 dpad, int part, int nparts,
                double *best_val, i64 *best_i, i64 *best_j) {
    const int nwg = 512;
    DevBuf<MaxRec> recs;
    recs.ensure(nwg);
    const size_t lds = (size_t)2 * 2 * MP_BK * MP_LD * sizeof(double);
    {
        ScopedKernelTimer t(c, "max_pair_dist");
        hipLaunchKernelGGL(max_pair_kernel, dim3(nwg), dim3(256), lds, c->stream, Xc, rnorm, n, ldn, dpad, part,
                           nparts, recs.p);
    }
    std::vector<MaxRec> h(nwg);
    HIP_CHECK(hipMemcpyAsync(h.data(), recs.p, sizeof(MaxRec) * nwg, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    double bv = -1.0;
    i64 bi = 0, bj = 0;
    for (int k = 0; k < nwg; k++)
        if (h[k].val > bv || (h[k].val == bv && (h[k].i < bi || (h[k].i == bi && h[k].j < bj)))) {
            bv = h[k].val; bi = h[k].i; bj = h[k].j;
        }
    *best_val = bv; *best_i = bi; *best_j = bj;
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
void k_pair_dist(cge_ctx *c, const double *Xr, i64 d, const i32 *pi, const i32 *pj, i64 S, double den, double *out) {
    if (S <= 0) return;
    hipLaunchKernelGGL(pair_dist_kernel, dim3(grid_for(S, 128)), dim3(128), 0, c->stream, Xr, d, pi, pj, S, den, out);
}
