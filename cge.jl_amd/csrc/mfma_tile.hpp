// mfma_tile.hpp -- the fp64 MFMA 128x128 Gram-tile core shared by the distance kernels
// (kernels_dist.hip) and the covariance SYRK (kernels_lm.hip).  gfx950 only.
//
// MFMA operand maps (cdna_hip_programming.md §3, f64 note): A lane l -> A[row l&15][k l>>4],
// B lane l -> B[k l>>4][col l&15]; C/D lane l, reg r -> row (l>>4) + 4r, col l&15.
#pragma once
#include "common.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define MP_BM 128          // tile rows  (I side)
#define MP_BN 128          // tile cols  (J side)
#define MP_BK 16           // k-chunk
#define MP_LD (MP_BM + 16) // LDS row stride in doubles: (2*LD) % 64 == 32 => the 2 k-rows of a half-wave hit disjoint banks
#define MP_SB 32           // super-block edge in tiles


// One 128x128 Gram tile G = A_tile * B_tile^T over the whole contraction dimension, operands delivered by
// loader functors: la(kc, q) / lb(kc, q) return this thread's two adjacent elements (columns c2, c2+1 of the
// tile) of k-row kc*MP_BK + wave + 4q.  SAME = true: B is A (a diagonal tile of a SYRK) -- staged once.
// LDS: 2 stages x (A[,B]) x MP_BK x MP_LD doubles.  All 256 threads must call it (barriers inside).
// fa / fb turn what a loader returned into the staged pair at the moment it is stored to LDS, i.e. AFTER the MFMAs of
// the chunk in between: arithmetic on a loaded value inside the loader itself puts an `s_waitcnt vmcnt(0)` right behind
// the load and the prefetch is gone (the covariance kernel did that: half of its wave cycles were spent parked).
template <bool SAME, class LA, class LB, class FA, class FB>
__device__ __forceinline__ void gram_tile_128_ld(LA la, LB lb, FA fa, FB fb, i64 nchunk, double *lds, d4 (&acc)[4][4],
                                                 int wave, int c2, int wr, int wc, int lr, int lk) {
    const size_t stage_doubles = (size_t)(SAME ? 1 : 2) * MP_BK * MP_LD;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    decltype(la((i64)0, 0)) ra[4];
    decltype(lb((i64)0, 0)) rb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        ra[q] = la((i64)0, q);
        if (!SAME) rb[q] = lb((i64)0, q);
    }
    __syncthreads(); // the previous tile's readers are done with both stages
    {
        double *As = lds, *Bs = lds + (size_t)MP_BK * MP_LD;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            *reinterpret_cast<d2 *>(As + (wave + 4 * q) * MP_LD + c2) = fa(ra[q]);
            if (!SAME) *reinterpret_cast<d2 *>(Bs + (wave + 4 * q) * MP_LD + c2) = fb(rb[q]);
        }
    }
    __syncthreads();
    for (i64 kc = 0; kc < nchunk; kc++) {
        const int s = (int)(kc & 1);
        const bool more = kc + 1 < nchunk;
        if (more) { // issue the next chunk's global loads; they land while the MFMAs run
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ra[q] = la(kc + 1, q);
                if (!SAME) rb[q] = lb(kc + 1, q);
            }
        }
        const double *As = lds + (size_t)s * stage_doubles;
        const double *Bs = SAME ? As : As + (size_t)MP_BK * MP_LD;
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
                *reinterpret_cast<d2 *>(An + (wave + 4 * q) * MP_LD + c2) = fa(ra[q]);
                if (!SAME) *reinterpret_cast<d2 *>(Bn + (wave + 4 * q) * MP_LD + c2) = fb(rb[q]);
            }
        }
        __syncthreads();
    }
}

// The dense-operand form: pa/pb = this thread's first load address (operand base + wave*ld + tile offset +
// 2*lane); a k-row of a tile is 128 contiguous doubles; chunk kc covers k-rows [kc*MP_BK, (kc+1)*MP_BK).
__device__ __forceinline__ void gram_tile_128(const double *__restrict__ pa, const double *__restrict__ pb, i64 lda,
                                              i64 ldb, i64 nchunk, double *lds, d4 (&acc)[4][4], int wave, int c2,
                                              int wr, int wc, int lr, int lk) {
    auto same = [](d2 v) { return v; };
    gram_tile_128_ld<false>(
        [&](i64 kc, int q) { return *reinterpret_cast<const d2 *>(pa + (kc * MP_BK + 4 * q) * lda); },
        [&](i64 kc, int q) { return *reinterpret_cast<const d2 *>(pb + (kc * MP_BK + 4 * q) * ldb); }, same, same, nchunk,
        lds, acc, wave, c2, wr, wc, lr, lk);
}

// The diagonal tile of a SYRK, G = A_tile * A_tile^T: only the 36 upper-triangular 16x16 blocks of the 8x8 block
// grid are computed.  Wave w owns block-rows w and 7-w (8-w + w+1 = 9 blocks each way: the four waves are balanced);
// slot q < 8-w is block (w, w+q), slot q >= 8-w is block (7-w, 7-w + q-(8-w)).  Same staging as gram_tile_128_ld
// with SAME = true.  acc[q] has the MFMA C/D layout of its block: lane l, register r -> row (l>>4)+4r, col l&15.
template <int W, class LA, class FA>
__device__ __forceinline__ void syrk_tile_128_wave(LA la, FA fa, i64 nchunk, double *lds, d4 (&acc)[9], int wave, int c2,
                                                   int lr, int lk) {
    const size_t stage_doubles = (size_t)MP_BK * MP_LD;
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
    decltype(la((i64)0, 0)) ra[4];
#pragma unroll
    for (int q = 0; q < 4; q++) ra[q] = la((i64)0, q);
    __syncthreads(); // the previous tile's readers are done with both stages
#pragma unroll
    for (int q = 0; q < 4; q++) *reinterpret_cast<d2 *>(lds + (wave + 4 * q) * MP_LD + c2) = fa(ra[q]);
    __syncthreads();
    for (i64 kc = 0; kc < nchunk; kc++) {
        const int s = (int)(kc & 1);
        const bool more = kc + 1 < nchunk;
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; q++) ra[q] = la(kc + 1, q);
        }
        const double *As = lds + (size_t)s * stage_doubles;
#pragma unroll
        for (int ks = 0; ks < MP_BK / 4; ks++) {
            const double *rowp = As + (ks * 4 + lk) * MP_LD + lr;
            double f[8]; // operand fragments of block-columns W..7 (block-rows W and 7-W are among them)
#pragma unroll
            for (int b = W; b < 8; b++) f[b] = rowp[16 * b];
#pragma unroll
            for (int q = 0; q < 8 - W; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[W], f[W + q], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q <= W; q++)
                acc[8 - W + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[7 - W], f[7 - W + q], acc[8 - W + q], 0, 0, 0);
        }
        if (more) {
            double *An = lds + (size_t)(s ^ 1) * stage_doubles;
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<d2 *>(An + (wave + 4 * q) * MP_LD + c2) = fa(ra[q]);
        }
        __syncthreads();
    }
}
// block coordinates of slot q of wave w (see above)
__device__ __forceinline__ void syrk_slot_block(int w, int q, int &bi, int &bj) {
    if (q < 8 - w) { bi = w; bj = w + q; } else { bi = 7 - w; bj = 7 - w + (q - (8 - w)); }
}
