// kernels_fitp.hip -- the Chung-Lu fixed points of wGCL / wGCL_directed (src/divergence.jl:150-168, :434-467) as ONE
// persistent launch per alpha: the upper triangle of GD = (1 - D)^alpha lives in REGISTERS for the whole fit.
//
// Why: one iteration streams 8*N*(N+1)/2 algorithmic bytes (64 MB at N = 4000) for 2 flops per byte, a thousand
// times per score.  As one launch per iteration that is ~27 us (the matrix from the Infinity Cache plus a launch
// boundary); here the matrix is read once per alpha and an iteration costs a few KB of exchange between workgroups.
//
// Layout.  The matrix is cut into 64 x 64 tiles; tile (I, J), I <= J, belongs to one wave (tile t -> wave t mod 4G,
// slot t / 4G; G workgroups of 4 waves, at most TPW slots per wave).  Inside a tile lane l = 8*rq + cq holds the
// 8 x 8 block rows 8*rq.., columns 8*cq.. (64 doubles).  One pass over the block gives both products of the
// symmetric pair: p = (T_i*T_j)*g_ij is added to the row sum of i and to the column sum of j (the reference's own
// product, src/divergence.jl:155-158).  The 8 partial rows / columns of a lane are combined across the 8 lanes that
// share rq / cq by a transposing butterfly (4 + 2 + 1 exchanges), which leaves one finished row (column) per lane.
//
// An iteration:  A) every wave: tile products -> partial vectors P[block][other block][64];
//                B) workgroup sb (one 16-row quarter of a block): S_i = sum of the block's Nt partial vectors in a
//                   fixed order, T_i += eps*T_i*(w_i/S_i - 1), f = max|w_i - S_i| (:160-166);
//                C) `while f > delta`, decided by every workgroup from the same maxima.
// Two kernels share this scheme, fit_flow_kernel (undirected) and fit_flow_dir_kernel (directed): THE DATA IS ITS OWN SIGNAL --
// a slot that has not been delivered holds a sentinel and consumers poll the values they need (the header of fit_flow_kernel
// has the re-arming argument).  Every value that crosses workgroups is stored and loaded with agent-scope relaxed atomics
// (sc1: write-through stores, L1-bypassing loads).  Every spin is bounded: on a timeout the launch sets `fail`, every
// workgroup leaves, and the host falls back to one launch per iteration (fit_symtile_kernel + fit_symreduce_kernel below,
// which is also what score graphs beyond the register file use).  All sums have a fixed order: a run is bitwise
// reproducible.  (Rounds 1-4 carried two more forms -- two XCD-hierarchical grid barriers per iteration, 15 us, and per-block
// dependency counters, 10.7 us against 6.1 -- as options 3 / 4 of "fit_persistent"; removed in round 5.)
#include "common.hpp"
#include "pow_parts.hpp"

namespace {

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, RLX_AGENT); }
typedef double dbl2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, RLX_AGENT); }
// the same with a wave-uniform base and a 32-bit element index (the buffers of these kernels are far below 4 GB): the
// address is base (scalar registers) + one 32-bit vector offset, so an index that is invariant over the iterations costs
// one vector register to keep instead of two -- fit_flow_kernel runs at the 256-register limit
__device__ __forceinline__ double ld_sc1_at(const double *base, unsigned idx) {
    return ld_sc1(reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (idx << 3)));
}
__device__ __forceinline__ void st_sc1_at(double *base, unsigned idx, double v) {
    st_sc1(reinterpret_cast<double *>(reinterpret_cast<char *>(base) + (idx << 3)), v);
}

// combine v[0..8) across the 8 lanes that differ in lane bits SH, SH+1, SH+2: afterwards the lane whose three bits
// spell q holds sum_lanes v[q].  Additions are pairwise in a fixed tree.
template <int SH>
__device__ __forceinline__ double transpose_reduce8(const double (&v)[8], int lane) {
    double w4[4], w2[2];
    const bool h2 = (lane >> (SH + 2)) & 1, h1 = (lane >> (SH + 1)) & 1, h0 = (lane >> SH) & 1;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const double keep = h2 ? v[q + 4] : v[q], send = h2 ? v[q] : v[q + 4];
        w4[q] = keep + __shfl_xor(send, 4 << SH);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const double keep = h1 ? w4[q + 2] : w4[q], send = h1 ? w4[q] : w4[q + 2];
        w2[q] = keep + __shfl_xor(send, 2 << SH);
    }
    const double keep = h0 ? w2[1] : w2[0], send = h0 ? w2[0] : w2[1];
    return keep + __shfl_xor(send, 1 << SH);
}

// ---- the fit with the data as its own signal ------------------------------------------------------------------------
// No counters, no grid barrier: a slot that has not
// been delivered yet holds SENTINEL (a NaN payload that arithmetic never produces), and a consumer polls the values it
// needs until none of them is the sentinel.  A hand-off is then one write-through store and one L1-bypassing load --
// the producer neither drains its stores nor signals, the consumer needs no barrier between a poll and its loads.
// Slots are re-armed by their consumer, off the critical path:
//   T      ring of 4 vectors; T_{k+1} goes to ring[(k+1)&3].  The reducer of a quarter block arms ring[(k+3)&3] (it
//          held T_{k-1}, whose readers have all delivered P_k) while it publishes T_{k+1}; tiles poll that slot for
//          T_{k+3} only after they consumed T_{k+2}, which the same wave stored after an `s_waitcnt vmcnt(0)` that
//          covers the arming store.  T_0 is read from the caller's vector, the result is written to `Tout`.
//   P      two buffers by the parity of k; the reducer arms the entries it has just read, and the tile that rewrites
//          them two iterations later has by then consumed a T_{k+2} stored after the arming stores were drained.
//   f      three buffers (k mod 3): f_k is stored with T_{k+1}, read by every reducer in iteration k+1, armed again by
//          its writer in iteration k+2 -- one iteration before the next value lands there.
// Everything is armed by a fill before the launch.  `done` / `fail` are looked at every 64 polls; a poll never sees a stale
// value, only the sentinel or the value it waits for.
#define FLOW_SENTINEL_WORD 0x7FF8DEADu
#define FLOW_SENTINEL 0x7FF8DEAD7FF8DEADull
__device__ __forceinline__ bool armed(double v) { return (unsigned long long)__double_as_longlong(v) == FLOW_SENTINEL; }
// fit_flow_kernel: a converged reducer stores this where T_{k+1} would go, so the tile waves of workgroups WITHOUT a quarter
// block (G > 4 Nt), which poll that slot, leave at once instead of at their next look at `done` (every 64 polls)
#define FLOW_FINISHED 0x7FF8D0D07FF8D0D0ull
__device__ __forceinline__ bool finished_mark(double v) { return (unsigned long long)__double_as_longlong(v) == FLOW_FINISHED; }
__device__ __forceinline__ double sentinel() { return __longlong_as_double((long long)FLOW_SENTINEL); }
// every 64th unsuccessful poll: 1 = the fit is over, 2 = abandoned (wave-uniform)
// (`done` / `fail` are armed with the sentinel word like everything else of these kernels' buffers: they are SET when they
// hold 1 -- testing them against zero, as rounds 1-2 did, made every wait of more than 64 polls end the fit as "converged")
__device__ __forceinline__ int flow_check(unsigned &spins, unsigned *fail, unsigned *done, long long deadline) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) != 0) return 0;
    if (__hip_atomic_load(done, RLX_AGENT) == 1u) return 1;
    if (wall_clock64() > deadline || __hip_atomic_load(fail, RLX_AGENT) == 1u) {
        __hip_atomic_store(fail, 1u, RLX_AGENT);
        return 2;
    }
    return 0;
}
// max over a row of 16 lanes / over the wave, by DPP (no LDS crossbar); every lane gets the result
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double row16_max(double v) {
    v = fmax(v, dpp_mov64<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmax(v, dpp_mov64<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmax(v, dpp_mov64<0x141>(v)); // row_half_mirror
    v = fmax(v, dpp_mov64<0x140>(v)); // row_mirror
    return v;
}
__device__ __forceinline__ double readlane64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_max(double v) {
    v = row16_max(v);
    return fmax(fmax(readlane64(v, 0), readlane64(v, 16)), fmax(readlane64(v, 32), readlane64(v, 48)));
}
// NW waves per workgroup.  With 8 waves of one tile each (two waves per SIMD, 256 registers apiece) the whole tile sits in
// architectural VGPRs and the two waves of a SIMD hide each other's latencies; with 4 waves of two tiles half of the
// matrix lives in accumulation registers and is copied back before use.  The quarter blocks are always reduced by the
// first 256 threads, with the additions of fit_dataflow_kernel in the same order.
#define FLOW_RLD 66 // row stride (doubles) of the transposing-reduction scratch: 16-byte accesses of 8 lanes hit 8 banks groups
// ---- what rides on the fit's launch (round 5): the rest of the alpha's chain, src/divergence.jl:146, :170-176, :178-213, :226-234 ----
// FUSED = true (landmark mode, sweep relabelled by community; wgcl_host.cpp decides):
//   prologue  g = 2^(alpha * log2(1 - D)) from the stored logarithm (pow_parts.hpp: exp2_matrix_upper_kernel's own function and
//             bits) -- the power matrix is never written;
//   epilogue  when the fit has converged the tile and (from the ring) the final T are at hand: the products
//             P_ij = (T_i T_j) g_ij are summed per (row community, column community) rectangle of the tile in the fixed order
//             of bvec_tile_kernel (kernels_fit.hip) -> `partial`, which bvec_bins_kernel folds into vect_B; and the first
//             CGE_PARTIAL_BLOCKS workgroups tally the local score's sampled pairs exactly as auc_landmark_kernel's blocks do.
// Neither launch re-reads the matrix: bvec_rows / bvec_zsum / bvec_fold / exp2_matrix_upper / auc_landmark and one read +
// one write of GD per alpha are gone from the chain.
#define FLOW_NP 32 // pieces (runs of one community inside an 8-column chunk) per 64-column block at most: LDS per wave FLOW_NP x 66 doubles
struct FlowFused { // by value: what the prologue needs, and where the epilogue finds the rest (device memory: loaded after the loop,
                   // so that none of it is held in registers across it -- the loop runs at the register limit)
    const double *Lh; const float *Ll; double alpha; // log2(1 - D) in two parts (k_pow_prepare)
    const cge_fit_fused *epi;                        // device copy of the sweep's table for this sample set
    int want;                                        // bit 0: vect_B's tile partials, bit 1: the local score's tallies
};
// the sum of the first 256 threads' values, block_sum_256's additions (kernels_fit.hip); every thread of the workgroup calls it
__device__ __forceinline__ double flow_sum_256(double v, double *sh, int tid) {
    if (tid < 256) sh[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}
template <int TPW, int NW, bool FUSED, int NSB> // NSB: quarter blocks a workgroup may have to reduce (1 when 4*Nt <= G)
__global__ __launch_bounds__(64 * NW) void fit_flow_kernel(const double *__restrict__ GD, i64 N, int Nt, const double *T0,
                                                            double *Tout, i64 Tld, const double *__restrict__ w, double eps,
                                                            double delta, int max_iters, double *ring, double *P, double *fq,
                                                            unsigned *sync, int *flags, long long timeout_ticks, int test_naps,
                                                            const FlowFused fz) {
    // (4*Nt <= NSB*G, checked by the host)
    __shared__ double red[2][NSB][16][17]; // by the parity of k: no barrier is needed to recycle it
    __shared__ double fred[2][4];
    __shared__ __attribute__((aligned(16))) double tsh[NW][2][64];     // per wave: T of the tile's row block / column block
    // per wave: the two transposing reductions of an iteration ([2][8][FLOW_RLD]); the fused epilogue stages its pieces there
    constexpr int RSH_W = FUSED ? (FLOW_NP * FLOW_RLD > 2 * 8 * FLOW_RLD ? FLOW_NP * FLOW_RLD : 2 * 8 * FLOW_RLD) : 2 * 8 * FLOW_RLD;
    __shared__ __attribute__((aligned(16))) double rsh_all[NW][RSH_W];
    // fused epilogue, filled by the prologue (so the epilogue waits for no global load): per wave the communities of its tile's
    // row / column runs, and {segI, segJ (the rows / columns that start a run, 64-bit masks), fc[I], fc[J], ns[J], base[I][J]}
    static_assert(!FUSED || TPW == 1, "the fused form keeps one tile per wave");
    __shared__ i32 segtab[FUSED ? NW : 1][2][64];
    __shared__ i32 ehdr[FUSED ? NW : 1][8];
    __shared__ int lds_exit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, G = gridDim.x;
    const int rq = lane >> 3, cq = lane & 7;
    const int NT = Nt * (Nt + 1) / 2;
    const i64 Psz = (i64)Nt * Nt * 64;
    unsigned *fail = sync + 1, *done = sync + 2;
    long long deadline = wall_clock64() + timeout_ticks; // re-armed at every iteration: it bounds one hand-off, not the whole fit

    double g[TPW][8][8];
    int tI[TPW], tJ[TPW];
#ifdef CGE_FLOW_CLOCK
    const long long ck0 = wall_clock64();
#endif
    if (FUSED && (fz.want & 2) && tid < 256) { // the local score's two powers per sample, ahead of everything (nothing is live yet): the
        const cge_fit_fused *ep0 = fz.epi; // epilogue's tally then waits for loads only.  Block vb = this workgroup's, as below
        const i64 S0 = ep0->S;
        const int GR0 = G < 4 * Nt ? G : 4 * Nt;
        if (wg < GR0)
            for (int vb = wg; vb < CGE_PARTIAL_BLOCKS; vb += GR0)
                for (i64 q = (i64)vb * 256 + tid; q < S0; q += (i64)CGE_PARTIAL_BLOCKS * 256) {
                    ep0->apw[q] = pow(1.0 - ep0->dpos[q], fz.alpha);
                    ep0->apw[S0 + q] = pow(1.0 - ep0->dneg[q], fz.alpha);
                }
    }
    if (FUSED && (fz.want & 1)) { // vect_B's tile geometry into LDS (this wave's tile: slot 0)
        const cge_fit_fused *ep0 = fz.epi;
        const int t = wg * NW + wave;
        if (t < NT) { // uniform per wave
            int I = 0, rem = t;
            while (rem >= Nt - I) { rem -= Nt - I; I++; }
            const int J = I + rem;
            const i64 vI = (i64)64 * I + lane, vJ = (i64)64 * J + lane;
            // communities of the block's rows / columns (-1 beyond the matrix): a set bit of segI / segJ starts a run
            const i32 cI = vI < N ? ep0->comm[vI] : -1, cJ = vJ < N ? ep0->comm[vJ] : -1;
            const i32 cIp = __shfl_up(cI, 1), cJp = __shfl_up(cJ, 1);
            const unsigned long long segI = __ballot(lane == 0 || cI != cIp), segJ = __ballot(lane == 0 || cJ != cJp);
            const unsigned long long below = (1ull << lane) - 1ull;
            if ((segI >> lane) & 1ull) segtab[wave][0][__popcll(segI & below)] = cI;
            if ((segJ >> lane) & 1ull) segtab[wave][1][__popcll(segJ & below)] = cJ;
            if (lane == 0) {
                ehdr[wave][0] = (i32)(unsigned)segI; ehdr[wave][1] = (i32)(unsigned)(segI >> 32);
                ehdr[wave][2] = (i32)(unsigned)segJ; ehdr[wave][3] = (i32)(unsigned)(segJ >> 32);
                ehdr[wave][4] = ep0->fc[I]; ehdr[wave][5] = ep0->fc[J]; ehdr[wave][6] = ep0->ns[J]; ehdr[wave][7] = ep0->base[I * Nt + J];
            }
        }
    }
    if (test_naps > 0 && (wg * NW + wave) < NT) // testing (option fit_persistent_test_delay): the tile waves start late
        for (int q = 0; q < test_naps; q++) __builtin_amdgcn_s_sleep(127);
#pragma unroll
    for (int s = 0; s < TPW; s++) {
        const int t = (wg * NW + wave) + s * NW * G;
        tI[s] = -1;
        tJ[s] = -1;
        if (t < NT) {
            int I = 0, rem = t;
            while (rem >= Nt - I) { rem -= Nt - I; I++; }
            tI[s] = I;
            tJ[s] = I + rem;
        }
        if (FUSED) { // the stored logarithm -> this alpha's power, in place (the element stays 0.0 outside the matrix).  Every
            // load is unconditional (indices clamped into the matrix, the result selected afterwards): no divergent branches
            const int Ic = tI[s] < 0 ? 0 : tI[s], Jc = tJ[s] < 0 ? 0 : tJ[s];
            const unsigned Nu = (unsigned)N;
            // The logarithm sits TILE-BLOCKED (k_pow_prepare, blocked form): the 64 doubles (and 64 floats) of a lane's 8 x 8 block are
            // contiguous, a wave's tile is 32 KB (16 KB) of consecutive memory -- 16-byte loads, every line used whole.  (Rounds
            // 4-5 read the row-major matrix: 64 eight-byte loads per lane at a stride of 64 B, 12-26 us of prologue.)
            const size_t tb = ((size_t)(tI[s] < 0 ? 0 : (wg * NW + wave) + s * NW * G) * 64 + (size_t)lane) * 64;
            const dbl2f *bh = reinterpret_cast<const dbl2f *>(fz.Lh + tb);
#pragma unroll
            for (int a = 0; a < 8; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const dbl2f v2 = bh[a * 4 + b];
                    g[s][a][2 * b] = v2.x;
                    g[s][a][2 * b + 1] = v2.y;
                }
            typedef float flt4f __attribute__((ext_vector_type(4)));
            const flt4f *bl = reinterpret_cast<const flt4f *>(fz.Ll + tb);
#pragma unroll
            for (int h = 0; h < 2; h++) { // the float parts by half tiles: 32 registers beside the 128 of the tile
                float ll[4][8];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        const flt4f v4 = bl[(4 * h + a) * 2 + b];
                        ll[a][4 * b] = v4.x; ll[a][4 * b + 1] = v4.y; ll[a][4 * b + 2] = v4.z; ll[a][4 * b + 3] = v4.w;
                    }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const bool in = tI[s] >= 0 && 64u * Ic + 8u * rq + 4 * h + a < Nu && 64u * Jc + 8u * cq + b < Nu;
                        const double e = exp2_parts(fz.alpha, g[s][4 * h + a][b], ll[a][b]);
                        g[s][4 * h + a][b] = in ? e : 0.0;
                    }
            }
            continue;
        }
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const i64 row = (i64)64 * tI[s] + 8 * rq + a;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const i64 col = (i64)64 * tJ[s] + 8 * cq + b;
                g[s][a][b] = (tI[s] >= 0 && row < N && col < N) ? GD[row * N + col] : 0.0;
            }
        }
    }
#ifdef CGE_FLOW_CLOCK // (a build flag, diagnostics only: wall-clock stamps of the launch's sections, printed by two waves)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long ck1 = wall_clock64();
#endif
    // the rows this thread updates (threads 0..15 only): the current iterate and the target stay in registers
    const int r16 = tid & 15, qg = (tid >> 4) & 15;
    const bool reducer = tid < 256;
    double tcur[NSB], wrow[NSB];
#pragma unroll
    for (int i = 0; i < NSB; i++) {
        const int sb = wg + i * G;
        const i64 row = (i64)64 * (sb >> 2) + 16 * (sb & 3) + r16;
        const bool mine = sb < 4 * Nt && tid < 16 && row < N;
        tcur[i] = mine ? T0[row] : 0.0;
        wrow[i] = mine ? w[row] : 0.0;
    }
    if (tid == 0) lds_exit = 0;
    __syncthreads();

    int k = 0, converged = 0, failed = 0, left_on = 0;
    if (timeout_ticks <= 0) max_iters = 0; // test hook: abandon at once
    for (;;) {
        deadline = wall_clock64() + timeout_ticks;
        if (k >= max_iters) { failed = 1; break; }
        const double *Tk = (k == 0) ? T0 : ring + (i64)(k & 3) * Tld;
        double *Pk = P + (i64)(k & 1) * Psz;
        int bad = 0; // wave-uniform: 1 = over, 2 = abandoned
        // The arming stores of the previous iteration (and its T) have landed before anything of this iteration is stored:
        // waited for here, where the wave would otherwise only wait for the other workgroups' T to become visible.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- 1. tile products with T_k, each wave on its own -----------------------------------------------------------
#pragma unroll
        for (int s = 0; s < TPW; s++) {
            if (tI[s] < 0 || bad) continue; // uniform per wave
            const int I = tI[s], J = tJ[s];
            double vi, vj;
            unsigned spins = 0;
            for (;;) {
                vi = ld_sc1_at(Tk, 64u * (unsigned)I + (unsigned)lane);
                vj = ld_sc1_at(Tk, 64u * (unsigned)J + (unsigned)lane);
                if (__any(finished_mark(vi) || finished_mark(vj))) { bad = 1; break; } // the fit ended with iteration k - 1
                if (__all(!armed(vi) && !armed(vj))) break;
                bad = flow_check(spins, fail, done, deadline);
                if (bad) break;
            }
            if (bad) continue;
            tsh[wave][0][lane] = vi;
            tsh[wave][1][lane] = vj;
            __builtin_amdgcn_wave_barrier();
            double ti[8], tj[8], pr[8], pc[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                ti[q] = tsh[wave][0][8 * rq + q];
                tj[q] = tsh[wave][1][8 * cq + q];
                pr[q] = 0.0;
                pc[q] = 0.0;
            }
#pragma unroll
            for (int a = 0; a < 8; a++)
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    pr[a] = fma(g[s][a][b], tj[b], pr[a]); // factored: the row's own T_i is applied by the reducer
                    pc[b] = fma(g[s][a][b], ti[a], pc[b]);
                }
            // The transposing reductions of transpose_reduce8 (same pairs, same bits) through LDS: the partial of lane
            // (rq, cq) for row 8*rq + a goes to R[cq][8*rq + a], lane l then adds the eight partials of row l as
            // ((u0+u4)+(u2+u6)) + ((u1+u5)+(u3+u7)); the same for the columns with the roles of rq and cq exchanged.
            double(*R)[FLOW_RLD] = reinterpret_cast<double(*)[FLOW_RLD]>(rsh_all[wave]);
            double(*Cc)[FLOW_RLD] = reinterpret_cast<double(*)[FLOW_RLD]>(rsh_all[wave] + 8 * FLOW_RLD);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                R[cq][8 * rq + q] = pr[q];
                Cc[rq][8 * cq + q] = pc[q];
            }
            __builtin_amdgcn_wave_barrier();
            double u[8], v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                u[q] = R[q][lane];
                v[q] = Cc[q][lane];
            }
            __builtin_amdgcn_wave_barrier();
            const double rsum = ((u[0] + u[4]) + (u[2] + u[6])) + ((u[1] + u[5]) + (u[3] + u[7]));
            st_sc1_at(Pk, ((unsigned)I * (unsigned)Nt + (unsigned)J) * 64u + (unsigned)lane, rsum);
            if (I != J) {
                const double csum = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
                st_sc1_at(Pk, ((unsigned)J * (unsigned)Nt + (unsigned)I) * 64u + (unsigned)lane, csum);
            }
        }
        // ---- 2. the quarter blocks this workgroup reduces -----------------------------------------------------------------
        bool stop = false;
#pragma unroll
        for (int i = 0; i < NSB; i++) {
            const int sb = wg + i * G;
            if (sb >= 4 * Nt) break; // uniform
            const int b = sb >> 2, rib = 16 * (sb & 3) + r16;
            const bool fcheck = k > 0 && i == 0; // `while diff > delta` on f of iteration k-1
            double pv[4] = {0.0, 0.0, 0.0, 0.0}, fv = 0.0, fx[2] = {0.0, 0.0};
            const double *fp = fq + (i64)((k + 2) % 3) * 4 * Nt; // f of iteration k-1
            if (!bad && fcheck && reducer) { // stored an iteration ago: asked for ahead of the partial vectors
#pragma unroll
                for (int u = 0; u < 2; u++)
                    if (tid + 256 * u < 4 * Nt) fx[u] = ld_sc1_at(fp, (unsigned)(tid + 256 * u));
            }
            if (!bad && reducer) {
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int q = qg + 16 * u;
                        if (q < Nt) {
                            pv[u] = ld_sc1_at(Pk, ((unsigned)b * (unsigned)Nt + (unsigned)q) * 64u + (unsigned)rib);
                            ok = ok && !armed(pv[u]);
                        }
                    }
                    if (__all(ok)) break;
                    bad = flow_check(spins, fail, done, deadline);
                    if (bad) break;
                }
            }
            if (!bad && fcheck && reducer) {
                unsigned spins = 0;
                for (;;) {
                    if (__all(!armed(fx[0]) && !armed(fx[1]))) break;
                    bad = flow_check(spins, fail, done, deadline);
                    if (bad) break;
#pragma unroll
                    for (int u = 0; u < 2; u++)
                        if (tid + 256 * u < 4 * Nt) fx[u] = ld_sc1_at(fp, (unsigned)(tid + 256 * u));
                }
                fv = fmax(fx[0], fx[1]);
            }
            if (bad && lane == 0) atomicOr(&lds_exit, bad);
            if (reducer) {
                if (fcheck) {
                    fv = wave_max(fv);
                    if (lane == 0) fred[k & 1][wave] = fv;
                }
                red[k & 1][i][qg][r16] = ((pv[0] + pv[1]) + pv[2]) + pv[3];
            }
            __syncthreads();
            const int ex = lds_exit;
            if (ex) { failed = (ex & 2) != 0; converged = !failed; stop = true; break; } // uniform
            if (fcheck) {
                const double f = fmax(fmax(fred[k & 1][0], fred[k & 1][1]), fmax(fred[k & 1][2], fred[k & 1][3]));
                if (!(f > delta)) { // uniform; nothing of iteration k is published -- only the mark that ends the waiting
                    if (tid < 16) {
#pragma unroll
                        for (int i2 = 0; i2 < NSB; i2++) {
                            const int sb2 = wg + i2 * G;
                            if (sb2 < 4 * Nt)
                                st_sc1(ring + (i64)((k + 1) & 3) * Tld + (i64)64 * (sb2 >> 2) + 16 * (sb2 & 3) + r16,
                                       __longlong_as_double((long long)FLOW_FINISHED));
                        }
                    }
                    converged = 1;
                    stop = true;
                    break;
                }
            }
            if (tid < 16) { // the update first: it is what the other workgroups wait for
                double S = red[k & 1][i][0][r16];
#pragma unroll
                for (int u = 1; u < 16; u++) S += red[k & 1][i][u][r16];
                const i64 row = (i64)64 * b + rib;
                double fr = 0.0, tnew = 0.0;
                if (row < N) {
                    S *= tcur[i]; // S_i = T_i * sum_j g_ij T_j: the tiles summed g * T
                    tnew = tcur[i] + (eps * tcur[i]) * (wrow[i] / S - 1.0);
                    fr = fabs(wrow[i] - S);
                }
                st_sc1_at(ring + (i64)((k + 1) & 3) * Tld, (unsigned)row, tnew);
                tcur[i] = tnew;
                fr = row16_max(fr);
                if (r16 == 0) st_sc1_at(fq + (i64)(k % 3) * 4 * Nt, (unsigned)sb, fr);
                st_sc1_at(ring + (i64)((k + 3) & 3) * Tld, (unsigned)row, sentinel());
                if (r16 == 0) st_sc1_at(fq + (i64)((k + 1) % 3) * 4 * Nt, (unsigned)sb, sentinel());
            }
            if (reducer) {
#pragma unroll
                for (int u = 0; u < 4; u++) { // arm the entries just read (their next writer is two iterations away)
                    const int q = qg + 16 * u;
                    if (q < Nt) st_sc1_at(Pk, ((unsigned)b * (unsigned)Nt + (unsigned)q) * 64u + (unsigned)rib, sentinel());
                }
            }
        }
        if (wg >= 4 * Nt && bad) { left_on = bad; break; } // no quarter block, no barrier in the loop: each wave leaves on its own
        if (stop) {
            if (converged && tid == 0) __hip_atomic_store(done, 1u, RLX_AGENT);
            break;
        }
        k++;
    }
    if (converged && tid < 16) { // T_k: every reducer holds its rows
#pragma unroll
        for (int i = 0; i < NSB; i++) {
            const int sb = wg + i * G;
            const i64 row = (i64)64 * (sb >> 2) + 16 * (sb & 3) + r16;
            if (sb < 4 * Nt && row < N) Tout[row] = tcur[i];
        }
    }
    if (wg == 0 && tid == 0) {
        flags[0] = converged;
        flags[1] = k; // iterations done: T_k is final
        flags[2] = failed || !converged;
        flags[3] = 0;
    }
#ifdef CGE_FLOW_CLOCK
    const long long ck2 = wall_clock64();
#endif
    if (!FUSED) return;
    // ---- the rest of the alpha's chain, from the tile and the final iterate --------------------------------------------------
    // A workgroup with a quarter block left the loop as a whole, at iteration k (T_k is final).  A wave of a workgroup without
    // one left on its own when it met the end mark / `done` while polling T_k: for it T_{k-1} is final.  T_final sits in its
    // ring slot, complete and not re-armed: every tile consumed it, and the converging iteration published nothing.
    // (lane ids the compiler cannot see through: nothing of the epilogue is computed ahead of the loop and kept in registers
    // across it -- the loop runs at the 256-register limit)
    int lane_e = lane, tid_e = tid;
    asm volatile("" : "+v"(lane_e), "+v"(tid_e));
    const cge_fit_fused *ep = fz.epi;
    const int want = fz.want;
    const int rq_e = lane_e >> 3, cq_e = lane_e & 7, wave_e = tid_e >> 6;
    const bool wg_reduces = wg < 4 * Nt;
    const bool ok = wg_reduces ? (converged != 0) : (left_on == 1 && k >= 1); // (wave-uniform; an abandoned fit computes nothing)
    const int kfin = wg_reduces ? k : k - 1;
    const double *Tf = (kfin == 0) ? T0 : ring + (i64)(kfin & 3) * Tld;
    if ((want & 1) && ok) {
        double(*R)[FLOW_RLD] = reinterpret_cast<double(*)[FLOW_RLD]>(rsh_all[wave_e]);
        double *const partial = ep->partial; // (the one global load of this part, asked for ahead of the arithmetic)
        {
            constexpr int s = 0;
            if (tI[s] >= 0) { // uniform per wave
            const int I = tI[s], J = tJ[s];
            const unsigned long long segI = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(ehdr[wave_e][1]) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane(ehdr[wave_e][0]);
            const unsigned long long segJ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(ehdr[wave_e][3]) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane(ehdr[wave_e][2]);
            // T_final of the tile's row and column block: what this wave staged for its last products (tsh is written once per
            // iteration, after the poll that a finished fit never passes)
            double ti[8], tj[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                ti[q] = tsh[wave_e][0][8 * rq_e + q];
                tj[q] = tsh[wave_e][1][8 * cq_e + q];
            }
            // (a) pieces: a piece is a run of columns of one community inside this lane_e's 8-column chunk; its eight row sums
            // (ascending column) go to R[piece][row].  pm: the columns that start a piece.
            const unsigned long long pm = segJ | 0x0101010101010101ull;
            const unsigned startb = (unsigned)(pm >> (8 * cq_e)) & 0xFFu, endb = (startb >> 1) | 0x80u;
            int pc = __popcll(pm & ((1ull << (8 * cq_e)) - 1ull));
            double cs[8];
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool st = (startb >> b) & 1u;
#pragma unroll
                for (int a = 0; a < 8; a++) {
                    const bool dead = (I == J) && (8 * rq_e + a > 8 * cq_e + b); // the reference sums j >= i only (:229)
                    const double pv = dead ? 0.0 : __dmul_rn(__dmul_rn(ti[a], tj[b]), g[s][a][b]);
                    cs[a] = st ? pv : __dadd_rn(cs[a], pv);
                }
                if ((endb >> b) & 1u) {
#pragma unroll
                    for (int a = 0; a < 8; a++) R[pc][8 * rq_e + a] = cs[a];
                    pc++;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // (b) lane = row: the pieces of a community's run, ascending, into the run's sum -> R[run][row].  All pieces are
            // requested first (the tile's registers are free by now), so the adds do not each wait for an LDS round trip; a lane
            // touches its own column of R only.
            int nrunJ = 0;
            {
                const int npieces = __popcll(pm);
                double xp[FLOW_NP];
#pragma unroll
                for (int q = 0; q < FLOW_NP; q++) xp[q] = q < npieces ? R[q][lane_e] : 0.0;
                unsigned long long m = pm;
                int run = -1;
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < FLOW_NP; q++) {
                    if (q < npieces) { // uniform
                        const int cpos = __builtin_ctzll(m);
                        m &= m - 1ull;
                        if ((segJ >> cpos) & 1ull) {
                            if (run >= 0) R[run][lane_e] = acc;
                            run++;
                            acc = xp[q];
                        } else
                            acc = __dadd_rn(acc, xp[q]);
                    }
                }
                R[run][lane_e] = acc;
                nrunJ = run + 1;
            }
            __builtin_amdgcn_wave_barrier();
            // (c) lane = column run: the rows of a row run, ascending -> one partial per (row community, column community);
            // again every operand is requested before the first add
            {
                const int run = lane_e < nrunJ ? lane_e : nrunJ - 1;
                const i32 ccol = segtab[wave_e][1][run];
                const bool live = lane_e < nrunJ && ccol >= 0;
                const i32 fcI = __builtin_amdgcn_readfirstlane(ehdr[wave_e][4]), fcJ = __builtin_amdgcn_readfirstlane(ehdr[wave_e][5]),
                          nsJ = __builtin_amdgcn_readfirstlane(ehdr[wave_e][6]);
                double *out = partial + (i64)__builtin_amdgcn_readfirstlane(ehdr[wave_e][7]) + (ccol - fcJ);
                double xr[64];
#pragma unroll
                for (int r = 0; r < 64; r += 2) {
                    const dbl2f v2 = *reinterpret_cast<const dbl2f *>(&R[run][r]);
                    xr[r] = v2.x;
                    xr[r + 1] = v2.y;
                }
                double acc = 0.0;
                int rrun = -1;
                i32 crow = -1;
#pragma unroll
                for (int r = 0; r < 64; r++) { // uniform
                    if ((segI >> r) & 1ull) {
                        if (rrun >= 0 && crow >= 0 && live) out[(i64)(crow - fcI) * nsJ] = acc;
                        rrun++;
                        crow = segtab[wave_e][0][rrun];
                        acc = xr[r];
                    } else
                        acc = __dadd_rn(acc, xr[r]);
                }
                if (crow >= 0 && live) out[(i64)(crow - fcI) * nsJ] = acc;
            }
            }
        }
    }
#ifdef CGE_FLOW_CLOCK
    const long long ck3 = wall_clock64();
#endif
    if ((want & 2) && wg_reduces) { // (`converged` is uniform over such a workgroup: the barriers below are safe)
        double *sh = &rsh_all[0][0]; // >= 256 doubles; every wave_e is past its own use of it once the barrier below is passed
        const int GR = G < 4 * Nt ? G : 4 * Nt;
        __syncthreads();
        for (int vb = wg; vb < CGE_PARTIAL_BLOCKS; vb += GR) { // uniform
            double num = 0.0;
            const i64 S = ep->S;
            if (converged && tid_e < 256)
                for (i64 q = (i64)vb * 256 + tid_e; q < S; q += (i64)CGE_PARTIAL_BLOCKS * 256) {
                    // auc_landmark_kernel's arithmetic on the prepared operands (k_auc_prepare) and this launch's own powers
                    i32 ix[4];
                    double f[8];
#pragma unroll
                    for (int u = 0; u < 4; u++) ix[u] = ep->aidx[u * S + q];
#pragma unroll
                    for (int u = 0; u < 8; u++) f[u] = ep->afac[u * S + q];
                    const double pp = ep->apw[q], pn = ep->apw[S + q], wq = ep->wts[q];
                    const double t_i = ld_sc1(Tf + ix[0]), t_j = ld_sc1(Tf + ix[1]), t_u = ld_sc1(Tf + ix[2]), t_v = ld_sc1(Tf + ix[3]);
                    const double ai = (t_i * f[0]) / f[1], aj = (t_j * f[2]) / f[3];
                    const double au = (t_u * f[4]) / f[5], av = (t_v * f[6]) / f[7];
                    const double pos = (ai * aj) * pp;
                    const double neg = (au * av) * pn;
                    num += (pos > neg ? 1.0 : 0.0) * wq;
                }
            num = flow_sum_256(num, sh, tid_e);
            if (tid_e == 0 && converged) { ep->auc_part[2 * vb] = num; ep->auc_part[2 * vb + 1] = ep->aden[vb]; }
        }
    }
#ifdef CGE_FLOW_CLOCK
    if (lane_e == 0 && ((wg == 0 && wave_e == 0) || (wg == 130 && wave_e == 5)))
        printf("flow clock wg %d wave %d: prologue %lld  loop %lld (%d iterations)  vect_B epilogue %lld  tallies %lld  (10 ns ticks)\n", wg, wave_e,
               ck1 - ck0, ck2 - ck1, k, ck3 - ck2, wall_clock64() - ck3);
#endif
}

// ---- one Chung-Lu iteration per launch pair, over the UPPER tiles only --------------------------------------------
// For landmark counts beyond the register file (N > ~4900: config 5 has 12 000, exact mode N = n) the fit is one
// launch (pair) per iteration and bound by the matrix it streams.  fit_step_kernel (kernels_fit.hip) reads whole rows,
// 8 N^2 bytes per iteration; this form reads every 64 x 64 tile on or above the diagonal once and uses it for both
// products of the symmetric pair, exactly as the persistent kernels do with their register-resident tiles: same lane
// layout (lane 8*rq + cq holds the 8 x 8 block), same transposing reductions, partial vectors P[block][other block][64].
// The second kernel adds a block's Nt partial vectors (four interleaved sequential sums, then ((s0+s1)+s2)+s3 -- a fixed
// order for any Nt), updates T and publishes f: an atomic max on the bit pattern of a non-negative double (exact,
// order-free) in fring[k % 3]; the launch of iteration k + 1 starts by reading f_k -- when it is <= delta the fit is over
// (iters = k + 1, the final T is the one iteration k wrote, exactly the reference's `while diff > delta`) and every later
// launch of the batch returns at `*done`; fring[(k+1) % 3] is cleared by iteration k for iteration k + 1.  Traffic per iteration: 4 N^2 (+ N^2 / 4 for the partial vectors) instead of 8 N^2 bytes.
__global__ __launch_bounds__(256) void fit_symtile_kernel(const double *__restrict__ GD, const double *__restrict__ T, i64 N,
                                                          int Nt, i64 NT, double delta, int k,
                                                          const unsigned long long *__restrict__ fring,
                                                          const int *__restrict__ done, double *__restrict__ P) {
    if (*done) return;
    if (k > 0) {
        const double fprev = __longlong_as_double((long long)fring[(k - 1) % 3]);
        if (!(fprev > delta)) return; // the fit ended with iteration k - 1 (the reduce kernel records it)
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rq = lane >> 3, cq = lane & 7;
    const i64 t = (i64)blockIdx.x * 4 + wave;
    if (t >= NT) return;
    // tile t of the row-major upper triangle: start(I) = I*Nt - I*(I-1)/2
    const double bb = 2.0 * Nt + 1.0;
    i64 I = (i64)((bb - sqrt(bb * bb - 8.0 * (double)t)) * 0.5);
    if (I < 0) I = 0;
    if (I > Nt - 1) I = Nt - 1;
    while (I + 1 < Nt && (I + 1) * Nt - (I + 1) * I / 2 <= t) I++;
    while (I > 0 && I * Nt - I * (I - 1) / 2 > t) I--;
    const i64 J = I + (t - (I * Nt - I * (I - 1) / 2));
    const i64 r0 = 64 * I + 8 * rq, c0 = 64 * J + 8 * cq;
    double g[8][8];
    if ((N & 1) == 0 && r0 + 8 <= N && c0 + 8 <= N) { // the common case: 16-byte loads, no guards
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const dbl2f *row = reinterpret_cast<const dbl2f *>(GD + (r0 + a) * N + c0);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const dbl2f v = row[b];
                g[a][2 * b] = v.x;
                g[a][2 * b + 1] = v.y;
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int b = 0; b < 8; b++) g[a][b] = (r0 + a < N && c0 + b < N) ? GD[(r0 + a) * N + c0 + b] : 0.0;
    }
    double ti[8], tj[8], pr[8], pc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        ti[q] = r0 + q < N ? T[r0 + q] : 0.0;
        tj[q] = c0 + q < N ? T[c0 + q] : 0.0;
        pr[q] = 0.0;
        pc[q] = 0.0;
    }
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) {
            pr[a] = fma(g[a][b], tj[b], pr[a]); // factored: the row's own T_i is applied by the reducer
            pc[b] = fma(g[a][b], ti[a], pc[b]);
        }
    const double rsum = transpose_reduce8<0>(pr, lane); // row 8*rq + cq of the tile
    P[(I * Nt + J) * 64 + lane] = rsum;
    if (I != J) {
        const double csum = transpose_reduce8<3>(pc, lane); // column 8*cq + rq of the tile
        P[(J * Nt + I) * 64 + 8 * cq + rq] = csum;
    }
}
__global__ __launch_bounds__(256) void fit_symreduce_kernel(const double *__restrict__ P, const double *__restrict__ T,
                                                            double *__restrict__ Tout, const double *__restrict__ w, i64 N,
                                                            int Nt, double eps, double delta, int k,
                                                            unsigned long long *__restrict__ fring, int *__restrict__ done,
                                                            int *__restrict__ iters) {
    __shared__ double red[4][64];
    if (*done) return;
    if (k > 0) {
        const double fprev = __longlong_as_double((long long)fring[(k - 1) % 3]);
        if (!(fprev > delta)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) { *iters = k; *done = 1; }
            return;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { fring[(k + 1) % 3] = 0ULL; *iters = k + 1; }
    const int r = threadIdx.x & 63, u = threadIdx.x >> 6;
    const i64 b = blockIdx.x;
    const double *Pb = P + b * Nt * 64 + r;
    double acc = 0.0;
    int q = u;
    for (; q + 12 < Nt; q += 16) { // four loads in flight; the additions stay in q order
        const double p0 = Pb[(i64)q * 64], p1 = Pb[(i64)(q + 4) * 64], p2 = Pb[(i64)(q + 8) * 64], p3 = Pb[(i64)(q + 12) * 64];
        acc += p0; acc += p1; acc += p2; acc += p3;
    }
    for (; q < Nt; q += 4) acc += Pb[(i64)q * 64];
    red[u][r] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        const i64 row = 64 * b + r;
        double f = 0.0;
        if (row < N) {
            const double ti = T[row], wi = w[row];
            const double S = ti * (((red[0][r] + red[1][r]) + red[2][r]) + red[3][r]); // the tiles summed g * T
            Tout[row] = ti + (eps * ti) * (wi / S - 1.0);
            f = fabs(wi - S);
        }
        f = wave_max(f);
        if (r == 0) {
            const unsigned long long fb = (unsigned long long)__double_as_longlong(f);
            if (fb > __hip_atomic_load(&fring[k % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&fring[k % 3], fb);
        }
    }
}
// ---- the directed fit with the data as its own signal (the scheme of fit_flow_kernel) -------------------------------------
// Per tile four partial vectors (e1 = (Tin_i*Tout_j)*g feeds Sin_i and Sout_j, e2 = (Tin_j*Tout_i)*g feeds Sout_i and
// Sin_j, the diagonal term twice), computed in two passes over the register block so that a wave of 256 registers holds
// the tile; the same additions in the same order as fit_dataflow_dir_kernel.  T ring: 4 x (Tin, Tout); P: 2 parities x
// (Sin, Sout) x Nt x Nt x 64; f: 3 x 4Nt.  T_0 = T0 (Tin, Tout: Tld doubles each, zero beyond N); on convergence the
// reducers write their rows of the final iterate to Tin_out / Tout_out (N doubles each), otherwise nothing is written.
template <int NW, int NSB> // NSB: quarter blocks a workgroup may have to reduce (1 when 4*Nt <= G)
__global__ __launch_bounds__(64 * NW) void fit_flow_dir_kernel(const double *__restrict__ GD, i64 N, int Nt, const double *T0,
                                                                double *Tin_out, double *Tout_out, i64 Tld,
                                                                const double *__restrict__ deg_in,
                                                                const double *__restrict__ deg_out, double eps0, double f0,
                                                                double delta, int max_iters, double *ring, double *P,
                                                                double *fq, unsigned *sync, int *flags,
                                                                long long timeout_ticks, int test_naps) {
    __shared__ double red[2][NSB][2][16][17]; // [parity of k][quarter block][Sin / Sout]
    __shared__ double fred[2][4];
    __shared__ __attribute__((aligned(16))) double tsh[NW][4][64];          // Tin_I, Tout_I, Tin_J, Tout_J
    __shared__ __attribute__((aligned(16))) double rsh[NW][2][8][FLOW_RLD]; // one pass: a row and a column reduction
    __shared__ int lds_exit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, G = gridDim.x;
    const int rq = lane >> 3, cq = lane & 7;
    const int NT = Nt * (Nt + 1) / 2;
    const i64 Pstride = (i64)Nt * Nt * 64, Psz = 2 * Pstride;
    unsigned *fail = sync + 1, *done = sync + 2;
    long long deadline = wall_clock64() + timeout_ticks; // re-armed at every iteration: it bounds one hand-off, not the whole fit

    double g[8][8];
    int tI = -1, tJ = -1;
    if (test_naps > 0 && (wg * NW + wave) < NT) // testing (option fit_persistent_test_delay): the tile waves start late
        for (int q = 0; q < test_naps; q++) __builtin_amdgcn_s_sleep(127);
    {
        const int t = wg * NW + wave;
        if (t < NT) {
            int I = 0, rem = t;
            while (rem >= Nt - I) { rem -= Nt - I; I++; }
            tI = I;
            tJ = I + rem;
        }
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const i64 row = (i64)64 * tI + 8 * rq + a;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const i64 col = (i64)64 * tJ + 8 * cq + b;
                g[a][b] = (tI >= 0 && row < N && col < N) ? GD[row * N + col] : 0.0;
            }
        }
    }
    const int r16 = tid & 15, qg = (tid >> 4) & 15;
    const bool reducer = tid < 256;
    double tin_cur[NSB], tout_cur[NSB], din[NSB], dout[NSB];
#pragma unroll
    for (int i = 0; i < NSB; i++) {
        const int sb = wg + i * G;
        const i64 row = (i64)64 * (sb >> 2) + 16 * (sb & 3) + r16;
        const bool mine = sb < 4 * Nt && tid < 16 && row < N;
        tin_cur[i] = mine ? T0[row] : 0.0;
        tout_cur[i] = mine ? T0[Tld + row] : 0.0;
        din[i] = mine ? deg_in[row] : 0.0;
        dout[i] = mine ? deg_out[row] : 0.0;
    }
    if (tid == 0) lds_exit = 0;
    __syncthreads();

    int k = 0, converged = 0, failed = 0;
    double eps = eps0, fprev = f0;
    if (timeout_ticks <= 0) max_iters = 0; // test hook: abandon at once
    for (;;) {
        deadline = wall_clock64() + timeout_ticks;
        if (k >= max_iters) { failed = 1; break; }
        const double *Tk = (k == 0) ? T0 : ring + (i64)(k & 3) * 2 * Tld; // Tin at Tk, Tout at Tk + Tld
        double *Pk = P + (i64)(k & 1) * Psz;
        int bad = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the arming stores of the previous iteration have landed
        // ---- 1. tile products with T_k ---------------------------------------------------------------------------------
        if (tI >= 0) {
            const int I = tI, J = tJ;
            double v0, v1, v2, v3;
            unsigned spins = 0;
            for (;;) {
                v0 = ld_sc1(Tk + 64 * I + lane);
                v1 = ld_sc1(Tk + Tld + 64 * I + lane);
                v2 = ld_sc1(Tk + 64 * J + lane);
                v3 = ld_sc1(Tk + Tld + 64 * J + lane);
                if (__all(!armed(v0) && !armed(v1) && !armed(v2) && !armed(v3))) break;
                bad = flow_check(spins, fail, done, deadline);
                if (bad) break;
            }
            if (!bad) {
                tsh[wave][0][lane] = v0;
                tsh[wave][1][lane] = v1;
                tsh[wave][2][lane] = v2;
                tsh[wave][3][lane] = v3;
                __builtin_amdgcn_wave_barrier();
                const bool diag_tile = I == J;
                double(*R)[FLOW_RLD] = rsh[wave][0];
                double(*Cc)[FLOW_RLD] = rsh[wave][1];
#pragma unroll 1
                for (int pass = 0; pass < 2; pass++) { // one copy of the code: the two passes share their registers
                    // FACTORED (as fit_dataflow_dir_kernel): the row's own factor Tin_i / Tout_i is applied by the reducer.
                    // pass 0: g * Tout -> rows: Sin partial of block I, columns: Sin partial of block J
                    // pass 1: g * Tin  -> rows: Sout partial of block I, columns: Sout partial of block J
                    const double *trow = tsh[wave][1 - pass], *tcol = tsh[wave][3 - pass]; // Tout_I, Tout_J / Tin_I, Tin_J
                    double ta[8], tb[8], pr[8], pc[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        ta[q] = trow[8 * rq + q];
                        tb[q] = tcol[8 * cq + q];
                        pr[q] = 0.0;
                        pc[q] = 0.0;
                    }
#pragma unroll
                    for (int a = 0; a < 8; a++)
#pragma unroll
                        for (int b = 0; b < 8; b++) {
                            double gv = g[a][b];
                            if (diag_tile && rq == cq && a == b) gv += gv; // the j == i term counts twice
                            pr[a] = fma(gv, tb[b], pr[a]);
                            pc[b] = fma(gv, ta[a], pc[b]);
                        }
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        R[cq][8 * rq + q] = pr[q];
                        Cc[rq][8 * cq + q] = pc[q];
                    }
                    __builtin_amdgcn_wave_barrier();
                    double u[8], v[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        u[q] = R[q][lane];
                        v[q] = Cc[q][lane];
                    }
                    __builtin_amdgcn_wave_barrier();
                    const double rsum = ((u[0] + u[4]) + (u[2] + u[6])) + ((u[1] + u[5]) + (u[3] + u[7]));
                    // pass 0 -> the Sin plane (0), pass 1 -> the Sout plane (1): rows for block I, columns for block J
                    st_sc1(Pk + (i64)pass * Pstride + ((i64)I * Nt + J) * 64 + lane, rsum);
                    if (!diag_tile) {
                        const double csum = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
                        st_sc1(Pk + (i64)pass * Pstride + ((i64)J * Nt + I) * 64 + lane, csum);
                    }
                }
            }
        }
        // ---- 2. the quarter blocks this workgroup reduces -----------------------------------------------------------------
        bool stop = false;
#pragma unroll
        for (int i = 0; i < NSB; i++) {
            const int sb = wg + i * G;
            if (sb >= 4 * Nt) break; // uniform
            const int b = sb >> 2, rib = 16 * (sb & 3) + r16;
            const bool fcheck = k > 0 && i == 0;
            double pi[4] = {0.0, 0.0, 0.0, 0.0}, po[4] = {0.0, 0.0, 0.0, 0.0}, fv = 0.0, fx[2] = {0.0, 0.0};
            const double *fp = fq + (i64)((k + 2) % 3) * 4 * Nt; // f of iteration k-1
            if (!bad && fcheck && reducer) {
#pragma unroll
                for (int u = 0; u < 2; u++)
                    if (tid + 256 * u < 4 * Nt) fx[u] = ld_sc1(fp + tid + 256 * u);
            }
            if (!bad && reducer) {
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int q = qg + 16 * u;
                        if (q < Nt) {
                            pi[u] = ld_sc1(Pk + ((i64)b * Nt + q) * 64 + rib);
                            po[u] = ld_sc1(Pk + Pstride + ((i64)b * Nt + q) * 64 + rib);
                            ok = ok && !armed(pi[u]) && !armed(po[u]);
                        }
                    }
                    if (__all(ok)) break;
                    bad = flow_check(spins, fail, done, deadline);
                    if (bad) break;
                }
            }
            if (!bad && fcheck && reducer) {
                unsigned spins = 0;
                for (;;) {
                    if (__all(!armed(fx[0]) && !armed(fx[1]))) break;
                    bad = flow_check(spins, fail, done, deadline);
                    if (bad) break;
#pragma unroll
                    for (int u = 0; u < 2; u++)
                        if (tid + 256 * u < 4 * Nt) fx[u] = ld_sc1(fp + tid + 256 * u);
                }
                fv = fmax(fx[0], fx[1]);
            }
            if (bad && lane == 0) atomicOr(&lds_exit, bad);
            if (reducer) {
                if (fcheck) {
                    fv = wave_max(fv);
                    if (lane == 0) fred[k & 1][wave] = fv;
                }
                red[k & 1][i][0][qg][r16] = ((pi[0] + pi[1]) + pi[2]) + pi[3];
                red[k & 1][i][1][qg][r16] = ((po[0] + po[1]) + po[2]) + po[3];
            }
            __syncthreads();
            const int ex = lds_exit;
            if (ex) { failed = (ex & 2) != 0; converged = !failed; stop = true; break; } // uniform
            if (fcheck) { // the step-size rule replayed from the same f history by every workgroup, then `while diff > delta`
                const double f = fmax(fmax(fred[k & 1][0], fred[k & 1][1]), fmax(fred[k & 1][2], fred[k & 1][3]));
                if (f > fprev) eps *= 0.99;
                fprev = f;
                if (!(f > delta)) { converged = 1; stop = true; break; }
            }
            if (tid < 16) {
                double Si = red[k & 1][i][0][0][r16], So = red[k & 1][i][1][0][r16];
#pragma unroll
                for (int u = 1; u < 16; u++) { Si += red[k & 1][i][0][u][r16]; So += red[k & 1][i][1][u][r16]; }
                Si *= tin_cur[i]; // the row's own factor (the tiles summed g * Tout / g * Tin)
                So *= tout_cur[i];
                const i64 row = (i64)64 * b + rib;
                double fr = 0.0, nin = 0.0, nout = 0.0;
                if (row < N) {
                    nin = tin_cur[i];
                    nout = tout_cur[i];
                    if (din[i] > 0) { nin = tin_cur[i] + (eps * tin_cur[i]) * (din[i] / Si - 1.0); fr = fmax(fr, fabs(din[i] - Si)); }
                    if (dout[i] > 0) { nout = tout_cur[i] + (eps * tout_cur[i]) * (dout[i] / So - 1.0); fr = fmax(fr, fabs(dout[i] - So)); }
                }
                double *Tn = ring + (i64)((k + 1) & 3) * 2 * Tld, *Ta = ring + (i64)((k + 3) & 3) * 2 * Tld;
                st_sc1(Tn + row, nin);
                st_sc1(Tn + Tld + row, nout);
                tin_cur[i] = nin;
                tout_cur[i] = nout;
                fr = row16_max(fr);
                if (r16 == 0) st_sc1(fq + (i64)(k % 3) * 4 * Nt + sb, fr);
                st_sc1(Ta + row, sentinel());
                st_sc1(Ta + Tld + row, sentinel());
                if (r16 == 0) st_sc1(fq + (i64)((k + 1) % 3) * 4 * Nt + sb, sentinel());
            }
            if (reducer) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = qg + 16 * u;
                    if (q < Nt) {
                        st_sc1(Pk + ((i64)b * Nt + q) * 64 + rib, sentinel());
                        st_sc1(Pk + Pstride + ((i64)b * Nt + q) * 64 + rib, sentinel());
                    }
                }
            }
        }
        if (wg >= 4 * Nt && bad) break;
        if (stop) {
            if (converged && tid == 0) __hip_atomic_store(done, 1u, RLX_AGENT);
            break;
        }
        k++;
    }
    if (converged && tid < 16) {
#pragma unroll
        for (int i = 0; i < NSB; i++) {
            const int sb = wg + i * G;
            const i64 row = (i64)64 * (sb >> 2) + 16 * (sb & 3) + r16;
            if (sb < 4 * Nt && row < N) {
                Tin_out[row] = tin_cur[i];
                Tout_out[row] = tout_cur[i];
            }
        }
    }
    if (wg == 0 && tid == 0) {
        flags[0] = converged;
        flags[1] = k;
        flags[2] = failed || !converged;
        flags[3] = 0;
    }
}

} // namespace

void k_fit_sym_step(cge_ctx *c, const double *GD, const double *Tin, double *Tout, const double *w, i64 N, double eps,
                    double delta, int k, unsigned long long *fring, int *done, int *iters) {
    const int Nt = (int)((N + 63) / 64);
    const i64 NT = (i64)Nt * (Nt + 1) / 2;
    c->fp_P.ensure((size_t)Nt * Nt * 64);
    ScopedKernelTimer t(c, "fit_symv");
    hipLaunchKernelGGL(fit_symtile_kernel, dim3((unsigned)((NT + 3) / 4)), dim3(256), 0, c->stream, GD, Tin, N, Nt, NT, delta, k,
                       fring, done, c->fp_P.p);
    hipLaunchKernelGGL(fit_symreduce_kernel, dim3((unsigned)Nt), dim3(256), 0, c->stream, c->fp_P.p, Tin, Tout, w, N, Nt, eps,
                       delta, k, fring, done, iters);
}

// geometry of fit_flow_kernel for N vertices on this device; false: the form does not apply
static bool flow_geometry(i64 N, i64 Tld, int *G_out, int *NW_out, int *tpw_out) {
    const int Nt = (int)((N + 63) / 64);
    const i64 NT = (i64)Nt * (Nt + 1) / 2;
    int dev = 0, cus = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0 || Tld < (i64)Nt * 64) return false;
    // one tile per wave: 4 waves per CU while they cover the triangle, else 8 (two per SIMD, 256 registers apiece).  Eight
    // tiles per CU is what the register file holds: N <= 4032 on 256 CUs (two tiles on each of four waves hold no more).
    const int NW = (NT <= (i64)4 * cus) ? 4 : 8;
    const int G = (int)std::min<i64>(cus, std::max<i64>((NT + NW - 1) / NW, (i64)4 * Nt));
    const int tpw = (int)((NT + (i64)NW * G - 1) / ((i64)NW * G));
    if (tpw > 1 || Nt > 64 || 4 * Nt > 2 * G) return false; // (a reducer adds up to 64 partial vectors, a workgroup reduces at most two quarter blocks)
    *G_out = G; *NW_out = NW; *tpw_out = tpw;
    return true;
}
bool k_fit_flow_fused_applies(cge_ctx *c, i64 N) {
    int G = 0, NW = 0, tpw = 0;
    return flow_geometry(N, (N + 63) / 64 * 64, &G, &NW, &tpw) && tpw == 1 && 4 * (int)((N + 63) / 64) <= G;
}
// where the hand-off slots of fit_flow_kernel for N vertices live and how they are armed (the buffer is made if need be)
bool k_fit_flow_arm_region(cge_ctx *c, i64 N, i64 Tld, uint4 **ptr, i64 *n16, unsigned *word) {
    int G = 0, NW = 0, tpw = 0;
    if (!flow_geometry(N, Tld, &G, &NW, &tpw)) return false;
    const int Nt = (int)((N + 63) / 64);
    const size_t psz = (size_t)Nt * Nt * 64, n_ring = (size_t)4 * Tld, n_fq = (size_t)3 * 4 * Nt, n_sync = 32;
    const size_t doubles = n_sync + n_ring + n_fq + 2 * psz;
    if (doubles % 2) return false; // (16-byte units)
    c->fp_flow.ensure(doubles);
    *ptr = reinterpret_cast<uint4 *>(c->fp_flow.p);
    *n16 = (i64)(doubles / 2);
    *word = FLOW_SENTINEL_WORD;
    return true;
}
bool k_fit_flow_enqueue(cge_ctx *c, const double *GD, i64 N, const double *T0, double *Tout, i64 Tld, const double *w,
                        double eps, double delta, int *dev_flags, const cge_fit_fused *ff, const cge_fit_fused *ff_dev) {
    const bool fused = ff != nullptr;
    FlowFused fz{};
    if (fused) {
        static_assert(CGE_FLOW_NP == FLOW_NP, "piece limit");
        fz.Lh = ff->Lh; fz.Ll = ff->Ll; fz.alpha = ff->alpha;
        fz.epi = ff_dev;
        fz.want = (ff->partial ? 1 : 0) | (ff->auc_part ? 2 : 0);
    }
    const int Nt = (int)((N + 63) / 64);
    int G = 0, NW = 0, tpw = 0;
    if (!flow_geometry(N, Tld, &G, &NW, &tpw)) return false;
    if (fused && (tpw != 1 || 4 * Nt > G)) return false; // (callers ask k_fit_flow_fused_applies first)
    const size_t psz = (size_t)Nt * Nt * 64, n_ring = (size_t)4 * Tld, n_fq = (size_t)3 * 4 * Nt;
    const size_t n_sync = 32; // fail / done words, armed with everything else
    c->fp_flow.ensure(n_sync + n_ring + n_fq + 2 * psz);
    const bool one = 4 * Nt <= G; // every workgroup reduces at most one quarter block
    const void *fn;
#define FLOW_PICK(F, B) (NW == 8 ? (const void *)fit_flow_kernel<1, 8, F, B> : (const void *)fit_flow_kernel<1, 4, F, B>)
    if (fused) fn = NW == 8 ? (const void *)fit_flow_kernel<1, 8, true, 1> : (const void *)fit_flow_kernel<1, 4, true, 1>;
    else fn = one ? FLOW_PICK(false, 1) : FLOW_PICK(false, 2);
#undef FLOW_PICK
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * NW, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return false;
    }
    hipStream_t st = c->stream;
    const i64 arm_words = (i64)(2 * (n_sync + n_ring + n_fq + 2 * psz));
    if (c->flow_armed_words != arm_words) // (else: armed by the last launch of the previous alpha's chain, bins_js_kernel)
        HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->fp_flow.p, (int)FLOW_SENTINEL_WORD, (size_t)arm_words, st));
    c->flow_armed_words = 0;
    const double *aGD = GD, *aT0 = T0, *aW = w;
    double *aTout = Tout, *aRing = c->fp_flow.p + n_sync, *aFq = aRing + n_ring, *aP = aFq + n_fq;
    i64 aN = N, aTld = Tld;
    int aNt = Nt, aMax = 2000000;
    double aEps = eps, aDelta = delta;
    unsigned *aSync = (unsigned *)c->fp_flow.p;
    int *aFlags = dev_flags;
    long long aTicks = c->opt_fit_test_timeout ? 0LL : CGE_FIT_TIMEOUT_TICKS; // per iteration (0: the test hook)
    int aNaps = c->opt_fit_test_delay;
    void *args[] = {&aGD, &aN, &aNt, &aT0, &aTout, &aTld, &aW, &aEps, &aDelta, &aMax, &aRing, &aP, &aFq, &aSync, &aFlags, &aTicks,
                    &aNaps, &fz};
    hipError_t e;
    {
        ScopedKernelTimer tm(c, "fit_persistent");
        e = hipLaunchKernel(fn, dim3((unsigned)G), dim3(64 * NW), args, 0, st);
    }
    if (e != hipSuccess) CGE_THROW(CGE_E_HIP, "fit launch failed: %s", hipGetErrorString(e));
    return true;
}

// Directed fit of one alpha from Tin / Tout (N doubles each, updated in place on success).  Returns false when the
// persistent path does not apply or was abandoned; Tin / Tout are then untouched.
bool k_fit_persistent_dir(cge_ctx *c, const double *GD, i64 N, double *Tin, double *Tout, const double *deg_in,
                          const double *deg_out, double eps0, double f0, double delta, i64 *iters, int *dev_flags, bool *enqueued_only) {
    if (enqueued_only) *enqueued_only = false;
    const int Nt = (int)((N + 63) / 64);
    const i64 NT = (i64)Nt * (Nt + 1) / 2, Tld = (i64)Nt * 64;
    int dev = 0, cus = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) return false;
    hipStream_t st = c->stream;
    c->fp_Td.ensure((size_t)4 * Tld);
    if (Nt <= 64) { // the data as its own signal: one tile per wave on 4 or 8 waves per CU
        const int NW = (NT <= (i64)4 * cus) ? 4 : 8;
        const int Gf = (int)std::min<i64>(cus, std::max<i64>((NT + NW - 1) / NW, (i64)4 * Nt));
        if (NT <= (i64)NW * Gf && 4 * Nt <= 2 * Gf) {
            const bool one = 4 * Nt <= Gf; // every workgroup reduces at most one quarter block
            const void *fn = NW == 8 ? (one ? (const void *)fit_flow_dir_kernel<8, 1> : (const void *)fit_flow_dir_kernel<8, 2>)
                                     : (one ? (const void *)fit_flow_dir_kernel<4, 1> : (const void *)fit_flow_dir_kernel<4, 2>);
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * NW, 0) == hipSuccess && per_cu >= 1) {
                const size_t pstride = (size_t)Nt * Nt * 64, n_sync = 32, n_ring = (size_t)8 * Tld, n_fq = (size_t)3 * 4 * Nt,
                             n_p = 4 * pstride;
                c->fp_flow.ensure(n_sync + n_ring + n_fq + n_p);
                c->flow_armed_words = 0; // (the undirected fit's slots live in the same buffer)
                c->fp_flags.ensure(4);
                HIP_CHECK(hipMemsetAsync(c->fp_Td.p, 0, sizeof(double) * 2 * Tld, st));
                HIP_CHECK(hipMemcpyAsync(c->fp_Td.p, Tin, sizeof(double) * N, hipMemcpyDeviceToDevice, st));
                HIP_CHECK(hipMemcpyAsync(c->fp_Td.p + Tld, Tout, sizeof(double) * N, hipMemcpyDeviceToDevice, st));
                HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->fp_flow.p, (int)FLOW_SENTINEL_WORD,
                                            2 * (n_sync + n_ring + n_fq + n_p), st));
                const double *aGD = GD, *aT0 = c->fp_Td.p, *aDi = deg_in, *aDo = deg_out;
                double *aTi = Tin, *aTo = Tout, *aRing = c->fp_flow.p + n_sync, *aFq = aRing + n_ring, *aP = aFq + n_fq;
                i64 aN = N, aTld = Tld;
                int aNt = Nt, aMax = 2000000;
                double aEps = eps0, aF0 = f0, aDelta = delta;
                unsigned *aSync = (unsigned *)c->fp_flow.p;
                int *aFlags = dev_flags ? dev_flags : c->fp_flags.p;
                long long aTicks = c->opt_fit_test_timeout ? 0LL : CGE_FIT_TIMEOUT_TICKS;
                int aNaps = c->opt_fit_test_delay;
                void *args[] = {&aGD, &aN, &aNt, &aT0, &aTi, &aTo, &aTld, &aDi, &aDo, &aEps, &aF0, &aDelta, &aMax, &aRing, &aP, &aFq,
                                &aSync, &aFlags, &aTicks, &aNaps};
                hipError_t e;
                {
                    ScopedKernelTimer tm(c, "fit_persistent");
                    e = hipLaunchKernel(fn, dim3((unsigned)Gf), dim3(64 * NW), args, 0, st);
                }
                if (e != hipSuccess) CGE_THROW(CGE_E_HIP, "directed fit launch failed: %s", hipGetErrorString(e));
                if (dev_flags) { // enqueue only: the caller reads the verdict (flags) behind whatever it queues after the fit
                    *enqueued_only = true;
                    *iters = 0;
                    return true;
                }
                int hf[4];
                HIP_CHECK(hipMemcpyAsync(hf, c->fp_flags.p, sizeof(hf), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (hf[2] || !hf[0]) { // abandoned: Tin / Tout are untouched
                    note_fit_fallback(c);
                    return false;
                }
                *iters = hf[1];
                return true;
            }
            (void)hipGetLastError();
        }
    }
    return false; // (the form does not apply to this size: the caller iterates with one launch pair per iteration)
}
