// kernels_fits.hip -- the Chung-Lu fixed point of wGCL (src/divergence.jl:150-168) as ONE persistent launch per alpha with the
// WHOLE matrix GD = (1 - D)^alpha on the chip, in row strips: ONE hand-off between workgroups per iteration.
//
// Why.  The tile form (kernels_fitp.hip, rounds 1-5) keeps the upper triangle in registers; a tile (I, J) then feeds the sums
// of block I and of block J, so an iteration is two all-to-all exchanges through memory (T to the tiles, partial vectors to
// the reducers) at ~2 us each: 5.9 us per iteration, 42 iterations per alpha, 25 alphas per score -- the dominant kernel of
// the step, and none of it arithmetic.  The chip holds more than the triangle: 256 CUs x (512 KB of vector registers + 160 KB
// of LDS) = 168 MB against 4096^2 x 8 B = 134 MB.  With the FULL matrix resident, workgroup `sb` (one per CU, 512 threads)
// owns the 16 rows [16 sb, 16 sb + 16) over all columns, computes their sums from T alone and updates its own 16 entries of
// T: the only thing that crosses workgroups is T itself -- one exchange per iteration.
//
// Layout.  Thread t of a workgroup holds the columns t + 512 s, s = 0..7, of its strip's 16 rows: 128 doubles, rows 0..11 in
// registers (192 VGPRs), rows 12..15 in LDS (128 KB, [row][slot][thread]: conflict-free).  N <= 4096 vertices, ceil(N/16)
// workgroups (all co-resident: one per CU).
//
// An iteration k (every workgroup the same):
//   poll   T_k at the thread's 8 columns (each value of T is loaded once per workgroup) and, threads 0..nstrips-1, f_{k-1};
//   sums   acc[r] = sum_s g[r][s] T[s] per thread, combined over the 64 lanes by a transposing reduction on DPP moves
//          (8 rows -> one row per lane, then the two halves of a row of 16), over the 8 waves and the 4 lane rows through LDS:
//          fixed order, bitwise reproducible;
//   `while diff > delta` (:151) on max f_{k-1}, the same decision in every workgroup, so the whole grid leaves together;
//   update T_i += eps T_i (w_i / S_i - 1), f = max |w_i - S_i| (:160-166) by 16 threads; T_{k+1} and f_k are published.
// THE DATA IS ITS OWN SIGNAL (fit_flow.hpp): a slot that has not been delivered holds a sentinel, consumers poll the values
// they need; stores are write-through, loads bypass L1.  T lives in a ring of 4 vectors: a workgroup that publishes T_{k+1}
// has consumed all of T_k, so every workgroup has published T_k, so every workgroup has finished with T_{k-1} -- its slot is
// armed again for T_{k+3} (the arming store is drained by the `s_waitcnt vmcnt(0)` at the top of the next iteration, two
// publications before anybody polls that slot).  f: three buffers by k mod 3, armed the same way.  Every spin is bounded
// (1 s per hand-off -> `fail` -> the host redoes the alpha with one launch per iteration).
//
// FUSED (landmark mode, the sweep relabelled by community; wgcl_host.cpp decides): the rest of the alpha's chain rides on
// this launch.  Prologue: g = 2^(alpha log2(1 - D)) from the stored logarithm (pow_parts.hpp) -- the power matrix is never
// written.  Epilogue, from the strip and the final T still in registers: vect_B's products P_ij = (T_i T_j) g_ij, j >= i
// (:226-234), summed per (row run, column segment) -- a row run is a maximal run of one community inside the strip, a column
// segment one inside 64 aligned columns -- rows ascending, then columns ascending; bins_js_kernel (kernels_fit.hip) adds the
// <= 4 partials of a community pair.  And the first CGE_PARTIAL_BLOCKS workgroups tally the local score's sampled pairs
// (:178-213) exactly as auc_landmark_kernel's blocks do.
#include "common.hpp"
#include "pow_parts.hpp"
#include "fit_flow.hpp"

namespace {

#define ST_R 16   // rows of a strip: lane j of every row of 16 lanes owns row j
#define ST_T 512  // threads of a workgroup = 32 column groups x 16 rows
#define ST_NU 4   // column slots of a thread: it LOADS T at the two columns 1024 u + 2 tid + h, u < 4, h < 2 ...
#define ST_UR 3   // ... and HOLDS the matrix at row j, columns 1024 u + 32 G + 2 n + h, n < 16 (G = tid / 16): u < ST_UR in registers
#define ST_NF 256 // entries of an f buffer (strips at most)
#define ST_STAGE 4 // column segments per chunk the epilogue stages at a time

struct StripFused { // by value: what the prologue needs, and where the epilogue finds the rest
    const double *Lh; const float *Ll; double alpha;
    const cge_fit_fused *epi; // device copy of the sweep's table for this sample set
    int want;                 // bit 0: vect_B's partials, bit 1: the local score's tallies
};

constexpr int ST_GL = (ST_NU - ST_UR) * 32 * ST_T; // doubles of the strip's LDS part: gl[(n * 512 + tid) * 2 + h]
constexpr int ST_WORK = 4096;                      // the rest of the LDS (32 KB): the prologue's staging; then what follows
constexpr int ST_RED = 0;                          // red[parity][wave][lane]               1024
constexpr int ST_RED2 = 1024;                      // red2[parity][lane of wave 0]           128
constexpr int ST_FRED = 1152;                      // fred[parity][8]                         16
constexpr int ST_TROW = 1168;                      // T and w of the strip's rows             32
constexpr int ST_EXIT = 1200;                      // (int)
constexpr int ST_CS = 1208;                        // the epilogue's segment table (130 ints)
constexpr int ST_STG = 1280;                       // stg[ST_STAGE][512]                    2048
static_assert(ST_STG + ST_STAGE * ST_T <= ST_WORK, "LDS work region");
constexpr int ST_LDS_DOUBLES = ST_GL + ST_WORK;

typedef double st_d2 __attribute__((ext_vector_type(2)));
// acc += (lane N of the reader's row of 16 lanes of `bc`) * a as ONE instruction (gfx950's fp64 FMA takes a DPP broadcast on its first
// factor at the plain FMA's rate, profiles/microbench_dpp_fmac.hip); `bc` must have been written two instructions earlier
template <int N>
__device__ __forceinline__ void st_fmac_bcast(double &acc, double bc, double a) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(a), "n"(N));
}
// sum over n of bcast_n(t0) * e[n][0] + bcast_n(t1) * e[n][1] into two chains
template <int N0, int N1>
struct StripDot {
    static __device__ __forceinline__ void run(double &a0, double &a1, double t0, double t1, const double (&e)[16][2]) {
        st_fmac_bcast<N0>(a0, t0, e[N0][0]);
        st_fmac_bcast<N0>(a1, t1, e[N0][1]);
        StripDot<N0 + 1, N1>::run(a0, a1, t0, t1, e);
    }
};
template <int N1>
struct StripDot<N1, N1> {
    static __device__ __forceinline__ void run(double &, double &, double, double, const double (&)[16][2]) {}
};
// the staging of the prologue: element (row r, local column cl < 256) of a half block, 16-byte units swizzled by the row so that
// the writers (a row, consecutive columns) and the readers (16 rows, the same column) are both free of bank conflicts
__device__ __forceinline__ int st_stage_at(int r, int cl) { return r * 256 + ((((cl >> 1) ^ r)) << 1) + (cl & 1); }
// the sum of the first 256 threads' values, block_sum_256's additions (kernels_fit.hip); every thread of the workgroup calls it
__device__ __forceinline__ double strip_sum_256(double v, double *sh, int tid) {
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

template <bool FUSED>
__global__ __launch_bounds__(ST_T) void fit_strip_kernel(const double *__restrict__ GD, i64 N, const double *T0, double *Tout, i64 Tld,
                                                         const double *__restrict__ w, double eps, double delta, int max_iters,
                                                         double *ring, double *fq, unsigned *sync, int *flags,
                                                         long long timeout_ticks, int test_naps, const StripFused fz) {
    extern __shared__ __attribute__((aligned(16))) double st_lds[];
    double *const gl = st_lds;           // the matrix elements of slot u = 3
    double *const wk = st_lds + ST_GL;   // the work region
    double *const red = wk + ST_RED, *const red2 = wk + ST_RED2, *const fred = wk + ST_FRED, *const trow = wk + ST_TROW;
    int *const lds_exit = reinterpret_cast<int *>(wk + ST_EXIT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sb = blockIdx.x, nstrips = gridDim.x;
    const int jrow = tid & 15, G = tid >> 4;
    const unsigned Nu = (unsigned)N;
    const int Ue = (int)((N + 1023) >> 10); // column slots in use (uniform)
    unsigned *fail = sync + 1, *done = sync + 2;
#ifdef CGE_FLOW_CLOCK
    const long long ck0 = wall_clock64();
#endif
    if (FUSED && (fz.want & 2) && tid < 256) { // the local score's two powers per sample, ahead of everything (nothing is live yet)
        const cge_fit_fused *ep0 = fz.epi;
        const i64 S0 = ep0->S;
        for (int vb = sb; vb < CGE_PARTIAL_BLOCKS; vb += nstrips)
            for (i64 q = (i64)vb * 256 + tid; q < S0; q += (i64)CGE_PARTIAL_BLOCKS * 256) {
                ep0->apw[q] = pow(1.0 - ep0->dpos[q], fz.alpha);
                ep0->apw[S0 + q] = pow(1.0 - ep0->dneg[q], fz.alpha);
            }
    }
    if (test_naps > 0 && (sb & 1)) // testing (option fit_persistent_test_delay): every other strip starts late
        for (int q = 0; q < test_naps; q++) __builtin_amdgcn_s_sleep(127);

    // ---- the strip: this alpha's power from the stored logarithm (FUSED) or the caller's matrix.  Half blocks of 256 columns are
    // loaded with the lanes along the columns (coalesced), staged in LDS and picked up by the lanes that own them (a row each):
    // 8 of the 32 column groups per half block.  The loads of the next half block are in flight meanwhile. ----------------------------
    double e[ST_UR][16][2];
#pragma unroll
    for (int u = 0; u < ST_UR; u++)
#pragma unroll
        for (int n = 0; n < 16; n++) e[u][n][0] = e[u][n][1] = 0.0;
    {
        const int cl = tid & 255, r0 = tid >> 8; // this thread loads column cl of the half block, rows r0 + 2 i
        double lh[8];
        float ll[8];
        auto load_half = [&](int hb) {
            const unsigned col = 256u * (unsigned)hb + (unsigned)cl, colc = min(col, Nu - 1u);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned row = 16u * (unsigned)sb + (unsigned)(r0 + 2 * i), rowc = min(row, Nu - 1u);
                if (FUSED) { // unconditional (indices clamped into the matrix), the result selected afterwards
                    lh[i] = fz.Lh[(i64)rowc * N + colc];
                    ll[i] = fz.Ll[(i64)rowc * N + colc];
                } else
                    lh[i] = GD[(i64)rowc * N + colc];
            }
        };
        const int nhb = (int)((N + 255) >> 8); // half blocks in use (uniform)
        load_half(0);
#pragma unroll
        for (int hb = 0; hb < 4 * ST_NU; hb++) {
            if (hb >= nhb) { // uniform: beyond the matrix; the LDS part must still read as zeros
                if (hb >= 4 * ST_UR && (G >> 3) == (hb & 3)) {
#pragma unroll
                    for (int n = 0; n < 16; n++) *reinterpret_cast<st_d2 *>(&gl[(n * ST_T + tid) * 2]) = st_d2{0.0, 0.0};
                }
                continue;
            }
            const unsigned col = 256u * (unsigned)hb + (unsigned)cl;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned row = 16u * (unsigned)sb + (unsigned)(r0 + 2 * i);
                const bool in = row < Nu && col < Nu;
                const double v = FUSED ? exp2_parts(fz.alpha, lh[i], ll[i]) : lh[i];
                wk[st_stage_at(r0 + 2 * i, cl)] = in ? v : 0.0;
            }
            if (hb + 1 < nhb) load_half(hb + 1);
            __syncthreads();
            if ((G >> 3) == (hb & 3)) { // (uniform per wave: a wave is four consecutive groups)
                const int lc = 32 * (G & 7);
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const st_d2 v2 = *reinterpret_cast<const st_d2 *>(&wk[jrow * 256 + ((((lc >> 1) + n) ^ jrow) << 1)]);
                    if (hb < 4 * ST_UR) {
                        e[hb < 4 * ST_UR ? hb >> 2 : 0][n][0] = v2.x;
                        e[hb < 4 * ST_UR ? hb >> 2 : 0][n][1] = v2.y;
                    } else
                        *reinterpret_cast<st_d2 *>(&gl[(n * ST_T + tid) * 2]) = v2;
                }
            }
            __syncthreads();
        }
    }
#ifdef CGE_FLOW_CLOCK
    const long long ck1 = wall_clock64();
#endif
    // the current iterate and the target of the strip's rows live in LDS: trow[0..16) = T, trow[16..32) = w -- 16 threads use them
    if (tid < 16) {
        const unsigned myrow = 16u * (unsigned)sb + (unsigned)tid;
        const double t0 = myrow < Nu ? T0[myrow] : 0.0;
        trow[tid] = t0;
        trow[16 + tid] = myrow < Nu ? w[myrow] : 0.0;
        st_sc1_at(ring, myrow, t0); // T_0 goes through the ring like every other iterate: the loop has one form of poll
    }
    if (tid == 0) *lds_exit = 0;
    __syncthreads();

    int k = 0, converged = 0, failed = 0;
    double tc[ST_NU][2]; // T_k at the columns this thread loads (0 beyond the matrix)
#ifdef CGE_FLOW_CLOCK
    long long ca_poll = 0, ca_comp = 0, ca_bar = 0, ca_upd = 0, ca_rounds = 0;
#define SCK(acc) { const long long n_ = wall_clock64(); acc += n_ - ckt; ckt = n_; }
#else
#define SCK(acc)
#endif
    if (timeout_ticks <= 0) max_iters = 0; // test hook: abandon at once
    for (;;) {
        const long long deadline = wall_clock64() + timeout_ticks; // bounds one hand-off, not the whole fit
#ifdef CGE_FLOW_CLOCK
        long long ckt = wall_clock64();
#endif
        if (k >= max_iters) {
            failed = 1;
            if (tid == 0) __hip_atomic_store(fail, 1u, RLX_AGENT);
            break;
        }
        const double *Tk = ring + (i64)(k & 3) * 4096; // (T_0 too: every strip put its rows there before the loop)
        const double *fp = fq + (i64)((k + 2) % 3) * ST_NF; // f of iteration k - 1
        const int par = k & 1;
        // the arming stores of the previous iteration (and its T) have landed before anything of this iteration is stored
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int bad = 0; // wave-uniform: 2 = abandoned
        // ---- 1. f of iteration k - 1: what `while diff > delta` needs -- and the sign that T_k is out.  Read straight from memory, T
        // costs every workgroup 32 KB per iteration, 8 MB over the chip: 1.8 us at the 4.4 TB/s such loads get (measured: the poll
        // rounds of a first form of this kernel, which started asking for T at once, found most of it missing and asked again).  A
        // strip stores f right behind its rows of T, so the 2 KB of f are polled first (rounds of ~1 us, no bandwidth to speak of) and
        // T is asked for when all of f is there -- normally once.  (A hint only: every value of T is still checked for the sentinel.)
        if (k > 0) {
            if (wave < 4) { // (nstrips <= 256: one value per thread of waves 0..3)
                double fx = 0.0;
                const bool wantf = tid < nstrips;
                unsigned spins = 0;
                for (;;) {
                    if (wantf) fx = ld_sc1_at(fp, (unsigned)tid);
#ifdef CGE_FLOW_CLOCK
                    ca_rounds++;
#endif
                    if (!__any(wantf && armed(fx))) break;
                    bad = flow_check(spins, fail, done, deadline);
                    if (bad) break;
                }
                const double fv = wave_max(bad ? 0.0 : fx);
                if (lane == 0) fred[par * 8 + wave] = fv;
                if (bad && lane == 0) atomicOr(lds_exit, bad);
            }
            SCK(ca_poll)
            __syncthreads();
            SCK(ca_bar)
            if (*lds_exit) { failed = 1; break; } // uniform
            const double f = fmax(fmax(fred[par * 8 + 0], fred[par * 8 + 1]), fmax(fred[par * 8 + 2], fred[par * 8 + 3]));
            if (!(f > delta)) { converged = 1; break; } // T_k is final (the epilogue loads it); the same decision in every workgroup
        }
        // ---- 2. T_k and the strip's sums.  One 32-bit offset (the thread) and a wave-uniform base per slot, spelled out: the compiler
        // builds 64-bit vector addresses for these loads otherwise, and the strip leaves no registers for them.  16 bytes per lane and
        // load, unconditional -- beyond the matrix they stay inside the ring slot (4096 doubles whatever N) and are discarded.  The
        // slots are used as they arrive.  This lane's row over its 128 columns: T of column 32 G + 2 n + h sits in lane n of the row
        // of 16 -- a DPP broadcast on the FMA's first factor, no cross-lane reduction.  Eight chains (slot x parity of the column).
        double a[ST_NU][2];
#pragma unroll
        for (int u = 0; u < ST_NU; u++) a[u][0] = a[u][1] = 0.0;
        {
            const unsigned voff = (unsigned)tid << 4;
            const double *b0 = Tk, *b1 = Tk + 1024, *b2 = Tk + 2048, *b3 = Tk + 3072;
            const unsigned c0 = 2u * (unsigned)tid; // column of .x in slot 0
            st_d2 x0, x1, x2, x3;
            asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\t"
                         "global_load_dwordx4 %1, %4, %6 sc1\n\t"
                         "global_load_dwordx4 %2, %4, %7 sc1\n\t"
                         "global_load_dwordx4 %3, %4, %8 sc1"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                         : "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(b3)
                         : "memory");
            unsigned todo = (1u << Ue) - 1u; // slots whose sums are still to be made (uniform)
            unsigned spins = 0;
            bool first = true;
            for (;;) {
#define ST_SLOT(U, X, CNT)                                                                                                          \
                if (todo & (1u << U)) {                                                                                             \
                    if (first) asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(X) : : "memory");                                    \
                    const bool i0 = c0 + 1024u * U < Nu, i1 = c0 + 1024u * U + 1u < Nu;                                             \
                    if (!__any((i0 && armed(X.x)) || (i1 && armed(X.y)))) {                                                         \
                        tc[U][0] = i0 ? X.x : 0.0;                                                                                  \
                        tc[U][1] = i1 ? X.y : 0.0;                                                                                  \
                        asm volatile("s_nop 1" : "+v"(tc[U][0]), "+v"(tc[U][1])); /* VALU write -> DPP read: two wait states */      \
                        ST_DOT(U)                                                                                                   \
                        todo &= ~(1u << U);                                                                                         \
                    }                                                                                                               \
                }
#define ST_DOT(U) if (U < ST_UR) StripDot<0, 16>::run(a[U][0], a[U][1], tc[U][0], tc[U][1], e[U < ST_UR ? U : 0]); else ST_DOT_LDS(U)
#define ST_DOT_LDS(U) {                                                                                                             \
                        _Pragma("unroll") for (int hq = 0; hq < 2; hq++) { /* the LDS slot, half of it at a time */                  \
                            double le[16][2];                                                                                       \
                            _Pragma("unroll") for (int n = 0; n < 8; n++) {                                                         \
                                const st_d2 v2 = *reinterpret_cast<const st_d2 *>(&gl[((8 * hq + n) * ST_T + tid) * 2]);            \
                                le[8 * hq + n][0] = v2.x;                                                                           \
                                le[8 * hq + n][1] = v2.y;                                                                           \
                            }                                                                                                       \
                            if (hq == 0) StripDot<0, 8>::run(a[U][0], a[U][1], tc[U][0], tc[U][1], le);                             \
                            else StripDot<8, 16>::run(a[U][0], a[U][1], tc[U][0], tc[U][1], le);                                    \
                        }                                                                                                           \
                    }
                ST_SLOT(0, x0, 3) ST_SLOT(1, x1, 2) ST_SLOT(2, x2, 1) ST_SLOT(3, x3, 0)
#undef ST_SLOT
#undef ST_DOT
#undef ST_DOT_LDS
                if (first) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "memory");
                first = false;
#ifdef CGE_FLOW_CLOCK
                ca_rounds += 1000;
#endif
                if (!todo) break;
                bad = flow_check(spins, fail, done, deadline);
                if (bad) break;
                // later rounds ask only for the slots that still hold a sentinel somewhere in the wave
                if (todo & 1u) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "+v"(x0) : "v"(voff), "s"(b0) : "memory");
                if (todo & 2u) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "+v"(x1) : "v"(voff), "s"(b1) : "memory");
                if (todo & 4u) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "+v"(x2) : "v"(voff), "s"(b2) : "memory");
                if (todo & 8u) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "+v"(x3) : "v"(voff), "s"(b3) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "memory");
            }
        }
        const double v = bad ? 0.0 : ((a[0][0] + a[0][1]) + (a[1][0] + a[1][1])) + ((a[2][0] + a[2][1]) + (a[3][0] + a[3][1]));
        if (bad && lane == 0) atomicOr(lds_exit, bad);
        red[(par * 8 + wave) * 64 + lane] = v; // row (lane & 15), column group 4 wave + lane / 16
        SCK(ca_comp)
        __syncthreads(); // red alternates by the parity of k
        SCK(ca_bar)
        if (*lds_exit) { failed = 1; break; } // uniform
        if (wave == 0) {
            // (a thread id the compiler cannot see through: the addresses of this block are computed here, from one register, instead
            // of being kept across the loop in registers the strip does not leave)
            int tu = tid;
            asm volatile("" : "+v"(tu));
            double x = red[(par * 8 + 0) * 64 + tu];
#pragma unroll
            for (int w2 = 1; w2 < 8; w2++) x += red[(par * 8 + w2) * 64 + tu];
            red2[par * 64 + tu] = x;
            __builtin_amdgcn_wave_barrier();
            if (tu < 16) {
                const unsigned urow = 16u * (unsigned)sb + (unsigned)tu;
                double S = ((red2[par * 64 + tu] + red2[par * 64 + tu + 16]) + red2[par * 64 + tu + 32]) + red2[par * 64 + tu + 48];
                double fr = 0.0, tnew = 0.0;
                const double tcur = trow[tu], wrow = trow[16 + tu];
                if (urow < Nu) {
                    S *= tcur; // S_i = T_i * sum_j g_ij T_j
                    tnew = tcur + (eps * tcur) * (wrow / S - 1.0);
                    fr = fabs(wrow - S);
                }
                unsigned long long sbits = FLOW_SENTINEL;
                asm volatile("" : "+s"(sbits));
                const double sent = __longlong_as_double((long long)sbits);
                st_sc1_at(ring + (i64)((k + 1) & 3) * 4096, urow, tnew);
                trow[tu] = tnew;
                fr = row16_max(fr);
                if (tu == 0) st_sc1_at(fq + (i64)(k % 3) * ST_NF, (unsigned)sb, fr);
                st_sc1_at(ring + (i64)((k + 3) & 3) * 4096, urow, sent);
                if (tu == 0) st_sc1_at(fq + (i64)((k + 1) % 3) * ST_NF, (unsigned)sb, sent);
            }
        }
        SCK(ca_upd)
        k++;
    }
    if (converged && tid < 16 && 16u * (unsigned)sb + (unsigned)tid < Nu) Tout[16 * sb + tid] = trow[tid]; // T_k: every strip holds its rows
    if (sb == 0 && tid == 0) {
        flags[0] = converged;
        flags[1] = k; // iterations done: T_k is final
        flags[2] = failed || !converged;
        flags[3] = 0;
    }
#ifdef CGE_FLOW_CLOCK
    const long long ck2 = wall_clock64();
#endif
    if (!FUSED) return;
    if (!converged) return; // (uniform over the grid: an abandoned fit computes nothing)
    // ---- the rest of the alpha's chain, from the strip and the final iterate (tc = T_k at the columns this thread loaded) ---------
    const cge_fit_fused *ep = fz.epi;
    int te = tid; // (a thread id the compiler cannot see through: nothing of the epilogue is computed ahead of the loop and kept across it)
    asm volatile("" : "+v"(te));
    if (fz.want & 1) {
        // TODO v2
    }
#ifdef CGE_FLOW_CLOCK
    const long long ck3 = wall_clock64();
#endif
    if (fz.want & 2) {
        double *sh = wk + ST_STG; // >= 256 doubles
        const double *Tf = ring + (i64)(k & 3) * 4096; // T_k, complete and not armed again: the converging iteration published nothing
        __syncthreads();
        for (int vb = sb; vb < CGE_PARTIAL_BLOCKS; vb += nstrips) { // uniform
            double num = 0.0;
            const i64 S = ep->S;
            if (te < 256)
                for (i64 q = (i64)vb * 256 + te; q < S; q += (i64)CGE_PARTIAL_BLOCKS * 256) {
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
            num = strip_sum_256(num, sh, te);
            if (te == 0) { ep->auc_part[2 * vb] = num; ep->auc_part[2 * vb + 1] = ep->aden[vb]; }
        }
    }
#ifdef CGE_FLOW_CLOCK
    if ((tid == 0 || tid == 320) && (sb == 0 || sb == 130))
        printf("strip clock wg %d wave %d: prologue %lld  loop %lld (%d iterations: poll %lld in %lld rounds, sums %lld, barrier %lld, update %lld)  vect_B epilogue %lld  tallies %lld  (10 ns ticks)\n", sb, tid >> 6,
               ck1 - ck0, ck2 - ck1, k, ca_poll, ca_rounds, ca_comp, ca_bar, ca_upd, ck3 - ck2, wall_clock64() - ck3);
#endif
}

bool strip_geometry(i64 N, i64 Tld, int *G_out) {
    if (N < 1 || N > (i64)ST_R * ST_NF || N > (i64)ST_NU * 1024) return false;
    const int G = (int)((N + ST_R - 1) / ST_R);
    int dev = 0, cus = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (G > cus || Tld < (i64)G * ST_R) return false;
    *G_out = G;
    return true;
}

} // namespace

bool k_fit_strip_applies(cge_ctx *c, i64 N) {
    int G = 0;
    return strip_geometry(N, (N + 63) / 64 * 64, &G);
}
// where the hand-off slots of fit_strip_kernel live and how they are armed (the buffer is made if need be)
bool k_fit_strip_arm_region(cge_ctx *c, i64 N, i64 Tld, uint4 **ptr, i64 *n16, unsigned *word) {
    int G = 0;
    if (!strip_geometry(N, Tld, &G)) return false;
    const size_t n_sync = 32, n_ring = (size_t)4 * 4096, n_fq = (size_t)3 * ST_NF;
    const size_t doubles = n_sync + n_ring + n_fq;
    if (doubles % 2) return false; // (16-byte units)
    c->fp_flow.ensure(doubles);
    *ptr = reinterpret_cast<uint4 *>(c->fp_flow.p);
    *n16 = (i64)(doubles / 2);
    *word = FLOW_SENTINEL_WORD;
    return true;
}
bool k_fit_strip_enqueue(cge_ctx *c, const double *GD, i64 N, const double *T0, double *Tout, i64 Tld, const double *w, double eps,
                         double delta, int *dev_flags, const cge_fit_fused *ff, const cge_fit_fused *ff_dev) {
    const bool fused = ff != nullptr;
    StripFused fz{};
    if (fused) {
        fz.Lh = ff->Lh; fz.Ll = ff->Ll; fz.alpha = ff->alpha;
        fz.epi = ff_dev;
        fz.want = (ff->partial ? 1 : 0) | (ff->auc_part ? 2 : 0);
    }
    int G = 0;
    if (!strip_geometry(N, Tld, &G)) return false;
    const void *fn = fused ? (const void *)fit_strip_kernel<true> : (const void *)fit_strip_kernel<false>;
    const size_t lds_bytes = ST_LDS_DOUBLES * sizeof(double);
    static bool attr_set[2] = {false, false};
    if (!attr_set[fused ? 1 : 0]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[fused ? 1 : 0] = true;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, ST_T, lds_bytes) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return false;
    }
    const size_t n_sync = 32, n_ring = (size_t)4 * 4096, n_fq = (size_t)3 * ST_NF;
    c->fp_flow.ensure(n_sync + n_ring + n_fq);
    hipStream_t st = c->stream;
    const i64 arm_words = (i64)(2 * (n_sync + n_ring + n_fq));
    if (c->flow_armed_words != arm_words) // (else: armed by the last launch of the previous alpha's chain, bins_js_kernel)
        HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)c->fp_flow.p, (int)FLOW_SENTINEL_WORD, (size_t)arm_words, st));
    c->flow_armed_words = 0;
    const double *aGD = GD, *aT0 = T0, *aW = w;
    double *aTout = Tout, *aRing = c->fp_flow.p + n_sync, *aFq = aRing + n_ring;
    i64 aN = N, aTld = Tld;
    int aMax = 2000000;
    double aEps = eps, aDelta = delta;
    unsigned *aSync = (unsigned *)c->fp_flow.p;
    int *aFlags = dev_flags;
    long long aTicks = c->opt_fit_test_timeout ? 0LL : CGE_FIT_TIMEOUT_TICKS; // per iteration (0: the test hook)
    int aNaps = c->opt_fit_test_delay;
    void *args[] = {&aGD, &aN, &aT0, &aTout, &aTld, &aW, &aEps, &aDelta, &aMax, &aRing, &aFq, &aSync, &aFlags, &aTicks, &aNaps, &fz};
    hipError_t e;
    {
        ScopedKernelTimer tm(c, "fit_persistent");
        e = hipLaunchKernel(fn, dim3((unsigned)G), dim3(ST_T), args, lds_bytes, st);
    }
    if (e != hipSuccess) CGE_THROW(CGE_E_HIP, "fit launch failed: %s", hipGetErrorString(e));
    return true;
}
