// fit_flow.hpp -- what the persistent Chung-Lu fits share (kernels_fitp.hip: the directed fit over register-resident upper tiles;
// kernels_fits.hip: the undirected fit over register-resident row strips): the sentinel hand-off primitives -- THE DATA IS ITS
// OWN SIGNAL, see the header of kernels_fits.hip -- and small DPP helpers.
#pragma once
#include "common.hpp"

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

} // namespace
