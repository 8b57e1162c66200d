// Peak-rate microbenchmark for v_mfma_f64_16x16x4_f64 on gfx950: back-to-back MFMAs on register operands,
// NACC independent accumulators per wave, 1 / 2 / 4 waves per SIMD, every CU busy.  Reports TFLOP/s.
// Build + run: hipcc --offload-arch=gfx950 -O3 profiles/microbench_f64_mfma.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = (d4){0.0, 0.0, 0.0, 0.0};
    double a[4], b[4];
    for (int q = 0; q < 4; q++) { a[q] = a0 + threadIdx.x * 1e-3 + q; b[q] = b0 - threadIdx.x * 1e-3 - q; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k & 3], b[(k >> 2) & 3], acc[k], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NACC; k++) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(double *out, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 160000 / NACC;
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        const int grid = 256 * wg_per_cu;
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0, 2.0); // warm-up
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 2.0);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)grid * 4 * iters * NACC;
        printf("acc %2d waves/SIMD %d: %6.2f ms  %6.2f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", NACC, wg_per_cu, ms,
               n_mfma * 2048.0 / (ms * 1e-3) / 1e12, (ms * 1e6) / (n_mfma / 1024.0));
    }
}
int main() {
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 256 * 8 * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    run<4>(out, e0, e1);
    run<8>(out, e0, e1);
    run<16>(out, e0, e1);
    run<32>(out, e0, e1);
    return 0;
}
