// Does a DPP broadcast ride free on an fp64 FMA on gfx950?  v_fmac_f64_dpp ... row_newbcast:N takes its first factor from lane N of
// the reader's row of 16 lanes -- the shape of the eigen-solver's rank-2 update (a[r][j] -= u[r] * w_j, w_j the same for all
// lanes), which today reads every w_j as an LDS broadcast.  Measures, per wave-instruction: plain v_fma_f64 on registers, the
// DPP form, and the LDS-broadcast + FMA pair; and checks the DPP semantics.
// Build + run: hipcc --offload-arch=gfx950 -O3 profiles/microbench_dpp_fmac.hip -o /tmp/mbdpp && /tmp/mbdpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NACC 16
template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(const double *__restrict__ in, double *__restrict__ out, int iters) {
    __shared__ double sh[64];
    const int lane = threadIdx.x & 63;
    double acc[NACC], a[NACC];
    const double u = in[lane];
    if (threadIdx.x < 64) sh[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NACC; k++) { acc[k] = 0.0; a[k] = in[64 + lane] + k; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) {
            if (MODE == 0) {
                acc[k] = fma(u, a[k], acc[k]);
                asm volatile("" : "+v"(acc[k]));
            } else if (MODE == 1) {
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[k]) : "v"(u), "v"(a[k]));
            } else {
                double b;
                asm volatile("ds_read_b64 %0, %1" : "=v"(b) : "v"((unsigned)(8 * ((k + it) & 63))) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc[k] = fma(b, a[k], acc[k]);
                asm volatile("" : "+v"(acc[k]));
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NACC; k++) s += acc[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void semantics_kernel(const double *__restrict__ in, double *__restrict__ out) {
    const int lane = threadIdx.x;
    double acc = 0.0, one = 1.0;
    const double u = in[lane];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(one));
    out[lane] = acc; // expected: in[16 * (lane / 16) + 5]
}
template <int MODE>
double run(const double *in, double *out, int wg_per_cu) {
    const int iters = 20000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: wg_per_cu waves per SIMD x iters x NACC
    return (double)ms * 1e6 / ((double)wg_per_cu * iters * NACC); // ns per wave-instruction per SIMD
}
int main() {
    double *in, *out;
    (void)hipMalloc(&in, 128 * 8);
    (void)hipMalloc(&out, 256 * 8 * 256 * 8);
    std::vector<double> h(128);
    for (int i = 0; i < 128; i++) h[i] = 1.0 + i * 0.001;
    (void)hipMemcpy(in, h.data(), 128 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(semantics_kernel, dim3(1), dim3(64), 0, 0, in, out);
    std::vector<double> o(64);
    (void)hipMemcpy(o.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) bad += o[l] != h[16 * (l / 16) + 5];
    printf("row_newbcast:5 semantics: %s (lane 0 -> %.3f, lane 17 -> %.3f, lane 63 -> %.3f)\n", bad ? "UNEXPECTED" : "lane 5 of the own row", o[0], o[17], o[63]);
    for (int w = 1; w <= 4; w *= 2) {
        printf("waves/SIMD %d: v_fma_f64 %.2f ns   v_fmac_f64_dpp row_newbcast %.2f ns   ds_read_b64 broadcast + v_fma_f64 %.2f ns  (per wave-instruction per SIMD)\n",
               w, run<0>(in, out, w), run<1>(in, out, w), run<2>(in, out, w));
    }
    return 0;
}
