// What a launch of the landmark phase can hope for from HBM: (a) streaming reads, (b) gathers of 1 KB rows (one d = 128 fp64
// row per wave-load) through an index, at the volume of one runsplit batch (~600 MB) and at 4 GB.  Reports GB/s per launch.
// Build + run: hipcc --offload-arch=gfx950 -O3 profiles/microbench_hbm_rows.hip -o /tmp/mbrows && /tmp/mbrows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
typedef double d2 __attribute__((ext_vector_type(2)));
// one wave per RF rows of 128 doubles (16 bytes per lane), rows taken through idx (nullptr: consecutive)
template <int RF>
__global__ __launch_bounds__(256) void rows_kernel(const double *__restrict__ X, const int *__restrict__ idx, long long n_rows,
                                                   double *__restrict__ out) {
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x & 63;
    const long long j0 = w * RF;
    if (j0 >= n_rows) return;
    d2 v[RF];
#pragma unroll
    for (int u = 0; u < RF; u++) {
        const long long j = j0 + u < n_rows ? j0 + u : j0;
        const long long r = idx ? idx[j] : j;
        v[u] = *(const d2 *)(X + r * 128 + 2 * lane);
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < RF; u++) s += v[u][0] + v[u][1];
    if (s == 1.2345e300) out[0] = s; // never true: keeps the loads
}
template <int RF>
float time_rows(const double *X, const int *idx, long long n_rows, double *out, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long long waves = (n_rows + RF - 1) / RF;
    const unsigned grid = (unsigned)((waves * 64 + 255) / 256);
    hipLaunchKernelGGL(rows_kernel<RF>, dim3(grid), dim3(256), 0, 0, X, idx, n_rows, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rows_kernel<RF>, dim3(grid), dim3(256), 0, 0, X, idx, n_rows, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const long long n_big = 4LL << 20; // 4 Mi rows x 1 KB = 4 GB
    double *X, *out;
    int *idx;
    if (hipMalloc(&X, n_big * 1024) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&idx, n_big * 4);
    (void)hipMemset(X, 0, n_big * 1024);
    std::mt19937_64 rng(1);
    for (long long n_tab : {1LL << 20, 4LL << 20}) {            // rows of the table the gather draws from (1 GB / 4 GB)
        for (long long n_rows : {600000LL, 4LL << 20}) {        // rows one launch reads
            if (n_rows > n_tab && n_rows != 600000) continue;
            std::vector<int> h(n_rows);
            std::vector<int> perm(n_tab);
            for (long long i = 0; i < n_tab; i++) perm[i] = (int)i;
            std::shuffle(perm.begin(), perm.end(), rng);
            for (long long i = 0; i < n_rows; i++) h[i] = perm[i % n_tab];
            (void)hipMemcpy(idx, h.data(), n_rows * 4, hipMemcpyHostToDevice);
            const double gb = n_rows * 1024.0 / 1e9;
            float t;
            t = time_rows<1>(X, nullptr, n_rows, out, 20); printf("table %lld rows, launch %lld rows: stream  RF1 %.4f ms %7.0f GB/s\n", n_tab, n_rows, t, gb / t * 1e3);
            t = time_rows<4>(X, nullptr, n_rows, out, 20); printf("table %lld rows, launch %lld rows: stream  RF4 %.4f ms %7.0f GB/s\n", n_tab, n_rows, t, gb / t * 1e3);
            t = time_rows<1>(X, idx, n_rows, out, 20);     printf("table %lld rows, launch %lld rows: gather  RF1 %.4f ms %7.0f GB/s\n", n_tab, n_rows, t, gb / t * 1e3);
            t = time_rows<4>(X, idx, n_rows, out, 20);     printf("table %lld rows, launch %lld rows: gather  RF4 %.4f ms %7.0f GB/s\n", n_tab, n_rows, t, gb / t * 1e3);
            t = time_rows<8>(X, idx, n_rows, out, 20);     printf("table %lld rows, launch %lld rows: gather  RF8 %.4f ms %7.0f GB/s\n", n_tab, n_rows, t, gb / t * 1e3);
            // sorted index (ascending rows with gaps): what a batch's row list looks like inside a community
            std::sort(h.begin(), h.end());
            (void)hipMemcpy(idx, h.data(), n_rows * 4, hipMemcpyHostToDevice);
            t = time_rows<4>(X, idx, n_rows, out, 20);     printf("table %lld rows, launch %lld rows: sorted  RF4 %.4f ms %7.0f GB/s\n", n_tab, n_rows, t, gb / t * 1e3);
        }
    }
    return 0;
}
