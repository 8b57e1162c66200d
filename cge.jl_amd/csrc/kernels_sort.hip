// kernels_sort.hip -- per-group ascending sort of the projections z on the device (rss rule on sorted
// order, src/landmarks.jl:155-210).  rocPRIM's segmented radix sort is stable, so equal z keep their
// original (ascending index) order -- the same order a stable sortperm gives.
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

__global__ void iota_local_kernel(const i32 *__restrict__ row_task, const i32 *__restrict__ task_row_off, i64 R,
                                  i32 *__restrict__ idx) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < R) idx[j] = (i32)(j - task_row_off[row_task[j]]);
}
// srows[j] = vertex at sorted rank j of its task
__global__ void sorted_gather_kernel(const i32 *__restrict__ rows, const i32 *__restrict__ row_task,
                                     const i32 *__restrict__ task_row_off, const i32 *__restrict__ perm, i64 R,
                                     i32 *__restrict__ srows) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < R) srows[j] = rows[task_row_off[row_task[j]] + perm[j]];
}
// status: 0 = sorted-order path applies, 1 = generic path (NaN, or a tie at the maximum: the arg-max of the
// reference is then not the last rank), 2 = homogeneous (argmin == argmax, src/landmarks.jl:165-167)
__global__ void sort_status_kernel(const double *__restrict__ zs, const i32 *__restrict__ task_row_off, i64 T,
                                   i32 *__restrict__ status) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const i64 o = task_row_off[t], k = task_row_off[t + 1] - o;
    const double a = zs[o], b = zs[o + k - 1], b2 = zs[o + k - 2];
    int st = 0;
    if (a != a || b != b) st = 1; // radix order puts NaNs at the two ends
    else if (a == b) st = 2;
    else if (b == b2) st = 1;
    status[t] = st;
}

void k_segmented_sort_z(cge_ctx *c, const double *z, const i32 *rows, const i32 *row_task, const i32 *task_row_off,
                        i64 R, i64 T, double *zs, i32 *perm, i32 *srows, i32 *status) {
    ScopedKernelTimer tm(c, "segmented_sort");
    c->sort_idx.ensure(R);
    const unsigned nb = (unsigned)((R + 255) / 256);
    hipLaunchKernelGGL(iota_local_kernel, dim3(nb), dim3(256), 0, c->stream, row_task, task_row_off, R, c->sort_idx.p);
    size_t bytes = 0;
    HIP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, bytes, z, zs, c->sort_idx.p, perm, (unsigned)R, (unsigned)T,
                                                  task_row_off, task_row_off + 1, 0, 64, c->stream));
    c->sort_tmp.ensure(bytes);
    HIP_CHECK(rocprim::segmented_radix_sort_pairs(c->sort_tmp.p, bytes, z, zs, c->sort_idx.p, perm, (unsigned)R,
                                                  (unsigned)T, task_row_off, task_row_off + 1, 0, 64, c->stream));
    hipLaunchKernelGGL(sorted_gather_kernel, dim3(nb), dim3(256), 0, c->stream, rows, row_task, task_row_off, perm, R, srows);
    hipLaunchKernelGGL(sort_status_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, c->stream, zs, task_row_off, T,
                       status);
}
