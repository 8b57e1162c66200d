/* cge_hip_testing.h -- host-only hooks of libcge_hip.so used by the CPU test-suite (no GPU needed).
 * They expose the product's own host routines (NOT the oracle's) so that `-m "not gpu"` tests can
 * check them against the oracle. */
#ifndef CGE_HIP_TESTING_H
#define CGE_HIP_TESTING_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* principal eigenvector of a symmetric d x d matrix (replaces eigvecs(A)[:, end], src/landmarks.jl:99) */
int cge_host_eig_top(const double *A, int64_t d, double *v);
/* the sampler's counter-based draw of positive rows: pos_idx[k] in 1..m (no device work) */
int cge_host_pos_draw(int64_t seed, int64_t stream_id, int64_t S, int64_t m, int64_t *pos_idx);
/* kernel-level hook (needs the GPU): principal eigenvectors of T symmetric d x d matrices (row-major, back to
 * back) by the batched device solvers that landmarks uses (d <= 128: register-resident; d <= 512: global-memory
 * resident); returns CGE_E_ARG for d > 512 */
int cge_group_eig(void *ctx, const double *A, int64_t T, int64_t d, double *v);
/* out[i] = (1 - x[i])^alpha on the device (host vectors): method 0 = the library pow, 1 = exp2(alpha * log2(1 - x)) with
 * the logarithm in double + float parts, as the alpha sweep computes GD = (1 - D)^alpha (src/divergence.jl:142-148) */
int cge_pow_test(void *ctx, const double *x, int64_t n, double alpha, int method, double *out);
/* needs the GPU and a communicator (cge_comm_init_rccl; one rank is enough): host array -> device -> the in-library
 * ncclAllReduce (op 0 sum / 1 max of doubles, 2 sum of the words as int64) -> host */
int cge_rccl_selftest(void *ctx, double *host_inout, int64_t count, int op);
/* kernel-level hook (needs the GPU): the per-group ascending STABLE sort of the projections as runsplit calls it (LDS pieces
 * + rank merge for long groups, rocPRIM otherwise): group t = rows [task_row_off[t], task_row_off[t+1]) of z; zs_out = sorted
 * values, perm_out[j] = index INSIDE its group of the element at sorted position j.  Order: ascending, -0.0 == 0.0, ties by
 * index (a group with NaNs is routed to the host path by its status) */
int cge_segment_sort_test(void *ctx, const double *z, const int32_t *task_row_off, int64_t T, double *zs_out, int32_t *perm_out);
/* kernel-level hook (needs the GPU): per row of 64 doubles, lane 0's sum by the shuffle tree (out_ref) and by the gfx950
 * lane swaps the projection kernel uses instead (out_new): the same pairs in the same order, so the same bits */
int cge_wave_tree_test(void *ctx, const double *x, int64_t n_rows, double *out_ref, double *out_new);
/* testing knobs of a context (needs the GPU; results must not depend on them):
 *   "fit_persistent_test_delay"   n: the tile waves of the persistent fits nap n x ~3 us before their first load (start skew, as
 *                                 under contention);
 *   "fit_persistent_test_timeout" 1: the persistent fit abandons every launch at once (the fallback path runs);
 *   "test_bvec_plain"             1: vect_B by the kernels of score graphs beyond the LDS budget / 512 communities.              */
int cge_set_test_option(void *ctx, const char *key, int64_t value);
#ifdef __cplusplus
}
#endif
#endif
