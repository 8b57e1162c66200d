/*
 * cge_hip.h -- C-ABI of libcge_hip.so: the MI355X (gfx950) implementation of CGE.jl's
 * divergence-scoring hot path.  This is the boundary a Julia `ccall`, a Python `ctypes`
 * or a C/C++ caller binds to.  Plain pointers and sizes only; no C++/torch types.
 *
 * What each entry point replaces in the reference (KrainskiL/CGE.jl v2.0.2):
 *   cge_landmarks_*      landmarks()        src/landmarks.jl:365-466  (+ runsplit :279-345, split rules :92-267)
 *   cge_wgcl             wGCL()             src/divergence.jl:27-257
 *   cge_wgcl (directed)  wGCL_directed()    src/divergence.jl:282-561
 *   cge_score            example/CGE_CLI.jl:4-24 (landmarks() followed by wGCL*() on device-resident data)
 *   cge_draw_samples     the `sample(E,..)` / `sample(NE,..)` draws, src/divergence.jl:184-210
 *   cge_js / cge_idx     JS() / idx()       src/auxilary.jl:34-52, :57-59
 *
 * Conventions (exactly the reference's, so a Julia Array can be passed as is):
 *   - vertex / community / landmark ids are 1-based int64;
 *   - matrices are column-major: edges (m x 2) = two contiguous int64 columns `src`,`dst`;
 *     an embedding (n x d) = d contiguous columns of n doubles;
 *   - host pointers are BORROWED for the duration of the call, never retained, never written
 *     (outputs excepted); outputs are caller-allocated;
 *   - every function returns an int status (0 = ok, <0 = error; cge_last_error() has the text);
 *     nothing throws or exits across the boundary;
 *   - a cge_ctx is bound to ONE GPU and is not re-entrant: one host thread drives it (cge_score
 *     itself starts a second host thread on a shadow context of its own; nothing of that crosses
 *     the boundary).  Multi-GPU runs are one process (one ctx) per GPU; the cross-GPU exchange
 *     steps are issued by the library itself on an RCCL communicator (cge_rccl_unique_id +
 *     cge_comm_init_rccl: ncclAllReduce on the ctx stream), or, where no RCCL communicator can be
 *     made (gloo tests, two ranks on one GPU), go through the cge_collectives hook.
 *   - there is NO CPU fallback: without a gfx950 device cge_create fails with CGE_E_HIP.
 */
#ifndef CGE_HIP_H
#define CGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGE_ABI_VERSION 1

/* status codes; the Julia wrapper maps them back to the reference's exceptions */
#define CGE_OK 0
#define CGE_E_ASSERT (-1)        /* an @assert of the reference (src/divergence.jl:50,81,303,363; src/landmarks.jl:93,...) */
#define CGE_E_HOMOGENEOUS (-2)   /* ErrorException("Trying to split homogenous cluster")  src/landmarks.jl:166 */
#define CGE_E_EMPTY_CLUSTER (-3) /* ErrorException("Unexpected empty cluster generated")  src/landmarks.jl:298,306,325,333 */
#define CGE_E_HIP (-4)           /* HIP runtime error / no device */
#define CGE_E_COLLECTIVE (-5)    /* a collective hook failed */
#define CGE_E_OOM (-6)
#define CGE_E_ARG (-7)           /* bad argument at the boundary */

/* split rule enum: the reference passes the function itself (src/auxilary.jl:64-67) */
#define CGE_METHOD_RSS 0
#define CGE_METHOD_RSS2 1
#define CGE_METHOD_SIZE 2
#define CGE_METHOD_DIAMETER 3

typedef struct cge_ctx cge_ctx;

/* ---- context ------------------------------------------------------------------------------ */
/* `stream`: a hipStream_t to launch on (NULL = the library creates its own).  Passing the host
 * framework's stream keeps its events/collectives ordered with the library's kernels.          */
int cge_create(cge_ctx **out, int device, void *stream);
void cge_destroy(cge_ctx *ctx);
const char *cge_last_error(const cge_ctx *ctx);
int cge_abi_version(void);
int cge_set_host_threads(cge_ctx *ctx, int n_threads); /* host worker pool for eigenvectors / cut rules */

/* Cross-GPU hooks (optional).  `buf` is a DEVICE pointer into the ctx's exchange buffer (obtain it
 * with cge_exchange_buffer and wrap it once on the host side); count is in doubles.  The hook must
 * order itself after work already enqueued on the ctx stream and leave the reduced data in place
 * (stream-ordered or synchronous).  op: 0 = sum, 1 = max, 2 = sum of the 8-byte words taken as
 * int64 (an exact gather of disjoint shards written into zero-filled buffers).                       */
typedef struct {
    int (*allreduce_f64)(void *user, void *buf, int64_t count, int op);
    void *user;
    int rank, world;
} cge_collectives;
int cge_set_collectives(cge_ctx *ctx, const cge_collectives *coll);
/* Further operations of the hook (optional; set AFTER cge_set_collectives, cleared by it): where they are present the library
 * moves gathers as gathers and the N x N landmark-pair matrix by row blocks instead of all-reducing whole buffers (what
 * ncclAllGather / ncclReduceScatter do on the in-library communicator).  Both work in place on `buf` (a device pointer into
 * the exchange buffer), `words` 8-byte words PER RANK, world * words in all:
 *   allgather(user, buf, words):          rank r contributes block [r * words, (r + 1) * words); afterwards every rank holds all;
 *   reduce_scatter_f64(user, buf, words): afterwards block r of rank r holds the sums (doubles) of that block over the ranks;
 *                                         the other blocks are unspecified.                                                  */
typedef struct {
    int (*allgather)(void *user, void *buf, int64_t words_per_rank);
    int (*reduce_scatter_f64)(void *user, void *buf, int64_t words_per_rank);
} cge_collectives_ext;
int cge_set_collectives_ext(cge_ctx *ctx, const cge_collectives_ext *ext);
int cge_exchange_buffer(cge_ctx *ctx, int64_t min_doubles, void **dev_ptr, int64_t *capacity_doubles);
/* or hand the library a caller-owned device buffer (e.g. a torch tensor) to use as exchange buffer */
int cge_set_exchange_buffer(cge_ctx *ctx, void *dev_ptr, int64_t capacity_doubles);

/* In-library collectives: RCCL over xGMI, one rank per GPU (csrc/collectives.cpp; librccl is opened at run time).  Rank 0
 * calls cge_rccl_unique_id and hands the CGE_RCCL_ID_BYTES bytes to every rank by whatever means the host has (MPI, a file,
 * torch.distributed); every rank then calls cge_comm_init_rccl on its context (collective: all ranks must call it).  From
 * then on the exchange steps of the path are ncclAllReduce calls issued by the library on the ctx stream -- no callback, no
 * host synchronisation per step, no exchange buffer to provide.  Takes precedence over a hook set with
 * cge_set_collectives.  cge_comm_finalize (also done by cge_destroy) releases the communicator.                       */
#define CGE_RCCL_ID_BYTES 128
int cge_rccl_unique_id(void *id_out);
int cge_comm_init_rccl(cge_ctx *ctx, const void *id, int rank, int world);
int cge_comm_finalize(cge_ctx *ctx);

/* ---- resident inputs (the ORIGINAL graph; uploaded once, H2D + layout change) ---------------
 * Calling cge_set_graph with another vertex count drops the resident embedding and vertex data (they were sized for the
 * old vertex set); cge_landmarks_run / cge_score check that all resident inputs agree and fail with CGE_E_ARG otherwise.
 * cge_wgcl in exact mode (empty v_to_l) makes ITS graph the resident one (the sampler rejects against it).
 * The ids are validated while they stream into the (already re-allocated) device arrays: a call that fails on a vertex id
 * outside 1..n leaves NO resident graph (the previous one is released, landmark state is invalidated); upload again.   */
/* src/dst: the two columns of the reference's `edges::Matrix{Int}` (src/auxilary.jl:106); w: eweights */
int cge_set_graph(cge_ctx *ctx, const int64_t *src, const int64_t *dst, const double *w, int64_t m, int64_t n);
/* embedding::Matrix{Float64} n x d column-major (src/auxilary.jl:164) */
int cge_set_embedding(cge_ctx *ctx, const double *X_colmajor, int64_t n, int64_t d);
/* The same for an embedding that already lives in THIS GPU's memory (a host framework's tensor; 41 GB at configuration 5
 * would otherwise cross PCIe): n x d doubles, row-major (row_major = 1: a vertex's d features contiguous) or column-major
 * like Julia's Matrix (0).  Copied (the caller keeps ownership and may free the buffer on return).                      */
int cge_set_embedding_device(cge_ctx *ctx, const double *X_dev, int64_t n, int64_t d, int row_major);
/* comm::Matrix{Int} n x 1 (src/auxilary.jl:122-139) and vweight (src/auxilary.jl:104-110) */
int cge_set_vertex_data(cge_ctx *ctx, const int64_t *comm, const double *vweights, int64_t n);

/* ---- landmarks(): src/landmarks.jl:365-466 --------------------------------------------------- */
/* clusters::Vector{Vector{Int}} as CSR: members clusters_flat[off[c] .. off[c+1]) (1-based ids).
 * Runs on the resident inputs; results stay on the device (for cge_wgcl / cge_score) and can be
 * fetched.  `*N_out` = number of landmarks, `*n_ledges_out` = rows of landmark_edges (w > 0),
 * `*truncated` = 1 when the reference's @warn at :374 would have fired.                          */
int cge_landmarks_run(cge_ctx *ctx, const int64_t *clusters_flat, const int64_t *clusters_off, int64_t n_clusters,
                      int64_t land, int64_t forced, int method, int directed, int64_t *N_out,
                      int64_t *n_ledges_out, int *truncated);
/* sizes of the landmark state produced by the last cge_landmarks_run / cge_score on this ctx */
int cge_landmarks_info(cge_ctx *ctx, int64_t *N_out, int64_t *n_ledges_out, int *truncated);
/* the reference's 7-tuple (src/landmarks.jl:465): dii[N], embed[N x d col-major], cluster[N],
 * landmark_edges[n_ledges x 2 col-major], weights[n_ledges], lweight[N], v_to_l[n]              */
int cge_landmarks_fetch(cge_ctx *ctx, double *dii, double *embed, int64_t *cluster, int64_t *ledges,
                        double *lweights_e, double *lweight, int64_t *v_to_l);
/* runsplit() alone (src/landmarks.jl:279-345): 0-based group ids, for parity tests              */
int cge_runsplit(cge_ctx *ctx, const int64_t *clusters_flat, const int64_t *clusters_off, int64_t n_clusters,
                 int64_t nland, int64_t forced, int method, int64_t *group_ids);

/* ---- local-score samples --------------------------------------------------------------------- */
/* Counter-based draws (documented in DESIGN.md; the Julia stream cannot be reproduced):
 * pos_idx[k] uniform in 1..m (rows of the original edge list, `sample(E,S,replace=true)`),
 * (neg_i,neg_j)[k] uniform over NON-edges of the resident graph (`sample(NE,S,replace=true)`),
 * by rejection against the resident edge list, i<j for undirected, ordered i!=j for directed.
 * `stream_id` selects an independent stream (the alpha index when the run is unseeded).          */
int cge_draw_samples(cge_ctx *ctx, int64_t seed, int64_t stream_id, int64_t S, int directed, int64_t *pos_idx,
                     int64_t *neg_i, int64_t *neg_j);

/* ---- wGCL() / wGCL_directed(): src/divergence.jl:27-31, :282-286 -------------------------------- */
typedef struct {
    /* the 15 reference arguments, pointer + length */
    const int64_t *edges_src, *edges_dst; /* edges (m x 2)                     */
    const double *eweights;              /* m                                  */
    int64_t m;
    const int64_t *comm;                 /* comm (N x 1)                       */
    int64_t n_comm;
    const double *embed;                 /* N x d column-major                 */
    int64_t embed_rows, d;
    const double *distances;             /* N                                  */
    int64_t n_distances;
    const double *vweights;              /* N                                  */
    const double *init_vweights;         /* n or empty                         */
    int64_t n_init;
    const int64_t *v_to_l;               /* n or empty (empty <=> exact mode, src/divergence.jl:44) */
    int64_t n_v_to_l;
    const int64_t *init_edges_src, *init_edges_dst; /* m_init x 2 or empty    */
    int64_t m_init;
    const double *init_eweights;
    const double *init_embed;            /* n x d column-major or empty        */
    int split;                           /* --split-global                     */
    int64_t seed;                        /* -1 = unseeded                      */
    int64_t auc_samples;
    int verbose;
    int directed;                        /* 0: wGCL, 1: wGCL_directed          */
    /* optional pre-drawn samples so the caller can own the RNG (n_sets = 1, or one set per alpha);
     * NULL => the library draws them with cge_draw_samples(seed, ...)                              */
    const int64_t *pos_idx, *neg_i, *neg_j, *pos_idx2;
    int64_t n_sample_sets;
} cge_wgcl_args;

/* per-alpha trace (optional, for parity tests and the bench's evaluation count)                 */
typedef struct {
    int64_t n_alpha;
    int64_t iters[64];
    double div[64];
    double auc[64];
} cge_trace;

/* Host-array form: uploads what it is given, runs on the GPU, returns the reference's vector:
 * [best_alpha, best_div, best_div_ext, best_div_int, best_alpha_auc, best_auc, best_auc_err]
 * (*out_len = 7; 6 for the directed star-graph early return, src/divergence.jl:332-334).          */
int cge_wgcl(cge_ctx *ctx, const cge_wgcl_args *args, double out[7], int *out_len, cge_trace *trace);

/* Device-resident form of example/CGE_CLI.jl:10-24: landmarks() on the resident inputs, then
 * wGCL*() in landmark mode, with no host round trip of the landmark graph.  land = -1 runs
 * the exact mode on the resident graph (distances = zeros, CGE_CLI.jl:4).                         */
typedef struct {
    const int64_t *clusters_flat, *clusters_off;
    int64_t n_clusters;
    int64_t land, forced;
    int method, directed, split;
    int64_t seed, auc_samples;
} cge_score_args;
int cge_score(cge_ctx *ctx, const cge_score_args *args, double out[7], int *out_len, cge_trace *trace);

/* ---- louvain_clust(): src/clustering.jl:14-68 (called by parseargs when `-c` is omitted, src/auxilary.jl:115-121) ----
 * The communities the reference writes to <file>.ecg: LEVEL 1 of Louvain (`hierarchy -l 1`: the partition after the first
 * pass of local moving), here computed on the resident graph (cge_set_graph; weights honoured) by synchronous rounds on
 * the device.  comm_out[v] (v = 0 .. n-1) in 0 .. *n_comm-1; *modularity = modularity of that partition.  The reference
 * visits the vertices in an unseeded random order, so partitions are comparable in quality, not in identity.            */
int cge_louvain(cge_ctx *ctx, int64_t *comm_out, int64_t *n_comm, double *modularity, int64_t *rounds);

/* ---- small helpers (same arithmetic as the reference helpers, host side) -------------------- */
int64_t cge_idx(int64_t n, int64_t i, int64_t j);                                   /* src/auxilary.jl:57-59 */
int cge_js(cge_ctx *ctx, const double *vC, const double *vB, int64_t len, const uint8_t *vI, int internal,
           double *out);                                                            /* src/auxilary.jl:34-52 (device) */

/* ---- kernel-level entry points (device work units; used by the parity tests and the bench) -- */
/* per-edge scatter (src/landmarks.jl:433-451 + src/divergence.jl:59-63) on the resident edge list,
 * rows [e0,e1): wedges_out (N x N doubles, [a-1 + (b-1)*N], a<=b when !directed) and vect_C
 * (packed p(C) when !directed, C*C when directed) are HOST outputs (may be NULL).                */
int cge_edge_scatter(cge_ctx *ctx, const int64_t *v_to_l, int64_t N, int64_t C, int directed, int64_t e0,
                     int64_t e1, double *wedges_out, double *vect_C_out);
/* point-set diameter `hi` = maximum(full_graph_D) (src/divergence.jl:104-113) of the resident
 * embedding.  Shard `part` of `nparts` visits its share of the pair tiles (nparts = 1: all);
 * returns the distance of the shard's arg-max pair computed with dist()'s own arithmetic
 * (src/auxilary.jl:14-20), so the max over shards is the reference's `hi`.                      */
int cge_max_pair_dist(cge_ctx *ctx, int part, int nparts, double *hi, int64_t *arg_i, int64_t *arg_j);

/* ---- options / statistics --------------------------------------------------------------------- */
/* "diameter": 0 = auto (exact landmark-pair pruning, brute-force MFMA kernel when pruning is weak),
 *             1 = always brute force, 2 = always pruned.  All three return the same exact value.
 * "diameter_f32": how the point-to-reference maxima that bound the landmark pairs of the pruned diameter are computed.
 *             2 (default) = on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) with every operand split into two bf16 terms,
 *             1 = by fp32-input MFMA (v_mfma_f32_32x32x2_f32); both as rigorous UPPER bounds (fp64 norms, an error margin
 *             of 1.05 (3 (K + 2) 2^-23 + 3.2 2^-16) resp. 1.01 (K + 3) 2^-24 on the dot products, a non-finite accumulator
 *             counted as +Inf; values beyond 2^+-100 send the pass to fp64; K > 128 takes 1 for 2).  0 = in fp64.  The
 *             surviving pairs are evaluated in fp64 as always, so the diameter keeps its bits in every mode.
 * "landmark_edges": 1 = cge_score also builds the N x N landmark-pair matrix and its positive-entry count, i.e. all that
 *             landmarks() returns (src/landmarks.jl:433-463); 0 (default) = on the first cge_landmarks_fetch /
 *             cge_landmarks_info (an undirected score does not read it).  bench.py sets 1.
 * "shard_ingest": N > 1, set AFTER the collectives (cge_comm_init_rccl / cge_set_collectives) and BEFORE the uploads:
 *             1 = cge_set_graph keeps only this rank's slice [m r / W, m (r+1) / W) of the edge list resident (the scatter
 *             passes run over what a rank holds, the sampler's look-ups are exchanged) and cge_set_embedding uploads n / W
 *             rows per rank and all-gathers them device to device; every rank still passes the WHOLE arrays.  Entry points
 *             that need the whole list on one rank (exact-mode cge_score, cge_louvain, cge_edge_scatter, cge_draw_samples,
 *             caller-drawn samples of cge_wgcl) return CGE_E_ARG on a sharded list.  0 (default): every rank uploads and
 *             keeps everything.  Same results.
 * "fit_max_iterations": a Chung-Lu fit (src/divergence.jl:150-168, :434-467) that has not met `diff <= delta` after this many
 *             iterations returns CGE_E_ASSERT "Chung-Lu fit did not converge" (default 2 000 000).  The reference's loop is
 *             unbounded: on a directed graph without a fixed point within delta (e.g. two landmarks, tests/test_gpu_parity.py::
 *             test_directed_fit_without_a_fixed_point_raises) it never returns -- a documented divergence.
 * "shard_rows": N > 1, set AFTER the collectives and BEFORE cge_set_embedding / cge_set_embedding_device, with the communities
 *             already resident (cge_set_vertex_data first): 1 = the EMBEDDING ROWS ARE SHARDED BY COMMUNITY -- communities by
 *             decreasing size, each to the rank with the fewest rows so far; a rank uploads and keeps the rows of its own
 *             communities only (stat "rows_resident" ~ n / W; every rank still passes the whole array) and runs all splits,
 *             forced and global, of those communities.  What crosses ranks: per runsplit round four words per split group
 *             (status, low size, two values) for the replicated heap; v_to_l once (4 B per vertex); the landmarks' centroids /
 *             weights / d_ii once; the bound matrix of the diameter (all-reduce max), the seed row of its sweep and the rows of
 *             the few candidate landmarks of its exact stage; the rows of the sampled pairs of the local score once per seeded
 *             draw; row hashes for the unique-row clamp.  The clusters handed to the landmark phase must refine the community
 *             vector (CGE_E_ARG otherwise).  Entry points that need every row on one rank (exact-mode cge_score,
 *             cge_max_pair_dist, the brute-force diameter option) return CGE_E_ARG.  A later cge_set_vertex_data with ANOTHER
 *             community vector drops the resident rows (upload the embedding again).  0 (default): every rank holds every
 *             row.  Same landmark ids, diameter bits and score.
 * "fit_persistent": how the Chung-Lu fixed point (src/divergence.jl:150-168, :434-467) is launched.
 *             0 = auto: one persistent launch per alpha (the matrix register-resident, the workgroups exchanging
 *                 partial sums and iterates by polling the data itself) for score graphs of >= 128 vertices that fit
 *                 the register file (N <= ~4900 undirected, ~4000 directed on 256 CUs), else one launch per iteration;
 *             1 = always one launch per iteration; 2 = persistent whenever it fits.  Same iterates and iteration counts in
 *             all modes; sums are grouped differently between the launch-per-iteration and the persistent form (last-bit
 *             differences of the score vector).
 * "fit_fused": 1 (default) = in landmark mode (undirected, >= 256 landmarks) the sweep is relabelled by community and the rest
 *             of an alpha's chain rides on the launch of the persistent fit: (1 - D)^alpha in its prologue from the stored
 *             logarithm (no power matrix is written), the community-pair sums of P = T_i T_j (1 - D_ij)^alpha (vect_B,
 *             src/divergence.jl:226-234) per 64 x 64 tile and the tallies of the sampled pairs (:178-213) in its epilogue.
 *             0 = separate launches on the graph as given.  Same iteration counts; the sums of vect_B are grouped differently
 *             (last-bit differences).
 * "wedges_reduce_scatter": N > 1: 1 = the N x N landmark-pair matrix of landmarks() is reduce-scattered by row blocks
 *             (ncclReduceScatter / the hook's reduce_scatter_f64) instead of all-reduced; its consumers then work on row
 *             blocks and cge_landmarks_fetch of the edge list becomes COLLECTIVE (every rank must call it: the blocks are
 *             all-gathered).  0 (default): all-reduce, every fetch is local.
 * "shard_runsplit": with collectives set (N > 1): 1 (default) = the forced per-community phase of runsplit and the big
 *             batches of its global phase are split over the ranks and gathered by one all-reduce each (hook op 2);
 *             0 = runsplit replicated on every rank; 2 = every batch is split (tests).  Same landmark ids in all modes.
 * "shard_samples": with collectives set (N > 1): 1 (default) = from 10^5 local-score samples on, and with the in-library
 *             RCCL communicator, rank r tallies the samples [S r / W, S (r + 1) / W) of every alpha and the block tallies are
 *             all-reduced (2 x 64 doubles per alpha, on the stream); 2 = always (also through the hook; tests); 0 = never.
 *             The sums are grouped differently from a one-rank run (last-bit differences of elements 5-7).
 * "pow_exp2": 1 (default) = GD = (1 - D)^alpha as exp2(alpha * log2(1 - D)) with log2 computed once per score to ~70
 *             bits (a double and a float per entry; below one ulp, like the library pow); 0 = the library pow per alpha.
 * "test_bvec_plain": testing hook, 1 = vect_B through the kernels that serve score graphs of more than 8192 vertices /
 *             512 communities (no LDS staging); same additions in the same order, hence the same bits.
 * "exact_relabel": exact mode (v_to_l empty) beyond 8192 vertices: 1 (default) = the score graph is relabelled by community
 *             inside the sweep, so that vect_B's row sums are contiguous pieces of a row; 0 = vertices as given (A/B and
 *             tests; same iteration counts, scores equal up to the rounding of the fit's summation order).
 * "bvec_blocks": 1 = sweeps from 256 vertices on relabel the score graph by community and sum vect_B by 64 x 64 tiles (one read of
 *             GD); 0 (default) = row bins + row sums + fold below 8192 vertices (beyond that the sweep is relabelled and uses the
 *             tiles anyway).  Same results up to the rounding of the summation order; measured no faster at the headline.
 * "fit_persistent_test_timeout": testing hook, 1 = every persistent launch gives up at once (the host then
 *             restores the iterate and falls back to one launch per iteration).                              */
int cge_set_option(cge_ctx *ctx, const char *key, int64_t value);
/* "landmarks" (N of the last run, no side effects), "diameter_path" (1 brute / 2 pruned), "diameter_candidate_pairs", "diameter_candidate_tiles", "diameter_refs" (reference points) of the last run;
 * "collective_calls" / "collective_bytes" = all-reduces issued by the in-library RCCL path since cge_create;
 * "diameter_bits" = the bit pattern of the last `hi` (reinterpret the int64 as a double);
 * "fit_persistent_alphas" = alphas of the last sweep fitted by a persistent launch, "fit_persistent_fallbacks" = persistent fits abandoned since the context was created, "fit_iterations" = Chung-Lu
 * iterations of the last sweep (all alphas); "landmark_batches" / "landmark_batch_rows" / "landmark_splits" =
 * device batches of the last runsplit, the rows they covered, the groups they split; "cut_tie_tasks" = groups of the size / diameter rules
 * of the last runsplit that held a row exactly ON the cut (settled by the reference's sequential rule; read from the device: synchronises);
 * "edges_resident" / "edges_total" (option shard_ingest), "rows_resident" / "rows_total" (option shard_rows: the embedding rows this
 * rank holds -- about n / W -- and n), "embedding_words_resident" (doubles of the resident row matrix)   */
int cge_get_stat(cge_ctx *ctx, const char *key, int64_t *value);

/* ---- text tables (the step right before the path: `readdlm` at src/auxilary.jl:80-168) -------------------------
 * Parallel reader of a whitespace-delimited numeric table (edge list, community file, embedding): the file is
 * mapped, cut at line boundaries into one piece per thread and parsed with exactly-rounded conversions.  Blank lines
 * are skipped; a first line whose field count differs from the second line's is taken as a header (node2vec's
 * "n d") and skipped, as the reference's retry with skipstart = 1 does (:151-156).  Needs no GPU and no context.
 * n_threads <= 0: one per hardware thread (at most 32).  `err` (optional) receives the message of a failure.
 * open: map the file, find the shape (one pass that counts lines).  parse: convert straight into the caller's matrix,
 * row-major or column-major (Julia's Matrix{Float64}), `out` = rows*cols doubles; no intermediate copy is held, so
 * a table of B bytes of text costs rows*cols*8 bytes of memory and nothing else.                                    */
int cge_text_table_open(const char *path, int n_threads, int64_t *rows, int64_t *cols, int *header_skipped, void **handle,
                        char *err, int64_t err_len);
int cge_text_table_parse(void *handle, double *out, int column_major, char *err, int64_t err_len);
void cge_text_table_close(void *handle);

/* ---- profiling ------------------------------------------------------------------------------ */
/* When enabled, every launch of the named kernels is bracketed by hipEvents on the ctx stream.   */
int cge_profile_enable(cge_ctx *ctx, int on);
int cge_profile_reset(cge_ctx *ctx);
/* restrict the timers to the comma-separated names (NULL or "": every timer); fewer event pairs on the stream        */
int cge_profile_select(cge_ctx *ctx, const char *names);
/* name: "edge_scatter", "max_pair_dist", "fit_symv", ...; returns launches and total milliseconds */
int cge_profile_get(cge_ctx *ctx, const char *name, int64_t *launches, double *total_ms);
int cge_profile_names(cge_ctx *ctx, char *buf, int64_t buf_len); /* comma-separated */
/* wall-clock phase timers of the last cge_score call: "landmarks","aggregate","scatter","dist",
 * "diameter","sweep","samples" (milliseconds, host clock around stream-synchronised phases)      */
int cge_phase_ms(cge_ctx *ctx, const char *phase, double *ms);
int cge_phase_names(cge_ctx *ctx, char *buf, int64_t buf_len); /* comma-separated, incl. the lm_* sub-phases */

#ifdef __cplusplus
}
#endif
#endif /* CGE_HIP_H */
