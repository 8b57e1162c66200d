/*
 * cge_driver.c -- a plain C caller of the C-ABI in include/cge_hip.h (no Python, no Julia, no torch): the call order
 * of the reference's command-line script example/CGE_CLI.jl:3-25 --
 *   parseargs (src/auxilary.jl:61-220; here the subset -g -c -e -l -f -m --seed --samples-local -d --split-global,
 *   files read with the library's cge_text_table_* reader)  ->  landmarks() (cge_set_* + cge_landmarks_run +
 *   cge_landmarks_fetch)  ->  wGCL() / wGCL_directed() on the landmark graph with the init_* copies (cge_wgcl)  ->
 *   println(results) on stdout, one "." per alpha + newline on stderr (src/divergence.jl:140,255).
 * Without -l (and below 10 000 vertices) it runs the exact mode, as the script does.
 *
 *   gcc -O2 -Iinclude examples/cge_driver.c -o cge_driver -Lcge.jl_amd/csrc/build -lcge_hip -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cge_hip.h"

static void die(const char *what, const char *msg) {
    fprintf(stderr, "cge_driver: %s: %s\n", what, msg ? msg : "");
    exit(1);
}
#define CK(ctx, call)                                                  \
    do {                                                               \
        int _rc = (call);                                              \
        if (_rc != CGE_OK) {                                           \
            fprintf(stderr, "cge_driver: %s -> %d: %s\n", #call, _rc, cge_last_error(ctx)); \
            exit(1);                                                   \
        }                                                              \
    } while (0)

static const char *flag(int argc, char **argv, const char *name) {
    for (int i = 1; i + 1 < argc; i++)
        if (!strcmp(argv[i], name)) return argv[i + 1];
    return NULL;
}
static int has(int argc, char **argv, const char *name) {
    for (int i = 1; i < argc; i++)
        if (!strcmp(argv[i], name)) return 1;
    return 0;
}

/* readdlm(fn, Float64) -> column-major matrix */
static double *read_table(const char *path, int64_t *rows, int64_t *cols) {
    char err[512] = {0};
    int hdr = 0;
    void *h = NULL;
    if (cge_text_table_open(path, 0, rows, cols, &hdr, &h, err, sizeof err) != CGE_OK) die(path, err);
    double *M = (double *)malloc(sizeof(double) * (size_t)(*rows) * (size_t)(*cols));
    if (!M) die(path, "out of memory");
    if (cge_text_table_parse(h, M, 1, err, sizeof err) != CGE_OK) die(path, err);
    cge_text_table_close(h);
    return M;
}

/* shortest decimal string that reads back as the same double (what Julia's println shows) */
static void print_shortest(double x) {
    char b[40];
    if (isnan(x)) { printf("NaN"); return; }
    if (isinf(x)) { printf(x > 0 ? "Inf" : "-Inf"); return; }
    for (int p = 1; p <= 17; p++) {
        snprintf(b, sizeof b, "%.*g", p, x);
        if (strtod(b, NULL) == x) break;
    }
    char *e = strchr(b, 'e');
    if (e) { /* "1e-05" -> "1.0e-5" (Julia's exponent form) */
        *e = 0;
        printf("%s%se%d", b, strchr(b, '.') ? "" : ".0", atoi(e + 1));
        return;
    }
    if (!strchr(b, '.')) strcat(b, ".0");
    printf("%s", b);
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

int main(int argc, char **argv) {
    const char *fg = flag(argc, argv, "-g"), *fc = flag(argc, argv, "-c"), *fe = flag(argc, argv, "-e");
    if (!fg || !fc || !fe) die("usage", "cge_driver -g edgelist -c communities -e embedding [-l n] [-f n] [-m rss|rss2|size|diameter] "
                                        "[--seed n] [--samples-local n] [-d] [--split-global] [--force-exact]");
    const int directed = has(argc, argv, "-d"), split = has(argc, argv, "--split-global");
    int64_t seed = flag(argc, argv, "--seed") ? atoll(flag(argc, argv, "--seed")) : -1;
    int64_t samples = flag(argc, argv, "--samples-local") ? atoll(flag(argc, argv, "--samples-local")) : 10000;
    int64_t forced = flag(argc, argv, "-f") ? atoll(flag(argc, argv, "-f")) : 4;
    int method = CGE_METHOD_RSS;
    const char *ms = flag(argc, argv, "-m");
    if (ms) {
        if (!strcmp(ms, "rss")) method = CGE_METHOD_RSS;
        else if (!strcmp(ms, "rss2")) method = CGE_METHOD_RSS2;
        else if (!strcmp(ms, "size")) method = CGE_METHOD_SIZE;
        else if (!strcmp(ms, "diameter")) method = CGE_METHOD_DIAMETER;
        else die("-m", "unknown split rule");
    }

    /* ---- edge list (src/auxilary.jl:86-110): 2 or 3 columns, 0- or 1-based ---- */
    int64_t m, ecols;
    double *E = read_table(fg, &m, &ecols);
    if (ecols != 2 && ecols != 3) die(fg, "Expected 2 or 3 columns in edgelist file");
    double vmin = E[0], vmax = E[0];
    for (int64_t k = 0; k < 2 * m; k++) { if (E[k] < vmin) vmin = E[k]; if (E[k] > vmax) vmax = E[k]; }
    if (vmin != 0.0 && vmin != 1.0) die(fg, "Vertices should be either 0-based or 1-based");
    const int64_t shift = vmin == 0.0 ? 1 : 0, n = (int64_t)vmax + shift;
    int64_t *src = malloc(sizeof(int64_t) * m), *dst = malloc(sizeof(int64_t) * m);
    double *ew = malloc(sizeof(double) * m), *vw = calloc((size_t)n, sizeof(double));
    for (int64_t e = 0; e < m; e++) {
        src[e] = (int64_t)E[e] + shift;
        dst[e] = (int64_t)E[e + m] + shift;
        ew[e] = ecols == 3 ? E[e + 2 * m] : 1.0;
        vw[src[e] - 1] += ew[e]; /* :107-110 */
        vw[dst[e] - 1] += ew[e];
    }
    free(E);

    /* ---- communities (:122-139): 1 column, or (id, community) pairs ---- */
    int64_t crow, ccols;
    double *Cm = read_table(fc, &crow, &ccols);
    if (crow != n) die(fc, "No. communities differ from no. nodes");
    int64_t *comm = malloc(sizeof(int64_t) * n);
    if (ccols == 1)
        for (int64_t i = 0; i < n; i++) comm[i] = (int64_t)Cm[i];
    else if (ccols == 2) { /* sort by the id column (ids are a permutation of 0..n-1 or 1..n) */
        double idmin = Cm[0];
        for (int64_t i = 0; i < n; i++) if (Cm[i] < idmin) idmin = Cm[i];
        for (int64_t i = 0; i < n; i++) comm[(int64_t)(Cm[i] - idmin)] = (int64_t)Cm[i + n];
    } else
        die(fc, "Expected 1 or 2 columns in communities file");
    free(Cm);
    int64_t cmin = comm[0], C = 0;
    for (int64_t i = 0; i < n; i++) if (comm[i] < cmin) cmin = comm[i];
    if (cmin != 0 && cmin != 1) die(fc, "Communities should be either 0-based or 1-based");
    for (int64_t i = 0; i < n; i++) { comm[i] += (cmin == 0); if (comm[i] > C) C = comm[i]; }

    /* ---- embedding (:150-167): optional id column ---- */
    int64_t xrow, xcols;
    double *X0 = read_table(fe, &xrow, &xcols);
    if (xrow != n) die(fe, "No. rows in embedding and no. vertices in a graph differ.");
    int ids = 1;
    for (int64_t i = 0; i < n && ids; i++) ids = X0[i] == floor(X0[i]);
    int64_t d = ids ? xcols - 1 : xcols;
    double *X = malloc(sizeof(double) * (size_t)n * (size_t)d);
    if (ids) { /* sort rows by the integral first column (stable: ids are distinct) */
        int64_t *key = malloc(sizeof(int64_t) * 2 * n);
        for (int64_t i = 0; i < n; i++) { key[2 * i] = (int64_t)X0[i]; key[2 * i + 1] = i; }
        qsort(key, (size_t)n, 2 * sizeof(int64_t), cmp_i64);
        for (int64_t k = 0; k < d; k++)
            for (int64_t i = 0; i < n; i++) X[i + k * n] = X0[key[2 * i + 1] + (k + 1) * n];
        free(key);
    } else
        memcpy(X, X0, sizeof(double) * (size_t)n * (size_t)d);
    free(X0);

    /* ---- landmark count (:175-197) ---- */
    int64_t land = -1;
    if (has(argc, argv, "-l")) {
        const char *lv = flag(argc, argv, "-l");
        char *end = NULL;
        land = lv ? strtoll(lv, &end, 10) : 0;
        if (!lv || end == lv || *end) land = (int64_t)llround(4.0 * sqrt((double)n));
    }
    if (has(argc, argv, "-f") && land == -1) land = 1;
    if (n >= 10000 && !has(argc, argv, "--force-exact") && land == -1) {
        land = (int64_t)llround(4.0 * sqrt((double)n));
        if (4 * C > land) land = 4 * C;
    }

    cge_ctx *ctx = NULL;
    if (cge_create(&ctx, 0, NULL) != CGE_OK) die("cge_create", "no MI355X visible (there is no CPU fallback)");
    double out[7];
    int out_len = 7;
    cge_trace tr;
    memset(&tr, 0, sizeof tr);
    cge_wgcl_args a;
    memset(&a, 0, sizeof a);
    a.split = split; a.seed = seed; a.auc_samples = samples; a.directed = directed;
    double *zeros = calloc((size_t)n, sizeof(double));
    if (land != -1) { /* CGE_CLI.jl:10-17 */
        int64_t *flat = malloc(sizeof(int64_t) * n), *off = calloc((size_t)C + 1, sizeof(int64_t)), *cur;
        for (int64_t i = 0; i < n; i++) off[comm[i]]++;
        for (int64_t c = 0; c < C; c++) off[c + 1] += off[c];
        cur = malloc(sizeof(int64_t) * (C + 1));
        memcpy(cur, off, sizeof(int64_t) * (C + 1));
        for (int64_t i = 0; i < n; i++) flat[cur[comm[i] - 1]++] = i + 1; /* clusters[c] = members in ascending id (:199-208) */
        CK(ctx, cge_set_graph(ctx, src, dst, ew, m, n));
        CK(ctx, cge_set_embedding(ctx, X, n, d));
        CK(ctx, cge_set_vertex_data(ctx, comm, vw, n));
        int64_t N = 0, ne = 0;
        int trunc = 0;
        CK(ctx, cge_landmarks_run(ctx, flat, off, C, land, forced, method, directed, &N, &ne, &trunc));
        if (trunc) fprintf(stderr, "Warning: Requested number of clusters larger than unique no. embeddings. Truncating.\n");
        double *dii = malloc(sizeof(double) * N), *lemb = malloc(sizeof(double) * N * d), *lw = malloc(sizeof(double) * ne),
               *lweight = malloc(sizeof(double) * N);
        int64_t *lcomm = malloc(sizeof(int64_t) * N), *ledges = malloc(sizeof(int64_t) * 2 * ne), *v2l = malloc(sizeof(int64_t) * n);
        CK(ctx, cge_landmarks_fetch(ctx, dii, lemb, lcomm, ledges, lw, lweight, v2l));
        a.edges_src = ledges; a.edges_dst = ledges + ne; a.eweights = lw; a.m = ne;
        a.comm = lcomm; a.n_comm = N; a.embed = lemb; a.embed_rows = N; a.d = d;
        a.distances = dii; a.n_distances = N; a.vweights = lweight;
        a.init_vweights = vw; a.n_init = n; a.v_to_l = v2l; a.n_v_to_l = n;
        a.init_edges_src = src; a.init_edges_dst = dst; a.m_init = m; a.init_eweights = ew; a.init_embed = X;
        CK(ctx, cge_wgcl(ctx, &a, out, &out_len, &tr));
    } else { /* exact mode: distances = zeros, empty init_* and v_to_l (CGE_CLI.jl:4-9) */
        a.edges_src = src; a.edges_dst = dst; a.eweights = ew; a.m = m;
        a.comm = comm; a.n_comm = n; a.embed = X; a.embed_rows = n; a.d = d;
        a.distances = zeros; a.n_distances = n; a.vweights = vw;
        CK(ctx, cge_wgcl(ctx, &a, out, &out_len, &tr));
    }
    for (int64_t k = 0; k < tr.n_alpha; k++) fputc('.', stderr);
    fputc('\n', stderr);
    printf("[");
    for (int k = 0; k < out_len; k++) { if (k) printf(", "); print_shortest(out[k]); }
    printf("]\n");
    cge_destroy(ctx); /* before exit(): streams and events must not outlive the HIP runtime */
    return 0;
}
