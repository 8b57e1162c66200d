/*
 * cge_oracle.c -- CPU restatement (plain C99, fp64, single thread, loop for loop)
 * of the divergence-scoring hot path of KrainskiL/CGE.jl v2.0.2.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (cge.jl_amd/) never links, imports or executes anything in oracle/.
 *
 * Parity status: PINNED for result elements 1-4 by the reference's only
 * known-answer vector (README.md:88-100, checked in tests/test_oracle_golden.py);
 * UNPINNED for (a) the local-score sampling stream (Julia RNG + StatsBase.sample +
 * Base.Set iteration order, src/divergence.jl:137,184-210) -- the oracle takes the
 * sampled pairs as INPUT -- and (b) the sign of LAPACK's eigenvector
 * (src/landmarks.jl:99,162,225,254), hence raw landmark ids; the oracle fixes the
 * sign so that the component of largest magnitude is positive.
 *
 * Conventions: exactly Julia's -- indices are 1-based int64, matrices are
 * column-major (edges m x 2 = two contiguous columns; embedding n x d = d
 * contiguous columns of length n).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define ORC_OK 0
#define ORC_E_ASSERT (-1)       /* a reference @assert would have fired          */
#define ORC_E_HOMOGENEOUS (-2)  /* "Trying to split homogenous cluster"          */
#define ORC_E_EMPTY (-3)        /* "Unexpected empty cluster generated"          */
#define ORC_E_ALLOC (-4)

typedef int64_t i64;

/* ------------------------------------------------------------------------- */
/* src/auxilary.jl:57-59  idx(n,i,j): packed upper-triangular index, 1-based  */
i64 orc_idx(i64 n, i64 i, i64 j) {
    return n * (i - 1) - (i - 1) * (i - 2) / 2 + j - i + 1;
}

/* src/auxilary.jl:14-20  dist(v1,v2,embed): Euclidean distance of two rows   */
double orc_dist(i64 v1, i64 v2, const double *embed, i64 n, i64 d) {
    if (v1 == v2) return 0.0;
    double s = 0.0;
    for (i64 k = 0; k < d; k++) {
        double t = embed[(v1 - 1) + k * n] - embed[(v2 - 1) + k * n];
        s += t * t;
    }
    return sqrt(s);
}

/* src/auxilary.jl:34-52  JS(vC,vB,vI,internal) with the +1 prior.
 * vI == NULL  <=> empty indicator (use every bin).                           */
double orc_js(const double *vC, const double *vB, i64 len, const uint8_t *vI, int internal) {
    double sp1 = 0.0, sp2 = 0.0;
    i64 cnt = 0;
    for (i64 k = 0; k < len; k++) {
        if (vI && ((vI[k] != 0) != (internal != 0))) continue;
        sp1 += vC[k];
        sp2 += vB[k];
        cnt++;
    }
    sp1 += (double)cnt;
    sp2 += (double)cnt;
    double f = 0.0;
    for (i64 k = 0; k < len; k++) {
        if (vI && ((vI[k] != 0) != (internal != 0))) continue;
        double p = (vC[k] + 1.0) / sp1;
        double q = (vB[k] + 1.0) / sp2;
        double m = (p + q) / 2.0;
        f += p * log(p / m) + q * log(q / m);
    }
    return f / 2.0;
}

/* ------------------------------------------------------------------------- */
/* Landmark heap: src/landmarks.jl:5-46                                       */
typedef struct {
    i64 *what;   /* 1-based vertex ids, owned */
    i64 len;
    double value;
} lm_t;

typedef struct {
    lm_t *a;     /* a[1..len] (slot 0 unused) */
    i64 len, cap;
} heap_t;

static int heap_init(heap_t *h) {
    h->cap = 64;
    h->len = 0;
    h->a = (lm_t *)malloc(sizeof(lm_t) * (size_t)(h->cap + 1));
    return h->a ? ORC_OK : ORC_E_ALLOC;
}
static void heap_free(heap_t *h) {
    for (i64 i = 1; i <= h->len; i++) free(h->a[i].what);
    free(h->a);
    h->a = NULL;
    h->len = 0;
}
/* landmark_put!  src/landmarks.jl:12-25 (takes ownership of what)            */
static int heap_put(heap_t *h, i64 *what, i64 len, double value) {
    if (h->len + 1 > h->cap) {
        h->cap *= 2;
        lm_t *na = (lm_t *)realloc(h->a, sizeof(lm_t) * (size_t)(h->cap + 1));
        if (!na) return ORC_E_ALLOC;
        h->a = na;
    }
    lm_t p = {what, len, value};
    h->len += 1;
    i64 i = h->len, j;
    while ((j = i / 2) >= 1) {
        if (value < h->a[j].value) {
            h->a[i] = h->a[j];
            i = j;
        } else
            break;
    }
    h->a[i] = p;
    return ORC_OK;
}
/* landmark_pop!  src/landmarks.jl:27-46                                      */
static lm_t heap_pop(heap_t *h) {
    lm_t x = h->a[1];
    lm_t y = h->a[h->len];
    h->len -= 1;
    if (h->len > 0) {
        i64 i = 1, len = h->len, l;
        while ((l = 2 * i) <= len) {
            i64 r = 2 * i + 1;
            i64 j = (r > len || h->a[l].value < h->a[r].value) ? l : r;
            if (h->a[j].value < y.value) {
                h->a[i] = h->a[j];
                i = j;
            } else
                break;
        }
        h->a[i] = y;
    }
    return x;
}

/* ------------------------------------------------------------------------- */
/* WSSE triple: src/landmarks.jl:50-67                                        */
typedef struct { double ss, s, ws; } wsse_t;
static inline double wsse_val(wsse_t x) { return x.ss - x.s * x.s / x.ws; }

/* view(embedding, idxs, :) / view(w, idxs)                                   */
typedef struct {
    const double *emb; /* n x d column-major */
    const double *w;   /* n                  */
    i64 n, d;
    const i64 *idxs;   /* 1-based rows       */
    i64 k;
} view_t;
#define VM(v, j, c) ((v)->emb[((v)->idxs[(j)] - 1) + (c) * (v)->n])
#define VW(v, j) ((v)->w[(v)->idxs[(j)] - 1])

/* total_rss(m,w): src/landmarks.jl:269 with WSSE(x,w) src/landmarks.jl:56-61 */
static double total_rss(const view_t *v) {
    double tot = 0.0;
    for (i64 c = 0; c < v->d; c++) {
        wsse_t a = {0.0, 0.0, 0.0};
        for (i64 j = 0; j < v->k; j++) {
            double x = VM(v, j, c), w = VW(v, j);
            a.ss += w * (x * x);
            a.s += w * x;
            a.ws += w;
        }
        tot += wsse_val(a);
    }
    return tot;
}

/* Largest-eigenvalue eigenvector of a symmetric d x d matrix (row-major A,
 * destroyed) by cyclic Jacobi.  Stands in for `eigvecs(A)[:, end]`
 * (src/landmarks.jl:99,162,225,254; LAPACK syevr, ascending eigenvalues).
 * Sign convention (the reference's is unspecified): largest-|.| component > 0. */
/* Fixture generation only: a caller-supplied routine for the eigenvector of the largest eigenvalue
 * (oracle.py::use_lapack_eig hands in LAPACK's syevr through scipy -- the routine Julia's `eigvecs`
 * itself calls).  The sign rule below is applied to its result as to Jacobi's.  NULL = Jacobi.    */
typedef int (*orc_eig_fn)(const double *A, i64 d, double *v);
static orc_eig_fn g_eig_hook = 0;
void orc_set_eig_hook(orc_eig_fn fn) { g_eig_hook = fn; }

static void eig_sign_rule(double *vout, i64 d) {
    i64 big = 0;
    for (i64 k = 1; k < d; k++)
        if (fabs(vout[k]) > fabs(vout[big])) big = k;
    if (vout[big] < 0.0)
        for (i64 k = 0; k < d; k++) vout[k] = -vout[k];
}

static int eig_top(double *A, i64 d, double *vout) {
    if (g_eig_hook) {
        int rc = g_eig_hook(A, d, vout);
        if (rc != 0) return ORC_E_ASSERT;
        eig_sign_rule(vout, d);
        return ORC_OK;
    }
    double *V = (double *)malloc(sizeof(double) * (size_t)(d * d));
    if (!V) return ORC_E_ALLOC;
    for (i64 i = 0; i < d; i++)
        for (i64 j = 0; j < d; j++) V[i * d + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0, diag = 0.0;
        for (i64 i = 0; i < d; i++) {
            diag += A[i * d + i] * A[i * d + i];
            for (i64 j = i + 1; j < d; j++) off += A[i * d + j] * A[i * d + j];
        }
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
        for (i64 p = 0; p < d - 1; p++) {
            for (i64 q = p + 1; q < d; q++) {
                double apq = A[p * d + q];
                if (apq == 0.0) continue;
                double app = A[p * d + p], aqq = A[q * d + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (i64 k = 0; k < d; k++) { /* columns p,q */
                    double akp = A[k * d + p], akq = A[k * d + q];
                    A[k * d + p] = c * akp - s * akq;
                    A[k * d + q] = s * akp + c * akq;
                }
                for (i64 k = 0; k < d; k++) { /* rows p,q */
                    double apk = A[p * d + k], aqk = A[q * d + k];
                    A[p * d + k] = c * apk - s * aqk;
                    A[q * d + k] = s * apk + c * aqk;
                }
                for (i64 k = 0; k < d; k++) {
                    double vkp = V[k * d + p], vkq = V[k * d + q];
                    V[k * d + p] = c * vkp - s * vkq;
                    V[k * d + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    i64 best = 0;
    for (i64 i = 1; i < d; i++)
        if (A[i * d + i] > A[best * d + best]) best = i;
    i64 big = 0;
    for (i64 k = 0; k < d; k++) {
        vout[k] = V[k * d + best];
        if (fabs(vout[k]) > fabs(vout[big])) big = k;
    }
    if (vout[big] < 0.0)
        for (i64 k = 0; k < d; k++) vout[k] = -vout[k];
    free(V);
    return ORC_OK;
}

/* Common prefix of all four rules (src/landmarks.jl:97-99,160-162,223-225,
 * 252-254): mu = matrix_w_mean (:71-81); y = (m - mu) .* sqrt.(w); A = y'y;
 * v = eigvecs(A)[:, end]; z = y*v.                                           */
static int pca_project(const view_t *v, double *z) {
    i64 d = v->d, k = v->k;
    double *mu = (double *)malloc(sizeof(double) * (size_t)d);
    double *y = (double *)malloc(sizeof(double) * (size_t)(k * d)); /* y[j*d+c] */
    double *A = (double *)calloc((size_t)(d * d), sizeof(double));
    double *ev = (double *)malloc(sizeof(double) * (size_t)d);
    if (!mu || !y || !A || !ev) { free(mu); free(y); free(A); free(ev); return ORC_E_ALLOC; }
    double sw = 0.0;
    for (i64 j = 0; j < k; j++) sw += VW(v, j);
    for (i64 c = 0; c < d; c++) {
        double r = 0.0;
        for (i64 j = 0; j < k; j++) r += VM(v, j, c) * VW(v, j);
        mu[c] = r / sw;
    }
    for (i64 j = 0; j < k; j++) {
        double sq = sqrt(VW(v, j));
        for (i64 c = 0; c < d; c++) y[j * d + c] = (VM(v, j, c) - mu[c]) * sq;
    }
    for (i64 j = 0; j < k; j++)
        for (i64 a = 0; a < d; a++) {
            double ya = y[j * d + a];
            for (i64 b = a; b < d; b++) A[a * d + b] += ya * y[j * d + b];
        }
    for (i64 a = 0; a < d; a++)
        for (i64 b = 0; b < a; b++) A[a * d + b] = A[b * d + a];
    int rc = eig_top(A, d, ev);
    if (rc == ORC_OK)
        for (i64 j = 0; j < k; j++) {
            double s = 0.0;
            for (i64 c = 0; c < d; c++) s += y[j * d + c] * ev[c];
            z[j] = s;
        }
    free(mu); free(y); free(A); free(ev);
    return rc;
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}
/* Statistics.median: sorted middle; even length -> x/2 + y/2 (Statistics.middle) */
static double median_of(const double *z, const i64 *sel, i64 cnt, double *scratch) {
    for (i64 i = 0; i < cnt; i++) scratch[i] = sel ? z[sel[i]] : z[i];
    qsort(scratch, (size_t)cnt, sizeof(double), cmp_double);
    if (cnt & 1) return scratch[cnt / 2];
    return scratch[cnt / 2 - 1] / 2.0 + scratch[cnt / 2] / 2.0;
}

/* sum(wsse, rss) over the d columns                                          */
static double sum_wsse(const wsse_t *r, i64 d) {
    double s = 0.0;
    for (i64 c = 0; c < d; c++) s += wsse_val(r[c]);
    return s;
}
/* out[c] = base[c] + WSSE(view(m,sel,c), view(w,sel))  (empty sel -> zeros)  */
static void add_wsse_set(const view_t *v, const wsse_t *base, const i64 *sel, i64 cnt, wsse_t *out) {
    for (i64 c = 0; c < v->d; c++) {
        wsse_t a = {0.0, 0.0, 0.0};
        for (i64 t = 0; t < cnt; t++) {
            double x = VM(v, sel[t], c), w = VW(v, sel[t]);
            a.ss += w * (x * x);
            a.s += w * x;
            a.ws += w;
        }
        out[c].ss = base[c].ss + a.ss;
        out[c].s = base[c].s + a.s;
        out[c].ws = base[c].ws + a.ws;
    }
}

/* A rule returns two lists of LOCAL 0-based positions (into the view).       */
typedef struct { i64 *low, *high; i64 nlow, nhigh; } split_t;

/* split_cluster_rss: src/landmarks.jl:155-210                                */
static int rule_rss(const view_t *v, split_t *out) {
    i64 k = v->k, d = v->d;
    if (k <= 1) return ORC_E_ASSERT;
    out->low = (i64 *)malloc(sizeof(i64) * (size_t)k);
    out->high = (i64 *)malloc(sizeof(i64) * (size_t)k);
    out->nlow = out->nhigh = 0;
    if (k == 2) { out->low[0] = 0; out->nlow = 1; out->high[0] = 1; out->nhigh = 1; return ORC_OK; }
    double *z = (double *)malloc(sizeof(double) * (size_t)k);
    double *scr = (double *)malloc(sizeof(double) * (size_t)k);
    i64 *gray = (i64 *)malloc(sizeof(i64) * (size_t)k);
    i64 *t1 = (i64 *)malloc(sizeof(i64) * (size_t)k), *t2 = (i64 *)malloc(sizeof(i64) * (size_t)k);
    wsse_t *rl = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d), *rh = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d);
    wsse_t *rlt = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d), *rht = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d);
    int rc = pca_project(v, z);
    if (rc != ORC_OK) goto done;
    i64 imin = 0, imax = 0; /* argmin/argmax: first occurrence */
    for (i64 j = 1; j < k; j++) {
        if (z[j] < z[imin]) imin = j;
        if (z[j] > z[imax]) imax = j;
    }
    if (imin == imax) { rc = ORC_E_HOMOGENEOUS; goto done; }
    out->low[out->nlow++] = imin;
    out->high[out->nhigh++] = imax;
    i64 ng = 0;
    for (i64 j = 0; j < k; j++)
        if (j != imin && j != imax) gray[ng++] = j;
    for (i64 c = 0; c < d; c++) { /* :169-170 */
        double x1 = VM(v, imin, c), w1 = VW(v, imin), x2 = VM(v, imax, c), w2 = VW(v, imax);
        rl[c].ss = x1 * x1 * w1; rl[c].s = x1 * w1; rl[c].ws = w1;
        rh[c].ss = x2 * x2 * w2; rh[c].s = x2 * w2; rh[c].ws = w2;
    }
    double med = median_of(z, NULL, k, scr);
    for (;;) {
        i64 n1 = 0, n2 = 0;
        for (i64 t = 0; t < ng; t++) {
            if (z[gray[t]] < med) t1[n1++] = gray[t];
            else t2[n2++] = gray[t];
        }
        add_wsse_set(v, rl, t1, n1, rlt);
        add_wsse_set(v, rh, t2, n2, rht);
        if (sum_wsse(rlt, d) < sum_wsse(rht, d)) {
            if (n1 == 0) break;
            memcpy(rl, rlt, sizeof(wsse_t) * (size_t)d);
            for (i64 t = 0; t < n1; t++) out->low[out->nlow++] = t1[t];
            memcpy(gray, t2, sizeof(i64) * (size_t)n2);
            ng = n2;
        } else {
            if (n2 == 0) break;
            memcpy(rh, rht, sizeof(wsse_t) * (size_t)d);
            for (i64 t = 0; t < n2; t++) out->high[out->nhigh++] = t2[t];
            memcpy(gray, t1, sizeof(i64) * (size_t)n1);
            ng = n1;
        }
        if (ng == 0) break;
        med = median_of(z, gray, ng, scr);
    }
    if (ng > 0) { /* :200-208 */
        add_wsse_set(v, rl, gray, ng, rlt);
        add_wsse_set(v, rh, gray, ng, rht);
        double a = fmax(sum_wsse(rlt, d), sum_wsse(rh, d));
        double b = fmax(sum_wsse(rl, d), sum_wsse(rht, d));
        if (a < b) for (i64 t = 0; t < ng; t++) out->low[out->nlow++] = gray[t];
        else for (i64 t = 0; t < ng; t++) out->high[out->nhigh++] = gray[t];
    }
done:
    free(z); free(scr); free(gray); free(t1); free(t2); free(rl); free(rh); free(rlt); free(rht);
    return rc;
}

/* stable ascending sortperm (merge sort), src/landmarks.jl:100               */
static void sortperm_stable(const double *z, i64 k, i64 *p) {
    i64 *tmp = (i64 *)malloc(sizeof(i64) * (size_t)k);
    for (i64 i = 0; i < k; i++) p[i] = i;
    for (i64 width = 1; width < k; width *= 2) {
        for (i64 lo = 0; lo < k; lo += 2 * width) {
            i64 mid = lo + width < k ? lo + width : k, hi = lo + 2 * width < k ? lo + 2 * width : k;
            i64 a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = (z[p[b]] < z[p[a]]) ? p[b++] : p[a++];
            while (a < mid) tmp[o++] = p[a++];
            while (b < hi) tmp[o++] = p[b++];
        }
        memcpy(p, tmp, sizeof(i64) * (size_t)k);
    }
    free(tmp);
}

static inline wsse_t wsse_one(double x, double w) { /* WSSE(x::Real,w::Real) :62 */
    wsse_t r = {w * (x * x), w * x, w};
    return r;
}

/* split_cluster_rss2: src/landmarks.jl:92-147                                */
static int rule_rss2(const view_t *v, split_t *out) {
    i64 k = v->k, d = v->d;
    if (k <= 1) return ORC_E_ASSERT;
    out->low = (i64 *)malloc(sizeof(i64) * (size_t)k);
    out->high = (i64 *)malloc(sizeof(i64) * (size_t)k);
    out->nlow = out->nhigh = 0;
    if (k == 2) { out->low[0] = 0; out->nlow = 1; out->high[0] = 1; out->nhigh = 1; return ORC_OK; }
    double *z = (double *)malloc(sizeof(double) * (size_t)k);
    i64 *p = (i64 *)malloc(sizeof(i64) * (size_t)k);
    wsse_t *rl = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d), *rh = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d);
    wsse_t *rlt = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d), *rht = (wsse_t *)malloc(sizeof(wsse_t) * (size_t)d);
    int rc = pca_project(v, z);
    if (rc != ORC_OK) goto done;
    sortperm_stable(z, k, p);
    i64 low = 0, high = k - 1; /* 0-based positions in p */
    for (i64 c = 0; c < d; c++) {
        rl[c] = wsse_one(VM(v, p[0], c), VW(v, p[0]));
        rh[c] = wsse_one(VM(v, p[k - 1], c), VW(v, p[k - 1]));
    }
    while (low + 1 < high) {
        if (sum_wsse(rl, d) < sum_wsse(rh, d)) {
            low += 1;
            for (i64 c = 0; c < d; c++) {
                wsse_t a = wsse_one(VM(v, p[low], c), VW(v, p[low]));
                rl[c].ss += a.ss; rl[c].s += a.s; rl[c].ws += a.ws;
            }
        } else {
            high -= 1;
            for (i64 c = 0; c < d; c++) {
                wsse_t a = wsse_one(VM(v, p[high], c), VW(v, p[high]));
                rh[c].ss += a.ss; rh[c].s += a.s; rh[c].ws += a.ws;
            }
        }
    }
    int moved_low = 0;
    while (low > 0) { /* :119 `low > 1` in 1-based */
        for (i64 c = 0; c < d; c++) {
            wsse_t a = wsse_one(VM(v, p[low], c), VW(v, p[low]));
            rlt[c].ss = rl[c].ss - a.ss; rlt[c].s = rl[c].s - a.s; rlt[c].ws = rl[c].ws - a.ws;
            rht[c].ss = rh[c].ss + a.ss; rht[c].s = rh[c].s + a.s; rht[c].ws = rh[c].ws + a.ws;
        }
        if (fmax(sum_wsse(rlt, d), sum_wsse(rht, d)) < fmax(sum_wsse(rl, d), sum_wsse(rh, d))) {
            moved_low = 1;
            low -= 1; high -= 1;
            memcpy(rl, rlt, sizeof(wsse_t) * (size_t)d);
            memcpy(rh, rht, sizeof(wsse_t) * (size_t)d);
        } else
            break;
    }
    if (!moved_low) {
        while (high < k - 1) { /* :133 `high < length(p)` in 1-based */
            for (i64 c = 0; c < d; c++) {
                wsse_t a = wsse_one(VM(v, p[high], c), VW(v, p[high]));
                rlt[c].ss = rl[c].ss + a.ss; rlt[c].s = rl[c].s + a.s; rlt[c].ws = rl[c].ws + a.ws;
                rht[c].ss = rh[c].ss - a.ss; rht[c].s = rh[c].s - a.s; rht[c].ws = rh[c].ws - a.ws;
            }
            if (fmax(sum_wsse(rlt, d), sum_wsse(rht, d)) < fmax(sum_wsse(rl, d), sum_wsse(rh, d))) {
                low += 1; high += 1;
                memcpy(rl, rlt, sizeof(wsse_t) * (size_t)d);
                memcpy(rh, rht, sizeof(wsse_t) * (size_t)d);
            } else
                break;
        }
    }
    for (i64 i = 0; i <= low; i++) out->low[out->nlow++] = p[i];
    for (i64 i = high; i < k; i++) out->high[out->nhigh++] = p[i];
done:
    free(z); free(p); free(rl); free(rh); free(rlt); free(rht);
    return rc;
}

/* split_cluster_size (:218-238) and split_cluster_diameter (:247-267) share
 * the tie rule; only the cut point differs.                                  */
static int rule_cut(const view_t *v, split_t *out, int use_median) {
    i64 k = v->k;
    if (use_median ? (k <= 1) : (v->d <= 1)) return ORC_E_ASSERT; /* :219 vs :248 */
    out->low = (i64 *)malloc(sizeof(i64) * (size_t)(k > 2 ? k : 2));
    out->high = (i64 *)malloc(sizeof(i64) * (size_t)(k > 2 ? k : 2));
    out->nlow = out->nhigh = 0;
    if (k == 2) { out->low[0] = 0; out->nlow = 1; out->high[0] = 1; out->nhigh = 1; return ORC_OK; }
    double *z = (double *)malloc(sizeof(double) * (size_t)k);
    double *scr = (double *)malloc(sizeof(double) * (size_t)k);
    int rc = pca_project(v, z);
    if (rc == ORC_OK) {
        double cut;
        if (use_median)
            cut = median_of(z, NULL, k, scr);
        else {
            double lo = z[0], hi = z[0];
            for (i64 j = 1; j < k; j++) { if (z[j] < lo) lo = z[j]; if (z[j] > hi) hi = z[j]; }
            cut = (lo + hi) / 2.0; /* mean(extrema(z)) */
        }
        for (i64 j = 0; j < k; j++) {
            if (z[j] == cut) {
                if (out->nlow < out->nhigh) out->low[out->nlow++] = j;
                else out->high[out->nhigh++] = j;
            } else if (z[j] < cut)
                out->low[out->nlow++] = j;
            else
                out->high[out->nhigh++] = j;
        }
    }
    free(z); free(scr);
    return rc;
}

static int apply_rule(int method, const view_t *v, split_t *out) {
    switch (method) {
    case 0: return rule_rss(v, out);
    case 1: return rule_rss2(v, out);
    case 2: return rule_cut(v, out, 1);
    case 3: return rule_cut(v, out, 0);
    }
    return ORC_E_ASSERT;
}

/* Pop one group, split it with `rule`, push both children (the body shared by
 * src/landmarks.jl:290-307 and :317-334).                                    */
static int split_top(heap_t *h, const double *emb, const double *w, i64 n, i64 d, int method) {
    lm_t g = heap_pop(h);
    view_t v = {emb, w, n, d, g.what, g.len};
    split_t sp = {0};
    int rc = apply_rule(method, &v, &sp);
    if (rc != ORC_OK) { free(sp.low); free(sp.high); free(g.what); return rc; }
    i64 *parts[2] = {sp.low, sp.high};
    i64 cnts[2] = {sp.nlow, sp.nhigh};
    for (int s = 0; s < 2 && rc == ORC_OK; s++) {
        i64 c = cnts[s];
        if (c == 0) { rc = ORC_E_EMPTY; break; }
        i64 *ids = (i64 *)malloc(sizeof(i64) * (size_t)c);
        for (i64 t = 0; t < c; t++) ids[t] = g.what[parts[s][t]];
        if (c > 1) {
            view_t cv = {emb, w, n, d, ids, c};
            rc = heap_put(h, ids, c, -total_rss(&cv));
        } else
            rc = heap_put(h, ids, c, DBL_EPSILON); /* eps() */
    }
    free(sp.low); free(sp.high); free(g.what);
    return rc;
}

typedef struct { const i64 *p; i64 len; } clus_t;
static int cmp_clus(const void *a, const void *b) { /* lexicographic isless on Vector{Int} */
    const clus_t *x = (const clus_t *)a, *y = (const clus_t *)b;
    i64 m = x->len < y->len ? x->len : y->len;
    for (i64 i = 0; i < m; i++) {
        if (x->p[i] < y->p[i]) return -1;
        if (x->p[i] > y->p[i]) return 1;
    }
    return (x->len > y->len) - (x->len < y->len);
}

/* runsplit: src/landmarks.jl:279-345.  group_ids out: 0-based, length n.     */
int orc_runsplit(const double *emb, const double *w, i64 n, i64 d, const i64 *cl_flat, const i64 *cl_off,
                 i64 ncl, i64 nland, i64 s, int method, i64 *group_ids) {
    heap_t H, L;
    int rc = heap_init(&H);
    if (rc) return rc;
    clus_t *cs = (clus_t *)malloc(sizeof(clus_t) * (size_t)(ncl > 0 ? ncl : 1));
    for (i64 c = 0; c < ncl; c++) { cs[c].p = cl_flat + cl_off[c]; cs[c].len = cl_off[c + 1] - cl_off[c]; }
    qsort(cs, (size_t)ncl, sizeof(clus_t), cmp_clus); /* sort(initial_clusters) :281 */
    for (i64 c = 0; c < ncl && rc == ORC_OK; c++) {
        if (cs[c].len <= s) {
            for (i64 t = 0; t < cs[c].len && rc == ORC_OK; t++) {
                i64 *one = (i64 *)malloc(sizeof(i64));
                one[0] = cs[c].p[t];
                rc = heap_put(&H, one, 1, DBL_EPSILON);
            }
        } else {
            rc = heap_init(&L);
            if (rc) break;
            i64 *all = (i64 *)malloc(sizeof(i64) * (size_t)cs[c].len);
            memcpy(all, cs[c].p, sizeof(i64) * (size_t)cs[c].len);
            view_t v = {emb, w, n, d, all, cs[c].len};
            rc = heap_put(&L, all, cs[c].len, -total_rss(&v));
            while (rc == ORC_OK && L.len < s) rc = split_top(&L, emb, w, n, d, method);
            while (rc == ORC_OK && L.len > 0) {
                lm_t g = heap_pop(&L);
                rc = heap_put(&H, g.what, g.len, g.value);
            }
            heap_free(&L);
        }
    }
    while (rc == ORC_OK && H.len < nland) rc = split_top(&H, emb, w, n, d, method);
    if (rc == ORC_OK) {
        for (i64 i = 0; i < n; i++) group_ids[i] = -1;
        for (i64 g = 1; g <= H.len; g++)
            for (i64 t = 0; t < H.a[g].len; t++) group_ids[H.a[g].what[t] - 1] = g - 1;
        for (i64 i = 0; i < n; i++)
            if (group_ids[i] < 0) rc = ORC_E_ASSERT; /* :343 */
    }
    heap_free(&H);
    free(cs);
    return rc;
}

/* count unique rows: `size(unique(embedding, dims=1), 1)`  src/landmarks.jl:371 */
static const double *g_emb; static i64 g_n, g_d;
static int cmp_rows(const void *a, const void *b) {
    i64 x = *(const i64 *)a, y = *(const i64 *)b;
    for (i64 c = 0; c < g_d; c++) {
        double u = g_emb[x + c * g_n], v = g_emb[y + c * g_n];
        if (u < v) return -1;
        if (u > v) return 1;
    }
    return 0;
}
i64 orc_unique_rows(const double *emb, i64 n, i64 d) {
    i64 *ix = (i64 *)malloc(sizeof(i64) * (size_t)n);
    for (i64 i = 0; i < n; i++) ix[i] = i;
    g_emb = emb; g_n = n; g_d = d;
    qsort(ix, (size_t)n, sizeof(i64), cmp_rows);
    i64 u = n > 0 ? 1 : 0;
    for (i64 i = 1; i < n; i++)
        if (cmp_rows(&ix[i - 1], &ix[i]) != 0) u++;
    free(ix);
    return u;
}

/* ------------------------------------------------------------------------- */
/* landmarks(): src/landmarks.jl:365-466.  Result object + getters.           */
typedef struct {
    i64 N, n, d, n_ledges;
    int truncated; /* the @warn at :374 fired */
    double *dii, *embed, *lweight, *lw_edges;
    i64 *cluster, *ledges, *v_to_l;
} orc_lm_result;

void orc_landmarks_free(orc_lm_result *r) {
    if (!r) return;
    free(r->dii); free(r->embed); free(r->lweight); free(r->lw_edges);
    free(r->cluster); free(r->ledges); free(r->v_to_l);
    free(r);
}

int orc_landmarks(const i64 *edges, i64 m, const double *weights, const double *vweights, i64 n,
                  const i64 *cl_flat, const i64 *cl_off, i64 ncl, const i64 *comm, const double *emb, i64 d,
                  i64 land, i64 forced, int method, int directed, orc_lm_result **out) {
    *out = NULL;
    orc_lm_result *r = (orc_lm_result *)calloc(1, sizeof(orc_lm_result));
    if (!r) return ORC_E_ALLOC;
    r->n = n; r->d = d;
    i64 uniq = orc_unique_rows(emb, n, d);
    if (land > uniq) { land = uniq; r->truncated = 1; } /* :373-376 */
    r->v_to_l = (i64 *)malloc(sizeof(i64) * (size_t)n);
    int rc = orc_runsplit(emb, vweights, n, d, cl_flat, cl_off, ncl, land, forced, method, r->v_to_l);
    if (rc != ORC_OK) { orc_landmarks_free(r); return rc; }
    i64 N = 0;
    for (i64 i = 0; i < n; i++) { r->v_to_l[i] += 1; if (r->v_to_l[i] > N) N = r->v_to_l[i]; } /* :379,:384 */
    r->N = N;
    r->embed = (double *)calloc((size_t)(N * d), sizeof(double)); /* N x d column-major */
    r->lweight = (double *)calloc((size_t)N, sizeof(double));
    r->dii = (double *)calloc((size_t)N, sizeof(double));
    r->cluster = (i64 *)calloc((size_t)N, sizeof(i64));
    for (i64 i = 0; i < n; i++) { /* :391-397 */
        i64 what = r->v_to_l[i] - 1;
        r->lweight[what] += vweights[i];
        for (i64 j = 0; j < d; j++) r->embed[what + j * N] += vweights[i] * emb[i + j * n];
    }
    for (i64 i = 0; i < N; i++)
        for (i64 j = 0; j < d; j++) r->embed[i + j * N] /= r->lweight[i]; /* :400-404 */
    for (i64 i = 0; i < n; i++) { /* :408-415 */
        i64 what = r->v_to_l[i] - 1;
        double dist = 0.0;
        for (i64 j = 0; j < d; j++) {
            double t = r->embed[what + j * N] - emb[i + j * n];
            dist += t * t;
        }
        r->dii[what] += dist;
    }
    for (i64 i = 0; i < N; i++)
        if (r->lweight[i] > 0) r->dii[i] = sqrt(r->dii[i] / r->lweight[i]); /* :418-423 */
    for (i64 i = 0; i < n; i++) r->cluster[r->v_to_l[i] - 1] = comm[i]; /* :427-429 */

    double *wedges = (double *)calloc((size_t)(N * N), sizeof(double)); /* [a-1 + (b-1)*N] */
    i64 cap = directed ? N * N : N * (N + 1) / 2;
    r->ledges = (i64 *)malloc(sizeof(i64) * (size_t)(2 * cap > 0 ? 2 * cap : 2));
    r->lw_edges = (double *)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    i64 *tmp_a = (i64 *)malloc(sizeof(i64) * (size_t)(cap > 0 ? cap : 1));
    i64 *tmp_b = (i64 *)malloc(sizeof(i64) * (size_t)(cap > 0 ? cap : 1));
    i64 ne = 0;
    for (i64 e = 0; e < m; e++) { /* :435-438 / :448-451 */
        i64 a = r->v_to_l[edges[e] - 1], b = r->v_to_l[edges[e + m] - 1];
        if (!directed && a > b) { i64 t = a; a = b; b = t; }
        wedges[(a - 1) + (b - 1) * N] += weights[e];
    }
    for (i64 i = 1; i <= N; i++) /* rows in N*(i-1)+j order (:441-446) or idx order (:454-459): same nesting */
        for (i64 j = directed ? 1 : i; j <= N; j++) {
            double wv = wedges[(i - 1) + (j - 1) * N];
            if (wv > 0) { tmp_a[ne] = i; tmp_b[ne] = j; r->lw_edges[ne] = wv; ne++; } /* :461 */
        }
    for (i64 e = 0; e < ne; e++) { r->ledges[e] = tmp_a[e]; r->ledges[e + ne] = tmp_b[e]; } /* ne x 2 column-major */
    r->n_ledges = ne;
    free(wedges); free(tmp_a); free(tmp_b);
    *out = r;
    return ORC_OK;
}

i64 orc_lm_N(const orc_lm_result *r) { return r->N; }
i64 orc_lm_n_ledges(const orc_lm_result *r) { return r->n_ledges; }
int orc_lm_truncated(const orc_lm_result *r) { return r->truncated; }
void orc_lm_get(const orc_lm_result *r, double *dii, double *embed, i64 *cluster, i64 *ledges, double *lw_edges,
                double *lweight, i64 *v_to_l) {
    memcpy(dii, r->dii, sizeof(double) * (size_t)r->N);
    memcpy(embed, r->embed, sizeof(double) * (size_t)(r->N * r->d));
    memcpy(cluster, r->cluster, sizeof(i64) * (size_t)r->N);
    memcpy(ledges, r->ledges, sizeof(i64) * (size_t)(2 * r->n_ledges));
    memcpy(lw_edges, r->lw_edges, sizeof(double) * (size_t)r->n_ledges);
    memcpy(lweight, r->lweight, sizeof(double) * (size_t)r->N);
    memcpy(v_to_l, r->v_to_l, sizeof(i64) * (size_t)r->n);
}

/* ------------------------------------------------------------------------- */
/* Point-set diameter = `hi` of extrema(full_graph_D) (src/divergence.jl:104-113;
 * lo is 0 because the diagonal entries stay 0).  Brute force, O(n^2 d).      */
double orc_max_pair_dist(const double *emb, i64 n, i64 d) {
    /* row-major copy for locality; arithmetic identical to orc_dist */
    double *x = (double *)malloc(sizeof(double) * (size_t)(n * d));
    for (i64 i = 0; i < n; i++)
        for (i64 k = 0; k < d; k++) x[i * d + k] = emb[i + k * n];
    double best = 0.0;
    for (i64 i = 0; i < n; i++)
        for (i64 j = i + 1; j < n; j++) {
            double s = 0.0;
            const double *a = x + i * d, *b = x + j * d;
            for (i64 k = 0; k < d; k++) { double t = a[k] - b[k]; s += t * t; }
            if (s > best) best = s;
        }
    free(x);
    return sqrt(best);
}

/* Full-size fixtures only (tests/golden/make_oracle_fixture_fullsize.py): at n = 10^6 the loop above is ~10^14
 * flop, so the fixture generator hands in the diameter instead -- found by an independent exact method (triangle-
 * inequality branch and bound over community pairs in numpy, tests/diameter_ref.py, the arg-max pair re-evaluated with
 * orc_dist's own arithmetic) that tests/test_oracle_golden.py checks against orc_max_pair_dist where the loop can run.
 * 0 (the default) = run the reference's loop.  Nothing else in the oracle changes.                                    */
static double orc_known_diameter = 0.0;
void orc_set_known_diameter(double hi) { orc_known_diameter = hi; }
static double orc_diameter(const double *emb, i64 n, i64 d) {
    return orc_known_diameter > 0.0 ? orc_known_diameter : orc_max_pair_dist(emb, n, d);
}

/* Sampled pairs for the local score.  The reference draws them with
 * Random.seed!/StatsBase.sample (src/divergence.jl:184-210); the stream cannot
 * be reproduced without Julia, so the oracle takes the draws as input:
 *   pos_idx[t*S + k] : 1-based row of the ORIGINAL edge list (E[k], :131-134)
 *   neg_i/neg_j[t*S + k] : a non-edge pair of original vertices (NE, :137)
 * n_sets = 1 (seeded: the same draw at every alpha) or >= number of alphas
 * (unseeded: a fresh draw per alpha).                                        */
typedef struct {
    i64 S, n_sets;
    const i64 *pos_idx, *neg_i, *neg_j;
    const i64 *pos_idx2; /* directed exact mode only: the second, overwriting draw (:510) */
} orc_samples;

/* trace of the alpha sweep, for debugging/parity tests (all optional)        */
typedef struct {
    i64 n_alpha;        /* alphas visited                                      */
    i64 iters[64];      /* Chung-Lu iterations per alpha                       */
    double div[64];     /* JS score per alpha (NaN when skipped)               */
    double auc[64];     /* 1-AUC per alpha (NaN when skipped)                  */
} orc_trace;

/* wGCL: src/divergence.jl:27-257                                             */
int orc_wgcl(const i64 *edges, i64 m, const double *eweights, const i64 *comm, i64 n_comm_rows, const double *embed,
             i64 embed_rows, i64 d, const double *distances, i64 n_dist, const double *vweights,
             const double *init_vweights, i64 n_init, const i64 *v_to_l, i64 n_vtol, const i64 *init_edges,
             i64 m_init, const double *init_eweights, const double *init_embed, int split,
             const orc_samples *smp, double *out7, orc_trace *tr) {
    const double epsilon = 0.25, delta = 0.001, AlphaMax = 10.0, AlphaStep = 0.25; /* :34-37 */
    int alpha_div_counter = 5, alpha_auc_counter = 5, skip_div = 0, skip_auc = 0;  /* :38-39 */
    i64 N = 0;
    for (i64 e = 0; e < 2 * m; e++) if (edges[e] > N) N = edges[e]; /* :41 */
    int landmarks = n_vtol > 0; /* :44 */
    if (n_comm_rows != N) return ORC_E_ASSERT; /* :50 */
    i64 C = 0;
    for (i64 i = 0; i < N; i++) if (comm[i] > C) C = comm[i]; /* :51 */
    i64 vlen = C * (C + 1) / 2;
    double *vC = (double *)calloc((size_t)vlen, sizeof(double)), *vB = (double *)calloc((size_t)vlen, sizeof(double));
    for (i64 e = 0; e < m; e++) { /* :59-63 */
        i64 a = comm[edges[e] - 1], b = comm[edges[e + m] - 1];
        i64 j = a < b ? a : b, k = a < b ? b : a;
        vC[orc_idx(C, j, k) - 1] += eweights[e];
    }
    uint8_t *vI = (uint8_t *)calloc((size_t)vlen, 1); /* :66-71 */
    { i64 j = 1; for (i64 i = 1; i <= C; i++) { vI[j - 1] = 1; j += (C - i + 1); } }
    double best_div = INFINITY, best_div_ext = INFINITY, best_div_int = INFINITY, best_auc_err = INFINITY,
           best_auc = INFINITY; /* typemax(Float64) :72 */
    double best_alpha = -1.0, best_alpha_auc = -1.0;
    i64 plen = N * (N + 1) / 2;
    if (n_dist != N) { free(vC); free(vB); free(vI); return ORC_E_ASSERT; } /* :81 */
    double *D = (double *)malloc(sizeof(double) * (size_t)plen);
    double *GD = (double *)malloc(sizeof(double) * (size_t)plen);
    double *P = (double *)malloc(sizeof(double) * (size_t)plen);
    double lo = INFINITY, hi = -INFINITY;
    for (i64 i = 1; i <= N; i++) /* :82-91 */
        for (i64 j = i; j <= N; j++) {
            double v = (i == j) ? distances[i - 1] : orc_dist(i, j, embed, embed_rows, d);
            D[orc_idx(N, i, j) - 1] = v;
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
    for (i64 k = 0; k < plen; k++) D[k] = (D[k] - lo) / (hi - lo); /* :93 */

    i64 adj_n = N, adj_m = m; /* :95-102 */
    const i64 *adj_edges = edges; const double *adj_ew = eweights;
    double fhi = 1.0;
    if (landmarks) {
        adj_n = n_init; adj_edges = init_edges; adj_ew = init_eweights; adj_m = m_init;
        fhi = orc_diameter(init_embed, adj_n, d); /* :104-114, lo == 0 */
    }
    double *T = (double *)malloc(sizeof(double) * (size_t)N), *S = (double *)malloc(sizeof(double) * (size_t)N);
    for (i64 i = 0; i < N; i++) T[i] = 1.0; /* :118 */
    i64 SS = smp ? smp->S : 0;
    double *pos = (double *)malloc(sizeof(double) * (size_t)(SS > 0 ? SS : 1));
    double *neg = (double *)malloc(sizeof(double) * (size_t)(SS > 0 ? SS : 1));
    double *aw = (double *)malloc(sizeof(double) * (size_t)(SS > 0 ? SS : 1));
    if (tr) tr->n_alpha = 0;

    i64 n_alpha_total = (i64)floor((AlphaMax + delta) / AlphaStep + 1e-9); /* 0.25:0.25:10.001 -> 40 */
    for (i64 ia = 1; ia <= n_alpha_total; ia++) {
        double alpha = AlphaStep * (double)ia; /* exact in binary for step 0.25 */
        for (i64 k = 0; k < plen; k++) GD[k] = pow(1.0 - D[k], alpha); /* :142-148 */
        double diff = 1.0; i64 iters = 0;
        while (diff > delta) { /* :150-168 */
            for (i64 i = 0; i < N; i++) S[i] = 0.0;
            for (i64 i = 1; i <= N; i++)
                for (i64 j = i; j <= N; j++) {
                    double tmp = T[i - 1] * T[j - 1] * GD[orc_idx(N, i, j) - 1];
                    S[i - 1] += tmp;
                    if (i != j) S[j - 1] += tmp;
                }
            double f = 0.0;
            for (i64 i = 0; i < N; i++) {
                double move = epsilon * T[i] * (vweights[i] / S[i] - 1.0);
                T[i] += move;
                double a = fabs(vweights[i] - S[i]);
                if (a > f) f = a; /* max(f, NaN) subtleties ignored: inputs are finite */
            }
            diff = f; iters++;
        }
        for (i64 i = 1; i <= N; i++) /* :170-176 */
            for (i64 j = i; j <= N; j++) {
                i64 k = orc_idx(N, i, j) - 1;
                P[k] = T[i - 1] * T[j - 1] * GD[k];
            }
        double auc_val = NAN, div_val = NAN;
        if (!skip_auc) { /* :178-224 */
            i64 set = (smp->n_sets == 1) ? 0 : (ia - 1);
            const i64 *pidx = smp->pos_idx + set * SS, *ni = smp->neg_i + set * SS, *nj = smp->neg_j + set * SS;
            for (i64 k = 0; k < SS; k++) {
                i64 e = pidx[k] - 1;
                i64 a = adj_edges[e], b = adj_edges[e + adj_m];
                i64 i = a < b ? a : b, j = a < b ? b : a; /* E tuple :133 */
                i64 u = ni[k] < nj[k] ? ni[k] : nj[k], v = ni[k] < nj[k] ? nj[k] : ni[k];
                if (landmarks) { /* :184-199 */
                    double Ti = T[v_to_l[i - 1] - 1] * init_vweights[i - 1] / vweights[v_to_l[i - 1] - 1];
                    double Tj = T[v_to_l[j - 1] - 1] * init_vweights[j - 1] / vweights[v_to_l[j - 1] - 1];
                    double dd = orc_dist(i, j, init_embed, adj_n, d) / fhi;
                    pos[k] = Ti * Tj * pow(1.0 - dd, alpha);
                    double Tu = T[v_to_l[u - 1] - 1] * init_vweights[u - 1] / vweights[v_to_l[u - 1] - 1];
                    double Tv = T[v_to_l[v - 1] - 1] * init_vweights[v - 1] / vweights[v_to_l[v - 1] - 1];
                    double dn = orc_dist(u, v, init_embed, adj_n, d) / fhi;
                    neg[k] = Tu * Tv * pow(1.0 - dn, alpha);
                } else { /* :201-210 */
                    pos[k] = P[orc_idx(N, i, j) - 1];
                    neg[k] = P[orc_idx(N, u, v) - 1];
                }
                aw[k] = adj_ew[e];
            }
            double num = 0.0, den = 0.0; /* :213 */
            for (i64 k = 0; k < SS; k++) { num += (pos[k] > neg[k] ? 1.0 : 0.0) * aw[k]; den += aw[k]; }
            double auc = 1.0 - num / den;
            auc_val = auc;
            if (auc < best_auc) {
                best_auc = auc;
                best_auc_err = 1.96 * sqrt(auc * (1.0 - auc) / (double)SS); /* :217 */
                best_alpha_auc = alpha;
                alpha_auc_counter = 5;
            } else {
                alpha_auc_counter -= 1;
                skip_auc = (alpha_auc_counter == 0);
            }
        }
        if (!skip_div) { /* :226-252 */
            memset(vB, 0, sizeof(double) * (size_t)vlen);
            for (i64 i = 1; i <= N; i++)
                for (i64 j = i; j <= N; j++) {
                    i64 a = comm[i - 1], b = comm[j - 1];
                    i64 k = a < b ? a : b, l = a < b ? b : a;
                    vB[orc_idx(C, k, l) - 1] += P[orc_idx(N, i, j) - 1];
                }
            double f, div_int = 0.0, div_ext = 0.0;
            if (!split)
                f = orc_js(vC, vB, vlen, NULL, 1);
            else {
                div_int = orc_js(vC, vB, vlen, vI, 1);
                div_ext = orc_js(vC, vB, vlen, vI, 0);
                f = (div_int + div_ext) / 2.0;
            }
            div_val = f;
            if (f < best_div) {
                best_div = f; best_alpha = alpha;
                best_div_ext = !split ? 0.0 : div_ext;
                best_div_int = !split ? 0.0 : div_int;
                alpha_div_counter = 5;
            } else {
                alpha_div_counter -= 1;
                skip_div = (alpha_div_counter == 0);
            }
        }
        if (tr && tr->n_alpha < 64) {
            tr->iters[tr->n_alpha] = iters; tr->div[tr->n_alpha] = div_val; tr->auc[tr->n_alpha] = auc_val;
            tr->n_alpha++;
        }
        if (skip_div && skip_auc) break; /* :253 */
    }
    out7[0] = best_alpha; out7[1] = best_div; out7[2] = best_div_ext; out7[3] = best_div_int;
    out7[4] = best_alpha_auc; out7[5] = best_auc; out7[6] = best_auc_err; /* :256 */
    free(vC); free(vB); free(vI); free(D); free(GD); free(P); free(T); free(S); free(pos); free(neg); free(aw);
    return ORC_OK;
}

/* wGCL_directed: src/divergence.jl:282-561.  Returns out_len = 6 for the
 * star-graph early return (:332-334), else 7.                                */
int orc_wgcl_directed(const i64 *edges, i64 m, const double *eweights, const i64 *comm, i64 n_comm_rows,
                      const double *embed, i64 embed_rows, i64 d, const double *distances, i64 n_dist,
                      const double *vweights, const double *init_vweights, i64 n_init, const i64 *v_to_l,
                      i64 n_vtol, const i64 *init_edges, i64 m_init, const double *init_eweights,
                      const double *init_embed, int split, const orc_samples *smp, double *out7, int *out_len,
                      orc_trace *tr) {
    const double delta = 0.001, AlphaMax = 10.0, AlphaStep = 0.25; /* :288-290 */
    int alpha_div_counter = 5, alpha_auc_counter = 5, skip_div = 0, skip_auc = 0;
    i64 N = 0;
    for (i64 e = 0; e < 2 * m; e++) if (edges[e] > N) N = edges[e]; /* :294 */
    int landmarks = n_vtol > 0;
    if (n_comm_rows != N) return ORC_E_ASSERT; /* :303 */
    i64 C = 0;
    for (i64 i = 0; i < N; i++) if (comm[i] > C) C = comm[i];
    double *deg_in = (double *)calloc((size_t)N, sizeof(double)), *deg_out = (double *)calloc((size_t)N, sizeof(double));
    i64 *star = (i64 *)calloc((size_t)N, sizeof(i64));
    for (i64 e = 0; e < m; e++) { /* :311-319 */
        i64 v1 = edges[e], v2 = edges[e + m];
        deg_out[v1 - 1] += eweights[e]; deg_in[v2 - 1] += eweights[e];
        star[v1 - 1] += 1; star[v2 - 1] += 1;
    }
    { /* :322-334 */
        int has_nm1 = 0, has_2nm1 = 0; i64 sum = 0, cnt2 = 0;
        for (i64 i = 0; i < N; i++) {
            if (star[i] == N - 1) has_nm1 = 1;
            if (star[i] == 2 * (N - 1)) has_2nm1 = 1;
            sum += star[i];
            if (star[i] == 2) cnt2++;
        }
        int is_star = (has_nm1 && sum == 2 * (N - 1)) || (has_2nm1 && cnt2 == N - 1);
        if (is_star) {
            out7[0] = -1.0; for (int k = 1; k < 6; k++) out7[k] = 0.0;
            *out_len = 6;
            free(deg_in); free(deg_out); free(star);
            return ORC_OK;
        }
    }
    free(star);
    *out_len = 7;
    i64 vlen = C * C;
    double *vC = (double *)calloc((size_t)vlen, sizeof(double)), *vB = (double *)calloc((size_t)vlen, sizeof(double));
    for (i64 e = 0; e < m; e++) /* :341-345 */
        vC[(comm[edges[e] - 1] - 1) * C + comm[edges[e + m] - 1] - 1] += eweights[e];
    uint8_t *vI = (uint8_t *)calloc((size_t)vlen, 1);
    for (i64 i = 0; i < vlen; i += C + 1) vI[i] = 1; /* :348-351 */
    double best_div = INFINITY, best_div_ext = INFINITY, best_div_int = INFINITY, best_auc_err = INFINITY,
           best_auc = INFINITY;
    double best_alpha = -1.0, best_alpha_auc = -1.0;
    i64 plen = N * (N + 1) / 2;
    if (n_dist != N) { free(vC); free(vB); free(vI); free(deg_in); free(deg_out); return ORC_E_ASSERT; } /* :363 */
    double *D = (double *)malloc(sizeof(double) * (size_t)plen);
    double *GD = (double *)malloc(sizeof(double) * (size_t)plen);
    double *P = (double *)malloc(sizeof(double) * (size_t)(N * N));
    double lo = INFINITY, hi = -INFINITY;
    for (i64 i = 1; i <= N; i++) /* :364-373 */
        for (i64 j = i; j <= N; j++) {
            double v = (i == j) ? distances[i - 1] : orc_dist(i, j, embed, embed_rows, d);
            D[orc_idx(N, i, j) - 1] = v;
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
    for (i64 k = 0; k < plen; k++) D[k] = (D[k] - lo) / (hi - lo);
    i64 adj_n = N, adj_m = m;
    const i64 *adj_edges = edges; const double *adj_ew = eweights;
    double fhi = 1.0;
    if (landmarks) {
        adj_n = n_init; adj_edges = init_edges; adj_ew = init_eweights; adj_m = m_init;
        fhi = orc_diameter(init_embed, adj_n, d); /* :386-397 */
    }
    double *Tin = (double *)malloc(sizeof(double) * (size_t)N), *Tout = (double *)malloc(sizeof(double) * (size_t)N);
    double *Sin = (double *)malloc(sizeof(double) * (size_t)N), *Sout = (double *)malloc(sizeof(double) * (size_t)N);
    for (i64 i = 0; i < N; i++) { Tin[i] = deg_in[i] == 0 ? 0.0 : 1.0; Tout[i] = deg_out[i] == 0 ? 0.0 : 1.0; } /* :399-402 */
    i64 SS = smp ? smp->S : 0;
    double *pos = (double *)malloc(sizeof(double) * (size_t)(SS > 0 ? SS : 1));
    double *neg = (double *)malloc(sizeof(double) * (size_t)(SS > 0 ? SS : 1));
    double *aw = (double *)malloc(sizeof(double) * (size_t)(SS > 0 ? SS : 1));
    if (tr) tr->n_alpha = 0;
    i64 n_alpha_total = (i64)floor((AlphaMax + delta) / AlphaStep + 1e-9);
    for (i64 ia = 1; ia <= n_alpha_total; ia++) {
        double alpha = AlphaStep * (double)ia;
        for (i64 k = 0; k < plen; k++) GD[k] = pow(1.0 - D[k], alpha); /* :426-432 */
        double diff = 1.0, epsilon = 0.9; i64 iters = 0; /* :434-435 */
        while (diff > delta) { /* :436-467 */
            for (i64 i = 0; i < N; i++) { Sin[i] = 0.0; Sout[i] = 0.0; }
            for (i64 i = 1; i <= N; i++)
                for (i64 j = i; j <= N; j++) { /* j == i included: diagonal counted twice (:439-449) */
                    double g = GD[orc_idx(N, i, j) - 1];
                    double tmp1 = Tin[i - 1] * Tout[j - 1] * g;
                    double tmp2 = Tin[j - 1] * Tout[i - 1] * g;
                    Sin[i - 1] += tmp1; Sin[j - 1] += tmp2;
                    Sout[i - 1] += tmp2; Sout[j - 1] += tmp1;
                }
            double f = 0.0;
            for (i64 i = 0; i < N; i++) {
                if (deg_in[i] > 0) {
                    Tin[i] += epsilon * Tin[i] * (deg_in[i] / Sin[i] - 1.0);
                    double a = fabs(deg_in[i] - Sin[i]); if (a > f) f = a;
                }
                if (deg_out[i] > 0) {
                    Tout[i] += epsilon * Tout[i] * (deg_out[i] / Sout[i] - 1.0);
                    double a = fabs(deg_out[i] - Sout[i]); if (a > f) f = a;
                }
            }
            if (f > diff) epsilon *= 0.99; /* :462-464 */
            diff = f; iters++;
        }
        for (i64 i = 1; i <= N; i++) /* :470-476 */
            for (i64 j = 1; j <= N; j++) {
                i64 a = i < j ? i : j, b = i < j ? j : i;
                P[N * (i - 1) + j - 1] = Tout[i - 1] * Tin[j - 1] * GD[orc_idx(N, a, b) - 1];
            }
        double auc_val = NAN, div_val = NAN;
        if (!skip_auc) { /* :478-528 */
            i64 set = (smp->n_sets == 1) ? 0 : (ia - 1);
            const i64 *pidx = smp->pos_idx + set * SS, *ni = smp->neg_i + set * SS, *nj = smp->neg_j + set * SS;
            const i64 *pidx2 = smp->pos_idx2 ? smp->pos_idx2 + set * SS : pidx;
            for (i64 k = 0; k < SS; k++) {
                i64 e = pidx[k] - 1;
                i64 i = adj_edges[e], j = adj_edges[e + adj_m]; /* orientation kept (:415-418) */
                i64 u = ni[k], v = nj[k];
                if (landmarks) { /* :482-501 */
                    i64 a = i < j ? i : j, b = i < j ? j : i;
                    double To = Tout[v_to_l[i - 1] - 1] * init_vweights[i - 1] / vweights[v_to_l[i - 1] - 1];
                    double Ti = Tin[v_to_l[j - 1] - 1] * init_vweights[j - 1] / vweights[v_to_l[j - 1] - 1];
                    pos[k] = To * Ti * pow(1.0 - orc_dist(a, b, init_embed, adj_n, d) / fhi, alpha);
                    double Uo = Tout[v_to_l[u - 1] - 1] * init_vweights[u - 1] / vweights[v_to_l[u - 1] - 1];
                    double Vi = Tin[v_to_l[v - 1] - 1] * init_vweights[v - 1] / vweights[v_to_l[v - 1] - 1];
                    i64 a2 = u < v ? u : v, b2 = u < v ? v : u;
                    neg[k] = Uo * Vi * pow(1.0 - orc_dist(a2, b2, init_embed, adj_n, d) / fhi, alpha);
                } else { /* :503-513: pos is overwritten by a second draw (:510), weights keep the first */
                    i64 e2 = pidx2[k] - 1;
                    i64 i2 = adj_edges[e2], j2 = adj_edges[e2 + adj_m];
                    pos[k] = P[N * (i2 - 1) + j2 - 1];
                    neg[k] = P[N * (u - 1) + v - 1];
                }
                aw[k] = adj_ew[e];
            }
            double num = 0.0, den = 0.0;
            for (i64 k = 0; k < SS; k++) { num += (pos[k] > neg[k] ? 1.0 : 0.0) * aw[k]; den += aw[k]; }
            double auc = 1.0 - num / den; /* :517 */
            auc_val = auc;
            if (auc < best_auc) {
                best_auc = auc;
                best_auc_err = 1.96 * sqrt(auc * (1.0 - auc) / (double)SS);
                best_alpha_auc = alpha;
                alpha_auc_counter = 5;
            } else {
                alpha_auc_counter -= 1;
                skip_auc = (alpha_auc_counter == 0);
            }
        }
        if (!skip_div) { /* :530-556 */
            memset(vB, 0, sizeof(double) * (size_t)vlen);
            for (i64 i = 1; i <= N; i++)
                for (i64 j = 1; j <= N; j++)
                    vB[(comm[i - 1] - 1) * C + comm[j - 1] - 1] += P[N * (i - 1) + j - 1];
            double f, div_int = 0.0, div_ext = 0.0;
            if (!split)
                f = orc_js(vC, vB, vlen, NULL, 1);
            else {
                div_int = orc_js(vC, vB, vlen, vI, 1);
                div_ext = orc_js(vC, vB, vlen, vI, 0);
                f = (div_int + div_ext) / 2.0;
            }
            div_val = f;
            if (f < best_div) {
                best_div = f; best_alpha = alpha;
                best_div_ext = !split ? 0.0 : div_ext;
                best_div_int = !split ? 0.0 : div_int;
                alpha_div_counter = 5;
            } else {
                alpha_div_counter -= 1;
                skip_div = (alpha_div_counter == 0);
            }
        }
        if (tr && tr->n_alpha < 64) {
            tr->iters[tr->n_alpha] = iters; tr->div[tr->n_alpha] = div_val; tr->auc[tr->n_alpha] = auc_val;
            tr->n_alpha++;
        }
        if (skip_div && skip_auc) break;
    }
    out7[0] = best_alpha; out7[1] = best_div; out7[2] = best_div_ext; out7[3] = best_div_int;
    out7[4] = best_alpha_auc; out7[5] = best_auc; out7[6] = best_auc_err;
    free(vC); free(vB); free(vI); free(D); free(GD); free(P); free(Tin); free(Tout); free(Sin); free(Sout);
    free(deg_in); free(deg_out); free(pos); free(neg); free(aw);
    return ORC_OK;
}

/* Exposed pieces for unit tests of the restatement itself.                   */
int orc_split(const double *emb, const double *w, i64 n, i64 d, const i64 *idxs, i64 k, int method, i64 *low,
              i64 *nlow, i64 *high, i64 *nhigh) {
    view_t v = {emb, w, n, d, idxs, k};
    split_t sp = {0};
    int rc = apply_rule(method, &v, &sp);
    if (rc == ORC_OK) {
        memcpy(low, sp.low, sizeof(i64) * (size_t)sp.nlow); *nlow = sp.nlow;
        memcpy(high, sp.high, sizeof(i64) * (size_t)sp.nhigh); *nhigh = sp.nhigh;
    }
    free(sp.low); free(sp.high);
    return rc;
}
double orc_total_rss(const double *emb, const double *w, i64 n, i64 d, const i64 *idxs, i64 k) {
    view_t v = {emb, w, n, d, idxs, k};
    return total_rss(&v);
}
int orc_eig_top(const double *A, i64 d, double *v) {
    double *B = (double *)malloc(sizeof(double) * (size_t)(d * d));
    memcpy(B, A, sizeof(double) * (size_t)(d * d));
    int rc = eig_top(B, d, v);
    free(B);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Louvain, level 1 -- what `louvain_clust` (src/clustering.jl:14-68) writes to <file>.ecg.
 * The reference shells out to three executables of louvain_jll (no version pin: Project.toml has no [compat]; the
 * package wraps the "generic Louvain" code of Blondel, Guillaume, Lambiotte, Lefebvre, v0.3): `convert` (edge list ->
 * binary graph, every edge in both directions), `louvain -l -1 -q 0` (all levels, modularity), `hierarchy -l 1` (the
 * partition after the FIRST pass of local moving).  This is a restatement of that published algorithm (one_level() of
 * louvain.cpp): nodes in turn are taken out of their community and put into the neighbouring community of largest gain
 * dnc - tot[c] * k / m2 (own community first, so ties stay), passes repeat while nodes move and the modularity improves
 * by more than 1e-6; communities are then renumbered 0.. in ascending order of their old id.  UNPINNED: the executable
 * visits the nodes in a rand()-shuffled order seeded from time and pid, so no two runs of the reference agree; here the
 * order is 0..n-1.  edges: m x 2 column-major, 1-based; weights may be NULL (unit).  comm_out: n entries, 0-based.   */
int orc_louvain_level1(const i64 *edges, const double *weights, i64 m, i64 n, i64 *comm_out, i64 *n_comm, double *quality) {
    i64 *deg = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    for (i64 e = 0; e < m; e++) {
        const i64 u = edges[e] - 1, v = edges[e + m] - 1;
        deg[u + 1]++;
        if (u != v) deg[v + 1]++;
    }
    for (i64 i = 0; i < n; i++) deg[i + 1] += deg[i];
    i64 *adj = (i64 *)malloc(sizeof(i64) * (size_t)(deg[n] > 0 ? deg[n] : 1));
    double *aw = (double *)malloc(sizeof(double) * (size_t)(deg[n] > 0 ? deg[n] : 1));
    i64 *cur = (i64 *)malloc(sizeof(i64) * (size_t)n);
    for (i64 i = 0; i < n; i++) cur[i] = deg[i];
    for (i64 e = 0; e < m; e++) {
        const i64 u = edges[e] - 1, v = edges[e + m] - 1;
        const double w = weights ? weights[e] : 1.0;
        adj[cur[u]] = v; aw[cur[u]++] = w;
        if (u != v) { adj[cur[v]] = u; aw[cur[v]++] = w; }
    }
    double *k = (double *)calloc((size_t)n, sizeof(double)), *tot = (double *)malloc(sizeof(double) * (size_t)n);
    double *in = (double *)calloc((size_t)n, sizeof(double)), *nw = (double *)malloc(sizeof(double) * (size_t)n);
    i64 *n2c = (i64 *)malloc(sizeof(i64) * (size_t)n), *npos = (i64 *)malloc(sizeof(i64) * (size_t)n);
    double m2 = 0.0;
    for (i64 i = 0; i < n; i++) {
        for (i64 q = deg[i]; q < deg[i + 1]; q++) { k[i] += aw[q]; if (adj[q] == i) in[i] += aw[q]; }
        m2 += k[i];
        tot[i] = k[i];
        n2c[i] = i;
        nw[i] = -1.0;
    }
    double cur_q = 0.0, new_q = 0.0;
    for (i64 c = 0; c < n; c++) if (m2 > 0 && tot[c] > 0) new_q += in[c] / m2 - (tot[c] / m2) * (tot[c] / m2);
    i64 moves;
    do {
        cur_q = new_q;
        moves = 0;
        for (i64 node = 0; node < n; node++) {
            const i64 own = n2c[node];
            i64 nlast = 1;
            npos[0] = own;
            nw[own] = 0.0;
            double self = 0.0;
            for (i64 q = deg[node]; q < deg[node + 1]; q++) {
                const i64 v = adj[q];
                if (v == node) { self += aw[q]; continue; }
                const i64 c = n2c[v];
                if (nw[c] == -1.0) { nw[c] = 0.0; npos[nlast++] = c; }
                nw[c] += aw[q];
            }
            tot[own] -= k[node];
            in[own] -= 2.0 * nw[own] + self;
            i64 best = own;
            double best_w = nw[own], best_inc = m2 > 0 ? nw[own] - tot[own] * k[node] / m2 : 0.0;
            for (i64 q = 1; q < nlast; q++) {
                const i64 c = npos[q];
                const double inc = m2 > 0 ? nw[c] - tot[c] * k[node] / m2 : 0.0;
                if (inc > best_inc) { best = c; best_w = nw[c]; best_inc = inc; }
            }
            tot[best] += k[node];
            in[best] += 2.0 * best_w + self;
            n2c[node] = best;
            if (best != own) moves++;
            for (i64 q = 0; q < nlast; q++) nw[npos[q]] = -1.0;
        }
        new_q = 0.0;
        for (i64 c = 0; c < n; c++) if (m2 > 0 && tot[c] > 0) new_q += in[c] / m2 - (tot[c] / m2) * (tot[c] / m2);
    } while (moves > 0 && new_q - cur_q > 1e-6);
    for (i64 i = 0; i < n; i++) npos[i] = -1;
    for (i64 i = 0; i < n; i++) npos[n2c[i]] = 0;
    i64 nc = 0;
    for (i64 c = 0; c < n; c++) if (npos[c] == 0) npos[c] = nc++;
    for (i64 i = 0; i < n; i++) comm_out[i] = npos[n2c[i]];
    if (n_comm) *n_comm = nc;
    if (quality) *quality = new_q;
    free(deg); free(adj); free(aw); free(cur); free(k); free(tot); free(in); free(nw); free(n2c); free(npos);
    return ORC_OK;
}
