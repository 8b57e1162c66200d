// landmarks_host.cpp -- runsplit() for the MI355X build (reference: src/landmarks.jl:279-345).
//
// Split of work:
//   device : the member lists (an int32 arena: a group is a range, a split appends its two children), per-group
//            weighted mean, covariance y'y (fp64 MFMA SYRK), principal eigenvector (register-resident batched
//            solver), projection z = y v, stable segmented sort of z, the cut of all four rules (rss: WSSE prefix
//            sums along sorted z + all median-cut rounds in one launch; rss2: two-pointer walk; size / diameter:
//            side flags with the sequential tie rule), children values / means / member lists, and finally v2l and
//            the landmark -> members index -- batched over many groups per launch (kernels_lm.hip, kernels_sort.hip);
//   host   : the heap (bit-for-bit the reference's sift rules, :12-46) and its replay from (offset, length, value)
//            triples, and the generic round-based rss path for groups with a tie at max z or NaNs.
// The reference splits strictly one group at a time.  Here the nodes of the split tree that can still be
// popped are split speculatively in batches and the heap is then REPLAYED in the reference's order from the
// cached results, so ids (= positions in the heap array, :337-342) come out identical while the device sees
// hundreds of groups per launch.  host_eig_top below is the d > 512 fallback of the device eigen-solvers.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <deque>
#include <memory>
#include <thread>
#include <unordered_map>

#include "common.hpp"

namespace {

// accumulates wall time of a scope into ctx->phases (reported by cge_phase_ms)
struct PhaseAcc {
    cge_ctx *c;
    const char *name;
    double t0;
    PhaseAcc(cge_ctx *ctx, const char *n) : c(ctx), name(n), t0(now_ms()) {}
    ~PhaseAcc() { c->phases.ms[name] += now_ms() - t0; }
};

// A group of the split tree.  Its members live on the device: the range [off, off + len) of the member arena
// (c->lm_arena: 0-based vertex ids in the reference's order).  A split writes the two children behind each other into
// a fresh range [coff, coff + len): low first (nlow entries), then high.
struct Group {
    i64 off = -1, len = 0;
    std::vector<i64> what; // host copy (1-based ids), fetched only for the generic round-based rss path
    double value = 0.0;    // heap key: -total_rss, or eps() for singletons
    bool has_split = false;
    int rc = CGE_OK;
    i64 coff = -1, nlow = 0; // children ranges until the child groups are materialised
    double vlow = 0.0, vhigh = 0.0;
    Group *clo = nullptr, *chi = nullptr; // child groups, created as soon as the split is known
    // weighted mean of the rows (matrix_w_mean, src/landmarks.jl:71-81) when it is already known from the parent's
    // split (the WSSE column sums of a child are sum w x and sum w): d doubles at this offset of the means arena
    // (c->lm_means), -1 = compute it on the device.  cmean_off: the two children's means, low then high.
    i64 mean_off = -1, cmean_off = -1;
    Group *parent = nullptr; // (children of a cached split)
    int owner = 0;           // option shard_rows: the rank that holds this group's rows (off / coff / mean_off are -1 elsewhere)
};

// The groups of one runsplit call: stable addresses, handed out from chunks of 4096 (a std::deque<Group> takes one 512-byte
// allocation per three groups: ~3000 malloc / free pairs per call at the headline, 0.25 ms of the landmark phase).
struct GroupPool {
    static constexpr size_t CH = 4096;
    std::vector<Group *> chunks;
    size_t used = 0; // groups handed out
    GroupPool() {}
    GroupPool(const GroupPool &) = delete;
    GroupPool &operator=(const GroupPool &) = delete;
    ~GroupPool() {
        for (size_t i = 0; i < used; i++) chunks[i / CH][i % CH].~Group();
        for (Group *p : chunks) ::operator delete(p);
    }
    void emplace_back() {
        if (used == chunks.size() * CH) chunks.push_back(static_cast<Group *>(::operator new(sizeof(Group) * CH)));
        new (&chunks[used / CH][used % CH]) Group();
        used++;
    }
    Group &back() { return chunks[(used - 1) / CH][(used - 1) % CH]; }
};

inline cge_ctx *root_of(cge_ctx *x) { return x; } // (rounds 3-4 ran half batches on shadow contexts of a root; removed in round 5)
// Growth copies the arena into a larger allocation: everything that may still read or write the old one has to be done.
template <typename T>
void arena_grow(cge_ctx *c, DevBuf<T> &buf, i64 used, i64 need) {
    const i64 cap = std::max<i64>(need + need / 2, 1024);
    T *fresh = nullptr;
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipMalloc((void **)&fresh, (size_t)cap * sizeof(T)));
    if (buf.p && used > 0) HIP_CHECK(hipMemcpyAsync(fresh, buf.p, sizeof(T) * used, hipMemcpyDeviceToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (buf.p) (void)hipFree(buf.p);
    buf.p = fresh;
    buf.n = (size_t)cap;
}
// reserve `cnt` entries of the member arena (ranges handed out earlier stay valid as offsets)
i64 arena_alloc(cge_ctx *x, i64 cnt) {
    cge_ctx *c = root_of(x);
    const i64 need = c->lm_arena_used + cnt;
    if ((i64)c->lm_arena.n < need || !c->lm_arena.p) arena_grow(c, c->lm_arena, c->lm_arena_used, need);
    const i64 at = c->lm_arena_used;
    c->lm_arena_used = need;
    return at;
}
// reserve `cnt` doubles of the means arena (same growth rule as the member arena)
i64 means_alloc(cge_ctx *x, i64 cnt) {
    cge_ctx *c = root_of(x);
    const i64 need = c->lm_means_used + cnt;
    if ((i64)c->lm_means.n < need || !c->lm_means.p) arena_grow(c, c->lm_means, c->lm_means_used, need);
    const i64 at = c->lm_means_used;
    c->lm_means_used = need;
    return at;
}

// 1-based binary min-heap on value with the reference's exact sift rules (src/landmarks.jl:12-46)
struct Heap {
    struct Ent { // the key sits beside the pointer: sifting compares neighbours of one array instead of chasing a pointer per level
        double value;
        Group *g;
    };
    std::vector<Ent> a{Ent{0.0, nullptr}};
    size_t len() const { return a.size() - 1; }
    Group *at(size_t i) const { return a[i].g; } // 1-based position = landmark id
    void put(Group *g) {
        const Ent e{g->value, g};
        a.push_back(e);
        size_t i = a.size() - 1, j;
        while ((j = i / 2) >= 1) {
            if (e.value < a[j].value) {
                a[i] = a[j];
                i = j;
            } else
                break;
        }
        a[i] = e;
    }
    Group *pop() {
        Group *x = a[1].g;
        const Ent y = a.back();
        a.pop_back();
        const size_t n = a.size() - 1;
        if (n > 0) {
            size_t i = 1, l;
            while ((l = 2 * i) <= n) {
                const size_t r = l + 1;
                const size_t j = (r > n || a[l].value < a[r].value) ? l : r;
                if (a[j].value < y.value) {
                    a[i] = a[j];
                    i = j;
                } else
                    break;
            }
            a[i] = y;
        }
        return x;
    }
    Group *top() const { return a[1].g; }
};

// all host-parallel loops go through the ctx's persistent pool
template <typename F>
void parallel_for(cge_ctx *c, i64 n, F &&fn) {
    if (n <= 0) return;
    if (!c->pool || n == 1) {
        for (i64 i = 0; i < n; i++) fn(i);
        return;
    }
    const std::function<void(i64)> f = std::forward<F>(fn);
    c->pool->run(n, f);
}

struct Wsse {
    double ss = 0, s = 0, ws = 0;
};
inline double wsse_val(const Wsse &x) { return x.ss - x.s * x.s / x.ws; }
inline double sum_wsse(const std::vector<Wsse> &r) {
    double t = 0.0;
    for (const auto &x : r) t += wsse_val(x);
    return t;
}


// Statistics.median: middle of the sorted values, even length -> x/2 + y/2
double median_sel(const double *z, const std::vector<i64> *sel, i64 k, std::vector<double> &scr) {
    const i64 cnt = sel ? (i64)sel->size() : k;
    scr.resize(cnt);
    for (i64 i = 0; i < cnt; i++) scr[i] = sel ? z[(*sel)[i]] : z[i];
    const i64 mid = cnt / 2;
    std::nth_element(scr.begin(), scr.begin() + mid, scr.end());
    const double hi = scr[mid];
    if (cnt & 1) return hi;
    const double lo = *std::max_element(scr.begin(), scr.begin() + mid);
    return lo / 2.0 + hi / 2.0;
}


} // namespace

// ------------------------------------------------------------------------------------------------
// Principal eigenvector of a symmetric d x d matrix: Householder tridiagonalisation, bisection for
// the largest eigenvalue, inverse iteration, back-transformation.  Replaces `eigvecs(A)[:, end]`
// (src/landmarks.jl:99 etc.).  Sign convention: the component of largest magnitude is positive.
void host_eig_top(const double *Ain, i64 d, double *vout) {
    if (d == 1) { vout[0] = 1.0; return; }
    std::vector<double> A(Ain, Ain + d * d);
    std::vector<double> diag(d), off(d, 0.0), beta(d, 0.0), p(d), vv(d);
    for (i64 k = 0; k + 2 < d; k++) {
        const i64 r = d - k - 1; // trailing size
        double *xcol = &vv[0];   // x = A[k+1.., k]
        for (i64 i = 0; i < r; i++) xcol[i] = A[(k + 1 + i) * d + k];
        const double alpha = xcol[0];
        double sigma = 0.0;
        for (i64 i = 1; i < r; i++) sigma += xcol[i] * xcol[i];
        if (sigma == 0.0) {
            beta[k] = 0.0;
            off[k] = alpha;
            continue;
        }
        const double mu = std::sqrt(alpha * alpha + sigma);
        const double v0 = (alpha <= 0.0) ? alpha - mu : -sigma / (alpha + mu);
        const double bk = 2.0 * v0 * v0 / (sigma + v0 * v0);
        xcol[0] = 1.0;
        for (i64 i = 1; i < r; i++) xcol[i] /= v0;
        beta[k] = bk;
        off[k] = mu;
        // p = beta * B v, B = A[k+1.., k+1..]
        for (i64 i = 0; i < r; i++) {
            const double *Bi = &A[(k + 1 + i) * d + (k + 1)];
            double s = 0.0;
            for (i64 j = 0; j < r; j++) s += Bi[j] * xcol[j];
            p[i] = bk * s;
        }
        double pv = 0.0;
        for (i64 i = 0; i < r; i++) pv += p[i] * xcol[i];
        const double K = 0.5 * bk * pv;
        for (i64 i = 0; i < r; i++) p[i] -= K * xcol[i]; // w
        for (i64 i = 0; i < r; i++) {
            double *Bi = &A[(k + 1 + i) * d + (k + 1)];
            const double vi = xcol[i], wi = p[i];
            for (i64 j = 0; j < r; j++) Bi[j] -= vi * p[j] + wi * xcol[j];
        }
        for (i64 i = 1; i < r; i++) A[(k + 1 + i) * d + k] = xcol[i]; // keep the reflector (v[0] = 1 implicit)
    }
    for (i64 i = 0; i < d; i++) diag[i] = A[i * d + i];
    off[d - 2] = A[(d - 1) * d + (d - 2)];
    // largest eigenvalue by bisection on the Sturm count
    double lo = diag[0], hi = diag[0], tnorm = 0.0;
    for (i64 i = 0; i < d; i++) {
        const double rad = (i > 0 ? std::fabs(off[i - 1]) : 0.0) + (i + 1 < d ? std::fabs(off[i]) : 0.0);
        lo = std::min(lo, diag[i] - rad);
        hi = std::max(hi, diag[i] + rad);
        tnorm = std::max(tnorm, std::fabs(diag[i]) + rad);
    }
    const double tiny = std::max(tnorm, DBL_MIN) * DBL_EPSILON;
    auto count_below = [&](double x) { // number of eigenvalues < x
        i64 cnt = 0;
        double q = diag[0] - x;
        if (q < 0) cnt++;
        for (i64 i = 1; i < d; i++) {
            if (q == 0.0) q = tiny;
            q = diag[i] - x - off[i - 1] * off[i - 1] / q;
            if (q < 0) cnt++;
        }
        return cnt;
    };
    hi += tiny;
    for (int it = 0; it < 200; it++) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) break;
        if (count_below(mid) >= d) hi = mid; else lo = mid;
    }
    const double lam = 0.5 * (lo + hi);
    // inverse iteration on (T - lam I), tridiagonal LU with partial pivoting
    std::vector<double> dl(d), dd(d), du(d), du2(d, 0.0), y(d);
    std::vector<char> swp(d, 0);
    for (i64 i = 0; i < d; i++) {
        dd[i] = diag[i] - lam;
        dl[i] = (i + 1 < d) ? off[i] : 0.0; // sub-diagonal below row i
        du[i] = (i + 1 < d) ? off[i] : 0.0; // super-diagonal right of row i
    }
    for (i64 i = 0; i + 1 < d; i++) {
        if (std::fabs(dd[i]) >= std::fabs(dl[i])) {
            if (dd[i] == 0.0) dd[i] = tiny;
            const double f = dl[i] / dd[i];
            dl[i] = f;
            dd[i + 1] -= f * du[i];
            du2[i] = 0.0;
        } else {
            const double f = dd[i] / dl[i];
            dd[i] = dl[i];
            dl[i] = f;
            const double t = du[i];
            du[i] = dd[i + 1];
            dd[i + 1] = t - f * dd[i + 1];
            if (i + 2 < d) {
                du2[i] = du[i + 1];
                du[i + 1] = -f * du[i + 1];
            }
            swp[i] = 1;
        }
    }
    if (dd[d - 1] == 0.0) dd[d - 1] = tiny;
    for (i64 i = 0; i < d; i++) y[i] = 1.0 + 0.01 * (double)((i * 2654435761u) % 97) / 97.0;
    for (int it = 0; it < 4; it++) {
        for (i64 i = 0; i + 1 < d; i++) {
            if (!swp[i])
                y[i + 1] -= dl[i] * y[i];
            else {
                const double t = y[i];
                y[i] = y[i + 1];
                y[i + 1] = t - dl[i] * y[i];
            }
        }
        y[d - 1] /= dd[d - 1];
        if (d > 1) y[d - 2] = (y[d - 2] - du[d - 2] * y[d - 1]) / dd[d - 2];
        for (i64 i = d - 3; i >= 0; i--) y[i] = (y[i] - du[i] * y[i + 1] - du2[i] * y[i + 2]) / dd[i];
        double nrm = 0.0, amax = 0.0;
        for (i64 i = 0; i < d; i++) amax = std::max(amax, std::fabs(y[i]));
        if (!(amax > 0.0) || !std::isfinite(amax)) { // degenerate: fall back to a unit vector
            for (i64 i = 0; i < d; i++) y[i] = (i == 0) ? 1.0 : 0.0;
            break;
        }
        for (i64 i = 0; i < d; i++) { y[i] /= amax; nrm += y[i] * y[i]; }
        nrm = std::sqrt(nrm);
        for (i64 i = 0; i < d; i++) y[i] /= nrm;
    }
    // back-transform: x = H_0 H_1 ... H_{d-3} y
    for (i64 k = d - 3; k >= 0; k--) {
        if (beta[k] == 0.0) continue;
        const i64 r = d - k - 1;
        double s = y[k + 1];
        for (i64 i = 1; i < r; i++) s += A[(k + 1 + i) * d + k] * y[k + 1 + i];
        s *= beta[k];
        y[k + 1] -= s;
        for (i64 i = 1; i < r; i++) y[k + 1 + i] -= s * A[(k + 1 + i) * d + k];
    }
    double nrm = 0.0;
    for (i64 i = 0; i < d; i++) nrm += y[i] * y[i];
    nrm = std::sqrt(nrm);
    if (!(nrm > 0.0) || !std::isfinite(nrm)) { // zero / degenerate matrix: any unit vector is an eigenvector
        for (i64 i = 0; i < d; i++) y[i] = (i == 0) ? 1.0 : 0.0;
        nrm = 1.0;
    }
    i64 big = 0;
    for (i64 i = 0; i < d; i++) {
        vout[i] = y[i] / nrm;
        if (std::fabs(vout[i]) > std::fabs(vout[big])) big = i;
    }
    if (vout[big] < 0.0)
        for (i64 i = 0; i < d; i++) vout[i] = -vout[i];
}

// ------------------------------------------------------------------------------------------------
namespace {

// ---- batched device work --------------------------------------------------------------------------------
// A batch = a list of groups; `rows` (device: c->ls_rows) holds their 0-based vertex ids back to back in the groups'
// own member order; every group is cut into chunks of CH rows (one workgroup each).
struct Batch {
    i64 T = 0, R = 0, NC = 0, max_len = 0; // max_len: rows of the longest group
    i32 *rows = nullptr, *row_task = nullptr; // host-built batches only: pinned staging owned by the ctx
    std::vector<i32> chunk_task, chunk_beg, chunk_end, task_chunk_off, task_row_off, task_off;
};
// chunk tables from the groups' lengths; `lens` = what.size() for a host-built batch
void batch_tables(Batch &B, const std::vector<i64> &lens) {
    const i64 CH = CGE_CHUNK_ROWS, T = (i64)lens.size();
    B.T = T;
    B.chunk_task.clear(); B.chunk_beg.clear(); B.chunk_end.clear();
    B.task_chunk_off.assign(T + 1, 0);
    B.task_row_off.assign(T + 1, 0);
    i64 pos = 0;
    B.max_len = 0;
    for (i64 t = 0; t < T; t++) {
        B.task_row_off[t] = (i32)pos;
        B.task_chunk_off[t] = (i32)B.chunk_task.size();
        const i64 k = lens[t];
        B.max_len = std::max(B.max_len, k);
        for (i64 s = 0; s < k; s += CH) {
            B.chunk_task.push_back((i32)t);
            B.chunk_beg.push_back((i32)(pos + s));
            B.chunk_end.push_back((i32)(pos + std::min(k, s + CH)));
        }
        pos += k;
    }
    B.R = pos;
    B.task_row_off[T] = (i32)pos;
    B.task_chunk_off[T] = (i32)B.chunk_task.size();
    B.NC = (i64)B.chunk_task.size();
}
// The other direction: device arrays -> one device staging area (one kernel) -> pinned memory (one copy).  fetch()
// waits for its copy (fetch_async() + wait() split the two halves); the pointers returned by get() are valid until the next
// gatherer of the same context is used.
struct WordGatherer {
    cge_ctx *c;
    std::vector<void *> ddst;
    std::vector<const void *> dsrc;
    std::vector<i64> words, offs;
    i64 tot = 0;
    explicit WordGatherer(cge_ctx *c_) : c(c_) {}
    template <class T>
    size_t add(const T *device, i64 count) { // returns the index of the item
        static_assert(sizeof(T) % 4 == 0, "4-byte words");
        dsrc.push_back(device);
        const i64 w = count * (i64)(sizeof(T) / 4);
        words.push_back(w);
        offs.push_back(tot);
        tot += w + (w & 1);
        return words.size() - 1;
    }
    void fetch() {
        fetch_async();
        wait();
    }
    void wait() { HIP_CHECK(hipEventSynchronize(c->copy_done)); }
    void fetch_async() { // ... and the event c->copy_done behind the copy
        c->pin_res.ensure((size_t)std::max<i64>(tot, 1));
        c->dev_res.ensure((size_t)std::max<i64>(tot, 1));
        ddst.resize(dsrc.size());
        for (size_t q = 0; q < dsrc.size(); q++) ddst[q] = c->dev_res.p + offs[q];
        for (size_t q0 = 0; q0 < dsrc.size(); q0 += CGE_WORD_SEGS) {
            const int n = (int)std::min<size_t>(CGE_WORD_SEGS, dsrc.size() - q0);
            k_copy_words(c, n, &ddst[q0], &dsrc[q0], &words[q0]);
        }
        if (tot > 0)
            HIP_CHECK(hipMemcpyAsync(c->pin_res.p, c->dev_res.p, sizeof(i32) * (size_t)tot, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipEventRecord(c->copy_done, c->stream));
    }
    template <class T>
    const T *get(size_t item) const { return reinterpret_cast<const T *>(c->pin_res.p + offs[item]); }
};
void upload_tables(cge_ctx *c, const Batch &B, WordPacker &pk) { // grow-only scratch owned by the ctx
    const i64 d = c->d;
    c->ls_rows.ensure(B.R); c->ls_row_task.ensure(B.R); c->ls_ct.ensure(B.NC); c->ls_cb.ensure(B.NC);
    c->ls_ce.ensure(B.NC); c->ls_tco.ensure(B.T + 1); c->sp_tro.ensure(B.T + 1);
    c->ls_part.ensure((size_t)B.NC * std::max(d * d, 2 * (2 * d + 1)));
    c->ls_side.ensure(B.R);
    c->ls_sums.ensure((size_t)B.T * 2 * (2 * d + 1));
    pk.add(c->ls_ct.p, B.chunk_task.data(), B.NC);
    pk.add(c->ls_cb.p, B.chunk_beg.data(), B.NC);
    pk.add(c->ls_ce.p, B.chunk_end.data(), B.NC);
    pk.add(c->ls_tco.p, B.task_chunk_off.data(), B.T + 1);
    pk.add(c->sp_tro.p, B.task_row_off.data(), B.T + 1);
}
// the usual batch: rows gathered on the device from the groups' arena ranges
void build_batch(cge_ctx *c, Group *const *groups, i64 T, Batch &B) {
    std::vector<i64> lens(T);
    B.task_off.resize(T);
    for (i64 t = 0; t < T; t++) {
        lens[t] = groups[t]->len;
        B.task_off[t] = (i32)groups[t]->off;
    }
    batch_tables(B, lens);
}
// `moff` (optional): the groups' offsets into the means arena, for k_gather_means -- it rides along
void upload_batch(cge_ctx *c, const Batch &B, const i64 *moff = nullptr) {
    WordPacker pk(c);
    upload_tables(c, B, pk);
    c->ls_toff.ensure(B.T);
    pk.add(c->ls_toff.p, B.task_off.data(), B.T);
    if (moff) {
        c->ls_moff.ensure(B.T);
        pk.add(c->ls_moff.p, moff, B.T);
    }
    pk.flush();
    k_gather_rows(c, c->lm_arena.p, c->ls_toff.p, c->sp_tro.p, c->ls_ct.p, c->ls_cb.p, c->ls_ce.p, B.NC, c->ls_rows.p,
                  c->ls_row_task.p);
}
// a batch from host member lists (the generic rss path): rows staged through pinned memory
void build_batch_host(cge_ctx *c, Group *const *groups, i64 T, Batch &B) {
    std::vector<i64> lens(T);
    for (i64 t = 0; t < T; t++) lens[t] = (i64)groups[t]->what.size();
    batch_tables(B, lens);
    c->pin_rows[1].ensure(B.R);
    c->pin_row_task[1].ensure(B.R);
    B.rows = c->pin_rows[1].p;
    B.row_task = c->pin_row_task[1].p;
    parallel_for(c, T, [&](i64 t) {
        const Group *g = groups[t];
        const i64 o = B.task_row_off[t], k = (i64)g->what.size();
        for (i64 j = 0; j < k; j++) {
            B.rows[o + j] = (i32)(g->what[j] - 1);
            B.row_task[o + j] = (i32)t;
        }
    });
}
void upload_batch_host(cge_ctx *c, const Batch &B) {
    WordPacker pk(c);
    upload_tables(c, B, pk);
    pk.flush();
    HIP_CHECK(hipMemcpyAsync(c->ls_rows.p, B.rows, sizeof(i32) * B.R, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(c->ls_row_task.p, B.row_task, sizeof(i32) * B.R, hipMemcpyHostToDevice, c->stream));
}
// sums[t][q] = { sum w x^2 [d], sum w x [d], sum w } over the rows of task t with side == q+1 (device)
// (the side flags are already in c->ls_side; the result lands in the pinned buffer c->pin_sums)
const double *side_sums_resident(cge_ctx *c, const Batch &B) {
    const i64 d = c->d, width = 2 * (2 * d + 1);
    hipStream_t st = c->stream;
    k_group_side_sums(c, c->Xr.p, lm_vw(c), c->ls_rows.p, c->ls_side.p, c->ls_cb.p, c->ls_ce.p, B.NC, c->ls_tco.p, B.T, d,
                      c->ls_part.p, c->ls_sums.p);
    c->pin_sums.ensure((size_t)B.T * width);
    HIP_CHECK(hipMemcpyAsync(c->pin_sums.p, c->ls_sums.p, sizeof(double) * B.T * width, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return c->pin_sums.p;
}
const double *side_sums(cge_ctx *c, const Batch &B, const std::vector<unsigned char> &side) {
    HIP_CHECK(hipMemcpyAsync(c->ls_side.p, side.data(), (size_t)B.R, hipMemcpyHostToDevice, c->stream));
    return side_sums_resident(c, B);
}
// sum over the columns of wsse(ss, s, ws)  (total_rss, src/landmarks.jl:269, given the column sums)
inline double rss_from_sums(const double *q, i64 d) {
    const double ws = q[2 * d];
    double tot = 0.0;
    for (i64 c2 = 0; c2 < d; c2++) tot += q[c2] - q[d + c2] * q[d + c2] / ws;
    return tot;
}
inline double sum_wsse_plus(const std::vector<Wsse> &base, const double *q, i64 d) { // sum(wsse, base .+ WSSE(set))
    const double ws = q[2 * d];
    double tot = 0.0;
    for (i64 c2 = 0; c2 < d; c2++) {
        const double ss = base[c2].ss + q[c2], s1 = base[c2].s + q[d + c2], w = base[c2].ws + ws;
        tot += ss - s1 * s1 / w;
    }
    return tot;
}
inline void add_sums(std::vector<Wsse> &base, const double *q, i64 d) {
    const double ws = q[2 * d];
    for (i64 c2 = 0; c2 < d; c2++) {
        base[c2].ss += q[c2];
        base[c2].s += q[d + c2];
        base[c2].ws += ws;
    }
}

// -total_rss of every group of a batch (roots of the local heaps), on the device
void device_group_values(cge_ctx *c, std::vector<Group *> &groups) {
    if (groups.empty()) return;
    const i64 d = c->d, width = 2 * (2 * d + 1);
    Batch B;
    build_batch(c, groups.data(), (i64)groups.size(), B);
    upload_batch(c, B);
    std::vector<unsigned char> side(B.R, 1);
    const double *sums = side_sums(c, B, side);
    for (i64 t = 0; t < B.T; t++) groups[t]->value = -rss_from_sums(&sums[(size_t)t * width], d);
}

// split_cluster_rss (src/landmarks.jl:155-210) for a whole batch: the 1-D logic (arg-min/max, medians,
// which half joins which side) runs on the host from z; the WSSE column sums of the candidate halves
// (:184-185, :201-202) come from one device pass per round over the rows that are still undecided.
struct RssState {
    std::vector<i64> low, high, gray, t1, t2;
    std::vector<Wsse> rl, rh;
    double med = 0.0;
    int phase = 0; // 0 = median rounds, 1 = leftover decision (:200-208), 2 = done
    int rc = CGE_OK;
};
void rule_rss_batched(cge_ctx *c, const Batch &B, Group *const *groups, const double *z, std::vector<RssState> &st) {
    const i64 T = B.T, d = c->d, width = 2 * (2 * d + 1);
    cge_ctx *hr = root_of(c); // the host mirrors live in the root context
    cge_ensure_host_embedding(hr); // generic round-based path only (ties at the maximum of z, NaNs)
    const double *hX = hr->h_Xr.data(), *hw = lm_hvw(hr); // (option shard_rows: this rank's rows, local ids)
    st.assign(T, RssState());
    parallel_for(c, T, [&](i64 t) {
        RssState &S = st[t];
        const Group *g = groups[t];
        const i64 k = (i64)g->what.size();
        const double *zt = &z[B.task_row_off[t]];
        i64 imin = 0, imax = 0;
        for (i64 j = 1; j < k; j++) {
            if (zt[j] < zt[imin]) imin = j;
            if (zt[j] > zt[imax]) imax = j;
        }
        if (imin == imax) { S.rc = CGE_E_HOMOGENEOUS; S.phase = 2; return; }
        S.low.assign(1, imin);
        S.high.assign(1, imax);
        S.gray.reserve(k);
        for (i64 j = 0; j < k; j++)
            if (j != imin && j != imax) S.gray.push_back(j);
        S.rl.resize(d);
        S.rh.resize(d);
        const double *x1 = hX + (g->what[imin] - 1) * d, *x2 = hX + (g->what[imax] - 1) * d;
        const double w1 = hw[g->what[imin] - 1], w2 = hw[g->what[imax] - 1];
        for (i64 q = 0; q < d; q++) { // :169-170
            S.rl[q] = {x1[q] * x1[q] * w1, x1[q] * w1, w1};
            S.rh[q] = {x2[q] * x2[q] * w2, x2[q] * w2, w2};
        }
        std::vector<double> scr;
        S.med = median_sel(zt, nullptr, k, scr);
    });
    // device state: 0 = gray for every row; the two seed rows are decided from the start
    std::vector<unsigned char> state0(B.R, 0);
    for (i64 t = 0; t < T; t++) {
        if (st[t].phase == 2) continue;
        state0[B.task_row_off[t] + st[t].low[0]] = 1;
        state0[B.task_row_off[t] + st[t].high[0]] = 2;
    }
    hipStream_t stream = c->stream;
    c->ls_state.ensure(B.R);
    c->ls_params.ensure((size_t)4 * T);
    c->pin_params.ensure((size_t)4 * T);
    HIP_CHECK(hipMemcpyAsync(c->ls_state.p, state0.data(), (size_t)B.R, hipMemcpyHostToDevice, stream));
    double *prm = c->pin_params.p; // {prev_med, absorb, cur_med, mode} per task
    for (i64 t = 0; t < T; t++) { prm[4 * t] = 0.0; prm[4 * t + 1] = 0.0; prm[4 * t + 2] = st[t].med; prm[4 * t + 3] = st[t].phase == 2 ? 0.0 : 1.0; }
    for (;;) {
        i64 n_active = 0;
        for (i64 t = 0; t < T; t++) n_active += (st[t].phase != 2);
        if (n_active == 0) break;
        // host copy of this round's halves (needed to update the member lists afterwards)
        parallel_for(c, T, [&](i64 t) {
            RssState &S = st[t];
            if (S.phase != 0) return;
            const double *zt = &z[B.task_row_off[t]];
            S.t1.clear();
            S.t2.clear();
            for (i64 j : S.gray) (zt[j] < S.med ? S.t1 : S.t2).push_back(j);
        });
        HIP_CHECK(hipMemcpyAsync(c->ls_params.p, prm, sizeof(double) * 4 * T, hipMemcpyHostToDevice, stream));
        k_rss_side(c, c->ls_z.p, c->ls_row_task.p, B.R, c->ls_params.p, c->ls_state.p, c->ls_side.p);
        const double *sums = side_sums_resident(c, B);
        parallel_for(c, T, [&](i64 t) {
            RssState &S = st[t];
            double *pt = prm + 4 * t;
            if (S.phase == 2) { pt[1] = 0.0; pt[3] = 0.0; return; }
            const double *q1 = &sums[(size_t)t * width], *q2 = q1 + (2 * d + 1);
            const double *zt = &z[B.task_row_off[t]];
            pt[0] = S.med; // becomes prev_med of the next round
            pt[1] = 0.0;
            if (S.phase == 0) {
                if (sum_wsse_plus(S.rl, q1, d) < sum_wsse_plus(S.rh, q2, d)) {
                    if (S.t1.empty()) { S.phase = 1; pt[3] = 2.0; return; }
                    add_sums(S.rl, q1, d);
                    S.low.insert(S.low.end(), S.t1.begin(), S.t1.end());
                    S.gray.swap(S.t2);
                    pt[1] = 1.0;
                } else {
                    if (S.t2.empty()) { S.phase = 1; pt[3] = 2.0; return; }
                    add_sums(S.rh, q2, d);
                    S.high.insert(S.high.end(), S.t2.begin(), S.t2.end());
                    S.gray.swap(S.t1);
                    pt[1] = 2.0;
                }
                if (S.gray.empty()) { S.phase = 2; pt[3] = 0.0; return; }
                std::vector<double> scr;
                S.med = median_sel(zt, &S.gray, 0, scr);
                pt[2] = S.med;
                pt[3] = 1.0;
            } else { // :200-208, sums of the whole leftover are in q1
                const double a = std::max(sum_wsse_plus(S.rl, q1, d), sum_wsse(S.rh));
                const double b = std::max(sum_wsse(S.rl), sum_wsse_plus(S.rh, q1, d));
                auto &dst = (a < b) ? S.low : S.high;
                dst.insert(dst.end(), S.gray.begin(), S.gray.end());
                S.gray.clear();
                S.phase = 2;
                pt[3] = 0.0;
            }
        });
    }
}

// What a rule leaves behind for every task of a batch: the children's member lists are already in the arena range of
// the batch (task t at base + task_row_off[t], low first); the host gets the sizes, values and means.
struct CutResult {
    std::vector<i32> nlow;
    std::vector<double> vlow, vhigh;
    std::vector<char> done; // 0 = this task still needs the generic host path
};

// One (half) batch in flight on one context (the root or its second lane): everything from the batch tables to the
// rule's cut is ENQUEUED on that context's stream without a host synchronisation (lane_enqueue), the results are read
// when the host comes back for them (lane_collect).  Two of these run half a chain out of phase (compute_splits).
struct LaneRun {
    cge_ctx *x = nullptr; // where it runs
    Group *const *groups = nullptr;
    i64 T = 0;
    int method = 0;
    Batch B;
    i64 base = 0, mbase = -1; // children ranges, children means (arena offsets)
    CutResult cr;
    std::unique_ptr<WordGatherer> wg;
    size_t i_status = 0, i_meta = 0, i_vals = 0, i_nlow = 0;
};

// split_cluster_rss on sorted order (kernels_lm.hip: k_sorted_prefix + k_rss_rounds; kernels_sort.hip): the device
// sorts z per task, scans the WSSE terms along that order, runs all median-cut rounds of every task in one launch, and
// writes the children's member lists in the reference's order (seed first, then every absorbed batch in ascending
// original index, :163-164, :189, :194, :204-206) by one stable radix pass over per-row bucket keys.
// Tasks the rank-range argument does not cover (a tie at the maximum of z, NaNs) are left to the generic path.
void rule_rss_sorted_enqueue(LaneRun &L) {
    cge_ctx *c = L.x;
    const Batch &B = L.B;
    const i64 T = B.T, R = B.R, d = c->d, W = 2 * d + 1;
    c->sp_srows.ensure(R); c->sp_zs.ensure(R); c->sp_perm.ensure(R); c->sp_status.ensure(T);
    c->sp_ctot.ensure((size_t)B.NC * W); c->sp_coff.ensure((size_t)B.NC * W);
    c->sp_prefix.ensure((size_t)(R / CGE_PREFIX_STRIDE + B.NC + 1) * W);
    c->sp_meta.ensure(2 * T); c->sp_rounds.ensure((size_t)T * 3 * CGE_RR_MAXROUNDS); c->sp_vals.ensure(2 * T);
    c->ls_keys.ensure(R); c->ls_nlow.ensure(T);
    k_segmented_sort_z(c, c->ls_z.p, c->ls_rows.p, c->ls_row_task.p, c->sp_tro.p, R, T, c->sp_zs.p, c->sp_perm.p,
                       c->sp_srows.p, c->sp_status.p, B.max_len);
    k_sorted_prefix(c, c->Xr.p, lm_vw(c), c->sp_srows.p, c->ls_cb.p, c->ls_ce.p, B.NC, c->ls_tco.p, T, d, c->sp_ctot.p,
                    c->sp_coff.p, c->sp_prefix.p);
    k_rss_rounds(c, c->Xr.p, lm_vw(c), c->sp_srows.p, c->sp_zs.p, c->sp_tro.p, c->ls_tco.p, c->sp_prefix.p, c->sp_coff.p,
                 T, d, c->sp_meta.p, c->sp_rounds.p, c->sp_vals.p, c->lm_means.p + L.mbase); // the children's means stay on the device
    k_rss_child_keys(c, c->sp_perm.p, c->ls_row_task.p, c->sp_tro.p, c->sp_meta.p, c->sp_rounds.p, R, T, c->ls_keys.p,
                     c->ls_nlow.p);
    // what the host needs for the heap (status, rounds, children values and sizes) is fetched BEFORE the children's member
    // lists are written: the host replays the heap and builds the next batch while that sort still runs (it only feeds the
    // device-side arena; the next batch queues behind it on the same stream)
    L.wg.reset(new WordGatherer(c));
    L.i_status = L.wg->add(c->sp_status.p, T); L.i_meta = L.wg->add(c->sp_meta.p, 2 * T);
    L.i_vals = L.wg->add(c->sp_vals.p, 2 * T); L.i_nlow = L.wg->add(c->ls_nlow.p, T);
    L.wg->fetch_async();
    k_sort_children(c, c->ls_keys.p, c->ls_rows.p, c->sp_tro.p, c->ls_cb.p, c->ls_ce.p, c->ls_tco.p, B.NC, R, T, 7, c->lm_arena.p + L.base);
}
void rule_rss_sorted_collect(LaneRun &L) {
    const i64 T = L.B.T, d = L.x->d;
    CutResult &out = L.cr;
    L.wg->wait();
    const i32 *status = L.wg->get<i32>(L.i_status), *meta = L.wg->get<i32>(L.i_meta);
    const double *vals = L.wg->get<double>(L.i_vals);
    std::memcpy(out.nlow.data(), L.wg->get<i32>(L.i_nlow), sizeof(i32) * T);
    for (i64 t = 0; t < T; t++) {
        Group *g = L.groups[t];
        if (status[t] == 2) { g->rc = CGE_E_HOMOGENEOUS; out.done[t] = 1; continue; }
        if (status[t] == 1 || meta[2 * t + 1] != 0) { out.done[t] = 0; continue; }
        out.vlow[t] = vals[2 * t];
        out.vhigh[t] = vals[2 * t + 1];
        g->cmean_off = L.mbase + 2 * t * d;
        g->rc = CGE_OK;
        out.done[t] = 1;
    }
}

// split_cluster_rss2 on the device (kernels_lm.hip: rss2_walk_kernel).  The children are rank ranges of the sorted
// order, in that order (`p[1:low]`, `p[high:end]`, src/landmarks.jl:151): the sorted rows ARE the two lists.
void rule_rss2_enqueue(LaneRun &L) {
    cge_ctx *c = L.x;
    const Batch &B = L.B;
    const i64 T = B.T, R = B.R, d = c->d;
    hipStream_t st = c->stream;
    c->sp_srows.ensure(R); c->sp_zs.ensure(R); c->sp_perm.ensure(R); c->sp_status.ensure(T);
    c->sp_meta.ensure(2 * T); c->sp_vals.ensure(2 * T);
    k_segmented_sort_z(c, c->ls_z.p, c->ls_rows.p, c->ls_row_task.p, c->sp_tro.p, R, T, c->sp_zs.p, c->sp_perm.p,
                       c->sp_srows.p, c->sp_status.p, B.max_len);
    HIP_CHECK(hipMemcpyAsync(c->lm_arena.p + L.base, c->sp_srows.p, sizeof(i32) * R, hipMemcpyDeviceToDevice, st));
    c->r2_rows = R;
    k_rss2_walk(c, c->Xr.p, lm_vw(c), c->sp_srows.p, c->sp_tro.p, T, d, c->sp_meta.p, c->sp_vals.p, c->lm_means.p + L.mbase);
    L.wg.reset(new WordGatherer(c));
    L.i_meta = L.wg->add(c->sp_meta.p, 2 * T); L.i_vals = L.wg->add(c->sp_vals.p, 2 * T);
    L.wg->fetch_async();
}
void rule_rss2_collect(LaneRun &L) {
    const i64 T = L.B.T, d = L.x->d;
    CutResult &out = L.cr;
    L.wg->wait();
    const i32 *meta = L.wg->get<i32>(L.i_meta);
    const double *vals = L.wg->get<double>(L.i_vals);
    for (i64 t = 0; t < T; t++) {
        Group *g = L.groups[t];
        out.nlow[t] = meta[2 * t] + 1; // low = ranks [0, lo], high = ranks [hi, k) with hi == lo + 1
        out.vlow[t] = vals[2 * t];
        out.vhigh[t] = vals[2 * t + 1];
        g->cmean_off = L.mbase + 2 * t * d;
        g->rc = CGE_OK;
        out.done[t] = 1;
    }
}

// split_cluster_size / split_cluster_diameter on the device (kernels_lm.hip: cut_sides_kernel): the side of every row,
// the children's WSSE column sums (values and means) by the side-sums pass, and the two member lists -- the rows of
// either side in the rows' own order -- by a stable one-bit sort.
void rule_cut_enqueue(LaneRun &L, bool use_median) {
    cge_ctx *c = L.x;
    const Batch &B = L.B;
    const i64 T = B.T, R = B.R, d = c->d, width = 2 * (2 * d + 1);
    hipStream_t st = c->stream;
    if (use_median) { // the median needs the sorted projections
        c->sp_srows.ensure(R); c->sp_zs.ensure(R); c->sp_perm.ensure(R); c->sp_status.ensure(T);
        k_segmented_sort_z(c, c->ls_z.p, c->ls_rows.p, c->ls_row_task.p, c->sp_tro.p, R, T, c->sp_zs.p, c->sp_perm.p,
                           c->sp_srows.p, c->sp_status.p, B.max_len);
    }
    c->ls_nlow.ensure(T);
    k_cut_sides(c, c->ls_z.p, use_median ? c->sp_zs.p : nullptr, c->sp_tro.p, T, use_median ? 1 : 0, c->ls_side.p, c->ls_nlow.p,
                root_of(c)->cut_ties.p);
    c->pin_res.ensure((size_t)T); // pinned: the copies do not stall the host, the event below covers them
    HIP_CHECK(hipMemcpyAsync(c->pin_res.p, c->ls_nlow.p, sizeof(i32) * T, hipMemcpyDeviceToHost, st));
    k_group_side_sums(c, c->Xr.p, lm_vw(c), c->ls_rows.p, c->ls_side.p, c->ls_cb.p, c->ls_ce.p, B.NC, c->ls_tco.p, B.T, d,
                      c->ls_part.p, c->ls_sums.p);
    // the children's values and means from the sums, on the device: the means go straight into the arena (the next batch reads
    // them there), the host reads two doubles per task (round 4; it used to fetch the sums, divide and upload the means: a
    // round trip and a stream synchronisation per batch, thirty times per score with the cut rules)
    (void)width;
    c->sp_vals.ensure(2 * T);
    k_side_values_means(c, c->ls_sums.p, T, d, c->sp_vals.p, c->lm_means.p + L.mbase);
    c->pin_sums.ensure((size_t)2 * T);
    HIP_CHECK(hipMemcpyAsync(c->pin_sums.p, c->sp_vals.p, sizeof(double) * 2 * T, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(c->copy_done, st));
    // the children's member lists last: the host already has what its heap needs and builds the next batch meanwhile
    k_sort_children(c, c->ls_side.p, c->ls_rows.p, c->sp_tro.p, c->ls_cb.p, c->ls_ce.p, c->ls_tco.p, L.B.NC, R, T, 2, c->lm_arena.p + L.base);
}
void rule_cut_collect(LaneRun &L) {
    cge_ctx *c = L.x;
    const i64 T = L.B.T, d = c->d, width = 2 * (2 * d + 1);
    CutResult &out = L.cr;
    HIP_CHECK(hipEventSynchronize(c->copy_done));
    const double *vals = c->pin_sums.p; // (-total_rss of the two children, computed beside the means on the device)
    std::memcpy(out.nlow.data(), c->pin_res.p, sizeof(i32) * T);
    (void)width;
    for (i64 t = 0; t < T; t++) {
        Group *g = L.groups[t];
        out.vlow[t] = vals[2 * t];
        out.vhigh[t] = vals[2 * t + 1];
        g->cmean_off = L.mbase + 2 * t * d;
        g->rc = CGE_OK;
        out.done[t] = 1;
    }
    // (the means are in the arena already, written on this lane's stream; a second lane starts behind an event of this stream)
}

// The generic round-based rss path for the tasks the sorted-order kernels declined (ties at the maximum of z, NaNs):
// their member lists and projections come to the host, rule_rss_batched runs on a sub-batch, and the children's
// lists go back into the tasks' arena ranges.
void rss_generic_tasks(cge_ctx *c, const Batch &B, Group *const *groups, i64 base, const std::vector<i64> &todo,
                       CutResult &out) {
    const i64 d = c->d, width = 2 * (2 * d + 1);
    hipStream_t st = c->stream;
    std::vector<Group *> fg;
    std::vector<std::vector<i32>> rows(todo.size());
    std::vector<double> fz;
    std::vector<i64> zoff(todo.size() + 1, 0);
    for (size_t q = 0; q < todo.size(); q++) zoff[q + 1] = zoff[q] + groups[todo[q]]->len;
    fz.resize(zoff.back());
    for (size_t q = 0; q < todo.size(); q++) {
        const i64 t = todo[q], o = B.task_row_off[t], k = groups[t]->len;
        rows[q].resize(k);
        HIP_CHECK(hipMemcpyAsync(rows[q].data(), c->ls_rows.p + o, sizeof(i32) * k, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(fz.data() + zoff[q], c->ls_z.p + o, sizeof(double) * k, hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    for (size_t q = 0; q < todo.size(); q++) {
        Group *g = groups[todo[q]];
        g->what.resize(rows[q].size());
        for (size_t j = 0; j < rows[q].size(); j++) g->what[j] = (i64)rows[q][j] + 1;
        fg.push_back(g);
    }
    Batch FB;
    build_batch_host(c, fg.data(), (i64)fg.size(), FB);
    upload_batch_host(c, FB);
    c->ls_z.ensure(FB.R);
    HIP_CHECK(hipMemcpyAsync(c->ls_z.p, fz.data(), sizeof(double) * FB.R, hipMemcpyHostToDevice, st));
    std::vector<RssState> rs;
    rule_rss_batched(c, FB, fg.data(), fz.data(), rs);
    // children: values and means by one side-sums pass over the sub-batch, lists straight into the arena
    std::vector<unsigned char> side(FB.R, 0);
    for (size_t q = 0; q < fg.size(); q++) {
        if (rs[q].rc != CGE_OK) continue;
        unsigned char *sd = &side[FB.task_row_off[q]];
        for (i64 j : rs[q].low) sd[j] = 1;
        for (i64 j : rs[q].high) sd[j] = 2;
    }
    const double *sums = side_sums(c, FB, side);
    for (size_t q = 0; q < fg.size(); q++) {
        const i64 t = todo[q];
        Group *g = fg[q];
        g->rc = rs[q].rc;
        out.done[t] = 1;
        if (g->rc != CGE_OK) continue;
        std::vector<i32> kids;
        kids.reserve(g->what.size());
        for (i64 j : rs[q].low) kids.push_back((i32)(g->what[j] - 1));
        for (i64 j : rs[q].high) kids.push_back((i32)(g->what[j] - 1));
        out.nlow[t] = (i32)rs[q].low.size();
        if (kids.size() == g->what.size())
            HIP_CHECK(hipMemcpy(c->lm_arena.p + base + B.task_row_off[t], kids.data(), sizeof(i32) * kids.size(),
                                hipMemcpyHostToDevice));
        else
            g->rc = CGE_E_EMPTY_CLUSTER;
        const double *q1 = &sums[(size_t)q * width], *q2 = q1 + (2 * d + 1);
        out.vlow[t] = -rss_from_sums(q1, d);
        out.vhigh[t] = -rss_from_sums(q2, d);
        std::vector<double> m2(2 * d);
        for (i64 c2 = 0; c2 < d; c2++) {
            m2[c2] = q1[d + c2] / q1[2 * d];
            m2[d + c2] = q2[d + c2] / q2[2 * d];
        }
        g->cmean_off = means_alloc(c, 2 * d);
        HIP_CHECK(hipMemcpy(c->lm_means.p + g->cmean_off, m2.data(), sizeof(double) * 2 * d, hipMemcpyHostToDevice));
        g->what.clear();
        g->what.shrink_to_fit();
    }
}

// Enqueue the split of every task of a (half) batch on L.x: mean, covariance, principal eigenvector, projection, the
// rule's 1-D cut, the children's member lists, values and means -- no host synchronisation (except the d > 512 host
// eigen-solver).  `after_cov` (optional): recorded behind the covariance, the point the other lane's start waits for.
void lane_enqueue(LaneRun &L) {
    cge_ctx *c = L.x, *root = root_of(c);
    const i64 d = c->d, T = L.T;
    hipStream_t st = c->stream;
    Batch &B = L.B;
    bool have_means = true; // known from the parents' splits: gathered from the means arena, no pass over the rows
    for (i64 t = 0; t < T && have_means; t++) have_means = L.groups[t]->mean_off >= 0;
    {
        PhaseAcc pa(root, "lm_pack");
        build_batch(c, L.groups, T, B);
        std::vector<i64> moff;
        if (have_means) {
            moff.resize(T);
            for (i64 t = 0; t < T; t++) moff[t] = L.groups[t]->mean_off;
        }
        upload_batch(c, B, have_means ? moff.data() : nullptr);
    }
    const i64 R = B.R, NC = B.NC;
    root->stat_lm_batches++;
    root->stat_lm_rows += R;
    root->stat_lm_splits += T;
    c->ls_mean.ensure((size_t)T * d); c->ls_sw.ensure(T);
    c->ls_cov.ensure((size_t)T * d * d);
    c->ls_vec.ensure((size_t)T * d); c->ls_z.ensure(R);
    {
        PhaseAcc pa(root, "lm_pca_dev");
        double *covp = c->ls_cov.p;
        {
            ScopedKernelTimer tm(c, "group_stats");
            if (have_means) // the offsets went up with the batch's tables
                k_gather_means(c, c->lm_means.p, c->ls_moff.p, T, d, c->ls_mean.p);
            else {
                k_group_mean(c, c->Xr.p, lm_vw(c), c->ls_rows.p, c->ls_ct.p, c->ls_cb.p, c->ls_ce.p, NC, c->ls_tco.p, T, d,
                             c->ls_part.p, c->ls_mean.p, c->ls_sw.p);
            }
            k_group_cov(c, c->Xr.p, lm_vw(c), c->ls_rows.p, c->ls_ct.p, c->ls_cb.p, c->ls_ce.p, NC, c->ls_tco.p, T, d,
                        c->ls_mean.p, c->ls_part.p, covp);
        }
        if (!k_group_eig(c, covp, T, d, c->ls_vec.p)) { // d > 512: host solver on a worker pool
            std::vector<double> cov((size_t)T * d * d), vec((size_t)T * d);
            HIP_CHECK(hipMemcpyAsync(cov.data(), covp, sizeof(double) * cov.size(), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            parallel_for(c, T, [&](i64 t) { host_eig_top(&cov[(size_t)t * d * d], d, &vec[(size_t)t * d]); });
            HIP_CHECK(hipMemcpyAsync(c->ls_vec.p, vec.data(), sizeof(double) * vec.size(), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st)); // vec goes out of scope
        }
        {
            ScopedKernelTimer tm(c, "group_project");
            k_group_project(c, c->Xr.p, lm_vw(c), c->ls_rows.p, c->ls_row_task.p, R, d, c->ls_mean.p, c->ls_vec.p,
                            c->ls_z.p);
        }
    }
    // ---- the cut: children lists into the arena, sizes / values / means to the host -----------------------------
    L.cr.nlow.assign(T, 0);
    L.cr.vlow.assign(T, 0.0);
    L.cr.vhigh.assign(T, 0.0);
    L.cr.done.assign(T, 0);
    if (L.method == CGE_METHOD_RSS) rule_rss_sorted_enqueue(L);
    else if (L.method == CGE_METHOD_RSS2) rule_rss2_enqueue(L);
    else rule_cut_enqueue(L, L.method == CGE_METHOD_SIZE);
}
// ... and book the results when they have arrived
void lane_collect(LaneRun &L) {
    cge_ctx *c = L.x, *root = root_of(c);
    const i64 T = L.T;
    const Batch &B = L.B;
    {
        PhaseAcc pa(root, "lm_cut");
        if (L.method == CGE_METHOD_RSS) {
            rule_rss_sorted_collect(L);
            std::vector<i64> todo;
            for (i64 t = 0; t < T; t++)
                if (!L.cr.done[t]) todo.push_back(t);
            if (!todo.empty()) rss_generic_tasks(c, B, L.groups, L.base, todo, L.cr);
        } else if (L.method == CGE_METHOD_RSS2)
            rule_rss2_collect(L);
        else
            rule_cut_collect(L);
    }
    for (i64 t = 0; t < T; t++) {
        Group *g = L.groups[t];
        if (g->rc != CGE_OK) continue;
        const i64 nl = L.cr.nlow[t], nh = g->len - nl;
        if (nl <= 0 || nh <= 0) { g->rc = CGE_E_EMPTY_CLUSTER; continue; }
        g->coff = L.base + B.task_row_off[t];
        g->nlow = nl;
        g->vlow = nl > 1 ? L.cr.vlow[t] : DBL_EPSILON;
        g->vhigh = nh > 1 ? L.cr.vhigh[t] : DBL_EPSILON;
    }
}

// Compute the split of every task, on the device.  The host only books the results.
// TWO LANES: the chain of a batch is a dozen dependent kernels of very different character -- covariance (MFMA),
// eigen-solver (126 dependent Householder steps: latency, most of the chip idle), projection / sort / scan / rounds (HBM
// gathers).  A batch is therefore cut into two halves that run on two streams half a chain out of phase (the second
// half starts when the first half's covariance is done), so one half's eigen-solver overlaps the other's memory-bound
// kernels, and the host side of one half (tables, result parsing) overlaps the device side of the other.  The halves
// are independent (disjoint groups, disjoint arena ranges); results do not depend on the split.
void compute_splits(cge_ctx *c, std::vector<Group *> &tasks, int method) {
    const i64 d = c->d;
    std::vector<Group *> big;
    for (Group *g : tasks) {
        const i64 k = g->len;
        g->has_split = true;
        g->rc = CGE_OK;
        // the rules' own asserts (:93,:156,:219 `size(m,1) > 1`; :248 `size(m,2) > 1`)
        if ((method != CGE_METHOD_DIAMETER && k <= 1) || (method == CGE_METHOD_DIAMETER && d <= 1)) {
            g->rc = CGE_E_ASSERT;
            continue;
        }
        if (k <= 1) { g->rc = CGE_E_EMPTY_CLUSTER; continue; } // diameter rule on one row: `low` comes out empty
        if (k == 2) { // `return [1], [2]`: the two members stay where they are
            g->coff = g->off;
            g->nlow = 1;
            g->vlow = g->vhigh = DBL_EPSILON;
            g->cmean_off = -1;
            continue;
        }
        big.push_back(g);
    }
    if (big.empty()) return;

    // Sub-batches bound the covariance buffers (T * d*d doubles) to ~1 GiB.
    const i64 max_tasks = std::max<i64>(1, (i64)(1ull << 27) / (d * d));
    for (size_t b0 = 0; b0 < big.size(); b0 += (size_t)max_tasks) {
        const size_t b1 = std::min(big.size(), b0 + (size_t)max_tasks);
        LaneRun L;
        L.x = c;
        L.groups = &big[b0];
        L.T = (i64)(b1 - b0);
        L.method = method;
        i64 r = 0;
        for (i64 t = 0; t < L.T; t++) r += L.groups[t]->len;
        L.base = arena_alloc(c, r); // the children of task t: [base + task_row_off[t], + len)
        L.mbase = means_alloc(c, 2 * L.T * d);
        lane_enqueue(L);
        lane_collect(L);
    }
}

// N > 1, option shard_rows: a group's rows live on ONE rank (its community's owner), which computes the split; what the
// replicated heap needs of it -- status, size of the low child, the two children's values: four 8-byte words per group --
// is gathered by one all-reduce into a zero-filled buffer (op 2: integer sum of the words, exact).  The member lists, the
// means and the arena offsets never leave the owner.
// Owner-only work between two collectives must not leave the other ranks waiting in the next one: an error raised by it is
// caught (guarded_work), travels with the words of that exchange -- one slot per rank behind them -- and AFTER the exchange
// every rank throws: the failing rank its own error, the others the same code ("every rank leaves by the same door").
struct RankError { int code = 0; std::string msg; };
template <class F> static RankError guarded_work(F &&work) {
    try { work(); } catch (const CgeError &e) { return RankError{e.code ? e.code : CGE_E_ASSERT, e.msg}; }
    return RankError{};
}
static void throw_if_a_rank_failed(cge_ctx *c, const RankError &mine, const double *slots, const char *where) {
    if (mine.code) throw CgeError{mine.code, mine.msg};
    for (int r = 0; r < c->coll.world; r++)
        if (slots[r] != 0.0)
            CGE_THROW((int)slots[r], "%s: rank %d failed with code %d (its message is on that rank); every rank stops here", where, r, (int)slots[r]);
}
void exchange_group_words(cge_ctx *c, std::vector<double> &w);
// the exchange of exchange_group_words with the ranks' verdicts behind the words
static void exchange_group_words_checked(cge_ctx *c, std::vector<double> &w, const RankError &mine, const char *where) {
    const size_t n0 = w.size();
    const int W = c->has_coll ? c->coll.world : 1;
    w.resize(n0 + (size_t)W, 0.0);
    if (mine.code) w[n0 + (size_t)c->coll.rank] = (double)mine.code;
    exchange_group_words(c, w);
    std::vector<double> slots(w.begin() + (std::ptrdiff_t)n0, w.end());
    w.resize(n0);
    throw_if_a_rank_failed(c, mine, slots.data(), where);
}
void exchange_group_words(cge_ctx *c, std::vector<double> &w) { // in: this rank's words, zeros elsewhere; out: everybody's
    if (w.empty()) return;
    PhaseAcc px(c, "lm_exchange");
    DevBuf<double> &X = c->samp_xchg; // (free during the landmark phase)
    X.ensure(w.size());
    HIP_CHECK(hipMemcpyAsync(X.p, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, c->stream));
    cge_allreduce_dev(c, X.p, (i64)w.size(), 2);
    HIP_CHECK(hipMemcpyAsync(w.data(), X.p, sizeof(double) * w.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
void compute_splits_rowsharded(cge_ctx *c, std::vector<Group *> &batch, int method) {
    const i64 T = (i64)batch.size();
    const int me = c->coll.rank;
    std::vector<Group *> mine;
    for (Group *g : batch)
        if (g->owner == me) mine.push_back(g);
    const RankError err = guarded_work([&] { if (!mine.empty()) compute_splits(c, mine, method); });
    std::vector<double> w((size_t)4 * T, 0.0);
    for (i64 t = 0; t < T && !err.code; t++) {
        const Group *g = batch[t];
        if (g->owner != me) continue;
        w[4 * t] = (double)(g->rc - 1); // (never the all-zero word: a group nobody answered for is detected below)
        w[4 * t + 1] = (double)g->nlow;
        w[4 * t + 2] = g->vlow;
        w[4 * t + 3] = g->vhigh;
    }
    exchange_group_words_checked(c, w, err, "runsplit (shard_rows)");
    for (i64 t = 0; t < T; t++) {
        Group *g = batch[t];
        if (g->owner == me) continue;
        if (w[4 * t] == 0.0) CGE_THROW(CGE_E_COLLECTIVE, "runsplit (shard_rows): no rank answered for a group of rank %d", g->owner);
        g->has_split = true;
        g->rc = (int)w[4 * t] + 1;
        g->coff = -1;
        g->cmean_off = -1;
        if (g->rc != CGE_OK) continue;
        g->nlow = (i64)w[4 * t + 1];
        g->vlow = w[4 * t + 2];
        g->vhigh = w[4 * t + 3];
    }
}

// N > 1, global phase: the groups of a batch are independent, so every rank cuts its share (longest first, each to the
// least loaded rank) and the results -- status, cut position, children values, children member lists and means -- are
// gathered by one all-reduce into zero-filled buffers (hook op 2, exact).  EVERY rank, owner or not, then re-allocates the
// children ranges and means in the batch's order and takes the gathered values: all ranks stay in identical state.
// Worth it only for batches with enough rows to outweigh the exchange.
void compute_splits_sharded(cge_ctx *c, std::vector<Group *> &batch, int method) {
    if (c->rows_sharded) { compute_splits_rowsharded(c, batch, method); return; }
    const i64 W = c->has_coll ? c->coll.world : 1, d = c->d, T = (i64)batch.size();
    i64 R = 0;
    for (Group *g : batch) R += g->len;
    const i64 s_words = (R + 1) / 2, m_per = 5 + 2 * d; // per group: rc, nlow, vlow, vhigh, means flag, two means
    const i64 x_need = s_words + T * m_per;
    // (option value 2: every batch, whatever its size -- the tests)
    if (W <= 1 || !c->opt_shard_forced || (c->opt_shard_forced == 1 && (T < 2 * W || R * d < ((i64)1 << 23))) ||
        !cge_exchange_fits(c, (size_t)(x_need + W))) {
        compute_splits(c, batch, method);
        return;
    }
    hipStream_t st = c->stream;
    std::vector<i64> ord(T), load(W, 0), prefix(T + 1, 0);
    std::vector<int> owner(T, 0);
    for (i64 t = 0; t < T; t++) { ord[t] = t; prefix[t + 1] = prefix[t] + batch[t]->len; }
    std::stable_sort(ord.begin(), ord.end(), [&](i64 a, i64 b) { return batch[a]->len > batch[b]->len; });
    for (i64 t : ord) {
        const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        owner[t] = r;
        load[r] += batch[t]->len;
    }
    const int me = c->coll.rank;
    const i64 A0 = c->lm_arena_used, M0 = c->lm_means_used;
    std::vector<Group *> mine;
    for (i64 t = 0; t < T; t++)
        if (owner[t] == me) mine.push_back(batch[t]);
    const RankError err = guarded_work([&] { if (!mine.empty()) compute_splits(c, mine, method); });
    PhaseAcc px(c, "lm_exchange");
    double *X = c->xptr;
    i32 *S = reinterpret_cast<i32 *>(X);
    double *Mg = X + s_words;
    HIP_CHECK(hipMemsetAsync(X, 0, sizeof(double) * (x_need + W), st)); // (W verdict slots behind the words)
    const double my_verdict = (double)err.code;
    if (err.code) HIP_CHECK(hipMemcpyAsync(X + x_need + me, &my_verdict, sizeof(double), hipMemcpyHostToDevice, st));
    std::vector<i64> seg, moff, mslot;
    std::vector<double> hg((size_t)T * 5, 0.0);
    for (i64 t = 0; t < T && !err.code; t++) {
        if (owner[t] != me) continue;
        Group *g = batch[t];
        hg[5 * t] = (double)g->rc;
        if (g->rc != CGE_OK) continue;
        hg[5 * t + 1] = (double)g->nlow;
        hg[5 * t + 2] = g->vlow;
        hg[5 * t + 3] = g->vhigh;
        hg[5 * t + 4] = g->cmean_off >= 0 ? 1.0 : 0.0;
        seg.push_back(g->coff); seg.push_back(prefix[t]); seg.push_back(g->len);
        if (g->cmean_off >= 0) { // the two means are adjacent in the means arena: two rows of d for the gather kernel
            moff.push_back(g->cmean_off); mslot.push_back(2 * t);
            moff.push_back(g->cmean_off + d); mslot.push_back(2 * t + 1);
        }
    }
    DevBuf<i64> d_seg, d_moff;
    if (!seg.empty()) {
        d_seg.ensure(seg.size());
        HIP_CHECK(hipMemcpyAsync(d_seg.p, seg.data(), sizeof(i64) * seg.size(), hipMemcpyHostToDevice, st));
        k_copy_segments(c, c->lm_arena.p, d_seg.p, (i64)seg.size() / 3, S);
    }
    HIP_CHECK(hipMemcpy2DAsync(Mg, sizeof(double) * m_per, hg.data(), sizeof(double) * 5, sizeof(double) * 5, (size_t)T,
                               hipMemcpyHostToDevice, st));
    if (!moff.empty()) { // slot q = 2t + (0 low / 1 high) -> Mg + t * m_per + 5 + (q & 1) * d: stride d over a view that starts at
        // Mg + 5 only works when m_per == 2d; use the generic form: one record per HALF group of stride m_per / 2 is not
        // integral, so the kernel gets (slot, stride, lead) per row through two launches (low halves, high halves)
        std::vector<i64> off_lo, off_hi, slot_t;
        for (size_t q = 0; q < moff.size(); q += 2) { off_lo.push_back(moff[q]); off_hi.push_back(moff[q + 1]); slot_t.push_back(mslot[q] / 2); }
        const size_t nq = slot_t.size();
        d_moff.ensure(3 * nq);
        HIP_CHECK(hipMemcpyAsync(d_moff.p, off_lo.data(), sizeof(i64) * nq, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_moff.p + nq, off_hi.data(), sizeof(i64) * nq, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_moff.p + 2 * nq, slot_t.data(), sizeof(i64) * nq, hipMemcpyHostToDevice, st));
        k_gather_means_slots(c, c->lm_means.p, d_moff.p, d_moff.p + 2 * nq, (i64)nq, d, m_per, 5, Mg);
        k_gather_means_slots(c, c->lm_means.p, d_moff.p + nq, d_moff.p + 2 * nq, (i64)nq, d, m_per, 5 + d, Mg);
        HIP_CHECK(hipStreamSynchronize(st)); // off_lo / off_hi / slot_t go out of scope
    }
    cge_allreduce_dev(c, X, x_need + W, 2);
    {
        std::vector<double> slots((size_t)W);
        HIP_CHECK(hipMemcpyAsync(slots.data(), X + x_need, sizeof(double) * W, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        throw_if_a_rank_failed(c, err, slots.data(), "runsplit (sharded batch)");
    }
    // every rank: the same ranges and means in the batch's order, the gathered values
    c->lm_arena_used = A0;
    c->lm_means_used = M0;
    const i64 base = arena_alloc(c, R), mbase = means_alloc(c, T * 2 * d);
    HIP_CHECK(hipMemcpyAsync(c->lm_arena.p + base, S, sizeof(i32) * R, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpy2DAsync(c->lm_means.p + mbase, sizeof(double) * 2 * d, Mg + 5, sizeof(double) * m_per, sizeof(double) * 2 * d,
                               (size_t)T, hipMemcpyDeviceToDevice, st));
    std::vector<double> all((size_t)T * 5);
    HIP_CHECK(hipMemcpy2DAsync(all.data(), sizeof(double) * 5, Mg, sizeof(double) * m_per, sizeof(double) * 5, (size_t)T,
                               hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (i64 t = 0; t < T; t++) {
        Group *g = batch[t];
        g->has_split = true;
        g->rc = (int)all[5 * t];
        if (g->rc != CGE_OK) continue;
        g->coff = base + prefix[t];
        g->nlow = (i64)all[5 * t + 1];
        g->vlow = all[5 * t + 2];
        g->vhigh = all[5 * t + 3];
        g->cmean_off = all[5 * t + 4] != 0.0 ? mbase + t * 2 * d : -1;
    }
}

void throw_rc(int rc) {
    switch (rc) {
    case CGE_E_HOMOGENEOUS: CGE_THROW(rc, "Trying to split homogenous cluster");
    case CGE_E_EMPTY_CLUSTER: CGE_THROW(rc, "Unexpected empty cluster generated");
    default: CGE_THROW(rc, "AssertionError: size(m, 1) > 1");
    }
}

// create the child groups of every freshly split task (single-threaded: the pool is not thread-safe)
void materialise_children(std::vector<Group *> &tasks, GroupPool &pool, i64 c_d) {
    for (Group *g : tasks) {
        if (g->rc != CGE_OK || g->clo) continue;
        pool.emplace_back();
        g->clo = &pool.back();
        g->clo->off = g->coff;
        g->clo->len = g->nlow;
        g->clo->value = g->vlow;
        g->clo->mean_off = g->cmean_off;
        g->clo->parent = g;
        g->clo->owner = g->owner;
        pool.emplace_back();
        g->chi = &pool.back();
        g->chi->off = g->coff >= 0 ? g->coff + g->nlow : -1; // (-1: another rank's rows, option shard_rows)
        g->chi->len = g->len - g->nlow;
        g->chi->value = g->vhigh;
        g->chi->mean_off = g->cmean_off >= 0 ? g->cmean_off + c_d : -1;
        g->chi->parent = g;
        g->chi->owner = g->owner;
    }
}

// pop the top group of `h`, push its two children (the body of src/landmarks.jl:290-307 / :317-334)
void replay_one(Heap &h) {
    Group *g = h.pop();
    if (g->rc != CGE_OK) throw_rc(g->rc);
    h.put(g->clo);
    h.put(g->chi);
}

// Bring every heap to its target length.  Each round: replay as far as the cached splits allow; then split,
// in ONE device batch, the most valuable unsplit nodes KNOWN so far -- heap members and descendants of cached
// splits alike (speculative tree expansion) -- as many per heap as pops are still missing.  The number of
// rounds is the depth of the relevant split tree, not the length of the pop sequence; the replay order (and
// with it every landmark id) is exactly the reference's.
// `speculate` = false: only the current top of each heap is split per round (no wasted splits; used for the
// many small per-community heaps of the forced phase, which need s-1 rounds anyway).
void advance_heaps(cge_ctx *c, std::vector<Heap *> &heaps, const std::vector<i64> &targets, int method,
                   GroupPool &pool, bool speculate) {
    std::vector<Group *> frontier, stack, keep; // (reused by every round: no allocation per round)
    std::vector<double> vals;
    for (;;) {
        PhaseAcc *ph = new PhaseAcc(c, "lm_heap"); // replay + choice of the next batch (host)
        std::vector<Group *> batch;
        for (size_t q = 0; q < heaps.size(); q++) {
            Heap &h = *heaps[q];
            while ((i64)h.len() < targets[q] && h.top()->has_split) replay_one(h);
            if ((i64)h.len() >= targets[q]) continue;
            if (!speculate) {
                batch.push_back(h.top());
                continue;
            }
            const i64 remaining = targets[q] - (i64)h.len();
            frontier.clear(); stack.clear(); vals.clear();
            // Exactly `remaining` more pops will happen.  A node can only be among them if its value ranks within
            // `remaining` among ALL known unpopped nodes (cached splits and unsplit ones alike): nodes still to be
            // discovered only add competitors.  So every unsplit node above that threshold is a candidate and
            // every one below it is certainly never popped.
            for (size_t i = 1; i <= h.len(); i++) stack.push_back(h.at(i));
            while (!stack.empty()) {
                Group *g = stack.back();
                stack.pop_back();
                if (g->len > 1 || g == h.top() || g->has_split) vals.push_back(g->value);
                if (!g->has_split) {
                    if (g->len > 1 || g == h.top()) frontier.push_back(g);
                } else if (g->rc == CGE_OK) {
                    stack.push_back(g->clo);
                    stack.push_back(g->chi);
                }
            }
            if (frontier.empty()) continue;
            double thr = INFINITY;
            if ((i64)vals.size() > remaining) {
                std::nth_element(vals.begin(), vals.begin() + (remaining - 1), vals.end());
                thr = vals[remaining - 1];
            }
            {
                keep.clear();
                for (Group *g : frontier)
                    if (g->value <= thr || g == h.top()) keep.push_back(g);
                frontier.swap(keep);
            }
            // Every candidate COULD be popped, but a good half of a large frontier never is (its competitors' children
            // outrank it).  Splitting only the most valuable part per round costs a round or two more and saves the
            // eigen-problems of the rest; what is left over is reconsidered, with more known, in the next round.
            // (never fewer than 256 at a time: the last pops would otherwise trickle through many tiny rounds)
            const int spec_pct = (method == CGE_METHOD_SIZE || method == CGE_METHOD_DIAMETER) ? 10
                                 : (c->d > 128 ? 25 : 40); // wide embeddings: a wasted split costs a memory-resident eigen-problem
            i64 take = std::max<i64>(std::min<i64>(remaining, 256), (i64)((double)remaining * spec_pct / 100.0));
            // The register-resident eigen-solver of 64 < d <= 128 holds two matrices per CU: 512 at a time, and a batch of 800
            // costs two rounds (0.98 ms) where 512 cost one (0.51).  The batch is cut DOWN to a multiple of 512: one round more
            // at the headline (8 batches), eigen-solver 4.7 -> 4.0 ms, landmarks 13.1 -> 12.3 ms (rounding up: 13.6).
            if (c->d > 64 && c->d <= 128 && take > 512) take = take / 512 * 512;
            if ((i64)frontier.size() > take) {
                std::nth_element(frontier.begin(), frontier.begin() + (take - 1), frontier.end(),
                                 [](const Group *a, const Group *b) { return a->value < b->value; });
                const double cut = frontier[take - 1]->value;
                keep.clear();
                for (Group *g : frontier)
                    if (g->value <= cut || g == h.top()) keep.push_back(g);
                frontier.swap(keep);
            }
            batch.insert(batch.end(), frontier.begin(), frontier.end());
        }
        delete ph;
        if (batch.empty()) break;
        if (speculate) compute_splits_sharded(c, batch, method); // the global phase (N > 1: split over the ranks)
        else compute_splits(c, batch, method);
        PhaseAcc pm(c, "lm_materialise");
        materialise_children(batch, pool, c->d);
    }
}

} // namespace

// group_ids[i] = 0-based group (= heap position - 1) of vertex i; also leaves on the device c->v2l (the same, int32)
// and the landmark -> members index c->lm_memoff / c->lm_mem (ascending inside a landmark), mirrored in
// c->h_mem_off / c->h_mem when `want_index`.
void host_runsplit(cge_ctx *c, const i64 *cl_flat, const i64 *cl_off, i64 ncl, i64 nland, i64 forced, int method,
                   std::vector<i64> &group_ids, bool want_index) {
    const i64 n = c->n, d = c->d;
    hipStream_t st = c->stream;
    const bool RS = c->rows_sharded; // option shard_rows: this rank holds (and splits) the rows of its own communities only
    const int me = RS ? c->coll.rank : 0;
    if (!c->Xr.p || c->Xr.n < (size_t)(lm_rows(c) * d) || (i64)c->h_vw.size() != n)
        CGE_THROW(CGE_E_ARG, "runsplit: embedding / vertex weights are not resident");
    GroupPool pool;
    Heap H;
    PhaseAcc *pinit = new PhaseAcc(c, "lm_init");
    // sort(initial_clusters): lexicographic (:281)
    std::vector<i64> order(ncl);
    for (i64 i = 0; i < ncl; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](i64 a, i64 b) {
        return std::lexicographical_compare(cl_flat + cl_off[a], cl_flat + cl_off[a + 1], cl_flat + cl_off[b],
                                            cl_flat + cl_off[b + 1]);
    });
    // the member arena starts as the clusters themselves (0-based), cluster q at cl_off[q]
    const i64 total = cl_off[ncl];
    c->lm_arena_used = 0;
    c->lm_means_used = 0;
    c->stat_lm_batches = c->stat_lm_rows = c->stat_lm_splits = 0;
    c->cut_ties.ensure(1);
    HIP_CHECK(hipMemsetAsync(c->cut_ties.p, 0, sizeof(int), c->stream));
    // option shard_rows: cl_owner[q] = the rank that holds cluster q's rows (the owner of its first member's community; a
    // cluster has to lie inside one rank's rows), a_off[q] = where an OWNED cluster starts in this rank's arena (local ids)
    std::vector<int> cl_owner;
    std::vector<i64> a_off_v;
    if (RS) {
        cl_owner.assign(ncl, 0);
        a_off_v.assign(ncl + 1, 0);
        int bad = 0;
        for (i64 q = 0; q < ncl; q++) {
            a_off_v[q + 1] = a_off_v[q];
            if (cl_off[q + 1] <= cl_off[q]) continue;
            const i64 v0 = cl_flat[cl_off[q]];
            if (v0 < 1 || v0 > n) { bad = 1; continue; }
            cl_owner[q] = c->comm_owner[c->h_comm[v0 - 1]];
            if (cl_owner[q] == me) a_off_v[q + 1] += cl_off[q + 1] - cl_off[q];
        }
        c->pin_rows[0].ensure(std::max<i64>(a_off_v[ncl], 1));
        i32 *stage = c->pin_rows[0].p;
        std::atomic<int> abad{bad};
        parallel_for(c, ncl, [&](i64 q) {
            const bool mine = cl_owner[q] == me;
            for (i64 t = cl_off[q]; t < cl_off[q + 1]; t++) {
                const i64 v = cl_flat[t];
                if (v < 1 || v > n) { abad.store(1); continue; }
                const i32 l = c->h_glob2loc[v - 1];
                if (mine != (l >= 0)) { abad.store(2); continue; } // a cluster that spans two ranks' rows
                if (mine) stage[a_off_v[q] + (t - cl_off[q])] = l;
            }
        });
        // every rank must leave by the same door: the verdicts are exchanged before anybody throws
        std::vector<double> verdict(1, (double)abad.load());
        {
            DevBuf<double> &X = c->samp_xchg;
            X.ensure(1);
            HIP_CHECK(hipMemcpyAsync(X.p, verdict.data(), sizeof(double), hipMemcpyHostToDevice, st));
            cge_allreduce_dev(c, X.p, 1, 1);
            HIP_CHECK(hipMemcpyAsync(verdict.data(), X.p, sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        if (verdict[0] == 1.0) CGE_THROW(CGE_E_ARG, "cluster member out of range 1..%lld", (long long)n);
        if (verdict[0] != 0.0)
            CGE_THROW(CGE_E_ARG, "option shard_rows: a cluster spans the rows of two ranks (the rows are sharded by the community "
                                 "vector of cge_set_vertex_data: clusters must refine it)");
        if (a_off_v[ncl] > 0) {
            arena_alloc(c, a_off_v[ncl]);
            HIP_CHECK(hipMemcpyAsync(c->lm_arena.p, stage, sizeof(i32) * a_off_v[ncl], hipMemcpyHostToDevice, st));
        }
    } else {
        c->pin_rows[0].ensure(total);
        i32 *stage = c->pin_rows[0].p;
        std::atomic<int> bad{0};
        parallel_for(c, 64, [&](i64 part) {
            for (i64 q = total * part / 64; q < total * (part + 1) / 64; q++) {
                if (cl_flat[q] < 1 || cl_flat[q] > n) { bad.store(1); continue; }
                stage[q] = (i32)(cl_flat[q] - 1);
            }
        });
        if (bad.load()) CGE_THROW(CGE_E_ARG, "cluster member out of range 1..%lld", (long long)n);
        arena_alloc(c, total);
        HIP_CHECK(hipMemcpyAsync(c->lm_arena.p, stage, sizeof(i32) * total, hipMemcpyHostToDevice, st));
    }
    // where cluster q starts in this rank's arena (-1: another rank's rows)
    auto a_off = [&](i64 q) -> i64 { return !RS ? cl_off[q] : (cl_owner[q] == me ? a_off_v[q] : -1); };

    // ---- forced per-community phase (:282-313): every big community owns a local heap -----------
    struct Local { Heap h; i64 pos; };
    std::vector<Local> locals;
    bool sharded_forced = false;
    // global-heap insertion order must follow the sorted community order, so first split all the
    // local heaps (independent of each other), then insert community by community.
    for (i64 q = 0; q < ncl; q++) {
        const i64 cidx = order[q], len = cl_off[cidx + 1] - cl_off[cidx];
        if (len > forced) {
            locals.push_back(Local{Heap(), q});
            pool.emplace_back();
            Group *g = &pool.back();
            g->off = a_off(cidx);
            g->len = len;
            g->owner = RS ? cl_owner[cidx] : 0;
            locals.back().h.put(g);
        }
    }
    delete pinit;
    if (!locals.empty()) {
        if (forced <= 1) { // with forced >= 2 every root is popped from its one-element heap: its value is never compared
            PhaseAcc pa(c, "lm_roots");
            std::vector<Group *> roots;
            for (auto &L : locals)
                if (!RS || L.h.top()->owner == me) roots.push_back(L.h.top());
            device_group_values(c, roots);
            // (option shard_rows: the other ranks' roots get their values with the forced phase's exchange below)
        }
        // N > 1: the local heaps are independent of each other, so every rank splits its share of the communities
        // (balanced by rows) and the results -- member lists in pop order, lengths, heap values, means -- are gathered by ONE
        // all-reduce into zero-filled buffers (op 2: integer sum of the words, exact).  Afterwards EVERY rank, owner or not,
        // rebuilds the groups from the gathered data, so all ranks continue from identical state.
        const i64 nbig = (i64)locals.size(), W = c->has_coll ? c->coll.world : 1;
        const i64 s_words = (total + 1) / 2, m_per = 3 + d; // per group: length, value, mean flag, mean
        const i64 x_need = s_words + nbig + nbig * forced * m_per;
        const bool shard = !RS && W > 1 && c->opt_shard_forced && forced >= 2 && cge_exchange_fits(c, (size_t)(x_need + W));
        std::vector<int> owner(nbig, 0);
        if (RS) {
            // option shard_rows: every rank runs the local heaps of ITS communities; the replicated global heap needs, of every
            // group they end with, its length and value only (pop order): one gather of 1 + 2 max(forced, 1) words per community
            const i64 per = 1 + 2 * std::max<i64>(forced, 1);
            std::vector<Heap *> hs;
            std::vector<i64> tg;
            for (i64 b = 0; b < nbig; b++) {
                owner[b] = locals[b].h.top()->owner;
                if (owner[b] == me) { hs.push_back(&locals[b].h); tg.push_back(forced); }
            }
            std::vector<double> w((size_t)nbig * per, 0.0);
            std::vector<std::vector<Group *>> popped(nbig);
            // (a reference error -- "Trying to split homogenous cluster", an empty child -- is raised by the owner of the
            // community alone: it must reach the other ranks, who would otherwise wait in the exchange for ever)
            const RankError err = guarded_work([&] {
                if (!hs.empty()) advance_heaps(c, hs, tg, method, pool, false);
                for (i64 b = 0; b < nbig; b++) {
                    if (owner[b] != me) continue;
                    Heap &L = locals[b].h;
                    if ((i64)L.len() > std::max<i64>(forced, 1)) CGE_THROW(CGE_E_ASSERT, "forced phase: a local heap grew beyond its target");
                    while (L.len() > 0) popped[b].push_back(L.pop()); // pop order = the order in which the global heap receives them (:309-312)
                    w[b * per] = (double)popped[b].size();
                    for (size_t s_ = 0; s_ < popped[b].size(); s_++) {
                        w[b * per + 1 + 2 * s_] = (double)popped[b][s_]->len;
                        w[b * per + 2 + 2 * s_] = popped[b][s_]->value;
                    }
                }
            });
            exchange_group_words_checked(c, w, err, "forced phase (shard_rows)");
            for (i64 b = 0; b < nbig; b++) {
                Heap &L = locals[b].h;
                L = Heap();
                const i64 cnt = (i64)w[b * per];
                if (cnt < 1 || cnt > std::max<i64>(forced, 1)) CGE_THROW(CGE_E_COLLECTIVE, "forced phase (shard_rows): no rank answered for a community of rank %d", owner[b]);
                for (i64 s_ = 0; s_ < cnt; s_++) {
                    Group *g;
                    if (owner[b] == me)
                        g = popped[b][s_];
                    else {
                        pool.emplace_back();
                        g = &pool.back();
                        g->off = -1;
                        g->len = (i64)w[b * per + 1 + 2 * s_];
                        g->value = w[b * per + 2 + 2 * s_];
                        g->owner = owner[b];
                    }
                    L.a.push_back(Heap::Ent{g->value, g}); // kept in pop order: the merge below reads the array front to back
                }
            }
            sharded_forced = true;
        }
        if (shard) { // longest first, each to the least loaded rank (deterministic)
            std::vector<i64> ord(nbig), load(W, 0);
            for (i64 b = 0; b < nbig; b++) ord[b] = b;
            std::stable_sort(ord.begin(), ord.end(), [&](i64 a, i64 b) { return locals[a].h.top()->len > locals[b].h.top()->len; });
            for (i64 b : ord) {
                const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                owner[b] = r;
                load[r] += locals[b].h.top()->len;
            }
        }
        const int me_f = shard ? c->coll.rank : 0;
        RankError ferr;
        if (!RS) {
            std::vector<Heap *> hs;
            std::vector<i64> tg;
            for (i64 b = 0; b < nbig; b++)
                if (owner[b] == me_f) { hs.push_back(&locals[b].h); tg.push_back(forced); }
            // (sharded: an error of this rank's heaps -- the reference's own "Trying to split homogenous cluster" included --
            // travels with the exchange below, so that every rank stops instead of waiting for this one)
            ferr = guarded_work([&] { if (!hs.empty()) advance_heaps(c, hs, tg, method, pool, false); });
            if (!shard && ferr.code) throw CgeError{ferr.code, ferr.msg};
        }
        if (shard) {
            const int me = me_f;
            PhaseAcc px(c, "lm_exchange");
            double *X = c->xptr;
            i32 *S = reinterpret_cast<i32 *>(X);
            double *Mc = X + s_words, *Mg = Mc + nbig; // counts per community; per group {len, value, flag, mean[d]}
            HIP_CHECK(hipMemsetAsync(X, 0, sizeof(double) * (x_need + W), st)); // (W verdict slots behind the words)
            std::vector<i64> seg, moff;
            std::vector<double> hc(nbig, 0.0), hg((size_t)nbig * forced * 3, 0.0);
            std::vector<i64> gslot; // slot (b * forced + s) of every owned group, in the order of `moff`
            if (!ferr.code) ferr = guarded_work([&] {
            for (i64 b = 0; b < nbig; b++) {
                if (owner[b] != me) continue;
                Heap &L = locals[b].h;
                const i64 cidx = order[locals[b].pos];
                i64 at = cl_off[cidx], s_ = 0;
                if ((i64)L.len() > forced) CGE_THROW(CGE_E_ASSERT, "forced phase: a local heap grew beyond its target");
                hc[b] = (double)L.len();
                while (L.len() > 0) { // pop order = the order in which the global heap receives them (:309-312)
                    Group *g = L.pop();
                    seg.push_back(g->off); seg.push_back(at); seg.push_back(g->len);
                    const size_t sl = (size_t)(b * forced + s_);
                    hg[3 * sl] = (double)g->len;
                    hg[3 * sl + 1] = g->value;
                    hg[3 * sl + 2] = g->mean_off >= 0 ? 1.0 : 0.0;
                    if (g->mean_off >= 0) { moff.push_back(g->mean_off); gslot.push_back((i64)sl); }
                    at += g->len;
                    s_++;
                }
                if (at != cl_off[cidx + 1]) CGE_THROW(CGE_E_ASSERT, "forced phase: groups do not cover their community");
            }
            });
            const double my_verdict = (double)ferr.code;
            if (ferr.code) {
                HIP_CHECK(hipMemcpyAsync(X + x_need + me, &my_verdict, sizeof(double), hipMemcpyHostToDevice, st));
                seg.clear(); moff.clear(); gslot.clear();
                std::fill(hc.begin(), hc.end(), 0.0);
            }
            // member lists -> their community's range of S; {len, value, flag} and the means -> Mg
            DevBuf<i64> d_seg, d_moff;
            if (!seg.empty()) {
                d_seg.ensure(seg.size());
                HIP_CHECK(hipMemcpyAsync(d_seg.p, seg.data(), sizeof(i64) * seg.size(), hipMemcpyHostToDevice, st));
                k_copy_segments(c, c->lm_arena.p, d_seg.p, (i64)seg.size() / 3, S);
            }
            HIP_CHECK(hipMemcpyAsync(Mc, hc.data(), sizeof(double) * nbig, hipMemcpyHostToDevice, st));
            // the three scalars of every slot (strided into Mg) and the means
            HIP_CHECK(hipMemcpy2DAsync(Mg, sizeof(double) * m_per, hg.data(), sizeof(double) * 3, sizeof(double) * 3,
                                       (size_t)nbig * forced, hipMemcpyHostToDevice, st));
            if (!moff.empty()) {
                d_moff.ensure(2 * moff.size());
                HIP_CHECK(hipMemcpyAsync(d_moff.p, moff.data(), sizeof(i64) * moff.size(), hipMemcpyHostToDevice, st));
                HIP_CHECK(hipMemcpyAsync(d_moff.p + moff.size(), gslot.data(), sizeof(i64) * gslot.size(), hipMemcpyHostToDevice, st));
                k_gather_means_slots(c, c->lm_means.p, d_moff.p, d_moff.p + moff.size(), (i64)moff.size(), d, m_per, 3, Mg);
            }
            cge_allreduce_dev(c, X, x_need + W, 2);
            {
                std::vector<double> slots((size_t)W);
                HIP_CHECK(hipMemcpyAsync(slots.data(), X + x_need, sizeof(double) * W, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                throw_if_a_rank_failed(c, ferr, slots.data(), "forced phase (sharded)");
            }
            // every rank: communities back into the start of the arena, one means block, fresh groups
            HIP_CHECK(hipMemcpyAsync(c->lm_arena.p, S, sizeof(i32) * total, hipMemcpyDeviceToDevice, st));
            c->lm_arena_used = total;
            std::vector<double> all((size_t)nbig + (size_t)nbig * forced * m_per);
            HIP_CHECK(hipMemcpyAsync(all.data(), Mc, sizeof(double) * all.size(), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            c->lm_means_used = 0;
            const i64 mbase = means_alloc(c, nbig * forced * d);
            HIP_CHECK(hipMemcpy2DAsync(c->lm_means.p + mbase, sizeof(double) * d, Mg + 3, sizeof(double) * m_per, sizeof(double) * d,
                                       (size_t)nbig * forced, hipMemcpyDeviceToDevice, st));
            for (i64 b = 0; b < nbig; b++) {
                Heap &L = locals[b].h;
                L = Heap(); // (an unsplit root of another rank's community, or empty after the pops above)
                const i64 cidx = order[locals[b].pos], cnt = (i64)all[b];
                i64 at = cl_off[cidx];
                for (i64 s_ = 0; s_ < cnt; s_++) {
                    const double *rec = &all[(size_t)nbig + (size_t)(b * forced + s_) * m_per];
                    pool.emplace_back();
                    Group *g = &pool.back();
                    g->off = at;
                    g->len = (i64)rec[0];
                    g->value = rec[1];
                    g->mean_off = rec[2] != 0.0 ? mbase + (b * forced + s_) * d : -1;
                    at += g->len;
                    L.a.push_back(Heap::Ent{g->value, g}); // kept in pop order: the merge below reads the array front to back
                }
                if (at != cl_off[cidx + 1]) CGE_THROW(CGE_E_ASSERT, "forced phase: gathered groups do not cover their community");
            }
            HIP_CHECK(hipStreamSynchronize(st)); // seg / moff staging goes out of scope
            sharded_forced = true;
        }
    }
    PhaseAcc *pmerge = new PhaseAcc(c, "lm_merge");
    size_t li = 0;
    for (i64 q = 0; q < ncl; q++) {
        const i64 cidx = order[q], len = cl_off[cidx + 1] - cl_off[cidx];
        if (len <= forced) {
            for (i64 t = cl_off[cidx]; t < cl_off[cidx + 1]; t++) {
                pool.emplace_back();
                Group *g = &pool.back();
                g->off = a_off(cidx) >= 0 ? a_off(cidx) + (t - cl_off[cidx]) : -1;
                g->len = 1;
                g->value = DBL_EPSILON; // eps() (:284)
                g->owner = RS ? cl_owner[cidx] : 0;
                H.put(g);
            }
        } else {
            Heap &L = locals[li++].h;
            if (sharded_forced) { // already in pop order
                for (size_t q2 = 1; q2 < L.a.size(); q2++) H.put(L.a[q2].g);
            } else
                while (L.len() > 0) H.put(L.pop()); // :309-312
        }
    }
    delete pmerge;
    // ---- global phase (:316-335) ----------------------------------------------------------------------------
    {
        std::vector<Heap *> hs{&H};
        std::vector<i64> tg{nland};
        advance_heaps(c, hs, tg, method, pool, true);
    }
    // ---- the heap array is the numbering (:337-342): v2l and the landmark index, on the device -----------------------
    PhaseAcc pfin(c, "lm_final");
    const i64 NG = (i64)H.len();
    const i64 nrows = lm_rows(c);
    c->pin_small.ensure((size_t)2 * NG + 2);
    i32 *goff = c->pin_small.p, *glen = goff + NG;
    c->h_mem_off.assign(NG + 1, 0);
    c->h_gl_off.assign(NG + 1, 0);
    c->lm_owner.assign(NG, 0);
    for (i64 g = 0; g < NG; g++) {
        const Group *G = H.at(g + 1);
        const bool mine = !RS || G->owner == me;
        goff[g] = mine ? (i32)G->off : 0; // (option shard_rows: another rank's group is an empty range here)
        glen[g] = mine ? (i32)G->len : 0;
        c->h_mem_off[g + 1] = c->h_mem_off[g] + glen[g];
        c->h_gl_off[g + 1] = c->h_gl_off[g] + (i32)G->len;
        c->lm_owner[g] = G->owner;
    }
    DevBuf<i32> &v2l_rows = RS ? c->v2l_loc : c->v2l; // landmark of every row this rank holds
    c->lm_goff.ensure(NG); c->lm_glen.ensure(NG); c->v2l.ensure(n); v2l_rows.ensure(nrows); c->lm_mem.ensure(nrows); c->lm_memoff.ensure(NG + 1);
    HIP_CHECK(hipMemcpyAsync(c->lm_goff.p, goff, sizeof(i32) * NG, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(c->lm_glen.p, glen, sizeof(i32) * NG, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(c->lm_memoff.p, c->h_mem_off.data(), sizeof(i32) * (NG + 1), hipMemcpyHostToDevice, st));
    const i64 unassigned = k_groups_to_index(c, c->lm_arena.p, c->lm_goff.p, c->lm_glen.p, NG, nrows, v2l_rows.p, c->lm_mem.p);
    bool cover_bad = unassigned != 0 || c->h_mem_off[NG] != nrows || c->h_gl_off[NG] != n;
    if (RS) {
        // v_to_l of ALL vertices on every rank (4 bytes per vertex: the table of the per-edge passes and of the local score):
        // every rank writes landmark + 1 of its rows at their global ids into a zero-filled vector, the ranks add the words
        const i64 words = (n + 1) / 2;
        DevBuf<i32> &tmp = c->sort_idx2;
        tmp.ensure(2 * words);
        HIP_CHECK(hipMemsetAsync(tmp.p, 0, sizeof(i32) * 2 * words, st));
        k_scatter_i32(c, v2l_rows.p, c->loc2glob.p, nrows, 1, tmp.p);
        cge_allreduce_dev(c, reinterpret_cast<double *>(tmp.p), words, 2);
        k_add_i32(c, tmp.p, n, -1, c->v2l.p);
        std::vector<double> verdict(1, cover_bad ? 1.0 : 0.0);
        exchange_group_words(c, verdict); // (sum: any rank's failure fails all)
        cover_bad = verdict[0] != 0.0;
    }
    if (cover_bad)
        CGE_THROW(CGE_E_ASSERT, "AssertionError: all(>=(0), group_ids) [%lld rows without a group, %lld of %lld rows in the groups here, %lld of %lld over all ranks]",
                  (long long)unassigned, (long long)c->h_mem_off[NG], (long long)nrows, (long long)c->h_gl_off[NG], (long long)n); // :343
    c->lm_index_on_device = want_index;
    c->h_mem.clear(); // the member lists stay on the device (c->lm_mem); host copies are made by whoever asks for them
    if (want_index) { // the fused path: v2l stays on the device too (landmarks_fetch copies it when it is asked for)
        HIP_CHECK(hipStreamSynchronize(st)); // goff / glen are pinned staging of this call
        group_ids.clear();
        return;
    }
    c->h_v2l0.resize(n);
    HIP_CHECK(hipMemcpyAsync(c->h_v2l0.data(), c->v2l.p, sizeof(i32) * n, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    group_ids.resize(n);
    parallel_for(c, 64, [&](i64 part) {
        for (i64 i = n * part / 64; i < n * (part + 1) / 64; i++) group_ids[i] = c->h_v2l0[i];
    });
}
