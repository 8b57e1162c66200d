// textio.cpp -- parallel reader for the numeric text tables that parseargs() hands to the hot path
// (reference: `readdlm` calls at src/auxilary.jl:80-168).  Host only, no GPU.
//
// The three inputs of CGE_CLI.jl are whitespace-delimited numeric tables: the edge list (2 or 3 columns), the
// community file (1 or 2 columns) and the embedding (d or d+1 columns, optionally with node2vec's "n d" header line).
// At the sizes of BASELINE.json a single-threaded `readdlm` of a multi-GB embedding takes longer than the whole
// scoring pass on the GPU, so the file is mapped, cut at line boundaries into one piece per thread, and every piece
// is parsed with std::from_chars (exactly rounded, locale-free).  What `readdlm` does is kept: blank lines are
// skipped, any run of spaces / tabs (or a comma) separates fields, every row must have the same number of fields.
// A first line whose field count differs from the second line's is taken as a header and skipped (the reference
// retries with `skipstart = 1`, src/auxilary.jl:151-156).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cge_hip.h"

namespace {

struct TextTable {
    std::string path;
    const char *buf = nullptr; // the mapped file
    size_t len = 0;
    int64_t rows = 0, cols = 0;
    int header_skipped = 0;
    std::vector<const char *> cut; // piece t = [cut[t], cut[t+1]), starts after a newline
    std::vector<int64_t> row0;     // first data row of piece t
    ~TextTable() {
        if (buf) munmap((void *)buf, len);
    }
};

inline bool is_sep(char ch) { return ch == ' ' || ch == '\t' || ch == ',' || ch == '\r'; }

// number of fields of the line [p, e)
int64_t count_fields(const char *p, const char *e) {
    int64_t n = 0;
    while (p < e) {
        while (p < e && is_sep(*p)) p++;
        if (p >= e) break;
        n++;
        while (p < e && !is_sep(*p)) p++;
    }
    return n;
}
inline bool blank_line(const char *p, const char *e) {
    while (p < e && is_sep(*p)) p++;
    return p >= e;
}
// parse the fields of one line into out[0], out[stride], ...; false on a malformed number or a wrong field count
bool parse_line(const char *p, const char *e, int64_t cols, double *out, int64_t stride) {
    int64_t c = 0;
    while (p < e) {
        while (p < e && is_sep(*p)) p++;
        if (p >= e) break;
        if (c >= cols) return false;
        const char *q = p;
        while (q < e && !is_sep(*q)) q++;
        const char *b = (*p == '+') ? p + 1 : p;
        double v = 0.0;
        const auto res = std::from_chars(b, q, v);
        if (res.ec != std::errc() || res.ptr != q) return false;
        out[c++ * stride] = v;
        p = q;
    }
    return c == cols;
}

template <class F>
void for_pieces(int P, F &&fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < P; t++) th.emplace_back(fn, t);
    fn(0);
    for (auto &x : th) x.join();
}

// map the file, find the column count / header, cut it into pieces and count the data rows of each
int scan_table(const char *path, TextTable &T, std::string &err, int n_threads) {
    T.path = path;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { err = T.path + " is not a file"; return CGE_E_ARG; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); err = T.path + " is not a file"; return CGE_E_ARG; }
    const size_t len = (size_t)st.st_size;
    if (len == 0) { close(fd); err = T.path + " is empty"; return CGE_E_ARG; }
    const char *buf = (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (buf == MAP_FAILED) { err = "cannot map " + T.path; return CGE_E_ARG; }
    T.buf = buf;
    T.len = len;
    const char *end = buf + len;
    // first two non-blank lines: column count and header detection
    const char *p = buf, *first = nullptr, *first_e = nullptr, *second = nullptr, *second_e = nullptr;
    while (p < end && !second) {
        const char *e = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!e) e = end;
        if (!blank_line(p, e)) {
            if (!first) { first = p; first_e = e; } else { second = p; second_e = e; }
        }
        p = e < end ? e + 1 : end;
    }
    if (!first) { err = T.path + " holds no data"; return CGE_E_ARG; }
    const char *body = buf;
    T.cols = count_fields(first, first_e);
    if (second && count_fields(second, second_e) != T.cols) { // header line (node2vec: "n d")
        T.cols = count_fields(second, second_e);
        T.header_skipped = 1;
        body = first_e < end ? first_e + 1 : end;
    }
    const int P = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, (size_t)(end - body) / (1 << 16) + 1));
    T.cut.assign(P + 1, end);
    T.cut[0] = body;
    for (int t = 1; t < P; t++) {
        const char *q = body + (size_t)(end - body) * t / P;
        const char *e = (const char *)memchr(q, '\n', (size_t)(end - q));
        T.cut[t] = e ? e + 1 : end;
    }
    for (int t = 1; t <= P; t++) T.cut[t] = std::max(T.cut[t], T.cut[t - 1]);
    std::vector<int64_t> nlines(P, 0);
    for_pieces(P, [&](int t) {
        int64_t n = 0;
        for (const char *q = T.cut[t]; q < T.cut[t + 1];) {
            const char *e = (const char *)memchr(q, '\n', (size_t)(T.cut[t + 1] - q));
            if (!e) e = T.cut[t + 1];
            if (!blank_line(q, e)) n++;
            q = e + 1;
        }
        nlines[t] = n;
    });
    T.row0.assign(P + 1, 0);
    for (int t = 0; t < P; t++) T.row0[t + 1] = T.row0[t] + nlines[t];
    T.rows = T.row0[P];
    return CGE_OK;
}

// parse every piece straight into the caller's matrix (row-major, or column-major = Julia's Matrix{Float64})
int parse_table(const TextTable &T, double *out, bool column_major, std::string &err) {
    const int P = (int)T.cut.size() - 1;
    const int64_t R = T.rows, Cn = T.cols;
    std::vector<int64_t> bad(P, -1);
    for_pieces(P, [&](int t) {
        int64_t r = T.row0[t];
        for (const char *q = T.cut[t]; q < T.cut[t + 1];) {
            const char *e = (const char *)memchr(q, '\n', (size_t)(T.cut[t + 1] - q));
            if (!e) e = T.cut[t + 1];
            if (!blank_line(q, e)) {
                const bool ok = column_major ? parse_line(q, e, Cn, out + r, R) : parse_line(q, e, Cn, out + (size_t)r * Cn, 1);
                if (!ok && bad[t] < 0) bad[t] = r;
                r++;
            }
            q = e + 1;
        }
    });
    for (int t = 0; t < P; t++)
        if (bad[t] >= 0) {
            char msg[64];
            snprintf(msg, sizeof(msg), ": data row %lld is not %lld numeric fields", (long long)(bad[t] + 1), (long long)Cn);
            err = T.path + msg;
            return CGE_E_ARG;
        }
    return CGE_OK;
}

} // namespace

extern "C" {

int cge_text_table_open(const char *path, int n_threads, int64_t *rows, int64_t *cols, int *header_skipped, void **handle,
                        char *err, int64_t err_len) {
    if (!path || !rows || !cols || !handle) return CGE_E_ARG;
    *handle = nullptr;
    TextTable *T = new (std::nothrow) TextTable();
    if (!T) return CGE_E_OOM;
    std::string msg;
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    int rc;
    try {
        rc = scan_table(path, *T, msg, n_threads);
    } catch (const std::bad_alloc &) {
        rc = CGE_E_OOM;
        msg = "out of host memory";
    }
    if (rc != CGE_OK) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", msg.c_str());
        delete T;
        return rc;
    }
    *rows = T->rows;
    *cols = T->cols;
    if (header_skipped) *header_skipped = T->header_skipped;
    *handle = T;
    return CGE_OK;
}

int cge_text_table_parse(void *handle, double *out, int column_major, char *err, int64_t err_len) {
    TextTable *T = (TextTable *)handle;
    if (!T || !out) return CGE_E_ARG;
    std::string msg;
    int rc;
    try {
        rc = parse_table(*T, out, column_major != 0, msg);
    } catch (const std::bad_alloc &) {
        rc = CGE_E_OOM;
        msg = "out of host memory";
    }
    if (rc != CGE_OK && err && err_len > 0) snprintf(err, (size_t)err_len, "%s", msg.c_str());
    return rc;
}

void cge_text_table_close(void *handle) { delete (TextTable *)handle; }

} // extern "C"
