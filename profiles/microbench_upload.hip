// microbench_upload.hip -- what the host-to-device path of cge_set_embedding can reach on this box (round 5):
// a 1 GiB pageable buffer to the device (a) by one hipMemcpy, (b) pinned in place with hipHostRegister and copied directly,
// (c) through pinned staging buffers filled by T host threads (the library's staged_upload) for several chunk sizes / thread
// counts / ring depths.  Prints ms and GB/s.   hipcc -O2 -o /tmp/mbu profiles/microbench_upload.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = (size_t)1 << 30;
    char *h = (char *)aligned_alloc(4096, bytes);
    memset(h, 1, bytes);
    char *d;
    CK(hipMalloc(&d, bytes));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        printf("(a) hipMemcpy from pageable: %.2f ms  %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipHostUnregister(h));
        double t3 = now();
        printf("(b) register %.2f ms + copy %.2f ms (%.1f GB/s) + unregister %.2f ms = %.2f ms  %.1f GB/s\n", t1 - t0, t2 - t1,
               bytes / (t2 - t1) / 1e6, t3 - t2, t3 - t0, bytes / (t3 - t0) / 1e6);
    }
    // (b2) register in pieces, copy piece k while piece k+1 is being pinned
    for (size_t piece : {(size_t)64 << 20, (size_t)256 << 20}) {
        double t0 = now();
        for (size_t off = 0; off < bytes; off += piece) {
            CK(hipHostRegister(h + off, piece, hipHostRegisterDefault));
            CK(hipMemcpyAsync(d + off, h + off, piece, hipMemcpyHostToDevice, st));
        }
        CK(hipStreamSynchronize(st));
        double t1 = now();
        for (size_t off = 0; off < bytes; off += piece) CK(hipHostUnregister(h + off));
        double t2 = now();
        printf("(b2) pieces of %zu MiB: register+copy pipelined %.2f ms (%.1f GB/s), unregister %.2f ms\n", piece >> 20, t1 - t0,
               bytes / (t1 - t0) / 1e6, t2 - t1);
    }
    for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20})
        for (int nt : {4, 8, 16})
            for (int depth : {2, 4}) {
                std::vector<char *> stage(depth);
                std::vector<hipEvent_t> ev(depth);
                for (int b = 0; b < depth; b++) { CK(hipHostMalloc(&stage[b], chunk)); memset(stage[b], 0, chunk); CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)); }
                double best = 1e30;
                for (int rep = 0; rep < 2; rep++) {
                    double t0 = now();
                    size_t k = 0;
                    for (size_t off = 0; off < bytes; off += chunk, k++) {
                        const int b = (int)(k % depth);
                        if (k >= (size_t)depth) CK(hipEventSynchronize(ev[b]));
                        std::vector<std::thread> th;
                        const size_t per = chunk / nt;
                        for (int t = 1; t < nt; t++) th.emplace_back([&, t] { memcpy(stage[b] + t * per, h + off + t * per, per); });
                        memcpy(stage[b], h + off, per);
                        for (auto &x : th) x.join();
                        CK(hipMemcpyAsync(d + off, stage[b], chunk, hipMemcpyHostToDevice, st));
                        CK(hipEventRecord(ev[b], st));
                    }
                    CK(hipStreamSynchronize(st));
                    best = std::min(best, now() - t0);
                }
                printf("(c) staged chunk %2zu MiB, %2d threads, ring of %d: %.2f ms  %.1f GB/s\n", chunk >> 20, nt, depth, best, bytes / best / 1e6);
                for (int b = 0; b < depth; b++) { CK(hipHostFree(stage[b])); CK(hipEventDestroy(ev[b])); }
            }
    // host memcpy alone (no GPU): what the fill threads can move
    {
        char *p;
        CK(hipHostMalloc(&p, (size_t)256 << 20));
        for (int nt : {1, 4, 8, 16}) {
            double t0 = now();
            for (int r = 0; r < 4; r++) {
                std::vector<std::thread> th;
                const size_t per = ((size_t)256 << 20) / nt;
                for (int t = 0; t < nt; t++) th.emplace_back([&, t] { memcpy(p + t * per, h + (size_t)r * ((size_t)256 << 20) + t * per, per); });
                for (auto &x : th) x.join();
            }
            double t1 = now();
            printf("(d) host memcpy pageable -> pinned, %2d threads: %.1f GB/s\n", nt, bytes / (t1 - t0) / 1e6);
        }
    }
    return 0;
}
