// How fast can the caller's pageable rotation array (72 B per edge, C4: 180 MB) reach HBM (round 4)?  It is the longest item on the set-up path of
// desc_pgd_solve (DESC_PGD.m:14's RijMat; 13-15 ms at C4).  Strategies, wall time each (best of `reps`), on `mb` megabytes of malloc'ed memory:
//   0  one hipMemcpy from pageable memory
//   1  hipHostRegister the whole array + hipMemcpy in 16 MiB pieces + hipHostUnregister      (what copy_rij does)
//   2  the same, registration and copies pipelined piece by piece on two threads
//   3  two pinned bounce buffers (allocated once, outside the timing), T threads memcpy a piece into one while the other is DMA'd
// build: hipcc -O3 --offload-arch=gfx950 -pthread -o tools/probes/upload_probe tools/probes/upload_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

static void par_memcpy(char* dst, const char* src, size_t bytes, int T) {
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { const size_t a = bytes * t / T, b = bytes * (t + 1) / T; memcpy(dst + a, src + a, b - a); });
    for (auto& x : th) x.join();
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 180;
    const int reps = argc > 2 ? atoi(argv[2]) : 4;
    const size_t total = mb << 20, piece = (size_t)16 << 20;
    CK(hipSetDevice(0));
    char* h = (char*)malloc(total);
    for (size_t i = 0; i < total; i += 4096) h[i] = (char)i;       // touch every page
    char* d; CK(hipMalloc(&d, total));
    char* bounce[2]; CK(hipHostMalloc(&bounce[0], piece)); CK(hipHostMalloc(&bounce[1], piece));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
    CK(hipMemcpy(d, h, piece, hipMemcpyHostToDevice));             // warm the runtime
    auto report = [&](const char* what, double best, const char* extra = "") { printf("%-78s %7.2f ms  %6.1f GB/s %s\n", what, best, total / best * 1e-6, extra); };
    {   double best = 1e30;
        for (int r = 0; r < reps; ++r) { auto t0 = clk::now(); CK(hipMemcpy(d, h, total, hipMemcpyHostToDevice)); best = std::min(best, ms_since(t0)); }
        report("0  hipMemcpy, pageable, one call", best); }
    {   double best = 1e30, breg = 0, bcp = 0, bun = 0;
        for (int r = 0; r < reps; ++r) {
            auto t0 = clk::now(); CK(hipHostRegister(h, total, hipHostRegisterDefault)); const double t_reg = ms_since(t0);
            auto t1 = clk::now();
            for (size_t off = 0; off < total; off += piece) CK(hipMemcpy(d + off, h + off, std::min(piece, total - off), hipMemcpyHostToDevice));
            const double t_cp = ms_since(t1);
            auto t2 = clk::now(); CK(hipHostUnregister(h)); const double t_un = ms_since(t2);
            const double t = ms_since(t0);
            if (t < best) { best = t; breg = t_reg; bcp = t_cp; bun = t_un; }
        }
        char extra[128]; snprintf(extra, sizeof extra, "(register %.2f + copies %.2f + unregister %.2f)", breg, bcp, bun);
        report("1  register all, 16 MiB copies, unregister", best, extra); }
    {   double best = 1e30;
        const int np = (int)((total + piece - 1) / piece);
        for (int r = 0; r < reps; ++r) {
            std::atomic<int> registered{0}, copied{0};
            auto t0 = clk::now();
            std::thread reg([&] {
                (void)hipSetDevice(0);
                for (int k = 0; k < np; ++k) { const size_t off = (size_t)k * piece; CK(hipHostRegister(h + off, std::min(piece, total - off), hipHostRegisterDefault)); registered.store(k + 1); }
                for (int k = 0; k < np; ++k) { while (copied.load() <= k) std::this_thread::yield(); CK(hipHostUnregister(h + (size_t)k * piece)); }
            });
            for (int k = 0; k < np; ++k) {
                while (registered.load() <= k) std::this_thread::yield();
                const size_t off = (size_t)k * piece;
                CK(hipMemcpy(d + off, h + off, std::min(piece, total - off), hipMemcpyHostToDevice));
                copied.store(k + 1);
            }
            const double t_copy_done = ms_since(t0);
            reg.join();
            best = std::min(best, t_copy_done);
        }
        report("2  register piece k+1 while piece k is copied (time until the last copy ends)", best); }
    for (int T : {1, 2, 4, 8}) {
        double best = 1e30;
        for (int r = 0; r < reps; ++r) {
            auto t0 = clk::now();
            int k = 0;
            for (size_t off = 0; off < total; off += piece, ++k) {
                const size_t nb = std::min(piece, total - off);
                if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));                 // the DMA out of this bounce buffer is done
                par_memcpy(bounce[k & 1], h + off, nb, T);
                CK(hipMemcpyAsync(d + off, bounce[k & 1], nb, hipMemcpyHostToDevice, st));
                CK(hipEventRecord(ev[k & 1], st));
            }
            CK(hipStreamSynchronize(st));
            best = std::min(best, ms_since(t0));
        }
        char what[128]; snprintf(what, sizeof what, "3  two pinned bounce buffers of 16 MiB, %d thread(s) filling them", T);
        report(what, best);
    }
    return 0;
}
