// Can the mirror sums of a SMALL graph ride on the sweep as 64-bit fixed-point atomics on global memory (round 4)?  The sums are integers since
// this round, so adds in any order give the same bits; what is not known is what agent-scope atomics cost on this part (eight L2s: they are
// resolved behind them).  The probe issues `n_ops` no-return 64-bit adds (one or two per thread, like a cycle with one or two sampled mirrors)
// at random places of an array of `n_addr` words -- C1 (n = 200, p = 0.5): ~0.36 M adds on 20 K words -- and compares with the same launch doing
// plain 8-byte stores and doing nothing.  Prints us per launch, averaged over back-to-back launches.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/atomic_probe tools/probes/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_ops(unsigned long long* T, const uint32_t* idx, const double* w, int n_threads, int per_thread) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_threads) return;
    const double v = w[t];                                         // the thread's "new weight"
    const unsigned long long fx = (unsigned long long)__double2ll_rn(v * 0x1p50);
    for (int q = 0; q < per_thread; ++q) {
        const uint32_t a = idx[(size_t)q * n_threads + t];
        if (MODE == 0) __hip_atomic_fetch_add(&T[a], fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) T[a] = fx;
    }
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { int n_threads, per_thread, n_addr; const char* what; };
    const Case cases[] = {
        {300000, 1, 20000, "C1-like: 0.30 M threads, 1 add each, 20 K words"},
        {300000, 2, 20000, "C1-like: 0.30 M threads, 2 adds each, 20 K words"},
        {180000, 2, 20000, "C1-like: 0.18 M threads, 2 adds each, 20 K words"},
        {1500000, 2, 100000, "1.5 M threads, 2 adds each, 100 K words"},
        {300000, 2, 2000, "0.30 M threads, 2 adds each, 2 K words (heavier collisions)"},
    };
    for (const Case& c : cases) {
        std::vector<uint32_t> idx((size_t)c.n_threads * c.per_thread);
        uint64_t s = 88172645463325252ull;
        for (auto& x : idx) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % (uint64_t)c.n_addr); }
        std::vector<double> w(c.n_threads, 0.25);
        uint32_t* d_idx; double* d_w; unsigned long long* d_T;
        CK(hipMalloc(&d_idx, 4 * idx.size())); CK(hipMalloc(&d_w, 8ull * c.n_threads)); CK(hipMalloc(&d_T, 8ull * c.n_addr));
        CK(hipMemcpy(d_idx, idx.data(), 4 * idx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, w.data(), 8ull * c.n_threads, hipMemcpyHostToDevice));
        CK(hipMemset(d_T, 0, 8ull * c.n_addr));
        const int grid = (c.n_threads + 255) / 256;
        float ms[3];
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int l = 0; l < launches; ++l) {
                    if (mode == 0) hipLaunchKernelGGL(k_ops<0>, dim3(grid), dim3(256), 0, st, d_T, d_idx, d_w, c.n_threads, c.per_thread);
                    if (mode == 1) hipLaunchKernelGGL(k_ops<1>, dim3(grid), dim3(256), 0, st, d_T, d_idx, d_w, c.n_threads, c.per_thread);
                    if (mode == 2) hipLaunchKernelGGL(k_ops<2>, dim3(grid), dim3(256), 0, st, d_T, d_idx, d_w, c.n_threads, c.per_thread);
                }
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                best = t < best ? t : best;
            }
            ms[mode] = best / launches;
        }
        unsigned long long chk = 0; CK(hipMemcpy(&chk, d_T, 8, hipMemcpyDeviceToHost));
        printf("%-62s  atomics %7.2f us   stores %7.2f us   neither %7.2f us   (%.1f G adds/s)\n", c.what, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3,
               (double)c.n_threads * c.per_thread / (ms[0] * 1e-3) * 1e-9);
        CK(hipFree(d_idx)); CK(hipFree(d_w)); CK(hipFree(d_T));
    }
    return 0;
}
