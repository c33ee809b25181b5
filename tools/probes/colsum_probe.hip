// What bounds k_colsum_node (round 4; DESC_PGD.m:185-191 in node form): the kernel reads, per (node, incident segment), ONE short run of
// consecutive weights -- the cycles of that segment whose mirror was sampled, ~12.5 doubles = 100 bytes at C4 -- at a scattered place of the
// 1 GB weight array, plus a sequential 2-byte column index per entry.  This probe reproduces only that access pattern in the kernel's launch
// shape (one 256-thread workgroup per node, 16 lanes per run, 4 runs per wave instruction, 8 instructions in flight per wave, 2 halves) with
// NO LDS accumulation and no records to decode: `runs_per_node` runs of `len` doubles per node, starting at
//   mode 0: random 8-byte-aligned offsets of the array          (the kernel's pattern for the smaller endpoint)
//   mode 1: random offsets, but 4 consecutive runs adjacent      (the larger endpoint: the i of a band sharing j are neighbours)
//   mode 2: sequential runs (a pure stream in the same launch shape)
//   mode 3: random offsets aligned to 128-byte lines
// Prints us per launch and GB/s of useful bytes.   build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/colsum_probe tools/probes/colsum_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int U = 8;
__global__ __launch_bounds__(256) void k_probe(const double* w, const uint32_t* start, const uint16_t* midx, int runs_per_node, int len, double* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane >> 4, l16 = lane & 15;
    const uint32_t* st = start + (size_t)blockIdx.x * runs_per_node;
    const uint16_t* mi = midx + (size_t)blockIdx.x * runs_per_node * len;
    double acc = 0.0;
    for (int g0 = wv; 4 * g0 < runs_per_node; g0 += 4 * U) {
        double v[2 * U]; uint32_t c[2 * U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tt = 4 * (g0 + 4 * u) + sub;
            v[2 * u] = 0.0; v[2 * u + 1] = 0.0; c[2 * u] = 0; c[2 * u + 1] = 0;
            if (tt < runs_per_node) {
                const uint32_t s0 = st[tt];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = l16 + 16 * h;
                    if (q < len) { c[2 * u + h] = mi[(size_t)tt * len + q]; v[2 * u + h] = w[(size_t)s0 + q]; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * U; ++u) acc += v[u] * (double)(c[u] | 1u);
    }
    if (acc == 1.2345e300) out[blockIdx.x] = acc;
}
int main(int argc, char** argv) {
    const int nodes = argc > 1 ? atoi(argv[1]) : 5000, rpn = argc > 2 ? atoi(argv[2]) : 1000, len = argc > 3 ? atoi(argv[3]) : 13;
    const size_t words = (size_t)125000000;                // 1 GB of weights
    double* w; uint32_t* start; uint16_t* midx; double* out;
    CK(hipMalloc(&w, words * 8 + 4096)); CK(hipMemset(w, 0, words * 8 + 4096));
    CK(hipMalloc(&start, (size_t)nodes * rpn * 4)); CK(hipMalloc(&midx, (size_t)nodes * rpn * len * 2 + 64)); CK(hipMemset(midx, 0, (size_t)nodes * rpn * len * 2 + 64));
    CK(hipMalloc(&out, nodes * 8));
    std::vector<uint32_t> hs((size_t)nodes * rpn);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"random 8-byte-aligned runs", "random, 4 adjacent runs per cluster", "sequential runs (stream)", "random 128-byte-aligned runs"};
    for (int mode = 0; mode < 4; ++mode) {
        uint64_t x = 88172645463325252ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        for (size_t r = 0; r < hs.size(); ++r) {
            if (mode == 0) hs[r] = (uint32_t)(rnd() % (words - 64));
            else if (mode == 1) hs[r] = (r & 3) ? hs[r - 1] + 50 : (uint32_t)(rnd() % (words - 256));
            else if (mode == 2) hs[r] = (uint32_t)((r * (size_t)(len + 12)) % (words - 64));
            else hs[r] = (uint32_t)((rnd() % (words / 16 - 8)) * 16);
        }
        CK(hipMemcpy(start, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_probe, dim3(nodes), dim3(256), 0, 0, w, start, midx, rpn, len, out);
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k_probe, dim3(nodes), dim3(256), 0, 0, w, start, midx, rpn, len, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, bytes = (double)nodes * rpn * len * 10.0 + (double)nodes * rpn * 4.0;
        printf("%-40s %d nodes x %d runs x %d doubles: %8.1f us per launch, %6.2f TB/s of useful bytes (weights + column indices + run starts)\n", names[mode], nodes, rpn, len, us, bytes / us / 1e6);
    }
    return 0;
}
