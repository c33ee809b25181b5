// What a grid-wide barrier costs on this part (round 4; the question behind a multi-iteration launch for the reference's demo sizes,
// Demo/compare_algorithms.m:10): one DESC_PGD iteration is two phases with a global dependency between them (the mirror sums of DESC_PGD.m:185-191
// need every weight of the previous sweep, the sweep of :193-229 needs every sum), so a persistent kernel pays TWO barriers per iteration.
// The probe launches `nwg` workgroups of 256 threads (all co-resident: nwg <= 2048), has them run `rounds` barriers on a monotonic counter
// (release add + acquire spin at agent scope, one lane per workgroup, __syncthreads around it) and prints the time per barrier:
//   mode 0: barrier only
//   mode 1: + one dependent global load per thread between barriers (the shortest possible "phase")
//   mode 2: only the workgroups with blockIdx % 8 == 0 take part (one XCD: workgroups are dealt round-robin to the 8 XCDs)
// Every spin is bounded (SPIN_CAP polls, then the workgroup sets an error word and leaves): the grid always drains.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/grid_barrier_probe tools/probes/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int SPIN_CAP = 4 << 20;

__global__ __launch_bounds__(256) void k_barriers(unsigned* counter, unsigned* err, const int* chase, int n_chase, int rounds, int mode, int members, int* sink) {
    if (mode == 2 && (blockIdx.x & 7) != 0) return;
    int pos = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)n_chase);
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    for (int r = 1; r <= rounds; ++r) {
        if (mode == 1) pos = chase[pos];
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)members * (unsigned)r;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > SPIN_CAP) { s_bad = 1; atomicExch(err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (s_bad || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;     // one workgroup gave up: everybody leaves
    }
    if (pos == -1) sink[0] = pos;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipSetDevice(0));
    unsigned *d_counter, *d_err; int *d_chase, *d_sink;
    const int n_chase = 1 << 20;
    std::vector<int> chase(n_chase);
    for (int i = 0; i < n_chase; ++i) chase[i] = (int)(((uint64_t)i * 2654435761ull + 12345ull) % (uint64_t)n_chase);
    CK(hipMalloc(&d_counter, 4)); CK(hipMalloc(&d_err, 4)); CK(hipMalloc(&d_chase, 4ull * n_chase)); CK(hipMalloc(&d_sink, 4));
    CK(hipMemcpy(d_chase, chase.data(), 4ull * n_chase, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int grids[] = {32, 64, 128, 256, 512, 1024, 2048};
    for (int mode = 0; mode < 3; ++mode)
        for (int nwg : grids) {
            if (mode == 2 && nwg < 256) continue;
            const int members = mode == 2 ? nwg / 8 : nwg;
            float best = 1e30f, best1 = 1e30f; unsigned err = 0;
            for (int rep = 0; rep < 5 && !err; ++rep)
                for (int rr : {1, rounds}) {                        // a launch with ONE barrier gives the launch's own time
                    CK(hipMemsetAsync(d_counter, 0, 4, st)); CK(hipMemsetAsync(d_err, 0, 4, st));
                    CK(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(k_barriers, dim3(nwg), dim3(256), 0, st, d_counter, d_err, d_chase, n_chase, rr, mode, members, d_sink);
                    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
                    if (err) break;
                    if (rr == 1) best1 = ms < best1 ? ms : best1; else best = ms < best ? ms : best;
                }
            if (err) { printf("mode %d  workgroups %4d  GAVE UP (a spin reached its cap)\n", mode, nwg); continue; }
            printf("mode %d  workgroups %4d (members %4d)  launch with 1 barrier %7.2f us   %d barriers %8.2f us   per barrier %6.2f us\n",
                   mode, nwg, members, best1 * 1e3, rounds, best * 1e3, (best - best1) * 1e3 / (rounds - 1));
        }
    return 0;
}
