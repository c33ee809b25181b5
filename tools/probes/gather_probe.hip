// How the CU's address unit prices an 8-byte gather instruction as a function of WHICH lanes share a 128-byte line (round 3).
// Every workgroup gathers from its own 16 KiB window (L1-resident after the first touch), 8 independent loads in flight per wave, 8 waves per CU;
// the patterns differ only in the lane -> line map.  Prints ns per wave instruction and CU.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/gather_probe tools/probes/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int LINES = 128;                  // 16 KiB window = 128 lines of 16 doubles
__device__ __forceinline__ int line_of(int pat, int lane) {
    switch (pat) {
        case 0: return 0;                   // one line for the whole wave
        case 1: return lane >> 2;           // the 4 lanes of a quad share a line (16 lines)
        case 2: return lane & 15;           // lanes 16 apart share a line (16 lines)
        case 3: return lane;                // 64 different lines
        case 4: return lane >> 1;           // adjacent pairs share (32 lines)
        case 5: return lane & 31;           // lanes 32 apart share (32 lines)
        case 6: return lane >> 4;           // the 16 lanes of a row share (4 lines)
        default: return lane & 3;           // lanes 4 apart share (4 lines)
    }
}
template <bool BUF>
__global__ __launch_bounds__(512) void k_gather(const double* tab, int pat, int iters, double* sink) {
    const int lane = threadIdx.x & 63;
    const double* win = tab + (size_t)blockIdx.x * LINES * 16;
    const int ln = line_of(pat, lane), word = (lane * 5) & 15;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, LINES * 128, 0x00020000);
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int l2 = (ln + 7 * u + 13 * it) & (LINES - 1);
            const int idx = l2 * 16 + word;
            if (BUF) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, idx * 8, 0, 0);
                __builtin_memcpy(&v[u], &r, 8);
            } else v[u] = win[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
// contiguous loads of W bytes per lane (the sweep's streams): lane L reads bytes [W * L, W * (L + 1)) of a run that moves through the window
template <int W>
__global__ __launch_bounds__(512) void k_stream(const double* tab, int iters, double* sink) {
    const int lane = threadIdx.x & 63;
    const char* win = (const char*)(tab + (size_t)blockIdx.x * LINES * 16);
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int off = ((u * 17 + it * 29) * 128 + W * lane) & (LINES * 128 - 1) & ~(W - 1);
            if (W == 4) v[u] = (double)*(const unsigned*)(win + off);
            else if (W == 8) v[u] = *(const double*)(win + off);
            else { const double2 t = *(const double2*)(win + off); v[u] = t.x + t.y; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
template <int W>
__global__ __launch_bounds__(512) void k_store(double* tab, int iters) {
    const int lane = threadIdx.x & 63;
    char* win = (char*)(tab + (size_t)blockIdx.x * LINES * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int off = ((u * 17 + it * 29 + (threadIdx.x >> 6) * 5) * 128 + W * lane) & (LINES * 128 - 1) & ~(W - 1);
            if (W == 8) *(double*)(win + off) = (double)it;
            else *(double2*)(win + off) = double2{(double)it, 1.0};
        }
    }
}
int main() {
    const int G = 256, iters = 2000;
    double *tab, *sink;
    CK(hipMalloc(&tab, (size_t)G * LINES * 128)); CK(hipMalloc(&sink, 8)); CK(hipMemset(tab, 0, (size_t)G * LINES * 128));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[8] = {"one line", "quads share (16 lines)", "lanes 16 apart share (16 lines)", "64 lines", "pairs share (32 lines)", "lanes 32 apart share (32 lines)",
                            "rows of 16 share (4 lines)", "lanes 4 apart share (4 lines)"};
    for (int buf = 0; buf < 2; ++buf)
        for (int pat = 0; pat < 8; ++pat) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (buf) hipLaunchKernelGGL(k_gather<true>, dim3(G), dim3(512), 0, 0, tab, pat, iters, sink);
                else hipLaunchKernelGGL(k_gather<false>, dim3(G), dim3(512), 0, 0, tab, pat, iters, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_cu = 8.0 * iters * 8;            // 8 waves x iters x 8 loads
            printf("%-6s %-34s %8.3f ms  %7.2f ns per wave instruction and CU (%5.1f cycles at 2.4 GHz)\n", buf ? "buffer" : "global", names[pat], ms,
                   ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
        }
    for (int w = 0; w < 5; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (w == 0) hipLaunchKernelGGL(k_stream<4>, dim3(G), dim3(512), 0, 0, tab, iters, sink);
            else if (w == 1) hipLaunchKernelGGL(k_stream<8>, dim3(G), dim3(512), 0, 0, tab, iters, sink);
            else if (w == 2) hipLaunchKernelGGL(k_stream<16>, dim3(G), dim3(512), 0, 0, tab, iters, sink);
            else if (w == 3) hipLaunchKernelGGL(k_store<8>, dim3(G), dim3(512), 0, 0, tab, iters);
            else hipLaunchKernelGGL(k_store<16>, dim3(G), dim3(512), 0, 0, tab, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_cu = 8.0 * iters * 8;
        const char* nm[5] = {"contiguous load, 4 B per lane", "contiguous load, 8 B per lane", "contiguous load, 16 B per lane", "contiguous store, 8 B per lane", "contiguous store, 16 B per lane"};
        printf("%-40s %8.3f ms  %7.2f ns per wave instruction and CU (%5.1f cycles at 2.4 GHz)\n", nm[w], ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    }
    return 0;
}
