// FETCH_SIZE calibration for the access widths of k_sweep_band (MI355X_MICROARCH.md, HBM: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels read a known number of
// bytes from 1 GiB arrays (far beyond L2 and the Infinity Cache) with 16-, 8- and 4-byte loads per lane, contiguous
// across the lanes of a wave; run under `rocprofv3 --pmc FETCH_SIZE` and compare the counter with the byte count.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/fetch_calib tools/probes/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class T> __global__ __launch_bounds__(1024) void k_read(const T* p, size_t n, double* sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) {
        const T v = p[i];
        const unsigned* u = reinterpret_cast<const unsigned*>(&v);
        acc += (double)u[0];
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
int main() {
    const size_t bytes = 1ull << 30;
    void* buf; double* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 8)); CK(hipMemset(buf, 1, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_read<double2>), dim3(1024), dim3(1024), 0, 0, (const double2*)buf, bytes / 16, sink);
        hipLaunchKernelGGL((k_read<double>), dim3(1024), dim3(1024), 0, 0, (const double*)buf, bytes / 8, sink);
        hipLaunchKernelGGL((k_read<unsigned>), dim3(1024), dim3(1024), 0, 0, (const unsigned*)buf, bytes / 4, sink);
    }
    CK(hipDeviceSynchronize());
    printf("each kernel reads %zu bytes\n", bytes);
    return 0;
}
