// desc_device_problem: the measurement graph resident in HBM, shared by every entry point of the library.
// (Nothing in the reference corresponds to this: there, RijMat and Ind are MATLAB arrays every function indexes.)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <new>
#include <vector>

#include "device_utils.h"

using namespace desc;

namespace desc {
// The caller's rotation array is pinned in place for the duration of the copy when it is large: a registered buffer is copied
// by DMA at PCIe speed, a pageable one is staged through the runtime's bounce buffers (DESC_UPLOAD_PIN=0 disables).
static hipError_t copy_rij(double* d_rij, const double* rij, int64_t m) {
    if (m <= 0) return hipSuccess;
    const char* pin_env = std::getenv("DESC_UPLOAD_PIN");
    const bool pin = m >= (1 << 18) && !(pin_env && std::atoi(pin_env) == 0);
    bool pinned = false;
    if (pin) { pinned = hipHostRegister((void*)rij, sizeof(double) * 9 * m, hipHostRegisterDefault) == hipSuccess; if (!pinned) (void)hipGetLastError(); }
    // In pieces of 16 MiB: desc_pgd_solve runs this copy on a helper thread while the structure builder uploads the edge list on the caller's;
    // behind ONE 180 MB transfer (C4) those two 10 MB copies waited up to 23 ms (profiles/r04_e2e_laps.txt), behind a piece at most ~1 ms.
    hipError_t e = hipSuccess;
    const size_t total = sizeof(double) * 9 * (size_t)m, piece = (size_t)16 << 20;
    for (size_t off = 0; off < total && e == hipSuccess; off += piece)
        e = hipMemcpy((char*)d_rij + off, (const char*)rij + off, std::min(piece, total - off), hipMemcpyHostToDevice);
    if (pinned) (void)hipHostUnregister((void*)rij);
    return e;
}

int upload_rij(const desc_problem* prob, int32_t device, double** d_rij) {
    *d_rij = nullptr;
    if (!prob || !prob->rij || prob->m <= 0) return DESC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { (void)hipGetLastError(); return DESC_ERR_HIP; }
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return DESC_ERR_HIP; }
    void* q = nullptr;
    if (dev_alloc(&q, sizeof(double) * 9 * (size_t)prob->m) != hipSuccess) { (void)hipGetLastError(); return DESC_ERR_HIP; }
    if (copy_rij((double*)q, prob->rij, prob->m) != hipSuccess) { (void)hipGetLastError(); dev_free(q); return DESC_ERR_HIP; }
    *d_rij = (double*)q;
    return DESC_OK;
}
void release_rij(double* d_rij, int32_t device) {
    if (!d_rij) return;
    (void)hipSetDevice(device);
    dev_free(d_rij);
}
}  // namespace desc

extern "C" {

int desc_problem_upload(const desc_problem* prob, int32_t device, desc_device_problem** out) {
    if (!out) return fail(DESC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    return no_throw("desc_problem_upload", [&]() -> int {
    int rc = validate_problem(prob, true);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DESC_ERR_HIP, "no HIP device visible: the library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(DESC_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    DESC_HIP(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();
    desc_device_problem* dp = new (std::nothrow) desc_device_problem();
    if (!dp) return fail(DESC_ERR_INVALID, "out of host memory");
    dp->device = device; dp->n = prob->n; dp->m = prob->m;
    const int64_t n = dp->n, m = dp->m;
    auto bail = [&](int code) { desc_problem_free(dp); return code; };
    auto A = [&](void** q, size_t bytes) { return dev_alloc(q, bytes) == hipSuccess; };
    if (!A((void**)&dp->d_ii, sizeof(int32_t) * m) || !A((void**)&dp->d_jj, sizeof(int32_t) * m) || !A((void**)&dp->d_rowptr, sizeof(int32_t) * (n + 1)) ||
        !A((void**)&dp->d_adj, sizeof(int32_t) * 2 * m) || !A((void**)&dp->d_adj_eid, sizeof(int32_t) * 2 * m) || !A((void**)&dp->d_rij, sizeof(double) * 9 * m))
        return bail(fail(DESC_ERR_HIP, "out of device memory for the problem (m = %lld)", (long long)m));
    // the big copy (72 B per edge; synchronous copies from pageable memory run at ~12 GB/s, asynchronous ones at ~3 GB/s
    // on this runtime) overlaps the host-side CSR pass, which runs on a helper thread (share 1 of run_threads: an exception in it
    // is rethrown here after the join and becomes an error code)
    hvec<int32_t> adj, adj_eid;
    hipError_t e = hipSuccess;
    try {
    run_threads(2, [&](int share) {
    if (share == 1) {
        dp->ii.assign(prob->ind_i, prob->ind_i + m); dp->jj.assign(prob->ind_j, prob->ind_j + m);
        build_csr(n, m, prob->ind_i, prob->ind_j, dp->rowptr, adj, adj_eid);
        return;
    }
    e = copy_rij(dp->d_rij, prob->rij, m);
    if (m && e == hipSuccess) e = hipMemcpy(dp->d_ii, prob->ind_i, sizeof(int32_t) * m, hipMemcpyHostToDevice);
    if (m && e == hipSuccess) e = hipMemcpy(dp->d_jj, prob->ind_j, sizeof(int32_t) * m, hipMemcpyHostToDevice);
    });
    } catch (...) { desc_problem_free(dp); throw; }
    if (e == hipSuccess) e = hipMemcpy(dp->d_rowptr, dp->rowptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice);
    if (m && e == hipSuccess) e = hipMemcpy(dp->d_adj, adj.data(), sizeof(int32_t) * 2 * m, hipMemcpyHostToDevice);
    if (m && e == hipSuccess) e = hipMemcpy(dp->d_adj_eid, adj_eid.data(), sizeof(int32_t) * 2 * m, hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(fail(DESC_ERR_HIP, "problem upload: %s", hipGetErrorString(e)));
    dp->ms_upload = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = dp;
    return DESC_OK;
    });
}

void desc_problem_free(desc_device_problem* dp) {
    if (!dp) return;
    (void)hipSetDevice(dp->device);
    for (void* q : {(void*)dp->d_ii, (void*)dp->d_jj, (void*)dp->d_rowptr, (void*)dp->d_adj, (void*)dp->d_adj_eid, (void*)dp->d_rij})
        dev_free(q);
    delete dp;
}

}  // extern "C"
