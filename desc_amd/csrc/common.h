// Internal declarations shared by the host-side translation units of libdesc_amd.so.
#pragma once
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "desc_amd.h"
#include "hostmem.h"
using desc::hvec;      // every host vector of the library: pooled large blocks (hostmem.h)

namespace desc {

// thread-local error text behind desc_last_error()
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

// splitmix64 finaliser; the sampling key is defined on top of it (see desc_sample_key)
inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
inline uint64_t sample_key(uint64_t seed, uint64_t edge, uint64_t k) {
    uint64_t a = mix64(seed ^ ((edge + 1) * 0x9E3779B97F4A7C15ull));
    return mix64(a ^ ((k + 1) * 0xD1B54A32D192ED03ull));
}

}  // namespace desc

// The sampled 3-cycle structure (DESC_PGD.m:29-127) in host memory.
struct desc_structure {
    int64_t n = 0, m = 0, m_pos = 0, m_cycle = 0;
    int32_t n_sample = 0, max_cnt = 0;
    hvec<int32_t> codeg, pos_edge;
    hvec<int64_t> cum_ind;
    hvec<int32_t> rowptr_host;                // CSR row starts of the graph, when the builder made them (device builder)
    hvec<int32_t> k, e_jk, e_ki, ikj, jki;   // per-cycle arrays on the host (valid iff host_cycles)
    double ms_build = 0.0;
    // A structure built on the device stays there in a lean form: the sampled third vertices `k`
    // (natural order), per edge-with-cycles the selection threshold (largest selected key and its k;
    // all ones when the edge was not sampled) -- enough to decide "was cycle (ik;j) sampled?" without
    // the mirror maps -- plus the CSR adjacency the solver's layout kernels re-use.  e_jk, e_ki, ikj,
    // jki are derived (device kernels) and copied to the host only when somebody asks
    // (desc_structure_get, the gather layout).
    bool host_cycles = true;
    int dev = -1;
    void* ev_fill = nullptr;                  // hipEvent_t recorded behind the kernel that fills d_k / d_tau / d_ktau: the device builder returns while it
                                              // runs (the host plans the solver's layout meanwhile); consumers on other streams wait for it
    void* fill_stream = nullptr;              // hipStream_t (pooled, non-blocking) the compaction and cycle-sampling kernels of the device builder run on
    hvec<void*> d_build_blocks;               // scratch of those kernels: lives until structure_free_device (freeing a block waits for the device, i.e. for the fill)
    int32_t *d_k = nullptr;
    int32_t* d_codeg = nullptr;               // m, one of d_build_blocks: the host's copy (desc_structure_get) is made on demand, structure_ensure_host
    unsigned long long* d_tau = nullptr;      // m, indexed by edge id (defined for edges with cycles)
    int32_t* d_ktau = nullptr;                // m
    int32_t *d_rowptr = nullptr, *d_adj = nullptr, *d_adj_eid = nullptr;   // n+1, 2m, 2m
    int32_t *d_ii = nullptr, *d_jj = nullptr;                              // m
    int32_t *d_pos = nullptr, *d_cum = nullptr, *d_poe = nullptr;          // m_pos, m_pos+1, m (edge -> index in pos_edge, -1)
    unsigned long long* d_bits = nullptr;     // n x words adjacency bitmaps
    uint32_t* d_rank = nullptr;               // n x words: neighbours of v in words < w (position of k in row v = rank + popcount below)
    int32_t words = 0;
    int32_t max_deg = 0;
    uint64_t seed = 0;
};

// A problem resident in HBM (desc_problem_upload): edge list, rotations and the CSR index of the undirected graph --
// what DESC()'s three stages (PGD, GCW, refinement), Spectral and CEMP each used to rebuild and re-upload.
struct desc_device_problem {
    int device = 0;
    int64_t n = 0, m = 0;
    hvec<int32_t> ii, jj, rowptr;      // host copies of the index data (O(m) host passes: degrees, validation)
    int32_t *d_ii = nullptr, *d_jj = nullptr, *d_rowptr = nullptr, *d_adj = nullptr, *d_adj_eid = nullptr;
    double* d_rij = nullptr;
    double ms_upload = 0.0;
};

namespace desc {
// a-1..a-3 on the host (structure_host.cpp)
int build_structure_host(const desc_problem* prob, int32_t n_sample_min, uint64_t seed,
                         desc_structure* out);
// a-1..a-3 on the device (structure_device.hip)
int build_structure_device(const desc_problem* prob, int32_t n_sample_min, uint64_t seed,
                           int32_t device, desc_structure* out);
int validate_problem(const desc_problem* prob, bool need_rij);
// CSR adjacency of the undirected graph (neighbours ascending, edge id per slot); Ind must be sorted by (i,j).
// Multithreaded for large m; the result does not depend on the thread count.
void build_csr(int64_t n, int64_t m, const int32_t* ii, const int32_t* jj, hvec<int32_t>& rowptr, hvec<int32_t>& adj,
               hvec<int32_t>& adj_eid);
// device-resident structures (structure_device.hip)
int structure_ensure_host(desc_structure* s);      // copy the per-cycle arrays to the host if they live on the device
void structure_free_device(desc_structure* s);
// desc_pgd_solve: the rotations (72 B per edge, the one large host -> device copy of a call) go up on a helper thread while the structure is
// built from the edge list; the solver handle is then created on that device copy.  upload_rij: *d_rij = a device block holding prob->rij
// (NULL and an error code if there is no device, no memory or the copy failed: the caller carries on without it); release_rij frees it.
int upload_rij(const desc_problem* prob, int32_t device, double** d_rij);
// incremented by the device structure builder once the edge list is on the device: desc_pgd_solve's helper thread starts pinning and copying
// the rotations only then (hipHostRegister of 180 MB holds the runtime for ~10 ms: the builder's two small copies waited behind it)
extern std::atomic<uint64_t> g_ind_upload_count;
void release_rij(double* d_rij, int32_t device);
int pgd_create_with_rij(const desc_problem* prob, const double* d_rij, const desc_structure* s, int32_t device, desc_pgd** out);
// CEMP.m:44-65 on the device: nsample cycles per edge-with-cycles, with replacement.  The four arrays are
// hipMalloc'ed on `device` (caller frees); DESC_ERR_TOO_LARGE when a codegree exceeds the LDS staging budget.
int build_cemp_samples_device(const desc_device_problem* dp, int32_t nsample, uint64_t seed, int64_t* m_pos,
                              int32_t** o_pos, int32_t** o_k, int32_t** o_ejk, int32_t** o_eki, uint32_t** o_pk = nullptr, int32_t* o_max_deg = nullptr);
// the desc_problem view of a device problem's host index copies (rij = NULL)
inline desc_problem host_view(const desc_device_problem* dp) { return desc_problem{dp->n, dp->m, dp->ii.data(), dp->jj.data(), nullptr}; }
int build_cemp_samples_host(const desc_problem* prob, int32_t nsample, uint64_t seed, hvec<int32_t>& pos_edge,
                            hvec<int32_t>& kk, hvec<int32_t>& e_jk, hvec<int32_t>& e_ki);
// Worker threads of the host-side passes.  An exception that escapes a thread body (bad_alloc in a vector, system_error from the
// thread constructor) would call std::terminate and take the MATLAB / Python host down: every body runs behind a catch, the
// calling thread takes share 0 itself, all workers are joined, and the first exception is rethrown on the calling thread -- where
// the no_throw guard of the entry point turns it into an error code.
template <class F>
void run_threads(int T, F&& body) {              // body(t), t = 0 .. T-1
    if (T <= 1) { body(0); return; }
    std::exception_ptr err;
    std::mutex mu;
    auto guarded = [&](int t) {
        try { body(t); }
        catch (...) { std::lock_guard<std::mutex> lk(mu); if (!err) err = std::current_exception(); }
    };
    std::vector<std::thread> th;
    try {
        th.reserve((size_t)T - 1);
        for (int t = 1; t < T; ++t) th.emplace_back(guarded, t);
        guarded(0);
    } catch (...) { std::lock_guard<std::mutex> lk(mu); if (!err) err = std::current_exception(); }
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
}
// No C++ exception may cross the C ABI (a MATLAB or Python host would be aborted): the entry points that allocate host memory run
// their bodies through this guard.
template <class F>
int no_throw(const char* what, F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(DESC_ERR_INVALID, "%s: out of host memory", what); }
    catch (const std::exception& e) { return fail(DESC_ERR_INVALID, "%s: %s", what, e.what()); }
    catch (...) { return fail(DESC_ERR_INVALID, "%s: unknown C++ exception", what); }
}
}  // namespace desc
