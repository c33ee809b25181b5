// Host side of the C ABI that needs no device code: error text, parameter
// defaults, structure objects, and the one-shot desc_pgd_solve.
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>

#include "common.h"

namespace desc {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
}
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace desc

using namespace desc;

extern "C" {

const char* desc_last_error(void) { return g_err.c_str(); }
const char* desc_version(void) { return "desc_amd 0.1 (gfx950)"; }

uint64_t desc_sample_key(uint64_t seed, uint64_t edge, uint64_t k) { return sample_key(seed, edge, k); }

void desc_params_default(desc_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->iters = 100;                      // Demo/compare_algorithms.m:39
    p->step_kind = DESC_STEP_CONSTANT;   // Demo/compare_algorithms.m:43
    p->lr = 0.01;
    p->beta1 = 0.9; p->beta2 = 0.999; p->decay_interval = 25;
    p->patience = 30;                    // DESC_PGD.m:180
    p->stop_tol = 1e-5;                  // DESC_PGD.m:243
    p->n_sample_min = 30;                // DESC_PGD.m:43
    p->build_where = DESC_BUILD_DEVICE;
}

static int structure_build_checked(const desc_problem* prob, int32_t n_sample_min, uint64_t seed, int32_t where, int32_t device, desc_structure** out,
                                   bool validated);
int desc_structure_build(const desc_problem* prob, int32_t n_sample_min, uint64_t seed,
                         int32_t where, int32_t device, desc_structure** out) {
    return structure_build_checked(prob, n_sample_min, seed, where, device, out, false);
}
// validated: the caller (desc_pgd_solve) has just run validate_problem on this edge list
static int structure_build_checked(const desc_problem* prob, int32_t n_sample_min, uint64_t seed, int32_t where, int32_t device, desc_structure** out,
                                   bool validated) {
    if (!out) return fail(DESC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validated ? DESC_OK : validate_problem(prob, false);
    if (rc) return rc;
    if (n_sample_min < 1) return fail(DESC_ERR_INVALID, "n_sample_min must be >= 1");
    desc_structure* s = new (std::nothrow) desc_structure();
    if (!s) return fail(DESC_ERR_INVALID, "out of host memory");
    rc = no_throw("desc_structure_build", [&]() -> int {
        return (where == DESC_BUILD_DEVICE) ? build_structure_device(prob, n_sample_min, seed, device, s)
                                            : build_structure_host(prob, n_sample_min, seed, s);
    });
    if (rc) { structure_free_device(s); delete s; return rc; }
    *out = s;
    return DESC_OK;
}

int desc_structure_import(int64_t n, int64_t m, int64_t m_pos, int32_t n_sample,
                          const int32_t* pos_edge, const int64_t* cum_ind,
                          const int32_t* k, const int32_t* e_jk, const int32_t* e_ki,
                          const int32_t* ikj, const int32_t* jki, desc_structure** out) {
    return no_throw("desc_structure_import", [&]() -> int {
    if (!out) return fail(DESC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 0 || m < 0 || m_pos < 0 || m_pos > m) return fail(DESC_ERR_INVALID, "bad sizes");
    if (!cum_ind || (m_pos > 0 && !pos_edge)) return fail(DESC_ERR_INVALID, "NULL array");
    if (cum_ind[0] != 0) return fail(DESC_ERR_INVALID, "cum_ind[0] != 0");
    int64_t mc = cum_ind[m_pos];
    if (mc >= (1ll << 31) - 1) return fail(DESC_ERR_TOO_LARGE, "m_cycle too large");
    if (mc > 0 && (!k || !e_jk || !e_ki || !ikj || !jki)) return fail(DESC_ERR_INVALID, "NULL cycle array");
    int32_t max_cnt = 0;
    for (int64_t l = 0; l < m_pos; ++l) {
        int64_t c = cum_ind[l + 1] - cum_ind[l];
        if (c < 1) return fail(DESC_ERR_INVALID, "segment %lld is empty", (long long)l);
        if (pos_edge[l] < 0 || pos_edge[l] >= m || (l > 0 && pos_edge[l] <= pos_edge[l - 1]))
            return fail(DESC_ERR_INVALID, "pos_edge must be ascending edge ids");
        if (c > max_cnt) max_cnt = (int32_t)c;
    }
    for (int64_t c = 0; c < mc; ++c) {
        if (e_jk[c] < 0 || e_jk[c] >= m || e_ki[c] < 0 || e_ki[c] >= m) return fail(DESC_ERR_INVALID, "edge id out of range at cycle %lld", (long long)c);
        if (ikj[c] < -1 || ikj[c] >= mc || jki[c] < -1 || jki[c] >= mc) return fail(DESC_ERR_INVALID, "mirror index out of range at cycle %lld", (long long)c);
        if (k[c] < 0 || k[c] >= n) return fail(DESC_ERR_INVALID, "k out of range at cycle %lld", (long long)c);
    }
    desc_structure* s = new (std::nothrow) desc_structure();
    if (!s) return fail(DESC_ERR_INVALID, "out of host memory");
    s->n = n; s->m = m; s->m_pos = m_pos; s->m_cycle = mc; s->n_sample = n_sample; s->max_cnt = max_cnt;
    s->codeg.assign(m, 0);
    s->pos_edge.assign(pos_edge, pos_edge + m_pos);
    s->cum_ind.assign(cum_ind, cum_ind + m_pos + 1);
    for (int64_t l = 0; l < m_pos; ++l) s->codeg[pos_edge[l]] = (int32_t)(cum_ind[l + 1] - cum_ind[l]);  // sampled count; true codegree unknown
    s->k.assign(k, k + mc); s->e_jk.assign(e_jk, e_jk + mc); s->e_ki.assign(e_ki, e_ki + mc);
    s->ikj.assign(ikj, ikj + mc); s->jki.assign(jki, jki + mc);
    *out = s;
    return DESC_OK;
    });
}

int desc_structure_get(const desc_structure* cs, desc_structure_view* v) {
    if (!cs || !v) return fail(DESC_ERR_INVALID, "NULL argument");
    desc_structure* s = const_cast<desc_structure*>(cs);      // lazily materialises the host copy of a device-built structure
    int rc = no_throw("desc_structure_get", [&]() -> int { return structure_ensure_host(s); });
    if (rc) return rc;
    v->n = s->n; v->m = s->m; v->m_pos = s->m_pos; v->m_cycle = s->m_cycle;
    v->n_sample = s->n_sample; v->max_cnt = s->max_cnt;
    v->codeg = s->codeg.data(); v->pos_edge = s->pos_edge.data(); v->cum_ind = s->cum_ind.data();
    v->k = s->k.data(); v->e_jk = s->e_jk.data(); v->e_ki = s->e_ki.data();
    v->ikj = s->ikj.data(); v->jki = s->jki.data();
    return DESC_OK;
}

int desc_structure_sizes(const desc_structure* s, desc_structure_info* info) {
    if (!s || !info) return fail(DESC_ERR_INVALID, "NULL argument");
    info->n = s->n; info->m = s->m; info->m_pos = s->m_pos; info->m_cycle = s->m_cycle;
    info->n_sample = s->n_sample; info->max_cnt = s->max_cnt;
    info->built_where = s->d_k ? DESC_BUILD_DEVICE : DESC_BUILD_HOST;
    info->host_resident = s->host_cycles ? 1 : 0;
    info->ms_build = s->ms_build;
    return DESC_OK;
}

void desc_structure_free(desc_structure* s) { if (s) { structure_free_device(s); delete s; } }
// (Releasing the host side -- tens of megabytes of index vectors, 10+ ms of page-table work at C4 -- on a background thread was
//  tried: it contends with the launching thread for the process's memory map and the PGD loop that follows lost what it saved.
//  Doing it on the calling thread while the device runs the first iterations was tried too: the unmapping stalls the kernel that is
//  running -- one sweep of 16-26 ms instead of 1.2 ms in the rocprofv3 trace at C4 -- for as long as it takes.)

int desc_pgd_solve(const desc_problem* prob, const desc_params* p, desc_result* r) {
    return no_throw("desc_pgd_solve", [&]() -> int {
    if (!p || !r) return fail(DESC_ERR_INVALID, "NULL argument");
    auto t0 = std::chrono::steady_clock::now();
    auto t_lap = t0;
    const char* tenv = std::getenv("DESC_DEBUG_TIMING");
    const bool timing = tenv && std::atoi(tenv) != 0;
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[desc_amd] solve %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    desc_structure* s = nullptr;
    int rc = validate_problem(prob, true);
    if (rc) return rc;
    lap("validate");
    // the structure needs only the edge list: the rotations (72 B per edge) go up on a helper thread meanwhile (C4: 13 ms hidden)
    double* d_rij = nullptr;
    const char* ov = std::getenv("DESC_DEBUG_OVERLAP_UPLOAD");
    const bool overlap_upload = ov ? std::atoi(ov) == 2 || (std::atoi(ov) != 0 && prob->m >= (1 << 16)) : prob->m >= (1 << 16);     // 0 off, 2 whatever the size (tests)
    const uint64_t ind0 = g_ind_upload_count.load();
    std::atomic<bool> builder_done{false};
    try {
    run_threads(overlap_upload ? 2 : 1, [&](int share) {
        if (share == 1) {
            // after the builder's own (small) uploads: see g_ind_upload_count
            while (g_ind_upload_count.load() == ind0 && !builder_done.load()) std::this_thread::sleep_for(std::chrono::microseconds(20));
            (void)upload_rij(prob, p->device, &d_rij);       // failed: d_rij stays NULL, the handle uploads for itself
            return;
        }
        struct Done { std::atomic<bool>& f; ~Done() { f.store(true); } } done{builder_done};
        rc = structure_build_checked(prob, p->n_sample_min > 0 ? p->n_sample_min : 30, p->seed, p->build_where, p->device, &s, true);
        if (rc == DESC_ERR_TOO_LARGE && p->build_where == DESC_BUILD_DEVICE)      // device budget exceeded: host builder
            rc = structure_build_checked(prob, p->n_sample_min > 0 ? p->n_sample_min : 30, p->seed, DESC_BUILD_HOST, p->device, &s, true);
    });
    } catch (...) { release_rij(d_rij, p->device); if (s) { structure_free_device(s); delete s; } throw; }
    if (rc) { release_rij(d_rij, p->device); return rc; }
    double ms_structure = s->ms_build;
    lap("structure");
    desc_pgd* h = nullptr;
    rc = d_rij ? pgd_create_with_rij(prob, d_rij, s, p->device, &h) : desc_pgd_create(prob, s, p->device, &h);
    release_rij(d_rij, p->device);
    lap("create");
    structure_free_device(s);
    lap("structure free dev");
    delete s;
    lap("structure free host");
    if (rc) return rc;
    rc = desc_pgd_run(h, p, r);
    lap("run + download");
    desc_pgd_destroy(h);
    lap("destroy");
    r->ms_structure = ms_structure;
    r->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
    });
}

}  // extern "C"
