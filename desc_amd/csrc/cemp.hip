// CEMP: cycle-edge message passing, Algorithms/CEMP.m:24-132.
//   :44-65   nsample 3-cycles per edge-with-cycles, sampled WITH replacement
//   :70-103  S0Mat(s,l) = |acos((tr(Rij Rjk Rki)-1)/2)|/pi, SVec = column means, 1 without cycles
//   :107-128 T rounds: w = exp(-beta (s_ik + s_jk)), column-normalised, s_ij = sum w .* S0
// The reference signs IndMat (CEMP.m:75-76) but only ever uses abs() of it; orientation
// matters solely for fetching R_jk / R_ki (stored block or its transpose).
// Same sweep shape as the PGD hot path with a fixed number of cycles per edge: one wave per
// edge (nsample <= 64 lanes; longer samples loop), two gathers of S per cycle, two wave
// reductions.  HBM-bound (24 B streamed + 2 gathers per cycle and round).
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "device_utils.h"

namespace desc {
namespace {

__device__ __forceinline__ double abs_acos_ext_c(double x) {
    if (x > 1.0) return acosh(x);
    if (x < -1.0) return hypot(M_PI, acosh(-x));
    return acos(x);
}

__global__ __launch_bounds__(256) void k_cemp_s0(const int32_t* pos_edge, const int32_t* ind_i, const int32_t* ind_j, const int32_t* kk,
                                                 const int32_t* e_jk, const int32_t* e_ki, const double* rij, double* S0,
                                                 double* S_a, double* S_b, int m_pos, int nsample) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        double A[9];
        load_block9(rij + 9 * (int64_t)e, A);
        double acc = 0.0;
        for (int s = lane; s < nsample; s += 64) {
            const int64_t c = l * nsample + s;
            const int k = kk[c];
            double pb[9], pc[9];                           // the two gathered blocks, in registers (16-byte loads)
            load_block9(rij + 9 * (int64_t)e_jk[c], pb);
            load_block9(rij + 9 * (int64_t)e_ki[c], pc);
            const bool tb = !(j < k), tc = !(k < i);
            double tr = 0.0;
            for (int r = 0; r < 3; ++r) {
                double P[3];
                for (int q = 0; q < 3; ++q) {
                    double a2 = 0.0;
                    for (int u = 0; u < 3; ++u) a2 = a2 + A[r + 3 * u] * (tb ? pb[q + 3 * u] : pb[u + 3 * q]);
                    P[q] = a2;
                }
                double a3 = 0.0;
                for (int u = 0; u < 3; ++u) a3 = a3 + P[u] * (tc ? pc[r + 3 * u] : pc[u + 3 * r]);
                tr = tr + a3;
            }
            const double d = abs_acos_ext_c((tr - 1.0) / 2.0) / M_PI;
            S0[c] = d;
            acc += d;
        }
        acc = group_sum<64>(acc);
        if (lane == 0) { const double mean = acc / (double)nsample; S_a[e] = mean; S_b[e] = mean; }      // :102
    }
}

__global__ __launch_bounds__(256) void k_cemp_round(const int32_t* pos_edge, const int32_t* e_jk, const int32_t* e_ki, const double* S0,
                                                    const double* S_old, double* S_new, int m_pos, int nsample, double beta) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        if (nsample <= 4 * 64) {                       // weights stay in registers: one pass over the samples
            double wr[4] = {0.0, 0.0, 0.0, 0.0}, dr[4] = {0.0, 0.0, 0.0, 0.0};
            double wsum = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = lane + 64 * u;
                if (s < nsample) {
                    const int64_t c = l * nsample + s;
                    wr[u] = exp(-beta * (S_old[e_ki[c]] + S_old[e_jk[c]]));      // :118-120
                    dr[u] = S0[c];
                    wsum += wr[u];
                }
            }
            wsum = group_sum<64>(wsum);
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (lane + 64 * u < nsample) acc += (wr[u] / wsum) * dr[u];   // :122-125
            acc = group_sum<64>(acc);
            if (lane == 0) S_new[pos_edge[l]] = acc;
            continue;
        }
        double wsum = 0.0;
        for (int s = lane; s < nsample; s += 64) {
            const int64_t c = l * nsample + s;
            wsum += exp(-beta * (S_old[e_ki[c]] + S_old[e_jk[c]]));                  // :118-120
        }
        wsum = group_sum<64>(wsum);
        double acc = 0.0;
        for (int s = lane; s < nsample; s += 64) {
            const int64_t c = l * nsample + s;
            const double w = exp(-beta * (S_old[e_ki[c]] + S_old[e_jk[c]]));
            acc += (w / wsum) * S0[c];                                               // :122-125
        }
        acc = group_sum<64>(acc);
        if (lane == 0) S_new[pos_edge[l]] = acc;
    }
}

__global__ void k_fill1(double* p, int64_t n, double v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

struct DevC {
    hvec<void*> p;
    ~DevC() { for (void* q : p) dev_free(q); }
    template <class T> int alloc(T** out, size_t count) {
        void* q = nullptr;
        DESC_HIP(dev_alloc(&q, sizeof(T) * (count ? count : 1)));
        p.push_back(q); *out = (T*)q;
        return DESC_OK;
    }
};

}  // namespace
}  // namespace desc

using namespace desc;

extern "C" int desc_cemp_run(const desc_problem* prob, const double* beta, int32_t n_beta, int32_t max_iter, int32_t nsample,
                             uint64_t seed, int32_t device, double* s_vec, double* ms_total) {
    if (!prob || !s_vec || !beta) return fail(DESC_ERR_INVALID, "NULL argument");
    auto t0 = std::chrono::steady_clock::now();
    desc_device_problem* dp = nullptr;
    int rc = desc_problem_upload(prob, device, &dp);
    if (rc) return rc;
    rc = desc_cemp_run_dev(dp, beta, n_beta, max_iter, nsample, seed, s_vec, nullptr);
    desc_problem_free(dp);
    if (ms_total) *ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" int desc_cemp_run_dev(const desc_device_problem* dp, const double* beta, int32_t n_beta, int32_t max_iter, int32_t nsample,
                                 uint64_t seed, double* s_vec, double* ms_total) {
    return no_throw("desc_cemp_run_dev", [&]() -> int {
    if (!dp || !s_vec || !beta) return fail(DESC_ERR_INVALID, "NULL argument");
    if (n_beta < 1 || max_iter < 0 || nsample < 1) return fail(DESC_ERR_INVALID, "need n_beta >= 1, max_iter >= 0, nsample >= 1");
    int rc = DESC_OK;
    auto t0 = std::chrono::steady_clock::now();
    DESC_HIP(hipSetDevice(dp->device));
    const int64_t m = dp->m;
    // samples: on the device; graphs beyond the device sampler's staging budget fall back to the host sampler
    int64_t mp = 0;
    int32_t *d_pos = nullptr, *d_k = nullptr, *d_ejk = nullptr, *d_eki = nullptr;
    struct Owned { int32_t **a, **b, **c, **d; ~Owned() { for (int32_t** q : {a, b, c, d}) if (*q) dev_free(*q); } } owned{&d_pos, &d_k, &d_ejk, &d_eki};
    rc = build_cemp_samples_device(dp, nsample, seed, &mp, &d_pos, &d_k, &d_ejk, &d_eki);
    if (rc == DESC_ERR_TOO_LARGE) {
        hvec<int32_t> pos_edge, kk, e_jk, e_ki;
        const desc_problem hv = host_view(dp);
        if ((rc = build_cemp_samples_host(&hv, nsample, seed, pos_edge, kk, e_jk, e_ki))) return rc;
        mp = (int64_t)pos_edge.size();
        const int64_t mch = mp * nsample;
        if (mch >= (1ll << 31)) return fail(DESC_ERR_TOO_LARGE, "m_pos * nsample exceeds 2^31");
        if (mp) {
            DESC_HIP(dev_alloc((void**)&d_pos, sizeof(int32_t) * mp)); DESC_HIP(dev_alloc((void**)&d_k, sizeof(int32_t) * mch));
            DESC_HIP(dev_alloc((void**)&d_ejk, sizeof(int32_t) * mch)); DESC_HIP(dev_alloc((void**)&d_eki, sizeof(int32_t) * mch));
            DESC_HIP(hipMemcpy(d_pos, pos_edge.data(), sizeof(int32_t) * mp, hipMemcpyHostToDevice));
            DESC_HIP(hipMemcpy(d_k, kk.data(), sizeof(int32_t) * mch, hipMemcpyHostToDevice));
            DESC_HIP(hipMemcpy(d_ejk, e_jk.data(), sizeof(int32_t) * mch, hipMemcpyHostToDevice));
            DESC_HIP(hipMemcpy(d_eki, e_ki.data(), sizeof(int32_t) * mch, hipMemcpyHostToDevice));
        }
    } else if (rc) return rc;
    const int64_t mc = mp * nsample;
    DevC D;
    const int32_t *d_ii = dp->d_ii, *d_jj = dp->d_jj; const double* d_rij = dp->d_rij; double *d_S0, *d_S[2];
    if ((rc = D.alloc(&d_S0, mc)) || (rc = D.alloc(&d_S[0], m)) || (rc = D.alloc(&d_S[1], m))) return rc;
    if (m) {
        const int g = (int)std::min<int64_t>(1024, (m + 255) / 256);
        hipLaunchKernelGGL(k_fill1, dim3(g), dim3(256), 0, 0, d_S[0], m, 1.0);     // SVec(~IndPosbin) = 1 (:103)
        hipLaunchKernelGGL(k_fill1, dim3(g), dim3(256), 0, 0, d_S[1], m, 1.0);
    }
    int cur = 0;
    if (mp) {
        const int g = (int)std::min<int64_t>(8192, (mp + 3) / 4);
        hipLaunchKernelGGL(k_cemp_s0, dim3(g), dim3(256), 0, 0, d_pos, d_ii, d_jj, d_k, d_ejk, d_eki, d_rij, d_S0, d_S[0], d_S[1], (int)mp, nsample);
        for (int it = 0; it < max_iter; ++it) {                                     // :107
            const double b = beta[it < n_beta ? it : n_beta - 1];                   // :30-34: missing betas repeat the last one
            hipLaunchKernelGGL(k_cemp_round, dim3(g), dim3(256), 0, 0, d_pos, d_ejk, d_eki, d_S0, d_S[cur], d_S[cur ^ 1], (int)mp, nsample, b);
            cur ^= 1;
        }
        DESC_HIP(hipGetLastError());
    }
    DESC_HIP(hipDeviceSynchronize());
    if (m) DESC_HIP(hipMemcpy(s_vec, d_S[cur], sizeof(double) * m, hipMemcpyDeviceToHost));
    if (ms_total) *ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DESC_OK;
    });
}
