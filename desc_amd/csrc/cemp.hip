// CEMP: cycle-edge message passing, Algorithms/CEMP.m:24-132.
//   :44-65   nsample 3-cycles per edge-with-cycles, sampled WITH replacement
//   :70-103  S0Mat(s,l) = |acos((tr(Rij Rjk Rki)-1)/2)|/pi, SVec = column means, 1 without cycles
//   :107-128 T rounds: w = exp(-beta (s_ik + s_jk)), column-normalised, s_ij = sum w .* S0
// The reference signs IndMat (CEMP.m:75-76) but only ever uses abs() of it; orientation
// matters solely for fetching R_jk / R_ki (stored block or its transpose).
// Same sweep shape as the PGD hot path with a fixed number of cycles per edge: one wave per
// edge (nsample <= 64 lanes; longer samples loop), two gathers of S per cycle, two wave
// reductions.
// Round 4: the rounds run on a CSR-ALIGNED copy of SVec (every edge value in both endpoint rows, as the PGD path's Sfull) in 2-D tiles
// (band of i-rows) x (block of j): the band's rows of S sit in the LDS (S({k,i}) = an LDS read), the rows of the j-block stay in the L2s
// because the grid walks the tiles j-block-major (S({j,k}) = a gather inside row j), and a tile's edges -- Ind is sorted by (i, j) -- are
// contiguous runs of the per-cycle arrays.  No host-side plan: a tile finds its edges by two binary searches per row.  Per cycle and round:
// 12 B streamed (packed row positions + S0) instead of 16 B + two random 8-byte gathers over the 20 MB of SVec (C4: 1.82 ms per round).
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
                                                 double* S_a, double* S_b, int m_pos, int nsample, const int32_t* slot_a, const int32_t* slot_b) {
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
        if (lane == 0) {                                                                                     // :102
            const double mean = acc / (double)nsample;
            if (slot_a) { const int sa = slot_a[e], sb = slot_b[e]; S_a[sa] = mean; S_a[sb] = mean; S_b[sa] = mean; S_b[sb] = mean; }      // CSR-aligned copies
            else { S_a[e] = mean; S_b[e] = mean; }
        }
    }
}

// The same with the rotation blocks of node i's row staged in the LDS (round 4; CSR-aligned path: `pk` holds the position of k in row i): one
// 512-thread workgroup per node i, its edges (i, j) -- consecutive in the (i, j)-sorted list: node_seg[i] .. node_seg[i + 1] -- a wave each, R_ki of
// every sample an LDS read (80 B per slot for 16-byte reads); R_jk stays a gather from the edge table.  Same arithmetic in the same order.
__global__ __launch_bounds__(512) void k_cemp_s0_staged(const int32_t* pos_edge, const int32_t* ind_i, const int32_t* ind_j, const int32_t* kk, const int32_t* e_jk,
                                                        const uint32_t* pk, const int32_t* rowptr, const int32_t* adj_eid, const int32_t* node_seg, const double* rij, double* S0,
                                                        double* S_a, double* S_b, int n, int nsample, const int32_t* slot_a, const int32_t* slot_b) {
    extern __shared__ double s_blk[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int l0 = node_seg[i], l1 = node_seg[i + 1];
        if (l0 == l1) continue;
        const int r0 = rowptr[i], d = rowptr[i + 1] - r0;
        __syncthreads();
        for (int t = threadIdx.x; t < d; t += 512) {
            double Bk[9];
            load_block9(rij + 9 * (int64_t)adj_eid[r0 + t], Bk);
            for (int q = 0; q < 9; ++q) s_blk[10 * t + q] = Bk[q];
        }
        __syncthreads();
        for (int64_t l = l0 + wv; l < l1; l += 8) {
            const int e = pos_edge[l], j = ind_j[e];
            double A[9];
            load_block9(rij + 9 * (int64_t)e, A);
            double acc = 0.0;
            for (int s = lane; s < nsample; s += 64) {
                const int64_t c = l * nsample + s;
                const int k = kk[c];
                double pb[9], pc[9];
                load_block9(rij + 9 * (int64_t)e_jk[c], pb);
                const double* sb = s_blk + 10 * (int)(pk[c] & 0xFFFFu);
                const double2 c0 = *reinterpret_cast<const double2*>(sb), c1 = *reinterpret_cast<const double2*>(sb + 2), c2 = *reinterpret_cast<const double2*>(sb + 4),
                              c3 = *reinterpret_cast<const double2*>(sb + 6);
                pc[0] = c0.x; pc[1] = c0.y; pc[2] = c1.x; pc[3] = c1.y; pc[4] = c2.x; pc[5] = c2.y; pc[6] = c3.x; pc[7] = c3.y; pc[8] = sb[8];
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
                const double dd = abs_acos_ext_c((tr - 1.0) / 2.0) / M_PI;
                S0[c] = dd;
                acc += dd;
            }
            acc = group_sum<64>(acc);
            if (lane == 0) {                                                                                 // :102
                const double mean = acc / (double)nsample;
                const int sa = slot_a[e], sb2 = slot_b[e];
                S_a[sa] = mean; S_a[sb2] = mean; S_b[sa] = mean; S_b[sb2] = mean;
            }
        }
    }
}
// first natural segment (edge with cycles, ascending edge id) whose smaller endpoint is >= v; node_seg[n] = m_pos
__global__ __launch_bounds__(256) void k_cemp_node_seg(const int32_t* pos_edge, const int32_t* ind_i, int64_t m_pos, int n, int32_t* node_seg) {
    for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < m_pos; l += (int64_t)gridDim.x * 256) {
        const int i = ind_i[pos_edge[l]], prev = l > 0 ? ind_i[pos_edge[l - 1]] : -1;
        for (int v = prev + 1; v <= i; ++v) node_seg[v] = (int32_t)l;
        if (l == m_pos - 1) for (int v = i + 1; v <= n; ++v) node_seg[v] = (int32_t)m_pos;
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
            const double rws = 1.0 / wsum;                 // one division per edge (the reference divides every weight: the same to 1 ulp)
#pragma unroll
            for (int u = 0; u < 4; ++u) if (lane + 64 * u < nsample) acc += (wr[u] * rws) * dr[u];   // :122-125
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

// slots of every edge in its two endpoint rows of the CSR-aligned arrays
__global__ __launch_bounds__(256) void k_cemp_slots(const int32_t* ind_i, const int32_t* ind_j, const int32_t* rowptr, const int32_t* adj, int32_t* slot_a, int32_t* slot_b, int64_t m) {
    auto slot = [&](int v, int u) {
        const int r = rowptr[v]; int lo = 0, hi = rowptr[v + 1] - r;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[r + mid] < u) lo = mid + 1; else hi = mid; }
        return r + lo;
    };
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (int64_t)gridDim.x * 256) { slot_a[e] = slot(ind_i[e], ind_j[e]); slot_b[e] = slot(ind_j[e], ind_i[e]); }
}
__global__ __launch_bounds__(256) void k_cemp_extract(const double* Sfull, const int32_t* slot_a, double* S_vec, int64_t m) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (int64_t)gridDim.x * 256) S_vec[e] = Sfull[slot_a[e]];
}

// One round (CEMP.m:107-128) on the CSR-aligned copy, one workgroup per tile (band of BI consecutive nodes i) x (block of JB consecutive j); tiles are
// numbered j-block-major, so the workgroups in flight at any moment gather S({j,k}) from the same JB rows.  poe: edge -> index among the edges
// with cycles (NULL: every edge has cycles and the map is the identity).
constexpr int CEMP_MAXB = 16;                      // nodes per band at most
__global__ __launch_bounds__(256) void k_cemp_round_tile(const int32_t* rowptr, const int32_t* adj, const int32_t* adj_eid, const int32_t* poe, const int32_t* slot_b,
                                                         const uint32_t* pk, const double* S0, const double* S_old, double* S_new, int n, int BI, int JB, int n_iband,
                                                         int nsample, double beta) {
    extern __shared__ double s_rows[];             // rows of the band: S_old[rowptr[i0] .. rowptr[i1])
    __shared__ int s_a[CEMP_MAXB], s_cnt[CEMP_MAXB + 1], s_rb[CEMP_MAXB];      // per node of the band: first slot of its run, edges in the tile, its row's offset in s_rows
    const int jb = blockIdx.x / n_iband, ib = blockIdx.x % n_iband;
    const int i0 = ib * BI, i1 = min(n, i0 + BI);
    const int j_lo = jb * JB, j_hi = min(n, j_lo + JB);
    if (j_hi <= i0 + 1) return;                    // every j of the block is <= every i of the band: no edge (i, j), i < j
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row_lo = rowptr[i0], row_len = rowptr[i1] - row_lo;
    if (tid < i1 - i0) {                           // the band's edges with j in the block: a contiguous run of row i's slots
        const int i = i0 + tid, r0 = rowptr[i], d = rowptr[i + 1] - r0;
        auto below = [&](int x) { int lo = 0, hi = d; while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[r0 + mid] < x) lo = mid + 1; else hi = mid; } return lo; };
        const int a = below(max(j_lo, i + 1)), b = below(j_hi);
        s_rb[tid] = r0 - row_lo;
        s_a[tid] = r0 + a; s_cnt[tid] = max(b - a, 0);        // a node of the band beyond the block's last j: no edge (b < a)
    }
    __syncthreads();
    int total = 0;
    for (int t = 0; t < i1 - i0; ++t) total += s_cnt[t];
    if (total == 0) return;
    for (int t = tid; t < row_len; t += 256) s_rows[t] = S_old[row_lo + t];
    __syncthreads();
    if (nsample <= 64) {
        // one sample per lane: CEMP_U edges of the tile per wave and trip, every stage's loads of all of them issued before any is used (each edge is
        // a chain of four dependent memory round trips -- slot -> row start of j and the packed word -> the gather in row j -> ... --: one edge at a
        // time the round measured 1.46 ms at C4, profiles/r04_cemp_tiles.txt)
        constexpr int CEMP_U = 4;                  // 2 / 4 / 6 / 8 edges in flight: 0.98 / 0.97 / 0.98 / 1.04 ms (profiles/r04_cemp_scalar_setup.txt): the round is issue-bound
        for (int x0 = wv; x0 < total; x0 += 4 * CEMP_U) {
            int rbi[CEMP_U], slot[CEMP_U], ee[CEMP_U], jn[CEMP_U]; bool on[CEMP_U];
#pragma unroll
            for (int u = 0; u < CEMP_U; ++u) {
                const int x = x0 + 4 * u;
                on[u] = x < total;
                int t = 0, y = on[u] ? x : 0;
                while (y >= s_cnt[t]) { y -= s_cnt[t]; ++t; }
                // everything about the edge is the same in all 64 lanes: scalar registers and scalar loads from here on (round 1.12 -> 0.97 ms at C4)
                rbi[u] = __builtin_amdgcn_readfirstlane(s_rb[t]); slot[u] = __builtin_amdgcn_readfirstlane(s_a[t] + y);
                ee[u] = uniform_load(adj_eid, slot[u]); jn[u] = uniform_load(adj, slot[u]);
            }
            int ll[CEMP_U], rbj[CEMP_U]; uint32_t pw[CEMP_U]; double dd[CEMP_U];
#pragma unroll
            for (int u = 0; u < CEMP_U; ++u) {
                ll[u] = poe ? uniform_load(poe, ee[u]) : ee[u];
                if (ll[u] < 0) on[u] = false;                   // no cycles: SVec stays 1 (:103, :126)
                rbj[u] = uniform_load(rowptr, jn[u]);
            }
#pragma unroll
            for (int u = 0; u < CEMP_U; ++u) {
                const int64_t c = (int64_t)max(ll[u], 0) * nsample + min(lane, nsample - 1);
                pw[u] = pk[c]; dd[u] = S0[c];
            }
            double sj[CEMP_U];
#pragma unroll
            for (int u = 0; u < CEMP_U; ++u) sj[u] = S_old[rbj[u] + (int)(pw[u] >> 16)];
#pragma unroll
            for (int u = 0; u < CEMP_U; ++u) {
                const bool act = lane < nsample;
                const double si = s_rows[rbi[u] + (int)(pw[u] & 0xFFFFu)];
                const double w = act ? exp(-beta * (si + sj[u])) : 0.0;                                     // :118-120  s_ik + s_jk
                const double rws = 1.0 / group_sum<64>(w);      // one division per edge (the reference divides every weight: the same to 1 ulp)
                const double acc = group_sum<64>(act ? (w * rws) * dd[u] : 0.0);                            // :122-125
                if (lane == 0 && on[u]) { S_new[slot[u]] = acc; S_new[slot_b[ee[u]]] = acc; }
            }
        }
        return;
    }
    for (int x = wv; x < total; x += 4) {          // one wave per edge of the tile
        int t = 0, y = x;
        while (y >= s_cnt[t]) { y -= s_cnt[t]; ++t; }
        const int i = i0 + t, slot = s_a[t] + y;   // the edge (i, j): slot of j in row i
        const int e = adj_eid[slot], j = adj[slot];
        const int l = poe ? poe[e] : e;
        if (l < 0) continue;                       // no cycles: SVec stays 1 (:103, :126)
        const int rbi = rowptr[i] - row_lo, rbj = rowptr[j];
        double wr[4] = {0.0, 0.0, 0.0, 0.0}, dr[4] = {0.0, 0.0, 0.0, 0.0};
        double wsum = 0.0, acc = 0.0;
        if (nsample <= 4 * 64) {                   // weights stay in registers: one pass over the samples
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sidx = lane + 64 * u;
                if (sidx < nsample) {
                    const int64_t c = (int64_t)l * nsample + sidx;
                    const uint32_t p = pk[c];
                    wr[u] = exp(-beta * (s_rows[rbi + (int)(p & 0xFFFFu)] + S_old[rbj + (int)(p >> 16)]));      // :118-120  s_ik + s_jk
                    dr[u] = S0[c];
                    wsum += wr[u];
                }
            }
            wsum = group_sum<64>(wsum);
            const double rws = 1.0 / wsum;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (lane + 64 * u < nsample) acc += (wr[u] * rws) * dr[u];                 // :122-125
        } else {
            for (int sidx = lane; sidx < nsample; sidx += 64) {
                const uint32_t p = pk[(int64_t)l * nsample + sidx];
                wsum += exp(-beta * (s_rows[rbi + (int)(p & 0xFFFFu)] + S_old[rbj + (int)(p >> 16)]));
            }
            wsum = group_sum<64>(wsum);
            for (int sidx = lane; sidx < nsample; sidx += 64) {
                const int64_t c = (int64_t)l * nsample + sidx;
                const uint32_t p = pk[c];
                const double w = exp(-beta * (s_rows[rbi + (int)(p & 0xFFFFu)] + S_old[rbj + (int)(p >> 16)]));
                acc += (w / wsum) * S0[c];
            }
        }
        acc = group_sum<64>(acc);
        if (lane == 0) { S_new[slot] = acc; S_new[slot_b[e]] = acc; }
    }
}

__global__ void k_fill1(double* p, int64_t n, double v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ __launch_bounds__(256) void k_cemp_inverse(const int32_t* pos_edge, int32_t* poe, int64_t m_pos) {
    for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < m_pos; l += (int64_t)gridDim.x * 256) poe[pos_edge[l]] = (int32_t)l;
}
int env_int_c(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }

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
    uint32_t* d_pk = nullptr;
    int32_t max_deg = 0;
    struct Owned { int32_t **a, **b, **c, **d; uint32_t** e; ~Owned() { for (int32_t** q : {a, b, c, d}) if (*q) dev_free(*q); if (*e) dev_free(*e); } } owned{&d_pos, &d_k, &d_ejk, &d_eki, &d_pk};
    const bool want_tiles = env_int_c("DESC_DEBUG_CEMP_TILES", 1) != 0;
    rc = build_cemp_samples_device(dp, nsample, seed, &mp, &d_pos, &d_k, &d_ejk, &d_eki, want_tiles ? &d_pk : nullptr, &max_deg);
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
    const int64_t n = dp->n;
    DevC D;
    const int32_t *d_ii = dp->d_ii, *d_jj = dp->d_jj; const double* d_rij = dp->d_rij; double *d_S0, *d_S[2];
    // tiles: the device sampler delivered the packed row positions (rows shorter than 2^16); the band of one tile must fit the LDS
    const bool tiles = d_pk != nullptr && mp > 0 && max_deg > 0 && (size_t)max_deg * sizeof(double) <= 64 * 1024;
    const int64_t slen = tiles ? 2 * m : m;                                          // CSR-aligned: every edge value in both endpoint rows
    int32_t *d_slot_a = nullptr, *d_slot_b = nullptr, *d_poe = nullptr; double* d_out = nullptr;
    if ((rc = D.alloc(&d_S0, mc)) || (rc = D.alloc(&d_S[0], slen)) || (rc = D.alloc(&d_S[1], slen))) return rc;
    if (tiles && ((rc = D.alloc(&d_slot_a, m)) || (rc = D.alloc(&d_slot_b, m)) || (rc = D.alloc(&d_out, m)))) return rc;
    if (m) {
        const int g = (int)std::min<int64_t>(1024, (slen + 255) / 256);
        hipLaunchKernelGGL(k_fill1, dim3(g), dim3(256), 0, 0, d_S[0], slen, 1.0);     // SVec(~IndPosbin) = 1 (:103)
        hipLaunchKernelGGL(k_fill1, dim3(g), dim3(256), 0, 0, d_S[1], slen, 1.0);
        if (tiles) hipLaunchKernelGGL(k_cemp_slots, dim3((unsigned)std::min<int64_t>(4096, (m + 255) / 256)), dim3(256), 0, 0, d_ii, d_jj, dp->d_rowptr, dp->d_adj, d_slot_a, d_slot_b, m);
    }
    int cur = 0;
    if (mp) {
        const int g = (int)std::min<int64_t>(8192, (mp + 3) / 4);
        const size_t lds0 = (size_t)max_deg * 10 * sizeof(double);
        if (tiles && lds0 <= 150 * 1024 && env_int_c("DESC_DEBUG_STAGED_LAYOUT", 1) != 0) {       // node i's rotation blocks in the LDS
            int32_t* d_node_seg = nullptr;
            if ((rc = D.alloc(&d_node_seg, n + 1))) return rc;
            hipLaunchKernelGGL(k_cemp_node_seg, dim3((unsigned)std::min<int64_t>(4096, (mp + 255) / 256)), dim3(256), 0, 0, d_pos, d_ii, mp, (int)n, d_node_seg);
            if (lds0 > 64 * 1024) DESC_HIP(hipFuncSetAttribute((const void*)k_cemp_s0_staged, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
            hipLaunchKernelGGL(k_cemp_s0_staged, dim3((unsigned)std::max<int64_t>(1, n)), dim3(512), lds0, 0, d_pos, d_ii, d_jj, d_k, d_ejk, d_pk, dp->d_rowptr, dp->d_adj_eid, d_node_seg, d_rij,
                               d_S0, d_S[0], d_S[1], (int)n, nsample, d_slot_a, d_slot_b);
        } else
        hipLaunchKernelGGL(k_cemp_s0, dim3(g), dim3(256), 0, 0, d_pos, d_ii, d_jj, d_k, d_ejk, d_eki, d_rij, d_S0, d_S[0], d_S[1], (int)mp, nsample,
                           tiles ? d_slot_a : (const int32_t*)nullptr, tiles ? d_slot_b : (const int32_t*)nullptr);
        // tile shape: the band's rows in <= 32 KiB of LDS (several workgroups per CU), the rows of a j-block ~2.5 MiB (they share an XCD's L2 with the streams)
        const int64_t avg_deg = std::max<int64_t>(1, 2 * m / std::max<int64_t>(1, n));
        int BI = (int)std::max<int64_t>(1, std::min<int64_t>(CEMP_MAXB, (32 * 1024 / 8) / std::max(max_deg, 1)));
        BI = std::max(1, std::min(BI, env_int_c("DESC_DEBUG_CEMP_BI", BI)));
        const int JB = (int)std::max<int64_t>(32, env_int_c("DESC_DEBUG_CEMP_JB", (int)std::max<int64_t>(32, (5ll << 19) / (8 * avg_deg))));      // measured: flat from ~2.5 MiB of rows on (profiles/r04_cemp_tile_scan.txt)
        const int n_iband = (int)((n + BI - 1) / BI), n_jblock = (int)((n + JB - 1) / JB);
        const size_t lds = (size_t)BI * (size_t)max_deg * sizeof(double);
        if (tiles) {
            if (mp != m) {                                                           // edge -> index among the edges with cycles
                if ((rc = D.alloc(&d_poe, m))) return rc;
                DESC_HIP(hipMemsetAsync(d_poe, 0xFF, sizeof(int32_t) * m, 0));
                hipLaunchKernelGGL(k_cemp_inverse, dim3((unsigned)std::min<int64_t>(4096, (mp + 255) / 256)), dim3(256), 0, 0, d_pos, d_poe, mp);
            }
            if (lds > 64 * 1024) DESC_HIP(hipFuncSetAttribute((const void*)k_cemp_round_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        for (int it = 0; it < max_iter; ++it) {                                     // :107
            const double b = beta[it < n_beta ? it : n_beta - 1];                   // :30-34: missing betas repeat the last one
            if (tiles)
                hipLaunchKernelGGL(k_cemp_round_tile, dim3((unsigned)((int64_t)n_iband * n_jblock)), dim3(256), lds, 0, dp->d_rowptr, dp->d_adj, dp->d_adj_eid, d_poe, d_slot_b,
                                   d_pk, d_S0, d_S[cur], d_S[cur ^ 1], (int)n, BI, JB, n_iband, nsample, b);
            else
                hipLaunchKernelGGL(k_cemp_round, dim3(g), dim3(256), 0, 0, d_pos, d_ejk, d_eki, d_S0, d_S[cur], d_S[cur ^ 1], (int)mp, nsample, b);
            cur ^= 1;
        }
        DESC_HIP(hipGetLastError());
    }
    if (tiles && m) hipLaunchKernelGGL(k_cemp_extract, dim3((unsigned)std::min<int64_t>(2048, (m + 255) / 256)), dim3(256), 0, 0, d_S[cur], d_slot_a, d_out, m);
    DESC_HIP(hipDeviceSynchronize());
    if (m) DESC_HIP(hipMemcpy(s_vec, tiles ? d_out : d_S[cur], sizeof(double) * m, hipMemcpyDeviceToHost));
    if (ms_total) *ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DESC_OK;
    });
}
