// Device construction of the sampled 3-cycle structure.
//
// Same result, bit for bit, as the host builder (structure_host.cpp) and the oracle:
//   DESC_PGD.m:23-34   adjacency and per-edge codegree            k_bitmaps, k_codeg
//   DESC_PGD.m:36-54   edges with cycles, n_sample, cum_ind        host, O(m) on the codegrees
//   DESC_PGD.m:79-96   common-neighbour lists, keyed sampling      k_fill_cycles
//   DESC_PGD.m:103-127 mirror-cycle maps IKJ / JKI                 k_mirror
// Method: adjacency rows as bitmaps in HBM (n^2/8 bytes: 3 MB at n = 5000, L2 resident);
// one wave per edge ANDs the two rows (lanes over 64-bit words, popcount + wave scan give
// the ascending enumeration of common neighbours without any sort); a per-word prefix
// popcount turns a neighbour id into its CSR slot and hence its edge id.  Sampling keeps
// the n_sample smallest desc_sample_key values by rank counting in LDS (keys are staged
// once per edge; every lane counts how many keys precede its own).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "device_utils.h"

namespace desc {
namespace {

constexpr int MAX_CODEG_LDS = 1024;      // common neighbours staged per edge when sampling

__device__ __forceinline__ uint64_t d_mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ uint64_t d_sample_key(uint64_t seed, uint64_t edge, uint64_t k) {
    const uint64_t a = d_mix64(seed ^ ((edge + 1) * 0x9E3779B97F4A7C15ull));
    return d_mix64(a ^ ((k + 1) * 0xD1B54A32D192ED03ull));
}

// inclusive prefix sum of an int over the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ void k_bitmaps(const int32_t* rowptr, const int32_t* adj, unsigned long long* bits, int n, int words) {
    // one workgroup per node: its row of `words` 64-bit words is private to the workgroup
    for (int v = blockIdx.x; v < n; v += gridDim.x) {
        unsigned long long* row = bits + (size_t)v * words;
        for (int t = rowptr[v] + threadIdx.x; t < rowptr[v + 1]; t += blockDim.x)
            atomicOr(&row[adj[t] >> 6], 1ull << (adj[t] & 63));
    }
}
// rank[v][w] = number of neighbours of v in words < w
__global__ void k_rank(const unsigned long long* bits, uint32_t* rank, int n, int words) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t acc = 0;
    for (int w = 0; w < words; ++w) { rank[(size_t)v * words + w] = acc; acc += (uint32_t)__popcll(bits[(size_t)v * words + w]); }
}
// codegree of every edge: popcount(row_i & row_j), one wave per edge
__global__ __launch_bounds__(256) void k_codeg(const int32_t* ind_i, const int32_t* ind_j, const unsigned long long* bits,
                                               int32_t* codeg, int64_t m, int words) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t e = wid; e < m; e += nw) {
        const unsigned long long* a = bits + (size_t)ind_i[e] * words;
        const unsigned long long* b = bits + (size_t)ind_j[e] * words;
        int c = 0;
        for (int w = lane; w < words; w += 64) c += __popcll(a[w] & b[w]);
        c = wave_incl_scan(c, lane);
        if (lane == 63) codeg[e] = c;
    }
}

// Cycle lists (DESC_PGD.m:79-96): one wave per edge-with-cycles.
__global__ __launch_bounds__(256) void k_fill_cycles(const int32_t* pos_edge, const int32_t* cum, const int32_t* ind_i,
                                                     const int32_t* ind_j, const unsigned long long* bits, const uint32_t* rank,
                                                     const int32_t* rowptr, const int32_t* adj_eid, int32_t* kk, int32_t* e_jk,
                                                     int32_t* e_ki, int64_t m_pos, int words, int n_sample, uint64_t seed,
                                                     int lds_cap) {
    extern __shared__ unsigned long long smem[];      // per wave: keys[lds_cap] then (k, ejk, eki)[lds_cap] as int32
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long* keys = smem + (size_t)wv * lds_cap;
    int32_t* tri = (int32_t*)(smem + (size_t)4 * lds_cap) + (size_t)wv * 3 * lds_cap;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        const int base = cum[l], cnt = cum[l + 1] - base;
        const unsigned long long* a = bits + (size_t)i * words;
        const unsigned long long* b = bits + (size_t)j * words;
        const uint32_t* ra = rank + (size_t)i * words;
        const uint32_t* rb = rank + (size_t)j * words;
        const int r0i = rowptr[i], r0j = rowptr[j];
        // enumerate the common neighbours in ascending order; `run` = how many precede this word
        int run = 0;
        bool sampling = false;
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int w = w0 + lane;
            unsigned long long x = 0, aw = 0, bw = 0;
            if (w < words) { aw = a[w]; bw = b[w]; x = aw & bw; }
            const int pc = __popcll(x);
            const int incl = wave_incl_scan(pc, lane);
            int pos = run + incl - pc;
            run += __shfl(incl, 63, 64);
            while (x) {
                const int bit = __ffsll((long long)x) - 1;
                const unsigned long long below = (1ull << bit) - 1ull;
                const int k = w * 64 + bit;
                const int eki = adj_eid[r0i + ra[w] + __popcll(aw & below)];
                const int ejk = adj_eid[r0j + rb[w] + __popcll(bw & below)];
                if (pos < lds_cap) { tri[3 * pos] = k; tri[3 * pos + 1] = ejk; tri[3 * pos + 2] = eki; keys[pos] = d_sample_key(seed, (uint64_t)e, (uint64_t)k); }
                ++pos;
                x &= x - 1;
            }
        }
        const int cd = run;                           // codegree
        sampling = cd >= n_sample;                    // DESC_PGD.m:83 (>=)
        __builtin_amdgcn_wave_barrier();
        if (!sampling) {
            for (int t = lane; t < cd; t += 64) { kk[base + t] = tri[3 * t]; e_jk[base + t] = tri[3 * t + 1]; e_ki[base + t] = tri[3 * t + 2]; }
        } else {
            // keep the n_sample smallest (key, k); common neighbours are distinct, positions ascend with k
            int outbase = 0;
            for (int t0 = 0; t0 < cd; t0 += 64) {
                const int t = t0 + lane;
                bool sel = false;
                if (t < cd) {
                    const unsigned long long kt = keys[t];
                    int rk = 0;
                    for (int u = 0; u < cd; ++u) { const unsigned long long ku = keys[u]; rk += (ku < kt) || (ku == kt && u < t); }
                    sel = rk < n_sample;
                }
                const unsigned long long mk = __ballot(sel);
                if (sel) {
                    const int o = base + outbase + __popcll(mk & ((1ull << lane) - 1ull));
                    kk[o] = tri[3 * t]; e_jk[o] = tri[3 * t + 1]; e_ki[o] = tri[3 * t + 2];
                }
                outbase += __popcll(mk);
            }
        }
        (void)cnt;
        __builtin_amdgcn_wave_barrier();
    }
}

// Mirror maps (DESC_PGD.m:103-127): binary search of j in the sampled list of edge {i,k}
// and of i in that of edge {j,k}; one wave per edge-with-cycles, lanes over its cycles.
__global__ __launch_bounds__(256) void k_mirror(const int32_t* pos_edge, const int32_t* cum, const int32_t* pos_of_edge,
                                                const int32_t* ind_i, const int32_t* ind_j, const int32_t* kk,
                                                const int32_t* e_jk, const int32_t* e_ki, int32_t* ikj, int32_t* jki,
                                                int64_t m_pos) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        for (int c = cum[l] + lane; c < cum[l + 1]; c += 64) {
            int IK = pos_of_edge[e_ki[c]], lo = cum[IK], hi = cum[IK + 1];
            const int end1 = hi;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (kk[mid] < j) lo = mid + 1; else hi = mid; }
            ikj[c] = (lo < end1 && kk[lo] == j) ? lo : -1;
            int JK = pos_of_edge[e_jk[c]];
            lo = cum[JK]; hi = cum[JK + 1];
            const int end2 = hi;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (kk[mid] < i) lo = mid + 1; else hi = mid; }
            jki[c] = (lo < end2 && kk[lo] == i) ? lo : -1;
        }
    }
}

struct DevBuf {
    std::vector<void*> p;
    ~DevBuf() { for (void* q : p) if (q) (void)hipFree(q); }
    template <class T> int alloc(T** out, size_t count) {
        void* q = nullptr;
        DESC_HIP(hipMalloc(&q, sizeof(T) * (count ? count : 1)));
        p.push_back(q); *out = (T*)q;
        return DESC_OK;
    }
};

}  // namespace

int build_structure_device(const desc_problem* prob, int32_t n_sample_min, uint64_t seed, int32_t device, desc_structure* s) {
    auto t0 = std::chrono::steady_clock::now();
    const int64_t n = prob->n, m = prob->m;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DESC_ERR_HIP, "no HIP device visible for DESC_BUILD_DEVICE");
    if (device < 0 || device >= ndev) return fail(DESC_ERR_INVALID, "device %d out of range", device);
    const int64_t words = (n + 63) / 64;
    if ((double)n * (double)words * 12.0 > 64.0 * 1073741824.0)
        return fail(DESC_ERR_TOO_LARGE, "adjacency bitmaps of n = %lld nodes do not fit the device-build budget; use DESC_BUILD_HOST", (long long)n);
    DESC_HIP(hipSetDevice(device));
    s->n = n; s->m = m;

    // CSR on the host (one pass; Ind is sorted by (i,j), so rows come out ascending)
    std::vector<int32_t> rowptr((size_t)n + 1, 0), adj((size_t)2 * m), adj_eid((size_t)2 * m);
    for (int64_t e = 0; e < m; ++e) { rowptr[prob->ind_i[e] + 1]++; rowptr[prob->ind_j[e] + 1]++; }
    for (int64_t v = 0; v < n; ++v) rowptr[v + 1] += rowptr[v];
    {
        std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
        for (int64_t e = 0; e < m; ++e) {
            const int32_t i = prob->ind_i[e], j = prob->ind_j[e];
            adj[fill[i]] = j; adj_eid[fill[i]++] = (int32_t)e;
            adj[fill[j]] = i; adj_eid[fill[j]++] = (int32_t)e;
        }
    }
    DevBuf D;
    int rc;
    int32_t *d_rowptr, *d_adj, *d_adj_eid, *d_ii, *d_jj, *d_codeg;
    unsigned long long* d_bits; uint32_t* d_rank;
    if ((rc = D.alloc(&d_rowptr, n + 1)) || (rc = D.alloc(&d_adj, 2 * m)) || (rc = D.alloc(&d_adj_eid, 2 * m)) ||
        (rc = D.alloc(&d_ii, m)) || (rc = D.alloc(&d_jj, m)) || (rc = D.alloc(&d_codeg, m)) ||
        (rc = D.alloc(&d_bits, (size_t)n * words)) || (rc = D.alloc(&d_rank, (size_t)n * words))) return rc;
    DESC_HIP(hipMemcpy(d_rowptr, rowptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
    if (m) {
        DESC_HIP(hipMemcpy(d_adj, adj.data(), sizeof(int32_t) * 2 * m, hipMemcpyHostToDevice));
        DESC_HIP(hipMemcpy(d_adj_eid, adj_eid.data(), sizeof(int32_t) * 2 * m, hipMemcpyHostToDevice));
        DESC_HIP(hipMemcpy(d_ii, prob->ind_i, sizeof(int32_t) * m, hipMemcpyHostToDevice));
        DESC_HIP(hipMemcpy(d_jj, prob->ind_j, sizeof(int32_t) * m, hipMemcpyHostToDevice));
    }
    DESC_HIP(hipMemset(d_bits, 0, sizeof(unsigned long long) * (size_t)n * words));
    if (n > 0) {
        hipLaunchKernelGGL(k_bitmaps, dim3((unsigned)std::min<int64_t>(n, 4096)), dim3(256), 0, 0, d_rowptr, d_adj, d_bits, (int)n, (int)words);
        hipLaunchKernelGGL(k_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_bits, d_rank, (int)n, (int)words);
    }
    if (m > 0)
        hipLaunchKernelGGL(k_codeg, dim3((unsigned)std::min<int64_t>(8192, (m + 3) / 4)), dim3(256), 0, 0, d_ii, d_jj, d_bits, d_codeg, m, (int)words);
    DESC_HIP(hipGetLastError());
    s->codeg.assign((size_t)m, 0);
    if (m) DESC_HIP(hipMemcpy(s->codeg.data(), d_codeg, sizeof(int32_t) * m, hipMemcpyDeviceToHost));

    // edges with cycles, median, n_sample, cum_ind  (DESC_PGD.m:36-51) -- O(m) on the host
    s->pos_edge.clear();
    std::vector<int32_t> pos_cd;
    int32_t max_codeg = 0;
    for (int64_t e = 0; e < m; ++e) if (s->codeg[e] > 0) { s->pos_edge.push_back((int32_t)e); pos_cd.push_back(s->codeg[e]); max_codeg = std::max(max_codeg, s->codeg[e]); }
    s->m_pos = (int64_t)s->pos_edge.size();
    int32_t n_sample = n_sample_min;
    if (s->m_pos > 0) {
        const size_t h = pos_cd.size() / 2;
        std::nth_element(pos_cd.begin(), pos_cd.begin() + h, pos_cd.end());
        double med = pos_cd[h];
        if ((pos_cd.size() & 1) == 0) med = 0.5 * ((double)*std::max_element(pos_cd.begin(), pos_cd.begin() + h) + med);
        n_sample = std::max(n_sample_min, (int32_t)std::ceil(med / 4.0));
    }
    s->n_sample = n_sample;
    s->cum_ind.assign((size_t)s->m_pos + 1, 0);
    s->max_cnt = 0;
    for (int64_t l = 0; l < s->m_pos; ++l) {
        const int32_t cnt = std::min(s->codeg[s->pos_edge[l]], n_sample);
        s->cum_ind[l + 1] = s->cum_ind[l] + cnt;
        s->max_cnt = std::max(s->max_cnt, cnt);
    }
    s->m_cycle = s->cum_ind[s->m_pos];
    if (s->m_cycle >= (1ll << 31) - 1) return fail(DESC_ERR_TOO_LARGE, "m_cycle = %lld exceeds 2^31-2", (long long)s->m_cycle);
    if (max_codeg > MAX_CODEG_LDS)
        return fail(DESC_ERR_TOO_LARGE, "an edge has %d common neighbours (> %d): use DESC_BUILD_HOST", max_codeg, MAX_CODEG_LDS);
    const int64_t mp = s->m_pos, mc = s->m_cycle;
    s->k.clear(); s->e_jk.clear(); s->e_ki.clear(); s->ikj.clear(); s->jki.clear();
    s->host_cycles = (mp == 0);                       // per-cycle arrays stay in HBM until somebody asks for them
    if (mp > 0) {
        std::vector<int32_t> cum32((size_t)mp + 1), pos_of_edge((size_t)m, -1);
        for (int64_t l = 0; l <= mp; ++l) cum32[l] = (int32_t)s->cum_ind[l];
        for (int64_t l = 0; l < mp; ++l) pos_of_edge[s->pos_edge[l]] = (int32_t)l;
        int32_t *d_pos, *d_cum, *d_poe, *d_k, *d_ejk, *d_eki, *d_ikj, *d_jki;
        if ((rc = D.alloc(&d_pos, mp)) || (rc = D.alloc(&d_cum, mp + 1)) || (rc = D.alloc(&d_poe, m))) return rc;
        // the five per-cycle arrays outlive this call: owned by the structure object
        s->dev = device;
        DESC_HIP(hipMalloc((void**)&s->d_k, sizeof(int32_t) * mc));
        DESC_HIP(hipMalloc((void**)&s->d_ejk, sizeof(int32_t) * mc));
        DESC_HIP(hipMalloc((void**)&s->d_eki, sizeof(int32_t) * mc));
        DESC_HIP(hipMalloc((void**)&s->d_ikj, sizeof(int32_t) * mc));
        DESC_HIP(hipMalloc((void**)&s->d_jki, sizeof(int32_t) * mc));
        d_k = s->d_k; d_ejk = s->d_ejk; d_eki = s->d_eki; d_ikj = s->d_ikj; d_jki = s->d_jki;
        DESC_HIP(hipMemcpy(d_pos, s->pos_edge.data(), sizeof(int32_t) * mp, hipMemcpyHostToDevice));
        DESC_HIP(hipMemcpy(d_cum, cum32.data(), sizeof(int32_t) * (mp + 1), hipMemcpyHostToDevice));
        DESC_HIP(hipMemcpy(d_poe, pos_of_edge.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
        // LDS per wave: keys (8 B) + 3 int32 per staged common neighbour
        int cap = 64;
        while (cap < max_codeg) cap <<= 1;
        const size_t lds = (size_t)4 * cap * (8 + 12);
        if (lds > 64 * 1024)
            DESC_HIP(hipFuncSetAttribute((const void*)k_fill_cycles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const unsigned g = (unsigned)std::min<int64_t>(8192, (mp + 3) / 4);
        hipLaunchKernelGGL(k_fill_cycles, dim3(g), dim3(256), lds, 0, d_pos, d_cum, d_ii, d_jj, d_bits, d_rank, d_rowptr, d_adj_eid,
                           d_k, d_ejk, d_eki, mp, (int)words, (int)n_sample, seed, cap);
        hipLaunchKernelGGL(k_mirror, dim3(g), dim3(256), 0, 0, d_pos, d_cum, d_poe, d_ii, d_jj, d_k, d_ejk, d_eki, d_ikj, d_jki, mp);
        DESC_HIP(hipGetLastError());
        DESC_HIP(hipDeviceSynchronize());
    }
    s->ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DESC_OK;
}

int structure_ensure_host(desc_structure* s) {
    if (!s || s->host_cycles) return DESC_OK;
    const int64_t mc = s->m_cycle;
    DESC_HIP(hipSetDevice(s->dev));
    s->k.resize((size_t)mc); s->e_jk.resize((size_t)mc); s->e_ki.resize((size_t)mc); s->ikj.resize((size_t)mc); s->jki.resize((size_t)mc);
    DESC_HIP(hipMemcpy(s->k.data(), s->d_k, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->e_jk.data(), s->d_ejk, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->e_ki.data(), s->d_eki, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->ikj.data(), s->d_ikj, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->jki.data(), s->d_jki, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    s->host_cycles = true;
    return DESC_OK;
}

void structure_free_device(desc_structure* s) {
    if (!s || s->dev < 0) return;
    (void)hipSetDevice(s->dev);
    for (int32_t* q : {s->d_k, s->d_ejk, s->d_eki, s->d_ikj, s->d_jki}) if (q) (void)hipFree(q);
    s->d_k = s->d_ejk = s->d_eki = s->d_ikj = s->d_jki = nullptr;
    s->dev = -1;
}

}  // namespace desc
