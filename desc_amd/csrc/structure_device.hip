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
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <thread>
#include <type_traits>
#include <vector>

#include "device_utils.h"

namespace desc {
namespace {

constexpr int MAX_CODEG_LDS = 1024;      // common neighbours staged per edge when sampling

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
// The same bitmaps straight from the edge list (no CSR needed first): two atomic ORs per edge into the L2-resident rows
__global__ void k_bitmaps_edges(const int32_t* ind_i, const int32_t* ind_j, unsigned long long* bits, int64_t m, int words) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = ind_i[e], j = ind_j[e];
        atomicOr(&bits[(size_t)i * words + (j >> 6)], 1ull << (j & 63));
        atomicOr(&bits[(size_t)j * words + (i >> 6)], 1ull << (i & 63));
    }
}
// degree and number of smaller neighbours of every node, from the bitmaps and their rank table
__global__ void k_degrees(const unsigned long long* bits, const uint32_t* rank, int32_t* deg, int32_t* low, int n, int words) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const size_t last = (size_t)v * words + (words - 1), own = (size_t)v * words + (v >> 6);
    deg[v] = (int32_t)(rank[last] + (uint32_t)__popcll(bits[last]));
    low[v] = (int32_t)(rank[own] + (uint32_t)__popcll(bits[own] & ((1ull << (v & 63)) - 1ull)));
}
// rowptr = exclusive scan of deg (n + 1 entries), upstart = exclusive scan of deg - low = id of the first edge (v, .) in the
// (i,j)-sorted edge list.  One workgroup: n is the number of nodes.
__global__ __launch_bounds__(1024) void k_scan_rows(const int32_t* deg, const int32_t* low, int32_t* rowptr, int32_t* upstart, int n) {
    __shared__ int sa[1024], sb[1024];
    __shared__ int carry_a, carry_b;
    if (threadIdx.x == 0) { carry_a = 0; carry_b = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int v = base + threadIdx.x;
        const int a = v < n ? deg[v] : 0, b = v < n ? deg[v] - low[v] : 0;
        sa[threadIdx.x] = a; sb[threadIdx.x] = b;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int xa = threadIdx.x >= d ? sa[threadIdx.x - d] : 0, xb = threadIdx.x >= d ? sb[threadIdx.x - d] : 0;
            __syncthreads();
            sa[threadIdx.x] += xa; sb[threadIdx.x] += xb;
            __syncthreads();
        }
        if (v < n) { rowptr[v] = carry_a + sa[threadIdx.x] - a; upstart[v] = carry_b + sb[threadIdx.x] - b; }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_a += sa[1023]; carry_b += sb[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) rowptr[n] = carry_a;
}
// CSR adjacency (neighbours ascending) and the edge id of every slot, from the bitmaps: 16 lanes per node row, lanes over its
// words.  Slot of neighbour u in row v = rowptr[v] + rank[v][word] + (bits below u in that word); the edge {v,u} is the
// (position of the larger endpoint among the smaller one's larger neighbours)-th edge of the smaller endpoint: Ind is sorted by (i,j).
__global__ __launch_bounds__(256) void k_csr_from_bits(const unsigned long long* bits, const uint32_t* rank, const int32_t* rowptr, const int32_t* low,
                                                       const int32_t* upstart, int32_t* adj, int32_t* adj_eid, int n, int words) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int v = row0; v < n; v += nrows) {
        const int r0 = rowptr[v], lv = low[v], us = upstart[v];
        for (int w = l16; w < words; w += 16) {
            unsigned long long x = bits[(size_t)v * words + w];
            int pos = r0 + (int)rank[(size_t)v * words + w];
            while (x) {
                const int u = w * 64 + __ffsll((long long)x) - 1;
                int e;
                if (u > v) e = us + (pos - r0 - lv);
                else {
                    const size_t wu = (size_t)u * words + (v >> 6);
                    const int idx = (int)(rank[wu] + (uint32_t)__popcll(bits[wu] & ((1ull << (v & 63)) - 1ull)));     // position of v in row u
                    e = upstart[u] + (idx - low[u]);
                }
                adj[pos] = u; adj_eid[pos] = e;
                ++pos;
                x &= x - 1;
            }
        }
    }
}
// rank[v][w] = number of neighbours of v in words < w
__global__ void k_rank(const unsigned long long* bits, uint32_t* rank, int n, int words) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t acc = 0;
    for (int w = 0; w < words; ++w) { rank[(size_t)v * words + w] = acc; acc += (uint32_t)__popcll(bits[(size_t)v * words + w]); }
}
// codegree of every edge: popcount(row_i & row_j), one wave per edge; plus the histogram of the
// codegrees (median and maximum on the host in O(n)): per-workgroup bins in LDS when they fit
constexpr int HIST_LDS_BINS = 8192;
__global__ __launch_bounds__(256) void k_codeg(const int32_t* ind_i, const int32_t* ind_j, const unsigned long long* bits,
                                               int32_t* codeg, int32_t* hist, int64_t m, int words, int nbins) {
    __shared__ int32_t lh[HIST_LDS_BINS];
    const bool local = nbins <= HIST_LDS_BINS;
    if (local) { for (int t = threadIdx.x; t < nbins; t += 256) lh[t] = 0; __syncthreads(); }
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t e = wid; e < m; e += nw) {
        const unsigned long long* a = bits + (size_t)ind_i[e] * words;
        const unsigned long long* b = bits + (size_t)ind_j[e] * words;
        int c = 0;
        for (int w = lane; w < words; w += 64) c += __popcll(a[w] & b[w]);
        c = wave_incl_scan(c, lane);
        if (lane == 63) { codeg[e] = c; atomicAdd(local ? &lh[c] : &hist[c], 1); }
    }
    if (local) {
        __syncthreads();
        for (int t = threadIdx.x; t < nbins; t += 256) if (lh[t]) atomicAdd(&hist[t], lh[t]);
    }
}

// Cycle lists (DESC_PGD.m:79-96): one wave per edge-with-cycles.  Writes the sampled third
// vertices (ascending) and the edge's selection threshold, indexed by edge id: a common neighbour k
// of edge e is kept iff (key(e,k), k) <= (tau[e], ktau[e]) lexicographically (all ones when
// codeg < n_sample).
__global__ __launch_bounds__(256) void k_fill_cycles(const int32_t* pos_edge, const int32_t* cum, const int32_t* ind_i,
                                                     const int32_t* ind_j, const unsigned long long* bits, int32_t* kk,
                                                     unsigned long long* tau, int32_t* ktau, int64_t m_pos, int words,
                                                     int n_sample, uint64_t seed, int lds_cap, int exact_only) {
    extern __shared__ unsigned long long smem[];      // per wave: keys[lds_cap] then k[lds_cap] as int32
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long* keys = smem + (size_t)wv * lds_cap;
    int32_t* ks = (int32_t*)(smem + (size_t)4 * lds_cap) + (size_t)wv * lds_cap;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        const int base = cum[l];
        const unsigned long long* a = bits + (size_t)i * words;
        const unsigned long long* b = bits + (size_t)j * words;
        // enumerate the common neighbours in ascending order; `run` = how many precede this word
        int run = 0;
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int w = w0 + lane;
            unsigned long long x = 0;
            if (w < words) x = a[w] & b[w];
            const int pc = __popcll(x);
            const int incl = wave_incl_scan(pc, lane);
            int pos = run + incl - pc;
            run += __shfl(incl, 63, 64);
            while (x) {
                const int k = w * 64 + __ffsll((long long)x) - 1;
                if (pos < lds_cap) ks[pos] = k;
                ++pos;
                x &= x - 1;
            }
        }
        const int cd = run;                           // codegree
        __builtin_amdgcn_wave_barrier();
        // the keys in a pass of their own over the compact list: every lane busy (in the loop above a lane holds 0 ... 6 neighbours and the wave
        // waits for the fullest one -- the 64-bit mixing is most of this kernel's instructions)
        unsigned long long kr[4] = {~0ull, ~0ull, ~0ull, ~0ull};      // this lane's keys of the first 256 neighbours: the probes below count them in registers
        bool kon[4] = {false, false, false, false};
        if (cd >= n_sample) {
            const int cdl = min(cd, lds_cap);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = lane + 64 * u;
                if (t < cdl) { kr[u] = d_sample_key(seed, (uint64_t)e, (uint64_t)ks[t]); keys[t] = kr[u]; kon[u] = true; }
            }
            for (int t = lane + 256; t < cdl; t += 64) keys[t] = d_sample_key(seed, (uint64_t)e, (uint64_t)ks[t]);
            __builtin_amdgcn_wave_barrier();
        }
        if (cd < n_sample) {                          // DESC_PGD.m:83 samples iff codeg >= n_sample
            for (int t = lane; t < cd; t += 64) kk[base + t] = ks[t];
            if (lane == 0) { tau[e] = ~0ull; ktau[e] = 0x7FFFFFFF; }
        } else {
            // Keep the n_sample smallest keys.  A separator g with exactly n_sample keys <= g is found
            // by bisection on the key VALUE (keys are uniform 64-bit hashes: the first probe is the
            // expected quantile, ~log2(cd) probes follow; one probe = one ballot pass over the keys).
            // Round 4: the next probe is interpolated between the bracket's ends from their counts (the keys are uniform: 2-4 probes instead of the
            // ~10 of plain halving, whose second probe already sat half-way to the far end of the key range); every third probe halves, so the
            // bracket shrinks geometrically whatever the keys look like.  Any separator with exactly n_sample keys <= g selects the same set.
            unsigned long long lo = 0, hi = ~0ull, g = ~0ull;
            int c_lo = 0, c_hi = cd;                             // keys <= lo (none known below the first probe), keys <= hi
            bool have_lo = false, found = (cd == n_sample) && !exact_only;
            if (!found) g = (unsigned long long)(((double)n_sample + 0.5) / (double)cd * 18446744073709549568.0);
            for (int it = 0; !found && !exact_only && it < 96; ++it) {
                int c = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) c += __popcll(__ballot(kon[u] && kr[u] <= g));
                for (int t0 = 256; t0 < cd; t0 += 64) { const int t = t0 + lane; c += __popcll(__ballot(t < cd && keys[t] <= g)); }
                if (c == n_sample) { found = true; break; }
                if (c < n_sample) { lo = g; c_lo = c; have_lo = true; } else { hi = g; c_hi = c; }
                const unsigned long long b = have_lo ? lo : 0ull;
                const unsigned long long span = hi - b;
                if (span <= 1ull) break;                         // no value left in between: duplicate keys straddle the cut
                unsigned long long step = span / 2ull;
                if (it % 3 != 2) {
                    const double frac = ((double)(n_sample - c_lo) + 0.5) / (double)(c_hi - c_lo + 1);
                    step = (unsigned long long)((double)span * frac);
                    step = step < 1ull ? 1ull : (step > span - 1ull ? span - 1ull : step);
                }
                g = b + step;
            }
            if (found) {
                int outbase = 0;
                for (int t0 = 0; t0 < cd; t0 += 64) {
                    const int t = t0 + lane;
                    const bool sel = t < cd && keys[t] <= g;
                    const unsigned long long mk = __ballot(sel);
                    if (sel) kk[base + outbase + __popcll(mk & ((1ull << lane) - 1ull))] = ks[t];
                    outbase += __popcll(mk);
                }
                if (lane == 0) { tau[e] = g; ktau[e] = 0x7FFFFFFF; }
            } else {
                // duplicate keys at the cut (probability ~1e-14 per edge): exact ranking of (key, k) pairs;
                // common neighbours are distinct, positions ascend with k
                int outbase = 0;
                for (int t0 = 0; t0 < cd; t0 += 64) {
                    const int t = t0 + lane;
                    bool sel = false;
                    if (t < cd) {
                        const unsigned long long kt = keys[t];
                        int rk = 0;
                        for (int u = 0; u < cd; ++u) { const unsigned long long ku = keys[u]; rk += (ku < kt) || (ku == kt && u < t); }
                        sel = rk < n_sample;
                        if (rk == n_sample - 1) { tau[e] = kt; ktau[e] = ks[t]; }     // the last one kept
                    }
                    const unsigned long long mk = __ballot(sel);
                    if (sel) kk[base + outbase + __popcll(mk & ((1ull << lane) - 1ull))] = ks[t];
                    outbase += __popcll(mk);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// CEMP's sampler (CEMP.m:57-65): nsample draws WITH replacement per edge-with-cycles,
// draw t = CoInd[key(seed, e, t) mod codeg]; one wave per edge.
__global__ __launch_bounds__(256) void k_cemp_samples(const int32_t* pos_edge, const int32_t* ind_i, const int32_t* ind_j,
                                                      const unsigned long long* bits, const uint32_t* rank, const int32_t* rowptr,
                                                      const int32_t* adj_eid, int32_t* kk, int32_t* e_jk, int32_t* e_ki, uint32_t* pk, int64_t m_pos,
                                                      int words, int nsample, uint64_t seed, int lds_cap) {
    extern __shared__ unsigned long long smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int32_t* ks = (int32_t*)smem + (size_t)wv * lds_cap;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        const unsigned long long* a = bits + (size_t)i * words;
        const unsigned long long* b = bits + (size_t)j * words;
        int run = 0;
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int w = w0 + lane;
            unsigned long long x = 0;
            if (w < words) x = a[w] & b[w];
            const int pc = __popcll(x);
            const int incl = wave_incl_scan(pc, lane);
            int pos = run + incl - pc;
            run += __shfl(incl, 63, 64);
            while (x) {
                if (pos < lds_cap) ks[pos] = w * 64 + __ffsll((long long)x) - 1;
                ++pos;
                x &= x - 1;
            }
        }
        const int cd = run;
        __builtin_amdgcn_wave_barrier();
        const int r0i = rowptr[i], r0j = rowptr[j];
        for (int t = lane; t < nsample; t += 64) {
            const int k = ks[d_sample_key(seed, (uint64_t)e, (uint64_t)t) % (uint64_t)cd];
            const size_t wi = (size_t)i * words + (k >> 6), wj = (size_t)j * words + (k >> 6);
            const unsigned long long below = (1ull << (k & 63)) - 1ull;
            const int64_t c = l * nsample + t;
            const int xi = (int)(rank[wi] + __popcll(bits[wi] & below)), xj = (int)(rank[wj] + __popcll(bits[wj] & below));     // idx_i(k), idx_j(k)
            kk[c] = k;
            e_ki[c] = adj_eid[r0i + xi];
            e_jk[c] = adj_eid[r0j + xj];
            if (pk) pk[c] = (uint32_t)xi | (uint32_t)xj << 16;          // S({k,i}) / S({j,k}) inside the CSR-aligned rows i and j (csrc/cemp.hip)
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// e_jk / e_ki of every cycle (DESC_PGD.m:87-88) from the sampled k: binary search of k in the CSR
// rows of j and i.  Only run when the full index structure is exported (desc_structure_get) or
// the gather layout needs it.
__global__ __launch_bounds__(256) void k_cycle_edges(const int32_t* pos_edge, const int32_t* cum, const int32_t* ind_i,
                                                     const int32_t* ind_j, const int32_t* kk, const int32_t* rowptr,
                                                     const int32_t* adj, const int32_t* adj_eid, int32_t* e_jk, int32_t* e_ki,
                                                     int64_t m_pos) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        const int ri = rowptr[i], di = rowptr[i + 1] - ri, rj = rowptr[j], dj = rowptr[j + 1] - rj;
        for (int c = cum[l] + lane; c < cum[l + 1]; c += 64) {
            const int k = kk[c];
            int lo = 0, hi = di;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[ri + mid] < k) lo = mid + 1; else hi = mid; }
            e_ki[c] = adj_eid[ri + lo];
            lo = 0; hi = dj;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[rj + mid] < k) lo = mid + 1; else hi = mid; }
            e_jk[c] = adj_eid[rj + lo];
        }
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

__global__ void k_iota(int32_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (int32_t)i;
}

struct DevBuf {
    hvec<void*> p;
    ~DevBuf() { release(); }
    void release() { for (void* q : p) dev_free(q); p.clear(); }
    template <class T> int alloc(T** out, size_t count) {
        void* q = nullptr;
        DESC_HIP(dev_alloc(&q, sizeof(T) * (count ? count : 1)));
        p.push_back(q); *out = (T*)q;
        return DESC_OK;
    }
};

// Compaction of the edges with cycles (DESC_PGD.m:36-37) and the prefix sums of their sampled cycle counts min(codeg, n_sample) (:45-49), as three
// launches: per tile of 1024 edges the number of edges with cycles and of their cycles (k_tile_sums), an exclusive scan of the tile totals by one
// workgroup (k_scan_tiles), and the scatter with the in-tile scan redone in the LDS (k_compact_tiles).  (The first form used hipCUB's device scans:
// 2.4 ms of host time per build in their size queries and launches, profiles/r04_e2e_laps_c4.txt "compaction (launched)".)
constexpr int SCAN_TILE = 1024;
__device__ __forceinline__ void tile_counts(const int32_t* codeg, int n_sample, int64_t m, int64_t e0, int f[4], int c[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int32_t cd = e0 + u < m ? codeg[e0 + u] : 0;
        f[u] = cd > 0 ? 1 : 0;
        c[u] = cd > 0 ? min(cd, n_sample) : 0;
    }
}
__global__ __launch_bounds__(256) void k_tile_sums(const int32_t* codeg, int n_sample, int64_t m, int32_t* tile_f, long long* tile_c) {
    __shared__ int sf[4], sc[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int f[4], c[4];
    tile_counts(codeg, n_sample, m, (int64_t)blockIdx.x * SCAN_TILE + 4 * threadIdx.x, f, c);
    int tf = f[0] + f[1] + f[2] + f[3], tc = c[0] + c[1] + c[2] + c[3];
    tf = wave_incl_scan(tf, lane); tc = wave_incl_scan(tc, lane);
    if (lane == 63) { sf[wv] = tf; sc[wv] = tc; }
    __syncthreads();
    if (threadIdx.x == 0) { tile_f[blockIdx.x] = sf[0] + sf[1] + sf[2] + sf[3]; tile_c[blockIdx.x] = (long long)sc[0] + sc[1] + sc[2] + sc[3]; }
}
// exclusive scan of the tile totals, in place, by ONE workgroup of 1024 threads (a few thousand tiles)
__global__ __launch_bounds__(1024) void k_scan_tiles(int32_t* tile_f, long long* tile_c, int nt) {
    __shared__ long long sa[1024], sb[1024];
    __shared__ long long carry_f, carry_c;
    if (threadIdx.x == 0) { carry_f = 0; carry_c = 0; }
    __syncthreads();
    for (int base = 0; base < nt; base += 1024) {
        const int t = base + threadIdx.x;
        const long long vf = t < nt ? tile_f[t] : 0, vc = t < nt ? tile_c[t] : 0;
        sa[threadIdx.x] = vf; sb[threadIdx.x] = vc;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const long long xf = threadIdx.x >= d ? sa[threadIdx.x - d] : 0, xc = threadIdx.x >= d ? sb[threadIdx.x - d] : 0;
            __syncthreads();
            sa[threadIdx.x] += xf; sb[threadIdx.x] += xc;
            __syncthreads();
        }
        if (t < nt) { tile_f[t] = (int32_t)(carry_f + sa[threadIdx.x] - vf); tile_c[t] = carry_c + sb[threadIdx.x] - vc; }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_f += sa[1023]; carry_c += sb[1023]; }
        __syncthreads();
    }
}
// pos[l] = e, poe[e] = l (-1: no cycles), cum[l] = cycles before edge-with-cycles l (int32 for the kernels, int64 for the host's cum_ind)
__global__ __launch_bounds__(256) void k_compact_tiles(const int32_t* codeg, int n_sample, const int32_t* tile_f, const long long* tile_c, int32_t* pos, int32_t* poe,
                                                       int32_t* cum, long long* cumc, int64_t m, int64_t mp) {
    __shared__ int sf[4], sc[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t e0 = (int64_t)blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
    int f[4], c[4];
    tile_counts(codeg, n_sample, m, e0, f, c);
    const int tf = f[0] + f[1] + f[2] + f[3], tc = c[0] + c[1] + c[2] + c[3];
    const int inf = wave_incl_scan(tf, lane), inc = wave_incl_scan(tc, lane);
    if (lane == 63) { sf[wv] = inf; sc[wv] = inc; }
    __syncthreads();
    int l = tile_f[blockIdx.x] + inf - tf;
    long long cy = tile_c[blockIdx.x] + inc - tc;
    for (int w = 0; w < wv; ++w) { l += sf[w]; cy += sc[w]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t e = e0 + u;
        if (e >= m) break;
        if (!f[u]) { poe[e] = -1; continue; }
        pos[l] = (int32_t)e; poe[e] = l;
        cum[l] = (int32_t)cy; cumc[l] = cy;
        if (l == mp - 1) { cum[mp] = (int32_t)(cy + c[u]); cumc[mp] = cy + c[u]; }
        ++l; cy += c[u];
    }
}

}  // namespace

int build_structure_device(const desc_problem* prob, int32_t n_sample_min, uint64_t seed, int32_t device, desc_structure* s) {
    auto t0 = std::chrono::steady_clock::now();
    auto t_lap = t0;
    const char* tenv = getenv("DESC_DEBUG_TIMING");
    const bool timing = tenv && atoi(tenv) != 0;
    auto lap = [&](const char* what, bool wait = true) {
        if (!timing) return;
        if (wait) (void)hipDeviceSynchronize();
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[desc_amd] structure_device %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    const int64_t n = prob->n, m = prob->m;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DESC_ERR_HIP, "no HIP device visible for DESC_BUILD_DEVICE");
    if (device < 0 || device >= ndev) return fail(DESC_ERR_INVALID, "device %d out of range", device);
    const int64_t words = (n + 63) / 64;
    if ((double)n * (double)words * 12.0 > 64.0 * 1073741824.0)
        return fail(DESC_ERR_TOO_LARGE, "adjacency bitmaps of n = %lld nodes do not fit the device-build budget; use DESC_BUILD_HOST", (long long)n);
    DESC_HIP(hipSetDevice(device));
    s->n = n; s->m = m; s->dev = device; s->seed = seed;

    // device arrays that outlive this call are owned by the structure object (structure_free_device)
    auto keep = [&](auto** out, size_t count) -> int {
        void* q = nullptr;
        DESC_HIP(dev_alloc(&q, sizeof(**out) * (count ? count : 1)));
        *out = (std::remove_reference_t<decltype(*out)>)q;
        return DESC_OK;
    };
    DevBuf D;
    int rc;
    int32_t *d_codeg, *d_hist, *d_deg, *d_low, *d_upstart;
    unsigned long long* d_bits;
    s->words = (int32_t)words;
    if ((rc = keep(&s->d_rowptr, n + 1)) || (rc = keep(&s->d_adj, 2 * m)) || (rc = keep(&s->d_adj_eid, 2 * m)) ||
        (rc = keep(&s->d_ii, m)) || (rc = keep(&s->d_jj, m)) || (rc = D.alloc(&d_codeg, m)) || (rc = D.alloc(&d_hist, n + 1)) ||
        (rc = D.alloc(&d_deg, n)) || (rc = D.alloc(&d_low, n)) || (rc = D.alloc(&d_upstart, n)) ||
        (rc = keep(&s->d_bits, (size_t)n * words)) || (rc = keep(&s->d_rank, (size_t)n * words))) return rc;
    d_bits = s->d_bits;
    lap("alloc");
    // The CSR index of the graph is made on the device (round 3; the host pass over the edges and the upload of its 2 x 2m ints
    // were 7-9 ms of a 200 ms solve at C4): bitmaps from the edge list -> rank table -> degrees -> row starts -> adjacency and edge ids.
    if (m) {
        DESC_HIP(hipMemcpy(s->d_ii, prob->ind_i, sizeof(int32_t) * m, hipMemcpyHostToDevice));
        DESC_HIP(hipMemcpy(s->d_jj, prob->ind_j, sizeof(int32_t) * m, hipMemcpyHostToDevice));
    }
    g_ind_upload_count.fetch_add(1);
    lap("upload Ind");
    DESC_HIP(hipMemsetAsync(d_bits, 0, sizeof(unsigned long long) * (size_t)n * words, 0));
    DESC_HIP(hipMemsetAsync(d_hist, 0, sizeof(int32_t) * (n + 1), 0));
    hvec<int32_t> rowptr((size_t)n + 1, 0);
    if (n > 0) {
        if (m) hipLaunchKernelGGL(k_bitmaps_edges, dim3((unsigned)std::min<int64_t>(4096, (m + 255) / 256)), dim3(256), 0, 0, s->d_ii, s->d_jj, d_bits, m, (int)words);
        hipLaunchKernelGGL(k_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_bits, s->d_rank, (int)n, (int)words);
        hipLaunchKernelGGL(k_degrees, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_bits, s->d_rank, d_deg, d_low, (int)n, (int)words);
        hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, 0, d_deg, d_low, s->d_rowptr, d_upstart, (int)n);
        hipLaunchKernelGGL(k_csr_from_bits, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (n * 16 + 255) / 256))), dim3(256), 0, 0,
                           d_bits, s->d_rank, s->d_rowptr, d_low, d_upstart, s->d_adj, s->d_adj_eid, (int)n, (int)words);
        DESC_HIP(hipGetLastError());
        DESC_HIP(hipMemcpy(rowptr.data(), s->d_rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
    }
    if (rowptr[n] != 2 * m) return fail(DESC_ERR_INVALID, "Ind lists an edge twice (the adjacency holds %lld of %lld entries)", (long long)rowptr[n], (long long)(2 * m));
    s->max_deg = 0;
    for (int64_t v = 0; v < n; ++v) s->max_deg = std::max(s->max_deg, rowptr[v + 1] - rowptr[v]);
    s->rowptr_host = std::move(rowptr);               // the solver's host-side plan needs the row starts again
    lap("device csr");
    if (m > 0)      // a codegree is at most the smaller degree: max_deg + 1 bins (in LDS up to 8192; n + 1 would push n > 8191 onto global atomics -- 30 ms at n = 10000)
        hipLaunchKernelGGL(k_codeg, dim3((unsigned)std::min<int64_t>(2048, (m + 3) / 4)), dim3(256), 0, 0, s->d_ii, s->d_jj, d_bits, d_codeg, d_hist, m, (int)words,
                           (int)std::min<int64_t>(n + 1, (int64_t)s->max_deg + 1));
    DESC_HIP(hipGetLastError());
    hvec<int32_t> hist((size_t)n + 1, 0);
    DESC_HIP(hipMemcpy(hist.data(), d_hist, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));      // synchronous: every kernel above has finished
    lap("bitmaps+codeg+hist");

    // edges with cycles, median, n_sample, m_cycle  (DESC_PGD.m:36-51) -- all from the codegree histogram (codeg <= n-2): O(n) on the host
    // instead of a selection / a pass over m values.
    int64_t mp = 0, mc_total = 0;
    int32_t max_codeg = 0;
    for (int64_t c = 1; c <= n; ++c) if (hist[c]) { mp += hist[c]; max_codeg = (int32_t)c; }
    s->m_pos = mp;
    int32_t n_sample = n_sample_min;
    if (mp > 0) {
        auto kth = [&](int64_t r) {        // r-th smallest positive codegree, r = 0-based
            int64_t acc = 0;
            for (int64_t c = 1; c <= n; ++c) { acc += hist[c]; if (acc > r) return (double)c; }
            return (double)max_codeg;
        };
        const double med = (mp & 1) ? kth(mp / 2) : 0.5 * (kth(mp / 2 - 1) + kth(mp / 2));     // MATLAB median (:43)
        n_sample = std::max(n_sample_min, (int32_t)std::ceil(med / 4.0));
    }
    for (int64_t c = 1; c <= n; ++c) mc_total += (int64_t)hist[c] * std::min<int64_t>(c, n_sample);      // sum of min(codeg, n_sample)  (:45-51)
    s->n_sample = n_sample;
    s->max_cnt = std::min(max_codeg, n_sample);
    s->m_cycle = mc_total;
    if (s->m_cycle >= (1ll << 31) - 1) return fail(DESC_ERR_TOO_LARGE, "m_cycle = %lld exceeds 2^31-2", (long long)s->m_cycle);
    if (max_codeg > MAX_CODEG_LDS)
        return fail(DESC_ERR_TOO_LARGE, "an edge has %d common neighbours (> %d): use DESC_BUILD_HOST", max_codeg, MAX_CODEG_LDS);
    lap("host median");
    const int64_t mc = s->m_cycle;
    s->k.clear(); s->e_jk.clear(); s->e_ki.clear(); s->ikj.clear(); s->jki.clear();
    s->host_cycles = (mp == 0);                       // per-cycle arrays stay in HBM until somebody asks for them
    // (the host's copies of the per-edge tables are sized AFTER the kernels below are on their way: 40 MB of first touches at C4, ~2 ms the device would idle through)
    if (mp == 0) { s->codeg.assign((size_t)m, 0); s->pos_edge.clear(); s->cum_ind.assign(1, 0); }
    if (mp > 0) {
        // Round 4: the compaction of the edges with cycles and the prefix sums of their cycle counts (DESC_PGD.m:36-37, 45-54) are a tiled device
        // scan + scatter (three launches) on a stream of their own, the cycle-sampling kernel follows on the same stream at once, and the
        // host copies pos_edge / cum_ind / codeg down for its planning WHILE that kernel runs.  (Round 3 did the compaction on the host: 4.2 ms
        // at C4 with the device idle, the three tables went back up, and only then was the sampling kernel launched.)
        hipStream_t fs = nullptr;
        DESC_HIP(stream_acquire(&fs));
        s->fill_stream = (void*)fs;
        int32_t* d_tile_f = nullptr; long long *d_tile_c = nullptr, *d_cumc = nullptr;
        auto tmp = [&](auto** out, size_t count) -> int {
            void* q = nullptr;
            DESC_HIP(dev_alloc(&q, sizeof(**out) * (count ? count : 1)));
            s->d_build_blocks.push_back(q);
            *out = (std::remove_reference_t<decltype(*out)>)q;
            return DESC_OK;
        };
        const int nt = (int)((m + SCAN_TILE - 1) / SCAN_TILE);
        if ((rc = keep(&s->d_pos, mp)) || (rc = keep(&s->d_cum, mp + 1)) || (rc = keep(&s->d_poe, m)) || (rc = keep(&s->d_k, mc)) ||
            (rc = keep(&s->d_tau, m)) || (rc = keep(&s->d_ktau, m)) || (rc = tmp(&d_tile_f, nt)) || (rc = tmp(&d_tile_c, nt)) || (rc = tmp(&d_cumc, mp + 1))) return rc;
        hipLaunchKernelGGL(k_tile_sums, dim3(nt), dim3(256), 0, fs, d_codeg, (int)n_sample, m, d_tile_f, d_tile_c);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, fs, d_tile_f, d_tile_c, nt);
        hipLaunchKernelGGL(k_compact_tiles, dim3(nt), dim3(256), 0, fs, d_codeg, (int)n_sample, d_tile_f, d_tile_c, s->d_pos, s->d_poe, s->d_cum, d_cumc, m, mp);
        DESC_HIP(hipGetLastError());
        hipEvent_t ev_c = nullptr;
        DESC_HIP(hipEventCreateWithFlags(&ev_c, hipEventDisableTiming));
        DESC_HIP(hipEventRecord(ev_c, fs));
        lap("compaction (launched)", false);
        // LDS per wave: key (8 B) + k (4 B) per staged common neighbour
        int cap = 64;
        while (cap < max_codeg) cap <<= 1;
        const size_t lds = (size_t)4 * cap * (8 + 4);
        if (lds > 64 * 1024)
            DESC_HIP(hipFuncSetAttribute((const void*)k_fill_cycles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const char* tenv2 = getenv("DESC_DEBUG_EXACT_SELECT");     // tests: force the exact (tie-safe) ranking path
        const unsigned g = (unsigned)std::min<int64_t>(8192, (mp + 3) / 4);
        hipLaunchKernelGGL(k_fill_cycles, dim3(g), dim3(256), lds, fs, s->d_pos, s->d_cum, s->d_ii, s->d_jj, d_bits, s->d_k, s->d_tau, s->d_ktau,
                           mp, (int)words, (int)n_sample, seed, cap, (tenv2 && atoi(tenv2) != 0) ? 1 : 0);
        DESC_HIP(hipGetLastError());
        // The sampled cycles (5.8 ms of kernel at C4) are not waited for: the caller goes on to plan the solver's layout on the host (9 ms at
        // C4) while they are drawn.  Whoever reads d_k / d_tau / d_ktau waits for ev_fill (setup_node, structure_ensure_host).
        // DESC_DEBUG_SYNC_FILL=1: wait here (A/B, timing laps).
        hipEvent_t ev = nullptr;
        DESC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        s->ev_fill = (void*)ev;
        DESC_HIP(hipEventRecord(ev, fs));
        // the host's copies of the per-edge tables, under the sampling kernel (the copies wait for the compaction only)
        s->d_codeg = d_codeg;                           // the codegrees go to the host only if somebody asks for them (structure_ensure_host)
        s->pos_edge.resize((size_t)mp);
        s->cum_ind.resize((size_t)mp + 1);
        {
            const hipError_t ew = hipEventSynchronize(ev_c);
            (void)hipEventDestroy(ev_c);
            if (ew != hipSuccess) return fail(DESC_ERR_HIP, "edge compaction kernels failed: %s", hipGetErrorString(ew));
        }
        DESC_HIP(hipMemcpy(s->pos_edge.data(), s->d_pos, sizeof(int32_t) * mp, hipMemcpyDeviceToHost));
        DESC_HIP(hipMemcpy(s->cum_ind.data(), d_cumc, sizeof(int64_t) * (mp + 1), hipMemcpyDeviceToHost));
        if (s->cum_ind[mp] != mc) return fail(DESC_ERR_STATE, "cycle prefix sums (%lld) disagree with the codegree histogram (%lld)", (long long)s->cum_ind[mp], (long long)mc);
        lap("tables to host", false);
        // the builder's own scratch (codegrees, histogram, degree tables) is still being read by the kernels in flight: the structure keeps it
        for (void* q : D.p) s->d_build_blocks.push_back(q);
        D.p.clear();
        const char* sf = getenv("DESC_DEBUG_SYNC_FILL");
        if (sf && atoi(sf) != 0) DESC_HIP(hipDeviceSynchronize());
        lap("fill (launched)", false);
    } else if (m) DESC_HIP(hipMemcpy(s->codeg.data(), d_codeg, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    s->ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DESC_OK;
}

int build_cemp_samples_device(const desc_device_problem* dp, int32_t nsample, uint64_t seed, int64_t* m_pos,
                              int32_t** o_pos, int32_t** o_k, int32_t** o_ejk, int32_t** o_eki, uint32_t** o_pk, int32_t* o_max_deg) {
    const int64_t n = dp->n, m = dp->m;
    const int64_t words = (n + 63) / 64;
    *m_pos = 0; *o_pos = *o_k = *o_ejk = *o_eki = nullptr;
    if (o_pk) *o_pk = nullptr;
    int32_t max_deg = 0;
    for (int64_t v = 0; v < n; ++v) max_deg = std::max(max_deg, dp->rowptr[v + 1] - dp->rowptr[v]);
    if (o_max_deg) *o_max_deg = max_deg;
    if ((double)n * (double)words * 12.0 > 64.0 * 1073741824.0) return fail(DESC_ERR_TOO_LARGE, "adjacency bitmaps do not fit the device budget");
    DESC_HIP(hipSetDevice(dp->device));
    DevBuf D;
    int rc;
    const int32_t *d_rowptr = dp->d_rowptr, *d_adj = dp->d_adj, *d_adj_eid = dp->d_adj_eid, *d_ii = dp->d_ii, *d_jj = dp->d_jj;   // CSR index: the device problem's
    int32_t *d_codeg, *d_hist;
    unsigned long long* d_bits; uint32_t* d_rank;
    if ((rc = D.alloc(&d_codeg, m)) || (rc = D.alloc(&d_hist, n + 1)) ||
        (rc = D.alloc(&d_bits, (size_t)n * words)) || (rc = D.alloc(&d_rank, (size_t)n * words))) return rc;
    DESC_HIP(hipMemset(d_bits, 0, sizeof(unsigned long long) * (size_t)n * words));
    DESC_HIP(hipMemset(d_hist, 0, sizeof(int32_t) * (n + 1)));
    if (n > 0) {
        hipLaunchKernelGGL(k_bitmaps, dim3((unsigned)std::min<int64_t>(n, 4096)), dim3(256), 0, 0, d_rowptr, d_adj, d_bits, (int)n, (int)words);
        hipLaunchKernelGGL(k_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_bits, d_rank, (int)n, (int)words);
    }
    if (m > 0)
        hipLaunchKernelGGL(k_codeg, dim3((unsigned)std::min<int64_t>(2048, (m + 3) / 4)), dim3(256), 0, 0, d_ii, d_jj, d_bits, d_codeg, d_hist, m, (int)words, (int)(n + 1));
    DESC_HIP(hipGetLastError());
    // edges with cycles: the codegree histogram says how many have none.  Usually (dense measurement graphs) every edge lies
    // on a triangle and pos_edge is the identity: filled on the device, no O(m) round trip through the host.
    hvec<int32_t> hist((size_t)n + 1), pos_edge;
    DESC_HIP(hipMemcpy(hist.data(), d_hist, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
    int32_t max_codeg = 0;
    for (int64_t c = 1; c <= n; ++c) if (hist[c]) max_codeg = (int32_t)c;
    const bool all_pos = m > 0 && hist[0] == 0;
    if (!all_pos) {
        hvec<int32_t> codeg((size_t)m);
        if (m) DESC_HIP(hipMemcpy(codeg.data(), d_codeg, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
        for (int64_t e = 0; e < m; ++e) if (codeg[e] > 0) pos_edge.push_back((int32_t)e);
    }
    const int64_t mp = all_pos ? m : (int64_t)pos_edge.size(), mc = mp * nsample;
    if (mc >= (1ll << 31)) return fail(DESC_ERR_TOO_LARGE, "m_pos * nsample exceeds 2^31");
    if (max_codeg > 4 * MAX_CODEG_LDS) return fail(DESC_ERR_TOO_LARGE, "an edge has %d common neighbours: host sampler", max_codeg);
    *m_pos = mp;
    if (mp == 0) return DESC_OK;
    auto keep = [&](int32_t** out, size_t count) -> int {
        void* q = nullptr;
        DESC_HIP(dev_alloc(&q, sizeof(int32_t) * (count ? count : 1)));
        *out = (int32_t*)q;
        return DESC_OK;
    };
    const bool want_pk = o_pk != nullptr && max_deg < 65536;            // 16 bits per row position
    if ((rc = keep(o_pos, mp)) || (rc = keep(o_k, mc)) || (rc = keep(o_ejk, mc)) || (rc = keep(o_eki, mc)) || (want_pk && (rc = keep((int32_t**)o_pk, mc)))) {
        for (int32_t** q : {o_pos, o_k, o_ejk, o_eki}) { if (*q) dev_free(*q); *q = nullptr; }
        if (o_pk && *o_pk) { dev_free(*o_pk); *o_pk = nullptr; }
        return rc;
    }
    if (all_pos) hipLaunchKernelGGL(k_iota, dim3((unsigned)std::min<int64_t>(2048, (mp + 255) / 256)), dim3(256), 0, 0, *o_pos, mp);
    else DESC_HIP(hipMemcpy(*o_pos, pos_edge.data(), sizeof(int32_t) * mp, hipMemcpyHostToDevice));
    int cap = 64;
    while (cap < max_codeg) cap <<= 1;
    const size_t lds = (size_t)4 * cap * 4;
    if (lds > 64 * 1024) DESC_HIP(hipFuncSetAttribute((const void*)k_cemp_samples, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_cemp_samples, dim3((unsigned)std::min<int64_t>(8192, (mp + 3) / 4)), dim3(256), lds, 0, *o_pos, d_ii, d_jj, d_bits, d_rank,
                       d_rowptr, d_adj_eid, *o_k, *o_ejk, *o_eki, want_pk ? *o_pk : (uint32_t*)nullptr, mp, (int)words, (int)nsample, seed, cap);
    DESC_HIP(hipGetLastError());
    DESC_HIP(hipDeviceSynchronize());
    return DESC_OK;
}

// Full index structure of a device-built structure on the host: derive e_jk / e_ki and the mirror
// maps on the device, copy everything down once.
static std::atomic<int64_t> g_host_exports{0};
std::atomic<uint64_t> g_ind_upload_count{0};

int structure_ensure_host(desc_structure* s) {
    if (!s || s->host_cycles) return DESC_OK;
    g_host_exports.fetch_add(1);
    const int64_t mc = s->m_cycle, mp = s->m_pos;
    DESC_HIP(hipSetDevice(s->dev));
    if (s->ev_fill) {       // the sampled cycles come from a kernel on the builder's own stream: wait for it, and report ITS failure as such
        const hipError_t ef = hipEventSynchronize((hipEvent_t)s->ev_fill);
        if (ef != hipSuccess) return fail(DESC_ERR_HIP, "cycle sampling kernel failed: %s", hipGetErrorString(ef));
    }
    if (s->d_codeg && (int64_t)s->codeg.size() != s->m) {
        s->codeg.resize((size_t)s->m);
        DESC_HIP(hipMemcpy(s->codeg.data(), s->d_codeg, sizeof(int32_t) * s->m, hipMemcpyDeviceToHost));
    }
    DevBuf D;
    int rc;
    int32_t *d_ejk, *d_eki, *d_ikj, *d_jki;
    if ((rc = D.alloc(&d_ejk, mc)) || (rc = D.alloc(&d_eki, mc)) || (rc = D.alloc(&d_ikj, mc)) || (rc = D.alloc(&d_jki, mc))) return rc;
    const unsigned g = (unsigned)std::min<int64_t>(8192, (mp + 3) / 4);
    hipLaunchKernelGGL(k_cycle_edges, dim3(g), dim3(256), 0, 0, s->d_pos, s->d_cum, s->d_ii, s->d_jj, s->d_k, s->d_rowptr, s->d_adj, s->d_adj_eid,
                       d_ejk, d_eki, mp);
    hipLaunchKernelGGL(k_mirror, dim3(g), dim3(256), 0, 0, s->d_pos, s->d_cum, s->d_poe, s->d_ii, s->d_jj, s->d_k, d_ejk, d_eki, d_ikj, d_jki, mp);
    DESC_HIP(hipGetLastError());
    s->k.resize((size_t)mc); s->e_jk.resize((size_t)mc); s->e_ki.resize((size_t)mc); s->ikj.resize((size_t)mc); s->jki.resize((size_t)mc);
    DESC_HIP(hipMemcpy(s->k.data(), s->d_k, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->e_jk.data(), d_ejk, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->e_ki.data(), d_eki, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->ikj.data(), d_ikj, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    DESC_HIP(hipMemcpy(s->jki.data(), d_jki, sizeof(int32_t) * mc, hipMemcpyDeviceToHost));
    s->host_cycles = true;
    return DESC_OK;
}

void structure_free_device(desc_structure* s) {
    if (!s || s->dev < 0) return;
    (void)hipSetDevice(s->dev);
    if (s->ev_fill) {
        if (hipEventSynchronize((hipEvent_t)s->ev_fill) != hipSuccess)      // nobody consumed the cycles: say so at least
            fprintf(stderr, "[desc_amd] the cycle-sampling kernel of a device-built structure failed: %s\n", hipGetErrorString(hipGetLastError()));
        (void)hipEventDestroy((hipEvent_t)s->ev_fill); s->ev_fill = nullptr;
    }
    if (s->fill_stream) { stream_release((hipStream_t)s->fill_stream); s->fill_stream = nullptr; }
    for (void* q : s->d_build_blocks) dev_free(q);
    s->d_build_blocks.clear();
    for (void* q : {(void*)s->d_k, (void*)s->d_tau, (void*)s->d_ktau, (void*)s->d_rowptr, (void*)s->d_adj, (void*)s->d_adj_eid, (void*)s->d_ii,
                    (void*)s->d_jj, (void*)s->d_pos, (void*)s->d_cum, (void*)s->d_poe, (void*)s->d_bits, (void*)s->d_rank})
        dev_free(q);
    s->d_k = nullptr; s->d_tau = nullptr; s->d_ktau = nullptr; s->d_bits = nullptr; s->d_rank = nullptr;
    s->d_rowptr = s->d_adj = s->d_adj_eid = s->d_ii = s->d_jj = s->d_pos = s->d_cum = s->d_poe = nullptr;
    s->dev = -1;
}

}  // namespace desc

extern "C" int64_t desc_structure_host_exports(void) { return desc::g_host_exports.load(); }

