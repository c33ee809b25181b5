// Host construction of the sampled 3-cycle structure.
//
// Reproduces, without any n x n / n x m_pos dense array:
//   DESC_PGD.m:23-34   adjacency and per-edge codegree  (CoDeg = (A*A).*A)
//   DESC_PGD.m:36-54   edges with cycles, n_sample = max(ceil(median/4),30), cum_ind
//   DESC_PGD.m:79-96   per-edge common-neighbour list, sampling when codeg >= n_sample
//   DESC_PGD.m:103-127 mirror-cycle maps IKJ / JKI
// `datasample` is replaced by the keyed selection documented in desc_amd.h.
//
// Method: adjacency rows as bitmaps (AND + popcount gives the codegree, set bits of
// the AND enumerate the common neighbours in ascending order, a per-word prefix
// popcount turns a neighbour id into its CSR slot and hence its edge id).  Graphs
// whose bitmaps would exceed 1 GiB fall back to merging sorted CSR rows.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

#include "common.h"

namespace desc {
namespace {

struct Graph {
    int64_t n, m;
    const int32_t* ii;
    const int32_t* jj;
    hvec<int64_t> rowptr;     // n+1
    hvec<int32_t> col, eid;   // 2m, neighbours ascending per row
    bool use_bits = false;
    int64_t words = 0;
    hvec<uint64_t> bits;      // n*words
    hvec<uint32_t> rank;      // n*words: neighbours of the row before this word
};

template <class F>
void parallel_for(int64_t count, F&& body) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 32u)), std::max<int64_t>(1, count / 2048));
    if (nt <= 1) { body(0, count, 0); return; }
    // interleaved blocks: low-numbered edges have larger neighbour lists on average
    const int64_t chunk = 1024;
    run_threads(nt, [&](int t) {
        for (int64_t a = (int64_t)t * chunk; a < count; a += (int64_t)nt * chunk)
            body(a, std::min(count, a + chunk), t);
    });
}

void build_graph(Graph& g) {
    const int64_t n = g.n, m = g.m;
    g.rowptr.assign(n + 1, 0);
    for (int64_t e = 0; e < m; ++e) { g.rowptr[g.ii[e] + 1]++; g.rowptr[g.jj[e] + 1]++; }
    for (int64_t v = 0; v < n; ++v) g.rowptr[v + 1] += g.rowptr[v];
    g.col.resize(2 * m); g.eid.resize(2 * m);
    hvec<int64_t> fill(g.rowptr.begin(), g.rowptr.end() - 1);
    // Edges are sorted by (i,j): a row receives its smaller neighbours (from edges
    // (x,v), x<v) before its larger ones (edges (v,x)), each run ascending.
    for (int64_t e = 0; e < m; ++e) {
        int32_t i = g.ii[e], j = g.jj[e];
        g.col[fill[i]] = j; g.eid[fill[i]++] = (int32_t)e;
        g.col[fill[j]] = i; g.eid[fill[j]++] = (int32_t)e;
    }
    g.words = (n + 63) / 64;
    g.use_bits = (double)n * (double)g.words * 8.0 <= 1073741824.0;
    if (g.use_bits) {
        g.bits.assign((size_t)n * g.words, 0);
        g.rank.assign((size_t)n * g.words, 0);
        parallel_for(n, [&](int64_t a, int64_t b, int) {
            for (int64_t v = a; v < b; ++v) {
                uint64_t* row = &g.bits[(size_t)v * g.words];
                for (int64_t t = g.rowptr[v]; t < g.rowptr[v + 1]; ++t) row[g.col[t] >> 6] |= 1ull << (g.col[t] & 63);
                uint32_t acc = 0;
                uint32_t* rk = &g.rank[(size_t)v * g.words];
                for (int64_t w = 0; w < g.words; ++w) { rk[w] = acc; acc += (uint32_t)__builtin_popcountll(row[w]); }
            }
        });
    }
}

inline int32_t codeg_of(const Graph& g, int32_t i, int32_t j) {
    if (g.use_bits) {
        const uint64_t* a = &g.bits[(size_t)i * g.words];
        const uint64_t* b = &g.bits[(size_t)j * g.words];
        int32_t c = 0;
        for (int64_t w = 0; w < g.words; ++w) c += __builtin_popcountll(a[w] & b[w]);
        return c;
    }
    int64_t a = g.rowptr[i], ae = g.rowptr[i + 1], b = g.rowptr[j], be = g.rowptr[j + 1];
    int32_t c = 0;
    while (a < ae && b < be) {
        int32_t x = g.col[a], y = g.col[b];
        c += (x == y); a += (x <= y); b += (y <= x);
    }
    return c;
}

struct Tri { int32_t k, ejk, eki; };

// ascending common neighbours of (i,j) with the ids of edges {j,k} and {k,i}
inline void common_of(const Graph& g, int32_t i, int32_t j, hvec<Tri>& out) {
    out.clear();
    if (g.use_bits) {
        const uint64_t* a = &g.bits[(size_t)i * g.words];
        const uint64_t* b = &g.bits[(size_t)j * g.words];
        const uint32_t* ra = &g.rank[(size_t)i * g.words];
        const uint32_t* rb = &g.rank[(size_t)j * g.words];
        for (int64_t w = 0; w < g.words; ++w) {
            uint64_t x = a[w] & b[w];
            while (x) {
                int bit = __builtin_ctzll(x);
                uint64_t below = (1ull << bit) - 1;
                int64_t pa = g.rowptr[i] + ra[w] + __builtin_popcountll(a[w] & below);
                int64_t pb = g.rowptr[j] + rb[w] + __builtin_popcountll(b[w] & below);
                out.push_back({(int32_t)(w * 64 + bit), g.eid[pb], g.eid[pa]});
                x &= x - 1;
            }
        }
        return;
    }
    int64_t a = g.rowptr[i], ae = g.rowptr[i + 1], b = g.rowptr[j], be = g.rowptr[j + 1];
    while (a < ae && b < be) {
        int32_t x = g.col[a], y = g.col[b];
        if (x == y) out.push_back({x, g.eid[b], g.eid[a]});
        a += (x <= y); b += (y <= x);
    }
}

}  // namespace

// CEMP's cycle sample (Algorithms/CEMP.m:44-65): nsample third vertices per edge-with-cycles,
// drawn WITH replacement; datasample's RNG is replaced by  CoInd[sample_key(seed, edge, t) mod codeg].
int build_cemp_samples_host(const desc_problem* prob, int32_t nsample, uint64_t seed, hvec<int32_t>& pos_edge,
                            hvec<int32_t>& kk, hvec<int32_t>& e_jk, hvec<int32_t>& e_ki) {
    Graph g; g.n = prob->n; g.m = prob->m; g.ii = prob->ind_i; g.jj = prob->ind_j;
    build_graph(g);
    hvec<int32_t> cd((size_t)g.m);
    parallel_for(g.m, [&](int64_t a, int64_t b, int) { for (int64_t e = a; e < b; ++e) cd[e] = codeg_of(g, g.ii[e], g.jj[e]); });
    pos_edge.clear();
    for (int64_t e = 0; e < g.m; ++e) if (cd[e] > 0) pos_edge.push_back((int32_t)e);
    const int64_t mp = (int64_t)pos_edge.size();
    kk.assign((size_t)mp * nsample, 0); e_jk.assign((size_t)mp * nsample, 0); e_ki.assign((size_t)mp * nsample, 0);
    parallel_for(mp, [&](int64_t a, int64_t b, int) {
        hvec<Tri> tri;
        for (int64_t l = a; l < b; ++l) {
            const int32_t e = pos_edge[l];
            common_of(g, g.ii[e], g.jj[e], tri);
            for (int32_t t = 0; t < nsample; ++t) {
                const Tri& q = tri[sample_key(seed, (uint64_t)e, (uint64_t)t) % tri.size()];
                kk[(size_t)l * nsample + t] = q.k; e_jk[(size_t)l * nsample + t] = q.ejk; e_ki[(size_t)l * nsample + t] = q.eki;
            }
        }
    });
    return DESC_OK;
}

void build_csr(int64_t n, int64_t m, const int32_t* ii, const int32_t* jj, hvec<int32_t>& rowptr, hvec<int32_t>& adj,
               hvec<int32_t>& adj_eid) {
    rowptr.assign((size_t)n + 1, 0); adj.resize((size_t)2 * m); adj_eid.resize((size_t)2 * m);
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, (int64_t)(16 << 20) / std::max<int64_t>(n, 1)));
    if (m < (1 << 18)) T = 1;
    if (T <= 1) {
        // Edges are sorted by (i,j): a row receives its smaller neighbours (from edges (x,v), x<v)
        // before its larger ones (edges (v,x)), each run ascending.
        for (int64_t e = 0; e < m; ++e) { rowptr[ii[e] + 1]++; rowptr[jj[e] + 1]++; }
        for (int64_t v = 0; v < n; ++v) rowptr[v + 1] += rowptr[v];
        hvec<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
        for (int64_t e = 0; e < m; ++e) {
            const int32_t i = ii[e], j = jj[e];
            adj[fill[i]] = j; adj_eid[fill[i]++] = (int32_t)e;
            adj[fill[j]] = i; adj_eid[fill[j]++] = (int32_t)e;
        }
        return;
    }
    // T contiguous chunks of edges; per chunk the number of edges ending (lo) / starting (up) at every node,
    // turned into per-chunk start offsets inside the lower / upper part of each row: the same slots as the
    // serial pass, whatever T is
    hvec<int32_t> lo((size_t)T * n, 0), up((size_t)T * n, 0), lowtot((size_t)n);
    auto run = [&](auto&& body) { run_threads(T, [&](int t) { body(t, m * t / T, m * (t + 1) / T); }); };
    run([&](int t, int64_t a, int64_t b) {
        int32_t* l = &lo[(size_t)t * n]; int32_t* u = &up[(size_t)t * n];
        for (int64_t e = a; e < b; ++e) { u[ii[e]]++; l[jj[e]]++; }
    });
    for (int64_t v = 0; v < n; ++v) {
        int32_t accl = 0, accu = 0;
        for (int t = 0; t < T; ++t) {
            const int32_t cl = lo[(size_t)t * n + v], cu = up[(size_t)t * n + v];
            lo[(size_t)t * n + v] = accl; up[(size_t)t * n + v] = accu;
            accl += cl; accu += cu;
        }
        lowtot[v] = accl;
        rowptr[v + 1] = rowptr[v] + accl + accu;
    }
    run([&](int t, int64_t a, int64_t b) {
        int32_t* l = &lo[(size_t)t * n]; int32_t* u = &up[(size_t)t * n];
        for (int64_t e = a; e < b; ++e) {
            const int32_t i = ii[e], j = jj[e];
            const int32_t pu = rowptr[i] + lowtot[i] + u[i]++, pl = rowptr[j] + l[j]++;
            adj[pu] = j; adj_eid[pu] = (int32_t)e;
            adj[pl] = i; adj_eid[pl] = (int32_t)e;
        }
    });
}

int validate_problem(const desc_problem* prob, bool need_rij) {
    if (!prob) return fail(DESC_ERR_INVALID, "problem is NULL");
    if (prob->n < 0 || prob->m < 0) return fail(DESC_ERR_INVALID, "negative n or m");
    if (prob->m > 0 && (!prob->ind_i || !prob->ind_j)) return fail(DESC_ERR_INVALID, "ind_i / ind_j is NULL");
    if (need_rij && prob->m > 0 && !prob->rij) return fail(DESC_ERR_INVALID, "rij is NULL");
    if (prob->m >= (1ll << 30)) return fail(DESC_ERR_TOO_LARGE, "m = %lld exceeds 2^30-1", (long long)prob->m);
    if (prob->n >= (1ll << 31)) return fail(DESC_ERR_TOO_LARGE, "n = %lld exceeds 2^31-1", (long long)prob->n);
    // first offending row, if any (large edge lists: the rows are checked in chunks by several threads, the smallest offending row is reported --
    // the same one a sequential scan finds)
    const int64_t m = prob->m;
    auto first_bad = [&](int64_t a, int64_t b) -> int64_t {
        for (int64_t e = a; e < b; ++e) {
            const int32_t i = prob->ind_i[e], j = prob->ind_j[e];
            if (i < 0 || j >= prob->n || i >= j) return e;
            if (e > 0) {
                const int32_t pi = prob->ind_i[e - 1], pj = prob->ind_j[e - 1];
                if (pi > i || (pi == i && pj >= j)) return e;
            }
        }
        return -1;
    };
    int64_t bad = -1;
    if (m >= (1 << 20)) {
        const int T = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 8u));
        std::vector<int64_t> found((size_t)T, -1);
        run_threads(T, [&](int t) { found[(size_t)t] = first_bad(m * t / T, m * (t + 1) / T); });
        for (int t = 0; t < T && bad < 0; ++t) bad = found[(size_t)t];
    } else bad = first_bad(0, m);
    if (bad >= 0) {
        const int64_t e = bad;
        const int32_t i = prob->ind_i[e], j = prob->ind_j[e];
        if (i < 0 || j >= prob->n || i >= j)
            return fail(DESC_ERR_INVALID, "edge %lld = (%d,%d): need 0 <= i < j < n = %lld", (long long)e, i, j, (long long)prob->n);
        return fail(DESC_ERR_INVALID, "Ind is not strictly sorted by (i,j) at row %lld (DESC_PGD.m:5 requires it)", (long long)e);
    }
    return DESC_OK;
}

int build_structure_host(const desc_problem* prob, int32_t n_sample_min, uint64_t seed, desc_structure* s) {
    auto t0 = std::chrono::steady_clock::now();
    Graph g; g.n = prob->n; g.m = prob->m; g.ii = prob->ind_i; g.jj = prob->ind_j;
    const int64_t m = g.m;
    build_graph(g);

    s->n = g.n; s->m = m;
    s->codeg.assign(m, 0);
    parallel_for(m, [&](int64_t a, int64_t b, int) {
        for (int64_t e = a; e < b; ++e) s->codeg[e] = codeg_of(g, g.ii[e], g.jj[e]);
    });

    // edges with cycles, median of their codegree (DESC_PGD.m:36-43)
    s->pos_edge.clear();
    hvec<int32_t> pos_cd;
    for (int64_t e = 0; e < m; ++e) if (s->codeg[e] > 0) { s->pos_edge.push_back((int32_t)e); pos_cd.push_back(s->codeg[e]); }
    s->m_pos = (int64_t)s->pos_edge.size();
    int32_t n_sample = n_sample_min;            // median([]) = NaN, max(NaN,30) = 30
    if (s->m_pos > 0) {
        size_t h = pos_cd.size() / 2;
        std::nth_element(pos_cd.begin(), pos_cd.begin() + h, pos_cd.end());
        double med = pos_cd[h];
        if ((pos_cd.size() & 1) == 0) {
            int32_t lower = *std::max_element(pos_cd.begin(), pos_cd.begin() + h);
            med = 0.5 * ((double)lower + med);
        }
        n_sample = std::max(n_sample_min, (int32_t)std::ceil(med / 4.0));
    }
    s->n_sample = n_sample;
    s->cum_ind.assign(s->m_pos + 1, 0);
    s->max_cnt = 0;
    for (int64_t l = 0; l < s->m_pos; ++l) {
        int32_t cnt = std::min(s->codeg[s->pos_edge[l]], n_sample);     // :45
        s->cum_ind[l + 1] = s->cum_ind[l] + cnt;
        s->max_cnt = std::max(s->max_cnt, cnt);
    }
    s->m_cycle = s->cum_ind[s->m_pos];
    if (s->m_cycle >= (1ll << 31) - 1) return fail(DESC_ERR_TOO_LARGE, "m_cycle = %lld exceeds 2^31-2", (long long)s->m_cycle);
    const int64_t mc = s->m_cycle;
    s->k.assign(mc, 0); s->e_jk.assign(mc, 0); s->e_ki.assign(mc, 0); s->ikj.assign(mc, -1); s->jki.assign(mc, -1);

    // cycle lists (DESC_PGD.m:79-96)
    parallel_for(s->m_pos, [&](int64_t a, int64_t b, int) {
        hvec<Tri> tri;
        hvec<std::pair<uint64_t, int32_t>> keyed;
        for (int64_t l = a; l < b; ++l) {
            int32_t e = s->pos_edge[l];
            common_of(g, g.ii[e], g.jj[e], tri);
            int64_t lo = s->cum_ind[l];
            int32_t cd = (int32_t)tri.size();
            if (cd >= n_sample) {                                        // :83 (>=)
                keyed.resize(cd);
                for (int32_t t = 0; t < cd; ++t) keyed[t] = {sample_key(seed, (uint64_t)e, (uint64_t)tri[t].k), t};
                // common neighbours are distinct, so (key, position) orders exactly like (key, k)
                std::nth_element(keyed.begin(), keyed.begin() + n_sample, keyed.end());
                std::sort(keyed.begin(), keyed.begin() + n_sample,
                          [](const auto& x, const auto& y) { return x.second < y.second; });
                for (int32_t t = 0; t < n_sample; ++t) {
                    const Tri& q = tri[keyed[t].second];
                    s->k[lo + t] = q.k; s->e_jk[lo + t] = q.ejk; s->e_ki[lo + t] = q.eki;
                }
            } else {
                for (int32_t t = 0; t < cd; ++t) { s->k[lo + t] = tri[t].k; s->e_jk[lo + t] = tri[t].ejk; s->e_ki[lo + t] = tri[t].eki; }
            }
        }
    });

    // mirror maps (DESC_PGD.m:103-127): binary search of j in the sampled list of
    // edge {i,k}, and of i in the sampled list of edge {j,k}
    hvec<int32_t> pos_of_edge(m, -1);                             // CoDeg_pos_ind_long (:53-54)
    for (int64_t l = 0; l < s->m_pos; ++l) pos_of_edge[s->pos_edge[l]] = (int32_t)l;
    parallel_for(s->m_pos, [&](int64_t a, int64_t b, int) {
        for (int64_t l = a; l < b; ++l) {
            int32_t e = s->pos_edge[l], i = g.ii[e], j = g.jj[e];
            for (int64_t c = s->cum_ind[l]; c < s->cum_ind[l + 1]; ++c) {
                int32_t IK = pos_of_edge[s->e_ki[c]], JK = pos_of_edge[s->e_jk[c]];
                const int32_t* kb = s->k.data();
                const int32_t* p = std::lower_bound(kb + s->cum_ind[IK], kb + s->cum_ind[IK + 1], j);
                if (p != kb + s->cum_ind[IK + 1] && *p == j) s->ikj[c] = (int32_t)(p - kb);
                p = std::lower_bound(kb + s->cum_ind[JK], kb + s->cum_ind[JK + 1], i);
                if (p != kb + s->cum_ind[JK + 1] && *p == i) s->jki[c] = (int32_t)(p - kb);
            }
        }
    });
    s->ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DESC_OK;
}

}  // namespace desc
