// Host-side planners of the node layout (declared in node_plan.h; used by csrc/pgd.hip: setup_node, desc_debug_band_plan).
#include "node_plan.h"

#include <algorithm>
#include <chrono>
#include <cstdio>

namespace desc {


// row_cap > 0: bands = maximal runs of consecutive nodes whose CSR rows hold <= row_cap entries together (the band
// sweep keeps them in the LDS); row_cap == 0: bands of a fixed number of nodes sized for the L2 (k_sweep_node).
int make_node_plan(const desc_problem* prob, const desc_structure* s, int max_deg, int world, int max_seg, int row_cap, NodePlan& P, int xparts) {
    const int64_t mp = s->m_pos, n = prob->n, m = prob->m;
    const bool timing = env_int("DESC_DEBUG_TIMING", 0) > 1;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[desc_amd] node plan %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    const int32_t* ii = prob->ind_i; const int32_t* jj = prob->ind_j; const int32_t* pe = s->pos_edge.data();
    if ((int64_t)s->rowptr_host.size() == n + 1) P.rowptr = s->rowptr_host;      // the device builder already made them
    else {
        P.rowptr.assign((size_t)n + 1, 0);
        for (int64_t e = 0; e < m; ++e) { P.rowptr[ii[e] + 1]++; P.rowptr[jj[e] + 1]++; }
        for (int64_t v = 0; v < n; ++v) P.rowptr[v + 1] += P.rowptr[v];
    }
    P.band_lo.clear();
    if (row_cap > 0) {
        P.band = 0;
        for (int64_t v = 0; v < n;) {
            P.band_lo.push_back((int32_t)v);
            int64_t e = v + 1;                               // a band holds at least one node (max_deg <= row_cap is the caller's check)
            while (e < n && P.rowptr[e + 1] - P.rowptr[v] <= row_cap) ++e;
            v = e;
        }
    } else {
        int band = env_int("DESC_DEBUG_BAND", 0);
        if (band <= 0)   // rows of one band should stay in an XCD's 4 MiB L2 next to the streamed arrays: ~1 MiB
            band = (int)std::max<int64_t>(8, std::min<int64_t>(512, (1 << 20) / (8 * (int64_t)std::max(1, max_deg))));
        P.band = band;
        for (int64_t v = 0; v < n; v += band) P.band_lo.push_back((int32_t)v);
    }
    P.band_lo.push_back((int32_t)n);
    const int64_t nb = (int64_t)P.band_lo.size() - 1;
    // order by (band(i), j, i): Ind -- and with it pos_edge -- is sorted by (i, j), so a band is a
    // contiguous range of pos_edge and one stable counting sort by j per band (bands in parallel)
    // gives the order in O(m_pos + bands * n)
    P.order.resize((size_t)mp);
    P.cum2.assign((size_t)mp + 1, 0);
    P.bstart.assign((size_t)nb + 1, mp);       // first position of every band in pos_edge
    {
        int64_t l = 0;                         // pos_edge is sorted by (i, j): the first edge with i >= band_lo[b], by binary search from the previous band's start
        for (int64_t b = 0; b <= nb; ++b) {
            const int32_t want = P.band_lo[b];
            l = std::partition_point(pe + l, pe + mp, [&](int32_t e) { return ii[e] < want; }) - pe;
            P.bstart[b] = l;
        }
    }
    lap("bands + starts");
    host_parallel(nb, [&](int64_t b0, int64_t b1) {
        hvec<int32_t> cnt((size_t)n + 1);
        for (int64_t b = b0; b < b1; ++b) {
            const int64_t lo = P.bstart[b], hi = P.bstart[b + 1];
            if (lo == hi) continue;
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int64_t l = lo; l < hi; ++l) cnt[jj[pe[l]] + 1]++;
            for (int64_t v = 0; v < n; ++v) cnt[v + 1] += cnt[v];
            for (int64_t l = lo; l < hi; ++l) {
                const int64_t q = lo + cnt[jj[pe[l]]]++;
                P.order[q] = (int32_t)l;
                P.cum2[q + 1] = (int32_t)(s->cum_ind[l + 1] - s->cum_ind[l]);        // cycle count of the segment at device position q: summed below
            }
        }
    }, mp >= (1 << 18) ? 1 : nb + 1);          // small problems: one thread
    lap("per-band sorts");
    for (int64_t q = 0; q < mp; ++q) P.cum2[q + 1] += P.cum2[q];
    lap("prefix sums");
    // Several ranks: a rank owns whole bands (its exchange layout is indexed by node ranges, k_xpos); the cuts go to the band boundaries
    // that split the cycles most evenly (a band is ~0.4 % of the work at C4), and chunks do not straddle them.
    hvec<int64_t> cut_seg;                                   // device position of the first segment of every rank, + mp
    P.rank_node.assign((size_t)world + 1, (int32_t)n);
    P.rank_node[0] = 0;
    cut_seg.assign((size_t)world + 1, mp);
    cut_seg[0] = 0;
    hvec<int64_t> cut_band((size_t)world + 1, nb);            // band index of every rank's first band
    cut_band[0] = 0;
    if (world > 1) {
        const int64_t total = P.cum2[mp];
        int64_t b = 0;
        for (int r = 1; r < world; ++r) {
            const int64_t want = total * r / world;
            while (b < nb && (int64_t)P.cum2[P.bstart[b]] < want) ++b;          // first band that starts at or beyond the target
            if (b > 0 && b <= nb && want - (int64_t)P.cum2[P.bstart[b - 1]] < (int64_t)P.cum2[P.bstart[std::min(b, nb)]] - want && P.bstart[b - 1] > cut_seg[r - 1]) --b;
            const int64_t bb = std::min(b, nb);
            cut_seg[r] = std::max<int64_t>(P.bstart[bb], cut_seg[r - 1]);
            cut_band[r] = std::max<int64_t>(bb, cut_band[r - 1]);
            P.rank_node[r] = bb < nb ? P.band_lo[bb] : (int32_t)n;
            if (P.rank_node[r] < P.rank_node[r - 1]) P.rank_node[r] = P.rank_node[r - 1];
        }
    }
    // exchange parts of every rank: whole bands again, cut where the cycles of the rank split most evenly
    P.xparts = std::max(1, xparts);
    P.vnode.assign((size_t)world * P.xparts + 1, (int32_t)n);
    P.vseg.assign((size_t)world * P.xparts + 1, mp);
    for (int r = 0; r < world; ++r) {
        const int64_t b0 = cut_band[r], b1 = cut_band[r + 1];
        const int64_t c0 = P.cum2[P.bstart[b0]], c1 = P.cum2[P.bstart[b1]];
        int64_t b = b0;
        for (int c = 0; c < P.xparts; ++c) {
            if (c > 0) {
                const int64_t want = c0 + (c1 - c0) * c / P.xparts;
                while (b < b1 && (int64_t)P.cum2[P.bstart[b]] < want) ++b;
                if (b > b0 && want - (int64_t)P.cum2[P.bstart[b - 1]] < (int64_t)P.cum2[P.bstart[b]] - want && b - 1 >= b0) --b;     // the nearer boundary
            }
            P.vnode[(size_t)r * P.xparts + c] = b < nb ? P.band_lo[b] : (int32_t)n;
            P.vseg[(size_t)r * P.xparts + c] = P.bstart[b];
        }
    }
    for (size_t t = 1; t < P.vnode.size(); ++t) {             // monotone (empty ranks / parts collapse)
        if (P.vnode[t] < P.vnode[t - 1]) P.vnode[t] = P.vnode[t - 1];
        if (P.vseg[t] < P.vseg[t - 1]) P.vseg[t] = P.vseg[t - 1];
    }
    // (the first part of a rank starts where the rank starts)
    for (int r = 0; r < world; ++r) { P.vnode[(size_t)r * P.xparts] = P.rank_node[r]; P.vseg[(size_t)r * P.xparts] = cut_seg[r]; }
    P.chunk_seg.clear();
    P.chunk_seg.push_back(0);
    P.rank_chunk.assign((size_t)world + 1, 0);
    {
        int rnext = 1;
        for (int64_t q = 0; q < mp;) {     // chunks: <= CHUNK_CAP cycles and <= CHUNK_SEG segments
            while (rnext < world && cut_seg[rnext] <= q) P.rank_chunk[rnext++] = (int64_t)P.chunk_seg.size() - 1;
            const int64_t stop = rnext < world ? cut_seg[rnext] : mp;
            const int64_t lim = std::min<int64_t>(stop, q + max_seg);         // last segment boundary within CHUNK_CAP cycles: binary search (cum2 increases strictly)
            const int64_t e = (int64_t)(std::upper_bound(P.cum2.begin() + q, P.cum2.begin() + lim + 1, (int64_t)P.cum2[q] + CHUNK_CAP,
                                                         [](int64_t v, int32_t c) { return v < (int64_t)c; }) - P.cum2.begin()) - 1;
            q = e;                          // max_cnt <= MAX_SEG_CYCLES <= CHUNK_CAP: always advances
            P.chunk_seg.push_back((int32_t)q);
        }
        const int64_t nch = (int64_t)P.chunk_seg.size() - 1;
        while (rnext <= world) P.rank_chunk[rnext++] = nch;
    }
    lap("chunks");
    return DESC_OK;
}

// The work of every workgroup of the band sweep as a list of pieces (band rows + a range of that band's segments).
//  * Sfull fits the L2s (small graphs): one contiguous range of segments per workgroup (equal cycle counts), split at band
//    boundaries -- each workgroup loads one or two bands.
//  * otherwise S({j,k}) rows would be fetched from the Infinity Cache once per (band, j) (measured: 8 % of the iteration at C4,
//    17 % at C5): the (band, j) plane is cut into units (band b, block of JB consecutive j) whose j-rows (~1.5 MiB) fit an
//    XCD's L2, and the units are dealt in j-block-major order to the least-loaded workgroup (deterministic list scheduling), so
//    that at any moment all workgroups gather from the same block of rows.
// Host only: no device call (also reachable through desc_debug_band_plan, which the CPU tests and sanitizer builds use).
void plan_band_pieces(const desc_problem* prob, const desc_structure* s, const NodePlan& P, int64_t seg_lo, int64_t seg_hi, int64_t cyc_lo, int64_t mcl,
                      int G, hvec<PieceDesc>& pieces, hvec<int32_t>& piece_ptr, int& band_rows, bool& jmajor_out, int* tail_first_out, int* n_tail_out, int max_tail) {
    const hvec<int32_t>& cum2 = P.cum2;
    const int64_t n = prob->n, m = prob->m;
    const bool timing = env_int("DESC_DEBUG_TIMING", 0) > 1;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[desc_amd] band plan %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    band_rows = 0;
    const int64_t nbands = (int64_t)P.band_lo.size() - 1;
    hvec<hvec<PieceDesc>> per_wg((size_t)G);
    hvec<PieceDesc> tail;                                    // shared tail pieces, j-block-major (BandSweepArgs)
    auto piece_of = [&](int64_t bd, int64_t q0, int64_t q1) {
        const int32_t row_lo = P.rowptr[P.band_lo[bd]], row_len = P.rowptr[P.band_lo[bd + 1]] - row_lo;
        band_rows = std::max(band_rows, (int)row_len);
        return PieceDesc{row_lo, row_len, (int32_t)q0, (int32_t)q1};
    };
    // largest x in [lo, e] with cycles(lo..x) <= room (cum2 increases strictly: a binary search instead of a walk over the segments)
    auto reach = [&](int64_t lo, int64_t e, int64_t room) {
        if ((int64_t)cum2[e] - cum2[lo] <= room) return e;                   // the whole range fits (most units): no search
        return (int64_t)(std::upper_bound(cum2.begin() + lo, cum2.begin() + e + 1, (int64_t)cum2[lo] + room,
                                          [](int64_t v, int32_t c) { return v < (int64_t)c; }) - cum2.begin()) - 1;
    };
    const int jmajor_env = env_int("DESC_DEBUG_JMAJOR", -1);
    // (C3, 6.4 MB of S: contiguous 0.167 ms, units 0.184.  Round 4: a rank of a sharded run with < 150 K cycles per workgroup -- C4 over 8 GPUs:
    //  61 K -- takes contiguous ranges too: a unit's row load and pipeline fill cost as much as its cycles there; measured one rank at a time,
    //  profiles/r04_shard_w8_c4_{default,jmajor0}.json: 1033 -> 272 pieces per rank, sweep 196 -> 173 us)
    //  C5 over 8 GPUs -- 73 K cycles per workgroup, but 80 MB of S and segments of 30 cycles -- is the other way round: units 250 us, contiguous 287 us
    //  (profiles/r04_shard_w8_c5_v2*.json); the rule below separates the two measured cases by the size of S, nothing deeper.
    const bool small_share = mcl < (int64_t)150000 * G && (int64_t)2 * m * 8 <= (48ll << 20);
    const bool jmajor = jmajor_env >= 0 ? jmajor_env != 0 : ((int64_t)2 * m * 8 > (12ll << 20) && !small_share);
    jmajor_out = jmajor;
    // what a piece costs besides its cycles -- the load of the band's rows into the LDS and the fill of the register pipeline -- in cycle
    // units.  Measured per workgroup with DESC_DEBUG_WGCLOCK (tools/wg_clock.py, least squares of the durations on the plan): 11 us per piece
    // at C2 (= 7600 cycles at 1.5 ns per cycle), 9 us at C3 (4800), 8-13 us at C4 (4000-6400).
    // Adopted: 6144 for the contiguous ranges (C2 sweep 130 -> 117 us, C3 121.5 -> 111 us against equal-cycle ranges; 4096 / 8192 within 2 %),
    // 4096 for the j-block-major units (C4: 1205 vs 1212 us at 6144 / 8192) -- profiles/r03_piece_cost.txt.
    const int64_t PC = std::max(0, env_int("DESC_DEBUG_PIECE_COST", jmajor ? 4096 : 6144));
    if (!jmajor) {
        // Contiguous ranges, one per workgroup, equal in cycles + PC per piece (a range that crosses a band boundary is two pieces and loads
        // two sets of rows).  Round 2 made the ranges equal in cycles alone: at C2 / C3 the workgroups with 2-3 pieces finished 10-20 % after
        // the others (durations 97-124 us at C2, correlation 0.86 with the piece count) and the kernel waited for them.
        int64_t q = seg_lo, bd = 0;
        while (bd + 1 < nbands && P.bstart[bd + 1] <= q) ++bd;
        for (int b = 0; b < G && q < seg_hi; ++b) {
            // bands that still begin inside what is left: each of them costs one more piece somewhere
            int64_t bands_left = 0;
            for (int64_t t = bd + 1; t < nbands && P.bstart[t] < seg_hi; ++t) ++bands_left;
            const int64_t left = (cyc_lo + mcl) - cum2[q];
            const int64_t target = (left + PC * ((G - b) + bands_left) + (G - b) - 1) / (G - b);
            int64_t load = 0;
            while (q < seg_hi) {
                while (bd + 1 < nbands && P.bstart[bd + 1] <= q) ++bd;
                const int64_t band_end = std::min<int64_t>(seg_hi, P.bstart[bd + 1]);
                int64_t e = band_end;
                if (b + 1 < G) {
                    const int64_t room = target - load - PC;
                    if (room <= 0 && load > 0) break;                       // not even the row load fits: the next workgroup starts here
                    int64_t lo2 = q, hi2 = band_end;                         // largest e with cycles(q..e) <= room
                    while (lo2 < hi2) { const int64_t mid = (lo2 + hi2 + 1) >> 1; if (cum2[mid] - cum2[q] <= room) lo2 = mid; else hi2 = mid - 1; }
                    e = std::max<int64_t>(lo2, q + 1);
                }
                per_wg[b].push_back(piece_of(bd, q, e));
                load += PC + (cum2[e] - cum2[q]);
                q = e;
                if (b + 1 < G && load >= target) break;
            }
        }
    } else {
        const int64_t avg_deg = std::max<int64_t>(1, 2 * m / std::max<int64_t>(1, n));
        int64_t JB = env_int("DESC_DEBUG_JBLOCK", 0);
        if (JB <= 0) {
            // rows of a j-block ~1.5 MiB (they share an XCD's 4 MiB L2 with the streams), but wide enough that a unit streams
            // >= 32 K cycles for the ~150 KB of band rows it loads (sparse graphs with short segments: C5 0.55 -> 0.60), up to 4 MiB
            const int64_t jb_l2 = std::max<int64_t>(32, (3ll << 19) / (8 * avg_deg));
            const double cyc_per_pair = (double)mcl / std::max(1.0, 0.5 * (double)nbands * (double)n);     // cycles per (band, j) pair
            const int64_t jb_amort = (int64_t)(32768.0 / std::max(cyc_per_pair, 1.0));
            JB = std::min<int64_t>(std::max(jb_l2, jb_amort), std::max<int64_t>(jb_l2, (4ll << 20) / (8 * avg_deg)));
        }
        const int64_t cap = std::max<int64_t>(16384, mcl / (4 * (int64_t)G));        // cycles per unit at most
        const int64_t nJ = (n + JB - 1) / JB;
        auto j_of = [&](int64_t q) { return (int64_t)prob->ind_j[s->pos_edge[P.order[q]]]; };
        // position of the first segment of band bd with j >= jlim, inside the rank's range
        hvec<int64_t> cur((size_t)nbands), bend((size_t)nbands);
        for (int64_t bd = 0; bd < nbands; ++bd) {
            cur[bd] = std::min(std::max(P.bstart[bd], seg_lo), seg_hi);
            bend[bd] = std::min(std::max(P.bstart[bd + 1], seg_lo), seg_hi);
        }
        // min-heap over (load, wg): the next unit goes to the workgroup that would be free first
        hvec<std::pair<int64_t, int>> heap; heap.reserve((size_t)G);
        for (int b = 0; b < G; ++b) heap.push_back({0, b});
        auto cmp = [](const std::pair<int64_t, int>& x, const std::pair<int64_t, int>& y) { return x > y; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        // Band affinity (round 3): the next unit of a band goes to the workgroup that took the band's previous unit -- whose LDS still holds
        // the band's rows: its piece is simply extended, no row load -- unless that workgroup is more than `slack` cycles ahead of the least
        // loaded one (then plain list scheduling, as in round 2).  Measured (profiles/r03_band_affinity.txt, sweep averages in one call):
        // C4 3636 -> 1738 pieces, 1183 -> 1148 us (-3 %) at a slack of 16 K cycles (~half a unit); 4 K -1 %, 8 K -2 %, 32 K 0, 64 K +2 %,
        // 256 K +23 % (the workgroups drift apart in j and lose the L2 locality of the j rows); C5 (529 bands on 256 workgroups: every
        // workgroup alternates between two bands, little to merge) within noise.  DESC_DEBUG_AFFINITY = slack in K cycles, 0 = off.
        const int64_t aff_slack = (int64_t)env_int("DESC_DEBUG_AFFINITY", 16) * 1024;
        if (aff_slack > 0) {
            hvec<int64_t> load((size_t)G, 0);
            hvec<int> last_wg((size_t)nbands, -1);
            // How even the lists end up (tools/wg_clock.py, DESC_DEBUG_WGCLOCK): with units of ~34 K cycles and the affinity slack the plan's
            // cycle counts spread +-5 % at C4 (455 K .. 509 K) and the workgroups' measured times follow them (correlation 0.74; mean 1105,
            // max 1158 us).  Two remedies, both here:
            //  * fit to target (DESC_DEBUG_FIT per mille, default 0): in the last part of the cycles a unit is cut where the workgroup that takes
            //    it reaches the common target load -- the lists end level to a segment.  Measured: no gain at C4, C5 slightly worse; what is left
            //    of the spread is not in the plan (even XCDs run 1.5 % slower than odd ones: profiles/r03_experiments.txt);
            //  * shared tail (DESC_DEBUG_TAIL per mille, default 20): the last part is queued as small pieces for whichever workgroup finishes
            //    first (k_sweep_band).  First measured with 10 % of the cycles in the queue: it levels the end times (max - mean 4.8 % -> 1.4 %) but
            //    its ~700 small pieces each load their band rows and the mean rises by as much (profiles/r03_experiments.txt).  With the final
            //    kernel and 1.5-4 % in the queue: C4 996-1014 -> 984-999 us, C5 1585-1596 -> 1569-1580 (about -1 %, profiles/r03_piece_cost.txt):
            //    2 % is the default.
            const int64_t tail_target = tail_first_out ? mcl * std::max(0, std::min(500, env_int("DESC_DEBUG_TAIL", 20))) / 1000 : 0;
            const int64_t tail_cap = std::max<int64_t>(4096, tail_target / std::max(1, max_tail - 64));
            const int64_t fit_from = mcl - mcl * std::max(0, std::min(500, env_int("DESC_DEBUG_FIT", 0))) / 1000;
            int64_t fit_target = -1;                         // common final load, fixed when the fitting phase starts
            int64_t dealt = 0;
            // where every band's segments cross the j-block limits: found band by band on all host threads (three dependent random reads per probe),
            // so that the dealing loop below -- sequential by nature -- only does arithmetic
            hvec<int64_t> cross((size_t)(nbands * nJ));
            host_parallel(nbands, [&](int64_t b0, int64_t b1) {
                for (int64_t bd = b0; bd < b1; ++bd) {
                    int64_t from = cur[bd];
                    for (int64_t J = 0; J < nJ; ++J) {
                        int64_t a0 = from, a1 = bend[bd];
                        const int64_t jlim = (J + 1) * JB;
                        while (a0 < a1) { const int64_t mid = (a0 + a1) >> 1; if (j_of(mid) < jlim) a0 = mid + 1; else a1 = mid; }
                        cross[(size_t)(bd * nJ + J)] = a0;
                        from = a0;
                    }
                }
            }, 1);
            lap("j-block crossings");
            for (int64_t J = 0; J < nJ; ++J) {
                for (int64_t bd = 0; bd < nbands; ++bd) {
                    int64_t lo = cur[bd]; const int64_t hi = bend[bd];
                    if (lo >= hi) continue;
                    const int64_t e = cross[(size_t)(bd * nJ + J)];
                    cur[bd] = e;
                    while (lo < e) {
                        int64_t x = reach(lo, e, cap);
                        if (x == lo) x = lo + 1;
                        if (tail_target > 0 && dealt >= mcl - tail_target && (int64_t)tail.size() < max_tail) {      // the rest of the sweep: queue
                            x = reach(lo, e, tail_cap);
                            if (x == lo) x = lo + 1;
                            tail.push_back(piece_of(bd, lo, x));
                            dealt += cum2[x] - cum2[lo];
                            lo = x;
                            continue;
                        }
                        // least-loaded workgroup, lowest index on ties: the minimum first (a plain reduction the compiler vectorises), then its first holder
                        int64_t lmin = load[0];
                        for (int w = 1; w < G; ++w) lmin = std::min(lmin, load[w]);
                        int wmin = 0;
                        while (load[wmin] != lmin) ++wmin;
                        const int wl = last_wg[bd];
                        const bool merge = wl >= 0 && load[wl] <= load[wmin] + aff_slack && !per_wg[wl].empty() && per_wg[wl].back().seg_hi == (int32_t)lo &&
                                           per_wg[wl].back().row_lo == P.rowptr[P.band_lo[bd]];
                        const int wt = merge ? wl : wmin;
                        if (dealt >= fit_from) {
                            if (fit_target < 0) {            // what is left + what is dealt + a row load per workgroup, shared equally
                                int64_t sum = 0;
                                for (int w = 0; w < G; ++w) sum += load[w];
                                fit_target = (sum + (mcl - dealt) + PC * (int64_t)G + G - 1) / G;
                            }
                            const int64_t room = fit_target - load[wt] - (merge ? 0 : PC);
                            if (room < 2048 && load[wmin] + PC + 2048 > fit_target) fit_target += 4096;       // everybody is full: raise the bar a little
                            else if (room >= 2048) {
                                const int64_t y = reach(lo, x, room);
                                if (y > lo) x = y;           // cut the unit where this workgroup reaches the target
                                else x = lo + 1;
                            } else {                         // the band's resident workgroup is full: the least loaded one takes the unit instead
                                last_wg[bd] = -1;
                                continue;
                            }
                        }
                        dealt += cum2[x] - cum2[lo];
                        if (merge) {
                            per_wg[wl].back().seg_hi = (int32_t)x;                 // same rows, contiguous segments: one longer piece
                            load[wl] += cum2[x] - cum2[lo];
                        } else {
                            per_wg[wmin].push_back(piece_of(bd, lo, x));
                            load[wmin] += cum2[x] - cum2[lo] + PC;
                            last_wg[bd] = wmin;
                        }
                        lo = x;
                    }
                }
            }
        } else
        for (int64_t J = 0; J < nJ; ++J) {
            const int64_t jlim = (J + 1) * JB;
            for (int64_t bd = 0; bd < nbands; ++bd) {
                int64_t lo = cur[bd]; const int64_t hi = bend[bd];
                if (lo >= hi) continue;
                int64_t a0 = lo, a1 = hi;                   // first q in [lo, hi) with j(q) >= jlim
                while (a0 < a1) { const int64_t mid = (a0 + a1) >> 1; if (j_of(mid) < jlim) a0 = mid + 1; else a1 = mid; }
                const int64_t e = a0;
                cur[bd] = e;
                while (lo < e) {                             // split units above the cap
                    int64_t x = reach(lo, e, cap);
                    if (x == lo) x = lo + 1;
                    std::pop_heap(heap.begin(), heap.end(), cmp);
                    auto& top = heap.back();
                    per_wg[top.second].push_back(piece_of(bd, lo, x));
                    top.first += cum2[x] - cum2[lo] + PC;    // + the row load and pipeline fill of a piece, in cycle units
                    std::push_heap(heap.begin(), heap.end(), cmp);
                    lo = x;
                }
            }
        }
    }
    pieces.clear();
    lap("units dealt");
    piece_ptr.assign((size_t)G + 1, 0);
    for (int b = 0; b < G; ++b) {
        pieces.insert(pieces.end(), per_wg[b].begin(), per_wg[b].end());
        piece_ptr[b + 1] = (int32_t)pieces.size();
    }
    if (tail_first_out) { *tail_first_out = (int)pieces.size(); *n_tail_out = (int)tail.size(); }
    pieces.insert(pieces.end(), tail.begin(), tail.end());
    if (pieces.empty()) pieces.push_back(PieceDesc{0, 0, 0, 0});
}

// LDS budget of a band's rows.  DESC_DEBUG_ROW_CAP (tests only) shrinks it -- never below the longest row -- so that a small graph is cut into
// many bands: piece boundaries, the ranks' whole-band ranges and the exchange layout of world = 8 are then exercised at oracle sizes.
int band_row_cap(int max_deg) {
    const int v = env_int("DESC_DEBUG_ROW_CAP", 0);
    return v > 0 ? std::min(BAND_ROW_CAP, std::max(v, std::max(max_deg, 2))) : BAND_ROW_CAP;
}


}  // namespace desc
