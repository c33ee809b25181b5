// Host-side plan of the NODE layout of the DESC_PGD hot path (DESC_PGD.m:182-261 on the device, csrc/pgd.hip): the band-major order of the
// edges with cycles, the chunks of k_sweep_node, the ranks' ranges, and the work lists ("pieces") of k_sweep_band.  Host only: no device call.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <thread>

#include "common.h"

namespace desc {

constexpr int CHUNK_LDS = 960;             // entries of the LDS image of a chunk: one 16-byte vector of w per thread
constexpr int CHUNK_CAP = CHUNK_LDS - 4;   // cycles per chunk: the image starts up to 3 entries before the chunk (16-byte alignment)
constexpr int MAX_SEG_CYCLES = 256;        // longest segment the node layout takes (64 lanes x 4 cycles)
constexpr int BAND_ROW_CAP = 19200;        // doubles of LDS for the band rows (150 KiB of the CU's 160 KiB)
constexpr int MAX_TAIL_PIECES = 768;       // shared tail of the band sweep: at most this many queued pieces (their partials: SHARD_PARTS - grid)
struct alignas(16) PieceDesc { int32_t row_lo, row_len, seg_lo, seg_hi; };   // CSR slots of the band, device-order segments

inline int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

template <class F>
void host_parallel(int64_t count, F&& body, int64_t grain = 65536) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, count / grain));
    if (nt <= 1) { body(0, count); return; }
    run_threads(nt, [&](int t) { body(count * t / nt, count * (t + 1) / nt); });
}

// Host-side plan of the node layout: band-major order of the edges with cycles, chunking,
// and the chunk ranges of the `world` ranks (contiguous, equal numbers of chunks).
struct NodePlan {
    int band = 0;                     // nodes per band (fixed-size bands), 0: LDS-sized bands
    hvec<int32_t> band_lo;     // nbands+1: first node of every band
    hvec<int64_t> bstart;      // nbands+1: first device position (= position in pos_edge) of every band
    hvec<int32_t> rowptr;      // n+1: CSR row starts (degrees prefix-summed)
    hvec<int32_t> order;       // device position -> index into s->pos_edge
    hvec<int32_t> cum2;        // m_pos+1, device order, global cycle numbering
    hvec<int32_t> chunk_seg;   // nchunks+1
    hvec<int64_t> rank_chunk;  // world+1
    hvec<int32_t> rank_node;   // world+1: rank r owns the nodes [rank_node[r], rank_node[r+1]) -- whole bands -- and the segments whose smaller endpoint they are
    // Exchange parts (round 4): every rank's node range is cut once more into `xparts` sub-ranges of whole bands with about equal cycles.  The pair
    // (rank r, part c) is a virtual owner vo = r * xparts + c of the exchange layout: the reduce-scatter of the mirror sums runs part by part, part
    // c + 1 travels while part c is swept.  vnode / vseg: first node / first device-order segment of every virtual owner, + the end.
    int xparts = 1;
    hvec<int32_t> vnode;       // world * xparts + 1
    hvec<int64_t> vseg;        // world * xparts + 1
};

// row_cap > 0: bands = maximal runs of consecutive nodes whose CSR rows hold <= row_cap entries together (the band
// sweep keeps them in the LDS); row_cap == 0: bands of a fixed number of nodes sized for the L2 (k_sweep_node).
int make_node_plan(const desc_problem* prob, const desc_structure* s, int max_deg, int world, int max_seg, int row_cap, NodePlan& P, int xparts = 1);
// the work of every workgroup of the band sweep as a list of pieces (see node_plan.cpp)
void plan_band_pieces(const desc_problem* prob, const desc_structure* s, const NodePlan& P, int64_t seg_lo, int64_t seg_hi, int64_t cyc_lo, int64_t mcl,
                      int G, hvec<PieceDesc>& pieces, hvec<int32_t>& piece_ptr, int& band_rows, bool& jmajor_out, int* tail_first_out = nullptr, int* n_tail_out = nullptr,
                      int max_tail = MAX_TAIL_PIECES);
// LDS budget of a band's rows (DESC_DEBUG_ROW_CAP shrinks it for tests)
int band_row_cap(int max_deg);

}  // namespace desc
