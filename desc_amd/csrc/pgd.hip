// HIP kernels and solver handle for the DESC projected-gradient hot path on gfx950.
//
// Reference text reproduced (Algorithms/DESC_PGD.m, identical in DESC.m:16-261):
//   :129-147  cycle inconsistency S0_long = |acos((tr(Rij Rjk Rki)-1)/2)|/pi
//   :148-157  wijk = 1/cnt, S_vec = 1 / segment mean of S0
//   :185-191  mirror-weight sums (scalar per edge, broadcast to masked positions)
//   :193-207  gradient, tangent projection, plugin step (Utils/ConstantStepSize.m:9-11,
//             PiecewiseStepSize.m:13-18, HybridGradient.m:23-41)
//   :208-230  per-edge simplex projection, new S_vec
//   :232-257  average_change, objective, early stop (evaluated on the device)
//
// Two layouts of the same arithmetic, three sweep kernels (SWEEP VARIANTS):
//
//  NODE layout (default).  Edges with cycles are stored band-major: nodes are cut into bands
//  of B consecutive ids and the edges (i,j) of a band are ordered by (j,i), so that a
//  run of consecutive segments shares j (its row of S stays in the CU's L1) while the
//  band's i-rows stay in the XCD's L2.  S_vec is kept CSR-aligned (`Sfull`, every edge
//  value stored in both endpoint rows) so S({j,k}) = Sfull[rowptr[j] + idx_j(k)] is a
//  gather inside one contiguous row.  Per cycle one packed word holds idx_i(k), idx_j(k)
//  and the two mirror-present bits (DESC_PGD.m:113,124).  The mirror sums are column
//  sums of per-node weight matrices: T1(i,j) = sum_k w(ik;j) = column j of node i,
//  T2(i,j) = column i of node j.  k_colsum_node streams, per endpoint, the cycles of every
//  incident segment whose mirror was sampled and accumulates the columns in LDS (per-wave
//  private copies, fixed order -> bitwise reproducible); k_sweep_node then needs no gather
//  of w at all.
//    HBM traffic per cycle and iteration: sweep 28 B (w r/w 16, S0 8, packed word 4)
//    + column sums (w 8 + a 2-byte column index for the mirrored cycles only), plus the row
//    gathers of S served by L1/L2.
//    Two sweeps run on this layout.  k_sweep_band (default since round 2, segments <= 256 cycles): bands sized so that their
//    CSR rows fit the LDS of a CU, one workgroup per CU (8 waves; shapes: band_shape()), S({k,i}) from the LDS,
//    register-pipelined waves, work dealt in j-block-major units so that the rows of S({j,k}) stay in the L2s; the Adam plugin on
//    its own instances up to 64 cycles.  k_sweep_node (round 1; tiny graphs, rows beyond the LDS, Adam on longer segments):
//    L2-sized bands, 512-thread workgroups, chunks staged through the LDS.
//
//  GATHER (fallback: max degree >= 32768, more LDS than a workgroup may hold, or
//  segments longer than 256 cycles).  Natural edge order; per cycle e_jk, e_ki, ikj, jki
//  and element-granular gathers S[e_jk], S[e_ki], w[ikj], w[jki]  (72 B per cycle as
//  SURVEY.md 8d counts them).
//
// The objective of iteration t needs S_vec of every edge after iteration t, so it
// cannot be fused into sweep t; it is accumulated for free inside sweep t+1 (which
// gathers exactly those values) and once more by an objective kernel after the last
// sweep.  The early-stop test of iteration t is therefore evaluated on the device
// during sweep t+1; when it fires, sweep t+1's output is discarded (the Jacobi double
// buffers still hold iteration t) and later launches return at once.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "device_utils.h"
#include "node_plan.h"

namespace desc {

struct DevState {
    int32_t stop;          // 1 once the patience rule fired
    int32_t misses;        // DESC_PGD.m:181
    int32_t iters_run;     // iteration at which the loop broke
    int32_t final_parity;  // which double buffer holds the final iterate
    int32_t last_tested;   // last iteration whose stop test (:243-256) has been applied: a download between
                           // iterations evaluates the objective early, the next sweep must not count it again
};

struct StepArgs {
    const double* adam_m;                 // HybridGradient.m_t / v_t before this GetStep call ...
    const double* adam_v;
    double* adam_m_out;                   // ... and after it: double-buffered like w, because the stop decision for
    double* adam_v_out;                   // iteration t falls during sweep t+1, whose update must be discardable
    double step;                          // step size of this GetStep call
    double lr, beta1, beta2, bc1, bc2;    // Adam (HybridGradient.m:28-35)
};

__device__ __forceinline__ double abs_acos_ext(double x) {
    // MATLAB abs(acos(x)) with the complex extension outside [-1,1] (DESC_PGD.m:147)
    if (x > 1.0) return acosh(x);
    if (x < -1.0) return hypot(M_PI, acosh(-x));
    return acos(x);
}

// The mirror-weight sums travel as 64-bit fixed-point integers (k_colsum_node: order-independent adds; sharded runs reduce-scatter them as
// ncclInt64, so the totals -- and with them S_vec and the weights -- are bitwise the same for every number of ranks).  `t` holds the bits.
__device__ __forceinline__ double fx_to_double(double t, double fx_inv) { return (double)__double_as_longlong(t) * fx_inv; }

template <int STEP>
__device__ __forceinline__ double apply_step(const StepArgs& a, double w, double g, int64_t c) {
    if (STEP == DESC_STEP_HYBRID) {               // HybridGradient.m:28-35 (strategy 0)
        double mt = (a.beta1 * a.adam_m[c]) + (1.0 - a.beta1) * g;
        double vt = (a.beta2 * a.adam_v[c]) + (1.0 - a.beta2) * (g * g);
        a.adam_m_out[c] = mt; a.adam_v_out[c] = vt;
        double cm = mt / a.bc1, cv = vt / a.bc2;
        return w + (-a.lr * cm / (sqrt(cv) + 1e-8));
    }
    return w + (-a.step * g);                     // ConstantStepSize.m:10 / PiecewiseStepSize.m:17
}

// Simplex projection threshold (DESC_PGD.m:215-223) for one segment held one value per
// lane in a group of G lanes: T with sum(max(ws - T, 0)) = 1.  Michelot's fixed point
// reaches the same active set as the reference's sort-and-scan (the first sorted i with
// sum(w(i:end)-w(i)) < 1).
template <int G>
__device__ __forceinline__ double simplex_threshold(double ws, bool act0, int lane) {
    bool act = act0;
    double T = 0.0;
    for (;;) {
        const double s = group_sum<G>(act ? ws : 0.0);
        const int na = group_count<G>(act, lane);
        T = (s - 1.0) / (double)max(na, 1);
        const bool keep = act && (ws > T);
        const bool changed = keep != act;
        act = keep;
        if (!__any(changed)) break;
    }
    return T;
}

// product-of-three trace in the reference's accumulation order (DESC_PGD.m:137-146)
__device__ __forceinline__ double cycle_trace(const double* A, const double* pb, bool tb, const double* pc, bool tc) {
    double Bm[9], Cm[9], B[9], C[9];
    load_block9(pb, Bm); load_block9(pc, Cm);
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) {
            B[r + 3 * s] = tb ? Bm[s + 3 * r] : Bm[r + 3 * s];
            C[r + 3 * s] = tc ? Cm[s + 3 * r] : Cm[r + 3 * s];
        }
    double tr = 0.0;
    for (int r = 0; r < 3; ++r) {
        double P[3];
        for (int s = 0; s < 3; ++s) {
            double acc = 0.0;
            for (int u = 0; u < 3; ++u) acc = acc + A[r + 3 * u] * B[u + 3 * s];
            P[s] = acc;
        }
        double acc = 0.0;
        for (int u = 0; u < 3; ++u) acc = acc + P[u] * C[u + 3 * r];
        tr = tr + acc;
    }
    return tr;
}

// the same with the third factor already loaded (k_layout_node_dev<.., STAGED>: node i's blocks sit in the LDS)
__device__ __forceinline__ double cycle_trace_regs(const double* A, const double* pb, bool tb, const double* Cm, bool tc) {
    double Bm[9], B[9], C[9];
    load_block9(pb, Bm);
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) {
            B[r + 3 * s] = tb ? Bm[s + 3 * r] : Bm[r + 3 * s];
            C[r + 3 * s] = tc ? Cm[s + 3 * r] : Cm[r + 3 * s];
        }
    double tr = 0.0;
    for (int r = 0; r < 3; ++r) {
        double P[3];
        for (int s = 0; s < 3; ++s) {
            double acc = 0.0;
            for (int u = 0; u < 3; ++u) acc = acc + A[r + 3 * u] * B[u + 3 * s];
            P[s] = acc;
        }
        double acc = 0.0;
        for (int u = 0; u < 3; ++u) acc = acc + P[u] * C[u + 3 * r];
        tr = tr + acc;
    }
    return tr;
}

// deterministic workgroup partials (blockDim 256 or 512)
__device__ __forceinline__ void block_partials(double obj_acc, double chg_acc, double* partials, int lb) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    obj_acc = group_sum<64>(obj_acc);
    chg_acc = group_sum<64>(chg_acc);
    __shared__ double sh[16];
    if (lane == 0) { sh[wv] = obj_acc; sh[8 + wv] = chg_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double o = 0.0, c = 0.0;
        for (int k = 0; k < nw; ++k) { o += sh[k]; c += sh[8 + k]; }
        partials[2 * lb] = o;
        partials[2 * lb + 1] = c;
    }
}

// ===========================================================================
// GATHER variant
// ===========================================================================
struct SweepArgs {
    const int32_t* cum;       // m_pos+1
    const int32_t* pos_edge;  // m_pos
    const int32_t* e_jk;
    const int32_t* e_ki;
    const int32_t* ikj;
    const int32_t* jki;
    const double* S0;
    const double* w_old;
    double* w_new;
    const double* S_old;
    double* S_new;
    const double* nv_tab;     // nv_tab[c] = 1/sqrt(c)   (DESC_PGD.m:199)
    double* partials;         // [grid][2]: objective of the old iterate, sum |dS|
    const DevState* state;
    StepArgs st;
    int32_t m_pos;
};

template <int G, int STEP>
__global__ __launch_bounds__(256) void k_sweep(SweepArgs a) {
    if (a.state->stop) return;
    constexpr int EPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int sub = lane / G, gl = lane % G;
    const int nb = gridDim.x;
    const int lb = xcd_logical_block(blockIdx.x, nb);
    const int per_block = (a.m_pos + nb - 1) / nb;
    const int lo_edge = lb * per_block;
    const int hi_edge = min(a.m_pos, lo_edge + per_block);

    double obj_acc = 0.0, chg_acc = 0.0;
    for (int l0 = lo_edge + wv * EPW; l0 < hi_edge; l0 += 4 * EPW) {
        const int l = l0 + sub;
        const bool edge_ok = l < hi_edge;
        int base = 0, cnt = 0;
        if (edge_ok) { base = a.cum[l]; cnt = a.cum[l + 1] - base; }
        const bool act0 = gl < cnt;
        const int64_t c = (int64_t)base + gl;

        double w = 0.0, d = 0.0, ssum = 0.0, wa = 0.0, wb = 0.0;
        int ia = -1, ib = -1;
        if (act0) {
            const int ejk = a.e_jk[c], eki = a.e_ki[c];
            ia = a.ikj[c]; ib = a.jki[c];
            w = a.w_old[c]; d = a.S0[c];
            ssum = a.S_old[ejk] + a.S_old[eki];
            if (ia >= 0) wa = a.w_old[ia];
            if (ib >= 0) wb = a.w_old[ib];
        }
        obj_acc += w * ssum;                       // objective of the iterate being read (:233, one sweep late)
        // mirror-weight sums: one scalar per edge, applied to masked positions only (:189-190)
        const double T1 = group_sum<G>(wa), T2 = group_sum<G>(wb);
        double g = ssum + ((ia >= 0 ? T1 : 0.0) + (ib >= 0 ? T2 : 0.0)) * d;          // :193
        // tangent projection grad - (grad*nv')*nv, nv = ones/sqrt(cnt)  (:199-201)
        const double nv = act0 ? a.nv_tab[cnt] : 0.0;
        const double dot = group_sum<G>(act0 ? g * nv : 0.0);
        g = g - dot * nv;
        const double ws = act0 ? apply_step<STEP>(a.st, w, g, c) : 0.0;               // :207
        const double T = simplex_threshold<G>(ws, act0, lane);          // :215-223
        const double wn = act0 ? fmax(ws - T, 0.0) : 0.0;                             // :224
        const double snew = group_sum<G>(wn * d);                                     // :229
        if (act0) a.w_new[c] = wn;
        if (edge_ok && gl == 0) {
            const int e = a.pos_edge[l];
            chg_acc += fabs(snew - a.S_old[e]);                                       // :232
            a.S_new[e] = snew;
        }
    }
    block_partials(obj_acc, chg_acc, a.partials, lb);
}

// Segments longer than 64 cycles: one wave per edge, several passes over the segment;
// w_new doubles as scratch (each lane re-reads only what it wrote itself).
template <int STEP>
__global__ __launch_bounds__(256) void k_sweep_big(SweepArgs a) {
    if (a.state->stop) return;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int nb = gridDim.x;
    const int lb = xcd_logical_block(blockIdx.x, nb);
    const int per_block = (a.m_pos + nb - 1) / nb;
    const int lo_edge = lb * per_block;
    const int hi_edge = min(a.m_pos, lo_edge + per_block);

    double obj_acc = 0.0, chg_acc = 0.0;
    for (int l = lo_edge + wv; l < hi_edge; l += 4) {
        const int base = a.cum[l], cnt = a.cum[l + 1] - base;
        double t1 = 0.0, t2 = 0.0;
        for (int t = lane; t < cnt; t += 64) {
            const int64_t c = (int64_t)base + t;
            const int ia = a.ikj[c], ib = a.jki[c];
            if (ia >= 0) t1 += a.w_old[ia];
            if (ib >= 0) t2 += a.w_old[ib];
        }
        const double T1 = group_sum<64>(t1), T2 = group_sum<64>(t2);
        const double nv = a.nv_tab[cnt];
        double dotp = 0.0;
        for (int t = lane; t < cnt; t += 64) {
            const int64_t c = (int64_t)base + t;
            const double ssum = a.S_old[a.e_jk[c]] + a.S_old[a.e_ki[c]];
            obj_acc += a.w_old[c] * ssum;
            const double g = ssum + ((a.ikj[c] >= 0 ? T1 : 0.0) + (a.jki[c] >= 0 ? T2 : 0.0)) * a.S0[c];
            a.w_new[c] = g;
            dotp += g * nv;
        }
        const double dot = group_sum<64>(dotp);
        for (int t = lane; t < cnt; t += 64) {
            const int64_t c = (int64_t)base + t;
            const double g = a.w_new[c] - dot * nv;
            a.w_new[c] = apply_step<STEP>(a.st, a.w_old[c], g, c);
        }
        double T = -INFINITY;
        int prev_n = -1;
        for (;;) {
            double s = 0.0; int na = 0;
            for (int t = lane; t < cnt; t += 64) {
                const double x = a.w_new[(int64_t)base + t];
                if (x > T) { s += x; ++na; }
            }
            s = group_sum<64>(s);
            na = (int)group_sum<64>((double)na);
            if (na == prev_n) break;
            prev_n = na;
            T = (s - 1.0) / (double)max(na, 1);
        }
        double sn = 0.0;
        for (int t = lane; t < cnt; t += 64) {
            const int64_t c = (int64_t)base + t;
            const double wn = fmax(a.w_new[c] - T, 0.0);
            a.w_new[c] = wn;
            sn += wn * a.S0[c];
        }
        const double snew = group_sum<64>(sn);
        if (lane == 0) {
            const int e = a.pos_edge[l];
            chg_acc += fabs(snew - a.S_old[e]);
            a.S_new[e] = snew;
        }
    }
    block_partials(obj_acc, chg_acc, a.partials, lb);
}

__global__ __launch_bounds__(256) void k_objective(const double* w, const double* S, const int32_t* e_jk,
                                                   const int32_t* e_ki, int64_t m_cycle, double* partials,
                                                   const DevState* st) {
    if (st->stop) return;
    double acc = 0.0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < m_cycle; c += (int64_t)gridDim.x * 256)
        acc += w[c] * (S[e_jk[c]] + S[e_ki[c]]);
    block_partials(acc, 0.0, partials, blockIdx.x);
}

// wijk = 1/cnt, S_vec(IJ) = wijk_seg * S0_seg'  (DESC_PGD.m:151-157); one wave per edge
__global__ __launch_bounds__(256) void k_init(const int32_t* cum, const int32_t* pos_edge, const double* S0,
                                              double* w, double* S_a, double* S_b, int m_pos) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int base = cum[l], cnt = cum[l + 1] - base;
        const double w0 = 1.0 / (double)cnt;
        double s = 0.0;
        for (int t = lane; t < cnt; t += 64) { w[(int64_t)base + t] = w0; s += w0 * S0[(int64_t)base + t]; }
        s = group_sum<64>(s);
        if (lane == 0) { S_a[pos_edge[l]] = s; S_b[pos_edge[l]] = s; }
    }
}

// Cycle inconsistency (DESC_PGD.m:129-147): one wave per edge, lanes over its cycles.
// R_jk = RijMat4d(:,:,j,k) is the stored block of edge {j,k} if j<k, its transpose
// otherwise; likewise R_ki (:65-66,89-91).
__global__ __launch_bounds__(256) void k_cycle_d(const int32_t* cum, const int32_t* pos_edge, const int32_t* ind_i,
                                                 const int32_t* ind_j, const int32_t* kk, const int32_t* e_jk,
                                                 const int32_t* e_ki, const double* rij, double* S0, int m_pos) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int base = cum[l], cnt = cum[l + 1] - base;
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        double A[9];
        for (int t = 0; t < 9; ++t) A[t] = rij[9 * (int64_t)e + t];
        for (int q = lane; q < cnt; q += 64) {
            const int64_t c = (int64_t)base + q;
            const int k = kk[c];
            const double tr = cycle_trace(A, rij + 9 * (int64_t)e_jk[c], !(j < k), rij + 9 * (int64_t)e_ki[c], !(k < i));
            S0[c] = abs_acos_ext((tr - 1.0) / 2.0) / M_PI;
        }
    }
}

// ===========================================================================
// NODE variant
// ===========================================================================
// per edge-with-cycles, in device (band-major) order
struct EdgeInfo {
    int32_t rb_i;     // rowptr[i]
    int32_t rb_j;     // rowptr[j]
    int32_t slot_a;   // rowptr[i] + idx_i(j): this edge's slot in row i of Sfull / Tfull
    int32_t slot_b;   // rowptr[j] + idx_j(i): this edge's slot in row j
};
// packed per-cycle word: bits 0-14 idx_i(k), bit 15 cycle (ik;j) sampled (IKJ_appears, :113),
//                        bits 16-30 idx_j(k), bit 31 cycle (jk;i) sampled (JKI_appears, :124)


constexpr int SWEEP_THREADS = 512;

// first / end cycle and first / end segment of a chunk, one 16-byte scalar load
struct alignas(16) ChunkDesc { int32_t c0, c1, l0, l1; };
__device__ __forceinline__ ChunkDesc uniform_load_desc(const ChunkDesc* p, int i) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef const v4i __attribute__((address_space(4)))* cptr_t;
    const v4i v = ((cptr_t)(unsigned long long)p)[i];
    return ChunkDesc{v.x, v.y, v.z, v.w};
}

struct NodeSweepArgs {
    const int32_t* cum;        // m_pos+1, device order
    const EdgeInfo* einfo;     // m_pos
    const uint32_t* pk;        // m_cycle
    const double* S0;
    const double* w_old;
    double* w_new;
    const double* S_old;       // Sfull, 2m
    double* S_new;
    const double* Tfull;       // 2m: Tfull[rowptr[v]+t] = sum_k w(v,k; nbr_t), masked; sharded runs: T1, T2 per owned segment
    const int2* xt;            // sharded runs (else NULL): per device-order segment {ta, tb}: Tfull (= the reduce-scattered sums of this rank) holds
                               // T1 of the segment at ta, T2 at tb (exchange layout: see k_xpos); ta is also its place in the all-gather slice
    double* s_slice;           // sharded runs: the new S of segment l goes to s_slice[ta] (this rank's all-gather slice)
    const double* nv_tab;
    double* partials;
    const DevState* state;
    const ChunkDesc* chunk_desc;   // nchunks: {c0, c1, l0, l1}
    StepArgs st;
    int32_t nchunks;
    int32_t max_cnt;
    uint32_t csr_bytes;        // bytes of the CSR-aligned arrays (S_old, S_new, Tfull of a one-rank run: 2m doubles) -- num_records of their buffer descriptors
    uint32_t t_bytes;          // bytes of Tfull (sharded runs: this rank's part of the reduce-scattered sums)
    uint32_t slice_bytes;      // bytes of s_slice (sharded runs)
    uint32_t seg_count;        // segments in cum / einfo / xt (device order, all ranks): cum has seg_count + 1 entries
    double fx_inv;             // 2^-fx_bits: Tfull holds fixed-point integers (fx_to_double)
};

struct StreamRegs { uint4 pk; double2 w, d; };   // one 16-byte vector of each streamed array

// LDS image of one chunk.  Entry q' = (cycle - a0) where a0 = c0 & ~3 is the chunk's first
// cycle rounded down to a 16-byte boundary of the packed-word array, so that every
// streaming load is a 16-byte access.
struct alignas(16) ChunkBuf {
    double w[CHUNK_LDS], d[CHUNK_LDS];
    uint32_t pk[CHUNK_LDS];
};

// The sweep (dominant kernel).  512-thread persistent workgroups take chunks of consecutive
// segments (<= CHUNK_CAP cycles, <= 8 * 64/LPS segments) round-robin: at any moment the
// resident workgroups read one contiguous window of the streamed arrays (DRAM row-buffer
// locality; disjoint far-apart streams per workgroup measured ~half the bandwidth).
// pk, w and S0 of a chunk arrive with 16-byte loads, two chunks ahead, and are parked in a
// triple-buffered LDS image.  Everything per segment lives in the registers of the 16 lanes
// that own the segment (LPS lanes x E cycles per lane = 16x1, 16x2, 32x2, 32x4, 64x4 for segments of up to 16, 32, 64, 128, 256 cycles; a
// chunk = one pass of the 8 waves x 64/LPS lane groups): the lane group
// that will compute segment t of chunk c+1 loads its records, issues its row gathers
// (addresses from the packed words already parked in LDS) and keeps their sums until the
// arithmetic of chunk c+1; E cycles per lane, reductions = E in-lane adds + 4 DPP steps
// shared by the 4 segments of a wave; simplex threshold by Michelot's fixed point with a
// division-free comparison (same active set as the reference's sort-and-scan, :215-223);
// weights and S leave with 8-byte stores straight from registers (16 lanes x 8 B = 128
// contiguous bytes).  One workgroup barrier per chunk.
//   iteration c:  park stream(c+1) -> LDS[(c+1)%3] | barrier | gathers(c+1) | records(c+2),
//                 stream(c+3) | arithmetic(c) out of LDS[c%3] + registers | land gathers and
//                 records | stores(c)
// gfx950 retires loads AND stores through one in-order counter (vmcnt): every pipelined
// load is unconditional (clamped index instead of a branch) so the compiler can count
// them, and a load that is waited for after a store point would make the wave wait for the
// stores too, so gathers and records are consumed (landed) after the arithmetic but before
// the stores of the iteration that issued them.  Chunk descriptors are prefetched four
// chunks ahead with one scalar 16-byte load (constant cache: off vmcnt).
// (the Adam variant carries two more streamed values per cycle through the arithmetic: it gets the
//  register budget of 3 waves per SIMD instead of spilling)
template <int LPS, int E, int STEP>
__global__ __launch_bounds__(SWEEP_THREADS, STEP == DESC_STEP_HYBRID ? 3 : 4) void k_sweep_node(NodeSweepArgs a) {
    __shared__ ChunkBuf X[3];
    __shared__ double s_nv[MAX_SEG_CYCLES + 1];
    if (a.state->stop) return;
    constexpr int NT = SWEEP_THREADS;
    static_assert(NT == 512 && (LPS == 16 || LPS == 32 || LPS == 64) && LPS * E <= MAX_SEG_CYCLES, "8 waves x 64/LPS lane groups per chunk");
    static_assert(CHUNK_LDS / 2 <= NT, "one 16-byte vector of w per thread");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int SPW = 64 / LPS;                      // segments per wave: a chunk holds at most 8 * SPW segments
    const int grp = lane / LPS, r16 = lane % LPS;      // LPS lanes per segment, E cycles per lane
    const int ts = wv * SPW + grp;                     // segment slot of this lane group
    const int nb = gridDim.x, lb = blockIdx.x;         // chunks dealt round-robin (see k_sweep_node)
    const int nk = lb < a.nchunks ? (a.nchunks - lb + nb - 1) / nb : 0;
    double obj_acc = 0.0, chg_acc = 0.0;
    if (nk == 0) { block_partials(obj_acc, chg_acc, a.partials, lb); return; }
    if (tid <= MAX_SEG_CYCLES) s_nv[tid] = tid <= a.max_cnt ? a.nv_tab[tid] : 0.0;

    struct RecRaw { int b0, b1; EdgeInfo ei; int2 x; };
    struct Rec { int qb, cnt, rbi, rbj, sa, sb, seg; };   // qb: image entry of the segment's first cycle; sharded runs: seg = ta, sb = tb
    struct Gat { double sjk[E], ski[E], T1, T2, So; };
    struct Landed { double ss[E], T1, T2, So; };       // S(jk)+S(ki) per cycle, mirror sums, old S of the segment
    struct Carry { StreamRegs s; Landed g; Rec r; };

    // past the end every load is redirected to this workgroup's last chunk: harmless and branch-free
    auto desc_of = [&](int k) -> ChunkDesc { return uniform_load_desc(a.chunk_desc, lb + min(k, nk - 1) * nb); };
    auto load_rec = [&](const ChunkDesc d) -> RecRaw {
        const int t = d.l0 + min(ts, d.l1 - d.l0 - 1);
        RecRaw r;
        r.b0 = a.cum[t]; r.b1 = a.cum[t + 1]; r.ei = a.einfo[t];
        r.x = a.xt ? a.xt[t] : int2{0, 0};
        return r;
    };
    auto land_rec = [&](const ChunkDesc d, const RecRaw q) -> Rec {
        Rec r;
        r.qb = q.b0 - (d.c0 & ~3);
        r.cnt = ts < d.l1 - d.l0 ? q.b1 - q.b0 : 0;
        r.rbi = q.ei.rb_i; r.rbj = q.ei.rb_j; r.sa = q.ei.slot_a; r.sb = a.xt ? q.x.y : q.ei.slot_b;
        r.seg = q.x.x;
        return r;
    };
    auto load_stream = [&](const ChunkDesc d) -> StreamRegs {   // 16-byte loads; image entry 0 = cycle a0 = c0 & ~3
        const int a0 = d.c0 & ~3, last = d.c1 - 1 - a0;
        const int v2 = min(tid, last >> 1), v4 = min(tid, last >> 2);
        StreamRegs sr;
        sr.w = *reinterpret_cast<const double2*>(a.w_old + a0 + 2 * (int64_t)v2);
        sr.d = *reinterpret_cast<const double2*>(a.S0 + a0 + 2 * (int64_t)v2);
        sr.pk = *reinterpret_cast<const uint4*>(a.pk + a0 + 4 * (int64_t)v4);
        return sr;
    };
    auto park = [&](const ChunkDesc d, const StreamRegs sr, ChunkBuf& xb) {
        const int last = d.c1 - 1 - (d.c0 & ~3);
        if (tid <= (last >> 1)) { *reinterpret_cast<double2*>(&xb.w[2 * tid]) = sr.w; *reinterpret_cast<double2*>(&xb.d[2 * tid]) = sr.d; }
        if (tid <= (last >> 2)) *reinterpret_cast<uint4*>(&xb.pk[4 * tid]) = sr.pk;
    };
    auto issue_gathers = [&](const Rec r, const ChunkBuf& xb) -> Gat {
        Gat g;
#pragma unroll
        for (int e = 0; e < E; ++e) {                // unconditional: idle lanes repeat the segment's first cycle
            const int idx = r16 + LPS * e;
            const uint32_t p = xb.pk[r.qb + (idx < r.cnt ? idx : 0)];
            const int si = r.rbi + (int)(p & 0x7FFFu), sj = r.rbj + (int)((p >> 16) & 0x7FFFu);
            g.sjk[e] = a.S_old[sj]; g.ski[e] = a.S_old[si];
        }
        const int ta = a.xt ? r.seg : r.sa, tb = r.sb;
        g.T1 = a.Tfull[ta];                      // column j of node i = sum(wijk(IKJ(mask)))  (:189)
        g.T2 = a.Tfull[tb];                      // column i of node j = sum(wijk(JKI(mask)))  (:190)
        g.So = a.S_old[r.sa];
        return g;
    };

    // one pipeline iteration: chunk k of this workgroup is computed; c.s = stream registers of
    // chunk k+1, c.g = gathers of chunk k, c.r = records of chunk k+1; r0 = records of chunk k
    auto iterate = [&](int k, const Carry c, const Rec r0, const ChunkDesc d0, const ChunkDesc d1, const ChunkDesc d2,
                       const ChunkDesc d3) -> Carry {
        Carry o;
        Gat gn;
        ChunkBuf& xc = X[k % 3];
        ChunkBuf& xn = X[(k + 1) % 3];
        park(d1, c.s, xn);
        __syncthreads();
        gn = issue_gathers(c.r, xn);
        const RecRaw q2 = load_rec(d2);
        o.s = load_stream(d3);
        // ---- arithmetic of chunk k
        const int a0 = d0.c0 & ~3;
        const int cnt = r0.cnt;
        const double T1 = fx_to_double(c.g.T1, a.fx_inv), T2 = fx_to_double(c.g.T2, a.fx_inv), nv = cnt > 0 ? s_nv[cnt] : 0.0;
        double ws[E];
        uint32_t okm = 0;
        double part = 0.0;
        // waves whose four lane groups own no segment of this chunk skip the arithmetic (wave-uniform)
        if (__ballot(cnt > 0) != 0ull) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = r16 + LPS * e;
                const bool ok = idx < cnt;
                const int q = r0.qb + idx;
                uint32_t p = 0; double w = 0.0, d = 0.0;
                if (ok) { p = xc.pk[q]; w = xc.w[q]; d = xc.d[q]; okm |= 1u << e; }
                const double ss = c.g.ss[e];                                                         // S(jk)+S(ki)
                obj_acc += w * ss;                                                                   // :233, one sweep late
                const double g = ss + (((p & 0x8000u) ? T1 : 0.0) + ((p & 0x80000000u) ? T2 : 0.0)) * d;   // :193
                ws[e] = g;
                part += ok ? g * nv : 0.0;
            }
            const double dot = group_sum<LPS>(part);                                                    // :199-201
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int q = r0.qb + r16 + LPS * e;
                const double g = ws[e] - dot * nv;
                ws[e] = ((okm >> e) & 1u) ? apply_step<STEP>(a.st, xc.w[q], g, (int64_t)a0 + q) : 0.0;   // :207
            }
            // simplex projection threshold (:215-223), Michelot fixed point; the comparison
            // ws*na > s-1 is ws > (s-1)/na without a division per pass
            uint32_t act = okm;
            double s1v = 0.0; int na = 1;
            for (;;) {
                double sp = 0.0;
#pragma unroll
                for (int e = 0; e < E; ++e) sp += ((act >> e) & 1u) ? ws[e] : 0.0;
                s1v = group_sum<LPS>(sp) - 1.0;
                if (LPS == 16) na = max(group16_sum((int)__popc(act)), 1);
                else {                                 // wider groups: count by ballots (scalar popcounts)
                    na = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) na += group_count<LPS>((act >> e) & 1u, lane);
                    na = max(na, 1);
                }
                const double nad = (double)na;
                uint32_t keep = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) keep |= (((act >> e) & 1u) && ws[e] * nad > s1v) ? 1u << e : 0u;
                const bool changed = keep != act;
                act = keep;
                if (!__any(changed)) break;
            }
            const double T = s1v / (double)na;
            double sn = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double wn = fmax(ws[e] - T, 0.0);                                              // :224
                ws[e] = wn;
                if ((okm >> e) & 1u) sn += wn * xc.d[r0.qb + r16 + LPS * e];
            }
            part = group_sum<LPS>(sn);                                                                  // :229
        }
        // gathers of chunk k+1 and records of chunk k+2 are consumed before this iteration's stores
        // (see the header); the gathers shrink to their sums
#pragma unroll
        for (int e = 0; e < E; ++e) o.g.ss[e] = gn.sjk[e] + gn.ski[e];
        o.g.T1 = gn.T1; o.g.T2 = gn.T2; o.g.So = gn.So;
        o.r = land_rec(d2, q2);
        // pin the landing here: without it the scheduler sinks these adds below the stores
#pragma unroll
        for (int e = 0; e < E; ++e) asm volatile("" : "+v"(o.g.ss[e]));
        asm volatile("" : "+v"(o.g.T1), "+v"(o.g.T2), "+v"(o.g.So));
        asm volatile("" : "+v"(o.r.qb), "+v"(o.r.cnt), "+v"(o.r.rbi), "+v"(o.r.rbj), "+v"(o.r.sa), "+v"(o.r.sb), "+v"(o.r.seg));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((okm >> e) & 1u) a.w_new[(int64_t)a0 + r0.qb + r16 + LPS * e] = ws[e];
        if (cnt > 0 && r16 == 0) {
            chg_acc += fabs(part - c.g.So);                                                      // :232
            // sharded runs: only the all-gather slice is written; k_unpack_S scatters S of every edge, this rank's included
            if (a.s_slice) a.s_slice[r0.seg] = part;
            else { a.S_new[r0.sa] = part; a.S_new[r0.sb] = part; }
        }
        return o;
    };

    // ---- prologue: chunk 0 parked and its gathers in flight, streams of chunks 1 and 2 in flight
    ChunkDesc D0 = desc_of(0), D1 = desc_of(1), D2 = desc_of(2), D3 = desc_of(3);
    Rec R0;
    Carry CA, CB;
    {
        const RecRaw q0 = load_rec(D0);
        const StreamRegs s0 = load_stream(D0);
        const RecRaw q1 = load_rec(D1);
        CA.s = load_stream(D1);                  // CA: consumed by even iterations
        CB.s = load_stream(D2);
        R0 = land_rec(D0, q0);
        park(D0, s0, X[0]);
        __syncthreads();
        const Gat g0 = issue_gathers(R0, X[0]);
#pragma unroll
        for (int e = 0; e < E; ++e) CA.g.ss[e] = g0.sjk[e] + g0.ski[e];
        CA.g.T1 = g0.T1; CA.g.T2 = g0.T2; CA.g.So = g0.So;
        CA.r = land_rec(D1, q1);
    }
    // Two iterations per trip (the register sets alternate).  The second one always runs -- with
    // an odd number of chunks it computes nothing (cnt = 0) -- so that the compiler sees one
    // straight-line loop body and can count the loads in flight across the back edge.
    for (int k = 0; k < nk; k += 2) {
        {
            const ChunkDesc Dn = desc_of(k + 4);
            const Carry o = iterate(k, CA, R0, D0, D1, D2, D3);       // parks CA.s (chunk k+1), consumes CA.g (chunk k)
            R0 = CA.r; CB.g = o.g; CB.r = o.r; CA.s = o.s;            // CA.s now streams chunk k+3
            D0 = D1; D1 = D2; D2 = D3; D3 = Dn;
        }
        {
            if (k + 1 >= nk) R0.cnt = 0;
            const ChunkDesc Dn = desc_of(k + 5);
            const Carry o = iterate(k + 1, Carry{CB.s, CB.g, CB.r}, R0, D0, D1, D2, D3);
            R0 = CB.r; CA.g = o.g; CA.r = o.r; CB.s = o.s;
            D0 = D1; D1 = D2; D2 = D3; D3 = Dn;
        }
    }
    block_partials(obj_acc, chg_acc, a.partials, lb);
}

// SMALL graphs (round 4; the reference's own demo sizes, Demo/compare_algorithms.m:10: n = 100-200, configs[0]): an iteration of a few hundred
// thousand cycles is not bandwidth- but latency-bound -- k_sweep_node's staged chunk pipeline is a chain of ~6 dependent memory round trips for one
// chunk per workgroup (C1: 11.7 us per sweep for ~1 us of traffic).  Here a lane group of G lanes takes a whole segment (one cycle per lane) straight
// from global memory: records -> packed words / weights / S0 -> the gathers of S and the mirror sums -> arithmetic -> stores, three dependent round
// trips.  Same per-segment arithmetic as the gather layout's k_sweep (group reductions over G lanes, Michelot threshold), on the node layout's arrays.
template <int G, int STEP>
__global__ __launch_bounds__(256) void k_sweep_small(NodeSweepArgs a, int n_seg) {
    if (a.state->stop) return;
    constexpr int EPW = 64 / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane / G, gl = lane % G;
    double obj_acc = 0.0, chg_acc = 0.0;
    for (int l0 = ((int)blockIdx.x * 4 + wv) * EPW; l0 < n_seg; l0 += (int)gridDim.x * 4 * EPW) {
        const int l = l0 + sub;
        const bool seg_ok = l < n_seg;
        int base = 0, cnt = 0;
        EdgeInfo ei{0, 0, 0, 0};
        if (seg_ok) { base = a.cum[l]; cnt = a.cum[l + 1] - base; ei = a.einfo[l]; }
        const bool act0 = gl < cnt;
        const int64_t c = (int64_t)base + gl;
        uint32_t p = 0; double w = 0.0, d = 0.0, ssum = 0.0;
        if (act0) {
            p = a.pk[c]; w = a.w_old[c]; d = a.S0[c];
            ssum = a.S_old[ei.rb_j + (int)((p >> 16) & 0x7FFFu)] + a.S_old[ei.rb_i + (int)(p & 0x7FFFu)];          // S(jk) + S(ki)
        }
        double T1 = 0.0, T2 = 0.0, So = 0.0;
        if (seg_ok) { T1 = fx_to_double(a.Tfull[ei.slot_a], a.fx_inv); T2 = fx_to_double(a.Tfull[ei.slot_b], a.fx_inv); So = a.S_old[ei.slot_a]; }
        obj_acc += w * ssum;                                                                                     // :233, one sweep late
        double g = ssum + (((p & 0x8000u) ? T1 : 0.0) + ((p & 0x80000000u) ? T2 : 0.0)) * d;                     // :193
        const double nv = act0 ? a.nv_tab[cnt] : 0.0;
        const double dot = group_sum<G>(act0 ? g * nv : 0.0);                                                    // :199-201
        g = g - dot * nv;
        const double ws = act0 ? apply_step<STEP>(a.st, w, g, c) : 0.0;                                          // :207
        const double T = simplex_threshold<G>(ws, act0, lane);                                                   // :215-223
        const double wn = act0 ? fmax(ws - T, 0.0) : 0.0;                                                        // :224
        const double snew = group_sum<G>(wn * d);                                                                // :229
        if (act0) a.w_new[c] = wn;
        if (seg_ok && cnt > 0 && gl == 0) {
            chg_acc += fabs(snew - So);                                                                          // :232
            a.S_new[ei.slot_a] = snew; a.S_new[ei.slot_b] = snew;
        }
    }
    block_partials(obj_acc, chg_acc, a.partials, blockIdx.x);
}

// ===========================================================================
// BAND sweep: the same arithmetic with the i-rows of S in the LDS
// ===========================================================================
// k_sweep_node is bound by L1-miss sector traffic: every segment gathers ~cnt values out of row i of Sfull
// (8 useful bytes per 64-byte sector, a different row for every segment of a chunk).  Here the nodes are cut
// into bands whose CSR rows fit the LDS of one CU together (<= row_cap doubles: 36 rows at C2, 18 at C4 / C5);
// the edges keep the (band(i), j, i) order, every workgroup owns ONE contiguous range of segments (equal cycle
// counts; pure streaming measured no penalty for contiguous ranges -- tools/probes/stream_probe.hip), split into
// "pieces" at band boundaries, loads the piece's band rows once (coalesced) and serves S({k,i}) and the segment's
// old S from the LDS.  S({j,k}) stays a gather through L1: consecutive segments of a band share j.
// 1024 threads = 16 waves per workgroup (one workgroup per CU); no barrier inside a piece.  A wave takes
// SPW = 64/LPS consecutive segments per iteration; its software pipeline keeps, per lane group,
//   records (cum, EdgeInfo: scalar loads, off vmcnt)  4 iterations ahead,
//   the streamed pk / w / S0 of its cycles            3 iterations ahead (registers, 4 rotating sets),
//   the gathers of S({j,k}), T1, T2 (+ LDS reads)     1 iteration ahead (2 sets),
// issued in the order gathers -> stream -> arithmetic -> stores: loads and stores retire through one in-order
// counter on gfx950, so everything an iteration waits for was issued before the previous iteration's stores.
// BUFFER instructions (round 3).  The band sweep and the column sums address global memory as `descriptor in SGPRs + 32-bit byte offset per
// lane` (buffer_load / buffer_store ... offen) instead of a 64-bit address per lane (global_load / global_store): half the address data per
// instruction on the way to the address unit -- the busiest unit of the CU in this kernel (TA_BUSY 76 % at C4, section 5 of DESIGN.md) -- and no
// 64-bit address arithmetic (v_lshl_add_u64, sign extensions: 12 % of the VALU instructions of an iteration).  Measured, rocprofv3 averages in
// one call (profiles/r03_buffer_instructions.txt): only the S({j,k}) gathers C4 1171 -> 1119 us; + T1/T2 and the S stores 1122; + the three
// streams and the weight stores 1084-1091 (-7.2 %); C5 1737 -> 1670-1682, C3 105.0 -> 100.4, C2 110.8 -> 107.8.  Cache-policy bits on the
// gathers (same file): nt 1873 us, sc1 / sc0 sc1 1253 us -- no.  Offsets are 32 bits: the descriptors of the per-cycle arrays are rebuilt per
// piece with the piece's first cycle as base (any m_cycle), the CSR-aligned arrays (2m doubles) must stay below 4 GiB (setup_node: else no band sweep).
// Out-of-range offsets read 0 and store nothing (num_records) instead of faulting.  (The round-2 global-load forms are in the history, tree ca0abdf.)
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const T* p, uint32_t bytes = 0xFFFFFFFFu) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);          // raw buffer (stride 0), 32-bit data format
}
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    double d; __builtin_memcpy(&d, &v, 8); return d;
}
__device__ __forceinline__ uint32_t buf_load_u32(__amdgpu_buffer_rsrc_t rs, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0); }
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_load_f64x2(__amdgpu_buffer_rsrc_t rs, uint32_t off, double& a0, double& a1) {      // 16 bytes, 8-byte aligned
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    const u32x2_t lo = {v.x, v.y}, hi = {v.z, v.w};
    __builtin_memcpy(&a0, &lo, 8); __builtin_memcpy(&a1, &hi, 8);
}
__device__ __forceinline__ void buf_load_u32x2(__amdgpu_buffer_rsrc_t rs, uint32_t off, uint32_t& a0, uint32_t& a1) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    a0 = v.x; a1 = v.y;
}
__device__ __forceinline__ void buf_store_f64x2(__amdgpu_buffer_rsrc_t rs, uint32_t off, double d0, double d1) {
    u32x2_t lo, hi; __builtin_memcpy(&lo, &d0, 8); __builtin_memcpy(&hi, &d1, 8);
    const u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t rs, uint32_t off, double d) {
    u32x2_t v; __builtin_memcpy(&v, &d, 8);
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off, 0, 0);
}

struct BandSweepArgs {
    NodeSweepArgs n;
    const PieceDesc* pieces;
    const int32_t* piece_ptr;      // grid + 1
    int32_t row_cap;
    unsigned long long* wg_clock;  // diagnostics (DESC_DEBUG_WGCLOCK=1; else NULL): per workgroup {start, end} of the constant 100 MHz clock
    // Dynamic tail (round 3): the last few per cent of the sweep's work are not in any workgroup's list but in a shared queue of small pieces
    // pieces[tail_first .. tail_first + n_tail), handed out by tickets to whichever workgroup has finished its list (the lists are balanced by
    // a cost model; the workgroups' real times differ by +-5 %).  Every tail piece has its own pair of partials (slot gridDim.x + ticket), so
    // the objective and |dS| sums do not depend on who took which piece.  *ticket is zeroed by the column-sum launch that precedes a sweep.
    int32_t tail_first, n_tail;
    int32_t* ticket;
};

__device__ __forceinline__ PieceDesc uniform_load_piece(const PieceDesc* p, int i) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef const v4i __attribute__((address_space(4)))* cptr_t;
    const v4i v = ((cptr_t)(unsigned long long)p)[i];
    return PieceDesc{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ EdgeInfo uniform_load_einfo(const EdgeInfo* p, int i) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef const v4i __attribute__((address_space(4)))* cptr_t;
    const v4i v = ((cptr_t)(unsigned long long)p)[i];
    return EdgeInfo{v.x, v.y, v.z, v.w};
}

template <int LPS, int E, int STEP, int NT, bool XT = false>      // XT: sharded run (exchange positions {ta, tb} per segment; its own instances keep
__global__ __launch_bounds__(NT, 1) void k_sweep_band(BandSweepArgs b) {     // the extra records out of the one-GPU kernels' scalar registers)
    extern __shared__ double s_dyn[];                  // [row_cap] band rows of S_old, then the nv table
    const NodeSweepArgs& a = b.n;
    double* s_rows = s_dyn;
    double* s_nv = s_dyn + b.row_cap;
    __shared__ double s_part[2][NT / 64];
    // segments of up to 64 cycles: 1024 threads (16 waves, <= 128 VGPRs each); up to 256 cycles (4 cycles per lane: twice the
    // registers per pipeline stage) and the Adam plugin (its two moments are two more streams in and out): 512 threads (8 waves, <= 256 VGPRs).
    static_assert((LPS == 8 || LPS == 16 || LPS == 32 || LPS == 64) && LPS * E <= MAX_SEG_CYCLES && (NT == 512 || NT == 1024) && (STEP != DESC_STEP_HYBRID || NT == 512), "band sweep instances");
    constexpr bool ADAM = STEP == DESC_STEP_HYBRID;
    constexpr int EA = ADAM ? E : 1;
    if (a.state->stop) return;
    constexpr int NW = NT / 64, SPW = 64 / LPS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane / LPS, rr = lane % LPS;
    if (tid <= MAX_SEG_CYCLES) s_nv[tid] = tid <= a.max_cnt ? a.nv_tab[tid] : 0.0;
    double obj_acc = 0.0, chg_acc = 0.0;

    constexpr bool VREC = SPW > 4;         // (for every shape, as buffer loads: C4 +4.7 %, C5 +3.6 %, C2 +2 % -- the scalar loads keep the records out of the address unit) eight segments per wave: their records would not fit the SGPRs -- per-lane vector loads instead
    struct RecRaw { int c0[VREC ? 1 : SPW], c1[VREC ? 1 : SPW]; EdgeInfo ei[VREC ? 1 : SPW]; int xa[XT ? (VREC ? 1 : SPW) : 1], xb[XT ? (VREC ? 1 : SPW) : 1]; int t0; };      // wave-uniform (SGPRs); VREC: this lane group's record (VGPRs)
    struct Rec { int c0, cnt, rbi, rbj, sa, sb, seg; };                   // this lane group's segment; sharded runs: seg = ta, sb = tb (exchange positions)
    // Which cycles of its segment a lane owns.  Round 2: rr + LPS * e (every stream instruction reads one contiguous run of 8-byte words).
    // Round 3 (PAIR): the adjacent pairs 2 rr + 2 LPS * (e / 2) + {0, 1}, so that the 8-byte streams (old weights, S0, Adam moments) are read
    // 16 bytes per lane: the CU's address unit takes 16.4 cycles for a contiguous 8-byte-per-lane load and 16.3 for a 16-byte one
    // (tools/probes/gather_probe.hip, profiles/r03_gather_probe.txt), i.e. half the cycles per byte.  The packed words become 8-byte loads
    // (16.4 cycles for two cycles' worth instead of 2 x 6.2), the weight stores 16-byte stores (43.6 = 2 x 21.9: no change).
    // Only the constant / piecewise-step shapes for segments of up to 64 cycles (LPS <= 16: what the BASELINE workloads run).  Measured on the others:
    // unsampled C5 (<64,4>) 8.05 -> 7.89 ms; the Adam instances (<32,2> at C2 / C4) no gain (2.19 -> 2.27 ms at C4, 0.251 -> 0.267 at C2) -- both
    // keep the round-2 map, and with it results bitwise equal to k_sweep_node's (which two tests assert).
    constexpr bool PAIR = (E % 2 == 0) && LPS <= 16 && !ADAM;
    // (four adjacent cycles per lane -- the packed words as ONE 16-byte load -- measured on top: C4 1025-1054 vs 989-1034 us, C2 +3 %: not adopted)
    auto cidx = [&](int e) { return PAIR ? 2 * rr + (e & 1) + 2 * LPS * (e >> 1) : rr + LPS * e; };
    struct Str { uint32_t pk[E]; double w[E], d[E], am[EA], av[EA]; };     // am / av: HybridGradient.m_t / v_t (Adam only)
    struct Gat { double sj[E], si[E], T1, T2, So; };

    const int p0 = uniform_load(b.piece_ptr, blockIdx.x), p1 = uniform_load(b.piece_ptr, blockIdx.x + 1);
    unsigned long long wg_c0 = 0;
    if (b.wg_clock && tid == 0) { b.wg_clock[2 * blockIdx.x] = wall_clock64(); wg_c0 = clock64(); }
    __shared__ int s_ticket;
    // deterministic workgroup partials of what has been accumulated since the last flush -> pair `slot`
    auto flush_partials = [&](int slot) {
        const double o1 = group_sum<64>(obj_acc), c1 = group_sum<64>(chg_acc);
        __syncthreads();
        if (lane == 0) { s_part[0][wv] = o1; s_part[1][wv] = c1; }
        __syncthreads();
        if (tid == 0) {
            double o = 0.0, c = 0.0;
            for (int k = 0; k < NW; ++k) { o += s_part[0][k]; c += s_part[1][k]; }
            a.partials[2 * slot] = o;
            a.partials[2 * slot + 1] = c;
        }
        obj_acc = 0.0; chg_acc = 0.0;
    };
    // num_records = the arrays' real lengths: an offset past the end reads 0 / stores nothing instead of touching a neighbouring block
#ifndef DESC_RSRC_UNBOUNDED          // A/B builds only: 1 = the round-3 descriptors (num_records 0xFFFFFFFF)
#define DESC_RSRC_UNBOUNDED 0
#endif
    const uint32_t nb_csr = DESC_RSRC_UNBOUNDED ? 0xFFFFFFFFu : a.csr_bytes, nb_t = DESC_RSRC_UNBOUNDED ? 0xFFFFFFFFu : a.t_bytes, nb_sl = DESC_RSRC_UNBOUNDED ? 0xFFFFFFFFu : a.slice_bytes;
    const __amdgpu_buffer_rsrc_t rs_S = make_rsrc(a.S_old, nb_csr), rs_T = make_rsrc(a.Tfull, nb_t), rs_Sn = make_rsrc(a.S_new, nb_csr);      // CSR-aligned: 2m doubles < 4 GiB
    const __amdgpu_buffer_rsrc_t rs_sl = XT ? make_rsrc(a.s_slice, nb_sl) : rs_Sn;                        // sharded runs: this rank's all-gather slice
    const __amdgpu_buffer_rsrc_t rs_cum = make_rsrc(a.cum, (a.seg_count + 1u) * 4u), rs_ei = make_rsrc(a.einfo, a.seg_count * 16u);          // per-lane records (VREC shapes only; C3 -1 %)
    int pc = p0, ticket = -1;
    for (;;) {
        if (ticket < 0 && pc >= p1) {                      // own list done: its partials, then the shared tail
            flush_partials(blockIdx.x);
            if (b.n_tail == 0) break;
            ticket = 0;
        }
        if (ticket >= 0) {
            if (tid == 0) s_ticket = atomicAdd(b.ticket, 1);
            __syncthreads();
            ticket = __builtin_amdgcn_readfirstlane(s_ticket);
            __syncthreads();
            if (ticket >= b.n_tail) break;
            pc = b.tail_first + ticket;
        }
        const PieceDesc pd = uniform_load_piece(b.pieces, pc);
        __syncthreads();                                   // every wave is done with the previous band's rows
        const int nit = (pd.seg_hi - pd.seg_lo + NW * SPW - 1) / (NW * SPW);
        // the piece's cycles are one contiguous range of the per-cycle arrays: Rec::c0 counts from its start
        const int c_lo = uniform_load(a.cum, pd.seg_lo);
        const uint32_t c_len = (uint32_t)(uniform_load(a.cum, pd.seg_hi) - c_lo) + 8u;           // + the 16-byte tail the arrays are padded by
        const __amdgpu_buffer_rsrc_t rs_pk = make_rsrc(a.pk + c_lo, c_len * 4u), rs_w = make_rsrc(a.w_old + c_lo, c_len * 8u),
                                     rs_d = make_rsrc(a.S0 + c_lo, c_len * 8u), rs_wn = make_rsrc(a.w_new + c_lo, c_len * 8u);
        __amdgpu_buffer_rsrc_t rs_am = rs_w, rs_av = rs_w, rs_amo = rs_wn, rs_avo = rs_wn;       // Adam moments (HybridGradient.m:28-35), read and written per cycle
        if constexpr (ADAM) {
            rs_am = make_rsrc(a.st.adam_m + c_lo, c_len * 8u); rs_av = make_rsrc(a.st.adam_v + c_lo, c_len * 8u);
            rs_amo = make_rsrc(a.st.adam_m_out + c_lo, c_len * 8u); rs_avo = make_rsrc(a.st.adam_v_out + c_lo, c_len * 8u);
        }

        auto load_raw = [&](int it) -> RecRaw {            // past the end: the piece's last segment, cnt = 0
            RecRaw q;
            q.t0 = pd.seg_lo + (it * NW + wv) * SPW;
            if constexpr (VREC) {
                const int t = min(q.t0 + grp, pd.seg_hi - 1);
                {
                    const u32x2_t cc = __builtin_amdgcn_raw_buffer_load_b64(rs_cum, t * 4, 0, 0);
                    const u32x4_t ee = __builtin_amdgcn_raw_buffer_load_b128(rs_ei, t * 16, 0, 0);
                    q.c0[0] = (int)cc.x; q.c1[0] = (int)cc.y;
                    q.ei[0].rb_i = (int)ee.x; q.ei[0].rb_j = (int)ee.y; q.ei[0].slot_a = (int)ee.z; q.ei[0].slot_b = (int)ee.w;
                }
                if constexpr (XT) { const int2 x = a.xt[t]; q.xa[0] = x.x; q.xb[0] = x.y; }
                return q;
            }
#pragma unroll
            for (int s2 = 0; s2 < SPW; ++s2) {
                const int t = min(q.t0 + s2, pd.seg_hi - 1);
                q.c0[s2] = uniform_load(a.cum, t); q.c1[s2] = uniform_load(a.cum, t + 1);
                q.ei[s2] = uniform_load_einfo(a.einfo, t);
                if constexpr (XT) { q.xa[s2] = uniform_load((const int32_t*)a.xt, 2 * t); q.xb[s2] = uniform_load((const int32_t*)a.xt, 2 * t + 1); }
            }
            return q;
        };
        auto land = [&](const RecRaw& q) -> Rec {
            Rec r{};
            if constexpr (VREC) {
                r.c0 = q.c0[0] - c_lo; r.cnt = q.t0 + grp < pd.seg_hi ? q.c1[0] - q.c0[0] : 0;
                r.rbi = q.ei[0].rb_i - pd.row_lo; r.rbj = q.ei[0].rb_j; r.sa = q.ei[0].slot_a; r.sb = XT ? q.xb[0] : q.ei[0].slot_b;
                r.seg = XT ? q.xa[0] : 0;
                return r;
            }
#pragma unroll
            for (int s2 = 0; s2 < SPW; ++s2)
                if (grp == s2) {
                    r.c0 = q.c0[s2] - c_lo; r.cnt = q.t0 + s2 < pd.seg_hi ? q.c1[s2] - q.c0[s2] : 0;
                    r.rbi = q.ei[s2].rb_i - pd.row_lo; r.rbj = q.ei[s2].rb_j; r.sa = q.ei[s2].slot_a; r.sb = XT ? q.xb[s2] : q.ei[s2].slot_b;
                    r.seg = XT ? q.xa[s2] : 0;
                }
            return r;
        };
        auto load_stream = [&](const Rec& r) -> Str {      // unconditional: idle lanes repeat a valid cycle of the segment
            Str x;
            if constexpr (PAIR) {
#pragma unroll
                for (int h = 0; h < E / 2; ++h) {
                    // the pair's first cycle, clamped into the segment; its second one may lie past the segment's end (an odd count, idle lanes):
                    // that is the next segment's first cycle or the arrays' padding -- readable, masked in compute; its packed word is replaced by
                    // the first one's, so that the gathers formed from it stay inside rows i and j
                    const int first = min(2 * rr + 2 * LPS * h, max(r.cnt, 1) - 1);
                    const uint32_t cr = (uint32_t)(r.c0 + first);
                    buf_load_u32x2(rs_pk, cr * 4u, x.pk[2 * h], x.pk[2 * h + 1]);
                    buf_load_f64x2(rs_w, cr * 8u, x.w[2 * h], x.w[2 * h + 1]);
                    buf_load_f64x2(rs_d, cr * 8u, x.d[2 * h], x.d[2 * h + 1]);
                    if (ADAM) { buf_load_f64x2(rs_am, cr * 8u, x.am[(2 * h) % EA], x.am[(2 * h + 1) % EA]); buf_load_f64x2(rs_av, cr * 8u, x.av[(2 * h) % EA], x.av[(2 * h + 1) % EA]); }
                    if (first + 1 >= r.cnt) x.pk[2 * h + 1] = x.pk[2 * h];
                }
                return x;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int cr = r.c0 + min(rr + LPS * e, max(r.cnt, 1) - 1);
                x.pk[e] = buf_load_u32(rs_pk, (uint32_t)cr * 4u); x.w[e] = buf_load_f64(rs_w, (uint32_t)cr * 8u); x.d[e] = buf_load_f64(rs_d, (uint32_t)cr * 8u);
                if (ADAM) { x.am[e % EA] = buf_load_f64(rs_am, (uint32_t)cr * 8u); x.av[e % EA] = buf_load_f64(rs_av, (uint32_t)cr * 8u); }
            }
            return x;
        };
        auto issue_gathers = [&](const Rec& r, const Str& x) -> Gat {
            Gat g;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t p = x.pk[e];
                g.sj[e] = buf_load_f64(rs_S, (uint32_t)(r.rbj + (int)((p >> 16) & 0x7FFFu)) * 8u);      // S({j,k}): a gather inside row j (L1 / L2)
                g.si[e] = s_rows[r.rbi + (int)(p & 0x7FFFu)];                                              // S({k,i}): the band's rows in the LDS
            }
            const int ta = XT ? r.seg : r.sa, tb = r.sb;
                        // this and the single S store: C4 1059.5 -> 1045.8 / 1083 -> 1059 us, C2 within noise (profiles/r03_buffer_instructions.txt)
            g.T1 = buf_load_f64(rs_T, (uint32_t)((lane & 1) ? tb : ta) * 8u);
            g.T2 = 0.0;
            g.So = s_rows[r.sa - pd.row_lo];
            return g;
        };
        // arithmetic + stores of one segment group (DESC_PGD.m:193-233), everything in registers
        auto compute = [&](const Rec& r, const Str& x, const Gat& g0) {
            const int cnt = r.cnt;
            Gat g = g0;
            const double tconv = fx_to_double(g0.T1, a.fx_inv);       // this lane's load: T1 (even lanes) or T2 (odd lanes)
            g.T1 = dpp_mov_f64<0xA0>(tconv);        // quad_perm:[0,0,2,2]: the even lane's value = T1
            g.T2 = dpp_mov_f64<0xF5>(tconv);        // quad_perm:[1,1,3,3]: the odd lane's value = T2
            double ws[E], mo[EA], vo[EA];
            uint32_t okm = 0;
            double part = 0.0;
            if (__ballot(cnt > 0) != 0ull) {               // wave-uniform; no memory operation inside
                const double nv = cnt > 0 ? s_nv[cnt] : 0.0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    // idle lanes (ok false) hold a copy of the segment's last cycle (load_stream clamps): what they compute from it is finite and
                    // discarded below (ws[e] = 0, no part in any sum), so only the sums mask them
                    const bool ok = cidx(e) < cnt;
                    const uint32_t p = x.pk[e];
                    const double w = ok ? x.w[e] : 0.0, d = x.d[e];
                    if (ok) okm |= 1u << e;
                    const double ss = g.sj[e] + g.si[e];                                                 // S(jk)+S(ki)
                    obj_acc += w * ss;                                                                   // :233, one sweep late
                    const double gr = ss + (((p & 0x8000u) ? g.T1 : 0.0) + ((p & 0x80000000u) ? g.T2 : 0.0)) * d;   // :193
                    ws[e] = gr;
                    part += ok ? gr * nv : 0.0;
                }
                const double dot = group_sum<LPS>(part);                                                 // :199-201
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double gr = ws[e] - dot * nv;
                    if (ADAM) {                            // HybridGradient.m:28-35 (strategy 0): apply_step with the moments in registers
                        const double mt = (a.st.beta1 * x.am[e % EA]) + (1.0 - a.st.beta1) * gr;
                        const double vt = (a.st.beta2 * x.av[e % EA]) + (1.0 - a.st.beta2) * (gr * gr);
                        mo[e % EA] = mt; vo[e % EA] = vt;
                        const double cm = mt / a.st.bc1, cv = vt / a.st.bc2;
                        ws[e] = ((okm >> e) & 1u) ? x.w[e] + (-a.st.lr * cm / (sqrt(cv) + 1e-8)) : 0.0;
                    } else
                    ws[e] = ((okm >> e) & 1u) ? apply_step<STEP>(a.st, x.w[e], gr, 0) : 0.0;             // :207
                }
                uint32_t act = okm;                        // simplex threshold (:215-223), Michelot fixed point
                double s1v = 0.0; int na = 1;
                for (;;) {
                    double sp = 0.0;
#pragma unroll
                    for (int e = 0; e < E; ++e) sp += ((act >> e) & 1u) ? ws[e] : 0.0;
                    s1v = group_sum<LPS>(sp) - 1.0;
                    if (LPS == 8) na = max(group8_sum((int)__popc(act)), 1);
                    else if (LPS == 16) na = max(group16_sum((int)__popc(act)), 1);
                    else {
                        na = 0;
#pragma unroll
                        for (int e = 0; e < E; ++e) na += group_count<LPS>((act >> e) & 1u, lane);
                        na = max(na, 1);
                    }
                    const double nad = (double)na;
                    uint32_t keep = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) keep |= (((act >> e) & 1u) && ws[e] * nad > s1v) ? 1u << e : 0u;
                    const bool changed = keep != act;
                    act = keep;
                    if (!__any(changed)) break;
                }
                const double T = s1v / (double)na;
                double sn = 0.0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double wn = fmax(ws[e] - T, 0.0);                                              // :224
                    ws[e] = wn;
                    if ((okm >> e) & 1u) sn += wn * x.d[e];
                }
                part = group_sum<LPS>(sn);                                                               // :229
            }
            if constexpr (PAIR) {                      // 16-byte stores of the pairs; the last pair of an odd segment: its first cycle alone
#pragma unroll
                for (int h = 0; h < E / 2; ++h) {
                    const uint32_t off = (uint32_t)(r.c0 + cidx(2 * h)) * 8u;
                    const bool ok0 = (okm >> (2 * h)) & 1u, ok1 = (okm >> (2 * h + 1)) & 1u;
                    if (ok1) {
                        buf_store_f64x2(rs_wn, off, ws[2 * h], ws[2 * h + 1]);
                        if (ADAM) { buf_store_f64x2(rs_amo, off, mo[(2 * h) % EA], mo[(2 * h + 1) % EA]); buf_store_f64x2(rs_avo, off, vo[(2 * h) % EA], vo[(2 * h + 1) % EA]); }
                    } else if (ok0) {
                        buf_store_f64(rs_wn, off, ws[2 * h]);
                        if (ADAM) { buf_store_f64(rs_amo, off, mo[(2 * h) % EA]); buf_store_f64(rs_avo, off, vo[(2 * h) % EA]); }
                    }
                }
            } else
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((okm >> e) & 1u) {
                    buf_store_f64(rs_wn, (uint32_t)(r.c0 + rr + LPS * e) * 8u, ws[e]);
                    if (ADAM) { buf_store_f64(rs_amo, (uint32_t)(r.c0 + rr + LPS * e) * 8u, mo[e % EA]); buf_store_f64(rs_avo, (uint32_t)(r.c0 + rr + LPS * e) * 8u, vo[e % EA]); }
                }
            if (!XT && cnt > 0 && rr < 2) buf_store_f64(rs_Sn, (uint32_t)(rr == 0 ? r.sa : r.sb) * 8u, part);
            if (cnt > 0 && rr == 0) {
                chg_acc += fabs(part - g.So);                                                            // :232
                if constexpr (XT) buf_store_f64(rs_sl, (uint32_t)r.seg * 8u, part);       // sharded: k_unpack_S copies it into the CSR-aligned replica
            }
        };

        // ---- prologue: records 0..3, streams 0..2, gathers 0
        Rec R0, R1, R2, R3; Str S0, S1, S2, S3; Gat G0, G1;
        RecRaw Q;
        { const RecRaw q = load_raw(0); R0 = land(q); S0 = load_stream(R0); }
        { const RecRaw q = load_raw(1); R1 = land(q); S1 = load_stream(R1); }
        { const RecRaw q = load_raw(2); R2 = land(q); S2 = load_stream(R2); }
        Q = load_raw(3);
        // the band's rows of S_old -> LDS, while the first streams are in flight
        {
            // up to 19 loads in flight per thread: one batch for 1024 threads, two for 512 (38 at once cost the 64 x 4 instance 3.5 %)
            constexpr int RL = (BAND_ROW_CAP + 1023) / 1024;
            // 16 bytes per lane (half the address-unit cycles per byte, see PAIR): thread t takes the doubles 2 t, 2 t + 1 of every batch of 2 NT.
            // The clamped load of a one-entry band would end one double past the rows -- such a band (one node of degree 1, no triangle) never has a
            // segment and therefore never a piece; all the same rs_S carries the array's real length (the read returns 0) and the CSR-aligned arrays
            // are allocated two doubles longer.
            constexpr int RL2 = (RL + 1) / 2;
            for (int base = 0; base < pd.row_len; base += 2 * RL2 * NT) {
                double v0[RL2], v1[RL2];
#pragma unroll
                for (int u = 0; u < RL2; ++u)
                    buf_load_f64x2(rs_S, (uint32_t)(pd.row_lo + min(base + 2 * (u * NT + tid), max(pd.row_len - 2, 0))) * 8u, v0[u], v1[u]);
#pragma unroll
                for (int u = 0; u < RL2; ++u) {
                    const int t0 = base + 2 * (u * NT + tid);
                    if (t0 + 1 < pd.row_len) { s_rows[t0] = v0[u]; s_rows[t0 + 1] = v1[u]; }
                    else if (t0 < pd.row_len) s_rows[t0] = pd.row_len >= 2 ? v1[u] : v0[u];      // the row's last double: the clamped load ended on it
                }
            }
        }
        __syncthreads();
        G0 = issue_gathers(R0, S0);
        // one iteration: Ra/Sa/Ga = group g (computed), Rb/Sb = g+1 (gathers issued into Gb), Rd/Sd <- g+3
        auto step = [&](int g, const Rec& Ra, const Str& Sa, const Gat& Ga, const Rec& Rb, const Str& Sb, Gat& Gb, Rec& Rd, Str& Sd) {
            const RecRaw qn = load_raw(g + 4);
            Gb = issue_gathers(Rb, Sb);
            Rd = land(Q);
            Sd = load_stream(Rd);
            compute(Ra, Sa, Ga);
            Q = qn;
        };
        for (int g = 0; g < nit; g += 4) {                 // the tail iterations past nit compute nothing (cnt = 0)
            step(g,     R0, S0, G0, R1, S1, G1, R3, S3);
            step(g + 1, R1, S1, G1, R2, S2, G0, R0, S0);
            step(g + 2, R2, S2, G0, R3, S3, G1, R1, S1);
            step(g + 3, R3, S3, G1, R0, S0, G0, R2, S2);
        }
        if (ticket >= 0) flush_partials((int)gridDim.x + ticket);      // a tail piece: its own pair of partials
        else ++pc;
    }
    if (b.wg_clock && tid == 0) {
        b.wg_clock[2 * blockIdx.x + 1] = wall_clock64();
        b.wg_clock[2 * gridDim.x + blockIdx.x] = clock64() - wg_c0;          // shader-clock cycles of the same interval: the clock the box ran at
    }
}

// Sum the block partials in a fixed order, record the traces and run the early-stop
// rule of DESC_PGD.m:243-256 for the iteration whose objective just became known.
// t = 1-based index of the sweep that produced the partials.  Called by one full wave.
struct FinArgs {
    const double* partials; DevState* st; double* obj_trace; double* avg_trace;
    int64_t m; double stop_tol; int32_t nparts, t, patience, last_only;      // t == 0: nothing to do
    int64_t rank_stride; int32_t nranks;     // sharded runs: the partials of rank r start at partials + r * rank_stride (0 / 0: one rank)
};
// The sum of the workgroup partials is defined over 256 VIRTUAL lanes: virtual lane v adds the (rank, index) pairs v, v + 256, ... in that order, the
// 64 virtual lanes of a quarter are combined by the fixed DPP butterfly and the four quarters are added in index order.  A 256-thread block
// (finalize_block: the bookkeeping workgroup of the column-sum and unpack launches) gives a quarter to each of its waves; a single wave
// (finalize_wave: k_finalize) walks the four quarters one after the other -- the SAME bits either way, and the same on every rank.  Eight pairs are
// loaded before any is added (one at a time the 8 x 1024 pairs of a world-8 run were 128 dependent round trips for one wave: ~50 of the 85 us of
// k_unpack_S, profiles/r04_shard_w8_c4_rocprof.txt -- 35 us in the launches that book-keep nothing).
__device__ __forceinline__ void finalize_quarter(const FinArgs& f, int quarter, int lane, double& o, double& ch) {
    o = 0.0; ch = 0.0;
    const int E = max(f.nranks, 1) * f.nparts;
    for (int base = 64 * quarter + lane; base < E; base += 256 * 8) {
        double vo[8], vc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + 256 * u;
            vo[u] = 0.0; vc[u] = 0.0;
            if (idx < E) {
                const double* q = f.partials + (int64_t)(idx / f.nparts) * f.rank_stride + 2 * (int64_t)(idx % f.nparts);
                vo[u] = q[0]; vc[u] = q[1];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { o += vo[u]; ch += vc[u]; }
    }
    o = group_sum<64>(o); ch = group_sum<64>(ch);
}
__device__ __forceinline__ void finalize_book(const FinArgs& f, double o, double ch);
// called by all 256 threads of a block
__device__ __forceinline__ void finalize_block(const FinArgs f) {
    __shared__ double s_fin[4][2];
    if (f.t <= 0 || f.st->stop) return;          // t == 0: no sweep to book-keep (uniform over the block)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double o, ch;
    finalize_quarter(f, wv, lane, o, ch);
    if (lane == 0) { s_fin[wv][0] = o; s_fin[wv][1] = ch; }
    __syncthreads();
    if (threadIdx.x == 0)
        finalize_book(f, ((s_fin[0][0] + s_fin[1][0]) + s_fin[2][0]) + s_fin[3][0], ((s_fin[0][1] + s_fin[1][1]) + s_fin[2][1]) + s_fin[3][1]);
}
// called by one full wave
__device__ __forceinline__ void finalize_wave(const FinArgs f) {
    if (f.t <= 0 || f.st->stop) return;
    const int lane = threadIdx.x & 63;
    double q0[4], q1[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) finalize_quarter(f, w, lane, q0[w], q1[w]);
    if (lane == 0) finalize_book(f, ((q0[0] + q0[1]) + q0[2]) + q0[3], ((q1[0] + q1[1]) + q1[2]) + q1[3]);
}
// traces and the stop rule of DESC_PGD.m:232-257 for the iteration whose sums just became known (one thread)
__device__ __forceinline__ void finalize_book(const FinArgs& f, double o, double ch) {
    DevState* st = f.st;
    // last_only: the partials come from the objective kernel after the final sweep t and
    // hold obj(t).  Otherwise they come from sweep t: obj(t-1) and sum|dS| of sweep t.
    const int t = f.t;
    const int it = f.last_only ? t : t - 1;          // iteration whose objective is o
    if (!f.last_only) f.avg_trace[t - 1] = ch / (double)f.m;                            // :232
    if (it >= 1) {
        f.obj_trace[it - 1] = o;                                                        // :233
        if (it <= st->last_tested) return;          // already tested (objective evaluated early by a download)
        st->last_tested = it;
        if (it > 1 && f.obj_trace[it - 2] - f.obj_trace[it - 1] < f.stop_tol) {         // :243
            st->misses += 1;
            if (st->misses >= f.patience) {                                             // :245-246
                st->stop = 1; st->iters_run = it; st->final_parity = it & 1;
            }
        } else {
            st->misses = 0;                                                             // :255
        }
    }
}
__global__ __launch_bounds__(64) void k_finalize(FinArgs f) { finalize_wave(f); }

// Mirror-weight column sums (DESC_PGD.m:185-191 in node form).  One workgroup per node
// v: for every incident edge {v,u} (CSR order) stream its segment; a cycle with third
// vertex t adds its weight to column idx_v(t) if the reverse cycle ({v,t};u) was sampled.
// Each wave owns a private copy of the columns in LDS (segments are dealt to waves
// round-robin; third vertices inside a segment are distinct), copies are added in a
// fixed order at the end -> bitwise reproducible.  Loads of COLSUM_U segments are in
// flight per wave at once.
// Round 3, what bounds it (profiles/r03_colsum_bound.txt): with the LDS adds replaced by plain stores or removed altogether the kernel takes
// 227.3 / 224.7 us instead of 227.0 at C4 -- the adds are free, the time is the scattered ~100-byte runs of `w` (one per incident segment,
// ~3 TB/s of real traffic).  A flat variant (a node's contributing cycles as ONE contiguous range of two static streams, column + cycle
// index, every lane of every load busy, no segment records) was built and measured: 250 vs 228 us at C4, 29.7 vs 27.0 at C2, 404 vs 392 at
// C5 -- its 4 extra bytes per entry cost more than the idle lanes of the 16-lanes-per-segment form; removed again.  Two adjacent cycles per lane
// (16-byte loads, what gained 4-5 % in the band sweep): 309 vs 238 us -- half as many independent requests in flight per wave; removed too.
constexpr int COLSUM_U = 8;
// The LAST workgroup of the launch does no column at all: it runs the bookkeeping of the previous sweep (traces,
// stop rule) that used to be a separate one-wave launch per iteration.  If it sets the stop flag while the
// node workgroups of this launch are running, they finish a T that nobody reads: the sweep that follows returns at once.
// The column of every contributing cycle comes from `midx`: a static stream of 16-bit column indices in exactly the order this
// kernel consumes them (node-major, then incident segment, then contributing cycle; `moff` = start of a CSR slot's run), read
// sequentially -- instead of the 4-byte packed words of the cycles, which sit in scattered 50-byte runs next to the weights
// (round 2: -25 % of this pass's sectors).
// Walks the segments [0, nrun) of one node's run (records in LDS: sb = first contributing cycle, sc = their number, sm = start in midx)
// in groups of SPI = 64 / CL consecutive segments, first group g_first, then every g_step-th, adding each contributing cycle's weight to
// column midx[..] of `cols`.  COLSUM_U groups are in flight per wave.
//
// The columns are 64-bit FIXED-POINT sums (round 4): a weight lies in [0, 1], a column adds at most one cycle of every incident segment, so
// round(w * 2^fx_bits) with fx_bits = 62 - ceil(log2(max degree + 1)) never overflows and costs <= 2^-(fx_bits+1) per term (2.2e-16 at C4,
// below the rounding of a double sum of the same terms).  Integer adds commute: the sum does not depend on the order in which lanes, waves or
// the LDS unit apply them, so the pass is bitwise reproducible BY CONSTRUCTION, and ONE copy of the columns per workgroup serves all four
// waves (a quarter of the accumulator LDS: 3 -> 7 workgroups per CU at C4).  History: rounds 1-3 kept per-wave copies of doubles and relied on
// the LDS unit resolving same-address lanes of one ds_add_f64 in a fixed order (observed, not specified); the order-explicit form of that
// (one segment of an instruction after the other) measured +10 % on this kernel (profiles/r04_ab_colsum_ordered_and_descriptors.txt: C4 230 ->
// 253 us, C2 27.0 -> 29.5); the fixed-point form had measured -5 % in round 3 (profiles/r03_colsum_bound.txt) and was shelved then.
template <int CL>
__device__ __forceinline__ void colsum_walk(unsigned long long* cols, const int* sb, const int* sc, const uint32_t* sm, int nrun, int g_first, int g_step,
                                            const uint16_t* midx, const double* w, int lane, double fx_scale) {
    // Inside a segment the cycles are ordered [(ik;j) only | both mirrors | (jk;i) only |
    // none], so either endpoint reads one run: only the `nact` cycles that contribute to its
    // columns (~n_sample/codeg of them; the smaller endpoint the first two classes, the larger
    // the middle two).  A wave instruction serves 4 segments x 16 lanes; segments
    // with more than 16 contributing cycles take further passes.  Every load of a batch is
    // issued before any result is touched (a use between loads would make the compiler wait
    // for each one).
    // CL lanes per segment, i.e. 2 CL slots per segment and round.  16 is right when a segment has 12-25 contributing cycles per endpoint
    // (C2, C3, C4); with ~9 (C5: 30 sampled cycles, 30 % of them with a sampled mirror) half of a 16-lane group's loads and adds are
    // idle: 8 lanes and 8 segments per instruction there -- C5 389.5 -> 370 us; at C4 / C3 / C2 the narrower groups cost +2.6 / +2.7 / +15 %
    // (second rounds), so the host picks by the average run length (setup_node).
    constexpr int SPI = 64 / CL;          // segments per wave instruction
    const int sub = lane / CL, l16 = lane % CL;
    for (int g0 = g_first; SPI * g0 < nrun; g0 += g_step * COLSUM_U) {
        // pieces 0 and 1 (contributing cycles 0..31 of each segment) are loaded together;
        // segments with more take further rounds
        for (int round = 0; round < MAX_SEG_CYCLES / (2 * CL); ++round) {
            uint32_t pv[2 * COLSUM_U]; double wvv[2 * COLSUM_U];
            bool more = false;
#pragma unroll
            for (int u = 0; u < COLSUM_U; ++u) {
                const int tt = SPI * (g0 + g_step * u) + sub;
                pv[2 * u] = 0xFFFFu; pv[2 * u + 1] = 0xFFFFu; wvv[2 * u] = 0.0; wvv[2 * u + 1] = 0.0;
                if (tt < nrun) {
                    const int nact = sc[tt] & 0x1FF;
                    more |= nact > 2 * CL * (round + 1);
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int q = l16 + CL * (2 * round + h2);
                        if (q < nact) {
                            const int64_t c = (int64_t)sb[tt] + q;          // (as buffer loads, like the band sweep's: no change, 239.1 vs 239.0 us at C4)
                            pv[2 * u + h2] = midx[(size_t)sm[tt] + q];
                            wvv[2 * u + h2] = w[c];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2 * COLSUM_U; ++u)
                if (pv[u] != 0xFFFFu) atomicAdd(&cols[pv[u]], (unsigned long long)__double2ll_rn(wvv[u] * fx_scale));      // ds_add_u64
            if (!__any(more)) break;
        }
    }
}

constexpr int COLSUM_SHORT = 128;         // sharded runs: rows whose owned run has at most this many segments are summed by ONE wave (four nodes per workgroup)
// LDS: `copies` x stride_cols 64-bit columns (1: the workgroup's shared copy; 4: one per wave, when the launch has short-run nodes), then 3 ints per incident edge
template <int CL>
__global__ __launch_bounds__(256) void k_colsum_node(const int32_t* rowptr, const int2* adj_seg, const uint32_t* moff, const uint16_t* midx, const double* w,
                                                     double* Tfull, int n, int stride_cols, const DevState* st, const int32_t* xpos, FinArgs fin, int32_t* tail_ticket, const int32_t* node_order,
                                                     const int2* node_run, int n_long, int copies, double fx_scale) {
    if (blockIdx.x == gridDim.x - 1) {
        if (fin.st) finalize_block(fin);
        if (threadIdx.x < 8 && tail_ticket) tail_ticket[threadIdx.x] = 0;      // the sweep launches that follow (one per exchange part) hand out their tail pieces from 0
        return;
    }
    if (st->stop) return;
    extern __shared__ unsigned long long acc[];
    int* seg_base = (int*)(acc + copies * stride_cols);
    int* seg_cf = seg_base + stride_cols;             // slot_record: {first contributing cycle, their number (| flag)}
    uint32_t* seg_mo = (uint32_t*)(seg_cf + stride_cols);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // node_order lists the nodes to visit (the dispatcher hands workgroups out in index order: see setup_node for the order), first the
    // n_long nodes that get a workgroup each, then the short-run nodes of a sharded run, four per workgroup
    if ((int)blockIdx.x < n_long) {
        const int vi = blockIdx.x;
        const int v = node_order ? node_order[vi] : vi;
        const int r0 = rowptr[v], deg = rowptr[v + 1] - r0;
        if (deg == 0) return;
        // Sharded runs: the segments of row v this rank owns -- those whose smaller endpoint lies in its node range -- are ONE contiguous run
        // [run.x, run.y) of the row's CSR slots (neighbours ascending); only that run is loaded and walked.  One rank: the whole row.
        const int2 run = node_run ? node_run[v] : int2{0, deg};
        const int ra = r0 + run.x, nrun = run.y - run.x;
        for (int t = threadIdx.x; t < deg; t += 256) acc[t] = 0ull;
        for (int t = threadIdx.x; t < nrun; t += 256) {    // CSR-aligned segment records: one coalesced load
            const int2 rec = adj_seg[ra + t];
            seg_base[t] = rec.x; seg_cf[t] = rec.y; seg_mo[t] = moff[ra + t];
        }
        __syncthreads();
        // groups of segments are dealt to the four waves round-robin; all add into the one shared copy
        colsum_walk<CL>(acc, seg_base, seg_cf, seg_mo, nrun, wv, 4, midx, w, lane, fx_scale);
        __syncthreads();
        for (int t = threadIdx.x; t < deg; t += 256)
            Tfull[xpos ? xpos[r0 + t] : r0 + t] = __longlong_as_double((long long)acc[t]);   // the fixed-point bits (fx_to_double); xpos: owner-sorted exchange layout
        return;
    }
    // Short runs (sharded ranks: most rows hold only a few dozen of a rank's segments -- at C4 over 8 GPUs rank 0 visits 4670 rows with ~66 of
    // its segments each): one WAVE per node, its copy of the columns wave-private, no barrier; four nodes per workgroup.  Measured per rank
    // before (a workgroup per node): 50-79 us by rank for ~29 us worth of entries (profiles/r04_shard_w8_c4_default.json).
    const int vi = n_long + ((int)blockIdx.x - n_long) * 4 + wv;
    if (vi >= n) return;
    const int v = node_order[vi];
    const int r0 = rowptr[v], deg = rowptr[v + 1] - r0;
    const int2 run = node_run[v];
    const int ra = r0 + run.x, nrun = min(run.y - run.x, COLSUM_SHORT);
    unsigned long long* mine = acc + wv * stride_cols;
    int* sb = seg_base + wv * COLSUM_SHORT; int* sc = seg_cf + wv * COLSUM_SHORT; uint32_t* sm = seg_mo + wv * COLSUM_SHORT;
    for (int t = lane; t < deg; t += 64) mine[t] = 0ull;
    for (int t = lane; t < nrun; t += 64) {
        const int2 rec = adj_seg[ra + t];
        sb[t] = rec.x; sc[t] = rec.y; sm[t] = moff[ra + t];
    }
    __builtin_amdgcn_wave_barrier();                  // one wave: its LDS operations execute in program order
    colsum_walk<CL>(mine, sb, sc, sm, nrun, 0, 1, midx, w, lane, fx_scale);
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < deg; t += 64) Tfull[xpos ? xpos[r0 + t] : r0 + t] = __longlong_as_double((long long)mine[t]);
}

// Setup of the node layout: packs idx_i(k) / idx_j(k) / mirror bits of every cycle and
// evaluates its inconsistency.  kf = k | (ikj>=0)<<30 | (jki>=0)<<31, already in device
// (band-major) order.
__global__ __launch_bounds__(256) void k_layout_node(const int32_t* cum, const int32_t* pos_edge,
                                                     const int32_t* ind_i, const int32_t* ind_j, const uint32_t* kf,
                                                     const int32_t* rowptr, const int32_t* adj, const int32_t* adj_eid,
                                                     const double* rij, uint32_t* pk, double* S0, int m_pos) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int base = cum[l], cnt = cum[l + 1] - base;
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        const int ri = rowptr[i], di = rowptr[i + 1] - ri, rj = rowptr[j], dj = rowptr[j + 1] - rj;
        double A[9];
        for (int t = 0; t < 9; ++t) A[t] = rij[9 * (int64_t)e + t];
        for (int q = lane; q < cnt; q += 64) {
            const uint32_t x = kf[(int64_t)base + q];
            const int k = (int)(x & 0x3FFFFFFFu);
            int lo = 0, hi = di;                                  // idx_i(k): position of k in row i
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[ri + mid] < k) lo = mid + 1; else hi = mid; }
            const int xi = min(lo, max(di - 1, 0));               // clamp: memory-safe even for a bogus imported structure
            lo = 0; hi = dj;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[rj + mid] < k) lo = mid + 1; else hi = mid; }
            const int xj = min(lo, max(dj - 1, 0));
            pk[(int64_t)base + q] = (uint32_t)xi | ((x >> 30) & 1u) << 15 | (uint32_t)xj << 16 | ((x >> 31) & 1u) << 31;
            const double tr = cycle_trace(A, rij + 9 * (int64_t)adj_eid[rj + xj], !(j < k), rij + 9 * (int64_t)adj_eid[ri + xi], !(k < i));
            S0[(int64_t)base + q] = abs_acos_ext((tr - 1.0) / 2.0) / M_PI;
        }
    }
}

// The same for a device-built structure, in place: reads the sampled k (natural order), decides
// the two mirror-present bits from the selection thresholds of the partner edges -- cycle (ik;j)
// was sampled iff (key(e_ik, j), j) <= (tau, ktau) of edge {i,k} -- and does the within-segment
// re-ordering [(ik;j) only | both mirrors | (jk;i) only | none] in the wave (stable: ascending k
// inside a class, exactly like the host path).  Segments have <= MAX_SEG_CYCLES cycles (pieces of 64).
// STAGED (round 4; natural order only: the segments of a node are then consecutive, node_seg[i] .. node_seg[i + 1]): one workgroup per node i
// first copies the rotation blocks of ALL edges incident to i -- row i of the CSR, 72 B per slot, 80 B apart in the LDS for 16-byte reads -- so that
// R_ki of every cycle of every segment (i, .) is an LDS read; R_jk stays a gather from the edge table.  Half of the kernel's 2 x 72 B of gathers per
// cycle (125 M cycles at C4: 18 GB out of the caches) become 0.36 GB of staging.
template <int NP, bool STAGED>                       // NP: pieces of 64 cycles per segment: 1, 2 or 4; STAGED: 512 threads (8 waves share a node's blocks), else 256
__global__ __launch_bounds__(STAGED ? 512 : 256) void k_layout_node_dev(const int32_t* cum, const int32_t* src_start, const int32_t* pos_edge,
                                                         const int32_t* ind_i, const int32_t* ind_j, const int32_t* nat_k,
                                                         const unsigned long long* tau, const int32_t* ktau,
                                                         uint64_t seed, const int32_t* rowptr, const unsigned long long* bits,
                                                         const uint32_t* rank, int words, const int32_t* adj_eid, const double* rij,
                                                         uint32_t* pk, double* S0, uint8_t* seg_perm, uint32_t* seg_counts, int m_pos, const int32_t* node_seg, int n_nodes) {
    extern __shared__ double s_blk[];                // STAGED: 10 doubles per slot of row i (9 used)
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
    const int n_outer = STAGED ? n_nodes : 1;
    for (int node = STAGED ? blockIdx.x : 0; node < n_outer; node += STAGED ? gridDim.x : 1) {
    int64_t l_first = wid, l_end = m_pos, l_step = nw;
    if (STAGED) {
        l_first = node_seg[node] + (threadIdx.x >> 6); l_end = node_seg[node + 1]; l_step = 8;
        if (node_seg[node] == l_end) continue;       // no segment with this smaller endpoint (uniform over the workgroup)
        const int r0n = rowptr[node], dn = rowptr[node + 1] - r0n;
        __syncthreads();                             // the previous node's blocks are no longer read
        for (int t = threadIdx.x; t < dn; t += 512) {
            double Bk[9];
            load_block9(rij + 9 * (int64_t)adj_eid[r0n + t], Bk);
            for (int q = 0; q < 9; ++q) s_blk[10 * t + q] = Bk[q];
        }
        __syncthreads();
    }
    for (int64_t l = l_first; l < l_end; l += l_step) {
        const int base = cum[l], cnt = cum[l + 1] - base, src = src_start[l];
        const int e = pos_edge[l], i = ind_i[e], j = ind_j[e];
        const int ri = rowptr[i], di = rowptr[i + 1] - ri, rj = rowptr[j], dj = rowptr[j + 1] - rj;
        double A[9];
        for (int t = 0; t < 9; ++t) A[t] = rij[9 * (int64_t)e + t];
        // pass 1: per cycle the packed word, the two partner edges and its class; class sizes per piece
        uint32_t word[NP]; int eik[NP], ejk[NP], kk[NP], cls[NP];
        unsigned long long cm[NP][4];
        int ncls[4] = {0, 0, 0, 0};
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
            const int t = pc * 64 + lane;
            const bool on = t < cnt;
            int k = 0, xi = 0, xj = 0, ei2 = e, ej2 = e;
            bool fi = false, fj = false;
            if (on) {
                k = nat_k[(int64_t)src + t];
                // idx_i(k), idx_j(k): positions of k in rows i and j = rank of its bit in the adjacency bitmaps
                const size_t wi = (size_t)i * words + (k >> 6), wj = (size_t)j * words + (k >> 6);
                const unsigned long long below = (1ull << (k & 63)) - 1ull;
                xi = min((int)(rank[wi] + __popcll(bits[wi] & below)), max(di - 1, 0));
                xj = min((int)(rank[wj] + __popcll(bits[wj] & below)), max(dj - 1, 0));
                ei2 = adj_eid[ri + xi]; ej2 = adj_eid[rj + xj];
                {   // both partner edges lie on the triangle {i,j,k}: they have cycles, their thresholds are defined
                    const unsigned long long key = d_sample_key(seed, (uint64_t)ei2, (uint64_t)j), th = tau[ei2];
                    fi = key < th || (key == th && j <= ktau[ei2]);                            // IKJ_appears (:113)
                }
                {
                    const unsigned long long key = d_sample_key(seed, (uint64_t)ej2, (uint64_t)i), th = tau[ej2];
                    fj = key < th || (key == th && i <= ktau[ej2]);                            // JKI_appears (:124)
                }
            }
            word[pc] = (uint32_t)xi | (fi ? 1u : 0u) << 15 | (uint32_t)xj << 16 | (fj ? 1u : 0u) << 31;
            eik[pc] = ei2; ejk[pc] = ej2; kk[pc] = k;
            cls[pc] = !on ? 4 : fi ? (fj ? 1 : 0) : (fj ? 2 : 3);
#pragma unroll
            for (int c = 0; c < 4; ++c) { cm[pc][c] = __ballot(cls[pc] == c); ncls[c] += __popcll(cm[pc][c]); }
        }
        if (lane == 0) seg_counts[l] = (uint32_t)ncls[0] | (uint32_t)(ncls[0] + ncls[1]) << 9 | (uint32_t)ncls[2] << 18;
        // pass 2: stable placement (class, then natural order) and the cycle inconsistency
        const int cbase[4] = {0, ncls[0], ncls[0] + ncls[1], ncls[0] + ncls[1] + ncls[2]};
        const unsigned long long lt = (1ull << lane) - 1ull;
        int seen[4] = {0, 0, 0, 0};
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
            if (pc * 64 >= cnt) break;                 // wave-uniform
            const int c = cls[pc];
            if (c < 4) {
                const int o = cbase[c] + seen[c] + __popcll(cm[pc][c] & lt);
                const int k = kk[pc];
                pk[(int64_t)base + o] = word[pc];
                seg_perm[(int64_t)base + o] = (uint8_t)(pc * 64 + lane);
                double tr;
                if (STAGED) {
                    double Cm[9];
                    const double* sb = s_blk + 10 * (int)(word[pc] & 0x7FFFu);          // slot idx_i(k) of row i: the block of edge {i, k}
                    const double2 c0 = *reinterpret_cast<const double2*>(sb), c1 = *reinterpret_cast<const double2*>(sb + 2), c2 = *reinterpret_cast<const double2*>(sb + 4),
                                  c3 = *reinterpret_cast<const double2*>(sb + 6);
                    Cm[0] = c0.x; Cm[1] = c0.y; Cm[2] = c1.x; Cm[3] = c1.y; Cm[4] = c2.x; Cm[5] = c2.y; Cm[6] = c3.x; Cm[7] = c3.y; Cm[8] = sb[8];
                    tr = cycle_trace_regs(A, rij + 9 * (int64_t)ejk[pc], !(j < k), Cm, !(k < i));
                } else tr = cycle_trace(A, rij + 9 * (int64_t)ejk[pc], !(j < k), rij + 9 * (int64_t)eik[pc], !(k < i));
                S0[(int64_t)base + o] = abs_acos_ext((tr - 1.0) / 2.0) / M_PI;
            }
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) seen[c2] += __popcll(cm[pc][c2]);
        }
    }
    }
}
// node_seg[v] = first natural segment (edge with cycles, ascending edge id) whose smaller endpoint is >= v; node_seg[n] = m_pos.  Ind is sorted by
// (i, j): the thread at every change of i fills the gap of nodes without segments below it.
__global__ __launch_bounds__(256) void k_node_seg_start(const int32_t* pos_edge, const int32_t* ind_i, int64_t m_pos, int n, int32_t* node_seg) {
    for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < m_pos; l += (int64_t)gridDim.x * 256) {
        const int i = ind_i[pos_edge[l]], prev = l > 0 ? ind_i[pos_edge[l - 1]] : -1;
        for (int v = prev + 1; v <= i; ++v) node_seg[v] = (int32_t)l;
        if (l == m_pos - 1) for (int v = i + 1; v <= n; ++v) node_seg[v] = (int32_t)m_pos;
    }
}
// Device-order segment tables from the plan's permutation (device-built structures): segment q of the device order is the
// structure's edge-with-cycles number order[q]
__global__ __launch_bounds__(256) void k_seg_tables(const int32_t* order, const int32_t* nat_pos_edge, const int32_t* nat_cum, int32_t* cum,
                                                    int32_t cyc_lo, int32_t* src_start, int32_t* pos_edge2, int32_t* devpos, int64_t m_pos) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q <= m_pos; q += (int64_t)gridDim.x * 256) {
        cum[q] -= cyc_lo;                                  // local cycle numbering (meaningful for owned segments)
        if (q == m_pos) break;
        const int32_t l = order[q], e = nat_pos_edge[l];
        src_start[q] = nat_cum[l];
        pos_edge2[q] = e; devpos[e] = (int32_t)q;
    }
}
// Round 4: a one-rank handle computes the packed words, S0_long and the in-segment class order in the structure's NATURAL order as soon as the
// sampled cycles exist -- while the host is still planning the band-major order (8-9 ms at C4) -- and moves whole segments to their device
// positions afterwards: segment q of the device order is natural segment order[q].  One wave per segment; ~26 bytes per cycle, bandwidth-bound.
__global__ __launch_bounds__(256) void k_permute_segments(const int32_t* order, const int32_t* nat_cum, const int32_t* cum, const uint32_t* pk_nat, const double* S0_nat,
                                                          const uint8_t* perm_nat, const uint32_t* counts_nat, uint32_t* pk, double* S0, uint8_t* seg_perm, uint32_t* counts,
                                                          int64_t m_pos) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    // PU segments in flight per wave: a segment alone is a chain of three dependent round trips (order -> starts -> data) for ~1 KB moved
    constexpr int PU = 4;
    for (int64_t q0 = wid * PU; q0 < m_pos; q0 += nw * PU) {
        int32_t l[PU]; int64_t src[PU], dst[PU]; int cnt[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) l[u] = order[min(q0 + u, m_pos - 1)];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int64_t q = min(q0 + u, m_pos - 1);
            src[u] = nat_cum[l[u]]; dst[u] = cum[q];
            cnt[u] = q0 + u < m_pos ? cum[q + 1] - cum[q] : 0;
        }
        uint32_t a[PU]; double b[PU]; uint8_t c[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (lane < cnt[u]) { a[u] = pk_nat[src[u] + lane]; b[u] = S0_nat[src[u] + lane]; c[u] = perm_nat[src[u] + lane]; }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            if (lane < cnt[u]) { pk[dst[u] + lane] = a[u]; S0[dst[u] + lane] = b[u]; seg_perm[dst[u] + lane] = c[u]; }
            for (int t = lane + 64; t < cnt[u]; t += 64) { pk[dst[u] + t] = pk_nat[src[u] + t]; S0[dst[u] + t] = S0_nat[src[u] + t]; seg_perm[dst[u] + t] = perm_nat[src[u] + t]; }
            if (lane == 0 && q0 + u < m_pos) counts[q0 + u] = counts_nat[l[u]];
        }
    }
}
// per-edge slots of the CSR-aligned arrays, on the device (device-built structures): eslot[e] = this
// edge's slot in its smaller endpoint's row; einfo of the edges with cycles in device order
__global__ __launch_bounds__(256) void k_edge_slots(const int32_t* ind_i, const int32_t* ind_j, const int32_t* rowptr, const int32_t* adj,
                                                    const int32_t* pos_edge2, int32_t* eslot, EdgeInfo* einfo, int64_t m, int64_t m_pos) {
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    auto slot = [&](int v, int u) {                   // position of u in row v
        const int r = rowptr[v]; int lo = 0, hi = rowptr[v + 1] - r;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[r + mid] < u) lo = mid + 1; else hi = mid; }
        return r + lo;
    };
    for (int64_t e = t0; e < m; e += nt) eslot[e] = slot(ind_i[e], ind_j[e]);
    for (int64_t q = t0; q < m_pos; q += nt) {
        const int e = pos_edge2[q], i = ind_i[e], j = ind_j[e];
        einfo[q] = EdgeInfo{rowptr[i], rowptr[j], slot(i, j), slot(j, i)};
    }
}
// Record of a CSR slot (v,u) for the column-sum pass: {first cycle of the run of segment {v,u} that contributes to node v's
// columns, length of the run | (v is the smaller endpoint) << 31}.  counts = n_ionly | n_i << 9 | n_jonly << 18 of the segment
// (class order [(ik;j) only | both | (jk;i) only | none]): the smaller endpoint reads [0, n_i), the larger [n_ionly, n_i + n_jonly).
__host__ __device__ inline int2 slot_record(int32_t seg_first, uint32_t counts, bool v_is_i) {
    const int n_io = counts & 0x1FFu, n_i = (counts >> 9) & 0x1FFu, n_jo = (counts >> 18) & 0x1FFu;
    return v_is_i ? int2{seg_first, (int)((uint32_t)n_i | 0x80000000u)} : int2{seg_first + n_io, n_i - n_io + n_jo};
}
// CSR-aligned segment records for the column-sum pass: 16 lanes per node row
__global__ __launch_bounds__(256) void k_adj_seg(const int32_t* rowptr, const int32_t* adj, const int32_t* adj_eid, const int32_t* devpos,
                                                 const int32_t* cum, const uint32_t* seg_counts, int seg_lo, int seg_hi, int2* adj_seg, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int v = row0; v < n; v += nrows)
        for (int t = rowptr[v] + l16; t < rowptr[v + 1]; t += 16) {
            const int q = devpos[adj_eid[t]];
            int2 rec{0, 0};
            if (q >= seg_lo && q < seg_hi) rec = slot_record(cum[q], seg_counts[q], v < adj[t]);
            adj_seg[t] = rec;
        }
}

// ---- the column-index stream of the column-sum pass
__device__ __forceinline__ int slot_nact(int2 rec) { return rec.y & 0x1FF; }         // contributing cycles of the segment behind a CSR slot, seen from the row's node
// entries per node row (16 lanes per row)
__global__ __launch_bounds__(256) void k_midx_rowsum(const int32_t* rowptr, const int2* adj_seg, uint32_t* rowsum, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int vb = row0 - (row0 % 4); vb < n; vb += nrows) {
        const int v = vb + (row0 % 4);
        int acc = 0;
        if (v < n) for (int t = rowptr[v] + l16; t < rowptr[v + 1]; t += 16) acc += slot_nact(adj_seg[t]);
        acc = group16_sum(acc);
        if (v < n && l16 == 0) rowsum[v] = (uint32_t)acc;
    }
}
// moff[t] = rowbase[v] + (entries of the earlier slots of row v).  16 lanes per row, each walking one contiguous sixteenth of
// the row's slots (offsets by a 16-lane exclusive scan of the chunk totals)
__global__ __launch_bounds__(256) void k_midx_fill(const int32_t* rowptr, const int2* adj_seg, const uint32_t* rowbase,
                                                   uint32_t* moff, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    for (int vb = row0 - (row0 % 4); vb < n; vb += nrows) {      // the 4 rows of a wave advance together (full-wave shuffles)
        const int v = vb + (row0 % 4);
        int t0 = 0, t1 = 0;
        if (v < n) {
            const int r0 = rowptr[v], r1 = rowptr[v + 1], chunk = (r1 - r0 + 15) / 16;
            t0 = min(r0 + l16 * chunk, r1); t1 = min(t0 + chunk, r1);
        }
        int mine = 0;
        for (int t = t0; t < t1; ++t) mine += slot_nact(adj_seg[t]);
        int incl = mine;                                          // inclusive scan over the 16 lanes of the row
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { const int up = __shfl_up(incl, d, 16); if (l16 >= d) incl += up; }
        if (v >= n) continue;
        uint32_t o = rowbase[v] + (uint32_t)(incl - mine);
        for (int t = t0; t < t1; ++t) { moff[t] = o; o += (uint32_t)slot_nact(adj_seg[t]); }
    }
}
// ... and the entries themselves: 16 lanes per CSR slot, lanes over the slot's contributing cycles (coalesced both ways)
__global__ __launch_bounds__(256) void k_midx_entries(const int2* adj_seg, const uint32_t* moff, const uint32_t* pk, uint16_t* midx, int64_t nslots) {
    const int l16 = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4, ng = ((int64_t)gridDim.x * 256) >> 4;
    for (int64_t t = g0; t < nslots; t += ng) {
        const int2 rec = adj_seg[t];
        const bool v_is_i = rec.y < 0;
        const int nact = slot_nact(rec);
        const uint32_t o = moff[t];
        for (int q = l16; q < nact; q += 16) {
            const uint32_t p = pk[(int64_t)rec.x + q];
            midx[(size_t)o + q] = (uint16_t)((v_is_i ? p : p >> 16) & 0x7FFFu);
        }
    }
}

// ---- exchange layout of the sharded runs (round 3) -------------------------------------------------------------------
// Rank r owns the nodes [node_lo[r], node_lo[r+1]) (whole bands) and with them the edges (i, j), i < j, whose SMALLER endpoint it owns:
// a contiguous range [e_lo[r], e_lo[r+1]) of the (i,j)-sorted edge list.  Its part of the reduce-scatter buffer (t_part = 2 t_half doubles):
//   [0, t_half)        T1 of its edges, in edge order            -> node i's column-sum workgroup writes the T1 of ALL its larger
//                                                                   neighbours as ONE contiguous run (CSR order of row i = edge order)
//   [t_half, 2 t_half) T2 of its edges, ordered by (j, i)        -> node j's workgroup writes its smaller neighbours' T2 as at most
//                                                                   `world` contiguous runs (the smaller neighbours owned by one rank
//                                                                   are consecutive in row j)
// (round 2 ordered both halves by the sweep's segment order (band, j, i): every rank wrote all 2m slots 8 bytes at a time, +58 us per
//  iteration at C4 whatever the number of ranks).  The all-gather slice of a rank = S of its edges in edge order, so the unpack copies
//  the larger-neighbour half of every CSR row contiguously and gathers the other half.
// prefB[r * n + v] = number of T2 entries of owner r that precede node v's run = sum over v' < v of |{u < v', u owned by r}|
__device__ __forceinline__ int nbrs_below(const int32_t* adj, int r0, int r1, int x) {       // neighbours of the row smaller than x
    int lo = 0, hi = r1 - r0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (adj[r0 + mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(1024) void k_prefB(const int32_t* rowptr, const int32_t* adj, const int32_t* node_lo, int32_t* prefB, int n) {
    __shared__ int sa[1024];
    __shared__ int carry;
    const int r = blockIdx.x, vlo = node_lo[r], vhi = node_lo[r + 1];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int v = base + threadIdx.x;
        int c = 0;
        if (v < n) { const int r0 = rowptr[v], r1 = rowptr[v + 1]; c = nbrs_below(adj, r0, r1, min(vhi, v)) - nbrs_below(adj, r0, r1, min(vlo, v)); }
        sa[threadIdx.x] = c;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int x = threadIdx.x >= d ? sa[threadIdx.x - d] : 0;
            __syncthreads();
            sa[threadIdx.x] += x;
            __syncthreads();
        }
        if (v < n) prefB[(int64_t)r * n + v] = carry + sa[threadIdx.x] - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sa[1023];
        __syncthreads();
    }
}
// Sharded column sums: the run of row v's CSR slots whose segments {v, u} belong to the rank that owns the nodes [lo, hi) -- those with
// min(v, u) in [lo, hi): u >= lo when v itself is owned, lo <= u < hi when v >= hi, none when v < lo.  Neighbours are ascending: one run.
__global__ __launch_bounds__(256) void k_node_runs(const int32_t* rowptr, const int32_t* adj, int lo, int hi, int2* runs, int n) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const int r0 = rowptr[v], r1 = rowptr[v + 1];
    int2 run{0, 0};
    if (v >= lo) {
        run.x = nbrs_below(adj, r0, r1, lo);
        run.y = v < hi ? r1 - r0 : nbrs_below(adj, r0, r1, hi);
    }
    runs[v] = run;
}
__device__ __forceinline__ int owner_of_node(const int32_t* node_lo, int world, int v) { int r = 0; while (r + 1 < world && v >= node_lo[r + 1]) ++r; return r; }
// Exchange parts (round 4): the owners of the layout are VIRTUAL -- vo = rank * xparts + part, node ranges node_lo[vo .. vo + 1), nv = world * xparts of
// them -- and the send buffer is part-major: block (part * world + rank) holds [T1 | T2] of that virtual owner's edges, so that the reduce-scatter of
// part c is ONE collective over the contiguous blocks [c * world, (c + 1) * world) and can travel while part c - 1 is swept.  xparts = 1: round 3's layout.
__device__ __forceinline__ int xblock_of(int vo, int xparts, int world) { return (vo % xparts) * world + vo / xparts; }
// xpos[t]: where the column sum of CSR slot t = (v, u) goes in the send buffer; spos[t]: where S of edge {v, u} sits in the gathered slices (per REAL rank)
__global__ __launch_bounds__(256) void k_xpos(const int32_t* rowptr, const int32_t* adj, const int32_t* adj_eid, const int32_t* node_lo, const int32_t* e_lo,
                                              const int32_t* prefB, int nv, int xparts, int world, int64_t t_part, int64_t slice_len, int32_t* xpos, int32_t* spos, int n) {
    const int l16 = threadIdx.x & 15;
    const int row0 = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    const int64_t t_half = t_part / 2;
    for (int v = row0; v < n; v += nrows) {
        const int r0 = rowptr[v], r1 = rowptr[v + 1];
        const int ov = owner_of_node(node_lo, nv, v);
        for (int t = r0 + l16; t < r1; t += 16) {
            const int u = adj[t], e = adj_eid[t];
            if (u > v) {                                         // T1 of edge (v, u); S of the edge: both with the owner of v
                xpos[t] = (int32_t)((int64_t)xblock_of(ov, xparts, world) * t_part + (e - e_lo[ov]));
                spos[t] = (int32_t)((int64_t)(ov / xparts) * slice_len + (e - e_lo[(ov / xparts) * xparts]));
            } else {                                             // column u of node v = T2 of edge (u, v), owned by the owner of u
                const int ou = owner_of_node(node_lo, nv, u);
                const int first = nbrs_below(adj, r0, r1, node_lo[ou]);          // first smaller neighbour of v that ou owns
                xpos[t] = (int32_t)((int64_t)xblock_of(ou, xparts, world) * t_part + t_half + prefB[(int64_t)ou * n + v] + ((t - r0) - first));
                spos[t] = (int32_t)((int64_t)(ou / xparts) * slice_len + (e - e_lo[(ou / xparts) * xparts]));
            }
        }
    }
}
// {ta, tb} of the segments this rank owns (device order): positions of their T1 / T2 inside the block of their virtual owner (= inside the part's
// reduce-scattered sums); ta + (first edge of the part - first edge of the rank) is the segment's place in the rank's all-gather slice
__global__ __launch_bounds__(256) void k_xt(const int32_t* pos_edge2, const EdgeInfo* einfo, const int32_t* ind_i, const int32_t* ind_j, const int32_t* rowptr,
                                            const int32_t* adj, const int32_t* node_lo, const int32_t* e_lo, const int32_t* prefB, int nv, int64_t t_half,
                                            int seg_lo, int seg_hi, int2* xt, int n) {
    for (int q = seg_lo + blockIdx.x * 256 + threadIdx.x; q < seg_hi; q += gridDim.x * 256) {
        const int e = pos_edge2[q], i = ind_i[e], j = ind_j[e];
        const int vo = owner_of_node(node_lo, nv, i);
        const int r0 = rowptr[j], r1 = rowptr[j + 1];
        const int first = nbrs_below(adj, r0, r1, node_lo[vo]);
        xt[q] = int2{e - e_lo[vo], (int)(t_half + prefB[(int64_t)vo * n + j] + ((einfo[q].slot_b - r0) - first))};
    }
}

__global__ __launch_bounds__(256) void k_init_node(const int32_t* cum, const EdgeInfo* einfo, const double* S0,
                                                   double* w, double* S_a, double* S_b, int m_pos, double* s_slice, const int2* xt) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int base = cum[l], cnt = cum[l + 1] - base;
        const double w0 = 1.0 / (double)cnt;
        double s = 0.0;
        for (int t = lane; t < cnt; t += 64) { w[(int64_t)base + t] = w0; s += w0 * S0[(int64_t)base + t]; }
        s = group_sum<64>(s);
        if (lane == 0) {
            const EdgeInfo ei = einfo[l];
            S_a[ei.slot_a] = s; S_a[ei.slot_b] = s; S_b[ei.slot_a] = s; S_b[ei.slot_b] = s;
            if (s_slice) s_slice[xt[l].x] = s;
        }
    }
}

__global__ __launch_bounds__(256) void k_objective_node(const int32_t* cum, const EdgeInfo* einfo, const uint32_t* pk,
                                                        const double* w, const double* S, int m_pos, double* partials,
                                                        const DevState* st) {
    if (st->stop) return;
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
    double acc = 0.0;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int base = cum[l], cnt = cum[l + 1] - base;
        const EdgeInfo ei = einfo[l];
        for (int t = lane; t < cnt; t += 64) {
            const uint32_t p = pk[(int64_t)base + t];
            acc += w[(int64_t)base + t] * (S[ei.rb_j + (int)((p >> 16) & 0x7FFFu)] + S[ei.rb_i + (int)(p & 0x7FFFu)]);
        }
    }
    block_partials(acc, 0.0, partials, blockIdx.x);
}

// S_vec in the caller's edge order from the CSR-aligned copy
__global__ void k_extract_S(const double* Sfull, const int32_t* eslot, double* S_vec, int64_t m) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x)
        S_vec[e] = Sfull[eslot[e]];
}
// per-cycle vector between natural order and device order (dir 0: natural -> device, 1: device -> natural)
__global__ __launch_bounds__(256) void k_reorder_cycles(const int32_t* cum, const int32_t* src_start, const uint8_t* seg_perm,
                                                        const double* in, double* out, int m_pos, int dir) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t l = wid; l < m_pos; l += nw) {
        const int base = cum[l], cnt = cum[l + 1] - base, src = src_start[l];
        for (int t = lane; t < cnt; t += 64) {
            const int o = seg_perm[(int64_t)base + t];          // natural offset of the cycle stored at device slot t
            if (dir == 0) out[(int64_t)base + t] = in[(int64_t)src + o];
            else out[(int64_t)src + o] = in[(int64_t)base + t];
        }
    }
}

// ---- multi-GPU exchange helpers (node variant) --------------------------------
// A rank's all-gather slice = [S of the edges it owns, padded to the largest shard | SHARD_PARTS pairs of workgroup
// partials (objective, sum |dS|)].  The sweep kernels write both straight into the slice; after the all-gather
// k_unpack_S scatters S of every edge into both of its CSR slots, and its extra last workgroup adds all ranks'
// partials in rank order (identical sums, hence identical stop decisions, on every rank) and runs the stop rule.
constexpr int SHARD_PARTS = 1024;           // >= band grid + tail pieces (MAX_TAIL_PIECES)
__global__ __launch_bounds__(256) void k_unpack_S(const int32_t* spos, int64_t nslots, const double* sall, double* S_a, double* S_b, FinArgs fin) {
    if (blockIdx.x == 0) {                      // the bookkeeping workgroup: dispatched first, runs under the copy
        if (fin.t > 0) finalize_block(fin);
        return;
    }
    // once the stop rule has fired the slices hold the discarded sweep: the double buffers must keep the final iterate
    if (fin.st->stop || !S_a) return;
    // every CSR slot from its place in the gathered slices: coalesced writes; reads contiguous for the larger-neighbour half of a row
    // (four independent position -> value chains in flight per thread.  Round 4, rocprofv3 over the eight emulated ranks of C4 on one card,
    //  profiles/r04_shard_w8_c4_rocprof.txt: 73 us on average but 35 us at best -- the eight ranks' tables (8 x 80 MB) evict each other from the
    //  256 MB Infinity Cache between a rank's turns, which a rank on its own GPU does not suffer.  A tile-ordered form -- slots visited
    //  (tile of source nodes)-major so that the gathered values stay in the L2 -- measured 81 us on average, 53 at best: not adopted.)
    const int64_t stride = (int64_t)(gridDim.x - 1) * 256;
    for (int64_t t = (int64_t)(blockIdx.x - 1) * 256 + threadIdx.x; t < nslots; t += 4 * stride) {
        int32_t q[4]; double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = spos[min(t + u * stride, nslots - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = sall[q[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t + u * stride < nslots) {
                S_a[t + u * stride] = v[u];
                if (S_b) S_b[t + u * stride] = v[u];
            }
    }
}

// ===========================================================================
// shared small kernels
// ===========================================================================
__global__ void k_fill(double* p, int64_t n, double v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// self-test kernel for the group reductions (tests/test_gpu_parity.py)
__global__ void k_selftest_group_sum(const double* in, double* out, int G) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const double v = in[t];
    out[t] = (G == 16) ? group_sum<16>(v) : (G == 32) ? group_sum<32>(v) : group_sum<64>(v);
}

}  // namespace desc

using namespace desc;

// ------------------------------------------------------------------- handle --
enum { VARIANT_GATHER = 1, VARIANT_NODE = 2 };

struct desc_pgd {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0, m = 0, m_pos = 0, m_cycle = 0;
    int32_t max_cnt = 0, n_sample = 0, max_deg = 0;
    int variant = VARIANT_GATHER;
    int G = 64;                 // gather variant: lanes per segment (0 = big fallback); node variant: cycles per lane
    int lps = 16;               // node variant: lanes per segment
    bool band_jmajor = false;   // band sweep: j-block-major units (large graphs) instead of contiguous ranges
    int grid = 0;               // sweep grid (multiple of 8)
    int obj_grid = 0;
    int colsum_grid = 0, colsum_stride = 0, colsum_cl = 16;     // colsum_cl: lanes per segment in k_colsum_node (8 when the runs are short)
    int colsum_fx_bits = 47;                                     // fixed-point position of the column sums: 62 - ceil(log2(max degree + 1))
    int colsum_nodes = 0;                                        // nodes k_colsum_node visits (sharded: those with segments of this rank in their row)
    int colsum_long = 0;                                         // nodes that get a whole workgroup in k_colsum_node (the rest: a wave each, sharded runs)
    int64_t colsum_entries = 0;                                  // (cycle, endpoint) pairs the column sums read: cycles whose mirror was sampled, per endpoint
    int64_t n_pieces = 0, n_bands = 0, piece_row_entries = 0;    // band sweep plan: pieces, bands, CSR entries of band rows loaded per sweep
    int band = 0;
    bool band_ok = false;       // k_sweep_band applies (segments <= 64 cycles, rows fit the LDS)
    bool small_ok = false;      // k_sweep_small: one rank, < 2 M cycles, segments of at most 64 cycles (the reference's demo sizes)
    int small_grid = 0, small_g = 64;
    int band_grid = 0, band_rows = 0;
    size_t band_lds = 0;
    PieceDesc* d_pieces = nullptr;
    int32_t* d_piece_ptr = nullptr;
    unsigned long long* d_wg_clock = nullptr;   // diagnostics: DESC_DEBUG_WGCLOCK
    int band_tail_first = 0, band_ntail = 0;    // shared tail of the band sweep (BandSweepArgs)
    int32_t* d_ticket = nullptr;
    int32_t* d_node_order = nullptr;            // column sums: the nodes gone through (one workgroup each, dispatched in index order); sharded: only those with owned segments
    int2* d_node_run = nullptr;                 // sharded runs: per node the run [x, y) of its row's CSR slots whose segments this rank owns (NULL: whole rows)
    hvec<void*> allocs;
    int uc_default = 0;         // which streamed arrays go to uncached memory (dalloc_stream)
    // common
    int32_t* d_cum = nullptr;
    double *d_S0 = nullptr, *d_w[2] = {nullptr, nullptr}, *d_S[2] = {nullptr, nullptr};
    double *d_adam_m[2] = {nullptr, nullptr}, *d_adam_v[2] = {nullptr, nullptr}, *d_nv = nullptr, *d_partials = nullptr;
    double *d_obj = nullptr, *d_avg = nullptr, *d_scratch = nullptr;
    DevState* d_state = nullptr;
    // gather variant
    int32_t *d_pos_edge = nullptr, *d_ejk = nullptr, *d_eki = nullptr, *d_ikj = nullptr, *d_jki = nullptr;
    // node variant
    EdgeInfo* d_einfo = nullptr;
    uint32_t* d_pk = nullptr;
    uint32_t* d_moff = nullptr;  // per CSR slot: start of its run in d_midx
    uint16_t* d_midx = nullptr;  // column index of every contributing cycle, in the order the column-sum pass reads them
    int2* d_adj_seg = nullptr;   // per CSR slot: slot_record() = {first contributing cycle of the incident segment, their number | (row node is the smaller endpoint) << 31}
    int32_t *d_rowptr = nullptr, *d_src_start = nullptr, *d_eslot = nullptr;
    ChunkDesc* d_chunk_desc = nullptr;
    int nchunks = 0;
    double *d_T = nullptr, *d_Svec = nullptr;
    uint8_t* d_seg_perm = nullptr;     // natural in-segment offset of every device-order cycle
    // sharding (world == 1: the whole problem)
    int rank = 0, world = 1;
    int64_t seg_lo = 0, seg_hi = 0;     // device-order range of segments owned by this rank
    int64_t cyc_lo = 0, cyc_hi = 0;     // their cycles
    int ch_lo = 0;                      // first chunk owned
    int64_t slice_len = 0;              // doubles per rank in the S exchange buffer
    int64_t slice_S = 0;                // ... of which S values (the rest: SHARD_PARTS pairs of partials)
    desc_collectives coll{};            // fused protocol: the caller's collectives (RCCL entry points + communicator)
    hipStream_t comm_stream = nullptr;  // second stream: exchange + unpack overlap the next column-sum pass
    hipEvent_t ev_col = nullptr, ev_rs = nullptr, ev_sw = nullptr, ev_ag = nullptr;
    StepArgs cur_step{}; bool cur_adam = false;      // sharded sweeps: the step of the sweep in progress (its exchange parts are separate launches)
    int unpack_pending = 0;             // fused protocol: sweep whose all-gathered S still has to be unpacked (on the compute stream, behind ev_ag)
    bool own_xbuf = false, force_coll = false;
    hvec<int64_t> rank_seg;      // world+1 segment boundaries
    double* x_T = nullptr;              // caller-bound exchange buffers (device): owner-sorted partial mirror sums (send)
    double* x_Trecv = nullptr;          // reduce-scattered mirror sums of the owned segments
    int32_t* d_xpos = nullptr;          // 2m: CSR slot -> position of its column sum in x_T (k_xpos)
    int32_t* d_spos = nullptr;          // 2m: CSR slot -> position of its edge's S in the gathered slices x_sall
    int2* d_xt = nullptr;               // m_pos (device order; owned segments filled): {ta, tb} = places of T1 / T2 in x_Trecv, ta also in the slice
    int64_t t_part = 0;                 // words per block of the exchange layout (one block per virtual owner = (rank, part))
    int xparts = 1;                     // exchange parts per rank (NodePlan::xparts): one reduce-scatter and one sweep launch each
    hvec<int64_t> xseg;                 // xparts + 1: device-order segment boundaries of this rank's parts
    hvec<int64_t> xslice_off;           // xparts: first edge of the part - first edge of the rank (offset of the part's S in the rank's slice)
    hvec<int> xpiece_base, xpiece_ptr_base, xtail_first, xntail;      // per part: where its pieces / piece_ptr start in d_pieces / d_piece_ptr, its shared tail
    hipEvent_t ev_rsx[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // fused protocol: reduce-scatter of part c done
    double* x_sall = nullptr;           //   world * slice_len
    bool borrowed_stream = false, objective_done = false;
    int last_parts = 0;                 // workgroup partials the last sharded sweep wrote
    int final_obj_T = -1;               // sweep count for which download already evaluated the objective
    int pending_fin = 0, pending_parts = 0;   // sweep whose bookkeeping (k_finalize) rides on the next column-sum launch
    int32_t* d_rank_seg = nullptr;
    int trace_cap = 0;          // allocated length of d_obj / d_avg (the largest budget this handle has seen)
    int iters_cap = 0;          // max(1, params.iters) of the last reset: what the caller's trace buffers hold
    // run state
    desc_params p{};
    bool armed = false;
    int t_done = 0;             // sweeps enqueued since reset
    int t_plugin = 0;           // plugin counter (PiecewiseStepSize.t / HybridGradient.t)
    double ms_upload = 0, ms_cycle_d = 0, ms_pgd = 0;
    std::string kname;
    std::string last_sweep;     // the sweep instance launched last, with its template arguments (desc_debug_last_sweep: tests assert which kernel ran)
};

namespace {

template <class T>
int dalloc(desc_pgd* h, T** p, size_t count) {
    *p = nullptr;
    void* q = nullptr;
    DESC_HIP(dev_alloc(&q, sizeof(T) * (count > 0 ? count : 1)));
    h->allocs.push_back(q);
    *p = (T*)q;
    return DESC_OK;
}
// The arrays the band sweep streams through exactly once per launch and never writes -- S0 (8 B per cycle) and the packed words (4 B) --
// live in UNCACHED device memory (MTYPE UC; devmem.hip): their lines do not stay in the L2, where the rows of S the sweep gathers from then
// keep their place.  Measured (profiles/r03_uncached_streams.txt, rocprofv3 sweep averages in one call): C4 1170 -> 1130 us (-3.4 %); the
// weights too (read, written, read again by the column sums): +17 %, not done; C5 / C2 / C3: see the file.  DESC_DEBUG_UNCACHED overrides
// the mask (1 weights, 2 S0, 4 packed words, 8 the column-index stream of the column sums, 16 their slot records; default 6 on graphs
// whose S does not fit the L2s, else 0).
template <class T>
int dalloc_stream(desc_pgd* h, T** p, size_t count, int bit) {
    const char* ev = std::getenv("DESC_DEBUG_UNCACHED");
    const int mask = ev ? std::atoi(ev) : h->uc_default;
    if (!(mask & bit)) return dalloc(h, p, count);
    *p = nullptr;
    void* q = nullptr;
    DESC_HIP(dev_alloc_uncached(&q, sizeof(T) * (count > 0 ? count : 1)));
    h->allocs.push_back(q);
    *p = (T*)q;
    return DESC_OK;
}
void dfree(desc_pgd* h, void* q) {
    if (!q) return;
    for (auto& x : h->allocs) if (x == q) { x = nullptr; break; }
    dev_free(q);
}

int set_device(const desc_pgd* h) { DESC_HIP(hipSetDevice(h->device)); return DESC_OK; }
int64_t local_cycles(const desc_pgd* h) { return h->variant == VARIANT_NODE ? h->cyc_hi - h->cyc_lo : h->m_cycle; }

void free_all(desc_pgd* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();                              // once for all blocks and both streams (dev_free would wait per block)
    for (void* q : h->allocs) dev_free_idle(q);
    for (hipEvent_t e : {h->ev_col, h->ev_rs, h->ev_sw, h->ev_ag}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_rsx) if (e) (void)hipEventDestroy(e);
    stream_release(h->comm_stream);
    if (!h->borrowed_stream) stream_release(h->stream);
    delete h;
}

template <class T>
int upload(desc_pgd* h, T* dst, const T* src, size_t count) {
    if (count) DESC_HIP(hipMemcpyAsync(dst, src, sizeof(T) * count, hipMemcpyHostToDevice, h->stream));
    return DESC_OK;
}

StepArgs make_step(desc_pgd* h, bool* adam, int rd, int wr) {
    const desc_params& p = h->p;
    StepArgs s{};
    s.adam_m = h->d_adam_m[rd]; s.adam_v = h->d_adam_v[rd]; s.adam_m_out = h->d_adam_m[wr]; s.adam_v_out = h->d_adam_v[wr];
    // one GetStep call per iteration (DESC_PGD.m:207): the plugin counter advances first
    const int tp = ++h->t_plugin;
    s.lr = p.lr; s.beta1 = p.beta1; s.beta2 = p.beta2; s.bc1 = 1.0; s.bc2 = 1.0;
    s.step = p.lr;
    *adam = false;
    if (p.step_kind == DESC_STEP_PIECEWISE) {
        s.step = p.lr / (std::trunc((double)tp / p.decay_interval) + 1.0);            // PiecewiseStepSize.m:16
    } else if (p.step_kind == DESC_STEP_HYBRID) {
        if (p.hybrid_strategy == 0) {
            *adam = true;
            s.bc1 = 1.0 - std::pow(p.beta1, (double)tp);                             // HybridGradient.m:32-33
            s.bc2 = 1.0 - std::pow(p.beta2, (double)tp);
        } else {
            s.step = 100.0 * (p.lr / (std::trunc((double)tp / p.decay_interval) + 1.0));   // HybridGradient.m:39
        }
    }
    return s;
}

template <int STEP>
void launch_gather(desc_pgd* h, const SweepArgs& a) {
    dim3 grid(h->grid), block(256);
    { char nm[64]; if (h->G) snprintf(nm, sizeof nm, "k_sweep<%d,%d>", h->G, STEP); else snprintf(nm, sizeof nm, "k_sweep_big<%d>", STEP); h->last_sweep = nm; }
    switch (h->G) {
        case 16: hipLaunchKernelGGL((k_sweep<16, STEP>), grid, block, 0, h->stream, a); break;
        case 32: hipLaunchKernelGGL((k_sweep<32, STEP>), grid, block, 0, h->stream, a); break;
        case 64: hipLaunchKernelGGL((k_sweep<64, STEP>), grid, block, 0, h->stream, a); break;
        default: hipLaunchKernelGGL((k_sweep_big<STEP>), grid, block, 0, h->stream, a); break;
    }
}
template <int STEP>
void launch_node(desc_pgd* h, const NodeSweepArgs& a) {
    dim3 grid(h->grid), block(SWEEP_THREADS);
    {
        int L = h->lps, E = h->G;
        if (!((L == 16 && (E == 1 || E == 2)) || (L == 32 && (E == 2 || E == 4)))) { L = 64; E = 4; }      // the switch's default
        char nm[64]; snprintf(nm, sizeof nm, "k_sweep_node<%d,%d,%d>%s", L, E, STEP, a.xt ? " sharded" : ""); h->last_sweep = nm;
    }
    switch (h->lps * 8 + h->G) {              // lps lanes per segment, G = cycles per lane (E)
        case 16 * 8 + 1: hipLaunchKernelGGL((k_sweep_node<16, 1, STEP>), grid, block, 0, h->stream, a); break;
        case 16 * 8 + 2: hipLaunchKernelGGL((k_sweep_node<16, 2, STEP>), grid, block, 0, h->stream, a); break;
        case 32 * 8 + 2: hipLaunchKernelGGL((k_sweep_node<32, 2, STEP>), grid, block, 0, h->stream, a); break;
        case 32 * 8 + 4: hipLaunchKernelGGL((k_sweep_node<32, 4, STEP>), grid, block, 0, h->stream, a); break;
        default: hipLaunchKernelGGL((k_sweep_node<64, 4, STEP>), grid, block, 0, h->stream, a); break;
    }
}

// Band sweep instances by the longest segment: lanes per segment x cycles per lane, threads per workgroup.
//   <= 16 cycles: 16 x 1, 1024     <= 32: 16 x 2, 512 (8 x 4, 512 on small graphs)     <= 64: 16 x 4, 512     <= 128: 32 x 4, 512     <= 256: 64 x 4, 512
// (33..64 cycles: 8 waves with 4 cycles per lane beat 16 waves of 32 x 2 by 5 % at C2 and C4 -- the DPP reductions are shared by four
//  segments per wave instead of two.  17..32 cycles: 8 lanes x 4 cycles, eight segments per wave with their records in per-lane vector
//  loads (they would spill the SGPRs), gains 4-5 % where S stays in the L2 (C3) and loses 1-2 % at C5; there 16 x 2 on 8 waves beats
//  the same shape on 16 waves by 2-4 % (12 waves: 1 %; 4 waves lose 7-22 % everywhere): fewer waves keep more of the j rows in the caches --
//  the PMC traffic of the C4 sweep fell from 6.2 to 5.6 GB with the 8-wave shape.)
// The Adam plugin: 512-thread instances (the moments ride in the stream sets) up to 64 cycles: 16 x 1, 16 x 2, 32 x 2.
struct BandShape { int lps, E; };
BandShape band_shape(const desc_pgd* h, bool adam) {
    const int c = h->max_cnt;
    if (adam) return c <= 16 ? BandShape{16, 1} : c <= 32 ? BandShape{16, 2} : BandShape{32, 2};
    if (c > 16 && c <= 32 && !h->band_jmajor) return BandShape{8, 4};      // graphs whose S stays in the L2 (contiguous ranges)
    return c <= 16 ? BandShape{16, 1} : c <= 32 ? BandShape{16, 2} : c <= 64 ? BandShape{16, 4} : c <= 128 ? BandShape{32, 4} : BandShape{64, 4};
}
bool band_adam_ok(const desc_pgd* h) { return h->band_ok && h->max_cnt <= 64; }
template <int STEP, bool XT>
const void* band_kernel(const desc_pgd* h) {
    const BandShape sh = band_shape(h, STEP == DESC_STEP_HYBRID);
    if constexpr (STEP == DESC_STEP_HYBRID) {
        if (h->max_cnt > 64) return nullptr;           // 4 cycles per lane + the moments do not fit the registers: k_sweep_node (band_adam_ok)
        switch (sh.lps * 8 + sh.E) {
            case 16 * 8 + 1: return (const void*)k_sweep_band<16, 1, STEP, 512, XT>;
            case 16 * 8 + 2: return (const void*)k_sweep_band<16, 2, STEP, 512, XT>;
            default: return (const void*)k_sweep_band<32, 2, STEP, 512, XT>;
        }
    } else {
        switch (sh.lps * 8 + sh.E) {
            case 16 * 8 + 1: return (const void*)k_sweep_band<16, 1, STEP, 1024, XT>;
            case 16 * 8 + 2: return (const void*)k_sweep_band<16, 2, STEP, 512, XT>;
            case 16 * 8 + 4: return (const void*)k_sweep_band<16, 4, STEP, 512, XT>;
            case 8 * 8 + 4: return (const void*)k_sweep_band<8, 4, STEP, 512, XT>;
            case 32 * 8 + 4: return (const void*)k_sweep_band<32, 4, STEP, 512, XT>;
            default: return (const void*)k_sweep_band<64, 4, STEP, 512, XT>;
        }
    }
}
template <int LPS, int E, int STEP, int NT>
void launch_band_shape(desc_pgd* h, const BandSweepArgs& b) {
    { char nm[64]; snprintf(nm, sizeof nm, "k_sweep_band<%d,%d,%d,%d,%s>", LPS, E, STEP, NT, b.n.xt ? "XT" : "one-rank"); h->last_sweep = nm; }
    if (b.n.xt) hipLaunchKernelGGL((k_sweep_band<LPS, E, STEP, NT, true>), dim3(h->band_grid), dim3(NT), h->band_lds, h->stream, b);
    else hipLaunchKernelGGL((k_sweep_band<LPS, E, STEP, NT, false>), dim3(h->band_grid), dim3(NT), h->band_lds, h->stream, b);
}
template <int STEP>
void launch_band(desc_pgd* h, const NodeSweepArgs& a, int part = 0) {
    const BandShape sh = band_shape(h, STEP == DESC_STEP_HYBRID);
    // the plan of exchange part `part` (one-rank handles: the only one)
    BandSweepArgs b{a, h->d_pieces + h->xpiece_base[part], h->d_piece_ptr + h->xpiece_ptr_base[part], h->band_rows, h->d_wg_clock, h->xtail_first[part], h->xntail[part],
                    h->d_ticket + part};
    if constexpr (STEP == DESC_STEP_HYBRID) {
        switch (sh.lps * 8 + sh.E) {
            case 16 * 8 + 1: launch_band_shape<16, 1, STEP, 512>(h, b); break;
            case 16 * 8 + 2: launch_band_shape<16, 2, STEP, 512>(h, b); break;
            default: launch_band_shape<32, 2, STEP, 512>(h, b); break;
        }
    } else {
        switch (sh.lps * 8 + sh.E) {
            case 16 * 8 + 1: launch_band_shape<16, 1, STEP, 1024>(h, b); break;
            case 16 * 8 + 2: launch_band_shape<16, 2, STEP, 512>(h, b); break;
            case 16 * 8 + 4: launch_band_shape<16, 4, STEP, 512>(h, b); break;
            case 8 * 8 + 4: launch_band_shape<8, 4, STEP, 512>(h, b); break;
            case 32 * 8 + 4: launch_band_shape<32, 4, STEP, 512>(h, b); break;
            default: launch_band_shape<64, 4, STEP, 512>(h, b); break;
        }
    }
}
template <int STEP>
void launch_small(desc_pgd* h, const NodeSweepArgs& a) {
    const dim3 grid(h->small_grid), block(256);
    { char nm[64]; snprintf(nm, sizeof nm, "k_sweep_small<%d,%d>", h->small_g, STEP); h->last_sweep = nm; }
    const int ns = (int)(h->seg_hi - h->seg_lo);
    switch (h->small_g) {
        case 16: hipLaunchKernelGGL((k_sweep_small<16, STEP>), grid, block, 0, h->stream, a, ns); break;
        case 32: hipLaunchKernelGGL((k_sweep_small<32, STEP>), grid, block, 0, h->stream, a, ns); break;
        default: hipLaunchKernelGGL((k_sweep_small<64, STEP>), grid, block, 0, h->stream, a, ns); break;
    }
}
void launch_sweep_node_layout(desc_pgd* h, const NodeSweepArgs& a, bool adam, int part = 0) {
    if (h->small_ok && a.partials == h->d_partials) { if (adam) launch_small<DESC_STEP_HYBRID>(h, a); else launch_small<DESC_STEP_CONSTANT>(h, a); }      // the plain one-rank path only
    else if (h->band_ok && !adam) launch_band<DESC_STEP_CONSTANT>(h, a, part);
    else if (adam && band_adam_ok(h)) launch_band<DESC_STEP_HYBRID>(h, a, part);
    else if (adam) launch_node<DESC_STEP_HYBRID>(h, a);
    else launch_node<DESC_STEP_CONSTANT>(h, a);
}
// workgroups (= partial pairs) of the sweep kernel that serves this step kind
int sweep_parts(const desc_pgd* h, bool adam, bool plain_path = false) {
    if (plain_path && h->variant == VARIANT_NODE && h->small_ok) return h->small_grid;
    return h->variant == VARIANT_NODE && (adam ? band_adam_ok(h) : h->band_ok) ? h->band_grid + h->band_ntail : h->grid;
}

// second half of d_partials: the objective kernel of a download writes there, so the partials of the last sweep stay intact and
// book-keeping that sweep again (a replayed column-sum launch after a flush or a download) rewrites the same numbers
size_t parts_cap(const desc_pgd* h) { return (size_t)std::max(std::max(std::max(h->grid, h->obj_grid), h->band_grid + h->band_ntail), h->small_grid); }
double* obj_partials(const desc_pgd* h) { return h->d_partials + 2 * parts_cap(h); }
FinArgs fin_args(const desc_pgd* h, const double* partials, int nparts, int t, int last_only) {
    return FinArgs{partials, h->d_state, h->d_obj, h->d_avg, h->m, h->p.stop_tol, nparts, t, h->p.patience, last_only, 0, 1};
}
// the bookkeeping of the last enqueued sweep, if it is still waiting for a column-sum launch to ride on
void flush_finalize(desc_pgd* h) {
    if (!h->pending_fin) return;
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, h->stream, fin_args(h, h->d_partials, h->pending_parts, h->pending_fin, 0));
    h->pending_fin = 0;
}

// column copies in the LDS of k_colsum_node: one shared by the workgroup; four (one per wave) when the launch has short-run nodes, a wave each
int colsum_copies(const desc_pgd* h) { return h->colsum_long < h->colsum_nodes ? 4 : 1; }
size_t colsum_lds(const desc_pgd* h) { return (size_t)h->colsum_stride * (colsum_copies(h) * sizeof(unsigned long long) + 3 * sizeof(int)); }
// the column-sum pass over the weights `w` -> T (through xpos: the exchange layout of a sharded run); `fin`: the bookkeeping that rides on it
void launch_colsum(desc_pgd* h, hipStream_t st, const double* w, double* T, const int32_t* xpos, const FinArgs& fin) {
    const dim3 grid(h->colsum_grid + 1), block(256);
    const size_t lds = colsum_lds(h);
    const int copies = colsum_copies(h);
    const double fx_scale = std::ldexp(1.0, h->colsum_fx_bits);
    if (h->colsum_cl == 8)
        hipLaunchKernelGGL(k_colsum_node<8>, grid, block, lds, st, h->d_rowptr, h->d_adj_seg, h->d_moff, h->d_midx, w, T, h->colsum_nodes, h->colsum_stride, h->d_state,
                           xpos, fin, h->d_ticket, h->d_node_order, h->d_node_run, h->colsum_long, copies, fx_scale);
    else
        hipLaunchKernelGGL(k_colsum_node<16>, grid, block, lds, st, h->d_rowptr, h->d_adj_seg, h->d_moff, h->d_midx, w, T, h->colsum_nodes, h->colsum_stride, h->d_state,
                           xpos, fin, h->d_ticket, h->d_node_order, h->d_node_run, h->colsum_long, copies, fx_scale);
}

// enqueue sweep number t (1-based) and its finalize; ev0/ev1 bracket the kernels of the sweep proper
int enqueue_sweep(desc_pgd* h, int t, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    const int rd = (t - 1) & 1, wr = t & 1;
    bool adam = false;
    const StepArgs st = make_step(h, &adam, rd, wr);
    if (ev0) (void)hipEventRecord(ev0, h->stream);
    if (h->variant == VARIANT_NODE) {
        const FinArgs fin = fin_args(h, h->d_partials, h->pending_parts, h->pending_fin, 0);      // the previous sweep's bookkeeping rides on this launch
        launch_colsum(h, h->stream, h->d_w[rd], h->d_T, nullptr, fin);
        h->pending_fin = 0;
        NodeSweepArgs a{};
        a.cum = h->d_cum; a.einfo = h->d_einfo; a.pk = h->d_pk; a.S0 = h->d_S0; a.w_old = h->d_w[rd]; a.w_new = h->d_w[wr];
        a.S_old = h->d_S[rd]; a.S_new = h->d_S[wr]; a.Tfull = h->d_T; a.xt = nullptr; a.nv_tab = h->d_nv; a.partials = h->d_partials;
        a.state = h->d_state; a.st = st; a.chunk_desc = h->d_chunk_desc + h->ch_lo; a.nchunks = h->nchunks; a.max_cnt = h->max_cnt;
        a.csr_bytes = (uint32_t)(16 * h->m); a.t_bytes = a.csr_bytes; a.slice_bytes = 0; a.seg_count = (uint32_t)h->m_pos; a.fx_inv = std::ldexp(1.0, -h->colsum_fx_bits);
        launch_sweep_node_layout(h, a, adam);
    } else {
        SweepArgs a{};
        a.cum = h->d_cum; a.pos_edge = h->d_pos_edge; a.e_jk = h->d_ejk; a.e_ki = h->d_eki; a.ikj = h->d_ikj; a.jki = h->d_jki;
        a.S0 = h->d_S0; a.w_old = h->d_w[rd]; a.w_new = h->d_w[wr]; a.S_old = h->d_S[rd]; a.S_new = h->d_S[wr];
        a.nv_tab = h->d_nv; a.partials = h->d_partials; a.state = h->d_state; a.st = st;
        a.m_pos = (int32_t)h->m_pos;
        if (adam) launch_gather<DESC_STEP_HYBRID>(h, a); else launch_gather<DESC_STEP_CONSTANT>(h, a);
    }
    if (ev1) (void)hipEventRecord(ev1, h->stream);
    h->pending_fin = t; h->pending_parts = sweep_parts(h, adam, true);
    if (h->variant != VARIANT_NODE) flush_finalize(h);         // no column-sum launch to ride on
    DESC_HIP(hipGetLastError());
    return DESC_OK;
}

// the next n sweeps.  (hipGraph replays of 10 captured iterations were built in round 3 and measured no gain -- C1 16.5 vs 16.6 us per
// iteration, C2 0.150 vs 0.152 ms, profiles/r03_graph_ab.txt: each kernel is a chain of dependent memory round trips, the launches were not
// what an iteration waits for -- and removed in round 4.)
int enqueue_iterations(desc_pgd* h, int n) {
    for (; n > 0; --n) { const int rc = enqueue_sweep(h, ++h->t_done); if (rc) return rc; }
    return DESC_OK;
}

int choose_grid(desc_pgd* h) {
    const int epw = h->G ? 64 / h->G : 1;
    int64_t want = (h->m_pos + 4 * epw - 1) / (4 * epw);
    want = std::min<int64_t>(want, 2048);
    want = std::max<int64_t>(want, 8);
    h->grid = (int)((want + 7) / 8 * 8);
    return DESC_OK;
}

// ---------------------------------------------------------------- GATHER setup
int setup_gather(desc_pgd* h, const desc_problem* prob, const desc_structure* s, const double* shared_rij) {
    const int64_t m = h->m, mp = h->m_pos, mc = h->m_cycle;
    int rc;
    if ((rc = structure_ensure_host(const_cast<desc_structure*>(s)))) return rc;
    if ((rc = dalloc(h, &h->d_pos_edge, mp))) return rc;
    if ((rc = dalloc(h, &h->d_ejk, mc))) return rc;
    if ((rc = dalloc(h, &h->d_eki, mc))) return rc;
    if ((rc = dalloc(h, &h->d_ikj, mc))) return rc;
    if ((rc = dalloc(h, &h->d_jki, mc))) return rc;
    if ((rc = dalloc(h, &h->d_S[0], m))) return rc;
    if ((rc = dalloc(h, &h->d_S[1], m))) return rc;
    h->G = h->max_cnt <= 16 ? 16 : h->max_cnt <= 32 ? 32 : h->max_cnt <= 64 ? 64 : 0;
    choose_grid(h);
    h->obj_grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (mc + 255) / 256));
    char nm[64];
    if (h->G) snprintf(nm, sizeof nm, "k_sweep<%d,", h->G); else snprintf(nm, sizeof nm, "k_sweep_big<");
    h->kname = nm;

    auto t0 = std::chrono::steady_clock::now();
    hvec<int32_t> cum32((size_t)mp + 1);
    for (int64_t l = 0; l <= mp; ++l) cum32[l] = (int32_t)s->cum_ind[l];
    int32_t *d_k = nullptr, *d_ii = nullptr, *d_jj = nullptr; double* d_rij = nullptr;
    if ((rc = dalloc(h, &d_k, mc))) return rc;
    if ((rc = dalloc(h, &d_ii, m))) return rc;
    if ((rc = dalloc(h, &d_jj, m))) return rc;
    if (shared_rij) d_rij = const_cast<double*>(shared_rij);
    else if ((rc = dalloc(h, &d_rij, 9 * (size_t)m))) return rc;
    if ((rc = upload(h, h->d_cum, cum32.data(), (size_t)mp + 1))) return rc;
    if ((rc = upload(h, h->d_pos_edge, s->pos_edge.data(), (size_t)mp))) return rc;
    if ((rc = upload(h, h->d_ejk, s->e_jk.data(), (size_t)mc))) return rc;
    if ((rc = upload(h, h->d_eki, s->e_ki.data(), (size_t)mc))) return rc;
    if ((rc = upload(h, h->d_ikj, s->ikj.data(), (size_t)mc))) return rc;
    if ((rc = upload(h, h->d_jki, s->jki.data(), (size_t)mc))) return rc;
    if ((rc = upload(h, d_k, s->k.data(), (size_t)mc))) return rc;
    if ((rc = upload(h, d_ii, prob->ind_i, (size_t)m))) return rc;
    if ((rc = upload(h, d_jj, prob->ind_j, (size_t)m))) return rc;
    if (!shared_rij && (rc = upload(h, d_rij, prob->rij, 9 * (size_t)m))) return rc;
    DESC_HIP(hipStreamSynchronize(h->stream));
    h->ms_upload = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    hipEvent_t e0, e1;
    DESC_HIP(hipEventCreate(&e0)); DESC_HIP(hipEventCreate(&e1));
    (void)hipEventRecord(e0, h->stream);
    if (mp > 0) {
        int g = (int)std::min<int64_t>(4096, (mp + 3) / 4);
        hipLaunchKernelGGL(k_cycle_d, dim3(g), dim3(256), 0, h->stream, h->d_cum, h->d_pos_edge, d_ii, d_jj, d_k,
                           h->d_ejk, h->d_eki, d_rij, h->d_S0, (int)mp);
    }
    (void)hipEventRecord(e1, h->stream);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    h->ms_cycle_d = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    dfree(h, d_k); dfree(h, d_ii); dfree(h, d_jj); if (!shared_rij) dfree(h, d_rij);
    if (e != hipSuccess) return fail(DESC_ERR_HIP, "k_cycle_d: %s", hipGetErrorString(e));
    return DESC_OK;
}

int setup_node(desc_pgd* h, const desc_problem* prob, const desc_structure* s, const double* shared_rij) {
    const int64_t n = h->n, m = h->m, mp = h->m_pos;
    auto t0 = std::chrono::steady_clock::now();
    auto t_lap = t0;
    const bool timing = env_int("DESC_DEBUG_TIMING", 0) != 0;
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[desc_amd] setup_node %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    int rc;
    NodePlan P;
    // A structure built on this device brings its CSR, edge tables and sampled k along in HBM: the
    // per-edge tables and the cycle layout are then made by kernels and the host only plans.
    const bool dev_cycles = s->dev == h->device && s->d_k != nullptr;
    double* d_rij = nullptr;
    // node_seg != NULL: the staged form (natural order: one workgroup per node, the node's rotation blocks in the LDS)
    auto launch_layout_dev = [&](const int32_t* cum, const int32_t* src_start, const int32_t* pos, const int32_t* rowptr_d, uint32_t* pk, double* S0, uint8_t* perm,
                                 uint32_t* counts, int64_t nseg, const int32_t* node_seg) -> int {
        const int g = node_seg ? (int)std::max<int64_t>(1, n) : (int)std::min<int64_t>(4096, (nseg + 3) / 4);
        const size_t lds = node_seg ? (size_t)h->max_deg * 10 * sizeof(double) : 0;
        auto go = [&](auto kern) -> int {
            if (lds > 64 * 1024) DESC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(g), dim3(node_seg ? 512 : 256), lds, h->stream, cum, src_start, pos, s->d_ii, s->d_jj, s->d_k, s->d_tau, s->d_ktau, (uint64_t)s->seed, rowptr_d,
                               s->d_bits, s->d_rank, (int)s->words, s->d_adj_eid, d_rij, pk, S0, perm, counts, (int)nseg, node_seg, (int)n);
            return DESC_OK;
        };
        if (node_seg) return h->max_cnt <= 64 ? go(k_layout_node_dev<1, true>) : h->max_cnt <= 128 ? go(k_layout_node_dev<2, true>) : go(k_layout_node_dev<4, true>);
        return h->max_cnt <= 64 ? go(k_layout_node_dev<1, false>) : h->max_cnt <= 128 ? go(k_layout_node_dev<2, false>) : go(k_layout_node_dev<4, false>);
    };
    // One rank: packed words, S0_long (DESC_PGD.m:129-147) and the class order of every segment are computed NOW, in the structure's natural
    // order, under the host-side planning below; k_permute_segments moves them into the band-major order once that exists.  (Round 3 ran this
    // kernel -- 8.7 ms at C4 -- after the plan: the device idled through the planning, then the host through the kernel.)
    uint32_t *d_pk_nat = nullptr, *d_counts_nat = nullptr; double* d_S0_nat = nullptr; uint8_t* d_perm_nat = nullptr;
    const bool early = dev_cycles && h->world == 1 && mp > 0 && env_int("DESC_DEBUG_EARLY_LAYOUT", 1) != 0;
    if (early) {
        if (shared_rij) d_rij = const_cast<double*>(shared_rij);
        else {
            if ((rc = dalloc(h, &d_rij, 9 * (size_t)m))) return rc;
            if (m) DESC_HIP(hipMemcpy(d_rij, prob->rij, sizeof(double) * 9 * (size_t)m, hipMemcpyHostToDevice));
        }
        if ((rc = dalloc(h, &d_pk_nat, h->m_cycle + 8)) || (rc = dalloc(h, &d_S0_nat, h->m_cycle + 8)) || (rc = dalloc(h, &d_perm_nat, h->m_cycle + 8)) ||
            (rc = dalloc(h, &d_counts_nat, mp))) return rc;
        // the rotation blocks of a node's row in the LDS (80 B per slot) when the longest row fits: R_ki is then an LDS read (k_layout_node_dev<., STAGED>)
        int32_t* d_node_seg = nullptr;
        const bool staged = (size_t)h->max_deg * 80 <= 150 * 1024 && env_int("DESC_DEBUG_STAGED_LAYOUT", 1) != 0;
        if (staged) {
            if ((rc = dalloc(h, &d_node_seg, (size_t)n + 1))) return rc;
            hipLaunchKernelGGL(k_node_seg_start, dim3((unsigned)std::min<int64_t>(4096, (mp + 255) / 256)), dim3(256), 0, h->stream, s->d_pos, s->d_ii, mp, (int)n, d_node_seg);
        }
        if (s->ev_fill) DESC_HIP(hipStreamWaitEvent(h->stream, (hipEvent_t)s->ev_fill, 0));
        if ((rc = launch_layout_dev(s->d_cum, s->d_cum, s->d_pos, s->d_rowptr, d_pk_nat, d_S0_nat, d_perm_nat, d_counts_nat, mp, d_node_seg))) return rc;
        DESC_HIP(hipGetLastError());
    }
    // band sweep (i-rows of S in the LDS): segments of up to 64 cycles, every CSR row fits the LDS; DESC_DEBUG_VARIANT=2 keeps
    // the L2-sized bands of k_sweep_node for comparison
    int ncu = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    }
    // (tiny problems stay on k_sweep_node: with < 2 M cycles a 1024-thread workgroup per CU is mostly pipeline fill -- C1: 31 vs 20 us)
    const int force_band = env_int("DESC_DEBUG_VARIANT", 0);
    h->band_ok = force_band != VARIANT_NODE && h->max_cnt <= MAX_SEG_CYCLES && h->max_deg <= BAND_ROW_CAP && (h->m_cycle >= (2 << 20) || force_band == 3) &&
                 (int64_t)2 * m * 8 < (1ll << 32);          // 32-bit byte offsets into the CSR-aligned arrays (buffer instructions): m < 2.7e8 edges
    // exchange parts: the reduce-scatter of part c + 1 travels while part c is swept (band sweep only; DESC_SHARD_PARTS overrides, 1 = one reduce-scatter)
    // (segments of up to 64 cycles: every step plugin then runs on the band sweep; longer ones fall back to k_sweep_node for Adam, which sweeps in one launch)
    const int xparts = (h->world > 1 && h->band_ok && h->max_cnt <= 64) ? std::max(1, std::min(8, env_int("DESC_SHARD_PARTS", 2))) : 1;
    if ((rc = make_node_plan(prob, s, h->max_deg, h->world, h->max_cnt <= 32 ? 32 : h->max_cnt <= 128 ? 16 : 8, h->band_ok ? band_row_cap(h->max_deg) : 0, P, xparts))) return rc;   // 8 waves x 64/lps segments
    h->xparts = P.xparts;
    h->band = P.band;
    lap("plan");
    const hvec<int32_t>& cum2 = P.cum2;
    const int64_t nch_all = (int64_t)P.chunk_seg.size() - 1;
    const int64_t ch_a = P.rank_chunk[h->rank], ch_b = P.rank_chunk[h->rank + 1];
    h->ch_lo = (int)ch_a; h->nchunks = (int)(ch_b - ch_a);
    h->seg_lo = P.chunk_seg[ch_a]; h->seg_hi = P.chunk_seg[ch_b];
    h->cyc_lo = cum2[h->seg_lo]; h->cyc_hi = cum2[h->seg_hi];
    h->rank_seg.assign((size_t)h->world + 1, 0);
    int64_t max_local = 0;
    for (int r = 0; r <= h->world; ++r) h->rank_seg[r] = P.chunk_seg[P.rank_chunk[r]];
    // exchange layout (k_xpos): a (virtual) owner has the edges whose smaller endpoint lies in its node range -- a contiguous range of the sorted edge list
    const int nv = h->world * h->xparts;
    hvec<int32_t> e_lo((size_t)nv + 1, (int32_t)m);
    for (int vo = 0; vo <= nv; ++vo)
        e_lo[vo] = (int32_t)(std::lower_bound(prob->ind_i, prob->ind_i + m, P.vnode[vo]) - prob->ind_i);
    int64_t max_local_v = 0;
    for (int r = 0; r < h->world; ++r) max_local = std::max<int64_t>(max_local, e_lo[(size_t)(r + 1) * h->xparts] - e_lo[(size_t)r * h->xparts]);
    for (int vo = 0; vo < nv; ++vo) max_local_v = std::max<int64_t>(max_local_v, e_lo[vo + 1] - e_lo[vo]);
    h->slice_len = max_local + 2 * SHARD_PARTS;     // S of the owned edges (edge order), then the workgroup partials (see k_unpack_S)
    h->slice_S = max_local;
    h->t_part = 2 * std::max<int64_t>(max_local_v, 1);
    if ((int64_t)nv * h->t_part >= (1ll << 31) - 1 || (int64_t)h->world * h->slice_len >= (1ll << 31) - 1)
        return fail(DESC_ERR_TOO_LARGE, "exchange buffers of %d ranks x %lld edges exceed the 32-bit positions of the exchange layout", h->world, (long long)max_local);
    h->xseg.assign((size_t)h->xparts + 1, 0); h->xslice_off.assign((size_t)h->xparts, 0);
    for (int c = 0; c <= h->xparts; ++c) h->xseg[c] = c < h->xparts ? std::min(std::max(P.vseg[(size_t)h->rank * h->xparts + c], h->seg_lo), h->seg_hi) : h->seg_hi;
    h->xseg[0] = h->seg_lo;
    for (int c = 0; c < h->xparts; ++c) h->xslice_off[c] = e_lo[(size_t)h->rank * h->xparts + c] - e_lo[(size_t)h->rank * h->xparts];
    const int64_t mcl = h->cyc_hi - h->cyc_lo;            // local cycles
    const int64_t nsl = h->seg_hi - h->seg_lo;            // local segments
    // band sweep: the work of every workgroup as a list of pieces (plan_band_pieces)
    hvec<PieceDesc> pieces; hvec<int32_t> piece_ptr;
    if (h->band_ok) {
        h->band_grid = ncu;
        bool jmajor = false;
        // one plan per exchange part (its segments only): a part is swept by a launch of its own
        h->xpiece_base.assign((size_t)h->xparts, 0); h->xpiece_ptr_base.assign((size_t)h->xparts, 0); h->xtail_first.assign((size_t)h->xparts, 0); h->xntail.assign((size_t)h->xparts, 0);
        h->band_rows = 0;
        for (int c = 0; c < h->xparts; ++c) {
            hvec<PieceDesc> pc; hvec<int32_t> pp; int rows_c = 0, tf = 0, nt = 0;
            const int64_t q0 = h->xseg[c], q1 = h->xseg[c + 1];
            const int max_tail = h->world > 1 ? std::max(0, std::min(MAX_TAIL_PIECES, SHARD_PARTS / h->xparts - h->band_grid)) : MAX_TAIL_PIECES;
            plan_band_pieces(prob, s, P, q0, q1, cum2[q0], (int64_t)cum2[q1] - cum2[q0], h->band_grid, pc, pp, rows_c, jmajor, &tf, &nt, max_tail);
            h->xpiece_base[c] = (int)pieces.size(); h->xpiece_ptr_base[c] = (int)piece_ptr.size(); h->xtail_first[c] = tf; h->xntail[c] = nt;
            pieces.insert(pieces.end(), pc.begin(), pc.end());
            piece_ptr.insert(piece_ptr.end(), pp.begin(), pp.end());
            h->band_rows = std::max(h->band_rows, rows_c);
        }
        h->band_tail_first = h->xtail_first[0]; h->band_ntail = h->xntail[0];
        for (int c = 1; c < h->xparts; ++c) h->band_ntail = std::max(h->band_ntail, h->xntail[c]);       // sweep_parts(): the largest part's count
        h->band_jmajor = jmajor;
        h->n_pieces = (int64_t)pieces.size(); h->n_bands = (int64_t)P.band_lo.size() - 1;
        for (const PieceDesc& pd : pieces) h->piece_row_entries += pd.row_len;
        if (timing) fprintf(stderr, "[desc_amd] band sweep: %zu bands, %zu pieces over %d workgroups, %s, rows <= %d\n", P.band_lo.size() - 1, pieces.size(),
                            h->band_grid, jmajor ? "j-block-major units" : "contiguous ranges", h->band_rows);
    }

    // CSR adjacency (neighbours ascending; single pass because Ind is sorted by (i,j))
    hvec<int32_t> rowptr, adj, adj_eid, eslot, eslot_b;
    if (!dev_cycles) {
        rowptr.assign((size_t)n + 1, 0); adj.resize((size_t)2 * m); adj_eid.resize((size_t)2 * m);
        for (int64_t e = 0; e < m; ++e) { rowptr[prob->ind_i[e] + 1]++; rowptr[prob->ind_j[e] + 1]++; }
        for (int64_t v = 0; v < n; ++v) rowptr[v + 1] += rowptr[v];
        // slot of every edge in its two endpoint rows (eslot: smaller endpoint, for S_vec extraction)
        eslot.resize((size_t)m); eslot_b.resize((size_t)m);
        hvec<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
        for (int64_t e = 0; e < m; ++e) {
            const int32_t i = prob->ind_i[e], j = prob->ind_j[e];
            eslot[e] = fill[i]; eslot_b[e] = fill[j];
            adj[fill[i]] = j; adj_eid[fill[i]++] = (int32_t)e;
            adj[fill[j]] = i; adj_eid[fill[j]++] = (int32_t)e;
        }
    }
    lap("csr+eslot");
    // device-order tables of the segments: local cycle starts, start in the structure's natural cycle order, edge id; position of
    // every edge.  Device-built structures: k_seg_tables makes them from the plan's permutation (no host pass, two uploads instead of four)
    hvec<int32_t> cum_loc, src_start, pos_edge2, devpos;
    hvec<EdgeInfo> einfo;
    if (!dev_cycles) {
    cum_loc.resize((size_t)mp + 1); src_start.resize((size_t)mp); pos_edge2.resize((size_t)mp); devpos.assign((size_t)m, -1);
    cum_loc[mp] = cum2[mp] - (int32_t)h->cyc_lo;
    host_parallel(mp, [&](int64_t a, int64_t b) {
        for (int64_t q = a; q < b; ++q) {
            cum_loc[q] = cum2[q] - (int32_t)h->cyc_lo;                              // local cycle numbering (meaningful for owned segments)
            const int32_t l = P.order[q], e = s->pos_edge[l];
            src_start[q] = (int32_t)s->cum_ind[l];
            pos_edge2[q] = e; devpos[e] = (int32_t)q;
        }
    });
    }
    if (!dev_cycles) {
        einfo.resize((size_t)mp);
        host_parallel(mp, [&](int64_t a, int64_t b) {
            for (int64_t q = a; q < b; ++q) {
                const int32_t e = pos_edge2[q], i = prob->ind_i[e], j = prob->ind_j[e];
                einfo[q] = EdgeInfo{rowptr[i], rowptr[j], eslot[e], eslot_b[e]};
            }
        });
    }
    lap("einfo");
    hvec<uint32_t> kf; hvec<uint8_t> seg_perm; hvec<uint32_t> seg_counts; hvec<int2> adj_seg;
    if (!dev_cycles) {
        if ((rc = structure_ensure_host(const_cast<desc_structure*>(s)))) return rc;
    // k with the two mirror-present bits of the owned cycles, device order.  Inside a segment
    // the cycles are stored [(ik;j) only | both mirrors sampled | (jk;i) only | none] (ascending k
    // within a class): the column-sum pass then reads, for either endpoint, ONE run with the cycles that contribute
    // (a fraction ~n_sample/codeg of them), the per-segment arithmetic is order independent.
    kf.assign((size_t)mcl, 0u); seg_perm.assign((size_t)mcl, 0); seg_counts.assign((size_t)mp, 0u);   // counts: n_ionly | n_i << 9 | n_jonly << 18
    host_parallel(nsl, [&](int64_t a, int64_t b) {
        for (int64_t q = h->seg_lo + a; q < h->seg_lo + b; ++q) {
            const int64_t src = src_start[q], dst = cum_loc[q], cnt = cum2[q + 1] - cum2[q];
            int n_cls[4] = {0, 0, 0, 0};                      // class 0 i-only, 1 both, 2 j-only, 3 none
            auto cls = [&](int64_t t) { const bool fi = s->ikj[src + t] >= 0, fj = s->jki[src + t] >= 0; return fi ? (fj ? 1 : 0) : (fj ? 2 : 3); };
            for (int64_t t = 0; t < cnt; ++t) n_cls[cls(t)]++;
            int pos[4] = {0, n_cls[0], n_cls[0] + n_cls[1], n_cls[0] + n_cls[1] + n_cls[2]};
            for (int64_t t = 0; t < cnt; ++t) {
                const int o = pos[cls(t)]++;
                kf[dst + o] = (uint32_t)s->k[src + t] | (s->ikj[src + t] >= 0 ? 1u << 30 : 0u) | (s->jki[src + t] >= 0 ? 1u << 31 : 0u);
                seg_perm[dst + o] = (uint8_t)t;
            }
            seg_counts[q] = (uint32_t)n_cls[0] | (uint32_t)(n_cls[0] + n_cls[1]) << 9 | (uint32_t)n_cls[2] << 18;
        }
    });
    adj_seg.resize((size_t)2 * m);
    host_parallel(n, [&](int64_t a, int64_t b) {
        for (int64_t v = a; v < b; ++v)
            for (int32_t t = rowptr[v]; t < rowptr[v + 1]; ++t) {
                const int32_t q = devpos[adj_eid[t]];
                int2 rec{0, 0};
                if (q >= h->seg_lo && q < h->seg_hi) {       // only segments this rank owns
                    rec = slot_record(cum_loc[q], seg_counts[q], v < adj[t]);
                }
                adj_seg[t] = rec;
            }
    });
    }
    lap("host pack");
    // chunk tables: all chunks, local cycle numbering
    hvec<ChunkDesc> chunk_desc((size_t)std::max<int64_t>(nch_all, 1));
    for (int64_t t = 0; t < nch_all; ++t)
        chunk_desc[t] = ChunkDesc{cum2[P.chunk_seg[t]] - (int32_t)h->cyc_lo, cum2[P.chunk_seg[t + 1]] - (int32_t)h->cyc_lo, P.chunk_seg[t], P.chunk_seg[t + 1]};

    h->uc_default = (h->band_ok && (int64_t)2 * m * 8 > (12ll << 20)) ? 6 : 0;      // S beyond the L2s: S0 and the packed words in uncached memory (dalloc_stream)
    if ((rc = dalloc_stream(h, &h->d_S0, mcl + 8, 2))) return rc;   // +8: 16-byte tail reads
    if ((rc = dalloc_stream(h, &h->d_w[0], mcl + 8, 1))) return rc;
    if ((rc = dalloc_stream(h, &h->d_w[1], mcl + 8, 1))) return rc;
    if ((rc = dalloc_stream(h, &h->d_pk, mcl + 8, 4))) return rc;
    if ((rc = dalloc(h, &h->d_seg_perm, mcl))) return rc;
    if ((rc = dalloc(h, &h->d_einfo, mp))) return rc;
    if ((rc = dalloc(h, &h->d_rowptr, n + 1))) return rc;
    if ((rc = dalloc_stream(h, &h->d_adj_seg, 2 * m, 16))) return rc;
    if ((rc = dalloc(h, &h->d_src_start, mp))) return rc;
    if ((rc = dalloc(h, &h->d_eslot, m))) return rc;
    if ((rc = dalloc_stream(h, &h->d_S[0], 2 * m + 2, 32))) return rc;      // bits 32 / 64: experiments (profiles/r03_experiments.txt), never default; + 2: a 16-byte row load may end one double past 2m
    if ((rc = dalloc_stream(h, &h->d_S[1], 2 * m + 2, 32))) return rc;
    if ((rc = dalloc_stream(h, &h->d_T, 2 * m + 2, 64))) return rc;
    if ((rc = dalloc(h, &h->d_Svec, m))) return rc;
    if ((rc = dalloc(h, &h->d_chunk_desc, chunk_desc.size()))) return rc;
    if ((rc = dalloc(h, &h->d_rank_seg, (size_t)h->world + 1))) return rc;
    if ((rc = dalloc(h, &h->d_xpos, 2 * m))) return rc;
    if ((rc = dalloc(h, &h->d_spos, 2 * m))) return rc;
    if ((rc = dalloc(h, &h->d_xt, mp))) return rc;
    if (h->band_ok) {
        if ((rc = dalloc(h, &h->d_pieces, pieces.size()))) return rc;
        if ((rc = dalloc(h, &h->d_piece_ptr, piece_ptr.size()))) return rc;
        if (env_int("DESC_DEBUG_WGCLOCK", 0) && (rc = dalloc(h, &h->d_wg_clock, 3 * (size_t)h->band_grid))) return rc;
        if ((rc = dalloc(h, &h->d_ticket, 8))) return rc;                    // one ticket counter per exchange part
        DESC_HIP(hipMemsetAsync(h->d_ticket, 0, 8 * sizeof(int32_t), h->stream));
    }
    lap("alloc");
    int32_t *d_ii = nullptr, *d_jj = nullptr, *d_adj = nullptr, *d_adj_eid = nullptr, *d_pos_edge2 = nullptr;
    uint32_t* d_kf = nullptr;
    if (dev_cycles) {               // borrowed from the structure for the duration of this call
        d_ii = s->d_ii; d_jj = s->d_jj; d_adj = s->d_adj; d_adj_eid = s->d_adj_eid;
    } else {
        if ((rc = dalloc(h, &d_ii, m))) return rc;
        if ((rc = dalloc(h, &d_jj, m))) return rc;
        if ((rc = dalloc(h, &d_adj, 2 * m))) return rc;
        if ((rc = dalloc(h, &d_adj_eid, 2 * m))) return rc;
        if ((rc = dalloc(h, &d_kf, mcl))) return rc;
    }
    if ((rc = dalloc(h, &d_pos_edge2, mp))) return rc;
    const bool rij_up = d_rij != nullptr;                             // the early layout already has the rotations on the device
    if (rij_up) {}
    else if (shared_rij) d_rij = const_cast<double*>(shared_rij);     // the device problem's copy: no 72-B-per-edge upload
    else if ((rc = dalloc(h, &d_rij, 9 * (size_t)m))) return rc;
    hvec<int32_t> rank_seg32(h->rank_seg.begin(), h->rank_seg.end());
    int32_t *d_devpos = nullptr, *d_order_keep = nullptr;
    if (dev_cycles && s->ev_fill) {       // the structure's sampled cycles may still be on their way; a fault of that kernel is reported as its own
        const hipError_t ef = hipEventSynchronize((hipEvent_t)s->ev_fill);
        if (ef != hipSuccess) return fail(DESC_ERR_HIP, "cycle sampling kernel failed: %s", hipGetErrorString(ef));
    }
    if (dev_cycles) {
        int32_t* d_order = nullptr;
        if ((rc = dalloc(h, &d_devpos, m)) || (rc = dalloc(h, &d_order, mp))) return rc;
        if ((rc = upload(h, h->d_cum, cum2.data(), (size_t)mp + 1)) || (rc = upload(h, d_order, P.order.data(), (size_t)mp))) return rc;
        DESC_HIP(hipMemsetAsync(d_devpos, 0xFF, sizeof(int32_t) * std::max<int64_t>(1, m), h->stream));
        hipLaunchKernelGGL(k_seg_tables, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (mp + 256) / 256))), dim3(256), 0, h->stream,
                           d_order, s->d_pos, s->d_cum, h->d_cum, (int32_t)h->cyc_lo, h->d_src_start, d_pos_edge2, d_devpos, mp);
        DESC_HIP(hipStreamSynchronize(h->stream));       // d_order, the host sources of the copies
        if (early) d_order_keep = d_order; else dfree(h, d_order);
    } else {
    if ((rc = upload(h, h->d_cum, cum_loc.data(), (size_t)mp + 1))) return rc;
    if ((rc = upload(h, h->d_src_start, src_start.data(), (size_t)mp))) return rc;
    if ((rc = upload(h, d_pos_edge2, pos_edge2.data(), (size_t)mp))) return rc;
    }
    if ((rc = upload(h, h->d_chunk_desc, chunk_desc.data(), chunk_desc.size()))) return rc;
    if ((rc = upload(h, h->d_rank_seg, rank_seg32.data(), rank_seg32.size()))) return rc;
    // (a synchronous copy from pageable memory runs 2-3x faster than an asynchronous one on this runtime)
    if (!shared_rij && !rij_up && m) DESC_HIP(hipMemcpy(d_rij, prob->rij, sizeof(double) * 9 * (size_t)m, hipMemcpyHostToDevice));
    if (h->band_ok) {
        if ((rc = upload(h, h->d_pieces, pieces.data(), pieces.size()))) return rc;
        if ((rc = upload(h, h->d_piece_ptr, piece_ptr.data(), piece_ptr.size()))) return rc;
    }
    if (dev_cycles) {
        DESC_HIP(hipMemcpyAsync(h->d_rowptr, s->d_rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToDevice, h->stream));
        hipLaunchKernelGGL(k_edge_slots, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (m + 255) / 256))), dim3(256), 0, h->stream,
                           d_ii, d_jj, s->d_rowptr, d_adj, d_pos_edge2, h->d_eslot, h->d_einfo, m, mp);
    } else {
        if ((rc = upload(h, h->d_einfo, einfo.data(), (size_t)mp))) return rc;
        if ((rc = upload(h, h->d_rowptr, rowptr.data(), (size_t)n + 1))) return rc;
        if ((rc = upload(h, h->d_adj_seg, adj_seg.data(), (size_t)2 * m))) return rc;
        if ((rc = upload(h, h->d_eslot, eslot.data(), (size_t)m))) return rc;
        if ((rc = upload(h, d_ii, prob->ind_i, (size_t)m))) return rc;
        if ((rc = upload(h, d_jj, prob->ind_j, (size_t)m))) return rc;
        if ((rc = upload(h, d_adj, adj.data(), (size_t)2 * m))) return rc;
        if ((rc = upload(h, d_adj_eid, adj_eid.data(), (size_t)2 * m))) return rc;
        if ((rc = upload(h, d_kf, kf.data(), (size_t)mcl))) return rc;
        if ((rc = upload(h, h->d_seg_perm, seg_perm.data(), (size_t)mcl))) return rc;
    }
    DESC_HIP(hipStreamSynchronize(h->stream));
    h->ms_upload = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    lap("upload");

    // lanes per segment x cycles per lane >= longest segment
    // (33..64 cycles: 32 lanes x 2 cycles keeps all 8 waves of the workgroup busy in the arithmetic --
    //  measured 2 % faster than 16 lanes x 4 cycles at C2 and C4)
    h->lps = h->max_cnt <= 32 ? 16 : h->max_cnt <= 128 ? 32 : 64;
    h->G = h->max_cnt <= 16 ? 1 : h->max_cnt <= 64 ? 2 : 4;
    {   // persistent grid: exactly the workgroups that are co-resident (registers / LDS decide)
        int per_cu = 0;
        const void* kfn = (const void*)k_sweep_node<32, 4, DESC_STEP_CONSTANT>;    // the largest-register instance
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, SWEEP_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        int64_t want = std::min<int64_t>(std::max(h->nchunks, 1), (int64_t)ncu * per_cu);
        h->grid = (int)(std::max<int64_t>(want, 8) + 7) / 8 * 8;
    }
    h->obj_grid = (int)std::min<int64_t>(SHARD_PARTS, std::max<int64_t>(1, (nsl + 3) / 4));     // sharded runs: its partials travel in the all-gather slice
    h->colsum_stride = (h->max_deg + 1) | 1;                 // odd stride: the 4 copies start on different banks
    // one node per workgroup: the dispatcher balances
    // ... in REVERSE node order (round 3).  The sweep before it ends with the last j-blocks, i.e. the weights of the segments with the largest
    // j are the freshest lines in the L2s, and the sweep after it begins with the first j-blocks, i.e. gathers T2 of the smallest j first: going
    // through the nodes from n-1 down to 0 the pass starts on what the sweep just wrote and ends on what the next sweep reads first.  C4
    // (profiles/r03_node_order.txt, alternating in one call): column sums 238 -> 232 us, the SWEEP 1193 -> 1129 us (-5.4 %); a hashed
    // shuffle: sweep -4 % but column sums +4 %; longest row first (the scheduling argument): erratic; C5, C2, C3: within noise.
    // DESC_DEBUG_NODE_ORDER: 0 ascending (round 2), 1 longest row first, 2 reverse (default), 3 hashed shuffle.
    // Sharded runs (round 4): only the nodes that have segments of this rank in their row are visited, and of their row only the run of CSR
    // slots those segments occupy (k_node_runs).  Measured per rank of 8 at C4 before that (profiles/r04_shard_w8_c4_before.json): 132 us
    // for an eighth of the cycles against 233 us for all of them on one GPU -- every rank started all n workgroups and walked whole rows.
    {
        hvec<int32_t> ord;
        hvec<int2> runs;
        ord.reserve((size_t)n);
        if (h->world > 1 && n > 0) {
            if ((rc = dalloc(h, &h->d_node_run, (size_t)n))) return rc;
            hipLaunchKernelGGL(k_node_runs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_rowptr, d_adj, P.rank_node[h->rank], P.rank_node[h->rank + 1],
                               h->d_node_run, (int)n);
            runs.resize((size_t)n);
            DESC_HIP(hipStreamSynchronize(h->stream));
            DESC_HIP(hipMemcpy(runs.data(), h->d_node_run, sizeof(int2) * (size_t)n, hipMemcpyDeviceToHost));
            for (int64_t v = 0; v < n; ++v) if (runs[v].y > runs[v].x) ord.push_back((int32_t)v);
        } else
            for (int64_t v = 0; v < n; ++v) ord.push_back((int32_t)v);
        const int mode = env_int("DESC_DEBUG_NODE_ORDER", 2);
        if (mode == 2) std::reverse(ord.begin(), ord.end());
        else if (mode == 3) std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return mix64((uint64_t)x) < mix64((uint64_t)y); });
        else if (mode == 1) std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return P.rowptr[x + 1] - P.rowptr[x] > P.rowptr[y + 1] - P.rowptr[y]; });
        // sharded: rows with a short run of this rank's segments go to the back of the list -- k_colsum_node gives them a wave each
        h->colsum_long = (int)ord.size();
        if (!runs.empty() && 4 * COLSUM_SHORT <= ((h->max_deg + 1) | 1) && env_int("DESC_DEBUG_COLSUM_SHORT", 1)) {
            const auto mid = std::stable_partition(ord.begin(), ord.end(), [&](int32_t v) { return runs[v].y - runs[v].x > COLSUM_SHORT; });
            h->colsum_long = (int)(mid - ord.begin());
        }
        h->colsum_nodes = (int)ord.size();
        h->colsum_grid = h->colsum_long + (h->colsum_nodes - h->colsum_long + 3) / 4;     // workgroups (0: a rank without segments); the launch adds the bookkeeping workgroup
        if ((rc = dalloc(h, &h->d_node_order, ord.size())) || (rc = upload(h, h->d_node_order, ord.data(), ord.size()))) return rc;
        DESC_HIP(hipStreamSynchronize(h->stream));
    }
    {
        int bits = 0;
        while ((1 << bits) < h->max_deg + 1) ++bits;
        h->colsum_fx_bits = 62 - bits;
        const size_t lds = colsum_lds(h);
        if (lds > 64 * 1024)
        {
            DESC_HIP(hipFuncSetAttribute((const void*)k_colsum_node<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            DESC_HIP(hipFuncSetAttribute((const void*)k_colsum_node<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
    }
    if (h->band_ok) {      // the band rows + the nv table in dynamic LDS: more than the 64 KiB default
        h->band_lds = ((size_t)h->band_rows + MAX_SEG_CYCLES + 1) * sizeof(double);
        for (const void* kb : {band_kernel<DESC_STEP_CONSTANT, false>(h), band_kernel<DESC_STEP_HYBRID, false>(h), band_kernel<DESC_STEP_CONSTANT, true>(h), band_kernel<DESC_STEP_HYBRID, true>(h)})
            if (kb && hipFuncSetAttribute(kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->band_lds) != hipSuccess) { (void)hipGetLastError(); h->band_ok = false; }
    }
    // the reference's demo sizes: the latency-lean sweep (DESC_DEBUG_VARIANT = 2 / 3 keep k_sweep_node / the band sweep for the cross-checks)
    h->small_ok = force_band == 0 && !h->band_ok && h->world == 1 && h->max_cnt <= 64 && h->m_cycle < (2 << 20) && env_int("DESC_DEBUG_SMALL", 1) != 0;
    h->small_g = h->max_cnt <= 16 ? 16 : h->max_cnt <= 32 ? 32 : 64;
    h->small_grid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (nsl + 4 * (64 / h->small_g) - 1) / (4 * (64 / h->small_g))));
    char nm[64];
    if (h->small_ok) snprintf(nm, sizeof nm, "k_sweep_small<%d,", h->small_g);
    else if (h->band_ok) { const BandShape sh = band_shape(h, false); snprintf(nm, sizeof nm, "k_sweep_band<%d,%d,", sh.lps, sh.E); }
    else snprintf(nm, sizeof nm, "k_sweep_node<%d,%d,", h->lps, h->G);
    h->kname = nm;

    lap("occupancy/attrs");
    hipEvent_t e0, e1;
    DESC_HIP(hipEventCreate(&e0)); DESC_HIP(hipEventCreate(&e1));
    (void)hipEventRecord(e0, h->stream);
    if (nsl > 0 && !dev_cycles) {
        int g = (int)std::min<int64_t>(4096, (nsl + 3) / 4);
        hipLaunchKernelGGL(k_layout_node, dim3(g), dim3(256), 0, h->stream, h->d_cum + h->seg_lo, d_pos_edge2 + h->seg_lo, d_ii, d_jj,
                           d_kf, h->d_rowptr, d_adj, d_adj_eid, d_rij, h->d_pk, h->d_S0, (int)nsl);
    } else if (dev_cycles) {
        // the structure's per-cycle arrays never left the device: lay them out in place
        uint32_t* d_counts = nullptr;
        if ((rc = dalloc(h, &d_counts, mp))) return rc;
        DESC_HIP(hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * std::max<int64_t>(1, mp), h->stream));
        if (early) {            // computed in natural order under the planning: move the segments to their device positions
            hipLaunchKernelGGL(k_permute_segments, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, (mp + 3) / 4))), dim3(256), 0, h->stream,
                               d_order_keep, s->d_cum, h->d_cum, d_pk_nat, d_S0_nat, d_perm_nat, d_counts_nat, h->d_pk, h->d_S0, h->d_seg_perm, d_counts, mp);
        } else if (nsl > 0)
            if ((rc = launch_layout_dev(h->d_cum + h->seg_lo, h->d_src_start + h->seg_lo, d_pos_edge2 + h->seg_lo, h->d_rowptr, h->d_pk, h->d_S0, h->d_seg_perm, d_counts + h->seg_lo, nsl, nullptr))) return rc;
        hipLaunchKernelGGL(k_adj_seg, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (n * 16 + 255) / 256))), dim3(256), 0, h->stream,
                           h->d_rowptr, d_adj, d_adj_eid, d_devpos, h->d_cum, d_counts, (int)h->seg_lo, (int)h->seg_hi, h->d_adj_seg, (int)n);
        DESC_HIP(hipStreamSynchronize(h->stream));
        dfree(h, d_devpos); dfree(h, d_counts);
        dfree(h, d_order_keep); dfree(h, d_pk_nat); dfree(h, d_S0_nat); dfree(h, d_perm_nat); dfree(h, d_counts_nat);
    }
    {   // exchange layout of the sharded runs (both paths have the CSR index, the edge list and the segment tables on the device by now)
        int32_t *d_node_lo = nullptr, *d_elo = nullptr, *d_prefB = nullptr;
        if ((rc = dalloc(h, &d_node_lo, (size_t)nv + 1)) || (rc = dalloc(h, &d_elo, (size_t)nv + 1)) || (rc = dalloc(h, &d_prefB, (size_t)nv * std::max<int64_t>(n, 1)))) return rc;
        if ((rc = upload(h, d_node_lo, P.vnode.data(), (size_t)nv + 1)) || (rc = upload(h, d_elo, e_lo.data(), (size_t)nv + 1))) return rc;
        if (n > 0) {
            const unsigned g16 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (n * 16 + 255) / 256));
            hipLaunchKernelGGL(k_prefB, dim3(nv), dim3(1024), 0, h->stream, h->d_rowptr, d_adj, d_node_lo, d_prefB, (int)n);
            hipLaunchKernelGGL(k_xpos, dim3(g16), dim3(256), 0, h->stream, h->d_rowptr, d_adj, d_adj_eid, d_node_lo, d_elo, d_prefB, nv, h->xparts, h->world, h->t_part, h->slice_len,
                               h->d_xpos, h->d_spos, (int)n);
            if (nsl > 0)
                hipLaunchKernelGGL(k_xt, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (nsl + 255) / 256))), dim3(256), 0, h->stream, d_pos_edge2, h->d_einfo,
                                   d_ii, d_jj, h->d_rowptr, d_adj, d_node_lo, d_elo, d_prefB, nv, h->t_part / 2, (int)h->seg_lo, (int)h->seg_hi, h->d_xt, (int)n);
        }
        DESC_HIP(hipStreamSynchronize(h->stream));            // e_lo / rank_node: host sources of the copies
        DESC_HIP(hipGetLastError());
        dfree(h, d_node_lo); dfree(h, d_elo); dfree(h, d_prefB);
    }
    {   // the column-index stream of the column-sum pass (needs adj_seg and pk: both complete on the stream by now)
        uint32_t *d_rowsum = nullptr, *d_rowbase = nullptr;
        if ((rc = dalloc(h, &d_rowsum, n)) || (rc = dalloc(h, &d_rowbase, n)) || (rc = dalloc_stream(h, &h->d_moff, 2 * m, 16))) return rc;
        const unsigned g16 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (n * 16 + 255) / 256));
        hipLaunchKernelGGL(k_midx_rowsum, dim3(g16), dim3(256), 0, h->stream, h->d_rowptr, h->d_adj_seg, d_rowsum, (int)n);
        hvec<uint32_t> rs((size_t)n), rb((size_t)n);
        DESC_HIP(hipStreamSynchronize(h->stream));
        DESC_HIP(hipMemcpy(rs.data(), d_rowsum, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
        uint64_t tot = 0;
        for (int64_t v = 0; v < n; ++v) { rb[v] = (uint32_t)tot; tot += rs[v]; }
        if (tot >= (1ull << 32)) return fail(DESC_ERR_TOO_LARGE, "column-index stream exceeds 2^32 entries");
        h->colsum_entries = (int64_t)tot;
        if ((rc = dalloc_stream(h, &h->d_midx, (size_t)tot + 8, 8))) return rc;
        {   // average run length (contributing cycles per segment and endpoint) decides the lanes per segment of the column sums
            const double avg_run = nsl > 0 ? (double)tot / (2.0 * (double)nsl) : 0.0;
            const int forced_cl = env_int("DESC_DEBUG_COLSUM_CL", 0);
            h->colsum_cl = forced_cl == 8 || forced_cl == 16 ? forced_cl : (avg_run > 0.0 && avg_run <= 10.5 ? 8 : 16);
            if (timing) fprintf(stderr, "[desc_amd] column sums: %.2f contributing cycles per segment and endpoint on average -> %d lanes per segment\n", avg_run, h->colsum_cl);
        }
        if ((rc = upload(h, d_rowbase, rb.data(), (size_t)n))) return rc;
        hipLaunchKernelGGL(k_midx_fill, dim3(g16), dim3(256), 0, h->stream, h->d_rowptr, h->d_adj_seg, d_rowbase, h->d_moff, (int)n);
        hipLaunchKernelGGL(k_midx_entries, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, (2 * m * 16 + 255) / 256))), dim3(256), 0, h->stream,
                           h->d_adj_seg, h->d_moff, h->d_pk, h->d_midx, (int64_t)2 * m);
        DESC_HIP(hipStreamSynchronize(h->stream));
        dfree(h, d_rowsum); dfree(h, d_rowbase);
    }
    (void)hipEventRecord(e1, h->stream);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    h->ms_cycle_d = ms;
    lap("layout kernels");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (!dev_cycles) { dfree(h, d_ii); dfree(h, d_jj); dfree(h, d_adj); dfree(h, d_adj_eid); dfree(h, d_kf); }
    dfree(h, d_pos_edge2); if (!shared_rij) dfree(h, d_rij);
    if (e != hipSuccess) return fail(DESC_ERR_HIP, "k_layout_node: %s", hipGetErrorString(e));
    return DESC_OK;
}

// per-cycle host vector <-> device (handles the node layout's segment permutation)
int cycles_to_device(desc_pgd* h, const double* host, double* dev) {
    if (h->m_cycle == 0) return DESC_OK;
    if (h->variant != VARIANT_NODE) { DESC_HIP(hipMemcpyAsync(dev, host, sizeof(double) * h->m_cycle, hipMemcpyHostToDevice, h->stream)); return DESC_OK; }
    DESC_HIP(hipMemcpyAsync(h->d_scratch, host, sizeof(double) * h->m_cycle, hipMemcpyHostToDevice, h->stream));
    int g = (int)std::min<int64_t>(4096, (h->m_pos + 3) / 4);
    hipLaunchKernelGGL(k_reorder_cycles, dim3(g), dim3(256), 0, h->stream, h->d_cum + h->seg_lo, h->d_src_start + h->seg_lo, h->d_seg_perm, h->d_scratch, dev, (int)(h->seg_hi - h->seg_lo), 0);
    return DESC_OK;
}
int cycles_to_host(desc_pgd* h, const double* dev, double* host) {
    if (h->m_cycle == 0) return DESC_OK;
    if (h->variant != VARIANT_NODE) { DESC_HIP(hipMemcpy(host, dev, sizeof(double) * h->m_cycle, hipMemcpyDeviceToHost)); return DESC_OK; }
    int g = (int)std::min<int64_t>(4096, (h->m_pos + 3) / 4);
    hipLaunchKernelGGL(k_reorder_cycles, dim3(g), dim3(256), 0, h->stream, h->d_cum + h->seg_lo, h->d_src_start + h->seg_lo, h->d_seg_perm, dev, h->d_scratch, (int)(h->seg_hi - h->seg_lo), 1);
    DESC_HIP(hipStreamSynchronize(h->stream));
    DESC_HIP(hipMemcpy(host, h->d_scratch, sizeof(double) * h->m_cycle, hipMemcpyDeviceToHost));      // synchronous: nothing stays pending into caller memory
    return DESC_OK;
}
int ensure_scratch(desc_pgd* h) {
    if (h->d_scratch || h->variant != VARIANT_NODE) return DESC_OK;
    return dalloc(h, &h->d_scratch, h->m_cycle);
}

}  // namespace

extern "C" {

int desc_device_count(void) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) return fail(DESC_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return c;
}

int desc_pgd_create(const desc_problem* prob, const desc_structure* s, int32_t device, desc_pgd** out) {
    return desc_pgd_create_shard(prob, s, device, 0, 1, out);
}

static int create_impl(const desc_problem* prob, const double* shared_rij, const desc_structure* s, int32_t device, int32_t rank, int32_t world, desc_pgd** out);

int desc_pgd_create_shard(const desc_problem* prob, const desc_structure* s, int32_t device, int32_t rank, int32_t world,
                          desc_pgd** out) {
    return no_throw("desc_pgd_create_shard", [&]() -> int {
    return create_impl(prob, nullptr, s, device, rank, world, out);
    });
}
// the same with the problem already resident in HBM (desc_problem_upload): nothing but O(m_pos) plan tables is uploaded
int desc_pgd_create_dev(const desc_device_problem* dp, const desc_structure* s, int32_t rank, int32_t world, desc_pgd** out) {
    return no_throw("desc_pgd_create_dev", [&]() -> int {
    if (!dp) return fail(DESC_ERR_INVALID, "NULL argument");
    const desc_problem hv = host_view(dp);
    return create_impl(&hv, dp->d_rij, s, dp->device, rank, world, out);
    });
}

static int create_impl(const desc_problem* prob, const double* shared_rij, const desc_structure* s, int32_t device, int32_t rank, int32_t world,
                       desc_pgd** out) {
    if (!out) return fail(DESC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!prob || !s) return fail(DESC_ERR_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(DESC_ERR_INVALID, "rank %d / world %d", rank, world);
    if (prob->m != s->m) return fail(DESC_ERR_INVALID, "structure was built for m = %lld, problem has m = %lld", (long long)s->m, (long long)prob->m);
    if (prob->m > 0 && !prob->rij && !shared_rij) return fail(DESC_ERR_INVALID, "rij is NULL");
    if (prob->n < s->n) return fail(DESC_ERR_INVALID, "problem n smaller than structure n");
    int ndev = desc_device_count();
    if (ndev < 0) return ndev;
    if (ndev == 0) return fail(DESC_ERR_HIP, "no HIP device visible: the DESC_PGD hot path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(DESC_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);

    desc_pgd* h = new (std::nothrow) desc_pgd();
    if (!h) return fail(DESC_ERR_INVALID, "out of host memory");
    h->device = device; h->rank = rank; h->world = world;
    h->n = prob->n; h->m = s->m; h->m_pos = s->m_pos; h->m_cycle = s->m_cycle; h->max_cnt = s->max_cnt; h->n_sample = s->n_sample;
    if (s->max_deg > 0) h->max_deg = s->max_deg;          // device-built structures carry it
    else {
        hvec<int32_t> deg((size_t)h->n, 0);
        for (int64_t e = 0; e < h->m; ++e) { deg[prob->ind_i[e]]++; deg[prob->ind_j[e]]++; }
        for (int32_t d : deg) h->max_deg = std::max(h->max_deg, d);
    }
    int rc = DESC_OK;
    hipError_t he = hipSetDevice(device);
    if (he == hipSuccess) he = stream_acquire(&h->stream);
    if (he != hipSuccess) { rc = fail(DESC_ERR_HIP, "device %d: %s", device, hipGetErrorString(he)); free_all(h); return rc; }

    // variant: NODE unless the packed per-cycle word or the LDS column copies do not fit
    const int forced = env_int("DESC_DEBUG_VARIANT", 0);
    const bool node_ok = h->max_deg < 32768 && (size_t)(h->max_deg + 2) * 44 <= 150 * 1024 && h->max_cnt <= MAX_SEG_CYCLES && h->m_pos > 0;
    h->variant = (forced == VARIANT_GATHER || !node_ok) ? VARIANT_GATHER : VARIANT_NODE;
    if (world > 1 && h->variant != VARIANT_NODE) {
        rc = fail(DESC_ERR_INVALID, "multi-GPU sharding needs the node layout (max degree < 2^15, segments <= 64 cycles)");
        free_all(h); return rc;
    }

    const int64_t mp = h->m_pos, mc = h->m_cycle;
    auto A = [&](int r) { if (!rc) rc = r; };
    A(dalloc(h, &h->d_cum, mp + 1));
    if (h->variant != VARIANT_NODE) { A(dalloc(h, &h->d_S0, mc)); A(dalloc(h, &h->d_w[0], mc)); A(dalloc(h, &h->d_w[1], mc)); }
    A(dalloc(h, &h->d_nv, (size_t)h->max_cnt + 1));
    A(dalloc(h, &h->d_state, 1));
    if (!rc) {
        hvec<double> nv((size_t)h->max_cnt + 1, 0.0);
        for (int c = 1; c <= h->max_cnt; ++c) nv[c] = 1.0 / std::pow((double)c, 0.5);   // ones/(nsample^0.5), DESC_PGD.m:199
        rc = upload(h, h->d_nv, nv.data(), nv.size());
        if (!rc) { hipError_t e = hipStreamSynchronize(h->stream); if (e != hipSuccess) rc = fail(DESC_ERR_HIP, "upload: %s", hipGetErrorString(e)); }
    }
    if (!rc) rc = (h->variant == VARIANT_NODE) ? setup_node(h, prob, s, shared_rij) : setup_gather(h, prob, s, shared_rij);
    if (!rc) rc = dalloc(h, &h->d_partials, 4 * parts_cap(h));   // sweep partials, then the objective kernel's
    if (rc) { free_all(h); return rc; }
    *out = h;
    return DESC_OK;
}

void desc_pgd_destroy(desc_pgd* h) { free_all(h); }

}  // extern "C"
namespace desc {
int pgd_create_with_rij(const desc_problem* prob, const double* d_rij, const desc_structure* s, int32_t device, desc_pgd** out) {
    return create_impl(prob, d_rij, s, device, 0, 1, out);
}
}  // namespace desc
extern "C" {

int desc_pgd_sizes(const desc_pgd* h, int64_t* m, int64_t* m_pos, int64_t* m_cycle, int32_t* max_cnt) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    if (m) *m = h->m;
    if (m_pos) *m_pos = h->m_pos;
    if (m_cycle) *m_cycle = h->m_cycle;
    if (max_cnt) *max_cnt = h->max_cnt;
    return DESC_OK;
}

const char* desc_pgd_kernel_name(const desc_pgd* h) { return h ? h->kname.c_str() : ""; }

int desc_pgd_get_s0(desc_pgd* h, double* s0) {
    if (!h || !s0) return fail(DESC_ERR_INVALID, "NULL argument");
    int rc = set_device(h); if (rc) return rc;
    if ((rc = ensure_scratch(h))) return rc;
    return cycles_to_host(h, h->d_S0, s0);
}

int desc_pgd_reset(desc_pgd* h, const desc_params* p) {
    if (!h || !p) return fail(DESC_ERR_INVALID, "NULL argument");
    if (p->iters < 0) return fail(DESC_ERR_INVALID, "iters < 0");
    if (p->step_kind < 0 || p->step_kind > 2) return fail(DESC_ERR_INVALID, "unknown step_kind %d", p->step_kind);
    if ((p->step_kind == DESC_STEP_PIECEWISE || (p->step_kind == DESC_STEP_HYBRID && p->hybrid_strategy == 1)) && !(p->decay_interval > 0))
        return fail(DESC_ERR_INVALID, "decay_interval must be > 0");
    int rc = set_device(h); if (rc) return rc;
    h->p = *p;
    if (h->p.patience <= 0) h->p.patience = 30;
    {   // diagnostics only: never set in production runs
        const int gr = env_int("DESC_DEBUG_GRID", 0);
        if (gr >= 8 && gr / 8 * 8 != h->grid) {
            dfree(h, h->d_partials); h->d_partials = nullptr;
            h->grid = gr / 8 * 8;
            rc = dalloc(h, &h->d_partials, 4 * parts_cap(h)); if (rc) return rc;
        }
    }
    h->t_done = 0; h->t_plugin = p->t0; h->ms_pgd = 0; h->objective_done = false; h->final_obj_T = -1; h->pending_fin = 0;

    const int cap = std::max(1, p->iters);
    h->iters_cap = cap;
    if (cap > h->trace_cap) {
        dfree(h, h->d_obj); dfree(h, h->d_avg);
        h->d_obj = h->d_avg = nullptr; h->trace_cap = 0;
        rc = dalloc(h, &h->d_obj, cap); if (rc) return rc;
        rc = dalloc(h, &h->d_avg, cap); if (rc) return rc;
        h->trace_cap = cap;
    }
    if (p->step_kind == DESC_STEP_HYBRID && p->hybrid_strategy == 0 && !h->d_adam_m[0])
        for (int q = 0; q < 2; ++q) {
            rc = dalloc(h, &h->d_adam_m[q], local_cycles(h)); if (rc) return rc;
            rc = dalloc(h, &h->d_adam_v[q], local_cycles(h)); if (rc) return rc;
        }
    DESC_HIP(hipMemsetAsync(h->d_state, 0, sizeof(DevState), h->stream));
    DESC_HIP(hipMemsetAsync(h->d_obj, 0, sizeof(double) * h->trace_cap, h->stream));
    DESC_HIP(hipMemsetAsync(h->d_avg, 0, sizeof(double) * h->trace_cap, h->stream));
    if (h->d_adam_m[0])
        for (int q = 0; q < 2; ++q) {
            DESC_HIP(hipMemsetAsync(h->d_adam_m[q], 0, sizeof(double) * std::max<int64_t>(1, local_cycles(h)), h->stream));
            DESC_HIP(hipMemsetAsync(h->d_adam_v[q], 0, sizeof(double) * std::max<int64_t>(1, local_cycles(h)), h->stream));
        }
    const int64_t slen = h->variant == VARIANT_NODE ? 2 * h->m : h->m;
    if (slen > 0) {                                                       // S_vec = ones(1,m)  (:148)
        int g = (int)std::min<int64_t>(1024, (slen + 255) / 256);
        hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, h->stream, h->d_S[0], slen, 1.0);
        hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, h->stream, h->d_S[1], slen, 1.0);
    }
    if (h->x_sall && h->slice_S > 0) {          // sharded: S of the edges this rank owns, edge order; edges without cycles keep 1 (:148)
        int g = (int)std::min<int64_t>(1024, (h->slice_S + 255) / 256);
        hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, h->stream, h->x_sall + (int64_t)h->rank * h->slice_len, h->slice_S, 1.0);
    }
    if (h->m_pos > 0) {
        int g = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (h->m_pos + 3) / 4));
        if (h->variant == VARIANT_NODE) {
            for (int c = 0; c < h->xparts; ++c) {        // per exchange part: its segments' initial S goes behind the parts before it in the slice
                const int64_t q0 = h->xseg[c], q1 = h->xseg[c + 1];
                if (q1 > q0)
                    hipLaunchKernelGGL(k_init_node, dim3(g), dim3(256), 0, h->stream, h->d_cum + q0, h->d_einfo + q0, h->d_S0, h->d_w[0], h->d_S[0], h->d_S[1], (int)(q1 - q0),
                                       h->x_sall ? h->x_sall + (int64_t)h->rank * h->slice_len + h->xslice_off[c] : (double*)nullptr, h->d_xt + q0);
            }
        } else
            hipLaunchKernelGGL(k_init, dim3(g), dim3(256), 0, h->stream, h->d_cum, h->d_pos_edge, h->d_S0, h->d_w[0], h->d_S[0], h->d_S[1], (int)h->m_pos);
    }
    DESC_HIP(hipGetLastError());
    h->armed = true;
    return DESC_OK;
}

int desc_pgd_iterate(desc_pgd* h, int32_t n_iters) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    if (!h->armed) return fail(DESC_ERR_STATE, "desc_pgd_reset must be called first");
    if (n_iters < 0 || h->t_done + n_iters > h->iters_cap) return fail(DESC_ERR_INVALID, "iterating past params.iters = %d", h->p.iters);
    int rc = set_device(h); if (rc) return rc;
    if (h->m_pos == 0) { h->t_done += n_iters; h->t_plugin += n_iters; return DESC_OK; }
    return enqueue_iterations(h, n_iters);
}

int desc_pgd_iterate_timed(desc_pgd* h, int32_t n_iters, float* ms_total, float* ms_main_kernel_avg) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    if (!h->armed) return fail(DESC_ERR_STATE, "desc_pgd_reset must be called first");
    if (n_iters < 0 || h->t_done + n_iters > h->iters_cap) return fail(DESC_ERR_INVALID, "iterating past params.iters = %d", h->p.iters);
    int rc = set_device(h); if (rc) return rc;
    hipEvent_t e0, e1;
    DESC_HIP(hipEventCreate(&e0)); DESC_HIP(hipEventCreate(&e1));
    hvec<hipEvent_t> ev;
    if (ms_main_kernel_avg) { ev.resize(2 * (size_t)n_iters); for (auto& e : ev) DESC_HIP(hipEventCreate(&e)); }
    DESC_HIP(hipEventRecord(e0, h->stream));
    if (h->m_pos > 0 && !ms_main_kernel_avg) { rc = enqueue_iterations(h, n_iters); if (rc) return rc; }
    else if (h->m_pos > 0)                                   // per-iteration events: direct launches
        for (int q = 0; q < n_iters; ++q) {
            rc = enqueue_sweep(h, ++h->t_done, ev[2 * q], ev[2 * q + 1]);
            if (rc) return rc;
        }
    else { h->t_done += n_iters; h->t_plugin += n_iters; }
    DESC_HIP(hipEventRecord(e1, h->stream));
    DESC_HIP(hipStreamSynchronize(h->stream));
    float ms = 0; DESC_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms_total) *ms_total = ms;
    if (ms_main_kernel_avg) {
        double acc = 0; int cntk = 0;
        if (h->m_pos > 0)
            for (int q = 0; q < n_iters; ++q) { float x = 0; DESC_HIP(hipEventElapsedTime(&x, ev[2 * q], ev[2 * q + 1])); acc += x; ++cntk; }
        *ms_main_kernel_avg = cntk ? (float)(acc / cntk) : 0.f;
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    h->ms_pgd += ms;
    return DESC_OK;
}

namespace { int shard_unpack_pending(desc_pgd* h); }

int desc_pgd_sync(desc_pgd* h) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    int rc = set_device(h); if (rc) return rc;
    if ((rc = shard_unpack_pending(h))) return rc;           // fused protocol: a sync leaves every enqueued iteration complete, its unpacking included
    DESC_HIP(hipStreamSynchronize(h->stream));
    if (h->comm_stream) DESC_HIP(hipStreamSynchronize(h->comm_stream));
    DESC_HIP(hipGetLastError());
    return DESC_OK;
}

namespace { int shard_unpack_pending(desc_pgd* h); }

int desc_pgd_download(desc_pgd* h, desc_result* r) {
    if (!h || !r) return fail(DESC_ERR_INVALID, "NULL argument");
    if (!h->armed) return fail(DESC_ERR_STATE, "nothing to download: call desc_pgd_reset first");
    if (!r->s_vec && h->m > 0) return fail(DESC_ERR_INVALID, "result.s_vec is NULL");
    int rc = set_device(h); if (rc) return rc;
    const int T = h->t_done;
    flush_finalize(h);
    if ((rc = shard_unpack_pending(h))) return rc;           // fused protocol: S and the stop rule of the last sweep
    if (h->comm_stream) DESC_HIP(hipStreamSynchronize(h->comm_stream));
    // objective of the last sweep (DESC_PGD.m:233) and its stop test
    if (h->m_pos > 0 && T >= 1 && h->world == 1 && h->final_obj_T != T) {
        h->final_obj_T = T;
        if (h->variant == VARIANT_NODE)
            hipLaunchKernelGGL(k_objective_node, dim3(h->obj_grid), dim3(256), 0, h->stream, h->d_cum + h->seg_lo, h->d_einfo + h->seg_lo, h->d_pk,
                               h->d_w[T & 1], h->d_S[T & 1], (int)(h->seg_hi - h->seg_lo), obj_partials(h), h->d_state);
        else
            hipLaunchKernelGGL(k_objective, dim3(h->obj_grid), dim3(256), 0, h->stream, h->d_w[T & 1], h->d_S[T & 1], h->d_ejk,
                               h->d_eki, h->m_cycle, obj_partials(h), h->d_state);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, h->stream, fin_args(h, obj_partials(h), h->obj_grid, T, 1));
    }
    DevState st{};
    DESC_HIP(hipStreamSynchronize(h->stream));
    DESC_HIP(hipMemcpy(&st, h->d_state, sizeof st, hipMemcpyDeviceToHost));      // synchronous: `st` is a stack slot
    DESC_HIP(hipGetLastError());
    int iters_run = T, par = T & 1;
    if (h->m_pos > 0 && st.stop) { iters_run = st.iters_run; par = st.final_parity; }
    if (h->m_pos == 0) {
        // no cycles at all: every sweep is a no-op, objective 0, so the reference
        // breaks after `patience` misses counted from iteration 2 (DESC_PGD.m:243-246)
        if (T >= h->p.patience + 1) iters_run = h->p.patience + 1;
        par = 0;
    }
    r->iters_run = iters_run;
    r->t_end = h->p.t0 + iters_run;
    if (h->m > 0) {
        if (h->variant == VARIANT_NODE) {
            int g = (int)std::min<int64_t>(1024, (h->m + 255) / 256);
            hipLaunchKernelGGL(k_extract_S, dim3(g), dim3(256), 0, h->stream, h->d_S[par], h->d_eslot, h->d_Svec, h->m);
            DESC_HIP(hipStreamSynchronize(h->stream));
            DESC_HIP(hipMemcpy(r->s_vec, h->d_Svec, sizeof(double) * h->m, hipMemcpyDeviceToHost));
        } else {
            DESC_HIP(hipMemcpy(r->s_vec, h->d_S[par], sizeof(double) * h->m, hipMemcpyDeviceToHost));
        }
    }
    if (h->world > 1 && (r->w || r->adam_m || r->adam_v))
        return fail(DESC_ERR_INVALID, "per-cycle outputs (w, Adam state) are not gathered across ranks");
    if (r->w || (h->d_adam_m[0] && r->adam_m && r->adam_v)) { rc = ensure_scratch(h); if (rc) return rc; }
    if (r->w) { rc = cycles_to_host(h, h->d_w[par], r->w); if (rc) return rc; }
    if (r->obj_trace && iters_run > 0) {
        if (h->m_pos > 0) DESC_HIP(hipMemcpy(r->obj_trace, h->d_obj, sizeof(double) * iters_run, hipMemcpyDeviceToHost));
        else std::memset(r->obj_trace, 0, sizeof(double) * iters_run);
    }
    if (r->avg_change_trace && iters_run > 0) {
        if (h->m_pos > 0) DESC_HIP(hipMemcpy(r->avg_change_trace, h->d_avg, sizeof(double) * iters_run, hipMemcpyDeviceToHost));
        else std::memset(r->avg_change_trace, 0, sizeof(double) * iters_run);
    }
    if (h->d_adam_m[0] && r->adam_m && r->adam_v) {          // the state after exactly iters_run GetStep calls
        rc = cycles_to_host(h, h->d_adam_m[par], r->adam_m); if (rc) return rc;
        rc = cycles_to_host(h, h->d_adam_v[par], r->adam_v); if (rc) return rc;
    }
    r->ms_upload = h->ms_upload; r->ms_cycle_d = h->ms_cycle_d; r->ms_pgd = h->ms_pgd;
    return DESC_OK;
}

// HybridGradient keeps m_t / v_t between calls (handle object): after desc_pgd_reset, the caller's moments of a run that
// continues (t0 > 0) go to the parity sweep 1 reads
static int upload_adam_state(desc_pgd* h, const desc_params* p, const desc_result* r) {
    if (!(p->step_kind == DESC_STEP_HYBRID && p->hybrid_strategy == 0 && p->t0 > 0 && r->adam_m && r->adam_v && h->m_cycle > 0)) return DESC_OK;
    int rc = ensure_scratch(h); if (rc) return rc;
    rc = cycles_to_device(h, r->adam_m, h->d_adam_m[0]); if (rc) return rc;
    DESC_HIP(hipStreamSynchronize(h->stream));
    rc = cycles_to_device(h, r->adam_v, h->d_adam_v[0]); if (rc) return rc;
    DESC_HIP(hipStreamSynchronize(h->stream));
    return DESC_OK;
}

// params.make_plots = true (DESC_PGD.m:235-239) as a composition of device rows: one sweep, one S_vec download and one GCW
// eigen-solve per iteration.  The alignment against R_orig (:238, GlobalSOdCorrectRight) stays with the caller.
int desc_pgd_run_traced(desc_pgd* h, const desc_device_problem* dp, const desc_params* p, const double* err_vec, double gcw_tol,
                        int32_t gcw_max_iters, double* svec_errors, double* R_est_all, desc_result* r) {
    return no_throw("desc_pgd_run_traced", [&]() -> int {
    if (!h || !dp || !p || !r || !err_vec || !svec_errors || !R_est_all) return fail(DESC_ERR_INVALID, "NULL argument");
    if (!r->s_vec || !r->obj_trace || !r->avg_change_trace) return fail(DESC_ERR_INVALID, "the traced run needs s_vec, obj_trace and avg_change_trace");
    if (dp->m != h->m || dp->n != h->n) return fail(DESC_ERR_INVALID, "device problem and solver handle describe different graphs");
    auto t0 = std::chrono::steady_clock::now();
    int rc = desc_pgd_reset(h, p); if (rc) return rc;
    if ((rc = upload_adam_state(h, p, r))) return rc;        // the moments stay on the device between the one-iteration pieces
    desc_result mid = *r;                                    // same buffers; w / Adam state only in the final download
    mid.w = nullptr; mid.adam_m = nullptr; mid.adam_v = nullptr;
    int done = 0;
    while (done < p->iters) {
        if ((rc = desc_pgd_iterate(h, 1))) return rc;
        ++done;
        if ((rc = desc_pgd_download(h, &mid))) return rc;    // also evaluates this iteration's objective and stop test
        if (mid.iters_run < done) break;                     // the patience rule fired at an earlier iteration (:243-246)
        double acc = 0.0;
        for (int64_t e = 0; e < h->m; ++e) acc += std::fabs(err_vec[e] - mid.s_vec[e]);
        svec_errors[done - 1] = h->m > 0 ? acc / (double)h->m : 0.0;                                     // :236
        if ((rc = desc_gcw_run_dev(dp, mid.s_vec, gcw_tol, gcw_max_iters, R_est_all + (size_t)(done - 1) * 9 * (size_t)h->n, nullptr))) return rc;   // :237
        if (p->progress) p->progress(p->progress_user, done, mid.avg_change_trace[done - 1], mid.obj_trace[done - 1]);
        else if (p->verbose) { printf("iter %d: average change in S_vec %f, objective value: %f\n", done, mid.avg_change_trace[done - 1], mid.obj_trace[done - 1]); fflush(stdout); }
    }
    rc = desc_pgd_download(h, r);
    r->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
    });
}

int desc_pgd_run(desc_pgd* h, const desc_params* p, desc_result* r) {
    return no_throw("desc_pgd_run", [&]() -> int {
    if (!h || !p || !r) return fail(DESC_ERR_INVALID, "NULL argument");
    auto t0 = std::chrono::steady_clock::now();
    int rc = desc_pgd_reset(h, p); if (rc) return rc;
    if ((rc = upload_adam_state(h, p, r))) return rc;
    // progress lines (DESC_PGD.m:241) are streamed while the loop runs: at every poll of the stop flag the traces of the
    // iterations that became final since the last poll are fetched and handed to the callback (or printed)
    const bool stream_lines = (p->progress != nullptr || p->verbose) && h->m_pos > 0;
    int chunk = p->check_every > 0 ? p->check_every : 32;
    if (stream_lines && p->check_every <= 0) chunk = 10;
    int reported = 0;
    hvec<double> lo, la;
    auto emit = [&](int it, double avg, double obj) {
        if (p->progress) p->progress(p->progress_user, it, avg, obj);
        else { printf("iter %d: average change in S_vec %f, objective value: %f\n", it, avg, obj); fflush(stdout); }
    };
    auto report_upto = [&](int upto) -> int {              // iterations reported+1 .. upto have both trace entries final
        if (upto <= reported) return DESC_OK;
        lo.resize((size_t)upto - reported); la.resize((size_t)upto - reported);
        DESC_HIP(hipMemcpy(lo.data(), h->d_obj + reported, sizeof(double) * (upto - reported), hipMemcpyDeviceToHost));
        DESC_HIP(hipMemcpy(la.data(), h->d_avg + reported, sizeof(double) * (upto - reported), hipMemcpyDeviceToHost));
        for (int it = reported + 1; it <= upto; ++it) emit(it, la[it - 1 - reported], lo[it - 1 - reported]);
        reported = upto;
        return DESC_OK;
    };
    int left = p->iters;
    while (left > 0) {
        const int nq = std::min(left, chunk);
        float ms = 0;
        rc = desc_pgd_iterate_timed(h, nq, &ms, nullptr); if (rc) return rc;
        left -= nq;
        if (left > 0 && h->m_pos > 0) {
            DevState st{};
            flush_finalize(h);
            DESC_HIP(hipStreamSynchronize(h->stream));
            DESC_HIP(hipMemcpy(&st, h->d_state, sizeof st, hipMemcpyDeviceToHost));
            // the objective of iteration t becomes known with sweep t+1: lines up to t_done - 1 (or the stop iteration) are final
            if (stream_lines && (rc = report_upto(st.stop ? st.iters_run : h->t_done - 1))) return rc;
            if (st.stop) break;
        }
    }
    rc = desc_pgd_download(h, r); if (rc) return rc;
    if (stream_lines) { if ((rc = report_upto(r->iters_run))) return rc; }
    else if ((p->verbose || p->progress) && r->obj_trace && r->avg_change_trace)      // no cycles at all: constant traces
        for (int it = 1; it <= r->iters_run; ++it) emit(it, r->avg_change_trace[it - 1], r->obj_trace[it - 1]);
    r->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DESC_OK;
    });
}

// ------------------------------------------------------------- multi-GPU pieces --
int desc_pgd_shard_info(const desc_pgd* h, desc_shard_info* info) {
    if (!h || !info) return fail(DESC_ERR_INVALID, "NULL argument");
    info->rank = h->rank; info->world = h->world;
    info->t_len = (int64_t)h->world * h->xparts * h->t_part + 1; info->t_part = h->t_part; info->slice_len = h->slice_len; info->xparts = h->xparts;
    info->seg_lo = h->seg_lo; info->seg_hi = h->seg_hi; info->cyc_lo = h->cyc_lo; info->cyc_hi = h->cyc_hi;
    info->m_pos = h->m_pos; info->m_cycle = h->m_cycle;
    return DESC_OK;
}

int desc_pgd_shard_bind(desc_pgd* h, double* T_send, double* T_recv, double* sall, void* hip_stream) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    if (h->variant != VARIANT_NODE) return fail(DESC_ERR_STATE, "sharding needs the node layout");
    int rc = set_device(h); if (rc) return rc;
    DESC_HIP(hipStreamSynchronize(h->stream));
    if (sweep_parts(h, false) > SHARD_PARTS || h->grid > SHARD_PARTS) return fail(DESC_ERR_STATE, "sweep grid exceeds the partials area of the exchange slice");
    if (!T_send && !T_recv && !sall) {       // library-owned exchange buffers (the fused protocol does not need torch tensors)
        const int64_t t_len = (int64_t)h->world * h->xparts * h->t_part + 1;
        if ((rc = dalloc(h, &T_send, (size_t)t_len))) return rc;
        if (h->world > 1) { if ((rc = dalloc(h, &T_recv, (size_t)h->xparts * h->t_part))) return rc; }
        else T_recv = T_send;                // one rank: the owner-sorted partial sums ARE the totals
        if ((rc = dalloc(h, &sall, (size_t)(h->world * h->slice_len)))) return rc;
        DESC_HIP(hipMemsetAsync(T_send, 0, sizeof(double) * t_len, h->stream));
        DESC_HIP(hipMemsetAsync(sall, 0, sizeof(double) * h->world * h->slice_len, h->stream));
        DESC_HIP(hipStreamSynchronize(h->stream));
        h->own_xbuf = true;
    } else if (!T_send || !T_recv || !sall) return fail(DESC_ERR_INVALID, "NULL exchange buffer");
    h->x_T = T_send; h->x_Trecv = T_recv; h->x_sall = sall;
    if (hip_stream) {          // run on the caller's stream so collectives and kernels are ordered
        if (h->stream) { (void)hipStreamSynchronize(h->stream); stream_release(h->stream); }
        h->stream = (hipStream_t)hip_stream;
        h->borrowed_stream = true;
    }
    return DESC_OK;
}

namespace {
// the pieces of one sharded iteration, each enqueued on the stream given
// One rank and nothing forcing the exchange layout: the column sums stay in CSR order (coalesced row writes) exactly as in
// the unsharded path.  With more ranks every rank writes all 2m slots of the owner-sorted layout, 8 bytes at a time
// (measured at C4: +74 us on the column-sum pass, independent of the number of ranks).
bool shard_direct(const desc_pgd* h) { return h->world == 1 && !h->force_coll && h->comm_stream != nullptr; }
int shard_enqueue_colsum(desc_pgd* h, hipStream_t st) {
    const int rd = h->t_done & 1;
    const bool direct = shard_direct(h);
    launch_colsum(h, st, h->d_w[rd], direct ? h->d_T : h->x_T, direct ? nullptr : h->d_xpos, FinArgs{});
    DESC_HIP(hipGetLastError());
    return DESC_OK;
}
double* my_slice(desc_pgd* h) { return h->x_sall + (int64_t)h->rank * h->slice_len; }
// sweep number t of the exchange parts [part_lo, part_hi): one launch each (part c reads the sums of ITS reduce-scatter: x_Trecv + c * t_part).
// first: this call opens sweep t (plugin step, iteration counter); the parts of one sweep share the step.
int shard_enqueue_sweep(desc_pgd* h, hipStream_t st, int part_lo = 0, int part_hi = -1) {
    if (part_hi < 0) part_hi = h->xparts;
    const bool first = part_lo == 0;
    if (first && h->t_done + 1 > h->iters_cap) return fail(DESC_ERR_INVALID, "iterating past params.iters = %d", h->p.iters);
    if (first) { ++h->t_done; h->cur_step = make_step(h, &h->cur_adam, (h->t_done - 1) & 1, h->t_done & 1); }
    const int t = h->t_done, rd = (t - 1) & 1, wr = t & 1;
    const bool adam = h->cur_adam;
    // a handle on the band sweep runs its parts as separate launches; k_sweep_node (tiny graphs, Adam on long segments) has one part
    const bool by_parts = adam ? band_adam_ok(h) : h->band_ok;
    for (int c = part_lo; c < part_hi; ++c) {
        if (!by_parts && c > 0) break;
        NodeSweepArgs a{};
        a.cum = h->d_cum; a.einfo = h->d_einfo; a.pk = h->d_pk; a.S0 = h->d_S0; a.w_old = h->d_w[rd]; a.w_new = h->d_w[wr];
        a.S_old = h->d_S[rd]; a.S_new = h->d_S[wr]; a.xt = h->d_xt; a.nv_tab = h->d_nv;
        a.Tfull = h->x_Trecv + (int64_t)c * h->t_part;
        // S and the workgroup partials go straight into the slice: the part's S behind the parts before it, its partials in its share of the partial area
        a.s_slice = my_slice(h) + h->xslice_off[c];
        a.partials = my_slice(h) + h->slice_S + 2 * (int64_t)c * (SHARD_PARTS / h->xparts);
        a.csr_bytes = (uint32_t)(16 * h->m); a.t_bytes = (uint32_t)(8 * h->t_part); a.slice_bytes = (uint32_t)(8 * (h->slice_S - h->xslice_off[c]));
        a.seg_count = (uint32_t)h->m_pos; a.fx_inv = std::ldexp(1.0, -h->colsum_fx_bits);
        if (shard_direct(h)) { a.Tfull = h->d_T; a.xt = nullptr; a.s_slice = nullptr; a.t_bytes = a.csr_bytes; }
        a.state = h->d_state; a.st = h->cur_step; a.chunk_desc = h->d_chunk_desc + h->ch_lo; a.nchunks = h->nchunks;
        a.max_cnt = h->max_cnt;
        const hipStream_t keep = h->stream; h->stream = st;
        launch_sweep_node_layout(h, a, adam, c);
        h->stream = keep;
    }
    h->last_parts = sweep_parts(h, adam);
    DESC_HIP(hipGetLastError());
    return DESC_OK;
}
// after the all-gather: S of every edge into the CSR-aligned copy (+ traces and stop rule of sweep t when fin_t > 0)
int shard_enqueue_unpack(desc_pgd* h, hipStream_t st, double* S_a, double* S_b, int fin_t, int last_only, int /*nparts*/) {
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (2 * h->m + 255) / 256));
    // the whole partial area of every slice is added (sweep grids may differ between ranks on tiny problems; unused pairs are zero)
    FinArgs f = fin_args(h, h->x_sall + h->slice_S, SHARD_PARTS, fin_t, last_only);
    f.rank_stride = h->slice_len; f.nranks = h->world;
    hipLaunchKernelGGL(k_unpack_S, dim3(g + 1), dim3(256), 0, st, h->d_spos, (int64_t)2 * h->m, h->x_sall, S_a, S_b, f);
    if (last_only)        // the objective pass filled more pairs than a sweep does: clear this rank's area for further iterations
        DESC_HIP(hipMemsetAsync(my_slice(h) + h->slice_S, 0, sizeof(double) * 2 * SHARD_PARTS, st));
    DESC_HIP(hipGetLastError());
    return DESC_OK;
}
int shard_enqueue_objective(desc_pgd* h, hipStream_t st) {
    const int T = h->t_done;
    // exactly SHARD_PARTS workgroups on every rank (idle ones write zeros): the partial areas of all slices are summed in full
    hipLaunchKernelGGL(k_objective_node, dim3(SHARD_PARTS), dim3(256), 0, st, h->d_cum + h->seg_lo, h->d_einfo + h->seg_lo, h->d_pk,
                       h->d_w[T & 1], h->d_S[T & 1], (int)(h->seg_hi - h->seg_lo), my_slice(h) + h->slice_S, h->d_state);
    DESC_HIP(hipGetLastError());
    return DESC_OK;
}
}  // namespace

// step 1 of an iteration: partial column sums of the segments this rank owns -> T_send, grouped by the
// rank that owns each edge (then: reduce-scatter(sum) T_send -> T_recv)
int desc_pgd_shard_colsum(desc_pgd* h) {
    if (!h || !h->armed || !h->x_T) return fail(DESC_ERR_STATE, "shard not armed / bound");
    int rc = set_device(h); if (rc) return rc;
    return shard_enqueue_colsum(h, h->stream);
}

// step 2: sweep the owned chunks (T_recv holds the global sums); S of the owned edges and the workgroup partials
// are written straight into this rank's slice of sall (then: all-gather sall)
int desc_pgd_shard_sweep(desc_pgd* h) {
    if (!h || !h->armed || !h->x_T) return fail(DESC_ERR_STATE, "shard not armed / bound");
    int rc = set_device(h); if (rc) return rc;
    return shard_enqueue_sweep(h, h->stream);
}

// step 3: (sall gathered) scatter S of every edge into the CSR-aligned copy, add the partials of all ranks in
// rank order, record traces, early-stop rule.
// initial: 1 = after desc_pgd_reset, before the first all-gather (the reset already put the initial S of the owned
// edges into the slice: nothing to do); 2 = after that all-gather: distribute the initial S_vec, no traces.
int desc_pgd_shard_finish(desc_pgd* h, int32_t initial) {
    if (!h || !h->armed || !h->x_sall) return fail(DESC_ERR_STATE, "shard not armed / bound");
    int rc = set_device(h); if (rc) return rc;
    const int t = h->t_done, wr = t & 1;
    if (initial && t != 0) return fail(DESC_ERR_STATE, "initial exchange after iterations");
    if (initial == 1) return DESC_OK;
    if (initial == 2) return shard_enqueue_unpack(h, h->stream, h->d_S[0], h->d_S[1], 0, 0, 0);
    return shard_enqueue_unpack(h, h->stream, h->d_S[wr], nullptr, t, 0, h->last_parts);
}

// final objective of a sharded run: phase 0 puts this rank's partials into its slice (then: all-gather sall),
// phase 1 adds the partials of all ranks and runs the stop rule for the last iteration.
int desc_pgd_shard_objective(desc_pgd* h, int32_t phase) {
    if (!h || !h->armed || !h->x_sall) return fail(DESC_ERR_STATE, "shard not armed / bound");
    int rc = set_device(h); if (rc) return rc;
    const int T = h->t_done;
    if (T < 1) return DESC_OK;
    if (phase == 0) return shard_enqueue_objective(h, h->stream);
    rc = shard_enqueue_unpack(h, h->stream, nullptr, nullptr, T, 1, h->obj_grid);
    h->objective_done = true; h->final_obj_T = T;
    return rc;
}

// ---- the fused protocol: whole iterations enqueued from C, exchange overlapped with the next column-sum pass
int desc_pgd_shard_set_collectives(desc_pgd* h, const desc_collectives* c) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    if (h->variant != VARIANT_NODE) return fail(DESC_ERR_STATE, "sharding needs the node layout");
    if (h->world > 1 && (!c || !c->reduce_scatter || !c->all_gather)) return fail(DESC_ERR_INVALID, "world > 1 needs both collectives");
    int rc = set_device(h); if (rc) return rc;
    h->coll = c ? *c : desc_collectives{};
    // diagnostics: call the collectives even with one rank (plumbing test of the RCCL entry points on a one-GPU box)
    h->force_coll = env_int("DESC_DEBUG_FORCE_COLLECTIVES", 0) != 0 && h->coll.reduce_scatter && h->coll.all_gather;
    if (!h->x_sall && (rc = desc_pgd_shard_bind(h, nullptr, nullptr, nullptr, nullptr))) return rc;
    if (!h->comm_stream) {
        DESC_HIP(stream_acquire(&h->comm_stream));
        for (hipEvent_t* e : {&h->ev_col, &h->ev_rs, &h->ev_sw, &h->ev_ag}) DESC_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        for (int c = 0; c < 8; ++c) DESC_HIP(hipEventCreateWithFlags(&h->ev_rsx[c], hipEventDisableTiming));
    }
    return DESC_OK;
}

namespace {
int coll_fail(int code, const char* what) { return fail(DESC_ERR_HIP, "%s failed with code %d", what, code); }
int shard_all_gather(desc_pgd* h) {
    if (h->world == 1 && !h->force_coll) return DESC_OK;
    const int rcc = h->coll.all_gather(my_slice(h), h->x_sall, (size_t)h->slice_len, 8 /* ncclDouble */, h->coll.comm, h->comm_stream);
    return rcc ? coll_fail(rcc, "all_gather") : DESC_OK;
}
}  // namespace

// reset + the initial exchange (every rank ends up with the full initial S_vec)
int desc_pgd_shard_start(desc_pgd* h, const desc_params* p) {
    if (!h || !p) return fail(DESC_ERR_INVALID, "NULL argument");
    if (!h->comm_stream) return fail(DESC_ERR_STATE, "desc_pgd_shard_set_collectives must be called first");
    int rc = desc_pgd_reset(h, p); if (rc) return rc;
    DESC_HIP(hipEventRecord(h->ev_sw, h->stream));
    DESC_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_sw, 0));
    if ((rc = shard_all_gather(h))) return rc;
    DESC_HIP(hipEventRecord(h->ev_ag, h->comm_stream));
    DESC_HIP(hipStreamWaitEvent(h->stream, h->ev_ag, 0));
    if ((rc = shard_enqueue_unpack(h, h->stream, h->d_S[0], h->d_S[1], 0, 0, 0))) return rc;
    h->unpack_pending = 0;
    return DESC_OK;
}

namespace {
// the all-gather of sweep t has been enqueued on the exchange stream (ev_ag behind it): its unpacking + the stop rule, on the compute stream
int shard_unpack_pending(desc_pgd* h) {
    if (!h->unpack_pending) return DESC_OK;
    const int t = h->unpack_pending;
    h->unpack_pending = 0;
    DESC_HIP(hipStreamWaitEvent(h->stream, h->ev_ag, 0));
    // one rank on the direct path: the sweep already wrote S of every edge; only the bookkeeping is left
    return shard_enqueue_unpack(h, h->stream, shard_direct(h) ? nullptr : h->d_S[t & 1], nullptr, t, 0, h->last_parts);
}
}  // namespace

// n iterations (round 4):
//   compute stream   colsum(t+1) | unpack(t) + stop rule | sweep(t+1)
//   exchange stream  AG(t)       | RS(t+1) . . . . . . . |            AG(t+1)
// colsum(t+1) needs only the weights of sweep t: it runs under the all-gather of S(t); the unpacking of S(t) runs under the reduce-scatter
// of the mirror sums.  (Round 3 had the unpack on the exchange stream IN FRONT of the reduce-scatter: measured per rank of 8 at C4,
// profiles/r04_shard_w8_c4_*.json, all-gather + unpack (~60 + 90 us) outlast the column sums (50-80 us), so the reduce-scatter started late.)
int desc_pgd_shard_iterate(desc_pgd* h, int32_t n_iters) {
    if (!h || !h->armed) return fail(DESC_ERR_STATE, "desc_pgd_shard_start must be called first");
    if (!h->comm_stream) return fail(DESC_ERR_STATE, "desc_pgd_shard_set_collectives must be called first");
    if (n_iters < 0 || h->t_done + n_iters > h->iters_cap) return fail(DESC_ERR_INVALID, "iterating past params.iters = %d", h->p.iters);
    int rc = set_device(h); if (rc) return rc;
    for (int q = 0; q < n_iters; ++q) {
        if ((rc = shard_enqueue_colsum(h, h->stream))) return rc;
        const bool exchange = h->world > 1 || h->force_coll;
        if (exchange) {
            DESC_HIP(hipEventRecord(h->ev_col, h->stream));
            DESC_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_col, 0));
            for (int c = 0; c < h->xparts; ++c) {                          // part c: blocks [c * world, (c + 1) * world) of the send buffer
                const int rcc = h->coll.reduce_scatter(h->x_T + (int64_t)c * h->world * h->t_part, h->x_Trecv + (int64_t)c * h->t_part, (size_t)h->t_part,
                                                       4 /* ncclInt64: fixed-point mirror sums, order-independent */, 0 /* ncclSum */, h->coll.comm, h->comm_stream);
                if (rcc) return coll_fail(rcc, "reduce_scatter");
                DESC_HIP(hipEventRecord(h->ev_rsx[c], h->comm_stream));
            }
        }
        if ((rc = shard_unpack_pending(h))) return rc;                  // S of the previous sweep into the CSR-aligned replica, under the reduce-scatter
        for (int c = 0; c < h->xparts; ++c) {                           // part c is swept while part c + 1 is still on the wire
            if (exchange) DESC_HIP(hipStreamWaitEvent(h->stream, h->ev_rsx[c], 0));
            if ((rc = shard_enqueue_sweep(h, h->stream, c, c + 1))) return rc;
        }
        DESC_HIP(hipEventRecord(h->ev_sw, h->stream));
        DESC_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_sw, 0));
        if ((rc = shard_all_gather(h))) return rc;
        DESC_HIP(hipEventRecord(h->ev_ag, h->comm_stream));
        h->unpack_pending = h->t_done;
    }
    return DESC_OK;
}

// start + iterate (polling the stop flag every check_every iterations) + final objective + download
int desc_pgd_shard_run(desc_pgd* h, const desc_params* p, desc_result* r) {
    if (!h || !p || !r) return fail(DESC_ERR_INVALID, "NULL argument");
    auto t0 = std::chrono::steady_clock::now();
    int rc = desc_pgd_shard_start(h, p); if (rc) return rc;
    const int chunk = p->check_every > 0 ? p->check_every : 32;
    int left = p->iters;
    while (left > 0) {
        const int nq = std::min(left, chunk);
        if ((rc = desc_pgd_shard_iterate(h, nq))) return rc;
        left -= nq;
        if (left > 0) { int32_t stop = 0; if ((rc = desc_pgd_stopped(h, &stop))) return rc; if (stop) break; }
    }
    if (h->t_done >= 1) {
        if ((rc = shard_unpack_pending(h))) return rc;
        if ((rc = shard_enqueue_objective(h, h->stream))) return rc;
        DESC_HIP(hipEventRecord(h->ev_sw, h->stream));
        DESC_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_sw, 0));
        if ((rc = shard_all_gather(h))) return rc;
        DESC_HIP(hipEventRecord(h->ev_ag, h->comm_stream));
        DESC_HIP(hipStreamWaitEvent(h->stream, h->ev_ag, 0));
        if ((rc = shard_enqueue_unpack(h, h->stream, nullptr, nullptr, h->t_done, 1, h->obj_grid))) return rc;
        h->objective_done = true; h->final_obj_T = h->t_done;
    }
    rc = desc_pgd_download(h, r); if (rc) return rc;
    r->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DESC_OK;
}

// 1 once the device-side patience rule has fired (synchronises the handle's stream)
// Diagnostics (tools/wg_clock.py): {start, end} of every workgroup of the LAST band sweep on the constant 100 MHz clock, when the handle
// was created with DESC_DEBUG_WGCLOCK=1.  Returns the number of workgroups written (0: not recorded).
int desc_debug_wg_clock(desc_pgd* h, uint64_t* out, int32_t cap) {
    if (!h || !out) return fail(DESC_ERR_INVALID, "NULL argument");
    if (!h->d_wg_clock || !h->band_ok) return 0;
    if (set_device(h)) return 0;
    const int nwg = std::min(cap, h->band_grid);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return 0;
    if (hipMemcpy(out, h->d_wg_clock, sizeof(uint64_t) * 2 * (size_t)nwg, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (2 * (int64_t)cap >= 3 * (int64_t)h->band_grid && nwg == h->band_grid &&          // room in out[2 * cap] for a third word per workgroup: its shader-clock cycles, at out[2 * nwg + w]
        hipMemcpy(out + 2 * (size_t)nwg, h->d_wg_clock + 2 * (size_t)nwg, sizeof(uint64_t) * (size_t)nwg, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return nwg;
}

// Diagnostics (tools/wg_clock.py): what the piece scheduler gave every workgroup of the band sweep -- out[4 * w + {0,1,2,3}] = cycles,
// segments, pieces, CSR entries of the band rows its pieces load.  Returns the number of workgroups.
int desc_debug_wg_plan(desc_pgd* h, int64_t* out, int32_t cap) {
    return no_throw("desc_debug_wg_plan", [&]() -> int {
    if (!h || !out) return fail(DESC_ERR_INVALID, "NULL argument");
    if (!h->band_ok || !h->d_pieces) return 0;
    if (set_device(h)) return 0;
    const int G = std::min(cap, h->band_grid);
    hvec<int32_t> pp((size_t)h->band_grid + 1), cum((size_t)h->m_pos + 1);
    if (hipMemcpy(pp.data(), h->d_piece_ptr, sizeof(int32_t) * pp.size(), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (hipMemcpy(cum.data(), h->d_cum, sizeof(int32_t) * cum.size(), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    hvec<PieceDesc> pc((size_t)pp[h->band_grid]);
    if (!pc.empty() && hipMemcpy(pc.data(), h->d_pieces, sizeof(PieceDesc) * pc.size(), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (int w = 0; w < G; ++w) {
        int64_t cyc = 0, seg = 0, rows = 0;
        for (int q = pp[w]; q < pp[w + 1]; ++q) { cyc += cum[pc[q].seg_hi] - cum[pc[q].seg_lo]; seg += pc[q].seg_hi - pc[q].seg_lo; rows += pc[q].row_len; }
        out[4 * w] = cyc; out[4 * w + 1] = seg; out[4 * w + 2] = pp[w + 1] - pp[w]; out[4 * w + 3] = rows;
    }
    return G;
    });
}

// What the layout of this handle moves per iteration (bench.py: the floor of its own traffic, next to the 72-byte algorithmic count):
// out[0] = (cycle, endpoint) pairs read by the column sums, out[1] = pieces of the band sweep, out[2] = bands, out[3] = CSR entries of band
// rows loaded into the LDS per sweep, out[4] = local cycles, out[5] = local segments.  Returns the number of values written.
int desc_pgd_layout_stats(const desc_pgd* h, int64_t* out, int32_t cap) {
    if (!h || !out) return fail(DESC_ERR_INVALID, "NULL argument");
    const int64_t v[6] = {h->colsum_entries, h->n_pieces, h->n_bands, h->piece_row_entries, local_cycles(h), h->variant == VARIANT_NODE ? h->seg_hi - h->seg_lo : h->m_pos};
    int k = 0;
    for (; k < 6 && k < cap; ++k) out[k] = v[k];
    return k;
}

// Diagnostics (tests): name and template arguments of the sweep kernel this handle launched last ("" before the first sweep).
const char* desc_debug_last_sweep(const desc_pgd* h) { return h ? h->last_sweep.c_str() : ""; }

// Diagnostics (tests/test_gpu_sharded.py): the exchange layout of a sharded handle.  xpos, spos: 2m entries each (CSR slot -> place of its
// column sum in the reduce-scatter send buffer / place of its edge's S in the gathered slices); xt: 2 * (seg_hi - seg_lo) entries, {ta, tb}
// of the owned segments in device order; slot_ab: the same count, {slot_a, slot_b} = the CSR slots of those segments' edges.
int desc_debug_shard_layout(desc_pgd* h, int32_t* xpos, int32_t* spos, int32_t* xt, int32_t* slot_ab) {
    return no_throw("desc_debug_shard_layout", [&]() -> int {
    if (!h || !xpos || !spos || !xt || !slot_ab) return fail(DESC_ERR_INVALID, "NULL argument");
    if (h->variant != VARIANT_NODE) return fail(DESC_ERR_STATE, "sharding needs the node layout");
    int rc = set_device(h); if (rc) return rc;
    DESC_HIP(hipStreamSynchronize(h->stream));
    if (h->m > 0) {
        DESC_HIP(hipMemcpy(xpos, h->d_xpos, sizeof(int32_t) * 2 * (size_t)h->m, hipMemcpyDeviceToHost));
        DESC_HIP(hipMemcpy(spos, h->d_spos, sizeof(int32_t) * 2 * (size_t)h->m, hipMemcpyDeviceToHost));
    }
    const int64_t nsl = h->seg_hi - h->seg_lo;
    if (nsl > 0) {
        DESC_HIP(hipMemcpy(xt, h->d_xt + h->seg_lo, sizeof(int2) * (size_t)nsl, hipMemcpyDeviceToHost));
        hvec<EdgeInfo> ei((size_t)nsl);
        DESC_HIP(hipMemcpy(ei.data(), h->d_einfo + h->seg_lo, sizeof(EdgeInfo) * (size_t)nsl, hipMemcpyDeviceToHost));
        for (int64_t q = 0; q < nsl; ++q) { slot_ab[2 * q] = ei[q].slot_a; slot_ab[2 * q + 1] = ei[q].slot_b; }
    }
    return DESC_OK;
    });
}

int desc_pgd_stopped(desc_pgd* h, int32_t* stopped) {
    if (!h || !stopped) return fail(DESC_ERR_INVALID, "NULL argument");
    int rc = set_device(h); if (rc) return rc;
    DevState st{};
    flush_finalize(h);
    if ((rc = shard_unpack_pending(h))) return rc;
    if (h->comm_stream) DESC_HIP(hipStreamSynchronize(h->comm_stream));
    DESC_HIP(hipStreamSynchronize(h->stream));
    DESC_HIP(hipMemcpy(&st, h->d_state, sizeof st, hipMemcpyDeviceToHost));      // synchronous: `st` is a stack slot
    *stopped = st.stop;
    return DESC_OK;
}


int desc_device_synchronize(int32_t device) {
    DESC_HIP(hipSetDevice(device));
    DESC_HIP(hipDeviceSynchronize());
    return DESC_OK;
}
int desc_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes) {
    if (bytes) DESC_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return DESC_OK;
}
int desc_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes) {
    if (bytes) DESC_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return DESC_OK;
}

// Test hook, host only (no device call): the band-sweep work plan of `rank` of `world` for `grid` workgroups -- bands, ranks' segment
// ranges, pieces -- checked for its invariants (every owned segment in exactly one piece, a piece inside one band, band rows within
// the LDS budget).  stats[8] = {bands, pieces, largest band's row entries, max / min cycles per workgroup, j-block-major?, seg_lo, seg_hi}.
int desc_debug_band_plan(const desc_problem* prob, const desc_structure* s, int32_t world, int32_t rank, int32_t grid, int64_t* stats) {
    if (!prob || !s || !stats || world < 1 || rank < 0 || rank >= world || grid < 1) return fail(DESC_ERR_INVALID, "bad argument");
    if (s->m_pos == 0) return fail(DESC_ERR_INVALID, "no edge with cycles");
    int32_t max_deg = 0;
    {
        hvec<int32_t> deg((size_t)prob->n, 0);
        for (int64_t e = 0; e < prob->m; ++e) { deg[prob->ind_i[e]]++; deg[prob->ind_j[e]]++; }
        for (int32_t d : deg) max_deg = std::max(max_deg, d);
    }
    if (max_deg > BAND_ROW_CAP) return fail(DESC_ERR_TOO_LARGE, "a row exceeds the LDS budget");
    NodePlan P;
    int rc = make_node_plan(prob, s, max_deg, world, s->max_cnt <= 32 ? 32 : s->max_cnt <= 128 ? 16 : 8, band_row_cap(max_deg), P);
    if (rc) return rc;
    const int64_t seg_lo = P.chunk_seg[P.rank_chunk[rank]], seg_hi = P.chunk_seg[P.rank_chunk[rank + 1]];
    const int64_t cyc_lo = P.cum2[seg_lo], mcl = P.cum2[seg_hi] - cyc_lo;
    hvec<PieceDesc> pieces; hvec<int32_t> piece_ptr; int band_rows = 0; bool jmajor = false;
    plan_band_pieces(prob, s, P, seg_lo, seg_hi, cyc_lo, mcl, grid, pieces, piece_ptr, band_rows, jmajor);
    const int64_t nbands = (int64_t)P.band_lo.size() - 1;
    hvec<uint8_t> seen((size_t)(seg_hi - seg_lo), 0);
    int64_t wmax = 0, wmin = INT64_MAX;
    for (int b = 0; b < grid; ++b) {
        int64_t cyc = 0;
        for (int32_t pc = piece_ptr[b]; pc < piece_ptr[b + 1]; ++pc) {
            const PieceDesc& pd = pieces[pc];
            if (pd.seg_lo < seg_lo || pd.seg_hi > seg_hi || pd.seg_lo >= pd.seg_hi) return fail(DESC_ERR_STATE, "piece %d out of the rank's range", pc);
            int64_t bd = 0;
            while (bd + 1 < nbands && P.bstart[bd + 1] <= pd.seg_lo) ++bd;
            if (pd.seg_hi > P.bstart[bd + 1]) return fail(DESC_ERR_STATE, "piece %d straddles two bands", pc);
            if (pd.row_lo != P.rowptr[P.band_lo[bd]] || pd.row_len != P.rowptr[P.band_lo[bd + 1]] - pd.row_lo || pd.row_len > BAND_ROW_CAP)
                return fail(DESC_ERR_STATE, "piece %d carries the wrong band rows", pc);
            for (int64_t q = pd.seg_lo; q < pd.seg_hi; ++q) {
                const int32_t i = prob->ind_i[s->pos_edge[P.order[q]]];
                if (i < P.band_lo[bd] || i >= P.band_lo[bd + 1]) return fail(DESC_ERR_STATE, "segment %lld is not in the band of its piece", (long long)q);
                if (seen[q - seg_lo]++) return fail(DESC_ERR_STATE, "segment %lld is in two pieces", (long long)q);
            }
            cyc += P.cum2[pd.seg_hi] - P.cum2[pd.seg_lo];
        }
        wmax = std::max(wmax, cyc); wmin = std::min(wmin, cyc);
    }
    for (size_t q = 0; q < seen.size(); ++q) if (!seen[q]) return fail(DESC_ERR_STATE, "segment %lld is in no piece", (long long)(q + seg_lo));
    stats[0] = nbands; stats[1] = (int64_t)pieces.size(); stats[2] = band_rows; stats[3] = wmax; stats[4] = wmin; stats[5] = jmajor ? 1 : 0;
    stats[6] = seg_lo; stats[7] = seg_hi;
    return DESC_OK;
}

// test hook: group_sum over a buffer of 64*k doubles
int desc_selftest_group_sum(const double* in, double* out, int32_t count, int32_t G, int32_t device) {
    if (!in || !out || count <= 0 || count % 64) return fail(DESC_ERR_INVALID, "count must be a positive multiple of 64");
    if (G != 16 && G != 32 && G != 64) return fail(DESC_ERR_INVALID, "G must be 16, 32 or 64");
    DESC_HIP(hipSetDevice(device));
    double *di = nullptr, *dout = nullptr;
    DESC_HIP(hipMalloc((void**)&di, sizeof(double) * count));
    DESC_HIP(hipMalloc((void**)&dout, sizeof(double) * count));
    DESC_HIP(hipMemcpy(di, in, sizeof(double) * count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest_group_sum, dim3(count / 64), dim3(64), 0, 0, di, dout, G);
    DESC_HIP(hipDeviceSynchronize());
    DESC_HIP(hipMemcpy(out, dout, sizeof(double) * count, hipMemcpyDeviceToHost));
    (void)hipFree(di); (void)hipFree(dout);
    return DESC_OK;
}

}  // extern "C"
