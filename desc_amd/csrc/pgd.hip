// HIP kernels and solver handle for the DESC projected-gradient hot path on gfx950.
//
// Reference text reproduced (Algorithms/DESC_PGD.m, identical in DESC.m:16-261):
//   k_cycle_d        :129-147  cycle inconsistency S0_long = |acos((tr(Rij Rjk Rki)-1)/2)|/pi
//   k_init           :148-157  wijk = 1/cnt, S_vec = 1 / segment mean of S0
//   k_sweep<G,STEP>  :185-230  mirror sums, gradient, tangent projection, plugin step
//                              (Utils/ConstantStepSize.m:9-11, PiecewiseStepSize.m:13-18,
//                              HybridGradient.m:23-41), simplex projection, new S_vec
//   k_sweep_big<STEP>          the same for segments longer than 64 cycles
//   k_objective      :233      obj = wijk*(S_vec(Ind_jk)'+S_vec(Ind_ki)')
//   k_finalize       :232,243-257  average_change, early-stop bookkeeping (on device)
//
// Data layout in HBM (all struct-of-arrays, cycles of one edge contiguous, edges in
// Ind order, third vertices ascending inside a segment):
//   per cycle : e_jk,e_ki,ikj,jki int32; S0 f64; w[2] f64 (Jacobi double buffer)
//   per edge  : S[2] f64 (double buffer), pos_edge int32, cum int32
// One sweep moves 72 B per cycle: w r/w 16, S0 8, 4 index words 16,
// 2 gathers of S 16, 2 gathers of w 16.  HBM-bound; no MFMA.
//
// The objective of iteration t needs S_vec of *every* edge after iteration t, so it
// cannot be fused into sweep t; it is accumulated for free inside sweep t+1 (which
// gathers exactly those values) and once more by k_objective after the last sweep.
// The early-stop test of iteration t is therefore evaluated on the device during
// sweep t+1; when it fires, sweep t+1's output is discarded (the Jacobi double
// buffers still hold iteration t) and later launches return at once.
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "device_utils.h"

namespace desc {

struct DevState {
    int32_t stop;          // 1 once the patience rule fired
    int32_t misses;        // DESC_PGD.m:181
    int32_t iters_run;     // iteration at which the loop broke
    int32_t final_parity;  // which double buffer holds the final iterate
};

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
    double* adam_m;
    double* adam_v;
    const double* nv_tab;     // nv_tab[c] = 1/sqrt(c)   (DESC_PGD.m:199)
    double* partials;         // [grid][2]: objective of the old iterate, sum |dS|
    const DevState* state;
    double step;              // step size of this call of GetStep
    double lr, beta1, beta2, bc1, bc2;   // Adam
    int32_t m_pos;
};

__device__ __forceinline__ double abs_acos_ext(double x) {
    // MATLAB abs(acos(x)) with the complex extension outside [-1,1] (DESC_PGD.m:147)
    if (x > 1.0) return acosh(x);
    if (x < -1.0) return hypot(M_PI, acosh(-x));
    return acos(x);
}

template <int STEP>
__device__ __forceinline__ double apply_step(const SweepArgs& a, double w, double g, int64_t c) {
    if (STEP == DESC_STEP_HYBRID) {               // HybridGradient.m:28-35 (strategy 0)
        double mt = (a.beta1 * a.adam_m[c]) + (1.0 - a.beta1) * g;
        double vt = (a.beta2 * a.adam_v[c]) + (1.0 - a.beta2) * (g * g);
        a.adam_m[c] = mt; a.adam_v[c] = vt;
        double cm = mt / a.bc1, cv = vt / a.bc2;
        return w + (-a.lr * cm / (sqrt(cv) + 1e-8));
    }
    return w + (-a.step * g);                     // ConstantStepSize.m:10 / PiecewiseStepSize.m:17
}

// ---------------------------------------------------------------------------
// Main sweep: G lanes per edge segment (cnt <= G), 64/G segments per wave step.
// ---------------------------------------------------------------------------
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
        // objective of the iterate being read (DESC_PGD.m:233, one sweep late)
        obj_acc += w * ssum;
        // mirror-weight sums: one scalar per edge, applied to masked positions only (:189-190)
        const double T1 = group_sum<G>(wa), T2 = group_sum<G>(wb);
        double g = ssum + ((ia >= 0 ? T1 : 0.0) + (ib >= 0 ? T2 : 0.0)) * d;          // :193
        // tangent projection grad - (grad*nv')*nv, nv = ones/sqrt(cnt)  (:199-201)
        const double nv = act0 ? a.nv_tab[cnt] : 0.0;
        const double dot = group_sum<G>(act0 ? g * nv : 0.0);
        g = g - dot * nv;
        double ws = act0 ? apply_step<STEP>(a, w, g, c) : 0.0;                        // :207

        // simplex projection (:215-224): threshold T with sum(max(w-T,0)) = 1.
        // Michelot's fixed point gives the same active set as the reference's
        // sort-and-scan (the first sorted i with sum(w(i:end)-w(i)) < 1).
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
        const double wn = act0 ? fmax(ws - T, 0.0) : 0.0;                             // :224
        const double snew = group_sum<G>(wn * d);                                     // :229
        if (act0) a.w_new[c] = wn;
        if (edge_ok && gl == 0) {
            const int e = a.pos_edge[l];
            chg_acc += fabs(snew - a.S_old[e]);                                       // :232
            a.S_new[e] = snew;
        }
    }
    // deterministic block partials
    obj_acc = group_sum<64>(obj_acc);
    chg_acc = group_sum<64>(chg_acc);
    __shared__ double sh[8];
    if (lane == 0) { sh[wv] = obj_acc; sh[4 + wv] = chg_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.partials[2 * lb] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        a.partials[2 * lb + 1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    }
}

// ---------------------------------------------------------------------------
// Fallback sweep for segments longer than 64 cycles: one wave per edge, several
// passes over the segment; w_new doubles as scratch (each lane re-reads only what
// it wrote itself).
// ---------------------------------------------------------------------------
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
            a.w_new[c] = apply_step<STEP>(a, a.w_old[c], g, c);
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
    obj_acc = group_sum<64>(obj_acc);
    chg_acc = group_sum<64>(chg_acc);
    __shared__ double sh[8];
    if (lane == 0) { sh[wv] = obj_acc; sh[4 + wv] = chg_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.partials[2 * lb] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        a.partials[2 * lb + 1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    }
}

// Sum the block partials in a fixed order, record the traces and run the early-stop
// rule of DESC_PGD.m:243-256 for the iteration whose objective just became known.
// t = 1-based index of the sweep that produced the partials.
__global__ __launch_bounds__(256) void k_finalize(const double* partials, int nparts, DevState* st,
                                                  double* obj_trace, double* avg_trace, int t, int64_t m,
                                                  int patience, double stop_tol, int last_only) {
    if (st->stop) return;
    __shared__ double sh[2][256];
    double o = 0.0, ch = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { o += partials[2 * i]; ch += partials[2 * i + 1]; }
    sh[0][threadIdx.x] = o; sh[1][threadIdx.x] = ch;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    // last_only: the partials come from k_objective after the final sweep t: they
    // hold obj(t).  Otherwise they come from sweep t: obj(t-1) and sum|dS| of sweep t.
    const int it = last_only ? t : t - 1;          // iteration whose objective is sh[0][0]
    if (!last_only) avg_trace[t - 1] = sh[1][0] / (double)m;                          // :232
    if (it >= 1) {
        obj_trace[it - 1] = sh[0][0];                                                 // :233
        if (it > 1 && obj_trace[it - 2] - obj_trace[it - 1] < stop_tol) {             // :243
            st->misses += 1;
            if (st->misses >= patience) {                                             // :245-246
                st->stop = 1; st->iters_run = it; st->final_parity = it & 1;
            }
        } else {
            st->misses = 0;                                                           // :255
        }
    }
}

__global__ __launch_bounds__(256) void k_objective(const double* w, const double* S, const int32_t* e_jk,
                                                   const int32_t* e_ki, int64_t m_cycle, double* partials,
                                                   const DevState* st) {
    if (st->stop) return;
    double acc = 0.0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < m_cycle; c += (int64_t)gridDim.x * 256)
        acc += w[c] * (S[e_jk[c]] + S[e_ki[c]]);
    acc = group_sum<64>(acc);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]); partials[2 * blockIdx.x + 1] = 0.0; }
}

__global__ void k_fill(double* p, int64_t n, double v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
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
// otherwise; likewise R_ki (:65-66,89-91).  Products are accumulated in the
// reference's order (sum over the middle index 1..3 starting from zero).
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
            const double* pb = rij + 9 * (int64_t)e_jk[c];
            const double* pc = rij + 9 * (int64_t)e_ki[c];
            double B[9], C[9];
            const bool tb = !(j < k), tc = !(k < i);
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) {
                    B[r + 3 * s] = tb ? pb[s + 3 * r] : pb[r + 3 * s];
                    C[r + 3 * s] = tc ? pc[s + 3 * r] : pc[r + 3 * s];
                }
            double tr = 0.0;
            for (int r = 0; r < 3; ++r) {
                // row r of P = A*B, then (P*C)(r,r)
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
            S0[c] = abs_acos_ext((tr - 1.0) / 2.0) / M_PI;
        }
    }
}

// self-test kernel for the group reductions (tests/test_gpu_primitives.py)
__global__ void k_selftest_group_sum(const double* in, double* out, int G) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const double v = in[t];
    out[t] = (G == 16) ? group_sum<16>(v) : (G == 32) ? group_sum<32>(v) : group_sum<64>(v);
}

}  // namespace desc

using namespace desc;

// ------------------------------------------------------------------- handle --
struct desc_pgd {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0, m = 0, m_pos = 0, m_cycle = 0;
    int32_t max_cnt = 0, n_sample = 0;
    int G = 64;                 // lanes per segment; 0 = big fallback
    int grid = 0;               // sweep grid (multiple of 8)
    int obj_grid = 0;
    // device buffers
    int32_t *d_cum = nullptr, *d_pos_edge = nullptr, *d_ejk = nullptr, *d_eki = nullptr, *d_ikj = nullptr, *d_jki = nullptr;
    double *d_S0 = nullptr, *d_w[2] = {nullptr, nullptr}, *d_S[2] = {nullptr, nullptr};
    double *d_adam_m = nullptr, *d_adam_v = nullptr, *d_nv = nullptr, *d_partials = nullptr;
    double *d_obj = nullptr, *d_avg = nullptr;
    DevState* d_state = nullptr;
    int trace_cap = 0;
    // run state
    desc_params p{};
    bool armed = false;
    int t_done = 0;             // sweeps enqueued since reset
    int t_plugin = 0;           // plugin counter (PiecewiseStepSize.t / HybridGradient.t)
    double ms_upload = 0, ms_cycle_d = 0, ms_pgd = 0;
    std::string kname;
};

namespace {

template <class T>
int dmalloc(T** p, size_t count) {
    *p = nullptr;
    DESC_HIP(hipMalloc((void**)p, sizeof(T) * (count > 0 ? count : 1)));
    return DESC_OK;
}

int set_device(const desc_pgd* h) { DESC_HIP(hipSetDevice(h->device)); return DESC_OK; }

void free_all(desc_pgd* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    void* ptrs[] = {h->d_cum, h->d_pos_edge, h->d_ejk, h->d_eki, h->d_ikj, h->d_jki, h->d_S0, h->d_w[0], h->d_w[1],
                    h->d_S[0], h->d_S[1], h->d_adam_m, h->d_adam_v, h->d_nv, h->d_partials, h->d_obj, h->d_avg, h->d_state};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

template <int STEP>
void launch_sweep(desc_pgd* h, const SweepArgs& a) {
    dim3 grid(h->grid), block(256);
    switch (h->G) {
        case 16: hipLaunchKernelGGL((k_sweep<16, STEP>), grid, block, 0, h->stream, a); break;
        case 32: hipLaunchKernelGGL((k_sweep<32, STEP>), grid, block, 0, h->stream, a); break;
        case 64: hipLaunchKernelGGL((k_sweep<64, STEP>), grid, block, 0, h->stream, a); break;
        default: hipLaunchKernelGGL((k_sweep_big<STEP>), grid, block, 0, h->stream, a); break;
    }
}

// enqueue sweep number t (1-based) and its finalize
int enqueue_sweep(desc_pgd* h, int t, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    const desc_params& p = h->p;
    const int rd = (t - 1) & 1, wr = t & 1;
    SweepArgs a{};
    a.cum = h->d_cum; a.pos_edge = h->d_pos_edge; a.e_jk = h->d_ejk; a.e_ki = h->d_eki; a.ikj = h->d_ikj; a.jki = h->d_jki;
    a.S0 = h->d_S0; a.w_old = h->d_w[rd]; a.w_new = h->d_w[wr]; a.S_old = h->d_S[rd]; a.S_new = h->d_S[wr];
    a.adam_m = h->d_adam_m; a.adam_v = h->d_adam_v; a.nv_tab = h->d_nv; a.partials = h->d_partials; a.state = h->d_state;
    a.m_pos = (int32_t)h->m_pos;
    // one GetStep call per iteration (DESC_PGD.m:207): the plugin counter advances first
    const int tp = ++h->t_plugin;
    a.lr = p.lr; a.beta1 = p.beta1; a.beta2 = p.beta2; a.bc1 = 1.0; a.bc2 = 1.0;
    a.step = p.lr;
    bool adam = false;
    if (p.step_kind == DESC_STEP_PIECEWISE) {
        a.step = p.lr / (std::trunc((double)tp / p.decay_interval) + 1.0);            // PiecewiseStepSize.m:16
    } else if (p.step_kind == DESC_STEP_HYBRID) {
        if (p.hybrid_strategy == 0) {
            adam = true;
            a.bc1 = 1.0 - std::pow(p.beta1, (double)tp);                             // HybridGradient.m:32-33
            a.bc2 = 1.0 - std::pow(p.beta2, (double)tp);
        } else {
            a.step = 100.0 * (p.lr / (std::trunc((double)tp / p.decay_interval) + 1.0));   // HybridGradient.m:39
        }
    }
    if (ev0) (void)hipEventRecord(ev0, h->stream);
    if (adam) launch_sweep<DESC_STEP_HYBRID>(h, a); else launch_sweep<DESC_STEP_CONSTANT>(h, a);
    if (ev1) (void)hipEventRecord(ev1, h->stream);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, h->stream, h->d_partials, h->grid, h->d_state, h->d_obj,
                       h->d_avg, t, h->m, p.patience, p.stop_tol, 0);
    DESC_HIP(hipGetLastError());
    return DESC_OK;
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
    if (!out) return fail(DESC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!prob || !s) return fail(DESC_ERR_INVALID, "NULL argument");
    if (prob->m != s->m) return fail(DESC_ERR_INVALID, "structure was built for m = %lld, problem has m = %lld", (long long)s->m, (long long)prob->m);
    if (prob->m > 0 && !prob->rij) return fail(DESC_ERR_INVALID, "rij is NULL");
    int ndev = desc_device_count();
    if (ndev < 0) return ndev;
    if (ndev == 0) return fail(DESC_ERR_HIP, "no HIP device visible: the DESC_PGD hot path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(DESC_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);

    desc_pgd* h = new (std::nothrow) desc_pgd();
    if (!h) return fail(DESC_ERR_INVALID, "out of host memory");
    h->device = device;
    h->n = s->n; h->m = s->m; h->m_pos = s->m_pos; h->m_cycle = s->m_cycle; h->max_cnt = s->max_cnt; h->n_sample = s->n_sample;
    int rc = DESC_OK;
    auto t0 = std::chrono::steady_clock::now();
#define TRY(x) do { rc = (x); if (rc) { free_all(h); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { rc = fail(DESC_ERR_HIP, "%s: %s", #x, hipGetErrorString(_e)); free_all(h); return rc; } } while (0)
    TRYHIP(hipSetDevice(device));
    TRYHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const int64_t m = h->m, mp = h->m_pos, mc = h->m_cycle;
    TRY(dmalloc(&h->d_cum, mp + 1)); TRY(dmalloc(&h->d_pos_edge, mp));
    TRY(dmalloc(&h->d_ejk, mc)); TRY(dmalloc(&h->d_eki, mc)); TRY(dmalloc(&h->d_ikj, mc)); TRY(dmalloc(&h->d_jki, mc));
    TRY(dmalloc(&h->d_S0, mc)); TRY(dmalloc(&h->d_w[0], mc)); TRY(dmalloc(&h->d_w[1], mc));
    TRY(dmalloc(&h->d_S[0], m)); TRY(dmalloc(&h->d_S[1], m));
    TRY(dmalloc(&h->d_nv, (size_t)h->max_cnt + 1));
    TRY(dmalloc(&h->d_state, 1));

    // lanes per segment and grid
    h->G = h->max_cnt <= 16 ? 16 : h->max_cnt <= 32 ? 32 : h->max_cnt <= 64 ? 64 : 0;
    const int epw = h->G ? 64 / h->G : 1;
    int64_t want = (mp + 4 * epw - 1) / (4 * epw);
    if (want > 2048) want = 2048;
    if (want < 8) want = 8;
    h->grid = (int)((want + 7) / 8 * 8);
    h->obj_grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (mc + 255) / 256));
    TRY(dmalloc(&h->d_partials, 2 * (size_t)std::max(h->grid, h->obj_grid)));
    char nm[64];
    if (h->G) snprintf(nm, sizeof nm, "k_sweep<%d,", h->G); else snprintf(nm, sizeof nm, "k_sweep_big<");
    h->kname = nm;

    // uploads
    {
        std::vector<int32_t> cum32((size_t)mp + 1);
        for (int64_t l = 0; l <= mp; ++l) cum32[l] = (int32_t)s->cum_ind[l];
        TRYHIP(hipMemcpyAsync(h->d_cum, cum32.data(), sizeof(int32_t) * (mp + 1), hipMemcpyHostToDevice, h->stream));
        TRYHIP(hipStreamSynchronize(h->stream));
    }
    int32_t *d_k = nullptr, *d_ii = nullptr, *d_jj = nullptr; double* d_rij = nullptr;
    auto cleanup_tmp = [&]() { if (d_k) (void)hipFree(d_k); if (d_ii) (void)hipFree(d_ii); if (d_jj) (void)hipFree(d_jj); if (d_rij) (void)hipFree(d_rij); };
#define TRY2(x) do { rc = (x); if (rc) { cleanup_tmp(); free_all(h); return rc; } } while (0)
#define TRYHIP2(x) do { hipError_t _e = (x); if (_e != hipSuccess) { rc = fail(DESC_ERR_HIP, "%s: %s", #x, hipGetErrorString(_e)); cleanup_tmp(); free_all(h); return rc; } } while (0)
    TRY2(dmalloc(&d_k, mc)); TRY2(dmalloc(&d_ii, m)); TRY2(dmalloc(&d_jj, m)); TRY2(dmalloc(&d_rij, 9 * (size_t)m));
    if (mp > 0) TRYHIP2(hipMemcpyAsync(h->d_pos_edge, s->pos_edge.data(), sizeof(int32_t) * mp, hipMemcpyHostToDevice, h->stream));
    if (mc > 0) {
        TRYHIP2(hipMemcpyAsync(h->d_ejk, s->e_jk.data(), sizeof(int32_t) * mc, hipMemcpyHostToDevice, h->stream));
        TRYHIP2(hipMemcpyAsync(h->d_eki, s->e_ki.data(), sizeof(int32_t) * mc, hipMemcpyHostToDevice, h->stream));
        TRYHIP2(hipMemcpyAsync(h->d_ikj, s->ikj.data(), sizeof(int32_t) * mc, hipMemcpyHostToDevice, h->stream));
        TRYHIP2(hipMemcpyAsync(h->d_jki, s->jki.data(), sizeof(int32_t) * mc, hipMemcpyHostToDevice, h->stream));
        TRYHIP2(hipMemcpyAsync(d_k, s->k.data(), sizeof(int32_t) * mc, hipMemcpyHostToDevice, h->stream));
    }
    if (m > 0) {
        TRYHIP2(hipMemcpyAsync(d_ii, prob->ind_i, sizeof(int32_t) * m, hipMemcpyHostToDevice, h->stream));
        TRYHIP2(hipMemcpyAsync(d_jj, prob->ind_j, sizeof(int32_t) * m, hipMemcpyHostToDevice, h->stream));
        TRYHIP2(hipMemcpyAsync(d_rij, prob->rij, sizeof(double) * 9 * m, hipMemcpyHostToDevice, h->stream));
    }
    {
        std::vector<double> nv((size_t)h->max_cnt + 1, 0.0);
        for (int c = 1; c <= h->max_cnt; ++c) nv[c] = 1.0 / std::pow((double)c, 0.5);   // ones/(nsample^0.5), DESC_PGD.m:199
        TRYHIP2(hipMemcpyAsync(h->d_nv, nv.data(), sizeof(double) * nv.size(), hipMemcpyHostToDevice, h->stream));
        TRYHIP2(hipStreamSynchronize(h->stream));
    }
    h->ms_upload = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    // a-4: cycle inconsistencies
    {
        hipEvent_t e0, e1;
        TRYHIP2(hipEventCreate(&e0)); TRYHIP2(hipEventCreate(&e1));
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
        if (e != hipSuccess) { rc = fail(DESC_ERR_HIP, "k_cycle_d: %s", hipGetErrorString(e)); cleanup_tmp(); free_all(h); return rc; }
    }
    cleanup_tmp();
#undef TRY
#undef TRYHIP
#undef TRY2
#undef TRYHIP2
    *out = h;
    return DESC_OK;
}

void desc_pgd_destroy(desc_pgd* h) { free_all(h); }

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
    if (h->m_cycle > 0) DESC_HIP(hipMemcpy(s0, h->d_S0, sizeof(double) * h->m_cycle, hipMemcpyDeviceToHost));
    return DESC_OK;
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
    h->t_done = 0; h->t_plugin = p->t0; h->ms_pgd = 0;
    const int cap = std::max(1, p->iters);
    if (cap > h->trace_cap) {
        if (h->d_obj) (void)hipFree(h->d_obj);
        if (h->d_avg) (void)hipFree(h->d_avg);
        h->d_obj = h->d_avg = nullptr; h->trace_cap = 0;
        rc = dmalloc(&h->d_obj, cap); if (rc) return rc;
        rc = dmalloc(&h->d_avg, cap); if (rc) return rc;
        h->trace_cap = cap;
    }
    if (p->step_kind == DESC_STEP_HYBRID && p->hybrid_strategy == 0 && !h->d_adam_m) {
        rc = dmalloc(&h->d_adam_m, h->m_cycle); if (rc) return rc;
        rc = dmalloc(&h->d_adam_v, h->m_cycle); if (rc) return rc;
    }
    DESC_HIP(hipMemsetAsync(h->d_state, 0, sizeof(DevState), h->stream));
    DESC_HIP(hipMemsetAsync(h->d_obj, 0, sizeof(double) * h->trace_cap, h->stream));
    DESC_HIP(hipMemsetAsync(h->d_avg, 0, sizeof(double) * h->trace_cap, h->stream));
    if (h->d_adam_m) {
        DESC_HIP(hipMemsetAsync(h->d_adam_m, 0, sizeof(double) * std::max<int64_t>(1, h->m_cycle), h->stream));
        DESC_HIP(hipMemsetAsync(h->d_adam_v, 0, sizeof(double) * std::max<int64_t>(1, h->m_cycle), h->stream));
    }
    if (h->m > 0) {                                                       // S_vec = ones(1,m)  (:148)
        int g = (int)std::min<int64_t>(1024, (h->m + 255) / 256);
        hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, h->stream, h->d_S[0], h->m, 1.0);
        hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, h->stream, h->d_S[1], h->m, 1.0);
    }
    if (h->m_pos > 0) {
        int g = (int)std::min<int64_t>(4096, (h->m_pos + 3) / 4);
        hipLaunchKernelGGL(k_init, dim3(g), dim3(256), 0, h->stream, h->d_cum, h->d_pos_edge, h->d_S0, h->d_w[0],
                           h->d_S[0], h->d_S[1], (int)h->m_pos);
    }
    DESC_HIP(hipGetLastError());
    h->armed = true;
    return DESC_OK;
}

int desc_pgd_iterate(desc_pgd* h, int32_t n_iters) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    if (!h->armed) return fail(DESC_ERR_STATE, "desc_pgd_reset must be called first");
    if (n_iters < 0 || h->t_done + n_iters > h->trace_cap) return fail(DESC_ERR_INVALID, "iterating past params.iters = %d", h->p.iters);
    int rc = set_device(h); if (rc) return rc;
    if (h->m_pos == 0) { h->t_done += n_iters; h->t_plugin += n_iters; return DESC_OK; }
    for (int q = 0; q < n_iters; ++q) { rc = enqueue_sweep(h, ++h->t_done); if (rc) return rc; }
    return DESC_OK;
}

int desc_pgd_iterate_timed(desc_pgd* h, int32_t n_iters, float* ms_total, float* ms_main_kernel_avg) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    if (!h->armed) return fail(DESC_ERR_STATE, "desc_pgd_reset must be called first");
    if (n_iters < 0 || h->t_done + n_iters > h->trace_cap) return fail(DESC_ERR_INVALID, "iterating past params.iters = %d", h->p.iters);
    int rc = set_device(h); if (rc) return rc;
    hipEvent_t e0, e1;
    DESC_HIP(hipEventCreate(&e0)); DESC_HIP(hipEventCreate(&e1));
    std::vector<hipEvent_t> ev;
    if (ms_main_kernel_avg) { ev.resize(2 * (size_t)n_iters); for (auto& e : ev) DESC_HIP(hipEventCreate(&e)); }
    DESC_HIP(hipEventRecord(e0, h->stream));
    if (h->m_pos > 0)
        for (int q = 0; q < n_iters; ++q) {
            rc = enqueue_sweep(h, ++h->t_done, ms_main_kernel_avg ? ev[2 * q] : nullptr, ms_main_kernel_avg ? ev[2 * q + 1] : nullptr);
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

int desc_pgd_sync(desc_pgd* h) {
    if (!h) return fail(DESC_ERR_INVALID, "NULL handle");
    int rc = set_device(h); if (rc) return rc;
    DESC_HIP(hipStreamSynchronize(h->stream));
    DESC_HIP(hipGetLastError());
    return DESC_OK;
}

int desc_pgd_download(desc_pgd* h, desc_result* r) {
    if (!h || !r) return fail(DESC_ERR_INVALID, "NULL argument");
    if (!h->armed) return fail(DESC_ERR_STATE, "nothing to download: call desc_pgd_reset first");
    if (!r->s_vec && h->m > 0) return fail(DESC_ERR_INVALID, "result.s_vec is NULL");
    int rc = set_device(h); if (rc) return rc;
    const int T = h->t_done;
    // objective of the last sweep (DESC_PGD.m:233) and its stop test
    if (h->m_pos > 0 && T >= 1) {
        hipLaunchKernelGGL(k_objective, dim3(h->obj_grid), dim3(256), 0, h->stream, h->d_w[T & 1], h->d_S[T & 1], h->d_ejk,
                           h->d_eki, h->m_cycle, h->d_partials, h->d_state);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, h->stream, h->d_partials, h->obj_grid, h->d_state, h->d_obj,
                           h->d_avg, T, h->m, h->p.patience, h->p.stop_tol, 1);
    }
    DevState st{};
    DESC_HIP(hipMemcpyAsync(&st, h->d_state, sizeof st, hipMemcpyDeviceToHost, h->stream));
    DESC_HIP(hipStreamSynchronize(h->stream));
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
    if (h->m > 0) DESC_HIP(hipMemcpy(r->s_vec, h->d_S[par], sizeof(double) * h->m, hipMemcpyDeviceToHost));
    if (r->w && h->m_cycle > 0) DESC_HIP(hipMemcpy(r->w, h->d_w[par], sizeof(double) * h->m_cycle, hipMemcpyDeviceToHost));
    if (r->obj_trace && iters_run > 0) {
        if (h->m_pos > 0) DESC_HIP(hipMemcpy(r->obj_trace, h->d_obj, sizeof(double) * iters_run, hipMemcpyDeviceToHost));
        else std::memset(r->obj_trace, 0, sizeof(double) * iters_run);
    }
    if (r->avg_change_trace && iters_run > 0) {
        if (h->m_pos > 0) DESC_HIP(hipMemcpy(r->avg_change_trace, h->d_avg, sizeof(double) * iters_run, hipMemcpyDeviceToHost));
        else std::memset(r->avg_change_trace, 0, sizeof(double) * iters_run);
    }
    if (h->d_adam_m && r->adam_m && r->adam_v && h->m_cycle > 0) {
        DESC_HIP(hipMemcpy(r->adam_m, h->d_adam_m, sizeof(double) * h->m_cycle, hipMemcpyDeviceToHost));
        DESC_HIP(hipMemcpy(r->adam_v, h->d_adam_v, sizeof(double) * h->m_cycle, hipMemcpyDeviceToHost));
    }
    r->ms_upload = h->ms_upload; r->ms_cycle_d = h->ms_cycle_d; r->ms_pgd = h->ms_pgd;
    return DESC_OK;
}

int desc_pgd_run(desc_pgd* h, const desc_params* p, desc_result* r) {
    if (!h || !p || !r) return fail(DESC_ERR_INVALID, "NULL argument");
    auto t0 = std::chrono::steady_clock::now();
    int rc = desc_pgd_reset(h, p); if (rc) return rc;
    if (p->step_kind == DESC_STEP_HYBRID && p->hybrid_strategy == 0 && p->t0 > 0 && r->adam_m && r->adam_v && h->m_cycle > 0) {
        // HybridGradient keeps m_t / v_t between calls (handle object)
        DESC_HIP(hipMemcpyAsync(h->d_adam_m, r->adam_m, sizeof(double) * h->m_cycle, hipMemcpyHostToDevice, h->stream));
        DESC_HIP(hipMemcpyAsync(h->d_adam_v, r->adam_v, sizeof(double) * h->m_cycle, hipMemcpyHostToDevice, h->stream));
    }
    const int chunk = p->check_every > 0 ? p->check_every : 32;
    int left = p->iters;
    while (left > 0) {
        const int nq = std::min(left, chunk);
        float ms = 0;
        rc = desc_pgd_iterate_timed(h, nq, &ms, nullptr); if (rc) return rc;
        left -= nq;
        if (left > 0 && h->m_pos > 0) {
            DevState st{};
            DESC_HIP(hipMemcpy(&st, h->d_state, sizeof st, hipMemcpyDeviceToHost));
            if (st.stop) break;
        }
    }
    rc = desc_pgd_download(h, r); if (rc) return rc;
    if (p->verbose && r->obj_trace && r->avg_change_trace)
        for (int it = 1; it <= r->iters_run; ++it)                       // DESC_PGD.m:241
            printf("iter %d: average change in S_vec %f, objective value: %f\n", it, r->avg_change_trace[it - 1], r->obj_trace[it - 1]);
    r->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
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
