/* ORACLE -- test infrastructure only.  Never linked, loaded or called by the
 * product path (desc_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / timed CPU baseline.
 *
 * Sparse CPU restatement of the reference's Algorithms/DESC_PGD.m (same text
 * inlined at Algorithms/DESC.m:16-261), split the way SURVEY.md section 8
 * splits it:
 *   oracle_build_structure  DESC_PGD.m:19-54, 79-127  (graph, codegree, sampling
 *                           budget, cycle lists, mirror-cycle maps) -- sparse:
 *                           no n x n / n x m_pos dense arrays
 *   oracle_cycle_d          DESC_PGD.m:129-147        (cycle inconsistency S0_long)
 *   oracle_pgd_run          DESC_PGD.m:148-261        (init + PGD loop, step-size
 *                           plugins Utils/ConstantStepSize.m:9-11,
 *                           PiecewiseStepSize.m:13-18, HybridGradient.m:23-41)
 *
 * PARITY UNPINNED: the reference has no tests / fixtures / golden vectors and
 * cannot be executed here (MATLAB only).  This file is pinned by hand-derived
 * known-answer cases and by agreement with the independent dense literal
 * restatement oracle/desc_pgd_literal.py (tests/test_oracle.py).
 *
 * All indices crossing this interface are 0-based.  rij is m x 9, block l holds
 * RijMat(:,:,l) in MATLAB (column-major) order: element (r,c) at rij[9*l+r+3*c].
 * `datasample` (DESC_PGD.m:84) is replaced by a deterministic keyed selection
 * (the n_sample smallest values of sample_key(seed, edge, k)); the reference's
 * own choice depends on MATLAB's global RNG stream and is not reproducible.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- deterministic sampling key (shared definition with the product; the
 *      product re-implements it independently in desc_amd/csrc) ------------- */
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
uint64_t oracle_sample_key(uint64_t seed, uint64_t edge, uint64_t k) {
    uint64_t a = mix64(seed ^ ((edge + 1) * 0x9E3779B97F4A7C15ull));
    return mix64(a ^ ((k + 1) * 0xD1B54A32D192ED03ull));
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* MATLAB abs(acos(x)) incl. complex extension (DESC_PGD.m:147) */
static double abs_acos_ext(double x) {
    if (x > 1.0) return acosh(x);
    if (x < -1.0) return hypot(M_PI, acosh(-x));
    return acos(x);
}

static int cmp_i64(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}
static int cmp_dbl(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}
typedef struct { uint64_t key; int32_t k; } keyed_t;
static int cmp_keyed(const void* a, const void* b) {
    const keyed_t* x = (const keyed_t*)a; const keyed_t* y = (const keyed_t*)b;
    if (x->key != y->key) return (x->key > y->key) - (x->key < y->key);
    return (x->k > y->k) - (x->k < y->k);
}

/* ------------------------------------------------------------------------
 * Structure (DESC_PGD.m:19-54, 79-127), sparse.
 * Two-call protocol: pass NULL output arrays to get the sizes first.
 * ---------------------------------------------------------------------- */
typedef struct {
    int64_t n, m, m_pos, m_cycle;
    int32_t n_sample;
} oracle_sizes;

/* CSR adjacency of the undirected graph, neighbours ascending, with the edge id
 * of each (node, neighbour) pair -- the sparse stand-in for AdjMat / IndMat
 * (DESC_PGD.m:23-24, 67-68). */
static void build_csr(int64_t n, int64_t m, const int32_t* ii, const int32_t* jj,
                      int64_t** rowptr_o, int32_t** col_o, int32_t** eid_o) {
    int64_t* rowptr = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
    for (int64_t e = 0; e < m; ++e) { rowptr[ii[e] + 1]++; rowptr[jj[e] + 1]++; }
    for (int64_t v = 0; v < n; ++v) rowptr[v + 1] += rowptr[v];
    int32_t* col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * m > 0 ? 2 * m : 1));
    int32_t* eid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * m > 0 ? 2 * m : 1));
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    memcpy(fill, rowptr, sizeof(int64_t) * (size_t)n);
    /* edges are sorted by (i, j): inserting in edge order keeps both the
     * "j > v" and the "i < v" parts ascending, but the i<v part of a row is
     * filled while scanning earlier rows, so it lands before the j>v part. */
    for (int64_t e = 0; e < m; ++e) {
        int32_t i = ii[e], j = jj[e];
        col[fill[i]] = j; eid[fill[i]++] = (int32_t)e;
        col[fill[j]] = i; eid[fill[j]++] = (int32_t)e;
    }
    /* do not rely on input order: sort each row by neighbour id */
    for (int64_t v = 0; v < n; ++v) {
        int64_t a = rowptr[v], b = rowptr[v + 1];
        for (int64_t x = a + 1; x < b; ++x) {          /* insertion sort, rows are nearly sorted */
            int32_t c = col[x], d = eid[x]; int64_t y = x - 1;
            while (y >= a && col[y] > c) { col[y + 1] = col[y]; eid[y + 1] = eid[y]; --y; }
            col[y + 1] = c; eid[y + 1] = d;
        }
    }
    free(fill);
    *rowptr_o = rowptr; *col_o = col; *eid_o = eid;
}

/* list common neighbours of (i,j) ascending (find(AdjMat(:,i).*AdjMat(:,j)), :82);
 * returns count; if kk != NULL also writes k, edge{j,k}, edge{k,i}. */
static int64_t common_nbrs(const int64_t* rowptr, const int32_t* col, const int32_t* eid,
                           int32_t i, int32_t j, int32_t* kk, int32_t* ejk, int32_t* eki) {
    int64_t a = rowptr[i], ae = rowptr[i + 1], b = rowptr[j], be = rowptr[j + 1], c = 0;
    while (a < ae && b < be) {
        if (col[a] < col[b]) ++a;
        else if (col[a] > col[b]) ++b;
        else {
            if (kk) { kk[c] = col[a]; eki[c] = eid[a]; ejk[c] = eid[b]; }
            ++c; ++a; ++b;
        }
    }
    return c;
}

int oracle_build_structure(int64_t n, int64_t m, const int32_t* ii, const int32_t* jj,
                           int32_t n_sample_min, uint64_t seed,
                           oracle_sizes* sz,
                           int32_t* codeg /* m, may be NULL */,
                           int32_t* pos_edge /* m_pos */, int64_t* cum_ind /* m_pos+1 */,
                           int32_t* kk, int32_t* e_jk, int32_t* e_ki,
                           int32_t* ikj, int32_t* jki /* m_cycle each */) {
    int64_t* rowptr; int32_t* col; int32_t* eid;
    build_csr(n, m, ii, jj, &rowptr, &col, &eid);

    /* codegree per edge (:29-34) */
    int32_t* cd = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    int64_t m_pos = 0;
    #pragma omp parallel for schedule(dynamic, 256) reduction(+:m_pos)
    for (int64_t e = 0; e < m; ++e) {
        cd[e] = (int32_t)common_nbrs(rowptr, col, eid, ii[e], jj[e], NULL, NULL, NULL);
        if (cd[e] > 0) ++m_pos;
    }
    /* n_sample = max(ceil(median(CoDeg_vec_pos)/4), 30)  (:43); median([]) = NaN -> 30 */
    int32_t n_sample = n_sample_min;
    if (m_pos > 0) {
        int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * (size_t)m_pos);
        int64_t t = 0;
        for (int64_t e = 0; e < m; ++e) if (cd[e] > 0) tmp[t++] = cd[e];
        qsort(tmp, (size_t)m_pos, sizeof(int64_t), cmp_i64);
        double med = (m_pos & 1) ? (double)tmp[m_pos / 2]
                                 : 0.5 * ((double)tmp[m_pos / 2 - 1] + (double)tmp[m_pos / 2]);
        free(tmp);
        int32_t q = (int32_t)ceil(med / 4.0);
        if (q > n_sample) n_sample = q;
    }
    /* cum_ind (:45-51) */
    int64_t m_cycle = 0;
    for (int64_t e = 0; e < m; ++e) if (cd[e] > 0) m_cycle += cd[e] < n_sample ? cd[e] : n_sample;
    sz->n = n; sz->m = m; sz->m_pos = m_pos; sz->m_cycle = m_cycle; sz->n_sample = n_sample;
    if (codeg) memcpy(codeg, cd, sizeof(int32_t) * (size_t)m);
    if (!pos_edge) { free(cd); free(rowptr); free(col); free(eid); return 0; }

    int64_t* pos_of_edge = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));  /* CoDeg_pos_ind_long (:53-54), -1 = none */
    {
        int64_t l = 0; cum_ind[0] = 0;
        for (int64_t e = 0; e < m; ++e) {
            pos_of_edge[e] = -1;
            if (cd[e] > 0) {
                pos_edge[l] = (int32_t)e; pos_of_edge[e] = l;
                cum_ind[l + 1] = cum_ind[l] + (cd[e] < n_sample ? cd[e] : n_sample);
                ++l;
            }
        }
    }
    /* cycle lists (:79-96); kept ascending in k inside each segment */
    #pragma omp parallel
    {
        int32_t cap = 16;
        int32_t* tk = (int32_t*)malloc(sizeof(int32_t) * cap);
        int32_t* tjk = (int32_t*)malloc(sizeof(int32_t) * cap);
        int32_t* tki = (int32_t*)malloc(sizeof(int32_t) * cap);
        keyed_t* keys = (keyed_t*)malloc(sizeof(keyed_t) * cap);
        #pragma omp for schedule(dynamic, 256)
        for (int64_t l = 0; l < m_pos; ++l) {
            int32_t e = pos_edge[l], i = ii[e], j = jj[e];
            int32_t c = cd[e];
            if (c > cap) {
                cap = c * 2;
                tk = (int32_t*)realloc(tk, sizeof(int32_t) * cap);
                tjk = (int32_t*)realloc(tjk, sizeof(int32_t) * cap);
                tki = (int32_t*)realloc(tki, sizeof(int32_t) * cap);
                keys = (keyed_t*)realloc(keys, sizeof(keyed_t) * cap);
            }
            common_nbrs(rowptr, col, eid, i, j, tk, tjk, tki);
            int64_t lo = cum_ind[l];
            if (c >= n_sample) {                       /* :83  (note >=) */
                for (int32_t t = 0; t < c; ++t) { keys[t].key = oracle_sample_key(seed, (uint64_t)e, (uint64_t)tk[t]); keys[t].k = t; }
                qsort(keys, (size_t)c, sizeof(keyed_t), cmp_keyed);
                /* keep the n_sample smallest keys, then restore ascending k */
                int64_t* sel = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_sample);
                for (int32_t t = 0; t < n_sample; ++t) sel[t] = keys[t].k;
                qsort(sel, (size_t)n_sample, sizeof(int64_t), cmp_i64);
                for (int32_t t = 0; t < n_sample; ++t) {
                    kk[lo + t] = tk[sel[t]]; e_jk[lo + t] = tjk[sel[t]]; e_ki[lo + t] = tki[sel[t]];
                }
                free(sel);
            } else {
                for (int32_t t = 0; t < c; ++t) { kk[lo + t] = tk[t]; e_jk[lo + t] = tjk[t]; e_ki[lo + t] = tki[t]; }
            }
        }
        free(tk); free(tjk); free(tki); free(keys);
    }
    /* mirror maps (:103-127): IKJ(c) = global index of cycle (ik; j) if j was
     * sampled for edge {i,k}; JKI(c) = index of cycle (jk; i) likewise; -1 absent */
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t l = 0; l < m_pos; ++l) {
        int32_t e = pos_edge[l], i = ii[e], j = jj[e];
        for (int64_t c = cum_ind[l]; c < cum_ind[l + 1]; ++c) {
            int64_t IK = pos_of_edge[e_ki[c]], JK = pos_of_edge[e_jk[c]];
            ikj[c] = -1; jki[c] = -1;
            for (int64_t t = cum_ind[IK]; t < cum_ind[IK + 1]; ++t) if (kk[t] == j) { ikj[c] = (int32_t)t; break; }
            for (int64_t t = cum_ind[JK]; t < cum_ind[JK + 1]; ++t) if (kk[t] == i) { jki[c] = (int32_t)t; break; }
        }
    }
    free(pos_of_edge); free(cd); free(rowptr); free(col); free(eid);
    return 0;
}

/* ------------------------------------------------------------------------
 * Cycle inconsistency (DESC_PGD.m:129-147)
 * ---------------------------------------------------------------------- */
static inline void load_block(const double* rij, int32_t e, int transpose, double R[9]) {
    const double* b = rij + 9 * (int64_t)e;
    if (!transpose) { for (int t = 0; t < 9; ++t) R[t] = b[t]; }
    else { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r + 3 * c] = b[c + 3 * r]; }
}
void oracle_cycle_d(int64_t m_pos, const int32_t* ii, const int32_t* jj, const double* rij,
                    const int32_t* pos_edge, const int64_t* cum_ind,
                    const int32_t* kk, const int32_t* e_jk, const int32_t* e_ki,
                    double* S0_long) {
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t l = 0; l < m_pos; ++l) {
        int32_t e = pos_edge[l], i = ii[e], j = jj[e];
        double A[9]; load_block(rij, e, 0, A);                 /* Rij0Mat (:129) */
        for (int64_t c = cum_ind[l]; c < cum_ind[l + 1]; ++c) {
            int32_t k = kk[c];
            double B[9], C[9], P[9];
            load_block(rij, e_jk[c], !(j < k), B);             /* RijMat4d(:,:,j,k) (:65-66,89) */
            load_block(rij, e_ki[c], !(k < i), C);             /* RijMat4d(:,:,k,i) (:91)      */
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {       /* :137-139 */
                double s = 0.0; for (int t = 0; t < 3; ++t) s += A[r + 3 * t] * B[t + 3 * q];
                P[r + 3 * q] = s;
            }
            double tr = 0.0;                                               /* :141-146 */
            for (int r = 0; r < 3; ++r) { double s = 0.0; for (int t = 0; t < 3; ++t) s += P[r + 3 * t] * C[t + 3 * r]; tr += s; }
            S0_long[c] = abs_acos_ext((tr - 1.0) / 2.0) / M_PI;            /* :147 */
        }
    }
}

/* ------------------------------------------------------------------------
 * Init + PGD loop (DESC_PGD.m:148-261)
 * step_kind 0 = ConstantStepSize(lr); 1 = PiecewiseStepSize(lr, decay);
 *           2 = HybridGradient(lr, b1, b2, decay) with .strategy = hybrid_strategy
 * t0 = plugin counter value on entry (handle objects keep state between calls)
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t iters, step_kind;
    double lr, beta1, beta2, decay_interval;
    int32_t hybrid_strategy, t0;
    int32_t patience;      /* 30  (:180) */
    double stop_tol;       /* 1e-5 (:243) */
} oracle_params;

int oracle_pgd_run(int64_t m, int64_t m_pos, const int32_t* pos_edge, const int64_t* cum_ind,
                   const int32_t* e_jk, const int32_t* e_ki, const int32_t* ikj, const int32_t* jki,
                   const double* S0_long, const oracle_params* p,
                   double* S_vec /* m */, double* wijk /* m_cycle, out */,
                   double* obj_vals /* iters */, double* avg_changes /* iters */,
                   double* adam_m /* m_cycle or NULL */, double* adam_v /* m_cycle or NULL */) {
    int64_t m_cycle = cum_ind[m_pos];
    double* grad_long = (double*)malloc(sizeof(double) * (size_t)(m_cycle > 0 ? m_cycle : 1));
    double* S_last = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    double* S_old = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    double* w_old = (double*)malloc(sizeof(double) * (size_t)(m_cycle > 0 ? m_cycle : 1));
    int own_adam = 0;
    if (p->step_kind == 2 && !adam_m) {
        adam_m = (double*)calloc((size_t)(m_cycle > 0 ? m_cycle : 1), sizeof(double));
        adam_v = (double*)calloc((size_t)(m_cycle > 0 ? m_cycle : 1), sizeof(double));
        own_adam = 1;
    }
    int64_t max_cnt = 1;
    for (int64_t l = 0; l < m_pos; ++l) if (cum_ind[l + 1] - cum_ind[l] > max_cnt) max_cnt = cum_ind[l + 1] - cum_ind[l];

    for (int64_t e = 0; e < m; ++e) S_vec[e] = 1.0;                        /* :148 */
    #pragma omp parallel for schedule(static)
    for (int64_t l = 0; l < m_pos; ++l) {                                  /* :151-157 */
        int64_t lo = cum_ind[l], hi = cum_ind[l + 1];
        double sw = 0.0; for (int64_t c = lo; c < hi; ++c) sw += 1.0;
        double s = 0.0;
        for (int64_t c = lo; c < hi; ++c) { wijk[c] = 1.0 / sw; s += wijk[c] * S0_long[c]; }
        S_vec[pos_edge[l]] = s;
    }
    memcpy(S_last, S_vec, sizeof(double) * (size_t)m);                     /* :167 */

    int32_t t = p->t0, misses = 0, it = 0, iters_run = 0;
    for (it = 1; it <= p->iters; ++it) {                                   /* :182 */
        iters_run = it;
        memcpy(w_old, wijk, sizeof(double) * (size_t)m_cycle);
        memcpy(S_old, S_vec, sizeof(double) * (size_t)m);
        /* the plugin is called once per iteration on the whole vector (:207) */
        ++t;
        double step_size = p->lr;
        double bc1 = 1.0, bc2 = 1.0;
        if (p->step_kind == 1) step_size = p->lr / (trunc((double)t / p->decay_interval) + 1.0);
        if (p->step_kind == 2 && p->hybrid_strategy == 1) step_size = 100.0 * (p->lr / (trunc((double)t / p->decay_interval) + 1.0));
        if (p->step_kind == 2 && p->hybrid_strategy == 0) { bc1 = 1.0 - pow(p->beta1, (double)t); bc2 = 1.0 - pow(p->beta2, (double)t); }

        #pragma omp parallel
        {
            double* ws = (double*)malloc(sizeof(double) * (size_t)max_cnt);
            #pragma omp for schedule(static)
            for (int64_t l = 0; l < m_pos; ++l) {
                int64_t lo = cum_ind[l], hi = cum_ind[l + 1], cnt = hi - lo;
                /* mirror sums: scalar per edge, broadcast to masked positions only (:185-191) */
                double T1 = 0.0, T2 = 0.0;
                for (int64_t c = lo; c < hi; ++c) { if (ikj[c] >= 0) T1 += w_old[ikj[c]]; if (jki[c] >= 0) T2 += w_old[jki[c]]; }
                /* :193 */
                for (int64_t c = lo; c < hi; ++c) {
                    double sik = ikj[c] >= 0 ? T1 : 0.0, sjk = jki[c] >= 0 ? T2 : 0.0;
                    grad_long[c] = S_old[e_jk[c]] + S_old[e_ki[c]] + (sik + sjk) * S0_long[c];
                }
                /* :195-204 tangent projection, nv = ones/sqrt(cnt) */
                double nv = 1.0 / pow((double)cnt, 0.5), dot = 0.0;
                for (int64_t c = lo; c < hi; ++c) dot += grad_long[c] * nv;
                for (int64_t c = lo; c < hi; ++c) grad_long[c] = grad_long[c] - dot * nv;
                /* :207 GetStep */
                for (int64_t c = lo; c < hi; ++c) {
                    double g = grad_long[c], step;
                    if (p->step_kind == 2 && p->hybrid_strategy == 0) {
                        adam_m[c] = (p->beta1 * adam_m[c]) + (1.0 - p->beta1) * g;
                        adam_v[c] = (p->beta2 * adam_v[c]) + (1.0 - p->beta2) * (g * g);
                        double cm = adam_m[c] / bc1, cv = adam_v[c] / bc2;
                        step = -p->lr * cm / (sqrt(cv) + 1e-8);
                    } else {
                        step = -step_size * g;
                    }
                    wijk[c] = w_old[c] + step;
                }
                /* :208-224 simplex projection, literal sort-and-scan */
                for (int64_t c = 0; c < cnt; ++c) ws[c] = wijk[lo + c];
                qsort(ws, (size_t)cnt, sizeof(double), cmp_dbl);
                int64_t Ti = 0; double tail = 0.0;
                for (int64_t i1 = 0; i1 < cnt; ++i1) {
                    double s = 0.0; for (int64_t q = i1; q < cnt; ++q) s += ws[q] - ws[i1];
                    if (s < 1.0) { Ti = i1; tail = s; break; }
                }
                double T = ws[Ti] - (1.0 - tail) / (double)(cnt - Ti);
                double s = 0.0;
                for (int64_t c = lo; c < hi; ++c) { double v = wijk[c] - T; wijk[c] = v > 0.0 ? v : 0.0; s += wijk[c] * S0_long[c]; }
                S_vec[pos_edge[l]] = s;                                     /* :229 */
            }
            free(ws);
        }
        /* :232-233 */
        double ac = 0.0, obj = 0.0;
        #pragma omp parallel for schedule(static) reduction(+:ac)
        for (int64_t e = 0; e < m; ++e) ac += fabs(S_vec[e] - S_last[e]);
        ac /= (double)m;
        #pragma omp parallel for schedule(static) reduction(+:obj)
        for (int64_t c = 0; c < m_cycle; ++c) obj += wijk[c] * (S_vec[e_jk[c]] + S_vec[e_ki[c]]);
        obj_vals[it - 1] = obj; avg_changes[it - 1] = ac;
        if (it > 1 && obj_vals[it - 2] - obj_vals[it - 1] < p->stop_tol) {  /* :243 */
            ++misses;
            if (misses >= p->patience) break;                               /* :245-246 */
        } else misses = 0;                                                  /* :255 */
        memcpy(S_last, S_vec, sizeof(double) * (size_t)m);                  /* :257 */
    }
    free(grad_long); free(S_last); free(S_old); free(w_old);
    if (own_adam) { free(adam_m); free(adam_v); }
    return iters_run;
}


/* ------------------------------------------------------------------------
 * The same loop (DESC_PGD.m:148-261, ConstantStepSize / PiecewiseStepSize only) carried in long double (x87: 64-bit
 * significand, eps 5.4e-20) from the first iteration to the last: a yardstick for how much of a difference between two
 * double-precision implementations is round-off amplified by the iteration (tests: the lr = 1 fuzz case of round 2).
 * Inputs and outputs are double; nothing is rounded to double in between.  Serial.
 * ---------------------------------------------------------------------- */
static int cmp_ld(const void* a, const void* b) {
    long double x = *(const long double*)a, y = *(const long double*)b;
    return (x > y) - (x < y);
}
int oracle_pgd_run_ld(int64_t m, int64_t m_pos, const int32_t* pos_edge, const int64_t* cum_ind,
                      const int32_t* e_jk, const int32_t* e_ki, const int32_t* ikj, const int32_t* jki,
                      const double* S0_long, const oracle_params* p,
                      double* S_vec_out /* m */, double* wijk_out /* m_cycle */, double* obj_vals /* iters */) {
    typedef long double real;
    if (p->step_kind == 2) return -1;
    int64_t m_cycle = cum_ind[m_pos];
    size_t nc = (size_t)(m_cycle > 0 ? m_cycle : 1), ne = (size_t)(m > 0 ? m : 1);
    real* w = (real*)malloc(sizeof(real) * nc);
    real* w_old = (real*)malloc(sizeof(real) * nc);
    real* g = (real*)malloc(sizeof(real) * nc);
    real* S = (real*)malloc(sizeof(real) * ne);
    real* S_old = (real*)malloc(sizeof(real) * ne);
    int64_t max_cnt = 1;
    for (int64_t l = 0; l < m_pos; ++l) if (cum_ind[l + 1] - cum_ind[l] > max_cnt) max_cnt = cum_ind[l + 1] - cum_ind[l];
    real* ws = (real*)malloc(sizeof(real) * (size_t)max_cnt);
    for (int64_t e = 0; e < m; ++e) S[e] = 1.0L;
    for (int64_t l = 0; l < m_pos; ++l) {
        int64_t lo = cum_ind[l], hi = cum_ind[l + 1];
        real s = 0.0L;
        for (int64_t c = lo; c < hi; ++c) { w[c] = 1.0L / (real)(hi - lo); s += w[c] * (real)S0_long[c]; }
        S[pos_edge[l]] = s;
    }
    int32_t t = p->t0, misses = 0, it = 0, iters_run = 0;
    for (it = 1; it <= p->iters; ++it) {
        iters_run = it;
        memcpy(w_old, w, sizeof(real) * (size_t)m_cycle);
        memcpy(S_old, S, sizeof(real) * (size_t)m);
        ++t;
        real step_size = (real)p->lr;
        if (p->step_kind == 1) step_size = (real)p->lr / (truncl((real)t / (real)p->decay_interval) + 1.0L);
        for (int64_t l = 0; l < m_pos; ++l) {
            int64_t lo = cum_ind[l], hi = cum_ind[l + 1], cnt = hi - lo;
            real T1 = 0.0L, T2 = 0.0L;
            for (int64_t c = lo; c < hi; ++c) { if (ikj[c] >= 0) T1 += w_old[ikj[c]]; if (jki[c] >= 0) T2 += w_old[jki[c]]; }
            for (int64_t c = lo; c < hi; ++c)
                g[c] = S_old[e_jk[c]] + S_old[e_ki[c]] + ((ikj[c] >= 0 ? T1 : 0.0L) + (jki[c] >= 0 ? T2 : 0.0L)) * (real)S0_long[c];
            real nv = 1.0L / sqrtl((real)cnt), dot = 0.0L;
            for (int64_t c = lo; c < hi; ++c) dot += g[c] * nv;
            for (int64_t c = lo; c < hi; ++c) w[c] = w_old[c] - step_size * (g[c] - dot * nv);
            for (int64_t c = 0; c < cnt; ++c) ws[c] = w[lo + c];
            qsort(ws, (size_t)cnt, sizeof(real), cmp_ld);
            int64_t Ti = 0; real tail = 0.0L;
            for (int64_t i1 = 0; i1 < cnt; ++i1) {
                real s = 0.0L; for (int64_t q = i1; q < cnt; ++q) s += ws[q] - ws[i1];
                if (s < 1.0L) { Ti = i1; tail = s; break; }
            }
            real T = ws[Ti] - (1.0L - tail) / (real)(cnt - Ti);
            real s = 0.0L;
            for (int64_t c = lo; c < hi; ++c) { real v = w[c] - T; w[c] = v > 0.0L ? v : 0.0L; s += w[c] * (real)S0_long[c]; }
            S[pos_edge[l]] = s;
        }
        real obj = 0.0L;
        for (int64_t c = 0; c < m_cycle; ++c) obj += w[c] * (S[e_jk[c]] + S[e_ki[c]]);
        obj_vals[it - 1] = (double)obj;
        if (it > 1 && obj_vals[it - 2] - obj_vals[it - 1] < p->stop_tol) { if (++misses >= p->patience) break; } else misses = 0;
    }
    for (int64_t e = 0; e < m; ++e) S_vec_out[e] = (double)S[e];
    for (int64_t c = 0; c < m_cycle; ++c) wijk_out[c] = (double)w[c];
    free(w); free(w_old); free(g); free(S); free(S_old); free(ws);
    return iters_run;
}
