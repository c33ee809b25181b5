/* desc_amd.h -- C ABI of libdesc_amd.so (MI355X / gfx950 HIP implementation of the
 * DESC projected-gradient hot path).
 *
 * The reference (ColeWyeth/DESC) is pure MATLAB and has no FFI of its own; the
 * boundary this library replaces is the MATLAB function signature
 *     S_vec = DESC_PGD(Ind, RijMat, params)            Algorithms/DESC_PGD.m:14
 * (the same body is inlined in Algorithms/DESC.m:16-261, called from
 * Demo/compare_algorithms.m:72).  A MEX shim (matlab/desc_pgd_mex.c) or any other
 * FFI (ctypes: desc_amd/_lib.py) binds exactly the entry points below.  Plain
 * pointers and sizes only; no exceptions cross the boundary; every function
 * returns DESC_OK (0) or a negative error code, with text from desc_last_error().
 *
 * Conventions
 *   - all indices are 0-based int32 (the MATLAB side subtracts 1);
 *   - edges are rows (ind_i[l], ind_j[l]) with ind_i < ind_j, strictly sorted by
 *     (ind_i, ind_j) -- the order DESC_PGD.m:5,31-34 silently relies on;
 *   - rij is m x 9 doubles, block l = RijMat(:,:,l) in MATLAB column-major order
 *     (element (r,c) at rij[9*l + r + 3*c]), i.e. mxGetDoubles(RijMat) unchanged;
 *   - all arithmetic is IEEE double (the reference has no single precision).
 *   - the caller owns every host buffer it passes; the library owns device memory.
 *   - threads: any entry point may be called from any host thread, and DISTINCT objects (structures, device problems,
 *     solver handles, one-shot solves) may be in use on distinct threads at the same time -- a MATLAB worker pool or a
 *     serving process; the block / stream pools are locked, desc_last_error() is per thread.  One object must not be
 *     used from two threads at once.  (tests/test_gpu_parity.py::test_concurrent_solves_from_several_host_threads)
 */
#ifndef DESC_AMD_H
#define DESC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DESC_OK              0
#define DESC_ERR_INVALID    -1   /* bad argument / unsorted Ind / out-of-range index     */
#define DESC_ERR_HIP        -2   /* HIP runtime error (no device, OOM, launch failure)   */
#define DESC_ERR_TOO_LARGE  -3   /* m_cycle or m exceeds the 2^31-1 / 2^30 index budget  */
#define DESC_ERR_STATE      -4   /* call order violated (e.g. iterate before reset)      */

/* step-size plugin kinds: params.Gradient of the reference (DESC_PGD.m:207) */
#define DESC_STEP_CONSTANT   0   /* Utils/ConstantStepSize.m:9-11   step = -lr*g                       */
#define DESC_STEP_PIECEWISE  1   /* Utils/PiecewiseStepSize.m:13-18 step = -lr/(fix(t/decay)+1)*g      */
#define DESC_STEP_HYBRID     2   /* Utils/HybridGradient.m:23-41    Adam, or 100*lr/(fix(t/decay)+1)   */

/* where a-1..a-3 (graph, codegree, sampling, mirror maps) are built */
#define DESC_BUILD_HOST      0
#define DESC_BUILD_DEVICE    1

const char* desc_last_error(void);
const char* desc_version(void);
/* number of visible HIP devices, or a negative error code */
int desc_device_count(void);

/* ---------------------------------------------------------------- problem -- */
typedef struct desc_problem {
    int64_t n;              /* number of nodes = max(Ind(:))            DESC_PGD.m:21 */
    int64_t m;              /* number of edges                          DESC_PGD.m:22 */
    const int32_t* ind_i;   /* m, 0-based, ind_i[l] < ind_j[l]          DESC_PGD.m:19 */
    const int32_t* ind_j;   /* m                                        DESC_PGD.m:20 */
    const double* rij;      /* m*9, may be NULL for structure-only calls  DESC_PGD.m:7 */
} desc_problem;

/* A problem resident in the HBM of `device`: edge list, rotations and the CSR index of the graph.  DESC()'s three stages
 * (DESC_PGD -> GCW -> refinement, Algorithms/DESC.m:16-313) and the stand-alone Spectral / CEMP all read the same Ind /
 * RijMat; uploading them once removes two of three 72-B-per-edge host->device copies and CSR passes of a DESC() call.
 * The *_dev entry points below take it in place of a desc_problem.  The caller may free its host arrays afterwards. */
typedef struct desc_device_problem desc_device_problem;   /* opaque, library-owned */
int desc_problem_upload(const desc_problem* prob, int32_t device, desc_device_problem** out);
void desc_problem_free(desc_device_problem* dp);

/* -------------------------------------------------------------- structure -- */
/* The sampled 3-cycle structure of DESC_PGD.m:29-127, sparse.  Cycle c of the
 * l-th edge-with-cycles (edge id pos_edge[l] = (i,j)) lives at
 * cum_ind[l] <= c < cum_ind[l+1] and has third vertex k[c]; inside a segment the
 * k are ascending. */
typedef struct desc_structure desc_structure;   /* opaque, library-owned */

typedef struct desc_structure_view {
    int64_t n, m;
    int64_t m_pos;            /* edges with >=1 triangle                       DESC_PGD.m:50 */
    int64_t m_cycle;          /* sampled cycles in total                       DESC_PGD.m:51 */
    int32_t n_sample;         /* max(ceil(median(codeg>0)/4), n_sample_min)    DESC_PGD.m:43 */
    int32_t max_cnt;          /* longest segment                                             */
    const int32_t* codeg;     /* m      codegree of every edge (0 = no triangle) DESC_PGD.m:29-40 */
    const int32_t* pos_edge;  /* m_pos  CoDeg_pos_ind (0-based edge ids)       DESC_PGD.m:36 */
    const int64_t* cum_ind;   /* m_pos+1                                       DESC_PGD.m:49 */
    const int32_t* k;         /* m_cycle  IJK                                  DESC_PGD.m:93 */
    const int32_t* e_jk;      /* m_cycle  Ind_jk: edge id of {j,k}             DESC_PGD.m:87 */
    const int32_t* e_ki;      /* m_cycle  Ind_ki: edge id of {k,i}             DESC_PGD.m:88 */
    const int32_t* ikj;       /* m_cycle  IKJ: cycle (ik;j), -1 if not sampled DESC_PGD.m:116 */
    const int32_t* jki;       /* m_cycle  JKI: cycle (jk;i), -1 if not sampled DESC_PGD.m:125 */
} desc_structure_view;

/* Build the structure from the edge list.  `datasample` (DESC_PGD.m:84, MATLAB
 * global RNG) is replaced by a counter-based keyed selection: an edge with
 * codeg >= n_sample keeps the n_sample common neighbours k with the smallest
 * desc_sample_key(seed, edge, k).  where = DESC_BUILD_HOST | DESC_BUILD_DEVICE.
 * DESC_BUILD_DEVICE keeps the structure in the HBM of `device` in a lean form (sampled
 * third vertices + per-edge selection thresholds + adjacency); a solver created on the
 * same device lays it out in place without a host round trip.  DESC_ERR_TOO_LARGE means
 * the graph exceeds the device builder's staging budget: use DESC_BUILD_HOST. */
int desc_structure_build(const desc_problem* prob, int32_t n_sample_min, uint64_t seed,
                         int32_t where, int32_t device, desc_structure** out);
/* Adopt a caller-supplied structure (oracle-parity runs; arrays are copied). */
int desc_structure_import(int64_t n, int64_t m, int64_t m_pos, int32_t n_sample,
                          const int32_t* pos_edge, const int64_t* cum_ind,
                          const int32_t* k, const int32_t* e_jk, const int32_t* e_ki,
                          const int32_t* ikj, const int32_t* jki, desc_structure** out);
/* Host view of the full index structure.  For a device-built structure the first call derives
 * e_jk, e_ki and the mirror maps on the device and copies them down (O(m_cycle)); later calls
 * are free.  Not thread-safe per structure; the view stays valid until desc_structure_free.
 * Callers that only need the sizes use desc_structure_sizes, which never touches the device. */
int desc_structure_get(const desc_structure* s, desc_structure_view* view);
/* Sizes and build facts of a structure (DESC_PGD.m:43,50,51) without materialising anything: O(1), no
 * device work, no host copy -- what a binding needs to size the per-cycle in/out vectors
 * (HybridGradient.m_t / v_t) before desc_pgd_create. */
typedef struct desc_structure_info {
    int64_t n, m;
    int64_t m_pos;            /* edges with >=1 triangle                       DESC_PGD.m:50 */
    int64_t m_cycle;          /* sampled cycles in total                       DESC_PGD.m:51 */
    int32_t n_sample;         /* DESC_PGD.m:43 */
    int32_t max_cnt;          /* longest segment */
    int32_t built_where;      /* DESC_BUILD_HOST | DESC_BUILD_DEVICE (imported structures: HOST) */
    int32_t host_resident;    /* 1 when the per-cycle index arrays exist in host memory */
    double  ms_build;         /* wall clock of desc_structure_build, milliseconds */
} desc_structure_info;
int desc_structure_sizes(const desc_structure* s, desc_structure_info* info);
/* Number of times, in this process, a device-built structure was exported to host memory
 * (k_cycle_edges + k_mirror + five O(m_cycle) copies).  Diagnostics: the solver path must not need it. */
int64_t desc_structure_host_exports(void);
void desc_structure_free(desc_structure* s);
uint64_t desc_sample_key(uint64_t seed, uint64_t edge, uint64_t k);

/* ----------------------------------------------------------------- solver -- */
typedef struct desc_params {
    int32_t iters;            /* params.iters                                  DESC_PGD.m:170 */
    int32_t step_kind;        /* DESC_STEP_*                                   DESC_PGD.m:207 */
    double  lr;               /* learning_rate / lr property of the plugin                    */
    double  beta1, beta2;     /* HybridGradient.beta_1/beta_2                                 */
    double  decay_interval;   /* Piecewise/Hybrid decay_interval                              */
    int32_t hybrid_strategy;  /* HybridGradient.strategy: 0 Adam, 1 decayed plain step        */
    int32_t t0;               /* plugin counter t on entry (handle objects keep state)        */
    int32_t patience;         /* 30                                            DESC_PGD.m:180 */
    double  stop_tol;         /* 1e-5                                          DESC_PGD.m:243 */
    int32_t n_sample_min;     /* 30                                            DESC_PGD.m:43  */
    uint64_t seed;            /* sampling seed                                                */
    int32_t verbose;          /* print the reference's progress lines          DESC_PGD.m:241 */
    int32_t device;           /* HIP device ordinal                                           */
    int32_t build_where;      /* DESC_BUILD_*  (one-shot desc_pgd_solve only)                 */
    int32_t check_every;      /* host polls the device stop flag every this many iterations
                                 (0 = library default); results do not depend on it           */
    /* progress lines of DESC_PGD.m:241, streamed while the loop runs (every check_every iterations, 10 by default when
     * a callback or verbose is set): called once per finished iteration, in order, from the thread inside desc_pgd_run /
     * desc_pgd_solve.  NULL with verbose != 0: the reference's fprintf line on stdout.  A MEX shim passes a function that
     * calls mexPrintf. */
    void (*progress)(void* user, int32_t iter, double average_change, double objective);
    void* progress_user;
} desc_params;
void desc_params_default(desc_params* p);

typedef struct desc_result {
    double* s_vec;            /* m        out, caller-allocated: S_vec         DESC_PGD.m:229 */
    double* obj_trace;        /* iters    out or NULL: obj_vals                DESC_PGD.m:233 */
    double* avg_change_trace; /* iters    out or NULL: average_change          DESC_PGD.m:232 */
    double* w;                /* m_cycle  out or NULL: wijk                    DESC_PGD.m:224 */
    double* adam_m;           /* m_cycle  in/out or NULL: HybridGradient.m_t                  */
    double* adam_v;           /* m_cycle  in/out or NULL: HybridGradient.v_t                  */
    int32_t iters_run;        /* iterations executed (early stop, DESC_PGD.m:243-246)         */
    int32_t t_end;            /* plugin counter after the run                                 */
    /* timings, milliseconds */
    double ms_structure;      /* a-1..a-3 build                                               */
    double ms_upload;         /* host -> HBM                                                  */
    double ms_cycle_d;        /* a-4 kernel                                                   */
    double ms_pgd;            /* PGD loop, device time (HIP events)                           */
    double ms_total;          /* wall clock of the call                                       */
} desc_result;

typedef struct desc_pgd desc_pgd;   /* opaque solver handle: one per (problem, device) */

/* Upload problem + structure to HBM and evaluate the cycle inconsistencies
 * S0_long (DESC_PGD.m:129-147).  The structure may be freed afterwards. */
int desc_pgd_create(const desc_problem* prob, const desc_structure* s, int32_t device,
                    desc_pgd** out);
/* the same for a problem already resident in HBM (same device); rank 0 of world 1 = the whole problem */
int desc_pgd_create_dev(const desc_device_problem* dp, const desc_structure* s, int32_t rank, int32_t world, desc_pgd** out);
void desc_pgd_destroy(desc_pgd* h);
/* Full run: init (DESC_PGD.m:148-167) + loop (:182-261) + download. */
int desc_pgd_run(desc_pgd* h, const desc_params* p, desc_result* r);
/* params.make_plots = true (DESC_PGD.m:235-239; the figure of DESC.m:315-344 is drawn from these traces): desc_pgd_run with, after
 * every iteration t, svec_errors[t-1] = mean|err_vec - S_vec| (:236, err_vec = params.ErrVec, m doubles) and the rotation estimate
 * GCW(S_vec) (:237) in R_est_all[(t-1) * 9n ...] (caller-allocated: iters entries / iters * 9n doubles; 3 x 3 x n column-major per
 * iteration).  MSE_means / MSE_medians (:238) = the caller's GlobalSOdCorrectRight(R_est, params.R_orig) of each estimate.
 * dp: the same problem resident on the handle's device.  r->s_vec, obj_trace, avg_change_trace are required.  Entries past
 * r->iters_run are untouched.  r->adam_m / adam_v as in desc_pgd_run. */
int desc_pgd_run_traced(desc_pgd* h, const desc_device_problem* dp, const desc_params* p, const double* err_vec, double gcw_tol,
                        int32_t gcw_max_iters, double* svec_errors, double* R_est_all, desc_result* r);
/* Pieces of desc_pgd_run, for benchmarks and the multi-GPU driver.  All work is
 * enqueued on the handle's stream; _sync waits for it. */
int desc_pgd_reset(desc_pgd* h, const desc_params* p);            /* :148-167                 */
int desc_pgd_iterate(desc_pgd* h, int32_t n_iters);              /* enqueue n_iters sweeps   */
int desc_pgd_iterate_timed(desc_pgd* h, int32_t n_iters, float* ms_total, float* ms_main_kernel_avg);
int desc_pgd_sync(desc_pgd* h);
int desc_pgd_download(desc_pgd* h, desc_result* r);              /* finishes objective trace */
int desc_pgd_get_s0(desc_pgd* h, double* s0 /* m_cycle */);       /* S0_long                  */
int desc_pgd_sizes(const desc_pgd* h, int64_t* m, int64_t* m_pos, int64_t* m_cycle, int32_t* max_cnt);
/* what this handle's layout streams per iteration (bench.py's `floor_bytes`): out[0] = (cycle, endpoint) pairs the mirror-sum pass reads
 * (DESC_PGD.m:185-191: cycles whose mirror was sampled), out[1] = pieces of the band sweep, out[2] = bands, out[3] = CSR entries of band rows
 * staged in the LDS per sweep, out[4] / out[5] = cycles / segments this rank owns.  Returns how many values were written (<= cap). */
int desc_pgd_layout_stats(const desc_pgd* h, int64_t* out, int32_t cap);
/* name of the main-sweep kernel variant chosen for this handle (rocprof cross-reference) */
const char* desc_pgd_kernel_name(const desc_pgd* h);

/* ------------------------------------------------------------- multi-GPU -- */
/* One process per GPU.  The edges with cycles (in the library's band-major order) are cut
 * into `world` contiguous ranges of equal cycle count; rank r keeps the per-cycle state
 * (weights, inconsistencies, packed indices) of its range only, and a full replica of the
 * O(m) vectors.  One PGD iteration = desc_pgd_shard_colsum -> reduce-scatter(sum) of T_send
 * into T_recv -> desc_pgd_shard_sweep -> all-gather of sall -> desc_pgd_shard_finish.  The collectives are
 * the caller's (torch.distributed over RCCL in desc_amd/sharded.py); T and sall are device
 * buffers the caller allocates and binds.  All ranks take identical stop decisions because
 * every rank adds the gathered scalar partials in rank order. */
typedef struct desc_shard_info {
    int32_t rank, world;
    int64_t t_len;            /* 8-byte words in T_send = world*xparts*t_part + 1 (mirror sums as fixed-point integers, one block of t_part words per
                                 (exchange part, rank), part-major; last = unused slot) */
    int64_t t_part;           /* words per block: T1 | T2 of the edges of one (rank, part), padded to the largest; T_recv holds xparts blocks */
    int64_t slice_len;        /* doubles per rank in sall: S of the owned edges (padded to the largest shard), then the
                                 workgroup partials (objective, sum |dS|) of the rank's last sweep                      */
    int64_t seg_lo, seg_hi;   /* owned range of edges-with-cycles (library order)             */
    int64_t cyc_lo, cyc_hi;   /* owned range of cycles                                        */
    int64_t m_pos, m_cycle;   /* global counts                                                */
    int32_t xparts;           /* exchange parts per rank (round 4): the reduce-scatter runs part by part -- for c in 0..xparts-1:
                                 reduce_scatter(T_send + c*world*t_part  ->  T_recv + c*t_part, t_part words) -- so that part c + 1 travels
                                 while part c is swept (the fused protocol does; the piecewise calls sweep all parts in desc_pgd_shard_sweep) */
    int32_t reserved;
} desc_shard_info;
int desc_pgd_create_shard(const desc_problem* prob, const desc_structure* s, int32_t device,
                          int32_t rank, int32_t world, desc_pgd** out);
int desc_pgd_shard_info(const desc_pgd* h, desc_shard_info* info);
/* T_send: t_len 8-byte words, zero-initialised by the caller (this rank's partial mirror sums, grouped by
 * owning (part, rank): block b = part*world + rank = [b*t_part, (b+1)*t_part)); the words are 64-bit fixed-point INTEGERS: the caller's
 * reduce-scatter must add them as int64 (ncclInt64 / torch.int64), not as doubles.  T_recv: xparts*t_part words (the caller's
 * reduce-scatter(sum) of every rank's T_send, part by part: see desc_shard_info.xparts); sall: world*slice_len doubles, zero-initialised; all on `device`.
 * All three NULL: the library allocates them itself (the fused protocol below needs no caller buffers).
 * hip_stream: the stream the caller's collectives are ordered on (NULL keeps the handle's own). */
int desc_pgd_shard_bind(desc_pgd* h, double* T_send, double* T_recv, double* sall, void* hip_stream);
/* Piecewise protocol (one call per step, the caller runs the collectives in between; kept for drivers that own
 * the communication, e.g. torch.distributed with a backend other than RCCL, and for single-GPU emulation tests). */
int desc_pgd_shard_colsum(desc_pgd* h);
int desc_pgd_shard_sweep(desc_pgd* h);
/* initial: 0 = after an iteration's all-gather; 1 = after reset, before the first all-gather (no-op: the reset
 * already put the initial S of the owned edges into the slice); 2 = unpack the initial S_vec (after that all-gather). */
int desc_pgd_shard_finish(desc_pgd* h, int32_t initial);
/* objective of the last iterate: phase 0 puts this rank's partials into its slice (then all-gather sall),
 * phase 1 adds the partials of all ranks and runs the stop rule for the last iteration. */
int desc_pgd_shard_objective(desc_pgd* h, int32_t phase);

/* Fused protocol: the library enqueues whole iterations itself -- column sums, reduce-scatter, sweep, all-gather,
 * unpack + stop rule -- on two streams, so that the all-gather of S_vec and its unpacking overlap the next
 * iteration's column-sum pass (which needs only the weights).  The collectives are the caller's: two function
 * pointers with the signatures of RCCL's ncclReduceScatter / ncclAllGather (hipStream_t stream; the reduce-scatter is
 * called with datatype 4 = ncclInt64 and op 0 = ncclSum -- the mirror sums of DESC_PGD.m:189-190 travel as 64-bit fixed-point
 * integers, whose sum does not depend on the order the ranks' parts are added in --, the all-gather with datatype 8 =
 * ncclDouble) and the communicator they take; a binding passes the addresses of the RCCL
 * symbols of the process (desc_amd/sharded.py: the librccl.so PyTorch ships, communicator created with
 * ncclCommInitRank from an id broadcast over torch.distributed) or its own trampolines.  world == 1: both may
 * be NULL.  Nothing in the reference corresponds to this (it is single-process MATLAB). */
typedef struct desc_collectives {
    void* comm;               /* ncclComm_t (opaque to the library) */
    int (*reduce_scatter)(const void* sendbuff, void* recvbuff, size_t recvcount, int datatype, int op, void* comm, void* stream);
    int (*all_gather)(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, void* stream);
} desc_collectives;
int desc_pgd_shard_set_collectives(desc_pgd* h, const desc_collectives* c);
int desc_pgd_shard_start(desc_pgd* h, const desc_params* p);          /* reset + initial exchange of S_vec        */
int desc_pgd_shard_iterate(desc_pgd* h, int32_t n_iters);             /* enqueue n_iters iterations               */
/* start + iterate (the stop flag is polled every p->check_every iterations; identical on all ranks) + objective of
 * the last iterate + download (S_vec, traces; per-cycle outputs are not gathered across ranks). */
int desc_pgd_shard_run(desc_pgd* h, const desc_params* p, desc_result* r);
/* *stopped = 1 once the device-side patience rule (DESC_PGD.m:243-246) has fired; waits for
 * the handle's stream. */
int desc_pgd_stopped(desc_pgd* h, int32_t* stopped);

/* ------------------------------------------------- Spectral / GCW (next row f-1) -- */
/* Top-3 eigenvectors ('la') of the 3n x 3n block connection matrix + per-node projection onto
 * SO(3).  weights == NULL, normalize_rows == 0: Algorithms/Spectral.m:18-46.
 * weights[l] = 1/(S_vec[l]^1.5 + 1e-8), normalize_rows == 1: Utils/GCW.m:9-36 (the matrix
 * D^-1*W .* Rij_blk; iterated in its symmetric similar form, eigenvectors mapped back and
 * re-normalised).  R_out: n*9 doubles, 3x3xn in MATLAB column-major order.  Rotations are
 * defined up to one global right rotation (compare after Utils/Rotation_Alignment.m). */
typedef struct desc_spectral_info {
    int32_t iters;            /* outer (filter + Rayleigh-Ritz) steps              */
    int32_t products;         /* block matrix products (3n x 6 each)               */
    int32_t converged;        /* residual <= tol reached                           */
    int32_t reserved;
    double  residual;         /* max relative residual of the three Ritz pairs     */
    double  eigenvalues[3];   /* the three largest eigenvalues                     */
    double  ms_total;
} desc_spectral_info;
int desc_spectral_run(const desc_problem* prob, const double* weights, int32_t normalize_rows, double tol,
                      int32_t max_iters, int32_t device, double* R_out, desc_spectral_info* info);
int desc_spectral_run_dev(const desc_device_problem* dp, const double* weights, int32_t normalize_rows, double tol,
                          int32_t max_iters, double* R_out, desc_spectral_info* info);
/* R_est = GCW(Ind, AdjMat, RijMat, SVec) -- Utils/GCW.m:9-36 -- with the weights 1/(SVec.^1.5 + 1e-8) (GCW.m:20) and the
 * weighted degrees formed on the device from s_vec (m doubles, host). */
int desc_gcw_run_dev(const desc_device_problem* dp, const double* s_vec, double tol, int32_t max_iters, double* R_out,
                     desc_spectral_info* info);

/* ------------------------------------------------------------ CEMP (next row f-2) -- */
/* SVec = CEMP(Ind, RijMat, CEMP_parameters) -- Algorithms/CEMP.m:24-132.  beta[0..n_beta-1] =
 * CEMP_parameters.reweighting (missing entries repeat the last one, CEMP.m:30-34), max_iter =
 * .max_iter, nsample = .nsample (cycles per edge, sampled with replacement, CEMP.m:64; MATLAB's
 * RNG is replaced by CoInd[desc_sample_key(seed, edge, t) mod codeg]).  s_vec: m doubles out. */
int desc_cemp_run(const desc_problem* prob, const double* beta, int32_t n_beta, int32_t max_iter, int32_t nsample,
                  uint64_t seed, int32_t device, double* s_vec, double* ms_total);
int desc_cemp_run_dev(const desc_device_problem* dp, const double* beta, int32_t n_beta, int32_t max_iter, int32_t nsample,
                      uint64_t seed, double* s_vec, double* ms_total);

/* ----------------------------------------- DESC refinement tail (next row f-3) -- */
/* Reweighted Lie-algebraic averaging, Algorithms/DESC.m:265-313 with Utils/Weighted_LAA.m,
 * Build_Amatrix.m, R2Q.m, q2R.m.  s_vec: m (the PGD output), R_init: n*9 (3x3xn column-major, the
 * GCW output, DESC.m:263), R_out: n*9.  stop_threshold <= 0 -> 1e-3, max_iters <= 0 -> 100
 * (DESC.m:272).  MATLAB's sparse-QR least squares is replaced by f64 PCG on the normal equations; the
 * weights span 1e-4..1e4 (DESC.m:279-282), squared in the normal equations, so a solve may stop at its
 * iteration cap: info->cg_unconverged / cg_residual say so, and a warning is printed to stderr. */
typedef struct desc_refine_info {
    int32_t iters;            /* refinement steps executed                                     */
    int32_t cg_iters;         /* conjugate-gradient steps in total                             */
    int32_t verbose;          /* in: print the reference's "Iter %d: ||dR||= %f" lines          */
    int32_t cg_unconverged;   /* refinement steps whose PCG solve stopped at its iteration cap
                                 before reaching |r| <= 1e-13 |b| (0 = every solve converged)   */
    double  score;            /* last mean rotation update (Weighted_LAA.m:40)                 */
    double  ms_total;
    double  cg_residual;      /* largest relative residual |r|/|b| left by any of the solves   */
} desc_refine_info;
int desc_refine_run(const desc_problem* prob, const double* s_vec, const double* R_init, double stop_threshold,
                    int32_t max_iters, int32_t device, double* R_out, desc_refine_info* info);
int desc_refine_run_dev(const desc_device_problem* dp, const double* s_vec, const double* R_init, double stop_threshold,
                        int32_t max_iters, double* R_out, desc_refine_info* info);

/* One-shot: what the MEX shim calls.  Builds the structure (p->build_where),
 * uploads, runs, downloads, frees. */
int desc_pgd_solve(const desc_problem* prob, const desc_params* p, desc_result* r);

/* Host-side marshalling of the reference's argument formats (no device code; a binding that already holds 0-based int32 endpoints and
 * MATLAB's own 3 x 3 x m memory -- the MEX shim -- needs neither).
 * desc_marshal_edges: `Ind` as the caller of DESC_PGD.m:14 holds it -- m x 2 node ids, 1-based, Ind(:,1) < Ind(:,2), as doubles (MATLAB),
 * int64 or int32 -- read through element strides (row_stride, col_stride: MATLAB's column-major m x 2 = (1, m), NumPy's row-major = (2, 1))
 * into the ABI's 0-based int32 endpoints in ONE threaded pass.  *n_out = max(Ind(:)) (DESC_PGD.m:21); *sorted_out = 1 if the rows are
 * strictly ascending by (i, j) (what DESC_PGD.m:5 requires and desc_problem expects), 0 if the caller still has to sort (and to look for
 * duplicates).  DESC_ERR_INVALID: a value that is not an integer, an id < 1, or a row with Ind(:,1) >= Ind(:,2) (first offending row in the text).
 * desc_marshal_rij: out[9 l + r + 3 c] = R[r stride_r + c stride_c + perm[l] stride_l] -- any strided 3 x 3 x m array of doubles (NumPy's
 * C order: strides (3m, m, 1); MATLAB's order (1, 3, 9) is the ABI's own and needs no call) into desc_problem.rij, edges permuted by
 * `perm` (sorted position -> caller's row; NULL: identity). */
#define DESC_DTYPE_F64 0
#define DESC_DTYPE_I64 1
#define DESC_DTYPE_I32 2
int desc_marshal_edges(const void* ind, int32_t dtype, int64_t m, int64_t row_stride, int64_t col_stride,
                       int32_t* ind_i, int32_t* ind_j, int64_t* n_out, int32_t* sorted_out);
int desc_marshal_rij(const double* R, int64_t m, int64_t stride_r, int64_t stride_c, int64_t stride_l, const int64_t* perm, double* out);

/* The library parks the device blocks of destroyed handles / structures / problems for reuse by the next call (up to
 * DESC_CACHE_MB megabytes per process, default 8192: hipFree + hipMalloc of the gigabyte-sized per-cycle arrays cost 10-20 ms
 * per solve), and likewise the large host-side index vectors of its setup (up to DESC_HOST_CACHE_MB, default 1024: first-touch
 * page faults and munmap of ~200 MB cost 40 ms per solve at n = 5000).  desc_trim_memory returns everything parked to the
 * driver / the C++ runtime; result: device bytes released. */
int64_t desc_trim_memory(void);

/* Binding utilities: synchronous copies between host memory and device memory of the library's own HIP runtime
 * (a binding that implements the collectives itself, e.g. staged through host memory, must not load a second
 * runtime), after draining every stream of `device`. */
int desc_device_synchronize(int32_t device);
int desc_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes);
int desc_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes);

/* Test hook, host only (runs without a GPU): plans the band sweep of `rank` of `world` for `grid` workgroups and checks the plan's
 * invariants.  stats: 8 values out (bands, pieces, largest band's row entries, max / min cycles per workgroup, j-block-major?,
 * first / end segment of the rank). */
int desc_debug_band_plan(const desc_problem* prob, const desc_structure* s, int32_t world, int32_t rank, int32_t grid, int64_t* stats);

/* Diagnostics hook (tools/wg_clock.py): {start, end} (constant 100 MHz clock) of every workgroup of the last band sweep of a handle created
 * with DESC_DEBUG_WGCLOCK=1; returns how many workgroups were written to out[2 * cap] (0: not recorded).  When all G workgroups fit and
 * 2 * cap >= 3 * G, out[2 * G + w] = shader-clock cycles workgroup w ran for (cycles / duration = the clock frequency of the launch). */
int desc_debug_wg_clock(desc_pgd* h, uint64_t* out, int32_t cap);
/* ... and what the piece scheduler gave each of them: out[4 * w + {0,1,2,3}] = cycles, segments, pieces, CSR entries of the band rows loaded. */
int desc_debug_wg_plan(desc_pgd* h, int64_t* out, int32_t cap);

/* Test hooks (tests/test_gpu_sharded.py).  desc_debug_last_sweep: name and template arguments of the sweep kernel the handle launched last
 * ("k_sweep_band<16,4,0,512,XT>": the sharded instance), "" before the first sweep.  desc_debug_shard_layout: the exchange layout of a
 * (possibly sharded) handle -- xpos / spos: 2m entries each (CSR slot -> place of its mirror sum, DESC_PGD.m:189-190, in the reduce-scatter
 * send buffer / place of its edge's S in the gathered slices); xt: {ta, tb} per owned segment (device order), slot_ab: the CSR slots of
 * the same segments' edges; both 2 * (seg_hi - seg_lo) entries. */
const char* desc_debug_last_sweep(const desc_pgd* h);
int desc_debug_shard_layout(desc_pgd* h, int32_t* xpos, int32_t* spos, int32_t* xt, int32_t* slot_ab);

/* Measurement hook (tools/next_rows_bench.py; SURVEY.md 8d "document the MFMA measurement rather than assume"): the 3x3-block SpMM of the
 * connection matrix (Spectral.m:27-37) with unit weights, `reps` products in its vector-FMA form and in a v_mfma_f64_4x4x4 form on the
 * same operand.  out[4]: ms per product (vector FMA), ms per product (MFMA; -1: operand layout not identified), max |difference|, layout code. */
int desc_debug_spmm_variants(const desc_device_problem* dp, int32_t reps, double* out);

/* Test hook: sums `in` over aligned groups of G = 16/32/64 lanes with the kernels'
 * DPP / permlane-swap reduction; every element of a group receives the group total. */
int desc_selftest_group_sum(const double* in, double* out, int32_t count, int32_t G, int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* DESC_AMD_H */
