/* desc_pgd_mex.c -- thin MEX shim over the C ABI of libdesc_amd.so (include/desc_amd.h).
 *
 *   [S_sorted, info] = desc_pgd_mex(Ind0, RijMat, opt, adam_m, adam_v)
 *     Ind0    m x 2 int32, 0-based, rows sorted by (i,j)
 *     RijMat  3 x 3 x m double  -- passed through untouched: MATLAB's column-major layout of a
 *             3x3xm array IS the library's m x 9 layout (element (r,c,l) at 9*l + r + 3*c)
 *     opt     struct: iters, step_kind, lr, beta1, beta2, decay_interval, hybrid_strategy, t0,
 *             seed, device, verbose (1: the reference's per-iteration line, printed while the loop runs),
 *             make_plots (1: DESC_PGD.m:235-239 -- needs ErrVec, 1 x m in the sorted edge order)
 *     adam_m/adam_v   [] or 1 x m_cycle (HybridGradient.m_t / v_t carried between calls)
 *   info: iters_run, t_end, obj_vals, avg_change, adam_m, adam_v, ms_structure, ms_pgd, ms_total; with make_plots also
 *         svec_errors (1 x iters_run) and R_est_all (3 x 3 x n x iters_run: GCW(S_vec) after every iteration)
 *
 * Build (on a machine with MATLAB + ROCm; NOT possible in the build container: no mex.h):
 *   mex -I../include desc_pgd_mex.c -L../desc_amd -ldesc_amd
 * This file only marshals; every numerical statement lives behind desc_pgd_solve().  The same
 * entry point is exercised from Python/ctypes by tests/ (desc_amd/_lib.py).
 */
#include <string.h>

#include "mex.h"
#include "desc_amd.h"

/* DESC_PGD.m:241, streamed by the library while the loop runs */
static void mex_progress(void* user, int32_t it, double avg, double obj) {
    (void)user;
    mexPrintf("iter %d: average change in S_vec %f, objective value: %f\n", (int)it, avg, obj);
}

static double field_or(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxGetField(s, 0, name);
    return (f && !mxIsEmpty(f)) ? mxGetScalar(f) : dflt;
}

/* the library parks device and host blocks between calls (DESC_CACHE_MB / DESC_HOST_CACHE_MB): give them back when MATLAB
 * clears the MEX file or exits */
static void release_parked_blocks(void) { (void)desc_trim_memory(); }
static int at_exit_registered = 0;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (!at_exit_registered) { mexAtExit(release_parked_blocks); at_exit_registered = 1; }
    if (nrhs < 3) mexErrMsgIdAndTxt("desc_amd:nargin", "usage: desc_pgd_mex(Ind0, RijMat, opt, adam_m, adam_v)");
    if (!mxIsInt32(prhs[0]) || mxGetN(prhs[0]) != 2) mexErrMsgIdAndTxt("desc_amd:Ind", "Ind0 must be m x 2 int32");
    const mwSize m = mxGetM(prhs[0]);
    const mwSize* dims = mxGetDimensions(prhs[1]);
    const mwSize nd = mxGetNumberOfDimensions(prhs[1]);
    if (!mxIsDouble(prhs[1]) || mxIsComplex(prhs[1]) || dims[0] != 3 || dims[1] != 3 || (m > 1 ? (nd != 3 || dims[2] != m) : 0))
        mexErrMsgIdAndTxt("desc_amd:RijMat", "RijMat must be a real double 3 x 3 x m array");
    if (!mxIsStruct(prhs[2])) mexErrMsgIdAndTxt("desc_amd:opt", "opt must be a struct");

    const int32_t* ind = (const int32_t*)mxGetData(prhs[0]);       /* column-major: [i(0..m-1), j(0..m-1)] */
    desc_problem prob;
    prob.m = (int64_t)m;
    prob.ind_i = ind;
    prob.ind_j = ind + m;
    prob.rij = mxGetPr(prhs[1]);
    int32_t nmax = -1;
    for (mwSize e = 0; e < m; ++e) if (ind[m + e] > nmax) nmax = ind[m + e];
    prob.n = (int64_t)nmax + 1;                                      /* n = max(Ind(:)), DESC_PGD.m:21 */

    desc_params p;
    desc_params_default(&p);
    p.iters = (int32_t)field_or(prhs[2], "iters", 100);
    p.step_kind = (int32_t)field_or(prhs[2], "step_kind", 0);
    p.lr = field_or(prhs[2], "lr", 0.01);
    p.beta1 = field_or(prhs[2], "beta1", 0.9);
    p.beta2 = field_or(prhs[2], "beta2", 0.999);
    p.decay_interval = field_or(prhs[2], "decay_interval", 25);
    p.hybrid_strategy = (int32_t)field_or(prhs[2], "hybrid_strategy", 0);
    p.t0 = (int32_t)field_or(prhs[2], "t0", 0);
    p.seed = (uint64_t)field_or(prhs[2], "seed", 0);
    p.device = (int32_t)field_or(prhs[2], "device", 0);
    p.verbose = 0;
    if (field_or(prhs[2], "verbose", 0) != 0) p.progress = mex_progress;    /* per-iteration lines through mexPrintf, as they happen */

    const int make_plots = field_or(prhs[2], "make_plots", 0) != 0;
    /* The common call -- no per-iteration GCW, no moment vectors to carry -- is ONE library call (desc_pgd_solve): the rotations go up
     * while the structure is built.  Otherwise the structure comes first: the per-cycle vectors (HybridGradient.m_t / v_t) are sized by it. */
    const int one_call = !make_plots && !(p.step_kind == DESC_STEP_HYBRID && p.hybrid_strategy == 0);
    desc_structure* st = NULL;
    int rc = DESC_OK;
    mwSize mc = 0;
    double ms_structure = 0.0;
    if (!one_call) {
        rc = desc_structure_build(&prob, p.n_sample_min, p.seed, p.build_where, p.device, &st);
        if (rc == DESC_ERR_TOO_LARGE && p.build_where == DESC_BUILD_DEVICE)   /* beyond the device builder's staging budget */
            rc = desc_structure_build(&prob, p.n_sample_min, p.seed, DESC_BUILD_HOST, p.device, &st);
        if (rc != DESC_OK) mexErrMsgIdAndTxt("desc_amd:structure", "%s", desc_last_error());
        desc_structure_info v;                     /* O(1): the structure stays in HBM */
        if (desc_structure_sizes(st, &v) != DESC_OK) {
            desc_structure_free(st);
            mexErrMsgIdAndTxt("desc_amd:structure", "%s", desc_last_error());
        }
        mc = (mwSize)v.m_cycle;
        ms_structure = v.ms_build;
    }

    plhs[0] = mxCreateDoubleMatrix(1, m, mxREAL);
    mxArray* obj = mxCreateDoubleMatrix(1, p.iters > 0 ? p.iters : 1, mxREAL);
    mxArray* avg = mxCreateDoubleMatrix(1, p.iters > 0 ? p.iters : 1, mxREAL);
    mxArray* am = mxCreateDoubleMatrix(1, mc, mxREAL);
    mxArray* av = mxCreateDoubleMatrix(1, mc, mxREAL);
    if (nrhs >= 5 && !mxIsEmpty(prhs[3]) && !mxIsEmpty(prhs[4])) {
        if (mxGetNumberOfElements(prhs[3]) != mc || mxGetNumberOfElements(prhs[4]) != mc) {
            desc_structure_free(st);
            mexErrMsgIdAndTxt("desc_amd:adam", "HybridGradient state has a different length than this problem's cycle vector");
        }
        memcpy(mxGetPr(am), mxGetPr(prhs[3]), sizeof(double) * mc);
        memcpy(mxGetPr(av), mxGetPr(prhs[4]), sizeof(double) * mc);
    }
    desc_result r;
    memset(&r, 0, sizeof r);
    r.s_vec = mxGetPr(plhs[0]);
    r.obj_trace = mxGetPr(obj);
    r.avg_change_trace = mxGetPr(avg);
    if (p.step_kind == DESC_STEP_HYBRID && p.hybrid_strategy == 0) { r.adam_m = mxGetPr(am); r.adam_v = mxGetPr(av); }

    mxArray *se = NULL, *rall = NULL;
    desc_pgd* h = NULL;
    if (one_call) {
        rc = desc_pgd_solve(&prob, &p, &r);
        ms_structure = r.ms_structure;
    } else if (!make_plots) {
        rc = desc_pgd_create(&prob, st, p.device, &h);
        desc_structure_free(st);
        if (rc == DESC_OK) rc = desc_pgd_run(h, &p, &r);
    } else {                                   /* DESC_PGD.m:235-239: svec_errors and GCW(S_vec) after every iteration */
        const mxArray* ev = mxGetField(prhs[2], 0, "ErrVec");
        if (!ev || !mxIsDouble(ev) || mxGetNumberOfElements(ev) != m) {
            desc_structure_free(st);
            mexErrMsgIdAndTxt("desc_amd:make_plots", "opt.make_plots needs opt.ErrVec with one entry per edge");
        }
        const mwSize it = p.iters > 0 ? p.iters : 1;
        mwSize rd[4]; rd[0] = 3; rd[1] = 3; rd[2] = (mwSize)prob.n; rd[3] = it;
        se = mxCreateDoubleMatrix(1, it, mxREAL);
        rall = mxCreateNumericArray(4, rd, mxDOUBLE_CLASS, mxREAL);
        desc_device_problem* dp = NULL;
        rc = desc_problem_upload(&prob, p.device, &dp);
        if (rc == DESC_OK) rc = desc_pgd_create_dev(dp, st, 0, 1, &h);
        desc_structure_free(st);
        if (rc == DESC_OK) rc = desc_pgd_run_traced(h, dp, &p, mxGetPr(ev), 1e-13, 500, mxGetPr(se), mxGetPr(rall), &r);
        if (dp) desc_problem_free(dp);
        if (rc == DESC_OK && (mwSize)r.iters_run < it) {          /* early stop: trim to the iterations that ran */
            mxSetN(se, (mwSize)r.iters_run);
            rd[3] = (mwSize)r.iters_run;
            mxSetDimensions(rall, rd, 4);
        }
    }
    if (h) desc_pgd_destroy(h);
    if (rc != DESC_OK) mexErrMsgIdAndTxt("desc_amd:run", "%s", desc_last_error());

    if (nlhs > 1) {
        const char* names[] = {"iters_run", "t_end", "obj_vals", "avg_change", "adam_m", "adam_v", "ms_structure", "ms_pgd", "ms_total",
                               "svec_errors", "R_est_all"};
        plhs[1] = mxCreateStructMatrix(1, 1, make_plots ? 11 : 9, names);
        if (make_plots) { mxSetField(plhs[1], 0, "svec_errors", se); mxSetField(plhs[1], 0, "R_est_all", rall); }
        mxSetField(plhs[1], 0, "iters_run", mxCreateDoubleScalar(r.iters_run));
        mxSetField(plhs[1], 0, "t_end", mxCreateDoubleScalar(r.t_end));
        mxSetField(plhs[1], 0, "obj_vals", obj);
        mxSetField(plhs[1], 0, "avg_change", avg);
        mxSetField(plhs[1], 0, "adam_m", am);
        mxSetField(plhs[1], 0, "adam_v", av);
        mxSetField(plhs[1], 0, "ms_structure", mxCreateDoubleScalar(ms_structure));
        mxSetField(plhs[1], 0, "ms_pgd", mxCreateDoubleScalar(r.ms_pgd));
        mxSetField(plhs[1], 0, "ms_total", mxCreateDoubleScalar(r.ms_total));
    }
}
